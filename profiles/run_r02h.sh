#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r02h
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest exit $?" >> $O/pytest.log
timeout 900 python profiles/host_path_timing.py > $O/host_path.log 2>&1
tail -8 $O/pytest.log; cat $O/host_path.log
