"""Identity of a model's kernel sources: what profiles/make_traffic.py stamps
the committed counter results (profiles/traffic.json) with and bench.py
checks before quoting them, so that a kernel change without a new counter
collection shows as "stale" instead of silently keeping the old numbers."""

import glob
import hashlib
import os
import re

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     "csrc")
_KERNEL_FILE = {"hbvedu": "hbvedu.hip", "abc": "abc.hip", "gr4j": "gr4j.hip",
                "cemaneige": "cemaneige.hip", "cemaneigegr4j": "cemaneige.hip",
                "cemaneigehystgr4j": "snownext_hyst.hip",
                "cemaneigegr4jice": "snownext.hip",
                "cemaneigehystgr4jice": "snownext.hip"}


_COMMENTS = re.compile(r"//[^\n]*|/\*.*?\*/", re.S)


def _code_of(text):
    """The source without its comments and with runs of white space
    collapsed: an edit of a comment is not a change of the kernel."""
    return " ".join(_COMMENTS.sub(" ", text).split())


def kernel_source_id(model):
    """sha256 (16 hex digits) over the CODE (comments and white space
    dropped) of the model's kernel file and of every header of
    rrmpg_amd/csrc, in name order."""
    files = sorted(glob.glob(os.path.join(_CSRC, "*.h")))
    files.append(os.path.join(_CSRC, _KERNEL_FILE[model]))
    h = hashlib.sha256()
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "r", encoding="utf-8", errors="replace") as fh:
            h.update(_code_of(fh.read()).encode())
    return h.hexdigest()[:16]
