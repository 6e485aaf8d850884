"""Host-side checks of the two numerical building blocks the HBV-Edu kernel
uses instead of OCML's general pow / hipcc's division expansion.  Both headers
are portable; small g++ harnesses (tests/native/) exercise the very source the
kernels compile.

  * fastmath.h : every fast form's worst error vs 80-bit libm inside its stated bound
  * invdiv.h  : bit-identical to `a / b` on 2e7 random + adversarial pairs
"""

import os
import re
import subprocess

from .conftest import REPO

NATIVE = os.path.join(REPO, "tests", "native")


def _build_and_run(src, tmp_path, *args):
    exe = os.path.join(str(tmp_path), os.path.splitext(src)[0])
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", exe,
                           os.path.join(NATIVE, src), "-lm"])
    return subprocess.run([exe, *args], capture_output=True, text=True,
                          check=True).stdout


def test_fastpow_accuracy(tmp_path):
    out = _build_and_run("fastmath_harness.cpp", tmp_path, "400000")
    vals = dict(re.findall(r"^(\w+) ([0-9.]+)", out, flags=re.M))
    # HBV-Edu's default since round 5: the power from the soil alone
    # (fastpow_soil) against powl of the exact quotient -- a sane run's box,
    # the whole guard box (FC 1e-3..1e6, soil within 2^9 of it, |Beta| <= 64)
    # and the stated bound
    # (6 + 3 |zz| + |y| (1 + 3 |log2 FC| + 3 |log2 soil|)) 2^-53
    assert float(vals["soil_worst_rel53_sane"]) < 250, out       # 2.8e-14
    assert float(vals["soil_worst_rel53_box"]) < 6000, out       # 6.7e-13
    assert int(vals["soil_worst_over_bound_x100"]) <= 100, out
    assert int(vals["soil_guard_rejected"]) == 0, out
    assert int(vals["soil_special_ok"]) == 1, out
    assert float(vals["worst_ulp_tanh_gr4j"]) < 3.0, out
    assert float(vals["worst_ulp_tanh_wide"]) < 3.0, out
    assert int(vals["tanh_special_ok"]) == 1, out
    # GR4J's default inside |a| <= 1: the [9/8] Pade pair
    assert float(vals["worst_ulp_tanh_rational"]) < 4.0, out
    assert int(vals["tanh_rational_special_ok"]) == 1, out
    assert float(vals["worst_ulp_fast_sqrt"]) < 0.75, out
    assert float(vals["worst_ulp_inv_fourth_root"]) < 2.0, out
    assert int(vals["r4_special_ok"]) == 1, out
    assert float(vals["worst_ulp_inv_fourth_root3"]) < 1.5, out
    assert int(vals["r4_3_exact_ok"]) == 1, out
    assert float(vals["worst_ulp_inv_fourth_root_poly"]) < 0.55, out
    assert int(vals["r4_poly_ends_ok"]) == 1, out
    assert float(vals["worst_ulp_fast_div"]) < 1.0, out
    assert int(vals["fast_div_exact_ok"]) == 1, out


def test_invariant_division_is_bit_exact(tmp_path):
    out = _build_and_run("invdiv_harness.cpp", tmp_path, "20000000")
    vals = dict(re.findall(r"^(\w+) ([0-9]+)", out, flags=re.M))
    assert int(vals["mismatches"]) == 0, out
    assert int(vals["checked"]) > 15_000_000, out
    # numerators below the strict vote's 2^-900 (GR4J's own quotients admit
    # them, gr4j_core.h gr4j_num_ok): the 3-FMA form is faithful there --
    # never more than one ulp from the IEEE quotient
    assert int(vals["tiny_checked"]) > 4_000_000, out
    assert int(vals["tiny_worst_ulps"]) <= 1, out
    # the faithful form a * RN(1/b) of HBV-Edu's and GR4J's own quotients: any
    # numerator, within 1.5 ulp; zeros, infinities and NaN as the division
    assert int(vals["faithful_checked"]) > 9_000_000, out
    assert int(vals["faithful_worst_ulps_x100"]) <= 150, out
    assert int(vals["faithful_special_bad"]) == 0, out


def test_pow2_tables_header_matches_its_generator(tmp_path):
    """rrmpg_amd/csrc/pow2_tables.h (fastpow_soil's tables) is what
    csrc/tools/gen_pow2_tables.py writes; 512 logarithm entries over [1/2, 1)
    whose products with their subinterval stay within 2^-9.99 of 1, 256
    entries 2^(j/256)."""
    import importlib.util
    import pytest
    pytest.importorskip("mpmath")
    gen = os.path.join(REPO, "rrmpg_amd", "csrc", "tools",
                       "gen_pow2_tables.py")
    committed = os.path.join(REPO, "rrmpg_amd", "csrc", "pow2_tables.h")
    with open(committed) as fp:
        want = fp.read()
    spec = importlib.util.spec_from_file_location("gen_pow2_tables", gen)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = str(tmp_path / "pow2_tables.h")
    mod.main(fresh)
    with open(fresh) as fp:
        assert fp.read() == want
    rows = re.findall(r"^    \{(\S+), (\S+)\}, ", want, flags=re.M)
    assert len(rows) == 512
    import math
    for i, (invc, lnc) in enumerate(rows):
        invc, lnc = float.fromhex(invc), float.fromhex(lnc)
        lo, hi = 0.5 + i / 1024, 0.5 + (i + 1) / 1024
        assert abs(lo * invc - 1) <= 2 ** -9.99
        assert abs(hi * invc - 1) <= 2 ** -9.99
        assert abs(lnc + math.log(invc)) < 2e-16
    exps = re.findall(r"0x1\.[0-9a-f]+p\+0", want.split(
        "FP_SOIL_EXP_TABLE_INIT")[1])
    assert len(exps) == 256
    assert float.fromhex(exps[128]) == 2 ** 0.5
    assert "#define FP_SOIL_LOG_A0 -0x1.0000000000000p-1" in want
