"""GPU fuzz tests: parameters far outside every model's bounds (zeros,
negatives, denormals, huge values, infinities, NaN) exercise the guarded
fallbacks of the fast paths (fastpow domain, invariant-divisor division
ranges, tanh clamp, padded unit hydrographs) -- results must still match the
oracle: same NaN / inf pattern everywhere; finite values within 1e-10
relative for the in-bounds sets (every even column) and within 1e-6 for the
wild ones, whose dynamics are ill-conditioned (e.g. Beta = 100 amplifies a
1-ulp difference of the power a hundredfold per step)."""

import os

import numpy as np
import pytest

from .conftest import golden, snow_same

#: RR_FUZZ_SEED=<k> shifts every generator seed (soak runs: the committed
#: runs use 0)
SEED = int(os.environ.get("RR_FUZZ_SEED", "0"))

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def _jitter(rng, x):
    """x with every non-zero entry moved one ulp up or down at random: a
    perturbation injected on every day, as the one-ulp difference between two
    correct `pow`s / `tanh`s is."""
    x = np.asarray(x, dtype=np.float64)
    up = rng.random(x.shape) < 0.5
    y = np.where(up, np.nextafter(x, np.inf), np.nextafter(x, -np.inf))
    return np.where(x == 0, x, y)


def _lost_days(b, runs):
    """First day from which a WILD set has lost its digits, per set (T =
    never): the day one of the oracle's own perturbed `runs` (the one-ulp
    probes of _same, and chaos probes perturbed by 1e-9 relative -- more than
    a rounding can absorb) has drifted more than 1e-3 from the unperturbed
    series `b`, relative to what the series has reached BY THAT DAY (a
    blow-up at the end must not hide a chaotic phase of millimetres).
    K_0 = 7.5 with a threshold in the loop is chaotic: one ulp decides which
    days the store spills, and a week later the runs are whole millimetres
    -- or an overflow -- apart.  Such a set is compared up to that day (the
    only rule that ever stops following the oracle, and it is about the
    ORACLE's conditioning: there is no rule for overflow any more -- sets
    that are not civil run the reference's own sequence on the GPU,
    csrc/hbvedu.hip hbv_civil_lane, csrc/gr4j_reference.h); in-bounds sets
    are never excused."""
    b = np.asarray(b)
    T, n = b.shape[0], b.shape[-1]
    with np.errstate(all="ignore"):
        bb = b.reshape(T, -1, n)
        fb = np.isfinite(bb)
        top = np.maximum.accumulate(
            np.where(fb, np.abs(bb), 0.0).max(axis=1), axis=0)
        scale = np.maximum(np.abs(bb),
                           1e-2 * np.maximum(top, 1e-9)[:, None, :])
        dev = np.zeros((T, n))
        for x in runs:
            xx = np.asarray(x).reshape(T, -1, n)
            fx = np.isfinite(xx)
            d = np.where(fb & fx, np.abs(xx - bb) / scale,
                         np.where(fb != fx, np.inf, 0.0))
            dev = np.maximum(dev, d.max(axis=1))
        gone = np.maximum.accumulate(dev, axis=0) > 1e-3
    lost = np.where(gone.any(axis=0), gone.argmax(axis=0), T)
    lost[::2] = T
    return lost


def _same(a, b, what, b_perturbed=None, horizon=None, chaos=None):
    """horizon: _lost_days' days (sets are only compared before theirs).
    b_perturbed: the oracle's own result(s) for slightly perturbed inputs
    (one array or a list): initial states moved by one ulp, the forcing
    jittered by one ulp per day (_jitter), the parameters moved by one ulp (a
    routing store of x3 = 0.5 mm against inflows of millimetres loses three
    digits a day to the cancellation in r + uh1 + exchange -- the reference's
    own fp64 arithmetic is then 1e-13 off the exact value per day, measured
    against a 40-digit evaluation -- and no jitter of the forcing shows that).  Where that alone moves a set's
    series by `amp` (relative), the set is ill-conditioned -- a wild Beta or
    recession constant makes the dynamics unstable (K_0 = 7.5: every day above
    the threshold multiplies a difference by -6.5) -- and a one-ulp difference
    between two correct `pow`s, injected every day, grows the same way: such a
    set is compared at 1000 * amp (a wild one at 1e5 * amp) instead of the flat
    tolerance (its NaN / inf pattern still has to match exactly).
    chaos: further oracle runs for _lost_days only."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    if b_perturbed is not None:
        pr = (b_perturbed if isinstance(b_perturbed, (list, tuple))
              else [b_perturbed])
        lost = _lost_days(b, list(pr) + list(chaos or []))
        assert (lost[1::2] < b.shape[0]).mean() < 0.5, \
            what + ": too many sets excused"
        horizon = lost if horizon is None else np.minimum(horizon, lost)
    if horizon is not None:
        days = np.arange(b.shape[0]).reshape((-1,) + (1,) * (b.ndim - 1))
        dead = days >= np.asarray(horizon)         # last axis = sets
        a, b = np.where(dead, 0.0, a), np.where(dead, 0.0, b)
        if b_perturbed is not None:
            pr = (b_perturbed if isinstance(b_perturbed, (list, tuple))
                  else [b_perturbed])
            b_perturbed = [np.where(dead, 0.0, np.asarray(x)) for x in pr]
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), what + ": NaN pattern"
    inf = np.isinf(b)
    assert np.array_equal(np.isinf(a), inf), what + ": inf pattern"
    assert np.array_equal(a[inf], b[inf]), what + ": inf sign"
    ok = ~(nan_b | inf)
    # |a - b| <= rtol * |b| + atol, atol scaled by the largest finite value of
    # the set's series (a store that empties leaves a cancellation residue of
    # ~1 ulp of what it held the day before).  Last axis = parameter sets;
    # even sets are in bounds (rtol 1e-10), odd ones wild (rtol 1e-6).
    with np.errstate(all="ignore"):
        fin = np.where(ok, np.abs(b), 0.0)
        colmax = fin.reshape(-1, fin.shape[-1]).max(axis=0)
        diff = np.where(ok, np.abs(a - b), 0.0)
        rtol = np.where(np.arange(b.shape[-1]) % 2 == 0, RTOL, 1e-6)
        if b_perturbed is not None:
            probes = (b_perturbed if isinstance(b_perturbed, (list, tuple))
                      else [b_perturbed])
            for b2 in probes:
                b2 = np.asarray(b2)
                both = ok & np.isfinite(b2)
                d2 = np.where(both, np.abs(b2 - b), 0.0)
                scale = np.maximum(fin, 1e-2 * np.maximum(colmax, 1e-9))
                amp = (d2 / scale).reshape(-1, b.shape[-1]).max(axis=0)
                # (a probe that overflows or changes a degenerate set's
                # regime: "do not compare the values of this set")
                amp = np.nan_to_num(amp, nan=1.0, posinf=1.0)
                # (wild sets: a few probes sample the sensitivity of a
                # threshold-ridden system -- a routing store of 0.5 mm that
                # empties every other day -- poorly: two more decades)
                mult = np.where(np.arange(b.shape[-1]) % 2 == 0, 1e3, 1e5)
                rtol = np.maximum(rtol, np.minimum(mult * amp, 1e3))
        # (wild sets: nothing below 1e-9 mm counts, the stated absolute
        # floor -- an exact zero against what a differently rounded
        # cancellation of stores of 1e5 mm leaves)
        floor = np.where(np.arange(b.shape[-1]) % 2 == 0, 0.0, 1e-9)
        # the probes must not loosen the in-bounds majority
        even = rtol[::2]
        assert (even > 100 * RTOL).mean() < 0.2, what + ": probes too loose"
        tol = rtol * fin + 1e-2 * rtol * np.maximum(colmax, 1e-9) + floor
        excess = np.where(diff <= tol, 0.0, diff - tol)
    assert excess.max(initial=0.0) <= 0, "%s: excess %g at %s" % (
        what, excess.max(), np.unravel_index(np.argmax(excess), excess.shape))


def _records(cls, flat):
    p = np.zeros(flat.shape[0], dtype=cls._dtype)
    for k, name in enumerate(cls._param_list):
        p[name] = flat[:, k]
    return p


WILD = np.array([0.0, -0.0, 1.0, -1.0, 5e-324, 1e-310, 2.3e-308, 1e-200,
                 1e200, 1e308, -1e308, np.inf, -np.inf, np.nan, 0.5, 2.0,
                 -0.25, 100.0, 1e-5, 7.5])


def _wild_params(rng, lo, hi, n):
    """n sets inside the bounds, then every parameter of every 2nd set
    replaced at random by a wild value."""
    k = lo.size
    flat = lo + (hi - lo) * rng.random((n, k))
    mask = rng.random((n, k)) < 0.25
    mask[::2] = False
    flat[mask] = rng.choice(WILD, mask.sum())
    return flat


@pytest.fixture(scope="module")
def models():
    from rrmpg_amd import _lib
    _lib.load()
    _lib.require_gpu()
    import rrmpg_amd.models as m
    return m


def test_hbvedu_fuzz(models, oracle, hbv_variant):
    g = golden("syn_hbvedu")
    rng = np.random.default_rng(100 + 1000 * SEED)
    lo = np.array([-1, 3, 100, 1, .01, 90, .05, .01, .01, .01, 2.])
    hi = np.array([1, 7, 200, 7, .07, 180, .2, .1, .05, .05, 5.])
    flat = _wild_params(rng, lo, hi, 640)
    # A negative finite Beta makes the soil update singular at soil -> 0
    # (prec_eff = lw * (FC/soil)**|Beta| throws the store through zero and
    # back): two correct one-ulp-apart powers end up orders of magnitude apart
    # within days, which no flat tolerance can separate from a real error.
    # Those sets keep their wild Beta's magnitude here; negative finite Beta
    # has its own test with a day-by-day conditioning bound,
    # test_hbvedu_negative_beta_within_its_conditioning.
    neg = np.isfinite(flat[:, 3]) & (flat[:, 3] < 0)
    flat[neg, 3] = -flat[neg, 3]
    t = 400
    with np.errstate(all="ignore"):
        ref = oracle.simulate_hbvedu(g["temp"][:t], g["prec"][:t],
                                     g["month"][:t] - 1, g["PE_m"], g["T_m"],
                                     (0., 100., 3., 10.), flat,
                                     return_storage=True, nthreads=8)
        # conditioning probe: the same sets from initial states one ulp up
        # (moving the parameters instead would change a wild set's regime:
        # Beta = -1 is fine for a negative base, -1 + 1 ulp is NaN)
        ref2 = oracle.simulate_hbvedu(g["temp"][:t], g["prec"][:t],
                                      g["month"][:t] - 1, g["PE_m"], g["T_m"],
                                      tuple(np.nextafter(v, np.inf) for v in
                                            (0., 100., 3., 10.)), flat,
                                      return_storage=True, nthreads=8)
        # ... and with the precipitation jittered by one ulp on every wet day
        # (an instability that sets in late has forgotten the initial states)
        ref3 = oracle.simulate_hbvedu(g["temp"][:t],
                                      _jitter(rng, g["prec"][:t]),
                                      g["month"][:t] - 1, g["PE_m"], g["T_m"],
                                      (0., 100., 3., 10.), flat,
                                      return_storage=True, nthreads=8)
        # ... and with the monthly tables moved by one ulp: the evaporation's
        # own rounding (a wild C = 100 makes a soil below the wilting point
        # grow by half a day's worth every day; neither probe above reaches
        # the soil of a set whose Beta = 0 decouples it from everything else)
        ref4 = oracle.simulate_hbvedu(g["temp"][:t], g["prec"][:t],
                                      g["month"][:t] - 1,
                                      _jitter(rng, g["PE_m"]),
                                      _jitter(rng, g["T_m"]),
                                      (0., 100., 3., 10.), flat,
                                      return_storage=True, nthreads=8)
    out = models.HBVEdu().simulate(g["temp"][:t], g["prec"][:t],
                                   g["month"][:t], g["PE_m"], g["T_m"], 0.,
                                   100., 3., 10., return_storage=True,
                                   params=_records(models.HBVEdu, flat))
    with np.errstate(all="ignore"):
        # chaos probe (_same): every initial state larger by 1e-9
        ref5 = oracle.simulate_hbvedu(g["temp"][:t], g["prec"][:t],
                                      g["month"][:t] - 1, g["PE_m"], g["T_m"],
                                      tuple(v * (1 + 1e-9) for v in
                                            (0., 100., 3., 10.)), flat,
                                      return_storage=True, nthreads=8)
    # No overflow rule: a set that is not civil (csrc/hbvedu.hip
    # hbv_civil_lane: every wild value of WILD but a few harmless ones) is
    # computed with the reference's own sequence, so infinities and NaN
    # appear where and as the reference produces them -- compared day by
    # day over the whole series.
    # What remains is the oracle's OWN conditioning: a set whose soil has
    # gone chaotic (Beta = 100 around soil = FC, K_0 = 7.5 with a threshold
    # in the loop) is lost in every series from the day the oracle's own
    # perturbed runs are 1e-3 apart: the discharge remembers a spike of
    # effective precipitation that one trajectory saw and the other did not
    horizon = np.full(flat.shape[0], t)
    for k in (0, 2, 3, 4):
        horizon = np.minimum(horizon, _lost_days(
            ref[k], [ref2[k], ref3[k], ref4[k], ref5[k]]))
    assert (horizon[::2] == t).all() and (horizon[1::2] == t).mean() > 0.5
    for a, b, b2, b3, b4, b5, n in zip(out, ref, ref2, ref3, ref4, ref5,
                                       ["qsim", "snow", "soil", "s1", "s2"]):
        _same(a, b, "hbv " + n, [b2, b3, b4] if n != "snow" else None,
              horizon=None if n == "snow" else horizon,
              chaos=None if n == "snow" else [b5])
    assert np.isnan(ref[0]).any() and np.isfinite(ref[0]).any()
    # the probe must not loosen the well-conditioned majority
    with np.errstate(all="ignore"):
        q, q2 = ref[0], ref2[0]
        okq = np.isfinite(q) & np.isfinite(q2)
        amp = np.where(okq, np.abs(q2 - q) / np.maximum(np.abs(q), 1e-9), 0)
    assert (amp.max(axis=0)[::2] < 1e-12).all()      # in-bounds sets
    assert (amp.max(axis=0)[1::2] < 1e-9).mean() > 0.5   # most wild ones too


def test_hbvedu_civil_set_that_runs_away(models, oracle, hbv_variant):
    """Fuzz seed 684, set 443, kept as a case of its own: every parameter
    inside the civil box, but FC = 1 mm under an initial soil of 100 mm makes
    (soil / FC)**Beta 1e4 -- the soil overshoots, and within two months the
    stores are infinite.  There the fast forms are no longer the reference's
    statements (max(0, s1 - L) * K_0 with s1 = inf, K_0 = 0 is NaN; the
    folded form's hardware maximum gives 0), and the near-surface store read
    inf where the reference has NaN, a day late.  The reference kernel
    behind the fast one now takes every all-civil wave whose last output row
    (or sum of squares) is not finite and computes it again, each lane with
    the reference's own day from the first day that starts with a store that
    is not finite (csrc/hbvedu.hip): the NaN / inf pattern is the oracle's
    day by day, the set's wave-mates keep their bits, and the set itself has
    the same bits among civil wave-mates, among wild ones, and with the
    discharge as the only output."""
    g = golden("syn_hbvedu")
    rng = np.random.default_rng(684)
    lo = np.array([-1, 3, 100, 1, .01, 90, .05, .01, .01, .01, 2.])
    hi = np.array([1, 7, 200, 7, .07, 180, .2, .1, .05, .05, 5.])
    n, t = 200, 400
    flat = lo + (hi - lo) * rng.random((n, 11))
    wild = np.array([-0.6172643189812488, 1e-05, 1.0, 2.0,
                     0.013688696087420371, 105.2038075127825, 0.0,
                     0.07708904560274418, 0.036591165513947574,
                     0.040856502449478624, 4.866631712418977])
    where = [3, 64, 131, 199]
    runaway = flat.copy()
    runaway[where] = wild
    runaway[131, 2] = 0.5           # (another run-away: FC = 0.5 mm)
    args = (g["temp"][:t], g["prec"][:t], g["month"][:t], g["PE_m"],
            g["T_m"], 0., 100., 3., 10.)
    with np.errstate(all="ignore"):
        ref = oracle.simulate_hbvedu(args[0], args[1], args[2] - 1, args[3],
                                     args[4], (0., 100., 3., 10.), runaway,
                                     return_storage=True, nthreads=8)
    assert np.isnan(ref[0][:, where]).any() and np.isinf(ref[3][:, 3]).any()
    out = models.HBVEdu().simulate(*args, return_storage=True,
                                   params=_records(models.HBVEdu, runaway))
    tame = models.HBVEdu().simulate(*args, return_storage=True,
                                    params=_records(models.HBVEdu, flat))
    others = np.setdiff1d(np.arange(n), where)
    for a, b, c, name in zip(out, ref, tame,
                             ["qsim", "snow", "soil", "s1", "s2"]):
        for i in where:
            x, r = a[:, i], b[:, i]
            assert np.array_equal(np.isnan(x), np.isnan(r)), (name, i)
            assert np.array_equal(np.isinf(x), np.isinf(r)), (name, i)
            fin = np.isfinite(r)
            assert np.allclose(x[fin], r[fin], rtol=1e-9, atol=0), (name, i)
        # the wave-mates: the bits they have without the run-away sets
        assert np.array_equal(a[:, others], c[:, others]), name
    # ... in the company of wild sets (the reference kernel's waves from the
    # start), and with qsim alone
    company = runaway.copy()
    company[[0, 70, 140, 198], 5] = np.nan
    mixed = models.HBVEdu().simulate(*args, return_storage=True,
                                     params=_records(models.HBVEdu, company))
    for a, b, name in zip(out, mixed, ["qsim", "snow", "soil", "s1", "s2"]):
        assert np.array_equal(a[:, where], b[:, where], equal_nan=True), name
    q_only = models.HBVEdu().simulate(*args,
                                      params=_records(models.HBVEdu, runaway))
    assert np.array_equal(np.asarray(q_only), out[0], equal_nan=True)
    # ... and the sums of squares, alone (the headline mode: nothing but the
    # sums tells the reference kernel) and next to the discharge
    from rrmpg_amd.models import hbvedu as hbv_mod
    forcing = (args[0], args[1], (args[2] - 1).astype(np.int8), args[3],
               args[4])
    qobs = np.abs(np.asarray(tame[0][:, 7]))
    recs = _records(models.HBVEdu, runaway)
    _, sse_alone = hbv_mod._run(forcing, args[5:], recs, False, False, qobs)
    (q_too, *_), sse_with = hbv_mod._run(forcing, args[5:], recs, True, False,
                                         qobs)
    assert np.array_equal(sse_alone, sse_with, equal_nan=True)
    assert np.array_equal(q_too, out[0], equal_nan=True)
    with np.errstate(all="ignore"):
        want = ((ref[0] - qobs[:, None]) ** 2).sum(axis=0)
    assert np.array_equal(np.isnan(sse_alone), np.isnan(want))
    assert np.allclose(sse_alone[others], want[others], rtol=1e-9)


def test_hbvedu_sets_do_not_feel_their_wave_mates(models, oracle):
    """Which sequence a set gets -- the fast forms or the reference's own
    (csrc/hbvedu.hip hbv_civil_lane) -- depends on ITS parameters only: a
    civil set in a wave of wild ones (the reference kernel's wave) has the
    bits it has among civil sets (the fast kernels'), and a wild set is the
    oracle's day by day, NaN / inf pattern and all, wherever it sits."""
    g = golden("syn_hbvedu")
    rng = np.random.default_rng(107 + 1000 * SEED)
    lo = np.array([-1, 3, 100, 1, .01, 90, .05, .01, .01, .01, 2.])
    hi = np.array([1, 7, 200, 7, .07, 180, .2, .1, .05, .05, 5.])
    n, t = 256, 500
    civil = lo + (hi - lo) * rng.random((n, 11))
    mixed = civil.copy()
    wild = {1: (6, -0.5), 5: (2, 0.0), 64: (3, np.inf), 65: (1, -3.0),
            130: (9, 1e200), 200: (5, np.nan), 255: (10, -np.inf)}
    for row, (col, val) in wild.items():
        mixed[row, col] = val
    args = (g["temp"][:t], g["prec"][:t], g["month"][:t], g["PE_m"], g["T_m"])
    inits = (0., 0., 3., 10.)          # an empty soil store: outside the box
    sim = lambda flat: models.HBVEdu().simulate(
        *args, *inits, return_storage=True,
        params=_records(models.HBVEdu, flat))
    a, b = sim(civil), sim(mixed)
    keep = np.array([k for k in range(n) if k not in wild])
    for x, y in zip(a, b):
        assert np.array_equal(x[:, keep], y[:, keep])
    with np.errstate(all="ignore"):
        ref = oracle.simulate_hbvedu(args[0], args[1], args[2] - 1, args[3],
                                     args[4], inits, mixed,
                                     return_storage=True, nthreads=8)
    rows = np.array(sorted(wild))
    for x, r, name in zip(b, ref, ["qsim", "snow", "soil", "s1", "s2"]):
        x, r = x[:, rows], r[:, rows]
        assert np.array_equal(np.isnan(x), np.isnan(r)), name
        assert np.array_equal(np.isinf(x), np.isinf(r)), name
        fin = np.isfinite(r)
        assert np.array_equal(np.sign(x[~fin & ~np.isnan(r)]),
                              np.sign(r[~fin & ~np.isnan(r)])), name
        with np.errstate(all="ignore"):
            err = np.abs(x[fin] - r[fin]) / np.maximum(np.abs(r[fin]), 1e-9)
        # (the reference's own sequence, with this library's pow in place of
        # glibc's)
        assert err.max(initial=0.0) < 1e-9, (name, err.max())


def test_hbvedu_negative_beta_within_its_conditioning(models, oracle,
                                                       hbv_variant):
    """Negative finite Beta (excluded from the fuzz above): the soil update
    prec_eff = lw * (soil/FC)**Beta is singular at soil -> 0, so from some day
    on the reference's OWN result moves by orders of magnitude when an input
    moves by one ulp.  Until then the GPU has to follow the oracle: day by
    day, a set is compared while the oracle's sensitivity to one-ulp
    perturbations (initial states one ulp up; precipitation jittered by one
    ulp a day; the monthly tables moved by one ulp), accumulated up to that
    day, stays below 1e-11 -- at 1e8 x that sensitivity (never tighter than the flat 1e-10, which is what most
    of the compared days get), NaN pattern included: the kernel's power is
    good to (4 + 3 |Beta log2(soil/FC)| + |Beta| / 4) ulp (fastmath.h
    fastpow_tab_lite), up to 200 ulp for these sets as the soil runs dry,
    where the probes move an input by one; the rest is a factor of 5e5 for
    what three probes can sample of a singular system (measured over 200
    seeds: error / sensitivity up to 1.9e7).  The snow series does not see Beta and stays
    bit-exact throughout."""
    g = golden("syn_hbvedu")
    rng = np.random.default_rng(104 + 1000 * SEED)
    lo = np.array([-1, 3, 100, 1, .01, 90, .05, .01, .01, .01, 2.])
    hi = np.array([1, 7, 200, 7, .07, 180, .2, .1, .05, .05, 5.])
    n, t = 256, 400
    flat = lo + (hi - lo) * rng.random((n, 11))
    flat[:, 3] = -rng.choice([0.25, 0.5, 1.0, 1.5, 2.5, 4.0, 7.0], n)
    args = (g["temp"][:t], g["prec"][:t], g["month"][:t] - 1, g["PE_m"],
            g["T_m"])
    inits = (0., 100., 3., 10.)
    with np.errstate(all="ignore"):
        ref = oracle.simulate_hbvedu(*args, inits, flat, return_storage=True,
                                     nthreads=8)
        probes = [
            oracle.simulate_hbvedu(*args, tuple(np.nextafter(v, np.inf)
                                                for v in inits), flat,
                                   return_storage=True, nthreads=8),
            oracle.simulate_hbvedu(args[0], _jitter(rng, args[1]), *args[2:],
                                   inits, flat, return_storage=True,
                                   nthreads=8),
            # (the evaporation's own rounding: monthly tables moved by one ulp)
            oracle.simulate_hbvedu(*args[:3], _jitter(rng, args[3]),
                                   _jitter(rng, args[4]), inits, flat,
                                   return_storage=True, nthreads=8)]
    out = models.HBVEdu().simulate(g["temp"][:t], g["prec"][:t],
                                   g["month"][:t], g["PE_m"], g["T_m"],
                                   *inits, return_storage=True,
                                   params=_records(models.HBVEdu, flat))
    assert np.array_equal(out[1], ref[1], equal_nan=True)       # snow
    compared = 0
    with np.errstate(all="ignore"):
        # sensitivity of every series of a set, accumulated over time
        amp = np.zeros((t, n))
        for k in (0, 2, 3, 4):
            scale = np.maximum(np.abs(ref[k]), 1e-6)
            for pr in probes:
                d = np.abs(pr[k] - ref[k]) / scale
                amp = np.maximum(amp, np.where(np.isfinite(d), d, np.inf))
        amp = np.maximum.accumulate(amp, axis=0)
        well = amp <= 1e-11                  # [t, n]: still well-conditioned
        for k, name in ((0, "qsim"), (2, "soil"), (3, "s1"), (4, "s2")):
            a, b = out[k], ref[k]
            assert np.array_equal(np.isnan(a)[well], np.isnan(b)[well]), name
            fin = well & np.isfinite(b)
            tol = np.maximum(1e8 * amp, RTOL) * np.maximum(np.abs(b), 1e-6)
            bad = fin & ~(np.abs(a - b) <= tol)
            worst = (np.abs(a - b) / (np.maximum(amp, 1e-16) *
                                      np.maximum(np.abs(b), 1e-6)))[bad]
            assert not bad.any(), ("%s: %d values beyond the bound, first at "
                                   "%s; error / sensitivity up to %.3g") \
                % (name, bad.sum(), np.argwhere(bad)[0], worst.max())
            compared += int(fin.sum())
    # the bound is not vacuous: two fifths of all set-days are compared,
    # every set for its first days and nine in ten of them for fifty
    assert compared > 0.3 * 4 * t * n, compared / (4.0 * t * n)
    assert well[:5].all() and well[:50].mean() > 0.9


def test_gr4j_fuzz(models, oracle, gr4j_variant):
    g = golden("syn_gr4j")
    rng = np.random.default_rng(101 + 1000 * SEED)
    lo, hi = np.array([100, -5, 20, 1.1]), np.array([1200, 3, 300, 2.9])
    flat = _wild_params(rng, lo, hi, 640)
    # x4 must still give 1..20 ordinates (anything else is a loud error)
    bad = ~((flat[:, 3] > 0) & (flat[:, 3] <= 20))
    flat[bad, 3] = rng.uniform(0.2, 19.5, bad.sum())
    t = 400
    def probes(f):
        """oracle runs for the conditioning probes of _same: parameters (all
        but x4, which fixes the hydrographs' lengths) and evapotranspiration
        moved by one ulp"""
        fj = f.copy()
        fj[:, :3] = _jitter(rng, f[:, :3])
        with np.errstate(all="ignore"):
            return [oracle.simulate_gr4j(g["prec"][:t], g["etp"][:t],
                                         (0.6, 0.7), fj, return_storage=True,
                                         nthreads=8),
                    oracle.simulate_gr4j(g["prec"][:t],
                                         _jitter(rng, g["etp"][:t]),
                                         (0.6, 0.7), f, return_storage=True,
                                         nthreads=8)]

    with np.errstate(all="ignore"):
        ref = oracle.simulate_gr4j(g["prec"][:t], g["etp"][:t], (0.6, 0.7),
                                   flat, return_storage=True, nthreads=8)
    pr = probes(flat)
    # (no overflow rule: a set that is not civil -- csrc/gr4j_reference.h --
    # is computed with the reference's own sequence, infinities and NaN where
    # and as the reference produces them, compared over the whole series)
    for sl in (slice(0, 640), slice(0, 64)):   # both start at an even set
        out = models.GR4J().simulate(g["prec"][:t], g["etp"][:t], 0.6, 0.7,
                                     return_storage=True,
                                     params=_records(models.GR4J, flat[sl]))
        for j, (a, b, n) in enumerate(zip(out, ref,
                                          ["qsim", "s_store", "r_store"])):
            _same(a, b[:, sl], "gr4j " + n, [q[j][:, sl] for q in pr])
    # register tiers too: all x4 <= 3 / <= 5 / <= 10
    for cap in (2.9, 4.9, 9.9):
        f2 = flat.copy()
        f2[:, 3] = rng.uniform(0.3, cap, 640)
        with np.errstate(all="ignore"):
            ref = oracle.simulate_gr4j(g["prec"][:t], g["etp"][:t],
                                       (0.6, 0.7), f2, nthreads=8)
        out = models.GR4J().simulate(g["prec"][:t], g["etp"][:t], 0.6, 0.7,
                                     params=_records(models.GR4J, f2))
        _same(out, ref, "gr4j tier %g" % cap, [q[0] for q in probes(f2)])


def test_snow_models_fuzz(models, oracle):
    from rrmpg_amd.models import _snowgr4j as core
    h = golden("syn_cemaneigehystgr4j")
    rng = np.random.default_rng(102 + 1000 * SEED)
    t = 500
    layers = tuple(h[k][:t] for k in ("layer_prec", "layer_mean",
                                      "frac_solid", "etp"))
    forcing = (layers[0], layers[1], layers[3], layers[2])
    fice = np.array([0.0, 0.1, 0.3, 0.6, 0.9])
    lo = np.array([0, 0, 1, 0, 10, -5, 20, 1.1, 0.])
    hi = np.array([1, 10, 1000, 1, 1200, 3, 5000, 10, 30.])
    for (hyst, ice), cls, cols in [
            ((True, True), models.CemaneigeHystGR4JIce, list(range(9))),
            ((True, False), models.CemaneigeHystGR4J, list(range(8))),
            ((False, True), models.CemaneigeGR4JIce, [0, 1, 4, 5, 6, 7, 8])]:
        flat = _wild_params(rng, lo[cols], hi[cols], 320)
        ix4 = cls._param_list.index('x4')
        bad = ~((flat[:, ix4] > 0) & (flat[:, ix4] <= 20))
        flat[bad, ix4] = rng.uniform(0.2, 9.9, bad.sum())
        inits = (3.0, -0.2, 0.4, 0.5, 0.6)
        with np.errstate(all="ignore"):
            ref = oracle.simulate_snow_gr4j(hyst, ice, *forcing, inits, flat,
                                            frac_ice=fice if ice else None,
                                            return_storages=True, nthreads=8)
            # conditioning probe for the GR4J part: evapotranspiration
            # jittered by one ulp a day (the snow states do not see it)
            ref2 = oracle.simulate_snow_gr4j(
                hyst, ice, forcing[0], forcing[1], _jitter(rng, forcing[2]),
                forcing[3], inits, flat, frac_ice=fice if ice else None,
                return_storages=True, nthreads=8)
            # ... and the layer precipitation (its melt reaches the stores)
            ref3 = oracle.simulate_snow_gr4j(
                hyst, ice, _jitter(rng, forcing[0]), forcing[1], forcing[2],
                forcing[3], inits, flat, frac_ice=fice if ice else None,
                return_storages=True, nthreads=8)
            # ... and x1, x2, x3 (cancellation in the routing store, _same)
            fj = flat.copy()
            ix1 = cls._param_list.index('x1')
            fj[:, ix1:ix1 + 3] = _jitter(rng, flat[:, ix1:ix1 + 3])
            ref4 = oracle.simulate_snow_gr4j(
                hyst, ice, *forcing, inits, fj,
                frac_ice=fice if ice else None, return_storages=True,
                nthreads=8)
        out, _ = core.run(hyst, ice, layers, fice if ice else None, inits,
                          _records(cls, flat), True, True, None)
        gr4j_part = ("qsim", "s_store", "r_store")
        # (no overflow rule: see test_gr4j_fuzz)
        for k, a in out.items():
            if a is not None:
                _same(a, ref[k], "%s %s" % (cls.__name__, k),
                      [ref2[k], ref3[k], ref4[k]] if k in gr4j_part else None)
    # Cemaneige alone with wild CTG / Kf
    flat = _wild_params(rng, np.array([0., 0.]), np.array([1., 10.]), 320)
    with np.errstate(all="ignore"):
        ref = oracle.simulate_cemaneige(layers[0], layers[1], layers[2],
                                        (3.0, -0.2), flat,
                                        return_storages=True, nthreads=8)
    from rrmpg_amd.models import cemaneige as cmod
    out, _ = cmod._run(layers[:3], (3.0, -0.2),
                       _records(models.Cemaneige, flat), True, True, None)
    for a, b, n in zip(out, ref, ["outflow", "G", "eTG"]):
        _same(a, b, "cemaneige " + n)


def test_cemaneigegr4j_fuzz(models, oracle, fused_variant):
    """The fused kernel -- both of its variants -- on wild parameter blocks
    (special values, extreme ranges) with all storages, plus a ragged number
    of sets so that the tail wave is exercised."""
    from rrmpg_amd.models import cemaneigegr4j as fmod
    h = golden("syn_cemaneigehystgr4j")
    rng = np.random.default_rng(103 + 1000 * SEED)
    t = 500
    layers = tuple(h[k][:t] for k in ("layer_prec", "layer_mean",
                                      "frac_solid", "etp"))
    lo = np.array([0, 0, 10, -5, 20, 0.5])
    hi = np.array([1, 10, 1200, 3, 5000, 2.9])
    flat = _wild_params(rng, lo, hi, 331)
    bad = ~((flat[:, 5] > 0) & (flat[:, 5] <= 20))
    flat[bad, 5] = rng.uniform(0.2, 9.9, bad.sum())
    inits = (3.0, -0.2, 0.4, 0.5)
    with np.errstate(all="ignore"):
        ref = oracle.simulate_cemaneigegr4j(layers[0], layers[1], layers[3],
                                            layers[2], inits, flat,
                                            return_storages=True, nthreads=8)
        # conditioning probes for the GR4J part (see _same)
        ref2 = oracle.simulate_cemaneigegr4j(
            layers[0], layers[1], _jitter(rng, layers[3]), layers[2], inits,
            flat, return_storages=True, nthreads=8)
        ref3 = oracle.simulate_cemaneigegr4j(
            _jitter(rng, layers[0]), layers[1], layers[3], layers[2], inits,
            flat, return_storages=True, nthreads=8)
        fj = flat.copy()
        fj[:, 2:5] = _jitter(rng, flat[:, 2:5])          # x1, x2, x3
        ref4 = oracle.simulate_cemaneigegr4j(
            layers[0], layers[1], layers[3], layers[2], inits, fj,
            return_storages=True, nthreads=8)
    out, _ = fmod._run(layers, inits, _records(models.CemaneigeGR4J, flat),
                       True, True, None)
    gr4j_part = ("qsim", "s_store", "r_store")
    # (no overflow rule: see test_gr4j_fuzz)
    for a, b, b2, b3, b4, n in zip(out, ref, ref2, ref3, ref4,
                                   ["qsim", "G", "eTG", "s_store", "r_store"]):
        _same(a, b, "cemaneigegr4j " + n,
              [b2, b3, b4] if n in gr4j_part else None)


@pytest.mark.parametrize("poison", ["nan_temp", "inf_temp", "negative_snow",
                                    "none"])
def test_snow_kernels_with_poisoned_forcing(models, oracle, poison):
    """The snow kernels' SANE form (one v_min for the thermal state's clamp, no
    sign check of the pack in the idle vote) is only entered when the pre-pass
    found every temperature finite and no snowfall negative, and the wave's
    parameters and the initial states allow it.  Forcing that breaks those
    premises must give what the reference gives -- NaNs where it has them."""
    from rrmpg_amd.models import _snowgr4j as core
    from rrmpg_amd.models import cemaneige as cmod
    from rrmpg_amd.models import cemaneigegr4j as fmod
    h = golden("syn_cemaneigehystgr4j")
    rng = np.random.default_rng(104 + 1000 * SEED)
    t = 400
    lp, lm, fr, etp = (h[k][:t].copy() for k in ("layer_prec", "layer_mean",
                                                 "frac_solid", "etp"))
    if poison == "nan_temp":
        lm[137, 2] = np.nan
    elif poison == "inf_temp":
        lm[90, 0] = np.inf
        lm[200, 4] = -np.inf
    elif poison == "negative_snow":
        lp[50, 1] = -3.0            # prec * frac < 0 on a frost day
        fr[50, 1] = 1.0
    n = 130                         # two full waves and a tail
    for inits_c in [(3.0, -0.2), (0.0, 0.0)]:
        flat = np.column_stack([rng.uniform(0, 1, n), rng.uniform(0, 10, n)])
        with np.errstate(all="ignore"):
            ref = oracle.simulate_cemaneige(lp, lm, fr, inits_c, flat,
                                            return_storages=True, nthreads=8)
        out, _ = cmod._run((lp, lm, fr), inits_c,
                           _records(models.Cemaneige, flat), True, True, None)
        for a, b, name in zip(out, ref, ["outflow", "G", "eTG"]):
            # (NaN pattern exact in all three; the thermal state bit for bit)
            snow_same(a, b, exact=name == "eTG", what=(poison, name))
    flat = np.column_stack([rng.uniform(0, 1, n), rng.uniform(0, 10, n),
                            rng.uniform(10, 1200, n), rng.uniform(-5, 3, n),
                            rng.uniform(20, 300, n), rng.uniform(0.5, 2.9, n)])
    inits = (3.0, -0.2, 0.4, 0.5)
    with np.errstate(all="ignore"):
        ref = oracle.simulate_cemaneigegr4j(lp, lm, etp, fr, inits, flat,
                                            return_storages=True, nthreads=8)
    out, _ = fmod._run((lp, lm, fr, etp), inits,
                       _records(models.CemaneigeGR4J, flat), True, True, None)
    for a, b, name in zip(out, ref, ["qsim", "G", "eTG", "s_store", "r_store"]):
        _same(a, b, "%s cemaneigegr4j %s" % (poison, name))
    snow_same(out[1], ref[1], what=poison)
    snow_same(out[2], ref[2], exact=True, what=poison)
    fice = np.array([0.0, 0.1, 0.3, 0.6, 0.9])
    flat = np.column_stack([rng.uniform(0, 1, n), rng.uniform(0, 10, n),
                            rng.uniform(10, 1200, n), rng.uniform(-5, 3, n),
                            rng.uniform(20, 300, n), rng.uniform(1.1, 2.9, n),
                            rng.uniform(0, 30, n)])
    inits5 = (3.0, -0.2, 0.4, 0.5, 0.6)
    with np.errstate(all="ignore"):
        ref = oracle.simulate_snow_gr4j(False, True, lp, lm, etp, fr, inits5,
                                        flat, frac_ice=fice,
                                        return_storages=True, nthreads=8)
    out, _ = core.run(False, True, (lp, lm, fr, etp), fice, inits5,
                      _records(models.CemaneigeGR4JIce, flat), True, True,
                      None)
    for k, a in out.items():
        if a is not None:
            _same(a, ref[k], "%s ice %s" % (poison, k))
    # hysteresis (+ ice): its SANE form additionally needs Kf >= +0, Thacc > 0
    # inside the 3-FMA quotient's range, finite Rsp -- lanes that break this
    # (every third wave has some) send their wave through the general form
    for ice in (False, True):
        cols = 9 if ice else 8
        flat = np.column_stack([rng.uniform(0, 1, n), rng.uniform(0, 10, n),
                                rng.uniform(1, 1000, n), rng.uniform(0, 1, n),
                                rng.uniform(10, 1200, n), rng.uniform(-5, 3, n),
                                rng.uniform(20, 300, n),
                                rng.uniform(1.1, 2.9, n),
                                rng.uniform(0, 30, n)])[:, :cols]
        flat[0, 2] = 0.0            # Thacc = 0: 0/0 on days without snowfall
        flat[1, 2] = -50.0          # negative Thacc
        flat[2, 1] = -0.0           # Kf = -0
        flat[3, 1] = 0.0            # Kf = +0 (allowed)
        flat[4, 3] = np.nan         # Rsp
        flat[5, 3] = 0.0            # Rsp = 0: Thmelt = 0
        flat[6, 1] = np.inf         # Kf = inf (allowed)
        cls = models.CemaneigeHystGR4JIce if ice else models.CemaneigeHystGR4J
        with np.errstate(all="ignore"):
            ref = oracle.simulate_snow_gr4j(True, ice, lp, lm, etp, fr, inits5,
                                            flat, frac_ice=fice if ice else None,
                                            return_storages=True, nthreads=8)
        out, _ = core.run(True, ice, (lp, lm, fr, etp), fice if ice else None,
                          inits5, _records(cls, flat), True, True, None)
        for k, a in out.items():
            if a is None:
                continue
            if k in ("G", "eTG", "sca"):
                snow_same(a, ref[k], exact=k != "G", what=(poison, ice, k))
            else:
                _same(a, ref[k], "%s hyst %s" % (poison, k))


def test_infinite_etp_where_no_reference_kernel_follows(models, oracle):
    """etp = +inf on a day with finite rain: the reference's dry arm gives
    p_r = perc + (0 - 0), a number (gr4j_model.py:102-123).  The fast GR4J
    kernels form that excess as (net - frac) times a factor 0.0 -- NaN here --
    and leave such launches to the one-lane reference kernel behind them;
    the two kinds of kernel that have none (the HBM-scratch tier for x4 > 20,
    the coupled kernels for more than eight layers) select the excess per
    lane instead (round 5's advisor finding: they returned NaN)."""
    from rrmpg_amd.models import cemaneigegr4j as fmod
    g = golden("syn_gr4j")
    rng = np.random.default_rng(107 + 1000 * SEED)
    t = 300
    prec, etp = g["prec"][:t].copy(), g["etp"][:t].copy()
    etp[[40, 41, 200]] = np.inf
    n = 130
    flat = np.column_stack([rng.uniform(100, 1200, n), rng.uniform(-5, 3, n),
                            rng.uniform(20, 300, n), rng.uniform(20.5, 33, n)])
    with np.errstate(all="ignore"):
        ref = oracle.simulate_gr4j(prec, etp, (0.6, 0.7), flat,
                                   return_storage=True, nthreads=8)
    assert np.isfinite(ref[0]).all()
    out = models.GR4J().simulate(prec, etp, 0.6, 0.7, return_storage=True,
                                 params=_records(models.GR4J, flat))
    for a, b, name in zip(out, ref, ["qsim", "s_store", "r_store"]):
        _same(a, b, "gr4j x4 > 20, etp = inf: " + name)
    # nine layers (> RR_CEMANEIGE_MAX_LAYERS): the five of the fixture + four
    h = golden("syn_cemaneigehystgr4j")
    lp, lm, fr = (np.concatenate([h[k][:t], h[k][:t, :4]], axis=1)
                  for k in ("layer_prec", "layer_mean", "frac_solid"))
    etp9 = h["etp"][:t].copy()
    etp9[[40, 41, 200]] = np.inf
    flat = np.column_stack([rng.uniform(0, 1, n), rng.uniform(0, 10, n),
                            rng.uniform(100, 1200, n), rng.uniform(-5, 3, n),
                            rng.uniform(20, 300, n), rng.uniform(0.5, 9.5, n)])
    inits = (3.0, -0.2, 0.4, 0.5)
    with np.errstate(all="ignore"):
        ref = oracle.simulate_cemaneigegr4j(lp, lm, etp9, fr, inits, flat,
                                            return_storages=True, nthreads=8)
    assert np.isfinite(ref[0]).all()
    out, _ = fmod._run((lp, lm, fr, etp9), inits,
                       _records(models.CemaneigeGR4J, flat), True, True, None)
    for a, b, name in zip(out, ref, ["qsim", "G", "eTG", "s_store", "r_store"]):
        _same(a, b, "nine layers, etp = inf: " + name)


@pytest.mark.parametrize("poison", ["nan_prec", "negative_prec",
                                    "minus_zero_prec", "nan_snow_init",
                                    "minus_zero_snow_init", "none"])
def test_hbvedu_with_poisoned_snow_inputs(models, oracle, hbv_variant, poison):
    """HBV-Edu's TAME copy of the time loop takes min(snow, melt) with one
    v_min_f64, which differs from numba's min for a NaN pack and for a melt of
    -0 against an empty pack.  It is only entered when the pre-pass found no
    precipitation that is NaN, negative or -0, the initial pack is a number
    that is not negative, and no lane has a negative degree-day factor.
    Inputs that break those premises must still give the reference's values,
    bit for bit in the snow series."""
    from rrmpg_amd.models import hbvedu as hmod
    g = golden("syn_hbvedu")
    rng = np.random.default_rng(105 + 1000 * SEED)
    t = 400
    temp, prec = g["temp"][:t].copy(), g["prec"][:t].copy()
    month0 = (g["month"][:t] - 1).astype(np.int8)
    snow_init = 0.0
    if poison == "nan_prec":
        prec[33] = np.nan
    elif poison == "negative_prec":
        prec[np.argmin(temp[:200])] = -4.0          # a frost day: pack < 0
    elif poison == "minus_zero_prec":
        prec[prec == 0.0] = -0.0
    elif poison == "nan_snow_init":
        snow_init = np.nan
    elif poison == "minus_zero_snow_init":
        snow_init = -0.0
    n = 200
    lo = np.array([-1, 3, 100, 1, .01, 90, .05, .01, .01, .01, 2.])
    hi = np.array([1, 7, 200, 7, .07, 180, .2, .1, .05, .05, 5.])
    flat = rng.uniform(lo, hi, (n, 11))
    flat[::7, 1] = 0.0              # DD = +0: melt is +0 on warm days
    flat[3::11, 1] = -0.0           # DD = -0 (one lane in some waves)
    flat[5::50, 1] = -2.0           # a negative degree-day factor
    inits = (snow_init, 100., 3., 10.)
    with np.errstate(all="ignore"):
        ref = oracle.simulate_hbvedu(temp, prec, month0, g["PE_m"], g["T_m"],
                                     inits, flat, return_storage=True,
                                     nthreads=8)
    out, _ = hmod._run((temp, prec, month0, g["PE_m"], g["T_m"]), inits,
                       _records(models.HBVEdu, flat), True, True, None)
    assert np.array_equal(out[1], ref[1], equal_nan=True), poison + ": snow"
    assert np.array_equal(np.signbit(out[1]), np.signbit(ref[1])), poison
    for a, b, name in zip(out, ref, ["qsim", "snow", "soil", "s1", "s2"]):
        nan_a, nan_b = np.isnan(a), np.isnan(b)
        assert np.array_equal(nan_a, nan_b), (poison, name)
        ok = ~nan_b & np.isfinite(b)
        assert np.allclose(a[ok], b[ok], rtol=1e-9, atol=1e-9), (poison, name)
    # a block of tame lanes only (every premise holds when poison == "none")
    tame = flat[(flat[:, 1] > 0)][:64]
    with np.errstate(all="ignore"):
        ref = oracle.simulate_hbvedu(temp, prec, month0, g["PE_m"], g["T_m"],
                                     inits, tame, return_storage=True)
    out, _ = hmod._run((temp, prec, month0, g["PE_m"], g["T_m"]), inits,
                       _records(models.HBVEdu, tame), True, True, None)
    assert np.array_equal(out[1], ref[1], equal_nan=True), poison + ": snow"
    assert np.array_equal(np.signbit(out[1]), np.signbit(ref[1])), poison


def test_kernel_variants_agree_bit_for_bit(models):
    """The size-dependent kernel choices must not change a single bit: a sweep
    sharded over GPUs (other launch sizes, hence other variants) has to give
    the numbers of the unsharded one.  HBV-Edu: plain loop / prefetch loop /
    LDS forcing, each with and without the TAME copy (the launch size decides:
    300 sets vs 90,000 sets sit in different windows); CemaneigeGR4J: the
    many-waves and the small-sweep kernel."""
    from rrmpg_amd import _lib
    from rrmpg_amd.models import cemaneigegr4j as fmod
    from rrmpg_amd.models import hbvedu as hmod
    g = golden("syn_hbvedu")
    rng = np.random.default_rng(106 + 1000 * SEED)
    t = 300
    forcing = (g["temp"][:t], g["prec"][:t], (g["month"][:t] - 1).astype(np.int8),
               g["PE_m"], g["T_m"])
    lo = np.array([-1, 3, 100, 1, .01, 90, .05, .01, .01, .01, 2.])
    hi = np.array([1, 7, 200, 7, .07, 180, .2, .1, .05, .05, 5.])
    n = 300
    flat = rng.uniform(lo, hi, (n, 11))
    rec = _records(models.HBVEdu, flat)
    inits = (0., 100., 3., 10.)
    base = None
    for v in (-1, 0, 3):
        with _lib.debug_option("hbv_variant", v):
            out, _ = hmod._run(forcing, inits, rec, True, True, None)
        if base is None:
            base = out
        for a, b in zip(out, base):
            assert np.array_equal(a, b), "HBV variant %d" % v
    # the time-tiled persistent form of the plain loop (million-set sweeps):
    # pieces of the time axis pulled from a work queue, states handed from
    # piece to piece through HBM -- every output and the score sum bit for bit
    qobs = rng.uniform(0, 5, t)
    with _lib.debug_option("hbv_variant", 0):
        ref_out, ref_sse = hmod._run(forcing, inits, rec, True, True, qobs)
        for tiles in (2, 3, 4, 7, 64):
            with _lib.debug_option("time_tiles", tiles):
                out, sse = hmod._run(forcing, inits, rec, True, True, qobs)
                only, sse2 = hmod._run(forcing, inits, rec, False, False, qobs)
            for a, b in zip(out, ref_out):
                assert np.array_equal(a, b), "HBV tiles %d" % tiles
            assert np.array_equal(sse, ref_sse) and np.array_equal(sse2, ref_sse)
    # ... and of the multi-catchment launch (the queue's slots run over
    # catchments x waves): three catchments with their own forcing and inits
    import torch
    from rrmpg_amd import device as dev
    C = 3
    ctemp = np.stack([forcing[0] + c for c in range(C)])
    cprec = np.stack([forcing[1] * (1 + 0.3 * c) for c in range(C)])
    cmonth = np.stack([g["month"][:t]] * C)
    cinits = np.array([[0., 100., 3., 10.], [5., 80., 1., 2.], [0., 150., 0., 0.]])
    cens = dev.HBVEduCatchments(ctemp, cprec, cmonth, np.stack([g["PE_m"]] * C),
                                np.stack([g["T_m"]] * C), cinits)
    cpar = torch.from_numpy(np.stack([flat[::-1], flat, flat * 1.01])
                            .copy()).cuda()
    cq = torch.from_numpy(np.stack([qobs, qobs * 2, qobs + 1])).cuda()
    with _lib.debug_option("hbv_variant", 0):
        with _lib.debug_option("time_tiles", 0):
            q0 = cens.new_output(n)
            s0 = cens.run(cpar, q0, qobs=cq).clone()
        for tiles in (2, 5):
            with _lib.debug_option("time_tiles", tiles):
                q1 = cens.new_output(n)
                st1 = [cens.new_output(n) for _ in range(4)]
                s1 = cens.run(cpar, q1, storages=st1, qobs=cq).clone()
                s2 = cens.run(cpar, None, qobs=cq).clone()
            torch.cuda.synchronize()
            # (bit patterns: the third catchment's sets produce NaNs)
            bits = lambda x: x.view(torch.int64)
            assert torch.equal(bits(q0), bits(q1)), tiles
            assert torch.equal(bits(s0), bits(s1)), tiles
            assert torch.equal(bits(s0), bits(s2)), tiles
    # the same sets inside a launch of two waves per SIMD (no TAME copy)
    big = np.tile(flat, (300, 1))[:90_000]
    outb, _ = hmod._run(forcing, inits, _records(models.HBVEdu, big), True,
                        False, None)
    assert np.array_equal(outb[0][:, :n], base[0])
    assert np.array_equal(outb[0][:, 60_000:60_000 + n], base[0])
    # fused kernel variants
    h = golden("syn_cemaneigehystgr4j")
    layers = tuple(h[k][:t] for k in ("layer_prec", "layer_mean",
                                      "frac_solid", "etp"))
    lo = np.array([0, 0, 10, -5, 20, 0.5])
    hi = np.array([1, 10, 1200, 3, 300, 4.9])
    flat = rng.uniform(lo, hi, (n, 6))
    rec = _records(models.CemaneigeGR4J, flat)
    base = None
    for v in (1, 2, 3, 0):
        with _lib.debug_option("fused_variant", v):
            out, _ = fmod._run(layers, (3.0, -0.2, 0.4, 0.5), rec, True, True,
                               None)
        if base is None:
            base = out
        for a, b in zip(out, base):
            assert np.array_equal(a, b), "fused variant %d" % v
    # ... and the scores of the many-waves kernel
    fq = rng.uniform(0, 3, t)
    with _lib.debug_option("fused_variant", 1):
        _, fsse = fmod._run(layers, (3.0, -0.2, 0.4, 0.5), rec, False, False,
                            fq)
    # the score-only forms of every variant: the sums of the sweep that
    # writes its series
    for v in (0, 2, 3):
        with _lib.debug_option("fused_variant", v):
            _, sse_v = fmod._run(layers, (3.0, -0.2, 0.4, 0.5), rec, False,
                                 False, fq)
        assert np.array_equal(sse_v, fsse), "fused variant %d, scores" % v
    # Cemaneige: the time-tiled form (million-set sweeps) against the plain loop
    cm = models.Cemaneige()
    crec = _records(models.Cemaneige, rng.uniform([0, 0], [1, 10], (n, 2)))
    ckw = dict(prec=h["prec"][:t] if "prec" in h else h["layer_prec"][:t, 0],
               mean_temp=h["layer_mean"][:t, 0] + 2,
               min_temp=h["layer_mean"][:t, 0] - 3,
               max_temp=h["layer_mean"][:t, 0] + 6, met_station_height=500,
               altitudes=[550, 620, 700, 785, 920], snow_pack_init=2.0,
               thermal_state_init=-0.3, return_storages=True, params=crec)
    with _lib.debug_option("time_tiles", 0):
        cbase = cm.simulate(**ckw)
    for tiles in (2, 3, 5):
        with _lib.debug_option("time_tiles", tiles):
            cout = cm.simulate(**ckw)
        for a, b in zip(cout, cbase):
            assert np.array_equal(a, b), "Cemaneige tiles %d" % tiles
    # GR4J: one wave per 64 sets / production and routing in two waves
    from rrmpg_amd.models import gr4j as gmod
    lo = np.array([10, -5, 20, 0.5])
    hi = np.array([1200, 3, 300, 4.9])
    for hi_x4 in (2.9, 4.9):            # both register tiers with a pipe form
        hi[3] = hi_x4
        flat = rng.uniform(lo, hi, (n, 4))
        rec = _records(models.GR4J, flat)
        qobs = rng.uniform(0, 3, t)
        base = None
        for v in (1, 0):
            with _lib.debug_option("gr4j_variant", v):
                out, sse = gmod._run(h["layer_prec"][:t, 0], h["etp"][:t], 0.4,
                                     0.5, rec, True, True, qobs)
            if base is None:
                base = list(out) + [sse]
            for a, b in zip(list(out) + [sse], base):
                assert np.array_equal(a, b), "GR4J variant %d" % v
        # the time-tiled persistent form of the optimistic kernel
        for tiles in (2, 3, 4, 9):
            with _lib.debug_option("time_tiles", tiles):
                out, sse = gmod._run(h["layer_prec"][:t, 0], h["etp"][:t], 0.4,
                                     0.5, rec, True, True, qobs)
                _, sse2 = gmod._run(h["layer_prec"][:t, 0], h["etp"][:t], 0.4,
                                    0.5, rec, False, False, qobs)
            for a, b in zip(list(out) + [sse, sse2], base + [base[-1]]):
                assert np.array_equal(a, b), "GR4J tiles %d" % tiles


def test_reference_kernels_hand_gr4j_the_reference_snow_outflow(models,
                                                                oracle):
    """Soak seed 200, set 281 of the ice-melt coupling, kept as a case of
    its own: x1 = 1e308 mm forms tanh(p_n / x1) among the subnormals, so
    p_n - p_s is a rounding residue of +-1e-17 mm whose sign one ulp of the
    snow outflow decides; under x3 = -0.25 a positive residue left in the
    routing store makes the next day's exchange term x2 (r / x3)**3.5 a NaN,
    which numba's max(0, .) swallows together with the day's discharge -- 0
    where the reference has 9e302.  The one-lane kernels of the sets that are
    not civil therefore run the reference's own snow day (csrc/snow_core.h
    cema_ref_day, csrc/snownext_kernels.h cema_hyst_day<.., REF>): the routing
    store is empty on the very days the reference's is, and the discharge
    follows it.  No probes, no excuses: rtol 1e-9 on every day (pow / tanh
    of the device library against glibc's), zeros and non-finite values in
    the same places."""
    from rrmpg_amd.models import _snowgr4j as core
    h = golden("syn_cemaneigehystgr4j")
    t = 500
    layers = tuple(h[k][:t] for k in ("layer_prec", "layer_mean",
                                      "frac_solid", "etp"))
    forcing = (layers[0], layers[1], layers[3], layers[2])
    fice = np.array([0.0, 0.1, 0.3, 0.6, 0.9])
    inits = (3.0, -0.2, 0.4, 0.5, 0.6)
    ice_set = [0.40534207254510635, -1e308, 1e308, -2.9360458153483835,
               -0.25, 9.7846766997872123, 1e-200]
    # the same set behind the hysteresis routine (Thacc, Rsp in bounds)
    for (hyst, ice), cls, row in [
            ((False, True), models.CemaneigeGR4JIce, ice_set),
            ((True, True), models.CemaneigeHystGR4JIce,
             ice_set[:2] + [300.0, 0.4] + ice_set[2:]),
            ((True, False), models.CemaneigeHystGR4J,
             ice_set[:2] + [300.0, 0.4] + ice_set[2:6])]:
        # (an in-bounds neighbour on either side: the set sits in a wave
        # with civil ones, as in the soak)
        civil = [0.5, 4.0] + ([300.0, 0.4] if hyst else []) + \
                [350.0, 0.5, 90.0, 2.2] + ([5.0] if ice else [])
        flat = np.array([civil, row, civil])
        with np.errstate(all="ignore"):
            ref = oracle.simulate_snow_gr4j(
                hyst, ice, *forcing, inits, flat,
                frac_ice=fice if ice else None, return_storages=True,
                nthreads=1)
        out, _ = core.run(hyst, ice, layers, fice if ice else None, inits,
                          _records(cls, flat), True, True, None)
        for k in ("qsim", "s_store", "r_store"):
            a, b = np.asarray(out[k])[..., 1], np.asarray(ref[k])[..., 1]
            what = "%s %s" % (cls.__name__, k)
            assert np.array_equal(np.isnan(a), np.isnan(b)), what
            assert np.array_equal(np.isinf(a), np.isinf(b)), what
            assert np.array_equal(a == 0, b == 0), what + ": zeros"
            fin = np.isfinite(b)
            with np.errstate(all="ignore"):
                assert np.all(np.abs(a[fin] - b[fin]) <=
                              1e-9 * np.abs(b[fin])), what
        assert np.isfinite(np.asarray(ref["qsim"])[..., 1]).any()
