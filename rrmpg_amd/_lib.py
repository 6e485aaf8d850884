"""ctypes binding of librrhip.so (include/rrhip.h) -- the only compute path.

There is deliberately no fallback: if the shared library is missing, or no
MI355X is visible when a simulation is requested, a RuntimeError is raised.
"""

import ctypes
import importlib.util
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librrhip.so")

_f64p = ctypes.POINTER(ctypes.c_double)
_i8p = ctypes.POINTER(ctypes.c_int8)
_i64 = ctypes.c_int64
_dbl = ctypes.c_double
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t

RR_OK = 0
ERROR_NAMES = {-1: "RR_E_NULL", -2: "RR_E_SIZE", -3: "RR_E_HIP",
               -4: "RR_E_PARAM", -5: "RR_E_NODEVICE", -6: "RR_E_WORKSPACE"}

RR_OPT_UNSET = -2 ** 63


class CallOptions(ctypes.Structure):
    """include/rrhip.h rr_call_options: per-call options of the host-pointer
    family (rr_<model>_simulate_opt), indexed by RR_OPT_*."""
    _fields_ = [("struct_bytes", ctypes.c_size_t),
                ("value", ctypes.c_int64 * 16)]


_optp = ctypes.POINTER(CallOptions)

# name -> (restype, argtypes); mirrors include/rrhip.h one to one
_SIGNATURES = {
    "rr_version": (ctypes.c_int, []),
    "rr_device_count": (ctypes.c_int, []),
    "rr_last_error": (ctypes.c_char_p, []),
    "rr_set_device": (ctypes.c_int, [ctypes.c_int]),
    "rr_get_device": (ctypes.c_int, []),
    "rr_release_cached_memory": (ctypes.c_int, []),
    "rr_shard_bounds": (ctypes.c_int, [_i64, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(_i64),
                                       ctypes.POINTER(_i64)]),
    "rr_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_comm_init": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p),
                                    ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p]),
    "rr_allgather_metric": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p,
                                           _i64, ctypes.c_void_p, _i64,
                                           ctypes.c_void_p]),
    "rr_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "rr_comm_init_all": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_int)]),
    "rr_comm_group_start": (ctypes.c_int, []),
    "rr_comm_group_end": (ctypes.c_int, []),
    "rr_debug_set_option": (ctypes.c_int, [ctypes.c_int, _i64]),
    "rr_debug_get_option": (_i64, [ctypes.c_int]),
    "rr_call_options_init": (None, [_optp]),
    "rr_call_options_set": (ctypes.c_int, [_optp, ctypes.c_int, _i64]),
    "rr_thread_options": (ctypes.c_int, [_optp]),
    "rr_column_sums_dev": (ctypes.c_int, [_vp, _i64, _vp, _i64, _i64, _vp,
                                          _vp]),
    "rr_column_sums_shifted_dev": (ctypes.c_int,
                                   [_vp, _i64, _vp, _i64, _i64, _dbl, _vp,
                                    _vp]),
    "rr_sample_params_dev": (ctypes.c_int,
                             [ctypes.c_uint64, ctypes.c_int, _f64p, _f64p,
                              ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                              _i64, _i64, _i64, _vp, _vp]),
    "rr_abc_workspace_bytes": (_sz, [_i64, _i64]),
    "rr_abc_simulate_dev": (ctypes.c_int, [_vp, _i64, _dbl, _vp, _i64, _vp,
                                           _vp, _i64, _vp, _vp, _vp, _sz,
                                           _vp]),
    "rr_abc_simulate": (ctypes.c_int, [_f64p, _i64, _dbl, _f64p, _i64, _f64p,
                                       _f64p, _f64p, _f64p]),
    "rr_abc_simulate_opt": (ctypes.c_int, ([_f64p, _i64, _dbl, _f64p, _i64, _f64p,
                                       _f64p, _f64p, _f64p]) + [_optp]),
    "rr_hbvedu_workspace_bytes": (_sz, [_i64, _i64]),
    "rr_hbvedu_simulate_dev": (ctypes.c_int,
                               [_vp] * 5 + [_i64] + [_dbl] * 4 + [_vp, _i64]
                               + [_vp] * 5 + [_i64, _vp, _vp, _vp, _sz, _vp]),
    "rr_hbvedu_simulate": (ctypes.c_int,
                           [_f64p, _f64p, _i8p, _f64p, _f64p, _i64]
                           + [_dbl] * 4 + [_f64p, _i64] + [_f64p] * 7),
    "rr_hbvedu_simulate_opt": (ctypes.c_int, ([_f64p, _f64p, _i8p, _f64p, _f64p, _i64]
                           + [_dbl] * 4 + [_f64p, _i64] + [_f64p] * 7) + [_optp]),
    "rr_hbvedu_catchments_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "rr_hbvedu_simulate_catchments_dev": (ctypes.c_int,
                                          [_vp] * 5 + [_i64, _i64, _vp, _vp,
                                                       _i64] + [_vp] * 5
                                          + [_i64, _vp, _vp, _vp, _sz, _vp]),
    "rr_gr4j_workspace_bytes": (_sz, [_i64, _i64]),
    "rr_gr4j_workspace_bytes_x4": (_sz, [_i64, _i64, _dbl]),
    "rr_gr4j_plan_status": (ctypes.c_int, [_vp, _vp]),
    "rr_gr4j_simulate_dev": (ctypes.c_int,
                             [_vp, _vp, _i64, _dbl, _dbl, _vp, _i64]
                             + [_vp] * 3 + [_i64, _vp, _vp, _vp, _sz, _vp]),
    "rr_gr4j_simulate": (ctypes.c_int,
                         [_f64p, _f64p, _i64, _dbl, _dbl, _f64p, _i64]
                         + [_f64p] * 5),
    "rr_gr4j_simulate_opt": (ctypes.c_int, ([_f64p, _f64p, _i64, _dbl, _dbl, _f64p, _i64]
                         + [_f64p] * 5) + [_optp]),
    "rr_cemaneige_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "rr_cemaneige_simulate_dev": (ctypes.c_int,
                                  [_vp] * 3 + [_i64, _i64, _dbl, _dbl, _vp,
                                               _i64] + [_vp] * 3
                                  + [_i64, _vp, _vp, _vp, _sz, _vp]),
    "rr_cemaneige_simulate": (ctypes.c_int,
                              [_f64p] * 3 + [_i64, _i64, _dbl, _dbl, _f64p,
                                             _i64] + [_f64p] * 5),
    "rr_cemaneige_simulate_opt": (ctypes.c_int, ([_f64p] * 3 + [_i64, _i64, _dbl, _dbl, _f64p,
                                             _i64] + [_f64p] * 5) + [_optp]),
    "rr_cemaneige_layers_workspace_bytes": (_sz, [_i64]),
    "rr_cemaneige_layers_dev": (ctypes.c_int,
                                [_vp] * 4 + [_i64, _f64p, _i64, _dbl, _f64p]
                                + [_vp] * 4 + [_sz, _vp]),
    "rr_cemaneigegr4j_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "rr_cemaneigegr4j_workspace_bytes_x4": (_sz, [_i64, _i64, _i64, _dbl]),
    "rr_cemaneigegr4j_simulate_dev": (ctypes.c_int,
                                      [_vp] * 4 + [_i64, _i64] + [_dbl] * 4
                                      + [_vp, _i64] + [_vp] * 5
                                      + [_i64, _vp, _vp, _vp, _sz, _vp]),
    "rr_cemaneigegr4j_simulate": (ctypes.c_int,
                                  [_f64p] * 4 + [_i64, _i64] + [_dbl] * 4
                                  + [_f64p, _i64] + [_f64p] * 7),
    "rr_cemaneigegr4j_simulate_opt": (ctypes.c_int, ([_f64p] * 4 + [_i64, _i64] + [_dbl] * 4
                                  + [_f64p, _i64] + [_f64p] * 7) + [_optp]),
    "rr_snowgr4j_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "rr_snowgr4j_workspace_bytes_x4": (_sz, [_i64, _i64, _i64, _dbl]),
    "rr_cemaneigehystgr4j_simulate_dev": (
        ctypes.c_int, [_vp] * 4 + [_i64, _i64] + [_dbl] * 5 + [_vp, _i64]
        + [_vp] * 6 + [_i64, _vp, _vp, _vp, _sz, _vp]),
    "rr_cemaneigehystgr4j_simulate": (
        ctypes.c_int, [_f64p] * 4 + [_i64, _i64] + [_dbl] * 5 + [_f64p, _i64]
        + [_f64p] * 8),
    "rr_cemaneigehystgr4j_simulate_opt": (ctypes.c_int, ([_f64p] * 4 + [_i64, _i64] + [_dbl] * 5 + [_f64p, _i64]
        + [_f64p] * 8) + [_optp]),
    "rr_cemaneigegr4jice_simulate_dev": (
        ctypes.c_int, [_vp] * 5 + [_i64, _i64] + [_dbl] * 4 + [_vp, _i64]
        + [_vp] * 6 + [_i64, _vp, _vp, _vp, _sz, _vp]),
    "rr_cemaneigegr4jice_simulate": (
        ctypes.c_int, [_f64p] * 5 + [_i64, _i64] + [_dbl] * 4 + [_f64p, _i64]
        + [_f64p] * 8),
    "rr_cemaneigegr4jice_simulate_opt": (ctypes.c_int, ([_f64p] * 5 + [_i64, _i64] + [_dbl] * 4 + [_f64p, _i64]
        + [_f64p] * 8) + [_optp]),
    "rr_cemaneigehystgr4jice_simulate_dev": (
        ctypes.c_int, [_vp] * 5 + [_i64, _i64] + [_dbl] * 5 + [_vp, _i64]
        + [_vp] * 8 + [_i64, _vp, _vp, _vp, _sz, _vp]),
    "rr_cemaneigehystgr4jice_simulate": (
        ctypes.c_int, [_f64p] * 5 + [_i64, _i64] + [_dbl] * 5 + [_f64p, _i64]
        + [_f64p] * 10),
    "rr_cemaneigehystgr4jice_simulate_opt": (ctypes.c_int, ([_f64p] * 5 + [_i64, _i64] + [_dbl] * 5 + [_f64p, _i64]
        + [_f64p] * 10) + [_optp]),
}

_lib = None


def exported_names():
    """Every entry point include/rrhip.h declares."""
    return sorted(_SIGNATURES)


def _share_hip_runtime_with_torch():
    """Make librrhip and PyTorch use ONE HIP runtime in this process.

    The PyTorch ROCm wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7) and asks for it as "libamdhip64.so".  If librrhip pulled
    in the system runtime first, a later `import torch` would load a second
    copy and see no GPU.  Loading torch's copy first makes librrhip's
    libamdhip64.so.7 dependency resolve to it.  Without torch installed the
    system runtime is used.
    """
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib",
                        "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load librrhip.so (once).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is None:
        _share_hip_runtime_with_torch()
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "librrhip.so is not built (%s). Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C rrmpg_amd/csrc`. There is no CPU fallback."
                % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    return load().rr_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != RR_OK:
        raise RuntimeError("%s failed with %s: %s"
                           % (what, ERROR_NAMES.get(rc, rc), last_error()))


def device_count():
    return int(load().rr_device_count())


def set_device(index):
    """Run this process's later sweeps on HIP device `index` (one process per
    GPU: pass the local rank)."""
    check(load().rr_set_device(int(index)), "rr_set_device")


# include/rrhip.h RR_OPT_*
OPTIONS = {"hbv_variant": 1, "gr4j_force_lds": 2, "max_block_cols": 3,
           "gather_threads": 4, "fused_variant": 5, "gr4j_variant": 6,
           "host_shards": 7, "time_tiles": 8, "warm_records": 9}


_tls = threading.local()


def _new_options(values):
    lib = load()
    opt = CallOptions()
    lib.rr_call_options_init(ctypes.byref(opt))
    for name, value in values.items():
        if value is None:
            continue
        check(lib.rr_call_options_set(ctypes.byref(opt), OPTIONS[name],
                                      int(value)), "rr_call_options_set")
    return opt


class call_options:
    """``with call_options(host_shards=3): model.simulate(...)`` -- per-call
    options (include/rrhip.h rr_call_options) for every host-pointer call this
    THREAD makes inside the block: they travel as the `opt` argument of
    rr_<model>_simulate_opt, so concurrent callers in other threads are not
    affected (this is what ``monte_carlo(gpus=...)`` and ``sharding.sweep``
    use).  Nested blocks merge, the inner one winning."""

    def __init__(self, **values):
        for name in values:
            if name not in OPTIONS:
                raise KeyError("unknown option %r (one of %s)"
                               % (name, sorted(OPTIONS)))
        self.values = values

    def __enter__(self):
        self.prev = getattr(_tls, "stack", None)
        merged = dict(self.prev[0]) if self.prev else {}
        # (None: "not set here" -- the enclosing block's value stays)
        merged.update({k: v for k, v in self.values.items() if v is not None})
        _tls.stack = (merged, _new_options(merged))
        return self

    def __exit__(self, *exc):
        _tls.stack = self.prev
        return False


def host_shards_of(gpus):
    """The RR_OPT_HOST_SHARDS value of a ``gpus=`` argument: None -> unset
    (the current device), 'all' -> -1 (one shard per visible device), a
    positive int -> that many shards.  (0 is rejected since round 4: it
    used to mean "the current device", which None says; an integral float
    such as 2.0 counts as its int.)"""
    if gpus is None:
        return None
    if isinstance(gpus, str):
        if gpus == "all":
            return -1
        raise ValueError("gpus must be a positive int, 'all' or None")
    if isinstance(gpus, bool) or int(gpus) != gpus or int(gpus) < 1:
        raise ValueError("gpus must be a positive int, 'all' or None")
    return int(gpus)


def opts_ptr():
    """The `opt` argument of a host-pointer call made by this thread now:
    the innermost call_options block's struct, or NULL."""
    stack = getattr(_tls, "stack", None)
    return ctypes.byref(stack[1]) if stack else None


class thread_options:
    """``with thread_options(hbv_variant=0): ens.run(...)`` -- standing
    options of the calling thread (rr_thread_options) for the *_simulate_dev
    family, which has no per-call argument.  Nested blocks merge, the inner
    one winning (None: the enclosing block's value stays); on exit the
    enclosing block's options stand again (cleared behind the outermost)."""

    def __init__(self, **values):
        for name in values:
            if name not in OPTIONS:
                raise KeyError("unknown option %r (one of %s)"
                               % (name, sorted(OPTIONS)))
        self.values = values

    def __enter__(self):
        self.prev = getattr(_tls, "standing", None)
        merged = dict(self.prev) if self.prev else {}
        merged.update({k: v for k, v in self.values.items() if v is not None})
        opt = _new_options(merged)
        check(load().rr_thread_options(ctypes.byref(opt)),
              "rr_thread_options")
        _tls.standing = merged
        return self

    def __exit__(self, *exc):
        _tls.standing = self.prev
        if self.prev:
            opt = _new_options(self.prev)
            check(load().rr_thread_options(ctypes.byref(opt)),
                  "rr_thread_options")
        else:
            check(load().rr_thread_options(None), "rr_thread_options")
        return False


class debug_option:
    """``with debug_option("hbv_variant", 0): ...`` pins a PROCESS-WIDE
    measurement / test option of the library (rr_debug_set_option) and
    restores the previous value afterwards.  Tests and A/B timing only -- not
    thread safe by nature; product code passes call_options instead."""

    def __init__(self, name, value):
        self.opt, self.value = OPTIONS[name], int(value)

    def __enter__(self):
        lib = load()
        self.prev = int(lib.rr_debug_get_option(self.opt))
        check(lib.rr_debug_set_option(self.opt, self.value),
              "rr_debug_set_option")
        return self

    def __exit__(self, *exc):
        check(load().rr_debug_set_option(self.opt, self.prev),
              "rr_debug_set_option")
        return False


def require_gpu():
    if device_count() < 1:
        raise RuntimeError("rrmpg_amd needs an AMD MI355X (gfx950) GPU: no "
                           "HIP device is visible and there is no CPU path.")


def f64(a):
    """C-contiguous float64 view/copy + its pointer (None passes NULL)."""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_f64p)


def f64s(*arrays):
    """Contiguous float64 versions of `arrays` and their pointers.

    Returns (keepalive, pointers): hold `keepalive` for as long as the
    pointers are in use.
    """
    keep = [np.ascontiguousarray(a, dtype=np.float64) for a in arrays]
    return keep, [a.ctypes.data_as(_f64p) for a in keep]


def params_block(params, k):
    """The structured parameter array as the C-ABI's double[N][k] block.

    The reference's _dtype is a packed all-float64 record, so its buffer
    already is that block; this only makes it
    contiguous and reinterprets it.
    """
    p = np.ascontiguousarray(params)
    if p.dtype.itemsize != 8 * k:
        raise TypeError("parameter records must hold %d float64 fields" % k)
    flat = p.view(np.float64).reshape(-1, k)
    return flat, flat.ctypes.data_as(_f64p), flat.shape[0]
