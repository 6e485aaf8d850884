"""HBM-resident sweeps: the *_simulate_dev family of the C-ABI driven with
torch-owned device memory.

The model classes (rrmpg_amd.models) keep the reference's host-array seam:
numpy in, numpy out, one PCIe round trip per call.  For million-set sweeps the
[timesteps, sets] discharge array (87.7 GB at 1M x 30 yr) should not cross
PCIe at all, so this module keeps forcing, parameters, outputs and the
per-set scores on the GPU.  torch is used only as the allocator / stream
owner; every kernel is librrhip's.

One process drives one GPU; several GPUs = several processes, each with its
own shard of the parameter-set axis (rrmpg_amd.sharding).
"""

import ctypes

import numpy as np
import torch

from . import _lib


def _dev_tensor(a, device, dtype=torch.float64):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    np_dtype = {torch.float64: np.float64, torch.int8: np.int8}[dtype]
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np_dtype)).to(device)


def _ptr(t):
    return None if t is None else t.data_ptr()


class _Ensemble:
    """Common plumbing: resident forcing, workspace, stream, launch."""

    NUM_PARAMS = 0
    # GR4J-family ensembles: the largest x4 the parameter blocks of ``run``
    # may hold.  Up to 20 the unit hydrographs live on chip; set a larger
    # value and the workspace grows by the unit-hydrograph scratch the longer
    # ones run from (N * (6 ceil(x4) + 2) * 8 B; rr_*_workspace_bytes_x4).  A
    # block beyond it writes nothing and ``check()`` raises.
    max_x4 = 20.0

    def __init__(self, device):
        self.lib = _lib.load()
        _lib.require_gpu()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("rrmpg_amd.device needs a GPU device")
        self._ws = None
        self.num_timesteps = 0

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8,
                                   device=self.device)
        return self._ws

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def replica(self, device=None):
        """A second ensemble over the same forcing with a workspace of its
        own, so that both can have sweeps in flight: on this GPU (the forcing
        tensors are shared, nothing is copied) or, with `device`, on another
        GPU (the forcing -- 1.4 MB at most -- is copied there once).  What
        the shards of ``monte_carlo(sampler='device', gpus=G)`` run on."""
        import copy
        new = copy.copy(self)
        new._ws = None
        if device is not None and torch.device(device) != self.device:
            new.device = torch.device(device)
            for key, val in vars(self).items():
                if isinstance(val, torch.Tensor):
                    setattr(new, key, val.to(new.device))
        return new

    def upload_params(self, params):
        """Structured numpy parameter array (or [N, k] float array) -> device
        block double[N][k]."""
        if isinstance(params, torch.Tensor):
            return params.to(self.device, torch.float64).contiguous()
        if params.dtype.names:
            flat, _, _ = _lib.params_block(params, self.NUM_PARAMS)
        else:
            flat = np.ascontiguousarray(params, dtype=np.float64)
        return torch.from_numpy(flat.copy()).to(self.device)

    # Row pitch of the outputs handed out here, in doubles: 128 B.  A wave
    # stores 512 contiguous bytes of a row; with a dense pitch that is not a
    # multiple of 64 B every such store straddles two partly written memory
    # lines and a million-set sweep loses up to 40 % (measured:
    # profiles/r04_row_pitch.txt).  The C-ABI takes any ld >= N.
    ROW_PITCH = 16

    def new_output(self, num_sets, rows_per_t=1):
        """[T, num_sets] (or [T, rows_per_t, num_sets]) output tensor whose
        rows start on 128-byte boundaries: a view of a buffer with the row
        pitch rounded up to ``ROW_PITCH`` doubles."""
        ld = -(-max(int(num_sets), 1) // self.ROW_PITCH) * self.ROW_PITCH
        shape = ((self.num_timesteps, ld) if rows_per_t == 1 else
                 (self.num_timesteps, rows_per_t, ld))
        return torch.empty(shape, dtype=torch.float64,
                           device=self.device)[..., :num_sets]

    def _check_tensor(self, t, shape, what):
        """The kernels take raw pointers: a tensor of the wrong dtype, device,
        shape or layout would mean silent out-of-bounds HBM traffic, so every
        caller-provided tensor is checked here.  The last axis (parameter
        sets) must be dense; the row stride may exceed it (a column block of
        a wider array)."""
        if t is None:
            return
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s must be a torch tensor" % what)
        if t.dtype != torch.float64:
            raise TypeError("%s must be float64, got %s" % (what, t.dtype))
        if t.device != self.device:
            raise ValueError("%s lives on %s, the ensemble on %s"
                             % (what, t.device, self.device))
        if tuple(t.shape) != tuple(shape):
            raise ValueError("%s must have shape %s, got %s"
                             % (what, tuple(shape), tuple(t.shape)))
        if t.numel() and t.stride(-1) != 1:
            raise ValueError("%s: the parameter-set axis must be contiguous"
                             % what)

    def _check_outputs(self, n, outs2d=(), outs3d=(), layers=1):
        """2-D outputs [T, n] and 3-D storages [T, layers, n] of one call must
        share one leading dimension ld (>= n); returns it."""
        t = self.num_timesteps
        lds = []
        for k, o in enumerate(outs2d):
            self._check_tensor(o, (t, n), "output %d" % k)
            if o is not None and t > 1:
                lds.append(o.stride(0))
        for k, o in enumerate(outs3d):
            self._check_tensor(o, (t, layers, n), "storage %d" % k)
            if o is None:
                continue
            if layers > 1:
                lds.append(o.stride(1))
                if t > 1 and o.stride(0) != layers * o.stride(1):
                    raise ValueError("3-D storages must be [T][L][ld] with "
                                     "dense layer planes")
            elif t > 1:
                lds.append(o.stride(0))
        if len(set(lds)) > 1:
            raise ValueError("all outputs of one call must share the row "
                             "stride, got %s" % sorted(set(lds)))
        return int(lds[0]) if lds else n

    def _common(self, params, qobs, sse):
        if not isinstance(params, torch.Tensor) or params.dim() != 2 \
                or params.shape[1] != self.NUM_PARAMS:
            raise ValueError("params must be a tensor [N, %d]"
                             % self.NUM_PARAMS)
        n = params.shape[0]
        self._check_tensor(params, (n, self.NUM_PARAMS), "params")
        if not params.is_contiguous():
            raise ValueError("params must be contiguous (double[N][k])")
        if qobs is not None:
            self._check_tensor(qobs, (self.num_timesteps,), "qobs")
            if sse is None:
                sse = torch.empty(n, dtype=torch.float64, device=self.device)
            else:
                self._check_tensor(sse, (n,), "sse")
        return n, sse

    def _layer_shapes(self, etp=None):
        """[T, L] layer forcing of the snow models: equal shapes, etp [T]."""
        if self.prec.dim() != 2 or self.temp.shape != self.prec.shape \
                or self.frac.shape != self.prec.shape:
            raise RuntimeError("The layer arrays (precipitation, mean "
                               "temperature, solid fraction) must be "
                               "[timesteps, layers] and of the same size.")
        self.num_timesteps, self.num_layers = (int(x) for x in
                                               self.prec.shape)
        if etp is not None and (etp.dim() != 1
                                or etp.numel() != self.num_timesteps):
            raise RuntimeError("etp must hold one value per timestep.")

    def check(self):
        """GR4J-family ensembles only: wait for the sweeps enqueued so far and
        raise if the last parameter block held a set the kernels cannot run
        (ceil(x4) < 1 or NaN, or x4 > 20) -- such a sweep writes nothing.
        ``run`` itself never synchronises (rr_gr4j_plan_status)."""
        if self._ws is not None and getattr(self, "HAS_GR4J", False):
            _lib.check(self.lib.rr_gr4j_plan_status(_ptr(self._ws),
                                                    self._stream()),
                       "rr_gr4j_plan_status")


def _same_length(what, *arrays):
    n = {int(a.shape[0]) for a in arrays}
    if len(n) != 1:
        raise RuntimeError("%s must be of the same size." % what)


class HBVEduEnsemble(_Ensemble):
    """HBV-Edu over N parameter sets, everything resident in HBM
    (rr_hbvedu_simulate_dev)."""

    NUM_PARAMS = 11

    def __init__(self, temp, prec, month, PE_m, T_m, snow_init=0., soil_init=0.,
                 s1_init=0., s2_init=0., device="cuda:0"):
        """month: 1..12 as in HBVEdu.simulate; it is decremented here."""
        super().__init__(device)
        self.temp = _dev_tensor(temp, self.device)
        self.prec = _dev_tensor(prec, self.device)
        month0 = np.asarray(month, dtype=np.int64) - 1
        if month0.min() < 0 or month0.max() > 11:
            raise ValueError("The month array must be between an integer1 "
                             "(Jan) and 12 (Dec).")
        self.month0 = _dev_tensor(month0.astype(np.int8), self.device,
                                  torch.int8)
        self.PE_m = _dev_tensor(PE_m, self.device)
        self.T_m = _dev_tensor(T_m, self.device)
        self.inits = tuple(float(v) for v in (snow_init, soil_init, s1_init,
                                              s2_init))
        self.num_timesteps = int(self.prec.numel())
        _same_length("The arrays of the temperature, precipitation and month",
                     self.temp.reshape(-1), self.prec.reshape(-1),
                     self.month0.reshape(-1))
        if self.PE_m.numel() != 12 or self.T_m.numel() != 12:
            raise RuntimeError("The monthly arrays must be of length 12.")

    def run(self, params, qsim=None, storages=None, qobs=None, sse=None):
        """Enqueue one sweep on the current stream (asynchronous).

        params: device tensor [N, 11]; qsim: optional [T, N] output tensor;
        storages: optional 4-tuple of [T, N] tensors; qobs: optional [T]
        device tensor -> returns the per-set squared-error sums [N].
        """
        n, sse = self._common(params, qobs, sse)
        t = self.num_timesteps
        wsb = self.lib.rr_hbvedu_workspace_bytes(t, n)
        ws = self._workspace(wsb)
        st = tuple(storages) if storages else (None,) * 4
        ld = self._check_outputs(n, (qsim,) + st)
        rc = self.lib.rr_hbvedu_simulate_dev(
            _ptr(self.temp), _ptr(self.prec), _ptr(self.month0),
            _ptr(self.PE_m), _ptr(self.T_m), t, *self.inits, _ptr(params), n,
            _ptr(qsim), *[_ptr(x) for x in st], ld, _ptr(qobs),
            _ptr(sse) if qobs is not None else None, _ptr(ws), wsb,
            self._stream())
        _lib.check(rc, "rr_hbvedu_simulate_dev")
        return sse if qobs is not None else None


class ABCEnsemble(_Ensemble):
    """ABC model over N parameter sets (rr_abc_simulate_dev)."""

    NUM_PARAMS = 3

    def __init__(self, prec, initial_state=0., device="cuda:0"):
        super().__init__(device)
        self.prec = _dev_tensor(prec, self.device)
        self.initial_state = float(initial_state)
        self.num_timesteps = int(self.prec.numel())

    def run(self, params, qsim=None, storage=None, qobs=None, sse=None):
        n, sse = self._common(params, qobs, sse)
        t = self.num_timesteps
        wsb = self.lib.rr_abc_workspace_bytes(t, n)
        ws = self._workspace(wsb)
        ld = self._check_outputs(n, (qsim, storage))
        rc = self.lib.rr_abc_simulate_dev(
            _ptr(self.prec), t, self.initial_state, _ptr(params), n,
            _ptr(qsim), _ptr(storage), ld, _ptr(qobs),
            _ptr(sse) if qobs is not None else None, _ptr(ws), wsb,
            self._stream())
        _lib.check(rc, "rr_abc_simulate_dev")
        return sse if qobs is not None else None


class GR4JEnsemble(_Ensemble):
    """GR4J over N parameter sets (rr_gr4j_simulate_dev)."""

    NUM_PARAMS = 4
    HAS_GR4J = True

    def __init__(self, prec, etp, s_init=0., r_init=0., device="cuda:0"):
        super().__init__(device)
        self.prec = _dev_tensor(prec, self.device)
        self.etp = _dev_tensor(etp, self.device)
        self.inits = (float(s_init), float(r_init))
        self.num_timesteps = int(self.prec.numel())
        _same_length("The arrays of precipitation and evapotranspiration",
                     self.prec.reshape(-1), self.etp.reshape(-1))

    def run(self, params, qsim=None, storages=None, qobs=None, sse=None):
        """Enqueue one sweep on the current stream; never synchronises.  The
        unit-hydrograph storage (registers for ceil(x4) <= 3 / 5 / 10, LDS up
        to x4 = 20, an HBM scratch up to ``self.max_x4`` beyond) is chosen on
        the GPU; a block with an unusable x4 leaves the outputs untouched --
        ``check()`` reports it."""
        n, sse = self._common(params, qobs, sse)
        t = self.num_timesteps
        wsb = self.lib.rr_gr4j_workspace_bytes_x4(t, n, float(self.max_x4))
        ws = self._workspace(wsb)
        st = tuple(storages) if storages else (None,) * 2
        ld = self._check_outputs(n, (qsim,) + st)
        rc = self.lib.rr_gr4j_simulate_dev(
            _ptr(self.prec), _ptr(self.etp), t, *self.inits, _ptr(params), n,
            _ptr(qsim), *[_ptr(x) for x in st], ld, _ptr(qobs),
            _ptr(sse) if qobs is not None else None, _ptr(ws), wsb,
            self._stream())
        _lib.check(rc, "rr_gr4j_simulate_dev")
        return sse if qobs is not None else None


class CemaneigeEnsemble(_Ensemble):
    """Cemaneige snow routine over N parameter sets
    (rr_cemaneige_simulate_dev).  Forcing: [T, L] layer arrays as produced by
    rrmpg_amd.models.cemaneige.prepare_snow_inputs."""

    NUM_PARAMS = 2

    def __init__(self, layer_prec, layer_mean_temp, frac_solid_prec,
                 snow_pack_init=0., thermal_state_init=0., device="cuda:0"):
        super().__init__(device)
        self.prec = _dev_tensor(layer_prec, self.device)
        self.temp = _dev_tensor(layer_mean_temp, self.device)
        self.frac = _dev_tensor(frac_solid_prec, self.device)
        self.inits = (float(snow_pack_init), float(thermal_state_init))
        self._layer_shapes()

    def run(self, params, outflow=None, storages=None, qobs=None, sse=None):
        n, sse = self._common(params, qobs, sse)
        t, nl = self.num_timesteps, self.num_layers
        wsb = self.lib.rr_cemaneige_workspace_bytes(t, nl, n)
        ws = self._workspace(wsb)
        st = tuple(storages) if storages else (None,) * 2
        ld = self._check_outputs(n, (outflow,), st, nl)
        rc = self.lib.rr_cemaneige_simulate_dev(
            _ptr(self.prec), _ptr(self.temp), _ptr(self.frac), t, nl,
            *self.inits, _ptr(params), n, _ptr(outflow),
            *[_ptr(x) for x in st], ld, _ptr(qobs),
            _ptr(sse) if qobs is not None else None, _ptr(ws), wsb,
            self._stream())
        _lib.check(rc, "rr_cemaneige_simulate_dev")
        return sse if qobs is not None else None


class CemaneigeGR4JEnsemble(_Ensemble):
    """Fused Cemaneige -> GR4J over N parameter sets
    (rr_cemaneigegr4j_simulate_dev)."""

    NUM_PARAMS = 6
    HAS_GR4J = True

    def __init__(self, layer_prec, layer_mean_temp, frac_solid_prec, etp,
                 snow_pack_init=0., thermal_state_init=0., s_init=0.,
                 r_init=0., device="cuda:0"):
        super().__init__(device)
        self.prec = _dev_tensor(layer_prec, self.device)
        self.temp = _dev_tensor(layer_mean_temp, self.device)
        self.frac = _dev_tensor(frac_solid_prec, self.device)
        self.etp = _dev_tensor(etp, self.device)
        self.inits = tuple(float(v) for v in (snow_pack_init,
                                              thermal_state_init, s_init,
                                              r_init))
        self._layer_shapes(self.etp)

    def run(self, params, qsim=None, storages=None, qobs=None, sse=None):
        """storages: optional (G, eTG, s_store, r_store).  Asynchronous; see
        GR4JEnsemble.run for the x4 rule and ``check()``."""
        n, sse = self._common(params, qobs, sse)
        t, nl = self.num_timesteps, self.num_layers
        wsb = self.lib.rr_cemaneigegr4j_workspace_bytes_x4(
            t, nl, n, float(self.max_x4))
        ws = self._workspace(wsb)
        st = tuple(storages) if storages else (None,) * 4
        ld = self._check_outputs(n, (qsim, st[2], st[3]), st[:2], nl)
        rc = self.lib.rr_cemaneigegr4j_simulate_dev(
            _ptr(self.prec), _ptr(self.temp), _ptr(self.etp), _ptr(self.frac),
            t, nl, *self.inits, _ptr(params), n, _ptr(qsim),
            *[_ptr(x) for x in st], ld, _ptr(qobs),
            _ptr(sse) if qobs is not None else None, _ptr(ws), wsb,
            self._stream())
        _lib.check(rc, "rr_cemaneigegr4j_simulate_dev")
        return sse if qobs is not None else None


class HBVEduCatchments(_Ensemble):
    """HBV-Edu for C independent catchments x N parameter sets each, one
    launch (rr_hbvedu_simulate_catchments_dev; BASELINE.json configs[4]).

    Forcing arrays are [C, T] (month 1..12), monthly tables [C, 12], inits
    [C, 4] = (snow, soil, s1, s2); params [C, N, 11]; outputs [C, T, N];
    qobs [C, T] -> sse [C, N].
    """

    NUM_PARAMS = 11

    def __init__(self, temp, prec, month, PE_m, T_m, inits, device="cuda:0"):
        super().__init__(device)
        self.temp = _dev_tensor(np.atleast_2d(temp), self.device)
        self.prec = _dev_tensor(np.atleast_2d(prec), self.device)
        month0 = np.atleast_2d(np.asarray(month, dtype=np.int64)) - 1
        if month0.min() < 0 or month0.max() > 11:
            raise ValueError("The month array must be between an integer1 "
                             "(Jan) and 12 (Dec).")
        self.month0 = _dev_tensor(month0.astype(np.int8), self.device,
                                  torch.int8)
        self.PE_m = _dev_tensor(np.atleast_2d(PE_m), self.device)
        self.T_m = _dev_tensor(np.atleast_2d(T_m), self.device)
        self.inits = _dev_tensor(np.atleast_2d(inits), self.device)
        self.num_catchments, self.num_timesteps = (int(x) for x in
                                                   self.prec.shape)
        c = self.num_catchments
        if (self.temp.shape != self.prec.shape
                or self.month0.shape != self.prec.shape
                or self.PE_m.shape != (c, 12) or self.T_m.shape != (c, 12)
                or self.inits.shape != (c, 4)):
            raise ValueError("inconsistent multi-catchment forcing shapes")

    def new_output(self, num_sets):
        return torch.empty((self.num_catchments, self.num_timesteps, num_sets),
                           dtype=torch.float64, device=self.device)

    def run(self, params, qsim=None, storages=None, qobs=None, sse=None):
        """params: device tensor [C, N, 11]."""
        c, t = self.num_catchments, self.num_timesteps
        if params.dim() != 3 or params.shape[0] != c or params.shape[2] != 11:
            raise ValueError("params must be [C, N, 11]")
        n = params.shape[1]
        self._check_tensor(params, (c, n, 11), "params")
        if not params.is_contiguous():
            raise ValueError("params must be contiguous (double[C][N][11])")
        if qobs is not None:
            self._check_tensor(qobs, (c, t), "qobs")
            if not qobs.is_contiguous():
                raise ValueError("qobs must be contiguous")
            if sse is None:
                sse = torch.empty((c, n), dtype=torch.float64,
                                  device=self.device)
            else:
                self._check_tensor(sse, (c, n), "sse")
        for k, o in enumerate((qsim,) + (tuple(storages) if storages
                                         else ())):
            self._check_tensor(o, (c, t, n), "output %d" % k)
            if o is not None and not o.is_contiguous():
                raise ValueError("multi-catchment outputs must be contiguous "
                                 "[C, T, N]")
        wsb = self.lib.rr_hbvedu_catchments_workspace_bytes(t, c, n)
        ws = self._workspace(wsb)
        st = storages or (None,) * 4
        rc = self.lib.rr_hbvedu_simulate_catchments_dev(
            _ptr(self.temp), _ptr(self.prec), _ptr(self.month0),
            _ptr(self.PE_m), _ptr(self.T_m), t, c, _ptr(self.inits),
            _ptr(params), n, _ptr(qsim), *[_ptr(x) for x in st], n,
            _ptr(qobs), _ptr(sse) if qobs is not None else None, _ptr(ws), wsb,
            self._stream())
        _lib.check(rc, "rr_hbvedu_simulate_catchments_dev")
        return sse if qobs is not None else None


class SnowGR4JEnsemble(_Ensemble):
    """Next-tier couplings over N parameter sets, resident in HBM:
    hyst=True, ice=False -> CemaneigeHystGR4J
    hyst=False, ice=True -> CemaneigeGR4JIce
    hyst=True, ice=True  -> CemaneigeHystGR4JIce
    (rr_cemaneigehystgr4j_simulate_dev & co.)."""

    HAS_GR4J = True

    def __init__(self, hyst, ice, layer_prec, layer_mean_temp, frac_solid_prec,
                 etp, frac_ice=None, snow_pack_init=0., thermal_state_init=0.,
                 sca_init=0., s_init=0., r_init=0., device="cuda:0"):
        super().__init__(device)
        if not (hyst or ice):
            raise ValueError("use CemaneigeGR4JEnsemble for the plain model")
        self.hyst, self.ice = bool(hyst), bool(ice)
        self.NUM_PARAMS = 6 + (2 if hyst else 0) + (1 if ice else 0)
        self.prec = _dev_tensor(layer_prec, self.device)
        self.temp = _dev_tensor(layer_mean_temp, self.device)
        self.frac = _dev_tensor(frac_solid_prec, self.device)
        self.etp = _dev_tensor(etp, self.device)
        self.frac_ice = (_dev_tensor(frac_ice, self.device) if ice else None)
        self.inits = tuple(float(v) for v in (snow_pack_init,
                                              thermal_state_init, sca_init,
                                              s_init, r_init))
        self._layer_shapes(self.etp)
        if ice and self.frac_ice.numel() != self.num_layers:
            raise ValueError("frac_ice must hold one value per elevation "
                             "layer.")

    def storage_names(self):
        """The state series this variant can materialise, in the order of the
        reference's return tuple."""
        names = ["G", "eTG", "s_store", "r_store"]
        if self.hyst:
            names.append("sca")
        if self.ice:
            names.append("icemelt")
        if self.hyst and self.ice:
            names.append("snowmelt")
        return names

    def new_storages(self, num_sets):
        """Device tensors for every state series of ``storage_names()``:
        G, eTG, sca are [T, L, N], the others [T, N]."""
        return {k: self.new_output(num_sets, self.num_layers
                                   if k in ("G", "eTG", "sca") else 1)
                for k in self.storage_names()}

    def run(self, params, qsim=None, qobs=None, sse=None, storages=None):
        """Discharge, fused per-set squared error and/or (storages: the dict
        of ``new_storages``, all of them or none) every state series.
        Asynchronous; see GR4JEnsemble.run for the x4 rule and ``check()``."""
        n, sse = self._common(params, qobs, sse)
        t, nl = self.num_timesteps, self.num_layers
        wsb = self.lib.rr_snowgr4j_workspace_bytes_x4(t, nl, n,
                                                      float(self.max_x4))
        ws = self._workspace(wsb)
        st = storages or {}
        if st and sorted(st) != sorted(self.storage_names()):
            raise ValueError("pass all of %s or none" % self.storage_names())
        two = [qsim] + [st.get(k) for k in ("s_store", "r_store", "icemelt",
                                            "snowmelt")]
        three = [st.get(k) for k in ("G", "eTG", "sca")]
        ld = self._check_outputs(n, two, three, nl)
        sse_p = _ptr(sse) if qobs is not None else None
        i = self.inits
        tail = (ld, _ptr(qobs), sse_p, _ptr(ws), wsb, self._stream())
        g = lambda k: _ptr(st.get(k))               # noqa: E731
        if self.hyst and self.ice:
            rc = self.lib.rr_cemaneigehystgr4jice_simulate_dev(
                _ptr(self.prec), _ptr(self.temp), _ptr(self.etp),
                _ptr(self.frac_ice), _ptr(self.frac), t, nl, *i, _ptr(params),
                n, _ptr(qsim), g("G"), g("eTG"), g("s_store"), g("r_store"),
                g("sca"), g("icemelt"), g("snowmelt"), *tail)
        elif self.hyst:
            rc = self.lib.rr_cemaneigehystgr4j_simulate_dev(
                _ptr(self.prec), _ptr(self.temp), _ptr(self.etp),
                _ptr(self.frac), t, nl, *i, _ptr(params), n, _ptr(qsim),
                g("G"), g("eTG"), g("s_store"), g("r_store"), g("sca"), *tail)
        else:
            rc = self.lib.rr_cemaneigegr4jice_simulate_dev(
                _ptr(self.prec), _ptr(self.temp), _ptr(self.etp),
                _ptr(self.frac_ice), _ptr(self.frac), t, nl, i[0], i[1], i[3],
                i[4], _ptr(params), n, _ptr(qsim), g("G"), g("eTG"),
                g("s_store"), g("r_store"), g("icemelt"), *tail)
        _lib.check(rc, "rr_snowgr4j_simulate_dev")
        return sse if qobs is not None else None


def snow_layers(prec, mean_temp, min_temp, max_temp, met_station_height,
                altitudes=(), device="cuda:0", numpy_exp=True):
    """Station series -> the [T, L] layer forcing of the Cemaneige family,
    computed on the GPU (rr_cemaneige_layers_dev): the resident counterpart of
    the host preprocessing in rrmpg_amd.models.cemaneige_utils (reference:
    rrmpg/models/cemaneige_utils.py).  Inputs: numpy arrays or device
    tensors [T].  Returns device tensors (layer_prec, layer_mean_temp,
    frac_solid_prec), ready for CemaneigeEnsemble & co.

    numpy_exp=True hands the L precipitation factors over as numpy computes
    them, so the result equals the host preprocessing bit for bit; False
    lets the library use the C library's exp (as numba does).
    """
    lib = _lib.load()
    _lib.require_gpu()
    dev = torch.device(device)
    series = [_dev_tensor(a, dev).reshape(-1) for a in (prec, mean_temp,
                                                         min_temp, max_temp)]
    t = int(series[0].numel())
    if any(int(a.numel()) != t for a in series):
        raise RuntimeError("All meteorological input arrays must have the "
                           "same length.")
    alts = np.asarray(altitudes if len(altitudes) else [met_station_height],
                      dtype=np.float64)
    nl = int(alts.size)
    factor = None
    if numpy_exp:
        factor = np.where(
            alts <= 4000, np.exp((alts - met_station_height) * 0.0004),
            np.exp((4000 - met_station_height) * 0.0004)
            if met_station_height <= 4000 else 1.0).astype(np.float64)
    outs = [torch.empty((t, nl), dtype=torch.float64, device=dev)
            for _ in range(3)]
    wsb = lib.rr_cemaneige_layers_workspace_bytes(nl)
    ws = torch.empty(int(wsb), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rr_cemaneige_layers_dev(
            *[_ptr(a) for a in series], t, _lib.f64(alts)[1], nl,
            float(met_station_height),
            None if factor is None else _lib.f64(factor)[1],
            *[_ptr(o) for o in outs], _ptr(ws), wsb,
            torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "rr_cemaneige_layers_dev")
    if t:
        torch.cuda.current_stream(dev).synchronize()   # ws, factors go away
    return tuple(outs)


def column_sums(qsim, obs, shift=0.0):
    """Per-column {sum q, sum q^2, sum q*obs, sum (obs-q)^2} of a resident
    discharge tensor qsim [T, N] against obs [T] (both on the GPU), in one
    HBM-bandwidth-bound pass (rr_column_sums_shifted_dev).  Feed the result
    (moved to the host) to rrmpg_amd.utils.metrics.scores_from_sums with the
    same `shift`: the first three sums are taken of q - shift and obs - shift,
    and shift = mean(obs) keeps the variance / correlation scores (KGE,
    alpha, r) free of cancellation for large, nearly constant series."""
    lib = _lib.load()
    if qsim.dim() != 2 or qsim.stride(1) != 1 or qsim.dtype != torch.float64:
        raise ValueError("qsim must be a float64 [T, N] tensor with "
                         "contiguous columns")
    t, n = qsim.shape
    obs = obs.to(qsim.device, torch.float64).contiguous()
    if obs.numel() != t:
        raise ValueError("Arrays must have the same size.")
    sums = torch.empty((n, 4), dtype=torch.float64, device=qsim.device)
    rc = lib.rr_column_sums_shifted_dev(
        qsim.data_ptr(), qsim.stride(0), obs.data_ptr(), t, n, float(shift),
        sums.data_ptr(), torch.cuda.current_stream(qsim.device).cuda_stream)
    _lib.check(rc, "rr_column_sums_shifted_dev")
    return sums


def sample_params(model, num, key, n_total=None, first=0, device=None):
    """Draw `num` parameter sets of `model` inside its default bounds directly
    in HBM (rr_sample_params_dev) -- the resident counterpart of
    ``model.get_random_params(num)`` (reference: rrmpg/models/basemodel.py:
    68-91, ABC rule abcmodel.py:70-103) without the N x k x 8 B upload.

    Returns a float64 tensor [num, k] in _param_list order, ready for the
    ensembles' ``run``.  The draws are rows ``first .. first+num-1`` of the
    population a host reproduces with::

        rng = numpy.random.Generator(numpy.random.Philox(key=key))
        cols = [rng.uniform(lo, hi, size=n_total) for each parameter
                in the model's draw order]

    (draw order = _param_list order; for the ABC model a, c, then b with the
    upper bound 1 - a, as the reference draws them).  ``host_population`` below
    is exactly that restatement.
    """
    lib = _lib.load()
    _lib.require_gpu()
    names = list(model._param_list)
    k = len(names)
    n_total = num if n_total is None else int(n_total)
    lo = np.array([model._default_bounds[p][0] for p in names], np.float64)
    hi = np.array([model._default_bounds[p][1] for p in names], np.float64)
    pos, one_minus = _draw_plan(model)
    dev = torch.device("cuda", torch.cuda.current_device()) \
        if device is None else torch.device(device)
    out = torch.empty((num, k), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rr_sample_params_dev(
            int(key), k, _lib.f64(lo)[1], _lib.f64(hi)[1],
            pos.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), one_minus,
            n_total, int(first), int(num), out.data_ptr(),
            torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "rr_sample_params_dev")
    return out


def _draw_plan(model):
    """(rank of each parameter in the draw order, index of the parameter
    whose upper bound is 1 - first parameter or 0)."""
    names = list(model._param_list)
    if names == ["a", "b", "c"]:            # ABC: a, c, then b | a
        return np.array([0, 2, 1], dtype=np.intc), 1
    return np.arange(len(names), dtype=np.intc), 0


def host_population(model, n_total, key):
    """The population ``sample_params`` draws from, built on the host with
    numpy's Philox generator -- float64 array [n_total, k]."""
    names = list(model._param_list)
    pos, one_minus = _draw_plan(model)
    rng = np.random.Generator(np.random.Philox(key=int(key)))
    out = np.empty((n_total, len(names)))
    for rank in range(len(names)):
        j = int(np.nonzero(pos == rank)[0][0])
        lo, hi = model._default_bounds[names[j]]
        if one_minus and j == one_minus:
            hi = 1 - out[:, 0]
        out[:, j] = rng.uniform(lo, hi, size=n_total)
    return out
