"""Interface to the ABC-Model (GPU ensemble engine).

Same class surface as the reference's rrmpg/models/abcmodel.py (ABCModel
:25-232, _loss :235-255); ``simulate`` evaluates ALL parameter sets with one
call into librrhip (rr_abc_simulate) instead of a Python loop over run_abcmodel.
"""

import numbers

import numpy as np

from .. import _lib
from ..utils.array_checks import check_for_negatives, validate_array_input
from .basemodel import BaseModel, new_outputs, out_ptr


class ABCModel(BaseModel):
    """Interface to the ABC-Model.

    Classical linear educational model (Fiering, "Streamflow synthesis",
    Harvard University Press, 1967).  If no model parameters are passed upon
    initialization, a random parameter set is generated.

    Beyond the reference's interface (every default is the reference's
    behaviour): ``simulate(params=...)`` runs any number of parameter sets in
    ONE GPU call; ``fit(batched=True)`` evaluates a whole
    differential-evolution generation per GPU sweep (about 100 times faster,
    another optimiser trajectory than the reference's sequential search, which
    stays the default); ``rrmpg_amd.tools.monte_carlo(model, num, qobs,
    return_qsim=False, sampler='device', gpus=G)`` draws, sweeps and scores
    the sets in the GPUs' memory.

    Args:
        params: (optional) Dictionary containing all model parameters as
            separate key/value pairs.
    """

    _param_list = ['a', 'b', 'c']

    _default_bounds = {'a': (0, 1),
                       'b': (0, 1),
                       'c': (0, 1)}

    _dtype = np.dtype([('a', np.float64),
                       ('b', np.float64),
                       ('c', np.float64)])

    def __init__(self, params=None):
        super().__init__(params=params)

    def get_random_params(self, num=1):
        """Generate random parameter sets that satisfy a + b <= 1.

        'a' and 'c' are drawn first (num values each), then one 'b' per set
        from U(0, 1 - a) -- the same stream of numpy.random draws as the
        reference (abcmodel.py:85-103), so a seeded run gives identical sets.
        """
        params = np.zeros(num, dtype=self._dtype)
        bnds = self._default_bounds
        params['a'][:] = np.random.uniform(low=bnds['a'][0],
                                           high=bnds['a'][1], size=num)
        params['c'][:] = np.random.uniform(low=bnds['c'][0],
                                           high=bnds['c'][1], size=num)
        # one draw per set, upper bound 1 - a[i]; a single broadcast call
        # consumes the generator exactly like num scalar calls
        params['b'][:] = np.random.uniform(low=bnds['b'][0],
                                           high=(1 - params['a']), size=num)
        return params

    def simulate(self, prec, initial_state=0, return_storage=False,
                 params=None):
        """Simulate the streamflow for the passed precipitation.

        Args:
            prec: Precipitation for each timestep (list, numpy array or
                pandas.Series).
            initial_state: (optional) Initial value for the storage.
            return_storage: (optional) Boolean, whether to return the
                simulated storage for each timestep as well.
            params: (optional) Numpy array of parameter sets of the model's
                custom dtype; all are evaluated at once on the GPU.  Defaults
                to the parameters stored in the model object.

        Returns:
            qsim [timesteps, sets], and optionally storage of the same shape.

        Raises:
            ValueError: If one of the inputs contains invalid values.
            TypeError: If one of the inputs has an incorrect datatype.
        """
        prec, initial_state = _validate(prec, initial_state)
        if not isinstance(return_storage, bool):
            raise TypeError("The return_storage arg must be a boolean.")
        params = self._resolve_params(params)
        qsim, storage, _ = _run(prec, initial_state, params, True,
                                return_storage, None)
        if return_storage:
            return qsim, storage
        return qsim

    def fit(self, qobs, prec, initial_state=0, batched=False):
        """Fit the model to a timeseries of discharge.

        Uses scipy's differential evolution, as the reference does
        (abcmodel.py:188-232); every candidate is one GPU call that returns
        only its squared-error sum.

        batched (extension; default False): False is the reference's own
        call -- one candidate per loss evaluation, immediate updating --
        which reproduces its seeded runs evaluation by evaluation
        (tests/test_gpu_fit_reference.py); True: one GPU sweep per generation
        -- a DIFFERENT optimiser trajectory than the reference's (a seeded
        fit ends in other, equally good parameters), a hundred times faster.

        Returns:
            res: A scipy OptimizeResult class object.
        """
        qobs = validate_array_input(qobs, np.float64, 'qobs')
        prec, initial_state = _validate(prec, initial_state)
        args = (prec, initial_state, qobs, self._dtype)
        return self._differential_evolution(_loss, args, batched)

    # used by rrmpg_amd.tools.monte_carlo: qsim and/or fused per-set SSE
    def _sweep(self, params, qobs, want_qsim, prec, initial_state=0):
        prec, initial_state = _validate(prec, initial_state)
        params = self._resolve_params(params)
        qsim, _, sse = _run(prec, initial_state, params, want_qsim, False,
                            qobs)
        return qsim, sse


    def _resident(self, prec, initial_state=0, device=None):
        """simulate()'s forcing as an HBM-resident ensemble
        (rrmpg_amd.device.ABCEnsemble) after simulate()'s own checks."""
        from .. import device as rrdev
        prec, initial_state = _validate(prec, initial_state)
        return rrdev.ABCEnsemble(
            prec, initial_state,
            **({} if device is None else {"device": device}))


def _validate(prec, initial_state):
    prec = validate_array_input(prec, np.float64, 'precipitation')
    if check_for_negatives(prec):
        raise ValueError("In the precipitation array are negative values.")
    if not isinstance(initial_state, numbers.Number) or initial_state < 0:
        raise TypeError("The variable 'initial_state' must be a numercial "
                        "scaler greate than 0.")
    return prec, float(initial_state)


def _run(prec, initial_state, params, want_qsim, want_storage, qobs):
    """One batched GPU call (include/rrhip.h: rr_abc_simulate)."""
    lib = _lib.load()
    _lib.require_gpu()
    block, p_ptr, n = _lib.params_block(params, 3)
    t = prec.shape[0]
    qsim, storage = new_outputs((t, n), (want_qsim, want_storage))
    qobs_arr, qobs_ptr = _lib.f64(qobs)
    sse = np.zeros(n) if qobs is not None else None
    if qobs is not None and qobs_arr.shape[0] != t:
        raise ValueError("Arrays must have the same size.")
    keep, (prec_ptr,) = _lib.f64s(prec)
    rc = lib.rr_abc_simulate_opt(prec_ptr, t, initial_state, p_ptr, n,
                             out_ptr(qsim), out_ptr(storage), qobs_ptr,
                             out_ptr(sse),
        _lib.opts_ptr())
    del keep
    _lib.check(rc, "rr_abc_simulate")
    return qsim, storage, sse


def _loss(X, *args):
    """Return the loss value (MSE) for the current parameter set."""
    prec, initial_state, qobs, dtype = args
    params = ABCModel._params_from_population(X)
    _, _, sse = _run(prec, initial_state, params, False, False, qobs)
    mse = sse / prec.shape[0]
    return mse if np.ndim(X) == 2 else mse[0]
