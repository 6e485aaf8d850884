"""Parent class of all rainfall-runoff models of rrmpg_amd.models.

Same public surface as the reference's BaseModel (reference:
rrmpg/models/basemodel.py:20-175): parameter list / default bounds / custom
dtype as class attributes, random sampling, get/set of parameters.  On top of
that it holds the two helpers every model's ``simulate`` shares here: turning
the ``params`` argument into the C-ABI's parameter block and calling the
batched GPU entry point once for all parameter sets.
"""

import numbers

import numpy as np

from numpy.random import uniform

from .. import _lib


class BaseModel(object):
    """Basic model class for all rainfall-runoff models."""

    # List of strings containing all model parameters
    _param_list = []

    # Dict containing the default parameter bounds
    _default_bounds = {}

    # Custom numpy datatype: packed float64 record in _param_list order.  Its
    # buffer is handed to the GPU library unchanged (double[N][k]).
    _dtype = np.dtype([])

    def __init__(self, params=None):
        """Initialize a new hydrological model.

        Args:
            params: (optional) Dictionary containing all model parameters as
                separate key/value pairs.  Random parameters within the
                default bounds are generated if nothing is passed.

        Raises:
            AttributeError: If a model parameter is missing in the passed
                dictionary.
        """
        if params:
            missings = [p for p in self._param_list if p not in params.keys()]
            if len(missings) > 0:
                raise AttributeError("Missing the following model parameters: "
                                     "{}".format(missings))
        else:
            params = self.get_random_params()
        self.set_params(params)

    def get_random_params(self, num=1):
        """Generate random sets of model parameters in the default bounds.

        One ``numpy.random.uniform(size=num)`` draw per parameter in
        _param_list order, so ``np.random.seed(s)`` gives the same sets as the
        reference (basemodel.py:83-91).

        Returns:
            A numpy array of the model's custom dtype with num entries.
        """
        params = np.zeros(num, dtype=self._dtype)
        for param in self._param_list:
            low, high = self._default_bounds[param]
            params[param] = uniform(low=low, high=high, size=num)
        return params

    def get_params(self):
        """Return a dict with all model parameters and their current value."""
        return {param: getattr(self, param) for param in self._param_list}

    def set_params(self, params):
        """Set model parameters to values passed in params.

        Args:
            params: Either a dictionary of parameter name/value pairs (one,
                many or all parameters) or a numpy array / record of the
                model's own custom dtype.

        Raises:
            ValueError: If any parameter is not a numerical value.
            AttributeError: If a key matches none of the parameter names.
            TypeError: If a numpy array does not have the model's dtype or
                the input is neither a dict nor a numpy.ndarray.
        """
        if isinstance(params, dict):
            for param, value in params.items():
                if param not in self._param_list:
                    raise AttributeError(
                        "Unknow parameter '{}'.Name must match one of the "
                        "model parameters.Use {}.get_parameter_names() to get "
                        "a list of valid names.".format(
                            param, self.__class__.__name__))
                if not isinstance(value, numbers.Number):
                    raise ValueError("The value of parameter '{}'must be "
                                     "numerical".format(param))
                setattr(self, param, value)
        elif isinstance(params, (np.void, np.ndarray)):
            if params.dtype != self._dtype:
                raise TypeError("The parameter array has the wrong data type. "
                                "It must be the custom data type of the "
                                "model.")
            for param in self._param_list:
                value = params[param]
                setattr(self, param,
                        value if isinstance(params, np.void) else value[0])
        else:
            raise TypeError("Wrong input data type. Must be either a dict or "
                            "a numpy.ndarray")

    def get_parameter_names(self):
        """Return the list of parameter names."""
        return self._param_list

    def get_default_bounds(self):
        """Return the dictionary containing the default parameter bounds."""
        return self._default_bounds

    def get_dtype(self):
        """Return the custom model datatype."""
        return self._dtype

    # ------------------------------------------------------------------
    # shared plumbing of the models' simulate() / _loss()
    # ------------------------------------------------------------------
    def _resolve_params(self, params):
        """The `params` argument of simulate() as a 1-D structured array.

        None -> the parameters stored in the model object; a single record
        (numpy.void) -> one-element array (reference: hbvedu.py:173-188).
        """
        if params is None:
            params = np.zeros(1, dtype=self._dtype)
            for param in self._param_list:
                params[param] = getattr(self, param)
        else:
            if not hasattr(params, "dtype") or params.dtype != self._dtype:
                raise TypeError("The model parameters must be a numpy array "
                                "of the models own custom data type.")
            if isinstance(params, np.void):
                params = np.expand_dims(params, params.ndim)
        return params

    def _sweep(self, params, qobs, want_qsim, **kwargs):
        """What rrmpg_amd.tools.monte_carlo runs for every model: simulate all
        parameter sets, return (qsim or None, per-set squared-error sums or
        None).  This generic form goes through ``simulate`` exactly as the
        reference's monte_carlo does (monte_carlo.py:64) and scores the
        returned array; the model classes override it with a call that
        accumulates the squared errors inside the GPU kernel."""
        qsim = self.simulate(params=params, **kwargs)
        if isinstance(qsim, tuple):          # return_storage(s)=True was set
            qsim = qsim[0]
        sse = None
        if qobs is not None:
            if qsim.shape[0] != len(qobs):
                raise ValueError("Arrays must have the same size.")
            d = np.asarray(qobs, dtype=np.float64)[:, None] - qsim
            sse = np.einsum("tn,tn->n", d, d)
        return (qsim if want_qsim else None), sse

    @classmethod
    def _params_from_population(cls, X):
        """Parameter records from the optimiser's candidates.

        X: one candidate vector [k] -> 1 record, or a population [k, S] (scipy
        differential_evolution with vectorized=True) -> S records.
        """
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = X[:, None]
        params = np.zeros(X.shape[1], dtype=cls._dtype)
        for row, name in zip(X, cls._param_list):
            params[name] = row
        return params

    def _differential_evolution(self, loss, args, batched):
        """scipy's differential evolution over the default bounds.

        batched=False (the default of every ``fit``) is the reference's call
        exactly (one candidate per loss evaluation, updating='immediate';
        e.g. reference hbvedu.py:305): the same optimiser trajectory as the
        reference for a seeded run, at the reference's speed.
        batched=True (opt-in): scipy gets a vectorised loss, so each
        generation's whole population is ONE GPU sweep (updating='deferred').
        A GPU runs a single candidate no faster than numba does (the time
        loop is serial: ~1 ms for ten years either way), but a population of
        165 in the same ~1 ms.
        """
        from scipy import optimize
        bnds = tuple([self._default_bounds[p] for p in self._param_list])
        if batched:
            return optimize.differential_evolution(
                loss, bounds=bnds, args=args, vectorized=True,
                updating='deferred')
        return optimize.differential_evolution(loss, bounds=bnds, args=args)


def new_outputs(shape, wanted):
    """Zero-initialised output arrays (None where not requested)."""
    return [np.zeros(shape, np.float64) if w else None for w in wanted]


def out_ptr(arr):
    return None if arr is None else arr.ctypes.data_as(_lib._f64p)
