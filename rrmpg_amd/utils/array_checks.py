"""Input coercion helpers with the reference's behaviour and messages.

Mirror of rrmpg/utils/array_checks.py (reference: check_for_negatives :15-32,
validate_array_input :35-73).  These run once per call on O(T) data and stay
on the host.
"""

import numpy as np

try:  # pandas is optional here; the reference accepts pandas.Series
    import pandas as pd
    _ARRAY_TYPES = (list, np.ndarray, pd.Series)
except Exception:  # pragma: no cover
    pd = None
    _ARRAY_TYPES = (list, np.ndarray)


def check_for_negatives(arr):
    """Return True if the array contains at least one negative number."""
    return bool(np.any(np.asarray(arr) < 0))


def validate_array_input(arr, dtype, arr_name):
    """Convert a list / numpy.ndarray / pandas.Series to a flat numpy array.

    Always returns a fresh, flattened copy of the requested dtype (so callers
    may modify it in place, e.g. HBVEdu's ``month -= 1``).

    Raises:
        ValueError: if the data is not purely numerical.
        TypeError: if arr is neither a list, a numpy.ndarray nor a
            pandas.Series.
    """
    if isinstance(arr, _ARRAY_TYPES):
        try:
            arr = np.array(arr, dtype=dtype).flatten()
        except Exception:
            raise ValueError("The data in the parameter array '{}' must be "
                             "purely numerical.".format(arr_name))
    else:
        raise TypeError("The array {} must be either a list, numpy.ndarray or "
                        "pandas.Series".format(arr_name))
    return arr
