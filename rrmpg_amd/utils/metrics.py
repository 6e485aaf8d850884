"""Evaluation metrics used on the hot path.

Mirror of the functions of rrmpg/utils/metrics.py that the Monte-Carlo sweep
and the optimiser's loss use (reference: calc_nse :29-77, calc_rmse :81-107,
calc_mse :110-136).  The per-set squared-error sums of a sweep are produced
on the GPU (the kernels' fused `sse` output); `mse_from_sse` / `nse_from_sse`
turn them into the same scores without touching qsim.
"""

import numpy as np

from .array_checks import validate_array_input


def _pair(obs, sim):
    obs = validate_array_input(obs, np.float64, 'obs')
    sim = validate_array_input(sim, np.float64, 'sim')
    if len(obs) != len(sim):
        raise ValueError("Arrays must have the same size.")
    return obs, sim


def _nse_denominator(obs):
    denominator = np.sum((obs - np.mean(obs)) ** 2)
    if denominator == 0:
        raise RuntimeError(
            "The Nash-Sutcliffe-Efficiency coefficient is not defined for the "
            "case, that all values in the observations are equal. Maybe you "
            "should use the Mean-Squared-Error instead.")
    return denominator


def calc_nse(obs, sim):
    """Nash-Sutcliffe model efficiency: 1 - sum((sim-obs)^2)/sum((obs-mean)^2).

    Raises:
        ValueError: arrays of unequal size or non-numeric values.
        TypeError: unsupported array type.
        RuntimeError: all observations equal (NSE undefined).
    """
    obs, sim = _pair(obs, sim)
    denominator = _nse_denominator(obs)
    numerator = np.sum((sim - obs) ** 2)
    return 1 - numerator / denominator


def calc_rmse(obs, sim):
    """Root mean squared error."""
    obs, sim = _pair(obs, sim)
    return np.sqrt(np.mean((obs - sim) ** 2))


def calc_mse(obs, sim):
    """Mean squared error."""
    obs, sim = _pair(obs, sim)
    return np.mean((obs - sim) ** 2)


def mse_from_sse(sse, num_timesteps):
    """Per-set MSE from the kernels' fused squared-error sums."""
    return np.asarray(sse, dtype=np.float64) / num_timesteps


def nse_from_sse(sse, obs):
    """Per-set NSE from the kernels' fused squared-error sums."""
    obs = validate_array_input(obs, np.float64, 'obs')
    return 1 - np.asarray(sse, dtype=np.float64) / _nse_denominator(obs)


# --- the remaining scores of the reference's metrics module (used by the
# hysteresis models' loss_metric="kge"; reference: calc_kge :139-188,
# calc_alpha_nse :191-232, calc_beta_nse :235-281, calc_r :284-299)
def _std_obs(obs, msg):
    std_obs = np.std(obs)
    if std_obs == 0:
        raise RuntimeError(msg)
    return std_obs


def calc_kge(obs, sim):
    """Kling-Gupta-Efficiency (Gupta et al. 2009):
    1 - sqrt((r-1)^2 + (alpha-1)^2 + (beta-1)^2).

    Raises:
        ValueError / TypeError: as calc_mse.
        RuntimeError: if the mean or the standard deviation of the
            observations equals 0.
    """
    from scipy.stats import pearsonr
    obs, sim = _pair(obs, sim)
    mean_obs = np.mean(obs)
    if mean_obs == 0:
        raise RuntimeError("KGE not definied if the mean of the observations "
                           "equals 0.")
    std_obs = _std_obs(obs, "KGE not definied if the standard deviation of "
                            "the observations equals 0.")
    r = pearsonr(obs, sim)[0]
    alpha = np.std(sim) / std_obs
    beta = np.mean(sim) / mean_obs
    return 1 - np.sqrt((r - 1) ** 2 + (alpha - 1) ** 2 + (beta - 1) ** 2)


def calc_alpha_nse(obs, sim):
    """Alpha decomposition of the NSE: std(sim) / std(obs)."""
    obs, sim = _pair(obs, sim)
    return np.std(sim) / _std_obs(obs, "Not definied if the standard "
                                       "deviation of the observations equals "
                                       "0.")


def calc_beta_nse(obs, sim):
    """Beta decomposition of the NSE: (mean(sim) - mean(obs)) / std(obs)."""
    obs, sim = _pair(obs, sim)
    std_obs = _std_obs(obs, "Not definied if the standard deviation of the "
                            "observations equals 0.")
    mean_obs = np.mean(obs)
    if mean_obs == 0:
        raise RuntimeError("Not definied if the mean of the observations "
                           "equals 0.")
    return (np.mean(sim) - mean_obs) / std_obs


def calc_r(obs, sim):
    """Pearson r (scipy.stats.pearsonr result, as the reference returns it)."""
    from scipy.stats import pearsonr
    obs, sim = _pair(obs, sim)
    return pearsonr(obs, sim)


ALL_SCORES = ("mse", "rmse", "nse", "kge", "alpha", "beta", "r")


def scores_from_sums(sums, obs, only=None, shift=0.0):
    """Every score of this module for N simulated series at once, from the
    per-column sums the GPU produces in one pass (rrmpg_amd.device.
    column_sums: {sum q, sum q^2, sum q*obs, sum (obs-q)^2} per column, the
    first three about `shift` -- pass the value given to column_sums).

    Returns a dict of arrays [N]: mse, rmse, nse, kge, alpha, beta, r -- or
    only those named in `only`.  Same definitions as calc_mse / calc_rmse /
    calc_nse / calc_kge / calc_alpha_nse / calc_beta_nse / calc_r, and, like
    them, a score raises RuntimeError for observations it is not defined for
    (NSE: all equal; KGE / alpha / beta: standard deviation or mean 0) only
    when it is asked for: `only=("mse",)` accepts any observations, as
    calc_mse does (reference: rrmpg/utils/metrics.py:110-136).
    """
    obs = validate_array_input(obs, np.float64, 'obs')
    sums = np.asarray(sums, dtype=np.float64).reshape(-1, 4)
    want = ALL_SCORES if only is None else tuple(only)
    unknown = set(want) - set(ALL_SCORES)
    if unknown:
        raise ValueError("unknown score(s): %s" % sorted(unknown))
    t = obs.size
    s_q, s_qq, s_qo, s_dd = sums.T
    out = {}
    mse = s_dd / t
    if "mse" in want:
        out["mse"] = mse
    if "rmse" in want:
        out["rmse"] = np.sqrt(mse)
    if "nse" in want:
        out["nse"] = 1 - s_dd / _nse_denominator(obs)
    moments = [k for k in ("kge", "alpha", "beta", "r") if k in want]
    if not moments:
        return out
    mean_obs, std_obs = np.mean(obs), np.std(obs)
    if mean_obs == 0 and ("kge" in want or "beta" in want):
        raise RuntimeError("KGE not definied if the mean of the observations "
                           "equals 0.")
    if std_obs == 0 and ("kge" in want or "alpha" in want or "beta" in want):
        raise RuntimeError("KGE not definied if the standard deviation of "
                           "the observations equals 0.")
    # moments about `shift` (c): mean and covariance move with it, the
    # variance does not
    mean_qc = s_q / t                          # mean(q) - c
    mean_oc = mean_obs - shift
    mean_q = mean_qc + shift
    var_q = np.maximum(s_qq / t - mean_qc ** 2, 0.0)
    std_q = np.sqrt(var_q)
    cov = s_qo / t - mean_qc * mean_oc
    with np.errstate(divide="ignore", invalid="ignore"):
        r = cov / (std_q * std_obs)
        alpha = std_q / std_obs
        if "kge" in want:
            out["kge"] = 1 - np.sqrt((r - 1) ** 2 + (alpha - 1) ** 2
                                     + (mean_q / mean_obs - 1) ** 2)
        if "alpha" in want:
            out["alpha"] = alpha
        if "beta" in want:
            out["beta"] = (mean_q - mean_obs) / std_obs
        if "r" in want:
            out["r"] = r
    return out
