"""Forcing preprocessing for the Cemaneige snow routine (host side).

These three functions are parameter-independent and O(T*L) once per
``simulate`` call, so they stay on the host as vectorised numpy (SURVEY.md
section 8a row A11); their [T, L] outputs are uploaded once and shared by
every parameter set on the device.

Same names, arguments and results as the reference's
rrmpg/models/cemaneige_utils.py (calculate_solid_fraction :15-98,
extrapolate_precipitation :100-158, extrapolate_temperature :160-207).
"""

import numpy as np


def calculate_solid_fraction(prec, altitudes, mean_temp, min_temp, max_temp):
    """Fraction of solid precipitation per timestep and elevation layer.

    Args:
        prec: [t, n] precipitation per elevation layer (shape only is used).
        altitudes: [n] median elevation of each layer.
        mean_temp, min_temp, max_temp: [t, n] daily temperatures per layer.

    Returns:
        [t, n] array with the fraction of solid precipitation.

    Layers below 1500 m use the min/max-temperature rule, layers at or above
    1500 m the mean-temperature rule (reference: cemaneige_utils.py:52-96).
    """
    altitudes = np.asarray(altitudes, dtype=np.float64)
    mean_temp = np.asarray(mean_temp, dtype=np.float64)
    min_temp = np.asarray(min_temp, dtype=np.float64)
    max_temp = np.asarray(max_temp, dtype=np.float64)
    z_thresh = 1500
    num_timesteps, num_layers = np.shape(prec)[0], len(altitudes)
    solid_fraction = np.zeros((num_timesteps, num_layers), dtype=np.float64)

    low = altitudes < z_thresh
    with np.errstate(divide="ignore", invalid="ignore"):
        # < 1500 m: 1 if max <= 0, 0 if min >= 0, else 1 - max / (max - min)
        between = 1 - (max_temp / (max_temp - min_temp))
        low_frac = np.where(max_temp <= 0, 1.0,
                            np.where(min_temp >= 0, 0.0, between))
        # >= 1500 m: 0 if mean >= 3, 1 if mean <= 0, else 1 - (mean + 1) / 4
        high_frac = np.where(mean_temp >= 3, 0.0,
                             np.where(mean_temp <= 0, 1.0,
                                      1 - (mean_temp + 1) / 4))
    solid_fraction[:, low] = low_frac[:, low]
    solid_fraction[:, ~low] = high_frac[:, ~low]
    return solid_fraction


def extrapolate_precipitation(prec, altitudes, met_station_height):
    """Extrapolate station precipitation to the layer elevations.

    Exponential gradient of 0.0004 / m, capped at 4000 m
    (reference: cemaneige_utils.py:120-156).

    Returns:
        [t, n] precipitation per elevation layer.
    """
    prec = np.asarray(prec, dtype=np.float64)
    altitudes = np.asarray(altitudes, dtype=np.float64)
    beta_altitude = 0.0004
    z_thresh = 4000
    layer_prec = np.zeros((prec.shape[0], len(altitudes)), dtype=np.float64)
    for l, alt in enumerate(altitudes):
        if alt <= z_thresh:
            layer_prec[:, l] = prec * np.exp((alt - met_station_height)
                                             * beta_altitude)
        elif met_station_height <= z_thresh:
            layer_prec[:, l] = prec * np.exp((z_thresh - met_station_height)
                                             * beta_altitude)
        else:
            layer_prec[:, l] = prec
    return layer_prec


def extrapolate_temperature(min_temp, mean_temp, max_temp, altitudes,
                            met_station_height):
    """Extrapolate station temperatures to the layer elevations.

    Linear lapse rate of -0.0065 K / m (reference: cemaneige_utils.py:185-205).

    Returns:
        layer_min_temp, layer_mean_temp, layer_max_temp: [t, n] arrays.
    """
    min_temp = np.asarray(min_temp, dtype=np.float64)
    mean_temp = np.asarray(mean_temp, dtype=np.float64)
    max_temp = np.asarray(max_temp, dtype=np.float64)
    altitudes = np.asarray(altitudes, dtype=np.float64)
    theta_temp = -0.0065
    delta_temp = (altitudes - met_station_height) * theta_temp
    return (min_temp[:, None] + delta_temp[None, :],
            mean_temp[:, None] + delta_temp[None, :],
            max_temp[:, None] + delta_temp[None, :])
