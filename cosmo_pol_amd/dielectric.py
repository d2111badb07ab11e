"""Dielectric factor |K|^2 of liquid water (used only when the configuration
leaves radar/K_squared undefined; reference: cosmo_pol/hydrometeors/
dielectric.py:49-92, Liebe's double-Debye water model at 10 degC)."""
import numpy as np

from .constants import T_K_SQUARED


def dielectric_water(t, f):
    """Complex refractive index m of pure liquid water; t [K], f [GHz]."""
    theta = 1 - 300. / t
    eps0 = 77.66 - 103.3 * theta
    eps1 = 0.0671 * eps0
    eps2 = 3.52 + 7.52 * theta
    gamma1 = 20.20 + 146.5 * theta + 316 * theta ** 2
    gamma2 = 39.8 * gamma1
    a, b = eps0 - eps1, eps1 - eps2
    d1, d2 = 1 + (f / gamma1) ** 2, 1 + (f / gamma2) ** 2
    eps = complex(a / d1 + b / d2 + eps2, (a / d1) * (f / gamma1) + (b / d2) * (f / gamma2))
    return np.sqrt(eps)


def K_squared(frequency):
    m = dielectric_water(T_K_SQUARED, frequency)
    k = (m ** 2 - 1) / (m ** 2 + 2)
    return np.abs(k) ** 2
