"""Oracle constants (TEST INFRASTRUCTURE ONLY).

Restates the numerical constants of the reference that are part of the
numerical contract:
  global   : cosmo_pol/constants/global_constants.py:111-190
  1-moment : cosmo_pol/constants/constants_1mom.py:46-104
  2-moment : cosmo_pol/constants/constants_2mom.py:16-177
Types matter (NumPy-2 promotion: python floats are "weak", np.float64 scalars
are not), so every derived constant is formed with the same kind of scalar
(python float vs np.float64 via scipy.special.gamma / np.exp) as upstream.
"""
import numpy as np
import scipy.special as spe

EPS = np.finfo(float).eps
C_LIGHT = 299792458.
RHO_W = 1000. / (1000 ** 3)
RHO_I = 916. / (1000 ** 3)
RHO_0 = 1.225
KE = 4. / 3.
MAX_MODEL_HEIGHT = 35000
T0 = 273.15
T_K_SQUARED = 283.15
SIMULATED_VARIABLES = ['ZH', 'DSPECTRUM', 'RVEL', 'ZV', 'PHIDP', 'ZDR', 'RHOHV', 'KDP']

GPM_SENSITIVITY = 12
GPM_RADIAL_RES_KA = 250
GPM_RADIAL_RES_KU = 125
GPM_KA_FREQUENCY = 35.6
GPM_KU_FREQUENCY = 13.6
GPM_3DB_BEAMWIDTH = 0.5


class Derived(object):
    """Config-dependent constants (global_constants.py:166-190)."""

    def __init__(self, config):
        r = config['radar']
        self.WAVELENGTH = C_LIGHT / (r['frequency'] * 1E09) * 1000
        self.PULSE_WIDTH = 2 * r['radial_resolution']
        self.RADAR_CONSTANT_DB = None
        sens = r.get('sensitivity')
        if isinstance(sens, (list, tuple)) and len(sens) == 3:
            self.RADAR_CONSTANT_DB = (180 - 10 * np.log10(np.pi ** 3 * r['3dB_beamwidth'] ** 2
                                      * r['K_squared'] * self.PULSE_WIDTH / 1000)
                                      - 2 * sens[1]
                                      + 10 * np.log10(1024 * np.log(2) * (self.WAVELENGTH / 1000.) ** 2))
        if r.get('type', 'ground') == 'ground':
            self.RANGE_RADAR = np.arange(r['radial_resolution'] / 2., r['range'],
                                         r['radial_resolution'])


class _NS(object):
    pass


def _build_1mom():
    c = _NS()
    # graupel (constants_1mom.py:46-57)
    c.N0_G = 4 * 1E3
    c.BM_G = 3.1
    c.BV_G = 0.89
    c.AM_G = 169.6 * (1000 ** -c.BM_G)
    c.AV_G = 442.0 * (1000 ** -c.BV_G)
    c.MU_G = 0.0
    c.D_MIN_G = 0.2
    c.D_MAX_G = 15
    c.LAMBDA_FACTOR_G = c.AM_G * c.N0_G * spe.gamma(c.BM_G + 1)
    c.VEL_FACTOR_G = spe.gamma(c.MU_G + c.BV_G + 1)
    c.NTOT_FACTOR_G = spe.gamma(c.MU_G + 1)
    # snow (:59-69)
    c.BM_S = 2.
    c.BV_S = 0.25
    c.AM_S = 0.038 * (1000 ** -c.BM_S)
    c.AV_S = 4.9 * (1000 ** -c.BV_S)
    c.MU_S = 0.0
    c.D_MIN_S = 0.2
    c.D_MAX_S = 20
    c.LAMBDA_FACTOR_S = spe.gamma(c.BM_S + 1)
    c.VEL_FACTOR_S = spe.gamma(c.MU_S + c.BV_S + 1)
    c.NTOT_FACTOR_S = spe.gamma(c.MU_S + 1)
    # rain (:71-85)
    c.MU_R = 0.5
    n00 = 8E6 / (1000 ** (1 + c.MU_R)) * (0.01) ** (-c.MU_R)
    c.N0_R = 0.1 * n00 * np.exp(3.2 * c.MU_R)
    c.BM_R = 3.
    c.BV_R = 0.5
    c.AM_R = np.pi / 6. * RHO_W
    c.AV_R = 130 * (1000 ** -c.BV_R)
    c.D_MIN_R = 0.1
    c.D_MAX_R = 8
    c.LAMBDA_FACTOR_R = c.AM_R * c.N0_R * spe.gamma(1. + c.BM_R + c.MU_R)
    c.VEL_FACTOR_R = spe.gamma(c.MU_R + c.BV_R + 1)
    c.NTOT_FACTOR_R = spe.gamma(c.MU_R + 1)
    # ice crystals (:87-100)
    c.BM_I = 3
    c.AM_I = 130 * (1000 ** -c.BM_I)
    c.MU_I = 0.0
    c.D_MIN_I = 0.05
    c.D_MAX_I = 2
    c.AV_I = 0.9655930341942476
    c.BV_I = 1.2019867549668874
    c.LAMBDA_FACTOR_I = spe.gamma(c.BM_I + 1)
    c.NTOT_FACTOR_I = spe.gamma(c.MU_I + 1)
    c.VEL_FACTOR_I = spe.gamma(c.MU_I + c.BV_I + 1)
    return c


def phi_23_ice(x):
    """Field et al. (2005) double-normalised PSD, moments 2-3
    (constants_1mom.py:104)."""
    return (490.6 * np.exp(-20.78 * x) + 17.46 * x ** (0.6357) * np.exp(-3.290 * x))


def _massic_to_diam(c, tag, am_, bm_, av_, bv_, nu_, mu_, xmin, xmax, dmin, dmax):
    """constants_2mom.py: conversion of the mass-based generalized gamma
    parameters to diameter-based ones (same statement order)."""
    bm = 1. / bm_
    am = am_ ** (-1 / bm_)
    bv = bv_ / bm_
    av = av_ * am_ ** (-bv_ / bm_)
    nu = nu_ / bm_
    mu = (mu_ + 1) / bm_ - 1
    lf = 1. / am * spe.gamma((mu + 1) / nu) / spe.gamma((mu + bm + 1) / nu)
    am = am * 1000 ** (-bm)
    av = av * 1000 ** (-bv)
    vf = spe.gamma((mu + bv + 1) / nu)
    nf = spe.gamma((mu + 1) / nu)
    for k, v in dict(BM=bm, AM=am, BV=bv, AV=av, NU=nu, MU=mu, LAMBDA_FACTOR=lf,
                     VEL_FACTOR=vf, NTOT_FACTOR=nf, X_MIN=xmin, X_MAX=xmax,
                     D_MIN=dmin, D_MAX=dmax).items():
        setattr(c, k + '_' + tag, v)


def _build_2mom():
    c = _NS()
    _massic_to_diam(c, 'G', 0.142, 0.314, 86.89371, 0.268325, 1. / 3., 1.0, 1E-09, 5E-04, 0.2, 15)
    _massic_to_diam(c, 'S', 2.4, 0.455, 4.2, 0.092, 0.5, 0.0, 1E-10, 2E-05, 0.2, 20)
    _massic_to_diam(c, 'R', 0.124, 1. / 3., 114.0137, 0.23437, 1. / 3., 0.0, 2.6E-10, 3E-06, 0.2, 8)
    _massic_to_diam(c, 'H', 0.1366, 1. / 3., 39.3, 1. / 6., 1. / 3., 1.0, 2.6E-09, 5E-04, 0.2, 15)
    _massic_to_diam(c, 'I', 0.124, 0.302, 317, 0.363, 1. / 3., 0.0, 1E-12, 1E-6, 0.05, 2)
    c.C_1 = 9.65
    c.C_2 = 10.3
    c.C_3 = 600. * 1000. ** (-1)
    return c


C1 = _build_1mom()
C2 = _build_2mom()
