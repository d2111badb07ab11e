"""Physical and microphysical constants of the forward operator.

Values are part of the numerical contract with the reference:
  global   : cosmo_pol/constants/global_constants.py:111-190
  1-moment : cosmo_pol/constants/constants_1mom.py:46-104
  2-moment : cosmo_pol/constants/constants_2mom.py:16-177
Scalar kinds (python float vs np.float64) follow upstream because NumPy-2
promotion of host-side derived tables depends on them.
"""
import numpy as np
import scipy.special as spe

EPS = np.finfo(float).eps
C_LIGHT = 299792458.
RHO_W = 1000. / (1000 ** 3)      # kg mm-3
RHO_I = 916. / (1000 ** 3)
RHO_0 = 1.225
KE = 4. / 3.
MAX_MODEL_HEIGHT = 35000
T0 = 273.15
T_K_SQUARED = 283.15
M_AIR = complex(1, 0)
SIMULATED_VARIABLES = ['ZH', 'DSPECTRUM', 'RVEL', 'ZV', 'PHIDP', 'ZDR', 'RHOHV', 'KDP']

GPM_SENSITIVITY = 12
GPM_RADIAL_RES_KA = 250
GPM_RADIAL_RES_KU = 125
GPM_NO_BINS_KA = 88 + 1
GPM_NO_BINS_KU = 176 + 1
GPM_KA_FREQUENCY = 35.6
GPM_KU_FREQUENCY = 13.6
GPM_3DB_BEAMWIDTH = 0.5


class _Table(object):
    def __repr__(self):
        return repr(self.__dict__)


def _one_moment():
    c = _Table()
    c.N0_G, c.BM_G, c.BV_G, c.MU_G, c.D_MIN_G, c.D_MAX_G = 4 * 1E3, 3.1, 0.89, 0.0, 0.2, 15
    c.AM_G = 169.6 * (1000 ** -c.BM_G)
    c.AV_G = 442.0 * (1000 ** -c.BV_G)
    c.LAMBDA_FACTOR_G = c.AM_G * c.N0_G * spe.gamma(c.BM_G + 1)
    c.VEL_FACTOR_G = spe.gamma(c.MU_G + c.BV_G + 1)
    c.NTOT_FACTOR_G = spe.gamma(c.MU_G + 1)

    c.BM_S, c.BV_S, c.MU_S, c.D_MIN_S, c.D_MAX_S = 2., 0.25, 0.0, 0.2, 20
    c.AM_S = 0.038 * (1000 ** -c.BM_S)
    c.AV_S = 4.9 * (1000 ** -c.BV_S)
    c.LAMBDA_FACTOR_S = spe.gamma(c.BM_S + 1)
    c.VEL_FACTOR_S = spe.gamma(c.MU_S + c.BV_S + 1)
    c.NTOT_FACTOR_S = spe.gamma(c.MU_S + 1)

    c.MU_R, c.BM_R, c.BV_R, c.D_MIN_R, c.D_MAX_R = 0.5, 3., 0.5, 0.1, 8
    c.N0_R = 0.1 * (8E6 / (1000 ** (1 + c.MU_R)) * (0.01) ** (-c.MU_R)) * np.exp(3.2 * c.MU_R)
    c.AM_R = np.pi / 6. * RHO_W
    c.AV_R = 130 * (1000 ** -c.BV_R)
    c.LAMBDA_FACTOR_R = c.AM_R * c.N0_R * spe.gamma(1. + c.BM_R + c.MU_R)
    c.VEL_FACTOR_R = spe.gamma(c.MU_R + c.BV_R + 1)
    c.NTOT_FACTOR_R = spe.gamma(c.MU_R + 1)

    c.BM_I, c.MU_I, c.D_MIN_I, c.D_MAX_I = 3, 0.0, 0.05, 2
    c.AM_I = 130 * (1000 ** -c.BM_I)
    c.AV_I, c.BV_I = 0.9655930341942476, 1.2019867549668874
    c.LAMBDA_FACTOR_I = spe.gamma(c.BM_I + 1)
    c.NTOT_FACTOR_I = spe.gamma(c.MU_I + 1)
    c.VEL_FACTOR_I = spe.gamma(c.MU_I + c.BV_I + 1)
    return c


# mass-based generalized-gamma parameters of the 2-moment scheme
# (D = a x^b; v = av x^bv; nu, mu), mean-mass limits and diameter range
_TWO_MOMENT_MASSIC = {
    'G': (0.142, 0.314, 86.89371, 0.268325, 1. / 3., 1.0, 1E-09, 5E-04, 0.2, 15),
    'S': (2.4, 0.455, 4.2, 0.092, 0.5, 0.0, 1E-10, 2E-05, 0.2, 20),
    'R': (0.124, 1. / 3., 114.0137, 0.23437, 1. / 3., 0.0, 2.6E-10, 3E-06, 0.2, 8),
    'H': (0.1366, 1. / 3., 39.3, 1. / 6., 1. / 3., 1.0, 2.6E-09, 5E-04, 0.2, 15),
    'I': (0.124, 0.302, 317, 0.363, 1. / 3., 0.0, 1E-12, 1E-6, 0.05, 2),
}


def _two_moment():
    c = _Table()
    for t, (am_, bm_, av_, bv_, nu_, mu_, xmin, xmax, dmin, dmax) in _TWO_MOMENT_MASSIC.items():
        bm = 1. / bm_
        am = am_ ** (-1 / bm_)
        bv = bv_ / bm_
        av = av_ * am_ ** (-bv_ / bm_)
        nu = nu_ / bm_
        mu = (mu_ + 1) / bm_ - 1
        lam_f = 1. / am * spe.gamma((mu + 1) / nu) / spe.gamma((mu + bm + 1) / nu)
        vals = dict(BM=bm, AM=am * 1000 ** (-bm), BV=bv, AV=av * 1000 ** (-bv), NU=nu, MU=mu,
                    LAMBDA_FACTOR=lam_f, VEL_FACTOR=spe.gamma((mu + bv + 1) / nu),
                    NTOT_FACTOR=spe.gamma((mu + 1) / nu), X_MIN=xmin, X_MAX=xmax,
                    D_MIN=dmin, D_MAX=dmax)
        for k, v in vals.items():
            setattr(c, k + '_' + t, v)
    c.C_1, c.C_2, c.C_3 = 9.65, 10.3, 600. * 1000. ** (-1)
    return c


C1 = _one_moment()
C2 = _two_moment()


def phi_23_ice(x):
    """Field et al. (2005) double-normalised ice PSD (constants_1mom.py:104)."""
    return (490.6 * np.exp(-20.78 * x) + 17.46 * x ** (0.6357) * np.exp(-3.290 * x))


class DerivedConstants(object):
    """Config-dependent constants (global_constants.py:166-190); rebuilt by
    RadarOperator whenever the configuration changes (reference: update())."""

    def __init__(self, config):
        r = config['radar']
        self.NVEL = C_LIGHT / (4 * 1E-6 * r['PRI'] * r['frequency'] * 1E09)
        self.VRES = 2 * self.NVEL / r['FFT_length']
        self.VARRAY = np.arange(-self.NVEL, self.NVEL + self.VRES, self.VRES)   # global_constants.py:171
        self.WAVELENGTH = C_LIGHT / (r['frequency'] * 1E09) * 1000
        self.PULSE_WIDTH = 2 * r['radial_resolution']
        self.RADAR_CONSTANT_DB = None
        sens = r.get('sensitivity')
        if isinstance(sens, (list, tuple)) and len(sens) == 3:
            self.RADAR_CONSTANT_DB = (
                180 - 10 * np.log10(np.pi ** 3 * r['3dB_beamwidth'] ** 2 * r['K_squared']
                                    * self.PULSE_WIDTH / 1000)
                - 2 * sens[1] + 10 * np.log10(1024 * np.log(2) * (self.WAVELENGTH / 1000.) ** 2))
        self.RANGE_RADAR = None
        if r.get('type', 'ground') == 'ground':
            self.RANGE_RADAR = np.arange(r['radial_resolution'] / 2., r['range'],
                                         r['radial_resolution'])
