"""Oracle particle-size distributions (TEST INFRASTRUCTURE ONLY).

Restates the PSD part of cosmo_pol/hydrometeors/hydrometeors.py that the
radar-observable path uses (set_psd / get_N / get_M / get_V / integrate_M /
integrate_V); the LUT-generation-only members (aspect ratios, canting,
dielectric mixing) are out of scope.  NumPy expressions are kept in the same
operand order and with the same scalar kinds as upstream so that the dtype
promotion (NumPy-2 rules; quirk Q11) and rounding match.

  generic N(D), V(D), M(D), integrate_V : hydrometeors.py:128-210
  2-moment set_psd                       : hydrometeors.py:212-256
  Rain / Snow / Graupel / Hail           : hydrometeors.py:627-1171
  IceParticle                            : hydrometeors.py:1177-1373
  melting particles                      : hydrometeors.py:303-478, 1393-1481
  vlinspace                              : utilities.py:158-173
"""
import numpy as np

from . import constants as K
from .constants import C1, C2


def vlinspace(a, b, N, endpoint=True):
    a, b = np.asanyarray(a), np.asanyarray(b)
    return a[..., None] + (b - a)[..., None] / (N - endpoint) * np.arange(N)


class GammaPSD(object):
    """N(D) = N0 D^mu exp(-lambda D^nu); m = a D^b; v = alpha D^beta."""
    tag = None

    def __init__(self, scheme):
        self.scheme = scheme if scheme in ('1mom', '2mom') else '1mom'
        c = C1 if self.scheme == '1mom' else C2
        t = self.tag
        self.nbins_D = 1024
        self.d_min = getattr(c, 'D_MIN_' + t)
        self.d_max = getattr(c, 'D_MAX_' + t)
        self.a = getattr(c, 'AM_' + t)
        self.b = getattr(c, 'BM_' + t)
        self.alpha = getattr(c, 'AV_' + t)
        self.beta = getattr(c, 'BV_' + t)
        self.mu = getattr(c, 'MU_' + t)
        self.nu = getattr(c, 'NU_' + t) if self.scheme == '2mom' else self._nu_1mom
        self.lambda_factor = getattr(c, 'LAMBDA_FACTOR_' + t)
        self.vel_factor = getattr(c, 'VEL_FACTOR_' + t)
        self.ntot_factor = getattr(c, 'NTOT_FACTOR_' + t)
        if self.scheme == '2mom':
            self.x_min = getattr(c, 'X_MIN_' + t)
            self.x_max = getattr(c, 'X_MAX_' + t)
        self.N0 = None
        self.lambda_ = None
        self.ntot = None

    _nu_1mom = 1

    # hydrometeors.py:128-147
    def get_N(self, D):
        if len(self.lambda_.shape) >= D.ndim:
            op = lambda x, y: np.outer(x, y)
        else:
            op = lambda x, y: np.multiply(x[:, None], y)
        if np.isscalar(self.N0):
            return self.N0 * D ** self.mu * np.exp(-op(self.lambda_, D ** self.nu))
        return op(self.N0, D ** self.mu) * np.exp(-op(self.lambda_, D ** self.nu))

    def get_V(self, D):
        return self.alpha * D ** self.beta

    def get_M(self, D):
        return self.a * D ** self.b

    # hydrometeors.py:178-199
    def integrate_V(self):
        v = (self.vel_factor * self.N0 * self.alpha / self.nu *
             self.lambda_ ** (-(self.beta + self.mu + 1) / self.nu))
        if self.scheme == '2mom':
            n = self.ntot
        else:
            n = self.ntot_factor * self.N0 / self.nu * self.lambda_ ** (-(self.mu + 1) / self.nu)
        if np.isscalar(v):
            v = np.array([v])
            n = np.array([n])
        return v, n

    # hydrometeors.py:212-256
    def set_psd_2mom(self, qn, q):
        with np.errstate(divide='ignore', invalid='ignore'):
            x_mean = np.minimum(np.maximum(q * 1.0 / (qn + K.EPS), self.x_min), self.x_max)
            if len(x_mean.shape) > 1:
                x_mean = np.squeeze(x_mean)
            lam = np.asarray((self.lambda_factor * x_mean) ** (-self.nu / self.b))
            lam[q == 0] = np.nan
            if lam.shape == ():
                lam = np.array([lam])
            n0 = np.asarray((self.nu / self.ntot_factor) * qn * lam ** ((self.mu + 1) / self.nu))
            n0 = n0 * 1000 ** (-(1 + self.mu))
            lam = lam * 1000 ** (-self.nu)
            self.N0 = n0.T
            self.lambda_ = lam.T
            self.ntot = qn

    def _finish_1mom(self, lam, q):
        lam[q == 0] = np.nan
        self.lambda_ = lam
        self.ntot = self.ntot_factor * self.N0 * self.lambda_ ** (self.mu - 1)


class Rain(GammaPSD):
    tag = 'R'
    _nu_1mom = 1.0

    def __init__(self, scheme):
        GammaPSD.__init__(self, scheme)
        if self.scheme == '1mom':
            self.N0 = C1.N0_R

    def set_psd(self, *args):        # hydrometeors.py:747-772
        if self.scheme == '2mom':
            return self.set_psd_2mom(*args)
        with np.errstate(divide='ignore', invalid='ignore'):
            lam = np.array((self.lambda_factor / args[0]) ** (1. / (4. + self.mu)))
            self._finish_1mom(lam, args[0])


class Graupel(GammaPSD):
    tag = 'G'

    def __init__(self, scheme):
        GammaPSD.__init__(self, scheme)
        self.N0 = C1.N0_G          # also in 2mom until set_psd (hydrometeors.py:1008)

    def set_psd(self, *args):        # hydrometeors.py:1025-1049 (quirk Q2: exponent 1/(4+mu))
        if self.scheme == '2mom':
            return self.set_psd_2mom(*args)
        with np.errstate(divide='ignore', invalid='ignore'):
            lam = np.array((self.lambda_factor / args[0]) ** (1. / (4. + self.mu)))
            self._finish_1mom(lam, args[0])


class Snow(GammaPSD):
    tag = 'S'

    def set_psd(self, *args):        # hydrometeors.py:879-907
        if self.scheme == '2mom':
            return self.set_psd_2mom(*args)
        self.N0 = 13.5 * (5.65 * 10 ** 5 * np.exp(-0.107 * (args[0] - 273.15))) / 1000
        with np.errstate(divide='ignore', invalid='ignore'):
            lam = np.array((self.a * self.N0 * self.lambda_factor / args[1]) ** (1. / (self.b + 1)))
            self._finish_1mom(lam, args[1])


class Hail(GammaPSD):
    tag = 'H'

    def __init__(self, scheme='2mom'):
        GammaPSD.__init__(self, '2mom')

    def set_psd(self, *args):
        return self.set_psd_2mom(*args)


class IceParticle(GammaPSD):
    tag = 'I'

    def __init__(self, scheme):
        GammaPSD.__init__(self, scheme)
        self.x_min = C2.X_MIN_I
        self.x_max = C2.X_MAX_I

    def get_N(self, D):              # hydrometeors.py:1231-1250
        if self.scheme == '1mom':
            x = self.lambda_[:, None] * D / 1000.
            return self.N0[:, None] * K.phi_23_ice(x)
        return (self.N0[:, None] * D ** self.mu * np.exp(-self.lambda_[:, None] * D ** self.nu))

    @staticmethod
    def mom_2(T, QM):                # hydrometeors.py:1277-1299
        n = 3
        T = T - K.T0
        a = 5.065339 - 0.062659 * T - 3.032362 * n + 0.029469 * T * n \
            - 0.000285 * T ** 2 + 0.312550 * n ** 2 + 0.000204 * T ** 2 * n \
            + 0.003199 * T * n ** 2 - 0.015952 * n ** 3
        a = 10 ** (a)
        b = 0.476221 - 0.015896 * T + 0.165977 * n + 0.007468 * T * n \
            - 0.000141 * T ** 2 + 0.060366 * n ** 2 + 0.000079 * T ** 2 * n \
            + 0.000594 * T * n ** 2 - 0.003577 * n ** 3
        return (QM / a) ** (1 / b)

    def set_psd(self, arg1, arg2):   # hydrometeors.py:1302-1373
        QM = arg2.astype(np.float64)
        if self.scheme == '1mom':
            T = arg1
            Q2 = self.mom_2(T, QM / C1.BM_I)
            N0 = Q2 ** ((self.b + 1) / (self.b - 2)) * QM ** ((2 + 1) / (2 - self.b))
            N0 /= 10 ** 5
            lam = (Q2 / QM) ** (1 / (self.b - 2))
            D = np.linspace(self.d_min, self.d_max, self.nbins_D)
            x = lam[:, None] * D.T / 1000
            N = N0[:, None] * K.phi_23_ice(x)
            QM_est = np.nansum(self.a * D ** self.b * N, axis=1) * (D[1] - D[0])
            N0 = N0 / QM_est * QM
            self.N0 = N0.T
            self.lambda_ = lam.T
            self.ntot = np.nansum(N, axis=1) * (D[1] - D[0])
        else:
            QN = arg1.astype(np.float64)
            with np.errstate(divide='ignore', invalid='ignore'):
                x_mean = np.minimum(np.maximum(QM * 1.0 / (QN + K.EPS), self.x_min), self.x_max)
                if len(x_mean.shape) > 1:
                    x_mean = np.squeeze(x_mean)
                lam = np.array((self.lambda_factor * x_mean) ** (-self.nu / self.b))
                lam[QM == 0] = float('nan')
                if not lam.shape:
                    lam = np.array([lam])
                N0 = np.asarray((self.nu / self.ntot_factor) * QN * lam ** ((self.mu + 1) / self.nu))
                N0 = N0 * 1000 ** (-(1 + self.mu))
                lam = lam * 1000 ** (-self.nu)
            self.N0 = N0.T
            self.lambda_ = lam.T
            self.ntot = self.ntot_factor * self.N0 * self.lambda_ ** (-self.mu - 1)

    def integrate_V(self):           # hydrometeors.py:1256-1275 (sums over ALL gates)
        D = np.linspace(self.d_min, self.d_max, self.nbins_D)
        dD = D[1] - D[0]
        N = self.get_N(D)
        v = np.sum(N * self.get_V(D)) * dD
        n = np.sum(N) * dD
        if np.isscalar(v):
            v = np.array([v])
            n = np.array([n])
        return v, n


class Melting(object):
    """Melting snow / graupel (hydrometeors.py:303-478)."""
    solid_cls = None

    def __init__(self, scheme):
        self.scheme = scheme if scheme in ('1mom', '2mom') else '1mom'
        self.nbins_D = 1024
        self.rain = Rain(self.scheme)
        self.solid = self.solid_cls(self.scheme)
        self.prop_factor = None
        self.d_min = None
        self.d_max = None

    @property
    def f_wet(self):
        return self._f_wet

    @f_wet.setter
    def f_wet(self, fw):             # hydrometeors.py:332-339
        self._f_wet = fw
        self.d_max = fw * self.rain.d_max + (1 - fw) * self.solid.d_max
        self.d_min = fw * self.rain.d_min + (1 - fw) * self.solid.d_min

    @staticmethod
    def _rowmul(x, y):
        return np.multiply(x[:, None], y)

    def get_M(self, D):              # hydrometeors.py:393-413
        M_rain = self.rain.get_M(D)
        M_dry = self.solid.get_M(D)
        op = (lambda x, y: np.outer(x, y)) if len(self.f_wet.shape) >= M_dry.ndim else self._rowmul
        return op(self.f_wet ** 2, M_rain) + op((1 - self.f_wet ** 2), M_dry)

    def _D_r(self, d):
        rho_m = self.get_M(d) / (np.pi / 6 * d ** 3)
        return (rho_m / K.RHO_W) ** (1 / 3.) * d

    def get_V(self, D):              # hydrometeors.py:415-439
        V_rain = self.rain.get_V(self._D_r(D))
        V_dry = self.solid.get_V(D)
        phi = 0.246 * self.f_wet + (1 - 0.246) * self.f_wet ** 7
        op = (lambda x, y: np.outer(x, y)) if len(self.f_wet.shape) > V_dry.ndim else self._rowmul
        return op(phi, V_rain) + op((1 - phi), V_dry)

    def get_N(self, D):              # hydrometeors.py:372-390
        if self.prop_factor is not None:
            op = (lambda x, y: np.outer(x, y)) if len(self.prop_factor.shape) > D.ndim else self._rowmul
        else:
            op = lambda x, y: y
        dDr = (self._D_r(D + 0.01) - self._D_r(D)) / 0.01
        return (op(self.prop_factor, self.rain.get_N(self._D_r(D)))
                * self.rain.get_V(self._D_r(D)) / self.get_V(D) * dDr)

    def integrate_M(self):           # hydrometeors.py:462-478
        D = vlinspace(self.d_min, self.d_max, self.nbins_D)
        dD = D[:, 1] - D[:, 0]
        if np.isscalar(self.d_min):
            return np.sum(self.get_N(D) * self.get_M(D)) * dD
        return np.sum(self.get_N(D) * self.get_M(D), axis=1) * dD

    def integrate_V(self):           # hydrometeors.py:441-460
        D = vlinspace(self.d_min, self.d_max, self.nbins_D)
        dD = D[:, 1] - D[:, 0]
        N = self.get_N(D)
        return np.sum(N * self.get_V(D), axis=1) * dD, np.sum(N, axis=1) * dD


class MeltingSnow(Melting):
    solid_cls = Snow

    def set_psd(self, T, q, fw):     # hydrometeors.py:1398-1434 (1mom)
        self.prop_factor = None
        with np.errstate(divide='ignore', invalid='ignore'):
            T = np.array(T)
            q = np.array(q)
            fw = np.array(fw)
            self.solid.set_psd(T, q)
            self.rain.set_psd(q)
            self.f_wet = fw
            self.prop_factor = q / self.integrate_M()


class MeltingGraupel(Melting):
    solid_cls = Graupel

    def set_psd(self, q, fw):        # hydrometeors.py:1446-1481 (1mom)
        self.prop_factor = None
        with np.errstate(divide='ignore', invalid='ignore'):
            q = np.array(q)
            fw = np.array(fw)
            self.solid.set_psd(q)
            self.rain.set_psd(q)
            self.f_wet = fw
            self.prop_factor = q / self.integrate_M()


def create_hydrometeor(h, scheme='1mom'):
    return {'R': Rain, 'S': Snow, 'G': Graupel, 'H': Hail, 'I': IceParticle,
            'mS': MeltingSnow, 'mG': MeltingGraupel}[h](scheme)
