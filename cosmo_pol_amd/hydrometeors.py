"""Host-side hydrometeor descriptions staged next to each scattering table.

The per-gate microphysics (PSD parameters, N(D), the PSD integral) runs in the
HIP kernels of csrc/cpol_psd.inl.  What is evaluated here, once per staged
table, are the per-diameter-bin factors that do not depend on the gate, with
the very NumPy expressions (operand kinds and order) the reference evaluates
per call, so that dtype promotion and rounding are identical:

  D**mu (x N0 for fixed intercepts), D**nu   hydrometeors/hydrometeors.py:143-147
  ice normalisation grid, a*D**b             hydrometeors/hydrometeors.py:1331-1337
  axes limits / steps of the table           lookup/lut.py:193-213, 336-341
  hydrometeor list and its order             scatter/doppler_scatter.py:99-106
  d_min / d_max taken from the table axis    scatter/doppler_scatter.py:116-122
"""
import numpy as np

from . import _native as N
from . import constants as K

BASE_VARIABLES = ['U', 'V', 'W', 'QR_v', 'QS_v', 'QG_v', 'QI_v', 'RHO', 'T']
BASE_VARIABLES_2MOM = ['QH_v', 'QNH_v', 'QNR_v', 'QNS_v', 'QNG_v', 'QNI_v']


def hydrometeor_list(config):
    """Order of doppler_scatter.py:99-106 (melting species right after R,S,G)."""
    mp = config['microphysics']
    h = ['R', 'S', 'G']
    if mp['with_melting']:
        h += ['mS', 'mG']
    if mp['scheme'] == '2mom':
        h += ['H']
    if mp['with_ice_crystals']:
        h += ['I']
    return h


def variable_list(config):
    v = list(BASE_VARIABLES)
    if config['microphysics']['scheme'] == '2mom':
        v += BASE_VARIABLES_2MOM
    return v


def _consts(h, scheme):
    c = K.C1 if (scheme == '1mom' and h != 'H') else K.C2
    g = lambda name: getattr(c, name + '_' + h)
    nu = g('NU') if c is K.C2 else (1.0 if h == 'R' else 1)
    return dict(a=g('AM'), b=g('BM'), alpha=g('AV'), beta=g('BV'), mu=g('MU'), nu=nu,
                lambda_factor=g('LAMBDA_FACTOR'), ntot_factor=g('NTOT_FACTOR'),
                vel_factor=g('VEL_FACTOR'), d_min=g('D_MIN'), d_max=g('D_MAX'),
                x_min=getattr(c, 'X_MIN_' + h, 0.0), x_max=getattr(c, 'X_MAX_' + h, 0.0))


def build_hydro(h, scheme, lut, var_index):
    """-> (HydroDesc, table float64 [n_e,n_t,n_d,12], pre, dnu, aux)"""
    d = N.HydroDesc()
    table = np.ascontiguousarray(lut.value_table, dtype=np.float64)
    n_e, n_t, n_d, n_c = table.shape
    if n_c != N.N_SZ:
        raise ValueError('lookup table of %s has %d columns, expected 12' % (h, n_c))
    d.n_e, d.n_t, d.n_d = n_e, n_t, n_d
    ax_e = lut.axes_names['e']
    second = 'wc' if h in ('mS', 'mG') else 't'
    ax_t = lut.axes_names[second]
    d.e_lo = np.float32(lut.axes_limits[ax_e][0])
    d.e_step = np.float32(lut.axes_step[ax_e])
    d.t_lo = np.float32(lut.axes_limits[ax_t][0])
    d.t_step = np.float32(lut.axes_step[ax_t])
    d.second_axis_f64 = 1 if h in ('mS', 'mG') else 0
    d.var_t = var_index['T']
    d.var_qn = -1
    d.var_q = -1
    pre = dnu = aux = None

    if h in ('mS', 'mG'):
        solid = 'S' if h == 'mS' else 'G'
        if scheme != '1mom':
            raise NotImplementedError('melting hydrometeors exist in the 1-moment scheme only')
        cs, cr = _consts(solid, '1mom'), _consts('R', '1mom')
        d.psd_family = N.PSD_MELTING
        d.rule = N.RULE_MELTING_SNOW if h == 'mS' else N.RULE_MELTING_GRAUPEL
        d.solid_rule = N.RULE_SNOW_1MOM if h == 'mS' else N.RULE_GRAUPEL_1MOM
        d.q_source = N.Q_MELT_SNOW if h == 'mS' else N.Q_MELT_GRAUPEL
        d.a, d.b, d.alpha, d.beta = cs['a'], cs['b'], cs['alpha'], cs['beta']
        d.mu, d.nu = cs['mu'], float(cs['nu'])
        d.lambda_factor = cs['lambda_factor']
        d.lam_exponent = (1. / (cs['b'] + 1)) if solid == 'S' else (1. / (4. + cs['mu']))
        d.n0_fixed = K.C1.N0_G if solid == 'G' else 0.0
        d.r_a, d.r_b, d.r_alpha, d.r_beta = cr['a'], cr['b'], cr['alpha'], cr['beta']
        d.r_n0, d.r_mu = K.C1.N0_R, cr['mu']
        d.r_lambda_factor = cr['lambda_factor']
        d.r_lam_exponent = 1. / (4. + cr['mu'])
        # the melting classes own fresh Rain/Snow/Graupel partners whose d_min/d_max
        # are the module constants, not the table axis (hydrometeors.py:325-339)
        d.r_dmin, d.r_dmax = cr['d_min'], cr['d_max']
        d.s_dmin, d.s_dmax = cs['d_min'], cs['d_max']
        d.dD = 0.0
        aux = None
        if MELT_TABLES:
            aux = melting_fw_tables(d, n_d, n_t, float(d.t_lo), float(d.t_step))
            d.tab_degree = N.MELT_DEGREE if aux is not None else 0
        return d, table, None, None, aux

    c = _consts(h, scheme)
    D = np.asarray(lut.axes[lut.axes_names['d']])          # float32 [n_d]
    if D.ndim != 1 or D.shape[0] != n_d:
        raise ValueError('diameter axis of %s has shape %s' % (h, D.shape))
    d.dD = float(D[1] - D[0])                               # np.float32 difference
    d.a, d.b, d.alpha, d.beta = c['a'], c['b'], c['alpha'], c['beta']
    d.mu, d.nu = c['mu'], float(c['nu'])
    d.lambda_factor, d.ntot_factor, d.vel_factor = (c['lambda_factor'], c['ntot_factor'],
                                                    c['vel_factor'])
    d.q_source = N.Q_MODEL
    d.var_q = var_index['Q' + h + '_v']

    if scheme == '1mom' and h == 'I':
        d.psd_family = N.PSD_ICE_FIELD
        d.rule = N.RULE_ICE_1MOM
        # normalisation grid: np.linspace(d_min, d_max, nbins) with d_min/d_max the
        # float32 end points of the table axis (doppler_scatter.py:117-122)
        Dn = np.linspace(D[0], D[-1], n_d)
        aDb = c['a'] * Dn ** c['b']
        dDn = Dn[1] - Dn[0]
        Vn = c['alpha'] * Dn ** c['beta']                # get_V on the same grid (:1267-1270)
        aux = np.concatenate([D.astype(np.float64), np.asarray(Dn, dtype=np.float64),
                              np.asarray(aDb, dtype=np.float64), [np.float64(dDn)],
                              np.asarray(Vn, dtype=np.float64)])
        fast = _ice_recurrence_aux(d, D, Dn, aDb, Vn, n_d)
        if fast is not None:
            aux = np.concatenate([aux, fast])
            tabs = ice_norm_tables(Dn, aDb, Vn) if ICE_TABLES else None
            if tabs is not None:
                aux = np.concatenate([aux, tabs])
                d.tab_degree = N.ICE_DEGREE
        return d, table, None, None, aux

    d.psd_family = N.PSD_GAMMA
    mu, nu = c['mu'], c['nu']
    dnu = np.asarray(D ** nu, dtype=np.float64)
    if scheme == '2mom':
        d.rule = N.RULE_TWO_MOMENT
        d.var_qn = var_index['QN' + h + '_v']
        d.x_min, d.x_max = c['x_min'], c['x_max']
        d.lam_exponent = -nu / c['b']
        d.n0_exponent = (mu + 1) / nu
        d.c_n0 = 1000 ** (-(1 + mu))
        d.c_lam = 1000 ** (-nu)
        pre = np.asarray(D ** mu, dtype=np.float64)
    elif h == 'R':
        d.rule = N.RULE_RAIN_1MOM
        d.lam_exponent = 1. / (4. + mu)
        d.n0_fixed = K.C1.N0_R
        pre = np.asarray(K.C1.N0_R * D ** mu, dtype=np.float64)     # np.float64 * float32
    elif h == 'G':
        d.rule = N.RULE_GRAUPEL_1MOM
        d.lam_exponent = 1. / (4. + mu)                               # quirk Q2
        d.n0_fixed = K.C1.N0_G
        pre = np.asarray(K.C1.N0_G * D ** mu, dtype=np.float64)     # python float * float32
    elif h == 'S':
        d.rule = N.RULE_SNOW_1MOM
        d.lam_exponent = 1. / (c['b'] + 1)
        pre = np.asarray(D ** mu, dtype=np.float64)
    else:
        raise ValueError('hydrometeor %s does not exist in the %s scheme' % (h, scheme))
    if scheme == '2mom' and h == 'I':
        # IceParticle.integrate_V is numeric in both schemes (hydrometeors.py:1256-1275)
        Dn = np.linspace(D[0], D[-1], n_d)
        d.numeric_intv, d.uniform_grid = 1, 0
        aux = np.concatenate([np.asarray(Dn ** mu, dtype=np.float64),
                              np.asarray(Dn ** nu, dtype=np.float64),
                              np.asarray(c['alpha'] * Dn ** c['beta'], dtype=np.float64),
                              [np.float64(Dn[1] - Dn[0])]])
        return d, table, pre, dnu, aux
    aux = _uniform_grid_aux(d, D, nu, n_d, pre)
    return d, table, pre, dnu, aux


PSD_WAVES = 8        # csrc/cpol_psd.inl: CPOL_PSD_WAVES (bins are split over 8 wavefronts)
PSD_WAVES_U = 8      # csrc/cpol_psd.inl: CPOL_PSD_WAVES_U (recurrence flavour; 8 or 10)


def _uniform_grid_aux(d, D, nu, n_d, pre):
    """nu == 1: exp(-lambda D_k) is advanced by a geometric recurrence inside each
    wave chunk of ceil(n_d/10) bins.  aux = [h, dlt_0 .. dlt_{n_d-1}, (pre_k, pre_k dlt_k,
    pre_k dlt_k^2 / 2, -pre_k dlt_k^3 / 6) ...] with h the mean grid step and dlt_k = D_k - D_k0 - (k-k0) h
    the (float32-rounding sized) departure of node k from the uniform grid anchored at
    its chunk start k0; the per-bin quadruple is one 32-byte scalar load in the kernel."""
    if float(nu) != 1.0:
        d.uniform_grid = 0
        return None
    D64 = np.asarray(D, dtype=np.float64)
    h = (D64[-1] - D64[0]) / (n_d - 1)
    ch = -(-n_d // PSD_WAVES_U)             # the kernel's chunk: ceil(n_d / waves)
    k = np.arange(n_d)
    k0 = (k // ch) * ch
    dlt = D64 - D64[k0] - (k - k0) * h
    # the second-order correction is accurate to ~|lambda*dlt|^3/6: require a
    # grid that really is uniform up to float32 rounding
    if np.max(np.abs(dlt)) > 4e-6 * max(1.0, float(np.max(np.abs(D64)))):
        d.uniform_grid = 0
        return None
    d.uniform_grid = 1
    p64 = np.asarray(pre, dtype=np.float64)
    quads = np.stack([p64, p64 * dlt, p64 * dlt * dlt / 2.0, -p64 * dlt ** 3 / 6.0], axis=1).ravel()
    return np.concatenate([[h], dlt, quads]).astype(np.float64)


def spectrum_tables(h, scheme, lut):
    """Doppler scheme 3 (get_doppler_spectrum, doppler_scatter.py:672-695): float32 radar
    cross sections rcs = 2 pi (Z11 - Z12 - Z21 + Z22) of every table slice and bin, and
    the float32 grid D = np.linspace(d_min, d_max, n_d) with D**mu, D**nu evaluated as
    NumPy does on float32 arrays.  Returns (rcs32 [n_e, n_t, n_d], dgrid [3, n_d])."""
    t = np.asarray(lut.value_table, dtype=np.float64)
    rcs = (2 * np.pi * (t[..., 0] - t[..., 1] - t[..., 2] + t[..., 3])).astype(np.float32)
    if h in ('mS', 'mG'):
        # melting species: the cross sections of every (elevation, wet fraction) slice alone -- the diameter grid and N(D)
        # belong to the wet fraction of the gate and are evaluated by the kernel (cpol_spectrum.inl)
        return np.ascontiguousarray(rcs), np.zeros((3, t.shape[2]), dtype=np.float32)
    c = _consts(h, scheme)
    d_ax = np.asarray(lut.axes[lut.axes_names['d']])
    D = np.zeros(t.shape[2], dtype='float32')
    D[:] = np.linspace(d_ax[0], d_ax[-1], t.shape[2])
    with np.errstate(divide='ignore'):
        dgrid = np.stack([D, (D ** c['mu']).astype(np.float32), (D ** c['nu']).astype(np.float32)])
    return np.ascontiguousarray(rcs), np.ascontiguousarray(dgrid, dtype=np.float32)


ICE_POWER = 0.6357          # exponent of the second term of PHI_23_I (constants_1mom.py)


def _ice_recurrence_aux(d, D, Dn, aDb, Vn, n_d):
    """1-moment ice: PHI_23(x) = 490.6 exp(-20.78 x) + 17.46 x^0.6357 exp(-3.29 x) with
    x = lambda D / 1000 on two (nearly) uniform grids -- the float32 table axis and the
    normalisation grid (np.linspace between float32 end points: float32 nodes too).  Both exponentials advance by geometric recurrences inside
    a wave chunk; x^0.6357 splits into an item factor and the per-bin table D^0.6357.
    Appended to aux (after [D, Dn, a Dn^b, dDn, Vn] = 4 n_d + 1 values, padded to
    4 n_d + 4): [h_lut, h_norm, 0, 0] then per bin (dlt, dlt^2/2, dlt^3/6, D^p) for the
    table axis and (Dn^p, a Dn^b, V(Dn), dlt_n) for the normalisation grid."""
    if n_d % PSD_WAVES != 0:
        return None
    ch = n_d // PSD_WAVES
    k = np.arange(n_d)
    k0 = (k // ch) * ch
    D64 = np.asarray(D, dtype=np.float64)
    Dn64 = np.asarray(Dn, dtype=np.float64)
    h_l = (D64[-1] - D64[0]) / (n_d - 1)
    h_n = (Dn64[-1] - Dn64[0]) / (n_d - 1)
    dlt = D64 - D64[k0] - (k - k0) * h_l
    dltn = Dn64 - Dn64[k0] - (k - k0) * h_n
    scale = max(1.0, float(np.max(np.abs(D64))))
    if np.max(np.abs(dlt)) > 4e-6 * scale or np.max(np.abs(dltn)) > 4e-6 * scale \
            or D64[0] <= 0 or Dn64[0] <= 0:
        return None
    d.uniform_grid = 1
    ql = np.stack([dlt, dlt * dlt / 2.0, dlt ** 3 / 6.0, D64 ** ICE_POWER], axis=1).ravel()
    qn = np.stack([Dn64 ** ICE_POWER, np.asarray(aDb, dtype=np.float64),
                   np.asarray(Vn, dtype=np.float64), dltn], axis=1).ravel()
    return np.concatenate([[0.0, 0.0, 0.0], [h_l, h_n, 0.0, 0.0], ql, qn]).astype(np.float64)


def doppler_weights(h, scheme, lut):
    """Doppler scheme 2 (doppler_scatter.py:283-296): per table slice and diameter bin
    (w rcs V, w rcs) with rcs = 2 pi (Z11 - Z12 - Z21 + Z22), V = get_V(D) on the table's
    diameter axis and w the unit-spaced trapezoid weights of np.trapz (1/2 at both ends).
    Melting species evaluate V(D) per gate in the kernel: their first column is unused."""
    t = np.asarray(lut.value_table, dtype=np.float64)
    n_d = t.shape[2]
    rcs = 2 * np.pi * (t[..., 0] - t[..., 1] - t[..., 2] + t[..., 3])
    w = np.ones(n_d)
    w[0] = w[-1] = 0.5
    out = np.empty(t.shape[:3] + (2,), dtype=np.float64)
    out[..., 1] = rcs * w
    if h in ('mS', 'mG'):
        out[..., 0] = out[..., 1]
    else:
        c = _consts(h, scheme)
        D = np.asarray(lut.axes[lut.axes_names['d']])
        V = np.asarray(c['alpha'] * D ** c['beta'], dtype=np.float64)     # _Hydrometeor.get_V
        out[..., 0] = (V * rcs) * w
    return out


def t_function_tables(which=('snow_n0', 'ice_mom2_a')):
    """float32 functions of the temperature over EVERY float32 value in [128 K, 512 K),
    evaluated here with the reference's own NumPy expressions (hydrometeors.py:896 and
    :1284-1292): NumPy's float32 exp / power are not correctly rounded (1-2 ulp off in 20-40 %
    of the arguments), and a 1-ulp difference of the snow intercept shows up above 1e-5 in
    K_DP.  The device looks the value up by the bit pattern of T (cpol_stage_t_function), so it
    uses the bits the reference would have used on this host.  {name: float32 [2**24]}"""
    out = {k: _TFUN_CACHE[k] for k in which if k in _TFUN_CACHE}       # once per process
    which = [k for k in which if k not in out]
    if not which:
        return out
    # (not disk-cached: reading 2 x 64 MB back and checking them against this host's NumPy takes as
    # long as evaluating them, 0.3 s; tablecache.py keeps the slow tables)
    T = np.arange(N.TFUN_FIRST_BITS, N.TFUN_FIRST_BITS + N.TFUN_COUNT, dtype=np.uint32).view(np.float32)
    with np.errstate(over='ignore', under='ignore', invalid='ignore'):
        if 'snow_n0' in which:
            out['snow_n0'] = 13.5 * (5.65 * 10 ** 5 * np.exp(-0.107 * (T - 273.15))) / 1000
        if 'ice_mom2_a' in which:
            n = 3
            Tc = T - K.T0
            a = 5.065339 - 0.062659 * Tc - 3.032362 * n + 0.029469 * Tc * n \
                - 0.000285 * Tc ** 2 + 0.312550 * n ** 2 + 0.000204 * Tc ** 2 * n \
                + 0.003199 * Tc * n ** 2 - 0.015952 * n ** 3
            out['ice_mom2_a'] = 10 ** (a)
    for k, v in out.items():
        assert v.dtype == np.float32, (k, v.dtype)
        _TFUN_CACHE[k] = v
    return out


_TFUN_CACHE = {}


# melting species: the fw-only factors of N(D) from polynomial tables (False: every (item, bin)
# evaluates the melting-particle relations directly, as round 1 did)
MELT_TABLES = True


def _melting_factors(d, fw, n_d, dtype=np.longdouble):
    """The four factors of the melting-particle PSD that depend on the wet fraction only
    (hydrometeors.py:333-339, 372-439, 457-478), on the per-item diameter grid, in extended
    precision: D_r, G = N_r0 sqrt(D_r) V_r / V dD_r/dD, G M, G V   -> each [len(fw), n_d]."""
    fw = np.asarray(fw, dtype=dtype)[:, None]
    k = np.arange(n_d, dtype=dtype)[None, :]
    f = lambda x: dtype(x)                                             # noqa: E731
    d_max = fw * f(d.r_dmax) + (1 - fw) * f(d.s_dmax)
    d_min = fw * f(d.r_dmin) + (1 - fw) * f(d.s_dmin)
    step = (d_max - d_min) / dtype(n_d - 1)
    D = d_min + step * k
    D2 = D + f(0.01)
    fw2 = fw * fw
    M = fw2 * (f(d.r_a) * D ** 3) + (1 - fw2) * (f(d.a) * D ** f(d.b))
    M2 = fw2 * (f(d.r_a) * D2 ** 3) + (1 - fw2) * (f(d.a) * D2 ** f(d.b))
    c6 = dtype(6.0e6) / dtype(np.pi)                                   # rho_w = 1e-6 kg mm-3
    third = dtype(1) / dtype(3)
    Dr = (c6 * M) ** third
    sq = np.sqrt(Dr)
    dDr = ((c6 * M2) ** third - Dr) * dtype(100)
    Vr = f(d.r_alpha) * sq
    Vd = f(d.alpha) * D ** f(d.beta)
    phi = f(0.246) * fw + (1 - f(0.246)) * fw ** 7
    V = phi * Vr + (1 - phi) * Vd
    G = (f(d.r_n0) * sq) * Vr / V * dDr
    return Dr, G, G * M, G * V


def melting_fw_tables(d, n_d, n_t, t_lo, t_step, deg=None):
    """Disk-cached (tablecache.memo) form of _melting_fw_tables: keyed by the descriptor's constants,
    the axes and the builder's source."""
    from . import tablecache
    dd = type(d).from_buffer_copy(bytes(d))
    dd.table_id = 0
    dd.var_q = dd.var_qn = dd.var_t = 0            # (where the species sits in the variable list does not matter)
    parts = ['melt_fw', bytes(dd), n_d, n_t, float(t_lo), float(t_step), deg, np.__version__,
             tablecache.source_of(_melting_fw_tables, _melting_factors)]
    n_out = 2 * n_t + n_t * n_d * N.MELT_FUNCS * ((N.MELT_DEGREE if deg is None else deg) + 1)
    return tablecache.memo('melt_fw', parts, lambda: _melting_fw_tables(d, n_d, n_t, t_lo, t_step, deg),
                           verify=lambda v: v.dtype == np.float64 and v.shape == (n_out,) and bool(np.isfinite(v).all()))


def _melting_fw_tables(d, n_d, n_t, t_lo, t_step, deg=None):
    """aux[] of a melting species (cpol_stage_hydro): per wet-fraction bin b of the table's
    second axis the centre and 1 / half-width of its fw interval, then per (b, diameter bin) the
    monomial coefficients in u = (fw - centre) / half-width of the degree-`deg` Chebyshev
    interpolants of D_r, G, G M, G V (nodes and transform in extended precision).  Bin 0 reaches
    down to fw = 0 and the last bin up to fw = 1 (lookup_line clips the index, lut.py:336-341).
    The coefficient tail of every interpolant is checked against 1e-14 of the function."""
    from numpy.polynomial import chebyshev as cheb
    deg = N.MELT_DEGREE if deg is None else deg
    nn = deg + 1
    ld = np.longdouble
    j = np.arange(nn)
    x = np.cos(np.pi * (j + 0.5) / nn).astype(ld)                      # Chebyshev nodes
    Tn = np.cos(np.pi * np.outer(np.arange(nn), (j + 0.5)) / nn).astype(ld)
    c2m = np.array([cheb.cheb2poly(np.eye(nn)[n]).tolist() + [0.0] * (nn - n - 1) for n in range(nn)])
    c2m = c2m.astype(ld)                                               # [n][power]
    head = np.empty((n_t, 2))
    out = np.empty((n_t, n_d, N.MELT_FUNCS, nn))
    worst = 0.0
    for b in range(n_t):
        lo = 0.0 if b == 0 else t_lo + b * t_step
        hi = 1.0 if b == n_t - 1 else t_lo + (b + 1) * t_step
        mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
        head[b] = mid, 1.0 / half
        fw = ld(mid) + ld(half) * x
        for q, F in enumerate(_melting_factors(d, fw, n_d)):
            co = (ld(2) / nn) * (Tn @ F)                               # [n, n_d] Chebyshev coefficients
            co[0] *= ld(0.5)
            scale = np.max(np.abs(F), axis=0)
            worst = max(worst, float(np.max(np.abs(co[-1]) / scale)))  # |c_deg|: the size of the first dropped terms
            out[b, :, q, :] = (c2m.T @ co).T.astype(np.float64)       # monomial, lowest power first
    if worst > 3e-14:
        return None            # wet-fraction bins too wide for this degree (coarse test tables): direct form
    return np.concatenate([head.ravel(), out.ravel()])


# 1-moment ice: the three normalisation sums from polynomial tables in log2(lambda) (False: summed
# inside the diameter-bin loop of every item, as round 1 did)
ICE_TABLES = True
ICE_LOG2_LO, ICE_LOG2_HI, ICE_PANELS_PER_OCTAVE = -24, 15, 8


def ice_norm_tables(Dn, aDb, Vn, deg=None):
    """Disk-cached (tablecache.memo) form of _ice_norm_tables."""
    from . import tablecache
    parts = ['ice_norm', np.asarray(Dn, dtype=np.float64), np.asarray(aDb, dtype=np.float64),
             np.asarray(Vn, dtype=np.float64), deg, ICE_LOG2_LO, ICE_LOG2_HI, ICE_PANELS_PER_OCTAVE, ICE_POWER,
             np.__version__, tablecache.source_of(_ice_norm_tables)]
    n_pan = (ICE_LOG2_HI - ICE_LOG2_LO) * ICE_PANELS_PER_OCTAVE
    n_out = 4 + n_pan * 3 * ((N.ICE_DEGREE if deg is None else deg) + 1)
    return tablecache.memo('ice_norm', parts, lambda: _ice_norm_tables(Dn, aDb, Vn, deg),
                           verify=lambda v: v.dtype == np.float64 and v.shape == (n_out,) and bool(np.isfinite(v).all()))


def _ice_norm_tables(Dn, aDb, Vn, deg=None):
    """1-moment ice crystals: the renormalisation of the Field et al. (2005) PSD
    (hydrometeors.py:1331-1339) and IceParticle.integrate_V (:1256-1275) sum
    phi_23(lambda D_k / 1000) x (a D_k^b | V(D_k) | 1) over the 1024 nodes of the normalisation
    grid -- functions of the item's lambda only.  Tabulated here as degree-`deg` Chebyshev
    interpolants (monomial coefficients in u in [-1, 1]) on panels of 1/8 octave of lambda
    between 2^-24 and 2^15 (beyond, the kernel falls back to summing), nodes and transform in
    extended precision: < 3e-15 relative.  -> [log2 lo, panels per octave, n_panels, 0] +
    [n_panels][3][deg + 1]."""
    from numpy.polynomial import chebyshev as cheb
    deg = N.ICE_DEGREE if deg is None else deg
    nn = deg + 1
    ld = np.longdouble
    ppo = ICE_PANELS_PER_OCTAVE
    n_pan = (ICE_LOG2_HI - ICE_LOG2_LO) * ppo
    j = np.arange(nn)
    x = np.cos(np.pi * (j + 0.5) / nn).astype(ld)
    Tn = np.cos(np.pi * np.outer(np.arange(nn), (j + 0.5)) / nn).astype(ld)
    c2m = np.array([cheb.cheb2poly(np.eye(nn)[n]).tolist() + [0.0] * (nn - n - 1) for n in range(nn)]).astype(ld)
    Dn_, aDb_, Vn_ = (np.asarray(v, dtype=np.float64).astype(ld) for v in (Dn, aDb, Vn))
    p = np.arange(n_pan, dtype=ld)[:, None]
    log2lam = ld(ICE_LOG2_LO) + (p + (x[None, :] + 1) / 2) / ld(ppo)          # [n_pan, nn]
    lam = np.exp2(log2lam).reshape(-1)
    xk = lam[:, None] * Dn_[None, :] / ld(1000)
    phi = ld(490.6) * np.exp(-ld(20.78) * xk) + ld(17.46) * xk ** ld(ICE_POWER) * np.exp(-ld(3.290) * xk)
    F = np.stack([(phi * aDb_).sum(1), (phi * Vn_).sum(1), phi.sum(1)]).reshape(3, n_pan, nn)
    co = (ld(2) / nn) * np.einsum('nj,fpj->fpn', Tn, F)                        # Chebyshev coefficients
    co[..., 0] *= ld(0.5)
    tail = float(np.max(np.abs(co[..., -1]) / np.max(np.abs(F), axis=2)))
    if tail > 3e-14:
        return None
    mono = np.einsum('np,fqn->qfp', c2m, co).astype(np.float64)                # [n_pan, 3, nn]
    return np.concatenate([[float(ICE_LOG2_LO), float(ppo), float(n_pan), 0.0], mono.ravel()])
