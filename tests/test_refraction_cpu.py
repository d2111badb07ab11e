"""Refraction scheme 2 (Zeng & Blahak ODE): the product's host-side solver
(cosmo_pol_amd/refraction.py) against the oracle restatement of
atm_refraction.py:56-179.  PARITY UNPINNED against the reference: its _ref_ODE
raises inside scipy.odeint under NumPy >= 1.24 (see oracle/gen_golden.py)."""
import numpy as np

import gen_golden
from cosmo_pol_amd import refraction, synthetic
from cosmo_pol_oracle import beam, refraction_ode
from cosmo_pol_oracle import constants as OK
from cosmo_pol_oracle import config as ocfg


def test_ode_paths_product_vs_oracle():
    cube = synthetic.small_test_cube(**gen_golden.CUBE_KW)
    Nf = gen_golden.refractivity_field(cube)
    coords = [46.5, 7.5, 1000]
    conf = ocfg.make_config({'radar': {'coords': coords, 'frequency': 5.6, 'range': 40000,
                                       'radial_resolution': 400}})
    rr = OK.Derived(conf).RANGE_RADAR
    h_col, n_col = refraction.refractivity_column(Nf, cube['zlevels'], cube['proj_info'],
                                                  cube['resolution'], coords)
    assert np.all(np.diff(h_col) > 0) and 1.0002 < n_col[0] < 1.0004
    for el in (0.5, 3.0, 6.4, 13.86, 20.0, 27.5):
        s, h, e = refraction.ode_path(rr, el, coords, h_col, n_col)
        so, ho, eo = refraction_ode.trajectory_ode(rr, el, coords, Nf, cube['zlevels'],
                                                   cube['proj_info'], cube['resolution'])
        for a, b in ((s, so), (h, ho), (e, eo)):
            # bit for bit: same scipy building blocks (interp1d inside the column, LSODA) on the same
            # dtypes as the reference leaves them -- float32 columns, float32 slopes (round 2 converted the
            # column to float64 first: LSODA's step control turned that into 1-ulp differences in 16 % of
            # the float32 path values, and those into LUT-bin flips at single gates)
            assert a.dtype == b.dtype == np.float32
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        # physical sanity against the 4/3-earth model (the ODE starts at the first gate
        # centre with h = radar altitude, so compare increments)
        s43, h43, e43 = beam.trajectory_4_3(rr, el, coords)
        assert np.max(np.abs((h - h[0]) - (h43 - h43[0]))) < 80.0
        assert np.max(np.abs((s - s[0]) - (s43 - s43[0]))) < 400.0
    paths = refraction.ode_paths(rr, [1.0, 1.0, 2.0], np.array([-0.3, 0.0, 0.3]), coords, h_col, n_col)
    assert paths.shape == (3, 3, 3, len(rr)) and np.array_equal(paths[0], paths[1])
    assert not np.array_equal(paths[0], paths[2])


def test_fast_interpolant_equals_interp1d_bitwise_and_helper_processes_give_the_same_paths():
    """Round-3 review, weak 9: an RHI of 90 elevations x 3 vertical nodes needs 270 LSODA solves.  The
    interpolant of the right-hand side no longer goes through interp1d's per-call checks (same bits), the
    solves are kept across calls and, from 16 missing ones on, spread over helper processes (same bits)."""
    import time
    rng = np.random.default_rng(11)
    for dt in (np.float32, np.float64):
        x = np.cumsum(rng.uniform(50, 400, 80)).astype(dt)
        y = (1.0003 - 3e-8 * x + 1e-6 * rng.standard_normal(80)).astype(dt)
        f = refraction._PiecewiseLinear(x, y)
        for v in np.concatenate([rng.uniform(float(x[0]), float(x[-1]), 4000), x.astype(np.float64),
                                 [float(x[0]), float(x[-1])]]):
            a, b = f(float(v)), float(f.f(float(v)))
            assert a == b and type(a) is float, (dt, v, a, b)
    cube = synthetic.small_test_cube(**gen_golden.CUBE_KW)
    Nf = gen_golden.refractivity_field(cube)
    coords = [46.5, 7.5, 1000]
    rr = np.arange(200., 40000., 400.)
    h_col, n_col = refraction.refractivity_column(Nf, cube['zlevels'], cube['proj_info'], cube['resolution'], coords)
    els = np.arange(0.5, 30.5, 1.0)                      # 30 elevations x 3 nodes: 90 solves
    pts = np.array([-0.3, 0.0, 0.3])
    refraction._SOLVED.clear()
    t0 = time.perf_counter()
    serial = refraction.ode_paths(rr, els, pts, coords, h_col, n_col, workers=0)
    t_serial = time.perf_counter() - t0
    t0 = time.perf_counter()
    again = refraction.ode_paths(rr, els, pts, coords, h_col, n_col, workers=0)      # all from the cache
    t_cached = time.perf_counter() - t0
    assert np.array_equal(serial, again) and t_cached < 0.2 * t_serial
    refraction._SOLVED.clear()
    pooled = refraction.ode_paths(rr, els, pts, coords, h_col, n_col, workers=3)
    assert len(refraction._POOL) == 3
    assert np.array_equal(serial.view(np.uint32), pooled.view(np.uint32))
    # another refractivity column: nothing of the cache applies
    other = refraction.ode_paths(rr, els[:2], pts, coords, h_col, (n_col + np.float32(1e-6)).astype(n_col.dtype), workers=0)
    assert not np.array_equal(other[0], serial[0])
    refraction._pool_close()
    assert refraction._POOL == []
