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
