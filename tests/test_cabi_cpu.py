"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every
symbol include/cosmo_pol_amd.h declares (no compute without a GPU), the ctypes
structures match the header, and the host-side logic of the product."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    with open(os.path.join(ROOT, 'include', 'cosmo_pol_amd.h')) as f:
        return f.read()


def test_library_exports_every_declared_symbol():
    from cosmo_pol_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _native.load_library()        # (not a bare CDLL: one HIP runtime per process, see below)
    declared = set(re.findall(r'\b(cpol_[a-z_]+)\s*\(', _header()))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof of every struct as gcc sees the header == ctypes mirror."""
    import subprocess
    from cosmo_pol_amd import _native as N
    pairs = [('cpol_hydro_desc', N.HydroDesc), ('cpol_sweep_params', N.SweepParams),
             ('cpol_ray_tables_t', N.RayTables), ('cpol_outputs', N.Outputs),
             ('cpol_counters_t', N.Counters)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cosmo_pol_amd.h"',
             'int main(void){']
    for cname, cls in pairs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append('return 0;}')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = str(tmp_path / 'layout')
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', exe, str(src)])
    got = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got['%s.%s' % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_no_gpu_means_loud_failure():
    """Without a HIP device the product must raise, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from cosmo_pol_amd import RadarOperator, _native
    with pytest.raises(_native.NativeError):
        RadarOperator(config={'radar': {'coords': [46.5, 7.5, 1000.], 'frequency': 5.6}}, luts={})


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'cosmo_pol_amd')
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.inl', '.h')):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+(oracle|cosmo_pol_oracle|ref_shim)', src, re.M), fn
                assert 'interp_twin' not in src and '_ref/' not in src, fn


def test_config_defaults_and_validation(capsys):
    from cosmo_pol_amd import config as cfg
    c = cfg.sanity_check({'radar': {'coords': [46.0, 7.0, 500], 'frequency': 5.6,
                                    'radial_resolution': 7}})        # invalid -> default
    assert c['radar']['radial_resolution'] == 500
    assert 'Invalid value entered for key: radar/radial_resolution' in capsys.readouterr().out
    assert c['integration']['nh_GH'] == 3 and c['integration']['nv_GH'] == 9
    assert c['microphysics']['with_ice_crystals'] == 1 and c['radar']['K_squared'] == 0.93
    with pytest.raises(ValueError):
        cfg.sanity_check({'radar': {'frequency': 5.6}})
    with pytest.raises(ValueError):
        cfg.sanity_check({'radar': {'coords': [1, 2], 'frequency': 5.6}})
    c = cfg.sanity_check({'radar': {'coords': [46.0, 7.0, 500], 'frequency': 5.6, 'K_squared': None},
                          'attenuation': {'correction': 1}})          # unknown section kept
    assert abs(c['radar']['K_squared'] - 0.9304396232221553) < 1e-15
    assert c['attenuation'] == {'correction': 1}
    assert 2.3 in cfg.Range(1.2, 5.3) and 1.9 not in cfg.Range(2.0, 10.0) and 3 not in cfg.Range(1.2, 5.3)


def test_reference_option_file_parses():
    from cosmo_pol_amd import config as cfg
    yml = os.path.join(ROOT, 'tests', 'golden', 'CH_PPI_like.yml')
    c = cfg.sanity_check(cfg.init(yml))
    assert c['radar']['frequency'] == 5.6 and c['radar']['radial_resolution'] == 150
    assert c['integration']['weight_threshold'] == 0.999 and c['doppler']['scheme'] == 2


def test_host_geometry_matches_oracle(golden):
    """Sub-beam weights / order and per-ray constants of the product's host side
    against the oracle (which is pinned to the reference's golden vectors)."""
    from cosmo_pol_amd import config as cfg
    from cosmo_pol_amd import geometry as geo
    from cosmo_pol_oracle import beam, geodesy
    from cosmo_pol_oracle import config as ocfg
    g = golden('quadrature')
    for ci in range(6):
        nh, nv, thr = g['case_%d' % ci]
        over = {'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6, '3dB_beamwidth': 1.3},
                'integration': {'nh_GH': int(nh), 'nv_GH': int(nv), 'weight_threshold': float(thr)}}
        sub = geo.gauss_hermite_subbeams(cfg.sanity_check(over))
        pts = np.stack([sub.pts_hor[sub.sub_h] + 10.0, sub.pts_ver[sub.sub_v] + 5.0], axis=1)
        assert np.array_equal(pts, g['pts_%d' % ci])
        assert np.array_equal(sub.sub_w, g['w_%d' % ci])
    assert geo.get_earth_radius(7.0) == beam.earth_radius(7.0)
    conf = cfg.sanity_check({'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6},
                             'integration': {'nh_GH': 3, 'nv_GH': 3}})
    sub = geo.gauss_hermite_subbeams(conf)
    az = np.array([0., 33., 271.5])
    el = np.array([1., 2., 88.])
    traj, gt = geo.ray_tables(conf['radar']['coords'], az, el, sub)
    for r in range(3):
        for i in range(3):
            k = geodesy.direct_ray_constants(46.5, 7.5, sub.pts_hor[i] + az[r])
            exp = [k['sin_a1'], k['cos_a1'], k['sigma1'], k['sin_alpha'], k['bA'], k['B'], k['C']]
            assert np.array_equal(gt[r, i, :7], np.array(exp, dtype=np.float64))
        for j in range(3):
            e = np.deg2rad(sub.pts_ver[j] + el[r])
            assert np.array_equal(traj[r, j], [e, np.sin(e), np.cos(e), sub.pts_ver[j] + el[r]])


def test_host_hydro_tables_follow_numpy_promotion():
    from cosmo_pol_amd import hydrometeors as hyd
    from cosmo_pol_amd import synthetic
    from cosmo_pol_amd import constants as K
    vi = {v: i for i, v in enumerate(hyd.BASE_VARIABLES + hyd.BASE_VARIABLES_2MOM)}
    lut = synthetic.make_lut('R', 5.6, '1mom', n_e=2, n_t=2)
    d, table, pre, dnu, aux = hyd.build_hydro('R', '1mom', lut, vi)
    D = lut.axes[2]
    assert D.dtype == np.float32 and pre.dtype == np.float64
    assert np.array_equal(pre, (K.C1.N0_R * np.sqrt(D)).astype(np.float64))
    assert np.array_equal(dnu, D.astype(np.float64))
    assert d.dD == float(D[1] - D[0]) and d.n_d == 1024 and d.var_q == vi['QR_v']
    lut = synthetic.make_lut('G', 5.6, '1mom', n_e=2, n_t=2)
    d, table, pre, dnu, aux = hyd.build_hydro('G', '1mom', lut, vi)
    assert np.all(pre == 4000.0)
    lut = synthetic.make_lut('mS', 5.6, '1mom', n_e=2, n_t=3)
    d, table, pre, dnu, aux = hyd.build_hydro('mS', '1mom', lut, vi)
    assert d.second_axis_f64 == 1 and (d.r_dmin, d.r_dmax, d.s_dmin, d.s_dmax) == (0.1, 8, 0.2, 20)
    assert hyd.hydrometeor_list({'microphysics': {'with_melting': 1, 'scheme': '2mom',
                                                  'with_ice_crystals': 1}}) == \
        ['R', 'S', 'G', 'mS', 'mG', 'H', 'I']


def test_lut_file_roundtrip(tmp_path):
    from cosmo_pol_amd import lut as plut
    from cosmo_pol_amd import synthetic
    from cosmo_pol_oracle import lut as olut
    for h in ('R', 'mG'):
        L = synthetic.make_lut(h, 5.6, '1mom', n_e=3, n_t=4)
        fn = str(tmp_path / plut.lut_filename(h, 5.6, '1mom'))
        plut.save_lut(L, fn)
        for loader in (plut.load_lut, olut.load_lut):     # either side reads the format
            M = loader(fn)
            assert np.array_equal(M.value_table, L.value_table)
            assert M.axes_names == L.axes_names
            for a, b in zip(M.axes, L.axes):
                assert np.array_equal(np.asarray(a), np.asarray(b))
            e = np.array([0.5, 3.9, 77.0], dtype=np.float32)
            assert np.array_equal(M.lookup_line(e=e), L.lookup_line(e=e))
    assert os.path.basename(fn) == 'lut_SZ_mG_5_6_1mom.lut'
    d = tmp_path / 'lut_tmatrix_masc'
    d.mkdir()
    plut.save_lut(synthetic.make_lut('R', 5.6, '1mom', n_e=2, n_t=2), str(d / 'lut_SZ_R_5_6_1mom.lut'))
    out = plut.load_all_lut('1mom', ['R'], 5.6, 'tmatrix_masc', lut_dir=str(tmp_path))
    assert out['R'].value_table.shape == (2, 2, 1024, 12)
    with pytest.raises(IOError):
        plut.load_all_lut('1mom', ['S'], 5.6, 'tmatrix_masc', lut_dir=str(tmp_path))


def test_nyquist_table(tmp_path):
    from cosmo_pol_amd import config as cfg
    fn = tmp_path / 'nyq.txt'
    fn.write_text('elevation,azimuth,nyquist\n-0.2,0,8.3\n1.0,0,9.6\n1.0,180,10.5\n2.5,0,11.\n')
    c = cfg.sanity_check({'radar': {'coords': [46.0, 7.0, 500], 'frequency': 5.6,
                                    'nyquist_velocity': str(fn)}})
    nyq = c['radar']['nyquist_velocity']
    assert np.array_equal(nyq([0.9, 1.2, 2.4, -5.0], [10., 170., 300., 0.]), [9.6, 10.5, 11., 8.3])
    c2 = cfg.sanity_check(c)                      # re-validation keeps the table object
    assert c2['radar']['nyquist_velocity'] is nyq


def test_ray_tables_c_side_matches_numpy():
    """cpol_ray_tables (libm, for C / C++ hosts) against geometry.ray_tables (NumPy, what the
    Python host uses): pure host code, callable without a GPU."""
    import ctypes as C
    from cosmo_pol_amd import _native as N, geometry as geo
    lib = N.load_library()
    conf = {'radar': {'3dB_beamwidth': 1.0}, 'integration': {'scheme': 1, 'nh_GH': 3, 'nv_GH': 5,
                                                                'weight_threshold': 1.0}}
    sub = geo.gauss_hermite_subbeams(conf)
    coords = [46.5, 7.5, 1000.0]
    az = np.array([0.0, 33.3, 181.0, 359.5])
    el = np.array([0.5, 3.0, 45.0, 89.0])
    traj, geo_t = geo.ray_tables(coords, az, el, sub)
    p = N.SweepParams()
    p.n_rays, p.n_hnodes, p.n_vnodes = len(az), len(sub.pts_hor), len(sub.pts_ver)
    p.sin_u1, p.cos_u1, _ = geo.radar_site_constants(coords)
    t_c = np.zeros_like(traj)
    g_c = np.zeros_like(geo_t)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)          # noqa: E731
    rc = lib.cpol_ray_tables(C.byref(p), ptr(az), ptr(el), ptr(np.ascontiguousarray(sub.pts_hor)),
                             ptr(np.ascontiguousarray(sub.pts_ver)), ptr(t_c), ptr(g_c))
    assert rc == 0
    np.testing.assert_allclose(t_c, traj, rtol=0, atol=1e-15)
    np.testing.assert_allclose(g_c, geo_t, rtol=1e-15, atol=1e-15)


def test_c_host_links_and_runs(tmp_path):
    """The boundary is a C ABI: a plain-C translation unit includes the header, links the
    library and runs (the GPU-free entry points; cpol_create must fail cleanly here)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, 'cosmo_pol_amd', 'csrc')
    exe = str(tmp_path / 'host_check')
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(root, 'include'),
           os.path.join(root, 'tests', 'c_host', 'host_check.c'), '-o', exe,
           '-L', libdir, '-lcosmo_pol_hip', '-lm', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'C_HOST_OK' in r.stdout, r.stdout + r.stderr


def test_t_function_tables_are_the_reference_expressions_bit_for_bit():
    """hydrometeors.t_function_tables: float32 functions of T over every float32 in
    [128, 512) K, equal to the oracle's (= the reference's) NumPy expressions bit for bit."""
    from cosmo_pol_amd import _native as N, hydrometeors as hyd
    from cosmo_pol_oracle import psd
    tabs = hyd.t_function_tables()
    assert set(tabs) == {'snow_n0', 'ice_mom2_a'}
    rng = np.random.default_rng(2)
    T = np.concatenate([rng.uniform(128.0, 511.9, 5000), [128.0, 200.0, 273.15, 276.0]]).astype(np.float32)
    idx = T.view(np.uint32) - N.TFUN_FIRST_BITS
    assert idx.max() < N.TFUN_COUNT
    sn = psd.create_hydrometeor('S', '1mom')
    sn.set_psd(T, np.full(T.shape, 1e-4, dtype=np.float32))
    assert np.asarray(sn.N0).dtype == np.float32
    assert np.array_equal(tabs['snow_n0'][idx].view(np.uint32), np.asarray(sn.N0).view(np.uint32))
    # ice: (QM / a) ** (1 / b) with a = 10 ** poly(T): recover a from the oracle's own statement
    n, Tc = 3, T - 273.15
    a = 5.065339 - 0.062659 * Tc - 3.032362 * n + 0.029469 * Tc * n \
        - 0.000285 * Tc ** 2 + 0.312550 * n ** 2 + 0.000204 * Tc ** 2 * n \
        + 0.003199 * Tc * n ** 2 - 0.015952 * n ** 3
    assert np.array_equal(tabs['ice_mom2_a'][idx].view(np.uint32), (10 ** a).view(np.uint32))


def test_one_hip_runtime_whatever_the_import_order():
    """libcosmo_pol_hip.so first, torch afterwards (a user who builds a RadarOperator and then
    touches torch.cuda): the process must map ONE libamdhip64 -- with two, the second runtime to
    initialise reports "No HIP GPUs are available" (gpurun_out/t_r2f.log of round 2).  Fresh
    interpreter, because the order is the point."""
    import subprocess
    import sys
    code = ('import sys; sys.path.insert(0, %r)\n'
            'from cosmo_pol_amd import _native as N\n'
            'N.load_library()\n'
            'assert "torch" not in sys.modules\n'
            'first = N.hip_runtimes_mapped()\n'
            'import torch\n'
            'torch.cuda.is_available()\n'
            'both = N.hip_runtimes_mapped()\n'
            'print(len(first), len(both))\n' % ROOT)
    out = subprocess.check_output([sys.executable, '-c', code], env=dict(os.environ, CPOL_HIP_RUNTIME=''))
    assert out.split() == [b'1', b'1'], out
    # the other order (what bench.py and most tests do)
    code2 = ('import sys; sys.path.insert(0, %r)\n'
             'import torch\n'
             'from cosmo_pol_amd import _native as N\n'
             'N.load_library()\n'
             'print(len(N.hip_runtimes_mapped()))\n' % ROOT)
    assert subprocess.check_output([sys.executable, '-c', code2]).split() == [b'1']


def test_every_product_module_imports():
    """A syntax / import error in a product module must show up without a GPU."""
    import importlib
    import pkgutil
    import cosmo_pol_amd
    for m in pkgutil.iter_modules(cosmo_pol_amd.__path__):
        if m.name == 'pyart_wrapper':
            continue                        # needs Py-ART (guarded at its call sites)
        importlib.import_module('cosmo_pol_amd.' + m.name)
    import bench  # noqa: F401
    import __graft_entry__  # noqa: F401
