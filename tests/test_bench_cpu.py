"""bench.py pieces that run without a GPU: the all-core CPU legs (child interpreter, time-boxed)
and the shape of the roofline helpers."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_pool_child_reports_every_leg():
    """`bench.py --cpu-pool-child` (what cpu_baseline_pool starts with a hard time limit): a fresh
    interpreter without torch, fork pools of P = os.cpu_count() (and fewer) workers over the
    oracle's per-radial path, every leg time-boxed; one JSON line with the documented keys."""
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-pool-child', '--small',
                        '--workload', 'c2'], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{') and '"partial"' not in ln]
    assert lines, r.stdout[-500:]
    d = json.loads(lines[-1])
    assert d['cores'] == os.cpu_count() and d['unit'] == 'gates/s'
    assert d['value'] and d['value'] > 0
    assert str(os.cpu_count()) in d['by_workers']
    assert d['reference_style'] and d['reference_style']['value'] > 0
    assert 'torch' not in r.stderr.lower() or 'error' not in r.stderr.lower()


def test_stage_roofline_is_recomputable_from_the_committed_profile(tmp_path, monkeypatch):
    """roofline.frac of the bench line = HBM bytes per launch of the committed PMC summary / live
    duration / 8 TB/s for the dominant stage; hbm_alg_frac = SURVEY 8(d) bytes over the same time."""
    import bench
    prof = {'_note': 'x',
            'k_psd_lookup(HydroSet, ItabSet, LookupArgs)': {'avg_us': 25.0, 'hbm_bytes': 29.0e6},
            'k_final<256>(FinalArgs, ScanRayArgs)': {'avg_us': 30.0, 'hbm_bytes': 41.0e6},
            'k_interp_sweep(ModelDev, InterpArgs)': {'avg_us': 22.0, 'hbm_bytes': 51.0e6},
            'k_itab_fit(ItabFitArgs)': {'avg_us': 300.0, 'hbm_bytes': 9e8}}
    (tmp_path / 'profiles').mkdir()
    (tmp_path / 'profiles' / 'r3_c2_iso_summary.json').write_text(json.dumps(prof))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    ms = {'interp': 0.022, 'classify': 0.016, 'bucket': 0.008, 'psd': 0.026, 'final': 0.031}
    r = bench.roofline_of_dominant_stage('c2_iso', ms, 180000, 195751, 180000, 9, 80)
    assert r['stage'] == 'final' and r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['traffic'] - 41.0e6) < 1 and abs(r['achieved'] - 41.0e6 / 0.031e-3 / 1e9) < 1e-6
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0 < r['frac'] < 1
    assert r['algorithmic_bytes_per_launch'] == 180000 * 48
    st = r['stages']
    assert st['psd']['algorithmic_bytes'] == 195751 * bench.LUT_SLICE_BYTES and st['psd']['hbm_alg_frac'] > 1
    assert st['interp']['algorithmic_bytes'] == 180000 * (4 * 80 * 4 + 9 * 8 * 4)
    assert st['classify']['traffic'] is None and 'k_itab_fit' not in str(st)
    assert abs(r['whole_sweep']['traffic'] - (29.0e6 + 41.0e6 + 51.0e6)) < 1


def test_profile_summary_counts_scalar_cache_reads_in_full(tmp_path):
    """tools/profile_summary.py: hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB for kernels that read through
    the vector caches (rocprofv3 reports half of those bytes on gfx950, profiles/r3_fetch_calibration.txt)
    but FETCH_SIZE + WRITE_SIZE for k_subbeam_sum<true>, whose coefficient rows come through the scalar
    cache and are reported in full; the vector rule's figure stays as hbm_bytes_upper."""
    def counters(d, name, rows):
        os.makedirs(d)
        with open(os.path.join(d, 'x_counter_collection.csv'), 'w') as f:
            f.write('Kernel_Name,Counter_Name,Counter_Value\n')
            for k, v in rows:
                f.write('"%s",%s,%g\n' % (k, name, v))
    coop = 'k_subbeam_sum_scalar(HydroSet, ItabSet, SubsumArgs)'
    gather = 'void k_subbeam_sum_gather<1, 2>(HydroSet, ItabSet, SubsumArgs)'
    stats = tmp_path / 'stats.csv'
    stats.write_text('"Name","Calls","TotalDurationNs","AverageNs"\n"%s",2,4000000,2000000\n"%s",2,1000000,500000\n'
                     % (coop, gather))
    counters(str(tmp_path / 'F'), 'FETCH_SIZE', [(coop, 5000.0), (gather, 400.0)])
    counters(str(tmp_path / 'W'), 'WRITE_SIZE', [(coop, 600.0), (gather, 100.0)])
    counters(str(tmp_path / 'S'), 'SQ_INSTS_VALU', [(coop, 1e6), (gather, 2e5)])
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'profile_summary.py'), str(stats),
                        str(tmp_path / 'F'), str(tmp_path / 'W'), str(tmp_path / 'S')],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-1000:]
    d = json.loads(r.stdout)
    assert d[coop]['hbm_bytes'] == (5000.0 + 600.0) * 1024 and d[coop]['hbm_bytes_upper'] == (2 * 5000.0 + 600.0) * 1024
    assert d[gather]['hbm_bytes'] == (2 * 400.0 + 100.0) * 1024 and 'hbm_bytes_upper' not in d[gather]
    assert d[coop]['avg_us'] == 2000.0 and d[coop]['SQ_INSTS_VALU'] == 1e6


def test_stage_kernels_cover_every_kernel_of_a_sweep():
    """Every kernel name of the committed profiles of a sweep (this round's) belongs to a stage of bench.py's
    roofline (or to table building / staging): no kernel's bytes fall out of `roofline.stages`."""
    import bench
    known = [k for ks in bench.STAGE_KERNELS.values() for k in ks]
    integrating = ('k_psd_uniform', 'k_psd<', 'k_psd_ice2', 'k_psd_melting', 'k_spec_', 'k_ml_weights')
    for name in ('c2_iso', 'c3_el3_iso', 'c4_volume_iso', 'c4_share8_iso', 'c5_ku_iso'):
        prof, path = bench.load_profile_summary(name)
        assert prof is not None and path.endswith('r4_%s_summary.json' % name)
        if name in ('c2_iso', 'c5_ku_iso'):                  # single-beam sweeps run the fused kernel
            assert any('k_gate1' in k for k in prof) and not any('k_classify' in k for k in prof)
        for kernel, c in prof.items():
            if kernel.startswith('_') or not isinstance(c, dict) or kernel.startswith('__amd'):
                continue
            if any(t in kernel for t in bench.TABLE_BUILD_KERNELS) or any(t in kernel for t in integrating):
                continue
            assert any(k in kernel for k in known), (name, kernel)
