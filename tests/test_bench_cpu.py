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
    interpreter without torch, fork pools of P = min(os.cpu_count(), 64) (and fewer) workers over the
    oracle's per-radial path, every leg time-boxed; one JSON line with the documented keys."""
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-pool-child', '--small',
                        '--workload', 'c2'], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{') and '"partial"' not in ln]
    assert lines, r.stdout[-500:]
    d = json.loads(lines[-1])
    assert d['cores'] == min(os.cpu_count(), 64) and d['host_cpus'] == os.cpu_count() and d['unit'] == 'gates/s'
    assert d['value'] and d['value'] > 0
    assert str(min(os.cpu_count(), 64)) in d['by_workers']
    assert d['reference_style'] and d['reference_style']['value'] > 0
    assert 'torch' not in r.stderr.lower() or 'error' not in r.stderr.lower()


def test_stage_roofline_is_recomputable_from_the_committed_profile(tmp_path, monkeypatch):
    """roofline.frac of the bench line = HBM bytes per launch of the committed PMC summary / live
    duration / 8 TB/s for the dominant stage; alg_8d = SURVEY 8(d) bytes over the same time."""
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
    assert r['alg_8d']['bytes_per_launch'] == 180000 * 48
    st = r['stages']
    assert st['psd']['algorithmic_bytes_8d'] == 195751 * bench.LUT_SLICE_BYTES and st['psd']['alg_8d_frac'] > 1
    assert st['interp']['algorithmic_bytes_8d'] == 180000 * (4 * 80 * 4 + 9 * 8 * 4)
    assert st['classify']['traffic'] is None and 'k_itab_fit' not in str(st)
    assert abs(r['whole_sweep']['traffic'] - (29.0e6 + 41.0e6 + 51.0e6)) < 1
    # the bound that binds: VALU issue = wave instructions x 4 cycles over 1024 SIMDs x 2.4 GHz x the live duration
    assert r['valu_f64']['bound'] == 'valu_f64' and r['valu_f64']['frac'] is None      # (this canned profile has no SQ counters)


def test_valu_issue_fraction():
    import bench
    f = bench.valu_issue_frac(5437184.0, 27.56e-6)          # k_gate1_species of profiles/r4_c2_iso_summary.json
    assert abs(f - 5437184.0 * 4 / (1024 * 2.4e9 * 27.56e-6)) < 1e-12 and 0.3 < f < 0.35
    assert bench.valu_issue_frac(None, 1.0) is None and bench.valu_issue_frac(1.0, None) is None


def _canned_detail():
    """A full result dict of the default N = 1 run as bench.py builds it (the shape of round 4's 22.6-KB line: nested
    child runs, per-stage tables, long notes), with a NaN and an Infinity planted."""
    note = 'x' * 700
    stages = {st: {'kernels': ['k_' + st], 'live_ms': 0.02, 'profile_us': 21.7, 'traffic': 4.1e7, 'achieved': 1900.0,
                   'frac': 0.24, 'valu_wave_instructions': 5.4e6, 'valu_frac': 0.33, 'algorithmic_bytes_8d': 9.6e9,
                   'alg_8d_frac': 43.0} for st in ('interp', 'classify', 'bucket', 'psd', 'final')}
    roof = {'kernel': 'k_gate1_species', 'stage': 'classify', 'bound': 'hbm', 'achieved': 1013.5684417848793, 'peak': 8000.0,
            'unit': 'GB/s', 'frac': 0.12669605522310992, 'traffic': 30293837.88679245,
            'traffic_source': 'profiles/r5_c2_iso_summary.json', 'avg_launch_ms': 0.029888300225138664,
            'profile_avg_us': 27.56, 'valu_f64': {'bound': 'valu_f64', 'frac': 0.3211, 'wave_instructions_per_launch': 5437184.0,
                                                  'peak_G_wave_instructions_per_s': 614.4, 'whole_sweep_frac': 0.27},
            'alg_8d': {'bytes_per_launch': 9621553152, 'frac': 40.2, 'whole_sweep_bytes': 9.9e9, 'whole_sweep_frac': 14.6,
                       'note': note},
            'whole_sweep': {'live_ms': 0.0848, 'traffic': 79237393.5, 'frac': 0.1167, 'valu_wave_instructions': 1.25e7,
                            'valu_frac': 0.24},
            'timed_region': {'ms_per_sweep': 0.04097689379705116, 'traffic_per_sweep': 79237393.50943395,
                             'achieved': 1933.709126462293, 'unit': 'GB/s', 'frac': 0.2417, 'valu_frac': 0.497, 'note': note},
            'stages': stages, 'note': note, 'integrating_kernel': {'note': note, 'frac': float('nan')}}
    cpu = {'value': 62779.57020777228, 'unit': 'gates/s', 'cores': 1, 'kind': 'port', 'sample': 'y' * 400,
           'samples_gates_per_s': [60151.5, 62126.8, 62779.5, 63310.1, 70145.7], 'radials_per_s': 125.5, 'host_cpus': 256,
           'all_cores': {'value': None, 'error': 'z' * 300, 'by_workers': {'16': 1.0e6, '64': None, '256': None}}}
    child = {'value': 4.8e8, 'unit': 'gates/s', 'n_gpus': 1, 'ms_per_step': 1.865, 'steps': 20, 'warmup': 3, 'scaling': 'weak',
             'roofline': dict(roof), 'setup_s': {'a': 1.0}, 'stages_ms': {'interp': 0.03}, 'api_ms': {'note': note},
             'single_sweep_ms': [{'elevation': e, 'device_ms': 0.13} for e in range(5)], 'cpu_baseline': dict(cpu),
             'workload': 'w' * 300, 'child_wall_s': 18.7, 'command': 'python bench.py --workload c3'}
    return {
        'metric': 'range-gates/sec', 'value': 4392719489.463923, 'unit': 'gates/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5,
        'ms_per_step': 0.3278151503764093, 'ms_per_sweep': 0.04097689379705116, 'sweeps_per_step': 8,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'value_definition': 'v2 ' + 'd' * 150,
        'config': {'workload': 'c2: step = 8 sweeps per GPU of the 360-az x 500-gate C-band PPI ' + 'w' * 600,
                   'rays_per_gpu': 360, 'gates_per_ray': 500, 'lanes': 3, 'sweeps_per_step': 8, 'parallelism': 'single GPU',
                   'small': False},
        'timed_region_repeats': {'n': 5, 'note': note}, 'roofline': roof,
        'stages_ms': {'interp': 0.024, 'note': note}, 'counters': {'n_valid_items': 195751},
        'host_outputs': {'value': 6.9e8, 'unit': 'gates/s', 'ms_per_sweep': 0.26, 'sweeps_timed': 20,
                         'ms_per_sweep_repeats': [0.64, 0.61, 0.257, 0.258, float('inf')], 'note': note},
        'value_cached_geometry': 9.6e8, 'refraction2_rhi_90x3_ms': {'note': note},
        'single_sweep_latency_ms': {'note': note}, 'process_group_backend': None, 'n_ranks_seen_by_rccl': 1,
        'host_placement': {'note': note}, 'setup_s': {'synthetic_inputs': 2.9}, 'cpu_baseline': cpu,
        'gpu_over_cpu_core': 69970.5, 'c3': dict(child), 'c4_volume_one_gpu': dict(child, speedup_vs_single_gpu=0.99),
        'c5': {'error': 'e' * 400}, 'c4_speedup_vs_single_gpu': 0.99, 'c4_speedup_single_volume': 1.0, 'gather_check': None}


def test_final_line_is_small_and_complete(tmp_path, capsys):
    """The line the driver parses (round 4: a 22.6-KB line left BENCH_r04.parsed = null): < 4 KB, strict JSON (no NaN /
    Infinity), the contract's keys at the top level with `roofline` and `cpu_baseline`, the SURVEY 8(d) step beside
    `value`; everything else in the side file and on '#detail ' lines BEFORE it."""
    import bench
    d = _canned_detail()
    line = bench.compact_line(d)
    text = json.dumps(line, allow_nan=False)                 # (raises on NaN / Infinity)
    assert len(text) < 4096, len(text)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'value_host_outputs',
              'host_outputs_ms_per_sweep_repeats', 'c4_speedup_vs_single_gpu', 'n_ranks_seen_by_rccl'):
        assert k in line, k
    assert line['value'] == 4392720000.0 and line['steps'] == 20 and line['n_gpus'] == 1
    assert line['config']['workload'].startswith('c2: step = 8 sweeps') and line['config']['lanes'] == 3
    for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'avg_launch_ms'):
        assert line['roofline'][k] is not None, k
    assert line['roofline']['bound'] == 'hbm' and line['roofline']['valu_f64']['bound'] == 'valu_f64'
    assert abs(line['roofline']['frac'] - line['roofline']['achieved'] / line['roofline']['peak']) < 1e-5
    assert line['roofline']['timed_region']['frac'] == 0.2417 and line['roofline']['alg_8d']['void'] is True
    assert set(line['cpu_baseline']) == {'value', 'unit', 'cores', 'kind', 'host_cpus', 'sample'}
    assert line['cpu_baseline']['cores'] == 1 and line['cpu_baseline']['kind'] == 'port'
    assert line['host_outputs_ms_per_sweep_repeats'][-1] is None          # the planted Infinity
    assert line['other_configs']['c5'] == {'error': 'e' * 80} and line['other_configs']['c3']['value'] == 4.8e8
    # emit(): the side file, the prefixed detail lines, then the compact line LAST
    path = tmp_path / 'bench_detail.json'
    bench.emit(d, str(path))
    out = capsys.readouterr().out.splitlines()
    assert json.loads(out[-1]) == json.loads(json.dumps(bench.compact_line(d))) and len(out[-1]) < 4096
    assert all(ln.startswith('#detail ') for ln in out[:-1]) and len(out) > 5
    assert [ln for ln in out if ln.startswith('{')] == [out[-1]]
    full = json.loads(path.read_text())
    assert full['roofline']['stages']['psd']['alg_8d_frac'] == 43.0 and full['c3']['api_ms']['note'] == 'x' * 700
    assert full['roofline']['integrating_kernel']['frac'] is None          # the planted NaN
    # a result with absurdly long strings everywhere still fits
    d['config']['parallelism'] = 'p' * 5000
    d['roofline']['kernel'] = 'k' * 5000
    d['cpu_baseline']['sample'] = 's' * 5000
    assert len(json.dumps(bench.compact_line(d))) < 4096


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
    integrating = ('k_psd_uniform', 'k_psd<', 'k_psd_ice2', 'k_psd_melting', 'k_psd_rare', 'k_spec_', 'k_ml_weights')
    for name in ('c2_iso', 'c3_el3_iso', 'c4_volume_iso', 'c4_share8_iso', 'c5_ku_iso'):
        prof, path = bench.load_profile_summary(name)
        assert prof is not None and path.endswith(('r6_%s_summary.json' % name, 'r5_%s_summary.json' % name, 'r4_%s_summary.json' % name))
        if name in ('c2_iso', 'c5_ku_iso'):                  # single-beam sweeps run the fused kernel
            assert any('k_gate1' in k for k in prof) and not any('k_classify' in k for k in prof)
        for kernel, c in prof.items():
            if kernel.startswith('_') or not isinstance(c, dict) or kernel.startswith('__amd'):
                continue
            if any(t in kernel for t in bench.TABLE_BUILD_KERNELS) or any(t in kernel for t in integrating):
                continue
            assert any(k in kernel for k in known), (name, kernel)
