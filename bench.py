#!/usr/bin/env python
"""bench.py -- range-gates/s of the cosmo_pol hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workloads (BASELINE.json `configs`):

  c2 (default at every N; configs[1], the configuration the metric is quoted on)
      360-azimuth x 500-gate C-band PPI, rain + snow + graupel 1-moment, one sub-beam,
      attenuation on, synthetic COSMO-1-like cube (80 x 774 x 1158) and full-size synthetic
      scattering tables.  A step = 8 complete sweeps through the C ABI (cpol_run_sweep with
      device outputs), one at each of 8 elevations (1.0 + 0.05 k deg): every kernel of the launch
      sequence runs and the ten radar fields of every sweep are left in HBM -- `value` is the rate
      with the inputs resident in HBM when the timed region starts (cube, scattering and integral
      tables; the per-ray constants of the 8 elevations are computed on the host before the timed
      region and the library keeps the table sets of the last 8 scan geometries on the device).
      Consecutive sweeps run on three LANES (cpol_fork: shared cube and tables, own stream and work
      buffers), as the sweeps of a volume scan do in the product.  (`ms_per_sweep` = ms_per_step / 8;
      one sweep per step would make 20 steps a 0.9 ms timed region.)
      `host_outputs` of the same line = the step as the reference's API hands it over, PCIe
      included (rounds 1-3 reported THIS as `value`): a new elevation out of 16 every step, its
      per-ray tables computed on the host inside the step and uploaded, all kernels, and EVERY
      output array -- nine polarimetric observables, RVEL, the radial mask and the gate coordinates
      (lats, lons, dist, heights): 15 arrays, 76 B per gate -- copied to page-locked host memory.
      `value_cached_geometry`: that step with one fixed elevation (round 2's headline).
      N > 1 = WEAK scaling of exactly the headline step, so that the driver's series N = 1, 2, 4, 8
      compares like with like: every rank runs the N = 1 step unchanged, rank r on elevation
      (step + r) mod 8 (a scan spread over the GPUs sweep by sweep).  Sweeps are independent, so a
      step holds no collective; the timed region ends with ONE all-gather (RCCL) of every rank's
      last sweep, which is in HBM already (`gather_check`: rank 0 recomputes those sweeps and compares
      the gathered blocks bit for bit).  `value` = N x gates per sweep x K / MAX-over-ranks time.

  c3 (configs[2])   5-elevation volume (360 x 500 each), R,S,G,mS,mG,I with melting layer and ice
      crystals, one sub-beam.  A step = one volume: per-sweep tables up, kernels, all outputs down.
  c4 (configs[3]; runs after c2 in the default N > 1 run)   the c3 volume with the 7 x 7 Gauss-Hermite antenna
      quadrature (49 sub-beams).  STRONG scaling: the 360 azimuths of every sweep are split into
      contiguous blocks of ceil(360 / N) rays, one per rank (cosmo_pol_amd/distributed.py); a
      rank runs its rays of ALL five sweeps as one launch sequence and ONE all-gather per volume
      (RCCL over xGMI) assembles the volume on every rank; rank 0 copies it to page-locked host
      memory.  A step = one volume.  After the timed region rank 0 runs the same volume alone
      (`single_gpu_same_workload`) and compares the gathered result with it bit for bit
      (`gather_check`).
  c5 (configs[4])   GPM-DPR Ku (200 x 49 rays) and Ka (200 x 25) swaths over the 2-moment cube
      through RadarOperator.get_GPM_swath.  A step = both swaths; gates = the gates kept (h < 35 km).

At N = 1 with the default workload the line also carries `c3`, `c4_volume_one_gpu` and `c5`:
the same script run with --workload c3 / c4 / c5 in child processes after the c2 operator is
closed (driver-timed like the rest of the run).  At N > 1 with the default workload it carries
`c4_strong_scaling`: `--workload c4 --gpus N` run by one child process per rank on the same GPUs
after the c2 part has closed its operator and process group (its `speedup_vs_single_gpu` is the
strong-scaling figure of north_star's "azimuths sharded, one gather at the end").

OUTPUT (rank 0).  The LAST stdout line is ONE compact JSON object (< 4 KB: `compact_line`, guarded by
tests/test_bench_cpu.py): metric, value, unit, n_gpus, steps, warmup, ms_per_step, dtype, data,
config{workload, ...}, roofline{...}, cpu_baseline{...}, value_host_outputs (+ its five repeats),
c4_speedup_vs_single_gpu, n_ranks_seen_by_rccl.  Everything else -- per-stage tables, notes, the c3 / c4 /
c5 child runs, repeats, latencies -- goes to `bench_detail.json` beside this script and to EARLIER stdout
lines that start with `#detail ` (one section per line), never to the last line.

`roofline` describes the dominant kernel of the workload: `traffic` = HBM-side bytes per launch of the
rocprofv3 PMC passes committed under profiles/ (read at run time, never hard-coded), `avg_launch_ms` = its
live HIP-event duration on the library's stream (cpol_enable_timing / cpol_counters), `achieved` = traffic /
avg_launch_ms, `frac` = achieved / 8 TB/s; `timed_region` = the bytes of one whole sweep over ms_per_sweep of
the timed region itself; `valu_f64` = the resource that binds (VALU issue: wave instructions x 4 cycles over
the SIMD cycles of the same duration); `alg_8d` = SURVEY 8(d)'s algorithmic bytes, kept with the note that
the integral tables void it (a sweep gathers 1056 B of coefficients per item, not a 49152-B table slice).
`cpu_baseline` times the CPU oracle (the restatement of the reference algorithm, per radial, un-batched) on
this host: one pinned core (median of 5 samples); the fork-pool legs (radar_operator.py:402,431) run AFTER
the GPU measurements, time-boxed, and are reported in the detail file.
"""
import argparse
import contextlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

RADAR_FIELDS = ['ZH', 'ZV', 'ZDR', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'ATT_H', 'ATT_V']
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s
N_SIMD, PEAK_CLOCK = 1024, 2.4e9  # 256 CUs x 4 SIMDs, 2.4 GHz
VALU_F64_CYCLES = 4               # a wave64 v_*_f64 occupies its SIMD for 4 cycles
LUT_SLICE_BYTES = 1024 * 12 * 4   # B_l per valid item (float32 staging figure)
C4_ELEVATIONS = [0.5, 1.5, 3.0, 5.0, 8.0]


def bench_config(small, workload='c2'):
    rng = 30000 if small else 150000
    conf = {'radar': {'coords': [46.5, 7.5, 1000], 'frequency': 5.6, 'range': rng,
                      'radial_resolution': 300, '3dB_beamwidth': 1., 'K_squared': 0.93,
                      'type': 'ground', 'sensitivity': [-5, 10000]},
            'refraction': {'scheme': 1},
            'integration': {'scheme': 1, 'nh_GH': 1, 'nv_GH': 1, 'weight_threshold': 1.},
            'doppler': {'scheme': 1},
            'microphysics': {'scheme': '1mom', 'with_melting': 0, 'with_ice_crystals': 0,
                             'with_attenuation': 1}}
    if workload in ('c3', 'c4'):
        conf['microphysics'].update(with_melting=1, with_ice_crystals=1)
    if workload == 'c4':
        conf['integration'].update(nh_GH=7, nv_GH=7)
    if workload == 'c5':
        conf['radar']['type'] = 'GPM'
        conf['microphysics'].update(scheme='2mom', with_melting=0, with_ice_crystals=1)
    return conf


def hydrometeors_of(workload):
    if workload == 'c5':
        return ('R', 'S', 'G', 'H', 'I')
    return ('R', 'S', 'G') if workload == 'c2' else ('R', 'S', 'G', 'mS', 'mG', 'I')


def make_inputs(workload, small=False):
    """(configuration, hydrometeors, synthetic COSMO cube, scattering tables) of a BASELINE workload: what main() stages and
    what tests/test_gpu_headline.py builds its operators from -- one definition, so that the test pins the path the line times."""
    from cosmo_pol_amd import synthetic
    conf = bench_config(small, workload)
    hyds = hydrometeors_of(workload)
    cube_h = tuple(h for h in hyds if h in ('R', 'S', 'G', 'H', 'I'))
    n_e = 8 if small else None
    if workload == 'c5':
        cube = (synthetic.small_test_cube(hydrometeors=cube_h, two_moment=True) if small
                else synthetic.make_cube(hydrometeors=cube_h, two_moment=True, **synthetic.BENCH_GRID))
        _sets = {}

        def luts(hl, freq, scheme):                   # (Ku / Ka / C sets, each built once)
            key = (tuple(hl), freq, scheme)
            if key not in _sets:
                _sets[key] = synthetic.make_all_luts(hl, freq, scheme, n_e=n_e)
            return _sets[key]
    elif small:
        cube = synthetic.small_test_cube(hydrometeors=cube_h)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    else:
        cube = synthetic.make_cube(hydrometeors=cube_h, **synthetic.BENCH_GRID)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    return conf, hyds, cube, luts


def load_profile_summary(workload):
    """profiles/r6_<workload>_summary.json (tools/profile_summary.py; r5_* / r4_* / r3_* / r2_* if this round's file is
    missing): per kernel the mean FETCH_SIZE / WRITE_SIZE / SQ counters per dispatch, `hbm_bytes`
    (FETCH_SIZE weighted per kernel as profiles/README.md states, + WRITE_SIZE) and the
    kernel-trace duration."""
    for tag in ('r6', 'r5', 'r4', 'r3', 'r2'):
        path = os.path.join(ROOT, 'profiles', '%s_%s_summary.json' % (tag, workload))
        try:
            with open(path) as f:
                return json.load(f), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None, None


# stage of the launch sequence (cpol_counters_t.ms_*) -> its kernels in the rocprofv3 summaries
STAGE_KERNELS = {'interp': ('k_interp_sweep', 'k_interp_classify', 'k_interp_gate1', 'k_trajectory'),   # (k_interp_classify: the gate kernel that also classifies)
                 'classify': ('k_classify', 'k_ml_weights', 'k_gate1'),          # (k_gate1_species / k_gate1_ray / k_gate1: the single-beam gate kernel)
                 'bucket': ('k_bucket_scan', 'k_bucket_scatter'),
                 'psd': ('k_psd_lookup', 'k_subbeam_sum', 'k_psd_rare'),     # (k_psd_rare: the integrating flavours, one launch, idle in a sweep)
                 'final': ('k_final', 'k_rvel_terms', 'k_ice_first', 'k_scan_rays', 'k_upload_tables')}
TABLE_BUILD_KERNELS = ('k_itab_', 'k_stage_', 'k_spaceborne_first_gate')      # cpol_prepare / cpol_stage_model / the swath's
                                                                              # gate windows (once per swath geometry): not part of a sweep


def stage_rooflines(prof_name, stage_ms, n_sbg, n_valid, n_gates, n_vars, nz, launches_per_run=1):
    """Every stage of ONE sweep against the HBM roofline and against VALU issue.  Per stage: live HIP-event
    duration (cpol_counters_t, one isolated sweep on one lane), HBM-side bytes and VALU wave instructions per
    sweep of its kernels from the committed rocprofv3 PMC passes of the SAME isolated sweep
    (profiles/<prof_name>_summary.json), `frac` = those bytes / live duration / 8 TB/s, `valu_frac` = VALU issue,
    and the ALGORITHMIC bytes of SURVEY 8(d) (`algorithmic_bytes_8d`: interp N_sbg x (4 nz 4 + n_vars 8 4) B,
    psd N_valid x 49152 B, final N_gates x 48 B)."""
    prof, prof_path = load_profile_summary(prof_name)
    alg = {'interp': n_sbg * (4 * nz * 4 + n_vars * 8 * 4), 'psd': n_valid * LUT_SLICE_BYTES,
           'final': n_gates * 48, 'classify': 0, 'bucket': 0}
    if prof and any('k_gate1' in name for name in prof):
        # the single-beam fused kernel (stage `classify`) does the PSD stage's work: SURVEY 8(d)'s bytes of the PSD stage
        # are booked there.  (They are VOID as a roofline: the integral tables replace the 49152-B slice per item by
        # 1056 B of polynomial coefficients, so B_alg / t exceeds the HBM peak many times over; kept for the record.)
        alg['classify'] = alg['psd']
        alg['psd'] = 0
    out = {}
    for st, kernels in STAGE_KERNELS.items():
        ms = stage_ms.get(st)
        traffic = prof_us = valu = None
        used = []
        if prof:
            for name, c in prof.items():
                if name.startswith('_') or not isinstance(c, dict) or not any(k in name for k in kernels):
                    continue
                if c.get('hbm_bytes') is None:
                    continue
                per_sweep = c.get('per_sweep', 1.0)          # dispatches of this kernel per sweep
                traffic = (traffic or 0.0) + c['hbm_bytes'] * per_sweep
                prof_us = (prof_us or 0.0) + (c.get('avg_us') or 0.0) * per_sweep
                if c.get('SQ_INSTS_VALU') is not None:
                    valu = (valu or 0.0) + c['SQ_INSTS_VALU'] * per_sweep
                used.append(name.split('(')[0])
        t = ms * 1e-3 if ms and ms > 0 else None
        out[st] = {'kernels': used, 'live_ms': ms, 'profile_us': prof_us, 'traffic': traffic,
                   'achieved': traffic / t / 1e9 if (t and traffic) else None,
                   'frac': traffic / t / 1e9 / HBM_PEAK_GBS if (t and traffic) else None,
                   'valu_wave_instructions': valu,
                   'valu_frac': valu_issue_frac(valu, t),
                   'algorithmic_bytes_8d': alg[st],
                   'alg_8d_frac': alg[st] / t / 1e9 / HBM_PEAK_GBS if (t and alg[st]) else None}
    return out, prof_path


def valu_issue_frac(wave_instructions, seconds):
    """VALU issue: a wave64 vector instruction occupies its SIMD for 4 cycles (f32, f64 and integer alike on
    CDNA4; transcendentals and f64 division longer: a lower bound) -> instructions x 4 over the cycles of
    1024 SIMDs at 2.4 GHz during `seconds`."""
    if not wave_instructions or not seconds:
        return None
    return wave_instructions * VALU_F64_CYCLES / (N_SIMD * PEAK_CLOCK * seconds)


def roofline_of_dominant_stage(prof_name, stage_ms, n_sbg, n_valid, n_gates, n_vars, nz, note=''):
    """The bench line's `roofline`: the stage with the longest live duration, against HBM (bytes of the committed
    PMC passes over the live duration) and against the resource that binds (`valu_f64`: VALU issue)."""
    stages, prof_path = stage_rooflines(prof_name, stage_ms, n_sbg, n_valid, n_gates, n_vars, nz)
    live = {k: v['live_ms'] for k, v in stages.items() if v['live_ms']}
    if not live:
        return {'bound': 'hbm', 'achieved': None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': None,
                'traffic': None, 'stages': stages}
    dom = max(live, key=live.get)
    d = stages[dom]
    total_ms = sum(live.values())
    total_traffic = sum(v['traffic'] or 0.0 for v in stages.values())
    total_alg = sum(v['algorithmic_bytes_8d'] for v in stages.values())
    total_valu = sum(v['valu_wave_instructions'] or 0.0 for v in stages.values())
    return {'kernel': ' + '.join(d['kernels']) or dom, 'stage': dom, 'bound': 'hbm',
            'achieved': d['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': d['frac'],
            'traffic': d['traffic'], 'traffic_source': prof_path,
            'avg_launch_ms': d['live_ms'], 'profile_avg_us': d['profile_us'],
            'valu_f64': {'bound': 'valu_f64', 'frac': d['valu_frac'],
                         'wave_instructions_per_launch': d['valu_wave_instructions'],
                         'peak_G_wave_instructions_per_s': N_SIMD * PEAK_CLOCK / VALU_F64_CYCLES / 1e9,
                         'whole_sweep_frac': valu_issue_frac(total_valu, total_ms * 1e-3)},
            'alg_8d': {'bytes_per_launch': d['algorithmic_bytes_8d'], 'frac': d['alg_8d_frac'],
                       'whole_sweep_bytes': total_alg,
                       'whole_sweep_frac': total_alg / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       'note': 'SURVEY 8(d) B_alg; void as a roofline where > 1: the integral tables replace the '
                               '49152-B slice per item by 1056 B of coefficients'},
            'whole_sweep': {'live_ms': total_ms, 'traffic': total_traffic,
                            'frac': total_traffic / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if total_traffic else None,
                            'valu_wave_instructions': total_valu,
                            'valu_frac': valu_issue_frac(total_valu, total_ms * 1e-3)},
            'stages': stages,
            'note': ('frac = HBM-side bytes per launch of the PMC passes (profiles/; FETCH_SIZE counts fabric reads, so '
                     'Infinity-Cache hits are included: an upper bound of DRAM traffic) / live HIP-event duration of '
                     'the stage (one isolated sweep, one lane) / 8 TB/s; valu_f64.frac = VALU wave instructions x 4 '
                     'cycles / (1024 SIMDs x 2.4 GHz x the same duration).  ' + note)}


# ------------------------------------------------------------------------ the line the driver parses
COMPACT_LIMIT = 4096


def _r(x, sig=6):
    """Numbers of the compact line: 6 significant digits; NaN / Infinity never reach the line."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        if x != x or x in (float('inf'), float('-inf')):
            return None
        return float('%.*g' % (sig, x))
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def compact_line(d):
    """The final stdout line from the full result dict `d` of a workload: the contract's keys at the top level,
    the roofline of the dominant kernel (HBM + the VALU bound + the timed region), the one-core CPU baseline,
    the SURVEY 8(d) step (`value_host_outputs`), the multi-GPU answers; < COMPACT_LIMIT bytes whatever `d`
    holds (notes and tables stay in bench_detail.json)."""
    cfg = d.get('config') or {}
    roof = d.get('roofline') or {}
    cpu = d.get('cpu_baseline') or None
    ho = d.get('host_outputs') or {}
    out = {k: d.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                                 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    out['config'] = {'workload': str(cfg.get('workload', ''))[:300]}
    for k in ('rays_per_gpu', 'rays_per_gpu_per_sweep', 'gates_per_ray', 'sub_beams', 'lanes', 'sweeps_per_step', 'small'):
        if k in cfg:
            out['config'][k] = cfg[k]
    if isinstance(cfg.get('parallelism'), str):
        out['config']['parallelism'] = cfg['parallelism'][:160]
    r = {k: roof.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source',
                                  'avg_launch_ms')}
    if isinstance(r.get('kernel'), str):
        r['kernel'] = r['kernel'][:80]
    tr = roof.get('timed_region') or {}
    if tr:
        r['timed_region'] = {k: tr.get(k) for k in ('ms_per_sweep', 'traffic_per_sweep', 'achieved', 'frac', 'valu_frac')}
    vf = roof.get('valu_f64') or {}
    if vf:
        r['valu_f64'] = {k: vf.get(k) for k in ('bound', 'frac', 'wave_instructions_per_launch', 'whole_sweep_frac')}
    a8 = roof.get('alg_8d') or {}
    if a8:
        r['alg_8d'] = {'bytes_per_launch': a8.get('bytes_per_launch'), 'frac': a8.get('frac'),
                       'void': bool((a8.get('frac') or 0) > 1.0)}
    out['roofline'] = r
    if cpu:
        out['cpu_baseline'] = {k: cpu.get(k) for k in ('value', 'unit', 'cores', 'kind', 'host_cpus')}
        out['cpu_baseline']['sample'] = str(cpu.get('sample', ''))[:200]
        if d.get('value') and cpu.get('value'):
            out['gpu_over_cpu_core'] = d['value'] / cpu['value']
    if ho:
        out['value_host_outputs'] = ho.get('value')
        out['host_outputs_ms_per_sweep'] = ho.get('ms_per_sweep')
        out['host_outputs_ms_per_sweep_repeats'] = ho.get('ms_per_sweep_repeats')
        out['host_outputs_d2h_bytes_per_step'] = ho.get('d2h_bytes_per_step')
        out['host_outputs_d2h_GBs'] = ho.get('d2h_GBs')
        if (ho.get('with_the_mask_as_one_byte_per_gate') or {}).get('value'):
            out['value_host_outputs_mask_bytes'] = ho['with_the_mask_as_one_byte_per_gate']['value']
        if (d.get('host_outputs_new_geometry') or {}).get('value'):
            out['value_host_outputs_new_geometry'] = d['host_outputs_new_geometry']['value']
        if ho.get('value') and cpu and cpu.get('value'):
            out['host_outputs_over_cpu_core'] = ho['value'] / cpu['value']
    for k in ('ms_per_sweep', 'value_definition', 'c4_speedup_vs_single_gpu', 'c4_speedup_single_volume', 'c4_gather_check',
              'speedup_vs_single_gpu', 'speedup_single_volume', 'gather_check', 'result_check', 'n_ranks_seen_by_rccl', 'collective'):
        if d.get(k) is not None:
            out[k] = d[k][:200] if isinstance(d[k], str) else d[k]
    others = {}
    for k in ('c3', 'c4_volume_one_gpu', 'c5', 'c4_strong_scaling'):
        c = d.get(k)
        if isinstance(c, dict):
            others[k] = ({'value': c.get('value'), 'ms_per_step': c.get('ms_per_step'),
                          'roofline_frac': (c.get('roofline') or {}).get('frac'),
                          'cpu_baseline': (c.get('cpu_baseline') or {}).get('value')}
                         if 'error' not in c else {'error': str(c['error'])[:80]})
    if others:
        out['other_configs'] = others
    out['detail'] = d.get('detail_file', 'bench_detail.json')
    out = _r(out)
    # whatever a caller put into `d`: the line stays below the limit (drop the optional parts first)
    for victim in ('other_configs', 'host_outputs_ms_per_sweep_repeats'):
        if len(json.dumps(out)) < COMPACT_LIMIT:
            break
        out.pop(victim, None)
    if len(json.dumps(out)) >= COMPACT_LIMIT:
        out['config'] = {'workload': out['config']['workload'][:120]}
        out['roofline'] = {k: out['roofline'].get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')}
    return out


def emit(d, detail_path=None):
    """Rank 0: the full result to `bench_detail.json` (beside this script; CPOL_BENCH_DETAIL: another path) and to
    stdout lines that start with '#detail ' (one top-level section per line), THEN the compact line -- the last
    line of stdout, the one the driver parses."""
    path = detail_path or os.environ.get('CPOL_BENCH_DETAIL') or os.path.join(ROOT, 'bench_detail.json')
    d['detail_file'] = os.path.basename(path)
    try:
        with open(path, 'w') as f:
            json.dump(_r(d, 9), f, indent=1)
    except OSError as e:                              # (a read-only checkout: the '#detail' lines still carry everything)
        print('[bench] could not write %s: %s' % (path, e), file=sys.stderr)
    scalars = {k: v for k, v in d.items() if not isinstance(v, (dict, list))}
    print('#detail scalars ' + json.dumps(_r(scalars, 9)))
    for k, v in d.items():
        if isinstance(v, (dict, list)):
            print('#detail %s %s' % (k, json.dumps(_r(v, 9))))
    line = json.dumps(compact_line(d))
    assert len(line) < COMPACT_LIMIT and not line.startswith('#')
    sys.stdout.flush()
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', choices=['auto', 'c2', 'c3', 'c4', 'c5'], default='auto',
                    help="auto: c2 (the metric's configuration; weak scaling at N > 1) followed, at N > 1, by "
                         "the c4 strong-scaling run as an extra; c4 at N > 1: strong scaling as the main line")
    ap.add_argument('--repeats', type=int, default=5, help='repeats of the timed region (c2)')
    ap.add_argument('--small', action='store_true', help='small cube / tables (debugging, tests)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0,
                    help='budget of the one-core CPU-oracle baseline (0 = skip all CPU legs)')
    ap.add_argument('--cpu-pool-child', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_pool_child:
        return cpu_pool_child(args)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run for --gpus > 1')
        args.gpus = world
    workload = args.workload if args.workload != 'auto' else 'c2'
    if args.steps is None:
        args.steps = {'c2': 20, 'c3': 20, 'c4': 10, 'c5': 5}[workload]      # (c2: the driver's own `--steps 20 --warmup 5`; five repeats of the region)
    if args.warmup is None:
        args.warmup = {'c2': 5, 'c3': 3, 'c4': 2, 'c5': 1}[workload]
    if workload in ('c3', 'c5') and world > 1:
        raise SystemExit('--workload %s is a single-GPU extra; the multi-GPU workloads are c4 (strong) and c2 (weak)' % workload)
    # debugging aids for a one-GPU box: CPOL_BENCH_BACKEND=gloo CPOL_BENCH_ONE_DEVICE=1 runs the
    # N-rank code path with every rank on GPU 0 (never used by the driver)
    backend = os.environ.get('CPOL_BENCH_BACKEND', 'nccl')
    if os.environ.get('CPOL_BENCH_ONE_DEVICE'):
        local_rank = 0
    from cosmo_pol_amd import RadarOperator
    t0 = time.time()
    conf, hyds, cube, luts = make_inputs(workload, args.small)
    t_gen = time.time() - t0

    # The one-core CPU baseline first (a bounded sample: --cpu-seconds of oracle work on one pinned core).  The
    # fork-pool legs run in a child interpreter AFTER every GPU measurement of this run: up to
    # 256 NumPy workers leave the host in a state (clocks, page cache, run queues) that the host-bound
    # `host_outputs` step of round 4 was measured in by accident.
    cpu_res = None
    if world == 1 and args.cpu_seconds > 0:
        print('[bench] CPU baseline: one pinned core ...', file=sys.stderr, flush=True)
        if workload == 'c5':
            cpu_res = cpu_baseline_c5(conf, cube, luts, args.cpu_seconds, args.small)
        else:
            el_cpu = 1.0 if workload == 'c2' else C4_ELEVATIONS[2]
            # (c4: a radial of 49 sub-beams takes the oracle about a second: one radial per sample at least)
            cpu_res = cpu_baseline(conf, cube, luts, np.arange(0, 360, 1.0), el_cpu, args.cpu_seconds,
                                   n_samples=5 if workload == 'c2' else 2 if workload == 'c4' else 3)

    # this rank's threads onto the cores next to its GPU (after the CPU legs, which use every core);
    # CPOL_NUMA_BIND=0 leaves the affinity as launched
    from cosmo_pol_amd import bind_to_device_numa_node
    numa = bind_to_device_numa_node(local_rank)
    torch.cuda.set_device(local_rank)
    t0 = time.time()
    with contextlib.redirect_stdout(sys.stderr):      # the operator's notices: stdout carries ONE JSON line
        op = RadarOperator(config=conf, luts=luts, output_variables='only_radar', device=local_rank)
        t_tables = time.time() - t0                   # scattering tables -> HBM + integral tables (cpol_prepare)
        op.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
    op._ctx.synchronize()
    t_stage = time.time() - t0
    t0 = time.time()
    op._ctx.prepare()
    op._ctx.synchronize()
    t_rebuild = time.time() - t0                      # ~0: the tables exist (built inside set_lut)
    # lanes: contexts forked from the operator's (shared cube / tables, own stream and work
    # buffers).  They keep the library's own non-blocking streams, created back to back before
    # any other stream of the process so that each gets its own hardware queue (measured:
    # torch-created streams, or forking after other streams exist, cost 6-25 % through queue
    # sharing).  The process group comes AFTER the lanes for the same reason.
    n_lanes = max(1, int(os.environ.get('CPOL_BENCH_LANES', '3')))
    lanes = [op._lane(i) for i in range(n_lanes)]
    # (c4 runs the product's distributed path, which needs a process group, at N = 1 as well: a one-rank
    # group of the same backend -- RCCL executes on every one-GPU run of the workload)
    # (CPOL_BENCH_FORCE_COLLECTIVES=1: the N > 1 code path of c2 -- process group, all-gather on the side
    # stream, gather check -- with whatever N is, also 1: how a one-GPU box rehearses the RCCL calls)
    force = bool(os.environ.get('CPOL_BENCH_FORCE_COLLECTIVES'))
    grouped = world > 1 or workload == 'c4' or force
    if grouped:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    env = dict(op=op, lanes=lanes, n_lanes=n_lanes, world=world, rank=rank, local_rank=local_rank,
               args=args, cube=cube, conf=conf, torch=torch, dist=dist, workload=workload, luts=luts, force=force)
    out = {'c2': run_c2, 'c3': run_c3, 'c4': run_c4, 'c5': run_c5}[workload](env)
    if rank == 0:
        out['process_group_backend'] = dist.get_backend() if grouped else None
        out['n_ranks_seen_by_rccl'] = int(dist.get_world_size()) if grouped and dist.get_backend() == 'nccl' else None
        out['host_placement'] = {'gpu_pci': numa['pci'], 'numa_node': numa['node'],
                                 'cores_bound': numa['bound'], 'cores_allowed': len(os.sched_getaffinity(0)),
                                 'note': 'rank threads restricted to the cores next to the GPU (0 = affinity left as '
                                         'launched); result blocks are page-locked memory of that node'}
        out['setup_s'] = {'synthetic_inputs': t_gen, 'stage_to_hbm': t_stage,
                          'of_which_scattering_and_integral_tables': t_tables, 'prepare_again': t_rebuild}
        if cpu_res is not None:
            out['cpu_baseline'] = cpu_res
            out['gpu_over_cpu_core'] = out['value'] / cpu_res['value']
    extras = args.workload == 'auto' and not os.environ.get('CPOL_BENCH_NO_EXTRAS')
    port = [None]
    if world > 1:
        if extras:                                    # a rendezvous port for the c4 run that follows
            if rank == 0:
                import socket
                with socket.socket() as sk:
                    sk.bind(('127.0.0.1', 0))
                    port[0] = sk.getsockname()[1]
            dist.broadcast_object_list(port, src=0)
        dist.barrier()
    op.close()
    if grouped:
        dist.destroy_process_group()
    if world > 1 and extras:
        # BASELINE configs[3] in the same run: the C4 volume strong-scaled over the same N GPUs
        # (`bench.py --workload c4 --gpus N`, one child per rank, after this rank's operator and
        # process group are closed)
        small = ['--small'] if args.small else []
        res = child_run('c4', ['--gpus', str(world), '--steps', '10', '--warmup', '2'] + small,
                        env={'MASTER_PORT': str(port[0]), 'TORCHELASTIC_USE_AGENT_STORE': 'False'},
                        quiet=rank != 0, timeout=240)   # (own store on the new port: the launcher's serves the old one)
        if rank == 0:
            out['c4_strong_scaling'] = res
            # the figures that answer north_star's multi-GPU target, at the top level of the line
            out['c4_speedup_vs_single_gpu'] = (res or {}).get('speedup_vs_single_gpu')
            out['c4_speedup_single_volume'] = (res or {}).get('speedup_single_volume')
            out['c4_gather_check'] = (res or {}).get('gather_check')
            out['c4_collective'] = (res or {}).get('collective')
    if rank == 0:
        if world == 1 and extras:
            # the other BASELINE configurations on this GPU, one child process each (after the c2
            # operator is closed): driver-timed like the rest of the run
            small = ['--small'] if args.small else []
            # (each with its own bounded one-core CPU baseline: SURVEY 8(d) for every BASELINE configuration)
            cpu = ['--cpu-seconds', '0' if args.cpu_seconds <= 0 else '8']
            out['c3'] = child_run('c3', ['--steps', '20', '--warmup', '3'] + small + cpu)
            out['c4_volume_one_gpu'] = child_run('c4', ['--steps', '6', '--warmup', '2'] + small + cpu)
            out['c5'] = child_run('c5', ['--steps', '3', '--warmup', '1'] + small + cpu)
            c4 = out['c4_volume_one_gpu'] or {}
            out['c4_speedup_vs_single_gpu'] = c4.get('speedup_vs_single_gpu')         # (N = 1: the path's own overhead)
            out['c4_speedup_single_volume'] = c4.get('speedup_single_volume')
            if out.get('n_ranks_seen_by_rccl') is None:
                out['n_ranks_seen_by_rccl'] = c4.get('n_ranks_seen_by_rccl')
            if args.cpu_seconds > 0 and cpu_res is not None and workload == 'c2':
                # the fork-pool legs, last of all (time-boxed; the line never waits longer than the limit)
                out['cpu_baseline']['all_cores'] = cpu_baseline_pool(workload, args.small)
        emit(out)


def child_run(workload, flags, env=None, quiet=False, timeout=600):
    """Default workload only: `bench.py --workload <workload>` in a child process of this rank (same
    GPU, same rank environment), so that the line also carries the other BASELINE configurations:
    at N = 1 c3, c4 (the single-GPU reference of the strong-scaling runs) and c5; at N > 1 the c4
    strong-scaling run, every rank starting its own child (`quiet`: a rank whose child prints nothing)."""
    import subprocess
    import tempfile
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', workload] + (flags if '--cpu-seconds' in flags else flags + ['--cpu-seconds', '0'])
    t0 = time.time()
    fd, detail = tempfile.mkstemp(prefix='cpol_bench_%s_' % workload, suffix='.json')
    os.close(fd)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout,
                           env=dict(os.environ, CPOL_BENCH_NO_EXTRAS='1', CPOL_BENCH_DETAIL=detail, **(env or {})))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if quiet and r.returncode == 0:
            return None
        if r.returncode != 0 or not line:
            return {'error': (r.stderr or r.stdout)[-400:]}
        try:
            with open(detail) as f:                           # the full result (the child's last line is compact)
                d = json.load(f)
        except (OSError, ValueError):
            d = json.loads(line[-1])
        keep = {k: d.get(k) for k in ('value', 'unit', 'n_gpus', 'ms_per_step', 'steps', 'warmup', 'scaling',
                                      'gather_check', 'speedup_vs_single_gpu', 'single_gpu_same_workload',
                                      'per_rank', 'roofline', 'setup_s', 'stages_ms', 'counters', 'api_ms', 'per_band',
                                      'single_sweep_ms', 'host_submit_ms_per_step', 'submit_loop_ms_per_step', 'speedup_device_only', 'collective', 'host_block_pinned',
                                      'collectives_in_timed_region', 'n_ranks_seen_by_backend', 'n_ranks_seen_by_rccl',
                                      'process_group_backend', 'cpu_baseline', 'speedup_single_volume',
                                      'single_volume_blocking') if k in d}
        keep['workload'] = d['config']['workload']
        keep['child_wall_s'] = time.time() - t0
        keep['command'] = 'python bench.py ' + ' '.join(cmd[2:])
        return keep
    except Exception as e:                                   # the headline line must not depend on an extra
        return {'error': repr(e)[:400]}
    finally:
        try:
            os.unlink(detail)
        except OSError:
            pass


# ------------------------------------------------------------------------------------------ c2
C2_ELEVATIONS = [1.0 + 0.05 * k for k in range(16)]     # one per step, round robin (> the host cache of ray tables)
GEOM_FIELDS = ['lats', 'lons', 'dist', 'heights']


def stage_ms_of(c):
    return {'interp': c.ms_interp, 'classify': c.ms_classify, 'bucket': c.ms_bucket, 'psd': c.ms_psd,
            'final': c.ms_final}


def check_last_sweep(env, slabs, rvel_slabs, els, az, k_last, n_buf, n_lanes, n_cycle):
    """`result_check` of the line (round-5 review, item 1b): what the LAST sweep of the timed region left in HBM -- which launch
    forms it took, read from the library -- against (i) the host-output hand-over of the same elevation (`step_full`'s call:
    NaN pattern equal, <= 1e-5; it is the same arithmetic, so the expected answer is "bitwise") and (ii), when the CPU legs
    are on, the first radials of the one-core `cpu_baseline` leg -- whose oracle results that leg keeps -- against the same
    rays through the product: the six ratio / power variables at the pure relative 1e-5, the three phase-like ones on the
    scale of the variable over the ray, as smoke() does.  The oracle is called by the cpu_baseline leg alone."""
    op, rank, args, torch = env['op'], env['rank'], env['args'], env['torch']
    lane = k_last % n_lanes
    forms = op._lane(lane).launch_forms()
    el = els[(k_last + rank) % n_cycle]
    got = slabs[k_last % n_buf].cpu().numpy()
    got_rvel = rvel_slabs[k_last % n_buf].cpu().numpy()
    with contextlib.redirect_stdout(sys.stderr):
        host = op.simulate_rays(az, el, lane=0)
    out = {'sweep': 'the last of the timed region (lane %d, elevation %.2f deg)' % (lane, float(el[0])), 'launch_forms': forms,
           'vs_host_outputs': {}, 'ok': True}
    bitwise, worst = True, 0.0
    for i, f in enumerate(RADAR_FIELDS):
        a, b = got[i].astype(np.float64), host[f].astype(np.float64)
        same_nan = bool(np.array_equal(np.isnan(a), np.isnan(b)))
        ok = np.isfinite(b) & np.isfinite(a) & (b != 0)
        rel = float(np.max(np.abs(a[ok] - b[ok]) / np.abs(b[ok]))) if ok.any() else 0.0
        bitwise = bitwise and bool(np.array_equal(got[i], host[f], equal_nan=True))
        worst = max(worst, rel)
        out['ok'] = out['ok'] and same_nan and rel <= 1e-5
    rv_same = bool(np.array_equal(got_rvel, host['RVEL'], equal_nan=True))
    bitwise = bitwise and rv_same
    out['ok'] = out['ok'] and rv_same
    out['vs_host_outputs'] = {'bitwise': bitwise, 'worst_rel': worst, 'n_finite_ZH': int(np.isfinite(got[0]).sum()),
                              'RVEL_bitwise': rv_same}
    out['ok'] = out['ok'] and out['vs_host_outputs']['n_finite_ZH'] > 0
    if _CPU_LEG_RAYS:
        # (ii) against the oracle, through what the cpu_baseline leg kept of its first radials (this function never calls the
        # oracle): the same rays at the leg's elevation through the product's blocking call, without the sensitivity cut
        keys = sorted(_CPU_LEG_RAYS)
        with contextlib.redirect_stdout(sys.stderr):
            mine = op.simulate_rays([k[0] for k in keys], [k[1] for k in keys], apply_sensitivity=False, lane=0)
        worst_o, ok_o = {}, True
        for r, key in enumerate(keys):
            ref = _CPU_LEG_RAYS[key]
            for f in RADAR_FIELDS:
                a, b = np.asarray(mine[f][r], dtype=np.float64), np.asarray(ref[f], dtype=np.float64)
                same = bool(np.array_equal(np.isnan(a), np.isnan(b)))
                ok_o = ok_o and same
                fin = np.isfinite(b)
                if not same or not fin.any():
                    continue
                scale = np.abs(b[fin]) if f not in ('KDP', 'PHIDP', 'DELTA_HV') else np.maximum(np.abs(b[fin]), np.max(np.abs(b[fin])))
                with np.errstate(divide='ignore', invalid='ignore'):
                    rel = np.nan_to_num(np.abs(a[fin] - b[fin]) / scale)
                worst_o[f] = max(worst_o.get(f, 0.0), float(rel.max()))
        ok_o = ok_o and all(v <= 1e-5 for v in worst_o.values())
        out['vs_oracle'] = {'rays': ['az %.0f el %.2f' % k for k in keys], 'worst_rel': worst_o, 'ok': ok_o,
                            'note': 'the radials the one-core CPU leg computed first, kept by that leg; the three phase-like '
                                    'variables on the scale of the variable over the ray, the others purely relative'}
        out['ok'] = out['ok'] and ok_o
    return out


def LazyMaskProbe(res):
    """Builds the float64 mask of a pinned result from its one-byte form once more (timing probe of run_c2)."""
    from cosmo_pol_amd.radar_operator import _mask_from_sum
    return _mask_from_sum(res['mask_sum8'], res['n_sub'])()


def run_c2(env):
    from cosmo_pol_amd import RadarOperator
    op, lanes, n_lanes, world, rank = env['op'], env['lanes'], env['n_lanes'], env['world'], env['rank']
    args, torch, dist, cube = env['args'], env['torch'], env['dist'], env['cube']
    az = np.arange(0, 360, 1.0)
    el = np.full(len(az), 1.0)
    els = [np.full(len(az), e) for e in C2_ELEVATIONS]
    n_rays, n_gates = len(az), len(op.constants.RANGE_RADAR)
    dev = torch.device('cuda', env['local_rank'])
    counter = [0]

    # N > 1 = weak scaling of the SAME step: every rank simulates one full sweep per step exactly as
    # at N = 1, rank r taking elevation (k + r) mod 16 of step k (a volume scan spread over the
    # GPUs sweep by sweep).  The sweeps are independent: no collective inside a step.  The timed
    # region ends with the one gather north_star names: each rank's last sweep, left in HBM,
    # all-gathered over RCCL on a side stream.
    weak = world > 1 or env.get('force', False)
    n_buf = max(2, n_lanes)
    slabs = [torch.empty((len(RADAR_FIELDS), n_rays, n_gates), dtype=torch.float32, device=dev)
             for _ in range(n_buf)]
    dev_outs = [{k: sl[i].data_ptr() for i, k in enumerate(RADAR_FIELDS)} for sl in slabs]
    # (round 6: the tenth field, the float64 radial velocity, is left in HBM too -- it was computed all along, into a buffer of
    # the library's own; the nine float32 observables alone travel in the all-gather that ends an N > 1 region)
    rvel_slabs = [torch.empty((n_rays, n_gates), dtype=torch.float64, device=dev) for _ in range(n_buf)]
    for d_o, rv in zip(dev_outs, rvel_slabs):
        d_o['RVEL'] = rv.data_ptr()
    gathered = torch.empty(world * slabs[0].numel(), dtype=torch.float32, device=dev) if weak else None
    lane_streams = [torch.cuda.ExternalStream(l.stream_ptr(), device=dev) for l in lanes] if weak else None
    comm_stream = torch.cuda.Stream() if weak else None
    last_gathered_step = [None]
    n_cycle = 8                               # elevations of the headline step (= the host cache of per-ray constants)
    # cycles of the 8 elevations per step (round 6: 10 -> a step = 80 sweeps, the driver's 20 steps = 1 600 sweeps = ~55 ms of GPU
    # time per repeat instead of 5.5 ms; `ms_per_sweep` stays the figure to compare across rounds)
    cycles_per_step = max(1, int(os.environ.get('CPOL_BENCH_CYCLES_PER_STEP', '2' if args.small else '10')))
    sweeps_per_step = n_cycle * cycles_per_step

    def step_hbm():
        """The headline step: `cycles_per_step` CYCLES of the 8 elevations -- sweeps of the configs[1] PPI, each at another
        elevation than the one before (per-ray constants held by the host, table sets resident on the device), all kernels,
        the ten radar fields of every sweep left in HBM.  (A sweep lasts ~34 us: the driver's 20 steps of one cycle were a
        5.5-ms timed region, which its utilisation sampler never saw; and the one all-gather that ends the region at N > 1
        would outweigh a shorter region.)"""
        for _ in range(sweeps_per_step):
            k = counter[0]
            counter[0] += 1
            op.simulate_rays(az, els[(k + rank) % n_cycle], device_outputs=dev_outs[k % n_buf], lane=k % n_lanes)

    def step_full():
        """The step as the reference's API hands it over (rounds 1-3: the headline; now `host_outputs`): a NEW
        elevation out of 16, its per-ray tables computed on the host and uploaded, all kernels, all 15 output
        arrays to page-locked host memory."""
        k = counter[0]
        counter[0] += 1
        return op.simulate_rays(az, els[(k + rank) % len(els)], pinned=True, lane=k % n_lanes)

    def end_of_region_gather():
        """N > 1: ONE all-gather of every rank's last sweep (in HBM already), behind that sweep's lane."""
        k = counter[0] - 1
        computed = torch.cuda.Event()
        computed.record(lane_streams[k % n_lanes])
        comm_stream.wait_event(computed)
        with torch.cuda.stream(comm_stream):
            dist.all_gather_into_tensor(gathered, slabs[k % n_buf].view(-1))
        last_gathered_step[0] = k

    def step_cached():                      # round 2's headline: fixed geometry, tables resident
        lane = counter[0] % n_lanes
        counter[0] += 1
        return op.simulate_rays(az, el, pinned=True, lane=lane)

    def step_device():
        b = counter[0] % n_buf
        lane = counter[0] % n_lanes
        counter[0] += 1
        op.simulate_rays(az, el, device_outputs=dev_outs[b], lane=lane)

    def fence():
        for i in range(n_lanes):
            op.wait(i)                      # also surfaces a deferred domain error
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, n):
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        t_submit = time.perf_counter() - t0
        if weak:
            end_of_region_gather()
        fence()
        elapsed = time.perf_counter() - t0
        if weak:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, t_submit

    step = step_hbm
    if weak:
        with torch.cuda.stream(comm_stream):     # communicator set-up belongs to the setup phase
            dist.all_gather_into_tensor(gathered, slabs[0].view(-1))
        fence()
    for _ in range(2):                            # set-up: every lane's work buffers exist, the host holds the per-ray
        step()                                    # constants of the n_cycle elevations
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    runs = [timed(step, args.steps) for _ in range(max(1, args.repeats))]
    per_step = sorted(1e3 * e / args.steps for e, _ in runs)
    elapsed = statistics.median(e for e, _ in runs)
    t_submit = statistics.median(s for _, s in runs)
    gates_per_sweep = world * n_rays * n_gates
    gates_per_step = sweeps_per_step * gates_per_sweep
    value = gates_per_step * args.steps / elapsed
    result_check = check_last_sweep(env, slabs, rvel_slabs, els, az, counter[0] - 1, n_buf, n_lanes, n_cycle) if rank == 0 else None

    # rank 0 recomputes the last sweep of every rank and compares it with the gathered block, bit for bit
    gather_ok = None
    if weak:
        if rank == 0:
            blocks = gathered.view(world, -1)
            gather_ok = True
            for r in range(world):
                chk = (last_gathered_step[0] + 1) % n_buf         # (not the slab that was gathered)
                op.simulate_rays(az, els[(last_gathered_step[0] + r) % n_cycle], device_outputs=dev_outs[chk], lane=0)
                op.wait(0)
                gather_ok = gather_ok and bool(torch.equal(torch.nan_to_num(blocks[r]),
                                                           torch.nan_to_num(slabs[chk].view(-1))))
        fence()

    extra = {}
    iso = cnt = None
    d2h_full = n_rays * n_gates * (len(RADAR_FIELDS) * 4 + 8 + 8 + 8 + 8 + 4 + 4)     # 9 observables, RVEL, mask, lats, lons, dist, heights
    d2h_cycle = n_rays * n_gates * (len(RADAR_FIELDS) * 4 + 8 + 8)                    # ... of a repeated geometry: the gate coordinates are the host's
    if not weak:
        # THE SURVEY 8(d) STEP ON A REPEATED SCAN GEOMETRY (round 6; `value_host_outputs`): a volume scan cycles through its
        # elevations -- a radar repeats its scan strategy every few minutes -- so the host holds the per-ray constants of the
        # 8 elevations, the device their tables, and the gate coordinates (lats, lons, dist, heights: functions of site,
        # azimuth and elevation alone) were copied ONCE per table set: every later result carries the same read-only arrays.
        # All 15 arrays are delivered complete; 11 cross PCIe (9 observables, RVEL, the float64 mask: 52 B per gate).
        def step_host_cycle():
            k = counter[0]
            counter[0] += 1
            return op.simulate_rays(az, els[(k + rank) % n_cycle], pinned=True, lane=k % n_lanes)
        op.compact_mask = False                  # (the default: the mask as the device's float64 array -- nothing left to do on the host)
        for _ in range(max(4 * n_cycle, 4 * args.steps)):
            last_host = step_host_cycle()
        fence()
        assert all(k in last_host for k in RADAR_FIELDS + ['RVEL', 'mask', 'lats', 'lons', 'dist', 'heights'])
        assert last_host['mask'].dtype == np.float64 and 'mask_sum8' not in last_host
        settle_c = [timed(step_host_cycle, n_cycle * args.steps) for _ in range(2)]
        runs_c = [timed(step_host_cycle, n_cycle * args.steps) for _ in range(max(1, args.repeats))]
        e_cy = statistics.median(e for e, _ in runs_c)
        n_sw = n_cycle * args.steps
        extra['host_outputs'] = {
            'value': gates_per_sweep * n_sw / e_cy, 'unit': 'gates/s', 'ms_per_sweep': 1e3 * e_cy / n_sw,
            'sweeps_timed': n_sw, 'ms_per_sweep_repeats': [1e3 * e / n_sw for e, _ in runs_c],
            'ms_per_sweep_settling_regions_not_counted': [1e3 * e / n_sw for e, _ in settle_c],
            'host_submit_ms_per_sweep': 1e3 * statistics.median(t for _, t in runs_c) / n_sw,
            'd2h_bytes_per_step': d2h_cycle, 'd2h_GBs': d2h_cycle * n_sw / e_cy / 1e9,
            'arrays_delivered': 15, 'arrays_copied_per_sweep': 11,
            'note': 'SURVEY 8(d) step on a repeated scan geometry: the 8 elevations of the headline step in turn, every result '
                    'handed over as the reference hands it over (all 15 arrays complete in host memory); the gate coordinates '
                    'of a table set are copied once and shared read-only between results; PCIe-bound.  '
                    '`with_the_mask_as_one_byte_per_gate`: RadarOperator.compact_mask = True (the mask crosses PCIe as the sum '
                    'of the sub-beams\' codes and becomes float64 when first read, on the reader\'s thread).  '
                    '`host_outputs_new_geometry`: a NEW geometry every step (rounds 1-5\'s definition)'}
        # the same step with the mask as one byte per gate over PCIe (RadarOperator.compact_mask = True), and what making the
        # float64 mask from its bytes then costs the thread that reads it
        op.compact_mask = True
        for _ in range(2 * n_cycle):
            last_host = step_host_cycle()
        fence()
        runs_f = [timed(step_host_cycle, n_cycle * args.steps) for _ in range(3)]
        op.compact_mask = False
        e_f = statistics.median(e for e, _ in runs_f)
        t0 = time.perf_counter()
        for _ in range(5):
            LazyMaskProbe(last_host)
        extra['host_outputs']['with_the_mask_as_one_byte_per_gate'] = {
            'value': gates_per_sweep * n_sw / e_f, 'ms_per_sweep': 1e3 * e_f / n_sw,
            'd2h_bytes_per_step': d2h_cycle - 7 * n_rays * n_gates,
            'mask_widening_ms_per_sweep_when_read': 1e3 * (time.perf_counter() - t0) / 5}
        del last_host

        # the step with the reference's hand-over (host arrays), PCIe included: rounds 1-3's headline
        op.reuse_device_tables = False          # nothing of the scan geometry stays on the device
        for _ in range(max(2 * n_lanes, 8 * args.steps)):      # (its page-locked blocks and staging slots exist after ~150 sweeps)
            step_full()
        fence()
        # Three timed regions that do not count.  Measured (tools/host_mode_probe.py, profiles/r5_host_mode_probe.txt): in the two
        # synchronisation periods that follow a process's FIRST fence over these sweeps, the first hipMemcpyAsync of every
        # sweep (the 35-KB table upload; the result copy when the upload is made by a kernel) takes ~340 us with ~200 minor page
        # faults, whatever came before (160 or 1 600 warm sweeps, CPU load, a pause); from the third period on it takes 2 us and
        # the step is PCIe-bound.  The HIP runtime's doing (gone under AMD_LOG_LEVEL=4: timing-dependent); round 4's "slow mode".
        settle = [timed(step_full, args.steps) for _ in range(3)]
        runs_h = [timed(step_full, args.steps) for _ in range(max(1, args.repeats))]
        e_h = statistics.median(e for e, _ in runs_h)
        op.reuse_device_tables = True
        extra['host_outputs_new_geometry'] = {
            'value': gates_per_sweep * args.steps / e_h, 'unit': 'gates/s', 'ms_per_sweep': 1e3 * e_h / args.steps,
            'sweeps_timed': args.steps, 'ms_per_sweep_repeats': [1e3 * e / args.steps for e, _ in runs_h],
            'ms_per_sweep_settling_regions_not_counted': [1e3 * e / args.steps for e, _ in settle],
            'host_submit_ms_per_sweep': 1e3 * statistics.median(t for _, t in runs_h) / args.steps,
            'd2h_bytes_per_step': d2h_full, 'd2h_GBs': d2h_full * args.steps / e_h / 1e9,
            'note': 'the same sweep handed over as the reference hands it over: a new elevation out of 16 every step '
                    '(more than the host cache of per-ray tables holds: computed inside the step and uploaded), all '
                    'kernels, all 15 output arrays (9 observables, RVEL, mask, lats, lons, dist, heights) copied to '
                    'page-locked host memory; PCIe-bound; this was `value` in rounds 1-3 and `value_host_outputs` in rounds 4-5'}
        # round 2's headline: fixed elevation, per-ray tables resident, gate coordinates copied once
        for _ in range(2 * n_lanes):
            step_cached()
        e_c, _ = timed(step_cached, args.steps)
        extra['value_cached_geometry'] = gates_per_sweep * args.steps / e_c
        for _ in range(3):
            step_device()
        fence()
        # the PSD stage with three sweeps in flight (events on lane 0 only, 2 per sweep)
        op._ctx.enable_timing(2)
        timed(step_device, args.steps)
        cnt = op._ctx.counters()
        op._ctx.enable_timing(False)
        # ONE sweep at a time on ONE lane, events around every stage: the durations the roofline
        # uses, and what profiles/r3_c2_iso_* profiles (tools/stage_times.py --config c2)
        n_it = max(20, args.steps // 2)
        op._ctx.enable_timing(True)
        for _ in range(n_it):
            op.simulate_rays(az, el, device_outputs=dev_outs[0], lane=0)
        fence()
        iso = op._ctx.counters()
        op._ctx.enable_timing(False)
        # latency of ONE sweep through the API (nothing else in flight)
        lat_rays, lat_ppi = [], []
        for _ in range(7):
            t0 = time.perf_counter()
            op.simulate_rays(az, el)
            lat_rays.append(1e3 * (time.perf_counter() - t0))
        with contextlib.redirect_stdout(sys.stderr):
            for _ in range(5):
                t0 = time.perf_counter()
                op.get_PPI(elevations=[1.0], az_step=1.0)
                lat_ppi.append(1e3 * (time.perf_counter() - t0))
        # refraction scheme 2 at scan scale (host side): the ray paths of a 90-elevation RHI with 3 vertical
        # quadrature nodes = 270 LSODA solves (cosmo_pol_amd/refraction.py), on an exponential refractivity column
        from cosmo_pol_amd import refraction
        zc = np.ascontiguousarray(cube['zlevels'][::-1, cube['zlevels'].shape[1] // 2, cube['zlevels'].shape[2] // 2])
        n_col = (1.0 + 315e-6 * np.exp(-zc.astype(np.float64) / 7350.0)).astype(np.float32)
        rr = np.asarray(op.constants.RANGE_RADAR, dtype=np.float64)
        coords = env['conf']['radar']['coords']
        ode = {}
        for tag, w in (('solved_in_this_process', 0), ('helper_processes_first_call', None), ('helper_processes', None)):
            refraction._SOLVED.clear()
            t0 = time.perf_counter()
            refraction.ode_paths(rr, np.arange(0.5, 90.5, 1.0), np.array([-0.3, 0.0, 0.3]), coords, zc, n_col, workers=w)
            ode[tag] = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        refraction.ode_paths(rr, np.arange(0.5, 90.5, 1.0), np.array([-0.3, 0.0, 0.3]), coords, zc, n_col)
        ode['same_model_state_again'] = 1e3 * (time.perf_counter() - t0)
        ode['helpers'] = len(refraction._POOL)
        refraction._pool_close()
        ode['note'] = ('ms of host time for the 270 ray paths (500 gates each) of a 90-elevation RHI with 3 vertical nodes, '
                       'refraction scheme 2; round 3: 2.1 s (interp1d inside the right-hand side)')
        extra['refraction2_rhi_90x3_ms'] = ode
        extra['single_sweep_latency_ms'] = {
            'simulate_rays_blocking_host_outputs': statistics.median(lat_rays),
            'get_PPI_one_elevation_with_packaging': statistics.median(lat_ppi),
            'device_only_isolated': iso.ms_total,
            'note': 'median wall time of one call with nothing else in flight; get_PPI adds the dB / '
                    'masked-array packaging the reference does on the host too'}
    else:
        op._ctx.enable_timing(True)
        for _ in range(5):
            op.simulate_rays(az, el, device_outputs=dev_outs[0], lane=0)
        fence()
        iso = cnt = op._ctx.counters()
        op._ctx.enable_timing(False)

    if rank != 0:
        return None
    n_valid, n_sbg = int(iso.n_valid_items), int(iso.n_subbeam_gates)
    n_vars, nz = len(op._staged_vars), cube['zlevels'].shape[0]
    roof = roofline_of_dominant_stage(
        'c2_iso', stage_ms_of(iso), n_sbg, n_valid, n_rays * n_gates, n_vars, nz,
        note='c2 sweep at 1.0 deg elevation: %d valid items, all on integral tables (%d).' % (n_valid, int(iso.n_table_items)))
    roof['psd_stage_ms_with_three_lanes_in_flight'] = cnt.ms_psd
    ws = roof.get('whole_sweep') or {}
    if ws.get('traffic'):
        ms_sweep = 1e3 * elapsed / args.steps / sweeps_per_step
        roof['timed_region'] = {
            'ms_per_sweep': ms_sweep, 'traffic_per_sweep': ws['traffic'],
            'achieved': ws['traffic'] / (ms_sweep * 1e-3) / 1e9, 'unit': 'GB/s',
            'frac': ws['traffic'] / (ms_sweep * 1e-3) / (HBM_PEAK_GBS * 1e9),
            'valu_frac': valu_issue_frac(ws.get('valu_wave_instructions'), ms_sweep * 1e-3),
            'note': 'HBM-side bytes / VALU wave instructions of one sweep (the committed PMC passes of the isolated sweep) '
                    'over the time per sweep of the timed region itself (sweeps of three lanes in flight)'}
    if not weak:
        # the integrating kernel itself (it builds the integral tables at staging time and takes the
        # items outside them): a second operator with the tables switched off, one lane
        os.environ['CPOL_ITAB'] = '0'
        try:
            with contextlib.redirect_stdout(sys.stderr):
                op2 = RadarOperator(config=env['conf'], luts=env['luts'], output_variables='only_radar',
                                    device=env['local_rank'], lanes=1)
                op2.load_model_arrays(cube['data'], cube['zlevels'], cube['proj_info'], cube['resolution'])
            for _ in range(3):
                op2.simulate_rays(az, el, device_outputs=dev_outs[0])
            op2.wait()
            op2._ctx.enable_timing(True)
            for _ in range(max(10, args.steps // 4)):
                op2.simulate_rays(az, el, device_outputs=dev_outs[0])
            op2.wait()
            c2 = op2._ctx.counters()
            op2.close()
        finally:
            del os.environ['CPOL_ITAB']
        roof['integrating_kernel'] = roofline('c2_direct', 'k_psd_uniform<false>', c2.ms_psd, int(c2.n_valid_items),
                                              int(c2.n_valid_items) * LUT_SLICE_BYTES)
        roof['integrating_kernel']['sweep_device_total_ms'] = c2.ms_total
        roof['integrating_kernel']['note'] = (
            'CPOL_ITAB=0: every item integrated over its 1024 diameter bins by k_psd_uniform (one lane, '
            'isolated) -- the kernel that evaluates the table nodes at staging time and finishes the items '
            'outside the tables; f64-VALU bound: ' + roof['integrating_kernel']['note'])
    out = {
        'metric': 'range-gates/sec', 'value': value, 'unit': 'gates/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'ms_per_sweep': 1e3 * elapsed / args.steps / sweeps_per_step,
        'sweeps_per_step': sweeps_per_step, 'ms_per_cycle_of_8_sweeps': 1e3 * elapsed / args.steps / cycles_per_step,
        'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'value_definition': 'v2 (round 4 on): sweeps at 8 elevations in turn (round 6: 80 per step), inputs and outputs resident in HBM; '
                            'value_host_outputs = the SURVEY 8(d) step on a repeated scan geometry (round 6: all 15 arrays '
                            'delivered complete, 11 copied per sweep: 52 B per gate); value_host_outputs_new_geometry = a new geometry every step '
                            '(v1: `value` in rounds 1-3, `value_host_outputs` in rounds 4-5)',
        'config': {'workload': 'c2: step = %d sweeps per GPU (%d cycles of 8 elevations) of the 360-az x 500-gate C-band PPI (R+S+G 1-mom, 1 sub-beam), '
                               'inputs and outputs resident in HBM; synthetic %s cube; each sweep = one cpol_run_sweep at one of '
                               '8 elevations (1.0 + 0.05 k deg, k = (sweep + rank) mod 8; per-ray constants on the host and '
                               'table sets on the device before the timed region), every kernel of the launch sequence, the '
                               '10 radar fields (9 float32 observables + the float64 radial velocity) left in HBM; the SURVEY 8(d) '
                               'step with host outputs: value_host_outputs%s'
                               % (sweeps_per_step, cycles_per_step, 'x'.join(map(str, cube['zlevels'].shape)),
                                  '' if not weak else '; the timed region ends with one all-gather of every rank\'s '
                                                      'last sweep (its 9 float32 observables, device to device)'),
                   'rays_per_gpu': n_rays, 'gates_per_ray': n_gates, 'lanes': n_lanes, 'sweeps_per_step': sweeps_per_step,
                   'parallelism': ('weak scaling: the sweeps of a scan are independent, rank r simulates elevation '
                                   '(sweep + r) mod 8 with the N = 1 step unchanged, no collective inside a step, ONE '
                                   'all-gather (RCCL) at the end of the timed region') if weak else 'single GPU',
                   'small': bool(args.small)},
        'timed_region_repeats': {'n': len(runs), 'ms_per_step_min': per_step[0],
                                 'ms_per_step_median': statistics.median(per_step),
                                 'ms_per_step_max': per_step[-1],
                                 'note': 'each repeat = exactly `steps` steps between two fences; '
                                         '`value` and `ms_per_step` are the median repeat'},
        'roofline': roof,
        'stages_ms': {'interp': iso.ms_interp, 'classify': iso.ms_classify,
                      'bucket': iso.ms_bucket, 'psd': iso.ms_psd, 'final': iso.ms_final,
                      'device_total': iso.ms_total, 'launches_per_sweep': 3,
                      'note': 'one sweep at a time on one lane, HIP events around every stage (the pass after the timed '
                              'region); single-beam fast path of a context with lanes: k_interp_sweep, k_gate1_ray '
                              '(`classify`: items outside the tables integrated in place, no integrating launch), '
                              'k_scan_rays (`final`: range scans); `bucket` and `psd` hold no kernel'},
        'counters': {'n_subbeam_gates': n_sbg, 'n_valid_items': n_valid,
                     'n_work_units': int(iso.n_work_units), 'n_table_items': int(iso.n_table_items)},
        'host_submit_ms_per_step': 1e3 * t_submit / args.steps,
        'host_submit_ms_per_sweep': 1e3 * t_submit / args.steps / sweeps_per_step,
        'gather_check': gather_ok,
        'result_check': None if result_check is None else bool(result_check['ok']),
        'result_check_detail': result_check,
    }
    out.update(extra)
    return out


def roofline(workload, kernel, ms_stage, n_valid, psd_bytes, ms_isolated=None, valu_scale=1.0):
    """The PSD x table stage against its bound.  It is f64-VALU bound: the
    LUT slices are shared by the items of a work unit through the scalar cache, so the HBM
    figure the survey defines (B_alg = N_valid x 49152 B over the kernel time) exceeds the
    HBM peak and is kept as `hbm_alg_frac` only.  frac = VALU wave-instructions per launch
    (SQ_INSTS_VALU of the committed PMC pass; all f64, 4 cycles each on a SIMD) over the
    instruction slots of 1024 SIMDs at 2.4 GHz during the live kernel time."""
    prof, prof_path = load_profile_summary(workload)
    valu = traffic = prof_us = None
    kernels = kernel if isinstance(kernel, (list, tuple)) else [kernel]
    if prof:
        for want in kernels:
            for name, c in prof.items():
                if name.startswith('_') or want not in name or 'SQ_INSTS_VALU' not in c:
                    continue
                valu = (valu or 0) + c['SQ_INSTS_VALU'] * valu_scale
                traffic = (traffic or 0) + c.get('hbm_bytes', 0) * valu_scale
                prof_us = (prof_us or 0) + c.get('avg_us', 0)
                break
    kernel = ' + '.join(kernels)
    analytic = -(-n_valid // 64) * 1024 * 16          # recurrence flavour: 16 v_*_f64 per (item row, bin)
    peak = N_SIMD * PEAK_CLOCK / VALU_F64_CYCLES / 1e9     # G wave-instructions / s
    t = ms_stage * 1e-3 if ms_stage and ms_stage > 0 else None
    n_inst = valu if valu else analytic
    achieved = n_inst / t / 1e9 if t else None
    r = {'kernel': kernel + ' (the PSD x table stage; 1 launch / sweep in c2, 3 flavours in c4)',
         'bound': 'valu_f64', 'achieved': achieved, 'peak': peak, 'unit': 'G wave-instructions/s',
         'frac': achieved / peak if achieved else None,
         'traffic': traffic,
         'valu_wave_instructions_per_launch': n_inst,
         'valu_source': prof_path if valu else 'analytic: ceil(N_valid/64) x 1024 bins x 16 f64 ops',
         'essential_fma_frac': (-(-n_valid // 64) * 1024 * 12 / t / 1e9 / peak) if t else None,
         'avg_stage_ms': ms_stage,
         'profile_avg_us': prof_us,
         'hbm_alg_frac': (psd_bytes / t / 1e9 / HBM_PEAK_GBS) if t else None,
         'hbm_physical_frac': (traffic / t / 1e9 / HBM_PEAK_GBS) if (t and traffic) else None,
         'algorithmic_bytes_per_launch': psd_bytes,
         'note': 'bound = f64 VALU issue (SIMD instruction slots); hbm_alg_frac is the survey\'s '
                 'B_alg / t / 8 TB/s (> 1 because LUT slices are shared on chip); traffic = HBM bytes '
                 'per launch from the PMC passes in profiles/ ((2 FETCH_SIZE + WRITE_SIZE) KiB)'}
    if ms_isolated:
        ti = ms_isolated * 1e-3
        r['isolated'] = {'avg_stage_ms': ms_isolated, 'frac': n_inst / ti / 1e9 / peak,
                         'essential_fma_frac': -(-n_valid // 64) * 1024 * 12 / ti / 1e9 / peak,
                         'note': 'same sweep with one lane only (no overlap with other sweeps)'}
    return r


# ------------------------------------------------------------------------------------------ c3
def run_c3(env):
    """BASELINE configs[2] on one GPU: 5 elevations x (360 x 500), R,S,G,mS,mG,I, one sub-beam.  A step
    = one volume through the C ABI, a volume scan that REPEATS (round 6; rounds 1-5 re-uploaded the per-ray tables and
    copied the gate coordinates with every sweep): the per-ray tables of the five elevations stay on the device,
    per sweep the kernels run and 11 arrays (9 observables, RVEL, the mask) go down into
    page-locked host memory, the gate coordinates of the unchanged geometry come from the host's cache -- all 15
    arrays are delivered; every sweep on a lane of its own (5 lanes), so that the arrays of a volume stay valid
    until the next volume starts.  `api_ms`: the same volume through
    RadarOperator.get_PPI (the drop-in call: the five sweeps queued on three lanes the same way, one wait, + dB fields,
    masked arrays, scan container -- built on first access)."""
    op, args, torch, cube = env['op'], env['args'], env['torch'], env['cube']
    az = np.arange(0, 360, 1.0 if not args.small else 4.0)
    n_el, n_rays, n_gates = len(C4_ELEVATIONS), len(az), len(op.constants.RANGE_RADAR)
    els = [np.full(n_rays, e) for e in C4_ELEVATIONS]
    for i in range(n_el):
        op._lane(i)
    def volume():
        # (one call per sweep, each on a lane of its own: the 9.4 MB copy of one sweep overlaps the kernels
        # of the next.  The five sweeps as ONE launch sequence -- get_PPI's form for scans with sub-beams, and for this scan
        # until round 6 -- give one 47 MB copy per volume that overlaps nothing: 1.37 against 0.93 ms per volume in this loop)
        return [op.simulate_rays(az, els[e], pinned=True, lane=e) for e in range(n_el)]

    def fence():
        for i in range(n_el):
            op.wait(i)
        torch.cuda.synchronize()

    for _ in range(2):
        volume()
    fence()
    for _ in range(args.warmup):
        volume()
    fence()
    # (five timed regions of `steps` volumes, the median: the single 30-ms region of rounds 5-6 gave 1.51 and 2.10 ms per volume on two
    # boxes with identical device times)
    # The first regions are not the step either: the results of a region are all in flight until its fence, so the operator's pool of
    # page-locked blocks grows to `steps` volumes' worth (hipHostMalloc, and the first copy into a new block blocks the submitting
    # thread for its whole duration: 298 us per sweep, tools/c3_settle.py) -- three settling regions, as c2's host-output step has.
    regions, settling = [], []
    for r in range(8):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            volume()
        t_sub = time.perf_counter() - t0
        fence()
        (settling if r < 3 else regions).append((time.perf_counter() - t0, t_sub))
    elapsed, t_submit = sorted(regions)[len(regions) // 2]
    gates = n_el * n_rays * n_gates
    # one sweep at a time, events around every stage (3 deg: what profiles/r3_c3_el3_iso_* profiles)
    slab = torch.empty((len(RADAR_FIELDS), n_rays, n_gates), dtype=torch.float32, device='cuda')
    ptrs = {k: slab[i].data_ptr() for i, k in enumerate(RADAR_FIELDS)}
    per_sweep, iso3 = [], None
    for e in range(n_el):
        for _ in range(2):
            op.simulate_rays(az, els[e], device_outputs=ptrs, lane=0)
        op.wait(0)
        op._ctx.enable_timing(True)
        for _ in range(5):
            op.simulate_rays(az, els[e], device_outputs=ptrs, lane=0)
        op.wait(0)
        c = op._ctx.counters()
        op._ctx.enable_timing(False)
        per_sweep.append({'elevation': C4_ELEVATIONS[e], 'device_ms': c.ms_total, 'n_valid_items': int(c.n_valid_items),
                          'n_table_items': int(c.n_table_items)})
        if C4_ELEVATIONS[e] == 3.0:
            iso3 = c
    api = []
    with contextlib.redirect_stdout(sys.stderr):
        op.lanes = 3
        for _ in range(1 if args.small else 12):      # (the pool of page-locked blocks of these calls settles, as above)
            op.get_PPI(C4_ELEVATIONS, azimuths=az)
        for _ in range(max(3, args.steps // 2)):
            t0 = time.perf_counter()
            op.get_PPI(C4_ELEVATIONS, azimuths=az)
            api.append(1e3 * (time.perf_counter() - t0))
    n_vars, nz = len(op._staged_vars), cube['zlevels'].shape[0]
    roof = roofline_of_dominant_stage('c3_el3_iso', stage_ms_of(iso3), int(iso3.n_subbeam_gates), int(iso3.n_valid_items),
                                      n_rays * n_gates, n_vars, nz, note='c3 sweep at 3 deg elevation.')
    # (9 observables, RVEL, the float64 mask; the gate coordinates of the repeated geometry are the host's: not copied)
    d2h = gates * (len(RADAR_FIELDS) * 4 + 8 + (1 if getattr(op, 'compact_mask', False) else 8))       # (compact_mask: off by default)
    return {
        'metric': 'range-gates/sec', 'value': gates * args.steps / elapsed, 'unit': 'gates/s', 'n_gpus': 1,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'c3: 5-elevation volume %s, %d x %d gates each, R,S,G,mS,mG,I 1-moment + melting layer, '
                               '1 sub-beam; one volume per step through the C ABI, the scan repeating: kernels, all 15 output '
                               'arrays of every sweep delivered in host memory (11 copied, the gate coordinates of the '
                               'unchanged geometry from the cache); 5 lanes'
                               % (C4_ELEVATIONS, n_rays, n_gates),
                   'd2h_bytes_per_step': d2h, 'lanes': n_el, 'small': bool(args.small)},
        'roofline': roof,
        'stages_ms': dict(stage_ms_of(iso3), device_total=iso3.ms_total, sweep='3 deg'),
        'single_sweep_ms': per_sweep,
        'api_ms': {'get_PPI_volume_median': statistics.median(api), 'get_PPI_volume_min': min(api),
                   'note': 'RadarOperator.get_PPI(5 elevations): the same volume through the drop-in call (a single-beam scan: '
                           'its sweeps queued on the operator\'s lanes with page-locked outputs, one wait at the end -- round 6; '
                           'rounds 4-5: ONE launch sequence and one copy, 1.37 ms; gate coordinates of the unchanged geometry '
                           'from the cache, RadarScan with fields built on first access)'},
        'host_submit_ms_per_step': 1e3 * t_submit / args.steps,
        'ms_per_step_repeats': [round(1e3 * r[0] / args.steps, 4) for r in regions],
        'ms_per_step_settling_regions_not_counted': [round(1e3 * r[0] / args.steps, 4) for r in settling],
    }


# ------------------------------------------------------------------------------------------ c5
def run_c5(env):
    """BASELINE configs[4] on one GPU: GPM-DPR Ku (200 scans x 49 rays, 125 m gates) and Ka (200 x 25,
    250 m) swaths over the 2-moment cube (R,S,G,H,I) through RadarOperator.get_GPM_swath -- each call
    switches the table set to the band's frequency (tables of a set seen before stay resident).  A step
    = both swaths; gates = the gates kept (below 35 km, above the topography of the reference's cut)."""
    from cosmo_pol_amd import gpm
    op, args, torch = env['op'], env['args'], env['torch']
    n_scans = 200 if not args.small else 12
    centre = (46.5, 7.5)
    swaths = {'Ku': gpm.synthetic_swath(n_scans=n_scans, n_rays=49, centre=centre, cross_track_deg=17.0,
                                        scan_spacing_m=3000.0 if not args.small else 1500.0),
              'Ka': gpm.synthetic_swath(n_scans=n_scans, n_rays=25, centre=centre, cross_track_deg=8.5,
                                        scan_spacing_m=3000.0 if not args.small else 1500.0)}
    if args.small:
        for sw in swaths.values():                  # keep the footprints inside the small test cube
            for k in ('Latitude', 'Longitude'):
                c0 = centre[0] if k == 'Latitude' else centre[1]
                sw[k] = c0 + (sw[k] - c0) * 0.12
    per_band, first = {}, {}
    with contextlib.redirect_stdout(sys.stderr):
        for band, sw in swaths.items():
            t0 = time.perf_counter()
            out = op.get_GPM_swath(sw, band)         # first call: this band's tables are built and staged
            first[band] = time.perf_counter() - t0
            per_band[band] = {'rays': int(out.azimuths.size), 'kept_gates': int(np.sum(out.n_kept)),
                              'finite_ZH': int(np.isfinite(out.raw['ZH']).sum()), 'first_call_s': first[band]}
        for _ in range(args.warmup):
            for band, sw in swaths.items():
                op.get_GPM_swath(sw, band)
        torch.cuda.synchronize()
        t_band = {b: 0.0 for b in swaths}
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for band, sw in swaths.items():
                t1 = time.perf_counter()
                op.get_GPM_swath(sw, band)
                t_band[band] += time.perf_counter() - t1
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    # one Ku swath at a time, HIP events around every stage of its launch sequence (what profiles/r4_c5_ku_iso_* profiles:
    # tools/stage_times.py --config c5)
    iso = None
    with contextlib.redirect_stdout(sys.stderr):
        op.get_GPM_swath(swaths['Ku'], 'Ku')
        op._ctx.enable_timing(True)
        for _ in range(3):
            op.get_GPM_swath(swaths['Ku'], 'Ku')
        iso = op._ctx.counters()
        op._ctx.enable_timing(False)
    for b in swaths:
        per_band[b]['swath_ms'] = 1e3 * t_band[b] / args.steps
        per_band[b]['gates_per_s'] = per_band[b]['kept_gates'] * args.steps / t_band[b]
    gates = sum(v['kept_gates'] for v in per_band.values())
    return {
        'metric': 'range-gates/sec', 'value': gates * args.steps / elapsed, 'unit': 'gates/s', 'n_gpus': 1,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'c5: GPM-DPR Ku (%d x 49 rays, 125 m gates) + Ka (%d x 25, 250 m) swaths over the 2-moment '
                               'cube (R,S,G,H,I) through RadarOperator.get_GPM_swath (blocking host outputs, SimulatedGPM '
                               'packaging included); one step = both swaths; gates = gates kept below 35 km'
                               % (n_scans, n_scans), 'small': bool(args.small)},
        'roofline': roofline_of_dominant_stage(
            'c5_ku_iso', stage_ms_of(iso), int(iso.n_subbeam_gates), int(iso.n_valid_items), int(iso.n_subbeam_gates),
            len(op._staged_vars), env['cube']['zlevels'].shape[0],
            note='c5: the launch sequence of ONE Ku swath (%d rays x %d gates, one sub-beam, 2-moment R,S,G,H,I: the '
                 'single-beam fast path), events around every stage; `value` itself is an API-level figure (blocking '
                 'host outputs, SimulatedGPM packaging).' % (per_band['Ku']['rays'], int(iso.n_subbeam_gates) // max(1, per_band['Ku']['rays']))),
        'stages_ms': dict(stage_ms_of(iso), device_total=iso.ms_total, swath='Ku'),
        'per_band': per_band,
    }


# ------------------------------------------------------------------------------------------ c4
def run_c4(env):
    """Strong scaling of the C4 volume THROUGH THE PRODUCT'S DISTRIBUTED PATH: RadarOperator.submit_volume
    (what RadarOperator(distributed=True).get_PPI runs) -- the azimuths of every sweep sharded over the
    ranks, a rank's rays of all five sweeps as ONE launch sequence, ONE collective per volume
    (cosmo_pol_amd/distributed.py::ShardedVolumeRunner: a gather to rank 0, who alone assembles the volume
    and copies it to page-locked host memory; CPOL_BENCH_C4_GATHER=all: an all-gather, every rank receives
    and copies), nothing waits: consecutive volumes alternate over the lanes, so that the collective and
    the copy of volume k overlap the kernels of volume k + 1.  The nine polarimetric observables travel
    (36 B per gate); `api_ms.get_PPI_distributed` times the blocking drop-in call with all 15 arrays."""
    import collections
    from cosmo_pol_amd.distributed import VolumeLayout
    op, lanes, n_lanes, world, rank = env['op'], env['lanes'], env['n_lanes'], env['world'], env['rank']
    args, torch, dist, cube = env['args'], env['torch'], env['dist'], env['cube']
    dev = torch.device('cuda', env['local_rank'])
    az_all = np.arange(0, 360, 1.0 if not args.small else 4.0)
    n_az, n_gates = len(az_all), len(op.constants.RANGE_RADAR)
    n_el = len(C4_ELEVATIONS)
    sweeps = [(az_all, np.full(n_az, e)) for e in C4_ELEVATIONS]
    fields = [(k, np.float32) for k in RADAR_FIELDS]
    lay = VolumeLayout(fields, [n_az] * n_el, world, n_gates)
    az, el = lay.local_rays(rank, sweeps)
    n_loc = len(az) // n_el
    op.lanes = n_lanes                                  # (the runner keeps lanes + 1 sets of device buffers)
    # (round 6: the all-gather is the default -- the rooted `dist.gather` has run on one rank only so far; CPOL_BENCH_C4_GATHER=root
    # opts into it, tests/test_gpu_zz_two_gpus.py run both forms wherever two GPUs are visible)
    op.gather_to = 0 if os.environ.get('CPOL_BENCH_C4_GATHER', 'all') == 'root' else None
    runner = op._dist_runner()
    pending = collections.deque()
    last = [None]
    counter = [0]

    def volume():
        k = counter[0]
        counter[0] += 1
        pending.append(op.submit_volume(sweeps, fields=RADAR_FIELDS, lane=k % n_lanes))
        while len(pending) > n_lanes:                   # results older than the lanes in flight: complete by now
            last[0] = pending.popleft().wait() or last[0]

    def fence():
        while pending:
            last[0] = pending.popleft().wait() or last[0]
        for i in range(n_lanes):
            op.wait(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    volume()                                            # communicator set-up is setup, not a step
    fence()
    for _ in range(max(args.warmup, n_lanes)):          # (every lane's work buffers exist)
        volume()
    fence()
    n_coll0 = runner.n_collectives
    t0 = time.perf_counter()
    for _ in range(args.steps):
        volume()
    t_submit = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    n_coll = runner.n_collectives - n_coll0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    gates_per_step = n_el * n_az * n_gates
    value = gates_per_step * args.steps / elapsed
    gathered_volume = last[0]                           # rank 0 (or every rank): the last volume of the timed region

    # ONE VOLUME AT A TIME through the same path (what a single get_PPI(distributed=True) sees): submit, wait, next
    def blocking_region(group, n):
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            op.submit_volume(sweeps, fields=RADAR_FIELDS, lane=0, group=group).wait()
        op.wait(0)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    op.submit_volume(sweeps, fields=RADAR_FIELDS, lane=0).wait()
    t_block = blocking_region(None, args.steps)
    if world > 1:
        tt = torch.tensor([t_block], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_block = float(tt.item())

    # THE SAME TWO REGIMES ON ONE GPU: rank 0 alone, through the same code path on a one-rank group (the other ranks
    # wait at the fence).  Numerator and denominator of both speedups are then like for like: distributed path,
    # 9 fields gathered, assembled on the device and copied to page-locked host memory; pipelined over the lanes
    # (`speedup_vs_single_gpu`), or one volume at a time (`speedup_single_volume`).
    one = None
    solo = dist.new_group([0]) if world > 1 else None    # (every rank makes the call; N = 1: the default group is it)
    if rank == 0:
        n_it = max(3, args.steps // 2)
        pend = collections.deque()
        for k in range(n_lanes + 1):                     # buffers of the one-rank layout, every lane
            pend.append(op.submit_volume(sweeps, fields=RADAR_FIELDS, lane=k % n_lanes, group=solo))
        while pend:
            pend.popleft().wait()
        for i in range(n_lanes):
            op.wait(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_it):
            pend.append(op.submit_volume(sweeps, fields=RADAR_FIELDS, lane=k % n_lanes, group=solo))
            while len(pend) > n_lanes:
                pend.popleft().wait()
        while pend:
            pend.popleft().wait()
        for i in range(n_lanes):
            op.wait(i)
        torch.cuda.synchronize()
        t1_pipe = (time.perf_counter() - t0) / n_it
        t0 = time.perf_counter()
        for _ in range(n_it):
            op.submit_volume(sweeps, fields=RADAR_FIELDS, lane=0, group=solo).wait()
        op.wait(0)
        torch.cuda.synchronize()
        t1_block = (time.perf_counter() - t0) / n_it
        one = {'pipelined_ms_per_volume': 1e3 * t1_pipe, 'single_volume_ms': 1e3 * t1_block, 'volumes_timed': n_it,
               'note': 'rank 0 alone runs the whole volume through the same distributed path on a one-rank group: '
                       '%d lanes alternating (pipelined) and submit().wait() per volume (single volume)' % n_lanes}
    fence()

    # the blocking drop-in call on every rank: RadarOperator(distributed=True).get_PPI, all 15 arrays
    op.distributed = True
    api = []
    with contextlib.redirect_stdout(sys.stderr):
        op.get_PPI(C4_ELEVATIONS, azimuths=az_all)
        for _ in range(max(3, args.steps // 3)):
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            op.get_PPI(C4_ELEVATIONS, azimuths=az_all)
            api.append(1e3 * (time.perf_counter() - t0))
    op.distributed = False
    fence()
    ptrs0 = None
    if n_loc > 0:
        blk0 = torch.zeros(lay.block.nbytes, dtype=torch.uint8, device=dev)
        ptrs0 = {k: blk0.data_ptr() + lay.block.offsets[k] for k, _ in fields}

    # per-rank work of one volume (untimed pass, one lane, events around every stage)
    mine = {'rank': rank, 'rays_per_sweep': n_loc}
    iso = None
    if n_loc > 0:
        op._ctx.enable_timing(True)
        for _ in range(3):
            op.simulate_rays(az, el, device_outputs=ptrs0, lane=0)
        op.wait(0)
        iso = op._ctx.counters()
        op._ctx.enable_timing(False)
        mine.update(n_valid_items=int(iso.n_valid_items), n_table_items=int(iso.n_table_items),
                    n_work_units=int(iso.n_work_units), volume_device_ms_isolated=float(iso.ms_total),
                    stages_ms=stage_ms_of(iso))
    per_rank = [None] * world
    if world > 1:
        dist.all_gather_object(per_rank, mine)
    else:
        per_rank = [mine]

    # rank 0 alone: the same volume on ONE GPU (the strong-scaling reference), and the check
    # that the gathered volume equals it bit for bit
    single = gather_ok = None
    fence()
    if rank == 0:
        lay1 = VolumeLayout(fields, [n_az] * n_el, 1, n_gates)
        full = torch.zeros(lay1.block.nbytes, dtype=torch.uint8, device=dev)
        fptrs = {k: full.data_ptr() + lay1.block.offsets[k] for k, _ in fields}
        az1, el1 = lay1.local_rays(0, sweeps)
        n_it = max(2, args.steps // 3)
        op.simulate_rays(az1, el1, device_outputs=fptrs, lane=0)
        op.wait(0)
        t0 = time.perf_counter()
        for _ in range(n_it):
            op.simulate_rays(az1, el1, device_outputs=fptrs, lane=0)
        op.wait(0)
        t1 = (time.perf_counter() - t0) / n_it
        single = {'ms_per_volume': 1e3 * t1, 'value': gates_per_step / t1,
                  'through_the_distributed_path': one,
                  'note': 'rank 0 runs the whole 5 x %d-ray volume alone after the timed region: ms_per_volume / value = '
                          'one launch sequence at a time with the outputs left in HBM (device work only; NOT the '
                          'denominator of the speedups); through_the_distributed_path = the regimes of the speedups' % n_az}
        ref = lay1.assemble(full.cpu().numpy())
        got = gathered_volume
        gather_ok = bool(got is not None and all(np.array_equal(got[s][k], ref[s][k], equal_nan=True)
                                                 for s in range(n_el) for k, _ in fields))
    fence()
    if rank != 0:
        return None
    busiest = max((q for q in per_rank if q.get('n_valid_items')), key=lambda q: q['n_valid_items'])
    n_vars, nz = len(op._staged_vars), cube['zlevels'].shape[0]
    n_sbg_b = n_el * busiest['rays_per_sweep'] * 49 * n_gates
    roof = roofline_of_dominant_stage(
        'c4_volume_iso' if world == 1 else 'c4_share%d_iso' % world, busiest['stages_ms'], n_sbg_b,
        busiest['n_valid_items'], n_el * busiest['rays_per_sweep'] * n_gates, n_vars, nz,
        note='c4: the five-elevation launch sequence of the busiest rank (%d rays per sweep), one lane, the pass '
             'after the timed region; traffic from the profile of the same launch sequence when committed '
             '(N = 1: profiles/*_c4_volume_iso_*; N = 8: *_c4_share8_iso_*).' % busiest['rays_per_sweep'])
    return {
        'metric': 'range-gates/sec', 'value': value, 'unit': 'gates/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'c4: 5-elevation volume %s, %d x %d gates each, R,S,G,mS,mG,I 1-moment + melting '
                               'layer, 7x7 Gauss-Hermite sub-beams (49 per radial), synthetic %s cube; one '
                               'volume per step; sub-beam gates are not counted as gates'
                               % (C4_ELEVATIONS, n_az, n_gates, 'x'.join(map(str, cube['zlevels'].shape))),
                   'rays_per_gpu_per_sweep': lay.bounds[0][0][1] - lay.bounds[0][0][0], 'gates_per_ray': n_gates,
                   'sub_beams': 49, 'lanes': n_lanes,
                   'parallelism': 'RadarOperator.submit_volume (the path of RadarOperator(distributed=True).get_PPI): '
                                  'azimuths of every sweep sharded in contiguous blocks of ceil(360/N) rays; a rank runs '
                                  'its rays of all 5 sweeps as one launch sequence (consecutive volumes on alternating '
                                  'lanes); ONE collective of the device blocks per volume (%s); the receiving rank puts '
                                  'the rows into scan order on the device and copies the volume (9 float32 fields) to '
                                  'page-locked host memory' % runner.collective,
                   'small': bool(args.small)},
        'roofline': roof,
        'per_rank': per_rank,
        'counters': {'n_valid_items_per_volume': sum(q.get('n_valid_items', 0) for q in per_rank),
                     'n_subbeam_gates_per_volume': n_el * n_az * n_gates * 49},
        'single_gpu_same_workload': single,
        # both speedups compare like with like: the distributed path on N ranks against the same path on one rank
        'speedup_vs_single_gpu': (one['pipelined_ms_per_volume'] / (1e3 * elapsed / args.steps)) if one else None,
        'speedup_single_volume': (one['single_volume_ms'] / (1e3 * t_block)) if one else None,
        'single_volume_blocking': {'ms_per_volume': 1e3 * t_block, 'value': gates_per_step / t_block,
                                   'note': 'submit_volume().wait() per step, nothing else in flight (max over ranks)'},
        'speedup_device_only': (value / single['value']) if single else None,
        'submit_loop_ms_per_step': 1e3 * t_submit / args.steps,
        'gather_check': gather_ok,
        'collective': runner.collective,
        'host_block_pinned': runner.host_pinned,     # the device-to-host copy behind the collective lands in memory the HIP runtime page-locked
        'collectives_in_timed_region': n_coll,
        'n_ranks_seen_by_backend': int(dist.get_world_size()),
        'api_ms': {'get_PPI_distributed_median': statistics.median(api), 'get_PPI_distributed_min': min(api),
                   'note': 'RadarOperator(distributed=True%s).get_PPI(5 elevations) on every rank, blocking: the same '
                           'sharded launch sequence and collective with all 15 arrays of the scan (76 B per gate), '
                           'RadarScan packaging on the receiving rank(s); wall time on rank 0'
                           % ('' if op.gather_to is None else ', gather_to=0')},
    }


# ------------------------------------------------------------------------------- CPU baselines
_POOL_STATE = {}


def _oracle_inputs(conf, cube, luts):
    from cosmo_pol_oracle import beam
    from cosmo_pol_oracle import config as ocfg
    from cosmo_pol_oracle import lut as olut
    oconf = ocfg.make_config(conf)
    order = ['U', 'V', 'W', 'QR_v', 'QS_v', 'QG_v', 'QI_v', 'RHO', 'T']
    oc = beam.ModelCube({n: cube['data'][n] for n in order}, cube['zlevels'], cube['proj_info'],
                        cube['resolution'], order)
    ol = {}
    for h, s in luts.items():
        L = olut.LookupTable()
        L.axes, L.axes_names, L.axes_limits, L.axes_step = s.axes, s.axes_names, s.axes_limits, s.axes_step
        L.value_table = s.value_table
        ol[h] = L
    return oconf, oc, ol


# what the one-core CPU leg keeps of its first radials: the oracle's observables of rays 0..3 (azimuth order, the leg's
# elevation, no sensitivity cut) -- `result_check` compares the GPU's sweep of that elevation with them, so that the
# oracle is touched by the cpu_baseline leg alone
_CPU_LEG_RAYS = {}


def _one_radial(inputs, a, el, keep=None):
    from cosmo_pol_oracle import beam, scatter
    oconf, oc, ol = inputs
    subs = beam.interpolate_radial(oc, oconf, float(a), float(el))
    obs = scatter.radar_observables(subs, ol, oconf)
    if keep is not None and len(keep) < 4 and (float(a), float(el)) not in keep:
        keep[(float(a), float(el))] = {k: np.array(v, copy=True) for k, v in obs.values.items()}
    return len(obs.values['ZH'])


def _pool_radial(a):
    return _one_radial(_POOL_STATE['inputs'], a, _POOL_STATE['el'])


def _sample_one_core(one, n_units, budget_s, n_samples, what):
    """`one(k) -> gates` of unit k on ONE pinned core (sched_setaffinity, what `taskset` does): one warm-up
    unit, then `n_samples` samples of budget/n_samples seconds each (at least one unit per sample) over
    consecutive units; median gates/s."""
    old = os.sched_getaffinity(0)
    os.sched_setaffinity(0, {sorted(old)[len(old) // 2]})
    try:
        one(0)
        rates, n_done, k = [], 0, 0
        for _ in range(n_samples):
            t0 = time.perf_counter()
            gates = units = 0
            while True:
                gates += one(k % n_units)
                k += 1
                units += 1
                if time.perf_counter() - t0 > budget_s / n_samples:
                    break
            dt = time.perf_counter() - t0
            rates.append((gates / dt, units / dt))
            n_done += units
    finally:
        os.sched_setaffinity(0, old)
    rates.sort()
    med = rates[len(rates) // 2]
    return {'value': med[0], 'unit': 'gates/s', 'cores': 1, 'kind': 'port',
            'sample': '%d %s in %d samples of %.1f s, median; oracle/cosmo_pol_oracle pinned to one host core '
                      '(sched_setaffinity)' % (n_done, what, n_samples, budget_s / n_samples),
            'samples_gates_per_s': [r[0] for r in rates], 'radials_per_s': med[1],
            'host_cpus': os.cpu_count()}


def cpu_baseline(conf, cube, luts, az, el, budget_s, n_samples=5):
    """the CPU oracle (restatement of the reference algorithm: per radial,
    per-variable C gate kernel, float64 LUT gather + einsum) on ONE core, radials in azimuth order."""
    inputs = _oracle_inputs(conf, cube, luts)
    n_sub = conf['integration']['nh_GH'] * conf['integration']['nv_GH']
    return _sample_one_core(lambda k: _one_radial(inputs, az[k], el, keep=_CPU_LEG_RAYS), len(az), budget_s, n_samples,
                            'radials in azimuth order (el %.1f deg, %d sub-beam%s each; output gates counted)'
                            % (el, n_sub, '' if n_sub == 1 else 's'))


def cpu_baseline_c5(conf, cube, luts_of, budget_s, small, n_samples=3):
    """The CPU oracle on Ku-band swath rays of the c5 workload (oracle/cosmo_pol_oracle/gpm.py: the intended
    behaviour of compute_trajectory_GPM, atm_refraction.py:222-272; 2-moment R,S,G,H,I): rays of the swath
    centre line in scan order, gates = gates kept below 35 km."""
    from cosmo_pol_amd import gpm
    from cosmo_pol_oracle import beam, scatter
    from cosmo_pol_oracle import config as ocfg
    from cosmo_pol_oracle import gpm as ogpm
    from cosmo_pol_oracle import lut as olut
    freq, res_m = gpm.band_settings('Ku')
    hyds = hydrometeors_of('c5')
    over = {k: dict(v) for k, v in conf.items()}
    over['radar'].update(frequency=freq, radial_resolution=res_m, sensitivity=12.0, type='GPM')
    over['radar']['3dB_beamwidth'] = 0.5
    oconf = ocfg.make_config(over)
    order = ['U', 'V', 'W', 'QR_v', 'QS_v', 'QG_v', 'QI_v', 'RHO', 'T', 'QH_v', 'QNH_v', 'QNR_v', 'QNS_v', 'QNG_v', 'QNI_v']
    oc = beam.ModelCube({n: cube['data'][n] for n in order}, cube['zlevels'], cube['proj_info'], cube['resolution'], order)
    ol = {}
    for h, s in luts_of(hyds, freq, '2mom').items():
        L = olut.LookupTable()
        L.axes, L.axes_names, L.axes_limits, L.axes_step = s.axes, s.axes_names, s.axes_limits, s.axes_step
        L.value_table = s.value_table
        ol[h] = L
    n_scans = 24 if not small else 6
    sw = gpm.synthetic_swath(n_scans=n_scans, n_rays=5, centre=(46.5, 7.5), cross_track_deg=4.0,
                             scan_spacing_m=25000.0 if not small else 1500.0)
    if small:
        for k in ('Latitude', 'Longitude'):
            c0 = 46.5 if k == 'Latitude' else 7.5
            sw[k] = c0 + (sw[k] - c0) * 0.12
    az, el, rng, sat = ogpm.swath_angles(sw)

    def one(k):
        i = k % n_scans
        subs, _, n = ogpm.interpolate_swath_ray(oc, oconf, az[i, 2], el[i, 2], rng[i, 2], sat[i])
        scatter.radar_observables(subs, ol, oconf, doppler=False)
        return n
    return _sample_one_core(one, n_scans, budget_s, n_samples,
                            'Ku rays (centre line of a %d-scan swath across the domain, 125-m gates kept below 35 km)' % n_scans)


def _pool_leg(ctx, procs, az, budget_s, chunk, first_result_timeout=45.0, **pool_kw):
    """Radials per second of a fork pool of `procs` workers, time-boxed: tasks are streamed with
    imap_unordered and the pool is terminated when the budget is used (so that a host on which
    256 NumPy workers thrash the memory system still finishes).  A leg whose first result does not
    arrive in `first_result_timeout` seconds is given up (None)."""
    import multiprocessing as mp

    def tasks():
        i = 0
        while True:
            yield az[i % len(az)]
            i += 1
    t0 = time.perf_counter()
    pool = ctx.Pool(processes=procs, **pool_kw)
    gates = n = 0
    t1 = t2 = None
    try:
        it = pool.imap_unordered(_pool_radial, tasks(), chunksize=1)     # (chunksize 1: next(timeout) exists)
        while True:
            try:
                g = it.next(timeout=first_result_timeout if t1 is None else 20.0)
            except mp.TimeoutError:
                break
            now = time.perf_counter()
            if t1 is None:
                t1 = now                          # steady state starts at the first result
                continue
            gates += g
            n += 1
            t2 = now
            if now - t1 > budget_s:
                break
    finally:
        pool.terminate()
        pool.join()
    if t1 is None or t2 is None or n == 0:
        return None
    return {'gates_per_s': gates / max(t2 - t1, 1e-9), 'radials': n, 'seconds': t2 - t1,
            'startup_s': t1 - t0, 'gates_per_s_with_startup': gates / max(t2 - t0, 1e-9)}


def cpu_pool_child(args):
    """The all-core CPU legs, run in a FRESH interpreter (no torch, no GPU state, no threads to
    inherit across fork) started by cpu_baseline_pool with a hard time limit.  SURVEY 8(d)(ii): the
    per-radial oracle under a fork pool mapped over azimuths, as radar_operator.py:402,431, with
    P = os.cpu_count() processes.  Also: fewer workers (NumPy's [n_valid, 1024, 12] float64
    temporaries make the per-radial algorithm memory-bound long before 256 cores are busy) and
    `reference_style` = Pool(P, maxtasksperchild=1), exactly as the reference creates its pool."""
    import multiprocessing as mp
    from cosmo_pol_amd import synthetic
    workload = args.workload if args.workload != 'auto' else 'c2'
    conf = bench_config(args.small, workload)
    hyds = hydrometeors_of(workload)
    cube_h = tuple(h for h in hyds if h in ('R', 'S', 'G', 'I'))
    if args.small:
        cube = synthetic.small_test_cube(hydrometeors=cube_h)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom', n_e=8)
    else:
        cube = synthetic.make_cube(hydrometeors=cube_h, **synthetic.BENCH_GRID)
        luts = synthetic.make_all_luts(hyds, 5.6, '1mom')
    az = np.arange(0, 360, 1.0)
    budget_s = 4.0
    n_cpu = max(1, os.cpu_count() or 1)
    _POOL_STATE['inputs'] = _oracle_inputs(conf, cube, luts)     # inherited by fork, not pickled
    _POOL_STATE['el'] = 1.0 if workload == 'c2' else C4_ELEVATIONS[2]
    ctx = mp.get_context('fork')
    legs = {}
    # (P = os.cpu_count() itself only up to 64 workers: with 256 NumPy workers over a 2.9 GB parent the leg has twice failed to
    # deliver within the 100-s box on the 256-core hosts of this pool -- round 4 measured 1.8e5 gates/s there when it did)
    for procs in sorted({min(n_cpu, 64), min(n_cpu, 16)}):       # small pools first
        print('[bench] CPU pool leg: %d worker processes ...' % procs, file=sys.stderr, flush=True)
        legs[procs] = _pool_leg(ctx, procs, az, budget_s, chunk=4, first_result_timeout=25.0)
        print(json.dumps({'partial': {str(k): (v or {}).get('gates_per_s') for k, v in legs.items()}}), flush=True)
    print('[bench] CPU pool leg: reference style (a fork per radial) ...', file=sys.stderr, flush=True)
    n_ref = min(n_cpu, 64)                     # (a fork per radial from 256 parents-of-2.9-GB stalls the host: bounded)
    ref = _pool_leg(ctx, n_ref, az, 3.0, chunk=1, first_result_timeout=20.0, maxtasksperchild=1)
    n_full = max(legs)
    full = legs.get(n_full)
    done = {k: v for k, v in legs.items() if v}
    best_p = max(done, key=lambda k: done[k]['gates_per_s']) if done else None
    out = {'value': full['gates_per_s'] if full else None, 'unit': 'gates/s', 'cores': n_full, 'host_cpus': n_cpu,
           'sample': ('%d radials in %.1f s of steady state, fork pool of %d persistent worker processes '
                      '(P = min(os.cpu_count(), 64)), pool start-up %.1f s excluded'
                      % (full['radials'], full['seconds'], n_full, full['startup_s'])) if full
                     else 'no result within the time box (P = %d workers)' % n_full,
           'with_pool_startup': full['gates_per_s_with_startup'] if full else None,
           'by_workers': {str(k): (v['gates_per_s'] if v else None) for k, v in sorted(legs.items())},
           'best': {'workers': best_p, 'value': done[best_p]['gates_per_s']} if done else None,
           'reference_style': {'value': ref['gates_per_s'],
                               'sample': '%d radials in %.1f s, Pool(%d, maxtasksperchild=1) as '
                                         'radar_operator.py:402,431 (a fork per radial)'
                                         % (ref['radials'], ref['seconds'], n_ref)} if ref else None}
    print(json.dumps(out), flush=True)


def cpu_baseline_pool(workload, small, limit_s=100.0):
    """Runs cpu_pool_child in a child interpreter of its own session with a hard time limit; the
    bench line never waits longer than `limit_s` for the all-core legs."""
    import signal
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-pool-child', '--workload', workload]
    if small:
        cmd.append('--small')
    print('[bench] CPU pool legs in a child interpreter (limit %.0f s) ...' % limit_s, file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=sys.stderr, text=True, start_new_session=True,
                         env=dict(os.environ, OMP_NUM_THREADS='1'))
    try:
        out, _ = p.communicate(timeout=limit_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)          # the session this function started, nothing else
        except OSError:
            pass
        out, _ = p.communicate()
        lines = [ln for ln in (out or '').splitlines() if ln.startswith('{')]
        part = json.loads(lines[-1]).get('partial') if lines else None
        return {'value': None, 'unit': 'gates/s', 'cores': os.cpu_count(),
                'error': 'all-core legs exceeded %.0f s and were stopped' % limit_s, 'by_workers': part}
    lines = [ln for ln in (out or '').splitlines() if ln.startswith('{') and '"partial"' not in ln]
    if p.returncode != 0 or not lines:
        return {'value': None, 'unit': 'gates/s', 'cores': os.cpu_count(),
                'error': 'child exited with %s' % p.returncode}
    return json.loads(lines[-1])


if __name__ == '__main__':
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    main()
