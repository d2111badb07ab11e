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


def test_lookup_roofline_prices_the_bytes_the_kernel_must_move():
    import bench

    class C(object):
        n_table_items = 195751
        n_valid_items = 195751
        ms_psd = 0.030

    r = bench.roofline_lookup('c2', C(), 195751 * bench.LUT_SLICE_BYTES, None, n_fields_read=12)
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    must = 195751 * 124
    assert abs(r['achieved'] - must / 0.030e-3 / 1e9) < 1e-6
    assert 0 < r['frac'] < 1 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert r['l1_gather']['bytes_per_item'] == 12 * 11 * 8 and 0 < r['l1_gather']['frac'] < 1
    assert r['hbm_alg_frac'] > 1                      # the survey's B_alg: what the reference algorithm reads
