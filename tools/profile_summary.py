#!/usr/bin/env python
"""Merges the rocprofv3 passes of tools/take_profiles.sh into ONE summary that bench.py reads
(profiles/r2_<workload>_summary.json): per kernel the mean SQ_INSTS_VALU / SQ_WAVE_CYCLES /
SQ_WAIT_ANY / FETCH_SIZE / WRITE_SIZE per dispatch, the HBM bytes per dispatch with the gfx950
correction of MI355X_MICROARCH.md ((2 FETCH_SIZE + WRITE_SIZE) KiB) and the kernel-trace
average duration.

usage: profile_summary.py <kernel_stats.csv> <pmc_fetch_dir> <pmc_write_dir> <pmc_sq_dir> > summary.json"""
import csv
import glob
import json
import sys
from collections import defaultdict


def pmc(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row.get('Kernel_Name', '')][row['Counter_Name']].append(float(row['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


SCALAR_LOAD_KERNELS = ('k_subbeam_sum<true', 'k_subbeam_sum_scalar')       # coefficient rows through s_load (round 4's default form,
                                                                           # k_subbeam_sum_lds, reads through the vector path: the common rule)


def main():
    stats, dF, dW, dS = sys.argv[1:5]
    extra = sys.argv[5:]                                   # optional passes: memory-side request split, L2 hit rate
    out = {'_note': 'means per dispatch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE '
                    'counts 64 B per 128-B request of the vector caches; kernels that read through the scalar cache: '
                    'FETCH_SIZE + WRITE_SIZE, hbm_bytes_upper = the vector rule); avg_us from rocprofv3 --kernel-trace --stats'}
    with open(stats) as fh:
        for row in csv.DictReader(fh):
            name = row.get('Name') or row.get('KernelName') or ''
            out[name] = {'calls': int(float(row.get('Calls', 0))),
                         'avg_us': float(row.get('AverageNs', row.get('Average', 0))) / 1e3}
    for d, keep in ((dF, ('FETCH_SIZE',)), (dW, ('WRITE_SIZE',)),
                    (dS, ('SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
                          'SQ_ACTIVE_INST_ANY', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'))):
        for k, cs in pmc(d).items():
            e = out.setdefault(k, {})
            for c in keep:
                if c in cs:
                    e[c] = cs[c]
    for d in extra:
        for k, cs in pmc(d).items():
            e = out.setdefault(k, {})
            for c in ('TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_DRAM_sum', 'TCC_EA0_RDREQ_32B_sum', 'TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_REQ_sum',
                      'SQ_LDS_IDX_ACTIVE', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_ACTIVE_INST_LDS', 'SQ_INSTS_SALU'):
                if c in cs:
                    e[c] = cs[c]
    for k, e in out.items():
        if isinstance(e, dict) and e.get('TCC_EA0_RDREQ_sum'):
            # of the L2's memory-side read requests, those routed to the DRAM side (the rest: other agents' memory);
            # Infinity-Cache hits are not told apart here -- the MALL sits behind this interface
            e['rdreq_dram_share'] = e.get('TCC_EA0_RDREQ_DRAM_sum', 0.0) / e['TCC_EA0_RDREQ_sum']
        if isinstance(e, dict) and e.get('TCC_REQ_sum'):
            e['l2_hit_rate'] = e.get('TCC_HIT_sum', 0.0) / e['TCC_REQ_sum']
        if isinstance(e, dict) and e.get('SQ_LDS_IDX_ACTIVE') and e.get('GRBM_GUI_ACTIVE'):
            # cycles the LDS index stage of a CU is active (summed over 256 CUs) over the kernel's cycles (GRBM_GUI_ACTIVE: summed over 8 XCDs)
            e['lds_busy_frac'] = (e['SQ_LDS_IDX_ACTIVE'] / 256.0) / (e['GRBM_GUI_ACTIVE'] / 8.0)
    for k, e in out.items():
        if isinstance(e, dict) and 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
            e['hbm_bytes'] = (2 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024
            if any(tag in k for tag in SCALAR_LOAD_KERNELS):
                # reads through the scalar cache are counted in full (profiles/r3_fetch_calibration.txt:
                # k_sread96 1.000 of the 64-B sectors), only the kernel's few vector reads at one half:
                # FETCH_SIZE + WRITE_SIZE is the lower bound, 2 FETCH_SIZE + WRITE_SIZE the upper one
                e['hbm_bytes_upper'] = e['hbm_bytes']
                e['hbm_bytes'] = (e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
