#!/usr/bin/env python
"""Accuracy gate of the integral tables (cpol_prepare): per staged slot the worst deviation of a
block's polynomial from the integrating kernel at the block's check points (1-D tables: mid-panel,
u = 0.37, and near the panel edge, u = 0.96; `check` = both, `edge` = the second alone, over the
accepted run of panels), where, and the time of the build / of the check.
usage: itab_report.py [case ...] [--full] [--mie] [--detail]   (cases of tests/_cases.py; --full: the
bench's full-size R,S,G,mS,mG,I tables; --mie: the cases with closed-form Mie tables, cosmo_pol_amd/mie.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def show(tag, op):
    rep = op._ctx.itab_report()
    for j, h in enumerate(op._staged_hydro):
        at = int(rep['at'][j])
        print('%-14s %-3s check %+.3e  edge %.3e  at (block %d, fn %d) mod 2^24   n_bad %d   build %.2f ms  check %.2f ms'
              % (tag, h, rep['check'][j], rep['check_edge'][j], at // 15, at % 15, rep['n_bad'][j], rep['build_ms'][j], rep['check_ms'][j]))
        d = op._ctx.itab_detail(j)
        if d is not None and '--detail' not in sys.argv:
            print('    accepted panels %s of %d' % (d['accepted_panels'], d['n_pan']))
        if d is not None and '--detail' in sys.argv:
            print('    n_pan %d  log2_lo %g  ppo %d  d0 %g  accepted panels %s' % (d['n_pan'], d['log2_lo'], d['ppo'], d['d0'], d['accepted_panels']))
            print('    by function:', ' '.join('%.1e' % x for x in d['by_fn']))
            bp, be = d['by_pan'], d['by_pan_edge']
            for p0 in range(0, len(bp), 16):
                print('    panels %3d.. (log2 lambda %6.2f):' % (p0, d['log2_lo'] + p0 / d['ppo']),
                      ' '.join('%.0e' % x for x in bp[p0:p0 + 16]))
                print('        edge point alone              :', ' '.join('%.0e' % x for x in be[p0:p0 + 16]))


def main():
    from cosmo_pol_amd import RadarOperator
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    if '--full' in sys.argv:
        import bench
        from cosmo_pol_amd import synthetic
        hy = bench.hydrometeors_of('c4')
        op = RadarOperator(config=bench.bench_config(False, 'c3'), luts=synthetic.make_all_luts(hy, 5.6, '1mom'),
                           output_variables='only_radar')
        show('full c3', op)
        op.close()
    import _cases
    for name in args or ['c3_melt_ice', 'c2_rsg', 'c5_2mom', 'c5_ka_2mom']:
        over = _cases.gen_golden.radial_case_inputs(name)[0]
        conf, _, _, _, luts, _ = _cases.radial_case(name)
        if '--mie' in sys.argv:
            import _rough
            luts = _rough.roughen_all(luts, 'mie', frequency=conf['radar']['frequency'], scheme=conf['microphysics']['scheme'])
            name = name + '/mie'
        op = RadarOperator(config=over, luts=luts, output_variables='only_radar')
        show(name, op)
        op.close()


if __name__ == '__main__':
    main()
