#!/usr/bin/env python
"""Per-panel verdicts of the integral-table accuracy gate for a golden case with roughened tables:
   python tools/gate_detail.py <case> <kind>      (tests/_rough.py kinds, incl. 'mie')"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import _cases  # noqa: E402
import _rough  # noqa: E402
from cosmo_pol_amd import RadarOperator  # noqa: E402

name, kind = sys.argv[1], sys.argv[2]
conf, az, el, ocube, luts, cube = _cases.radial_case(name)
over = _cases.gen_golden.radial_case_inputs(name)[0]
rl = luts if kind == 'smooth' else _rough.roughen_all(luts, kind, frequency=conf['radar']['frequency'],
                                                      scheme=conf['microphysics']['scheme'])
op = RadarOperator(config=over, luts=rl, output_variables='only_radar')
rep = op._ctx.itab_report()
np.set_printoptions(precision=1, linewidth=200)
for j, h in enumerate(op._staged_hydro):
    d = op._ctx.itab_detail(j)
    print(h, 'check %.2e edge %.2e n_bad %d' % (rep['check'][j], rep['check_edge'][j], rep['n_bad'][j]),
          'accepted', None if d is None else d['accepted_panels'], 'of', None if d is None else d['n_pan'])
    if d is not None:
        print('   by_fn ', d['by_fn'])
        print('   by_pan first 16', d['by_pan'][:16])
        print('   by_pan last 8  ', d['by_pan'][-8:])
op.close()
