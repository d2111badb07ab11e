#!/usr/bin/env python
"""tools/make_luts.py -- write / inspect scattering lookup tables in the on-disk
format of the reference (tar of value_table / axes / axes_names / axes_step /
axes_limits .npy members, cosmo_pol/lookup/lut.py:78-154), under the directory
layout `load_all_lut` expects:

    <lut_dir>/lut_<scattering>/lut_SZ_<H>_<freq with '_' for '.'>_<scheme>.lut

  python tools/make_luts.py write  --lut-dir DIR [--frequency 5.6] [--scheme 1mom]
                                   [--hydrometeors R S G ...] [--scattering tmatrix_masc]
                                   [--n-e N] [--n-t N] [--seed S]
  python tools/make_luts.py info   FILE.lut [...]

`write` produces the SYNTHETIC tables of cosmo_pol_amd/synthetic.py (Rayleigh-
spheroid model in the exact reference layout) or, with --model mie, closed-form Mie
tables (cosmo_pol_amd/mie.py: resonant in D; pytmatrix is not available, so
the table VALUES are not the reference's -- real cosmo_pol .lut files drop into
the same directory and are read by the same loader).  `RadarOperator(...,
lut_dir=DIR)` then stages them exactly as it would stage real tables.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from cosmo_pol_amd import lut as lutmod  # noqa: E402
from cosmo_pol_amd import synthetic  # noqa: E402

FOLDERS = {'tmatrix': 'lut_tmatrix', 'tmatrix_masc': 'lut_tmatrix_masc', 'dda': 'lut_dda'}
DEFAULT_HYDROMETEORS = {'1mom': ['R', 'S', 'G', 'I', 'mS', 'mG'],
                        '2mom': ['R', 'S', 'G', 'H', 'I', 'mS', 'mG']}


def write(lut_dir, frequency, scheme, hydrometeors, scattering, n_e=None, n_t=None,
          seed=20260301, quiet=False, model='spheroid'):
    folder = os.path.join(lut_dir, FOLDERS[scattering])
    os.makedirs(folder, exist_ok=True)
    written = []
    for h in hydrometeors:
        table = synthetic.make_lut(h, frequency, scheme, seed, n_e, n_t)
        if model == 'mie' and h not in ('mS', 'mG'):
            # closed-form Mie series of the equal-volume sphere (resonances along D), the two polarisations
            # split by the Rayleigh spheroid ratio (cosmo_pol_amd/mie.py); the melting species keep the spheroid model
            from cosmo_pol_amd import mie
            table = mie.mie_table_like(table, h, frequency, scheme)
        path = os.path.join(folder, lutmod.lut_filename(h, frequency, scheme))
        lutmod.save_lut(table, path)
        written.append(path)
        if not quiet:
            print('%s  %s  %.1f MB' % (path, 'x'.join(map(str, table.value_table.shape)),
                                       os.path.getsize(path) / 1e6))
    return written


def info(path):
    t = lutmod.load_lut(path)
    print(path)
    print('  value_table %s %s' % (t.value_table.shape, t.value_table.dtype))
    for name, i in sorted(t.axes_names.items(), key=lambda kv: kv[1]):
        ax = np.asarray(t.axes[i])
        print('  axis %d %-3s shape %-12s first %.6g last %.6g step %s'
              % (i, name, ax.shape, ax.ravel()[0], ax.ravel()[-1], np.ravel(t.axes_step[i])[0]))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    sub = ap.add_subparsers(dest='cmd', required=True)
    w = sub.add_parser('write')
    w.add_argument('--lut-dir', required=True)
    w.add_argument('--frequency', type=float, default=5.6)
    w.add_argument('--scheme', choices=['1mom', '2mom'], default='1mom')
    w.add_argument('--hydrometeors', nargs='+', default=None)
    w.add_argument('--scattering', choices=sorted(FOLDERS), default='tmatrix_masc')
    w.add_argument('--n-e', type=int, default=None, help='truncate the elevation axis (small tables)')
    w.add_argument('--n-t', type=int, default=None, help='truncate the temperature / wet-fraction axis')
    w.add_argument('--seed', type=int, default=20260301)
    w.add_argument('--model', choices=['spheroid', 'mie'], default='spheroid',
                   help="spheroid: smooth Rayleigh-spheroid tables (default); mie: Lorenz-Mie series of the "
                        "equal-volume sphere for the non-melting species (resonant in D at Ku / Ka band)")
    i = sub.add_parser('info')
    i.add_argument('files', nargs='+')
    a = ap.parse_args(argv)
    if a.cmd == 'write':
        write(a.lut_dir, a.frequency, a.scheme, a.hydrometeors or DEFAULT_HYDROMETEORS[a.scheme],
              a.scattering, a.n_e, a.n_t, a.seed, model=a.model)
    else:
        for f in a.files:
            info(f)


if __name__ == '__main__':
    main()
