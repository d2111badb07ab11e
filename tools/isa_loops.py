#!/usr/bin/env python
"""Instruction mix of the loops of a kernel in cosmo_pol_amd/csrc/cosmo_pol_hip.gfx950.s
(`make -C cosmo_pol_amd/csrc asm`):  python tools/isa_loops.py <mangled-name-substring> ..."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    s = open(os.path.join(ROOT, 'cosmo_pol_amd', 'csrc', 'cosmo_pol_hip.gfx950.s')).read()
    names = re.findall(r'^(_Z\w+):', s, flags=re.M)
    for want in sys.argv[1:]:
        for name in [n for n in names if want in n]:
            i = s.index('\n' + name + ':')
            f = s[i:s.index('s_endpgm', i)]
            lines = f.split('\n')
            labels = {}
            for idx, l in enumerate(lines):
                m = re.match(r'^(\.LBB\d+_\d+):', l)
                if m:
                    labels[m.group(1)] = idx
            print(name, 'lines', len(lines))
            for idx, l in enumerate(lines):
                m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
                if m and m.group(1) in labels and labels[m.group(1)] < idx:
                    body = lines[labels[m.group(1)]:idx + 1]
                    cnt = lambda pat: sum(1 for x in body if re.match(r'\s+' + pat, x))   # noqa: E731
                    print('  loop %5d-%5d: %4d instrs | f64 VALU %4d (fma %d mul %d add %d other %d) | other VALU %3d | '
                          's_load %2d s_waitcnt %2d ds %2d vmem %2d' % (
                              labels[m.group(1)], idx, sum(1 for x in body if re.match(r'\s+[sv]_|\s+ds_|\s+global_|\s+scratch_', x)),
                              cnt(r'v_\w+_f64'), cnt('v_fma_f64'), cnt('v_mul_f64'), cnt('v_add_f64'),
                              cnt(r'v_\w+_f64') - cnt('v_fma_f64') - cnt('v_mul_f64') - cnt('v_add_f64'),
                              cnt(r'v_(?!\w+_f64)'), cnt('s_load'), cnt('s_waitcnt'), cnt('ds_'),
                              cnt('(global|scratch|buffer)_')))


if __name__ == '__main__':
    main()
