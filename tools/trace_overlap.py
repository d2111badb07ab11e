#!/usr/bin/env python
"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last N sweeps, the timeline of
kernel starts / ends per stream (lane) relative to the first kernel, plus busy fractions.
usage: trace_overlap.py <dir> [n_kernels]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rows = []
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:28],
                             r.get('Stream_Id', r.get('Queue_Id', '?'))))
    rows.sort()
    # take a window in the middle of the timed region: sweeps on the device-output path
    mid = int(len(rows) * float(sys.argv[3])) if len(sys.argv) > 3 else len(rows) * 2 // 5
    win = rows[mid:mid + n]
    t0 = win[0][0]
    for s, e, name, q in win:
        print('%9.1f %9.1f  %6.1f us  q=%s  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
    # union busy time and PSD busy time over the window
    span = (max(e for _, e, _, _ in win) - t0) / 1e3
    psd = [(s, e) for s, e, nme, _ in win if 'k_psd' in nme]
    def union(iv):
        iv = sorted(iv); tot = 0; cur_s, cur_e = iv[0]
        for s, e in iv[1:]:
            if s > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s, e
            else: cur_e = max(cur_e, e)
        return (tot + cur_e - cur_s) / 1e3
    print('window %.1f us; any-kernel busy %.1f us; psd busy (union) %.1f us; psd sum %.1f us; n_psd %d'
          % (span, union([(s, e) for s, e, _, _ in win]), union(psd), sum(e - s for s, e in psd) / 1e3, len(psd)))


if __name__ == '__main__':
    main()
