#!/usr/bin/env python
"""Summarises a rocprofv3 --pmc CSV directory: per kernel name, mean of every counter
over its dispatches.  usage: pmc_summary.py <dir> [kernel-substring] > summary.json"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', '')
                if want and want not in name:
                    continue
                acc[name[:80]][row['Counter_Name']].append(float(row['Counter_Value']))
    out = {k: {c: {'mean': sum(v) / len(v), 'n': len(v)} for c, v in cs.items()} for k, cs in acc.items()}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
