#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of one python command, on the GPU box:
#   tools/kernel_times.sh <pattern> <script.py> [args...]      e.g.  "lookup|classify" tools/stage_times.py --config c4
root=${GRAFT_REPO_ROOT:-$(pwd)}
pat=$1; shift
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python3 $script "$@" > /dev/null 2>&1
f=$(find /tmp/kt1 -name "*kernel_stats.csv" | head -1)
grep -E -i "$pat" $f | python3 -c "
import csv, sys
for r in csv.reader(sys.stdin):
    print('%-60s calls=%s avg_us=%.1f' % (r[0][:58], r[1], float(r[3]) / 1000.0))"
