#!/bin/bash
# Per-kernel average durations (rocprofv3 --kernel-trace --stats) of one python command, on the GPU box:
#   [KT_ENV="A=1 B=2"] tools/kernel_times.sh <pattern> <script.py> [args...]
#   e.g.  tools/kernel_times.sh "lookup|classify" tools/stage_times.py --config c4
# KT_ENV: variables exported before the profiler starts (never an `env` hop behind `--`).  Output of the
# profiled program goes to gpurun_out/kernel_times.log; a pass without a stats file ends with status 3
# before anything reads it (no reader is ever started without a file operand).
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
pat=$1; shift
script=$root/$1; shift
[ -f "$script" ] || { echo "kernel_times: no such script: $script" >&2; exit 2; }
mkdir -p "$root/gpurun_out"
for kv in ${KT_ENV:-}; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -- python3 "$script" "$@" >> "$root/gpurun_out/kernel_times.log" 2>&1 \
    || { echo "kernel_times: the profiled command failed (status $?): gpurun_out/kernel_times.log" >&2; exit 3; }
f=$(find /tmp/kt1 -name "*kernel_stats.csv" -type f | sort | tail -n 1)
[ -n "$f" ] && [ -s "$f" ] || { echo "kernel_times: no kernel_stats.csv under /tmp/kt1" >&2; exit 3; }
echo "== ${KT_ENV:-default} :: $*"
python3 - "$f" "$pat" <<'PY'
import csv, re, sys
pat = re.compile(sys.argv[2], re.I)
for r in csv.reader(open(sys.argv[1])):
    if r and pat.search(r[0]):
        try:
            print('%-60s calls=%s avg_us=%.1f' % (r[0][:58], r[1], float(r[3]) / 1000.0))
        except (ValueError, IndexError):
            pass
PY
