#!/bin/bash
# ONE rocprofv3 --pmc pass (plus kernel trace) of a python script, summarised per kernel:
#   tools/pmc_pass.sh <name> "<COUNTER ...>" [kernel-substring] -- <script.py> [args...]
#   -> gpurun_out/pmc_<name>.json   (mean per dispatch of every counter, per kernel)
# The program itself follows `--` (no env / bash hop); counters that do not fit one pass fail the pass: split them.
set -u
name=$1; counters=$2; shift 2
want=""
if [ "$1" != "--" ]; then want=$1; shift; fi
shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
script=$root/$1; shift
[ -f "$script" ] || { echo "pmc_pass: no such script: $script" >&2; exit 2; }
for kv in ${PC_ENV:-}; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$name
echo "[pmc_pass] $name: $counters"
# shellcheck disable=SC2086
rocprofv3 --pmc $counters --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 "$script" "$@" > "$root/gpurun_out/pmc_$name.stdout.log" 2> "$root/gpurun_out/pmc_$name.log" \
    || { echo "pmc_pass: pass failed (status $?): gpurun_out/pmc_$name.log" >&2; tail -3 "$root/gpurun_out/pmc_$name.log" >&2; exit 3; }
python3 "$root/tools/pmc_summary.py" /tmp/pmc_$name $want > "$root/gpurun_out/pmc_$name.json" || exit 3
echo "done $name"
