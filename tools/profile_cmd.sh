#!/bin/bash
# rocprofv3 passes of an arbitrary python command (run on the GPU box), merged into one summary:
#   [PC_ENV="A=1 B=2"] tools/profile_cmd.sh <name> <script.py> [args...]
#   -> gpurun_out/profiles_<name>/<name>_kernel_stats.csv and <name>_summary.json
# kernel-trace + stats, then separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters), as
# MI355X_MICROARCH.md prescribes; the program itself follows `--` (no env / bash hop).
# Every ad-hoc profile goes through this script.  Rules it keeps (a run of round 3 was killed for
# silence after `head` with an empty operand sat on stdin): no output goes to /dev/null, every pass
# appends to a log under gpurun_out/ and prints a progress line, a pass that leaves no result file
# ends the script with status 3 before anything reads it, and no reader is ever called without a file.
set -u
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$name
mkdir -p "$out"
script=$root/$1; shift
[ -f "$script" ] || { echo "profile_cmd: no such script: $script" >&2; exit 2; }
for kv in ${PC_ENV:-}; do export "$kv"; done     # (exported here: never an `env` hop behind `--`)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/pmcF /tmp/pmcW /tmp/pmcS

need() {    # need <dir> <pattern> <what>: the newest matching file, or exit 3
    local f
    f=$(find "$1" -name "$2" -type f 2>/dev/null | sort | tail -n 1)
    if [ -z "$f" ] || [ ! -s "$f" ]; then
        echo "profile_cmd: $3 left no $2 under $1 (see $out/*.log)" >&2
        exit 3
    fi
    printf '%s' "$f"
}

echo "[profile_cmd] $name: kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 "$script" "$@" > "$out/stdout_kt.log" 2> "$out/kt.log" \
    || { echo "profile_cmd: the kernel-trace pass failed (status $?): $out/kt.log" >&2; exit 3; }
stats=$(need /tmp/kt "*kernel_stats.csv" "the kernel-trace pass") || exit 3
cp "$stats" "$out/${name}_kernel_stats.csv"
for pass in F:FETCH_SIZE W:WRITE_SIZE "S:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    tag=${pass%%:*}; counters=${pass#*:}
    echo "[profile_cmd] $name: pmc pass $tag ($counters)"
    # shellcheck disable=SC2086
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d /tmp/pmc$tag -- python3 "$script" "$@" >> "$out/stdout_pmc$tag.log" 2>> "$out/pmc$tag.log" \
        || { echo "profile_cmd: pmc pass $tag failed (status $?): $out/pmc$tag.log" >&2; exit 3; }
    need /tmp/pmc$tag "*counter_collection.csv" "pmc pass $tag" > /dev/null || exit 3
done
# optional passes (a counter this ROCm does not list only costs the pass): where the L2's memory-side reads go (DRAM-side
# against all: Infinity-Cache hits are not told apart at this interface), the L2 hit rate, and the LDS (cycles its index
# stage is active per CU: the bound of the kernels that broadcast table rows from it)
for pass in "D:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "L:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "X:SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
    tag=${pass%%:*}; counters=${pass#*:}
    rm -rf /tmp/pmc$tag
    echo "[profile_cmd] $name: optional pmc pass $tag ($counters)"
    # shellcheck disable=SC2086
    rocprofv3 --pmc $counters --kernel-trace --output-format csv -d /tmp/pmc$tag -- python3 "$script" "$@" >> "$out/stdout_pmc$tag.log" 2>> "$out/pmc$tag.log" \
        || echo "[profile_cmd] optional pass $tag failed (status $?): $out/pmc$tag.log"
done
python3 "$root/tools/profile_summary.py" "$out/${name}_kernel_stats.csv" /tmp/pmcF /tmp/pmcW /tmp/pmcS /tmp/pmcD /tmp/pmcL /tmp/pmcX > "$out/${name}_summary.json" \
    || { echo "profile_cmd: profile_summary.py failed" >&2; exit 3; }
[ -s "$out/${name}_summary.json" ] || { echo "profile_cmd: empty summary" >&2; exit 3; }
grep -h "^{" "$out/stdout_kt.log" | cut -c1-300 || true
echo "done $name"
