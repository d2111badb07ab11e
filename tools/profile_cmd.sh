#!/bin/bash
# rocprofv3 passes of an arbitrary python command (run on the GPU box), merged into one summary:
#   tools/profile_cmd.sh <name> <script.py> [args...]
#   -> gpurun_out/profiles_<name>/<name>_kernel_stats.csv and <name>_summary.json
# kernel-trace + stats, then separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ counters), as
# MI355X_MICROARCH.md prescribes; the program itself follows `--` (no env / bash hop).
set -e
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$name
mkdir -p $out
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/pmcF /tmp/pmcW /tmp/pmcS
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $script "$@" > $out/stdout_kt.log 2> $out/kt.log
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $out/${name}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcF -- python3 $script "$@" > /dev/null 2> $out/pmcF.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcW -- python3 $script "$@" > /dev/null 2> $out/pmcW.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcS -- python3 $script "$@" > /dev/null 2> $out/pmcS.log
python3 $root/tools/profile_summary.py $out/${name}_kernel_stats.csv /tmp/pmcF /tmp/pmcW /tmp/pmcS > $out/${name}_summary.json
grep -h "^{" $out/stdout_kt.log | cut -c1-300
echo "done $name"
