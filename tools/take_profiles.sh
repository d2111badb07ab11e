#!/bin/bash
# Collects the rocprofv3 summaries committed under profiles/ (run on the GPU box):
#   tools/take_profiles.sh <tag> [workload]    e.g. r2 c2   /   r2 c4
# kernel-trace stats of the bench command, then separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# SQ counters) as MI355X_MICROARCH.md prescribes (never --pmc together with a trace domain
# other than --kernel-trace), merged by tools/profile_summary.py into the summary bench.py reads.
set -e
tag=${1:-run}
wl=${2:-c2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_${tag}_${wl}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/pmcF /tmp/pmcW /tmp/pmcS
if [ "$wl" = "c2" ]; then long="--steps 200 --warmup 10 --repeats 1"; short="--steps 5 --warmup 1 --repeats 1"; else long="--steps 4 --warmup 1"; short="--steps 1 --warmup 1"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $root/bench.py --workload $wl $long --cpu-seconds 0 > $out/bench_under_rocprof.json 2> $out/kt.log
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $out/${tag}_${wl}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcF -- python3 $root/bench.py --workload $wl $short --cpu-seconds 0 > /dev/null 2> $out/pmcF.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcW -- python3 $root/bench.py --workload $wl $short --cpu-seconds 0 > /dev/null 2> $out/pmcW.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcS -- python3 $root/bench.py --workload $wl $short --cpu-seconds 0 > /dev/null 2> $out/pmcS.log
python3 $root/tools/profile_summary.py $out/${tag}_${wl}_kernel_stats.csv /tmp/pmcF /tmp/pmcW /tmp/pmcS > $out/${tag}_${wl}_summary.json
echo done
