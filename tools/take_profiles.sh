#!/bin/bash
# Collects the rocprofv3 summaries committed under profiles/ (run on the GPU box):
#   tools/take_profiles.sh <tag>     e.g. r1_final2
# kernel-trace stats of the default bench command, then separate PMC passes (FETCH_SIZE,
# WRITE_SIZE, SQ counters) as MI355X_MICROARCH.md prescribes.
set -e
tag=${1:-run}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/pmcF /tmp/pmcW /tmp/pmcS
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $root/bench.py --steps 200 --warmup 10 --cpu-seconds 0 > $out/bench_under_rocprof.json 2> $out/kt.log
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcF -- python3 $root/bench.py --steps 5 --warmup 1 --cpu-seconds 0 > /dev/null 2> $out/pmcF.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcW -- python3 $root/bench.py --steps 5 --warmup 1 --cpu-seconds 0 > /dev/null 2> $out/pmcW.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcS -- python3 $root/bench.py --steps 5 --warmup 1 --cpu-seconds 0 > /dev/null 2> $out/pmcS.log
python3 $root/tools/pmc_summary.py /tmp/pmcF > $out/${tag}_pmc_fetch.json
python3 $root/tools/pmc_summary.py /tmp/pmcW > $out/${tag}_pmc_write.json
python3 $root/tools/pmc_summary.py /tmp/pmcS k_psd > $out/${tag}_pmc_sq_psd.json
echo done
