#!/bin/bash
# The bench command itself under rocprofv3 (run on the GPU box):
#   tools/take_profiles.sh <tag> [workload]    e.g. r4bench c2   ->  gpurun_out/profiles_<tag>_<workload>/
# A thin wrapper over tools/profile_cmd.sh (kernel-trace stats, then separate FETCH_SIZE / WRITE_SIZE / SQ-counter
# passes; exits with status 3 before anything reads a file a pass did not leave).  The isolated launch sequences
# bench.py takes its roofline bytes from are profiled with tools/profile_cmd.sh <name> tools/stage_times.py ...
set -u
tag=${1:-run}
wl=${2:-c2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$wl" = "c2" ]; then args="--steps 20 --warmup 2 --repeats 1"; else args="--steps 2 --warmup 1"; fi
# shellcheck disable=SC2086
CPOL_BENCH_NO_EXTRAS=1 exec "$root/tools/profile_cmd.sh" "${tag}_${wl}" bench.py --workload "$wl" $args --cpu-seconds 0
