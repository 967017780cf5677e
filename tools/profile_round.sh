#!/bin/bash
# The three profile artefacts of a build, as committed under profiles/ (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag>      ->  gpurun_out/<tag>_bench_kernel_stats.csv, <tag>_step_trace.txt, <tag>_pmc_hbm_traffic.{csv,json}
# 1. rocprofv3 --kernel-trace --stats on the default bench command (graph replays): per-kernel totals + one step in launch order
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE; each with --kernel-trace only) on the eager step, reduced by tools/pmc_traffic.py
set -e
tag=${1:-roundX}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/stats -o s -- python $root/bench.py --no-cpu-baseline --steps 20 > $out/${tag}_bench_line.txt 2>&1
cp $(find /tmp/prof_$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
python $root/tools/step_trace.py $(find /tmp/prof_$tag/stats -name "*kernel_trace.csv" | head -1) > $out/${tag}_step_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$tag/$c -o p -- python $root/bench.py --eager --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
done
python $root/tools/pmc_traffic.py $(find /tmp/prof_$tag/FETCH_SIZE -name "*counter_collection.csv" | head -1) \
    $(find /tmp/prof_$tag/WRITE_SIZE -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_hbm_traffic bf16x3
tail -1 $out/${tag}_bench_line.txt | cut -c1-200
