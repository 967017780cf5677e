#!/bin/bash
# The three profile artefacts of a build, as committed under profiles/ (run on the GPU box from the repo root):
#   tools/profile_round.sh <tag>      ->  gpurun_out/<tag>_bench_kernel_stats.csv, <tag>_step_trace.txt, <tag>_pmc_hbm_traffic.{csv,json},
#                                         <tag>_eval_kernel_stats.csv, <tag>_eval_step_trace.txt
# 1. rocprofv3 --kernel-trace --stats on the default bench command (graph replays): per-kernel totals + one step in launch order
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE; each with --kernel-trace only) on the eager step, reduced by tools/pmc_traffic.py
set -e
tag=${1:-roundX}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/stats -o s -- python $root/bench.py --no-cpu-baseline --sustain 0 --no-eval-leg --no-dexycb-leg --no-study-leg --no-jpeg-leg --no-mixed-leg --no-rccl-leg --no-dropin-leg --steps 20 > $out/${tag}_bench_line.txt 2>&1
cp $(find /tmp/prof_$tag/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
python $root/tools/step_trace.py $(find /tmp/prof_$tag/stats -name "*kernel_trace.csv" | head -1) > $out/${tag}_step_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$tag/$c -o p -- python $root/bench.py --eager --no-cpu-baseline --sustain 0 --no-eval-leg --no-dexycb-leg --no-study-leg --no-jpeg-leg --no-mixed-leg --no-rccl-leg --no-dropin-leg --steps 3 --warmup 1 > /dev/null 2>&1
done
python $root/tools/pmc_traffic.py $(find /tmp/prof_$tag/FETCH_SIZE -name "*counter_collection.csv" | head -1) \
    $(find /tmp/prof_$tag/WRITE_SIZE -name "*counter_collection.csv" | head -1) $out/${tag}_pmc_hbm_traffic bf16x3
# 3. BASELINE configs[1] (eval-mode forward, bs 64): the same kernel-trace summary + one forward in launch order
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/eval -o e -- python $root/bench.py --eval --steps 20 > $out/${tag}_eval_line.txt 2>&1
cp $(find /tmp/prof_$tag/eval -name "*kernel_stats.csv" | head -1) $out/${tag}_eval_kernel_stats.csv
python $root/tools/step_trace.py $(find /tmp/prof_$tag/eval -name "*kernel_trace.csv" | head -1) sam_stage2 -8 > $out/${tag}_eval_step_trace.txt
tail -1 $out/${tag}_bench_line.txt | cut -c1-200
tail -1 $out/${tag}_eval_line.txt | cut -c1-200
# 4. the real-frame JPEG decode: kernel-trace summary of tools/bench_jpeg.py + one ab_jpeg_decode_batch call in launch order
cd $root
bash tools/trace_jpeg.sh > $out/${tag}_jpeg_call_trace.txt 2>&1
cp $(find /tmp/pj3 -name "*kernel_trace.csv" | head -1) /tmp/pj3_trace.csv
python - <<PY > $out/${tag}_jpeg_kernel_stats.csv
import csv, collections
rows = list(csv.DictReader(open('/tmp/pj3_trace.csv')))
agg = collections.defaultdict(list)
for r in rows:
    if 'jpeg' in r['Kernel_Name']:
        agg[r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"')
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'"{k}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{min(v)},{max(v)}')
PY
timeout 200 python tools/bench_jpeg.py --sub-bytes 128 2>&1 | tail -2 >> $out/${tag}_jpeg_call_trace.txt
tail -3 $out/${tag}_jpeg_call_trace.txt
