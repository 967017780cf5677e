#!/bin/bash
# BatchNorm / pooling family: per-launch bytes, microseconds and TB/s of one step.   tools/bn_table.sh <out-file>   (run on the GPU box from the repo root)
root=$(pwd); out=${1:-$root/gpurun_out/bn_table.txt}; case $out in /*) ;; *) out=$root/$out;; esac
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bnt
B="python $root/bench.py --no-cpu-baseline --sustain 0 --no-eval-leg --no-dexycb-leg --no-study-leg --no-jpeg-leg --no-mixed-leg --no-rccl-leg --no-dropin-leg"
rocprofv3 --kernel-trace --output-format csv -d /tmp/bnt/t -o t -- $B --steps 20 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/bnt/$c -o p -- $B --eager --steps 3 --warmup 1 > /dev/null 2>&1
done
mkdir -p $(dirname $out); python $root/tools/bn_table.py $(find /tmp/bnt/FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/bnt/WRITE_SIZE -name "*counter_collection.csv" | head -1) \
    $(find /tmp/bnt/t -name "*kernel_trace.csv" | head -1) "${@:2}" > $out 2>&1
cat $out
