# A/B of one environment switch on the default bench step, alternating processes on ONE box:  tools/ab_env.sh AB_C3_STACK 0 1 [pairs]
var=$1; a=$2; b=$3; pairs=${4:-2}
for i in $(seq 1 $pairs); do
  for v in $a $b; do
    env $var=$v timeout 300 python bench.py --no-cpu-baseline --sustain 0 --no-eval-leg --no-dexycb-leg --no-study-leg --no-jpeg-leg --no-mixed-leg --no-rccl-leg --no-dropin-leg 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$var=$v', d['value'], 'samples/s', d['ms_per_step'], 'ms/step; conv stack', d['roofline']['conv_ms_per_step'], 'ms')"
  done
done
