for s in 3 0 3; do timeout 300 python bench.py --no-cpu-baseline --no-eval-leg --no-dexycb-leg --no-study-leg --no-jpeg-leg --sustain $s 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('sustain $s:', d['value'], d['roofline']['conv_ms_per_step'], d['roofline']['frac'])"; done
