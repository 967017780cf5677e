# one-rank RCCL leg: the default three-stage schedule and no split (AB_DDP_SPLIT=0), each with and without the next batch's render on the side
# stream (--no-render-overlap), against the plain step; alternating on one box.  (split2_norender needs the two-stage patch of DESIGN 13.6.)
for i in 1 2 3; do
  for v in base split3 split3_norender unsplit unsplit_norender; do
    cmd="python bench.py --rccl-single-rank"; env="X=1"
    case $v in
      base) cmd="python bench.py --no-cpu-baseline --sustain 0 --no-eval-leg --no-dexycb-leg --no-study-leg --no-jpeg-leg --no-mixed-leg --no-rccl-leg --no-dropin-leg";;
      split3_norender) cmd="$cmd --no-render-overlap";;
      unsplit) env="AB_DDP_SPLIT=0";;
      unsplit_norender) cmd="$cmd --no-render-overlap"; env="AB_DDP_SPLIT=0";;
    esac
    env $env timeout 300 $cmd 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', d['ms_per_step'], 'ms/step', d['config'].get('parallelism'), 'render_overlap', d['config'].get('render_overlap'))"
  done
done
