# per-step time of the BatchNorm kernel family with and without the in-apply finalize (AB_BNFIN_FUSE), from rocprofv3 kernel traces
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
rm -rf /tmp/bnab$v
AB_BNFIN_FUSE=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bnab$v -o s -- python /root/repo/bench.py --no-cpu-baseline --sustain 0 --no-eval-leg --no-dexycb-leg --no-study-leg --no-jpeg-leg --no-mixed-leg --no-rccl-leg --no-dropin-leg --steps 20 > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob('/tmp/bnab$v/**/*kernel_stats.csv', recursive=True)[0]
tot = 0; rows = []
for r in csv.DictReader(open(f)):
    n = r['Name']
    if n.startswith('void bn_') or n.startswith('bn_') or 'bn_fin' in n or 'bn_apply' in n or 'bn_bwd' in n or 'bn_finalize' in n:
        rows.append((float(r['TotalDurationNs']) / 25 / 1e3, int(r['Calls']) / 25, float(r['AverageNs']) / 1e3, n[:70]))
        tot += float(r['TotalDurationNs']) / 25 / 1e3
print("AB_BNFIN_FUSE=$v: BatchNorm family %.1f us per step (25 steps incl. warm-up)" % tot)
for t, c, a, n in sorted(rows, reverse=True): print("   %8.1f us  %5.1f calls  avg %6.1f  %s" % (t, c, a, n))
PY
done
