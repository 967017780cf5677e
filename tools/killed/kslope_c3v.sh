cd /tmp && export TMPDIR=/tmp
for v in 0 1; do for C in 128 256 512; do
rm -rf /tmp/ks; AB_C3V=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o p -- python /root/repo/tools/one_c3v.py 16 $C 20 256 > /dev/null 2>&1
python - <<PY
import csv, glob
fs = glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(fs[0])):
    n = r['Name']
    if 'conv3x3' in n: print("C3V=$v Cin=$C", n[:50], f"avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}")
PY
done; done
