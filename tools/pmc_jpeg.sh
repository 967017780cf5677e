cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pj2 -o p -- python /root/repo/tools/bench_jpeg.py --iters 2 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pj2/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
rows = list(csv.DictReader(open(f)))
for r in rows:
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k, v in agg.items():
    if 'jpeg' not in k: continue
    n = cnt[(k, 'SQ_WAVES')]
    print(k, 'launches', n, {c: round(x / n) for c, x in v.items()})
# per-dispatch for the sync kernel: first 4 dispatches
seen = 0
byd = collections.defaultdict(dict)
for r in rows:
    if 'jpeg_sync' in r['Kernel_Name']:
        byd[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
for d in sorted(byd, key=int)[:3]:
    print(d, byd[d])
PY
