# one step of `bench.py --rccl-single-rank` in launch order per stream (rocprofv3 --kernel-trace): where the multi-rank schedule spends its extra time
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pr1 -o t -- python /root/repo/bench.py --rccl-single-rank --sustain 0 --steps 12 --warmup 3 "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pr1/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'clip_adam' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
# the step between two raster_setup launches; render overlaps, so take a window and report per-queue
t0 = int(rows[a]['Start_Timestamp'])
busy = collections.defaultdict(float); last = {}
names = collections.Counter()
for r in rows[a:b]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    q = r.get('Queue_Id', '?')
    busy[q] += (en - st) / 1e3
    n = r['Kernel_Name'].split('(')[0][:50]
    if q != '4' or 'ccl' in n.lower(): names[(q, n)] += (en - st) / 1e3
span = (int(rows[b]['Start_Timestamp']) - t0) / 1e3
print('span between two clip_adam launches: %.1f us' % span)
for q, t in busy.items(): print('queue', q, 'kernel time %.1f us' % t)
for (q, n), t in names.items(): print('  queue', q, n, '%.1f us' % t)
# gaps > 15 us on the busiest queue
mq = max(busy, key=busy.get)
prev = None
for r in rows[a:b]:
    if r.get('Queue_Id', '?') != mq: continue
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev is not None and st - prev > 8000:
        print('  gap %.1f us before %s at %.1f us' % ((st - prev) / 1e3, r['Kernel_Name'].split('(')[0][:40], (st - t0) / 1e3))
    prev = max(prev or 0, en)
PY
