cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
cat > /tmp/run_mixed.py <<'PY'
import sys, os, yaml
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_mixed
cfg = yaml.safe_load(open('/root/repo/config/ho3dv2_clasbased_artiboost_mi355x.yaml'))
cfg["DATA_PRESET"]["IMAGE_SIZE"] = [256, 256]
bench_mixed.train_loop(cfg, steps=16, modes=("same stream, frames of 4 batches decoded per call one group ahead on a side stream",))
PY
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pm -o t -- python /tmp/run_mixed.py > /tmp/pm_out.txt 2>&1
tail -1 /tmp/pm_out.txt
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pm/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'clip_adam' in r['Kernel_Name']]
a, b = idx[-6], idx[-2]          # four steps = one decode group
t0, t1 = int(rows[a]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
print('span of 4 steps: %.1f us -> %.1f us per step' % ((t1 - t0) / 1e3, (t1 - t0) / 4e3))
busy = collections.defaultdict(float); fam = collections.defaultdict(float)
for r in rows[a:b]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy[r.get('Queue_Id', '?')] += d
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('<')[0].split('(')[0][:40]
    fam[(r.get('Queue_Id', '?'), n)] += d
for q, t in busy.items(): print('queue', q, '%.1f us per step' % (t / 4))
mq = max(busy, key=busy.get)
for (q, n), t in sorted(fam.items(), key=lambda kv: -kv[1])[:50]:
    if 'conv' in n or 'wgrad' in n or n.startswith('bn_'): continue
    print('  q%s %-40s %.1f us per step' % (q, n, t / 4))
# idle on the main queue
prev = None; idle = 0
for r in rows[a:b]:
    if r.get('Queue_Id', '?') != mq: continue
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev is not None and st > prev: idle += st - prev
    prev = max(prev or 0, en)
print('idle on the main queue: %.1f us per step' % (idle / 4e3))
PY
