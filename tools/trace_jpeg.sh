# one ab_jpeg_decode_batch call in launch order (rocprofv3 --kernel-trace of tools/bench_jpeg.py): start offset, duration, kernel
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pj3
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pj3 -o t -- python /root/repo/tools/bench_jpeg.py --iters 3 "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pj3/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'jpeg_tables' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
    print(f"{(st - t0) / 1e3:9.1f} us {(en - st) / 1e3:8.1f} us  {name}  grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
PY
