# SQ counters of the render chain's kernels (two passes of eight counters over tools/one_render.py), per-launch means
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr1 /tmp/pr2 /tmp/pr0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr0 -o s -- python /root/repo/tools/one_render.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pr1 -o p -- python /root/repo/tools/one_render.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/pr2 -o p -- python /root/repo/tools/one_render.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/pr0/**/*kernel_stats.csv', recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:10]:
        print(f"{r['Name'].split('(')[0][:50]:52s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ('/tmp/pr1', '/tmp/pr2'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs: print('no counters in', d); continue
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:40]
        if not any(t in k for t in ('raster', 'jitter', 'gauss', 'warp')): continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, v in sorted(agg.items()):
    n = max(cnt[(k, c)] for c in v)
    w = v.get('SQ_WAVE_CYCLES', 0) or 1
    print(f"{k:42s} n={n:3d} " + ' '.join(f"{c[3:]}={x / n:.3g}" for c, x in sorted(v.items())))
    print(f"{'':42s}   active/wave={v.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f} valu/wave={v.get('SQ_ACTIVE_INST_VALU', 0) / w:.3f} wait_any/wave={v.get('SQ_WAIT_ANY', 0) / w:.3f} wait_inst/wave={v.get('SQ_WAIT_INST_ANY', 0) / w:.3f} wait_lds/wave={v.get('SQ_WAIT_INST_LDS', 0) / w:.3f} lds_conflict/lds_active={v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
PY
