# kernel durations + SQ counters of the 3x3 kernels of tools/one_c3v.py: tools/pmc_c3v.sh H C   (env AB_C3V etc. pass through)
H=${1:-16}; C=${2:-256}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv0 /tmp/pv1 /tmp/pv2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv0 -o p -- python /root/repo/tools/one_c3v.py $H $C 20 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pv1 -o p -- python /root/repo/tools/one_c3v.py $H $C 4 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pv2 -o p -- python /root/repo/tools/one_c3v.py $H $C 4 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, re
fs = glob.glob('/tmp/pv0/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(fs[0])):
    n = r['Name']
    if 'conv3x3' in n or 'c3v' in n:
        print(f"{n[:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ('/tmp/pv1', '/tmp/pv2'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs: print('no counters in', d); continue
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].split('(')[0][:64]
        if 'conv3x3' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, v in sorted(agg.items()):
    n = max(cnt[(k, c)] for c in v)
    w = v.get('SQ_WAVE_CYCLES', 0) or 1
    print(k, 'n =', n)
    print('   ' + ' '.join(f"{c[3:]}={x / n:.4g}" for c, x in sorted(v.items())))
    print(f"   active/wave={v.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f} wait_any/wave={v.get('SQ_WAIT_ANY', 0) / w:.3f} wait_inst/wave={v.get('SQ_WAIT_INST_ANY', 0) / w:.3f} wait_lds/wave={v.get('SQ_WAIT_INST_LDS', 0) / w:.3f}  mfma_busy/busy_cycles={v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(v.get('SQ_BUSY_CYCLES', 1), 1):.3f}  lds_conflict/lds_active={v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
PY
