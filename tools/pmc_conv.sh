# SQ counters of the MFMA kernels of one eager step (two passes of 8 SQ counters): where the wave cycles of conv3x3 / wgrad3x3 go
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc1 /tmp/pc2
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pc1 -o p -- python /root/repo/bench.py --eager --no-cpu-baseline --sustain 0 --no-eval-leg --no-dexycb-leg --no-study-leg --no-jpeg-leg --no-mixed-leg --no-rccl-leg --no-dropin-leg --steps 2 --warmup 1 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pc2 -o p -- python /root/repo/bench.py --eager --no-cpu-baseline --sustain 0 --no-eval-leg --no-dexycb-leg --no-study-leg --no-jpeg-leg --no-mixed-leg --no-rccl-leg --no-dropin-leg --steps 2 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, re, os
FILTER = os.environ.get('PMC_FILTER', 'conv3x3,wgrad3x3,conv_gemm2,wgrad_gemm2').split(',')
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ('/tmp/pc1', '/tmp/pc2'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs: print('no counters in', d); continue
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0][:64]
        if not any(t in k for t in FILTER): continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, v in sorted(agg.items()):
    n = max(cnt[(k, c)] for c in v)
    w = v.get('SQ_WAVE_CYCLES', 0) or 1
    print(f"{k:66s} n={n:3d} " + ' '.join(f"{c[3:]}={x / n:.3g}" for c, x in sorted(v.items())))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
        print(f"{'':66s}   mfma_busy/(4*wave_cycles/waves-per-simd..)  busy/BUSY_CYCLES={v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(v.get('SQ_BUSY_CYCLES', 1), 1):.3f}  active/wave={v.get('SQ_ACTIVE_INST_ANY', 0) / w:.3f} wait_any/wave={v.get('SQ_WAIT_ANY', 0) / w:.3f} wait_inst/wave={v.get('SQ_WAIT_INST_ANY', 0) / w:.3f} wait_lds/wave={v.get('SQ_WAIT_INST_LDS', 0) / w:.3f}")
    if 'SQ_LDS_BANK_CONFLICT' in v:
        print(f"{'':66s}   lds_conflict/lds_active={v['SQ_LDS_BANK_CONFLICT'] / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f}")
PY
