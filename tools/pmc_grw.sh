# HBM traffic and SQ counters of gemm_rw_kernel on the standalone probe (tools/probe_grw_0 built with GRW_ABL=0): three separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
SAM=${1:-0}
rm -rf /tmp/g1 /tmp/g2 /tmp/g3 /tmp/g4
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/g1 -o p -- $R/tools/probe_grw_0 $SAM 3 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/g2 -o p -- $R/tools/probe_grw_0 $SAM 3 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/g3 -o p -- $R/tools/probe_grw_0 $SAM 3 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/g4 -o p -- $R/tools/probe_grw_0 $SAM 3 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in ('/tmp/g1', '/tmp/g2', '/tmp/g3', '/tmp/g4'):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs: print('no counters in', d); continue
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if 'gemm_rw' not in r['Kernel_Name']: continue
        agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for c in sorted(agg):
        v = agg[c] / n[c]
        extra = ''
        if c == 'FETCH_SIZE': extra = f'  -> {v * 1024 * 2 / 1e6:.1f} MB per launch (x2: gfx950 wide-load correction; counter unit KB)'
        if c == 'WRITE_SIZE': extra = f'  -> {v * 1024 / 1e6:.1f} MB per launch (counter unit KB, uncalibrated)'
        print(f'{c:28s} {v:14.4g} per launch (n={n[c]}){extra}')
PY
