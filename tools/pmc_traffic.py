"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md
"HBM" section): FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B; confirmed here on sam_bwd, whose
logits read is 90112 KB and reports 45276 KB), WRITE_SIZE is taken as is (sam_bwd writes 90112 KB, reports 90112.0 KB).
Counter unit: KB.

    python tools/pmc_traffic.py <fetch.csv[.gz]> <write.csv[.gz]> <steps-in-run> <out-prefix>
"""
import collections, csv, gzip, io, json, sys

CONV = ("conv_gemm_kernel", "conv_gemm2_kernel", "conv3x3_kernel", "wgrad_kernel", "wgrad3x3_kernel", "wgrad_reduce")


def load(path):
    f = io.TextIOWrapper(gzip.open(path)) if path.endswith(".gz") else open(path)
    d = collections.OrderedDict()
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0]
        a = d.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return d


def main():
    fetch, write, steps, out = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
    rows = []
    for k in fetch:
        n = fetch[k][0]
        rd = 2.0 * fetch[k][1] * 1024 / steps
        wr = write.get(k, [0, 0.0])[1] * 1024 / steps
        rows.append((k, n / steps, rd, wr))
    rows.sort(key=lambda r: -(r[2] + r[3]))
    with open(out + ".csv", "w") as f:
        f.write("kernel,launches_per_step,hbm_read_MB_per_step,hbm_write_MB_per_step\n")
        for k, n, rd, wr in rows:
            f.write(f"\"{k}\",{n:.2f},{rd/1e6:.2f},{wr/1e6:.2f}\n")
    conv = [r for r in rows if r[0].replace("void ", "").split("<")[0] in CONV]
    summ = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --eager --steps 3 --warmup 1`",
            "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1; KB -> bytes",
            "steps": steps,
            "conv_stack_bytes_per_step": sum(r[2] + r[3] for r in conv),
            "conv_stack_launches_per_step": sum(r[1] for r in conv),
            "all_kernels_bytes_per_step": sum(r[2] + r[3] for r in rows)}
    json.dump(summ, open(out + ".json", "w"), indent=1)
    print(json.dumps(summ))


if __name__ == "__main__":
    main()
