"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md
"HBM" section), next to the ALGORITHMIC bytes of each kernel family (every tensor the family must touch, once), so that
the wasted-traffic ratio is in the file.

Corrections: FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B; confirmed on sam_bwd, whose logits read is
90112 KB at bf16 and reports 45276 KB), WRITE_SIZE is taken as is (sam_bwd writes 90112 KB, reports 90112.0 KB).  Counter
unit: KB.  The passes come from `bench.py --eager --steps 3 --warmup 1` (counters are per dispatch; graph replays are not
attributed): kernels are normalised per step by the launch count of a marker kernel of their phase (learner: pose_loss_kernel,
render: raster_shade_kernel, optimizer: clip_adam_kernel).

    python tools/pmc_traffic.py <fetch.csv[.gz]> <write.csv[.gz]> <out-prefix> [bf16x3|bf16]
"""
import collections
import csv
import gzip
import io
import json
import os
import sys

CONV = ("conv_gemm_kernel", "conv_gemm2_kernel", "conv3x3_kernel", "conv3x3r_kernel", "conv2x2_kernel", "convp_kernel", "gemm_rw_kernel",
        "stem_halo_kernel", "stem_halo_x3_kernel", "wgrad_kernel", "wgrad3x3_kernel", "wgrad_gemm2_kernel", "wgrad_reduce", "wgrad_reduce_batch_kernel")
BN = ("bn_apply_kernel", "bn_apply_x3_kernel", "bn_bwd_reduce_kernel", "bn_bwd_apply_kernel", "bn_bwd_apply_x3_kernel", "bn_finalize_kernel",
      "bn_bwd_finalize_kernel", "bn_fin_apply_x3_kernel", "bn_fin_bwd_apply_x3_kernel", "maxpool_fwd_x3_kernel", "pool_win_bn_reduce_kernel",
      "pool_bwd_bn_reduce_kernel", "pool_bwd_bn_apply_x3_kernel", "col_stats_kernel", "col_stats_x3_kernel", "maxpool_fwd_kernel", "maxpool_bwd_kernel", "split_f32_kernel", "avgpool_fwd_kernel",
      "avgpool_bwd_kernel")
HEAD = ("sam_stage1", "sam_stage2", "sam_bwd", "sam_bias_finalize", "pose_loss_kernel", "pose_loss_finalize", "colsum_finalize_kernel", "linear_nt_kernel",
        "linear_wgrad_kernel")
RENDER = ("raster_setup_kernel", "raster_shade_kernel", "gauss_blur_kernel", "jitter_stats_kernel", "warp_jitter_kernel", "zero_words_kernel",
          "mano_lbs_kernel")
OPTIM = ("sqnorm_kernel", "norm_finalize_kernel", "clip_adam_kernel", "transpose_oki_batch_kernel", "cast_f32_bf16_kernel")
FAMILIES = (("conv stack (fwd, dgrad, wgrad, slab reduce)", CONV), ("BatchNorm / ReLU / pooling / operand split", BN),
            ("soft-argmax head, pose+loss, box MLP", HEAD), ("render chain", RENDER), ("clip + Adam + weight repack", OPTIM))


def base(name):
    return name.replace("void ", "").split("(")[0].split("<")[0].strip()


def load(path):
    f = io.TextIOWrapper(gzip.open(path)) if path.endswith(".gz") else open(path)
    d = collections.OrderedDict()
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0]
        a = d.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return d


def algorithmic(dtype, B=64, S=256):
    """Compulsory HBM bytes per step of each family: every tensor touched once per pass that needs it (B = 64, 256 x 256,
    ResNet-34 + IntegralDeconvHead 22 x 32 (padded depth) x 32 x 32).  e = bytes per activation element."""
    e = 4 if dtype != "bf16" else 2
    convs, bns = [], []           # (in_elems, out_elems, w_elems, has_dgrad, addend_elems) ; (elems, residual, keeps_f32)
    H = S // 2
    convs.append((B * (S + 6) * (S + 8) * 4, B * H * H * 64, 64 * 7 * 8 * 4, False, 0))
    bns.append((B * H * H * 64, False))
    H //= 2
    cin = 64
    for li, (c, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for b in range(n):
            s = 2 if (b == 0 and li > 0) else 1
            Hi, Ho = H, H // s
            i, o = B * Hi * Hi * cin, B * Ho * Ho * c
            convs.append((i, o, c * 9 * cin, True, i))            # conv1 (its data gradient merges the residual gradient)
            bns.append((o, False))
            convs.append((o, o, c * 9 * c, True, 0))              # conv2
            bns.append((o, True))
            if s == 2 or cin != c:
                convs.append((i, o, c * cin, True, i))
                bns.append((o, False))
            cin, H = c, Ho
    f = B * H * H * 512
    convs.append((f, 4 * f // 2, 512 * 16 * 256, True, 0)); bns.append((4 * f // 2, False))      # ConvT 512 -> 256
    convs.append((2 * f, 8 * f, 256 * 16 * 256, True, 0)); bns.append((8 * f, False))           # ConvT 256 -> 256
    logits = B * (S // 8) ** 2 * 22 * 32
    convs.append((8 * f, logits, 704 * 256, True, 0))
    conv = 0
    for i, o, w, dg, add in convs:
        conv += (i + w) * e + o * e                               # forward
        if dg:
            conv += (o + w) * e + i * e + add * e                 # data gradient (+ residual-branch addend)
        conv += (i + o) * e + w * 4                               # weight gradient (fp32 out)
    bn = 0
    for t, res in bns:
        bn += t * e * (2 + (2 if res else 0))                     # apply: y (+res) in, activation out (+fp32 copy kept for the residual)
        bn += t * e * (2 + (1 if res else 0)) + t * e * (1 + (1 if res else 0))   # backward: dout, y (+out) in; dy (+dz) out
    head = 3 * logits * e + 2 * B * 512 * 4                        # logits read fwd, read + written bwd
    render = 129e6                                                # 2.0 MB per sample
    P = 25.56e6                                                    # flat parameter elements (incl. padding)
    optim = P * 4 * (4 + 3) + (P * 4 if dtype != "f32" else 0)     # p, g, m, v read; p, m, v written (+ compute-precision copy)
    return dict(zip((f[0] for f in FAMILIES), (conv, bn, head, render, optim)))


def main():
    if sys.argv[1] == "--from-csv":
        # re-aggregate the per-kernel table of an earlier reduction (the family lists above grow with the kernels: a table written before a kernel
        # was listed counted it in no family):  pmc_traffic.py --from-csv profiles/<tag>_pmc_hbm_traffic.csv <out-prefix> [dtype]
        src, out = sys.argv[2], sys.argv[3]
        dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
        rows = []
        for line in open(src):
            parts = next(csv.reader([line])) if line.startswith('"') else []
            if len(parts) == 4:
                try:
                    rows.append((parts[0], float(parts[1]), float(parts[2]) * 1e6, float(parts[3]) * 1e6))
                except ValueError:
                    pass
        marker = json.load(open(src[:-4] + ".json")).get("steps_in_run") if os.path.exists(src[:-4] + ".json") else None
    else:
        fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
        dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
        cnt = {base(k): v[0] for k, v in fetch.items()}
        marker = {"learn": cnt.get("pose_loss_kernel", 1), "render": cnt.get("raster_shade_kernel", 1), "optim": cnt.get("clip_adam_kernel", 1)}
        rows = []
        for k in fetch:
            b = base(k)
            steps = marker["render"] if b in RENDER else marker["optim"] if b in OPTIM else marker["learn"]
            rd = 2.0 * fetch[k][1] * 1024 / steps
            wr = write.get(k, [0, 0.0])[1] * 1024 / steps
            rows.append((k, fetch[k][0] / steps, rd, wr))
    rows.sort(key=lambda r: -(r[2] + r[3]))
    alg = algorithmic(dtype)
    fam_rows = []
    for fname, members in FAMILIES:
        sel = [r for r in rows if base(r[0]) in members]
        meas = sum(r[2] + r[3] for r in sel)
        fam_rows.append({"family": fname, "launches_per_step": round(sum(r[1] for r in sel), 1), "hbm_bytes_per_step": round(meas),
                         "algorithmic_bytes_per_step": round(alg[fname]), "traffic_over_algorithmic": round(meas / alg[fname], 2)})
    with open(out + ".csv", "w") as f:
        f.write("kernel,launches_per_step,hbm_read_MB_per_step,hbm_write_MB_per_step\n")
        for k, n, rd, wr in rows:
            f.write(f"\"{k}\",{n:.2f},{rd/1e6:.2f},{wr/1e6:.2f}\n")
        f.write("\nfamily,launches_per_step,hbm_MB_per_step,algorithmic_MB_per_step,traffic_over_algorithmic\n")
        for r in fam_rows:
            f.write(f"\"{r['family']}\",{r['launches_per_step']},{r['hbm_bytes_per_step']/1e6:.1f},{r['algorithmic_bytes_per_step']/1e6:.1f},{r['traffic_over_algorithmic']}\n")
    summ = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --eager --steps 3 --warmup 1`",
            "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1; KB -> bytes", "dtype": dtype,
            "steps_in_run": marker,
            "conv_stack_bytes_per_step": fam_rows[0]["hbm_bytes_per_step"],
            "conv_stack_algorithmic_bytes_per_step": fam_rows[0]["algorithmic_bytes_per_step"],
            "conv_stack_launches_per_step": fam_rows[0]["launches_per_step"],
            "families": fam_rows,
            "all_kernels_bytes_per_step": round(sum(r[2] + r[3] for r in rows if base(r[0]) in sum((m for _, m in FAMILIES), ())))}
    json.dump(summ, open(out + ".json", "w"), indent=1)
    print(json.dumps(summ, indent=1))


if __name__ == "__main__":
    main()
