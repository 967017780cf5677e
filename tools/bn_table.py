"""Per-launch bandwidth table of the BatchNorm / pooling kernel family (round-5 review item 6): for every launch of one training step --
HBM bytes (rocprofv3 PMC passes FETCH_SIZE x 2 [gfx950 correction, tools/pmc_traffic.py] + WRITE_SIZE, eager step), duration in the REPLAYED
step (rocprofv3 --kernel-trace of the default bench command), TB/s -- grouped by (kernel, grid size) = by layer shape.

    python tools/bn_table.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel_trace.csv> [name-substring ...]
Launches are matched between the eager PMC runs and the replayed trace by (kernel name, k-th occurrence within the step)."""
import collections
import csv
import sys

FAMILY = ("bn_", "maxpool", "pool_", "col_stats", "split_f32", "avgpool")


def short(name):
    return name.replace("void ", "").split("(")[0].strip()


def steps_of(rows, name_key, order_key, marker="clip_adam_kernel"):
    rows = sorted(rows, key=order_key)
    ends = [i for i, r in enumerate(rows) if short(r[name_key]).startswith(marker)]
    return rows, ends


def one_step_counters(path):
    rows = list(csv.DictReader(open(path)))
    key = (lambda r: int(r["Dispatch_Id"])) if "Dispatch_Id" in rows[0] else (lambda r: 0)
    rows, ends = steps_of(rows, "Kernel_Name", key)
    lo, hi = ends[-2] + 1, ends[-1] + 1                     # the last complete step of the run
    occ, out = collections.Counter(), {}
    for r in rows[lo:hi]:
        n = short(r["Kernel_Name"])
        out[(n, occ[n])] = (float(r["Counter_Value"]), r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        occ[n] += 1
    return out


def one_step_trace(path):
    rows = list(csv.DictReader(open(path)))
    rows, ends = steps_of(rows, "Kernel_Name", lambda r: int(r["Start_Timestamp"]))
    best = None
    for k in range(1, len(ends)):                            # the shortest replayed step without stamp launches (tools/step_trace.py)
        a, b = ends[k - 1] + 1, ends[k] + 1
        if any("wall_stamp_kernel" in r["Kernel_Name"] for r in rows[a:b]):
            continue
        span = int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])
        if best is None or span < best[0]:
            best = (span, a, b)
    occ, out = collections.Counter(), {}
    for r in rows[best[1]:best[2]]:
        n = short(r["Kernel_Name"])
        out[(n, occ[n])] = ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")))
        occ[n] += 1
    return out


def main():
    fetch, write, trace = one_step_counters(sys.argv[1]), one_step_counters(sys.argv[2]), one_step_trace(sys.argv[3])
    want = sys.argv[4:] or FAMILY
    groups = collections.OrderedDict()
    for (n, k), (us, grid) in trace.items():
        if not any(w in n for w in want):
            continue
        f, w = fetch.get((n, k)), write.get((n, k))
        if f is None or w is None:
            continue
        g = groups.setdefault((n, grid), [0, 0.0, 0.0, 0.0, []])
        g[0] += 1; g[1] += us; g[2] += 2.0 * f[0] * 1024; g[3] += w[0] * 1024; g[4].append(us)
    tot_us = tot_b = 0.0
    print(f"{'kernel':44s} {'grid':>9s} {'n':>3s} {'us/launch':>9s} {'MB read':>8s} {'MB write':>8s} {'TB/s':>6s} {'us/step':>8s}")
    for (n, grid), (c, us, rd, wr, lst) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        tot_us += us; tot_b += rd + wr
        print(f"{n[:44]:44s} {grid:>9s} {c:3d} {us / c:9.1f} {rd / c / 1e6:8.1f} {wr / c / 1e6:8.1f} {(rd + wr) / us / 1e6:6.2f} {us:8.1f}")
    print(f"family: {tot_us:.1f} us per step, {tot_b / 1e9:.2f} GB per step, {tot_b / tot_us / 1e6:.2f} TB/s")


if __name__ == "__main__":
    main()
