"""Per-launch kernel durations of ONE replayed step from a rocprofv3 --kernel-trace CSV (launch order kept).
usage: python tools/step_trace.py <kernel_trace.csv> [step_marker_kernel=clip_adam_kernel] [which_step=auto]
which_step = auto: the step with the SHORTEST span among those that carry no wall_stamp_kernel launch (bench.py's roofline leg replays a
stamped copy of the step and runs a few eager steps after the timed loop: neither is the step the headline is quoted on)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "clip_adam_kernel"
which = sys.argv[3] if len(sys.argv) > 3 else "auto"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(marker)]
if which == "auto":
    best = None
    for k in range(1, len(ends)):
        a, b = ends[k - 1] + 1, ends[k] + 1
        if any("wall_stamp_kernel" in r["Kernel_Name"] for r in rows[a:b]):
            continue
        span = int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])
        if best is None or span < best[0]:
            best = (span, a, b)
    _, lo, hi = best
else:
    which = int(which)
    lo, hi = ends[which - 1] + 1, ends[which] + 1
t0 = int(rows[lo]["Start_Timestamp"])
tot = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    tot += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {name[:70]:70s} grid={r.get('Grid_Size_X', '?')}/{r.get('Workgroup_Size_X', '?')} lds={r.get('LDS_Block_Size', '?')}")
print(f"kernel time {tot / 1e6:.3f} ms, span {(int(rows[hi - 1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
