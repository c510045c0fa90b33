"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: every launch in start order with its duration and the idle
gap in front of it; totals of busy / idle time, and the launches grouped by name with the gaps they sit behind.

    python tools/step_timeline.py <dir with *kernel_trace.csv> [--anchor k_rotate] [--full]

A step is cut from the last-but-one launch of the anchor kernel (the first kernel of a step) to the last one."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "k_rotate"
    full = "--full" in sys.argv
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tsde::", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(marks) < 2:
        raise SystemExit(f"fewer than two launches of {anchor}")
    step = rows[marks[-2]:marks[-1]]
    t0 = step[0][0]
    busy = idle = 0.0
    prev_end = t0
    by = defaultdict(lambda: [0, 0.0, 0.0])
    for s, e, n in step:
        gap = max(0, s - prev_end) * 1e-3
        dur = (e - s) * 1e-3
        overlap = s < prev_end
        busy += (e - max(s, prev_end)) * 1e-3 if e > prev_end else 0.0
        idle += gap
        b = by[n[:70]]
        b[0] += 1
        b[1] += dur
        b[2] += gap
        if full:
            print(f"{(s - t0) * 1e-3:10.1f} us  +{gap:6.1f} gap  {dur:8.1f} us {'|' if overlap else ' '} {n[:100]}")
        prev_end = max(prev_end, e)
    span = (prev_end - t0) * 1e-3
    print(f"step: {len(step)} launches, span {span:.1f} us, busy {busy:.1f} us, idle {idle:.1f} us")
    print(f"{'launches':>8} {'kernel us':>10} {'gaps us':>9}  name")
    for n, (c, du, g) in sorted(by.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:60]:
        print(f"{c:8d} {du:10.1f} {g:9.1f}  {n}")


if __name__ == "__main__":
    main()
