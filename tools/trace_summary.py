"""Per-launch view of a rocprofv3 --kernel-trace CSV: for every kernel name, the launches grouped by grid size with their
count and mean duration (which call site of a shared kernel -- e.g. the weight-gradient reduction -- costs what).

    python tools/trace_summary.py <dir with *kernel_trace.csv> [name filter] [--skip-first N]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"].split("(")[0]
                if filt and filt not in name:
                    continue
                grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]))
                rows[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    tot = defaultdict(float)
    for (name, grid), v in rows.items():
        tot[name] += sum(v)
    for name in sorted(tot, key=lambda n: -tot[n])[:40]:
        print(f"{tot[name]:12.1f} us  {name[:90]}")
        for (n2, grid), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            if n2 == name:
                print(f"      grid {grid[0]:>7} x{grid[1]:<3} wg {grid[2]:<5} launches {len(v):>5}  mean {sum(v) / len(v):9.1f} us  total {sum(v):10.1f} us")


if __name__ == "__main__":
    main()
