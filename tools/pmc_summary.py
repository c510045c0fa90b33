"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into the per-kernel HBM traffic table kept under
profiles/ and the traffic json bench.py reads for `roofline.traffic`.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirF> -- python3 bench.py --steps 3 --warmup 1 \
        --no-cpu-baseline --no-train-step --streams 1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dirW> -- python3 bench.py ... (same)
    python tools/pmc_summary.py <dirF> <dirW> profiles/r01_final_pmc_summary.md profiles/r01_final_traffic.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are reported in
KiB; FETCH_SIZE tallies 128-B read requests at 64 B on gfx950, so wide coalesced reads are doubled.
"""
import csv
import glob
import json
import os
import re
import sys


def per_kernel_max(d, counter):
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                out[name] = max(out.get(name, 0.0), float(row["Counter_Value"]))
    return out


def main():
    dir_f, dir_w, md_path, json_path = sys.argv[1:5]
    fetch, write = per_kernel_max(dir_f, "FETCH_SIZE"), per_kernel_max(dir_w, "WRITE_SIZE")
    names = sorted(set(fetch) | set(write), key=lambda n: -(2 * fetch.get(n, 0) + write.get(n, 0)))
    lines = ["# HBM traffic per launch (rocprofv3 PMC, BASELINE config 2, single stream)", "",
             "    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
             "--no-cpu-baseline --no-train-step --streams 1",
             "    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py (same flags)", "",
             "Largest launch of each kernel, KiB as rocprofv3 reports them; double FETCH on gfx950 for wide coalesced reads "
             "(MI355X_MICROARCH.md, HBM section).", "", "| kernel | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB |", "|---|---|---|"]
    for n in names:
        if n.startswith("tsde::"):
            lines.append(f"| {n} | {fetch.get(n, 0):.0f} | {write.get(n, 0):.0f} |")
    with open(md_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    edge = [n for n in names if "k_edge_kv2" in n or "k_edge_kv<" in n]
    best = max(edge, key=lambda n: write.get(n, 0)) if edge else None
    if best:
        fr, wr = fetch.get(best, 0.0), write.get(best, 0.0)
        doc = {"k_edge_kv[aa]": {
            "kernel": best, "FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB": wr, "hbm_bytes_per_launch": int((2 * fr + wr) * 1024),
            "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes: python3 bench.py --steps 3 --warmup 1 "
                    "--no-cpu-baseline --no-train-step --streams 1 (config 2, 3.43 M (t,edge) pairs per launch); FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); algorithmic bytes: 20 B in + 288 B out per "
                    "pair = 1.06 GB + 44 MB of unique q rows"}}
        with open(json_path, "w") as f:
            json.dump(doc, f, indent=1)
    print("\n".join(lines[:30]))


if __name__ == "__main__":
    main()
