"""Summarise the rocprofv3 --pmc passes collected by tools/collect_counters.sh into one markdown table per program (largest
launch of each kernel) and the traffic json bench.py quotes under `roofline.traffic_profiled`.

    python tools/counter_summary.py <dir with pmc_fwd_*/ and pmc_sde_*/> out.md traffic.json

Units follow MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES counts cycles; FETCH_SIZE / WRITE_SIZE are KiB, and FETCH_SIZE tallies 128-B read requests at 64 B
on gfx950, so wide coalesced reads are doubled before they are compared with a byte count.
"""
import csv
import glob
import json
import os
import re
import sys


def collect(root, prefix):
    """{kernel: {counter: value of the launch with the largest SQ_WAVE_CYCLES-ish footprint}}: per counter the max over launches"""
    out = {}
    for d in sorted(glob.glob(os.path.join(root, prefix + "*"))):
        if not os.path.isdir(d):
            continue
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").strip()
                    if not name.startswith("tsde::"):
                        continue
                    name = name[len("tsde::"):]
                    c, v = row["Counter_Name"], float(row["Counter_Value"])
                    k = out.setdefault(name, {})
                    k[c] = max(k.get(c, 0.0), v)
    return out


def table(title, cmd, data, order_key="SQ_WAVE_CYCLES", top=14):
    cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
            "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM",
            "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS",
            "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"]
    names = sorted(data, key=lambda n: -data[n].get(order_key, 0.0))[:top]
    lines = [f"## {title}", "", f"    {cmd}", "", "Largest launch of each kernel (per counter, over the launches of the run).", "",
             "| kernel | " + " | ".join(c.replace("SQ_", "") for c in cols) + " |", "|---|" + "---|" * len(cols)]
    for n in names:
        lines.append(f"| {n} | " + " | ".join(f"{data[n][c]:.4g}" if c in data[n] else "" for c in cols) + " |")
    lines += ["", "Derived (fractions of the wave cycles: where a resident wave's time goes):", "",
              "| kernel | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | VALU / wave-cycles | MFMA pipe busy / (SIMD-cycles of the launch) | VALU+MFMA co-executing / MFMA busy | VALU per MFMA | HBM bytes (2 x FETCH + WRITE) |",
              "|---|---|---|---|---|---|---|---|---|"]
    for n in names:
        d = data[n]
        wc = d.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        busy = d.get("SQ_BUSY_CYCLES", 0.0)                  # per-SE busy cycles summed; GRBM_GUI_ACTIVE / 8 = kernel cycles
        gui = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        simd_cycles = gui * 1024 if gui > 0 else 0.0
        mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        hbm = (2 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024
        lines.append(f"| {n} | {d.get('SQ_WAIT_ANY', 0) / wc:.2f} | {d.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | {d.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | "
                     f"{d.get('SQ_ACTIVE_INST_VALU', 0) / wc:.2f} | {(mf / simd_cycles if simd_cycles else 0):.2f} | "
                     f"{(d.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / mf if mf else 0):.2f} | "
                     f"{(d.get('SQ_INSTS_VALU', 0) / d['SQ_INSTS_MFMA'] if d.get('SQ_INSTS_MFMA') else 0):.1f} | {hbm:.4g} |")
    return lines


def main():
    root, md_path, json_path = sys.argv[1:4]
    fwd, sde, stress = collect(root, "pmc_fwd_"), collect(root, "pmc_sde_"), collect(root, "pmc_stress_")
    lines = [f"# SQ / TCC counters, {os.environ.get('ROUND', 'this')} build (fp16x3 split precision, fused edge attention with matrix-core first "
             "layers, Philox4x32-7)", ""]
    lines += table("forward, 32 scenes x 256 agents, K=6, 20 steps, one stream",
                   "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --windows 1 "
                   "--no-cpu-baseline --no-train-step --no-secondary --streams 1   (one pass per counter set)", fwd)
    lines += [""] + table("step-granular SDE step, 786 432 rows", "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- "
                          "python3 tools/sde_step_bench.py 786432 30", sde)
    if stress:
        lines += ["", "## stress shape (BASELINE configs[4]): 8 scenes x 1024 agents, K=20, 50 steps, bf16 state storage -- HBM counters", "",
                  "    rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/stress_forward.py --storage bf16 --iters 3", "",
                  "| kernel | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes (2 x FETCH + WRITE) |", "|---|---|---|---|"]
        for n in sorted(stress, key=lambda k: -(2 * stress[k].get("FETCH_SIZE", 0) + stress[k].get("WRITE_SIZE", 0)))[:12]:
            fr, wr = stress[n].get("FETCH_SIZE", 0.0), stress[n].get("WRITE_SIZE", 0.0)
            lines.append(f"| {n} | {fr:.0f} | {wr:.0f} | {(2 * fr + wr) * 1024:.4g} |")
    with open(md_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    doc = {}
    for tag, key in (("k_edge_kv[aa]", "k_edge_attn2"), ("k_global_attn", "k_global_attn"), ("k_seg_merge", "k_seg_merge")):
        cand = [n for n in fwd if key in n and ("FETCH_SIZE" in fwd[n] or "WRITE_SIZE" in fwd[n])]
        if cand:
            n = max(cand, key=lambda x: fwd[x].get("FETCH_SIZE", 0) + fwd[x].get("WRITE_SIZE", 0))
            fr, wr = fwd[n].get("FETCH_SIZE", 0.0), fwd[n].get("WRITE_SIZE", 0.0)
            doc[tag] = {"kernel": n, "FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB": wr, "hbm_bytes_per_launch": int((2 * fr + wr) * 1024),
                        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the bench command at --streams 1; FETCH doubled "
                                "per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B)"}
    cand = [n for n in sde if "k_sde_step" in n]
    if cand:
        n = cand[0]
        fr, wr = sde[n].get("FETCH_SIZE", 0.0), sde[n].get("WRITE_SIZE", 0.0)
        doc["k_sde_step@786432"] = {"kernel": n, "FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB": wr, "hbm_bytes_per_launch": int((2 * fr + wr) * 1024),
                                    "algorithmic_bytes_per_launch": 786432 * 512}
    with open(json_path, "w") as f:
        json.dump(doc, f, indent=1)
    print("\n".join(lines[-12:]))


if __name__ == "__main__":
    main()
