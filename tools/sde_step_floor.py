"""The instruction floor of k_sde_step: instruction classes of one tile-step (16 paths x one Euler-Maruyama step of one wave), read off
the compiler's listing, times the measured issue cost of each class (tools/microbench/, profiles/r03_microbench_*.log) = SIMD cycles,
against the in-kernel phase clocks (profiles/r04_sde_step_phase_stamps.log).  Writes the table of profiles/r05_sde_step_floor.md.

    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -DTSDE_NO_SLP=1 -DTSDE_PRODUCT=1 -std=c++17 --cuda-device-only -S -o /tmp/dec.s trajsde_amd/csrc/decoder.hip
    python tools/sde_step_floor.py /tmp/dec.s [stamped SIMD cycles per tile-step = 5912] [measured HBM fraction = 0.304]
"""
import collections
import re
import sys

# measured issue costs, SIMD cycles per wave-instruction (profiles/r03_microbench_trans_rate.log: 284.8 / 64 and 540.2 / 64 cycles;
# profiles/r03_microbench_coexec.log: 48 matrix instructions per 1536 / 2 cycles; v_mad_u64_u32: tools/microbench, a quarter-rate op;
# the matrix instruction's hold on the issue port: coexec mode 16, one wave issuing 48 x (matrix + 4 fma) in 1044 cycles = 21.75 per
# group, of which the four fma are 17.8 -> ~4 cycles)
COST = {"valu": 4.45, "trans": 8.44, "mad64": 16.0, "mfma_issue": 4.0, "mfma_pipe": 16.0}
TRANS = ("v_exp_", "v_rcp_", "v_log_", "v_sqrt_", "v_sin_", "v_cos_", "v_rsq_")


def main(path, stamped=5912.0, frac=0.304):
    lines = open(path).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if re.match(r"_ZN4tsde10k_sde_stepILb0E.*:", l))      # the fp32-state instantiation
    end = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[i0:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), len(body)) < i:
            loops.append((labels[m.group(1)], i))
    def n_mfma(ab):
        return sum(1 for l in body[ab[0]:ab[1] + 1] if "v_mfma" in l)
    def n_mad(ab):
        return sum(1 for l in body[ab[0]:ab[1] + 1] if "v_mad_u64_u32" in l)
    # the tile loop of the in-kernel-Philox path: innermost loop that holds the products AND the generator
    a, b = min((ab for ab in loops if n_mfma(ab) >= 100 and n_mad(ab) > 0), key=lambda ab: ab[1] - ab[0])
    c = collections.Counter(l.split()[0] for l in (x.strip() for x in body[a:b + 1]) if l and l[0] not in ".;")
    valu = {k: v for k, v in c.items() if k.startswith("v_") and "mfma" not in k}
    trans = sum(v for k, v in valu.items() if k.startswith(TRANS))
    mad64 = valu.get("v_mad_u64_u32", 0)
    plain = sum(valu.values()) - trans - mad64
    mfma = sum(v for k, v in c.items() if "mfma" in k)
    lds = sum(v for k, v in c.items() if k.startswith("ds_"))
    salu = sum(v for k, v in c.items() if k.startswith("s_"))
    rows = [("full-rate vector (fma / add / mul / xor / cvt / select ...)", plain, COST["valu"]),
            ("transcendental (exp, rcp for 64 tanh + 1 sigmoid; log, sqrt, sin, cos of Box-Muller)", trans, COST["trans"]),
            ("v_mad_u64_u32 (Philox4x32-7 multiplies)", mad64, COST["mad64"]),
            ("matrix instructions: hold on the vector issue port", mfma, COST["mfma_issue"])]
    tot = sum(n * k for _, n, k in rows)
    print(f"| class | instructions per tile-step | cycles each | SIMD cycles |\n|---|---|---|---|")
    for name, n, k in rows:
        print(f"| {name} | {n} | {k:g} | {n * k:.0f} |")
    print(f"| **vector issue port, total** | {sum(valu.values())} vector + {mfma} matrix | | **{tot:.0f}** |")
    print(f"| matrix pipe (overlaps the vector work of the SIMD's other waves) | {mfma} | {COST['mfma_pipe']:g} | {mfma * COST['mfma_pipe']:.0f} |")
    print(f"| LDS reads (ds_read_b128: weight fragments), scalar | {lds}, {salu} | | (own pipes) |")
    print()
    print(f"stamped: {stamped:.0f} SIMD cycles per tile-step (four resident waves: {4 * stamped:.0f} cycles of a wave's clock); "
          f"issue floor {tot:.0f} = {tot / stamped:.2f} of it.")
    print(f"measured HBM fraction {frac:.3f} -> at the floor {frac * stamped / tot:.3f}; 0.40 needs <= {stamped * frac / 0.40:.0f} cycles "
          f"per tile-step, i.e. {100 * (1 - stamped * frac / 0.40 / tot):.0f} % fewer issue cycles than the floor itself.")


if __name__ == "__main__":
    main(sys.argv[1], *(float(x) for x in sys.argv[2:4]))
