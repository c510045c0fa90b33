"""Price the loop of a kernel from its gfx950 listing with the per-instruction SIMD costs that tools/microbench/vissue.hip measured
(profiles/r06_microbench_vissue.log): the static counterpart of a phase-stamp run.

    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -DTSDE_NO_SLP=1 -DTSDE_PRODUCT=1 -std=c++17 --cuda-device-only -S -o /tmp/attn.s trajsde_amd/csrc/attn.hip
    python tools/issue_model.py /tmp/attn.s k_edge_attn2ILi2ELb0ELb0ELi0ELb1E [--loop N] [--mfma-hold 5.2]

For every backward-branch loop of the kernel (or the N-th largest) it prints the instruction count by cost class and three sums:
  lone      cycles one wave alone needs to ISSUE the loop body: max(4.15, pipe cycles) per vector instruction (a wave issues one
            vector instruction per ~4.15 cycles; an 8-cycle instruction holds it 8), 16 per 16x16x32 matrix instruction when they run
            back to back (the pipe), LDS / memory / scalar instructions at their issue cost;
  pipe      SIMD cycles of the vector PIPE: 2 per full-rate instruction (two or more waves interleave them), 4 and 8 for the others;
  port      SIMD cycles of the shared issue port per wave: the vector pipe cycles above + `mfma-hold` cycles per matrix instruction
            (vissue "V wave beside an M wave": a 16x16x32 instruction takes ~5.2 cycles of its partner's issue).
  pair      SIMD cycles of one iteration of BOTH waves of a two-wave SIMD as vissue's "both alternate 48 mfma / 192 V" mode runs them:
            2 x matrix pipe + 2 x sum of what a vector instruction of each kind ADDS to the matrix pipe there ((alternate - 1536) / 384:
            1.35 full-rate, 2.68 half-rate, 6.97 quarter-rate): two free-running waves that alternate matrix and vector phases overlap
            that much and no more, whatever the kind -- the empirical floor of a two-wave kernel with this instruction mix.
The matrix pipe itself (16 cycles per 16x16x32, 32 per 16x16x4 f32 / 32x32x16) is printed beside them: with W waves per SIMD a loop
iteration of all W waves cannot take less than max(W * matrix pipe, W * port)."""
import collections
import re
import sys

FULL = {"v_fma_f32", "v_mul_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_bitop3_b32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b64", "v_add_co_u32", "v_addc_co_u32",
        "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_lshrrev_b32", "v_not_b32", "v_and_or_b32", "v_or3_b32", "v_add3_u32", "v_xad_u32", "v_xor3_b32"}
QUARTER = {"v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_exp_f32", "v_log_f32", "v_rsq_f32", "v_rcp_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32",
           "v_permlane16_swap_b32", "v_permlane32_swap_b32", "v_fma_f16", "v_rcp_iflag_f32"}
SIXTEEN = set()
WIDE = {"v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add_u64", "v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32"}   # 64-bit results: 5.3 lone, 4.5 beside a second wave


def base(op):
    return re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)


def classify(op):
    b = base(op)
    if b.startswith("v_mfma"):
        return "mfma"
    if b.startswith("ds_"):
        return "lds"
    if b.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if b.startswith("s_"):
        return "scalar"
    if not b.startswith("v_"):
        return None
    if b in FULL:
        return "v2"
    if b in QUARTER:
        return "v8"
    if b in WIDE:
        return "v16"
    return "v4"                                                 # cvt / max / min / cndmask / perm / shifts / packed / mix_f32 / cmp / ...


def mfma_pipe(op):
    return 32 if ("16x16x4" in op or "32x32x16" in op or "32x32x2" in op) else 16


def loops_of(body):
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
    out = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), len(body)) < i:
            out.append((labels[m.group(1)], i))
    return out


def main():
    args = sys.argv[1:]
    path, key = args[0], args[1]
    which = int(args[args.index("--loop") + 1]) if "--loop" in args else None
    hold = float(args[args.index("--mfma-hold") + 1]) if "--mfma-hold" in args else 5.2
    lines = open(path).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if re.match(r"_ZN4tsde\d+" + re.escape(key) + r".*:", l))
    end = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[i0:end]
    loops = sorted(loops_of(body), key=lambda ab: ab[0] - ab[1])
    if which is not None:
        loops = loops[which:which + 1]
    for a, b in loops[:3]:
        cls, ops, pipe_m = collections.Counter(), collections.Counter(), 0
        for l in body[a:b + 1]:
            l = l.strip()
            if not l or l.startswith((";", ".")) or l.endswith(":"):
                continue
            op = l.split()[0]
            c = classify(op)
            if c is None:
                continue
            cls[c] += 1
            ops[(c, base(op))] += 1
            if c == "mfma":
                pipe_m += mfma_pipe(op)
        nv = cls["v2"] + cls["v4"] + cls["v8"] + cls["v16"]
        lone_v = 4.15 * (cls["v2"] + cls["v4"]) + 8.15 * cls["v8"] + 5.3 * cls["v16"]
        pipe_v = 2.07 * cls["v2"] + 4.08 * cls["v4"] + 8.07 * cls["v8"] + 4.5 * cls["v16"]
        other = 4.0 * cls["lds"] + 4.0 * cls["vmem"]
        print(f"loop lines {a}..{b}: {nv} vector ({cls['v2']} full-rate, {cls['v4']} half, {cls['v8']} quarter, {cls['v16']} 64-bit), "
              f"{cls['mfma']} matrix, {cls['lds']} LDS, {cls['vmem']} memory, {cls['scalar']} scalar")
        print(f"  matrix pipe              {pipe_m:8.0f} cycles per wave-iteration")
        print(f"  lone wave, issue         {lone_v + pipe_m + other:8.0f}   (vector {lone_v:.0f} + matrix pipe {pipe_m} + LDS / memory issue {other:.0f})")
        print(f"  vector pipe (>= 2 waves) {pipe_v:8.0f}")
        print(f"  issue port (>= 2 waves)  {pipe_v + hold * cls['mfma'] + other:8.0f}   (vector pipe + {hold} x matrix + LDS / memory issue)")
        add_v = 1.35 * cls["v2"] + 2.68 * cls["v4"] + 6.97 * cls["v8"] + 2.7 * cls["v16"]
        print(f"  pair, alternating waves  {2 * pipe_m + 2 * add_v:8.0f}   (2 x matrix pipe {2 * pipe_m} + 2 x vector additions {add_v:.0f}; both waves of a two-wave SIMD)")
        top = sorted(ops.items(), key=lambda kv: -kv[1])[:28]
        print("  " + ", ".join(f"{n} {o}[{c}]" for (c, o), n in top))


if __name__ == "__main__":
    main()
