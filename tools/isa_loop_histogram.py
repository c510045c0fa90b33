"""Instruction histogram of the loops of one kernel in a gfx950 assembly listing.

    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -DTSDE_NO_SLP=1 -std=c++17 --cuda-device-only -S -o /tmp/attn.s trajsde_amd/csrc/attn.hip
    python tools/isa_loop_histogram.py /tmp/attn.s k_edge_attn2ILi512ELb0ELb0E

Prints, per backward-branch loop of the kernel, the number of VALU / matrix / LDS / scalar / memory instructions in the loop
body and the most frequent opcodes -- the static counterpart of the SQ_INSTS_* counters in profiles/."""
import collections
import re
import sys


def main(path, key, top=40):
    lines = open(path).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if re.match(r"_ZN4tsde\d+" + re.escape(key) + r".*:", l))
    end = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[i0:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), len(body)) < i:
            loops.append((labels[m.group(1)], i))
    for ln in body:
        if re.search(r"; (NumVgprs|ScratchSize|Occupancy|NumAgprs|LDSByteSize)", ln) or re.search(r"\.(vgpr_count|sgpr_count)", ln):
            print(ln.strip())
    for a, b in loops:
        c = collections.Counter()
        for l in body[a:b + 1]:
            l = l.strip()
            if not l or l[0] in ".;":
                continue
            c[l.split()[0]] += 1
        cls = lambda p: sum(v for k, v in c.items() if p(k))
        print(f"loop lines {a}..{b}: {sum(c.values())} instructions | VALU {cls(lambda k: k.startswith('v_') and 'mfma' not in k)}"
              f" | MFMA {cls(lambda k: 'mfma' in k)} | LDS {cls(lambda k: k.startswith('ds_'))} | SALU {cls(lambda k: k.startswith('s_'))}"
              f" | VMEM {cls(lambda k: k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))}")
        if sum(c.values()) > 200:
            for k, v in c.most_common(top):
                print(f"     {v:5d}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
