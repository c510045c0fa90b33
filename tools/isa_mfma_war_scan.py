"""Scan a gfx950 assembly listing for vector instructions that overwrite a SrcA / SrcB register of a matrix instruction issued just before
(the write-after-read window of v_mfma_*_16x16x32: see csrc/tile.hpp split_pair).  Prints every such pair with the number of instructions
and of s_nop wait states between them.

    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -DTSDE_NO_SLP=1 -std=c++17 --cuda-device-only -S -o /tmp/x.s trajsde_amd/csrc/node_bwd.hip
    python tools/isa_mfma_war_scan.py /tmp/x.s [max distance in instructions, default 6]"""
import re
import sys


def regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main(path, maxd=6):
    kernel = None
    ins = []
    total = 0
    for line in open(path):
        l = line.strip()
        m = re.match(r"(_Z\w+):", l)
        if m:
            kernel, ins = m.group(1), []
            continue
        if not l or l[0] in ".;" or l.endswith(":"):
            continue
        ins.append(l)
        op = l.split()[0]
        if not (op.startswith(("v_", "ds_read", "global_load", "buffer_load")) and not op.startswith("v_mfma")) or " " not in l:
            continue
        dst = regs(l.split(None, 1)[1].split(",")[0])
        if op.startswith("v_cmp") or not dst:
            continue
        nops = 0
        for back in range(2, maxd + 2):
            if back > len(ins):
                break
            p = ins[-back]
            if p.startswith("s_nop"):
                nops += int(p.split()[1]) + 1
                continue
            if p.startswith("v_mfma"):
                ops = [o.strip() for o in p.split(None, 1)[1].split(",")]
                src = regs(ops[1]) | regs(ops[2])
                if src & dst:
                    total += 1
                    print(f"{kernel[:60]:60s} +{back - 1} instr, {nops} wait states: {p[:70]} | {l[:60]}")
                    break
    print(f"{total} pairs")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6)
