"""Loops that pay one memory latency per iteration: a static scan of gfx950 assembly listings.

    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -DTSDE_NO_SLP=1 -std=c++17 --cuda-device-only -S -o /tmp/x.s trajsde_amd/csrc/x.hip
    python tools/isa_serial_loads.py /tmp/x.s [...]

`for (i ..) dst[i] = f(src[i])` compiles to load, s_waitcnt vmcnt(0), use, branch (HISTORY.md section 5 "Round 5"): however independent the
iterations are, each costs a round trip to L2 / HBM.  For every innermost backward-branch loop the scan reports the number of vector-memory
loads in the body and the largest number of them requested before a `s_waitcnt vmcnt(n)` that waits for (nearly) all of them; loops with one or
two loads in flight and no matrix instructions are listed first -- candidates for requesting a batch of loads before the first use."""
import re
import sys


def kernels(lines):
    name, start = None, 0
    for i, l in enumerate(lines):
        m = re.match(r"(_ZN4tsde\w+):", l)
        if m:
            name, start = m.group(1), i
        elif l.startswith(".Lfunc_end") and name:
            yield name, lines[start:i]
            name = None


def demangle_short(n):
    m = re.match(r"_ZN4tsde(\d+)", n)
    k = int(m.group(1))
    p = len(m.group(0))
    return n[p:p + k] + n[p + k:p + k + 24]


def scan(body):
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), len(body)) < i:
            loops.append((labels[m.group(1)], i))
    inner = [(a, b) for a, b in loops if not any((c > a or d < b) and c >= a and d <= b for c, d in loops if (c, d) != (a, b))]
    out = []
    for a, b in inner:
        loads = mfma = insts = 0
        in_flight = 0
        worst = 0            # largest number of loads outstanding when a wait drains (nearly) all
        waits = 0
        for l in body[a:b + 1]:
            l = l.strip()
            if not l or l[0] in ".;":
                continue
            op = l.split()[0]
            insts += 1
            if "mfma" in op:
                mfma += 1
            if op.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in l:
                loads += 1
                in_flight += 1
            m = re.search(r"vmcnt\((\d+)\)", l)
            if op == "s_waitcnt" and m:
                n = int(m.group(1))
                if in_flight > n:
                    waits += 1
                    worst = max(worst, in_flight)
                    in_flight = n
        if loads:
            out.append((a, b, insts, loads, waits, worst, mfma))
    return out


def main(paths):
    rows = []
    for path in paths:
        lines = open(path).read().split("\n")
        for name, body in kernels(lines):
            for a, b, insts, loads, waits, worst, mfma in scan(body):
                rows.append((worst if worst else 99, mfma > 0, path.split("/")[-1], demangle_short(name), a, b, insts, loads, waits, worst, mfma))
    rows.sort()
    print(f"{'file':18s} {'kernel':44s} {'lines':>13s} {'insts':>6s} {'loads':>6s} {'waits':>6s} {'max in flight':>14s} {'mfma':>5s}")
    for _, _, f, k, a, b, insts, loads, waits, worst, mfma in rows:
        print(f"{f:18s} {k:44s} {a:6d}-{b:<6d} {insts:6d} {loads:6d} {waits:6d} {worst:14d} {mfma:5d}")


if __name__ == "__main__":
    main(sys.argv[1:])
