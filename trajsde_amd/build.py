"""Build libtrajsde_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m trajsde_amd.build [--force]
    TRAJSDE_SPLIT=bf16x6 python -m trajsde_amd.build --force     # three bf16 pieces / six products instead of fp16x3
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtrajsde_hip.so")
# The product library carries ONE form of every kernel (-DTSDE_PRODUCT).  The measured-slower alternative forms that are kept as
# cross-checks -- the 32x32x16 edge attention (edge32.hip), the software-pipelined and the one-tile edge attention, the matrix-core
# global attention (gattn.hip) -- are compiled into a second library of the same C-ABI, loaded (TRAJSDE_LIB) by the tests and A/B
# tools that select them with their TRAJSDE_* switches; the product library refuses those switches.
ALT_LIB = os.path.join(HERE, "variants", "libtrajsde_alt.so")
ALT_SOURCES = ("attn.hip", "edge32.hip", "gattn.hip", "stages.hip")     # the units TSDE_PRODUCT changes
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize is a CORRECTNESS flag here, not a tuning one.  With the SLP vectoriser on, the compiler pairs the fp32 arithmetic
# around the matrix products into packed instructions (v_pk_fma_f32 / v_pk_mul_f32 on register pairs shuffled together by v_pk_mov_b32),
# and identical launches then disagree in their low-order bits: one element of a pair, in lanes 48-63 of a 16-row tile, off by the size
# of ONE dropped split-precision term (2^-11 of it), in a few tiles per million, differently every run -- every backward call of the
# aggregator at 64 x 128 agents, 40 % of the encoder's.  Without it: 0 differing words in 40 calls and across 8 processes; same speed
# (forward +-0, training step -2 %).  The mechanism is not pinned down: the simple hazards measure as the compiler assumes
# (tools/microbench/mfma_war.hip: results readable 7 wait states behind the instruction, also by packed reads; operands free at once).
# (It is also what lets the operand split compile to three instructions per pair: csrc/tile.hpp split_pair.)
# Guards: tests/test_gpu_backward.py *_bitwise_identical, test_full_size_training_step_agrees_between_kernel_forms.  DESIGN.md section 5.
# -DTSDE_NO_SLP=1 ties the sources to the flag: csrc/tile.hpp refuses to compile without it, so no other build recipe
# (a user's HIPCC line, an IDE, a future setup.py) can produce the library with the vectoriser on by accident.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-DTSDE_NO_SLP=1", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# split-precision flavour of the matrix products (csrc/tile.hpp): fp16x3 (default) or the older bf16x6
if os.environ.get("TRAJSDE_SPLIT", "fp16x3") == "bf16x6":
    FLAGS.append("-DTSDE_SPLIT_H3=0")
FLAGS += os.environ.get("TRAJSDE_CXXFLAGS", "").split()       # experiments: extra -D switches of the kernels
# recur.hip: the cooperative recurrence keeps its 256 weight registers per lane in the accumulation registers (pin_agpr) and needs the
# products' accumulators in ordinary registers for that -- the compiler's other choice runs every product through a[0:3] (recur.hip)
PER_FILE_FLAGS = {"recur.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale() -> bool:
    if not os.path.isfile(LIB) or not os.path.isfile(ALT_LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "trajsde_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    objs, alt_objs = [], []
    procs = []
    for src in sources():
        name = os.path.basename(src)
        extra = PER_FILE_FLAGS.get(name, [])
        obj = os.path.join(CSRC, name[:-4] + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([HIPCC, *FLAGS, "-DTSDE_PRODUCT=1", *extra, "-c", src, "-o", obj], stdout=subprocess.PIPE,
                                            stderr=subprocess.STDOUT)))
        if name in ALT_SOURCES:                              # the same unit with the alternative forms compiled in
            alt = os.path.join(CSRC, name[:-4] + ".alt.o")
            alt_objs.append(alt)
            procs.append((src, subprocess.Popen([HIPCC, *FLAGS, *extra, "-c", src, "-o", alt], stdout=subprocess.PIPE,
                                                stderr=subprocess.STDOUT)))
        else:
            alt_objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    os.makedirs(os.path.dirname(ALT_LIB), exist_ok=True)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", ALT_LIB, *alt_objs])
    if verbose:
        print(f"built {LIB}\nbuilt {ALT_LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
