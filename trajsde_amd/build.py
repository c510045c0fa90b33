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
ALT_SOURCES = ("attn.hip", "edge32.hip", "gattn.hip", "gattn_f32.hip", "stages.hip")     # the units TSDE_PRODUCT changes
# The strict-precision twin: the product library with 24-bit operands (three bf16 pieces, six products: -DTSDE_SPLIT_H3=0), same
# C-ABI.  bench.py times it in a child process (`strict24` key of the bench line) so that the headline's precision asterisk --
# fp16x3 operands are 22-bit -- always has a current number beside it.  Not built when TRAJSDE_SPLIT already selects bf16x6.
STRICT_LIB = os.path.join(HERE, "variants", "libtrajsde_strict24.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize is a CORRECTNESS flag here, not a tuning one.  With the SLP vectoriser on, the compiler pairs the fp32 arithmetic
# around the matrix products into packed instructions (v_pk_fma_f32 / v_pk_mul_f32 on register pairs shuffled together by v_pk_mov_b32),
# and identical launches then disagree in their low-order bits: one element of a pair, in lanes 48-63 of a 16-row tile, off by the size
# of ONE dropped split-precision term (2^-11 of it), in a few tiles per million, differently every run -- every backward call of the
# aggregator at 64 x 128 agents, 40 % of the encoder's.  Without it: 0 differing words in 40 calls and across 8 processes; same speed
# (forward +-0, training step -2 %).  The mechanism is not pinned down: the simple hazards measure as the compiler assumes
# (tools/microbench/mfma_war.hip: results readable 7 wait states behind the instruction, also by packed reads; operands free at once).
# (It is also what lets the operand split compile to three instructions per pair: csrc/tile.hpp split_pair.)
# Guards: tests/test_gpu_backward.py *_bitwise_identical, test_full_size_training_step_agrees_between_kernel_forms.  HISTORY.md section 5.
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
_PROBED = {}


def usable_flags(flags):
    """`flags` if this hipcc accepts them, else [] with a warning.  The per-file options are INTERNAL LLVM options (`-mllvm ...`):
    a ROCm that renames or drops one must not fail the whole build -- the kernels stay correct without it, only slower (the
    compiler then routes every product of recur.hip through a[0:3])."""
    key = tuple(flags)
    if key not in _PROBED:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void trajsde_flag_probe() {}\n")
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", *flags, "-c", src, "-o", os.path.join(d, "probe.o")],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        _PROBED[key] = r.returncode == 0
        if not _PROBED[key]:
            print(f"[trajsde_amd.build] warning: {HIPCC} rejects {' '.join(flags)} -- building without it (slower recurrence, same results):\n"
                  + r.stdout.decode()[-400:], file=sys.stderr)
    return list(flags) if _PROBED[key] else []


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _want_strict() -> bool:
    """the strict-precision twin is a product of `build_strict()` (bench.py's `strict24` leg, __graft_entry__.build()) -- not of every
    import: it compiles every unit a third time for an artefact only the bench uses.  TRAJSDE_BUILD_STRICT24=1 puts it back into build()."""
    return "-DTSDE_SPLIT_H3=0" not in FLAGS and os.environ.get("TRAJSDE_BUILD_STRICT24", "0") == "1"


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith(".o")] + [os.path.join(os.path.dirname(HERE), "include", "trajsde_hip.h")]


def _stale() -> bool:
    if not os.path.isfile(LIB) or not os.path.isfile(ALT_LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def _run_all(jobs, what="hipcc"):
    """run the compile commands, at most one per host core at a time (three builds of every unit started at once took ~3x the
    peak host memory of one)"""
    limit = max(2, os.cpu_count() or 2)
    pending, running, outputs = list(jobs), [], []
    while pending or running:
        while pending and len(running) < limit:
            src, cmd = pending.pop(0)
            running.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        src, p = running.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            for _, q in running:
                q.kill()
            raise RuntimeError(f"{what} failed on {src}:\n{out.decode()}")
        outputs.append(out)
    return outputs


def _unit_flags(name):
    return usable_flags(PER_FILE_FLAGS[name]) if (name in PER_FILE_FLAGS and "-DTSDE_SPLIT_H3=0" not in FLAGS) else []


def build_strict(force: bool = False, verbose: bool = True) -> str:
    """variants/libtrajsde_strict24.so: the product library with 24-bit operands (three bf16 pieces, six products), same C-ABI"""
    if "-DTSDE_SPLIT_H3=0" in FLAGS:
        raise RuntimeError("TRAJSDE_SPLIT=bf16x6 already builds the main library strict")
    if not force and os.path.isfile(STRICT_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(STRICT_LIB) for d in _deps()):
        return STRICT_LIB
    jobs, objs = [], []
    for src in sources():
        sobj = os.path.join(CSRC, os.path.basename(src)[:-4] + ".s24.o")
        objs.append(sobj)
        # (without the per-file options: the bf16x6 recurrence pins nothing in the accumulation registers, and this compiler
        #  crashes on its fp32 matrix instructions under -amdgpu-mfma-vgpr-form)
        jobs.append((src, [HIPCC, *FLAGS, "-DTSDE_SPLIT_H3=0", "-DTSDE_PRODUCT=1", "-c", src, "-o", sobj]))
    for out in _run_all(jobs):
        if verbose and out.strip():
            print(out.decode())
    os.makedirs(os.path.dirname(STRICT_LIB), exist_ok=True)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", STRICT_LIB, *objs])
    if verbose:
        print(f"built {STRICT_LIB}")
    return STRICT_LIB


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    objs, alt_objs, jobs = [], [], []
    for src in sources():
        name = os.path.basename(src)
        extra = _unit_flags(name)
        obj = os.path.join(CSRC, name[:-4] + ".o")
        objs.append(obj)
        jobs.append((src, [HIPCC, *FLAGS, "-DTSDE_PRODUCT=1", *extra, "-c", src, "-o", obj]))
        if name in ALT_SOURCES:                              # the same unit with the alternative forms compiled in
            alt = os.path.join(CSRC, name[:-4] + ".alt.o")
            alt_objs.append(alt)
            jobs.append((src, [HIPCC, *FLAGS, *extra, "-c", src, "-o", alt]))
        else:
            alt_objs.append(obj)
    for out in _run_all(jobs):
        if verbose and out.strip():
            print(out.decode())
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    os.makedirs(os.path.dirname(ALT_LIB), exist_ok=True)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", ALT_LIB, *alt_objs])
    if verbose:
        print(f"built {LIB}\nbuilt {ALT_LIB}")
    if _want_strict():
        build_strict(force=True, verbose=verbose)
    return LIB


def build_sanitized(out_lib: str, obj_dir: str) -> str:
    """The product library once more with the HOST side under AddressSanitizer + UndefinedBehaviorSanitizer (device code
    unsanitised: -fno-gpu-sanitize; GPU ASan is not available on this pool).  For the CPU suite (tests/test_cabi_cpu.py): the
    size queries, argument checks, job tables and workspace carving of the entry points run on the host and can be driven
    without a GPU.  Load it in a process started with LD_PRELOAD=<asan_runtime()>."""
    os.makedirs(obj_dir, exist_ok=True)
    san = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-shared-libsan"]
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".san.o")
        objs.append(obj)
        jobs.append((src, [HIPCC, *FLAGS, "-DTSDE_PRODUCT=1", *san, "-c", src, "-o", obj]))
    _run_all(jobs, "hipcc (sanitized)")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *san, "-o", out_lib, *objs])
    return out_lib


def asan_runtime() -> str:
    """path of the shared AddressSanitizer runtime of the compiler that built build_sanitized()'s library"""
    out = subprocess.check_output([os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang++"),
                                   "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()
    if not os.path.isfile(out):
        import glob
        hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
        if not hits:
            raise FileNotFoundError("libclang_rt.asan-x86_64.so")
        out = hits[0]
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--strict" in sys.argv or "--all" in sys.argv:
        build_strict(force="--force" in sys.argv)
