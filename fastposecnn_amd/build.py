"""Build libfpc_hip.so (gfx950) in-tree with hipcc.  `python -m fastposecnn_amd.build`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun
snapshot.  -ffp-contract=off: every float op is separately rounded, matching the CPU oracle
bit for bit on the integer-valued outputs (inlier counts, winners, class ids, labels).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libfpc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(HERE, "..", "include")]


# per-file additions.  vote_count.hip: the MFMA results of k_vote_count are consumed by the VALU at once, so they must
# land in VGPRs (the default AGPR form costs one v_accvgpr_read per element), and SLP-packing its f32 subtractions into
# v_pk_add_f32 loses the |.| source modifier (52 extra v_and per step).  The flag is experimental: it is kept away from
# every other kernel (k_vote_plan built with it faulted on a null `keep` it had checked for).
FILE_FLAGS = {"vote_count.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
              # one wave per SIMD beside its own MFMAs: a packed f32 instruction costs ~13 cycles more than the two scalar ones it
              # replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); plain -O3 SLP-packs the splits' subtractions
              "wino_w4.hip": ["-fno-slp-vectorize"], "wino128.hip": ["-fno-slp-vectorize"], "wino_h2.hip": ["-fno-slp-vectorize"], # wino_h3.hip: its 48-slot loop body is unrolled from four nested loops whose body names every slot's item; the size estimate
              # BEFORE the slot conditions fold exceeds the default pragma-unroll threshold (the loop then stays rolled and the accumulators
              # go to scratch)
              "wino_h3.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=4000000"]}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    inc = os.path.join(HERE, "..", "include")
    hdrs += [os.path.join(inc, f) for f in os.listdir(inc)]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, objdir, dep_mtime, force, extra):
    obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), dep_mtime):
        return obj
    subprocess.check_call([HIPCC, *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), *extra, "-c", src, "-o", obj])
    return obj


def build(force=False, verbose=False, extra=()):
    """Objects are cached per flag set (a diagnostic build with -DFPC_STAMP_* never shares objects with the product
    build), and the library records the flag set it was linked from: a different one relinks."""
    extra = list(extra)
    tag = hashlib.sha256(" ".join([HIPCC, *FLAGS, repr(sorted(FILE_FLAGS.items())), *extra]).encode()).hexdigest()[:12]
    objdir = os.path.join(OBJ, tag)
    os.makedirs(objdir, exist_ok=True)
    srcs = sources()
    dep = _deps_mtime()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, objdir, dep, force, extra), srcs))
    stamp = os.path.join(OBJ, "linked_from")
    linked = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if (force or linked != tag or not os.path.exists(LIB)
            or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs)):
        subprocess.check_call([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, "-lz"])      # zlib: png_decode.hip
        with open(stamp, "w") as f:
            f.write(tag)
        from . import isa_lint
        try:
            isa_lint.check(LIB)          # no `s_mov vcc` next to a v_div_fmas (csrc/common.hpp: div_ieee)
        except FileNotFoundError as e:   # no llvm-objdump on this machine: the library is usable, the lint did not run
            print("fastposecnn_amd.build: ISA lint skipped (%s)" % e, file=sys.stderr)
        except Exception:
            os.remove(stamp)             # the next build links and lints again
            raise
    if verbose:
        print("built", LIB, "(diagnostic flags: %s)" % " ".join(extra) if extra else "")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
