"""Builds the gfx950 C-ABI library (dualdiff_amd/lib/libdualdiff_hip.so) with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only container; the
resulting .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIBNAME = "libdualdiff_hip.so"
SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "elementwise.hip", "xattn.hip", "gemm8.hip", "tokens.hip"]
# per-source extra flags: the attention softmax lives on the MFMA results, so ask LLVM for the
# VGPR-destination form of MFMA (gfx950 has a unified register file) instead of AGPR accumulators
# that cost a v_accvgpr_read/write per touched element.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-honor-nans"],
               # round 5: the same form for the GEMM / conv kernels — four-wave kernels with a 512-register budget otherwise get
               # AGPR accumulators (and, in dd_gemm3's rotating schedule, v_accvgpr shuffles per K-step).  Same-box A/B, three
               # alternating rounds: fp16 87.32 -> 87.34, bf16 89.80 -> 89.97, 4 scenes batched 112.7 -> 113.4 (+0.6 %)
               "gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def lib_path():
    # DD_HIP_LIB: load an alternative build of the same C-ABI (A/B experiments on kernel variants)
    return os.environ.get("DD_HIP_LIB") or os.path.join(LIBDIR, LIBNAME)


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, "..", "include", "dualdiff_hip.h"))
    return deps


def needs_build():
    lib = lib_path()
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps() if os.path.exists(d))


def build_native(force=False, verbose=True):
    """Compile every HIP source for gfx950 and link the shared library. Returns its path."""
    if not force and not needs_build():
        return lib_path()
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    # -O2, not -O3: same arithmetic, 0.5 % smaller code and +0.4 % on the step in three alternating same-box pairs (86.49 /
    # 86.55 / 86.56 against 86.19 / 86.17 / 86.09; -Os the same; profiles/r04_experiments.txt #5) — the kernels' hot loops are
    # unrolled by hand, what -O3 adds is code around them that every launch fetches cold
    flags = ["--offload-arch=" + ARCH, "-O2", "-std=c++17", "-fPIC", "-Wno-unused-value",
             "-DNDEBUG"] + os.environ.get("DD_HIP_DEFINES", "").split()

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + flags + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = lib_path() + ".tmp"
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(tmp, lib_path())
    if verbose:
        print("[dualdiff_amd] built %s" % lib_path(), file=sys.stderr)
    return lib_path()


if __name__ == "__main__":
    build_native(force="--force" in sys.argv)
