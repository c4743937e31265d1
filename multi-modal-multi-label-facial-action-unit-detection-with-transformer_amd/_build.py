"""Build libavformer_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libavformer_hip.so"
SOURCES = ["api.hip", "norm_elem.hip", "gemm_f32.hip", "attn_f32.hip", "attn_f32_mfma.hip", "attn_f32x3.hip", "gemm_bf16.hip", "gemm_ws.hip", "gemm_mx8.hip", "attn_bf16.hip", "attn_bwd_merged.hip", "layer.hip", "optim.hip", "layer_small.hip", "heads.hip"]
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "gemm_nt.hpp"), os.path.join(os.path.dirname(HERE), "include", "avformer_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# per-source extra flags.  attn_bwd_merged.hip: MFMA results in architectural VGPRs (the kernel pins its long-lived
# accumulators to AGPRs itself; see the comment at m_mfma_pair_acc)
# ... and no packed fp32 forms there (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32, which hipcc selects on its own): beside an MFMA
# stream a packed instruction gets 42 % of its issue rate, a scalar one 77-88 % (tools/diag/mfma_valu_overlap.hip), and in
# that kernel ONE wave per SIMD issues both streams.  C3: -0.8 % of the step (profiles/ab/r04_attn_bwd_unpacked.json).  The
# feature is passed to both compilation passes; the host pass answers "not a recognized feature for this target (ignoring)".
EXTRA_FLAGS = {"gemm_f32.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attn_f32_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attn_f32x3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "attn_bwd_merged.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]}


def built_lib_path() -> str:
    """where build() puts the library"""
    return os.path.join(LIBDIR, LIBNAME)


def lib_path() -> str:
    """what _lib.load() opens.  AVF_LIB_PATH: another build of the library (tuning aid: A/B of two builds inside one GPU-box
    call, e.g. a copy of the previous build kept as lib/ab_base.so)"""
    return os.environ.get("AVF_LIB_PATH") or built_lib_path()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip source to an object and link the shared library.  Returns its path.
    Serialised across processes with a file lock (eight ranks of a node may import at once)."""
    import fcntl
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS + [os.path.abspath(__file__)]):  # (this file holds the flags)
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("compiled", os.path.basename(s))
        return o

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    out = built_lib_path()
    if force or jobs or _stale(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("linked", out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))
