"""Builds libclimsim_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m climsim_amd.build [--force]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libclimsim_amd.so")
SOURCES = ["api.hip", "prep.hip", "gemm.hip", "rec.hip", "head.hip", "mlp_api.hip", "mlp_train.hip", "cnn_api.hip", "cnn_train.hip", "gen.hip", "crps.hip", "evalm.hip", "derive.hip", "online.hip", "phys.hip", "phys_rad.hip", "phys_train.hip", "stoch.hip", "stoch_bwd.hip", "train_api.hip", "train_rec.hip", "train_misc.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "train.h"), os.path.join(CSRC, "pack.h"), os.path.join(CSRC, "rh_to_q.h"), os.path.join(CSRC, "stoch.h"), os.path.join(CSRC, "phys.h"),
           os.path.join(HERE, "..", "include", "climsim_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None, objdir=None):
    """Default: the product library, objects beside the sources.  `extra_flags` / `out` / `objdir` build a DIAGNOSTIC variant
    of the same sources elsewhere (e.g. -DCSA_FAST_GATES=0 -> tools/bin/libclimsim_amd_exactgates.so, loaded through
    CSA_LIB_PATH by tests/reports/stagewise_report.py)."""
    out = out or LIB
    objdir = objdir or CSRC
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([HIPCC] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj])
        objs.append(obj)
    if jobs:      # independent translation units: compile them side by side (CSA_BUILD_JOBS, default = host cores, at most 8)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        nj = max(1, min(int(os.environ.get("CSA_BUILD_JOBS", os.cpu_count() or 1)), 8, len(jobs)))
        with ThreadPoolExecutor(nj) as ex:
            list(ex.map(run, jobs))
    if force or _stale(out, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def build_exact_gates(verbose=False):
    """Diagnostic variant: libm exp2f + IEEE division in the LSTM gate activations instead of v_exp_f32 / v_rcp_f32."""
    d = os.path.join(HERE, "..", "tools", "bin")
    return build(verbose=verbose, extra_flags=["-DCSA_FAST_GATES=0"], out=os.path.join(d, "libclimsim_amd_exactgates.so"),
                 objdir=os.path.join(d, "obj_exactgates"))


if __name__ == "__main__":
    if "--exact-gates" in sys.argv:
        print(build_exact_gates(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
