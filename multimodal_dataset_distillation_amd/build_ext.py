"""Builds csrc/*.hip + engine.cpp into the in-tree C-ABI shared library libmdd_hip.so for gfx950.

`python -m multimodal_dataset_distillation_amd.build_ext` (or `__graft_entry__.build()`).
hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libmdd_hip.so")
SOURCES = ["flat_ops.hip", "ws.hip", "elementwise.hip", "linear.hip", "head.hip", "conv_gemm.hip",
           "conv_wgrad.hip", "retrieval.hip", "collective.hip", "vit.hip", "attn.hip", "engine.hip"]
HEADERS = ["common.h", "kernels.h", "engine.h", "conv_gemm_epilogue.inc", os.path.join("..", "..", "include", "mdd_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
         "-I", os.path.join(HERE, "..", "include")]


def _digest(paths):
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build_variant(tag, defines, verbose=True):
    """Experiment build: every source compiled with extra -D flags into variants/libmdd_hip.<tag>.so
    (selected at run time with MDD_HIP_LIB=...).  Not used by the product path."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    vdir = os.path.join(HERE, "variants")
    odir = os.path.join(vdir, "_obj_" + tag)
    os.makedirs(odir, exist_ok=True)
    flags = FLAGS + ["-D" + d for d in defines]

    def one(s):
        obj = os.path.join(odir, s + ".o")
        r = subprocess.run([hipcc] + flags + ["-c", os.path.join(CSRC, s), "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr[-6000:]))
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, SOURCES))
    lib = os.path.join(vdir, "libmdd_hip.%s.so" % tag)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    if verbose:
        print("[build] variant", lib, flush=True)
    return lib


def build(verbose=True, force=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS if os.path.exists(os.path.join(CSRC, h))]
    hdig = _digest(hdrs)
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        stamp = obj + ".sha1"
        dig = _digest([src]) + hdig
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.exists(stamp)
                and open(stamp).read() == dig):
            continue
        jobs.append((src, obj, stamp, dig))

    def compile_one(job):
        src, obj, stamp, dig = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-6000:]))
        with open(stamp, "w") as f:
            f.write(dig)
        if verbose:
            print("[build] compiled", os.path.basename(src), flush=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("[build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":   # --variant TAG DEF1 DEF2=3 ...
        build_variant(sys.argv[2], sys.argv[3:])
    else:
        build(force="--force" in sys.argv)
