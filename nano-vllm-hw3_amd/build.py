"""Build libqlinear.so (HIP, gfx950) in-tree with hipcc.  No GPU is needed to build."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "nanovllm", "_lib")
LIB = os.path.join(OUT_DIR, "libqlinear.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "qlinear.h"))
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))      # every translation unit of csrc/

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc] + FLAGS + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
