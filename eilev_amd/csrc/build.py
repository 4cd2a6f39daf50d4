"""Builds eilev_amd/csrc/libeilev_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "misc.hip", "pipeline.hip", "backward.hip", "comm.hip", "patch.hip", "gemv.hip"]
HEADERS = ["common.h", "gemm_a4.h", "gemm_a4_loop.inc", "attn_frame3.h", os.path.join("..", "..", "include", "eilev.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]
LIB = os.path.join(HERE, "libeilev_hip.so")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    gen, inc = os.path.join(HERE, "gen_a4_loop.py"), os.path.join(HERE, "gemm_a4_loop.inc")
    if _stale(inc, [gen]):  # the hand-scheduled K loop of gemm_a4_kernel as inline-asm text
        with open(inc, "w") as f:
            subprocess.check_call([sys.executable, gen], stdout=f)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    # gemm.hip is by far the longest compile (the persistent kernel's instances): two objects, built side by side
    units = [(src, src.replace(".hip", ".o"), []) for src in SOURCES if src != "gemm.hip"]
    units += [("gemm.hip", "gemm.o", ["-DEILEV_GEMM_PART=1"]), ("gemm.hip", "gemm_ext.o", ["-DEILEV_GEMM_PART=2"]),
              ("gemm.hip", "gemm_a4.o", ["-DEILEV_GEMM_PART=3"])]
    for src, obj, extra in units:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, obj)
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([_hipcc(), *FLAGS, *extra, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=len(SOURCES) + 1) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn, file=sys.stderr)
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
