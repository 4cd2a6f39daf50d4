"""Builds eilev_amd/csrc/libeilev_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "gemm_pp4_ext.hip", "norm.hip", "attention.hip", "misc.hip", "pipeline.hip", "backward.hip", "comm.hip", "gemv.hip"]
HEADERS = ["common.h", "gemm_common.h", "gemm_tiled.h", "gemm_pp4.h", "gemm_w6.h", "gemm_skinny.h", "attn_frame3.h", os.path.join("..", "..", "include", "eilev.h")]
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


def build_hip(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=()) -> str:
    """``variant`` / ``extra_flags``: an A/B build of the same sources with extra -D flags into build/<variant>/ and
    libeilev_hip_<variant>.so (tools/gemm_ab.py compares two libraries in one process on one box)."""
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build", variant) if variant else os.path.join(HERE, "build")
    lib = os.path.join(HERE, f"libeilev_hip_{variant}.so") if variant else LIB
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    # (gemm.hip and gemm_pp4_ext.hip are by far the longest compiles — the persistent kernel's instances — and run side by side)
    units = [(src, src.replace(".hip", ".o"), []) for src in SOURCES]
    for src, obj, extra in units:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, obj)
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([_hipcc(), *FLAGS, *extra, *extra_flags, "-c", s, "-o", o])

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
    if force or jobs or _stale(lib, objs + [os.path.join(HERE, "exports.map")]):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl", "-Wl,--version-script=" + os.path.join(HERE, "exports.map")])
    return lib


if __name__ == "__main__":
    # python build.py [--force] [--variant NAME -DX=1 ...]
    args = [a for a in sys.argv[1:] if a != "--force"]
    name = ""
    if "--variant" in args:
        i = args.index("--variant")
        name = args[i + 1]
        del args[i:i + 2]
    print(build_hip(force="--force" in sys.argv, verbose=True, variant=name, extra_flags=tuple(args)))
