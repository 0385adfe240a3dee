"""Build libmoe_gfx950.so in-tree with plain hipcc (no torch extension, no pybind): the C ABI of
``include/moe_gfx950.h`` is the whole interface.  ``python -m moe_llava_amd.build`` or
``__graft_entry__.build()``."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmoe_gfx950.so")
SOURCES = ["moe_routing.hip", "moe_routing_top1.hip", "moe_gemm.hip", "moe_gemv.hip", "moe_wgrad.hip", "moe_backward.hip", "moe_block.hip"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def kernel_source_sha() -> str:
    """sha256 over the kernel sources and the header (what a profile of the library was taken on)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(PKG_DIR), "include", "moe_gfx950.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(PKG_DIR), "include", "moe_gfx950.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = True, extra_flags=(), jobs: int = 0) -> str:
    """Compile every source to an object file (in parallel, only the stale ones unless ``force``) and link them."""
    if not force and not needs_build() and not extra_flags:
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = hipcc_path()
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    common = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    common.append(os.path.join(os.path.dirname(PKG_DIR), "include", "moe_gfx950.h"))
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", *extra_flags]
    stamp = os.path.join(objdir, "flags.txt")
    flag_text = " ".join(flags)
    if not os.path.exists(stamp) or open(stamp).read() != flag_text:
        force = True

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        deps = [path, *common]
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
            cmd = [hipcc, *flags, "-c", path, "-o", obj]
            if verbose:
                print("[moe_llava_amd.build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    jobs = jobs or min(len(SOURCES), int(os.environ.get("MAX_JOBS", "0")) or os.cpu_count() or 4, 8)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    with open(stamp, "w") as f:
        f.write(flag_text)
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", LIB_PATH]
    if verbose:
        print("[moe_llava_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv,
                  extra_flags=[a for a in sys.argv[1:] if a.startswith("-") and a != "--force"])
    print(LIB_PATH)
