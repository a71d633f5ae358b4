"""Build libcasync_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the
library is a plain C-ABI shared object loaded with ctypes (calipsync_amd/_lib.py)."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libcasync_hip.so")
SOURCES = ["runtime.hip", "gemm.hip", "ops.hip", "ir_fused.hip", "attention.hip", "frame_ops.hip", "engine.hip"]
HEADERS = ["common.h", os.path.join("..", "..", "include", "casync_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the casync HIP engine cannot be built")
    return exe


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _header_paths():
    return [os.path.join(CSRC, h) for h in HEADERS]


def is_stale() -> bool:
    """True when the library is missing or older than any of its sources / headers."""
    return _newer(LIB_PATH, [os.path.join(CSRC, s) for s in SOURCES] + _header_paths())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 (one object per source, in parallel) and link
    calipsync_amd/lib/libcasync_hip.so."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: str):
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if not force and not _newer(obj, [path] + _header_paths()):
            return obj, None
        cmd = [hipcc] + FLAGS + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        return obj, (res.stdout + res.stderr if res.returncode else None)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    errors = [err for _, err in results if err]
    if errors:
        raise RuntimeError("hipcc failed:\n" + "\n".join(errors))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + [obj for obj, _ in results]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
