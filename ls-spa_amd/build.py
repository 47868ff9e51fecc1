"""Build liblsspa_hip.so (hand-written HIP for gfx950) in-tree.

    python ls-spa_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The shared library lands in
``ls-spa_amd/lib/`` (git-ignored, but it travels to the GPU box with the tree).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIBNAME = "liblsspa_hip.so"
SOURCES = ["k_factor.hip", "k_small.hip", "k_lift.hip", "k_gram.hip", "k_error.hip", "lsspa_comm.hip", "lsspa_api.hip",
           "host_perms.cpp"]
HEADERS = ["tiles.h", "kernels.h", "comm.h", os.path.join("..", "..", "include", "lsspa.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# developer A/B builds: LSSPA_CXXFLAGS="-DSOMETHING" python ls-spa_amd/build.py --force --out ls-spa_amd/lib/ab/new.so
FLAGS += os.environ.get("LSSPA_CXXFLAGS", "").split()
# per-file flags.  k_small.hip: its register-resident kernel needs more than 256 registers a wave, and with such a
# budget the compiler gives every matrix instruction an AGPR result by default -- the pivot chain's vector
# instructions then copy both accumulator tiles to VGPRs and back for every pivot (7.3 k instead of 6.0 k cycles per
# 16 x 16 block, measured).  VGPR-form matrix instructions leave only the spill traffic on the AGPR side.
FILE_FLAGS = {"k_small.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the LS-SPA engine needs the ROCm toolchain to build")
    return exe


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _newest_input() -> float:
    paths = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return max(os.path.getmtime(p) for p in paths)


def build_native(force: bool = False, verbose: bool = True, out: str | None = None) -> str:
    out = out or lib_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest_input():
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o").replace(".cpp", ".o"))
        if src.endswith(".cpp"):      # host-only C++ (no device pass: x86 intrinsics inside)
            cmd = [hipcc, "-x", "c++", "-O3", "-std=c++17", "-fPIC", "-Wall", "-c", os.path.join(CSRC, src), "-o", obj]
        else:
            cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"built {out}")
    return out


if __name__ == "__main__":
    build_native(force="--force" in sys.argv,
                 out=os.path.abspath(sys.argv[sys.argv.index("--out") + 1]) if "--out" in sys.argv else None)
