"""Build libresshift_hip.so (gfx950) in-tree with hipcc.  No GPU is needed: hipcc cross-compiles.

    python -m resshift_amd.build [--force]

The shared library lands next to this file so that it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build" + ("_" + hashlib.sha256(os.environ["RS_BUILD_DEFS"].encode()).hexdigest()[:8] if os.environ.get("RS_BUILD_DEFS") else ""))
LIB = os.environ.get("RS_BUILD_OUT") or os.path.join(HERE, "libresshift_hip.so")   # (RS_BUILD_OUT / RS_BUILD_DEFS: A/B builds of one source tree)
SOURCES = ["igemm.hip", "igemm2.hip", "igemm3.hip", "igemm4.hip", "igemm4s.hip", "wino.hip", "igemm_split.hip", "swin_mlp.hip", "direct_conv.hip", "norm_attn.hip", "win_attn_split.hip", "ae_attn.hip", "ae_attn_split.hip", "elementwise.hip", "engine.hip"]
HEADERS = ["common.h", "igemm_common.h", "igemm4_kernel.h", "gn_tail.h", os.path.join("..", "..", "include", "resshift_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
FLAGS += [f"-D{d}" for d in os.environ.get("RS_BUILD_DEFS", "").split() if d]
if os.environ.get("RS_BUILD_ABLATE"):   # extra instantiations for the K-loop timing ablations (scripts/igemm_bench.py + RS_IGEMM_DBG)
    FLAGS.append("-DRS_SPLIT_ABLATE")


def _digest() -> str:
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    cmd = [HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    stamp = LIB + ".stamp"   # beside the library: the object directory does not travel to the GPU box (.gpurunignore)
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    if verbose:
        print(f"[resshift_amd.build] compiling {len(SOURCES)} HIP sources for gfx950 ...", flush=True)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    # the library must resolve all of its own symbols (catches dropped kernel stubs); needs only libamdhip64, no GPU
    chk = subprocess.run([sys.executable, "-c", f"import ctypes; ctypes.CDLL({LIB!r})"], capture_output=True, text=True)
    if chk.returncode != 0:
        raise RuntimeError(f"built library does not load:\n{chk.stderr[-2000:]}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    if verbose:
        print(f"[resshift_amd.build] built {LIB}", flush=True)
    return LIB


def build_testhooks(verbose: bool = False) -> str:
    """libresshift_hip_testhooks.so: the same objects with engine.hip recompiled under -DRS_TEST_HOOKS (the RS_FAKE_DEVICE plumbing hook of
    tests/_fake_device_plumbing.py).  Never loaded by the product: resshift_amd._lib loads libresshift_hip.so unless RESSHIFT_HIP_LIB says otherwise."""
    build(force=False, verbose=verbose)
    out = os.path.join(HERE, "libresshift_hip_testhooks.so")
    stamp = out + ".stamp"
    dig = _digest() + "+hooks"
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return out
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES if s != "engine.hip"]
    if not all(os.path.exists(o) for o in objs):   # (the library was built elsewhere: compile everything once)
        os.makedirs(OBJ, exist_ok=True)
        with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
            list(ex.map(_compile, [s for s in SOURCES if s != "engine.hip"]))
    hobj = os.path.join(OBJ, "engine_testhooks.o")
    r = subprocess.run([HIPCC, *FLAGS, "-DRS_TEST_HOOKS", "-c", os.path.join(CSRC, "engine.hip"), "-o", hobj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for engine.hip (-DRS_TEST_HOOKS):\n{r.stderr[-4000:]}")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, hobj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
