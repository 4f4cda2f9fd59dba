#!/usr/bin/env python3
"""Builds tensor4all-rs_amd/lib/libt4a_gpu.so (gfx950) with hipcc.

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the bit-exact pivot contract needs
separately rounded multiply/subtract exactly like the Rust reference.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "build")
SOURCES = ["kernels_rrlu.hip", "kernels_rrlu_reg.hip", "rrlu_xcd_plan.hip", "kernels_rrlu_xcd2.hip", "kernels_rrlu_xcd2_group.hip", "kernels_rrlu_xcd2m.hip", "kernels_rrlu_wg.hip", "kernels_rrlu_wg_group.hip", "kernels_rrlu_w1.hip", "kernels_rrlu_global.hip", "kernels_pi.hip", "kernels_chain.hip", "kernels_small.hip", "kernels_dense.hip", "kernels_linalg.hip",
           "kernels_tt.hip", "pool.hip", "engine.hip", "rook.hip", "tt.hip", "globalsearch.hip", "tci2.hip", "tci2_chain.hip", "tci2_small.hip", "conversion.hip", "patching.hip", "tree.hip", "quantics.hip", "tensorops.hip", "aci.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-fvisibility=hidden"] + os.environ.get("T4A_EXTRA_FLAGS", "").split()  # e.g. -DT4A_RRLU_TRACE (tools/trace_arrivals.py)
# per-source flags.  kernels_dense.hip: keep MFMA accumulators in VGPRs — in AGPR form the compiler moves all of them between the
# two register files at every k-step of the GEMM loop (32 v_accvgpr reads + writes behind a pipeline drain)
# kernels_linalg.hip (Jacobi SVD, Householder QR: compared at tolerances, no pivot decision hangs on a rounding): fused multiply-adds —
# its one-workgroup kernels are bound by the instructions they issue, a separately rounded multiply and add is two of them
FILE_FLAGS = {"kernels_dense.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "kernels_linalg.hip": ["-ffp-contract=fast"]}
HEADERS = ["common.hpp", "stdrng.hpp", "pishard.hpp", "kernels.hpp", "kernels_rrlu_xcd_common.hpp", "kernels_rrlu_w1_body.hpp", "engine.hpp", "tci2.hpp", "tt.hpp", "globalsearch.hpp", "rook.hpp", "patching.hpp", "tree.hpp", "quantics.hpp", "tensorops.hpp", "aci.hpp", "../../include/t4a_gpu.h",
           "../../include/t4a_testfunctions.h"]


_INC_CACHE = {}


def _includes(path):
    """Headers a source includes with quotes, transitively (only files that exist next to it / relative to it)."""
    path = os.path.normpath(path)
    if path in _INC_CACHE:
        return _INC_CACHE[path]
    _INC_CACHE[path] = set()
    found = set()
    try:
        with open(path) as f:
            text = f.read()
    except OSError:
        return found
    import re
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        cand = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if os.path.exists(cand):
            found.add(cand)
            found |= _includes(cand)
    _INC_CACHE[path] = found
    return found


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    # objects built with other flags (T4A_EXTRA_FLAGS: trace / A-B / development builds) must not be mixed into this build
    stamp = os.path.join(OBJ, "flags.stamp")
    flags_now = " ".join(FLAGS) + " | " + repr(sorted(FILE_FLAGS.items()))
    try:
        with open(stamp) as f:
            flags_then = f.read()
    except OSError:
        flags_then = None
    if flags_then != flags_now:
        force = True
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + sorted(_includes(src))):
            jobs.append([hipcc] + FLAGS + FILE_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    lib = os.path.join(OUT, "libt4a_gpu.so")
    if force or jobs or not os.path.exists(lib):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    with open(stamp, "w") as f:
        f.write(flags_now)
    # test-hook twin (tests/test_gpu_chain.py: fault injection into the issue of a pending fill_site_tensors): tci2.hip compiled with
    # -DT4A_TEST_HOOKS, every other object shared with the production library — which therefore carries no injector
    hooks = os.path.join(OUT, "libt4a_gpu_testhooks.so")
    src = os.path.join(CSRC, "tci2.hip")
    hobj = os.path.join(OBJ, "tci2_testhooks.obj")
    if force or jobs or _newer(hobj, [src] + sorted(_includes(src))) or not os.path.exists(hooks):
        run([hipcc] + FLAGS + ["-DT4A_TEST_HOOKS", "-c", src, "-o", hobj])
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", hooks] + [o for o in objs if os.path.basename(o) != "tci2.o"] + [hobj])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
