"""Builds vampnet_amd/libvampnet_hip.so (gfx950) in-tree with hipcc.  No torch involved."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvampnet_hip.so")
SOURCES = ["engine.hip", "gemm_f32.hip", "attention_f32.hip", "elementwise.hip", "sampling.hip", "conv1d_f32.hip",
           "train.hip", "train_kernels.hip", "attention_train.hip", "torch_rng.hip", "gemm_x3.hip", "attention_x3.hip", "attention_train_x3.hip", "comm.hip", "codec.hip", "codec_plan.hip", "preprocess.hip", "devmem.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(CSRC, h) for h in ("vn_common.h", "vn_model.h", "vn_train.h", "attention_x3_dev.h")] + [
        os.path.join(HERE, "..", "include", "vampnet_hip.h"), os.path.join(HERE, "..", "include", "vampnet_hip_debug.h")]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [hipcc, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
