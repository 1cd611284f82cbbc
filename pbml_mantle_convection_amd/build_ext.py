"""Builds libmantle_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and
by `python -m pbml_mantle_convection_amd.build_ext`.  The built .so is git-ignored but travels
to the GPU box with the source snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmantle_hip.so")
SOURCES = ["elementwise.hip", "conv_api.hip", "conv_f32.hip", "conv_bf16.hip", "conv_rr_bf16.hip", "loss.hip", "optim.hip"]
HEADERS = ["common.h", "conv_common.h", "conv_rr.h", os.path.join("..", "..", "include", "mantle_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


# conv_rr_bf16.hip: the loader waves' activation transform must stay single-issue f32 FMAs (see rr_gelu_bf16x8)
EXTRA_FLAGS = {"conv_rr_bf16.hip": ["-fno-slp-vectorize"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
