"""Build libplatipy_hip.so (gfx950) in-tree with hipcc.

The shared library is git-ignored but travels with the tree to the GPU box.  `python -m
platipy_amd._build` or `__graft_entry__.build()` rebuilds it when a source is newer.
"""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libplatipy_hip.so")
SOURCES = ["pp_api.hip", "pp_fir.hip", "pp_resample.hip", "pp_demons.hip", "pp_iir.hip", "pp_fusion.hip", "pp_cc.hip", "pp_dist.hip", "pp_morph.hip", "pp_linear.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "platipy_amd.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wall", "-Wno-unused-function"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_hip(force=False, verbose=False):
    """Compile every .hip source for gfx950 and link libplatipy_hip.so.  Returns the path."""
    if not force and not stale():
        return LIB
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libplatipy_hip.so")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
