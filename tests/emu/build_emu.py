"""TEST INFRASTRUCTURE ONLY: compile the unmodified platipy_amd/csrc/*.hip kernel sources with g++
against the CPU stand-in for the HIP runtime (tests/emu/include/hip/hip_runtime.h), producing
tests/emu/_build/libplatipy_emu.so.  The CPU test suite drives the C ABI of that library to check
kernel indexing / LDS / barrier logic against the oracle where no GPU exists.  The product
(platipy_amd/) never loads it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "platipy_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libplatipy_emu.so")
SOURCES = ["pp_api.hip", "pp_fir.hip", "pp_resample.hip", "pp_demons.hip", "pp_iir.hip", "pp_fusion.hip", "pp_cc.hip", "pp_dist.hip", "pp_morph.hip", "pp_linear.hip"]
FLAGS = ["-O2", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-Wno-unknown-pragmas", "-Wno-attributes",
         "-I", os.path.join(HERE, "include")]


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps += [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "include", "hip", "hip_runtime.h"),
             os.path.join(ROOT, "include", "platipy_amd.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OUT, src.replace(".hip", ".o"))
        procs.append((src, subprocess.Popen(["g++"] + FLAGS + ["-x", "c++", "-c", os.path.join(CSRC, src), "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    obj = os.path.join(OUT, "hipemu.o")
    procs.append(("hipemu.cpp", subprocess.Popen(["g++"] + FLAGS + ["-c", os.path.join(HERE, "hipemu.cpp"), "-o", obj],
                                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"g++ failed on {src}:\n{out.decode(errors='replace')}")
    subprocess.check_call(["g++", "-shared", "-pthread", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
