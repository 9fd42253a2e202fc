#!/bin/bash
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for lib in libplatipy_hip.so libplatipy_abl_NOWARP.so; do
  echo "#### $lib"
  cp platipy_amd/csrc/$lib /tmp/lib_under_test.so
  PP_LIB_OVERRIDE=/tmp/lib_under_test.so python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, json, subprocess
sys.path.insert(0, os.getcwd())
from platipy_amd import _lib
_lib.DEFAULT_LIB = os.environ["PP_LIB_OVERRIDE"]
sys.argv = ["bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-registration", "--no-atlas"]
import runpy, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("ms_per_step %.4f" % d["ms_per_step"])
for k, v in d["kernels"].items(): print("   %-28s %.4f ms" % (k, v["avg_ms"]))
PY
done
