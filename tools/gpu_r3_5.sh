#!/bin/bash
# fixed-sample cache in the linear stage: per-level cost with and without, then the bench's atlas legs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3b
{
echo "== PP_NO_FIXED_SAMPLES=1"
PP_NO_FIXED_SAMPLES=1 timeout 300 python tools/profile_linear.py 2>&1 | grep -v amdgpu.ids
echo "== fixed samples cached (default)"
timeout 300 python tools/profile_linear.py 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3b/bench2.json 2> gpurun_out/r3b/bench2.err
python - <<PY
import json
d = json.load(open("gpurun_out/r3b/bench2.json"))
print({k: d[k] for k in ("ms_per_step", "registration_s")})
print("atlas chain s", d["multi_atlas"]["seconds"], "per min", d["multi_atlas"]["atlases_per_min"], "dice", d["multi_atlas"]["dice_vs_template_label"])
print("4 streams s", d["multi_atlas_streams"]["seconds"], "per min", d["multi_atlas_streams"]["atlases_per_min"])
PY
} 2>&1 | tee gpurun_out/r3b/fixed_samples.txt
