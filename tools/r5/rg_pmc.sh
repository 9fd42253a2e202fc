#!/bin/bash
# round 5: what the recursive Gaussian's kernels wait for -- SQ counters of the three directional passes (stand-alone, 512 x 512 x 256)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
OUT=gpurun_out/rg_pmc; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OLDPWD/$OUT -o $tag -- $OLDPWD/tools/kbench/sbench $OLDPWD/platipy_amd/csrc/libplatipy_hip.so 512 512 256 2 > $OLDPWD/$OUT/$tag.log 2>&1 )
done
python - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/rg_pmc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_rg_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("| kernel | " + " | ".join(names) + " |")
print("|---|" + "---|" * len(names))
for k in sorted(acc):
    print("| " + k + " | " + " | ".join("%.4g" % (sum(acc[k][c]) / len(acc[k][c])) if c in acc[k] else "" for c in names) + " |")
PY
