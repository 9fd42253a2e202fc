#!/bin/bash
# Evidence run of a round on the GPU box: the GPU suite, bench.py plain, bench.py under rocprofv3 --kernel-trace --stats, and the
# PMC passes (each counter set in its own run, kernel-trace only) of the SAME command, reduced to profiles-ready files:
#   tools/gpu_final.sh <round number> <commit>
#   gpurun_out/r<N>final/bench.json                 the bench line
#   gpurun_out/r<N>final/bench_torchrun1.json       the same under `torchrun --nproc-per-node 1` (RCCL communicator of one rank)
#   gpurun_out/r<N>final/bench_under_rocprof.json   the bench line of the traced run (HIP-event kernel times to compare)
#   gpurun_out/r<N>final/kernel_stats.md            rocprofv3 per-kernel statistics of that run
#   gpurun_out/r<N>final/pmc.json / pmc_counters.md calibrated HBM-side bytes per launch + every raw counter
#   gpurun_out/r<N>final/standalone_kernels.txt     tools/kbench/sbench: every once-per-level kernel alone, algorithmic GB/s
#   gpurun_out/r<N>final/registration_kernels.md    per-kernel breakdown of ONE whole config-2 registration (the second of two)
# and profiles/round<N>_pmc.json / _pmc_counters.md, which the bench line of the same build reads its measured traffic from.
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROUND=${1:-5}
OUT=gpurun_out/r${ROUND}final
mkdir -p $OUT
COMMIT=${2:-unknown}
BENCH="python bench.py --steps 20 --warmup 3 --no-registration --no-atlas --no-cpu-baseline"
echo "== pytest -m gpu"; PP_STATS_DIR=$OUT/parity_stats timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OLDPWD/$OUT/pmc -o $tag -- bash -c "cd $OLDPWD && $BENCH --pmc-calibration" > $OLDPWD/$OUT/pmc_$tag.log 2>&1 )
  tail -c 200 $OUT/pmc_$tag.log | head -c 200; echo
done
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_counters.md
python tools/pmc_reduce.py $OUT/pmc_counters.md $COMMIT 512 512 256 > $OUT/pmc.json
cat $OUT/pmc.json
# the bench line reads its measured traffic from profiles/round*_pmc.json (matched on the kernel sources' hash): put this
# run's capture there first, so that the committed line and the committed capture come from the same build
cp $OUT/pmc.json profiles/round${ROUND}_pmc.json
cp $OUT/pmc_counters.md profiles/round${ROUND}_pmc_counters.md
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
echo "== bench.py --gpus 1 under torchrun (one-rank RCCL communicator)"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err
tail -c 400 $OUT/bench_torchrun1.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/trace -o trace -- bash -c "cd $OLDPWD && $BENCH" > $OLDPWD/$OUT/bench_under_rocprof.json 2> $OLDPWD/$OUT/trace.err )
python - <<PY > $OUT/kernel_stats.md
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | calls | total us | avg us | min us | max us |\n|---|---|---|---|---|---|")
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"| {k[:110]} | {len(v)} | {sum(v):.1f} | {sum(v)/len(v):.2f} | {min(v):.2f} | {max(v):.2f} |")
PY

echo "== stand-alone kernels"
timeout 300 tools/kbench/sbench platipy_amd/csrc/libplatipy_hip.so 512 512 256 5 > $OUT/standalone_kernels.txt 2>&1
cat $OUT/standalone_kernels.txt
echo "== one whole registration, per kernel"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$OUT/regtrace -o reg -- bash -c "cd $OLDPWD && python tools/profile_registration.py" > $OLDPWD/$OUT/registration_run.log 2>&1 )
grep registration_s $OUT/registration_run.log
python - <<PY > $OUT/registration_kernels.md
import csv, glob, collections, re
rows = []
for f in glob.glob("$OUT/regtrace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fused = [i for i, r in enumerate(rows) if "k_fused2_force_smooth" in r["Kernel_Name"]]
# tools/profile_registration.py runs the registration twice (warm-up, then timed).  The second one starts behind the
# largest device-idle gap between the first run's last fused launch and the second run's first (the host-side
# synchronize + print between the two calls).
lo, hi = fused[len(fused) // 2 - 1], fused[len(fused) // 2]
start = max(range(lo + 1, hi + 1), key=lambda i: int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]))
sel = rows[start:]
agg = collections.defaultdict(list)
for r in sel:
    m = re.search(r"(k_\w+(?:<[^>]*>)?|__amd\w+|at::native::\w+)", r["Kernel_Name"])
    agg[m.group(1) if m else r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
tot = sum(sum(v) for v in agg.values())
print(f"One fast_symmetric_forces_demons_registration at 512x512x256, [8,4,1] x [10,10,10] (second of two runs): {len(sel)} launches, "
      f"{tot / 1e3:.3f} ms of kernel time inside a device span of {span / 1e3:.3f} ms.\n")
print("| kernel | calls | total us | avg us | % of kernel time |\n|---|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"| {k} | {len(v)} | {sum(v):.1f} | {sum(v) / len(v):.2f} | {100 * sum(v) / tot:.1f} |")
PY
head -30 $OUT/registration_kernels.md
echo "== clocks and power under the fused kernels"
( timeout 60 tools/kbench/kbench platipy_amd/csrc/libplatipy_hip.so 512 512 256 3000 "PP_FUSED_MASK=1" > $OUT/kbench_long.txt 2>&1 ) &
sleep 1.5
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -i "power (W)\|sclk\|mclk" ; sleep 0.4; done > $OUT/power_clocks.txt
wait
rocm-smi --showmaxpower 2>&1 | grep -i "max" >> $OUT/power_clocks.txt
cat $OUT/power_clocks.txt $OUT/kbench_long.txt
echo "== randomised parity sweep (HIP vs oracle)"
timeout 600 python tools/fuzz_parity.py 31 16 > $OUT/fuzz_parity.txt 2>&1; tail -3 $OUT/fuzz_parity.txt
echo "== configs end to end"
timeout 900 python tools/run_configs.py > $OUT/configs.txt 2> $OUT/configs.err; cut -c1-300 $OUT/configs.txt
