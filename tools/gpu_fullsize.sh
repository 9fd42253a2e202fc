#!/bin/bash
set +e
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_fullsize.py -m gpu -x -q 2>&1 | tail -5
echo "== 2 ranks on one GPU over gloo (smoke test of the N > 1 bench path only)"
PP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -4
