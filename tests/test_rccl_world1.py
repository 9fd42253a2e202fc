"""The RCCL code path on the one GPU there is: the 8-GPU node is the driver's, but a
process group with backend "nccl" and world_size 1 runs every exchange of the multi-atlas path and of bench.py's N > 1
skeleton through the real communicator -- so the first 8-GPU lease cannot die on `device_id=`, a dtype or a tensor on the
wrong device.  The worker runs in its own process (tests/rccl_world1_worker.py); `bench.py --gpus 1` under torchrun too."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return env


@pytest.mark.gpu
def test_multiatlas_exchanges_and_bench_ranks_over_rccl_world_1(gpu_backend):
    port = 33500 + (os.getpid() % 2000)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py")], env=_env(port), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL_WORLD1 ")][-1]
    out = json.loads(line[len("RCCL_WORLD1 "):])
    assert out["ok"] and out["rccl_mapped"], out
    assert len(out["iar_kept"]) >= 3, out
    assert {"crop_allreduce", "fusion_allreduce", "fusion_layout"} <= set(out["exchange_ms"]), out
    assert "fusion_reduce" in out["exchange_ms_reduce"], out


@pytest.mark.gpu
def test_bench_line_under_torchrun_with_one_rank(gpu_backend, tmp_path):
    """`torchrun --nproc-per-node 1 bench.py --gpus 1`: WORLD_SIZE=1 makes bench.Ranks open the RCCL communicator
    (device_id=cuda:0); the line must report rccl_ranks == 1 and the metric of the plain run."""
    port = 35500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2", "--repeats", "2", "--no-cpu-baseline", "--no-registration"]
    env = _env(port)
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["value"] > 0, line
    # the atlas legs ran their exchanges through the communicator (one rank: the all_reduce is the identity, but it is timed)
    assert isinstance(line["multi_atlas"], dict) and "fusion_allreduce" in line["multi_atlas"]["exchange_ms"], line["multi_atlas"]
