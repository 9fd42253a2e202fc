import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from platipy_amd import _lib
dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
for streams in (1, 2, 4, 6):
    dt, dice, _ = bench.multi_atlas_streams_leg(ctx, (256, 512, 512), (1.0, 1.0, 1.0), dev, 0, 1, per_gpu=6, streams=streams)
    print(f"6 atlases, {streams} streams: {dt:.3f} s ({360 / dt:.0f} atlases/min), dice {dice:.4f}")
