#!/usr/bin/env python
"""bench.py -- demons-iteration throughput on MI355X (BASELINE.json metric, config 2).

A "step" is ONE full fast-symmetric-forces demons iteration (warp, ESM update, smooth update,
add, smooth field, metric/RMS reduction) on a 512x512x256 fp32 volume pair that is already
resident in HBM -- the finest pyramid level of config 2, where 98 % of the reference's voxel
iterations are spent.  `value` = voxels x steps x ranks / wall time, in Mvoxels/s.

  python bench.py [--gpus N --steps K --warmup W]            (N > 1: launched by torch.distributed.run)

N > 1 is the multi-atlas shape of the path: every rank registers its own atlas to the target with
no data-path collective inside the demons loop (weak scaling); timing is barrier-bracketed and the
maximum over ranks.  Extra keys: `roofline` (dominant kernel, algorithmic bytes / HIP-event time),
`cpu_baseline` (the oracle timed on this host's cores, rank 0, N == 1 only), `kernels` (per-kernel
breakdown) and `registration_s` (one whole 3-level config-2 registration through the drop-in API).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from platipy_amd import _lib  # noqa: E402

# Bytes per voxel per launch, two models side by side:
#   compulsory : what the schedule that actually runs must move through HBM -- every input read once, every output
#                written once, halos and gather re-reads excluded.  Fused: A reads F, M.D (8) and writes U (12);
#                B reads D, U, M (28) and writes D', M.D' (16).  The roofline fraction is computed on these (<= 1 by
#                construction).
#   contract   : SURVEY 8(d)'s figure, which counts every separable pass of the STAGED schedule as its own sweep
#                (warp 20, force 20, smooth-update 72, add + first field pass 36, remaining field passes 48 = 196).
#                It describes the staged kernels exactly; for the fused kernels it is reported for continuity only.
# Generation-2 fused kernels: A reads F, M o D and D (4 + 4 + 12) and writes S = D + G_u * update (12); B reads S (12) and
# the moving image once (4) and writes D' (12) and M o D' (4).  With PP_FUSED_SUM=0 (and generation 1) A writes the
# smoothed update alone (20) and B reads D and U (44).  Either way an iteration moves 64 compulsory bytes per voxel.
# The library names what it ran: "k_fused2_*" = SUM mode, "k_fused2_*/sep" = the update stored alone (PP_FUSED_SUM=0, or one of the
# two kernels in generation 1 because of its radius), so the byte model follows the schedule, not an environment variable.
COMPULSORY_BYTES = {
    "k_fused2_force_smooth": 32, "k_fused2_add_smooth_warp": 32, "k_fused2_force_smooth/sep": 20, "k_fused2_add_smooth_warp/sep": 44,
    "k_fused_force_smooth": 20, "k_fused_add_smooth_warp": 44,
    "k_warp_same_grid": 20, "k_demons_force": 20, "k_conv_axis x3 (update)": 72, "k_conv_axis x3 (add+field)": 84,
}
CONTRACT_BYTES = {
    "k_fused2_force_smooth": 20 + 72, "k_fused2_add_smooth_warp": 36 + 48 + 20,
    "k_fused2_force_smooth/sep": 20 + 72, "k_fused2_add_smooth_warp/sep": 36 + 48 + 20,
    "k_fused_force_smooth": 20 + 72, "k_fused_add_smooth_warp": 36 + 48 + 20,
    "k_warp_same_grid": 20, "k_demons_force": 20, "k_conv_axis x3 (update)": 72, "k_conv_axis x3 (add+field)": 84,
}
CONTRACT_BYTES_ITER = 196
COMPULSORY_BYTES_ITER_FUSED = 64
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def kernel_source_sha16():
    """sha256 (first 16 hex digits) over the demons kernel sources: a committed PMC capture is attached to the bench line only
    when it was taken of exactly these sources (`roofline.traffic` must be of HEAD's kernels)."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "platipy_amd", "csrc")
    for f in ("pp_demons.hip", "pp_demons_fused2.h", "pp_warp_sample.h", "pp_internal.h"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc(nx, ny, nz):
    """The newest profiles/round*_pmc.json of this grid whose `kernel_source_sha16` matches the sources in the tree."""
    import glob

    sha = kernel_source_sha16()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if list(pmc.get("size", [])) == [nx, ny, nz] and pmc.get("kernel_source_sha16") == sha:
            pmc["_file"] = os.path.relpath(path, ROOT)
            return pmc
    return None


def synth_pair(ctx, shape, spacing, seed, device, warp_seed=None, label=None):
    """SURVEY 8d synthetic CT pair, generated on the GPU: ellipsoid body, ellipsoidal organs,
    sigma-1.5-voxel blur, N(0,5^2) noise; moving = fixed warped by a smooth <= 6 mm field + noise.
    warp_seed: draw the field (and the moving image's noise) from its own stream, so that several atlases are
    independent warps of one template (seeds 2000 + i).  label: a uint8 template label; the return
    value then gains the label seen through the same field (nearest neighbour), i.e. the atlas's own contour."""
    nz, ny, nx = shape
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.arange(nx, device=device, dtype=torch.float32).view(1, 1, nx)
    y = torch.arange(ny, device=device, dtype=torch.float32).view(1, ny, 1)
    z = torch.arange(nz, device=device, dtype=torch.float32).view(nz, 1, 1)
    vol = torch.full(shape, -1000.0, device=device)
    body = ((x - nx / 2) / (0.42 * nx)) ** 2 + ((y - ny / 2) / (0.40 * ny)) ** 2 + ((z - nz / 2) / (0.46 * nz)) ** 2 < 1
    vol[body] = 0.0
    for _ in range(12):
        c = torch.rand(3, generator=g) * 0.5 + 0.25
        r = torch.rand(3, generator=g) * 0.14 + 0.04
        val = float(torch.rand(1, generator=g)) * 600.0 - 200.0
        m = ((x - float(c[0]) * nx) / (float(r[0]) * nx)) ** 2 + ((y - float(c[1]) * ny) / (float(r[1]) * ny)) ** 2 + \
            ((z - float(c[2]) * nz) / (float(r[2]) * nz)) ** 2 < 1
        vol[m & body] = val
        del m
    del body
    size = (nx, ny, nz)
    clean = torch.empty_like(vol)
    ctx.discrete_gaussian(vol, clean, size, (1, 1, 1), (2.25, 2.25, 2.25), 0.01, 32, False)
    ctx.sync()
    del vol
    gd = torch.Generator(device=device).manual_seed(seed + 1)
    fixed = clean + 5.0 * torch.randn(shape, device=device, generator=gd)
    if warp_seed is not None:
        gd = torch.Generator(device=device).manual_seed(warp_seed)
    coarse = torch.randn((1, 3, 8, 16, 16), device=device, generator=gd)
    dvf = torch.nn.functional.interpolate(coarse, size=shape, mode="trilinear", align_corners=True)[0].contiguous()
    dvf *= 6.0 / float(torch.sqrt((dvf ** 2).sum(0)).max())
    geom = _lib.make_geom(size, spacing)
    moving = torch.empty_like(clean)
    ctx.warp(clean, dvf, geom, -1000.0, moving)
    ctx.sync()
    moving += 5.0 * torch.randn(shape, device=device, generator=gd)
    if label is not None:
        import platipy_amd as pa

        field = pa.Image(dvf, spacing)
        warped = pa.registration.apply_transform(pa.Image(label, spacing), transform=pa.DisplacementFieldTransform(field),
                                                 default_value=0, interpolator=pa.sitkNearestNeighbor).tensor
        del clean, dvf
        return fixed.contiguous(), moving.contiguous(), geom, warped.contiguous()
    del clean, dvf
    return fixed.contiguous(), moving.contiguous(), geom


def multi_atlas_leg(ctx, fixed, moving, spacing, rank, world, device, seed=1234, linear_overrides=None):
    """Config 4 shape: one atlas per GPU, whole chain (quick crop registration, affine, demons, propagation of
    the CT and one structure, local weight map, fusion all-reduce, post-processing) with the reference pipeline's
    default settings (multiatlas/run.py:47-103) except that atlases are already in HBM.  Returns seconds."""
    import copy

    import platipy_amd as pa
    from platipy_amd.projects.multiatlas import MUTLIATLAS_SETTINGS_DEFAULTS, run_segmentation

    nz, ny, nx = fixed.shape
    x = torch.arange(nx, device=device, dtype=torch.float32).view(1, 1, nx)
    y = torch.arange(ny, device=device, dtype=torch.float32).view(1, ny, 1)
    z = torch.arange(nz, device=device, dtype=torch.float32).view(nz, 1, 1)
    label = (((x - 0.5 * nx) / (0.2 * nx)) ** 2 + ((y - 0.5 * ny) / (0.18 * ny)) ** 2 + ((z - 0.5 * nz) / (0.25 * nz)) ** 2 < 1).to(torch.uint8)
    ids = [f"{i:03d}" for i in range(world)]
    # this rank's atlas = the template seen through its own smooth field (seed 2000 + rank) + the template's label seen
    # through the same field
    # (the target is the seed-1234 template on EVERY rank: the pipeline replicates the target, only atlases are sharded)
    fixed, atlas_ct, _, atlas_label = synth_pair(ctx, tuple(fixed.shape), spacing, seed, device, warp_seed=2000 + rank, label=label)
    atlases = {ids[rank]: {"CT Image": pa.Image(atlas_ct, spacing), "HEART": pa.Image(atlas_label, spacing)}}
    st = copy.deepcopy(MUTLIATLAS_SETTINGS_DEFAULTS)
    st["atlas_settings"]["atlas_id_list"] = ids
    st["atlas_settings"]["atlas_structure_list"] = ["HEART"]
    st["label_fusion_settings"]["vote_type"] = "local"
    st["linear_registration_settings"].update(linear_overrides or {})
    target = pa.Image(fixed, spacing)
    run_segmentation(target, st, atlases=atlases)            # warm-up: workspaces, RCCL communicator
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    res, _ = run_segmentation(target, st, atlases=atlases)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    fused = res["HEART"].tensor > 0
    dice = float(2 * (fused & (label > 0)).sum() / (fused.sum() + (label > 0).sum()))
    return dt, int(fused.sum()), dice


def multi_atlas_streams_leg(ctx, shape, spacing, device, rank=0, world=1, per_gpu=4, streams=4, seed=1234, linear_overrides=None):
    """Config 5's shape: `per_gpu` atlases on EVERY GPU (independent warps of one template, seeds 2000 + i), their
    chains overlapped on `streams` HIP streams (one worker thread + pp_ctx per stream), iterative atlas selection when
    there are enough atlases for it (>= 8), then fusion on the survivors.  Returns (seconds, Dice, atlases removed)."""
    import copy

    import platipy_amd as pa
    from platipy_amd.projects.multiatlas import MUTLIATLAS_SETTINGS_DEFAULTS, run_segmentation

    nz, ny, nx = shape
    x = torch.arange(nx, device=device, dtype=torch.float32).view(1, 1, nx)
    y = torch.arange(ny, device=device, dtype=torch.float32).view(1, ny, 1)
    z = torch.arange(nz, device=device, dtype=torch.float32).view(nz, 1, 1)
    label = (((x - 0.5 * nx) / (0.2 * nx)) ** 2 + ((y - 0.5 * ny) / (0.18 * ny)) ** 2 + ((z - 0.5 * nz) / (0.25 * nz)) ** 2 < 1).to(torch.uint8)
    total = per_gpu * world
    ids = [f"{i:03d}" for i in range(total)]
    atlases, target = {}, None
    for i in range(rank, total, world):              # the pipeline's rule: atlas i belongs to rank i % world
        target, ct, _, lab = synth_pair(ctx, shape, spacing, seed, device, warp_seed=2000 + i, label=label)
        atlases[ids[i]] = {"CT Image": pa.Image(ct, spacing), "HEART": pa.Image(lab, spacing)}
    st = copy.deepcopy(MUTLIATLAS_SETTINGS_DEFAULTS)
    st["atlas_settings"]["atlas_id_list"] = ids
    st["atlas_settings"]["atlas_structure_list"] = ["HEART"]
    st["label_fusion_settings"]["vote_type"] = "local"
    st["linear_registration_settings"].update(linear_overrides or {})
    if total >= 8:
        st["iar_settings"]["reference_structure"] = "HEART"
    target = pa.Image(target, spacing)
    run_segmentation(target, st, atlases=atlases, streams_per_gpu=streams)      # warm-up: per-stream workspaces, communicator
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    res, _ = run_segmentation(target, st, atlases=atlases, streams_per_gpu=streams)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    fused = res["HEART"].tensor > 0
    dice = float(2 * (fused & (label > 0)).sum() / (fused.sum() + (label > 0).sum()))
    return dt, dice, list(getattr(run_segmentation, "last_iar_removed", []))


EXCHANGE_LABELS = ["crop_allreduce", "iar_exchange", "fusion_layout", "fusion_allreduce", "fusion_reduce", "contour_allreduce", "other"]


def exchange_ms_over_ranks(ranks, mine):
    """The pipeline's per-label exchange times (projects/multiatlas.py::_Dist), maximum over ranks; a fixed label list so that
    every rank makes the same collective call."""
    return ranks.max_dict(mine or {}, EXCHANGE_LABELS)


def cpu_baseline(fixed, moving, spacing, budget_s=15.0):
    """The reference's CPU path timed beside the GPU, on the metric's own configuration (the full bench pair, not a crop).
    SimpleITK (the reference's own arithmetic) is used when importable; otherwise the oracle (C/OpenMP restatement, fp64
    field like ITK) stands in, labelled "port".  A reported baseline, not a target: the GPU/CPU ratio says nothing about
    kernel quality (the roofline fraction does) and is deliberately not computed here."""
    f = fixed.cpu().numpy()
    m = moving.cpu().numpy()
    nz, ny, nx = f.shape
    nvox = f.size
    try:
        import SimpleITK as sitk  # noqa: F401

        def run(n):
            flt = sitk.FastSymmetricForcesDemonsRegistrationFilter()
            flt.SetNumberOfThreads(os.cpu_count())
            flt.SetSmoothUpdateField(True)
            flt.SetSmoothDisplacementField(True)
            flt.SetStandardDeviations([1.5 / s for s in spacing])
            flt.SetNumberOfIterations(n)
            flt.SetMaximumRMSError(0.0)
            fi, mi = sitk.GetImageFromArray(f), sitk.GetImageFromArray(m)
            fi.SetSpacing(spacing)
            mi.SetSpacing(spacing)
            t0 = time.perf_counter()
            flt.Execute(fi, mi)
            return time.perf_counter() - t0

        kind, cores = "reference", os.cpu_count()
    except ImportError:
        from oracle import oracle as O

        def run(n):
            flt = O.DemonsFilter()
            flt.SetSmoothUpdateField(True)
            flt.SetSmoothDisplacementField(True)
            flt.SetStandardDeviations([1.5 / s for s in spacing])
            flt.SetNumberOfIterations(n)
            flt.SetMaximumRMSError(0.0)
            t0 = time.perf_counter()
            flt.Execute(O.Vol(f, spacing), O.Vol(m, spacing))
            return time.perf_counter() - t0

        kind, cores = "port", O.lib().orc_num_threads()
    t1 = run(1)                                # also pages the arrays in and starts the thread pool
    n = int(max(1, min(10, (budget_s - t1) / max(t1, 1e-3))))
    t = run(n)
    return {"value": nvox * n / t / 1e6, "unit": "Mvoxels/s per demons iter", "cores": cores, "kind": kind,
            "sample": f"the whole {nx}x{ny}x{nz} bench pair, {n} iteration(s) in {t:.1f} s after a 1-iteration warm-up of {t1:.1f} s" +
                      ("" if kind == "reference" else " (SimpleITK unavailable: C/OpenMP fp64 restatement stands in)")}


class Ranks:
    """The N > 1 skeleton of the bench (one process per GPU, launched by torch.distributed.run): process group, barrier,
    max-over-ranks and a gather of every rank's own time.  `backend` "nccl" is RCCL on ROCm; "gloo" exists so that the
    plumbing can be exercised without GPUs (tests/test_bench_plumbing.py) -- the timed work itself never runs on gloo."""

    def __init__(self, world, device, backend=None):
        self.world, self.device, self.dist = world, device, None
        # (also a world of ONE rank when a launcher set WORLD_SIZE -- `torchrun --nproc-per-node 1 bench.py --gpus 1`: the
        # RCCL communicator, `device_id=` and every collective below then run on the one GPU there is, so the plumbing of
        # the N > 1 line is exercised wherever a single MI355X is; a plain `python bench.py` opens no process group)
        launched = all(os.environ.get(k) for k in ("WORLD_SIZE", "RANK", "MASTER_PORT"))     # (a launcher's complete rendezvous)
        if world > 1 or launched:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = backend or os.environ.get("PP_BENCH_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
            self.dist = dist
        self.rank = self.dist.get_rank() if self.dist else 0

    def _cdev(self):
        return self.device if (self.dist and self.dist.get_backend() == "nccl") else "cpu"

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max(self, x):
        if not self.dist:
            return float(x)
        tt = torch.tensor([x], dtype=torch.float64, device=self._cdev())
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return float(tt.item())

    def count(self):
        """Ranks taking part in the job's collectives: an all_reduce(SUM) of ones (collective: every rank calls it)."""
        if not self.dist:
            return 1
        tt = torch.ones(1, dtype=torch.float32, device=self._cdev())
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.SUM)
        return int(round(float(tt.item())))

    def max_dict(self, d, keys):
        """Element-wise maximum over ranks of d[k] for the given keys (missing -> 0); collective."""
        vals = [float(d.get(k, 0.0)) for k in keys]
        if self.dist and keys:
            tt = torch.tensor(vals, dtype=torch.float64, device=self._cdev())
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            vals = [float(v) for v in tt.tolist()]
        return dict(zip(keys, vals))

    def gather(self, x):
        """-> every rank's value, in rank order (on every rank)."""
        if not self.dist:
            return [float(x)]
        mine = torch.tensor([x], dtype=torch.float64, device=self._cdev())
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [float(t.item()) for t in out]

    def timed(self, fn, sync):
        """Barrier + sync on both sides of fn(); -> (max over ranks, [per-rank seconds], load-imbalance fields)."""
        self.barrier()
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        own = time.perf_counter() - t0          # this rank's own work, before it waits for the others
        self.barrier()
        dt = time.perf_counter() - t0
        per_rank = self.gather(own)
        dt = self.max(dt)
        mean = sum(per_rank) / len(per_rank)
        return dt, per_rank, {"per_rank_s": per_rank, "slowest_rank": int(max(range(len(per_rank)), key=per_rank.__getitem__)),
                              "imbalance": (max(per_rank) / mean - 1.0) if mean > 0 else 0.0}

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def self_launch(n):
    """Re-run this command as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same flags>` on
    127.0.0.1 with a free port, and return its exit status.  The children see WORLD_SIZE and do not come here again."""
    import socket
    import subprocess
    import sys

    if os.environ.get("PP_BENCH_SELF_LAUNCHED") == "1":
        raise SystemExit("bench.py: launched itself but WORLD_SIZE is still unset")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, PP_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC: what RCCL needs on this host driver)
    return subprocess.call(cmd, env=env)


def stub_main(args, world, rank):
    """PP_BENCH_STUB=1 (tests/test_bench_plumbing.py only): everything around the timed work -- process group, barrier-
    bracketed timing, maximum over ranks, gather, rank 0's single line -- with a sleep in place of the kernels, so that the
    way `--gpus N` starts can be exercised where no GPU exists.  The line says so and carries no value."""
    ranks = Ranks(world, torch.device("cpu"), backend=os.environ.get("PP_BENCH_BACKEND", "gloo"))
    try:
        for _ in range(args.warmup):
            time.sleep(0.001)
        dt, per_rank, balance = ranks.timed(lambda: time.sleep(0.002 * args.steps * (1 + rank)), lambda: None)
        n = ranks.count()
        if rank == 0:
            print(json.dumps({"metric": "Mvoxels/s per demons iter, 512x512x256 fp32", "value": None, "stub": True, "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "ranks": balance,
                              "rccl_ranks": n, "data": "none: PP_BENCH_STUB=1, the kernels did not run"}), flush=True)
    finally:
        ranks.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 256], metavar=("NX", "NY", "NZ"))
    ap.add_argument("--variant", choices=["auto", "fused", "staged"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-registration", action="store_true")
    ap.add_argument("--no-atlas", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="time the region without the per-launch HIP events (no roofline block)")
    ap.add_argument("--kernel-events-every", type=int, default=5, metavar="K",
                    help="bracket every K-th launch of each kernel with HIP events inside the timed region (a pair of events costs the "
                         "stream ~7 us, 1.4 %% of a 0.5 ms kernel: with K = 5 the timed region carries a fifth of that and `avg_launch_ms` "
                         "is the mean of steps * repeats / K bracketed launches per kernel); 1 brackets every launch")
    ap.add_argument("--repeats", type=int, default=5, help="the timed block of --steps iterations is run this many times; `value` and "
                    "`ms_per_step` are the MEDIAN block's, min / max are reported beside it (boxes differ by several per cent run to run)")
    ap.add_argument("--pmc-calibration", action="store_true",
                    help="first run two kernels with known byte counts (16 B/lane and 4 B/lane), so that a rocprofv3 --pmc pass of this "
                         "command can calibrate FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md prescribes (tools/gpu_pmc2.sh)")
    args = ap.parse_args()

    # `python bench.py --gpus N` started plainly -- the way the driver starts N = 1 -- launches itself: one process per GPU
    # under torch.distributed.run on this node (the driver's own N > 1 command sets WORLD_SIZE and comes straight through).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it plainly, or under torch.distributed.run with --nproc-per-node {args.gpus})")
    if os.environ.get("PP_BENCH_STUB") == "1":
        return stub_main(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    ranks = Ranks(world, device)

    nx, ny, nz = args.size
    shape, spacing = (nz, ny, nx), (1.0, 1.0, 1.0)
    nvox = nx * ny * nz
    stream = torch.cuda.current_stream().cuda_stream
    ctx = _lib.Context(local_rank, stream)
    fixed, moving, geom = synth_pair(ctx, shape, spacing, 1234 + 100 * rank, device)
    field = torch.zeros((3,) + shape, device=device)
    if args.pmc_calibration:
        a = torch.rand(3 * nvox, device=device)
        b = torch.rand(3 * nvox, device=device) + 1.0
        c = torch.empty_like(a)
        torch.cuda.synchronize()
        torch.add(a, 1.0, out=c)            # 16 B/lane: reads 12 * nvox bytes, writes 12 * nvox
        ctx.fuse_divide(a, b, c, 3 * nvox)  # 16 B/lane (k_map4): reads 24 * nvox bytes, writes 12 * nvox
        ctx.sum_sq_diff(a, b, 3 * nvox)     # 4 B/lane loads: reads 24 * nvox bytes
        torch.cuda.synchronize()
        del a, b, c

    p = ctx.default_demons_params()
    p.smooth_update = 1
    p.smooth_displacement = 1
    p.sigma_d_vox[:] = [1.5 / s for s in spacing]  # deformable.py:253-257
    p.max_rms_error = 0.0                          # fixed iteration count for timing
    p.variant = {"auto": _lib.DEMONS_AUTO, "fused": _lib.DEMONS_FUSED, "staged": _lib.DEMONS_STAGED}[args.variant]

    def max_over_ranks(x):
        return ranks.max(x)

    if args.warmup > 0:
        p.iterations = args.warmup
        ctx.demons_execute(fixed, moving, geom, p, field, want_stats=False)
    torch.cuda.synchronize()
    ctx.profile_enable(not args.no_kernel_events, every=max(1, args.kernel_events_every))
    p.iterations = args.steps
    # The timed region -- exactly --steps iterations between barrier + synchronize on both sides, maximum over ranks -- is
    # run --repeats times; the MEDIAN block is the one reported (per-kernel HIP events accumulate over all of them).
    blocks = []
    for _ in range(max(1, args.repeats)):
        blocks.append(ranks.timed(lambda: ctx.demons_execute(fixed, moving, geom, p, field, want_stats=False), torch.cuda.synchronize))
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
    dt, per_rank, balance = blocks[order[len(order) // 2]]
    repeats = {"n": len(blocks), "ms_per_step_min": blocks[order[0]][0] * 1e3 / args.steps,
               "ms_per_step_median": dt * 1e3 / args.steps, "ms_per_step_max": blocks[order[-1]][0] * 1e3 / args.steps,
               "reported": "median"}
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    # What an EMPTY bracket reads on this stream: kernels[*].avg_ms are event-to-event times, so each carries this much beyond
    # the kernel's own duration (which is what rocprofv3 reports) -- that is why they can add up to more than ms_per_step.
    pair_ms = []
    for _ in range(64):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        eb.record()
        eb.synchronize()
        pair_ms.append(ea.elapsed_time(eb))
    event_pair_ms = sorted(pair_ms)[len(pair_ms) // 2]

    out = None
    if rank == 0:
        kernels = {}
        for name, (launches, total_ms) in prof.items():
            if launches == 0:
                continue
            avg_ms = total_ms / launches
            cb, kb = COMPULSORY_BYTES.get(name), CONTRACT_BYTES.get(name)
            kernels[name] = {"launches": launches, "avg_ms": avg_ms,
                             "compulsory_bytes_per_voxel": cb,
                             "compulsory_GBps": (cb * nvox / (avg_ms * 1e-3) / 1e9) if cb else None,
                             "frac_of_peak": (cb * nvox / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if cb else None,
                             "contract_bytes_per_voxel": kb,
                             "contract_GBps": (kb * nvox / (avg_ms * 1e-3) / 1e9) if kb else None}
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"]) if kernels else None
        # Measured HBM-side traffic of the same kernels: rocprofv3 --pmc passes of THIS command (tools/gpu_pmc2.sh ->
        # profiles/round2_pmc.json, which names the commit and the raw counter file it was reduced from; FETCH_SIZE
        # calibrated as MI355X_MICROARCH.md prescribes).  Counters cannot be read inside an untraced run, so the figure
        # is attached only when the committed capture is of this size and these kernels.
        pmc = load_pmc(nx, ny, nz)
        if pmc:
            for name, k in kernels.items():
                tb = pmc.get("hbm_bytes_per_launch", {}).get(name)
                if tb:
                    k["traffic_bytes_per_voxel"] = tb / nvox
                    k["traffic_GBps"] = tb / (k["avg_ms"] * 1e-3) / 1e9
        roofline = None
        if dom and kernels[dom]["compulsory_GBps"]:
            a = kernels[dom]["compulsory_GBps"]
            traffic = pmc.get("hbm_bytes_per_launch", {}).get(dom) if pmc else None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": a / HBM_PEAK_GBS, "traffic": traffic,
                        "traffic_source": ({"file": pmc.get("_file"), "commit": pmc.get("commit"), "raw": pmc.get("raw"),
                                            "kernel_source_sha16": pmc.get("kernel_source_sha16")} if traffic else
                                           {"note": "no committed PMC capture of these kernel sources (sha %s)" % kernel_source_sha16()}),
                        "bytes_model": "compulsory (inputs once + outputs once of the schedule that runs)",
                        "algorithmic_bytes_per_launch": kernels[dom]["compulsory_bytes_per_voxel"] * nvox,
                        "avg_launch_ms": kernels[dom]["avg_ms"],
                        "contract_model": {"bytes_per_voxel": kernels[dom]["contract_bytes_per_voxel"],
                                           "GBps": kernels[dom]["contract_GBps"],
                                           "note": "SURVEY 8(d) counts separable passes the fused kernel does not perform; "
                                                   "not a bandwidth"}}
        ms_per_step = dt * 1e3 / args.steps
        fused_run = any(k.startswith("k_fused") for k in kernels)
        iter_bytes = COMPULSORY_BYTES_ITER_FUSED if fused_run else CONTRACT_BYTES_ITER
        iter_gbps = iter_bytes * nvox / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "Mvoxels/s per demons iter, 512x512x256 fp32",
            "value": world * nvox * args.steps / dt / 1e6,
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"config 2 finest level: one fast-symmetric-forces demons iteration on a "
                                   f"{nx}x{ny}x{nz} fp32 CT-like pair per GPU, sigma_u 1.0 vox, sigma_d 1.5 mm, "
                                   f"schedule {args.variant}", "parallelism": f"1 atlas-to-target registration per GPU x{world}"},
            "ranks": balance,
            "repeats": repeats,
            # HIP events bracket every K-th launch of each kernel inside the timed region (kernels[*].launches = bracketed launches)
            "kernel_events_every": (None if args.no_kernel_events else max(1, args.kernel_events_every)),
            "kernel_event_pair_ms": event_pair_ms,
            "kernel_events_note": "kernels[*].avg_ms (and roofline.avg_launch_ms) are HIP event-to-event times on the launch stream: each "
                                  "includes the bracket's own cost (kernel_event_pair_ms: an empty bracket on this stream) and the lost "
                                  "back-to-back overlap, so they are upper bounds of the kernel durations rocprofv3 reports and may sum to "
                                  "more than ms_per_step; roofline.frac is therefore conservative",
            "roofline": roofline,
            "roofline_iteration": {"achieved": iter_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": iter_gbps / HBM_PEAK_GBS,
                                   "compulsory_bytes_per_voxel": iter_bytes,
                                   "contract_bytes_per_voxel": CONTRACT_BYTES_ITER,
                                   "contract_model_GBps": CONTRACT_BYTES_ITER * nvox / (ms_per_step * 1e-3) / 1e9},
            "kernels": kernels,
        }

    # one whole config-2 registration (3 levels, 10/10/10) through the drop-in API, for the record
    if not args.no_registration and world == 1 and (nx, ny, nz) == (512, 512, 256):
        try:
            from platipy_amd.image import Image
            from platipy_amd.registration.deformable import fast_symmetric_forces_demons_registration as reg

            fi, mi = Image(fixed, spacing), Image(moving, spacing)
            reg(fi, mi)  # warm-up: workspace allocation
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reg(fi, mi)
            torch.cuda.synchronize()
            out["registration_s"] = time.perf_counter() - t0
        except Exception as e:  # keep the headline line even if the optional leg fails
            out["registration_s"] = f"failed: {e!r}"

    if not args.no_atlas and (nx, ny, nz) == (512, 512, 256):
        # The multi-atlas legs are extras around the headline metric.  At N > 1 a rank that fails before a collective
        # would leave the others waiting: a watchdog prints the line without the unfinished leg and ends the process.
        watchdog = None
        if world > 1:
            import threading

            def bail():
                if rank == 0:
                    out.setdefault("multi_atlas", "timed out")
                    out.setdefault("multi_atlas_streams", "timed out")
                    print(json.dumps(out), flush=True)
                os._exit(0)

            watchdog = threading.Timer(float(os.environ.get("PP_BENCH_ATLAS_TIMEOUT", "240")), bail)
            watchdog.daemon = True
            watchdog.start()
        try:
            dt_a, nvox_label, dice_a = multi_atlas_leg(ctx, fixed, moving, spacing, rank, world, device)
            rs = __import__("platipy_amd").projects.multiatlas.run_segmentation
            xms = exchange_ms_over_ranks(ranks, getattr(rs, "last_exchange_ms", {}))
            if world > 1:
                dt_a = max_over_ranks(dt_a)
            if rank == 0:
                out["multi_atlas"] = {"atlases": world, "atlases_per_gpu": 1, "structures": 1, "seconds": dt_a,
                                      "atlases_per_min": 60.0 * world / dt_a, "fused_label_voxels": nvox_label,
                                      "dice_vs_template_label": dice_a,
                                      "fusion_allreduce_bytes": getattr(rs, "last_fusion_payload_bytes", None),
                                      # where an N > 1 run's time goes besides the chains: every exchange of the pipeline,
                                      # milliseconds, maximum over ranks (0 at N = 1: nothing is exchanged)
                                      "fusion_allreduce_ms": xms.get("fusion_allreduce", 0.0), "crop_allreduce_ms": xms.get("crop_allreduce", 0.0),
                                      "iar_exchange_ms": xms.get("iar_exchange", 0.0), "exchange_ms": xms,
                                      "settings": "multiatlas/run.py defaults (affine GD-line-search 16/8/4 x50; demons isotropic "
                                                  "6/3/1.5 mm x150/125/100; local vote), atlases resident in HBM; linear stage with "
                                                  "ITK's sampling semantics (round 6's default: seeded jitter, filtered gradient image)"}
                if world == 1:     # the same leg as rounds 1-5 timed it: samples on the lattice, the interpolant's gradient
                    dt_l, _, dice_l = multi_atlas_leg(ctx, fixed, moving, spacing, rank, world, device, linear_overrides={"itk_sampling": False})
                    out["multi_atlas"].update({"seconds_lattice_sampling": dt_l, "atlases_per_min_lattice_sampling": 60.0 / dt_l,
                                               "dice_lattice_sampling": dice_l})
        except Exception as e:
            if rank == 0:
                out["multi_atlas"] = f"failed: {e!r}"
        try:
            dt_s, dice_s, removed = multi_atlas_streams_leg(ctx, (nz, ny, nx), spacing, device, rank, world)
            xms = exchange_ms_over_ranks(ranks, getattr(__import__("platipy_amd").projects.multiatlas.run_segmentation, "last_exchange_ms", {}))
            if world > 1:
                dt_s = max_over_ranks(dt_s)
            if rank == 0:
                out["multi_atlas_streams"] = {"atlases": 4 * world, "atlases_per_gpu": 4, "hip_streams": 4, "seconds": dt_s,
                                              "fusion_allreduce_ms": xms.get("fusion_allreduce", 0.0), "iar_exchange_ms": xms.get("iar_exchange", 0.0),
                                              "exchange_ms": xms,
                                              "atlases_per_min": 60.0 * 4 * world / dt_s, "dice_vs_template_label": dice_s,
                                              "iterative_atlas_removal": ("on, removed %s" % removed) if 4 * world >= 8 else "off (< 8 atlases)",
                                              "settings": "as multi_atlas; 4 independent atlas warps per GPU, chains overlapped on 4 HIP "
                                                          "streams (config 5's shape)"}
                if world == 1:
                    dt_l, _, _ = multi_atlas_streams_leg(ctx, (nz, ny, nx), spacing, device, rank, world, linear_overrides={"itk_sampling": False})
                    out["multi_atlas_streams"].update({"seconds_lattice_sampling": dt_l, "atlases_per_min_lattice_sampling": 60.0 * 4 / dt_l})
        except Exception as e:
            if rank == 0:
                out["multi_atlas_streams"] = f"failed: {e!r}"
        if watchdog is not None:
            watchdog.cancel()

    if rank == 0:
        out["rccl_ranks"] = ranks.count()     # ranks that answered an all_reduce of ones (1 without a process group)
    else:
        ranks.count()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fixed, moving, spacing)
        print(json.dumps(out))
    ranks.close()


if __name__ == "__main__":
    main()
