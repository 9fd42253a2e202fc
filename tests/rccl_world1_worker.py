"""Worker of tests/test_rccl_world1.py (run as a script in its own process): a torch.distributed process group with
backend "nccl" (RCCL on ROCm) and ONE rank on cuda:0, through which the multi-atlas exchanges (projects/multiatlas._Dist,
label/iar.run_iar_distributed) and bench.py's Ranks run exactly as they would on 8 GPUs -- device_id=, collectives on
device tensors, dtypes, the reduce-to-root path.  With one rank every collective is the identity, so the results must
equal the run without a process group bit for bit.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    import platipy_amd as pa
    from platipy_amd.label.iar import run_iar, run_iar_distributed
    from platipy_amd.projects import multiatlas
    from tests.test_multiatlas import _data, _iar_case, _settings

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ids = ["001", "002"]
    target, _, _, atlases = _data(pa, ids)
    st = _settings(ids)
    plain, plain_prob = multiatlas.run_segmentation(target, st, atlases=atlases)
    assert multiatlas.run_segmentation.last_world_size == 1
    t_iar, iar_ids, aset_w = _iar_case(pa)
    for i in iar_ids:
        aset_w[i]["DIR"]["Weight Map"] = pa.label.compute_weight_map(t_iar, aset_w[i]["DIR"]["CT Image"], vote_type="global")
    kept_plain = run_iar(atlas_set=aset_w, reference_structure="HEART", min_best_atlases=3, z_score_statistic="mad",
                         outlier_method="iqr", outlier_factor=1.5)
    _, _, aset = _iar_case(pa)

    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    try:
        d = multiatlas._Dist()
        assert d.dist is not None and d.world == 1 and dist.get_backend() == "nccl"
        # (i) run_segmentation with 2 atlases through _Dist: crop all_reduce, layout MIN, fusion all_reduce, then reduce
        res, prob = multiatlas.run_segmentation(target, st, atlases=atlases)
        ms = dict(multiatlas.run_segmentation.last_exchange_ms)
        assert {"crop_allreduce", "fusion_allreduce", "fusion_layout"} <= set(ms), ms
        for k in plain:
            assert np.array_equal(res[k].numpy(), plain[k].numpy()), k
            assert np.array_equal(prob[k].numpy(), plain_prob[k].numpy()), k
        res2, prob2 = multiatlas.run_segmentation(target, st, atlases=atlases, fusion_collective="reduce")
        ms2 = dict(multiatlas.run_segmentation.last_exchange_ms)
        assert "fusion_reduce" in ms2 and "fusion_allreduce" not in ms2, ms2
        for k in plain:
            assert np.array_equal(res2[k].numpy(), plain[k].numpy()), k
        # distributed atlas removal (the pipeline takes this path when world > 1): every exchange of it on device tensors
        # through the communicator, same atlases kept as the single-process run_iar
        weights = {i: float(pa.label.compute_weight_map(t_iar, aset[i]["DIR"]["CT Image"], vote_type="global").tensor.flatten()[0])
                   for i in iar_ids}
        kept = run_iar_distributed(multiatlas._Dist(), aset, iar_ids, iar_ids, "HEART", t_iar, weights, min_best_atlases=3,
                                   z_score_statistic="mad", outlier_method="iqr", outlier_factor=1.5)
        assert sorted(kept) == sorted(kept_plain), (kept, list(kept_plain))
        out["iar_kept"] = sorted(kept)
        out["exchange_ms"] = {k: round(float(v), 4) for k, v in ms.items()}
        out["exchange_ms_reduce"] = {k: round(float(v), 4) for k, v in ms2.items()}
        # (ii) bench.Ranks on the communicator that already exists: count / max / max_dict / gather / timed on device tensors
        import bench

        os.environ["WORLD_SIZE"] = "1"
        r = bench.Ranks.__new__(bench.Ranks)
        r.world, r.device, r.dist, r.rank = 1, dev, dist, 0
        assert r._cdev() == dev
        assert r.count() == 1 and r.max(2.5) == 2.5 and r.gather(1.25) == [1.25]
        assert r.max_dict({"a": 1.0, "b": 3.0}, ["a", "b", "c"]) == {"a": 1.0, "b": 3.0, "c": 0.0}
        x = torch.ones(1 << 20, device=dev)
        dt, per_rank, extra = r.timed(lambda: x.mul_(2.0), torch.cuda.synchronize)
        assert dt > 0 and len(per_rank) == 1 and extra["slowest_rank"] == 0
        out["rccl_mapped"] = any("librccl" in ln for ln in open("/proc/self/maps"))
        assert out["rccl_mapped"], "librccl is not among the mapped libraries"
        out["ok"] = True
    finally:
        dist.destroy_process_group()
    print("RCCL_WORLD1 " + json.dumps(out))


if __name__ == "__main__":
    main()
