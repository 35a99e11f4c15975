"""world_size-2 gloo test (CPU) of the layer-sharded calibration protocol (omniquant_amd/parallel.py): partition,
pipelined teacher boundary send/recv, per-rank calibration and the final gather.  The block compute is supplied by
the CPU oracle here (tests may use it); on GPUs the same protocol drives the HIP engine."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def test_shard_bounds_partition():
    from omniquant_amd.parallel import shard_bounds
    for n in (1, 2, 7, 32, 40, 80):
        for w in (1, 2, 3, 4, 8):
            cover = []
            for r in range(w):
                lo, hi = shard_bounds(n, w, r)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))
            sizes = [shard_bounds(n, w, r)[1] - shard_bounds(n, w, r)[0] for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _worker(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from oracle import ref_cpu as R
    from omniquant_amd.parallel import calibrate_sharded
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    g, m = load_golden("g4_traj_llama_w3a16g32_lwc.npz")
    T = lambda a: torch.from_numpy(np.asarray(a)).float()
    layers = [{k[len(f"w{i}."):]: T(v) for k, v in g.items() if k.startswith(f"w{i}.")} for i in range(m["n_layers"])]
    spec = R.QuantSpec(m["wbits"], m["abits"], m["group_size"], m["lwc"], m["let"])
    inps, mask, pos = T(g["inps"]), T(g["mask"]), torch.from_numpy(g["position_ids"])

    def teacher(lo, hi, bank):
        for i in range(lo, hi):
            blk = R.Block(m["family"], m["config"], layers[i], spec)
            with torch.no_grad():
                for j in range(bank.shape[0]):
                    bank[j] = blk.forward(bank[j][None], mask, pos, None, False)[0]
        return bank

    def calib(lo, hi, tbank, sbank):
        res = R.calibrate(m["family"], m["config"], layers[lo:hi], spec, tbank, mask, pos, epochs=m["epochs"],
                          let_lr=m["let_lr"], lwc_lr=m["lwc_lr"])
        return {lo + i: {k: v.cpu() for k, v in d.items()} for i, d in enumerate(res["omni"])}

    merged, (lo, hi) = calibrate_sharded(m["n_layers"], inps, teacher, calib)
    if rank == 0:
        torch.save({"merged": merged, "bounds": (lo, hi)}, os.path.join(outdir, "r0.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_protocol_two_ranks_gloo():
    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_worker, args=(2, initfile, d), nprocs=2, join=True)
        out = torch.load(os.path.join(d, "r0.pt"), weights_only=False)
    merged = out["merged"]
    g, m = load_golden("g4_traj_llama_w3a16g32_lwc.npz")
    assert sorted(merged.keys()) == list(range(m["n_layers"]))
    # layer 0 is calibrated by rank 0 from the true input: identical to the sequential reference trajectory
    for k in [k for k in g if k.startswith("omni.0.")]:
        n = k[len("omni.0."):]
        ref = g[k].astype(np.float64)
        assert np.abs(merged[0][n].double().numpy() - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1e-6) + 1e-3
    # layer 1 is calibrated by rank 1 with the TEACHER activation as student input (documented deviation):
    # same keys/shapes/dtype, finite, and close to the sequential result
    for k in [k for k in g if k.startswith("omni.1.")]:
        n = k[len("omni.1."):]
        t = merged[1][n]
        assert t.dtype == torch.float16 and tuple(t.shape) == g[k].shape and torch.isfinite(t).all()
        assert np.abs(t.double().numpy() - g[k].astype(np.float64)).max() < 0.5
