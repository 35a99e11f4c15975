"""Two-rank run of the layer-sharded protocol with the REAL HIP engine: two processes share cuda:0 and talk over gloo
(RCCL cannot put two ranks on one device; on a multi-GPU node the same code runs with backend "nccl").  Rank 0
calibrates layer 0, rank 1 layer 1 with the teacher activation as student input (the documented deviation).  Checked
against a single-process run of exactly that schedule."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_golden
from gpu_helpers import T, make_args, make_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FIX = "g4_traj_llama_w3a16g32_lwc.npz"


def _layers(g, m, cfg):
    from omniquant_amd.synthetic import make_layer
    layers = []
    for i in range(m["n_layers"]):
        w = {k[len(f"w{i}."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(f"w{i}.")}
        layers.append(make_layer(cfg, weights=w, device=DEV))
    return layers


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    from omniquant_amd.parallel import calibrate_sharded, hip_callables
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, m = load_golden(FIX)
        cfg, args = make_cfg(m), make_args(m)
        layers = _layers(g, m, cfg)
        pos = torch.from_numpy(g["position_ids"]).to(DEV)
        teacher, calib = hip_callables(layers, cfg, args, T(g["mask"], DEV), pos, compute_dtype=torch.float32)
        merged, (lo, hi) = calibrate_sharded(m["n_layers"], T(g["inps"], DEV), teacher, calib)
        assert (lo, hi) == (rank, rank + 1)
        if rank == 0:
            torch.save(merged, out_path)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_calibration_on_one_gpu(tmp_path):
    from omniquant_amd.calibrate import calibrate_layers
    from omniquant_amd.parallel import hip_callables
    g, m = load_golden(FIX)
    assert m["n_layers"] == 2
    out_path = str(tmp_path / "merged.pt")
    mp.spawn(_worker, args=(2, 29541, out_path), nprocs=2, join=True)
    merged = torch.load(out_path, weights_only=True)
    assert sorted(merged.keys()) == [0, 1]
    # layer 0 is calibrated exactly as in the sequential engine: it must match the reference-generated trajectory
    for k in [k for k in g if k.startswith("omni.0.")]:
        ref = g[k].astype(np.float64)
        got = merged[0][k[len("omni.0."):]].double().numpy()
        assert np.abs(got - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1e-6) + 1e-3, k
    # layer 1: same schedule in one process (teacher bank of layer 1 as both teacher and student input)
    cfg, args = make_cfg(m), make_args(m)
    layers = _layers(g, m, cfg)
    pos = torch.from_numpy(g["position_ids"]).to(DEV)
    mask = T(g["mask"], DEV)
    teacher, _ = hip_callables(layers, cfg, args, mask, pos, compute_dtype=torch.float32)
    bank = teacher(0, 1, T(g["inps"], DEV).clone())
    _, omni, _, _ = calibrate_layers(layers[1:2], cfg, args, bank, mask, pos, None, None, None, True, torch.float32,
                                     layer_offset=1, student_inps=bank.clone())
    for n, t in omni[1].items():
        assert torch.equal(t, merged[1][n]), n          # the engine is bit-reproducible
