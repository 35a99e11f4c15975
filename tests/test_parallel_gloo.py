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


def _pipe_worker(rank, world, initfile, outdir, chunk):
    sys.path.insert(0, ROOT)
    from omniquant_amd.parallel import pipeline_teacher_boundaries, shard_bounds
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    inps = torch.randn(7, 5, 8, generator=g)              # 7 samples: the last message is ragged for chunk 2 and 3
    calls = []

    def teacher(lo, hi, bank):                             # "layer i": x -> tanh(x) * (i + 2) + i, per sample
        calls.append(bank.shape[0])
        for i in range(lo, hi):
            bank = torch.tanh(bank) * (i + 2) + i
        return bank

    lo, hi = shard_bounds(5, world, rank)                  # 5 layers over 3 ranks: 2 + 2 + 1
    bank = pipeline_teacher_boundaries(inps, lo, hi, teacher, None, chunk)
    torch.save({"bank": bank, "calls": calls, "bounds": (lo, hi)}, os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("chunk", [None, 2, 3, 100])
def test_chunked_teacher_pipeline_three_ranks_gloo(chunk):
    """The streamed pre-pass hands every rank exactly the activations the sequential chain produces at its boundary,
    for any message size (one message, ragged last message, message larger than the bank), and a rank forwards one
    message at a time (so its successor can start after the first one)."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_pipe_worker, args=(3, os.path.join(d, "init"), d, chunk), nprocs=3, join=True)
        outs = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(3)]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(7, 5, 8, generator=g)
    expect = [x.clone()]
    for i in range(5):
        x = torch.tanh(x) * (i + 2) + i
        expect.append(x.clone())
    for r, o in enumerate(outs):
        lo, hi = o["bounds"]
        assert torch.equal(o["bank"], expect[lo]), f"rank {r}: wrong boundary bank"
    n_msg = 1 if (chunk is None or chunk >= 7) else -(-7 // chunk)
    assert len(outs[0]["calls"]) == n_msg and len(outs[1]["calls"]) == n_msg and outs[2]["calls"] == []
    assert sum(outs[0]["calls"]) == 7


def _dtype_worker(rank, world, initfile, outdir, tdt):
    sys.path.insert(0, ROOT)
    from omniquant_amd.parallel import pipeline_teacher_boundaries, shard_bounds
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(3)
    inps = torch.randn(5, 4, 8, generator=g).half()        # fp16 bank, as the reference's Catcher records it
    tdtype = getattr(torch, tdt)

    def teacher(lo, hi, bank):                             # computes (and returns) in ITS dtype, like hip_callables' teacher
        bank = bank.to(tdtype)
        for i in range(lo, hi):
            bank = bank * 1.5 + (i + 1)
        return bank

    lo, hi = shard_bounds(2, world, rank)
    bank = pipeline_teacher_boundaries(inps, lo, hi, teacher, None, 2)
    torch.save({"bank": bank}, os.path.join(outdir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tdt", ["bfloat16", "float32"])
def test_boundary_messages_travel_in_the_bank_dtype(tdt):
    """An fp16 bank with a teacher that computes in bf16 / f32: the message is converted to the bank's dtype before it is
    sent (the receive buffer has the bank's dtype and neither backend checks), so rank 1 gets VALUES, not reinterpreted bits."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_dtype_worker, args=(2, os.path.join(d, "init"), d, tdt), nprocs=2, join=True)
        got = torch.load(os.path.join(d, "r1.pt"), weights_only=True)["bank"]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 4, 8, generator=g).half()
    want = (x.to(getattr(torch, tdt)) * 1.5 + 1).half()
    assert got.dtype == torch.float16 and torch.equal(got, want)
