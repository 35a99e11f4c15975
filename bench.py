#!/usr/bin/env python3
"""Headline benchmark: calibration sample-steps/s of the OmniQuant LWC/LET inner loop on MI355X.

A "step" is one calibration sample-step (quantize/omniquant.py:214-230 of the reference): LET re-parameterisation
+ weight fake-quant of the block's 7 linears, quantised block forward on one [1, 2048, H] sample, MSE against the
teacher output, backward, grad-norm and AdamW -- all on the HIP path, replayed from one hipGraph.  Inputs
(weights, the 128x2048 activation banks) are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--config llama-7b-w4a4|llama-7b-w3a16g128|...]

Two timed regions, both bracketed by barrier + synchronize and reduced with MAX over ranks:
  1. the contract's region: W warm-up + exactly K sample-steps per rank (`value`, `ms_per_step`).  There is no
     collective inside the step loop of the layer-sharded engine, so every rank steps its own block;
  2. `end_to_end`: the REAL sharded engine (omniquant_amd.parallel.calibrate_sharded over world x blocks-per-rank
     layers): streamed teacher pre-pass with RCCL send/recv of the boundary activation bank, per-layer teacher pass,
     hipGraph capture, the sample-steps, fold, propagate and the final parameter gather.
N>1: launched by torch.distributed.run, one rank per GPU.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (arch, wbits, abits, group, lwc, let, let_lr, alpha, aug_loss)
    "llama-7b-w4a4": ("llama-7b", 4, 4, None, True, True, 5e-3, 0.5, False),
    "llama-7b-w3a16g128": ("llama-7b", 3, 16, 128, True, False, 5e-3, 0.5, False),
    "llama-2-13b-w4a4": ("llama-2-13b", 4, 4, None, True, True, 1e-3, 0.75, False),
    "llama-2-70b-w2a16g64": ("llama-2-70b", 2, 16, 64, True, False, 5e-3, 0.5, False),
    "opt-125m-w4a16": ("opt-125m", 4, 16, None, True, False, 5e-3, 0.5, False),
}
WORKLOAD = {
    "llama-7b-w4a4": "LLaMA-7B W4A4 --lwc --let, 128x2048 calib, one decoder block per GPU (BASELINE configs[2])",
    "llama-7b-w3a16g128": "LLaMA-7B W3A16g128 --lwc, 128x2048 calib, one decoder block per GPU (BASELINE configs[1])",
    "llama-2-13b-w4a4": "LLaMA-2-13B W4A4 --lwc --let, 128x2048 calib, one decoder block per GPU",
    "llama-2-70b-w2a16g64": "LLaMA-2-70B W2A16g64 --lwc, 128x2048 calib, one decoder block per GPU",
    "opt-125m-w4a16": "OPT-125m W4A16 --lwc, 16x2048 calib",
}
SEQLEN = 2048
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def linear_flops(cfg, T):
    """fprop + dgrad + wgrad of the block's linears (2.49 TFLOP for LLaMA-7B)."""
    H = cfg.hidden_size
    hd = H // cfg.num_attention_heads
    kv = hd * getattr(cfg, "num_key_value_heads", cfg.num_attention_heads)
    if cfg.family == "llama":
        P = 2 * H * H + 2 * H * kv + 3 * H * cfg.intermediate_size
    else:
        P = 4 * H * H + 2 * H * cfg.ffn_dim
    return 3 * 2 * T * P, P


def build_block(name, rank, dev, nsamples, seed=0):
    from omniquant_amd.calibrate import default_args, decoder_layer_class, register_let_parameters, forward_bank
    from omniquant_amd.optim import BlockOptimizer
    from omniquant_amd.calibrate import StepRunner
    from omniquant_amd import synthetic as S
    arch, wbits, abits, group, lwc, let, let_lr, alpha, aug = CONFIGS[name]
    cfg = S.make_config(arch)
    args = default_args(wbits=wbits, abits=abits, group_size=group, lwc=lwc, let=let, let_lr=let_lr, alpha=alpha,
                        aug_loss=aug, net=arch, nsamples=nsamples)
    torch.manual_seed(seed + rank)
    layer = S.make_layer(cfg, seed=seed + rank, device=dev)
    q = decoder_layer_class(cfg.family)(cfg, layer, args).to(dev)
    q.compute_dtype = torch.bfloat16
    H = cfg.hidden_size
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    ch = torch.exp(0.5 * torch.randn(H, device=dev, generator=g))
    quant_inps = (torch.randn(nsamples, SEQLEN, H, device=dev, generator=g) * ch).to(torch.bfloat16)
    fp_inps = quant_inps.clone()
    mask = S.causal_mask(SEQLEN, dev)
    pos = torch.arange(SEQLEN, device=dev)[None]
    is_llama = cfg.family == "llama"
    q.set_quant_state(False, False)
    forward_bank(q, fp_inps, fp_inps, mask, pos, is_llama, chunk=4)          # teacher targets
    q.set_quant_state(False, True)
    q.let = let
    if let:
        sc, sh = S.synth_act_stats(cfg, 1)
        prefix = "model.layers" if is_llama else "model.decoder.layers"
        sc = {k.replace(f"{prefix}.0.", f"{prefix}.{0}."): v for k, v in sc.items()}
        register_let_parameters(q, cfg.family, sc, sh, alpha, 0, dev)
    opt = BlockOptimizer(q, args.let_lr, args.lwc_lr, args.wd)
    runner = StepRunner(q, opt, mask, pos, (1, SEQLEN, H), torch.bfloat16, aug, is_llama, use_graph=True)
    return cfg, args, q, opt, runner, quant_inps, fp_inps


def gemm_roofline(runner, quant_inps, fp_inps, cfg, n_prof=3):
    """Per-launch duration of the MFMA GEMM kernel, measured with HIP events on the launch stream while the real
    sample-step runs eagerly (same data, same cache state as the timed loop)."""
    from omniquant_amd import ops
    import inspect
    rec = []
    orig = ops.gemm
    sig = inspect.signature(orig)

    def timed(a, b, c, M, N, K, *rest, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        orig(a, b, c, M, N, K, *rest, **kw)
        e1.record(torch.cuda.current_stream())
        ba = sig.bind(a, b, c, M, N, K, *rest, **kw)
        ba.apply_defaults()
        g = ba.arguments
        nb = g["batch_o"] * g["batch_i"]
        rec.append((e0, e1, 2.0 * M * N * K * nb, (M, N, K, int(g["a_kc"]), int(g["b_kc"]), nb, int(g["tri"]))))

    runner.use_graph = False
    ops.gemm = timed
    try:
        for j in range(n_prof):
            runner.run(quant_inps[j:j + 1], fp_inps[j:j + 1])
        torch.cuda.synchronize()
    finally:
        ops.gemm = orig
        runner.use_graph = True
    H = cfg.hidden_size
    lin = [(e0.elapsed_time(e1) * 1e-3, fl) for e0, e1, fl, sh in rec
           if sh[5] == 1 and (min(sh[:3]) >= 1024 or (sh[0] * sh[1] * sh[2] >= 2048 * H * H))]
    if os.environ.get("OQ_BENCH_SHAPES"):
        import collections
        by = collections.OrderedDict()
        for e0, e1, fl, sh in rec:
            a = by.setdefault(sh, [0, 0.0, 0.0])
            a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += fl
        for sh, (n, t, fl) in by.items():
            tri = {0: 1.0, 1: 0.5, 2: 0.5, 3: 0.5}.get(sh[6], 1.0)
            print(f"[gemm] M={sh[0]:6d} N={sh[1]:6d} K={sh[2]:6d} a_kc={sh[3]} b_kc={sh[4]} batch={sh[5]:3d} tri={sh[6]} "
                  f"n/step={n // n_prof:2d} avg={1e6 * t / n:8.1f} us  {fl * tri / t / 1e12:7.1f} TF/s", file=sys.stderr)
    tot_t = sum(t for t, _ in lin)
    tot_f = sum(f for _, f in lin)
    n = len(lin)
    all_t = sum(e0.elapsed_time(e1) * 1e-3 for e0, e1, _, _ in rec)
    return dict(launches_per_step=n // n_prof, avg_launch_ms=1e3 * tot_t / max(n, 1), tflops=tot_f / tot_t / 1e12,
                gemm_ms_per_step=1e3 * all_t / n_prof, linear_gemm_ms_per_step=1e3 * tot_t / n_prof)


PROFILE_COUNTERS = "r2_gemm_counters.json"
ROOFLINE_KERNEL = "gemm_bf16_p3_kernel"


def pmc_counters(name):
    """rocprofv3 PMC results of the dominant kernel from the committed passes (profiles/README.md): HBM-side bytes per
    launch (FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950) and the MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES)).  Hardware counters
    cannot be read from inside this process, so the numbers come from profiles/ -- and ONLY when that file was
    collected for this config and for the kernel the run reports; otherwise both are null."""
    path = os.path.join(ROOT, "profiles", PROFILE_COUNTERS)
    try:
        with open(path) as f:
            ent = json.load(f).get(name)
    except (OSError, ValueError):
        return None, None
    if not ent or ent.get("kernel") != ROOFLINE_KERNEL:
        return None, None
    return ent.get("bytes_per_launch"), ent.get("mfma_busy_frac")


def cpu_baseline(name, n_steps=3, n_warm=1):
    """CPU baseline: the oracle (pure-PyTorch fp32 restatement of the reference loop, pinned to the reference by the
    golden fixtures) timed on this box's host cores on a bounded sample of the SAME workload.  Baseline only."""
    from oracle import ref_cpu as R
    from omniquant_amd import synthetic as S
    arch, wbits, abits, group, lwc, let, let_lr, alpha, aug = CONFIGS[name]
    # the GPU box gives one GPU's job a 16-core CPU share (more threads only oversubscribe the shared host)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    cfg = S.make_config(arch)
    layer = S.make_layer(cfg, seed=0, device="cpu")
    weights = {n: p.detach().float() for n, p in layer.named_parameters()}
    cdict = dict(hidden_size=cfg.hidden_size, num_attention_heads=cfg.num_attention_heads,
                 num_key_value_heads=getattr(cfg, "num_key_value_heads", cfg.num_attention_heads), rms_norm_eps=1e-6)
    blk = R.Block(cfg.family, cdict, weights, R.QuantSpec(wbits, abits, group, lwc, let), max_pos=SEQLEN)
    if let:
        sc, sh = S.synth_act_stats(cfg, 1)
        blk.register_let(sc, sh, alpha, 0, "model.layers" if cfg.family == "llama" else "model.decoder.layers")
    opt = R.AdamW([{"params": blk.let_params(), "lr": let_lr}, {"params": blk.lwc_params(), "lr": 1e-2}])
    n_all = n_steps + n_warm
    x = S.make_calib_inputs(n_all, SEQLEN, cfg.hidden_size, dtype=torch.float32)
    mask = S.causal_mask(SEQLEN)
    pos = torch.arange(SEQLEN)[None]
    with torch.no_grad():
        tgt = torch.stack([blk.forward(x[j:j + 1], mask, pos, None, False)[0] for j in range(n_all)])
    for j in range(n_warm):
        R.train_step(blk, opt, x[j:j + 1], tgt[j:j + 1], mask, pos)
    t0 = time.time()
    for j in range(n_warm, n_all):
        R.train_step(blk, opt, x[j:j + 1], tgt[j:j + 1], mask, pos)
    dt = time.time() - t0
    return dict(value=n_steps / dt, unit="sample-steps/s", cores=cores, kind="port",
                sample=f"{n_warm} warm-up + {n_steps} timed sample-steps of one {arch} block (T={SEQLEN}, fp32 oracle, "
                       f"{cores} threads, {dt / n_steps:.1f} s per step)")


def end_to_end(name, rank, world, dev, dist, n_samples, blocks_per_rank, chunk):
    """Second timed region: the real layer-sharded engine on world x blocks_per_rank synthetic layers.  Returns the
    wall time (MAX over ranks) of calibrate_sharded -- teacher pre-pass pipeline, calibration of every owned block
    (1 epoch x n_samples sample-steps each, hipGraph-replayed), fold, propagate, gather of the learned parameters."""
    from omniquant_amd import synthetic as S
    from omniquant_amd.calibrate import default_args
    from omniquant_amd.parallel import calibrate_sharded, hip_callables, shard_bounds
    arch, wbits, abits, group, lwc, let, let_lr, alpha, aug = CONFIGS[name]
    cfg = S.make_config(arch)
    L = world * blocks_per_rank
    args = default_args(wbits=wbits, abits=abits, group_size=group, lwc=lwc, let=let, let_lr=let_lr, alpha=alpha,
                        aug_loss=aug, net=arch, nsamples=n_samples, epochs=1)
    lo, hi = shard_bounds(L, world, rank)
    layers = [S.make_layer(cfg, seed=100 + i, device=dev) if lo <= i < hi else None for i in range(L)]
    H = cfg.hidden_size
    if rank == 0:
        g = torch.Generator(device=dev).manual_seed(11)
        ch = torch.exp(0.5 * torch.randn(H, device=dev, generator=g))
        inps = (torch.randn(n_samples, SEQLEN, H, device=dev, generator=g) * ch).to(torch.bfloat16)
    else:
        inps = torch.empty(n_samples, SEQLEN, H, device=dev, dtype=torch.bfloat16)
    mask = S.causal_mask(SEQLEN, dev)
    pos = torch.arange(SEQLEN, device=dev)[None]
    sc, sh = S.synth_act_stats(cfg, L) if let else (None, None)
    teacher, calib = hip_callables(layers, cfg, args, mask, pos, sc, sh, None, torch.bfloat16, True)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    if dist is not None:
        merged, _ = calibrate_sharded(L, inps, teacher, calib, chunk=chunk)
    else:
        merged = calib(0, L, inps, inps.clone())
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    if rank == 0:
        assert merged is not None and sorted(merged.keys()) == list(range(L)), "gather lost a layer"
    steps = L * n_samples
    return dict(value=steps / dt, unit="sample-steps/s", wall_s=dt, layers=L, blocks_per_rank=blocks_per_rank,
                calib_samples=n_samples, epochs=1, boundary_message_samples=chunk,
                includes="streamed teacher pre-pass (RCCL send/recv of the boundary bank between ranks), per-layer teacher "
                         "pass, hipGraph capture, sample-steps, fold, propagate, final parameter gather")


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n, argv, port=None, python=None):
    """The `torch.distributed.run` command line of an N-rank run of this script on one node (rendezvous on 127.0.0.1)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    """Runs launch_command() as a child process group, passes its stdout / stderr through and returns its exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this image
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(launch_command(n, argv), env=env, cwd=ROOT)
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~1 s of warm-up and ~4 s timed at the default config, so clocks and the driver's GPU-busy sampling see
    # the steady state (128 steps were 0.5 s)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--config", default="llama-7b-w4a4", choices=list(CONFIGS))
    ap.add_argument("--nsamples", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--e2e-samples", type=int, default=16, help="calibration samples of the end-to-end region")
    ap.add_argument("--blocks-per-rank", type=int, default=1, help="decoder blocks per rank in the end-to-end region")
    ap.add_argument("--boundary-chunk", type=int, default=4, help="samples per boundary message of the teacher pre-pass")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start one fresh process per GPU and relay rank 0's JSON line.  Nothing in this
        # process has touched the GPU (no HIP call, no torch.cuda query), and the children are started, never exec'd into.
        sys.exit(self_launch(a.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {a.gpus}` "
                         f"(it launches its own ranks) or under torch.distributed.run with --nproc-per-node {a.gpus}")
    dist = None
    if os.environ.get("OQ_BENCH_LAUNCH_ONLY"):      # tests/test_host_logic_cpu.py: rendezvous of the ranks only (gloo, no GPU)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = [None] * world
        dist.all_gather_object(seen, (rank, local_rank))
        if rank == 0:
            print(json.dumps({"launch_only": True, "n_gpus": a.gpus, "world": world, "ranks": seen}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    from omniquant_amd import _capi
    _capi.load()      # raises if the HIP extension is missing: no fallback

    nsamples = min(a.nsamples, 16) if a.config.startswith("opt") else a.nsamples
    cfg, args, q, opt, runner, quant_inps, fp_inps = build_block(a.config, rank, dev, nsamples)
    n = quant_inps.shape[0]

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for j in range(a.warmup):
        runner.run(quant_inps[j % n:j % n + 1], fp_inps[j % n:j % n + 1])
    sync()
    t0 = time.perf_counter()
    for j in range(a.steps):
        runner.run(quant_inps[j % n:j % n + 1], fp_inps[j % n:j % n + 1])
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    loss = float(runner.loss)
    assert loss == loss, "NaN loss in the timed region"

    roof = gemm_roofline(runner, quant_inps, fp_inps, cfg)
    e2e = None
    if not a.no_end_to_end and not a.config.startswith("opt"):
        del runner, opt, q
        torch.cuda.empty_cache()
        e2e = end_to_end(a.config, rank, world, dev, dist, a.e2e_samples, a.blocks_per_rank, a.boundary_chunk)
    if rank == 0:
        flops, P = linear_flops(cfg, SEQLEN)
        value = world * a.steps / dt
        traffic, mfma_busy = pmc_counters(a.config)
        out = {
            "metric": "calibration_sample_steps_per_sec", "value": value, "unit": "sample-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": WORKLOAD[a.config], "name": a.config, "seq_len": SEQLEN, "batch_size": 1,
                       "calib_samples": n, "blocks_per_gpu": 1, "sharding": f"layers x{world}",
                       "n_ranks_seen": int(dist.get_world_size()) if dist is not None else 1,
                       "per_gpu_sample_steps_per_sec": value / world, "last_loss": loss,
                       "linear_tflop_per_step": flops / 1e12},
            "roofline": {"bound": "mfma", "kernel": ROOFLINE_KERNEL + " (fprop+dgrad+wgrad of the fake-quant linears)",
                         "achieved": roof["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": roof["tflops"] / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "mfma_busy_frac": mfma_busy,
                         "avg_launch_ms": roof["avg_launch_ms"], "launches_per_step": roof["launches_per_step"],
                         "linear_gemm_ms_per_step": roof["linear_gemm_ms_per_step"],
                         "all_gemm_ms_per_step": roof["gemm_ms_per_step"]},
        }
        if e2e is not None:
            out["end_to_end"] = e2e
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.config)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
