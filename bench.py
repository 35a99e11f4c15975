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
    # the reference's own recipe for this model (scripts/llama/llama-7b/w4a4.sh:4 adds --aug_loss: a second MSE term against the
    # teacher output of the STUDENT input, quantize/omniquant.py:168-171,221-222)
    "llama-7b-w4a4-aug": ("llama-7b", 4, 4, None, True, True, 5e-3, 0.5, True),
    "llama-7b-w3a16g128": ("llama-7b", 3, 16, 128, True, False, 5e-3, 0.5, False),
    "llama-2-13b-w4a4": ("llama-2-13b", 4, 4, None, True, True, 1e-3, 0.75, False),
    "llama-2-70b-w2a16g64": ("llama-2-70b", 2, 16, 64, True, False, 5e-3, 0.5, False),
    "opt-125m-w4a16": ("opt-125m", 4, 16, None, True, False, 5e-3, 0.5, False),
}
WORKLOAD = {
    "llama-7b-w4a4": "LLaMA-7B W4A4 --lwc --let, 128x2048 calib, one decoder block per GPU (BASELINE configs[2])",
    "llama-7b-w4a4-aug": "LLaMA-7B W4A4 --lwc --let --aug_loss (scripts/llama/llama-7b/w4a4.sh), 128x2048 calib, one decoder block per GPU",
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


_DT_SIZE = {0: 4, 1: 2, 2: 2}      # OQ_F32, OQ_F16, OQ_BF16 (include/oq_hip.h)


def _hbm_bytes(name, a):
    """Algorithmic HBM bytes of one call of a quantiser-class entry point (what the kernel must read and
    write once), from its C-ABI arguments.  None for entry points that are not priced."""
    import ctypes
    from omniquant_amd import _capi as C
    sz = _DT_SIZE
    if name == "oq_fakequant_fwd":
        return a[2] * a[3] * (sz[a[1]] + sz[a[14]] + (1 if a[20] else 0))
    if name == "oq_fakequant_bwd":
        return a[2] * a[3] * (sz[a[1]] + sz[a[16]] + (sz[a[21]] if a[20] else 0))
    if name == "oq_fakequant_fwd_multi":
        arr = (C.FakeQuantFwdArgs * a[1]).from_address(a[0])
        return sum(t.rows * t.cols * (sz[t.w_dtype] + sz[t.y_dtype] + (1 if t.codes else 0)) for t in arr)
    if name == "oq_fakequant_bwd_multi":
        arr = (C.FakeQuantBwdArgs * a[1]).from_address(a[0])
        return sum(t.rows * t.cols * (sz[t.w_dtype] + sz[t.g_dtype] + (sz[t.gx_dtype] if t.gx else 0)) for t in arr)
    if name == "oq_norm_quant_fwd":
        return a[2] * a[3] * (sz[a[1]] + sz[a[10]] + (1 if a[17] else 0))
    if name == "oq_norm_quant_bwd":
        n_g = 1 + (1 if a[2] else 0) + (1 if a[3] else 0) + (1 if a[19] else 0)
        return a[6] * a[7] * (sz[a[4]] + sz[a[5]] * (n_g + 1))
    if name == "oq_silu_mul_quant_fwd":
        return a[3] * a[4] * (2 * sz[a[2]] + sz[a[8]] + (1 if a[13] else 0))
    if name == "oq_silu_mul_quant_bwd":
        return a[5] * a[6] * (2 * sz[a[3]] + 3 * sz[a[4]])
    if name == "oq_qkv_rope_quant_fwd":
        return a[2] * (a[4] + a[5] + a[6]) * a[7] * (sz[a[1]] + sz[a[14]])
    if name == "oq_qkv_rope_quant_bwd":
        return a[2] * (a[4] + a[5] + a[6]) * a[7] * (sz[a[1]] + 2 * sz[a[16]])
    return None


KERNEL_OF = {"oq_fakequant_fwd_multi": "letq_fwd_multi_kernel (LET + LWC weight quantiser, several matrices)",
             "oq_fakequant_bwd_multi": "letq_bwd_multi_kernel (its backward: reduces dW to LWC / LET gradients)",
             "oq_fakequant_fwd": "rowq_fwd_kernel / letq_fwd_kernel (weight or per-token quantiser)",
             "oq_fakequant_bwd": "rowq_bwd_kernel / letq_bwd_kernel",
             "oq_norm_quant_fwd": "normq_fwd_kernel (RMSNorm -> per-token quantiser)", "oq_norm_quant_bwd": "normq_bwd_kernel",
             "oq_silu_mul_quant_fwd": "rowq_fwd_kernel<PRO=1> (silu * up -> per-token quantiser)",
             "oq_silu_mul_quant_bwd": "rowq_bwd_kernel<PRO=1>", "oq_qkv_rope_quant_fwd": "ropeq_fwd_kernel (RoPE -> head quantisers)",
             "oq_qkv_rope_quant_bwd": "ropeq_bwd_kernel"}


def kernel_rooflines(runner, quant_inps, fp_inps, cfg, n_prof=3):
    """Per-launch durations measured with HIP events on the launch stream (every C-ABI call is bracketed) while the real
    sample-step runs eagerly -- same data, same cache state as the timed loop.  Returns the MFMA GEMM figures (bf16 and int8
    launches of the fake-quant linears) and the HBM-class quantiser kernels' achieved bandwidth."""
    from omniquant_amd import _capi as C
    rec = []
    orig = C.call

    def timed(name, *args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        nbytes = _hbm_bytes(name, args)      # (the multi-matrix entry points pass a struct array that lives only during the call)
        e0.record(st)
        orig(name, *args)
        e1.record(st)
        rec.append((name, e0, e1, args, nbytes))

    # the eager step's host side is slower than its short kernels: without a head start the GPU idles between launches and an
    # event pair would time the host.  A queue of graph replays of the same step in front of every profiled step gives the host
    # its head start AND keeps the chip at the clocks of the timed loop (a spin kernel was tried first: the GEMMs behind 60 ms of
    # idle-power spinning measured 10 % slower than rocprofv3 sees them inside replays).
    def head_start():
        if runner.graph is not None:
            for _ in range(24):
                runner.graph.replay()
        else:
            torch.cuda._sleep(100_000_000)
    # what a pair of event records costs on the GPU timeline with NOTHING between them (two marker packets through the command
    # processor): measured here on the warm, busy GPU and subtracted from every interval (~5 us per launch)
    head_start()
    pairs = []
    for _ in range(64):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record(torch.cuda.current_stream())
        b_.record(torch.cuda.current_stream())
        pairs.append((a_, b_))
    torch.cuda.synchronize()
    gaps = sorted(a_.elapsed_time(b_) * 1e-3 for a_, b_ in pairs)
    ev_over = gaps[len(gaps) // 2]                     # median, seconds
    runner.use_graph = False
    C.call = timed
    try:
        for j in range(n_prof):
            head_start()
            runner.run(quant_inps[j:j + 1], fp_inps[j:j + 1], fp_inps[j:j + 1] if runner.t2 is not None else None)
        torch.cuda.synchronize()
    finally:
        C.call = orig
        runner.use_graph = True
    H = cfg.hidden_size
    lin, lin8, allg, hbm = [], [], 0.0, {}
    shapes = []
    for name, e0, e1, a, nbytes in rec:
        dt = max(e0.elapsed_time(e1) * 1e-3 - ev_over, 1e-7)
        if name in ("oq_gemm", "oq_gemm_ws"):
            M, N, K, nb, tri = a[5], a[6], a[7], a[16] * a[17], a[24]
            allg += dt
            shapes.append(((M, N, K, a[11], a[12], nb, tri, "bf16"), dt, 2.0 * M * N * K * nb))
            if nb == 1 and (min(M, N, K) >= 1024 or M * N * K >= 2048 * H * H):
                lin.append((dt, 2.0 * M * N * K))
        elif name == "oq_gemm_i8":
            M, N, K = a[13], a[14], a[15]
            allg += dt
            shapes.append(((M, N, K, 1, 1, 1, 0, "int8"), dt, 2.0 * M * N * K))
            lin8.append((dt, 2.0 * M * N * K))
        else:
            b = nbytes
            if b is not None:
                h = hbm.setdefault(name, [0, 0.0, 0.0])
                h[0] += 1; h[1] += dt; h[2] += float(b)
    if os.environ.get("OQ_BENCH_SHAPES"):
        import collections
        by = collections.OrderedDict()
        for sh, t, fl in shapes:
            acc = by.setdefault(sh, [0, 0.0, 0.0])
            acc[0] += 1; acc[1] += t; acc[2] += fl
        for sh, (n, t, fl) in by.items():
            tri = {0: 1.0, 1: 0.5, 2: 0.5, 3: 0.5}.get(sh[6], 1.0)
            print(f"[gemm] {sh[7]:4s} M={sh[0]:6d} N={sh[1]:6d} K={sh[2]:6d} a_kc={sh[3]} b_kc={sh[4]} batch={sh[5]:3d} tri={sh[6]} "
                  f"n/step={n // n_prof:2d} avg={1e6 * t / n:8.1f} us  {fl * tri / t / 1e12:7.1f} T/s", file=sys.stderr)
        for name, (n, t, b) in hbm.items():
            print(f"[hbm]  {name:26s} n/step={n // n_prof:2d} avg={1e6 * t / n:8.1f} us  {b / t / 1e9:8.1f} GB/s", file=sys.stderr)

    def agg(v):
        t, f = sum(x for x, _ in v), sum(x for _, x in v)
        return dict(launches_per_step=len(v) // n_prof, avg_launch_ms=1e3 * t / max(len(v), 1), tflops=(f / t / 1e12) if t else 0.0,
                    ms_per_step=1e3 * t / n_prof, tflop_per_step=f / n_prof / 1e12)

    out = dict(bf16=agg(lin), int8=agg(lin8) if lin8 else None, gemm_ms_per_step=1e3 * allg / n_prof, hbm=[],
               event_pair_overhead_us=1e6 * ev_over)
    for name, (n, t, b) in sorted(hbm.items(), key=lambda kv: -kv[1][1]):
        out["hbm"].append(dict(entry=name, kernel=KERNEL_OF.get(name, name), launches_per_step=n // n_prof, us_per_step=1e6 * t / n_prof,
                               avg_launch_us=1e6 * t / n, algorithmic_bytes_per_launch=b / n, achieved_gbs=b / t / 1e9,
                               frac=b / t / 1e9 / HBM_PEAK_GBS))
    return out


PROFILE_COUNTERS = "r3_gemm_counters.json"
ROOFLINE_KERNEL = "gemm_bf16_p3_kernel + gemm_bf16_w256_kernel"
MFMA_I8_PEAK_TOPS = 5000.0         # dense int8 = 2x bf16 (MI355X_MICROARCH.md, matrix-core table)


def pmc_counters(name):
    """rocprofv3 PMC results of the dominant kernel from the COMMITTED passes (profiles/README.md): HBM-side bytes per launch
    (FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
    and the MFMA-busy fraction.  Hardware counters cannot be read from inside this process: the numbers are NOT measured
    by this run, so they are reported together with their provenance (file and the commit they were collected at), and
    only when that file was collected for this config and for the kernel the run reports; otherwise null."""
    path = os.path.join(ROOT, "profiles", PROFILE_COUNTERS)
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None, None, None
    ent = doc.get(name)
    if not ent or ent.get("kernel") != ROOFLINE_KERNEL:
        return None, None, None
    src = {"file": "profiles/" + PROFILE_COUNTERS, "collected_at_commit": doc.get("_head"), "note": "rocprofv3 --pmc passes of this "
           "command, committed; not measured by this run"}
    return ent.get("bytes_per_launch"), ent.get("mfma_busy_frac"), src


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

    aug = CONFIGS[a.config][8]          # --aug_loss: the step also reads fp_inps_2 (= the teacher output of the student input; with
    fp2 = fp_inps if aug else None      # identical synthetic banks that is the same tensor as fp_inps)
    for j in range(a.warmup):
        runner.run(quant_inps[j % n:j % n + 1], fp_inps[j % n:j % n + 1], fp2[j % n:j % n + 1] if aug else None)
    sync()
    t0 = time.perf_counter()
    for j in range(a.steps):
        runner.run(quant_inps[j % n:j % n + 1], fp_inps[j % n:j % n + 1], fp2[j % n:j % n + 1] if aug else None)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    loss = float(runner.loss)
    assert loss == loss, "NaN loss in the timed region"

    roof = kernel_rooflines(runner, quant_inps, fp_inps, cfg)
    out = None
    if rank == 0:
        flops, P = linear_flops(cfg, SEQLEN)
        value = world * a.steps / dt
        traffic, mfma_busy, src = pmc_counters(a.config)
        b16, i8 = roof["bf16"], roof["int8"]
        out = {
            "metric": "calibration_sample_steps_per_sec", "value": value, "unit": "sample-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": WORKLOAD[a.config], "name": a.config, "seq_len": SEQLEN, "batch_size": 1,
                       "calib_samples": n, "blocks_per_gpu": 1, "sharding": f"layers x{world}",
                       "n_ranks_seen": int(dist.get_world_size()) if dist is not None else 1,
                       "per_gpu_sample_steps_per_sec": value / world, "last_loss": loss,
                       "linear_tflop_per_step": flops / 1e12,
                       "fprop_arithmetic": "int8 codes on v_mfma_i32_16x16x64_i8 (exact), fp32 epilogue" if i8 else "bf16"},
            # dominant kernel of the step: the bf16 MFMA GEMM (dgrad + wgrad of the fake-quant linears; + fprop when the integer
            # path is off).  achieved = algorithmic flops of those launches / their summed HIP-event durations.
            "roofline": {"bound": "mfma", "kernel": ROOFLINE_KERNEL + (" (dgrad + wgrad of the fake-quant linears)" if i8 else
                                                                       " (fprop+dgrad+wgrad of the fake-quant linears)"),
                         "achieved": b16["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": b16["tflops"] / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "mfma_busy_frac": mfma_busy,
                         "traffic_and_busy_from_profile": src,
                         "avg_launch_ms": b16["avg_launch_ms"], "launches_per_step": b16["launches_per_step"],
                         "linear_gemm_ms_per_step": b16["ms_per_step"], "tflop_per_step": b16["tflop_per_step"],
                         "all_gemm_ms_per_step": roof["gemm_ms_per_step"],
                         "event_pair_overhead_us_subtracted": roof["event_pair_overhead_us"]},
        }
        if i8:
            out["roofline_int8"] = {"bound": "mfma", "kernel": "gemm_i8_p3_kernel (fprop of the fake-quant linears on integer codes)",
                                    "achieved": i8["tflops"], "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                                    "frac": i8["tflops"] / MFMA_I8_PEAK_TOPS, "avg_launch_ms": i8["avg_launch_ms"],
                                    "launches_per_step": i8["launches_per_step"], "ms_per_step": i8["ms_per_step"],
                                    "top_per_step": i8["tflop_per_step"]}
        if roof["hbm"]:
            top = roof["hbm"][0]
            # the quantiser reductions are the HBM-class kernels of the step: the one with the most time per step leads
            out["roofline_hbm"] = {"bound": "hbm", "kernel": top["kernel"], "achieved": top["achieved_gbs"], "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": top["frac"], "avg_launch_us": top["avg_launch_us"],
                                   "launches_per_step": top["launches_per_step"],
                                   "algorithmic_bytes_per_launch": top["algorithmic_bytes_per_launch"],
                                   "all_quantiser_kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in h.items()}
                                                             for h in roof["hbm"]]}
    # ---- second region: the real sharded engine.  Its rank-to-rank exchange (RCCL send / recv) is the one part of this file no
    # single-GPU box can rehearse, so it runs behind a watchdog: whatever happens to it, rank 0 still prints the ONE JSON line
    # of the step loop above (with the failure recorded under "end_to_end") and every rank leaves.
    emitted = []

    def emit(e2e):
        if emitted:
            return
        emitted.append(1)
        if rank == 0:
            if e2e is not None:
                out["end_to_end"] = e2e
            if world == 1 and not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(a.config)
            print(json.dumps(out), flush=True)

    e2e, failed = None, False
    if not a.no_end_to_end and not a.config.startswith("opt"):
        del runner, opt, q
        torch.cuda.empty_cache()
        timer = None
        if world > 1:
            import threading
            limit = float(os.environ.get("OQ_BENCH_E2E_TIMEOUT", "240"))

            def on_timeout():
                emit({"error": f"end-to-end region did not finish within {limit:.0f} s on rank {rank}"})
                os._exit(0)
            timer = threading.Timer(limit, on_timeout)
            timer.daemon = True
            timer.start()
        try:
            e2e = end_to_end(a.config, rank, world, dev, dist, a.e2e_samples, a.blocks_per_rank, a.boundary_chunk)
        except Exception as ex:        # noqa: BLE001 -- recorded, never swallowed silently
            if world == 1:
                raise
            e2e, failed = {"error": repr(ex)[:400]}, True
        if timer is not None:
            timer.cancel()
    emit(e2e)
    if failed:
        os._exit(0)       # the other ranks may be blocked inside the exchange: no barrier; their watchdogs end them
    if dist is not None:
        if world > 1:                 # a rank whose watchdog fired is gone: do not wait for it for ever
            import threading
            t = threading.Timer(90.0, lambda: os._exit(0))
            t.daemon = True
            t.start()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
