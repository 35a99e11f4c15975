"""-m gpu: ONE production-mode calibration sample-step at FULL block shape vs the CPU oracle, for every BASELINE.json
configuration.  "Production mode" is exactly what bench.py times: bf16 activations, StepRunner's hipGraph replay,
gemm_bf16_p3_kernel (incl. its causal modes), the fused causal attention kernels, the bf16 quantiser instantiations,
gradient routing into the optimiser arena.  The oracle (oracle/ref_cpu.py::train_step, fp32, pinned to the reference by
tests/golden) runs the same step on the host cores (8-40 s per block on the GPU box).

Compared (reference step: quantize/omniquant.py:214-230):
  * the step's loss;
  * every LWC / LET gradient: cosine and relative L2 error per tensor;
  * the fake-quantised temporary weights (`temp_weight`, models/int_llama_layer.py:279-307) of every linear: within one
    bf16 ulp of the oracle's fp32 values, except the rare elements whose rounding decision sits on a tie.

Bar, against the oracle's PLAIN fp32 step: loss <= 1.5e-3 relative; per gradient tensor
  * weight-only configurations (W3A16g128, W2A16g64): cosine >= 0.99, relative L2 <= 5e-2;
  * weight-activation configurations (W4A4): cosine >= 0.99 / L2 <= 0.12, on the attention-score path (everything behind the
    per-head 4-bit q / k quantisers) cosine >= 0.985 / L2 <= 0.18 -- the level at which the fp32 step agrees with a float64
    evaluation of itself (0.984 .. 0.987, tests/diag/fp32_vs_fp64.py).  The production mode reaches it because no activation
    is rounded to 16 bits in front of a 4-bit rounding decision: integer fprop, attention on the integer grid, fp32 side
    channels for the attention output and the hidden states.  The same bars are held against the oracle's STORAGE MODEL of
    the product path (`Block.forward(act_dtype=torch.bfloat16, int_fprop=True, wide=True)`: fp32 arithmetic, every tensor
    the product materialises in bf16 rounded, value and gradient).  With the A/B switches that put 16-bit tensors back
    (OQ_WIDE=0, OQ_GRID_ATTN=0, OQ_INT_FPROP=0) only the storage-model bar and a coarse floor against fp32 apply.
Measured values are printed and written to gpurun_out/fullsize_parity.json when that directory exists.
Every shape is the full one (T = 2048; the LLaMA-2-70B block's CPU step takes about a minute on the GPU box's host cores)."""
import json
import math
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# name: (arch, wbits, abits, group, let, T, alpha)
CASES = {
    "llama-7b-w4a4-let": ("llama-7b", 4, 4, None, True, 2048, 0.5),              # BASELINE configs[2] (headline)
    "llama-7b-w3a16g128": ("llama-7b", 3, 16, 128, False, 2048, 0.5),            # configs[1]
    "llama-2-13b-w4a4-let": ("llama-2-13b", 4, 4, None, True, 2048, 0.75),       # configs[3]: H=5120, I=13824
    "llama-2-70b-w2a16g64-gqa": ("llama-2-70b", 2, 16, 64, False, 2048, 0.5),    # configs[4]: GQA 64/8, W2 g64
}


def _bf16_ulp(ref):
    a = np.maximum(np.abs(ref), 2.0 ** -126)
    return 2.0 ** (np.floor(np.log2(a)) - 7)


@pytest.mark.parametrize("name", list(CASES))
def test_production_step_vs_oracle(name):
    from oracle import ref_cpu as R
    from omniquant_amd import synthetic as S
    from omniquant_amd.calibrate import StepRunner, decoder_layer_class, default_args, register_let_parameters
    from omniquant_amd.linear import QuantLinear
    from omniquant_amd.optim import BlockOptimizer
    arch, wbits, abits, group, let, T, alpha = CASES[name]
    T = int(os.environ.get("OQ_TEST_T", T))            # diagnostics only
    cfg = S.make_config(arch)
    H = cfg.hidden_size
    try:
        cores = min(len(os.sched_getaffinity(0)), 32)
    except AttributeError:
        cores = 8
    torch.set_num_threads(max(cores, 1))
    layer = S.make_layer(cfg, seed=0, device="cpu")
    weights = {n: p.detach().float() for n, p in layer.named_parameters()}
    # the calibration bank is DATA stored in 16 bits on both sides (reference: fp16 `inps`, quantize/omniquant.py:77-79; here
    # bf16): every run below -- fp32 oracle, storage model, HIP -- sees the same bf16-representable sample
    x = S.make_calib_inputs(1, T, H, dtype=torch.float32).to(torch.bfloat16).float()
    mask, pos = S.causal_mask(T), torch.arange(T)[None]
    sc, sh = S.synth_act_stats(cfg, 1)
    # ---- oracle: teacher target + one train_step ---------------------------------------------------------------------
    cd = dict(hidden_size=H, num_attention_heads=cfg.num_attention_heads,
              num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=1e-6)
    t0 = time.time()
    blk = R.Block("llama", cd, weights, R.QuantSpec(wbits, abits, group, True, let), max_pos=T)
    if let:
        blk.register_let(sc, sh, alpha, 0, "model.layers")
    with torch.no_grad():
        tgt = blk.forward(x, mask, pos, None, False)
    temps = blk.temporaries()
    ref_tw = {n: temps[n + ".weight"].detach() for n in blk.order}
    out = blk.forward(x, mask, pos, temps=temps, act_quant=True)
    loss_o = torch.nn.functional.mse_loss(tgt, out)
    loss_o.backward()
    ref_grad = {n: p.grad.detach().clone() for n, p in blk.params.items()}
    del temps, out
    t_cpu = time.time() - t0
    # ---- oracle again in the bf16 storage model (weight-activation configurations only) ------------------------------
    emu_grad, loss_e, use_int, wide = None, None, False, False
    if abits < 16:
        blk2 = R.Block("llama", cd, weights, R.QuantSpec(wbits, abits, group, True, let), max_pos=T)
        if let:
            blk2.register_let(sc, sh, alpha, 0, "model.layers")
        bf = torch.bfloat16
        from omniquant_amd import ops as _ops
        use_int = _ops.int_fprop_on() and group is None and abits <= 8 and wbits <= 8       # the product's integer fprop (oq_gemm_i8)
        wide = use_int and _ops.wide_on() and _ops.grid_attention_on()       # the product's un-rounded side channels
        temps2 = blk2.temporaries(store_dtype=bf, int_fprop=use_int)
        out2 = blk2.forward(x.to(bf).float(), mask, pos, temps=temps2, act_quant=True, act_dtype=bf, int_fprop=use_int, wide=wide)
        loss_e = torch.nn.functional.mse_loss(tgt.to(bf).float(), out2)
        loss_e.backward()
        emu_grad = {n: p.grad.detach().clone() for n, p in blk2.params.items()}
        loss_e = float(loss_e)
        del temps2, out2, blk2
    # ---- HIP path, production mode -----------------------------------------------------------------------------------
    args = default_args(wbits=wbits, abits=abits, group_size=group, lwc=True, let=let, alpha=alpha, net=arch, nsamples=1)
    q = decoder_layer_class("llama")(cfg, layer.to(DEV), args).to(DEV)
    q.compute_dtype = torch.bfloat16
    q.set_quant_state(False, True)
    q.let = let
    if let:
        register_let_parameters(q, "llama", sc, sh, alpha, 0, DEV)
    opt = BlockOptimizer(q, args.let_lr, args.lwc_lr, args.wd)
    opt.clear_grads_in_step = False     # the gradients of the step are compared below (production clears them in oq_adamw_step)
    runner = StepRunner(q, opt, mask.to(DEV), pos.to(DEV), (1, T, H), torch.bfloat16, False, True, use_graph=True)
    runner.run(x.to(DEV).to(torch.bfloat16), tgt.to(DEV).to(torch.bfloat16))
    torch.cuda.synchronize()
    assert runner.graph is not None, "the step did not go through the hipGraph"
    loss = float(runner.loss)
    rep = {"case": name, "cpu_seconds": t_cpu, "loss": loss, "loss_ref": float(loss_o), "grads": {}, "temp_weight": {}}
    rep["loss_rel"] = abs(loss - float(loss_o)) / float(loss_o)
    if loss_e is not None:
        rep["loss_rel_bf16_model"] = abs(loss - loss_e) / loss_e
    fails = []

    def cmp(g, r):
        g, r = g.double().reshape(-1), r.double().reshape(-1)
        return float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-300)), float((g - r).norm() / (r.norm() + 1e-300))

    # the same step evaluated in float64 (tests/diag/fp32_vs_fp64.py wrote the fixture from the oracle): how far the production
    # step and the fp32 oracle step each are from it
    f64 = None
    f64_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_f64_llama7b_w4a4_t2048.npz")
    if name == "llama-7b-w4a4-let" and T == 2048 and os.path.exists(f64_path):
        f64 = np.load(f64_path)
        rep["loss_rel_f64"] = abs(loss - float(f64["loss"])) / float(f64["loss"])
        rep["oracle_fp32_loss_rel_f64"] = abs(float(loss_o) - float(f64["loss"])) / float(f64["loss"])
    for n, p in q.named_parameters():
        g = p.grad.detach().cpu()
        cos, l2 = cmp(g, ref_grad[n])
        rep["grads"][n] = {"cos": cos, "l2": l2, "norm_ref": float(ref_grad[n].norm())}
        if f64 is not None:
            r64 = torch.from_numpy(f64["grad." + n])
            c64, l64 = cmp(g, r64)
            o64, ol64 = cmp(ref_grad[n], r64)
            rep["grads"][n].update({"cos_f64": c64, "l2_f64": l64, "oracle_fp32_vs_f64_cos": o64, "oracle_fp32_vs_f64_l2": ol64})
        if emu_grad is not None:
            cos, l2 = cmp(g, emu_grad[n])       # the bar is held against the bf16 storage model of the same step
            rep["grads"][n].update({"cos_bf16_model": cos, "l2_bf16_model": l2})
            c2, l22 = cmp(emu_grad[n], ref_grad[n])
            rep["grads"][n].update({"oracle_bf16_vs_fp32_cos": c2, "oracle_bf16_vs_fp32_l2": l22})
        score = any(t in n for t in ("q_proj", "k_proj", "qkt_smooth"))
        if emu_grad is None:
            # weight-only configurations: held against the oracle's plain fp32 step
            ok = cos >= 0.99 and l2 <= 5e-2
        elif wide:
            # Integer fprop + attention on the integer grid + fp32 side channels: no activation is rounded to 16 bits in front of
            # a 4-bit rounding decision any more (gate | up excepted), and the bar is held against the PLAIN fp32 oracle step as
            # well as against the storage model.  Measured: 7B 0.9924 .. 0.9934 on the attention-score path (relative L2 0.115 ..
            # 0.123), >= 0.9991 elsewhere; 13B 0.9940 .. 0.9945 / >= 0.9962.  The fp32 step itself is only defined to that
            # accuracy: evaluated in float64 it moves by cosine 0.984 .. 0.987 on the score path (tests/diag/fp32_vs_fp64.py) --
            # fp32 summation noise flips a few 1e-5 of the 4-bit head-quantiser decisions and every flip moves a whole
            # attention row (tests/diag/forward_flips.py).
            c32, l32 = rep["grads"][n]["cos"], rep["grads"][n]["l2"]
            ok = (c32 >= 0.985 and l32 <= 0.18 and cos >= 0.985 and l2 <= 0.18) if score else \
                 (c32 >= 0.99 and l32 <= 0.12 and cos >= 0.99 and l2 <= 0.12)
        else:
            # A/B modes (OQ_WIDE=0 / OQ_GRID_ATTN=0 / OQ_INT_FPROP=0): bf16 tensors in front of 4-bit decisions decorrelate the
            # quantisation-residue gradients from an fp32 run by precision mode; the bar is held against the storage model only
            if score:
                ok = cos >= 0.97 and l2 <= 0.25
            elif "v_proj" in n or "out_smooth" in n:      # behind P @ V as well
                ok = cos >= 0.98 and l2 <= 0.2
            else:
                ok = cos >= 0.995 and l2 <= 0.1
            c32 = rep["grads"][n]["cos"]
            floor = 0.86 if score else (0.90 if ("v_proj" in n or "out_smooth" in n) else 0.975)
            if use_int and c32 < floor:
                fails.append((n, "vs plain fp32", round(c32, 5), "floor", floor))
        if not ok:
            fails.append((n, round(cos, 5), round(l2, 4), round(rep["grads"][n]["cos"], 5), round(rep["grads"][n]["l2"], 4)))
    mods = {n: m for n, m in q.named_modules() if isinstance(m, QuantLinear)}
    for n, ref in ref_tw.items():
        got = mods[n].temp_weight.detach().float().cpu().numpy().astype(np.float64)
        ref = ref.numpy().astype(np.float64)
        d = np.abs(got - ref)
        off = d > _bf16_ulp(ref)                        # > 1 bf16 ulp: only a flipped rounding decision can do that
        step = np.abs(ref).max() / (2 ** (wbits - 1))
        rep["temp_weight"][n] = {"frac_beyond_1ulp": float(off.mean()), "max_err_in_steps": float(d.max() / step)}
        assert off.mean() <= 2e-4 and d.max() <= 2.5 * step, (n, rep["temp_weight"][n])
    print(json.dumps({k: v for k, v in rep.items() if k not in ("grads", "temp_weight")}),
          "worst grad:", min((v["cos"], k) for k, v in rep["grads"].items()), max((v["l2"], k) for k, v in rep["grads"].items()))
    if os.path.isdir("gpurun_out"):
        path = os.path.join("gpurun_out", "fullsize_parity.json")
        allrep = json.load(open(path)) if os.path.exists(path) else {}
        allrep[name] = rep
        json.dump(allrep, open(path, "w"), indent=1)
    assert math.isfinite(loss) and rep["loss_rel"] <= (1.5e-3 if (wide or emu_grad is None) else 1e-2), rep["loss_rel"]
    if loss_e is not None:
        assert rep["loss_rel_bf16_model"] <= 1e-2, rep["loss_rel_bf16_model"]
        worst = min((v["cos_bf16_model"], k) for k, v in rep["grads"].items())
        print("vs bf16 storage model: worst cos", worst, "worst l2", max((v["l2_bf16_model"], k) for k, v in rep["grads"].items()))
    if f64 is not None:
        worst_hip = min((v["cos_f64"], k) for k, v in rep["grads"].items())
        worst_o32 = min((v["oracle_fp32_vs_f64_cos"], k) for k, v in rep["grads"].items())
        print("vs the float64 evaluation of the step: production", worst_hip, "| fp32 oracle", worst_o32)
        if wide:
            # the production step is as close to the float64 step as the reference's own fp32 arithmetic is (margin: the
            # two distances are both made of a few hundred flipped 4-bit decisions)
            for k, v in rep["grads"].items():
                if v["cos_f64"] < v["oracle_fp32_vs_f64_cos"] - 0.02:
                    fails.append((k, "vs float64", round(v["cos_f64"], 5), "fp32 oracle:", round(v["oracle_fp32_vs_f64_cos"], 5)))
    assert not fails, fails
