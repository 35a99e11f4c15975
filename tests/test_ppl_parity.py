"""End-to-end parity on a small random-init LLaMA-style model: block-wise calibration on the HIP path (float32 parity
mode, hipGraph step) vs the CPU oracle, then the post-quant perplexity of both (arithmetic of the reference's
evaluate(), main.py:116-148: nll = CE * seqlen per window, ppl = exp(sum nll / (nsamples * seqlen))).
North-star bar: learned tensors and post-quant PPL within 1e-3 relative.

What is and is not achievable: the two trajectories agree to ~1e-6 step by step until the first 4-bit activation
rounding decision flips (an fp32 summation-order difference of 1 ulp in a GEMM output that sits on a rounding
boundary); from there on AdamW's sign-like updates decorrelate near-zero learnables (the LET shifts) and the
trajectories differ at the level of the quantisation noise -- exactly as two runs of the REFERENCE differ when its
inputs are perturbed by 1 ulp.  The test therefore measures that noise floor with the oracle itself (same
calibration, inputs perturbed by 1e-6 relative) and requires the HIP path to stay within a small multiple of it."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CFG = dict(hidden_size=128, intermediate_size=256, num_attention_heads=4, num_key_value_heads=4, rms_norm_eps=1e-6)
VOCAB, T, NSAMP, NLAYERS = 256, 64, 8, 3


def _model(seed=0, CFG=CFG, T=T, NSAMP=NSAMP):
    g = torch.Generator().manual_seed(seed)
    H, I = CFG["hidden_size"], CFG["intermediate_size"]
    emb = torch.randn(VOCAB, H, generator=g) * 0.5
    emb[:, 7] *= 6.0
    head = torch.randn(VOCAB, H, generator=g) * 0.02      # tame logits: PPL stays O(vocab), as for a real LM head
    fnorm = 1 + 0.1 * torch.randn(H, generator=g)
    layers = []
    for _ in range(NLAYERS):
        w = {}
        for n, shp in (("self_attn.q_proj", (H, H)), ("self_attn.k_proj", (H, H)), ("self_attn.v_proj", (H, H)),
                       ("self_attn.o_proj", (H, H)), ("mlp.gate_proj", (I, H)), ("mlp.up_proj", (I, H)),
                       ("mlp.down_proj", (H, I))):
            w[n + ".weight"] = (torch.randn(shp, generator=g) * 0.08).half().float()
        w["input_layernorm.weight"] = (1 + 0.1 * torch.randn(H, generator=g)).half().float()
        w["post_attention_layernorm.weight"] = (1 + 0.1 * torch.randn(H, generator=g)).half().float()
        layers.append(w)
    tokens = torch.randint(0, VOCAB, (NSAMP, T), generator=g)
    return emb, head, fnorm, layers, tokens


def _ppl(hidden, fnorm, head, tokens):
    """hidden [n, T, H] after the last block -> final RMSNorm -> lm_head -> shifted CE -> reference PPL arithmetic."""
    T = tokens.shape[1]
    var = hidden.pow(2).mean(-1, keepdim=True)
    h = fnorm * (hidden * torch.rsqrt(var + 1e-6))
    logits = h @ head.T
    nlls = []
    for i in range(tokens.shape[0]):
        loss = torch.nn.functional.cross_entropy(logits[i, :-1].float(), tokens[i, 1:])
        nlls.append(loss.float() * T)
    return float(torch.exp(torch.stack(nlls).sum() / (tokens.shape[0] * T)))


def test_post_quant_ppl_matches_oracle():
    from omniquant_amd.calibrate import calibrate_layers, default_args
    from omniquant_amd.synthetic import make_config, make_layer
    emb, head, fnorm, layers, tokens = _model()
    inps = emb[tokens]                                   # layer-0 inputs, as the reference's Catcher records them
    mask = torch.triu(torch.full((T, T), torch.finfo(torch.float32).min), 1)[None, None]
    pos = torch.arange(T)[None]
    g = torch.Generator().manual_seed(5)
    names = ["self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj"]
    sc = {f"model.layers.{i}.{n}": torch.rand(CFG["hidden_size"], generator=g) * 3 + 0.2 for i in range(NLAYERS) for n in names}
    sh = {k: torch.zeros(CFG["hidden_size"]) for k in sc}
    epochs = 3
    # ---- oracle (CPU fp32) ----
    spec = R.QuantSpec(4, 4, None, True, True)
    ref = R.calibrate("llama", CFG, layers, spec, inps, mask, pos, sc, sh, epochs=epochs)
    ppl_ref = _ppl(ref["quant_out"][-1], fnorm, head, tokens)
    ppl_fp = _ppl(ref["fp_out"][-1], fnorm, head, tokens)
    # ---- HIP path (fp32 parity mode, graph-replayed steps) ----
    cfg = make_config(None, family="llama", hidden_size=CFG["hidden_size"], inter=CFG["intermediate_size"], heads=4, kv_heads=4)
    args = default_args(wbits=4, abits=4, lwc=True, let=True, epochs=epochs, nsamples=NSAMP, net="llama")
    hip_layers = [make_layer(cfg, weights={k: v for k, v in w.items()}, device=DEV) for w in layers]
    _, omni, losses, (qi, fi) = calibrate_layers(hip_layers, cfg, args, inps.to(DEV), mask.to(DEV), pos.to(DEV), sc, sh,
                                                 use_graph=True, compute_dtype=torch.float32)
    ppl_hip = _ppl(qi.float().cpu(), fnorm, head, tokens)
    ppl_fp_hip = _ppl(fi.float().cpu(), fnorm, head, tokens)
    # noise floor of the reference algorithm itself: the same calibration with inputs perturbed by 1e-6 relative
    # (5 seeds).  The oracle's own post-quant PPL scatters by ~1 % on this tiny eval set once a rounding decision flips.
    samples = [ppl_ref]
    for seed in (9, 10, 11, 12, 13):
        gp = torch.Generator().manual_seed(seed)
        refp = R.calibrate("llama", CFG, layers, spec, inps * (1 + 1e-6 * torch.randn(inps.shape, generator=gp)), mask,
                           pos, sc, sh, epochs=epochs)
        samples.append(_ppl(refp["quant_out"][-1], fnorm, head, tokens))
    lo_s, hi_s = min(samples), max(samples)
    spread = hi_s - lo_s
    print(f"PPL fp {ppl_fp:.4f} (hip {ppl_fp_hip:.4f}); post-quant oracle {ppl_ref:.4f} hip {ppl_hip:.4f}; "
          f"oracle under 1e-6 input noise: {', '.join(f'{v:.3f}' for v in samples[1:])} (spread {spread:.3f})")
    assert abs(ppl_fp_hip - ppl_fp) / ppl_fp < 1e-4
    # the HIP result must be statistically indistinguishable from a re-run of the reference algorithm
    assert lo_s - 0.5 * spread - 1e-3 * ppl_ref <= ppl_hip <= hi_s + 0.5 * spread + 1e-3 * ppl_ref, (ppl_hip, samples)
    assert abs(ppl_hip - ppl_ref) / ppl_ref < 2e-2
    # before the first rounding flip the trajectories are identical; afterwards they stay statistically close
    l_hip, l_ref = np.asarray(losses), np.asarray(ref["losses"])
    np.testing.assert_allclose(l_hip[:8], l_ref[:8], rtol=1e-4)
    # (per-step losses of single samples scatter once the runs have decorrelated: compare epoch means)
    np.testing.assert_allclose(l_hip.reshape(-1, NSAMP).mean(1), l_ref.reshape(-1, NSAMP).mean(1), rtol=0.1)
    # layer 0 LWC clips (not chaotic: O(1) values with O(1) gradients) stay within the north-star 1e-3 * few
    for n, t in ref["omni"][0].items():
        if "bound_factor" in n:
            got, want = omni[0][n].double().numpy(), t.double().numpy()
            assert np.abs(got - want).max() <= 2e-2 * np.abs(want).max(), n


def test_post_quant_ppl_bf16_mode_close():
    """Production bf16 mode on the same model: not elementwise identical by construction, PPL delta must stay small."""
    from omniquant_amd.calibrate import calibrate_layers, default_args
    from omniquant_amd.synthetic import make_config, make_layer
    emb, head, fnorm, layers, tokens = _model()
    inps = emb[tokens]
    mask = torch.triu(torch.full((T, T), torch.finfo(torch.float32).min), 1)[None, None]
    pos = torch.arange(T)[None]
    spec = R.QuantSpec(4, 16, None, True, False)
    ref = R.calibrate("llama", CFG, layers, spec, inps, mask, pos, epochs=3)
    ppl_ref = _ppl(ref["quant_out"][-1], fnorm, head, tokens)
    cfg = make_config(None, family="llama", hidden_size=CFG["hidden_size"], inter=CFG["intermediate_size"], heads=4, kv_heads=4)
    args = default_args(wbits=4, abits=16, lwc=True, let=False, epochs=3, nsamples=NSAMP, net="llama")
    hip_layers = [make_layer(cfg, weights={k: v for k, v in w.items()}, device=DEV) for w in layers]
    _, omni, losses, (qi, fi) = calibrate_layers(hip_layers, cfg, args, inps.to(DEV), mask.to(DEV), pos.to(DEV),
                                                 use_graph=True, compute_dtype=torch.bfloat16)
    ppl_hip = _ppl(qi.float().cpu(), fnorm, head, tokens)
    print(f"bf16 mode: post-quant PPL oracle {ppl_ref:.4f} hip {ppl_hip:.4f}")
    assert math.isfinite(ppl_hip) and abs(ppl_hip - ppl_ref) / ppl_ref < 2e-2, (ppl_hip, ppl_ref)


def test_post_quant_ppl_bf16_w4a4_let_production_kernels():
    """The headline configuration (W4A4 --lwc --let) in production mode -- bf16 MFMA GEMMs, fused causal attention
    (head_dim 128, T 256), hipGraph-replayed steps -- on a 3-layer random-init LLaMA: post-quant PPL vs the CPU oracle's.
    The oracle's own PPL scatters once a rounding decision flips (see the module docstring); the bf16 result must sit
    inside that scatter widened by the north-star PPL tolerance."""
    from omniquant_amd.calibrate import calibrate_layers, default_args
    from omniquant_amd.synthetic import make_config, make_layer
    cfg_d = dict(hidden_size=256, intermediate_size=512, num_attention_heads=2, num_key_value_heads=2, rms_norm_eps=1e-6)
    Tn, ns, epochs = 256, 8, 3
    emb, head, fnorm, layers, tokens = _model(seed=1, CFG=cfg_d, T=Tn, NSAMP=ns)
    inps = emb[tokens]
    mask = torch.triu(torch.full((Tn, Tn), torch.finfo(torch.float32).min), 1)[None, None]
    pos = torch.arange(Tn)[None]
    g = torch.Generator().manual_seed(5)
    names = ["self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj"]
    sc = {f"model.layers.{i}.{n}": torch.rand(256, generator=g) * 3 + 0.2 for i in range(NLAYERS) for n in names}
    sh = {k: torch.zeros(256) for k in sc}
    spec = R.QuantSpec(4, 4, None, True, True)
    ref = R.calibrate("llama", cfg_d, layers, spec, inps, mask, pos, sc, sh, epochs=epochs)
    ppl_ref = _ppl(ref["quant_out"][-1], fnorm, head, tokens)
    ppl_fp = _ppl(ref["fp_out"][-1], fnorm, head, tokens)
    samples = [ppl_ref]
    for seed in (9, 10, 11):
        gp = torch.Generator().manual_seed(seed)
        refp = R.calibrate("llama", cfg_d, layers, spec, inps * (1 + 1e-6 * torch.randn(inps.shape, generator=gp)), mask,
                           pos, sc, sh, epochs=epochs)
        samples.append(_ppl(refp["quant_out"][-1], fnorm, head, tokens))
    cfg = make_config(None, family="llama", hidden_size=256, inter=512, heads=2, kv_heads=2)
    args = default_args(wbits=4, abits=4, lwc=True, let=True, epochs=epochs, nsamples=ns, net="llama")
    hip_layers = [make_layer(cfg, weights={k: v for k, v in w.items()}, device=DEV) for w in layers]
    from omniquant_amd import ops
    used = {"n": 0}
    orig = ops.FusedCausalAttnFn.apply

    class _Spy:
        @staticmethod
        def apply(*a):
            used["n"] += 1
            return orig(*a)
    ops.FusedCausalAttnFn, keep = _Spy, ops.FusedCausalAttnFn
    try:
        _, omni, losses, (qi, fi) = calibrate_layers(hip_layers, cfg, args, inps.to(DEV), mask.to(DEV), pos.to(DEV), sc, sh,
                                                     use_graph=True, compute_dtype=torch.bfloat16)
    finally:
        ops.FusedCausalAttnFn = keep
    assert used["n"] > 0
    ppl_hip = _ppl(qi.float().cpu(), fnorm, head, tokens)
    ppl_fp_hip = _ppl(fi.float().cpu(), fnorm, head, tokens)
    lo_s, hi_s = min(samples), max(samples)
    spread = hi_s - lo_s
    print(f"bf16 W4A4+LET: PPL fp {ppl_fp:.4f} (hip {ppl_fp_hip:.4f}); post-quant oracle {ppl_ref:.4f} hip {ppl_hip:.4f}; "
          f"oracle under 1e-6 input noise {', '.join(f'{v:.3f}' for v in samples[1:])}")
    assert abs(ppl_fp_hip - ppl_fp) / ppl_fp < 5e-3                       # bf16 teacher pass
    assert math.isfinite(ppl_hip)
    assert lo_s - spread - 1e-2 * ppl_ref <= ppl_hip <= hi_s + spread + 1e-2 * ppl_ref, (ppl_hip, samples)
    l_hip, l_ref = np.asarray(losses), np.asarray(ref["losses"])
    np.testing.assert_allclose(l_hip.reshape(-1, ns).mean(1), l_ref.reshape(-1, ns).mean(1), rtol=0.1)


def test_sharding_boundaries_cost_little_ppl():
    """SURVEY.md 8e's validity criterion for the layer-sharded engine: every chunk boundary replaces the student input of
    the chunk's first block (the output of the already calibrated, quantised predecessor -- quantize/omniquant.py:219,245)
    by the TEACHER activation.  Worst case = one decoder block per rank = a boundary after EVERY layer.  Single GPU, no
    processes: the 3-layer model is calibrated (a) sequentially and (b) block by block with the teacher bank as student
    input; the post-quant PPL of both quantised models (run as a chain from the true input, as evaluation does) and the
    deviation of the learned tensors are reported, and the PPL delta must stay inside the oracle's own run-to-run scatter
    plus the north-star's 0.05."""
    from omniquant_amd.calibrate import calibrate_layers, default_args, forward_bank
    from omniquant_amd.synthetic import make_config, make_layer
    emb, head, fnorm, layers, tokens = _model()
    inps = emb[tokens]
    mask = torch.triu(torch.full((T, T), torch.finfo(torch.float32).min), 1)[None, None]
    pos = torch.arange(T)[None]
    g = torch.Generator().manual_seed(5)
    names = ["self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj"]
    sc = {f"model.layers.{i}.{n}": torch.rand(CFG["hidden_size"], generator=g) * 3 + 0.2 for i in range(NLAYERS) for n in names}
    sh = {k: torch.zeros(CFG["hidden_size"]) for k in sc}
    epochs = 3
    cfg = make_config(None, family="llama", hidden_size=CFG["hidden_size"], inter=CFG["intermediate_size"], heads=4, kv_heads=4)
    args = default_args(wbits=4, abits=4, lwc=True, let=True, epochs=epochs, nsamples=NSAMP, net="llama")
    mk = lambda: [make_layer(cfg, weights={k: v for k, v in w.items()}, device=DEV) for w in layers]
    x0, m_d, p_d = inps.to(DEV), mask.to(DEV), pos.to(DEV)

    def chain_ppl(qlayers):
        bank = x0.float().clone()
        for q in qlayers:
            q.compute_dtype = torch.float32
            q.float()
            forward_bank(q, bank, bank, m_d.float(), p_d, True)
        return _ppl(bank.cpu(), fnorm, head, tokens)

    # (a) sequential = the reference's schedule
    q_seq, omni_seq, _, (qi, fi) = calibrate_layers(mk(), cfg, args, x0, m_d, p_d, sc, sh, use_graph=True,
                                                    compute_dtype=torch.float32)
    # evaluated as main.py does: the returned layers are fp16 (qlayer.half(), quantize/omniquant.py:247), so the chain differs from
    # the engine's own fp32 propagate bank by the fp16 rounding of the folded weights (same for both schedules)
    ppl_seq = chain_ppl(q_seq)
    ppl_seq_bank = _ppl(qi.float().cpu(), fnorm, head, tokens)
    assert abs(ppl_seq - ppl_seq_bank) / ppl_seq < 2e-2
    # (b) a boundary after every layer: teacher bank in, teacher bank as student input (parallel.calibrate_sharded, G = L)
    hip_layers = mk()
    teacher = x0.float().clone()
    q_shard, omni_shard = [], {}
    for i in range(NLAYERS):
        ql, om, _, (_, fp_next) = calibrate_layers(hip_layers[i:i + 1], cfg, args, teacher, m_d, p_d, sc, sh, use_graph=True,
                                                   compute_dtype=torch.float32, layer_offset=i, student_inps=teacher)
        q_shard += ql
        omni_shard.update(om)
        teacher = fp_next.float()
    ppl_shard = chain_ppl(q_shard)
    # the reference algorithm's own scatter (inputs perturbed by 1e-6 relative), as in test_post_quant_ppl_matches_oracle
    spec = R.QuantSpec(4, 4, None, True, True)
    samples = []
    for seed in (None, 9, 10, 11):
        xin = inps if seed is None else inps * (1 + 1e-6 * torch.randn(inps.shape, generator=torch.Generator().manual_seed(seed)))
        samples.append(_ppl(R.calibrate("llama", CFG, layers, spec, xin, mask, pos, sc, sh, epochs=epochs)["quant_out"][-1],
                            fnorm, head, tokens))
    spread = max(samples) - min(samples)
    dev = {}
    for i in range(NLAYERS):
        worst = 0.0
        for n, t in omni_seq[i].items():
            a, b = t.double(), omni_shard[i][n].double()
            worst = max(worst, float((a - b).abs().max() / a.abs().max().clamp_min(1e-6)))
        dev[i] = worst
    print(f"sharding validity: post-quant PPL sequential {ppl_seq:.4f} (engine's fp32 bank {ppl_seq_bank:.4f}), boundary after every layer {ppl_shard:.4f} "
          f"(delta {ppl_shard - ppl_seq:+.4f} = {100 * (ppl_shard - ppl_seq) / ppl_seq:+.3f} %); oracle scatter under 1e-6 input "
          f"noise {min(samples):.3f}..{max(samples):.3f}; learned-tensor deviation per layer (max rel) "
          + ", ".join(f"L{i} {v:.3e}" for i, v in dev.items()))
    assert dev[0] == 0.0                                   # layer 0 sees the true input either way: bit-identical
    assert math.isfinite(ppl_shard)
    assert abs(ppl_shard - ppl_seq) <= spread + 0.05 + 1e-2 * ppl_seq, (ppl_seq, ppl_shard, samples)
