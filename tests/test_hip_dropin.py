"""-m gpu tests of the drop-in entry points a reference caller hits first (main.py -> quantize/omniquant.py::omniquant,
models/int_llama_layer.py's calls into QuantMatMul) and of the public decoder-layer surface outside the calibration
engine's habits: batched masks that differ per sample, fp16 hidden states, grouped-query attention with LET, W16."""
import math
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ---- a stub `lm` with the attribute surface quantize/omniquant.py:30-113 touches -------------------------------------
class _Inner(nn.Module):
    def __init__(self, cfg, n_layers, vocab, seed):
        super().__init__()
        from omniquant_amd.synthetic import make_layer
        g = torch.Generator().manual_seed(seed)
        self.embed_tokens = nn.Embedding(vocab, cfg.hidden_size)
        with torch.no_grad():
            self.embed_tokens.weight.copy_(torch.randn(vocab, cfg.hidden_size, generator=g))
        self.embed_tokens = self.embed_tokens.half()
        self.layers = nn.ModuleList([make_layer(cfg, seed=seed + 1 + i) for i in range(n_layers)])
        self.norm = nn.Identity()


class _HFLikeModel(nn.Module):
    """forward(ids): embeds, builds the additive mask the way transformers does for an fp16 model
    (finfo(float16).min above the diagonal, [bs,1,T,T], fp16) and calls layer(h, attention_mask=, position_ids=)."""

    def __init__(self, cfg, n_layers=2, vocab=64, seed=0):
        super().__init__()
        self.config = cfg
        self.model = _Inner(cfg, n_layers, vocab, seed)

    def forward(self, ids):
        h = self.model.embed_tokens(ids)
        T = ids.shape[1]
        mask = torch.triu(torch.full((T, T), torch.finfo(torch.float16).min, dtype=torch.float16, device=ids.device), 1)
        mask = mask[None, None].expand(ids.shape[0], 1, T, T)
        pos = torch.arange(T, device=ids.device)[None]
        for layer in self.model.layers:
            h = layer(h, attention_mask=mask, position_ids=pos)[0]
        return h


class _LM:
    def __init__(self, model, seqlen):
        self.model, self.seqlen, self.device = model, seqlen, torch.device(DEV)
        self._device = self.device


class _Log:
    def __init__(self):
        self.lines = []

    def info(self, m):
        self.lines.append(str(m))


def _setup(epochs, out_dir, resume=None, seed=0, **kw):
    from omniquant_amd.calibrate import default_args
    from omniquant_amd.synthetic import make_config
    cfg = make_config(None, family="llama", hidden_size=256, inter=512, heads=2, kv_heads=2)
    T, ns = 256, 4
    lm = _LM(_HFLikeModel(cfg, 2, 64, seed), T)
    args = default_args(wbits=4, abits=4, lwc=True, let=True, epochs=epochs, nsamples=ns, net="llama-tiny-stub",
                        output_dir=out_dir, resume=resume, **kw)
    g = torch.Generator().manual_seed(7)
    loader = [(torch.randint(0, 64, (1, T), generator=g), None) for _ in range(ns + 2)]
    return lm, args, loader


def test_omniquant_entry_writes_checkpoint_and_resumes(tmp_path):
    """omniquant(lm, args, dataloader, act_scales, act_shifts, logger) end to end on a 2-layer random LLaMA
    (quantize/omniquant.py:20-289): Catcher capture of the layer-0 inputs with an HF-style fp16 mask (must take the
    causal fast path, i.e. the fused attention), on-the-fly LET statistics, bf16 hipGraph steps, per-layer rewrite of
    output_dir/omni_parameters.pth, the returned fp16 model runs on fp16 hidden states; then `--resume` + `--epochs 0`
    reproduces the folded weights from the checkpoint alone."""
    from omniquant_amd import ops
    from omniquant_amd.calibrate import omniquant
    from omniquant_amd.llama_block import QuantLlamaDecoderLayer
    out = str(tmp_path)
    lm, args, loader = _setup(2, out)
    log = _Log()
    used = {"fused": 0}
    orig = ops.FusedCausalAttnFn.apply

    class _Spy:
        @staticmethod
        def apply(*a):
            used["fused"] += 1
            return orig(*a)
    ops.FusedCausalAttnFn, keep = _Spy, ops.FusedCausalAttnFn
    try:
        model = omniquant(lm, args, loader, None, None, log)
    finally:
        ops.FusedCausalAttnFn = keep
    assert used["fused"] > 0, "the HF fp16 mask (-65504) must select the causal fast path"
    assert model is lm.model and all(isinstance(l, QuantLlamaDecoderLayer) for l in model.model.layers)
    assert any("=== Start quantize layer 1 ===" in s for s in log.lines) and any("loss:" in s for s in log.lines)
    path = os.path.join(out, "omni_parameters.pth")
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    assert sorted(ckpt.keys()) == [0, 1]
    for i in (0, 1):
        names = set(ckpt[i].keys())
        assert {"qkt_smooth_scale", "qkv_smooth_scale", "qkv_smooth_shift", "out_smooth_scale", "fc1_smooth_shift",
                "self_attn.q_proj.weight_quantizer.upbound_factor", "mlp.down_proj.weight_quantizer.lowbound_factor"} <= names
        assert len(names) == 21 and all(v.dtype == torch.float16 for v in ckpt[i].values())
        lin = model.model.layers[i].self_attn.q_proj
        assert lin.weight.dtype == torch.float16 and lin.weight_quantizer.scales.shape == (256, 1)
    # the returned model keeps the reference's dtype contract: fp16 in, fp16 out (evaluate() feeds fp16 hidden states)
    model.model.embed_tokens.to(DEV)
    for l in model.model.layers:
        l.to(DEV)
    with torch.no_grad():
        y = model(loader[0][0].to(DEV))
    assert y.dtype == torch.float16 and bool(torch.isfinite(y.float()).all())
    w_first = [model.model.layers[i].mlp.down_proj.weight.detach().float().cpu().clone() for i in (0, 1)]
    wq_first = [model.model.layers[i].self_attn.q_proj.weight.detach().float().cpu().clone() for i in (0, 1)]
    # ---- --resume <ckpt> --epochs 0: no training, learned tensors come from the file, fold is deterministic ----------
    lm2, args2, loader2 = _setup(0, None, resume=path)
    model2 = omniquant(lm2, args2, loader2, None, None, _Log())
    for i in (0, 1):
        a = model2.model.layers[i].mlp.down_proj.weight.detach().float().cpu()
        b = model2.model.layers[i].self_attn.q_proj.weight.detach().float().cpu()
        # fp16 learnables round-trip through the file (quirk Q10): the fold sees fp16(gamma), fp16(scale) instead of the
        # fp32 values, which moves every grid point by ~1e-3 relative and flips a rounding decision for a few elements
        # (one 4-bit step = up to ~0.15 of the row's range)
        for got, ref in ((a, w_first[i]), (b, wq_first[i])):
            mx = float(ref.abs().max())
            d = (got - ref).abs()
            assert float(d.max()) <= 0.2 * mx and float((d > 1e-2 * mx).float().mean()) < 0.03
    # and without the checkpoint the epochs-0 fold is the plain init (differs from the trained one)
    lm3, args3, loader3 = _setup(0, None)
    model3 = omniquant(lm3, args3, loader3, None, None, _Log())
    c = model3.model.layers[0].self_attn.q_proj.weight.detach().float().cpu()
    assert float((c - wq_first[0]).abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_quantmatmul_forward_reference_call_shapes(dtype):
    """QuantMatMul.forward(x1, x2) with the reference's operands (models/int_llama_layer.py:143,161):
    [bs,nh,T,hd] @ [bs,nh,hd,T] and [bs,nh,T,T] @ [bs,nh,T,hd]; OPT's bmm shapes [bs*nh,T,hd] @ [bs*nh,hd,T]."""
    from omniquant_amd.matmul import QuantMatMul
    from omniquant_amd.calibrate import default_args
    a = default_args(abits=4)
    mm = QuantMatMul(a.q_quant_params, a.k_quant_params, matmul_func=torch.matmul).to(DEV)
    g = torch.Generator().manual_seed(0)
    bs, nh, T, hd = 2, 3, 64, 32
    q = torch.randn(bs, nh, T, hd, generator=g).to(DEV).to(dtype)
    kt = torch.randn(bs, nh, hd, T, generator=g).to(DEV).to(dtype)
    v = torch.randn(bs, nh, T, hd, generator=g).to(DEV).to(dtype)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    s = mm(q, kt)
    ref = torch.matmul(q.double(), kt.double())
    assert s.shape == (bs, nh, T, T) and s.dtype == dtype
    assert float((s.double() - ref).abs().max()) <= tol * float(ref.abs().max())
    p = torch.softmax(ref, -1).to(dtype)
    o = mm(p, v)
    refo = torch.matmul(p.double(), v.double())
    assert o.shape == (bs, nh, T, hd)
    assert float((o.double() - refo).abs().max()) <= tol * float(refo.abs().max())
    bm = QuantMatMul(a.q_quant_params, a.k_quant_params, matmul_func=torch.bmm).to(DEV)
    s3 = bm(q.reshape(bs * nh, T, hd), kt.reshape(bs * nh, hd, T))
    assert s3.shape == (bs * nh, T, T)
    assert float((s3.double() - ref.reshape(bs * nh, T, T)).abs().max()) <= tol * float(ref.abs().max())
    # quant_x1 / quant_x2 obey the act-quant switch exactly like the reference (quantize/int_matmul.py:31-39)
    assert mm.quant_x1(q) is q
    mm.set_quant_state(False, True)
    q4 = mm.quant_x1(q.float())
    assert q4 is not q and int(torch.unique(q4[0, 0, 0]).numel()) <= 16


def _tiny_block(let, kv_heads=2, wbits=4, abits=4, seed=5, hidden=256, heads=2):
    from omniquant_amd.calibrate import default_args, register_let_parameters
    from omniquant_amd.synthetic import make_config, make_layer, synth_act_stats
    from omniquant_amd.llama_block import QuantLlamaDecoderLayer
    cfg = make_config(None, family="llama", hidden_size=hidden, inter=2 * hidden, heads=heads, kv_heads=kv_heads)
    args = default_args(wbits=wbits, abits=abits, lwc=True, let=let, epochs=1, nsamples=2, net="llama")
    layer = make_layer(cfg, seed=seed, device=DEV)
    q = QuantLlamaDecoderLayer(cfg, layer, args).to(DEV)
    q.set_quant_state(False, True)
    q.let = let
    if let:
        sc, sh = synth_act_stats(cfg, 1)
        register_let_parameters(q, "llama", sc, sh, 0.5, 0, DEV)
    with torch.no_grad():
        for p in q.parameters():
            p.data = p.data.float()
    return cfg, args, layer, q


def test_batched_masks_that_differ_per_sample():
    """[bs,1,T,T] masks that are NOT the same for every sample (left padding): each sample must get its own mask, as
    the reference's `attn_weights + attention_mask` does (models/int_llama_layer.py:150-157)."""
    from omniquant_amd.synthetic import make_calib_inputs
    _, _, _, q = _tiny_block(True)
    q.compute_dtype = torch.float32
    T = 64
    x = make_calib_inputs(2, T, 256, dtype=torch.float32).to(DEV)
    neg = torch.finfo(torch.float32).min
    m = torch.triu(torch.full((T, T), neg, device=DEV), 1)[None, None].repeat(2, 1, 1, 1)
    m[1, 0, :, :5] = neg                       # sample 1: first five key positions are padding
    m[1, 0, :5, :5] = torch.triu(torch.full((5, 5), neg, device=DEV), 1)   # keep every row with one live key
    pos = torch.arange(T, device=DEV)[None]
    q.smooth_and_quant_temporary()
    with torch.no_grad():
        both = q(x, attention_mask=m, position_ids=pos)[0]
        one0 = q(x[:1], attention_mask=m[:1], position_ids=pos)[0]
        one1 = q(x[1:], attention_mask=m[1:], position_ids=pos)[0]
    assert torch.equal(both[0], one0[0]) and torch.equal(both[1], one1[0])
    with torch.no_grad():
        wrong = q(x[1:], attention_mask=m[:1], position_ids=pos)[0]      # sample 1 under sample 0's mask differs
    assert float((wrong[0] - one1[0]).abs().max()) > 1e-3
    with pytest.raises(NotImplementedError):                               # per-sample position ids are refused, not ignored
        q(x, attention_mask=m, position_ids=torch.stack([pos[0], pos[0] + 1]))


def test_gqa_with_let_is_refused_and_gqa_lwc_runs():
    """Grouped-query attention: LET pairs q/k/v/o rows one to one (models/transformation.py:44-69) and the reference
    dies on a broadcast error; the HIP path must be as loud (no out-of-bounds vector reads).  LWC-only GQA runs."""
    from omniquant_amd.synthetic import make_calib_inputs, causal_mask
    _, _, _, q = _tiny_block(True, kv_heads=1)
    with pytest.raises(NotImplementedError):
        q.smooth_and_quant_temporary()
    _, _, _, q2 = _tiny_block(False, kv_heads=1, abits=16)
    q2.compute_dtype = torch.float32
    x = make_calib_inputs(1, 64, 256, dtype=torch.float32).to(DEV)
    q2.smooth_and_quant_temporary()
    y = q2(x, attention_mask=causal_mask(64, DEV), position_ids=torch.arange(64, device=DEV)[None])[0]
    y.float().pow(2).mean().backward()
    assert bool(torch.isfinite(y).all()) and q2.self_attn.k_proj.weight_quantizer.upbound_factor.grad is not None


def test_w16_weights_pass_through_with_let_vs_oracle():
    """--wbits 16 --abits 4 --let: the weight quantizer is the identity (quantize/quantizer.py:109-110), LET still
    re-parameterises the weights; one sample-step (loss + LET gradients) vs the CPU oracle, fold keeps scales None."""
    from oracle import ref_cpu as R
    from omniquant_amd.synthetic import make_calib_inputs, causal_mask, synth_act_stats
    cfg, args, layer, q = _tiny_block(True, wbits=16, abits=4, seed=9)
    weights = {n: p.detach().float().cpu() for n, p in layer.named_parameters()}
    q.compute_dtype = torch.float32
    T = 32
    x = make_calib_inputs(1, T, 256, dtype=torch.float32)
    tgt = make_calib_inputs(1, T, 256, seed=4, dtype=torch.float32)
    mask, pos = causal_mask(T), torch.arange(T)[None]
    cd = dict(hidden_size=256, num_attention_heads=2, num_key_value_heads=2, rms_norm_eps=1e-6)
    blk = R.Block("llama", cd, weights, R.QuantSpec(16, 4, None, True, True), max_pos=T)
    sc, sh = synth_act_stats(cfg, 1)
    blk.register_let(sc, sh, 0.5, 0, "model.layers")
    out_o = blk.forward(x, mask, pos, temps=blk.temporaries(), act_quant=True)
    loss_o = torch.nn.functional.mse_loss(tgt, out_o)
    loss_o.backward()
    q.smooth_and_quant_temporary()
    out = q(x.to(DEV), attention_mask=mask.to(DEV), position_ids=pos.to(DEV))[0]
    loss = torch.nn.functional.mse_loss(tgt.to(DEV), out)
    loss.backward()
    assert abs(float(loss) - float(loss_o)) <= 1e-3 * float(loss_o)
    for n, p in q.named_parameters():
        ref = blk.params[n].grad
        if "bound_factor" in n:
            assert ref is None and (p.grad is None or float(p.grad.abs().max()) == 0.0), n
            continue
        den = float(ref.norm()) + 1e-20
        assert float((p.grad.cpu() - ref).norm()) / den < 2e-2, n
    q.clear_temp_variable()
    q.smooth_and_quant_inplace()
    q.register_scales_and_zeros()
    assert q.mlp.down_proj.weight_quantizer.scales is None


def test_opt_block_w4a4_let_at_opt125m_width_vs_oracle():
    """OPT decoder block at OPT-125m width (768 / 3072 / 12 heads), W4A4 + LET, fp32 parity mode: here the LayerNorm ->
    input-quantiser pairs run the fused oq_norm_quant kernels (hidden >= 512) with the sibling-gradient side channel.
    One sample-step vs the CPU oracle: loss and every LWC / LET gradient."""
    from oracle import ref_cpu as R
    from omniquant_amd.calibrate import default_args, register_let_parameters
    from omniquant_amd.synthetic import make_config, make_layer, make_calib_inputs, causal_mask, synth_act_stats
    from omniquant_amd.opt_block import QuantOPTDecoderLayer
    from omniquant_amd import ops
    cfg = make_config("opt-125m")
    args = default_args(wbits=4, abits=4, lwc=True, let=True, epochs=1, nsamples=1, net="opt-125m")
    layer = make_layer(cfg, seed=4, device=DEV)
    weights = {n: p.detach().float().cpu() for n, p in layer.named_parameters()}
    q = QuantOPTDecoderLayer(cfg, layer, args).to(DEV)
    q.compute_dtype = torch.float32
    q.set_quant_state(False, True)
    q.let = True
    sc, sh = synth_act_stats(cfg, 1)
    register_let_parameters(q, "opt", sc, sh, 0.5, 0, DEV)
    with torch.no_grad():
        for p_ in q.parameters():
            p_.data = p_.data.float()
    T = 64
    x = make_calib_inputs(1, T, 768, dtype=torch.float32)
    tgt = make_calib_inputs(1, T, 768, seed=9, dtype=torch.float32)
    mask = causal_mask(T)
    cd = dict(hidden_size=768, num_attention_heads=12)
    blk = R.Block("opt", cd, weights, R.QuantSpec(4, 4, None, True, True), max_pos=T)
    blk.register_let(sc, sh, 0.5, 0, "model.decoder.layers")
    out_o = blk.forward(x, mask, None, temps=blk.temporaries(), act_quant=True)
    loss_o = torch.nn.functional.mse_loss(tgt, out_o)
    loss_o.backward()
    calls = {"n": 0}
    orig = ops.NormQuantFn.apply

    class _Spy:
        @staticmethod
        def apply(*a):
            calls["n"] += 1
            return orig(*a)
    ops.NormQuantFn, keep = _Spy, ops.NormQuantFn
    try:
        q.smooth_and_quant_temporary()
        out = q(x.to(DEV), attention_mask=mask.to(DEV))[0]
        loss = torch.nn.functional.mse_loss(tgt.to(DEV), out)
        loss.backward()
    finally:
        ops.NormQuantFn = keep
    assert calls["n"] == 2, "both LayerNorm -> quantiser pairs must take the fused kernels"
    assert abs(float(loss) - float(loss_o)) <= 2e-3 * float(loss_o)
    for n, p_ in q.named_parameters():
        ref = blk.params[n].grad
        rel = float((p_.grad.cpu() - ref).norm()) / (float(ref.norm()) + 1e-20)
        assert rel < 3e-2, (n, rel)          # a few 4-bit activation rounding ties may flip between the two fp32 runs
