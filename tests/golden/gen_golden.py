#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

Imports the reference's own hot-path modules (quantize.quantizer, quantize.int_linear,
quantize.int_matmul, quantize.omni_norm, models.transformation, models.int_llama_layer,
models.int_opt_layer) on CPU / fp32 and records input -> output vectors as small .npz
fixtures next to this file.  Nothing under tests/, bench.py or smoke() ever reads
/root/reference; they only read the .npz files written here.

The reference's engine (quantize/omniquant.py) cannot run verbatim on CPU,
so the per-block sequence of quantize/omniquant.py:165-250 is driven here step by step
with fp32 tensors: teacher pass -> LET init -> AdamW loop -> clear_temp_variable ->
smooth_and_quant_inplace -> propagate -> register_scales_and_zeros -> half -> omni_state_dict.

Generator-side shims (written here, not reference code):
  * transformers>=5 LlamaAttention has no `rotary_emb` and a 4-arg apply_rotary_pos_emb;
    the reference block (models/int_llama_layer.py:70,124-125) expects the 4.31-era API.
    `_OldRotary` + `_old_apply_rotary` restate that published formula.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [act_stats | round2]
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

torch.set_num_threads(4)
torch.use_deterministic_algorithms(False)

from quantize.quantizer import UniformAffineQuantizer  # noqa: E402
from quantize.int_linear import QuantLinear  # noqa: E402,F401
import models.int_llama_layer as ref_llama  # noqa: E402
import models.int_opt_layer as ref_opt  # noqa: E402
import models.transformation as ref_tf  # noqa: E402

from transformers.models.llama.configuration_llama import LlamaConfig  # noqa: E402
from transformers.models.llama import modeling_llama as hf_llama  # noqa: E402
from transformers.models.opt.configuration_opt import OPTConfig  # noqa: E402
from transformers.models.opt import modeling_opt as hf_opt  # noqa: E402


# --------------------------------------------------------------------------------------
# shims
# --------------------------------------------------------------------------------------
class _OldRotary(torch.nn.Module):
    """cos/sin cache with the transformers-4.31 call signature rotary(x, seq_len=...)."""

    def __init__(self, dim, max_pos=2048, base=10000.0):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        t = torch.arange(max_pos, dtype=torch.float32)
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        self.register_buffer("cos_cached", emb.cos()[None, None, :, :], persistent=False)
        self.register_buffer("sin_cached", emb.sin()[None, None, :, :], persistent=False)

    def forward(self, x, seq_len=None):
        return (self.cos_cached[:, :, :seq_len, ...].to(dtype=x.dtype),
                self.sin_cached[:, :, :seq_len, ...].to(dtype=x.dtype))


def _rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def _old_apply_rotary(q, k, cos, sin, position_ids):
    cos = cos.squeeze(1).squeeze(0)[position_ids].unsqueeze(1)
    sin = sin.squeeze(1).squeeze(0)[position_ids].unsqueeze(1)
    return (q * cos) + (_rotate_half(q) * sin), (k * cos) + (_rotate_half(k) * sin)


ref_llama.apply_rotary_pos_emb = _old_apply_rotary


def np32(t):
    return t.detach().to(torch.float32).cpu().clone().numpy()   # clone: never alias live tensors


def save(name, arrays, meta=None):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.ascontiguousarray(v)
    if meta is not None:
        out["__meta__"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB, {len(out)} arrays")


# --------------------------------------------------------------------------------------
# G1: UniformAffineQuantizer forward/backward (quantize/quantizer.py:84-147)
# --------------------------------------------------------------------------------------
def gen_quantizer():
    g = torch.Generator().manual_seed(0)
    cases = []
    arrays = {}

    def run_case(tag, x, n_bits, group_size=None, lwc=False, symmetric=False,
                 dynamic_method="per_channel", rand_bounds=True, need_gx=True):
        x = x.clone().requires_grad_(need_gx)
        q = UniformAffineQuantizer(n_bits=n_bits, symmetric=symmetric, per_channel_axes=[0],
                                   dynamic_method=dynamic_method, group_size=group_size,
                                   shape=tuple(x.shape) if lwc else None, lwc=lwc)
        if lwc and rand_bounds:
            with torch.no_grad():
                q.upbound_factor.copy_(4.0 + 1.5 * torch.randn(q.upbound_factor.shape, generator=g))
                q.lowbound_factor.copy_(4.0 + 1.5 * torch.randn(q.lowbound_factor.shape, generator=g))
        y = q(x)
        G = torch.randn(y.shape, generator=g)
        if y.requires_grad:
            (y * G).sum().backward()
        i = len(cases)
        arrays[f"c{i}_x"] = np32(x)
        arrays[f"c{i}_G"] = np32(G)
        arrays[f"c{i}_y"] = np32(y)
        arrays[f"c{i}_scale"] = np32(q.scale)
        arrays[f"c{i}_zp"] = np32(q.round_zero_point)
        if need_gx:
            arrays[f"c{i}_gx"] = np32(x.grad)
        if lwc:
            arrays[f"c{i}_up"] = np32(q.upbound_factor)
            arrays[f"c{i}_low"] = np32(q.lowbound_factor)
            arrays[f"c{i}_gup"] = np32(q.upbound_factor.grad)
            arrays[f"c{i}_glow"] = np32(q.lowbound_factor.grad)
        cases.append(dict(tag=tag, n_bits=n_bits, group_size=group_size, lwc=lwc,
                          symmetric=symmetric, dynamic_method=dynamic_method,
                          shape=list(x.shape), deficiency=q.deficiency))

    # weight-style, per-channel and grouped, lwc on/off, several bit widths
    for n_bits in (2, 3, 4, 6):
        for group in (None, 64, 128):
            x = torch.randn(8, 256, generator=g) * 0.05
            x[:, 7] *= 12.0  # outlier column exercises clipping
            run_case(f"w{n_bits}g{group}_lwc", x, n_bits, group, lwc=True)
    run_case("w4_nolwc", torch.randn(16, 512, generator=g), 4, None, lwc=False)
    run_case("w3g128_nolwc", torch.randn(16, 512, generator=g), 3, 128, lwc=False)
    run_case("w4_lwc_init4", torch.randn(16, 512, generator=g) * 0.02, 4, None, lwc=True, rand_bounds=False)
    run_case("w8_lwc", torch.randn(8, 256, generator=g), 8, None, lwc=True)

    # ties: duplicated max / min inside a row and inside a group (amax/amin split grads)
    x = torch.randn(8, 256, generator=g)
    x[0, 3] = x[0, 200] = 5.0
    x[0, 9] = x[0, 100] = x[0, 101] = -4.0
    x[1, 0] = x[1, 64] = x[1, 128] = 3.5
    run_case("ties_w4", x, 4, None, lwc=True)
    run_case("ties_w4g64", x, 4, 64, lwc=True)

    # exact half-way rounding (round-half-to-even): range 15 -> scale 1 with 4 bits
    x = torch.zeros(4, 32)
    x[:, 0] = 0.0
    x[:, 1] = 15.0
    x[:, 2:] = torch.arange(30).float().view(1, -1) * 0.5
    run_case("halfway_w4", x, 4, None, lwc=False)

    # symmetric (quantize/quantizer.py:136-140), incl. deficiency padding (:65-69,85-87)
    run_case("sym_w4", torch.randn(8, 256, generator=g), 4, None, lwc=True, symmetric=True)
    run_case("sym_w4g128", torch.randn(8, 256, generator=g), 4, 128, lwc=True, symmetric=True)
    run_case("sym_w4g96_deficient", torch.randn(8, 256, generator=g), 4, 96, lwc=True, symmetric=True)

    # activation-style: per-token on [bs,T,K] and head-wise on [1,4,16,32]
    for n_bits in (4, 6, 8):
        run_case(f"a{n_bits}_tok", torch.randn(2, 16, 64, generator=g) * 3, n_bits, None,
                 dynamic_method="per_token")
    run_case("a4_head", torch.randn(1, 4, 16, 32, generator=g), 4, None, dynamic_method="per_token")
    run_case("a4_tok_positive", torch.rand(1, 8, 64, generator=g) + 100.0, 4, None,
             dynamic_method="per_token")

    # constant row: scale==0 -> NaN (quirk Q1, quantize/quantizer.py:144-145)
    x = torch.randn(4, 64, generator=g)
    x[2, :] = 0.75
    run_case("const_row_nan", x, 4, None, dynamic_method="per_token", need_gx=False)

    # pass-through gates (quantize/quantizer.py:109-110)
    q16 = UniformAffineQuantizer(n_bits=16, metric="fix0to1")
    xx = torch.randn(3, 5, generator=g)
    assert q16(xx) is xx
    save("g1_quantizer.npz", arrays, meta=dict(cases=cases))


# --------------------------------------------------------------------------------------
# block builders
# --------------------------------------------------------------------------------------
def make_args(wbits, abits, group_size, lwc, let, symmetric=False):
    a = types.SimpleNamespace()
    a.weight_quant_params = {"n_bits": wbits, "per_channel_axes": [0], "symmetric": symmetric,
                             "dynamic_method": "per_channel", "group_size": group_size, "lwc": lwc}
    act = {"n_bits": abits, "per_channel_axes": [], "symmetric": False, "dynamic_method": "per_token"}
    a.act_quant_params = dict(act)
    a.q_quant_params = dict(act)
    a.k_quant_params = dict(act)
    a.v_quant_params = dict(act)
    a.p_quant_params = {"n_bits": 16, "metric": "fix0to1"}
    a.let, a.lwc = let, lwc
    return a


LLAMA_TINY = dict(hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=4,
                  num_hidden_layers=2, vocab_size=128, max_position_embeddings=64, rms_norm_eps=1e-6)
OPT_TINY = dict(hidden_size=64, ffn_dim=256, num_attention_heads=4, num_hidden_layers=2, vocab_size=128,
                max_position_embeddings=64, word_embed_proj_dim=64)


def llama_layer(cfg, seed):
    torch.manual_seed(seed)
    layer = hf_llama.LlamaDecoderLayer(cfg, 0)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if "layernorm" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) * 0.08)
        # outlier input channels (what LET is for)
        for lin in (layer.self_attn.q_proj, layer.self_attn.k_proj, layer.self_attn.v_proj,
                    layer.mlp.up_proj, layer.mlp.gate_proj):
            lin.weight[:, 5] *= 6.0
    layer = layer.half()
    hd = cfg.hidden_size // cfg.num_attention_heads
    layer.self_attn.rotary_emb = _OldRotary(hd, cfg.max_position_embeddings)
    return layer


def opt_layer(cfg, seed):
    torch.manual_seed(seed)
    layer = hf_opt.OPTDecoderLayer(cfg)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if "layer_norm.weight" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif "bias" in n:
                p.copy_(torch.randn_like(p) * 0.05)
            else:
                p.copy_(torch.randn_like(p) * 0.08)
    return layer.half()


def causal_mask(T, dtype=torch.float32):
    m = torch.full((T, T), torch.finfo(dtype).min, dtype=dtype)
    m = torch.triu(m, diagonal=1)
    return m[None, None]


def layer_weights(layer, prefix):
    return {f"{prefix}{n}": p.detach().cpu().numpy() for n, p in layer.named_parameters()}


def let_linears(is_llama):
    if is_llama:
        return {"q_proj": "qkv", "o_proj": "out", "up_proj": "fc1"}
    return {"q_proj": "qkv", "out_proj": "out", "fc1": "fc1"}


def register_let(qlayer, is_llama, act_scales, act_shifts, alpha, layer_idx, prefix):
    """quantize/omniquant.py:182-197 driven by hand (fp32)."""
    pairs = let_linears(is_llama)
    dtype = torch.float32
    qlayer.register_parameter("qkt_smooth_scale", torch.nn.Parameter(
        torch.ones(qlayer.self_attn.q_proj.out_features, dtype=dtype)))
    for name, module in qlayer.named_modules():
        if isinstance(module, QuantLinear):
            for key in pairs.keys():
                if key in name:
                    act = act_scales[f"{prefix}.{layer_idx}.{name}"].to(dtype=dtype).clamp(min=1e-5)
                    weight = module.weight.max(dim=0)[0].clamp(min=1e-5)
                    scale = (act.pow(alpha) / weight.pow(1 - alpha)).clamp(min=1e-5)
                    if not is_llama:
                        shift = act_shifts[f"{prefix}.{layer_idx}.{name}"].to(dtype=dtype)
                    else:
                        shift = torch.zeros_like(scale)
                    qlayer.register_parameter(f"{pairs[key]}_smooth_shift", torch.nn.Parameter(shift))
                    qlayer.register_parameter(f"{pairs[key]}_smooth_scale", torch.nn.Parameter(scale))


def synth_act_stats(is_llama, cfg, n_layers, prefix, g):
    H = cfg.hidden_size
    scales, shifts = {}, {}
    for i in range(n_layers):
        for name in let_linears(is_llama).keys():
            full = ("self_attn." + name) if name in ("q_proj", "o_proj", "out_proj") else (
                "mlp." + name if is_llama else name)
            s = (torch.rand(H, generator=g) * 3 + 0.2)
            s[5] = 25.0
            scales[f"{prefix}.{i}.{full}"] = s
            shifts[f"{prefix}.{i}.{full}"] = torch.randn(H, generator=g) * 0.3
    return scales, shifts


def build_qlayer(is_llama, cfg, layer, args):
    if is_llama:
        q = ref_llama.QuantLlamaDecoderLayer(cfg, layer, args)
    else:
        q = ref_opt.QuantOPTDecoderLayer(cfg, layer, args)
    return q


def fwd(qlayer, x, mask, pos, is_llama):
    if is_llama:
        return qlayer(x, attention_mask=mask, position_ids=pos)[0]
    return qlayer(x, attention_mask=mask)[0]


# --------------------------------------------------------------------------------------
# G2 + G3: LET temporaries and one full block step (fwd, loss, every grad)
# --------------------------------------------------------------------------------------
def gen_block_step(is_llama, tag, wbits, abits, group, lwc, let, T=16, randomize=True, cfg_over=None, save_tmp=True):
    fam = "llama" if is_llama else "opt"
    cfg_dict = dict(LLAMA_TINY if is_llama else OPT_TINY, **(cfg_over or {}))
    cfg = LlamaConfig(**cfg_dict) if is_llama else OPTConfig(**cfg_dict)
    prefix = "model.layers" if is_llama else "model.decoder.layers"
    g = torch.Generator().manual_seed(11)
    layer = llama_layer(cfg, 3) if is_llama else opt_layer(cfg, 3)
    arrays = layer_weights(layer, "w.")
    args = make_args(wbits, abits, group, lwc, let)
    qlayer = build_qlayer(is_llama, cfg, layer, args)
    qlayer.set_quant_state(weight_quant=False, act_quant=True)
    qlayer.let = let
    act_scales, act_shifts = synth_act_stats(is_llama, cfg, 1, prefix, g)
    if let:
        register_let(qlayer, is_llama, act_scales, act_shifts, 0.5, 0, prefix)
        for k, v in act_scales.items():
            arrays["act_scales." + k] = np32(v)
        for k, v in act_shifts.items():
            arrays["act_shifts." + k] = np32(v)
    with torch.no_grad():
        qlayer.float()
        if randomize:
            for n, p in qlayer.named_parameters():
                if "bound_factor" in n:
                    p.copy_(4.0 + 1.0 * torch.randn(p.shape, generator=g))
                elif "smooth_shift" in n:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.2)
                elif "smooth_scale" in n:
                    p.mul_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            if let:
                # values under the 1e-2 truncation threshold (models/transformation.py:5-20)
                qlayer.qkt_smooth_scale[3] = 0.004
                qlayer.qkt_smooth_scale[4] = -0.003
                qlayer.fc1_smooth_scale[9] = 0.001
    for n, p in qlayer.named_parameters():
        arrays["p0." + n] = np32(p)

    x = torch.randn(1, T, cfg.hidden_size, generator=g)
    x[..., 5] *= 8.0
    target = torch.randn(1, T, cfg.hidden_size, generator=g)
    mask = causal_mask(T)
    pos = torch.arange(T)[None]
    arrays.update(x=np32(x), target=np32(target), mask=np32(mask), position_ids=pos.numpy())

    qlayer.smooth_and_quant_temporary()
    # G2: temporaries (left out of the larger fixtures: they would dominate the file)
    for n, m in qlayer.named_modules():
        if not save_tmp:
            break
        if hasattr(m, "temp_weight") and m.temp_weight is not None:
            arrays["tmp." + n + ".temp_weight"] = np32(m.temp_weight)
        if hasattr(m, "temp_bias") and getattr(m, "temp_bias") is not None:
            arrays["tmp." + n + ".temp_bias"] = np32(m.temp_bias)
    for n, p in qlayer.named_parameters():
        arrays["p_trunc." + n] = np32(p)   # after in-place truncate_number
    out = fwd(qlayer, x, mask, pos, is_llama)
    loss = torch.nn.functional.mse_loss(target, out)
    loss.backward()
    arrays["out"] = np32(out)
    arrays["loss"] = np32(loss)
    for n, p in qlayer.named_parameters():
        arrays["grad." + n] = np32(p.grad) if p.grad is not None else np.zeros(0, np.float32)
    # fp (quant off) forward, for the teacher path
    qlayer.clear_temp_variable()
    for m in qlayer.modules():
        if hasattr(m, "use_temporary_parameter"):
            m.use_temporary_parameter = False
    qlayer.set_quant_state(weight_quant=False, act_quant=False)
    with torch.no_grad():
        arrays["out_fp"] = np32(fwd(qlayer, x, mask, pos, is_llama))
    meta = dict(family=fam, wbits=wbits, abits=abits, group_size=group, lwc=lwc, let=let, T=T,
                config=cfg_dict, alpha=0.5, layer_prefix=prefix)
    save(f"g3_step_{fam}_{tag}.npz", arrays, meta)


# --------------------------------------------------------------------------------------
# G4: learned-parameter trajectory, 2 layers x 4 samples x 2 epochs
# --------------------------------------------------------------------------------------
def gen_trajectory(is_llama, tag, wbits, abits, group, lwc, let, aug_loss=False,
                   let_lr=5e-3, lwc_lr=1e-2, alpha=0.5, T=16, nsamples=4, epochs=2, n_layers=2, cfg_over=None,
                   batch_size=1):
    fam = "llama" if is_llama else "opt"
    cfg_dict = dict(LLAMA_TINY if is_llama else OPT_TINY, **(cfg_over or {}))
    cfg = LlamaConfig(**cfg_dict) if is_llama else OPTConfig(**cfg_dict)
    prefix = "model.layers" if is_llama else "model.decoder.layers"
    g = torch.Generator().manual_seed(23)
    layers = [llama_layer(cfg, 100 + i) if is_llama else opt_layer(cfg, 100 + i) for i in range(n_layers)]
    arrays = {}
    for i, l in enumerate(layers):
        arrays.update(layer_weights(l, f"w{i}."))
    act_scales, act_shifts = synth_act_stats(is_llama, cfg, n_layers, prefix, g)
    for k, v in act_scales.items():
        arrays["act_scales." + k] = np32(v)
    for k, v in act_shifts.items():
        arrays["act_shifts." + k] = np32(v)
    args = make_args(wbits, abits, group, lwc, let)
    inps = torch.randn(nsamples, T, cfg.hidden_size, generator=g)
    inps[..., 5] *= 8.0
    arrays["inps"] = np32(inps)
    mask = causal_mask(T)
    pos = torch.arange(T)[None]
    arrays.update(mask=np32(mask), position_ids=pos.numpy())

    quant_inps = inps.clone()
    fp_inps = inps.clone()
    fp_inps_2 = inps.clone() if aug_loss else None
    loss_func = torch.nn.MSELoss()
    losses, norms = [], []
    for i in range(n_layers):
        qlayer = build_qlayer(is_llama, cfg, layers[i], args)
        with torch.no_grad():
            qlayer.float()   # CPU has no autocast: teacher pass in fp32
        qlayer.set_quant_state(weight_quant=False, act_quant=False)
        with torch.no_grad():
            for j in range(nsamples):
                fp_inps[j] = fwd(qlayer, fp_inps[j].unsqueeze(0), mask, pos, is_llama)[0]
                if aug_loss:
                    fp_inps_2[j] = fwd(qlayer, quant_inps[j].unsqueeze(0), mask, pos, is_llama)[0]
        arrays[f"fp_out.{i}"] = np32(fp_inps)
        qlayer.set_quant_state(weight_quant=False, act_quant=True)
        qlayer.let = let
        if let:
            register_let(qlayer, is_llama, act_scales, act_shifts, alpha, i, prefix)
        for n, p in qlayer.named_parameters():
            arrays[f"init.{i}.{n}"] = np32(p)
        opt = torch.optim.AdamW(
            [{"params": qlayer.let_parameters(True), "lr": let_lr},
             {"params": qlayer.lwc_parameters(), "lr": lwc_lr}], weight_decay=0.0)
        bs = batch_size
        mask_b = mask.repeat(bs, 1, 1, 1) if bs > 1 else mask       # quantize/omniquant.py:139-141 (attention_mask_batch)
        for ep in range(epochs):
            for j in range(0, nsamples // bs * bs, bs):               # quantize/omniquant.py:214-219
                qlayer.smooth_and_quant_temporary()
                out = fwd(qlayer, quant_inps[j:j + bs], mask_b, pos, is_llama)
                loss = loss_func(fp_inps[j:j + bs], out)
                if aug_loss:
                    loss = loss + loss_func(fp_inps_2[j:j + bs], out)
                losses.append(float(loss))
                opt.zero_grad()
                loss.backward()
                params = [p for p in qlayer.omni_parameters(True) if p.grad is not None]
                norm = torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in params]), 2.0)
                norms.append(float(norm))
                opt.step()
        qlayer.clear_temp_variable()
        for n, p in qlayer.named_parameters():
            arrays[f"trained32.{i}.{n}"] = np32(p)
        qlayer.smooth_and_quant_inplace()
        with torch.no_grad():
            for j in range(nsamples):
                quant_inps[j] = fwd(qlayer, quant_inps[j].unsqueeze(0), mask, pos, is_llama)[0]
        arrays[f"quant_out.{i}"] = np32(quant_inps)
        qlayer.register_scales_and_zeros()
        for n, b in qlayer.named_buffers():
            if n.endswith("weight") or n.endswith("bias") or n.endswith("scales") or n.endswith("zeros"):
                arrays[f"folded32.{i}.{n}"] = np32(b)
        qlayer.half()
        sd = qlayer.omni_state_dict()
        for n, p in sd.items():
            assert p.dtype == torch.float16
            arrays[f"omni.{i}.{n}"] = p.detach().cpu().numpy()
    arrays["losses"] = np.asarray(losses, np.float64)
    arrays["norms"] = np.asarray(norms, np.float64)
    meta = dict(family=fam, wbits=wbits, abits=abits, group_size=group, lwc=lwc, let=let, aug_loss=aug_loss,
                let_lr=let_lr, lwc_lr=lwc_lr, alpha=alpha, T=T, nsamples=nsamples, epochs=epochs,
                n_layers=n_layers, config=cfg_dict, layer_prefix=prefix, batch_size=batch_size)
    save(f"g4_traj_{fam}_{tag}.npz", arrays, meta)


# --------------------------------------------------------------------------------------
# G5: truncate_number + LET-init formula in isolation
# --------------------------------------------------------------------------------------
def gen_misc():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, generator=g) * 0.02
    x[0], x[1], x[2] = 0.0, 0.00999, -0.01
    y = ref_tf.truncate_number(x.clone())
    xg = x.clone().requires_grad_(True)
    yg = ref_tf.truncate_number(xg)
    G = torch.randn(64, generator=g)
    (yg * G).sum().backward()
    act = torch.rand(32, generator=g) * 4
    act[0] = 0.0
    W = torch.randn(48, 32, generator=g) * 0.1
    W[:, 1] = -W[:, 1].abs()      # all-negative column: signed max < 0 -> clamp(1e-5) (quirk Q4)
    arrays = dict(trunc_x=x, trunc_y=y, trunc_G=G, trunc_gx=xg.grad, let_act=act, let_W=W)
    for alpha in (0.5, 0.75):
        a = act.clamp(min=1e-5)
        w = W.max(dim=0)[0].clamp(min=1e-5)
        arrays[f"let_scale_a{alpha}"] = (a.pow(alpha) / w.pow(1 - alpha)).clamp(min=1e-5)
    save("g5_misc.npz", arrays)


def gen_act_stats():
    """G6: the reference's own get_act_scales / get_act_shifts (generate_act_scale_shift.py:25-94) driven over a
    tiny two-linear model and 5 'samples'; fixtures hold every hooked linear input and the resulting dicts."""
    import generate_act_scale_shift as ref_gen      # the reference's pre-pass script (its hooks, its arithmetic)
    g = torch.Generator().manual_seed(11)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = torch.nn.Linear(24, 40)
            self.fc2 = torch.nn.Linear(40, 16)

        def forward(self, x):
            return self.fc2(torch.relu(self.fc1(x)) - 0.3)

    model = Tiny()
    with torch.no_grad():
        for p_ in model.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * 0.4)
    xs = [torch.randn(1, 33, 24, generator=g) * (1 + i) for i in range(5)]
    xs[2][0, 5, 3] = 9.5            # a large positive outlier in one sample
    xs[3][0, 7, 4] = -12.0          # and a large negative one (abs-max must catch it)
    dataloader = [(x,) for x in xs]
    scales = ref_gen.get_act_scales(model, dataloader, num_samples=5)
    shifts = ref_gen.get_act_shifts(model, dataloader, num_samples=5)
    with torch.no_grad():
        h = [torch.relu(model.fc1(x)) - 0.3 for x in xs]
    arrays = dict(x_fc1=torch.cat(xs), x_fc2=torch.cat(h))
    for k in ("fc1", "fc2"):
        arrays[f"scale_{k}"] = scales[k]
        arrays[f"shift_{k}"] = shifts[k]
    save("g6_act_stats.npz", arrays)


def gen_round2():
    """Round-2 additions: grouped-query attention (models/int_llama_layer.py:138-139 repeat_kv; the LLaMA-2-70B shape
    class, LWC only -- LET cannot pair q/k/v rows under GQA) and one block step at head_dim 128 / T 256, the smallest
    shape on which the HIP path's causal tile-skipping GEMM modes and fused attention kernels run."""
    gqa = dict(num_key_value_heads=2)
    gen_block_step(True, "gqa_w4a4_lwc", 4, 4, None, True, False, cfg_over=gqa)
    gen_trajectory(True, "gqa_w2a16g16_lwc", 2, 16, 16, True, False, cfg_over=gqa)
    hd128 = dict(hidden_size=256, intermediate_size=512, num_attention_heads=2, num_key_value_heads=2,
                 max_position_embeddings=256)
    gen_block_step(True, "hd128_w4a4_lwc_let", 4, 4, None, True, True, T=256, cfg_over=hd128, save_tmp=False)


def gen_round2b():
    """--batch_size 2 (quantize/omniquant.py:139-141,214-219: two samples per step, the mask repeated per sample, the MSE
    a mean over both): one LLaMA W4A4 + LET and one OPT W4A16 trajectory."""
    gen_trajectory(True, "w4a4_lwc_let_bs2", 4, 4, None, True, True, batch_size=2)
    gen_trajectory(False, "w4a16_lwc_bs2", 4, 16, None, True, False, batch_size=2)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "round2b":
        gen_round2b()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "act_stats":
        gen_act_stats()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        gen_round2()
        sys.exit(0)
    gen_quantizer()
    gen_misc()
    # (family, tag, wbits, abits, group, lwc, let)
    gen_block_step(True, "w4a4_lwc_let", 4, 4, None, True, True)
    gen_block_step(True, "w3a16g32_lwc", 3, 16, 32, True, False)
    gen_block_step(True, "w4a16_lwc", 4, 16, None, True, False)
    gen_block_step(False, "w4a4_lwc_let", 4, 4, None, True, True)
    gen_block_step(False, "w4a16_lwc", 4, 16, None, True, False)
    gen_trajectory(True, "w4a4_lwc_let", 4, 4, None, True, True)
    gen_trajectory(True, "w4a4_lwc_let_aug", 4, 4, None, True, True, aug_loss=True)
    gen_trajectory(True, "w3a16g32_lwc", 3, 16, 32, True, False)
    gen_trajectory(True, "w2a16g16_lwc_aug", 2, 16, 16, True, False, aug_loss=True)
    gen_trajectory(False, "w4a16_lwc", 4, 16, None, True, False)
    gen_trajectory(False, "w4a4_lwc_let", 4, 4, None, True, True)
    gen_act_stats()
    gen_round2()
    gen_round2b()
