"""Random-init decoder layers and calibration inputs of a named architecture (no network, no checkpoints).

The blocks only need an object exposing the HF layer attribute names (self_attn.q_proj ... input_layernorm), so
these light stand-ins avoid a transformers dependency on the hot path.  Used by bench.py, smoke() and tests.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

ARCH = {
    # name: (family, hidden, intermediate/ffn, heads, kv_heads, layers)
    "llama-tiny": ("llama", 64, 128, 4, 4, 2),
    "llama-7b": ("llama", 4096, 11008, 32, 32, 32),
    "llama-2-13b": ("llama", 5120, 13824, 40, 40, 40),
    "llama-2-70b": ("llama", 8192, 28672, 64, 8, 80),
    "opt-tiny": ("opt", 64, 256, 4, 4, 2),
    "opt-125m": ("opt", 768, 3072, 12, 12, 12),
}


def make_config(name=None, **over):
    if name is not None:
        fam, H, I, nh, nkv, L = ARCH[name]
    else:
        fam = over.pop("family")
        H, I, nh, nkv, L = over.pop("hidden_size"), over.pop("inter"), over.pop("heads"), over.pop("kv_heads"), over.pop("layers", 1)
    if fam == "llama":
        cfg = SimpleNamespace(family="llama", hidden_size=H, intermediate_size=I, num_attention_heads=nh,
                              num_key_value_heads=nkv, num_hidden_layers=L, hidden_act="silu",
                              max_position_embeddings=4096, rms_norm_eps=1e-6, rope_theta=10000.0, use_cache=False)
    else:
        cfg = SimpleNamespace(family="opt", hidden_size=H, ffn_dim=I, num_attention_heads=nh, num_hidden_layers=L,
                              attention_dropout=0.0, dropout=0.1, enable_bias=True, do_layer_norm_before=True,
                              max_position_embeddings=2048, use_cache=False)
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


class _RMSNorm(nn.Module):
    def __init__(self, H, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(H))
        self.variance_epsilon = eps


class LlamaLayerStub(nn.Module):
    """Same attribute names as transformers' LlamaDecoderLayer (what models/int_llama_layer.py:33-41,72-95 read)."""

    def __init__(self, cfg):
        super().__init__()
        H, I = cfg.hidden_size, cfg.intermediate_size
        hd = H // cfg.num_attention_heads
        kvd = hd * cfg.num_key_value_heads
        self.self_attn = nn.Module()
        self.self_attn.q_proj = nn.Linear(H, H, bias=False)
        self.self_attn.k_proj = nn.Linear(H, kvd, bias=False)
        self.self_attn.v_proj = nn.Linear(H, kvd, bias=False)
        self.self_attn.o_proj = nn.Linear(H, H, bias=False)
        self.mlp = nn.Module()
        self.mlp.gate_proj = nn.Linear(H, I, bias=False)
        self.mlp.up_proj = nn.Linear(H, I, bias=False)
        self.mlp.down_proj = nn.Linear(I, H, bias=False)
        self.input_layernorm = _RMSNorm(H, cfg.rms_norm_eps)
        self.post_attention_layernorm = _RMSNorm(H, cfg.rms_norm_eps)


class OPTLayerStub(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H, F = cfg.hidden_size, cfg.ffn_dim
        self.self_attn = nn.Module()
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            setattr(self.self_attn, n, nn.Linear(H, H, bias=True))
        self.self_attn_layer_norm = nn.LayerNorm(H, eps=1e-5)
        self.fc1 = nn.Linear(H, F, bias=True)
        self.fc2 = nn.Linear(F, H, bias=True)
        self.final_layer_norm = nn.LayerNorm(H, eps=1e-5)


def make_layer(cfg, seed=0, device="cpu", std=0.02, outliers=True, weights=None, dtype=torch.float16):
    """Random-init layer (SURVEY.md 8d synthetic recipe): weights N(0, std) in fp16 with 0.1 % outlier input
    columns x20; norm weights 1 + 0.1 N(0,1).  `weights` (dict name -> tensor) overrides the random init."""
    layer = LlamaLayerStub(cfg) if cfg.family == "llama" else OPTLayerStub(cfg)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if weights is not None:
                p.copy_(weights[n].to(p.dtype))
                continue
            if "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * std)
                if outliers and p.dim() == 2:
                    k = max(1, p.shape[1] // 1000)
                    idx = torch.randperm(p.shape[1], generator=g)[:k]
                    p[:, idx] *= 20.0
    layer = layer.to(dtype)
    return layer.to(device)


def make_calib_inputs(nsamples, T, H, seed=1, device="cpu", dtype=torch.float16):
    """randn activations with per-channel scale exp(0.5 N(0,1))."""
    g = torch.Generator().manual_seed(seed)
    ch = torch.exp(0.5 * torch.randn(H, generator=g))
    x = torch.randn(nsamples, T, H, generator=g) * ch
    return x.to(dtype).to(device)


def causal_mask(T, device="cpu"):
    m = torch.full((T, T), torch.finfo(torch.float32).min, dtype=torch.float32, device=device)
    return torch.triu(m, diagonal=1)[None, None]


def synth_act_stats(cfg, n_layers, seed=3):
    """Synthetic act_scales / act_shifts dicts keyed like generate_act_scale_shift.py output."""
    g = torch.Generator().manual_seed(seed)
    H = cfg.hidden_size
    fam = cfg.family
    prefix = "model.layers" if fam == "llama" else "model.decoder.layers"
    names = ["self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj"] if fam == "llama" else \
        ["self_attn.q_proj", "self_attn.out_proj", "fc1"]
    scales, shifts = {}, {}
    for i in range(n_layers):
        for n in names:
            scales[f"{prefix}.{i}.{n}"] = torch.rand(H, generator=g) * 3 + 0.2
            shifts[f"{prefix}.{i}.{n}"] = torch.randn(H, generator=g) * 0.1
    return scales, shifts
