"""INTEGRATION.md section 1, exercised: the reference's OWN block code running over our four operator classes.

A maintainer who swaps the four imports keeps models/int_llama_layer.py (block forward, :103-179, :213-267;
smooth_and_quant_temporary, :279-307) and models/transformation.py (:24-69) as they are: those files assign
`temp_weight` / `temp_bias` from OUTSIDE with plain torch expressions, call `weight_quantizer(temp_weight)` under autograd,
flip `use_temporary_parameter`, call `lin(x)`, `qkt_matmul.quant_x1 / quant_x2`, `qkt_matmul(q, k.transpose(2, 3))`,
`pv_matmul(p, v)` and the norms with `temp_weight / temp_bias`.  The driver below is written from that call sequence (it is
not the reference's text) and uses nothing but the public surface of UniformAffineQuantizer / QuantLinear / QuantMatMul /
OmniLlamaRMSNorm; every other operation is eager torch, as in the reference.  Output, loss and all 21 gradients are compared
with the reference-generated fixture in fp32 at the north-star's 1e-3."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from gpu_helpers import T, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _rope(q, k, position_ids, theta=10000.0):
    """transformers-4.31 rotary embedding: cos/sin cache indexed by position_ids, broadcast over heads."""
    hd = q.shape[-1]
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32, device=q.device) / hd))
    fr = torch.outer(position_ids[0].float(), inv)
    emb = torch.cat((fr, fr), dim=-1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]
    return q * cos + _rotate_half(q) * sin, k * cos + _rotate_half(k) * sin


class _RefStyleBlock(nn.Module):
    """A LLaMA decoder block assembled the way models/int_llama_layer.py does, from the four drop-in classes only."""

    def __init__(self, g, meta):
        super().__init__()
        from omniquant_amd import QuantLinear, QuantMatMul, OmniLlamaRMSNorm
        from omniquant_amd.calibrate import default_args
        c = meta["config"]
        a = default_args(wbits=meta["wbits"], abits=meta["abits"], group_size=meta["group_size"], lwc=meta["lwc"], let=meta["let"])
        H, I = c["hidden_size"], c["intermediate_size"]
        self.nh, self.hd, self.H = c["num_attention_heads"], H // c["num_attention_heads"], H

        def lin(name, o, i):
            m = nn.Linear(i, o, bias=False)
            m.weight.data = torch.from_numpy(g[f"w.{name}.weight"]).clone()          # fp16 master, as the HF model holds it
            return QuantLinear(m, a.weight_quant_params, a.act_quant_params)

        def norm(name):
            m = SimpleNamespace(weight=nn.Parameter(torch.from_numpy(g[f"w.{name}.weight"]).clone()))
            return OmniLlamaRMSNorm(m, eps=c["rms_norm_eps"])

        self.q_proj, self.k_proj, self.v_proj, self.o_proj = (lin(f"self_attn.{n}_proj", H, H) for n in "qkvo")
        self.gate_proj, self.up_proj, self.down_proj = lin("mlp.gate_proj", I, H), lin("mlp.up_proj", I, H), lin("mlp.down_proj", H, I)
        self.ln1, self.ln2 = norm("input_layernorm"), norm("post_attention_layernorm")
        self.qkt_matmul = QuantMatMul(a.q_quant_params, a.k_quant_params, matmul_func=torch.matmul)
        self.pv_matmul = QuantMatMul(a.p_quant_params, a.v_quant_params, matmul_func=torch.matmul)
        for n in ("qkt_smooth_scale", "qkv_smooth_shift", "qkv_smooth_scale", "out_smooth_shift", "out_smooth_scale",
                  "fc1_smooth_shift", "fc1_smooth_scale"):
            self.register_parameter(n, nn.Parameter(torch.from_numpy(g["p0." + n]).clone()))

    def names(self):
        return {"self_attn.q_proj": self.q_proj, "self_attn.k_proj": self.k_proj, "self_attn.v_proj": self.v_proj,
                "self_attn.o_proj": self.o_proj, "mlp.gate_proj": self.gate_proj, "mlp.up_proj": self.up_proj,
                "mlp.down_proj": self.down_proj}

    # ---- models/transformation.py:24-69 + models/int_llama_layer.py:279-307, as a call sequence -------------------------
    def smooth_and_quant_temporary(self):
        with torch.no_grad():
            for n, p in self.named_parameters():
                if "smooth_scale" in n:
                    small = p.abs() < 1e-2
                    p.data[small] = torch.sign(p.data[small]) * 1e-2
        for ln, fcs, s, b in ((self.ln1, (self.q_proj, self.k_proj, self.v_proj), self.qkv_smooth_scale, self.qkv_smooth_shift),
                              (self.ln2, (self.up_proj, self.gate_proj), self.fc1_smooth_scale, self.fc1_smooth_shift)):
            ln.use_temporary_parameter = True
            ln.temp_bias = (-1 * b) / s
            ln.temp_weight = ln.weight / s
            for fc in fcs:
                fc.use_temporary_parameter = True
                fc.temp_bias = fc.weight @ b
                fc.temp_weight = fc.weight * s.view(1, -1)
        v, o = self.v_proj, self.o_proj
        o.use_temporary_parameter = True
        v.temp_bias = (v.temp_bias - self.out_smooth_shift) / self.out_smooth_scale
        v.temp_weight = v.temp_weight / self.out_smooth_scale.view(-1, 1)
        o.temp_bias = o.weight @ self.out_smooth_shift
        o.temp_weight = o.weight * self.out_smooth_scale.view(1, -1)
        q, k, s = self.q_proj, self.k_proj, self.qkt_smooth_scale
        q.temp_weight = q.temp_weight / s.view(-1, 1)
        q.temp_bias = q.temp_bias / s.view(-1)
        k.temp_weight = k.temp_weight * s.view(-1, 1)
        k.temp_bias = k.temp_bias * s.view(-1)
        self.down_proj.temp_weight = self.down_proj.weight
        for m in self.names().values():
            m.temp_weight = m.weight_quantizer(m.temp_weight)            # <- the drop-in quantiser, under autograd
            if not hasattr(m, "temp_bias"):
                m.temp_bias = m.bias
            m.use_temporary_parameter = True

    # ---- models/int_llama_layer.py:103-179, 213-267 as a call sequence ---------------------------------------------------
    def forward(self, x, mask, position_ids):
        bs, Tn, _ = x.shape
        nh, hd = self.nh, self.hd
        res = x
        h = self.ln1(x)
        q = self.q_proj(h).view(bs, Tn, nh, hd).transpose(1, 2)
        k = self.k_proj(h).view(bs, Tn, nh, hd).transpose(1, 2)
        v = self.v_proj(h).view(bs, Tn, nh, hd).transpose(1, 2)
        q, k = _rope(q, k, position_ids)
        q = self.qkt_matmul.quant_x1(q)
        k = self.qkt_matmul.quant_x2(k)
        w = self.qkt_matmul(q, k.transpose(2, 3)) / math.sqrt(hd)
        w = w + mask
        w = torch.max(w, torch.tensor(torch.finfo(w.dtype).min, device=w.device))
        w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
        w = self.pv_matmul.quant_x1(w)
        v = self.pv_matmul.quant_x2(v)
        a = self.pv_matmul(w, v).transpose(1, 2).reshape(bs, Tn, self.H)
        x = res + self.o_proj(a)
        res = x
        h = self.ln2(x)
        return res + self.down_proj(torch.nn.functional.silu(self.gate_proj(h)) * self.up_proj(h))


def test_reference_style_block_over_the_dropin_modules():
    g, meta = load_golden("g3_step_llama_w4a4_lwc_let.npz")
    blk = _RefStyleBlock(g, meta).to(DEV)
    for n, m in blk.names().items():
        m.set_quant_state(weight_quant=False, act_quant=True)            # quantize/omniquant.py:180
        for side in ("upbound_factor", "lowbound_factor"):
            getattr(m.weight_quantizer, side).data = T(g[f"p0.{n}.weight_quantizer.{side}"], DEV)
    blk.qkt_matmul.set_quant_state(False, True)
    blk.pv_matmul.set_quant_state(False, True)
    blk.float()                                                           # quantize/omniquant.py:205 (quirk Q9)
    x, tgt, mask = T(g["x"], DEV), T(g["target"], DEV), T(g["mask"], DEV)
    pos = torch.from_numpy(g["position_ids"]).to(DEV)
    blk.smooth_and_quant_temporary()
    worst_tmp = 0.0
    for k_, ref in g.items():
        if k_.startswith("tmp."):
            mod, kind = k_[4:].rsplit(".temp_", 1)
            owner = blk.names().get(mod) or {"input_layernorm": blk.ln1, "post_attention_layernorm": blk.ln2}[mod]
            worst_tmp = max(worst_tmp, rel_err(getattr(owner, "temp_" + kind).float(), ref))
    out = blk(x, mask, pos)
    loss = torch.nn.functional.mse_loss(tgt, out)
    loss.backward()
    e_out = rel_err(out, g["out"])
    e_loss = abs(float(loss.detach()) - float(g["loss"][0])) / abs(float(g["loss"][0]))
    grads = {}
    for n in ("qkt_smooth_scale", "qkv_smooth_shift", "qkv_smooth_scale", "out_smooth_shift", "out_smooth_scale",
              "fc1_smooth_shift", "fc1_smooth_scale"):
        grads[n] = getattr(blk, n).grad
    for n, m in blk.names().items():
        for side in ("upbound_factor", "lowbound_factor"):
            grads[f"{n}.weight_quantizer.{side}"] = getattr(m.weight_quantizer, side).grad
    assert len(grads) == 21 and all(v is not None for v in grads.values())
    e_grad = {n: rel_err(v, g["grad." + n]) for n, v in grads.items()}
    print(f"reference-style block over the drop-in modules: temporaries {worst_tmp:.2e}, out {e_out:.2e}, loss {e_loss:.2e}, "
          f"worst grad {max(e_grad.values()):.2e} ({max(e_grad, key=e_grad.get)})")
    assert worst_tmp < 1e-3 and e_out < 1e-3 and e_loss < 1e-3
    assert max(e_grad.values()) < 1e-3, e_grad
