"""Diagnostic (not a test; CPU only): rounding_points.py one level finer, for the FORWARD rounding points that were left once the
Linears' forward products were integer-exact.  The oracle's LLaMA-7B W4A4 --lwc --let step is re-run with
one bf16 rounding point switched on at a time (value and gradient rounded where the product path stored the tensor in bf16):
  in_qk    the fake-quantised q, k as attention operands        core_p   the probabilities in front of P V
  in_v     the fake-quantised v                                  core_o   the stored attention output (in front of the o_proj input quantiser)
  core_bwd the attention backward's own roundings (dO, P for dV, dS)
  hidden   the hidden states (o_proj / down_proj output + residual)   mlp_pre  gate | up in front of silu * up -> quantiser
Columns: cosine of the gradient against the plain fp32 step for qkt_smooth_scale, q / k / v_proj LWC, out / qkv / fc1 smooth
scale, down_proj LWC.

    python tests/diag/forward_rounding_points.py [T]
"""
import math
import os
import sys

import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "diag"))
import rounding_points as D
import torch.nn.functional as F
R, S, BF = D.R, D.S, D.BF

class AttnModel2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, scale, fl):
        r = lambda t: t.to(BF).to(t.dtype)
        rq = r if fl.get("in_qk") else (lambda t: t)
        rvv = r if fl.get("in_v") else (lambda t: t)
        rp = r if fl.get("core_p") else (lambda t: t)
        ro = r if fl.get("core_o") else (lambda t: t)
        q2, k2, v2 = rq(q), rq(k), rvv(v)
        s = torch.matmul(q2, k2.transpose(2, 3)) * scale
        s = torch.max(s + mask, torch.tensor(torch.finfo(s.dtype).min))
        pt = torch.exp(s - s.amax(dim=-1, keepdim=True))
        l = pt.sum(dim=-1, keepdim=True)
        o = torch.matmul(rp(pt), v2) / l
        orr = ro(o)
        ctx.save_for_backward(q2, k2, v2, pt / l, orr)
        ctx.scale, ctx.fl = scale, fl
        return orr
    @staticmethod
    def backward(ctx, go):
        q, k, v, p, orr = ctx.saved_tensors
        fl = ctx.fl
        r = (lambda t: t.to(BF).to(t.dtype)) if fl.get("core_bwd") else (lambda t: t)
        go = r(go)
        dv = torch.matmul(r(p).transpose(2, 3), go)
        dp = torch.matmul(go, v.transpose(2, 3))
        d = (go * orr).sum(-1, keepdim=True)
        ds = r(p * (dp - d) * ctx.scale)
        return torch.matmul(ds, k), torch.matmul(ds.transpose(2, 3), q), dv, None, None, None

def step(blk, x, tgt, mask, pos, fl):
    for p in blk.params.values(): p.grad = None
    t = blk.temporaries(); nm = blk.names; bs, T, H = x.shape
    rv = lambda tt, on: D.RoundVG.apply(tt, on, on)
    def lin(xq, name):
        y = D.LinModel.apply(xq, t[name + ".weight"], False, True)
        return y + t[name + ".bias"] if t[name + ".bias"] is not None else y
    aq = lambda z: R.fake_quant(z, blk.spec.abits)
    h = aq(blk._norm(x, nm["ln1"], t))
    q = lin(h, nm["q"]).view(bs, T, blk.nh, blk.hd).transpose(1, 2)
    k = lin(h, nm["k"]).view(bs, T, blk.nkv, blk.hd).transpose(1, 2)
    v = lin(h, nm["v"]).view(bs, T, blk.nkv, blk.hd).transpose(1, 2)
    cos, sin = blk.cos[:T][pos].unsqueeze(1), blk.sin[:T][pos].unsqueeze(1)
    q = q * cos + R._rot_half(q) * sin
    k = k * cos + R._rot_half(k) * sin
    q, k, v = aq(q), aq(k), aq(v)
    o = AttnModel2.apply(q, k, v, mask, 1.0 / math.sqrt(blk.hd), fl).transpose(1, 2).reshape(bs, T, H)
    hid = fl.get("hidden", False)
    h1 = rv(x + lin(aq(o), nm["o"]), hid)
    h2 = aq(blk._norm(h1, nm["ln2"], t))
    mp = fl.get("mlp_pre", False)
    gate, up = rv(lin(h2, "mlp.gate_proj"), mp), rv(lin(h2, "mlp.up_proj"), mp)
    out = rv(h1 + lin(aq(F.silu(gate) * up), "mlp.down_proj"), hid)
    loss = F.mse_loss(tgt, out); loss.backward()
    return float(loss.detach()), {n: p.grad.detach().clone() for n, p in blk.params.items()}

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.set_num_threads(int(os.environ.get('OQ_THREADS', os.cpu_count() or 8)))
cfg = S.make_config("llama-7b"); H = cfg.hidden_size
layer = S.make_layer(cfg, seed=0, device="cpu")
weights = {n: p.detach().float() for n, p in layer.named_parameters()}
x = S.make_calib_inputs(1, T, H, dtype=torch.float32).to(BF).float()
mask, pos = S.causal_mask(T), torch.arange(T)[None]
sc, sh = S.synth_act_stats(cfg, 1)
cd = dict(hidden_size=H, num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=1e-6)
blk = R.Block("llama", cd, weights, R.QuantSpec(4, 4, None, True, True), max_pos=T)
blk.register_let(sc, sh, 0.5, 0, "model.layers")
with torch.no_grad(): tgt = blk.forward(x, mask, pos, None, False).to(BF).float()
l0, g0 = step(blk, x, tgt, mask, pos, {})
keys = ["qkt_smooth_scale", "self_attn.q_proj.weight_quantizer.upbound_factor", "self_attn.k_proj.weight_quantizer.upbound_factor",
        "self_attn.v_proj.weight_quantizer.upbound_factor", "out_smooth_scale", "qkv_smooth_scale", "fc1_smooth_scale",
        "mlp.down_proj.weight_quantizer.upbound_factor"]
ALLF = ["in_qk", "in_v", "core_p", "core_o", "core_bwd", "hidden", "mlp_pre"]
variants = [("all seven on (the production mode before section 3c)", ALLF)] + [("only " + f, [f]) for f in ALLF] + [
    ("all - in_qk", [f for f in ALLF if f != "in_qk"]),
    ("all - in_qk - core_o", [f for f in ALLF if f not in ("in_qk", "core_o")]),
    ("all - in_qk - core_o - core_p", [f for f in ALLF if f not in ("in_qk", "core_o", "core_p")]),
    ("all - in_qk - core_o - core_p - in_v", [f for f in ALLF if f not in ("in_qk", "core_o", "core_p", "in_v")]),
    ("all - in_qk - core_o - core_p - in_v - hidden", ["core_bwd", "mlp_pre"]),
]
for name, on in variants:
    l, g = step(blk, x, tgt, mask, pos, {f: True for f in on})
    cs = []
    for kx in keys:
        a, b = g[kx].double().reshape(-1), g0[kx].double().reshape(-1)
        cs.append(float(torch.dot(a, b) / (a.norm() * b.norm())))
    print(f"{name:50s} loss_rel {abs(l - l0) / l0:.1e} | " + " ".join(f"{c:.4f}" for c in cs), flush=True)
