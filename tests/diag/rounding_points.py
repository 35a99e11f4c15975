"""Diagnostic (not a test; CPU only): WHICH bf16 rounding point of the production mode decorrelates the q/k-path gradients
from the fp32 oracle step?  One LLaMA-7B-shaped W4A4 --lwc --let sample-step of the oracle (oracle/ref_cpu.py) is re-run with
the product path's rounding points switched on one group at a time; gradients are compared with the plain fp32 step.

    python tests/diag/rounding_points.py [T]

Rounding groups (fp32 arithmetic everywhere, values AND gradients rounded where a tensor is materialised in bf16):
  lin_ops_fwd : the two operands of every Linear's FORWARD product rounded (bf16 GEMM operands; off = integer-exact fprop)
  lin_ops_bwd : the operands of dgrad / wgrad rounded (they stay bf16 on the integer path)
  pre         : q|k|v and gate|up projection outputs rounded before RoPE / silu*up -> quantiser
  hidden      : o_proj output (+ residual) rounded before the second norm -> quantiser; block output rounded
  attn_in     : the head-wise fake-quantised q, k, v rounded (operands of the attention products)
  attn_core   : the fused attention's own rounding points (P for the PV product, stored O, D = rowsum(dO*O), dS)
  grads       : activation gradients rounded where the product path stores them (same sites as the values above)
"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R          # noqa: E402  (test infrastructure may use the oracle)
from omniquant_amd import synthetic as S  # noqa: E402

BF = torch.bfloat16


class RoundVG(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, rv, rg):
        ctx.rg = rg
        return t.to(BF).to(t.dtype) if rv else t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return (g.to(BF).to(g.dtype) if ctx.rg else g), None, None


class LinModel(torch.autograd.Function):
    """y = x @ w.T with independently rounded operands in forward and backward."""
    @staticmethod
    def forward(ctx, x, w, rf, rb):
        r = lambda t: t.to(BF).to(t.dtype)
        ctx.save_for_backward(r(x) if rb else x, r(w) if rb else w)
        return F.linear(r(x) if rf else x, r(w) if rf else w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        return (g2 @ w).view_as(x), g2.t() @ x2, None, None


class AttnModel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, scale, core):
        s = torch.matmul(q, k.transpose(2, 3)) * scale
        s = torch.max(s + mask, torch.tensor(torch.finfo(s.dtype).min))
        pt = torch.exp(s - s.amax(dim=-1, keepdim=True))
        l = pt.sum(dim=-1, keepdim=True)
        r = (lambda t: t.to(BF).to(t.dtype)) if core else (lambda t: t)
        o = torch.matmul(r(pt), v) / l
        orr = r(o)
        ctx.save_for_backward(q, k, v, pt / l, orr)
        ctx.scale, ctx.core = scale, core
        return orr

    @staticmethod
    def backward(ctx, go):
        q, k, v, p, orr = ctx.saved_tensors
        r = (lambda t: t.to(BF).to(t.dtype)) if ctx.core else (lambda t: t)
        go = r(go)
        dv = torch.matmul(r(p).transpose(2, 3), go)
        dp = torch.matmul(go, v.transpose(2, 3))
        d = (go * orr).sum(-1, keepdim=True)
        ds = r(p * (dp - d) * ctx.scale)
        return torch.matmul(ds, k), torch.matmul(ds.transpose(2, 3), q), dv, None, None, None


def step(blk, x, tgt, mask, pos, fl):
    """One llama sample-step of `blk` with the rounding groups in `fl` on.  Returns (loss, grads)."""
    for p in blk.params.values():
        p.grad = None
    g = fl.get("grads", False)
    rv = lambda t, on: RoundVG.apply(t, on, on and g)
    t = blk.temporaries()
    nm = blk.names
    bs, T, H = x.shape

    def lin(xq, name):
        return LinModel.apply(xq, t[name + ".weight"], fl.get("lin_ops_fwd", False), fl.get("lin_ops_bwd", False)) + t[name + ".bias"] \
            if t[name + ".bias"] is not None else LinModel.apply(xq, t[name + ".weight"], fl.get("lin_ops_fwd", False), fl.get("lin_ops_bwd", False))

    aq = lambda z: R.fake_quant(z, blk.spec.abits)
    h = aq(blk._norm(x, nm["ln1"], t))
    pre = fl.get("pre", False)
    q = rv(lin(h, nm["q"]), pre).view(bs, T, blk.nh, blk.hd).transpose(1, 2)
    k = rv(lin(h, nm["k"]), pre).view(bs, T, blk.nkv, blk.hd).transpose(1, 2)
    v = rv(lin(h, nm["v"]), pre).view(bs, T, blk.nkv, blk.hd).transpose(1, 2)
    cos, sin = blk.cos[:T][pos].unsqueeze(1), blk.sin[:T][pos].unsqueeze(1)
    q = q * cos + R._rot_half(q) * sin
    k = k * cos + R._rot_half(k) * sin
    ai = fl.get("attn_in", False)
    q, k, v = rv(aq(q), ai), rv(aq(k), ai), rv(aq(v), ai)
    o = AttnModel.apply(q, k, v, mask, 1.0 / math.sqrt(blk.hd), fl.get("attn_core", False)).transpose(1, 2).reshape(bs, T, H)
    hid = fl.get("hidden", False)
    h1 = rv(x + lin(aq(o), nm["o"]), hid)
    h2 = aq(blk._norm(h1, nm["ln2"], t))
    gate, up = rv(lin(h2, "mlp.gate_proj"), pre), rv(lin(h2, "mlp.up_proj"), pre)
    out = rv(h1 + lin(aq(F.silu(gate) * up), "mlp.down_proj"), hid)
    loss = F.mse_loss(tgt, out)
    loss.backward()
    return float(loss.detach()), {n: p.grad.detach().clone() for n, p in blk.params.items()}


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    arch = sys.argv[2] if len(sys.argv) > 2 else "llama-7b"
    torch.set_num_threads(int(os.environ.get("OQ_THREADS", os.cpu_count() or 8)))
    cfg = S.make_config(arch)
    H = cfg.hidden_size
    layer = S.make_layer(cfg, seed=0, device="cpu")
    weights = {n: p.detach().float() for n, p in layer.named_parameters()}
    x = S.make_calib_inputs(1, T, H, dtype=torch.float32).to(BF).float()
    mask, pos = S.causal_mask(T), torch.arange(T)[None]
    sc, sh = S.synth_act_stats(cfg, 1)
    cd = dict(hidden_size=H, num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=1e-6)
    blk = R.Block("llama", cd, weights, R.QuantSpec(4, 4, None, True, True), max_pos=T)
    blk.register_let(sc, sh, 0.5, 0, "model.layers")
    with torch.no_grad():
        tgt = blk.forward(x, mask, pos, None, False).to(BF).float()
    t0 = time.time()
    l0, g0 = step(blk, x, tgt, mask, pos, {})
    print(f"fp32 step: loss {l0:.6f} ({time.time() - t0:.1f} s)", flush=True)
    ALL = ["lin_ops_fwd", "lin_ops_bwd", "pre", "hidden", "attn_in", "attn_core", "grads"]
    variants = [("bf16 everywhere (round-2 production mode)", ALL),
                ("integer fprop + fp32 pre (this round)", [f for f in ALL if f not in ("lin_ops_fwd", "pre")]),
                ("only lin_ops_fwd", ["lin_ops_fwd"]), ("only lin_ops_bwd", ["lin_ops_bwd"]), ("only pre", ["pre"]),
                ("only hidden", ["hidden"]), ("only attn_in", ["attn_in"]), ("only attn_core", ["attn_core"]),
                ("only grads (+ sites)", ["grads", "pre", "hidden", "attn_in"]),
                ("int fprop, + fp32 hidden", [f for f in ALL if f not in ("lin_ops_fwd", "pre", "hidden")]),
                ("int fprop, + fp32 hidden, exact attn_in", [f for f in ALL if f not in ("lin_ops_fwd", "pre", "hidden", "attn_in")])]
    keys = ["qkt_smooth_scale", "self_attn.q_proj.weight_quantizer.upbound_factor", "self_attn.k_proj.weight_quantizer.upbound_factor",
            "self_attn.v_proj.weight_quantizer.upbound_factor", "out_smooth_scale", "qkv_smooth_scale", "fc1_smooth_scale",
            "mlp.down_proj.weight_quantizer.upbound_factor"]
    for name, on in variants:
        t0 = time.time()
        l, g = step(blk, x, tgt, mask, pos, {f: True for f in on})
        cs = []
        for kx in keys:
            a, b = g[kx].double().reshape(-1), g0[kx].double().reshape(-1)
            cs.append(float(torch.dot(a, b) / (a.norm() * b.norm())))
        print(f"{name:45s} loss_rel {abs(l - l0) / l0:.1e} | " + " ".join(f"{c:.4f}" for c in cs) + f"  ({time.time() - t0:.0f} s)", flush=True)
    print("columns: " + ", ".join(k.replace("self_attn.", "").replace(".weight_quantizer.upbound_factor", ".up") for k in keys))


if __name__ == "__main__":
    main()
