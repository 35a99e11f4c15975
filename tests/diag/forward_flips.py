"""Diagnostic (not a test; needs the GPU): WHERE does the production-mode forward of one LLaMA-7B-shaped W4A4 --lwc --let
block leave the fp32 oracle's forward?  Every quantiser output and every side-channel tensor of the HIP step is compared
with the oracle's value at the same point: fraction of elements that sit on another grid point (a flipped 4-bit rounding
decision) and relative L2 error.

    python tests/diag/forward_flips.py [T]
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R          # noqa: E402  (test infrastructure may use the oracle)
from omniquant_amd import synthetic as S  # noqa: E402
from omniquant_amd import ops            # noqa: E402

DEV = "cuda:0"


def spy(name, log):
    orig = getattr(ops, name)

    class _Spy:
        @staticmethod
        def apply(*a):
            out = orig.apply(*a)
            log.append((name, a, out))
            return out
    setattr(ops, name, _Spy)
    return orig


def report(what, got, ref, step=None):
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().reshape(-1)
    l2 = float((got - ref).norm() / ref.norm())
    msg = f"{what:34s} rel L2 {l2:.3e}  max abs {float((got - ref).abs().max()):.3e}"
    if step is not None:
        st = step.detach().double().cpu().reshape(-1)
        flips = ((got - ref).abs() > 0.5 * st).double().mean()
        msg += f"  flipped decisions {float(flips):.3e}"
    print(msg, flush=True)


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 32))
    from omniquant_amd.calibrate import decoder_layer_class, default_args, register_let_parameters
    cfg = S.make_config("llama-7b")
    H, nh, hd = cfg.hidden_size, cfg.num_attention_heads, cfg.hidden_size // cfg.num_attention_heads
    layer = S.make_layer(cfg, seed=0, device="cpu")
    weights = {n: p.detach().float() for n, p in layer.named_parameters()}
    x = S.make_calib_inputs(1, T, H, dtype=torch.float32).to(torch.bfloat16).float()
    mask, pos = S.causal_mask(T), torch.arange(T)[None]
    sc, sh = S.synth_act_stats(cfg, 1)
    cd = dict(hidden_size=H, num_attention_heads=nh, num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=1e-6)
    blk = R.Block("llama", cd, weights, R.QuantSpec(4, 4, None, True, True), max_pos=T)
    blk.register_let(sc, sh, 0.5, 0, "model.layers")
    # ---- oracle forward with its intermediates (same expressions as Block.forward, fp32) ----
    with torch.no_grad():
        t = blk.temporaries()
        nm = blk.names
        aq = lambda z: R.fake_quant(z, 4)

        def seg_step(z):                       # the quantiser's scale per segment, broadcast to the elements
            return ((z.amax(-1, keepdim=True) - z.amin(-1, keepdim=True)) / 15.0).expand_as(z)
        lin = lambda z, n: F.linear(z, t[n + ".weight"], t[n + ".bias"])
        o_ = {}
        n1 = blk._norm(x, nm["ln1"], t)
        o_["h"], o_["h_step"] = aq(n1), seg_step(n1)
        q = lin(o_["h"], nm["q"]).view(1, T, nh, hd).transpose(1, 2)
        k = lin(o_["h"], nm["k"]).view(1, T, nh, hd).transpose(1, 2)
        v = lin(o_["h"], nm["v"]).view(1, T, nh, hd).transpose(1, 2)
        cos, sin = blk.cos[:T][pos].unsqueeze(1), blk.sin[:T][pos].unsqueeze(1)
        q = q * cos + R._rot_half(q) * sin
        k = k * cos + R._rot_half(k) * sin
        for nme, z in (("q", q), ("k", k), ("v", v)):
            o_[nme], o_[nme + "_step"] = aq(z).transpose(1, 2), seg_step(z).transpose(1, 2)     # [1, T, nh, hd]
        s = torch.matmul(o_["q"].transpose(1, 2), o_["k"].transpose(1, 2).transpose(2, 3)) / math.sqrt(hd) + mask
        s = torch.max(s, torch.tensor(torch.finfo(s.dtype).min))
        att = F.softmax(s, dim=-1, dtype=torch.float32)
        o = torch.matmul(att, o_["v"].transpose(1, 2)).transpose(1, 2).reshape(1, T, H)
        o_["o"] = o
        o_["oq"], o_["oq_step"] = aq(o), seg_step(o)
        h1 = x + lin(o_["oq"], nm["o"])
        o_["h1"] = h1
        n2 = blk._norm(h1, nm["ln2"], t)
        o_["h2"], o_["h2_step"] = aq(n2), seg_step(n2)
        act = F.silu(lin(o_["h2"], "mlp.gate_proj")) * lin(o_["h2"], "mlp.up_proj")
        o_["act"], o_["act_step"] = aq(act), seg_step(act)
        o_["out"] = h1 + lin(o_["act"], "mlp.down_proj")
    # ---- HIP forward, production mode, eager, with spies ----
    args = default_args(wbits=4, abits=4, group_size=None, lwc=True, let=True, alpha=0.5, net="llama-7b", nsamples=1)
    ql = decoder_layer_class("llama")(cfg, layer.to(DEV), args).to(DEV)
    ql.compute_dtype = torch.bfloat16
    ql.set_quant_state(False, True)
    ql.let = True
    register_let_parameters(ql, "llama", sc, sh, 0.5, 0, DEV)
    log = []
    for name in ("NormQuantFn", "QKVRopeQuantFn", "FusedCausalAttnFn", "FakeQuantFn", "StackedGateUpFn", "LinearFn"):
        spy(name, log)
    with torch.no_grad():
        ql.smooth_and_quant_temporary()
        n_w = len(log)
        out = ql(x.to(DEV).to(torch.bfloat16), attention_mask=mask.to(DEV), position_ids=pos.to(DEV))[0]
    torch.cuda.synchronize()
    ev = log[n_w:]
    print("HIP forward nodes:", [e[0] for e in ev])
    nq = [e for e in ev if e[0] == "NormQuantFn"]
    report("norm1 -> quant (values)", nq[0][2][0].float(), o_["h"], o_["h_step"])
    qkv = [e for e in ev if e[0] == "QKVRopeQuantFn"][0]
    stashes = qkv[1][11]
    for nme, y, st in zip("qkv", qkv[2], stashes):
        val = y.float() * st["scale"].reshape(1, T, -1, 1) if qkv[1][-1] else y.float()
        report(f"{nme} head quantiser (values)", val, o_[nme], o_[nme + "_step"])
    at = [e for e in ev if e[0] == "FusedCausalAttnFn"][0]
    wide = at[1][5].get("wide") if len(at[1]) > 5 and at[1][5] is not None else None
    report("attention output (f32 channel)" if wide is not None else "attention output (bf16)", (wide if wide is not None else at[2]).float().view(1, T, H), o_["o"])
    fq = [e for e in ev if e[0] == "FakeQuantFn"]
    report("o_proj input quantiser (values)", fq[0][2][0].float().view(1, T, H), o_["oq"], o_["oq_step"])
    lf = [e for e in ev if e[0] == "LinearFn"]
    st0 = lf[0][1][7] if len(lf[0][1]) > 7 else None
    h1 = st0["wide"] if st0 and st0.get("wide") is not None else lf[0][2]
    report("hidden after attention", h1.float().view(1, T, H), o_["h1"])
    report("norm2 -> quant (values)", nq[1][2][0].float(), o_["h2"], o_["h2_step"])
    gu = [e for e in ev if e[0] == "StackedGateUpFn"][0]
    report("silu*up -> quant (values)", gu[2].float().view(1, T, -1), o_["act"], o_["act_step"])
    st1 = lf[1][1][7] if len(lf[1][1]) > 7 else None
    o2 = st1["wide"] if st1 and st1.get("wide") is not None else out
    report("block output", o2.float().view(1, T, H), o_["out"])


if __name__ == "__main__":
    main()
