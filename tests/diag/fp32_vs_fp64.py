"""Diagnostic (not a test; CPU only): how far is the reference's fp32 arithmetic from ITSELF?  One LLaMA-7B-shaped W4A4 --lwc
--let sample-step of the oracle (oracle/ref_cpu.py, the reference-pinned fp32 restatement) against the same step evaluated
in float64.  fp32 summation noise (1e-6 relative in a 4096-term projection) flips a few 1e-5 of the 4-bit head-quantiser
decisions; every flip moves a whole attention row, the next quantisers amplify it, and the block output of the two runs
differs by 10-20 % (relative L2) while the loss agrees to 5e-4.  This is the floor any implementation is measured against
when it is compared with "the fp32 step" (tests/test_fullsize_parity.py).

    python tests/diag/fp32_vs_fp64.py [T] [out.npz]

With an output path the float64 step's loss and gradients are saved (tests/golden/oracle_f64_llama7b_w4a4_t2048.npz was made
this way); tests/test_fullsize_parity.py reports the production step's distance to them next to its distance to the fp32 step.
"""
import os
import sys

import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch.nn.functional as F
from oracle import ref_cpu as R
from omniquant_amd import synthetic as S
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.set_num_threads(int(os.environ.get('OQ_THREADS', os.cpu_count() or 8)))
cfg = S.make_config("llama-7b"); H = cfg.hidden_size
layer = S.make_layer(cfg, seed=0, device="cpu")
weights = {n: p.detach().float() for n, p in layer.named_parameters()}
x = S.make_calib_inputs(1, T, H, dtype=torch.float32).to(torch.bfloat16).float()
mask, pos = S.causal_mask(T), torch.arange(T)[None]
sc, sh = S.synth_act_stats(cfg, 1)
cd = dict(hidden_size=H, num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=1e-6)

def run(dt):
    blk = R.Block("llama", cd, weights, R.QuantSpec(4, 4, None, True, True), max_pos=T)
    blk.register_let(sc, sh, 0.5, 0, "model.layers")
    if dt == torch.float64:
        blk.w = {k: v.double() for k, v in blk.w.items()}
        for n, p in blk.params.items():
            p.data = p.data.double()
        blk.cos, blk.sin = blk.cos.double(), blk.sin.double()
        def _norm(x_, name, t=None):           # Block._norm without its float32 variance
            w = t[name + ".weight"] if t and name + ".weight" in t else blk.w[name + ".weight"]
            b = t[name + ".bias"] if t and name + ".bias" in t else blk.w.get(name + ".bias")
            xh = x_ * torch.rsqrt(x_.pow(2).mean(-1, keepdim=True) + blk.eps)
            return w * xh + b if b is not None else w * xh
        blk._norm = _norm
    xx, mm = x.to(dt), mask.to(dt)
    with torch.no_grad():
        tgt = blk.forward(xx, mm, pos, None, False)
    temps = blk.temporaries()
    out = blk.forward(xx, mm, pos, temps=temps, act_quant=True)
    loss = F.mse_loss(tgt, out); loss.backward()
    return float(loss), {n: p.grad.detach().double().clone() for n, p in blk.params.items()}, out.detach().double(), tgt.detach().double()

l32, g32, o32, t32 = run(torch.float32)
l64, g64, o64, t64 = run(torch.float64)
print(f"loss fp32 {l32:.6f} fp64 {l64:.6f} rel {abs(l32-l64)/l64:.2e}; block output rel L2 {float((o32-o64).norm()/o64.norm()):.3e}; teacher rel L2 {float((t32-t64).norm()/t64.norm()):.3e}")
for n in sorted(g32, key=lambda n: float(torch.dot(g32[n].reshape(-1), g64[n].reshape(-1)) / (g32[n].norm() * g64[n].norm()))):
    a, b = g32[n].reshape(-1), g64[n].reshape(-1)
    print(f"  {n:55s} cos {float(torch.dot(a,b)/(a.norm()*b.norm())):.5f} l2 {float((a-b).norm()/b.norm()):.4f}")

if len(sys.argv) > 2:
    import numpy as np
    np.savez_compressed(sys.argv[2], loss=np.float64(l64), T=np.int64(T), **{"grad." + n: g.numpy() for n, g in g64.items()})
    print("saved", sys.argv[2])
