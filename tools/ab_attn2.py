"""Time fused vs unfused causal attention fwd+bwd at the LLaMA-7B shape (tools only)."""
import math, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniquant_amd import ops
T, nh, hd = 2048, 32, 128
dev = "cuda"
q, k, v, go = (torch.randn(1, T, nh, hd, device=dev).to(torch.bfloat16) for _ in range(4))
mask = torch.triu(torch.full((T, T), torch.finfo(torch.float32).min, device=dev), 1)
scale = 1 / math.sqrt(hd)


def fused():
    qd, kd, vd = (t.detach().requires_grad_(True) for t in (q, k, v))
    o = ops.FusedCausalAttnFn.apply(qd, kd, vd, scale)
    o.backward(go)


def unfused():
    qd, kd, vd = (t.detach().requires_grad_(True) for t in (q, k, v))
    p = ops.SoftmaxFn.apply(ops.AttnScoresFn.apply(qd, kd, True), mask, scale, True)
    o = ops.AttnPVFn.apply(p, vd, True)
    o.backward(go)


def fwd_only():
    ops.FusedCausalAttnFn.apply(q, k, v, scale)


for name, fn in (("fused fwd+bwd", fused), ("unfused fwd+bwd", unfused), ("fused fwd", fwd_only)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:18s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
