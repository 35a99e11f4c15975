"""Determinism probe of the fused attention kernels in isolation (tools only)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniquant_amd import ops, _capi as C
dev = "cuda"
for (T, nh) in ((256, 2), (512, 4), (2048, 32)):
    g = torch.Generator().manual_seed(T)
    q, k, v, go = (torch.randn(1, T, nh, 128, generator=g).to(torch.bfloat16).to(dev) for _ in range(4))
    outs, lses, gqs, gks, gvs = [], [], [], [], []
    for it in range(12):
        qd, kd, vd = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        o = ops.FusedCausalAttnFn.apply(qd, kd, vd, 1 / math.sqrt(128))
        o.backward(go)
        outs.append(o.detach().clone()); gqs.append(qd.grad.clone()); gks.append(kd.grad.clone()); gvs.append(vd.grad.clone())
    torch.cuda.synchronize()
    for name, lst in (("o", outs), ("gq", gqs), ("gk", gks), ("gv", gvs)):
        bad = [int((lst[i].float() != lst[0].float()).sum()) for i in range(1, len(lst))]
        print(T, nh, name, "mismatch counts vs run 0:", bad)
