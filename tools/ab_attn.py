import sys, os, torch, statistics, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd import ops
dev = "cuda:0"
bs, T, nh, hd = 1, 2048, 32, 128
q = torch.randn(bs, T, nh, hd, device=dev).bfloat16().requires_grad_(True)
k = torch.randn(bs, T, nh, hd, device=dev).bfloat16().requires_grad_(True)
v = torch.randn(bs, T, nh, hd, device=dev).bfloat16().requires_grad_(True)
go = torch.randn(bs, T, nh, hd, device=dev).bfloat16()
def fwdbwd(causal):
    s = ops.AttnScoresFn.apply(q, k, causal)
    p = ops.SoftmaxFn.apply(s, None, 1 / math.sqrt(hd), causal)
    o = ops.AttnPVFn.apply(p, v, causal)
    o.backward(go)
variants = [dict(OQ_GEMM_NO_P3="1"), dict()]
res = {j: [] for j in range(len(variants))}
for rnd in range(4):
    for j, env in enumerate(variants):
        os.environ.pop("OQ_GEMM_NO_P3", None); os.environ.update(env)
        for _ in range(2): fwdbwd(True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fwdbwd(True)
        e1.record(); torch.cuda.synchronize()
        res[j].append(e0.elapsed_time(e1) / 5 * 1e3)
for j, env in enumerate(variants):
    print(env, f"attention fwd+bwd (causal) {statistics.median(res[j]):.1f} us")
