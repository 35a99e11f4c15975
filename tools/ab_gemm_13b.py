"""LLaMA-2-13B: the three launches whose contraction oq_gemm_ws splits, split vs unsplit (tools only)."""
import sys, os, torch, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd import ops
dev = "cuda:0"
T, H, I = 2048, 5120, 13824
shapes = [("dgrad gate|up", T, H, 2 * I, True, False), ("dgrad qkv", T, H, 3 * H, True, False), ("fprop down", T, H, I, True, True),
          ("fprop o", T, H, H, True, True)]
for name, M, N, K, akc, bkc in shapes:
    a = torch.randn((M, K) if akc else (K, M), device=dev).bfloat16()
    b = torch.randn((N, K) if bkc else (K, N), device=dev).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for flag in ("0", "1", "0", "1"):
        os.environ["OQ_GEMM_SPLITK"] = flag
        for _ in range(3): ops.gemm(a, b, c, M, N, K, K if akc else M, K if bkc else N, N, akc, bkc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(a, b, c, M, N, K, K if akc else M, K if bkc else N, N, akc, bkc)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(flag, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * M * N * K
    u, s_ = min(res["0"]), min(res["1"])
    print(f"{name:14s} M={M} N={N} K={K}: unsplit {u:7.1f} us ({fl/u/1e6:5.0f} TF/s)   split {s_:7.1f} us ({fl/s_/1e6:5.0f} TF/s)   {100*(u-s_)/u:+.1f} %", flush=True)
