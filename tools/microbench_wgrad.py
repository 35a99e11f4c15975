"""The four weight-gradient launches of a LLaMA-7B (or 13B) step: 256x128x64 kernel vs the 256x256x32 kernel, same process."""
import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd import ops
dev = "cuda:0"
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = 2048
H, I = (4096, 11008) if len(sys.argv) < 2 or sys.argv[1] == "7b" else (5120, 13824)
shapes = [("wgrad q|k|v", 3 * H, H), ("wgrad o", H, H), ("wgrad gate|up", 2 * I, H), ("wgrad down", H, I)]
tot = {"0": 0.0, "1": 0.0}
for name, M, N in shapes:
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(T, M, device=dev, generator=g).bfloat16(); b = torch.randn(T, N, device=dev, generator=g).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    for flag in ("0", "1", "0", "1"):
        os.environ["OQ_GEMM_W256"] = flag
        t = timeit(lambda: ops.gemm(a, b, c, M, N, T, M, N, N, False, False))
        res[flag] = min(res.get(flag, 1e9), t)
    fl = 2.0 * M * N * T
    for f in ("0", "1"): tot[f] += res[f]
    print(f"{name:14s} M={M:6d} N={N:6d} K={T}  256x128x64 {res['0']:7.1f} us {fl/res['0']/1e6:7.1f} TF/s | 256x256x32 {res['1']:7.1f} us {fl/res['1']/1e6:7.1f} TF/s")
print(f"four wgrad launches: {tot['0']:.1f} -> {tot['1']:.1f} us")
