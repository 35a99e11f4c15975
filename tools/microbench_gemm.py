import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd import ops
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = 2048
shapes = [("fprop qkvo", T, 4096, 4096, True, True), ("fprop gate/up", T, 11008, 4096, True, True), ("fprop down", T, 4096, 11008, True, True),
          ("dgrad qkvo", T, 4096, 4096, True, False), ("dgrad gate/up", T, 4096, 11008, True, False), ("dgrad down", T, 11008, 4096, True, False),
          ("wgrad qkvo", 4096, 4096, T, False, False), ("wgrad gate/up", 11008, 4096, T, False, False), ("wgrad down", 4096, 11008, T, False, False)]
tot_t = tot_f = 0
for name, M, N, K, akc, bkc in shapes:
    a = torch.randn((M, K) if akc else (K, M), device=dev).bfloat16()
    b = torch.randn((N, K) if bkc else (K, N), device=dev).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(a, b, c, M, N, K, K if akc else M, K if bkc else N, N, akc, bkc)
    us = timeit(f)
    ref = (a.float() if akc else a.float().T) @ (b.float().T if bkc else b.float())
    err = (c.float() - ref).abs().max().item() / ref.abs().max().item()
    fl = 2.0 * M * N * K
    mult = {"qkvo": 4, "gate/up": 2, "down": 1}[name.split()[1]]
    tot_t += us * mult; tot_f += fl * mult
    print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d}  {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  relerr {err:.2e}")
print(f"step linears: {tot_t/1e3:.3f} ms  {tot_f/tot_t/1e6:.1f} TF/s")
print("--- layout vs shape")
for name, M, N, K in [("4096x4096x2048", 4096, 4096, 2048), ("2048x4096x4096", 2048, 4096, 4096), ("2048x4096x11008", 2048, 4096, 11008)]:
    for akc in (True, False):
        for bkc in (True, False):
            a = torch.randn((M, K) if akc else (K, M), device=dev).bfloat16()
            b = torch.randn((N, K) if bkc else (K, N), device=dev).bfloat16()
            c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            us = timeit(lambda: ops.gemm(a, b, c, M, N, K, K if akc else M, K if bkc else N, N, akc, bkc))
            print(f"{name} akc={akc} bkc={bkc}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s")
