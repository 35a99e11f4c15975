"""Sibling projections as ONE GEMM launch (weights stacked along the output dimension) against one launch per matrix:
q/k/v (3 x [4096,4096]) and gate/up (2 x [11008,4096]) of a LLaMA-7B block, fprop / dgrad / wgrad.  Interleaved rounds in
one process, median of 5.  Usage: python tools/microbench_gemm_merged.py [H I T]"""
import sys, os, torch, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd import ops
dev = "cuda:0"
H, I, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 11008, 2048)
# name, parts, (M, N, K, akc, bkc) of one part, (M, N, K) merged
cases = [("fprop qkv", 3, (T, H, H, True, True), (T, 3 * H, H)),
         ("dgrad qkv", 3, (T, H, H, True, False), (T, H, 3 * H)),
         ("wgrad qkv", 3, (H, H, T, False, False), (3 * H, H, T)),
         ("fprop gate/up", 2, (T, I, H, True, True), (T, 2 * I, H)),
         ("dgrad gate/up", 2, (T, H, I, True, False), (T, H, 2 * I)),
         ("wgrad gate/up", 2, (I, H, T, False, False), (2 * I, H, T))]


def mk(M, N, K, akc, bkc):
    a = torch.randn((M, K) if akc else (K, M), device=dev).bfloat16()
    b = torch.randn((N, K) if bkc else (K, N), device=dev).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    return a, b, c, (M, N, K, K if akc else M, K if bkc else N, N, akc, bkc)


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, parts, (M, N, K, akc, bkc), (Mm, Nm, Km) in cases:
    sep = [mk(M, N, K, akc, bkc) for _ in range(parts)]
    mer = mk(Mm, Nm, Km, akc, bkc)

    def run_sep():
        for a, b, c, args in sep:
            ops.gemm(a, b, c, *args)

    def run_mer():
        a, b, c, args = mer
        ops.gemm(a, b, c, *args)
    ts, tm = [], []
    for _ in range(5):
        ts.append(timeit(run_sep)); tm.append(timeit(run_mer))
    us_s, us_m = statistics.median(ts), statistics.median(tm)
    fl = 2.0 * Mm * Nm * Km
    print(f"{name:14s} {parts} launches {us_s:7.1f} us ({fl/us_s/1e6:6.0f} TF/s)   merged M={Mm} N={Nm} K={Km}: {us_m:7.1f} us "
          f"({fl/us_m/1e6:6.0f} TF/s)   {100*(us_s-us_m)/us_s:+.1f} %", flush=True)
