"""hipBLASLt / rocBLAS (torch.matmul) on the 7B step's GEMM shapes: a practical ceiling for the hand-written kernel
(tools only; nothing in the product path calls a BLAS library)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniquant_amd import ops
dev = "cuda"
T, H, I = 2048, 4096, 11008


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


# (name, M, N, K, a_kc, b_kc): C[M,N] = sum_k A(m,k) B(n,k) -- the 12 linear launches of a step (q|k|v and gate|up stacked)
shapes = [("fprop qkv", T, 3 * H, H, 1, 1), ("fprop o", T, H, H, 1, 1), ("fprop gate|up", T, 2 * I, H, 1, 1), ("fprop down", T, H, I, 1, 1),
          ("dgrad down", T, I, H, 1, 0), ("dgrad gate|up", T, H, 2 * I, 1, 0), ("dgrad o", T, H, H, 1, 0), ("dgrad qkv", T, H, 3 * H, 1, 0),
          ("wgrad down", H, I, T, 0, 0), ("wgrad gate|up", 2 * I, H, T, 0, 0), ("wgrad o", H, H, T, 0, 0), ("wgrad qkv", 3 * H, H, T, 0, 0)]
tot = [0.0, 0.0, 0.0]
for name, M, N, K, akc, bkc in shapes:
    A = (torch.randn((M, K) if akc else (K, M), device=dev) * 0.5).to(torch.bfloat16)
    B = (torch.randn((N, K) if bkc else (K, N), device=dev) * 0.5).to(torch.bfloat16)
    Cm = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    lda = K if akc else M
    ldb = K if bkc else N
    ours = lambda: ops.gemm(A, B, Cm, M, N, K, lda, ldb, N, bool(akc), bool(bkc))
    Am = A if akc else A.t()
    Bm = B.t() if bkc else B
    blas = lambda: torch.matmul(Am, Bm, out=Cm)
    t1, t2 = timeit(ours), timeit(blas)
    fl = 2.0 * M * N * K
    print(f"{name:14s} M={M:6d} N={N:6d} K={K:6d}  ours {fl / t1 / 1e12:7.1f} TF/s ({t1 * 1e6:6.1f} us)   torch.matmul {fl / t2 / 1e12:7.1f} TF/s ({t2 * 1e6:6.1f} us)")
    tot[0] += t1; tot[1] += t2; tot[2] += fl
print(f"step's 12 launches: ours {tot[0] * 1e3:.3f} ms ({tot[2] / tot[0] / 1e12:.0f} TF/s)   torch.matmul {tot[1] * 1e3:.3f} ms ({tot[2] / tot[1] / 1e12:.0f} TF/s)")
