"""The four fprop launches of a LLaMA-7B W4A4 step: bf16 p3 kernel vs the int8 kernel (oq_gemm_i8), same process, same box."""
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
T = int(os.environ.get("T", 2048))
H, I = (4096, 11008) if len(sys.argv) < 2 or sys.argv[1] == "7b" else (5120, 13824)
shapes = [("fprop q|k|v", T, 3 * H, H), ("fprop o", T, H, H), ("fprop gate|up", T, 2 * I, H), ("fprop down", T, H, I)]
tb = ti = 0.0
for name, M, N, K in shapes:
    g = torch.Generator(device=dev).manual_seed(0)
    qa = torch.randint(0, 16, (M, K), device=dev, generator=g); qw = torch.randint(0, 16, (N, K), device=dev, generator=g)
    za = torch.randint(0, 16, (M,), device=dev, generator=g).float(); zw = torch.randint(0, 16, (N,), device=dev, generator=g).float()
    sa = torch.rand(M, device=dev, generator=g) * 0.1 + 0.01; sw = torch.rand(N, device=dev, generator=g) * 0.01 + 0.001
    a16 = ((qa - za[:, None]) * sa[:, None]).bfloat16(); w16 = ((qw - zw[:, None]) * sw[:, None]).bfloat16()
    A = ops.IntCodes(qa.to(torch.int8), sa, za, qa.sum(1).float(), 4)
    W = ops.IntCodes(qw.to(torch.int8), sw, zw, qw.sum(1).float(), 4)
    c16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); c32 = torch.empty(M, N, device=dev, dtype=torch.float32)
    t_b = timeit(lambda: ops.gemm(a16, w16, c16, M, N, K, K, K, N, True, True))
    t_i = timeit(lambda: ops.gemm_i8(A, W, c16))
    t_i32 = timeit(lambda: ops.gemm_i8(A, W, c32))
    fl = 2.0 * M * N * K
    tb += t_b; ti += t_i
    print(f"{name:14s} M={M:5d} N={N:6d} K={K:6d}  bf16 {t_b:7.1f} us {fl/t_b/1e6:7.1f} TF/s | i8->bf16 {t_i:7.1f} us {fl/t_i/1e6:7.1f} TOP/s "
          f"| i8->f32 {t_i32:7.1f} us {fl/t_i32/1e6:7.1f} TOP/s")
print(f"four fprop launches: bf16 {tb:.1f} us, i8 {ti:.1f} us")
