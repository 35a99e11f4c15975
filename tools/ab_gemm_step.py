"""The 12 linear GEMM launches of a LLaMA-7B sample-step with stacked sibling projections, per environment variant
(interleaved rounds in one process, median of 5): tools/ab_gemm_step.py OQ_GEMM_PERSIST=2 OQ_GEMM_PERSIST=1 ..."""
import sys, os, torch, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd import ops
dev = "cuda:0"
variants = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]] or [dict(OQ_GEMM_PERSIST="2"), dict(OQ_GEMM_PERSIST="1")]
KEYS = set(k for v in variants for k in v)
T, H, I = 2048, 4096, 11008
shapes = [("fprop qkv", T, 3 * H, H, True, True), ("fprop o", T, H, H, True, True), ("fprop gate|up", T, 2 * I, H, True, True),
          ("fprop down", T, H, I, True, True),
          ("dgrad down", T, I, H, True, False), ("dgrad gate|up", T, H, 2 * I, True, False), ("dgrad o", T, H, H, True, False),
          ("dgrad qkv", T, H, 3 * H, True, False),
          ("wgrad down", H, I, T, False, False), ("wgrad gate|up", 2 * I, H, T, False, False), ("wgrad o", H, H, T, False, False),
          ("wgrad qkv", 3 * H, H, T, False, False)]
bufs = []
for name, M, N, K, akc, bkc in shapes:
    a = torch.randn((M, K) if akc else (K, M), device=dev).bfloat16()
    b = torch.randn((N, K) if bkc else (K, N), device=dev).bfloat16()
    bufs.append((a, b, torch.empty(M, N, device=dev, dtype=torch.bfloat16)))


def run(i):
    name, M, N, K, akc, bkc = shapes[i]; a, b, c = bufs[i]
    ops.gemm(a, b, c, M, N, K, K if akc else M, K if bkc else N, N, akc, bkc)


res = {j: [[] for _ in shapes] for j in range(len(variants))}
for rnd in range(5):
    for j, env in enumerate(variants):
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        for i in range(len(shapes)):
            for _ in range(2): run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(i)
            e1.record(); torch.cuda.synchronize()
            res[j][i].append(e0.elapsed_time(e1) / 10 * 1e3)
print("shape".ljust(16) + "tiles  " + "  ".join(str(v).ljust(28) for v in variants))
tot = [0.0] * len(variants)
flops = 0.0
for i, (name, M, N, K, akc, bkc) in enumerate(shapes):
    tiles = ((M + 255) // 256) * ((N + 127) // 128)
    fl = 2.0 * M * N * K; flops += fl
    cells = []
    for j in range(len(variants)):
        us = statistics.median(res[j][i]); tot[j] += us
        cells.append(f"{us:7.1f} us {fl / us / 1e6:6.0f} TF/s".ljust(28))
    print(name.ljust(16) + f"{tiles:5d}  " + "  ".join(cells))
print("step".ljust(23) + "  ".join(f"{t / 1e3:7.3f} ms {flops / t / 1e6:6.0f} TF/s".ljust(28) for t in tot))
print("--- K sweep, M = N = 4096 (512 tiles = 2 rounds), per layout and variant: us")
for akc, bkc in ((True, True), (True, False), (False, False)):
    for K in (1024, 2048, 4096, 8192):
        M = N = 4096
        a = torch.randn((M, K) if akc else (K, M), device=dev).bfloat16()
        b = torch.randn((N, K) if bkc else (K, N), device=dev).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        cells = []
        for env in variants:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            ts = []
            for _ in range(3):
                for _ in range(2): ops.gemm(a, b, c, M, N, K, K if akc else M, K if bkc else N, N, akc, bkc)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm(a, b, c, M, N, K, K if akc else M, K if bkc else N, N, akc, bkc)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10 * 1e3)
            cells.append(f"{statistics.median(ts):7.1f}")
        print(f"a_kc={int(akc)} b_kc={int(bkc)} K={K:5d}: " + "  ".join(cells))
