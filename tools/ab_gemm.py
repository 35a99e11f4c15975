import sys, os, torch, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd import ops
dev = "cuda:0"
variants = [dict(OQ_GEMM_NO_P3="1"), dict(OQ_GEMM_STAGGER="0"), dict(OQ_GEMM_STAGGER="1")]
if len(sys.argv) > 1:
    variants = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
KEYS = set(k for v in variants for k in v) | {"OQ_GEMM_NO_P3", "OQ_GEMM_STAGGER"}
T = 2048
shapes = [("fprop qkvo", T, 4096, 4096, True, True, 4), ("fprop gate/up", T, 11008, 4096, True, True, 2), ("fprop down", T, 4096, 11008, True, True, 1),
          ("dgrad qkvo", T, 4096, 4096, True, False, 4), ("dgrad gate/up", T, 4096, 11008, True, False, 2), ("dgrad down", T, 11008, 4096, True, False, 1),
          ("wgrad qkvo", 4096, 4096, T, False, False, 4), ("wgrad gate/up", 11008, 4096, T, False, False, 2), ("wgrad down", 4096, 11008, T, False, False, 1)]
bufs = []
for name, M, N, K, akc, bkc, mult in shapes:
    a = torch.randn((M, K) if akc else (K, M), device=dev).bfloat16()
    b = torch.randn((N, K) if bkc else (K, N), device=dev).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bufs.append((a, b, c))
def run(i):
    name, M, N, K, akc, bkc, mult = shapes[i]; a, b, c = bufs[i]
    ops.gemm(a, b, c, M, N, K, K if akc else M, K if bkc else N, N, akc, bkc)
res = {j: [[] for _ in shapes] for j in range(len(variants))}
for rnd in range(5):                      # interleaved rounds in ONE process (guide 5.4 rule 24)
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
for j, env in enumerate(variants):
    tot_t = tot_f = 0
    line = []
    for i, (name, M, N, K, akc, bkc, mult) in enumerate(shapes):
        us = statistics.median(res[j][i]); tot_t += us * mult; tot_f += 2.0 * M * N * K * mult
        line.append(f"{us:6.1f}")
    print(env, " ".join(line), f"| step {tot_t/1e3:.3f} ms {tot_f/tot_t/1e6:.0f} TF/s")
