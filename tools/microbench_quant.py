import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd import _capi as C
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rows, cols = 4096, 4096
W = (torch.randn(rows, cols, device=dev) * 0.02).half()
G = torch.randn(rows, cols, device=dev).bfloat16()
cm = torch.rand(cols, device=dev) + 0.5; rd = torch.rand(rows, device=dev) + 0.5; sh = torch.randn(cols, device=dev)
up = torch.full((rows, 1), 4.0, device=dev); low = up.clone()
gws = torch.randn(rows, device=dev)
g_up, g_low = torch.empty_like(up), torch.empty_like(low)
g_cm, g_sh = torch.empty(cols, device=dev), torch.empty(cols, device=dev)
g_rd = torch.empty(rows, device=dev)
ws_n = C.size_call("oq_fakequant_bwd_workspace", rows, cols); ws = torch.empty(ws_n, device=dev)
y = torch.empty(rows, cols, device=dev, dtype=torch.bfloat16)
sc, zp, wsh = torch.empty(rows, device=dev), torch.empty(rows, device=dev), torch.empty(rows, device=dev)
xmn, xmx = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
P = C.fptr; st = C.stream()
def bwd(cm_, rd_, sh_, gws_, gcm_, gsh_, grd_, x=W, g=G, gx=None):
    C.call("oq_fakequant_bwd", C.ptr(x), C.dt(x), rows, cols, cols, 4, 0, P(cm_), P(rd_), None, P(sh_), P(up), P(low), P(xmn), P(xmx),
           C.ptr(g), C.dt(g), P(gws_), P(g_up), P(g_low), C.ptr(gx), C.dt(g), P(gcm_), P(gsh_), P(grd_), None, P(ws), ws_n, st)
def fwd(cm_, rd_, sh_, wsh_):
    C.call("oq_fakequant_fwd", C.ptr(W), C.dt(W), rows, cols, cols, 4, 0, P(cm_), P(rd_), None, P(sh_), P(up), P(low),
           C.ptr(y), 2, P(sc), P(zp), P(xmn), P(xmx), P(wsh_), None, None, st)
fwd(cm, rd, sh, wsh)
print("bwd LET full        ", timeit(lambda: bwd(cm, rd, sh, gws, g_cm, g_sh, g_rd)))
print("bwd LET no colgrads ", timeit(lambda: bwd(cm, rd, None, None, None, None, g_rd)))
print("bwd LET colmul only ", timeit(lambda: bwd(cm, None, None, None, g_cm, None, None)))
print("bwd LWC only        ", timeit(lambda: bwd(None, None, None, None, None, None, None)))
print("fwd LET full        ", timeit(lambda: fwd(cm, rd, sh, wsh)))
print("fwd LWC only        ", timeit(lambda: fwd(None, None, None, None)))
X = torch.randn(2048, 4096, device=dev).bfloat16(); GX = torch.empty_like(X)
xmn = X.float().amin(1).contiguous(); xmx = X.float().amax(1).contiguous()
rows = 2048
up = low = None
g_up = g_low = None
print("bwd act per-token   ", timeit(lambda: bwd(None, None, None, None, None, None, None, X, G[:2048].contiguous(), GX)))
z = torch.empty(64 * 1024 * 1024, device=dev, dtype=torch.bfloat16); z2 = torch.empty_like(z)
print("copy 128MB+128MB (torch)", timeit(lambda: z2.copy_(z)), "us ->", 2 * z.numel() * 2 / 1e6 / timeit(lambda: z2.copy_(z)), "TB/s")
