"""Quantiser kernels, per shape of a LLaMA-7B W4A4 sample-step: segment kernels (OQ_ROWQ=0) vs wave-per-row kernels, in one
process on one box (tools only).  Prints microseconds and algorithmic TB/s."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from omniquant_amd import _capi as C  # noqa: E402

dev = "cuda:0"
P = C.fptr


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
    return e0.elapsed_time(e1) / n * 1e3


def case(name, rows, cols, let, lwc, act, nbytes_f, nbytes_b):
    g = torch.Generator(device=dev).manual_seed(0)
    if act:
        W = torch.randn(rows, cols, device=dev, generator=g).bfloat16()
    else:
        W = (torch.randn(rows, cols, device=dev, generator=g) * 0.02).half()
    G = torch.randn(rows, cols, device=dev, generator=g).bfloat16()
    cm = torch.rand(cols, device=dev) + 0.5 if let else None
    rd = torch.rand(rows, device=dev) + 0.5 if let else None
    sh = torch.randn(cols, device=dev) if let else None
    up = torch.full((rows, 1), 4.0, device=dev) if lwc else None
    low = up.clone() if lwc else None
    gws = torch.randn(rows, device=dev) if let else None
    g_up = torch.empty(rows, 1, device=dev) if lwc else None
    g_low = torch.empty(rows, 1, device=dev) if lwc else None
    g_cm = torch.empty(cols, device=dev) if let else None
    g_sh = torch.empty(cols, device=dev) if let else None
    g_rd = torch.empty(rows, device=dev) if let else None
    gx = torch.empty_like(G) if act else None
    y = torch.empty(rows, cols, device=dev, dtype=torch.bfloat16)
    sc, zp, wsh = (torch.empty(rows, device=dev) for _ in range(3))
    xmn, xmx = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    st = C.stream()
    want_codes = os.environ.get("OQ_MB_CODES", "0") != "0"       # OQ_MB_CODES=1: also write the integer side channel
    codes_t = torch.empty(rows, cols, device=dev, dtype=torch.int8) if want_codes else None
    csum_t = torch.empty(rows, device=dev) if want_codes else None
    CODES, CSUM = C.ptr(codes_t), P(csum_t)

    def fwd():
        C.call("oq_fakequant_fwd", C.ptr(W), C.dt(W), rows, cols, cols, 4, 0, P(cm), P(rd), None, P(sh), P(up), P(low),
               C.ptr(y), 2, P(sc), P(zp), P(xmn), P(xmx), P(wsh if let else None),
               CODES if os.environ.get("OQ_ROWQ") != "0" else None, CSUM if os.environ.get("OQ_ROWQ") != "0" else None, st)

    def bwd():
        ws_n = C.size_call("oq_fakequant_bwd_workspace", rows, cols)
        ws = torch.empty(ws_n, device=dev) if let else None
        C.call("oq_fakequant_bwd", C.ptr(W), C.dt(W), rows, cols, cols, 4, 0, P(cm), P(rd), None, P(sh), P(up), P(low), P(xmn), P(xmx),
               C.ptr(G), C.dt(G), P(gws), P(g_up), P(g_low), C.ptr(gx), C.dt(G), P(g_cm), P(g_sh), P(g_rd), None, P(ws),
               ws_n if let else 0, st)

    res = {}
    outs = {}
    for mode in ("0", "1"):
        os.environ["OQ_ROWQ"] = mode
        fwd()
        tf = timeit(fwd)
        bwd()
        tb = timeit(bwd)
        res[mode] = (tf, tb)
        outs[mode] = [t.clone() if t is not None else None for t in (y, sc, zp, wsh if let else None, g_up, g_low, g_cm, g_sh, g_rd, gx)]
    os.environ["OQ_ROWQ"] = "1"
    # agreement of the two kernel families (same helper arithmetic: forward bit-identical)
    same = all(a is None or torch.equal(a, b) for a, b in zip(outs["0"][:3], outs["1"][:3]))
    gerr = 0.0
    for a, b in zip(outs["0"][3:], outs["1"][3:]):
        if a is not None:
            gerr = max(gerr, float((a.float() - b.float()).abs().max() / (a.float().abs().max() + 1e-30)))
    (f0, b0), (f1, b1) = res["0"], res["1"]
    print(f"{name:34s} fwd {f0:7.1f} -> {f1:7.1f} us ({nbytes_f / f1 / 1e6:4.2f} TB/s)   bwd {b0:7.1f} -> {b1:7.1f} us "
          f"({nbytes_b / b1 / 1e6:4.2f} TB/s)   fwd bit-identical: {same}, max rel grad diff {gerr:.1e}")
    return f0, b0, f1, b1


tot = [0.0, 0.0, 0.0, 0.0]
for spec, mult in (
        (("W 4096x4096 LET+LWC", 4096, 4096, True, True, False, 4096 * 4096 * 4, 4096 * 4096 * 4), 4),
        (("W 11008x4096 LET+LWC", 11008, 4096, True, True, False, 11008 * 4096 * 4, 11008 * 4096 * 4), 2),
        (("W 4096x11008 LWC", 4096, 11008, False, True, False, 4096 * 11008 * 4, 4096 * 11008 * 4), 1),
        (("A 2048x4096 per-token", 2048, 4096, False, False, True, 2048 * 4096 * 4, 2048 * 4096 * 6), 3),
        (("A 2048x11008 per-token", 2048, 11008, False, False, True, 2048 * 11008 * 4, 2048 * 11008 * 6), 1)):
    r = case(*spec)
    for i in range(4):
        tot[i] += r[i] * mult
print(f"per 7B W4A4 step (x launches): fwd {tot[0]:.0f} -> {tot[2]:.0f} us, bwd {tot[1]:.0f} -> {tot[3]:.0f} us")
