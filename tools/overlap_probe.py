"""Probe (GPU): do an MFMA-bound GEMM launch and an HBM-bound weight-quantiser launch overlap when they are issued on two
streams -- eagerly, and as two branches of one captured hipGraph?  Prints the serial and the two-stream times."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniquant_amd import ops

DEV = "cuda:0"


def main():
    torch.manual_seed(0)
    T, K, N = 2048, 4096, 22016
    gy = torch.randn(T, N, device=DEV).to(torch.bfloat16)
    x = torch.randn(T, K, device=DEV).to(torch.bfloat16)
    gw = torch.empty(N, K, dtype=torch.bfloat16, device=DEV)
    w = (torch.randn(11008, 4096, device=DEV) * 0.02).half()
    cm = torch.rand(4096, device=DEV) + 0.5
    up = torch.full((11008, 1), 4.0, device=DEV)
    lo = torch.full((11008, 1), 4.0, device=DEV)

    def gemm():
        ops.gemm(gy, x, gw, N, K, T, N, K, K, False, False)

    big = torch.randn(64 * 1024 * 1024, device=DEV)
    big16 = torch.empty(64 * 1024 * 1024, dtype=torch.bfloat16, device=DEV)
    mode = os.environ.get("PROBE", "letq")

    def quant():
        if mode == "cast":                       # a small-register streaming kernel (384 MB of traffic)
            from omniquant_amd import _capi as C
            C.call("oq_cast", C.ptr(big), C.dt(big), C.ptr(big16), C.dt(big16), big.numel(), C.stream())
            return
        for _ in range(2):
            ops.fake_quant(w, 4, None, up, lo, False, torch.bfloat16, {}, col_mul=cm)

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    s2 = torch.cuda.Stream()

    def both_serial():
        gemm()
        quant()

    def both_streams():
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            quant()
        gemm()
        cur.wait_stream(s2)

    with torch.no_grad():
        tg, tq = timeit(gemm), timeit(quant)
        ts, tp = timeit(both_serial), timeit(both_streams)
        print(f"eager : gemm {tg:.1f} us, quantiser {tq:.1f} us, serial {ts:.1f} us, two streams {tp:.1f} us")
        for name, fn in (("serial", both_serial), ("two branches", both_streams)):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            t = timeit(g.replay)
            print(f"graph : {name} {t:.1f} us")


if __name__ == "__main__":
    main()
