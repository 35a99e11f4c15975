"""oq_gemm_i8: the integer-exact fprop of the fake-quant Linear (quantize/int_linear.py:59-62 on the grids of
quantize/quantizer.py:84-105).  The int32 accumulators must equal a torch int64 matmul of the codes bit for bit, and the
finished output must equal the fp64 evaluation of  s_a s_w sum (q_a - z_a)(q_w - z_w) + bias + addend  to fp32 rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _side(rows, K, bits, gen, zp_range=None):
    off = 128 if bits == 8 else 0                                               # the quantisers' storage convention
    q = torch.randint(0, 1 << bits, (rows, K), generator=gen)                   # grid codes
    codes = (q - off).to(torch.int8)
    scale = (torch.rand(rows, generator=gen) * 0.1 + 0.01).float()
    lo, hi = zp_range if zp_range else (0, (1 << bits) - 1)
    zp = torch.randint(lo, hi + 1, (rows,), generator=gen).float()
    csum = codes.double().sum(1).float()
    return q, codes, scale, zp, csum


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (2048, 4096, 4096), (2048, 1536, 11008), (300, 392, 1280),
                                   (8, 128, 256), (257, 136, 384), (16, 64, 128), (5, 24, 96), (1000, 4096 + 8, 512)])
@pytest.mark.parametrize("abits,wbits", [(4, 4), (8, 4), (6, 6), (8, 8)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_gemm_i8_exact(M, N, K, abits, wbits, out_dtype):
    from omniquant_amd import ops
    if (M * N * K > 2 ** 33) and (abits, wbits) != (4, 4):
        pytest.skip("large shape once")
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + abits * 11 + wbits)
    # zero-points beyond the grid on some rows (all-positive / all-negative rows give them): the correction terms must hold
    qa, ca, sa, za, csa = _side(M, K, abits, g, zp_range=(-40, 300))
    qw, cw, sw, zw, csw = _side(N, K, wbits, g)
    bias = torch.randn(N, generator=g)
    addend = torch.randn(M, N, generator=g).to(out_dtype)
    A = ops.IntCodes(ca.to(DEV), sa.to(DEV), za.to(DEV), csa.to(DEV), abits)
    W = ops.IntCodes(cw.to(DEV), sw.to(DEV), zw.to(DEV), csw.to(DEV), wbits)
    # (1) raw integer accumulators: scales 1, zero-points at the offset (z' = 0), no bias -> the output IS sum codes_a * codes_b
    one = lambda n: torch.ones(n, device=DEV)
    offa, offw = (128.0 if abits == 8 else 0.0), (128.0 if wbits == 8 else 0.0)
    A1 = ops.IntCodes(A.codes, one(M), torch.full((M,), offa, device=DEV), A.csum, abits)
    W1 = ops.IntCodes(W.codes, one(N), torch.full((N,), offw, device=DEV), W.csum, wbits)
    c = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_i8(A1, W1, c)
    want_int = (ca.to(DEV).double() @ cw.to(DEV).double().T)       # |values| < 2^53: exact in float64
    assert want_int.abs().max() < 2 ** 24
    assert torch.equal(c.double(), want_int), "int32 accumulators differ from the integer matmul"
    # (2) the finished output
    out = torch.empty(M, N, dtype=out_dtype, device=DEV)
    ops.gemm_i8(A, W, out, bias=bias.to(DEV), addend=addend.to(DEV))
    da = (qa.double() - za.double()[:, None]) * sa.double()[:, None]
    dw = (qw.double() - zw.double()[:, None]) * sw.double()[:, None]
    want = da.to(DEV) @ dw.to(DEV).T + bias.double().to(DEV) + addend.double().to(DEV)
    tol = 3e-6 if out_dtype == torch.float32 else 4e-3
    err = (out.double() - want).abs().max() / want.abs().max()
    assert err < tol, err
