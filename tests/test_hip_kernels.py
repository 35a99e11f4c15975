"""-m gpu parity tests of the individual HIP kernels, called through the C ABI (ctypes), against
(a) the golden vectors generated from the reference and (b) plain torch float64/float32 restatements."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype).to(DEV)


def assert_close(got, ref, rtol, atol, what="", max_bad_frac=0.0):
    got = got.detach().double().cpu().numpy()
    ref = np.asarray(ref, np.float64).reshape(got.shape)
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert (nan_g == nan_r).all(), f"{what}: NaN pattern differs"
    bad = np.abs(got - ref)[~nan_r] > (atol + rtol * np.abs(ref[~nan_r]))
    frac = bad.mean() if bad.size else 0.0
    assert frac <= max_bad_frac, f"{what}: {bad.sum()} / {bad.size} out of tolerance (max abs err " \
                                 f"{np.abs(got - ref)[~nan_r].max():.3e})"


G1, G1META = load_golden("g1_quantizer.npz")


@pytest.mark.parametrize("i", range(len(G1META["cases"])))
def test_fakequant_vs_reference_golden(i):
    """UniformAffineQuantizer fwd + bwd on the HIP path vs vectors produced by the reference's own class."""
    from omniquant_amd import UniformAffineQuantizer, OQError
    c = G1META["cases"][i]
    shape = c["shape"]
    q = UniformAffineQuantizer(n_bits=c["n_bits"], symmetric=c["symmetric"], per_channel_axes=[0],
                               dynamic_method=c["dynamic_method"], group_size=c["group_size"],
                               shape=tuple(shape) if c["lwc"] else None, lwc=c["lwc"]).to(DEV)
    if c["lwc"]:
        with torch.no_grad():
            q.upbound_factor.copy_(T(G1[f"c{i}_up"]))
            q.lowbound_factor.copy_(T(G1[f"c{i}_low"]))
    need_gx = f"c{i}_gx" in G1
    x = T(G1[f"c{i}_x"]).requires_grad_(need_gx)
    y = q(x)      # incl. the ragged symmetric case (deficiency > 0): zero-padded last group, quantizer.py:85-87,125-128
    # scale may differ by an ulp (sigmoid/exp implementation), which can flip a rounding tie for a few elements:
    # allow <=0.2 % of elements to be off by one quantisation step, everything else tight.
    step = float(np.nanmax(G1[f"c{i}_scale"])) if not np.isnan(G1[f"c{i}_scale"]).all() else 1.0
    assert_close(q.scale, G1[f"c{i}_scale"], 2e-6, 1e-9, c["tag"] + " scale")
    assert_close(q.round_zero_point, G1[f"c{i}_zp"], 0, 0, c["tag"] + " zp", max_bad_frac=0.002)
    assert_close(y, G1[f"c{i}_y"], 1e-5, 1e-7, c["tag"] + " y", max_bad_frac=0.002)
    assert_close(y, G1[f"c{i}_y"], 0, 1.001 * step, c["tag"] + " y (one step)")
    if y.requires_grad:
        (y * T(G1[f"c{i}_G"])).sum().backward()
        big = c["tag"] == "a4_tok_positive"
        if need_gx:
            assert_close(x.grad, G1[f"c{i}_gx"], 1e-4, 5e-4 if big else 2e-5, c["tag"] + " gx", max_bad_frac=0.002)
        if c["lwc"]:
            sc = max(np.abs(G1[f"c{i}_gup"]).max(), np.abs(G1[f"c{i}_glow"]).max(), 1e-6)
            assert_close(q.upbound_factor.grad, G1[f"c{i}_gup"], 1e-3, 2e-4 * sc, c["tag"] + " gup", max_bad_frac=0.01)
            assert_close(q.lowbound_factor.grad, G1[f"c{i}_glow"], 1e-3, 2e-4 * sc, c["tag"] + " glow", max_bad_frac=0.01)


@pytest.mark.parametrize("in_dtype,out_dtype", [(torch.float16, torch.bfloat16), (torch.float16, torch.float32),
                                                (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("rows,cols,group", [(64, 4096, None), (32, 11008, None), (48, 4096, 128), (16, 1024, 64),
                                             (8, 28672, None), (8, 8192, 64)])
def test_fakequant_let_fused_vs_oracle(in_dtype, out_dtype, rows, cols, group):
    """Fused LET transform + LWC fake quant + W@shift (fwd and every gradient) at production row lengths."""
    from oracle import ref_cpu as R
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    W = (torch.randn(rows, cols, generator=g) * 0.02).to(in_dtype)
    cm = (torch.rand(cols, generator=g) + 0.5)
    rd = (torch.rand(rows, generator=g) + 0.5)
    rm = (torch.rand(rows, generator=g) + 0.5)
    sh = torch.randn(cols, generator=g) * 0.1
    nseg = rows * (cols // (group or cols))
    up = 4 + torch.randn(nseg, 1, generator=g)
    low = 4 + torch.randn(nseg, 1, generator=g)
    G = torch.randn(rows, cols, generator=g)
    Gs = torch.randn(rows, generator=g)
    # oracle (fp32 CPU)
    leaves = [t.clone().requires_grad_(True) for t in (cm, rd, rm, sh, up, low)]
    ocm, ord_, orm, osh, oup, olow = leaves
    Wf = W.float()
    xo = ((Wf * ocm.view(1, -1)) / ord_.view(-1, 1)) * orm.view(-1, 1)
    yo = R.fake_quant(xo, 4, group, oup, olow)
    wso = Wf @ osh
    ((yo * G).sum() + (wso * Gs).sum()).backward()
    # HIP
    dl = [t.clone().to(DEV).requires_grad_(True) for t in (cm, rd, rm, sh, up, low)]
    dcm, drd, drm, dsh, dup, dlow = dl
    y, ws = ops.fake_quant(W.to(DEV), 4, group, dup, dlow, False, out_dtype, None, dcm, drd, drm, dsh)
    assert y.dtype == out_dtype
    tol = 1e-5 if out_dtype == torch.float32 else 8e-3
    step = float(yo.detach().abs().max()) / 7
    assert_close(y.float(), yo.detach().numpy(), tol, tol * 1e-2, "y", max_bad_frac=0.002)
    assert_close(y.float(), yo.detach().numpy(), 0, 1.01 * step + 1e-2 * step, "y one-step")
    assert_close(ws, wso.detach().numpy(), 1e-4, 1e-5, "wshift")
    ((y.float() * G.to(DEV)).sum() + (ws * Gs.to(DEV)).sum()).backward()
    for name, a, b in zip(("col_mul", "row_div", "row_mul", "shift", "up", "low"), dl, leaves):
        ref = b.grad.numpy()
        sc = max(np.abs(ref).max(), 1e-9)
        gt = 2e-3 if out_dtype == torch.float32 else 2e-2     # bf16 y => the incoming gradient is bf16-rounded
        assert_close(a.grad / sc, ref / sc, 5e-3, gt, "grad " + name, max_bad_frac=0.01)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (16, 64, 64), (200, 136, 72), (2048, 512, 1024)])
def test_gemm_layouts(dtype, a_kc, b_kc, M, N, K):
    """C = A(m,k) B(n,k) for every operand layout, ragged tiles included; asymmetric integer-valued data makes a
    transposed or permuted fragment show up as an O(1) error."""
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randint(-3, 4, (M, K), generator=g).float() + 0.25 * torch.arange(K).remainder(3)[None, :]
    B = torch.randint(-3, 4, (N, K), generator=g).float() - 0.5 * torch.arange(N).remainder(2)[:, None]
    ref = (A.double() @ B.double().T)
    a = (A if a_kc else A.T.contiguous()).to(dtype).to(DEV)
    b = (B if b_kc else B.T.contiguous()).to(dtype).to(DEV)
    bias = torch.randn(N, generator=g)
    for out_dtype in (torch.float32, dtype):
        c = torch.full((M, N), float("nan"), dtype=out_dtype, device=DEV)
        ops.gemm(a, b, c, M, N, K, K if a_kc else M, K if b_kc else N, N, a_kc, b_kc, bias=bias.to(DEV), alpha=0.5)
        want = 0.5 * ref + bias.double()[None, :]
        tol = 1e-5 if out_dtype == torch.float32 else 1e-2
        assert_close(c.float(), want.numpy(), tol, tol, f"gemm {dtype} akc={a_kc} bkc={b_kc} out={out_dtype}")


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("M,N,K", [(768, 1408, 128), (2304, 2688, 64), (520, 3080, 192), (4096, 1032, 64)])
def test_gemm_strip_ordered_tiles_and_lds_epilogue_exact(a_kc, b_kc, M, N, K):
    """The 256x128 MFMA kernel on tile grids that are ragged in every way the strip-ordered tile ids (8 n-tiles per strip,
    m-major inside, narrower last strip) and the LDS epilogue (whole-line stores, M / N edges inside a tile) have to get
    right: several strips, a last strip of 3 / 5 / 1 n-tiles, more tiles than CUs, partial edge tiles.  Integer-valued
    operands make every output exact in fp32, so a tile computed twice, skipped or stored at the wrong place is an exact
    mismatch; bias, alpha = 1 and the addend (output accumulation) go through the same epilogue."""
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    bias = torch.randint(-5, 6, (N,), generator=g).float()
    add = torch.randint(-9, 10, (M, N), generator=g).float()
    want = A @ B.T + bias[None, :] + add
    a = (A if a_kc else A.T.contiguous()).bfloat16().to(DEV)
    b = (B if b_kc else B.T.contiguous()).bfloat16().to(DEV)
    c = add.clone().to(DEV)
    ops.gemm(a, b, c, M, N, K, K if a_kc else M, K if b_kc else N, N, a_kc, b_kc, bias=bias.to(DEV), addend=c)
    assert torch.equal(c.cpu(), want)
    cb = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm(a, b, cb, M, N, K, K if a_kc else M, K if b_kc else N, N, a_kc, b_kc)
    assert torch.equal(cb.cpu().float(), (A @ B.T).bfloat16().float())


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 2048), (12288, 4096, 256), (2304, 2688, 64), (520, 3080, 96), (4096, 1032, 32),
                                   (256, 264, 160), (11008, 4096, 128)])
def test_gemm_weight_gradient_kernel_exact(monkeypatch, M, N, K):
    """gemm_bf16_w256_kernel (256x256x32 tiles, four-stage LDS ring; the weight-gradient shape class: both operands
    k-strided): exact on integer-valued operands -- full rounds, ragged M / N edges, a last strip narrower than 8 n-tiles,
    K of 1, 2, 3, 5 and 64 K-tiles (the ring's prologue, steady state and tail), bias and the output addend -- and
    bit-identical to the 256x128x64 kernel on random bf16 data (same contraction order per output element)."""
    from omniquant_amd import ops
    monkeypatch.setenv("OQ_GEMM_W256_MIN_TILES", "1")
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    bias = torch.randint(-5, 6, (N,), generator=g).float()
    add = torch.randint(-9, 10, (M, N), generator=g).float()
    want = A @ B.T + bias[None, :] + add
    a, b = A.T.contiguous().bfloat16().to(DEV), B.T.contiguous().bfloat16().to(DEV)
    c = add.clone().to(DEV)
    ops.gemm(a, b, c, M, N, K, M, N, N, False, False, bias=bias.to(DEV), addend=c)
    assert torch.equal(c.cpu(), want)
    cb = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm(a, b, cb, M, N, K, M, N, N, False, False)
    assert torch.equal(cb.cpu().float(), (A @ B.T).bfloat16().float())
    ar, br = torch.randn(K, M, generator=g).bfloat16().to(DEV), torch.randn(K, N, generator=g).bfloat16().to(DEV)
    c1 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    c2 = torch.empty_like(c1)
    ops.gemm(ar, br, c1, M, N, K, M, N, N, False, False)
    monkeypatch.setenv("OQ_GEMM_W256", "0")
    ops.gemm(ar, br, c2, M, N, K, M, N, N, False, False)
    assert torch.equal(c1, c2)


@pytest.mark.parametrize("a_kc,b_kc,M,N,K", [(True, False, 2048, 5120, 27648), (True, True, 2048, 5120, 15360), (False, False, 2048, 5120, 15360)])
def test_gemm_split_contraction_exact(a_kc, b_kc, M, N, K):
    """The LLaMA-2-13B launches with N = 5120 (320 tiles of 256 x 128 = 1.25 rounds of the 256 CUs, 216-432 K-tiles long):
    oq_gemm_workspace() asks for S x M x N fp32 words and oq_gemm_ws runs S x 320 work items over 1/S of the contraction each,
    then adds the partial outputs in index order.  Integer-valued operands make every partial sum exact in fp32, so the result
    must EQUAL the unsplit one and the exact product, with bias and output accumulation, in fp32 and in bf16."""
    from omniquant_amd import ops, _capi as C
    assert C.size_call("oq_gemm_workspace", M, N, K, C._DT[torch.bfloat16], 1, 0) >= 3 * M * N * 4
    assert C.size_call("oq_gemm_workspace", 2048, 4096, 4096, C._DT[torch.bfloat16], 1, 0) == 0      # whole rounds: no split
    g = torch.Generator().manual_seed(K)
    A = torch.randint(-3, 4, (M, K), generator=g).float()
    B = torch.randint(-3, 4, (N, K), generator=g).float()
    bias = torch.randint(-5, 6, (N,), generator=g).float().to(DEV)
    add = torch.randint(-9, 10, (M, N), generator=g).float().to(DEV)
    a = (A if a_kc else A.T.contiguous()).bfloat16().to(DEV)
    b = (B if b_kc else B.T.contiguous()).bfloat16().to(DEV)
    want = A.to(DEV) @ B.to(DEV).T + bias[None, :] + add               # exact: |sum| < 2^24
    c = add.clone()
    ops.gemm(a, b, c, M, N, K, K if a_kc else M, K if b_kc else N, N, a_kc, b_kc, bias=bias, addend=c)
    assert torch.equal(c, want)
    os.environ["OQ_GEMM_SPLITK"] = "0"
    try:
        c0 = add.clone()
        ops.gemm(a, b, c0, M, N, K, K if a_kc else M, K if b_kc else N, N, a_kc, b_kc, bias=bias, addend=c0)
    finally:
        del os.environ["OQ_GEMM_SPLITK"]
    assert torch.equal(c0, want)
    cb = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.gemm(a, b, cb, M, N, K, K if a_kc else M, K if b_kc else N, N, a_kc, b_kc, alpha=0.5)
    assert torch.equal(cb.float(), (0.5 * (A.to(DEV) @ B.to(DEV).T)).bfloat16().float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nh,nkv", [(4, 4), (8, 2)])
def test_attention_gemms(dtype, nh, nkv):
    """AttnScoresFn / AttnPVFn (strided-batched, GQA via zero stride) fwd + bwd vs torch."""
    from omniquant_amd import ops
    bs, Tn, hd = 2, 128, 64
    g = torch.Generator().manual_seed(nh)
    q = torch.randn(bs, Tn, nh, hd, generator=g)
    k = torch.randn(bs, Tn, nkv, hd, generator=g)
    v = torch.randn(bs, Tn, nkv, hd, generator=g)
    rep = nh // nkv
    ql, kl, vl = (t.clone().to(dtype).float().requires_grad_(True) for t in (q, k, v))
    kr = kl.repeat_interleave(rep, dim=2)
    vr = vl.repeat_interleave(rep, dim=2)
    s_ref = torch.einsum("bthd,bshd->bhts", ql, kr)
    p_ref = torch.softmax(s_ref / math.sqrt(hd), dim=-1)
    o_ref = torch.einsum("bhts,bshd->bthd", p_ref.to(dtype).float(), vr)
    Go = torch.randn(o_ref.shape, generator=g)
    (o_ref * Go).sum().backward()
    qd, kd, vd = (t.clone().to(dtype).to(DEV).requires_grad_(True) for t in (q, k, v))
    s = ops.AttnScoresFn.apply(qd, kd)
    p = ops.SoftmaxFn.apply(s, None, 1.0 / math.sqrt(hd))
    o = ops.AttnPVFn.apply(p, vd)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert_close(s.float(), s_ref.detach().numpy(), tol, tol, "scores")
    assert_close(o.float(), o_ref.detach().numpy(), tol, tol, "out")
    (o.float() * Go.to(DEV)).sum().backward()
    for n, a, b in (("gq", qd, ql), ("gk", kd, kl), ("gv", vd, vl)):
        sc = float(b.grad.abs().max())
        assert_close(a.grad.float() / sc, (b.grad / sc).numpy(), tol, tol, n)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("is_ln", [False, True])
@pytest.mark.parametrize("rows,cols", [(37, 64), (128, 4096), (16, 8192)])
def test_norm(dtype, is_ln, rows, cols):
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(cols)
    x = torch.randn(rows, cols, generator=g).to(dtype)
    w = 1 + 0.1 * torch.randn(cols, generator=g)
    b = 0.1 * torch.randn(cols, generator=g)
    G = torch.randn(rows, cols, generator=g)
    xl, wl, bl = x.float().clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    if is_ln:
        yr = torch.nn.functional.layer_norm(xl, (cols,), wl, bl, 1e-5)
    else:
        yr = wl * (xl * torch.rsqrt(xl.pow(2).mean(-1, keepdim=True) + 1e-6)) + bl
    (yr * G).sum().backward()
    xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = ops.NormFn.apply(xd, wd, bd, 1e-5 if is_ln else 1e-6, is_ln)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert_close(y.float(), yr.detach().numpy(), tol, tol, "y")
    (y.float() * G.to(DEV)).sum().backward()
    assert_close(xd.grad.float(), xl.grad.numpy(), max(tol, 1e-4), max(tol, 1e-4), "gx")
    sc = float(wl.grad.abs().max())
    assert_close(wd.grad / sc, (wl.grad / sc).numpy(), 1e-3, 1e-3 if dtype == torch.float32 else 2e-2, "gw")
    sb = float(bl.grad.abs().max())
    assert_close(bd.grad / sb, (bl.grad / sb).numpy(), 1e-3, 1e-3 if dtype == torch.float32 else 2e-2, "gb")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rope_silu_relu_softmax_mse(dtype):
    from omniquant_amd import ops, _capi as C
    g = torch.Generator().manual_seed(0)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    # rope
    bs, Tn, nh, hd = 2, 16, 4, 32
    x = torch.randn(bs, Tn, nh, hd, generator=g).to(dtype)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(Tn).float(), inv)
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos(), emb.sin()
    xl = x.float().clone().requires_grad_(True)
    rot = torch.cat((-xl[..., hd // 2:], xl[..., :hd // 2]), -1)
    yr = xl * cos[None, :, None, :] + rot * sin[None, :, None, :]
    G = torch.randn(yr.shape, generator=g)
    (yr * G).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    y = ops.RopeFn.apply(xd, cos.to(DEV).contiguous(), sin.to(DEV).contiguous())
    assert_close(y.float(), yr.detach().numpy(), tol, tol, "rope")
    (y.float() * G.to(DEV)).sum().backward()
    assert_close(xd.grad.float(), xl.grad.numpy(), tol, tol, "rope grad")
    # silu*mul, relu
    a, b = torch.randn(64, 256, generator=g).to(dtype), torch.randn(64, 256, generator=g).to(dtype)
    al, bl = a.float().clone().requires_grad_(True), b.float().clone().requires_grad_(True)
    yr = torch.nn.functional.silu(al) * bl
    G = torch.randn(64, 256, generator=g)
    (yr * G).sum().backward()
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = ops.SiluMulFn.apply(ad, bd)
    assert_close(y.float(), yr.detach().numpy(), tol, tol, "silu_mul")
    (y.float() * G.to(DEV)).sum().backward()
    assert_close(ad.grad.float(), al.grad.numpy(), tol, tol, "silu ggate")
    assert_close(bd.grad.float(), bl.grad.numpy(), tol, tol, "silu gup")
    ad2 = a.to(DEV).requires_grad_(True)
    y = ops.ReluFn.apply(ad2)
    (y.float() * G.to(DEV)).sum().backward()
    assert_close(y.float(), torch.relu(a.float()).numpy(), 0, 0, "relu")
    assert_close(ad2.grad.float(), (G * (a.float() > 0)).to(dtype).float().numpy(), tol, tol, "relu grad")
    # masked softmax (causal, finfo.min mask, clamp)
    rows, cols = 4 * 48, 48
    s = (torch.randn(rows, cols, generator=g) * 4).to(dtype)
    mask = torch.triu(torch.full((48, 48), torch.finfo(torch.float32).min), 1)
    sl = s.float().clone().requires_grad_(True)
    z = torch.max(sl.view(4, 48, 48) * 0.3 + mask, torch.tensor(torch.finfo(torch.float32).min))
    pr = torch.softmax(z, -1).view(rows, cols)
    G = torch.randn(rows, cols, generator=g)
    (pr.to(dtype).float() * G).sum().backward()
    sd = s.to(DEV).requires_grad_(True)
    p = ops.SoftmaxFn.apply(sd, mask.to(DEV), 0.3)
    assert_close(p.float(), pr.detach().numpy(), tol, tol, "softmax")
    (p.float() * G.to(DEV)).sum().backward()
    assert_close(sd.grad.float(), sl.grad.numpy(), 3 * tol, 3 * tol, "softmax grad")
    # fused mse + grad
    out, t1, t2 = (torch.randn(8, 64, generator=g).to(dtype) for _ in range(3))
    loss = torch.zeros(1 + 1024, device=DEV)
    gg = torch.empty(8, 64, dtype=dtype, device=DEV)
    o_d, t1_d, t2_d = out.to(DEV), t1.to(DEV), t2.to(DEV)
    C.call("oq_mse_fwd_bwd", C.ptr(o_d), C.dt(o_d), C.ptr(t1_d), C.ptr(t2_d), C.dt(o_d), out.numel(), 1.0, C.fptr(loss), C.ptr(gg),
           C.stream())
    ol = out.float().clone().requires_grad_(True)
    lr = torch.nn.functional.mse_loss(t1.float(), ol) + torch.nn.functional.mse_loss(t2.float(), ol)
    lr.backward()
    assert abs(float(loss[0]) - float(lr)) <= 1e-5 * abs(float(lr)) + 1e-7
    assert_close(gg.float(), ol.grad.numpy(), tol, tol * 1e-2, "mse grad")


def test_adamw_gradnorm_truncate_vs_torch():
    from omniquant_amd import _capi as C
    g = torch.Generator().manual_seed(1)
    n, n_let = 5000, 1200
    p0 = torch.randn(n, generator=g)
    pt_let = p0[:n_let].clone().requires_grad_(True)
    pt_lwc = p0[n_let:].clone().requires_grad_(True)
    opt = torch.optim.AdamW([{"params": [pt_let], "lr": 5e-3}, {"params": [pt_lwc], "lr": 1e-2}], weight_decay=0.0)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step, norm, ws = torch.zeros(1, device=DEV), torch.zeros(2, device=DEV), torch.zeros(2048, device=DEV)
    for it in range(5):
        gr = torch.randn(n, generator=g) * (10.0 ** (it - 2))
        pt_let.grad, pt_lwc.grad = gr[:n_let].clone(), gr[n_let:].clone()
        opt.step()
        gd = gr.to(DEV)
        C.call("oq_gradnorm", C.fptr(gd), n, C.fptr(norm), C.fptr(ws), C.stream())
        C.call("oq_adamw", C.fptr(p), C.fptr(gd), C.fptr(m), C.fptr(v), n, n_let, 5e-3, 1e-2, 0.9, 0.999, 1e-8, 0.0,
               C.fptr(step), C.fptr(norm), C.stream())
        assert abs(float(norm[0]) - float(gr.norm())) <= 1e-5 * float(gr.norm())
        assert float(norm[1]) == 1.0
    want = torch.cat([pt_let.detach(), pt_lwc.detach()])
    assert_close(p, want.numpy(), 1e-5, 1e-6, "adamw params")
    assert float(step) == 5.0
    # non-finite gradient: the step is skipped on the device
    gd = torch.randn(n, generator=g).to(DEV)
    gd[17] = float("inf")
    before = p.clone()
    C.call("oq_gradnorm", C.fptr(gd), n, C.fptr(norm), C.fptr(ws), C.stream())
    C.call("oq_adamw", C.fptr(p), C.fptr(gd), C.fptr(m), C.fptr(v), n, n_let, 5e-3, 1e-2, 0.9, 0.999, 1e-8, 0.0,
           C.fptr(step), C.fptr(norm), C.stream())
    assert float(norm[1]) == 0.0 and torch.equal(p, before) and float(step) == 5.0
    # truncate_number vs golden
    g5, _ = load_golden("g5_misc.npz")
    x = T(g5["trunc_x"]).contiguous()
    C.call("oq_truncate", C.fptr(x), x.numel(), 1e-2, C.stream())
    assert_close(x, g5["trunc_y"], 0, 0, "truncate")


def test_adamw_step_fused_vs_torch():
    """oq_adamw_step = grad norm + skip-on-inf + AdamW + truncate_number of the head of the arena + gradient clearing in
    three launches (the grad-norm's second stage also advances the step counter)."""
    from omniquant_amd import _capi as C
    g = torch.Generator().manual_seed(3)
    n, n_let, n_tr = 70000, 9000, 4000          # > 256 workgroups of 256: the partial stage is capped and strided
    p0 = torch.randn(n, generator=g) * 0.02     # many |p| < 1e-2: truncation matters
    thr = 1e-2

    def trunc(x):
        y = x.clone()
        small = y.abs() < thr
        y[small] = torch.sign(y[small]) * thr
        return y

    pt_let = p0[:n_let].clone().requires_grad_(True)
    pt_lwc = p0[n_let:].clone().requires_grad_(True)
    opt = torch.optim.AdamW([{"params": [pt_let], "lr": 5e-3}, {"params": [pt_lwc], "lr": 1e-2}], weight_decay=0.01)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step, norm, ws = torch.zeros(1, device=DEV), torch.zeros(2, device=DEV), torch.zeros(2048, device=DEV)
    loss_t = torch.zeros(1, device=DEV)
    log = torch.zeros(1 + 2 * 4, device=DEV)           # room for 4 of the 7 steps: the rest must only be counted
    logged = []

    def fused(gd):
        loss_t.fill_(float(len(logged)) + 0.5)
        C.call("oq_adamw_step", C.fptr(p), C.fptr(gd), C.fptr(m), C.fptr(v), n, n_let, n_tr, thr, 1, 5e-3, 1e-2, 0.9, 0.999,
               1e-8, 0.01, C.fptr(step), C.fptr(norm), C.fptr(ws), C.fptr(loss_t), C.fptr(log), 4, C.stream())
        logged.append((float(loss_t), float(norm[0])))

    for it in range(6):
        gr = torch.randn(n, generator=g) * (10.0 ** (it - 3))
        pt_let.grad, pt_lwc.grad = gr[:n_let].clone(), gr[n_let:].clone()
        opt.step()
        with torch.no_grad():                   # the reference truncates at the top of the next step
            pt_let[:n_tr] = trunc(pt_let[:n_tr])
        gd = gr.to(DEV)
        fused(gd)
        assert abs(float(norm[0]) - float(gr.norm())) <= 1e-5 * float(gr.norm())
        assert float(norm[1]) == 1.0 and float(step) == it + 1
        assert float(gd.abs().max()) == 0.0, "gradient arena not cleared"
    want = torch.cat([pt_let.detach(), pt_lwc.detach()])
    assert_close(p, want.numpy(), 1e-5, 1e-6, "fused adamw params")
    assert float(p[:n_tr].abs().min()) >= float(torch.tensor(thr, dtype=torch.float32))
    # non-finite gradient: no update, no step increment, gradients still cleared
    gd = torch.randn(n, generator=g).to(DEV)
    gd[n - 3] = float("nan")
    before, mb, vb = p.clone(), m.clone(), v.clone()
    fused(gd)
    assert float(norm[1]) == 0.0 and torch.equal(p, before) and torch.equal(m, mb) and torch.equal(v, vb) and float(step) == 6.0
    assert float(gd.abs().max()) == 0.0
    # bit-identical for identical input
    g1 = torch.randn(n, generator=g).to(DEV)
    g2 = g1.clone()
    p_snap, m_snap, v_snap, s_snap = p.clone(), m.clone(), v.clone(), step.clone()
    fused(g1)
    p1 = p.clone()
    p.copy_(p_snap); m.copy_(m_snap); v.copy_(v_snap); step.copy_(s_snap)
    fused(g2)
    assert torch.equal(p, p1) and float(step) == 7.0
    # the on-device step log: every call counted (skipped and repeated ones too), the first 4 recorded as (loss, norm)
    host = log.tolist()
    assert host[0] == float(len(logged)) == 9.0
    for k in range(4):
        assert host[1 + 2 * k] == logged[k][0] and host[2 + 2 * k] == logged[k][1]


@pytest.mark.parametrize("direct", ["1", "0"])
def test_colsum_bias_gradient_kernels(direct, monkeypatch):
    """oq_colsum (bias gradients): the one-launch kernel (a workgroup owns 64 columns and walks all rows) and the two-launch
    slab kernels, against a float64 sum; ragged row counts, both dtypes, bit-identical on repetition."""
    from omniquant_amd import _capi as C
    monkeypatch.setenv("OQ_COLSUM_DIRECT", direct)
    g = torch.Generator().manual_seed(5)
    for rows, cols, dt in ((2048, 4096, torch.bfloat16), (2048, 11008, torch.bfloat16), (2047, 2048, torch.bfloat16),
                           (97, 4096, torch.float32), (16, 512, torch.bfloat16), (9000, 2048, torch.float32), (7, 64, torch.float32)):
        x = torch.randn(rows, cols, generator=g).to(dt).to(DEV)
        ws_n = C.size_call("oq_colsum_workspace", rows, cols)
        ws = torch.empty(ws_n, device=DEV)
        outs = []
        for _ in range(2):
            out = torch.full((cols,), float("nan"), device=DEV)
            C.call("oq_colsum", C.ptr(x), C.dt(x), rows, cols, C.fptr(out), C.fptr(ws), ws_n, C.stream())
            outs.append(out)
        want = x.double().sum(0)
        err = float((outs[0].double() - want).abs().max())
        assert err <= 2e-6 * float(x.double().abs().sum(0).max()) + 1e-6, (rows, cols, dt, err)
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("n_back", [4, 3])
@pytest.mark.parametrize("routed", [True, False])
@pytest.mark.parametrize("cols", [4096, 5120])
def test_weight_quant_batch_matches_single_launches(n_back, routed, cols):
    """ops.WeightQuantBatch: the LET weights of one shape quantised in ONE multi-matrix launch per direction
    (oq_fakequant_fwd_multi / oq_fakequant_bwd_multi) give bit-identical outputs and gradients to one launch per matrix;
    a sibling whose gradient never arrives (n_back = 3) does not leave the others unlaunched.  routed: the gradients go to
    the optimiser's sinks / collector (the engine's case, backward launches are merged too) or back to autograd (then the
    backward launches stay separate: AccumulateGrad copies what it is handed at once)."""
    from omniquant_amd import ops
    rows = 512                     # cols: 4096 (LLaMA-7B: 2 chunks per lane) and 5120 (LLaMA-2-13B: 3 chunks per lane)
    modes = [dict(rd=True), dict(rm=True), dict(rd=True), dict()]          # q, k, v, o

    class Collector:
        def __init__(self):
            self.got = {}

        def add(self, par, g):
            self.got[par._name] = g

    def run(batched):
        outs, leaves = [], []
        coll = Collector()
        gen = torch.Generator().manual_seed(12)
        ctxm = ops.WeightQuantBatch() if batched else None
        if ctxm is not None:
            ctxm.__enter__()
        for i, m in enumerate(modes):
            W = (torch.randn(rows, cols, generator=gen) * 0.02).half().to(DEV)
            named = dict(cm=(torch.rand(cols, generator=gen) + 0.5), sh=torch.randn(cols, generator=gen),
                         rd=(torch.rand(rows, generator=gen) + 0.5) if m.get("rd") else None,
                         rm=(torch.rand(rows, generator=gen) + 0.5) if m.get("rm") else None,
                         up=torch.full((rows, 1), 4.0), low=torch.full((rows, 1), 3.5))
            t = {}
            for k, v in named.items():
                if v is None:
                    t[k] = None
                    continue
                v = v.to(DEV).requires_grad_(True)
                v._name = f"{i}.{k}"
                if routed:
                    v._oq_grad_sink = torch.zeros_like(v)
                    v._oq_collector = coll
                t[k] = v
            y, ws = ops.fake_quant(W, 4, up=t["up"], low=t["low"], out_dtype=torch.bfloat16, col_mul=t["cm"], row_div=t["rd"],
                                   row_mul=t["rm"], shift=t["sh"])
            outs.append((y, ws))
            leaves.append(t)
        if ctxm is not None:
            ctxm.__exit__(None, None, None)
        gen2 = torch.Generator().manual_seed(13)
        loss = 0.0
        for i, (y, ws) in enumerate(outs[:n_back]):
            G = torch.randn(rows, cols, generator=gen2).to(DEV)
            gw = torch.randn(rows, generator=gen2).to(DEV)
            loss = loss + (y.float() * G).sum() + (ws * gw).sum()
        loss.backward()
        ops.WeightQuantBatch.flush_pending()        # what optim.GradCollector.flush does first
        torch.cuda.synchronize()
        grads = {}
        for i, t in enumerate(leaves):
            for k, v in t.items():
                if v is None:
                    continue
                if not routed:
                    grads[v._name] = None if v.grad is None else v.grad.clone()
                elif k in ("up", "low"):
                    grads[v._name] = v._oq_grad_sink.clone()
                else:
                    grads[v._name] = coll.got.get(v._name)
        return [(y.detach().clone(), ws.detach().clone()) for y, ws in outs], grads

    o1, g1 = run(False)
    o2, g2 = run(True)
    for (ya, wa), (yb, wb) in zip(o1, o2):
        assert torch.equal(ya, yb) and torch.equal(wa, wb)
    assert g1.keys() == g2.keys()
    for k in g1:
        if int(k.split(".")[0]) < n_back:
            assert g1[k] is not None and g2[k] is not None and torch.equal(g1[k], g2[k]), k
            assert float(g1[k].abs().max()) > 0, k
    assert not ops.WeightQuantBatch.pending


def test_copy_samples_one_launch():
    """oq_copy_samples: the step's input sample and teacher output(s) reach the hipGraph's static buffers in one launch."""
    from omniquant_amd import _capi as C
    g = torch.Generator().manual_seed(2)
    for n, numel, dt in ((2, 2048 * 4096, torch.bfloat16), (3, 8 * 1000, torch.float32), (1, 8, torch.bfloat16)):
        src = [torch.randn(numel, generator=g).to(dt).to(DEV) for _ in range(n)]
        dst = [torch.zeros(numel, dtype=dt, device=DEV) for _ in range(n)]
        flat = [C.ptr(t) for s_, d_ in zip(src, dst) for t in (s_, d_)] + [None] * (6 - 2 * n)
        C.call("oq_copy_samples", n, *flat, numel * src[0].element_size(), C.stream())
        for s_, d_ in zip(src, dst):
            assert torch.equal(s_, d_)
    with pytest.raises(C.OQError):
        C.call("oq_copy_samples", 1, C.ptr(src[0]), C.ptr(dst[0]), None, None, None, None, 24, C.stream())


@pytest.mark.parametrize("heads", [(4, 4, 4), (8, 2, 2)])
@pytest.mark.parametrize("use_sib", [False, True])
def test_merged_qkv_rope_quant_matches_three_nodes(heads, use_sib):
    """ops.QKVRopeQuantFn (q, k, v projections into one buffer, RoPE + head-wise quant of all heads in one launch per
    direction, one bias column sum) against three ops.LinearRopeQuantFn nodes: bit-identical outputs, weight / bias
    gradients and input-gradient pieces; also with grouped-query head counts."""
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(21)
    bs, T, K, hd = 1, 64, 512, 128
    nhq, nhk, nhv = heads
    x = torch.randn(bs, T, K, generator=g).bfloat16().to(DEV)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.outer(torch.arange(T).float(), inv)
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(DEV).contiguous(), emb.sin().to(DEV).contiguous()
    Ws = [(torch.randn(n * hd, K, generator=g) * 0.05).bfloat16().to(DEV) for n in heads]
    Bs = [torch.randn(n * hd, generator=g).to(DEV) for n in heads]
    Gs = [torch.randn(bs, T, n, hd, generator=g).bfloat16().to(DEV) for n in heads]

    def run(merged):
        xl = x.clone().requires_grad_(True)
        ws = [w.clone().requires_grad_(True) for w in Ws]
        bb = [b.clone().requires_grad_(True) for b in Bs]
        sib = ops.SiblingGrads() if use_sib else None
        if merged:
            outs = ops.QKVRopeQuantFn.apply(xl, ws[0], bb[0], ws[1], bb[1], ws[2], bb[2], cos, sin, 4, hd, [{}, {}, {}], sib)
        else:
            outs = [ops.LinearRopeQuantFn.apply(xl, ws[i], bb[i], cos if i < 2 else None, sin if i < 2 else None, 4, hd, {}, sib)
                    for i in range(3)]
        loss = sum((o.float() * G.float()).sum() for o, G in zip(outs, Gs))
        loss.backward()
        torch.cuda.synchronize()
        gx = xl.grad.float().clone()
        if sib is not None:
            for part in sib.take():
                gx = gx + part.float()
        return [o.detach().clone() for o in outs], [w.grad.clone() for w in ws], [b.grad.clone() for b in bb], gx

    o1, w1, b1, x1 = run(False)
    o2, w2, b2, x2 = run(True)
    for a, b in zip(o1 + w1 + b1, o2 + w2 + b2):
        assert a.shape == b.shape and torch.equal(a, b)
    if use_sib:
        assert torch.equal(x1, x2)      # the same three pieces, added in the same order here
    else:
        # separate nodes: autograd adds three bf16 tensors; merged: the dgrad GEMMs accumulate in their epilogue
        assert float((x1 - x2).abs().max()) <= 2e-2 * float(x1.abs().max())


@pytest.mark.parametrize("heads", [(4, 4, 4), (8, 2, 2)])
@pytest.mark.parametrize("use_sib", [False, True])
def test_stacked_qkv_weights_run_as_one_gemm_per_direction(heads, use_sib):
    """ops.QKVRopeQuantFn on q / k / v weights (and biases) that are row blocks of ONE buffer (what
    block_common._weight_slabs hands the weight quantisers): one GEMM with N = Nq + Nk + Nv forward, one dgrad GEMM with
    K = Nq + Nk + Nv, one wgrad GEMM with M = Nq + Nk + Nv -- against the same node on three separate tensors (one GEMM per
    matrix): outputs, weight and bias gradients bit-identical (every element is accumulated over the contraction in the same
    order); the input gradient is summed in the fp32 accumulators instead of rounding three bf16 terms first."""
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(33)
    bs, T, K, hd = 1, 64, 512, 128
    x = torch.randn(bs, T, K, generator=g).bfloat16().to(DEV)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.outer(torch.arange(T).float(), inv)
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(DEV).contiguous(), emb.sin().to(DEV).contiguous()
    Ns = [n * hd for n in heads]
    Wall = (torch.randn(sum(Ns), K, generator=g) * 0.05).bfloat16().to(DEV)
    Ball = torch.randn(sum(Ns), generator=g).to(DEV)
    Gs = [torch.randn(bs, T, n, hd, generator=g).bfloat16().to(DEV) for n in heads]
    offs = [0, Ns[0], Ns[0] + Ns[1], sum(Ns)]

    def run(stacked):
        xl = x.clone().requires_grad_(True)
        if stacked:
            wl, bl = Wall.clone().requires_grad_(True), Ball.clone().requires_grad_(True)
            ws = [wl[offs[i]:offs[i + 1]] for i in range(3)]
            bb = [bl[offs[i]:offs[i + 1]] for i in range(3)]
            assert ops.stacked_rows(ws) is not None and ops.stacked_vectors(bb) is not False
        else:
            ws = [Wall[offs[i]:offs[i + 1]].clone().requires_grad_(True) for i in range(3)]
            bb = [Ball[offs[i]:offs[i + 1]].clone().requires_grad_(True) for i in range(3)]
            assert ops.stacked_rows(ws) is None
        sib = ops.SiblingGrads() if use_sib else None
        outs = ops.QKVRopeQuantFn.apply(xl, ws[0], bb[0], ws[1], bb[1], ws[2], bb[2], cos, sin, 4, hd, [{}, {}, {}], sib)
        loss = sum((o.float() * G.float()).sum() for o, G in zip(outs, Gs))
        loss.backward()
        torch.cuda.synchronize()
        gx = xl.grad.float().clone()
        nparts = 0
        if sib is not None:
            for part in sib.take():
                gx = gx + part.float()
                nparts += 1
        gw = wl.grad.clone() if stacked else torch.cat([w.grad for w in ws])
        gb = bl.grad.clone() if stacked else torch.cat([b.grad for b in bb])
        return [o.detach().clone() for o in outs], gw, gb, gx, nparts

    o1, w1, b1, x1, p1 = run(False)
    o2, w2, b2, x2, p2 = run(True)
    for a, b in zip(o1 + [w1, b1], o2 + [w2, b2]):
        assert a.shape == b.shape and torch.equal(a, b)
    if use_sib:
        assert (p1, p2) == (2, 0)       # stacked: ONE input-gradient tensor, nothing left for the norm backward to add
    assert float((x1 - x2).abs().max()) <= 2e-2 * float(x1.abs().max())
    # the stacked dgrad is the better-rounded one: compare both with an fp32 reference of the sum
    assert float((x1 - x2).norm()) <= 1e-2 * float(x1.norm())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("nbits", [4, 0])
@pytest.mark.parametrize("use_sib", [False, True])
def test_stacked_gate_up_matches_separate_projections(dtype, nbits, use_sib):
    """ops.StackedGateUpFn (gate | up weights as row blocks of one buffer: one GEMM per direction, silu*up (-> per-token
    quant) kernels reading / writing the column blocks of the stacked buffers through a row stride) against two ops.LinearFn
    projections followed by ops.SiluMulQuantFn (nbits 4) or ops.SiluMulFn (nbits 0): activation and weight gradients
    bit-identical, bias gradients equal up to fp32 summation order, input gradient equal up to the rounding of the two bf16
    terms the separate path adds."""
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(5 + nbits)
    T, K, I = 96, 256, 1032        # I is not a multiple of the GEMM's N-tile: the up block starts mid-tile
    x = torch.randn(1, T, K, generator=g).to(dtype).to(DEV)
    Wall = (torch.randn(2 * I, K, generator=g) * 0.08).to(dtype).to(DEV)
    Ball = (torch.randn(2 * I, generator=g) * 0.1).to(DEV)
    G = torch.randn(1, T, I, generator=g).to(dtype).to(DEV)

    def run(stacked):
        xl = x.clone().requires_grad_(True)
        sib = ops.SiblingGrads() if use_sib else None
        stash = {}
        if stacked:
            wl, bl = Wall.clone().requires_grad_(True), Ball.clone().requires_grad_(True)
            y = ops.StackedGateUpFn.apply(xl, wl[:I], bl[:I], wl[I:], bl[I:], nbits, stash, sib)
        else:
            ws = [Wall[:I].clone().requires_grad_(True), Wall[I:].clone().requires_grad_(True)]
            bb = [Ball[:I].clone().requires_grad_(True), Ball[I:].clone().requires_grad_(True)]
            gate = ops.LinearFn.apply(xl, ws[0], bb[0], None, sib)
            up = ops.LinearFn.apply(xl, ws[1], bb[1], None, sib)
            y = ops.SiluMulQuantFn.apply(gate, up, nbits, stash) if nbits else ops.SiluMulFn.apply(gate, up)
        (y.float() * G.float()).sum().backward()
        torch.cuda.synchronize()
        gx = xl.grad.float().clone()
        if sib is not None:
            for part in sib.take():
                gx = gx + part.float()
        gw = wl.grad.clone() if stacked else torch.cat([w.grad for w in ws])
        gb = bl.grad.clone() if stacked else torch.cat([b.grad for b in bb])
        return y.detach().clone(), gw, gb, gx, stash

    y1, w1, b1, x1, s1 = run(False)
    y2, w2, b2, x2, s2 = run(True)
    assert torch.equal(y1, y2)
    assert torch.equal(w1, w2)
    assert torch.allclose(b1, b2, rtol=1e-5, atol=1e-5 * float(b1.abs().max()))   # one column sum over 2*I columns: other slabs
    if nbits:
        assert torch.equal(s1["scale"], s2["scale"]) and torch.equal(s1["zp"], s2["zp"])
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert float((x1 - x2).norm()) <= tol * float(x1.norm())


@pytest.mark.parametrize("heads", [(4, 4, 4), (8, 2, 2)])
def test_qkv_rope_split_identity_grid_matches_unfused_chain(heads):
    """ops.QKVRopeQuantFn with nbits = 16 (identity grid: the weight-only configurations' q / k / v path -- one stacked GEMM,
    RoPE + split in one launch per direction) against the unfused chain it replaces (three ops.LinearFn projections,
    ops.RopeFn on q and k; models/int_llama_layer.py:116-125): outputs and weight / bias gradients bit-identical, the input
    gradient equal up to the bf16 rounding of the three terms the unfused chain adds."""
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(44)
    bs, T, K, hd = 1, 64, 512, 128
    x = torch.randn(bs, T, K, generator=g).bfloat16().to(DEV)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.outer(torch.arange(T).float(), inv)
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().to(DEV).contiguous(), emb.sin().to(DEV).contiguous()
    Ns = [n * hd for n in heads]
    offs = [0, Ns[0], Ns[0] + Ns[1], sum(Ns)]
    Wall = (torch.randn(sum(Ns), K, generator=g) * 0.05).bfloat16().to(DEV)
    Ball = torch.randn(sum(Ns), generator=g).to(DEV)
    Gs = [torch.randn(bs, T, n, hd, generator=g).bfloat16().to(DEV) for n in heads]

    def run(fused):
        xl = x.clone().requires_grad_(True)
        wl, bl = Wall.clone().requires_grad_(True), Ball.clone().requires_grad_(True)
        ws = [wl[offs[i]:offs[i + 1]] for i in range(3)]
        bb = [bl[offs[i]:offs[i + 1]] for i in range(3)]
        if fused:
            outs = ops.QKVRopeQuantFn.apply(xl, ws[0], bb[0], ws[1], bb[1], ws[2], bb[2], cos, sin, 16, hd, None, None)
        else:
            pre = [ops.LinearFn.apply(xl, ws[i], bb[i], None, None).view(bs, T, heads[i], hd) for i in range(3)]
            outs = [ops.RopeFn.apply(pre[0], cos, sin), ops.RopeFn.apply(pre[1], cos, sin), pre[2]]
        loss = sum((o.float() * G.float()).sum() for o, G in zip(outs, Gs))
        loss.backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in outs], wl.grad.clone(), bl.grad.clone(), xl.grad.float().clone()

    o1, w1, b1, x1 = run(False)
    o2, w2, b2, x2 = run(True)
    for a, b in zip(o1, o2):
        assert a.shape == b.shape and torch.equal(a, b)
    assert torch.equal(w1, w2)
    assert torch.allclose(b1, b2, rtol=1e-5, atol=1e-5 * float(b1.abs().max()))
    assert float((x1 - x2).norm()) <= 1e-2 * float(x1.norm())


def test_bad_arguments_raise():
    """Error convention of the boundary: negative rc -> OQError with the library's message; CPU tensors refused."""
    from omniquant_amd import ops, OQError, _capi as C
    x = torch.randn(4, 20, device=DEV)          # 20 % 8 != 0: served by the generic-segment kernels, not an error
    assert ops.fake_quant(x, 4).shape == x.shape
    with pytest.raises(OQError):                # ... which have no LET transform
        ops.fake_quant(x, 4, col_mul=torch.ones(20, device=DEV))
    with pytest.raises(OQError):
        ops.fake_quant(torch.randn(4, 64), 4)   # CPU tensor
    with pytest.raises(OQError):
        ops.fake_quant(torch.randn(4, 64, device=DEV), 1)
    a = torch.randn(16, 12, device=DEV, dtype=torch.bfloat16)   # K=12 not a multiple of 8
    with pytest.raises(OQError):
        ops.gemm(a, a, torch.empty(16, 16, device=DEV), 16, 16, 12, 12, 12, 16, True, True)


def test_full_size_properties():
    """LLaMA-7B row shapes (BASELINE configs 1/2), size-independent checks: quantisation grid properties of the
    fake-quant kernel and a checksum-of-checksums for the bf16 MFMA GEMM in all three layouts."""
    from omniquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(0)
    W = (torch.randn(4096, 11008, device=DEV, generator=g) * 0.02).half()
    stash = {}
    for bits, group in ((4, None), (3, 128)):
        nseg = 4096 * (11008 // (group or 11008))
        up = torch.full((nseg, 1), 4.0, device=DEV)
        y = ops.fake_quant(W, bits, group, up, up.clone(), False, torch.float32, stash)
        s = stash["scale"].view(4096, -1)
        seg = group or 11008
        yv, xv = y.view(4096, -1, seg), W.float().view(4096, -1, seg)
        lev = torch.round(yv / s[..., None] + stash["zp"].view(4096, -1)[..., None])
        assert lev.min() >= 0 and lev.max() <= 2 ** bits - 1                     # on the grid
        hi = torch.sigmoid(torch.tensor(4.0)) * xv.amax(-1, keepdim=True)
        lo = torch.sigmoid(torch.tensor(4.0)) * xv.amin(-1, keepdim=True)
        inside = (xv <= hi) & (xv >= lo)
        err = (yv - xv).abs()
        assert (err[inside] <= 0.5001 * s[..., None].expand_as(err)[inside] + 1e-7).all()   # half-step bound
        y2 = ops.fake_quant(W, bits, group, up, up.clone(), False, torch.float32, None)
        assert torch.equal(y, y2)                                                  # deterministic
    Tn, K, N = 2048, 4096, 11008
    X = torch.randn(Tn, K, device=DEV, generator=g).bfloat16()
    Wb = (torch.randn(N, K, device=DEV, generator=g) * 0.02).bfloat16()
    Y = torch.empty(Tn, N, device=DEV, dtype=torch.float32)
    ops.gemm(X, Wb, Y, Tn, N, K, K, K, N, True, True)
    chk = X.float() @ Wb.float().sum(0)                # sum_n Y[t,n] = X[t,:] . sum_n W[n,:]
    assert torch.allclose(Y.sum(1), chk, rtol=2e-3, atol=2e-2 * float(chk.abs().max()))
    dY = torch.randn(Tn, N, device=DEV, generator=g).bfloat16()
    dX = torch.empty(Tn, K, device=DEV, dtype=torch.float32)
    ops.gemm(dY, Wb, dX, Tn, K, N, N, K, K, True, False)
    chk = dY.float() @ Wb.float().sum(1)
    assert torch.allclose(dX.sum(1), chk, rtol=2e-3, atol=2e-2 * float(chk.abs().max()))
    dW = torch.empty(N, K, device=DEV, dtype=torch.float32)
    ops.gemm(dY, X, dW, N, K, Tn, N, K, K, False, False)
    chk = dY.float().T @ X.float().sum(1)
    assert torch.allclose(dW.sum(1), chk, rtol=2e-3, atol=2e-2 * float(chk.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Tn", [128, 512, 768])
def test_causal_attention_fast_path(dtype, Tn):
    """Causal fast path (tile-skipping GEMM modes + causal softmax) vs the dense path with the additive mask:
    forward output and all three input gradients must agree (masked probabilities are exactly 0 in both)."""
    from omniquant_amd import ops
    bs, nh, nkv, hd = 1, 4, 2, 64
    g = torch.Generator().manual_seed(Tn)
    q = torch.randn(bs, Tn, nh, hd, generator=g)
    k = torch.randn(bs, Tn, nkv, hd, generator=g)
    v = torch.randn(bs, Tn, nkv, hd, generator=g)
    Go = torch.randn(bs, Tn, nh, hd, generator=g).to(dtype).to(DEV)
    mask = torch.triu(torch.full((Tn, Tn), torch.finfo(torch.float32).min), 1).to(DEV)
    assert ops.mask_is_causal(mask)
    assert not ops.mask_is_causal(torch.zeros(Tn, Tn, device=DEV))
    outs = []
    for causal in (False, True):
        qd, kd, vd = (t.clone().to(dtype).to(DEV).requires_grad_(True) for t in (q, k, v))
        s = ops.AttnScoresFn.apply(qd, kd, causal)
        p = ops.SoftmaxFn.apply(s, mask, 1.0 / math.sqrt(hd), causal)
        o = ops.AttnPVFn.apply(p, vd, causal)
        (o.float() * Go.float()).sum().backward()
        outs.append((o, qd.grad, kd.grad, vd.grad))
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for name, a, b in zip(("o", "gq", "gk", "gv"), outs[0], outs[1]):
        sc = float(a.float().abs().max())
        assert_close(b.detach().float() / sc, (a.detach().float() / sc).cpu().numpy(), tol, tol, f"causal {name}")


@pytest.mark.parametrize("Tn,nh,nkv", [(128, 2, 2), (256, 4, 2), (768, 3, 3), (1024, 2, 1)])
def test_fused_causal_attention(Tn, nh, nkv):
    """oq_attn_fwd / oq_attn_bwd (+ the causal dQ GEMM) vs an fp64 restatement of
    models/int_llama_layer.py:143-163 on the same bf16 inputs, and vs the unfused HIP kernels.
    Tolerance: bf16 probabilities / gradients -> 2e-2 of the tensor's max (same bar as the unfused bf16 path)."""
    from omniquant_amd import ops
    bs, hd = 1, 128
    g = torch.Generator().manual_seed(Tn + nh)
    q = (torch.randn(bs, Tn, nh, hd, generator=g) * 1.5).to(torch.bfloat16)
    k = torch.randn(bs, Tn, nkv, hd, generator=g).to(torch.bfloat16)
    v = torch.randn(bs, Tn, nkv, hd, generator=g).to(torch.bfloat16)
    Go = torch.randn(bs, Tn, nh, hd, generator=g).to(torch.bfloat16)
    scale = 1.0 / math.sqrt(hd)
    # fp64 reference
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    rep = nh // nkv
    kk = kr.repeat_interleave(rep, dim=2)
    vv = vr.repeat_interleave(rep, dim=2)
    s = torch.einsum("bthd,bshd->bhts", qr, kk) * scale
    s = s + torch.triu(torch.full((Tn, Tn), float("-inf"), dtype=torch.float64), 1)
    o_ref = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), vv)
    (o_ref * Go.double()).sum().backward()
    ref = (o_ref.detach(), qr.grad, kr.grad, vr.grad)
    # fused HIP path
    qd, kd, vd = (t.clone().to(DEV).requires_grad_(True) for t in (q, k, v))
    assert ops.fused_attention_supported(qd, True)
    o = ops.FusedCausalAttnFn.apply(qd, kd, vd, scale)
    (o.float() * Go.to(DEV).float()).sum().backward()
    got = (o, qd.grad, kd.grad, vd.grad)
    # unfused HIP path
    mask = torch.triu(torch.full((Tn, Tn), torch.finfo(torch.float32).min), 1).to(DEV)
    qu, ku, vu = (t.clone().to(DEV).requires_grad_(True) for t in (q, k, v))
    pu = ops.SoftmaxFn.apply(ops.AttnScoresFn.apply(qu, ku, True), mask, scale, True)
    ou = ops.AttnPVFn.apply(pu, vu, True)
    (ou.float() * Go.to(DEV).float()).sum().backward()
    unf = (ou, qu.grad, ku.grad, vu.grad)
    for name, a, r, u in zip(("o", "gq", "gk", "gv"), got, ref, unf):
        sc = float(r.abs().max())
        err = float((a.detach().double().cpu() - r).abs().max()) / sc
        err_u = float((u.detach().double().cpu() - r).abs().max()) / sc
        assert err < 2e-2, f"fused {name}: {err} (unfused path: {err_u})"
        assert err < 2.0 * err_u + 4e-3, f"fused {name} is less accurate than the unfused kernels: {err} vs {err_u}"


def test_act_stats_kernel_vs_golden_and_oracle():
    """oq_act_stats (LET-init statistics fused into the teacher pass) vs the reference's own
    generate_act_scale_shift.py functions (G6 fixture, rows padded to the kernel's 32-row minimum by replaying whole
    samples) and vs the oracle at the full [4, 2048, 4096] bf16 size fed in two chunks (bit-exact: max/min/abs are
    exact, the running average is the same fp32 expression in the same order)."""
    from omniquant_amd.actstats import ActStatCollector
    from oracle import ref_cpu as R
    g, _ = load_golden("g6_act_stats.npz")
    for k in ("fc1", "fc2"):
        x = torch.from_numpy(g[f"x_{k}"])                       # [5, 33, K] fp32
        col = ActStatCollector()
        col.update(k, x[:2].to(DEV))                            # chunked, as the engine's forward_bank does
        col.update(k, x[2:].to(DEV))
        np.testing.assert_array_equal(col.scales[k].cpu().numpy(), g[f"scale_{k}"])
        np.testing.assert_array_equal(col.shifts[k].cpu().numpy(), g[f"shift_{k}"])
    gen = torch.Generator().manual_seed(3)
    big = (torch.randn(4, 2048, 4096, generator=gen) * torch.exp(0.5 * torch.randn(4096, generator=gen))).to(torch.bfloat16)
    col = ActStatCollector()
    col.update("a", big[:3].to(DEV))
    col.update("a", big[3:].to(DEV))
    sc, sh = R.act_stats([big[i:i + 1] for i in range(4)])
    np.testing.assert_array_equal(col.scales["a"].cpu().numpy(), sc.numpy())
    np.testing.assert_array_equal(col.shifts["a"].cpu().numpy(), sh.numpy())
    with pytest.raises(Exception):
        col.update("cpu", torch.randn(2, 64, 64))               # no CPU fallback


def test_fused_attention_is_bitwise_reproducible():
    """No atomics, no timing-dependent arithmetic: repeated launches on the same inputs must agree bit for bit
    (this caught an MFMA -> inline-asm VALU hazard in the first version of attn_fwd_kernel)."""
    from omniquant_amd import ops
    for Tn, nh in ((512, 4), (2048, 8)):
        g = torch.Generator().manual_seed(Tn)
        q, k, v, go = (torch.randn(1, Tn, nh, 128, generator=g).to(torch.bfloat16).to(DEV) for _ in range(4))
        ref = None
        for _ in range(6):
            qd, kd, vd = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
            o = ops.FusedCausalAttnFn.apply(qd, kd, vd, 1.0 / math.sqrt(128))
            o.backward(go)
            cur = (o.detach().float(), qd.grad.float(), kd.grad.float(), vd.grad.float())
            if ref is None:
                ref = cur
            else:
                for name, a, b in zip(("o", "gq", "gk", "gv"), cur, ref):
                    assert int((a != b).sum()) == 0, f"{name} differs between two launches (T={Tn})"


@pytest.mark.parametrize("bits,group", [(4, None), (4, 128), (3, 128), (2, 64), (8, None)])
def test_real_quant_packing(bits, group):
    """oq_pack_weights vs the oracle's restatement of AutoGPTQ's pack() (PARITY UNPINNED: the library is not vendored
    in the reference) -- bit-exact -- and an unpack/dequantise round trip that must give back the fake-quant weight."""
    from omniquant_amd import ops
    from omniquant_amd.realquant import pack_linear, PackedLinear
    from oracle import ref_cpu as R
    out, inn = 96, 256
    g = torch.Generator().manual_seed(bits * 7 + (group or 0))
    W = torch.randn(out, inn, generator=g) * 0.05
    stash = {}
    wq = ops.fake_quant(W.to(DEV), bits, group, stash=stash, out_dtype=torch.float32)        # folded weight, scales, zeros
    sc, zp = stash["scale"], stash["zp"]
    packed = pack_linear(wq, sc, zp, bits, group)
    qw_ref, qz_ref = R.autogptq_pack(wq.cpu(), sc.cpu().reshape(out, -1), zp.cpu().reshape(out, -1), bits, group)
    assert torch.equal(packed["qweight"].cpu(), qw_ref)
    assert torch.equal(packed["qzeros"].cpu(), qz_ref)
    pl = PackedLinear(bits, group, inn, out, packed)
    back = pl.dequantize().float().cpu()
    ng = inn // (group or inn)
    want = (wq.cpu().reshape(out, ng, -1) / sc.cpu().reshape(out, ng, 1)).round() * sc.cpu().reshape(out, ng, 1).half().float()
    assert float((back - want.reshape(out, inn)).abs().max()) <= 2e-3 * float(wq.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_addend_and_sibling_projections(dtype):
    """oq_gemm's epilogue addend: (1) LinearFn(x, w, b, residual) == LinearFn(x, w, b) + residual (bit-exact in f32:
    same operation order); (2) SiblingLinearFn (dgrad accumulated in the epilogue) vs three independent LinearFn."""
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(17)
    T_, K, N = 256, 256, 384
    x = torch.randn(T_, K, generator=g).to(dtype).to(DEV)
    ws = [(torch.randn(N, K, generator=g) * 0.1).to(dtype).to(DEV) for _ in range(3)]
    bs = [torch.randn(N, generator=g).to(DEV), None, torch.randn(N, generator=g).to(DEV)]
    res = torch.randn(T_, N, generator=g).to(dtype).to(DEV)
    y0 = ops.LinearFn.apply(x, ws[0], bs[0])
    y1 = ops.LinearFn.apply(x, ws[0], bs[0], res)
    if dtype == torch.float32:
        assert torch.equal(y1, y0 + res)
    else:
        assert float((y1.float() - (y0.float() + res.float())).abs().max()) <= 2e-2 * float(y0.float().abs().max())
    Gs = [torch.randn(T_, N, generator=g).to(dtype).to(DEV) for _ in range(3)]
    xa = x.clone().requires_grad_(True)
    wa = [w.clone().requires_grad_(True) for w in ws]
    ya = [ops.LinearFn.apply(xa, w, b) for w, b in zip(wa, bs)]
    sum((y.float() * G.float()).sum() for y, G in zip(ya, Gs)).backward()
    xb = x.clone().requires_grad_(True)
    wb = [w.clone().requires_grad_(True) for w in ws]
    args = []
    for w, b in zip(wb, bs):
        args += [w, b]
    yb = ops.SiblingLinearFn.apply(xb, *args)
    sum((y.float() * G.float()).sum() for y, G in zip(yb, Gs)).backward()
    for a, b in zip(ya, yb):
        assert torch.equal(a, b)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((xa.grad.float() - xb.grad.float()).abs().max()) <= tol * float(xa.grad.float().abs().max())
    for a, b in zip(wa, wb):
        assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("rows,cols,seg,sym", [(16, 252, None, False), (8, 256, 96, True), (8, 192, 96, False),
                                               (4, 8 * 512 * 8 + 8, None, False), (6, 100, 40, True), (5, 77, 7, False)])
@pytest.mark.parametrize("nbits", [2, 4])
def test_fakequant_generic_segments_vs_oracle(rows, cols, seg, sym, nbits):
    """Segmentations the vector kernels do not take (row length not a multiple of 8, groups that are not 8*2^k wide,
    ragged last group = zero-padded, rows longer than 32768) run the generic one-wave-per-segment kernels: forward,
    LWC gradients and input gradient vs the CPU oracle (quantize/quantizer.py:84-147 incl. :85-87,:125-128)."""
    from oracle import ref_cpu as R
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(rows * cols + nbits)
    x = torch.randn(rows, cols, generator=g)
    if seg and cols % seg:
        x[1] = -x[1].abs()            # an all-negative ragged group: its amax is the padding zero
        x[2] = x[2].abs()
    nseg = rows * (-(-cols // (seg or cols)))
    up = 4 + torch.randn(nseg, 1, generator=g)
    low = 4 + torch.randn(nseg, 1, generator=g)
    G = torch.randn(rows, cols, generator=g)
    if seg and cols % seg and not sym:
        pytest.skip("ragged groups are symmetric-only in the reference (quantizer.py:69)")
    xo, uo, lo_ = x.clone().requires_grad_(True), up.clone().requires_grad_(True), low.clone().requires_grad_(True)
    yo, so, zo = R.fake_quant(xo, nbits, seg, uo, lo_, sym, return_qparams=True)
    (yo * G).sum().backward()
    xd, ud, ld = (t.clone().to(DEV).requires_grad_(True) for t in (x, up, low))
    stash = {}
    yd = ops.fake_quant(xd, nbits, seg, ud, ld, sym, None, stash)
    assert_close(stash["scale"], so.detach().numpy(), 2e-6, 1e-9, "scale")
    assert_close(stash["zp"], zo.detach().numpy(), 0, 0, "zp", max_bad_frac=0.01)
    assert_close(yd, yo.detach().numpy(), 1e-5, 1e-6, "y", max_bad_frac=0.005)
    (yd * G.to(DEV)).sum().backward()
    sc = max(float(uo.grad.abs().max()), float(lo_.grad.abs().max()), 1e-6)
    assert_close(ud.grad, uo.grad.numpy(), 2e-3, 1e-3 * sc, "gup", max_bad_frac=0.02)
    assert_close(ld.grad, lo_.grad.numpy(), 2e-3, 1e-3 * sc, "glow", max_bad_frac=0.02)
    assert_close(xd.grad, xo.grad.numpy(), 1e-4, 1e-4, "gx", max_bad_frac=0.005)


def test_identity_grid_keeps_the_let_transform():
    """n_bits >= 16: the reference's quantizer returns its input unchanged (quantizer.py:109-110), so with LET the
    temporary weight is just the transformed weight and the gradients are those of the transform alone."""
    from omniquant_amd.quantizer import UniformAffineQuantizer
    g = torch.Generator().manual_seed(3)
    rows, cols = 48, 512
    W = (torch.randn(rows, cols, generator=g) * 0.05).half()
    cm, rd, rm = (torch.rand(n, generator=g) + 0.5 for n in (cols, rows, rows))
    sh = torch.randn(cols, generator=g) * 0.1
    G, Gs = torch.randn(rows, cols, generator=g), torch.randn(rows, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (cm, rd, rm, sh)]
    xo = ((W.float() * leaves[0].view(1, -1)) / leaves[1].view(-1, 1)) * leaves[2].view(-1, 1)
    wso = W.float() @ leaves[3]
    ((xo * G).sum() + (wso * Gs).sum()).backward()
    q = UniformAffineQuantizer(n_bits=16, per_channel_axes=[0], dynamic_method="per_channel", shape=(rows, cols), lwc=True).to(DEV)
    dl = [t.clone().to(DEV).requires_grad_(True) for t in (cm, rd, rm, sh)]
    y, ws = q.quantize(W.to(DEV), out_dtype=torch.float32, col_mul=dl[0], row_div=dl[1], row_mul=dl[2], shift=dl[3])
    assert q.scale is None and q.round_zero_point is None
    assert_close(y, xo.detach().numpy(), 1e-6, 1e-9, "identity y")
    assert_close(ws, wso.detach().numpy(), 1e-4, 1e-6, "wshift")
    ((y * G.to(DEV)).sum() + (ws * Gs.to(DEV)).sum()).backward()
    for n, a, b in zip(("col_mul", "row_div", "row_mul", "shift"), dl, leaves):
        sc = float(b.grad.abs().max())
        assert_close(a.grad / sc, b.grad.numpy() / sc, 1e-3, 1e-4, "grad " + n)
    assert q.upbound_factor.grad is None
    w16 = q.quantize(W.to(DEV), out_dtype=torch.bfloat16)             # no LET: the weight itself, cast
    assert w16.dtype == torch.bfloat16 and torch.equal(w16, W.to(DEV).to(torch.bfloat16))
    q.register_scales_and_zeros()                                     # tolerates scale = None like the reference
    assert q.scales is None and q.zeros is None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(64, 11008), (33, 4096), (16, 13824), (5, 28672), (7, 520)])
def test_silu_mul_quant_fused_vs_oracle(dtype, rows, cols):
    """y = per_token_fake_quant(silu(gate) * up) in one kernel per direction (models/int_llama_layer.py:44-45 +
    quantize/int_linear.py:59-60) vs the CPU oracle's two steps: output, both input gradients (clip mask, straight-through
    rounding and amax / amin tie terms included) and the stashed scale / zero-point."""
    from oracle import ref_cpu as R
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    gate = (torch.randn(rows, cols, generator=g) * 1.5).to(dtype)
    up = torch.randn(rows, cols, generator=g).to(dtype)
    G = torch.randn(rows, cols, generator=g).to(dtype)
    go, uo = gate.float().clone().requires_grad_(True), up.float().clone().requires_grad_(True)
    yo, so, zo = R.fake_quant(torch.nn.functional.silu(go) * uo, 4, return_qparams=True)
    (yo * G.float()).sum().backward()
    gd, ud = gate.to(DEV).requires_grad_(True), up.to(DEV).requires_grad_(True)
    assert ops.silu_mul_quant_supported(gd, 4)
    stash = {}
    yd = ops.SiluMulQuantFn.apply(gd, ud, 4, stash)
    assert yd.dtype == dtype
    f32 = dtype == torch.float32
    step = float(so.max())
    assert_close(stash["scale"], so.detach().numpy(), 5e-6, 1e-9, "scale")
    assert_close(yd.float(), yo.detach().numpy(), 1e-5 if f32 else 8e-3, 1e-6 if f32 else 1e-2 * step, "y", max_bad_frac=0.003)
    assert_close(yd.float(), yo.detach().numpy(), 0, 1.02 * step, "y (one step)")
    (yd.float() * G.to(DEV).float()).sum().backward()
    for name, a, b in (("dgate", gd.grad, go.grad), ("dup", ud.grad, uo.grad)):
        sc = float(b.abs().max())
        assert_close(a.float() / sc, b.numpy() / sc, 2e-3 if f32 else 1e-2, 2e-4 if f32 else 4e-3, name, max_bad_frac=0.003)


@pytest.mark.parametrize("in_dtype,out_dtype", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("rot", [True, False])
def test_rope_quant_fused_vs_oracle(in_dtype, out_dtype, rot):
    """RoPE + per-(token, head) fake quant in one kernel per direction (models/int_llama_layer.py:124-125,140-143,161)
    vs the CPU oracle's two steps (transformers-4.31 rotate_half formula + quantize/quantizer.py:84-147), forward and the
    gradient w.r.t. the un-rotated input; rot=False is the v path (quantiser only)."""
    from oracle import ref_cpu as R
    from omniquant_amd import _capi as C
    g = torch.Generator().manual_seed(5 + int(rot))
    bs, Tn, nh, hd = 2, 24, 3, 128
    x = torch.randn(bs, Tn, nh, hd, generator=g).to(in_dtype)
    G = torch.randn(bs, Tn, nh, hd, generator=g).to(out_dtype)
    cos, sin = R._rope_tables(hd, Tn)
    xo = x.float().clone().requires_grad_(True)
    xr = xo
    if rot:
        c, s = cos[None, :, None, :], sin[None, :, None, :]
        xr = xo * c + R._rot_half(xo) * s
    yo, so, zo = R.fake_quant(xr, 4, return_qparams=True)
    (yo * G.float()).sum().backward()
    xd = x.to(DEV).contiguous()
    rows = bs * Tn
    y = torch.empty(rows, nh * hd, device=DEV, dtype=out_dtype)
    sc, zp, mn, mx = (torch.empty(rows * nh, device=DEV) for _ in range(4))
    cd, sd = (cos.to(DEV).contiguous(), sin.to(DEV).contiguous()) if rot else (None, None)
    assert C.size_call("oq_rope_quant_supported", C._DT[in_dtype], hd) == 1
    C.call("oq_rope_quant_fwd", C.ptr(xd), C.dt(xd), rows, Tn, nh, hd, C.fptr(cd), C.fptr(sd), 4, C.ptr(y), C.dt(y),
           C.fptr(sc), C.fptr(zp), C.fptr(mn), C.fptr(mx), C.stream())
    f32 = out_dtype == torch.float32 and in_dtype == torch.float32
    step = float(so.detach().max())
    assert_close(sc, so.detach().numpy().reshape(-1), 5e-6, 1e-9, "scale")
    assert_close(y.float().view(bs, Tn, nh, hd), yo.detach().numpy(), 1e-5 if f32 else 8e-3, 1e-6 if f32 else 1e-2 * step, "y",
                 max_bad_frac=0.003)
    Gd = G.to(DEV).contiguous()
    gx = torch.empty(rows, nh * hd, device=DEV, dtype=out_dtype)
    C.call("oq_rope_quant_bwd", C.ptr(xd), C.dt(xd), rows, Tn, nh, hd, C.fptr(cd), C.fptr(sd), 4, C.fptr(mn), C.fptr(mx),
           C.ptr(Gd), C.dt(Gd), C.ptr(gx), C.stream())
    scg = float(xo.grad.abs().max())
    assert_close(gx.float().view(bs, Tn, nh, hd) / scg, xo.grad.numpy() / scg, 2e-3 if f32 else 1e-2, 2e-4 if f32 else 6e-3, "gx",
                 max_bad_frac=0.003)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("is_ln,rows,cols", [(False, 37, 4096), (False, 20, 5120), (True, 33, 768), (False, 9, 8192), (True, 16, 1024)])
def test_norm_quant_fused_vs_oracle(dtype, is_ln, rows, cols):
    """y = per_token_fake_quant(norm(x)) in one kernel per direction (quantize/omni_norm.py:26-34,52-63 followed by
    quantize/int_linear.py:59-60) vs the CPU oracle's two steps: output, dL/dx (incl. the residual-path addend), and the
    norm weight / bias gradients (what LET's temp_weight / temp_bias receive)."""
    from oracle import ref_cpu as R
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 2 + 0.3).to(dtype)
    x[:, 5] *= 6
    w = 1 + 0.2 * torch.randn(cols, generator=g)
    b = 0.1 * torch.randn(cols, generator=g)
    G = torch.randn(rows, cols, generator=g).to(dtype)
    Gp = torch.randn(rows, cols, generator=g).to(dtype)
    xo, wo, bo = x.float().clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    if is_ln:
        h = torch.nn.functional.layer_norm(xo, (cols,), wo, bo, eps=1e-5)
    else:
        h = wo * (xo * torch.rsqrt(xo.pow(2).mean(-1, keepdim=True) + 1e-6)) + bo
    yo, so, zo = R.fake_quant(h, 4, return_qparams=True)
    ((yo * G.float()).sum() + (xo * Gp.float()).sum()).backward()
    xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    assert ops.norm_quant_supported(xd, 4)
    stash = {}
    yd, res = ops.NormQuantFn.apply(xd, wd, bd, 1e-5 if is_ln else 1e-6, is_ln, 4, stash)
    f32 = dtype == torch.float32
    step = float(so.detach().max())
    assert_close(stash["scale"], so.detach().numpy(), 2e-5, 1e-9, "scale")
    assert_close(yd.float(), yo.detach().numpy(), 1e-5 if f32 else 8e-3, 1e-6 if f32 else 1e-2 * step, "y", max_bad_frac=0.004)
    ((yd.float() * G.to(DEV).float()).sum() + (res.float() * Gp.to(DEV).float()).sum()).backward()
    for name, a, ref in (("gx", xd.grad, xo.grad), ("gw", wd.grad, wo.grad), ("gb", bd.grad, bo.grad)):
        sc = float(ref.abs().max())
        assert_close(a.float() / sc, ref.numpy() / sc, 3e-3 if f32 else 2e-2, 3e-4 if f32 else 8e-3, name, max_bad_frac=0.004)


def test_sibling_gradients_are_summed_inside_the_norm_backward():
    """Three projections read one fused norm -> quant output: with the SiblingGrads side channel only the first consumer's
    input gradient goes through autograd, the other two are added inside oq_norm_quant_bwd.  Must equal autograd's own
    accumulation (OQ_NO_SIBLING_GRADS=1) up to fp32 summation order."""
    import os
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(3)
    rows, K, N = 48, 1024, 256
    x = torch.randn(rows, K, generator=g)
    w = 1 + 0.1 * torch.randn(K, generator=g)
    Ws = [torch.randn(N, K, generator=g) * 0.05 for _ in range(3)]
    Gs = [torch.randn(rows, N, generator=g) for _ in range(3)]

    def run():
        xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        stash = {}
        y, res = ops.NormQuantFn.apply(xd, wd, None, 1e-6, False, 4, stash)
        sib = stash.get("sib")
        tot = res.sum() * 0.01
        for W_, G_ in zip(Ws, Gs):
            tot = tot + (ops.LinearFn.apply(y, W_.to(DEV), None, None, sib) * G_.to(DEV)).sum()
        tot.backward()
        return xd.grad.clone(), wd.grad.clone(), sib

    gx1, gw1, sib = run()
    assert sib is not None and sib.parts == [] and not sib.primary          # consumed by the norm backward
    os.environ["OQ_NO_SIBLING_GRADS"] = "1"
    try:
        gx0, gw0, sib0 = run()
    finally:
        del os.environ["OQ_NO_SIBLING_GRADS"]
    assert sib0 is None
    assert float((gx1 - gx0).abs().max()) <= 1e-5 * float(gx0.abs().max())
    assert float((gw1 - gw0).abs().max()) <= 1e-5 * float(gw0.abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("bs,T,nkv,rep,hd", [(1, 256, 8, 8, 128), (2, 33, 2, 4, 64), (1, 5, 1, 3, 8)])
def test_group_sum_gqa(dtype, bs, T, nkv, rep, hd):
    """oq_group_sum: dK / dV of a shared key-value head = sum over its query heads (backward of repeat_kv,
    models/int_llama_layer.py:136-141), fp32 accumulation in head order, both tensors in one launch."""
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(bs + T + rep)
    a = torch.randn(bs, T, nkv * rep, hd, generator=g).to(dtype).to(DEV)
    b = torch.randn(bs, T, nkv * rep, hd, generator=g).to(dtype).to(DEV)
    ya, yb = ops.group_sum(a, nkv, rep, b)
    for x, y in ((a, ya), (b, yb)):
        want = x.float().view(bs, T, nkv, rep, hd)
        acc = want[:, :, :, 0].clone()
        for r in range(1, rep):
            acc = acc + want[:, :, :, r]
        assert torch.equal(y, acc.to(dtype))
    (y1, none) = ops.group_sum(a, nkv, rep)
    assert none is None and torch.equal(y1, ya)


@pytest.mark.parametrize("rows,cols,seg", [(4096, 4096, 128), (300, 11008, 128), (1024, 8192, 64), (77, 1024, 32), (64, 2048, 256),
                                           (16, 4096, 512), (5, 128, 64)])
@pytest.mark.parametrize("nbits,symmetric", [(3, False), (2, False), (4, True)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_groupq_kernels_match_segment_kernels(monkeypatch, rows, cols, seg, nbits, symmetric, out_dtype):
    """Grouped weights without LET on the lanes-per-group kernels (oq_groupq.hip) vs the round-1 segment kernels (OQ_GROUPQ=0,
    themselves pinned by the reference's G1 vectors): forward bit-identical (same helper arithmetic), LWC gradients equal up
    to the summation order inside a group.  Includes a constant group (scale 0 -> NaN group, quirk Q1) and a NaN input."""
    from omniquant_amd import ops
    g = torch.Generator().manual_seed(rows + cols + seg + nbits)
    w = (torch.randn(rows, cols, generator=g) * 0.02).half()
    w[0, :seg] = 0.125                       # constant group
    if rows > 2:
        w[2, seg + 3] = float("nan")
    nseg = rows * (cols // seg)
    up = (4.0 + 0.5 * torch.randn(nseg, 1, generator=g))
    low = (4.0 + 0.5 * torch.randn(nseg, 1, generator=g))
    gy = torch.randn(rows, cols, generator=g).to(out_dtype)
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("OQ_GROUPQ", flag)
        u, l = up.clone().to(DEV).requires_grad_(True), low.clone().to(DEV).requires_grad_(True)
        stash = {}
        y, _ = ops.FakeQuantFn.apply(w.to(DEV), None, None, None, None, u, l, nbits, seg, symmetric, out_dtype, stash)
        y.backward(gy.to(DEV))
        res.append((y.detach(), stash["scale"], stash["zp"], u.grad, l.grad))
    (y0, s0, z0, gu0, gl0), (y1, s1, z1, gu1, gl1) = res
    same = lambda a, b: torch.equal(torch.nan_to_num(a.float(), nan=7.0), torch.nan_to_num(b.float(), nan=7.0))
    assert same(y0, y1) and same(s0, s1) and same(z0, z1)
    ok = torch.isfinite(gu0) & torch.isfinite(gl0)
    assert torch.equal(ok, torch.isfinite(gu1) & torch.isfinite(gl1))
    for a, b in ((gu0, gu1), (gl0, gl1)):
        d = (a[ok] - b[ok]).abs().max()
        assert float(d) <= 2e-5 * float(a[ok].abs().max()) + 1e-12, float(d)


@pytest.mark.parametrize("shapes", [
    # (N, K) of each dW; T = tokens.  7B: 3088 tiles of 256x256 = 12 rounds + 16 -> one tile column of down_proj is peeled
    [(12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)],
    [(1024, 512), (512, 512), (768, 1024)],                     # small: no peel (rem too large), 3 items
    [(4096, 4096), (200, 512)],                                 # an item the kernel does not take (< 256 rows): single launches
])
def test_wgrad_group_equals_single_launches(shapes):
    """oq_wgrad_group (all weight gradients of a backward pass in one launch of the 256x256x32 kernel, a peeled strip with a split
    contraction) vs one oq_gemm per item: bit-identical outside the peeled strip, fp32-summation-order close inside it."""
    import ctypes
    from omniquant_amd import ops, _capi as C
    T = 2048 if shapes[0][0] > 4096 else 512
    g = torch.Generator().manual_seed(len(shapes) * 7 + T)
    items, refs, keep = [], [], []
    for N, K in shapes:
        gy = (torch.randn(T, N, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
        x = torch.randint(-7, 9, (T, K), generator=g).to(torch.bfloat16).to(DEV)
        ref = torch.empty(N, K, dtype=torch.bfloat16, device=DEV)
        ops.gemm(gy, x, ref, N, K, T, N, K, K, False, False)
        gw = torch.full((N, K), float("nan"), dtype=torch.bfloat16, device=DEV)
        items.append((gy, x, gw, N, K))
        refs.append(ref)
    arr = (C.WgradItem * len(items))()
    for i, (gy, x, gw, N, K) in enumerate(items):
        arr[i] = C.WgradItem(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), N, K, T, N, K, K)
    ws_bytes = C.size_call("oq_wgrad_group_workspace", ctypes.addressof(arr), len(items))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    C.call("oq_wgrad_group", ctypes.addressof(arr), len(items), ws.data_ptr(), ws_bytes, C.stream())
    torch.cuda.synchronize()
    n_diff = 0
    for (gy, x, gw, N, K), ref in zip(items, refs):
        assert bool(torch.isfinite(gw.float()).all()), "an output tile was not written"
        same = gw == ref
        if not bool(same.all()):
            bad = (~same).nonzero()
            rows, cols = bad[:, 0], bad[:, 1]
            # differences only inside ONE strip of 256 rows or columns at the end of ONE item, and at bf16 rounding level
            assert int(rows.min()) >= N - 256 or int(cols.min()) >= K - 256, (N, K, int(rows.min()), int(cols.min()))
            n_diff += 1
            scale = float(ref.float().abs().max())
            assert float((gw.float() - ref.float()).abs().max()) <= 2.0 ** -7 * scale
    assert n_diff <= 1
    if shapes[0] == (12288, 4096):
        # the LLaMA-7B step: the strip exists and went through the split contraction (its values differ in summation order)
        assert n_diff == 1 or all(bool((i[2] == r).all()) for i, r in zip(items, refs))
