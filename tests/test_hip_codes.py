"""Integer side channel of the quantiser kernels (ops.IntCodes): the int8 grid codes and per-row code sums that
oq_gemm_i8 contracts.  quantize/quantizer.py:93-99 computes x_int = clamp(round(x / scale) + zp, 0, 2^n - 1) and returns
(x_int - zp) * scale; the kernels store x_int next to that value, so  (codes - zp) * scale == y  must hold bit for bit in
fp32 (and after the single bf16 rounding in production mode), the codes must lie on the grid, and csum must be their row sum
(NaN for a row whose scale is 0 / NaN: the reference's all-NaN row, quirk Q1)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check(y, ic, nbits, nan_rows=()):
    codes = ic.codes.float() + (128.0 if nbits == 8 else 0.0)          # grid codes
    rows = codes.shape[0]
    ok = torch.ones(rows, dtype=torch.bool, device=codes.device)
    for r in nan_rows:
        ok[r] = False
    assert codes[ok].min() >= 0 and codes[ok].max() <= 2 ** nbits - 1
    rec = (codes - ic.zp[:, None]) * ic.scale[:, None]
    y2 = y.reshape(rows, -1)
    assert torch.equal(rec.to(y2.dtype)[ok], y2[ok]), "(codes - zp) * scale != y"
    want = ic.codes.double().sum(1)
    assert torch.equal(ic.csum.double()[ok], want[ok]), "csum != row sum of the stored codes"
    for r in nan_rows:
        assert torch.isnan(ic.csum[r]) and torch.isnan(y2[r]).all()


@pytest.mark.parametrize("rows,cols", [(64, 4096), (37, 11008), (16, 512), (2048, 4096), (130, 1024), (5, 5120), (9, 13824)])
@pytest.mark.parametrize("nbits", [4, 8, 6])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_per_token_codes(rows, cols, nbits, dtype):
    from omniquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(rows + cols + nbits)
    x = (torch.randn(rows, cols, device=DEV, generator=g) * torch.exp(0.5 * torch.randn(cols, device=DEV, generator=g))).to(dtype)
    x[0] = x[0].abs() + 3.0               # an all-positive row: zero-point below the grid
    if rows > 4:
        x[3] = 1.25                       # constant row: scale 0 -> NaN row (quirk Q1)
    stash = {"want_int": True}
    y, _ = ops.FakeQuantFn.apply(x, None, None, None, None, None, None, nbits, cols, False, dtype, stash)
    assert "int" in stash
    _check(y, stash["int"], nbits, nan_rows=(3,) if rows > 4 else ())
    # identical values with and without the side channel
    y0, _ = ops.FakeQuantFn.apply(x, None, None, None, None, None, None, nbits, cols, False, dtype, {})
    assert torch.equal(torch.nan_to_num(y0.float()), torch.nan_to_num(y.float()))


@pytest.mark.parametrize("rows,cols", [(4096, 4096), (301, 4096), (64, 5120), (11008, 4096), (3, 4096), (768, 768), (130, 2048),
                                       (17, 8192), (5, 512)])
@pytest.mark.parametrize("mode", ["lwc", "let_rd", "let_rm", "let_plain"])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_weight_codes(rows, cols, mode, out_dtype):
    from omniquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(rows * 3 + cols)
    w = (torch.randn(rows, cols, device=DEV, generator=g) * 0.02).half()
    up = torch.full((rows, 1), 4.0, device=DEV) + 0.3 * torch.randn(rows, 1, device=DEV, generator=g)
    low = torch.full((rows, 1), 4.0, device=DEV) + 0.3 * torch.randn(rows, 1, device=DEV, generator=g)
    cm = rd = rm = sh = None
    if mode != "lwc":
        cm = torch.rand(cols, device=DEV, generator=g) + 0.5
        sh = torch.randn(cols, device=DEV, generator=g)
        if mode == "let_rd":
            rd = torch.rand(rows, device=DEV, generator=g) + 0.5
        if mode == "let_rm":
            rm = torch.rand(rows, device=DEV, generator=g) + 0.5
    stash = {"want_int": True}
    y, _ = ops.FakeQuantFn.apply(w, cm, rd, rm, sh, up, low, 4, cols, False, out_dtype, stash)
    assert "int" in stash
    _check(y, stash["int"], 4)
    y0, _ = ops.FakeQuantFn.apply(w, cm, rd, rm, sh, up, low, 4, cols, False, out_dtype, {})
    assert torch.equal(y0, y)


def test_weight_codes_multi_launch_into_stacked_destinations():
    """q, k, v, o in ONE multi-matrix launch, q | k | v writing values, codes and per-row vectors into stacked buffers
    (block_common._weight_slabs): the stacked IntCodes must equal the three single-matrix results."""
    from omniquant_amd import ops
    n, K = 512, 4096
    g = torch.Generator(device=DEV).manual_seed(0)
    ws = [(torch.randn(n, K, device=DEV, generator=g) * 0.02).half() for _ in range(4)]
    cm = torch.rand(K, device=DEV, generator=g) + 0.5
    sh = torch.randn(K, device=DEV, generator=g)
    rds = [torch.rand(n, device=DEV, generator=g) + 0.5, None, torch.rand(n, device=DEV, generator=g) + 0.5, None]
    rms = [None, torch.rand(n, device=DEV, generator=g) + 0.5, None, None]
    ups = [torch.full((n, 1), 4.0, device=DEV) for _ in range(4)]
    slab = torch.empty(3 * n, K, dtype=torch.bfloat16, device=DEV)
    codes = torch.empty(3 * n, K, dtype=torch.int8, device=DEV)
    vec = torch.empty(3, 3 * n, device=DEV)
    dests = [ops.WeightDest(slab[i * n:(i + 1) * n], None, codes[i * n:(i + 1) * n], vec[0, i * n:(i + 1) * n].view(n, 1),
                            vec[1, i * n:(i + 1) * n].view(n, 1), vec[2, i * n:(i + 1) * n]) for i in range(3)] + [None]
    stashes = [{"want_int": True} for _ in range(4)]
    outs = []
    with ops.WeightQuantBatch():
        for i in range(4):
            outs.append(ops.FakeQuantFn.apply(ws[i], cm, rds[i], rms[i], sh, ups[i], ups[i], 4, K, False, torch.bfloat16, stashes[i],
                                              dests[i])[0])
    ints = [s["int"] for s in stashes]
    for i in range(4):
        _check(outs[i], ints[i], 4)
        st = {"want_int": True}
        y1, _ = ops.FakeQuantFn.apply(ws[i], cm, rds[i], rms[i], sh, ups[i], ups[i], 4, K, False, torch.bfloat16, st)
        assert torch.equal(y1, outs[i]) and torch.equal(st["int"].codes, ints[i].codes) and torch.equal(st["int"].csum, ints[i].csum)
    allq = ops.stacked_int(ints[:3])
    assert allq is not None and tuple(allq.codes.shape) == (3 * n, K) and allq.codes.data_ptr() == codes.data_ptr()
    assert torch.equal(allq.csum, vec[2]) and torch.equal(allq.scale, vec[0])


@pytest.mark.parametrize("rows,cols", [(2048, 4096), (77, 4096), (33, 5120), (10, 8192), (6, 1024)])
@pytest.mark.parametrize("is_ln", [False, True])
def test_norm_quant_codes(rows, cols, is_ln):
    from omniquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(rows + cols)
    x = (torch.randn(rows, cols, device=DEV, generator=g) * 2).bfloat16()
    w = 1 + 0.1 * torch.randn(cols, device=DEV, generator=g)
    b = 0.1 * torch.randn(cols, device=DEV, generator=g) if is_ln else None
    stash = {"want_int": True}
    y, _ = ops.NormQuantFn.apply(x, w, b, 1e-6, is_ln, 4, stash)
    _check(y, stash["int"], 4)
    y0, _ = ops.NormQuantFn.apply(x, w, b, 1e-6, is_ln, 4, {})
    assert torch.equal(y0, y)


@pytest.mark.parametrize("rows,I", [(2048, 11008), (19, 11008), (40, 13824), (8, 1024)])
@pytest.mark.parametrize("pre_dtype", [torch.bfloat16, torch.float32])
def test_silu_mul_quant_codes_and_f32_preactivations(rows, I, pre_dtype):
    """The fused silu * up -> down_proj input quantiser on the column blocks of a stacked gate | up buffer: codes, and fp32
    pre-activations with bf16 outputs / gradients (the integer path keeps the exact projection result in fp32)."""
    from omniquant_amd import _capi as C
    from omniquant_amd import ops
    g = torch.Generator(device=DEV).manual_seed(rows + I)
    pre = torch.randn(rows, 2 * I, device=DEV, generator=g).to(pre_dtype)
    y = torch.empty(rows, I, dtype=torch.bfloat16, device=DEV)
    scale, zp, xmin, xmax = (torch.empty(rows, 1, device=DEV) for _ in range(4))
    codes = torch.empty(rows, I, dtype=torch.int8, device=DEV)
    csum = torch.empty(rows, device=DEV)
    es = pre.element_size()
    C.call("oq_silu_mul_quant_fwd", pre.data_ptr(), pre.data_ptr() + I * es, C.dt(pre), rows, I, 2 * I, 4, C.ptr(y), C.dt(y),
           C.fptr(scale), C.fptr(zp), C.fptr(xmin), C.fptr(xmax), C.ptr(codes), C.fptr(csum), C.stream())
    ic = ops.IntCodes(codes, scale.view(-1), zp.view(-1), csum, 4)
    _check(y, ic, 4)
    # value check against torch on the same pre-activations
    p32 = pre.float()
    prod = torch.nn.functional.silu(p32[:, :I]) * p32[:, I:]
    lo, hi = prod.amin(1, keepdim=True), prod.amax(1, keepdim=True)
    s = (hi - lo) / 15
    z = (-lo / s).round()
    ref = ((prod / s).round() + z).clamp(0, 15)
    assert (ref != (codes.float())).float().mean() < 2e-3          # silu via v_exp / v_rcp: a few elements sit on the other side
    # backward with bf16 gradients: on bf16-representable pre-activations the f32-input kernels must reproduce the bf16-input
    # kernels bit for bit (same arithmetic, only the load differs)
    if pre_dtype != torch.float32:
        return
    pre_b = pre.bfloat16()
    pre_f = pre_b.float()
    res = []
    for pz in (pre_b, pre_f):
        y_, sc_, zp_, mn_, mx_ = torch.empty_like(y), *(torch.empty(rows, 1, device=DEV) for _ in range(4))
        e = pz.element_size()
        C.call("oq_silu_mul_quant_fwd", pz.data_ptr(), pz.data_ptr() + I * e, C.dt(pz), rows, I, 2 * I, 4, C.ptr(y_), C.dt(y_),
               C.fptr(sc_), C.fptr(zp_), C.fptr(mn_), C.fptr(mx_), None, None, C.stream())
        gy = torch.randn(rows, I, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7)).bfloat16()
        gpre = torch.empty(rows, 2 * I, dtype=torch.bfloat16, device=DEV)
        C.call("oq_silu_mul_quant_bwd", pz.data_ptr(), pz.data_ptr() + I * e, C.ptr(gy), C.dt(pz), C.dt(gy), rows, I, 2 * I, 4,
               C.fptr(mn_), C.fptr(mx_), gpre.data_ptr(), gpre.data_ptr() + I * 2, C.stream())
        res.append((y_, sc_, zp_, gpre))
    for a, b in zip(*res):
        assert torch.equal(a, b)
