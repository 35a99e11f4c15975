"""Attention on the head quantisers' integer grid (oq_attn_fwd_grid / oq_attn_bwd_grid, oq_qkv_rope_quant_fwd(out_grid)).

Reference chain: models/int_llama_layer.py:124-125 (RoPE) -> qkt_matmul.quant_x1 / quant_x2, pv_matmul.quant_x2
(:140-143,161; quantize/quantizer.py:84-147) -> qkt_matmul / softmax(fp32) / pv_matmul (:143-163).  The fake-quantised
q / k / v are coordinate * scale; the grid kernels contract the coordinates and apply the scales in f32, so the forward must
agree with an fp64 evaluation of the reference chain to f32 accuracy (1e-5 here), not to 16-bit accuracy."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _problem(Tn, nh, nkv, seed, hd=128):
    g = torch.Generator().manual_seed(seed)
    nq = torch.randint(-15, 16, (1, Tn, nh, hd), generator=g).float()
    nk = torch.randint(-12, 16, (1, Tn, nkv, hd), generator=g).float()
    nv = torch.randint(-15, 13, (1, Tn, nkv, hd), generator=g).float()
    nht = nh + 2 * nkv
    sc = (0.02 + 0.2 * torch.rand(Tn, nht, 1, generator=g)).float()          # merged per-(token, head) scales
    go = torch.randn(1, Tn, nh, hd, generator=g).to(torch.bfloat16)
    return nq, nk, nv, sc, go


def _reference(nq, nk, nv, sc, go, nh, nkv):
    Tn, hd = nq.shape[1], nq.shape[3]
    sq, sk, sv = sc[:, :nh], sc[:, nh:nh + nkv], sc[:, nh + nkv:]
    q = (nq.double() * sq.double()[None]).requires_grad_(True)               # the fake-quantised VALUES
    k = (nk.double() * sk.double()[None]).requires_grad_(True)
    v = (nv.double() * sv.double()[None]).requires_grad_(True)
    rep = nh // nkv
    kk, vv = k.repeat_interleave(rep, dim=2), v.repeat_interleave(rep, dim=2)
    s = torch.einsum("bthd,bshd->bhts", q, kk) / math.sqrt(hd)
    s = s + torch.triu(torch.full((Tn, Tn), float("-inf"), dtype=torch.float64), 1)
    o = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), vv)
    (o * go.double()).sum().backward()
    return o.detach(), q.grad, k.grad, v.grad


@pytest.mark.parametrize("Tn,nh,nkv", [(128, 2, 2), (256, 4, 2), (512, 3, 1), (1024, 2, 2)])
def test_grid_attention_vs_fp64(Tn, nh, nkv):
    from omniquant_amd import ops
    nq, nk, nv, sc, go = _problem(Tn, nh, nkv, Tn + nh)
    o_ref, gq_ref, gk_ref, gv_ref = _reference(nq, nk, nv, sc, go, nh, nkv)
    scd = sc.to(DEV)
    grid = (scd[:, :nh], scd[:, nh:nh + nkv], scd[:, nh + nkv:])
    qd, kd, vd = (t.to(torch.bfloat16).to(DEV).requires_grad_(True) for t in (nq, nk, nv))
    stash = {}
    o = ops.FusedCausalAttnFn.apply(qd, kd, vd, 1.0 / math.sqrt(128), grid, stash)
    o32 = stash["wide"]
    (o.float() * go.to(DEV).float()).sum().backward()
    sc_o = float(o_ref.abs().max())
    err32 = float((o32.double().cpu() - o_ref).abs().max()) / sc_o
    err16 = float((o.double().cpu() - o_ref).abs().max()) / sc_o
    assert err32 < 1e-5, f"f32 output: {err32}"
    assert err16 < 6e-3, f"bf16 copy: {err16}"
    assert torch.equal(o, o32.to(torch.bfloat16)), "the bf16 output is the rounded f32 output"
    for name, a, r in (("gq", qd.grad, gq_ref), ("gk", kd.grad, gk_ref), ("gv", vd.grad, gv_ref)):
        a, r = a.double().cpu().reshape(-1), r.reshape(-1)
        cos = float(torch.dot(a, r) / (a.norm() * r.norm()))
        rel = float((a - r).norm() / r.norm())
        assert cos > 0.9999 and rel < 1e-2, f"{name}: cos {cos} rel {rel}"      # bf16 gradient operands (dO, dS, P)


def test_grid_attention_is_deterministic_and_matches_value_kernels_loosely():
    """Same problem through the value kernels (q, k, v rounded to bf16): the two paths agree at bf16 level, and the grid path is
    bit-reproducible run to run."""
    from omniquant_amd import ops
    Tn, nh, nkv = 512, 4, 4
    nq, nk, nv, sc, go = _problem(Tn, nh, nkv, 7)
    scd = sc.to(DEV)
    grid = (scd[:, :nh], scd[:, nh:nh + nkv], scd[:, nh + nkv:])
    outs = []
    for _ in range(2):
        qd, kd, vd = (t.to(torch.bfloat16).to(DEV).requires_grad_(True) for t in (nq, nk, nv))
        o = ops.FusedCausalAttnFn.apply(qd, kd, vd, 1.0 / math.sqrt(128), grid, {})
        (o.float() * go.to(DEV).float()).sum().backward()
        outs.append((o.detach().clone(), qd.grad.clone(), kd.grad.clone(), vd.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    qv, kv, vv = ((n.to(DEV) * s[None]).to(torch.bfloat16).requires_grad_(True) for n, s in zip((nq, nk, nv), grid))
    ov = ops.FusedCausalAttnFn.apply(qv, kv, vv, 1.0 / math.sqrt(128))
    (ov.float() * go.to(DEV).float()).sum().backward()
    for name, a, b in zip(("o", "gq", "gk", "gv"), outs[0], (ov, qv.grad, kv.grad, vv.grad)):
        rel = float((a.float() - b.float()).norm() / b.float().norm())
        assert rel < 3e-2, f"{name}: {rel}"


def test_rope_quant_grid_output_times_scale_is_the_value_output():
    """oq_qkv_rope_quant_fwd(out_grid=1) stores clamp(round(x/s)+z, 0, Q) - z; times the scale it is bit-for-bit the f32 value
    output of the same call without the flag; the coordinates are integers in [-Q, Q] + the zero-point's range."""
    from omniquant_amd import _capi as C
    rows, T, nhq, nhk, hd = 256, 256, 4, 2, 128
    nht = nhq + 2 * nhk
    g = torch.Generator().manual_seed(11)
    pre = (torch.randn(rows, nht * hd, generator=g) * 2).to(DEV)
    pos = torch.arange(T, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    emb = torch.cat([torch.outer(pos, inv)] * 2, dim=-1)
    cos, sin = emb.cos().contiguous().to(DEV), emb.sin().contiguous().to(DEV)

    def run(out_grid, dtype):
        ys = [torch.empty((rows, n * hd), dtype=dtype, device=DEV) for n in (nhq, nhk, nhk)]
        vec = [torch.empty((rows * nht, 1), dtype=torch.float32, device=DEV) for _ in range(4)]
        C.call("oq_qkv_rope_quant_fwd", C.ptr(pre), C.dt(pre), rows, T, nhq, nhk, nhk, hd, C.fptr(cos), C.fptr(sin), 4,
               C.ptr(ys[0]), C.ptr(ys[1]), C.ptr(ys[2]), C.dt(ys[0]), out_grid, *(C.fptr(v) for v in vec), C.stream())
        return ys, vec

    yv, vec_v = run(0, torch.float32)
    yg, vec_g = run(1, torch.bfloat16)
    for a, b in zip(vec_v, vec_g):
        assert torch.equal(a, b)
    scale = vec_g[0].view(rows, nht)
    h0 = 0
    for y_val, y_grid, n in zip(yv, yg, (nhq, nhk, nhk)):
        coords = y_grid.float().view(rows, n, hd)
        assert torch.equal(coords, coords.round()) and float(coords.abs().max()) <= 15.0
        assert torch.equal(coords * scale[:, h0:h0 + n, None], y_val.view(rows, n, hd))
        h0 += n


def test_grid_entry_points_refuse_bad_arguments():
    """Error behaviour of the new entry points: an OQError with a message, nothing launched."""
    from omniquant_amd import _capi as C
    T, nh, hd = 128, 2, 128
    q = torch.zeros(1, T, nh, hd, dtype=torch.bfloat16, device=DEV)
    o = torch.empty_like(q)
    lse = torch.empty(1, nh, T, dtype=torch.float32, device=DEV)
    sc = torch.ones(T, nh, dtype=torch.float32, device=DEV)
    with pytest.raises(C.OQError, match="stride"):
        C.call("oq_attn_fwd_grid", C.ptr(q), C.ptr(q), C.ptr(q), C.fptr(sc), C.fptr(sc), C.fptr(sc), 1, C.ptr(o), None,
               C.fptr(lse), 1, T, nh, nh, hd, 0.1, 1, C.stream())
    with pytest.raises(C.OQError, match="head_dim"):
        C.call("oq_attn_fwd_grid", C.ptr(q), C.ptr(q), C.ptr(q), C.fptr(sc), C.fptr(sc), C.fptr(sc), nh, C.ptr(o), None,
               C.fptr(lse), 1, T, nh, nh, 64, 0.1, 1, C.stream())
    with pytest.raises(C.OQError, match="null"):
        C.call("oq_attn_fwd_grid", C.ptr(q), C.ptr(q), C.ptr(q), None, C.fptr(sc), C.fptr(sc), nh, C.ptr(o), None,
               C.fptr(lse), 1, T, nh, nh, hd, 0.1, 1, C.stream())
    pre = torch.zeros(T, 3 * nh * hd, dtype=torch.float32, device=DEV)
    cs = torch.ones(T, hd, dtype=torch.float32, device=DEV)
    ys = [torch.empty(T, nh * hd, dtype=torch.bfloat16, device=DEV) for _ in range(3)]
    with pytest.raises(C.OQError, match="grid coordinates"):
        C.call("oq_qkv_rope_quant_fwd", C.ptr(pre), C.dt(pre), T, T, nh, nh, nh, hd, C.fptr(cs), C.fptr(cs), 16,
               C.ptr(ys[0]), C.ptr(ys[1]), C.ptr(ys[2]), C.dt(ys[0]), 1, None, None, None, None, C.stream())


def test_grid_attention_nan_segment_poisons_what_attends_to_it():
    """Quirk Q1 (a head segment whose values are all equal: scale 0 -> NaN) reaches the grid kernels as NaN coordinates.  A NaN
    QUERY segment gives a NaN output row for that (token, head) only; a NaN KEY segment poisons every query at or after its
    position (the causal kernels never touch masked keys: DESIGN.md section 4, deviation 3c) and nothing before it; other heads
    are untouched."""
    from omniquant_amd import ops
    Tn, nh = 256, 2
    nq, nk, nv, sc, go = _problem(Tn, nh, nh, 3)
    nq[0, 40, 0, :] = float("nan")            # query 40 of head 0
    nk[0, 100, 1, :] = float("nan")           # key 100 of head 1
    scd = sc.to(DEV)
    grid = (scd[:, :nh], scd[:, nh:2 * nh], scd[:, 2 * nh:])
    qd, kd, vd = (t.to(torch.bfloat16).to(DEV) for t in (nq, nk, nv))
    stash = {}
    ops.FusedCausalAttnFn.apply(qd, kd, vd, 1.0 / math.sqrt(128), grid, stash)
    o = stash["wide"][0].cpu()                # [T, nh, hd]
    row_nan = torch.isnan(o).all(dim=-1)      # [T, nh]
    row_fin = torch.isfinite(o).all(dim=-1)
    assert bool(row_nan[40, 0]) and bool(row_fin[:40, 0].all()) and bool(row_fin[41:, 0].all())
    assert bool(row_fin[:100, 1].all()) and bool(row_nan[100:, 1].all())
