"""Pins oracle/ref_cpu.py to the reference: every golden vector in tests/golden/*.npz was produced by
importing the reference's own modules (tests/golden/gen_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import ref_cpu as R

torch.set_num_threads(4)


def T(a):
    return torch.from_numpy(np.asarray(a)).float()


def close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert (nan_a == nan_b).all(), f"{what}: NaN pattern differs"
    np.testing.assert_allclose(a[~nan_a], b[~nan_b], rtol=rtol, atol=atol, err_msg=what)


G1, G1META = load_golden("g1_quantizer.npz")


@pytest.mark.parametrize("i", range(len(G1META["cases"])))
def test_quantizer_forward_backward(i):
    c = G1META["cases"][i]
    x = T(G1[f"c{i}_x"]).requires_grad_(f"c{i}_gx" in G1)
    up = T(G1[f"c{i}_up"]).requires_grad_(True) if c["lwc"] else None
    low = T(G1[f"c{i}_low"]).requires_grad_(True) if c["lwc"] else None
    y, s, z = R.fake_quant(x, c["n_bits"], c["group_size"], up, low, c["symmetric"], return_qparams=True)
    close(y.detach(), G1[f"c{i}_y"], what=c["tag"] + " y", rtol=0, atol=0)      # bit-exact
    close(s.detach(), G1[f"c{i}_scale"], what=c["tag"] + " scale", rtol=0, atol=0)
    close(z.detach(), G1[f"c{i}_zp"], what=c["tag"] + " zp", rtol=0, atol=0)
    if y.requires_grad:
        (y * T(G1[f"c{i}_G"])).sum().backward()
        if x.grad is not None:
            close(x.grad, G1[f"c{i}_gx"], what=c["tag"] + " gx", rtol=1e-6, atol=1e-7)
        if c["lwc"]:
            close(up.grad, G1[f"c{i}_gup"], what=c["tag"] + " gup", rtol=1e-5, atol=1e-6)
            close(low.grad, G1[f"c{i}_glow"], what=c["tag"] + " glow", rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("i", [i for i, c in enumerate(G1META["cases"])
                               if not c["symmetric"] and f"c{i}_gx" in G1 and c["n_bits"] < 16])
def test_quantizer_closed_form_backward(i):
    c = G1META["cases"][i]
    x = T(G1[f"c{i}_x"])
    up = T(G1[f"c{i}_up"]) if c["lwc"] else None
    low = T(G1[f"c{i}_low"]) if c["lwc"] else None
    gx, gup, glow = R.fake_quant_backward(x, T(G1[f"c{i}_G"]), c["n_bits"], c["group_size"], up, low)
    # gs = sum G*((q-z) - x/s) cancels catastrophically when |x/s| is large (x ~ 100, s ~ 1/15):
    # fp32 noise there is ~1e-7 * |x/s| per element, in the reference's autograd as well.
    atol = 5e-4 if c["tag"] == "a4_tok_positive" else 2e-5
    close(gx, G1[f"c{i}_gx"], what=c["tag"] + " gx", rtol=1e-5, atol=atol)
    if c["lwc"]:
        close(gup, G1[f"c{i}_gup"], what=c["tag"] + " gup", rtol=1e-4, atol=2e-5)
        close(glow, G1[f"c{i}_glow"], what=c["tag"] + " glow", rtol=1e-4, atol=2e-5)


def test_truncate_and_let_init():
    g, _ = load_golden("g5_misc.npz")
    close(R.truncate_small(T(g["trunc_x"])), g["trunc_y"], rtol=0, atol=0)
    for alpha in (0.5, 0.75):
        close(R.let_init_scale(T(g["let_act"]), T(g["let_W"]), alpha), g[f"let_scale_a{alpha}"], rtol=1e-6)


def _spec(meta):
    return R.QuantSpec(meta["wbits"], meta["abits"], meta["group_size"], meta["lwc"], meta["let"])


def _block_from_step(g, meta):
    weights = {k[2:]: T(v) for k, v in g.items() if k.startswith("w.")}
    blk = R.Block(meta["family"], meta["config"], weights, _spec(meta))
    if meta["let"]:
        sc = {k[len("act_scales."):]: T(v) for k, v in g.items() if k.startswith("act_scales.")}
        sh = {k[len("act_shifts."):]: T(v) for k, v in g.items() if k.startswith("act_shifts.")}
        blk.register_let(sc, sh, meta["alpha"], 0, meta["layer_prefix"])
    for n in list(blk.params.keys()):
        blk.params[n] = T(g["p0." + n]).requires_grad_(True)
    return blk


STEP_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g3_step_*.npz")))


@pytest.mark.parametrize("fname", STEP_FILES)
def test_block_step(fname):
    g, meta = load_golden(fname)
    blk = _block_from_step(g, meta)
    assert {("p0." + n) for n in blk.params} == {k for k in g if k.startswith("p0.")}
    x, tgt, mask = T(g["x"]), T(g["target"]), T(g["mask"])
    pos = torch.from_numpy(g["position_ids"])
    temps = blk.temporaries()
    for k, v in g.items():
        if k.startswith("tmp."):
            mod, kind = k[4:].rsplit(".temp_", 1)
            close(temps[f"{mod}.{kind}"].detach(), v, rtol=1e-6, atol=1e-7, what=k)
        if k.startswith("p_trunc."):
            close(blk.params[k[8:]].detach(), v, rtol=0, atol=0, what=k)
    out = blk.forward(x, mask, pos, temps=temps, act_quant=True)
    close(out.detach(), g["out"], rtol=2e-4, atol=2e-5, what="out")
    loss = torch.nn.functional.mse_loss(tgt, out)
    close(loss.detach().reshape(1), g["loss"].reshape(1), rtol=1e-5, what="loss")
    loss.backward()
    for n, p in blk.params.items():
        ref = g["grad." + n]
        scale = max(np.abs(ref).max(), 1e-12)
        close(p.grad / scale, ref / scale, rtol=2e-3, atol=2e-5, what="grad " + n)
    with torch.no_grad():
        close(blk.forward(x, mask, pos, temps=None, act_quant=False), g["out_fp"], rtol=1e-5, atol=1e-5, what="fp")


@pytest.mark.parametrize("fname", ["g3_step_llama_hd128_w4a4_lwc_let.npz", "g3_step_llama_w4a4_lwc_let.npz",
                                   "g3_step_opt_w4a4_lwc_let.npz"])
@pytest.mark.parametrize("int_fprop,wide", [(False, False), (True, False), (True, True)])
def test_storage_model_is_the_pinned_step_when_nothing_is_rounded(fname, int_fprop, wide):
    """The precision-mode emulation of oracle/ref_cpu.py (Block.forward(act_dtype=..., int_fprop=..., wide=...),
    temporaries(store_dtype, int_fprop)) is what the production-mode GPU tests are held against.  It must be the reference-pinned fp32 step in
    everything but its rounding points: with float32 as the "storage" dtype every rounding is the identity, and the model
    -- fused-attention branch, integer-fprop Linear (_IntFpropLinear), fused producer branches and all -- has to reproduce
    the golden step: output, loss and every gradient, at the fixture's own tolerances."""
    g, meta = load_golden(fname)
    blk = _block_from_step(g, meta)
    x, tgt, mask = T(g["x"]), T(g["target"]), T(g["mask"])
    pos = torch.from_numpy(g["position_ids"])
    temps = blk.temporaries(store_dtype=torch.float32, int_fprop=int_fprop)
    out = blk.forward(x, mask, pos, temps=temps, act_quant=True, act_dtype=torch.float32, int_fprop=int_fprop, wide=wide)
    # (the model's fused-attention branch evaluates the softmax as exp(s - max) / sum: fp32 op-order noise of a few 1e-5)
    close(out.detach(), g["out"], rtol=1e-3, atol=1e-4, what="out")
    loss = torch.nn.functional.mse_loss(tgt, out)
    close(loss.detach().reshape(1), g["loss"].reshape(1), rtol=1e-5, what="loss")
    loss.backward()
    for n, p in blk.params.items():
        ref = g["grad." + n]
        scale = max(np.abs(ref).max(), 1e-12)
        close(p.grad / scale, ref / scale, rtol=5e-3, atol=5e-5, what="grad " + n)


TRAJ_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g4_traj_*.npz")))


@pytest.mark.parametrize("fname", TRAJ_FILES)
def test_trajectory(fname):
    g, m = load_golden(fname)
    layers = [{k[len(f"w{i}."):]: T(v) for k, v in g.items() if k.startswith(f"w{i}.")} for i in range(m["n_layers"])]
    sc = {k[len("act_scales."):]: T(v) for k, v in g.items() if k.startswith("act_scales.")}
    sh = {k[len("act_shifts."):]: T(v) for k, v in g.items() if k.startswith("act_shifts.")}
    res = R.calibrate(m["family"], m["config"], layers, _spec(m), T(g["inps"]), T(g["mask"]),
                      torch.from_numpy(g["position_ids"]), sc, sh, epochs=m["epochs"], let_lr=m["let_lr"],
                      lwc_lr=m["lwc_lr"], alpha=m["alpha"], aug_loss=m["aug_loss"], prefix=m["layer_prefix"],
                      batch_size=m.get("batch_size", 1))
    close(res["losses"], g["losses"], rtol=2e-3, what="losses")
    close(res["norms"], g["norms"], rtol=2e-2, what="norms")
    for i in range(m["n_layers"]):
        close(res["fp_out"][i], g[f"fp_out.{i}"], rtol=1e-4, atol=1e-4, what=f"fp_out {i}")
        keys = [k for k in g if k.startswith(f"omni.{i}.")]
        assert {k[len(f"omni.{i}."):] for k in keys} == set(res["omni"][i].keys())
        for k in keys:
            n = k[len(f"omni.{i}."):]
            ref = g[k].astype(np.float64)
            got = res["omni"][i][n].float().numpy()
            assert res["omni"][i][n].dtype == torch.float16
            # north-star bar: learned tensors within 1e-3 relative (to the tensor's scale)
            tol = 1e-3 * max(np.abs(ref).max(), 1e-6)
            assert np.abs(got - ref).max() <= tol + 1e-3 * 0, f"{n}: {np.abs(got-ref).max()} > {tol}"
        close(res["quant_out"][i], g[f"quant_out.{i}"], rtol=5e-2, atol=5e-2, what=f"quant_out {i}")
        # ---- fp32 learned tensors before the fp16 cast (trained32.*) -----------------------------------------------
        for k in [k for k in g if k.startswith(f"trained32.{i}.")]:
            n = k[len(f"trained32.{i}."):]
            ref = g[k].astype(np.float64)
            got = res["trained"][i][n].double().numpy().reshape(ref.shape)
            assert np.abs(got - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1e-6), f"trained32 {n}"
        # ---- the folded model (models/int_llama_layer.py:315-332,365-368): weights, LET biases, norm parameters,
        #      weight_quantizer.scales / zeros as registered by register_scales_and_zeros --------------------------------
        fold = res["folded"][i]
        for k in [k for k in g if k.startswith(f"folded32.{i}.")]:
            n = k[len(f"folded32.{i}."):]
            ref = g[k].astype(np.float64)
            if n.endswith("weight_quantizer.scales") or n.endswith("weight_quantizer.zeros"):
                lin = n.rsplit(".weight_quantizer.", 1)[0]
                sc_, zp_ = res["qparams"][i][lin]
                got = (sc_ if n.endswith("scales") else zp_).double().numpy().reshape(ref.shape)
                if n.endswith("zeros"):
                    # an integer: a 1e-3-level difference in the learned bounds may move it by one for a few rows
                    assert np.abs(got - ref).max() <= 1.0 and (got != ref).mean() <= 0.02, f"folded zeros {n}"
                else:
                    assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), f"folded scales {n}"
                continue
            if n not in fold:
                assert np.abs(ref).max() == 0.0, f"folded32 {n} missing from the oracle's fold"
                continue
            got = fold[n].double().numpy().reshape(ref.shape)
            if n.endswith("proj.weight") or n.endswith("fc1.weight") or n.endswith("fc2.weight"):
                # fake-quantised values: equal up to 1e-3 of the tensor except the few elements whose rounding decision
                # flipped (one quantisation step)
                step = np.abs(ref).max() / (2 ** (m["wbits"] - 1))
                d = np.abs(got - ref)
                assert d.max() <= 2.5 * step and (d > 2e-3 * np.abs(ref).max()).mean() <= 0.01, f"folded weight {n}"
            else:
                assert np.abs(got - ref).max() <= 1e-3 * max(np.abs(ref).max(), 1e-6), f"folded {n}"


def test_act_stats_vs_reference_prepass():
    """G6: oracle restatement of generate_act_scale_shift.py:25-94 vs the reference's own functions (bit-exact)."""
    g, _ = load_golden("g6_act_stats.npz")
    for k in ("fc1", "fc2"):
        x = torch.from_numpy(g[f"x_{k}"])
        sc, sh = R.act_stats([x[i:i + 1] for i in range(x.shape[0])])
        np.testing.assert_array_equal(sc.numpy(), g[f"scale_{k}"])
        np.testing.assert_array_equal(sh.numpy(), g[f"shift_{k}"])
