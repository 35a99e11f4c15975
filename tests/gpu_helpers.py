"""Helpers shared by the -m gpu parity tests and __graft_entry__.smoke(): build HIP blocks from the golden
fixtures and compare with the reference-generated vectors / the CPU oracle."""
import numpy as np
import torch

from conftest import load_golden


def T(a, device="cpu", dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype).to(device)


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double().reshape(a.shape)
    den = max(float(b.abs().max()), 1e-12)
    return float((a - b).abs().max()) / den


def make_args(meta):
    from omniquant_amd.calibrate import default_args
    return default_args(wbits=meta["wbits"], abits=meta["abits"], group_size=meta["group_size"], lwc=meta["lwc"],
                        let=meta["let"], net=meta["family"], alpha=meta.get("alpha", 0.5),
                        aug_loss=meta.get("aug_loss", False), epochs=meta.get("epochs", 1),
                        let_lr=meta.get("let_lr", 5e-3), lwc_lr=meta.get("lwc_lr", 1e-2),
                        nsamples=meta.get("nsamples", 1), batch_size=meta.get("batch_size", 1))


def make_cfg(meta):
    from omniquant_amd.synthetic import make_config
    c = meta["config"]
    if meta["family"] == "llama":
        return make_config(None, family="llama", hidden_size=c["hidden_size"], inter=c["intermediate_size"],
                           heads=c["num_attention_heads"], kv_heads=c["num_key_value_heads"],
                           rms_norm_eps=c["rms_norm_eps"])
    return make_config(None, family="opt", hidden_size=c["hidden_size"], inter=c["ffn_dim"],
                       heads=c["num_attention_heads"], kv_heads=c["num_attention_heads"])


def build_block(g, meta, wprefix, device, dtype):
    """HIP block from fixture weights (fp16 master exactly as the reference got them)."""
    from omniquant_amd.synthetic import make_layer
    from omniquant_amd.calibrate import decoder_layer_class
    cfg = make_cfg(meta)
    weights = {k[len(wprefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(wprefix)}
    layer = make_layer(cfg, weights=weights, device=device)
    args = make_args(meta)
    q = decoder_layer_class(meta["family"])(cfg, layer, args).to(device)
    q.compute_dtype = dtype
    return q, cfg, args


def run_block_step_parity(fname, dtype=torch.float32, device="cuda:0", keep_grads=False):
    """One sample-step on the HIP path vs the golden vectors of the reference.  Returns max relative errors."""
    from omniquant_amd.calibrate import register_let_parameters
    g, meta = load_golden(fname)
    q, cfg, args = build_block(g, meta, "w.", device, dtype)
    q.set_quant_state(weight_quant=False, act_quant=True)
    q.let = meta["let"]
    if meta["let"]:
        sc = {k[len("act_scales."):]: T(v) for k, v in g.items() if k.startswith("act_scales.")}
        sh = {k[len("act_shifts."):]: T(v) for k, v in g.items() if k.startswith("act_shifts.")}
        register_let_parameters(q, meta["family"], sc, sh, meta["alpha"], 0, device)
    with torch.no_grad():
        for n, p in q.named_parameters():
            p.data = T(g["p0." + n], device).reshape(p.shape)
    x, tgt = T(g["x"], device, dtype), T(g["target"], device, dtype)
    mask = T(g["mask"], device)
    pos = torch.from_numpy(g["position_ids"]).to(device)
    q.smooth_and_quant_temporary()
    errs = {"tmp": 0.0}
    for k, v in g.items():
        if k.startswith("tmp."):
            mod, kind = k[4:].rsplit(".temp_", 1)
            got = getattr(q.get_submodule(mod), "temp_" + kind)
            errs["tmp"] = max(errs["tmp"], rel_err(got.float(), v))
    out = q(x, attention_mask=mask, position_ids=pos)[0] if meta["family"] == "llama" else q(x, attention_mask=mask)[0]
    errs["out"] = rel_err(out.float(), g["out"])
    loss = torch.nn.functional.mse_loss(tgt.float(), out.float())
    errs["loss"] = abs(float(loss) - float(g["loss"].reshape(-1)[0])) / abs(float(g["loss"].reshape(-1)[0]))
    errs["loss_value"] = float(loss)
    loss.backward()
    errs["grad"] = 0.0
    errs["per_grad"] = {}
    errs["cos"], errs["l2"] = {}, {}          # per-tensor cosine / relative L2 error (what bf16 mode is judged by)
    for n, p in q.named_parameters():
        e = rel_err(p.grad, g["grad." + n])
        errs["per_grad"][n] = e
        errs["grad"] = max(errs["grad"], e)
        a = p.grad.detach().double().cpu().reshape(-1)
        b = torch.as_tensor(np.asarray(g["grad." + n])).double().reshape(-1)
        errs["cos"][n] = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
        errs["l2"][n] = float((a - b).norm() / (b.norm() + 1e-300))
    if keep_grads:
        errs["grads"] = {n: p.grad.detach().float().cpu().clone() for n, p in q.named_parameters()}
    return errs


def fp16_ulp(ref):
    """ulp of float16 at |ref| (elementwise, numpy float64): |fp16(a) - fp16(b)| <= |a - b| + ulp."""
    a = np.maximum(np.abs(np.asarray(ref, np.float64)), 2.0 ** -14)
    return 2.0 ** (np.floor(np.log2(a)) - 10)


def assert_learned_close(got, ref, what, rel=1e-3):
    """North-star bar for a learned tensor stored in float16: 1e-3 of the tensor's magnitude plus one fp16 ulp of
    each element (two fp32 values closer than that can still round to neighbouring float16 numbers)."""
    got = np.asarray(got, np.float64).reshape(np.asarray(ref).shape)
    ref = np.asarray(ref, np.float64)
    tol = rel * max(np.abs(ref).max(), 1e-6) + fp16_ulp(ref)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{what}: {bad.sum()}/{bad.size} beyond 1e-3*max + 1 fp16 ulp, worst {np.abs(got - ref).max():.3e} " \
                          f"(max |ref| {np.abs(ref).max():.3e})"


def assert_folded_close(qlayer, g, i, wbits):
    """The folded block (models/int_llama_layer.py:315-332, 365-368) vs the fixture's folded32.*: fake-quant weights,
    LET biases, norm parameters and the registered weight_quantizer.scales / zeros.  The block is float16 by now
    (qlayer.half(), quantize/omniquant.py:247)."""
    sd = dict(qlayer.named_buffers())
    seen = 0
    for k in [k for k in g if k.startswith(f"folded32.{i}.")]:
        n = k[len(f"folded32.{i}."):]
        ref = g[k].astype(np.float64)
        if n not in sd or sd[n] is None:
            assert np.abs(ref).max() == 0.0, f"folded32 {n} missing from the HIP block"
            continue
        got = sd[n].detach().double().cpu().numpy().reshape(ref.shape)
        seen += 1
        mx = max(np.abs(ref).max(), 1e-6)
        if n.endswith("weight_quantizer.zeros"):
            assert np.abs(got - ref).max() <= 1.0 and (got != ref).mean() <= 0.02, f"layer {i} {n}"
        elif n.endswith("proj.weight") or n.endswith("fc1.weight") or n.endswith("fc2.weight"):
            step = mx / (2 ** (wbits - 1))
            d = np.abs(got - ref)
            assert d.max() <= 2.5 * step and (d > 2e-3 * mx + fp16_ulp(ref)).mean() <= 0.01, f"layer {i} {n}: {d.max()}"
        else:
            assert_learned_close(got, ref, f"layer {i} {n}")
    assert seen >= 14


def oracle_block_from_step(g, meta):
    """CPU oracle block initialised from a g3_step fixture (weights, LET statistics, the fixture's parameter values)."""
    from oracle import ref_cpu as R
    weights = {k[2:]: T(v) for k, v in g.items() if k.startswith("w.")}
    spec = R.QuantSpec(meta["wbits"], meta["abits"], meta["group_size"], meta["lwc"], meta["let"])
    blk = R.Block(meta["family"], meta["config"], weights, spec, max_pos=max(int(meta.get("T", 16)), 16))
    if meta["let"]:
        sc = {k[len("act_scales."):]: T(v) for k, v in g.items() if k.startswith("act_scales.")}
        sh = {k[len("act_shifts."):]: T(v) for k, v in g.items() if k.startswith("act_shifts.")}
        blk.register_let(sc, sh, meta["alpha"], 0, meta["layer_prefix"])
    for n in list(blk.params.keys()):
        blk.params[n] = T(g["p0." + n]).reshape(blk.params[n].shape).requires_grad_(True)
    return blk


def oracle_step_bf16_model(g, meta):
    """One sample-step of the oracle on a fixture's inputs in the bf16 STORAGE model (fp32 arithmetic, every tensor the
    product path materialises rounded to bf16): returns (loss, {name: grad})."""
    blk = oracle_block_from_step(g, meta)
    bf = torch.bfloat16
    x, tgt, mask = T(g["x"]).to(bf).float(), T(g["target"]).to(bf).float(), T(g["mask"])
    pos = torch.from_numpy(g["position_ids"])
    from omniquant_amd import ops
    cols = [v.shape[1] for k, v in g.items() if k.startswith("w.") and k.endswith("proj.weight")]
    # the product's integer fprop (oq_gemm_i8) runs when its quantiser kernels can emit codes for these row lengths
    use_int = (ops.int_fprop_on() and meta["abits"] <= 8 and meta["wbits"] <= 8 and not meta["group_size"]
               and all(c % 128 == 0 and ops.int_codes_supported(c, c, meta["wbits"], False) and ops.int_codes_supported(c, c, meta["wbits"], True)
                       for c in cols))
    wide = ops.wide_on() and ops.grid_attention_on()                     # the product's un-rounded side channels
    temps = blk.temporaries(store_dtype=bf, int_fprop=use_int)
    out = blk.forward(x, mask, pos, temps=temps, act_quant=True, act_dtype=bf, int_fprop=use_int, wide=wide)
    loss = torch.nn.functional.mse_loss(tgt, out)
    loss.backward()
    return float(loss.detach()), {n: p.grad.detach().clone() for n, p in blk.params.items()}
