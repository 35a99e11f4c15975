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
                        nsamples=meta.get("nsamples", 1))


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


def run_block_step_parity(fname, dtype=torch.float32, device="cuda:0"):
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
    loss.backward()
    errs["grad"] = 0.0
    errs["per_grad"] = {}
    for n, p in q.named_parameters():
        e = rel_err(p.grad, g["grad." + n])
        errs["per_grad"][n] = e
        errs["grad"] = max(errs["grad"], e)
    return errs
