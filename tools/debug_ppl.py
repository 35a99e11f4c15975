import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import numpy as np, torch
import test_ppl_parity as TP
from oracle import ref_cpu as R
from omniquant_amd.calibrate import calibrate_layers, default_args
from omniquant_amd.synthetic import make_config, make_layer
DEV = "cuda:0"
emb, head, fnorm, layers, tokens = TP._model()
inps = emb[tokens]; T = TP.T
mask = torch.triu(torch.full((T, T), torch.finfo(torch.float32).min), 1)[None, None]; pos = torch.arange(T)[None]
g = torch.Generator().manual_seed(5)
names = ["self_attn.q_proj", "self_attn.o_proj", "mlp.up_proj"]
sc = {f"model.layers.{i}.{n}": torch.rand(128, generator=g) * 3 + 0.2 for i in range(TP.NLAYERS) for n in names}
sh = {k: torch.zeros(128) for k in sc}
for (ab, let) in ((4, True), (4, False), (16, True)):
    spec = R.QuantSpec(4, ab, None, True, let)
    ref = R.calibrate("llama", TP.CFG, layers, spec, inps, mask, pos, sc, sh, epochs=3)
    cfg = make_config(None, family="llama", hidden_size=128, inter=256, heads=4, kv_heads=4)
    args = default_args(wbits=4, abits=ab, lwc=True, let=let, epochs=3, nsamples=TP.NSAMP, net="llama")
    hl = [make_layer(cfg, weights=dict(w), device=DEV) for w in layers]
    _, omni, losses, (qi, fi) = calibrate_layers(hl, cfg, args, inps.to(DEV), mask.to(DEV), pos.to(DEV), sc, sh, use_graph=False, compute_dtype=torch.float32)
    l0, l1 = np.asarray(ref["losses"]), np.asarray(losses)
    print(f"abits={ab} let={let}: ppl oracle {TP._ppl(ref['quant_out'][-1], fnorm, head, tokens):.4f} hip {TP._ppl(qi.float().cpu(), fnorm, head, tokens):.4f}")
    rel = np.abs(l1 - l0) / l0
    print("   loss rel diff per step (first 24 = layer0):", np.round(rel[:24], 5).tolist())
    for i in range(TP.NLAYERS):
        worst = max((float((omni[i][n].double() - t.double()).abs().max() / max(float(t.double().abs().max()), 1e-6)), n) for n, t in ref["omni"][i].items())
        print("   layer", i, "worst learned-tensor rel diff", worst)
