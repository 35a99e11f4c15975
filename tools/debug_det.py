"""Run-to-run determinism probe of a block forward (tools only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniquant_amd.calibrate import default_args, register_let_parameters
from omniquant_amd.synthetic import make_config, make_layer, make_calib_inputs, causal_mask, synth_act_stats
from omniquant_amd.llama_block import QuantLlamaDecoderLayer
from omniquant_amd import ops
DEV = "cuda:0"
cfg = make_config(None, family="llama", hidden_size=256, inter=512, heads=2, kv_heads=2)
args = default_args(wbits=4, abits=4, lwc=True, let=True, epochs=1, nsamples=2, net="llama")
Tn = 256
x = make_calib_inputs(2, Tn, 256, dtype=torch.bfloat16).to(DEV)
mask = causal_mask(Tn).to(DEV)
pos = torch.arange(Tn, device=DEV)[None]
sc, sh = synth_act_stats(cfg, 1)
q = QuantLlamaDecoderLayer(cfg, make_layer(cfg, seed=7, device=DEV), args).to(DEV)
q.compute_dtype = torch.bfloat16
q.set_quant_state(weight_quant=False, act_quant=True)
q.let = True
register_let_parameters(q, "llama", sc, sh, 0.5, 0, DEV)
with torch.no_grad():
    for p_ in q.parameters():
        p_.data = p_.data.float()
rec = {}
def hook(name):
    def f(m, i, o):
        t = o[0] if isinstance(o, tuple) else o
        rec.setdefault(name, []).append(t.detach().clone())
    return f
for n, m in q.named_modules():
    if n:
        m.register_forward_hook(hook(n))
orig = ops.FusedCausalAttnFn.apply
def attn(*a):
    o = orig(*a)
    rec.setdefault("ATTN", []).append(o.detach().clone())
    for i, t in enumerate(a[:3]):
        rec.setdefault(f"ATTN_in{i}", []).append(t.detach().clone())
    return o
ops.FusedCausalAttnFn.apply = attn
for bs in (1,):
    rec.clear()
    for it in range(8):
        for p_ in q.parameters():
            p_.grad = None
        q.smooth_and_quant_temporary()
        for n, m in q.named_modules():
            if hasattr(m, "temp_weight") and getattr(m, "use_temporary_parameter", False) and m.temp_weight is not None:
                rec.setdefault("TW." + n, []).append(m.temp_weight.detach().clone())
        out = q(x[:bs], attention_mask=mask.expand(bs, -1, -1, -1), position_ids=pos)[0]
        (out.float() ** 2).mean().backward()
        for n, p_ in q.named_parameters():
            if p_.grad is not None:
                rec.setdefault("GRAD." + n, []).append(p_.grad.detach().clone())
        q.clear_temp_variable()
    torch.cuda.synchronize()
    print("bs", bs)
    for n, lst in rec.items():
        bad = [i for i in range(1, len(lst)) if int((lst[i].float() != lst[0].float()).sum()) != 0]
        if bad:
            print("   NONDET", n, "runs", bad, "count", [int((lst[i].float() != lst[0].float()).sum()) for i in bad][:4])
print("done")
