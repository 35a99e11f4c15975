import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniquant_amd.calibrate import calibrate_block, default_args
from omniquant_amd.synthetic import make_config, make_layer, make_calib_inputs, causal_mask, synth_act_stats
from omniquant_amd.llama_block import QuantLlamaDecoderLayer
DEV = "cuda:0"
cfg = make_config(None, family="llama", hidden_size=256, inter=512, heads=2, kv_heads=2)
Tn = 256
x = make_calib_inputs(4, Tn, 256, dtype=torch.bfloat16).to(DEV)
mask = causal_mask(Tn).to(DEV)
pos = torch.arange(Tn, device=DEV)[None]
sc, sh = synth_act_stats(cfg, 1)
for let in (False, True):
    for graph in (False, True):
        args = default_args(wbits=4, abits=4, lwc=True, let=let, epochs=2, nsamples=4, net="llama")
        outs = []
        for _ in range(3):
            q = QuantLlamaDecoderLayer(cfg, make_layer(cfg, seed=5, device=DEV), args).to(DEV)
            res = calibrate_block(q, args, "llama", 0, x.clone(), x.clone(), None, mask, pos, sc if let else None, sh if let else None, use_graph=graph)
            outs.append(res)
        same = all(all(torch.equal(outs[0]["omni"][k], o["omni"][k]) for k in outs[0]["omni"]) for o in outs[1:])
        print("let", let, "graph", graph, "omni identical:", same, "losses[1]:", [o["losses"][1] for o in outs], "norms[0]:", [o["norms"][0] for o in outs])
