import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from gpu_helpers import T, make_args, make_cfg
from omniquant_amd.calibrate import *
from omniquant_amd.synthetic import make_layer
from omniquant_amd.optim import BlockOptimizer
DEV="cuda:0"
fname = sys.argv[1] if len(sys.argv)>1 else "g4_traj_llama_w4a4_lwc_let.npz"
dtype = torch.bfloat16 if (len(sys.argv)>2 and sys.argv[2]=="bf16") else torch.float32
g, m = load_golden(fname)
cfg, args = make_cfg(m), make_args(m)
sc = {k[len("act_scales."):]: T(v) for k, v in g.items() if k.startswith("act_scales.")}
sh = {k[len("act_shifts."):]: T(v) for k, v in g.items() if k.startswith("act_shifts.")}
pos = torch.from_numpy(g["position_ids"]).to(DEV)
mask = T(g["mask"], DEV)
inps = T(g["inps"], DEV, dtype)
def run(use_graph):
    w = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w0.")}
    layer = make_layer(cfg, weights=w, device=DEV)
    q = decoder_layer_class(m["family"])(cfg, layer, args).to(DEV)
    q.compute_dtype = dtype
    q.set_quant_state(False, False)
    with torch.no_grad():
        fp = q(inps, attention_mask=mask.expand(inps.shape[0],-1,-1,-1), position_ids=pos)[0]
    q.set_quant_state(False, True); q.let = m["let"]
    if m["let"]: register_let_parameters(q, m["family"], sc, sh, m["alpha"], 0, DEV)
    opt = BlockOptimizer(q, args.let_lr, args.lwc_lr, 0.0)
    r = StepRunner(q, opt, mask, pos, (1,)+tuple(inps.shape[1:]), dtype, False, True, use_graph)
    hist=[]
    for j in range(4):
        r.run(inps[j:j+1], fp[j:j+1])
        torch.cuda.synchronize()
        hist.append((float(r.loss), float(opt.norm[0]), opt.flat.clone(), opt.grad.clone()))
    return hist, opt
h0, o0 = run(False)
h1, o1 = run(True)
names=[]; off=0
for n,p in o0.named:
    names.append((n,off,off+p.numel())); off+=p.numel()
for j in range(4):
    print("step",j,"loss",h0[j][0],h1[j][0],"norm",h0[j][1],h1[j][1])
    for n,a,b in names:
        dg=(h0[j][3][a:b]-h1[j][3][a:b]).abs().max().item(); dp=(h0[j][2][a:b]-h1[j][2][a:b]).abs().max().item()
        if dg>1e-6*max(1e-9,h0[j][3][a:b].abs().max().item()) or dp>1e-6:
            print("   ",n,"dgrad",dg,"gmax",h0[j][3][a:b].abs().max().item(),"dparam",dp)
print("---- huge grads in graph mode")
for j in range(4):
    gr = h1[j][3]
    for n,a,b in names:
        mx = gr[a:b].abs().max().item()
        if mx > 1e3 or mx != mx:
            idx = int(gr[a:b].abs().argmax())
            print(j, n, "max", mx, "at", idx, "of", b-a, "eager val", h0[j][3][a+idx].item())
