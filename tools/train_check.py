"""Functional check at full size: the calibration loss of a LLaMA-7B-shaped block must go DOWN over epochs (bf16
production mode, hipGraph step)."""
import sys, os, time, torch, logging
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd.calibrate import default_args, calibrate_layers
from omniquant_amd import synthetic as S
logging.basicConfig(level=logging.INFO, format="%(message)s")
log = logging.getLogger("check")
dev = "cuda:0"
name = sys.argv[1] if len(sys.argv) > 1 else "llama-7b"
cfg = S.make_config(name)
let = len(sys.argv) < 3 or sys.argv[2] != "lwc"
args = default_args(wbits=4, abits=4 if let else 16, lwc=True, let=let, epochs=4, nsamples=16, net=name, aug_loss=False)
layers = [S.make_layer(cfg, seed=i, device=dev) for i in range(2)]
inps = S.make_calib_inputs(16, 2048, cfg.hidden_size, device=dev, dtype=torch.bfloat16)
mask = S.causal_mask(2048, dev)
pos = torch.arange(2048, device=dev)[None]
sc, sh = S.synth_act_stats(cfg, 2)
t0 = time.time()
q, omni, losses, _ = calibrate_layers(layers, cfg, args, inps, mask, pos, sc, sh, logger=log)
torch.cuda.synchronize()
print("wall", time.time() - t0, "s for", len(losses), "sample-steps + teacher/propagate")
import numpy as np
L = np.asarray(losses).reshape(2, 4, 16).mean(-1)
print("epoch-mean losses per layer:\n", L)
assert (L[:, -1] < L[:, 0]).all(), "loss did not decrease"
print("omni keys:", len(omni[0]), list(omni[0].keys())[:4])
