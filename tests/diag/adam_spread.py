"""Diagnostic (CPU, not a test): how far do the learned tensors of two ORACLE calibrations drift apart after the 6 steps of
tests/test_hip_block.py::test_stacked_sibling_gemms_leave_the_calibration_unchanged when the inputs differ by one bf16 ulp?
AdamW's first updates are sign-like (|update| ~ lr whatever the gradient's size), so elements whose gradient is of noise
magnitude move by up to 2 * lr per step in opposite directions.  This is the oracle-side number behind that test's bound."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R          # noqa: E402
from omniquant_amd import synthetic as S  # noqa: E402

H, I, T, NS, EPOCHS = 1024, 1536, 256, 3, 2
cfg = S.make_config(None, family="llama", hidden_size=H, inter=I, heads=8, kv_heads=8)
layer = S.make_layer(cfg, seed=11, device="cpu")
weights = {n: p.detach().float() for n, p in layer.named_parameters()}
cd = dict(hidden_size=H, intermediate_size=I, num_attention_heads=8, num_key_value_heads=8, rms_norm_eps=1e-6)
x = S.make_calib_inputs(NS, T, H, dtype=torch.bfloat16).float()
mask, pos = S.causal_mask(T), torch.arange(T)[None]
sc, sh = S.synth_act_stats(cfg, 1)
spec = R.QuantSpec(4, 4, None, True, True)
runs = []
for seed in (None, 1, 2, 3):
    xin = x
    if seed is not None:
        g = torch.Generator().manual_seed(seed)
        xin = x * (1 + (2.0 ** -9) * (torch.rand(x.shape, generator=g) - 0.5))        # within one bf16 ulp
    runs.append(R.calibrate("llama", cd, [weights], spec, xin, mask, pos, sc, sh, epochs=EPOCHS)["omni"][0])
for k in runs[0]:
    d = [float((runs[0][k].float() - r[k].float()).abs().mean()) for r in runs[1:]]
    print(f"{k:55s} mean|a| {float(runs[0][k].float().abs().mean()):.4f}  mean|delta| under 1-ulp input noise: " + " ".join(f"{v:.2e}" for v in d))
