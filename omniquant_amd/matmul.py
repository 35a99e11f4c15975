"""QuantMatMul on the HIP path.  Surface = reference quantize/int_matmul.py:7-43.

`matmul_func` keeps the reference meaning (torch.bmm for OPT, torch.matmul for LLaMA) but both resolve to the
strided-batched MFMA GEMM; the block code calls `scores()` / `apply_probs()` which take the head-interleaved
[bs, T, heads, head_dim] layout straight from the projection GEMMs (no transposes, no copies)."""
import torch
import torch.nn as nn

from . import ops
from .quantizer import UniformAffineQuantizer


class QuantMatMul(nn.Module):
    def __init__(self, x1_quant_params: dict = {}, x2_quant_params: dict = {}, disable_act_quant=False,
                 matmul_func=torch.bmm):
        super().__init__()
        self.use_act_quant = False
        self.i_cluster_counts = None
        self.x1_quantizer = UniformAffineQuantizer(**x1_quant_params)
        self.x2_quantizer = UniformAffineQuantizer(**x2_quant_params)
        self.matmul_func = matmul_func
        self.disable_act_quant = disable_act_quant

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant

    def quant_x1(self, x1):
        if self.use_act_quant:
            x1 = self.x1_quantizer(x1)
        return x1

    def quant_x2(self, x2):
        if self.use_act_quant:
            x2 = self.x2_quantizer(x2)
        return x2

    # ---- HIP GEMM entry points on the [bs, T, heads, hd] layout ------------------------------------------
    def scores(self, q, k, causal=False):
        """q [bs,T,nh,hd], k [bs,Tk,nkv,hd] -> q @ k^T  [bs,nh,T,Tk].  causal=True: only tiles on/below the
        diagonal are computed (the rest is never read by the causal softmax)."""
        return ops.AttnScoresFn.apply(q, k, causal)

    def apply_probs(self, p, v, causal=False):
        """p [bs,nh,T,Tk], v [bs,Tk,nkv,hd] -> p @ v  [bs,T,nh,hd]"""
        return ops.AttnPVFn.apply(p, v, causal)

    def forward(self, x1, x2):
        """Generic x1 @ x2 with x1 [..., M, K], x2 [..., K, N] (reference call shape)."""
        lead = x1.shape[:-2]
        M, K = x1.shape[-2:]
        N = x2.shape[-1]
        a = x1.reshape(-1, 1, M, K).transpose(1, 2)            # [B, M, 1, K]  as q-layout with 1 head
        b = x2.transpose(-1, -2).reshape(-1, 1, N, K).transpose(1, 2)   # [B, N, 1, K]
        out = ops.AttnScoresFn.apply(a.contiguous(), b.contiguous())    # [B, 1, M, N]
        return out.view(*lead, M, N)
