"""UniformAffineQuantizer on the HIP path.  Surface = reference quantize/quantizer.py:22-152."""
import math

import torch
import torch.nn as nn

from . import ops

CLIPMIN = 1e-5


def round_ste(x: torch.Tensor):
    """Straight-through rounding (quantize/quantizer.py:15-19).  Kept for API parity; the HIP fake-quant kernel
    applies the same estimator internally."""
    return (x.round() - x).detach() + x


class UniformAffineQuantizer(nn.Module):
    def __init__(self, n_bits: int = 8, symmetric: bool = False, per_channel_axes=[], metric="minmax",
                 dynamic=False, dynamic_method="per_cluster", group_size=None, shape=None, lwc=False):
        super().__init__()
        self.symmetric = symmetric
        assert 2 <= n_bits <= 16, "bitwidth not supported"
        self.n_bits = n_bits
        self.qmin = 0
        self.qmax = 2 ** n_bits - 1
        self.per_channel_axes = per_channel_axes
        self.metric = metric
        self.cluster_counts = None
        self.cluster_dim = None
        self.scale = None
        self.zero_point = None
        self.round_zero_point = None
        self.cached_xmin = None
        self.cached_xmax = None
        self.dynamic = dynamic
        self.dynamic_method = dynamic_method
        self.deficiency = 0
        self.lwc = lwc
        init_value = 4.0   # sigmoid(4) = 0.982 (quirk Q3)
        if lwc:
            if group_size:
                dim1 = int(shape[0] * math.ceil(shape[1] / group_size))
                self.deficiency = shape[-1] % group_size
                if self.deficiency > 0:
                    self.deficiency = group_size - self.deficiency
                    assert self.symmetric
            else:
                dim1 = shape[0]
            self.upbound_factor = nn.Parameter(torch.ones((dim1, 1)) * init_value)
            self.lowbound_factor = nn.Parameter(torch.ones((dim1, 1)) * init_value)
        self.sigmoid = nn.Sigmoid()
        self.enable = True
        self.group_size = group_size

    def change_n_bits(self, n_bits):
        self.n_bits = n_bits
        self.qmin = 0
        self.qmax = 2 ** n_bits - 1

    # ---- the fused HIP path ---------------------------------------------------------------------------
    def _segment(self, x):
        if self.group_size:
            assert len(x.shape) == 2, "only support linear layer now"
            if self.deficiency > 0:
                raise NotImplementedError("ragged weight groups (in_features % group_size != 0) are not "
                                          "implemented on the HIP path")
            return self.group_size
        return x.shape[-1]

    def quantize(self, x, out_dtype=None, col_mul=None, row_div=None, row_mul=None, shift=None):
        """Dynamic calibration + fake quant in ONE kernel, optionally fused with the LET weight transform
        x' = ((x*col_mul)/row_div)*row_mul and the by-product x @ shift.  Sets self.scale/round_zero_point."""
        stash = {}
        up = self.upbound_factor if self.lwc else None
        low = self.lowbound_factor if self.lwc else None
        res = ops.fake_quant(x, self.n_bits, self._segment(x), up, low, self.symmetric, out_dtype, stash,
                             col_mul, row_div, row_mul, shift)
        self.scale, self.round_zero_point = stash["scale"], stash["zp"]
        return res

    def forward(self, x: torch.Tensor):
        if self.n_bits >= 16 or not self.enable:
            return x
        if self.metric == "fix0to1":
            return x.mul_(2 ** self.n_bits - 1).round_().div_(2 ** self.n_bits - 1)
        if self.dynamic_method == "per_token" or self.dynamic_method == "per_channel":
            return self.quantize(x)
        raise NotImplementedError()

    # kept for API parity: in the reference these are two passes; here both are served by the fused kernel
    def per_token_dynamic_calibration(self, x):
        self.quantize(x.detach())

    def fake_quant(self, x, scale=None, round_zero_point=None):
        return self.quantize(x)

    def register_scales_and_zeros(self):
        self.register_buffer("scales", self.scale)
        self.register_buffer("zeros", self.round_zero_point)
        del self.scale
        del self.round_zero_point
