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
            return self.group_size
        return x.shape[-1]

    def _identity(self):
        return self.n_bits >= 16 or not self.enable

    def quantize(self, x, out_dtype=None, col_mul=None, row_div=None, row_mul=None, shift=None, out=None, want_int=False,
                 src=None):
        """Dynamic calibration + fake quant in ONE kernel, optionally fused with the LET weight transform
        x' = ((x*col_mul)/row_div)*row_mul and the by-product x @ shift.  Sets self.scale/round_zero_point.

        n_bits >= 16 or a disabled quantizer is the identity of the reference's forward() gate
        (quantize/quantizer.py:109-110): the tensor comes back unchanged (cast to out_dtype), scale / zero-point stay
        None; with LET arguments the transform still runs (kernel's identity grid).
        A ragged last group (in_features % group_size != 0, symmetric only as in the reference, :64-69) is zero-padded
        inside the kernel exactly like :85-87 / :125-128.
        out: destination tensor of the fake-quantised values (block_common stacks sibling weights in one buffer); ignored
        by the identity branch.
        want_int: also produce the integer side channel (ops.IntCodes: the grid codes as int8 + per-row code sums) when the
        kernels can; it is attached to the returned tensor as `_oq_int` and kept in `self.int_codes` (None when not produced).
        src: float32 tensor with x's values before their rounding to 16 bits (ops.wide_of): the kernel reads it in place of x."""
        let = not (col_mul is None and row_div is None and row_mul is None and shift is None)
        self.int_codes = None
        if self._identity():
            self.scale = self.round_zero_point = None
            if not let:
                return ops.cast(x, out_dtype) if out_dtype is not None else x
            res = ops.fake_quant(x, 16, x.shape[-1], None, None, False, out_dtype, None, col_mul, row_div, row_mul, shift)
            return res
        stash = {"want_int": True} if (want_int and self.n_bits <= 8 and x.dim() >= 2) else {}
        up = self.upbound_factor if self.lwc else None
        low = self.lowbound_factor if self.lwc else None
        seg = self._segment(x)
        if self.group_size and x.shape[-1] % seg != 0:
            assert self.symmetric, "ragged weight groups are only defined for the symmetric grid (quantizer.py:69)"
        res = ops.fake_quant(x, self.n_bits, seg, up, low, self.symmetric, out_dtype, stash,
                             col_mul, row_div, row_mul, shift, out=out, src=src)
        self.scale, self.round_zero_point = stash["scale"], stash["zp"]
        (res[0] if isinstance(res, tuple) else res)._oq_fq_out = True      # its backward knows ops.WgradQueue's protocol
        ic = stash.get("int")
        if ic is not None:
            self.int_codes = ic
            y = res[0] if isinstance(res, tuple) else res
            y._oq_int = ic             # consumers (linear._hip_linear, the fused block nodes) pick it up from the tensor
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
        """API parity with quantize/quantizer.py:84-105.  Called with this quantizer's own freshly calibrated
        scale / zero-point (the only way the reference calls it) it is the fused kernel; explicit foreign parameters are
        applied with the reference's formula on the device (not on the calibration path)."""
        if scale is None or (scale is self.scale and round_zero_point is self.round_zero_point):
            return self.quantize(x)
        shape = x.shape
        xg = x
        if self.group_size:
            assert len(shape) == 2, "only support linear layer now"
            if self.deficiency > 0:
                xg = torch.cat((x, x.new_zeros(shape[0], self.deficiency)), dim=1)
            xg = xg.reshape(-1, self.group_size)
        x_int = round_ste(xg / scale)
        if round_zero_point is not None:
            x_int = x_int.add(round_zero_point)
        x_int = x_int.clamp(self.qmin, self.qmax)
        if round_zero_point is not None:
            x_int = x_int.sub(round_zero_point)
        y = x_int.mul(scale)
        if self.group_size:
            y = y.reshape(shape[0], -1)
            if self.deficiency > 0:
                y = y[:, :-self.deficiency]
        return y

    def register_scales_and_zeros(self):
        self.register_buffer("scales", self.scale)
        self.register_buffer("zeros", self.round_zero_point)
        del self.scale
        del self.round_zero_point
