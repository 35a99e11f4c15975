"""QuantLinear on the HIP path.  Surface = reference quantize/int_linear.py:11-69."""
import os

import torch
import torch.nn as nn

from . import ops
from .quantizer import UniformAffineQuantizer


def _hip_linear(input, weight, bias=None, residual=None, sib=None, wide=False):
    """wide: also keep the un-rounded float32 result as `_oq_wide` on the (bf16) output -- integer fprop only (ops.wide_on)."""
    xint, wint = getattr(input, "_oq_int", None), getattr(weight, "_oq_int", None)
    if weight.dtype != input.dtype:
        weight = ops.cast(weight, input.dtype)
    if xint is None or wint is None or not ops.int_fprop_on():
        xint = wint = None
    stash = None
    if wide and xint is not None and input.dtype == torch.bfloat16 and ops.wide_on():
        stash = {"want_wide": True, "res_wide": ops.wide_of(residual) if residual is not None else None}
    # both operands carry their integer codes (ops.IntCodes): the fprop runs on the int8 MFMA, exactly
    y = ops.LinearFn.apply(input, weight, bias, residual, sib, xint, wint, stash)
    if stash is not None and stash.get("wide") is not None:
        y._oq_wide = stash["wide"].view(y.shape)
    return y


class QuantLinear(nn.Module):
    """Fake-quantised Linear.  `fwd_func` is the MFMA GEMM (bf16 tiles, or exact f32 in parity mode)."""

    def __init__(self, org_module: nn.Linear, weight_quant_params: dict = {}, act_quant_params: dict = {},
                 disable_input_quant=False):
        super().__init__()
        self.fwd_kwargs = dict()
        self.fwd_func = _hip_linear
        # .detach(): share the storage (as the reference does) but never keep an nn.Parameter inside a buffer --
        # a Parameter assigned to temp_weight/temp_bias would register itself as a learnable (quirk Q9)
        self.register_buffer("weight", org_module.weight.detach())
        if org_module.bias is not None:
            self.register_buffer("bias", org_module.bias.detach())
        else:
            self.bias = None
        self.in_features = org_module.in_features
        self.out_features = org_module.out_features
        self.use_weight_quant = False
        self.use_act_quant = False
        self.weight_quantizer = UniformAffineQuantizer(**weight_quant_params, shape=org_module.weight.shape)
        if not disable_input_quant:
            self.act_quantizer = UniformAffineQuantizer(**act_quant_params)
        else:
            self.act_quantizer = None
        self.disable_input_quant = disable_input_quant
        self.use_temporary_parameter = False
        self._wcache, self._wcache_key = None, None

    def _weight_as(self, dtype):
        """raw / folded weight in the activation dtype; the cast is cached (teacher and propagate passes call
        the same frozen weight for every sample)."""
        w = self.weight
        if w.dtype == dtype:
            return w
        key = (w.data_ptr(), w._version, dtype)
        if self._wcache_key != key:
            self._wcache, self._wcache_key = ops.cast(w, dtype), key
            ic = getattr(w, "_oq_int", None)          # a folded weight keeps the integer codes of its fold (block_common)
            if ic is not None:
                self._wcache._oq_int = ic
        return self._wcache

    def drop_cache(self):
        self._wcache, self._wcache_key = None, None

    def int_fprop_eligible(self, dtype):
        """True when this linear's fprop can run on integer codes (ops.gemm_i8) for activations of `dtype`: production mode
        (bf16), the plain dynamic per-token input quantiser and a per-channel (ungrouped) weight quantiser, both on grids
        of at most 8 bits, and the HIP GEMM as forward function.  OQ_INT_FPROP=0 switches the path off."""
        aq, wq = self.act_quantizer, self.weight_quantizer
        return (ops.int_fprop_on() and dtype == torch.bfloat16 and self.fwd_func is _hip_linear and not self.fwd_kwargs
                and self.use_act_quant and aq is not None and not self.disable_input_quant and aq.enable and 2 <= aq.n_bits <= 8
                and not aq.group_size and aq.metric != "fix0to1" and aq.dynamic_method == "per_token"
                and wq.enable and 2 <= wq.n_bits <= 8 and not wq.group_size and wq.metric != "fix0to1"
                and self.in_features % 128 == 0 and self.out_features % 8 == 0      # the int8 MFMA kernel's K-tile / store width
                # (the weight may or may not be LET-transformed: its quantiser must be able to emit codes either way)
                and ops.int_codes_supported(self.in_features, self.in_features, wq.n_bits, False)
                and ops.int_codes_supported(self.in_features, self.in_features, wq.n_bits, True))

    def weight_has_codes(self):
        """The weight this forward will use carries integer codes: the step's temporaries do when the linear is eligible
        (block_common asks their quantisers for them), a folded weight does when its fold produced them."""
        return self.use_temporary_parameter or getattr(self.weight, "_oq_int", None) is not None

    def quantize_input(self, input):
        """Per-token fake quant of the input, exposed so that sibling projections reading the SAME tensor with
        identical quantizer settings (q/k/v; gate/up) share one quantisation pass (reference quirk Q7: the three
        results are bit-identical)."""
        if self.use_act_quant and not self.disable_input_quant:
            aq = self.act_quantizer
            if input.is_cuda and aq.dynamic_method == "per_token" and aq.metric != "fix0to1" and not aq._identity():
                # + integer codes for the int8 fprop (same values) when this linear takes it; read from the input's un-rounded
                # side channel if it has one (ops.wide_of)
                want = self.int_fprop_eligible(input.dtype) and self.weight_has_codes()
                src = ops.wide_of(input) if input.dtype == torch.bfloat16 else None
                if want or src is not None:
                    return aq.quantize(input, want_int=want, src=src)
            return aq(input)
        return input

    def _resolve(self, dtype):
        """(weight, bias) this forward uses: the step's temporaries (produced lazily if deferred), the on-the-fly
        fake-quant weight, or the raw / folded weight cast to the activation dtype."""
        if self.use_temporary_parameter:
            lazy = self.__dict__.pop("_lazy_temp", None)
            if lazy is not None:                    # deferred fake-quant of this weight (block_common._let_temporaries)
                lazy()
            weight, bias = self.temp_weight, self.temp_bias
            ev = self.__dict__.get("_temp_ready")
            if ev is not None:                      # temp weight was produced on the block's weight stream
                cur = torch.cuda.current_stream()
                cur.wait_event(ev)
                weight.record_stream(cur)
                if bias is not None:
                    bias.record_stream(cur)
            return weight, bias
        if self.use_weight_quant:
            return self.weight_quantizer.quantize(self.weight, out_dtype=dtype), self.bias
        return self._weight_as(dtype), self.bias

    def forward(self, input: torch.Tensor, input_is_quantized: bool = False, residual=None, sib=None, wide=False):
        """residual (HIP-path extension): added to the output inside the GEMM's store (the block's residual add).
        sib: ops.SiblingGrads collector of the tensor `input` came from (fused norm -> quant), or None.
        wide: ask the HIP GEMM for the un-rounded float32 copy of the output (`_oq_wide`, see _hip_linear)."""
        sink = self.__dict__.get("_stat_sink")
        if sink is not None:                        # LET-init statistics ride on the FP teacher pass (actstats.py)
            sink[0].update(sink[1], input)
        weight, bias = self._resolve(input.dtype)
        if not input_is_quantized:
            input = self.quantize_input(input)
        if (sib is not None or wide) and self.fwd_func is _hip_linear:
            return self.fwd_func(input, weight, bias, residual=residual, sib=sib, wide=wide, **self.fwd_kwargs)
        if residual is not None:
            return self.fwd_func(input, weight, bias, residual=residual, **self.fwd_kwargs)
        return self.fwd_func(input, weight, bias, **self.fwd_kwargs)

    @staticmethod
    def forward_siblings(mods, input, sib=None):
        """Projections reading the SAME (already act-quantised) input: one autograd node whose backward accumulates
        the input gradient in the dgrad GEMMs' epilogue (ops.SiblingLinearFn).  Falls back to independent calls when a
        module does not use the HIP GEMM as its fwd_func."""
        # opt-in (OQ_SIBLING=1): measured 1 % SLOWER than independent projections + autograd's add launches on the 7B
        # step (same-box A/B 245.9 vs 248.6 sample-steps/s): the epilogue's read-modify-write of the 16.8 MB gradient
        # costs more inside the MFMA kernel than in a streaming add
        if os.environ.get("OQ_SIBLING", "0") == "0" or any(m.fwd_func is not _hip_linear or m.fwd_kwargs for m in mods):
            return tuple(m(input, True, sib=sib) for m in mods)
        wb = []
        for m in mods:
            sink = m.__dict__.get("_stat_sink")
            if sink is not None:
                sink[0].update(sink[1], input)
            w, b = m._resolve(input.dtype)
            if w.dtype != input.dtype:
                w = ops.cast(w, input.dtype)
            wb += [w, b]
        return ops.SiblingLinearFn.apply(input, *wb)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
