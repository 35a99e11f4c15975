"""QuantLinear on the HIP path.  Surface = reference quantize/int_linear.py:11-69."""
import torch
import torch.nn as nn

from . import ops
from .quantizer import UniformAffineQuantizer


def _hip_linear(input, weight, bias=None):
    if weight.dtype != input.dtype:
        weight = ops.cast(weight, input.dtype)
    return ops.LinearFn.apply(input, weight, bias)


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
        return self._wcache

    def drop_cache(self):
        self._wcache, self._wcache_key = None, None

    def quantize_input(self, input):
        """Per-token fake quant of the input, exposed so that sibling projections reading the SAME tensor with
        identical quantizer settings (q/k/v; gate/up) share one quantisation pass (reference quirk Q7: the three
        results are bit-identical)."""
        if self.use_act_quant and not self.disable_input_quant:
            return self.act_quantizer(input)
        return input

    def forward(self, input: torch.Tensor, input_is_quantized: bool = False):
        sink = self.__dict__.get("_stat_sink")
        if sink is not None:                        # LET-init statistics ride on the FP teacher pass (actstats.py)
            sink[0].update(sink[1], input)
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
        elif self.use_weight_quant:
            weight, bias = self.weight_quantizer.quantize(self.weight, out_dtype=input.dtype), self.bias
        else:
            weight, bias = self._weight_as(input.dtype), self.bias
        if not input_is_quantized:
            input = self.quantize_input(input)
        return self.fwd_func(input, weight, bias, **self.fwd_kwargs)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
