"""OmniLayerNorm / OmniLlamaRMSNorm on the HIP path.  Surface = reference quantize/omni_norm.py:11-63."""
import torch
import torch.nn as nn

from . import ops


def _plain_act_quantizer(lin):
    """The QuantLinear's input quantiser if it is the plain dynamic per-token asymmetric one main.py builds and is
    switched on (then norm -> quant can be one kernel), else None."""
    q = getattr(lin, "act_quantizer", None)
    if (q is None or not lin.use_act_quant or lin.disable_input_quant or not q.enable or not (2 <= q.n_bits < 16)
            or q.symmetric or q.lwc or q.group_size or q.metric == "fix0to1" or q.dynamic_method != "per_token"
            or lin.__dict__.get("_stat_sink") is not None):
        return None
    return q


def _f32_cached(mod, name):
    """float32 copy of a frozen (fp16) norm weight / bias buffer, cached on the module: the LWC-only configurations read the
    raw norm parameters every step, and a per-step cast would be a launch of another library inside the captured step."""
    t = getattr(mod, name, None)
    if t is None or t.dtype == torch.float32:
        return t
    cache = mod.__dict__.setdefault("_f32_cache", {})
    key = (t.data_ptr(), t._version, t.dtype)
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        cache[name] = hit = (key, t.detach().float().contiguous())
    return hit[1]


class _FusedQuantMixin:
    def forward_quant(self, x, lin, is_ln, eps):
        """(fake_quant(norm(x)), x, sibling-gradient collector) through the fused kernels when `lin`'s input quantiser
        allows it, else None.  The collector (ops.SiblingGrads, may be None) is to be handed to every consumer of y."""
        q = _plain_act_quantizer(lin)
        if q is None or not ops.norm_quant_supported(x, q.n_bits):
            return None
        if self.use_temporary_parameter:
            weight, bias = self.temp_weight, self.temp_bias
        else:
            weight, bias = _f32_cached(self, "weight"), _f32_cached(self, "bias")
        # integer side channel for the int8 fprop of the linears that read y (ops.IntCodes), when they can use it
        stash = {"want_int": True} if (lin.weight_has_codes() and lin.int_fprop_eligible(x.dtype)) else {}
        wide = ops.wide_of(x) if x.dtype == torch.bfloat16 else None     # un-rounded hidden state (ops.wide_on)
        y, res = ops.NormQuantFn.apply(x, weight, bias, eps, is_ln, q.n_bits, stash, wide)
        if wide is not None:
            res._oq_wide = wide           # the residual path keeps the side channel (down_proj / fc2 add it in float32)
        q.scale, q.round_zero_point = stash["scale"], stash["zp"]
        if stash.get("int") is not None:
            y._oq_int = stash["int"]
        return y, res, stash.get("sib")


class OmniLayerNorm(_FusedQuantMixin, nn.Module):
    def __init__(self, ori_layer_norm) -> None:
        super().__init__()
        self.use_act_quant = True
        self.register_buffer("weight", ori_layer_norm.weight.detach())
        if ori_layer_norm.bias is not None:
            self.register_buffer("bias", ori_layer_norm.bias.detach())
        else:
            self.bias = None
        self.eps = ori_layer_norm.eps
        self.norm_func = ops.NormFn.apply
        self.normalized_shape = ori_layer_norm.normalized_shape
        self.use_temporary_parameter = False

    def forward(self, x):
        if self.use_temporary_parameter:
            weight, bias = self.temp_weight, self.temp_bias
        else:
            weight, bias = _f32_cached(self, "weight"), _f32_cached(self, "bias")
        return ops.NormFn.apply(x, weight, bias, self.eps, True)

    def forward_with_residual(self, x):
        """(norm(x), x) as one autograd node: see ops.NormResidualFn."""
        if self.use_temporary_parameter:
            weight, bias = self.temp_weight, self.temp_bias
        else:
            weight, bias = _f32_cached(self, "weight"), _f32_cached(self, "bias")
        return ops.NormResidualFn.apply(x, weight, bias, self.eps, True)

    def set_quant_state(self, use_weight_quant, use_act_quant):
        self.use_act_quant = use_act_quant


class OmniLlamaRMSNorm(_FusedQuantMixin, nn.Module):
    def __init__(self, ori_norm, eps=1e-6):
        super().__init__()
        self.register_buffer("weight", ori_norm.weight.detach())
        self.bias = None
        self.variance_epsilon = eps
        self.use_temporary_parameter = False

    def forward(self, hidden_states):
        if self.use_temporary_parameter:
            weight, bias = self.temp_weight, self.temp_bias
        else:
            weight, bias = _f32_cached(self, "weight"), _f32_cached(self, "bias")
        return ops.NormFn.apply(hidden_states, weight, bias, self.variance_epsilon, False)

    def forward_with_residual(self, hidden_states):
        """(norm(x), x) as one autograd node: see ops.NormResidualFn."""
        if self.use_temporary_parameter:
            weight, bias = self.temp_weight, self.temp_bias
        else:
            weight, bias = _f32_cached(self, "weight"), _f32_cached(self, "bias")
        return ops.NormResidualFn.apply(hidden_states, weight, bias, self.variance_epsilon, False)
