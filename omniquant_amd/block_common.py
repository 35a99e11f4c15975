"""Shared machinery of the quantised decoder blocks (LLaMA / OPT): quant-state switches, the fused
smooth-and-quant step, the final fold, parameter selectors and the omni state dict.

Reference: models/int_llama_layer.py:269-368 and models/int_opt_layer.py:348-452 (the two copies are
identical up to module names).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import let as L
from .linear import QuantLinear
from .matmul import QuantMatMul


class QuantBlockMixin:
    #: activation / temp-weight dtype of the HIP path: torch.bfloat16 (MFMA bf16) or torch.float32 (exact)
    compute_dtype = torch.bfloat16
    let = False

    # subclasses define: _let_names() -> dict(q,k,v,o,fc1=[...],last, ln1, ln2)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        for m in self.modules():
            if isinstance(m, (QuantLinear, QuantMatMul)):
                m.set_quant_state(weight_quant, act_quant)

    def _quant_linears(self):
        return [m for m in self.modules() if isinstance(m, QuantLinear)]

    def _truncate_scales(self):
        with torch.no_grad():
            for name, p in self.named_parameters():
                if "smooth_scale" in name:
                    L.truncate_number(p)

    def smooth_and_quant_temporary(self):
        """LET re-parameterisation + weight fake-quant of every linear: ONE fused kernel per weight matrix
        (reference: ~25 eager passes per matrix, models/int_llama_layer.py:279-307)."""
        nm = self._let_names()
        dt = self.compute_dtype
        if self.let:
            self._truncate_scales()
            specs = L.block_let_specs(nm, self, None)
            for ln, key in ((nm["ln1"], "qkv"), (nm["ln2"], "fc1")):
                ln.temp_weight, ln.temp_bias = L.ln_temporaries(
                    ln, getattr(self, f"{key}_smooth_scale"), getattr(self, f"{key}_smooth_shift"))
                ln.use_temporary_parameter = True
            for mod, sp in specs.items():
                if sp.shift is not None:
                    mod.temp_weight, ws = mod.weight_quantizer.quantize(
                        mod.weight, out_dtype=dt, col_mul=sp.col_mul, row_div=sp.row_div, row_mul=sp.row_mul,
                        shift=sp.shift)
                    mod.temp_bias = sp.bias_fn(ws)
                else:
                    mod.temp_weight = mod.weight_quantizer.quantize(mod.weight, out_dtype=dt)
                    mod.temp_bias = mod.bias
                mod.use_temporary_parameter = True
        else:
            for mod in self._quant_linears():
                mod.temp_weight = mod.weight_quantizer.quantize(mod.weight, out_dtype=dt)
                mod.temp_bias = mod.bias
                mod.use_temporary_parameter = True

    def clear_temp_variable(self):
        nm = self._let_names()
        for mod in self._quant_linears() + [nm["ln1"], nm["ln2"]]:
            if hasattr(mod, "temp_weight"):
                del mod.temp_weight
            if hasattr(mod, "temp_bias"):
                del mod.temp_bias

    @torch.no_grad()
    def smooth_and_quant_inplace(self):
        """Final fold (models/int_llama_layer.py:315-332): weights <- fake_quant(LET(W)) in float32, LET biases
        and norm parameters become buffers."""
        nm = self._let_names()
        if self.let:
            self._truncate_scales()
            specs = L.block_let_specs(nm, self, None)
            for ln, key in ((nm["ln1"], "qkv"), (nm["ln2"], "fc1")):
                tw, tb = L.ln_temporaries(ln, getattr(self, f"{key}_smooth_scale"), getattr(self, f"{key}_smooth_shift"))
                ln.use_temporary_parameter = False
                if hasattr(ln, "bias"):
                    del ln.bias
                ln.register_buffer("bias", tb)
                ln.weight = tw
            for mod, sp in specs.items():
                if sp.shift is not None:
                    w, ws = mod.weight_quantizer.quantize(mod.weight, out_dtype=torch.float32, col_mul=sp.col_mul,
                                                          row_div=sp.row_div, row_mul=sp.row_mul, shift=sp.shift)
                    b = sp.bias_fn(ws)
                    if hasattr(mod, "bias"):
                        del mod.bias
                    mod.register_buffer("bias", b)
                else:
                    w = mod.weight_quantizer.quantize(mod.weight, out_dtype=torch.float32)
                mod.weight = w
                mod.use_temporary_parameter = False
        else:
            for mod in self._quant_linears():
                mod.weight = mod.weight_quantizer.quantize(mod.weight, out_dtype=torch.float32)
                mod.use_temporary_parameter = False
        for ln in (nm["ln1"], nm["ln2"]):
            ln.use_temporary_parameter = False

    def let_parameters(self, use_shift=True):
        template = "smooth" if use_shift else "smooth_scale"
        return iter([m for n, m in self.named_parameters() if n.find(template) > -1])

    def lwc_parameters(self):
        return iter([m for n, m in self.named_parameters() if n.find("bound_factor") > -1])

    def omni_parameters(self, use_shift=True):
        template = "smooth" if use_shift else "smooth_scale"
        return iter([m for n, m in self.named_parameters()
                     if n.find("bound_factor") > -1 or n.find(template) > -1])

    def omni_state_dict(self, destination=None, prefix="", keep_vars=False):
        if destination is None:
            destination = OrderedDict()
        for name, param in self.named_parameters():
            if name.find("smooth") > -1 or name.find("bound_factor") > -1:
                destination[prefix + name] = param if keep_vars else param.detach()
        return destination

    def register_scales_and_zeros(self):
        for mod in self._quant_linears():
            mod.weight_quantizer.register_scales_and_zeros()
