"""Shared machinery of the quantised decoder blocks (LLaMA / OPT): quant-state switches, the fused
smooth-and-quant step, the final fold, parameter selectors and the omni state dict.

Reference: models/int_llama_layer.py:269-368 and models/int_opt_layer.py:348-452 (the two copies are
identical up to module names).
"""
from collections import OrderedDict

import os

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

    # ---- block boundary dtype -------------------------------------------------------------------------------------
    # The kernels compute in bfloat16 (MFMA bf16) or float32 (parity mode).  A float16 hidden state -- what the
    # reference's fp16 model hands to the layers it gets back from omniquant() (main.py evaluate) -- is converted to the
    # compute dtype on entry and the result is returned in float16 again, so the returned model keeps its dtype contract.
    def _enter(self, hidden_states):
        from . import ops
        if hidden_states.dtype == torch.float16:
            cd = self.compute_dtype if self.compute_dtype in (torch.bfloat16, torch.float32) else torch.bfloat16
            return ops.cast(hidden_states, cd), torch.float16
        return hidden_states, None

    @staticmethod
    def _leave(hidden_states, back):
        from . import ops
        return hidden_states if back is None else ops.cast(hidden_states, back)

    def _truncate_scales(self):
        arena = self.__dict__.get("_arena_truncate")
        if arena is not None:
            arena()                      # all LET scales are contiguous in the optimizer's arena: one launch
            return
        with torch.no_grad():
            for name, p in self.named_parameters():
                if "smooth_scale" in name:
                    L.truncate_number(p)

    def _f32c(self, t):
        """float32 copy of a frozen norm weight / bias buffer, cached (they never change during calibration)."""
        if t is None:
            return None
        if t.dtype == torch.float32:
            return t
        cache = self.__dict__.setdefault("_f32_cache", {})
        key = (id(t), t._version)
        hit = cache.get(id(t))
        if hit is None or hit[0] != key:
            cache[id(t)] = (key, t.detach().float().contiguous(), t)     # keep `t` alive so id() stays unique
        return cache[id(t)][1]

    # ---- second HIP stream for the weight path -----------------------------------------------------------------
    # The weight fake-quant kernels (HBM/VALU-bound, no LDS) and their backward depend only on the learnables and on
    # dL/dW, not on the activations, so they can run beside the MFMA-bound GEMMs of the activation path.  When
    # `_fq_stream` is set (StepRunner does it), they are enqueued on that stream; each QuantLinear waits for its
    # own weight's event right before its GEMM.  Autograd runs every backward node on the stream of its forward, so
    # the fake-quant BACKWARD kernels land on the side stream as well and overlap the remaining dgrad/wgrad GEMMs.
    def _weight_stream(self):
        """(stream, forked): the side stream when one is configured, else the current stream."""
        st = self.__dict__.get("_fq_stream")
        if st is None:
            return torch.cuda.current_stream(), False
        return st, True

    def _publish(self, mod, weight, bias, side, forked):
        mod.temp_weight, mod.temp_bias = weight, bias
        mod.use_temporary_parameter = True
        if forked:
            ev = torch.cuda.Event()
            ev.record(side)
            mod._temp_ready = ev
        else:
            mod._temp_ready = None

    def _weight_slabs(self, nm, dtype):
        """{module: destination of its fake-quant weight}: q | k | v (and gate | up) are written back to back into one
        buffer each, so that the sibling projections run as ONE GEMM per direction (ops.stacked_rows).  Per step, from the
        caching allocator (inside the captured graph's pool the addresses are static)."""
        dest = {}
        if os.environ.get("OQ_STACKED_GEMM", "1") == "0":
            return dest
        for group in ([nm["q"], nm["k"], nm["v"]], list(nm["fc1"])):
            # (the reference lists fc1 as [up_proj, gate_proj]; the MLP consumes gate first)
            if group and len(group) > 1 and all(isinstance(m, QuantLinear) for m in group) and \
                    len({(m.in_features, m.weight.device) for m in group}) == 1 and \
                    not any(m.weight_quantizer._identity() for m in group):
                if group[0] is not nm["q"]:
                    group = sorted(group, key=lambda m: 0 if m is getattr(getattr(self, "mlp", None), "gate_proj", None) else 1)
                from . import ops
                dev = group[0].weight.device
                rows_all, cols = sum(m.out_features for m in group), group[0].in_features
                slab = torch.empty((rows_all, cols), dtype=dtype, device=dev)
                # the LET by-product w @ shift of gate / up IS their bias (no vector kernel in between): stacked as well
                ws = torch.empty((rows_all,), dtype=torch.float32, device=dev) if group[0] is not nm["q"] else None
                # integer side channel (ops.IntCodes) of the siblings, stacked the same way: codes + scale | zp | csum vectors
                codes = vec = None
                if all(m.int_fprop_eligible(dtype) for m in group):
                    codes = torch.empty((rows_all, cols), dtype=torch.int8, device=dev)
                    vec = torch.empty((3, rows_all), dtype=torch.float32, device=dev)
                r = 0
                for m in group:
                    n = m.out_features
                    d = ops.WeightDest(slab[r:r + n], None if ws is None else ws[r:r + n])
                    if codes is not None:
                        d.codes, d.scale, d.zp, d.csum = codes[r:r + n], vec[0, r:r + n].view(n, 1), vec[1, r:r + n].view(n, 1), vec[2, r:r + n]
                    dest[m] = d
                    r += n
        return dest

    def _let_temporaries(self, out_dtype, lazy_mlp=False, stack=False):
        """Fused LET path: 6 fused transform+fake-quant kernels (one per smoothed linear) + ONE vector kernel for
        every norm weight/bias and projection bias.  Returns {module: (temp_weight, temp_bias)} and the norm temps.
        lazy_mlp: the MLP weights (no coupling with the vector kernel) are NOT quantised here; instead
        `lazy[mod]` is a closure that does it right before the module's GEMM, so that the 0.27 GB of MLP fake-quant
        weights do not flush the attention projections' weights out of the 256 MB Infinity Cache before they are used
        (same kernels, same values, different issue order)."""
        nm = self._let_names()
        from . import ops
        hidden = self.qkv_smooth_scale.numel()
        for key in ("q", "k", "v"):
            if nm[key].out_features != hidden:
                # smooth_q_k / smooth_fc_fc pair q, k, v and o row by row (models/transformation.py:44-69): the reference
                # dies on a broadcast error for grouped-query attention; fail as loudly instead of reading out of bounds
                raise NotImplementedError(f"LET with {key}_proj.out_features = {nm[key].out_features} != hidden size "
                                          f"{hidden}: grouped-query attention has no LET (use --lwc only)")
        specs = L.block_let_specs(nm, self, None)
        f = self._f32c
        mlp = set(nm["fc1"]) | {nm["last"]} if lazy_mlp else set()
        wq, ws, lazy = {}, {}, {}
        dest = self._weight_slabs(nm, out_dtype) if stack else {}

        def quant(mod, sp):
            # + integer codes for the int8 fprop (ops.IntCodes): of the step's temporaries, and of the FOLDED weights (float32
            # values, same codes) for the propagate pass that follows the fold
            want_int = mod.int_fprop_eligible(out_dtype if stack else self.compute_dtype)
            if sp.shift is not None:
                return mod.weight_quantizer.quantize(mod.weight, out_dtype=out_dtype, col_mul=sp.col_mul,
                                                     row_div=sp.row_div, row_mul=sp.row_mul, shift=sp.shift, out=dest.get(mod),
                                                     want_int=want_int)
            return mod.weight_quantizer.quantize(mod.weight, out_dtype=out_dtype, out=dest.get(mod), want_int=want_int), None

        def mlp_bias(mod, wsh):
            if mod is nm["last"]:
                return mod.bias
            return wsh if mod.bias is None else f(mod.bias) + wsh

        # issue order: the attention projections' weights are quantised LAST-USED-FIRST (o, v, k, q), so that the ones the
        # first GEMMs read are the most recently written and still sit in the 256 MB Infinity Cache (same kernels, same values)
        order = list(specs.items())
        if os.environ.get("OQ_WQ_ORDER", "1") != "0":
            rank = {nm["o"]: 0, nm["v"]: 1, nm["k"]: 2, nm["q"]: 3}
            order.sort(key=lambda it: rank.get(it[0], -1))
        fc1_group = [m for m in nm["fc1"] if m in mlp and m in specs]
        # gate / up in one quantiser launch per direction: both feed ONE stacked GEMM now, so nothing is gained any more by
        # quantising each right before "its" GEMM (+0.3 % same-box; before the stacked launches it cost 0.9 %)
        batch_fc1 = len(fc1_group) > 1 and os.environ.get("OQ_WQ_BATCH_MLP", "1") != "0"
        with ops.WeightQuantBatch():        # the weights quantised here and now (q, k, v, o) share one launch per direction
            for mod, sp in order:
                if mod in mlp:
                    if batch_fc1 and mod in fc1_group:
                        def make(mod=mod):
                            # whichever of gate / up is needed first quantises both in one launch
                            res = {}
                            with ops.WeightQuantBatch():
                                for m2 in fc1_group:
                                    res[m2] = quant(m2, specs[m2])
                            for m2 in fc1_group:
                                m2.temp_weight, m2.temp_bias = res[m2][0], mlp_bias(m2, res[m2][1])
                                if m2 is not mod:
                                    m2.__dict__.pop("_lazy_temp", None)
                    else:
                        def make(mod=mod, sp=sp):
                            w_, ws_ = quant(mod, sp)
                            mod.temp_weight, mod.temp_bias = w_, mlp_bias(mod, ws_)
                    lazy[mod] = make
                else:
                    wq[mod], ws[mod] = quant(mod, sp)
        q, k, v, o, ln1, ln2 = nm["q"], nm["k"], nm["v"], nm["o"], nm["ln1"], nm["ln2"]
        ln1_tw, ln1_tb, ln2_tw, ln2_tb, b_q, b_k, b_v, b_o = ops.LetVectorsFn.apply(
            self.qkv_smooth_scale, self.qkv_smooth_shift, self.out_smooth_scale, self.out_smooth_shift,
            self.fc1_smooth_scale, self.fc1_smooth_shift, self.qkt_smooth_scale,
            ws[q], ws[k], ws[v], ws[o], f(ln1.weight), f(getattr(ln1, "bias", None)), f(ln2.weight),
            f(getattr(ln2, "bias", None)), f(q.bias), f(k.bias), f(v.bias), f(o.bias))
        bias = {q: b_q, k: b_k, v: b_v, o: b_o}
        for mod in list(wq):
            if mod not in bias:
                bias[mod] = mlp_bias(mod, ws[mod])
        self.__dict__["_lazy_let"] = lazy
        return wq, bias, (ln1_tw, ln1_tb), (ln2_tw, ln2_tb)

    def smooth_and_quant_temporary(self):
        """LET re-parameterisation + weight fake-quant of every linear: ONE fused kernel per weight matrix plus one
        vector kernel (reference: ~25 eager passes per matrix, models/int_llama_layer.py:279-307)."""
        nm = self._let_names()
        dt = self.compute_dtype
        main = torch.cuda.current_stream()
        side, forked = self._weight_stream()
        if self.let:
            self._truncate_scales()
            if forked:
                side.wait_stream(main)      # parameters were updated (AdamW / truncate) on the main stream
            lazy_mlp = bool(self.__dict__.get("_lazy_mlp_quant")) and not forked
            with torch.cuda.stream(side):
                wq, bias, t1, t2 = self._let_temporaries(dt, lazy_mlp, stack=True)
                for mod in wq:
                    self._publish(mod, wq[mod], bias[mod], side, forked)
                for mod, fn in self.__dict__.pop("_lazy_let", {}).items():
                    mod.temp_weight, mod.temp_bias = None, None
                    mod.use_temporary_parameter = True
                    mod._temp_ready = None
                    mod.__dict__["_lazy_temp"] = fn
            if forked:
                main.wait_stream(side)      # norm weights/biases are needed at the very start of the block
                # (the per-linear events above are then already satisfied; LWC-only blocks below overlap more)
            for ln, (tw, tb) in ((nm["ln1"], t1), (nm["ln2"], t2)):
                ln.temp_weight, ln.temp_bias = tw, tb
                ln.use_temporary_parameter = True
        else:
            if forked:
                side.wait_stream(main)
            lazy_mlp = bool(self.__dict__.get("_lazy_mlp_quant")) and not forked
            mlp = set(nm["fc1"]) | {nm["last"]} if lazy_mlp else set()
            with torch.cuda.stream(side):
                dest = self._weight_slabs(nm, dt)      # (allocated on the stream that writes them)
                mods = list(self._quant_linears())
                if os.environ.get("OQ_WQ_ORDER", "1") != "0":       # last-used-first, see _let_temporaries
                    rank = {nm["o"]: 0, nm["v"]: 1, nm["k"]: 2, nm["q"]: 3}
                    mods.sort(key=lambda m: rank.get(m, -1))
                for mod in mods:
                    if mod in mlp:
                        def make(mod=mod):
                            mod.temp_weight, mod.temp_bias = mod.weight_quantizer.quantize(
                                mod.weight, out_dtype=dt, out=dest.get(mod), want_int=mod.int_fprop_eligible(dt)), mod.bias
                        mod.temp_weight, mod.temp_bias = None, None
                        mod.use_temporary_parameter = True
                        mod._temp_ready = None
                        mod.__dict__["_lazy_temp"] = make
                    else:
                        self._publish(mod, mod.weight_quantizer.quantize(mod.weight, out_dtype=dt, out=dest.get(mod),
                                                                         want_int=mod.int_fprop_eligible(dt)), mod.bias, side, forked)

    def clear_temp_variable(self):
        nm = self._let_names()
        for mod in self._quant_linears() + [nm["ln1"], nm["ln2"]]:
            mod.__dict__.pop("_lazy_temp", None)
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
            wq, bias, t1, t2 = self._let_temporaries(torch.float32)
            for ln, (tw, tb) in ((nm["ln1"], t1), (nm["ln2"], t2)):
                ln.use_temporary_parameter = False
                if hasattr(ln, "bias"):
                    del ln.bias
                ln.register_buffer("bias", tb)
                ln.weight = tw
            for mod in wq:
                if bias[mod] is not None and mod is not nm["last"]:
                    if hasattr(mod, "bias"):
                        del mod.bias
                    mod.register_buffer("bias", bias[mod])
                mod.weight = wq[mod]
                mod.use_temporary_parameter = False
        else:
            for mod in self._quant_linears():
                mod.weight = mod.weight_quantizer.quantize(mod.weight, out_dtype=torch.float32,
                                                           want_int=mod.int_fprop_eligible(self.compute_dtype))
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
