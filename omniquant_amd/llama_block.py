"""Quantised LLaMA decoder block on the HIP path.  Surface = reference models/int_llama_layer.py.

Data layout: projections produce [bs, T, heads*hd]; q/k/v stay in that head-interleaved layout (viewed as
[bs, T, heads, hd]) through RoPE, head-wise fake-quant and both attention GEMMs (strided-batched, no
transposes).  GQA (num_key_value_heads < num_attention_heads) is handled by the GEMM's zero batch stride.
"""
import math
import os
from typing import Optional, Tuple

import torch
from torch import nn

from . import ops
from .block_common import QuantBlockMixin
from .linear import QuantLinear
from .matmul import QuantMatMul
from .norm import OmniLlamaRMSNorm


def _rope_theta(config):
    rp = getattr(config, "rope_parameters", None)
    if isinstance(rp, dict) and "rope_theta" in rp:
        return float(rp["rope_theta"])
    return float(getattr(config, "rope_theta", 10000.0))


class QuantLlamaMLP(nn.Module):
    def __init__(self, org_module: nn.Module, hidden_size: int, intermediate_size: int, hidden_act: str, args=None):
        super().__init__()
        self.gate_proj = QuantLinear(org_module.gate_proj, args.weight_quant_params, args.act_quant_params)
        self.down_proj = QuantLinear(org_module.down_proj, args.weight_quant_params, args.act_quant_params)
        self.up_proj = QuantLinear(org_module.up_proj, args.weight_quant_params, args.act_quant_params)
        if hidden_act not in ("silu", "swish"):
            raise NotImplementedError(f"hidden_act {hidden_act}: only SiLU has a HIP kernel")

    def forward(self, x, residual=None, input_is_quantized=False, sib=None):
        from .linear import _hip_linear
        # the block OUTPUT feeds no rounding decision (only the loss): its float32 side channel is off unless asked for
        wide_out = os.environ.get("OQ_WIDE_OUT", "0") != "0"
        xq = x if input_is_quantized else self.gate_proj.quantize_input(x)   # gate/up share one act-quant pass
        dq = self.down_proj.act_quantizer
        fuse_q = (self.down_proj.use_act_quant and dq is not None and not self.down_proj.disable_input_quant and dq.enable
                  and dq.n_bits < 16 and not dq.symmetric and dq.metric != "fix0to1" and not dq.group_size
                  and dq.dynamic_method == "per_token" and not dq.lwc and self.down_proj.__dict__.get("_stat_sink") is None)
        g_, u_ = self.gate_proj, self.up_proj
        if all(m.use_temporary_parameter and m.fwd_func is _hip_linear and not m.fwd_kwargs
               and m.__dict__.get("_stat_sink") is None for m in (g_, u_)) and xq.dtype in (torch.bfloat16, torch.float32) \
                and g_.out_features % 8 == 0:
            xint = getattr(xq, "_oq_int", None)
            (wg, bg), (wu, bu) = g_._resolve(xq.dtype), u_._resolve(xq.dtype)
            if wg.dtype == xq.dtype and ops.stacked_rows([wg, wu]) is not None and ops.stacked_vectors([bg, bu]) is not False:
                # the two fake-quant weights are row blocks of one buffer (block_common._weight_slabs): gate | up as ONE GEMM
                # per direction, silu*up (-> down_proj input quantiser) reading the column blocks in place
                nb = dq.n_bits if (fuse_q and ops.silu_mul_quant_supported(xq.new_empty((0, g_.out_features)), dq.n_bits)) else 0
                # integer codes of the operands (ops.IntCodes), when all three carry them: exact int8 fprop
                wints = (getattr(wg, "_oq_int", None), getattr(wu, "_oq_int", None))
                if xint is None or any(w is None for w in wints) or not ops.int_fprop_on():
                    xint = wints = None
                stash = {"want_int": True} if (nb and self.down_proj.weight_has_codes()
                                               and self.down_proj.int_fprop_eligible(xq.dtype)) else {}
                act = ops.StackedGateUpFn.apply(xq, wg, bg, wu, bu, nb, stash, sib, xint, wints)
                if nb:
                    dq.scale, dq.round_zero_point = stash["scale"], stash["zp"]
                    if stash.get("int") is not None:
                        act._oq_int = stash["int"]
                return self.down_proj(act, input_is_quantized=bool(nb), residual=residual, wide=wide_out)
        gate, up = QuantLinear.forward_siblings([self.gate_proj, self.up_proj], xq, sib)
        if fuse_q and ops.silu_mul_quant_supported(gate, dq.n_bits):
            # act_fn(gate) * up and the down_proj input quantiser in ONE kernel: the product is never stored
            stash = {"want_int": True} if (self.down_proj.weight_has_codes()
                                           and self.down_proj.int_fprop_eligible(gate.dtype)) else {}
            act = ops.SiluMulQuantFn.apply(gate, up, dq.n_bits, stash)
            dq.scale, dq.round_zero_point = stash["scale"], stash["zp"]
            if stash.get("int") is not None:
                act._oq_int = stash["int"]
            return self.down_proj(act, input_is_quantized=True, residual=residual, wide=wide_out)
        return self.down_proj(ops.SiluMulFn.apply(gate, up), residual=residual, wide=wide_out)   # residual add fused into the GEMM store


class QuantLlamaAttention(nn.Module):
    def __init__(self, org_module: nn.Module, config, args=None):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = getattr(config, "num_key_value_heads", None) or self.num_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.max_position_embeddings = config.max_position_embeddings
        if (self.head_dim * self.num_heads) != self.hidden_size:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                             f" and `num_heads`: {self.num_heads}).")
        self.rope_theta = _rope_theta(config)
        self._rope_cache = None
        self.k_proj = QuantLinear(org_module.k_proj, args.weight_quant_params, args.act_quant_params)
        self.v_proj = QuantLinear(org_module.v_proj, args.weight_quant_params, args.act_quant_params)
        self.q_proj = QuantLinear(org_module.q_proj, args.weight_quant_params, args.act_quant_params)
        self.o_proj = QuantLinear(org_module.o_proj, args.weight_quant_params, args.act_quant_params)
        self.qkt_matmul = QuantMatMul(args.q_quant_params, args.k_quant_params, matmul_func=torch.matmul)
        self.pv_matmul = QuantMatMul(args.p_quant_params, args.v_quant_params, matmul_func=torch.matmul)
        self.use_weight_quant = False
        self.use_act_quant = False

    def _rope_tables(self, position_ids, T, device):
        """cos/sin [T, hd] f32 gathered by position_ids (transformers-4.31 LlamaRotaryEmbedding formula; the reference
        indexes its cos/sin cache with position_ids, models/int_llama_layer.py:124-125).  position_ids: None (= arange),
        [T] or [bs, T] whose rows are all equal -- per-row positions would need per-sample tables, which the calibration
        loop never produces; they are rejected instead of silently using row 0.  The cache holds the position_ids tensor
        itself (identity + version counter), so a new tensor at a recycled address can never hit a stale entry."""
        c = self._rope_cache
        if c is not None and c[0] == (T, str(device)) and c[1] is position_ids and \
                (position_ids is None or c[2] == position_ids._version):
            return c[3], c[4]
        hd = self.head_dim
        inv = 1.0 / (self.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32, device=device) / hd))
        if position_ids is None:
            pos = torch.arange(T, dtype=torch.float32, device=device)
        else:
            pid = position_ids.reshape(-1, position_ids.shape[-1])
            if pid.shape[-1] != T:
                raise ValueError(f"position_ids of length {pid.shape[-1]} for a sequence of length {T}")
            if pid.shape[0] > 1 and not bool((pid == pid[:1]).all()):
                raise NotImplementedError("per-sample position_ids (rows differ) are not supported on the HIP path")
            pos = pid[0].to(device=device, dtype=torch.float32)
        fr = torch.outer(pos, inv)
        emb = torch.cat((fr, fr), dim=-1)
        cos, sin = emb.cos().contiguous(), emb.sin().contiguous()
        self._rope_cache = ((T, str(device)), position_ids, None if position_ids is None else position_ids._version, cos, sin)
        return cos, sin

    def _fused_rope_quant(self, hq):
        """True when q / k / v can take the fused projection -> RoPE -> head-wise quant node: the three head quantisers are
        the plain dynamic per-token asymmetric ones (what main.py builds), head_dim 128, HIP GEMM projections."""
        from .linear import _hip_linear
        if not (self.qkt_matmul.use_act_quant and self.pv_matmul.use_act_quant):
            return False
        for qz in (self.qkt_matmul.x1_quantizer, self.qkt_matmul.x2_quantizer, self.pv_matmul.x2_quantizer):
            if not (qz.enable and 2 <= qz.n_bits < 16 and not qz.symmetric and not qz.lwc and not qz.group_size
                    and qz.metric != "fix0to1" and qz.dynamic_method == "per_token"):
                return False
        for lin in (self.q_proj, self.k_proj, self.v_proj):
            if lin.fwd_func is not _hip_linear or lin.fwd_kwargs or lin.__dict__.get("_stat_sink") is not None:
                return False
        return ops.rope_quant_supported(hq.dtype, self.head_dim)

    def _fused_rope_split(self, hq):
        """True when q / k / v can take the same node on the IDENTITY grid: the three head quantisers are off (or >= 16 bit:
        weight-only configurations, quantize/quantizer.py:109-110), head_dim 128, HIP GEMM projections, bf16.  The projections
        then run as one stacked GEMM per direction and RoPE + the q | k | v split are one launch per direction."""
        from .linear import _hip_linear
        if os.environ.get("OQ_MERGED_QKV", "1") == "0" or hq.dtype != torch.bfloat16:
            return False
        for mm, qz in ((self.qkt_matmul, self.qkt_matmul.x1_quantizer), (self.qkt_matmul, self.qkt_matmul.x2_quantizer),
                       (self.pv_matmul, self.pv_matmul.x2_quantizer)):
            if mm.use_act_quant and qz.enable and qz.n_bits < 16:
                return False
        for lin in (self.q_proj, self.k_proj, self.v_proj):
            if lin.fwd_func is not _hip_linear or lin.fwd_kwargs or lin.__dict__.get("_stat_sink") is not None:
                return False
        return ops.rope_quant_supported(hq.dtype, self.head_dim)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                output_attentions=False, use_cache=False, residual=None, input_is_quantized=False, sib=None):
        if past_key_value is not None or use_cache or output_attentions:
            raise NotImplementedError("the calibration hot path runs without KV cache / attention outputs")
        bsz, q_len, _ = hidden_states.size()
        nh, nkv, hd = self.num_heads, self.num_key_value_heads, self.head_dim
        hq = hidden_states if input_is_quantized else self.q_proj.quantize_input(hidden_states)   # q/k/v share one pass
        cos, sin = self._rope_tables(position_ids, q_len, hidden_states.device)
        causal = ops.mask_is_causal(attention_mask)   # exact causal mask -> the masked half is skipped everywhere
        pq = self.pv_matmul.x1_quantizer
        p_identity = (not self.pv_matmul.use_act_quant) or pq.n_bits >= 16 or not pq.enable
        fused_qkv = self._fused_rope_quant(hq)
        grid_scales = None
        split_qkv = (not fused_qkv) and self._fused_rope_split(hq)
        if split_qkv:
            wbs = []
            for lin in (self.q_proj, self.k_proj, self.v_proj):
                w, b = lin._resolve(hq.dtype)
                wbs.append((w if w.dtype == hq.dtype else ops.cast(w, hq.dtype), b))
            q, k, v = ops.QKVRopeQuantFn.apply(hq, wbs[0][0], wbs[0][1], wbs[1][0], wbs[1][1], wbs[2][0], wbs[2][1], cos, sin,
                                               16, hd, None, sib)
            fused_qkv = True            # (q, k, v are final: the head quantisers are the identity)
        elif fused_qkv:
            # projection -> RoPE -> head-wise fake quant as one node per tensor: the projection output stays fp32 inside it
            trio = ((self.q_proj, self.qkt_matmul.x1_quantizer, True), (self.k_proj, self.qkt_matmul.x2_quantizer, True),
                    (self.v_proj, self.pv_matmul.x2_quantizer, False))
            wbs = []
            for lin, _, _ in trio:
                w, b = lin._resolve(hq.dtype)
                if w.dtype != hq.dtype:
                    w = ops.cast(w, hq.dtype)
                wbs.append((w, b))
            same_bits = len({qz.n_bits for _, qz, _ in trio}) == 1
            if same_bits and os.environ.get("OQ_MERGED_QKV", "1") != "0":
                # one node for the three: RoPE + head quant of all heads in one launch per direction, one bias column sum
                stashes = [{}, {}, {}]
                # integer codes of the shared input and of the three weights (ops.IntCodes), when all carry them: exact int8 fprop
                xint = getattr(hq, "_oq_int", None)
                wints = tuple(getattr(w, "_oq_int", None) for w, _ in wbs)
                if xint is None or any(w is None for w in wints) or not ops.int_fprop_on():
                    xint = wints = None
                # attention on the head quantisers' integer grid (bf16 production mode, fused causal kernels): q, k, v come back
                # as grid coordinates, their scales stay in the merged per-(token, head) vector
                grid = (ops.grid_attention_on() and hq.dtype == torch.bfloat16 and trio[0][1].n_bits <= 8 and p_identity
                        and bool(causal) and ops.fused_attention_shape_supported(hq.dtype, q_len, hd))
                q, k, v = ops.QKVRopeQuantFn.apply(hq, wbs[0][0], wbs[0][1], wbs[1][0], wbs[1][1], wbs[2][0], wbs[2][1], cos, sin,
                                                   trio[0][1].n_bits, hd, stashes, sib, xint, wints, grid)
                for (_, qz, _), st in zip(trio, stashes):
                    qz.scale, qz.round_zero_point = st["scale"], st["zp"]
                if grid:
                    grid_scales = tuple(st["scale"] for st in stashes)
            else:
                outs = []
                for (lin, qz, rot), (w, b) in zip(trio, wbs):
                    stash = {}
                    outs.append(ops.LinearRopeQuantFn.apply(hq, w, b, cos if rot else None, sin if rot else None, qz.n_bits, hd, stash, sib))
                    qz.scale, qz.round_zero_point = stash["scale"], stash["zp"]
                q, k, v = outs
        else:
            q, k, v = QuantLinear.forward_siblings([self.q_proj, self.k_proj, self.v_proj], hq, sib)
            q, k, v = q.view(bsz, q_len, nh, hd), k.view(bsz, q_len, nkv, hd), v.view(bsz, q_len, nkv, hd)
            q = ops.RopeFn.apply(q, cos, sin)
            k = ops.RopeFn.apply(k, cos, sin)
            # head-wise (per head, per token) fake quant over head_dim; repeat_kv commutes with it
            q = self.qkt_matmul.quant_x1(q)
            k = self.qkt_matmul.quant_x2(k)
        mask = None
        if attention_mask is not None:
            if attention_mask.size() != (bsz, 1, q_len, q_len):
                raise ValueError(f"Attention mask should be of size {(bsz, 1, q_len, q_len)}, but is "
                                 f"{attention_mask.size()}")
            mask = attention_mask            # [bs,1,T,T]: every sample keeps its own mask (ops.SoftmaxFn)
        wide = None
        if grid_scales is not None:
            # (decided above: exact causal mask, identity p-quantiser, fused kernels cover the shape)
            stash = {} if ops.wide_on() else None
            attn = ops.FusedCausalAttnFn.apply(q, k, v, 1.0 / math.sqrt(hd), grid_scales, stash)
            wide = stash.get("wide") if stash is not None else None
        elif p_identity and ops.fused_attention_supported(q, causal):
            # exact causal mask + identity p-quantiser (the reference default, 16 bit): one fused kernel per direction,
            # the [nh, T, T] scores / probabilities never touch HBM
            if not fused_qkv:
                v = self.pv_matmul.quant_x2(v)
            attn = ops.FusedCausalAttnFn.apply(q, k, v, 1.0 / math.sqrt(hd))
        else:
            scores = self.qkt_matmul.scores(q, k, causal)               # [bs, nh, T, T], unscaled
            probs = ops.SoftmaxFn.apply(scores, mask, 1.0 / math.sqrt(hd), causal)   # scale, +mask, clamp, f32 softmax
            probs = self.pv_matmul.quant_x1(probs)
            if not fused_qkv:
                v = self.pv_matmul.quant_x2(v)
            attn = self.pv_matmul.apply_probs(probs, v, causal)        # [bs, T, nh, hd]
        attn = attn.view(bsz, q_len, self.hidden_size)
        if wide is not None:
            attn._oq_wide = wide.view(bsz, q_len, self.hidden_size)     # un-rounded output for the o_proj input quantiser
        attn = self.o_proj(attn, residual=residual, wide=True)           # (+ residual in the GEMM store)
        return attn, None, None

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        for m in self.modules():
            if isinstance(m, (QuantLinear, QuantMatMul)):
                m.set_quant_state(weight_quant, act_quant)


class QuantLlamaDecoderLayer(QuantBlockMixin, nn.Module):
    def __init__(self, config, ori_layer, args):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = QuantLlamaAttention(org_module=ori_layer.self_attn, config=config, args=args)
        self.mlp = QuantLlamaMLP(org_module=ori_layer.mlp, hidden_size=self.hidden_size,
                                 intermediate_size=config.intermediate_size, hidden_act=config.hidden_act, args=args)
        self.input_layernorm = OmniLlamaRMSNorm(ori_layer.input_layernorm, eps=ori_layer.input_layernorm.variance_epsilon)
        self.post_attention_layernorm = OmniLlamaRMSNorm(ori_layer.post_attention_layernorm,
                                                         eps=ori_layer.post_attention_layernorm.variance_epsilon)

    def _let_names(self):
        a, m = self.self_attn, self.mlp
        return dict(q=a.q_proj, k=a.k_proj, v=a.v_proj, o=a.o_proj, fc1=[m.up_proj, m.gate_proj], last=m.down_proj,
                    ln1=self.input_layernorm, ln2=self.post_attention_layernorm)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                output_attentions=False, use_cache=False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        hidden_states, back = self._enter(hidden_states)
        residual = hidden_states
        a = self.self_attn
        same_q = a.q_proj.act_quantizer is not None and all(
            getattr(m.act_quantizer, "n_bits", None) == a.q_proj.act_quantizer.n_bits and m.use_act_quant == a.q_proj.use_act_quant
            for m in (a.k_proj, a.v_proj))
        # norm -> input quantiser of q/k/v (resp. gate/up) as ONE kernel when those quantisers are the plain dynamic
        # per-token ones: the normalised activations are never stored (quantize/omni_norm.py + quantize/int_linear.py:59-60)
        fq = self.input_layernorm.forward_quant(hidden_states, a.q_proj, False, self.input_layernorm.variance_epsilon) if same_q else None
        sib1 = sib2 = None
        if fq is not None:
            h, residual, sib1 = fq
        else:
            h = self.input_layernorm(hidden_states)
        # the two residual adds (models/int_llama_layer.py:246,264) are folded into the o_proj / down_proj GEMM stores
        hidden_states, _, _ = self.self_attn(hidden_states=h, attention_mask=attention_mask, position_ids=position_ids,
                                             past_key_value=past_key_value, output_attentions=output_attentions,
                                             use_cache=use_cache, residual=residual, input_is_quantized=fq is not None, sib=sib1)
        m = self.mlp
        same_m = m.gate_proj.act_quantizer is not None and getattr(m.up_proj.act_quantizer, "n_bits", None) == m.gate_proj.act_quantizer.n_bits \
            and m.up_proj.use_act_quant == m.gate_proj.use_act_quant
        fq2 = self.post_attention_layernorm.forward_quant(hidden_states, m.gate_proj, False,
                                                          self.post_attention_layernorm.variance_epsilon) if same_m else None
        if fq2 is not None:
            h, res, sib2 = fq2
        else:
            h, res = self.post_attention_layernorm.forward_with_residual(hidden_states)   # residual-path grad joins in norm bwd
        hidden_states = self.mlp(h, residual=res, input_is_quantized=fq2 is not None, sib=sib2)
        return (self._leave(hidden_states, back),)
