"""Quantised OPT decoder block on the HIP path.  Surface = reference models/int_opt_layer.py.

q (after the 1/sqrt(hd) scaling), k and v are fake-quantised per token over the FULL hidden dim before the
head split (models/int_opt_layer.py:96-97,124-130), then viewed as [bs, T, heads, hd] for the attention GEMMs.
"""
from typing import Optional, Tuple

import torch
from torch import nn

from . import ops
from .block_common import QuantBlockMixin
from .linear import QuantLinear
from .matmul import QuantMatMul
from .norm import OmniLayerNorm


class QuantOPTAttention(nn.Module):
    def __init__(self, org_module: nn.Module, embed_dim: int, num_heads: int, dropout: float = 0.0,
                 is_decoder: bool = False, bias: bool = True, args=None, disable_act_quant=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        if (self.head_dim * num_heads) != self.embed_dim:
            raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {self.embed_dim}"
                             f" and `num_heads`: {num_heads}).")
        self.scaling = self.head_dim ** -0.5
        self.is_decoder = is_decoder
        self.k_proj = QuantLinear(org_module.k_proj, args.weight_quant_params, args.act_quant_params)
        self.v_proj = QuantLinear(org_module.v_proj, args.weight_quant_params, args.act_quant_params)
        self.q_proj = QuantLinear(org_module.q_proj, args.weight_quant_params, args.act_quant_params)
        self.out_proj = QuantLinear(org_module.out_proj, args.weight_quant_params, args.act_quant_params)
        self.qkt_matmul = QuantMatMul(args.q_quant_params, args.k_quant_params, matmul_func=torch.bmm)
        self.pv_matmul = QuantMatMul(args.p_quant_params, args.v_quant_params, matmul_func=torch.bmm)
        self.use_weight_quant = False
        self.use_act_quant = False

    def forward(self, hidden_states, key_value_states=None, past_key_value=None, attention_mask=None,
                layer_head_mask=None, output_attentions=False, residual=None, input_is_quantized=False, sib=None):
        if key_value_states is not None or past_key_value is not None or layer_head_mask is not None or output_attentions:
            raise NotImplementedError("the calibration hot path runs self-attention without cache / head masks")
        bsz, tgt_len, _ = hidden_states.size()
        nh, hd = self.num_heads, self.head_dim
        hq = hidden_states if input_is_quantized else self.q_proj.quantize_input(hidden_states)   # q/k/v share one pass
        q, k, v = QuantLinear.forward_siblings([self.q_proj, self.k_proj, self.v_proj], hq, sib)
        q = self.qkt_matmul.quant_x1(ops.ScaleFn.apply(q, self.scaling))
        k = self.qkt_matmul.quant_x2(k)
        v = self.pv_matmul.quant_x2(v)
        q, k, v = (t.view(bsz, tgt_len, nh, hd) for t in (q, k, v))
        mask = None
        if attention_mask is not None:
            if attention_mask.size() != (bsz, 1, tgt_len, tgt_len):
                raise ValueError(f"Attention mask should be of size {(bsz, 1, tgt_len, tgt_len)}, but is "
                                 f"{attention_mask.size()}")
            mask = attention_mask            # [bs,1,T,T]: every sample keeps its own mask (ops.SoftmaxFn)
        causal = ops.mask_is_causal(attention_mask)
        scores = self.qkt_matmul.scores(q, k, causal)
        probs = ops.SoftmaxFn.apply(scores, mask, 1.0, causal)
        probs = self.pv_matmul.quant_x1(probs)
        attn = self.pv_matmul.apply_probs(probs, v, causal).view(bsz, tgt_len, self.embed_dim)
        return self.out_proj(attn, residual=residual), None, None      # (+ residual in the GEMM store)

    def set_quant_state(self, weight_quant: bool = False, act_quant: bool = False):
        self.use_weight_quant = weight_quant
        self.use_act_quant = act_quant
        for m in self.modules():
            if isinstance(m, (QuantLinear, QuantMatMul)):
                m.set_quant_state(weight_quant, act_quant)


class QuantOPTDecoderLayer(QuantBlockMixin, nn.Module):
    def __init__(self, config, ori_layer, args):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.self_attn = QuantOPTAttention(org_module=ori_layer.self_attn, embed_dim=self.embed_dim,
                                           num_heads=config.num_attention_heads, dropout=config.attention_dropout,
                                           is_decoder=True, bias=config.enable_bias, args=args)
        self.do_layer_norm_before = config.do_layer_norm_before
        self.dropout = config.dropout
        self.self_attn_layer_norm = OmniLayerNorm(ori_layer.self_attn_layer_norm)
        self.fc1 = QuantLinear(ori_layer.fc1, weight_quant_params=args.weight_quant_params,
                               act_quant_params=args.act_quant_params)
        self.fc2 = QuantLinear(ori_layer.fc2, weight_quant_params=args.weight_quant_params,
                               act_quant_params=args.act_quant_params)
        self.final_layer_norm = OmniLayerNorm(ori_layer.final_layer_norm)
        self.type = ori_layer.fc1.weight.dtype

    def _let_names(self):
        a = self.self_attn
        return dict(q=a.q_proj, k=a.k_proj, v=a.v_proj, o=a.out_proj, fc1=[self.fc1], last=self.fc2,
                    ln1=self.self_attn_layer_norm, ln2=self.final_layer_norm)

    def forward(self, hidden_states, attention_mask=None, layer_head_mask=None, output_attentions=False,
                use_cache=False, past_key_value=None, **kwargs) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        if use_cache or output_attentions:
            raise NotImplementedError("the calibration hot path runs without KV cache / attention outputs")
        hidden_states, back = self._enter(hidden_states)
        residual = hidden_states
        a = self.self_attn
        fq = None
        if self.do_layer_norm_before:
            same = a.q_proj.act_quantizer is not None and all(
                getattr(m.act_quantizer, "n_bits", None) == a.q_proj.act_quantizer.n_bits and m.use_act_quant == a.q_proj.use_act_quant
                for m in (a.k_proj, a.v_proj))
            # LayerNorm -> input quantiser of q/k/v as one kernel when it is the plain dynamic per-token one (llama_block.py)
            fq = self.self_attn_layer_norm.forward_quant(hidden_states, a.q_proj, True, self.self_attn_layer_norm.eps) if same else None
        sib1 = sib2 = None
        if fq is not None:
            h, residual, sib1 = fq
        else:
            h = self.self_attn_layer_norm(hidden_states) if self.do_layer_norm_before else hidden_states
        hidden_states, _, _ = self.self_attn(hidden_states=h, past_key_value=past_key_value, attention_mask=attention_mask,
                                             layer_head_mask=layer_head_mask, output_attentions=output_attentions,
                                             residual=residual, input_is_quantized=fq is not None,
                                             sib=sib1)      # residual add folded into out_proj's GEMM store
        if not self.do_layer_norm_before:
            hidden_states = self.self_attn_layer_norm(hidden_states)
        fq2 = None
        if self.do_layer_norm_before:
            fq2 = self.final_layer_norm.forward_quant(hidden_states, self.fc1, True, self.final_layer_norm.eps)
            if fq2 is not None:
                h, residual, sib2 = fq2
            else:
                h, residual = self.final_layer_norm.forward_with_residual(hidden_states)    # residual-path grad joins in norm bwd
        else:
            h = residual = hidden_states
        hidden_states = self.fc2(ops.ReluFn.apply(self.fc1(h, fq2 is not None, sib=sib2)), residual=residual)
        if not self.do_layer_norm_before:
            hidden_states = self.final_layer_norm(hidden_states)
        return (self._leave(hidden_states, back),)
