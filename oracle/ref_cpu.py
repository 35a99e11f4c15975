"""CPU oracle for the OmniQuant calibration hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm.  It is the
checker, never the product: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it.  `omniquant_amd/` never does (tests/test_no_oracle_leak.py
enforces that).

Parity status: PINNED.  Every function here is checked against golden vectors produced by
importing the reference itself in the build container (tests/golden/gen_golden.py ->
tests/golden/*.npz; tests/test_oracle_vs_golden.py).

Besides the pinned fp32 arithmetic, `Block.forward(act_dtype=...)` / `Block.temporaries(store_dtype=...)` can round
every tensor the product path materialises in bf16 (value and gradient) while keeping fp32 arithmetic: a CPU model of
the bf16 production mode, used only by tests/test_fullsize_parity.py to separate "precision mode" from "kernel bug".
The default (None) is the reference's arithmetic and the only mode the golden vectors pin.

Reference lines restated (all under /root/reference):
  fake_quant_*            quantize/quantizer.py:15-19 (round_ste), :84-105 (fake_quant),
                          :122-147 (per_token_dynamic_calibration)
  fake_quant_backward     closed form of the autograd of the above
  truncate_small          models/transformation.py:5-20
  let_init_scale          quantize/omniquant.py:189-191
  Block.temporaries       models/transformation.py:24-69, models/int_llama_layer.py:279-307,
                          models/int_opt_layer.py:385-413
  Block.forward           models/int_llama_layer.py:103-179,213-267 / models/int_opt_layer.py:81-213,268-346,
                          quantize/int_linear.py:48-65, quantize/int_matmul.py:31-43, quantize/omni_norm.py:26-63
  Block.fold              models/transformation.py:71-114, models/int_llama_layer.py:315-332
  adamw_step/grad_norm    torch.optim.AdamW semantics used at quantize/omniquant.py:207-208; utils.py:11-24
  calibrate               quantize/omniquant.py:157-250 (fp32, GradScaler disabled as on CPU)
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

CLIPMIN = 1e-5


# ----------------------------------------------------------------------------------------------
# quantizer
# ----------------------------------------------------------------------------------------------
def _rne_ste(t):
    """round-half-to-even forward, identity backward."""
    return t + (torch.round(t) - t).detach()


def fake_quant(x, n_bits, group_size=None, up=None, low=None, symmetric=False, return_qparams=False):
    """Dynamic min/max uniform affine fake quantisation over the last dim (or over groups of the
    last dim of a 2-D tensor).  `up`/`low` are the LWC logits [rows*groups, 1] or None."""
    if n_bits >= 16:
        return (x, None, None) if return_qparams else x
    shape = x.shape
    deficiency = 0
    xg = x
    if group_size:
        assert x.dim() == 2, "grouped quantisation is defined for 2-D weights only"
        rem = shape[1] % group_size
        if rem:
            assert symmetric, "ragged groups are only defined for the symmetric grid"
            deficiency = group_size - rem
            xg = torch.cat((x, x.new_zeros(shape[0], deficiency)), dim=1)
        xg = xg.reshape(-1, group_size)
    lo = xg.amin(dim=-1, keepdim=True)
    hi = xg.amax(dim=-1, keepdim=True)
    if up is not None:
        hi = torch.sigmoid(up) * hi
        lo = torch.sigmoid(low) * lo
    if symmetric:
        levels = 2 ** (n_bits - 1) - 1
        scale = (torch.maximum(hi.abs(), lo.abs()) / levels).clamp(min=CLIPMIN, max=1e4)
        zp = torch.full_like(scale, float(levels))
    else:
        scale = (hi - lo) / (2 ** n_bits - 1)          # NOT clamped (reference quirk Q1)
        zp = (-lo / scale).clamp(min=-1e4, max=1e4).round()   # plain round: zero gradient
    q = (_rne_ste(xg / scale) + zp).clamp(0, 2 ** n_bits - 1)
    y = (q - zp) * scale
    if group_size:
        y = y.reshape(shape[0], -1)
        if deficiency:
            y = y[:, :-deficiency]
    if return_qparams:
        return y, scale, zp
    return y


def fake_quant_backward(x, G, n_bits, group_size=None, up=None, low=None):
    """Closed-form gradient of asymmetric `fake_quant` (no autograd).  Returns (gx, gup, glow).
    Ties in amax/amin share the gradient equally, as torch.amax/amin do."""
    shape = x.shape
    xg = x.reshape(-1, group_size) if group_size else x.reshape(-1, shape[-1])
    Gg = G.reshape(xg.shape)
    Q = float(2 ** n_bits - 1)
    hi = xg.amax(-1, keepdim=True)
    lo = xg.amin(-1, keepdim=True)
    su = torch.sigmoid(up) if up is not None else torch.ones_like(hi)
    sl = torch.sigmoid(low) if low is not None else torch.ones_like(lo)
    s = (su * hi - sl * lo) / Q
    z = (-(sl * lo) / s).clamp(-1e4, 1e4).round()
    v = torch.round(xg / s) + z
    m = ((v >= 0) & (v <= Q)).to(x.dtype)
    q = v.clamp(0, Q)
    dy_ds = (q - z) - m * (xg / s)
    gs = (Gg * dy_ds).sum(-1, keepdim=True)
    gup = gs * hi / Q * su * (1 - su)
    glow = gs * (-lo / Q) * sl * (1 - sl)
    is_hi = (xg == hi).to(x.dtype)
    is_lo = (xg == lo).to(x.dtype)
    gx = Gg * m + is_hi / is_hi.sum(-1, keepdim=True) * gs * su / Q \
        - is_lo / is_lo.sum(-1, keepdim=True) * gs * sl / Q
    return gx.reshape(shape), gup, glow


def truncate_small(t, thr=1e-2):
    """|t| < thr  ->  sign(t)*thr  (sign(0) = 0 stays 0)."""
    out = t.clone()
    sel = out.abs() < thr
    out[sel] = out[sel].sign() * thr
    return out


def let_init_scale(act_absmax, weight, alpha):
    a = act_absmax.clamp(min=1e-5)
    w = weight.max(dim=0)[0].clamp(min=1e-5)           # signed column max (quirk Q4)
    return (a.pow(alpha) / w.pow(1 - alpha)).clamp(min=1e-5)


# ----------------------------------------------------------------------------------------------
# precision-mode emulation (NOT part of the reference algorithm)
# ----------------------------------------------------------------------------------------------
class _RoundSTE(torch.autograd.Function):
    """Value and gradient rounded to `dtype` and back: stands for a tensor that is MATERIALISED in that dtype between
    two kernels (the product path's bf16 production mode), while all arithmetic stays fp32 as in the kernels."""

    @staticmethod
    def forward(ctx, t, dtype):
        ctx.dtype = dtype
        return t.to(dtype).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class _GradRoundSTE(torch.autograd.Function):
    """Identity forward, gradient rounded to `dtype`: a tensor whose GRADIENT is materialised in that dtype."""

    @staticmethod
    def forward(ctx, t, dtype):
        ctx.dtype = dtype
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class _FusedAttnModel(torch.autograd.Function):
    """Storage model of the product path's fused causal attention (omniquant_amd/csrc/oq_attn.hip), fp32 arithmetic:
    forward  p~ = exp(s - rowmax), l = sum p~ (fp32); the operand of the P @ v product is p~ rounded to `dtype`
             (un-normalised, as the kernel's online softmax feeds its MFMA), O = (p~_r @ v) / l, stored in `dtype`;
    backward as the kernel does it: P = softmax (fp32, recomputed), dV = P_r^T dO with P rounded, dP = dO v^T,
             D = rowsum(dO * O_r) with the STORED output, dS = scale * P * (dP - D) rounded to `dtype` (the kernel keeps
             dS^T in bf16 for both dK and the dQ GEMM), dQ = dS k, dK = dS^T q.
    The mathematics is that of models/int_llama_layer.py:143-163; only the rounding points are the kernel's.
    grid: the kernels' integer-grid mode (oq_attn_*_grid) -- the forward rounds nothing (q, k, v arrive as exact grid
    coordinates x scale, the probabilities meet v as a bf16 high + low pair = 16 mantissa bits, the output is handed on in
    fp32); the backward keeps its rounding points (dO, P for dV, dS) except that D uses the un-rounded output."""

    @staticmethod
    def forward(ctx, q, k, v, mask, scale, dtype, grid=False):
        s = torch.matmul(q, k.transpose(2, 3)) * scale
        if mask is not None:
            s = torch.max(s + mask, torch.tensor(torch.finfo(s.dtype).min))
        pt = torch.exp(s - s.amax(dim=-1, keepdim=True))
        l = pt.sum(dim=-1, keepdim=True)
        o = torch.matmul(pt if grid else pt.to(dtype).to(pt.dtype), v) / l
        orr = o.to(dtype).to(o.dtype)
        p = pt / l
        ctx.save_for_backward(q, k, v, p, o if grid else orr)       # (grid: D is taken from the un-rounded output)
        ctx.scale, ctx.dtype = scale, dtype
        return o if grid else orr

    @staticmethod
    def backward(ctx, go):
        q, k, v, p, orr = ctx.saved_tensors
        dt = ctx.dtype
        go = go.to(dt).to(go.dtype)
        dv = torch.matmul(p.to(dt).to(p.dtype).transpose(2, 3), go)
        dp = torch.matmul(go, v.transpose(2, 3))
        d = (go * orr).sum(-1, keepdim=True)
        ds = (p * (dp - d) * ctx.scale).to(dt).to(p.dtype)
        dq = torch.matmul(ds, k)
        dk = torch.matmul(ds.transpose(2, 3), q)
        return dq, dk, dv, None, None, None, None


class _IntFpropLinear(torch.autograd.Function):
    """Storage model of a Linear on the product path's INTEGER fprop (oq_gemm_i8): the forward product is exact -- the two
    operands are contracted as integer codes, never rounded -- while dgrad / wgrad run on the operands as they are stored
    for the bf16 MFMA (rounded to `dtype`).  Arithmetic fp32; the mathematics is quantize/int_linear.py:62."""

    @staticmethod
    def forward(ctx, x, w, dtype):
        ctx.save_for_backward(x.to(dtype).to(x.dtype), w.to(dtype).to(w.dtype))
        return F.linear(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        return (g2 @ w).view_as(x), g2.t() @ x2, None


def _rounder(dtype):
    """None -> identity (the reference's pure-fp32 CPU arithmetic, the pinned mode)."""
    if dtype is None:
        return lambda t: t
    return lambda t: _RoundSTE.apply(t, dtype)


# ----------------------------------------------------------------------------------------------
# block description
# ----------------------------------------------------------------------------------------------
def act_stats(samples):
    """LET-init activation statistics of one linear input (generate_act_scale_shift.py:29-37,63-72).
    samples: iterable of [..., K] tensors, one per calibration sample, IN ORDER.  Returns (act_scale[K], act_shift[K]):
    running max of max|x| per channel; (max+min)/2 of the first sample, then 0.99*prev + 0.01*(max+min)/2."""
    scale = shift = None
    for x in samples:
        t = x.reshape(-1, x.shape[-1]).detach()
        amax = t.abs().max(dim=0)[0].float()
        mid = (t.max(dim=0)[0].float() + t.min(dim=0)[0].float()) / 2
        scale = amax if scale is None else torch.max(scale, amax)
        shift = mid if shift is None else 0.99 * shift + 0.01 * mid
    return scale, shift


def autogptq_pack(W, scales, zeros, bits, group_size=None):
    """numpy restatement of AutoGPTQ qlinear_cuda.QuantLinear.pack (v0.4.x) as called at quantize/omniquant.py:264-265.
    PARITY UNPINNED: the library is not vendored in the reference.  W [out,in] fake-quant weight, scales/zeros
    [out, groups].  Returns (qweight [in/32*bits, out] int32, qzeros [groups, out/32*bits] int32)."""
    import numpy as np
    out, inn = W.shape
    group = group_size or inn
    g_idx = np.arange(inn) // group
    s = scales.reshape(out, -1).t().contiguous().float()          # [groups, out]
    z = zeros.reshape(out, -1).t().contiguous().float()
    sz = z * s
    inw = torch.round((W.float().t() + sz[g_idx]) / s[g_idx]).to(torch.int32).numpy().astype(np.uint32)     # [in, out]

    def pack_rows(iw):
        q = np.zeros((iw.shape[0] // 32 * bits, iw.shape[1]), dtype=np.uint32)
        i = row = 0
        while row < q.shape[0]:
            if bits in (2, 4, 8):
                for j in range(i, i + 32 // bits):
                    q[row] |= iw[j] << np.uint32(bits * (j - i))
                i += 32 // bits
                row += 1
            else:
                for j in range(i, i + 10):
                    q[row] |= iw[j] << np.uint32(3 * (j - i))
                i += 10
                q[row] |= iw[i] << np.uint32(30)
                row += 1
                q[row] |= (iw[i] >> np.uint32(2)) & np.uint32(1)
                i += 1
                for j in range(i, i + 10):
                    q[row] |= iw[j] << np.uint32(3 * (j - i) + 1)
                i += 10
                q[row] |= iw[i] << np.uint32(31)
                row += 1
                q[row] |= (iw[i] >> np.uint32(1)) & np.uint32(3)
                i += 1
                for j in range(i, i + 10):
                    q[row] |= iw[j] << np.uint32(3 * (j - i) + 2)
                i += 10
                row += 1
        return q

    qweight = pack_rows(inw)
    zi = (z - 1).numpy().astype(np.uint32)                     # [groups, out]
    qzeros = pack_rows(np.ascontiguousarray(zi.T)).T           # pack along the output channels
    return torch.from_numpy(qweight.astype(np.int32)), torch.from_numpy(np.ascontiguousarray(qzeros).astype(np.int32))


class QuantSpec:
    def __init__(self, wbits=4, abits=16, group_size=None, lwc=True, let=False, symmetric=False):
        self.wbits, self.abits, self.group_size = wbits, abits, group_size
        self.lwc, self.let, self.symmetric = lwc, let, symmetric


LLAMA_LINEARS = ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                 "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"]
# registration order of the reference modules (k, v, q, o; gate, down, up) decides parameter order
LLAMA_MODULE_ORDER = ["self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj", "self_attn.o_proj",
                      "mlp.gate_proj", "mlp.down_proj", "mlp.up_proj"]
OPT_MODULE_ORDER = ["self_attn.k_proj", "self_attn.v_proj", "self_attn.q_proj", "self_attn.out_proj",
                    "fc1", "fc2"]


def _rope_tables(hd, n_pos, base=10000.0):
    inv = 1.0 / (base ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.outer(torch.arange(n_pos, dtype=torch.float32), inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(t):
    h = t.shape[-1] // 2
    return torch.cat((-t[..., h:], t[..., :h]), dim=-1)


class Block:
    """One decoder block (family 'llama' or 'opt') with LWC / LET learnables.

    weights : dict name -> fp32 tensor (HF parameter names of the decoder layer)
    params  : OrderedDict name -> leaf tensor(requires_grad) using the reference's state-dict keys
    """

    def __init__(self, family, cfg, weights, spec, max_pos=2048):
        self.family, self.cfg, self.spec = family, dict(cfg), spec
        self.w = {k: v.clone().float() for k, v in weights.items()}
        self.H = cfg["hidden_size"]
        self.nh = cfg["num_attention_heads"]
        self.hd = self.H // self.nh
        self.nkv = cfg.get("num_key_value_heads", self.nh)
        self.eps = cfg.get("rms_norm_eps", 1e-6) if family == "llama" else 1e-5
        self.order = LLAMA_MODULE_ORDER if family == "llama" else OPT_MODULE_ORDER
        self.params = OrderedDict()
        self.folded = False
        self.qparams = {}
        if family == "llama":
            self.cos, self.sin = _rope_tables(self.hd, max_pos)
        if spec.lwc:
            for n in self.order:
                W = self.w[n + ".weight"]
                rows = W.shape[0] * (math.ceil(W.shape[1] / spec.group_size) if spec.group_size else 1)
                for side in ("upbound_factor", "lowbound_factor"):
                    self.params[f"{n}.weight_quantizer.{side}"] = torch.full((rows, 1), 4.0, requires_grad=True)

    # ---- naming helpers --------------------------------------------------------------------
    @property
    def names(self):
        f = self.family == "llama"
        return dict(q="self_attn.q_proj", k="self_attn.k_proj", v="self_attn.v_proj",
                    o="self_attn.o_proj" if f else "self_attn.out_proj",
                    ln1="input_layernorm" if f else "self_attn_layer_norm",
                    ln2="post_attention_layernorm" if f else "final_layer_norm",
                    fc1=["mlp.up_proj", "mlp.gate_proj"] if f else ["fc1"],
                    last="mlp.down_proj" if f else "fc2")

    def register_let(self, act_scales, act_shifts, alpha, layer_idx, prefix):
        """quantize/omniquant.py:182-197.  Parameter order follows the reference: LWC params are
        module parameters (registered at construction); LET params are appended afterwards."""
        nm = self.names
        self.params["qkt_smooth_scale"] = torch.ones(self.w[nm["q"] + ".weight"].shape[0], requires_grad=True)
        # named_modules order: k, v, q(->qkv), o(->out), [gate, down,] up/fc1(->fc1)
        for mod, key in ((nm["q"], "qkv"), (nm["o"], "out"), (nm["fc1"][0], "fc1")):
            scale = let_init_scale(act_scales[f"{prefix}.{layer_idx}.{mod}"].float(), self.w[mod + ".weight"], alpha)
            if self.family == "llama":
                shift = torch.zeros_like(scale)
            else:
                shift = act_shifts[f"{prefix}.{layer_idx}.{mod}"].float().clone()
            self.params[f"{key}_smooth_shift"] = shift.requires_grad_(True)
            self.params[f"{key}_smooth_scale"] = scale.detach().clone().requires_grad_(True)

    def let_params(self):
        return [p for n, p in self.params.items() if "smooth" in n]

    def lwc_params(self):
        return [p for n, p in self.params.items() if "bound_factor" in n]

    # ---- temporaries (LET re-parameterisation + weight fake-quant) ---------------------------
    def _bias(self, name):
        return self.w.get(name + ".bias")

    def _wq(self, name, W):
        s = self.spec
        up = self.params.get(f"{name}.weight_quantizer.upbound_factor")
        low = self.params.get(f"{name}.weight_quantizer.lowbound_factor")
        y, sc, zp = fake_quant(W, s.wbits, s.group_size, up, low, s.symmetric, return_qparams=True)
        self.qparams[name] = (sc, zp)
        return y

    def temporaries(self, store_dtype=None, int_fprop=False):
        """Returns dict of temp tensors: '<linear>.weight', '<linear>.bias', '<ln>.weight', '<ln>.bias'.
        store_dtype (precision-mode emulation only): the fake-quantised weights are rounded to that dtype, as the
        product path stores them for its bf16 MFMA GEMMs; biases and norm parameters stay fp32 there too.
        int_fprop (precision-mode emulation only): the weights stay unrounded here -- forward(int_fprop=True) contracts
        them exactly and rounds them for the backward products only (_IntFpropLinear)."""
        nm, P, W = self.names, self.params, self.w
        rw = _rounder(None if int_fprop else store_dtype)
        t = {}
        if self.spec.let:
            with torch.no_grad():
                for n, p in P.items():
                    if "smooth_scale" in n:
                        p.data = truncate_small(p.data)
            for ln, fcs, key in ((nm["ln1"], [nm["q"], nm["k"], nm["v"]], "qkv"), (nm["ln2"], nm["fc1"], "fc1")):
                sc, sh = P[f"{key}_smooth_scale"], P[f"{key}_smooth_shift"]
                lb = W.get(ln + ".bias")
                t[ln + ".bias"] = (lb - sh) / sc if lb is not None else (-1 * sh) / sc
                t[ln + ".weight"] = W[ln + ".weight"] / sc
                for fc in fcs:
                    b = self._bias(fc)
                    t[fc + ".bias"] = b + W[fc + ".weight"] @ sh if b is not None else W[fc + ".weight"] @ sh
                    t[fc + ".weight"] = W[fc + ".weight"] * sc.view(1, -1)
            sc, sh = P["out_smooth_scale"], P["out_smooth_shift"]
            v, o = nm["v"], nm["o"]
            t[v + ".bias"] = (t[v + ".bias"] - sh) / sc.view(-1)
            t[v + ".weight"] = t[v + ".weight"] / sc.view(-1, 1)
            ob = self._bias(o)
            t[o + ".bias"] = ob + W[o + ".weight"] @ sh if ob is not None else W[o + ".weight"] @ sh
            t[o + ".weight"] = W[o + ".weight"] * sc.view(1, -1)
            s = P["qkt_smooth_scale"]
            q, k = nm["q"], nm["k"]
            t[q + ".weight"] = t[q + ".weight"] / s.view(-1, 1)
            t[q + ".bias"] = t[q + ".bias"] / s.view(-1)
            t[k + ".weight"] = t[k + ".weight"] * s.view(-1, 1)
            t[k + ".bias"] = t[k + ".bias"] * s.view(-1)
            t[nm["last"] + ".weight"] = W[nm["last"] + ".weight"]
        else:
            for n in self.order:
                t[n + ".weight"] = W[n + ".weight"]
        for n in self.order:
            t[n + ".weight"] = rw(self._wq(n, t[n + ".weight"]))
            if n + ".bias" not in t:
                t[n + ".bias"] = self._bias(n)
        return t

    # ---- forward -----------------------------------------------------------------------------
    def _aq(self, x, on):
        return fake_quant(x, self.spec.abits) if on else x

    def _norm(self, x, name, t):
        w = t[name + ".weight"] if t and name + ".weight" in t else self.w[name + ".weight"]
        b = t[name + ".bias"] if t and name + ".bias" in t else self.w.get(name + ".bias")
        if self.family == "llama":
            var = x.float().pow(2).mean(-1, keepdim=True)
            xh = x * torch.rsqrt(var + self.eps)
            return (w * xh + b).to(x.dtype) if b is not None else (w * xh).to(x.dtype)
        return F.layer_norm(x, (self.H,), w, b, eps=self.eps)

    def _lin(self, x, name, t, act_quant, rnd=None, residual=None, round_out=True, int_dtype=None):
        w = t[name + ".weight"] if t else self.w[name + ".weight"]
        b = t[name + ".bias"] if t else self.w.get(name + ".bias")
        xin = self._aq(x, act_quant)
        if int_dtype is not None and act_quant and self.spec.abits < 16:
            # integer fprop: exact forward product of the two quantised operands, bf16 operands in the backward products
            y = _IntFpropLinear.apply(xin, w, int_dtype)
            if b is not None:
                y = y + b
        else:
            if rnd is not None and act_quant and self.spec.abits < 16:
                xin = rnd(xin)                              # the quantised activation is stored before the GEMM
            y = F.linear(xin, w, b)
        if residual is not None:
            y = residual + y                                # (the product path adds the residual in the GEMM's store)
        if rnd is not None and round_out:
            return rnd(y)
        if int_dtype is not None:
            return _GradRoundSTE.apply(y, int_dtype)        # value kept in fp32 for a fused quantiser, its gradient is stored in bf16
        return y

    def forward(self, x, mask=None, position_ids=None, temps=None, act_quant=True, act_dtype=None, int_fprop=False, wide=False):
        """x [bs,T,H].  temps=None -> raw (or folded) weights; act_quant toggles every activation quantizer.
        act_dtype (precision-mode emulation only, default None = the reference's fp32 arithmetic): every activation the
        product path materialises between two kernels is rounded to that dtype (value and gradient), arithmetic stays
        fp32 -- the CPU model of the bf16 production mode.
        int_fprop (with act_dtype, weight-activation configurations on grids of at most 8 bits without weight groups): the
        product path's integer fprop -- every Linear's forward product is exact (_IntFpropLinear; `temps` must come from
        temporaries(int_fprop=True)), and the q | k | v projection output, which feeds the fused RoPE -> head quantisers, stays
        fp32 (head_dim 128).
        wide (llama, head_dim 128): the product path's un-rounded side channels in front of every remaining 4-bit rounding
        decision -- the fused attention on the head quantisers' integer grid (q, k, v never rounded, output handed to the
        o_proj input quantiser in fp32) and, with int_fprop, the hidden state after attention (o_proj output + residual) kept
        in fp32 next to its bf16 copy for the second norm -> quantiser and the down_proj residual add; the GRADIENTS of those
        tensors are still stored in `act_dtype`."""
        nm = self.names
        bs, T, H = x.shape
        R_ = _rounder(act_dtype)
        rnd = R_ if act_dtype is not None else None
        aq4 = act_quant and self.spec.abits < 16
        idt = act_dtype if (int_fprop and act_dtype is not None and aq4 and self.spec.abits <= 8 and self.spec.wbits <= 8
                            and not self.spec.group_size) else None
        # (with activation quantisation on, the product path fuses norm -> input quantiser: the norm output is not stored)
        fused_nq = act_dtype is not None and aq4 and 512 <= H <= 8192 and H % 8 == 0
        h = self._norm(x, nm["ln1"], temps)
        if not fused_nq:
            h = R_(h)
        if self.family == "llama":
            # with head-wise activation quantisation on (head_dim 128), the product path rotates and quantises the stored
            # projection output in one kernel: the rotated tensor is never stored
            keep = act_dtype is not None and aq4 and self.hd == 128
            pre_f32 = idt is not None and keep          # integer path: the projection output reaches the fused quantiser in fp32
            wd = bool(wide) and keep and mask is not None         # grid attention (the causal fused kernels) + its fp32 output
            wd_h = bool(wide) and idt is not None                 # fp32 hidden states: written by the integer fprop's epilogue
            q = self._lin(h, nm["q"], temps, act_quant, rnd, round_out=not pre_f32, int_dtype=idt).view(bs, T, self.nh, self.hd).transpose(1, 2)
            k = self._lin(h, nm["k"], temps, act_quant, rnd, round_out=not pre_f32, int_dtype=idt).view(bs, T, self.nkv, self.hd).transpose(1, 2)
            v = self._lin(h, nm["v"], temps, act_quant, rnd, round_out=not pre_f32, int_dtype=idt).view(bs, T, self.nkv, self.hd).transpose(1, 2)
            cos = self.cos[:T][position_ids].unsqueeze(1)
            sin = self.sin[:T][position_ids].unsqueeze(1)
            q = q * cos + _rot_half(q) * sin
            k = k * cos + _rot_half(k) * sin
            if not keep:
                q, k = R_(q), R_(k)
            rep = self.nh // self.nkv
            if rep > 1:
                k = k[:, :, None].expand(bs, self.nkv, rep, T, self.hd).reshape(bs, self.nh, T, self.hd)
                v = v[:, :, None].expand(bs, self.nkv, rep, T, self.hd).reshape(bs, self.nh, T, self.hd)
            q, k = self._aq(q, act_quant), self._aq(k, act_quant)
            GR_ = (lambda t_: _GradRoundSTE.apply(t_, act_dtype)) if wd else None
            if aq4:
                q, k = (GR_(q), GR_(k)) if wd else (R_(q), R_(k))
            v = self._aq(v, act_quant)
            if aq4:
                v = GR_(v) if wd else R_(v)
            if act_dtype is not None and self.hd == 128:
                # the product path's fused kernels (bf16, head_dim 128, causal): their own rounding points
                o = _FusedAttnModel.apply(q, k, v, mask, 1.0 / math.sqrt(self.hd), act_dtype, wd)
                o = o.transpose(1, 2).reshape(bs, T, H)
                if wd:
                    o = GR_(o)
            else:
                att = R_(torch.matmul(q, k.transpose(2, 3))) / math.sqrt(self.hd) if act_dtype is not None else \
                    torch.matmul(q, k.transpose(2, 3)) / math.sqrt(self.hd)
                if mask is not None:
                    att = att + mask
                    att = torch.max(att, torch.tensor(torch.finfo(att.dtype).min))
                att = F.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
                o = R_(torch.matmul(R_(att), v).transpose(1, 2).reshape(bs, T, H))
            # (the first hidden state's side channel is read by the fused norm -> quantiser kernel; the plain norm reads bf16)
            h = self._lin(o, nm["o"], temps, act_quant, rnd, residual=x, int_dtype=idt, round_out=not (wd_h and fused_nq))
            h2 = self._norm(h, nm["ln2"], temps)
            if not fused_nq:
                h2 = R_(h2)
            # (gate | up are stored in bf16 on the integer path too: keeping them fp32 was measured to change nothing)
            gate = self._lin(h2, "mlp.gate_proj", temps, act_quant, rnd, int_dtype=idt)
            up = self._lin(h2, "mlp.up_proj", temps, act_quant, rnd, int_dtype=idt)
            act = F.silu(gate) * up
            if not aq4:
                act = R_(act)          # stored before the GEMM; with activation quantisation on, the product path fuses
                #                        silu*up into the down_proj input quantiser and the product stays fp32
            return self._lin(act, "mlp.down_proj", temps, act_quant, rnd, residual=h, int_dtype=idt)     # (the block output feeds no rounding decision: bf16, as stored)
        # ---- OPT: q/k/v are quantised per token over the full hidden dim before the head split
        scaling = self.hd ** -0.5
        q = self._aq(R_(self._lin(h, nm["q"], temps, act_quant, rnd) * scaling), act_quant)
        k = self._aq(self._lin(h, nm["k"], temps, act_quant, rnd), act_quant)
        v = self._aq(self._lin(h, nm["v"], temps, act_quant, rnd), act_quant)
        if aq4:
            q, k, v = R_(q), R_(k), R_(v)

        def split(t_):
            return t_.view(bs, T, self.nh, self.hd).transpose(1, 2).contiguous().view(bs * self.nh, T, self.hd)
        q, k, v = split(q), split(k), split(v)
        att = R_(torch.bmm(q, k.transpose(1, 2)))
        if mask is not None:
            att = att.view(bs, self.nh, T, T) + mask
            att = torch.max(att, torch.tensor(torch.finfo(att.dtype).min)).view(bs * self.nh, T, T)
        att = R_(F.softmax(att, dim=-1))
        o = R_(torch.bmm(att, v).view(bs, self.nh, T, self.hd).transpose(1, 2).reshape(bs, T, H))
        h = self._lin(o, nm["o"], temps, act_quant, rnd, residual=x).reshape(-1, H)
        h2 = R_(self._norm(h, nm["ln2"], temps))
        f = R_(F.relu(self._lin(h2, "fc1", temps, act_quant, rnd)))
        f = self._lin(f, "fc2", temps, act_quant, rnd, residual=h)
        return f.view(bs, T, H)

    # ---- final fold --------------------------------------------------------------------------
    @torch.no_grad()
    def fold(self):
        """smooth_and_quant_inplace: after this, self.w holds folded + fake-quantised weights."""
        nm, P, W = self.names, self.params, self.w
        if self.spec.let:
            for n, p in P.items():
                if "smooth_scale" in n:
                    p.data = truncate_small(p.data)
            for ln, fcs, key in ((nm["ln1"], [nm["q"], nm["k"], nm["v"]], "qkv"), (nm["ln2"], nm["fc1"], "fc1")):
                sc, sh = P[f"{key}_smooth_scale"], P[f"{key}_smooth_shift"]
                lb = W.get(ln + ".bias")
                W[ln + ".bias"] = (lb - sh) / sc if lb is not None else (-1 * sh) / sc
                W[ln + ".weight"] = W[ln + ".weight"] / sc
                for fc in fcs:
                    b = self._bias(fc)
                    W[fc + ".bias"] = b + W[fc + ".weight"] @ sh if b is not None else W[fc + ".weight"] @ sh
                    W[fc + ".weight"] = W[fc + ".weight"] * sc.view(1, -1)
            sc, sh = P["out_smooth_scale"], P["out_smooth_shift"]
            v, o = nm["v"], nm["o"]
            W[v + ".bias"] = (W[v + ".bias"] - sh) / sc.view(-1)
            W[v + ".weight"] = W[v + ".weight"] / sc.view(-1, 1)
            ob = self._bias(o)
            W[o + ".bias"] = ob + W[o + ".weight"] @ sh if ob is not None else W[o + ".weight"] @ sh
            W[o + ".weight"] = W[o + ".weight"] * sc.view(1, -1)
            s = P["qkt_smooth_scale"]
            q, k = nm["q"], nm["k"]
            W[q + ".weight"] = W[q + ".weight"] / s.view(-1, 1)
            W[q + ".bias"] = W[q + ".bias"] / s.view(-1)
            W[k + ".weight"] = W[k + ".weight"] * s.view(-1, 1)
            W[k + ".bias"] = W[k + ".bias"] * s.view(-1)
        for n in self.order:
            W[n + ".weight"] = self._wq(n, W[n + ".weight"]).detach()
        self.folded = True

    def omni_state_dict(self):
        """fp16 learnables, reference key order (quirk Q10)."""
        return OrderedDict((n, p.detach().half()) for n, p in self.params.items())


# ----------------------------------------------------------------------------------------------
# optimiser (torch.optim.AdamW, amsgrad=False, maximize=False) and grad norm
# ----------------------------------------------------------------------------------------------
class AdamW:
    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.groups, self.b1, self.b2, self.eps, self.wd = groups, betas[0], betas[1], eps, weight_decay
        self.t = 0
        self.m = [[torch.zeros_like(p) for p in g["params"]] for g in groups]
        self.v = [[torch.zeros_like(p) for p in g["params"]] for g in groups]

    @torch.no_grad()
    def step(self):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for gi, g in enumerate(self.groups):
            lr = g["lr"]
            for pi, p in enumerate(g["params"]):
                if p.grad is None:
                    continue
                m, v = self.m[gi][pi], self.v[gi][pi]
                p.mul_(1 - lr * self.wd)
                m.lerp_(p.grad, 1 - self.b1)
                v.mul_(self.b2).addcmul_(p.grad, p.grad, value=1 - self.b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(m, denom, value=-lr / bc1)

    def zero_grad(self):
        for g in self.groups:
            for p in g["params"]:
                p.grad = None


def grad_norm(params):
    gs = [p.grad for p in params if p.grad is not None]
    return torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in gs]), 2.0)


# ----------------------------------------------------------------------------------------------
# engine
# ----------------------------------------------------------------------------------------------
def train_step(block, opt, x, target, mask, pos, target2=None):
    """One sample-step: temporaries -> forward -> MSE -> backward -> grad-norm -> AdamW."""
    temps = block.temporaries()
    out = block.forward(x, mask, pos, temps=temps, act_quant=True)
    loss = F.mse_loss(target, out)
    if target2 is not None:
        loss = loss + F.mse_loss(target2, out)
    opt.zero_grad()
    loss.backward()
    norm = grad_norm(list(block.params.values()))
    opt.step()
    return float(loss.detach()), float(norm)


def calibrate(family, cfg, layer_weights, spec, inps, mask, pos, act_scales=None, act_shifts=None,
              epochs=2, let_lr=5e-3, lwc_lr=1e-2, wd=0.0, alpha=0.5, aug_loss=False, prefix="model.layers", batch_size=1):
    """Sequential block-wise calibration.  batch_size samples per step with the mask repeated per sample
    (quantize/omniquant.py:139-141,214-219).  Returns dict with per-step losses/norms, the
    fp16 omni_state_dict per layer, folded weights, qparams and the propagated activations."""
    quant_inps = inps.clone()
    fp_inps = inps.clone()
    fp_inps_2 = inps.clone() if aug_loss else None
    n = inps.shape[0]
    res = dict(losses=[], norms=[], omni=[], folded=[], qparams=[], fp_out=[], quant_out=[], trained=[], blocks=[])
    for i, lw in enumerate(layer_weights):
        blk = Block(family, cfg, lw, spec)
        with torch.no_grad():
            for j in range(n):
                fp_inps[j] = blk.forward(fp_inps[j][None], mask, pos, temps=None, act_quant=False)[0]
                if aug_loss:
                    fp_inps_2[j] = blk.forward(quant_inps[j][None], mask, pos, temps=None, act_quant=False)[0]
        res["fp_out"].append(fp_inps.clone())
        if spec.let:
            blk.register_let(act_scales, act_shifts, alpha, i, prefix)
        opt = AdamW([{"params": blk.let_params(), "lr": let_lr}, {"params": blk.lwc_params(), "lr": lwc_lr}],
                    weight_decay=wd)
        bs = batch_size
        mask_b = mask.repeat(bs, 1, 1, 1) if (bs > 1 and mask is not None) else mask
        for _ in range(epochs):
            for j in range(0, n // bs * bs, bs):
                l, g = train_step(blk, opt, quant_inps[j:j + bs], fp_inps[j:j + bs], mask_b, pos,
                                  fp_inps_2[j:j + bs] if aug_loss else None)
                res["losses"].append(l)
                res["norms"].append(g)
        res["trained"].append(OrderedDict((k, v.detach().clone()) for k, v in blk.params.items()))
        blk.fold()
        with torch.no_grad():
            for j in range(n):
                quant_inps[j] = blk.forward(quant_inps[j][None], mask, pos, temps=None, act_quant=True)[0]
        res["quant_out"].append(quant_inps.clone())
        res["folded"].append({k: v.clone() for k, v in blk.w.items()})
        res["qparams"].append({k: (s.detach().clone(), z.detach().clone()) for k, (s, z) in blk.qparams.items()})
        res["omni"].append(blk.omni_state_dict())
        res["blocks"].append(blk)
    return res
