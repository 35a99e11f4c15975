"""Learnable Equivalent Transformation (LET) algebra on the HIP path.

Reference: models/transformation.py.  The per-step (`*_temporary`) transforms are NOT materialised as separate
elementwise passes: `fused_temporaries()` feeds the scale/shift vectors straight into the fake-quant kernel
(oq_fakequant_fwd: x = ((W*col_mul)/row_div)*row_mul, wshift = W@shift), which reads each weight once.  Only
[hidden]-sized vector arithmetic (biases, norm weights) is done with torch ops.  The `*_inplace` fold uses the
same kernel with float32 output.
"""
import torch

from . import _capi as C


def truncate_number(number, threshold=1e-2):
    """|x| < thr -> sign(x)*thr, identity gradient (models/transformation.py:5-20).  In place on the HIP path
    (the reference rebinds .data to the truncated copy, which is equivalent)."""
    data = number.data if isinstance(number, torch.nn.Parameter) else number
    if not data.is_contiguous():
        raise C.OQError("truncate_number: parameter must be contiguous")
    C.call("oq_truncate", C.fptr(data), data.numel(), float(threshold), C.stream())
    return number


def _f(t):
    return None if t is None else t.float()


class LetSpec:
    """Per-linear description of the LET transform: W' = ((W*col_mul)/row_div)*row_mul, b' from wshift."""

    def __init__(self, col_mul=None, row_div=None, row_mul=None, shift=None, bias_fn=None):
        self.col_mul, self.row_div, self.row_mul, self.shift, self.bias_fn = col_mul, row_div, row_mul, shift, bias_fn


def ln_temporaries(ln, scales, shifts):
    """temp weight/bias of the norm in front of a smoothed fc group (models/transformation.py:24-33)."""
    if hasattr(ln, "bias") and ln.bias is not None:
        tb = (_f(ln.bias) - shifts) / scales
    else:
        tb = (-1 * shifts) / scales
    tw = _f(ln.weight) / scales
    return tw, tb


def block_let_specs(names, P, biases):
    """Builds the LetSpec of every linear of a block.

    names: dict with keys q,k,v,o (module objects), fc1 (list of modules), last (module)
    P: object with attributes qkv_smooth_scale/shift, out_smooth_scale/shift, fc1_smooth_scale/shift, qkt_smooth_scale
    Bias algebra restates models/transformation.py:34-69 with ws = W @ shift coming out of the fused kernel.
    """
    qkv_s, qkv_sh = P.qkv_smooth_scale, P.qkv_smooth_shift
    out_s, out_sh = P.out_smooth_scale, P.out_smooth_shift
    fc1_s, fc1_sh = P.fc1_smooth_scale, P.fc1_smooth_shift
    qkt = P.qkt_smooth_scale

    def base(mod):   # bias + W@shift
        b = _f(mod.bias) if mod.bias is not None else None
        return (lambda ws: b + ws) if b is not None else (lambda ws: ws)

    specs = {}
    q, k, v, o = names["q"], names["k"], names["v"], names["o"]
    bq, bk, bv, bo = base(q), base(k), base(v), base(o)
    specs[q] = LetSpec(col_mul=qkv_s, row_div=qkt, shift=qkv_sh, bias_fn=lambda ws: bq(ws) / qkt.view(-1))
    specs[k] = LetSpec(col_mul=qkv_s, row_mul=qkt, shift=qkv_sh, bias_fn=lambda ws: bk(ws) * qkt.view(-1))
    specs[v] = LetSpec(col_mul=qkv_s, row_div=out_s, shift=qkv_sh,
                       bias_fn=lambda ws: (bv(ws) - out_sh) / out_s.view(-1))
    specs[o] = LetSpec(col_mul=out_s, shift=out_sh, bias_fn=bo)
    for fc in names["fc1"]:
        specs[fc] = LetSpec(col_mul=fc1_s, shift=fc1_sh, bias_fn=base(fc))
    specs[names["last"]] = LetSpec()
    return specs
