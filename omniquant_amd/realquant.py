"""Real-quant packing on the HIP path (`--real_quant`, quantize/omniquant.py:255-278).

The reference hands every folded QuantLinear to AutoGPTQ's `qlinear_cuda.QuantLinear.pack` (third-party, not vendored:
PARITY UNPINNED, see csrc/oq_pack.hip).  `pack_linear` produces the same buffers (`qweight`, `qzeros`, `scales`, `g_idx`,
`bias`, AutoGPTQ names and shapes) with one HIP kernel; `PackedLinear` is a minimal container with a `dequantize()`
that inverts the packing (used by the tests' round trip; inference kernels are out of scope)."""
import torch
import torch.nn as nn

from . import _capi as C


def pack_linear(weight, scales, zeros, bits, group_size=None, bias=None):
    """weight [out,in] = folded fake-quant weight; scales/zeros [out*groups,1] or [out,groups] (the `scales`/`zeros`
    buffers `register_scales_and_zeros` creates).  Returns a dict of AutoGPTQ-style buffers."""
    if not weight.is_cuda:
        raise C.OQError("pack_linear: GPU tensors expected (no CPU fallback)")
    out, inn = weight.shape
    group = group_size or inn
    ng = inn // group
    w = weight.detach().contiguous()
    s = scales.detach().float().reshape(out, ng).contiguous()
    z = zeros.detach().float().reshape(out, ng).contiguous()
    qweight = torch.empty((inn // 32 * bits, out), dtype=torch.int32, device=w.device)
    qzeros = torch.empty((ng, out // 32 * bits), dtype=torch.int32, device=w.device)
    C.call("oq_pack_weights", C.ptr(w), C.dt(w), out, inn, group, bits, C.fptr(s), C.fptr(z), C.ptr(qweight), C.ptr(qzeros),
           C.stream())
    g_idx = (torch.arange(inn, device=w.device, dtype=torch.int32) // group)
    return dict(qweight=qweight, qzeros=qzeros, scales=s.t().contiguous().half(), g_idx=g_idx,
                bias=None if bias is None else bias.detach().half())


def unpack(q, bits, n):
    """Inverse of the AutoGPTQ word packing along dim 0: q [n/32*bits, m] int32 -> codes [n, m] int64."""
    qu = q.to(torch.int64) & 0xFFFFFFFF
    m = qu.shape[1]
    if bits != 3:
        per = 32 // bits
        sh = torch.arange(per, device=q.device, dtype=torch.int64) * bits
        codes = (qu[:, None, :] >> sh[None, :, None]) & ((1 << bits) - 1)
        return codes.reshape(-1, m)[:n]
    w3 = qu.reshape(-1, 3, m)
    a, b, c = w3[:, 0], w3[:, 1], w3[:, 2]
    cols = []
    for j in range(10):
        cols.append((a >> (3 * j)) & 7)
    cols.append(((a >> 30) & 3) | ((b & 1) << 2))
    for j in range(10):
        cols.append((b >> (3 * j + 1)) & 7)
    cols.append(((b >> 31) & 1) | ((c & 3) << 1))
    for j in range(10):
        cols.append((c >> (3 * j + 2)) & 7)
    return torch.stack(cols, dim=1).reshape(-1, m)[:n]


class PackedLinear(nn.Module):
    def __init__(self, bits, group_size, in_features, out_features, packed):
        super().__init__()
        self.bits, self.group_size, self.infeatures, self.outfeatures = bits, group_size or in_features, in_features, out_features
        for k, v in packed.items():
            if v is not None:
                self.register_buffer(k, v)
            else:
                setattr(self, k, None)

    def dequantize(self):
        """fp16 weight [out, in] = scales * (q - (qzeros + 1))  (AutoGPTQ's dequantisation)."""
        q = unpack(self.qweight, self.bits, self.infeatures)                              # [in, out]
        z = unpack(self.qzeros.t().contiguous(), self.bits, self.outfeatures).t() + 1      # [groups, out]
        g = self.g_idx.long()
        w = self.scales.float()[g] * (q - z[g]).float()
        return w.t().contiguous().half()


def pack_block(qlayer, wbits):
    """Replace every folded QuantLinear of a calibrated block by a PackedLinear (quantize/omniquant.py:255-278)."""
    from .linear import QuantLinear
    for name, module in list(qlayer.named_modules()):
        if not isinstance(module, QuantLinear):
            continue
        wq = module.weight_quantizer
        packed = pack_linear(module.weight.float(), wq.scales, wq.zeros, wbits, wq.group_size, module.bias)
        pl = PackedLinear(wbits, wq.group_size, module.in_features, module.out_features, packed)
        parent = qlayer
        parts = name.split(".")
        for p_ in parts[:-1]:
            parent = getattr(parent, p_)
        setattr(parent, parts[-1], pl)
    return qlayer
