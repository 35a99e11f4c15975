"""Activation statistics for the LET initialisation on the HIP path.

Reference: generate_act_scale_shift.py:25-94 -- an offline pre-pass that hooks every nn.Linear of the FP model and
records, per input channel, max|x| over all calibration samples (`act_scales`) and a running average of
(max+min)/2 (`act_shifts`).  Here the same numbers fall out of the FP teacher pass the calibration engine runs anyway
(quantize/omniquant.py:165-172): `ActStatCollector` is attached to the QuantLinears whose input LET smooths and the
bank's activations are reduced by `oq_act_stats` as they stream by (one extra read of each activation, no extra
forward).  Keys are the reference's (`model.layers.{i}.self_attn.q_proj`, ...), so the dicts are interchangeable with
the `act_scales/*.pt` / `act_shifts/*.pt` files `main.py:313-315` loads."""
import torch

from . import _capi as C


class ActStatCollector:
    def __init__(self):
        self.scales, self.shifts, self.seen = {}, {}, {}
        self._ws = None

    def update(self, name, x):
        """x [nsamp, T, K] (or [T, K]): fold its samples, in order, into the statistics of `name`."""
        if not x.is_cuda:
            raise C.OQError("ActStatCollector: GPU tensor expected (no CPU fallback)")
        x = x.detach().contiguous()
        if x.dim() == 2:
            x = x[None]
        nsamp, rows, cols = x.shape[0], x.numel() // (x.shape[0] * x.shape[-1]), x.shape[-1]
        if name not in self.scales:
            self.scales[name] = torch.zeros(cols, dtype=torch.float32, device=x.device)
            self.shifts[name] = torch.zeros(cols, dtype=torch.float32, device=x.device)
            self.seen[name] = 0
        need = C.size_call("oq_act_stats_workspace", nsamp, cols)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.float32, device=x.device)
        C.call("oq_act_stats", C.ptr(x), C.dt(x), nsamp, rows, cols, C.fptr(self.scales[name]), C.fptr(self.shifts[name]),
               self.seen[name], C.fptr(self._ws), self._ws.numel(), C.stream())
        self.seen[name] += nsamp

    def attach(self, qlayer, prefix, layer_idx, only=None):
        """Register this collector on the QuantLinears of a block (all of them, or those whose name contains one of
        `only`); returns the list of modules to pass to detach()."""
        from .linear import QuantLinear
        mods = []
        for name, m in qlayer.named_modules():
            if isinstance(m, QuantLinear) and (only is None or any(k in name for k in only)):
                m.__dict__["_stat_sink"] = (self, f"{prefix}.{layer_idx}.{name}")
                mods.append(m)
        return mods

    @staticmethod
    def detach(mods):
        for m in mods:
            m.__dict__.pop("_stat_sink", None)
