"""Poison every torch.empty/empty_like used by the op layer with NaN: an uninitialised read shows up as NaN."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import omniquant_amd.ops as ops
_empty, _empty_like = torch.empty, torch.empty_like
class _T:
    def __getattr__(self, k): return getattr(torch, k)
    @staticmethod
    def empty(*a, **k):
        t = _empty(*a, **k)
        if t.is_floating_point(): t.fill_(float("nan"))
        return t
    @staticmethod
    def empty_like(*a, **k):
        t = _empty_like(*a, **k)
        if t.is_floating_point(): t.fill_(float("nan"))
        return t
ops.torch = _T()
import omniquant_amd.calibrate as cal
cal.torch = _T() if False else cal.torch
sys.argv = [sys.argv[0]] + sys.argv[1:]
exec(open(os.path.join(R, "tools", "debug_graph.py")).read())
