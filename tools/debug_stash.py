import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import omniquant_amd.ops as ops
STASH = []
_fq_b = ops.FakeQuantFn.backward
def fq_b(ctx, gy, gws):
    out = _fq_b(ctx, gy, gws)
    STASH.append(("fq rows=%d cols=%d" % (ctx.cfg[0], ctx.cfg[1]), dict(gy=gy, gws=gws, g_cm=out[1], g_rd=out[2], g_rm=out[3], g_sh=out[4], g_up=out[5])))
    return out
ops.FakeQuantFn.backward = staticmethod(fq_b)
_n_b = ops.NormFn.backward
def n_b(ctx, gy):
    out = _n_b(ctx, gy)
    STASH.append(("norm", dict(gy=gy, gx=out[0], gw=out[1], gb=out[2])))
    return out
ops.NormFn.backward = staticmethod(n_b)
_l_b = ops.LinearFn.backward
def l_b(ctx, gy):
    out = _l_b(ctx, gy)
    STASH.append(("linear N=%d" % ctx.saved_tensors[1].shape[0], dict(gy=gy, gx=out[0], gw=out[1], gb=out[2])))
    return out
ops.LinearFn.backward = staticmethod(l_b)
import omniquant_amd.calibrate as cal
_run = cal.StepRunner.run
def run(self, *a, **k):
    _run(self, *a, **k)
    torch.cuda.synchronize()
    if self.use_graph:
        print("== after replay", self.steps)
        # only the entries of the captured step (last len/3) are live
        n = len(STASH) // 3
        for name, d in STASH[-n:]:
            for k2, v in d.items():
                if v is None: continue
                m = v.float().abs().max().item()
                if not (m < 1e4): print("   BAD", name, k2, m, tuple(v.shape), v.dtype)
cal.StepRunner.run = run
exec(open(os.path.join(R, "tools", "debug_graph.py")).read())
