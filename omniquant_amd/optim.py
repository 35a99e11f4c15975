"""Optimiser side of the calibration step on the HIP path.

Reference: utils.py:11-46 (NativeScalerWithGradNormCount, ampscaler_get_grad_norm) and the two-group
torch.optim.AdamW built at quantize/omniquant.py:207-208.

MI355X-native layout: every learnable of a block (LET scales, LET shifts, LWC bounds) is re-homed into ONE
contiguous float32 arena; the nn.Parameters keep their identity, names and shapes but become views, and so do
their .grad tensors.  zero_grad = one memset, grad-norm = one reduction, AdamW = one kernel with the step
counter and the finite-flag on the device (graph-replayable, no host sync, no loss scaling needed for bf16).
"""
import torch

from . import _capi as C


class BlockOptimizer:
    def __init__(self, qlayer, let_lr=5e-3, lwc_lr=1e-2, weight_decay=0.0, use_shift=True,
                 betas=(0.9, 0.999), eps=1e-8):
        named = list(qlayer.named_parameters())
        scales = [(n, p) for n, p in named if "smooth_scale" in n]
        shifts = [(n, p) for n, p in named if "smooth_shift" in n] if use_shift else []
        lwc = [(n, p) for n, p in named if "bound_factor" in n]
        self.named = scales + shifts + lwc
        if not self.named:
            raise ValueError("BlockOptimizer: the block has no learnable LWC/LET parameters")
        dev = self.named[0][1].device
        self.n_scale = sum(p.numel() for _, p in scales)
        self.n_let = self.n_scale + sum(p.numel() for _, p in shifts)
        self.n = self.n_let + sum(p.numel() for _, p in lwc)
        self.flat = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.grad)
        self.exp_avg_sq = torch.zeros_like(self.grad)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(2, dtype=torch.float32, device=dev)
        self._ws = torch.zeros(512, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for _, p in self.named:
                k = p.numel()
                self.flat[off:off + k].copy_(p.data.reshape(-1).float())
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                p.requires_grad_(True)
                off += k
        if self.n_scale:
            qlayer.__dict__["_arena_truncate"] = self.truncate_scales     # block's truncate_number -> one launch
        self.let_lr, self.lwc_lr, self.wd = float(let_lr), float(lwc_lr), float(weight_decay)
        self.betas, self.eps = betas, float(eps)

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()

    def truncate_scales(self, thr=1e-2):
        """truncate_number over all LET scales in one launch (they are the head of the arena)."""
        if self.n_scale:
            C.call("oq_truncate", C.fptr(self.flat), self.n_scale, float(thr), C.stream())

    def grad_norm(self):
        C.call("oq_gradnorm", C.fptr(self.grad), self.n, C.fptr(self.norm), C.fptr(self._ws), C.stream())
        return self.norm[0]

    def step(self):
        """grad-norm + AdamW; the update is skipped on device when a gradient is non-finite."""
        norm = self.grad_norm()
        C.call("oq_adamw", C.fptr(self.flat), C.fptr(self.grad), C.fptr(self.exp_avg), C.fptr(self.exp_avg_sq),
               self.n, self.n_let, self.let_lr, self.lwc_lr, self.betas[0], self.betas[1], self.eps, self.wd,
               C.fptr(self.step_count), C.fptr(self.norm), C.stream())
        return norm


class NativeScalerWithGradNormCount:
    """API-compatible stand-in for utils.py:26-52.  bf16/f32 need no loss scaling: scale is 1; the
    skip-on-inf behaviour of GradScaler.step lives inside oq_adamw."""
    state_dict_key = "amp_scaler"

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True,
                 retain_graph=False):
        loss.backward(create_graph=create_graph, retain_graph=retain_graph)
        if not update_grad:
            return None
        if clip_grad is not None:
            raise NotImplementedError("gradient clipping is not used by the calibration loop")
        return optimizer.step()

    def state_dict(self):
        return {"scale": 1.0}

    def load_state_dict(self, state_dict):
        pass
