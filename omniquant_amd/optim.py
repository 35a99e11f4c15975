"""Optimiser side of the calibration step on the HIP path.

Reference: utils.py:11-46 (NativeScalerWithGradNormCount, ampscaler_get_grad_norm) and the two-group
torch.optim.AdamW built at quantize/omniquant.py:207-208.

MI355X-native layout: every learnable of a block (LET scales, LET shifts, LWC bounds) is re-homed into ONE
contiguous float32 arena; the nn.Parameters keep their identity, names and shapes but become views, and so do
their .grad tensors.  zero_grad = one memset, grad-norm = one reduction, AdamW = one kernel with the step
counter and the finite-flag on the device (graph-replayable, no host sync, no loss scaling needed for bf16).
"""
import ctypes
import os

import torch

from . import _capi as C

SUM_MAX_ENTRIES, SUM_MAX_SRC = 16, 6      # OQ_SUM_MAX_ENTRIES / OQ_SUM_MAX_SRC of include/oq_hip.h


class GradCollector:
    """Gathers the partial gradients of SHARED learnables straight into the gradient arena.

    A LET vector such as qkv_smooth_scale feeds four backward kernels (the fused fake-quant of q, k and v and the
    LET vector kernel).  Left to autograd, their four results are summed by three tiny `add` launches plus one more
    into .grad -- 22 launches (~4.7 us each) per step for a LLaMA block.  Instead the autograd Functions of ops.py
    hand those partial tensors to this collector (and return None to autograd), and `flush()` -- called once per
    step, after backward -- writes grad_slot = sum(partials) for every parameter with ONE oq_sum_vectors launch, in
    the fixed order the partials were produced.  Single-use learnables (the LWC bounds) do not come here at all:
    their backward kernel writes into the arena directly (`_oq_grad_sink`)."""

    def __init__(self):
        self.pending = {}     # id(param) -> (grad view, [partial tensors])

    def add(self, param, partial):
        ent = self.pending.get(id(param))
        if ent is None:
            ent = self.pending[id(param)] = (param._oq_grad_sink, [])
        ent[1].append(partial.contiguous().view(-1))

    def flush(self):
        from .ops import WeightQuantBatch
        WeightQuantBatch.flush_pending()      # batched weight-quantiser backward launches whose siblings never arrived
        if not self.pending:
            return
        entries = list(self.pending.values())
        self.pending = {}
        for i in range(0, len(entries), SUM_MAX_ENTRIES):
            chunk = entries[i:i + SUM_MAX_ENTRIES]
            E = len(chunk)
            dst = (ctypes.c_void_p * E)()
            n = (ctypes.c_int64 * E)()
            nsrc = (ctypes.c_int * E)()
            src = (ctypes.c_void_p * (E * SUM_MAX_SRC))()
            keep = []
            for e, (gview, parts) in enumerate(chunk):
                if len(parts) > SUM_MAX_SRC:        # more partials than one launch takes: pre-sum the tail with torch
                    head, tail = parts[:SUM_MAX_SRC - 1], parts[SUM_MAX_SRC - 1:]
                    parts = head + [torch.stack(tail).sum(0)]
                keep.append(parts)
                dst[e] = C.fptr(gview.view(-1))
                n[e] = gview.numel()
                nsrc[e] = len(parts)
                for k, t in enumerate(parts):
                    if t.numel() != gview.numel():
                        raise C.OQError("GradCollector: partial gradient has the wrong size")
                    src[e * SUM_MAX_SRC + k] = C.fptr(t)
            C.call("oq_sum_vectors", E, dst, n, nsrc, src, C.stream())


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
        self._ws = torch.zeros(2048, dtype=torch.float32, device=dev)      # oq_adamw_step / oq_gradnorm workspace
        self.collector = GradCollector()
        off = 0
        with torch.no_grad():
            for _, p in self.named:
                k = p.numel()
                self.flat[off:off + k].copy_(p.data.reshape(-1).float())
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                p.requires_grad_(True)
                # gradient routing without autograd's accumulate kernels (ops.py reads these attributes)
                p._oq_grad_sink = p.grad
                p._oq_collector = self.collector
                off += k
        if self.n_scale:
            qlayer.__dict__["_arena_truncate"] = self.truncate_scales     # block's truncate_number -> one launch
        self.let_lr, self.lwc_lr, self.wd = float(let_lr), float(lwc_lr), float(weight_decay)
        self.betas, self.eps = betas, float(eps)
        # fused step (oq_adamw_step): the update also truncates the LET scales (what the next step would do first) and
        # clears the gradient arena, so the steady-state sample-step launches neither oq_truncate nor a fill.
        # OQ_FUSED_OPT=0 keeps the separate launches for A/B.
        self.fused = os.environ.get("OQ_FUSED_OPT", "1") != "0"
        self.truncate_thr = 1e-2
        # on-device step log (oq_adamw_step): (loss, gradient norm) of every step since the last reset; set by attach_step_log
        self.step_log = None
        self._log_loss = None
        self.clear_grads_in_step = True   # False: .grad survives step() as with torch (tests that read the gradients)
        self._grads_clean = True          # the arena is all zeros (set by the fused step, cleared by anything that may write it)
        self._scales_truncated = False    # the scales are known to satisfy |s| >= thr (set by the fused step)

    def attach_step_log(self, loss, n_steps):
        """Have every fused step append (loss[0], gradient norm) to a device buffer instead of the host fetching them per step
        (quantize/omniquant.py:223-231 does loss.item() each step).  `loss`: the 1-element float32 device tensor the step's
        loss kernel writes.  Read with read_step_log() -- one sync per epoch -- which also rewinds the log."""
        if loss.dtype != torch.float32 or loss.numel() != 1 or not loss.is_cuda:
            raise C.OQError("attach_step_log: loss must be a 1-element float32 GPU tensor")
        self._log_loss = loss
        self.step_log = torch.zeros(1 + 2 * int(n_steps), dtype=torch.float32, device=self.flat.device)

    def read_step_log(self):
        """(losses, norms) of the steps logged since the last call; ONE device-to-host copy, then the log is rewound."""
        if self.step_log is None:
            return [], []
        host = self.step_log.tolist()
        n = min(int(host[0]), (len(host) - 1) // 2)
        self.step_log[:1].zero_()
        return host[1:1 + 2 * n:2], host[2:2 + 2 * n:2]

    def zero_grad(self, set_to_none=False, lazy=False):
        """lazy=True (the engine's own step loop): skip the launch when the fused step has just cleared the arena."""
        if lazy and self._grads_clean:
            return
        self.grad.zero_()
        self._grads_clean = True

    def truncate_scales(self, thr=1e-2, force=False):
        """truncate_number over all LET scales in one launch (they are the head of the arena)."""
        if not self.n_scale:
            return
        if self._scales_truncated and not force and float(thr) == self.truncate_thr:
            return                        # the fused step already did it behind the last update
        C.call("oq_truncate", C.fptr(self.flat), self.n_scale, float(thr), C.stream())

    def collect_grads(self):
        """Finish the backward pass: one launch sums the partial gradients of the shared learnables into the arena."""
        self._grads_clean = False
        self.collector.flush()

    def grad_norm(self):
        self.collector.flush()
        C.call("oq_gradnorm", C.fptr(self.grad), self.n, C.fptr(self.norm), C.fptr(self._ws), C.stream())
        return self.norm[0]

    def step(self):
        """grad-norm + AdamW; the update is skipped on device when a gradient is non-finite."""
        if self.fused:
            self.collector.flush()
            C.call("oq_adamw_step", C.fptr(self.flat), C.fptr(self.grad), C.fptr(self.exp_avg), C.fptr(self.exp_avg_sq),
                   self.n, self.n_let, self.n_scale, self.truncate_thr, int(self.clear_grads_in_step), self.let_lr,
                   self.lwc_lr, self.betas[0], self.betas[1], self.eps, self.wd, C.fptr(self.step_count), C.fptr(self.norm),
                   C.fptr(self._ws), C.fptr(self._log_loss) if self.step_log is not None else None, C.fptr(self.step_log),
                   (self.step_log.numel() - 1) // 2 if self.step_log is not None else 0, C.stream())
            self._grads_clean = self.clear_grads_in_step
            self._scales_truncated = True
            return self.norm[0]
        self._grads_clean = False
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
        if hasattr(optimizer, "collect_grads"):
            optimizer.collect_grads()
        if not update_grad:
            return None
        if clip_grad is not None:
            raise NotImplementedError("gradient clipping is not used by the calibration loop")
        return optimizer.step()

    def state_dict(self):
        return {"scale": 1.0}

    def load_state_dict(self, state_dict):
        pass
