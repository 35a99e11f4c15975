"""Block-wise OmniQuant calibration engine on the HIP path.

Reference: quantize/omniquant.py:22-289.  Same per-layer sequence (teacher pass -> LET init -> AdamW loop over
`epochs x nsamples` sample-steps -> fold -> propagate -> save), re-designed for one MI355X:

  * the whole model's blocks and the three activation banks stay resident in HBM (288 GB): no per-layer
    CPU<->GPU shuttling (reference :159,:248);
  * one sample-step (LET+quant of 7 weights, forward, MSE, backward, grad-norm, AdamW) is a fixed sequence of
    HIP kernels captured ONCE per layer into a hipGraph and replayed; the loss and the gradient norm stay on the
    device and are read back once per epoch (the reference syncs with loss.item() every step, :223);
  * teacher / propagate passes run several samples per launch (per-token quantisation is batch-invariant).
"""
import math
import os
from collections import OrderedDict
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _capi as C
from .linear import QuantLinear
from .optim import BlockOptimizer


def get_named_linears(module):
    return {name: m for name, m in module.named_modules() if isinstance(m, QuantLinear)}


def quant_params_from_args(args):
    """The five quantizer kwarg dicts of main.py:268-303."""
    args.weight_quant_params = {"n_bits": args.wbits, "per_channel_axes": [0], "symmetric": getattr(args, "symmetric", False),
                                "dynamic_method": getattr(args, "w_dynamic_method", "per_channel"),
                                "group_size": getattr(args, "group_size", None), "lwc": args.lwc}
    act = {"n_bits": args.abits, "per_channel_axes": [], "symmetric": False,
           "dynamic_method": getattr(args, "a_dynamic_method", "per_token")}
    args.act_quant_params = dict(act)
    args.q_quant_params = dict(act)
    args.k_quant_params = dict(act)
    args.v_quant_params = dict(act)
    args.p_quant_params = {"n_bits": 16, "metric": "fix0to1"}
    return args


def default_args(**kw):
    """Hyper-parameters with the reference's defaults (main.py:193-229)."""
    a = SimpleNamespace(wbits=4, abits=4, group_size=None, alpha=0.5, let_lr=5e-3, lwc_lr=1e-2, wd=0.0, epochs=10,
                        let=False, lwc=False, aug_loss=False, symmetric=False, nsamples=128, batch_size=1,
                        deactive_amp=False, resume=None, real_quant=False, output_dir=None, net="llama-7b")
    for k, v in kw.items():
        setattr(a, k, v)
    return quant_params_from_args(a)


LET_PAIRS = {"llama": {"q_proj": "qkv", "o_proj": "out", "up_proj": "fc1"},
             "opt": {"q_proj": "qkv", "out_proj": "out", "fc1": "fc1"}}
LAYER_PREFIX = {"llama": "model.layers", "opt": "model.decoder.layers"}


def family_of(net):
    n = net.lower()
    if "llama" in n:
        return "llama"
    if "opt" in n:
        return "opt"
    raise ValueError("Only support for opt/llama/Llama-2 on the HIP path")


def decoder_layer_class(family):
    if family == "llama":
        from .llama_block import QuantLlamaDecoderLayer
        return QuantLlamaDecoderLayer
    from .opt_block import QuantOPTDecoderLayer
    return QuantOPTDecoderLayer


def register_let_parameters(qlayer, family, act_scales, act_shifts, alpha, layer_idx, dev, use_shift=True):
    """LET init of quantize/omniquant.py:182-197 (signed column max, llama shifts start at 0).  Once per layer."""
    pairs = LET_PAIRS[family]
    prefix = LAYER_PREFIX[family]
    is_llama = family == "llama"
    dtype = torch.float32
    qlayer.register_parameter("qkt_smooth_scale", nn.Parameter(
        torch.ones(qlayer.self_attn.q_proj.out_features, device=dev, dtype=dtype)))
    for name, module in qlayer.named_modules():
        if isinstance(module, QuantLinear):
            for key in pairs.keys():
                if key in name:
                    act = act_scales[f"{prefix}.{layer_idx}.{name}"].to(device=dev, dtype=dtype).clamp(min=1e-5)
                    weight = module.weight.float().max(dim=0)[0].clamp(min=1e-5)
                    scale = (act.pow(alpha) / weight.pow(1 - alpha)).clamp(min=1e-5)
                    if use_shift and not is_llama:
                        shift = act_shifts[f"{prefix}.{layer_idx}.{name}"].to(device=dev, dtype=dtype)
                    else:
                        shift = torch.zeros_like(scale)
                    qlayer.register_parameter(f"{pairs[key]}_smooth_shift", nn.Parameter(shift.clone()))
                    qlayer.register_parameter(f"{pairs[key]}_smooth_scale", nn.Parameter(scale.clone()))


class StepRunner:
    """One calibration sample-step (quantize/omniquant.py:214-230) as a replayable hipGraph."""

    def __init__(self, qlayer, opt, mask, position_ids, sample_shape, dtype, aug_loss, is_llama, use_graph=True):
        dev = opt.flat.device
        self.qlayer, self.opt, self.mask, self.pos, self.is_llama = qlayer, opt, mask, position_ids, is_llama
        self.x = torch.zeros(sample_shape, dtype=dtype, device=dev)
        self.t1 = torch.zeros(sample_shape, dtype=dtype, device=dev)
        self.t2 = torch.zeros(sample_shape, dtype=dtype, device=dev) if aug_loss else None
        self._loss_buf = torch.zeros(1 + 1024, dtype=torch.float32, device=dev)     # [0] loss, [1:] oq_mse scratch
        self.loss = self._loss_buf[:1]
        self.graph = None
        self.use_graph = use_graph
        self.steps = 0
        # opt-in: on ROCm 7.2 a hipGraph replays its parallel branches on ONE hardware queue (measured: zero overlap in
        # the rocprofv3 trace), so the second stream only pays off for un-captured (eager) stepping
        # MLP weights are fake-quantised right before their GEMMs instead of at the start of the step (cache locality;
        # identical kernels and values).  OQ_LAZY_MLP=0 restores the reference's order for A/B.
        if os.environ.get("OQ_LAZY_MLP", "1") != "0":
            qlayer.__dict__["_lazy_mlp_quant"] = True
        if os.environ.get("OQ_WEIGHT_STREAM", "0") != "0":
            qlayer.__dict__["_fq_stream"] = torch.cuda.Stream(device=dev)   # see QuantBlockMixin._weight_stream

    def _forward(self):
        if self.is_llama:
            return self.qlayer(self.x, attention_mask=self.mask, position_ids=self.pos)[0]
        return self.qlayer(self.x, attention_mask=self.mask)[0]

    def _step(self):
        from .ops import WeightQuantBatch
        WeightQuantBatch.drop_stale()       # nothing of a previous (possibly failed) step may be launched by this one
        self.qlayer.smooth_and_quant_temporary()
        out = self._forward()
        g = torch.empty_like(out)
        from . import ops
        wide = ops.wide_of(out) if out.dtype == torch.bfloat16 else None      # the block output's un-rounded side channel
        src = wide.contiguous() if wide is not None else out
        C.call("oq_mse_fwd_bwd", C.ptr(src), C.dt(src), C.ptr(self.t1), C.ptr(self.t2), C.dt(out), out.numel(), 1.0,
               C.fptr(self._loss_buf), C.ptr(g), C.stream())
        self.opt.zero_grad(lazy=True)       # a no-op once the fused optimiser step clears the arena behind each update
        ops.WgradQueue.enabled = True       # (with OQ_WGRAD_GROUP=1: the block's weight-gradient GEMMs are parked and grouped)
        try:
            out.backward(g)
        finally:
            ops.WgradQueue.enabled = False
        ops.WeightQuantBatch.flush_pending()        # (normally done by the engine callbacks already)
        self.opt.step()

    def _capture(self):
        """Warm up and capture on the SAME side stream: autograd binds each parameter's AccumulateGrad node to the
        stream of its first backward; if the capture ran on another stream the `grad +=` kernels would be forked
        onto the warm-up stream and race with the allocator's reuse of the incoming gradient buffers."""
        opt = self.opt
        snap = [t.clone() for t in (opt.flat, opt.exp_avg, opt.exp_avg_sq, opt.step_count)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):          # warm-up: allocator + lazily cached tables
                self._step()
            for t, c in zip((opt.flat, opt.exp_avg, opt.exp_avg_sq, opt.step_count), snap):
                t.copy_(c)
            if opt.step_log is not None:
                opt.step_log[:1].zero_()          # the warm-up steps are not part of the record
            # the captured step starts from scales the fused update keeps truncated: do it once for the restored values
            opt.truncate_scales(force=True)
            self.qlayer.clear_temp_variable()     # drop the warm-up autograd graph
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=s):
            self._step()

    def run(self, x, t1, t2=None):
        """x, t1, t2: [bs, T, H] device tensors of this sample.  Returns nothing; loss/norm stay on device."""
        pairs = [(x, self.x), (t1, self.t1)] + ([(t2, self.t2)] if self.t2 is not None else [])
        if all(s.is_contiguous() and s.dtype == d.dtype and s.shape == d.shape and s.is_cuda for s, d in pairs) \
                and (self.x.numel() * self.x.element_size()) % 16 == 0:
            flat = [C.ptr(t) for s, d in pairs for t in (s, d)] + [None] * (6 - 2 * len(pairs))
            C.call("oq_copy_samples", len(pairs), *flat, self.x.numel() * self.x.element_size(), C.stream())   # one launch
        else:
            for s, d in pairs:
                d.copy_(s)
        if self.use_graph:
            if self.graph is None:
                self._capture()
            self.graph.replay()
        else:
            self._step()
        self.steps += 1


@torch.no_grad()
def forward_bank(qlayer, bank, out, mask, pos, is_llama, chunk=4):
    """out[j] = qlayer(bank[j]) for every sample, `chunk` samples per launch."""
    n = bank.shape[0]
    for j in range(0, n, chunk):
        x = bank[j:j + chunk]
        m = mask.expand(x.shape[0], -1, -1, -1) if mask is not None else None
        y = qlayer(x, attention_mask=m, position_ids=pos)[0] if is_llama else qlayer(x, attention_mask=m)[0]
        out[j:j + chunk] = y


def calibrate_block(qlayer, args, family, layer_idx, quant_inps, fp_inps, fp_inps_2, mask, position_ids,
                    act_scales=None, act_shifts=None, resume_params=None, logger=None, use_graph=True,
                    compute_dtype=torch.bfloat16, bank_chunk=4):
    """Everything the reference does for ONE layer (quantize/omniquant.py:165-250) except moving it between
    devices.  quant_inps / fp_inps(/fp_inps_2) are updated in place.  Returns dict(losses, norms, omni)."""
    dev = quant_inps.device
    is_llama = family == "llama"
    qlayer.compute_dtype = compute_dtype
    nsamples = quant_inps.shape[0]
    # ---- teacher pass ----------------------------------------------------------------------------------
    qlayer.set_quant_state(weight_quant=False, act_quant=False)
    stat_mods, stats = [], None
    if args.let and act_scales is None:
        # no pre-computed statistics (generate_act_scale_shift.py was not run): gather them from the teacher pass
        from .actstats import ActStatCollector
        stats = ActStatCollector()
    if args.epochs > 0 or stats is not None:
        if args.aug_loss and args.epochs > 0:
            forward_bank(qlayer, quant_inps, fp_inps_2, mask, position_ids, is_llama, bank_chunk)
        if stats is not None:
            stat_mods = stats.attach(qlayer, LAYER_PREFIX[family], layer_idx, only=list(LET_PAIRS[family].keys()))
        forward_bank(qlayer, fp_inps, fp_inps, mask, position_ids, is_llama, bank_chunk)
        if stats is not None:
            stats.detach(stat_mods)
            act_scales, act_shifts = stats.scales, stats.shifts
    # ---- learnables ------------------------------------------------------------------------------------
    qlayer.set_quant_state(weight_quant=False, act_quant=True)
    qlayer.let = args.let
    use_shift = True
    if args.let:
        register_let_parameters(qlayer, family, act_scales, act_shifts, args.alpha, layer_idx, dev, use_shift)
    if resume_params is not None:
        qlayer.load_state_dict(resume_params, strict=False)
    losses, norms = [], []
    if args.epochs > 0:
        with torch.no_grad():
            for p in qlayer.parameters():       # learnables in float32 (weights keep their fp16 master)
                p.data = p.data.float()
        opt = BlockOptimizer(qlayer, args.let_lr, args.lwc_lr, args.wd, use_shift)
        bs = args.batch_size
        runner = StepRunner(qlayer, opt, mask.expand(bs, -1, -1, -1).contiguous() if mask is not None else None,
                            position_ids, (bs,) + tuple(quant_inps.shape[1:]), compute_dtype, args.aug_loss, is_llama,
                            use_graph)
        nsteps = nsamples // bs
        # loss and gradient norm of every step are appended to a device log by the step's own optimiser launch (captured in
        # the graph): the loop issues nothing per step besides the sample copy and the replay
        on_device_log = opt.fused            # (OQ_FUSED_OPT=0, the A/B path with separate launches, copies per step instead)
        if on_device_log:
            opt.attach_step_log(runner.loss, nsteps)
        else:
            loss_buf = torch.zeros(nsteps, dtype=torch.float32, device=dev)
            norm_buf = torch.zeros(nsteps, dtype=torch.float32, device=dev)
        for epoch in range(args.epochs):
            for j in range(nsteps):
                i0 = j * bs
                runner.run(quant_inps[i0:i0 + bs], fp_inps[i0:i0 + bs], fp_inps_2[i0:i0 + bs] if args.aug_loss else None)
                if not on_device_log:
                    loss_buf[j:j + 1].copy_(runner.loss)
                    norm_buf[j:j + 1].copy_(opt.norm[0:1])
            if on_device_log:
                ep_loss, ep_norm = opt.read_step_log()                   # ONE sync per epoch
                if len(ep_loss) != nsteps:
                    raise C.OQError(f"layer {layer_idx} epoch {epoch}: the step log holds {len(ep_loss)} of {nsteps} steps")
            else:
                ep_loss, ep_norm = loss_buf.tolist(), norm_buf.tolist()
            losses += ep_loss
            norms += ep_norm
            if not all(math.isfinite(v) for v in ep_loss):
                raise FloatingPointError(f"layer {layer_idx} epoch {epoch}: loss is NaN/inf")
            if logger:
                logger.info(f"layer {layer_idx} iter {epoch} loss:{sum(ep_loss) / len(ep_loss)} "
                            f"norm:{sum(ep_norm) / len(ep_norm)} max memory_allocated "
                            f"{torch.cuda.max_memory_allocated(dev) / 1024 ** 2} ")
        qlayer.clear_temp_variable()
        del runner, opt
    # ---- fold + propagate --------------------------------------------------------------------------------
    qlayer.smooth_and_quant_inplace()
    omni = None
    if args.epochs > 0:
        forward_bank(qlayer, quant_inps, quant_inps, mask, position_ids, is_llama, bank_chunk)
        qlayer.register_scales_and_zeros()
        qlayer.half()
        omni = OrderedDict((k, v.detach().cpu()) for k, v in qlayer.omni_state_dict().items())
    else:
        qlayer.register_scales_and_zeros()
        qlayer.half()
    for m in qlayer.modules():
        if isinstance(m, QuantLinear):
            m.drop_cache()
    return dict(losses=losses, norms=norms, omni=omni, act_scales=act_scales, act_shifts=act_shifts)


def calibrate_layers(layers, config, args, inps, attention_mask, position_ids=None, act_scales=None,
                     act_shifts=None, logger=None, use_graph=True, compute_dtype=torch.bfloat16, layer_offset=0,
                     student_inps=None, keep_on_device=True):
    """Sequential calibration of `layers` (HF decoder layers already on the GPU).
    inps [nsamples, T, H]: teacher input of the first layer; student_inps defaults to the same tensor.
    Returns (list of quantised layers, omni_parameters dict, per-step losses, final (quant_inps, fp_inps))."""
    family = family_of(args.net)
    DecoderLayer = decoder_layer_class(family)
    dev = inps.device
    quant_inps = (student_inps if student_inps is not None else inps).to(compute_dtype).clone()
    fp_inps = inps.to(compute_dtype).clone()
    fp_inps_2 = inps.to(compute_dtype).clone() if args.aug_loss else None
    mask = attention_mask.float().contiguous() if attention_mask is not None else None
    # omni_parameters.pth is {layer_idx: OrderedDict(name -> fp16 tensor)}: loadable without unpickling code
    resume = torch.load(args.resume, map_location="cpu", weights_only=True) if getattr(args, "resume", None) else {}
    omni_parameters = dict(resume) if resume else {}
    all_losses, qlayers = [], []
    for i, layer in enumerate(layers):
        gi = layer_offset + i
        if logger:
            logger.info(f"=== Start quantize layer {gi} ===")
        qlayer = DecoderLayer(config, layer.to(dev), args).to(dev)
        res = calibrate_block(qlayer, args, family, gi, quant_inps, fp_inps, fp_inps_2, mask, position_ids, act_scales,
                              act_shifts, resume.get(gi) if resume else None, logger, use_graph, compute_dtype)
        all_losses += res["losses"]
        if res["omni"] is not None:
            omni_parameters[gi] = res["omni"]
            if getattr(args, "output_dir", None):
                torch.save(omni_parameters, os.path.join(args.output_dir, "omni_parameters.pth"))
        if getattr(args, "real_quant", False):
            from .realquant import pack_block
            pack_block(qlayer, args.wbits)         # quantize/omniquant.py:255-278
        qlayers.append(qlayer if keep_on_device else qlayer.to("cpu"))
    return qlayers, omni_parameters, all_losses, (quant_inps, fp_inps)


def omniquant(lm, args, dataloader, act_scales, act_shifts, logger=None):
    """Drop-in for quantize/omniquant.py::omniquant.  Layer-0 input capture (:89-113) runs the HF embedding +
    first layer once per sample (plumbing); everything after that is the HIP path."""
    model, dev = lm.model, lm.device
    family = family_of(args.net)
    use_cache = model.config.use_cache
    model.config.use_cache = False
    if family == "llama":
        layers = model.model.layers
        embeds = [model.model.embed_tokens, model.model.norm]
    else:
        layers = model.model.decoder.layers
        embeds = [model.model.decoder.embed_tokens, model.model.decoder.embed_positions]
    for m in embeds:
        m.to(dev)
    layers[0] = layers[0].to(dev)
    compute_dtype = torch.float32 if (args.deactive_amp and args.epochs > 0) else torch.bfloat16
    inps = torch.zeros((args.nsamples, lm.seqlen, model.config.hidden_size), dtype=torch.float16, device=dev)
    cache = {"i": 0}

    class Catcher(nn.Module):
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, inp, **kwargs):
            inps[cache["i"]] = inp
            cache["i"] += 1
            cache["attention_mask"] = kwargs.get("attention_mask")
            cache["position_ids"] = kwargs.get("position_ids")
            raise ValueError

    layers[0] = Catcher(layers[0])
    with torch.no_grad():
        for batch in dataloader:
            if cache["i"] >= args.nsamples:
                break
            try:
                model(batch[0].to(dev))
            except ValueError:
                pass
    layers[0] = layers[0].module
    for m in embeds:
        m.cpu()
    attention_mask = cache["attention_mask"]
    if attention_mask is None:     # newer transformers hand the layer no mask for pure-causal inputs
        T = lm.seqlen
        attention_mask = torch.triu(torch.full((T, T), torch.finfo(torch.float32).min, device=dev), 1)[None, None]
    position_ids = cache["position_ids"] if family == "llama" else None
    if logger:
        logger.info("Starting ...")
    qlayers, omni, _, _ = calibrate_layers([l for l in layers], model.config, args, inps, attention_mask, position_ids,
                                           act_scales, act_shifts, logger, True, compute_dtype, keep_on_device=False)
    for i, q in enumerate(qlayers):
        layers[i] = q
    torch.cuda.empty_cache()
    model.config.use_cache = use_cache
    return model
