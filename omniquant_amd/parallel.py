"""Layer-sharded calibration across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-GPU calibration: blocks are trained strictly in sequence because block
i's student input is block i-1's trained output.  The sharding implemented here is the documented approximation:

  1. rank r owns the contiguous chunk  layers[lo(r):hi(r)];
  2. teacher pre-pass, pipelined: rank r receives the FP activation bank at its chunk boundary from rank r-1
     (point-to-point send/recv -- on MI355X one xGMI link, 2.15 GB at LLaMA-7B), forwards it through its own chunk
     with quantisation off and sends the result to rank r+1;
  3. every rank calibrates its chunk exactly like the single-GPU engine, except that the student input of its FIRST
     block is the teacher activation (G-1 boundaries deviate from the reference);
  4. learned parameters are gathered on rank 0 (gather_object of small fp16 dicts).

No collective runs inside the sample-step loop.  The compute is injected as two callables so the protocol can be
tested on CPU with gloo (tests/test_parallel_gloo.py) and used with the HIP engine on GPUs.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_layers, world, rank):
    """Contiguous, balanced partition: the first (n_layers % world) ranks get one extra layer."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} / world {world}")
    base, extra = divmod(n_layers, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def pipeline_teacher_boundaries(inps, lo, hi, teacher_chunk_forward, group=None):
    """Step 2.  `teacher_chunk_forward(lo, hi, bank) -> bank` runs layers [lo, hi) with quantisation off.
    Returns the FP activation bank at THIS rank's chunk input (rank 0: `inps` itself)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    # RCCL ("nccl") moves device buffers directly over xGMI; gloo (CPU tests, or ranks sharing one GPU) has no
    # device point-to-point, so the bank is staged through host memory there
    staged = inps.is_cuda and dist.get_backend(group) == "gloo"
    bank = inps
    if rank > 0:
        buf = torch.empty(inps.shape, dtype=inps.dtype, device="cpu" if staged else inps.device)
        dist.recv(buf, src=rank - 1, group=group)
        bank = buf.to(inps.device) if staged else buf
    if rank < world - 1:
        out = teacher_chunk_forward(lo, hi, bank.clone()).contiguous()
        dist.send(out.cpu() if staged else out, dst=rank + 1, group=group)
    return bank


def gather_omni_parameters(local, group=None):
    """Step 4.  local: {global_layer_idx: OrderedDict(name -> CPU tensor)}.  Returns the merged dict on rank 0."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(local, gathered, dst=0, group=group)
    if rank != 0:
        return None
    merged = {}
    for part in gathered:
        for k, v in part.items():
            if k in merged:
                raise RuntimeError(f"layer {k} calibrated by two ranks")
            merged[k] = v
    return dict(sorted(merged.items()))


def calibrate_sharded(n_layers, inps, teacher_chunk_forward, calibrate_chunk, group=None):
    """Full protocol.  `calibrate_chunk(lo, hi, teacher_bank, student_bank) -> {layer_idx: omni dict}`.
    Returns (merged omni parameters on rank 0 / None elsewhere, (lo, hi))."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(n_layers, world, rank)
    bank = pipeline_teacher_boundaries(inps, lo, hi, teacher_chunk_forward, group)
    local = calibrate_chunk(lo, hi, bank, bank.clone()) if hi > lo else {}
    merged = gather_omni_parameters(local, group)
    return merged, (lo, hi)


def hip_callables(layers, config, args, attention_mask, position_ids, act_scales=None, act_shifts=None, logger=None,
                  compute_dtype=torch.bfloat16, use_graph=True):
    """The two callables backed by the HIP engine (omniquant_amd.calibrate)."""
    from .calibrate import calibrate_layers, decoder_layer_class, family_of, forward_bank

    family = family_of(args.net)
    is_llama = family == "llama"

    def teacher_chunk_forward(lo, hi, bank):
        bank = bank.to(compute_dtype)
        mask = attention_mask.float().contiguous()
        for i in range(lo, hi):
            q = decoder_layer_class(family)(config, layers[i], args).to(bank.device)
            q.compute_dtype = compute_dtype
            q.set_quant_state(weight_quant=False, act_quant=False)
            forward_bank(q, bank, bank, mask, position_ids, is_llama)
        return bank

    def calibrate_chunk(lo, hi, teacher_bank, student_bank):
        _, omni, _, _ = calibrate_layers(layers[lo:hi], config, args, teacher_bank, attention_mask, position_ids,
                                         act_scales, act_shifts, logger, use_graph, compute_dtype, layer_offset=lo,
                                         student_inps=student_bank)
        return omni

    return teacher_chunk_forward, calibrate_chunk
