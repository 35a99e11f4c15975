"""Layer-sharded calibration across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-GPU calibration: blocks are trained strictly in sequence because block
i's student input is block i-1's trained output.  The sharding implemented here is the documented approximation:

  1. rank r owns the contiguous chunk  layers[lo(r):hi(r)];
  2. teacher pre-pass, pipelined: rank r receives the FP activation bank at its chunk boundary from rank r-1
     (point-to-point send/recv -- on MI355X one xGMI link, 2.15 GB at LLaMA-7B), forwards it through its own chunk
     with quantisation off and sends the result to rank r+1 -- streamed in messages of a few samples with the next
     receive and the previous send in flight under the compute, so all ranks work on the pre-pass at once;
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


def pipeline_teacher_boundaries(inps, lo, hi, teacher_chunk_forward, group=None, chunk=None):
    """Step 2.  `teacher_chunk_forward(lo, hi, bank) -> bank` runs layers [lo, hi) with quantisation off on ANY number
    of samples.  Returns the FP activation bank at THIS rank's chunk input (rank 0: `inps` itself).

    The bank is streamed in messages of `chunk` samples (None: one message, store-and-forward): rank r forwards message
    c through its layers and isend()s the result while it already computes message c+1, and it posts the irecv() of
    message c+1 before computing c -- so rank r+1 starts after ONE message has crossed r's layers, not the whole bank,
    and the last of G ranks starts after G-1 message latencies instead of G-1 bank passes.  Only point-to-point
    send/recv (RCCL over one xGMI link per boundary); message order per (src, dst) pair is FIFO in both backends."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = inps.shape[0]
    step = n if not chunk or chunk <= 0 else min(int(chunk), n)
    # RCCL ("nccl") moves device buffers directly over xGMI; gloo (CPU tests, or ranks sharing one GPU) has no
    # device point-to-point, so messages are staged through host memory there
    staged = inps.is_cuda and dist.get_backend(group) == "gloo"
    comm_dev = "cpu" if staged else inps.device
    bounds = [(a, min(a + step, n)) for a in range(0, n, step)]
    bank = inps
    recv_req = None
    if rank > 0:
        bank = torch.empty(inps.shape, dtype=inps.dtype, device=inps.device)
        rbuf = [torch.empty((b - a,) + tuple(inps.shape[1:]), dtype=inps.dtype, device=comm_dev) for a, b in bounds]
        recv_req = dist.irecv(rbuf[0], src=rank - 1, group=group)
    pending = []
    for c, (a, b) in enumerate(bounds):
        if rank > 0:
            recv_req.wait()
            if c + 1 < len(bounds):
                recv_req = dist.irecv(rbuf[c + 1], src=rank - 1, group=group)     # next message lands under this compute
            if rbuf[c].dtype != bank.dtype or tuple(rbuf[c].shape) != tuple(bank[a:b].shape):
                raise RuntimeError("boundary message does not have the bank's dtype / shape")
            bank[a:b].copy_(rbuf[c])
            rbuf[c] = None
        if rank < world - 1:
            out = teacher_chunk_forward(lo, hi, bank[a:b].clone())
            # the wire format is the BANK's dtype and shape: the receiver's buffer was allocated from `inps`, and neither
            # RCCL nor gloo checks that the two sides agree (a bf16 message landing in an fp16 buffer would be reinterpreted
            # bit for bit).  A teacher that computes in another dtype is converted here, once per message.
            if tuple(out.shape) != (b - a,) + tuple(inps.shape[1:]):
                raise RuntimeError(f"teacher_chunk_forward returned shape {tuple(out.shape)} for a message of "
                                   f"{(b - a,) + tuple(inps.shape[1:])}")
            out = out.to(inps.dtype).contiguous()
            msg = out.cpu() if staged else out
            pending.append((dist.isend(msg, dst=rank + 1, group=group), msg))       # keep the buffer alive until sent
    for req, _ in pending:
        req.wait()
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


def calibrate_sharded(n_layers, inps, teacher_chunk_forward, calibrate_chunk, group=None, chunk=None):
    """Full protocol.  `calibrate_chunk(lo, hi, teacher_bank, student_bank) -> {layer_idx: omni dict}`.
    `chunk`: samples per boundary message of the teacher pre-pass (None: the whole bank in one message).
    Returns (merged omni parameters on rank 0 / None elsewhere, (lo, hi))."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(n_layers, world, rank)
    bank = pipeline_teacher_boundaries(inps, lo, hi, teacher_chunk_forward, group, chunk)
    local = calibrate_chunk(lo, hi, bank, bank.clone()) if hi > lo else {}
    merged = gather_omni_parameters(local, group)
    return merged, (lo, hi)


def hip_callables(layers, config, args, attention_mask, position_ids, act_scales=None, act_shifts=None, logger=None,
                  compute_dtype=torch.bfloat16, use_graph=True):
    """The two callables backed by the HIP engine (omniquant_amd.calibrate)."""
    from .calibrate import calibrate_layers, decoder_layer_class, family_of, forward_bank

    family = family_of(args.net)
    is_llama = family == "llama"

    teachers = {}           # layer index -> FP block, built once (the pre-pass calls this once per streamed message)
    mask32 = attention_mask.float().contiguous() if attention_mask is not None else None

    def teacher_chunk_forward(lo, hi, bank):
        bank = bank.to(compute_dtype)
        for i in range(lo, hi):
            q = teachers.get(i)
            if q is None:
                q = decoder_layer_class(family)(config, layers[i], args).to(bank.device)
                q.compute_dtype = compute_dtype
                q.set_quant_state(weight_quant=False, act_quant=False)
                teachers[i] = q
            forward_bank(q, bank, bank, mask32, position_ids, is_llama)
        return bank

    def calibrate_chunk(lo, hi, teacher_bank, student_bank):
        _, omni, _, _ = calibrate_layers(layers[lo:hi], config, args, teacher_bank, attention_mask, position_ids,
                                         act_scales, act_shifts, logger, use_graph, compute_dtype, layer_offset=lo,
                                         student_inps=student_bank)
        return omni

    return teacher_chunk_forward, calibrate_chunk
