"""Host-side decisions of the HIP path that need no GPU: causal-mask detection (what selects the fused attention and
the tile-skipping GEMM modes) and the argument plumbing of the engine's defaults."""
import torch

from omniquant_amd import ops
from omniquant_amd.calibrate import default_args, family_of, LET_PAIRS, LAYER_PREFIX
from omniquant_amd.parallel import shard_bounds


def _causal(T):
    return torch.triu(torch.full((T, T), torch.finfo(torch.float32).min), 1)


def test_mask_is_causal_rules():
    assert ops.mask_is_causal(_causal(128))
    assert ops.mask_is_causal(_causal(512)[None, None])
    assert not ops.mask_is_causal(_causal(384))             # the causal GEMM modes contract in 256-blocks
    assert not ops.mask_is_causal(torch.zeros(128, 128))    # no mask at all is not "causal"
    assert not ops.mask_is_causal(None)
    m = _causal(128)
    m[5, 3] = -1.0                                          # a soft (finite) entry below the diagonal: dense path
    assert not ops.mask_is_causal(m)
    m2 = _causal(128)
    assert ops.mask_is_causal(m2)
    m2[0, 1] = 0.0                                          # edited in place after it was cached: version counter invalidates
    assert not ops.mask_is_causal(m2)
    assert not ops.mask_is_causal(torch.triu(torch.full((128, 128), -1e4), 1))   # -1e4 is not "exactly masked"
    # the mask transformers builds for an fp16 model (what the reference's Catcher records, quantize/omniquant.py:89-113):
    # finfo(float16).min above the diagonal, batch of identical masks
    hf = torch.triu(torch.full((256, 256), torch.finfo(torch.float16).min, dtype=torch.float16), 1)[None, None]
    assert ops.mask_is_causal(hf) and ops.mask_is_causal(hf.float().expand(2, -1, -1, -1))
    pad = hf.float().repeat(2, 1, 1, 1)
    pad[1, 0, :, :3] = torch.finfo(torch.float16).min       # sample 1 is left-padded: not the plain causal mask
    assert not ops.mask_is_causal(pad)
    # slices of one batched mask are judged one by one (the verdict is cached per view window, not per base tensor)
    assert ops.mask_is_causal(pad[:1]) and not ops.mask_is_causal(pad[1:]) and ops.mask_is_causal(pad[:1])


def test_engine_defaults_match_reference_cli():
    a = default_args()
    # main.py:193-229 defaults
    assert (a.wbits, a.abits, a.alpha, a.let_lr, a.lwc_lr, a.wd, a.nsamples, a.batch_size) == (4, 4, 0.5, 5e-3, 1e-2, 0.0, 128, 1)
    assert a.weight_quant_params["n_bits"] == 4 and a.weight_quant_params["dynamic_method"] == "per_channel"
    assert a.act_quant_params["dynamic_method"] == "per_token" and a.p_quant_params["n_bits"] == 16
    assert family_of("Llama-2-13b") == "llama" and family_of("opt-6.7b") == "opt"
    assert set(LET_PAIRS["llama"]) == {"q_proj", "o_proj", "up_proj"} and LAYER_PREFIX["opt"] == "model.decoder.layers"


def test_shard_bounds_cover_exactly_once():
    for n in (1, 7, 32, 80):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = shard_bounds(n, w, r)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
