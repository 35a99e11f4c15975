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


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE) must start N fresh rank processes itself -- torch.distributed.run on
    127.0.0.1 -- before anything touches a GPU, relay rank 0's single JSON line and return the children's exit code."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], port=29511, python="py")
    assert cmd[:3] == ["py", "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OQ_BENCH_LAUNCH_ONLY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["world"] == 2 and sorted(map(tuple, lines[0]["ranks"])) == [(0, 0), (1, 1)]
    # a world size that does not match --gpus is refused with a message, not an assert
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True,
                        timeout=300)
    assert r2.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in r2.stderr
