"""Host-side logic of the real-quant path on CPU: the oracle's restatement of AutoGPTQ's pack() (parity unpinned: the
library is not vendored in the reference) and the product's pure-torch `unpack` must be inverse to each other for every
bit width, including the 3-bit layout whose channels 10 and 21 straddle two words."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from omniquant_amd.realquant import unpack


@pytest.mark.parametrize("bits,group", [(2, 32), (3, 64), (4, None), (4, 32), (8, None)])
def test_oracle_pack_and_product_unpack_are_inverse(bits, group):
    g = torch.Generator().manual_seed(bits)
    out, inn = 64, 128
    ng = inn // (group or inn)
    Q = (1 << bits) - 1
    codes = torch.randint(0, Q + 1, (out, inn), generator=g)
    scales = torch.rand(out, ng, generator=g) * 0.1 + 0.01
    zeros = torch.randint(1, Q + 1, (out, ng), generator=g).float()       # AutoGPTQ stores zeros - 1: keep them >= 1
    gi = torch.arange(inn) // (group or inn)
    W = (codes.float() - zeros[:, gi]) * scales[:, gi]                    # exactly representable fake-quant weight
    qweight, qzeros = R.autogptq_pack(W, scales, zeros, bits, group)
    assert qweight.shape == (inn // 32 * bits, out) and qzeros.shape == (ng, out // 32 * bits)
    back = unpack(qweight, bits, inn)                                      # [in, out]
    np.testing.assert_array_equal(back.t().numpy(), codes.numpy())
    zback = unpack(qzeros.t().contiguous(), bits, out).t()                 # [groups, out]
    np.testing.assert_array_equal(zback.numpy(), (zeros.t() - 1).long().numpy())
