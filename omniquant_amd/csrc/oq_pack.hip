// Real-quant packing of a fake-quantised weight (quantize/omniquant.py:255-278 -> AutoGPTQ
// qlinear_cuda.QuantLinear.pack).  PARITY UNPINNED: AutoGPTQ is a third-party dependency that is not vendored in the
// reference tree and not pinned (README.md:40 installs an unpinned fork); this restates the published algorithm of
// auto_gptq/nn_modules/qlinear/qlinear_cuda.py (v0.4.x) and is checked against the CPU restatement in the test oracle and
// by an unpack round trip only.
//   q[o][i]   = round((W[o][i] + zeros[o][g] * scales[o][g]) / scales[o][g])        g = i / group
//   qweight   [in/32*bits, out] int32: 32/bits consecutive input channels per word (LSB first); 3 bit: 32 channels in
//             3 words with the two straddling channels split 2+1 / 1+2 bits exactly as AutoGPTQ does
//   qzeros    [in/group, out/32*bits] int32: (zeros - 1) packed the same way along the output channels
// Byte/integer work, HBM-bound: each weight element is read once (2-4 B) and bits/8 B are written.
#include "oq_common.h"

namespace {

template <typename T>
__device__ __forceinline__ uint32_t code_of(const T* w, const float* scales, const float* zeros, int64_t o, int64_t i,
                                            int64_t in, int64_t group, int64_t ngroups) {
    const int64_t g = i / group;
    const float s = scales[o * ngroups + g], z = zeros[o * ngroups + g];
    return (uint32_t)(int)rintf(((float)w[o * in + i] + z * s) / s);
}

// one thread per (output channel, word); consecutive threads = consecutive words of ONE weight row, so the 16-bit
// weight reads (the bulk of the traffic: 32/bits elements per 4-byte word written) are contiguous
template <typename T>
__global__ void __launch_bounds__(256) pack_weight_kernel(const T* w, const float* scales, const float* zeros, int64_t out,
                                                          int64_t in, int64_t group, int bits, uint32_t* qweight) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;      // word row
    const int64_t o = blockIdx.y;
    if (row >= in / 32 * bits) return;
    const int64_t ngroups = in / group;
    uint32_t word = 0;
    if (bits != 3) {
        const int per = 32 / bits;
        const int64_t i0 = row * per;
        for (int j = 0; j < per; ++j) word |= code_of(w, scales, zeros, o, i0 + j, in, group, ngroups) << (bits * j);
    } else {
        const int64_t i0 = (row / 3) * 32;          // 32 channels -> 3 words
        const int ph = (int)(row % 3);
        auto c = [&](int j) { return code_of(w, scales, zeros, o, i0 + j, in, group, ngroups); };
        if (ph == 0) {
            for (int j = 0; j < 10; ++j) word |= c(j) << (3 * j);
            word |= c(10) << 30;
        } else if (ph == 1) {
            word |= (c(10) >> 2) & 1u;
            for (int j = 0; j < 10; ++j) word |= c(11 + j) << (3 * j + 1);
            word |= c(21) << 31;
        } else {
            word |= (c(21) >> 1) & 3u;
            for (int j = 0; j < 10; ++j) word |= c(22 + j) << (3 * j + 2);
        }
    }
    qweight[row * out + o] = word;
}

// qzeros[g][word] over output channels; one thread per word
__global__ void __launch_bounds__(256) pack_zeros_kernel(const float* zeros, int64_t out, int64_t ngroups, int bits,
                                                         uint32_t* qzeros, int64_t words) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ngroups * words) return;
    const int64_t g = idx / words, wd = idx % words;
    auto z = [&](int64_t o) { return (uint32_t)(int)(zeros[o * ngroups + g] - 1.0f); };
    uint32_t word = 0;
    if (bits != 3) {
        const int per = 32 / bits;
        for (int j = 0; j < per; ++j) word |= z(wd * per + j) << (bits * j);
    } else {
        const int64_t o0 = (wd / 3) * 32;
        const int ph = (int)(wd % 3);
        if (ph == 0) {
            for (int j = 0; j < 10; ++j) word |= z(o0 + j) << (3 * j);
            word |= z(o0 + 10) << 30;
        } else if (ph == 1) {
            word |= (z(o0 + 10) >> 2) & 1u;
            for (int j = 0; j < 10; ++j) word |= z(o0 + 11 + j) << (3 * j + 1);
            word |= z(o0 + 21) << 31;
        } else {
            word |= (z(o0 + 21) >> 1) & 3u;
            for (int j = 0; j < 10; ++j) word |= z(o0 + 22 + j) << (3 * j + 2);
        }
    }
    qzeros[g * words + wd] = word;
}

}  // namespace

extern "C" int oq_pack_weights(const void* w, int dtype, int64_t out, int64_t in, int64_t group, int bits,
                               const float* scales, const float* zeros, int32_t* qweight, int32_t* qzeros, void* stream) {
    OQ_CHECK_ARG(w && scales && zeros && qweight && qzeros, "oq_pack_weights: null pointer");
    OQ_CHECK_ARG(bits == 2 || bits == 3 || bits == 4 || bits == 8, "oq_pack_weights: bits %d (2, 3, 4 or 8)", bits);
    OQ_CHECK_ARG(out > 0 && in > 0 && group > 0 && in % group == 0 && in % 32 == 0 && out % 32 == 0,
                 "oq_pack_weights: out=%lld in=%lld group=%lld (in %% group == 0, in and out multiples of 32)", (long long)out,
                 (long long)in, (long long)group);
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = in / 32 * bits;
    OQ_CHECK_ARG(out <= 65535, "oq_pack_weights: out=%lld too large for one launch (split the rows)", (long long)out);
    const dim3 grid((unsigned)((rows + 255) / 256), (unsigned)out);
    if (dtype == OQ_F32)
        hipLaunchKernelGGL((pack_weight_kernel<float>), grid, dim3(256), 0, st, (const float*)w, scales, zeros, out, in, group, bits, (uint32_t*)qweight);
    else if (dtype == OQ_F16)
        hipLaunchKernelGGL((pack_weight_kernel<f16_t>), grid, dim3(256), 0, st, (const f16_t*)w, scales, zeros, out, in, group, bits, (uint32_t*)qweight);
    else if (dtype == OQ_BF16)
        hipLaunchKernelGGL((pack_weight_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)w, scales, zeros, out, in, group, bits, (uint32_t*)qweight);
    else {
        oq_set_error("oq_pack_weights: dtype %d", dtype);
        return OQ_E_UNSUPPORTED;
    }
    const int64_t ngroups = in / group, words = out / 32 * bits;
    hipLaunchKernelGGL(pack_zeros_kernel, dim3((unsigned)((ngroups * words + 255) / 256)), dim3(256), 0, st, zeros, out, ngroups,
                       bits, (uint32_t*)qzeros, words);
    OQ_CHECK_LAUNCH("oq_pack_weights");
    return OQ_OK;
}
