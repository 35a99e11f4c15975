// Activation statistics for the LET initialisation, computed while the FP teacher pass streams the calibration bank.
//
// Replaces the offline pre-pass generate_act_scale_shift.py:25-94 (forward hooks on every nn.Linear):
//   act_scales[c] = max over samples of max_t |x[t, c]|                                    (get_act_scales :29-37)
//   act_shifts[c] = (max_t x + min_t x) / 2 for the first sample, then the running average
//                   0.99 * act_shifts + 0.01 * (max_t x + min_t x) / 2 per further sample   (get_act_shifts :63-72)
// HBM-bound: every activation element is read once (2 B for bf16); 16.8 MB per [2048, 4096] sample.
//   act_stats_partial_kernel  grid (column groups of 512, 32 row slabs, samples): per-column max / min of a slab
//   act_stats_final_kernel    one thread per column: folds the slabs and applies the samples IN ORDER (the running
//                             average is order dependent) -- deterministic, no atomics.
#include "oq_common.h"

namespace {

constexpr int SLABS = 32;

template <typename T>
__global__ void __launch_bounds__(256) act_stats_partial_kernel(const T* x, int64_t rows, int64_t cols, float* ws) {
    __shared__ float red[2][4][512];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * 512 + lane * 8;
    const int slab = blockIdx.y, s = blockIdx.z;
    const int64_t r0 = rows * slab / SLABS, r1 = rows * (slab + 1) / SLABS;
    float mx[8], mn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { mx[i] = -INFINITY; mn[i] = INFINITY; }
    if (c0 < cols) {
        const T* base = x + ((int64_t)s * rows) * cols + c0;
        for (int64_t r = r0 + w; r < r1; r += 4) {
            float v[8];
            Vec8<T>::load(base + r * cols, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) { mx[i] = nmax(mx[i], v[i]); mn[i] = nmin(mn[i], v[i]); }   // NaN propagates like torch.max
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[0][w][lane * 8 + i] = mx[i]; red[1][w][lane * 8 + i] = mn[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int64_t col = (int64_t)blockIdx.x * 512 + c;
        if (col >= cols) continue;
        float a = red[0][0][c], b = red[1][0][c];
#pragma unroll
        for (int k = 1; k < 4; ++k) { a = nmax(a, red[0][k][c]); b = nmin(b, red[1][k][c]); }
        float* o = ws + (((int64_t)s * SLABS + slab) * 2) * cols;
        o[col] = a;
        o[cols + col] = b;
    }
}

__global__ void __launch_bounds__(256) act_stats_final_kernel(const float* ws, int64_t nsamp, int64_t cols, int64_t seen,
                                                              float* scale, float* shift) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float sc = seen > 0 ? scale[c] : 0.f, sh = seen > 0 ? shift[c] : 0.f;
    for (int64_t s = 0; s < nsamp; ++s) {
        float a = -INFINITY, b = INFINITY;
        for (int k = 0; k < SLABS; ++k) {
            const float* o = ws + ((s * SLABS + k) * 2) * cols;
            a = nmax(a, o[c]);
            b = nmin(b, o[cols + c]);
        }
        const float amax = nmax(fabsf(a), fabsf(b));      // max |x| over the sample
        const float mid = (a + b) / 2.f;
        if (seen + s == 0) {
            sc = amax;
            sh = mid;
        } else {
            sc = nmax(sc, amax);
            sh = 0.99f * sh + 0.01f * mid;
        }
    }
    scale[c] = sc;
    shift[c] = sh;
}

}  // namespace

extern "C" int64_t oq_act_stats_workspace(int64_t nsamp, int64_t cols) { return nsamp * SLABS * 2 * cols; }

extern "C" int oq_act_stats(const void* x, int dtype, int64_t nsamp, int64_t rows, int64_t cols, float* scale,
                            float* shift, int64_t seen, float* workspace, int64_t workspace_floats, void* stream) {
    OQ_CHECK_ARG(x && scale && shift && workspace, "oq_act_stats: null pointer");
    OQ_CHECK_ARG(nsamp > 0 && nsamp <= 65535 && rows > 0 && cols > 0 && cols % 8 == 0 && seen >= 0,
                 "oq_act_stats: nsamp=%lld rows=%lld cols=%lld (multiple of 8)", (long long)nsamp, (long long)rows, (long long)cols);
    OQ_CHECK_ARG(oq_aligned16(x), "oq_act_stats: x must be 16-byte aligned");
    OQ_CHECK_ARG(workspace_floats >= nsamp * SLABS * 2 * cols, "oq_act_stats: workspace of %lld floats needed",
                 (long long)(nsamp * SLABS * 2 * cols));
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((cols + 511) / 512), SLABS, (unsigned)nsamp);
    if (dtype == OQ_F32)
        hipLaunchKernelGGL((act_stats_partial_kernel<float>), grid, dim3(256), 0, st, (const float*)x, rows, cols, workspace);
    else if (dtype == OQ_BF16)
        hipLaunchKernelGGL((act_stats_partial_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, rows, cols, workspace);
    else if (dtype == OQ_F16)
        hipLaunchKernelGGL((act_stats_partial_kernel<f16_t>), grid, dim3(256), 0, st, (const f16_t*)x, rows, cols, workspace);
    else {
        oq_set_error("oq_act_stats: dtype %d", dtype);
        return OQ_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(act_stats_final_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, st, workspace, nsamp, cols,
                       seen, scale, shift);
    OQ_CHECK_LAUNCH("oq_act_stats");
    return OQ_OK;
}
