// Optimiser step for the flat learnable arena of one block (gfx950).
// Replaces utils.py:11-24 (grad norm), utils.py:32-46 (GradScaler-style skip on non-finite grads) and the
// torch.optim.AdamW step configured at quantize/omniquant.py:207-208, plus truncate_number
// (models/transformation.py:5-20).  The reference launches ~100-190 tiny kernels for this; here it is two
// launches over one contiguous f32 buffer, with the step counter and the norm kept on device so the whole
// sample-step replays from a hipGraph without host round trips.
#include "oq_common.h"

namespace {

__global__ void __launch_bounds__(256) gradnorm_partial_kernel(const float* g, int64_t n, float* ws) {
    __shared__ float red[2][4];
    float ss = 0.f, bad = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        ss += v * v;
        if (!(fabsf(v) <= 3.4028234663852886e38f)) bad = 1.f;   // NaN or inf
    }
    ss = wave_sum(ss);
    bad = wave_max(bad);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[0][wid] = ss; red[1][wid] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        ws[2 * blockIdx.x + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    }
}

// deterministic second stage (fixed order)
__global__ void __launch_bounds__(64) gradnorm_final_kernel(const float* ws, int nblocks, float* out) {
    float ss = 0.f, bad = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) { ss += ws[2 * i]; bad = fmaxf(bad, ws[2 * i + 1]); }
    ss = wave_sum(ss);
    bad = wave_max(bad);
    if (threadIdx.x == 0) {
        out[0] = sqrtf(ss);
        out[1] = (bad == 0.f && ss == ss && ss <= 3.4028234663852886e38f) ? 1.f : 0.f;
    }
}

__global__ void __launch_bounds__(256) adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_let,
                                                    float lr_let, float lr_lwc, float b1, float b2, float eps, float wd,
                                                    const float* step_ptr, const float* norm) {
    if (norm && norm[1] == 0.f) return;   // non-finite gradients: skip the step (GradScaler semantics)
    const float step = step_ptr[0] + 1.f;
    const float bc1 = 1.f - powf(b1, step);
    const float bc2 = 1.f - powf(b2, step);
    const float bc2_sqrt = sqrtf(bc2);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float lr = i < n_let ? lr_let : lr_lwc;
        const float gi = g[i];
        float pi = p[i];
        pi = pi * (1.f - lr * wd);
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * (1.f - b1);                 // lerp_, as torch
        vi = vi * b2 + (1.f - b2) * (gi * gi);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

__global__ void step_inc_kernel(float* step_ptr, const float* norm) {
    if (norm && norm[1] == 0.f) return;
    step_ptr[0] += 1.f;
}

__global__ void truncate_kernel(float* x, int64_t n, float thr) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        if (fabsf(v) < thr) x[i] = (v > 0.f ? thr : (v < 0.f ? -thr : 0.f));
    }
}

constexpr int GN_BLOCKS = 256;
}  // namespace

extern "C" int oq_gradnorm(const float* g, int64_t n, float* norm_out, float* workspace, void* stream) {
    OQ_CHECK_ARG(g && norm_out && workspace && n > 0, "oq_gradnorm: bad args (workspace needs %d floats)", 2 * GN_BLOCKS);
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = (n + 255) / 256;
    nb = nb > GN_BLOCKS ? GN_BLOCKS : nb;
    hipLaunchKernelGGL(gradnorm_partial_kernel, dim3(nb), dim3(256), 0, st, g, n, workspace);
    hipLaunchKernelGGL(gradnorm_final_kernel, dim3(1), dim3(64), 0, st, workspace, (int)nb, norm_out);
    OQ_CHECK_LAUNCH("oq_gradnorm");
    return OQ_OK;
}

extern "C" int oq_adamw(float* p, const float* g, float* m, float* v, int64_t n, int64_t n_let, float lr_let,
                        float lr_lwc, float beta1, float beta2, float eps, float wd, float* step_ptr, const float* norm,
                        void* stream) {
    OQ_CHECK_ARG(p && g && m && v && step_ptr && n > 0 && n_let >= 0 && n_let <= n, "oq_adamw: bad args");
    hipStream_t st = (hipStream_t)stream;
    int64_t nb = (n + 255) / 256;
    nb = nb > 2048 ? 2048 : nb;
    hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, st, p, g, m, v, n, n_let, lr_let, lr_lwc, beta1, beta2, eps, wd,
                       (const float*)step_ptr, norm);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, step_ptr, norm);
    OQ_CHECK_LAUNCH("oq_adamw");
    return OQ_OK;
}

extern "C" int oq_truncate(float* x, int64_t n, float thr, void* stream) {
    OQ_CHECK_ARG(x && n > 0, "oq_truncate: bad args");
    int64_t nb = (n + 255) / 256;
    nb = nb > 1024 ? 1024 : nb;
    hipLaunchKernelGGL(truncate_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, thr);
    OQ_CHECK_LAUNCH("oq_truncate");
    return OQ_OK;
}
