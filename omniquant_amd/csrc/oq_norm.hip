// OmniLlamaRMSNorm / OmniLayerNorm forward + backward for gfx950 (quantize/omni_norm.py:26-34,:52-63).
// One workgroup per row (row kept in registers, <= 4 chunks of 8 per lane), f32 statistics.
// Backward column sums (gw, gb) are accumulated in registers over the rows a workgroup walks, written as one partial
// row per workgroup and finished by norm_colreduce_kernel in a fixed order (no atomics).
#include "oq_common.h"

namespace {
constexpr int MAXCH = 4;

__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    if (nw == 1) return v;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float a = red[0];
    for (int k = 1; k < nw; ++k) a += red[k];
    return a;
}

template <typename T, int NCH>
__global__ void __launch_bounds__(1024) norm_fwd_kernel(const T* x, int64_t rows, int64_t cols, const float* w,
                                                        const float* b, float eps, int ln, T* y, float* rstd_out,
                                                        float* mean_out) {
    __shared__ float red[16];
    const int t = threadIdx.x, BT = blockDim.x;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        float v[NCH][8];
        bool valid[NCH];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int64_t c0 = ((int64_t)j * BT + t) * 8;
            valid[j] = c0 < cols;
            if (valid[j]) {
                Vec8<T>::load(x + r * cols + c0, v[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) s += ln ? v[j][i] : v[j][i] * v[j][i];
            }
        }
        s = block_sum(s, red);
        float mean = 0.f, var;
        if (ln) {
            mean = s / (float)cols;
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                if (valid[j]) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float d = v[j][i] - mean; s2 += d * d; }
                }
            var = block_sum(s2, red) / (float)cols;
        } else {
            var = s / (float)cols;
        }
        const float rstd = rsqrtf(var + eps);
        if (t == 0) {
            rstd_out[r] = rstd;
            if (mean_out) mean_out[r] = mean;
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (!valid[j]) continue;
            const int64_t c0 = ((int64_t)j * BT + t) * 8;
            float o[8], wv[8], bv[8];
            Vec8<float>::load(w + c0, wv);               // 32-byte vector reads (were 16 scalar loads per chunk)
            if (b) {
                Vec8<float>::load(b + c0, bv);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (v[j][i] - mean) * rstd;
                o[i] = wv[i] * xh + bv[i];
            }
            Vec8<T>::store(y + r * cols + c0, o);
        }
    }
}

template <typename T, int NCH>
__global__ void __launch_bounds__(512) norm_bwd_kernel(const T* x, const T* gy, int64_t rows, int64_t cols,
                                                       const float* w, const float* rstd_in, const float* mean_in,
                                                       int ln, T* gx, float* ws, int want_b, const T* gx_add) {
    __shared__ float red[16];
    const int t = threadIdx.x, BT = blockDim.x;
    float aw[NCH][8], ab[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) { aw[j][i] = 0.f; ab[j][i] = 0.f; }
    float wv[NCH][8];                       // the norm weight of this thread's columns: loaded once, as vectors
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int64_t c0 = ((int64_t)j * BT + t) * 8;
        if (c0 < cols) {
            Vec8<float>::load(w + c0, wv[j]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[j][i] = 0.f;
        }
    }
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float rstd = rstd_in[r];
        const float mean = (ln && mean_in) ? mean_in[r] : 0.f;
        float xh[NCH][8], gh[NCH][8];   // normalised x, and gy*w
        bool valid[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int64_t c0 = ((int64_t)j * BT + t) * 8;
            valid[j] = c0 < cols;
            if (valid[j]) {
                float g[8];
                Vec8<T>::load(x + r * cols + c0, xh[j]);
                Vec8<T>::load(gy + r * cols + c0, g);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xh[j][i] = (xh[j][i] - mean) * rstd;
                    aw[j][i] += g[i] * xh[j][i];
                    ab[j][i] += g[i];
                    gh[j][i] = g[i] * wv[j][i];
                    s1 += gh[j][i];
                    s2 += gh[j][i] * xh[j][i];
                }
            }
        }
        s2 = block_sum(s2, red) / (float)cols;
        s1 = ln ? block_sum(s1, red) / (float)cols : 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (!valid[j]) continue;
            const int64_t c0 = ((int64_t)j * BT + t) * 8;
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (gh[j][i] - s1 - xh[j][i] * s2) * rstd;
            if (gx_add) {          // gradient that reached x on the residual path: added here instead of by a separate launch
                float ga[8];
                Vec8<T>::load(gx_add + r * cols + c0, ga);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] += ga[i];
            }
            Vec8<T>::store(gx + r * cols + c0, o);
        }
    }
    float* pw = ws + (int64_t)blockIdx.x * cols;
    float* pb = ws + ((int64_t)gridDim.x + blockIdx.x) * cols;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int64_t c0 = ((int64_t)j * BT + t) * 8;
        if (c0 < cols) {
            Vec8<float>::store(pw + c0, aw[j]);
            if (want_b) Vec8<float>::store(pb + c0, ab[j]);
        }
    }
}

__global__ void __launch_bounds__(256) norm_colreduce_kernel(const float* part, int nblocks, int64_t cols, float* out0,
                                                             float* out1) {
    const float* src = part + (int64_t)blockIdx.y * nblocks * cols;
    float* out = blockIdx.y == 0 ? out0 : out1;
    if (!out) return;
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int64_t c = (int64_t)blockIdx.x * 16 + cl;
    __shared__ float red[16][17];
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    if (c < cols) {          // 32 independent loads in flight per lane (see colreduce_kernel in oq_quant.hip)
        for (int base = 0; base < nblocks; base += 512) {
            float v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int b = base + rg + 16 * k;
                v[k] = b < nblocks ? src[(int64_t)b * cols + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) a[k & 7] += v[k];
        }
    }
    red[rg][cl] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (rg == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        out[c] = s;
    }
}

constexpr int64_t NORM_BWD_BLOCKS = 512;

int norm_threads(int64_t cols) {
    const int64_t lanes = (cols + 7) / 8;
    if (lanes <= 64) return 64;
    if (lanes <= 128) return 128;
    if (lanes <= 256) return 256;
    if (cols <= 8 * 512 * MAXCH) return 512;
    return 1024;
}
}  // namespace

extern "C" int oq_norm_fwd(const void* x, int dtype, int64_t rows, int64_t cols, const float* w, const float* b,
                           float eps, int is_layernorm, void* y, float* rstd, float* mean, void* stream) {
    OQ_CHECK_ARG(x && y && w && rstd, "oq_norm_fwd: null pointer");
    OQ_CHECK_ARG(oq_aligned16(w) && oq_aligned16(b), "oq_norm_fwd: weight / bias must be 16-byte aligned");
    OQ_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 8 * 1024 * MAXCH, "oq_norm_fwd: cols %lld", (long long)cols);
    OQ_CHECK_ARG(!is_layernorm || mean, "oq_norm_fwd: layernorm needs mean buffer");
    const int bt = norm_threads(cols);
    const int64_t grid = rows < 8192 ? rows : 8192;
    hipStream_t st = (hipStream_t)stream;
    // chunks of 8 elements per thread: 1, 2 or 4 (a template parameter: the row lives in registers, and sizing every
    // instantiation for 4 chunks cost half the occupancy at hidden sizes <= 4096)
    const int64_t per = ((cols + 7) / 8 + bt - 1) / bt;
    const int nch = per <= 1 ? 1 : (per <= 2 ? 2 : 4);
#define NORM_FWD(T_, N_) hipLaunchKernelGGL((norm_fwd_kernel<T_, N_>), dim3(grid), dim3(bt), 0, st, (const T_*)x, rows, cols, w, b, eps, is_layernorm, (T_*)y, rstd, mean)
    if (dtype == OQ_F32) {
        if (nch == 1) NORM_FWD(float, 1); else if (nch == 2) NORM_FWD(float, 2); else NORM_FWD(float, 4);
    } else if (dtype == OQ_BF16) {
        if (nch == 1) NORM_FWD(bf16_t, 1); else if (nch == 2) NORM_FWD(bf16_t, 2); else NORM_FWD(bf16_t, 4);
    } else {
        oq_set_error("oq_norm_fwd: dtype %d", dtype);
        return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_norm_fwd");
    return OQ_OK;
}

extern "C" int oq_norm_bwd(const void* x, const void* gy, int dtype, int64_t rows, int64_t cols, const float* w,
                           const float* rstd, const float* mean, int is_layernorm, void* gx, float* gw, float* gb,
                           const void* gx_addend, float* workspace, int64_t workspace_floats, void* stream) {
    OQ_CHECK_ARG(x && gy && gx && w && rstd, "oq_norm_bwd: null pointer");
    OQ_CHECK_ARG(oq_aligned16(w), "oq_norm_bwd: weight must be 16-byte aligned");
    OQ_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 8 * 512 * MAXCH, "oq_norm_bwd: cols %lld", (long long)cols);
    int bt = norm_threads(cols);
    if (bt > 512) bt = 512;
    const int64_t grid = rows < NORM_BWD_BLOCKS ? rows : NORM_BWD_BLOCKS;
    OQ_CHECK_ARG(gw && workspace && workspace_floats >= 2 * grid * cols, "oq_norm_bwd: gw and a workspace of %lld floats are required",
                 (long long)(2 * grid * cols));
    hipStream_t st = (hipStream_t)stream;
    const int64_t per = ((cols + 7) / 8 + bt - 1) / bt;
    const int nch = per <= 1 ? 1 : (per <= 2 ? 2 : 4);
#define NORM_BWD(T_, N_) hipLaunchKernelGGL((norm_bwd_kernel<T_, N_>), dim3(grid), dim3(bt), 0, st, (const T_*)x, (const T_*)gy, rows, cols, w, rstd, mean, is_layernorm, (T_*)gx, workspace, gb ? 1 : 0, (const T_*)gx_addend)
    if (dtype == OQ_F32) {
        if (nch == 1) NORM_BWD(float, 1); else if (nch == 2) NORM_BWD(float, 2); else NORM_BWD(float, 4);
    } else if (dtype == OQ_BF16) {
        if (nch == 1) NORM_BWD(bf16_t, 1); else if (nch == 2) NORM_BWD(bf16_t, 2); else NORM_BWD(bf16_t, 4);
    } else {
        oq_set_error("oq_norm_bwd: dtype %d", dtype);
        return OQ_E_UNSUPPORTED;
    }
    if (dtype == OQ_F32 || dtype == OQ_BF16) {
        const dim3 rg((unsigned)((cols + 15) / 16), gb ? 2 : 1);
        hipLaunchKernelGGL(norm_colreduce_kernel, rg, dim3(256), 0, st, workspace, (int)grid, cols, gw, gb);
    }
    OQ_CHECK_LAUNCH("oq_norm_bwd");
    return OQ_OK;
}

extern "C" int64_t oq_norm_bwd_workspace(int64_t rows, int64_t cols) {
    const int64_t grid = rows < NORM_BWD_BLOCKS ? rows : NORM_BWD_BLOCKS;
    return 2 * grid * cols;
}
