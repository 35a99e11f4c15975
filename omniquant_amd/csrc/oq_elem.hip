// Block glue kernels for gfx950: RoPE, SiLU*up, ReLU, masked softmax (fwd/bwd), fused MSE loss + gradient,
// residual add, scaling, dtype cast, column sums.  All HBM-bound streaming kernels: 16-byte vector accesses,
// grid-stride, f32 math.
// Reference lines: models/int_llama_layer.py:44-45,124-125,153-163; models/int_opt_layer.py:96,151-170,307;
// quantize/omniquant.py:220-222.
#include <cstdlib>

#include "oq_common.h"

namespace {

inline int64_t ew_grid(int64_t nvec, int bt = 256) {
    int64_t g = (nvec + bt - 1) / bt;
    return g < 1 ? 1 : (g > 4096 ? 4096 : g);
}

// ---- RoPE: x [T, heads, hd]; cos/sin [T, hd] ---------------------------------------------------------
// y = x*cos + rot_half(x)*sin ; rot_half(x) = cat(-x[hd/2:], x[:hd/2]).
// inverse (gradient): gx = g*cos + rot_half^T(g*sin), rot_half^T(u) = cat(u[hd/2:], -u[:hd/2]).
template <typename T>
__global__ void rope_kernel(const T* x, T* y, int64_t Tn, int64_t heads, int64_t hd, const float* cs, const float* sn,
                            int inverse) {
    const int64_t half = hd / 2;
    const int64_t nvec = Tn * heads * (half / 8);   // each item handles 8 elements of the first half + partner 8
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d0 = (v % (half / 8)) * 8;
        const int64_t th = v / (half / 8);
        const int64_t tt = th / heads;
        const T* px = x + th * hd;
        T* py = y + th * hd;
        float a[8], b[8], oa[8], ob[8], c1[8], s1[8], c2[8], s2[8];
        Vec8<T>::load(px + d0, a);
        Vec8<T>::load(px + half + d0, b);
        // tables are [T, hd] f32 and L2-resident: four 32-byte vector reads instead of 32 scalar ones
        Vec8<float>::load(cs + tt * hd + d0, c1);
        Vec8<float>::load(sn + tt * hd + d0, s1);
        Vec8<float>::load(cs + tt * hd + half + d0, c2);
        Vec8<float>::load(sn + tt * hd + half + d0, s2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (!inverse) {
                oa[i] = a[i] * c1[i] + (-b[i]) * s1[i];
                ob[i] = b[i] * c2[i] + a[i] * s2[i];
            } else {
                oa[i] = a[i] * c1[i] + b[i] * s2[i];
                ob[i] = b[i] * c2[i] - a[i] * s1[i];
            }
        }
        Vec8<T>::store(py + d0, oa);
        Vec8<T>::store(py + half + d0, ob);
    }
}

template <typename T, int OP>   // 0 silu*up fwd, 1 relu fwd, 2 add, 3 scale
__global__ void ew_fwd_kernel(const T* a, const T* b, T* y, int64_t n, float s) {
    const int64_t nvec = n / 8;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        float x[8], u[8], o[8];
        Vec8<T>::load(a + v * 8, x);
        if (OP == 0 || OP == 2) Vec8<T>::load(b + v * 8, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) o[i] = (x[i] / (1.f + expf(-x[i]))) * u[i];
            if (OP == 1) o[i] = x[i] > 0.f ? x[i] : 0.f;
            if (OP == 2) o[i] = x[i] + u[i];
            if (OP == 3) o[i] = x[i] * s;
        }
        Vec8<T>::store(y + v * 8, o);
    }
}

template <typename T>
__global__ void silu_mul_bwd_kernel(const T* gate, const T* up, const T* gy, T* gg, T* gu, int64_t n) {
    const int64_t nvec = n / 8;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        float x[8], u[8], g[8], og[8], ou[8];
        Vec8<T>::load(gate + v * 8, x);
        Vec8<T>::load(up + v * 8, u);
        Vec8<T>::load(gy + v * 8, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sg = 1.f / (1.f + expf(-x[i]));
            const float silu = x[i] * sg;
            ou[i] = g[i] * silu;
            og[i] = g[i] * u[i] * (sg * (1.f + x[i] * (1.f - sg)));
        }
        Vec8<T>::store(gg + v * 8, og);
        Vec8<T>::store(gu + v * 8, ou);
    }
}

// silu(gate) * up and its backward on [rows, cols] problems whose gate / up (ggate / gup) rows lie `ld` elements apart: the two
// column blocks of the buffer a stacked gate/up GEMM writes (resp. the stacked dgrad / wgrad GEMMs read); y / gy are dense.
template <typename T>
__global__ void silu_mul_fwd_2d_kernel(const T* gate, const T* up, T* y, int64_t rows, int64_t cols, int64_t ld) {
    const int64_t cv = cols / 8, nvec = rows * cv;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = v / cv, c = (v - r * cv) * 8;
        float x[8], u[8], o[8];
        Vec8<T>::load(gate + r * ld + c, x);
        Vec8<T>::load(up + r * ld + c, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (x[i] / (1.f + expf(-x[i]))) * u[i];
        Vec8<T>::store(y + r * cols + c, o);
    }
}

template <typename T>
__global__ void silu_mul_bwd_2d_kernel(const T* gate, const T* up, const T* gy, T* gg, T* gu, int64_t rows, int64_t cols, int64_t ld) {
    const int64_t cv = cols / 8, nvec = rows * cv;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = v / cv, c = (v - r * cv) * 8;
        float x[8], u[8], g[8], og[8], ou[8];
        Vec8<T>::load(gate + r * ld + c, x);
        Vec8<T>::load(up + r * ld + c, u);
        Vec8<T>::load(gy + r * cols + c, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sg = 1.f / (1.f + expf(-x[i]));
            const float silu = x[i] * sg;
            ou[i] = g[i] * silu;
            og[i] = g[i] * u[i] * (sg * (1.f + x[i] * (1.f - sg)));
        }
        Vec8<T>::store(gg + r * ld + c, og);
        Vec8<T>::store(gu + r * ld + c, ou);
    }
}

template <typename T>
__global__ void relu_bwd_kernel(const T* x, const T* gy, T* gx, int64_t n) {
    const int64_t nvec = n / 8;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        float a[8], g[8], o[8];
        Vec8<T>::load(x + v * 8, a);
        Vec8<T>::load(gy + v * 8, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = a[i] > 0.f ? g[i] : 0.f;
        Vec8<T>::store(gx + v * 8, o);
    }
}

// ---- masked softmax: one wave per row, row in registers when cols <= 64*8*SM_CH ------------------------
constexpr int SM_CH = 8;   // 8 chunks * 8 elems * 64 lanes = 4096 columns max per wave-row

template <typename T>
__global__ void __launch_bounds__(256) softmax_fwd_kernel(const T* s, T* p, int64_t rows, int64_t cols, float alpha,
                                                          const float* mask, int64_t mask_rows, int causal) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float lowest = -3.4028234663852886e38f;
    for (int64_t r = wave; r < rows; r += nwaves) {
        float v[SM_CH][8];
        float mx = -INFINITY;
        const float* mrow = mask ? mask + (r % mask_rows) * cols : nullptr;
        // causal: this row attends to columns <= tq; zeros are written up to the end of its 256-column block and
        // nothing beyond that is read or written (the causal GEMM modes never touch it)
        // tq is wave-uniform (one row per wave): chunk-level skipping is a scalar branch, lanes beyond the diagonal
        // inside a live chunk load (valid, possibly unwritten) memory and are masked by a select -> no divergence
        const int64_t tq = causal ? (int64_t)__builtin_amdgcn_readfirstlane((int)(r % cols)) : cols - 1;
        const int64_t zend = causal ? (((tq >> 8) + 1) << 8 < cols ? ((tq >> 8) + 1) << 8 : cols) : cols;
        const int nch = (int)(tq >> 9) + 1;          // 512 columns per wave chunk
#pragma unroll
        for (int j = 0; j < SM_CH; ++j) {
            const int64_t c0 = ((int64_t)j * 64 + lane) * 8;
            if (j < nch && c0 < cols) {
                Vec8<T>::load(s + r * cols + c0, v[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float z = v[j][i] * alpha;
                    if (mrow) z = fmaxf(z + mrow[c0 + i], lowest);
                    z = (c0 + i > tq) ? -INFINITY : z;
                    v[j][i] = z;
                    mx = fmaxf(mx, z);
                }
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < SM_CH; ++j) {
            const int64_t c0 = ((int64_t)j * 64 + lane) * 8;
            if (j < nch && c0 < cols) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[j][i] = expf(v[j][i] - mx); sum += v[j][i]; }
            }
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int j = 0; j < SM_CH; ++j) {
            const int64_t c0 = ((int64_t)j * 64 + lane) * 8;
            if (c0 < zend) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (j < nch) ? v[j][i] * inv : 0.f;    // exp(-inf) = 0 beyond tq
                Vec8<T>::store(p + r * cols + c0, o);
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) softmax_bwd_kernel(const T* p, const T* gp, T* gs, int64_t rows, int64_t cols,
                                                          float alpha, int causal) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        float pv[SM_CH][8], gv[SM_CH][8];
        float dot = 0.f;
        const int64_t tq = causal ? (int64_t)__builtin_amdgcn_readfirstlane((int)(r % cols)) : cols - 1;
        const int64_t zend = causal ? (((tq >> 8) + 1) << 8 < cols ? ((tq >> 8) + 1) << 8 : cols) : cols;
        const int nch = (int)(tq >> 9) + 1;
#pragma unroll
        for (int j = 0; j < SM_CH; ++j) {
            const int64_t c0 = ((int64_t)j * 64 + lane) * 8;
            if (j < nch && c0 < cols) {
                Vec8<T>::load(p + r * cols + c0, pv[j]);
                Vec8<T>::load(gp + r * cols + c0, gv[j]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool dead = c0 + i > tq;                  // dP beyond the diagonal may be unwritten memory
                    pv[j][i] = dead ? 0.f : pv[j][i];
                    gv[j][i] = dead ? 0.f : gv[j][i];
                    dot += pv[j][i] * gv[j][i];
                }
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int j = 0; j < SM_CH; ++j) {
            const int64_t c0 = ((int64_t)j * 64 + lane) * 8;
            if (c0 < zend) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (j < nch) ? pv[j][i] * (gv[j][i] - dot) * alpha : 0.f;
                Vec8<T>::store(gs + r * cols + c0, o);
            }
        }
    }
}

// ---- MSE loss + gradient -------------------------------------------------------------------------------
// TO: dtype of `out` (float32 when the block's output arrives through its un-rounded side channel)
template <typename T, typename TO>
__global__ void __launch_bounds__(256) mse_kernel(const TO* out, const T* t1, const T* t2, int64_t n, float gscale,
                                                  float* part, T* g) {
    __shared__ float red[4];
    const int64_t nvec = n / 8;
    const float inv_n = 1.f / (float)n;
    float acc = 0.f;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        float o[8], a[8], b[8], gg[8];
        Vec8<TO>::load(out + v * 8, o);
        Vec8<T>::load(t1 + v * 8, a);
        if (t2) Vec8<T>::load(t2 + v * 8, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float d = o[i] - a[i];
            acc += d * d;
            float gr = 2.f * d;
            if (t2) { const float d2 = o[i] - b[i]; acc += d2 * d2; gr += 2.f * d2; }
            gg[i] = gr * inv_n * gscale;
        }
        Vec8<T>::store(g + v * 8, gg);
    }
    acc = wave_sum(acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) red[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * inv_n;     // no atomics: fixed-order sum below
}

__global__ void __launch_bounds__(64) mse_final_kernel(const float* part, int nblocks, float* loss) {
    float a = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) a += part[i];
    a = wave_sum(a);
    if (threadIdx.x == 0) loss[0] = a;
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* x, TD* y, int64_t n) {
    const int64_t nvec = n / 8;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        float a[8];
        Vec8<TS>::load(x + v * 8, a);
        Vec8<TD>::store(y + v * 8, a);
    }
}


// column sums (bias gradients), deterministic: grid (column groups of 512, row slabs); a lane owns 8 columns (16-byte
// loads), the 4 waves of a block interleave the slab's rows and combine through LDS; slab partials go to the workspace
// and colsum_final_kernel adds them in slab order (no atomics: the order of a float sum must not depend on timing).
constexpr int COLSUM_MAX_SLABS = 128;

template <typename T>
__global__ void __launch_bounds__(256) colsum_partial_kernel(const T* x, int64_t rows, int64_t cols, float* ws) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * 512 + lane * 8;
    const int64_t r0 = rows * blockIdx.y / gridDim.y, r1 = rows * (blockIdx.y + 1) / gridDim.y;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    if (c0 < cols) {
        for (int64_t r = r0 + wid; r < r1; r += 4) {
            float v[8];
            Vec8<T>::load(x + r * cols + c0, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wid][lane * 8 + i] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int64_t col = (int64_t)blockIdx.x * 512 + c;
        if (col < cols) ws[(int64_t)blockIdx.y * cols + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// One-launch variant for the row counts of a calibration sample (<= 8192 rows): a workgroup owns 64 columns and walks ALL
// rows (thread = 8 columns x every 32nd row, four loads in flight), then folds its 32 row phases through LDS in a fixed
// order.  64-172 workgroups stream 17-45 MB in ~5 us -- no faster than the slab kernel, but the second launch (another
// ~5 us inside the hipGraph) is gone.
template <typename T>
__global__ void __launch_bounds__(256) colsum_direct_kernel(const T* x, int64_t rows, int64_t cols, float* out) {
    __shared__ float red[32][65];
    const int cch = threadIdx.x & 7, ro = threadIdx.x >> 3;
    const int64_t c0 = (int64_t)blockIdx.x * 64 + cch * 8;
    float acc[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[u][i] = 0.f;
    if (c0 < cols) {
        int64_t r = ro;
        for (; r + 96 < rows; r += 128) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) Vec8<T>::load(x + (r + 32 * u) * cols + c0, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[u][i] += v[u][i];
        }
        for (; r < rows; r += 32) {
            float v[8];
            Vec8<T>::load(x + r * cols + c0, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[0][i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[ro][cch * 8 + i] = (acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i]);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int64_t col = (int64_t)blockIdx.x * 64 + threadIdx.x;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k += 4) {
            a0 += red[k][threadIdx.x];
            a1 += red[k + 1][threadIdx.x];
            a2 += red[k + 2][threadIdx.x];
            a3 += red[k + 3][threadIdx.x];
        }
        if (col < cols) out[col] = (a0 + a1) + (a2 + a3);
    }
}

// out[c] = sum_k ws[k][c] in a fixed order: block = 64 columns x 4 slab groups; a thread adds the slabs k = sg, sg+4, ...
// of its column with four independent accumulators (loads in flight), the groups are combined through LDS.
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* ws, int slabs, int64_t cols, float* out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        int k = sg;
        for (; k + 12 < slabs; k += 16) {
            a0 += ws[(int64_t)k * cols + c];
            a1 += ws[(int64_t)(k + 4) * cols + c];
            a2 += ws[(int64_t)(k + 8) * cols + c];
            a3 += ws[(int64_t)(k + 12) * cols + c];
        }
        for (; k < slabs; k += 4) a0 += ws[(int64_t)k * cols + c];
    }
    red[sg][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (sg == 0 && c < cols) out[c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

}  // namespace

#define EW_CHECK(fn, n) OQ_CHECK_ARG((n) > 0 && (n) % 8 == 0, fn ": n=%lld must be a positive multiple of 8", (long long)(n))
#define DT_SWITCH(fn, dtype, CALL_F32, CALL_BF16)                 \
    do {                                                          \
        if ((dtype) == OQ_F32) { CALL_F32; }                      \
        else if ((dtype) == OQ_BF16) { CALL_BF16; }               \
        else { oq_set_error(fn ": dtype %d unsupported", dtype); return OQ_E_UNSUPPORTED; } \
        OQ_CHECK_LAUNCH(fn);                                      \
        return OQ_OK;                                             \
    } while (0)

extern "C" int oq_rope(const void* x, void* y, int dtype, int64_t T, int64_t heads, int64_t hd, const float* cos,
                       const float* sin, int inverse, void* stream) {
    OQ_CHECK_ARG(x && y && cos && sin, "oq_rope: null pointer");
    OQ_CHECK_ARG(oq_aligned16(cos) && oq_aligned16(sin), "oq_rope: cos/sin tables must be 16-byte aligned");
    OQ_CHECK_ARG(T > 0 && heads > 0 && hd > 0 && hd % 16 == 0, "oq_rope: head_dim %lld must be a multiple of 16", (long long)hd);
    const int64_t nvec = T * heads * (hd / 16);
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH("oq_rope", dtype,
              hipLaunchKernelGGL((rope_kernel<float>), dim3(ew_grid(nvec)), dim3(256), 0, st, (const float*)x, (float*)y, T, heads, hd, cos, sin, inverse),
              hipLaunchKernelGGL((rope_kernel<bf16_t>), dim3(ew_grid(nvec)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, T, heads, hd, cos, sin, inverse));
}

#define EW_FWD(NAME, OP, A, B, S)                                                                                          \
    hipStream_t st = (hipStream_t)stream;                                                                                  \
    DT_SWITCH(NAME, dtype,                                                                                                 \
              hipLaunchKernelGGL((ew_fwd_kernel<float, OP>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const float*)A, (const float*)B, (float*)y, n, S), \
              hipLaunchKernelGGL((ew_fwd_kernel<bf16_t, OP>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)y, n, S))

extern "C" int oq_silu_mul_fwd(const void* gate, const void* up, void* y, int dtype, int64_t n, void* stream) {
    EW_CHECK("oq_silu_mul_fwd", n);
    EW_FWD("oq_silu_mul_fwd", 0, gate, up, 0.f);
}
extern "C" int oq_relu_fwd(const void* x, void* y, int dtype, int64_t n, void* stream) {
    EW_CHECK("oq_relu_fwd", n);
    EW_FWD("oq_relu_fwd", 1, x, x, 0.f);
}
extern "C" int oq_add(const void* a, const void* b, void* y, int dtype, int64_t n, void* stream) {
    EW_CHECK("oq_add", n);
    EW_FWD("oq_add", 2, a, b, 0.f);
}
extern "C" int oq_scale(const void* a, float s, void* y, int dtype, int64_t n, void* stream) {
    EW_CHECK("oq_scale", n);
    EW_FWD("oq_scale", 3, a, a, s);
}

// ---- grouped-query attention: dK / dV of a key-value head = sum over the `rep` query heads that share it ------------------
// (repeat_kv's backward, models/int_llama_layer.py:136-141).  x [rows, nkv, rep, hd] -> y [rows, nkv, hd], fp32 sum in head
// order, one rounding; up to two tensors (dK and dV) in one launch.
namespace {
template <typename T>
__global__ void __launch_bounds__(256) group_sum_kernel(const T* x0, T* y0, const T* x1, T* y1, int64_t nvec_out, int rep, int hd8) {
    const T* x = blockIdx.y ? x1 : x0;
    T* y = blockIdx.y ? y1 : y0;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec_out; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = v / hd8, c = v - g * hd8;            // g = (row, kv head), c = 8-element chunk inside the head
        float acc[8];
        Vec8<T>::load(x + ((g * rep) * hd8 + c) * 8, acc);
        for (int r = 1; r < rep; ++r) {
            float t[8];
            Vec8<T>::load(x + ((g * rep + r) * hd8 + c) * 8, t);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += t[i];
        }
        Vec8<T>::store(y + v * 8, acc);
    }
}
}  // namespace

extern "C" int oq_group_sum(const void* x0, void* y0, const void* x1, void* y1, int dtype, int64_t groups, int rep, int64_t hd,
                            void* stream) {
    OQ_CHECK_ARG(x0 && y0 && ((x1 == nullptr) == (y1 == nullptr)), "oq_group_sum: null pointer");
    OQ_CHECK_ARG(groups > 0 && rep >= 1 && hd > 0 && hd % 8 == 0, "oq_group_sum: groups %lld rep %d hd %lld (hd multiple of 8)",
                 (long long)groups, rep, (long long)hd);
    OQ_CHECK_ARG(oq_aligned16(x0) && oq_aligned16(y0) && oq_aligned16(x1) && oq_aligned16(y1), "oq_group_sum: 16-byte alignment");
    const int hd8 = (int)(hd / 8);
    const int64_t nvec = groups * hd8;
    const dim3 grid((unsigned)((nvec + 255) / 256 < 4096 ? (nvec + 255) / 256 : 4096), x1 ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == OQ_BF16)
        hipLaunchKernelGGL((group_sum_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x0, (bf16_t*)y0, (const bf16_t*)x1, (bf16_t*)y1, nvec, rep, hd8);
    else if (dtype == OQ_F32)
        hipLaunchKernelGGL((group_sum_kernel<float>), grid, dim3(256), 0, st, (const float*)x0, (float*)y0, (const float*)x1, (float*)y1, nvec, rep, hd8);
    else {
        oq_set_error("oq_group_sum: dtype %d unsupported", dtype);
        return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_group_sum");
    return OQ_OK;
}

// ---- the step's input sample(s) -> the static buffers a replayed hipGraph reads: up to three 16-byte-granular copies in
// ONE launch (quantize/omniquant.py:216-219 indexes the banks in place; a captured graph needs fixed addresses)
namespace {
struct Copy3 {
    const u32x4* src[3];
    u32x4* dst[3];
    int64_t nvec;       // 16-byte vectors per pair
    int n;
};
__global__ void __launch_bounds__(256) copy_samples_kernel(Copy3 c) {
    const u32x4* s = c.src[blockIdx.y];
    u32x4* d = c.dst[blockIdx.y];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < c.nvec; i += 4 * stride) {          // four independent 16-byte loads in flight per thread
        const u32x4 a = s[i], b = s[i + stride], e = s[i + 2 * stride], f = s[i + 3 * stride];
        d[i] = a; d[i + stride] = b; d[i + 2 * stride] = e; d[i + 3 * stride] = f;
    }
    for (; i < c.nvec; i += stride) d[i] = s[i];
}
}  // namespace

extern "C" int oq_copy_samples(int n, const void* src0, void* dst0, const void* src1, void* dst1, const void* src2, void* dst2,
                               int64_t bytes_each, void* stream) {
    OQ_CHECK_ARG(n >= 1 && n <= 3 && bytes_each > 0 && bytes_each % 16 == 0, "oq_copy_samples: 1..3 pairs of a multiple of 16 bytes");
    Copy3 c{};
    const void* s[3] = {src0, src1, src2};
    void* d[3] = {dst0, dst1, dst2};
    for (int i = 0; i < n; ++i) {
        OQ_CHECK_ARG(s[i] && d[i] && oq_aligned16(s[i]) && oq_aligned16(d[i]), "oq_copy_samples: null or unaligned buffer %d", i);
        c.src[i] = (const u32x4*)s[i];
        c.dst[i] = (u32x4*)d[i];
    }
    c.nvec = bytes_each / 16;
    c.n = n;
    int64_t nb = (c.nvec + 4 * 256 - 1) / (4 * 256);
    nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
    hipLaunchKernelGGL(copy_samples_kernel, dim3((unsigned)nb, (unsigned)n), dim3(256), 0, (hipStream_t)stream, c);
    OQ_CHECK_LAUNCH("oq_copy_samples");
    return OQ_OK;
}

extern "C" int oq_silu_mul_bwd(const void* gate, const void* up, const void* gy, void* ggate, void* gup, int dtype,
                               int64_t n, void* stream) {
    EW_CHECK("oq_silu_mul_bwd", n);
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH("oq_silu_mul_bwd", dtype,
              hipLaunchKernelGGL((silu_mul_bwd_kernel<float>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const float*)gate, (const float*)up, (const float*)gy, (float*)ggate, (float*)gup, n),
              hipLaunchKernelGGL((silu_mul_bwd_kernel<bf16_t>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)gate, (const bf16_t*)up, (const bf16_t*)gy, (bf16_t*)ggate, (bf16_t*)gup, n));
}

extern "C" int oq_silu_mul_fwd_2d(const void* gate, const void* up, void* y, int dtype, int64_t rows, int64_t cols, int64_t ld,
                                  void* stream) {
    OQ_CHECK_ARG(gate && up && y, "oq_silu_mul_fwd_2d: null pointer");
    OQ_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && ld >= cols && ld % 8 == 0,
                 "oq_silu_mul_fwd_2d: rows %lld, cols %lld, ld %lld (multiples of 8, ld >= cols)", (long long)rows, (long long)cols, (long long)ld);
    OQ_CHECK_ARG(oq_aligned16(gate) && oq_aligned16(up) && oq_aligned16(y), "oq_silu_mul_fwd_2d: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nv = rows * cols / 8;
    DT_SWITCH("oq_silu_mul_fwd_2d", dtype,
              hipLaunchKernelGGL((silu_mul_fwd_2d_kernel<float>), dim3(ew_grid(nv)), dim3(256), 0, st, (const float*)gate, (const float*)up, (float*)y, rows, cols, ld),
              hipLaunchKernelGGL((silu_mul_fwd_2d_kernel<bf16_t>), dim3(ew_grid(nv)), dim3(256), 0, st, (const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)y, rows, cols, ld));
}

extern "C" int oq_silu_mul_bwd_2d(const void* gate, const void* up, const void* gy, void* ggate, void* gup, int dtype,
                                  int64_t rows, int64_t cols, int64_t ld, void* stream) {
    OQ_CHECK_ARG(gate && up && gy && ggate && gup, "oq_silu_mul_bwd_2d: null pointer");
    OQ_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && ld >= cols && ld % 8 == 0,
                 "oq_silu_mul_bwd_2d: rows %lld, cols %lld, ld %lld (multiples of 8, ld >= cols)", (long long)rows, (long long)cols, (long long)ld);
    OQ_CHECK_ARG(oq_aligned16(gate) && oq_aligned16(up) && oq_aligned16(gy) && oq_aligned16(ggate) && oq_aligned16(gup),
                 "oq_silu_mul_bwd_2d: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nv = rows * cols / 8;
    DT_SWITCH("oq_silu_mul_bwd_2d", dtype,
              hipLaunchKernelGGL((silu_mul_bwd_2d_kernel<float>), dim3(ew_grid(nv)), dim3(256), 0, st, (const float*)gate, (const float*)up, (const float*)gy, (float*)ggate, (float*)gup, rows, cols, ld),
              hipLaunchKernelGGL((silu_mul_bwd_2d_kernel<bf16_t>), dim3(ew_grid(nv)), dim3(256), 0, st, (const bf16_t*)gate, (const bf16_t*)up, (const bf16_t*)gy, (bf16_t*)ggate, (bf16_t*)gup, rows, cols, ld));
}

extern "C" int oq_relu_bwd(const void* x, const void* gy, void* gx, int dtype, int64_t n, void* stream) {
    EW_CHECK("oq_relu_bwd", n);
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH("oq_relu_bwd", dtype,
              hipLaunchKernelGGL((relu_bwd_kernel<float>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const float*)x, (const float*)gy, (float*)gx, n),
              hipLaunchKernelGGL((relu_bwd_kernel<bf16_t>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)gy, (bf16_t*)gx, n));
}

extern "C" int oq_softmax_fwd(const void* s, void* p, int dtype, int64_t rows, int64_t cols, float alpha,
                              const float* mask, int64_t mask_rows, int causal, void* stream) {
    OQ_CHECK_ARG(!causal || (!mask && rows % cols == 0), "oq_softmax_fwd: causal needs mask == NULL and square [T,T] problems");
    OQ_CHECK_ARG(s && p, "oq_softmax_fwd: null pointer");
    OQ_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 64 * 8 * SM_CH, "oq_softmax_fwd: cols %lld (multiple of 8, <= %d)", (long long)cols, 64 * 8 * SM_CH);
    OQ_CHECK_ARG(!mask || mask_rows > 0, "oq_softmax_fwd: mask_rows");
    const int64_t grid = (rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192;
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH("oq_softmax_fwd", dtype,
              hipLaunchKernelGGL((softmax_fwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)s, (float*)p, rows, cols, alpha, mask, mask_rows, causal),
              hipLaunchKernelGGL((softmax_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)s, (bf16_t*)p, rows, cols, alpha, mask, mask_rows, causal));
}

extern "C" int oq_softmax_bwd(const void* p, const void* gp, void* gs, int dtype, int64_t rows, int64_t cols,
                              float alpha, int causal, void* stream) {
    OQ_CHECK_ARG(!causal || rows % cols == 0, "oq_softmax_bwd: causal needs square [T,T] problems");
    OQ_CHECK_ARG(p && gp && gs, "oq_softmax_bwd: null pointer");
    OQ_CHECK_ARG(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 64 * 8 * SM_CH, "oq_softmax_bwd: cols %lld", (long long)cols);
    const int64_t grid = (rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192;
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH("oq_softmax_bwd", dtype,
              hipLaunchKernelGGL((softmax_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)p, (const float*)gp, (float*)gs, rows, cols, alpha, causal),
              hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)p, (const bf16_t*)gp, (bf16_t*)gs, rows, cols, alpha, causal));
}

extern "C" int oq_mse_fwd_bwd(const void* out, int out_dtype, const void* t1, const void* t2, int dtype, int64_t n, float gscale,
                              float* loss, void* g, void* stream) {
    EW_CHECK("oq_mse_fwd_bwd", n);
    OQ_CHECK_ARG(out && t1 && loss && g, "oq_mse_fwd_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t grid = ew_grid(n / 8) > OQ_MSE_MAX_BLOCKS ? OQ_MSE_MAX_BLOCKS : ew_grid(n / 8);
    if (dtype == OQ_F32 && out_dtype == OQ_F32)
        hipLaunchKernelGGL((mse_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)out, (const float*)t1, (const float*)t2, n, gscale, loss + 1, (float*)g);
    else if (dtype == OQ_BF16 && out_dtype == OQ_BF16)
        hipLaunchKernelGGL((mse_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)out, (const bf16_t*)t1, (const bf16_t*)t2, n, gscale, loss + 1, (bf16_t*)g);
    else if (dtype == OQ_BF16 && out_dtype == OQ_F32)
        hipLaunchKernelGGL((mse_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, st, (const float*)out, (const bf16_t*)t1, (const bf16_t*)t2, n, gscale, loss + 1, (bf16_t*)g);
    else {
        oq_set_error("oq_mse_fwd_bwd: dtypes %d (out %d) unsupported", dtype, out_dtype);
        return OQ_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, st, loss + 1, (int)grid, loss);
    OQ_CHECK_LAUNCH("oq_mse_fwd_bwd");
    return OQ_OK;
}

extern "C" int oq_cast(const void* x, int src_dtype, void* y, int dst_dtype, int64_t n, void* stream) {
    EW_CHECK("oq_cast", n);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ew_grid(n / 8)), blk(256);
    const int key = src_dtype * 3 + dst_dtype;
    switch (key) {
        case OQ_F32 * 3 + OQ_BF16: hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, blk, 0, st, (const float*)x, (bf16_t*)y, n); break;
        case OQ_F32 * 3 + OQ_F16: hipLaunchKernelGGL((cast_kernel<float, f16_t>), grid, blk, 0, st, (const float*)x, (f16_t*)y, n); break;
        case OQ_BF16 * 3 + OQ_F32: hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, blk, 0, st, (const bf16_t*)x, (float*)y, n); break;
        case OQ_F16 * 3 + OQ_F32: hipLaunchKernelGGL((cast_kernel<f16_t, float>), grid, blk, 0, st, (const f16_t*)x, (float*)y, n); break;
        case OQ_F16 * 3 + OQ_BF16: hipLaunchKernelGGL((cast_kernel<f16_t, bf16_t>), grid, blk, 0, st, (const f16_t*)x, (bf16_t*)y, n); break;
        case OQ_BF16 * 3 + OQ_F16: hipLaunchKernelGGL((cast_kernel<bf16_t, f16_t>), grid, blk, 0, st, (const bf16_t*)x, (f16_t*)y, n); break;
        default: oq_set_error("oq_cast: %d -> %d unsupported", src_dtype, dst_dtype); return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_cast");
    return OQ_OK;
}

static bool colsum_direct_enabled() {          // OQ_COLSUM_DIRECT=0: the two-launch slab kernels (A/B)
    const char* v = getenv("OQ_COLSUM_DIRECT");
    return !(v && v[0] == '0');
}

static int colsum_slabs(int64_t rows) {
    int64_t s = rows / 16;          // many small slabs: the partial kernel is latency-bound below ~2 workgroups per CU
    return (int)(s < 1 ? 1 : (s > COLSUM_MAX_SLABS ? COLSUM_MAX_SLABS : s));
}

extern "C" int64_t oq_colsum_workspace(int64_t rows, int64_t cols) { return (int64_t)colsum_slabs(rows) * cols; }

extern "C" int oq_colsum(const void* x, int dtype, int64_t rows, int64_t cols, float* out, float* workspace,
                         int64_t workspace_floats, void* stream) {
    OQ_CHECK_ARG(x && out && workspace && rows > 0 && cols > 0 && cols % 8 == 0, "oq_colsum: bad args (cols %% 8 == 0)");
    OQ_CHECK_ARG(oq_aligned16(x), "oq_colsum: x must be 16-byte aligned");
    const int slabs = colsum_slabs(rows);
    OQ_CHECK_ARG(workspace_floats >= (int64_t)slabs * cols, "oq_colsum: workspace of %lld floats needed",
                 (long long)((int64_t)slabs * cols));
    hipStream_t st = (hipStream_t)stream;
    if (rows <= 8192 && cols >= 2048 && colsum_direct_enabled()) {
        const dim3 gd((unsigned)((cols + 63) / 64));
        if (dtype == OQ_F32) hipLaunchKernelGGL((colsum_direct_kernel<float>), gd, dim3(256), 0, st, (const float*)x, rows, cols, out);
        else if (dtype == OQ_BF16) hipLaunchKernelGGL((colsum_direct_kernel<bf16_t>), gd, dim3(256), 0, st, (const bf16_t*)x, rows, cols, out);
        else {
            oq_set_error("oq_colsum: dtype %d unsupported", dtype);
            return OQ_E_UNSUPPORTED;
        }
        OQ_CHECK_LAUNCH("oq_colsum");
        return OQ_OK;
    }
    const dim3 grid((unsigned)((cols + 511) / 512), (unsigned)slabs);
    if (dtype == OQ_F32)
        hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, dim3(256), 0, st, (const float*)x, rows, cols, workspace);
    else if (dtype == OQ_BF16)
        hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, rows, cols, workspace);
    else {
        oq_set_error("oq_colsum: dtype %d unsupported", dtype);
        return OQ_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(256), 0, st, workspace, slabs, cols, out);
    OQ_CHECK_LAUNCH("oq_colsum");
    return OQ_OK;
}
