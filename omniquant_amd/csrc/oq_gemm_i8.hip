// Integer-exact fprop of the fake-quant Linear for gfx950: Y = dequant(X) @ dequant(W)^T without ever rounding the two
// dequantised operands.
//
// Replaces quantize/int_linear.py:59-62 (per-token fake-quant input x per-output-channel fake-quant weight -> F.linear) when
// both operands come from UniformAffineQuantizer grids (quantize/quantizer.py:84-105): with integer codes q_a, q_w, rounded
// zero-points z_a[t], z_w[n] and scales s_a[t], s_w[n]
//     Y[t][n] = s_a[t] * s_w[n] * sum_k (q_a[t][k] - z_a[t]) * (q_w[n][k] - z_w[n])  + bias[n] (+ addend)
// and the sum is an INTEGER.  The quantiser kernels store the codes as int8 (code - off; off = 0, or 128 for 8-bit grids so
// that they fit) next to their bf16 output; this kernel contracts them on v_mfma_i32_16x16x64_i8 (2x the bf16 rate, half the
// operand bytes per MAC) into int32 accumulators -- exact -- and the epilogue applies, in fp32,
//     sum = acc - za'[t] * Cw[n] - zw'[n] * (Ca[t] - K * za'[t])      za' = z_a - off_a, zw' = z_w - off_w,
//                                                                      Ca[t] = sum_k stored a-codes, Cw[n] likewise
// then the two scales, bias and the residual.  (The bf16 form rounds (q - z) * s of BOTH operands to 8 bits of mantissa first,
// which is what flips ~3 % of the next 4-bit rounding decisions downstream -- DESIGN.md section 4.)
//
// Structure = gemm_bf16_p3_kernel of oq_gemm.hip with both operands k-contiguous: 256x128 output tile, K-tile of 128 BYTES per
// row (128 codes; the same LDS image as 64 bf16), 8 waves x (4x4) MFMA tiles, LDS-DMA staging (global_load_lds, 16 B per
// lane, swizzle on the source address), 3-stage ring with a counted vmcnt, staggered wave groups, epilogue through LDS in
// whole 128-byte lines, strip-ordered XCD-aware tile ids.  Since A and B fragments are read with the SAME k permutation
// (16 consecutive bytes per lane), any consistent lane->k map of the instruction gives the exact dot product.
#include <stdlib.h>
#include "oq_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;

struct GemmI8P {
    const int8_t* a;
    const int8_t* b;
    void* c;
    const float* bias;
    const void* addend;
    int add_f32;                    // addend is float32 (else bf16)
    float* cw32;                    // optional second copy of the result in float32 (same ldc)
    const float *sa, *za, *ca;      // per output row t: scale, rounded zero-point, sum of the stored codes of the row
    const float *sw, *zw, *cw;      // per output column n (= weight row)
    int64_t M, N, K, lda, ldb, ldc;
    float off_a, off_w;             // stored code = grid code - off
    int tiles_n, tiles_m, strip;
};

constexpr int IBM = 256, IBN = 128, IBK = 128;     // IBK in bytes = codes
constexpr int IA_BYTES = IBM * IBK;                // 32 KiB
constexpr int IB_BYTES = IBN * IBK;                // 16 KiB
constexpr int ISTAGE = IA_BYTES + IB_BYTES;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16b(const int8_t* src, char* dst) {
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
}

// per-lane source pointer of LDS-DMA piece `piece` (8 rows x 128 B) at k0 = 0
__device__ __forceinline__ const int8_t* piece_src8(const int8_t* base, int64_t ld, int64_t i0, int64_t I, int piece, int lane) {
    const int row = piece * 8 + (lane >> 3);
    const int src_chunk = (lane & 7) ^ ((row >> 1) & 7);
    int64_t i = i0 + row;
    i = i < I ? i : I - 1;
    return base + i * ld + src_chunk * 16;
}

template <int OFF>
__device__ __forceinline__ i32x4 asm_ds_read_b128_i(uint32_t addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

template <typename TOUT>
__global__ void __launch_bounds__(512) gemm_i8_p3_kernel(GemmI8P p) {
    __shared__ __attribute__((aligned(16))) char smem[3 * ISTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    int64_t m0, n0;
    {
        // strip-ordered ids (see gemm_bf16_p3_kernel): strips of `W` n-tiles, m-major inside a strip
        const int W = p.strip;
        const int ssz = p.tiles_m * W, strip = tile / ssz, rem = tile - strip * ssz;
        const int w = p.tiles_n - strip * W < W ? p.tiles_n - strip * W : W;
        const int mi = rem / w;
        m0 = (int64_t)mi * IBM;
        n0 = (int64_t)(strip * W + (rem - mi * w)) * IBN;
    }
    const int8_t* ga[4];
    const int8_t* gb[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) ga[q] = piece_src8(p.a, p.lda, m0, p.M, wid + 8 * q, lane);
#pragma unroll
    for (int q = 0; q < 2; ++q) gb[q] = piece_src8(p.b, p.ldb, n0, p.N, wid + 8 * q, lane);

    i32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};

    uint32_t ba[2], bb[2];
    {
        const int r = lane & 15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ba[ks] = (uint32_t)((wm * 64 + r) * 128 + ((((ks * 4) + (lane >> 4)) ^ (r >> 1)) << 4));
            bb[ks] = (uint32_t)((wn * 64 + r) * 128 + ((((ks * 4) + (lane >> 4)) ^ (r >> 1)) << 4));
        }
    }
    const uint32_t lds_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int nt = (int)(p.K / IBK);

    auto issue = [&](int stage) {
        char* sa = smem + stage * ISTAGE + wid * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) { glds16b(ga[q], sa + q * 8192); ga[q] += IBK; }
        char* sb = sa + IA_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) { glds16b(gb[q], sb + q * 8192); gb[q] += IBK; }
    };
    auto issue_h1 = [&](int stage) {
        char* sa = smem + stage * ISTAGE + wid * 1024;
#pragma unroll
        for (int q = 0; q < 3; ++q) { glds16b(ga[q], sa + q * 8192); ga[q] += IBK; }
    };
    auto issue_h2 = [&](int stage, int which) {
        char* sa = smem + stage * ISTAGE + wid * 1024;
        if (which == 0) { glds16b(ga[3], sa + 3 * 8192); ga[3] += IBK; }
        else { glds16b(gb[which - 1], sa + IA_BYTES + (which - 1) * 8192); gb[which - 1] += IBK; }
    };
    // same staggered two-group schedule and hazard reasoning as gemm_bf16_p3_kernel<.., STAGGER = true, SPLIT = true>
    const bool grp_b = wid >= 4;
    issue(0);
    if (nt > 1) {
        issue(1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp_b) __builtin_amdgcn_s_barrier();
    int s_cur = 0, s_pre = 2;
    for (int t = 0; t < nt; ++t) {
        if (t + 2 < nt) issue_h1(s_pre);
        const uint32_t sao = lds_base + (uint32_t)(s_cur * ISTAGE);
        const uint32_t sbo = sao + IA_BYTES;
        i32x4 fa[2][4], fb[2][4];
        fa[0][0] = asm_ds_read_b128_i<0 * 2048>(ba[0] + sao);
        fa[0][1] = asm_ds_read_b128_i<1 * 2048>(ba[0] + sao);
        fa[0][2] = asm_ds_read_b128_i<2 * 2048>(ba[0] + sao);
        fa[0][3] = asm_ds_read_b128_i<3 * 2048>(ba[0] + sao);
        fb[0][0] = asm_ds_read_b128_i<0 * 2048>(bb[0] + sbo);
        fb[0][1] = asm_ds_read_b128_i<1 * 2048>(bb[0] + sbo);
        fb[0][2] = asm_ds_read_b128_i<2 * 2048>(bb[0] + sbo);
        fb[0][3] = asm_ds_read_b128_i<3 * 2048>(bb[0] + sbo);
        fa[1][0] = asm_ds_read_b128_i<0 * 2048>(ba[1] + sao);
        fa[1][1] = asm_ds_read_b128_i<1 * 2048>(ba[1] + sao);
        fa[1][2] = asm_ds_read_b128_i<2 * 2048>(ba[1] + sao);
        fa[1][3] = asm_ds_read_b128_i<3 * 2048>(ba[1] + sao);
        fb[1][0] = asm_ds_read_b128_i<0 * 2048>(bb[1] + sbo);
        fb[1][1] = asm_ds_read_b128_i<1 * 2048>(bb[1] + sbo);
        fb[1][2] = asm_ds_read_b128_i<2 * 2048>(bb[1] + sbo);
        fb[1][3] = asm_ds_read_b128_i<3 * 2048>(bb[1] + sbo);
        if (t + 1 < nt) {
            if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
        {
            const bool more = t + 2 < nt;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        // swapped operands (B as the A-operand): lane holds 4 consecutive n of one m, as in the bf16 kernel
                        acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
                    if (more && ks * 4 + i == 1) issue_h2(s_pre, 0);
                    if (more && ks * 4 + i == 3) issue_h2(s_pre, 1);
                    if (more && ks * 4 + i == 5) issue_h2(s_pre, 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        s_cur = s_cur == 2 ? 0 : s_cur + 1;
        s_pre = s_pre == 2 ? 0 : s_pre + 1;
    }
    if (!grp_b) __builtin_amdgcn_s_barrier();
    // ---- epilogue through LDS: park the wave's 64 x 64 int32 block, read it back row-wise, finish in fp32, store whole lines ----
    constexpr int RS = 272;
    char* epi = smem + wid * (64 * RS);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<i32x4*>(epi + (i * 16 + (lane & 15)) * RS + (j * 16 + (lane >> 4) * 4) * 4) = acc[i][j];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int64_t n = n0 + wn * 64 + (lane & 7) * 8;
    const bool ncol = n < p.N;                       // N % 8 == 0 (host): a lane's 8 columns are all inside or all outside
    float sw[8], zw[8], cw[8], bv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { sw[r] = 0.f; zw[r] = 0.f; cw[r] = 0.f; bv[r] = 0.f; }
    if (ncol) {
        Vec8<float>::load(p.sw + n, sw);
        Vec8<float>::load(p.zw + n, zw);
        Vec8<float>::load(p.cw + n, cw);
        if (p.bias) Vec8<float>::load(p.bias + n, bv);
#pragma unroll
        for (int r = 0; r < 8; ++r) zw[r] -= p.off_w;
    }
    const float Kf = (float)p.K;
    TOUT* C = reinterpret_cast<TOUT*>(p.c);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = q * 8 + (lane >> 3);
        const int64_t m = m0 + wm * 64 + row;
        const i32x4 lo = *reinterpret_cast<const i32x4*>(epi + row * RS + (lane & 7) * 32);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(epi + row * RS + (lane & 7) * 32 + 16);
        if (m >= p.M || !ncol) continue;
        const float s_a = p.sa[m];
        const float z_a = p.za[m] - p.off_a;
        const float a2 = p.ca[m] - Kf * z_a;
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float ai = (float)(r < 4 ? lo[r] : hi[r - 4]);
            float inner = fmaf(-z_a, cw[r], ai);
            inner = fmaf(-zw[r], a2, inner);
            v[r] = fmaf(s_a * sw[r], inner, bv[r]);
        }
        TOUT* dst = C + m * p.ldc + n;
        if (p.addend) {
            float a8[8];
            if (p.add_f32) Vec8<float>::load(reinterpret_cast<const float*>(p.addend) + m * p.ldc + n, a8);
            else Vec8<bf16_t>::load(reinterpret_cast<const bf16_t*>(p.addend) + m * p.ldc + n, a8);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += a8[r];
        }
        Vec8<TOUT>::store(dst, v);
        if (p.cw32) Vec8<float>::store(p.cw32 + m * p.ldc + n, v);
    }
}

// small / ragged problems (tests, tiny models): one thread per output element, same arithmetic
template <typename TOUT>
__global__ void __launch_bounds__(256) gemm_i8_ref_kernel(GemmI8P p) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= p.M * p.N) return;
    const int64_t m = idx / p.N, n = idx - m * p.N;
    const int8_t* ar = p.a + m * p.lda;
    const int8_t* br = p.b + n * p.ldb;
    int acc = 0;
    for (int64_t k = 0; k < p.K; ++k) acc += (int)ar[k] * (int)br[k];
    const float z_a = p.za[m] - p.off_a, z_w = p.zw[n] - p.off_w;
    const float a2 = p.ca[m] - (float)p.K * z_a;
    float inner = fmaf(-z_a, p.cw[n], (float)acc);
    inner = fmaf(-z_w, a2, inner);
    float v = fmaf(p.sa[m] * p.sw[n], inner, p.bias ? p.bias[n] : 0.f);
    TOUT* C = reinterpret_cast<TOUT*>(p.c);
    if (p.addend)
        v += p.add_f32 ? reinterpret_cast<const float*>(p.addend)[m * p.ldc + n]
                       : (float)reinterpret_cast<const bf16_t*>(p.addend)[m * p.ldc + n];
    C[m * p.ldc + n] = (TOUT)v;
    if (p.cw32) p.cw32[m * p.ldc + n] = v;
}

int dbg_env_i8(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace

extern "C" int oq_gemm_i8(const void* a_codes, const void* b_codes, void* c, float* c_wide, const float* bias,
                          const void* addend, int addend_dtype, const float* a_scale, const float* a_zp, const float* a_csum, const float* b_scale, const float* b_zp,
                          const float* b_csum, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                          int a_bits, int b_bits, int out_dtype, void* stream) {
    OQ_CHECK_ARG(a_codes && b_codes && c && a_scale && a_zp && a_csum && b_scale && b_zp && b_csum, "oq_gemm_i8: null operand");
    OQ_CHECK_ARG(M > 0 && N > 0 && K > 0, "oq_gemm_i8: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
    OQ_CHECK_ARG(out_dtype == OQ_F32 || out_dtype == OQ_BF16, "oq_gemm_i8: out dtype %d", out_dtype);
    OQ_CHECK_ARG(!addend || addend_dtype == OQ_F32 || addend_dtype == OQ_BF16, "oq_gemm_i8: addend dtype %d", addend_dtype);
    OQ_CHECK_ARG(a_bits >= 2 && a_bits <= 8 && b_bits >= 2 && b_bits <= 8, "oq_gemm_i8: code widths %d / %d (2..8 bit grids)", a_bits, b_bits);
    OQ_CHECK_ARG(lda >= K && ldb >= K && ldc >= N, "oq_gemm_i8: leading dimensions");
    // |sum_k a*b| <= K * 128 * 128 must stay inside int32
    OQ_CHECK_ARG(K <= (1ll << 16), "oq_gemm_i8: K = %lld too long for int32 accumulation", (long long)K);
    GemmI8P p{};
    p.a = (const int8_t*)a_codes; p.b = (const int8_t*)b_codes; p.c = c; p.bias = bias; p.addend = addend;
    p.add_f32 = addend_dtype == OQ_F32; p.cw32 = c_wide;
    p.sa = a_scale; p.za = a_zp; p.ca = a_csum; p.sw = b_scale; p.zw = b_zp; p.cw = b_csum;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.off_a = a_bits == 8 ? 128.f : 0.f;       // the quantisers store plain grid codes, 8-bit grids as code - 128
    p.off_w = b_bits == 8 ? 128.f : 0.f;
    hipStream_t st = (hipStream_t)stream;
    const bool fast = K % IBK == 0 && lda % 16 == 0 && ldb % 16 == 0 && oq_aligned16(a_codes) && oq_aligned16(b_codes) &&
                      M >= 8 && N >= 128 && N % 8 == 0 && ldc % 8 == 0 && oq_aligned16(c) && (!addend || oq_aligned16(addend)) && oq_aligned16(c_wide) &&
                      (!bias || oq_aligned16(bias)) && oq_aligned16(b_scale) && oq_aligned16(b_zp) && oq_aligned16(b_csum) &&
                      dbg_env_i8("OQ_GEMM_I8_REF", 0) == 0;
    if (fast) {
        const int64_t tm = (M + IBM - 1) / IBM, tn = (N + IBN - 1) / IBN;
        OQ_CHECK_ARG(tm * tn < (1ll << 30), "oq_gemm_i8: too many tiles");
        p.tiles_m = (int)tm; p.tiles_n = (int)tn;
        p.strip = 8;
        dim3 grid((unsigned)(tm * tn));
        if (out_dtype == OQ_F32) hipLaunchKernelGGL((gemm_i8_p3_kernel<float>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((gemm_i8_p3_kernel<bf16_t>), grid, dim3(512), 0, st, p);
    } else {
        const int64_t total = M * N;
        dim3 grid((unsigned)((total + 255) / 256));
        if (out_dtype == OQ_F32) hipLaunchKernelGGL((gemm_i8_ref_kernel<float>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_i8_ref_kernel<bf16_t>), grid, dim3(256), 0, st, p);
    }
    OQ_CHECK_LAUNCH("oq_gemm_i8");
    return OQ_OK;
}
