// Device helpers shared by the quantiser kernels (oq_quant.hip: segment kernels; oq_rowq.hip: wave-per-row kernels).
// Everything here restates quantize/quantizer.py:84-147 arithmetic; both kernel families MUST use these functions so that
// forward, backward and the two families agree bit for bit on scale / zero-point / rounding decisions.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "oq_common.h"

// One description of a fake-quant problem (forward and backward share it).  Plain struct with external linkage: it is
// passed between oq_quant.hip and oq_rowq.hip.
struct FQ {
    const void* w;
    int64_t rows, cols, seg;
    int nbits, symmetric;
    float inv_q;          // 1 / (2^nbits - 1), correctly rounded on the host
    const float *col_mul, *row_div, *row_mul, *shift, *up, *low;
    // fwd
    void* y;
    float *scale, *zp, *xmin, *xmax, *wshift;      // bwd READS xmin / xmax (written by the forward)
    // bwd
    const void* g;
    const float* g_wshift;
    float *g_up, *g_low;
    void* gx;
    float *g_col_mul, *g_shift, *g_row_div, *g_row_mul;
    float* ws;   // bwd workspace: [2][gridDim.x][cols] per-workgroup column partials
    // fused producers (oq_rowq.hip): x = silu(w) * w2 feeds the quantiser directly; the backward writes d/dw to gx, d/dw2 to gx2
    const void* w2;
    void* gx2;
    int64_t ldw;          // row stride (elements) of w / w2 and gx / gx2 in the fused-producer kernels; 0 = cols
    // integer side channel of the forward (whole-row segments, nbits <= 8; optional): the grid codes as int8 -- plain codes
    // 0 .. 2^nbits-1, for 8-bit grids code - 128 -- and per row the sum of the STORED codes (NaN for a row whose scale is 0 / NaN,
    // so that the integer GEMM's output is NaN where the reference's is).  Consumed by oq_gemm_i8.
    int8_t* codes;
    float* csum;
};

// 8 grid codes (floats holding integers 0 .. 255) -> 8 bytes; v_cvt_pk_u8_f32 converts and places one byte per instruction
__device__ __forceinline__ void oq_store_codes8(int8_t* dst, const float (&c)[8], bool eight_bit) {
    uint32_t lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_u8_f32(c[0], 0, lo);
    lo = __builtin_amdgcn_cvt_pk_u8_f32(c[1], 1, lo);
    lo = __builtin_amdgcn_cvt_pk_u8_f32(c[2], 2, lo);
    lo = __builtin_amdgcn_cvt_pk_u8_f32(c[3], 3, lo);
    hi = __builtin_amdgcn_cvt_pk_u8_f32(c[4], 0, hi);
    hi = __builtin_amdgcn_cvt_pk_u8_f32(c[5], 1, hi);
    hi = __builtin_amdgcn_cvt_pk_u8_f32(c[6], 2, hi);
    hi = __builtin_amdgcn_cvt_pk_u8_f32(c[7], 3, hi);
    if (eight_bit) { lo ^= 0x80808080u; hi ^= 0x80808080u; }     // code - 128 as a signed byte
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    *reinterpret_cast<u32x2*>(dst) = u32x2{lo, hi};
}

namespace {


// block-wide reduction of up to 3 values; op: 0 sum, 1 max, 2 min.  All threads must call.
// block-wide reduction of up to 4 values; op: 0 sum, 1 max, 2 min.  All threads must call.  The cross-wave step reads
// the (at most 8) per-wave partials of a value with one or two ds_read_b128 and combines them unconditionally: slots of
// waves that do not exist hold the operation's identity (written once by red_init), so there is no runtime loop.
// Every call site owns its `red` region (the identities are per operation).
__device__ __forceinline__ float red_identity(int op) { return op == 0 ? 0.f : (op == 1 ? -INFINITY : INFINITY); }

template <int NV>
__device__ __forceinline__ void red_init(const int (&op)[NV], float* red /*[NV*8]*/) {
    if (threadIdx.x < NV * 8) red[threadIdx.x] = red_identity(op[threadIdx.x >> 3]);
}

template <int NV>
__device__ __forceinline__ void block_reduce(float (&v)[NV], const int (&op)[NV], float* red /*[NV*8], red_init'ed*/,
                                             unsigned wave_uniform = 0 /* bit i: v[i] is already a per-wave value */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((wave_uniform >> i) & 1) continue;
        v[i] = op[i] == 0 ? wave_sum(v[i]) : (op[i] == 1 ? wave_max(v[i]) : wave_min(v[i]));
    }
    if (nw == 1) return;
    __syncthreads();   // protect `red` from the previous use
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[i * 8 + wid] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(red + i * 8);
        float r;
        if (op[i] == 0) r = (a[0] + a[1]) + (a[2] + a[3]);
        else if (op[i] == 1) r = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
        else r = fminf(fminf(a[0], a[1]), fminf(a[2], a[3]));
        if (nw > 4) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(red + i * 8 + 4);
            if (op[i] == 0) r += (b[0] + b[1]) + (b[2] + b[3]);
            else if (op[i] == 1) r = fmaxf(r, fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
            else r = fminf(r, fminf(fminf(b[0], b[1]), fminf(b[2], b[3])));
        }
        v[i] = r;
    }
}

struct QP {
    float s, z, su, sl, hi, lo;
};

// x / d for the LET row factor: reciprocal + one fma-residual correction (3 VALU ops instead of the ~10-op IEEE
// expansion whose v_div_* / v_rcp ops made these kernels VALU-bound).  Forward and backward use the SAME function,
// so x, its min/max, ties and the clip mask are self-consistent; vs. an IEEE divide the result differs by <= 1 ulp
// in rare cases, which is below the ulp-level differences sigmoid/exp already introduce.
__device__ __forceinline__ float div_nr(float a, float d, float inv_d) {
    const float q = a * inv_d;
    const float r = fmaf(-q, d, a);
    return fmaf(r, inv_d, q);
}

// rne(x / s) evaluated as rne(x * (1/s)) -- bit-identical to the IEEE quotient's rounding except when x/s lies within
// ~2 ulp of a half-integer; those (rare) lanes redo the exact division, so the result equals rintf(x / s) always.
__device__ __forceinline__ float rne_div(float x, float s, float inv_s, float* tq) {
    float t = x * inv_s;
    float r = rintf(t);
    if (fabsf(t - r) > fmaf(-4e-7f, fabsf(t), 0.5f)) {
        t = x / s;
        r = rintf(t);
    }
    *tq = t;
    return r;
}

// round_ste forward exactly as the reference composes it, (round(t) - t) + t: equals rintf(t) for every finite t
// and turns +-inf (scale == 0, quirk Q1) into NaN like the reference does.
__device__ __forceinline__ float rne_ste(float t) {
    const float r = rintf(t);
    return (r - t) + t;
}

// Scale / zero-point of one segment.  The asymmetric branch avoids the ~12-instruction IEEE division expansion (every
// thread of the row runs this once per row): (hs-ls)/Q is formed by the Markstein sequence q0 = a*(1/Q),
// q1 = fma(fma(-q0, Q, a), 1/Q, q0), which is the correctly rounded quotient for the integer divisors 2^n-1 used here
// (1/Q correctly rounded, exact fma residual); the zero-point uses the reciprocal-multiply + exact fallback of rne_div.
// `inv_s` is only ever used inside rne_div-style verified rounding, so v_rcp_f32 (1 ulp) is accurate enough.
__device__ __forceinline__ QP make_qp(float hi, float lo, bool lwc, float up_logit, float low_logit, int nbits,
                                      int symmetric, float invQ, float* inv_s_out) {
    QP q;
    q.hi = hi;
    q.lo = lo;
    q.su = lwc ? sigmoidf_(up_logit) : 1.0f;
    q.sl = lwc ? sigmoidf_(low_logit) : 1.0f;
    const float hs = lwc ? q.su * hi : hi;
    const float ls = lwc ? q.sl * lo : lo;
    if (symmetric) {
        const float lv = (float)((1 << (nbits - 1)) - 1);
        float s = fmaxf(fabsf(hs), fabsf(ls)) / lv;
        if (hs != hs || ls != ls) s = hs + ls;               // keep NaN
        q.s = (s != s) ? s : fminf(fmaxf(s, 1e-5f), 1e4f);
        q.z = lv;
        *inv_s_out = __builtin_amdgcn_rcpf(q.s);
    } else {
        const float Q = (float)((1 << nbits) - 1);
        const float a = hs - ls;
        const float q0 = a * invQ;
        float sc = fmaf(fmaf(-q0, Q, a), invQ, q0);          // == a / Q (not clamped: reference quirk Q1)
        if (!(fabsf(a) >= 1e-30f && fabsf(a) <= 1e30f)) sc = a / Q;   // zero / tiny / huge / NaN: plain division
        q.s = sc;
        const float inv_s = __builtin_amdgcn_rcpf(sc);
        *inv_s_out = inv_s;
        float zp = -ls * inv_s;
        const float rz = rintf(zp);
        if (!(fabsf(zp - rz) <= fmaf(-4e-7f, fabsf(zp), 0.5f)) || !(fabsf(zp) < 9.9e3f)) {
            zp = -ls / sc;                                   // near a rounding boundary, at the clamp, inf or NaN
            zp = (zp != zp) ? zp : fminf(fmaxf(zp, -1e4f), 1e4f);
            q.z = rintf(zp);
        } else {
            q.z = rz;
        }
    }
    return q;
}


// ---- wave-per-row geometry shared by oq_rowq.hip and the fused norm + quant kernels of oq_norm.hip ------------------
int64_t env_i(const char* n, int64_t d) {
    const char* v = getenv(n);
    return v ? atoll(v) : d;
}

struct RowGeo {
    int nw;     // waves per row (1, 2, 4, 8)
    int chn;    // 16-byte chunks per lane actually used (<= 8)
    int wpb;    // waves per workgroup
};

// rows of 512 .. 32768 elements, multiple of 8
bool row_geo(int64_t cols, int force_nw, RowGeo* g) {
    if (cols % 8 != 0 || cols < 512 || cols > 32768) return false;
    const int64_t chunks = cols / 8;
    // two waves per row by default (4 chunks per lane, ~110 VGPRs, 4 waves per SIMD): measured better than one wave with 8
    // chunks (175 VGPRs) on every 4096-wide kernel of the step, in-step 254.8 -> 259 sample-steps/s (tools/sweep_rowq.sh)
    int nw = force_nw > 0 ? force_nw : (chunks >= 128 ? 2 : 1);
    while (nw <= 8 && (chunks + 64 * nw - 1) / (64 * nw) > 8) nw <<= 1;
    if (nw > 8 || (nw & (nw - 1))) return false;
    g->nw = nw;
    g->chn = (int)((chunks + 64 * nw - 1) / (64 * nw));
    g->wpb = nw > 4 ? 8 : 4;
    return true;
}

// cross-wave exchange of 4 per-row values among the nw waves of a row (double-buffered: one barrier per row)
__device__ __forceinline__ void row_exchange(float* red, int& par, int wid, int rslot, int nw, int lane, float (&v)[4],
                                             const int (&op)[4]) {
    float* rr = red + par * 32;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rr[wid * 4 + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = op[k] == 0 ? 0.f : (op[k] == 1 ? -INFINITY : INFINITY);
    for (int i = 0; i < nw; ++i) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(rr + (rslot * nw + i) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = op[k] == 0 ? v[k] + q[k] : (op[k] == 1 ? fmaxf(v[k], q[k]) : fminf(v[k], q[k]));
    }
    par ^= 1;
}

int n_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace

// wave-per-row kernels (oq_rowq.hip).  Return OQ_OK when they took the problem, 1 when the shape is not theirs (the
// caller then runs the segment kernels), or a negative OQ_E_* code.
int oq_rowq_fwd(const FQ& p, int w_dtype, int y_dtype, void* stream);
int oq_rowq_bwd(const FQ& p, int w_dtype, int g_dtype, float* workspace, int64_t workspace_floats, int64_t* partial_rows,
                void* stream);
int64_t oq_letq_bwd_blocks(int64_t rows);
int oq_letq_fwd_multi(const FQ* ps, int n, int w_dtype, int y_dtype, void* stream);
int oq_letq_bwd_multi(FQ* ps, int n, int w_dtype, int g_dtype, const int64_t* workspace_floats, int64_t* parts, void* stream);
int64_t oq_rowq_bwd_blocks(int64_t rows, int64_t cols);     // workgroups that write column partials (workspace rows)

// lanes-per-group kernels for grouped weights without LET (oq_groupq.hip).  oq_groupq_fwd / _bwd return OQ_OK when launched,
// 1 when the dtype pair is not theirs, negative on error.
bool oq_groupq_eligible(const FQ& p, bool backward);
int oq_groupq_fwd(const FQ& p, int w_dtype, int y_dtype, void* stream);
int oq_groupq_bwd(const FQ& p, int w_dtype, int g_dtype, void* stream);
