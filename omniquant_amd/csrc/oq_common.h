// Shared device/host helpers for the gfx950 kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/oq_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

void oq_set_error(const char* fmt, ...);

#define OQ_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            oq_set_error(__VA_ARGS__);     \
            return OQ_E_ARG;               \
        }                                  \
    } while (0)

#define OQ_CHECK_LAUNCH(name)                                                      \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            oq_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return OQ_E_LAUNCH;                                                    \
        }                                                                          \
    } while (0)

// ---- element load/store: 8 consecutive elements per lane as f32 ---------------------------------
template <typename T>
struct Vec8;

template <>
struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        f32x4 a = *reinterpret_cast<const f32x4*>(p);
        f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        f32x4 a, b;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
        *reinterpret_cast<f32x4*>(p) = a;
        *reinterpret_cast<f32x4*>(p + 4) = b;
    }
};

template <>
struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __builtin_bit_cast(float, r[i] << 16);
            v[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        *reinterpret_cast<bf16x8*>(p) = o;
    }
};

template <>
struct Vec8<f16_t> {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    static __device__ __forceinline__ void load(const f16_t* p, float (&v)[8]) {
        h8 r = *reinterpret_cast<const h8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
    }
    static __device__ __forceinline__ void store(f16_t* p, const float (&v)[8]) {
        h8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (f16_t)v[i];
        *reinterpret_cast<h8*>(p) = o;
    }
};

// ---- raw (not yet converted) 8-element vectors: lets a kernel keep the NEXT row's loads in flight in few VGPRs --
template <typename T>
struct Raw8;
template <>
struct Raw8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const f32x4*>(p);
        b = *reinterpret_cast<const f32x4*>(p + 4);
    }
    __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
};
template <>
struct Raw8<bf16_t> {
    u32x4 a;
    __device__ __forceinline__ void load(const bf16_t* p) { a = *reinterpret_cast<const u32x4*>(p); }
    __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __builtin_bit_cast(float, a[i] << 16);
            v[2 * i + 1] = __builtin_bit_cast(float, a[i] & 0xffff0000u);
        }
    }
};
template <>
struct Raw8<f16_t> {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    h8 a;
    __device__ __forceinline__ void load(const f16_t* p) { a = *reinterpret_cast<const h8*>(p); }
    __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
};

template <typename T>
__device__ __forceinline__ float ld1(const T* p) { return (float)*p; }
template <typename T>
__device__ __forceinline__ void st1(T* p, float v) { *p = (T)v; }

__device__ __forceinline__ float vmax(float a, float b);
__device__ __forceinline__ float vmin(float a, float b);
// ---- wave-level reductions (width = power of two <= 64, lanes grouped contiguously) -------------
// Butterfly on DPP (quad_perm / row_half_mirror / row_mirror: 1 VALU op each, no LDS crossbar), ds_swizzle for the
// 16<->16 step and two v_readlane for the 32<->32 step.  Every lane of a group ends with the group's result.
// (__shfl_xor lowers to ds_bpermute_b32 inside a runtime loop: ~100+ cycles per step, it dominated these kernels.)
__device__ __forceinline__ float dpp_f(float v, const int ctrl_sel) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (ctrl_sel) {
        case 0: r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, true); break;    // quad_perm [1,0,3,2]
        case 1: r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, true); break;    // quad_perm [2,3,0,1]
        case 2: r = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, true); break;   // row_half_mirror
        case 3: r = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, true); break;   // row_mirror
        default: r = __builtin_amdgcn_ds_swizzle(x, 0x401F); break;                    // lane ^ 16
    }
    return __builtin_bit_cast(float, r);
}
#define OQ_WAVE_REDUCE(NAME, OP)                                                              \
    __device__ __forceinline__ float NAME(float v, int width = 64) {                          \
        if (width >= 2) { const float o = dpp_f(v, 0); v = OP(v, o); }                         \
        if (width >= 4) { const float o = dpp_f(v, 1); v = OP(v, o); }                         \
        if (width >= 8) { const float o = dpp_f(v, 2); v = OP(v, o); }                         \
        if (width >= 16) { const float o = dpp_f(v, 3); v = OP(v, o); }                        \
        if (width >= 32) { const float o = dpp_f(v, 4); v = OP(v, o); }                        \
        if (width >= 64) {                                                                     \
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));   \
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));  \
            v = OP(a, b);                                                                      \
        }                                                                                      \
        return v;                                                                              \
    }
__device__ __forceinline__ float oq_addf(float a, float b) { return a + b; }
OQ_WAVE_REDUCE(wave_sum, oq_addf)
OQ_WAVE_REDUCE(wave_max, vmax)
OQ_WAVE_REDUCE(wave_min, vmin)
// NaN-propagating variants: torch.amax/amin propagate NaN, fmaxf does not.
__device__ __forceinline__ float nmax(float a, float b) { return (a != a) ? a : ((b != b) ? b : fmaxf(a, b)); }
__device__ __forceinline__ float nmin(float a, float b) { return (a != a) ? a : ((b != b) ? b : fminf(a, b)); }

// 1 / (1 + e^-x) on the hardware transcendental units (v_exp_f32 = 2^x, v_rcp_f32; ~2 ulp): two of these run per row
// and per lane of the quantiser kernels, the OCML expf + IEEE divide expansion was ~35 instructions each.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// Raw v_max_f32 / v_min_f32: fmaxf/fminf compile to canonicalise + max (3 VALU ops); the single instruction returns
// the non-NaN operand exactly like fmaxf (IEEE maxNum) and is all the hot loops need.
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

static inline int oq_dtype_size(int dt) { return dt == OQ_F32 ? 4 : 2; }
static inline bool oq_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
