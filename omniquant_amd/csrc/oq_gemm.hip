// Fake-quant Linear / QuantMatMul GEMM for gfx950.
//
// Replaces F.linear at quantize/int_linear.py:62, torch.matmul / torch.bmm at quantize/int_matmul.py:41-43
// and the dgrad / wgrad GEMMs their autograd produces.  One strided-batched kernel covers every operand
// layout of forward and backward WITHOUT transposed copies:
//     C[m][n] = alpha * sum_k A(m,k) * B(n,k) + bias[n]
// where each operand is either k-contiguous (row-major [rows][k]) or k-strided ([k][rows]).
//
// bf16 path (production): 128x128x64 block tile, 4 waves (2x2), 64x64 per wave as 4x4
// v_mfma_f32_16x16x32_bf16 tiles, f32 accumulate.  Global -> registers -> LDS staging, double-buffered,
// one barrier per K-step, next tile's global loads in flight under the MFMAs.
//   * k-contiguous tiles sit in LDS as [128 rows][64 k] (128-B rows), 16-B chunk index XOR-swizzled with
//     (row>>1)&7 so every ds_read_b128 lane group covers 16 distinct slots (conflict-free).
//   * k-strided tiles sit in LDS as stored ([64 k][128 rows], 256-B rows) and are read with
//     ds_read_b64_tr_b16 (hardware transpose); 32-B chunk index XOR-swizzled with (k&3)|((k>>3)&1)<<2 so the 8
//     k-rows a 32-lane half touches spread over all 64 banks.
//   * operands are fed to the MFMA swapped (B as the A-operand) so each lane ends up holding 4 CONSECUTIVE
//     output columns: the epilogue stores 8/16 B per lane with no shuffles.
// f32 path (parity mode): exact f32 on v_mfma_f32_16x16x4_f32, 64x64x16 tile, scalar predicated loads: any
// shape, any alignment; numerics = k-ordered fmaf chain.
#include <stdlib.h>
#include "oq_common.h"

namespace {

int dbg_env_i(const char* name, int dflt) {     // tuning / A-B switches (read per call so one process can compare)
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Diagnostic build only (tools/gemm_stamps.hip compiles this file with -DOQ_GEMM_STAMPS): wave 0 of every workgroup writes
// s_memrealtime (100 MHz) at a few points into a buffer of its own.  In the product library no stamp executes.
#ifdef OQ_GEMM_STAMPS
__device__ unsigned long long* g_gemm_stamps = nullptr;      // [workgroup][8]
#define OQ_STAMP(slot)                                                                                      \
    do {                                                                                                    \
        if (g_gemm_stamps && threadIdx.x == 0)                                                              \
            g_gemm_stamps[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();              \
    } while (0)
#else
#define OQ_STAMP(slot) do { } while (0)
#endif

struct GemmP {
    const void* a;
    const void* b;
    void* c;
    const float* bias;
    const void* addend;      // optional: same dtype / leading dim / batch strides as C; C = alpha*A.B + bias + addend
    int64_t M, N, K, lda, ldb, ldc;
    int a_kc, b_kc;
    float alpha;
    int64_t batch_i, sa_o, sa_i, sb_o, sb_i, sc_o, sc_i;
    int tiles_n, tiles_m, nmajor, tri;
    int splitk, tiles_real;  // gemm_bf16_p3_kernel: contraction split into `splitk` parts (ids = part * tiles_real + tile), fp32
                             // partial tiles go to c + part * M * N (c = the workspace); a reduce launch finishes the output
    int blk48;               // gemm_bf16_p3_kernel: strip-ordered tile ids, ~4 x 8-tile blocks per XCD round (see the kernel)
    int epi_lds;             // gemm_bf16_p3_kernel: store the output tile through LDS in whole 128-byte lines
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand per stage

__device__ __forceinline__ int kstr_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// ---- global -> registers ---------------------------------------------------------------------------
// KC tile: 128 rows x 64 k.  thread: chunk = tid&7 (16 B of k), rows tid>>3 + 32*p.
// KS tile: 64 k-rows x 128 rows. thread: c16 = tid&15, k-rows tid>>4 + 16*p.
template <bool KC>
__device__ __forceinline__ void load_tile(const bf16_t* base, int64_t ld, int64_t i0, int64_t k0, int64_t I, int64_t K,
                                          int tid, u32x4 (&r)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (KC) {
            const int chunk = tid & 7, row = (tid >> 3) + 32 * p;
            const int64_t i = i0 + row, k = k0 + chunk * 8;
            if (i < I && k < K) v = *reinterpret_cast<const u32x4*>(base + i * ld + k);
        } else {
            const int c16 = tid & 15, kr = (tid >> 4) + 16 * p;
            const int64_t k = k0 + kr, i = i0 + c16 * 8;
            if (k < K && i < I) v = *reinterpret_cast<const u32x4*>(base + k * ld + i);
        }
        r[p] = v;
    }
}

// FAST path: row/column clamped once outside the K loop (out-of-range rows re-read the last valid row; their
// products only reach output rows/cols that are never stored), K % 64 == 0 guaranteed by the host: no predication,
// no per-iteration address arithmetic beyond one add.
template <bool KC>
__device__ __forceinline__ void tile_offsets(int64_t ld, int64_t i0, int64_t I, int tid, int64_t (&off)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (KC) {
            int64_t i = i0 + (tid >> 3) + 32 * p;
            i = i < I ? i : I - 1;
            off[p] = i * ld + (tid & 7) * 8;
        } else {
            int64_t i = i0 + (tid & 15) * 8;
            i = i < I ? i : I - 8;
            off[p] = (int64_t)((tid >> 4) + 16 * p) * ld + i;
        }
    }
}

template <bool KC>
__device__ __forceinline__ void load_tile_fast(const bf16_t* base, int64_t ld, int64_t k0, const int64_t (&off)[4],
                                               u32x4 (&r)[4]) {
    const int64_t step = KC ? k0 : k0 * ld;
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = *reinterpret_cast<const u32x4*>(base + off[p] + step);
}

template <bool KC>
__device__ __forceinline__ void store_tile(char* lds, int tid, const u32x4 (&r)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int off;
        if (KC) {
            const int chunk = tid & 7, row = (tid >> 3) + 32 * p;
            off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
        } else {
            const int c16 = tid & 15, kr = (tid >> 4) + 16 * p;
            off = kr * 256 + ((((c16 >> 1) ^ kstr_f(kr))) << 5) + ((c16 & 1) << 4);
        }
        *reinterpret_cast<u32x4*>(lds + off) = r[p];
    }
}

// fragment for the 16 rows starting at `row0` (multiple of 16, within the 128-row tile) and k-substep ks.
// returns lane's 8 bf16: element j = X[row0 + (lane&15)][ks*32 + 8*(lane>>4) + j]
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag(const char* lds, int row0, int ks, int lane) {
    if (KC) {
        const int row = row0 + (lane & 15);
        const int chunk = ks * 4 + (lane >> 4);
        const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
        return *reinterpret_cast<const bf16x8*>(lds + off);
    } else {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int k = ks * 32 + 8 * g + q;
        const int c32 = row0 >> 4;
        const int off0 = k * 256 + ((c32 ^ kstr_f(k)) << 5) + pp * 8;
        const int off1 = (k + 4) * 256 + ((c32 ^ kstr_f(k + 4)) << 5) + pp * 8;
        typedef __attribute__((address_space(3))) s16x4 lds_s4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + off0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + off1));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <bool AKC, bool BKC, typename TOUT, bool FAST>
__global__ void __launch_bounds__(256, 2) gemm_bf16_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    // XCD-aware tile mapping: workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2), so
    // give every XCD a CONTIGUOUS range of tiles, ordered so that the range splits the larger operand (bijective
    // for any grid size).
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    int64_t m0, n0;
    if (p.nmajor) { n0 = (int64_t)(tile / p.tiles_m) * BN; m0 = (int64_t)(tile % p.tiles_m) * BM; }
    else { m0 = (int64_t)(tile / p.tiles_n) * BM; n0 = (int64_t)(tile % p.tiles_n) * BN; }
    const int64_t bo = blockIdx.z / p.batch_i, bi = blockIdx.z % p.batch_i;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.a) + bo * p.sa_o + bi * p.sa_i;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.b) + bo * p.sb_o + bi * p.sb_i;
    TOUT* C = reinterpret_cast<TOUT*>(p.c) + bo * p.sc_o + bi * p.sc_i;

    f32x4 acc[4][4];   // [mt][nt], lane holds C[m = mt*16 + (lane&15)][n = nt*16 + (lane>>4)*4 + r]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (p.tri == 1 && n0 >= m0 + BM) return;               // causal: tile entirely above the diagonal
    u32x4 ra[4], rb[4];
    // causal contraction ranges (tri 2: k < m0+BM, tri 3: k >= m0), in K-tiles of 64
    int64_t kend = p.K;
    if (p.tri == 2 && m0 + BM < kend) kend = m0 + BM;
    const int64_t t0 = p.tri == 3 ? m0 / BK : 0;
    const int64_t nt = (kend + BK - 1) / BK;
    int64_t offa[4], offb[4];
    if (FAST) {
        tile_offsets<AKC>(p.lda, m0, p.M, tid, offa);
        tile_offsets<BKC>(p.ldb, n0, p.N, tid, offb);
        load_tile_fast<AKC>(A, p.lda, t0 * BK, offa, ra);
        load_tile_fast<BKC>(B, p.ldb, t0 * BK, offb, rb);
    } else {
        load_tile<AKC>(A, p.lda, m0, t0 * BK, p.M, p.K, tid, ra);
        load_tile<BKC>(B, p.ldb, n0, t0 * BK, p.N, p.K, tid, rb);
    }
    store_tile<AKC>(smem, tid, ra);
    store_tile<BKC>(smem + TILE_BYTES, tid, rb);
    __syncthreads();
    int cur = 0;
    for (int64_t t = t0; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if (more) {
            if (FAST) {
                load_tile_fast<AKC>(A, p.lda, (t + 1) * BK, offa, ra);
                load_tile_fast<BKC>(B, p.ldb, (t + 1) * BK, offb, rb);
            } else {
                load_tile<AKC>(A, p.lda, m0, (t + 1) * BK, p.M, p.K, tid, ra);
                load_tile<BKC>(B, p.ldb, n0, (t + 1) * BK, p.N, p.K, tid, rb);
            }
        }
        const char* sa = smem + cur * 2 * TILE_BYTES;
        const char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag<AKC>(sa, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag<BKC>(sb, wn * 64 + j * 16, ks, lane);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // swapped operands: D[row = n-sub][col = m-sub]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if (more) {
            char* da = smem + (cur ^ 1) * 2 * TILE_BYTES;
            store_tile<AKC>(da, tid, ra);
            store_tile<BKC>(da + TILE_BYTES, tid, rb);
        }
        __syncthreads();
        cur ^= 1;
    }
    // ---- epilogue: lane holds 4 consecutive n for one m ------------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;   // N % 4 == 0 is enforced on the host
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r] * p.alpha;
                if (p.bias) v[r] += p.bias[n + r];
            }
            if (p.addend) {          // residual / gradient accumulation fused into the store (may alias C)
                const TOUT* ad = reinterpret_cast<const TOUT*>(p.addend) + (C - reinterpret_cast<TOUT*>(p.c)) + m * p.ldc + n;
                if constexpr (sizeof(TOUT) == 4) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(ad);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += av[r];
                } else {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf4a;
                    const bf4a av = *reinterpret_cast<const bf4a*>(ad);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)av[r];
                }
            }
            TOUT* dst = C + m * p.ldc + n;
            if constexpr (sizeof(TOUT) == 4) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
                *reinterpret_cast<bf4*>(dst) = bf4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 256x128x64 kernel: 8 waves (4x2, 64x64 per wave), LDS-DMA staging (global_load_lds, 16 B per lane), 3-stage LDS
// ring (3 x 48 KiB) with a COUNTED vmcnt so one K-tile stays in flight across the single barrier per K-step.
// No registers or ds_write are spent on staging; swizzles are applied on the per-lane SOURCE address because the
// LDS-DMA destination is lane-linear (wave-uniform base + lane*16).
// ---------------------------------------------------------------------------------------------------
constexpr int P3_BM = 256, P3_BN = 128;
constexpr int P3_A_BYTES = P3_BM * BK * 2;   // 32 KiB
constexpr int P3_B_BYTES = P3_BN * BK * 2;   // 16 KiB
constexpr int P3_STAGE = P3_A_BYTES + P3_B_BYTES;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const bf16_t* src, char* dst) {
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
}

// per-lane source pointer of LDS-DMA piece `piece` (1 KiB of the operand tile) at k0 = 0.
//   KC: piece = 8 rows x 128 B.   KS: piece = (1024 / ROWB) k-rows of ROWB bytes (ROWB = 2 * tile rows).
template <bool KC, int ROWB>
__device__ __forceinline__ const bf16_t* piece_src(const bf16_t* base, int64_t ld, int64_t i0, int64_t I, int piece,
                                                   int lane) {
    if (KC) {
        const int row = piece * 8 + (lane >> 3);
        const int src_chunk = (lane & 7) ^ ((row >> 1) & 7);
        int64_t i = i0 + row;
        i = i < I ? i : I - 1;
        return base + i * ld + src_chunk * 8;
    } else {
        constexpr int LPR = ROWB / 16;              // lanes (16-B chunks) per k-row
        const int krow = piece * (64 / LPR) + lane / LPR;
        const int pc16 = lane % LPR;
        const int c32 = (pc16 >> 1) ^ kstr_f(krow);
        int64_t i = i0 + (c32 * 2 + (pc16 & 1)) * 8;
        i = i < I ? i : I - 8;
        return base + (int64_t)krow * ld + i;
    }
}

template <bool KC, int ROWB>
__device__ __forceinline__ bf16x8 read_frag3(const char* lds, int row0, int ks, int lane) {
    if (KC) {
        const int row = row0 + (lane & 15);
        const int chunk = ks * 4 + (lane >> 4);
        const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
        return *reinterpret_cast<const bf16x8*>(lds + off);
    } else {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int k = ks * 32 + 8 * g + q;
        const int c32 = row0 >> 4;
        const int off0 = k * ROWB + ((c32 ^ kstr_f(k)) << 5) + pp * 8;
        const int off1 = (k + 4) * ROWB + ((c32 ^ kstr_f(k + 4)) << 5) + pp * 8;
        typedef __attribute__((address_space(3))) s16x4 lds_s4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + off0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(lds + off1));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

// ---- fragment reads as inline asm --------------------------------------------------------------------------
// hipcc puts an `s_waitcnt vmcnt(0)` in front of every compiler-visible ds_read_b64_tr_b16 while an LDS-DMA is in
// flight (it cannot prove the transposed read does not alias the DMA destination), which drains the prefetched
// K-tile each iteration (measured: -25 % on every layout with a k-strided operand).  Reads issued from inline asm
// are invisible to that pass; their completion is waited for explicitly (counted lgkmcnt) and a sched_barrier keeps
// the MFMAs below the wait (cdna guide 5.4 rule 18).
template <int OFF>
__device__ __forceinline__ bf16x8 asm_ds_read_b128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ s16x4 asm_ds_read_tr(uint32_t addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}

// Per-lane LDS byte offsets (relative to the stage's operand base) of the fragment reads of one wave.
//   KC: base[ks]  (+ i*2048 immediate per 16-row tile i)       -> base[0..1]
//   KS: base[i]   (+ ks*32*ROWB and +4*ROWB immediates)         -> base[0..3]
template <bool KC, int ROWB>
__device__ __forceinline__ void frag_bases(int wave_row0, int lane, uint32_t (&base)[4]) {
    if (KC) {
        const int r = lane & 15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            base[ks] = (uint32_t)((wave_row0 + r) * 128 + ((((ks * 4) + (lane >> 4)) ^ (r >> 1)) << 4));
        base[2] = base[3] = 0;
    } else {
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int k = 8 * g + q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c32 = (wave_row0 >> 4) + i;
            base[i] = (uint32_t)(k * ROWB + ((c32 ^ kstr_f(k)) << 5) + pp * 8);
        }
    }
}

template <bool KC, int ROWB, int KS_, int I_>
__device__ __forceinline__ bf16x8 frag_read(const uint32_t (&b)[4], uint32_t stage_off) {
    if constexpr (KC) {
        return asm_ds_read_b128<I_ * 2048>(b[KS_] + stage_off);
    } else {
        const s16x4 lo = asm_ds_read_tr<KS_ * 32 * ROWB>(b[I_] + stage_off);
        const s16x4 hi = asm_ds_read_tr<KS_ * 32 * ROWB + 4 * ROWB>(b[I_] + stage_off);
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <bool AKC, bool BKC, typename TOUT, bool STAGGER, bool SPLIT>
__global__ void __launch_bounds__(512) gemm_bf16_p3_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) char smem[3 * P3_STAGE];   // ONE array: see cdna guide (second-object trap)
    OQ_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    int part = 0;
    if (p.splitk > 1) { part = tile / p.tiles_real; tile -= part * p.tiles_real; }
    int64_t m0, n0;
    if (p.blk48) {
        // tiles are numbered strip by strip (a strip = 8 n-tiles wide, the last one narrower), m-major inside a strip: ANY 32
        // consecutive ids -- what an XCD runs at a time -- cover about 4 x 8 tiles (1024 x 1024 outputs): 4 A-tiles + 8 B-tiles
        // = 256 KB of operands per K-step through that XCD's L2 instead of 8 + 4 tiles = 320 KB with plain n-major order.
        // No alignment of the tile counts is needed.
        const int W = p.blk48;                          // strip width in n-tiles (8)
        const int ssz = p.tiles_m * W, strip = tile / ssz, rem = tile - strip * ssz;
        const int w = p.tiles_n - strip * W < W ? p.tiles_n - strip * W : W;
        const int mi = rem / w;
        m0 = (int64_t)mi * P3_BM;
        n0 = (int64_t)(strip * W + (rem - mi * w)) * P3_BN;
    } else if (p.nmajor) { n0 = (int64_t)(tile / p.tiles_m) * P3_BN; m0 = (int64_t)(tile % p.tiles_m) * P3_BM; }
    else { m0 = (int64_t)(tile / p.tiles_n) * P3_BM; n0 = (int64_t)(tile % p.tiles_n) * P3_BN; }
    const int64_t bo = blockIdx.z / p.batch_i, bi = blockIdx.z % p.batch_i;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.a) + bo * p.sa_o + bi * p.sa_i;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.b) + bo * p.sb_o + bi * p.sb_i;
    TOUT* C = reinterpret_cast<TOUT*>(p.c) + bo * p.sc_o + bi * p.sc_i + (int64_t)part * p.M * p.N;

    // LDS-DMA pieces of this wave: A pieces wid + 8q (q < 4), B pieces wid + 8q (q < 2)
    const bf16_t* ga[4];
    const bf16_t* gb[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) ga[q] = piece_src<AKC, 2 * P3_BM>(A, p.lda, m0, p.M, wid + 8 * q, lane);
#pragma unroll
    for (int q = 0; q < 2; ++q) gb[q] = piece_src<BKC, 2 * P3_BN>(B, p.ldb, n0, p.N, wid + 8 * q, lane);
    const int64_t astep = AKC ? BK : (int64_t)BK * p.lda;
    const int64_t bstep = BKC ? BK : (int64_t)BK * p.ldb;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint32_t ba[4], bb[4];
    frag_bases<AKC, 2 * P3_BM>(wm * 64, lane, ba);
    frag_bases<BKC, 2 * P3_BN>(wn * 64, lane, bb);
    const uint32_t lds_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    if (p.tri == 1 && n0 >= m0 + P3_BM) return;            // causal: tile entirely above the diagonal
    int64_t kend = p.K;
    if (p.tri == 2 && m0 + P3_BM < kend) kend = m0 + P3_BM;
    int t0 = p.tri == 3 ? (int)(m0 / BK) : 0;
    if (p.splitk > 1) {                                    // this part's slice of the contraction
        const int64_t kpart = p.K / p.splitk;
        t0 = (int)(part * kpart / BK);
        kend = (part + 1) * kpart;
    }
    const int nt = (int)(kend / BK) - t0;                  // K-tiles of this tile's contraction range
    if (t0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ga[q] += (int64_t)t0 * astep;
#pragma unroll
        for (int q = 0; q < 2; ++q) gb[q] += (int64_t)t0 * bstep;
    }
    auto issue = [&](int stage) {
        char* sa = smem + stage * P3_STAGE + wid * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) { glds16(ga[q], sa + q * 8192); ga[q] += astep; }
        char* sb = sa + P3_A_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) { glds16(gb[q], sb + q * 8192); gb[q] += bstep; }
    };
    // ---- staggered two-group schedule ----------------------------------------------------------------------
    // Waves w and w+4 share a SIMD.  Each K-tile is split into a load half H1 (issue the LDS-DMA of tile t+2, read
    // tile t's fragments into registers) and a pure-MFMA half H2, separated by barriers; group B (waves 4-7) runs
    // one barrier interval behind group A, so on every SIMD one wave issues MFMAs while its partner issues
    // LDS-DMA / ds_reads.  Hazards (cdna guide: "one barrier MORE when two wave groups run staggered"):
    //   RAW  tile t is read in H1(t) (interval 2t for A, 2t+1 for B); every wave waits `vmcnt` for its own pieces of
    //        tile t at the end of its H1(t-1) and then passes a barrier (<= interval 2t-1 | 2t).
    //   WAR  the DMA of tile t+2 overwrites tile t-1's stage; all fragment reads of tile t-1 are COMPLETE
    //        (lgkmcnt(0)) before the barrier that ends H1(t-1) of either group (<= interval 2t-1 | 2t).
    const bool grp_b = STAGGER && wid >= 4;
    // SPLIT: the 6 LDS-DMA pieces of a tile are issued half in H1 and half between the MFMAs of H2, which shortens the
    // load half (the longer of the two) of every interval
    auto issue_h1 = [&](int stage) {
        char* sa = smem + stage * P3_STAGE + wid * 1024;
#pragma unroll
        for (int q = 0; q < 3; ++q) { glds16(ga[q], sa + q * 8192); ga[q] += astep; }
    };
    auto issue_h2 = [&](int stage, int which) {
        char* sa = smem + stage * P3_STAGE + wid * 1024;
        if (which == 0) { glds16(ga[3], sa + 3 * 8192); ga[3] += astep; }
        else { glds16(gb[which - 1], sa + P3_A_BYTES + (which - 1) * 8192); gb[which - 1] += bstep; }
    };
    issue(0);
    if (nt > 1) {
        issue(1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    OQ_STAMP(1);
    if (grp_b) __builtin_amdgcn_s_barrier();      // group B idles through interval 0
    int s_cur = 0, s_pre = 2;
    for (int t = 0; t < nt; ++t) {
        // ---------------- H1(t) ----------------
        if (t + 2 < nt) {
            if (SPLIT) issue_h1(s_pre);
            else issue(s_pre);
        }
        const uint32_t sao = lds_base + (uint32_t)(s_cur * P3_STAGE);
        const uint32_t sbo = sao + P3_A_BYTES;
        bf16x8 fa[2][4], fb[2][4];
        fa[0][0] = frag_read<AKC, 2 * P3_BM, 0, 0>(ba, sao);
        fa[0][1] = frag_read<AKC, 2 * P3_BM, 0, 1>(ba, sao);
        fa[0][2] = frag_read<AKC, 2 * P3_BM, 0, 2>(ba, sao);
        fa[0][3] = frag_read<AKC, 2 * P3_BM, 0, 3>(ba, sao);
        fb[0][0] = frag_read<BKC, 2 * P3_BN, 0, 0>(bb, sbo);
        fb[0][1] = frag_read<BKC, 2 * P3_BN, 0, 1>(bb, sbo);
        fb[0][2] = frag_read<BKC, 2 * P3_BN, 0, 2>(bb, sbo);
        fb[0][3] = frag_read<BKC, 2 * P3_BN, 0, 3>(bb, sbo);
        fa[1][0] = frag_read<AKC, 2 * P3_BM, 1, 0>(ba, sao);
        fa[1][1] = frag_read<AKC, 2 * P3_BM, 1, 1>(ba, sao);
        fa[1][2] = frag_read<AKC, 2 * P3_BM, 1, 2>(ba, sao);
        fa[1][3] = frag_read<AKC, 2 * P3_BM, 1, 3>(ba, sao);
        fb[1][0] = frag_read<BKC, 2 * P3_BN, 1, 0>(bb, sbo);
        fb[1][1] = frag_read<BKC, 2 * P3_BN, 1, 1>(bb, sbo);
        fb[1][2] = frag_read<BKC, 2 * P3_BN, 1, 2>(bb, sbo);
        fb[1][3] = frag_read<BKC, 2 * P3_BN, 1, 3>(bb, sbo);
        if (t + 1 < nt) {
            // my pieces of tile t+1 (issued one H1 ago) have landed; tile t+2 (just issued) stays in flight
            if (t + 2 < nt) {
                if (SPLIT) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");      // fragments are in registers: the stage may be overwritten
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---------------- H2(t): 32 MFMAs on registers ----------------
        __builtin_amdgcn_s_setprio(1);
        if (SPLIT) {
            const bool more = t + 2 < nt;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
                    // one DMA piece after MFMA groups 2, 4 and 6 (of 8)
                    if (more && ks * 4 + i == 1) issue_h2(s_pre, 0);
                    if (more && ks * 4 + i == 3) issue_h2(s_pre, 1);
                    if (more && ks * 4 + i == 5) issue_h2(s_pre, 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        s_cur = s_cur == 2 ? 0 : s_cur + 1;
        s_pre = s_pre == 2 ? 0 : s_pre + 1;
    }
    if (STAGGER && !grp_b) __builtin_amdgcn_s_barrier();     // group A's matching extra barrier
    OQ_STAMP(2);
    if (p.epi_lds) {
        // ---- epilogue through LDS: every global store covers whole 128-byte lines ---------------------------------
        // The MFMA accumulators hold, per lane, 4 consecutive columns of ONE row: stored from there, a wave instruction
        // touches 16 rows x 32 bytes (16 partial lines; measured 3.5-4.8 us per tile, every CU bursting at once).  After the
        // final barrier nobody reads the operand ring any more, so each wave parks its 64 x 64 fp32 block in a 17 KiB region
        // of its own (row stride 272 B: conflict-free ds_write_b128), reads it back row-wise and stores 16 B (bf16: 8 lanes
        // per 128-byte row) or 2 x 16 B (f32) per lane.  Same arithmetic, same single rounding: bit-identical results.
        constexpr int RS = 272;
        char* epi = smem + wid * (64 * RS);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            const int64_t nb = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (p.bias && nb < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + nb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][j][r] * p.alpha;
                    if (p.bias) v[r] += bv[r];
                }
                *reinterpret_cast<f32x4*>(epi + (i * 16 + (lane & 15)) * RS + (j * 16 + (lane >> 4) * 4) * 4) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t n = n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = q * 8 + (lane >> 3);
            const int64_t m = m0 + wm * 64 + row;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(epi + row * RS + (lane & 7) * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(epi + row * RS + (lane & 7) * 32 + 16);
            if (m >= p.M || n >= p.N) continue;
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            TOUT* dst = C + m * p.ldc + n;
            if (p.addend) {
                const TOUT* ad = reinterpret_cast<const TOUT*>(p.addend) + (C - reinterpret_cast<TOUT*>(p.c)) + m * p.ldc + n;
                if constexpr (sizeof(TOUT) == 4) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(ad), a1 = *reinterpret_cast<const f32x4*>(ad + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] += a0[r]; v[4 + r] += a1[r]; }
                } else {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(ad);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)av[r];
                }
            }
            if constexpr (sizeof(TOUT) == 4) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                bf16x8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = (bf16_t)v[r];
                *reinterpret_cast<bf16x8*>(dst) = o;
            }
        }
        OQ_STAMP(3);
#ifdef OQ_GEMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OQ_STAMP(4);
#endif
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r] * p.alpha;
                if (p.bias) v[r] += p.bias[n + r];
            }
            if (p.addend) {          // residual / gradient accumulation fused into the store (may alias C)
                const TOUT* ad = reinterpret_cast<const TOUT*>(p.addend) + (C - reinterpret_cast<TOUT*>(p.c)) + m * p.ldc + n;
                if constexpr (sizeof(TOUT) == 4) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(ad);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += av[r];
                } else {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf4a;
                    const bf4a av = *reinterpret_cast<const bf4a*>(ad);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)av[r];
                }
            }
            TOUT* dst = C + m * p.ldc + n;
            if constexpr (sizeof(TOUT) == 4) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
                *reinterpret_cast<bf4*>(dst) = bf4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            }
        }
    }
    OQ_STAMP(3);
#ifdef OQ_GEMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OQ_STAMP(4);
#endif
}

// ---------------------------------------------------------------------------------------------------
// 256x256x32 kernel for the WEIGHT-GRADIENT products (dW = dY^T X: both operands k-strided, short contraction K = tokens,
// huge output): the 256x128x64 kernel pays 48 one-KiB LDS-DMA pieces and 16 fragment reads per wave for every 32 MFMAs, and at
// K = 2048 its per-tile fixed costs (first K-tile's latency, epilogue start, workgroup turnover: 6-8 us) are a quarter of a
// tile's life.  Here a workgroup owns 256x256 outputs and walks K in tiles of 32: per wave still 32 MFMAs per barrier pair
// (64x128 outputs as 4x8 tiles, ONE k-step), but 32 pieces (4 per wave) and 12 fragment reads -- a third less staging per flop --
// and half as many output tiles.  K-tiles of 32 keep a stage at 32 KiB, so a FOUR-stage ring (128 KiB) holds the same
// two-tiles-ahead prefetch and staggered wave groups as the three-stage ring of the 64-deep kernel.  Epilogue through LDS in
// two halves of the wave's 64x128 block.  Same arithmetic per output element (k ascending in steps of 32 inside the MFMA):
// results are bit-identical to gemm_bf16_p3_kernel's.
// ---------------------------------------------------------------------------------------------------
constexpr int W2_BM = 256, W2_BN = 256, W2_BK = 32;
constexpr int W2_A_BYTES = W2_BM * W2_BK * 2;      // 16 KiB
constexpr int W2_B_BYTES = W2_BN * W2_BK * 2;      // 16 KiB
constexpr int W2_STAGE = W2_A_BYTES + W2_B_BYTES;  // 32 KiB
constexpr int W2_NSTAGE = 4;
constexpr int W2_EPI_RS = 272;                     // bytes per parked row (64 fp32 + pad)
constexpr int W2_LDS = (W2_NSTAGE * W2_STAGE > 8 * 64 * W2_EPI_RS) ? W2_NSTAGE * W2_STAGE : 8 * 64 * W2_EPI_RS;

// XCD-aware bijective remap of the linear workgroup id (same formula as the other kernels)
__device__ __forceinline__ int w256_tile_id() {
    const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// AHEAD: how many K-tiles the LDS-DMA runs ahead of the MFMAs (2 or 3).  With the four-stage ring the DMA of tile t+3 overwrites
// the stage of tile t-1, whose fragment reads both wave groups completed before the barrier that opens this interval (group B
// reads tile t-1 in the interval before group A's H1(t)).
template <typename TOUT, int AHEAD>
__device__ __forceinline__ void w256_body(const GemmP& p, const int tile, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;              // wave tile: rows wm*64 .. +64, columns wn*128 .. +128
    int64_t m0, n0;
    {
        const int W = p.blk48 > 0 ? p.blk48 : 8;        // strip-ordered ids, as in gemm_bf16_p3_kernel
        const int ssz = p.tiles_m * W, strip = tile / ssz, rem = tile - strip * ssz;
        const int w = p.tiles_n - strip * W < W ? p.tiles_n - strip * W : W;
        const int mi = rem / w;
        m0 = (int64_t)mi * W2_BM;
        n0 = (int64_t)(strip * W + (rem - mi * w)) * W2_BN;
    }
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.a);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.b);
    TOUT* C = reinterpret_cast<TOUT*>(p.c);
    // LDS-DMA pieces of this wave (1 KiB = 2 k-rows of 512 B): A pieces wid, wid + 8; B pieces wid, wid + 8
    const bf16_t* ga[2];
    const bf16_t* gb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        ga[q] = piece_src<false, 2 * W2_BM>(A, p.lda, m0, p.M, wid + 8 * q, lane);
        gb[q] = piece_src<false, 2 * W2_BN>(B, p.ldb, n0, p.N, wid + 8 * q, lane);
    }
    const int64_t astep = (int64_t)W2_BK * p.lda, bstep = (int64_t)W2_BK * p.ldb;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint32_t ba[4], bb0[4], bb1[4];
    frag_bases<false, 2 * W2_BM>(wm * 64, lane, ba);
    frag_bases<false, 2 * W2_BN>(wn * 128, lane, bb0);
    frag_bases<false, 2 * W2_BN>(wn * 128 + 64, lane, bb1);
    const uint32_t lds_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int nt = (int)(p.K / W2_BK);

    auto issue_a = [&](int stage) {
        char* sa = smem + stage * W2_STAGE + wid * 1024;
#pragma unroll
        for (int q = 0; q < 2; ++q) { glds16(ga[q], sa + q * 8192); ga[q] += astep; }
    };
    auto issue_b = [&](int stage, int q) {
        char* sb = smem + stage * W2_STAGE + W2_A_BYTES + wid * 1024;
        glds16(gb[q], sb + q * 8192);
        gb[q] += bstep;
    };
    // staggered two-group schedule of gemm_bf16_p3_kernel; hazards with FOUR stages: the DMA of tile t+2 overwrites the stage
    // of tile t-2, whose fragment reads completed two barrier pairs ago in either group.
    const bool grp_b = wid >= 4;
    issue_a(0); issue_b(0, 0); issue_b(0, 1);
    if (nt > 1) { issue_a(1); issue_b(1, 0); issue_b(1, 1); }
    if (AHEAD == 3 && nt > 2) { issue_a(2); issue_b(2, 0); issue_b(2, 1); }
    // tile 0 has landed when only the younger tiles' pieces (4 per tile and wave) are outstanding
    if (AHEAD == 3 && nt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp_b) __builtin_amdgcn_s_barrier();
    int s_cur = 0, s_pre = AHEAD;
    for (int t = 0; t < nt; ++t) {
        // ---------------- H1(t): DMA of tile t+AHEAD (A half), fragments of tile t ----------------
        const bool more = t + AHEAD < nt;
        if (more) issue_a(s_pre);
        const uint32_t sao = lds_base + (uint32_t)(s_cur * W2_STAGE);
        const uint32_t sbo = sao + W2_A_BYTES;
        bf16x8 fa[4], fb[8];
        fa[0] = frag_read<false, 2 * W2_BM, 0, 0>(ba, sao);
        fa[1] = frag_read<false, 2 * W2_BM, 0, 1>(ba, sao);
        fa[2] = frag_read<false, 2 * W2_BM, 0, 2>(ba, sao);
        fa[3] = frag_read<false, 2 * W2_BM, 0, 3>(ba, sao);
        fb[0] = frag_read<false, 2 * W2_BN, 0, 0>(bb0, sbo);
        fb[1] = frag_read<false, 2 * W2_BN, 0, 1>(bb0, sbo);
        fb[2] = frag_read<false, 2 * W2_BN, 0, 2>(bb0, sbo);
        fb[3] = frag_read<false, 2 * W2_BN, 0, 3>(bb0, sbo);
        fb[4] = frag_read<false, 2 * W2_BN, 0, 0>(bb1, sbo);
        fb[5] = frag_read<false, 2 * W2_BN, 0, 1>(bb1, sbo);
        fb[6] = frag_read<false, 2 * W2_BN, 0, 2>(bb1, sbo);
        fb[7] = frag_read<false, 2 * W2_BN, 0, 3>(bb1, sbo);
        if (t + 1 < nt) {
            // my four pieces of tile t+1 have landed; younger pieces stay in flight: the two A pieces just issued, and with
            // AHEAD == 3 the four of tile t+2
            if (AHEAD == 3) {
                if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---------------- H2(t): 32 MFMAs on registers, the B half of tile t+2's DMA in between ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            if (more && i == 0) issue_b(s_pre, 0);
            if (more && i == 2) issue_b(s_pre, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        s_cur = (s_cur + 1) & 3;
        s_pre = (s_pre + 1) & 3;
    }
    if (!grp_b) __builtin_amdgcn_s_barrier();
    // ---- epilogue through LDS, the wave's 64 x 128 block in two halves of 64 x 64 (same scheme as gemm_bf16_p3_kernel) ------
    char* epi = smem + wid * (64 * W2_EPI_RS);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            const int64_t nb = n0 + wn * 128 + h * 64 + j * 16 + (lane >> 4) * 4;
            if (p.bias && nb < p.N) bv = *reinterpret_cast<const f32x4*>(p.bias + nb);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][h * 4 + j][r] * p.alpha;
                    if (p.bias) v[r] += bv[r];
                }
                *reinterpret_cast<f32x4*>(epi + (i * 16 + (lane & 15)) * W2_EPI_RS + (j * 16 + (lane >> 4) * 4) * 4) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t n = n0 + wn * 128 + h * 64 + (lane & 7) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int row = q * 8 + (lane >> 3);
            const int64_t m = m0 + wm * 64 + row;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(epi + row * W2_EPI_RS + (lane & 7) * 32);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(epi + row * W2_EPI_RS + (lane & 7) * 32 + 16);
            if (m >= p.M || n >= p.N) continue;
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            TOUT* dst = C + m * p.ldc + n;
            if (p.addend) {
                const TOUT* ad = reinterpret_cast<const TOUT*>(p.addend) + m * p.ldc + n;
                float a8[8];
                Vec8<TOUT>::load(ad, a8);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += a8[r];
            }
            Vec8<TOUT>::store(dst, v);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the second half re-uses the wave's parking region
    }
}

template <typename TOUT, int AHEAD>
__global__ void __launch_bounds__(512) gemm_bf16_w256_kernel(GemmP p) {
    __shared__ __attribute__((aligned(16))) char smem[W2_LDS];
    w256_body<TOUT, AHEAD>(p, w256_tile_id(), smem);
}

// Several weight-gradient problems (same kernel, same tile time: K = tokens for all of them) in ONE launch: workgroup ids
// [start[i], start[i+1]) belong to problem i.  Separate launches each end in a partial round of the 256 CUs; together the
// tiles of a LLaMA-7B step's four wgrads make 12.06 rounds instead of 3 + 6 + 1 + 3 = 13 (the host peels the last 16 tiles off
// and runs them with a split contraction, oq_wgrad_group), LLaMA-2-13B's 18.9 instead of 21.
struct GemmGroupP {
    GemmP p[4];
    int start[5];
    int n;
};

template <typename TOUT>
__global__ void __launch_bounds__(512) gemm_bf16_w256_group_kernel(GemmGroupP g) {
    __shared__ __attribute__((aligned(16))) char smem[W2_LDS];
    int tile = w256_tile_id();
    int gi = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < g.n && tile >= g.start[i]) gi = i;
    tile -= g.start[gi];
    GemmP p = g.p[0];                   // (wave-uniform: scalar registers)
    if (gi == 1) p = g.p[1];
    else if (gi == 2) p = g.p[2];
    else if (gi == 3) p = g.p[3];
    w256_body<TOUT, 2>(p, tile, smem);
}

// ---------------------------------------------------------------------------------------------------
// Persistent variant of the 256x128x64 kernel: one workgroup per CU walks a list of output tiles and the LDS-DMA
// stream of K-tiles never drains at a tile boundary (the first K-tiles of the next output tile are already in flight
// while the current tile finishes and stores its result).  Removes the per-tile pipeline fill, which dominated
// short-K problems (wgrad K = 2048: 32 K-tiles; attention scores K = 128: 2 K-tiles).  Same staggered two-group
// schedule, ring and hazard reasoning as gemm_bf16_p3_kernel; "g" below is the workgroup's global K-tile counter.
// ---------------------------------------------------------------------------------------------------
struct TileCur {          // wave-uniform cursor over this workgroup's tile list
    int k;                // index into the list (tile id = remap(blockIdx.x + k * gridDim.x))
    int nt;               // K-tiles of the current tile
    int t0;               // first K-tile (causal mode 3)
    int64_t m0, n0, z;
    bool valid;
};

__device__ __forceinline__ void tile_decode(const GemmP& p, int total, int tiles_per_z, int w, TileCur& c) {
    // XCD-aware bijective remap of the linear work id (same formula as the non-persistent kernels)
    const int xcd = w & 7, q = total >> 3, r = total & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
    c.z = tile / tiles_per_z;
    const int tz = tile % tiles_per_z;
    if (p.nmajor) { c.n0 = (int64_t)(tz / p.tiles_m) * P3_BN; c.m0 = (int64_t)(tz % p.tiles_m) * P3_BM; }
    else { c.m0 = (int64_t)(tz / p.tiles_n) * P3_BM; c.n0 = (int64_t)(tz % p.tiles_n) * P3_BN; }
    int64_t kend = p.K;
    if (p.tri == 2 && c.m0 + P3_BM < kend) kend = c.m0 + P3_BM;
    c.t0 = p.tri == 3 ? (int)(c.m0 / BK) : 0;
    c.nt = (int)(kend / BK) - c.t0;
    if (p.tri == 1 && c.n0 >= c.m0 + P3_BM) c.nt = 0;          // tile above the diagonal: nothing to do
}

// advance to the next tile that has work; c.valid = false when the list is exhausted
__device__ __forceinline__ void tile_next(const GemmP& p, int total, int tiles_per_z, TileCur& c) {
    for (;;) {
        ++c.k;
        const int w = (int)blockIdx.x + c.k * (int)gridDim.x;
        if (w >= total) { c.valid = false; c.nt = 0; return; }
        tile_decode(p, total, tiles_per_z, w, c);
        if (c.nt > 0) { c.valid = true; return; }
    }
}

template <bool AKC, bool BKC, typename TOUT>
__global__ void __launch_bounds__(512) gemm_bf16_pp_kernel(GemmP p, int total, int tiles_per_z) {
    __shared__ __attribute__((aligned(16))) char smem[3 * P3_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const bool grp_b = wid >= 4;
    const int64_t astep = AKC ? BK : (int64_t)BK * p.lda;
    const int64_t bstep = BKC ? BK : (int64_t)BK * p.ldb;

    uint32_t ba[4], bb[4];
    frag_bases<AKC, 2 * P3_BM>(wm * 64, lane, ba);
    frag_bases<BKC, 2 * P3_BN>(wn * 64, lane, bb);
    const uint32_t lds_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    // ---- DMA cursor (runs two K-tiles ahead of the compute cursor) ------------------------------------------
    TileCur dc;
    dc.k = -1; dc.valid = true; dc.nt = 0;
    int d_rem = 0;                       // K-tiles of dc's tile not yet issued
    const bf16_t* ga[4];
    const bf16_t* gb[2];
    int d_stage = 0;
    auto dma_fetch_tile = [&]() {        // move dc to the next tile with work and set up its piece pointers
        tile_next(p, total, tiles_per_z, dc);
        if (!dc.valid) return;
        const int64_t bo = dc.z / p.batch_i, bi = dc.z % p.batch_i;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(p.a) + bo * p.sa_o + bi * p.sa_i;
        const bf16_t* B = reinterpret_cast<const bf16_t*>(p.b) + bo * p.sb_o + bi * p.sb_i;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            ga[q] = piece_src<AKC, 2 * P3_BM>(A, p.lda, dc.m0, p.M, wid + 8 * q, lane) + (int64_t)dc.t0 * astep;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            gb[q] = piece_src<BKC, 2 * P3_BN>(B, p.ldb, dc.n0, p.N, wid + 8 * q, lane) + (int64_t)dc.t0 * bstep;
        d_rem = dc.nt;
    };
    // returns false when there is nothing left to issue.  part 0: first half (3 A pieces), 1..3: one piece each,
    // 4: all six pieces
    bool d_have = false;                 // a K-tile is selected for issue (its first half may already be out)
    auto dma_select = [&]() -> bool {    // make sure a K-tile is selected
        if (d_have) return true;
        while (d_rem == 0) {
            if (!dc.valid) return false;
            dma_fetch_tile();
            if (!dc.valid) return false;
        }
        d_have = true;
        return true;
    };
    auto dma_issue_h1 = [&]() {
        char* sa = smem + d_stage * P3_STAGE + wid * 1024;
#pragma unroll
        for (int q = 0; q < 3; ++q) { glds16(ga[q], sa + q * 8192); ga[q] += astep; }
    };
    auto dma_issue_h2 = [&](int which) {
        char* sa = smem + d_stage * P3_STAGE + wid * 1024;
        if (which == 0) { glds16(ga[3], sa + 3 * 8192); ga[3] += astep; }
        else { glds16(gb[which - 1], sa + P3_A_BYTES + (which - 1) * 8192); gb[which - 1] += bstep; }
    };
    auto dma_done_tile = [&]() {         // the selected K-tile is completely issued
        d_have = false;
        --d_rem;
        d_stage = d_stage == 2 ? 0 : d_stage + 1;
    };
    auto dma_issue_full = [&]() {
        dma_issue_h1();
        dma_issue_h2(0); dma_issue_h2(1); dma_issue_h2(2);
        dma_done_tile();
    };

    // ---- compute cursor ---------------------------------------------------------------------------------------
    TileCur cc;
    cc.k = -1; cc.valid = true; cc.nt = 0;
    tile_next(p, total, tiles_per_z, cc);
    if (!cc.valid) return;               // this workgroup has no work at all (uniform over the workgroup)

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: K-tiles g = 0 and g = 1 completely issued
    int issued = 0;
    if (dma_select()) { dma_issue_full(); ++issued; }
    if (dma_select()) { dma_issue_full(); ++issued; }
    if (issued == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (grp_b) __builtin_amdgcn_s_barrier();

    int s_cur = 0;
    int t = 0;                           // K-tile index inside the compute tile
    for (;;) {
        // ---------------- H1(g) ----------------
        const bool more2 = dma_select();           // is there a K-tile g+2 ?
        if (more2) dma_issue_h1();
        const uint32_t sao = lds_base + (uint32_t)(s_cur * P3_STAGE);
        const uint32_t sbo = sao + P3_A_BYTES;
        bf16x8 fa[2][4], fb[2][4];
        fa[0][0] = frag_read<AKC, 2 * P3_BM, 0, 0>(ba, sao);
        fa[0][1] = frag_read<AKC, 2 * P3_BM, 0, 1>(ba, sao);
        fa[0][2] = frag_read<AKC, 2 * P3_BM, 0, 2>(ba, sao);
        fa[0][3] = frag_read<AKC, 2 * P3_BM, 0, 3>(ba, sao);
        fb[0][0] = frag_read<BKC, 2 * P3_BN, 0, 0>(bb, sbo);
        fb[0][1] = frag_read<BKC, 2 * P3_BN, 0, 1>(bb, sbo);
        fb[0][2] = frag_read<BKC, 2 * P3_BN, 0, 2>(bb, sbo);
        fb[0][3] = frag_read<BKC, 2 * P3_BN, 0, 3>(bb, sbo);
        fa[1][0] = frag_read<AKC, 2 * P3_BM, 1, 0>(ba, sao);
        fa[1][1] = frag_read<AKC, 2 * P3_BM, 1, 1>(ba, sao);
        fa[1][2] = frag_read<AKC, 2 * P3_BM, 1, 2>(ba, sao);
        fa[1][3] = frag_read<AKC, 2 * P3_BM, 1, 3>(ba, sao);
        fb[1][0] = frag_read<BKC, 2 * P3_BN, 1, 0>(bb, sbo);
        fb[1][1] = frag_read<BKC, 2 * P3_BN, 1, 1>(bb, sbo);
        fb[1][2] = frag_read<BKC, 2 * P3_BN, 1, 2>(bb, sbo);
        fb[1][3] = frag_read<BKC, 2 * P3_BN, 1, 3>(bb, sbo);
        // K-tile g+1 (if it exists) was issued completely one interval pair ago; only g+2's first half is newer.
        // (epilogue stores issued in between only make the counted wait stricter: VMEM ops retire in order)
        if (more2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---------------- H2(g) ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
                if (more2 && ks * 4 + i == 1) dma_issue_h2(0);
                if (more2 && ks * 4 + i == 3) dma_issue_h2(1);
                if (more2 && ks * 4 + i == 5) dma_issue_h2(2);
                __builtin_amdgcn_sched_barrier(0);
            }
        __builtin_amdgcn_s_setprio(0);
        if (more2) dma_done_tile();
        s_cur = s_cur == 2 ? 0 : s_cur + 1;
        ++t;
        bool finished = false;
        if (t == cc.nt) {
            // ---- epilogue of the compute tile (its successor's K-tiles are already streaming in) ----
            const int64_t bo = cc.z / p.batch_i, bi = cc.z % p.batch_i;
            TOUT* C = reinterpret_cast<TOUT*>(p.c) + bo * p.sc_o + bi * p.sc_i;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t m = cc.m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t n = cc.n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
                    if (m < p.M && n < p.N) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = acc[i][j][r] * p.alpha;
                            if (p.bias) v[r] += p.bias[n + r];
                        }
                        if (p.addend) {
                            const TOUT* ad = reinterpret_cast<const TOUT*>(p.addend) + (C - reinterpret_cast<TOUT*>(p.c)) + m * p.ldc + n;
                            if constexpr (sizeof(TOUT) == 4) {
                                const f32x4 av = *reinterpret_cast<const f32x4*>(ad);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] += av[r];
                            } else {
                                typedef __attribute__((ext_vector_type(4))) __bf16 bf4a;
                                const bf4a av = *reinterpret_cast<const bf4a*>(ad);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] += (float)av[r];
                            }
                        }
                        TOUT* dst = C + m * p.ldc + n;
                        if constexpr (sizeof(TOUT) == 4) {
                            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                        } else {
                            typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
                            *reinterpret_cast<bf4*>(dst) = bf4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        }
                    }
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            t = 0;
            tile_next(p, total, tiles_per_z, cc);
            finished = !cc.valid;
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (finished) break;
    }
    if (!grp_b) __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------------
// exact f32 path
// ---------------------------------------------------------------------------------------------------
constexpr int FM = 64, FN = 64, FK = 16;

template <typename TOUT>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmP p) {
    __shared__ float As[FK][FM + 1];
    __shared__ float Bs[FK][FN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tile = blockIdx.x;
    const int64_t m0 = (int64_t)(tile / p.tiles_n) * FM, n0 = (int64_t)(tile % p.tiles_n) * FN;
    const int64_t bo = blockIdx.z / p.batch_i, bi = blockIdx.z % p.batch_i;
    const float* A = reinterpret_cast<const float*>(p.a) + bo * p.sa_o + bi * p.sa_i;
    const float* B = reinterpret_cast<const float*>(p.b) + bo * p.sb_o + bi * p.sb_i;
    TOUT* C = reinterpret_cast<TOUT*>(p.c) + bo * p.sc_o + bi * p.sc_i;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (p.tri == 1 && n0 >= m0 + FM) return;
    int64_t kend = p.K;
    if (p.tri == 2 && m0 + FM < kend) kend = m0 + FM;
    const int64_t kbeg = p.tri == 3 ? m0 : 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += FK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e;   // 0..1023
            int kk, ii;
            // pick the fast-varying index along the contiguous global dim
            if (p.a_kc) { kk = idx & 15; ii = idx >> 4; } else { ii = idx & 63; kk = idx >> 6; }
            {
                const int64_t m = m0 + ii, k = k0 + kk;
                float v = 0.f;
                if (m < p.M && k < p.K) v = p.a_kc ? A[m * p.lda + k] : A[k * p.lda + m];
                As[kk][ii] = v;
            }
            if (p.b_kc) { kk = idx & 15; ii = idx >> 4; } else { ii = idx & 63; kk = idx >> 6; }
            {
                const int64_t n = n0 + ii, k = k0 + kk;
                float v = 0.f;
                if (n < p.N && k < p.K) v = p.b_kc ? B[n * p.ldb + k] : B[k * p.ldb + n];
                Bs[kk][ii] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < FK / 4; ++ks) {
            const int k = ks * 4 + (lane >> 4);
            float fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = As[k][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = Bs[k][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[row = (lane>>4)*4 + r][col = lane&15] with A rows = m, B cols = n
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
                const int64_t n = n0 + wn * 32 + j * 16 + (lane & 15);
                if (m < p.M && n < p.N) {
                    float v = acc[i][j][r] * p.alpha;
                    if (p.bias) v += p.bias[n];
                    if (p.addend) v += (float)(reinterpret_cast<const TOUT*>(p.addend) + (C - reinterpret_cast<TOUT*>(p.c)))[m * p.ldc + n];
                    C[m * p.ldc + n] = (TOUT)v;
                }
            }
}

thread_local int g_splitk_request = 1;       // set by oq_gemm_ws around its call of the launcher

// ---- split contraction: out = alpha * sum_s ws[s] + bias + addend, parts added in index order (deterministic) ------------
template <typename TOUT>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* ws, int S, int64_t M, int64_t N, TOUT* c, int64_t ldc,
                                                            const float* bias, const TOUT* addend, float alpha) {
    const int64_t nv = N / 8, total = M * nv;
    for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = v / nv, n = (v - m * nv) * 8;
        float acc[8];
        Vec8<float>::load(ws + m * N + n, acc);
        for (int s = 1; s < S; ++s) {
            float t[8];
            Vec8<float>::load(ws + ((int64_t)s * M + m) * N + n, t);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += t[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] *= alpha;
            if (bias) acc[i] += bias[n + i];
        }
        if (addend) {
            float a[8];
            Vec8<TOUT>::load(addend + m * ldc + n, a);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += a[i];
        }
        Vec8<TOUT>::store(c + m * ldc + n, acc);
    }
}

// How many parts the contraction of a bf16 problem is worth splitting into (1 = not at all): launches whose tile count is a
// little more than a whole number of rounds of the 256 CUs (LLaMA-2-13B: N = 5120 -> 320 tiles = 1.25 rounds of 216-432 K-tiles)
// run the ragged round at full length; S parts give ceil(S * tiles / 256) rounds of 1/S the length, plus per-tile fixed costs
// (~8 K-tiles' worth each, tools/gemm_stamps.hip) and one pass over S fp32 partial outputs.
int splitk_parts(int64_t M, int64_t N, int64_t K, int64_t batch, int tri_mode) {
    if (dbg_env_i("OQ_GEMM_SPLITK", 1) == 0 || tri_mode != 0 || batch != 1) return 1;
    if (dbg_env_i("OQ_GEMM_NO_P3", 0) || dbg_env_i("OQ_GEMM_PERSIST", 2) == 1) return 1;      // only the 256x128 kernel splits
    if (K % BK != 0 || M % 8 != 0 || N % 8 != 0 || M < 128 || N < 128) return 1;
    const int64_t tiles = ((M + P3_BM - 1) / P3_BM) * ((N + P3_BN - 1) / P3_BN), nt = K / BK;
    // costs in K-tile times (~0.8 us): 12 per tile for its fixed costs incl. the larger fp32 store, and the reduce pass at ~2.7 TB/s
    // effective; calibrated on the three 13B launches (tools/ab_gemm_13b.py: K = 27648 +25 %, 15360 +8 %, 13824 -2 % -> not split)
    const double fixed = 12.0;
    auto rounds = [&](int64_t t) { return (double)((t + 255) / 256); };
    const double base = rounds(tiles) * (nt + fixed);
    int best = 1;
    double best_cost = base * 0.88;                      // at least 12 % better by this model, or not at all
    for (int S = 2; S <= 4; ++S) {
        if (nt % S != 0 || nt / S < 32) continue;
        const double reduce = 1.5 * ((double)(S + 1) * M * N * 4.0 / 4.0e12) / 0.85e-6;
        const double cost = rounds(tiles * S) * ((double)nt / S + fixed) + reduce;
        if (cost < best_cost) { best_cost = cost; best = S; }
    }
    return best;
}

}  // namespace

extern "C" int64_t oq_gemm_workspace(int64_t M, int64_t N, int64_t K, int in_dtype, int64_t batch, int tri_mode) {
    if (in_dtype != OQ_BF16 || M <= 0 || N <= 0 || K <= 0) return 0;
    const int S = splitk_parts(M, N, K, batch, tri_mode);
    return S > 1 ? (int64_t)S * M * N * 4 : 0;
}

static int oq_gemm_impl(const void* a, const void* bm, void* c, const float* bias, const void* addend, int64_t M, int64_t N,
                        int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype,
                        float alpha, int64_t batch_o, int64_t batch_i, int64_t sa_o, int64_t sa_i, int64_t sb_o, int64_t sb_i,
                        int64_t sc_o, int64_t sc_i, int tri_mode, void* stream);

extern "C" int oq_gemm_ws(const void* a, const void* bm, void* c, const float* bias, const void* addend, int64_t M, int64_t N,
                          int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype,
                          float alpha, int64_t batch_o, int64_t batch_i, int64_t sa_o, int64_t sa_i, int64_t sb_o, int64_t sb_i,
                          int64_t sc_o, int64_t sc_i, int tri_mode, void* workspace, int64_t workspace_bytes, void* stream) {
    const int S = (in_dtype == OQ_BF16 && workspace && M > 0 && N > 0 && K > 0) ? splitk_parts(M, N, K, batch_o * batch_i, tri_mode) : 1;
    if (S <= 1 || workspace_bytes < (int64_t)S * M * N * 4 || !oq_aligned16(workspace) || ldc % 8 != 0 || !oq_aligned16(c) ||
        (addend && !oq_aligned16(addend)) || (out_dtype != OQ_F32 && out_dtype != OQ_BF16))
        return oq_gemm_impl(a, bm, c, bias, addend, M, N, K, lda, ldb, ldc, a_kc, b_kc, in_dtype, out_dtype, alpha, batch_o, batch_i,
                            sa_o, sa_i, sb_o, sb_i, sc_o, sc_i, tri_mode, stream);
    // S launches' worth of tiles in ONE launch: fp32 partial outputs [S][M][N] in the workspace, then the reduce
    g_splitk_request = S;
    const int rc = oq_gemm_impl(a, bm, workspace, nullptr, nullptr, M, N, K, lda, ldb, N, a_kc, b_kc, in_dtype, OQ_F32, 1.0f, 1, 1,
                                0, 0, 0, 0, 0, 0, 0, stream);
    g_splitk_request = 1;
    if (rc != OQ_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nvec = M * (N / 8);
    const unsigned grid = (unsigned)(nvec / 256 + 1 < 4096 ? nvec / 256 + 1 : 4096);
    if (out_dtype == OQ_F32)
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)workspace, S, M, N, (float*)c, ldc,
                           bias, (const float*)addend, alpha);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)workspace, S, M, N, (bf16_t*)c, ldc,
                           bias, (const bf16_t*)addend, alpha);
    OQ_CHECK_LAUNCH("oq_gemm_ws(reduce)");
    return OQ_OK;
}

extern "C" int oq_gemm(const void* a, const void* bm, void* c, const float* bias, const void* addend, int64_t M, int64_t N,
                       int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype,
                       float alpha, int64_t batch_o, int64_t batch_i, int64_t sa_o, int64_t sa_i, int64_t sb_o, int64_t sb_i,
                       int64_t sc_o, int64_t sc_i, int tri_mode, void* stream) {
    return oq_gemm_impl(a, bm, c, bias, addend, M, N, K, lda, ldb, ldc, a_kc, b_kc, in_dtype, out_dtype, alpha, batch_o, batch_i, sa_o,
                        sa_i, sb_o, sb_i, sc_o, sc_i, tri_mode, stream);
}

static int oq_gemm_impl(const void* a, const void* bm, void* c, const float* bias, const void* addend, int64_t M, int64_t N,
                       int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int a_kc, int b_kc, int in_dtype, int out_dtype,
                       float alpha, int64_t batch_o, int64_t batch_i, int64_t sa_o, int64_t sa_i, int64_t sb_o,
                       int64_t sb_i, int64_t sc_o, int64_t sc_i, int tri_mode, void* stream) {
    OQ_CHECK_ARG(a && bm && c, "oq_gemm: null operand");
    OQ_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch_o > 0 && batch_i > 0, "oq_gemm: empty problem M=%lld N=%lld K=%lld",
                 (long long)M, (long long)N, (long long)K);
    OQ_CHECK_ARG(batch_o * batch_i <= 65535, "oq_gemm: batch %lld too large", (long long)(batch_o * batch_i));
    OQ_CHECK_ARG(out_dtype == OQ_F32 || out_dtype == OQ_BF16, "oq_gemm: out dtype %d", out_dtype);
    GemmP p{};
    p.a = a; p.b = bm; p.c = c; p.bias = bias; p.addend = addend; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.a_kc = a_kc; p.b_kc = b_kc; p.alpha = alpha; p.batch_i = batch_i;
    p.sa_o = sa_o; p.sa_i = sa_i; p.sb_o = sb_o; p.sb_i = sb_i; p.sc_o = sc_o; p.sc_i = sc_i;
    OQ_CHECK_ARG(tri_mode >= 0 && tri_mode <= 3, "oq_gemm: tri_mode %d", tri_mode);
    OQ_CHECK_ARG(tri_mode == 0 || (tri_mode == 1 ? M == N : true), "oq_gemm: tri_mode 1 needs a square output");
    OQ_CHECK_ARG(tri_mode < 2 || (K % 256 == 0 || K < 256), "oq_gemm: causal contraction needs K %% 256 == 0 (or K < 256)");
    p.tri = tri_mode;
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == OQ_BF16) {
        // 16-byte vector loads / 8-16-byte stores
        OQ_CHECK_ARG(oq_aligned16(a) && oq_aligned16(bm) && oq_aligned16(c), "oq_gemm(bf16): base pointers 16-B aligned");
        OQ_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "oq_gemm(bf16): lda/ldb %% 8, ldc %% 4");
        OQ_CHECK_ARG(sa_o % 8 == 0 && sa_i % 8 == 0 && sb_o % 8 == 0 && sb_i % 8 == 0 && sc_o % 4 == 0 && sc_i % 4 == 0,
                     "oq_gemm(bf16): batch strides must keep 16-B alignment");
        OQ_CHECK_ARG((a_kc ? K : M) % 8 == 0 && (b_kc ? K : N) % 8 == 0 && N % 4 == 0,
                     "oq_gemm(bf16): contiguous dims must be multiples of 8 (M=%lld N=%lld K=%lld)", (long long)M,
                     (long long)N, (long long)K);
        const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
        p.tiles_n = (int)tn;
        p.tiles_m = (int)tm;
        p.nmajor = N >= M ? 1 : 0;
        dim3 grid((unsigned)(tm * tn), 1, (unsigned)(batch_o * batch_i));
        // fast path: no predication in the K loop (row clamping needs at least one full vector per operand)
        const bool fast = (K % BK == 0) && M >= 8 && N >= 8;
#define LAUNCH_BF16(AK, BK_, T)                                                                         \
    do {                                                                                                \
        if (fast) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK_, T, true>), grid, dim3(256), 0, st, p);  \
        else hipLaunchKernelGGL((gemm_bf16_kernel<AK, BK_, T, false>), grid, dim3(256), 0, st, p);      \
    } while (0)
        const int key = (a_kc ? 4 : 0) | (b_kc ? 2 : 0) | (out_dtype == OQ_F32 ? 1 : 0);
        const int use_p3 = dbg_env_i("OQ_GEMM_NO_P3", 0) ? 0 : 1;
        if (use_p3 && fast && M >= 128 && N >= 128) {
            const int64_t tm3 = (M + P3_BM - 1) / P3_BM, tn3 = (N + P3_BN - 1) / P3_BN;
            p.tiles_n = (int)tn3;
            p.tiles_m = (int)tm3;
            p.splitk = g_splitk_request;
            p.tiles_real = (int)(tm3 * tn3);
            dim3 grid3((unsigned)(tm3 * tn3 * (p.splitk > 1 ? p.splitk : 1)), 1, (unsigned)(batch_o * batch_i));
            // persistent tile loop: wins where the per-tile pipeline fill dominates (short K, batched attention
            // problems); the plain kernel's leaner main loop wins on the long-K linears (measured, tools/ab_gemm.py)
            const int persist = dbg_env_i("OQ_GEMM_PERSIST", 2);
            if (persist == 1 || (persist == 2 && (K <= 256 || batch_o * batch_i > 1))) {
                const int tiles_per_z = (int)(tm3 * tn3);
                const int64_t total64 = (int64_t)tiles_per_z * batch_o * batch_i;
                OQ_CHECK_ARG(total64 < (1ll << 30), "oq_gemm: too many tiles");
                const int total = (int)total64;
                static int n_cu = 0;
                if (n_cu == 0) {
                    int dev = 0;
                    hipDeviceProp_t prop;
                    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                        n_cu = prop.multiProcessorCount;
                    if (n_cu <= 0) n_cu = 256;
                }
                const int gridp = total < n_cu ? total : n_cu;     // one 144 KiB-LDS workgroup per CU
#define LAUNCH_PP(AK, BK_, T) hipLaunchKernelGGL((gemm_bf16_pp_kernel<AK, BK_, T>), dim3(gridp), dim3(512), 0, st, p, total, tiles_per_z)
                switch (key) {
                    case 0: LAUNCH_PP(false, false, bf16_t); break;
                    case 1: LAUNCH_PP(false, false, float); break;
                    case 2: LAUNCH_PP(false, true, bf16_t); break;
                    case 3: LAUNCH_PP(false, true, float); break;
                    case 4: LAUNCH_PP(true, false, bf16_t); break;
                    case 5: LAUNCH_PP(true, false, float); break;
                    case 6: LAUNCH_PP(true, true, bf16_t); break;
                    case 7: LAUNCH_PP(true, true, float); break;
                }
                OQ_CHECK_LAUNCH("oq_gemm(pp)");
                return OQ_OK;
            }
            // whole-line stores need 8-element granularity of every output address (and of the addend, which shares them)
            p.epi_lds = (dbg_env_i("OQ_GEMM_EPI_LDS", 1) != 0 && N % 8 == 0 && ldc % 8 == 0 && sc_o % 8 == 0 && sc_i % 8 == 0 &&
                         (reinterpret_cast<uintptr_t>(c) & 15) == 0 && (!addend || (reinterpret_cast<uintptr_t>(addend) & 15) == 0) &&
                         (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) ? 1 : 0;
            // weight-gradient shape class (both operands k-strided, one problem, at least one full round of 256 x 256 tiles):
            // the 256x256x32 kernel
            {
                const int64_t tm2 = (M + W2_BM - 1) / W2_BM, tn2 = (N + W2_BN - 1) / W2_BN;
                // rounds of the 256 CUs: a 256 x 256 tile takes ~1.76x the time of a 256 x 128 tile at K = 2048 (measured: -12 % at
                // whole rounds); ragged last rounds decide the rest
                const int64_t r1 = (tm3 * tn3 + 255) / 256, r2 = (tm2 * tn2 + 255) / 256;
                const bool pays = dbg_env_i("OQ_GEMM_W256_MIN_TILES", 256) < 256 || (double)r2 * 1.76 < (double)r1 + 1e-9;
                if (!a_kc && !b_kc && tri_mode == 0 && batch_o * batch_i == 1 && g_splitk_request <= 1 && p.epi_lds && pays &&
                    K % W2_BK == 0 && tm2 * tn2 >= dbg_env_i("OQ_GEMM_W256_MIN_TILES", 256) && dbg_env_i("OQ_GEMM_W256", 1) != 0) {
                    p.tiles_n = (int)tn2;
                    p.tiles_m = (int)tm2;
                    p.blk48 = 8;
                    dim3 gridw((unsigned)(tm2 * tn2));
                    const bool ahead3 = dbg_env_i("OQ_GEMM_W256_AHEAD", 3) == 3;
                    if (out_dtype == OQ_F32) {
                        if (ahead3) hipLaunchKernelGGL((gemm_bf16_w256_kernel<float, 3>), gridw, dim3(512), 0, st, p);
                        else hipLaunchKernelGGL((gemm_bf16_w256_kernel<float, 2>), gridw, dim3(512), 0, st, p);
                    } else {
                        if (ahead3) hipLaunchKernelGGL((gemm_bf16_w256_kernel<bf16_t, 3>), gridw, dim3(512), 0, st, p);
                        else hipLaunchKernelGGL((gemm_bf16_w256_kernel<bf16_t, 2>), gridw, dim3(512), 0, st, p);
                    }
                    OQ_CHECK_LAUNCH("oq_gemm(w256)");
                    return OQ_OK;
                }
            }
            {
                const int sw = dbg_env_i("OQ_GEMM_BLK48", 8);          // 0: plain n-/m-major order; else the strip width
                p.blk48 = (tri_mode == 0 && sw > 0) ? (sw == 1 ? 8 : sw) : 0;
            }
            const bool stagger = dbg_env_i("OQ_GEMM_STAGGER", 1) != 0;
            const bool split = dbg_env_i("OQ_GEMM_SPLIT", 1) != 0;
#define LAUNCH_P3(AK, BK_, T)                                                                                          \
    do {                                                                                                               \
        if (stagger && split) hipLaunchKernelGGL((gemm_bf16_p3_kernel<AK, BK_, T, true, true>), grid3, dim3(512), 0, st, p);  \
        else if (stagger) hipLaunchKernelGGL((gemm_bf16_p3_kernel<AK, BK_, T, true, false>), grid3, dim3(512), 0, st, p);     \
        else hipLaunchKernelGGL((gemm_bf16_p3_kernel<AK, BK_, T, false, false>), grid3, dim3(512), 0, st, p);                 \
    } while (0)
            switch (key) {
                case 0: LAUNCH_P3(false, false, bf16_t); break;
                case 1: LAUNCH_P3(false, false, float); break;
                case 2: LAUNCH_P3(false, true, bf16_t); break;
                case 3: LAUNCH_P3(false, true, float); break;
                case 4: LAUNCH_P3(true, false, bf16_t); break;
                case 5: LAUNCH_P3(true, false, float); break;
                case 6: LAUNCH_P3(true, true, bf16_t); break;
                case 7: LAUNCH_P3(true, true, float); break;
            }
            OQ_CHECK_LAUNCH("oq_gemm(p3)");
            return OQ_OK;
        }
        switch (key) {
            case 0: LAUNCH_BF16(false, false, bf16_t); break;
            case 1: LAUNCH_BF16(false, false, float); break;
            case 2: LAUNCH_BF16(false, true, bf16_t); break;
            case 3: LAUNCH_BF16(false, true, float); break;
            case 4: LAUNCH_BF16(true, false, bf16_t); break;
            case 5: LAUNCH_BF16(true, false, float); break;
            case 6: LAUNCH_BF16(true, true, bf16_t); break;
            case 7: LAUNCH_BF16(true, true, float); break;
        }
    } else if (in_dtype == OQ_F32) {
        const int64_t tm = (M + FM - 1) / FM, tn = (N + FN - 1) / FN;
        p.tiles_n = (int)tn;
        dim3 grid((unsigned)(tm * tn), 1, (unsigned)(batch_o * batch_i));
        if (out_dtype == OQ_F32)
            hipLaunchKernelGGL((gemm_f32_kernel<float>), grid, dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<bf16_t>), grid, dim3(256), 0, st, p);
    } else {
        oq_set_error("oq_gemm: in dtype %d unsupported", in_dtype);
        return OQ_E_UNSUPPORTED;
    }
    OQ_CHECK_LAUNCH("oq_gemm");
    return OQ_OK;
}

// ---------------------------------------------------------------------------------------------------
// The weight-gradient products of one backward pass as ONE launch of the 256x256x32 kernel (gemm_bf16_w256_group_kernel).
// item i: gw[N_i][K_i] = gy_i[T][N_i]^T x_i[T][K_i]  (bf16, both operands k-strided, no bias / addend).
// If the tiles of all items exceed a whole number of rounds of the CUs by a few tiles (<= a quarter round) and one item has a
// tile row / column of exactly that many tiles, that strip is peeled off and computed with a split contraction (256x128x64
// kernel, fp32 partials in `workspace`, fixed-order reduce): a quarter-length tail instead of a whole extra round.
// ---------------------------------------------------------------------------------------------------
struct OqWgradItem {
    const void* gy;
    const void* x;
    void* gw;
    int64_t N, K, T, ld_gy, ld_x, ld_gw;
};

static int wgrad_n_cu() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

static bool wgrad_item_groupable(const OqWgradItem& it) {
    return it.gy && it.x && it.gw && it.N >= 256 && it.K >= 256 && it.T >= 64 && it.T % W2_BK == 0 && it.N % 8 == 0 && it.K % 8 == 0 &&
           it.ld_gy % 8 == 0 && it.ld_x % 8 == 0 && it.ld_gw % 8 == 0 && oq_aligned16(it.gy) && oq_aligned16(it.x) && oq_aligned16(it.gw);
}

extern "C" int64_t oq_wgrad_group_workspace(const void* items_, int n) {
    // bytes the peeled strip's split contraction may need: 8 parts x (256 x the longest side) fp32
    const OqWgradItem* items = reinterpret_cast<const OqWgradItem*>(items_);
    int64_t mx = 0;
    for (int i = 0; i < n; ++i) {
        mx = items[i].N > mx ? items[i].N : mx;
        mx = items[i].K > mx ? items[i].K : mx;
    }
    return 8 * 256 * mx * 4;
}

extern "C" int oq_wgrad_group(const void* items_, int n, void* workspace, int64_t workspace_bytes, void* stream) {
    const OqWgradItem* items = reinterpret_cast<const OqWgradItem*>(items_);
    OQ_CHECK_ARG(items && n > 0 && n <= 16, "oq_wgrad_group: %d items (1 .. 16)", n);
    hipStream_t st = (hipStream_t)stream;
    auto single = [&](const OqWgradItem& it) {
        return oq_gemm_impl(it.gy, it.x, it.gw, nullptr, nullptr, it.N, it.K, it.T, it.ld_gy, it.ld_x, it.ld_gw, 0, 0, OQ_BF16, OQ_BF16,
                            1.0f, 1, 1, 0, 0, 0, 0, 0, 0, 0, stream);
    };
    bool ok = dbg_env_i("OQ_WGRAD_GROUP", 1) != 0 && n >= 2;
    for (int i = 0; i < n && ok; ++i) ok = wgrad_item_groupable(items[i]);
    if (!ok) {
        for (int i = 0; i < n; ++i)
            if (int rc = single(items[i])) return rc;
        return OQ_OK;
    }
    for (int base = 0; base < n; base += 4) {
        const int cnt = n - base < 4 ? n - base : 4;
        if (cnt == 1) {
            if (int rc = single(items[base])) return rc;
            continue;
        }
        OqWgradItem it[4];
        int64_t tm[4], tn[4], total = 0;
        for (int i = 0; i < cnt; ++i) {
            it[i] = items[base + i];
            tm[i] = (it[i].N + W2_BM - 1) / W2_BM;
            tn[i] = (it[i].K + W2_BN - 1) / W2_BN;
            total += tm[i] * tn[i];
        }
        // ---- peel a strip so that the rest is a whole number of rounds ----
        const int ncu = wgrad_n_cu();
        const int64_t rem = total % ncu;
        int peel_i = -1, peel_rows = 0, peel_cols = 0;       // tile rows (of gw) or tile columns taken off item peel_i
        if (rem != 0 && rem <= ncu / 4 && dbg_env_i("OQ_WGRAD_PEEL", 1) != 0 && workspace) {
            for (int i = cnt - 1; i >= 0 && peel_i < 0; --i) {
                if (it[i].N % W2_BM != 0 || it[i].K % W2_BN != 0) continue;
                if (rem % tm[i] == 0 && rem / tm[i] < tn[i]) { peel_i = i; peel_cols = (int)(rem / tm[i]); }
                else if (rem % tn[i] == 0 && rem / tn[i] < tm[i]) { peel_i = i; peel_rows = (int)(rem / tn[i]); }
            }
        }
        OqWgradItem strip{};
        int strip_parts = 1;
        if (peel_i >= 0) {
            strip = it[peel_i];
            const int64_t es = 2;
            if (peel_cols) {            // the last peel_cols * 256 columns of gw (= columns of x)
                const int64_t c0 = it[peel_i].K - (int64_t)peel_cols * W2_BN;
                strip.x = (const char*)it[peel_i].x + c0 * es;
                strip.gw = (char*)it[peel_i].gw + c0 * es;
                strip.K = (int64_t)peel_cols * W2_BN;
                it[peel_i].K = c0;
                tn[peel_i] -= peel_cols;
            } else {                    // the last peel_rows * 256 rows of gw (= columns of gy)
                const int64_t r0 = it[peel_i].N - (int64_t)peel_rows * W2_BM;
                strip.gy = (const char*)it[peel_i].gy + r0 * es;
                strip.gw = (char*)it[peel_i].gw + r0 * it[peel_i].ld_gw * es;
                strip.N = (int64_t)peel_rows * W2_BM;
                it[peel_i].N = r0;
                tm[peel_i] -= peel_rows;
            }
            // split contraction of the strip: S parts so that S x (256x128 tiles) fill about one round
            const int64_t tiles3 = ((strip.N + P3_BM - 1) / P3_BM) * ((strip.K + P3_BN - 1) / P3_BN);
            int S = 1;
            for (int s = 8; s >= 2; s /= 2)
                if (strip.T % (s * BK) == 0 && strip.T / s >= 256 && tiles3 * s <= ncu && (int64_t)s * strip.N * strip.K * 4 <= workspace_bytes) { S = s; break; }
            if (S < 2 || !oq_aligned16(workspace) || !oq_aligned16(strip.gw) || !oq_aligned16(strip.x) || !oq_aligned16(strip.gy)) {
                // cannot split: undo the peel
                it[peel_i] = items[base + peel_i];
                tm[peel_i] = (it[peel_i].N + W2_BM - 1) / W2_BM;
                tn[peel_i] = (it[peel_i].K + W2_BN - 1) / W2_BN;
                peel_i = -1;
            } else {
                strip_parts = S;
            }
        }
        GemmGroupP g{};
        g.n = cnt;
        int64_t start = 0;
        for (int i = 0; i < cnt; ++i) {
            GemmP& p = g.p[i];
            p.a = it[i].gy; p.b = it[i].x; p.c = it[i].gw;
            p.M = it[i].N; p.N = it[i].K; p.K = it[i].T; p.lda = it[i].ld_gy; p.ldb = it[i].ld_x; p.ldc = it[i].ld_gw;
            p.alpha = 1.0f; p.batch_i = 1; p.tiles_m = (int)tm[i]; p.tiles_n = (int)tn[i]; p.blk48 = 8; p.epi_lds = 1;
            g.start[i] = (int)start;
            start += tm[i] * tn[i];
        }
        g.start[cnt] = (int)start;
        OQ_CHECK_ARG(start > 0 && start < (1ll << 30), "oq_wgrad_group: %lld tiles", (long long)start);
        hipLaunchKernelGGL((gemm_bf16_w256_group_kernel<bf16_t>), dim3((unsigned)start), dim3(512), 0, st, g);
        OQ_CHECK_LAUNCH("oq_wgrad_group");
        if (peel_i >= 0) {
            const int S = strip_parts;
            g_splitk_request = S;
            const int rc = oq_gemm_impl(strip.gy, strip.x, workspace, nullptr, nullptr, strip.N, strip.K, strip.T, strip.ld_gy, strip.ld_x,
                                        strip.K, 0, 0, OQ_BF16, OQ_F32, 1.0f, 1, 1, 0, 0, 0, 0, 0, 0, 0, stream);
            g_splitk_request = 1;
            if (rc != OQ_OK) return rc;
            const int64_t nvec = strip.N * (strip.K / 8);
            const unsigned grid = (unsigned)(nvec / 256 + 1 < 4096 ? nvec / 256 + 1 : 4096);
            hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const float*)workspace, S, strip.N, strip.K,
                               (bf16_t*)strip.gw, strip.ld_gw, nullptr, (const bf16_t*)nullptr, 1.0f);
            OQ_CHECK_LAUNCH("oq_wgrad_group(reduce)");
        }
    }
    return OQ_OK;
}
