// common.h -- shared device helpers for libvadx (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vadx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace vadx {

void set_error(const char *fmt, ...);

#define VADX_HIP_TRY(expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            vadx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                            __LINE__);                                                       \
            return VADX_EHIP;                                                                \
        }                                                                                    \
    } while (0)

#define VADX_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            vadx::set_error(__VA_ARGS__);  \
            return VADX_EINVAL;            \
        }                                  \
    } while (0)

// Raise a kernel's dynamic-LDS limit once PER DEVICE (the attribute is per device; engines accept device="cuda:N", so a
// process-wide flag would leave the kernel at the 64 KB default on the second GPU) and thread-safely.
#define VADX_DYN_LDS(kernel_expr, bytes)                                                                              \
    do {                                                                                                              \
        static std::atomic<unsigned long long> done_{0};                                                              \
        int dev_ = 0;                                                                                                 \
        VADX_HIP_TRY(hipGetDevice(&dev_));                                                                            \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                          \
        if (!(done_.load(std::memory_order_acquire) & bit_)) {                                                        \
            VADX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel_expr),                             \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));              \
            done_.fetch_or(bit_, std::memory_order_release);                                                          \
        }                                                                                                             \
    } while (0)

// v_mfma_f32_16x16x4_f32: exact f32 FMA chain (A: lane l holds A[l&15][l>>4], B: B[l>>4][l&15],
// D: lane l reg r holds D[4*(l>>4)+r][l&15]).  32-cycle issue, 40-cycle dependent latency.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// gemm_pass: acc[nt][mt] += ACT x W^T over KB blocks of 16 k.
//
//   ACT lives in LDS "k-major": element (k, col) at act[k*lda + col]; for m-tile mt this lane's
//       column is moff[mt] + (lane & 15).  lda % 8 == 4 makes the ds_read_b32 pattern below
//       bank-conflict free (rows 4 apart land 16 banks apart).
//   W   lives in global/L2 FRAGMENT-MAJOR: a [N][K] matrix (N, K multiples of 16) is stored as
//       [N/16 tiles][K/16 blocks][64 lanes][4], the f32x4 of (tile, block S, lane 16q+i) holding
//       W[16*tile + i][16S + 4q + j], j = 0..3 -- exactly what that lane feeds to the 4 MFMAs of block S,
//       so a wave-wide weight load is ONE contiguous 1 KB run (8 full cache lines; a row-major
//       matrix costs 16 half-used lines per load, which made the L1/L2 path the limiter).
//       wrow[nt] = tile base + 4*lane (+ FRAG per 16 k of column offset); tile nt of a matrix with
//       leading dimension ldw starts at W + nt*16*ldw (same footprint as row-major).  The k consumed
//       by (block S, sub-step j, lane quarter q) is 16*S + 4*q + j -- a fixed permutation of the
//       contraction index applied to both operands.
//   SWAP=false: D rows = ACT columns (m), D cols = W rows (n)   -> next layer's k-major LDS
//   SWAP=true : D rows = W rows (n),      D cols = ACT columns  -> e.g. gate-major LSTM input
// Weights for block S+1 are requested before block S's MFMAs issue (L2 latency hiding).
// ---------------------------------------------------------------------------------------------
constexpr int FRAG = 256;     // floats per (16-row tile, 16-k block) fragment

// this lane's pointer into tile `tile` of a fragment-major matrix with leading dimension ldw, at k offset k0
__device__ __forceinline__ const float *frag_ptr(const float *W, int ldw, int tile, int k0, int lane) {
    return W + (size_t)tile * 16 * ldw + (size_t)k0 * 16 + lane * 4;
}

// host: element (row r, column k) of a fragment-major [rows][ldw] matrix
inline size_t frag_index(int ldw, int r, int k) {
    return (size_t)(r / 16) * 16 * ldw + (size_t)(k / 16) * FRAG + (size_t)(((k % 16) / 4) * 16 + (r % 16)) * 4 + (k % 4);
}

// Weight / table loads name the global address space explicitly.  Inside a non-inlined device function a plain
// `const float *` is a generic pointer and the load compiles to flat_load, which counts on BOTH vmcnt and lgkmcnt: the
// LDS waits of the GEMM loops then also wait for the weight prefetch they were meant to overlap.
typedef const __attribute__((address_space(1))) f32x4 *global_f32x4_ptr;
typedef const __attribute__((address_space(1))) float *global_f32_ptr;
__device__ __forceinline__ f32x4 ldg4(const float *p) { return *(global_f32x4_ptr)(p); }
__device__ __forceinline__ float ldg1(const float *p) { return *(global_f32_ptr)(p); }
__device__ __forceinline__ void stg1(float *p, float v) { *(__attribute__((address_space(1))) float *)(p) = v; }

// host: convert a row-major [rows][ldw] block (rows, ldw multiples of 16) to fragment-major in place
inline void frag_major_inplace(float *w, int rows, int ldw) {
    float *tmp = new float[(size_t)rows * ldw];
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < ldw; ++k) tmp[frag_index(ldw, r, k)] = w[(size_t)r * ldw + k];
    for (size_t e = 0; e < (size_t)rows * ldw; ++e) w[e] = tmp[e];
    delete[] tmp;
}

// VADX_GEMM_EXP: development-only what-if switches of gemm_rt (results are wrong when set): 1 activations not read from LDS,
// 2 every weight fragment from the same address (L1 instead of L2), 4 gemm_rt issues one VALU FMA in place of each MFMA
#ifndef VADX_GEMM_EXP
#define VADX_GEMM_EXP 0
#endif

template <int NT, int MT, int KB, bool SWAP>
__device__ __forceinline__ void gemm_pass(f32x4 (&acc)[NT][MT], const float *act, int lda,
                                          const int (&moff)[MT], const float *const (&wrow)[NT],
                                          int lane) {
    const int q = lane >> 4, i = lane & 15;
    const float *ap = act + (4 * q) * lda + i;
    f32x4 wcur[NT], wnxt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wcur[nt] = ldg4(wrow[nt]);
#pragma unroll 2
    for (int S = 0; S < KB; ++S) {
        const int Sn = (S + 1 < KB) ? S + 1 : S;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wnxt[nt] = ldg4(wrow[nt] + FRAG * Sn);
        __builtin_amdgcn_sched_barrier(0);      // keep the next block's weight loads ahead of this block's MFMAs
        const float *aps = ap + 16 * S * lda;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = aps[j * lda + moff[mt]];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float wj = wcur[nt][j];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = SWAP ? mfma16(wj, av[mt], acc[nt][mt]) : mfma16(av[mt], wj, acc[nt][mt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wcur[nt] = wnxt[nt];
    }
}

// Runtime-length variant (kb blocks of 16 k).  AFFINE=true applies (a + add[k]) * mul[k] to the LDS
// operand before the multiply (FSMN's CMVN on the LFR features, FSMN/Export_FSMN_VAD.py:86), with
// add/mul indexed like the weight row (K contiguous, this lane's 4 k per block).
template <int NT, int MT, bool SWAP, bool AFFINE = false>
__device__ __forceinline__ void gemm_rt(f32x4 (&acc)[NT][MT], const float *act, int lda, const int (&moff)[MT],
                                        const float *const (&wrow)[NT], int kb, int lane,
                                        const float *add = nullptr, const float *mul = nullptr) {
    // Software pipeline, one block (16 k) deep on BOTH operands: the LDS activations and the L2 weights of
    // block S+1 are requested before block S's MFMAs issue (with few m-tiles a block is only 4*MT*NT MFMAs, so
    // un-pipelined ds_read latency -- not bandwidth -- is what idles the matrix pipe).  Two register sets
    // alternate (loop unrolled by two) so no copies are needed.
    const int q = lane >> 4, i = lane & 15;
    const float *ap = act + (4 * q) * lda + i;
    f32x4 w0[NT], w1[NT];
    float a0[4][MT], a1[4][MT];
    f32x4 ad0 = {0.f, 0.f, 0.f, 0.f}, ml0 = {1.f, 1.f, 1.f, 1.f}, ad1 = ad0, ml1 = ml0;      // AFFINE terms ride the same pipeline
    auto fetch = [&](int S, f32x4 (&w)[NT], float (&a)[4][MT], f32x4 &ad, f32x4 &ml) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w[nt] = ldg4(wrow[nt] + FRAG * ((VADX_GEMM_EXP & 2) ? 0 : S));
        if (AFFINE) {
            ad = ldg4(add + 16 * S + 4 * q);
            ml = ldg4(mul + 16 * S + 4 * q);
        }
        const float *aps = ap + 16 * S * lda;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[j][mt] = (VADX_GEMM_EXP & 1) ? (float)(S + j + mt) : aps[j * lda + moff[mt]];
    };
    auto compute = [&](const f32x4 (&w)[NT], const float (&a)[4][MT], const f32x4 &a4, const f32x4 &m4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                av[mt] = a[j][mt];
                if (AFFINE) av[mt] = __fmul_rn(__fadd_rn(av[mt], a4[j]), m4[j]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float wj = w[nt][j];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (VADX_GEMM_EXP & 4) { acc[nt][mt][0] += wj * av[mt]; continue; }      // what-if: no MFMA (one VALU FMA keeps the operands live)
                    acc[nt][mt] = SWAP ? mfma16(wj, av[mt], acc[nt][mt]) : mfma16(av[mt], wj, acc[nt][mt]);
                }
            }
        }
    };
    fetch(0, w0, a0, ad0, ml0);
    for (int S = 0; S < kb; S += 2) {
        fetch(S + 1 < kb ? S + 1 : S, w1, a1, ad1, ml1);
        __builtin_amdgcn_sched_barrier(0);      // next block's operands stay ahead of this block's MFMAs
        compute(w0, a0, ad0, ml0);
        if (S + 1 < kb) {
            fetch(S + 2 < kb ? S + 2 : S + 1, w0, a0, ad0, ml0);
            __builtin_amdgcn_sched_barrier(0);
            compute(w1, a1, ad1, ml1);
        }
    }
}

// gemm_rt with the first TWO weight blocks already in registers (r0, r1; kb >= 4, even).  A layer that runs again and
// again on new activations (FireRed's point-wise pair walks the window tile by tile with the same two matrices) keeps
// those fragments resident, so a phase starts multiplying as soon as its LDS operands arrive and streams from block 2
// on -- instead of every wave opening every phase with an L2 round trip nobody else on the CU can cover (one workgroup
// per CU, all waves at the same barrier).
template <int NT, int MT>
__device__ __forceinline__ void gemm_rt_resident(f32x4 (&acc)[NT][MT], const float *act, int lda, const int (&moff)[MT],
                                                 const float *const (&wrow)[NT], int kb, int lane,
                                                 const f32x4 (&r0)[NT], const f32x4 (&r1)[NT]) {
    const int q = lane >> 4, i = lane & 15;
    const float *ap = act + (4 * q) * lda + i;
    f32x4 w0[NT], w1[NT];
    float a0[4][MT], a1[4][MT];
    auto fetch_w = [&](int S, f32x4 (&w)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w[nt] = ldg4(wrow[nt] + FRAG * S);
    };
    auto fetch_a = [&](int S, float (&a)[4][MT]) {
        const float *aps = ap + 16 * S * lda;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[j][mt] = aps[j * lda + moff[mt]];
    };
    auto compute = [&](const f32x4 (&w)[NT], const float (&a)[4][MT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(a[j][mt], w[nt][j], acc[nt][mt]);
    };
    fetch_w(2, w0);
    fetch_w(3, w1);
    fetch_a(0, a0);
    fetch_a(1, a1);
    __builtin_amdgcn_sched_barrier(0);
    compute(r0, a0);
    fetch_a(2, a0);
    __builtin_amdgcn_sched_barrier(0);
    compute(r1, a1);
    for (int S = 2; S < kb; S += 2) {          // w0 = block S, w1 = block S+1, a0 = operands of block S
        fetch_a(S + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        compute(w0, a0);
        if (S + 2 < kb) { fetch_w(S + 2, w0); fetch_a(S + 2, a0); }
        __builtin_amdgcn_sched_barrier(0);
        compute(w1, a1);
        if (S + 3 < kb) fetch_w(S + 3, w1);
    }
}

// All KB weight blocks of the wave's n-tiles resident in registers (w[S][nt]): no global traffic at all, the LDS operands
// of block S+1 are requested before block S's MFMAs issue.
template <int NT, int MT, int KB>
__device__ __forceinline__ void gemm_resident(f32x4 (&acc)[NT][MT], const float *act, int lda, const int (&moff)[MT], int lane,
                                              const f32x4 (&w)[KB][NT]) {
    const int q = lane >> 4, i = lane & 15;
    const float *ap = act + (4 * q) * lda + i;
    float a0[4][MT], a1[4][MT];
    auto fetch_a = [&](int S, float (&a)[4][MT]) {
        const float *aps = ap + 16 * S * lda;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[j][mt] = aps[j * lda + moff[mt]];
    };
    auto compute = [&](const f32x4 (&wb)[NT], const float (&a)[4][MT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = mfma16(a[j][mt], wb[nt][j], acc[nt][mt]);
    };
    fetch_a(0, a0);
#pragma unroll
    for (int S = 0; S < KB; S += 2) {
        if (S + 1 < KB) fetch_a(S + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        compute(w[S], a0);
        if (S + 2 < KB) fetch_a(S + 2, a0);
        __builtin_amdgcn_sched_barrier(0);
        if (S + 1 < KB) compute(w[S + 1 < KB ? S + 1 : S], a1);
    }
}

// Plain one-block look-ahead variant (weights only): measured faster than the operand-pipelined gemm_rt
// for the front-end's NT=2 x MT=2 DFT tiles, where a block already carries 16 MFMAs per 8 ds_reads.
template <int NT, int MT, bool SWAP>
__device__ __forceinline__ void gemm_rt_simple(f32x4 (&acc)[NT][MT], const float *act, int lda, const int (&moff)[MT],
                                               const float *const (&wrow)[NT], int kb, int lane) {
    const int q = lane >> 4, i = lane & 15;
    const float *ap = act + (4 * q) * lda + i;
    f32x4 wcur[NT], wnxt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wcur[nt] = ldg4(wrow[nt]);
    for (int S = 0; S < kb; ++S) {
        const int Sn = (S + 1 < kb) ? S + 1 : S;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wnxt[nt] = ldg4(wrow[nt] + FRAG * ((VADX_GEMM_EXP & 2) ? 0 : Sn));
        const float *aps = ap + 16 * S * lda;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = (VADX_GEMM_EXP & 1) ? (float)(S + j + mt) : aps[j * lda + moff[mt]];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float wj = wcur[nt][j];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = SWAP ? mfma16(wj, av[mt], acc[nt][mt]) : mfma16(av[mt], wj, acc[nt][mt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wcur[nt] = wnxt[nt];
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Gate non-linearities of the recurrent cells on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1 ulp each):
// a recurrent step is a serial chain MFMA -> gates -> next step, so libm's branchy expf/tanhf (~140 VALU instructions
// per unit) sat directly on the critical path.  |error| < 3e-7 per gate, far inside the 1e-4 score bar.
__device__ __forceinline__ float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float gate_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

}  // namespace vadx
