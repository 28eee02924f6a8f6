// layers.h -- dense layer on LDS-resident activations shared by the FSMN / FireRed / MarbleNet nets.
#pragma once
#include "common.h"

namespace vadx {

// layer<> spreads its work items over all waves of the workgroup (blockDim.x / 64)

struct LayerArgs {
    const float *W; int ldw, ntiles;      // W fragment-major (common.h), ldw = its padded K
    int npass, kb, kstep, cstep;          // K passes: weights advance kstep floats, act columns cstep
    const float *bias; int relu;
    const float *act; int lda, acol0;
    float *dst; int ldd, dcol0;
    const float *add, *mul;               // CMVN (AFFINE only)
    float *scratch = nullptr;             // optional LDS scratch (>= NW*256*MTT floats, not act/dst): leftover n-tiles are
                                          // then K-split over all waves instead of running as single-tile chains
};

// dst[n][m] = act[k][m] x W[n][k] (+bias, ReLU).  Work items: whole n-tiles (all MTT m-tiles of the tile share each
// weight fragment) for as many rounds as fill every wave; the remaining n-tiles are split into (n-tile, m-tile) pairs.
// (A layer with 9 n-tiles on 8 waves used to run entirely as 36 single-tile items: one dependent MFMA chain per
// weight load, i.e. latency-bound -- FSMN's in_linear1 took 20 % of the kernel for 13 % of its MFMAs.)
template <int MTT, bool AFFINE>
__device__ __forceinline__ void layer(const LayerArgs &a) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // opaque to the optimiser: a layer called inside a loop over tiles / windows is loop-invariant in everything but its LDS contents, and
    // LICM hoists every layer's per-lane weight pointers out of that loop (FSMN: 238 VGPRs spilled, 16 MB of scratch per XCD thrashing L2)
    asm volatile("" : "+v"(lane), "+v"(wave));
    const int q = lane >> 4, i = lane & 15;
    const int NW = blockDim.x >> 6;
    const int full = (a.ntiles / NW) * NW;
    // rounds are taken two at a time (n-tiles nt and nt + NW side by side: 2*MTT accumulators, one pipeline fill and
    // drain instead of two -- all waves fill at the same moment after a barrier, so that time is not hidden by anyone)
    int nt0 = wave;
    for (; nt0 + NW < full; nt0 += 2 * NW) {
        f32x4 acc[2][MTT];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) acc[h][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *row0 = frag_ptr(a.W, a.ldw, nt0, 0, lane), *row1 = frag_ptr(a.W, a.ldw, nt0 + NW, 0, lane);
        for (int ps = 0; ps < a.npass; ++ps) {
            const float *const wrow[2] = {row0 + ps * a.kstep * 16, row1 + ps * a.kstep * 16};
            int moff[MTT];
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) moff[mt] = a.acol0 + mt * 16 + ps * a.cstep;
            gemm_rt<2, MTT, false, AFFINE>(acc, a.act, a.lda, moff, wrow, a.kb, lane,
                                           AFFINE ? a.add + ps * a.kstep : nullptr, AFFINE ? a.mul + ps * a.kstep : nullptr);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nt = nt0 + h * NW;
            const float b = a.bias ? ldg1(a.bias + nt * 16 + i) : 0.f;
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[h][mt][r] + b; if (a.relu) v[r] = fmaxf(v[r], 0.f); }
                *reinterpret_cast<f32x4 *>(a.dst + (nt * 16 + i) * a.ldd + a.dcol0 + mt * 16 + 4 * q) = v;
            }
        }
    }
    for (int nt = nt0; nt < full; nt += NW) {
        f32x4 acc[1][MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) acc[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *row = frag_ptr(a.W, a.ldw, nt, 0, lane);
        for (int ps = 0; ps < a.npass; ++ps) {
            const float *const wrow[1] = {row + ps * a.kstep * 16};
            int moff[MTT];
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) moff[mt] = a.acol0 + mt * 16 + ps * a.cstep;
            gemm_rt<1, MTT, false, AFFINE>(acc, a.act, a.lda, moff, wrow, a.kb, lane,
                                           AFFINE ? a.add + ps * a.kstep : nullptr, AFFINE ? a.mul + ps * a.kstep : nullptr);
        }
        const float b = a.bias ? ldg1(a.bias + nt * 16 + i) : 0.f;
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[0][mt][r] + b; if (a.relu) v[r] = fmaxf(v[r], 0.f); }
            *reinterpret_cast<f32x4 *>(a.dst + (nt * 16 + i) * a.ldd + a.dcol0 + mt * 16 + 4 * q) = v;
        }
    }
    if (a.scratch && a.ntiles - full == 1) {
        // One leftover n-tile (e.g. 9 tiles on 8 waves): every wave contracts a slice of its K blocks for all MTT m-tiles
        // (full weight reuse, no dependent single-tile chain), the partial tiles meet in LDS and are summed in wave
        // order -- a fixed order, so the result is reproducible.
        const int nt = full, nblk = a.npass * a.kb;
        f32x4 acc[MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *row = frag_ptr(a.W, a.ldw, nt, 0, lane);
        const float *ap = a.act + (4 * q) * a.lda + i + a.acol0;
        for (int b = wave; b < nblk; b += NW) {
            const int ps = b / a.kb, S = b - ps * a.kb;
            const f32x4 w4 = ldg4(row + ps * a.kstep * 16 + S * FRAG);
            f32x4 a4 = {0.f, 0.f, 0.f, 0.f}, m4 = {1.f, 1.f, 1.f, 1.f};
            if (AFFINE) {
                a4 = ldg4(a.add + ps * a.kstep + 16 * S + 4 * q);
                m4 = ldg4(a.mul + ps * a.kstep + 16 * S + 4 * q);
            }
            const float *aps = ap + 16 * S * a.lda + ps * a.cstep;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MTT; ++mt) {
                    float av = aps[j * a.lda + mt * 16];
                    if (AFFINE) av = __fmul_rn(__fadd_rn(av, a4[j]), m4[j]);
                    acc[mt] = mfma16(av, w4[j], acc[mt]);
                }
        }
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt)
            *reinterpret_cast<f32x4 *>(a.scratch + ((wave * 16 + i) * MTT + mt) * 16 + 4 * q) = acc[mt];
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * MTT * 4; e += blockDim.x) {      // (row i2, m-tile, frame quad)
            const int i2 = e / (MTT * 4), rem = e - i2 * MTT * 4;
            f32x4 v = *reinterpret_cast<const f32x4 *>(a.scratch + (i2 * MTT) * 16 + rem * 4);
            for (int w2 = 1; w2 < NW; ++w2) v += *reinterpret_cast<const f32x4 *>(a.scratch + ((w2 * 16 + i2) * MTT) * 16 + rem * 4);
            const float b = a.bias ? ldg1(a.bias + nt * 16 + i2) : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] += b; if (a.relu) v[r] = fmaxf(v[r], 0.f); }
            *reinterpret_cast<f32x4 *>(a.dst + (nt * 16 + i2) * a.ldd + a.dcol0 + rem * 4) = v;
        }
        return;
    }
    for (int item = wave; item < (a.ntiles - full) * MTT; item += NW) {
        const int nt = full + item / MTT, mt = item - (item / MTT) * MTT;
        f32x4 acc[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
        const float *row = frag_ptr(a.W, a.ldw, nt, 0, lane);
        for (int ps = 0; ps < a.npass; ++ps) {
            const float *const wrow[1] = {row + ps * a.kstep * 16};
            const int moff[1] = {a.acol0 + mt * 16 + ps * a.cstep};
            gemm_rt<1, 1, false, AFFINE>(acc, a.act, a.lda, moff, wrow, a.kb, lane,
                                         AFFINE ? a.add + ps * a.kstep : nullptr, AFFINE ? a.mul + ps * a.kstep : nullptr);
        }
        const float b = a.bias ? ldg1(a.bias + nt * 16 + i) : 0.f;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[0][0][r] + b; if (a.relu) v[r] = fmaxf(v[r], 0.f); }
        *reinterpret_cast<f32x4 *>(a.dst + (nt * 16 + i) * a.ldd + a.dcol0 + mt * 16 + 4 * q) = v;
    }
}


}  // namespace vadx
