// layers_split.h -- dense layer on LDS-resident activations as split products (csrc/split_scheme.h: SC = SchemeB3, bf16 x 3, or SchemeH2,
// fp16 x 2): the split-product counterpart of layers.h for the FSMN / FireRed nets.
//
// Activations are SC::NP 16-bit planes [k / 8][NCOL columns][8] (NCOL a multiple of 16: a wave's ds_read_b128 of one (k-group, 16 columns)
// block per quarter is conflict-free); weights are A fragments [n-tile][32-k chunk][plane][QFRAG] streamed from L2 one chunk ahead.
// A layer chooses its output orientation by which operand the weights are:
//   OUT_PLANES  weights = A operand: D rows = output channels, a lane holds four consecutive channels of one column -> one 8-byte store
//               into each plane of the next layer's input (bias, ReLU, split3 in the epilogue);
//   OUT_F32     activations = A operand: D rows = columns (frames), a lane holds four consecutive frames of one channel -> one 16-byte
//               float32 store into a k-major [channel][frame] buffer (the FIR's input with its history columns, the softmax's logits).
// Same fragments either way (lane 16 g + i supplies eight consecutive k of row / column i).
#pragma once
#include "common.h"
#include "split_scheme.h"

namespace vadx {

struct QLayerArgs {
    const float *W;                 // A fragments of tile nt: W + nt * nchunks * NP * QFRAG; chunk kc, plane p at (kc * NP + p) * QFRAG
    int ntiles, nchunks;
    const float *bias;              // [ntiles * 16] or nullptr
    int relu;
    const unsigned char *act;       // input planes (LDS), plane p at act + p * act_pl
    int act_pl;
    unsigned char *dst;             // OUT_PLANES: planes base, plane stride dst_pl, NCOL = dst_ncol; OUT_F32: float buffer
    int dst_pl, dst_ncol;           // OUT_F32: dst_ncol = row stride in floats, dst_pl = first column
    float *exch;                    // unused (kept so that the argument lists of the two layer families line up)
};

// One group of NT n-tiles x MTT column tiles.  baddr(kgrp, mt) -> byte offset (inside a plane) of this lane's 16-byte B block of
// k-group kgrp in column tile mt, or of a block of zeros when kgrp lies beyond the layer's K.
// RING: chunk kc + RING - 1 is requested before chunk kc's MFMAs issue (2 = one chunk ahead; a deeper ring costs NT * NP * 4 registers per
// step and pays where few waves share a SIMD and a step's MFMAs are shorter than the L2 round trip).
template <typename SC, int NT, int MTT, bool OUT_PLANES, int RING = 2, typename BAddr>
__device__ __forceinline__ void qgemm_group(f32x4 (&hi)[NT][MTT], f32x4 (&lo)[NT][MTT], const float *const (&w)[NT], int kc0, int kc1,
                                            const unsigned char *act, int act_pl, BAddr baddr, int lane) {
    const int q = lane >> 4;
    constexpr int NP = SC::NP;
    typedef typename SC::frag frag;
    frag ar[RING][NT][NP];
    auto load_a = [&](int kc, frag (&a)[NT][NP]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#if defined(QG_WHATIF) && (QG_WHATIF & 1)
                if (kc > kc0) { a[nt][p] = ar[0][nt][p]; continue; }     // what-if: the weight stream costs nothing
#endif
                a[nt][p] = SC::ld(w[nt] + (size_t)(kc * NP + p) * QFRAG, lane);
            }
    };
    auto step = [&](int kc, const frag (&a)[NT][NP]) {
        frag b[MTT][NP];
        int boff[MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) boff[mt] = baddr(4 * kc + q, mt);
        // plane-major request order: the first products (a2 x b0, a1 x b1 ...) of every column tile can issue after MTT reads have landed
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) {
#if defined(QG_WHATIF) && (QG_WHATIF & 2)
                b[mt][p] = a[0][p];                 // what-if: the activation reads cost nothing
                continue;
#endif
                b[mt][p] = SC::lds(act + boff[mt] + p * act_pl);
            }
        SC::template products<NT, MTT, OUT_PLANES>(a, b, hi, lo);
    };
    if constexpr (RING == 2) {                      // (spelled out: this form allocates fewer registers than the general loop below)
        load_a(kc0, ar[0]);
        for (int kc = kc0; kc < kc1; kc += 2) {
            if (kc + 1 < kc1) load_a(kc + 1, ar[1]);
            __builtin_amdgcn_sched_barrier(0);      // the next chunk's fragments are requested before this chunk's MFMAs issue
            step(kc, ar[0]);
            if (kc + 1 < kc1) {
                if (kc + 2 < kc1) load_a(kc + 2, ar[0]);
                __builtin_amdgcn_sched_barrier(0);
                step(kc + 1, ar[1]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < RING - 1; ++j)
            if (kc0 + j < kc1) load_a(kc0 + j, ar[j]);
        for (int kc = kc0; kc < kc1; kc += RING) {
#pragma unroll
            for (int j = 0; j < RING; ++j) {
                if (kc + j < kc1) {
                    if (kc + j + RING - 1 < kc1) load_a(kc + j + RING - 1, ar[(j + RING - 1) % RING]);
                    __builtin_amdgcn_sched_barrier(0);
                    step(kc + j, ar[j]);
                }
            }
        }
    }
}

template <typename SC, int MTT, bool OUT_PLANES>
__device__ __forceinline__ void qlayer_store(const QLayerArgs &a, int nt, int mt, f32x4 v, int lane, float &amax) {
    const int q = lane >> 4, i = lane & 15;
    if (OUT_PLANES) {
        if (a.bias) v += ldg4(a.bias + nt * 16 + 4 * q);
        if (a.relu)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        u32x2 pp[SC::NP];
        SC::split4(v, pp, amax);
        const int g = 4 * nt + q;
        unsigned char *d = a.dst + ((g >> 1) * a.dst_ncol + mt * 16 + i) * 16 + (g & 1) * 8;
#pragma unroll
        for (int p = 0; p < SC::NP; ++p) *reinterpret_cast<u32x2 *>(d + p * a.dst_pl) = pp[p];
    } else {
        const float b = a.bias ? ldg1(a.bias + nt * 16 + i) : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] += b; if (a.relu) v[r] = fmaxf(v[r], 0.f); }
        *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(a.dst) + (nt * 16 + i) * a.dst_ncol + a.dst_pl + mt * 16 + 4 * q) = v;
    }
}

// dst = W x act (+bias, ReLU).  n-tiles go to the waves round-robin, two rounds side by side where there are that many; leftover
// n-tiles (9 tiles on 8 waves) run as single (n-tile, column tile) items.
template <typename SC, int MTT, bool OUT_PLANES, typename BAddr>
__device__ __forceinline__ void qlayer(const QLayerArgs &a, BAddr baddr, float &amax) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    asm volatile("" : "+v"(lane), "+v"(wave));      // nothing per-lane is hoisted out of the enclosing tile / window loops (layers.h)
    const int NW = blockDim.x >> 6;
    const int full = (a.ntiles / NW) * NW;
    const size_t tstride = (size_t)a.nchunks * SC::NP * QFRAG;
    int nt0 = wave;
    for (; nt0 + NW < full; nt0 += 2 * NW) {
        f32x4 hi[2][MTT], lo[2][MTT];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) { hi[h][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[h][mt] = hi[h][mt]; }
        const float *const w[2] = {a.W + nt0 * tstride, a.W + (nt0 + NW) * tstride};
        qgemm_group<SC, 2, MTT, OUT_PLANES>(hi, lo, w, 0, a.nchunks, a.act, a.act_pl, baddr, lane);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) qlayer_store<SC, MTT, OUT_PLANES>(a, nt0 + h * NW, mt, SC::join(hi[h][mt], lo[h][mt]), lane, amax);
    }
    for (; nt0 < full; nt0 += NW) {
        f32x4 hi[1][MTT], lo[1][MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) { hi[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[0][mt] = hi[0][mt]; }
        const float *const w[1] = {a.W + nt0 * tstride};
        qgemm_group<SC, 1, MTT, OUT_PLANES>(hi, lo, w, 0, a.nchunks, a.act, a.act_pl, baddr, lane);
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) qlayer_store<SC, MTT, OUT_PLANES>(a, nt0, mt, SC::join(hi[0][mt], lo[0][mt]), lane, amax);
    }
    // leftover n-tiles (9 tiles on 8 waves): (n-tile, column tile) items, one 16 x 16 output tile each
    for (int item = wave; item < (a.ntiles - full) * MTT; item += NW) {
        const int nt = full + item / MTT, mt_w = item - (item / MTT) * MTT;
        f32x4 hi[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}}, lo[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
        const float *const w[1] = {a.W + nt * tstride};
        auto b1 = [&](int kgrp, int) { return baddr(kgrp, mt_w); };
        qgemm_group<SC, 1, 1, OUT_PLANES>(hi, lo, w, 0, a.nchunks, a.act, a.act_pl, b1, lane);
        qlayer_store<SC, MTT, OUT_PLANES>(a, nt, mt_w, SC::join(hi[0][0], lo[0][0]), lane, amax);
    }
}

}  // namespace vadx
