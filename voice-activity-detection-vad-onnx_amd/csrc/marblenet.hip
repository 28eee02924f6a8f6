// marblenet.hip -- MarbleNet (NeMo Jasper, depthwise-separable 1-D convs) building blocks for gfx950.
// Reference wrapper: NVIDIA_Frame_VAD_Multilingual_MarbleNet/Export_NVIDIA_MarbleNet_VAD.py:222-275
// (encoder/decoder classes are NeMo's, un-vendored; BatchNorm is folded on the host like :58-151).
//
// sepconv_block_kernel = ONE fused launch per Jasper sub-block, tile = 1 clip x 32 output frames:
//   stage the input tile + receptive-field halo in LDS  [C_in][32*stride + (k-1)*dil]
//   -> depthwise FIR on the VALU  -> k-major [C_in][32]
//   -> pointwise 1x1 (+ folded BN bias) as an f32-MFMA GEMM, weights streamed from L2
//   -> optional residual branch (1x1 conv + folded BN of the block input) as a second GEMM
//   -> add, ReLU, coalesced store [B][C_out][T_out].
#include "common.h"
#include "layers.h"
#include "split_scheme.h"
#include "layers_split.h"

// MB_EXP: development-only cycle accounting of jasper_block2_kernel (tools/exp_marblenet.py): per-section clock64 sums of thread 0
#ifndef MB_EXP
#define MB_EXP 0
#endif
#if MB_EXP
__device__ unsigned long long mb_dbg[16];
#define MB_T0() long long mb_t_ = clock64()
#define MB_ACC(slot) do { if (threadIdx.x == 0) { const long long n_ = clock64(); atomicAdd(&mb_dbg[slot], (unsigned long long)(n_ - mb_t_)); mb_t_ = n_; } } while (0)
extern "C" int vadx_marblenet_debug_cycles(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mb_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(mb_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define MB_T0() do {} while (0)
#define MB_ACC(slot) do {} while (0)
#endif

#include <math.h>
#include <type_traits>

namespace vadx {
namespace marblenet {

constexpr int THREADS = 512;
constexpr int TILE = 32, A_LD = 36;     // frames per tile, k-major row stride
constexpr int IN_LD_MAX = 100;          // largest halo tile row stride (>= 31*stride + (k-1)*dil + 1, % 8 == 4)
constexpr int MAXC = 128;
// LDS is carved per launch from the block's real shape (Cfg::in_ld, channel counts): a 64-channel k=13 block needs
// 47 KB instead of the 125 KB worst case, i.e. three workgroups per CU instead of one -- a tile is a short serial chain
// (stage -> FIR -> GEMM -> store), so with one workgroup per CU the launch was pure latency (72 rounds x ~6 us).
// A block without a residual branch whose depthwise stage has consumed the input tile writes its output over that tile
// (out_alias): the prologue (80 -> 128 channels, k = 11, stride 2) drops from 56.8 to 38.4 KB -- three workgroups per CU, not two.
static bool out_aliases_in(int cinp, int coutp, int cresp, int in_ld, bool has_dw) {
    return has_dw && !cresp && (size_t)coutp * A_LD <= (size_t)cinp * in_ld;
}
static size_t lds_floats(int cinp, int coutp, int cresp, int in_ld, bool has_dw) {
    return (size_t)cinp * in_ld + (has_dw ? (size_t)cinp * A_LD : 0) + (out_aliases_in(cinp, coutp, cresp, in_ld, has_dw) ? 0 : (size_t)coutp * A_LD) +
           (cresp ? (size_t)(cresp + coutp) * A_LD : 0);
}

struct Cfg {
    int cin, cout, k, stride, dil, pad, has_dw, cres, relu, cinp, coutp, cresp, in_ld, out_alias;
};

// A FIR thread's register window / its eight outputs as 16-byte LDS accesses (rows and window starts are multiples of four floats).
// As single floats the window reads of a wave fell on 8 of the 32 banks (every address a multiple of four floats apart: row strides
// = 4 mod 8, starts = 0 mod 8) -- four-way conflicts on every read, half of all LDS cycles of these kernels (profiles/r03_marblenet).
template <int N>
__device__ __forceinline__ void window_load(float (&win)[N], const float *row) {
#pragma unroll
    for (int v = 0; v < (N + 3) / 4; ++v) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(row + 4 * v);      // (the last one may read <= 3 floats of row padding)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * v + e < N) win[4 * v + e] = t[e];
    }
}
__device__ __forceinline__ void store8(float *dst, const float (&o)[8]) {
    *reinterpret_cast<f32x4 *>(dst) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
}

// KT / DT / ST: compile-time kernel size, dilation and stride of the depthwise stage (0 = take them from `c` at run
// time: the generic fallback).  With constants the FIR is a register-window filter -- a thread owns 8 consecutive
// outputs of one channel, reads its (8-1)*stride + (k-1)*dil + 1 inputs and its k taps ONCE, and everything after
// is FMAs on registers -- and every staging load is unconditional (index clamped, value selected): a guarded load
// compiles to a branch plus a full wait, which used to serialise ~20 memory round trips per thread.
// ---- fp16 x 2 helpers of the single-block kernels (prologue, tail): 128 output channels = eight row tiles, one per wave
// split pass: k-major float32 rows src[ch][col] (ch < rows; beyond: zeros) -> B planes [kgroups][TILE][8] x 2 (item = (8 channels, frame):
// eight conflict-free reads, 24 VALU, one 16-byte store per plane)
__device__ __forceinline__ void split_rows_to_planes(const float *src, int ld, int rows, int kgroups, unsigned char *planes, float &amax) {
    const int pl = kgroups * TILE * 16;
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    for (int it = threadIdx.x; it < kgroups * TILE; it += THREADS) {
        const int kg = it / TILE, col = it - kg * TILE;
        f32x4 lo4, hi4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo4[e] = 8 * kg + e < rows ? src[(8 * kg + e) * ld + col] : 0.f;
            hi4[e] = 8 * kg + 4 + e < rows ? src[(8 * kg + 4 + e) * ld + col] : 0.f;
        }
        u32x2 a0, a1, b0, b1;
        vadx::split2x4(lo4, a0, a1, amax);
        vadx::split2x4(hi4, b0, b1, amax);
        *reinterpret_cast<u32x4_ *>(planes + (kg * TILE + col) * 16) = u32x4_{a0[0], a0[1], b0[0], b0[1]};
        *reinterpret_cast<u32x4_ *>(planes + pl + (kg * TILE + col) * 16) = u32x4_{a1[0], a1[1], b1[0], b1[1]};
    }
}
// eight row tiles = one per wave, both column tiles of the 32-frame tile (layers_split.h's qlayer<> carries a paired-tile path that does
// not fit the 80 registers of six waves per SIMD).  OUT_PLANES as in layers_split.h.
template <bool OUT_PLANES>
__device__ __forceinline__ void gemm8_h2(const vadx::QLayerArgs &a, float &amax) {
    typedef vadx::SchemeH2 SC;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    asm volatile("" : "+v"(lane), "+v"(wave));
    const int i = lane & 15;
    f32x4 hi[1][2], lo[1][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) { hi[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[0][mt] = hi[0][mt]; }
    const float *const w[1] = {a.W + (size_t)wave * a.nchunks * SC::NP * vadx::QFRAG};
    vadx::qgemm_group<SC, 1, 2, OUT_PLANES>(hi, lo, w, 0, a.nchunks, a.act, a.act_pl, [=](int kgrp, int mt) { return (kgrp * TILE + mt * 16 + i) * 16; }, lane);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) vadx::qlayer_store<SC, 2, OUT_PLANES>(a, wave, mt, SC::join(hi[0][mt], lo[0][mt]), lane, amax);
}
__device__ __forceinline__ void range_flag_words(unsigned *flag, float amax) {
    if (!(amax <= vadx::H_MAX)) { atomicOr(flag, 1u); atomicMax(flag + 1, __float_as_uint(amax)); }
}

// AR = 2: the prologue form only (depthwise, no residual branch, 128 filters): filter -> split pass -> gemm8_h2 (fp16 x 2 split products)
template <int KT, int DT, int ST, int AR = 0>
__global__ __launch_bounds__(THREADS, 2) void sepconv_block_kernel(
    Cfg c, const float *__restrict__ dw_w, const float *__restrict__ pw_w, const float *__restrict__ pw_b,
    const float *__restrict__ res_w, const float *__restrict__ res_b, const float *__restrict__ x,
    long long xs_b, long long xs_c, long long xs_t, int T_in, const float *__restrict__ xres,
    float *__restrict__ y, int T_out, int tiles, unsigned *__restrict__ range_flag) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int IN_LD = c.in_ld;
    float *IN = lds, *D = IN + c.cinp * IN_LD, *OUT = c.out_alias ? IN : D + (c.has_dw ? c.cinp * A_LD : 0), *RIN = OUT + c.coutp * A_LD,
          *ROUT = RIN + c.cresp * A_LD;           // (out_alias implies no residual branch: RIN / ROUT are unused then)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x - b * tiles) * TILE;
    const int K = KT ? KT : c.k, DIL = KT ? DT : c.dil, STR = KT ? ST : c.stride;
    const int width = (TILE - 1) * STR + (K - 1) * DIL + 1;
    const float *xb = x + (long long)b * xs_b;
    const int tin0 = t0 * STR - c.pad;
    if (xs_c == 1) {            // time-major source (front-end output): channel fastest -> lane = channel, wave = time
        for (int j0 = 0; j0 < width; j0 += 4 * (THREADS / 64)) {
            float v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = j0 + u * (THREADS / 64) + wave, ch = lane + 64 * h, ti = tin0 + j;
                    const int tc = ti < 0 ? 0 : (ti >= T_in ? T_in - 1 : ti), cc = ch < c.cin ? ch : c.cin - 1;
                    v[u][h] = xb[(long long)tc * xs_t + cc];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = j0 + u * (THREADS / 64) + wave, ch = lane + 64 * h, ti = tin0 + j;
                    if (j < width && ch < c.cinp) IN[ch * IN_LD + j] = (ch < c.cin && ti >= 0 && ti < T_in) ? v[u][h] : 0.f;
                }
        }
    } else {                    // channel-first source: time fastest -> lane = time, wave = channel
        for (int ch0 = 0; ch0 < c.cinp; ch0 += 4 * (THREADS / 64)) {
            float v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ch = ch0 + u * (THREADS / 64) + wave, j = lane + 64 * h, ti = tin0 + j;
                    const int tc = ti < 0 ? 0 : (ti >= T_in ? T_in - 1 : ti), cc = ch < c.cin ? ch : c.cin - 1;
                    v[u][h] = xb[(long long)cc * xs_c + (long long)tc * xs_t];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ch = ch0 + u * (THREADS / 64) + wave, j = lane + 64 * h, ti = tin0 + j;
                    if (j < width && ch < c.cinp) IN[ch * IN_LD + j] = (ch < c.cin && ti >= 0 && ti < T_in) ? v[u][h] : 0.f;
                }
        }
    }
    if (c.cres) {               // residual input tile [cres][32]: lane&31 = frame, 16 rows per pass
        const float *rb = xres + (long long)b * c.cres * T_out;
        const int m = tid & 31, tm = (t0 + m < T_out) ? t0 + m : T_out - 1;
        for (int ch0 = 0; ch0 < c.cresp; ch0 += 4 * (THREADS / 32)) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ch = ch0 + u * (THREADS / 32) + (tid >> 5), cc = ch < c.cres ? ch : c.cres - 1;
                v[u] = rb[(long long)cc * T_out + tm];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ch = ch0 + u * (THREADS / 32) + (tid >> 5);
                if (ch < c.cresp) RIN[ch * A_LD + m] = (ch < c.cres && t0 + m < T_out) ? v[u] : 0.f;
            }
        }
    }
    __syncthreads();
    const float *act = IN;
    int lda = IN_LD;
    if (c.has_dw) {
        if (KT) {               // register-window FIR: thread = (channel tid >> 2, 8 outputs starting at 8 * (tid & 3))
            constexpr int KK = KT ? KT : 1, WIN = 7 * (ST ? ST : 1) + (KK - 1) * (DT ? DT : 1) + 1;
            const int ch = tid >> 2, m0 = 8 * (tid & 3);
            if (ch < c.cinp) {
                const int cc = ch < c.cin ? ch : c.cin - 1;
                float wk[KK], win[WIN];
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) wk[kk] = dw_w[cc * KK + kk];
                const float *row = IN + ch * IN_LD + m0 * (ST ? ST : 1);
                window_load(win, row);
                float o8[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    float s2 = 0.f;
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) s2 = fmaf(wk[kk], win[o * (ST ? ST : 1) + kk * (DT ? DT : 1)], s2);
                    o8[o] = ch < c.cin ? s2 : 0.f;
                }
                store8(D + ch * A_LD + m0, o8);
            }
        } else {
            for (int e = tid; e < c.cinp * TILE; e += THREADS) {
                const int ch = e / TILE, m = e - ch * TILE;
                float s2 = 0.f;
                if (ch < c.cin) {
                    const float *row = IN + ch * IN_LD + m * c.stride;
                    const float *wk = dw_w + ch * c.k;
                    for (int kk = 0; kk < c.k; ++kk) s2 = fmaf(wk[kk], row[kk * c.dil], s2);
                }
                D[ch * A_LD + m] = s2;
            }
        }
        __syncthreads();
        act = D;
        lda = A_LD;
    }
    if constexpr (AR == vadx::VADX_AR_H2) {          // (launcher: has_dw, out_alias, no residual, coutp == 128; planes behind D)
        float amax = 0.f;
        const int kgroups = ((c.cinp + 31) / 32) * 4;
        unsigned char *DP = reinterpret_cast<unsigned char *>(D + c.cinp * A_LD);
        split_rows_to_planes(D, A_LD, c.cinp, kgroups, DP, amax);
        __syncthreads();
        gemm8_h2<false>(vadx::QLayerArgs{pw_w, c.coutp / 16, kgroups / 4, pw_b, c.relu ? 1 : 0, DP, kgroups * TILE * 16,
                                         reinterpret_cast<unsigned char *>(OUT), 0, A_LD, nullptr}, amax);
        range_flag_words(range_flag, amax);
    } else {
        LayerArgs a{pw_w, c.cinp, c.coutp / 16, 1, c.cinp / 16, 0, 0, pw_b, (c.relu && !c.cres) ? 1 : 0,
                    act, lda, 0, OUT, A_LD, 0, nullptr, nullptr};
        layer<2, false>(a);
    }
    if (c.cres) {
        LayerArgs r{res_w, c.cresp, c.coutp / 16, 1, c.cresp / 16, 0, 0, res_b, 0, RIN, A_LD, 0, ROUT, A_LD, 0, nullptr, nullptr};
        layer<2, false>(r);
    }
    __syncthreads();
    float *yb = y + (long long)b * c.cout * T_out;
    for (int e = tid; e < c.cout * TILE; e += THREADS) {
        const int ch = e / TILE, m = e - ch * TILE;
        if (t0 + m < T_out) {
            float v = OUT[ch * A_LD + m];
            if (c.cres) { v += ROUT[ch * A_LD + m]; if (c.relu) v = fmaxf(v, 0.f); }
            yb[(long long)ch * T_out + t0 + m] = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Fused Jasper block (the three residual blocks of MarbleNet 3x2x64: two separable sub-blocks + the residual 1x1 branch of
// the block input): ONE launch per block instead of two, and the 64-channel tensor between the sub-blocks never leaves LDS.
// Tile = 1 clip x 32 output frames; sub-block 0 is evaluated on the 32 + (K - 1) frames sub-block 1's depthwise filter needs
// (48 columns = 3 m-tiles: <= 1.5 x recompute of a layer that is a quarter of the block's work), masked to zero outside
// [0, T) -- the zero padding sub-block 1 sees in the reference.
//   IN  [cin][48 + K - 1]   block input, frames t0 - (K-1) ..; the residual 1x1 (operand: columns K - 1 ..) runs FIRST, into
//   ROUT[c2][32]; then depthwise 0 overwrites IN in place (every thread reads its windows, barrier, writes) -> D0 [cin][48];
//   H1  [c1][48]            relu(pw0 D0 + b0);  depthwise 1 -> D1 [c1][32] in IN's place;  pw1 -> OUT [c2][32] in H1's place.
// Three live regions instead of four: 53 KB for the 128-channel block -- three workgroups per CU like the 64-channel blocks
// (it ran two at 80 KB; a tile is a short serial chain, so this kernel lives on workgroups per CU).
// ---------------------------------------------------------------------------------------------
struct Blk2 {
    int cin, cinp, c1, c2, in_ld, T;      // c1 = c2 = 64 here; cinp = cin padded to 16
};
constexpr int W1 = 48, H_LD = 52;

// GEMM of the fused block: 64 output channels = four row tiles for eight waves.  As whole-tile work items (layer<>) that was one
// (row tile, column tile) chain per wave opened by an L2 round trip -- the three GEMMs took 61 % of the kernel for 1.5 us of MFMA
// issue per tile.  Here wave = (row tile nt = wave & 3, K half kh = wave >> 2): every wave multiplies its half of K for ALL MT column
// tiles, its weight fragments (KB/2 <= 4 of them) are requested a phase EARLY (kgemm_pre: before the depthwise filter in front of
// the GEMM) and the two halves meet in the destination rows: the kh = 1 wave stores its partial tile, the kh = 0 wave adds its own
// and the bias behind a barrier -- a fixed order, so the result is reproducible.
struct KPre { f32x4 w[4]; };

__device__ __forceinline__ KPre kgemm_pre(const float *__restrict__ W, int kdim) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    asm volatile("" : "+v"(lane), "+v"(wave));
    const int nt = wave & 3, kh = wave >> 2, nb = kdim >> 5;      // 16-k blocks per half: 2 (K = 64) or 4 (K = 128)
    const float *base = vadx::frag_ptr(W, kdim, nt, 16 * kh * nb, lane);
    KPre p;
#pragma unroll
    for (int u = 0; u < 4; ++u) p.w[u] = vadx::ldg4(base + vadx::FRAG * (u < nb ? u : nb - 1));      // unconditional, clamped
    return p;
}

template <int MT>
__device__ __forceinline__ void kgemm(const KPre &p, int kdim, const float *act, int lda, int acol0, float *dst, int ldd,
                                      const float *__restrict__ bias, bool relu) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    asm volatile("" : "+v"(lane), "+v"(wave));
    const int q = lane >> 4, i = lane & 15, nt = wave & 3, kh = wave >> 2, nb = kdim >> 5;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *ap = act + (16 * kh * nb + 4 * q) * lda + acol0 + i;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (u < nb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#if defined(MB_WHATIF) && (MB_WHATIF & 1)
                if (j || (u & 1)) continue;                  // what-if: one MFMA in eight (the matrix time of an fp16 x 2 product)
#endif
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = vadx::mfma16(ap[(16 * u + j) * lda + mt * 16], p.w[u][j], acc[mt]);
            }
        }
    }
    float *dp = dst + (nt * 16 + i) * ldd + 4 * q;
    if (kh == 1)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4 *>(dp + mt * 16) = acc[mt];
    __syncthreads();
    if (kh == 0) {
        const float b = bias[nt * 16 + i];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(dp + mt * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = (v[r] + acc[mt][r]) + b; if (relu) v[r] = fmaxf(v[r], 0.f); }
            *reinterpret_cast<f32x4 *>(dp + mt * 16) = v;
        }
    }
}

// The same GEMM on fp16 x 2 split products (split2.h): three v_mfma_f32_16x16x32_f16 per 32 channels and column tile instead of eight
// v_mfma_f32_16x16x4_f32 -- on this kernel the f32 MFMAs shared the vector datapath with the depthwise filters (with one MFMA in eight
// the three block launches ran 4.14 -> 2.72 ms).  The depthwise filters leave k-major float32 rows (a thread owns eight consecutive
// frames of ONE channel), so there are no planes to read: a wave splits its own operand -- lane (q, i) reads the eight channels
// 16 (e >> 2) + 4 q + (e & 3), e = 0..7, of frame i (the f32 kernel's conflict-free row pattern: quarters 4 rows = 16 banks apart),
// splits them (3 VALU per value) and uses them as the A operand; the weights are the B operand, their fragments packed on the host in
// the same k-slot order (vadx_frag_h2_host).  The four row-tile waves of a K half split the same values (4 x redundant, 28 VALU per
// fragment): still a third of the datapath time of the f32 MFMAs they replace.
struct KPreH { f16x8 w[2][2]; };         // [chunk of this wave's K half][plane]

__device__ __forceinline__ KPreH kgemm_pre_h(const float *__restrict__ W, int kdim) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    asm volatile("" : "+v"(lane), "+v"(wave));
    const int nt = wave & 3, kh = wave >> 2, nc = kdim >> 6;      // 32-k chunks per half: 1 (K = 64) or 2 (K = 128)
    const float *base = W + (size_t)((nt * 2 * nc + kh * nc) * 2) * vadx::HFRAG;
    KPreH p;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) p.w[u][pl] = vadx::ldh(base + (size_t)((u < nc ? u : nc - 1) * 2 + pl) * vadx::HFRAG, lane);      // unconditional, clamped
    return p;
}

template <int MT>
__device__ __forceinline__ void kgemm_h(const KPreH &p, int kdim, const float *act, int lda, int acol0, float *dst, int ldd,
                                        const float *__restrict__ bias, bool relu, float &amax) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    asm volatile("" : "+v"(lane), "+v"(wave));
    const int q = lane >> 4, i = lane & 15, nt = wave & 3, kh = wave >> 2, nc = kdim >> 6;
    f32x4 hi[MT], mid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { hi[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; mid[mt] = hi[mt]; }
    const float *ap = act + (32 * kh * nc + 4 * q) * lda + acol0 + i;
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u < nc) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 lo4, hi4;
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo4[j] = ap[(32 * u + j) * lda + mt * 16]; hi4[j] = ap[(32 * u + 16 + j) * lda + mt * 16]; }
                u32x2 a0, a1, b0, b1;
                vadx::split2x4(lo4, a0, a1, amax);
                vadx::split2x4(hi4, b0, b1, amax);
                const f16x8 x0 = __builtin_bit_cast(f16x8, u32x4_{a0[0], a0[1], b0[0], b0[1]});
                const f16x8 x1 = __builtin_bit_cast(f16x8, u32x4_{a1[0], a1[1], b1[0], b1[1]});
                mid[mt] = vadx::mfma_f16(x1, p.w[u][0], mid[mt]);
                mid[mt] = vadx::mfma_f16(x0, p.w[u][1], mid[mt]);
                hi[mt] = vadx::mfma_f16(x0, p.w[u][0], hi[mt]);
            }
        }
    }
    float *dp = dst + (nt * 16 + i) * ldd + 4 * q;
    if (kh == 1)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4 *>(dp + mt * 16) = vadx::join2(hi[mt], mid[mt]);
    __syncthreads();
    if (kh == 0) {
        const float b = bias[nt * 16 + i];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(dp + mt * 16);
            const f32x4 own = vadx::join2(hi[mt], mid[mt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = (v[r] + own[r]) + b; if (relu) v[r] = fmaxf(v[r], 0.f); }
            *reinterpret_cast<f32x4 *>(dp + mt * 16) = v;
        }
    }
}

// ---- AR = 3: the plane form of the fp16 x 2 GEMMs, for the 64-channel blocks (their 40 KB leave room for 12 KB of planes at three
// workgroups per CU; the 128-channel block sits at 53.2 KB and keeps kgemm_h).  The operand is split ONCE by a transposing pass (item = (8
// channels, column): eight conflict-free reads, 24 VALU, one 16-byte store per plane) instead of by each of the four row-tile waves, and
// with no split to share a wave takes a whole K: wave = (row tile, column-tile parity), no partial sums, no barrier inside the GEMM -- the
// pass's barrier takes its place.  Weights: vadx_frag_h2_host, VADX_H2_K_PLAIN.
__device__ __forceinline__ void split_rows_to_planes_n(const float *src, int ld, int col0, int kgroups, int ncol, unsigned char *planes, float &amax) {
    const int pl = kgroups * ncol * 16;
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    for (int it = threadIdx.x; it < kgroups * ncol; it += THREADS) {
        const int kg = it / ncol, col = it - kg * ncol;
        f32x4 lo4, hi4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo4[e] = src[(8 * kg + e) * ld + col0 + col]; hi4[e] = src[(8 * kg + 4 + e) * ld + col0 + col]; }
        u32x2 a0, a1, b0, b1;
        vadx::split2x4(lo4, a0, a1, amax);
        vadx::split2x4(hi4, b0, b1, amax);
        *reinterpret_cast<u32x4_ *>(planes + (kg * ncol + col) * 16) = u32x4_{a0[0], a0[1], b0[0], b0[1]};
        *reinterpret_cast<u32x4_ *>(planes + pl + (kg * ncol + col) * 16) = u32x4_{a1[0], a1[1], b1[0], b1[1]};
    }
}
__device__ __forceinline__ KPreH kgemm_pre_p(const float *__restrict__ W) {      // K = 64: this wave's row tile, both chunks, both planes
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    asm volatile("" : "+v"(lane), "+v"(wave));
    const float *base = W + (size_t)((wave & 3) * 2 * 2) * vadx::HFRAG;
    KPreH p;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) p.w[u][pl] = vadx::ldh(base + (size_t)(u * 2 + pl) * vadx::HFRAG, lane);
    return p;
}
template <int MT>
__device__ __forceinline__ void pgemm(const KPreH &p, const unsigned char *planes, int ncol, float *dst, int ldd, const float *__restrict__ bias, bool relu) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    asm volatile("" : "+v"(lane), "+v"(wave));
    const int q = lane >> 4, i = lane & 15, nt = wave & 3, mh = wave >> 2, plb = 8 * ncol * 16;
    const float b = bias[nt * 16 + i];
#pragma unroll
    for (int mt0 = 0; mt0 < MT; mt0 += 2) {
        const int mt = mt0 + mh;
        if (mt < MT) {
            f32x4 hi = {0.f, 0.f, 0.f, 0.f}, mid = hi;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned char *ap = planes + ((4 * u + q) * ncol + mt * 16 + i) * 16;
                const f16x8 a0 = *reinterpret_cast<const f16x8 *>(ap), a1 = *reinterpret_cast<const f16x8 *>(ap + plb);
                mid = vadx::mfma_f16(a1, p.w[u][0], mid);
                hi = vadx::mfma_f16(a0, p.w[u][0], hi);
                mid = vadx::mfma_f16(a0, p.w[u][1], mid);
            }
            f32x4 v = vadx::join2(hi, mid);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] += b; if (relu) v[r] = fmaxf(v[r], 0.f); }
            *reinterpret_cast<f32x4 *>(dst + (nt * 16 + i) * ldd + mt * 16 + 4 * q) = v;
        }
    }
}
// AR: 0 = float32 MFMAs (kgemm), 2 = fp16 x 2 split products (kgemm_h); one body for both
template <int AR> struct KPreOf { typedef KPre type; };
template <> struct KPreOf<vadx::VADX_AR_H2> { typedef KPreH type; };
template <> struct KPreOf<3> { typedef KPreH type; };          // 3 = fp16 x 2, plane form (pgemm)
template <int AR>
__device__ __forceinline__ typename KPreOf<AR>::type kpre(const float *__restrict__ W, int kdim) {
    if constexpr (AR == 3) return kgemm_pre_p(W);
    else if constexpr (AR == vadx::VADX_AR_H2) return kgemm_pre_h(W, kdim);
    else return kgemm_pre(W, kdim);
}
template <int AR, int MT>
__device__ __forceinline__ void kmul(const typename KPreOf<AR>::type &p, int kdim, const float *act, int lda, int acol0, float *dst, int ldd,
                                     const float *__restrict__ bias, bool relu, float &amax) {
    if constexpr (AR == vadx::VADX_AR_H2) kgemm_h<MT>(p, kdim, act, lda, acol0, dst, ldd, bias, relu, amax);
    else kgemm<MT>(p, kdim, act, lda, acol0, dst, ldd, bias, relu);
}

template <int K, int AR>
__global__ __launch_bounds__(THREADS, 6) void jasper_block2_kernel(
    Blk2 c, const float *__restrict__ dw0, const float *__restrict__ pw0, const float *__restrict__ b0,
    const float *__restrict__ dw1, const float *__restrict__ pw1, const float *__restrict__ b1,
    const float *__restrict__ rw, const float *__restrict__ rb, const float *__restrict__ x, float *__restrict__ y, int tiles,
    unsigned *__restrict__ range_flag) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef typename KPreOf<AR>::type KP;
    float amax = 0.f;
    constexpr int PAD = (K - 1) / 2, WIN0 = W1 + K - 1;
    const int IN_LD = c.in_ld;
    float *IN = lds, *H1 = IN + c.cinp * IN_LD, *ROUT = H1 + c.c1 * H_LD;
    float *D0 = IN, *D1 = IN, *OUT = H1;
    unsigned char *PL = reinterpret_cast<unsigned char *>(ROUT + c.c2 * A_LD);      // AR = 3: operand planes [8 k-groups][<= 48 columns][8] x 2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x - b * tiles) * TILE;
    const float *xb = x + (long long)b * c.cin * c.T;
    const int tin0 = t0 - 2 * PAD;
    MB_T0();
    const KP wres = kpre<AR>(rw, c.cinp);                    // in flight while the input tile is staged
    // ---- stage the block input (channel-first source: lane = time, wave = channel), unconditional clamped loads
    for (int ch0 = 0; ch0 < c.cinp; ch0 += 8 * (THREADS / 64)) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ch = ch0 + u * (THREADS / 64) + wave, ti = tin0 + lane;
            const int tc = ti < 0 ? 0 : (ti >= c.T ? c.T - 1 : ti), cc = ch < c.cin ? ch : c.cin - 1;
            v[u] = xb[(long long)cc * c.T + tc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ch = ch0 + u * (THREADS / 64) + wave, ti = tin0 + lane;
            if (lane < WIN0 && ch < c.cinp) IN[ch * IN_LD + lane] = (ch < c.cin && ti >= 0 && ti < c.T) ? v[u] : 0.f;
        }
    }
    MB_ACC(0);
    __syncthreads();
    MB_ACC(7);
    // pointwise 0's weights: in flight through the residual GEMM and depthwise 0 -- except at K = 17, whose 24-value window + 17 taps +
    // 16 prefetched weight registers no longer fit the 80 VGPRs of six waves per SIMD (20 B of scratch per lane doubled the kernel's
    // HBM writes): there they are requested behind the filter
    KP wpw0;
    if (K < 17) wpw0 = kpre<AR>(pw0, c.cinp);
    // residual 1x1 of the block input (frames t0 .. t0 + 31 sit at column 2 PAD), before the input is overwritten
    if constexpr (AR == 3) {
        split_rows_to_planes_n(IN, IN_LD, 2 * PAD, 8, TILE, PL, amax);
        __syncthreads();            // (the residual GEMM reads the planes only: depthwise 0 may overwrite IN behind this barrier)
        pgemm<2>(wres, PL, TILE, ROUT, A_LD, rb, false);
        MB_ACC(1);
    } else {
        kmul<AR, 2>(wres, c.cinp, IN, IN_LD, 2 * PAD, ROUT, A_LD, rb, false, amax);
        MB_ACC(1);
        __syncthreads();            // every residual operand is read: depthwise 0 may overwrite IN
    }
    MB_ACC(7);
    // ---- depthwise 0 IN PLACE: register-window FIR, item = (channel, 8 outputs).  The six items of a channel are six neighbouring
    // lanes of ONE wave: a wave's window reads all precede its writes in program order (the FMAs in between depend on them), and
    // no two waves share a channel, so the overwrite needs no further barrier.
    for (int ch0 = 0; ch0 < c.cinp; ch0 += 10 * (THREADS / 64)) {
        const int ch = ch0 + wave * 10 + lane / 6, m0 = 8 * (lane % 6);
        if (lane < 60 && ch < c.cinp) {
            const int cc = ch < c.cin ? ch : c.cin - 1;
            float wk[K], win[7 + K];
#pragma unroll
            for (int kk = 0; kk < K; ++kk) wk[kk] = dw0[cc * K + kk];
            float *row = IN + ch * IN_LD + m0;
            window_load(win, row);
            float o8[8];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float s2 = 0.f;
#pragma unroll
                for (int kk = 0; kk < K; ++kk) s2 = fmaf(wk[kk], win[o + kk], s2);
                o8[o] = ch < c.cin ? s2 : 0.f;
            }
            __builtin_amdgcn_wave_barrier();                // keep every lane's reads above every lane's writes in the schedule
            store8(row, o8);
        }
    }
    if (K >= 17) wpw0 = kpre<AR>(pw0, c.cinp);
    MB_ACC(2);
    __syncthreads();
    MB_ACC(7);
    KP wpw1;                                                 // pointwise 1's weights: in flight through pointwise 0 and depthwise 1
    if (K < 17) wpw1 = kpre<AR>(pw1, c.c1);
    // pointwise 0 + folded BN + ReLU on 48 columns
    if constexpr (AR == 3) {
        split_rows_to_planes_n(D0, IN_LD, 0, 8, W1, PL, amax);       // (every wave is past the residual GEMM: the barrier behind depthwise 0)
        __syncthreads();
        pgemm<3>(wpw0, PL, W1, H1, H_LD, b0, true);
    } else {
        kmul<AR, 3>(wpw0, c.cinp, D0, IN_LD, 0, H1, H_LD, b0, true, amax);
    }
    MB_ACC(3);
    __syncthreads();
    MB_ACC(7);
    // ---- depthwise 1 on H1 (column j = frame t0 - PAD + j; frames outside the clip are the conv's zero padding)
    for (int it = tid; it < c.c1 * (TILE / 8); it += THREADS) {
        const int ch = it / (TILE / 8), m0 = 8 * (it - ch * (TILE / 8));
        float wk[K], win[7 + K];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) wk[kk] = dw1[ch * K + kk];
        const float *row = H1 + ch * H_LD + m0;
        window_load(win, row);
#pragma unroll
        for (int u = 0; u < 7 + K; ++u) {
            const int fr = t0 - PAD + m0 + u;
            win[u] = (fr >= 0 && fr < c.T) ? win[u] : 0.f;
        }
        float o8[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float s2 = 0.f;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) s2 = fmaf(wk[kk], win[o + kk], s2);
            o8[o] = s2;
        }
        store8(D1 + ch * A_LD + m0, o8);                                // D1 sits in IN's region (D0 is dead since the barrier above)
    }
    if (K >= 17) wpw1 = kpre<AR>(pw1, c.c1);
    MB_ACC(4);
    __syncthreads();                // every H1 read is done: OUT may overwrite it
    MB_ACC(7);
    if constexpr (AR == 3) {
        split_rows_to_planes_n(D1, A_LD, 0, 8, TILE, PL, amax);
        __syncthreads();
        pgemm<2>(wpw1, PL, TILE, OUT, A_LD, b1, false);
    } else {
        kmul<AR, 2>(wpw1, c.c1, D1, A_LD, 0, OUT, A_LD, b1, false, amax);
    }
    MB_ACC(5);
    __syncthreads();
    MB_ACC(7);
    float *yb = y + (long long)b * c.c2 * c.T;
    for (int e = tid; e < c.c2 * TILE; e += THREADS) {
        const int ch = e / TILE, m = e - ch * TILE;
        if (t0 + m < c.T) yb[(long long)ch * c.T + t0 + m] = fmaxf(OUT[ch * A_LD + m] + ROUT[ch * A_LD + m], 0.f);
    }
    MB_ACC(6);
    if (AR >= vadx::VADX_AR_H2 && !(amax <= vadx::H_MAX)) {          // an operand left the fp16 range: the host recomputes this batch on float32
        atomicOr(range_flag, 1u);
        atomicMax(range_flag + 1, __float_as_uint(amax));
    }
}

// ---------------------------------------------------------------------------------------------
// Tail of the encoder + decoder in one launch: block 5 (depthwise k = 29, dilation 2, 64 -> 128, ReLU), block 6 (plain 1x1,
// 128 -> 128, ReLU), Linear(128 -> 2), softmax: the two 128-channel tensors and the encoder output never reach HBM -- only
// the two scores per frame leave the kernel (wrapper :265-274).
//   IN [64][32 + 56] -> D [64][32] -> H [128][32] -> OUT [128][32] (in IN's place) -> scores
// ---------------------------------------------------------------------------------------------
struct Tail { int cin, cmid, k, dil, in_ld, T; };

// AR = 2 (fp16 x 2 split products, layers_split.h: qlayer): the two 1x1 convs are GEMM -> GEMM, so only the filter's k-major float32 output
// needs a transposing split pass (item = (8 channels, frame): eight conflict-free reads, 24 VALU, one 16-byte store per plane into the dead
// input tile's place); block 5's conv takes the weights as its A operand and stores its ReLU output straight into block 6's B planes
// [channel / 8][frame][8] (8-byte stores, in H's place), block 6 takes the activations as A and stores float32 rows for the decoder.
template <int AR>
__global__ __launch_bounds__(THREADS, 6) void marblenet_tail_kernel(
    Tail c, const float *__restrict__ dw, const float *__restrict__ pw, const float *__restrict__ pb,
    const float *__restrict__ w6, const float *__restrict__ b6, const float *__restrict__ dec_w, const float *__restrict__ dec_b,
    const float *__restrict__ x, float *__restrict__ s0, float *__restrict__ s1, int tiles, unsigned *__restrict__ range_flag) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int K = 29, DIL = 2, PAD = 28, WIN0 = TILE + 2 * PAD;
    const int IN_LD = c.in_ld;
    float amax = 0.f;
    float *IN = lds, *D = IN + c.cin * IN_LD, *H = D + c.cin * A_LD, *OUT = IN;
    float *red = D;                                 // [16 parts][32 frames][2] once D is dead
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x - b * tiles) * TILE;
    const float *xb = x + (long long)b * c.cin * c.T;
    const int tin0 = t0 - PAD;
    for (int ch0 = 0; ch0 < c.cin; ch0 += 4 * (THREADS / 64)) {
        float v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ch = ch0 + u * (THREADS / 64) + wave, ti = tin0 + lane + 64 * h;
                const int tc = ti < 0 ? 0 : (ti >= c.T ? c.T - 1 : ti), cc = ch < c.cin ? ch : c.cin - 1;
                v[u][h] = xb[(long long)cc * c.T + tc];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ch = ch0 + u * (THREADS / 64) + wave, j = lane + 64 * h, ti = tin0 + j;
                if (j < WIN0 && ch < c.cin) IN[ch * IN_LD + j] = (ti >= 0 && ti < c.T) ? v[u][h] : 0.f;
            }
    }
    __syncthreads();
    for (int it = tid; it < c.cin * (TILE / 8); it += THREADS) {       // register-window FIR, dilation 2
        const int ch = it / (TILE / 8), m0 = 8 * (it - ch * (TILE / 8));
        float wk[K], win[7 + (K - 1) * DIL + 1];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) wk[kk] = dw[ch * K + kk];
        const float *row = IN + ch * IN_LD + m0;
        window_load(win, row);
        float o8[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float s2 = 0.f;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) s2 = fmaf(wk[kk], win[o + kk * DIL], s2);
            o8[o] = s2;
        }
        store8(D + ch * A_LD + m0, o8);
    }
    __syncthreads();
    if constexpr (AR == vadx::VADX_AR_H2) {
        unsigned char *DP = reinterpret_cast<unsigned char *>(IN), *HP = reinterpret_cast<unsigned char *>(H);      // the input tile is dead
        const int dp_pl = (c.cin / 8) * TILE * 16, hp_pl = (c.cmid / 8) * TILE * 16;                                  // bytes per plane
        split_rows_to_planes(D, A_LD, c.cin, c.cin / 8, DP, amax);
        __syncthreads();
        gemm8_h2<true>(vadx::QLayerArgs{pw, c.cmid / 16, c.cin / 32, pb, 1, DP, dp_pl, HP, hp_pl, TILE, nullptr}, amax);
        __syncthreads();            // D and its planes are dead
        gemm8_h2<false>(vadx::QLayerArgs{w6, c.cmid / 16, c.cmid / 32, b6, 1, HP, hp_pl, reinterpret_cast<unsigned char *>(OUT), 0, A_LD, nullptr}, amax);
    } else {
        {
            LayerArgs a{pw, c.cin, c.cmid / 16, 1, c.cin / 16, 0, 0, pb, 1, D, A_LD, 0, H, A_LD, 0, nullptr, nullptr};
            layer<2, false>(a);
        }
        __syncthreads();            // IN and D are dead
        {
            LayerArgs a{w6, c.cmid, c.cmid / 16, 1, c.cmid / 16, 0, 0, b6, 1, H, A_LD, 0, OUT, A_LD, 0, nullptr, nullptr};
            layer<2, false>(a);
        }
    }
    __syncthreads();
    {   // decoder: thread = (part of 8 channels, frame); partial logits meet in LDS and are summed in part order
        const int m = tid & 31, part = tid >> 5;
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ch = part * 8 + u;
            const float v = OUT[ch * A_LD + m];
            z0 = fmaf(dec_w[ch], v, z0);
            z1 = fmaf(dec_w[c.cmid + ch], v, z1);
        }
        red[(part * TILE + m) * 2] = z0;
        red[(part * TILE + m) * 2 + 1] = z1;
    }
    __syncthreads();
    if (tid < TILE && t0 + tid < c.T) {
        float z0 = dec_b[0], z1 = dec_b[1];
#pragma unroll
        for (int part = 0; part < THREADS / 32; ++part) { z0 += red[(part * TILE + tid) * 2]; z1 += red[(part * TILE + tid) * 2 + 1]; }
        const float mx = fmaxf(z0, z1), e0 = expf(z0 - mx), e1 = expf(z1 - mx), inv = 1.0f / (e0 + e1);
        s0[(long long)b * c.T + t0 + tid] = e0 * inv;
        s1[(long long)b * c.T + t0 + tid] = e1 * inv;
    }
    if (AR == vadx::VADX_AR_H2) range_flag_words(range_flag, amax);
}

// decoder Linear(C -> 2) + softmax (wrapper :270-274): one thread per (clip, frame)
__global__ void frame_classifier_kernel(const float *__restrict__ enc, const float *__restrict__ w, const float *__restrict__ bias,
                                        int B, int C, int T, float *__restrict__ s0, float *__restrict__ s1) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * T) return;
    const int b = (int)(idx / T), t = (int)(idx - (long long)b * T);
    const float *e = enc + (long long)b * C * T + t;
    float z0 = 0.f, z1 = 0.f;
    for (int c2 = 0; c2 < C; ++c2) {
        const float v = e[(long long)c2 * T];
        z0 = fmaf(w[c2], v, z0);
        z1 = fmaf(w[C + c2], v, z1);
    }
    z0 += bias[0]; z1 += bias[1];
    const float m = fmaxf(z0, z1), e0 = expf(z0 - m), e1 = expf(z1 - m), inv = 1.0f / (e0 + e1);
    s0[idx] = e0 * inv;
    s1[idx] = e1 * inv;
}

}  // namespace marblenet
}  // namespace vadx

using namespace vadx::marblenet;

extern "C" int vadx_sepconv_block(const vadx_sepconv_cfg *cfg, const float *dw_w, const float *pw_w, const float *pw_b,
                                  const float *res_w, const float *res_b, const float *x, int64_t xs_b, int64_t xs_c,
                                  int64_t xs_t, int t_in, const float *xres, float *y, int batch, int t_out, void *stream,
                                  const vadx_marblenet_cfg *mcfg) {
    VADX_REQUIRE(cfg && pw_w && pw_b && x && y, "vadx_sepconv_block: NULL argument");
    const int ar = vadx::arith_internal(mcfg ? mcfg->arithmetic : VADX_ARITH_AUTO, vadx::VADX_AR_F32);
    VADX_REQUIRE(ar == vadx::VADX_AR_F32 || ar == vadx::VADX_AR_H2, "vadx_sepconv_block: arithmetic must be AUTO / F32 or F16X2");
    unsigned *flag = mcfg ? static_cast<unsigned *>(mcfg->range_flag) : nullptr;
    Cfg c;
    c.cin = cfg->cin; c.cout = cfg->cout; c.k = cfg->kernel; c.stride = cfg->stride; c.dil = cfg->dilation;
    c.has_dw = cfg->depthwise ? 1 : 0; c.cres = cfg->residual_cin; c.relu = cfg->relu ? 1 : 0;
    c.pad = (c.dil * (c.k - 1)) / 2;
    c.cinp = (c.cin + 15) & ~15; c.coutp = (c.cout + 15) & ~15; c.cresp = (c.cres + 15) & ~15;
    VADX_REQUIRE(c.cin > 0 && c.cin <= MAXC && c.cout > 0 && c.cout <= MAXC && c.cres >= 0 && c.cres <= MAXC,
                 "vadx_sepconv_block: channels must be in [1,128]");
    VADX_REQUIRE(c.k >= 1 && c.stride >= 1 && c.dil >= 1 && (TILE - 1) * c.stride + (c.k - 1) * c.dil + 1 <= IN_LD_MAX,
                 "vadx_sepconv_block: receptive field of a 32-frame tile exceeds %d samples", IN_LD_MAX);
    {   const int width = (TILE - 1) * c.stride + (c.k - 1) * c.dil + 1;
        c.in_ld = c.has_dw ? (((width + 7) & ~7) + 4) : A_LD; }      // % 8 == 4; a plain 1x1 block feeds IN straight to the GEMM
    c.out_alias = out_aliases_in(c.cinp, c.coutp, c.cresp, c.in_ld, c.has_dw) ? 1 : 0;
    size_t lds_bytes = lds_floats(c.cinp, c.coutp, c.cresp, c.in_ld, c.has_dw) * sizeof(float);
    if (ar == vadx::VADX_AR_H2) {
        VADX_REQUIRE(flag && c.has_dw && !c.cres && c.out_alias && c.coutp == 128 && c.k == 11 && c.dil == 1 && c.stride == 2,
                     "vadx_sepconv_block: F16X2 is built for the MarbleNet prologue (depthwise k 11 stride 2, 128 filters, no residual; pw_w from "
                     "vadx_frag_h2_host VADX_H2_K_PLAIN) and needs mcfg->range_flag");
        lds_bytes += (size_t)((c.cinp + 31) / 32) * 4 * TILE * 16 * 2;      // the operand planes behind D
    }
    VADX_REQUIRE(c.has_dw ? dw_w != nullptr : (c.k == 1 && c.stride == 1), "vadx_sepconv_block: plain conv must be k=1, stride 1");
    VADX_REQUIRE(!c.cres || (res_w && res_b && xres), "vadx_sepconv_block: residual branch needs res_w/res_b/xres");
    VADX_REQUIRE(batch > 0 && t_in > 0 && t_out > 0 && t_out == (t_in + 2 * c.pad - c.dil * (c.k - 1) - 1) / c.stride + 1,
                 "vadx_sepconv_block: t_out=%d inconsistent with t_in=%d", t_out, t_in);
    const int tiles = (t_out + TILE - 1) / TILE;
    VADX_REQUIRE((long long)batch * tiles < (1LL << 31), "vadx_sepconv_block: too many tiles");
#define SEPCONV_LAUNCH(...)                                                                                                     \
    do {                                                                                                                        \
        VADX_DYN_LDS((sepconv_block_kernel<__VA_ARGS__>), 128 * 1024);                                                          \
        hipLaunchKernelGGL((sepconv_block_kernel<__VA_ARGS__>), dim3((unsigned)(batch * tiles)), dim3(THREADS),                 \
                           lds_bytes, static_cast<hipStream_t>(stream), c, dw_w, pw_w, pw_b, res_w, res_b, x,                   \
                           (long long)xs_b, (long long)xs_c, (long long)xs_t, t_in, xres, y, t_out, tiles, flag);                \
    } while (0)
    // the depthwise shapes of the published MarbleNet 3x2x64 get compile-time FIRs; anything else runs the generic kernel
    if (ar == vadx::VADX_AR_H2) SEPCONV_LAUNCH(11, 1, 2, 2);
    else if (c.has_dw && c.k == 11 && c.dil == 1 && c.stride == 2) SEPCONV_LAUNCH(11, 1, 2);
    else if (c.has_dw && c.k == 13 && c.dil == 1 && c.stride == 1) SEPCONV_LAUNCH(13, 1, 1);
    else if (c.has_dw && c.k == 15 && c.dil == 1 && c.stride == 1) SEPCONV_LAUNCH(15, 1, 1);
    else if (c.has_dw && c.k == 17 && c.dil == 1 && c.stride == 1) SEPCONV_LAUNCH(17, 1, 1);
    else if (c.has_dw && c.k == 29 && c.dil == 2 && c.stride == 1) SEPCONV_LAUNCH(29, 2, 1);
    else SEPCONV_LAUNCH(0, 0, 0);
#undef SEPCONV_LAUNCH
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}


extern "C" int vadx_marblenet_block2(int cin, int kernel, const float *dw0, const float *pw0, const float *b0, const float *dw1,
                                     const float *pw1, const float *b1, const float *res_w, const float *res_b, const float *x,
                                     float *y, int batch, int frames, void *stream, const vadx_marblenet_cfg *cfg) {
    VADX_REQUIRE(dw0 && pw0 && b0 && dw1 && pw1 && b1 && res_w && res_b && x && y, "vadx_marblenet_block2: NULL argument");
    const int ar = vadx::arith_internal(cfg ? cfg->arithmetic : VADX_ARITH_AUTO, vadx::VADX_AR_F32);
    VADX_REQUIRE(ar == vadx::VADX_AR_F32 || ar == vadx::VADX_AR_H2, "vadx_marblenet_block2: arithmetic must be AUTO / F32 (fragment-major weights) or "
                 "F16X2 (vadx_frag_h2_host weights)");
    VADX_REQUIRE(ar != vadx::VADX_AR_H2 || cfg->range_flag, "vadx_marblenet_block2: F16X2 needs cfg->range_flag (two device words)");
    unsigned *flag = cfg ? static_cast<unsigned *>(cfg->range_flag) : nullptr;
    VADX_REQUIRE((cin == 64 || cin == 128) && (kernel == 13 || kernel == 15 || kernel == 17) && batch > 0 && frames > 0,
                 "vadx_marblenet_block2: built for the published MarbleNet 3x2x64 residual blocks (cin 64/128, 64 filters, kernel 13/15/17)");
    Blk2 c;
    c.cin = cin; c.cinp = (cin + 15) & ~15; c.c1 = 64; c.c2 = 64; c.T = frames;
    const int width = W1 + kernel - 1;
    c.in_ld = width + ((4 - width % 8) + 8) % 8;             // smallest row stride >= width with stride % 8 == 4
    const int tiles = (frames + TILE - 1) / TILE;
    VADX_REQUIRE((long long)batch * tiles < (1LL << 31), "vadx_marblenet_block2: too many tiles");
    const bool planes = ar == vadx::VADX_AR_H2 && cin == 64;      // the plane form (AR = 3): 12 KB more, still three workgroups per CU
    const size_t lds = ((size_t)c.cinp * c.in_ld + (size_t)c.c1 * H_LD + (size_t)c.c2 * A_LD) * sizeof(float) + (planes ? 2 * 8 * W1 * 16 : 0);
    VADX_REQUIRE(c.cinp * c.in_ld >= c.c1 * A_LD, "vadx_marblenet_block2: the depthwise-1 output does not fit the input region");
#define BLK2_LAUNCH(KK, AR)                                                                                                    \
    do {                                                                                                                       \
        VADX_DYN_LDS((jasper_block2_kernel<KK, AR>), 128 * 1024);                                                              \
        hipLaunchKernelGGL((jasper_block2_kernel<KK, AR>), dim3((unsigned)(batch * tiles)), dim3(THREADS), lds,                \
                           static_cast<hipStream_t>(stream), c, dw0, pw0, b0, dw1, pw1, b1, res_w, res_b, x, y, tiles, flag);   \
    } while (0)
    if (planes) {
        if (kernel == 13) BLK2_LAUNCH(13, 3);
        else if (kernel == 15) BLK2_LAUNCH(15, 3);
        else BLK2_LAUNCH(17, 3);
    } else if (ar == vadx::VADX_AR_H2) {
        if (kernel == 13) BLK2_LAUNCH(13, 2);
        else if (kernel == 15) BLK2_LAUNCH(15, 2);
        else BLK2_LAUNCH(17, 2);
    } else {
        if (kernel == 13) BLK2_LAUNCH(13, 0);
        else if (kernel == 15) BLK2_LAUNCH(15, 0);
        else BLK2_LAUNCH(17, 0);
    }
#undef BLK2_LAUNCH
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

extern "C" int vadx_marblenet_tail(const float *dw, const float *pw, const float *pb, const float *w6, const float *b6,
                                   const float *dec_w, const float *dec_b, const float *x, float *score0, float *score1,
                                   int batch, int frames, void *stream, const vadx_marblenet_cfg *cfg) {
    VADX_REQUIRE(dw && pw && pb && w6 && b6 && dec_w && dec_b && x && score0 && score1, "vadx_marblenet_tail: NULL argument");
    const int ar = vadx::arith_internal(cfg ? cfg->arithmetic : VADX_ARITH_AUTO, vadx::VADX_AR_F32);
    VADX_REQUIRE(ar == vadx::VADX_AR_F32 || ar == vadx::VADX_AR_H2, "vadx_marblenet_tail: arithmetic must be AUTO / F32 (fragment-major weights) or "
                 "F16X2 (vadx_frag_h2_host weights, VADX_H2_K_PLAIN)");
    VADX_REQUIRE(ar != vadx::VADX_AR_H2 || cfg->range_flag, "vadx_marblenet_tail: F16X2 needs cfg->range_flag (two device words)");
    unsigned *flag = cfg ? static_cast<unsigned *>(cfg->range_flag) : nullptr;
    VADX_REQUIRE(batch > 0 && frames > 0, "vadx_marblenet_tail: bad shape");
    Tail c;
    c.cin = 64; c.cmid = 128; c.k = 29; c.dil = 2; c.T = frames;
    c.in_ld = ((TILE + 56 + 7) & ~7) + 4;
    const int tiles = (frames + TILE - 1) / TILE;
    VADX_REQUIRE((long long)batch * tiles < (1LL << 31), "vadx_marblenet_tail: too many tiles");
    const size_t lds = ((size_t)c.cin * c.in_ld + (size_t)c.cin * A_LD + (size_t)c.cmid * A_LD) * sizeof(float);
    if (ar == vadx::VADX_AR_H2) {
        VADX_DYN_LDS(marblenet_tail_kernel<2>, 128 * 1024);
        hipLaunchKernelGGL(marblenet_tail_kernel<2>, dim3((unsigned)(batch * tiles)), dim3(THREADS), lds, static_cast<hipStream_t>(stream),
                           c, dw, pw, pb, w6, b6, dec_w, dec_b, x, score0, score1, tiles, flag);
    } else {
        VADX_DYN_LDS(marblenet_tail_kernel<0>, 128 * 1024);
        hipLaunchKernelGGL(marblenet_tail_kernel<0>, dim3((unsigned)(batch * tiles)), dim3(THREADS), lds, static_cast<hipStream_t>(stream),
                           c, dw, pw, pb, w6, b6, dec_w, dec_b, x, score0, score1, tiles, flag);
    }
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

extern "C" int vadx_frame_classifier(const float *enc, const float *dec_w, const float *dec_b, int batch, int channels,
                                     int frames, float *score0, float *score1, void *stream) {
    VADX_REQUIRE(enc && dec_w && dec_b && score0 && score1, "vadx_frame_classifier: NULL argument");
    VADX_REQUIRE(batch > 0 && channels > 0 && frames > 0, "vadx_frame_classifier: bad shape");
    const long long n = (long long)batch * frames;
    hipLaunchKernelGGL(frame_classifier_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), enc, dec_w, dec_b, batch, channels, frames, score0, score1);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
