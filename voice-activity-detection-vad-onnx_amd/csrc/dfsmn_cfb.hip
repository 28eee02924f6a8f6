// dfsmn_cfb.hip -- one ICCRN gated conv block (CFB + CepsUnit, Export_DFSMN_VAD.py:76-154) as TWO streaming kernels around the
// frequency-axis LSTM, instead of the six launches (+ four statistics merges) of the unfused chain in dfsmn.hip:
//
//   cfb_front   x -> LN0 -> conv_gate / conv_input -> gx = g * xi, r = xi - gx            (:87-89)
//                 -> y1 = conv31(w1 * gx)   (the (3,1) conv of :90 on the UN-normalised gx, see below)      -> HBM
//                 -> S  = DFT_F(LN2(r))     (CepsUnit :134-138), LayerNorm statistics of S                  -> HBM
//   lstm_f      (dfsmn.hip, unchanged)  S -> LN -> bi-LSTM along F -> hf                                    -> HBM
//   cfb_back    hf, S -> Linear 40->40 -> complex product with S -> pinv inverse DFT (:141-153)
//                 -> out = inv1 * (y1 - mean1 * CW) + CB + ceps                                             -> HBM
//
// Per block the 20-channel tensors that cross HBM drop from 15-16 to 9-10 (gx, r, lo, ceps never exist in memory).
//
// Both kernels STREAM over the frequency axis in chunks of four bins = one MFMA k-step of the length-160 DFT: the DFT of all
// 20 channels of a 16-frame tile (20 x 10 row tiles of 16 x 16) is accumulated in registers -- 25 tiles = 100 accumulator
// VGPRs per wave, eight waves -- while the chunk's 1x1 / (3,1) convs run, so no 20 x 160 x 16 tensor (205 KB) ever has to sit in
// LDS; the 160 x 160 DFT table does instead (102 KB, fragment order, one conflict-free ds_read_b32 per MFMA).  Workgroups are
// persistent (one per CU) and walk the tiles.
//
// LayerNorm over (C, F) needs the frame's statistics before its first use, which a single streaming pass does not have.  LN is
// affine per frame, so it commutes with the linear ops behind it:
//     DFT(LN2(r))[c][m]   = inv2 * ( sum_f T[m][f] w2[c][f] r[c][f]  -  mean2 * TW[c][m] ) + TB[c][m]
//     conv31(LN1(gx))[co][f] = inv1 * ( conv31(w1 * gx)[co][f] - mean1 * CW[co][f] ) + CB[co][f]
// with TW = T w2, TB = T b2, CW = conv31(w1), CB = conv31(b1) + bias evaluated once on the host in float64.  The raw sums are
// accumulated while the statistics are gathered (shifted sums per lane, Chan merges: same scheme as dfsmn.hip), and the
// correction is ONE extra MFMA k-step per tile with A = (TW, TB) and B = (-mean, std + eps).  Re-association only: the block's
// output equals the unfused chain's to float32 rounding (tests/test_gpu_dfsmn.py).
#include "common.h"
#include "split3.h"

#include <stdlib.h>
#include <string.h>
#include <type_traits>

// CFB_EXP: development-only what-if switches (bit mask; results are wrong when set; tools/exp_cfb.py): 1 no workgroup barriers in the
// chunk loops, 2 no global loads of the chunk rows, 4 no gate / input conv MFMAs, 8 no (3,1) conv, 16 no DFT accumulation,
// 32 no gate arithmetic / statistics, 64 no y1 stores, 128 no Linear + complex product (back), 256 no epilogue stores,
// 512 no y1 loads in cfb_back's prologue, 1024 no prologue transposes
#ifndef CFB_EXP
#define CFB_EXP 0
#endif
#define CFB_SYNC() do { if (!(CFB_EXP & 1)) __syncthreads(); } while (0)
// bit 2048: cycle accounting (s_memtime deltas of lane 0 of the first producer and the first DFT wave, summed over all workgroups
// into cfb_dbg[slot]; vadx_cfb_debug_cycles reads / clears them): slot = 8 * kernel (0 front, 1 back) + 4 * role + section
#if CFB_EXP & 2048
__device__ unsigned long long cfb_dbg[32];
#define CFB_T0() long long cfb_t_ = __builtin_readcyclecounter(); unsigned long long cfb_a_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CFB_MARK(slot) do { const long long n_ = __builtin_readcyclecounter(); cfb_a_[(slot) & 7] += (unsigned long long)(n_ - cfb_t_); cfb_t_ = n_; } while (0)
#define CFB_FLUSH(base) do { if ((threadIdx.x & 255) == 0) for (int k_ = 0; k_ < 8; ++k_) if (cfb_a_[k_]) atomicAdd(&cfb_dbg[(base) + k_], cfb_a_[k_]); } while (0)
extern "C" int vadx_cfb_debug_cycles(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(cfb_dbg), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(cfb_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define CFB_T0() do {} while (0)
#define CFB_MARK(slot) do {} while (0)
#define CFB_FLUSH(base) do {} while (0)
#endif

namespace vadx {
namespace dfsmn_cfb {

struct View {
    const float *ptr;
    int c_total, c_off, c;
};
struct ViewW {
    float *ptr;
    int c_total, c_off, c;
};

constexpr int F = 160, CH = 20, CF = 81, NTH = 512;                 // bins, channels, ceps bins, threads
constexpr int KSF = 40, KSI = 41;                                   // k-steps of the forward / inverse table
constexpr int TBLF_FLOATS = 10 * KSF * 64, TBLI_FLOATS = 10 * KSI * 64;

// Global access as UNIFORM base + unsigned 32-bit BYTE offset: the form the hardware addresses as (scalar base, 32-bit lane offset).
// An element offset would have to be scaled by four in 64 bits (it may overflow 32), i.e. one 64-bit address pair per access.
__device__ __forceinline__ float ldg1o(const float *base, unsigned byte_off) { return *(global_f32_ptr)(reinterpret_cast<const char *>(base) + byte_off); }
__device__ __forceinline__ f32x4 ldg4o(const float *base, unsigned byte_off) { return *(global_f32x4_ptr)(reinterpret_cast<const char *>(base) + byte_off); }
__device__ __forceinline__ void stg1o(float *base, unsigned byte_off, float v) { *(__attribute__((address_space(1))) float *)(reinterpret_cast<char *>(base) + byte_off) = v; }
__device__ __forceinline__ void stg4o(float *base, unsigned byte_off, f32x4 v) { *(__attribute__((address_space(1))) f32x4 *)(reinterpret_cast<char *>(base) + byte_off) = v; }

__device__ __forceinline__ size_t ft_idx(int tile, int c_total, int c, int Fb, int f) {
    return (((size_t)tile * c_total + c) * Fb + f) * 16;
}

// running (count, shift, sum, sum of squares) of one lane's values of ONE frame; -> (n, mean, M2)
struct Acc1 {
    float K, s1, s2, n;
    __device__ __forceinline__ void init() { K = s1 = s2 = n = 0.f; }
    __device__ __forceinline__ void add(float v) {
        if (n == 0.f) K = v;
        const float d = v - K;
        s1 += d; s2 = fmaf(d, d, s2); n += 1.f;
    }
    __device__ __forceinline__ void addk(float v) { const float d = v - K; s1 += d; s2 = fmaf(d, d, s2); n += 1.f; }      // K set by the caller
    __device__ __forceinline__ void finish(float &cnt, float &mean, float &M2) const {
        cnt = n;
        const float inv = n > 0.f ? 1.0f / n : 0.f;
        mean = K + s1 * inv;
        M2 = fmaxf(s2 - s1 * s1 * inv, 0.f);
    }
};

__device__ __forceinline__ void chan_merge(float &na, float &ma, float &Ma, float nb, float mb, float Mb) {
    const float n = na + nb, f = n > 0.f ? nb / n : 0.f, d = mb - ma;
    ma += d * f;
    Ma += Mb + d * d * (na * f);
    na = n;
}

// merge over the four lane quarters (same column i), leaving the result in every lane
__device__ __forceinline__ void chan_merge_q(float &n, float &m, float &M) {
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        const float nb = __shfl_xor(n, off), mb = __shfl_xor(m, off), Mb = __shfl_xor(M, off);
        chan_merge(n, m, M, nb, mb, Mb);
    }
}

__device__ __forceinline__ float sum_q(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// LDS-only workgroup barrier: every wave's LDS writes are complete (lgkmcnt) before it arrives; outstanding GLOBAL loads / stores
// are NOT waited for (a __syncthreads() is free to, and the chunk loops keep a prefetch and the y1 / S stores in flight across it).
__device__ __forceinline__ void lds_barrier() {
    if (CFB_EXP & 1) return;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int WSP = 20, WS_TILE = 16 * WSP, WS_FLOATS = 2 * WS_TILE;      // per-wave 16 x 16 transpose scratch (two tiles), pitch 20: D-layout writes of the four lane
                                                   // quarters land 16 banks apart, rows stay 16-byte aligned for ds_read_b128

// D layout (lane (q, i) holds rows 4 q + r of column i) <-> row layout (lane l holds row l >> 2, columns 4 (l & 3) .. + 3) of one
// 16 x 16 tile through the wave's own LDS scratch: a tile then leaves / arrives as ONE 16-byte access per lane -- a wave covers the
// tile's 1 KB contiguously -- instead of four 4-byte ones (a wave's LDS operations execute in order: no barrier)
// (the empty asm statements are compiler fences: the scalar and the vector accesses go through differently typed pointers, and
// nothing else orders them for the optimiser).  Two scratch tiles per wave: tile n + 1 is written while tile n's read is in flight.
#define CFB_FENCE() asm volatile("" ::: "memory")
__device__ __forceinline__ void put_d(float *ws, int q, int i, f32x4 d) {
#pragma unroll
    for (int r = 0; r < 4; ++r) ws[(4 * q + r) * WSP + i] = d[r];
}
__device__ __forceinline__ f32x4 get_d(const float *ws, int q, int i) {
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = ws[(4 * q + r) * WSP + i];
    return d;
}
__device__ __forceinline__ void put_rows(float *ws, int lane, f32x4 v) { *reinterpret_cast<f32x4 *>(ws + (lane >> 2) * WSP + 4 * (lane & 3)) = v; }
__device__ __forceinline__ f32x4 get_rows(const float *ws, int lane) { return *reinterpret_cast<const f32x4 *>(ws + (lane >> 2) * WSP + 4 * (lane & 3)); }

struct FrontArgs {
    View a, b;
    const float *stats0;
    vadx_dfsmn_cfb_weights w;
    float *y1, *stats1, *li, *stats_li;
    int tiles;
};

// First half of the block, same scheme as cfb_back below: every wave does both kinds of work.  Per iteration j of EIGHT bins, wave w
//   * runs the gate / input 1x1 convs of bin 8 j + w (both output-channel tiles share the operand, read straight from global memory
//     one iteration ahead), the gate arithmetic and the LayerNorm statistics          -> R[j & 1] (ln2_w * r), ring (ln1_w * gx),
//   * runs the (3,1) conv of output bin 8 j - 9 + w on the ring (bins written before the last barrier)      -> y1 (global),
//   * accumulates DFT k-steps 2 (j - 1), 2 (j - 1) + 1 of all its 25 tiles from R[(j - 1) & 1],
// the convs' short dependent chains interleaved with the fifty independent DFT MFMAs; ONE LDS-only barrier per iteration.
constexpr int RP8 = 8 * 16;                        // R pitch per channel: [8 bins][16]
constexpr int NSLOT = 18;                          // ring: 18 bin slots per channel = exactly the bins live in an iteration (the (3,1) conv reads
                                                   // 8 j - 10 .. 8 j - 1, the 1x1 convs write 8 j .. 8 j + 7)
// ring pitch per channel: 304 = 48 mod 64 puts the k-quarters of a B read 16 banks apart; the 40-channel instantiation gives the 1280
// bytes of padding to conv_input's fragments instead (in registers they spill, and a scratch reload inside the loop waits out the
// whole in-order prefetch queue: 2.07 -> 1.5x ms) and lives with 2-way conflicts on the (3,1) conv's fifteen reads
template <int CIN> constexpr int ring_pitch() { return CIN == 20 ? NSLOT * 16 + 16 : NSLOT * 16; }
constexpr int NIT8 = F / 8;                        // 20 iterations of front work (+ 2 that drain the (3,1) conv and the DFT)

template <int CIN>
__global__ __launch_bounds__(NTH) void cfb_front_kernel(FrontArgs p) {
    constexpr int KS = CIN / 4, GP20 = ring_pitch<CIN>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *TBL = lds;                              // [10 row tiles][40 k-steps][64 lanes]
    float *R = TBL + TBLF_FLOATS;                  // [2][20][8 bins][16]: ln2_w * r of a chunk, double buffered
    float *GXW = R + 2 * CH * RP8;                 // [20][18 slots][16] pitch GP20: ln1_w * gx, ring over bins
    float *W31 = GXW + CH * GP20;                  // [2 row tiles][15 k-steps][64 lanes]: the (3,1) conv's A fragments
    float *WG = W31 + 2 * 15 * 64;                 // [2][KS][64]: conv_gate's A fragments; behind them conv_input's
    constexpr bool WI_LDS = true;
    float *WS = R;                                 // epilogue only (R and the ring are dead then): [8 waves][2 tiles] transpose scratch
    float *RED = GXW;                              //   "          : reduction scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, i = lane & 15;
    for (int e = tid; e < TBLF_FLOATS / 4; e += NTH) reinterpret_cast<f32x4 *>(TBL)[e] = ldg4(p.w.fwd_tbl + 4 * e);
    for (int e = tid; e < 2 * 15 * 64; e += NTH) W31[e] = p.w.conv_w[((e >> 6) / 15 * 16 + (e & 15)) * 60 + 4 * ((e >> 6) % 15) + ((e >> 4) & 3)];
    for (int e = tid; e < 2 * KS * 64; e += NTH) {
        const int idx = ((e >> 6) / KS * 16 + (e & 15)) * CIN + 4 * ((e >> 6) % KS) + ((e >> 4) & 3);
        WG[e] = p.w.gate_w[idx];
        if (WI_LDS) WG[2 * KS * 64 + e] = p.w.in_w[idx];
    }
    float wif[WI_LDS ? 1 : 2][WI_LDS ? 1 : KS], bi[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (!WI_LDS) {
#pragma unroll
            for (int s = 0; s < KS; ++s) wif[WI_LDS ? 0 : m][WI_LDS ? 0 : s] = p.w.in_w[(m * 16 + i) * CIN + 4 * s + q];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bi[m][r] = p.w.in_b[m * 16 + 4 * q + r];
    }
    const float *wg_w = WG + lane, *wi_w = WG + 2 * KS * 64 + lane;
    const int ac = p.a.c, ks_a = ac >> 2;
    const bool up_ok = q == 0;                     // rows 16 + 4 q + r of the upper tile exist for q == 0 only (20 channels)
    const int cg = wave >> 1, mg = wave & 1;       // DFT tiles: channels 5 cg .. + 4, row tiles 5 mg .. + 4; tile jj = (jj / 5, jj % 5)
    const float *tbl_w = TBL + (5 * mg * KSF) * 64 + lane;
    const float *r_w = R + (5 * cg) * RP8 + q * 16 + i;
    float *ws = WS + wave * WS_FLOATS;
    __syncthreads();

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        f32x4 acc[25];
#pragma unroll
        for (int jj = 0; jj < 25; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float mean0 = p.stats0[((size_t)tile * 16 + i) * 2], inv0 = p.stats0[((size_t)tile * 16 + i) * 2 + 1];
        const float mi0 = mean0 * inv0;
        // Every global access below is a UNIFORM base (scalar registers) plus an unsigned 32-bit per-lane offset: with per-lane 64-bit
        // pointers the address pairs alone overflowed the register file, and a scratch reload inside the loop waits out the whole
        // in-order prefetch queue.
        const float *abase = p.a.ptr + ((size_t)tile * p.a.c_total + p.a.c_off) * F * 16;
        const float *bbase = p.b.ptr ? p.b.ptr + ((size_t)tile * p.b.c_total + p.b.c_off - ac) * F * 16 : abase;
        const unsigned lq = (unsigned)(q * F * 16 + i);      // channel q of a k-step, frame i
        float x0[KS], x1[KS];                       // input rows of even / odd iterations (ping-pong: no copies), requested an iteration ahead
        auto load_x = [&](int j, float (&x)[KS]) {
            const unsigned f = (unsigned)min(8 * j + wave, F - 1);
            unsigned lql = lq;                      // laundered: left visible, the compiler forms one 64-bit pointer per k-step outside the
            asm volatile("" : "+v"(lql));           // loop and spills them; the reloads then wait out the whole prefetch queue
#pragma unroll
            for (int s = 0; s < KS; ++s) x[s] = (CFB_EXP & 2) ? 0.25f : ldg1o(s < ks_a ? abase : bbase, 4u * (lql + (unsigned)(s * 4 * F * 16) + f * 16u));
        };
        load_x(0, x0);
        for (int e = tid; e < CH * 16; e += NTH) GXW[(e >> 4) * GP20 + (NSLOT - 1) * 16 + (e & 15)] = 0.f;      // bin -1: zero padding of the (3,1) conv
        Acc1 sg, sr;
        sg.init(); sr.init();
        float *y1_t = p.y1 + (size_t)tile * (CH * F * 16);
        // One iteration, straight-line code: C31 = the (3,1) conv of output bin 8 j - 9 + w (its result is stored only if that bin exists),
        // DFT = k-steps 2 (j - 1), 2 (j - 1) + 1, FRONT = the 1x1 convs + gate arithmetic of bin 8 j + w.  Order: the L2-resident
        // tables of this bin are requested first, the (3,1) conv and the DFT run, then the 1x1 convs (their operand xc was requested an
        // iteration ago), the gate arithmetic, the LDS hand-over, and the y1 stores last.
        auto iter = [&](auto front_c, auto c31_c, auto dft_c, int j, const float (&xc)[KS]) {
            constexpr bool FRONT = decltype(front_c)::value, C31 = decltype(c31_c)::value, DFT = decltype(dft_c)::value;
            const int f = 8 * j + wave, fo = 8 * j - 9 + wave;
            unsigned qv = (unsigned)q, iv = (unsigned)i;      // laundered lane indices for this iteration's global addresses (see load_x)
            asm volatile("" : "+v"(qv), "+v"(iv));
            float w0c[KS];
            if (FRONT) {
#pragma unroll
                for (int s = 0; s < KS; ++s) w0c[s] = ldg1o(p.w.ln0_w, 4u * (qv * (unsigned)F + (unsigned)(4 * s * F + f)));
            }
            f32x4 a3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            const float *rb_p = r_w + ((j - 1) & 1) * CH * RP8;
            const int sl0 = (fo + NSLOT - 1) % NSLOT;    // slot of bin fo - 1 (fo >= -1)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float tf[5], rb[5];
                __builtin_amdgcn_sched_barrier(0);      // one k-step's fragment reads at a time
                if (DFT && !(CFB_EXP & 16)) {
#pragma unroll
                    for (int u = 0; u < 5; ++u) { tf[u] = tbl_w[(u * KSF + 2 * (j - 1) + ks) * 64]; rb[u] = rb_p[u * RP8 + ks * 64]; }
                }
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    if (C31 && !(CFB_EXP & 8)) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int s = (ks * 5 + g) * 2 + h;          // 0..19; the fifteen k-steps run at s < 15
                            if (s < 15) {
                                const int slot = sl0 + s / 5 >= NSLOT ? sl0 + s / 5 - NSLOT : sl0 + s / 5;
                                const float bv = GXW[(4 * (s % 5) + q) * GP20 + slot * 16 + i];
                                a3[0] = mfma16(W31[s * 64 + lane], bv, a3[0]);
                                a3[1] = mfma16(W31[(15 + s) * 64 + lane], bv, a3[1]);
                            }
                        }
                    }
                    if (DFT && !(CFB_EXP & 16)) {
#pragma unroll
                        for (int u = 0; u < 5; ++u) acc[g * 5 + u] = mfma16(tf[u], rb[g], acc[g * 5 + u]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- y1 rows of output bin fo (stores: behind every load this iteration still waits for -- a wave's memory counter retires
            // in order, and a load issued behind a store waits out the store's acknowledgement too)
            if (FRONT) {
                // the two output-channel tiles one after the other (together their accumulators and table rows are 48 more live registers
                // than this kernel has): table rows (GW, GB, ln1_w, ln2_w; L2-resident) requested, the tile's two chains of KS MFMAs
                // run, then the gate arithmetic of its rows
                float xw[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) xw[s] = xc[s] * w0c[s];
                const int slot = f % NSLOT;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    __builtin_amdgcn_sched_barrier(0);      // (keeps the second tile's loads and MFMAs behind the first tile's arithmetic)
                    f32x4 ta[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        ta[k] = ldg4o(p.w.front_tab, 4u * ((unsigned)((f * 4 + k) * CH) + (m == 0 ? 4u * qv : 16u)));
                    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ai = {bi[m][0], bi[m][1], bi[m][2], bi[m][3]};
                    if (!(CFB_EXP & 4)) {
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            ag = mfma16(wg_w[(m * KS + s) * 64], xw[s], ag);
                            ai = mfma16(WI_LDS ? wi_w[(m * KS + s) * 64] : wif[WI_LDS ? 0 : m][WI_LDS ? 0 : s], xc[s], ai);
                        }
                    }
                    if (m == 1 && !up_ok) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = m * 16 + 4 * q + r;
                        // LN0 commuted behind the gate conv: inv0 * (Wg (w0 x)) + (Wg b0 + bias) - mean0 inv0 (Wg w0)
                        const float pre = fmaf(inv0, ag[r], fmaf(-mi0, ta[0][r], ta[1][r]));
                        const float g = (CFB_EXP & 32) ? pre : gate_sigmoid(pre), xi = ai[r], gx = g * xi, rr = xi - gx;
                        if (!C31 && m == 0 && r == 0) { sg.K = gx; sr.K = rr; }      // the tile's first iteration (no (3,1) conv yet): the shifts
                        if (!(CFB_EXP & 32)) { sg.addk(gx); sr.addk(rr); }
                        GXW[co * GP20 + slot * 16 + i] = gx * ta[2][r];
                        R[(j & 1) * CH * RP8 + co * RP8 + wave * 16 + i] = rr * ta[3][r];
                    }
                }
            } else if (j == NIT8 && wave == 0) {
                for (int e = lane; e < CH * 16; e += 64) GXW[(e >> 4) * GP20 + (F % NSLOT) * 16 + (e & 15)] = 0.f;      // bin 160: zero padding
            }
            if (C31 && fo >= 0 && fo < F && !(CFB_EXP & (8 | 64))) {
                const unsigned o = qv * (unsigned)(4 * F * 16) + iv + (unsigned)(fo * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) stg1o(y1_t, 4u * (o + (unsigned)(r * F * 16)), a3[0][r]);
                if (up_ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg1o(y1_t, 4u * (o + (unsigned)((16 + r) * F * 16)), a3[1][r]);
                }
            }
            lds_barrier();
        };
        using T_ = std::true_type; using F_ = std::false_type;
        load_x(1, x1);
        iter(T_{}, F_{}, F_{}, 0, x0);
        for (int j = 1; j < NIT8 - 1; j += 2) {         // (1, 2), (3, 4), ... (17, 18)
            load_x(j + 1, x0);
            iter(T_{}, T_{}, T_{}, j, x1);
            load_x(j + 2, x1);
            iter(T_{}, T_{}, T_{}, j + 1, x0);
        }
        iter(T_{}, T_{}, T_{}, NIT8 - 1, x1);           // 19
        iter(F_{}, T_{}, T_{}, NIT8, x0);               // 20: drains the (3,1) conv (bins 151 .. 158) and the DFT's last two k-steps
        iter(F_{}, T_{}, F_{}, NIT8 + 1, x0);           // 21: bin 159
        // ---- LayerNorm statistics of gx (LN1, handed to cfb_back) and r (LN2, applied below): every wave saw one bin in eight
        float n1, m1, M1, n2, m2, M2;
        sg.finish(n1, m1, M1);
        sr.finish(n2, m2, M2);
        chan_merge_q(n1, m1, M1);
        chan_merge_q(n2, m2, M2);
        if (lane < 16) {
            float *o = RED + (wave * 16 + i) * 6;
            o[0] = n1; o[1] = m1; o[2] = M1; o[3] = n2; o[4] = m2; o[5] = M2;
        }
        lds_barrier();
        float *RES = RED + 8 * 16 * 6;
        if (tid < 16) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float *o = RED + (w * 16 + tid) * 6;
                chan_merge(a0, a1, a2, o[0], o[1], o[2]);
                chan_merge(b0, b1, b2, o[3], o[4], o[5]);
            }
            const float inv1 = 1.0f / (sqrtf(a2 / (a0 - 1.f)) + 1e-6f), sd2 = sqrtf(b2 / (b0 - 1.f)) + 1e-6f;
            p.stats1[((size_t)tile * 16 + tid) * 2] = a1;
            p.stats1[((size_t)tile * 16 + tid) * 2 + 1] = inv1;
            RES[tid * 4] = b1; RES[tid * 4 + 1] = sd2; RES[tid * 4 + 2] = 1.0f / sd2;
        }
        lds_barrier();
        const float mean2 = RES[i * 4], sd2 = RES[i * 4 + 1], inv2 = RES[i * 4 + 2];
        const float bfix = q == 0 ? -mean2 : (q == 1 ? sd2 : 0.f);
        float ssum = 0.f;
        int lane_l = lane, ql = q, il = i;
        asm volatile("" : "+v"(lane_l), "+v"(ql), "+v"(il));
        const float *fix_w = p.w.fwd_fix + ((5 * cg) * 10 + 5 * mg) * 64;
        const unsigned lane_u = (unsigned)lane_l;
        {
            float fx[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) fx[u] = ldg1o(fix_w, 4u * (lane_u + (unsigned)(u * 64)));
#pragma unroll
            for (int jj = 0; jj < 25; ++jj) {
                acc[jj] = mfma16(fx[jj % 5], bfix, acc[jj]);
                if (jj + 5 < 25) fx[jj % 5] = ldg1o(fix_w, 4u * (lane_u + (unsigned)((((jj + 5) / 5) * 10 + (jj + 5) % 5) * 64)));
                acc[jj] *= inv2;
                ssum += (acc[jj][0] + acc[jj][1]) + (acc[jj][2] + acc[jj][3]);
            }
        }
        // ---- statistics of S over (40 channels, 81 bins): 160 stored values + 2 structural zeros per channel, two passes in registers
        float *RD = RES + 64;
        ssum = sum_q(ssum);
        if (lane < 16) RD[wave * 16 + i] = ssum;
        lds_barrier();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += RD[w * 16 + i];
        const float nS = (float)(2 * CH * CF), meanS = tot / nS;
        float dsum = 0.f;
#pragma unroll
        for (int jj = 0; jj < 25; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc[jj][r] - meanS; dsum = fmaf(d, d, dsum); }
        dsum = sum_q(dsum);
        if (lane < 16) RD[128 + wave * 16 + i] = dsum;
        lds_barrier();
        if (tid < 16) {
            float M = 2.f * CH * meanS * meanS;                  // the imaginary parts of bins 0 and 80
#pragma unroll
            for (int w = 0; w < 8; ++w) M += RD[128 + w * 16 + tid];
            p.stats_li[((size_t)tile * 16 + tid) * 2] = meanS;
            p.stats_li[((size_t)tile * 16 + tid) * 2 + 1] = 1.0f / (sqrtf(M / (nS - 1.f)) + 1e-6f);
        }
        // ---- S -> li[40][81]: table rows 0..80 = cos bins of channel c, rows 81..159 = sin bins 1..79 of channel 20 + c; one 16-byte
        // store per lane and tile through the wave's transpose scratch
        if (!(CFB_EXP & 256)) {
            float *li_t = p.li + (size_t)tile * (2 * CH * CF * 16);
            CFB_FENCE();
            put_d(ws, ql, il, acc[0]);
#pragma unroll
            for (int jj = 0; jj < 25; ++jj) {
                const int c = 5 * cg + jj / 5, m = 5 * mg + jj % 5;
                CFB_FENCE();
                if (jj + 1 < 25) put_d(ws + ((jj + 1) & 1) * WS_TILE, ql, il, acc[jj + 1]);
                CFB_FENCE();
                const f32x4 v = get_rows(ws + (jj & 1) * WS_TILE, lane_l);
                CFB_FENCE();
                const int row = 16 * m + (lane_l >> 2);
                const int ch = row <= 80 ? c : CH + c, bin = row <= 80 ? row : row - 80;
                stg4o(li_t, 4u * (unsigned)((ch * CF + bin) * 16 + 4 * (lane_l & 3)), v);
            }
            for (int e = tid; e < CH * 2 * 16; e += NTH)
                li_t[((CH + (e >> 5)) * CF + (((e >> 4) & 1) ? 80 : 0)) * 16 + (e & 15)] = 0.f;
        }
        lds_barrier();                              // the scratch aliases R and the ring
    }
}

struct BackArgs {
    const float *hf, *li, *y1, *stats1;
    vadx_dfsmn_cfb_weights w;
    ViewW out;
    float *part;
    int tiles;
};

// Second half of the block.  Every wave does BOTH kinds of work here (the producer / DFT-wave split of cfb_front left each role
// waiting on the other: cycle accounting showed the producers stalled 42 % on their operand loads and the DFT waves 37 % in a
// load-bound prologue and at barriers -- memory time and MFMA time added up).  Per iteration of EIGHT ceps bins, wave w
//   * computes CepsUnit's Linear 40 -> 40 for bin 8 c + w (three row tiles sharing the operand, read straight from global memory
//     one iteration = ~7 us ahead) and the complex product with the spectrum  -> OB[c & 1],
//   * accumulates the pinv inverse DFT of the PREVIOUS chunk (two real-part and two imaginary-part k-steps: 100 MFMAs on its 25
//     accumulator tiles) from OB[(c - 1) & 1],
// the Linear's three dependent chains interleaved with the independent DFT MFMAs; ONE LDS-only barrier per iteration.
constexpr int OP8 = 2 * 8 * 16;                    // OB pitch per channel: [re | im][8 bins][16]
constexpr int NCH8 = 11;                           // chunks of eight ceps bins (81 -> 88)

__global__ __launch_bounds__(NTH) void cfb_back_kernel(BackArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *TI = lds;                               // [10 row tiles][41 k-steps][64 lanes]: 21 real-part + 20 imaginary-part steps
    float *OB = TI + TBLI_FLOATS;                  // [2][20][re | im][8 bins][16]: complex product, double buffered
    float *WL = OB + 2 * CH * OP8;                 // [3 row tiles][10 k-steps][64 lanes]: the Linear's A fragments (30 registers otherwise)
    float *WS = OB;                                // prologue / epilogue only (OB is dead then): [8 waves][2 tiles] transpose scratch
    float *RED = OB + 8 * WS_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, i = lane & 15;
    for (int e = tid; e < TBLI_FLOATS / 4; e += NTH) reinterpret_cast<f32x4 *>(TI)[e] = ldg4(p.w.inv_tbl + 4 * e);
    // Linear 40 -> 40 with rows permuted so that a lane's four D rows are (re c, re c+1, im c, im c+1), c = 8 mt + 2 q
    for (int e = tid; e < 3 * 10 * 64; e += NTH) WL[e] = p.w.lin_w[((e >> 6) / 10 * 16 + (e & 15)) * 40 + 4 * ((e >> 6) % 10) + ((e >> 4) & 3)];
    float bl[3][4];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) bl[m][r] = p.w.lin_b[m * 16 + 4 * q + r];
    const float *wl_w = WL + lane;
    const int cg = wave >> 1, mg = wave & 1;           // DFT tiles: channels 5 cg .. + 4, row tiles 5 mg .. + 4; tile jj = (jj / 5, jj % 5)
    const float *tbl_w = TI + (5 * mg * KSI) * 64 + lane;
    const float *o_w = OB + (5 * cg) * OP8 + q * 16 + i;
    float *ws = WS + wave * WS_FLOATS;
    __syncthreads();
    CFB_T0();

    f32x4 acc[25];
    // this lane's place in a tile's row layout + the wave's first tile (uniform bases + unsigned 32-bit offsets everywhere: see cfb_front)
    const unsigned rowoff = (unsigned)((lane >> 2) * 16 + 4 * (lane & 3) + ((5 * cg) * F + 16 * (5 * mg)) * 16);
    auto y1_load = [&](int tile) {                  // the accumulators START as y1 (row layout; transposed below)
        const float *y1_t = p.y1 + (size_t)tile * (CH * F * 16);
#pragma unroll
        for (int jj = 0; jj < 25; ++jj) acc[jj] = (CFB_EXP & 512) ? f32x4{0.f, 0.f, 0.f, 0.f} : ldg4o(y1_t, 4u * (rowoff + (unsigned)(((jj / 5) * F + 16 * (jj % 5)) * 16)));
    };
    if ((int)blockIdx.x < p.tiles) y1_load(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        // ---- operands of this wave's Linear: bin 8 c + w, requested one iteration ahead
        // (uniform bases + unsigned 32-bit per-lane offsets: see cfb_front)
        const float *hf_t = p.hf + (size_t)tile * (2 * CH * CF * 16);
        const float *li_t = p.li + (size_t)tile * (2 * CH * CF * 16);
        const unsigned lq = (unsigned)(q * CF * 16 + i), li_ = (unsigned)i;
        float h0[10], h1[10], s0[12], s1[12];       // operand sets of even / odd chunks (ping-pong: no copies, so a set's loads are
                                                    // waited for only where the next iteration first uses them)
        auto load = [&](int c, float (&h)[10], float (&sv)[12]) {
            const int bin = min(8 * c + wave, CF - 1);            // bins past 80 meet zero table rows: any finite value
            unsigned lql = lq + (unsigned)(bin * 16), lil = li_ + (unsigned)(bin * 16);      // laundered: the 22 offsets are formed at each call,
            asm volatile("" : "+v"(lql), "+v"(lil));                                            // not hoisted and spilled
#pragma unroll
            for (int s = 0; s < 10; ++s) h[s] = (CFB_EXP & 2) ? 0.25f : ldg1o(hf_t, 4u * (lql + (unsigned)(s * 4 * CF * 16)));
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int ca = min(8 * m + 2 * q, CH - 2);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    sv[4 * m + h2] = (CFB_EXP & 2) ? 0.5f : ldg1o(li_t, 4u * (lil + (unsigned)((ca + h2) * CF * 16)));
                    sv[4 * m + 2 + h2] = (CFB_EXP & 2) ? 0.5f : ldg1o(li_t, 4u * (lil + (unsigned)((CH + ca + h2) * CF * 16)));
                }
            }
        };
        load(0, h0, s0);
        // ---- acc = inv1 * (y1 - mean1 * CW) + CB: y1 arrived as one 16-byte load per lane and tile (requested during the previous
        // tile's epilogue), is turned into the MFMA D layout through the wave's LDS scratch; the (CW, CB) term is one k-step
        const float mean1 = p.stats1[((size_t)tile * 16 + i) * 2], inv1 = p.stats1[((size_t)tile * 16 + i) * 2 + 1];
        const float bfix = q == 0 ? -mean1 * inv1 : (q == 1 ? 1.f : 0.f);
        int ql = q, il = i, lane_l = lane;
        asm volatile("" : "+v"(ql), "+v"(il), "+v"(lane_l));      // keeps the tile addresses out of the tile loop's preheader
        const float *fix_w = p.w.out_fix + ((5 * cg) * 10 + 5 * mg) * 64;
        const unsigned lane_u = (unsigned)lane_l;
        {
            float fx[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) fx[u] = ldg1o(fix_w, 4u * (lane_u + (unsigned)(u * 64)));
            CFB_FENCE();
            put_rows(ws, lane_l, acc[0]);
#pragma unroll
            for (int jj = 0; jj < 25; ++jj) {
                CFB_FENCE();
                if (jj + 1 < 25) put_rows(ws + ((jj + 1) & 1) * WS_TILE, lane_l, acc[jj + 1]);
                CFB_FENCE();
                f32x4 d = get_d(ws + (jj & 1) * WS_TILE, ql, il);
                CFB_FENCE();
                d *= inv1;
                acc[jj] = mfma16(fx[jj % 5], bfix, d);
                if (jj + 5 < 25) fx[jj % 5] = ldg1o(fix_w, 4u * (lane_u + (unsigned)((((jj + 5) / 5) * 10 + (jj + 5) % 5) * 64)));
            }
        }
        CFB_MARK(0);
        lds_barrier();                              // the scratch aliases OB
        // One iteration: Linear + complex product of chunk c from (hc, sc) -> OB[c & 1], and NK k-steps of the inverse DFT of chunk
        // c - 1 from OB[(c - 1) & 1] (NK = 4: real 2 cm, 2 cm + 1, imaginary 2 cm, 2 cm + 1; NK = 1: the last chunk holds bin 80 only).
        // Straight-line code: the Linear's ten steps (three chains) sit at every other slot between the DFT's groups of five MFMAs.
        auto iter = [&](auto lin_c, auto nk_c, int c, const float (&hc)[10], const float (&sc)[12]) {
            constexpr bool LIN = decltype(lin_c)::value;
            constexpr int NK = decltype(nk_c)::value;
            f32x4 P[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) P[m] = f32x4{bl[m][0], bl[m][1], bl[m][2], bl[m][3]};
            const float *ob_p = o_w + ((c - 1) & 1) * CH * OP8;
            const int cm = c - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int part = ks >> 1, half = ks & 1;
                float tf[5], ob[5];
                __builtin_amdgcn_sched_barrier(0);      // one k-step's ten fragment reads at a time (hoisted together they spill)
                if (ks < NK && !(CFB_EXP & 16)) {
#pragma unroll
                    for (int u = 0; u < 5; ++u) { tf[u] = tbl_w[(u * KSI + part * 21 + 2 * cm + half) * 64]; ob[u] = ob_p[u * OP8 + part * 128 + half * 64]; }
                }
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    if (LIN && !(CFB_EXP & 128) && ((ks * 5 + g) & 1) == 0) {
                        const int s = (ks * 5 + g) >> 1;
#pragma unroll
                        for (int m = 0; m < 3; ++m) P[m] = mfma16(wl_w[(m * 10 + s) * 64], hc[s], P[m]);
                    }
                    if (ks < NK && !(CFB_EXP & 16)) {
#pragma unroll
                        for (int u = 0; u < 5; ++u) acc[g * 5 + u] = mfma16(tf[u], ob[g], acc[g * 5 + u]);
                    }
                }
            }
            if (LIN) {
                float *ob = OB + (c & 1) * CH * OP8 + wave * 16 + i;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int ca = 8 * m + 2 * q;
                    if (ca < CH) {
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const float sre = sc[4 * m + h2], sim = sc[4 * m + 2 + h2], pr = P[m][h2], pi = P[m][2 + h2];
                            ob[(ca + h2) * OP8] = pr * sre - pi * sim;
                            ob[(ca + h2) * OP8 + 128] = pr * sim + pi * sre;
                        }
                    }
                }
            }
            CFB_MARK(1);
            lds_barrier();
            CFB_MARK(2);
        };
        using T_ = std::true_type; using F_ = std::false_type;
        load(1, h1, s1);
        iter(T_{}, std::integral_constant<int, 0>{}, 0, h0, s0);
        for (int c = 1; c < NCH8 - 1; c += 2) {          // chunks (1, 2), (3, 4), ... (9, 10): all four k-steps of chunks 0 .. 9 are real
            load(c + 1, h0, s0);
            iter(T_{}, std::integral_constant<int, 4>{}, c, h1, s1);
            if (c + 2 < NCH8) load(c + 2, h1, s1);
            iter(T_{}, std::integral_constant<int, 4>{}, c + 1, h0, s0);
        }
        iter(F_{}, std::integral_constant<int, 1>{}, NCH8, h0, s0);        // chunk 10: bin 80, real part only
        // ---- out: one 16-byte store per lane and tile; as a tile's registers leave, the NEXT tile's y1 is requested into them, so that
        // its latency passes under this epilogue's reductions; (count, mean, M2) of the output per frame for the next LayerNorm
        float ssum = 0.f, dsum = 0.f;
#pragma unroll
        for (int jj = 0; jj < 25; ++jj) ssum += (acc[jj][0] + acc[jj][1]) + (acc[jj][2] + acc[jj][3]);
        ssum = sum_q(ssum);
        if (lane < 16) RED[wave * 16 + i] = ssum;
        lds_barrier();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += RED[w * 16 + i];
        const float nO = (float)(CH * F), meanO = tot / nO;
#pragma unroll
        for (int jj = 0; jj < 25; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc[jj][r] - meanO; dsum = fmaf(d, d, dsum); }
        dsum = sum_q(dsum);
        if (lane < 16) RED[128 + wave * 16 + i] = dsum;
        float *out_t = p.out.ptr + ((size_t)tile * p.out.c_total + p.out.c_off) * (F * 16);
        const int next = tile + gridDim.x;
        const float *y1_n = p.y1 + (size_t)(next < p.tiles ? next : tile) * (CH * F * 16);
        unsigned ro = rowoff;
        asm volatile("" : "+v"(ql), "+v"(il), "+v"(lane_l), "+v"(ro));
        CFB_FENCE();
        put_d(ws, ql, il, acc[0]);
#pragma unroll
        for (int jj = 0; jj < 25; ++jj) {
            const unsigned off = ro + (unsigned)(((jj / 5) * F + 16 * (jj % 5)) * 16);
            CFB_FENCE();
            if (jj + 1 < 25) put_d(ws + ((jj + 1) & 1) * WS_TILE, ql, il, acc[jj + 1]);
            CFB_FENCE();
            const f32x4 v = get_rows(ws + (jj & 1) * WS_TILE, lane_l);
            CFB_FENCE();
            if (!(CFB_EXP & 256)) stg4o(out_t, 4u * off, v);
            if (jj >= 1)        // tile jj - 1's registers are free now: the next tile's y1 lands in them during the reductions below
                acc[jj - 1] = (CFB_EXP & 512) ? f32x4{0.f, 0.f, 0.f, 0.f} : ldg4o(y1_n, 4u * (ro + (unsigned)((((jj - 1) / 5) * F + 16 * ((jj - 1) % 5)) * 16)));
        }
        acc[24] = (CFB_EXP & 512) ? f32x4{0.f, 0.f, 0.f, 0.f} : ldg4o(y1_n, 4u * (ro + (unsigned)((4 * F + 16 * 4) * 16)));
        lds_barrier();
        if (p.part && tid < 16) {
            float M = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) M += RED[128 + w * 16 + tid];
            float *o = p.part + (((size_t)tile * VADX_DFSMN_STAT_PARTS) * 16 + tid) * 4;
            o[0] = nO; o[1] = meanO; o[2] = M; o[3] = 0.f;
            float *z = o + 16 * 4;
            z[0] = z[1] = z[2] = z[3] = 0.f;
        }
        CFB_MARK(3);
    }
    CFB_FLUSH(8);
}

// =====================================================================================================================================
// The same two halves on bf16 x 3 exact-split products (csrc/split3.h): every matrix product of the block whose one operand is a
// constant -- the 1x1 convs, the (3,1) conv, the 160 x 160 forward / inverse DFT tables, CepsUnit's Linear -- runs as six
// v_mfma_f32_16x16x32_bf16 per K = 32 step.  The streaming structure follows the K granularity:
//   * the frequency axis goes in FIVE groups of 32 bins = one bf16 k-step of the DFT; the table is streamed from L2 as split fragments
//     (three planes no longer fit LDS), the DFT of all 20 channels of a 16-frame tile stays in registers as before (wave (cg, mg):
//     channels 5 cg .. + 4, row tiles 5 mg .. + 4);
//   * per group, wave w runs the 1x1 convs + gate arithmetic of bins 32 g + 4 w .. + 3, leaves ln1_w gx in its four ring slots and
//     ln2_w r as its half k-group of the group's B planes | barrier A | every wave runs the (3,1) conv of output bins 32 g + 4 w - 1 ..
//     + 2 from its own slots and its lower neighbour's last two (wave 0: the carry slots wave 7 left behind the previous group) and
//     the group's DFT k-step | barrier B.
// K orders are free, so the 1x1 convs take their input channels five per lane quarter (20 channels = one k-step, 3 of 8 slots zero),
// and the (3,1) conv takes (tap, channel quad) pairs: the ring holds 8-byte (quad, frame) units, exactly what a producer lane stores.
constexpr int RQ_ROW = 256, RQ_PLANE = CH * 4 * RQ_ROW;                    // B planes of one group: [channel][k-group][16 frames][8 bins] bf16
constexpr int RING_SLOT = 5 * 16 * 8, RING_PLANE = 4 * RING_SLOT, RING_WAVE = 3 * RING_PLANE;      // [4 slots][5 quads][16 frames][4 ch] bf16 per plane
constexpr int CARRY_PLANE = 2 * RING_SLOT;                                 // bins 32 g - 2, 32 g - 1 of the previous group
constexpr int QF_B = 1024;                                                 // bytes of one bf16 A fragment
template <int CIN> constexpr int front_split_lds() { return 3 * RQ_PLANE + 8 * RING_WAVE + 3 * CARRY_PLANE + (2 * 2 * 3 + 2 * 2 * (CIN / 20) * 3) * QF_B; }
static_assert(front_split_lds<40>() <= 160 * 1024, "LDS budget of cfb_front_split");
static_assert((8 * WS_FLOATS + 8 * 16 * 6 + 64 + 256) * 4 <= 3 * RQ_PLANE, "the split front kernel's epilogue scratch aliases the B planes");

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// five float32 values (k-slots 0..4 of a lane's eight; 5..7 are zero) -> the lane's B fragment in each plane
__device__ __forceinline__ void split5(const float (&v)[5], bf16x8 (&b)[3]) {
    unsigned t0[5], t1[5], t2[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        t0[e] = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(t0[e] & 0xffff0000u);
        t1[e] = __float_as_uint(r1);
        t2[e] = __float_as_uint(r1 - __uint_as_float(t1[e] & 0xffff0000u));
    }
    b[0] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_perm(t0[1], t0[0], 0x07060302u), __builtin_amdgcn_perm(t0[3], t0[2], 0x07060302u), t0[4] >> 16, 0u});
    b[1] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_perm(t1[1], t1[0], 0x07060302u), __builtin_amdgcn_perm(t1[3], t1[2], 0x07060302u), t1[4] >> 16, 0u});
    b[2] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_perm(t2[1], t2[0], 0x07060302u), __builtin_amdgcn_perm(t2[3], t2[2], 0x07060302u), t2[4] >> 16, 0u});
}

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char *p) { return *reinterpret_cast<const bf16x8 *>(p); }
// element k (wave-uniform, run-time) of a register vector: four selects instead of a scratch round trip
__device__ __forceinline__ void set_k(f32x4 &v, int k, float x) {
    v[0] = k == 0 ? x : v[0]; v[1] = k == 1 ? x : v[1]; v[2] = k == 2 ? x : v[2]; v[3] = k == 3 ? x : v[3];
}

struct FrontQArgs {
    FrontArgs f;
    const float *tbl_q;           // forward table as split A fragments [10 row tiles][5 chunks][3 planes][QFRAG]
};

template <int CIN>
__global__ __launch_bounds__(NTH) void cfb_front_split_kernel(FrontQArgs pq) {
    const FrontArgs &p = pq.f;
    constexpr int KC = CIN / 20;
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
    unsigned char *RQ = qlds;                                   // 3 planes
    unsigned char *RING = RQ + 3 * RQ_PLANE;                    // [8 waves][3 planes][4 slots]
    unsigned char *CARRY = RING + 8 * RING_WAVE;                // [3 planes][2 slots]
    unsigned char *W31 = CARRY + 3 * CARRY_PLANE;               // [2 row tiles][2 chunks][3 planes] fragments
    unsigned char *WG = W31 + 2 * 2 * 3 * QF_B;                 // [2][KC][3]
    unsigned char *WI = WG + 2 * KC * 3 * QF_B;
    float *WS = reinterpret_cast<float *>(RQ);                  // epilogue only (the planes are dead then): [8 waves][2 tiles] transpose scratch
    float *RED = WS + 8 * WS_FLOATS;                            //   "          : reduction scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, i = lane & 15;
    // ---- weights -> split A fragments in LDS (once per launch: the workgroup is persistent)
    for (int e = tid; e < 2 * 2 * 64 * 8; e += NTH) {           // (3,1) conv: slot 8 g + el <-> pair u = 2 g + el / 4 = (tap u / 5, quad u % 5)
        const int el = e & 7, l = (e >> 3) & 63, ck = (e >> 9) & 1, m = e >> 10;
        const int u = 2 * (4 * ck + (l >> 4)) + (el >> 2);
        // (second row tile: channel 16 + j on row 4 j -- a lane's D rows 4 q .. 4 q + 3 then hold ONE real channel, 16 + q, in r = 0)
        const int row = m == 0 ? (l & 15) : (((l & 3) == 0) ? 16 + ((l & 15) >> 2) : -1);
        const float w = (u < 15 && row >= 0) ? p.w.conv_w[row * 60 + (u / 5) * 20 + 4 * (u % 5) + (el & 3)] : 0.f;
        unsigned short h0, h1, h2;
        split3x1(w, h0, h1, h2);
        unsigned short *d = reinterpret_cast<unsigned short *>(W31 + (m * 2 + ck) * 3 * QF_B) + l * 8 + el;
        d[0] = h0; d[QF_B / 2] = h1; d[QF_B] = h2;
    }
    for (int e = tid; e < 2 * KC * 64 * 8; e += NTH) {          // 1x1 convs: slot 8 q + el of chunk ck <-> channel 20 ck + 5 q + el (el < 5)
        const int el = e & 7, l = (e >> 3) & 63, ck = (e >> 9) % KC, m = (e >> 9) / KC;
        const int ch = 20 * ck + 5 * (l >> 4) + el, row = m == 0 ? (l & 15) : (((l & 3) == 0) ? 16 + ((l & 15) >> 2) : -1);
        const float wg = (el < 5 && row >= 0) ? p.w.gate_w[row * CIN + ch] : 0.f, wi = (el < 5 && row >= 0) ? p.w.in_w[row * CIN + ch] : 0.f;
        unsigned short h0, h1, h2;
        split3x1(wg, h0, h1, h2);
        unsigned short *d = reinterpret_cast<unsigned short *>(WG + (m * KC + ck) * 3 * QF_B) + l * 8 + el;
        d[0] = h0; d[QF_B / 2] = h1; d[QF_B] = h2;
        split3x1(wi, h0, h1, h2);
        d = reinterpret_cast<unsigned short *>(WI + (m * KC + ck) * 3 * QF_B) + l * 8 + el;
        d[0] = h0; d[QF_B / 2] = h1; d[QF_B] = h2;
    }
    float bi[2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bi[0][r] = p.w.in_b[4 * q + r]; bi[1][r] = r == 0 ? p.w.in_b[16 + q] : 0.f; }
    unsigned char *ring = RING + wave * RING_WAVE;
    // the (3,1) conv's B operand: k-group 4 chunk + q is the pairs u = 2 (4 chunk + q) + half; (tap, byte offset inside a slot) per (chunk, half)
    int c31_tap[2][2], c31_off[2][2];
    bool c31_ok[2][2];
#pragma unroll
    for (int ck = 0; ck < 2; ++ck)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int u = 2 * (4 * ck + q) + h;
            c31_ok[ck][h] = u < 15;
            c31_tap[ck][h] = u < 15 ? u / 5 : 0;
            c31_off[ck][h] = ((u < 15 ? u % 5 : 0) * 16 + i) * 8;
        }
    const int cg = wave >> 1, mg = wave & 1;       // DFT tiles: channels 5 cg .. + 4, row tiles 5 mg .. + 4; tile jj = (jj / 5, jj % 5)
    float *ws = WS + wave * WS_FLOATS;
    __syncthreads();
    CFB_T0();

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        f32x4 acc[25];
#pragma unroll
        for (int jj = 0; jj < 25; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float mean0 = p.stats0[((size_t)tile * 16 + i) * 2], inv0 = p.stats0[((size_t)tile * 16 + i) * 2 + 1];
        const float mi0 = mean0 * inv0;
        const float *abase = p.a.ptr + ((size_t)tile * p.a.c_total + p.a.c_off) * F * 16;
        const float *bbase = p.b.ptr ? p.b.ptr + ((size_t)tile * p.b.c_total + p.b.c_off) * F * 16 : abase;
        float *y1_t = p.y1 + (size_t)tile * (CH * F * 16);
        Acc1 sg, sr;
        sg.init(); sr.init();
        if (wave == 7)                                          // bins -2, -1: the zero padding of the (3,1) conv
            for (int e = lane; e < 3 * CARRY_PLANE / 8; e += 64) reinterpret_cast<u32x2 *>(CARRY)[e] = u32x2{0u, 0u};
        float xn[KC][5], wn[KC][5];
        auto load_x = [&](int f) {
            const unsigned fc = (unsigned)min(f, F - 1);
            unsigned lql = (unsigned)((5 * q) * F * 16 + i), lqw = (unsigned)(5 * q * F);
            asm volatile("" : "+v"(lql), "+v"(lqw));
#pragma unroll
            for (int ck = 0; ck < KC; ++ck)
#pragma unroll
                for (int e = 0; e < 5; ++e) {
                    // (uniform base per channel slot + ONE per-lane offset: the slot strides are beyond a load's immediate offset, and as
                    // per-lane additions they were two vector instructions per load)
                    xn[ck][e] = (CFB_EXP & 2) ? 0.25f * (float)e : ldg1o((ck == 0 ? abase : bbase) + e * F * 16, 4u * (lql + fc * 16u));
                    wn[ck][e] = (CFB_EXP & 2) ? 1.5f : ldg1o(p.w.ln0_w + (20 * ck + e) * F, 4u * (lqw + fc));
                }
        };
        load_x(4 * wave);
        for (int g = 0; g < F / 32; ++g) {
            // ---- 1x1 convs + gate arithmetic of bins 32 g + 4 w + k
            f32x4 rw0[4], rw1 = {0.f, 0.f, 0.f, 0.f};           // ln2_w r of the four bins: [channel 4 q + r][bin k]; of channel 16 + q
#pragma unroll
            for (int r = 0; r < 4; ++r) rw0[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {                       // (not unrolled: hoisted together the four bins' operands spill)
                const int f = 32 * g + 4 * wave + k;
                // this bin's table rows (L2) are requested first: a wave's loads retire in order, and they are needed behind this bin's MFMAs
                unsigned qv = (unsigned)q;
                asm volatile("" : "+v"(qv));
                f32x4 ta[4];
                float tb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ta[t] = (CFB_EXP & 8192) ? f32x4{0.5f, 0.25f, 1.f, 2.f} : ldg4o(p.w.front_tab + (f * 4 + t) * CH, 16u * qv);
                    tb[t] = (CFB_EXP & 8192) ? 0.5f : ldg1o(p.w.front_tab + (f * 4 + t) * CH + 16, 4u * qv);
                }
                f32x4 agh[2], agl[2], aih[2], ail[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    agh[m] = f32x4{0.f, 0.f, 0.f, 0.f}; agl[m] = agh[m]; ail[m] = agh[m];
                    aih[m] = f32x4{bi[m][0], bi[m][1], bi[m][2], bi[m][3]};
                }
#pragma unroll
                for (int ck = 0; ck < KC; ++ck) {
                    bf16x8 bg[3], wa[2][3];
                    float xw[5];
#pragma unroll
                    for (int e = 0; e < 5; ++e) xw[e] = xn[ck][e] * wn[ck][e];
                    split5(xw, bg);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) wa[m][pl] = lds_frag(WG + ((m * KC + ck) * 3 + pl) * QF_B + lane * 16);
                    mfma_split6(wa[0], bg, agh[0], agl[0]);
                    mfma_split6(wa[1], bg, agh[1], agl[1]);
                }
#pragma unroll
                for (int ck = 0; ck < KC; ++ck) {
                    bf16x8 bx[3], wa[2][3];
                    split5(xn[ck], bx);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) wa[m][pl] = lds_frag(WI + ((m * KC + ck) * 3 + pl) * QF_B + lane * 16);
                    mfma_split6(wa[0], bx, aih[0], ail[0]);
                    mfma_split6(wa[1], bx, aih[1], ail[1]);
                }
                // the next bin's input rows (HBM) behind this bin's MFMAs: their registers were this bin's operands until here (a second
                // set, requested a whole bin ahead, measured no faster and spilled at 40 input channels)
                load_x(k < 3 ? f + 1 : f + 29);               // the next bin of this wave (k = 3: the next group's first)
                unsigned char *slot_w = ring + k * RING_SLOT;
                {   // channels 4 q + r
                    const f32x4 ag = agh[0] + agl[0], ai = aih[0] + ail[0];
                    f32x4 gxw;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // LN0 commuted behind the gate conv: inv0 * (Wg (w0 x)) + (Wg b0 + bias) - mean0 inv0 (Wg w0)
                        const float pre = fmaf(inv0, ag[r], fmaf(-mi0, ta[0][r], ta[1][r]));
                        const float gt = (CFB_EXP & 32) ? pre : gate_sigmoid(pre), xi = ai[r], gx = gt * xi, rr = xi - gx;
                        if (g == 0 && k == 0 && r == 0) { sg.K = gx; sr.K = rr; }      // the tile's first value of this lane: the shifts
                        sg.addk(gx); sr.addk(rr);
                        set_k(rw0[r], k, rr * ta[3][r]);
                        gxw[r] = gx * ta[2][r];
                    }
                    u32x2 p0, p1, p2;
                    split3x4(gxw, p0, p1, p2);
                    unsigned char *d = slot_w + (q * 16 + i) * 8;
                    *reinterpret_cast<u32x2 *>(d) = p0;
                    *reinterpret_cast<u32x2 *>(d + RING_PLANE) = p1;
                    *reinterpret_cast<u32x2 *>(d + 2 * RING_PLANE) = p2;
                }
                {   // channel 16 + q (row 4 q of the second tile)
                    const float pre = fmaf(inv0, agh[1][0] + agl[1][0], fmaf(-mi0, tb[0], tb[1]));
                    const float gt = (CFB_EXP & 32) ? pre : gate_sigmoid(pre), xi = aih[1][0] + ail[1][0], gx = gt * xi, rr = xi - gx;
                    sg.addk(gx); sr.addk(rr);
                    set_k(rw1, k, rr * tb[3]);
                    unsigned short h0, h1, h2;
                    split3x1(gx * tb[2], h0, h1, h2);
                    unsigned short *d = reinterpret_cast<unsigned short *>(slot_w + (4 * 16 + i) * 8) + q;
                    d[0] = h0; d[RING_PLANE / 2] = h1; d[RING_PLANE] = h2;
                }
            }
            if (!(CFB_EXP & 4096)) {
                // ln2_w r of the four bins -> positions 4 (w & 1) .. + 3 of k-group w / 2 of each channel's row: one 8-byte store per plane
                unsigned char *d0 = RQ + (wave >> 1) * RQ_ROW + i * 16 + 8 * (wave & 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    u32x2 p0, p1, p2;
                    split3x4(rw0[r], p0, p1, p2);
                    unsigned char *d = d0 + (4 * q + r) * 4 * RQ_ROW;
                    *reinterpret_cast<u32x2 *>(d) = p0;
                    *reinterpret_cast<u32x2 *>(d + RQ_PLANE) = p1;
                    *reinterpret_cast<u32x2 *>(d + 2 * RQ_PLANE) = p2;
                }
                u32x2 p0, p1, p2;
                split3x4(rw1, p0, p1, p2);
                unsigned char *d = d0 + (16 + q) * 4 * RQ_ROW;
                *reinterpret_cast<u32x2 *>(d) = p0;
                *reinterpret_cast<u32x2 *>(d + RQ_PLANE) = p1;
                *reinterpret_cast<u32x2 *>(d + 2 * RQ_PLANE) = p2;
            }
            CFB_MARK(0);
            lds_barrier();                                       // A: the group's ring slots and B planes are complete
            CFB_MARK(1);
            // ---- (3,1) conv of output bins fo = 32 g + 4 w - 1 + k (wave 7 of the last group: bin 159 too) from the bins fo - 1 .. fo + 1:
            // relative to this wave's slot 0 they are r = k + tap - 2 in [-2, 3] (4: bin 160, zero)
            // the DFT k-step's first two table fragments are requested before the (3,1) conv's y1 stores enter the memory queue
            const float *tq = pq.tbl_q + ((size_t)(5 * mg) * (F / 32) + g) * 3 * QFRAG;
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            bf16x8 tfr[3][3];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) tfr[u][pl] = ldq(tq + ((size_t)u * (F / 32) * 3 + pl) * QFRAG, lane_t);
            {
                const unsigned char *below = wave > 0 ? ring - RING_WAVE + 2 * RING_SLOT : CARRY;      // bins r = -2, -1 (slot pitch RING_SLOT either way)
                const int below_pl = wave > 0 ? RING_PLANE : CARRY_PLANE;
                const int nout = (g == F / 32 - 1 && wave == 7) ? 5 : 4;
                unsigned qv = (unsigned)q, iv = (unsigned)i;
                asm volatile("" : "+v"(qv), "+v"(iv));
#pragma unroll 1
                for (int k = 0; k < nout; ++k) {
                    const int fo = 32 * g + 4 * wave - 1 + k;
                    if (fo < 0) continue;
                    f32x4 a3h[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, a3l[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int ck = 0; ck < 2; ++ck) {
                        bf16x8 b3[3];
                        const unsigned char *src[2];
                        int spl[2];
                        bool ok[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int r = k + c31_tap[ck][h] - 2;              // per lane
                            ok[h] = c31_ok[ck][h] && r < 4;
                            const int rc = r < 4 ? r : 3;
                            src[h] = (rc < 0 ? below + (rc + 2) * RING_SLOT : ring + rc * RING_SLOT) + c31_off[ck][h];
                            spl[h] = rc < 0 ? below_pl : RING_PLANE;
                        }
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            u32x2 lo2 = *reinterpret_cast<const u32x2 *>(src[0] + pl * spl[0]);
                            u32x2 hi2 = *reinterpret_cast<const u32x2 *>(src[1] + pl * spl[1]);
                            if (!ok[0]) lo2 = u32x2{0u, 0u};
                            if (!ok[1]) hi2 = u32x2{0u, 0u};
                            b3[pl] = __builtin_bit_cast(bf16x8, u32x4{lo2[0], lo2[1], hi2[0], hi2[1]});
                        }
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            bf16x8 wa[3];
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) wa[pl] = lds_frag(W31 + ((m * 2 + ck) * 3 + pl) * QF_B + lane * 16);
                            mfma_split6(wa, b3, a3h[m], a3l[m]);
                        }
                    }
                    const unsigned o = qv * (unsigned)(4 * F * 16) + iv + (unsigned)(fo * 16);
                    const f32x4 y0 = a3h[0] + a3l[0];
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg1o(y1_t, 4u * (o + (unsigned)(r * F * 16)), y0[r]);
                    stg1o(y1_t, 4u * ((16u + qv) * (unsigned)(F * 16) + iv + (unsigned)(fo * 16)), a3h[1][0] + a3l[1][0]);
                }
            }
            CFB_MARK(2);
            // ---- the group's DFT k-step: 25 tiles, six products each into ONE accumulator (a lo set would be 100 more registers)
            {
                bf16x8 bq[5][3];
#pragma unroll
                for (int c5 = 0; c5 < 5; ++c5)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bq[c5][pl] = lds_frag(RQ + pl * RQ_PLANE + (((5 * cg + c5) * 4 + q) * 16 + i) * 16);
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    bf16x8 (&cur)[3] = tfr[u % 3];
                    if (u + 2 < 5) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) tfr[(u + 2) % 3][pl] = ldq(tq + ((size_t)(u + 2) * (F / 32) * 3 + pl) * QFRAG, lane_t);
                    }
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[2], bq[c5][0], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[1], bq[c5][1], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[0], bq[c5][2], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[1], bq[c5][0], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[0], bq[c5][1], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[0], bq[c5][0], acc[c5 * 5 + u]);
                }
            }
            CFB_MARK(3);
            lds_barrier();                                       // B: every read of this group's slots and planes is done
            CFB_MARK(1);
            if (wave == 7)                                       // bins 32 g + 30, 32 g + 31 for wave 0's (3,1) conv behind the next barrier A
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    for (int e = lane; e < CARRY_PLANE / 8; e += 64)
                        reinterpret_cast<u32x2 *>(CARRY + pl * CARRY_PLANE)[e] = reinterpret_cast<const u32x2 *>(ring + pl * RING_PLANE + 2 * RING_SLOT)[e];
        }
        // ---- LayerNorm statistics of gx (LN1, handed to cfb_back) and r (LN2, applied below): every wave saw one bin in eight
        float n1, m1, M1, n2, m2, M2;
        sg.finish(n1, m1, M1);
        sr.finish(n2, m2, M2);
        chan_merge_q(n1, m1, M1);
        chan_merge_q(n2, m2, M2);
        if (lane < 16) {
            float *o = RED + (wave * 16 + i) * 6;
            o[0] = n1; o[1] = m1; o[2] = M1; o[3] = n2; o[4] = m2; o[5] = M2;
        }
        lds_barrier();
        float *RES = RED + 8 * 16 * 6;
        if (tid < 16) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float *o = RED + (w * 16 + tid) * 6;
                chan_merge(a0, a1, a2, o[0], o[1], o[2]);
                chan_merge(c0, c1, c2, o[3], o[4], o[5]);
            }
            const float inv1 = 1.0f / (sqrtf(a2 / (a0 - 1.f)) + 1e-6f), sd2 = sqrtf(c2 / (c0 - 1.f)) + 1e-6f;
            p.stats1[((size_t)tile * 16 + tid) * 2] = a1;
            p.stats1[((size_t)tile * 16 + tid) * 2 + 1] = inv1;
            RES[tid * 4] = c1; RES[tid * 4 + 1] = sd2; RES[tid * 4 + 2] = 1.0f / sd2;
        }
        lds_barrier();
        const float mean2 = RES[i * 4], sd2 = RES[i * 4 + 1], inv2 = RES[i * 4 + 2];
        const float bfix = q == 0 ? -mean2 : (q == 1 ? sd2 : 0.f);
        float ssum = 0.f;
        int lane_l = lane, ql = q, il = i;
        asm volatile("" : "+v"(lane_l), "+v"(ql), "+v"(il));
        const float *fix_w = p.w.fwd_fix + ((5 * cg) * 10 + 5 * mg) * 64;
        const unsigned lane_u = (unsigned)lane_l;
        {
            float fx[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) fx[u] = ldg1o(fix_w, 4u * (lane_u + (unsigned)(u * 64)));
#pragma unroll
            for (int jj = 0; jj < 25; ++jj) {
                acc[jj] = mfma16(fx[jj % 5], bfix, acc[jj]);
                if (jj + 5 < 25) fx[jj % 5] = ldg1o(fix_w, 4u * (lane_u + (unsigned)((((jj + 5) / 5) * 10 + (jj + 5) % 5) * 64)));
                acc[jj] *= inv2;
                ssum += (acc[jj][0] + acc[jj][1]) + (acc[jj][2] + acc[jj][3]);
            }
        }
        // ---- statistics of S over (40 channels, 81 bins): 160 stored values + 2 structural zeros per channel, two passes in registers
        float *RD = RES + 64;
        ssum = sum_q(ssum);
        if (lane < 16) RD[wave * 16 + i] = ssum;
        lds_barrier();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += RD[w * 16 + i];
        const float nS = (float)(2 * CH * CF), meanS = tot / nS;
        float dsum = 0.f;
#pragma unroll
        for (int jj = 0; jj < 25; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc[jj][r] - meanS; dsum = fmaf(d, d, dsum); }
        dsum = sum_q(dsum);
        if (lane < 16) RD[128 + wave * 16 + i] = dsum;
        lds_barrier();
        if (tid < 16) {
            float M = 2.f * CH * meanS * meanS;                  // the imaginary parts of bins 0 and 80
#pragma unroll
            for (int w = 0; w < 8; ++w) M += RD[128 + w * 16 + tid];
            p.stats_li[((size_t)tile * 16 + tid) * 2] = meanS;
            p.stats_li[((size_t)tile * 16 + tid) * 2 + 1] = 1.0f / (sqrtf(M / (nS - 1.f)) + 1e-6f);
        }
        // ---- S -> li[40][81]: table rows 0..80 = cos bins of channel c, rows 81..159 = sin bins 1..79 of channel 20 + c; one 16-byte
        // store per lane and tile through the wave's transpose scratch
        {
            float *li_t = p.li + (size_t)tile * (2 * CH * CF * 16);
            CFB_FENCE();
            put_d(ws, ql, il, acc[0]);
#pragma unroll
            for (int jj = 0; jj < 25; ++jj) {
                const int c = 5 * cg + jj / 5, m = 5 * mg + jj % 5;
                CFB_FENCE();
                if (jj + 1 < 25) put_d(ws + ((jj + 1) & 1) * WS_TILE, ql, il, acc[jj + 1]);
                CFB_FENCE();
                const f32x4 v = get_rows(ws + (jj & 1) * WS_TILE, lane_l);
                CFB_FENCE();
                const int row = 16 * m + (lane_l >> 2);
                const int ch = row <= 80 ? c : CH + c, bin = row <= 80 ? row : row - 80;
                stg4o(li_t, 4u * (unsigned)((ch * CF + bin) * 16 + 4 * (lane_l & 3)), v);
            }
            for (int e = tid; e < CH * 2 * 16; e += NTH)
                li_t[((CH + (e >> 5)) * CF + (((e >> 4) & 1) ? 80 : 0)) * 16 + (e & 15)] = 0.f;
        }
        lds_barrier();                              // the scratch aliases the planes
        CFB_MARK(4);
    }
    CFB_FLUSH(16);
}

// ---- second half on split products.  CepsUnit's Linear 40 -> 40 (K = 40 as two k-steps of 5 channels per lane quarter) and the pinv
// inverse DFT.  The inverse contracts over 81 real + 79 imaginary parts = 160 values = FIVE bf16 k-steps with no padding once the
// k order is chosen freely: k-step j holds [re of ceps bins 16 j .. 16 j + 15 | im of the same bins], and the one slot that is
// structurally zero (im of bin 0) carries re of bin 80.  Per k-step, wave w runs the Linear + complex product of bins 16 j + 2 w,
// 16 j + 2 w + 1 (wave 7 also bin 80, in step 0, where there is no DFT work yet) -> OB[j & 1] (4-byte stores: two consecutive bins of
// a (channel, part, frame) row), and accumulates the DFT k-step j - 1 from OB[(j - 1) & 1]; one barrier per step.
constexpr int back_split_lds() { return 2 * 3 * RQ_PLANE + 3 * 2 * 3 * QF_B; }
static_assert(back_split_lds() <= 160 * 1024, "LDS budget of cfb_back_split");
static_assert((8 * WS_FLOATS + 256 + 64) * 4 <= 2 * 3 * RQ_PLANE, "the split back kernel's scratch aliases the B planes");

struct BackQArgs {
    BackArgs b;
    const float *tbl_q;           // pinv inverse table as split A fragments [10 row tiles][5 k-steps][3 planes][QFRAG], k order as above
};

__global__ __launch_bounds__(NTH) void cfb_back_split_kernel(BackQArgs pq) {
    const BackArgs &p = pq.b;
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
    unsigned char *OBQ = qlds;                                  // [2][3 planes][20 ch][4 k-groups: re lo, re hi, im lo, im hi][16 frames][8 bins]
    unsigned char *WLQ = OBQ + 2 * 3 * RQ_PLANE;                // [3 row tiles][2 chunks][3 planes] fragments of the Linear
    float *WS = reinterpret_cast<float *>(OBQ);                 // prologue / epilogue only: [8 waves][2 tiles] transpose scratch
    float *RED = WS + 8 * WS_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, i = lane & 15;
    // Linear 40 -> 40 (rows permuted on the host so that a lane's four D rows are (re c, re c+1, im c, im c+1), c = 8 mt + 2 q):
    // slot 8 q + el of chunk ck <-> input channel 20 ck + 5 q + el (el < 5)
    for (int e = tid; e < 3 * 2 * 64 * 8; e += NTH) {
        const int el = e & 7, l = (e >> 3) & 63, ck = (e >> 9) & 1, m = e >> 10;
        const float w = el < 5 ? p.w.lin_w[(m * 16 + (l & 15)) * 40 + 20 * ck + 5 * (l >> 4) + el] : 0.f;
        unsigned short h0, h1, h2;
        split3x1(w, h0, h1, h2);
        unsigned short *d = reinterpret_cast<unsigned short *>(WLQ + (m * 2 + ck) * 3 * QF_B) + l * 8 + el;
        d[0] = h0; d[QF_B / 2] = h1; d[QF_B] = h2;
    }
    float bl[3][4];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) bl[m][r] = p.w.lin_b[m * 16 + 4 * q + r];
    const int cg = wave >> 1, mg = wave & 1;           // DFT tiles: channels 5 cg .. + 4, row tiles 5 mg .. + 4; tile jj = (jj / 5, jj % 5)
    float *ws = WS + wave * WS_FLOATS;
    __syncthreads();

    f32x4 acc[25];
    const unsigned rowoff = (unsigned)((lane >> 2) * 16 + 4 * (lane & 3) + ((5 * cg) * F + 16 * (5 * mg)) * 16);
    auto y1_load = [&](int tile) {                  // the accumulators START as y1 (row layout; transposed below)
        const float *y1_t = p.y1 + (size_t)tile * (CH * F * 16);
#pragma unroll
        for (int jj = 0; jj < 25; ++jj) acc[jj] = ldg4o(y1_t, 4u * (rowoff + (unsigned)(((jj / 5) * F + 16 * (jj % 5)) * 16)));
    };
    if ((int)blockIdx.x < p.tiles) y1_load(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const float *hf_t = p.hf + (size_t)tile * (2 * CH * CF * 16);
        const float *li_t = p.li + (size_t)tile * (2 * CH * CF * 16);
        float hn[2][5], sn[12];                     // operands of the NEXT bin of this wave
        auto load_h = [&](int bin) {
            unsigned lql = (unsigned)((5 * q) * CF * 16 + i + bin * 16);
            asm volatile("" : "+v"(lql));
#pragma unroll
            for (int ck = 0; ck < 2; ++ck)
#pragma unroll
                for (int e = 0; e < 5; ++e) hn[ck][e] = ldg1o(hf_t + (20 * ck + e) * CF * 16, 4u * lql);
        };
        auto load_s = [&](int bin) {
            unsigned lil = (unsigned)(i + bin * 16);
            asm volatile("" : "+v"(lil));
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int ca = min(8 * m + 2 * q, CH - 2);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    sn[4 * m + h2] = ldg1o(li_t, 4u * (lil + (unsigned)((ca + h2) * CF * 16)));
                    sn[4 * m + 2 + h2] = ldg1o(li_t, 4u * (lil + (unsigned)((CH + ca + h2) * CF * 16)));
                }
            }
        };
        load_h(2 * wave);
        load_s(2 * wave);
        // ---- acc = inv1 * (y1 - mean1 * CW) + CB: y1 arrived as one 16-byte load per lane and tile (requested during the previous
        // tile's epilogue), is turned into the MFMA D layout through the wave's LDS scratch; the (CW, CB) term is one k-step
        const float mean1 = p.stats1[((size_t)tile * 16 + i) * 2], inv1 = p.stats1[((size_t)tile * 16 + i) * 2 + 1];
        const float bfix = q == 0 ? -mean1 * inv1 : (q == 1 ? 1.f : 0.f);
        int ql = q, il = i, lane_l = lane;
        asm volatile("" : "+v"(ql), "+v"(il), "+v"(lane_l));
        const float *fix_w = p.w.out_fix + ((5 * cg) * 10 + 5 * mg) * 64;
        const unsigned lane_u = (unsigned)lane_l;
        {
            float fx[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) fx[u] = ldg1o(fix_w, 4u * (lane_u + (unsigned)(u * 64)));
            CFB_FENCE();
            put_rows(ws, lane_l, acc[0]);
#pragma unroll
            for (int jj = 0; jj < 25; ++jj) {
                CFB_FENCE();
                if (jj + 1 < 25) put_rows(ws + ((jj + 1) & 1) * WS_TILE, lane_l, acc[jj + 1]);
                CFB_FENCE();
                f32x4 d = get_d(ws + (jj & 1) * WS_TILE, ql, il);
                CFB_FENCE();
                d *= inv1;
                acc[jj] = mfma16(fx[jj % 5], bfix, d);
                if (jj + 5 < 25) fx[jj % 5] = ldg1o(fix_w, 4u * (lane_u + (unsigned)((((jj + 5) / 5) * 10 + (jj + 5) % 5) * 64)));
            }
        }
        lds_barrier();                              // the scratch aliases the planes
        for (int g = 0; g <= F / 32; ++g) {
            if (g < F / 32) {
                // ---- Linear + complex product of bins 16 g + 2 w, + 1 (g = 0, wave 7: bin 80 as a third)
                const int nb = (g == 0 && wave == 7) ? 3 : 2;
                float o_re[3][2][2], o_im[3][2][2];             // [m][h2][bin of the pair]
                unsigned char *obw = OBQ + (g & 1) * 3 * RQ_PLANE;
#pragma unroll 1
                for (int b2 = 0; b2 < nb; ++b2) {
                    const int bin = b2 < 2 ? 16 * g + 2 * wave + b2 : CF - 1;
                    // the bin after this one in the wave's sequence (the last one's successor is never used: any valid bin)
                    const int nxt = b2 + 1 < nb ? (b2 == 0 ? bin + 1 : CF - 1) : min(16 * (g + 1) + 2 * wave, CF - 1);
                    f32x4 Ph[3], Pl[3];
#pragma unroll
                    for (int m = 0; m < 3; ++m) { Ph[m] = f32x4{bl[m][0], bl[m][1], bl[m][2], bl[m][3]}; Pl[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int ck = 0; ck < 2; ++ck) {
                        bf16x8 bh[3];
                        split5(hn[ck], bh);
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            bf16x8 wa[3];
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) wa[pl] = lds_frag(WLQ + ((m * 2 + ck) * 3 + pl) * QF_B + lane * 16);
                            mfma_split6(wa, bh, Ph[m], Pl[m]);
                        }
                    }
                    load_h(nxt);
                    float pr_[3][2][2];                         // [m][h2][re | im]
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const f32x4 P = Ph[m] + Pl[m];
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const float sre = sn[4 * m + h2], sim = sn[4 * m + 2 + h2], pr = P[h2], pi = P[2 + h2];
                            pr_[m][h2][0] = pr * sre - pi * sim;
                            pr_[m][h2][1] = pr * sim + pi * sre;
                        }
                    }
                    load_s(nxt);
                    if (b2 < 2) {
#pragma unroll
                        for (int m = 0; m < 3; ++m)
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                if (b2 == 0) { o_re[m][h2][0] = pr_[m][h2][0]; o_im[m][h2][0] = pr_[m][h2][1]; }
                                else { o_re[m][h2][1] = pr_[m][h2][0]; o_im[m][h2][1] = pr_[m][h2][1]; }
                            }
                    } else {
                        // bin 80: its real part rides in the slot of im (bin 0) = position 0 of k-group 2 of step 0
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            const int ca = 8 * m + 2 * q;
                            if (ca < CH) {
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) {
                                    unsigned short t0, t1, t2;
                                    split3x1(pr_[m][h2][0], t0, t1, t2);
                                    unsigned short *d = reinterpret_cast<unsigned short *>(obw + (((ca + h2) * 4 + 2) * 16 + i) * 16);
                                    d[0] = t0; d[RQ_PLANE / 2] = t1; d[RQ_PLANE] = t2;
                                }
                            }
                        }
                    }
                }
                // the pair's two bins -> positions 2 (w & 3), + 1 of k-group w / 4 (re) and 2 + w / 4 (im): one 4-byte store per plane
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int ca = 8 * m + 2 * q;
                    if (ca < CH) {
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            unsigned p0, p1, p2;
                            unsigned char *d = obw + (((ca + h2) * 4 + (wave >> 2)) * 16 + i) * 16 + 4 * (wave & 3);
                            split3x2(o_re[m][h2][0], o_re[m][h2][1], p0, p1, p2);
                            *reinterpret_cast<unsigned *>(d) = p0;
                            *reinterpret_cast<unsigned *>(d + RQ_PLANE) = p1;
                            *reinterpret_cast<unsigned *>(d + 2 * RQ_PLANE) = p2;
                            split3x2(o_im[m][h2][0], o_im[m][h2][1], p0, p1, p2);
                            d += 2 * RQ_ROW;
                            if (g == 0 && wave == 0) {          // im of bin 0 does not exist: its slot is bin 80's (wave 7 writes it)
                                reinterpret_cast<unsigned short *>(d)[1] = (unsigned short)(p0 >> 16);
                                reinterpret_cast<unsigned short *>(d + RQ_PLANE)[1] = (unsigned short)(p1 >> 16);
                                reinterpret_cast<unsigned short *>(d + 2 * RQ_PLANE)[1] = (unsigned short)(p2 >> 16);
                            } else {
                                *reinterpret_cast<unsigned *>(d) = p0;
                                *reinterpret_cast<unsigned *>(d + RQ_PLANE) = p1;
                                *reinterpret_cast<unsigned *>(d + 2 * RQ_PLANE) = p2;
                            }
                        }
                    }
                }
            }
            if (g > 0) {
                // ---- inverse DFT k-step g - 1: 25 tiles, six products each
                const unsigned char *obr = OBQ + ((g - 1) & 1) * 3 * RQ_PLANE;
                const float *tq = pq.tbl_q + ((size_t)(5 * mg) * (F / 32) + (g - 1)) * 3 * QFRAG;
                int lane_t = lane;
                asm volatile("" : "+v"(lane_t));
                bf16x8 tfr[3][3];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) tfr[u][pl] = ldq(tq + ((size_t)u * (F / 32) * 3 + pl) * QFRAG, lane_t);
                bf16x8 bq[5][3];
#pragma unroll
                for (int c5 = 0; c5 < 5; ++c5)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bq[c5][pl] = lds_frag(obr + pl * RQ_PLANE + (((5 * cg + c5) * 4 + q) * 16 + i) * 16);
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    bf16x8 (&cur)[3] = tfr[u % 3];
                    if (u + 2 < 5) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) tfr[(u + 2) % 3][pl] = ldq(tq + ((size_t)(u + 2) * (F / 32) * 3 + pl) * QFRAG, lane_t);
                    }
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[2], bq[c5][0], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[1], bq[c5][1], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[0], bq[c5][2], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[1], bq[c5][0], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[0], bq[c5][1], acc[c5 * 5 + u]);
#pragma unroll
                    for (int c5 = 0; c5 < 5; ++c5) acc[c5 * 5 + u] = mfma_bf16(cur[0], bq[c5][0], acc[c5 * 5 + u]);
                }
            }
            lds_barrier();
        }
        // ---- out: one 16-byte store per lane and tile; as a tile's registers leave, the NEXT tile's y1 is requested into them, so that
        // its latency passes under this epilogue's reductions; (count, mean, M2) of the output per frame for the next LayerNorm
        float ssum = 0.f, dsum = 0.f;
#pragma unroll
        for (int jj = 0; jj < 25; ++jj) ssum += (acc[jj][0] + acc[jj][1]) + (acc[jj][2] + acc[jj][3]);
        ssum = sum_q(ssum);
        if (lane < 16) RED[wave * 16 + i] = ssum;
        lds_barrier();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += RED[w * 16 + i];
        const float nO = (float)(CH * F), meanO = tot / nO;
#pragma unroll
        for (int jj = 0; jj < 25; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc[jj][r] - meanO; dsum = fmaf(d, d, dsum); }
        dsum = sum_q(dsum);
        if (lane < 16) RED[128 + wave * 16 + i] = dsum;
        float *out_t = p.out.ptr + ((size_t)tile * p.out.c_total + p.out.c_off) * (F * 16);
        const int next = tile + gridDim.x;
        const float *y1_n = p.y1 + (size_t)(next < p.tiles ? next : tile) * (CH * F * 16);
        unsigned ro = rowoff;
        asm volatile("" : "+v"(ql), "+v"(il), "+v"(lane_l), "+v"(ro));
        CFB_FENCE();
        put_d(ws, ql, il, acc[0]);
#pragma unroll
        for (int jj = 0; jj < 25; ++jj) {
            const unsigned off = ro + (unsigned)(((jj / 5) * F + 16 * (jj % 5)) * 16);
            CFB_FENCE();
            if (jj + 1 < 25) put_d(ws + ((jj + 1) & 1) * WS_TILE, ql, il, acc[jj + 1]);
            CFB_FENCE();
            const f32x4 v = get_rows(ws + (jj & 1) * WS_TILE, lane_l);
            CFB_FENCE();
            stg4o(out_t, 4u * off, v);
            if (jj >= 1)        // tile jj - 1's registers are free now: the next tile's y1 lands in them during the reductions below
                acc[jj - 1] = ldg4o(y1_n, 4u * (ro + (unsigned)((((jj - 1) / 5) * F + 16 * ((jj - 1) % 5)) * 16)));
        }
        acc[24] = ldg4o(y1_n, 4u * (ro + (unsigned)((4 * F + 16 * 4) * 16)));
        lds_barrier();
        if (p.part && tid < 16) {
            float M = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) M += RED[128 + w * 16 + tid];
            float *o = p.part + (((size_t)tile * VADX_DFSMN_STAT_PARTS) * 16 + tid) * 4;
            o[0] = nO; o[1] = meanO; o[2] = M; o[3] = 0.f;
            float *z = o + 16 * 4;
            z[0] = z[1] = z[2] = z[3] = 0.f;
        }
    }
}

static int cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int CIN>
constexpr size_t front_lds_bytes() { return (size_t)(TBLF_FLOATS + 2 * CH * RP8 + CH * ring_pitch<CIN>() + 2 * 15 * 64 + 4 * (CIN / 4) * 64) * sizeof(float); }
static_assert(front_lds_bytes<20>() <= 160 * 1024 && front_lds_bytes<40>() <= 160 * 1024, "LDS budget of cfb_front");
static_assert(8 * WS_FLOATS <= 2 * CH * RP8 && 8 * 16 * 6 + 64 + 256 <= CH * NSLOT * 16, "the front kernel's epilogue scratch aliases R and the ring");
constexpr size_t back_lds_bytes() { return (size_t)(TBLI_FLOATS + 2 * CH * OP8 + 3 * 10 * 64) * sizeof(float); }
static_assert(8 * WS_FLOATS + 256 <= 2 * CH * OP8, "the back kernel's scratch aliases OB");

}  // namespace dfsmn_cfb
}  // namespace vadx

using namespace vadx::dfsmn_cfb;

static bool weights_ok(const vadx_dfsmn_cfb_weights *w) {
    return w && w->ln0_w && w->gate_w && w->in_w && w->in_b && w->front_tab && w->conv_w && w->fwd_tbl && w->fwd_fix && w->lin_w &&
           w->lin_b && w->inv_tbl && w->out_fix;
}

extern "C" int vadx_dfsmn_cfb_front(const vadx_dfsmn_cfb_weights *w, const vadx_ft_view *a, const vadx_ft_view *b, const float *stats0,
                                    float *y1, float *stats1, float *li, float *stats_li, int tiles, void *stream) {
    VADX_REQUIRE(weights_ok(w) && a && a->ptr && stats0 && y1 && stats1 && li && stats_li && tiles > 0, "vadx_dfsmn_cfb_front: bad argument");
    FrontArgs p;
    p.a = View{a->ptr, a->c_total, a->c_off, a->c};
    p.b = b && b->ptr ? View{b->ptr, b->c_total, b->c_off, b->c} : View{nullptr, 0, 0, 0};
    p.stats0 = stats0; p.w = *w; p.y1 = y1; p.stats1 = stats1; p.li = li; p.stats_li = stats_li; p.tiles = tiles;
    const int cin = p.a.c + p.b.c;
    VADX_REQUIRE(p.a.c % 4 == 0, "vadx_dfsmn_cfb_front: the first view must hold a multiple of 4 channels (a k-step does not straddle the views)");
    const unsigned grid = (unsigned)(tiles < cu_count() ? tiles : cu_count());
    hipStream_t st = static_cast<hipStream_t>(stream);
    VADX_REQUIRE(w->front_arithmetic == VADX_ARITH_AUTO || w->front_arithmetic == VADX_ARITH_F32 || w->front_arithmetic == VADX_ARITH_BF16X3,
                 "vadx_dfsmn_cfb_front: front_arithmetic=%d (this kernel has VADX_ARITH_F32 and VADX_ARITH_BF16X3)", w->front_arithmetic);
    if (w->front_arithmetic != VADX_ARITH_F32 && w->fwd_tbl_q && (cin == 20 || cin == 40) && p.a.c == 20) {
        // bf16 x 3 split products (front_arithmetic = VADX_ARITH_F32 selects the f32-MFMA kernels below)
        FrontQArgs pq{p, w->fwd_tbl_q};
        if (cin == 20) {
            VADX_DYN_LDS(cfb_front_split_kernel<20>, front_split_lds<20>());
            hipLaunchKernelGGL(cfb_front_split_kernel<20>, dim3(grid), dim3(NTH), front_split_lds<20>(), st, pq);
        } else {
            VADX_DYN_LDS(cfb_front_split_kernel<40>, front_split_lds<40>());
            hipLaunchKernelGGL(cfb_front_split_kernel<40>, dim3(grid), dim3(NTH), front_split_lds<40>(), st, pq);
        }
        VADX_HIP_TRY(hipGetLastError());
        return VADX_OK;
    }
    if (cin == 20) {
        VADX_DYN_LDS(cfb_front_kernel<20>, front_lds_bytes<20>());
        hipLaunchKernelGGL(cfb_front_kernel<20>, dim3(grid), dim3(NTH), front_lds_bytes<20>(), st, p);
    } else if (cin == 40) {
        VADX_DYN_LDS(cfb_front_kernel<40>, front_lds_bytes<40>());
        hipLaunchKernelGGL(cfb_front_kernel<40>, dim3(grid), dim3(NTH), front_lds_bytes<40>(), st, p);
    } else {
        vadx::set_error("vadx_dfsmn_cfb_front: the input must have 20 or 40 channels (got %d)", cin);
        return VADX_EINVAL;
    }
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

extern "C" int vadx_dfsmn_cfb_back(const vadx_dfsmn_cfb_weights *w, const float *hf, const float *li, const float *y1, const float *stats1,
                                   const vadx_ft_view *out, float *part, int tiles, void *stream) {
    VADX_REQUIRE(weights_ok(w) && hf && li && y1 && stats1 && out && out->ptr && out->c == 20 && tiles > 0, "vadx_dfsmn_cfb_back: bad argument");
    BackArgs p;
    p.hf = hf; p.li = li; p.y1 = y1; p.stats1 = stats1; p.w = *w;
    p.out = ViewW{const_cast<float *>(out->ptr), out->c_total, out->c_off, out->c};
    p.part = part; p.tiles = tiles;
    const unsigned grid = (unsigned)(tiles < cu_count() ? tiles : cu_count());
    // The split-product form of this half is opt-in (back_arithmetic = VADX_ARITH_BF16X3): measured no faster than the f32-MFMA kernel
    // (0.975 against 0.943 ms per 3584 tiles) -- this half moves 824 KB per tile in 64-byte rows and waits on them, not on the matrix pipe.
    VADX_REQUIRE(w->back_arithmetic == VADX_ARITH_AUTO || w->back_arithmetic == VADX_ARITH_F32 || w->back_arithmetic == VADX_ARITH_BF16X3,
                 "vadx_dfsmn_cfb_back: back_arithmetic=%d (this kernel has VADX_ARITH_F32 and VADX_ARITH_BF16X3)", w->back_arithmetic);
    if (w->back_arithmetic == VADX_ARITH_BF16X3 && w->inv_tbl_q) {
        BackQArgs pq{p, w->inv_tbl_q};
        VADX_DYN_LDS(cfb_back_split_kernel, back_split_lds());
        hipLaunchKernelGGL(cfb_back_split_kernel, dim3(grid), dim3(NTH), back_split_lds(), static_cast<hipStream_t>(stream), pq);
        VADX_HIP_TRY(hipGetLastError());
        return VADX_OK;
    }
    VADX_DYN_LDS(cfb_back_kernel, back_lds_bytes());
    hipLaunchKernelGGL(cfb_back_kernel, dim3(grid), dim3(NTH), back_lds_bytes(), static_cast<hipStream_t>(stream), p);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
