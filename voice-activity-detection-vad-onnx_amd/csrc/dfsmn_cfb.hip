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

// CFB_EXP: development-only what-if switches (bit mask; results are wrong when set; tools/exp_cfb.py): 1 no workgroup barriers in the
// chunk loops, 2 no global loads of the chunk rows, 4 no gate / input conv MFMAs, 8 no (3,1) conv, 16 no DFT accumulation,
// 32 no gate arithmetic / statistics, 64 no y1 stores, 128 no Linear + complex product (back), 256 no epilogue stores
#ifndef CFB_EXP
#define CFB_EXP 0
#endif
#define CFB_SYNC() do { if (!(CFB_EXP & 1)) __syncthreads(); } while (0)

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

constexpr int F = 160, CH = 20, CF = 81, NTH = 512, TPW = 25;       // bins, channels, ceps bins, threads, DFT tiles per wave
constexpr int KSF = 40, KSI = 41;                                   // k-steps of the forward / inverse table
constexpr int XP = 80;                                              // LDS pitch per channel of a 4-bin chunk (4 * 16 + 16: the
                                                                    // four k-quarters of an MFMA B read land 16 banks apart)
constexpr int GP = 16 * 16 + 16;                                    // pitch per channel of the 16-slot gx ring
constexpr int RP = 64, OP = 128;
constexpr int TBLF_FLOATS = 10 * KSF * 64, TBLI_FLOATS = 10 * KSI * 64;
constexpr int RED_FLOATS = 8 * 16 * 6 + 16 * 8;

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

struct FrontArgs {
    View a, b;
    const float *stats0;
    vadx_dfsmn_cfb_weights w;
    float *y1, *stats1, *li, *stats_li;
    int tiles;
};

template <int CIN>
__global__ __launch_bounds__(NTH) void cfb_front_kernel(FrontArgs p) {
    constexpr int KS = CIN / 4, NIT = (CIN * 16 + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *TBL = lds;                              // [10 row tiles][40 k-steps][64 lanes]
    float *X = TBL + TBLF_FLOATS;                  // [CIN][4 bins][16] pitch XP
    float *WB = X + CIN * XP;                      // [CIN][4 bins] (LN0 weight, bias)
    float *R = WB + CIN * 8;                       // [20][4 bins][16]: w2 * r of the chunk
    float *GXW = R + CH * RP;                      // [20][16 slots][16] pitch GP: w1 * gx, ring over bins
    float *RED = GXW + CH * GP;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, i = lane & 15;
    const int fbin = wave & 3, mt = wave >> 2;
    const bool rows_ok = mt == 0 || q == 0;        // this lane's four output channels 16 mt + 4 q + r exist (20 channels)

    for (int e = tid; e < TBLF_FLOATS / 4; e += NTH) reinterpret_cast<f32x4 *>(TBL)[e] = ldg4(p.w.fwd_tbl + 4 * e);
    float wgf[KS], wif[KS], w31f[15], bg[4], bi[4];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        wgf[s] = p.w.gate_w[(mt * 16 + i) * CIN + 4 * s + q];
        wif[s] = p.w.in_w[(mt * 16 + i) * CIN + 4 * s + q];
    }
#pragma unroll
    for (int s = 0; s < 15; ++s) w31f[s] = p.w.conv_w[(mt * 16 + i) * 60 + 4 * s + q];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bg[r] = p.w.gate_b[mt * 16 + 4 * q + r]; bi[r] = p.w.in_b[mt * 16 + 4 * q + r]; }
    // DFT tile ownership: wave = (channel group cg of five channels, row-tile group mg of five tiles); tile jj = (cc, mm) = (jj / 5,
    // jj % 5) -> channel 5 cg + cc, row tile 5 mg + mm: every LDS address below is one base register plus a compile-time offset, and
    // a k-step needs five table fragments and five data fragments for its 25 MFMAs
    const int cg = wave >> 1, mg = wave & 1;
    const float *tbl_w = TBL + (5 * mg * KSF) * 64 + lane;
    const float *r_w = R + (5 * cg) * RP + q * 16 + i;
    const int ac = p.a.c;
    const int co0 = min(mt * 16 + 4 * q, CH - 4);  // clamped channel base for the per-lane LayerNorm weight loads

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        f32x4 acc[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float mean0 = p.stats0[((size_t)tile * 16 + i) * 2], inv0 = p.stats0[((size_t)tile * 16 + i) * 2 + 1];
        const float *abase = p.a.ptr + ((size_t)tile * p.a.c_total + p.a.c_off) * F * 16;
        const float *bbase = p.b.ptr ? p.b.ptr + ((size_t)tile * p.b.c_total + p.b.c_off) * F * 16 : abase;
        f32x4 pre[NIT];
        float pw0 = 1.f, pb0 = 0.f, w1n[4], w2n[4], w1c[4], w2c[4];
        auto request = [&](int j) {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int e = min(tid + NTH * u, CIN * 16 - 1), row = e >> 2, tq = e & 3, ci = row >> 2, fl = row & 3;
                const int f = 4 * j + fl;
                const float *src = ci < ac ? abase + (size_t)(ci * F + f) * 16 : bbase + (size_t)((ci - ac) * F + f) * 16;
                pre[u] = (CFB_EXP & 2) ? f32x4{0.1f, 0.2f, 0.3f, 0.4f} : ldg4(src + 4 * tq);
            }
            const int e = min(tid, CIN * 4 - 1), ci = e >> 2, f = 4 * j + (e & 3);
            pw0 = ldg1(p.w.ln0_w + ci * F + f);
            pb0 = ldg1(p.w.ln0_b + ci * F + f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                w1n[r] = ldg1(p.w.ln1_w + (co0 + r) * F + 4 * j + fbin);
                w2n[r] = ldg1(p.w.ln2_w + (co0 + r) * F + 4 * j + fbin);
            }
        };
        auto park = [&]() {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int e = tid + NTH * u, row = e >> 2, tq = e & 3, ci = row >> 2, fl = row & 3;
                if (e < CIN * 16) *reinterpret_cast<f32x4 *>(X + ci * XP + fl * 16 + 4 * tq) = pre[u];
            }
            if (tid < CIN * 4) { WB[2 * tid] = pw0; WB[2 * tid + 1] = pb0; }
#pragma unroll
            for (int r = 0; r < 4; ++r) { w1c[r] = w1n[r]; w2c[r] = w2n[r]; }
        };
        for (int e = tid; e < CH * 16; e += NTH) GXW[(e >> 4) * GP + 15 * 16 + (e & 15)] = 0.f;      // bin -1 (zero padding of the (3,1) conv)
        request(0);
        park();
        request(1);
        Acc1 sg, sr;
        sg.init(); sr.init();
        for (int j = 0; j <= KSF; ++j) {
            CFB_SYNC();                 // chunk j is parked; nobody reads R / the ring slots written below any more
            if (j < KSF) {
                // ---- gate / input 1x1 convs of bin 4 j + fbin, output-channel tile mt
                const int f = 4 * j + fbin;
                f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ai = {0.f, 0.f, 0.f, 0.f};
                float xv[KS], lv[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    xv[s] = X[(4 * s + q) * XP + fbin * 16 + i];
                    const float2 wb = *reinterpret_cast<const float2 *>(WB + 2 * ((4 * s + q) * 4 + fbin));
                    const float sc = inv0 * wb.x;
                    lv[s] = fmaf(xv[s], sc, wb.y - mean0 * sc);
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (CFB_EXP & 4) { ag[0] += lv[s]; ai[0] += xv[s]; continue; }
                    ag = mfma16(wgf[s], lv[s], ag);
                    ai = mfma16(wif[s], xv[s], ai);
                }
                if ((CFB_EXP & 32) && ag[0] != 123.f) {
                    if (rows_ok)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { GXW[(mt * 16 + 4 * q + r) * GP + (f & 15) * 16 + i] = ag[r]; R[(mt * 16 + 4 * q + r) * RP + fbin * 16 + i] = ai[r]; }
                } else
                if (rows_ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = mt * 16 + 4 * q + r;
                        const float g = gate_sigmoid(ag[r] + bg[r]), xi = ai[r] + bi[r], gx = g * xi, rr = xi - gx;
                        sg.add(gx);
                        sr.add(rr);
                        GXW[co * GP + (f & 15) * 16 + i] = gx * w1c[r];
                        R[co * RP + fbin * 16 + i] = rr * w2c[r];
                    }
                }
            } else {
                for (int e = tid; e < CH * 16; e += NTH) GXW[(e >> 4) * GP + (F & 15) * 16 + (e & 15)] = 0.f;      // bin 160 (zero padding)
            }
            CFB_SYNC();
            if (j + 1 < KSF) {
                park();                                   // chunk j + 1 (requested a phase ago) -> X
                if (j + 2 < KSF) request(j + 2);
            }
            // ---- (3,1) conv of output bin 4 j - 1 + fbin on the ring (bins fo - 1 .. fo + 1), raw: LN1 is applied by cfb_back
            const int fo = 4 * j - 1 + fbin;
            if (fo >= 0 && fo < F && !(CFB_EXP & 8)) {
                f32x4 a3 = {0.f, 0.f, 0.f, 0.f};
                float bv[15];
#pragma unroll
                for (int s = 0; s < 15; ++s) bv[s] = GXW[(4 * (s % 5) + q) * GP + ((fo + s / 5 - 1) & 15) * 16 + i];
#pragma unroll
                for (int s = 0; s < 15; ++s) a3 = mfma16(w31f[s], bv[s], a3);
                if (rows_ok && (!(CFB_EXP & 64) || a3[0] == 123.f)) {
                    float *dst = p.y1 + ft_idx(tile, CH, mt * 16 + 4 * q, F, fo) + i;
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg1(dst + (size_t)r * F * 16, a3[r]);
                }
            }
            // ---- DFT k-step j: 25 tiles (channel c, row tile m) per wave
            if (j < KSF && !(CFB_EXP & 16)) {
                float ta[5], rb[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) { ta[u] = tbl_w[(u * KSF + j) * 64]; rb[u] = r_w[u * RP]; }
#pragma unroll
                for (int jj = 0; jj < TPW; ++jj) acc[jj] = mfma16(ta[jj % 5], rb[jj / 5], acc[jj]);
            }
        }
        // ---- LayerNorm statistics of gx (LN1, handed to cfb_back) and r (LN2, applied here)
        float n1, m1, M1, n2, m2, M2;
        sg.finish(n1, m1, M1);
        sr.finish(n2, m2, M2);
        chan_merge_q(n1, m1, M1);
        chan_merge_q(n2, m2, M2);
        __syncthreads();
        if (lane < 16) {
            float *o = RED + (wave * 16 + i) * 6;
            o[0] = n1; o[1] = m1; o[2] = M1; o[3] = n2; o[4] = m2; o[5] = M2;
        }
        __syncthreads();
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
            RES[tid * 8] = b1; RES[tid * 8 + 1] = sd2; RES[tid * 8 + 2] = 1.0f / sd2;
        }
        __syncthreads();
        const float mean2 = RES[i * 8], sd2 = RES[i * 8 + 1], inv2 = RES[i * 8 + 2];
        const float bfix = q == 0 ? -mean2 : (q == 1 ? sd2 : 0.f);
        float ssum = 0.f;
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const float *fix_w = p.w.fwd_fix + ((5 * cg) * 10 + 5 * mg) * 64;
#pragma unroll
        for (int jj = 0; jj < TPW; ++jj) {
            acc[jj] = mfma16(ldg1(fix_w + ((jj / 5) * 10 + jj % 5) * 64 + lane_l), bfix, acc[jj]);
            acc[jj] *= inv2;
            ssum += (acc[jj][0] + acc[jj][1]) + (acc[jj][2] + acc[jj][3]);
        }
        // ---- statistics of S over (40 channels, 81 bins): 160 stored values + 2 structural zeros per channel, two passes in registers
        ssum = sum_q(ssum);
        if (lane < 16) RED[wave * 16 + i] = ssum;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += RED[w * 16 + i];
        const float nS = (float)(2 * CH * CF), meanS = tot / nS;
        float dsum = 0.f;
#pragma unroll
        for (int jj = 0; jj < TPW; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = acc[jj][r] - meanS; dsum = fmaf(d, d, dsum); }
        dsum = sum_q(dsum);
        __syncthreads();
        if (lane < 16) RED[wave * 16 + i] = dsum;
        __syncthreads();
        if (tid < 16) {
            float M = 2.f * CH * meanS * meanS;                  // the imaginary parts of bins 0 and 80
#pragma unroll
            for (int w = 0; w < 8; ++w) M += RED[w * 16 + tid];
            p.stats_li[((size_t)tile * 16 + tid) * 2] = meanS;
            p.stats_li[((size_t)tile * 16 + tid) * 2 + 1] = 1.0f / (sqrtf(M / (nS - 1.f)) + 1e-6f);
        }
        // ---- S -> li[40][81]: table rows 0..80 = cos bins of channel c, rows 81..159 = sin bins 1..79 of channel 20 + c
        // (32-bit offsets from the tile's base, formed from laundered lane indices: left visible, the compiler hoists the hundred
        // loop-invariant store addresses out of the tile loop and spills them)
        {
            int ql = q, il = i;
            asm volatile("" : "+v"(ql), "+v"(il));
            float *li_t = p.li + (size_t)tile * (2 * CH * CF * 16);
#pragma unroll
            for (int jj = 0; jj < TPW; ++jj) {
                const int c = 5 * cg + jj / 5, m = 5 * mg + jj % 5;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * m + 4 * ql + r;
                    const int ch = row <= 80 ? c : CH + c, bin = row <= 80 ? row : row - 80;
                    stg1(li_t + (ch * CF + bin) * 16 + il, acc[jj][r]);
                }
            }
        }
        for (int e = tid; e < CH * 2 * 16; e += NTH)
            p.li[ft_idx(tile, 2 * CH, CH + (e >> 5), CF, ((e >> 4) & 1) ? 80 : 0) + (e & 15)] = 0.f;
        __syncthreads();
    }
}

struct BackArgs {
    const float *hf, *li, *y1, *stats1;
    vadx_dfsmn_cfb_weights w;
    ViewW out;
    float *part;
    int tiles;
};

__global__ __launch_bounds__(NTH) void cfb_back_kernel(BackArgs p) {
    constexpr int NCH = 21, NIT = 3;               // chunks of four ceps bins (81 -> 84); float4 staging items per thread (1280 / 512)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *TI = lds;                               // [10 row tiles][41 k-steps][64 lanes]: 21 real-part + 20 imaginary-part steps
    float *HF = TI + TBLI_FLOATS;                  // [40][4 bins][16] pitch XP: LSTM output
    float *SX = HF + 2 * CH * XP;                  // [40][4 bins][16]: spectrum
    float *OB = SX + 2 * CH * XP;                  // [20][re | im][4 bins][16]: complex product
    float *RED = OB + CH * OP;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, i = lane & 15;
    for (int e = tid; e < TBLI_FLOATS / 4; e += NTH) reinterpret_cast<f32x4 *>(TI)[e] = ldg4(p.w.inv_tbl + 4 * e);
    // Linear 40 -> 40 with rows permuted so that a lane's four D rows are (re c, re c+1, im c, im c+1), c = 8 mt + 2 q
    const int mt0 = wave >> 2, fl0 = wave & 3;
    float wl0[10], wl2[10], bl0[4], bl2[4];
#pragma unroll
    for (int s = 0; s < 10; ++s) {
        wl0[s] = p.w.lin_w[(mt0 * 16 + i) * 40 + 4 * s + q];
        wl2[s] = p.w.lin_w[(2 * 16 + i) * 40 + 4 * s + q];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { bl0[r] = p.w.lin_b[mt0 * 16 + 4 * q + r]; bl2[r] = p.w.lin_b[2 * 16 + 4 * q + r]; }
    const int cg = wave >> 1, mg = wave & 1;           // tile ownership as in cfb_front
    const float *tbl_w = TI + (5 * mg * KSI) * 64 + lane;
    const float *o_w = OB + (5 * cg) * OP + q * 16 + i;

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        f32x4 acc[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 pre[NIT];
        auto request = [&](int jb) {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int e = min(tid + NTH * u, 1279), which = e >= 640, e2 = e - 640 * which, row = e2 >> 2, tq = e2 & 3;
                const int ch = row >> 2, bin = min(4 * jb + (row & 3), CF - 1);           // bins past 80 meet zero table rows: any finite value
                pre[u] = (CFB_EXP & 2) ? f32x4{0.1f, 0.2f, 0.3f, 0.4f} : ldg4((which ? p.li : p.hf) + ft_idx(tile, 2 * CH, ch, CF, bin) + 4 * tq);
            }
        };
        auto park = [&]() {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int e = tid + NTH * u, which = e >= 640, e2 = e - 640 * which, row = e2 >> 2, tq = e2 & 3;
                if (e < 1280) *reinterpret_cast<f32x4 *>((which ? SX : HF) + (row >> 2) * XP + (row & 3) * 16 + 4 * tq) = pre[u];
            }
        };
        auto lin_item = [&](int fl, int mt, const float (&wl)[10], const float (&bl)[4]) {
            f32x4 P = {bl[0], bl[1], bl[2], bl[3]};
            float hv[10];
#pragma unroll
            for (int s = 0; s < 10; ++s) hv[s] = HF[(4 * s + q) * XP + fl * 16 + i];
#pragma unroll
            for (int s = 0; s < 10; ++s) P = mfma16(wl[s], hv[s], P);
            const int ca = 8 * mt + 2 * q;
            if (ca < CH) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = ca + h;
                    const float sre = SX[c * XP + fl * 16 + i], sim = SX[(CH + c) * XP + fl * 16 + i];
                    const float pr = P[h], pi = P[2 + h];
                    OB[c * OP + fl * 16 + i] = pr * sre - pi * sim;
                    OB[c * OP + 64 + fl * 16 + i] = pr * sim + pi * sre;
                }
            }
        };
        request(0);
        park();
        request(1);
        for (int jb = 0; jb < NCH; ++jb) {
            CFB_SYNC();
            if (!(CFB_EXP & 128)) {
                lin_item(fl0, mt0, wl0, bl0);
                if (wave < 4) lin_item(wave, 2, wl2, bl2);
            }
            CFB_SYNC();
            if (jb + 1 < NCH) {
                park();
                if (jb + 2 < NCH) request(jb + 2);
            }
            if (!(CFB_EXP & 16)) {
                float ta[5], ob[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) { ta[u] = tbl_w[(u * KSI + jb) * 64]; ob[u] = o_w[u * OP]; }
#pragma unroll
                for (int jj = 0; jj < TPW; ++jj) acc[jj] = mfma16(ta[jj % 5], ob[jj / 5], acc[jj]);
            }
            if (jb < NCH - 1 && !(CFB_EXP & 16)) {
                float ta[5], ob[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) { ta[u] = tbl_w[(u * KSI + NCH + jb) * 64]; ob[u] = o_w[u * OP + 64]; }
#pragma unroll
                for (int jj = 0; jj < TPW; ++jj) acc[jj] = mfma16(ta[jj % 5], ob[jj / 5], acc[jj]);
            }
        }
        // ---- out = inv1 * (y1 - mean1 * CW) + CB + ceps: the (CW, CB) term is one more k-step, y1 is read in D layout
        const float mean1 = p.stats1[((size_t)tile * 16 + i) * 2], inv1 = p.stats1[((size_t)tile * 16 + i) * 2 + 1];
        const float bfix = q == 0 ? -mean1 * inv1 : (q == 1 ? 1.f : 0.f);
        float ssum = 0.f;
        int ql = q, il = i, lane_l = lane;
        asm volatile("" : "+v"(ql), "+v"(il), "+v"(lane_l));      // see cfb_front: keeps the epilogue's addresses out of the tile loop's preheader
        const float *y1_t = p.y1 + (size_t)tile * (CH * F * 16);
        float *out_t = p.out.ptr + ((size_t)tile * p.out.c_total + p.out.c_off) * (F * 16);
        const float *fix_w = p.w.out_fix + ((5 * cg) * 10 + 5 * mg) * 64;
#pragma unroll
        for (int jj = 0; jj < TPW; ++jj) {
            const int c = 5 * cg + jj / 5, m = 5 * mg + jj % 5;
            const int off = (c * F + 16 * m + 4 * ql) * 16 + il;
            const float *ysrc = y1_t + off;
            float yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = ldg1(ysrc + r * 16);
            acc[jj] = mfma16(ldg1(fix_w + ((jj / 5) * 10 + jj % 5) * 64 + lane_l), bfix, acc[jj]);
            float *dst = out_t + off;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[jj][r] = fmaf(inv1, yv[r], acc[jj][r]);
                stg1(dst + r * 16, acc[jj][r]);
                ssum += acc[jj][r];
            }
        }
        if (p.part) {            // (count, mean, M2) of the block's output per frame, in the partial-statistics format of dfsmn.hip
            ssum = sum_q(ssum);
            __syncthreads();
            if (lane < 16) RED[wave * 16 + i] = ssum;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += RED[w * 16 + i];
            const float nO = (float)(CH * F), meanO = tot / nO;
            float dsum = 0.f;
#pragma unroll
            for (int jj = 0; jj < TPW; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[jj][r] - meanO; dsum = fmaf(d, d, dsum); }
            dsum = sum_q(dsum);
            __syncthreads();
            if (lane < 16) RED[wave * 16 + i] = dsum;
            __syncthreads();
            if (tid < 16) {
                float M = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) M += RED[w * 16 + tid];
                float *o = p.part + (((size_t)tile * VADX_DFSMN_STAT_PARTS) * 16 + tid) * 4;
                o[0] = nO; o[1] = meanO; o[2] = M; o[3] = 0.f;
                float *z = o + 16 * 4;
                z[0] = z[1] = z[2] = z[3] = 0.f;
            }
        }
        __syncthreads();
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
constexpr size_t front_lds_bytes() { return (size_t)(TBLF_FLOATS + CIN * XP + CIN * 8 + CH * RP + CH * GP + RED_FLOATS) * sizeof(float); }
constexpr size_t back_lds_bytes() { return (size_t)(TBLI_FLOATS + 4 * CH * XP + CH * OP + RED_FLOATS) * sizeof(float); }

}  // namespace dfsmn_cfb
}  // namespace vadx

using namespace vadx::dfsmn_cfb;

static bool weights_ok(const vadx_dfsmn_cfb_weights *w) {
    return w && w->ln0_w && w->ln0_b && w->gate_w && w->gate_b && w->in_w && w->in_b && w->ln1_w && w->conv_w && w->ln2_w &&
           w->fwd_tbl && w->fwd_fix && w->lin_w && w->lin_b && w->inv_tbl && w->out_fix;
}

extern "C" int vadx_dfsmn_cfb_front(const vadx_dfsmn_cfb_weights *w, const vadx_ft_view *a, const vadx_ft_view *b, const float *stats0,
                                    float *y1, float *stats1, float *li, float *stats_li, int tiles, void *stream) {
    VADX_REQUIRE(weights_ok(w) && a && a->ptr && stats0 && y1 && stats1 && li && stats_li && tiles > 0, "vadx_dfsmn_cfb_front: bad argument");
    FrontArgs p;
    p.a = View{a->ptr, a->c_total, a->c_off, a->c};
    p.b = b && b->ptr ? View{b->ptr, b->c_total, b->c_off, b->c} : View{nullptr, 0, 0, 0};
    p.stats0 = stats0; p.w = *w; p.y1 = y1; p.stats1 = stats1; p.li = li; p.stats_li = stats_li; p.tiles = tiles;
    const int cin = p.a.c + p.b.c;
    const unsigned grid = (unsigned)(tiles < cu_count() ? tiles : cu_count());
    hipStream_t st = static_cast<hipStream_t>(stream);
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
    VADX_DYN_LDS(cfb_back_kernel, back_lds_bytes());
    hipLaunchKernelGGL(cfb_back_kernel, dim3(grid), dim3(NTH), back_lds_bytes(), static_cast<hipStream_t>(stream), p);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
