// fsmn.hip -- FSMN-VAD (FunASR encoder, explicit 4-cache signature) for gfx950.
// Reference: FSMN/Export_FSMN_VAD.py:75-101 (wrapper), FSMN/modeling_modified/encoder.py:78-83,
// 108-110,139-144,208-217 (encoder), FSMN/Inference_FSMN_VAD_ONNX.py:156-234 (host loop).
//
// One workgroup (8 waves) owns one analysis window (chunk) of one clip at a time and keeps every
// activation of a 64-frame (then 48-frame) tile in LDS, k-major [feature][frame]:
//   LFR x5 + CMVN (applied on the MFMA operand) -> Affine 400->A -> Affine A->L, ReLU
//   -> 4 x [ Linear L->128 | 20-tap causal depthwise FIR over (19-frame cache ++ tile) + skip |
//            Affine 128->L, ReLU ] -> Affine L->A' -> Affine A'->O -> softmax[:,0]
// All dense layers are f32-MFMA GEMMs with weights streamed from L2 (gemm_rt); the FIR, softmax,
// energy gate and the look-ahead vote are VALU/LDS work.  In "clips" mode the workgroup walks the
// overlapping windows of its clip in order, carrying the four FIR caches (global scratch = the
// reference's cache_0..3 tensors) and the adaptive noise floor exactly like the reference loop.
#include "rebalance.h"
#include "common.h"
#include "layers.h"
#include "layers_split.h"

#include <math.h>
#include <string.h>

// FS_EXP: development-only cycle accounting (tools/exp_fsmn.py): per-section clock64 sums of thread 0, all workgroups
#ifndef FS_EXP
#define FS_EXP 0
#endif
#if FS_EXP
__device__ unsigned long long fsmn_dbg[16];
#define FS_T0() long long fs_t_ = clock64()
#define FS_ACC(slot) do { if (threadIdx.x == 0) { const long long n_ = clock64(); atomicAdd(&fsmn_dbg[slot], (unsigned long long)(n_ - fs_t_)); fs_t_ = n_; } } while (0)
extern "C" int vadx_fsmn_debug_cycles(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fsmn_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(fsmn_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define FS_T0() do {} while (0)
#define FS_ACC(slot) do {} while (0)
#endif

namespace vadx {
namespace fsmn {

constexpr int THREADS = 512, NW = 8;
constexpr int PROJ = 128, LORDER = 20, HIST = LORDER - 1, NLAYER = 4, NMEL = 80, LFR_M = 5;
constexpr int A_LD = 68;                 // bufA / bufB row stride: 64 frames + 4 (% 8 == 4)
constexpr int P_LD = 84;                 // bufP row stride: 1 unused + 19 history + 64 frames
constexpr int P_CUR = 20;                // first "current frame" column of bufP (16-B aligned)
constexpr int BUFA_ROWS = 256, BUFB_ROWS = 144;
constexpr int BUFA = BUFA_ROWS * A_LD, BUFB = BUFB_ROWS * A_LD, BUFP = PROJ * P_LD;
constexpr int SMALL = 1024;              // ps[128] | red[512] | sc[128] | misc
constexpr int LDS_FLOATS = BUFA + BUFB + BUFP + SMALL;

struct Dev {
    int A, L, A2, O, Ap, Lp, A2p, Op, T;
    float ratio;
    int off_in1, off_b1, off_mean, off_var, off_in2, off_b2;
    int off_lin[NLAYER], off_fir[NLAYER], off_aff[NLAYER], off_baff[NLAYER];
    int off_out1, off_bo1, off_out2, off_bo2, total;
    // split-product copies of the dense layers (layers_split.h): A fragments [n-tile][32-k chunk][np planes][QFRAG] of ONE arithmetic, the one
    // dims->arithmetic names (split_scheme.h: np = 3 for bf16 x 3, 2 for fp16 x 2, 0 = none: float32 MFMAs on the fragment-major matrices)
    int arith, np, off_flag;                        // off_flag: [sticky range flag, bits of the largest |x|, pad, pad] of the fp16 x 2 kernels
    int split_ok;                                   // 0: dims outside the split tile's LDS map (Ap, A2p <= 160, Lp, Op <= 256)
    int nch_A, nch_L, nch_A2;                       // 32-k chunks of an A- / L- / A2-wide input
    int q_in1, q_b1, q_mbar, q_in2, q_lin[NLAYER], q_aff[NLAYER], q_out1, q_out2;
};
constexpr int NCH_IN1 = 13;                         // 5 x 80 LFR features = 400 -> 13 chunks of 32 (k-groups 50, 51 read a row of zeros)

static int r16(int x) { return (x + 15) & ~15; }

static int derive(const vadx_fsmn_dims *c, Dev *d) {
    memset(d, 0, sizeof(*d));
    if (c->input_affine_dim <= 0 || c->linear_dim <= 0 || c->output_affine_dim <= 0 || c->output_dim <= 0 || c->frames <= 0) return -1;
    d->A = c->input_affine_dim; d->L = c->linear_dim; d->A2 = c->output_affine_dim; d->O = c->output_dim; d->T = c->frames;
    d->Ap = r16(d->A); d->Lp = r16(d->L); d->A2p = r16(d->A2); d->Op = r16(d->O);
    d->ratio = c->speech_2_noise_ratio;
    if (d->Ap > BUFB_ROWS || d->A2p > BUFB_ROWS || d->Lp > BUFA_ROWS || d->Op > BUFA_ROWS) return -1;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 255) & ~255; return r; };      // every section on a 1 KiB boundary (a wave's fragment load = 1 KiB = 16 lines)
    d->off_in1 = take(d->Ap * 400); d->off_b1 = take(d->Ap); d->off_mean = take(400); d->off_var = take(400);
    d->off_in2 = take(d->Lp * d->Ap); d->off_b2 = take(d->Lp);
    for (int l = 0; l < NLAYER; ++l) {
        d->off_lin[l] = take(PROJ * d->Lp); d->off_fir[l] = take(PROJ * LORDER);
        d->off_aff[l] = take(d->Lp * PROJ); d->off_baff[l] = take(d->Lp);
    }
    d->off_out1 = take(d->A2p * d->Lp); d->off_bo1 = take(d->A2p);
    d->off_out2 = take(d->Op * d->A2p); d->off_bo2 = take(d->Op);
    d->nch_A = (d->Ap + 31) / 32; d->nch_L = (d->Lp + 31) / 32; d->nch_A2 = (d->A2p + 31) / 32;
    d->split_ok = d->Ap <= 160 && d->A2p <= 160 && d->Lp <= 256 && d->Op <= 256;
    // AUTO = bf16 x 3 where the split tile fits (float32's exponent range: safe for a caller that never reads the range flag -- the score
    // gate's uint8 output has no NaN to be poisoned with), float32 MFMAs otherwise; fp16 x 2 is an explicit request (VADX_ARITH_F16X2) by
    // callers that run the range protocol, as FsmnEngine does; an explicit split arithmetic on dims outside the tile is refused
    d->arith = vadx::arith_internal(c->arithmetic, d->split_ok ? vadx::VADX_AR_B3 : vadx::VADX_AR_F32);
    if (d->arith < 0 || (d->arith != vadx::VADX_AR_F32 && !d->split_ok)) return -1;
    d->np = d->arith == vadx::VADX_AR_B3 ? 3 : (d->arith == vadx::VADX_AR_H2 ? 2 : 0);
    const int np = d->np;
    d->q_in1 = take(d->Ap / 16 * NCH_IN1 * np * QFRAG); d->q_b1 = take(d->Ap); d->q_mbar = take(NMEL);
    d->q_in2 = take(d->Lp / 16 * d->nch_A * np * QFRAG);
    for (int l = 0; l < NLAYER; ++l) {
        d->q_lin[l] = take(PROJ / 16 * d->nch_L * np * QFRAG);
        d->q_aff[l] = take(d->Lp / 16 * (PROJ / 32) * np * QFRAG);
    }
    d->q_out1 = take(d->A2p / 16 * d->nch_L * np * QFRAG);
    d->q_out2 = take(d->Op / 16 * d->nch_A2 * np * QFRAG);
    d->off_flag = take(4);
    d->total = o;
    return 0;
}

// One tile of MTT*16 frames starting at frame f0 (nvalid of them inside the chunk).
//   lm     : this chunk's log-mel [T][80] (global)
//   cin/cout: FIR caches of this stream, [layer][128][19] (global); cout may alias cin
//   ps     : LDS, receives P(silence) for frames f0 .. f0+nvalid-1
template <int MTT>
__device__ __forceinline__ void tile(const Dev &d, const float *__restrict__ Pk, const float *__restrict__ lm,
                                     int f0, int nvalid, const float *const *cin, float *const *cout,
                                     float *bufA, float *bufB, float *bufP, float *ps, float *red) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));          // per tile: nothing derived from the thread index is hoisted out of the tile / window loops
    constexpr int NF = MTT * 16;

    FS_T0();
    // ---- stage log-mel with LFR edge replication: bufB[mel][c] = lm[clamp(f0 + c - 2, 0, T-1)][mel]
    {   // batches of four loads per thread in flight (the index is always clamped into the chunk)
        constexpr int NE = (NF + 4) * NMEL, NIT = (NE + THREADS - 1) / THREADS;
        for (int u0 = 0; u0 < NIT; u0 += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(tid + THREADS * (u0 + u), NE - 1), c = e / NMEL, mel = e - c * NMEL;
                int fr = f0 + c - 2;
                fr = fr < 0 ? 0 : (fr > d.T - 1 ? d.T - 1 : fr);
                // read-once stream: non-temporal, so that it does not push the weight set (1.67 MB, re-read by every tile) and the FIR
                // caches (rewritten by every tile) out of the 4 MB L2 of the XCD
                v[u] = __builtin_nontemporal_load((vadx::global_f32_ptr)(lm + (size_t)fr * NMEL + mel));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + THREADS * (u0 + u), c = e / NMEL, mel = e - c * NMEL;
                if (e < NE) bufB[mel * A_LD + c] = v[u];
            }
        }
    }
    FS_ACC(0);
    __syncthreads();
    FS_ACC(9);

    LayerArgs a;
    // in_linear1: K = 5 passes x 80 (LFR concat: frame offset j -> column offset j), CMVN on the operand
    a = LayerArgs{Pk + d.off_in1, 400, d.Ap / 16, LFR_M, NMEL / 16, NMEL, 1, Pk + d.off_b1, 0,
                  bufB, A_LD, 0, bufP, A_LD, 0, Pk + d.off_mean, Pk + d.off_var, bufA};      // bufA is idle: K-split scratch
    layer<MTT, true>(a);
    FS_ACC(1);
    __syncthreads();
    FS_ACC(9);
    // FIR caches: the five values a thread carries into bufP's history columns are requested one layer EARLY (layer 0's before in_linear2,
    // layer l + 1's before layer l's linear), so the round trip to the global cache hides behind a GEMM instead of opening every layer
    constexpr int NH = (PROJ * HIST + THREADS - 1) / THREADS;
    auto cache_fetch = [&](int l, float (&hv)[NH]) {
#pragma unroll
        for (int u = 0; u < NH; ++u) { const int e = tid + THREADS * u; hv[u] = ldg1(cin[l] + (e < PROJ * HIST ? e : PROJ * HIST - 1)); }
    };
    float hv[NH];
    cache_fetch(0, hv);
    // in_linear2 + ReLU
    a = LayerArgs{Pk + d.off_in2, d.Ap, d.Lp / 16, 1, d.Ap / 16, 0, 0, Pk + d.off_b2, 1,
                  bufP, A_LD, 0, bufA, A_LD, 0, nullptr, nullptr};
    layer<MTT, false>(a);
    FS_ACC(2);
    __syncthreads();
    FS_ACC(9);

    for (int l = 0; l < NLAYER; ++l) {
        // history columns 1..19 of bufP <- cache (previous tile / previous chunk)
        {
#pragma unroll
            for (int u = 0; u < NH; ++u) {
                const int e = tid + THREADS * u, ch = e / HIST, h = e - ch * HIST;
                if (e < PROJ * HIST) bufP[ch * P_LD + 1 + h] = hv[u];
            }
            cache_fetch(l + 1 < NLAYER ? l + 1 : l, hv);      // next layer's values (the last iteration's request is unused)
        }
        FS_ACC(3);
        a = LayerArgs{Pk + d.off_lin[l], d.Lp, PROJ / 16, 1, d.Lp / 16, 0, 0, nullptr, 0,
                      bufA, A_LD, 0, bufP, P_LD, P_CUR, nullptr, nullptr};
        layer<MTT, false>(a);
        FS_ACC(4);
        __syncthreads();
        FS_ACC(9);
        {   // FIR + skip: thread = (channel, quarter of the tile's frames)
            const int ch = tid >> 2, fq = tid & 3;
            constexpr int FPT = NF / 4;
            const float *wf = Pk + d.off_fir[l] + ch * LORDER;
            float w[LORDER];
#pragma unroll
            for (int k = 0; k < LORDER; ++k) w[k] = ldg1(wf + k);
            const float *seq = bufP + ch * P_LD + 1 + fq * FPT;       // seq[s], s = t + k
            float win[FPT + HIST];
#pragma unroll
            for (int s = 0; s < FPT + HIST; ++s) win[s] = seq[s];
#pragma unroll
            for (int t = 0; t < FPT; ++t) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < LORDER; ++k) s = fmaf(w[k], win[t + k], s);
                bufB[ch * A_LD + fq * FPT + t] = win[t + HIST] + s;
            }
            // new cache = last 19 entries of (history ++ valid frames)
            for (int e = tid; e < PROJ * HIST; e += THREADS) {
                const int c2 = e / HIST, h = e - c2 * HIST;
                stg1(cout[l] + e, bufP[c2 * P_LD + 1 + nvalid + h]);
            }
        }
        FS_ACC(5);
        __syncthreads();
        FS_ACC(9);
        a = LayerArgs{Pk + d.off_aff[l], PROJ, d.Lp / 16, 1, PROJ / 16, 0, 0, Pk + d.off_baff[l], 1,
                      bufB, A_LD, 0, bufA, A_LD, 0, nullptr, nullptr};
        layer<MTT, false>(a);
        FS_ACC(6);
        __syncthreads();
        FS_ACC(9);
    }
    a = LayerArgs{Pk + d.off_out1, d.Lp, d.A2p / 16, 1, d.Lp / 16, 0, 0, Pk + d.off_bo1, 0,
                  bufA, A_LD, 0, bufB, A_LD, 0, nullptr, nullptr, bufP};                      // bufP is idle: K-split scratch
    layer<MTT, false>(a);
    __syncthreads();
    a = LayerArgs{Pk + d.off_out2, d.A2p, d.Op / 16, 1, d.A2p / 16, 0, 0, Pk + d.off_bo2, 0,
                  bufB, A_LD, 0, bufA, A_LD, 0, nullptr, nullptr};
    layer<MTT, false>(a);
    FS_ACC(7);
    __syncthreads();

    FS_ACC(9);
    // ---- softmax over the O logits of each frame, keep class 0: thread = (part 0..7, frame 0..63)
    {
        const int m = tid & 63, part = tid >> 6;
        float mx = -INFINITY;
        if (m < NF) for (int n = part; n < d.O; n += NW) mx = fmaxf(mx, bufA[n * A_LD + m]);
        red[part * 64 + m] = mx;
        __syncthreads();
        float gm = red[m];
#pragma unroll
        for (int p2 = 1; p2 < NW; ++p2) gm = fmaxf(gm, red[p2 * 64 + m]);
        __syncthreads();
        float sm = 0.f;
        if (m < NF) for (int n = part; n < d.O; n += NW) sm += expf(bufA[n * A_LD + m] - gm);
        red[part * 64 + m] = sm;
        __syncthreads();
        if (part == 0 && m < nvalid) {
            float tot = 0.f;
#pragma unroll
            for (int p2 = 0; p2 < NW; ++p2) tot += red[p2 * 64 + m];
            ps[f0 + m] = expf(bufA[m] - gm) / tot;
        }
        __syncthreads();
    }
    FS_ACC(8);
}


// ---- the same tile on bf16 x 3 split products (layers_split.h) ---------------------------------------------------------------------
// Every dense layer runs as six v_mfma_f32_16x16x32_bf16 per K = 32 step on exactly split float32 operands (6/16 of the f32-MFMA matrix
// time, float32-class accuracy).  Activations that feed a GEMM live in LDS as three bf16 planes [k / 8][frame][8]; the two float32
// buffers that remain are the FIR's input P [128][history ++ frames] and the softmax's logits.
//   * CMVN: the reference's (x + mean) * var on the LFR features moves into the first layer, W1' = W1 var, b1' = b1 + W1' (mean - mbar),
//     evaluated in double on the host; the staged log-mel is pre-centred with mbar = the centre LFR position's means (one float32
//     add), so that the products keep the magnitude of centred features.  Re-association only.
//   * the FIR thread owns four consecutive channels x four frames, so its outputs leave as one 8-byte store per plane and frame.
// LDS map (bytes), 160 KB with the small block -- tensors of one phase never overlap (H = planes of an A-wide tensor padded to 160
// channels, HL = planes of an L-wide tensor [256][64], P = float32 [128][84], FO = FIR output planes [128][64]):
//   in1: LM @61440 -> H1 @0        in2: H1 @0 -> HL @61440 ("mid")
//   even layer: HL mid -> P @0 -> FO @110592 -> HL @0 ("low")         odd layer: HL low -> P @98304 -> FO @0 -> HL mid
//   out1: HL mid -> H2 @0          out2: H2 @0 -> logits f32 [256][68] @61440
constexpr int SQ_H = 0, SQ_LM = 61440, SQ_HLM = 61440, SQ_HLL = 0, SQ_PE = 0, SQ_PO = 98304, SQ_FOH = 110592, SQ_FOL = 0, SQ_LOG = 61440;
constexpr int SQ_ARENA = 159744, SQ_LDS_BYTES = SQ_ARENA + SMALL * 4;
static_assert(SQ_HLM + 256 * 64 * 6 <= SQ_ARENA && SQ_FOH + 128 * 64 * 6 <= SQ_ARENA && SQ_PO + BUFP * 4 <= SQ_FOH + 128 * 64 * 6 &&
              SQ_LOG + 256 * A_LD * 4 <= SQ_ARENA && SQ_LM + 11 * 80 * 16 * 3 <= SQ_ARENA && 160 * 64 * 6 <= SQ_HLM && SQ_LDS_BYTES <= 160 * 1024,
              "split tile LDS map");

template <typename SC, int MTT>
__device__ __forceinline__ void tile_split(const Dev &d, const float *__restrict__ Pk, const float *__restrict__ lm,
                                           int f0, int nvalid, const float *const *cin, float *const *cout,
                                           unsigned char *smem, float *ps, float *red, float &amax) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    constexpr int NP = SC::NP;
    constexpr int NF = MTT * 16, NCL = NF + 16;             // columns of a plane row: frames (log-mel: + LFR context)
    const int lane = tid & 63, i = lane & 15;
    auto grp_pl = [](int kgrps, int ncol) { return kgrps * ncol * 16; };      // bytes of one plane

    FS_T0();
    // ---- zero rows: k-groups of the A-wide tensors beyond Ap (their weights are zero, the operand must be finite), the log-mel's row 10
    {
        const int kg0 = d.Ap / 8, kg1 = 4 * d.nch_A, pl = grp_pl(kg1, NF);
        for (int e = tid; e < NP * (kg1 - kg0) * NF; e += THREADS) {
            const int p = e / ((kg1 - kg0) * NF), r = e - p * (kg1 - kg0) * NF;
            *reinterpret_cast<f32x4 *>(smem + SQ_H + p * pl + (kg0 * NF + r) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int e = tid; e < NP * NCL; e += THREADS) {
            const int p = e / NCL, c = e - p * NCL;
            *reinterpret_cast<f32x4 *>(smem + SQ_LM + p * grp_pl(11, NCL) + (10 * NCL + c) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // ---- stage log-mel with LFR edge replication, pre-centred, as planes: column c = frame clamp(f0 + c - 2), item = (column, 4 mels)
    {
        constexpr int NE = (NF + 4) * (NMEL / 4), NIT = (NE + THREADS - 1) / THREADS;
        f32x4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int e = min(tid + THREADS * u, NE - 1), c = e / (NMEL / 4), mg = e - c * (NMEL / 4);
            int fr = f0 + c - 2;
            fr = fr < 0 ? 0 : (fr > d.T - 1 ? d.T - 1 : fr);
            v[u] = __builtin_nontemporal_load((vadx::global_f32x4_ptr)(lm + (size_t)fr * NMEL + 4 * mg));
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int e = tid + THREADS * u, c = e / (NMEL / 4), mg = e - c * (NMEL / 4);
            if (e < NE) {
                const f32x4 x = v[u] + ldg4(Pk + d.q_mbar + 4 * mg);
                u32x2 pp[NP];
                SC::split4(x, pp, amax);
                unsigned char *dp = smem + SQ_LM + ((mg >> 1) * NCL + c) * 16 + (mg & 1) * 8;
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2 *>(dp + p * grp_pl(11, NCL)) = pp[p];
            }
        }
    }
    FS_ACC(0);
    __syncthreads();
    FS_ACC(9);

    auto plain = [&](int ncol) { return [=](int kgrp, int mt) { return (kgrp * ncol + mt * 16 + i) * 16; }; };
    QLayerArgs a;
    // in_linear1: K = 5 LFR positions x 80 mels = 50 k-groups; k-group G = (position j = G / 10, mel group G % 10) reads column + j
    a = QLayerArgs{Pk + d.q_in1, d.Ap / 16, NCH_IN1, Pk + d.q_b1, 0, smem + SQ_LM, grp_pl(11, NCL), smem + SQ_H, grp_pl(4 * d.nch_A, NF), NF, nullptr};
    qlayer<SC, MTT, true>(a, [=](int G, int mt) { const int j = G / 10, mg = G - 10 * j; return G < 50 ? (mg * NCL + mt * 16 + i + j) * 16 : (10 * NCL + mt * 16 + i) * 16; }, amax);
    FS_ACC(1);
    __syncthreads();
    FS_ACC(9);
    constexpr int NH = (PROJ * HIST + THREADS - 1) / THREADS;
    auto cache_fetch = [&](int l, float (&hv)[NH]) {
#pragma unroll
        for (int u = 0; u < NH; ++u) { const int e = tid + THREADS * u; hv[u] = ldg1(cin[l] + (e < PROJ * HIST ? e : PROJ * HIST - 1)); }
    };
    float hv[NH];
    cache_fetch(0, hv);
    // in_linear2 + ReLU
    a = QLayerArgs{Pk + d.q_in2, d.Lp / 16, d.nch_A, Pk + d.off_b2, 1, smem + SQ_H, grp_pl(4 * d.nch_A, NF), smem + SQ_HLM, grp_pl(4 * d.nch_L, NF), NF, nullptr};
    qlayer<SC, MTT, true>(a, plain(NF), amax);
    FS_ACC(2);
    __syncthreads();
    FS_ACC(9);

    for (int l = 0; l < NLAYER; ++l) {
        const bool even = (l & 1) == 0;
        unsigned char *HLin = smem + (even ? SQ_HLM : SQ_HLL), *HLout = smem + (even ? SQ_HLL : SQ_HLM);
        float *bufP = reinterpret_cast<float *>(smem + (even ? SQ_PE : SQ_PO));
        unsigned char *FO = smem + (even ? SQ_FOH : SQ_FOL);
        {   // history columns 1..19 of P <- cache (previous tile / previous chunk)
#pragma unroll
            for (int u = 0; u < NH; ++u) {
                const int e = tid + THREADS * u, ch = e / HIST, h = e - ch * HIST;
                if (e < PROJ * HIST) bufP[ch * P_LD + 1 + h] = hv[u];
            }
            cache_fetch(l + 1 < NLAYER ? l + 1 : l, hv);
        }
        FS_ACC(3);
        a = QLayerArgs{Pk + d.q_lin[l], PROJ / 16, d.nch_L, nullptr, 0, HLin, grp_pl(4 * d.nch_L, NF), reinterpret_cast<unsigned char *>(bufP), P_CUR, P_LD, nullptr};
        qlayer<SC, MTT, false>(a, plain(NF), amax);
        FS_ACC(4);
        // the FIR taps of this thread's four channels (80 floats) are requested BEFORE the barrier: their L2 round trip hides behind the
        // wait for the slowest wave of the GEMM instead of opening the FIR (per channel: load, wait, 80 FMAs -- four times in a row)
        const int cg = tid >> 4, fg = tid & 15;
        f32x4 fw[4][LORDER / 4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k4 = 0; k4 < LORDER / 4; ++k4) fw[c][k4] = ldg4(Pk + d.off_fir[l] + (4 * cg + c) * LORDER + 4 * k4);
        __syncthreads();
        FS_ACC(9);
        {   // FIR + skip: thread = (4 consecutive channels, 4 frames); the four channels of a frame leave as one 8-byte store per plane
            if (fg < NF / 4) {
                f32x4 o[4];                                               // o[t][c]
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ch = 4 * cg + c;
                    float w[LORDER], v[24];
#pragma unroll
                    for (int k4 = 0; k4 < LORDER / 4; ++k4) {
                        const f32x4 w4 = fw[c][k4];
                        w[4 * k4] = w4[0]; w[4 * k4 + 1] = w4[1]; w[4 * k4 + 2] = w4[2]; w[4 * k4 + 3] = w4[3];
                    }
                    const float *row = bufP + ch * P_LD + 4 * fg;         // v[1 + s] = seq[s] of the f32 tile (column 1 + 4 fg + s)
#pragma unroll
                    for (int b4 = 0; b4 < 6; ++b4) {
                        const f32x4 x4 = *reinterpret_cast<const f32x4 *>(row + 4 * b4);
                        v[4 * b4] = x4[0]; v[4 * b4 + 1] = x4[1]; v[4 * b4 + 2] = x4[2]; v[4 * b4 + 3] = x4[3];
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float s = 0.f;
#pragma unroll
                        for (int k = 0; k < LORDER; ++k) s = fmaf(w[k], v[1 + t + k], s);
                        o[t][c] = v[1 + t + HIST] + s;
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    u32x2 pp[NP];
                    SC::split4(o[t], pp, amax);
                    unsigned char *dp = FO + ((cg >> 1) * NF + 4 * fg + t) * 16 + (cg & 1) * 8;
#pragma unroll
                    for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2 *>(dp + p * grp_pl(PROJ / 8, NF)) = pp[p];
                }
            }
            for (int e = tid; e < PROJ * HIST; e += THREADS) {            // new cache = last 19 entries of (history ++ valid frames)
                const int c2 = e / HIST, h = e - c2 * HIST;
                stg1(cout[l] + e, bufP[c2 * P_LD + 1 + nvalid + h]);
            }
        }
        FS_ACC(5);
        __syncthreads();
        FS_ACC(9);
        a = QLayerArgs{Pk + d.q_aff[l], d.Lp / 16, PROJ / 32, Pk + d.off_baff[l], 1, FO, grp_pl(PROJ / 8, NF), HLout, grp_pl(4 * d.nch_L, NF), NF, nullptr};
        qlayer<SC, MTT, true>(a, plain(NF), amax);
        FS_ACC(6);
        __syncthreads();
        FS_ACC(9);
    }
    {   // zero rows of the A2-wide tensor (region 0 is free again)
        const int kg0 = d.A2p / 8, kg1 = 4 * d.nch_A2, pl = grp_pl(kg1, NF);
        for (int e = tid; e < NP * (kg1 - kg0) * NF; e += THREADS) {
            const int p = e / ((kg1 - kg0) * NF), r = e - p * (kg1 - kg0) * NF;
            *reinterpret_cast<f32x4 *>(smem + SQ_H + p * pl + (kg0 * NF + r) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    a = QLayerArgs{Pk + d.q_out1, d.A2p / 16, d.nch_L, Pk + d.off_bo1, 0, smem + SQ_HLM, grp_pl(4 * d.nch_L, NF), smem + SQ_H, grp_pl(4 * d.nch_A2, NF), NF, nullptr};
    qlayer<SC, MTT, true>(a, plain(NF), amax);
    __syncthreads();
    float *logits = reinterpret_cast<float *>(smem + SQ_LOG);
    a = QLayerArgs{Pk + d.q_out2, d.Op / 16, d.nch_A2, Pk + d.off_bo2, 0, smem + SQ_H, grp_pl(4 * d.nch_A2, NF), reinterpret_cast<unsigned char *>(logits), 0, A_LD, nullptr};
    qlayer<SC, MTT, false>(a, plain(NF), amax);
    FS_ACC(7);
    __syncthreads();
    FS_ACC(9);
    {   // softmax over the O logits of each frame, keep class 0 (as tile<>)
        const int m = tid & 63, part = tid >> 6;
        float mx = -INFINITY;
        if (m < NF) for (int n = part; n < d.O; n += NW) mx = fmaxf(mx, logits[n * A_LD + m]);
        red[part * 64 + m] = mx;
        __syncthreads();
        float gm = red[m];
#pragma unroll
        for (int p2 = 1; p2 < NW; ++p2) gm = fmaxf(gm, red[p2 * 64 + m]);
        __syncthreads();
        float sm = 0.f;
        if (m < NF) for (int n = part; n < d.O; n += NW) sm += expf(logits[n * A_LD + m] - gm);
        red[part * 64 + m] = sm;
        __syncthreads();
        if (part == 0 && m < nvalid) {
            float tot = 0.f;
#pragma unroll
            for (int p2 = 0; p2 < NW; ++p2) tot += red[p2 * 64 + m];
            ps[f0 + m] = expf(logits[m] - gm) / tot;
        }
        __syncthreads();
    }
    FS_ACC(8);
}

// score gate of one chunk (FSMN/Export_FSMN_VAD.py:87-101): returns noisy_dB (NaN if no frame is "noise")
__device__ __forceinline__ float gate(const Dev &d, const float *ps, const float *__restrict__ db, float thr,
                                      float noise_db, unsigned char *__restrict__ score_out, float *__restrict__ psil_out,
                                      float *sc, float *red) {
    const int tid = threadIdx.x;
    float part_sum = 0.f, part_cnt = 0.f;
    if (tid < 128) {
        for (int t = tid; t < d.T; t += 128) {
            const float p = ps[t];
            float s;
            if (d.ratio > 1.0f) s = p + powf(p, d.ratio);
            else if (d.ratio < 1.0f) s = p + 1.0f;
            else s = p + p;
            const float e = db[t];
            const bool cond = (s <= thr) && (e >= noise_db);
            sc[t] = cond ? 1.f : 0.f;
            if (score_out) score_out[t] = cond ? 1 : 0;
            if (psil_out) psil_out[t] = p;
            if (!cond) { part_sum += e; part_cnt += 1.f; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { part_sum += __shfl_xor(part_sum, o); part_cnt += __shfl_xor(part_cnt, o); }
        if ((tid & 63) == 0) { red[(tid >> 6) * 2] = part_sum; red[(tid >> 6) * 2 + 1] = part_cnt; }
    }
    __syncthreads();
    const float tot = red[0] + red[2], cnt = red[1] + red[3];
    __syncthreads();
    return tot / cnt;                      // 0/0 -> NaN like torch's mean of an empty tensor
}

// AR: the arithmetic of the dense layers (split_scheme.h: 0 float32 MFMAs, 1 bf16 x 3, 2 fp16 x 2)
template <int AR> struct SchemeOf { typedef vadx::SchemeB3 type; };
template <> struct SchemeOf<vadx::VADX_AR_H2> { typedef vadx::SchemeH2 type; };
template <int AR>
__device__ __forceinline__ void run_chunk(const Dev &d, const float *Pk, const float *lm, const float *const *cin,
                                          float *const *cout, float *lds, float &amax) {
    constexpr bool SPLIT = AR != vadx::VADX_AR_F32;
    typedef typename SchemeOf<AR>::type SC;
    float *bufA = lds, *bufB = lds + BUFA, *bufP = bufB + BUFB;
    float *small = SPLIT ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(lds) + SQ_ARENA) : bufP + BUFP;
    float *ps = small, *red = small + 128;
    unsigned char *smem = reinterpret_cast<unsigned char *>(lds);
    int f0 = 0;
    bool first = true;
    while (f0 < d.T) {
        const int left = d.T - f0;
        const float *const *ci = first ? cin : cout;         // later tiles continue from the updated cache
        if (SPLIT) {
            if (left > 48) tile_split<SC, 4>(d, Pk, lm, f0, left < 64 ? left : 64, ci, cout, smem, ps, red, amax), f0 += 64;
            else if (left > 32) tile_split<SC, 3>(d, Pk, lm, f0, left, ci, cout, smem, ps, red, amax), f0 += 48;
            else if (left > 16) tile_split<SC, 2>(d, Pk, lm, f0, left, ci, cout, smem, ps, red, amax), f0 += 32;
            else tile_split<SC, 1>(d, Pk, lm, f0, left, ci, cout, smem, ps, red, amax), f0 += 16;
        } else {
            if (left > 48) tile<4>(d, Pk, lm, f0, left < 64 ? left : 64, ci, cout, bufA, bufB, bufP, ps, red), f0 += 64;
            else if (left > 32) tile<3>(d, Pk, lm, f0, left, ci, cout, bufA, bufB, bufP, ps, red), f0 += 48;
            else if (left > 16) tile<2>(d, Pk, lm, f0, left, ci, cout, bufA, bufB, bufP, ps, red), f0 += 32;
            else tile<1>(d, Pk, lm, f0, left, ci, cout, bufA, bufB, bufP, ps, red), f0 += 16;
        }
        first = false;
    }
}

struct RunArgs {
    const float *logmel, *db;             // [B][T][80], [B][T]
    const float *cin[NLAYER]; float *cout[NLAYER];   // each [B][128][19]
    const float *thr, *noise_db;          // [B]
    unsigned char *score; float *noisy_db, *psil;    // [B][T], [B], [B][T] (psil optional)
};

// ORT-boundary equivalent: one chunk per stream, B independent streams.
template <int AR>
__global__ __launch_bounds__(THREADS, 2) void fsmn_run_kernel(Dev d, const float *__restrict__ Pk, RunArgs r) {
    constexpr bool SPLIT = AR != vadx::VADX_AR_F32;
    float amax = 0.f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x;
    const float *cin[NLAYER]; float *cout[NLAYER];
#pragma unroll
    for (int l = 0; l < NLAYER; ++l) { cin[l] = r.cin[l] + (size_t)b * PROJ * HIST; cout[l] = r.cout[l] + (size_t)b * PROJ * HIST; }
    run_chunk<AR>(d, Pk, r.logmel + (size_t)b * d.T * NMEL, cin, cout, lds, amax);
    if (AR == vadx::VADX_AR_H2) vadx::range_flag_raise(Pk + d.off_flag, amax);
    float *small = SPLIT ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(lds) + SQ_ARENA) : lds + BUFA + BUFB + BUFP;
    const float noisy = gate(d, small, r.db + (size_t)b * d.T, r.thr[b], r.noise_db[b], r.score + (size_t)b * d.T,
                             r.psil ? r.psil + (size_t)b * d.T : nullptr, small + 640, small + 128);
    if (threadIdx.x == 0) r.noisy_db[b] = noisy;
}

struct ClipArgs {
    const float *logmel, *db;             // [B*W][T][80], [B*W][T]
    float *cache;                         // scratch [B][4][128][19]
    int W, slide, lb;                     // windows per clip, slide_range, look_backward
    float thr, noise0, snr;               // one_minus_speech_threshold, initial noise floor (x0.1), SNR (x0.1)
    double speaking, silence_score;
    unsigned char *flags;                 // [B][W*slide + (T - slide)] silence flags
    float *noise_trace;                   // optional [B][W]
};

// Whole clips: the reference's while-loop (Inference_FSMN_VAD_ONNX.py:176-234), one workgroup per clip.
template <int AR>
__global__ __launch_bounds__(THREADS, 2) void fsmn_clips_kernel(Dev d, const float *__restrict__ Pk, ClipArgs c) {
    constexpr bool SPLIT = AR != vadx::VADX_AR_F32;
    float amax = 0.f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    float *small = SPLIT ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(lds) + SQ_ARENA) : lds + BUFA + BUFB + BUFP;
    float *ps = small, *red = small + 128, *sc = small + 640, *cnt = small + 768;
    float *cbase = c.cache + (size_t)b * NLAYER * PROJ * HIST;
    for (int e = tid; e < NLAYER * PROJ * HIST; e += THREADS) cbase[e] = 0.f;
    __syncthreads();
    const float *cin[NLAYER]; float *cout[NLAYER];
#pragma unroll
    for (int l = 0; l < NLAYER; ++l) { cin[l] = cbase + l * PROJ * HIST; cout[l] = cbase + l * PROJ * HIST; }
    float noise = c.noise0;
    int silence = 1;                       // carried by thread 0
    const int nflags = c.W * c.slide + (d.T - c.slide);
    unsigned char *fl = c.flags + (size_t)b * nflags;
    for (int k = 0; k < c.W; ++k) {
        const size_t widx = (size_t)b * c.W + k;
        run_chunk<AR>(d, Pk, c.logmel + widx * d.T * NMEL, cin, cout, lds, amax);
        const float noisy = gate(d, ps, c.db + widx * d.T, c.thr, noise, nullptr, nullptr, sc, red);
        // look-ahead vote: cnt[i] = #{ j in [1,lb) : sc[i+j] != 0 }
        if (tid < c.slide) {
            float s = 0.f;
            for (int j = 1; j < c.lb; ++j) s += sc[tid + j];
            cnt[tid] = s;
        }
        __syncthreads();
        if (tid == 0) {
            const double inv_lb = 1.0 / (double)c.lb;
            for (int i = 0; i < c.slide; ++i) {
                if (silence) {
                    if (sc[i] != 0.f) silence = !((1.0 + (double)cnt[i]) * inv_lb >= c.speaking);
                    else silence = 1;
                } else {
                    if (sc[i] != 1.f) silence = !((1.0 + (double)((c.lb - 1) - cnt[i])) * inv_lb <= c.silence_score);
                    else silence = 0;
                }
                fl[k * c.slide + i] = (unsigned char)silence;
            }
            if (k == c.W - 1)              // tail of the final chunk: plain rule (:223-234)
                for (int i = c.slide; i < d.T; ++i) {
                    silence = silence ? !(sc[i] != 0.f) : (sc[i] != 1.f);
                    fl[c.W * c.slide + (i - c.slide)] = (unsigned char)silence;
                }
        }
        if (noisy > 0.0f) noise = 0.5f * ((noise + noisy) + c.snr);
        if (c.noise_trace && tid == 0) c.noise_trace[widx] = noise;
        __syncthreads();
    }
    if (AR == vadx::VADX_AR_H2) vadx::range_flag_raise(Pk + d.off_flag, amax);
}

// frame energy in dB/10 of the prepped window (FSMN/Export_FSMN_VAD.py:93-97): one workgroup per window
__global__ void fsmn_energy_kernel(const int16_t *__restrict__ audio, long long row_stride, long long win_stride,
                                   int windows_per_clip, int window_len, int n_fft, int hop, int T,
                                   const float *__restrict__ means, float inv_ref, float *__restrict__ db) {
    const int widx = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int b = widx / windows_per_clip, w = widx - b * windows_per_clip;
    const int16_t *win = audio + (long long)b * row_stride + (long long)w * win_stride;
    const float mean = means[widx];
    const int nfr = (window_len - n_fft) / hop + 1;
    float *out = db + (size_t)widx * T;
    for (int f = wave; f < T; f += nw) {
        const int ff = f < nfr ? f : nfr - 1;          // last value repeated up to T frames
        float s = 0.f;
        // eight (sample, predecessor) pairs per lane are requested before any is used, unconditionally (clamped index, value selected
        // afterwards): the guarded predecessor load compiled to a branch plus a full wait -- one serialised round trip per sample.
        // Accumulation order per lane is unchanged (k ascending): the sums are bit-identical.
        for (int k0 = lane; k0 < n_fft; k0 += 64 * 8) {
            float xa[8], xb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = min(k0 + 64 * u, n_fft - 1), n = ff * hop + k;
                xa[u] = (float)win[n];
                xb[u] = (float)win[n > 0 ? n - 1 : 0];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 64 * u, n = ff * hop + k;
                if (k < n_fft) {
                    const float a = __fsub_rn(xa[u], mean);
                    float y = a;
                    if (n > 0) y = __fsub_rn(a, __fmul_rn(0.97f, __fsub_rn(xb[u], mean)));
                    y = __fmul_rn(y, inv_ref);
                    s = fmaf(y, y, s);
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) out[f] = log10f(s + 0.00002f);
    }
}

// Window mean AND frame energies in one pass over the PCM (round 6; replaces window_mean_kernel + fsmn_energy_kernel for the aligned case: the two
// read every window from memory once each, and the energy kernel visited every sample 3.2 times -- 512-sample frames at a hop of 160).
// One workgroup of 256 threads per window: a thread keeps its (up to) eight 16-byte runs of samples in registers, the workgroup sums them as
// integers (the mean, exactly as window_mean_kernel computes it), then every run contributes sum(y^2) of its eight prepped samples ONCE; four
// neighbouring lanes add up to a 32-sample granule, and a frame is sixteen consecutive granules (512 = 16 x 32, 160 = 5 x 32).
// Same formula as fsmn_energy_kernel, another summation order: the dB values agree to float32 rounding (1e-7 relative).
constexpr int ST_THREADS = 256, ST_MAXIT = 8, ST_MAXGRAN = (ST_THREADS * ST_MAXIT * 8) / 32;       // windows up to 16 384 samples
__global__ __launch_bounds__(ST_THREADS) void fsmn_stats_kernel(const int16_t *__restrict__ audio, long long row_stride, long long win_stride,
                                                                int windows_per_clip, int window_len, int T, float inv_ref,
                                                                float *__restrict__ means, float *__restrict__ db) {
    __shared__ float gran[ST_MAXGRAN];
    __shared__ long long wsum[ST_THREADS / 64];
    const int widx = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = widx / windows_per_clip, w = widx - b * windows_per_clip;
    const int16_t *win = audio + (long long)b * row_stride + (long long)w * win_stride;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const int nrun = window_len / 8;
    s16x8 x[ST_MAXIT];
    long long s = 0;
#pragma unroll
    for (int it = 0; it < ST_MAXIT; ++it) {
        const int r = it * ST_THREADS + tid;
        x[it] = *reinterpret_cast<const s16x8 *>(win + 8 * (r < nrun ? r : nrun - 1));
        if (r < nrun)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[it][e];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    long long tot = 0;
#pragma unroll
    for (int k = 0; k < ST_THREADS / 64; ++k) tot += wsum[k];
    const float mean = (float)((double)tot / (double)window_len);
    if (tid == 0) means[widx] = mean;
#pragma unroll
    for (int it = 0; it < ST_MAXIT; ++it) {
        const int r = it * ST_THREADS + tid;
        // the sample in front of this run: the previous lane's last sample (the previous run), the first lane of a wave reads it from memory
        float prev = (float)__shfl_up((int)x[it][7], 1);
        if (lane == 0) prev = (float)win[r > 0 ? 8 * (r < nrun ? r : nrun - 1) - 1 : 0];
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = __fsub_rn((float)x[it][e], mean);
            float y = a;
            if (e > 0 || r > 0) y = __fsub_rn(a, __fmul_rn(0.97f, __fsub_rn(prev, mean)));
            y = __fmul_rn(y, inv_ref);
            acc = fmaf(y, y, acc);
            prev = (float)x[it][e];
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if ((lane & 3) == 0 && r < nrun) gran[r >> 2] = acc;
    }
    __syncthreads();
    const int nfr = (window_len - 512) / 160 + 1;
    float *out = db + (size_t)widx * T;
    for (int f = tid; f < T; f += ST_THREADS) {
        const int ff = f < nfr ? f : nfr - 1;          // last value repeated up to T frames
        float e = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) e += gran[5 * ff + j];
        out[f] = log10f(e + 0.00002f);
    }
}

}  // namespace fsmn
}  // namespace vadx

using namespace vadx::fsmn;
using vadx::QFRAG;

extern "C" size_t vadx_fsmn_packed_floats(const vadx_fsmn_dims *dims) {
    Dev d;
    if (!dims || derive(dims, &d)) return 0;
    return (size_t)d.total;
}

extern "C" int vadx_fsmn_pack_host(const vadx_fsmn_dims *dims, const vadx_fsmn_weights_host *w_in, float *p) {
    Dev d;
    VADX_REQUIRE(dims && w_in && p, "vadx_fsmn_pack_host: NULL argument");
    VADX_REQUIRE(derive(dims, &d) == 0, "vadx_fsmn_pack_host: unsupported dims (affine <= 144, linear/output <= 256)");
    memset(p, 0, sizeof(float) * d.total);
    auto mat = [&](int off, const float *src, int rows, int cols, int ld) {
        for (int r = 0; r < rows; ++r) memcpy(p + off + (size_t)r * ld, src + (size_t)r * cols, cols * sizeof(float));
    };
    VADX_REQUIRE(w_in->in1_w && w_in->in1_b && w_in->in2_w && w_in->in2_b && w_in->out1_w && w_in->out1_b && w_in->out2_w && w_in->out2_b &&
                 w_in->cmvn_means && w_in->cmvn_vars, "vadx_fsmn_pack_host: NULL weight pointer");
    for (int l = 0; l < NLAYER; ++l)
        VADX_REQUIRE(w_in->lin_w[l] && w_in->fir_w[l] && w_in->aff_w[l] && w_in->aff_b[l], "vadx_fsmn_pack_host: NULL layer %d weight", l);
    // Exact power-of-two rebalancing of the affine chains (csrc/rebalance.h; ordinary checkpoints pass through untouched).  The projection p of
    // every FSMN block is visible outside the kernels (the FIR caches: cache_0..3 of the reference's session), so it stays at its true scale
    // and ends a segment: [in1, in2, lin_0], [aff_l, lin_(l+1)], [aff_3, out1, out2] (softmax after out2).  FIR + skip, ReLU and the
    // activation-free in1 -> in2 / out1 -> out2 pairs are positively homogeneous.
    std::vector<float> r_in1(w_in->in1_w, w_in->in1_w + (size_t)d.A * 400), r_in1b(w_in->in1_b, w_in->in1_b + d.A);
    std::vector<float> r_in2(w_in->in2_w, w_in->in2_w + (size_t)d.L * d.A), r_in2b(w_in->in2_b, w_in->in2_b + d.L);
    std::vector<float> r_out1(w_in->out1_w, w_in->out1_w + (size_t)d.A2 * d.L), r_out1b(w_in->out1_b, w_in->out1_b + d.A2);
    std::vector<float> r_out2(w_in->out2_w, w_in->out2_w + (size_t)d.O * d.A2), r_out2b(w_in->out2_b, w_in->out2_b + d.O);
    std::vector<float> r_lin[NLAYER], r_aff[NLAYER], r_affb[NLAYER];
    for (int l = 0; l < NLAYER; ++l) {
        r_lin[l].assign(w_in->lin_w[l], w_in->lin_w[l] + (size_t)PROJ * d.L);
        r_aff[l].assign(w_in->aff_w[l], w_in->aff_w[l] + (size_t)d.L * PROJ);
        r_affb[l].assign(w_in->aff_b[l], w_in->aff_b[l] + d.L);
    }
    int reb_min = 1000;
    vadx::rebalance_chain({{&r_in1, &r_in1b}, {&r_in2, &r_in2b}, {&r_lin[0], nullptr}}, &reb_min);
    for (int l = 0; l + 1 < NLAYER; ++l) vadx::rebalance_chain({{&r_aff[l], &r_affb[l]}, {&r_lin[l + 1], nullptr}}, &reb_min);
    vadx::rebalance_chain({{&r_aff[NLAYER - 1], &r_affb[NLAYER - 1]}, {&r_out1, &r_out1b}, {&r_out2, &r_out2b}}, &reb_min);
    vadx_fsmn_weights_host w_reb = *w_in;
    w_reb.in1_w = r_in1.data(); w_reb.in1_b = r_in1b.data(); w_reb.in2_w = r_in2.data(); w_reb.in2_b = r_in2b.data();
    w_reb.out1_w = r_out1.data(); w_reb.out1_b = r_out1b.data(); w_reb.out2_w = r_out2.data(); w_reb.out2_b = r_out2b.data();
    for (int l = 0; l < NLAYER; ++l) { w_reb.lin_w[l] = r_lin[l].data(); w_reb.aff_w[l] = r_aff[l].data(); w_reb.aff_b[l] = r_affb[l].data(); }
    const vadx_fsmn_weights_host *w = &w_reb;
    mat(d.off_in1, w->in1_w, d.A, 400, 400); memcpy(p + d.off_b1, w->in1_b, d.A * sizeof(float));
    memcpy(p + d.off_mean, w->cmvn_means, 400 * sizeof(float)); memcpy(p + d.off_var, w->cmvn_vars, 400 * sizeof(float));
    mat(d.off_in2, w->in2_w, d.L, d.A, d.Ap); memcpy(p + d.off_b2, w->in2_b, d.L * sizeof(float));
    for (int l = 0; l < NLAYER; ++l) {
        VADX_REQUIRE(w->lin_w[l] && w->fir_w[l] && w->aff_w[l] && w->aff_b[l], "vadx_fsmn_pack_host: NULL layer %d weight", l);
        mat(d.off_lin[l], w->lin_w[l], PROJ, d.L, d.Lp);
        memcpy(p + d.off_fir[l], w->fir_w[l], PROJ * LORDER * sizeof(float));
        mat(d.off_aff[l], w->aff_w[l], d.L, PROJ, PROJ); memcpy(p + d.off_baff[l], w->aff_b[l], d.L * sizeof(float));
    }
    mat(d.off_out1, w->out1_w, d.A2, d.L, d.Lp); memcpy(p + d.off_bo1, w->out1_b, d.A2 * sizeof(float));
    mat(d.off_out2, w->out2_w, d.O, d.A2, d.A2p); memcpy(p + d.off_bo2, w->out2_b, d.O * sizeof(float));
    // GEMM operands go fragment-major (common.h): one contiguous 1 KB run per wave-wide weight load
    vadx::frag_major_inplace(p + d.off_in1, d.Ap, 400);
    vadx::frag_major_inplace(p + d.off_in2, d.Lp, d.Ap);
    for (int l = 0; l < NLAYER; ++l) {
        vadx::frag_major_inplace(p + d.off_lin[l], PROJ, d.Lp);
        vadx::frag_major_inplace(p + d.off_aff[l], d.Lp, PROJ);
    }
    vadx::frag_major_inplace(p + d.off_out1, d.A2p, d.Lp);
    vadx::frag_major_inplace(p + d.off_out2, d.Op, d.A2p);
    // ---- split-product copies (layers_split.h): A fragments [n-tile][chunk][np planes][QFRAG] from the ORIGINAL row-major weights, in the
    // arithmetic dims->arithmetic names (none for float32 MFMAs)
    float wmax = 0.f;                                                  // largest |weight| handed to fp16 fragments
    auto qmat = [&](int off, int rows, int nch, auto wfn) {            // wfn(row, k) -> weight (0 outside the matrix)
        if (d.np == 0) return;
        for (int nt = 0; nt < (rows + 15) / 16; ++nt)
            for (int kc = 0; kc < nch; ++kc) {
                float *fr = p + off + (size_t)((nt * nch + kc) * d.np) * vadx::QFRAG;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) {
                        if (d.np == 3) vadx::SchemeB3::put_host(fr, i, k, wfn(16 * nt + i, 32 * kc + k), wmax);
                        else vadx::SchemeH2::put_host(fr, i, k, wfn(16 * nt + i, 32 * kc + k), wmax);
                    }
            }
    };
    // in_linear1 with the CMVN folded in (double): W1' = W1 var, b1' = b1 + sum_k W1' (mean_k - mbar_{k % 80}), mbar = the centre LFR position's means
    for (int m = 0; m < NMEL; ++m) p[d.q_mbar + m] = w->cmvn_means[2 * NMEL + m];
    qmat(d.q_in1, d.Ap, NCH_IN1, [&](int r, int k) {
        return (r < d.A && k < 400) ? (float)((double)w->in1_w[(size_t)r * 400 + k] * (double)w->cmvn_vars[k]) : 0.f; });
    for (int r = 0; r < d.A; ++r) {
        double acc = (double)w->in1_b[r];
        for (int k = 0; k < 400; ++k)
            acc += (double)(float)((double)w->in1_w[(size_t)r * 400 + k] * (double)w->cmvn_vars[k]) * ((double)w->cmvn_means[k] - (double)p[d.q_mbar + k % NMEL]);
        p[d.q_b1 + r] = (float)acc;
    }
    qmat(d.q_in2, d.Lp, d.nch_A, [&](int r, int k) { return (r < d.L && k < d.A) ? w->in2_w[(size_t)r * d.A + k] : 0.f; });
    for (int l = 0; l < NLAYER; ++l) {
        qmat(d.q_lin[l], PROJ, d.nch_L, [&](int r, int k) { return k < d.L ? w->lin_w[l][(size_t)r * d.L + k] : 0.f; });
        qmat(d.q_aff[l], d.Lp, PROJ / 32, [&](int r, int k) { return r < d.L ? w->aff_w[l][(size_t)r * PROJ + k] : 0.f; });
    }
    qmat(d.q_out1, d.A2p, d.nch_L, [&](int r, int k) { return (r < d.A2 && k < d.L) ? w->out1_w[(size_t)r * d.L + k] : 0.f; });
    qmat(d.q_out2, d.Op, d.nch_A2, [&](int r, int k) { return (r < d.O && k < d.A2) ? w->out2_w[(size_t)r * d.A2 + k] : 0.f; });
    VADX_REQUIRE(d.arith != vadx::VADX_AR_H2 || wmax <= vadx::H_MAX,
                 "vadx_fsmn_pack_host: a weight (|w| up to %g) is outside the fp16 range: pack with dims->arithmetic = VADX_ARITH_BF16X3", wmax);
    VADX_REQUIRE(d.arith != vadx::VADX_AR_H2 || reb_min >= vadx::REB_REFUSE,
                 "vadx_fsmn_pack_host: a weight tensor lies wholly below 2^%d (largest |w| < 2^%d after rebalancing), outside the fp16 range: pack with "
                 "dims->arithmetic = VADX_ARITH_BF16X3", vadx::REB_REFUSE, reb_min + 1);
    return VADX_OK;
}

// The fp16 x 2 kernels' sticky range flag (as vadx_silero_range_flag): [flag, largest |x|] of the launches since the last reset.
extern "C" int vadx_fsmn_range_flag(const vadx_fsmn_dims *dims, const float *packed, int reset, uint32_t *flag_host, float *amax_host, void *stream) {
    Dev d;
    VADX_REQUIRE(dims && packed && flag_host, "vadx_fsmn_range_flag: NULL argument");
    VADX_REQUIRE(derive(dims, &d) == 0, "vadx_fsmn_range_flag: unsupported dims");
    uint32_t w[2] = {0, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    VADX_HIP_TRY(hipMemcpyAsync(w, packed + d.off_flag, sizeof(w), hipMemcpyDeviceToHost, st));
    VADX_HIP_TRY(hipStreamSynchronize(st));
    if (reset && (w[0] | w[1])) VADX_HIP_TRY(hipMemsetAsync(const_cast<float *>(packed) + d.off_flag, 0, sizeof(w), st));
    *flag_host = w[0];
    if (amax_host) memcpy(amax_host, &w[1], sizeof(float));
    return VADX_OK;
}

static int set_lds_attr() {
    VADX_DYN_LDS(fsmn_run_kernel<0>, LDS_FLOATS * sizeof(float));
    VADX_DYN_LDS(fsmn_clips_kernel<0>, LDS_FLOATS * sizeof(float));
    VADX_DYN_LDS(fsmn_run_kernel<1>, SQ_LDS_BYTES);
    VADX_DYN_LDS(fsmn_clips_kernel<1>, SQ_LDS_BYTES);
    VADX_DYN_LDS(fsmn_run_kernel<2>, SQ_LDS_BYTES);
    VADX_DYN_LDS(fsmn_clips_kernel<2>, SQ_LDS_BYTES);
    return VADX_OK;
}

extern "C" int vadx_fsmn_energy(const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch,
                                int windows_per_clip, int window_len, int frames, const float *means, float *db,
                                void *stream) {
    VADX_REQUIRE(audio && means && db, "vadx_fsmn_energy: NULL argument");
    VADX_REQUIRE(batch > 0 && windows_per_clip > 0 && window_len >= 512 && frames > 0, "vadx_fsmn_energy: bad shape");
    const float inv_ref = (float)(1.0 / (sqrt((double)window_len) * 2e-5));
    hipLaunchKernelGGL(fsmn_energy_kernel, dim3((unsigned)(batch * windows_per_clip)), dim3(512), 0,
                       static_cast<hipStream_t>(stream), audio, (long long)row_stride, (long long)win_stride,
                       windows_per_clip, window_len, 512, 160, frames, means, inv_ref, db);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

extern "C" int vadx_fsmn_window_stats(const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch, int windows_per_clip,
                                      int window_len, int frames, float *means, float *db, void *stream) {
    VADX_REQUIRE(audio && means && db, "vadx_fsmn_window_stats: NULL argument");
    VADX_REQUIRE(batch > 0 && windows_per_clip > 0 && window_len >= 512 && frames > 0, "vadx_fsmn_window_stats: bad shape");
    const float inv_ref = (float)(1.0 / (sqrt((double)window_len) * 2e-5));
    const long long nwin = (long long)batch * windows_per_clip;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the one-pass kernel reads 16-byte runs: every window must start on a 16-byte boundary and hold a whole number of 32-sample granules
    const bool aligned = (reinterpret_cast<uintptr_t>(audio) & 15) == 0 && (row_stride & 7) == 0 && (win_stride & 7) == 0 &&
                         window_len % 32 == 0 && window_len <= ST_THREADS * ST_MAXIT * 8;
    if (aligned) {
        hipLaunchKernelGGL(fsmn_stats_kernel, dim3((unsigned)nwin), dim3(ST_THREADS), 0, st, audio, (long long)row_stride, (long long)win_stride,
                           windows_per_clip, window_len, frames, inv_ref, means, db);
        VADX_HIP_TRY(hipGetLastError());
        return VADX_OK;
    }
    int rc = vadx_frontend_window_means(audio, row_stride, win_stride, batch, windows_per_clip, window_len, 1.0f, means, stream);
    if (rc != VADX_OK) return rc;
    return vadx_fsmn_energy(audio, row_stride, win_stride, batch, windows_per_clip, window_len, frames, means, db, stream);
}

extern "C" int vadx_fsmn_run(const vadx_fsmn_dims *dims, const float *packed, const float *logmel, const float *db,
                             const float *const cache_in[4], float *const cache_out[4], const float *thr,
                             const float *noise_db, int batch, uint8_t *score, float *noisy_db, float *psil,
                             void *stream) {
    Dev d;
    VADX_REQUIRE(dims && packed && logmel && db && cache_in && cache_out && thr && noise_db && score && noisy_db,
                 "vadx_fsmn_run: NULL argument");
    VADX_REQUIRE(derive(dims, &d) == 0, "vadx_fsmn_run: unsupported dims");
    VADX_REQUIRE(batch > 0, "vadx_fsmn_run: batch must be positive");
    int rc = set_lds_attr();
    if (rc) return rc;
    RunArgs r;
    r.logmel = logmel; r.db = db; r.thr = thr; r.noise_db = noise_db; r.score = score; r.noisy_db = noisy_db; r.psil = psil;
    for (int l = 0; l < NLAYER; ++l) {
        VADX_REQUIRE(cache_in[l] && cache_out[l], "vadx_fsmn_run: NULL cache %d", l);
        r.cin[l] = cache_in[l]; r.cout[l] = cache_out[l];
    }
    if (d.arith == vadx::VADX_AR_H2)
        hipLaunchKernelGGL(fsmn_run_kernel<2>, dim3(batch), dim3(THREADS), SQ_LDS_BYTES, static_cast<hipStream_t>(stream), d, packed, r);
    else if (d.arith == vadx::VADX_AR_B3)
        hipLaunchKernelGGL(fsmn_run_kernel<1>, dim3(batch), dim3(THREADS), SQ_LDS_BYTES, static_cast<hipStream_t>(stream), d, packed, r);
    else
        hipLaunchKernelGGL(fsmn_run_kernel<0>, dim3(batch), dim3(THREADS), LDS_FLOATS * sizeof(float),
                           static_cast<hipStream_t>(stream), d, packed, r);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

extern "C" int vadx_fsmn_clips(const vadx_fsmn_dims *dims, const float *packed, const float *logmel, const float *db,
                               int batch, int windows_per_clip, const vadx_fsmn_loop_params *lp, float *cache_ws,
                               uint8_t *flags, float *noise_trace, void *stream) {
    Dev d;
    VADX_REQUIRE(dims && packed && logmel && db && lp && cache_ws && flags, "vadx_fsmn_clips: NULL argument");
    VADX_REQUIRE(derive(dims, &d) == 0, "vadx_fsmn_clips: unsupported dims");
    VADX_REQUIRE(batch > 0 && windows_per_clip > 0, "vadx_fsmn_clips: batch/windows must be positive");
    VADX_REQUIRE(lp->look_backward >= 0 && lp->look_backward < d.T && d.T - lp->look_backward <= 128 && d.T <= 112,
                 "vadx_fsmn_clips: look_backward=%d frames=%d unsupported", lp->look_backward, d.T);
    int rc = set_lds_attr();
    if (rc) return rc;
    ClipArgs c;
    c.logmel = logmel; c.db = db; c.cache = cache_ws; c.W = windows_per_clip;
    // the reference takes slide_range = score_len - look_backward BEFORE it bumps a zero look_backward to 1
    // (Inference_FSMN_VAD_ONNX.py:79-86): LOOK_BACKWARD = 0 means slide_range = T, a vote over one frame, an empty tail
    c.lb = lp->look_backward > 0 ? lp->look_backward : 1;
    c.slide = d.T - lp->look_backward; c.thr = lp->one_minus_speech_threshold; c.noise0 = lp->noise_db_init;
    c.snr = lp->snr_threshold; c.speaking = lp->speaking_score; c.silence_score = lp->silence_score;
    c.flags = flags; c.noise_trace = noise_trace;
    if (d.arith == vadx::VADX_AR_H2)
        hipLaunchKernelGGL(fsmn_clips_kernel<2>, dim3(batch), dim3(THREADS), SQ_LDS_BYTES, static_cast<hipStream_t>(stream), d, packed, c);
    else if (d.arith == vadx::VADX_AR_B3)
        hipLaunchKernelGGL(fsmn_clips_kernel<1>, dim3(batch), dim3(THREADS), SQ_LDS_BYTES, static_cast<hipStream_t>(stream), d, packed, c);
    else
        hipLaunchKernelGGL(fsmn_clips_kernel<0>, dim3(batch), dim3(THREADS), LDS_FLOATS * sizeof(float),
                           static_cast<hipStream_t>(stream), d, packed, c);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
