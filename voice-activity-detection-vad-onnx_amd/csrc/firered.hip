// firered.hip -- FireRedVAD / FireRedAED DetectModel (channel-first DFSMN) for gfx950, plus the
// VadPostprocessor decision kernel shared with MarbleNet.
// Reference: FireRedVAD/Export_FireRedVAD.py:185-326 (FSMN / DFSMNBlock / DFSMN / DetectModel),
// :420-467 (wrapper), FireRedVAD/Inference_FireRed_ONNX.py:102-304 (VadPostprocessor).
//
// One workgroup owns one stateless analysis window (T <= 112 frames, 98 for 16000 samples).
// Activations stay in LDS k-major [channel][frame] for the whole stack:
//   mem [P][T] (block input / FSMN output, updated in place), p [P][T] (FSMN input), h [H][32] tile.
// Pointwise convs = f32-MFMA GEMMs over 32-frame tiles (weights streamed from L2); the dilated
// depthwise look-back / look-ahead FIRs + skip run on the VALU over the whole window once per block.
#include "rebalance.h"
#include "common.h"
#include "layers.h"
#include "split_scheme.h"

#include <math.h>
#include <string.h>

// FR_EXP: development-only what-if switches (bit mask; results are wrong when set): 1 no FIR memory, 2 no barriers
// inside the point-wise pairs, 4 no second (H->P) layer, 8 no first (P->H) layer, 16 no dnn / output section,
// 32 no log-mel staging / zero fills
#ifndef FR_EXP
#define FR_EXP 0
#endif
#define FR_SYNC() do { if (!(FR_EXP & 2)) __syncthreads(); } while (0)

namespace vadx {
namespace firered {

constexpr int THREADS = 512;
constexpr int NMEL = 80, MAX_R = 16, MAX_M = 4, MAX_ODIM = 4, MAX_T = 112;
constexpr int M_LD = 116;               // mem / p row stride (112 + 4)
constexpr int H_LD = 36;                // h tile row stride (32 + 4)
constexpr int MAXP = 128, MAXH = 256;
constexpr int MEM_F = MAXP * M_LD, P_F = MAXP * M_LD, H_F = MAXH * H_LD;
constexpr int LDS_FLOATS = MEM_F + P_F + H_F;

struct Dev {
    int R, M, H, P, N1, S1, N2, S2, odim, T, Hp, Pp;
    int off_fc1, off_fc1b, off_fc2, off_fc2b;
    int off_lb[MAX_R], off_la[MAX_R], off_win[MAX_R], off_bfc1[MAX_R], off_bfc1b[MAX_R], off_bfc2[MAX_R];
    int off_dnn[MAX_M], off_dnnb[MAX_M], off_out, off_outb, total;
    // split-product copies of the point-wise pairs (Hp = 256, Pp = 128 only): A fragments [n-tile][32-k chunk][np planes][QFRAG] of the ONE
    // arithmetic cfg->arithmetic names (split_scheme.h: np = 3 bf16 x 3, 2 fp16 x 2, 0 none = float32 MFMAs)
    int split_ok, arith, np, off_flag, q_fc1, q_fc2, q_bfc1[MAX_R], q_bfc2[MAX_R];
    // M = 1: the dnn layer (P -> H, ReLU) and the output head (H -> odim, rows padded to Pp with zeros) as one more point-wise pair
    int q_dnn0, q_head, off_headb;
};

static int r16(int x) { return (x + 15) & ~15; }

static int derive(const vadx_firered_cfg *c, Dev *d) {
    memset(d, 0, sizeof(*d));
    if (c->idim != NMEL || c->R < 1 || c->R > MAX_R || c->M < 1 || c->M > MAX_M || c->H < 1 || c->P < 1 ||
        c->N1 < 1 || c->N1 > 32 || c->S1 < 1 || c->N2 < 0 || c->N2 > 32 || (c->N2 > 0 && c->S2 < 1) ||
        c->odim < 1 || c->odim > MAX_ODIM || c->frames < 1 || c->frames > MAX_T)
        return -1;
    d->R = c->R; d->M = c->M; d->H = c->H; d->P = c->P; d->N1 = c->N1; d->S1 = c->S1; d->N2 = c->N2; d->S2 = c->S2;
    d->odim = c->odim; d->T = c->frames; d->Hp = r16(c->H); d->Pp = r16(c->P);
    if (d->Hp > MAXH || d->Pp > MAXP) return -1;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 255) & ~255; return r; };      // every section on a 1 KiB boundary (a wave's fragment load = 1 KiB = 16 lines)
    d->off_fc1 = take(d->Hp * NMEL); d->off_fc1b = take(d->Hp);
    d->off_fc2 = take(d->Pp * d->Hp); d->off_fc2b = take(d->Pp);
    for (int r = 0; r < d->R; ++r) {
        d->off_lb[r] = take(d->Pp * d->N1);
        d->off_la[r] = take(d->Pp * (d->N2 > 0 ? d->N2 : 1));
        d->off_win[r] = take(d->Pp * 40);          // merged 40-tap window c[-19..20] per channel (fast FIR path)
        if (r > 0) { d->off_bfc1[r] = take(d->Hp * d->Pp); d->off_bfc1b[r] = take(d->Hp); d->off_bfc2[r] = take(d->Pp * d->Hp); }
    }
    for (int m = 0; m < d->M; ++m) { d->off_dnn[m] = take(d->Hp * (m == 0 ? d->Pp : d->Hp)); d->off_dnnb[m] = take(d->Hp); }
    d->off_out = take(16 * d->Hp); d->off_outb = take(16);      // output head padded to one 16-row MFMA tile
    d->split_ok = d->Hp == 256 && d->Pp == 128;
    // AUTO = bf16 x 3 where the split kernel applies (float32's exponent range: safe without the range protocol), float32 MFMAs otherwise;
    // fp16 x 2 is an explicit request (VADX_ARITH_F16X2) by callers that read vadx_firered_range_flag, as FireRedEngine does; an explicit
    // split arithmetic elsewhere is refused
    d->arith = vadx::arith_internal(c->arithmetic, d->split_ok ? vadx::VADX_AR_B3 : vadx::VADX_AR_F32);
    if (d->arith < 0 || (d->arith != vadx::VADX_AR_F32 && !d->split_ok)) return -1;
    d->np = d->arith == vadx::VADX_AR_B3 ? 3 : (d->arith == vadx::VADX_AR_H2 ? 2 : 0);
    if (d->np) {
        const int np = d->np;
        d->q_fc1 = take(16 * 3 * np * vadx::QFRAG);                // 80 mels -> 3 chunks of 32 (k-groups 10, 11 are zero rows)
        d->q_fc2 = take(8 * 8 * np * vadx::QFRAG);
        for (int r = 1; r < d->R; ++r) { d->q_bfc1[r] = take(16 * 4 * np * vadx::QFRAG); d->q_bfc2[r] = take(8 * 8 * np * vadx::QFRAG); }
        if (d->M == 1) { d->q_dnn0 = take(16 * 4 * np * vadx::QFRAG); d->q_head = take(8 * 8 * np * vadx::QFRAG); d->off_headb = take(d->Pp); }
    }
    d->off_flag = take(4);
    d->total = o;
    return 0;
}

// memory = p + lookback(p) + lookahead(p) (+ mem)   -- Export_FireRedVAD.py:213-236, :255-264
// thread = (channel, quarter of the window); the channel's taps sit in registers (loops fully unrolled
// to MAXTAP with guards: a runtime-indexed tap array would live in scratch and serialise on latency).
constexpr int MAXTAP = 32;

// Fast path for undilated filters with <= 20 taps each (the shipped FireRedVAD/AED configs): the
// look-back and look-ahead filters merge into one 40-tap window c[-19..20] around t; each thread slides
// it over 28 frames held in registers -- no guards, no address math in the inner loop.

__device__ __forceinline__ void fsmn_memory_fast(const Dev &d, const float *__restrict__ Pk, int r, bool skip,
                                                 const float *p, float *mem) {
    // thread = (channel PAIR, eighth of the window): both channels ride in the two halves of v_pk_fma_f32 operands
    // (twice the f32 FMA rate of the scalar form; the per-channel summation order is unchanged).
    constexpr int LB = 19, LA = 20, SEG = 14, WIN = SEG + LB + LA;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));       // per call: the clamped indices / in-range predicates below are invariant over the blocks, and hoisted out
                                        // of the block loop they stayed live through the point-wise pairs and spilled (15 registers, the kernel's scratch)
    const int cp = tid >> 3, part = tid & 7, ch = 2 * cp;
    if (ch >= d.Pp) return;
    f32x2 c[LB + 1 + LA];                       // c[j + LB], j = -19..20 ; c[LB] (j = 0) unused (zero)
    {   const f32x4 *w0 = reinterpret_cast<const f32x4 *>(Pk + d.off_win[r] + ch * 40), *w1 = w0 + 10;      // merged at pack time
#pragma unroll
        for (int g = 0; g < 10; ++g) {
            const f32x4 a4 = w0[g], b4 = w1[g];
#pragma unroll
            for (int e = 0; e < 4; ++e) c[4 * g + e] = f32x2{a4[e], b4[e]};
        }
    }
    const float *row0 = p + ch * M_LD, *row1 = row0 + M_LD;
    const int t0 = part * SEG;
    f32x2 win[WIN];
#pragma unroll
    for (int u = 0; u < WIN; ++u) {
        const int t = t0 - LB + u;
        const bool in = t >= 0 && t < d.T;
        const int tc = t < 0 ? 0 : (t > MAX_T - 1 ? MAX_T - 1 : t);      // unconditional loads (clamped) + select:
        const float v0 = row0[tc], v1 = row1[tc];                          // a guarded load costs a branch and a wait each
        win[u] = f32x2{in ? v0 : 0.f, in ? v1 : 0.f};
    }
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
        const int t = t0 + u;
        f32x2 lb = {0.f, 0.f}, la = {0.f, 0.f};
#pragma unroll
        for (int j = -LB; j <= 0; ++j) lb = __builtin_elementwise_fma(c[j + LB], win[u + LB + j], lb);      // same order as the generic path
#pragma unroll
        for (int j = 1; j <= LA; ++j) la = __builtin_elementwise_fma(c[j + LB], win[u + LB + j], la);
        // frames t >= T (< MAX_T) are inside the row: computing / storing them unconditionally is harmless (the next
        // layer's columns are independent and p is zero there) and keeps the loop branch-free
        f32x2 s2 = (win[u + LB] + lb) + la;
        if (skip) s2 += f32x2{mem[ch * M_LD + t], mem[(ch + 1) * M_LD + t]};
        mem[ch * M_LD + t] = s2[0];
        mem[(ch + 1) * M_LD + t] = s2[1];
    }
}

__device__ __forceinline__ void fsmn_memory_generic(const Dev &d, const float *__restrict__ Pk, int r, bool skip,
                                            const float *p, float *mem) {
    const int ch = threadIdx.x >> 2, part = threadIdx.x & 3;
    if (ch >= d.Pp) return;
    float wlb[MAXTAP], wla[MAXTAP];
#pragma unroll
    for (int k = 0; k < MAXTAP; ++k) {
        wlb[k] = k < d.N1 ? Pk[d.off_lb[r] + ch * d.N1 + k] : 0.f;
        wla[k] = k < d.N2 ? Pk[d.off_la[r] + ch * d.N2 + k] : 0.f;
    }
    const float *row = p + ch * M_LD;
    const int per = (d.T + 3) >> 2, t0 = part * per, t1 = (t0 + per < d.T) ? t0 + per : d.T;
    const bool ahead = d.N2 > 0 && d.T > 1;
    for (int t = t0; t < t1; ++t) {
        float lb = 0.f, la = 0.f;
#pragma unroll
        for (int k = 0; k < MAXTAP; ++k) {
            const int idx = t + (k - (d.N1 - 1)) * d.S1;
            if (k < d.N1 && idx >= 0) lb = fmaf(wlb[k], row[idx], lb);
        }
        float s = row[t] + lb;
        if (ahead) {
#pragma unroll
            for (int k = 0; k < MAXTAP; ++k) {
                const int idx = t + d.S2 + k * d.S2;
                if (k < d.N2 && idx < d.T) la = fmaf(wla[k], row[idx], la);
            }
            s += la;
        }
        if (skip) s += mem[ch * M_LD + t];
        mem[ch * M_LD + t] = s;
    }
}

__device__ __forceinline__ void fsmn_memory(const Dev &d, const float *__restrict__ Pk, int r, bool skip, const float *p, float *mem) {
    if (FR_EXP & 1) return;
    if (d.S1 == 1 && d.N1 <= 20 && (d.N2 == 0 || d.S2 == 1) && d.N2 <= 20 && d.T > 1 && d.T <= 112 && d.Pp <= 128) fsmn_memory_fast(d, Pk, r, skip, p, mem);
    else fsmn_memory_generic(d, Pk, r, skip, p, mem);
}

// pointwise pair on every 32-frame tile of the window: h = relu(W1 x src + b1); dst = W2 x h (+b2, relu)
// The DFSMN block's P -> H -> P pair at the published widths (H = 256: one pair of n-tiles per wave; P = 128: one
// n-tile per wave; 8 waves).  A wave's slice of BOTH matrices -- 8 blocks x 2 tiles of W1, 16 blocks x 1 tile of W2,
// 128 VGPRs -- is loaded once per block and stays in registers while the window's tiles stream through: no weight
// traffic inside the pair (it was re-streamed from L2 for every 32-frame tile, and every phase opened with an L2 round
// trip that nobody else on the CU could cover: one workgroup per CU, all waves at the same barrier).  Bias / ReLU /
// LDS store as in layer<>.  MT = 2 for a full 32-frame tile, 1 for a last tile of <= 16 frames.
template <int KB1>
struct PairResident {
    f32x4 w1[KB1][2], w2[16][1];
    float bias1[2], bias2;
};

// HAS2: the second (H -> P) layer exists; otherwise the H tile in `h` is the result (dnn layer before the output head)
template <int MT, int KB1, bool HAS2>
__device__ __forceinline__ void pair_tile_resident(const PairResident<KB1> &R, const float *src, float *dst, float *h, int f0, bool relu2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, i = lane & 15;
    int moff[MT], hoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { moff[mt] = f0 + 16 * mt; hoff[mt] = 16 * mt; }
    {   f32x4 acc[2][MT];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_resident<2, MT, KB1>(acc, src, M_LD, moff, lane, R.w1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[nt][mt][r] + R.bias1[nt], 0.f);
                *reinterpret_cast<f32x4 *>(h + ((wave + 8 * nt) * 16 + i) * H_LD + mt * 16 + 4 * q) = v;
            }
    }
    __syncthreads();
    if (HAS2) {
        f32x4 acc[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_resident<1, MT, 16>(acc, h, H_LD, hoff, lane, R.w2);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[0][mt][r] + R.bias2; if (relu2) v[r] = fmaxf(v[r], 0.f); }
            *reinterpret_cast<f32x4 *>(dst + (wave * 16 + i) * M_LD + f0 + mt * 16 + 4 * q) = v;
        }
        __syncthreads();
    }
}

template <int KB1, bool HAS2>
__device__ __forceinline__ void load_pair_resident(PairResident<KB1> &R, const float *W1, const float *b1, const float *W2, const float *b2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15;
    const float *row10 = frag_ptr(W1, KB1 * 16, wave, 0, lane), *row11 = frag_ptr(W1, KB1 * 16, wave + 8, 0, lane);
#pragma unroll
    for (int S = 0; S < KB1; ++S) { R.w1[S][0] = ldg4(row10 + FRAG * S); R.w1[S][1] = ldg4(row11 + FRAG * S); }
    if (HAS2) {
        const float *row2 = frag_ptr(W2, 256, wave, 0, lane);
#pragma unroll
        for (int S = 0; S < 16; ++S) R.w2[S][0] = ldg4(row2 + FRAG * S);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) R.bias1[nt] = ldg1(b1 + (wave + 8 * nt) * 16 + i);
    R.bias2 = (HAS2 && b2) ? ldg1(b2 + wave * 16 + i) : 0.f;
}

template <int KB1>
__device__ __forceinline__ void pointwise_pair_resident(const Dev &d, const float *W1, const float *b1, const float *src,
                                                        const float *W2, const float *b2, bool relu2, float *dst, float *h) {
    PairResident<KB1> R;
    load_pair_resident<KB1, true>(R, W1, b1, W2, b2);
    for (int f0 = 0; f0 < d.T; f0 += 32) {
        if (d.T - f0 <= 16) pair_tile_resident<1, KB1, true>(R, src, dst, h, f0, relu2);
        else pair_tile_resident<2, KB1, true>(R, src, dst, h, f0, relu2);
    }
}


// ---- the point-wise pair on bf16 x 3 split products (csrc/split3.h), published widths (H = 256, P = 128, 8 waves) ----------------------
// Same data flow as pointwise_pair_resident -- src and dst stay float32 [channel][frame] (the FIR memory works on them), a wave's slice of
// BOTH matrices is resident for the whole window (192 VGPRs of bf16 fragments) -- but the tile is 16 frames and each GEMM runs as six
// v_mfma_f32_16x16x32_bf16 per K = 32 step: 96 bf16 MFMAs per wave and tile instead of 256 f32 ones (6/16 of the matrix time).
// Per tile: the tile's src columns are split into three bf16 planes [k / 8][16 frames][8] (512 items, one per thread) | barrier | GEMM 1,
// weights = A operand: a lane ends with four consecutive H channels of one frame -> bias, ReLU, split, one 8-byte store per plane of the
// H tile | barrier | GEMM 2, activations = A operand: a lane ends with four consecutive frames of one P channel -> one float4 store into
// dst.  The planes live where the f32 kernel keeps its 32-frame H tile (12 KB + 24 KB = its 36 KB).
constexpr int QS_PL_SRC = 16 * 16 * 16, QS_PL_H = 32 * 16 * 16;     // bytes of one plane: src tile (<= 128 ch x 16 frames), H tile (256 x 16)
template <typename SC, int KC1>
struct PairSplit { typename SC::frag w1[2][KC1][SC::NP], w2[8][SC::NP]; f32x4 bias1[2]; float bias2; };

template <typename SC, int KC1>
__device__ __forceinline__ void load_pair_split(PairSplit<SC, KC1> &R, const float *Q1, const float *b1, const float *Q2, const float *b2, int tid) {
    constexpr int NP = SC::NP;
    const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, i = lane & 15;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kc = 0; kc < KC1; ++kc)
#pragma unroll
            for (int p = 0; p < NP; ++p) R.w1[nt][kc][p] = SC::ld(Q1 + (size_t)(((wave + 8 * nt) * KC1 + kc) * NP + p) * QFRAG, lane);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc)
#pragma unroll
        for (int p = 0; p < NP; ++p) R.w2[kc][p] = SC::ld(Q2 + (size_t)((wave * 8 + kc) * NP + p) * QFRAG, lane);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) R.bias1[nt] = ldg4(b1 + (wave + 8 * nt) * 16 + 4 * q);
    R.bias2 = b2 ? ldg1(b2 + wave * 16 + i) : 0.f;
}

template <typename SC, int KC1>
__device__ __forceinline__ void pointwise_pair_split(const Dev &d, const float *Q1, const float *b1, int ksrc, const float *src,
                                                     const float *Q2, const float *b2, bool relu2, float *dst, float *hbuf, float &amax) {
    constexpr int NP = SC::NP;
    typedef typename SC::frag frag;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));       // per call: the per-lane LDS offsets below are loop-invariant over the blocks, and hoisted out of the block
                                        // loop they stayed live through the FIR (186 registers of its own) and spilled there (layers.h has the same note)
    const int lane = tid & 63, wave = tid >> 6, q = lane >> 4, i = lane & 15;
    unsigned char *SQ = reinterpret_cast<unsigned char *>(hbuf), *HQ = SQ + 3 * QS_PL_SRC;       // (the plane map is sized for three planes; fp16 x 2 uses two of each)
    PairSplit<SC, KC1> R;
    load_pair_split<SC, KC1>(R, Q1, b1, Q2, b2, tid);
    if (ksrc < KC1 * 32) {             // k-groups beyond the source's channels (80 mels in 96 slots): rows of zeros, written once
        const int kg0 = ksrc / 8, n = (KC1 * 4 - kg0) * 16;
        for (int e = tid; e < NP * n; e += THREADS) {
            const int p = e / n, r = e - p * n;
            *reinterpret_cast<f32x4 *>(SQ + p * QS_PL_SRC + (kg0 * 16 + r) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int cg = tid >> 4, fr = tid & 15;
    auto split_tile = [&](int f0) {    // this thread's item: channels 4 cg .. 4 cg + 3 of frame f0 + fr -> the three planes of the src tile
        if (4 * cg < ksrc) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = src[(4 * cg + c) * M_LD + f0 + fr];
            u32x2 pp[NP];
            SC::split4(v, pp, amax);
            unsigned char *dp = SQ + ((cg >> 1) * 16 + fr) * 16 + (cg & 1) * 8;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2 *>(dp + p * QS_PL_SRC) = pp[p];
        }
    };
    split_tile(0);
    __syncthreads();
    for (int f0 = 0; f0 < d.T; f0 += 16) {
        {   // GEMM 1: H channels 16 (wave + 8 nt) + 4 q + r of frame i
            f32x4 hi[2][1], lo[2][1];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { hi[nt][0] = R.bias1[nt]; lo[nt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int kc = 0; kc < KC1; ++kc) {
                frag b[1][NP], a[2][NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    b[0][p] = SC::lds(SQ + p * QS_PL_SRC + ((4 * kc + q) * 16 + i) * 16);
                    a[0][p] = R.w1[0][kc][p];
                    a[1][p] = R.w1[1][kc][p];
                }
                SC::template products<2, 1, true>(a, b, hi, lo);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 v = SC::join(hi[nt][0], lo[nt][0]);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                u32x2 pp[NP];
                SC::split4(v, pp, amax);
                const int g = 4 * (wave + 8 * nt) + q;
                unsigned char *dp = HQ + ((g >> 1) * 16 + i) * 16 + (g & 1) * 8;
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2 *>(dp + p * QS_PL_H) = pp[p];
            }
        }
        __syncthreads();
        // the NEXT tile's split rides beside GEMM 2: GEMM 1 is done with the src planes (barrier above), GEMM 2 reads only the H tile, and
        // the barrier at the end of the iteration publishes both the new src planes and the free H tile
        if (f0 + 16 < d.T) split_tile(f0 + 16);
        {   // GEMM 2 (activations = A operand): frames f0 + 4 q + r of P channel 16 wave + i
            f32x4 hi[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}}, lo[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                frag b[1][NP], a[1][NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    b[0][p] = SC::lds(HQ + p * QS_PL_H + ((4 * kc + q) * 16 + i) * 16);
                    a[0][p] = R.w2[kc][p];
                }
                SC::template products<1, 1, false>(a, b, hi, lo);
            }
            f32x4 v = SC::join(hi[0][0], lo[0][0]);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = v[r] + R.bias2; if (relu2) v[r] = fmaxf(v[r], 0.f); }
            *reinterpret_cast<f32x4 *>(dst + (wave * 16 + i) * M_LD + f0 + 4 * q) = v;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void pointwise_pair(const Dev &d, const float *W1, const float *b1, int k1b, const float *src,
                                               const float *W2, const float *b2, int relu2, float *dst, float *h) {
    if (!FR_EXP && W2 && d.Hp == 256 && d.Pp == 128 && blockDim.x == 512) {
        if (k1b == 8) { pointwise_pair_resident<8>(d, W1, b1, src, W2, b2, relu2 != 0, dst, h); return; }      // DFSMN block
        if (k1b == 5) { pointwise_pair_resident<5>(d, W1, b1, src, W2, b2, relu2 != 0, dst, h); return; }      // fc1 (80 mels) / fc2
    }
    for (int f0 = 0; f0 < d.T; f0 += 32) {
        const bool half = (d.T - f0) <= 16;
        LayerArgs a{W1, k1b * 16, d.Hp / 16, 1, k1b, 0, 0, b1, 1, src, M_LD, f0, h, H_LD, 0, nullptr, nullptr};
        if (!(FR_EXP & 8)) { if (half) layer<1, false>(a); else layer<2, false>(a); }
        FR_SYNC();
        if (W2 && !(FR_EXP & 4)) {
            LayerArgs c{W2, d.Hp, d.Pp / 16, 1, d.Hp / 16, 0, 0, b2, relu2, h, H_LD, 0, dst, M_LD, f0, nullptr, nullptr};
            if (half) layer<1, false>(c); else layer<2, false>(c);
            FR_SYNC();
        }
    }
}

template <int AR> struct FrSchemeOf { typedef vadx::SchemeB3 type; };
template <> struct FrSchemeOf<vadx::VADX_AR_H2> { typedef vadx::SchemeH2 type; };
template <int AR>
__global__ __launch_bounds__(THREADS, 2) void firered_kernel(Dev d, const float *__restrict__ Pk,
                                                             const float *__restrict__ logmel, float *__restrict__ probs) {
    constexpr bool SPLIT = AR != vadx::VADX_AR_F32;
    typedef typename FrSchemeOf<AR>::type SC;
    float amax = 0.f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *mem = lds, *p = lds + MEM_F, *h = p + P_F;
    const int tid = threadIdx.x;
    long long tk0 = clock64(), tk_fir = 0, tk_pw = 0, tk_x;      // FR_EXP & 64: cycle accounting (development)
    const float *lm = logmel + (size_t)blockIdx.x * d.T * NMEL;
    // stage log-mel channel-first into `mem` rows 0..79; frames >= T are zero (finite operands)
    for (int e = tid; e < ((FR_EXP & 32) ? 0 : MAX_T * NMEL); e += THREADS) {
        const int t = e / NMEL, mel = e - t * NMEL;
        const float v = lm[(size_t)(t < d.T ? t : d.T - 1) * NMEL + mel];      // unconditional (clamped) load, then select
        mem[mel * M_LD + t] = t < d.T ? v : 0.f;
    }
    for (int e = tid; e < ((FR_EXP & 32) ? 0 : MAXP * M_LD); e += THREADS) p[e] = 0.f;
    __syncthreads();
    // dfsmn.fc1 (80->H, ReLU) ; dfsmn.fc2 (H->P, bias, ReLU) ; fsmn1
    if (SPLIT) pointwise_pair_split<SC, 3>(d, Pk + d.q_fc1, Pk + d.off_fc1b, NMEL, mem, Pk + d.q_fc2, Pk + d.off_fc2b, true, p, h, amax);
    else pointwise_pair(d, Pk + d.off_fc1, Pk + d.off_fc1b, NMEL / 16, mem, Pk + d.off_fc2, Pk + d.off_fc2b, 1, p, h);
    for (int e = tid; e < ((FR_EXP & 32) ? 0 : MAXP * M_LD); e += THREADS) mem[e] = 0.f;        // log-mel rows are dead now
    __syncthreads();
    tk_x = clock64();
    fsmn_memory(d, Pk, 0, false, p, mem);
    __syncthreads();
    tk_fir += clock64() - tk_x;
    for (int r = 1; r < d.R; ++r) {      // DFSMNBlock: fc1 (P->H, ReLU) ; fc2 (H->P, no bias) ; fsmn + skip
        tk_x = clock64();
        if (SPLIT) pointwise_pair_split<SC, 4>(d, Pk + d.q_bfc1[r], Pk + d.off_bfc1b[r], d.Pp, mem, Pk + d.q_bfc2[r], nullptr, false, p, h, amax);
        else pointwise_pair(d, Pk + d.off_bfc1[r], Pk + d.off_bfc1b[r], d.Pp / 16, mem, Pk + d.off_bfc2[r], nullptr, 0, p, h);
        tk_pw += clock64() - tk_x;
        tk_x = clock64();
        fsmn_memory(d, Pk, r, true, p, mem);
        __syncthreads();
        tk_fir += clock64() - tk_x;
    }
    // One dnn layer (the published configuration): dnn (P -> H, ReLU) + output head (H -> odim) are one more point-wise pair on split
    // products -- the head's rows padded to Pp with zeros (as many MFMAs as a block's second layer, at a third of the f32 MFMAs' cost) --
    // over the whole window at once instead of tile by tile on float32 MFMAs; logits land in p[o][t].
#ifndef FR_OLD_DNN
#define FR_OLD_DNN 0        /* development: 1 = the float32 tile-by-tile dnn + head (A/B timing) */
#endif
    if (SPLIT && d.M == 1 && !FR_EXP && !FR_OLD_DNN) {
        pointwise_pair_split<SC, 4>(d, Pk + d.q_dnn0, Pk + d.off_dnnb[0], d.Pp, mem, Pk + d.q_head, Pk + d.off_headb, false, p, h, amax);
        __syncthreads();
        for (int e = tid; e < d.odim * d.T; e += THREADS) {
            const int o = e / d.T, t = e - o * d.T;
            probs[((size_t)blockIdx.x * d.odim + o) * d.T + t] = sigmoidf_(p[o * M_LD + t]);
        }
        if (AR == vadx::VADX_AR_H2) vadx::range_flag_raise(Pk + d.off_flag, amax);
        return;
    }
    // dnns (P->H ReLU, then M-1 x H->H ReLU) and the 1x1 output conv + sigmoid, tile by tile
    float *h2 = p;                        // p is dead: second H-tile buffer for M > 1
    const bool dnn_resident = !FR_EXP && d.Hp == 256 && d.Pp == 128 && blockDim.x == 512;
    PairResident<8> RD;                   // dnn[0] (P -> H) stays in registers across the window's tiles, like the pairs
    if (dnn_resident) load_pair_resident<8, false>(RD, Pk + d.off_dnn[0], Pk + d.off_dnnb[0], nullptr, nullptr);
    for (int f0 = 0; f0 < ((FR_EXP & 16) ? 0 : d.T); f0 += 32) {
        const bool half = (d.T - f0) <= 16;
        if (dnn_resident) {
            if (half) pair_tile_resident<1, 8, false>(RD, mem, nullptr, h, f0, false);
            else pair_tile_resident<2, 8, false>(RD, mem, nullptr, h, f0, false);
        } else {
            LayerArgs a{Pk + d.off_dnn[0], d.Pp, d.Hp / 16, 1, d.Pp / 16, 0, 0, Pk + d.off_dnnb[0], 1, mem, M_LD, f0, h, H_LD, 0, nullptr, nullptr};
            if (half) layer<1, false>(a); else layer<2, false>(a);
            __syncthreads();
        }
        float *cur = h, *nxt = h2;
        for (int m = 1; m < d.M; ++m) {
            LayerArgs c{Pk + d.off_dnn[m], d.Hp, d.Hp / 16, 1, d.Hp / 16, 0, 0, Pk + d.off_dnnb[m], 1, cur, H_LD, 0, nxt, H_LD, 0, nullptr, nullptr};
            if (half) layer<1, false>(c); else layer<2, false>(c);
            __syncthreads();
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        // output head (odim <= 4 rows, zero padded to one MFMA tile) -> logits in `lg`, then sigmoid
        float *lg = p + MAXH * H_LD;            // behind h2
        {   LayerArgs o{Pk + d.off_out, d.Hp, 1, 1, d.Hp / 16, 0, 0, Pk + d.off_outb, 0, cur, H_LD, 0, lg, H_LD, 0, nullptr, nullptr};
            if (half) layer<1, false>(o); else layer<2, false>(o); }
        __syncthreads();
        if (tid < 32 * d.odim) {
            const int t = tid & 31, o = tid >> 5;
            if (f0 + t < d.T) probs[((size_t)blockIdx.x * d.odim + o) * d.T + f0 + t] = sigmoidf_(lg[o * H_LD + t]);
        }
        __syncthreads();
    }
    if ((FR_EXP & 64) && tid == 0) {
        float *dbg = probs + (size_t)blockIdx.x * d.odim * d.T;
        dbg[0] = (float)tk_fir; dbg[1] = (float)tk_pw; dbg[2] = (float)(clock64() - tk0);
    }
    if (AR == vadx::VADX_AR_H2) vadx::range_flag_raise(Pk + d.off_flag, amax);
}

// ---- streaming variant (FireRedVAD/Export_FireRedVAD.py:479-612): no look-ahead, the look-back context
// of every FSMN comes from an explicit cache [R][B][P][(N1-1)*S1] that is returned updated ---------------
__device__ __forceinline__ void fsmn_memory_stream(const Dev &d, const float *__restrict__ Pk, int r, bool skip,
                                                   const float *p, float *mem, const float *__restrict__ cin,
                                                   float *__restrict__ cout) {
    const int pad = (d.N1 - 1) * d.S1;
    for (int e = threadIdx.x; e < d.P * d.T; e += THREADS) {
        const int ch = e / d.T, t = e - ch * d.T;
        const float *row = p + ch * M_LD, *crow = cin + (size_t)ch * pad, *w = Pk + d.off_lb[r] + ch * d.N1;
        float lb = 0.f;
        for (int k = 0; k < d.N1; ++k) {
            const int j = t + k * d.S1;                       // index into cache ++ p
            lb = fmaf(w[k], j < pad ? crow[j] : row[j - pad], lb);
        }
        float s2 = row[t] + lb;
        if (skip) s2 += mem[ch * M_LD + t];
        mem[ch * M_LD + t] = s2;
    }
    for (int e = threadIdx.x; e < d.P * pad; e += THREADS) {      // new cache = last `pad` entries of cache ++ p
        const int ch = e / pad, j = e - ch * pad, src = d.T + j;
        cout[e] = src < pad ? cin[(size_t)ch * pad + src] : p[ch * M_LD + src - pad];
    }
}

__global__ __launch_bounds__(THREADS, 2) void firered_stream_kernel(Dev d, const float *__restrict__ Pk,
                                                                    const float *__restrict__ logmel,
                                                                    const float *__restrict__ caches_in,
                                                                    float *__restrict__ caches_out, int B,
                                                                    float *__restrict__ probs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *mem = lds, *p = lds + MEM_F, *h = p + P_F;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int pad = (d.N1 - 1) * d.S1;
    const float *lm = logmel + (size_t)b * d.T * NMEL;
    for (int e = tid; e < MAX_T * NMEL; e += THREADS) {
        const int t = e / NMEL, mel = e - t * NMEL;
        const float v = lm[(size_t)(t < d.T ? t : d.T - 1) * NMEL + mel];      // unconditional (clamped) load, then select
        mem[mel * M_LD + t] = t < d.T ? v : 0.f;
    }
    for (int e = tid; e < MAXP * M_LD; e += THREADS) p[e] = 0.f;
    __syncthreads();
    pointwise_pair(d, Pk + d.off_fc1, Pk + d.off_fc1b, NMEL / 16, mem, Pk + d.off_fc2, Pk + d.off_fc2b, 1, p, h);
    for (int e = tid; e < MAXP * M_LD; e += THREADS) mem[e] = 0.f;
    __syncthreads();
    auto cache = [&](const float *base, int r) { return base + ((size_t)r * B + b) * d.P * pad; };
    fsmn_memory_stream(d, Pk, 0, false, p, mem, cache(caches_in, 0), const_cast<float *>(cache(caches_out, 0)));
    __syncthreads();
    for (int r = 1; r < d.R; ++r) {
        pointwise_pair(d, Pk + d.off_bfc1[r], Pk + d.off_bfc1b[r], d.Pp / 16, mem, Pk + d.off_bfc2[r], nullptr, 0, p, h);
        fsmn_memory_stream(d, Pk, r, true, p, mem, cache(caches_in, r), const_cast<float *>(cache(caches_out, r)));
        __syncthreads();
    }
    float *h2 = p;
    for (int f0 = 0; f0 < d.T; f0 += 32) {
        const bool half = (d.T - f0) <= 16;
        LayerArgs a{Pk + d.off_dnn[0], d.Pp, d.Hp / 16, 1, d.Pp / 16, 0, 0, Pk + d.off_dnnb[0], 1, mem, M_LD, f0, h, H_LD, 0, nullptr, nullptr};
        if (half) layer<1, false>(a); else layer<2, false>(a);
        __syncthreads();
        float *cur = h, *nxt = h2;
        for (int m = 1; m < d.M; ++m) {
            LayerArgs c{Pk + d.off_dnn[m], d.Hp, d.Hp / 16, 1, d.Hp / 16, 0, 0, Pk + d.off_dnnb[m], 1, cur, H_LD, 0, nxt, H_LD, 0, nullptr, nullptr};
            if (half) layer<1, false>(c); else layer<2, false>(c);
            __syncthreads();
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        float *lg = p + MAXH * H_LD;            // behind h2
        {   LayerArgs o{Pk + d.off_out, d.Hp, 1, 1, d.Hp / 16, 0, 0, Pk + d.off_outb, 0, cur, H_LD, 0, lg, H_LD, 0, nullptr, nullptr};
            if (half) layer<1, false>(o); else layer<2, false>(o); }
        __syncthreads();
        if (tid < 32 * d.odim) {
            const int t = tid & 31, o = tid >> 5;
            if (f0 + t < d.T) probs[((size_t)b * d.odim + o) * d.T + f0 + t] = sigmoidf_(lg[o * H_LD + t]);
        }
        __syncthreads();
    }
}

// ---- VadPostprocessor: one clip per thread, working arrays laid out [frame][clip] (coalesced) -------
struct PostDev {
    int ws, min_sp, max_sp, min_si, merge, extend;
    float thr, inv_ws;
};

__global__ void vadpost_kernel(PostDev q, const float *__restrict__ probs, int stride, const int *__restrict__ nframes,
                               int B, float *__restrict__ wsm, signed char *__restrict__ wdec,
                               signed char *__restrict__ decisions, int *__restrict__ segs, int *__restrict__ counts, int cap) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int n = nframes[b];
    const float *pr = probs + (size_t)b * stride;
#define SM(i) wsm[(size_t)(i) * B + b]
#define DEC(i) wdec[(size_t)(i) * B + b]
    counts[b] = 0;
    if (n <= 0) return;
    // smoothing: float32 running sum exactly like np.cumsum, expanding mean on the first ws-1 frames
    if (q.ws > 1) {
        float cs = 0.f;
        for (int i = 0; i < n; ++i) { cs = __fadd_rn(cs, pr[i]); SM(i) = cs; }      // SM(i) = cumsum[i+1]
        for (int i = n - 1; i >= 0; --i) {
            float v;
            if (i < q.ws - 1) v = __fdiv_rn(SM(i), (float)(i + 1));
            else v = __fmul_rn(__fsub_rn(SM(i), (i - q.ws >= 0) ? SM(i - q.ws) : 0.f), q.inv_ws);
            SM(i) = v;                    // descending order: SM(i - ws) is still a cumsum when read
        }
    } else {
        for (int i = 0; i < n; ++i) SM(i) = pr[i];
    }
    // threshold + 4-state machine
    if (q.min_sp <= 0 && q.min_si <= 0) {
        for (int t = 0; t < n; ++t) DEC(t) = SM(t) >= q.thr ? 1 : 0;
    } else {
        int state = 0, t0 = 0, s0 = 0;
        for (int t = 0; t < n; ++t) {
            const bool hot = SM(t) >= q.thr;
            if (state == 0) { if (hot) { state = 1; t0 = t; } }
            else if (state == 1) {
                if (hot) { if (t - t0 >= q.min_sp) { state = 2; for (int u = t0; u < t; ++u) DEC(u) = 1; } }
                else state = 0;
            } else if (state == 2) { if (!hot) { state = 3; s0 = t; } }
            else { if (!hot) { if (t - s0 >= q.min_si) state = 0; } else state = 2; }
            DEC(t) = state >= 2 ? 1 : 0;
        }
    }
    // extend each rising edge left by the smoothing window
    if (q.ws > 1)
        for (int t = 1; t < n; ++t)
            if (DEC(t) == 1 && DEC(t - 1) == 0) { for (int u = (t >= q.ws ? t - q.ws : 0); u < t; ++u) DEC(u) = 1; }
    if (q.merge > 0) {
        int g0 = -1;
        for (int t = 1; t < n; ++t) {
            const int a = DEC(t - 1), c = DEC(t);
            if (a == 1 && c == 0 && g0 < 0) g0 = t;
            else if (a == 0 && c == 1 && g0 >= 0) { if (t - g0 < q.merge) for (int u = g0; u < t; ++u) DEC(u) = 1; g0 = -1; }
        }
    }
    if (q.extend > 0) {
        int dist = q.extend + 1;
        for (int t = 0; t < n; ++t) { if (DEC(t)) dist = 0; else if (++dist <= q.extend) DEC(t) = 1; }
        dist = q.extend + 1;
        for (int t = n - 1; t >= 0; --t) { if (DEC(t)) dist = 0; else if (++dist <= q.extend) DEC(t) = 1; }
    }
    // split segments longer than max_speech at the lowest raw probability of [pos+max/2, pos+max)
    {
        const int half = q.max_sp >> 1;
        int t = 0;
        while (t < n) {
            if (!DEC(t)) { ++t; continue; }
            const int a = t;
            while (t < n && DEC(t)) ++t;
            if (t - a > q.max_sp) {
                int pos = a;
                const int e = t;
                while (pos + q.max_sp < e) {
                    const int lo = pos + half, hi = (pos + q.max_sp < e) ? pos + q.max_sp : e;
                    if (lo >= hi) break;
                    int arg = lo;
                    float best = pr[lo];
                    for (int u = lo + 1; u < hi; ++u) if (pr[u] < best) { best = pr[u]; arg = u; }
                    DEC(arg) = 0;
                    pos = arg + 1;
                }
            }
        }
    }
    // decisions out + (start_frame, end_frame) pairs (end = first frame after the run)
    int ns = 0, start = -1;
    for (int t = 0; t < n; ++t) {
        const int v = DEC(t);
        decisions[(size_t)b * stride + t] = (signed char)v;
        if (v && start < 0) start = t;
        if (!v && start >= 0) { if (ns < cap) { segs[((size_t)b * cap + ns) * 2] = start; segs[((size_t)b * cap + ns) * 2 + 1] = t; } ++ns; start = -1; }
    }
    if (start >= 0) { if (ns < cap) { segs[((size_t)b * cap + ns) * 2] = start; segs[((size_t)b * cap + ns) * 2 + 1] = n; } ++ns; }
    counts[b] = ns;
#undef SM
#undef DEC
}

}  // namespace firered
}  // namespace vadx

using namespace vadx::firered;

extern "C" size_t vadx_firered_packed_floats(const vadx_firered_cfg *cfg) {
    Dev d;
    if (!cfg || derive(cfg, &d)) return 0;
    return (size_t)d.total;
}

extern "C" int vadx_firered_pack_host(const vadx_firered_cfg *cfg, const vadx_firered_weights_host *w_in, float *p) {
    Dev d;
    VADX_REQUIRE(cfg && w_in && p, "vadx_firered_pack_host: NULL argument");
    VADX_REQUIRE(derive(cfg, &d) == 0, "vadx_firered_pack_host: unsupported config (idim 80, H<=256, P<=128, frames<=112, odim<=4)");
    memset(p, 0, sizeof(float) * d.total);
    auto mat = [&](int off, const float *src, int rows, int cols, int ld) {
        for (int r = 0; r < rows; ++r) memcpy(p + off + (size_t)r * ld, src + (size_t)r * cols, cols * sizeof(float));
    };
    VADX_REQUIRE(w_in->fc1_w && w_in->fc1_b && w_in->fc2_w && w_in->fc2_b && w_in->out_w && w_in->out_b, "vadx_firered_pack_host: NULL weight pointer");
    for (int r = 1; r < d.R; ++r)
        VADX_REQUIRE(w_in->blk_fc1_w[r] && w_in->blk_fc1_b[r] && w_in->blk_fc2_w[r], "vadx_firered_pack_host: NULL block %d weight", r);
    for (int m = 0; m < d.M; ++m) VADX_REQUIRE(w_in->dnn_w[m] && w_in->dnn_b[m], "vadx_firered_pack_host: NULL dnn %d weight", m);
    // Exact power-of-two rebalancing of the point-wise pairs (csrc/rebalance.h; ordinary checkpoints pass through untouched).  The P-wide trunk
    // carries the skip connections and the streaming caches (caches_in / caches_out of the reference's stream session), so it stays at its true
    // scale and ends every segment: [fc1, fc2], [blk_fc1_r, blk_fc2_r], [dnn_0 .. dnn_(M-1), out] (sigmoid after out).
    std::vector<float> r_fc1(w_in->fc1_w, w_in->fc1_w + (size_t)d.H * NMEL), r_fc1b(w_in->fc1_b, w_in->fc1_b + d.H);
    std::vector<float> r_fc2(w_in->fc2_w, w_in->fc2_w + (size_t)d.P * d.H), r_fc2b(w_in->fc2_b, w_in->fc2_b + d.P);
    std::vector<float> r_out(w_in->out_w, w_in->out_w + (size_t)d.odim * d.H), r_outb(w_in->out_b, w_in->out_b + d.odim);
    std::vector<float> r_b1[16], r_b1b[16], r_b2[16], r_dnn[4], r_dnnb[4];
    int reb_min = 1000;
    vadx::rebalance_chain({{&r_fc1, &r_fc1b}, {&r_fc2, &r_fc2b}}, &reb_min);
    vadx_firered_weights_host w_reb = *w_in;
    w_reb.fc1_w = r_fc1.data(); w_reb.fc1_b = r_fc1b.data(); w_reb.fc2_w = r_fc2.data(); w_reb.fc2_b = r_fc2b.data();
    for (int r = 1; r < d.R; ++r) {
        r_b1[r].assign(w_in->blk_fc1_w[r], w_in->blk_fc1_w[r] + (size_t)d.H * d.P);
        r_b1b[r].assign(w_in->blk_fc1_b[r], w_in->blk_fc1_b[r] + d.H);
        r_b2[r].assign(w_in->blk_fc2_w[r], w_in->blk_fc2_w[r] + (size_t)d.P * d.H);
        vadx::rebalance_chain({{&r_b1[r], &r_b1b[r]}, {&r_b2[r], nullptr}}, &reb_min);
        w_reb.blk_fc1_w[r] = r_b1[r].data(); w_reb.blk_fc1_b[r] = r_b1b[r].data(); w_reb.blk_fc2_w[r] = r_b2[r].data();
    }
    {
        std::vector<vadx::RebLayer> tail;
        for (int m = 0; m < d.M; ++m) {
            r_dnn[m].assign(w_in->dnn_w[m], w_in->dnn_w[m] + (size_t)d.H * (m == 0 ? d.P : d.H));
            r_dnnb[m].assign(w_in->dnn_b[m], w_in->dnn_b[m] + d.H);
            tail.push_back({&r_dnn[m], &r_dnnb[m]});
            w_reb.dnn_w[m] = r_dnn[m].data(); w_reb.dnn_b[m] = r_dnnb[m].data();
        }
        tail.push_back({&r_out, &r_outb});
        vadx::rebalance_chain(tail, &reb_min);
        w_reb.out_w = r_out.data(); w_reb.out_b = r_outb.data();
    }
    const vadx_firered_weights_host *w = &w_reb;
    mat(d.off_fc1, w->fc1_w, d.H, NMEL, NMEL); memcpy(p + d.off_fc1b, w->fc1_b, d.H * sizeof(float));
    mat(d.off_fc2, w->fc2_w, d.P, d.H, d.Hp); memcpy(p + d.off_fc2b, w->fc2_b, d.P * sizeof(float));
    for (int r = 0; r < d.R; ++r) {
        VADX_REQUIRE(w->fsmn_lb[r] && (d.N2 == 0 || w->fsmn_la[r]), "vadx_firered_pack_host: NULL FSMN filter %d", r);
        memcpy(p + d.off_lb[r], w->fsmn_lb[r], (size_t)d.P * d.N1 * sizeof(float));
        if (d.N2 > 0) memcpy(p + d.off_la[r], w->fsmn_la[r], (size_t)d.P * d.N2 * sizeof(float));
        if (d.N1 <= 20 && d.N2 <= 20)          // look-back taps end at j = 0 (slot 19), look-ahead taps start at j = 1 (slot 20)
            for (int ch = 0; ch < d.P; ++ch) {
                for (int k = 0; k < d.N1; ++k) p[d.off_win[r] + ch * 40 + 19 - (d.N1 - 1) + k] = w->fsmn_lb[r][(size_t)ch * d.N1 + k];
                for (int k = 0; k < d.N2; ++k) p[d.off_win[r] + ch * 40 + 20 + k] = w->fsmn_la[r][(size_t)ch * d.N2 + k];
            }
        if (r > 0) {
            VADX_REQUIRE(w->blk_fc1_w[r] && w->blk_fc1_b[r] && w->blk_fc2_w[r], "vadx_firered_pack_host: NULL block %d weight", r);
            mat(d.off_bfc1[r], w->blk_fc1_w[r], d.H, d.P, d.Pp); memcpy(p + d.off_bfc1b[r], w->blk_fc1_b[r], d.H * sizeof(float));
            mat(d.off_bfc2[r], w->blk_fc2_w[r], d.P, d.H, d.Hp);
        }
    }
    for (int m = 0; m < d.M; ++m) {
        VADX_REQUIRE(w->dnn_w[m] && w->dnn_b[m], "vadx_firered_pack_host: NULL dnn %d weight", m);
        mat(d.off_dnn[m], w->dnn_w[m], d.H, m == 0 ? d.P : d.H, m == 0 ? d.Pp : d.Hp);
        memcpy(p + d.off_dnnb[m], w->dnn_b[m], d.H * sizeof(float));
    }
    mat(d.off_out, w->out_w, d.odim, d.H, d.Hp); memcpy(p + d.off_outb, w->out_b, d.odim * sizeof(float));
    // GEMM operands go fragment-major (common.h); the FIR taps stay row-major (VALU)
    vadx::frag_major_inplace(p + d.off_out, 16, d.Hp);
    vadx::frag_major_inplace(p + d.off_fc1, d.Hp, NMEL);
    vadx::frag_major_inplace(p + d.off_fc2, d.Pp, d.Hp);
    for (int r = 1; r < d.R; ++r) {
        vadx::frag_major_inplace(p + d.off_bfc1[r], d.Hp, d.Pp);
        vadx::frag_major_inplace(p + d.off_bfc2[r], d.Pp, d.Hp);
    }
    for (int m = 0; m < d.M; ++m) vadx::frag_major_inplace(p + d.off_dnn[m], d.Hp, m == 0 ? d.Pp : d.Hp);
    if (d.np) {              // split A fragments [n-tile][chunk][np planes][QFRAG] from the ORIGINAL row-major weights, in cfg->arithmetic
        float wmax = 0.f;
        auto qmat = [&](int off, int rows_p, int nch, const float *W, int rows, int cols) {
            for (int nt = 0; nt < rows_p / 16; ++nt)
                for (int kc = 0; kc < nch; ++kc) {
                    float *f3 = p + off + (size_t)((nt * nch + kc) * d.np) * vadx::QFRAG;
                    for (int i = 0; i < 16; ++i)
                        for (int k = 0; k < 32; ++k) {
                            const int r = 16 * nt + i, c = 32 * kc + k;
                            const float v = (r < rows && c < cols) ? W[(size_t)r * cols + c] : 0.f;
                            if (d.np == 3) vadx::SchemeB3::put_host(f3, i, k, v, wmax);
                            else vadx::SchemeH2::put_host(f3, i, k, v, wmax);
                        }
                }
        };
        qmat(d.q_fc1, d.Hp, 3, w->fc1_w, d.H, NMEL);
        qmat(d.q_fc2, d.Pp, 8, w->fc2_w, d.P, d.H);
        for (int r = 1; r < d.R; ++r) {
            qmat(d.q_bfc1[r], d.Hp, 4, w->blk_fc1_w[r], d.H, d.P);
            qmat(d.q_bfc2[r], d.Pp, 8, w->blk_fc2_w[r], d.P, d.H);
        }
        if (d.M == 1) {
            qmat(d.q_dnn0, d.Hp, 4, w->dnn_w[0], d.H, d.P);
            qmat(d.q_head, d.Pp, 8, w->out_w, d.odim, d.H);
            memcpy(p + d.off_headb, w->out_b, d.odim * sizeof(float));
        }
        VADX_REQUIRE(d.arith != vadx::VADX_AR_H2 || wmax <= vadx::H_MAX,
                     "vadx_firered_pack_host: a weight (|w| up to %g) is outside the fp16 range: pack with cfg->arithmetic = VADX_ARITH_BF16X3", wmax);
        VADX_REQUIRE(d.arith != vadx::VADX_AR_H2 || reb_min >= vadx::REB_REFUSE,
                     "vadx_firered_pack_host: a weight tensor lies wholly below 2^%d (largest |w| < 2^%d after rebalancing), outside the fp16 range: pack "
                     "with cfg->arithmetic = VADX_ARITH_BF16X3", vadx::REB_REFUSE, reb_min + 1);
    }
    return VADX_OK;
}

// The fp16 x 2 kernel's sticky range flag (as vadx_silero_range_flag)
extern "C" int vadx_firered_range_flag(const vadx_firered_cfg *cfg, const float *packed, int reset, uint32_t *flag_host, float *amax_host, void *stream) {
    Dev d;
    VADX_REQUIRE(cfg && packed && flag_host, "vadx_firered_range_flag: NULL argument");
    VADX_REQUIRE(derive(cfg, &d) == 0, "vadx_firered_range_flag: unsupported config");
    uint32_t w[2] = {0, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    VADX_HIP_TRY(hipMemcpyAsync(w, packed + d.off_flag, sizeof(w), hipMemcpyDeviceToHost, st));
    VADX_HIP_TRY(hipStreamSynchronize(st));
    if (reset && (w[0] | w[1])) VADX_HIP_TRY(hipMemsetAsync(const_cast<float *>(packed) + d.off_flag, 0, sizeof(w), st));
    *flag_host = w[0];
    if (amax_host) memcpy(amax_host, &w[1], sizeof(float));
    return VADX_OK;
}

extern "C" int vadx_firered_run(const vadx_firered_cfg *cfg, const float *packed, const float *logmel, int windows,
                                float *probs, void *stream) {
    Dev d;
    VADX_REQUIRE(cfg && packed && logmel && probs, "vadx_firered_run: NULL argument");
    VADX_REQUIRE(derive(cfg, &d) == 0, "vadx_firered_run: unsupported config");
    VADX_REQUIRE(windows > 0, "vadx_firered_run: windows must be positive");
    VADX_DYN_LDS(firered_kernel<0>, LDS_FLOATS * sizeof(float));
    VADX_DYN_LDS(firered_kernel<1>, LDS_FLOATS * sizeof(float));
    VADX_DYN_LDS(firered_kernel<2>, LDS_FLOATS * sizeof(float));
    if (d.arith == vadx::VADX_AR_H2)
        hipLaunchKernelGGL(firered_kernel<2>, dim3(windows), dim3(THREADS), LDS_FLOATS * sizeof(float), static_cast<hipStream_t>(stream), d, packed, logmel, probs);
    else if (d.arith == vadx::VADX_AR_B3)
        hipLaunchKernelGGL(firered_kernel<1>, dim3(windows), dim3(THREADS), LDS_FLOATS * sizeof(float), static_cast<hipStream_t>(stream), d, packed, logmel, probs);
    else
        hipLaunchKernelGGL(firered_kernel<0>, dim3(windows), dim3(THREADS), LDS_FLOATS * sizeof(float),
                           static_cast<hipStream_t>(stream), d, packed, logmel, probs);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

extern "C" size_t vadx_vadpost_workspace_bytes(int batch, int stride) {
    if (batch <= 0 || stride <= 0) return 0;
    return (size_t)batch * stride * (sizeof(float) + 1) + 64;
}

extern "C" int vadx_vadpost(const vadx_vadpost_params *prm, const float *probs, int stride, const int32_t *n_frames,
                            int batch, int8_t *decisions, int32_t *segments, int32_t *counts, int cap,
                            void *workspace, size_t workspace_bytes, void *stream) {
    VADX_REQUIRE(prm && probs && n_frames && decisions && segments && counts && workspace, "vadx_vadpost: NULL argument");
    VADX_REQUIRE(batch > 0 && stride > 0 && cap > 0, "vadx_vadpost: batch/stride/cap must be positive");
    if (workspace_bytes < vadx_vadpost_workspace_bytes(batch, stride)) {
        vadx::set_error("vadx_vadpost: workspace %zu B < required %zu B", workspace_bytes, vadx_vadpost_workspace_bytes(batch, stride));
        return VADX_ENOSPACE;
    }
    PostDev q;
    q.ws = prm->smooth_window_size < 1 ? 1 : prm->smooth_window_size;
    q.thr = prm->prob_threshold; q.min_sp = prm->min_speech_frame; q.max_sp = prm->max_speech_frame;
    q.min_si = prm->min_silence_frame; q.merge = prm->merge_silence_frame; q.extend = prm->extend_speech_frame;
    q.inv_ws = (float)(1.0 / (double)q.ws);
    float *wsm = static_cast<float *>(workspace);
    signed char *wdec = reinterpret_cast<signed char *>(wsm + (size_t)batch * stride);
    hipLaunchKernelGGL(vadpost_kernel, dim3((batch + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), q, probs,
                       stride, n_frames, batch, wsm, wdec, reinterpret_cast<signed char *>(decisions), segments, counts, cap);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}



// ---- StreamVadPostprocessor on device (FireRedVAD/Export_FireRedVAD.py:1161-1339): one thread per stream, the state carried between
// chunks in a 32-word record per stream.  Same float32 operations in the same order as the reference's per-frame loop (ring-buffer moving
// average: sum += p - buf[k]; sum / count; compare with the float32 threshold), same four-state machine, forced split at max_speech.
namespace vadx {
namespace firered {
struct StreamPostDev { int ws; float thr; int pad_start, min_sp, max_sp, min_si; };
constexpr int SP_WORDS = 32, SP_MAXWS = 16;      // record: [0,16) window buffer, 16 sum, 17 pos, 18 count, 19 frame_cnt, 20 state, 21 speech_cnt,
                                                 // 22 silence_cnt, 23 hit_max, 24 last_start, 25 last_end
__global__ void stream_vadpost_kernel(StreamPostDev q, const float *__restrict__ probs, long long stride, int frames, int streams,
                                      float *__restrict__ state, int reset, int flush, int *__restrict__ segs, int *__restrict__ counts, int cap) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= streams) return;
    float *rec = state + (size_t)s * SP_WORDS;
    int *ri = reinterpret_cast<int *>(rec);
    float buf[SP_MAXWS];
    float sum = 0.f;
    int pos = 0, cnt = 0, fc = 0, st = 0, sp = 0, si = 0, hit = 0, last_start = -1, last_end = -1;
    if (!reset) {
#pragma unroll
        for (int k = 0; k < SP_MAXWS; ++k) buf[k] = rec[k];
        sum = rec[16]; pos = ri[17]; cnt = ri[18]; fc = ri[19]; st = ri[20]; sp = ri[21]; si = ri[22]; hit = ri[23]; last_start = ri[24]; last_end = ri[25];
    } else {
#pragma unroll
        for (int k = 0; k < SP_MAXWS; ++k) buf[k] = 0.f;
    }
    int *out = segs + (size_t)s * cap * 2, n = 0;
    auto emit = [&](int a, int b) { if (n < cap) { out[2 * n] = a; out[2 * n + 1] = b; } ++n; };
    const float *pr = probs + (size_t)s * stride;
    for (int t = 0; t < frames; ++t) {
        const float p = pr[t];
        ++fc;
        float sm = p;
        if (q.ws > 1) {
            float old = 0.f;
#pragma unroll
            for (int k = 0; k < SP_MAXWS; ++k) if (k == pos) { old = buf[k]; buf[k] = p; }      // (register array: no dynamic indexing)
            sum = __fadd_rn(sum, __fsub_rn(p, old));
            pos = pos + 1 == q.ws ? 0 : pos + 1;
            if (cnt < q.ws) ++cnt;
            sm = __fdiv_rn(sum, (float)cnt);
        }
        const bool speech = sm >= q.thr;
        int e0 = 0, e1 = 0;
        bool ended = false;
        if (hit) { last_start = fc; hit = 0; }           // a forced split re-opens a segment on the next frame
        auto close = [&]() { e0 = last_start; e1 = fc; ended = true; last_start = -1; last_end = fc; };
        if (st == 0) {
            if (speech) { st = 1; sp = 1; } else { ++si; sp = 0; }
        } else if (st == 1) {
            if (speech) {
                ++sp;
                if (sp >= q.min_sp) {
                    st = 2;
                    int a = fc - sp + 1 - q.pad_start;
                    a = a < 1 ? 1 : a;
                    last_start = a > last_end + 1 ? a : last_end + 1;
                    si = 0;
                }
            } else { st = 0; si = 1; sp = 0; }
        } else {
            ++sp;
            if (speech) {
                st = 2; si = 0;
                if (sp >= q.max_sp) { hit = 1; sp = 0; close(); }
            } else if (st == 2) { st = 3; si = 1; }
            else {
                ++si;
                if (si >= q.min_si) { st = 0; sp = 0; close(); }
            }
        }
        if (ended && e0 > 0) emit(e0 - 1 > 0 ? e0 - 1 : 0, e1 - 1 > 0 ? e1 - 1 : 0);
    }
    if (flush && last_start > 0) emit(last_start - 1 > 0 ? last_start - 1 : 0, fc - 1);      // unterminated segment at the end of the stream
    counts[s] = n;
#pragma unroll
    for (int k = 0; k < SP_MAXWS; ++k) rec[k] = buf[k];
    rec[16] = sum; ri[17] = pos; ri[18] = cnt; ri[19] = fc; ri[20] = st; ri[21] = sp; ri[22] = si; ri[23] = hit; ri[24] = last_start; ri[25] = last_end;
}
}  // namespace firered
}  // namespace vadx

extern "C" size_t vadx_stream_vadpost_state_bytes(int streams) { return streams > 0 ? (size_t)streams * vadx::firered::SP_WORDS * 4 : 0; }

extern "C" int vadx_stream_vadpost(const vadx_stream_vadpost_params *prm, const float *probs, int64_t probs_stride, int frames, int streams,
                                   void *state, int reset, int flush, int32_t *segments, int32_t *counts, int cap, void *stream) {
    VADX_REQUIRE(prm && probs && state && segments && counts, "vadx_stream_vadpost: NULL argument");
    VADX_REQUIRE(streams > 0 && frames >= 0 && cap > 0 && probs_stride >= frames, "vadx_stream_vadpost: streams / frames / cap / stride");
    vadx::firered::StreamPostDev q;
    q.ws = prm->smooth_window_size < 1 ? 1 : prm->smooth_window_size;
    VADX_REQUIRE(q.ws <= vadx::firered::SP_MAXWS, "vadx_stream_vadpost: smooth_window_size %d > %d", q.ws, vadx::firered::SP_MAXWS);
    q.thr = prm->speech_threshold;
    q.pad_start = prm->pad_start_frame > q.ws ? prm->pad_start_frame : q.ws;
    q.min_sp = prm->min_speech_frame; q.max_sp = prm->max_speech_frame; q.min_si = prm->min_silence_frame;
    hipLaunchKernelGGL(vadx::firered::stream_vadpost_kernel, dim3((streams + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), q, probs,
                       (long long)probs_stride, frames, streams, static_cast<float *>(state), reset, flush, segments, counts, cap);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

extern "C" int vadx_firered_stream_run(const vadx_firered_cfg *cfg, const float *packed, const float *logmel, int streams,
                                       const float *caches_in, float *caches_out, float *probs, void *stream) {
    Dev d;
    VADX_REQUIRE(cfg && packed && logmel && caches_in && caches_out && probs, "vadx_firered_stream_run: NULL argument");
    VADX_REQUIRE(derive(cfg, &d) == 0, "vadx_firered_stream_run: unsupported config");
    VADX_REQUIRE(cfg->N2 == 0, "vadx_firered_stream_run: the streaming model has no look-ahead filter (N2 must be 0)");
    VADX_REQUIRE(streams > 0 && caches_in != caches_out, "vadx_firered_stream_run: streams must be positive, caches must not alias");
    VADX_DYN_LDS(firered_stream_kernel, LDS_FLOATS * sizeof(float));
    hipLaunchKernelGGL(firered_stream_kernel, dim3(streams), dim3(THREADS), LDS_FLOATS * sizeof(float),
                       static_cast<hipStream_t>(stream), d, packed, logmel, caches_in, caches_out, streams, probs);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
