// silero.hip -- Silero-VAD v5 (16 kHz) for gfx950: encoder tile kernel + persistent LSTM kernel +
// device segmenter.  See DESIGN.md "Silero path" for the layout rationale.
//
// Work decomposition (B clips, T windows of 512 samples per clip):
//   silero_encode_kernel : one workgroup per (16 clips) x (1 window).  Everything that does NOT
//       depend on the recurrent state runs here as a chain of f32-MFMA GEMMs over LDS-resident
//       activations (k-major), weights streamed from L2 straight into MFMA operand registers:
//       STFT conv (258x256 basis, 4 frames) -> |.| -> conv1..conv4 (+ReLU) -> W_ih x + b.
//       Output gx[T][B/16][wave][gate][lane][4] in exactly the register order the LSTM kernel reads.
//   silero_lstm_kernel   : one persistent workgroup per 16 clips; W_hh (512x128 f32 = 256 KB) lives
//       in the VGPRs of its 8 waves for all T steps; h is exchanged through double-buffered LDS.
//   silero_segments_kernel: get_speech_timestamps' state machine, one clip per thread.
#include "silero_common.h"
#include "rebalance.h"
#include "split3.h"
#include "split2.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef VADX_SILERO_ENCODER_DEFAULT
// 0 = exact-f32 MFMA encoder, 1 = bf16 x 3 split-product encoder (silero_split.hip).  The split encoder is the default since
// tests/test_gpu_silero.py::test_split_products_are_as_exact_as_f32_products showed it CLOSER to the float64 evaluation of the same
// float32 network than the f32-MFMA encoder (max 9.3e-6 against 1.6e-5 on gx of scale 30, mean 2.1e-7 against 3.4e-7) and every
// parity test of that file passes on both.
#define VADX_SILERO_ENCODER_DEFAULT 2
#endif

// VADX_EXP: development-only what-if switches for tools/exp_encoder.py (results are wrong when set).
#ifndef VADX_EXP
#define VADX_EXP 0
#endif
#define ENC_SKIP(n) ((VADX_EXP >> (n)) & 1)          // VADX_EXP is a bit mask of what-if switches
// bit 14: per-phase cycle accounting of wave 0 of every workgroup (s_memtime deltas summed into enc_dbg[slot]; read with
// vadx_silero_debug_cycles, tools/exp_encoder.py) -- slot k = time from the previous mark to mark k
#if (VADX_EXP >> 14) & 1
__device__ unsigned long long enc_dbg[16];
#define ENC_T0() long long enc_t_ = __builtin_readcyclecounter()
#define ENC_MARK(slot) do { if (threadIdx.x == 0) { const long long n_ = __builtin_readcyclecounter(); atomicAdd(&enc_dbg[slot], (unsigned long long)(n_ - enc_t_)); enc_t_ = n_; } } while (0)
extern "C" int vadx_silero_debug_cycles(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(enc_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(enc_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define ENC_T0() do {} while (0)
#define ENC_MARK(slot) do {} while (0)
#endif
#if (VADX_EXP >> 2) & 1
#define ENC_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define ENC_SYNC() __syncthreads()
#endif

namespace vadx {
namespace silero {


// ---- encoder LDS map (floats): 52 624 B per workgroup => THREE workgroups per CU ------------------
// One region is reused by every phase; a phase whose output would overwrite its own input keeps the
// result in registers across a barrier before storing it (STFT -> V, conv1 -> A1):
//   X  [16 clips][642]  raw windows in their global layout (row stride 642 = 2 mod 32: with the
//       STFT's k permutation the half-wave (clip i, q) hits bank 2i+q -> conflict free)
//   -> V  [129 ch][6 planes x 16 (+4)]  |STFT| in the Winograd F(4,3) input domain (V_j = sum_f BT[j][f] |X_f|), k-major;
//         + 256 floats of scratch behind it (bin 64 partial sums [8 waves][re|im][16 clips] / Nyquist magnitudes of the four
//         frames, combined by the lanes that write that row of V)
//   -> A1 [128 ch][4 frames x 16 (+4)]  conv1 output, with A2 [64][2x16 (+4)] (conv2 output) right behind it
//   -> A3 [64][16 (+4)], A4 [128][16 (+4)]   (A1 is dead after conv2)
// Conv zero padding is never stored: the Winograd input transform is written for zero frames -1 and 4, and the taps of
// conv2..4 that would read padding are simply not issued.
constexpr int V_LD = 100, A1_LD = 68, A2_LD = 36, A3_LD = 20, A4_LD = 20;
constexpr int V_SCR = 129 * V_LD;                  // fold: [8 waves][2][16 clips]; dense: [4 frames][16 clips]
constexpr int R0_FLOATS = V_SCR + 256;             // 13156  (X: 16*642 = 10272; A1 + A2: 128*68 + 64*36 = 11008)
constexpr int A2_OFF = 128 * A1_LD;
constexpr int A3_OFF = 0, A4_OFF = 64 * A3_LD;
constexpr int ENC_LDS_FLOATS = R0_FLOATS;
static_assert(16 * X_LDM <= R0_FLOATS && A2_OFF + 64 * A2_LD <= R0_FLOATS && 3 * ENC_LDS_FLOATS * 4 <= 160 * 1024, "encoder LDS map");

// Winograd F(4,3) over the window's four STFT frames (frames -1 and 4 are the conv's zero padding): rows of B^T restricted
// to the four real frames, and A^T is applied to the six accumulators in the epilogue of phase 2.
__device__ constexpr float WINO_BT[6][4] = {{0.f, -5.f, 0.f, 1.f}, {-4.f, -4.f, 1.f, 1.f}, {4.f, -4.f, -1.f, 1.f},
                                            {-2.f, -1.f, 2.f, 1.f}, {2.f, -1.f, -2.f, 1.f}, {4.f, 0.f, -5.f, 0.f}};
constexpr int ENC_THREADS = 512;


// k-major pass that accumulates into acc[0][A0 .. A0+MT) of a wider accumulator array
template <int MT, int KB, int A0, int AN>
__device__ __forceinline__ void gemm_pass_sub(f32x4 (&acc)[1][AN], const float *act, int lda, const int (&moff)[MT],
                                              const float *wrow, int lane) {
    const int q = lane >> 4, i = lane & 15;
    const float *ap = act + (4 * q) * lda + i;
    f32x4 wcur = *reinterpret_cast<const f32x4 *>(wrow), wnxt;
#pragma unroll 1
    for (int S = 0; S < KB; ++S) {
        const int Sn = (S + 1 < KB) ? S + 1 : S;
        wnxt = *reinterpret_cast<const f32x4 *>(wrow + FRAG * Sn);
        const float *aps = ap + 16 * S * lda;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float wj = wcur[j];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[0][A0 + mt] = mfma16(aps[j * lda + moff[mt]], wj, acc[0][A0 + mt]);
        }
        wcur = wnxt;
    }
}

// Three Winograd planes side by side: acc[p] += V_p x U_p over KB blocks of 16 input channels.  Plane p's activations sit
// 16 columns further right in the same LDS rows, its weights KB fragments further in this lane's stream; the three
// accumulators alternate, so consecutive MFMAs are independent.

template <int KB>
__device__ __forceinline__ void gemm_planes3(f32x4 &a0, f32x4 &a1, f32x4 &a2, const float *act, int lda, const float *w, int lane) {
    // (An explicitly software-pipelined form of this loop -- weight fragments and LDS operands of block S + 1 in flight
    // under the MFMAs of block S, ping-pong registers, counted waits -- was built and measured: 7.5 - 8.0 ms against 7.3 ms
    // for this plain form.  Six waves per SIMD already interleave at MFMA-pair granularity; the extra live registers only
    // cost spills at the 80-VGPR budget.)  The loop is unrolled by four all the same: rolled, every block carried a dozen v_mov
    // (wc = wn) and 64-bit pointer updates -- 37 VALU instructions per 12 MFMAs -- and VALU time adds to f32-MFMA time here.
    const int q = lane >> 4, i = lane & 15;
    const float *ap = act + (4 * q) * lda + i;
    f32x4 wc[3], wn[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) wc[p] = *reinterpret_cast<const f32x4 *>(w + p * KB * FRAG);
#pragma unroll 4
    for (int S = 0; S < KB; ++S) {
        const int Sn = ENC_SKIP(13) ? 0 : ((S + 1 < KB) ? S + 1 : S);
#pragma unroll
        for (int p = 0; p < 3; ++p) wn[p] = *reinterpret_cast<const f32x4 *>(w + (p * KB + Sn) * FRAG);
        const float *aps = ap + 16 * S * lda;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0 = mfma16(aps[j * lda], wc[0][j], a0);
            a1 = mfma16(aps[j * lda + 16], wc[1][j], a1);
            a2 = mfma16(aps[j * lda + 32], wc[2][j], a2);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) wc[p] = wn[p];
    }
}

// Single-tile chain (one 16x16 output tile per wave, K = 16*NB): the weights of blocks B0..NB-1 sit contiguously in
// this lane's row; activations for block b come from column offset coff(b) and k-row 16*(b % KPB).  With one MFMA
// per ds_read there is nothing to hide an L2 round trip behind, so the weight stream runs DEPTH blocks ahead, and
// two accumulators alternate so consecutive MFMAs are independent (32- instead of 40-cycle cadence).
template <int B0, int NB, int KPB, int DEPTH, typename ColOff>
__device__ __forceinline__ f32x4 gemm_chain(const float *act, int lda, const float *wrow, int lane, ColOff coff) {
    const int q = lane >> 4, i = lane & 15;
    const float *ap = act + (4 * q) * lda + i;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 w[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (B0 + d < NB) w[d] = *reinterpret_cast<const f32x4 *>(wrow + FRAG * (B0 + d));
    float a[2][4];
    auto fetch_act = [&](int b, float (&dst)[4]) {
        const float *aps = ap + 16 * (b % KPB) * lda + coff(b);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = aps[j * lda];
    };
    fetch_act(B0, a[0]);
#pragma unroll
    for (int b = B0; b < NB; ++b) {
        const f32x4 wc = w[(b - B0) % DEPTH];
        if (b + DEPTH < NB) w[(b - B0) % DEPTH] = *reinterpret_cast<const f32x4 *>(wrow + FRAG * (b + DEPTH));
        if (b + 1 < NB) fetch_act(b + 1, a[(b - B0 + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);          // requests above stay ahead of this block's MFMAs; bounded registers
        const float (&ac)[4] = a[(b - B0) & 1];
        acc0 = mfma16(ac[0], wc[0], acc0);
        acc1 = mfma16(ac[1], wc[1], acc1);
        acc0 = mfma16(ac[2], wc[2], acc0);
        acc1 = mfma16(ac[3], wc[3], acc1);
    }
    return acc0 + acc1;
}


template <typename SampleT>
__global__ __launch_bounds__(ENC_THREADS, 6) void silero_encode_kernel(
    const float *__restrict__ P, const SampleT *__restrict__ audio, float in_scale, long long n_samples,
    long long row_stride, long long origin, int B, int G, int T, int Gws, int g0, float *__restrict__ gx) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *X = lds, *V = lds, *A1 = lds;           // one region, one tenant per phase
    float *A3 = lds + A3_OFF, *A4 = lds + A4_OFF;
    float *A2 = lds + A2_OFF, *scr = lds + V_SCR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, i = lane & 15;
    const int grp = blockIdx.x % G, t = blockIdx.x / G;

    const bool fold = P[OFF_FOLD] != 0.f;      // uniform: the basis has the DFT symmetries -> folded STFT pass
    ENC_T0();
    // ---------------- phase 0: copy the 16 windows (576 samples each) + right reflect pad of 64
    // (folded pass: even / odd samples go to separate planes of the clip row)
    auto xslot = [fold](int pp) { return fold ? (pp & 1) * X_ODD + (pp >> 1) : pp; };
    {
        const long long base = (long long)t * 512 + origin;
        const bool vec_ok = ((row_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(audio) & (SampleIO<SampleT>::VEC_ALIGN - 1)) == 0) && n_samples >= 4 && !ENC_SKIP(1);
        // Work item e = (clip c, float4 p/4) of the 16 x 576-sample tile.  In the normal case every lane loads its
        // float4 UNCONDITIONALLY from a clamped address, all five loads back to back (a load under a condition -- even a
        // uniform one -- compiles to a branch plus a full wait: five serialised HBM round trips per tile); the rare edge
        // lanes (first / last windows, clips past B) patch their values afterwards in a branch that is normally skipped.
        // FAST PATH (every window except a clip's first and last few, and groups that run past the batch): wave w stages clips
        // 2w and 2w+1, lane l the float4s l, l+64, l+128 (< 144) of each -- the address is a wave-uniform row base plus
        // 16 * lane bytes (no divisions, no 64-bit per-lane arithmetic, no per-element conditions), six loads back to back,
        // and a float4 of samples (p .. p+3) leaves as two 8-byte LDS stores: (p, p+2) into the even plane, (p+1, p+3) into
        // the odd plane.  Only the 17 float4s that hold samples 511..574 also write the mirrored reflect-pad slots.
        // (Cycle accounting of the general path below: 8 k cycles to issue the five loads, 11 k for the LDS scatter -- a
        // sixth of the tile's time on index arithmetic -- against 1.8 k waiting for HBM.)
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const bool fast = fold && vec_ok && base >= 0 && base + 576 <= n_samples && (long long)grp * 16 + 16 <= B;
        if (fast) {
            f32x4 xv[2][3];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const SampleT *src = audio + ((long long)grp * 16 + 2 * wv + k2) * row_stride + base;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int f = j < 2 ? lane + 64 * j : min(lane + 128, 143);
                    xv[k2][j] = SampleIO<SampleT>::load4(src + 4 * f, in_scale);
                }
            }
#if (VADX_EXP >> 14) & 1
            ENC_MARK(11);
            if (xv[1][2][3] == 123.456f) X[0] = 0.f;
            ENC_MARK(12);
#endif
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                float *row = X + (2 * wv + k2) * X_LDM;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int f = lane + 64 * j;
                    if (j < 2 || lane < 16) {
                        const f32x4 v = xv[k2][j];
                        *reinterpret_cast<float2 *>(row + 2 * f) = float2{v[0], v[2]};
                        *reinterpret_cast<float2 *>(row + X_ODD + 2 * f) = float2{v[1], v[3]};
                        if (f >= 127) {                                               // samples 508..575: reflect pad (0, 64)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                const int pp = 4 * f + jj;
                                if (pp >= 511 && pp <= 574) row[xslot(1150 - pp)] = v[jj];
                            }
                        }
                    }
                }
            }
        } else {
        f32x4 x4[5];
        if (vec_ok) {
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int e = min(tid + ENC_THREADS * it, 16 * 144 - 1), c = e / 144, p = 4 * (e - c * 144);
                const long long b = (long long)grp * 16 + c, idx = base + p;
                const SampleT *src = audio + (b < B ? b : 0) * row_stride;
                const long long idc = idx < 0 ? 0 : (idx + 3 < n_samples ? idx : ((n_samples - 4) & ~3LL));
                x4[it] = SampleIO<SampleT>::load4(src + idc, in_scale);
            }
        }
#if (VADX_EXP >> 14) & 1
        ENC_MARK(11);                                   // loads issued
        if (x4[4][3] == 123.456f) X[0] = 0.f;           // (forces the wait here)
        ENC_MARK(12);                                   // loads landed
#endif
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int e = tid + ENC_THREADS * it;            // 16 clips x 144 float4
            if (e < 16 * 144) {
                const int c = e / 144, p = 4 * (e - c * 144);
                const long long b = (long long)grp * 16 + c;
                const bool bvalid = b < B;
                const SampleT *src = audio + (bvalid ? b : 0) * row_stride;
                const long long idx = base + p;
                float v[4];
                if (vec_ok && bvalid && idx >= 0 && idx + 3 < n_samples) {
                    v[0] = x4[it][0]; v[1] = x4[it][1]; v[2] = x4[it][2]; v[3] = x4[it][3];
                } else if (ENC_SKIP(1)) {
                    v[0] = v[1] = v[2] = v[3] = 1e-3f * (float)(p & 63);
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        v[jj] = (bvalid && idx + jj >= 0 && idx + jj < n_samples) ? SampleIO<SampleT>::load1(src + idx + jj, in_scale) : 0.f;
                }
                float *row = X + c * X_LDM;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int pp = p + jj;
                    row[xslot(pp)] = v[jj];
                    if (pp >= 511 && pp <= 574) row[xslot(1150 - pp)] = v[jj];       // reflect pad (0,64)
                }
            }
        }
        }   // general path
    }
#if (VADX_EXP >> 14) & 1
    ENC_MARK(13);                                       // LDS writes issued
#endif
    ENC_SYNC();
    ENC_MARK(0);

    // ---------------- phase 1: STFT conv -> magnitude (results stay in registers until every wave is done with X)
    if (fold) {
        // wave = (bin tile tl: bins 16 tl .. 16 tl + 15, frame pair fp); it produces bins k and 128 - k
        const int tl = wave & 3, fp = wave >> 2;
        // bin 64 (its own mirror) on the VALU, spread over all eight waves: wave = (frame f = wave & 3, half h of n = 1..128),
        // lane = (clip i, quarter q of the half): 16 taps each, summed over q by shuffles and over h through the scratch
        float b64re = 0.f, b64im = 0.f;
        if (!ENC_SKIP(7)) {
            const int f = wave & 3, h = wave >> 2, n0 = h * 64 + q * 16;
            // the lane's 16 + 16 coefficients as eight 16-byte loads, one group of four taps ahead of its use (they were 32 scalar
            // loads: 32 VMEM instructions per wave and tile).  This block runs BEFORE the folded MFMA pass: beside that pass's 32
            // accumulator registers the eight fragments spilled at the 80-VGPR budget
            static_assert(OFF_B64 % 4 == 0, "bin-64 table must be 16-byte aligned in the packed blob");
            const float *xr = X + i * X_LDM + 64 * f;
            f32x4 cre = *reinterpret_cast<const f32x4 *>(P + OFF_B64 + n0), cim = *reinterpret_cast<const f32x4 *>(P + OFF_B64 + 128 + n0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int un = u + 1 < 4 ? u + 1 : u;
                const f32x4 nre = *reinterpret_cast<const f32x4 *>(P + OFF_B64 + n0 + 4 * un);
                const f32x4 nim = *reinterpret_cast<const f32x4 *>(P + OFF_B64 + 128 + n0 + 4 * un);
#pragma unroll
                for (int k3 = 0; k3 < 4; ++k3) {
                    const int n = n0 + 4 * u + k3 + 1;                                // 1..128, mirror 256 - n
                    const float a = xr[(n & 1) * X_ODD + (n >> 1)], b = xr[(n & 1) * X_ODD + ((256 - n) >> 1)];
                    b64re = fmaf(a + b, cre[k3], b64re);
                    b64im = fmaf(a - b, cim[k3], b64im);
                }
                cre = nre;
                cim = nim;
            }
            b64re += __shfl_xor(b64re, 16); b64re += __shfl_xor(b64re, 32);
            b64im += __shfl_xor(b64im, 16); b64im += __shfl_xor(b64im, 32);
            if (h == 0) {                                                             // the n = 0 tap
                const float x0 = xr[0];
                b64re = fmaf(x0, P[OFF_B64 + 256], b64re);
                b64im = fmaf(x0, P[OFF_B64 + 257], b64im);
            }
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the bin-64 registers from overlapping the fold's accumulators
        f32x4 ere[2], eim[2], ore[2], oim[2];
        const float c0 = P[OFF_S0 + tl * 16 + i], s0 = P[OFF_S0 + 64 + tl * 16 + i];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x0 = X[(4 * q + r) * X_LDM + 64 * (2 * fp + f)];          // n = 0 belongs to the even class
                ere[f][r] = c0 * x0;
                eim[f][r] = s0 * x0;
                ore[f][r] = 0.f;
                oim[f][r] = 0.f;
            }
        if (!ENC_SKIP(4)) {
            const float *row = X + i * X_LDM + 128 * fp;                              // frame 2 fp starts 64 fp... x2 planes
            const float *wt = P + OFF_SF + tl * 16 * FRAG + lane * 4;                 // [E|O][re|im][4 blocks]
            stft_fold_class(ere, eim, row + 1 + q, row + 127 - q, wt, wt + 4 * FRAG);
            stft_fold_class(ore, oim, row + X_ODD + q, row + X_ODD + 127 - q, wt + 8 * FRAG, wt + 12 * FRAG);
        }
        ENC_SYNC();          // every wave is done reading X: V may now overwrite it
        ENC_MARK(1);
        // Winograd input transform across the two waves that hold a bin's four frames: the fp = 0 wave stores its share
        // c[j][0] |X_0| + c[j][1] |X_1| of every plane, the fp = 1 wave adds c[j][2] |X_2| + c[j][3] |X_3| behind a barrier
        // (one store, one add: the sum does not depend on timing).
        const int k = tl * 16 + i;
        f32x4 mk[2], mn[2];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = ere[f][r] + ore[f][r], pi = eim[f][r] + oim[f][r];
                const float nr = ere[f][r] - ore[f][r], ni = eim[f][r] - oim[f][r];
                mk[f][r] = mag_sqrt(pr * pr + pi * pi);
                mn[f][r] = mag_sqrt(nr * nr + ni * ni);
            }
        float *vk = V + k * V_LD + 4 * q, *vn = V + (128 - k) * V_LD + 4 * q;
        if (fp == 0) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                *reinterpret_cast<f32x4 *>(vk + j * 16) = WINO_BT[j][0] * mk[0] + WINO_BT[j][1] * mk[1];
                *reinterpret_cast<f32x4 *>(vn + j * 16) = WINO_BT[j][0] * mn[0] + WINO_BT[j][1] * mn[1];
            }
        }
        if (q == 0) { scr[wave * 32 + i] = b64re; scr[wave * 32 + 16 + i] = b64im; }      // partial (re, im) of bin 64, frame wave & 3
        ENC_SYNC();
        ENC_MARK(2);
        if (fp == 1) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x4 *pk = reinterpret_cast<f32x4 *>(vk + j * 16), *pn = reinterpret_cast<f32x4 *>(vn + j * 16);
                *pk = *pk + (WINO_BT[j][2] * mk[0] + WINO_BT[j][3] * mk[1]);
                *pn = *pn + (WINO_BT[j][2] * mn[0] + WINO_BT[j][3] * mn[1]);
            }
        } else if (tid < 96) {                                // bin 64 row of V: plane j = tid / 16, clip = tid % 16
            const int j = tid >> 4, c = tid & 15;
            float m[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const float re = scr[f * 32 + c] + scr[(f + 4) * 32 + c], im = scr[f * 32 + 16 + c] + scr[(f + 4) * 32 + 16 + c];
                m[f] = mag_sqrt(re * re + im * im);
            }
            V[64 * V_LD + j * 16 + c] = WINO_BT[j][0] * m[0] + WINO_BT[j][1] * m[1] + WINO_BT[j][2] * m[2] + WINO_BT[j][3] * m[3];
        }
    } else {
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *const wrow[2] = {P + OFF_STFT + (wave * 32 + i) * 256,
                                      P + OFF_STFT + (wave * 32 + 16 + i) * 256};
        const int koff[4] = {0, 128, 256, 384};
        gemm_pass_mmajor<2, 4, 16>(acc, X, X_LDM, koff, wrow, lane);
        float nyq = 0.f;
        if (wave < 4) {   // Nyquist bin: frame f = wave, lane = (clip i, k-quarter q)
            const int f = wave;
            const float *nre = P + OFF_NYQ + q * 64, *nim = P + OFF_NYQ + 256 + q * 64;
            const float *xp = X + i * X_LDM + 128 * f + q * 64;
            float sre = 0.f, sim = 0.f;
#pragma unroll 8
            for (int k = 0; k < 64; ++k) {
                const float x = xp[k];
                sre = fmaf(x, nre[k], sre);
                sim = fmaf(x, nim[k], sim);
            }
            sre += __shfl_xor(sre, 16); sre += __shfl_xor(sre, 32);
            sim += __shfl_xor(sim, 16); sim += __shfl_xor(sim, 32);
            nyq = mag_sqrt(sre * sre + sim * sim);
        }
        ENC_SYNC();          // every wave is done reading X: V may now overwrite it
        ENC_MARK(3);
        {   // this wave holds all four frames of its 16 bins: Winograd input transform in registers
            f32x4 m[4];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) m[f][r] = mag_sqrt(acc[0][f][r] * acc[0][f][r] + acc[1][f][r] * acc[1][f][r]);
            float *vr = V + (wave * 16 + i) * V_LD + 4 * q;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                *reinterpret_cast<f32x4 *>(vr + j * 16) = WINO_BT[j][0] * m[0] + WINO_BT[j][1] * m[1] + WINO_BT[j][2] * m[2] + WINO_BT[j][3] * m[3];
        }
        if (wave < 4 && q == 0) scr[wave * 16 + i] = nyq;
        ENC_SYNC();
        ENC_MARK(4);
        if (tid < 96) {                                       // Nyquist bin: plane j = tid / 16, clip = tid % 16
            const int j = tid >> 4, c = tid & 15;
            V[128 * V_LD + j * 16 + c] = WINO_BT[j][0] * scr[c] + WINO_BT[j][1] * scr[16 + c] + WINO_BT[j][2] * scr[32 + c] + WINO_BT[j][3] * scr[48 + c];
        }
    }
    ENC_SYNC();
    ENC_MARK(5);

    // ---------------- phase 2: conv1 129->128, k3 s1 p1, ReLU as Winograd F(4,3) over the window's four frames:
    // six plane GEMMs M_j = U_j V_j (U_j = G g packed on the host in float64) instead of the ten tap GEMMs a direct
    // conv issues for 4 output frames with zero frames either side (1536 instead of 2560 MFMAs per tile), then
    // Y = A^T M on the accumulators.  Rounding error stays in the float32 class (measured 3x a direct conv's on |STFT|
    // data, three orders below the 1e-4 score tolerance).
    {
        f32x4 acc[6];
        {   // input channel 128 (the Nyquist bin) on VALU
            const float *un = P + OFF_C1N + (wave * 16 + i) * 8;
            const f32x4 ua = *reinterpret_cast<const f32x4 *>(un), ub = *reinterpret_cast<const f32x4 *>(un + 4);
            const float *vn = V + 128 * V_LD + 4 * q;
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] = *reinterpret_cast<const f32x4 *>(vn + j * 16) * (j < 4 ? ua[j] : ub[j - 4]);
        }
        if (!ENC_SKIP(5)) {
            const float *w0 = P + OFF_C1 + wave * 6 * 8 * FRAG + lane * 4;      // [oc tile][plane][8 blocks]
            gemm_planes3<8>(acc[0], acc[1], acc[2], V, V_LD, w0, lane);
            gemm_planes3<8>(acc[3], acc[4], acc[5], V + 48, V_LD, w0 + 3 * 8 * FRAG, lane);
        }
        const float bias = P[OFF_B1 + wave * 16 + i];
        ENC_SYNC();          // every wave is done reading V: A1 may now overwrite it
        ENC_MARK(6);
        const f32x4 s12 = acc[1] + acc[2], d12 = acc[1] - acc[2], s34 = acc[3] + acc[4], d34 = acc[3] - acc[4];
        f32x4 y[4];
        y[0] = acc[0] + s12 + s34;
        y[1] = d12 + 2.f * d34;
        y[2] = s12 + 4.f * s34;
        y[3] = d12 + 8.f * d34 + acc[5];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(y[f][r] + bias, 0.f);
            *reinterpret_cast<f32x4 *>(&A1[(wave * 16 + i) * A1_LD + f * 16 + 4 * q]) = v;
        }
    }
    ENC_SYNC();
    ENC_MARK(7);

    // ---------------- phase 3: conv2 128->64, k3 s2 p1, ReLU (out frame fp reads in frames 2fp-1..2fp+1)
    {
        const int nt = wave & 3, fp = wave >> 2;
        const float *w0 = P + OFF_C2 + nt * 24 * FRAG + lane * 4;      // [oc tile][tap*8 + S]: 24 blocks of 16 k
        f32x4 a2 = {0.f, 0.f, 0.f, 0.f};
        if (ENC_SKIP(3)) {} else
        if (fp == 0) a2 = gemm_chain<8, 24, 8, 4>(A1, A1_LD, w0, lane, [](int b) { return (b / 8 - 1) * 16; });
        else a2 = gemm_chain<0, 24, 8, 4>(A1, A1_LD, w0, lane, [](int b) { return (b / 8 + 1) * 16; });
        const float bias = P[OFF_B2 + nt * 16 + i];
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(a2[r] + bias, 0.f);
        *reinterpret_cast<f32x4 *>(&A2[(nt * 16 + i) * A2_LD + fp * 16 + 4 * q]) = v;
    }
    ENC_SYNC();
    ENC_MARK(8);

    // ---------------- phase 4: conv3 64->64, k3 s2 p1, ReLU (1 out frame; tap 0 reads padding)
    if (wave < 4) {
        // [2 taps][64] contiguous: 8 blocks; tap ps reads A2 frame ps
        const f32x4 a3 = ENC_SKIP(3) ? f32x4{0.f, 0.f, 0.f, 0.f} : gemm_chain<0, 8, 4, 4>(A2, A2_LD, P + OFF_C3 + wave * 8 * FRAG + lane * 4, lane,
                                                [](int b) { return (b / 4) * 16; });
        const float bias = P[OFF_B3 + wave * 16 + i];
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(a3[r] + bias, 0.f);
        *reinterpret_cast<f32x4 *>(&A3[(wave * 16 + i) * A3_LD + 4 * q]) = v;
    }
    ENC_SYNC();
    ENC_MARK(9);

    // ---------------- phase 5: conv4 64->128, k3 s1 p1, ReLU (1 frame in/out; centre tap only)
    {
        const f32x4 a4 = ENC_SKIP(3) ? f32x4{0.f, 0.f, 0.f, 0.f} : gemm_chain<0, 4, 4, 4>(A3, A3_LD, P + OFF_C4 + wave * 4 * FRAG + lane * 4, lane, [](int) { return 0; });
        const float bias = P[OFF_B4 + wave * 16 + i];
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(a4[r] + bias, 0.f);
        *reinterpret_cast<f32x4 *>(&A4[(wave * 16 + i) * A4_LD + 4 * q]) = v;
    }
    ENC_SYNC();
    ENC_MARK(10);

    // ---------------- phase 6: LSTM input projection, gate-major (D rows = hidden units)
    {
        f32x4 acc[4][1];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            acc[g][0] = *reinterpret_cast<const f32x4 *>(P + OFF_BG + g * 128 + wave * 16 + 4 * q);
        if (!ENC_SKIP(6)) {        // W_ih tiles [gate][wave][8 blocks]; D rows = hidden units (gate-major for the LSTM kernel)
            const float *wl = P + OFF_IH + wave * 8 * FRAG + lane * 4;
            const float *ap = A4 + (4 * q) * A4_LD + i;
            f32x4 wcur[4], wnxt[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wcur[g] = *reinterpret_cast<const f32x4 *>(wl + g * 64 * FRAG);
#pragma unroll
            for (int S = 0; S < 8; ++S) {
                const int Sn = ENC_SKIP(13) ? 0 : ((S + 1 < 8) ? S + 1 : S);
#pragma unroll
                for (int g = 0; g < 4; ++g) wnxt[g] = *reinterpret_cast<const f32x4 *>(wl + (ENC_SKIP(13) ? 0 : g * 64 * FRAG) + Sn * FRAG);
                __builtin_amdgcn_sched_barrier(0);
                const float *aps = ap + 16 * S * A4_LD;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float av = aps[j * A4_LD];
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g][0] = mfma16(wcur[g][j], av, acc[g][0]);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) wcur[g] = wnxt[g];
            }
        }
        float *dst = gx + ((size_t)t * Gws + g0 + grp) * GX_TILE_FLOATS + (size_t)wave * 4 * 256 + lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (!ENC_SKIP(8) || acc[g][0][0] == 12345.f) *reinterpret_cast<f32x4 *>(dst + g * 256) = acc[g][0];
    }
    ENC_MARK(15);
}

// ---- persistent LSTM --------------------------------------------------------------------------
constexpr int HS_LD = 136;                       // h [16 clips][128] (+8 pad: conflict-free b128)
constexpr int LSTM_LDS_FLOATS = 2 * 16 * HS_LD + 2 * 8 * 16;
constexpr int LSTM_THREADS = 512;

__global__ __launch_bounds__(LSTM_THREADS, 2) void silero_lstm_kernel(
    const float *__restrict__ P, const float *__restrict__ gx, const float *__restrict__ state0,
    int B, int G, int T, float *__restrict__ probs, long long probs_stride,
    float *__restrict__ state_n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Hs = lds;                              // [2][16][HS_LD]
    float *part = lds + 2 * 16 * HS_LD;           // [2][8][16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int grp = blockIdx.x;
    const long long b = (long long)grp * 16 + n;
    const bool bvalid = b < B;
    const int u0 = wave * 16 + 4 * q;             // this lane's 4 hidden units

    // W_hh rows of this wave's 4 gate tiles, resident for the whole clip: 128 VGPRs
    float a[4][32];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float *row = P + OFF_HH + (g * 128 + wave * 16 + n) * 128 + 4 * q;
#pragma unroll
        for (int S = 0; S < 8; ++S) {
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(row + 16 * S);
            a[g][4 * S + 0] = w4[0]; a[g][4 * S + 1] = w4[1]; a[g][4 * S + 2] = w4[2]; a[g][4 * S + 3] = w4[3];
        }
    }
    f32x4 dw = *reinterpret_cast<const f32x4 *>(P + OFF_DW + u0);
    const float db = P[OFF_DB];

    f32x4 c = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
    if (state0 != nullptr && bvalid) {
        h = *reinterpret_cast<const f32x4 *>(state0 + b * 128 + u0);
        c = *reinterpret_cast<const f32x4 *>(state0 + ((long long)B + b) * 128 + u0);
    }
    *reinterpret_cast<f32x4 *>(&Hs[n * HS_LD + u0]) = h;
    __syncthreads();

    const float *gsrc = gx + (size_t)grp * GX_TILE_FLOATS + (size_t)wave * 4 * 256 + lane * 4;
    const size_t gstep = (size_t)G * GX_TILE_FLOATS;
    f32x4 gcur[4], gnxt[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) gcur[g] = *reinterpret_cast<const f32x4 *>(gsrc + g * 256);

    int cur = 0;
    for (int t = 0; t < T; ++t) {
        const int tn = (t + 1 < T) ? t + 1 : t;
#pragma unroll
        for (int g = 0; g < 4; ++g) gnxt[g] = ENC_SKIP(11) ? gcur[g] : *reinterpret_cast<const f32x4 *>(gsrc + tn * gstep + g * 256);

        // Gate-outer order: a gate's 32 MFMAs finish before the next gate's start, so its non-linearity (VALU +
        // transcendentals) runs in the shadow of the following gate's MFMAs instead of after all 128 of them; only
        // sigmoid(o) and the h update stay exposed.  (Within a gate the MFMAs are a dependent chain at a 40-cycle
        // cadence; the SIMD's other wave fills the gaps.)
        const float *hb = Hs + cur * 16 * HS_LD + n * HS_LD + 4 * q;
        f32x4 hv[8];
#pragma unroll
        for (int S = 0; S < 8; ++S) hv[S] = *reinterpret_cast<const f32x4 *>(hb + 16 * S);
        auto gate = [&](int g) {
            f32x4 acc = gcur[g];
#pragma unroll
            for (int S = 0; S < 8; ++S)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = mfma16(a[g][4 * S + j], hv[S][j], acc);
            return acc;
        };
        const f32x4 ai = gate(0), af = gate(1);
        f32x4 ig, fg, gg;
#pragma unroll
        for (int r = 0; r < 4; ++r) ig[r] = ENC_SKIP(9) ? ai[r] * 1e-3f : gate_sigmoid(ai[r]);
        const f32x4 ag = gate(2);
#pragma unroll
        for (int r = 0; r < 4; ++r) fg[r] = ENC_SKIP(9) ? af[r] * 1e-3f : gate_sigmoid(af[r]);
        const f32x4 ao = gate(3);
        float dpart = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gg[r] = ENC_SKIP(9) ? ag[r] * 1e-3f : gate_tanh(ag[r]);
            c[r] = fg[r] * c[r] + ig[r] * gg[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h[r] = ENC_SKIP(9) ? ao[r] * 1e-3f + c[r] * 1e-3f : gate_sigmoid(ao[r]) * gate_tanh(c[r]);
            dpart = fmaf(dw[r], fmaxf(h[r], 0.f), dpart);
        }
        const int nxt = cur ^ 1;
        *reinterpret_cast<f32x4 *>(&Hs[nxt * 16 * HS_LD + n * HS_LD + u0]) = h;
        dpart += __shfl_xor(dpart, 16);
        dpart += __shfl_xor(dpart, 32);
        if (q == 0) part[(nxt * 8 + wave) * 16 + n] = dpart;
        if (!ENC_SKIP(10)) __syncthreads();
        if (wave == 0 && lane < 16 && bvalid && !ENC_SKIP(12)) {
            float s = db;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += part[(nxt * 8 + w) * 16 + lane];
            probs[b * probs_stride + t] = sigmoidf_(s);
        }
        cur = nxt;
#pragma unroll
        for (int g = 0; g < 4; ++g) gcur[g] = gnxt[g];
    }
    if (state_n != nullptr && bvalid) {
        *reinterpret_cast<f32x4 *>(state_n + b * 128 + u0) = h;
        *reinterpret_cast<f32x4 *>(state_n + ((long long)B + b) * 128 + u0) = c;
    }
}

// ---- segmenter: get_speech_timestamps' state machine (utils_vad.py:374-476), one clip/thread ----
__global__ void silero_segments_kernel(const float *__restrict__ probs, int B, int T,
                                       const long long *__restrict__ n_samples,
                                       vadx_silero_seg_params prm, long long *__restrict__ segs,
                                       int *__restrict__ counts, int cap) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double sr = (double)prm.sampling_rate;
    const long long W = (prm.sampling_rate == 16000) ? 512 : 256;
    const double thr = prm.threshold;
    const double neg = (prm.neg_threshold < 0.0) ? fmax(thr - 0.15, 0.01) : prm.neg_threshold;
    const double min_speech = sr * prm.min_speech_duration_ms / 1000.0;
    const double pad = sr * prm.speech_pad_ms / 1000.0;
    const double max_speech = sr * prm.max_speech_duration_s - (double)W - 2.0 * pad;
    const double min_sil = sr * prm.min_silence_duration_ms / 1000.0;
    const double min_sil_at_max = sr * prm.min_silence_at_max_speech / 1000.0;
    const long long L = n_samples[b];
    const int nwin = (int)((L + W - 1) / W) < T ? (int)((L + W - 1) / W) : T;

    long long *out = segs + (size_t)b * cap * 2;
    int ns = 0;
    auto push = [&](long long s, long long e) {
        if (ns < cap) { out[2 * ns] = s; out[2 * ns + 1] = e; }
        ++ns;
    };

    // The state machine runs in WINDOW units on 32-bit integers: every position it handles is a multiple of W, and the
    // reference's float comparisons of such positions against second-derived thresholds are equivalent to integer
    // comparisons against the thresholds' floor / ceiling in windows (d*W > X <=> d > floor(X/W); d*W < X <=> d < ceil(X/W);
    // W is a power of two, so X/W is exact); a float32 probability compares against a double threshold like against the
    // smallest float32 not below it.  (The 64-bit / double version spent ~1200 cycles per step on conversions.)
    auto win_floor = [&](double x) { const double q = floor(x / (double)W); return q >= 2147483647.0 ? 2147483647 : (q <= -2147483648.0 ? (int)-2147483647 - 1 : (int)q); };
    auto win_ceil = [&](double x) { const double q = ceil(x / (double)W); return q >= 2147483647.0 ? 2147483647 : (q <= -2147483648.0 ? (int)-2147483647 - 1 : (int)q); };
    auto f32_not_below = [](double x) { float f = (float)x; if ((double)f < x) f = nextafterf(f, INFINITY); return f; };
    const float thr_f = f32_not_below(thr), neg_f = f32_not_below(neg);
    const int d_sil_at_max = win_floor(min_sil_at_max), d_max_speech = win_floor(max_speech), d_min_sil = win_ceil(min_sil),
              d_min_speech = win_floor(min_speech);
    auto pushw = [&](int s_, int e_) { push((long long)s_ * W, (long long)e_ * W); };

    bool triggered = false, have_cur = false, have_possible = false;
    int cur_start = 0, temp_end = 0, prev_end = 0, next_start = 0;      // window indices (0 doubles as "unset", like the reference)
    int best_end = 0, best_dur = 0;
    auto step = [&](int pos, float p) {          // one probability through the state machine (`continue` -> return)
        if (p >= thr_f && temp_end) {
            const int gap = pos - temp_end;
            if (gap > d_sil_at_max) {
                if (!have_possible || gap > best_dur) { best_end = temp_end; best_dur = gap; }
                have_possible = true;
            }
            temp_end = 0;
            if (next_start < prev_end) next_start = pos;
        }
        if (p >= thr_f && !triggered) {
            triggered = true; cur_start = pos; have_cur = true;
            return;
        }
        if (triggered && (pos - cur_start) > d_max_speech) {
            if (prm.use_max_poss_sil_at_max_speech && have_possible) {
                prev_end = best_end;
                pushw(cur_start, prev_end);
                have_cur = false;
                next_start = prev_end + best_dur;
                if (next_start < prev_end + pos) { cur_start = next_start; have_cur = true; }
                else triggered = false;
                prev_end = next_start = temp_end = 0;
                have_possible = false;
            } else if (prev_end) {
                pushw(cur_start, prev_end);
                have_cur = false;
                if (next_start < prev_end) triggered = false;
                else { cur_start = next_start; have_cur = true; }
                prev_end = next_start = temp_end = 0;
                have_possible = false;
            } else {
                pushw(cur_start, pos);
                have_cur = false;
                prev_end = next_start = temp_end = 0;
                triggered = false;
                have_possible = false;
                return;
            }
        }
        if (p < neg_f && triggered) {
            if (!temp_end) temp_end = pos;
            const int sil_now = pos - temp_end;
            if (!prm.use_max_poss_sil_at_max_speech && sil_now > d_sil_at_max) prev_end = temp_end;
            if (sil_now < d_min_sil) return;
            if ((temp_end - cur_start) > d_min_speech) pushw(cur_start, temp_end);
            have_cur = false;
            prev_end = next_start = temp_end = 0;
            triggered = false;
            have_possible = false;
            return;
        }
        };
    // probabilities are fetched eight at a time, unconditionally (clamped index): a per-step load sat on the serial path
    // of all T steps (0.23 ms per batch for what is a few microseconds of arithmetic).  (Staging them through LDS with
    // coalesced row reads was tried: no further gain -- what remains is the divergent state machine itself, ~1000
    // cycles per step for one wave per 64 clips.)
    const float *pr = probs + (size_t)b * T;
    for (int k0 = 0; k0 < nwin; k0 += 8) {
        float buf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) buf[j] = pr[k0 + j < T ? k0 + j : T - 1];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < nwin) step(k0 + j, buf[j]);
    }
    if (have_cur && (double)(L - (long long)cur_start * W) > min_speech) push((long long)cur_start * W, L);

    // +-speech_pad with midpoint split of short gaps (utils_vad.py:464-476)
    const int m = ns < cap ? ns : cap;
    for (int s = 0; s < m; ++s) {
        if (s == 0) out[0] = (long long)fmax(0.0, (double)out[0] - pad);
        if (s != m - 1) {
            const long long gap = out[2 * (s + 1)] - out[2 * s + 1];
            if ((double)gap < 2.0 * pad) {
                const long long half = gap >= 0 ? gap / 2 : -((-gap + 1) / 2);   // Python floor //
                out[2 * s + 1] += half;
                const long long ns0 = out[2 * (s + 1)] - half;
                out[2 * (s + 1)] = ns0 > 0 ? ns0 : 0;
            } else {
                out[2 * s + 1] = (long long)fmin((double)L, (double)out[2 * s + 1] + pad);
                out[2 * (s + 1)] = (long long)fmax(0.0, (double)out[2 * (s + 1)] - pad);
            }
        } else {
            out[2 * s + 1] = (long long)fmin((double)L, (double)out[2 * s + 1] + pad);
        }
    }
    counts[b] = ns;
}


}  // namespace silero
}  // namespace vadx

// =================================================================================================
// C ABI
// =================================================================================================
using namespace vadx::silero;
using vadx::FRAG;

extern "C" size_t vadx_silero_packed_floats(void) { return (size_t)PACKED_FLOATS; }

extern "C" int vadx_silero_pack_host(const vadx_silero_weights_host *w_in, float *p) {
    VADX_REQUIRE(w_in && p, "vadx_silero_pack_host: NULL argument");
    VADX_REQUIRE(w_in->stft_basis && w_in->lstm_w_ih && w_in->lstm_w_hh && w_in->lstm_b_ih && w_in->lstm_b_hh && w_in->dec_w && w_in->dec_b,
                 "vadx_silero_pack_host: NULL weight pointer");
    for (int k = 0; k < 4; ++k) VADX_REQUIRE(w_in->enc_w[k] && w_in->enc_b[k], "vadx_silero_pack_host: NULL encoder weight %d", k);
    memset(p, 0, sizeof(float) * PACKED_FLOATS);
    // conv1 -> ReLU -> conv2 -> ReLU -> conv3 -> ReLU -> conv4 -> ReLU -> W_ih is one chain of affine layers with only ReLU between them: exact
    // power-of-two rebalancing (csrc/rebalance.h) when a layer's weights sit outside [2^-10, 2^7); ordinary checkpoints pass through untouched.
    // gx (b_ih + b_hh, then the LSTM's non-linearities) stays at its true scale: W_ih is the segment's last layer.
    static const size_t enc_nw[4] = {128 * 129 * 3, 64 * 128 * 3, 64 * 64 * 3, 128 * 64 * 3}, enc_nb[4] = {128, 64, 64, 128};
    std::vector<float> rw[5], rb[4];
    for (int k = 0; k < 4; ++k) { rw[k].assign(w_in->enc_w[k], w_in->enc_w[k] + enc_nw[k]); rb[k].assign(w_in->enc_b[k], w_in->enc_b[k] + enc_nb[k]); }
    rw[4].assign(w_in->lstm_w_ih, w_in->lstm_w_ih + 512 * 128);
    int reb_min = 1000;
    vadx::rebalance_chain({{&rw[0], &rb[0]}, {&rw[1], &rb[1]}, {&rw[2], &rb[2]}, {&rw[3], &rb[3]}, {&rw[4], nullptr}}, &reb_min);
    vadx_silero_weights_host w_reb = *w_in;
    for (int k = 0; k < 4; ++k) { w_reb.enc_w[k] = rw[k].data(); w_reb.enc_b[k] = rb[k].data(); }
    w_reb.lstm_w_ih = rw[4].data();
    const vadx_silero_weights_host *w = &w_reb;
    // STFT basis rows regrouped per wave: [wave][re 16 bins | im 16 bins][256]
    for (int wv = 0; wv < 8; ++wv)
        for (int part = 0; part < 2; ++part)
            for (int i = 0; i < 16; ++i)
                for (int S = 0; S < 16; ++S)          // k permutation of the m-major STFT pass: slot 4q+j <- k = q+4j
                    for (int qq = 0; qq < 4; ++qq)
                        for (int j = 0; j < 4; ++j)
                            p[OFF_STFT + (size_t)(wv * 32 + part * 16 + i) * 256 + 16 * S + 4 * qq + j] =
                                w->stft_basis[(size_t)(part * 129 + wv * 16 + i) * 256 + 16 * S + qq + 4 * j];
    memcpy(p + OFF_NYQ, w->stft_basis + (size_t)128 * 256, 256 * sizeof(float));
    memcpy(p + OFF_NYQ + 256, w->stft_basis + (size_t)257 * 256, 256 * sizeof(float));
    // fragment-major slot of element (row i of its tile, contraction index k) inside a tile that starts at `base`
    auto frag = [](size_t base, int i, int k) { return base + ((size_t)(k / 16) * 64 + ((k % 16) / 4) * 16 + i) * 4 + (k % 4); };
    float hmax = 0.f;       // largest |weight| handed to the fp16 x 2 fragments (must stay inside the fp16 range)
    {   // folded basis: valid when the table has the time symmetry c[k][256-n] == c[k][n], s[k][256-n] == -s[k][n]
        // (n = 1..127; s[k][128] == 0) AND the frequency symmetry c[128-k][n] == (-1)^n c[k][n],
        // s[128-k][n] == -(-1)^n s[k][n], both up to f32 rounding of the table (1e-6 of the largest entry) -- which
        // every windowed real-DFT basis satisfies.  Otherwise the kernel takes the dense pass.
        const float *re = w->stft_basis, *im = w->stft_basis + (size_t)129 * 256;
        float amax = 0.f, dev = 0.f;
        for (size_t e = 0; e < (size_t)258 * 256; ++e) amax = fmaxf(amax, fabsf(w->stft_basis[e]));
        for (int k = 0; k <= 128; ++k) {
            for (int n = 1; n < 128; ++n) {
                dev = fmaxf(dev, fabsf(re[k * 256 + n] - re[k * 256 + 256 - n]));
                dev = fmaxf(dev, fabsf(im[k * 256 + n] + im[k * 256 + 256 - n]));
            }
            dev = fmaxf(dev, fabsf(im[k * 256 + 128]));
        }
        for (int k = 0; k < 64; ++k)
            for (int n = 0; n < 256; ++n) {
                const float sg = (n & 1) ? -1.f : 1.f;
                dev = fmaxf(dev, fabsf(re[(128 - k) * 256 + n] - sg * re[k * 256 + n]));
                dev = fmaxf(dev, fabsf(im[(128 - k) * 256 + n] + sg * im[k * 256 + n]));
            }
        const bool fold = dev <= 1e-6f * amax;
        p[OFF_FOLD] = fold ? 1.f : 0.f;
        if (fold) {
            // symmetrised coefficient of bin k (<= 64) at sample n: average of the four table entries that must agree
            auto C = [&](int k, int n) {
                const float sg = (n & 1) ? -1.f : 1.f;
                const int nm = (256 - n) & 255;
                return 0.25f * (re[k * 256 + n] + re[k * 256 + nm] + sg * (re[(128 - k) * 256 + n] + re[(128 - k) * 256 + nm]));
            };
            auto S = [&](int k, int n) {
                const float sg = (n & 1) ? -1.f : 1.f;
                const int nm = (256 - n) & 255;
                return 0.25f * (im[k * 256 + n] - im[k * 256 + nm] - sg * (im[(128 - k) * 256 + n] - im[(128 - k) * 256 + nm]));
            };
            for (int k = 0; k < 64; ++k) {
                const int tl = k / 16, i = k % 16;
                for (int cls = 0; cls < 2; ++cls)
                    for (int m = 0; m < 64; ++m) {
                        const int n = cls ? 2 * m + 1 : 2 * m + 2;
                        // n = 128 is its own mirror: x[128] gets added to itself, so its coefficient is halved
                        const float cv = (n == 128) ? 0.5f * C(k, 128) : C(k, n), sv = (n == 128) ? 0.f : S(k, n);
                        const int slot = 16 * (m / 16) + 4 * (m % 4) + (m % 16) / 4;     // m = 16S + q + 4j -> slot 16S + 4q + j
                        p[frag(OFF_SF + (size_t)((tl * 2 + cls) * 2 + 0) * 4 * FRAG, i, slot)] = cv;
                        p[frag(OFF_SF + (size_t)((tl * 2 + cls) * 2 + 1) * 4 * FRAG, i, slot)] = sv;
                        // the same coefficients as split fragments (silero_split.hip's STFT), pairs in natural order
                        vadx::qfrag_put(p + OFF_QSF + (size_t)((((tl * 2 + cls) * 2 + 0) * 2 + m / 32) * 3) * QF, i, m % 32, cv);
                        vadx::qfrag_put(p + OFF_QSF + (size_t)((((tl * 2 + cls) * 2 + 1) * 2 + m / 32) * 3) * QF, i, m % 32, sv);
                        // ... and as fp16 x 2 fragment pairs (silero_h2.hip)
                        hmax = fmaxf(hmax, vadx::hfrag_put(p + OFF_HSF + (size_t)((((tl * 2 + cls) * 2 + 0) * 2 + m / 32) * 2) * HF, i, m % 32, cv));
                        hmax = fmaxf(hmax, vadx::hfrag_put(p + OFF_HSF + (size_t)((((tl * 2 + cls) * 2 + 1) * 2 + m / 32) * 2) * HF, i, m % 32, sv));
                    }
                p[OFF_S0 + k] = 0.5f * (re[k * 256] + re[(128 - k) * 256]);
                p[OFF_S0 + 64 + k] = 0.5f * (im[k * 256] - im[(128 - k) * 256]);
            }
            for (int n = 1; n <= 128; ++n) {          // bin 64, time-folded only
                const float h = (n == 128) ? 0.5f : 1.f;
                p[OFF_B64 + n - 1] = h * 0.5f * (re[64 * 256 + n] + re[64 * 256 + ((256 - n) & 255)]);
                p[OFF_B64 + 128 + n - 1] = (n == 128) ? 0.f : 0.5f * (im[64 * 256 + n] - im[64 * 256 + 256 - n]);
            }
            p[OFF_B64 + 256] = re[64 * 256];
            p[OFF_B64 + 257] = im[64 * 256];
            // bin 64 as row 0 of a fifth bin tile of the fp16 x 2 STFT (its coefficients are OFF_B64's, per class)
            for (int cls = 0; cls < 2; ++cls)
                for (int m = 0; m < 64; ++m) {
                    const int n = cls ? 2 * m + 1 : 2 * m + 2;
                    hmax = fmaxf(hmax, vadx::hfrag_put(p + OFF_HSF + (size_t)((((4 * 2 + cls) * 2 + 0) * 2 + m / 32) * 2) * HF, 0, m % 32, p[OFF_B64 + n - 1]));
                    hmax = fmaxf(hmax, vadx::hfrag_put(p + OFF_HSF + (size_t)((((4 * 2 + cls) * 2 + 1) * 2 + m / 32) * 2) * HF, 0, m % 32, p[OFF_B64 + 128 + n - 1]));
                }
        }
    }
    {   // conv1 in the Winograd F(4,3) domain: U_j[co][ci] = sum_t G[j][t] g[co][ci][t], evaluated in float64
        static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                       {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
        for (int co = 0; co < 128; ++co)
            for (int j = 0; j < 6; ++j)
                for (int ci = 0; ci < 129; ++ci) {
                    const float *g = w->enc_w[0] + ((size_t)co * 129 + ci) * 3;
                    const float u = (float)(G[j][0] * (double)g[0] + G[j][1] * (double)g[1] + G[j][2] * (double)g[2]);
                    if (ci < 128) p[frag(OFF_C1 + (size_t)((co / 16) * 6 + j) * 8 * FRAG, co % 16, ci)] = u;
                    else p[OFF_C1N + co * 8 + j] = u;
                }
    }
    memcpy(p + OFF_B1, w->enc_b[0], 128 * sizeof(float));
    for (int co = 0; co < 64; ++co)
        for (int kk = 0; kk < 3; ++kk)
            for (int ci = 0; ci < 128; ++ci)
                p[frag(OFF_C2 + (size_t)(co / 16) * 24 * FRAG, co % 16, kk * 128 + ci)] = w->enc_w[1][((size_t)co * 128 + ci) * 3 + kk];
    memcpy(p + OFF_B2, w->enc_b[1], 64 * sizeof(float));
    for (int co = 0; co < 64; ++co)
        for (int ps = 0; ps < 2; ++ps)
            for (int ci = 0; ci < 64; ++ci)
                p[frag(OFF_C3 + (size_t)(co / 16) * 8 * FRAG, co % 16, ps * 64 + ci)] = w->enc_w[2][((size_t)co * 64 + ci) * 3 + (ps + 1)];
    memcpy(p + OFF_B3, w->enc_b[2], 64 * sizeof(float));
    for (int co = 0; co < 128; ++co)
        for (int ci = 0; ci < 64; ++ci)
            p[frag(OFF_C4 + (size_t)(co / 16) * 4 * FRAG, co % 16, ci)] = w->enc_w[3][((size_t)co * 64 + ci) * 3 + 1];
    memcpy(p + OFF_B4, w->enc_b[3], 128 * sizeof(float));
    for (int r = 0; r < 512; ++r)          // row r = gate*128 + unit; tile = gate*8 + unit/16
        for (int k = 0; k < 128; ++k) p[frag(OFF_IH + (size_t)(r / 16) * 8 * FRAG, r % 16, k)] = w->lstm_w_ih[(size_t)r * 128 + k];
    for (int r = 0; r < 512; ++r) p[OFF_BG + r] = w->lstm_b_ih[r] + w->lstm_b_hh[r];
    memcpy(p + OFF_HH, w->lstm_w_hh, 512 * 128 * sizeof(float));
    memcpy(p + OFF_DW, w->dec_w, 128 * sizeof(float));
    p[OFF_DB] = w->dec_b[0];
    // ---- the same conv / W_ih weights as bf16 x 3 fragments for the split-product encoder (silero_split.hip, split3.h)
    for (int rt = 0; rt < 8; ++rt)
        for (int kc = 0; kc < 4; ++kc)
            for (int tap = 0; tap < 3; ++tap) {
                float *f3 = p + OFF_Q1 + (size_t)(((rt * 4 + kc) * 3 + tap) * 3) * QF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) {
                        const int slot = 32 * kc + k, bin = slot <= 64 ? slot : 192 - slot;        // the order the STFT pass leaves the bins in
                        vadx::qfrag_put(f3, i, k, w->enc_w[0][((size_t)(16 * rt + i) * 129 + bin) * 3 + tap]);
                    }
            }
    for (int co = 0; co < 128; ++co)
        for (int tap = 0; tap < 3; ++tap) p[OFF_Q1N + co * 4 + tap] = w->enc_w[0][((size_t)co * 129 + 128) * 3 + tap];
    for (int rt = 0; rt < 4; ++rt)
        for (int kc = 0; kc < 4; ++kc)
            for (int tap = 0; tap < 3; ++tap) {
                float *f3 = p + OFF_Q2 + (size_t)(((rt * 4 + kc) * 3 + tap) * 3) * QF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) vadx::qfrag_put(f3, i, k, w->enc_w[1][((size_t)(16 * rt + i) * 128 + 32 * kc + k) * 3 + tap]);
            }
    for (int rt = 0; rt < 4; ++rt)
        for (int th = 0; th < 2; ++th)
            for (int kc = 0; kc < 2; ++kc) {
                float *f3 = p + OFF_Q3 + (size_t)(((rt * 2 + th) * 2 + kc) * 3) * QF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) vadx::qfrag_put(f3, i, k, w->enc_w[2][((size_t)(16 * rt + i) * 64 + 32 * kc + k) * 3 + th + 1]);
            }
    for (int rt = 0; rt < 8; ++rt)
        for (int kc = 0; kc < 2; ++kc) {
            float *f3 = p + OFF_Q4 + (size_t)((rt * 2 + kc) * 3) * QF;
            for (int i = 0; i < 16; ++i)
                for (int k = 0; k < 32; ++k) vadx::qfrag_put(f3, i, k, w->enc_w[3][((size_t)(16 * rt + i) * 64 + 32 * kc + k) * 3 + 1]);
        }
    for (int wv = 0; wv < 8; ++wv)
        for (int kc = 0; kc < 4; ++kc)
            for (int g = 0; g < 4; ++g) {
                float *f3 = p + OFF_QIH + (size_t)(((wv * 4 + kc) * 4 + g) * 3) * QF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) vadx::qfrag_put(f3, i, k, w->lstm_w_ih[(size_t)(g * 128 + wv * 16 + i) * 128 + 32 * kc + k]);
            }
    for (int wv = 0; wv < 8; ++wv)
        for (int g = 0; g < 4; ++g)
            for (int kc = 0; kc < 4; ++kc) {
                float *f3 = p + OFF_QHH + (size_t)(((wv * 4 + g) * 4 + kc) * 3) * QF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) vadx::qfrag_put(f3, i, k, w->lstm_w_hh[(size_t)(g * 128 + wv * 16 + i) * 128 + 32 * kc + k]);
            }
    // ---- ... and as fp16 x 2 fragment pairs for silero_h2.hip (split2.h), same orders
    for (int rt = 0; rt < 8; ++rt)
        for (int kc = 0; kc < 4; ++kc)
            for (int tap = 0; tap < 3; ++tap) {
                float *f2 = p + OFF_H1 + (size_t)(((rt * 4 + kc) * 3 + tap) * 2) * HF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) {
                        const int slot = 32 * kc + k, bin = slot <= 64 ? slot : 192 - slot;
                        hmax = fmaxf(hmax, vadx::hfrag_put(f2, i, k, w->enc_w[0][((size_t)(16 * rt + i) * 129 + bin) * 3 + tap]));
                    }
            }
    for (int rt = 0; rt < 4; ++rt)
        for (int kc = 0; kc < 4; ++kc)
            for (int tap = 0; tap < 3; ++tap) {
                float *f2 = p + OFF_H2 + (size_t)(((rt * 4 + kc) * 3 + tap) * 2) * HF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) hmax = fmaxf(hmax, vadx::hfrag_put(f2, i, k, w->enc_w[1][((size_t)(16 * rt + i) * 128 + 32 * kc + k) * 3 + tap]));
            }
    for (int rt = 0; rt < 4; ++rt)
        for (int th = 0; th < 2; ++th)
            for (int kc = 0; kc < 2; ++kc) {
                float *f2 = p + OFF_H3 + (size_t)(((rt * 2 + th) * 2 + kc) * 2) * HF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) hmax = fmaxf(hmax, vadx::hfrag_put(f2, i, k, w->enc_w[2][((size_t)(16 * rt + i) * 64 + 32 * kc + k) * 3 + th + 1]));
            }
    for (int rt = 0; rt < 8; ++rt)
        for (int kc = 0; kc < 2; ++kc) {
            float *f2 = p + OFF_H4 + (size_t)((rt * 2 + kc) * 2) * HF;
            for (int i = 0; i < 16; ++i)
                for (int k = 0; k < 32; ++k) hmax = fmaxf(hmax, vadx::hfrag_put(f2, i, k, w->enc_w[3][((size_t)(16 * rt + i) * 64 + 32 * kc + k) * 3 + 1]));
        }
    for (int wv = 0; wv < 8; ++wv)
        for (int kc = 0; kc < 4; ++kc)
            for (int g = 0; g < 4; ++g) {
                float *f2 = p + OFF_HIH + (size_t)(((wv * 4 + kc) * 4 + g) * 2) * HF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) hmax = fmaxf(hmax, vadx::hfrag_put(f2, i, k, w->lstm_w_ih[(size_t)(g * 128 + wv * 16 + i) * 128 + 32 * kc + k]));
            }
    for (int wv = 0; wv < 8; ++wv)
        for (int g = 0; g < 4; ++g)
            for (int kc = 0; kc < 4; ++kc) {
                float *f2 = p + OFF_HHH + (size_t)(((wv * 4 + g) * 4 + kc) * 2) * HF;
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 32; ++k) hmax = fmaxf(hmax, vadx::hfrag_put(f2, i, k, w->lstm_w_hh[(size_t)(g * 128 + wv * 16 + i) * 128 + 32 * kc + k]));
            }
    // the fp16 x 2 kernels need the folded STFT pass and every weight inside the fp16 range (NaN fails the comparison too)
    // ... and no weight tensor wholly below the smallest normal fp16 once the chain is rebalanced (csrc/rebalance.h)
    {
        std::vector<float> whh(w->lstm_w_hh, w->lstm_w_hh + 512 * 128);
        const int e = vadx::reb_exponent(whh);
        if (e > -100000 && e < reb_min) reb_min = e;
    }
    p[OFF_HFLAG] = (p[OFF_FOLD] != 0.f && hmax <= vadx::H_MAX && reb_min >= vadx::REB_REFUSE) ? 1.f : 0.f;
    return VADX_OK;
}

// Which kernel set a launch uses comes with the call (include/vadx.h: vadx_silero_cfg.arithmetic); NULL / VADX_ARITH_AUTO = the
// library's default.  Internal numbering: 0 = exact-f32 MFMAs, 1 = bf16 x 3 (silero_split.hip), 2 = fp16 x 2 (silero_h2.hip).
static int arith_of(const vadx_silero_cfg *cfg) {
    const int a = cfg ? cfg->arithmetic : VADX_ARITH_AUTO;
    switch (a) {
        case VADX_ARITH_AUTO: return VADX_SILERO_ENCODER_DEFAULT;
        case VADX_ARITH_F32: return 0;
        case VADX_ARITH_BF16X3: return 1;
        case VADX_ARITH_F16X2: return 2;
        default: return -1;
    }
}
#define VADX_SILERO_ARITH(cfg, who)                                                                               \
    const int arith = arith_of(cfg);                                                                              \
    VADX_REQUIRE(arith >= 0, who ": cfg->arithmetic=%d is not one of VADX_ARITH_*", (cfg) ? (cfg)->arithmetic : 0)

// The fp16 x 2 kernels' sticky range flag (silero_common.h: OFF_HFLAG): copies the two words [flag, bits of the largest |activation|]
// to the host (synchronises `stream`) and, with reset != 0, clears them on the device.
extern "C" int vadx_silero_range_flag(const float *packed, int reset, uint32_t *flag_host, float *amax_host, void *stream) {
    VADX_REQUIRE(packed && flag_host, "vadx_silero_range_flag: NULL argument");
    uint32_t w[2] = {0, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    VADX_HIP_TRY(hipMemcpyAsync(w, packed + OFF_HFLAG + 1, sizeof(w), hipMemcpyDeviceToHost, st));
    VADX_HIP_TRY(hipStreamSynchronize(st));
    if (reset && (w[0] | w[1]))
        VADX_HIP_TRY(hipMemsetAsync(const_cast<float *>(packed) + OFF_HFLAG + 1, 0, sizeof(w), st));
    *flag_host = w[0];
    if (amax_host) memcpy(amax_host, &w[1], sizeof(float));
    return VADX_OK;
}

extern "C" size_t vadx_silero_workspace_bytes(int batch, int steps) {
    if (batch <= 0 || steps <= 0) return 0;
    const size_t G = ((size_t)batch + 15) / 16;
    return G * (size_t)steps * GX_TILE_FLOATS * sizeof(float);
}

template <typename S>
static int silero_encode_launch(const float *packed, const S *src, float in_scale, long long n_valid, long long row_stride,
                                long long origin, int batch, int steps, void *ws, size_t ws_bytes, void *stream,
                                const vadx_silero_cfg *cfg, int first_group = 0, int total_batch = 0) {
    VADX_SILERO_ARITH(cfg, "silero");
    VADX_REQUIRE(packed && src && ws, "silero: NULL pointer argument");
    VADX_REQUIRE(batch > 0 && steps > 0, "silero: batch=%d steps=%d must be positive", batch, steps);
    VADX_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
                 "silero: packed weights and workspace must be 16-byte aligned");
    if (total_batch <= 0) total_batch = batch;          // the workspace is laid out for total_batch clips; this launch fills
    const int Gws = (total_batch + 15) / 16;            // groups [first_group, first_group + G) of it
    const int G = (batch + 15) / 16;
    VADX_REQUIRE(first_group >= 0 && first_group + G <= Gws, "silero: clip groups [%d, %d) outside the workspace's %d", first_group,
                 first_group + G, Gws);
    if (ws_bytes < vadx_silero_workspace_bytes(total_batch, steps)) {
        vadx::set_error("silero: workspace %zu B < required %zu B", ws_bytes, vadx_silero_workspace_bytes(total_batch, steps));
        return VADX_ENOSPACE;
    }
    const long long nblk = (long long)G * steps;
    VADX_REQUIRE(nblk < (1LL << 31), "silero: too many tiles (%lld)", nblk);
    if (arith == 2)
        return silero_encode_h2_launch<S>(packed, src, in_scale, n_valid, row_stride, origin, batch, G, steps, Gws, first_group,
                                          static_cast<float *>(ws), stream);
    if (arith == 1)
        return silero_encode_split_launch<S>(packed, src, in_scale, n_valid, row_stride, origin, batch, G, steps, Gws, first_group,
                                             static_cast<float *>(ws), stream);
    VADX_DYN_LDS(silero_encode_kernel<S>, ENC_LDS_FLOATS * sizeof(float));
    hipLaunchKernelGGL(silero_encode_kernel<S>, dim3((unsigned)nblk), dim3(ENC_THREADS), ENC_LDS_FLOATS * sizeof(float),
                       static_cast<hipStream_t>(stream), packed, src, in_scale, n_valid, row_stride, origin, batch, G, steps,
                       Gws, first_group, static_cast<float *>(ws));
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

static int silero_recur_launch(const float *packed, const void *ws, size_t ws_bytes, int batch, int steps,
                               const float *state0, float *probs, long long probs_stride, float *state_n, void *stream,
                               const vadx_silero_cfg *cfg) {
    VADX_SILERO_ARITH(cfg, "silero");
    VADX_REQUIRE(packed && ws && probs, "silero: NULL pointer argument");
    VADX_REQUIRE(batch > 0 && steps > 0, "silero: batch=%d steps=%d must be positive", batch, steps);
    if (ws_bytes < vadx_silero_workspace_bytes(batch, steps)) {
        vadx::set_error("silero: workspace %zu B < required %zu B", ws_bytes, vadx_silero_workspace_bytes(batch, steps));
        return VADX_ENOSPACE;
    }
    const int G = (batch + 15) / 16;
    if (arith == 2)
        return silero_lstm_h2_launch(packed, static_cast<const float *>(ws), state0, batch, G, steps, probs, probs_stride, state_n, stream);
    if (arith == 1)
        return silero_lstm_split_launch(packed, static_cast<const float *>(ws), state0, batch, G, steps, probs, probs_stride, state_n, stream);
    hipLaunchKernelGGL(silero_lstm_kernel, dim3(G), dim3(LSTM_THREADS), LSTM_LDS_FLOATS * sizeof(float),
                       static_cast<hipStream_t>(stream), packed, static_cast<const float *>(ws), state0, batch, G, steps,
                       probs, probs_stride, state_n);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

static int silero_run(const float *packed, const float *src, long long n_valid, long long row_stride,
                      long long origin, int batch, int steps, const float *state0, float *probs,
                      long long probs_stride, float *state_n, void *ws, size_t ws_bytes, void *stream, const vadx_silero_cfg *cfg) {
    int rc = silero_encode_launch(packed, src, 1.0f, n_valid, row_stride, origin, batch, steps, ws, ws_bytes, stream, cfg);
    if (rc != VADX_OK) return rc;
    return silero_recur_launch(packed, ws, ws_bytes, batch, steps, state0, probs, probs_stride, state_n, stream, cfg);
}

extern "C" int vadx_silero_encode(const float *packed, const float *audio, int batch, int64_t n_samples,
                                  int64_t row_stride, void *workspace, size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg) {
    VADX_REQUIRE(n_samples > 0 && row_stride >= n_samples, "vadx_silero_encode: n_samples=%lld row_stride=%lld",
                 (long long)n_samples, (long long)row_stride);
    const long long steps = (n_samples + 511) / 512;
    VADX_REQUIRE(steps < (1LL << 30), "vadx_silero_encode: clip too long");
    return silero_encode_launch(packed, audio, 1.0f, n_samples, row_stride, -64, batch, (int)steps, workspace, workspace_bytes, stream, cfg);
}

extern "C" int vadx_silero_encode_pcm16(const float *packed, const int16_t *audio, float scale, int batch, int64_t n_samples,
                                        int64_t row_stride, void *workspace, size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg) {
    VADX_REQUIRE(n_samples > 0 && row_stride >= n_samples, "vadx_silero_encode_pcm16: n_samples=%lld row_stride=%lld",
                 (long long)n_samples, (long long)row_stride);
    const long long steps = (n_samples + 511) / 512;
    VADX_REQUIRE(steps < (1LL << 30), "vadx_silero_encode_pcm16: clip too long");
    return silero_encode_launch(packed, audio, scale, n_samples, row_stride, -64, batch, (int)steps, workspace, workspace_bytes, stream, cfg);
}

extern "C" int vadx_silero_encode_pcm16_part(const float *packed, const int16_t *audio, float scale, int batch, int64_t n_samples,
                                             int64_t row_stride, int first_clip, int total_batch, void *workspace,
                                             size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg) {
    VADX_REQUIRE(n_samples > 0 && row_stride >= n_samples, "vadx_silero_encode_pcm16_part: n_samples=%lld row_stride=%lld",
                 (long long)n_samples, (long long)row_stride);
    VADX_REQUIRE(first_clip >= 0 && first_clip % 16 == 0 && first_clip + batch <= total_batch,
                 "vadx_silero_encode_pcm16_part: first_clip=%d must be a multiple of 16 and first_clip + batch <= total_batch=%d",
                 first_clip, total_batch);
    const long long steps = (n_samples + 511) / 512;
    VADX_REQUIRE(steps < (1LL << 30), "vadx_silero_encode_pcm16_part: clip too long");
    return silero_encode_launch(packed, audio, scale, n_samples, row_stride, -64, batch, (int)steps, workspace, workspace_bytes, stream,
                                cfg, first_clip / 16, total_batch);
}

extern "C" int vadx_silero_recur(const float *packed, const void *workspace, size_t workspace_bytes, int batch,
                                 int steps, const float *state0, float *probs, float *state_n, void *stream, const vadx_silero_cfg *cfg) {
    return silero_recur_launch(packed, workspace, workspace_bytes, batch, steps, state0, probs, steps, state_n, stream, cfg);
}

extern "C" int vadx_silero_encode_span(const float *packed, const float *audio, int batch, int64_t n_samples,
                                       int64_t row_stride, int first_step, int n_steps, void *workspace,
                                       size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg) {
    VADX_REQUIRE(n_samples > 0 && row_stride >= n_samples, "vadx_silero_encode_span: n_samples=%lld row_stride=%lld",
                 (long long)n_samples, (long long)row_stride);
    const long long steps = (n_samples + 511) / 512;
    VADX_REQUIRE(first_step >= 0 && n_steps > 0 && (long long)first_step + n_steps <= steps,
                 "vadx_silero_encode_span: span [%d, %d + %d) outside the clip's %lld windows", first_step, first_step, n_steps, steps);
    return silero_encode_launch(packed, audio, 1.0f, n_samples, row_stride, (long long)first_step * 512 - 64, batch, n_steps,
                                workspace, workspace_bytes, stream, cfg);
}

extern "C" int vadx_silero_recur_span(const float *packed, const void *workspace, size_t workspace_bytes, int batch,
                                      int n_steps, const float *state0, float *probs, int64_t probs_stride,
                                      float *state_n, void *stream, const vadx_silero_cfg *cfg) {
    VADX_REQUIRE(probs_stride >= n_steps, "vadx_silero_recur_span: probs_stride=%lld < n_steps=%d", (long long)probs_stride, n_steps);
    return silero_recur_launch(packed, workspace, workspace_bytes, batch, n_steps, state0, probs, probs_stride, state_n, stream, cfg);
}

extern "C" int vadx_silero_step(const float *packed, const float *input, const float *state, int64_t sr,
                                int batch, float *out, float *state_n, void *workspace,
                                size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg) {
    VADX_REQUIRE(sr == 16000, "Supported sampling rates: [16000] (got %lld)", (long long)sr);
    VADX_REQUIRE(state && state_n, "vadx_silero_step: state / state_n must not be NULL");
    return silero_run(packed, input, 576, 576, 0, batch, 1, state, out, 1, state_n, workspace, workspace_bytes, stream, cfg);
}

extern "C" int vadx_silero_clips(const float *packed, const float *audio, int batch, int64_t n_samples,
                                 int64_t row_stride, float *probs, float *state_n, void *workspace,
                                 size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg) {
    VADX_REQUIRE(n_samples > 0 && row_stride >= n_samples, "vadx_silero_clips: n_samples=%lld row_stride=%lld",
                 (long long)n_samples, (long long)row_stride);
    const long long steps = (n_samples + 511) / 512;
    VADX_REQUIRE(steps < (1LL << 30), "vadx_silero_clips: clip too long");
    return silero_run(packed, audio, n_samples, row_stride, -64, batch, (int)steps, nullptr, probs, steps,
                      state_n, workspace, workspace_bytes, stream, cfg);
}

extern "C" int vadx_silero_segments(const float *probs, int batch, int steps, const int64_t *n_samples,
                                    const vadx_silero_seg_params *params, int64_t *segments,
                                    int32_t *counts, int cap, void *stream) {
    VADX_REQUIRE(probs && n_samples && params && segments && counts, "vadx_silero_segments: NULL pointer argument");
    VADX_REQUIRE(batch > 0 && steps > 0 && cap > 0, "vadx_silero_segments: batch/steps/cap must be positive");
    VADX_REQUIRE(params->sampling_rate == 16000 || params->sampling_rate == 8000,
                 "Currently silero VAD models support 8000 and 16000 (or multiply of 16000) sample rates");
    // SEG_CLIPS clips per wave: the state machine diverges per clip (a wave pays for every path its lanes take), and 64 clips per wave
    // leave three quarters of the CUs without work at 4096 clips
#ifndef SEG_CLIPS
#define SEG_CLIPS 16
#endif
    hipLaunchKernelGGL(silero_segments_kernel, dim3((batch + SEG_CLIPS - 1) / SEG_CLIPS), dim3(SEG_CLIPS), 0, static_cast<hipStream_t>(stream),
                       probs, batch, steps, reinterpret_cast<const long long *>(n_samples), *params,
                       reinterpret_cast<long long *>(segments), counts, cap);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

