// silero_common.h -- what the two Silero encoder kernels share: the packed-blob map, the staged-window layout, the STFT passes.
// (csrc/silero.hip: exact-f32 MFMA encoder; csrc/silero_split.hip: bf16 x 3 split-product encoder, csrc/split3.h)
#pragma once
#include "common.h"

namespace vadx {
namespace silero {

// ---- packed weight blob (float offsets) -------------------------------------------------------
// Encoder GEMM weights are stored FRAGMENT-MAJOR: [16-row tile][16-k block S][lane = 16q+i][4], i.e. exactly the
// f32x4 each lane feeds to the four MFMAs of block S (row 16*tile+i, k = 16S+4q+j).  One wave-wide load is then
// one contiguous 1 KB run (8 full cache lines) instead of 16 half-used lines of a row-major matrix.
constexpr int OFF_STFT = 0;                        // [8 waves][re16|im16][256]  (bins 0..127), k-permuted per 16
constexpr int OFF_NYQ = OFF_STFT + 256 * 256;      // [2][256]                   (bin 128 re, im)
constexpr int C1_KP = 128;                         // input channels 0..127 on MFMA; channel 128 (Nyquist) on VALU
// conv1 runs in the Winograd F(4,3) domain (see "phase 2"): 6 transformed weight planes U_j = G g instead of 3 taps
constexpr int OFF_C1 = OFF_NYQ + 2 * 256;          // [8 oc tiles][6 planes][8 blocks][FRAG]
constexpr int OFF_C1N = OFF_C1 + 128 * 6 * C1_KP;  // [128][8]  the six transformed taps of input channel 128 (+2 pad)
constexpr int OFF_B1 = OFF_C1N + 128 * 8;          // [128]
constexpr int OFF_C2 = OFF_B1 + 128;               // [4 oc tiles][3 taps x 8 blocks][FRAG]
constexpr int OFF_B2 = OFF_C2 + 64 * 3 * 128;      // [64]
constexpr int OFF_C3 = OFF_B2 + 64;                // [4 oc tiles][2 taps x 4 blocks][FRAG]  taps 1,2 (tap 0 only sees padding)
constexpr int OFF_B3 = OFF_C3 + 64 * 2 * 64;       // [64]
constexpr int OFF_C4 = OFF_B3 + 64;                // [8 oc tiles][4 blocks][FRAG]  tap 1 only
constexpr int OFF_B4 = OFF_C4 + 128 * 64;          // [128]
constexpr int OFF_IH = OFF_B4 + 128;               // [4 gates][8 unit tiles][8 blocks][FRAG]
constexpr int OFF_BG = OFF_IH + 512 * 128;         // [512] b_ih + b_hh
constexpr int OFF_HH = OFF_BG + 512;               // [512][128]
constexpr int OFF_DW = OFF_HH + 512 * 128;         // [128]
constexpr int OFF_DB = OFF_DW + 128;               // [1] (+3 pad)
// Folded STFT basis (used when the table has the DFT's time and frequency symmetries, see pack_host and
// stft_fold_class): symmetrised coefficients of bins 0..63 for the even-n / odd-n classes.
constexpr int OFF_SF = OFF_DB + 4;                 // [4 bin tiles][E|O][re|im][4 blocks][FRAG]
constexpr int OFF_S0 = OFF_SF + 4 * 2 * 2 * 4 * 256;   // [2][64]   the n = 0 column (re, im) of bins 0..63
constexpr int OFF_B64 = OFF_S0 + 128;              // [2][128] bin 64, time-folded (re, im), n = 1..128; then [4]: n = 0 (re, im)
constexpr int OFF_FOLD = OFF_B64 + 256 + 4;        // [1] (+3 pad)  1.0 = folded pass valid
// ---- split-product encoder (csrc/silero_split.hip): the same conv / W_ih weights as bf16 x 3 A fragments (csrc/split3.h: one
// fragment = QFRAG floats = 64 lanes x 8 bf16 of one (16-row tile, 32-k chunk, plane)), in the order each wave streams them.
// conv1 is the DIRECT three-tap conv here (each tap fragment serves up to four frames; the Winograd planes serve one each and the matrix
// time they save is cheap on this pipe), its input channels in the order the STFT pass leaves them: slot s <= 64 = bin s,
// slot 64 + k = bin 128 - k (k = 1..63); bin 128 (Nyquist) stays a VALU term.
constexpr int QF = 256;                                   // = vadx::QFRAG
// (every fragment section below starts on a 1 KiB boundary of the blob -- VADX_FRAG_ALIGN floats; a wave's fragment load is 64 lanes x 16 B =
//  1 KiB contiguous, and until round 6 the sections sat 48 B past a 64-byte line: every load touched 17 lines instead of 16, every 16-lane quarter
//  5 instead of 4 -- a quarter more L1 traffic for the kernel whose binding resource turned out to be L1 throughput)
#ifndef VADX_FRAG_ALIGN
#define VADX_FRAG_ALIGN 256
#endif
constexpr int OFF_Q1 = (OFF_FOLD + 4 + VADX_FRAG_ALIGN - 1) / VADX_FRAG_ALIGN * VADX_FRAG_ALIGN;      // [8 oc tiles][4 chunks][3 taps][3 planes][QF]
constexpr int OFF_Q1N = OFF_Q1 + 8 * 4 * 3 * 3 * QF;      // [128 oc][4]: taps 0..2 of input channel 128 (+1 pad), f32
constexpr int OFF_Q2 = OFF_Q1N + 128 * 4;                 // [4 oc tiles][4 chunks][3 taps][3 planes][QF]
constexpr int OFF_Q3 = OFF_Q2 + 4 * 4 * 3 * 3 * QF;       // [4 oc tiles][2 taps (1, 2)][2 chunks][3 planes][QF]
constexpr int OFF_Q4 = OFF_Q3 + 4 * 2 * 2 * 3 * QF;       // [8 oc tiles][2 chunks][3 planes][QF]   centre tap
constexpr int OFF_QIH = OFF_Q4 + 8 * 2 * 3 * QF;          // [8 unit tiles][4 chunks][4 gates][3 planes][QF]
constexpr int OFF_QHH = OFF_QIH + 8 * 4 * 4 * 3 * QF;     // [8 unit tiles][4 gates][4 chunks][3 planes][QF]   W_hh for the split recurrent kernel
// folded STFT basis (OFF_SF's coefficients) as split fragments: [4 bin tiles][E|O][re|im][2 chunks of 32 pairs][3 planes][QF]; pair m of
// the class = sample n = 2 m + 2 (E) / 2 m + 1 (O), in natural order
constexpr int OFF_QSF = OFF_QHH + 8 * 4 * 4 * 3 * QF;
// ---- fp16 x 2 encoder (csrc/silero_h2.hip, csrc/split2.h): the same weights as fragment PAIRS (h0 plane, h1 * 2^11 plane; one fragment =
// HF floats = 64 lanes x 8 fp16), in the order each wave streams them
constexpr int HF = 256;                                   // = vadx::HFRAG
constexpr int OFF_H1 = OFF_QSF + 4 * 2 * 2 * 2 * 3 * QF;  // conv1 [8 oc tiles][4 chunks][3 taps][2 planes][HF], input slots as OFF_Q1
constexpr int OFF_H2 = OFF_H1 + 8 * 4 * 3 * 2 * HF;       // conv2 [4 oc tiles][4 chunks][3 taps][2][HF]
constexpr int OFF_H3 = OFF_H2 + 4 * 4 * 3 * 2 * HF;       // conv3 [4 oc tiles][2 taps (1, 2)][2 chunks][2][HF]
constexpr int OFF_H4 = OFF_H3 + 4 * 2 * 2 * 2 * HF;       // conv4 [8 oc tiles][2 chunks][2][HF]   centre tap
constexpr int OFF_HIH = OFF_H4 + 8 * 2 * 2 * HF;          // W_ih [8 unit tiles][4 chunks][4 gates][2][HF]
constexpr int OFF_HHH = OFF_HIH + 8 * 4 * 4 * 2 * HF;     // W_hh [8 unit tiles][4 gates][4 chunks][2][HF]
// folded STFT basis: [5 bin tiles][E|O][re|im][2 chunks of 32 pairs][2][HF]; tiles 0..3 = bins 0..63 (OFF_SF's coefficients), tile 4 row 0 =
// bin 64 (OFF_B64's time-folded coefficients, rows 1..15 zero); pair m of the class = sample n = 2 m + 2 (E) / 2 m + 1 (O)
constexpr int OFF_HSF = OFF_HHH + 8 * 4 * 4 * 2 * HF;
// [0] 1.0 = the fp16 x 2 kernels may run on this blob (every weight inside the fp16 range, folded basis); [1] sticky range flag, written
// by the kernels as an unsigned (non-zero: some activation left the fp16 range, the batch must be recomputed on the bf16 x 3 kernels);
// [2] bits of the largest |activation| seen by a flagged workgroup; [3] pad
constexpr int OFF_HFLAG = OFF_HSF + 5 * 2 * 2 * 2 * 2 * HF;
constexpr int PACKED_FLOATS = OFF_HFLAG + 4;

constexpr int X_LDM = 642;            // staged window row: 576 samples + 64 reflect pad (+2: bank = 2 clip + q, conflict free)
// gx: per (t, group) 8 waves x 4 gates x 64 lanes x 4 floats
constexpr int GX_TILE_FLOATS = 8 * 4 * 256;

// m-major operand variant of gemm_pass for the STFT: act element (m, k) at act[m*ldm + k], this lane's
// m = lane&15, the k consumed by (block S, sub-step j, quarter q) is 16S + q + 4j (the basis rows are
// packed with the matching permutation, so weights still arrive as one 16-B load per block).
// SWAP: the table is the A operand, so D rows = table rows (bins) and D columns = act rows (clips) -- the orientation the split-product
// encoder's LDS planes want (a lane then holds four consecutive bins of one clip).  Same registers either way.
template <int NT, int MT, int KB, bool SWAP = false>
__device__ __forceinline__ void gemm_pass_mmajor(f32x4 (&acc)[NT][MT], const float *act, int ldm,
                                                 const int (&koff)[MT], const float *const (&wrow)[NT], int lane) {
    const int q = lane >> 4, i = lane & 15;
    const float *ap = act + i * ldm + q;
    f32x4 wcur[NT], wnxt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wcur[nt] = *reinterpret_cast<const f32x4 *>(wrow[nt] + 4 * q);
#pragma unroll 1
    for (int S = 0; S < KB; ++S) {
        const int Sn = (S + 1 < KB) ? S + 1 : S;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wnxt[nt] = *reinterpret_cast<const f32x4 *>(wrow[nt] + 16 * Sn + 4 * q);
        const float *aps = ap + 16 * S;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = aps[4 * j + koff[mt]];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float wj = wcur[nt][j];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = SWAP ? mfma16(wj, av[mt], acc[nt][mt]) : mfma16(av[mt], wj, acc[nt][mt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wcur[nt] = wnxt[nt];
    }
}

// Folded STFT pass.  A windowed real-DFT basis (symmetric window) has two symmetries that the dense 258x256
// conv ignores:
//   time      c[k][256-n] =  c[k][n],  s[k][256-n] = -s[k][n]        -> contract x[n] +- x[256-n] over n = 1..128
//   frequency c[128-k][n] = (-1)^n c[k][n],  s[128-k][n] = -(-1)^n s[k][n]
//                                                                    -> bins k and 128-k share the partial sums over
//                                                                       even n (E) and odd n (O): X[k] = E + O, X[128-k] = +-(E - O)
// so bins 0..63 (4 tiles) over two 64-long contractions give all of bins 0..63 and 65..128: a quarter of the dense
// pass's MFMAs, for two VALU adds per operand pair.  (Bin 64 pairs with itself and goes through the VALU.)
// X is staged de-interleaved: per clip row an even-sample plane [0..320] and an odd-sample plane [321..641], so the
// contraction index of either class walks its plane with unit stride (bank = 2*clip + q: conflict free).
// Class E: n = 2m + 2 (plane index m + 1, mirror 127 - m); class O: n = 2m + 1 (plane index m, mirror 127 - m);
// contraction slot (block S, sub-step j, quarter q) <-> m = 16S + q + 4j.
constexpr int X_ODD = 322;            // offset of the odd plane inside a clip row (even: the staging code stores sample pairs 8 B wide)
template <bool SWAP = false>
__device__ __forceinline__ void stft_fold_class(f32x4 (&are)[2], f32x4 (&aim)[2], const float *fwd, const float *rev,
                                                const float *wre, const float *wim) {
    f32x4 cre = *reinterpret_cast<const f32x4 *>(wre), cim = *reinterpret_cast<const f32x4 *>(wim);
#pragma unroll
    for (int S = 0; S < 4; ++S) {
        const int Sn = (S + 1 < 4) ? S + 1 : S;
        const f32x4 nre = *reinterpret_cast<const f32x4 *>(wre + FRAG * Sn);
        const f32x4 nim = *reinterpret_cast<const f32x4 *>(wim + FRAG * Sn);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (forming the two frames' operands as f32x2 pairs for v_pk_add_f32 -- half the adds -- costs more in register
            // shuffles than it saves: 291 instead of 199 VALU instructions in this loop)
            float e[2], o[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const float a = fwd[16 * S + 4 * j + 64 * f], b = rev[64 * f - 16 * S - 4 * j];
                e[f] = a + b;
                o[f] = a - b;
            }
#pragma unroll
            for (int f = 0; f < 2; ++f) are[f] = SWAP ? mfma16(cre[j], e[f], are[f]) : mfma16(e[f], cre[j], are[f]);
#pragma unroll
            for (int f = 0; f < 2; ++f) aim[f] = SWAP ? mfma16(cim[j], o[f], aim[f]) : mfma16(o[f], cim[j], aim[f]);
        }
        cre = nre;
        cim = nim;
    }
}

// |STFT| magnitude: the bare v_sqrt_f32 (1 ulp).  sqrtf() is the correctly rounded, denormal-safe library routine -- about fifteen
// VALU instructions per value -- and on gfx950 VALU work does not hide under v_mfma_f32_16x16x4_f32: the two ADD UP on a SIMD
// (tools/mfma_valu_overlap.sh: 15.2 ns per MFMA alone, +1.8 ns per v_fma_f32 placed beside it, one or two waves per SIMD alike).
__device__ __forceinline__ float mag_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Sample fetch of phase 0: the reference feeds Silero float32 = int16 * 0.000030517578 (Silero/Inference_Silero_VAD_ONNX.py:83);
// the PCM16 instantiation reads the int16 samples themselves (half the HBM read, no f32 copy of the batch) and applies
// that very multiplication -- one f32 rounding, bit-identical to the host-side product.
template <typename SampleT> struct SampleIO;
template <> struct SampleIO<float> {
    static constexpr int VEC_ALIGN = 16;
    static __device__ __forceinline__ f32x4 load4(const float *p, float) { return *reinterpret_cast<const f32x4 *>(p); }
    static __device__ __forceinline__ float load1(const float *p, float) { return *p; }
};
template <> struct SampleIO<int16_t> {
    static constexpr int VEC_ALIGN = 8;
    static __device__ __forceinline__ f32x4 load4(const int16_t *p, float scale) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        const s16x4 v = *reinterpret_cast<const s16x4 *>(p);
        return f32x4{(float)v[0] * scale, (float)v[1] * scale, (float)v[2] * scale, (float)v[3] * scale};
    }
    static __device__ __forceinline__ float load1(const int16_t *p, float scale) { return (float)*p * scale; }
};

// host: launch of the split-product encoder (csrc/silero_split.hip); arguments as silero_encode_kernel's
template <typename S>
int silero_encode_split_launch(const float *packed, const S *src, float in_scale, long long n_valid, long long row_stride,
                               long long origin, int batch, int G, int steps, int Gws, int first_group, float *gx, void *stream);

// csrc/silero_h2.hip: the fp16 x 2 encoder and recurrent kernels (arguments as the bf16 x 3 launches)
template <typename S>
int silero_encode_h2_launch(const float *packed, const S *src, float in_scale, long long n_valid, long long row_stride,
                            long long origin, int batch, int G, int steps, int Gws, int first_group, float *gx, void *stream);
int silero_lstm_h2_launch(const float *packed, const float *gx, const float *state0, int batch, int G, int steps, float *probs,
                          long long probs_stride, float *state_n, void *stream);

int silero_lstm_split_launch(const float *packed, const float *gx, const float *state0, int batch, int G, int steps, float *probs,
                             long long probs_stride, float *state_n, void *stream);

}  // namespace silero
}  // namespace vadx
