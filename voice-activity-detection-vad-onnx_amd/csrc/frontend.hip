// frontend.hip -- fused signal front-end for gfx950 (reference: the five STFT_Process.py copies +
// the wrapper code around them): int16 PCM -> prep (DC / pre-emphasis / scaling) -> framed
// windowed DFT against the REFERENCE'S OWN float32 table -> |.|^2 -> mel -> log, one kernel.
//
// Tile = one analysis window of one clip x 32 (or 16) consecutive frames; 512 threads.
//   * The PCM span of the tile is read from HBM once (coalesced int16), prepped in registers and
//     laid out in LDS as the hop-polyphase matrix  X2[r][g] = s'[g*hop + r]  (r < hop): frame f,
//     tap k = a*hop + r is then X2[r][f + a], i.e. the framed DFT becomes ceil(taps/hop) GEMM
//     passes that differ only by a COLUMN offset a -- overlapping frames are served from LDS and
//     every sample is stored exactly once.
//   * DFT = f32 MFMA GEMM [frames x taps] x [taps x 2F]; table rows stream from L2 into operand
//     registers (each wave owns whole bin tiles, re+im accumulate side by side so |.|^2 is lane-local).
//   * power spectrum goes to LDS k-major [bin][frame]; mel = second MFMA GEMM that only visits the
//     16-bin blocks each 16-mel tile actually touches (triangular filters are banded); log epilogue,
//     time-major output [window][frame][mel] (mel contiguous: LFR rows are contiguous slices).
#include "common.h"

#include <string.h>

#include <type_traits>

// FE_EXP: development-only cycle accounting (tools/exp_frontend.py)
#ifndef FE_EXP
#define FE_EXP 0
#endif
#if FE_EXP
__device__ unsigned long long fe_dbg[8];
#define FE_T0() long long fe_t_ = clock64()
#define FE_ACC(slot) do { if (threadIdx.x == 0) { const long long n_ = clock64(); atomicAdd(&fe_dbg[slot], (unsigned long long)(n_ - fe_t_)); fe_t_ = n_; } } while (0)
extern "C" int vadx_frontend_debug_cycles(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fe_dbg), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(fe_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define FE_T0() do {} while (0)
#define FE_ACC(slot) do {} while (0)
#endif

namespace vadx {
namespace frontend {

constexpr int THREADS = 512;
constexpr int TF = 32;            // frames per full tile (2 m-tiles); a 16-frame variant handles tails
constexpr int X_LD = 36;          // X2 row stride  (>= TF + max passes, % 8 == 4)
constexpr int P_LD = 36;          // power row stride (>= TF, % 8 == 4, 16-B aligned rows)
constexpr int MAX_PASSES = 4;
constexpr int MAX_MEL_TILES = 8;

struct Dev {
    // geometry
    int prep, center_pad, tap0, taps, hop, n_bins, n_mels, log_mode, frames, window_len;
    int in_len;             // samples per window in the SOURCE buffer (= window_len unless the graph resamples, prep 6 / 7)
    float k0, k1, log_floor, rs_scale;
    // derived
    int passes, pass_kb[MAX_PASSES], pass_koff[MAX_PASSES];   // 16-blocks per pass, k offset in table row
    int Kp;                 // padded taps per table row
    int nbt;                // MFMA bin tiles (16 bins each)
    int nyq;                // 1 => bin n_bins-1 (n_bins % 16 == 1) is its own 16-row tile (row 0 = re, row 1 = im), K-split over the waves
    int Fp;                 // padded bins (power rows)
    int nmt, mel_kb_lo[MAX_MEL_TILES], mel_kb_hi[MAX_MEL_TILES];
    int off_dft, off_nyq, off_mel;    // float offsets in the packed blob
    int tiles32, tiles16;   // per window: number of 32-frame tiles, then 16-frame tiles
    int out_stride, out_off; // floats per output frame row / first column (lets several streams share one row)
};

static int round16(int x) { return (x + 15) & ~15; }

static int derive(const vadx_frontend_cfg *c, Dev *d) {
    memset(d, 0, sizeof(*d));
    d->prep = c->prep; d->center_pad = c->center_pad; d->tap0 = c->tap0; d->taps = c->taps; d->hop = c->hop;
    d->n_bins = c->n_bins; d->n_mels = c->n_mels; d->log_mode = c->log_mode; d->frames = c->frames;
    d->window_len = c->window_len; d->k0 = c->k0; d->k1 = c->k1; d->log_floor = c->log_floor;
    d->in_len = c->in_window_len > 0 ? c->in_window_len : c->window_len;
    d->rs_scale = c->rs_scale;
    if ((c->prep == 6 || c->prep == 7) && !(c->rs_scale > 0.f && c->in_window_len > 0)) return -1;
    if (c->prep < 0 || c->prep > 7) return -1;
    if (c->hop <= 0 || c->hop > 320 || c->hop % 16 || c->taps <= 0 || c->n_bins <= 0 || c->n_mels <= 0 || c->n_mels % 16 ||
        c->frames <= 0 || c->window_len <= 0)
        return -1;
    d->passes = (c->taps + c->hop - 1) / c->hop;
    if (d->passes > MAX_PASSES || c->n_mels / 16 > MAX_MEL_TILES) return -1;
    int koff = 0;
    for (int a = 0; a < d->passes; ++a) {
        const int rows = (a + 1) * c->hop <= c->taps ? c->hop : c->taps - a * c->hop;
        d->pass_kb[a] = round16(rows) / 16;
        d->pass_koff[a] = koff;
        koff += round16(rows);
    }
    d->Kp = koff;
    d->nyq = (c->n_bins % 16 == 1) ? 1 : 0;
    d->nbt = d->nyq ? c->n_bins / 16 : (c->n_bins + 15) / 16;
    d->Fp = round16(c->n_bins);
    d->nmt = c->n_mels / 16;
    d->off_dft = 0;
    d->off_nyq = d->off_dft + d->nbt * 32 * d->Kp;
    d->off_mel = d->off_nyq + 16 * d->Kp;
    d->tiles32 = c->frames / TF;
    const int rem = c->frames - d->tiles32 * TF;
    d->tiles16 = (rem + 15) / 16;
    if (rem > 16) { d->tiles32 += 1; d->tiles16 = 0; }
    d->out_stride = c->n_mels; d->out_off = 0;
    return 0;
}

// log of the mel energies: v_log_f32 (1 ulp in log2) times ln 2 instead of the library logf (~25 VALU instructions per value; VALU
// time adds to f32-MFMA time on gfx950).  |difference| <= 4e-6 over the feature range, two orders inside the feature tolerance.
#ifndef FE_FAST_LOG
#define FE_FAST_LOG 1
#endif
__device__ __forceinline__ float FE_LOG(float x) { return FE_FAST_LOG ? __builtin_amdgcn_logf(x) * 0.6931471805599453f : logf(x); }

struct FtOut { float *ptr; int tile0, c_total, c_off; };      // COMPLEX: FT destination (re -> c_off, im -> c_off+1)

template <int MT, bool COMPLEX = false>
__device__ __forceinline__ void tile_body(const Dev &d, const float *__restrict__ P, const int16_t *__restrict__ win,
                                          const float *__restrict__ fwin, float mean, int f0, float *__restrict__ out_win,
                                          float *X2, float *PW, FtOut ft = FtOut{nullptr, 0, 0, 0}) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, i = lane & 15;
    constexpr int NF = MT * 16;
    const int cols = NF + d.passes - 1;

    FE_T0();
    // ---- phase 0: prep + polyphase staging  X2[r][g] = s'[(f0+g)*hop + r]
    // wave = column g, lane = row r (+64 j): consecutive lanes read consecutive samples, no divisions.  All loads of a
    // column are issued first and UNCONDITIONALLY (indices clamped, values selected afterwards; which sources exist is
    // decided once, outside the loop): a load under any condition -- even a uniform one -- compiles to a branch plus a
    // full wait, which serialised ~100 memory round trips per tile (29 % of the kernel).
    // NJ = rows per lane (64 NJ >= hop): three for the 160-sample hops of FSMN / MarbleNet / FireRed, five up to hop 320 (DFSMN) --
    // with five everywhere, two fifths of the staging instructions of the 160-hop models produced rows that were never stored
    auto stage = [&](auto prep_c, auto nj_c) {
        constexpr int PREP = decltype(prep_c)::value, NJ = decltype(nj_c)::value;      // compile-time prep: ~15 instructions per sample
        constexpr bool HW = PREP != 4, HF = PREP >= 4;
        for (int g = wave; g < cols; g += THREADS / 64) {
            const int nb = (f0 + g) * d.hop + d.tap0 - d.center_pad;      // window index of row 0
            float xs[NJ], xms[NJ], as_[NJ], ams[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {                                 // hop <= 64 NJ
                const int n = nb + lane + 64 * j;
                const int nc = n < 0 ? 0 : (n >= d.window_len ? d.window_len - 1 : n), nm = nc > 0 ? nc - 1 : 0;
                xs[j] = HW ? (float)win[nc] : 0.f;
                xms[j] = HW ? (float)win[nm] : 0.f;
                as_[j] = HF ? fwin[nc] : 0.f;
                ams[j] = HF ? fwin[nm] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int r = lane + 64 * j, n = nb + r;
                const bool in = n >= 0 && n < d.window_len;
                const float x = xs[j], xm = n > 0 ? xms[j] : 0.f;
                float v;
                if (PREP == 0) {              // FSMN: (x-mean) - 0.97*(x[-1]-mean), first sample kept
                    const float a = __fsub_rn(x, mean);
                    v = (n > 0) ? __fsub_rn(a, __fmul_rn(0.97f, __fsub_rn(xm, mean))) : a;
                } else if (PREP == 1) {       // two-tap conv with zero history
                    v = __fadd_rn(__fmul_rn(xm, d.k0), __fmul_rn(x, d.k1));
                } else if (PREP == 2) {       // scale, then remove the window mean (mean is of the scaled signal)
                    v = __fsub_rn(__fmul_rn(x, d.k1), mean);
                } else {                      // DFSMN feature streams (Export_DFSMN_VAD.py:338-341)
                    float npe = 0.f, ape = 0.f;
                    if (PREP != 4) {          // near: a = k1*x - mean, pre-emphasis keeping a[0]
                        const float a = __fsub_rn(__fmul_rn(x, d.k1), mean);
                        npe = (n > 0) ? __fsub_rn(a, __fmul_rn(0.97f, __fsub_rn(__fmul_rn(xm, d.k1), mean))) : a;
                    }
                    if (PREP != 3) {          // AEC output (float source), same pre-emphasis
                        const float a = as_[j];
                        ape = (n > 0) ? __fsub_rn(a, __fmul_rn(0.97f, ams[j])) : a;
                    }
                    v = PREP == 3 ? npe : (PREP == 4 ? ape : __fsub_rn(npe, __fmul_rn(d.k0, ape)));   // 5: echo = near - k0*aec
                }
                if (r < d.hop) X2[r * X_LD + g] = in ? v : 0.f;
            }
        }
    };
    // In-graph linear resampling of exports built with IN_SAMPLE_RATE != 16000 (Export_NVIDIA_MarbleNet_VAD.py:237-254,
    // FireRedVAD/Export_FireRedVAD.py:431-449): F.interpolate(mode='linear', align_corners=False, scale_factor=16000/in_rate)
    // BEFORE the two-tap pre-emphasis when the input rate is higher (prep 6), AFTER it when it is lower (prep 7).  The window in
    // HBM holds in_len samples at the input rate; output sample n reads source position max(0, rs_scale * (n + 0.5) - 0.5)
    // (float32 arithmetic, as torch's area_pixel_compute_source_index), neighbours clamped to the window like torch clamps them.
    auto stage_rs = [&](auto prep_c, auto nj_c) {
        constexpr int PREP = decltype(prep_c)::value, NJ = decltype(nj_c)::value;
        auto src = [&](int n, int &i0, int &i1, float &lam) {
            float s = __fsub_rn(__fmul_rn(d.rs_scale, __fadd_rn((float)n, 0.5f)), 0.5f);
            s = s < 0.f ? 0.f : s;
            i0 = (int)s;
            i0 = i0 > d.in_len - 1 ? d.in_len - 1 : i0;
            i1 = i0 + (i0 < d.in_len - 1 ? 1 : 0);
            lam = __fsub_rn(s, (float)i0);
        };
        for (int g = wave; g < cols; g += THREADS / 64) {
            const int nb = (f0 + g) * d.hop + d.tap0 - d.center_pad;
            float xa[NJ], xb[NJ], xc[NJ], xd[NJ], la[NJ], lb[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nb + lane + 64 * j;
                const int nc = n < 0 ? 0 : (n >= d.window_len ? d.window_len - 1 : n);
                int i0, i1, j0, j1;
                src(nc, i0, i1, la[j]);
                if (PREP == 6) {              // r(n) and r(n-1): four source samples
                    src(nc > 0 ? nc - 1 : 0, j0, j1, lb[j]);
                    xa[j] = (float)win[i0]; xb[j] = (float)win[i1]; xc[j] = (float)win[j0]; xd[j] = (float)win[j1];
                } else {                      // p(i0) and p(i1): x[i-1], x[i] of both neighbours
                    lb[j] = 0.f;
                    xa[j] = (float)win[i0]; xb[j] = (float)win[i1];
                    const float pc = (float)win[i0 > 0 ? i0 - 1 : 0], pd = (float)win[i1 > 0 ? i1 - 1 : 0];      // unconditional loads from
                    xc[j] = i0 > 0 ? pc : 0.f;                                                                       // clamped indices, zero
                    xd[j] = i1 > 0 ? pd : 0.f;                                                                       // selected afterwards
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int r = lane + 64 * j, n = nb + r;
                const bool in = n >= 0 && n < d.window_len;
                const float l1 = la[j], l0 = __fsub_rn(1.f, l1);
                float v;
                if (PREP == 6) {
                    const float rn = __fadd_rn(__fmul_rn(l0, xa[j]), __fmul_rn(l1, xb[j]));
                    const float m1 = lb[j], m0 = __fsub_rn(1.f, m1);
                    const float rm = n > 0 ? __fadd_rn(__fmul_rn(m0, xc[j]), __fmul_rn(m1, xd[j])) : 0.f;
                    v = __fadd_rn(__fmul_rn(rm, d.k0), __fmul_rn(rn, d.k1));
                } else {
                    const float p0 = __fadd_rn(__fmul_rn(xc[j], d.k0), __fmul_rn(xa[j], d.k1));
                    const float p1 = __fadd_rn(__fmul_rn(xd[j], d.k0), __fmul_rn(xb[j], d.k1));
                    v = __fadd_rn(__fmul_rn(l0, p0), __fmul_rn(l1, p1));
                }
                if (r < d.hop) X2[r * X_LD + g] = in ? v : 0.f;
            }
        }
    };
    auto stage_any = [&](auto nj_c) {
        switch (d.prep) {       // uniform
            case 6: stage_rs(std::integral_constant<int, 6>{}, nj_c); break;
            case 7: stage_rs(std::integral_constant<int, 7>{}, nj_c); break;
            case 0: stage(std::integral_constant<int, 0>{}, nj_c); break;
            case 1: stage(std::integral_constant<int, 1>{}, nj_c); break;
            case 2: stage(std::integral_constant<int, 2>{}, nj_c); break;
            case 3: stage(std::integral_constant<int, 3>{}, nj_c); break;
            case 4: stage(std::integral_constant<int, 4>{}, nj_c); break;
            default: stage(std::integral_constant<int, 5>{}, nj_c); break;
        }
    };
    if (d.hop <= 192) stage_any(std::integral_constant<int, 3>{});
    else stage_any(std::integral_constant<int, 5>{});
    FE_ACC(4);
    __syncthreads();
    FE_ACC(0);

    // ---- phase 1: DFT GEMM, |.|^2 -> PW[bin][frame]
    for (int bt = wave; bt < d.nbt; bt += THREADS / 64) {
        f32x4 acc[2][MT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *re_row = vadx::frag_ptr(P + d.off_dft, d.Kp, bt * 2, 0, lane);      // tiles: [bt][re | im]
        const float *im_row = vadx::frag_ptr(P + d.off_dft, d.Kp, bt * 2 + 1, 0, lane);
        for (int a = 0; a < d.passes; ++a) {
            const float *const wrow[2] = {re_row + d.pass_koff[a] * 16, im_row + d.pass_koff[a] * 16};
            int moff[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) moff[mt] = mt * 16 + a;
            vadx::gemm_rt_simple<2, MT, false>(acc, X2, X_LD, moff, wrow, d.pass_kb[a], lane);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (COMPLEX) {           // raw spectrum into the frame-tiled layout [tile][c][bin][16 frames]
                const int bin = bt * 16 + i;
                if (bin < d.n_bins) {
                    const size_t base = (((size_t)(ft.tile0 + (f0 >> 4) + mt) * ft.c_total + ft.c_off) * d.n_bins + bin) * 16 + 4 * q;
                    *reinterpret_cast<f32x4 *>(ft.ptr + base) = acc[0][mt];
                    *reinterpret_cast<f32x4 *>(ft.ptr + base + (size_t)d.n_bins * 16) = acc[1][mt];
                }
                continue;
            }
            f32x4 pw;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pw[r] = __fadd_rn(__fmul_rn(acc[0][mt][r], acc[0][mt][r]), __fmul_rn(acc[1][mt][r], acc[1][mt][r]));
            *reinterpret_cast<f32x4 *>(&PW[(bt * 16 + i) * P_LD + mt * 16 + 4 * q]) = pw;
        }
    }
    if (COMPLEX) return;
    FE_ACC(1);
    if (d.nyq) {
        // Last bin (n_bins % 16 == 1, e.g. the Nyquist bin of a 512-point DFT): a 17th bin tile would hand one wave three
        // tiles instead of two (+50 % on the phase), and as a VALU dot product over all taps it cost a fifth of the kernel
        // (cycle accounting: "last bin" 21 % for the FSMN geometry).  It is ONE 16-row MFMA tile instead -- row 0 = the bin's
        // real table row, row 1 = its imaginary row -- whose K blocks are dealt round-robin to the eight waves (three or four
        // blocks each: 1/32 more MFMAs, evenly spread); the per-wave partial sums meet in the padding rows of PW.
        f32x4 nacc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) nacc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nblk = d.Kp / 16;
        for (int b = wave; b < nblk; b += THREADS / 64) {
            int a = 0, b0 = 0;
            while (a + 1 < d.passes && b >= b0 + d.pass_kb[a]) { b0 += d.pass_kb[a]; ++a; }
            const f32x4 w = vadx::ldg4(P + d.off_nyq + (size_t)b * vadx::FRAG + lane * 4);
            const float *aps = X2 + (16 * (b - b0) + 4 * q) * X_LD + i + a;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) nacc[mt] = vadx::mfma16(aps[j * X_LD + mt * 16], w[j], nacc[mt]);
        }
        float *scr = PW + (d.n_bins) * P_LD;               // rows n_bins .. Fp-1 (15 x P_LD floats >= 8 waves x 2 x NF)
        if (i < 2)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                *reinterpret_cast<f32x4 *>(scr + (wave * 2 + i) * NF + mt * 16 + 4 * q) = nacc[mt];
        for (int e = tid + 16 * NF; e < (d.Fp - d.n_bins) * P_LD; e += THREADS) scr[e] = 0.f;    // the rows' tail: finite for the mel GEMM
        __syncthreads();
        if (tid < NF) {
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < THREADS / 64; ++w8) { re += scr[(w8 * 2) * NF + tid]; im += scr[(w8 * 2 + 1) * NF + tid]; }
            PW[(d.n_bins - 1) * P_LD + tid] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
        }
        // (the padding rows keep the finite partial sums: their mel weights are exactly zero)
    } else {
        // zero the padded power rows (bins n_bins..Fp-1) so the mel GEMM multiplies 0 x 0
        for (int e = tid; e < (d.Fp - d.n_bins) * NF; e += THREADS) {
            const int r = e / NF, c = e - r * NF;
            PW[(d.n_bins + r) * P_LD + c] = 0.f;
        }
    }
    __syncthreads();
    FE_ACC(2);

    // ---- phase 2: banded mel GEMM + log, D rows = mel (SWAP) so each lane stores 4 consecutive mels
    for (int mtile = wave; mtile < d.nmt; mtile += THREADS / 64) {
        f32x4 acc[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int lo = d.mel_kb_lo[mtile], hi = d.mel_kb_hi[mtile];
        const float *const wrow[1] = {vadx::frag_ptr(P + d.off_mel, d.Fp, mtile, lo * 16, lane)};
        int moff[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) moff[mt] = mt * 16;
        if (hi > lo) vadx::gemm_rt_simple<1, MT, true>(acc, PW + lo * 16 * P_LD, P_LD, moff, wrow, hi - lo, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int f = f0 + mt * 16 + i;
            if (f < d.frames) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m = acc[0][mt][r];
                    v[r] = FE_LOG(d.log_mode ? __fadd_rn(m, d.log_floor) : fmaxf(m, d.log_floor));
                }
                *reinterpret_cast<f32x4 *>(out_win + (size_t)f * d.out_stride + d.out_off + mtile * 16 + 4 * q) = v;
            }
        }
    }
    FE_ACC(3);
}

__global__ __launch_bounds__(THREADS, 4) void frontend_logmel_kernel(
    Dev d, const float *__restrict__ P, const int16_t *__restrict__ audio, long long row_stride,
    long long win_stride, int windows_per_clip, const float *__restrict__ means, float *__restrict__ out,
    const float *__restrict__ faux) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *X2 = lds;
    float *PW = lds + d.hop * X_LD;                   // a pass never reads past row hop - 1: its rows round up to 16 <= hop (hop % 16 == 0)
    const int tiles = d.tiles32 + d.tiles16;
    const int widx = blockIdx.x / tiles, tile = blockIdx.x - widx * tiles;
    const int b = widx / windows_per_clip, w = widx - b * windows_per_clip;
    const int16_t *win = audio ? audio + (long long)b * row_stride + (long long)w * win_stride : nullptr;
    const float *fwin = faux ? faux + (size_t)widx * d.window_len : nullptr;      // float source: [window][window_len]
    float *out_win = out + (size_t)widx * d.frames * d.out_stride;
    const float mean = means ? means[widx] : 0.f;
    if (tile < d.tiles32) tile_body<2>(d, P, win, fwin, mean, tile * TF, out_win, X2, PW);
    else tile_body<1>(d, P, win, fwin, mean, d.tiles32 * TF + (tile - d.tiles32) * 16, out_win, X2, PW);
}

// raw complex STFT into the FT layout (DFSMN's two-stream STFT-B): every tile is 32 frames = 2 FT tiles
__global__ __launch_bounds__(THREADS, 4) void stft_complex_kernel(
    Dev d, const float *__restrict__ P, const int16_t *__restrict__ audio, long long row_stride,
    long long win_stride, int windows_per_clip, const float *__restrict__ means, float *__restrict__ ft_out,
    int ft_nt, int ft_ctotal, int ft_coff) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *X2 = lds;
    float *PW = lds + d.hop * X_LD;
    const int tiles = (d.frames + TF - 1) / TF;
    const int widx = blockIdx.x / tiles, tile = blockIdx.x - widx * tiles;
    const int b = widx / windows_per_clip, w = widx - b * windows_per_clip;
    const int16_t *win = audio + (long long)b * row_stride + (long long)w * win_stride;
    const float mean = means ? means[widx] : 0.f;
    FtOut ft{ft_out, widx * ft_nt, ft_ctotal, ft_coff};
    if (tile * TF + 16 < ft_nt * 16) tile_body<2, true>(d, P, win, nullptr, mean, tile * TF, nullptr, X2, PW, ft);
    else tile_body<1, true>(d, P, win, nullptr, mean, tile * TF, nullptr, X2, PW, ft);
}

// window means for the DC-removing preps: exact integer sum -> float (one wave per window)
__global__ void window_mean_kernel(const int16_t *__restrict__ audio, long long row_stride, long long win_stride,
                                   int windows_per_clip, int n_windows, int window_len, float scale,
                                   float *__restrict__ means) {
    const int widx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (widx >= n_windows) return;
    const int lane = threadIdx.x & 63;
    const int b = widx / windows_per_clip, w = widx - b * windows_per_clip;
    const int16_t *win = audio + (long long)b * row_stride + (long long)w * win_stride;
    long long s = 0;
    for (int n = lane; n < window_len; n += 64) s += win[n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) means[widx] = (float)((double)s * (double)scale / (double)window_len);
}

}  // namespace frontend
}  // namespace vadx

using namespace vadx::frontend;

extern "C" size_t vadx_frontend_packed_floats(const vadx_frontend_cfg *cfg) {
    Dev d;
    if (!cfg || derive(cfg, &d)) return 0;
    return (size_t)d.off_mel + (size_t)d.n_mels * d.Fp;
}

// Tables on the host in the reference's own layout: cos_tab/sin_tab [n_bins][n_fft] (windowed, the
// float32 values the reference registers as conv kernels), fbank [n_mels][n_bins].
extern "C" int vadx_frontend_pack_host(const vadx_frontend_cfg *cfg, const float *cos_tab, const float *sin_tab,
                                       int n_fft, const float *fbank, float *packed_host, int32_t *mel_kb) {
    Dev d;
    VADX_REQUIRE(cfg && cos_tab && sin_tab && fbank && packed_host && mel_kb, "vadx_frontend_pack_host: NULL argument");
    VADX_REQUIRE(derive(cfg, &d) == 0, "vadx_frontend_pack_host: unsupported geometry (hop %% 16, n_mels %% 16, passes <= 4)");
    VADX_REQUIRE(cfg->tap0 >= 0 && cfg->tap0 + cfg->taps <= n_fft, "vadx_frontend_pack_host: taps outside n_fft");
    const size_t total = (size_t)d.off_mel + (size_t)d.n_mels * d.Fp;
    memset(packed_host, 0, total * sizeof(float));
    // taps outside [tap0, tap0+taps) must be zero in the reference table (centre-padded window)
    for (int f = 0; f < d.n_bins; ++f)
        for (int t = 0; t < n_fft; ++t)
            if ((t < cfg->tap0 || t >= cfg->tap0 + cfg->taps) && (cos_tab[(size_t)f * n_fft + t] != 0.f || sin_tab[(size_t)f * n_fft + t] != 0.f)) {
                vadx::set_error("vadx_frontend_pack_host: table has a non-zero tap %d outside [%d,%d)", t, cfg->tap0, cfg->tap0 + cfg->taps);
                return VADX_EINVAL;
            }
    auto put_row = [&](float *dst, const float *src_row) {     // [Kp] <- taps regrouped per pass
        for (int a = 0; a < d.passes; ++a) {
            const int rows = (a + 1) * d.hop <= d.taps ? d.hop : d.taps - a * d.hop;
            memcpy(dst + d.pass_koff[a], src_row + cfg->tap0 + a * d.hop, rows * sizeof(float));
        }
    };
    for (int bt = 0; bt < d.nbt; ++bt)
        for (int i = 0; i < 16; ++i) {
            const int f = bt * 16 + i;
            if (f >= d.n_bins) continue;
            put_row(packed_host + d.off_dft + (size_t)(bt * 32 + i) * d.Kp, cos_tab + (size_t)f * n_fft);
            put_row(packed_host + d.off_dft + (size_t)(bt * 32 + 16 + i) * d.Kp, sin_tab + (size_t)f * n_fft);
        }
    if (d.nyq) {            // its own 16-row tile: row 0 = real table row, row 1 = imaginary, rows 2..15 zero
        put_row(packed_host + d.off_nyq, cos_tab + (size_t)(d.n_bins - 1) * n_fft);
        put_row(packed_host + d.off_nyq + d.Kp, sin_tab + (size_t)(d.n_bins - 1) * n_fft);
    }
    for (int m = 0; m < d.n_mels; ++m)
        memcpy(packed_host + d.off_mel + (size_t)m * d.Fp, fbank + (size_t)m * d.n_bins, d.n_bins * sizeof(float));
    for (int mt = 0; mt < d.nmt; ++mt) {       // banded: first/last 16-bin block with a non-zero weight
        int lo = d.Fp / 16, hi = 0;
        for (int m = mt * 16; m < mt * 16 + 16; ++m)
            for (int f = 0; f < d.n_bins; ++f)
                if (fbank[(size_t)m * d.n_bins + f] != 0.f) {
                    if (f / 16 < lo) lo = f / 16;
                    if (f / 16 + 1 > hi) hi = f / 16 + 1;
                }
        if (hi < lo) { lo = 0; hi = 0; }
        mel_kb[2 * mt] = lo;
        mel_kb[2 * mt + 1] = hi;
    }
    // GEMM operands go fragment-major (common.h)
    vadx::frag_major_inplace(packed_host + d.off_dft, d.nbt * 32, d.Kp);
    if (d.nyq) vadx::frag_major_inplace(packed_host + d.off_nyq, 16, d.Kp);
    vadx::frag_major_inplace(packed_host + d.off_mel, d.n_mels, d.Fp);
    return VADX_OK;
}

extern "C" int vadx_frontend_logmel(const vadx_frontend_cfg *cfg, const float *packed, const int32_t *mel_kb_host,
                                    const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch,
                                    int windows_per_clip, float *means_ws, float *out, void *stream) {
    Dev d;
    VADX_REQUIRE(cfg && packed && mel_kb_host && audio && out, "vadx_frontend_logmel: NULL argument");
    VADX_REQUIRE(derive(cfg, &d) == 0, "vadx_frontend_logmel: unsupported geometry");
    VADX_REQUIRE(batch > 0 && windows_per_clip > 0, "vadx_frontend_logmel: batch/windows must be positive");
    VADX_REQUIRE((windows_per_clip - 1) * win_stride + d.in_len <= row_stride,
                 "vadx_frontend_logmel: windows run past the clip row (pad the clip to the window grid first)");
    VADX_REQUIRE(cfg->prep == 1 || cfg->prep >= 6 || means_ws, "vadx_frontend_logmel: this prep mode needs a means workspace of batch*windows floats");
    VADX_REQUIRE(TF + d.passes - 1 <= X_LD, "vadx_frontend_logmel: too many passes");
    for (int mt = 0; mt < d.nmt; ++mt) { d.mel_kb_lo[mt] = mel_kb_host[2 * mt]; d.mel_kb_hi[mt] = mel_kb_host[2 * mt + 1]; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long nwin = (long long)batch * windows_per_clip;
    VADX_REQUIRE(nwin * (d.tiles32 + d.tiles16) < (1LL << 31), "vadx_frontend_logmel: too many tiles");
    const float *means = nullptr;
    if (cfg->prep != 1 && cfg->prep < 6) {
        const float scale = (cfg->prep == 2) ? cfg->k1 : 1.0f;
        hipLaunchKernelGGL(window_mean_kernel, dim3((unsigned)((nwin + 3) / 4)), dim3(256), 0, st, audio, (long long)row_stride,
                           (long long)win_stride, windows_per_clip, (int)nwin, cfg->window_len, scale, means_ws);
        VADX_HIP_TRY(hipGetLastError());
        means = means_ws;
    }
    const size_t lds = ((size_t)d.hop * X_LD + (size_t)d.Fp * P_LD) * sizeof(float);      // FSMN / FireRed: 52 992 B -- three workgroups per CU (it was 2.3 KB more: two)
    VADX_REQUIRE(lds <= 160 * 1024, "vadx_frontend_logmel: geometry needs %zu B of LDS", lds);
    if (lds > 64 * 1024)
        VADX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(frontend_logmel_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(frontend_logmel_kernel, dim3((unsigned)(nwin * (d.tiles32 + d.tiles16))), dim3(THREADS), lds, st, d,
                       packed, audio, (long long)row_stride, (long long)win_stride, windows_per_clip, means, out,
                       static_cast<const float *>(nullptr));
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}


// DFSMN feature streams: same fused kernel, sources = int16 near window (+ its scaled mean) and/or the float AEC
// waveform [windows][window_len]; writes n_mels columns at out_off of rows of out_stride floats.
extern "C" int vadx_frontend_logmel_ex(const vadx_frontend_cfg *cfg, const float *packed, const int32_t *mel_kb_host,
                                       const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch,
                                       int windows_per_clip, const float *means, const float *faux, int out_stride,
                                       int out_off, float *out, void *stream) {
    Dev d;
    VADX_REQUIRE(cfg && packed && mel_kb_host && out, "vadx_frontend_logmel_ex: NULL argument");
    VADX_REQUIRE(derive(cfg, &d) == 0, "vadx_frontend_logmel_ex: unsupported geometry");
    VADX_REQUIRE(cfg->prep >= 3 && cfg->prep <= 5, "vadx_frontend_logmel_ex: prep must be 3 (near), 4 (aec) or 5 (echo)");
    VADX_REQUIRE((cfg->prep == 4 || (audio && means)) && (cfg->prep == 3 || faux), "vadx_frontend_logmel_ex: missing source");
    VADX_REQUIRE(out_stride >= out_off + cfg->n_mels && out_stride % 4 == 0 && out_off % 4 == 0, "vadx_frontend_logmel_ex: bad output stride/offset");
    for (int mt = 0; mt < d.nmt; ++mt) { d.mel_kb_lo[mt] = mel_kb_host[2 * mt]; d.mel_kb_hi[mt] = mel_kb_host[2 * mt + 1]; }
    d.out_stride = out_stride; d.out_off = out_off;
    const long long nwin = (long long)batch * windows_per_clip;
    const size_t lds = ((size_t)d.hop * X_LD + (size_t)d.Fp * P_LD) * sizeof(float);      // FSMN / FireRed: 52 992 B -- three workgroups per CU (it was 2.3 KB more: two)
    VADX_REQUIRE(lds <= 160 * 1024, "vadx_frontend_logmel_ex: geometry needs %zu B of LDS", lds);
    if (lds > 64 * 1024)
        VADX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(frontend_logmel_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(frontend_logmel_kernel, dim3((unsigned)(nwin * (d.tiles32 + d.tiles16))), dim3(THREADS), lds,
                       static_cast<hipStream_t>(stream), d, packed, audio, (long long)row_stride, (long long)win_stride,
                       windows_per_clip, means, out, faux);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

// Raw complex STFT of int16 windows into an FT tensor [window*ft_nt + t/16][c_total][n_bins][16] at channels
// c_off (re) and c_off+1 (im); prep 2 (scale + window-mean removal).  means_ws as in vadx_frontend_logmel.
extern "C" int vadx_frontend_stft_ft(const vadx_frontend_cfg *cfg, const float *packed, const int16_t *audio,
                                     int64_t row_stride, int64_t win_stride, int batch, int windows_per_clip,
                                     float *means_ws, float *ft_out, int c_total, int c_off, void *stream) {
    Dev d;
    VADX_REQUIRE(cfg && packed && audio && means_ws && ft_out, "vadx_frontend_stft_ft: NULL argument");
    VADX_REQUIRE(derive(cfg, &d) == 0 && d.nyq == 0, "vadx_frontend_stft_ft: unsupported geometry");
    VADX_REQUIRE(cfg->prep == 2, "vadx_frontend_stft_ft: prep must be 2");
    VADX_REQUIRE((windows_per_clip - 1) * win_stride + cfg->window_len <= row_stride, "vadx_frontend_stft_ft: windows run past the clip row");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long nwin = (long long)batch * windows_per_clip;
    hipLaunchKernelGGL(window_mean_kernel, dim3((unsigned)((nwin + 3) / 4)), dim3(256), 0, st, audio, (long long)row_stride,
                       (long long)win_stride, windows_per_clip, (int)nwin, cfg->window_len, cfg->k1, means_ws);
    VADX_HIP_TRY(hipGetLastError());
    const int ft_nt = (d.frames + 15) / 16, tiles = (d.frames + TF - 1) / TF;
    const size_t lds = ((size_t)d.hop * X_LD) * sizeof(float);
    hipLaunchKernelGGL(stft_complex_kernel, dim3((unsigned)(nwin * tiles)), dim3(THREADS), lds, st, d, packed, audio,
                       (long long)row_stride, (long long)win_stride, windows_per_clip, means_ws, ft_out, ft_nt, c_total, c_off);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
