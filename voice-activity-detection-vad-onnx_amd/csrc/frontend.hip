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
// Three kernels share the staging and the mel + log phase: the dense product above (frontend_logmel_kernel: DFSMN's feature streams,
// the raw STFT, any geometry without a fold plan), and two FOLDED products of the log-mel path on 64-frame tiles
// (frontend_fold_kernel: mirror-paired taps + f16 residual, kinds 1 / 2, the default of the FSMN / MarbleNet / FireRed front-ends;
// frontend_fold3_kernel: time x frequency fold, opt-in) -- see "Folded DFT" below.
#include "common.h"
#include "layers_split.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include <type_traits>

// FE_EXP: development-only cycle accounting (tools/exp_frontend.py)
#ifndef FE_EXP
#define FE_EXP 0
#endif
// FE_WHATIF: development-only switches of the folded kernel (results are wrong when set): 1 no residual product, 2 every table fragment
// from block 0 (L1 instead of L2), 4 no u / v additions, 8 no symmetric product
#ifndef FE_WHATIF
#define FE_WHATIF 0
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
constexpr int MAX_REGIONS = 8;
constexpr int XF_LD = 68;         // folded kernel: X2 and power row stride (64 frames + passes - 1 <= 67, % 8 == 4)
constexpr int TF_FOLD = 64;       // frames per full tile of the folded kernel
constexpr double FOLD_MAX_RATIO = 1.5e-4;   // residual / table scale the f16 product may carry (x 2^-11 each operand: ~1.5e-7)
constexpr float RES_SCALE = 8192.f;   // residual table entries are stored as f16(value * 2^13)

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
    // folded DFT (cfg.fold != 0): pair regions of the symmetric part, the f16 residual, the 64-frame tiling
    int fold, f_regions, f_blocks[MAX_REGIONS], f_offA[MAX_REGIONS], f_offB[MAX_REGIONS], f_strB[MAX_REGIONS];
    int f_Pb, f_Kb32;       // 16-pair blocks of the symmetric part / 32-tap blocks of the residual
    int off_fold, off_res, off_plan;  // float offsets of the folded tables in the blob; off_plan: int32 [region][blocks, offA, offB, strB], then [mel tile][lo, hi]
                                      // (the kernel reads the regions from there: indexing the by-value Dev arrays dynamically costs scratch)
    float f_xscale, f_rinv; // power-of-two scale of the f16 samples; 1 / (f_xscale * RES_SCALE)
    int tiles64, tail_mt;   // 64-frame tiles of a window, then ONE tail tile of tail_mt m-tiles (0: none; 1 - 3: 16 - 48 frames)
    // split-product dense DFT (cfg.fold == 4): A fragments of the reference table at off_fold, [bin tile][re | im][32-tap chunk][plane][QFRAG]
    int s_nch, s_nbt;       // 32-tap chunks of a table row; 16-bin tiles (the last one zero-padded)
    // cfg.fold == 5: the same product on fp16 x 2 split operands (split_scheme.h: SchemeH2, two planes per fragment).  The prepped samples
    // are bounded by the int16 input, so they are pre-scaled by a power of two that puts their bound at 2^15 (inside the fp16 range, far
    // above its subnormals) and the power is scaled back -- both exact: no range check is needed here.
    int s_np;               // planes per fragment: 3 (fold 4) or 2 (fold 5)
    float s_xs, s_ps;       // sample pre-scale 2^e, power post-scale 2^(-2e) (1, 1 for fold 4)
};

static int round16(int x) { return (x + 15) & ~15; }


// ---------------------------------------------------------------------------------------------------------------------------
// Folded DFT (cfg.fold): only |X|^2 leaves the log-mel kernel, so the spectrum may be rotated by any per-bin phase.  About the
// window's centre c (rotation by 2 pi b c / n_fft) the real table row is even in the tap index and the imaginary row odd -- up
// to what the reference's float32 table arithmetic left (|residual| ~ 6e-5 of the table scale for its three front-ends).  Taps
// are therefore taken in mirror pairs (k, k' = T - k):
//     re' = sum_p E[p] (x_k + x_k')  + sum_k RX[k] x_k        im' = sum_p O[p] (x_k - x_k')  + sum_k IX[k] x_k
// E / O = f32 even / odd parts of the rotated REFERENCE table (half the f32 MFMAs of the dense product), RX / IX = the table
// minus what E / O reproduce, computed in double from the table's own bits and applied as an f16 MFMA product on f16 copies of
// the samples (16x the f32 rate; |RX| 2^-11 |x| 2^-11 relative rounding on a term that is itself 6e-5 of the sum).
// kind 1: c = tap0 + (taps-1)/2 (symmetric windows), kind 2: c = tap0 + taps/2 (periodic windows: tap 0 has no partner and is
// paired with a row of zeros, tap taps/2 is its own partner).
// ---------------------------------------------------------------------------------------------------------------------------
struct FoldPair { int k, kp, kind; };          // kind 0: (k, kp); 1: k == kp; 2: k with the zero row; -1: padding (zero weights)
struct FoldPlan {
    int T, regions, blocks[MAX_REGIONS], offA[MAX_REGIONS], offB[MAX_REGIONS], strB[MAX_REGIONS];
    std::vector<FoldPair> pairs;               // padded: 16 per block, region after region
};

static int fold_plan(const vadx_frontend_cfg *c, int kind, FoldPlan *pl) {
    const int taps = c->taps, hop = c->hop;
    pl->T = kind == 1 ? taps - 1 : taps;
    pl->regions = 0;
    pl->pairs.clear();
    std::vector<FoldPair> seq;
    int k = kind == 1 ? 0 : 1;
    for (; k < pl->T - k; ++k) seq.push_back(FoldPair{k, pl->T - k, 0});
    if (k == pl->T - k && k < taps) seq.push_back(FoldPair{k, k, 1});
    auto close_region = [&](int first, int last, bool zero_partner) -> int {      // seq[first..last]
        if (pl->regions >= MAX_REGIONS) return -1;
        const int len = last - first + 1, nb = (len + 15) / 16, rg = pl->regions++;
        const int k0 = seq[first].k, kp0 = seq[first].kp;
        pl->blocks[rg] = nb;
        pl->offA[rg] = (k0 % hop) * XF_LD + k0 / hop;
        if (k0 % hop + 16 * nb > hop + 16) return -1;                             // padding rows must stay inside X2 (hop + 16 rows)
        if (zero_partner) { pl->offB[rg] = hop * XF_LD; pl->strB[rg] = XF_LD; }
        else {
            pl->offB[rg] = (kp0 % hop) * XF_LD + kp0 / hop; pl->strB[rg] = -XF_LD;
            if (kp0 % hop - (16 * nb - 1) < 0) return -1;
        }
        for (int t = 0; t < 16 * nb; ++t) pl->pairs.push_back(t < len ? seq[first + t] : FoldPair{0, 0, -1});
        return 0;
    };
    int first = 0;
    for (int e = 1; e <= (int)seq.size(); ++e)
        if (e == (int)seq.size() || seq[e].k / hop != seq[first].k / hop || seq[e].kp / hop != seq[first].kp / hop) {
            if (e > first && close_region(first, e - 1, false)) return -1;
            first = e;
        }
    if (kind == 2) {                           // tap 0 alone
        seq.push_back(FoldPair{0, 0, 2});
        if (close_region((int)seq.size() - 1, (int)seq.size() - 1, true)) return -1;
    }
    std::vector<int> seen(taps, 0);            // every tap exactly once
    for (const FoldPair &pr : pl->pairs) {
        if (pr.kind < 0) continue;
        seen[pr.k]++;
        if (pr.kind == 0) seen[pr.kp]++;
    }
    for (int t = 0; t < taps; ++t) if (seen[t] != 1) return -1;
    return 0;
}

// E / O [n_bins][P] (f32) and RX / IX [n_bins][taps] (double) of the reference table for one plan; returns max |RX|,|IX| and max |table|
static void fold_tables(const vadx_frontend_cfg *c, const FoldPlan &pl, const float *cos_tab, const float *sin_tab, int n_fft,
                        std::vector<float> &E, std::vector<float> &O, std::vector<double> &RX, std::vector<double> &IX,
                        double *res_max, double *tab_max) {
    const int P = (int)pl.pairs.size(), taps = c->taps, nb = c->n_bins;
    const double cen = c->tap0 + pl.T / 2.0, two_pi = 6.283185307179586476925286766559;
    std::vector<float> e2((size_t)nb * P), o2((size_t)nb * P);
    std::vector<double> rx2((size_t)nb * taps), ix2((size_t)nb * taps), Rr(taps), Ir(taps);
    double best = -1.0;
    *tab_max = 0.0;
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
        double rmax = 0.0;
        for (int b = 0; b < nb; ++b) {
            const double phi = sgn * two_pi * b * cen / n_fft, cp = cos(phi), sp = sin(phi);
            for (int t = 0; t < taps; ++t) {
                const double R = cos_tab[(size_t)b * n_fft + c->tap0 + t], I = sin_tab[(size_t)b * n_fft + c->tap0 + t];
                Rr[t] = cp * R - sp * I; Ir[t] = sp * R + cp * I;
                if (fabs(R) > *tab_max) *tab_max = fabs(R);
                if (fabs(I) > *tab_max) *tab_max = fabs(I);
            }
            for (int p = 0; p < P; ++p) {
                const FoldPair &pr = pl.pairs[p];
                float e = 0.f, o = 0.f;
                if (pr.kind == 0) {
                    e = (float)(0.5 * (Rr[pr.k] + Rr[pr.kp])); o = (float)(0.5 * (Ir[pr.k] - Ir[pr.kp]));
                    rx2[(size_t)b * taps + pr.k] = Rr[pr.k] - (double)e; rx2[(size_t)b * taps + pr.kp] = Rr[pr.kp] - (double)e;
                    ix2[(size_t)b * taps + pr.k] = Ir[pr.k] - (double)o; ix2[(size_t)b * taps + pr.kp] = Ir[pr.kp] + (double)o;
                } else if (pr.kind == 1) {      // u = 2 x, v = 0
                    e = (float)(0.5 * Rr[pr.k]);
                    rx2[(size_t)b * taps + pr.k] = Rr[pr.k] - 2.0 * (double)e; ix2[(size_t)b * taps + pr.k] = Ir[pr.k];
                } else if (pr.kind == 2) {      // u = v = x
                    e = (float)Rr[pr.k]; o = (float)Ir[pr.k];
                    rx2[(size_t)b * taps + pr.k] = Rr[pr.k] - (double)e; ix2[(size_t)b * taps + pr.k] = Ir[pr.k] - (double)o;
                }
                e2[(size_t)b * P + p] = e; o2[(size_t)b * P + p] = o;
            }
        }
        for (size_t t = 0; t < rx2.size(); ++t) { rmax = fmax(rmax, fabs(rx2[t])); rmax = fmax(rmax, fabs(ix2[t])); }
        if (best < 0.0 || rmax < best) { best = rmax; E = e2; O = o2; RX = rx2; IX = ix2; }
    }
    *res_max = best;
}

// ---------------------------------------------------------------------------------------------------------------------------
// kind 3 = kind 2 composed with the FREQUENCY fold (periodic windows centred on an integer c, n_fft % 64 == 0).  With m = n - c the
// rotated rows of bins b and b' = n_fft/2 - b satisfy  R'[b'][m] = (-1)^m R'[b][m],  I'[b'][m] = -(-1)^m I'[b][m]  (cos / sin of
// pi m - w_b m), and the mirror pairs (m, -m) of kind 2 keep the parity of m.  So per bin b < n_fft/4 the four sums
//     Ce = sum_{m even} E u,  Co = sum_{m odd} E u,  Se = sum_{m even} O v,  So = sum_{m odd} O v
// give BOTH bins:  |X_b|^2 = (Ce + Co)^2 + (Se + So)^2,  |X_b'|^2 = (Ce - Co)^2 + (Se - So)^2  -- a quarter of the dense MACs.
// Bin n_fft/4 is its own mirror (its own K-split tile, as the last bin of kinds 1 / 2).  What the reference's float32 rows hold beyond
// this model (both rows of a mirror pair against the E / O of row b) rides in f16 residual tables that feed the same four
// accumulators: (RX_b + RX_b')/2 -> Ce, (RX_b - RX_b')/2 -> Co, (IX_b +- IX_b')/2 -> Se, So (sign of the b' row chosen by fit).
// Pairs are addressed through a per-slot offset table (two LDS word offsets per pair; a row of zeros for lone taps and padding),
// 16 slots = one block, slot tau of a block <-> MFMA contraction index (q, j) with tau = 2 q + (j & 1) + 8 (j >> 1): the four lane
// quarters of a k-step then read X2 rows four apart (pairs of one parity class are two rows apart), i.e. 16 banks apart.
// OPT-IN (cfg.fold = 3; vadx_frontend_fold_kind never answers 3): Ce and Co each grow to half of the STRONG bin of a mirror pair, so the
// weak one, Ce - Co, inherits float32 round-off relative to its partner -- measured against the double evaluation of the same table,
// bands within 26 dB of the frame's peak are as accurate as the dense product (<= 7e-6 on the log-mel), bands 50 - 70 dB down 2e-4 and
// the few beyond 1e-3 (dense: 4e-5 / 1e-4); FSMN's scores and decisions still meet the 1e-4 / bit-exact bar.  34 % faster than kind 2.
// ---------------------------------------------------------------------------------------------------------------------------
struct Fold3Plan {
    int Pc, Pb;                          // slots per parity class (multiple of 16), blocks in all (2 Pc / 16)
    std::vector<int> ka, kb, sgn;        // per slot: tap of the first member / of its partner (-1: the zero row; ka = -1: padding), class parity
};

static int fold3_slot_tau(int kappa) { const int q = kappa >> 2, j = kappa & 3; return 2 * q + (j & 1) + 8 * (j >> 1); }

static int fold3_plan(const vadx_frontend_cfg *c, int n_fft, Fold3Plan *pl) {
    const int taps = c->taps, h = taps / 2;
    if (taps % 2 || n_fft % 64 || c->n_bins != n_fft / 2 + 1 || c->tap0 + h != n_fft / 2) return -1;      // centre n_fft / 2: rotation = (-1)^b
    std::vector<int> ea, eb, oa, ob;
    for (int m = 1; m < h; ++m) { ((m & 1) ? oa : ea).push_back(h + m); ((m & 1) ? ob : eb).push_back(h - m); }
    ea.push_back(h); eb.push_back(-1);                      // m = 0: alone (u = v = x)
    ea.push_back(0); eb.push_back(-1);                      // m = -h (even: taps % 4 == 0 is implied by n_fft % 64 == 0 and the centre) alone
    if (h & 1) return -1;
    const size_t pc = ((std::max(ea.size(), oa.size()) + 15) / 16) * 16;
    pl->Pc = (int)pc; pl->Pb = (int)(2 * pc / 16);
    pl->ka.assign(2 * pc, -1); pl->kb.assign(2 * pc, -1); pl->sgn.assign(2 * pc, 0);
    for (size_t t = 0; t < ea.size(); ++t) { pl->ka[t] = ea[t]; pl->kb[t] = eb[t]; }
    for (size_t t = 0; t < oa.size(); ++t) { pl->ka[pc + t] = oa[t]; pl->kb[pc + t] = ob[t]; }
    for (size_t t = 0; t < 2 * pc; ++t) pl->sgn[t] = t < pc ? 1 : -1;
    return 0;
}

// kind-3 tables of the reference table: E / O [n_fft/4 + 1 bins][2 Pc slots] (f32), the four residual rows per bin b < n_fft/4 and the two of
// bin n_fft/4 (double, [bin][4][taps]); returns the largest residual and the table scale
static void fold3_tables(const vadx_frontend_cfg *c, const Fold3Plan &pl, const float *cos_tab, const float *sin_tab, int n_fft,
                         std::vector<float> &E, std::vector<float> &O, std::vector<double> &RES, double *res_max, double *tab_max) {
    const int taps = c->taps, h = taps / 2, nq = n_fft / 4, P = 2 * pl.Pc;
    E.assign((size_t)(nq + 1) * P, 0.f); O.assign((size_t)(nq + 1) * P, 0.f); RES.assign((size_t)(nq + 1) * 4 * taps, 0.0);
    std::vector<double> Rb(taps), Ib(taps), Rm(taps), Im(taps), mre(taps), mim(taps);
    *res_max = 0.0; *tab_max = 0.0;
    auto rot = [&](int b, std::vector<double> &R, std::vector<double> &I) {      // rotation about n_fft / 2 = the sign (-1)^b
        const double sg = (b & 1) ? -1.0 : 1.0;
        for (int t = 0; t < taps; ++t) {
            R[t] = sg * cos_tab[(size_t)b * n_fft + c->tap0 + t]; I[t] = sg * sin_tab[(size_t)b * n_fft + c->tap0 + t];
            *tab_max = fmax(*tab_max, fmax(fabs(R[t]), fabs(I[t])));
        }
    };
    std::vector<double> eb(P), ob(P);
    for (int b = 0; b <= nq; ++b) {
        rot(b, Rb, Ib);
        // even / odd parts of row b over the slots (double)
        for (int p = 0; p < P; ++p) {
            const int k = pl.ka[p], kp = pl.kb[p];
            eb[p] = k < 0 ? 0.0 : (kp >= 0 ? 0.5 * (Rb[k] + Rb[kp]) : Rb[k]);
            ob[p] = k < 0 ? 0.0 : (kp >= 0 ? 0.5 * (Ib[k] - Ib[kp]) : Ib[k]);
        }
        double tau = 1.0;
        if (b < nq) {
            // the mirror row carries the same parts up to the parity sign: R'[b'] ~ par E, I'[b'] ~ -par s O, and up to one global sign tau of
            // its imaginary row (the power does not see it: the sign that fits is taken).  The shared model is the AVERAGE of the two rows'
            // parts, so that each row's residual carries half of what the reference's rounding put between them (the residual's f16
            // product is as noisy as the f32 one at 1.2e-4 of the table scale, and negligible at half that).
            rot(n_fft / 2 - b, Rm, Im);
            double fit = 0.0;
            for (int p = 0; p < P; ++p) {
                const int k = pl.ka[p], kp = pl.kb[p];
                if (k < 0) continue;
                const double om = kp >= 0 ? 0.5 * (Im[k] - Im[kp]) : Im[k];
                fit += -pl.sgn[p] * om * ob[p];
            }
            tau = fit >= 0.0 ? 1.0 : -1.0;
            for (int p = 0; p < P; ++p) {
                const int k = pl.ka[p], kp = pl.kb[p];
                if (k < 0) continue;
                const double em = pl.sgn[p] * (kp >= 0 ? 0.5 * (Rm[k] + Rm[kp]) : Rm[k]);
                const double om = -pl.sgn[p] * tau * (kp >= 0 ? 0.5 * (Im[k] - Im[kp]) : Im[k]);
                eb[p] = 0.5 * (eb[p] + em); ob[p] = 0.5 * (ob[p] + om);
            }
        }
        std::fill(mre.begin(), mre.end(), 0.0); std::fill(mim.begin(), mim.end(), 0.0);
        for (int p = 0; p < P; ++p) {
            const int k = pl.ka[p], kp = pl.kb[p];
            if (k < 0) continue;
            const float e = (float)eb[p], o = (float)ob[p];
            if (kp >= 0) { mre[kp] = e; mim[kp] = -(double)o; }
            mre[k] = e; mim[k] = o;
            E[(size_t)b * P + p] = e; O[(size_t)b * P + p] = o;
        }
        double *res = RES.data() + (size_t)b * 4 * taps;
        if (b == nq) {                                      // its own mirror: plain residual rows (re, im)
            for (int t = 0; t < taps; ++t) { res[t] = Rb[t] - mre[t]; res[taps + t] = Ib[t] - mim[t]; }
        } else {
            for (int t = 0; t < taps; ++t) {
                const double par = ((t - h) & 1) ? -1.0 : 1.0;
                const double rb = Rb[t] - mre[t], ib = Ib[t] - mim[t];
                const double rm = Rm[t] - par * mre[t], im = tau * Im[t] + par * mim[t];      // residual of tau * Im against -par * mim
                // accumulators: A = Ce + .., B = Co + .. with A + B = re'_b, A - B = re'_b';  C = Se + .., D = So + .. with C + D = im'_b, C - D = -tau im'_b'
                res[t] = 0.5 * (rb + rm); res[2 * taps + t] = 0.5 * (rb - rm);
                res[taps + t] = 0.5 * (ib - im); res[3 * taps + t] = 0.5 * (ib + im);
            }
        }
        for (int t = 0; t < (b == nq ? 2 : 4) * taps; ++t) *res_max = fmax(*res_max, fabs(res[t]));
    }
}

// LDS of the folded kernel: X2 [hop + 16][XF_LD] f32 | XS [XF_LD][hop + 8] f16, both reused by the power rows [Fp][XF_LD]; then the
// partial sums of the last-bin tile [8 waves][2][64]
static size_t fold_pw_off_bytes(const Dev *d) {
    const size_t stage = (size_t)(d->hop + 16) * XF_LD * 4 + (size_t)XF_LD * (d->hop + 8) * 2, pw = (size_t)d->Fp * XF_LD * 4;
    return ((stage > pw ? stage : pw) + 15) & ~(size_t)15;
}
static size_t fold_lds_bytes(const Dev *d) { return fold_pw_off_bytes(d) + 8 * 2 * 64 * 4; }

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
    if (c->fold == 4 || c->fold == 5) {
        // dense product on split operands (frontend_split_kernel; 4 = bf16 x 3, 5 = fp16 x 2): the flat sample planes need hop = 160 (their
        // skew), the staging knows the int16 preps 0 - 2
        if (c->hop != 160 || c->prep > 2 || c->taps > 512 || c->n_bins > 17 * 16) return -1;
        d->fold = c->fold;
        d->s_np = c->fold == 4 ? 3 : 2;
        d->s_xs = d->s_ps = 1.f;
        if (c->fold == 5) {
            // |prepped sample| <= bound: prep 0 (x - mean) - 0.97 (x[-1] - mean) with |x - mean| < 65536; prep 1 k0 x[-1] + k1 x; prep 2 k1 x - mean
            const double bound = c->prep == 0 ? 65536.0 * 1.97 : (c->prep == 1 ? 32768.0 * (fabs((double)c->k0) + fabs((double)c->k1)) : 65536.0 * fabs((double)c->k1));
            if (!(bound > 0.0) || !(bound < 1e30)) return -1;
            const int e = (int)floor(log2(32768.0 / bound));
            d->s_xs = (float)ldexp(1.0, e);
            d->s_ps = (float)ldexp(1.0, -2 * e);
        }
        d->s_nch = (c->taps + 31) / 32;
        d->s_nbt = (c->n_bins + 15) / 16;
        d->off_fold = (d->off_mel + d->n_mels * d->Fp + 255) / 256 * 256;      // fragments on 1 KiB boundaries: a wave's 1 KiB load touches 16 lines, not 17
        d->off_plan = d->off_fold + d->s_nbt * 2 * d->s_nch * d->s_np * vadx::QFRAG;
        d->tiles64 = c->frames / TF_FOLD;
        const int rem64 = c->frames - d->tiles64 * TF_FOLD;
        d->tail_mt = (rem64 + 15) / 16;
        if (d->tail_mt == 4) { d->tiles64 += 1; d->tail_mt = 0; }
        return 0;
    }
    if (c->fold) {
        if (c->fold < 1 || c->fold > 3) return -1;
        if (!(c->prep <= 2 || c->prep >= 6) || d->nbt > 16) return -1;      // int16-derived samples; two power tiles per wave
        if (TF_FOLD + d->passes - 1 >= XF_LD || c->hop < 32) return -1;
        d->fold = c->fold;
        d->f_Kb32 = (c->taps + 31) / 32;
        d->off_fold = d->off_mel + d->n_mels * d->Fp;
        if (c->fold == 3) {
            Fold3Plan p3;
            if (fold3_plan(c, 2 * (c->n_bins - 1), &p3)) return -1;
            d->f_Pb = p3.Pb;
            const int ntl3 = (c->n_bins - 1) / 32 + 1;                                  // tiles of bins 0 .. n_fft/4 - 1, then bin n_fft/4's own
            d->off_res = d->off_fold + ntl3 * 32 * d->f_Pb * 16;
            d->off_plan = d->off_res + ntl3 * 4 * d->f_Kb32 * vadx::FRAG;
        } else {
            FoldPlan pl;
            if (fold_plan(c, c->fold, &pl)) return -1;
            d->f_regions = pl.regions; d->f_Pb = 0;
            for (int r = 0; r < pl.regions; ++r) {
                d->f_blocks[r] = pl.blocks[r]; d->f_offA[r] = pl.offA[r]; d->f_offB[r] = pl.offB[r]; d->f_strB[r] = pl.strB[r];
                d->f_Pb += pl.blocks[r];
            }
            d->off_res = d->off_fold + (d->nbt + d->nyq) * 32 * d->f_Pb * 16;
            d->off_plan = d->off_res + (d->nbt + d->nyq) * 2 * d->f_Kb32 * vadx::FRAG;
        }
        // bound of |sample| after prep: the f16 copies carry x * 2^e with |x + x'| 2^e <= 32768
        float M = 65536.f * 1.97f;                                                          // prep 0
        if (c->prep == 1 || c->prep >= 6) M = 32768.f * (fabsf(c->k0) + fabsf(c->k1));
        else if (c->prep == 2) M = 65536.f * fabsf(c->k1);
        if (!(M > 0.f)) return -1;
        d->f_xscale = exp2f(floorf(log2f(16384.f / M)));
        d->f_rinv = 1.0f / (d->f_xscale * RES_SCALE);
        d->tiles64 = c->frames / TF_FOLD;
        const int rem64 = c->frames - d->tiles64 * TF_FOLD;
        d->tail_mt = (rem64 + 15) / 16;
        if (d->tail_mt == 4) { d->tiles64 += 1; d->tail_mt = 0; }
        if (fold_lds_bytes(d) > 80 * 1024) return -1;
    }
    return 0;
}

static size_t packed_total(const Dev &d) {
    if (d.fold == 4 || d.fold == 5) return (size_t)d.off_plan + 2 * MAX_MEL_TILES + 4;       // + the mel bands
    if (d.fold == 3) return (size_t)d.off_plan + (size_t)d.f_Pb * 32 + 2 * MAX_MEL_TILES;      // [block][quarter][offA x 4 | offB x 4], then the mel bands
    if (d.fold) return (size_t)d.off_plan + 4 * MAX_REGIONS + 2 * MAX_MEL_TILES + 4;      // + last-bin mode (one int, padded to four)
    return (size_t)d.off_mel + (size_t)d.n_mels * d.Fp;
}

// log of the mel energies: v_log_f32 (1 ulp in log2) times ln 2 instead of the library logf (~25 VALU instructions per value; VALU
// time adds to f32-MFMA time on gfx950).  |difference| <= 4e-6 over the feature range, two orders inside the feature tolerance.
#ifndef FE_FAST_LOG
#define FE_FAST_LOG 1
#endif
__device__ __forceinline__ float FE_LOG(float x) { return FE_FAST_LOG ? __builtin_amdgcn_logf(x) * 0.6931471805599453f : logf(x); }

// Phase 0 of both tile bodies: prep + polyphase staging  X2[r][g] = s'[(f0+g)*hop + r]  for g < cols (XLD = row stride of X2).
// HALF: the same samples also go, scaled by the power of two `xscale` and rounded to f16, to XS[g][r] (row pitch xs_pitch
// halves) -- the operand of the folded kernel's residual product.
__device__ __forceinline__ _Float16 to_half_sat(float v) { return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }

template <int XLD, bool HALF>
__device__ __forceinline__ void stage_tile(const Dev &d, const int16_t *__restrict__ win, const float *__restrict__ fwin, float mean,
                                           int f0, int cols, float *X2, _Float16 *XS, int xs_pitch, float xscale) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
                if (r < d.hop) { const float sv = in ? v : 0.f; X2[r * XLD + g] = sv; if (HALF) XS[g * xs_pitch + r] = to_half_sat(__fmul_rn(sv, xscale)); }
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
                if (r < d.hop) { const float sv = in ? v : 0.f; X2[r * XLD + g] = sv; if (HALF) XS[g * xs_pitch + r] = to_half_sat(__fmul_rn(sv, xscale)); }
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
}

// Phase 2 of both tile bodies: banded mel GEMM over the power rows PW[bin][frame] (row stride p_ld) + log, D rows = mel (SWAP) so
// each lane stores 4 consecutive mels
// (bands: the folded kernel reads [mel tile][lo, hi] from the blob -- with its four tile variants inlined, the dynamic index into the
// by-value Dev arrays made the compiler keep a 384-byte copy of Dev in scratch)
template <int MT, bool BANDS = false, int NWV = THREADS / 64>
__device__ __forceinline__ void mel_phase(const Dev &d, const float *__restrict__ P, const float *PW, int p_ld, int f0,
                                          float *__restrict__ out_win, const int *__restrict__ bands = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, i = lane & 15;
    auto finish = [&](int mtile, int mt, const f32x4 &a) {
        const int f = f0 + mt * 16 + i;
        if (f < d.frames) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = FE_LOG(d.log_mode ? __fadd_rn(a[r], d.log_floor) : fmaxf(a[r], d.log_floor));
            *reinterpret_cast<f32x4 *>(out_win + (size_t)f * d.out_stride + d.out_off + mtile * 16 + 4 * q) = v;
        }
    };
    if (NWV < 8) {
        // few waves (the split kernel's four): (mel tile, column tile) items -- five mel tiles would be two rounds with one busy wave
        for (int item = wave; item < d.nmt * MT; item += NWV) {
            const int mtile = item / MT, mt1 = item - mtile * MT;
            f32x4 acc[1][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}};
            const int lo = BANDS ? bands[2 * mtile] : d.mel_kb_lo[mtile], hi = BANDS ? bands[2 * mtile + 1] : d.mel_kb_hi[mtile];
            const float *const wrow[1] = {vadx::frag_ptr(P + d.off_mel, d.Fp, mtile, lo * 16, lane)};
            const int moff[1] = {mt1 * 16};
            if (hi > lo) vadx::gemm_rt_simple<1, 1, true>(acc, PW + lo * 16 * p_ld, p_ld, moff, wrow, hi - lo, lane);
            finish(mtile, mt1, acc[0][0]);
        }
        return;
    }
    for (int mtile = wave; mtile < d.nmt; mtile += NWV) {
        f32x4 acc[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int lo = BANDS ? bands[2 * mtile] : d.mel_kb_lo[mtile], hi = BANDS ? bands[2 * mtile + 1] : d.mel_kb_hi[mtile];
        const float *const wrow[1] = {vadx::frag_ptr(P + d.off_mel, d.Fp, mtile, lo * 16, lane)};
        int moff[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) moff[mt] = mt * 16;
        if (hi > lo) vadx::gemm_rt_simple<1, MT, true>(acc, PW + lo * 16 * p_ld, p_ld, moff, wrow, hi - lo, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) finish(mtile, mt, acc[0][mt]);
    }
}

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
    // ---- phase 0: prep + polyphase staging  X2[r][g] = s'[(f0+g)*hop + r]  (stage_tile)
    stage_tile<X_LD, false>(d, win, fwin, mean, f0, cols, X2, nullptr, 0, 0.f);
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

    mel_phase<MT>(d, P, PW, P_LD, f0, out_win);
    FE_ACC(3);
}


typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) f16x8 *global_f16x8_ptr;
__device__ __forceinline__ f32x4 mfma16h(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// One tile of MT*16 frames of the folded kernel (see "Folded DFT" above).  Phase 1 keeps the power of the wave's (at most two)
// bin tiles in registers: the power rows reuse the LDS of X2 / XS once every wave is done with them.
template <int MT>
__device__ __forceinline__ void fold_tile(const Dev &d, const float *__restrict__ P, const int16_t *__restrict__ win, float mean,
                                          int f0, float *__restrict__ out_win, float *lds, float *NQ) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, i = lane & 15;
    constexpr int NF = MT * 16;
    const int cols = NF + d.passes - 1, xs_pitch = d.hop + 8;
    float *X2 = lds, *PW = lds;
    _Float16 *XS = reinterpret_cast<_Float16 *>(lds + (d.hop + 16) * XF_LD);

    FE_T0();
    for (int e = tid; e < 16 * XF_LD; e += THREADS) X2[d.hop * XF_LD + e] = 0.f;                      // the zero rows (lone taps, padding pairs)
    for (int e = tid; e < xs_pitch; e += THREADS) XS[cols * xs_pitch + e] = (_Float16)0.f;            // column read by the residual's padded taps
    stage_tile<XF_LD, true>(d, win, nullptr, mean, f0, cols, X2, XS, xs_pitch, d.f_xscale);
    FE_ACC(4);
    __syncthreads();
    FE_ACC(0);

    // 16 pairs (4 k-steps) of the symmetric part: u -> real accumulators, v -> imaginary ones
    auto fold_block = [&](const f32x4 &e4, const f32x4 &o4, const float *pa, const float *pb, int strB, f32x4 (&are)[MT], f32x4 (&aim)[MT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float u[MT], v[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float xa = pa[j * XF_LD + mt * 16], xb = pb[j * strB + mt * 16];
                u[mt] = (FE_WHATIF & 4) ? xa : __fadd_rn(xa, xb);
                v[mt] = (FE_WHATIF & 4) ? xb : __fsub_rn(xa, xb);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                are[mt] = vadx::mfma16(u[mt], e4[j], are[mt]);
                aim[mt] = vadx::mfma16(v[mt], o4[j], aim[mt]);
            }
        }
    };
    const int Kp = d.f_Pb * 16;
    const int *__restrict__ plan = reinterpret_cast<const int *>(P + d.off_plan);
    const int last_mode = plan[4 * MAX_REGIONS + 2 * MAX_MEL_TILES];      // 0: both products of the last bin's tile, 1: v x O only, 2: u x E only
    if (d.nyq) {
        // last bin (n_bins % 16 == 1): its 16-pair blocks and 32-tap residual blocks are dealt round-robin to the eight waves; the B tile of
        // the u product carries E in row 0, the one of the v product O in row 1, so one accumulator holds (re', im') in columns 0, 1
        f32x4 na[MT], nr[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { na[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; nr[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const f16x8 *rf = reinterpret_cast<const f16x8 *>(P + d.off_res + (size_t)(d.nbt * 2) * d.f_Kb32 * vadx::FRAG) + lane;
        for (int S = wave; S < d.f_Kb32; S += THREADS / 64) {
            const f16x8 w = *(global_f16x8_ptr)(rf + S * 64);
            const int k = 32 * S + 8 * q, a = k / d.hop, r = k - a * d.hop;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                nr[mt] = mfma16h(*reinterpret_cast<const f16x8 *>(XS + (mt * 16 + i + a) * xs_pitch + r), w, nr[mt]);
        }
        const float *fe = vadx::frag_ptr(P + d.off_fold, Kp, d.nbt * 2, 0, lane), *fo = vadx::frag_ptr(P + d.off_fold, Kp, d.nbt * 2 + 1, 0, lane);
        int gb = 0;
        for (int rg = 0; rg < d.f_regions; ++rg) {
            const int nblk = plan[4 * rg], strB = plan[4 * rg + 3];
            const float *pa = X2 + plan[4 * rg + 1] + 4 * q * XF_LD + i, *pb = X2 + plan[4 * rg + 2] + 4 * q * strB + i;
            for (int S = 0; S < nblk; ++S, ++gb) {
                if ((gb & 7) != wave) continue;
                const float *pas = pa + 16 * S * XF_LD, *pbs = pb + 16 * S * strB;
                if (last_mode == 0) fold_block(vadx::ldg4(fe + vadx::FRAG * gb), vadx::ldg4(fo + vadx::FRAG * gb), pas, pbs, strB, na, na);
                else {                           // one of the two parts is null (packed into the residual): half the MFMAs
                    const f32x4 w4 = vadx::ldg4((last_mode == 2 ? fe : fo) + vadx::FRAG * gb);
                    const float sg = last_mode == 2 ? 1.f : -1.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            na[mt] = vadx::mfma16(__fadd_rn(pas[j * XF_LD + mt * 16], __fmul_rn(sg, pbs[j * strB + mt * 16])), w4[j], na[mt]);
                }
            }
        }
        if (i < 2)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 t;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = __fmaf_rn(nr[mt][r], d.f_rinv, na[mt][r]);
                *reinterpret_cast<f32x4 *>(NQ + (wave * 2 + i) * 64 + mt * 16 + 4 * q) = t;
            }
    }
    // The wave's two row tiles (bt0 = wave, bt1 = wave + 8) go through ONE loop: the LDS operands are read and u / v formed once per
    // k-step for both (four accumulator sets), each tile's own accumulation order unchanged (MarbleNet 11.56 -> 11.05 ms, FSMN 23.15 -> 21.9;
    // FireRed's 13 tiles 8.41 -> 8.61).
    // A wave without a second tile (13 tiles on eight waves) skips the second tile's MFMAs (wave-uniform branch).
    f32x4 pw[2][MT];
    {
        const bool two = wave + THREADS / 64 < d.nbt;             // (wave-uniform)
        const int bt0 = wave < d.nbt ? wave : d.nbt - 1, bt1 = two ? wave + THREADS / 64 : bt0;
        f32x4 acc[4][MT];                                         // (re, im) of bt0, (re, im) of bt1
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // residual (f16): RX -> real, IX -> imaginary, both against the f16 samples of the frame; table fragments one block ahead
        if (!(FE_WHATIF & 1)) {
            const f16x8 *r0 = reinterpret_cast<const f16x8 *>(P + d.off_res + (size_t)(bt0 * 2) * d.f_Kb32 * vadx::FRAG) + lane;
            const f16x8 *r1 = reinterpret_cast<const f16x8 *>(P + d.off_res + (size_t)(bt1 * 2) * d.f_Kb32 * vadx::FRAG) + lane;
            const int rs = d.f_Kb32 * 64;
            f16x8 w0 = *(global_f16x8_ptr)(r0), w1 = *(global_f16x8_ptr)(r0 + rs), w2 = *(global_f16x8_ptr)(r1), w3 = *(global_f16x8_ptr)(r1 + rs);
            int a = 0, r = 8 * q;
            while (r >= d.hop) { r -= d.hop; ++a; }
            for (int S = 0; S < d.f_Kb32; ++S) {
                const int Sn = ((FE_WHATIF & 2) ? 0 : (S + 1 < d.f_Kb32 ? S + 1 : S)) * 64;
                const f16x8 n0 = *(global_f16x8_ptr)(r0 + Sn), n1 = *(global_f16x8_ptr)(r0 + rs + Sn);
                const f16x8 n2 = *(global_f16x8_ptr)(r1 + Sn), n3 = *(global_f16x8_ptr)(r1 + rs + Sn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f16x8 xb = *reinterpret_cast<const f16x8 *>(XS + (mt * 16 + i + a) * xs_pitch + r);
                    acc[0][mt] = mfma16h(xb, w0, acc[0][mt]);
                    acc[1][mt] = mfma16h(xb, w1, acc[1][mt]);
                    if (two) {
                        acc[2][mt] = mfma16h(xb, w2, acc[2][mt]);
                        acc[3][mt] = mfma16h(xb, w3, acc[3][mt]);
                    }
                }
                w0 = n0; w1 = n1; w2 = n2; w3 = n3;
                r += 32;
                while (r >= d.hop) { r -= d.hop; ++a; }
            }
#pragma unroll
            for (int a2 = 0; a2 < 4; ++a2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[a2][mt] *= d.f_rinv;
        }
        // symmetric part (f32): table fragments one block ahead (requesting the LDS operands a k-step ahead as well, on two register
        // sets, measured no gain: four waves per SIMD already cover that latency)
        if (!(FE_WHATIF & 8)) {
            const float *fe0 = vadx::frag_ptr(P + d.off_fold, Kp, bt0 * 2, 0, lane), *fo0 = vadx::frag_ptr(P + d.off_fold, Kp, bt0 * 2 + 1, 0, lane);
            const float *fe1 = vadx::frag_ptr(P + d.off_fold, Kp, bt1 * 2, 0, lane), *fo1 = vadx::frag_ptr(P + d.off_fold, Kp, bt1 * 2 + 1, 0, lane);
            f32x4 e0 = vadx::ldg4(fe0), o0 = vadx::ldg4(fo0), e1 = vadx::ldg4(fe1), o1 = vadx::ldg4(fo1);
            int gb = 0;
            for (int rg = 0; rg < d.f_regions; ++rg) {
                const int nblk = plan[4 * rg], strB = plan[4 * rg + 3];
                const float *pa = X2 + plan[4 * rg + 1] + 4 * q * XF_LD + i, *pb = X2 + plan[4 * rg + 2] + 4 * q * strB + i;
                for (int S = 0; S < nblk; ++S, ++gb) {
                    const int gn = vadx::FRAG * ((FE_WHATIF & 2) ? 0 : (gb + 1 < d.f_Pb ? gb + 1 : gb));
                    const f32x4 en0 = vadx::ldg4(fe0 + gn), on0 = vadx::ldg4(fo0 + gn), en1 = vadx::ldg4(fe1 + gn), on1 = vadx::ldg4(fo1 + gn);
                    __builtin_amdgcn_sched_barrier(0);      // the next block's table fragments are requested before this block's MFMAs issue
                    const float *pas = pa + 16 * S * XF_LD, *pbs = pb + 16 * S * strB;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float u[MT], v[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const float xa = pas[j * XF_LD + mt * 16], xb = pbs[j * strB + mt * 16];
                            u[mt] = (FE_WHATIF & 4) ? xa : __fadd_rn(xa, xb);
                            v[mt] = (FE_WHATIF & 4) ? xb : __fsub_rn(xa, xb);
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            acc[0][mt] = vadx::mfma16(u[mt], e0[j], acc[0][mt]);
                            acc[1][mt] = vadx::mfma16(v[mt], o0[j], acc[1][mt]);
                        }
                        if (two)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                acc[2][mt] = vadx::mfma16(u[mt], e1[j], acc[2][mt]);
                                acc[3][mt] = vadx::mfma16(v[mt], o1[j], acc[3][mt]);
                            }
                    }
                    e0 = en0; o0 = on0; e1 = en1; o1 = on1;
                }
            }
        }
#pragma unroll
        for (int slot = 0; slot < 2; ++slot)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pw[slot][mt][r] = __fadd_rn(__fmul_rn(acc[2 * slot][mt][r], acc[2 * slot][mt][r]), __fmul_rn(acc[2 * slot + 1][mt][r], acc[2 * slot + 1][mt][r]));
    }
    FE_ACC(1);
    __syncthreads();                     // every wave is done with X2 / XS: the power rows take their place
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int bt = wave + slot * (THREADS / 64);
        if (bt < d.nbt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4 *>(&PW[(bt * 16 + i) * XF_LD + mt * 16 + 4 * q]) = pw[slot][mt];
    }
    if (d.nyq) {
        if (tid < NF) {
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < THREADS / 64; ++w8) { re += NQ[(w8 * 2) * 64 + tid]; im += NQ[(w8 * 2 + 1) * 64 + tid]; }
            PW[(d.n_bins - 1) * XF_LD + tid] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
        }
    }
    for (int e = tid; e < (d.Fp - d.n_bins) * NF; e += THREADS) {      // padded power rows: 0 x 0 in the mel GEMM
        const int r = e / NF, c2 = e - r * NF;
        PW[(d.n_bins + r) * XF_LD + c2] = 0.f;
    }
    __syncthreads();
    FE_ACC(2);
    mel_phase<MT, true>(d, P, PW, XF_LD, f0, out_win, plan + 4 * MAX_REGIONS);
    FE_ACC(3);
}

__global__ __launch_bounds__(THREADS, 4) void frontend_fold_kernel(
    Dev d, const float *__restrict__ P, const int16_t *__restrict__ audio, long long row_stride,
    long long win_stride, int windows_per_clip, const float *__restrict__ means, float *__restrict__ out, int nq_off) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tiles = d.tiles64 + (d.tail_mt ? 1 : 0);
    const int widx = blockIdx.x / tiles, tile = blockIdx.x - widx * tiles;
    const int b = widx / windows_per_clip, w = widx - b * windows_per_clip;
    const int16_t *win = audio + (long long)b * row_stride + (long long)w * win_stride;
    float *out_win = out + (size_t)widx * d.frames * d.out_stride;
    const float mean = means ? means[widx] : 0.f;
    float *NQ = lds + nq_off;
    if (tile < d.tiles64) fold_tile<4>(d, P, win, mean, tile * TF_FOLD, out_win, lds, NQ);
    else if (d.tail_mt == 3) fold_tile<3>(d, P, win, mean, d.tiles64 * TF_FOLD, out_win, lds, NQ);
    else if (d.tail_mt == 2) fold_tile<2>(d, P, win, mean, d.tiles64 * TF_FOLD, out_win, lds, NQ);
    else fold_tile<1>(d, P, win, mean, d.tiles64 * TF_FOLD, out_win, lds, NQ);
}

// One tile of MT*16 frames of the kind-3 kernel (time x frequency fold, see "kind 3" above): wave w owns bins 16 w .. 16 w + 15 AND
// their mirrors n_fft/2 - b through four accumulator sets (Ce, Se, Co, So); bin n_fft/4 is K-split over the waves like the last bin
// of kinds 1 / 2.  Eight row tiles on eight waves: no second round.
template <int MT>
__device__ __forceinline__ void fold3_tile(const Dev &d, const float *__restrict__ P, const int16_t *__restrict__ win, float mean,
                                           int f0, float *__restrict__ out_win, float *lds, float *NQ) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, i = lane & 15;
    constexpr int NF = MT * 16;
    const int cols = NF + d.passes - 1, xs_pitch = d.hop + 8;
    float *X2 = lds, *PW = lds;
    _Float16 *XS = reinterpret_cast<_Float16 *>(lds + (d.hop + 16) * XF_LD);
    const int nq = (d.n_bins - 1) >> 1, nt3 = nq >> 4;            // bin n_fft/4; row tiles below it (8 for 257 bins)
    const int Kp = d.f_Pb * 16, pbc = d.f_Pb >> 1;                // pair slots; blocks per parity class
    const int *__restrict__ plan = reinterpret_cast<const int *>(P + d.off_plan);

    FE_T0();
    for (int e = tid; e < 16 * XF_LD; e += THREADS) X2[d.hop * XF_LD + e] = 0.f;                      // the zero rows (lone taps, padding pairs)
    for (int e = tid; e < xs_pitch; e += THREADS) XS[cols * xs_pitch + e] = (_Float16)0.f;            // column read by the residual's padded taps
    stage_tile<XF_LD, true>(d, win, nullptr, mean, f0, cols, X2, XS, xs_pitch, d.f_xscale);
    FE_ACC(4);
    __syncthreads();
    FE_ACC(0);

    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) i32x4 *global_i32x4_ptr;
    // 16 pair slots (4 k-steps) of one parity class: u -> `are`, v -> `aim`; the slots' X2 offsets come from the blob
    auto block3 = [&](const f32x4 &e4, const f32x4 &o4, const i32x4 &oa, const i32x4 &ob, f32x4 (&are)[MT], f32x4 (&aim)[MT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *pa = X2 + oa[j] + i, *pb = X2 + ob[j] + i;
            float u[MT], v[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float xa = pa[mt * 16], xb = pb[mt * 16];
                u[mt] = __fadd_rn(xa, xb); v[mt] = __fsub_rn(xa, xb);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                are[mt] = vadx::mfma16(u[mt], e4[j], are[mt]);
                aim[mt] = vadx::mfma16(v[mt], o4[j], aim[mt]);
            }
        }
    };
    const int *po = plan + q * 8;                                  // this lane quarter's offsets of block S: po + 32 S
    {   // bin n_fft/4 (its own mirror): blocks dealt round-robin to the waves; u tile row 0 = E, v tile row 1 = O -> one accumulator holds (re', im')
        f32x4 na[MT], nr[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { na[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; nr[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const f16x8 *rf = reinterpret_cast<const f16x8 *>(P + d.off_res + (size_t)(nt3 * 4) * d.f_Kb32 * vadx::FRAG) + lane;
        for (int S = wave; S < d.f_Kb32; S += THREADS / 64) {
            const f16x8 w = *(global_f16x8_ptr)(rf + S * 64);
            const int k = 32 * S + 8 * q, a = k / d.hop, r = k - a * d.hop;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                nr[mt] = mfma16h(*reinterpret_cast<const f16x8 *>(XS + (mt * 16 + i + a) * xs_pitch + r), w, nr[mt]);
        }
        const float *fe = vadx::frag_ptr(P + d.off_fold, Kp, nt3 * 2, 0, lane), *fo = vadx::frag_ptr(P + d.off_fold, Kp, nt3 * 2 + 1, 0, lane);
        for (int S = wave; S < d.f_Pb; S += THREADS / 64) {
            const i32x4 oa = *(global_i32x4_ptr)(po + 32 * S), ob = *(global_i32x4_ptr)(po + 32 * S + 4);
            block3(vadx::ldg4(fe + vadx::FRAG * S), vadx::ldg4(fo + vadx::FRAG * S), oa, ob, na, na);
        }
        if (i < 2)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 t;
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = __fmaf_rn(nr[mt][r], d.f_rinv, na[mt][r]);
                *reinterpret_cast<f32x4 *>(NQ + (wave * 2 + i) * 64 + mt * 16 + 4 * q) = t;
            }
    }
    f32x4 pw[2][MT];                                              // power of bins b (slot 0) and n_fft/2 - b (slot 1)
    if (wave < nt3) {
        const int bt = wave;
        f32x4 acc[4][MT];                                         // Ce, Se, Co, So
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        {   // residual (f16): four tables against the f16 samples of the frame, one per accumulator set
            const f16x8 *rb = reinterpret_cast<const f16x8 *>(P + d.off_res + (size_t)(bt * 4) * d.f_Kb32 * vadx::FRAG) + lane;
            const int rs = d.f_Kb32 * 64;
            f16x8 w0 = *(global_f16x8_ptr)(rb), w1 = *(global_f16x8_ptr)(rb + rs), w2 = *(global_f16x8_ptr)(rb + 2 * rs), w3 = *(global_f16x8_ptr)(rb + 3 * rs);
            int a = 0, r = 8 * q;
            while (r >= d.hop) { r -= d.hop; ++a; }
            for (int S = 0; S < d.f_Kb32; ++S) {
                const int Sn = S + 1 < d.f_Kb32 ? S + 1 : S;
                const f16x8 n0 = *(global_f16x8_ptr)(rb + Sn * 64), n1 = *(global_f16x8_ptr)(rb + rs + Sn * 64);
                const f16x8 n2 = *(global_f16x8_ptr)(rb + 2 * rs + Sn * 64), n3 = *(global_f16x8_ptr)(rb + 3 * rs + Sn * 64);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f16x8 xb = *reinterpret_cast<const f16x8 *>(XS + (mt * 16 + i + a) * xs_pitch + r);
                    acc[0][mt] = mfma16h(xb, w0, acc[0][mt]);
                    acc[1][mt] = mfma16h(xb, w1, acc[1][mt]);
                    acc[2][mt] = mfma16h(xb, w2, acc[2][mt]);
                    acc[3][mt] = mfma16h(xb, w3, acc[3][mt]);
                }
                w0 = n0; w1 = n1; w2 = n2; w3 = n3;
                r += 32;
                while (r >= d.hop) { r -= d.hop; ++a; }
            }
#pragma unroll
            for (int a2 = 0; a2 < 4; ++a2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[a2][mt] *= d.f_rinv;
        }
        // symmetric part (f32): table fragments and slot offsets one block ahead; even-class blocks feed (Ce, Se), odd-class blocks (Co, So)
        const float *fe = vadx::frag_ptr(P + d.off_fold, Kp, bt * 2, 0, lane), *fo = vadx::frag_ptr(P + d.off_fold, Kp, bt * 2 + 1, 0, lane);
        f32x4 ec = vadx::ldg4(fe), oc = vadx::ldg4(fo);
        i32x4 ac = *(global_i32x4_ptr)(po), bc = *(global_i32x4_ptr)(po + 4);
        for (int S = 0; S < pbc; ++S) {
            const f32x4 en = vadx::ldg4(fe + vadx::FRAG * (S + 1)), on = vadx::ldg4(fo + vadx::FRAG * (S + 1));      // (block pbc exists: the odd class)
            const i32x4 an = *(global_i32x4_ptr)(po + 32 * (S + 1)), bn = *(global_i32x4_ptr)(po + 32 * (S + 1) + 4);
            __builtin_amdgcn_sched_barrier(0);
            block3(ec, oc, ac, bc, acc[0], acc[1]);
            ec = en; oc = on; ac = an; bc = bn;
        }
        for (int S = pbc; S < d.f_Pb; ++S) {
            const int Sn = S + 1 < d.f_Pb ? S + 1 : S;
            const f32x4 en = vadx::ldg4(fe + vadx::FRAG * Sn), on = vadx::ldg4(fo + vadx::FRAG * Sn);
            const i32x4 an = *(global_i32x4_ptr)(po + 32 * Sn), bn = *(global_i32x4_ptr)(po + 32 * Sn + 4);
            __builtin_amdgcn_sched_barrier(0);
            block3(ec, oc, ac, bc, acc[2], acc[3]);
            ec = en; oc = on; ac = an; bc = bn;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float re = __fadd_rn(acc[0][mt][r], acc[2][mt][r]), im = __fadd_rn(acc[1][mt][r], acc[3][mt][r]);
                const float rm = __fsub_rn(acc[0][mt][r], acc[2][mt][r]), imm = __fsub_rn(acc[1][mt][r], acc[3][mt][r]);
                pw[0][mt][r] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
                pw[1][mt][r] = __fadd_rn(__fmul_rn(rm, rm), __fmul_rn(imm, imm));
            }
    }
    FE_ACC(1);
    __syncthreads();                     // every wave is done with X2 / XS: the power rows take their place
    if (wave < nt3) {
        const int b = wave * 16 + i;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            *reinterpret_cast<f32x4 *>(&PW[b * XF_LD + mt * 16 + 4 * q]) = pw[0][mt];
            *reinterpret_cast<f32x4 *>(&PW[(2 * nq - b) * XF_LD + mt * 16 + 4 * q]) = pw[1][mt];
        }
    }
    if (tid < NF) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < THREADS / 64; ++w8) { re += NQ[(w8 * 2) * 64 + tid]; im += NQ[(w8 * 2 + 1) * 64 + tid]; }
        PW[nq * XF_LD + tid] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
    }
    for (int e = tid; e < (d.Fp - d.n_bins) * NF; e += THREADS) {      // padded power rows: 0 x 0 in the mel GEMM
        const int r = e / NF, c2 = e - r * NF;
        PW[(d.n_bins + r) * XF_LD + c2] = 0.f;
    }
    __syncthreads();
    FE_ACC(2);
    mel_phase<MT, true>(d, P, PW, XF_LD, f0, out_win, plan + d.f_Pb * 32);
    FE_ACC(3);
}

__global__ __launch_bounds__(THREADS, 4) void frontend_fold3_kernel(
    Dev d, const float *__restrict__ P, const int16_t *__restrict__ audio, long long row_stride,
    long long win_stride, int windows_per_clip, const float *__restrict__ means, float *__restrict__ out, int nq_off) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tiles = d.tiles64 + (d.tail_mt ? 1 : 0);
    const int widx = blockIdx.x / tiles, tile = blockIdx.x - widx * tiles;
    const int b = widx / windows_per_clip, w = widx - b * windows_per_clip;
    const int16_t *win = audio + (long long)b * row_stride + (long long)w * win_stride;
    float *out_win = out + (size_t)widx * d.frames * d.out_stride;
    const float mean = means ? means[widx] : 0.f;
    float *NQ = lds + nq_off;
    if (tile < d.tiles64) fold3_tile<4>(d, P, win, mean, tile * TF_FOLD, out_win, lds, NQ);
    else if (d.tail_mt == 3) fold3_tile<3>(d, P, win, mean, d.tiles64 * TF_FOLD, out_win, lds, NQ);
    else if (d.tail_mt == 2) fold3_tile<2>(d, P, win, mean, d.tiles64 * TF_FOLD, out_win, lds, NQ);
    else fold3_tile<1>(d, P, win, mean, d.tiles64 * TF_FOLD, out_win, lds, NQ);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Dense DFT product on bf16 x 3 split operands (cfg.fold == 4; csrc/split3.h, layers_split.h).  The reference's own float32 table, split
// exactly into three bf16 planes on the host, times the prepped samples, split exactly by the lanes that stage them: six
// v_mfma_f32_16x16x32_bf16 per 32 taps, i.e. 6/16 of the dense f32 product's matrix time and 3/4 of the folded f32 product's -- with no
// fold plan, no residual table and no symmetry assumption about the table (any window, any centre).
//   * a tile of 64 frames needs 63 hop + taps CONSECUTIVE prepped samples; they sit in LDS once, as three flat bf16 planes.  The eight
//     taps 8 G .. 8 G + 7 of frame c are the 16-byte block 20 c + G of a plane (hop = 160 = 20 blocks): the MFMA's B fragment is read
//     straight out of the sample stream, frames overlap for free.  Block b lives at slot b + 2 (b / 80) (the skew spreads frames 4 apart
//     over different banks).
//   * wave = bin tile (real and imaginary rows side by side, all four column tiles); the power goes to PW[bin][frame] float32 for the
//     banded mel GEMM + log of the other kernels (mel_phase).
// LDS: planes 3 x 21.5 KB, then the power rows [Fp][68] f32 (74 KB at 257 bins) in their place: two workgroups of four waves per CU.
__device__ __forceinline__ int sq_slot(int b) { return b + 2 * ((b * 3277) >> 18); }      // b / 80 for b < 13 000
constexpr int SQ_BLOCKS = 63 * 20 + 4 * 16;                                               // blocks of a 64-frame tile, taps <= 512
constexpr int SQ_PLANE_BYTES = (SQ_BLOCKS + 2 * (SQ_BLOCKS / 80) + 2 + 7) / 8 * 8 * 16;
constexpr int SQ_THREADS = 256, SQ_WAVES = SQ_THREADS / 64;
#ifndef FE_RING
#define FE_RING 2          /* table fragments requested FE_RING - 1 chunks ahead (layers_split.h: qgemm_group) */
#endif
static size_t split_lds_bytes(const Dev *d) {
    const size_t pw = (size_t)d->Fp * XF_LD * 4;
    return pw > 3 * (size_t)SQ_PLANE_BYTES ? pw : 3 * (size_t)SQ_PLANE_BYTES;
}

// Four waves per workgroup and <= 80 KB of LDS: TWO workgroups per CU, each in a phase of its own -- one stages or runs its mel GEMM
// (VALU, loads, f32 MFMA) while the other's split products own the bf16 pipe.  The power rows take the PLANES' place: a wave keeps the
// power of its bin tiles in registers (<= 4 rounds x MT fragments) until every wave has read its last sample block.
template <typename SC, int MT>
__device__ __forceinline__ void split_tile(const Dev &d, const float *__restrict__ P, const int16_t *__restrict__ win, float mean,
                                           int f0, float *__restrict__ out_win, unsigned char *smem) {
    constexpr int NP = SC::NP;
    float amax = 0.f;             // (the samples are bounded by construction, see Dev::s_xs: nothing reads this)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, i = lane & 15;
    constexpr int NF = MT * 16;
    float *PW = reinterpret_cast<float *>(smem);
    FE_T0();
    // ---- phase 0: prep + split: item = block of eight consecutive samples (all nine loads unconditional, from clamped indices)
    {
        const int n0 = f0 * d.hop + d.tap0 - d.center_pad, nblk = (NF - 1) * 20 + 4 * d.s_nch;
        for (int b0 = tid; b0 < nblk; b0 += SQ_THREADS) {
            float x[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const int n = n0 + 8 * b0 + e - 1;
                x[e] = (float)win[n < 0 ? 0 : (n >= d.window_len ? d.window_len - 1 : n)];
            }
            f32x4 v[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int n = n0 + 8 * b0 + e;
                const float xc = x[e + 1], xm = n > 0 ? x[e] : 0.f;
                float r;
                if (d.prep == 0) {            // FSMN: (x - mean) - 0.97 (x[-1] - mean), first sample kept
                    const float a = __fsub_rn(xc, mean);
                    r = (n > 0) ? __fsub_rn(a, __fmul_rn(0.97f, __fsub_rn(x[e], mean))) : a;
                } else if (d.prep == 1) {     // two-tap conv with zero history
                    r = __fadd_rn(__fmul_rn(xm, d.k0), __fmul_rn(xc, d.k1));
                } else {                      // scale, then remove the window mean
                    r = __fsub_rn(__fmul_rn(xc, d.k1), mean);
                }
                v[e >> 2][e & 3] = (n >= 0 && n < d.window_len) ? (NP == 2 ? __fmul_rn(r, d.s_xs) : r) : 0.f;
            }
            u32x2 pa[NP], pb[NP];
            SC::split4(v[0], pa, amax);
            SC::split4(v[1], pb, amax);
            unsigned char *dp = smem + sq_slot(b0) * 16;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4 *>(dp + p * SQ_PLANE_BYTES) = u32x4{pa[p][0], pa[p][1], pb[p][0], pb[p][1]};
        }
    }
    FE_ACC(4);
    __syncthreads();
    FE_ACC(0);
    // ---- phase 1: DFT as split products, |.|^2 kept in registers
    const float *tab = P + d.off_fold;
    const size_t tstride = (size_t)d.s_nch * NP * vadx::QFRAG;
    // rounds of four bin tiles; one leftover tile (17 = 4 x 4 + 1, 13 = 3 x 4 + 1) goes out as (bin tile, column tile) items instead of
    // a round with one busy wave; two or three leftover tiles are a (partial) round of their own (s_nbt <= 17: at most four rounds)
    const int rem = d.s_nbt & 3, rounds = (d.s_nbt >> 2) + (rem > 1 ? 1 : 0);
    f32x4 pw[4][MT], pl = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rounds; ++r) {
        const int bt = 4 * r + wave;
        if (bt >= d.s_nbt) break;                  // (wave-uniform; no barrier below)
        f32x4 hi[2][MT], lo[2][MT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { hi[a][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[a][mt] = hi[a][mt]; }
        const float *const w[2] = {tab + (size_t)(2 * bt) * tstride, tab + (size_t)(2 * bt + 1) * tstride};
        vadx::qgemm_group<SC, 2, MT, true, FE_RING>(hi, lo, w, 0, d.s_nch, smem, SQ_PLANE_BYTES,
                                       [=](int G, int mt) { return sq_slot(20 * (16 * mt + i) + G) * 16; }, lane);
        f32x4 p[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const f32x4 rev = SC::join(hi[0][mt], lo[0][mt]), imv = SC::join(hi[1][mt], lo[1][mt]);
                const float re = rev[rr], im = imv[rr];
                p[mt][rr] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
                if (NP == 2) p[mt][rr] = __fmul_rn(p[mt][rr], d.s_ps);
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {         // (constant indices: the rows stay in registers)
            if (r == 0) pw[0][mt] = p[mt];
            else if (r == 1) pw[1][mt] = p[mt];
            else if (r == 2) pw[2][mt] = p[mt];
            else pw[3][mt] = p[mt];
        }
    }
    if (rem == 1 && wave < MT) {
        const int bt = d.s_nbt - 1, mt1 = wave;
        f32x4 hi[2][1], lo[2][1];
#pragma unroll
        for (int a = 0; a < 2; ++a) { hi[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[a][0] = hi[a][0]; }
        const float *const w[2] = {tab + (size_t)(2 * bt) * tstride, tab + (size_t)(2 * bt + 1) * tstride};
        vadx::qgemm_group<SC, 2, 1, true>(hi, lo, w, 0, d.s_nch, smem, SQ_PLANE_BYTES,
                                      [=](int G, int) { return sq_slot(20 * (16 * mt1 + i) + G) * 16; }, lane);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const f32x4 rev = SC::join(hi[0][0], lo[0][0]), imv = SC::join(hi[1][0], lo[1][0]);
            const float re = rev[rr], im = imv[rr];
            pl[rr] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
            if (NP == 2) pl[rr] = __fmul_rn(pl[rr], d.s_ps);
        }
    }
    FE_ACC(1);
    __syncthreads();                               // every sample block has been read: the power rows take the planes' place
    FE_ACC(2);
    for (int r = 0; r < rounds; ++r) {
        const int bt = 4 * r + wave;
        if (bt >= d.s_nbt) break;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = r == 0 ? pw[0][mt] : (r == 1 ? pw[1][mt] : (r == 2 ? pw[2][mt] : pw[3][mt]));
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) PW[(bt * 16 + 4 * q + rr) * XF_LD + mt * 16 + i] = v[rr];
        }
    }
    if (rem == 1 && wave < MT)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) PW[((d.s_nbt - 1) * 16 + 4 * q + rr) * XF_LD + wave * 16 + i] = pl[rr];
    __syncthreads();
    mel_phase<MT, true, SQ_WAVES>(d, P, PW, XF_LD, f0, out_win, reinterpret_cast<const int *>(P + d.off_plan));
    FE_ACC(3);
}

template <typename SC>
__global__ __launch_bounds__(SQ_THREADS, 2) void frontend_split_kernel(
    Dev d, const float *__restrict__ P, const int16_t *__restrict__ audio, long long row_stride,
    long long win_stride, int windows_per_clip, const float *__restrict__ means, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsmem[];
    const int tiles = d.tiles64 + (d.tail_mt ? 1 : 0);
    const int widx = blockIdx.x / tiles, tile = blockIdx.x - widx * tiles;
    const int b = widx / windows_per_clip, w = widx - b * windows_per_clip;
    const int16_t *win = audio + (long long)b * row_stride + (long long)w * win_stride;
#if FE_WHATIF & 16      // what-if: every workgroup stages window 0 of clip 0 (L2-warm samples): what a perfect prefetch of the samples could save
    win = audio;
#endif
    float *out_win = out + (size_t)widx * d.frames * d.out_stride;
    const float mean = means ? means[widx] : 0.f;
#if FE_EXP
    const long long fe_c0 = clock64(), fe_w0 = wall_clock64();
#endif
    if (tile < d.tiles64) split_tile<SC, 4>(d, P, win, mean, tile * TF_FOLD, out_win, fsmem);
    else if (d.tail_mt == 3) split_tile<SC, 3>(d, P, win, mean, d.tiles64 * TF_FOLD, out_win, fsmem);
    else if (d.tail_mt == 2) split_tile<SC, 2>(d, P, win, mean, d.tiles64 * TF_FOLD, out_win, fsmem);
    else split_tile<SC, 1>(d, P, win, mean, d.tiles64 * TF_FOLD, out_win, fsmem);
#if FE_EXP
    if (threadIdx.x == 0) {     // shader clock against the constant 100 MHz counter
        atomicAdd(&fe_dbg[5], (unsigned long long)(clock64() - fe_c0));
        atomicAdd(&fe_dbg[6], (unsigned long long)(wall_clock64() - fe_w0));
    }
#endif
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
    return packed_total(d);
}

// Tables on the host in the reference's own layout: cos_tab/sin_tab [n_bins][n_fft] (windowed, the
// float32 values the reference registers as conv kernels), fbank [n_mels][n_bins].
extern "C" int vadx_frontend_pack_host(const vadx_frontend_cfg *cfg, const float *cos_tab, const float *sin_tab,
                                       int n_fft, const float *fbank, float *packed_host, int32_t *mel_kb) {
    Dev d;
    VADX_REQUIRE(cfg && cos_tab && sin_tab && fbank && packed_host && mel_kb, "vadx_frontend_pack_host: NULL argument");
    VADX_REQUIRE(derive(cfg, &d) == 0, "vadx_frontend_pack_host: unsupported geometry (hop %% 16, n_mels %% 16, passes <= 4)");
    VADX_REQUIRE(cfg->tap0 >= 0 && cfg->tap0 + cfg->taps <= n_fft, "vadx_frontend_pack_host: taps outside n_fft");
    const size_t total = packed_total(d);
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
    if (d.fold == 4 || d.fold == 5) {   // the reference table itself as split fragments (three bf16 planes, exact, or two fp16 planes): [bin tile][re | im][chunk][plane][QFRAG]
        float wmax = 0.f;
        for (int bt = 0; bt < d.s_nbt; ++bt)
            for (int part = 0; part < 2; ++part)
                for (int kc = 0; kc < d.s_nch; ++kc) {
                    float *f3 = packed_host + d.off_fold + (size_t)(((bt * 2 + part) * d.s_nch + kc) * d.s_np) * vadx::QFRAG;
                    for (int i = 0; i < 16; ++i)
                        for (int k = 0; k < 32; ++k) {
                            const int f = bt * 16 + i, t = 32 * kc + k;
                            const float v = (f < d.n_bins && t < d.taps) ? (part ? sin_tab : cos_tab)[(size_t)f * n_fft + cfg->tap0 + t] : 0.f;
                            if (d.s_np == 3) vadx::SchemeB3::put_host(f3, i, k, v, wmax);
                            else vadx::SchemeH2::put_host(f3, i, k, v, wmax);
                        }
                }
        VADX_REQUIRE(d.fold != 5 || wmax <= vadx::H_MAX, "vadx_frontend_pack_host: a table entry (|.| up to %g) is outside the fp16 range: use fold = 4", wmax);
        int32_t *pi = reinterpret_cast<int32_t *>(packed_host + d.off_plan);
        for (int mt = 0; mt < d.nmt; ++mt) { pi[2 * mt] = mel_kb[2 * mt]; pi[2 * mt + 1] = mel_kb[2 * mt + 1]; }
        return VADX_OK;
    }
    if (d.fold == 3) {
        Fold3Plan p3;
        VADX_REQUIRE(fold3_plan(cfg, n_fft, &p3) == 0, "vadx_frontend_pack_host: no kind-3 fold plan for this geometry");
        std::vector<float> E, O;
        std::vector<double> RES;
        double res_max = 0.0, tab_max = 0.0;
        fold3_tables(cfg, p3, cos_tab, sin_tab, n_fft, E, O, RES, &res_max, &tab_max);
        VADX_REQUIRE(res_max <= FOLD_MAX_RATIO * tab_max && res_max * RES_SCALE < 30000.0,
                     "vadx_frontend_pack_host: table is too far from the kind-3 model (residual %.3g of %.3g): ask vadx_frontend_fold_kind", res_max, tab_max);
        const int P = 2 * p3.Pc, nq = n_fft / 4, nt3 = nq / 16, ntl3 = nt3 + 1;
        // columns in MFMA contraction order: column 16 S + kappa of a row = slot 16 S + tau(kappa)
        auto put_cols = [&](float *dst, const float *src) { for (int cidx = 0; cidx < P; ++cidx) dst[cidx] = src[(cidx & ~15) + fold3_slot_tau(cidx & 15)]; };
        float *fm = packed_host + d.off_fold;                      // [ntl3][E rows 0..15 | O rows 16..31][P]
        for (int t = 0; t < nt3; ++t)
            for (int i = 0; i < 16; ++i) {
                put_cols(fm + (size_t)(t * 32 + i) * P, E.data() + (size_t)(t * 16 + i) * P);
                put_cols(fm + (size_t)(t * 32 + 16 + i) * P, O.data() + (size_t)(t * 16 + i) * P);
            }
        put_cols(fm + (size_t)(nt3 * 32) * P, E.data() + (size_t)nq * P);                 // bin n_fft/4: u tile row 0 = E, v tile row 1 = O
        put_cols(fm + (size_t)(nt3 * 32 + 17) * P, O.data() + (size_t)nq * P);
        vadx::frag_major_inplace(fm, ntl3 * 32, P);
        _Float16 *rh = reinterpret_cast<_Float16 *>(packed_host + d.off_res);           // [ntl3][4 accumulators][Kb32][64 lanes][8]
        for (int t = 0; t < ntl3; ++t)
            for (int a = 0; a < 4; ++a)
                for (int S = 0; S < d.f_Kb32; ++S)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int i = lane & 15, k = 32 * S + 8 * (lane >> 4) + e;
                            double v = 0.0;
                            if (k < d.taps) {
                                if (t < nt3) v = RES[((size_t)(t * 16 + i) * 4 + a) * d.taps + k];
                                else if (a == 0 && i < 2) v = RES[((size_t)nq * 4 + i) * d.taps + k];      // rows 0, 1 = (re, im) residual of bin n_fft/4
                            }
                            rh[((((size_t)t * 4 + a) * d.f_Kb32 + S) * 64 + lane) * 8 + e] = (_Float16)(float)(v * RES_SCALE);
                        }
        int32_t *pi = reinterpret_cast<int32_t *>(packed_host + d.off_plan);          // [block][quarter q][offA j0..3 | offB j0..3]: LDS word offsets in X2
        auto off = [&](int k) { return k < 0 ? d.hop * XF_LD : (k % d.hop) * XF_LD + k / d.hop; };
        for (int S = 0; S < p3.Pb; ++S)
            for (int q = 0; q < 4; ++q)
                for (int j = 0; j < 4; ++j) {
                    const int slot = 16 * S + fold3_slot_tau(4 * q + j);
                    pi[(S * 4 + q) * 8 + j] = off(p3.ka[slot]);
                    pi[(S * 4 + q) * 8 + 4 + j] = off(p3.kb[slot]);
                }
        for (int mt = 0; mt < d.nmt; ++mt) { pi[p3.Pb * 32 + 2 * mt] = mel_kb[2 * mt]; pi[p3.Pb * 32 + 2 * mt + 1] = mel_kb[2 * mt + 1]; }
    } else if (d.fold) {
        FoldPlan pl;
        VADX_REQUIRE(fold_plan(cfg, cfg->fold, &pl) == 0, "vadx_frontend_pack_host: no fold plan for this geometry");
        std::vector<float> E, O;
        std::vector<double> RX, IX;
        double res_max = 0.0, tab_max = 0.0;
        fold_tables(cfg, pl, cos_tab, sin_tab, n_fft, E, O, RX, IX, &res_max, &tab_max);
        VADX_REQUIRE(res_max <= FOLD_MAX_RATIO * tab_max && res_max * RES_SCALE < 30000.0,
                     "vadx_frontend_pack_host: table is not symmetric enough about the window centre for cfg.fold = %d (residual %.3g of %.3g): "
                     "ask vadx_frontend_fold_kind", cfg->fold, res_max, tab_max);
        const int P = d.f_Pb * 16, ntl = d.nbt + d.nyq, last = d.n_bins - 1;
        float *fm = packed_host + d.off_fold;                      // [ntl][E rows 0..15 | O rows 16..31][P]
        for (int t = 0; t < d.nbt; ++t)
            for (int i = 0; i < 16; ++i) {
                const int f = t * 16 + i;
                if (f >= d.n_bins || (d.nyq && f == last)) continue;
                memcpy(fm + (size_t)(t * 32 + i) * P, E.data() + (size_t)f * P, P * sizeof(float));
                memcpy(fm + (size_t)(t * 32 + 16 + i) * P, O.data() + (size_t)f * P, P * sizeof(float));
            }
        int last_mode = 0;     // the last bin's tile: 0 both products, 1 only v x O, 2 only u x E
        if (d.nyq) {           // u tile: E in row 0; v tile: O in row 1
            // For the Nyquist bin one of the two parts is round-off of the reference's table (cos / sin of pi n is 0 or +-1): about a half-integer
            // centre its even part, about an integer centre its odd part.  Such a part (below 1e-3 of the table scale) joins the residual rows and
            // its 16-pair MFMAs are skipped (half of the last-bin tile: 2.5 % of the kernel's f32 MFMAs).
            double emax = 0.0, omax = 0.0;
            for (int p2 = 0; p2 < P; ++p2) { emax = fmax(emax, fabs((double)E[(size_t)last * P + p2])); omax = fmax(omax, fabs((double)O[(size_t)last * P + p2])); }
            if (emax < 1e-3 * tab_max) last_mode = 1;
            else if (omax < 1e-3 * tab_max) last_mode = 2;
            for (int p2 = 0; p2 < P && last_mode; ++p2) {
                const FoldPair &pr = pl.pairs[p2];
                if (pr.kind < 0) continue;
                float &e = E[(size_t)last * P + p2], &o = O[(size_t)last * P + p2];
                double *rx = RX.data() + (size_t)last * d.taps, *ix = IX.data() + (size_t)last * d.taps;
                if (last_mode == 1) {          // the model's real part E u goes back into RX
                    if (pr.kind == 0) { rx[pr.k] += e; rx[pr.kp] += e; } else if (pr.kind == 1) rx[pr.k] += 2.0 * e; else rx[pr.k] += e;
                    e = 0.f;
                } else {                        // the model's imaginary part O v goes back into IX
                    if (pr.kind == 0) { ix[pr.k] += o; ix[pr.kp] -= o; } else if (pr.kind == 2) ix[pr.k] += o;
                    o = 0.f;
                }
            }
            for (int t2 = 0; t2 < d.taps && last_mode; ++t2)
                VADX_REQUIRE(fabs(RX[(size_t)last * d.taps + t2]) * RES_SCALE < 30000.0 && fabs(IX[(size_t)last * d.taps + t2]) * RES_SCALE < 30000.0,
                             "vadx_frontend_pack_host: last-bin residual out of the f16 range");
            memcpy(fm + (size_t)(d.nbt * 32) * P, E.data() + (size_t)last * P, P * sizeof(float));
            memcpy(fm + (size_t)(d.nbt * 32 + 16 + 1) * P, O.data() + (size_t)last * P, P * sizeof(float));
        }
        vadx::frag_major_inplace(fm, ntl * 32, P);
        int32_t *pi = reinterpret_cast<int32_t *>(packed_host + d.off_plan);
        for (int r = 0; r < d.f_regions; ++r) { pi[4 * r] = d.f_blocks[r]; pi[4 * r + 1] = d.f_offA[r]; pi[4 * r + 2] = d.f_offB[r]; pi[4 * r + 3] = d.f_strB[r]; }
        for (int mt = 0; mt < d.nmt; ++mt) { pi[4 * MAX_REGIONS + 2 * mt] = mel_kb[2 * mt]; pi[4 * MAX_REGIONS + 2 * mt + 1] = mel_kb[2 * mt + 1]; }      // the mel bands, for the kernel
        pi[4 * MAX_REGIONS + 2 * MAX_MEL_TILES] = last_mode;
        _Float16 *rh = reinterpret_cast<_Float16 *>(packed_host + d.off_res);      // [ntl][RX | IX][Kb32][64 lanes][8]
        for (int t = 0; t < ntl; ++t)
            for (int part = 0; part < 2; ++part)
                for (int S = 0; S < d.f_Kb32; ++S)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int i = lane & 15, k = 32 * S + 8 * (lane >> 4) + e;
                            double v = 0.0;
                            if (k < d.taps) {
                                if (t < d.nbt) {
                                    const int f = t * 16 + i;
                                    if (f < d.n_bins && !(d.nyq && f == last)) v = (part ? IX : RX)[(size_t)f * d.taps + k];
                                } else if (part == 0 && i < 2) v = (i ? IX : RX)[(size_t)last * d.taps + k];
                            }
                            rh[((((size_t)t * 2 + part) * d.f_Kb32 + S) * 64 + lane) * 8 + e] = (_Float16)(float)(v * RES_SCALE);
                        }
    }
    return VADX_OK;
}

// Which fold the reference table admits by default: 1 / 2 (see "Folded DFT") or 0 -- the geometry has no plan, or the table is further from
// the symmetric model than the f16 residual may carry (FOLD_MAX_RATIO of the table scale).  The caller stores the answer in cfg.fold
// BEFORE vadx_frontend_packed_floats / vadx_frontend_pack_host / the launches.
extern "C" int vadx_frontend_fold_kind(const vadx_frontend_cfg *cfg, const float *cos_tab, const float *sin_tab, int n_fft) {
    if (!cfg || !cos_tab || !sin_tab) return 0;
    int best = 0;
    double best_res = 0.0;
    for (int kind = 1; kind <= 2; ++kind) {
        vadx_frontend_cfg c = *cfg;
        c.fold = kind;
        Dev d;
        if (c.tap0 < 0 || c.tap0 + c.taps > n_fft || derive(&c, &d)) continue;
        FoldPlan pl;
        if (fold_plan(&c, kind, &pl)) continue;
        std::vector<float> E, O;
        std::vector<double> RX, IX;
        double res_max = 0.0, tab_max = 0.0;
        fold_tables(&c, pl, cos_tab, sin_tab, n_fft, E, O, RX, IX, &res_max, &tab_max);
        if (!(res_max <= FOLD_MAX_RATIO * tab_max && res_max * RES_SCALE < 30000.0)) continue;
        if (!best || res_max < best_res) { best = kind; best_res = res_max; }
    }
    return best;
}

extern "C" int vadx_frontend_window_means(const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch, int windows_per_clip,
                                          int window_len, float scale, float *means, void *stream) {
    VADX_REQUIRE(audio && means && batch > 0 && windows_per_clip > 0 && window_len > 0, "vadx_frontend_window_means: bad argument");
    const long long nwin = (long long)batch * windows_per_clip;
    hipLaunchKernelGGL(window_mean_kernel, dim3((unsigned)((nwin + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), audio,
                       (long long)row_stride, (long long)win_stride, windows_per_clip, (int)nwin, window_len, scale, means);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

static int logmel_impl(const vadx_frontend_cfg *cfg, const float *packed, const int32_t *mel_kb_host, const int16_t *audio, int64_t row_stride,
                       int64_t win_stride, int batch, int windows_per_clip, float *means_ws, bool means_given, float *out, void *stream);

extern "C" int vadx_frontend_logmel(const vadx_frontend_cfg *cfg, const float *packed, const int32_t *mel_kb_host,
                                    const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch,
                                    int windows_per_clip, float *means_ws, float *out, void *stream) {
    return logmel_impl(cfg, packed, mel_kb_host, audio, row_stride, win_stride, batch, windows_per_clip, means_ws, false, out, stream);
}

extern "C" int vadx_frontend_logmel_means(const vadx_frontend_cfg *cfg, const float *packed, const int32_t *mel_kb_host,
                                          const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch,
                                          int windows_per_clip, const float *means, float *out, void *stream) {
    VADX_REQUIRE(means, "vadx_frontend_logmel_means: NULL means");
    return logmel_impl(cfg, packed, mel_kb_host, audio, row_stride, win_stride, batch, windows_per_clip, const_cast<float *>(means), true, out, stream);
}

static int logmel_impl(const vadx_frontend_cfg *cfg, const float *packed, const int32_t *mel_kb_host, const int16_t *audio, int64_t row_stride,
                       int64_t win_stride, int batch, int windows_per_clip, float *means_ws, bool means_given, float *out, void *stream) {
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
        if (!means_given) {             // (vadx_frontend_logmel_means: the caller computed them, e.g. with vadx_fsmn_window_stats)
            const float scale = (cfg->prep == 2) ? cfg->k1 : 1.0f;
            hipLaunchKernelGGL(window_mean_kernel, dim3((unsigned)((nwin + 3) / 4)), dim3(256), 0, st, audio, (long long)row_stride,
                               (long long)win_stride, windows_per_clip, (int)nwin, cfg->window_len, scale, means_ws);
            VADX_HIP_TRY(hipGetLastError());
        }
        means = means_ws;
    }
    if (d.fold == 4 || d.fold == 5) {
        const size_t slds = split_lds_bytes(&d);
        VADX_REQUIRE(slds <= 160 * 1024, "vadx_frontend_logmel: geometry needs %zu B of LDS", slds);
        const long long nblk = nwin * (d.tiles64 + (d.tail_mt ? 1 : 0));
        VADX_REQUIRE(nblk < (1LL << 31), "vadx_frontend_logmel: too many tiles");
        if (d.fold == 4) {
            VADX_DYN_LDS(frontend_split_kernel<vadx::SchemeB3>, 160 * 1024);
            hipLaunchKernelGGL(frontend_split_kernel<vadx::SchemeB3>, dim3((unsigned)nblk), dim3(SQ_THREADS), slds, st, d, packed, audio, (long long)row_stride,
                               (long long)win_stride, windows_per_clip, means, out);
        } else {
            VADX_DYN_LDS(frontend_split_kernel<vadx::SchemeH2>, 160 * 1024);
            hipLaunchKernelGGL(frontend_split_kernel<vadx::SchemeH2>, dim3((unsigned)nblk), dim3(SQ_THREADS), slds, st, d, packed, audio, (long long)row_stride,
                               (long long)win_stride, windows_per_clip, means, out);
        }
        VADX_HIP_TRY(hipGetLastError());
        return VADX_OK;
    }
    if (d.fold) {
        const size_t flds = fold_lds_bytes(&d);
        const long long nblk = nwin * (d.tiles64 + (d.tail_mt ? 1 : 0));
        VADX_REQUIRE(nblk < (1LL << 31), "vadx_frontend_logmel: too many tiles");
        if (d.fold == 3) {
            VADX_DYN_LDS(frontend_fold3_kernel, 80 * 1024);
            hipLaunchKernelGGL(frontend_fold3_kernel, dim3((unsigned)nblk), dim3(THREADS), flds, st, d, packed, audio, (long long)row_stride,
                               (long long)win_stride, windows_per_clip, means, out, (int)(fold_pw_off_bytes(&d) / 4));
            VADX_HIP_TRY(hipGetLastError());
            return VADX_OK;
        }
        VADX_DYN_LDS(frontend_fold_kernel, 80 * 1024);
        hipLaunchKernelGGL(frontend_fold_kernel, dim3((unsigned)nblk), dim3(THREADS), flds, st, d, packed, audio, (long long)row_stride,
                           (long long)win_stride, windows_per_clip, means, out, (int)(fold_pw_off_bytes(&d) / 4));
        VADX_HIP_TRY(hipGetLastError());
        return VADX_OK;
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
