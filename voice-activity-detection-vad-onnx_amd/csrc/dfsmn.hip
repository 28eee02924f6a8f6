// dfsmn.hip -- DFSMN near+far VAD (SDAEC ICCRN echo canceller + mask-net) building blocks for gfx950.
// Reference: DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py:65-354.
//
// Activation layout ("FT", frame-tiled): a tensor with C channels over F bins and T frames is stored as
//     [tile = chunk*NT + t/16][C][F][16]        element (c, f, t) at ((tile*C + c)*F + f)*16 + t%16
// so the 16 frames of a tile are the 16 COLUMNS of every MFMA tile: every op of the ICCRN that is
// independent per frame (LayerNorm over (C,F), 1x1 / (3,1) convs, the length-160 DFT of CepsUnit, the
// bi-LSTMs that run ALONG the frequency axis) becomes an f32-MFMA GEMM whose other operand (weights,
// DFT tables) is stationary in VGPRs.  The LSTMs that run along TIME use 16 bins as the columns.
// LSTM recurrences keep h and c in registers: with gate rows ordered (unit-quad q, gate r) the D
// fragment of step t is exactly the B fragment of step t+1 -- no LDS, no shuffles.
#include "common.h"

#include <math.h>
#include <string.h>

namespace vadx {
namespace dfsmn {

struct View {            // channel slice [c_off, c_off + c) of an FT tensor with c_total channels
    const float *ptr;
    int c_total, c_off, c;
};
struct ViewW {
    float *ptr;
    int c_total, c_off, c;
};

__device__ __forceinline__ size_t ft_idx(int tile, int c_total, int c, int F, int f) {
    return (((size_t)tile * c_total + c) * F + f) * 16;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm statistics over (C, F) per frame: stats[tile][16][2] = (mean, 1/(std_unbiased + 1e-6))
// (LayerNorm.forward, Export_DFSMN_VAD.py:163-167).  Two passes (mean, then centred squares).
// ---------------------------------------------------------------------------------------------
__global__ void frame_stats_kernel(View a, View b, int F, float *__restrict__ stats) {
    __shared__ float red[16][17];
    const int tile = blockIdx.x, tid = threadIdx.x, t = tid & 15, part = tid >> 4;     // 256 threads: 16 parts
    const int n = (a.c + b.c) * F;
    auto at = [&](int e) -> float {
        const int c = e / F, f = e - c * F;
        return c < a.c ? a.ptr[ft_idx(tile, a.c_total, a.c_off + c, F, f) + t]
                       : b.ptr[ft_idx(tile, b.c_total, b.c_off + c - a.c, F, f) + t];
    };
    float s = 0.f;
    for (int e = part; e < n; e += 16) s += at(e);
    red[part][t] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int p2 = 0; p2 < 16; ++p2) mean += red[p2][t];
    mean /= (float)n;
    __syncthreads();
    float v = 0.f;
    for (int e = part; e < n; e += 16) { const float d = at(e) - mean; v = fmaf(d, d, v); }
    red[part][t] = v;
    __syncthreads();
    if (part == 0) {
        float var = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < 16; ++p2) var += red[p2][t];
        const float sd = sqrtf(var / (float)(n - 1));
        stats[((size_t)tile * 16 + t) * 2] = mean;
        stats[((size_t)tile * 16 + t) * 2 + 1] = 1.0f / (sd + 1e-6f);
    }
}

struct LN {              // LayerNorm applied on the fly to an input view: (x - mean) * inv * w[c][f] + b[c][f]
    const float *stats, *w, *b;        // stats NULL = identity
};

__device__ __forceinline__ float ln_apply(const LN &ln, int tile, int t, int cf, float x) {
    const float mean = ln.stats[((size_t)tile * 16 + t) * 2], inv = ln.stats[((size_t)tile * 16 + t) * 2 + 1];
    return (x - mean) * inv * ln.w[cf] + ln.b[cf];
}

// ---------------------------------------------------------------------------------------------
// pw_conv: convolution over channels with KF (1 or 3) taps along F ('same' zero padding), per frame.
//   rows = output channels (MT tiles of 16, zero padded), K = KF * C_in (KS k-steps of 4), cols = frames.
//   Weights W[MT*16][KF*C_in] are loaded once per wave into VGPRs; each wave walks the bins.
// MODE 0: out0 = act(conv(in) + b)                       act: 0 none, 1 sigmoid
// MODE 1: CFB front (:87-90): g = sigmoid(convG(LN(in)) + bg); xi = convI(in) + bi;
//                             out0 = g*xi; out1 = xi - g*xi
// MODE 2: CFB back  (:91-92): out0 = conv31(LN(in)) + b + add
// `in` is the channel concatenation of views a and b.
// ---------------------------------------------------------------------------------------------
struct PwArgs {
    View a, b;
    LN ln;
    const float *W, *bias;             // MODE 1: gate weights / bias
    const float *W2, *bias2;           // MODE 1: input-conv weights / bias
    View add;                          // MODE 2
    ViewW out0, out1;
    int F, co, act;
};

template <int MT, int KS, int KF, int MODE>
__global__ __launch_bounds__(256) void pw_conv_kernel(PwArgs p) {
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int q = lane >> 4, i = lane & 15;
    const int cin = p.a.c + p.b.c;
    constexpr int K = KS * 4;
    float wa[MT][KS], wb[MODE == 1 ? MT : 1][MODE == 1 ? KS : 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            wa[mt][s] = p.W[(size_t)(mt * 16 + i) * K + 4 * s + q];
            if (MODE == 1) wb[mt][s] = p.W2[(size_t)(mt * 16 + i) * K + 4 * s + q];
        }
    for (int f = wave; f < p.F; f += nw) {
        f32x4 acc[MT], acc2[MODE == 1 ? MT : 1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; if (MODE == 1) acc2[mt] = acc[mt]; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 4 * s + q;
            const int tap = (KF == 1) ? 0 : k / cin, c = (KF == 1) ? k : k - tap * cin;
            const int ff = f + tap - (KF - 1) / 2;
            float raw = 0.f, lnv = 0.f;
            if (ff >= 0 && ff < p.F) {
                raw = c < p.a.c ? p.a.ptr[ft_idx(tile, p.a.c_total, p.a.c_off + c, p.F, ff) + i]
                                : p.b.ptr[ft_idx(tile, p.b.c_total, p.b.c_off + c - p.a.c, p.F, ff) + i];
                lnv = p.ln.stats ? ln_apply(p.ln, tile, i, c * p.F + ff, raw) : raw;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = mfma16(wa[mt][s], lnv, acc[mt]);
                if (MODE == 1) acc2[mt] = mfma16(wb[mt][s], raw, acc2[mt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = mt * 16 + 4 * q + r;
                if (co < p.co) {
                    float v = acc[mt][r] + p.bias[co];
                    const size_t o0 = ft_idx(tile, p.out0.c_total, p.out0.c_off + co, p.F, f) + i;
                    if (MODE == 0) {
                        if (p.act == 1) v = sigmoidf_(v);
                        p.out0.ptr[o0] = v;
                    } else if (MODE == 1) {
                        const float g = sigmoidf_(v), xi = acc2[mt][r] + p.bias2[co], gx = g * xi;
                        p.out0.ptr[o0] = gx;
                        p.out1.ptr[ft_idx(tile, p.out1.c_total, p.out1.c_off + co, p.F, f) + i] = xi - gx;
                    } else {
                        p.out0.ptr[o0] = v + p.add.ptr[ft_idx(tile, p.add.c_total, p.add.c_off + co, p.F, f) + i];
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// dft_f: GEMM along the frequency axis, per channel:  out[c][m][t] = sum_k Tbl[m][k] * B_c[k][t]
//   FWD (CepsUnit :134-138): B_c = LN2(in)[c][f] (k = f, 160);  rows m = (cos k'<81 | sin k'<81) in
//        12 tiles of 16; out channel c <- cos rows, C + c <- sin rows, Fout = 81.
//   INV (:141-153): k = (re k'<81 | im k'<81) (164 padded); B_c = complex product of the LSTM output
//        (pr, pi) = lo[c], lo[C+c] with the raw spectrum (re, im) = li[c], li[C+c]; rows m = f (160).
// Table rows live in VGPRs (2 m-tiles per wave); the channel's B matrix is staged through LDS.
// ---------------------------------------------------------------------------------------------
struct DftArgs {
    View in;             // FWD: r (C ch, F=160).  INV: li (2C ch, F=81)
    View lo;             // INV only: LSTM+linear output (2C ch, F=81)
    LN ln;               // FWD only
    const float *tbl;    // [MTILES*16][KS*4] row-major, zero padded
    ViewW out;           // FWD: li (2C ch, 81).  INV: ceps_out (C ch, 160)
    int C;
};

template <bool INV>
__global__ __launch_bounds__(INV ? 320 : 384) void dft_f_kernel(DftArgs p) {
    constexpr int KS = INV ? 41 : 40, NWAVE = INV ? 5 : 6, KROWS = KS * 4;
    __shared__ float Bs[2][KROWS * 16];
    const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i = lane & 15;
    float ta[2][KS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < KS; ++s) ta[h][s] = p.tbl[(size_t)((wave * 2 + h) * 16 + i) * KROWS + 4 * s + q];
    auto stage = [&](int c, float *dst) {
        for (int e = tid; e < KROWS * 16; e += NWAVE * 64) {
            const int k = e >> 4, t = e & 15;
            float v = 0.f;
            if (!INV) {
                if (k < 160) v = ln_apply(p.ln, tile, t, c * 160 + k, p.in.ptr[ft_idx(tile, p.in.c_total, p.in.c_off + c, 160, k) + t]);
            } else if (k < 162) {
                const int kk = k < 81 ? k : k - 81;
                const float re = p.in.ptr[ft_idx(tile, p.in.c_total, p.in.c_off + c, 81, kk) + t];
                const float im = p.in.ptr[ft_idx(tile, p.in.c_total, p.in.c_off + p.C + c, 81, kk) + t];
                const float pr = p.lo.ptr[ft_idx(tile, p.lo.c_total, p.lo.c_off + c, 81, kk) + t];
                const float pi = p.lo.ptr[ft_idx(tile, p.lo.c_total, p.lo.c_off + p.C + c, 81, kk) + t];
                v = k < 81 ? pr * re - pi * im : pr * im + pi * re;
            }
            dst[e] = v;
        }
    };
    stage(0, Bs[0]);
    __syncthreads();
    for (int c = 0; c < p.C; ++c) {
        const float *B = Bs[c & 1];
        if (c + 1 < p.C) stage(c + 1, Bs[(c + 1) & 1]);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float bv = B[(4 * s + q) * 16 + i];
            acc[0] = mfma16(ta[0][s], bv, acc[0]);
            acc[1] = mfma16(ta[1][s], bv, acc[1]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = (wave * 2 + h) * 16 + 4 * q + r;
                if (!INV) {          // tiles 0..5 = cos rows, 6..11 = sin rows
                    const int kk = m < 96 ? m : m - 96;
                    if (kk < 81) p.out.ptr[ft_idx(tile, p.out.c_total, p.out.c_off + (m < 96 ? c : p.C + c), 81, kk) + i] = acc[h][r];
                } else {
                    p.out.ptr[ft_idx(tile, p.out.c_total, p.out.c_off + c, 160, m) + i] = acc[h][r];
                }
            }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// lstm_f: bidirectional LSTM (hidden 20) ALONG the frequency axis, batch = the tile's 16 frames
// (CH_LSTM_F :270-284: in_ch_lstm IN=4 F=160; CepsUnit IN=40 F=81 with LayerNorm on the input).
// One wave per direction.  Gate rows are ordered so that m-tile mt, fragment row 4q+r holds gate r of
// hidden unit 4*mt+q: the cell update is lane-local and the new h IS the next step's B fragment.
// Output channels: [0,20) forward h, [20,40) backward h.
// ---------------------------------------------------------------------------------------------
struct LstmFArgs {
    View in;
    LN ln;
    const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];      // per direction, torch layouts [80][IN], [80][20], [80]
    ViewW out;           // 40 channels
    int F;
};

template <int IN>
__global__ __launch_bounds__(128) void lstm_f_kernel(LstmFArgs p) {
    constexpr int KI = IN / 4, H = 20, MT = 5;
    const int tile = blockIdx.x, lane = threadIdx.x & 63, dir = threadIdx.x >> 6, q = lane >> 4, i = lane & 15;
    const int grow = (i & 3) * H + (i >> 2);          // A-fragment row i <-> gate (i&3), unit-in-quad (i>>2)
    float wi[MT][KI], wh[MT][MT], bias[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = grow + 4 * mt;
#pragma unroll
        for (int s = 0; s < KI; ++s) wi[mt][s] = p.w_ih[dir][(size_t)row * IN + 4 * s + q];
#pragma unroll
        for (int s = 0; s < MT; ++s) wh[mt][s] = p.w_hh[dir][(size_t)row * H + 4 * s + q];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[mt][r] = p.b_ih[dir][r * H + 4 * mt + q] + p.b_hh[dir][r * H + 4 * mt + q];
    }
    float h[MT], c[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { h[mt] = 0.f; c[mt] = 0.f; }
    auto load_x = [&](int f, float (&x)[KI]) {
#pragma unroll
        for (int s = 0; s < KI; ++s) {
            const int ch = 4 * s + q;
            const float v = p.in.ptr[ft_idx(tile, p.in.c_total, p.in.c_off + ch, p.F, f) + i];
            x[s] = p.ln.stats ? ln_apply(p.ln, tile, i, ch * p.F + f, v) : v;
        }
    };
    float xc[KI], xn[KI];
    load_x(dir ? p.F - 1 : 0, xc);
    for (int st = 0; st < p.F; ++st) {
        const int f = dir ? p.F - 1 - st : st;
        if (st + 1 < p.F) load_x(dir ? f - 1 : f + 1, xn);
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{bias[mt][0], bias[mt][1], bias[mt][2], bias[mt][3]};
#pragma unroll
        for (int s = 0; s < KI; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(wi[mt][s], xc[s], acc[mt]);
#pragma unroll
        for (int s = 0; s < MT; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(wh[mt][s], h[s], acc[mt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float ig = sigmoidf_(acc[mt][0]), fg = sigmoidf_(acc[mt][1]), gg = tanhf(acc[mt][2]), og = sigmoidf_(acc[mt][3]);
            c[mt] = fg * c[mt] + ig * gg;
            h[mt] = og * tanhf(c[mt]);
            p.out.ptr[ft_idx(tile, p.out.c_total, p.out.c_off + dir * H + 4 * mt + q, p.F, f) + i] = h[mt];
        }
#pragma unroll
        for (int s = 0; s < KI; ++s) xc[s] = xn[s];
    }
}

}  // namespace dfsmn
}  // namespace vadx

using namespace vadx::dfsmn;

static View mkview(const vadx_ft_view *v) { return View{v ? v->ptr : nullptr, v ? v->c_total : 0, v ? v->c_off : 0, v ? v->c : 0}; }
static ViewW mkvieww(const vadx_ft_view *v) { return ViewW{v ? const_cast<float *>(v->ptr) : nullptr, v ? v->c_total : 0, v ? v->c_off : 0, v ? v->c : 0}; }
static LN mkln(const vadx_ft_ln *l) { return LN{l ? l->stats : nullptr, l ? l->w : nullptr, l ? l->b : nullptr}; }

extern "C" int vadx_dfsmn_frame_stats(const vadx_ft_view *a, const vadx_ft_view *b, int F, int tiles, float *stats, void *stream) {
    VADX_REQUIRE(a && a->ptr && stats && F > 0 && tiles > 0, "vadx_dfsmn_frame_stats: bad argument");
    hipLaunchKernelGGL(frame_stats_kernel, dim3(tiles), dim3(256), 0, static_cast<hipStream_t>(stream), mkview(a), mkview(b), F, stats);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

template <int MT, int KS, int KF, int MODE>
static int launch_pw(const PwArgs &p, int tiles, void *stream) {
    hipLaunchKernelGGL((pw_conv_kernel<MT, KS, KF, MODE>), dim3(tiles), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

extern "C" int vadx_dfsmn_pw_conv(int mode, const vadx_ft_view *a, const vadx_ft_view *b, const vadx_ft_ln *ln,
                                  const float *w, const float *bias, const float *w2, const float *bias2,
                                  const vadx_ft_view *add, const vadx_ft_view *out0, const vadx_ft_view *out1,
                                  int F, int co, int kf, int act, int tiles, void *stream) {
    VADX_REQUIRE(a && a->ptr && w && bias && out0 && out0->ptr && F > 0 && tiles > 0 && co > 0, "vadx_dfsmn_pw_conv: bad argument");
    PwArgs p;
    p.a = mkview(a); p.b = mkview(b); p.ln = mkln(ln); p.W = w; p.bias = bias; p.W2 = w2; p.bias2 = bias2;
    p.add = mkview(add); p.out0 = mkvieww(out0); p.out1 = mkvieww(out1); p.F = F; p.co = co; p.act = act;
    const int cin = p.a.c + p.b.c, K = kf * cin, MT = (co + 15) / 16;
    VADX_REQUIRE(K % 4 == 0, "vadx_dfsmn_pw_conv: kf*cin must be a multiple of 4");
    const int KS = K / 4;
#define PW_CASE(mt, ks, kfv, md) if (MT == mt && KS == ks && kf == kfv && mode == md) return launch_pw<mt, ks, kfv, md>(p, tiles, stream)
    PW_CASE(2, 5, 1, 1);       // CFB front, 20 -> 20
    PW_CASE(2, 10, 1, 1);      // CFB front, 40 -> 20
    PW_CASE(2, 15, 3, 2);      // CFB back: conv (3,1) 20 -> 20 + ceps
    PW_CASE(2, 6, 1, 0);       // in_conv 24 -> 20
    PW_CASE(2, 10, 1, 0);      // in_ch_lstm linear 40 -> 20
    PW_CASE(3, 10, 1, 0);      // ceps linear 40 -> 40
    PW_CASE(1, 15, 1, 0);      // out_conv 60 -> 2
#undef PW_CASE
    vadx::set_error("vadx_dfsmn_pw_conv: unsupported shape (co=%d cin=%d kf=%d mode=%d)", co, cin, kf, mode);
    return VADX_EINVAL;
}

extern "C" int vadx_dfsmn_dft_f(int inverse, const vadx_ft_view *in, const vadx_ft_view *lo, const vadx_ft_ln *ln,
                                const float *tbl, const vadx_ft_view *out, int C, int tiles, void *stream) {
    VADX_REQUIRE(in && in->ptr && tbl && out && out->ptr && C > 0 && tiles > 0, "vadx_dfsmn_dft_f: bad argument");
    VADX_REQUIRE(inverse ? (lo && lo->ptr) : (ln && ln->stats), "vadx_dfsmn_dft_f: missing lo / ln");
    DftArgs p;
    p.in = mkview(in); p.lo = mkview(lo); p.ln = mkln(ln); p.tbl = tbl; p.out = mkvieww(out); p.C = C;
    if (inverse) hipLaunchKernelGGL(dft_f_kernel<true>, dim3(tiles), dim3(320), 0, static_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(dft_f_kernel<false>, dim3(tiles), dim3(384), 0, static_cast<hipStream_t>(stream), p);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

extern "C" int vadx_dfsmn_lstm_f(const vadx_ft_view *in, const vadx_ft_ln *ln, const float *const w_ih[2],
                                 const float *const w_hh[2], const float *const b_ih[2], const float *const b_hh[2],
                                 const vadx_ft_view *out, int F, int tiles, void *stream) {
    VADX_REQUIRE(in && in->ptr && out && out->ptr && w_ih && w_hh && b_ih && b_hh && F > 0 && tiles > 0, "vadx_dfsmn_lstm_f: bad argument");
    LstmFArgs p;
    p.in = mkview(in); p.ln = mkln(ln); p.out = mkvieww(out); p.F = F;
    for (int d = 0; d < 2; ++d) { p.w_ih[d] = w_ih[d]; p.w_hh[d] = w_hh[d]; p.b_ih[d] = b_ih[d]; p.b_hh[d] = b_hh[d]; }
    if (in->c == 4) hipLaunchKernelGGL(lstm_f_kernel<4>, dim3(tiles), dim3(128), 0, static_cast<hipStream_t>(stream), p);
    else if (in->c == 40) hipLaunchKernelGGL(lstm_f_kernel<40>, dim3(tiles), dim3(128), 0, static_cast<hipStream_t>(stream), p);
    else { vadx::set_error("vadx_dfsmn_lstm_f: input channels must be 4 or 40"); return VADX_EINVAL; }
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
