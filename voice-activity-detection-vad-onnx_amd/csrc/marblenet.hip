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

#include <math.h>

namespace vadx {
namespace marblenet {

constexpr int THREADS = 512;
constexpr int TILE = 32, A_LD = 36;     // frames per tile, k-major row stride
constexpr int IN_LD_MAX = 100;          // largest halo tile row stride (>= 31*stride + (k-1)*dil + 1, % 8 == 4)
constexpr int MAXC = 128;
// LDS is carved per launch from the block's real shape (Cfg::in_ld, channel counts): a 64-channel k=13 block needs
// 47 KB instead of the 125 KB worst case, i.e. three workgroups per CU instead of one -- a tile is a short serial chain
// (stage -> FIR -> GEMM -> store), so with one workgroup per CU the launch was pure latency (72 rounds x ~6 us).
static size_t lds_floats(int cinp, int coutp, int cresp, int in_ld, bool has_dw) {
    return (size_t)cinp * in_ld + (has_dw ? (size_t)cinp * A_LD : 0) + (size_t)coutp * A_LD + (cresp ? (size_t)(cresp + coutp) * A_LD : 0);
}

struct Cfg {
    int cin, cout, k, stride, dil, pad, has_dw, cres, relu, cinp, coutp, cresp, in_ld;
};

// KT / DT / ST: compile-time kernel size, dilation and stride of the depthwise stage (0 = take them from `c` at run
// time: the generic fallback).  With constants the FIR is a register-window filter -- a thread owns 8 consecutive
// outputs of one channel, reads its (8-1)*stride + (k-1)*dil + 1 inputs and its k taps ONCE, and everything after
// is FMAs on registers -- and every staging load is unconditional (index clamped, value selected): a guarded load
// compiles to a branch plus a full wait, which used to serialise ~20 memory round trips per thread.
template <int KT, int DT, int ST>
__global__ __launch_bounds__(THREADS, 2) void sepconv_block_kernel(
    Cfg c, const float *__restrict__ dw_w, const float *__restrict__ pw_w, const float *__restrict__ pw_b,
    const float *__restrict__ res_w, const float *__restrict__ res_b, const float *__restrict__ x,
    long long xs_b, long long xs_c, long long xs_t, int T_in, const float *__restrict__ xres,
    float *__restrict__ y, int T_out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int IN_LD = c.in_ld;
    float *IN = lds, *D = IN + c.cinp * IN_LD, *OUT = D + (c.has_dw ? c.cinp * A_LD : 0), *RIN = OUT + c.coutp * A_LD, *ROUT = RIN + c.cresp * A_LD;
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
#pragma unroll
                for (int u = 0; u < WIN; ++u) win[u] = row[u];
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    float s2 = 0.f;
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) s2 = fmaf(wk[kk], win[o * (ST ? ST : 1) + kk * (DT ? DT : 1)], s2);
                    D[ch * A_LD + m0 + o] = ch < c.cin ? s2 : 0.f;
                }
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
    {
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
                                  int64_t xs_t, int t_in, const float *xres, float *y, int batch, int t_out, void *stream) {
    VADX_REQUIRE(cfg && pw_w && pw_b && x && y, "vadx_sepconv_block: NULL argument");
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
    const size_t lds_bytes = lds_floats(c.cinp, c.coutp, c.cresp, c.in_ld, c.has_dw) * sizeof(float);
    VADX_REQUIRE(c.has_dw ? dw_w != nullptr : (c.k == 1 && c.stride == 1), "vadx_sepconv_block: plain conv must be k=1, stride 1");
    VADX_REQUIRE(!c.cres || (res_w && res_b && xres), "vadx_sepconv_block: residual branch needs res_w/res_b/xres");
    VADX_REQUIRE(batch > 0 && t_in > 0 && t_out > 0 && t_out == (t_in + 2 * c.pad - c.dil * (c.k - 1) - 1) / c.stride + 1,
                 "vadx_sepconv_block: t_out=%d inconsistent with t_in=%d", t_out, t_in);
    const int tiles = (t_out + TILE - 1) / TILE;
    VADX_REQUIRE((long long)batch * tiles < (1LL << 31), "vadx_sepconv_block: too many tiles");
#define SEPCONV_LAUNCH(KT, DT, ST)                                                                                              \
    do {                                                                                                                        \
        VADX_DYN_LDS((sepconv_block_kernel<KT, DT, ST>), 128 * 1024);                                                           \
        hipLaunchKernelGGL((sepconv_block_kernel<KT, DT, ST>), dim3((unsigned)(batch * tiles)), dim3(THREADS),                  \
                           lds_bytes, static_cast<hipStream_t>(stream), c, dw_w, pw_w, pw_b, res_w, res_b, x,                   \
                           (long long)xs_b, (long long)xs_c, (long long)xs_t, t_in, xres, y, t_out, tiles);                      \
    } while (0)
    // the depthwise shapes of the published MarbleNet 3x2x64 get compile-time FIRs; anything else runs the generic kernel
    if (c.has_dw && c.k == 11 && c.dil == 1 && c.stride == 2) SEPCONV_LAUNCH(11, 1, 2);
    else if (c.has_dw && c.k == 13 && c.dil == 1 && c.stride == 1) SEPCONV_LAUNCH(13, 1, 1);
    else if (c.has_dw && c.k == 15 && c.dil == 1 && c.stride == 1) SEPCONV_LAUNCH(15, 1, 1);
    else if (c.has_dw && c.k == 17 && c.dil == 1 && c.stride == 1) SEPCONV_LAUNCH(17, 1, 1);
    else if (c.has_dw && c.k == 29 && c.dil == 2 && c.stride == 1) SEPCONV_LAUNCH(29, 2, 1);
    else SEPCONV_LAUNCH(0, 0, 0);
#undef SEPCONV_LAUNCH
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
