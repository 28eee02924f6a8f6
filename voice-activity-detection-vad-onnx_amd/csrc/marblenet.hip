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
constexpr int IN_LD = 100;              // halo tile row stride (>= 31*stride + (k-1)*dil + 1, % 8 == 4)
constexpr int MAXC = 128;
constexpr int IN_F = MAXC * IN_LD, T_F = MAXC * A_LD;
constexpr int LDS_FLOATS = IN_F + 4 * T_F;

struct Cfg {
    int cin, cout, k, stride, dil, pad, has_dw, cres, relu, cinp, coutp, cresp;
};

__global__ __launch_bounds__(THREADS, 2) void sepconv_block_kernel(
    Cfg c, const float *__restrict__ dw_w, const float *__restrict__ pw_w, const float *__restrict__ pw_b,
    const float *__restrict__ res_w, const float *__restrict__ res_b, const float *__restrict__ x,
    long long xs_b, long long xs_c, long long xs_t, int T_in, const float *__restrict__ xres,
    float *__restrict__ y, int T_out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *IN = lds, *D = lds + IN_F, *OUT = D + T_F, *RIN = OUT + T_F, *ROUT = RIN + T_F;
    const int tid = threadIdx.x;
    const int b = blockIdx.x / tiles, t0 = (blockIdx.x - b * tiles) * TILE;
    const int width = (TILE - 1) * c.stride + (c.k - 1) * c.dil + 1;
    const float *xb = x + (long long)b * xs_b;
    const int tin0 = t0 * c.stride - c.pad;
    if (xs_c == 1) {            // time-major source (front-end output): channel fastest
        for (int e = tid; e < width * c.cinp; e += THREADS) {
            const int j = e / c.cinp, ch = e - j * c.cinp, ti = tin0 + j;
            IN[ch * IN_LD + j] = (ch < c.cin && ti >= 0 && ti < T_in) ? xb[(long long)ti * xs_t + ch] : 0.f;
        }
    } else {                    // channel-first source: time fastest
        for (int e = tid; e < width * c.cinp; e += THREADS) {
            const int ch = e / width, j = e - ch * width, ti = tin0 + j;
            IN[ch * IN_LD + j] = (ch < c.cin && ti >= 0 && ti < T_in) ? xb[(long long)ch * xs_c + (long long)ti * xs_t] : 0.f;
        }
    }
    if (c.cres) {
        const float *rb = xres + (long long)b * c.cres * T_out;
        for (int e = tid; e < c.cresp * TILE; e += THREADS) {
            const int ch = e / TILE, m = e - ch * TILE;
            RIN[ch * A_LD + m] = (ch < c.cres && t0 + m < T_out) ? rb[(long long)ch * T_out + t0 + m] : 0.f;
        }
    }
    __syncthreads();
    const float *act = IN;
    int lda = IN_LD;
    if (c.has_dw) {
        for (int e = tid; e < c.cinp * TILE; e += THREADS) {
            const int ch = e / TILE, m = e - ch * TILE;
            float s = 0.f;
            if (ch < c.cin) {
                const float *row = IN + ch * IN_LD + m * c.stride;
                const float *wk = dw_w + ch * c.k;
                for (int kk = 0; kk < c.k; ++kk) s = fmaf(wk[kk], row[kk * c.dil], s);
            }
            D[ch * A_LD + m] = s;
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
    VADX_REQUIRE(c.k >= 1 && c.stride >= 1 && c.dil >= 1 && (TILE - 1) * c.stride + (c.k - 1) * c.dil + 1 <= IN_LD,
                 "vadx_sepconv_block: receptive field of a 32-frame tile exceeds %d samples", IN_LD);
    VADX_REQUIRE(c.has_dw ? dw_w != nullptr : (c.k == 1 && c.stride == 1), "vadx_sepconv_block: plain conv must be k=1, stride 1");
    VADX_REQUIRE(!c.cres || (res_w && res_b && xres), "vadx_sepconv_block: residual branch needs res_w/res_b/xres");
    VADX_REQUIRE(batch > 0 && t_in > 0 && t_out > 0 && t_out == (t_in + 2 * c.pad - c.dil * (c.k - 1) - 1) / c.stride + 1,
                 "vadx_sepconv_block: t_out=%d inconsistent with t_in=%d", t_out, t_in);
    static bool done = false;
    if (!done) {
        VADX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sepconv_block_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * sizeof(float)));
        done = true;
    }
    const int tiles = (t_out + TILE - 1) / TILE;
    VADX_REQUIRE((long long)batch * tiles < (1LL << 31), "vadx_sepconv_block: too many tiles");
    hipLaunchKernelGGL(sepconv_block_kernel, dim3((unsigned)(batch * tiles)), dim3(THREADS), LDS_FLOATS * sizeof(float),
                       static_cast<hipStream_t>(stream), c, dw_w, pw_w, pw_b, res_w, res_b, x, (long long)xs_b,
                       (long long)xs_c, (long long)xs_t, t_in, xres, y, t_out, tiles);
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
