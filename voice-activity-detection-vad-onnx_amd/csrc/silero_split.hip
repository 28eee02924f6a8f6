// silero_split.hip -- Silero-VAD v5 (16 kHz) encoder tile kernel on bf16 x 3 split products (csrc/split3.h), gfx950.
//
// Same work decomposition, same inputs and the same gx output as silero_encode_kernel (csrc/silero.hip): one workgroup per
// (16 clips) x (1 window), STFT -> |.| -> conv1..conv4 (+ReLU) -> W_ih x + b.  What changes is the arithmetic of every GEMM whose one
// operand is a constant (all of them but the STFT's time-folded operand sums): weights split offline into three bf16 planes, activations
// split once by the lanes that produce them, six v_mfma_f32_16x16x32_bf16 per K = 32 step in place of eight v_mfma_f32_16x16x4_f32 --
// 6/16 of the matrix time at float32-class accuracy (split3.h), with the VALU work hiding beside the matrix pipe instead of adding to it.
//   * STFT: the folded float32 pass of silero.hip, operands swapped so that a lane ends up with four consecutive bins of one clip.
//   * conv1 is the direct three-tap conv (a tap fragment serves up to four frames), 1920 bf16 MFMAs; conv2 480, conv3 96, conv4 96,
//     W_ih 768: 3360 bf16 + 1024 f32 MFMAs per tile against 4480 f32.
//   * activations live in LDS as three bf16 planes [k / 8][16 clips][8]: a wave's B-fragment read is one contiguous 1 KiB
//     (ds_read_b128, conflict-free), a producer lane's four consecutive channels one 8-byte store per plane.
//   * weights stream from L2 as 1 KiB fragments, 976 KB per tile (877 KB for the f32 kernel); tools/l2_stream_probe.sh: every CU
//     streaming the same blob sustains 33 - 35 TB/s, this kernel needs ~20.
// Reference being reproduced: the `session.run` of Silero/modeling_modified/utils_vad.py:116-119 (see silero.hip).
#include "silero_common.h"
#include "split3.h"

// VADX_EXP: development-only what-if switches for tools/exp_encoder.py (results are wrong when set): bit 3 no conv2..4 MFMAs, 4 no STFT
// MFMAs, 5 no conv1 MFMAs, 6 no W_ih MFMAs, 11 no W_ih phase at all (neither fragments nor MFMAs), 13 every weight fragment from one address (L1 instead of L2), 12 no activation splits
// (planes written from the raw bits), 14 per-phase cycle accounting of wave 0 (sp_dbg, read with vadx_silero_split_debug_cycles)
#ifndef VADX_EXP
#define VADX_EXP 0
#endif
#define SP_SKIP(n) ((VADX_EXP >> (n)) & 1)
// VADX_SPLIT_STFT = 1 (or bit 15 of VADX_EXP): the folded STFT itself on split products -- built, parity-green on every test of
// tests/test_gpu_silero.py, and measured EQUAL to the f32-MFMA fold (5.60 against 5.62 ms per launch): it issues 96 bf16 instead of 128
// f32 MFMAs per wave and tile but streams 384 KB of table fragments per tile instead of 128 KB, and the fragment stream out of L2 is what
// holds this kernel (every fragment from one L1-resident address: 4.10 ms with this STFT, 4.65 with the f32 fold, 5.95 as shipped on
// the same box).  Not the default.
#ifndef VADX_SPLIT_STFT
#define VADX_SPLIT_STFT ((VADX_EXP >> 15) & 1)
#endif
#define SP_W(addr) (SP_SKIP(13) ? (P + vadx::silero::OFF_Q1) : (addr))        // what-if: every weight fragment from one (L1-resident) address
#if (VADX_EXP >> 14) & 1
__device__ unsigned long long sp_dbg[16];
#define SP_T0() long long sp_t_ = __builtin_readcyclecounter(); const long long sp_c0_ = sp_t_, sp_w0_ = wall_clock64()
// slots 14 / 15: shader cycles and 100 MHz ticks of the whole workgroup -> the clock the kernel sustains
#define SP_CLK() do { if (threadIdx.x == 0) { atomicAdd(&sp_dbg[14], (unsigned long long)(__builtin_readcyclecounter() - sp_c0_)); atomicAdd(&sp_dbg[15], (unsigned long long)(wall_clock64() - sp_w0_)); } } while (0)
#define SP_MARK(slot) do { if (threadIdx.x == 0) { const long long n_ = __builtin_readcyclecounter(); atomicAdd(&sp_dbg[slot], (unsigned long long)(n_ - sp_t_)); sp_t_ = n_; } } while (0)
extern "C" int vadx_silero_split_debug_cycles(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sp_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sp_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define SP_T0() do {} while (0)
#define SP_MARK(slot) do {} while (0)
#define SP_CLK() do {} while (0)
#endif

namespace vadx {
namespace silero {

constexpr int SP_THREADS = 512;
// ---- LDS map (BYTES): 75 776 B per workgroup => two workgroups per CU (eight waves of <= 128 VGPRs each)
//   R0 [0, 49152): X f32 [16 clips][642]  ->  |X| planes [3][4 frames][16 k-groups][16 clips][8 bf16]  ->  conv1 output planes (same shape)
//                  ->  conv3 output planes [3][8][16][8] at 0 and conv4 output planes [3][16][16][8] at 8192
//   R1 [49152, 61440): conv2 output planes [3][2 frames][8 k-groups][16][8]
//   scratch f32 [512]: bin-64 partial sums [8 waves][re|im][16 clips]; Nyquist magnitudes [4 frames][16 clips] at +256
//   stash 12 288 B: the first tile's conv4 planes (a workgroup encodes TWO tiles one after the other and runs W_ih once for both: every
//                   W_ih fragment -- 384 KB per tile, the phase that sat at the L2 ceiling -- then serves 32 columns instead of 16)
//   exchange f32: partial sums of the waves that own the other half of K -- conv2's inside its own output region, conv3's in R0
constexpr int SP_PL128 = 16384, SP_FR128 = 4096;
constexpr int SP_R1 = 49152, SP_PL2 = 4096, SP_FR2 = 2048;
constexpr int SP_C3 = 0, SP_PL3 = 2048;
constexpr int SP_C4 = 8192, SP_PL4 = 4096;
constexpr int SP_SCR = SP_R1 + 12288;
constexpr int SP_STASH = SP_SCR + 2048;     // conv4 output planes of the pair's FIRST tile, kept for the joint W_ih phase (12 288 B)
constexpr int SP_EXC2 = SP_R1;              // conv2's K-half exchange (8 KB) sits in conv2's own output region: read, barrier, then the planes
constexpr int SP_EXC3 = 32768;              // conv3's (4 KB) in R0 behind its output (conv1's planes are dead by then)
constexpr int SP_LDS_BYTES = SP_STASH + 12288;
static_assert(16 * X_LDM * 4 <= SP_R1 && 3 * SP_PL128 <= SP_R1 && SP_C4 + 3 * SP_PL4 <= SP_R1 && SP_C3 + 3 * SP_PL3 <= SP_C4 &&
              2 * SP_LDS_BYTES <= 160 * 1024, "split encoder LDS map");

// byte offset of (k-group kg8 = k / 8, clip) inside one plane of one frame
__device__ __forceinline__ int pl_off(int kg8, int clip) { return (kg8 * 16 + clip) * 16; }

// a producer lane's four consecutive channels 4 g .. 4 g + 3 of clip i: one 8-byte store into each of the three planes
__device__ __forceinline__ void store_split4(unsigned char *base, int plane_stride, int g, int i, const f32x4 v) {
    u32x2 p0, p1, p2;
    split3x4(v, p0, p1, p2);
    unsigned char *d = base + pl_off(g >> 1, i) + (g & 1) * 8;
    *reinterpret_cast<u32x2 *>(d) = p0;
    *reinterpret_cast<u32x2 *>(d + plane_stride) = p1;
    *reinterpret_cast<u32x2 *>(d + 2 * plane_stride) = p2;
}
__device__ __forceinline__ void store_split1(unsigned char *base, int plane_stride, int slot, int i, float v) {
    unsigned short h0, h1, h2;
    split3x1(v, h0, h1, h2);
    unsigned char *d = base + pl_off(slot >> 3, i) + (slot & 7) * 2;
    *reinterpret_cast<unsigned short *>(d) = h0;
    *reinterpret_cast<unsigned short *>(d + plane_stride) = h1;
    *reinterpret_cast<unsigned short *>(d + 2 * plane_stride) = h2;
}
// the wave's B fragments (three planes) of the 32-k chunk kc: lane 16 q + i reads k-group 4 kc + q of clip i
__device__ __forceinline__ void load_b3(bf16x8 (&b)[3], const unsigned char *base, int plane_stride, int kc, int q, int i) {
    const unsigned char *s = base + pl_off(4 * kc + q, i);
#pragma unroll
    for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8 *>(s + p * plane_stride);
}
__device__ __forceinline__ void load_a3(bf16x8 (&a)[3], const float *frag3, int lane) {
#pragma unroll
    for (int p = 0; p < 3; ++p) a[p] = ldq(frag3 + p * QF, lane);
}

// HALVES = 2 (an experiment kept for the record, not the default -- see VADX_SPLIT_HALVES): a 1024-thread workgroup whose two halves
// encode two neighbouring tiles in lockstep, each in its own LDS region.  Same registers, same LDS and the same 16 waves per CU as two
// 512-thread workgroups, but the halves stream the SAME weight fragments at the same moment, so the second request of every line is an
// L1 hit instead of an L2 read (the weight stream out of L2 is the kernel's largest stall: tools/exp_encoder.py, "every fragment from
// one address" 5.77 -> 4.51 ms).
template <typename SampleT, int HALVES, int NSUB>
__global__ __launch_bounds__(SP_THREADS * HALVES, 4) void silero_encode_split_kernel(
    const float *__restrict__ P, const SampleT *__restrict__ audio, float in_scale, long long n_samples,
    long long row_stride, long long origin, int B, int G, int T, int Gws, int g0, float *__restrict__ gx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int half = HALVES > 1 ? (int)(threadIdx.x >> 9) : 0;
    unsigned char *smem = smem_all + half * SP_LDS_BYTES;
    float *X = reinterpret_cast<float *>(smem);
    float *scr = reinterpret_cast<float *>(smem + SP_SCR), *nyq = scr + 256;

    int tid0 = threadIdx.x & (SP_THREADS - 1);
    const long long ntile = (long long)G * T;
    const bool fold = P[OFF_FOLD] != 0.f;      // uniform: the basis has the DFT symmetries -> folded STFT pass
    SP_T0();
#pragma unroll 1
    for (int sub = 0; sub < NSUB; ++sub) {
    // per tile: nothing derived from the thread index is hoisted out of the tile loop (every phase's per-lane LDS offsets and fragment
    // pointers are loop-invariant; hoisted, they spilled 1.8 KB per lane)
    asm volatile("" : "+v"(tid0));
    const int tid = tid0, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, i = lane & 15;
    // the workgroup's tiles: an odd tile count leaves the last workgroup's last slot without work -- it recomputes the last tile (the
    // barriers are workgroup-wide) and stores nothing
    const long long tile_raw = ((long long)blockIdx.x * HALVES + half) * NSUB + sub;
    const bool tile_ok = tile_raw < ntile;
    const int tile_id = (int)(tile_ok ? tile_raw : ntile - 1);
    const int grp = tile_id % G, t = tile_id / G;

    // ---------------- phase 0: the 16 windows (576 samples each) + right reflect pad of 64 (as silero_encode_kernel stages them:
    // folded pass -> even / odd samples in separate planes of the clip row)
    auto xslot = [fold](int pp) { return fold ? (pp & 1) * X_ODD + (pp >> 1) : pp; };
    {
        const long long base = (long long)t * 512 + origin;
        const bool vec_ok = ((row_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(audio) & (SampleIO<SampleT>::VEC_ALIGN - 1)) == 0) && n_samples >= 4;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const bool fast = fold && vec_ok && base >= 0 && base + 576 <= n_samples && (long long)grp * 16 + 16 <= B;
        if (fast) {     // wave w stages clips 2w and 2w+1: wave-uniform row base + 16 * lane bytes, six loads back to back
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
        } else {        // edge windows, short clips, groups past the batch: clamped unconditional loads, patched per element
            f32x4 x4[5];
            if (vec_ok) {
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    const int e = min(tid + SP_THREADS * it, 16 * 144 - 1), c = e / 144, p = 4 * (e - c * 144);
                    const long long b = (long long)grp * 16 + c, idx = base + p;
                    const SampleT *src = audio + (b < B ? b : 0) * row_stride;
                    const long long idc = idx < 0 ? 0 : (idx + 3 < n_samples ? idx : ((n_samples - 4) & ~3LL));
                    x4[it] = SampleIO<SampleT>::load4(src + idc, in_scale);
                }
            }
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int e = tid + SP_THREADS * it;             // 16 clips x 144 float4
                if (e < 16 * 144) {
                    const int c = e / 144, p = 4 * (e - c * 144);
                    const long long b = (long long)grp * 16 + c;
                    const bool bvalid = b < B;
                    const SampleT *src = audio + (bvalid ? b : 0) * row_stride;
                    const long long idx = base + p;
                    float v[4];
                    if (vec_ok && bvalid && idx >= 0 && idx + 3 < n_samples) {
                        v[0] = x4[it][0]; v[1] = x4[it][1]; v[2] = x4[it][2]; v[3] = x4[it][3];
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
        }
    }
    __syncthreads();
    SP_MARK(0);

    // ---------------- phase 1: STFT (float32 MFMAs, table = A operand) -> magnitudes -> the three bf16 planes of conv1's input.
    // Input-channel slot s of conv1: s <= 64 = bin s, s = 64 + k = bin 128 - k; bin 128 (Nyquist) goes to the scratch.
    if (fold && VADX_SPLIT_STFT) {
        // ---- the folded STFT itself on split products.  The fold's operands are sums / differences of sample PAIRS: per frame f and pair
        // index n = 1..128, e = x[128 f + n] + x[128 f + 256 - n] (cos part), o = the difference (sin part), in two classes (even / odd n:
        // the frequency fold).  Thread (clip, class, four consecutive pairs of the class) reads its 32 samples out of X ONCE, into
        // registers; the four frames then go in two pairs: its e / o values as 8-byte stores into the pair's B planes (which take X's
        // place: [E e | E o | O e | O o][plane][k-group][2 frames x 16 clips][8]) | barrier | wave (bin tile, frame of the pair): four
        // accumulators (E re, E im, O re, O im) x two k-steps x six products | barrier.  96 bf16 MFMAs per wave and tile instead of 128 f32,
        // no operand arithmetic beside the matrix instructions.
        const int cls = wave >> 2, pc = tid & 15, pj = (tid >> 4) & 15;
        float xa[4][4], xb[4][4];                     // [frame][k]: the pair's two samples (plane indices: see stft_fold_class)
        {
            const float *row = X + pc * X_LDM + (cls ? X_ODD : 1) + 4 * pj, *rowb = X + pc * X_LDM + (cls ? X_ODD : 0) + 127 - 4 * pj;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int k = 0; k < 4; ++k) { xa[f][k] = row[64 * f + k]; xb[f][k] = rowb[64 * f - k]; }
        }
        const int tl = wave >> 1, ct = wave & 1;      // GEMM role: bins 16 tl + 4 q + r (and 128 - them), frame 2 pair + ct, clip i
        float x0p[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) x0p[pr] = X[i * X_LDM + 64 * (2 * pr + ct)];        // n = 0 belongs to the even class
        float b64re = 0.f, b64im = 0.f;               // bin 64 (its own mirror) on the VALU: wave = (frame wave & 3, half of n = 1..128)
        {
            const int f = wave & 3, h = wave >> 2, n0 = h * 64 + q * 16;
            const float *xr = X + i * X_LDM + 64 * f;
            f32x4 cre = ldg4(P + OFF_B64 + n0), cim = ldg4(P + OFF_B64 + 128 + n0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int un = u + 1 < 4 ? u + 1 : u;
                const f32x4 nre = ldg4(P + OFF_B64 + n0 + 4 * un), nim = ldg4(P + OFF_B64 + 128 + n0 + 4 * un);
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
        const f32x4 c0 = ldg4(P + OFF_S0 + tl * 16 + 4 * q), s0 = ldg4(P + OFF_S0 + 64 + tl * 16 + 4 * q);
        __syncthreads();          // every sample is in registers: the pair planes may overwrite X
        SP_MARK(1);
        f32x4 mk[2], mn[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            // ---- this thread's e / o values of the pair's two frames -> planes
#pragma unroll
            for (int fl = 0; fl < 2; ++fl) {
                f32x4 ev, ov;
#pragma unroll
                for (int k = 0; k < 4; ++k) { ev[k] = xa[2 * pr + fl][k] + xb[2 * pr + fl][k]; ov[k] = xa[2 * pr + fl][k] - xb[2 * pr + fl][k]; }
                unsigned char *d = smem + ((2 * cls) * 3 * 8 + (pj >> 1)) * 512 + (16 * fl + pc) * 16 + 8 * (pj & 1);
                u32x2 p0, p1, p2;
                split3x4(ev, p0, p1, p2);
                *reinterpret_cast<u32x2 *>(d) = p0;
                *reinterpret_cast<u32x2 *>(d + 8 * 512) = p1;
                *reinterpret_cast<u32x2 *>(d + 16 * 512) = p2;
                split3x4(ov, p0, p1, p2);
                *reinterpret_cast<u32x2 *>(d + 24 * 512) = p0;
                *reinterpret_cast<u32x2 *>(d + 32 * 512) = p1;
                *reinterpret_cast<u32x2 *>(d + 40 * 512) = p2;
            }
            __syncthreads();
            // ---- the pair's GEMM: (class, part) = (E re, E im, O re, O im) x two k-steps
            f32x4 hi[4], lo[4];
            hi[0] = c0 * x0p[pr];
            hi[1] = s0 * x0p[pr];
            hi[2] = f32x4{0.f, 0.f, 0.f, 0.f};
            hi[3] = hi[2];
#pragma unroll
            for (int a4 = 0; a4 < 4; ++a4) lo[a4] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!SP_SKIP(4)) {
                const float *wq = P + OFF_QSF + (size_t)tl * (2 * 2 * 2 * 3 * QF);
                bf16x8 a[2][3];
                load_a3(a[0], SP_W(wq), lane);
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {          // s8 = (class, part, chunk): the fragment order of OFF_QSF
                    if (s8 + 1 < 8) load_a3(a[(s8 + 1) & 1], SP_W(wq + (s8 + 1) * 3 * QF), lane);
                    bf16x8 b[3];
                    const unsigned char *bp = smem + ((s8 >> 1) * 3 * 8 + 4 * (s8 & 1) + q) * 512 + (16 * ct + i) * 16;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const bf16x8 *>(bp + pl * 8 * 512);
                    mfma_split6(a[s8 & 1], b, hi[s8 >> 1], lo[s8 >> 1]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ere = hi[0][r] + lo[0][r], eim = hi[1][r] + lo[1][r], ore = hi[2][r] + lo[2][r], oim = hi[3][r] + lo[3][r];
                const float pre = ere + ore, pim = eim + oim, nre = ere - ore, nim = eim - oim;
                mk[pr][r] = mag_sqrt(pre * pre + pim * pim);
                mn[pr][r] = mag_sqrt(nre * nre + nim * nim);
            }
            __syncthreads();      // every wave is done reading the pair's planes
        }
        SP_MARK(1);
        const int g = 4 * tl + q;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int f = 2 * pr + ct;
            unsigned char *fr = smem + f * SP_FR128;
            store_split4(fr, SP_PL128, g, i, mk[pr]);                 // slots 4 g + r        = bins 4 g + r
            store_split4(fr, SP_PL128, 16 + g, i, mn[pr]);            // slots 64 + 4 g + r   = bins 128 - (4 g + r); g = 0, r = 0 is bin 128:
            if (g == 0) nyq[f * 16 + i] = mn[pr][0];                  //   it goes to the scratch, and slot 64 is rewritten below with bin 64
        }
        if (q == 0) { scr[wave * 32 + i] = b64re; scr[wave * 32 + 16 + i] = b64im; }
        __syncthreads();
        SP_MARK(2);
        if (tid < 64) {                                               // bin 64: frame tid / 16, clip tid % 16
            const int f = tid >> 4, c = tid & 15;
            const float re = scr[f * 32 + c] + scr[(f + 4) * 32 + c], im = scr[f * 32 + 16 + c] + scr[(f + 4) * 32 + 16 + c];
            store_split1(smem + f * SP_FR128, SP_PL128, 64, c, mag_sqrt(re * re + im * im));
        }
    } else if (fold) {
        const int tl = wave & 3, fp = wave >> 2;      // wave = (bin tile tl, frame pair fp): bins k = 16 tl + 4 q + r and 128 - k
        float b64re = 0.f, b64im = 0.f;               // bin 64 (its own mirror) on the VALU: wave = (frame wave & 3, half of n = 1..128)
        {
            const int f = wave & 3, h = wave >> 2, n0 = h * 64 + q * 16;
            const float *xr = X + i * X_LDM + 64 * f;
            f32x4 cre = ldg4(P + OFF_B64 + n0), cim = ldg4(P + OFF_B64 + 128 + n0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int un = u + 1 < 4 ? u + 1 : u;
                const f32x4 nre = ldg4(P + OFF_B64 + n0 + 4 * un), nim = ldg4(P + OFF_B64 + 128 + n0 + 4 * un);
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
        __builtin_amdgcn_sched_barrier(0);
        f32x4 ere[2], eim[2], ore[2], oim[2];
        const f32x4 c0 = ldg4(P + OFF_S0 + tl * 16 + 4 * q), s0 = ldg4(P + OFF_S0 + 64 + tl * 16 + 4 * q);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const float x0 = X[i * X_LDM + 64 * (2 * fp + f)];                        // n = 0 belongs to the even class
            ere[f] = c0 * x0;
            eim[f] = s0 * x0;
            ore[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            oim[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!SP_SKIP(4)) {
            const float *row = X + i * X_LDM + 128 * fp;
            const float *wt = P + OFF_SF + tl * 16 * FRAG + lane * 4;                 // [E|O][re|im][4 blocks]
            stft_fold_class<true>(ere, eim, row + 1 + q, row + 127 - q, wt, wt + 4 * FRAG);
            stft_fold_class<true>(ore, oim, row + X_ODD + q, row + X_ODD + 127 - q, wt + 8 * FRAG, wt + 12 * FRAG);
        }
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
        __syncthreads();          // every wave is done reading X: the planes may now overwrite it
        SP_MARK(1);
        const int g = 4 * tl + q;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            unsigned char *fr = smem + (2 * fp + f) * SP_FR128;
            store_split4(fr, SP_PL128, g, i, mk[f]);                  // slots 4 g + r        = bins 4 g + r
            store_split4(fr, SP_PL128, 16 + g, i, mn[f]);             // slots 64 + 4 g + r   = bins 128 - (4 g + r); g = 0, r = 0 is bin 128:
            if (g == 0) nyq[(2 * fp + f) * 16 + i] = mn[f][0];        //   it goes to the scratch, and slot 64 is rewritten below with bin 64
        }
        if (q == 0) { scr[wave * 32 + i] = b64re; scr[wave * 32 + 16 + i] = b64im; }
        __syncthreads();
        SP_MARK(2);
        if (tid < 64) {                                               // bin 64: frame tid / 16, clip tid % 16
            const int f = tid >> 4, c = tid & 15;
            const float re = scr[f * 32 + c] + scr[(f + 4) * 32 + c], im = scr[f * 32 + 16 + c] + scr[(f + 4) * 32 + 16 + c];
            store_split1(smem + f * SP_FR128, SP_PL128, 64, c, mag_sqrt(re * re + im * im));
        }
    } else {
        // dense pass (a basis without the DFT symmetries): wave w = bins 16 w .. 16 w + 15, all four frames
        f32x4 acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *const wrow[2] = {P + OFF_STFT + (wave * 32 + i) * 256, P + OFF_STFT + (wave * 32 + 16 + i) * 256};
        const int koff[4] = {0, 128, 256, 384};
        gemm_pass_mmajor<2, 4, 16, true>(acc, X, X_LDM, koff, wrow, lane);
        float nyqv = 0.f;
        if (wave < 4) {   // Nyquist bin: frame f = wave, lane = (clip i, k-quarter q)
            const float *nre = P + OFF_NYQ + q * 64, *nim = P + OFF_NYQ + 256 + q * 64;
            const float *xp = X + i * X_LDM + 128 * wave + q * 64;
            float sre = 0.f, sim = 0.f;
#pragma unroll 8
            for (int k = 0; k < 64; ++k) {
                const float x = xp[k];
                sre = fmaf(x, nre[k], sre);
                sim = fmaf(x, nim[k], sim);
            }
            sre += __shfl_xor(sre, 16); sre += __shfl_xor(sre, 32);
            sim += __shfl_xor(sim, 16); sim += __shfl_xor(sim, 32);
            nyqv = mag_sqrt(sre * sre + sim * sim);
        }
        __syncthreads();          // every wave is done reading X
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            f32x4 m;
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = mag_sqrt(acc[0][f][r] * acc[0][f][r] + acc[1][f][r] * acc[1][f][r]);
            unsigned char *fr = smem + f * SP_FR128;
            if (wave < 4) store_split4(fr, SP_PL128, 4 * wave + q, i, m);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bin = 16 * wave + 4 * q + r;
                    store_split1(fr, SP_PL128, bin == 64 ? 64 : 192 - bin, i, m[r]);
                }
            }
        }
        if (wave < 4 && q == 0) nyq[wave * 16 + i] = nyqv;
    }
    __syncthreads();
    SP_MARK(3);

    // ---------------- phase 2: conv1 129->128, k3 s1 p1, ReLU -- direct: out[f] = sum_tap W[tap] in[f + tap - 1], wave = 16 output channels
    {
        const int rt = wave;
        f32x4 hi[4], lo[4];
        {   // bias + input channel 128 (the Nyquist bin) on the VALU
            const f32x4 bias = ldg4(P + OFF_B1 + 16 * rt + 4 * q);
            f32x4 wn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wn[r] = ldg4(P + OFF_Q1N + (16 * rt + 4 * q + r) * 4);
            float nq[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) nq[f] = nyq[f * 16 + i];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = fmaf(wn[r][1], nq[f], bias[r]);
                    if (f > 0) v = fmaf(wn[r][0], nq[f - 1], v);
                    if (f < 3) v = fmaf(wn[r][2], nq[f + 1], v);
                    hi[f][r] = v;
                }
                lo[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float *wq = P + OFF_Q1 + rt * (4 * 3 * 3 * QF);
        bf16x8 a[2][3];
        load_a3(a[0], SP_W(wq), lane);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            bf16x8 b[4][3];
#pragma unroll
            for (int f = 0; f < 4; ++f) load_b3(b[f], smem + f * SP_FR128, SP_PL128, kc, q, i);
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int s = kc * 3 + tap;
                if (s + 1 < 12) load_a3(a[(s + 1) & 1], SP_W(wq + (s + 1) * 3 * QF), lane);
                const bf16x8 (&ac)[3] = a[s & 1];
                // six products per (frame, tap), frames innermost so that consecutive MFMAs hit different accumulators
#define SP_TERM(AP, BP, ACC)                                                                  \
    _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                           \
        const int fi = f + tap - 1;                                                           \
        if (fi >= 0 && fi < 4 && !SP_SKIP(5)) ACC[f] = mfma_bf16(ac[AP], b[fi][BP], ACC[f]);  \
    }
                SP_TERM(2, 0, lo) SP_TERM(1, 1, lo) SP_TERM(0, 2, lo) SP_TERM(1, 0, lo) SP_TERM(0, 1, lo) SP_TERM(0, 0, hi)
#undef SP_TERM
            }
        }
        __syncthreads();          // every wave is done reading the |X| planes: conv1's output may now overwrite them
        SP_MARK(4);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = fmaxf(hi[f][r] + lo[f][r], 0.f);
            store_split4(smem + f * SP_FR128, SP_PL128, 4 * rt + q, i, y);
        }
    }
    __syncthreads();
    SP_MARK(5);

    // ---------------- phase 3: conv2 128->64, k3 s2 p1, ReLU: out frame o reads in frames 2 o - 1 .. 2 o + 1; wave = (16 channels, half of K)
    {
        const int rt = wave & 3, kh = wave >> 2;
        f32x4 hi[2], lo[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) { hi[o] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[o] = hi[o]; }
        const float *wq = P + OFF_Q2 + (rt * 4 + 2 * kh) * (3 * 3 * QF);
        bf16x8 a[3][3];                    // (a step is 6 - 12 MFMAs: the fragment stream runs two steps ahead, see phase 6)
        load_a3(a[0], SP_W(wq), lane);
        load_a3(a[1], SP_W(wq + 3 * QF), lane);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 b[4][3];
#pragma unroll
            for (int f = 0; f < 4; ++f) load_b3(b[f], smem + f * SP_FR128, SP_PL128, 2 * kh + kk, q, i);
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int s = kk * 3 + tap;
                if (s + 2 < 6) load_a3(a[(s + 2) % 3], SP_W(wq + (s + 2) * 3 * QF), lane);
                const bf16x8 (&ac)[3] = a[s % 3];
#define SP_TERM(AP, BP, ACC)                                                                  \
    _Pragma("unroll") for (int o = 0; o < 2; ++o) {                                           \
        const int fi = 2 * o + tap - 1;                                                       \
        if (fi >= 0 && !SP_SKIP(3)) ACC[o] = mfma_bf16(ac[AP], b[fi][BP], ACC[o]);            \
    }
                SP_TERM(2, 0, lo) SP_TERM(1, 1, lo) SP_TERM(0, 2, lo) SP_TERM(1, 0, lo) SP_TERM(0, 1, lo) SP_TERM(0, 0, hi)
#undef SP_TERM
            }
        }
        f32x4 s2[2] = {hi[0] + lo[0], hi[1] + lo[1]};
        float *exc = reinterpret_cast<float *>(smem + SP_EXC2);
        if (kh == 1) {
#pragma unroll
            for (int o = 0; o < 2; ++o) *reinterpret_cast<f32x4 *>(exc + ((rt * 2 + o) * 64 + lane) * 4) = s2[o];
        }
        __syncthreads();
        f32x4 other[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (kh == 0) {
#pragma unroll
            for (int o = 0; o < 2; ++o) other[o] = *reinterpret_cast<const f32x4 *>(exc + ((rt * 2 + o) * 64 + lane) * 4);
        }
        __syncthreads();              // the exchange sits inside the planes' region: every partial sum is read before the first plane store
        if (kh == 0) {
            const f32x4 bias = ldg4(P + OFF_B2 + 16 * rt + 4 * q);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = fmaxf(s2[o][r] + other[o][r] + bias[r], 0.f);
                store_split4(smem + SP_R1 + o * SP_FR2, SP_PL2, 4 * rt + q, i, y);
            }
        }
    }
    __syncthreads();
    SP_MARK(6);

    // ---------------- phase 4: conv3 64->64, k3 s2 p1, ReLU (one output frame; tap 0 reads padding): wave = (16 channels, tap 1 | 2)
    {
        const int rt = wave & 3, th = wave >> 2;      // tap th + 1 reads conv2's frame th
        f32x4 hi = {0.f, 0.f, 0.f, 0.f}, lo = hi;
        const float *wq = P + OFF_Q3 + (rt * 2 + th) * (2 * 3 * QF);
        bf16x8 a[2][3], b[2][3];
        load_a3(a[0], SP_W(wq), lane);
        load_a3(a[1], SP_W(wq + 3 * QF), lane);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) load_b3(b[kc], smem + SP_R1 + th * SP_FR2, SP_PL2, kc, q, i);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) if (!SP_SKIP(3)) mfma_split6(a[kc], b[kc], hi, lo);
        f32x4 s3 = hi + lo;
        float *exc = reinterpret_cast<float *>(smem + SP_EXC3);
        if (th == 1) *reinterpret_cast<f32x4 *>(exc + (rt * 64 + lane) * 4) = s3;
        __syncthreads();
        if (th == 0) {
            const f32x4 bias = ldg4(P + OFF_B3 + 16 * rt + 4 * q);
            const f32x4 other = *reinterpret_cast<const f32x4 *>(exc + (rt * 64 + lane) * 4);
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = fmaxf(s3[r] + other[r] + bias[r], 0.f);
            store_split4(smem + SP_C3, SP_PL3, 4 * rt + q, i, y);
        }
    }
    __syncthreads();
    SP_MARK(7);

    // ---------------- phase 5: conv4 64->128, k3 s1 p1, ReLU (one frame in / out: centre tap only)
    {
        const int rt = wave;
        f32x4 hi = ldg4(P + OFF_B4 + 16 * rt + 4 * q), lo = {0.f, 0.f, 0.f, 0.f};
        const float *wq = P + OFF_Q4 + rt * (2 * 3 * QF);
        bf16x8 a[2][3], b[2][3];
        load_a3(a[0], SP_W(wq), lane);
        load_a3(a[1], SP_W(wq + 3 * QF), lane);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) load_b3(b[kc], smem + SP_C3, SP_PL3, kc, q, i);
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) if (!SP_SKIP(3)) mfma_split6(a[kc], b[kc], hi, lo);
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = fmaxf(hi[r] + lo[r], 0.f);
        store_split4(smem + (sub + 1 < NSUB ? SP_STASH : SP_C4), SP_PL4, 4 * rt + q, i, y);      // the last tile's planes stay in R0
    }
    __syncthreads();
    SP_MARK(8);
    }      // sub

    asm volatile("" : "+v"(tid0));
    const int tid = tid0, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i = lane & 15;
    (void)tid;
    // ---------------- phase 6: LSTM input projection for the workgroup's NSUB tiles at once, gate-major (D rows = hidden units
    // 16 wave + 4 q + r, columns = clips): every W_ih fragment is loaded once and multiplies NSUB column tiles
    {
        f32x4 hi[NSUB][4], lo[NSUB][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bg = ldg4(P + OFF_BG + g * 128 + wave * 16 + 4 * q);
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) { hi[sb][g] = bg; lo[sb][g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        // A step here is six MFMAs (one gate): a fragment requested one step ahead has ~100 cycles of this wave's MFMAs (~400 with the SIMD's
        // other waves) to cross from L2 -- too little (halving the W_ih fragment bytes changed nothing, serving them from L1 did): the
        // stream runs THREE steps ahead on four rotating register sets.
        const float *wq = P + OFF_QIH + wave * (4 * 4 * 3 * QF);
        constexpr int AHEAD = 3;
        bf16x8 a[AHEAD + 1][3];
#pragma unroll
        for (int s0 = 0; s0 < AHEAD; ++s0) load_a3(a[s0], SP_W(wq + (SP_SKIP(11) ? 0 : s0) * 3 * QF), lane);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            bf16x8 b[NSUB][3];
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) load_b3(b[sb], smem + (sb + 1 < NSUB ? SP_STASH : SP_C4), SP_PL4, kc, q, i);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int s = kc * 4 + g;
                if (s + AHEAD < 16 && !SP_SKIP(11)) load_a3(a[(s + AHEAD) % (AHEAD + 1)], SP_W(wq + (s + AHEAD) * 3 * QF), lane);
                const bf16x8 (&ac)[3] = a[s % (AHEAD + 1)];
                if (!SP_SKIP(6) && !SP_SKIP(11)) {
#define SP_TERM(AP, BP, ACC) _Pragma("unroll") for (int sb = 0; sb < NSUB; ++sb) ACC[sb][g] = mfma_bf16(ac[AP], b[sb][BP], ACC[sb][g]);
                    SP_TERM(2, 0, lo) SP_TERM(1, 1, lo) SP_TERM(0, 2, lo) SP_TERM(1, 0, lo) SP_TERM(0, 1, lo) SP_TERM(0, 0, hi)
#undef SP_TERM
                }
            }
        }
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) {
            const long long tile_raw = ((long long)blockIdx.x * HALVES + half) * NSUB + sb;
            const int tile_id = (int)(tile_raw < ntile ? tile_raw : ntile - 1);
            float *dst = gx + ((size_t)(tile_id / G) * Gws + g0 + tile_id % G) * GX_TILE_FLOATS + (size_t)wave * 4 * 256 + lane * 4;
            if (tile_raw < ntile)
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(dst + g * 256) = hi[sb][g] + lo[sb][g];
        }
    }
    SP_MARK(9);
    SP_CLK();
}

// ---- persistent LSTM on split products ---------------------------------------------------------
// Same decomposition as silero_lstm_kernel (silero.hip): one persistent workgroup per 16 clips, wave w owns hidden units 16 w .. 16 w + 15
// of all four gates, W_hh resident for the whole clip, h exchanged through double-buffered LDS, one barrier per step.  W_hh x h runs as
// bf16 x 3 split products: 96 v_mfma_f32_16x16x32_bf16 per wave and step instead of 128 v_mfma_f32_16x16x4_f32 (6/16 of the matrix
// time).  W_hh's planes 0 and 1 stay in VGPRs (128 registers, as many as the f32 rows took); plane 2 -- used by one product in six -- sits in
// LDS in fragment order (128 KB, one copy per CU: a wave reads its 1 KiB fragments back with ds_read_b128).  h is split by the lanes that
// produce it (four consecutive units of one clip = one 8-byte store per plane).
constexpr int LS_H = 0;                          // h planes [2 buffers][3 planes][16 k-groups][16 clips][8 bf16] = 2 x 12 288 B
constexpr int LS_HPL = 4096, LS_HBUF = 12288;
constexpr int LS_PART = 2 * LS_HBUF;             // f32 [2][8 waves][16 clips]
constexpr int LS_W2 = LS_PART + 1024;            // W_hh plane 2: [8 waves][4 gates][4 chunks][64 lanes][16 B] = 131 072 B
constexpr int LS_BYTES = LS_W2 + 131072;
static_assert(LS_BYTES <= 160 * 1024, "split LSTM LDS map");

__global__ __launch_bounds__(512, 2) void silero_lstm_split_kernel(
    const float *__restrict__ P, const float *__restrict__ gx, const float *__restrict__ state0,
    int B, int G, int T, float *__restrict__ probs, long long probs_stride, float *__restrict__ state_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *part = reinterpret_cast<float *>(smem + LS_PART);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int grp = blockIdx.x;
    const long long b = (long long)grp * 16 + n;
    const bool bvalid = b < B;
    const int u0 = wave * 16 + 4 * q;             // this lane's 4 hidden units

    // W_hh fragments of this wave's 4 gate tiles: planes 0, 1 -> 128 VGPRs for the whole clip, plane 2 -> LDS
    bf16x8 a[4][4][2];
    {
        const float *wq = P + OFF_QHH + (size_t)wave * (4 * 4 * 3 * QF);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const float *f3 = wq + (g * 4 + kc) * 3 * QF;
                a[g][kc][0] = ldq(f3, lane);
                a[g][kc][1] = ldq(f3 + QF, lane);
                *reinterpret_cast<bf16x8 *>(smem + LS_W2 + (((wave * 4 + g) * 4 + kc) * 64 + lane) * 16) = ldq(f3 + 2 * QF, lane);
            }
    }
    const f32x4 dw = ldg4(P + OFF_DW + u0);
    const float db = P[OFF_DB];

    f32x4 c = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
    if (state0 != nullptr && bvalid) {
        h = *reinterpret_cast<const f32x4 *>(state0 + b * 128 + u0);
        c = *reinterpret_cast<const f32x4 *>(state0 + ((long long)B + b) * 128 + u0);
    }
    store_split4(smem + LS_H, LS_HPL, 4 * wave + q, n, h);
    __syncthreads();

    const float *gsrc = gx + (size_t)grp * GX_TILE_FLOATS + (size_t)wave * 4 * 256 + lane * 4;
    const size_t gstep = (size_t)G * GX_TILE_FLOATS;
    f32x4 gcur[4], gnxt[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) gcur[g] = *reinterpret_cast<const f32x4 *>(gsrc + g * 256);

    const unsigned char *w2 = smem + LS_W2 + (size_t)(wave * 16 * 64 + lane) * 16;
    int cur = 0;
    for (int t = 0; t < T; ++t) {
        const int tn = (t + 1 < T) ? t + 1 : t;
#pragma unroll
        for (int g = 0; g < 4; ++g) gnxt[g] = *reinterpret_cast<const f32x4 *>(gsrc + tn * gstep + g * 256);

        f32x4 hi[4], lo[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { hi[g] = gcur[g]; lo[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const unsigned char *hb = smem + LS_H + cur * LS_HBUF;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            bf16x8 bb[3];
            load_b3(bb, hb, LS_HPL, kc, q, n);
            bf16x8 a2[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) a2[g] = *reinterpret_cast<const bf16x8 *>(w2 + (g * 4 + kc) * 64 * 16);
            // the six products, gates innermost so that consecutive MFMAs hit different accumulators
#pragma unroll
            for (int g = 0; g < 4; ++g) lo[g] = mfma_bf16(a2[g], bb[0], lo[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) lo[g] = mfma_bf16(a[g][kc][1], bb[1], lo[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) lo[g] = mfma_bf16(a[g][kc][0], bb[2], lo[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) lo[g] = mfma_bf16(a[g][kc][1], bb[0], lo[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) lo[g] = mfma_bf16(a[g][kc][0], bb[1], lo[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) hi[g] = mfma_bf16(a[g][kc][0], bb[0], hi[g]);
        }
        float dpart = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = gate_sigmoid(hi[0][r] + lo[0][r]), fg = gate_sigmoid(hi[1][r] + lo[1][r]);
            const float gg = gate_tanh(hi[2][r] + lo[2][r]), og = gate_sigmoid(hi[3][r] + lo[3][r]);
            c[r] = fg * c[r] + ig * gg;
            h[r] = og * gate_tanh(c[r]);
            dpart = fmaf(dw[r], fmaxf(h[r], 0.f), dpart);
        }
        const int nxt = cur ^ 1;
        store_split4(smem + LS_H + nxt * LS_HBUF, LS_HPL, 4 * wave + q, n, h);
        dpart += __shfl_xor(dpart, 16);
        dpart += __shfl_xor(dpart, 32);
        if (q == 0) part[(nxt * 8 + wave) * 16 + n] = dpart;
        __syncthreads();
        if (wave == 0 && lane < 16 && bvalid) {
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

int silero_lstm_split_launch(const float *packed, const float *gx, const float *state0, int batch, int G, int steps, float *probs,
                             long long probs_stride, float *state_n, void *stream) {
    VADX_DYN_LDS(silero_lstm_split_kernel, LS_BYTES);
    hipLaunchKernelGGL(silero_lstm_split_kernel, dim3(G), dim3(512), LS_BYTES, static_cast<hipStream_t>(stream), packed, gx, state0, batch, G,
                       steps, probs, probs_stride, state_n);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

// VADX_SPLIT_HALVES: 1 = 512-thread workgroups (two independent ones per CU); 2 = the lockstep pair.  Measured on one box, B = 4096 x 10 s
// (one tile per workgroup half): 5.75 ms against 6.35 ms -- the L1 hits of the pair do not pay for what the lockstep costs: two
// independent workgroups sit in DIFFERENT phases most of the time (one in its f32 STFT or its staging while the other streams
// weights), the pair never does.
// VADX_SPLIT_NSUB: tiles a workgroup encodes one after the other before ONE joint W_ih phase (2: every W_ih fragment serves 32 columns,
// at the price of the first tile's conv4 planes in a 12 KB stash and one more barrier in conv2).
#ifndef VADX_SPLIT_HALVES
#define VADX_SPLIT_HALVES 1
#endif
#ifndef VADX_SPLIT_NSUB
#define VADX_SPLIT_NSUB 1       // 2 measured no faster on the same box (5.61 against 5.59 ms): the W_ih phase waits on fragment LATENCY, not on L2 bandwidth
#endif
template <typename S>
int silero_encode_split_launch(const float *packed, const S *src, float in_scale, long long n_valid, long long row_stride,
                               long long origin, int batch, int G, int steps, int Gws, int first_group, float *gx, void *stream) {
    constexpr int HV = VADX_SPLIT_HALVES, NS = VADX_SPLIT_NSUB;
    VADX_DYN_LDS((silero_encode_split_kernel<S, HV, NS>), SP_LDS_BYTES * HV);
    const long long nblk = ((long long)G * steps + HV * NS - 1) / (HV * NS);
    hipLaunchKernelGGL((silero_encode_split_kernel<S, HV, NS>), dim3((unsigned)nblk), dim3(SP_THREADS * HV), SP_LDS_BYTES * HV, static_cast<hipStream_t>(stream),
                       packed, src, in_scale, n_valid, row_stride, origin, batch, G, steps, Gws, first_group, gx);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
template int silero_encode_split_launch<float>(const float *, const float *, float, long long, long long, long long, int, int, int, int, int, float *, void *);
template int silero_encode_split_launch<int16_t>(const float *, const int16_t *, float, long long, long long, long long, int, int, int, int, int, float *, void *);

}  // namespace silero
}  // namespace vadx
