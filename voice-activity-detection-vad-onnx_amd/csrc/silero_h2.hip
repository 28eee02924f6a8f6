// silero_h2.hip -- Silero-VAD v5 (16 kHz) encoder + recurrent kernels on fp16 x 2 split products (csrc/split2.h), gfx950.
//
// Same inputs, same packed blob (its OFF_H* sections) and the same gx layout as silero_encode_split_kernel / silero_encode_kernel; what
// changes against the bf16 x 3 kernel (csrc/silero_split.hip):
//   * every product -- the folded STFT included -- is THREE v_mfma_f32_16x16x32_f16 per K = 32 step (h0 h0 -> hi; h0 h1', h1' h0 -> mid,
//     joined as hi + 2^-11 mid) instead of six bf16 MFMAs: 2 160 fp16 MFMAs per 16-window tile (STFT 480, conv1 960, conv2 240, conv3 48,
//     conv4 48, W_ih 384) against 1 024 f32 + 3 360 bf16; a weight is 4 bytes of fragments instead of 6, an activation two LDS planes
//     instead of three;
//   * the STFT's operands e = x[n] + x[256 - n], o = x[n] - x[256 - n] are formed ONCE per sample pair (thread = clip x four pairs x parity
//     class, all four frames), split, and parked as B planes for all four frames (64 KB, in place of the staged window); wave (bin tile,
//     frame pair) then runs both classes and both parts of its two frames, so nothing has to meet through LDS; bin 64 (its own mirror) is
//     row 0 of a fifth bin tile whose (class, part) pieces are dealt over the eight waves -- no VALU dot product;
//   * a workgroup encodes NSUB = 2 tiles one after the other up to conv2 and runs conv3, conv4 and W_ih ONCE for both: those fragments
//     (328 of a tile's 750 KB) serve 32 columns instead of 16;
//   * the kernels keep the running max |x| of everything they split; a workgroup that saw a value outside the fp16 range raises the
//     blob's sticky flag (OFF_HFLAG + 1) and the host recomputes the batch on the bf16 x 3 kernels (vadx_silero_range_flag).
// Reference being reproduced: the `session.run` of Silero/modeling_modified/utils_vad.py:116-119 (see silero.hip).
#include "silero_common.h"
#include "split2.h"

// VADX_EXP: development-only what-if switches for tools/exp_encoder.py (results are wrong when set): bit 3 no conv2..4 MFMAs, 4 no STFT
// MFMAs, 5 no conv1 MFMAs, 6 no W_ih MFMAs, 13 every weight fragment from one address (L1 instead of L2), 14 per-phase cycle accounting
// of wave 0 (h2_dbg, read with vadx_silero_h2_debug_cycles)
#ifndef VADX_EXP
#define VADX_EXP 0
#endif
#define H2_SKIP(n) ((VADX_EXP >> (n)) & 1)
// H2_XP: which phases request their first weight fragments BEFORE the barrier that ends the phase in front of them (bit 0 STFT, 1 conv1,
// 2 conv2, 3 conv4 / W_ih), bit 4: conv2 requests its whole stream (six sets) up front
#ifndef H2_XP
#define H2_XP 1
#endif
#define H2_XP_ON(n) ((H2_XP >> (n)) & 1)
// Every phase re-derives its thread indices from a laundered copy of the thread id: nothing computed from them (LDS offsets, fragment
// pointers of LATER phases) can then be hoisted to the top of the tile loop, where it would sit in registers -- or in scratch -- through
// every phase in between.
#define H2_IDS()                                                                                           \
    int t_ = tid0;                                                                                         \
    asm volatile("" : "+v"(t_));                                                                           \
    const int tid = t_, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i = lane & 15;                    \
    (void)tid; (void)lane; (void)wave; (void)q; (void)i
// how many steps ahead of the MFMAs each phase's fragment stream runs (register sets = steps + 1)
#ifndef H2_AHS
#define H2_AHS 2      // STFT
#endif
#ifndef H2_AH1
#define H2_AH1 2      // conv1
#endif
#ifndef H2_AHI
#define H2_AHI 2      // W_ih (two tiles per workgroup)
#endif
#define H2_W(addr) (H2_SKIP(13) ? (P + vadx::silero::OFF_H1) : (addr))
// bit 15: the SECOND tile of a workgroup takes its conv1 / conv2 fragments from one L1-resident address (bit 12: its STFT fragments too): an upper
// bound on what sharing those weight streams between the two tiles of a workgroup (one pass over 128 columns) could save
#define H2_W12(addr) ((((H2_SKIP(15) || H2_SKIP(11)) && sub == 1) || H2_SKIP(13)) ? (P + vadx::silero::OFF_H1) : (addr))
// bit 11: ... and every lane reads the SAME 16 bytes (one L1 access per load instead of sixteen): what removing those loads altogether -- the
// second tile multiplying the fragments the first tile's pass already holds -- could save if L1 request throughput is the limit
#define H2_L12 (((H2_SKIP(11) && sub == 1) || H2_SKIP(10)) ? 0 : lane)
// bit 10: EVERY fragment load reads one 16-byte piece (all lanes the same address): the kernel without its L1 request stream
#define H2_LN (H2_SKIP(10) ? 0 : lane)
// bit 9: the W_ih fragments alone as single L1 accesses from one address (what a W_ih pass over more columns per fragment could approach);
// bit 8: the STFT's
#define H2_WIH(addr) ((H2_SKIP(9) || H2_SKIP(13)) ? (P + vadx::silero::OFF_H1) : (addr))
#define H2_LIH ((H2_SKIP(9) || H2_SKIP(10)) ? 0 : lane)
#define H2_WST(addr) ((H2_SKIP(8) || (H2_SKIP(12) && sub == 1) || H2_SKIP(13)) ? (P + vadx::silero::OFF_H1) : (addr))
#define H2_LST ((H2_SKIP(8) || H2_SKIP(10)) ? 0 : lane)
#define H2_WS(addr) (((H2_SKIP(12) && sub == 1) || H2_SKIP(13)) ? (P + vadx::silero::OFF_H1) : (addr))
#if (VADX_EXP >> 14) & 1
__device__ unsigned long long h2_dbg[16];
#define H2_T0() long long h2_t_ = __builtin_readcyclecounter(); const long long h2_c0_ = h2_t_, h2_w0_ = wall_clock64()
#define H2_CLK() do { if (threadIdx.x == 0) { atomicAdd(&h2_dbg[14], (unsigned long long)(__builtin_readcyclecounter() - h2_c0_)); atomicAdd(&h2_dbg[15], (unsigned long long)(wall_clock64() - h2_w0_)); } } while (0)
#define H2_MARK(slot) do { if (threadIdx.x == 0) { const long long n_ = __builtin_readcyclecounter(); atomicAdd(&h2_dbg[slot], (unsigned long long)(n_ - h2_t_)); h2_t_ = n_; } } while (0)
extern "C" int vadx_silero_h2_debug_cycles(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(h2_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(h2_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define H2_T0() do {} while (0)
#define H2_MARK(slot) do {} while (0)
#define H2_CLK() do {} while (0)
#endif

// H2_DUMP (debugging): per (tile, stage) an order-independent checksum of the LDS region the stage just produced, into a caller's buffer
// (vadx_silero_h2_dump): stage 0 operand planes, 1 |X| planes + scratch, 2 conv1 planes, 3 conv2 planes
#ifndef H2_DUMP
#define H2_DUMP 0
#endif
#ifndef H2_PK_NATURAL
#define H2_PK_NATURAL 0
#endif
// H2_TRACE (development, tools/h2_trace.py): sixteen workgroups spread over the grid record the shader clock of lane 0 of every wave right before
// and right after each barrier (H2_SYNC(k): marks 2 k, 2 k + 1; the per-tile barriers at + 32 per tile of the workgroup) -- a timeline of
// where the waves of a workgroup wait, with a few stores per phase as the only perturbation
#ifndef H2_TRACE
#define H2_TRACE 0
#endif
#ifndef H2_PAIR_T
#define H2_PAIR_T 1
#endif
#if H2_TRACE
__device__ unsigned long long h2_trace_buf[16][8][128];
extern "C" int vadx_silero_h2_trace(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(h2_trace_buf), sizeof(unsigned long long) * 16 * 8 * 128) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[16 * 8 * 128]; if (hipMemcpyToSymbol(HIP_SYMBOL(h2_trace_buf), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define H2_TR(m) do { if (h2_tr_slot >= 0 && (threadIdx.x & 63) == 0) h2_trace_buf[h2_tr_slot][threadIdx.x >> 6][(m) + h2_tr_off] = __builtin_readcyclecounter(); } while (0)
#define H2_SYNC(k) do { H2_TR(2 * (k)); __syncthreads(); H2_TR(2 * (k) + 1); } while (0)
#else
#define H2_TR(m) do {} while (0)
#define H2_SYNC(k) __syncthreads()
#endif
// H2_PRIO: wave priority (s_setprio) inside the GEMM loops, 0 elsewhere: the SIMD's arbiter then prefers the waves that feed the matrix pipe over
// co-resident waves in their VALU phases
#ifndef H2_PRIO
#define H2_PRIO 0
#endif

#define H2_PRIO_ON() do { if (H2_PRIO) __builtin_amdgcn_s_setprio(H2_PRIO); } while (0)
#define H2_PRIO_OFF() do { if (H2_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)
// (round 6, measured at config 2, built then removed: H2_PRIO = 1 -- 3.59 -> 3.85 ms, the waves in their VALU phases are starved and reach the
//  barriers later; the second tile's samples requested at the start of the first tile's conv2 phase instead of in its own staging phase --
//  3.59 -> 4.35 ms: the 24 registers do not exist beside conv2's, scratch 36 -> 148 B; a persistent grid of 512 workgroups walking the tile
//  pairs, the upper half started 0 / 24 000 / 48 000 / 72 000 shader cycles late so that a CU's two workgroups run a fixed fraction of a tile
//  period apart -- 3.55 -> 3.60 / 3.60 / 3.60 / 3.61 ms: no offset beats the dispatcher's own staggering.)
#if H2_DUMP
__device__ unsigned *h2_dump_ptr;
extern "C" int vadx_silero_h2_dump(unsigned *buf) { return hipMemcpyToSymbol(HIP_SYMBOL(h2_dump_ptr), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#define H2_SUM(stage, base, bytes) do { unsigned acc_ = 0; for (int o_ = threadIdx.x * 4; o_ < (bytes); o_ += 512 * 4) acc_ += *reinterpret_cast<const unsigned *>(smem + (base) + o_) * (unsigned)(2 * o_ + 1); \
    atomicAdd(h2_dump_ptr + (size_t)tile_id * 4 + (stage), acc_); } while (0)
#else
#define H2_SUM(stage, base, bytes) do {} while (0)
#endif

namespace vadx {
namespace silero {

constexpr int H2_THREADS = 512;
// ---- LDS map (BYTES): 81 920 B per workgroup => two workgroups per CU (eight waves of <= 128 VGPRs each)
//   R0 [0, 65536): X f32 [16 clips][642]
//                  -> STFT operand planes [E|O][e|o][2 planes][8 k-groups][4 frames x 16 clips][8 fp16]                (65 536 B)
//                  -> |X| planes [2][4 frames][16 k-groups][16 clips][8 fp16] at 0 (32 768 B) -> conv1 output planes (same shape, in place)
//                  -> after the last tile's conv2: conv3 output planes [tile][2][8][16][8] at 0, conv4 output planes [tile][2][16][16][8] at 8192
//      scratch (only while [32768, 65536) is free, i.e. from the STFT's last barrier on): Nyquist magnitudes f32 [4 frames][16 clips],
//              bin-64 partial sums f32 [8 waves][2 frames][16 clips]; conv2's K-half exchange (8 KB), conv3's (8 KB)
//   R1 [65536, 81920): conv2 output planes of the workgroup's tiles [tile][2 planes][2 frames][8 k-groups][16][8]
constexpr int H2_EO_PL = 8192, H2_EO_KG = 1024;                 // one (class, e|o, plane) block; one k-group row of 64 columns
constexpr int H2_PL128 = 16384, H2_FR128 = 4096;
constexpr int H2_SCR = 32768;                                   // f32 scratch [512]: nyq at +0, bin-64 partials at +256 floats
constexpr int H2_EXC2 = 36864, H2_EXC3 = 45056;
#ifndef H2_R1_BASE
#define H2_R1_BASE 65536      // debugging: 53248 keeps every LDS address below 64 KB (NSUB = 1 only)
#endif
constexpr int H2_R1 = H2_R1_BASE, H2_T2 = 8192, H2_PL2 = 4096, H2_FR2 = 2048;
constexpr int H2_C3 = 0, H2_T3 = 4096, H2_PL3 = 2048;
constexpr int H2_C4 = 8192, H2_T4 = 8192, H2_PL4 = 4096;
constexpr int H2_LDS_BYTES = H2_R1 + 2 * H2_T2;
static_assert(16 * X_LDM * 4 <= 65536 && 2 * H2_LDS_BYTES <= 160 * 1024 && H2_C4 + 2 * H2_T4 <= H2_SCR, "fp16 x 2 encoder LDS map");

__device__ __forceinline__ int hpl_off(int kg8, int clip) { return (kg8 * 16 + clip) * 16; }

// a producer lane's four consecutive channels 4 g .. 4 g + 3 of clip i: one 8-byte store into each of the two planes
__device__ __forceinline__ void store_h4(unsigned char *base, int plane_stride, int g, int i, const f32x4 v, float &amax) {
    u32x2 p0, p1;
    split2x4(v, p0, p1, amax);
    unsigned char *d = base + hpl_off(g >> 1, i) + (g & 1) * 8;
    *reinterpret_cast<u32x2 *>(d) = p0;
    *reinterpret_cast<u32x2 *>(d + plane_stride) = p1;
}
__device__ __forceinline__ void store_h1(unsigned char *base, int plane_stride, int slot, int i, float v, float &amax) {
    unsigned short h0, h1;
    split2x1(v, h0, h1, amax);
    unsigned char *d = base + hpl_off(slot >> 3, i) + (slot & 7) * 2;
    *reinterpret_cast<unsigned short *>(d) = h0;
    *reinterpret_cast<unsigned short *>(d + plane_stride) = h1;
}
// the wave's B fragments (two planes) of the 32-k chunk kc: lane 16 q + i reads k-group 4 kc + q of clip i
__device__ __forceinline__ void load_b2(f16x8 (&b)[2], const unsigned char *base, int plane_stride, int kc, int q, int i) {
    const unsigned char *s = base + hpl_off(4 * kc + q, i);
    b[0] = *reinterpret_cast<const f16x8 *>(s);
    b[1] = *reinterpret_cast<const f16x8 *>(s + plane_stride);
}
__device__ __forceinline__ void load_a2(f16x8 (&a)[2], const float *frag2, int lane) {
    a[0] = ldh(frag2, H2_LN);
    a[1] = ldh(frag2 + HF, lane);
}

template <typename SampleT, int NSUB>
#ifndef H2_WAVES_PER_SIMD
#define H2_WAVES_PER_SIMD 4
#endif
#ifndef H2_LDS_PAD
#define H2_LDS_PAD 0          // debugging: extra dynamic LDS per workgroup (> 0 forces one workgroup per CU)
#endif
__global__ __launch_bounds__(H2_THREADS, H2_WAVES_PER_SIMD) void silero_encode_h2_kernel(
    const float *__restrict__ P, const SampleT *__restrict__ audio, float in_scale, long long n_samples,
    long long row_stride, long long origin, int B, int G, int T, int Gws, int g0, float *__restrict__ gx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *X = reinterpret_cast<float *>(smem);
    float *scr = reinterpret_cast<float *>(smem + H2_SCR), *nyq = scr, *b64p = scr + 256;

    int tid0 = threadIdx.x;
    const long long ntile = (long long)G * T;
#if H2_TRACE
    const int h2_tr_slot = (blockIdx.x % 2503u == 1201u && blockIdx.x / 2503u < 16u) ? (int)(blockIdx.x / 2503u) : -1;
    int h2_tr_off = 0;
    H2_TR(126);
#endif
    float amax = 0.f;                   // running max |x| of everything this thread splits
    if (ldg1(P + OFF_HFLAG) == 0.f) {   // uniform: this blob cannot run on fp16 x 2 (basis without the fold, a weight outside the range)
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(reinterpret_cast<unsigned *>(const_cast<float *>(P)) + OFF_HFLAG + 1, 2u);
        return;
    }
    H2_T0();
    const long long blk = blockIdx.x;
    {
    // Cross-phase fragment prefetch: the first sets of a phase's weight stream are requested BEFORE the barrier that ends the phase in front of
    // it (global loads stay in flight across s_barrier, which only waits for lgkmcnt), so the L2 round trip that used to open every phase
    // runs under the previous phase's epilogue.  pre_* = those sets, named per consumer.
#pragma unroll 1
    for (int sub = 0; sub < NSUB; ++sub) {
    // per tile: nothing derived from the thread index is hoisted out of the tile loop (see silero_split.hip)
    // the workgroup's tiles: an odd tile count leaves the last workgroup's last slot without work -- it recomputes the last tile (the
    // barriers are workgroup-wide) and stores nothing
    // a workgroup's NSUB tiles are CONSECUTIVE WINDOWS of one clip group (H2_PAIR_T: window t = NSUB * (blk / G) + sub of group blk % G): the
    // second tile's samples are the next 2 KB of the same sixteen rows -- same pages, same DRAM rows as the loads the first tile has just made --
    // instead of sixteen rows 10 MB away (adjacent groups of one window, the round-5 order)
#if H2_PAIR_T
    const int grp = (int)(blk % G), t_raw = (int)(blk / G) * NSUB + sub;
    const bool tile_valid = t_raw < T;
    const int t = tile_valid ? t_raw : T - 1;
    const int tile_id = t * G + grp;
    (void)tile_id;
#else
    const long long tile_raw = blk * NSUB + sub;
    const int tile_id = (int)(tile_raw < ntile ? tile_raw : ntile - 1);
    const int grp = tile_id % G, t = tile_id / G;
#endif
#if H2_TRACE
    h2_tr_off = 32 * sub;
#endif

    // ---------------- phase 0: the 16 windows (576 samples each) + right reflect pad of 64, even / odd samples in separate planes of the
    // clip row (as silero_encode_kernel stages them for the folded pass)
    auto xslot = [](int pp) { return (pp & 1) * X_ODD + (pp >> 1); };
    {
        H2_IDS();
        const long long base = (long long)t * 512 + origin;
        const bool vec_ok = ((row_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(audio) & (SampleIO<SampleT>::VEC_ALIGN - 1)) == 0) && n_samples >= 4;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const bool fast = vec_ok && base >= 0 && base + 576 <= n_samples && (long long)grp * 16 + 16 <= B;
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
                    }
                }
                // right reflect pad (0, 64): padded sample 1150 - p = sample p for p = 511 .. 574.  p = 512 + 4 lane + jj sits in the third load of
                // lanes 0 .. 15 (slot parity = parity of jj, plane index 319 - 2 lane - ...), p = 511 in the second load of lane 63 -- spelled out:
                // the generic per-element test cost ~40 VALU instructions per clip for sixteen lanes' worth of stores
                if (lane < 16) {
                    const f32x4 v = xv[k2][2];
                    row[319 - 2 * lane] = v[0];
                    row[X_ODD + 318 - 2 * lane] = v[1];
                    row[318 - 2 * lane] = v[2];
                    if (lane < 15) row[X_ODD + 317 - 2 * lane] = v[3];
                }
                if (lane == 63) row[X_ODD + 319] = xv[k2][1][3];
            }
        } else {        // edge windows, short clips, groups past the batch: clamped unconditional loads, patched per element
            f32x4 x4[5];
            if (vec_ok) {
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    const int e = min(tid + H2_THREADS * it, 16 * 144 - 1), c = e / 144, p = 4 * (e - c * 144);
                    const long long b = (long long)grp * 16 + c, idx = base + p;
                    const SampleT *src = audio + (b < B ? b : 0) * row_stride;
                    const long long idc = idx < 0 ? 0 : (idx + 3 < n_samples ? idx : ((n_samples - 4) & ~3LL));
                    x4[it] = SampleIO<SampleT>::load4(src + idc, in_scale);
                }
            }
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int e = tid + H2_THREADS * it;             // 16 clips x 144 float4
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
    f16x8 pre_s[H2_AHS][2];           // the STFT's first sets: in flight while the samples go to registers and the operand planes are built
    if (H2_XP_ON(0)) {
        H2_IDS();
        const float *wq = P + OFF_HSF + (size_t)(wave & 3) * (2 * 2 * 2 * 2 * HF);
#pragma unroll
        for (int s0_ = 0; s0_ < H2_AHS; ++s0_) load_a2(pre_s[s0_], H2_WST(wq + s0_ * 2 * HF), H2_LST);
    }
    H2_SYNC(0);
    H2_MARK(0);

    // ---------------- phase 1: the folded STFT on split products -> magnitudes -> the two fp16 planes of conv1's input.
    // Per frame f (samples 128 f .. 128 f + 255 of the padded window) and pair index n = 1..128: e = x[n] + x[256 - n] (cos part),
    // o = the difference (sin part), in two classes (E: n = 2 m + 2, O: n = 2 m + 1: the frequency fold); bins k <= 63 of tile tl:
    //   X[k] = E + O, X[128 - k] = +-(E - O), with E / O = the class's partial sums (silero_common.h: stft_fold_class).
    // Input-channel slot s of conv1: s <= 64 = bin s, s = 64 + k = bin 128 - k; bin 128 (Nyquist) goes to the scratch.
    f16x8 pre_1[2][2], pre_2[2][2];
    {
        H2_IDS();
        const int tl = wave & 3, fp = wave >> 2;                  // GEMM role: bins 16 tl + 4 q + r (and 128 - them), frames 2 fp, 2 fp + 1, clip i
        f32x4 mk[2], mn[2];
        float b64[2];
        {
            // ---- operand role: thread (clip pc, four consecutive pairs 4 pj .. 4 pj + 3 of class cls) reads its 32 samples out of X
            const int cls = wave >> 2, pc = tid & 15, pj = (tid >> 4) & 15;
            float xa[4][4], xb[4][4];                             // [frame][k]: the pair's two samples
            {
                const float *row = X + pc * X_LDM + (cls ? X_ODD : 1) + 4 * pj, *rowb = X + pc * X_LDM + (cls ? X_ODD : 0) + 127 - 4 * pj;
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { xa[f][k] = row[64 * f + k]; xb[f][k] = rowb[64 * f - k]; }
            }
            float x0p[2];                                         // the n = 0 sample of the GEMM role's two frames (even class)
#pragma unroll
            for (int fr = 0; fr < 2; ++fr) x0p[fr] = X[i * X_LDM + 64 * (2 * fp + fr)];
            // bin 64's piece of this wave: (class bc, part bp) of the fifth tile for the wave's two frames; the n = 0 tap rides in the
            // even class's accumulator (row 0 = lanes q = 0, element 0)
            const int bc = (wave >> 1) & 1, bp = wave & 1;
#if H2_DUMP
            {   // stage 0: the samples this thread holds, weighted by who holds them
                unsigned acc_ = 0;
                for (int f = 0; f < 4; ++f) for (int k = 0; k < 4; ++k) acc_ += (__float_as_uint(xa[f][k]) * 3u + __float_as_uint(xb[f][k])) * (unsigned)(2 * (tid * 16 + f * 4 + k) + 1);
                atomicAdd(h2_dump_ptr + (size_t)tile_id * 4 + 0, acc_);
            }
#endif
            H2_SYNC(1);          // every sample is in registers: the operand planes may overwrite X
            H2_MARK(1);
#if H2_DUMP
            unsigned accr_ = 0, acce_ = 0;
#endif
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                f32x4 ev, ov;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // (plain v_add_f32 / v_sub_f32, spelled out: left to the compiler the pair sums become v_pk_add_f32 with a CROSS op_sel swizzle
                    //  -- xb is read in descending order -- and that form returned wrong sums in roughly every second tile once two workgroups
                    //  (four waves per SIMD) shared a CU, while the same binary was bit-exact at one workgroup per CU: tests/probes/h2_race.py,
                    //  DESIGN.md section 4e)
                    float e_, o_;
#if H2_PK_NATURAL     // development only (tests/probes/pk_hazard.py): the plain-C sums the compiler turns into cross-swizzled v_pk_add_f32
                    e_ = xa[f][k] + xb[f][k];
                    o_ = xa[f][k] - xb[f][k];
#if H2_PK_NATURAL == 2      // ... and nothing may follow the sums for a few cycles
                    asm volatile("s_nop 7" : "+v"(e_), "+v"(o_));
#endif
#else
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(e_) : "v"(xa[f][k]), "v"(xb[f][k]));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(o_) : "v"(xa[f][k]), "v"(xb[f][k]));
#endif
                    ev[k] = e_;
                    ov[k] = o_;
                }
                unsigned char *d = smem + (cls * 4) * H2_EO_PL + (pj >> 1) * H2_EO_KG + (16 * f + pc) * 16 + 8 * (pj & 1);
                u32x2 p0, p1;
                split2x4(ev, p0, p1, amax);
                *reinterpret_cast<u32x2 *>(d) = p0;
                *reinterpret_cast<u32x2 *>(d + H2_EO_PL) = p1;
                split2x4(ov, p0, p1, amax);
                *reinterpret_cast<u32x2 *>(d + 2 * H2_EO_PL) = p0;
                *reinterpret_cast<u32x2 *>(d + 3 * H2_EO_PL) = p1;
#if H2_PK_NATURAL == 3       // ... and the samples stay in their registers until the frame's planes are stored (no early reuse of a source register)
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("" :: "v"(xa[f][k]), "v"(xb[f][k]));
#endif
#if H2_DUMP
                for (int k = 0; k < 4; ++k) acce_ += (__float_as_uint(ev[k]) * 5u + __float_as_uint(ov[k])) * (unsigned)(2 * (tid * 16 + f * 4 + k) + 1);
                accr_ += (p0[0] * 3u + p0[1] * 7u + p1[0] * 11u + p1[1] * 13u) * (unsigned)(2 * (tid * 4 + f) + 1);
#endif
            }
#if H2_DUMP
            atomicAdd(h2_dump_ptr + (size_t)tile_id * 4 + 2, acce_);
            atomicAdd(h2_dump_ptr + (size_t)tile_id * 4 + 3, accr_);
#endif
            H2_SYNC(2);
            H2_MARK(2);
            H2_SUM(1, 0, 65536);
            // ---- the wave's GEMM: (class, part) = (E re, E im, O re, O im) x two chunks x its two frames
            const f32x4 c0 = ldg4(P + OFF_S0 + tl * 16 + 4 * q), s0 = ldg4(P + OFF_S0 + 64 + tl * 16 + 4 * q);
            const float b64n0 = bc ? 0.f : ldg1(P + OFF_B64 + 256 + bp);
            f32x4 hi[4][2], mid[4][2];
#pragma unroll
            for (int fr = 0; fr < 2; ++fr) {
                hi[0][fr] = c0 * x0p[fr];
                hi[1][fr] = s0 * x0p[fr];
                hi[2][fr] = f32x4{0.f, 0.f, 0.f, 0.f};
                hi[3][fr] = hi[2][fr];
#pragma unroll
                for (int a4 = 0; a4 < 4; ++a4) mid[a4][fr] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const float *wb = P + OFF_HSF + (size_t)((4 * 2 + bc) * 2 + bp) * (2 * 2 * HF);
            if (!H2_SKIP(4)) {
                const float *wq = P + OFF_HSF + (size_t)tl * (2 * 2 * 2 * 2 * HF);
                constexpr int AH = H2_AHS;
                f16x8 a[AH + 1][2];
#pragma unroll
                for (int s0_ = 0; s0_ < AH; ++s0_) {
                    if (H2_XP_ON(0)) { a[s0_][0] = pre_s[s0_][0]; a[s0_][1] = pre_s[s0_][1]; }
                    else load_a2(a[s0_], H2_W(wq + s0_ * 2 * HF), H2_LN);
                }
                H2_PRIO_ON();
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {          // s8 = (class, part, chunk) in OFF_HSF's order
                    if (s8 + AH < 8) load_a2(a[(s8 + AH) % (AH + 1)], H2_WST(wq + (s8 + AH) * 2 * HF), H2_LST);
                    f16x8 b[2][2];
#pragma unroll
                    for (int fr = 0; fr < 2; ++fr) {
                        const unsigned char *bs = smem + ((s8 >> 1) * 2) * H2_EO_PL + (4 * (s8 & 1) + q) * H2_EO_KG + (16 * (2 * fp + fr) + i) * 16;
                        b[fr][0] = *reinterpret_cast<const f16x8 *>(bs);
                        b[fr][1] = *reinterpret_cast<const f16x8 *>(bs + H2_EO_PL);
                    }
                    const f16x8 (&ac)[2] = a[s8 % (AH + 1)];      // the two frames alternate so that consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int fr = 0; fr < 2; ++fr) mid[s8 >> 1][fr] = mfma_f16(ac[1], b[fr][0], mid[s8 >> 1][fr]);
#pragma unroll
                    for (int fr = 0; fr < 2; ++fr) mid[s8 >> 1][fr] = mfma_f16(ac[0], b[fr][1], mid[s8 >> 1][fr]);
#pragma unroll
                    for (int fr = 0; fr < 2; ++fr) hi[s8 >> 1][fr] = mfma_f16(ac[0], b[fr][0], hi[s8 >> 1][fr]);
                }
                H2_PRIO_OFF();
            }
            f16x8 ab[2][2];                               // the bin-64 piece's two chunks: requested before the magnitudes, used after them
            load_a2(ab[0], H2_WST(wb), H2_LST);
            load_a2(ab[1], H2_WST(wb + 2 * HF), H2_LST);
            if (H2_XP_ON(1)) {   // conv1's first two sets
                const float *w1 = P + OFF_H1 + wave * (4 * 3 * 2 * HF);
                load_a2(pre_1[0], H2_W12(w1), H2_L12);
                load_a2(pre_1[1], H2_W12(w1 + 2 * HF), H2_L12);
            }
#pragma unroll
            for (int fr = 0; fr < 2; ++fr) {
                const f32x4 ere = join2(hi[0][fr], mid[0][fr]), eim = join2(hi[1][fr], mid[1][fr]);
                const f32x4 ore = join2(hi[2][fr], mid[2][fr]), oim = join2(hi[3][fr], mid[3][fr]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pre = ere[r] + ore[r], pim = eim[r] + oim[r], nre = ere[r] - ore[r], nim = eim[r] - oim[r];
                    mk[fr][r] = mag_sqrt(pre * pre + pim * pim);
                    mn[fr][r] = mag_sqrt(nre * nre + nim * nim);
                }
            }
#pragma unroll
            for (int fr = 0; fr < 2; ++fr) {
                f32x4 bhi = {b64n0 * x0p[fr], 0.f, 0.f, 0.f}, bmid = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    f16x8 b[2];
                    const unsigned char *bs = smem + ((bc * 2 + bp) * 2) * H2_EO_PL + (4 * ch + q) * H2_EO_KG + (16 * (2 * fp + fr) + i) * 16;
                    b[0] = *reinterpret_cast<const f16x8 *>(bs);
                    b[1] = *reinterpret_cast<const f16x8 *>(bs + H2_EO_PL);
                    if (!H2_SKIP(4)) mfma_split3(ab[ch], b, bhi, bmid);
                }
                b64[fr] = fmaf(bmid[0], H1_INV, bhi[0]);
            }
        }
        H2_SYNC(3);          // every wave is done reading the operand planes: the |X| planes may overwrite them
        H2_MARK(3);
        {
            H2_IDS();             // (fresh indices: the store offsets below must not be computed -- and parked -- in front of the GEMM)
            const int g = 4 * (wave & 3) + q;
#pragma unroll
            for (int fr = 0; fr < 2; ++fr) {
                const int f = 2 * (wave >> 2) + fr;
                unsigned char *frp = smem + f * H2_FR128;
                store_h4(frp, H2_PL128, g, i, mk[fr], amax);                  // slots 4 g + r        = bins 4 g + r
                store_h4(frp, H2_PL128, 16 + g, i, mn[fr], amax);             // slots 64 + 4 g + r   = bins 128 - (4 g + r); g = 0, r = 0 is bin 128:
                if (g == 0) nyq[f * 16 + i] = mn[fr][0];                      //   it goes to the scratch, and slot 64 is rewritten below with bin 64
                if (q == 0) b64p[(wave * 2 + fr) * 16 + i] = b64[fr];
            }
        }
        H2_SYNC(4);
        if (tid < 64) {                                                       // bin 64: frame tid / 16, clip tid % 16
            const int f = tid >> 4, c = tid & 15;
            const float *pp = b64p + ((f >> 1) * 8 + (f & 1)) * 16 + c;       // waves 4 (f >> 1) + (class, part), part fastest
            const float re = pp[0] + pp[2 * 2 * 16], im = pp[1 * 2 * 16] + pp[3 * 2 * 16];
            store_h1(smem + f * H2_FR128, H2_PL128, 64, c, mag_sqrt(re * re + im * im), amax);
        }
    }
    H2_SYNC(5);
    H2_MARK(4);

    // ---------------- phase 2: conv1 129->128, k3 s1 p1, ReLU -- direct: out[f] = sum_tap W[tap] in[f + tap - 1], wave = 16 output channels
    {
        H2_IDS();
        const int rt = wave;
        f32x4 hi[4], mid[4];
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
                mid[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float *wq = P + OFF_H1 + rt * (4 * 3 * 2 * HF);
        constexpr int AH = H2_AH1;
        f16x8 a[AH + 1][2];
#pragma unroll
        for (int s0_ = 0; s0_ < AH; ++s0_) {
            if (H2_XP_ON(1)) { a[s0_][0] = pre_1[s0_][0]; a[s0_][1] = pre_1[s0_][1]; }
            else load_a2(a[s0_], H2_W(wq + s0_ * 2 * HF), H2_LN);
        }
        H2_PRIO_ON();
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            f16x8 b[4][2];
#pragma unroll
            for (int f = 0; f < 4; ++f) load_b2(b[f], smem + f * H2_FR128, H2_PL128, kc, q, i);
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int s = kc * 3 + tap;
                if (s + AH < 12) load_a2(a[(s + AH) % (AH + 1)], H2_W12(wq + (s + AH) * 2 * HF), H2_L12);
                const f16x8 (&ac)[2] = a[s % (AH + 1)];
                // three products per (frame, tap), frames innermost so that consecutive MFMAs hit different accumulators
#define H2_TERM(AP, BP, ACC)                                                                  \
    _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                           \
        const int fi = f + tap - 1;                                                           \
        if (fi >= 0 && fi < 4 && !H2_SKIP(5)) ACC[f] = mfma_f16(ac[AP], b[fi][BP], ACC[f]);   \
    }
                H2_TERM(1, 0, mid) H2_TERM(0, 1, mid) H2_TERM(0, 0, hi)
#undef H2_TERM
            }
        }
        H2_PRIO_OFF();
        if (H2_XP_ON(2)) {   // conv2's first two sets
            const float *w2 = P + OFF_H2 + ((wave & 3) * 4 + 2 * (wave >> 2)) * (3 * 2 * HF);
            load_a2(pre_2[0], H2_W12(w2), H2_L12);
            load_a2(pre_2[1], H2_W12(w2 + 2 * HF), H2_L12);
        }
        H2_SYNC(6);          // every wave is done reading the |X| planes: conv1's output may now overwrite them
        H2_MARK(5);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const f32x4 s = join2(hi[f], mid[f]);
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = fmaxf(s[r], 0.f);
            store_h4(smem + f * H2_FR128, H2_PL128, 4 * rt + q, i, y, amax);
        }
    }
    H2_SYNC(7);
    H2_MARK(6);

    // ---------------- phase 3: conv2 128->64, k3 s2 p1, ReLU: out frame o reads in frames 2 o - 1 .. 2 o + 1; wave = (16 channels, half of K)
    {
        H2_IDS();
        const int rt = wave & 3, kh = wave >> 2;
        f32x4 hi[2], mid[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) { hi[o] = f32x4{0.f, 0.f, 0.f, 0.f}; mid[o] = hi[o]; }
        const float *wq = P + OFF_H2 + (rt * 4 + 2 * kh) * (3 * 2 * HF);
        constexpr int RING = H2_XP_ON(4) ? 6 : 3;      // the wave's whole stream (six sets) up front, or two steps ahead on three register sets
        f16x8 a[RING][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (H2_XP_ON(2)) { a[s][0] = pre_2[s][0]; a[s][1] = pre_2[s][1]; }
            else load_a2(a[s], H2_W(wq + s * 2 * HF), H2_LN);
        }
        if (RING == 6) {
#pragma unroll
            for (int s = 2; s < 6; ++s) load_a2(a[s % RING], H2_W12(wq + s * 2 * HF), H2_L12);
        }
        H2_PRIO_ON();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 b[4][2];
#pragma unroll
            for (int f = 0; f < 4; ++f) load_b2(b[f], smem + f * H2_FR128, H2_PL128, 2 * kh + kk, q, i);
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int s = kk * 3 + tap;
                if (RING == 3 && s + 2 < 6) load_a2(a[(s + 2) % RING], H2_W12(wq + (s + 2) * 2 * HF), H2_L12);
                const f16x8 (&ac)[2] = a[s % RING];
#define H2_TERM(AP, BP, ACC)                                                                  \
    _Pragma("unroll") for (int o = 0; o < 2; ++o) {                                           \
        const int fi = 2 * o + tap - 1;                                                       \
        if (fi >= 0 && !H2_SKIP(3)) ACC[o] = mfma_f16(ac[AP], b[fi][BP], ACC[o]);             \
    }
                H2_TERM(1, 0, mid) H2_TERM(0, 1, mid) H2_TERM(0, 0, hi)
#undef H2_TERM
            }
        }
        H2_PRIO_OFF();
        f32x4 s2[2] = {join2(hi[0], mid[0]), join2(hi[1], mid[1])};
        float *exc = reinterpret_cast<float *>(smem + H2_EXC2);
        if (kh == 1) {
#pragma unroll
            for (int o = 0; o < 2; ++o) *reinterpret_cast<f32x4 *>(exc + ((rt * 2 + o) * 64 + lane) * 4) = s2[o];
        }
        H2_SYNC(8);
        if (kh == 0) {
            const f32x4 bias = ldg4(P + OFF_B2 + 16 * rt + 4 * q);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const f32x4 other = *reinterpret_cast<const f32x4 *>(exc + ((rt * 2 + o) * 64 + lane) * 4);
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = fmaxf(s2[o][r] + other[r] + bias[r], 0.f);
                store_h4(smem + H2_R1 + sub * H2_T2 + o * H2_FR2, H2_PL2, 4 * rt + q, i, y, amax);
            }
        }
    }
    H2_SYNC(9);
    H2_MARK(7);
    }      // sub

#if H2_TRACE
    h2_tr_off = 64 - 2 * 10;          // the joint tail's barriers (H2_SYNC(10) ...) land at marks 64 ...
#endif
    constexpr int AHEAD = NSUB > 1 ? H2_AHI : 3;      // W_ih's stream runs this many steps ahead
    f16x8 pre_3[2][2], pre_4[2][2], pre_ih[AHEAD][2];
    {   H2_IDS();
        // conv3's two sets (not requested inside the tile loop: a value that only the last iteration defines would be carried, and spilled, around it)
        const float *w3 = P + OFF_H3 + ((wave & 3) * 2 + (wave >> 2)) * (2 * 2 * HF);
        load_a2(pre_3[0], H2_W(w3), H2_LN);
        load_a2(pre_3[1], H2_W(w3 + 2 * HF), H2_LN);
    }
    // ---------------- phase 4: conv3 64->64, k3 s2 p1, ReLU (one output frame; tap 0 reads padding), both tiles: wave = (16 channels, tap 1 | 2)
    {
        H2_IDS();
        const int rt = wave & 3, th = wave >> 2;      // tap th + 1 reads conv2's frame th
        f32x4 hi[NSUB], mid[NSUB];
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) { hi[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; mid[sb] = hi[sb]; }
        if (H2_XP_ON(3)) {   // conv4's two sets
            const float *w4 = P + OFF_H4 + wave * (2 * 2 * HF);
            load_a2(pre_4[0], H2_W(w4), H2_LN);
            load_a2(pre_4[1], H2_W(w4 + 2 * HF), H2_LN);
        }
        const f16x8 (&a)[2][2] = pre_3;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) {
                f16x8 b[2];
                load_b2(b, smem + H2_R1 + sb * H2_T2 + th * H2_FR2, H2_PL2, kc, q, i);
                if (!H2_SKIP(3)) mfma_split3(a[kc], b, hi[sb], mid[sb]);
            }
        float *exc = reinterpret_cast<float *>(smem + H2_EXC3);
        if (th == 1) {
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) *reinterpret_cast<f32x4 *>(exc + ((sb * 4 + rt) * 64 + lane) * 4) = join2(hi[sb], mid[sb]);
        }
        H2_SYNC(10);
        if (th == 0) {
            const f32x4 bias = ldg4(P + OFF_B3 + 16 * rt + 4 * q);
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) {
                const f32x4 s3 = join2(hi[sb], mid[sb]), other = *reinterpret_cast<const f32x4 *>(exc + ((sb * 4 + rt) * 64 + lane) * 4);
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = fmaxf(s3[r] + other[r] + bias[r], 0.f);
                store_h4(smem + H2_C3 + sb * H2_T3, H2_PL3, 4 * rt + q, i, y, amax);
            }
        }
    }
    H2_SYNC(11);
    H2_MARK(8);

    // ---------------- phase 5: conv4 64->128, k3 s1 p1, ReLU (one frame in / out: centre tap only), both tiles
    {
        H2_IDS();
        const int rt = wave;
        const f32x4 bias = ldg4(P + OFF_B4 + 16 * rt + 4 * q);
        f32x4 hi[NSUB], mid[NSUB];
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) { hi[sb] = bias; mid[sb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (H2_XP_ON(3)) {   // W_ih's first sets
            const float *wi = P + OFF_HIH + wave * (4 * 4 * 2 * HF);
#pragma unroll
            for (int s0_ = 0; s0_ < AHEAD; ++s0_) load_a2(pre_ih[s0_], H2_WIH(wi + s0_ * 2 * HF), H2_LIH);
        }
        if (!H2_XP_ON(3)) {
            const float *w4 = P + OFF_H4 + wave * (2 * 2 * HF);
            load_a2(pre_4[0], H2_W(w4), H2_LN);
            load_a2(pre_4[1], H2_W(w4 + 2 * HF), H2_LN);
        }
        const f16x8 (&a)[2][2] = pre_4;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) {
                f16x8 b[2];
                load_b2(b, smem + H2_C3 + sb * H2_T3, H2_PL3, kc, q, i);
                if (!H2_SKIP(3)) mfma_split3(a[kc], b, hi[sb], mid[sb]);
            }
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) {
            const f32x4 s = join2(hi[sb], mid[sb]);
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = fmaxf(s[r], 0.f);
            store_h4(smem + H2_C4 + sb * H2_T4, H2_PL4, 4 * rt + q, i, y, amax);
        }
        // nothing is split after this point: the workgroup's verdict on the fp16 range (one word per wave, read behind the barrier below)
        const bool bad = !(amax <= H_MAX);
        if (__ballot(bad) != 0ULL && lane == 0) reinterpret_cast<unsigned *>(scr)[128 + wave] = 1u;
        else if (lane == 0) reinterpret_cast<unsigned *>(scr)[128 + wave] = 0u;
    }
    H2_SYNC(12);
    H2_MARK(9);

    // ---------------- phase 6: LSTM input projection for the workgroup's tiles at once, gate-major (D rows = hidden units
    // 16 wave + 4 q + r, columns = clips): every W_ih fragment is loaded once and multiplies NSUB column tiles
    {
        H2_IDS();
        f32x4 hi[NSUB][4], mid[NSUB][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bg = ldg4(P + OFF_BG + g * 128 + wave * 16 + 4 * q);
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) { hi[sb][g] = bg; mid[sb][g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        const float *wq = P + OFF_HIH + wave * (4 * 4 * 2 * HF);
        f16x8 a[AHEAD + 1][2];
#pragma unroll
        for (int s0_ = 0; s0_ < AHEAD; ++s0_) {
            if (H2_XP_ON(3)) { a[s0_][0] = pre_ih[s0_][0]; a[s0_][1] = pre_ih[s0_][1]; }
            else load_a2(a[s0_], H2_W(wq + s0_ * 2 * HF), H2_LN);
        }
        H2_PRIO_ON();
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            f16x8 b[NSUB][2];
#pragma unroll
            for (int sb = 0; sb < NSUB; ++sb) load_b2(b[sb], smem + H2_C4 + sb * H2_T4, H2_PL4, kc, q, i);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int s = kc * 4 + g;
                if (s + AHEAD < 16) load_a2(a[(s + AHEAD) % (AHEAD + 1)], H2_WIH(wq + (s + AHEAD) * 2 * HF), H2_LIH);
                const f16x8 (&ac)[2] = a[s % (AHEAD + 1)];
                if (!H2_SKIP(6)) {
#define H2_TERM(AP, BP, ACC) _Pragma("unroll") for (int sb = 0; sb < NSUB; ++sb) ACC[sb][g] = mfma_f16(ac[AP], b[sb][BP], ACC[sb][g]);
                    H2_TERM(1, 0, mid) H2_TERM(0, 1, mid) H2_TERM(0, 0, hi)
#undef H2_TERM
                }
            }
        }
        H2_PRIO_OFF();
#pragma unroll
        for (int sb = 0; sb < NSUB; ++sb) {
#if H2_PAIR_T
            const int grp_s = (int)(blk % G), t_s = (int)(blk / G) * NSUB + sb;
            const long long tile_raw = t_s < T ? 0 : ntile;          // (valid / not, for the test below)
            float *dst = gx + ((size_t)(t_s < T ? t_s : T - 1) * Gws + g0 + grp_s) * GX_TILE_FLOATS + (size_t)wave * 4 * 256 + lane * 4;
#else
            const long long tile_raw = blk * NSUB + sb;
            const int tile_id = (int)(tile_raw < ntile ? tile_raw : ntile - 1);
            float *dst = gx + ((size_t)(tile_id / G) * Gws + g0 + tile_id % G) * GX_TILE_FLOATS + (size_t)wave * 4 * 256 + lane * 4;
#endif
            // a workgroup that split anything outside the fp16 range hands the recurrent kernel NaN, not numbers that look like gate
            // pre-activations: a caller that never reads vadx_silero_range_flag gets NaN scores for these clips, not plausible ones
            const unsigned *bw = reinterpret_cast<const unsigned *>(scr) + 128;
            const bool poison = __builtin_amdgcn_readfirstlane((int)(bw[0] | bw[1] | bw[2] | bw[3] | bw[4] | bw[5] | bw[6] | bw[7])) != 0;
            if (tile_raw < ntile) {
                if (!poison) {           // (workgroup-uniform: a branch, not 32 selects)
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(dst + g * 256) = join2(hi[sb][g], mid[sb][g]);
                } else {
                    const float qnan = __builtin_nanf("");
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4 *>(dst + g * 256) = f32x4{qnan, qnan, qnan, qnan};
                }
            }
        }
    }
    }      // blk
    // ---- range check: anything split above the largest finite fp16 raises the blob's sticky flag.  (A NaN sample does not: v_max3_f32
    // returns its non-NaN operands, so NaN never reaches amax -- it travels through the products instead.)
    if (!(amax <= H_MAX)) {
        unsigned *fl = reinterpret_cast<unsigned *>(const_cast<float *>(P)) + OFF_HFLAG + 1;
        atomicOr(fl, 1u);
        atomicMax(fl + 1, __float_as_uint(amax));
    }
    H2_MARK(10);
    H2_CLK();
#if H2_TRACE
    h2_tr_off = 0;
    H2_TR(127);
#endif
}

// ---- persistent LSTM on fp16 x 2 products ------------------------------------------------------
// Same decomposition as silero_lstm_split_kernel: one persistent workgroup per 16 clips, wave w owns hidden units 16 w .. 16 w + 15 of all
// four gates, h exchanged through double-buffered LDS planes, one barrier per step.  Both planes of W_hh stay in VGPRs for the whole
// clip (128 registers: no weight bytes in LDS at all); 48 v_mfma_f32_16x16x32_f16 per wave and step instead of 96 bf16.  |h| <= 1 and W_hh
// passed the pack-time range check, so nothing here can leave the fp16 range.
constexpr int LH_HPL = 4096, LH_HBUF = 8192;     // h planes [2 buffers][2 planes][16 k-groups][16 clips][8 fp16]
constexpr int LH_PART = 2 * LH_HBUF;             // f32 [2][8 waves][16 clips]
constexpr int LH_BYTES = LH_PART + 1024;

__global__ __launch_bounds__(512, 2) void silero_lstm_h2_kernel(
    const float *__restrict__ P, const float *__restrict__ gx, const float *__restrict__ state0,
    int B, int G, int T, float *__restrict__ probs, long long probs_stride, float *__restrict__ state_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *part = reinterpret_cast<float *>(smem + LH_PART);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int grp = blockIdx.x;
    const long long b = (long long)grp * 16 + n;
    const bool bvalid = b < B;
    const int u0 = wave * 16 + 4 * q;             // this lane's 4 hidden units
    if (ldg1(P + OFF_HFLAG) == 0.f) {   // uniform: this blob cannot run on fp16 x 2 (a weight outside the fp16 range): as the encoder, flag bit 1,
        if (tid == 0 && grp == 0) atomicOr(reinterpret_cast<unsigned *>(const_cast<float *>(P)) + OFF_HFLAG + 1, 2u);      // and no plausible score
        if (wave == 0 && lane < 16 && bvalid)
            for (int t = 0; t < T; ++t) probs[b * probs_stride + t] = __builtin_nanf("");
        return;
    }

    f16x8 a[4][4][2];
    {
        const float *wq = P + OFF_HHH + (size_t)wave * (4 * 4 * 2 * HF);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) load_a2(a[g][kc], wq + (g * 4 + kc) * 2 * HF, lane);
    }
    // the decoder's ReLU as (h + |h|) / 2 with the 1/2 folded into its weights: the same value bit for bit (both scalings are exact), but a NaN
    // state stays NaN in the score -- fmaxf(h, 0) would turn it into 0 and a poisoned clip (see the encoder's gx) into a plausible number
    const f32x4 dwh = ldg4(P + OFF_DW + u0) * 0.5f;
    const float db = P[OFF_DB];

    float amax = 0.f;
    f32x4 c = {0.f, 0.f, 0.f, 0.f}, h = {0.f, 0.f, 0.f, 0.f};
    if (state0 != nullptr && bvalid) {
        h = *reinterpret_cast<const f32x4 *>(state0 + b * 128 + u0);
        c = *reinterpret_cast<const f32x4 *>(state0 + ((long long)B + b) * 128 + u0);
    }
    store_h4(smem, LH_HPL, 4 * wave + q, n, h, amax);
    __syncthreads();

    const float *gsrc = gx + (size_t)grp * GX_TILE_FLOATS + (size_t)wave * 4 * 256 + lane * 4;
    const size_t gstep = (size_t)G * GX_TILE_FLOATS;
    // gx (the encoder's gate pre-activations, 32 KB per step and workgroup out of HBM: 2.6 GB per launch at config 2 = 4.3 TB/s) is requested
    // LH_GX_AHEAD steps ahead: with one step of lead a lone workgroup ran 1.42 us per step and the full grid 1.95 -- the loaded HBM's
    // latency exceeds a step.  The ring is indexed statically (the step loop is unrolled by its depth).
// (round 6, measured, removed: gate-major MFMA order -- a gate's twelve products finish before the next gate's start, so that its non-linearity
//  runs under the next gate's MFMAs, all four B fragment pairs held in registers -- 0.590 -> 0.597 ms with gx two steps ahead, 0.65 with three
//  (44 B of scratch): the compiler interleaves v_exp / v_rcp with the MFMAs as intended, the step is no shorter.)
#ifndef LH_GX_AHEAD
#define LH_GX_AHEAD 3
#endif
    constexpr int GA = LH_GX_AHEAD;
    f32x4 gq[GA][4];
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int tj = j < T ? j : T - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) gq[j][g] = *reinterpret_cast<const f32x4 *>(gsrc + tj * gstep + g * 256);
    }

    int cur = 0;
    for (int t0 = 0; t0 < T; t0 += GA) {
#pragma unroll
    for (int j = 0; j < GA; ++j) {
        const int t = t0 + j;
        if (t >= T) break;
        f32x4 hi[4], mid[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { hi[g] = gq[j][g]; mid[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        {
            const int tn = (t + GA < T) ? t + GA : T - 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) gq[j][g] = *reinterpret_cast<const f32x4 *>(gsrc + tn * gstep + g * 256);
        }
        const unsigned char *hb = smem + cur * LH_HBUF;
        float dpart = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            f16x8 bb[2];
            load_b2(bb, hb, LH_HPL, kc, q, n);
            // the three products, gates innermost so that consecutive MFMAs hit different accumulators
#pragma unroll
            for (int g = 0; g < 4; ++g) mid[g] = mfma_f16(a[g][kc][1], bb[0], mid[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) mid[g] = mfma_f16(a[g][kc][0], bb[1], mid[g]);
#pragma unroll
            for (int g = 0; g < 4; ++g) hi[g] = mfma_f16(a[g][kc][0], bb[0], hi[g]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = gate_sigmoid(fmaf(mid[0][r], H1_INV, hi[0][r])), fg = gate_sigmoid(fmaf(mid[1][r], H1_INV, hi[1][r]));
            const float gg = gate_tanh(fmaf(mid[2][r], H1_INV, hi[2][r])), og = gate_sigmoid(fmaf(mid[3][r], H1_INV, hi[3][r]));
            c[r] = fg * c[r] + ig * gg;
            h[r] = og * gate_tanh(c[r]);
            dpart = fmaf(dwh[r], h[r] + __builtin_fabsf(h[r]), dpart);
        }
        const int nxt = cur ^ 1;
        store_h4(smem + nxt * LH_HBUF, LH_HPL, 4 * wave + q, n, h, amax);
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
    }
    }
    if (state_n != nullptr && bvalid) {
        *reinterpret_cast<f32x4 *>(state_n + b * 128 + u0) = h;
        *reinterpret_cast<f32x4 *>(state_n + ((long long)B + b) * 128 + u0) = c;
    }
    if (!(amax <= H_MAX)) {           // only a caller-supplied initial state can do this
        unsigned *fl = reinterpret_cast<unsigned *>(const_cast<float *>(P)) + OFF_HFLAG + 1;
        atomicOr(fl, 1u);
        atomicMax(fl + 1, __float_as_uint(amax));
    }
    // ... and then this clip group's scores are NaN, not numbers (cf. the encoder's gx)
    __syncthreads();
    if (__ballot(!(amax <= H_MAX)) != 0ULL && lane == 0) reinterpret_cast<unsigned *>(part)[wave] = 1u;
    else if (lane == 0) reinterpret_cast<unsigned *>(part)[wave] = 0u;
    __syncthreads();
    {
        const unsigned *bw = reinterpret_cast<const unsigned *>(part);
        if ((bw[0] | bw[1] | bw[2] | bw[3] | bw[4] | bw[5] | bw[6] | bw[7]) != 0u && wave == 0 && lane < 16 && bvalid)
            for (int t = 0; t < T; ++t) probs[b * probs_stride + t] = __builtin_nanf("");
    }
}

int silero_lstm_h2_launch(const float *packed, const float *gx, const float *state0, int batch, int G, int steps, float *probs,
                          long long probs_stride, float *state_n, void *stream) {
    hipLaunchKernelGGL(silero_lstm_h2_kernel, dim3(G), dim3(512), LH_BYTES, static_cast<hipStream_t>(stream), packed, gx, state0, batch, G,
                       steps, probs, probs_stride, state_n);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

// VADX_H2_NSUB: tiles a workgroup encodes one after the other before ONE joint conv3 / conv4 / W_ih pass
#ifndef VADX_H2_NSUB
#define VADX_H2_NSUB 2
#endif
template <typename S>
int silero_encode_h2_launch(const float *packed, const S *src, float in_scale, long long n_valid, long long row_stride,
                            long long origin, int batch, int G, int steps, int Gws, int first_group, float *gx, void *stream) {
    constexpr int NS = VADX_H2_NSUB;
    constexpr int LDS = (H2_R1 + NS * H2_T2 > 65536 ? H2_R1 + NS * H2_T2 : 65536) + H2_LDS_PAD;
    VADX_DYN_LDS((silero_encode_h2_kernel<S, NS>), LDS);
    const long long nblk = H2_PAIR_T ? (long long)G * ((steps + NS - 1) / NS) : ((long long)G * steps + NS - 1) / NS;
    hipLaunchKernelGGL((silero_encode_h2_kernel<S, NS>), dim3((unsigned)nblk), dim3(H2_THREADS), LDS, static_cast<hipStream_t>(stream),
                       packed, src, in_scale, n_valid, row_stride, origin, batch, G, steps, Gws, first_group, gx);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
template int silero_encode_h2_launch<float>(const float *, const float *, float, long long, long long, long long, int, int, int, int, int, float *, void *);
template int silero_encode_h2_launch<int16_t>(const float *, const int16_t *, float, long long, long long, long long, int, int, int, int, int, float *, void *);

}  // namespace silero
}  // namespace vadx
