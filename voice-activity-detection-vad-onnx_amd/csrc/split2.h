// split2.h -- float32-class products on the fp16 matrix pipe ("fp16 x 2" split products), gfx950 only.
//
// A float32 value x is represented by TWO fp16 terms taken with round-to-nearest:
//     h0 = RN16(x),      h1 = RN16((x - h0) * 2^11)          x ~ h0 + h1 * 2^-11
// The residual x - h0 is exact in float32 and SIGNED (|x - h0| <= half an fp16 ulp of x), so the two 11-bit significands cover
// 11 + 1 + 11 = 23 bits: |x - h0 - h1 2^-11| <= 2^-23 |x| -- one float32 ulp at worst, 2^-24.8 rms -- for every |x| in
// [2^-14, 65504]; below 2^-14 (h0 subnormal) the error is absolute, <= 2^-36.  The residual term is stored SCALED by 2^11 so that it keeps
// its 11 bits wherever h0 is a normal number, and the two cross products go into an accumulator of their own ("mid") that joins the h0 h0
// accumulator ("hi") with one multiply by 2^-11 at the end:
//     w x  ~  w0 x0  +  2^-11 (w0 x1 + w1 x0)            (the dropped w1 x1 2^-22 term is <= 2^-22, 2^-25.6 rms, of the product)
// THREE v_mfma_f32_16x16x32_f16 per K = 32 step, against six bf16 MFMAs for the exact three-way bf16 split (split3.h) and eight
// v_mfma_f32_16x16x4_f32.  tools/f16x2_probe.sh (profiles/r05_f16x2_probe.txt), error / sum |w x| against a float64 evaluation of the SAME
// float32 operands: max 0.7 - 1.1e-7, rms 1.1 - 1.9e-8 (normal operands, K = 64 ... 512) -- the bf16 x 3 figures (0.7 - 0.9e-7, 1.0 - 1.5e-8),
// and below the f32 MFMA chain's (1.9 - 2.5e-7, 2.8e-8: its error is the accumulation's, one rounding per K = 4 step); rate 23.8 - 24.9 ns
// per K = 32 step and SIMD against 46 - 47 (bf16 x 3) and 108 - 114 (f32).
// What fp16 does NOT have is float32's exponent range: an activation beyond 65504 becomes +-inf.  The kernels that use this header keep a
// running max |x| of everything they split and raise a sticky flag the host reads with the results (silero_h2.hip: OFF_HFLAG); a flagged
// batch is recomputed on the bf16 x 3 kernels, whose terms have float32's range.
//
// Layouts are split3.h's with two planes: A operand (weights, constant) split offline, one "fragment" per (16-row tile, 32-k chunk, plane)
// = what a wave loads with ONE 16-byte-per-lane request (lane 16 g + i holds W[16 tile + i][32 chunk + 8 g + e], e = 0..7);
// B operand (activations, LDS) per plane [k / 8][16 columns][8 fp16].
#pragma once
#include <stdint.h>
#include <string.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifndef VADX_U32X2
#define VADX_U32X2
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#endif

namespace vadx {

constexpr int HFRAG = 256;          // floats (= 1 KiB) per fp16 A fragment: 64 lanes x 8 fp16
constexpr float H1_SCALE = 2048.f, H1_INV = 1.f / 2048.f;
constexpr float H_MAX = 65504.f;    // largest finite fp16

// ---- host: the two terms, round-to-nearest-even (the compiler's float -> _Float16 conversion)
inline void split2_host(float x, uint16_t &h0, uint16_t &h1) {
    const _Float16 a = (_Float16)x;
    const _Float16 b = (_Float16)((x - (float)a) * H1_SCALE);
    memcpy(&h0, &a, 2);
    memcpy(&h1, &b, 2);
}
// host: write W[row][k] (row in the tile 0..15, k in the chunk 0..31) into the fragment pair frag2[0..1] (each HFRAG floats).
// Returns |w| so that callers can refuse weights outside the fp16 range.
inline float hfrag_put(float *frag2, int row, int k, float w) {
    uint16_t h[2];
    split2_host(w, h[0], h[1]);
    for (int p = 0; p < 2; ++p) {
        uint16_t *f = reinterpret_cast<uint16_t *>(frag2 + (size_t)p * HFRAG);
        f[(size_t)((k / 8) * 16 + row) * 8 + (k % 8)] = h[p];
    }
    return w < 0.f ? -w : w;
}

#if defined(__HIPCC__)
__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

typedef const __attribute__((address_space(1))) f16x8 *global_f16x8_ptr;
__device__ __forceinline__ f16x8 ldh(const float *frag, int lane) { return *((global_f16x8_ptr)(frag) + lane); }

// four float32 values (consecutive k of one column) -> their two planes, four fp16 (8 bytes) each: v_cvt_pk_f16_f32 (RN) x 2,
// v_cvt_f32_f16 x 4, v_pk_add_f32 x 2, v_pk_mul_f32 x 2, v_cvt_pk_f16_f32 x 2 = 3 VALU per value.  `amax` is the running max |x|.
// (fp contraction is OFF inside the split: with it the compiler folds the multiply that PRODUCED x into the residual -- x - h0 becomes
//  fma(a, b, -h0) on the unrounded product -- and the planes then encode a value that is not the float32 x the caller also keeps, e.g. the
//  LSTM state a later launch resumes from: tests/test_gpu_silero.py::test_spanned_schedule_is_bitwise_identical caught one-ulp differences.)
__device__ __forceinline__ void split2x4(const f32x4 x, u32x2 &p0, u32x2 &p1, float &amax) {
#pragma clang fp contract(off)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ xa = {x[0], x[1]}, xb = {x[2], x[3]};
    const f16x2 a = __builtin_convertvector(xa, f16x2), b = __builtin_convertvector(xb, f16x2);
    const f32x2_ ra = (xa - __builtin_convertvector(a, f32x2_)) * H1_SCALE, rb = (xb - __builtin_convertvector(b, f32x2_)) * H1_SCALE;
    const f16x2 c = __builtin_convertvector(ra, f16x2), d = __builtin_convertvector(rb, f16x2);
    p0 = u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
    p1 = u32x2{__builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)};
    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[0])), __builtin_fabsf(x[1]));      // v_max3_f32 with |.| modifiers
    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(x[2])), __builtin_fabsf(x[3]));
    // (pinned here: nothing reads amax until the kernel's end, so the scheduler otherwise sinks the whole max chain behind the next GEMM and
    //  keeps every x alive for it -- in silero_h2.hip that was 12 spilled registers per thread and 2.3 GB of scratch traffic per launch)
    asm volatile("" : "+v"(amax));
    // (round 6, measured, not kept: (x - h0) 2^11 as fma(h0, -2^11, x 2^11) with the fp16 -> f32 conversion inside v_fma_mix_f32 -- the same
    //  bits in 12 instead of 14 VALU instructions per four values -- Silero encoder 3.57 -> 3.76 - 3.81 ms, recurrent kernel 0.577 -> 0.559)
}
// one float32 value -> its two fp16 terms
__device__ __forceinline__ void split2x1(float x, unsigned short &h0, unsigned short &h1, float &amax) {
#pragma clang fp contract(off)
    const _Float16 a = (_Float16)x;
    const _Float16 b = (_Float16)((x - (float)a) * H1_SCALE);
    h0 = __builtin_bit_cast(unsigned short, a);
    h1 = __builtin_bit_cast(unsigned short, b);
    amax = __builtin_fmaxf(amax, __builtin_fabsf(x));
    asm volatile("" : "+v"(amax));
}

// The three products of one K = 32 step for one (row tile, column tile): cross terms into `mid` (scale 2^11), h0 x h0 into `hi`.
__device__ __forceinline__ void mfma_split3(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x4 &hi, f32x4 &mid) {
    mid = mfma_f16(a[1], b[0], mid);
    mid = mfma_f16(a[0], b[1], mid);
    hi = mfma_f16(a[0], b[0], hi);
}
__device__ __forceinline__ f32x4 join2(const f32x4 hi, const f32x4 mid) {
    return f32x4{__builtin_fmaf(mid[0], H1_INV, hi[0]), __builtin_fmaf(mid[1], H1_INV, hi[1]), __builtin_fmaf(mid[2], H1_INV, hi[2]),
                 __builtin_fmaf(mid[3], H1_INV, hi[3])};
}
#endif

}  // namespace vadx
