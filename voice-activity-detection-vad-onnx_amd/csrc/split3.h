// split3.h -- exact float32 products on the bf16 matrix pipe ("bf16 x 3" split products), gfx950 only.
//
// A float32 value x splits EXACTLY into three bf16 terms by truncation, x = t0 + t1 + t2 (8 + 8 + 8 significand bits, float32's
// exponent range: no scaling, no overflow cases).  A product w * x of two such values is the sum of nine bf16 x bf16 products, each
// exact in float32; the six with i + j <= 2 carry everything down to 2^-24 relative, the three dropped ones are <= 3 * 2^-24 |w x|.
// v_mfma_f32_16x16x32_bf16 issues at 16x the rate of v_mfma_f32_16x16x4_f32, so six of them (one K = 32 step) cost 6/16 of the
// eight f32 MFMAs they replace -- and, unlike the f32-input MFMA, they do not share the vector ALU's datapath, so VALU work hides
// beside them (tools/bf16x3_probe.sh: 43 ns against 108 ns per K = 32 step per SIMD; 47 ns with two VALU instructions per MFMA).
// Measured error against a float64 evaluation of the same float32 operands (same probe, K = 64 ... 512, random and DFT-table x PCM
// operands), with the five small products in an accumulator of their own ("lo") that joins the t0 x t0 accumulator ("hi") at the end:
// max 7e-8 ... 2.2e-7 of sum |w x| against 2.1e-7 ... 4.4e-7 for the f32 MFMA chain, rms 1e-8 against 2.8e-8 -- the split product is
// the MORE accurate of the two (fewer roundings: the hi products of a K = 32 step meet in the matrix unit's internal adder tree).
//
// Layouts.  A operand (weights, constant): split offline on the host, one "fragment" per (16-row tile, 32-k chunk, plane) = what a wave
// loads with ONE 16-byte-per-lane request: lane 16 g + i holds W[16 tile + i][32 chunk + 8 g + e], e = 0..7, as eight bf16.
// B operand (activations, LDS): per plane [k / 8][16 columns][8 bf16]: the lane's eight k of one column are 16 contiguous bytes and a
// wave's 64 requests are one contiguous 1 KiB (conflict-free in any lane grouping).  The producer's D fragment (lane 16 g + i: rows
// 4 g .. 4 g + 3 of column i) is four consecutive k of column i: one 8-byte store per plane.
#pragma once
#include <stdint.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace vadx {

constexpr int QFRAG = 256;      // floats (= 1 KiB) per bf16 A fragment: 64 lanes x 8 bf16

// ---- host: exact three-way split by truncation
inline void split3_host(float x, uint16_t &h0, uint16_t &h1, uint16_t &h2) {
    union { float f; uint32_t u; } v, t;
    v.f = x;
    t.u = v.u & 0xffff0000u;
    h0 = (uint16_t)(t.u >> 16);
    const float r1 = x - t.f;
    v.f = r1;
    t.u = v.u & 0xffff0000u;
    h1 = (uint16_t)(t.u >> 16);
    const float r2 = r1 - t.f;
    v.f = r2;
    h2 = (uint16_t)(v.u >> 16);
}

// host: write W[row][k] (row in the tile 0..15, k in the chunk 0..31) of plane-split fragments frag[0..2] (each QFRAG floats)
inline void qfrag_put(float *frag3, int row, int k, float w) {
    uint16_t h[3];
    split3_host(w, h[0], h[1], h[2]);
    for (int p = 0; p < 3; ++p) {
        uint16_t *f = reinterpret_cast<uint16_t *>(frag3 + (size_t)p * QFRAG);
        f[(size_t)((k / 8) * 16 + row) * 8 + (k % 8)] = h[p];
    }
}

#if defined(__HIPCC__)
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// the wave's A fragment (one 16-byte load per lane from the fragment's 1 KiB)
typedef const __attribute__((address_space(1))) bf16x8 *global_bf16x8_ptr;
__device__ __forceinline__ bf16x8 ldq(const float *frag, int lane) { return *((global_bf16x8_ptr)(frag) + lane); }

// four float32 values (consecutive k of one column) -> their three planes, four bf16 (8 bytes) each
// (fp contraction is OFF inside the splits, as in split2.h: with it the multiply that PRODUCED x is folded into the residual -- x - t0 becomes
//  fma(a, b, -t0) on the unrounded product -- and the planes then encode a value that is not the float32 x the caller also keeps, e.g. the
//  LSTM state a later launch resumes from.  The SLP vectoriser used to hide this by turning the subtractions into packed adds; the library is
//  built without it now: tests/test_gpu_silero.py::test_spanned_schedule_is_bitwise_identical[split-*] caught the one-ulp differences.)
__device__ __forceinline__ void split3x4(const f32x4 x, u32x2 &p0, u32x2 &p1, u32x2 &p2) {
#pragma clang fp contract(off)
    unsigned xb[4], r1b[4], r2b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xb[j] = __float_as_uint(x[j]);
        const float r1 = x[j] - __uint_as_float(xb[j] & 0xffff0000u);
        r1b[j] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(r1b[j] & 0xffff0000u);
        r2b[j] = __float_as_uint(r2);
    }
    // v_perm_b32: the high halves of two registers side by side (element j in the low half, j + 1 in the high half)
    p0 = u32x2{__builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u), __builtin_amdgcn_perm(xb[3], xb[2], 0x07060302u)};
    p1 = u32x2{__builtin_amdgcn_perm(r1b[1], r1b[0], 0x07060302u), __builtin_amdgcn_perm(r1b[3], r1b[2], 0x07060302u)};
    p2 = u32x2{__builtin_amdgcn_perm(r2b[1], r2b[0], 0x07060302u), __builtin_amdgcn_perm(r2b[3], r2b[2], 0x07060302u)};
}

// two float32 values (consecutive k of one column) -> their three planes, two bf16 (4 bytes) each
__device__ __forceinline__ void split3x2(float x0, float x1, unsigned &p0, unsigned &p1, unsigned &p2) {
#pragma clang fp contract(off)
    const unsigned a = __float_as_uint(x0), b = __float_as_uint(x1);
    const float ra = x0 - __uint_as_float(a & 0xffff0000u), rb = x1 - __uint_as_float(b & 0xffff0000u);
    const unsigned a1 = __float_as_uint(ra), b1 = __float_as_uint(rb);
    const float sa = ra - __uint_as_float(a1 & 0xffff0000u), sb = rb - __uint_as_float(b1 & 0xffff0000u);
    p0 = __builtin_amdgcn_perm(b, a, 0x07060302u);
    p1 = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

// one float32 value -> its three bf16 terms
__device__ __forceinline__ void split3x1(float x, unsigned short &h0, unsigned short &h1, unsigned short &h2) {
#pragma clang fp contract(off)
    const unsigned xb = __float_as_uint(x);
    const float r1 = x - __uint_as_float(xb & 0xffff0000u);
    const unsigned r1b = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(r1b & 0xffff0000u);
    h0 = (unsigned short)(xb >> 16);
    h1 = (unsigned short)(r1b >> 16);
    h2 = (unsigned short)(__float_as_uint(r2) >> 16);
}

// The six products of one K = 32 step for one (row tile, column tile): small terms into `lo`, t0 x t0 into `hi`.
__device__ __forceinline__ void mfma_split6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 &hi, f32x4 &lo) {
    lo = mfma_bf16(a[2], b[0], lo);
    lo = mfma_bf16(a[1], b[1], lo);
    lo = mfma_bf16(a[0], b[2], lo);
    lo = mfma_bf16(a[1], b[0], lo);
    lo = mfma_bf16(a[0], b[1], lo);
    hi = mfma_bf16(a[0], b[0], hi);
}
#endif

}  // namespace vadx
