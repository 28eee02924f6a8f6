// split_scheme.h -- the two split-product arithmetics behind one interface, so that a kernel is written once and instantiated for both:
//   SchemeB3  bf16 x 3 (csrc/split3.h): operands split EXACTLY into three bf16 planes, six v_mfma_f32_16x16x32_bf16 per K = 32 step,
//             float32's exponent range;
//   SchemeH2  fp16 x 2 (csrc/split2.h): operands to one float32 ulp in two round-to-nearest fp16 planes (the second scaled by 2^11),
//             three v_mfma_f32_16x16x32_f16 per K = 32 step; values must stay inside the fp16 range -- every split feeds a running max |x|
//             (`amax`) that the kernel checks at its end (range flag in its packed blob).
// A scheme provides: NP (planes), frag (the MFMA operand vector), ld (one A fragment from global memory), products (the K = 32 step of an
// NT x MTT group of output tiles, small terms into `lo`, leading term into `hi`), join (hi, lo -> the float32 result), split4 (four
// consecutive k of one column -> NP 8-byte plane entries), put_host (one weight into the NP fragments of its (tile, chunk)).
#pragma once
#include "common.h"
#include "split2.h"
#include "split3.h"

namespace vadx {

constexpr int VADX_AR_F32 = 0, VADX_AR_B3 = 1, VADX_AR_H2 = 2;      // internal numbering of the arithmetics (include/vadx.h: VADX_ARITH_* - 1)
// NULL-safe mapping of a cfg's `arithmetic` field to the internal number; `dflt` = what VADX_ARITH_AUTO means for the entry point; -1 = invalid
inline int arith_internal(int a, int dflt) {
    return a == VADX_ARITH_AUTO ? dflt : (a == VADX_ARITH_F32 ? VADX_AR_F32 : (a == VADX_ARITH_BF16X3 ? VADX_AR_B3 : (a == VADX_ARITH_F16X2 ? VADX_AR_H2 : -1)));
}

struct SchemeB3 {
    static constexpr int NP = 3;
    static constexpr int ARITH = VADX_AR_B3;
    static constexpr bool RANGE_CHECK = false;
#if defined(__HIPCC__)
    typedef bf16x8 frag;
    static __device__ __forceinline__ frag ld(const float *f, int lane) { return ldq(f, lane); }
    static __device__ __forceinline__ frag lds(const unsigned char *p) { return *reinterpret_cast<const frag *>(p); }
    template <int NT, int MTT, bool A_IS_W>
    static __device__ __forceinline__ void products(const frag (&a)[NT][NP], const frag (&b)[MTT][NP], f32x4 (&hi)[NT][MTT], f32x4 (&lo)[NT][MTT]) {
        // six products per (n-tile, column tile), tiles innermost: consecutive MFMAs hit different accumulators
#define VADX_SCH_TERM(AP, BP, ACC)                                                                                             \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int mt = 0; mt < MTT; ++mt)                       \
        ACC[nt][mt] = A_IS_W ? mfma_bf16(a[nt][AP], b[mt][BP], ACC[nt][mt]) : mfma_bf16(b[mt][BP], a[nt][AP], ACC[nt][mt]);
        VADX_SCH_TERM(2, 0, lo) VADX_SCH_TERM(1, 1, lo) VADX_SCH_TERM(0, 2, lo) VADX_SCH_TERM(1, 0, lo) VADX_SCH_TERM(0, 1, lo) VADX_SCH_TERM(0, 0, hi)
#undef VADX_SCH_TERM
    }
    static __device__ __forceinline__ f32x4 join(const f32x4 hi, const f32x4 lo) { return hi + lo; }
    static __device__ __forceinline__ void split4(const f32x4 x, u32x2 (&p)[NP], float &) { split3x4(x, p[0], p[1], p[2]); }
#endif
    static void put_host(float *frags, int row, int k, float w, float &) { qfrag_put(frags, row, k, w); }
};

struct SchemeH2 {
    static constexpr int NP = 2;
    static constexpr int ARITH = VADX_AR_H2;
    static constexpr bool RANGE_CHECK = true;
#if defined(__HIPCC__)
    typedef f16x8 frag;
    static __device__ __forceinline__ frag ld(const float *f, int lane) { return ldh(f, lane); }
    static __device__ __forceinline__ frag lds(const unsigned char *p) { return *reinterpret_cast<const frag *>(p); }
    template <int NT, int MTT, bool A_IS_W>
    static __device__ __forceinline__ void products(const frag (&a)[NT][NP], const frag (&b)[MTT][NP], f32x4 (&hi)[NT][MTT], f32x4 (&lo)[NT][MTT]) {
#define VADX_SCH_TERM(AP, BP, ACC)                                                                                             \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int mt = 0; mt < MTT; ++mt)                       \
        ACC[nt][mt] = A_IS_W ? mfma_f16(a[nt][AP], b[mt][BP], ACC[nt][mt]) : mfma_f16(b[mt][BP], a[nt][AP], ACC[nt][mt]);
        VADX_SCH_TERM(1, 0, lo) VADX_SCH_TERM(0, 1, lo) VADX_SCH_TERM(0, 0, hi)
#undef VADX_SCH_TERM
    }
    static __device__ __forceinline__ f32x4 join(const f32x4 hi, const f32x4 lo) { return join2(hi, lo); }
    static __device__ __forceinline__ void split4(const f32x4 x, u32x2 (&p)[NP], float &amax) { split2x4(x, p[0], p[1], amax); }
#endif
    static void put_host(float *frags, int row, int k, float w, float &wmax) {
        const float a = hfrag_put(frags, row, k, w);
        if (!(a <= wmax)) wmax = a;          // (a NaN weight ends up in wmax too)
    }
};

#if defined(__HIPCC__)
// the end-of-kernel range check of a SchemeH2 kernel: `flag` = two words inside the kernel's packed blob [sticky flag, bits of the largest |x|]
__device__ __forceinline__ void range_flag_raise(const float *flag_words, float amax) {
    if (!(amax <= H_MAX)) {
        unsigned *fl = reinterpret_cast<unsigned *>(const_cast<float *>(flag_words));
        atomicOr(fl, 1u);
        atomicMax(fl + 1, __float_as_uint(amax));
    }
}
#endif

}  // namespace vadx
