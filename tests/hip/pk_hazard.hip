// Test-only build (tests/hip/libvadx_testhooks.so): standalone reproducer for the packed-f32 wrong-sum finding of DESIGN.md section 4e.
//
// silero_encode_h2_kernel (csrc/silero_h2.hip, phase 1) forms e = x[n] + x[256 - n], o = x[n] - x[256 - n] from samples it has just read out
// of LDS: xa ascending, xb DESCENDING.  Left to the compiler those sums became `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (the second
// source's halves swapped) and in round 5 that build returned wrong sums in about every second tile once two workgroups (four waves per
// SIMD) shared a CU.  This kernel keeps that phase and nothing else of the encoder: the staged window X f32 [16 clips][642] (even / odd
// sample planes), the same thread -> sample map, the same two barriers, then the pair sums three ways --
//   variant 0  scalar v_add_f32 / v_sub_f32 (inline asm: what the product kernel now does)
//   variant 1  v_pk_add_f32 with the CROSS swizzle, spelled in inline asm on the register pairs the LDS reads returned: (a) destination in
//              registers of its own, (b) destination = the swizzled source's pair, (c) destination = the plain source's pair
//   variant 2  plain C (whatever the compiler forms; tools/pk_scan.py on the test-hook library says what it formed)
// and compares 1 and 2 with 0 bit for bit in the kernel.  Between tiles a burst of fp16 MFMAs fed from LDS keeps the matrix pipe and the LDS
// busy the way the encoder's GEMM phases do, so that co-resident workgroups sit in different phases.  `lds_bytes` decides how many workgroups
// share a CU (<= 80 KB: two, i.e. four waves per SIMD; more: one).
// NOT part of the product ABI (include/vadx.h) and not linked into libvadx.so.
#include "../../voice-activity-detection-vad-onnx_amd/csrc/common.h"
#include "../../voice-activity-detection-vad-onnx_amd/csrc/split2.h"

using namespace vadx;

namespace {
constexpr int PK_LDM = 642, PK_ODD = 322;
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512, 4) void pk_hazard_kernel(const float *__restrict__ audio, long long row_stride, int tiles_per_wg, int mfma_burst,
                                                          unsigned *__restrict__ mism, float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *X = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned bad_pk = 0, bad_c = 0, bad_inb = 0, bad_ina = 0, bad_war = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int tl = 0; tl < tiles_per_wg; ++tl) {
        const long long base = ((long long)blockIdx.x * tiles_per_wg + tl) * 512;
        {   // staging as the encoder's fast path: wave w stages clips 2 w, 2 w + 1
            f32x4 xv[2][3];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const float *src = audio + (long long)(2 * wave + k2) * row_stride + base;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int f = j < 2 ? lane + 64 * j : min(lane + 128, 143);
                    xv[k2][j] = *reinterpret_cast<const f32x4 *>(src + 4 * f);
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                float *row = X + (2 * wave + k2) * PK_LDM;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int f = lane + 64 * j;
                    if (j < 2 || lane < 16) {
                        const f32x4 v = xv[k2][j];
                        *reinterpret_cast<float2 *>(row + 2 * f) = float2{v[0], v[2]};
                        *reinterpret_cast<float2 *>(row + PK_ODD + 2 * f) = float2{v[1], v[3]};
                    }
                }
            }
        }
        __syncthreads();
        const int cls = wave >> 2, pc = tid & 15, pj = (tid >> 4) & 15;
        float xa[4][4], xb[4][4];
        {
            const float *row = X + pc * PK_LDM + (cls ? PK_ODD : 1) + 4 * pj, *rowb = X + pc * PK_LDM + (cls ? PK_ODD : 0) + 127 - 4 * pj;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int k = 0; k < 4; ++k) { xa[f][k] = row[64 * f + k]; xb[f][k] = rowb[64 * f - k]; }
        }
        __syncthreads();          // as in the encoder: every sample is in registers, the planes may overwrite X
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            float e0[4], o0[4], e2[4], o2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0[k]) : "v"(xa[f][k]), "v"(xb[f][k]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(o0[k]) : "v"(xa[f][k]), "v"(xb[f][k]));
                e2[k] = xa[f][k] + xb[f][k];
                o2[k] = xa[f][k] - xb[f][k];
            }
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                // the register pairs as the descending read leaves them: (xb[k + 1], xb[k]) -- the cross swizzle swaps them back
                const f32x2_t a = {xa[f][k], xa[f][k + 1]}, b = {xb[f][k + 1], xb[f][k]};
                f32x2_t e1, o1;
                // 1a: destination in registers of its own (early clobber: never one of the sources)
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(e1) : "v"(a), "v"(b));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(o1) : "v"(a), "v"(b));
                bad_pk += (__float_as_uint(e1[0]) != __float_as_uint(e0[k])) + (__float_as_uint(e1[1]) != __float_as_uint(e0[k + 1]));
                bad_pk += (__float_as_uint(o1[0]) != __float_as_uint(o0[k])) + (__float_as_uint(o1[1]) != __float_as_uint(o0[k + 1]));
                // 1b: destination = the SWIZZLED source's pair (the low result lands in the register the high result still has to read)
                f32x2_t eb = b, ob = b;
                asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(eb) : "v"(a));
                asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(ob) : "v"(a));
                bad_inb += (__float_as_uint(eb[0]) != __float_as_uint(e0[k])) + (__float_as_uint(eb[1]) != __float_as_uint(e0[k + 1]));
                bad_inb += (__float_as_uint(ob[0]) != __float_as_uint(o0[k])) + (__float_as_uint(ob[1]) != __float_as_uint(o0[k + 1]));
                // 1d: the sources are overwritten by the instructions that follow at once (write-after-read: the split's conversions, as the
                //     compiler schedules them behind the sums in the encoder), in fixed registers so that the pattern is what is written here
                f32x2_t ed;
                asm volatile("v_mov_b32 v120, %1\n\tv_mov_b32 v121, %2\n\tv_mov_b32 v122, %3\n\tv_mov_b32 v123, %4\n\ts_nop 4\n\t"
                             "v_pk_add_f32 %0, v[120:121], v[122:123] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                             "v_cvt_pk_f16_f32 v122, %1, %2\n\tv_cvt_f32_f16_e32 v123, v122\n\tv_cvt_pk_f16_f32 v120, %3, %4\n\tv_mov_b32 v121, %1"
                             : "=&v"(ed) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]) : "v120", "v121", "v122", "v123");
                bad_war += (__float_as_uint(ed[0]) != __float_as_uint(e0[k])) + (__float_as_uint(ed[1]) != __float_as_uint(e0[k + 1]));
                // 1c: destination = the plain source's pair (control)
                f32x2_t ea = a, oa = a;
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(ea) : "v"(b));
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(oa) : "v"(b));
                bad_ina += (__float_as_uint(ea[0]) != __float_as_uint(e0[k])) + (__float_as_uint(ea[1]) != __float_as_uint(e0[k + 1]));
                bad_ina += (__float_as_uint(oa[0]) != __float_as_uint(o0[k])) + (__float_as_uint(oa[1]) != __float_as_uint(o0[k + 1]));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                bad_c += (__float_as_uint(e2[k]) != __float_as_uint(e0[k])) + (__float_as_uint(o2[k]) != __float_as_uint(o0[k]));
            // the planes: the sums overwrite X as fp16 pairs, as the encoder's operand planes do (keeps the LDS write traffic of the phase)
            float am = 0.f;
            u32x2 p0, p1;
            split2x4(f32x4{e2[0], e2[1], e2[2], e2[3]}, p0, p1, am);
            unsigned char *d = smem + (cls * 4) * 8192 + (pj >> 1) * 1024 + (16 * f + pc) * 16 + 8 * (pj & 1);
            *reinterpret_cast<u32x2 *>(d) = p0;
            *reinterpret_cast<u32x2 *>(d + 8192) = p1;
            split2x4(f32x4{o2[0], o2[1], o2[2], o2[3]}, p0, p1, am);
            *reinterpret_cast<u32x2 *>(d + 2 * 8192) = p0;
            *reinterpret_cast<u32x2 *>(d + 3 * 8192) = p1;
        }
        __syncthreads();
        // a GEMM phase's worth of matrix + LDS work on the planes just written
        const int q = lane >> 4, i = lane & 15;
#pragma unroll 1
        for (int m = 0; m < mfma_burst; ++m) {
            const unsigned char *bs = smem + ((m & 7) * 8192) + (q + 4 * ((m >> 3) & 1)) * 1024 + (16 * (wave >> 1) + i) * 16;
            const f16x8 b0 = *reinterpret_cast<const f16x8 *>(bs);
            acc = mfma_f16(b0, b0, acc);
        }
        __syncthreads();
    }
    if (bad_pk) atomicAdd(mism + 0, bad_pk);
    if (bad_c) atomicAdd(mism + 1, bad_c);
    if (bad_inb) atomicAdd(mism + 2, bad_inb);
    if (bad_ina) atomicAdd(mism + 3, bad_ina);
    if (bad_war) atomicAdd(mism + 4, bad_war);
    if (acc[0] == 12345.678f) sink[0] = acc[1];        // keeps the burst alive
}
// ---- second probe: is a PARTIAL s_waitcnt lgkmcnt(N) sound when ds_read2_b64 and ds_read2_b32 are mixed?
// The failing encoder builds differ from the passing one in one more way than the packed adds: the compiler starts the sums of the first
// frames while later LDS reads are still in flight, behind `s_waitcnt lgkmcnt(4 / 3 / 2 / 1)`, which is only sound if LDS returns data in issue
// order.  This kernel issues the encoder's six reads (two 16-byte ds_read2_b64 of the descending run, four 8-byte ds_read2_b32 of the
// ascending one) into sentinel-filled FIXED registers, snapshots what each partial wait claims has arrived, then waits for everything and
// compares: a snapshot that still holds the sentinel (or anything but the final value) means the counter released the wave early.
// MODE 1: the same reads, and behind each partial wait the encoder's own consumer -- v_pk_add_f32 with the cross swizzle on the registers that
// have just arrived (a = read #1 / #4, b = the HIGH / LOW pair of the 16-byte read #2, as the failing builds had them) -- against the same
// packed add and scalar adds issued after s_waitcnt lgkmcnt(0): [0] early packed sum != scalar, [1] second early packed sum != scalar,
// [2] late packed sum != scalar.
template <int MODE>
__global__ __launch_bounds__(512, 4) void lds_order_kernel(const float *__restrict__ audio, long long row_stride, int tiles_per_wg, int mfma_burst,
                                                          unsigned *__restrict__ mism, float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *X = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned bad[4] = {0, 0, 0, 0};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int tl = 0; tl < tiles_per_wg; ++tl) {
        const long long base = ((long long)blockIdx.x * tiles_per_wg + tl) * 512;
        {
            f32x4 xv[2][3];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const float *src = audio + (long long)(2 * wave + k2) * row_stride + base;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int f = j < 2 ? lane + 64 * j : min(lane + 128, 143);
                    xv[k2][j] = *reinterpret_cast<const f32x4 *>(src + 4 * f);
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                float *row = X + (2 * wave + k2) * PK_LDM;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int f = lane + 64 * j;
                    if (j < 2 || lane < 16) {
                        const f32x4 v = xv[k2][j];
                        *reinterpret_cast<float2 *>(row + 2 * f) = float2{v[0], v[2]};
                        *reinterpret_cast<float2 *>(row + PK_ODD + 2 * f) = float2{v[1], v[3]};
                    }
                }
            }
        }
        __syncthreads();
        const int cls = wave >> 2, pc = tid & 15, pj = (tid >> 4) & 15;
        // byte addresses as the encoder forms them: ascending run at row, descending run ending at rowb
        const unsigned a_up = (unsigned)((pc * PK_LDM + (cls ? PK_ODD : 1) + 4 * pj) * 4);
        const unsigned a_dn = (unsigned)((pc * PK_LDM + (cls ? PK_ODD : 0) + 127 - 4 * pj - 3 - 124) * 4);      // = rowb - 127 floats: offset0:62 -> rowb[-3..-2]
        unsigned s4a = 0, s2a = 0;
        if (MODE == 0) {
        unsigned s4b, s4c, s2b, s1a, f4a, f4b, f4c, f2a, f2b, f1a;
        asm volatile(
            "v_mov_b32 v100, 0xdeadbeef\n\tv_mov_b32 v101, 0xdeadbeef\n\tv_mov_b32 v102, 0xdeadbeef\n\tv_mov_b32 v103, 0xdeadbeef\n\t"
            "v_mov_b32 v104, 0xdeadbeef\n\tv_mov_b32 v105, 0xdeadbeef\n\tv_mov_b32 v106, 0xdeadbeef\n\tv_mov_b32 v107, 0xdeadbeef\n\t"
            "v_mov_b32 v108, 0xdeadbeef\n\tv_mov_b32 v109, 0xdeadbeef\n\tv_mov_b32 v110, 0xdeadbeef\n\tv_mov_b32 v111, 0xdeadbeef\n\t"
            "v_mov_b32 v112, 0xdeadbeef\n\tv_mov_b32 v113, 0xdeadbeef\n\tv_mov_b32 v114, 0xdeadbeef\n\tv_mov_b32 v115, 0xdeadbeef\n\t"
            "s_nop 4\n\t"
            "ds_read2_b32 v[100:101], %12 offset1:1\n\t"                    // #1  (the natural build's order)
            "ds_read2_b64 v[102:105], %13 offset0:62 offset1:63\n\t"        // #2
            "ds_read2_b64 v[106:109], %13 offset0:94 offset1:95\n\t"        // #3
            "ds_read2_b32 v[110:111], %12 offset0:2 offset1:3\n\t"          // #4
            "ds_read2_b32 v[112:113], %12 offset0:64 offset1:65\n\t"        // #5
            "ds_read2_b32 v[114:115], %12 offset0:66 offset1:67\n\t"        // #6
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v104\n\tv_mov_b32 %2, v105\n\t"        // #1, #2 claimed
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mov_b32 %3, v110\n\tv_mov_b32 %4, v102\n\t"                               // #3, #4 claimed (#2's low half read here, as the sums do)
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_mov_b32 %5, v112\n\t"                                                       // #5 claimed
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 4\n\t"
            "v_mov_b32 %6, v100\n\tv_mov_b32 %7, v104\n\tv_mov_b32 %8, v105\n\tv_mov_b32 %9, v110\n\tv_mov_b32 %10, v102\n\tv_mov_b32 %11, v112"
            : "=&v"(s4a), "=&v"(s4b), "=&v"(s4c), "=&v"(s2a), "=&v"(s2b), "=&v"(s1a), "=&v"(f4a), "=&v"(f4b), "=&v"(f4c), "=&v"(f2a), "=&v"(f2b), "=&v"(f1a)
            : "v"(a_up), "v"(a_dn)
            : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");
        bad[0] += (s4a != f4a) + (s4b != f4b) + (s4c != f4c);
        bad[1] += (s2a != f2a) + (s2b != f2b);
        bad[2] += (s1a != f1a);
        bad[3] += (f4a == 0xdeadbeefu);            // (never: the final values are data)
        } else {
            unsigned e0, e1, g0, g1, l0, l1, m0, m1, r0, r1, r2, r3;
            asm volatile(
                "v_mov_b32 v100, 0xdeadbeef\n\tv_mov_b32 v101, 0xdeadbeef\n\tv_mov_b32 v102, 0xdeadbeef\n\tv_mov_b32 v103, 0xdeadbeef\n\t"
                "v_mov_b32 v104, 0xdeadbeef\n\tv_mov_b32 v105, 0xdeadbeef\n\tv_mov_b32 v106, 0xdeadbeef\n\tv_mov_b32 v107, 0xdeadbeef\n\t"
                "v_mov_b32 v108, 0xdeadbeef\n\tv_mov_b32 v109, 0xdeadbeef\n\tv_mov_b32 v110, 0xdeadbeef\n\tv_mov_b32 v111, 0xdeadbeef\n\t"
                "v_mov_b32 v112, 0xdeadbeef\n\tv_mov_b32 v113, 0xdeadbeef\n\tv_mov_b32 v114, 0xdeadbeef\n\tv_mov_b32 v115, 0xdeadbeef\n\t"
                "s_nop 4\n\t"
                "ds_read2_b32 v[100:101], %12 offset1:1\n\t"
                "ds_read2_b64 v[102:105], %13 offset0:62 offset1:63\n\t"
                "ds_read2_b64 v[106:109], %13 offset0:94 offset1:95\n\t"
                "ds_read2_b32 v[110:111], %12 offset0:2 offset1:3\n\t"
                "ds_read2_b32 v[112:113], %12 offset0:64 offset1:65\n\t"
                "ds_read2_b32 v[114:115], %12 offset0:66 offset1:67\n\t"
                "s_waitcnt lgkmcnt(4)\n\t"
                "v_pk_add_f32 v[116:117], v[100:101], v[104:105] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                "s_waitcnt lgkmcnt(2)\n\t"
                "v_pk_add_f32 v[118:119], v[110:111], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "s_nop 7\n\t"
                "v_pk_add_f32 v[120:121], v[100:101], v[104:105] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                "v_pk_add_f32 v[122:123], v[110:111], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                "v_add_f32 %8, v100, v105\n\tv_add_f32 %9, v101, v104\n\tv_add_f32 %10, v110, v103\n\tv_add_f32 %11, v111, v102\n\t"
                "s_nop 4\n\t"
                "v_mov_b32 %0, v116\n\tv_mov_b32 %1, v117\n\tv_mov_b32 %2, v118\n\tv_mov_b32 %3, v119\n\t"
                "v_mov_b32 %4, v120\n\tv_mov_b32 %5, v121\n\tv_mov_b32 %6, v122\n\tv_mov_b32 %7, v123"
                : "=&v"(e0), "=&v"(e1), "=&v"(g0), "=&v"(g1), "=&v"(l0), "=&v"(l1), "=&v"(m0), "=&v"(m1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                : "v"(a_up), "v"(a_dn)
                : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
                  "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123");
            bad[0] += (e0 != r0) + (e1 != r1);
            bad[1] += (g0 != r2) + (g1 != r3);
            bad[2] += (l0 != r0) + (l1 != r1) + (m0 != r2) + (m1 != r3);
            s4a = e0; s2a = g0;
        }
        __syncthreads();
        // keep the planes' write traffic and a GEMM phase's worth of matrix + LDS work between tiles, as pk_hazard_kernel
        *reinterpret_cast<unsigned *>(smem + (tid * 4)) = s4a ^ s2a;
        __syncthreads();
        const int q = lane >> 4, i = lane & 15;
#pragma unroll 1
        for (int m = 0; m < mfma_burst; ++m) {
            const unsigned char *bs = smem + ((m & 7) * 8192) + (q + 4 * ((m >> 3) & 1)) * 1024 + (16 * (wave >> 1) + i) * 16;
            const f16x8 b0 = *reinterpret_cast<const f16x8 *>(bs);
            acc = mfma_f16(b0, b0, acc);
        }
        __syncthreads();
    }
    for (int k = 0; k < 4; ++k)
        if (bad[k]) atomicAdd(mism + k, bad[k]);
    if (acc[0] == 12345.678f) sink[0] = acc[1];
}
}  // namespace

/* As vadx_test_pk_hazard; mode 0 / 1 as above.  Mode 0 -- mism: device unsigned[4], zeroed by the caller: values that a partial s_waitcnt claimed had arrived and that differed from
 * what the register held after s_waitcnt lgkmcnt(0) -- [0] behind lgkmcnt(4) of 6 reads, [1] behind lgkmcnt(2), [2] behind lgkmcnt(1), [3] sentinel check. */
extern "C" int vadx_test_lds_order(const float *audio, long long row_stride, int nblocks, int tiles_per_wg, int mfma_burst, int lds_bytes,
                                   int mode, unsigned *mism, float *sink, void *stream) {
    VADX_REQUIRE(audio && mism && sink && nblocks > 0 && tiles_per_wg > 0 && lds_bytes >= 65536 && lds_bytes <= 160 * 1024, "vadx_test_lds_order: bad arguments");
    VADX_DYN_LDS(lds_order_kernel<0>, 160 * 1024);
    VADX_DYN_LDS(lds_order_kernel<1>, 160 * 1024);
    if (mode)
        hipLaunchKernelGGL(lds_order_kernel<1>, dim3(nblocks), dim3(512), lds_bytes, static_cast<hipStream_t>(stream), audio, row_stride, tiles_per_wg,
                           mfma_burst, mism, sink);
    else
        hipLaunchKernelGGL(lds_order_kernel<0>, dim3(nblocks), dim3(512), lds_bytes, static_cast<hipStream_t>(stream), audio, row_stride, tiles_per_wg,
                           mfma_burst, mism, sink);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}

/* audio: f32 [16][row_stride] device rows of at least nblocks * tiles_per_wg * 512 + 576 samples; mism: device unsigned[5], zeroed by the caller: sums that
 * differ from the scalar sums -- [0] forced cross-swizzled v_pk_add_f32, own destination, [1] the plain-C form, [2] forced, destination = swizzled
 * source, [3] forced, destination = plain source, [4] forced, sources overwritten by the next instructions. */
extern "C" int vadx_test_pk_hazard(const float *audio, long long row_stride, int nblocks, int tiles_per_wg, int mfma_burst, int lds_bytes,
                                   unsigned *mism, float *sink, void *stream) {
    VADX_REQUIRE(audio && mism && sink && nblocks > 0 && tiles_per_wg > 0 && lds_bytes >= 65536 && lds_bytes <= 160 * 1024, "vadx_test_pk_hazard: bad arguments");
    VADX_DYN_LDS(pk_hazard_kernel, 160 * 1024);
    hipLaunchKernelGGL(pk_hazard_kernel, dim3(nblocks), dim3(512), lds_bytes, static_cast<hipStream_t>(stream), audio, row_stride, tiles_per_wg,
                       mfma_burst, mism, sink);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
