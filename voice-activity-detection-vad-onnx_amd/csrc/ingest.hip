// ingest.hip -- device-side audio ingest with the arithmetic of the reference drivers' pydub chain
//   AudioSegment.from_file(p).set_channels(1).set_frame_rate(16000)        (e.g. FSMN/Inference_FSMN_VAD_ONNX.py:68)
// which is the stdlib audioop: tomono(0.5, 0.5) then ratecv (linear interpolation, no filter), 16-bit samples.
//
// audioop.ratecv walks the input with an integer phase d (rates reduced by their gcd to I -> O, d starts at -O): each
// input frame adds O, each output frame is emitted while d >= 0 as
//     cur_o = (int)((prev * d + cur * (O - d)) / O)   on samples scaled by 2^16, stored as cur_o >> 16
// and subtracts I.  In closed form output e uses m = ceil(e*I/O) + 1 consumed frames, d = (m-1)*O - e*I in [0, O),
// prev = x[m-2] (0 before the first frame), cur = x[m-1]; for O < 65536 the truncate-then-shift equals the floor
// division  out[e] = floor((prev*d + cur*(O-d)) / O)  on the 16-bit values.  tomono: floor((L + R) * 0.5).
// One thread per output sample: pure HBM-bound integer work.
#include "common.h"

namespace vadx {
namespace ingest {

__device__ __forceinline__ long long floordiv(long long a, long long b) {      // b > 0
    const long long qd = a / b;
    return (a % b != 0 && a < 0) ? qd - 1 : qd;
}

__global__ void ingest_kernel(const int16_t *__restrict__ src, long long src_stride, int channels, long long frames_in,
                              long long I, long long O, int16_t *__restrict__ dst, long long dst_stride, long long frames_out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= frames_out) return;
    const int16_t *row = src + (long long)blockIdx.y * src_stride;
    auto mono = [&](long long k) -> long long {
        if (k < 0) return 0;
        if (channels == 1) return row[k];
        return ((long long)row[2 * k] + (long long)row[2 * k + 1]) >> 1;       // floor((L + R) / 2)
    };
    const long long m = (e * I + O - 1) / O + 1, d = (m - 1) * O - e * I;
    const long long prev = mono(m - 2), cur = mono(m - 1);
    dst[(long long)blockIdx.y * dst_stride + e] = (int16_t)floordiv(prev * d + cur * (O - d), O);
}

static long long gcd(long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a; }

}  // namespace ingest
}  // namespace vadx

using namespace vadx::ingest;

extern "C" int64_t vadx_ingest_out_frames(int64_t frames_in, int in_rate, int out_rate) {
    if (frames_in <= 0 || in_rate <= 0 || out_rate <= 0) return 0;
    const long long g = gcd(in_rate, out_rate), I = in_rate / g, O = out_rate / g;
    return (int64_t)(((long long)frames_in - 1) * O / I + 1);
}

extern "C" int vadx_ingest_pcm16(const int16_t *src, int64_t src_stride, int channels, int64_t frames_in, int in_rate,
                                 int out_rate, int16_t *dst, int64_t dst_stride, int batch, void *stream) {
    VADX_REQUIRE(src && dst && batch > 0 && frames_in > 0, "vadx_ingest_pcm16: bad argument");
    VADX_REQUIRE(channels == 1 || channels == 2, "vadx_ingest_pcm16: 1 or 2 interleaved channels (got %d)", channels);
    VADX_REQUIRE(in_rate > 0 && out_rate > 0, "vadx_ingest_pcm16: rates must be positive");
    const long long g = gcd(in_rate, out_rate), I = in_rate / g, O = out_rate / g;
    VADX_REQUIRE(O < 65536 && I < (1LL << 31), "vadx_ingest_pcm16: reduced rates %lld -> %lld out of range", I, O);
    const int64_t frames_out = vadx_ingest_out_frames(frames_in, in_rate, out_rate);
    VADX_REQUIRE(src_stride >= frames_in * channels && dst_stride >= frames_out, "vadx_ingest_pcm16: row strides too small");
    hipLaunchKernelGGL(ingest_kernel, dim3((unsigned)((frames_out + 255) / 256), (unsigned)batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, (long long)src_stride, channels, (long long)frames_in, I, O, dst,
                       (long long)dst_stride, (long long)frames_out);
    VADX_HIP_TRY(hipGetLastError());
    return VADX_OK;
}
