/* cabi_silero.c -- a plain C (C99) client of libvadx.so: no Python, no torch.  It is what a native binding of the
 * reference's Silero seam (OnnxWrapper / get_speech_timestamps, Silero/modeling_modified/utils_vad.py:116-119,
 * 350-372, 374-476) would do: pack the weights once, upload, run whole clips, segment on the device.
 *
 *   cabi_silero weights.bin audio.bin B N out.bin [gain [unchecked]]
 * weights.bin: the float32 tensors of vadx_silero_weights_host in declaration order; audio.bin: f32 [B][N] (multiplied by `gain`, default 1);
 * out.bin: probs f32 [B][T] | counts int32 [B] | segments int64 [B][CAP][2].
 * The default arithmetic of the C ABI (cfg NULL / VADX_ARITH_AUTO = F16X2) has fp16's exponent range, so a caller runs the RANGE PROTOCOL of
 * include/vadx.h: after the launches, vadx_silero_range_flag (8 bytes + one stream synchronisation); a non-zero flag means some activation
 * left the fp16 range and the batch is recomputed on VADX_ARITH_BF16X3, whose terms have float32's range.  With a third extra argument
 * ("unchecked") the client skips the protocol on purpose: the flagged clips' scores must then be NaN, never plausible numbers.
 * Built and driven by tests/test_gpu_cabi_c.py, which compares out.bin with the Python host path bit for bit. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "vadx.h"

#define CAP 32
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_VADX(x) do { int rc_ = (x); if (rc_ != VADX_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, vadx_last_error()); return 3; } } while (0)

static float *read_floats(const char *path, size_t n) {
    FILE *f = fopen(path, "rb");
    float *p = (float *)malloc(n * sizeof(float));
    if (!f || !p || fread(p, sizeof(float), n, f) != n) { fprintf(stderr, "cannot read %zu floats from %s\n", n, path); exit(1); }
    fclose(f);
    return p;
}

int main(int argc, char **argv) {
    if (argc < 6 || argc > 8) { fprintf(stderr, "usage: %s weights.bin audio.bin B N out.bin [gain [unchecked]]\n", argv[0]); return 1; }
    const float gain = argc > 6 ? (float)atof(argv[6]) : 1.0f;
    const int unchecked = argc > 7;
    const int B = atoi(argv[3]);
    const long long N = atoll(argv[4]);
    const int T = (int)((N + 511) / 512);
    static const size_t wsz[15] = {258 * 256, 128 * 129 * 3, 64 * 128 * 3, 64 * 64 * 3, 128 * 64 * 3, 128, 64, 64, 128,
                                   512 * 128, 512 * 128, 512, 512, 128, 1};
    size_t total = 0, off[15];
    for (int i = 0; i < 15; ++i) { off[i] = total; total += wsz[i]; }
    float *w = read_floats(argv[1], total);
    vadx_silero_weights_host hw;
    hw.stft_basis = w + off[0];
    for (int i = 0; i < 4; ++i) { hw.enc_w[i] = w + off[1 + i]; hw.enc_b[i] = w + off[5 + i]; }
    hw.lstm_w_ih = w + off[9]; hw.lstm_w_hh = w + off[10]; hw.lstm_b_ih = w + off[11]; hw.lstm_b_hh = w + off[12];
    hw.dec_w = w + off[13]; hw.dec_b = w + off[14];

    if (vadx_abi_version() != VADX_ABI_VERSION) { fprintf(stderr, "unexpected ABI version\n"); return 1; }
    const size_t npk = vadx_silero_packed_floats();
    float *pk_host = (float *)malloc(npk * sizeof(float));
    CHECK_VADX(vadx_silero_pack_host(&hw, pk_host));

    float *audio_host = read_floats(argv[2], (size_t)B * (size_t)N);
    for (size_t e = 0; e < (size_t)B * (size_t)N; ++e) audio_host[e] *= gain;
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    float *pk, *audio, *probs;
    void *ws;
    int64_t *lens, *segs;
    int32_t *counts;
    const size_t ws_bytes = vadx_silero_workspace_bytes(B, T);
    CHECK_HIP(hipMalloc((void **)&pk, npk * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&audio, (size_t)B * N * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&probs, (size_t)B * T * sizeof(float)));
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    CHECK_HIP(hipMalloc((void **)&lens, B * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&segs, (size_t)B * CAP * 2 * sizeof(int64_t)));
    CHECK_HIP(hipMalloc((void **)&counts, B * sizeof(int32_t)));
    int64_t *lens_host = (int64_t *)malloc(B * sizeof(int64_t));
    for (int b = 0; b < B; ++b) lens_host[b] = N;
    CHECK_HIP(hipMemcpyAsync(pk, pk_host, npk * sizeof(float), hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(audio, audio_host, (size_t)B * N * sizeof(float), hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(lens, lens_host, B * sizeof(int64_t), hipMemcpyHostToDevice, stream));

    /* the two library calls of the hot path, stream-ordered behind the uploads; cfg NULL = VADX_ARITH_AUTO = F16X2 ... */
    CHECK_VADX(vadx_silero_clips(pk, audio, B, N, N, probs, NULL, ws, ws_bytes, stream, NULL));
    /* ... whose range protocol every caller of the default arithmetic runs: flag read (and cleared), recompute on BF16X3 if raised */
    int fallbacks = 0;
    if (!unchecked) {
        uint32_t flag = 0;
        float amax = 0.f;
        CHECK_VADX(vadx_silero_range_flag(pk, 1, &flag, &amax, stream));
        if (flag) {
            vadx_silero_cfg exact = {VADX_ARITH_BF16X3, {0, 0, 0}};
            CHECK_VADX(vadx_silero_clips(pk, audio, B, N, N, probs, NULL, ws, ws_bytes, stream, &exact));
            fallbacks = 1;
            fprintf(stderr, "range flag %u (largest |activation| %g): batch recomputed on VADX_ARITH_BF16X3\n", flag, (double)amax);
        }
    }
    vadx_silero_seg_params prm = {0.5, -1.0, 16000, 250.0, 1e30, 100.0, 30.0, 98.0, 1};
    CHECK_VADX(vadx_silero_segments(probs, B, T, lens, &prm, segs, counts, CAP, stream));

    /* error path: a sample rate the reference wrapper rejects must come back as VADX_EINVAL with its message */
    if (vadx_silero_step(pk, audio, audio, 44100, 1, probs, probs, ws, ws_bytes, stream, NULL) != VADX_EINVAL) {
        fprintf(stderr, "sr=44100 was not rejected\n");
        return 4;
    }

    float *probs_host = (float *)malloc((size_t)B * T * sizeof(float));
    int32_t *counts_host = (int32_t *)malloc(B * sizeof(int32_t));
    int64_t *segs_host = (int64_t *)malloc((size_t)B * CAP * 2 * sizeof(int64_t));
    CHECK_HIP(hipMemcpyAsync(probs_host, probs, (size_t)B * T * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(counts_host, counts, B * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(segs_host, segs, (size_t)B * CAP * 2 * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    FILE *f = fopen(argv[5], "wb");
    if (!f) return 1;
    fwrite(probs_host, sizeof(float), (size_t)B * T, f);
    fwrite(counts_host, sizeof(int32_t), B, f);
    fwrite(segs_host, sizeof(int64_t), (size_t)B * CAP * 2, f);
    fclose(f);
    size_t n_nan = 0;
    for (size_t e = 0; e < (size_t)B * T; ++e) n_nan += probs_host[e] != probs_host[e];
    printf("ok: %d clips x %d windows, first clip %d segments, range fallbacks %d, NaN scores %zu\n", B, T, counts_host[0], fallbacks, n_nan);
    return 0;
}
