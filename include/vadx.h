/* vadx.h -- C ABI of the MI355X-native batched VAD engine (libvadx.so).
 *
 * The reference (DakeQQ/Voice-Activity-Detection-VAD-ONNX) has no native FFI: its drop-in seam is
 * the ONNX-Runtime Python session object (`session.run(output_names, feeds)`), plus for Silero the
 * `OnnxWrapper` object.  Each entry point below is what a binding for that seam calls; the
 * comment on each names the reference interface it replaces (file:line under the reference root).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / hip types in signatures (`stream` is a hipStream_t
 *     passed as void*; NULL = the default stream);
 *   - all data pointers are DEVICE pointers unless the name ends in `_host`;
 *   - caller-owned buffers, no hidden allocation, stream-ordered (no device sync inside);
 *   - return 0 on success, a negative VADX_E* code otherwise; `vadx_last_error()` gives the text.
 */
#ifndef VADX_H
#define VADX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VADX_OK          0
#define VADX_EINVAL     -1   /* bad argument (shape, NULL pointer, unsupported sample rate ...) */
#define VADX_ENOSPACE   -2   /* workspace / output capacity too small */
#define VADX_EHIP       -3   /* a HIP runtime call failed */

/* THE ABI number: the library (csrc/capi.hip), the ctypes binding (vadx._lib.ABI_VERSION, parsed from this line),
 * the C client (tests/c/cabi_silero.c) and __graft_entry__.build() all read it from here and nowhere else.
 * 3: vadx_frontend_cfg grew `fold`; vadx_frontend_fold_kind, vadx_dfsmn_cfb_*, _lstm_t_ex, _ft_repack (round 3).
 * 4: vadx_silero_encoder_mode, vadx_gemm_mode; the Silero / FSMN packed blobs grew the bf16 x 3 weight fragments (round 4).
 * 5: vadx_dfsmn_cfb_weights grew fwd_tbl_q / inv_tbl_q (the struct must be zero-initialised at its ABI-5 size: an ABI-4 caller's shorter
 *    struct would have the new pointers read past its end -- the version check is the guard); vadx_stream_vadpost* (round 4).
 * 6: the arithmetic of an entry point comes WITH THE CALL (VADX_ARITH_* in a cfg / dims struct) instead of process-wide switches:
 *    vadx_silero_encoder_mode and vadx_gemm_mode are gone, every Silero launch takes a trailing `const vadx_silero_cfg *`;
 *    VADX_ARITH_F16X2 (fp16 x 2 split products) + vadx_silero_range_flag; the Silero packed blob grew the fp16 fragments;
 *    vadx_sepconv_block / vadx_marblenet_block2 / vadx_marblenet_tail take a trailing `const vadx_marblenet_cfg *` (NULL = float32 MFMAs),
 *    vadx_frag_h2_host / vadx_frag_h2_floats; vadx_dfsmn_lstm_f and vadx_dfsmn_lstm_t_ex take (arithmetic, range_flag) (round 5).
 * 7: no signature changed; three behaviours did (round 6).  (a) VADX_ARITH_AUTO of vadx_fsmn_dims / vadx_firered_cfg is BF16X3 (float32's
 *    exponent range: nothing for the caller to check); F16X2 is an explicit request there, by callers that read the range flag.  Silero's AUTO
 *    stays F16X2, and (b) a Silero workgroup whose activations left the fp16 range writes NaN instead of its gate pre-activations, so that the
 *    scores of those clips are NaN for a caller that never reads vadx_silero_range_flag (tests/c/cabi_silero.c reads it).  (c) The pack_host
 *    functions rebalance chains of affine layers by exact powers of two when a weight tensor sits outside [2^-10, 2^7) (csrc/rebalance.h) and
 *    refuse F16X2 for a blob that keeps a weight tensor wholly below 2^-14. */
#define VADX_ABI_VERSION 7

/* Arithmetic of the products whose one operand is a constant (every weight matrix, every DFT table) -- float32 RESULTS in all of them:
 *   F32     v_mfma_f32_16x16x4_f32 on the float32 operands themselves;
 *   BF16X3  operands split EXACTLY into three bf16 terms, six v_mfma_f32_16x16x32_bf16 per K = 32 step (csrc/split3.h): float32-class
 *           accuracy at 6/16 of the matrix time, float32's exponent range;
 *   F16X2   operands represented to one float32 ulp by two round-to-nearest fp16 terms, three v_mfma_f32_16x16x32_f16 per K = 32 step
 *           (csrc/split2.h): float32-class accuracy at 3/16 of the matrix time; activations must stay inside the fp16 range (|x| <= 65504),
 *           which the kernels check (vadx_silero_range_flag / vadx_fsmn_range_flag / vadx_firered_range_flag: the caller reads the flag with
 *           the results and recomputes a flagged batch on BF16X3 -- the RANGE PROTOCOL; tests/c/cabi_silero.c shows it);
 *   AUTO    the library's default for the entry point (what a zero-initialised cfg selects). */
#define VADX_ARITH_AUTO   0
#define VADX_ARITH_F32    1
#define VADX_ARITH_BF16X3 2
#define VADX_ARITH_F16X2  3

int         vadx_abi_version(void);      /* == VADX_ABI_VERSION of the header the library was built from */
const char *vadx_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Silero
 * ------------------------------------------------------------------------------------------- */

/* Original-layout float32 weights on the HOST (same tensors as the ONNX initialisers):
 * stft_basis [258][256]; enc_w[i] [c_out][c_in][3], enc_b[i] [c_out] for the four encoder convs
 * (129->128 s1, 128->64 s2, 64->64 s2, 64->128 s1); LSTM cell [512][128] x2 + biases [512] x2 in
 * torch gate order (i,f,g,o); decoder conv [128] + bias [1]. */
typedef struct vadx_silero_weights_host {
    const float *stft_basis;
    const float *enc_w[4];
    const float *enc_b[4];
    const float *lstm_w_ih, *lstm_w_hh, *lstm_b_ih, *lstm_b_hh;
    const float *dec_w, *dec_b;
} vadx_silero_weights_host;

/* Per-call configuration of the Silero launches; NULL = all defaults.  Zero-initialise it. */
typedef struct vadx_silero_cfg {
    int32_t arithmetic;     /* VADX_ARITH_*: AUTO = F16X2 (run the range protocol: vadx_silero_range_flag below).  The three kernel sets read the same packed blob and write the same workspace, so the
                             * encoder and recurrent launches of one batch may even differ.  Replaces nothing in the reference: onnxruntime has
                             * one CPU kernel set. */
    int32_t reserved[3];    /* must be zero */
} vadx_silero_cfg;

/* Number of floats of the packed (kernel-layout) weight blob. */
size_t vadx_silero_packed_floats(void);
/* Repack on the host (init time only); the caller uploads `packed_host` to the device once.
 * Replaces: onnxruntime.InferenceSession(path) construction, Silero/modeling_modified/utils_vad.py:39. */
int vadx_silero_pack_host(const vadx_silero_weights_host *w, float *packed_host);

/* Scratch needed by the two calls below, in bytes. `steps` = windows per clip (1 for vadx_silero_step). */
size_t vadx_silero_workspace_bytes(int batch, int steps);

/* One ORT-boundary call, batched:
 *   feeds  {'input': f32 [B,576], 'state': f32 [2,B,128], 'sr': int64 scalar}
 *   fetches(out f32 [B,1], stateN f32 [2,B,128])
 * Replaces: self.session.run(None, ort_inputs), Silero/modeling_modified/utils_vad.py:116-119.
 * sr must be 16000 (the reference wrapper's '16k' model path, utils_vad.py:62-64). */
int vadx_silero_step(const float *packed, const float *input, const float *state, int64_t sr,
                     int batch, float *out, float *state_n,
                     void *workspace, size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg);

/* Whole clips, batched: audio f32 [B][row_stride] (first n_samples valid per row, +-1 scale),
 * zero state and zero context at t=0, last window zero-padded; probs f32 [B][T], T = ceil(n/512).
 * state_n (optional, may be NULL) receives the final [2,B,128].
 * Replaces the window loop of get_speech_timestamps, utils_vad.py:350,359-372, and
 * OnnxWrapper.audio_forward, utils_vad.py:130-146 (context carry :111-114,:123 is done in-kernel). */
int vadx_silero_clips(const float *packed, const float *audio, int batch, int64_t n_samples,
                      int64_t row_stride, float *probs, float *state_n,
                      void *workspace, size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg);

/* The two halves of vadx_silero_clips as separate launches (same buffers, same results), so a
 * harness can time the state-independent encoder (STFT conv + conv stack + W_ih, the dominant
 * kernel) and the recurrent kernel separately.  `steps` = ceil(n_samples/512). */
int vadx_silero_encode(const float *packed, const float *audio, int batch, int64_t n_samples,
                       int64_t row_stride, void *workspace, size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg);
/* The same encoder fed int16 PCM: sample = (float)pcm * scale in one f32 rounding -- with scale = 0.000030517578f exactly the
 * float32 array the reference script builds on the host (Silero/Inference_Silero_VAD_ONNX.py:83), so results are
 * bit-identical to vadx_silero_encode on that array while the upload and the HBM read are half the bytes. */
int vadx_silero_encode_pcm16(const float *packed, const int16_t *audio, float scale, int batch, int64_t n_samples,
                             int64_t row_stride, void *workspace, size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg);
/* A slice of the batch: clips [first_clip, first_clip + batch) of a workspace laid out for total_batch clips (first_clip a
 * multiple of 16).  Lets the encoder of one uploaded chunk run while the next chunk is still crossing PCIe (bench.py's
 * feed-inclusive mode, SURVEY 8e "pinned-host staging double-buffered per GPU"); one vadx_silero_recur over total_batch
 * follows.  Results are identical to a single vadx_silero_encode_pcm16 over the whole batch. */
int vadx_silero_encode_pcm16_part(const float *packed, const int16_t *audio, float scale, int batch, int64_t n_samples,
                                  int64_t row_stride, int first_clip, int total_batch, void *workspace,
                                  size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg);
int vadx_silero_recur(const float *packed, const void *workspace, size_t workspace_bytes, int batch,
                      int steps, const float *state0, float *probs, float *state_n, void *stream, const vadx_silero_cfg *cfg);

/* The same two launches over a SPAN of windows [first_step, first_step + n_steps) of the clips, for recordings whose
 * whole-clip workspace (32 KB per 16-clip group and window) is too large: the caller walks the spans in order on one
 * stream with one span-sized workspace (vadx_silero_workspace_bytes(batch, n_steps)), as
 * vadx.silero.SileroEngine.clips_spanned does.  encode_span reads the clip rows with the usual zero context before
 * sample 0 and zero padding after n_samples.  recur_span: `probs` points at column first_step of the [B][probs_stride]
 * table, state0 (NULL = zeros) / state_n (may alias state0) carry [2,B,128] between spans.  Results are identical to
 * the single launches (same kernels, same arithmetic order).  (Running span k's recurrence on a second stream under
 * span k+1's encoder was measured and gains nothing: both kernels are matrix-pipe-bound) */
int vadx_silero_encode_span(const float *packed, const float *audio, int batch, int64_t n_samples,
                            int64_t row_stride, int first_step, int n_steps, void *workspace,
                            size_t workspace_bytes, void *stream, const vadx_silero_cfg *cfg);
int vadx_silero_recur_span(const float *packed, const void *workspace, size_t workspace_bytes, int batch,
                           int n_steps, const float *state0, float *probs, int64_t probs_stride,
                           float *state_n, void *stream, const vadx_silero_cfg *cfg);

/* Parameters of the segmenter; defaults of get_speech_timestamps (utils_vad.py:248-263). */
typedef struct vadx_silero_seg_params {
    double threshold;                 /* 0.5 */
    double neg_threshold;             /* <0 => max(threshold-0.15, 0.01) */
    int    sampling_rate;             /* 16000 */
    double min_speech_duration_ms;    /* 250 */
    double max_speech_duration_s;     /* inf */
    double min_silence_duration_ms;   /* 100 */
    double speech_pad_ms;             /* 30 */
    double min_silence_at_max_speech; /* 98 */
    int    use_max_poss_sil_at_max_speech; /* 1 */
} vadx_silero_seg_params;

/* Device-side dual-threshold segmenter, one clip per thread: probs f32 [B][T] -> padded sample
 * index pairs int64 [B][cap][2] + counts int32 [B] (count > cap => truncated, VADX_ENOSPACE is NOT
 * raised on device; the host checks counts).  n_samples int64 [B] = true clip lengths.
 * Replaces: the state machine + padding of get_speech_timestamps, utils_vad.py:374-476
 * (the seconds rounding :478-482 stays on the host: it is Python round()). */
int vadx_silero_segments(const float *probs, int batch, int steps, const int64_t *n_samples,
                         const vadx_silero_seg_params *params, int64_t *segments, int32_t *counts,
                         int cap, void *stream);

/* VADX_ARITH_F16X2 (csrc/split2.h: every product, the STFT included, as three v_mfma_f32_16x16x32_f16 per K = 32 step on operands
 * represented to one float32 ulp by two round-to-nearest fp16 terms).  fp16 terms do not have float32's exponent range: the F16X2
 * kernels keep the largest |activation| they split and raise a sticky flag inside the packed blob when one left the fp16
 * range (|x| > 65504), or when the blob cannot run in this mode at all (a weight outside the range, an STFT basis without the DFT
 * symmetries).  This call copies the flag (0 = every result since the last reset is valid) and, when non-zero, the largest magnitude
 * seen to the host (it synchronises `stream`); reset != 0 clears it.  A flagged batch must be recomputed with VADX_ARITH_BF16X3, whose
 * bf16 terms have float32's range (vadx.silero.SileroEngine and tests/c/cabi_silero.c do so).  A caller that skips this does not read
 * plausible numbers: a flagged workgroup hands the recurrent kernel NaN, and the scores of its clips are NaN from that window on (ABI 7).
 * The underflow side needs no protocol: pack_host rebalances layers whose weights sit outside [2^-10, 2^7) by exact powers of two
 * (csrc/rebalance.h) and marks a blob with a weight tensor wholly below 2^-14 as unusable for this arithmetic (flag bit 1 on a launch). */
int vadx_silero_range_flag(const float *packed, int reset, uint32_t *flag_host, float *amax_host, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused signal front-end: int16 PCM -> prep -> framed windowed DFT against the
 * reference's own float32 table -> |.|^2 -> mel -> log.
 * Replaces the graph prefix every exported model starts with:
 *   FSMN      FSMN/Export_FSMN_VAD.py:76-81      + FSMN/STFT_Process.py:144-157
 *   MarbleNet Export_NVIDIA_MarbleNet_VAD.py:236-262 + NVIDIA_.../STFT_Process.py:265-279
 *   FireRed   FireRedVAD/Export_FireRedVAD.py:428-461 + FireRedVAD/STFT_Process.py (same class)
 *   DFSMN     DFSMN/.../Export_DFSMN_VAD.py:318-325,342-348 + DFSMN/.../STFT_Process.py:154-172
 * ------------------------------------------------------------------------------------------- */
typedef struct vadx_frontend_cfg {
    int   prep;        /* 0: x-mean(window), then y[n]=a[n]-0.97*a[n-1], y[0]=a[0]        (FSMN)
                          1: y[n]=k0*x[n-1]+k1*x[n], x[-1]=0                             (MarbleNet, FireRed)
                          2: y = k1*x - mean(k1*x)                                       (DFSMN STFT-B)
                          3/4/5: DFSMN feature streams, see vadx_frontend_logmel_ex
                          6: linear resample (in_window_len -> window_len samples), THEN prep 1   (IN_SAMPLE_RATE > 16000)
                          7: prep 1 at the input rate, THEN linear resample                    (IN_SAMPLE_RATE < 16000)
                             = F.interpolate(mode='linear', align_corners=False) of the exports built for another input
                             rate (Export_NVIDIA_MarbleNet_VAD.py:237-254, FireRedVAD/Export_FireRedVAD.py:431-449) */
    float k0, k1;
    int   center_pad;  /* zeros on each side of the window (n_fft/2, or 0 for snip-edges) */
    int   tap0, taps;  /* non-zero span of the centre-padded analysis window inside n_fft */
    int   hop;         /* multiple of 16, <= 320, and taps <= 4 * hop (at most four hops per analysis window): any other geometry is REFUSED
                          (vadx_frontend_packed_floats returns 0, the other entry points VADX_EINVAL) -- there is no slower generic path */
    int   n_bins;      /* n_fft/2+1 */
    int   n_mels;      /* multiple of 16 */
    int   log_mode;    /* 0: log(max(x,floor))   1: log(x+floor) */
    float log_floor;
    int   frames;      /* frames per window */
    int   window_len;  /* samples per window (at 16 kHz, i.e. after the in-graph resample of prep 6 / 7) */
    int   in_window_len; /* prep 6 / 7: samples per window in the audio buffer (input rate); 0 otherwise */
    float rs_scale;    /* prep 6 / 7: source samples per output sample, float32(1 / scale_factor) as torch computes it */
    int   fold;        /* 0: dense DFT product.  1 / 2: folded product of vadx_frontend_logmel (mirror-paired taps about the window
                          centre + f16 residual; same table bits, half the f32 MFMAs) -- the value vadx_frontend_fold_kind returned
                          for this table; set it before vadx_frontend_packed_floats / _pack_host.  3: opt-in, periodic windows centred on
                          n_fft/2 only (FSMN): time x frequency fold, a quarter of the dense MACs, noisier on bands far below the
                          frame's peak (csrc/frontend.hip "kind 3"); _pack_host refuses it for a table that does not admit it.
                          4: dense product of the reference table itself on bf16 x 3 exactly split operands (csrc/split3.h: six bf16
                          MFMAs per 32 taps, float32-class accuracy, no symmetry assumption about the table); hop 160, preps 0 - 2,
                          taps <= 512 -- the Python front-end's default where it applies.
                          Ignored by _logmel_ex / _stft_ft callers' kernels (they take the dense tables, which every blob carries) */
} vadx_frontend_cfg;

/* Which fold (0 = none) the reference's windowed DFT table admits for this geometry: the table must equal, about the window
 * centre, an even real / odd imaginary pair to within what the f16 residual carries (1.5e-4 of the table scale). */
int vadx_frontend_fold_kind(const vadx_frontend_cfg *cfg, const float *cos_tab, const float *sin_tab, int n_fft);


size_t vadx_frontend_packed_floats(const vadx_frontend_cfg *cfg);
/* Host repack of the reference tables (cos/sin [n_bins][n_fft] windowed, fbank [n_mels][n_bins]);
 * mel_kb receives 2*n_mels/16 ints (banded mel ranges) that vadx_frontend_logmel wants back. */
int vadx_frontend_pack_host(const vadx_frontend_cfg *cfg, const float *cos_tab, const float *sin_tab,
                            int n_fft, const float *fbank, float *packed_host, int32_t *mel_kb);
/* audio int16 [batch][row_stride]; window w of clip b starts at b*row_stride + w*win_stride;
 * out f32 [batch*windows_per_clip][frames][n_mels]; means_ws: batch*windows floats (prep 0/2). */
int vadx_frontend_logmel(const vadx_frontend_cfg *cfg, const float *packed, const int32_t *mel_kb_host,
                         const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch,
                         int windows_per_clip, float *means_ws, float *out, void *stream);
/* vadx_frontend_logmel with the window means (prep 0 / 2) GIVEN by the caller instead of computed into a workspace, and the kernel that
 * computes them (mean of scale * x over each window: exact integer sum, one float32 rounding). */
int vadx_frontend_logmel_means(const vadx_frontend_cfg *cfg, const float *packed, const int32_t *mel_kb_host,
                               const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch,
                               int windows_per_clip, const float *means, float *out, void *stream);
int vadx_frontend_window_means(const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch, int windows_per_clip,
                               int window_len, float scale, float *means, void *stream);

/* DFSMN variants of the fused front-end (DFSMN/.../Export_DFSMN_VAD.py:322-325, 338-348):
 *   prep 3: near stream  a = k1*x - mean, pre-emphasis 0.97 keeping a[0]        (int16 source + means)
 *   prep 4: AEC stream   pre-emphasis of the float waveform `faux` [windows][window_len]
 *   prep 5: echo stream  near_pe - k0 * aec_pe                                  (both sources)
 * Each call writes n_mels columns at `out_off` of rows of `out_stride` floats (3 calls fill the 240-dim row). */
int vadx_frontend_logmel_ex(const vadx_frontend_cfg *cfg, const float *packed, const int32_t *mel_kb_host,
                            const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch,
                            int windows_per_clip, const float *means, const float *faux, int out_stride,
                            int out_off, float *out, void *stream);
/* Raw complex STFT (prep 2) into a frame-tiled tensor [window*ceil(frames/16) + t/16][c_total][n_bins][16]:
 * real part at channel c_off, imaginary at c_off+1 (the two-stream STFT-B of DFSMN_VAD.forward :322-325). */
int vadx_frontend_stft_ft(const vadx_frontend_cfg *cfg, const float *packed, const int16_t *audio,
                          int64_t row_stride, int64_t win_stride, int batch, int windows_per_clip,
                          float *means_ws, float *ft_out, int c_total, int c_off, void *stream);

/* ---------------------------------------------------------------------------------------------
 * FSMN-VAD
 * ------------------------------------------------------------------------------------------- */
typedef struct vadx_fsmn_dims {          /* FunASR FSMN(input 400, proj 128, lorder 20, 4 layers pinned in-tree) */
    int   input_affine_dim, linear_dim, output_affine_dim, output_dim;   /* external config: 140/250/140/248 */
    int   frames;                         /* T = window_len // 160 + 1 (101) */
    float speech_2_noise_ratio;           /* FSMN/Export_FSMN_VAD.py:34,87-92 */
    int   arithmetic;                     /* VADX_ARITH_* of the dense layers (0 = AUTO = BF16X3: float32's exponent range, nothing to check; F16X2
                                           * is for callers that read vadx_fsmn_range_flag with the results and recompute a flagged batch on
                                           * BF16X3, as vadx.fsmn.FsmnEngine does).  The packed blob carries the weight fragments of THIS
                                           * arithmetic only: pass the same dims to vadx_fsmn_pack_host and to every launch.  pack_host refuses
                                           * F16X2 when a weight lies outside the fp16 range (pack BF16X3 then). */
} vadx_fsmn_dims;

typedef struct vadx_fsmn_weights_host {  /* torch layouts: Linear weight [out][in]; conv_left [128][20] */
    const float *in1_w, *in1_b, *in2_w, *in2_b;
    const float *lin_w[4], *fir_w[4], *aff_w[4], *aff_b[4];
    const float *out1_w, *out1_b, *out2_w, *out2_b;
    const float *cmvn_means, *cmvn_vars;  /* [400] each, (x + means) * vars */
} vadx_fsmn_weights_host;

size_t vadx_fsmn_packed_floats(const vadx_fsmn_dims *dims);
/* The F16X2 kernels' sticky range flag, as vadx_silero_range_flag: flag != 0 = an activation of some launch since the last reset left the
 * fp16 range and that launch's results must be recomputed with a BF16X3 blob.  Synchronises `stream`. */
int vadx_fsmn_range_flag(const vadx_fsmn_dims *dims, const float *packed, int reset, uint32_t *flag_host, float *amax_host, void *stream);
int vadx_fsmn_pack_host(const vadx_fsmn_dims *dims, const vadx_fsmn_weights_host *w, float *packed_host);

/* Per-frame energy term of the score gate: log10(sum_512 (y/(sqrt(L)*2e-5))^2 + 2e-5) of the prepped
 * window, 97 real frames + last value repeated to `frames`; means = per-window DC (from
 * vadx_frontend_logmel's means workspace).  Replaces FSMN/Export_FSMN_VAD.py:93-97. */
int vadx_fsmn_energy(const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch,
                     int windows_per_clip, int window_len, int frames, const float *means, float *db,
                     void *stream);
/* The window means (prep 0: exact integer sum / window_len, what vadx_frontend_logmel computes into its workspace) AND the energy term above
 * in ONE pass over the PCM: means f32 [batch*windows_per_clip], db f32 [batch*windows_per_clip][frames].  Feed the means to
 * vadx_frontend_logmel_means.  (Windows that do not start on 16-byte boundaries or whose length is not a multiple of 32 take the two
 * separate kernels.)  Replaces nothing more than vadx_fsmn_energy does: FSMN/Export_FSMN_VAD.py:76-79, 93-97. */
int vadx_fsmn_window_stats(const int16_t *audio, int64_t row_stride, int64_t win_stride, int batch, int windows_per_clip,
                           int window_len, int frames, float *means, float *db, void *stream);

/* One ORT-boundary call for `batch` independent streams, after the front-end:
 *   feeds   audio -> (logmel [B][T][80], db [B][T]); cache_0..3 f32 [B][128][19];
 *           one_minus_speech_threshold f32 [B]; noise_average_dB f32 [B]
 *   fetches score u8 [B][T]; cache_0..3; noisy_dB f32 [B]      (+ optional P(silence) f32 [B][T])
 * Replaces ort_session_A.run(...), FSMN/Inference_FSMN_VAD_ONNX.py:177-187 (graph :75-101). */
int vadx_fsmn_run(const vadx_fsmn_dims *dims, const float *packed, const float *logmel, const float *db,
                  const float *const cache_in[4], float *const cache_out[4], const float *thr,
                  const float *noise_db, int batch, uint8_t *score, float *noisy_db, float *psil,
                  void *stream);

typedef struct vadx_fsmn_loop_params {   /* FSMN/Inference_FSMN_VAD_ONNX.py:16-23,79-81,159-171 */
    int    look_backward;                 /* frames: int(LOOK_BACKWARD*16000 // 160) = 30; 0 is allowed and means what it
                                           * means in the reference: slide_range = T, one-frame vote, empty tail */
    float  one_minus_speech_threshold;    /* 1.0 */
    float  noise_db_init;                 /* (BACKGROUND_NOISE_dB_INIT + SNR_THRESHOLD) * 0.1 */
    float  snr_threshold;                 /* SNR_THRESHOLD * 0.1 */
    double speaking_score, silence_score; /* 0.5, 0.5 */
} vadx_fsmn_loop_params;

/* Whole clips: every clip's windows in order with cache + noise-floor carry and the look-ahead
 * vote; flags u8 [B][W*(T-lb) + lb] = the reference's `saved` list (1 = silence).
 * cache_ws: scratch of B*4*128*19 floats.  noise_trace (optional) [B][W] = noise floor after each window.
 * Replaces the while-loop + tail, FSMN/Inference_FSMN_VAD_ONNX.py:176-234. */
int vadx_fsmn_clips(const vadx_fsmn_dims *dims, const float *packed, const float *logmel, const float *db,
                    int batch, int windows_per_clip, const vadx_fsmn_loop_params *lp, float *cache_ws,
                    uint8_t *flags, float *noise_trace, void *stream);

/* ---------------------------------------------------------------------------------------------
 * FireRedVAD / FireRedAED DetectModel and VadPostprocessor (row a16)
 * ------------------------------------------------------------------------------------------- */
typedef struct vadx_firered_cfg {        /* checkpoint `args` (FireRedVAD/Export_FireRedVAD.py:336-337) */
    int idim, R, M, H, P, N1, S1, N2, S2, odim;
    int frames;                           /* frames per window: (L-400)//160+1 = 98 */
    int arithmetic;                       /* VADX_ARITH_* of the point-wise layer pairs (0 = AUTO = BF16X3 where H = 256, P = 128; float32 MFMAs
                                           * elsewhere; F16X2 on request, with vadx_firered_range_flag).  As vadx_fsmn_dims.arithmetic: the blob
                                           * carries the fragments of this arithmetic only. */
} vadx_firered_cfg;

typedef struct vadx_firered_weights_host {   /* torch layouts, 1x1 convs as [out][in] */
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;          /* dfsmn.fc1 (CMVN already folded), dfsmn.fc2 */
    const float *fsmn_lb[16], *fsmn_la[16];              /* [P][N1], [P][N2] depthwise filters per FSMN */
    const float *blk_fc1_w[16], *blk_fc1_b[16], *blk_fc2_w[16];   /* DFSMNBlock r = 1..R-1 (index 0 unused) */
    const float *dnn_w[4], *dnn_b[4];
    const float *out_w, *out_b;
} vadx_firered_weights_host;

size_t vadx_firered_packed_floats(const vadx_firered_cfg *cfg);
/* The F16X2 kernel's sticky range flag, as vadx_silero_range_flag. */
int vadx_firered_range_flag(const vadx_firered_cfg *cfg, const float *packed, int reset, uint32_t *flag_host, float *amax_host, void *stream);
int vadx_firered_pack_host(const vadx_firered_cfg *cfg, const vadx_firered_weights_host *w, float *packed_host);
/* logmel f32 [windows][frames][80] (vadx_frontend_logmel, 'firered' geometry) -> probs f32 [windows][odim][frames].
 * Replaces ort_session_A.run([probs], {audio}), FireRedVAD/Inference_FireRed_ONNX.py:567-572
 * (graph: FireRedVAD/Export_FireRedVAD.py:420-467 after the front-end). */
int vadx_firered_run(const vadx_firered_cfg *cfg, const float *packed, const float *logmel, int windows,
                     float *probs, void *stream);

/* Streaming variant (FireRed Stream-VAD): one chunk of cfg->frames frames for `streams`
 * independent streams; caches f32 [R][streams][P][(N1-1)*S1] in -> out (must not alias); cfg->N2 == 0.
 * Replaces ort_session_C.run([probs, caches_out], {audio, caches_in}), Inference_FireRed_ONNX.py:798-801
 * (graph FireRedVAD/Export_FireRedVAD.py:479-612, 656-749 after the front-end). */
int vadx_firered_stream_run(const vadx_firered_cfg *cfg, const float *packed, const float *logmel, int streams,
                            const float *caches_in, float *caches_out, float *probs, void *stream);

typedef struct vadx_vadpost_params {     /* VadPostprocessor.__init__, Inference_FireRed_ONNX.py:108-120 */
    int   smooth_window_size;
    float prob_threshold;
    int   min_speech_frame, max_speech_frame, min_silence_frame, merge_silence_frame, extend_speech_frame;
} vadx_vadpost_params;

size_t vadx_vadpost_workspace_bytes(int batch, int stride);
/* probs f32 [B][stride] (n_frames[b] valid) -> decisions i8 [B][stride] + (start_frame, end_frame)
 * int32 pairs [B][cap][2] + counts [B]; one clip per thread.
 * Replaces VadPostprocessor.process + the edge extraction of decision_to_segment
 * (Inference_FireRed_ONNX.py:122-168, 181-304; MarbleNet copy Inference_NVIDIA_...:160-353);
 * the float32 seconds arithmetic + round(,3) of decision_to_segment :169-179 stays on the host. */
int vadx_vadpost(const vadx_vadpost_params *prm, const float *probs, int stride, const int32_t *n_frames,
                 int batch, int8_t *decisions, int32_t *segments, int32_t *counts, int cap,
                 void *workspace, size_t workspace_bytes, void *stream);

typedef struct vadx_stream_vadpost_params {     /* StreamVadPostprocessor.__init__, FireRedVAD/Export_FireRedVAD.py:1161-1190 */
    int   smooth_window_size;                   /* <= 16 */
    float speech_threshold;
    int   pad_start_frame, min_speech_frame, max_speech_frame, min_silence_frame;
} vadx_stream_vadpost_params;

size_t vadx_stream_vadpost_state_bytes(int streams);
/* The streaming decision state machine for `streams` independent streams, one thread per stream: probs f32 [streams][probs_stride]
 * (`frames` new frames each) -> the segments that END inside this chunk as (start_frame, end_frame) int32 pairs [streams][cap][2]
 * (0-based frames: multiply by 1 / frames_per_second on the host) + counts [streams] (count > cap => truncated).  `state`
 * (vadx_stream_vadpost_state_bytes, device) carries the moving-average ring buffer and the machine between calls; reset != 0 starts
 * every stream from the reference's reset() state; flush != 0 also reports the segment still open after the last frame, as the
 * reference's process_batch does at the end of its input, without closing it.  Same float32 operations in the same order as the
 * reference's per-frame loop (its comparisons are float32 under NumPy 2).
 * Replaces StreamVadPostprocessor.process_batch, FireRedVAD/Export_FireRedVAD.py:1161-1339 (driver Inference_FireRed_ONNX.py:788-822). */
int vadx_stream_vadpost(const vadx_stream_vadpost_params *prm, const float *probs, int64_t probs_stride, int frames, int streams,
                        void *state, int reset, int flush, int32_t *segments, int32_t *counts, int cap, void *stream);

/* ---------------------------------------------------------------------------------------------
 * MarbleNet building blocks: one fused launch per Jasper sub-block
 * (depthwise conv -> pointwise 1x1 + folded BatchNorm [-> + residual 1x1] -> ReLU) and the frame
 * classifier head.  Replaces the encoder/decoder calls of NVIDIA_VAD_Optimized.forward,
 * Export_NVIDIA_MarbleNet_VAD.py:265-274 (BatchNorm folded on the host as in :58-151).
 * ------------------------------------------------------------------------------------------- */
typedef struct vadx_sepconv_cfg {
    int cin, cout;            /* <= 128 */
    int kernel, stride, dilation;   /* 'same' padding = dilation*(kernel-1)/2 */
    int depthwise;            /* 0: plain 1x1 conv (kernel must be 1) */
    int residual_cin;         /* 0: no residual branch; else channels of the block input */
    int relu;
} vadx_sepconv_cfg;

/* x: element (b, c, t) at x[b*xs_b + c*xs_c + t*xs_t] (lets the first block read the time-major
 * log-mel directly); xres [B][residual_cin][t_out]; y [B][cout][t_out].  Weights on the device:
 * dw_w [cin][kernel]; pw_w [cout_pad16][cin_pad16], pw_b [cout_pad16]; res_w [cout_pad16][rescin_pad16]. */
struct vadx_marblenet_cfg;   /* below: arithmetic of the launch + the fp16 x 2 range flag words; NULL = float32 MFMAs */
int vadx_sepconv_block(const vadx_sepconv_cfg *cfg, const float *dw_w, const float *pw_w, const float *pw_b,
                       const float *res_w, const float *res_b, const float *x, int64_t xs_b, int64_t xs_c,
                       int64_t xs_t, int t_in, const float *xres, float *y, int batch, int t_out, void *stream,
                       const struct vadx_marblenet_cfg *mcfg);
/* Fused forms for the published MarbleNet 3x2x64 layout (what MarbleNetEngine launches; the per-sub-block entry above stays
 * for other Jasper stacks).  block2: one residual Jasper block = two separable sub-blocks (depthwise `kernel`, stride 1,
 * 64 filters each) + the residual 1x1 branch of the block input, x [B][cin][T] -> y [B][64][T]; weights as for
 * vadx_sepconv_block (dw [c][kernel]; pw / res fragment-major, BatchNorm folded; biases padded to 16).
 * tail: block 5 (depthwise k 29, dilation 2, 64 -> 128) -> block 6 (plain 1x1, 128 -> 128) -> Linear(128 -> 2) -> softmax,
 * x [B][64][T] -> score0 / score1 [B][T] (dec_w [2][128], dec_b [2]); the 128-channel tensors never reach HBM. */
/* cfg (may be NULL = float32 MFMAs): `arithmetic` VADX_ARITH_AUTO / VADX_ARITH_F32 -- pw0 / pw1 / res_w are fragment-major float32
 * (vadx_frag_major_host); VADX_ARITH_F16X2 -- they are fp16 x 2 fragments (vadx_frag_h2_host: K_QUARTER for block2 with cin = 128, K_PLAIN for block2 with cin = 64 and for tail's pw / w6) and the 1x1 convs run as split
 * products on the fp16 pipe (float32-class results, csrc/split2.h); `range_flag` then points at two device words {sticky flag, bits of the
 * largest |operand|} that a launch raises when an activation left the fp16 range -- the caller reads them with the results and recomputes
 * that batch on VADX_ARITH_F32 (MarbleNetEngine does).  VADX_ARITH_BF16X3 is refused (no such form of these kernels). */
typedef struct vadx_marblenet_cfg {
    int32_t arithmetic;       /* VADX_ARITH_* */
    int32_t reserved;
    void *range_flag;         /* device uint32_t[2], zeroed by the caller; required for VADX_ARITH_F16X2 */
} vadx_marblenet_cfg;
int vadx_marblenet_block2(int cin, int kernel, const float *dw0, const float *pw0, const float *b0, const float *dw1,
                          const float *pw1, const float *b1, const float *res_w, const float *res_b, const float *x,
                          float *y, int batch, int frames, void *stream, const vadx_marblenet_cfg *cfg);
int vadx_marblenet_tail(const float *dw, const float *pw, const float *pb, const float *w6, const float *b6,
                        const float *dec_w, const float *dec_b, const float *x, float *score0, float *score1,
                        int batch, int frames, void *stream, const vadx_marblenet_cfg *cfg);
/* enc [B][C][T] -> softmax(Linear(C->2)) split into score0 / score1 [B][T]. */
int vadx_frame_classifier(const float *enc, const float *dec_w, const float *dec_b, int batch, int channels,
                          int frames, float *score0, float *score1, void *stream);

/* ---------------------------------------------------------------------------------------------
 * DFSMN near+far: ICCRN building blocks on frame-tiled ("FT") activations
 *   [tile = chunk*NT + t/16][C][F][16]  -- element (c,f,t) at ((tile*C + c)*F + f)*16 + t%16.
 * Reference: DFSMN/near_and_far_end_audio/Export_DFSMN_VAD.py (LayerNorm :157-167, CFB :76-93,
 * CepsUnit :96-154, CH_LSTM_F :270-284, CH_LSTM_T :252-267, NET :170-249, DFSMN_VAD :287-354).
 * ------------------------------------------------------------------------------------------- */
typedef struct vadx_ft_view { const float *ptr; int c_total, c_off, c; } vadx_ft_view;   /* channel slice */
typedef struct vadx_ft_ln { const float *stats, *w, *b; } vadx_ft_ln;                    /* LayerNorm on the fly */

/* stats[tile][16][2] = (mean, 1/(unbiased std + 1e-6)) over (C,F) of cat(a, b) per frame. */
int vadx_dfsmn_frame_stats(const vadx_ft_view *a, const vadx_ft_view *b, int F, int tiles, float *stats, void *stream);
/* Fused alternative: pw_conv and the forward dft_f can emit, per frame, the (count, mean, sum of squared deviations) of
 * what each workgroup wrote -- "partial statistics", float [tiles][VADX_DFSMN_STAT_PARTS][16][4], caller-owned --
 * and this merges the partials of one tensor (part_b NULL) or of the channel concatenation of two into the same
 * stats[tile][16][2] that frame_stats produces, without re-reading the tensors. */
#define VADX_DFSMN_STAT_PARTS 2
int vadx_dfsmn_stats_merge(const float *part_a, const float *part_b, int tiles, float *stats, void *stream);
/* Shapes are instantiated at compile time: (co, cin, kf, mode) in {(20,20,1,1), (20,40,1,1), (20,20,3,2), (20,24,1,0), (20,40,1,0),
 * (40,40,1,0), (2,60,1,0)} -- the ICCRN's seven; anything else returns VADX_EINVAL.
 * mode 0: out0 = act(conv(cat(a,b)) + bias)        (kf taps along F, weights [ceil16(co)][kf*cin])
 * mode 1: CFB front: g = sigmoid(convG(LN(x)) + bias); xi = convI(x) + bias2; out0 = g*xi; out1 = xi - g*xi
 * mode 2: CFB back : out0 = conv31(LN(x)) + bias + add
 * part0 / part1 (optional, may be NULL): partial statistics of out0 / out1 (mode 1) for vadx_dfsmn_stats_merge. */
int vadx_dfsmn_pw_conv(int mode, const vadx_ft_view *a, const vadx_ft_view *b, const vadx_ft_ln *ln,
                       const float *w, const float *bias, const float *w2, const float *bias2,
                       const vadx_ft_view *add, const vadx_ft_view *out0, const vadx_ft_view *out1,
                       int F, int co, int kf, int act, int tiles, float *part0, float *part1, void *stream);
/* CepsUnit's length-160 real DFT along F (inverse=0: in C ch x 160 -> out 2C ch x 81, LayerNorm on the
 * input) and its pinv-based inverse fused with the complex product (inverse=1: in = spectrum, lo = LSTM
 * output, both 2C ch x 81 -> out C ch x 160).  tbl = float [160][160] row-major:
 *   forward: rows = cos bins 0..80 | sin bins 1..79 (the sine rows of bins 0 and 80 vanish; those two outputs are written as 0);
 *   inverse: rows = output bin f, columns k = re 0..80 | im 1..79 of the pseudo-inverse basis (its im(0), im(80) columns are 0).
 * part (optional, forward only): partial statistics of out. */
int vadx_dfsmn_dft_f(int inverse, const vadx_ft_view *in, const vadx_ft_view *lo, const vadx_ft_ln *ln,
                     const float *tbl, const vadx_ft_view *out, int C, int tiles, float *part, void *stream);
/* bi-LSTM (hidden 20) along F with the tile's 16 frames as the batch; in->c = 4 or 40; out 40 ch.  arithmetic: VADX_ARITH_* of the
 * 40-channel (CepsUnit) form -- AUTO = BF16X3, F32 = the float32-MFMA kernel, F16X2 = fp16 x 2 split products with range_flag = two
 * zeroed device words {sticky flag, bits of the largest |operand|} (the caller recomputes a flagged batch on BF16X3 / F32); range_flag
 * may be NULL for the other arithmetics. */
int vadx_dfsmn_lstm_f(const vadx_ft_view *in, const vadx_ft_ln *ln, const float *const w_ih[2],
                      const float *const w_hh[2], const float *const b_ih[2], const float *const b_hh[2],
                      const vadx_ft_view *out, int F, int tiles, void *stream, int arithmetic, void *range_flag);

/* One gated conv block (CFB :76-93 with its CepsUnit :96-154) as two streaming launches around vadx_dfsmn_lstm_f
 * (csrc/dfsmn_cfb.hip): the block's 20-channel intermediates gx, r, lo, ceps never reach memory.
 *   front: cat(a, b) (20 or 40 ch x 160, LN0 statistics stats0[tile][16][2] as frame_stats / stats_merge give them)
 *          -> y1 [tiles][20][160][16] = conv31(ln1_w * gx) WITHOUT LayerNorm 1's (mean, inv, bias) and the conv bias,
 *             stats1 [tile][16][2] = (mean, 1/(std + 1e-6)) of gx,
 *             li [tiles][40][81][16] = DFT_F(LN2(xi - gx)) (real | imaginary parts) and its statistics stats_li for the LSTM's LayerNorm;
 *   back:  hf (the LSTM output, [tiles][40][81][16]), li, y1, stats1 -> out (20-channel slice of an FT tensor)
 *          = conv31(LN1(gx)) + bias + ceps_unit(...), part (optional): its partial statistics for vadx_dfsmn_stats_merge.
 * All pointers are device pointers; the derived tables are built on the host (vadx/dfsmn.py: Iccrn._cfb_tables):
 *   ln0_w [cin][160]; gate_w / in_w [32][cin], in_b [32] (rows >= 20 zero); conv_w [32][60], column tap * 20 + ci;
 *   front_tab [160 bins][4][20 channels]: (conv_gate ln0_w, conv_gate ln0_b + gate bias, ln1_w, ln2_w) -- LayerNorm 0 is applied
 *           BEHIND the gate conv: pre-activation = inv0 * (Wg (ln0_w * x)) + front_tab[1] - mean0 * inv0 * front_tab[0];
 *   fwd_tbl [10 row tiles][40 k-steps][64 lanes]: lane (q, i) of fragment (m, s) = T[16 m + i][4 s + q], T = the 160 x 160 forward
 *           table of vadx_dfsmn_dft_f; fwd_fix [20 ch][10][64]: lane quarter q = 0 -> (T ln2_w[c])[16 m + i], q = 1 -> (T ln2_b[c])[..], else 0;
 *   lin_w [48][40], lin_b [48]: CepsUnit's Linear with rows permuted so that row 16 t + 4 q + r = (r >> 1 ? imaginary : real) output
 *           of channel 8 t + 2 q + (r & 1) (zero rows for channels >= 20);
 *   inv_tbl [10][41][64]: k-steps 0..20 = real parts of bins 4 s .. 4 s + 3, 21..40 = imaginary parts of bins 4 (s - 21) .. + 3
 *           (zero for bin 0 and bins > 80); out_fix [20][10][64]: q = 0 -> conv31(ln1_w)[c][16 m + i], q = 1 -> conv31(ln1_b) + bias. */
typedef struct vadx_dfsmn_cfb_weights {
    const float *ln0_w, *gate_w, *in_w, *in_b, *front_tab, *conv_w, *fwd_tbl, *fwd_fix, *lin_w, *lin_b, *inv_tbl, *out_fix;
    /* ABI 5: the two DFT tables as bf16 x 3 split A fragments (csrc/split3.h) for the split-product kernels, or NULL (the f32-MFMA
     * kernels run then): fwd_tbl_q [10 row tiles][5 chunks of 32 bins][3 planes][256 floats], lane 16 g + i of a fragment
     * holds T[16 tile + i][32 chunk + 8 g + e], e = 0..7, as eight bf16;  inv_tbl_q [10][5 chunks][3][256], same fragment layout, with
     * the 160 columns (81 real + 79 imaginary parts) ordered so that chunk j = [re of ceps bins 16 j .. 16 j + 15 | im of the same
     * bins] and the slot of the non-existent im of bin 0 holds re of bin 80. */
    const float *fwd_tbl_q, *inv_tbl_q;
    /* ABI 6: VADX_ARITH_* of the two halves.  front: AUTO = BF16X3 when fwd_tbl_q is given, else F32.  back: AUTO = F32 (its split form is
     * no faster: that half waits on 64-byte rows, not on the matrix pipe).  F16X2 is not built for these kernels and is refused. */
    int32_t front_arithmetic, back_arithmetic;
} vadx_dfsmn_cfb_weights;
int vadx_dfsmn_cfb_front(const vadx_dfsmn_cfb_weights *w, const vadx_ft_view *a, const vadx_ft_view *b, const float *stats0,
                         float *y1, float *stats1, float *li, float *stats_li, int tiles, void *stream);
int vadx_dfsmn_cfb_back(const vadx_dfsmn_cfb_weights *w, const float *hf, const float *li, const float *y1, const float *stats1,
                        const vadx_ft_view *out, float *part, int tiles, void *stream);

/* x4 = [mix_re, mix_im, |alpha| far_re, |alpha| far_im] (FT, 4 ch x 160), DFSMN_VAD.forward :326-335.
 * pow_far NULL: far power from channels 2, 3 (near + far model).  pow_far = [160][frames][10] floats: the near-end-only
 * model's baked far-end power (DFSMN/only_near_end_audio/Export_DFSMN_VAD.py:309, :327); channels 2, 3 of `in` then
 * hold its constant far spectrum. */
int vadx_dfsmn_alpha_scale(const float *in, float *out, int chunks, int nt, const float *w1, const float *b1,
                           const float *w2, const float *b2, const float *pow_far, int frames, void *stream);
/* LSTM along time, 16 bins per workgroup.  which = 0: NET.ch_lstm (in 20, hidden 40, 2 layers, Linear 40->20,
 * output multiplied element-wise with `mul` = e5);  which = 1: NET.out_ch_lstm (in 40, hidden 20, Linear 20->40).
 * wl rows zero-padded to a multiple of 16. */
int vadx_dfsmn_lstm_t(int which, const vadx_ft_view *in, const vadx_ft_ln *ln, const float *const w_ih[2], const float *const w_hh[2],
                      const float *const b_ih[2], const float *const b_hh[2], const float *wl, const float *bl,
                      const vadx_ft_view *mul, const vadx_ft_view *out, int F, int frames, int chunks, void *stream);
/* The same with an explicit frame stride between chunks (a multiple of 4): chunk c's frame t is column (c * frame_stride + t) % 16 of
 * tile (c * frame_stride + t) / 16.  vadx_dfsmn_lstm_t is frame_stride = 16 * ceil(frames / 16). */
/* arithmetic (ABI 6): VADX_ARITH_AUTO / VADX_ARITH_F32 = float32 MFMAs; VADX_ARITH_F16X2 = the two-layer net (which = 0) as fp16 x 2 split
 * products (csrc/split2.h; weights split by the kernel itself), range_flag = two zeroed device words {sticky flag, bits of the largest
 * |operand|} raised when an input or weight left the fp16 range (the caller then recomputes on VADX_ARITH_F32); which = 1 runs float32
 * MFMAs for every arithmetic. */
int vadx_dfsmn_lstm_t_ex(int which, const vadx_ft_view *in, const vadx_ft_ln *ln, const float *const w_ih[2], const float *const w_hh[2],
                         const float *const b_ih[2], const float *const b_hh[2], const float *wl, const float *bl,
                         const vadx_ft_view *mul, const vadx_ft_view *out, int F, int frames, int chunks, int frame_stride, void *stream,
                         int arithmetic, void *range_flag);
/* Copy an FT tensor [channels][F] of `chunks` windows between frame strides (both multiples of 4, >= frames): e.g. 112 -> 104 packs
 * 101-frame windows onto 6.5 tiles each (the per-frame kernels then process 7 % fewer tiles), 104 -> 112 unpacks the result.
 * dst holds ceil(chunks * dst_stride / 16) tiles. */
int vadx_dfsmn_ft_repack(const float *src, float *dst, int channels, int F, int frames, int chunks, int src_stride, int dst_stride, void *stream);
/* NET.istft :220-224: y_ft FT [2 ch x 160] -> out f32 [chunks][(frames-1)*160 + 1]; basis_t [320 j][320 ch]
 * (transposed inverse basis, row 319 zero), wsum_inv = the reference's window_sum_inv buffer; z_ws scratch
 * of chunks*frames*320 floats. */
int vadx_dfsmn_istft(const float *y_ft, const float *basis_t, const float *wsum_inv, float *z_ws, float *out,
                     int chunks, int frames, void *stream);
/* mask-net head: device pointers; linear weights zero-padded to multiples of 16 in BOTH dims
 * ([Hp][240], [H2p][Hp], [Hp][H2p]); conv1 [hidden][lorder]; shift already includes log(32768^2). */
typedef struct vadx_dfsmn_mask_weights {
    int hidden, fsmn_hidden, layers, lorder;
    const float *shift, *scale, *linear1_w, *linear1_b, *linear3_w, *linear3_b;
    const float *fsmn_linear_w[8], *fsmn_linear_b[8], *fsmn_project_w[8], *fsmn_conv_w[8];
} vadx_dfsmn_mask_weights;
/* feat f32 [chunks][frames][240] (3 x 80 log-fbank streams) -> vad f32 [chunks][frames]
 * (DFSMN_VAD.forward :349-353 + UniDeepFsmn.compute1, uni_deep_fsmn.py:311-329). */
int vadx_dfsmn_mask_net(const vadx_dfsmn_mask_weights *w, const float *feat, int chunks, int frames, float *vad,
                        void *stream);
/* Look-ahead vote + tail of the DFSMN driver (Inference_DFSMN_VAD_ONNX.py:231-273): vad f32 [B][W][frames]
 * -> flags u8 [B][W*(frames-lb) + lb] (1 = silence). */
int vadx_dfsmn_vote(const float *vad, int batch, int windows, int frames, int look_backward, double speaking_score,
                    double silence_score, uint8_t *flags, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Test hooks (used by tests/ only)
 * ------------------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------------
 * Audio ingest: the reference drivers' pydub chain set_channels(1).set_frame_rate(r)
 * (FSMN/Inference_FSMN_VAD_ONNX.py:68 and the other four drivers) = audioop.tomono(0.5, 0.5) + audioop.ratecv,
 * bit-exact on 16-bit PCM.  src: batch rows of frames_in interleaved frames (1 or 2 channels), row stride in
 * int16 elements; dst: batch rows of vadx_ingest_out_frames(frames_in, in_rate, out_rate) mono samples.
 * ------------------------------------------------------------------------------------------- */
int64_t vadx_ingest_out_frames(int64_t frames_in, int in_rate, int out_rate);
int vadx_ingest_pcm16(const int16_t *src, int64_t src_stride, int channels, int64_t frames_in, int in_rate,
                      int out_rate, int16_t *dst, int64_t dst_stride, int batch, void *stream);

/* Weight layout of every GEMM operand the kernels stream from L2 ("fragment-major"): a row-major
 * [rows][cols] matrix, zero-padded to multiples of 16 both ways, stored as
 * [rows/16 tiles][cols/16 blocks][64 lanes][4]: the float4 of (tile, block S, lane 16q+i) holds
 * W[16*tile + i][16*S + 4*q + 0..3], so one wave-wide load is one contiguous 1 KB run.  The *_pack_host
 * functions produce it themselves; entry points that take bare weight pointers (vadx_sepconv_block's
 * pw_w / res_w, vadx_dfsmn_mask_weights' linear1_w / fsmn_linear_w / fsmn_project_w)
 * expect buffers converted with this helper.  dst holds vadx_frag_major_floats(rows, cols) floats. */
size_t vadx_frag_major_floats(int rows, int cols);
int vadx_frag_major_host(const float *src, int rows, int cols, float *dst);
/* The fp16 x 2 counterpart for the entry points that take bare weight pointers and an `arithmetic` (vadx_marblenet_block2 / _tail):
 * [rows/16 tiles][cols/32 chunks][2 planes][64 lanes][8 fp16], lane 16q+i slot e = W[16*tile + i][32*chunk + k(q, e)], round-to-nearest
 * fp16 terms h0, h1 = (w - h0) * 2^11 (csrc/split2.h).  k_order: VADX_H2_K_PLAIN k = 8q + e (vadx_marblenet_tail's pw / w6),
 * VADX_H2_K_QUARTER k = 16*(e>>2) + 4q + (e&3) (vadx_marblenet_block2's pw0 / pw1 / res_w when cin = 128; with cin = 64 they are K_PLAIN).  dst holds vadx_frag_h2_floats(rows, cols)
 * floats; *wmax_out (optional) receives the largest |w|; fails when a weight is outside the fp16 range (keep that matrix on VADX_ARITH_F32). */
#define VADX_H2_K_PLAIN 0
#define VADX_H2_K_QUARTER 1
size_t vadx_frag_h2_floats(int rows, int cols);
int vadx_frag_h2_host(const float *src, int rows, int cols, int k_order, float *dst, float *wmax_out);

#ifdef __cplusplus
}
#endif
#endif /* VADX_H */
