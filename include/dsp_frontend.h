/*
 * dsp_frontend.h -- C ABI of the MI355X-native speech feature front-end (libdsp_frontend.so).
 *
 * Drop-in boundary.  The reference (AuCson/DSP-Speech-Recognition) is 100 % Python: its "FFI" for
 * this path is the module surface of features/{sigproc,base,endpoint}.py.  Each entry point below
 * names the reference function(s) (file:line, relative to the reference checkout) whose arithmetic
 * it replaces; the Python mirror in dsp-speech-recognition_amd/features/ binds them with ctypes
 * under the reference's own function names.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void*
 *     (NULL = default stream); every pointer prefixed d_ is DEVICE memory, h_ is host memory.
 *   - every function returns DSP_OK (0) or a negative DSP_E* code; dsp_last_error() gives the
 *     thread-local message.  Nothing falls back to a CPU path: without a GPU calls fail.
 *   - waveform buffers may start at any element-aligned address and dense batches may have any
 *     length: what the fused kernels cannot take directly (a start that is not 16-byte aligned, a
 *     dense length that is not a multiple of 4) is viewed as a ragged batch of the aligned buffer
 *     underneath, with offset tables built on the device.
 *   - utterances are concatenated: sample_offsets[B+1] (int64) into the wave buffer,
 *     frame_offsets[B+1] (int64) into the [sum T_b, D] output (row-major, fp32).
 *   - no global mutable state besides the thread-local error string; a plan is immutable after
 *     creation and may be used from several host threads / streams concurrently.
 */
#ifndef DSP_FRONTEND_H
#define DSP_FRONTEND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSP_ABI_VERSION 1

/* status codes */
#define DSP_OK          0
#define DSP_EINVAL     -1   /* bad argument / unsupported configuration */
#define DSP_EHIP       -2   /* HIP runtime error (message has hipGetErrorString) */
#define DSP_ENODEV     -3   /* no usable GPU */

/* waveform sample types */
#define DSP_WAVE_F32    0
#define DSP_WAVE_I16    1   /* what reader.py:80 yields (int16 PCM) */

/* which stage's output dsp_features_batch writes */
#define DSP_OUT_FRAMES   0  /* [sumT, L]        sigproc.framesig      sigproc.py:66-98   */
#define DSP_OUT_MAGSPEC  1  /* [sumT, NFFT/2+1] sigproc.magspec       sigproc.py:136-148 */
#define DSP_OUT_POWSPEC  2  /* [sumT, NFFT/2+1] sigproc.powspec       sigproc.py:151-158 */
#define DSP_OUT_FBANK    3  /* [sumT, M] + energy[sumT]  base.fbank   base.py:18-32      */
#define DSP_OUT_MFCC     4  /* [sumT, C]        base.mfcc             base.py:8-16       */

typedef struct dsp_plan dsp_plan;

/*
 * Host-built tables for one (rate, L, S, NFFT, M, C, preemph, lifter, window) tuple.  The host side
 * evaluates everything the reference evaluates in Python/fp64 per call -- round_half_up of the frame
 * sizes (sigproc.py:55-56,77-78), winfunc(L) (sigproc.py:89), get_filterbanks (base.py:40-58), the
 * DCT-II/ortho matrix (base.py:13, scipy.fftpack.dct) with the lifter (base.py:60-68) folded in --
 * once, in fp64, and hands them over rounded to fp32.
 */
typedef struct dsp_plan_desc {
    int32_t frame_len;        /* L  samples per frame (after round_half_up)                        */
    int32_t frame_step;       /* S  hop in samples                                                 */
    int32_t nfft;             /* NFFT: 2^k or 3*2^k, 16 <= NFFT <= 4096; frames longer are truncated
                                 (sigproc.py:143-147)                                              */
    int32_t nfilt;            /* M  mel filters (0 if the plan is only used up to POWSPEC)         */
    int32_t numcep;           /* C  cepstra kept, C <= M                                           */
    int32_t append_energy;    /* base.py:15 -- column 0 := log(frame energy)                       */
    float   preemph;          /* base.py:22 -- 0 disables the filter                               */
    const float*   h_window;      /* [L]                                                           */
    const int32_t* h_mel_start;   /* [M] first FFT bin of filter j with a stored weight            */
    const int32_t* h_mel_count;   /* [M] number of stored weights                                  */
    const float*   h_mel_weights; /* [sum count] row after row                                     */
    const float*   h_dct;         /* [C, M] row-major, lifter already multiplied in                */
} dsp_plan_desc;

/* ---- library / device plumbing ------------------------------------------------------------- */
int         dsp_abi_version(void);
const char* dsp_last_error(void);
int dsp_device_count(int* n);
int dsp_set_device(int device);
int dsp_get_device(int* device);   /* plans, buffers and workspaces belong to the device current at creation */
int dsp_malloc(void** d_ptr, size_t bytes);
int dsp_free(void* d_ptr);
int dsp_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream);
int dsp_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream);
int dsp_memset(void* d_dst, int value, size_t bytes, void* stream);
int dsp_stream_synchronize(void* stream);

/* ---- host-side geometry (no GPU needed) ---------------------------------------------------- */
/* T = 1 if n <= L else 1 + ceil((n - L) / S)                       sigproc.py:79-82 */
int dsp_frame_count(int64_t n_samples, int32_t frame_len, int32_t frame_step, int64_t* n_frames);
/* frame_offsets[0]=0, frame_offsets[b+1]=frame_offsets[b]+T_b      (h_ pointers)    */
int dsp_frame_offsets(const int64_t* h_sample_offsets, int32_t n_utt, int32_t frame_len,
                      int32_t frame_step, int64_t* h_frame_offsets);

/* ---- plans --------------------------------------------------------------------------------- */
int dsp_plan_create(const dsp_plan_desc* desc, dsp_plan** plan);
int dsp_plan_destroy(dsp_plan* plan);
/* 1 if the plan is served by a specialised fused kernel (NFFT = 512 or 1536), 0 if only by the generic one */
int dsp_plan_has_fast_path(const dsp_plan* plan);
/* testing aid: route the calling THREAD's feature / VAD / pitch calls through the generic kernels even when a
   specialised one applies (thread-local flag: other threads are unaffected; the kernels are independent
   implementations and must agree) */
int dsp_debug_force_generic(int on);
/* The matrix-pipe NFFT = 512 kernel (csrc/kernels_mfma512.h: the DFT, the mel filterbank and the DCT of
   sigproc.py:136-158 / base.py:8-32 as fp16 / bf16 (hi, lo) products on v_mfma_f32_16x16x32) is an OPT-IN path for
   dense batches: on = 1 routes the calling thread's dsp_features_batch(MFCC) / dsp_mfcc_delta_batch calls to it when it
   serves the plan and the batch, 2 to its frame-per-product form (csrc/kernels_mfma512t.h: one register-resident matrix
   per DFT stage, window and twiddle on the vector pipe), 0 keeps them on the vector-pipe kernels, -1 follows the
   environment (DSP_MFMA512=1 / 2). */
int dsp_debug_use_mfma512(int on);
/* bit 0: the plan has tables for kernels_mfma512.h (NFFT = 512, hop 160, <= 47 filters in the supported block pattern);
   bit 1: for kernels_mfma512t.h (NFFT = 512, hop 160, <= 47 filters) */
int dsp_plan_has_mfma512(const dsp_plan* plan);
/* testing aid: device workspaces (index tables, cepstra scratch) the library's pool currently holds */
int dsp_debug_pool_stats(long long* n_buffers, long long* bytes);
/* on = 1: dsp_plan_create on the calling thread runs every host-side table builder (window / twiddle / mel / DCT tables,
   the fused kernels' and the matrix-pipe kernels' operand tables) but keeps the tables in HOST memory and never touches a
   device -- for the sanitizer build (`make -C dsp-speech-recognition_amd/csrc asan`, tests/test_host_asan.py), which runs
   on a machine without a GPU.  Such a plan can only be destroyed; launches with it are refused. */
int dsp_debug_host_dry_run(int on);

/* ---- the hot path -------------------------------------------------------------------------- */
/*
 * y[0]=x[0]; y[n]=x[n]-coeff*x[n-1] per utterance.        sigproc.preemphasis  sigproc.py:178-185
 */
int dsp_preemphasis_batch(const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                          int32_t n_utt, int64_t n_samples_total, float coeff, float* d_out,
                          void* stream);

/*
 * Fused pre-emphasis -> framing*window -> rFFT -> |X|^2/NFFT -> mel -> log -> DCT*lifter -> energy
 * swap, stopping at `out_kind`.  Replaces sigproc.preemphasis/framesig/magspec/powspec and
 * base.fbank/mfcc (sigproc.py:66-98,136-158,178-185; base.py:8-32).
 *   uniform_samples > 0: every utterance has exactly that many samples (offsets may then be NULL).
 *   d_out2: energy[sumT] for DSP_OUT_FBANK, else ignored (may be NULL).
 *   ld_out: row stride of d_out in floats (0 = dense).
 */
int dsp_features_batch(const dsp_plan* plan, const void* d_wave, int wave_dtype,
                       const int64_t* d_sample_offsets, const int64_t* d_frame_offsets,
                       int32_t n_utt, int64_t n_frames_total, int64_t uniform_samples,
                       int out_kind, float* d_out, int64_t ld_out, float* d_out2, void* stream);

/*
 * d[t] = sum_{n=-N..N} n * x[clamp(t+n)] / (2 sum i^2), edge-replicated per utterance.
 *                                                                base.delta  base.py:70-79
 * Reads columns [0,D) of d_in (row stride ld_in), writes columns [0,D) of d_out (row stride ld_out).
 * If d_out_dd != NULL also writes delta(delta(x)) there in the same pass (model.py:76-77 pattern).
 *   uniform_frames > 0: every utterance has exactly that many frames (offsets may be NULL).
 */
int dsp_delta_batch(const float* d_in, int64_t ld_in, const int64_t* d_frame_offsets,
                    int32_t n_utt, int64_t n_frames_total, int64_t uniform_frames, int32_t D,
                    int32_t N, float* d_out, int64_t ld_out, float* d_out_dd, int64_t ld_out_dd,
                    void* stream);

/*
 * BASELINE config 2 in one call: out[sumT, 3C] = mfcc | delta_N | delta_N(delta_N).
 *                                        base.mfcc + base.delta x2   base.py:8-16,70-79
 */
int dsp_mfcc_delta_batch(const dsp_plan* plan, const void* d_wave, int wave_dtype,
                         const int64_t* d_sample_offsets, const int64_t* d_frame_offsets,
                         int32_t n_utt, int64_t n_frames_total, int64_t uniform_samples,
                         int32_t delta_n, float* d_out, void* stream);

/* c[:, n] *= lift[n]                                           base.lifter  base.py:60-68 */
int dsp_scale_columns(float* d_x, int64_t rows, int32_t cols, const float* d_scale, void* stream);

/* ---- endpointing (energy / ZCR) ------------------------------------------------------------ */
/*
 * Per frame of length L / hop S (rectangular window, sizes truncated by the caller as
 * sigproc.to_frames does, sigproc.py:11-19):
 *   amp_sum[t] = sum |x| (or sum x^2)   -> endpoint.get_amplitude = amp_sum / L   endpoint.py:109-126
 *   zcr[t]     = #{ i : x[i]*x[i+1] < 0 }, sign-pair test                         endpoint.py:182-198
 * fp64 accumulation; exact for int16 input.
 */
int dsp_vad_features_batch(const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                           const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total,
                           int64_t uniform_samples, int32_t frame_len, int32_t frame_step,
                           int32_t use_sq, double* d_amp_sum, int32_t* d_zcr, void* stream);

/*
 * The two-threshold state machine, one utterance per thread, fp64 thresholds:
 * endpoint.amplitude_rule (endpoint.py:133-179, use_acr=False), the mh=0.125 retry and the
 * <50-frame fallbacks of endpoint.basic_endpoint_detection (endpoint.py:42-62), endpoint.zcr_rule
 * (endpoint.py:201-220).  cfg_frame/cfg_step are the reference's cfg.frame / cfg.step.
 * Writes frame indices (left2,right2) per utterance to d_endpoints[2*b .. 2*b+1]; the caller maps
 * them to samples with int(left*cfg.step*rate) in fp64 (endpoint.py:64).
 */
int dsp_endpoint_rule_batch(const double* d_amp_sum, const int32_t* d_zcr,
                            const int64_t* d_frame_offsets, int32_t n_utt, int32_t frame_len,
                            double cfg_frame, double cfg_step, int32_t* d_endpoints, void* stream);

/*
 * endpoint.robust_endpoint_detection (endpoint.py:68-92), batched, in two calls.
 *
 * dsp_acr_gate_batch: the autocorrelation gate acr_rule of endpoint.amplitude_rule (endpoint.py:142-144) for EVERY frame
 * of the rectangular to_frames framing (sigproc.py:11-19): d_voiced[g] = 1 when
 *     max_{n in [lag_lo, lag_hi)} sigproc.acr(frame, n) / sigproc.acr(frame, 0) > thresh        (sigproc.py:48-53)
 * -- the caller passes lag_lo = rate // 500, lag_hi = rate // 50, thresh = 0.55 -- accumulated in fp64 (exact for
 * int16 PCM).  Same batch conventions as dsp_vad_features_batch; lags must not exceed frame_len.
 *
 * dsp_endpoint_rule_acr_batch: dsp_endpoint_rule_batch's state machine in the robust form -- amplitude_rule with
 * mh = 0.5 and use_acr=True (a segment only grows over frames whose d_voiced bit is set, endpoint.py:168-170), one
 * pass (no mh = 0.125 retry), then zcr_rule and the < 50-frame fallback (endpoint.py:73-90).
 */
int dsp_acr_gate_batch(const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                       const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total, int64_t uniform_samples,
                       int32_t frame_len, int32_t frame_step, int32_t lag_lo, int32_t lag_hi, double thresh,
                       uint8_t* d_voiced, void* stream);
int dsp_endpoint_rule_acr_batch(const double* d_amp_sum, const int32_t* d_zcr, const uint8_t* d_voiced,
                                const int64_t* d_frame_offsets, int32_t n_utt, int32_t frame_len, double cfg_frame,
                                double cfg_step, int32_t* d_endpoints, void* stream);

/*
 * Batch-layout handle: the index tables a ragged call would otherwise build with a small launch of its own (into a
 * pooled workspace) on EVERY call, built ONCE for a batch shape -- the frame offsets of `frame_len` / `frame_step`
 * framing of n_utt utterances.  dsp_vad_features_layout_batch is dsp_vad_features_batch with those tables: no table
 * launch, no pooled workspace, so the call can be captured into a HIP graph.  (The feature stage of configs[3] has
 * data-dependent offsets -- the trimmed clips -- and keeps its tables in the caller's work buffer instead:
 * dsp_endpoint_layout_segments_batch.)  The tables are built on `stream`; the handle belongs to the current device.
 */
typedef struct dsp_layout dsp_layout;
int dsp_layout_create(const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total, int32_t frame_len,
                      int32_t frame_step, void* stream, dsp_layout** out);
int dsp_layout_destroy(dsp_layout* layout);
int dsp_vad_features_layout_batch(const dsp_layout* layout, const void* d_wave, int wave_dtype,
                                  const int64_t* d_sample_offsets, const int64_t* d_frame_offsets, int32_t use_sq,
                                  double* d_amp_sum, int32_t* d_zcr, void* stream);

/*
 * Endpoint-trimmed copy (fp32 out): utterance b keeps samples [segments[2b], segments[2b+1]) relative
 * to its start and lands at d_dst_offsets[b]; with unit_variance != 0 it is divided by its population
 * standard deviation (zero -> 1), i.e. sig[left:right] -> sklearn scale(with_mean=False) as
 * model.py:62-63 does before feature extraction.  fp64 statistics.
 */
int dsp_trim_scale_batch(const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                         const int64_t* d_segments, const int64_t* d_dst_offsets, int32_t n_utt,
                         int32_t unit_variance, float* d_out, void* stream);

/*
 * configs[3] without the trimmed copy: MFCC | delta | delta-delta of sig[left:right] for every utterance of a ragged
 * batch, READ IN PLACE -- model.py:113-121 (endpoint_detect -> scale -> feature_extract_mfcc) without writing and
 * re-reading an fp32 copy of every clip.  d_segments = int64 [B, 2] (left, right) relative to each utterance's start
 * (dsp_endpoint_layout_batch writes them), d_frame_offsets = the frame prefix of the TRIMMED clips (same call),
 * n_frames_bound >= their total.  unit_variance != 0: the clip counts as divided by its population standard deviation
 * (sklearn scale(with_mean=False), model.py:62-63) -- computed as what that scaling does to the result: it adds
 * -ln(var) to c0 = log(energy) and leaves every other cepstrum, delta and delta-delta unchanged; var comes from fp64
 * sums the MFCC kernel accumulates while it stages the samples.  d_work: caller-owned scratch of
 * dsp_segments_workspace_bytes() bytes (tables, statistics, dense cepstra): no pooled workspace, no allocation, the call
 * is a fixed sequence of three launches on `stream` and can be captured into a HIP graph.
 * Served by the NFFT = 512 AND the NFFT = 1536 fused kernels (the size model.py:74 uses).  delta_n = 0: cepstra only,
 * written straight to d_out [sum T, C] (model.py:66-88 wants delta(3) of the MEAN-REMOVED cepstra, which is
 * dsp_model_finalize_segments_batch's job); with DSP_SEG_UNIT_VARIANCE the c0 column then still lacks its -ln(var) --
 * the statistics stay at the start of d_work and dsp_model_finalize_segments_batch applies the shift as it reads.
 * Returns 1 (not an error) when the plan / buffer is not served in place (no fused kernel, misaligned buffer,
 * unit variance without appendEnergy): use dsp_trim_scale_batch + dsp_mfcc_delta_batch then.
 */
#define DSP_SEG_UNIT_VARIANCE 1   /* flags: the clips count as divided by their standard deviation (see above) */
#define DSP_SEG_TABLES_READY 2    /* flags: d_work already holds the tables (dsp_endpoint_layout_segments_batch) */
int dsp_segments_workspace_bytes(const dsp_plan* plan, int32_t n_utt, int64_t n_frames_bound, size_t* bytes);
int dsp_mfcc_delta_segments_batch(const dsp_plan* plan, const void* d_wave, int wave_dtype,
                                  const int64_t* d_sample_offsets, const int64_t* d_segments,
                                  const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_bound,
                                  int32_t delta_n, int32_t flags, void* d_work, size_t work_bytes,
                                  float* d_out, void* stream);
/*
 * dsp_endpoint_layout_batch (below) for `plan`'s framing that ALSO leaves, in the same launch, everything
 * dsp_mfcc_delta_segments_batch would otherwise build with a launch of its own (frame-group and delta-tile tables of the
 * trimmed clips, zeroed statistics) in d_work: pass DSP_SEG_TABLES_READY to that call then.
 */
int dsp_endpoint_layout_segments_batch(const int32_t* d_endpoints, const int64_t* d_sample_offsets, int32_t n_utt,
                                       double cfg_step, double rate, const int64_t* d_jitter, int64_t* d_segments,
                                       int64_t* d_dst_offsets, int64_t* d_frame_offsets, const dsp_plan* plan,
                                       int64_t n_frames_bound, void* d_work, size_t work_bytes, void* stream);

/*
 * Device-side glue between dsp_endpoint_rule_batch and dsp_trim_scale_batch / dsp_features_batch, so that
 * the endpoint -> trim -> features pipeline of model.py:113-121 needs no host round trip:
 *   d_segments[2b], [2b+1] = int((left * cfg.step) * rate), int((right * cfg.step) * rate) in fp64, that
 *                            order, truncated (endpoint.py:64), clipped to the clip length as numpy slicing
 *                            sig[left:right] does (model.py:62); with d_jitter != NULL the per-utterance
 *                            offsets d_jitter[2b] (added to left, i.e. -s_l) and d_jitter[2b+1] (+s_r) of
 *                            model.py:54-60 are applied first and negatives raised to 0;
 *   d_dst_offsets[B+1]     = exclusive prefix of the trimmed lengths;
 *   d_frame_offsets[B+1]   = exclusive prefix of their frame counts at (frame_len, frame_step).
 * Ragged feature calls that follow may pass an upper bound as n_frames_total (e.g. the frame count of the
 * untrimmed clips): kernels take the true totals from d_frame_offsets.
 */
int dsp_endpoint_layout_batch(const int32_t* d_endpoints, const int64_t* d_sample_offsets, int32_t n_utt,
                              double cfg_step, double rate, int32_t frame_len, int32_t frame_step,
                              const int64_t* d_jitter, int64_t* d_segments, int64_t* d_dst_offsets,
                              int64_t* d_frame_offsets, void* stream);

/* ---- model.py glue behind the feature call ----------------------------- */
/*
 * What model.py:66-88 does to mfcc0 = mfcc(...) of every utterance, and the [200, B, 39] layout of
 * model.py:35-50,131-135, in one launch:
 *   x  = mfcc0 - mean(mfcc0)             scalar mean of the utterance's [T, C] block   model.py:75
 *   d1 = delta(x, N); d2 = delta(d1, N)  base.delta, edge replicated                  model.py:76-77
 *   z  = (x - mean_c) / std_c            per coefficient, population std, 0 -> 1       model.py:78
 *   d_out[t, b, 0:C | C:2C | 2C:3C] = z | d1 | d2 for t < min(T_b, max_len), zero rows up to max_len
 *   d_len0[b] = min(T_b, max_len)
 * d_mfcc: [sum T_b, C] (row stride ld_in, 0 = dense); d_out: [max_len, n_utt, 3C] fp32.
 * Statistics are accumulated in fp64.
 */
int dsp_model_finalize_batch(const float* d_mfcc, int64_t ld_in, const int64_t* d_frame_offsets,
                             int32_t n_utt, int32_t C, int32_t N, int32_t max_len, float* d_out,
                             int32_t* d_len0, void* stream);

/*
 * Optional amplitude stream of model.py:97-101 (cfg.use_timefeat), batched: d_amp_sum = per-frame sums of
 * |x| of the trimmed, scaled clips (dsp_vad_features_batch at frame_len = int(rate * cfg.frame)), one run of
 * frames per utterance.  Writes d_out[max_len, n_utt, 2]: column 0 = the z-scored frame amplitude
 * (endpoint.amplitude_feature, endpoint.py:128-131, through sklearn scale: population std, 0 -> 1), column 1 =
 * its first difference (model.py:29-33, T - 1 rows), both zero padded / truncated to max_len (model.py:35-50).
 */
/* dsp_model_finalize_batch for cepstra of segments read in place (dsp_mfcc_delta_segments_batch, delta_n = 0, same
   d_segments and d_work): with the unit-variance statistics in d_work, c0 gets its -ln(var) (model.py:62-63) at every
   read -- frames of exactly zero energy keep ln(eps), base.py:26 -- before the mean, the deltas and the z-score. */
int dsp_model_finalize_segments_batch(const float* d_mfcc, int64_t ld_in, const int64_t* d_frame_offsets,
                                      const int64_t* d_segments, const void* d_work, int32_t n_utt, int32_t C, int32_t N,
                                      int32_t max_len, float* d_out, int32_t* d_len0, void* stream);
int dsp_model_timefeat_batch(const double* d_amp_sum, const int64_t* d_frame_offsets, int32_t n_utt,
                             int32_t frame_len, int32_t max_len, float* d_out, void* stream);

/* ---- pitch scores ------------------------------------------------------ */
/*
 * Per frame of the (already 10 kHz) signal, rectangular frames of frame_len / hop frame_step as
 * sigproc.to_frames cuts them:
 *   centre clipping at the median of the non-negative samples (if center_clip != 0)
 *                                                         pitch.center_clip      pitch.py:145-155
 *   y = convolve(frame, taps)[:frame_len], f = |y|        sigproc.window         sigproc.py:22-46
 *   scores[t, n - lag_min] = sum_i f[i] f[i+n] / (frame_len - n), lag_min <= n < lag_max
 *                                                         sigproc.acr            sigproc.py:48-53
 * i.e. pitch.pitch_detect_frame_sr (pitch.py:112-132) for every frame of pitch.pitch_detect_sr
 * (pitch.py:96-107).  d_sig: fp32; d_taps: frame_len complex taps (re, im interleaved), built by the
 * host in fp64; d_scores: [sum T_b, lag_max - lag_min] fp32.  frame_len <= 1024.
 */
int dsp_pitch_scores_batch(const float* d_sig, const int64_t* d_sample_offsets,
                           const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total,
                           int64_t uniform_samples, int32_t frame_len, int32_t frame_step,
                           const float* d_taps, int32_t center_clip, int32_t lag_min, int32_t lag_max,
                           float* d_scores, void* stream);

/*
 * The tracker behind those scores, for a whole batch (one wavefront per utterance, fp64): pitch.smooth in place
 * (pitch.py:157-164: running mean over rows [i - 2, i + 2) of which the first two are already smoothed; the
 * reference's end-of-utterance window and its empty mean for a one-frame utterance included), pitch.max_pitch
 * (pitch.py:166-172: first arg-max, 1 / (1e-4 (bias + idx))) and the two octave-repair sweeps of
 * pitch.robust_max_pitch (pitch.py:191-206).  d_scores: [sum T_b, n_lags] fp32 as dsp_pitch_scores_batch writes
 * them; d_pitch: [sum T_b] fp64, Hz per frame.  degree must be 2 (the only value the reference uses).
 */
int dsp_pitch_track_batch(const float* d_scores, const int64_t* d_frame_offsets, int32_t n_utt, int32_t n_lags,
                          int32_t bias, int32_t degree, double* d_pitch, void* stream);

/*
 * The tracker's parts on fp64 score rows [sum T_b, n_lags] (what the reference's helpers take): flags bit 0 =
 * pitch.smooth(g, degree) in place (pitch.py:157-164), bit 1 = pitch.max_pitch (pitch.py:166-172) into d_pitch, bit 2 =
 * the octave-repair sweeps of pitch.robust_max_pitch (pitch.py:191-206; needs bit 1).  d_rows is overwritten by the
 * smoothed rows when bit 0 is set.
 */
int dsp_pitch_rows_batch(double* d_rows, const int64_t* d_frame_offsets, int32_t n_utt, int32_t n_lags, int32_t bias,
                         int32_t degree, int32_t flags, double* d_pitch, void* stream);

/*
 * Device-side glue of the optional streams of model.py:90-101 (no clip leaves the device):
 *   dsp_resample_layout_batch  from the sample offsets of a batch: with dst_rate > 0 the lengths preprocess.downsampling
 *                              (preprocess.py:21-28) leaves of every clip and their exclusive prefix d_dst_offsets[B+1];
 *                              always the exclusive prefix d_frame_offsets[B+1] of the frame counts of the (decimated)
 *                              clips at (frame_len, frame_step) (sigproc.py:79-82).  dst_rate = 0: frame offsets only.
 *   dsp_decimate_batch         the kept samples themselves (the k-th kept index is the first i with
 *                              i * dst_rate / src_rate > k - 1 + 1e-8, the reference's fp64 test); n_out_bound is an
 *                              upper bound of the output length (e.g. the input length), the true total is
 *                              d_dst_offsets[n_utt].
 *   dsp_model_pitchfeat_batch  [max_len, B, 2] = pitch / 150 and its first difference, zero padded (model.py:90-95,
 *                              35-50), from dsp_pitch_track_batch's d_pitch.
 */
int dsp_resample_layout_batch(const int64_t* d_src_offsets, int32_t n_utt, int64_t src_rate, int64_t dst_rate,
                              int32_t frame_len, int32_t frame_step, int64_t* d_dst_offsets, int64_t* d_frame_offsets,
                              void* stream);
int dsp_decimate_batch(const float* d_in, const int64_t* d_src_offsets, const int64_t* d_dst_offsets, int32_t n_utt,
                       int64_t n_out_bound, int64_t src_rate, int64_t dst_rate, float* d_out, void* stream);
int dsp_model_pitchfeat_batch(const double* d_pitch, const int64_t* d_frame_offsets, int32_t n_utt, int32_t max_len,
                              float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSP_FRONTEND_H */
