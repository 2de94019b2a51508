// Pitch scores of pitch.pitch_detect_sr (pitch.py:96-132), one workgroup per frame of the 10 kHz
// signal:
//   centre clipping at the median of the non-negative samples   pitch.center_clip   pitch.py:145-155
//   causal complex band-pass FIR truncated to the frame, |.|     sigproc.window      sigproc.py:22-46
//   autocorrelation at lags lag_min .. lag_max - 1               sigproc.acr         sigproc.py:48-53
// The FIR taps (inverse FFT of the ideal band, times the window, times 2 pi) are built on the host
// in fp64 and handed over as float2[L].  Everything else (smoothing across frames, arg-max, octave
// repair) is O(T * lags) sequential host logic in the Python mirror, as in the reference.
#pragma once

#include "dsp_common.h"

#define PITCH_THREADS 256
#define PITCH_MAX_L 1024

// frames: rectangular, frame t = samples [t S, t S + L) of the utterance, zero padded (to_frames).
__global__ __launch_bounds__(PITCH_THREADS) void pitch_scores_kernel(
    const float* __restrict__ sig, BatchGeom bg, int32_t L, int32_t S, int32_t P /* pow2 >= L */,
    const float2* __restrict__ taps, int32_t do_clip, int32_t lag_min, int32_t n_lags,
    float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float smem_p[];
    float* key = smem_p;                                   // [P]  sort keys
    float* cl = key + P;                                   // [L]  clipped frame, later |filtered|
    float2* h = reinterpret_cast<float2*>(cl + ((L + 3) & ~3));   // [L]  FIR taps
    float* f = reinterpret_cast<float*>(h + L);            // [L]  |filtered frame|
    __shared__ int s_m;
    const int tid = threadIdx.x;
    const int64_t g = blockIdx.x;
    int32_t utt;
    int64_t t, s0, nsamp;
    if (bg.uniform_frames <= 0 && g >= bg.frame_off[bg.n_utt]) return;   // the grid may be sized by an upper bound of the frame count
    dsp_locate(bg, g, utt, t, s0, nsamp);
    const int64_t first = t * (int64_t)S;
    if (tid == 0) s_m = 0;
    __syncthreads();
    // ---- load the frame; sort keys: non-negative samples, everything else +inf ----
    int mloc = 0;
    for (int i = tid; i < P; i += PITCH_THREADS) {
        float x = 0.f;
        const bool in = i < L;
        if (in && first + i < nsamp) x = sig[s0 + first + i];
        if (in) cl[i] = x;
        const bool nn = in && x >= 0.f;
        key[i] = nn ? x : __int_as_float(0x7f800000);
        mloc += nn ? 1 : 0;
    }
    for (int i = tid; i < L; i += PITCH_THREADS) h[i] = taps[i];
    if (do_clip) {
        if (mloc) atomicAdd(&s_m, mloc);
        __syncthreads();
        // bitonic sort, ascending
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P; i += PITCH_THREADS) {
                    const int p = i ^ j;
                    if (p > i) {
                        const float a = key[i], b = key[p];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { key[i] = b; key[p] = a; }
                    }
                }
                __syncthreads();
            }
        }
        const int m = s_m;
        // numpy.median: mean of the two middle order statistics; no non-negative sample -> NaN
        const float med = m > 0 ? 0.5f * (key[(m - 1) >> 1] + key[m >> 1]) : __int_as_float(0x7fc00000);
        for (int i = tid; i < L; i += PITCH_THREADS) {
            const float x = cl[i];
            cl[i] = x > med ? x - med : (x < -med ? x + med : 0.f);
        }
    }
    __syncthreads();
    // ---- y[k] = sum_{j <= k} c[j] h[k - j]; outputs k and L - 1 - k share a thread (L + 1 taps each) ----
    for (int k1 = tid; 2 * k1 < L; k1 += PITCH_THREADS) {
        const int k2 = L - 1 - k1;
        float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
        int j = 0;
        for (; j <= k1; ++j) {
            const float c = cl[j];
            const float2 ha = h[k1 - j], hb = h[k2 - j];
            ar = fmaf(c, ha.x, ar); ai = fmaf(c, ha.y, ai);
            br = fmaf(c, hb.x, br); bi = fmaf(c, hb.y, bi);
        }
        for (; j <= k2; ++j) {
            const float c = cl[j];
            const float2 hb = h[k2 - j];
            br = fmaf(c, hb.x, br); bi = fmaf(c, hb.y, bi);
        }
        f[k1] = sqrtf(fmaf(ar, ar, ai * ai));
        if (k2 != k1) f[k2] = sqrtf(fmaf(br, br, bi * bi));
    }
    __syncthreads();
    // ---- autocorrelation, one lag per thread ----
    for (int q = tid; q < n_lags; q += PITCH_THREADS) {
        const int n = lag_min + q;
        float acc = 0.f;
        if (n == 0) {
            for (int i = 0; i < L; ++i) acc = fmaf(f[i], f[i], acc);
            acc /= (float)L;
        } else if (n < L) {
            for (int i = 0; i + n < L; ++i) acc = fmaf(f[i], f[i + n], acc);
            acc /= (float)(L - n);
        } else {
            acc = __int_as_float(0x7fc00000);   // numpy: sum of an empty product / 0 -> nan
        }
        scores[g * n_lags + q] = acc;
    }
}

// Register-blocked variant, one wavefront per frame.  Every lane owns W consecutive FIR outputs at
// the bottom of the frame and W at the top (so all lanes do the same work), keeps the 2 W samples it
// needs in a register ring and reads one new sample per chunk and tap: 3 LDS reads per 4 W
// multiply-adds instead of 3 per 4.  Indices below zero fall into a zeroed guard band, so the tap
// loop is uniform and branch free.  The autocorrelation is blocked the same way (4 lags per lane).
// The clip level (median of the non-negative samples) comes from a 31-step bisection on the float bit
// patterns held in registers (ballot + popcount per step): no sort, no LDS, no barrier.
// Requires lag_min % 4 == 0, n_lags <= 256, ceil(L / (2 W)) <= 64 (so L <= 512).
#define PITCH2_GUARD 256   // zeros behind f[] (>= lag_max + 4)

template <int W>
__global__ __launch_bounds__(64) void pitch_scores_kernel_v2(
    const float* __restrict__ sig, BatchGeom bg, int32_t L, int32_t S, int32_t P, const float2* __restrict__ taps,
    int32_t do_clip, int32_t lag_min, int32_t n_lags, float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float smem_p[];
    const int Lp = (L + W - 1) / W * W;                    // outputs are produced in chunks of W
    float* cl0 = smem_p;                                   // [Lp zeros][Lp samples]: index i lives at cl0[Lp + i]
    float2* h = reinterpret_cast<float2*>(cl0 + 2 * Lp);   // [Lp] taps, zero beyond L
    float* f = reinterpret_cast<float*>(h + Lp);           // [Lp + PITCH2_GUARD]
    (void)P;
    const int lane = threadIdx.x;
    const int64_t g = blockIdx.x;
    int32_t utt;
    int64_t t, s0, nsamp;
    if (bg.uniform_frames <= 0 && g >= bg.frame_off[bg.n_utt]) return;   // the grid may be sized by an upper bound of the frame count
    dsp_locate(bg, g, utt, t, s0, nsamp);
    const int64_t first = t * (int64_t)S;
    float* cl = cl0 + Lp;
    // frame samples: lane owns elements lane + 64 r (r < 8, L <= 512); order statistics are found on
    // the bit patterns (non-negative floats order like unsigned integers), everything else is 0xffffffff
    float xr[8];
    uint32_t kb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = lane + 64 * r;
        float x = 0.f;
        const bool in = i < L;
        if (in && first + i < nsamp) x = sig[s0 + first + i];
        xr[r] = x;
        const bool nn = in && x >= 0.f;
        kb[r] = nn ? __float_as_uint(x + 0.f) : 0xffffffffu;        // x + 0 turns -0 into +0
    }
    for (int i = lane; i < Lp; i += 64) {
        cl0[i] = 0.f;                                      // guard band: samples before the frame
        if (i >= L) cl[i] = 0.f;
        h[i] = i < L ? taps[i] : make_float2(0.f, 0.f);
    }
    for (int i = lane; i < Lp + PITCH2_GUARD; i += 64) f[i] = 0.f;
    float med = 0.f;
    if (do_clip) {
        // wave-wide count of keys below a candidate: one v_cmp per register, popcount of the masks
        auto count_below = [&](uint32_t cand) {
            int c = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) c += __popcll(__ballot(kb[r] < cand));
            return c;
        };
        int m = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) m += __popcll(__ballot(kb[r] != 0xffffffffu));
        if (m > 0) {
            // k-th smallest (0-based): the largest v with fewer than k + 1 keys below it, bit by bit
            const int k1 = (m - 1) >> 1, k2 = m >> 1;
            uint32_t v1 = 0;
            for (int bit = 30; bit >= 0; --bit) {
                const uint32_t cand = v1 | (1u << bit);
                if (count_below(cand) <= k1) v1 = cand;
            }
            uint32_t v2 = v1;
            if (k2 != k1 && count_below(v1 + 1) < k2 + 1) {
                // the next distinct key above v1
                uint32_t mn = 0xffffffffu;
#pragma unroll
                for (int r = 0; r < 8; ++r) mn = (kb[r] > v1 && kb[r] < mn) ? kb[r] : mn;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const uint32_t other = (uint32_t)__shfl_xor((int)mn, o, 64);
                    mn = other < mn ? other : mn;
                }
                v2 = mn;
            }
            med = 0.5f * (__uint_as_float(v1) + __uint_as_float(v2));     // numpy.median
        } else {
            med = __int_as_float(0x7fc00000);              // no non-negative sample: NaN clip level
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int i = lane + 64 * r;
        if (i < L) {
            const float x = xr[r];
            cl[i] = do_clip ? (x > med ? x - med : (x < -med ? x + med : 0.f)) : x;
        }
    }
    __syncthreads();
    // ---- FIR: y[k] = sum_m h[m] c[k - m]; lane q owns outputs [W q, W q + W) and [Lp - W q - W, Lp - W q) ----
    const int nchunk = (Lp / W + 1) / 2;
    if (lane < nchunk) {
        const int kl = W * lane, kh = Lp - W * (lane + 1);
        float wl[W], wh[W];                                // ring: the sample at position p sits in slot p % W
        float alr[W], ali[W], ahr[W], ahi[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            wl[e] = cl[kl + e];
            wh[e] = cl[kh + e];
            alr[e] = ali[e] = ahr[e] = ahi[e] = 0.f;
        }
        for (int m0 = 0; m0 < Lp; m0 += W) {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const int m = m0 + j;
                if (m > 0) {                               // position k0 - m enters slot (-m) % W == (W - j) % W
                    wl[(W - j) % W] = cl[kl - m];
                    wh[(W - j) % W] = cl[kh - m];
                }
                const float2 hm = h[m];
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    const float vl = wl[(e - j + W) % W], vh = wh[(e - j + W) % W];
                    alr[e] = fmaf(hm.x, vl, alr[e]); ali[e] = fmaf(hm.y, vl, ali[e]);
                    ahr[e] = fmaf(hm.x, vh, ahr[e]); ahi[e] = fmaf(hm.y, vh, ahi[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < W; ++e) {
            if (kl + e < L) f[kl + e] = sqrtf(fmaf(alr[e], alr[e], ali[e] * ali[e]));
            if (kh + e < L) f[kh + e] = sqrtf(fmaf(ahr[e], ahr[e], ahi[e] * ahi[e]));
        }
    }
    __syncthreads();
    // ---- autocorrelation: lane q owns lags lag_min + 4 q .. + 3; f is zero beyond L ----
    const int nq = (n_lags + 3) / 4;
    for (int q = lane; q < nq; q += 64) {
        const int n0 = lag_min + 4 * q;                    // multiple of 4
        float w[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 3; ++e) w[e] = f[n0 + e];      // positions n0 .. n0 + 2 in slots 0..2
        for (int i0 = 0; i0 < L; i0 += 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j;                      // window positions i + n0 .. i + n0 + 3
                w[(j + 3) % 4] = f[i + n0 + 3];
                const float a = i < L ? f[i] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(a, w[(j + e) % 4], acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + e;
            if (4 * q + e < n_lags) {
                float v;
                if (n == 0 || n < L) v = acc[e] / (float)(L - n);
                else v = __int_as_float(0x7fc00000);
                scores[g * n_lags + 4 * q + e] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The tracker behind the scores: pitch.smooth (pitch.py:157-164), pitch.max_pitch (pitch.py:166-172) and the two
// octave-repair sweeps of pitch.robust_max_pitch (pitch.py:191-206) for a whole batch, one wavefront per utterance.
// Everything is sequential over the frames of an utterance in the reference, and two of its habits must be kept:
//   * smooth() works IN PLACE: row i becomes the mean of rows [max(i - 2, 0), right) where rows below i are already
//     smoothed and `right = i + 2 if i + 2 < T else T - 1` (so the last rows average over a window that EXCLUDES
//     the last row, and a one-frame utterance averages over nothing: NaN, whose arg-max is index 0);
//   * numpy's mean adds the rows in order and divides once; arg-max takes the first maximum (the first NaN if any).
// fp64 throughout, as the reference (its scores are this library's fp32 scores promoted to double).
// ------------------------------------------------------------------------------------------------
#define PITCH_TRACK_LDS_FRAMES 2048   // pitch values of an utterance kept in LDS for the repair sweeps (global memory beyond)

__device__ __forceinline__ void pitch_argmax_combine(double& v, int& ix, double ov, int oix) {
    // numpy.argmax order: a NaN beats everything, then the larger value, then the smaller index
    const bool vn = v != v, on = ov != ov;
    const bool take = (on && !vn) || (on == vn && (ov > v || (ov == v && oix < ix))) || (on && vn && oix < ix);
    if (take) { v = ov; ix = oix; }
}

__global__ __launch_bounds__(64) void pitch_track_kernel(const float* __restrict__ scores,
                                                         const int64_t* __restrict__ frame_off, int32_t n_lags,
                                                         int32_t bias, double* __restrict__ pitch) {
    __shared__ double s_pitch[PITCH_TRACK_LDS_FRAMES];
    const int u = blockIdx.x, lane = threadIdx.x;
    const int64_t base = frame_off[u];
    const int T = (int)(frame_off[u + 1] - base);
    if (T <= 0) return;
    const float* sc = scores + base * n_lags;
    double* out = pitch + base;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    double p2[4], p1[4];                    // smoothed rows i - 2 and i - 1 (lags lane, lane + 64, ...)
#pragma unroll
    for (int k = 0; k < 4; ++k) p2[k] = p1[k] = 0.0;
    for (int i = 0; i < T; ++i) {
        const int left = i - 2 >= 0 ? i - 2 : 0;
        const int right = i + 2 < T ? i + 2 : T - 1;          // exclusive
        const int cnt = right - left;
        double cur[4];
        double bv = -__longlong_as_double(0x7ff0000000000000ll);   // -inf, index "none": loses to every real candidate
        int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lag = lane + 64 * k;
            double acc = qnan;
            if (lag < n_lags && cnt > 0) {
                bool have = false;
                acc = 0.0;
                for (int r = left; r < right; ++r) {          // rows in order, as numpy's add.reduce over axis 0
                    const double v = r == i - 2 ? p2[k] : (r == i - 1 ? p1[k] : (double)sc[(int64_t)r * n_lags + lag]);
                    acc = have ? acc + v : v;
                    have = true;
                }
                acc = acc / (double)cnt;
            }
            cur[k] = acc;
            if (lag < n_lags) pitch_argmax_combine(bv, bi, acc, lag);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            pitch_argmax_combine(bv, bi, ov, oi);
        }
        if (lane == 0) {
            const double p = 1.0 / (0.0001 * (double)(bias + bi));        // pitch.py:169-170
            if (i < PITCH_TRACK_LDS_FRAMES) s_pitch[i] = p; else out[i] = p;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { p2[k] = p1[k]; p1[k] = cur[k]; }
    }
    __syncthreads();
    if (lane == 0) {
        auto get = [&](int i) { return i < PITCH_TRACK_LDS_FRAMES ? s_pitch[i] : out[i]; };
        auto put = [&](int i, double v) { if (i < PITCH_TRACK_LDS_FRAMES) s_pitch[i] = v; else out[i] = v; };
        const double C = 50.0;
        for (int i = 1; i < T; ++i) {                          // pitch.py:199-201
            const double p = get(i);
            if (fabs(2.0 * p - get(i - 1)) < C && p < 170.0) put(i, 2.0 * p);
        }
        for (int i = T - 2; i > 0; --i) {                      // pitch.py:202-204
            const double p = get(i);
            if (fabs(2.0 * p - get(i + 1)) < C && p < 170.0) put(i, 2.0 * p);
        }
    }
    __syncthreads();
    for (int i = lane; i < T && i < PITCH_TRACK_LDS_FRAMES; i += 64) out[i] = s_pitch[i];
}

// ------------------------------------------------------------------------------------------------
// Device-side glue of the optional streams of model.py:90-101, so that ModelFeatureBatch never brings the trimmed
// clips to the host: decimation to 10 kHz (preprocess.downsampling, preprocess.py:21-28), frame offsets of a
// framing on the device, the [max_len, B, 2] pitch stream, and the tracker's parts for fp64 score rows.
// ------------------------------------------------------------------------------------------------

// preprocess.py:21-28 keeps sample i whenever  i * dst / src > kept - 1 + 1e-8  (Python: exact integer product, one
// fp64 division).  For a decimation (dst < src) the k-th kept index is the smallest i that passes the test for tick k.
__device__ __forceinline__ bool dsp_dec_pass(int64_t i, int64_t k, int64_t src, int64_t dst) {
    return (double)(i * dst) / (double)src > (double)(k - 1) + 1e-8;
}
__device__ __forceinline__ int64_t dsp_dec_index(int64_t k, int64_t src, int64_t dst) {
    int64_t i = (int64_t)floor(((double)(k - 1) + 1e-8) * (double)src / (double)dst);
    if (i < 0) i = 0;
    while (!dsp_dec_pass(i, k, src, dst)) ++i;
    while (i > 0 && dsp_dec_pass(i - 1, k, src, dst)) --i;
    return i;
}
__device__ __forceinline__ int64_t dsp_dec_count(int64_t n, int64_t src, int64_t dst) {   // samples kept of n
    if (n <= 0) return 0;
    int64_t k = (int64_t)floor((double)((n - 1) * dst) / (double)src) + 1;
    while (dsp_dec_index(k, src, dst) < n) ++k;
    while (k > 0 && dsp_dec_index(k - 1, src, dst) >= n) --k;
    return k;
}

// One 1024-thread block: lengths n_b = src_off[b+1] - src_off[b]  ->  (optional) decimated lengths and their exclusive
// prefix dst_off, and the exclusive prefix frame_off of the frame counts of the (decimated) lengths at (L, S)
// (sigproc.py:79-82).  dst_rate == 0: no decimation (frame offsets of another framing of the same clips).
__global__ __launch_bounds__(1024) void resample_layout_kernel(const int64_t* __restrict__ src_off, int32_t n_utt,
                                                               int64_t src_rate, int64_t dst_rate, int32_t L, int32_t S,
                                                               int64_t* __restrict__ dst_off, int64_t* __restrict__ frame_off) {
    __shared__ int64_t part_s[16], part_f[16];
    const int tid = threadIdx.x;
    const int per = (n_utt + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, n_utt);
    int64_t sum_s = 0, sum_f = 0;
    for (int b = lo; b < hi; ++b) {
        int64_t n = src_off[b + 1] - src_off[b];
        if (dst_rate > 0) n = dsp_dec_count(n, src_rate, dst_rate);
        sum_s += n;
        sum_f += n <= L ? 1 : 1 + (n - L + S - 1) / S;
    }
    const int64_t in_s = dsp_wave_scan_i64(sum_s), in_f = dsp_wave_scan_i64(sum_f);
    if ((tid & 63) == 63) { part_s[tid >> 6] = in_s; part_f[tid >> 6] = in_f; }
    __syncthreads();
    int64_t base_s = in_s - sum_s, base_f = in_f - sum_f;
    for (int w = 0; w < (tid >> 6); ++w) { base_s += part_s[w]; base_f += part_f[w]; }
    for (int b = lo; b < hi; ++b) {
        int64_t n = src_off[b + 1] - src_off[b];
        if (dst_rate > 0) n = dsp_dec_count(n, src_rate, dst_rate);
        if (dst_off) dst_off[b] = base_s;
        frame_off[b] = base_f;
        base_s += n;
        base_f += n <= L ? 1 : 1 + (n - L + S - 1) / S;
    }
    if (hi == n_utt && lo < hi) {
        if (dst_off) dst_off[n_utt] = base_s;
        frame_off[n_utt] = base_f;
    }
    if (n_utt == 0 && tid == 0) { if (dst_off) dst_off[0] = 0; frame_off[0] = 0; }
}

// out[dst_off[b] + k] = in[src_off[b] + index of the k-th kept sample]  (preprocess.py:21-28), one thread per output
__global__ __launch_bounds__(256) void decimate_gather_kernel(const float* __restrict__ in, const int64_t* __restrict__ src_off,
                                                              const int64_t* __restrict__ dst_off, int32_t n_utt,
                                                              int64_t src_rate, int64_t dst_rate, float* __restrict__ out) {
    const int64_t total = dst_off[n_utt];
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int32_t u = dsp_find_utt(dst_off, n_utt, o);
        out[o] = in[src_off[u] + dsp_dec_index(o - dst_off[u], src_rate, dst_rate)];
    }
}

// model.py:90-95: out[t, b, 0] = pitch[t] / 150, out[t, b, 1] = their first difference (T - 1 rows), zero padded /
// truncated to max_len rows (model.py:35-50).  One wave per utterance.
__global__ __launch_bounds__(64) void pitchfeat_finalize_kernel(const double* __restrict__ pitch,
                                                                const int64_t* __restrict__ frame_off, int32_t n_utt,
                                                                int32_t max_len, float* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t base = frame_off[b];
    const int T = (int)(frame_off[b + 1] - base);
    const double* p = pitch + base;
    for (int t = lane; t < max_len; t += 64) {
        float z0 = 0.f, z1 = 0.f;
        if (t < T) {
            const double a = p[t] / 150.0;
            z0 = (float)a;
            if (t + 1 < T) z1 = (float)(p[t + 1] / 150.0 - a);
        }
        float* o = out + ((int64_t)t * n_utt + b) * 2;
        o[0] = z0;
        o[1] = z1;
    }
}

// The tracker's parts on fp64 score rows (the host helpers pitch.smooth / max_pitch / robust_max_pitch are batches of
// one through this): flags bit 0 = smooth in place over rows [i - degree, i + degree) (pitch.py:157-164: the rows
// before i are already smoothed; the window's end is exclusive and stops at T - 1; an empty window is a NaN row),
// bit 1 = arg-max -> 1 / (1e-4 (bias + idx)) (pitch.py:166-172), bit 2 = the two octave-repair sweeps
// (pitch.py:191-206).  `rows` is overwritten by the smoothed rows when bit 0 is set.  One wave per utterance.
__global__ __launch_bounds__(64) void pitch_rows_kernel(double* __restrict__ rows, const int64_t* __restrict__ frame_off,
                                                        int32_t n_lags, int32_t bias, int32_t degree, int32_t flags,
                                                        double* __restrict__ pitch) {
    const int u = blockIdx.x, lane = threadIdx.x;
    const int64_t base = frame_off[u];
    const int T = (int)(frame_off[u + 1] - base);
    if (T <= 0) return;
    double* g = rows + base * n_lags;
    double* out = pitch ? pitch + base : nullptr;
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    for (int i = 0; i < T; ++i) {
        double bv = -__longlong_as_double(0x7ff0000000000000ll);
        int bi = 0x7fffffff;
        for (int lag = lane; lag < n_lags; lag += 64) {
            double cur = g[(int64_t)i * n_lags + lag];
            if (flags & 1) {
                const int left = i - degree >= 0 ? i - degree : 0;
                const int right = i + degree < T ? i + degree : T - 1;      // exclusive
                double acc = qnan;
                bool have = false;
                for (int r = left; r < right; ++r) {                         // rows in order (rows < i: already smoothed, same lane wrote them)
                    const double v = g[(int64_t)r * n_lags + lag];
                    acc = have ? acc + v : v;
                    have = true;
                }
                cur = have ? acc / (double)(right - left) : qnan;
                g[(int64_t)i * n_lags + lag] = cur;
            }
            pitch_argmax_combine(bv, bi, cur, lag);
        }
        if (flags & 2) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                pitch_argmax_combine(bv, bi, ov, oi);
            }
            if (lane == 0) out[i] = 1.0 / (0.0001 * (double)(bias + bi));
        }
    }
    if ((flags & 4) && lane == 0) {
        const double C = 50.0;
        for (int i = 1; i < T; ++i) {
            const double p = out[i];
            if (fabs(2.0 * p - out[i - 1]) < C && p < 170.0) out[i] = 2.0 * p;
        }
        for (int i = T - 2; i > 0; --i) {
            const double p = out[i];
            if (fabs(2.0 * p - out[i + 1]) < C && p < 170.0) out[i] = 2.0 * p;
        }
    }
}
