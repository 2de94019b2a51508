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
