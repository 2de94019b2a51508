// Shared device/host helpers for the MI355X speech feature front-end (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dsp_frontend.h"

#define DSP_EPS_F32 2.220446049250313e-16f  // numpy.finfo(float).eps, base.py:26,30 (normal in fp32)

struct dsp_plan {
    int32_t L, S, nfft, K, M, C, append_energy;
    int32_t lfft;       // min(L, nfft): samples of a frame that reach the FFT (sigproc.py:143-147)
    int32_t mel_nnz;
    float preemph;
    // device tables
    float* d_window;        // [L]
    float2* d_twiddle;      // [nfft/2 + 1]  exp(-2 pi i k / nfft), computed in fp64 on the host
    int32_t* d_mel_start;   // [M]
    int32_t* d_mel_count;   // [M]
    int32_t* d_mel_off;     // [M] prefix offsets into d_mel_w
    float* d_mel_w;         // [mel_nnz]
    float* d_dct;           // [C, M]
    void* d_fast;           // tables of the specialised NFFT=512 kernel (NULL if not applicable)
    void* d_fast1536;       // tables of the specialised NFFT=1536 kernel (NULL if not applicable)
    void* d_mfma;           // tables of the matrix-pipe NFFT=512 kernel, kernels_mfma512.h (NULL if not applicable)
    void* d_mfmat;          // tables of its frame-per-product form, kernels_mfma512t.h (NULL if not applicable)
    int device;
    int dry_run;            // tables live in HOST memory (dsp_debug_host_dry_run): the plan can be destroyed, nothing else
};

// Table memory of a plan.  Normally device memory; under dsp_debug_host_dry_run(1) (the sanitizer build's CPU tests:
// `make asan`, tests/test_host_asan.py) plain host memory, so that every host-side table builder runs -- and is checked
// by ASan / UBSan -- on a machine without a GPU.  Plans made that way carry `dry_run` and refuse every launch.
extern thread_local int g_host_dry_run;
static inline hipError_t dsp_table_alloc_copy(void** d, const void* h, size_t bytes) {
    *d = nullptr;
    if (g_host_dry_run) {
        *d = malloc(bytes ? bytes : 1);
        if (!*d) return hipErrorOutOfMemory;
        memcpy(*d, h, bytes);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(d, bytes);
    if (e != hipSuccess) return e;
    e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(*d); *d = nullptr; }
    return e;
}
static inline void dsp_table_free(void* d, int dry_run) {
    if (!d) return;
    if (dry_run) free(d); else (void)hipFree(d);
}

// Batch geometry handed to every kernel by value.
struct BatchGeom {
    const int64_t* sample_off;  // [B+1] or nullptr when uniform
    const int64_t* frame_off;   // [B+1] or nullptr when uniform
    int64_t uniform_samples;    // > 0: all utterances have this many samples
    int64_t uniform_frames;     // frames per utterance when uniform
    int64_t total_frames;
    int32_t n_utt;
    // ragged batches only, both optional: utterance b is samples [seg[2b], seg[2b+1]) OF the range sample_off names
    // (endpoint-trimmed clips read in place), and per-utterance (sum (x - x0), sum (x - x0)^2) accumulators, x0 = the
    // segment's first sample (fp64 atomics; the unit-variance statistics of model.py:62-63)
    const int64_t* seg;
    double* stats;
};

// Largest b with off[b] <= g  (off is non-decreasing, off[0] == 0, off[n] == total > g).
__device__ __forceinline__ int32_t dsp_find_utt(const int64_t* __restrict__ off, int32_t n, int64_t g) {
    int32_t lo = 0, hi = n;  // invariant: off[lo] <= g < off[hi]
    while (hi - lo > 1) {
        int32_t mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// Resolve global frame g -> (utterance, frame-in-utterance, first sample of the utterance, its length).
__device__ __forceinline__ void dsp_locate(const BatchGeom& bg, int64_t g, int32_t& utt, int64_t& t,
                                           int64_t& s0, int64_t& nsamp) {
    if (bg.uniform_samples > 0) {
        utt = (int32_t)(g / bg.uniform_frames);
        t = g - (int64_t)utt * bg.uniform_frames;
        s0 = (int64_t)utt * bg.uniform_samples;
        nsamp = bg.uniform_samples;
    } else {
        utt = dsp_find_utt(bg.frame_off, bg.n_utt, g);
        t = g - bg.frame_off[utt];
        s0 = bg.sample_off[utt];
        nsamp = bg.sample_off[utt + 1] - s0;
    }
}

template <int DTYPE>
__device__ __forceinline__ float dsp_load_sample(const void* __restrict__ wave, int64_t i) {
    if constexpr (DTYPE == DSP_WAVE_I16) {
        return (float)reinterpret_cast<const int16_t*>(wave)[i];
    } else {
        return reinterpret_cast<const float*>(wave)[i];
    }
}

__device__ __forceinline__ float dsp_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
