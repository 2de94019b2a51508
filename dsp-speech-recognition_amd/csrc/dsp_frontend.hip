// C ABI of libdsp_frontend.so (see include/dsp_frontend.h).  gfx950 / ROCm only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "dsp_common.h"
#include "kernels_generic.h"
#include "kernels_fast512.h"
#include "kernels_fast1536.h"
#include "kernels_mfma512.h"
#include "kernels_mfma512t.h"
#include "kernels_vad.h"
#include "kernels_pitch.h"

thread_local int g_host_dry_run = 0;   // dsp_debug_host_dry_run: plan tables in host memory (sanitizer build, no GPU)

namespace {

thread_local std::string g_err;
thread_local int g_force_generic = 0;   // test switch, per calling thread: other threads' calls are unaffected

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(DSP_EHIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

template <typename T>
int upload(T** d, const T* h, size_t n) {
    *d = nullptr;
    if (n == 0) return DSP_OK;
    HIP_TRY(dsp_table_alloc_copy(reinterpret_cast<void**>(d), h, n * sizeof(T)));
    return DSP_OK;
}

int grid_for(int64_t work_items, int per_block) {
    int64_t blocks = (work_items + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    const int64_t cap = (int64_t)dsp_cu_count() * 8;  // CUs x 8 resident blocks; the kernels grid-stride beyond
    return (int)(blocks < cap ? blocks : cap);
}

bool factor_half_fft(int n2, std::vector<int>& radix) {
    radix.clear();
    if (n2 % 3 == 0) { radix.push_back(3); n2 /= 3; }
    while (n2 % 4 == 0) { radix.push_back(4); n2 /= 4; }
    if (n2 % 2 == 0) { radix.push_back(2); n2 /= 2; }
    return n2 == 1 && radix.size() <= DSP_MAX_RADIX_PASSES;
}

int check_geom(const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
               const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total,
               int64_t uniform_samples) {
    if (!d_wave) return fail(DSP_EINVAL, "d_wave is NULL");
    if (wave_dtype != DSP_WAVE_F32 && wave_dtype != DSP_WAVE_I16)
        return fail(DSP_EINVAL, "unsupported wave_dtype %d", wave_dtype);
    if (n_utt <= 0 || n_frames_total <= 0) return fail(DSP_EINVAL, "empty batch (n_utt=%d, frames=%lld)", n_utt, (long long)n_frames_total);
    if (uniform_samples <= 0 && (!d_sample_offsets || !d_frame_offsets))
        return fail(DSP_EINVAL, "ragged batch needs d_sample_offsets and d_frame_offsets");
    return DSP_OK;
}

BatchGeom make_geom(const int64_t* d_sample_offsets, const int64_t* d_frame_offsets, int32_t n_utt,
                    int64_t n_frames_total, int64_t uniform_samples, int32_t L, int32_t S) {
    BatchGeom bg;
    bg.sample_off = d_sample_offsets;
    bg.frame_off = d_frame_offsets;
    bg.uniform_samples = uniform_samples > 0 ? uniform_samples : 0;
    bg.uniform_frames = 0;
    if (uniform_samples > 0) {
        int64_t T;
        dsp_frame_count(uniform_samples, L, S, &T);
        bg.uniform_frames = T;
    }
    bg.total_frames = n_frames_total;
    bg.n_utt = n_utt;
    bg.seg = nullptr;
    bg.stats = nullptr;
    return bg;
}

GenericParams generic_params(const dsp_plan* p) {
    GenericParams P;
    memset(&P, 0, sizeof(P));
    P.L = p->L; P.S = p->S; P.nfft = p->nfft; P.K = p->K; P.M = p->M; P.C = p->C;
    P.lfft = p->lfft; P.append_energy = p->append_energy; P.preemph = p->preemph;
    P.window = p->d_window; P.tw = p->d_twiddle;
    P.mel_start = p->d_mel_start; P.mel_count = p->d_mel_count; P.mel_off = p->d_mel_off;
    P.mel_w = p->d_mel_w; P.dct = p->d_dct;
    std::vector<int> radix;
    factor_half_fft(p->nfft / 2, radix);
    P.n_pass = (int)radix.size();
    for (int i = 0; i < P.n_pass; ++i) P.radix[i] = radix[i];
    return P;
}

// sample_off'[b] = (dense ? b * n : sample_off[b]) + shift, frame_off'[b] = dense ? b * t : frame_off[b]
__global__ void offsets_view_kernel(const int64_t* __restrict__ src_sample, const int64_t* __restrict__ src_frame,
                                    int64_t* __restrict__ sample_off, int64_t* __restrict__ frame_off,
                                    int32_t n_utt, int64_t n, int64_t t, int64_t shift) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b <= n_utt; b += gridDim.x * blockDim.x) {
        sample_off[b] = (src_sample ? src_sample[b] : (int64_t)b * n) + shift;
        frame_off[b] = src_frame ? src_frame[b] : (int64_t)b * t;
    }
}

// The dense instantiations of the fused kernels need N % 4 == 0 (their 16-byte vectors must not
// straddle utterances) and every fused kernel needs a 16-byte aligned buffer (8 for int16).  A batch
// that violates either is still a perfectly good ragged batch of a buffer that starts a few samples
// earlier: write offset tables (arithmetic for dense input, shifted copies otherwise) into a pooled
// workspace and describe it that way.  `d_wave` is moved down to the aligned address.
// Returns nullptr and changes nothing if no view is needed or the workspace cannot be had.
DspWorkspace* fused_kernel_view(BatchGeom& bg, const void*& d_wave, int wave_dtype, hipStream_t st) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(d_wave);
    const size_t elem = wave_dtype == DSP_WAVE_I16 ? 2 : 4;
    const uintptr_t mis = addr % (4 * elem);               // bytes past the previous aligned vector
    const bool odd_dense = bg.uniform_samples > 0 && (bg.uniform_samples % 4) != 0;
    if ((mis == 0 && !odd_dense) || (mis % elem) != 0) return nullptr;
    DspWorkspace* w = dsp_workspace_pool().acquire(2 * ((size_t)bg.n_utt + 1) * sizeof(int64_t), st);
    if (!w) return nullptr;
    int64_t* so = static_cast<int64_t*>(w->ptr);
    int64_t* fo = so + bg.n_utt + 1;
    const int blocks = (bg.n_utt + 256) / 256 < 1024 ? (bg.n_utt + 256) / 256 : 1024;
    const bool dense = bg.uniform_samples > 0;
    offsets_view_kernel<<<blocks, 256, 0, st>>>(dense ? nullptr : bg.sample_off, dense ? nullptr : bg.frame_off, so, fo,
                                                bg.n_utt, bg.uniform_samples, bg.uniform_frames, (int64_t)(mis / elem));
    bg.sample_off = so;
    bg.frame_off = fo;
    bg.uniform_samples = 0;
    bg.uniform_frames = 0;
    d_wave = reinterpret_cast<const void*>(addr - mis);
    return w;
}

}  // namespace

extern "C" {

int dsp_abi_version(void) { return DSP_ABI_VERSION; }

int dsp_debug_force_generic(int on) {
    g_force_generic = on ? 1 : 0;
    return DSP_OK;
}

int dsp_debug_host_dry_run(int on) {
    g_host_dry_run = on ? 1 : 0;
    return DSP_OK;
}

int dsp_debug_use_mfma512(int on) {
    g_use_mfma512 = on < 0 ? -1 : (on > 2 ? 1 : on);
    return DSP_OK;
}

int dsp_plan_has_mfma512(const dsp_plan* plan) { return plan ? (plan->d_mfma ? 1 : 0) | (plan->d_mfmat ? 2 : 0) : 0; }

int dsp_debug_pool_stats(long long* n_buffers, long long* bytes) {
    dsp_workspace_pool().stats(n_buffers, bytes);
    return DSP_OK;
}

#ifdef F512_STAMPS
// diagnostic builds only: per-phase shader-clock sums of mfcc512_kernel (not declared in the public header)
int dsp_debug_read_stamps(unsigned long long* out, int n) {
    static std::vector<unsigned int> h(F512_NSTAMP * 8192);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(f512_stamp_sum), h.size() * 4));
    for (int i = 0; i < n && i < F512_NSTAMP; ++i) {
        out[i] = 0;
        for (int w = 0; w < 8192; ++w) out[i] += h[(size_t)w * F512_NSTAMP + i];
    }
    std::fill(h.begin(), h.end(), 0u);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(f512_stamp_sum), h.data(), h.size() * 4));
    return DSP_OK;
}
#endif

#ifdef M512T_STAMPS
// diagnostic builds only: per-phase shader-clock sums of mfcc512t_kernel (not declared in the public header)
int dsp_debug_read_stamps_m512t(unsigned long long* out, int n) {
    static std::vector<unsigned int> h(M512T_NSTAMP * 4096);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(m512t_stamp_sum), h.size() * 4));
    for (int i = 0; i < n && i < M512T_NSTAMP; ++i) {
        out[i] = 0;
        for (int w = 0; w < 4096; ++w) out[i] += h[(size_t)w * M512T_NSTAMP + i];
    }
    std::fill(h.begin(), h.end(), 0u);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(m512t_stamp_sum), h.data(), h.size() * 4));
    return DSP_OK;
}
#endif

#ifdef M512_STAMPS
// diagnostic builds only: per-phase shader-clock sums of mfcc512m_kernel (not declared in the public header)
int dsp_debug_read_stamps_m512(unsigned long long* out, int n) {
    static std::vector<unsigned int> h(M512_NSTAMP * 2048);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(m512_stamp_sum), h.size() * 4));
    for (int i = 0; i < n && i < M512_NSTAMP; ++i) {
        out[i] = 0;
        for (int w = 0; w < 2048; ++w) out[i] += h[(size_t)w * M512_NSTAMP + i];
    }
    std::fill(h.begin(), h.end(), 0u);
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(m512_stamp_sum), h.data(), h.size() * 4));
    return DSP_OK;
}
#endif

int dsp_plan_has_fast_path(const dsp_plan* plan) { return plan && (plan->d_fast || plan->d_fast1536) ? 1 : 0; }

const char* dsp_last_error(void) { return g_err.c_str(); }

int dsp_device_count(int* n) {
    if (!n) return fail(DSP_EINVAL, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(DSP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = c;
    return DSP_OK;
}

int dsp_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return DSP_OK;
}

int dsp_get_device(int* device) {
    if (!device) return fail(DSP_EINVAL, "dsp_get_device: NULL argument");
    HIP_TRY(hipGetDevice(device));
    return DSP_OK;
}

int dsp_malloc(void** d_ptr, size_t bytes) {
    if (!d_ptr) return fail(DSP_EINVAL, "d_ptr is NULL");
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
    return DSP_OK;
}

int dsp_free(void* d_ptr) {
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return DSP_OK;
}

int dsp_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return DSP_OK;
}

int dsp_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return DSP_OK;
}

int dsp_memset(void* d_dst, int value, size_t bytes, void* stream) {
    HIP_TRY(hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
    return DSP_OK;
}

int dsp_stream_synchronize(void* stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return DSP_OK;
}

int dsp_frame_count(int64_t n_samples, int32_t frame_len, int32_t frame_step, int64_t* n_frames) {
    if (!n_frames || frame_len <= 0 || frame_step <= 0 || n_samples < 0)
        return fail(DSP_EINVAL, "dsp_frame_count: bad arguments");
    if (n_samples <= frame_len) *n_frames = 1;
    else *n_frames = 1 + (n_samples - frame_len + frame_step - 1) / frame_step;
    return DSP_OK;
}

int dsp_frame_offsets(const int64_t* h_sample_offsets, int32_t n_utt, int32_t frame_len,
                      int32_t frame_step, int64_t* h_frame_offsets) {
    if (!h_sample_offsets || !h_frame_offsets || n_utt < 0)
        return fail(DSP_EINVAL, "dsp_frame_offsets: bad arguments");
    h_frame_offsets[0] = 0;
    for (int32_t b = 0; b < n_utt; ++b) {
        int64_t T;
        const int64_t n = h_sample_offsets[b + 1] - h_sample_offsets[b];
        if (n < 0) return fail(DSP_EINVAL, "sample_offsets not monotone at %d", b);
        int rc = dsp_frame_count(n, frame_len, frame_step, &T);
        if (rc != DSP_OK) return rc;
        h_frame_offsets[b + 1] = h_frame_offsets[b] + T;
    }
    return DSP_OK;
}

int dsp_plan_create(const dsp_plan_desc* d, dsp_plan** out) {
    if (!d || !out) return fail(DSP_EINVAL, "NULL desc/out");
    *out = nullptr;
    if (d->frame_len <= 0 || d->frame_step <= 0) return fail(DSP_EINVAL, "frame_len/frame_step must be > 0");
    std::vector<int> radix;
    if (d->nfft < 16 || d->nfft > 4096 || (d->nfft & 1) || !factor_half_fft(d->nfft / 2, radix))
        return fail(DSP_EINVAL, "unsupported nfft %d (need 2^k or 3*2^k in [16, 4096])", d->nfft);
    if (!d->h_window) return fail(DSP_EINVAL, "h_window is NULL");
    if (d->nfilt < 0 || d->numcep < 0 || d->numcep > d->nfilt)
        return fail(DSP_EINVAL, "need 0 <= numcep <= nfilt (got %d, %d)", d->numcep, d->nfilt);
    if (d->nfilt > d->nfft) return fail(DSP_EINVAL, "nfilt %d > nfft %d", d->nfilt, d->nfft);
    const int K = d->nfft / 2 + 1;
    std::vector<int32_t> off(d->nfilt > 0 ? d->nfilt : 1, 0);
    int64_t nnz = 0;
    if (d->nfilt > 0) {
        if (!d->h_mel_start || !d->h_mel_count || !d->h_mel_weights)
            return fail(DSP_EINVAL, "mel tables missing");
        if (d->numcep > 0 && !d->h_dct) return fail(DSP_EINVAL, "h_dct is NULL");
        for (int j = 0; j < d->nfilt; ++j) {
            if (d->h_mel_count[j] < 0 || d->h_mel_start[j] < 0 || d->h_mel_start[j] + d->h_mel_count[j] > K)
                return fail(DSP_EINVAL, "mel filter %d spans bins outside [0,%d)", j, K);
            off[j] = (int32_t)nnz;
            nnz += d->h_mel_count[j];
        }
    }
    int dev = 0;
    if (!g_host_dry_run) HIP_TRY(hipGetDevice(&dev));
    dsp_plan* p = new dsp_plan();
    memset(p, 0, sizeof(*p));
    p->dry_run = g_host_dry_run ? 1 : 0;
    if (p->dry_run) dev = -1;      // no device owns these tables: every launch path that checks the plan's device refuses
    p->L = d->frame_len; p->S = d->frame_step; p->nfft = d->nfft; p->K = K;
    p->M = d->nfilt; p->C = d->numcep; p->append_energy = d->append_energy ? 1 : 0;
    p->lfft = d->frame_len < d->nfft ? d->frame_len : d->nfft;
    p->preemph = d->preemph; p->mel_nnz = (int32_t)nnz; p->device = dev;
    std::vector<float2> tw(d->nfft);
    for (int k = 0; k < d->nfft; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)d->nfft;
        tw[k] = make_float2((float)cos(a), (float)sin(a));
    }
    int rc = DSP_OK;
    if (rc == DSP_OK) rc = upload(&p->d_window, d->h_window, (size_t)d->frame_len);
    if (rc == DSP_OK) rc = upload(&p->d_twiddle, tw.data(), tw.size());
    if (rc == DSP_OK && d->nfilt > 0) {
        rc = upload(&p->d_mel_start, d->h_mel_start, (size_t)d->nfilt);
        if (rc == DSP_OK) rc = upload(&p->d_mel_count, d->h_mel_count, (size_t)d->nfilt);
        if (rc == DSP_OK) rc = upload(&p->d_mel_off, off.data(), (size_t)d->nfilt);
        if (rc == DSP_OK) rc = upload(&p->d_mel_w, d->h_mel_weights, (size_t)nnz);
        if (rc == DSP_OK && d->numcep > 0) rc = upload(&p->d_dct, d->h_dct, (size_t)d->numcep * d->nfilt);
    }
    if (rc == DSP_OK) rc = fast512_plan_init(p, d, off.data());
    if (rc == DSP_OK && d->nfilt > 0 && d->numcep > 0) rc = fast1536_plan_init(p, d, off.data());
    if (rc == DSP_OK && d->nfilt > 0 && d->numcep > 0 && p->d_fast) rc = mfma512_plan_init(p, d);
    if (rc == DSP_OK && d->nfilt > 0 && d->numcep > 0 && p->d_fast) rc = mfma512t_plan_init(p, d);
    if (rc != DSP_OK) { dsp_plan_destroy(p); return rc; }
    *out = p;
    return DSP_OK;
}

int dsp_plan_destroy(dsp_plan* p) {
    if (!p) return DSP_OK;
    void* bufs[] = {p->d_window, p->d_twiddle, p->d_mel_start, p->d_mel_count, p->d_mel_off, p->d_mel_w, p->d_dct};
    for (void* b : bufs) dsp_table_free(b, p->dry_run);
    fast512_plan_free(p);
    fast1536_plan_free(p);
    mfma512_plan_free(p);
    mfma512t_plan_free(p);
    delete p;
    return DSP_OK;
}

int dsp_preemphasis_batch(const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                          int32_t n_utt, int64_t n_samples_total, float coeff, float* d_out, void* stream) {
    if (!d_wave || !d_out || !d_sample_offsets || n_utt <= 0 || n_samples_total <= 0)
        return fail(DSP_EINVAL, "dsp_preemphasis_batch: bad arguments");
    const int grid = grid_for(n_samples_total, 256);
    hipStream_t st = (hipStream_t)stream;
    if (wave_dtype == DSP_WAVE_I16)
        preemphasis_kernel<DSP_WAVE_I16><<<grid, 256, 0, st>>>(d_wave, d_sample_offsets, n_utt, n_samples_total, coeff, d_out);
    else if (wave_dtype == DSP_WAVE_F32)
        preemphasis_kernel<DSP_WAVE_F32><<<grid, 256, 0, st>>>(d_wave, d_sample_offsets, n_utt, n_samples_total, coeff, d_out);
    else
        return fail(DSP_EINVAL, "unsupported wave_dtype %d", wave_dtype);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

static int launch_generic(const dsp_plan* plan, const void* d_wave, int wave_dtype, const BatchGeom& bg,
                          int out_kind, float* d_out, int64_t ld_out, float* d_out2, hipStream_t st) {
    GenericParams P = generic_params(plan);
    const size_t lds = (size_t)DSP_GEN_WAVES * 2 * (plan->nfft / 2) * sizeof(float2);
    const int grid = grid_for(bg.total_frames, DSP_GEN_WAVES);
    if (wave_dtype == DSP_WAVE_I16) {
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void*)features_generic_kernel<DSP_WAVE_I16>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        features_generic_kernel<DSP_WAVE_I16><<<grid, 64 * DSP_GEN_WAVES, lds, st>>>(P, bg, d_wave, out_kind, d_out, ld_out, d_out2);
    } else {
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void*)features_generic_kernel<DSP_WAVE_F32>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        features_generic_kernel<DSP_WAVE_F32><<<grid, 64 * DSP_GEN_WAVES, lds, st>>>(P, bg, d_wave, out_kind, d_out, ld_out, d_out2);
    }
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

static int features_batch_impl(const dsp_plan* plan, const void* d_wave, int wave_dtype,
                               const int64_t* d_sample_offsets, const int64_t* d_frame_offsets, int32_t n_utt,
                               int64_t n_frames_total, int64_t uniform_samples, int out_kind, float* d_out,
                               int64_t ld_out, float* d_out2, void* stream, const DspRaggedTables* pre) {
    if (!plan || !d_out) return fail(DSP_EINVAL, "plan/d_out is NULL");
    {
        int dev = -1;
        HIP_TRY(hipGetDevice(&dev));
        if (dev != plan->device)   // the plan's tables live on the device it was created on
            return fail(DSP_EINVAL, "plan belongs to device %d, current device is %d", plan->device, dev);
    }
    int rc = check_geom(d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, n_utt, n_frames_total, uniform_samples);
    if (rc != DSP_OK) return rc;
    int width;
    switch (out_kind) {
        case DSP_OUT_FRAMES: width = plan->L; break;
        case DSP_OUT_MAGSPEC:
        case DSP_OUT_POWSPEC: width = plan->K; break;
        case DSP_OUT_FBANK:
            width = plan->M;
            if (plan->M <= 0) return fail(DSP_EINVAL, "plan has no mel filterbank");
            if (!d_out2) return fail(DSP_EINVAL, "DSP_OUT_FBANK needs d_out2 (energy)");
            break;
        case DSP_OUT_MFCC:
            width = plan->C;
            if (plan->M <= 0 || plan->C <= 0) return fail(DSP_EINVAL, "plan has no mel/DCT tables");
            break;
        default: return fail(DSP_EINVAL, "unknown out_kind %d", out_kind);
    }
    if (ld_out == 0) ld_out = width;
    if (ld_out < width) return fail(DSP_EINVAL, "ld_out %lld < row width %d", (long long)ld_out, width);
    BatchGeom bg = make_geom(d_sample_offsets, d_frame_offsets, n_utt, n_frames_total, uniform_samples, plan->L, plan->S);
    if (uniform_samples > 0 && bg.uniform_frames * n_utt != n_frames_total)
        return fail(DSP_EINVAL, "n_frames_total %lld != n_utt*T (%d*%lld)", (long long)n_frames_total, n_utt, (long long)bg.uniform_frames);
    hipStream_t st = (hipStream_t)stream;
    if (out_kind == DSP_OUT_MFCC && !g_force_generic && !pre && mfma512t_applicable(plan, bg, wave_dtype, 0)) {
        const int mrc = mfma512t_launch(plan, d_wave, wave_dtype, bg, 0, d_out, ld_out, st);   // dense batches: matrix-pipe kernel, frame per product
        if (mrc == DSP_OK) return DSP_OK;
        if (mrc < 0) return fail(mrc, "matrix-pipe MFCC kernel launch failed");
    }
    if (out_kind == DSP_OUT_MFCC && !g_force_generic && !pre && mfma512_applicable(plan, bg, wave_dtype, 0)) {
        const int mrc = mfma512_launch(plan, d_wave, wave_dtype, bg, 0, d_out, ld_out, st);   // dense batches: matrix-pipe kernel
        if (mrc == DSP_OK) return DSP_OK;
        if (mrc < 0) return fail(mrc, "matrix-pipe MFCC kernel launch failed");
    }
    if (out_kind == DSP_OUT_MFCC && !g_force_generic && (plan->d_fast || plan->d_fast1536)) {
        BatchGeom fg = bg;
        const void* fw = d_wave;
        DspWorkspace* view = fused_kernel_view(fg, fw, wave_dtype, st);
        int frc = 1;   // 1 = no fused kernel took it
        if (fast512_applicable(plan, fg, fw, wave_dtype))
            frc = fast512_launch(plan, fw, wave_dtype, fg, d_out, ld_out, st, pre);
        else if (fast1536_applicable(plan, fg, fw, wave_dtype))
            frc = fast1536_launch(plan, fw, wave_dtype, fg, d_out, ld_out, st, pre);
        if (view && dsp_workspace_pool().release(view, st) != 0 && frc == DSP_OK) frc = DSP_EHIP;
        if (frc == DSP_OK) return DSP_OK;
        if (frc < 0) return fail(frc, "fused kernel launch failed");
    }
    return launch_generic(plan, d_wave, wave_dtype, bg, out_kind, d_out, ld_out, d_out2, st);
}

int dsp_features_batch(const dsp_plan* plan, const void* d_wave, int wave_dtype,
                       const int64_t* d_sample_offsets, const int64_t* d_frame_offsets, int32_t n_utt,
                       int64_t n_frames_total, int64_t uniform_samples, int out_kind, float* d_out,
                       int64_t ld_out, float* d_out2, void* stream) {
    return features_batch_impl(plan, d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, n_utt, n_frames_total,
                               uniform_samples, out_kind, d_out, ld_out, d_out2, stream, nullptr);
}

int dsp_delta_batch(const float* d_in, int64_t ld_in, const int64_t* d_frame_offsets, int32_t n_utt,
                    int64_t n_frames_total, int64_t uniform_frames, int32_t D, int32_t N, float* d_out,
                    int64_t ld_out, float* d_out_dd, int64_t ld_out_dd, void* stream) {
    if (!d_in || !d_out) return fail(DSP_EINVAL, "dsp_delta_batch: NULL buffer");
    if (N < 1) return fail(DSP_EINVAL, "N must be an integer >= 1");  // base.py:71-72
    if (D <= 0 || n_utt <= 0 || n_frames_total <= 0) return fail(DSP_EINVAL, "dsp_delta_batch: empty input");
    if (uniform_frames <= 0 && !d_frame_offsets) return fail(DSP_EINVAL, "ragged batch needs d_frame_offsets");
    if (ld_in == 0) ld_in = D;
    if (ld_out == 0) ld_out = D;
    if (ld_out_dd == 0) ld_out_dd = D;
    BatchGeom bg;
    memset(&bg, 0, sizeof(bg));
    bg.frame_off = d_frame_offsets;
    bg.uniform_frames = uniform_frames > 0 ? uniform_frames : 0;
    bg.total_frames = n_frames_total;
    bg.n_utt = n_utt;
    int den = 0;
    for (int i = 1; i <= N; ++i) den += i * i;
    const float inv_den = (float)(1.0 / (2.0 * den));
    // tiled LDS kernel for uniform batches (tile table is arithmetic); per-element kernel otherwise
    const size_t lds = ((size_t)(DT_TILE + 4 * N) + (size_t)(DT_TILE + 2 * N)) * D * sizeof(float);
    if (uniform_frames > 0 && lds <= 64 * 1024) {
        const int64_t tiles = (uniform_frames + DT_TILE - 1) / DT_TILE;
        const int64_t blocks = tiles * n_utt;
        if (blocks > 0x7fffffff) return fail(DSP_EINVAL, "too many delta tiles");
        if (ld_in > 0x7fffff || ld_out > 0x7fffff || ld_out_dd > 0x7fffff) return fail(DSP_EINVAL, "row stride too large");
        if (D == 13)
            delta_tiled_kernel<13><<<(int)blocks, 256, lds, (hipStream_t)stream>>>(
                d_in, ld_in, bg, D, N, inv_den, d_out, ld_out, d_out_dd, ld_out_dd, (int32_t)tiles, nullptr);
        else
            delta_tiled_kernel<0><<<(int)blocks, 256, lds, (hipStream_t)stream>>>(
                d_in, ld_in, bg, D, N, inv_den, d_out, ld_out, d_out_dd, ld_out_dd, (int32_t)tiles, nullptr);
    } else if (uniform_frames <= 0 && lds <= 64 * 1024 && ld_in <= 0x7fffff && ld_out <= 0x7fffff && ld_out_dd <= 0x7fffff) {
        // ragged: per-utterance tile prefix in a pooled, event-guarded workspace, grid sized by a bound
        static_assert(DT_TILE == (1 << DT_SHIFT), "tile tables are built with shifts");
        const int64_t bound = n_frames_total / DT_TILE + n_utt;
        if (bound > 0x7fffffff) return fail(DSP_EINVAL, "too many delta tiles");
        hipStream_t st = (hipStream_t)stream;
        DspWorkspace* w = dsp_workspace_pool().acquire(((size_t)n_utt + 1) * sizeof(int64_t), st);
        if (!w) return fail(DSP_EHIP, "workspace allocation failed");
        int64_t* tile_off = static_cast<int64_t*>(w->ptr);
        prefix_ceil_kernel<<<1, 1024, 0, st>>>(d_frame_offsets, n_utt, DT_SHIFT, tile_off);
        if (D == 13)
            delta_tiled_kernel<13><<<(int)bound, 256, lds, st>>>(d_in, ld_in, bg, D, N, inv_den, d_out, ld_out,
                                                                 d_out_dd, ld_out_dd, 0, tile_off);
        else
            delta_tiled_kernel<0><<<(int)bound, 256, lds, st>>>(d_in, ld_in, bg, D, N, inv_den, d_out, ld_out,
                                                                d_out_dd, ld_out_dd, 0, tile_off);
        if (dsp_workspace_pool().release(w, st) != 0) return fail(DSP_EHIP, "workspace release failed");
    } else {
        delta_kernel<<<grid_for(n_frames_total * D, 256), 256, 0, (hipStream_t)stream>>>(
            d_in, ld_in, bg, D, N, inv_den, d_out, ld_out, d_out_dd, ld_out_dd);
    }
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_mfcc_delta_batch(const dsp_plan* plan, const void* d_wave, int wave_dtype,
                         const int64_t* d_sample_offsets, const int64_t* d_frame_offsets, int32_t n_utt,
                         int64_t n_frames_total, int64_t uniform_samples, int32_t delta_n, float* d_out,
                         void* stream) {
    if (!plan) return fail(DSP_EINVAL, "plan is NULL");
    if (delta_n < 1) return fail(DSP_EINVAL, "N must be an integer >= 1");  // base.py:71-72
#ifdef DSP_WS_MALLOC_ASYNC
    dsp_ws_diag_stream() = (hipStream_t)stream;
#endif
    const int C = plan->C;
    if (C <= 0) return fail(DSP_EINVAL, "plan has no mel/DCT tables");
    if (!d_out) return fail(DSP_EINVAL, "plan/d_out is NULL");
    {   // validate before the first launch or workspace acquire (the table kernels read the offset arrays)
        const int grc = check_geom(d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, n_utt, n_frames_total, uniform_samples);
        if (grc != DSP_OK) return grc;
    }
    int64_t uniform_frames = 0;
    if (uniform_samples > 0) dsp_frame_count(uniform_samples, plan->L, plan->S, &uniform_frames);
    hipStream_t st = (hipStream_t)stream;
    // Dense batches of the NFFT = 512 plans: ONE kernel writes the finished rows (kernels_fast512.h, "Fused delta").
    if (uniform_samples > 0 && !g_force_generic && plan->d_fast) {
        int dev = -1;
        HIP_TRY(hipGetDevice(&dev));
        if (dev != plan->device) return fail(DSP_EINVAL, "plan belongs to device %d, current device is %d", plan->device, dev);
        if (uniform_frames * n_utt != n_frames_total)
            return fail(DSP_EINVAL, "n_frames_total %lld != n_utt*T (%d*%lld)", (long long)n_frames_total, n_utt, (long long)uniform_frames);
        const BatchGeom fbg = make_geom(nullptr, nullptr, n_utt, n_frames_total, uniform_samples, plan->L, plan->S);
        if (mfma512t_applicable(plan, fbg, wave_dtype, delta_n)) {
            const int mrc = mfma512t_launch(plan, d_wave, wave_dtype, fbg, delta_n, d_out, 3 * (int64_t)C, st);
            if (mrc == DSP_OK) return DSP_OK;
            if (mrc < 0) return fail(mrc, "matrix-pipe MFCC + delta kernel launch failed");
        }
        if (mfma512_applicable(plan, fbg, wave_dtype, delta_n)) {
            const int mrc = mfma512_launch(plan, d_wave, wave_dtype, fbg, delta_n, d_out, 3 * (int64_t)C, st);
            if (mrc == DSP_OK) return DSP_OK;
            if (mrc < 0) return fail(mrc, "matrix-pipe MFCC + delta kernel launch failed");
        }
        const int frc = fast512_launch_fused(plan, d_wave, wave_dtype, fbg, delta_n, d_out, st);
        if (frc == DSP_OK) return DSP_OK;
        if (frc < 0) return fail(frc, "fused MFCC + delta kernel launch failed");
    }
    // Two passes, every byte written once as part of a full line: the MFCC kernel writes DENSE cepstra
    // [sum T, C] into a pooled scratch buffer, delta_rows_kernel turns them into whole 3C-float rows.
    // (Writing the 52-byte cepstra straight into the 156-byte rows cost 1.5x write amplification and a
    // strided re-read.)  Scratch above 256 MiB falls back to the in-place form.
    const size_t lds = ((size_t)(DT_TILE + 4 * delta_n) + (size_t)(DT_TILE + 2 * delta_n)) * C * sizeof(float);
    const size_t scratch_bytes = (size_t)n_frames_total * C * sizeof(float);
    if (n_frames_total > 0 && n_utt > 0 && lds <= 64 * 1024 && scratch_bytes <= ((size_t)256 << 20)) {
        const bool ragged = uniform_frames <= 0;
        int64_t tiles = 0, blocks = 0;
        if (!ragged) {
            tiles = (uniform_frames + DT_TILE - 1) / DT_TILE;
            blocks = tiles * n_utt;
        } else {
            if (!d_frame_offsets) return fail(DSP_EINVAL, "ragged batch needs d_frame_offsets");
            blocks = n_frames_total / DT_TILE + n_utt;
        }
        if (blocks <= 0x7fffffff) {
            // ragged: the delta tile table and the fused MFCC kernel's group tables come from ONE small launch
            const int gshift = plan->d_fast ? 3 : (plan->d_fast1536 ? 2 : 0);
            const int64_t gbound = gshift ? (n_frames_total >> gshift) + n_utt : 0;
            auto pad256 = [](size_t b) { return (b + 255) / 256 * 256; };
            const size_t tile_bytes = ragged ? pad256(((size_t)n_utt + 1) * sizeof(int64_t)) : 0;
            const size_t goff_bytes = ragged && gshift ? pad256(((size_t)n_utt + 1) * sizeof(int32_t)) : 0;
            const size_t gutt_bytes = ragged && gshift ? pad256((size_t)gbound * sizeof(int32_t)) : 0;
            DspWorkspace* w = dsp_workspace_pool().acquire(tile_bytes + goff_bytes + gutt_bytes + scratch_bytes, st);
            if (w) {
            char* wp = static_cast<char*>(w->ptr);
            int64_t* tile_off = ragged ? reinterpret_cast<int64_t*>(wp) : nullptr;
            DspRaggedTables pre;
            pre.shift = gshift;
            pre.group_off = reinterpret_cast<int32_t*>(wp + tile_bytes);
            pre.group_utt = reinterpret_cast<int32_t*>(wp + tile_bytes + goff_bytes);
            float* cep = reinterpret_cast<float*>(wp + tile_bytes + goff_bytes + gutt_bytes);
            const bool have_pre = ragged && gshift != 0 && gbound <= 0x3fffffff;
            if (have_pre) f512_build_group_tables(d_frame_offsets, n_utt, gshift, pre.group_off, pre.group_utt, st, tile_off);
            int rc = features_batch_impl(plan, d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, n_utt,
                                         n_frames_total, uniform_samples, DSP_OUT_MFCC, cep, (int64_t)C, nullptr, stream,
                                         have_pre ? &pre : nullptr);
            if (rc == DSP_OK) {
                BatchGeom bg;
                memset(&bg, 0, sizeof(bg));
                bg.frame_off = d_frame_offsets;
                bg.uniform_frames = uniform_frames > 0 ? uniform_frames : 0;
                bg.total_frames = n_frames_total;
                bg.n_utt = n_utt;
                int den = 0;
                for (int i = 1; i <= delta_n; ++i) den += i * i;
                const float inv_den = (float)(1.0 / (2.0 * den));
                if (ragged && !have_pre) prefix_ceil_kernel<<<1, 1024, 0, st>>>(d_frame_offsets, n_utt, DT_SHIFT, tile_off);
                if (C == 13)
                    delta_rows_kernel<13><<<(int)blocks, 256, lds, st>>>(cep, bg, C, delta_n, inv_den, d_out, (int32_t)tiles, tile_off);
                else
                    delta_rows_kernel<0><<<(int)blocks, 256, lds, st>>>(cep, bg, C, delta_n, inv_den, d_out, (int32_t)tiles, tile_off);
                if (hipGetLastError() != hipSuccess) rc = fail(DSP_EHIP, "delta_rows_kernel launch failed");
            }
            if (dsp_workspace_pool().release(w, st) != 0 && rc == DSP_OK) rc = fail(DSP_EHIP, "workspace release failed");
            return rc;
            }
            // no scratch to be had (device memory exhausted, or `stream` is being captured into a HIP graph): the
            // in-place form below needs none -- same values, 52-byte partial row writes instead of whole lines
        }
    }
    int rc = dsp_features_batch(plan, d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, n_utt,
                                n_frames_total, uniform_samples, DSP_OUT_MFCC, d_out, 3 * (int64_t)C, nullptr, stream);
    if (rc != DSP_OK) return rc;
    return dsp_delta_batch(d_out, 3 * (int64_t)C, d_frame_offsets, n_utt, n_frames_total, uniform_frames, C,
                           delta_n, d_out + C, 3 * (int64_t)C, d_out + 2 * C, 3 * (int64_t)C, stream);
}

int dsp_scale_columns(float* d_x, int64_t rows, int32_t cols, const float* d_scale, void* stream) {
    if (!d_x || !d_scale || rows <= 0 || cols <= 0) return fail(DSP_EINVAL, "dsp_scale_columns: bad arguments");
    scale_columns_kernel<<<grid_for(rows * cols, 256), 256, 0, (hipStream_t)stream>>>(d_x, rows, cols, d_scale);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

}  // extern "C"

// A batch shape's index tables, built once (include/dsp_frontend.h: dsp_layout).
struct dsp_layout {
    int32_t n_utt, frame_len, frame_step, shift;
    int64_t n_frames_total;
    int32_t* group_off;
    int32_t* group_utt;
    int32_t* group_off2;   // tables of the int16 VAD kernel's 8-frame groups, where it uses them (vad_scan_frames8)
    int32_t* group_utt2;
    int device;
};

static int vad_features_impl(const dsp_layout* layout, const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                             const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total,
                             int64_t uniform_samples, int32_t frame_len, int32_t frame_step, int32_t use_sq,
                             double* d_amp_sum, int32_t* d_zcr, void* stream);

extern "C" {

int dsp_layout_create(const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total, int32_t frame_len,
                      int32_t frame_step, void* stream, dsp_layout** out) {
    if (!out) return fail(DSP_EINVAL, "dsp_layout_create: out is NULL");
    *out = nullptr;
    if (!d_frame_offsets || n_utt <= 0 || n_frames_total <= 0 || frame_len <= 0 || frame_step <= 0)
        return fail(DSP_EINVAL, "dsp_layout_create: bad arguments");
    const int tile = vad_tile_frames(frame_len, frame_step);
    dsp_layout* l = new dsp_layout();
    memset(l, 0, sizeof(*l));
    l->n_utt = n_utt; l->frame_len = frame_len; l->frame_step = frame_step; l->n_frames_total = n_frames_total;
    if (hipGetDevice(&l->device) != hipSuccess) { delete l; return fail(DSP_EHIP, "dsp_layout_create: hipGetDevice failed"); }
    if (tile != 0) {
        l->shift = tile == 16 ? 4 : 2;
        const int64_t bound = n_frames_total / tile + n_utt;
        if (bound > 0x3fffffff) { delete l; return fail(DSP_EINVAL, "dsp_layout_create: batch too large"); }
        const bool second = vad_scan_frames8(frame_len, frame_step, DSP_WAVE_I16, 0);      // int16 callers take 8-frame groups
        const int64_t bound2 = second ? n_frames_total / 8 + n_utt : 0;
        const size_t n1 = (size_t)n_utt + 1 + (size_t)bound, n2 = second ? (size_t)n_utt + 1 + (size_t)bound2 : 0;
        if (hipMalloc(reinterpret_cast<void**>(&l->group_off), (n1 + n2) * sizeof(int32_t)) != hipSuccess) {
            delete l;
            return fail(DSP_EHIP, "dsp_layout_create: allocation failed");
        }
        l->group_utt = l->group_off + n_utt + 1;
        f512_build_group_tables(d_frame_offsets, n_utt, l->shift, l->group_off, l->group_utt, (hipStream_t)stream);
        if (second) {
            l->group_off2 = l->group_off + n1;
            l->group_utt2 = l->group_off2 + n_utt + 1;
            f512_build_group_tables(d_frame_offsets, n_utt, 3, l->group_off2, l->group_utt2, (hipStream_t)stream);
        }
        if (hipGetLastError() != hipSuccess) { (void)hipFree(l->group_off); delete l; return fail(DSP_EHIP, "dsp_layout_create: launch failed"); }
    }
    *out = l;
    return DSP_OK;
}

int dsp_layout_destroy(dsp_layout* layout) {
    if (!layout) return DSP_OK;
    if (layout->group_off) (void)hipFree(layout->group_off);
    delete layout;
    return DSP_OK;
}

int dsp_vad_features_layout_batch(const dsp_layout* layout, const void* d_wave, int wave_dtype,
                                  const int64_t* d_sample_offsets, const int64_t* d_frame_offsets, int32_t use_sq,
                                  double* d_amp_sum, int32_t* d_zcr, void* stream) {
    if (!layout) return fail(DSP_EINVAL, "dsp_vad_features_layout_batch: layout is NULL");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != layout->device) return fail(DSP_EINVAL, "layout belongs to device %d, current device is %d", layout->device, dev);
    return vad_features_impl(layout, d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, layout->n_utt,
                             layout->n_frames_total, 0, layout->frame_len, layout->frame_step, use_sq, d_amp_sum, d_zcr,
                             stream);
}

int dsp_vad_features_batch(const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                           const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total,
                           int64_t uniform_samples, int32_t frame_len, int32_t frame_step, int32_t use_sq,
                           double* d_amp_sum, int32_t* d_zcr, void* stream) {
    return vad_features_impl(nullptr, d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, n_utt, n_frames_total,
                             uniform_samples, frame_len, frame_step, use_sq, d_amp_sum, d_zcr, stream);
}

}  // extern "C"

static int vad_features_impl(const dsp_layout* layout, const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                             const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total,
                             int64_t uniform_samples, int32_t frame_len, int32_t frame_step, int32_t use_sq,
                             double* d_amp_sum, int32_t* d_zcr, void* stream) {
    if (!d_amp_sum || !d_zcr) return fail(DSP_EINVAL, "dsp_vad_features_batch: NULL output");
    if (frame_len <= 0 || frame_step <= 0) return fail(DSP_EINVAL, "frame_len/frame_step must be > 0");
    int rc = check_geom(d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, n_utt, n_frames_total, uniform_samples);
    if (rc != DSP_OK) return rc;
    BatchGeom bg = make_geom(d_sample_offsets, d_frame_offsets, n_utt, n_frames_total, uniform_samples, frame_len, frame_step);
    hipStream_t st = (hipStream_t)stream;
    const int tile = vad_tile_frames(frame_len, frame_step);
    if (!g_force_generic && tile != 0) {
        BatchGeom fg = bg;
        const void* fw = d_wave;
        DspWorkspace* view = fused_kernel_view(fg, fw, wave_dtype, st);
        const bool ok = vad_tile_applicable(fg, fw, wave_dtype, tile);
        DspRaggedTables pre;
        const bool have_pre = layout != nullptr && layout->group_off != nullptr && view == nullptr;
        if (have_pre) {
            pre.shift = layout->shift; pre.group_off = layout->group_off; pre.group_utt = layout->group_utt;
            if (layout->group_off2 != nullptr) { pre.shift2 = 3; pre.group_off2 = layout->group_off2; pre.group_utt2 = layout->group_utt2; }
        }
        if (ok) rc = vad_tile_launch(tile, frame_len, frame_step, use_sq, fg, fw, wave_dtype, d_amp_sum, d_zcr, st,
                                     have_pre ? &pre : nullptr);
        if (view && dsp_workspace_pool().release(view, st) != 0 && ok && rc == DSP_OK) rc = DSP_EHIP;
        if (ok) {
            if (rc != DSP_OK) return fail(rc, "vad tile kernel launch failed");
            return DSP_OK;
        }
    }
    const int grid = grid_for(n_frames_total, 4);
    if (wave_dtype == DSP_WAVE_I16)
        vad_features_kernel<DSP_WAVE_I16><<<grid, 256, 0, st>>>(d_wave, bg, frame_len, frame_step, use_sq, d_amp_sum, d_zcr);
    else
        vad_features_kernel<DSP_WAVE_F32><<<grid, 256, 0, st>>>(d_wave, bg, frame_len, frame_step, use_sq, d_amp_sum, d_zcr);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

extern "C" {

int dsp_trim_scale_batch(const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                         const int64_t* d_segments, const int64_t* d_dst_offsets, int32_t n_utt,
                         int32_t unit_variance, float* d_out, void* stream) {
    if (!d_wave || !d_sample_offsets || !d_segments || !d_dst_offsets || !d_out || n_utt <= 0)
        return fail(DSP_EINVAL, "dsp_trim_scale_batch: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (wave_dtype == DSP_WAVE_I16)
        trim_scale_kernel<DSP_WAVE_I16><<<n_utt, 256, 0, st>>>(d_wave, d_sample_offsets, d_segments, d_dst_offsets, unit_variance, d_out);
    else if (wave_dtype == DSP_WAVE_F32)
        trim_scale_kernel<DSP_WAVE_F32><<<n_utt, 256, 0, st>>>(d_wave, d_sample_offsets, d_segments, d_dst_offsets, unit_variance, d_out);
    else
        return fail(DSP_EINVAL, "unsupported wave_dtype %d", wave_dtype);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

namespace {
struct SegWork {
    size_t stats, tile, goff, gutt, tutt, cep, total;
};
// layout of the caller-owned work buffer of dsp_mfcc_delta_segments_batch (every part 256-byte aligned)
SegWork seg_work_layout(int32_t n_utt, int64_t n_frames_bound, int32_t C) {
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    SegWork w;
    w.stats = 0;
    w.tile = w.stats + pad((size_t)n_utt * 2 * sizeof(double));
    w.goff = w.tile + pad(((size_t)n_utt + 1) * sizeof(int64_t));
    w.gutt = w.goff + pad(((size_t)n_utt + 1) * sizeof(int32_t));
    w.tutt = w.gutt + pad(((size_t)(n_frames_bound >> 2) + (size_t)n_utt) * sizeof(int32_t));   // 4-frame groups (NFFT = 1536) at most
    w.cep = w.tutt + pad(((size_t)(n_frames_bound >> DT_SHIFT) + (size_t)n_utt) * sizeof(int32_t));
    w.total = w.cep + pad((size_t)n_frames_bound * (size_t)C * sizeof(float));
    return w;
}
}  // namespace

int dsp_segments_workspace_bytes(const dsp_plan* plan, int32_t n_utt, int64_t n_frames_bound, size_t* bytes) {
    if (!plan || !bytes || n_utt <= 0 || n_frames_bound <= 0) return fail(DSP_EINVAL, "dsp_segments_workspace_bytes: bad arguments");
    *bytes = seg_work_layout(n_utt, n_frames_bound, plan->C).total;
    return DSP_OK;
}

int dsp_mfcc_delta_segments_batch(const dsp_plan* plan, const void* d_wave, int wave_dtype,
                                  const int64_t* d_sample_offsets, const int64_t* d_segments,
                                  const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_bound,
                                  int32_t delta_n, int32_t flags, void* d_work, size_t work_bytes,
                                  float* d_out, void* stream) {
    const int unit_variance = flags & DSP_SEG_UNIT_VARIANCE;
    if (!plan || !d_out || !d_work || !d_segments) return fail(DSP_EINVAL, "dsp_mfcc_delta_segments_batch: NULL argument");
    if (delta_n < 0) return fail(DSP_EINVAL, "N must be an integer >= 1 (or 0: cepstra only)");  // base.py:71-72
    int rc = check_geom(d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, n_utt, n_frames_bound, 0);
    if (rc != DSP_OK) return rc;
    {
        int dev = -1;
        HIP_TRY(hipGetDevice(&dev));
        if (dev != plan->device) return fail(DSP_EINVAL, "plan belongs to device %d, current device is %d", plan->device, dev);
    }
    const int C = plan->C;
    const size_t lds = ((size_t)(DT_TILE + 4 * delta_n) + (size_t)(DT_TILE + 2 * delta_n)) * C * sizeof(float);
    BatchGeom bg = make_geom(d_sample_offsets, d_frame_offsets, n_utt, n_frames_bound, 0, plan->L, plan->S);
    // Served by the NFFT = 512 and NFFT = 1536 kernels on buffers they can read in place; everything else (and unit
    // variance without appendEnergy, where the scaling does not reduce to a shift of c0) reports 1: "use the
    // trimmed-copy path".
    const bool k512 = plan->d_fast && fast512_applicable(plan, bg, d_wave, wave_dtype);
    const bool k1536 = !k512 && plan->d_fast1536 && fast1536_applicable(plan, bg, d_wave, wave_dtype);
    if (g_force_generic || (!k512 && !k1536) || C <= 0 || lds > 64 * 1024 ||
        (unit_variance && !plan->append_energy) || (n_frames_bound >> 2) + n_utt > 0x3fffffff)
        return 1;
    const SegWork w = seg_work_layout(n_utt, n_frames_bound, C);
    if (work_bytes < w.total) return fail(DSP_EINVAL, "work buffer too small (%zu < %zu bytes)", work_bytes, w.total);
    hipStream_t st = (hipStream_t)stream;
    char* wp = static_cast<char*>(d_work);
    double* stats = unit_variance ? reinterpret_cast<double*>(wp + w.stats) : nullptr;
    int64_t* tile_off = reinterpret_cast<int64_t*>(wp + w.tile);
    DspRaggedTables pre;
    pre.shift = k512 ? 3 : 2;
    pre.group_off = reinterpret_cast<int32_t*>(wp + w.goff);
    pre.group_utt = reinterpret_cast<int32_t*>(wp + w.gutt);
    float* cep = delta_n >= 1 ? reinterpret_cast<float*>(wp + w.cep) : d_out;   // cepstra only: straight into the result
    // one small launch: group tables of the MFCC kernel, tile table of the delta pass, statistics zeroed -- unless
    // dsp_endpoint_layout_segments_batch has already left all of that in d_work
    if (!(flags & DSP_SEG_TABLES_READY))
        f512_build_group_tables(d_frame_offsets, n_utt, pre.shift, pre.group_off, pre.group_utt, st, tile_off, stats);
    bg.seg = d_segments;
    bg.stats = stats;
    rc = k512 ? fast512_launch(plan, d_wave, wave_dtype, bg, cep, (int64_t)C, st, &pre)
              : fast1536_launch(plan, d_wave, wave_dtype, bg, cep, (int64_t)C, st, &pre);
    if (rc != DSP_OK) return fail(rc < 0 ? rc : DSP_EHIP, "fused kernel launch failed");
    if (delta_n == 0) return DSP_OK;    // (unit variance: c0 still lacks -ln(var); dsp_model_finalize_segments_batch applies it)
    BatchGeom dg;
    memset(&dg, 0, sizeof(dg));
    dg.frame_off = d_frame_offsets;
    dg.total_frames = n_frames_bound;
    dg.n_utt = n_utt;
    int den = 0;
    for (int i = 1; i <= delta_n; ++i) den += i * i;
    const float inv_den = (float)(1.0 / (2.0 * den));
    const int64_t blocks = n_frames_bound / DT_TILE + n_utt;
    if (blocks > 0x7fffffff) return fail(DSP_EINVAL, "too many delta tiles");
    // (the tile -> utterance table exists only when the layout kernel built the tables)
    const int32_t* tile_utt = (flags & DSP_SEG_TABLES_READY) ? reinterpret_cast<const int32_t*>(wp + w.tutt) : nullptr;
    if (C == 13)
        delta_rows_kernel<13><<<(int)blocks, 256, lds, st>>>(cep, dg, C, delta_n, inv_den, d_out, 0, tile_off, d_segments, stats, tile_utt);
    else
        delta_rows_kernel<0><<<(int)blocks, 256, lds, st>>>(cep, dg, C, delta_n, inv_den, d_out, 0, tile_off, d_segments, stats, tile_utt);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_endpoint_layout_batch(const int32_t* d_endpoints, const int64_t* d_sample_offsets, int32_t n_utt,
                              double cfg_step, double rate, int32_t frame_len, int32_t frame_step,
                              const int64_t* d_jitter, int64_t* d_segments, int64_t* d_dst_offsets,
                              int64_t* d_frame_offsets, void* stream) {
    if (!d_endpoints || !d_sample_offsets || !d_segments || !d_dst_offsets || !d_frame_offsets || n_utt <= 0)
        return fail(DSP_EINVAL, "dsp_endpoint_layout_batch: bad arguments");
    if (!(cfg_step > 0.0) || !(rate > 0.0) || frame_len <= 0 || frame_step <= 0)
        return fail(DSP_EINVAL, "dsp_endpoint_layout_batch: step, rate, frame_len, frame_step must be > 0");
    endpoint_layout_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(d_endpoints, d_sample_offsets, n_utt, cfg_step, rate,
                                                               frame_len, frame_step, d_jitter, d_segments,
                                                               d_dst_offsets, d_frame_offsets);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_endpoint_layout_segments_batch(const int32_t* d_endpoints, const int64_t* d_sample_offsets, int32_t n_utt,
                                       double cfg_step, double rate, const int64_t* d_jitter, int64_t* d_segments,
                                       int64_t* d_dst_offsets, int64_t* d_frame_offsets, const dsp_plan* plan,
                                       int64_t n_frames_bound, void* d_work, size_t work_bytes, void* stream) {
    if (!d_endpoints || !d_sample_offsets || !d_segments || !d_dst_offsets || !d_frame_offsets || !plan || !d_work || n_utt <= 0)
        return fail(DSP_EINVAL, "dsp_endpoint_layout_segments_batch: bad arguments");
    if (!(cfg_step > 0.0) || !(rate > 0.0) || n_frames_bound <= 0)
        return fail(DSP_EINVAL, "dsp_endpoint_layout_segments_batch: step, rate, n_frames_bound must be > 0");
    if ((n_frames_bound >> 2) + n_utt > 0x3fffffff)   // the kernel's group and tile counters are int32 (as dsp_mfcc_delta_segments_batch checks)
        return fail(DSP_EINVAL, "dsp_endpoint_layout_segments_batch: batch too large");
    const SegWork w = seg_work_layout(n_utt, n_frames_bound, plan->C);
    if (work_bytes < w.total) return fail(DSP_EINVAL, "work buffer too small (%zu < %zu bytes)", work_bytes, w.total);
    char* wp = static_cast<char*>(d_work);
    endpoint_layout_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(
        d_endpoints, d_sample_offsets, n_utt, cfg_step, rate, plan->L, plan->S, d_jitter, d_segments, d_dst_offsets,
        d_frame_offsets, plan->d_fast ? 3 : 2 /* frames per group: 8 (NFFT = 512 kernel) or 4 (NFFT = 1536) */,
        reinterpret_cast<int32_t*>(wp + w.goff), reinterpret_cast<int32_t*>(wp + w.gutt),
        reinterpret_cast<int64_t*>(wp + w.tile), reinterpret_cast<double*>(wp + w.stats),
        reinterpret_cast<int32_t*>(wp + w.tutt));
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_model_finalize_batch(const float* d_mfcc, int64_t ld_in, const int64_t* d_frame_offsets, int32_t n_utt,
                             int32_t C, int32_t N, int32_t max_len, float* d_out, int32_t* d_len0, void* stream) {
    if (!d_mfcc || !d_frame_offsets || !d_out || !d_len0 || n_utt <= 0)
        return fail(DSP_EINVAL, "dsp_model_finalize_batch: bad arguments");
    if (N < 1) return fail(DSP_EINVAL, "N must be an integer >= 1");  // base.py:71-72
    if (C <= 0 || C > 32 || max_len <= 0) return fail(DSP_EINVAL, "need 0 < C <= 32 and max_len > 0");
    if (ld_in == 0) ld_in = C;
    if (ld_in < C) return fail(DSP_EINVAL, "ld_in %lld < C %d", (long long)ld_in, C);
    const size_t lds = ((size_t)(max_len + 2 * N) + (size_t)(max_len + N)) * C * sizeof(float);
    if (lds > 64 * 1024) return fail(DSP_EINVAL, "max_len * C too large for the LDS tile (%zu bytes)", lds);
    model_finalize_kernel<<<n_utt, 256, lds, (hipStream_t)stream>>>(d_mfcc, ld_in, d_frame_offsets, n_utt, C, N,
                                                                   max_len, d_out, d_len0);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_model_finalize_segments_batch(const float* d_mfcc, int64_t ld_in, const int64_t* d_frame_offsets,
                                      const int64_t* d_segments, const void* d_work, int32_t n_utt, int32_t C, int32_t N,
                                      int32_t max_len, float* d_out, int32_t* d_len0, void* stream) {
    if (!d_mfcc || !d_frame_offsets || !d_segments || !d_work || !d_out || !d_len0 || n_utt <= 0)
        return fail(DSP_EINVAL, "dsp_model_finalize_segments_batch: bad arguments");
    if (N < 1) return fail(DSP_EINVAL, "N must be an integer >= 1");  // base.py:71-72
    if (C <= 0 || C > 32 || max_len <= 0) return fail(DSP_EINVAL, "need 0 < C <= 32 and max_len > 0");
    if (ld_in == 0) ld_in = C;
    if (ld_in < C) return fail(DSP_EINVAL, "ld_in %lld < C %d", (long long)ld_in, C);
    const size_t lds = ((size_t)(max_len + 2 * N) + (size_t)(max_len + N)) * C * sizeof(float);
    if (lds > 64 * 1024) return fail(DSP_EINVAL, "max_len * C too large for the LDS tile (%zu bytes)", lds);
    // the statistics sit at the start of the work buffer of dsp_mfcc_delta_segments_batch (seg_work_layout)
    model_finalize_kernel<<<n_utt, 256, lds, (hipStream_t)stream>>>(d_mfcc, ld_in, d_frame_offsets, n_utt, C, N, max_len, d_out, d_len0,
                                                                   d_segments, static_cast<const double*>(d_work));
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_model_timefeat_batch(const double* d_amp_sum, const int64_t* d_frame_offsets, int32_t n_utt,
                             int32_t frame_len, int32_t max_len, float* d_out, void* stream) {
    if (!d_amp_sum || !d_frame_offsets || !d_out || n_utt <= 0 || frame_len <= 0 || max_len <= 0)
        return fail(DSP_EINVAL, "dsp_model_timefeat_batch: bad arguments");
    timefeat_finalize_kernel<<<n_utt, 64, 0, (hipStream_t)stream>>>(d_amp_sum, d_frame_offsets, n_utt, frame_len, max_len, d_out);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_pitch_scores_batch(const float* d_sig, const int64_t* d_sample_offsets, const int64_t* d_frame_offsets,
                           int32_t n_utt, int64_t n_frames_total, int64_t uniform_samples, int32_t frame_len,
                           int32_t frame_step, const float* d_taps, int32_t center_clip, int32_t lag_min,
                           int32_t lag_max, float* d_scores, void* stream) {
    if (!d_taps || !d_scores) return fail(DSP_EINVAL, "dsp_pitch_scores_batch: NULL taps/output");
    if (frame_len <= 0 || frame_len > PITCH_MAX_L || frame_step <= 0)
        return fail(DSP_EINVAL, "need 0 < frame_len <= %d and frame_step > 0", PITCH_MAX_L);
    if (lag_min < 0 || lag_max <= lag_min) return fail(DSP_EINVAL, "need 0 <= lag_min < lag_max");
    int rc = check_geom(d_sig, DSP_WAVE_F32, d_sample_offsets, d_frame_offsets, n_utt, n_frames_total, uniform_samples);
    if (rc != DSP_OK) return rc;
    if (n_frames_total > 0x7fffffff) return fail(DSP_EINVAL, "too many frames for one launch");
    BatchGeom bg = make_geom(d_sample_offsets, d_frame_offsets, n_utt, n_frames_total, uniform_samples, frame_len, frame_step);
    int P = 1;
    while (P < frame_len) P <<= 1;
    // register-blocked kernel (one wave per frame) where its shape constraints hold
    const int n_lags = lag_max - lag_min;
    const int W = frame_len <= 384 ? 3 : (frame_len <= 512 ? 4 : 0);
    if (W != 0 && (lag_min % 4) == 0 && n_lags <= 256 && lag_max + 8 <= PITCH2_GUARD && !g_force_generic) {
        const int Lp = (frame_len + W - 1) / W * W;
        const size_t lds2 = (2 * (size_t)Lp + 2 * (size_t)Lp + (size_t)Lp + PITCH2_GUARD) * sizeof(float);
        const float2* tp = reinterpret_cast<const float2*>(d_taps);
        if (W == 3)
            pitch_scores_kernel_v2<3><<<(int)n_frames_total, 64, lds2, (hipStream_t)stream>>>(
                d_sig, bg, frame_len, frame_step, P, tp, center_clip ? 1 : 0, lag_min, n_lags, d_scores);
        else
            pitch_scores_kernel_v2<4><<<(int)n_frames_total, 64, lds2, (hipStream_t)stream>>>(
                d_sig, bg, frame_len, frame_step, P, tp, center_clip ? 1 : 0, lag_min, n_lags, d_scores);
        HIP_TRY(hipGetLastError());
        return DSP_OK;
    }
    const size_t lds = ((size_t)P + (size_t)((frame_len + 3) & ~3) + 3 * (size_t)frame_len) * sizeof(float);
    pitch_scores_kernel<<<(int)n_frames_total, PITCH_THREADS, lds, (hipStream_t)stream>>>(
        d_sig, bg, frame_len, frame_step, P, reinterpret_cast<const float2*>(d_taps), center_clip ? 1 : 0, lag_min,
        lag_max - lag_min, d_scores);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_pitch_track_batch(const float* d_scores, const int64_t* d_frame_offsets, int32_t n_utt, int32_t n_lags,
                          int32_t bias, int32_t degree, double* d_pitch, void* stream) {
    if (!d_scores || !d_frame_offsets || !d_pitch || n_utt <= 0) return fail(DSP_EINVAL, "dsp_pitch_track_batch: bad arguments");
    if (n_lags <= 0 || n_lags > 256) return fail(DSP_EINVAL, "need 0 < n_lags <= 256");
    if (degree != 2) return fail(DSP_EINVAL, "smoothing degree %d is not served on the device (the reference only uses 2)", degree);
    pitch_track_kernel<<<n_utt, 64, 0, (hipStream_t)stream>>>(d_scores, d_frame_offsets, n_lags, bias, d_pitch);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_endpoint_rule_batch(const double* d_amp_sum, const int32_t* d_zcr, const int64_t* d_frame_offsets,
                            int32_t n_utt, int32_t frame_len, double cfg_frame, double cfg_step,
                            int32_t* d_endpoints, void* stream) {
    if (!d_amp_sum || !d_zcr || !d_frame_offsets || !d_endpoints || n_utt <= 0 || frame_len <= 0)
        return fail(DSP_EINVAL, "dsp_endpoint_rule_batch: bad arguments");
    if (!(cfg_frame > 0.0) || !(cfg_step > 0.0)) return fail(DSP_EINVAL, "cfg.frame / cfg.step must be > 0");
    if (2 * (int)(0.100 / cfg_step) > DSP_MAX_SIL)
        return fail(DSP_EINVAL, "cfg.step %g gives a silence window > %d frames", cfg_step, DSP_MAX_SIL);
    endpoint_rule_kernel<<<n_utt, 64, 0, (hipStream_t)stream>>>(
        d_amp_sum, d_zcr, d_frame_offsets, n_utt, frame_len, cfg_frame, cfg_step, d_endpoints);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_acr_gate_batch(const void* d_wave, int wave_dtype, const int64_t* d_sample_offsets,
                       const int64_t* d_frame_offsets, int32_t n_utt, int64_t n_frames_total, int64_t uniform_samples,
                       int32_t frame_len, int32_t frame_step, int32_t lag_lo, int32_t lag_hi, double thresh,
                       uint8_t* d_voiced, void* stream) {
    if (!d_voiced) return fail(DSP_EINVAL, "dsp_acr_gate_batch: d_voiced is NULL");
    if (frame_len <= 0 || frame_step <= 0) return fail(DSP_EINVAL, "frame_len / frame_step must be > 0");
    if (lag_lo < 1 || lag_hi <= lag_lo || lag_hi > frame_len)
        return fail(DSP_EINVAL, "lags [%d, %d) must lie in [1, frame_len = %d]", lag_lo, lag_hi, frame_len);
    const size_t lds = (size_t)4 * frame_len * sizeof(double);
    if (lds > 64 * 1024) return fail(DSP_EINVAL, "frame_len %d too long for the autocorrelation gate (<= 2048)", frame_len);
    int rc = check_geom(d_wave, wave_dtype, d_sample_offsets, d_frame_offsets, n_utt, n_frames_total, uniform_samples);
    if (rc != DSP_OK) return rc;
    BatchGeom bg = make_geom(d_sample_offsets, d_frame_offsets, n_utt, n_frames_total, uniform_samples, frame_len, frame_step);
    if (uniform_samples > 0 && bg.uniform_frames * n_utt != n_frames_total)
        return fail(DSP_EINVAL, "n_frames_total %lld != n_utt*T (%d*%lld)", (long long)n_frames_total, n_utt, (long long)bg.uniform_frames);
    if (n_frames_total <= 0) return DSP_OK;
    const int64_t blocks = (n_frames_total + 3) / 4;
    if (blocks > 0x7fffffff) return fail(DSP_EINVAL, "too many frames");
    hipStream_t st = (hipStream_t)stream;
    if (wave_dtype == DSP_WAVE_I16)
        acr_gate_kernel<DSP_WAVE_I16><<<(int)blocks, 256, lds, st>>>(d_wave, bg, frame_len, frame_step, lag_lo, lag_hi, thresh, d_voiced);
    else
        acr_gate_kernel<DSP_WAVE_F32><<<(int)blocks, 256, lds, st>>>(d_wave, bg, frame_len, frame_step, lag_lo, lag_hi, thresh, d_voiced);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_endpoint_rule_acr_batch(const double* d_amp_sum, const int32_t* d_zcr, const uint8_t* d_voiced,
                                const int64_t* d_frame_offsets, int32_t n_utt, int32_t frame_len, double cfg_frame,
                                double cfg_step, int32_t* d_endpoints, void* stream) {
    if (!d_amp_sum || !d_zcr || !d_voiced || !d_frame_offsets || !d_endpoints || n_utt <= 0 || frame_len <= 0)
        return fail(DSP_EINVAL, "dsp_endpoint_rule_acr_batch: bad arguments");
    if (!(cfg_frame > 0.0) || !(cfg_step > 0.0)) return fail(DSP_EINVAL, "cfg.frame / cfg.step must be > 0");
    if (2 * (int)(0.100 / cfg_step) > DSP_MAX_SIL)
        return fail(DSP_EINVAL, "cfg.step %g gives a silence window > %d frames", cfg_step, DSP_MAX_SIL);
    endpoint_rule_kernel<<<n_utt, 64, 0, (hipStream_t)stream>>>(
        d_amp_sum, d_zcr, d_frame_offsets, n_utt, frame_len, cfg_frame, cfg_step, d_endpoints, d_voiced);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_resample_layout_batch(const int64_t* d_src_offsets, int32_t n_utt, int64_t src_rate, int64_t dst_rate,
                              int32_t frame_len, int32_t frame_step, int64_t* d_dst_offsets, int64_t* d_frame_offsets,
                              void* stream) {
    if (!d_src_offsets || !d_frame_offsets || n_utt <= 0 || frame_len <= 0 || frame_step <= 0)
        return fail(DSP_EINVAL, "dsp_resample_layout_batch: bad arguments");
    if (dst_rate < 0 || (dst_rate > 0 && (src_rate <= 0 || dst_rate >= src_rate || !d_dst_offsets)))
        return fail(DSP_EINVAL, "dsp_resample_layout_batch: need 0 < dst_rate < src_rate and d_dst_offsets (a decimation), or dst_rate = 0");
    resample_layout_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(d_src_offsets, n_utt, src_rate, dst_rate, frame_len, frame_step,
                                                               d_dst_offsets, d_frame_offsets);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_decimate_batch(const float* d_in, const int64_t* d_src_offsets, const int64_t* d_dst_offsets, int32_t n_utt,
                       int64_t n_out_bound, int64_t src_rate, int64_t dst_rate, float* d_out, void* stream) {
    if (!d_in || !d_src_offsets || !d_dst_offsets || !d_out || n_utt <= 0)
        return fail(DSP_EINVAL, "dsp_decimate_batch: bad arguments");
    if (src_rate <= 0 || dst_rate <= 0 || dst_rate >= src_rate) return fail(DSP_EINVAL, "dsp_decimate_batch: need 0 < dst_rate < src_rate");
    if (n_out_bound <= 0) return DSP_OK;
    decimate_gather_kernel<<<grid_for(n_out_bound, 256), 256, 0, (hipStream_t)stream>>>(d_in, d_src_offsets, d_dst_offsets, n_utt,
                                                                                         src_rate, dst_rate, d_out);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_model_pitchfeat_batch(const double* d_pitch, const int64_t* d_frame_offsets, int32_t n_utt, int32_t max_len,
                              float* d_out, void* stream) {
    if (!d_pitch || !d_frame_offsets || !d_out || n_utt <= 0 || max_len <= 0)
        return fail(DSP_EINVAL, "dsp_model_pitchfeat_batch: bad arguments");
    pitchfeat_finalize_kernel<<<n_utt, 64, 0, (hipStream_t)stream>>>(d_pitch, d_frame_offsets, n_utt, max_len, d_out);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

int dsp_pitch_rows_batch(double* d_rows, const int64_t* d_frame_offsets, int32_t n_utt, int32_t n_lags, int32_t bias,
                         int32_t degree, int32_t flags, double* d_pitch, void* stream) {
    if (!d_rows || !d_frame_offsets || n_utt <= 0 || n_lags <= 0 || degree < 0)
        return fail(DSP_EINVAL, "dsp_pitch_rows_batch: bad arguments");
    if ((flags & 4) && !(flags & 2)) return fail(DSP_EINVAL, "dsp_pitch_rows_batch: the repair sweeps (4) need the arg-max (2)");
    if ((flags & 2) && !d_pitch) return fail(DSP_EINVAL, "dsp_pitch_rows_batch: d_pitch is NULL");
    pitch_rows_kernel<<<n_utt, 64, 0, (hipStream_t)stream>>>(d_rows, d_frame_offsets, n_lags, bias, degree, flags, d_pitch);
    HIP_TRY(hipGetLastError());
    return DSP_OK;
}

}  // extern "C"
