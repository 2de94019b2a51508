// Small device workspaces for the ragged-batch index tables (group / tile prefixes).
//
// The tables are produced and consumed by kernels of ONE call on ONE stream, so the host never
// needs to wait for them -- but the buffer must not be handed to another call while those kernels
// may still be running.  Each buffer therefore carries an event recorded behind its last user; a
// buffer is reused only once that event has completed, otherwise a new one is allocated (this only
// happens while warming up: steady state is lock + hipEventQuery, no allocation, no host sync).
//
// hipMallocAsync/hipFreeAsync were used here first; with them, ragged calls queued back to back on
// the legacy default stream gave intermittently wrong results on gfx950 / ROCm 7.2 (see
// tests/test_gpu_batch.py::test_ragged_calls_queued_back_to_back), hence the explicit event guard.
#pragma once

#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

struct DspWorkspace {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    int device = -1;
    bool leased = false;
};

#ifdef DSP_WS_MALLOC_ASYNC
// Diagnostic build only (tools/diag_malloc_async.py): the round-1 scheme this pool replaced -- every call
// takes its tables from hipMallocAsync on the call's stream and hands them back with hipFreeAsync.
inline hipStream_t& dsp_ws_diag_stream() {
    thread_local hipStream_t st = nullptr;
    return st;
}
#endif

class DspWorkspacePool {
  public:
    // Returns nullptr on HIP failure.  The buffer stays leased until release().
    DspWorkspace* acquire(size_t bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        if (bytes < 256) bytes = 256;
#ifdef DSP_WS_MALLOC_ASYNC
        {
            DspWorkspace* w = new DspWorkspace();
            if (hipMallocAsync(&w->ptr, bytes, dsp_ws_diag_stream()) != hipSuccess) { delete w; return nullptr; }
            w->bytes = bytes;
            w->device = dev;
            return w;
        }
#endif
        std::lock_guard<std::mutex> lk(mu_);
        for (DspWorkspace* w : all_) {
            if (w->leased || w->device != dev || w->bytes < bytes) continue;
            if (hipEventQuery(w->done) != hipSuccess) {  // still in flight (or error): skip
                (void)hipGetLastError();
                continue;
            }
            w->leased = true;
            return w;
        }
        DspWorkspace* w = new DspWorkspace();
        // round up so that slightly larger batches reuse the buffer
        size_t cap = 4096;
        while (cap < bytes) cap *= 2;
        if (hipMalloc(&w->ptr, cap) != hipSuccess ||
            hipEventCreateWithFlags(&w->done, hipEventDisableTiming) != hipSuccess) {
            if (w->ptr) (void)hipFree(w->ptr);
            delete w;
            return nullptr;
        }
        w->bytes = cap;
        w->device = dev;
        w->leased = true;
        all_.push_back(w);
        return w;
    }

    // Marks the buffer reusable once everything queued on `st` so far has finished.
    int release(DspWorkspace* w, hipStream_t st) {
#ifdef DSP_WS_MALLOC_ASYNC
        {
            const hipError_t e = hipFreeAsync(w->ptr, st);
            delete w;
            return e == hipSuccess ? 0 : -1;
        }
#endif
        hipError_t e = hipEventRecord(w->done, st);
        std::lock_guard<std::mutex> lk(mu_);
        w->leased = false;
        return e == hipSuccess ? 0 : -1;
    }

  private:
    std::mutex mu_;
    std::vector<DspWorkspace*> all_;  // lives for the process (a handful of small buffers)
};

inline DspWorkspacePool& dsp_workspace_pool() {
    static DspWorkspacePool* pool = new DspWorkspacePool();  // never destroyed: no teardown-order issues with HIP
    return *pool;
}

// Raises a kernel's dynamic-LDS limit when needed.  The attribute is per device, so the largest
// size already granted is remembered per device (`granted` is one static array per kernel
// instantiation); concurrent callers may both set it, which is harmless.
#define DSP_MAX_DEVICES 64
inline int dsp_ensure_dynamic_lds(const void* kernel, size_t bytes, size_t (&granted)[DSP_MAX_DEVICES]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DSP_MAX_DEVICES) return -1;
    if (bytes <= granted[dev]) return 0;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return -1;
    granted[dev] = bytes;
    return 0;
}


// Compute units of the current device (256 on MI355X), read once per device: the persistent kernels size
// their grids as CUs x resident workgroups per CU.
inline int dsp_cu_count() {
    static int cached[DSP_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DSP_MAX_DEVICES) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
