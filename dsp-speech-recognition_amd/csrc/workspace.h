// Device workspaces: the ragged-batch index tables (group / tile prefixes, KBs) and the dense cepstra scratch
// of dsp_mfcc_delta_batch (up to 256 MiB).
//
// A workspace is produced and consumed by kernels of ONE call on ONE stream, so the host never needs to wait
// for it -- but the buffer must not be handed to another call while those kernels may still be running.  Each
// buffer carries the stream of its last user and an event recorded behind that user:
//   * a call on the SAME stream may take the buffer at once (stream order already protects it),
//   * a call on another stream may take it once the event has completed,
//   * a stream under HIP-graph capture gets none (see acquire),
//   * otherwise a new buffer is allocated -- until the pool holds DSP_WS_POOL_CAP_MB (default 1024) MiB; past
//     that, completed buffers are freed, and if that is not enough the call WAITS for the best-fitting buffer
//     in flight instead of growing the pool (a host that queues thousands of steps ahead keeps a bounded pool).
// Selection is best fit, and a request never takes a buffer more than 8x its size while a new small one can
// still be allocated (the 256 MiB scratch is not handed to a 4 KB table).
//
// hipMallocAsync/hipFreeAsync were used here first; with them, ragged calls queued back to back on
// the legacy default stream gave intermittently wrong results on gfx950 / ROCm 7.2 (see
// tests/test_gpu_batch.py::test_ragged_calls_queued_back_to_back), hence the explicit event guard.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

struct DspWorkspace {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    hipStream_t last_stream = nullptr;   // stream `done` was recorded on
    bool used = false;                   // `done` has been recorded at least once
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
    // Returns nullptr on HIP failure (callers fall back or report).  The buffer stays leased until release().
    // `st` is the stream the caller will use it on.
    DspWorkspace* acquire(size_t bytes, hipStream_t st) {
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
        {   // A stream that is being captured into a HIP graph gets nothing: event queries and allocations are
            // not allowed there (hipErrorStreamCaptureUnsupported invalidates the capture), and a buffer baked into a
            // graph would be reused by every replay behind the pool's back.  Callers fall back or report.
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
            if (cs != hipStreamCaptureStatusNone) return nullptr;
        }
        std::lock_guard<std::mutex> lk(mu_);
        static const bool dbg = getenv("DSP_WS_DEBUG") != nullptr;
        if (dbg) {
            fprintf(stderr, "[ws] acquire %zu B on stream %p; pool:", bytes, (void*)st);
            for (DspWorkspace* w : all_) fprintf(stderr, " (%zu%s%s s=%p)", w->bytes, w->leased ? " leased" : "", w->used ? "" : " new", (void*)w->last_stream);
            fprintf(stderr, "\n");
        }
        // round up so that slightly larger batches reuse the buffer
        size_t want = 4096;
        while (want < bytes) want *= 2;
        const size_t cap = cap_bytes();
        const bool may_grow = total_ + want <= cap;
        // 1. best fit among the buffers last used on this stream: stream order is enough -- for an ordinary stream
        //    handle.  hipStreamPerThread is one handle value naming a different stream in every thread, so it never
        //    takes this shortcut (step 2 asks the buffer's event instead).
        const bool same_stream_ok = st != hipStreamPerThread;
        DspWorkspace* best = nullptr;
        for (DspWorkspace* w : all_)
            if (!w->leased && w->device == dev && w->bytes >= bytes && (!w->used || (same_stream_ok && w->last_stream == st)) &&
                (!best || w->bytes < best->bytes)) best = w;
        if (best && (best->bytes <= 8 * want || !may_grow)) return lease(best);
        // 2. best fit among the buffers of other streams whose last user has finished
        best = nullptr;
        for (DspWorkspace* w : all_) {
            if (w->leased || w->device != dev || w->bytes < bytes || (best && w->bytes >= best->bytes)) continue;
            if (w->used && (w->last_stream != st || !same_stream_ok)) {
                const hipError_t qe = hipEventQuery(w->done);
                if (qe != hipSuccess) {  // still in flight (or error)
                    if (dbg) fprintf(stderr, "[ws]   query %zu B: %s\n", w->bytes, hipGetErrorName(qe));
                    (void)hipGetLastError();
                    continue;
                }
            }
            best = w;
        }
        if (best && (best->bytes <= 8 * want || !may_grow)) return lease(best);
        // 3. grow, inside the cap: first give back completed buffers that nobody holds
        if (!may_grow) trim(dev, want, cap);
        if (total_ + want <= cap || all_.empty()) {
            DspWorkspace* w = new DspWorkspace();
            if (hipMalloc(&w->ptr, want) == hipSuccess && hipEventCreateWithFlags(&w->done, hipEventDisableTiming) == hipSuccess) {
                w->bytes = want;
                w->device = dev;
                total_ += want;
                all_.push_back(w);
                return lease(w);
            }
            (void)hipGetLastError();
            if (w->ptr) (void)hipFree(w->ptr);
            delete w;
        }
        // 4. the pool is full (or the device is): wait for the best-fitting buffer in flight
        best = nullptr;
        for (DspWorkspace* w : all_)
            if (!w->leased && w->device == dev && w->bytes >= bytes && (!best || w->bytes < best->bytes)) best = w;
        if (!best) return nullptr;
        if (best->used && (best->last_stream != st || !same_stream_ok) && hipEventSynchronize(best->done) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return lease(best);
    }

    // Marks the buffer reusable: at once by later work on `st`, by other streams once everything queued on
    // `st` so far has finished.
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
        w->last_stream = st;
        w->used = true;
        w->leased = false;
        return e == hipSuccess ? 0 : -1;
    }

    // diagnostics (tests): buffers and bytes currently held by the pool
    void stats(long long* n_buffers, long long* bytes) {
        std::lock_guard<std::mutex> lk(mu_);
        if (n_buffers) *n_buffers = (long long)all_.size();
        if (bytes) *bytes = (long long)total_;
    }

  private:
    static size_t cap_bytes() {
        static const size_t cap = [] {
            const char* e = getenv("DSP_WS_POOL_CAP_MB");
            const long mb = e ? atol(e) : 1024;
            return (size_t)(mb > 0 ? mb : 1024) << 20;
        }();
        return cap;
    }
    DspWorkspace* lease(DspWorkspace* w) {
        w->leased = true;
        return w;
    }
    // frees completed, unleased buffers (largest first) until `want` more bytes fit under `cap`
    void trim(int dev, size_t want, size_t cap) {
        while (total_ + want > cap) {
            size_t pick = all_.size();
            for (size_t i = 0; i < all_.size(); ++i) {
                DspWorkspace* w = all_[i];
                if (w->leased || w->device != dev) continue;
                if (w->used && hipEventQuery(w->done) != hipSuccess) { (void)hipGetLastError(); continue; }
                if (pick == all_.size() || w->bytes > all_[pick]->bytes) pick = i;
            }
            if (pick == all_.size()) return;
            DspWorkspace* w = all_[pick];
            (void)hipFree(w->ptr);
            (void)hipEventDestroy(w->done);
            total_ -= w->bytes;
            all_.erase(all_.begin() + (long)pick);
            delete w;
        }
    }

    std::mutex mu_;
    std::vector<DspWorkspace*> all_;  // lives for the process
    size_t total_ = 0;
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
