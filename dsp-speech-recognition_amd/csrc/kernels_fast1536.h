// Specialised fused MFCC kernel for NFFT = 1536 = 3 * 512: the transform size the reference's own
// driver uses (model.py:74 -- 30 ms frames at 44.1 / 48 kHz, 26 mel filters).
//
// Work decomposition (gfx950, wave64):
//   * one wavefront = 4 consecutive frames of one utterance, 16 lanes per frame (lane = 16 f + c);
//   * the 1536-point real DFT is split by one decimation-in-frequency radix-3 step,
//         X[3q + s] = FFT512(y_s)[q],  y_s[m] = (a + W3^s b + W3^2s c) W1536^(s m),
//         a, b, c = windowed samples m, m + 512, m + 1024,
//     and because the input is real only s = 0 (y_0 real, bins 3q, q <= 256) and s = 1 (y_1
//     complex, bins 3q + 1 for q < 256 and, mirrored, bins 1535 - 3q for q >= 256) are needed;
//   * each 512-point FFT is split m = 16 n1 + n2 exactly like the NFFT = 512 kernel: lane c owns
//     column n2 = c -> one in-register complex FFT32 over n1, twiddle W512^(c k1), exchange through
//     LDS (two rounds of 16 rows, XOR-swizzled 16-byte slots, conflict-free both ways), then lane c
//     owns rows k1 = c and c + 16 -> two in-register FFT16 over n2 -> Y[k1 + 32 k2]; the real y_0
//     takes a cheaper route (FFT16 of packed pairs + untangle -> 17 rows, f1536_rfft512);
//   * all power values stay in registers until both FFTs are done, so the per-wave LDS region is
//     reused three times (staged samples -> exchange buffer -> one 776-float spectrum row per
//     frame) and 8 waves fit a CU next to ~30 KB of tables;
//   * |X|^2 / 1536 -> LDS row -> table-driven sparse mel: the filters are cut into 64 equal slots,
//     lane s sums slot s for all four frames, the pieces of a filter are added up through LDS ->
//     log -> DCT*lifter partial sums (lane c owns filters c, c + 16, ...) -> 4-step DPP
//     all-reduce over the frame's 16 lanes -> store.
// Samples are read from HBM once per wave (3 S + 1536 of them, 16 B per lane), pre-emphasised on
// the fly and staged in LDS exactly as in kernels_fast512.h (whose staging helpers are reused).
#pragma once

#include "kernels_fast512.h"

#define F1536_PS_STRIDE 776    // 769 bins + padding; == 8 (mod 64)
#define F1536_XSTRIDE 544      // exchange buffer per frame: 16 rows x 32 floats + 32 (bank offset)
#define F1536_MAX_NI 4
#define F1536_R3_ROW 164       // floats per column row of the radix-3 table (== 36 mod 64: conflict-free b128 rows)
#define F1536_SLOTS 64         // mel work items per wave (one per lane)
#define F1536_PIECES 8         // a filter is cut into at most this many slots
#define F1536_PART_STRIDE F1536_PS_STRIDE   // partial sums overwrite the head of each frame's (dead) spectrum row
#ifndef F1536_WAVES
#define F1536_WAVES 8
#endif

struct F1536Params {
    const float* tables;   // device blob copied to LDS by every workgroup
    int32_t tab_floats;
    int32_t off_w3, off_tw, off_dct, off_melw, off_mels, off_pidx;
    int32_t melw_row;      // floats per mel slot row (odd number of 16-byte units: conflict-free b128)
    int32_t mel_blocks;    // 8-tap blocks per mel slot
    int32_t off_part;      // float offset of the partial-sum rows inside the wave region
    int32_t L, S, M, C, append_energy;
    float preemph;
    int32_t span_vec;      // ceil((3 S + 1536) / 4): 16-byte vectors staged per wave
    int32_t wave_floats;   // per-wave LDS region
    int64_t groups_per_utt, total_groups;
    const int32_t* group_off;   // ragged: [B+1] prefix of ceil(T_b / 4)
    const int32_t* group_utt;
};

struct Fast1536Plan {
    float* d_tables;
    F1536Params P;
    int variant;
};

// Sum over the 16 lanes of a frame (one DPP row); every lane ends with the total.
__device__ __forceinline__ float frame16_allreduce(float v) {
    v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);  // row_half_mirror
    v += dpp_f32<0x140>(v);  // row_mirror
    return v;
}

template <bool RAGGED>
__device__ __forceinline__ F512Group f1536_locate(const F1536Params& P, const BatchGeom& bg, int G) {
    F512Group g;
    if constexpr (RAGGED) {
        g.utt = P.group_utt[G];
        g.t0 = (G - P.group_off[g.utt]) * 4;
        g.s0 = bg.sample_off[g.utt];
        g.nsamp = (int)(bg.sample_off[g.utt + 1] - g.s0);
        if (bg.seg != nullptr) {   // the utterance is a segment of that range, read in place (kernels_fast512.h)
            const int64_t lo = bg.seg[2 * g.utt];
            g.nsamp = (int)(bg.seg[2 * g.utt + 1] - lo);
            g.s0 += lo;
        }
        g.row0 = bg.frame_off[g.utt];
        g.T = (int)(bg.frame_off[g.utt + 1] - g.row0);
    } else {
        const int gpu = (int)P.groups_per_utt;
        g.utt = G / gpu;
        g.t0 = (G - g.utt * gpu) * 4;
        g.nsamp = (int)bg.uniform_samples;
        g.T = (int)bg.uniform_frames;
        g.s0 = (int64_t)g.utt * bg.uniform_samples;
        g.row0 = (int64_t)g.utt * bg.uniform_frames;
    }
    return g;
}

// Complex 512-point FFT of one frame spread over its 16 lanes.  In: z[n1] = y[16 n1 + c].
// Out: pa[k2] = |Y[c + 32 k2]|^2 * scale, pb[k2] = |Y[16 + c + 32 k2]|^2 * scale.
__device__ __forceinline__ void f1536_cfft512(cpx (&z)[32], float* __restrict__ xb, const float2* __restrict__ s_tw,
                                              int c, float scale, float (&pa)[16], float (&pb)[16]) {
    FFTReg<32>::template run<32>(z);
#pragma unroll
    for (int k1 = 1; k1 < 32; ++k1) {
        const float2 t = s_tw[(k1 - 1) * 16 + c];
        z[k1] = cmulc(z[k1], t.x, t.y);
    }
    float* wr = xb + 2 * (c & 1);
    const int ch = c >> 1;
    // round A: rows 0..15.  Element (row r, column c) sits in 16-byte slot (c >> 1) ^ (r >> 1).
#pragma unroll
    for (int r = 0; r < 16; ++r)
        *reinterpret_cast<float2*>(wr + r * 32 + 4 * (ch ^ (r >> 1))) = make_float2(z[r].x, z[r].y);
    F512_FENCE();
    cpx u[16];
    const float* rd = xb + c * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(rd + 4 * (i ^ ch));
        u[2 * i] = {t.x, t.y};
        u[2 * i + 1] = {t.z, t.w};
    }
    F512_FENCE();
    // round B: rows 16..31 (LDS operations of one wave execute in order: the reads above are done)
#pragma unroll
    for (int r = 0; r < 16; ++r)
        *reinterpret_cast<float2*>(wr + r * 32 + 4 * (ch ^ (r >> 1))) = make_float2(z[16 + r].x, z[16 + r].y);
    F512_FENCE();
    FFTReg<16>::run(u);
#pragma unroll
    for (int k = 0; k < 16; ++k) pa[k] = scale * fmaf(u[k].x, u[k].x, u[k].y * u[k].y);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(rd + 4 * (i ^ ch));
        u[2 * i] = {t.x, t.y};
        u[2 * i + 1] = {t.z, t.w};
    }
    F512_FENCE();
    FFTReg<16>::run(u);
#pragma unroll
    for (int k = 0; k < 16; ++k) pb[k] = scale * fmaf(u[k].x, u[k].x, u[k].y * u[k].y);
}

// Real 512-point FFT of one frame spread over its 16 lanes.  In: r[n1] = y[16 n1 + c] (real).
// Column pass: FFT16 of r[2m] + i r[2m+1], untangled to the 17 rows k1 = 0..16 of the real-input
// FFT32 (rows 17..31 are their mirror images and never materialise).  Rows 0..15 go through the
// exchange and one FFT16 per lane; row 16 sits in the spare 17th row of the exchange buffer and is
// transformed by lane 0 alone.  All values carry a factor 2 (folded into `scale`).
// Out: pa[k2] = |Y[c + 32 k2]|^2 (lane c; k2 >= 8 are the mirror bins 512 - c - 32 k2 for c >= 1),
//      p16[k2] = |Y[16 + 32 k2]|^2 for k2 < 8 (valid in lane 0 only).
__device__ __forceinline__ void f1536_rfft512(const float (&r)[32], float* __restrict__ xb,
                                              const float2* __restrict__ s_tw, int c, float scale,
                                              float (&pa)[16], float (&p16)[8]) {
    cpx g[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) g[m] = {r[2 * m], r[2 * m + 1]};
    FFTReg<16>::run(g);
    cpx R[17];
    R[0] = {2.f * (g[0].x + g[0].y), 0.f};
    R[16] = {2.f * (g[0].x - g[0].y), 0.f};
    R[8] = {2.f * g[8].x, -2.f * g[8].y};
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const cpx a = g[k], b = g[16 - k];
        const cpx e = {a.x + b.x, a.y - b.y};             // A + conj(B)
        const cpx dd = {a.x - b.x, a.y + b.y};            // A - conj(B)
        const cpx o = {dd.y, -dd.x};                      // -i (A - conj(B))
        const cpx t = cmulc(o, DSP_COS32[k], -DSP_SIN32[k]);   // W32^k * o
        R[k] = {e.x + t.x, e.y + t.y};
        R[16 - k] = {e.x - t.x, t.y - e.y};               // conj(e - t)
    }
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        const float2 t = s_tw[(k1 - 1) * 16 + c];
        R[k1] = cmulc(R[k1], t.x, t.y);
    }
    {
        const float2 t = s_tw[15 * 16 + c];               // row 16 is real before the twiddle
        R[16] = {R[16].x * t.x, R[16].x * t.y};
    }
    float* wr = xb + 2 * (c & 1);
    const int ch = c >> 1;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr)
        *reinterpret_cast<float2*>(wr + rr * 32 + 4 * (ch ^ (rr >> 1))) = make_float2(R[rr].x, R[rr].y);
    *reinterpret_cast<float2*>(wr + 16 * 32 + 4 * ch) = make_float2(R[16].x, R[16].y);
    F512_FENCE();
    cpx u[16];
    const float* rd = xb + c * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(rd + 4 * (i ^ ch));
        u[2 * i] = {t.x, t.y};
        u[2 * i + 1] = {t.z, t.w};
    }
    F512_FENCE();
    FFTReg<16>::run(u);
#pragma unroll
    for (int k = 0; k < 16; ++k) pa[k] = scale * fmaf(u[k].x, u[k].x, u[k].y * u[k].y);
#pragma unroll
    for (int k = 0; k < 8; ++k) p16[k] = 0.f;
    if (c == 0) {
        cpx v[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(xb + 16 * 32 + 4 * i);
            v[2 * i] = {t.x, t.y};
            v[2 * i + 1] = {t.z, t.w};
        }
        FFTReg<16>::run(v);
#pragma unroll
        for (int k = 0; k < 8; ++k) p16[k] = scale * fmaf(v[k].x, v[k].x, v[k].y * v[k].y);
    }
    F512_FENCE();
}

template <int NI, int NC, int NSTAGE, int DTYPE, int WAVES, bool RAGGED>
__global__ __launch_bounds__(64 * WAVES, (WAVES > 8 ? 3 : 2)) void mfcc1536_kernel(F1536Params P, BatchGeom bg,
                                                                 const void* __restrict__ wave,
                                                                 float* __restrict__ out, int64_t ld_out) {
    extern __shared__ __attribute__((aligned(256))) float smem_f[];
    float* const smem = smem_f;
    const int tid = threadIdx.x;
    // Prologue as in kernels_fast512.h: the 33 KB of tables are requested FIRST (inline asm: the compiler would sink the
    // loads below the touches), then one dword per 16-byte vector of the wave's first group; the copy to LDS and the
    // barrier wait for the tables only, the touches stay in flight, and every wave waits for its own first fetch.
    constexpr int TABV = 5;                                    // 5 x 2048 floats cover 40 KB of tables
    typedef float f1536_v4 __attribute__((ext_vector_type(4)));
    f1536_v4 tabv_[TABV];
#pragma unroll
    for (int k = 0; k < TABV; ++k) {
        const int i = tid * 4 + 64 * WAVES * 4 * k;
        const float* src = P.tables + (i < P.tab_floats ? i : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(tabv_[k]) : "v"(src) : "memory");
    }
    if constexpr (!RAGGED) {
        float warm_[NSTAGE];
        int G0 = (int)blockIdx.x * WAVES + (tid >> 6);
        G0 = G0 < (int)P.total_groups ? G0 : (int)P.total_groups - 1;
        const int64_t e = (int64_t)(G0 / (int)P.groups_per_utt) * bg.uniform_samples +
                          (int64_t)(G0 % (int)P.groups_per_utt) * 4 * P.S;
        const int64_t lim = (int64_t)bg.n_utt * bg.uniform_samples - 4;
#pragma unroll
        for (int r = 0; r < NSTAGE; ++r) {
            int64_t idx = e + 4 * (tid & 63) + 256 * r;
            idx = idx < lim ? idx : lim;
            warm_[r] = dsp_load_sample<DTYPE>(wave, idx);
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTAGE) : "memory");   // the tables; the touches stay in flight
#pragma unroll
        for (int k = 0; k < TABV; ++k) {
            const int i = tid * 4 + 64 * WAVES * 4 * k;
            if (i < P.tab_floats) *reinterpret_cast<f1536_v4*>(smem + i) = tabv_[k];
        }
        for (int i = tid * 4 + 64 * WAVES * 4 * TABV; i < P.tab_floats; i += 64 * WAVES * 4)
            *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(P.tables + i);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NSTAGE; ++r) asm volatile("" :: "v"(warm_[r]));   // not used: keeps the touches alive
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < TABV; ++k) {
            const int i = tid * 4 + 64 * WAVES * 4 * k;
            if (i < P.tab_floats) *reinterpret_cast<f1536_v4*>(smem + i) = tabv_[k];
        }
        for (int i = tid * 4 + 64 * WAVES * 4 * TABV; i < P.tab_floats; i += 64 * WAVES * 4)
            *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(P.tables + i);
        __syncthreads();
    }
    const float* s_r3 = smem;   // radix-3 step: one row of F1536_R3_ROW floats per column c (window + W1536 twiddles, see host side)
    const float2* s_tw = reinterpret_cast<const float2*>(smem + P.off_tw);
    const float* s_dct = smem + P.off_dct;
    const float* s_melw = smem + P.off_melw;

    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* wbuf = smem + P.tab_floats + wid * P.wave_floats;
    const int total_groups = RAGGED ? P.group_off[bg.n_utt] : (int)P.total_groups;
    const int gstride = (int)gridDim.x * WAVES;

    // full rounds deal 8 consecutive groups to the 8 waves of a workgroup; the partial last round is dealt
    // wave-major (wave 0 of every workgroup first) so that no SIMD carries two groups above the average
    const int nfull = total_groups / gstride;
    for (int rnd = 0; rnd <= nfull; ++rnd) {
        int G;
        if (rnd < nfull) {
            G = __builtin_amdgcn_readfirstlane(rnd * gstride + (int)blockIdx.x * WAVES + wid);
        } else {
            G = __builtin_amdgcn_readfirstlane(nfull * gstride + (int)blockIdx.x + (int)gridDim.x * wid);
            if (G >= total_groups) break;
        }
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));  // keep lane-derived addresses out of loop-invariant registers
        const int f = lane >> 4, c = lane & 15;
        const F512Group grp = f1536_locate<RAGGED>(P, bg, G);
        const int t0 = grp.t0, T = grp.T, nsamp = grp.nsamp;
        const int base = t0 * P.S;
        const int64_t g0 = grp.s0 + base;
        // groups entirely inside their utterance stage without per-vector bookkeeping (kernels_fast512.h)
        const bool fast_stage = base + NSTAGE * 256 <= nsamp;
        const int d = (RAGGED && !fast_stage) ? (int)(g0 & 3) : 0;
        // Unit-variance statistics of segments read in place (bg.stats, as in kernels_fast512.h): every group adds the sums
        // of its own share [base, base + 4 S) of the utterance -- the last group everything up to the end -- of (x - x0)
        // and (x - x0)^2, x0 = the utterance's first sample.
        float st_s = 0.f, st_q = 0.f, st_ref = 0.f;
        int st_lim = 0;
        bool do_stats = false;
        if constexpr (RAGGED) {
            do_stats = bg.stats != nullptr && nsamp > 0;
            if (do_stats) {
                st_ref = dsp_load_sample<DTYPE>(wave, grp.s0);
                st_lim = t0 + 4 >= T ? nsamp : base + 4 * P.S;
            }
        }

        // ---- stage 3 S + 1536 (+ d) samples: 16 B loads, pre-emphasis, zero fill ----
        if (fast_stage) {
            F512Raw<DTYPE> raw[NSTAGE];
            const int64_t e0 = g0 + 4 * lane;
#pragma unroll
            for (int r = 0; r < NSTAGE; ++r)
                raw[r] = RAGGED ? f512_load_raw_unaligned<DTYPE>(wave, e0 + 256 * r) : f512_load_raw<DTYPE>(wave, e0 + 256 * r);
            float left = base > 0 ? dsp_load_sample<DTYPE>(wave, g0 - 1) : 0.f;
            const int span_vec = P.span_vec;
#pragma unroll
            for (int r = 0; r < NSTAGE; ++r) {
                float x[4];
                f512_unpack<DTYPE>(raw[r], x);
                const float prev = f512_shift_in(x[3], left);
                left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[3]), 63));
                float4 y;
                y.x = fmaf(-P.preemph, prev, x[0]);
                y.y = fmaf(-P.preemph, x[0], x[1]);
                y.z = fmaf(-P.preemph, x[1], x[2]);
                y.w = fmaf(-P.preemph, x[2], x[3]);
                if (lane + 64 * r < span_vec) *reinterpret_cast<float4*>(wbuf + 4 * lane + 256 * r) = y;
                if constexpr (RAGGED) {
                    if (do_stats) {
                        const int round_in = st_lim - (base + 256 * r);    // samples of this round inside the share (wave-uniform)
                        if (round_in >= 256) {
                            const float d0 = x[0] - st_ref, d1 = x[1] - st_ref, d2 = x[2] - st_ref, d3 = x[3] - st_ref;
                            st_s += (d0 + d1) + (d2 + d3);
                            st_q = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, st_q))));
                        } else if (round_in > 0) {
                            const int left_in_share = round_in - 4 * lane;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float dl = k < left_in_share ? x[k] - st_ref : 0.f;
                                st_s += dl;
                                st_q = fmaf(dl, dl, st_q);
                            }
                        }
                    }
                }
            }
        } else {
            const int64_t a0 = g0 - d;
            const int span_vec = RAGGED ? P.span_vec + 1 : P.span_vec;
            F512Raw<DTYPE> raw[NSTAGE];
#pragma unroll
            for (int r = 0; r < NSTAGE; ++r) {
                const int v = lane + 64 * r;
                const int rel = base - d + 4 * v;
                const bool touch = v < span_vec && rel + 3 >= 0 && rel < nsamp;
                raw[r] = f512_load_raw<DTYPE>(wave, touch ? a0 + 4 * v : 0);
            }
            float left = (base - d > 0) ? dsp_load_sample<DTYPE>(wave, a0 - 1) : 0.f;
#pragma unroll
            for (int r = 0; r < NSTAGE; ++r) {
                const int v = lane + 64 * r;
                const int rel = base - d + 4 * v;
                float x[4];
                f512_unpack<DTYPE>(raw[r], x);
                const float prev = f512_shift_in(x[3], left);
                left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[3]), 63));
                float4 y;
                y.x = fmaf(-P.preemph, prev, x[0]);
                y.y = fmaf(-P.preemph, x[0], x[1]);
                y.z = fmaf(-P.preemph, x[1], x[2]);
                y.w = fmaf(-P.preemph, x[2], x[3]);
                if constexpr (RAGGED) {
                    if (do_stats) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const bool in = v < span_vec && rel + k >= base && rel + k < st_lim;
                            const float dl = in ? x[k] - st_ref : 0.f;
                            st_s += dl;
                            st_q = fmaf(dl, dl, st_q);
                        }
                    }
                    if (rel + 0 == 0) y.x = x[0];
                    if (rel + 1 == 0) y.y = x[1];
                    if (rel + 2 == 0) y.z = x[2];
                    if (rel + 3 == 0) y.w = x[3];
                    if (rel + 0 < 0 || rel + 0 >= nsamp) y.x = 0.f;
                    if (rel + 1 < 0 || rel + 1 >= nsamp) y.y = 0.f;
                    if (rel + 2 < 0 || rel + 2 >= nsamp) y.z = 0.f;
                    if (rel + 3 < 0 || rel + 3 >= nsamp) y.w = 0.f;
                } else {
                    const uint32_t m = (v < span_vec && rel < nsamp) ? 0xffffffffu : 0u;
                    y.x = __uint_as_float(__float_as_uint(y.x) & m);
                    y.y = __uint_as_float(__float_as_uint(y.y) & m);
                    y.z = __uint_as_float(__float_as_uint(y.z) & m);
                    y.w = __uint_as_float(__float_as_uint(y.w) & m);
                }
                if (v < span_vec) *reinterpret_cast<float4*>(wbuf + 4 * v) = y;
            }
        }
        if constexpr (RAGGED) {
            if (do_stats) {
                float ws = st_s, wq = st_q;
                ws += dpp_f32<0xB1>(ws);  wq += dpp_f32<0xB1>(wq);     // quad_perm [1,0,3,2]
                ws += dpp_f32<0x4E>(ws);  wq += dpp_f32<0x4E>(wq);     // quad_perm [2,3,0,1]
                ws += dpp_f32<0x141>(ws); wq += dpp_f32<0x141>(wq);    // row_half_mirror
                ws += dpp_f32<0x140>(ws); wq += dpp_f32<0x140>(wq);    // row_mirror
                // the four row sums meet in fp64 in scalar registers: ONE pair of atomics per group instead of four
                double ds = 0.0, dq = 0.0;
#pragma unroll
                for (int rw = 0; rw < 64; rw += 16) {
                    ds += (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ws), rw));
                    dq += (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(wq), rw));
                }
                if (lane == 0) {
                    unsafeAtomicAdd(bg.stats + 2 * grp.utt, ds);
                    unsafeAtomicAdd(bg.stats + 2 * grp.utt + 1, dq);
                }
            }
        }
        F512_FENCE();

        // ---- radix-3 step: y_0 (real) and y_1 (complex) of column c, rows n1 = 0..31 ----
        float y0[32];
        cpx y1[32];
        {
            const float* xp = wbuf + d + f * P.S + c;
            // the column's window values and twiddles come as 40 conflict-free b128 reads of its own table row (they were
            // 96 + 32 narrow reads at a stride of 16 floats): per four rows n1, [w(m) x4 | w(m + 512) x4 | w(m + 1024) x4 |
            // W1536^m x4 (re, im)], m = 16 n1 + c
            const float4* rp = reinterpret_cast<const float4*>(s_r3 + c * F1536_R3_ROW);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 wa = rp[5 * q], wb = rp[5 * q + 1], we = rp[5 * q + 2], t0 = rp[5 * q + 3], t1 = rp[5 * q + 4];
                const float wav[4] = {wa.x, wa.y, wa.z, wa.w}, wbv[4] = {wb.x, wb.y, wb.z, wb.w}, wev[4] = {we.x, we.y, we.z, we.w};
                const float wx[4] = {t0.x, t0.z, t1.x, t1.z}, wy[4] = {t0.y, t0.w, t1.y, t1.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n1 = 4 * q + k;
                    const float a = xp[16 * n1] * wav[k];
                    const float b = xp[512 + 16 * n1] * wbv[k];
                    const float e = xp[1024 + 16 * n1] * wev[k];
                    const float sbe = b + e;
                    y0[n1] = a + sbe;
                    const float tr = fmaf(-0.5f, sbe, a);                     // Re(a + W3 b + W3^2 e)
                    const float ti = 0.86602540378443864676f * (e - b);        // Im(...)
                    y1[n1] = {fmaf(tr, wx[k], -ti * wy[k]), fmaf(tr, wy[k], ti * wx[k])};   // times W1536^(16 n1 + c)
                }
            }
        }
        F512_FENCE();

        constexpr float SC = 1.0f / 1536.0f;
        float* xb = wbuf + f * F1536_XSTRIDE;
        float p1a[16], p1b[16], p0a[16], p016[8];
        f1536_cfft512(y1, xb, s_tw, c, SC, p1a, p1b);
        F512_FENCE();
        f1536_rfft512(y0, xb, s_tw, c, 0.25f * SC, p0a, p016);
        F512_FENCE();

        // ---- power spectrum -> LDS row of this frame (every bin written exactly once) ----
        float* ps = wbuf + f * F1536_PS_STRIDE;
        float energy = 0.f;
        {
            float* q = ps + 3 * c;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                q[96 * k] = p0a[k];                    // bin 3 (c + 32 k)
                q[1 + 96 * k] = p1a[k];                // bin 3 (c + 32 k) + 1
                q[49 + 96 * k] = p1b[k];               // bin 3 (16 + c + 32 k) + 1
                energy += p0a[k] + (p1a[k] + p1b[k]);
            }
            float* m = ps - 3 * c;
#pragma unroll
            for (int k = 8; k < 16; ++k) {
                m[1535 - 96 * k] = p1a[k];             // bin 1536 - (3 (c + 32 k) + 1)
                m[1487 - 96 * k] = p1b[k];             // bin 1536 - (3 (16 + c + 32 k) + 1)
                energy += p1a[k] + p1b[k];
            }
            if (c == 0) {
                // lane 0: row 0 gives bins 96 k (k <= 8; its k >= 9 are duplicates), row 16 bins 48 + 96 k
                ps[768] = p0a[8];
                energy += p0a[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    ps[48 + 96 * k] = p016[k];
                    energy += p016[k];
                }
            } else {
                // rows 1..15: outputs k >= 8 are the mirror bins 3 (512 - c - 32 k) = residues 17..31
#pragma unroll
                for (int k = 8; k < 16; ++k) {
                    m[1536 - 96 * k] = p0a[k];
                    energy += p0a[k];
                }
                if (c < 8) ps[768 + c] = 0.f;          // row padding read by zero-weight mel taps
            }
        }
        energy = frame16_allreduce(energy);
        if (energy == 0.f) energy = DSP_EPS_F32;
        F512_FENCE();

        // ---- sparse mel triangles.  The filters are cut into 64 slots of at most 8 * mel_blocks taps
        //      (wide filters into several); lane s owns slot s for ALL four frames, so a weight is
        //      read from LDS once per four frames and every lane runs the same number of taps. ----
        float* part = wbuf + P.off_part;
        {
            const float4* wrow = reinterpret_cast<const float4*>(s_melw + lane * P.melw_row);
            const float* pbase = wbuf + __float_as_int(smem[P.off_mels + lane]);
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
            const int nb = P.mel_blocks;
            for (int b = 0; b < nb; ++b) {
                const float4 w0 = wrow[2 * b], w1 = wrow[2 * b + 1];
                float4 q0[4], q1[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4* pb = reinterpret_cast<const float4*>(pbase + g * F1536_PS_STRIDE);
                    q0[g] = pb[2 * b];
                    q1[g] = pb[2 * b + 1];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    a0[g] = fmaf(w0.x, q0[g].x, a0[g]);
                    a1[g] = fmaf(w0.y, q0[g].y, a1[g]);
                    a0[g] = fmaf(w0.z, q0[g].z, a0[g]);
                    a1[g] = fmaf(w0.w, q0[g].w, a1[g]);
                    a0[g] = fmaf(w1.x, q1[g].x, a0[g]);
                    a1[g] = fmaf(w1.y, q1[g].y, a1[g]);
                    a0[g] = fmaf(w1.z, q1[g].z, a0[g]);
                    a1[g] = fmaf(w1.w, q1[g].w, a1[g]);
                }
            }
            F512_FENCE();
#pragma unroll
            for (int g = 0; g < 4; ++g) part[g * F1536_PART_STRIDE + lane] = a0[g] + a1[g];
            if (lane < 4) part[lane * F1536_PART_STRIDE + F1536_SLOTS] = 0.f;   // the "no piece" slot
        }
        F512_FENCE();
        // gather the pieces of this lane's filters (c + 16 i) of its own frame, log
        float lm[NI];
        {
            const float* pf = part + f * F1536_PART_STRIDE;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int4* px = reinterpret_cast<const int4*>(smem + P.off_pidx + (i * 16 + c) * F1536_PIECES);
                const int4 x0 = px[0], x1 = px[1];
                float acc = (pf[x0.x] + pf[x0.y]) + (pf[x0.z] + pf[x0.w]);
                acc += (pf[x1.x] + pf[x1.y]) + (pf[x1.z] + pf[x1.w]);
                if (acc == 0.f) acc = DSP_EPS_F32;
                lm[i] = __logf(acc);
            }
        }
        F512_FENCE();

        // ---- DCT-II * lifter partial sums, all-reduce over the frame's 16 lanes ----
        float cep[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) cep[k] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4* dr = reinterpret_cast<const float4*>(s_dct + (i * 16 + c) * 20);
#pragma unroll
            for (int q = 0; q < (NC + 3) / 4; ++q) {
                const float4 dv = dr[q];
                if (4 * q + 0 < NC) cep[4 * q + 0] = fmaf(dv.x, lm[i], cep[4 * q + 0]);
                if (4 * q + 1 < NC) cep[4 * q + 1] = fmaf(dv.y, lm[i], cep[4 * q + 1]);
                if (4 * q + 2 < NC) cep[4 * q + 2] = fmaf(dv.z, lm[i], cep[4 * q + 2]);
                if (4 * q + 3 < NC) cep[4 * q + 3] = fmaf(dv.w, lm[i], cep[4 * q + 3]);
            }
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) cep[k] = frame16_allreduce(cep[k]);
        if (P.append_energy) cep[0] = __logf(energy);

        float v0 = cep[0];
#pragma unroll
        for (int k = 1; k < NC; ++k)
            if (c == k) v0 = cep[k];
        const int t = t0 + f;
        if (t < T && c < P.C) out[(grp.row0 + t) * ld_out + c] = v0;
        F512_FENCE();
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline bool fast1536_shape_ok(const dsp_plan_desc* d) {
    return d->nfft == 1536 && d->frame_len <= 1536 && d->frame_step >= 1 &&
           (3 * d->frame_step + 1536 + 3) / 4 + 1 <= 16 * 64 && d->nfilt >= 1 &&
           d->nfilt <= 16 * F1536_MAX_NI && d->numcep >= 1 && d->numcep <= 16;
}

// (Rounds 1-2 dealt the mel slots to lanes by a local search over a bank-conflict model of ds_read_b128 -- four groups
// of 16 lanes, 64 banks: 11 -> 8 modelled cycles per read.  Round 3 measured it away: with the search and with slots
// dealt in order the kernel shows the SAME SQ_LDS_BANK_CONFLICT (6 963 200 per dispatch, to the unit), the same
// SQ_LDS_IDX_ACTIVE and the same time (profiles/r3_f1536_conflicts.txt).  The counter -- and the clock -- do not see
// what that model predicts for these reads, so the slots are dealt in order and the search is gone.)
static inline int fast1536_plan_init(dsp_plan* p, const dsp_plan_desc* d, const int32_t* mel_off) {
    p->d_fast1536 = nullptr;
    if (!fast1536_shape_ok(d)) return DSP_OK;
    const int M = d->nfilt, C = d->numcep, L = d->frame_len;
    const int ni_real = (M + 15) / 16;
    const int span_vec = (3 * d->frame_step + 1536 + 3) / 4;
    const int nstage = (span_vec + 1 + 63) / 64;
    int variant, NI;
    if (ni_real <= 2 && C <= 13 && nstage <= 12) { variant = 0; NI = 2; }
    else { variant = 1; NI = F1536_MAX_NI; }
    Fast1536Plan* fp = new Fast1536Plan();
    memset(fp, 0, sizeof(*fp));

    std::vector<float> win(1536, 0.f);
    for (int n = 0; n < L; ++n) win[n] = d->h_window[n];
    std::vector<float> w3(2 * 512), tw(2 * 31 * 16);
    for (int m = 0; m < 512; ++m) {
        const double a = -2.0 * M_PI * (double)m / 1536.0;
        w3[2 * m] = (float)cos(a);
        w3[2 * m + 1] = (float)sin(a);
    }
    for (int k1 = 1; k1 < 32; ++k1)
        for (int c = 0; c < 16; ++c) {
            const double a = -2.0 * M_PI * (double)(c * k1) / 512.0;
            tw[2 * ((k1 - 1) * 16 + c)] = (float)cos(a);
            tw[2 * ((k1 - 1) * 16 + c) + 1] = (float)sin(a);
        }
    std::vector<float> dct((size_t)NI * 16 * 20, 0.f);
    for (int i = 0; i < NI; ++i)
        for (int c = 0; c < 16; ++c) {
            const int j = c + 16 * i;
            if (j >= M) continue;
            for (int k = 0; k < C; ++k) dct[((size_t)i * 16 + c) * 20 + k] = d->h_dct[(size_t)k * M + j];
        }
    // mel: cut the filters into at most 64 slots of `cap` taps (cap a multiple of 8, as small as
    // possible).  A slot is read from a bin that is a multiple of 4 (leading zero weights absorb the
    // misalignment) and never runs past the end of the 776-float spectrum row.
    bool ok = true;
    int cap = 0;
    for (int tryc = 8; tryc <= F1536_PS_STRIDE; tryc += 8) {
        int n = 0, worst = 0;
        for (int j = 0; j < M; ++j) {
            if (d->h_mel_count[j] <= 0) continue;
            const int span = (d->h_mel_start[j] & 3) + d->h_mel_count[j];
            const int pcs = (span + tryc - 1) / tryc;
            n += pcs;
            if (pcs > worst) worst = pcs;
        }
        if (n <= F1536_SLOTS && worst <= F1536_PIECES) { cap = tryc; break; }
    }
    if (cap == 0) ok = false;
    int melw_row = cap > 0 ? cap : 8;
    if (((melw_row / 4) & 1) == 0) melw_row += 4;
    std::vector<float> melw((size_t)F1536_SLOTS * melw_row, 0.f), mels(F1536_SLOTS, 0.f);
    std::vector<int32_t> pidx((size_t)NI * 16 * F1536_PIECES, F1536_SLOTS);   // default: the always-zero slot
    if (ok) {
        // pass 1: the slots in filter order; pass 2: deal them to lanes (f1536_assign_slots); pass 3: tables
        int start[64], lane_of_slot[64], n_slots = 0;
        for (int j = 0; j < M; ++j) {
            const int ms = d->h_mel_start[j], cnt = d->h_mel_count[j];
            if (cnt <= 0) continue;
            const int a0 = ms & ~3;
            const int pcs = ((ms - a0) + cnt + cap - 1) / cap;
            for (int pc = 0; pc < pcs; ++pc, ++n_slots) {
                int st0 = a0 + pc * cap;
                if (st0 + cap > F1536_PS_STRIDE) st0 = F1536_PS_STRIDE - cap;   // stays a multiple of 4
                start[n_slots] = st0;
            }
        }
        for (int sl = n_slots; sl < 64; ++sl) start[sl] = 0;
        for (int sl = 0; sl < 64; ++sl) lane_of_slot[sl] = sl;
        int slot = 0;
        for (int j = 0; j < M; ++j) {
            const int ms = d->h_mel_start[j], cnt = d->h_mel_count[j];
            if (cnt <= 0) continue;
            const int a0 = ms & ~3;
            const int pcs = ((ms - a0) + cnt + cap - 1) / cap;
            for (int pc = 0; pc < pcs; ++pc, ++slot) {
                const int lo = a0 + pc * cap, hi = lo + cap;            // bins this piece covers
                const int32_t st0 = start[slot];
                const int ln = lane_of_slot[slot];
                for (int bin = (lo > ms ? lo : ms); bin < hi && bin < ms + cnt; ++bin)
                    melw[(size_t)ln * melw_row + (bin - st0)] = d->h_mel_weights[mel_off[j] + (bin - ms)];
                memcpy(&mels[ln], &st0, 4);
                pidx[((size_t)(j / 16) * 16 + (j % 16)) * F1536_PIECES + pc] = ln;
            }
        }
        for (int sl = n_slots; sl < 64; ++sl) {      // unused lanes: zero weights, any in-row start
            const int32_t st0 = start[sl];
            memcpy(&mels[lane_of_slot[sl]], &st0, 4);
        }
    }
    auto pad64 = [](size_t n) { return (n + 63) / 64 * 64; };
    // radix-3 table: row c (F1536_R3_ROW = 164 floats: 36 c mod 64 runs over the 16 bank quads, so the 16 columns of a b128
    // lane group never meet on a bank), 8 chunks of 20 floats for rows n1 = 4 q .. 4 q + 3
    std::vector<float> r3((size_t)16 * F1536_R3_ROW, 0.f);
    for (int c = 0; c < 16; ++c)
        for (int q = 0; q < 8; ++q)
            for (int k = 0; k < 4; ++k) {
                const int m = 16 * (4 * q + k) + c;
                float* row = r3.data() + (size_t)c * F1536_R3_ROW + 20 * q;
                row[k] = win[m];
                row[4 + k] = win[512 + m];
                row[8 + k] = win[1024 + m];
                row[12 + 2 * k] = w3[2 * m];
                row[13 + 2 * k] = w3[2 * m + 1];
            }
    const size_t o_w3 = 0, o_tw = pad64(r3.size()), o_dct = o_tw + tw.size();
    const size_t o_melw = pad64(o_dct + dct.size()), o_mels = o_melw + melw.size();
    const size_t o_pidx = pad64(o_mels + mels.size());
    const size_t total = pad64(o_pidx + pidx.size());
    size_t wave_floats = 4 * F1536_PS_STRIDE;
    if ((size_t)4 * (span_vec + 1) > wave_floats) wave_floats = (size_t)4 * (span_vec + 1);
    if ((size_t)4 * F1536_XSTRIDE > wave_floats) wave_floats = 4 * F1536_XSTRIDE;
    // partial sums reuse the first 65 floats of each spectrum row: a wave's LDS operations execute in
    // order, so every lane's mel reads are done before the first partial sum lands
    const size_t off_part = 0;
    wave_floats = pad64(wave_floats);
    if (!ok || (total + F1536_WAVES * wave_floats) * 4 > 160 * 1024) {  // does not fit one CU's LDS: generic kernel
        delete fp;
        return DSP_OK;
    }
    std::vector<float> blob(total, 0.f);
    memcpy(blob.data(), r3.data(), r3.size() * 4);
    memcpy(blob.data() + o_tw, tw.data(), tw.size() * 4);
    memcpy(blob.data() + o_dct, dct.data(), dct.size() * 4);
    memcpy(blob.data() + o_melw, melw.data(), melw.size() * 4);
    memcpy(blob.data() + o_mels, mels.data(), mels.size() * 4);
    memcpy(blob.data() + o_pidx, pidx.data(), pidx.size() * 4);
    if (dsp_table_alloc_copy(reinterpret_cast<void**>(&fp->d_tables), blob.data(), total * 4) != hipSuccess) {
        delete fp;
        return DSP_EHIP;
    }
    fp->P.tables = fp->d_tables;
    fp->P.tab_floats = (int32_t)total;
    fp->P.off_w3 = (int32_t)o_w3; fp->P.off_tw = (int32_t)o_tw; fp->P.off_dct = (int32_t)o_dct;
    fp->P.off_melw = (int32_t)o_melw; fp->P.off_mels = (int32_t)o_mels; fp->P.melw_row = melw_row;
    fp->P.off_pidx = (int32_t)o_pidx; fp->P.mel_blocks = cap / 8; fp->P.off_part = (int32_t)off_part;
    fp->P.L = L; fp->P.S = d->frame_step; fp->P.M = M; fp->P.C = C;
    fp->P.append_energy = d->append_energy ? 1 : 0;
    fp->P.preemph = d->preemph;
    fp->P.span_vec = span_vec;
    fp->P.wave_floats = (int32_t)wave_floats;
    fp->variant = variant;
    p->d_fast1536 = fp;
    return DSP_OK;
}

static inline void fast1536_plan_free(dsp_plan* p) {
    Fast1536Plan* fp = static_cast<Fast1536Plan*>(p->d_fast1536);
    if (!fp) return;
    dsp_table_free(fp->d_tables, p->dry_run);
    delete fp;
    p->d_fast1536 = nullptr;
}

static inline bool fast1536_applicable(const dsp_plan* p, const BatchGeom& bg, const void* d_wave, int dtype) {
    if (!p->d_fast1536) return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_wave);
    if ((a % (dtype == DSP_WAVE_I16 ? 8 : 16)) != 0) return false;
    if (bg.uniform_samples > 0) {
        if ((bg.uniform_samples % 4) != 0) return false;
        return bg.uniform_samples <= 0x3fffffff && ((bg.uniform_frames + 3) / 4) * bg.n_utt <= 0x3fffffff;
    }
    return bg.total_frames / 4 + bg.n_utt <= 0x3fffffff;
}

template <int NI, int NC, int NSTAGE, int DTYPE, bool RAGGED>
static int fast1536_launch_k(const F1536Params& P, const void* d_wave, const BatchGeom& bg, float* d_out,
                             int64_t ld_out, int64_t groups_bound, hipStream_t st) {
    const size_t lds = ((size_t)P.tab_floats + (size_t)F1536_WAVES * P.wave_floats) * sizeof(float);
    const int64_t cap = dsp_cu_count();  // one 8-wave workgroup per CU (LDS bound); the kernel deals the partial last round
    int64_t blocks = (groups_bound + F1536_WAVES - 1) / F1536_WAVES;
    if (blocks > cap) blocks = cap;
    auto k = mfcc1536_kernel<NI, NC, NSTAGE, DTYPE, F1536_WAVES, RAGGED>;
    static size_t granted[DSP_MAX_DEVICES] = {};
    if (dsp_ensure_dynamic_lds((const void*)k, lds, granted) != 0) return DSP_EHIP;
    k<<<(int)blocks, 64 * F1536_WAVES, lds, st>>>(P, bg, d_wave, d_out, ld_out);
    return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
}

template <int NI, int NC, int NSTAGE>
static int fast1536_launch_t(F1536Params P, const void* d_wave, int dtype, const BatchGeom& bg, float* d_out,
                             int64_t ld_out, hipStream_t st, const DspRaggedTables* pre = nullptr) {
    if (bg.uniform_samples > 0) {
        P.groups_per_utt = (bg.uniform_frames + 3) / 4;
        P.total_groups = P.groups_per_utt * bg.n_utt;
        if (dtype == DSP_WAVE_I16)
            return fast1536_launch_k<NI, NC, NSTAGE, DSP_WAVE_I16, false>(P, d_wave, bg, d_out, ld_out, P.total_groups, st);
        return fast1536_launch_k<NI, NC, NSTAGE, DSP_WAVE_F32, false>(P, d_wave, bg, d_out, ld_out, P.total_groups, st);
    }
    const int64_t bound = bg.total_frames / 4 + bg.n_utt;  // >= sum ceil(T_b / 4)
    DspWorkspace* w = nullptr;
    if (pre != nullptr && pre->shift == 2) {
        P.group_off = pre->group_off;
        P.group_utt = pre->group_utt;
    } else {
        const size_t ws_bytes = ((size_t)bg.n_utt + 1 + (size_t)bound) * sizeof(int32_t);
        w = dsp_workspace_pool().acquire(ws_bytes, st);
        if (!w) return DSP_EHIP;
        int32_t* group_off = static_cast<int32_t*>(w->ptr);
        int32_t* group_utt = group_off + bg.n_utt + 1;
        f512_build_group_tables(bg.frame_off, bg.n_utt, 2, group_off, group_utt, st);
        P.group_off = group_off;
        P.group_utt = group_utt;
    }
    int rc;
    if (dtype == DSP_WAVE_I16)
        rc = fast1536_launch_k<NI, NC, NSTAGE, DSP_WAVE_I16, true>(P, d_wave, bg, d_out, ld_out, bound, st);
    else
        rc = fast1536_launch_k<NI, NC, NSTAGE, DSP_WAVE_F32, true>(P, d_wave, bg, d_out, ld_out, bound, st);
    if (w != nullptr && dsp_workspace_pool().release(w, st) != 0 && rc == DSP_OK) rc = DSP_EHIP;
    return rc;
}

static inline int fast1536_launch(const dsp_plan* p, const void* d_wave, int dtype, const BatchGeom& bg,
                                  float* d_out, int64_t ld_out, hipStream_t st, const DspRaggedTables* pre = nullptr) {
    const Fast1536Plan* fp = static_cast<const Fast1536Plan*>(p->d_fast1536);
    if (fp->variant == 0) return fast1536_launch_t<2, 13, 12>(fp->P, d_wave, dtype, bg, d_out, ld_out, st, pre);
    return fast1536_launch_t<4, 16, 16>(fp->P, d_wave, dtype, bg, d_out, ld_out, st, pre);
}
