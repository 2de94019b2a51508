// Tiled short-time amplitude / zero-crossing kernel for the endpointing path
// (endpoint.get_amplitude, endpoint.py:109-126; endpoint.get_zcr, endpoint.py:182-198; frames from
// sigproc.to_frames, sigproc.py:11-19).
//
// One wavefront = FR consecutive frames of one utterance, 64 / FR lanes per frame.  The
// (FR - 1) S + L samples the frames cover are read from HBM once (aligned 16 B per lane, the staging
// scheme of kernels_fast512.h without the pre-emphasis), converted to fp32 (exact for int16) and
// parked in LDS; every lane then walks its frame with plain b32 reads.  Sums are order independent,
// so frame f starts its walk rot_f elements into the frame, rot_f chosen such that the FR frames
// of a wave instruction hit disjoint LDS banks whatever the hop is.  fp64 accumulation.
#pragma once

#include <type_traits>

#include "kernels_fast512.h"

#define VAD_NSTAGE 12   // 12 x 64 lanes x 4 samples staged per wave at most
#define VAD_WAVES 4

struct VadParams {
    int32_t L, S, use_sq;
    int32_t span_vec;       // ceil(((FR - 1) S + L) / 4)
    int32_t wave_floats;    // per-wave LDS region (floats)
    int32_t off_a4, off_e;  // vad_vec_kernel: per-vector |x| sums / sign-change bits behind the samples
    int64_t groups_per_utt, total_groups;   // uniform batches
    const int32_t* group_off;               // ragged: [B+1] prefix of ceil(T_b / FR)
    const int32_t* group_utt;
};

// ACC: 0 = fp32 partial sums of |x| (exact for int16 input), 1 = fp64 sums of |x|, 2 = fp64 sums of x^2
template <int DTYPE, int FR, bool RAGGED, int ACC>
__global__ __launch_bounds__(64 * VAD_WAVES) void vad_tile_kernel(VadParams P, BatchGeom bg,
                                                                  const void* __restrict__ wave,
                                                                  double* __restrict__ amp_sum,
                                                                  int32_t* __restrict__ zcr) {
    constexpr int LPF = 64 / FR;   // lanes per frame
    constexpr int SHIFT = FR == 16 ? 4 : (FR == 8 ? 3 : 2);
    static_assert(FR == 16 || FR == 8 || FR == 4, "");
    extern __shared__ __attribute__((aligned(256))) float smem_f[];
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    float* wbuf = smem_f + wid * P.wave_floats;
    const int total_groups = RAGGED ? P.group_off[bg.n_utt] : (int)P.total_groups;
    const int gstride = (int)gridDim.x * VAD_WAVES;
    const int f = lane / LPF, q = lane % LPF;
    const int L = P.L, S = P.S;
    const int Lr = (L + LPF - 1) / LPF * LPF;            // walk length, multiple of LPF
    const int rot = ((LPF * f - S * f) % 64 + 64) % 64;   // (S f + rot) % 64 == LPF f: disjoint banks

    for (int G = __builtin_amdgcn_readfirstlane((int)blockIdx.x * VAD_WAVES + wid); G < total_groups; G += gstride) {
        int utt, t0, T, nsamp;
        int64_t s0, row0;
        if constexpr (RAGGED) {
            utt = P.group_utt[G];
            t0 = (G - P.group_off[utt]) << SHIFT;
            s0 = bg.sample_off[utt];
            nsamp = (int)(bg.sample_off[utt + 1] - s0);
            row0 = bg.frame_off[utt];
            T = (int)(bg.frame_off[utt + 1] - row0);
        } else {
            const int gpu = (int)P.groups_per_utt;
            utt = G / gpu;
            t0 = (G - utt * gpu) << SHIFT;
            nsamp = (int)bg.uniform_samples;
            T = (int)bg.uniform_frames;
            s0 = (int64_t)utt * bg.uniform_samples;
            row0 = (int64_t)utt * bg.uniform_frames;
        }
        const int base = t0 * S;
        const int64_t g0 = s0 + base;
        const int d = RAGGED ? (int)(g0 & 3) : 0;
        {
            const int64_t a0 = g0 - d;
            const int span_vec = RAGGED ? P.span_vec + 1 : P.span_vec;
            F512Raw<DTYPE> raw[VAD_NSTAGE];
#pragma unroll
            for (int r = 0; r < VAD_NSTAGE; ++r) {
                const int v = lane + 64 * r;
                const int rel = base - d + 4 * v;
                const bool touch = v < span_vec && rel + 3 >= 0 && rel < nsamp;
                raw[r] = f512_load_raw<DTYPE>(wave, touch ? a0 + 4 * v : 0);
            }
#pragma unroll
            for (int r = 0; r < VAD_NSTAGE; ++r) {
                const int v = lane + 64 * r;
                const int rel = base - d + 4 * v;
                float x[4];
                f512_unpack<DTYPE>(raw[r], x);
                float4 y = make_float4(x[0], x[1], x[2], x[3]);
                if (rel + 0 < 0 || rel + 0 >= nsamp) y.x = 0.f;   // zero padding of the last frame (sigproc.py:84-87)
                if (rel + 1 < 0 || rel + 1 >= nsamp) y.y = 0.f;
                if (rel + 2 < 0 || rel + 2 >= nsamp) y.z = 0.f;
                if (rel + 3 < 0 || rel + 3 >= nsamp) y.w = 0.f;
                if (v < span_vec) *reinterpret_cast<float4*>(wbuf + 4 * v) = y;
            }
        }
        F512_FENCE();

        const float* fr = wbuf + d + f * S;
        double acc = 0.0;
        int32_t cnt = 0;
        float accf = 0.f;
        const int last = L - 1;
        // eight steps at a time: all sixteen LDS reads are issued before the first use; no branches
        // (addresses are clamped, contributions masked; element L - 1 has no right neighbour)
        for (int j0 = 0; j0 < Lr; j0 += 8 * LPF) {
            float av[8], bv[8];
            int iv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u * LPF;
                int i = q + j + rot;
                if (i >= Lr) i -= Lr;
                if (j >= Lr) i = L;                       // step past the walk: masked below
                iv[u] = i;
                av[u] = fr[min(i, last)];
                bv[u] = fr[min(i + 1, last)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float a = iv[u] > last ? 0.f : av[u];
                const float b = bv[u];
                const uint32_t differ = (__float_as_uint(a) ^ __float_as_uint(b)) >> 31;
                const uint32_t pair = differ & (uint32_t)(a != 0.f) & (uint32_t)(b != 0.f) & (uint32_t)(iv[u] < last);
                cnt += (int32_t)pair;
                if constexpr (ACC == 0) accf += fabsf(a);
                else if constexpr (ACC == 1) acc += (double)fabsf(a);
                else acc += (double)a * (double)a;
            }
        }
        if constexpr (ACC == 0) acc = (double)accf;
#pragma unroll
        for (int o = LPF / 2; o > 0; o >>= 1) {
            acc += __shfl_xor(acc, o, 64);
            cnt += __shfl_xor(cnt, o, 64);
        }
        const int t = t0 + f;
        if (q == 0 && t < T) {
            amp_sum[row0 + t] = acc;
            zcr[row0 + t] = cnt;
        }
        F512_FENCE();
    }
}


// int16 input, sum |x|: every sample is visited ONCE.  While a vector of four samples is being staged
// its lane also forms a4 = |x0|+|x1|+|x2|+|x3| (exact in fp32) and four sign-change bits e_i for the
// pairs (i-1, i); a frame is then the sum of the ~L/4 vector totals strictly inside it plus two edge
// vectors resolved from the staged samples and a bit mask.  Same results as the walk above, ~3.5x
// fewer instructions.
template <int FR, bool RAGGED>
__global__ __launch_bounds__(64 * VAD_WAVES) void vad_vec_kernel(VadParams P, BatchGeom bg,
                                                                 const void* __restrict__ wave,
                                                                 double* __restrict__ amp_sum,
                                                                 int32_t* __restrict__ zcr) {
    constexpr int LPF = 64 / FR;
    constexpr int SHIFT = FR == 16 ? 4 : (FR == 8 ? 3 : 2);
    extern __shared__ __attribute__((aligned(256))) float smem_f[];
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    float* xs = smem_f + wid * P.wave_floats;
    float* a4s = xs + P.off_a4;
    int32_t* es = reinterpret_cast<int32_t*>(xs + P.off_e);
    const int total_groups = RAGGED ? P.group_off[bg.n_utt] : (int)P.total_groups;
    const int gstride = (int)gridDim.x * VAD_WAVES;
    const int f = lane / LPF, q = lane % LPF;
    const int L = P.L, S = P.S;

    for (int G = __builtin_amdgcn_readfirstlane((int)blockIdx.x * VAD_WAVES + wid); G < total_groups; G += gstride) {
        int utt, t0, T, nsamp;
        int64_t s0, row0;
        if constexpr (RAGGED) {
            utt = P.group_utt[G];
            t0 = (G - P.group_off[utt]) << SHIFT;
            s0 = bg.sample_off[utt];
            nsamp = (int)(bg.sample_off[utt + 1] - s0);
            row0 = bg.frame_off[utt];
            T = (int)(bg.frame_off[utt + 1] - row0);
        } else {
            const int gpu = (int)P.groups_per_utt;
            utt = G / gpu;
            t0 = (G - utt * gpu) << SHIFT;
            nsamp = (int)bg.uniform_samples;
            T = (int)bg.uniform_frames;
            s0 = (int64_t)utt * bg.uniform_samples;
            row0 = (int64_t)utt * bg.uniform_frames;
        }
        const int base = t0 * S;
        const int64_t g0 = s0 + base;
        const int d = RAGGED ? (int)(g0 & 3) : 0;
        {
            const int64_t a0 = g0 - d;
            const int span_vec = RAGGED ? P.span_vec + 1 : P.span_vec;
            F512Raw<DSP_WAVE_I16> raw[VAD_NSTAGE];
#pragma unroll
            for (int r = 0; r < VAD_NSTAGE; ++r) {
                const int v = lane + 64 * r;
                const int rel = base - d + 4 * v;
                const bool touch = v < span_vec && rel + 3 >= 0 && rel < nsamp;
                raw[r] = f512_load_raw<DSP_WAVE_I16>(wave, touch ? a0 + 4 * v : 0);
            }
            float left = 0.f;   // the pair (first staged sample - 1, first staged sample) is never inside a frame
#pragma unroll
            for (int r = 0; r < VAD_NSTAGE; ++r) {
                const int v = lane + 64 * r;
                const int rel = base - d + 4 * v;
                float x[4];
                f512_unpack<DSP_WAVE_I16>(raw[r], x);
                float4 y = make_float4(x[0], x[1], x[2], x[3]);
                if (rel + 0 < 0 || rel + 0 >= nsamp) y.x = 0.f;   // zero padding (sigproc.py:84-87)
                if (rel + 1 < 0 || rel + 1 >= nsamp) y.y = 0.f;
                if (rel + 2 < 0 || rel + 2 >= nsamp) y.z = 0.f;
                if (rel + 3 < 0 || rel + 3 >= nsamp) y.w = 0.f;
                const float prev = f512_shift_in(y.w, left);
                left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y.w), 63));
                // int16 products cannot underflow: x[i-1] * x[i] < 0 is the exact sign-pair test
                const uint32_t b0 = prev * y.x < 0.f, b1 = y.x * y.y < 0.f, b2 = y.y * y.z < 0.f, b3 = y.z * y.w < 0.f;
                const uint32_t bits = b0 | (b1 << 1) | (b2 << 2) | (b3 << 3);
                if (v < span_vec) {
                    *reinterpret_cast<float4*>(xs + 4 * v) = y;
                    a4s[v] = (fabsf(y.x) + fabsf(y.y)) + (fabsf(y.z) + fabsf(y.w));
                    es[v] = (int32_t)(bits | ((b0 + b1 + b2 + b3) << 8));
                }
            }
        }
        F512_FENCE();

        // frame f = samples [s, e) of the LDS image; vectors vL and vR hold its two ends
        const int s = d + f * S, e = s + L;
        const int vL = s >> 2, vR = e >> 2;
        float accf = 0.f;
        int32_t cnt = 0;
        for (int v0 = vL + 1 + q; v0 < vR; v0 += 8 * LPF) {
            float av[8];
            int32_t ev[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = min(v0 + u * LPF, vR - 1);
                av[u] = a4s[v];
                ev[u] = es[v];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = v0 + u * LPF < vR;
                accf += in ? av[u] : 0.f;
                cnt += in ? (ev[u] >> 8) : 0;
            }
        }
        if (q == 0) {
            // left edge: offsets k..3 of vector vL, pairs (i-1, i) only for i >= s + 1
            const int k = s & 3;
            const float4 xl = *reinterpret_cast<const float4*>(xs + 4 * vL);
            accf += (k <= 0 ? fabsf(xl.x) : 0.f) + (k <= 1 ? fabsf(xl.y) : 0.f) + (k <= 2 ? fabsf(xl.z) : 0.f) + fabsf(xl.w);
            cnt += __builtin_popcount((uint32_t)es[vL] & (0xFu << (k + 1)) & 0xFu);
            // right edge: offsets 0..m-1 of vector vR (absent when the frame ends on a vector boundary)
            const int m = e & 3;
            const float4 xr = *reinterpret_cast<const float4*>(xs + 4 * vR);
            accf += (m > 0 ? fabsf(xr.x) : 0.f) + (m > 1 ? fabsf(xr.y) : 0.f) + (m > 2 ? fabsf(xr.z) : 0.f);
            cnt += __builtin_popcount((uint32_t)es[vR] & ((1u << m) - 1u));
        }
        double acc = (double)accf;    // lane partials are exact integers below 2^24
#pragma unroll
        for (int o = LPF / 2; o > 0; o >>= 1) {
            acc += __shfl_xor(acc, o, 64);
            cnt += __shfl_xor(cnt, o, 64);
        }
        const int t = t0 + f;
        if (q == 0 && t < T) {
            amp_sum[row0 + t] = acc;
            zcr[row0 + t] = cnt;
        }
        F512_FENCE();
    }
}

// Vector-aligned frames (L % 4 == 0 and S % 4 == 0 -- every framing the endpoint path uses at 16 / 44.1 /
// 48 kHz): the wave loads its samples from the group's FIRST sample as it stands (vector loads at the
// element's own alignment), so every frame is a whole number of 4-sample vectors and nothing but two
// numbers per vector has to be staged: a4 = sum of |x| (or x^2) over the vector and the sign-change bits
// of its four (i - 1, i) pairs.  Every sample is visited once, for int16 AND fp32 input (SURVEY 8a-12/13:
// endpoint.py:109-126, 182-198); 8 bytes (12 with fp64 sums) of LDS per vector instead of 24, so three
// times as many waves fit a CU as with vad_vec_kernel.  A frame = its L / 4 vector totals; its count
// drops the first vector's pair (s - 1, s), which lies outside the frame.
//   sums: fp32 when exact (int16 input, |x|: a lane's partial stays below 2^24), fp64 otherwise.
template <int DTYPE, int FR, bool RAGGED, bool F32SUM>
__global__ __launch_bounds__(64 * VAD_WAVES) void vad_sum_kernel(VadParams P, BatchGeom bg,
                                                                 const void* __restrict__ wave,
                                                                 double* __restrict__ amp_sum,
                                                                 int32_t* __restrict__ zcr) {
    using acc_t = typename std::conditional<F32SUM, float, double>::type;
    constexpr int LPF = 64 / FR;
    constexpr int SHIFT = FR == 16 ? 4 : 2;
    extern __shared__ __attribute__((aligned(256))) float smem_f[];
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // per-wave region: nvec sums (acc_t) followed by nvec int32 bit words
    char* wb = reinterpret_cast<char*>(smem_f) + (size_t)wid * P.wave_floats * 4;
    acc_t* a4s = reinterpret_cast<acc_t*>(wb);
    int32_t* es = reinterpret_cast<int32_t*>(wb + (size_t)P.off_e * 4);
    const int total_groups = RAGGED ? P.group_off[bg.n_utt] : (int)P.total_groups;
    const int gstride = (int)gridDim.x * VAD_WAVES;
    const int f = lane / LPF, q = lane % LPF;
    const int vpf = P.L >> 2, vps = P.S >> 2;      // vectors per frame / per hop
    const int nvec = P.span_vec;                    // (FR - 1) S / 4 + L / 4

    for (int G = __builtin_amdgcn_readfirstlane((int)blockIdx.x * VAD_WAVES + wid); G < total_groups; G += gstride) {
        int utt, t0, T, nsamp;
        int64_t s0, row0;
        if constexpr (RAGGED) {
            utt = P.group_utt[G];
            t0 = (G - P.group_off[utt]) << SHIFT;
            s0 = bg.sample_off[utt];
            nsamp = (int)(bg.sample_off[utt + 1] - s0);
            row0 = bg.frame_off[utt];
            T = (int)(bg.frame_off[utt + 1] - row0);
        } else {
            const int gpu = (int)P.groups_per_utt;
            utt = G / gpu;
            t0 = (G - utt * gpu) << SHIFT;
            nsamp = (int)bg.uniform_samples;
            T = (int)bg.uniform_frames;
            s0 = (int64_t)utt * bg.uniform_samples;
            row0 = (int64_t)utt * bg.uniform_frames;
        }
        const int base = t0 * P.S;
        const int64_t g0 = s0 + base;
        // ---- one pass over the samples: all loads first, then sums and pair bits ----
        F512Raw<DTYPE> raw[VAD_NSTAGE];
#pragma unroll
        for (int r = 0; r < VAD_NSTAGE; ++r) {
            const int v = lane + 64 * r;
            const int rel = base + 4 * v;
            if (v < nvec && rel + 3 < nsamp) {
                raw[r] = f512_load_raw_unaligned<DTYPE>(wave, g0 + 4 * v);       // entirely inside the clip
            } else {                                                              // clip end: element by element
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = (v < nvec && rel + k < nsamp) ? dsp_load_sample<DTYPE>(wave, g0 + 4 * v + k) : 0.f;
                if constexpr (DTYPE == DSP_WAVE_I16) raw[r].v = make_short4((short)e[0], (short)e[1], (short)e[2], (short)e[3]);
                else raw[r].v = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        float left = 0.f;   // the pair (first staged sample - 1, first staged sample) is never inside a frame
#pragma unroll
        for (int r = 0; r < VAD_NSTAGE; ++r) {
            const int v = lane + 64 * r;
            float x[4];
            f512_unpack<DTYPE>(raw[r], x);
            const float prev = f512_shift_in(x[3], left);
            left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[3]), 63));
            uint32_t b0, b1, b2, b3;
            if constexpr (DTYPE == DSP_WAVE_I16) {
                // int16 products cannot underflow: x[i-1] * x[i] < 0 is the exact sign-pair test
                b0 = prev * x[0] < 0.f; b1 = x[0] * x[1] < 0.f; b2 = x[1] * x[2] < 0.f; b3 = x[2] * x[3] < 0.f;
            } else {
                // fp32: opposite sign bits and both non-zero (a product could underflow to 0)
                auto opp = [](float a, float b) -> uint32_t {
                    return ((__float_as_uint(a) ^ __float_as_uint(b)) >> 31) & (uint32_t)(a != 0.f) & (uint32_t)(b != 0.f);
                };
                b0 = opp(prev, x[0]); b1 = opp(x[0], x[1]); b2 = opp(x[1], x[2]); b3 = opp(x[2], x[3]);
            }
            acc_t a4;
            if (P.use_sq) {
                a4 = ((acc_t)x[0] * (acc_t)x[0] + (acc_t)x[1] * (acc_t)x[1]) + ((acc_t)x[2] * (acc_t)x[2] + (acc_t)x[3] * (acc_t)x[3]);
            } else {
                a4 = ((acc_t)fabsf(x[0]) + (acc_t)fabsf(x[1])) + ((acc_t)fabsf(x[2]) + (acc_t)fabsf(x[3]));
            }
            if (v < nvec) {
                a4s[v] = a4;
                es[v] = (int32_t)(b0 | ((b0 + b1 + b2 + b3) << 8));
            }
        }
        F512_FENCE();

        // ---- frame f = vectors [f vps, f vps + vpf); lane q takes every LPF-th of them ----
        const int v_lo = f * vps, v_hi = v_lo + vpf;
        acc_t acc = 0;
        int32_t cnt = 0;
        for (int v0 = v_lo + q; v0 < v_hi; v0 += 8 * LPF) {
            acc_t av[8];
            int32_t ev[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = min(v0 + u * LPF, v_hi - 1);
                av[u] = a4s[v];
                ev[u] = es[v];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = v0 + u * LPF < v_hi;
                acc += in ? av[u] : (acc_t)0;
                cnt += in ? (ev[u] >> 8) : 0;
            }
        }
        if (q == 0) cnt -= es[v_lo] & 1;        // the pair (s - 1, s) belongs to the previous sample
        double accd = (double)acc;               // fp32 lane partials are exact integers below 2^24
#pragma unroll
        for (int o = LPF / 2; o > 0; o >>= 1) {
            accd += __shfl_xor(accd, o, 64);
            cnt += __shfl_xor(cnt, o, 64);
        }
        const int t = t0 + f;
        if (q == 0 && t < T) {
            amp_sum[row0 + t] = accd;
            zcr[row0 + t] = cnt;
        }
        F512_FENCE();
    }
}

// int16 input, sum |x|, ANY frame length and hop -- the endpoint path's standard case (16-bit wav data, get_amplitude and
// get_zcr of endpoint.py:109-126, 182-198; 30 ms / 10 ms frames are 480 / 160 samples at 16 kHz, 1323 / 441 at 44.1 kHz)
// -- in integer arithmetic on packed halves: 473 vector instructions per group of 16 frames where vad_sum_kernel issued
// 811 (it was issue bound, 18 us per 49 MB; vad_vec_kernel, which served the frames that are not whole vectors, took 54 us
// per 69 MB):
//   * per 4-sample vector (two dwords w0 = x1:x0, w1 = x3:x2): sgn = clamp(x, -1, 1) on both halves (v_pk_min / max_i16),
//     |x0| + .. + |x3| = two v_dot2_i32_i16 of (x, sgn); the sign changes of the pairs (x[i-1], x[i]) are the halves
//     equal to -1 of sgn x (sgn shifted by one sample: v_alignbit over the previous dword, the previous LANE's last
//     dword through DPP), counted with one v_perm / v_and / v_bcnt.  Exact: nothing is rounded anywhere.
//   * both numbers go to LDS in the vectors' own order; every lane then takes 12 CONSECUTIVE vectors (three
//     conflict-free ds_read_b128 per array: a lane stride of 12 dwords spreads 16 lanes over all 64 banks), forms their
//     running sums, the wave scans the 64 lane totals with DPP, and the prefix sums go back in place.
//   * frame f = samples [f S, f S + L) of the group = a run of whole vectors, two prefix differences instead of a walk
//     over L/4 vectors, plus at most three samples in front and three behind, which the frame's lane reads from memory
//     itself at the group's start (with the sample in front of each run, for the pairs); a frame that starts on a vector
//     drops that vector's first pair.
#define VAD_SCAN_CH 12     // vectors per lane in the prefix phase: 64 x 12 = 64 x VAD_NSTAGE
typedef short vad_s2 __attribute__((ext_vector_type(2)));
typedef int vad_i4 __attribute__((ext_vector_type(4)));
typedef unsigned int vad_u2u __attribute__((ext_vector_type(2), aligned(2)));

// clamp(x, -1, 1) on both halves (hipcc expands __builtin_elementwise_min / max on short2 into compares and selects)
__device__ __forceinline__ vad_s2 vad_sgn2(vad_s2 x) {
    uint32_t t;
    asm("v_pk_min_i16 %0, %1, %2\n\tv_pk_max_i16 %0, %0, %3" : "=&v"(t) : "v"(__builtin_bit_cast(uint32_t, x)), "s"(0x00010001u), "s"(0xffffffffu));
    return __builtin_bit_cast(vad_s2, t);
}

template <int FR, bool RAGGED>
__global__ __launch_bounds__(64 * VAD_WAVES) void vad_scan_kernel(VadParams P, BatchGeom bg,
                                                                  const int16_t* __restrict__ wave,
                                                                  double* __restrict__ amp_sum,
                                                                  int32_t* __restrict__ zcr) {
    constexpr int SHIFT = FR == 16 ? 4 : (FR == 8 ? 3 : 2);
    // eight frames per wave (7 S + L samples: 4410 at 44.1 kHz, 4800 at 48 kHz) stage up to 20 rounds of 64 vectors; a lane
    // stride of 20 dwords in the prefix phase is conflict free for 16-byte reads like the stride of 12
    constexpr int NS = FR == 8 ? 20 : VAD_NSTAGE, CH = FR == 8 ? 20 : VAD_SCAN_CH;
    static_assert(CH * 64 >= NS * 64 && CH % 4 == 0, "the prefix phase covers every staged vector");
    extern __shared__ __attribute__((aligned(256))) float smem_f[];
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int32_t* pa = reinterpret_cast<int32_t*>(smem_f) + wid * (2 * 64 * CH);   // |x| sums per vector, then their prefix sums
    int32_t* pe = pa + 64 * CH;                                               // sign changes, likewise
    const int total_groups = RAGGED ? P.group_off[bg.n_utt] : (int)P.total_groups;
    const int gstride = (int)gridDim.x * VAD_WAVES;
    const int nr = (P.span_vec + 63) >> 6;          // rounds of 64 vectors (<= NS)

    // Where a group lives (two levels of dependent scalar loads for ragged batches).  Issuing the NEXT group's behind this
    // group's sample loads was measured and lost 1.4 us per 49 MB launch: the wait for them lands in front of the arithmetic.
    struct Loc { int t0, T, nsamp; int64_t s0, row0; };
    auto locate = [&](int G) -> Loc {
        Loc g;
        if constexpr (RAGGED) {
            const int utt = P.group_utt[G];
            g.t0 = (G - P.group_off[utt]) << SHIFT;
            g.s0 = bg.sample_off[utt];
            g.nsamp = (int)(bg.sample_off[utt + 1] - g.s0);
            g.row0 = bg.frame_off[utt];
            g.T = (int)(bg.frame_off[utt + 1] - g.row0);
        } else {
            const int gpu = (int)P.groups_per_utt;
            const int utt = G / gpu;
            g.t0 = (G - utt * gpu) << SHIFT;
            g.nsamp = (int)bg.uniform_samples;
            g.T = (int)bg.uniform_frames;
            g.s0 = (int64_t)utt * bg.uniform_samples;
            g.row0 = (int64_t)utt * bg.uniform_frames;
        }
        return g;
    };
    int G = __builtin_amdgcn_readfirstlane((int)blockIdx.x * VAD_WAVES + wid);
    if (G >= total_groups) return;
    Loc cur = locate(G);
    for (;;) {
        const int t0 = cur.t0, T = cur.T, nsamp = cur.nsamp;
        const int64_t s0 = cur.s0, row0 = cur.row0;
        const int base = t0 * P.S;
        const int16_t* gp = wave + s0 + base;
        // Frame `lane` = samples [flo, fhi) of the group = the whole vectors [vlo, vhi) plus up to three samples in front
        // and three behind (none when L and S are multiples of 4).  The lane reads those few, and the sample in front of
        // each run, itself: x[flo - 1 .. flo + 2] and x[4 vhi - 1 .. 4 vhi + 2], zeros outside the clip.  Used in the frame phase, so nothing waits for them here.
        const int flo = lane * P.S, fhi = flo + P.L;
        const int vlo = (flo + 3) >> 2, vhi = fhi >> 2;
        int hw[4], tw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ih = flo - 1 + k, it = 4 * vhi - 1 + k;
            const bool okh = lane < FR && ih >= 0 && base + ih < nsamp, okt = lane < FR && base + it < nsamp;
            // (lanes with nothing to read load from the result buffer instead -- always mapped, value dropped: the clip
            //  itself may be empty, and an empty LAST clip has no valid sample address at all)
            const int16_t* const idle = reinterpret_cast<const int16_t*>(amp_sum);
            hw[k] = *(okh ? gp + ih : idle);
            tw[k] = *(okt ? gp + it : idle);
            hw[k] = okh ? hw[k] : 0;
            tw[k] = okt ? tw[k] : 0;
        }
        // ---- all loads first, no branch on a lane's position: a vector that crosses the clip's end is read as the clip's
        //      LAST four samples and shifted down, zeros above (clips shorter than 4 samples: element by element) ----
        uint2 raw[NS];
        if (nsamp >= 4) {
            const int16_t* const tailp = gp + (nsamp - base - 4);
#pragma unroll
            for (int r = 0; r < NS; ++r)
                if (r < nr) {
                    const int16_t* vp = gp + 4 * (lane + 64 * r);
                    if (base + 256 * (r + 1) <= nsamp) {       // the whole round lies inside the clip (wave-uniform)
                        const vad_u2u t = *reinterpret_cast<const vad_u2u*>(vp);
                        raw[r] = make_uint2(t.x, t.y);
                    } else if (base + 256 * r >= nsamp) {       // the whole round lies behind the clip's end
                        raw[r] = make_uint2(0u, 0u);
                    } else {                                    // the one round in between
                        const int c = nsamp - (base + 4 * (lane + 64 * r));     // samples of this vector inside the clip
                        const vad_u2u t = *reinterpret_cast<const vad_u2u*>(c >= 4 ? vp : tailp);
                        raw[r] = make_uint2(t.x, t.y);                          // shifted into place below, once everything is on its way
                    }
                }
        } else {
#pragma unroll
            for (int r = 0; r < NS; ++r)
                if (r < nr) {
                    const int rel = base + 4 * (lane + 64 * r);
                    uint32_t e[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = rel + k < nsamp ? (uint32_t)(uint16_t)gp[4 * (lane + 64 * r) + k] : 0u;
                    raw[r] = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
                }
        }
        if (nsamp >= 4 && base + 256 * nr > nsamp) {    // the group reaches the clip's end (wave-uniform)
#pragma unroll
            for (int r = 0; r < NS; ++r)
                if (r < nr && base + 256 * (r + 1) > nsamp && base + 256 * r < nsamp) {
                    const int c = nsamp - (base + 4 * (lane + 64 * r));
                    const uint64_t q = ((uint64_t)raw[r].y << 32) | raw[r].x;
                    const uint64_t qs = c >= 4 ? q : (c <= 0 ? 0ull : q >> (16 * (4 - c)));
                    raw[r] = make_uint2((uint32_t)qs, (uint32_t)(qs >> 32));
                }
        }
        int left = 0;   // sgn of the two samples in front of lane 0's vector (the first staged sample has nothing in front)
#pragma unroll
        for (int r = 0; r < NS; ++r)
            if (r < nr) {
                const vad_s2 x0 = __builtin_bit_cast(vad_s2, raw[r].x), x1 = __builtin_bit_cast(vad_s2, raw[r].y);
                const vad_s2 g0 = vad_sgn2(x0), g1 = vad_sgn2(x1);
                const int g1i = __builtin_bit_cast(int, g1);
                const int gprev = __builtin_amdgcn_update_dpp(left, g1i, 0x138, 0xF, 0xF, false);   // wave_shr:1, lane 0 keeps `left`
                left = __builtin_amdgcn_readlane(g1i, 63);
                int a = __builtin_amdgcn_sdot2(x0, g0, 0, false);
                a = __builtin_amdgcn_sdot2(x1, g1, a, false);
                const uint32_t g0u = __builtin_bit_cast(uint32_t, g0);
                const vad_s2 t0v = __builtin_bit_cast(vad_s2, __builtin_amdgcn_alignbit(g0u, (uint32_t)gprev, 16));   // sgn x0 : sgn x[-1]
                const vad_s2 t1v = __builtin_bit_cast(vad_s2, __builtin_amdgcn_alignbit((uint32_t)g1i, g0u, 16));      // sgn x2 : sgn x1
                const vad_s2 p0 = g0 * t0v, p1 = g1 * t1v;                                                             // -1 where the pair changes sign
                const uint32_t m = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1), 0x03010705u) & 0x80808080u;
                pa[lane + 64 * r] = a;
                pe[lane + 64 * r] = __builtin_popcount(m);
            }
        F512_FENCE();
        // ---- prefix sums over the vectors: lane l owns vectors 12 l .. 12 l + 11 ----
        {
            vad_i4* qa = reinterpret_cast<vad_i4*>(pa + CH * lane);
            vad_i4* qe = reinterpret_cast<vad_i4*>(pe + CH * lane);
            int va[CH], ve[CH];
#pragma unroll
            for (int k = 0; k < CH / 4; ++k) {
                const vad_i4 x = qa[k], y = qe[k];
                va[4 * k] = x.x; va[4 * k + 1] = x.y; va[4 * k + 2] = x.z; va[4 * k + 3] = x.w;
                ve[4 * k] = y.x; ve[4 * k + 1] = y.y; ve[4 * k + 2] = y.z; ve[4 * k + 3] = y.w;
            }
#pragma unroll
            for (int k = 1; k < CH; ++k) { va[k] += va[k - 1]; ve[k] += ve[k - 1]; }
            const int oa = dsp_wave_scan_i32(va[CH - 1]) - va[CH - 1];   // sum of the lanes in front
            const int oe = dsp_wave_scan_i32(ve[CH - 1]) - ve[CH - 1];
#pragma unroll
            for (int k = 0; k < CH / 4; ++k) {
                vad_i4 x, y;
                x.x = va[4 * k] + oa; x.y = va[4 * k + 1] + oa; x.z = va[4 * k + 2] + oa; x.w = va[4 * k + 3] + oa;
                y.x = ve[4 * k] + oe; y.y = ve[4 * k + 1] + oe; y.z = ve[4 * k + 2] + oe; y.w = ve[4 * k + 3] + oe;
                qa[k] = x;
                qe[k] = y;
            }
        }
        F512_FENCE();
        // ---- frame f: two prefix differences over its whole vectors, the edge samples one by one ----
        if (lane < FR) {
            int sa = pa[vhi - 1], se = pe[vhi - 1];
            if (vlo > 0) {
                sa -= pa[vlo - 1];
                se -= pe[vlo - 1];
            }
            const int hr = 4 * vlo - flo, tq = fhi - 4 * vhi;      // samples in front of / behind the whole vectors
            auto cross = [](int a, int b) -> int { return a * b < 0 ? 1 : 0; };   // int16 x int16: exact
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < hr) {
                    sa += abs(hw[1 + k]);                          // x[flo + k]
                    if (k > 0) se += cross(hw[k], hw[1 + k]);      // pairs (i - 1, i), i = flo + 1 .. flo + hr - 1
                }
                if (k < tq) {
                    sa += abs(tw[1 + k]);                          // x[4 vhi + k]
                    se += cross(tw[k], tw[1 + k]);                 // pairs (i - 1, i), i = 4 vhi .. fhi - 1
                }
            }
            // the first whole vector's count starts with the pair (4 vlo - 1, 4 vlo): outside the frame when the frame starts
            // there (frame 0 of a group: that vector was staged with a zero in front, nothing to take back)
            if (hr == 0 && lane > 0) se -= cross(hw[0], hw[1]);
            const int t = t0 + lane;
            if (t < T) {
                amp_sum[row0 + t] = (double)sa;
                zcr[row0 + t] = se;
            }
        }
        F512_FENCE();
        G += gstride;
        if (G >= total_groups) break;
        cur = locate(G);
    }
}

// Picks the tile shape; returns 0 if the configuration has to take the one-wave-per-frame kernel.
static inline int vad_tile_frames(int32_t L, int32_t S) {
    if (L < 64 || S < 1) return 0;
    if ((15 * (int64_t)S + L + 3) / 4 + 1 <= 64 * VAD_NSTAGE) return 16;
    if ((3 * (int64_t)S + L + 3) / 4 + 1 <= 64 * VAD_NSTAGE) return 4;
    return 0;
}

// int16, sum |x|: does vad_scan_kernel take this framing with EIGHT frames per wave (up to 20 rounds of 64 vectors)?  Only where
// sixteen do not fit and four would otherwise be used.
static inline bool vad_scan_frames8(int32_t L, int32_t S, int dtype, int32_t use_sq) {
    if (dtype != DSP_WAVE_I16 || use_sq || L < 64 || S < 1) return false;
    if ((15 * (int64_t)S + L + 3) / 4 + 1 <= 64 * VAD_NSTAGE) return false;      // sixteen frames per wave fit
    return (7 * (int64_t)S + L + 3) / 4 <= 64 * 20 && (3 * (int64_t)S + L + 3) / 4 + 1 <= 64 * VAD_NSTAGE;
}

static inline bool vad_tile_applicable(const BatchGeom& bg, const void* d_wave, int dtype, int FR) {
    if (FR == 0) return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_wave);
    if ((a % (dtype == DSP_WAVE_I16 ? 8 : 16)) != 0) return false;
    if (bg.uniform_samples > 0) {
        if ((bg.uniform_samples % 4) != 0) return false;
        return bg.uniform_samples <= 0x3fffffff && ((bg.uniform_frames + FR - 1) / FR) * bg.n_utt <= 0x3fffffff;
    }
    return bg.total_frames / FR + bg.n_utt <= 0x3fffffff;
}

template <int DTYPE, int FR, bool RAGGED>
static int vad_tile_launch_k(const VadParams& P, const BatchGeom& bg, const void* d_wave, double* d_amp,
                             int32_t* d_zcr, int64_t groups_bound, hipStream_t st) {
    const size_t lds = (size_t)VAD_WAVES * P.wave_floats * sizeof(float);
    int64_t blocks = (groups_bound + VAD_WAVES - 1) / VAD_WAVES;
    const int64_t cap = (int64_t)dsp_cu_count() * 3;   // <= 48 KB per workgroup: three resident workgroups per CU
    if (blocks > cap) {
        const int64_t rounds = (blocks + cap - 1) / cap;
        blocks = (blocks + rounds - 1) / rounds;
    }
    // int16 samples: the |x| sum of one lane stays below 2^24, so fp32 partial sums are exact
    const bool f32_exact = DTYPE == DSP_WAVE_I16 && !P.use_sq && (P.L + 64 / FR - 1) / (64 / FR) < 512;
    static const bool force_walk = getenv("DSP_VAD_WALK") != nullptr;   // A/B aid: keep the per-frame walk
    static const bool no_sum = getenv("DSP_VAD_NOSUM") != nullptr;       // A/B aid: skip vad_sum_kernel
    static const bool no_scan = getenv("DSP_VAD_NOSCAN") != nullptr;   // A/B aid: keep the round-3 kernels for int16 input
    if constexpr (DTYPE == DSP_WAVE_I16) {
        // int16, sum |x|, any frame length and hop: integer partials and prefix sums (a vector's sum stays below 2^17, a
        // group's below 2^27)
        VadParams Q = P;
        Q.span_vec = ((FR - 1) * P.S + P.L + 3) / 4;
        if (!P.use_sq && !no_scan && !force_walk && !no_sum && Q.span_vec <= 64 * VAD_NSTAGE && P.L >= 8) {
            const size_t ldss = (size_t)VAD_WAVES * 2 * 64 * VAD_SCAN_CH * sizeof(int32_t);
            int64_t blockss = (groups_bound + VAD_WAVES - 1) / VAD_WAVES;
            const int64_t caps = (int64_t)dsp_cu_count() * 6;          // 24.6 KB per workgroup: six per CU
            if (blockss > caps) blockss = caps;
            vad_scan_kernel<FR, RAGGED><<<(int)blockss, 64 * VAD_WAVES, ldss, st>>>(Q, bg, static_cast<const int16_t*>(d_wave), d_amp, d_zcr);
            return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
        }
    }
    if ((P.L % 4) == 0 && (P.S % 4) == 0 && !force_walk && !no_sum) {
        // vector-aligned frames: visit-once kernel with 8 (12) bytes of LDS per vector, both input types
        VadParams Q = P;
        Q.span_vec = ((FR - 1) * P.S + P.L) / 4;
        const size_t sum_bytes = f32_exact ? 4 : 8;
        Q.off_e = (int32_t)((((size_t)Q.span_vec * sum_bytes + 15) / 16 * 16) / 4);       // floats
        Q.wave_floats = (int32_t)(((size_t)Q.off_e + Q.span_vec + 63) / 64 * 64);
        if (Q.span_vec <= 64 * VAD_NSTAGE) {
            const size_t lds3 = (size_t)VAD_WAVES * Q.wave_floats * sizeof(float);
            int64_t blocks3 = (groups_bound + VAD_WAVES - 1) / VAD_WAVES;
            int per_cu = (int)((size_t)(150 * 1024) / (lds3 ? lds3 : 1));
            if (per_cu > 8) per_cu = 8;
            if (per_cu < 1) per_cu = 1;
            const int64_t cap3 = (int64_t)dsp_cu_count() * per_cu;
            if (blocks3 > cap3) blocks3 = cap3;
            if (f32_exact) {
                auto k = vad_sum_kernel<DTYPE, FR, RAGGED, true>;
                static size_t granted[DSP_MAX_DEVICES] = {};
                if (lds3 > 48 * 1024 && dsp_ensure_dynamic_lds((const void*)k, lds3, granted) != 0) return DSP_EHIP;
                k<<<(int)blocks3, 64 * VAD_WAVES, lds3, st>>>(Q, bg, d_wave, d_amp, d_zcr);
            } else {
                auto k = vad_sum_kernel<DTYPE, FR, RAGGED, false>;
                static size_t granted[DSP_MAX_DEVICES] = {};
                if (lds3 > 48 * 1024 && dsp_ensure_dynamic_lds((const void*)k, lds3, granted) != 0) return DSP_EHIP;
                k<<<(int)blocks3, 64 * VAD_WAVES, lds3, st>>>(Q, bg, d_wave, d_amp, d_zcr);
            }
            return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
        }
    }
    if (f32_exact && !force_walk) {
        // visit-once kernel: samples + per-vector sums + sign bits per wave (6 floats per staged vector)
        VadParams Q = P;
        Q.off_a4 = 4 * (P.span_vec + 1);
        Q.off_e = 5 * (P.span_vec + 1);
        Q.wave_floats = (6 * (P.span_vec + 1) + 63) / 64 * 64;
        const size_t lds2 = (size_t)VAD_WAVES * Q.wave_floats * sizeof(float);
        auto k = vad_vec_kernel<FR, RAGGED>;
        static size_t granted[DSP_MAX_DEVICES] = {};
        if (lds2 > 48 * 1024 && dsp_ensure_dynamic_lds((const void*)k, lds2, granted) != 0) return DSP_EHIP;
        int64_t blocks2 = (groups_bound + VAD_WAVES - 1) / VAD_WAVES;
        const int64_t cap2 = (int64_t)dsp_cu_count() * 2;
        if (blocks2 > cap2) {
            const int64_t rounds = (blocks2 + cap2 - 1) / cap2;
            blocks2 = (blocks2 + rounds - 1) / rounds;
        }
        k<<<(int)blocks2, 64 * VAD_WAVES, lds2, st>>>(Q, bg, d_wave, d_amp, d_zcr);
        return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
    }
    if (f32_exact)
        vad_tile_kernel<DTYPE, FR, RAGGED, 0><<<(int)blocks, 64 * VAD_WAVES, lds, st>>>(P, bg, d_wave, d_amp, d_zcr);
    else if (!P.use_sq)
        vad_tile_kernel<DTYPE, FR, RAGGED, 1><<<(int)blocks, 64 * VAD_WAVES, lds, st>>>(P, bg, d_wave, d_amp, d_zcr);
    else
        vad_tile_kernel<DTYPE, FR, RAGGED, 2><<<(int)blocks, 64 * VAD_WAVES, lds, st>>>(P, bg, d_wave, d_amp, d_zcr);
    return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
}

template <int FR>
static int vad_tile_launch_t(VadParams P, const BatchGeom& bg, const void* d_wave, int dtype, double* d_amp,
                             int32_t* d_zcr, hipStream_t st, const DspRaggedTables* pre = nullptr) {
    P.span_vec = ((FR - 1) * P.S + P.L + 3) / 4;
    P.wave_floats = (4 * (P.span_vec + 1) + 63) / 64 * 64;
    constexpr int SHIFT = FR == 16 ? 4 : 2;
    if (bg.uniform_samples > 0) {
        P.groups_per_utt = (bg.uniform_frames + FR - 1) / FR;
        P.total_groups = P.groups_per_utt * bg.n_utt;
        if (dtype == DSP_WAVE_I16)
            return vad_tile_launch_k<DSP_WAVE_I16, FR, false>(P, bg, d_wave, d_amp, d_zcr, P.total_groups, st);
        return vad_tile_launch_k<DSP_WAVE_F32, FR, false>(P, bg, d_wave, d_amp, d_zcr, P.total_groups, st);
    }
    const int64_t bound = bg.total_frames / FR + bg.n_utt;
    DspWorkspace* w = nullptr;
    if (pre != nullptr && pre->shift == SHIFT) {   // tables of a dsp_layout: built once per batch shape
        P.group_off = pre->group_off;
        P.group_utt = pre->group_utt;
    } else {
        const size_t ws_bytes = ((size_t)bg.n_utt + 1 + (size_t)bound) * sizeof(int32_t);
        w = dsp_workspace_pool().acquire(ws_bytes, st);
        if (!w) return DSP_EHIP;
        int32_t* group_off = static_cast<int32_t*>(w->ptr);
        int32_t* group_utt = group_off + bg.n_utt + 1;
        f512_build_group_tables(bg.frame_off, bg.n_utt, SHIFT, group_off, group_utt, st);
        P.group_off = group_off;
        P.group_utt = group_utt;
    }
    int rc;
    if (dtype == DSP_WAVE_I16)
        rc = vad_tile_launch_k<DSP_WAVE_I16, FR, true>(P, bg, d_wave, d_amp, d_zcr, bound, st);
    else
        rc = vad_tile_launch_k<DSP_WAVE_F32, FR, true>(P, bg, d_wave, d_amp, d_zcr, bound, st);
    if (w != nullptr && dsp_workspace_pool().release(w, st) != 0 && rc == DSP_OK) rc = DSP_EHIP;
    return rc;
}

static inline int vad_tile_launch(int FR, int32_t L, int32_t S, int32_t use_sq, const BatchGeom& bg, const void* d_wave,
                                  int dtype, double* d_amp, int32_t* d_zcr, hipStream_t st,
                                  const DspRaggedTables* pre = nullptr) {
    VadParams P;
    memset(&P, 0, sizeof(P));
    P.L = L; P.S = S; P.use_sq = use_sq;
    if (FR == 16) return vad_tile_launch_t<16>(P, bg, d_wave, dtype, d_amp, d_zcr, st, pre);
    // int16 clips whose 16-frame groups do not fit (44.1 kHz: 30 ms / 10 ms = 1323 / 441 samples): eight frames per wave
    // (1103 of 1280 staged vectors; 1200 at 48 kHz) instead of four -- 1.25 x instead of 1.5 x of the samples read, half the prefix-scan
    // work per frame: 23.4 vs 30.0 us per 69 MB (same-job A/B)
    if (vad_scan_frames8(L, S, dtype, use_sq)) {
        static const bool no_scan = getenv("DSP_VAD_NOSCAN") != nullptr;
        static const bool force_walk = getenv("DSP_VAD_WALK") != nullptr, no_sum = getenv("DSP_VAD_NOSUM") != nullptr;
        if (!no_scan && !force_walk && !no_sum) {
            P.span_vec = (int32_t)((7 * (int64_t)S + L + 3) / 4);
            const size_t ldss = (size_t)VAD_WAVES * 2 * 64 * 20 * sizeof(int32_t);     // CH = 20 dwords per lane and array
            const int64_t cap = (int64_t)dsp_cu_count() * 3;                             // 40 KB per workgroup: three per CU
            if (bg.uniform_samples > 0) {
                P.groups_per_utt = (bg.uniform_frames + 7) / 8;
                P.total_groups = P.groups_per_utt * bg.n_utt;
                int64_t blocks = (P.total_groups + VAD_WAVES - 1) / VAD_WAVES;
                if (blocks > cap) blocks = cap;
                vad_scan_kernel<8, false><<<(int)blocks, 64 * VAD_WAVES, ldss, st>>>(P, bg, static_cast<const int16_t*>(d_wave), d_amp, d_zcr);
                return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
            }
            const int64_t bound = bg.total_frames / 8 + bg.n_utt;
            DspWorkspace* w = nullptr;
            if (pre != nullptr && pre->shift == 3) {
                P.group_off = pre->group_off;
                P.group_utt = pre->group_utt;
            } else if (pre != nullptr && pre->shift2 == 3) {
                P.group_off = pre->group_off2;
                P.group_utt = pre->group_utt2;
            } else {
                const size_t ws_bytes = ((size_t)bg.n_utt + 1 + (size_t)bound) * sizeof(int32_t);
                w = dsp_workspace_pool().acquire(ws_bytes, st);
                if (!w) return DSP_EHIP;
                int32_t* group_off = static_cast<int32_t*>(w->ptr);
                f512_build_group_tables(bg.frame_off, bg.n_utt, 3, group_off, group_off + bg.n_utt + 1, st);
                P.group_off = group_off;
                P.group_utt = group_off + bg.n_utt + 1;
            }
            int64_t blocks = (bound + VAD_WAVES - 1) / VAD_WAVES;
            if (blocks > cap) blocks = cap;
            vad_scan_kernel<8, true><<<(int)blocks, 64 * VAD_WAVES, ldss, st>>>(P, bg, static_cast<const int16_t*>(d_wave), d_amp, d_zcr);
            int rc = hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
            if (w != nullptr && dsp_workspace_pool().release(w, st) != 0 && rc == DSP_OK) rc = DSP_EHIP;
            return rc;
        }
    }
    return vad_tile_launch_t<4>(P, bg, d_wave, dtype, d_amp, d_zcr, st, pre);
}
