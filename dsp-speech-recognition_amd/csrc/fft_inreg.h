// In-register forward complex FFTs of 4/8/16/32 points, natural order in and out.
// Every index is a compile-time constant after unrolling, so the arrays live in VGPRs and the
// twiddles fold into literals / SGPRs.  Inputs with index >= NZ are known zeros (zero-padded
// frame tail) and are skipped at the first butterfly level.
#pragma once

#include <hip/hip_runtime.h>

#include "fft_consts.h"

#define DSP_HD __host__ __device__ __forceinline__

struct cpx {
    float x, y;
};

DSP_HD cpx cadd(cpx a, cpx b) { return {a.x + b.x, a.y + b.y}; }
DSP_HD cpx csub(cpx a, cpx b) { return {a.x - b.x, a.y - b.y}; }
DSP_HD cpx cmulc(cpx a, float wr, float wi) {  // a * (wr + i wi)
    return {fmaf(a.x, wr, -a.y * wi), fmaf(a.x, wi, a.y * wr)};
}

// v * W_N^K, W_N = exp(-2 pi i / N), K compile-time.
template <int N, int K>
DSP_HD cpx mulw(cpx v) {
    constexpr int k = ((K % N) + N) % N;
    constexpr float R = 0.70710678118654752440f;
    if constexpr (k == 0) return v;
    else if constexpr (4 * k == N) return {v.y, -v.x};          // * (-i)
    else if constexpr (2 * k == N) return {-v.x, -v.y};         // * (-1)
    else if constexpr (4 * k == 3 * N) return {-v.y, v.x};      // * (+i)
    else if constexpr (8 * k == N) return {R * (v.x + v.y), R * (v.y - v.x)};
    else if constexpr (8 * k == 3 * N) return {R * (v.y - v.x), -R * (v.x + v.y)};
    else if constexpr (8 * k == 5 * N) return {-R * (v.x + v.y), R * (v.x - v.y)};
    else if constexpr (8 * k == 7 * N) return {R * (v.x - v.y), R * (v.x + v.y)};
    else {
        static_assert(32 % N == 0, "twiddle tables cover N | 32");
        constexpr int idx = k * (32 / N);
        return cmulc(v, DSP_COS32[idx], -DSP_SIN32[idx]);
    }
}

// 4-point forward DFT of (a, b, c, d); NZ = how many leading inputs may be non-zero.
template <int NZ = 4>
DSP_HD void dft4(cpx& a, cpx& b, cpx& c, cpx& d) {
    static_assert(NZ >= 1 && NZ <= 4, "");
    cpx s0, s1, s2, s3;
    if constexpr (NZ >= 3) { s0 = cadd(a, c); s1 = csub(a, c); } else { s0 = a; s1 = a; }
    if constexpr (NZ == 4) { s2 = cadd(b, d); s3 = csub(b, d); }
    else if constexpr (NZ >= 2) { s2 = b; s3 = b; }
    if constexpr (NZ == 1) { b = a; c = a; d = a; return; }
    else {
        a = cadd(s0, s2);
        c = csub(s0, s2);
        b = {s1.x + s3.y, s1.y - s3.x};  // s1 - i s3
        d = {s1.x - s3.y, s1.y + s3.x};  // s1 + i s3
    }
}

template <int N>
struct FFTReg;

template <>
struct FFTReg<4> {
    template <int NZ = 4>
    static DSP_HD void run(cpx (&x)[4]) { dft4<NZ>(x[0], x[1], x[2], x[3]); }
};

template <>
struct FFTReg<8> {
    // 8 = 2 x 4: n = 2a + b (a<4, b<2), k = d + 4e (d<4, e<2)
    template <int NZ = 8>
    static DSP_HD void run(cpx (&x)[8]) {
        cpx y[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            cpx t0 = x[b], t1 = x[2 + b], t2 = x[4 + b], t3 = x[6 + b];
            dft4<4>(t0, t1, t2, t3);
            y[b][0] = t0; y[b][1] = t1; y[b][2] = t2; y[b][3] = t3;
        }
        y[1][1] = mulw<8, 1>(y[1][1]);
        y[1][2] = mulw<8, 2>(y[1][2]);
        y[1][3] = mulw<8, 3>(y[1][3]);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            x[d] = cadd(y[0][d], y[1][d]);
            x[d + 4] = csub(y[0][d], y[1][d]);
        }
    }
};

namespace fftdetail {
template <int N, int B, int D>
struct TwRow {  // y[b][d] *= W_N^(b*d) for b = B
    template <int NB>
    static DSP_HD void apply(cpx (&y)[NB][4]) {
        y[B][D] = mulw<N, B * D>(y[B][D]);
        if constexpr (D + 1 < 4) TwRow<N, B, D + 1>::apply(y);
    }
};
template <int N, int B, int NB>
struct TwAll {
    static DSP_HD void apply(cpx (&y)[NB][4]) {
        TwRow<N, B, 1>::apply(y);
        if constexpr (B + 1 < NB) TwAll<N, B + 1, NB>::apply(y);
    }
};
template <int NZ, int STRIDE, int B>
constexpr int nz_of() {  // how many of x[B], x[B+STRIDE], x[B+2*STRIDE], x[B+3*STRIDE] are < NZ
    int c = 0;
    for (int a = 0; a < 4; ++a) c += (B + a * STRIDE < NZ) ? 1 : 0;
    return c < 1 ? 1 : c;
}
template <int N, int NZ, int B>
struct Stage1 {  // 4-point DFTs over a (n = (N/4) a + b), for b = B .. N/4-1
    static DSP_HD void run(const cpx (&x)[N], cpx (&y)[N / 4][4]) {
        constexpr int NB = N / 4;
        cpx t0 = x[B], t1 = x[NB + B], t2 = x[2 * NB + B], t3 = x[3 * NB + B];
        dft4<nz_of<NZ, NB, B>()>(t0, t1, t2, t3);
        y[B][0] = t0; y[B][1] = t1; y[B][2] = t2; y[B][3] = t3;
#if defined(__HIP_DEVICE_COMPILE__) && defined(FFT_SEQ_BARRIERS)
        if constexpr ((B & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // bound live ranges: 2 butterflies in flight
#endif
        if constexpr (B + 1 < NB) Stage1<N, NZ, B + 1>::run(x, y);
    }
};
}  // namespace fftdetail

// N = 4 * NB (NB = 4 or 8): n = NB*a + b, k = d + 4e.
template <int N>
struct FFTReg {
    static_assert(N == 16 || N == 32, "");
    template <int NZ = N>
    static DSP_HD void run(cpx (&x)[N]) {
        constexpr int NB = N / 4;
        cpx y[NB][4];
        fftdetail::Stage1<N, NZ, 0>::run(x, y);
        fftdetail::TwAll<N, 1, NB>::apply(y);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            cpx u[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) u[b] = y[b][d];
            FFTReg<NB>::run(u);
#pragma unroll
            for (int e = 0; e < NB; ++e) x[d + 4 * e] = u[e];
#if defined(__HIP_DEVICE_COMPILE__) && defined(FFT_SEQ_BARRIERS)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }
};
