// NFFT = 512 feature kernel on the matrix pipe, frame-per-product form (gfx950): waveform -> MFCC (+ delta,
// delta-delta) rows for dense batches.  Same path as kernels_mfma512.h -- sigproc.preemphasis (sigproc.py:178-185) ->
// framesig * window (sigproc.py:66-98) -> powspec (sigproc.py:136-158) -> fbank / mfcc (base.py:8-32, 60-68) ->
// delta (base.py:70-79) -- with the two DFT stages turned round: the N side of both products is the DFT row k1 of ONE
// frame, not 16 frames (tables and operand maps: mfma512t_tables.h; CPU emulation: tools/mfma512t_emul.py).
//
// Why (round 4): kernels_mfma512.h keeps a 16-frame tile's stage-1 results in 128 registers, 64 KB of per-column
// stage-1 matrices in LDS and runs two waves per SIMD; no unit is busy more than a third of the time.  Here a frame is
// eight registers of samples, both DFT matrices are register resident (one per stage: the window multiplies the
// samples, the twiddle the stage-1 result, both on the vector pipe), nothing but the samples is staged per frame,
// and the stage-1 result IS the stage-2 operand (its rows are the contraction index): no transpose, ~10 KB of LDS per
// wave, three waves per SIMD.
//
// Work unit: a wave owns a PART of an utterance (a run of whole 16-frame tiles; 1..WAVES parts per utterance so that a
// small batch still fills the chip), the waves of an utterance sit in one workgroup.
//  1. staging, per half tile (8 frames): the half's samples (7 S + 512, read once through a bounds-checked descriptor:
//     samples before the utterance and beyond its end read as zero -- the reference's zero padding), pre-emphasised,
//     scaled by a power of two from the half's largest sample, stored COLUMN-MAJOR in LDS (column = sample index mod 16):
//     a lane's stage-1 operand is four conflict-free 8-byte reads;
//  2. per frame: window * samples -> fp16 (hi, lo) (two v_fma_mix per value), six MFMAs, twiddle, (hi, lo), six
//     MFMAs, power, bf16 (hi, lo) -> four registers of the tile's 64;
//  3. per tile: the bins 16 m from column 0 of the 16 frames (six MFMAs), the powers through an 8 KB exchange in LDS
//     (hi halves, then lo halves) onto the N side, mel blocks (bf16), log2, DCT * lifter (six MFMAs);
//  4. delta / delta-delta from a 24-row LDS buffer as in kernels_mfma512.h; the four rows either side of a boundary
//     between two parts wait for the neighbour's cepstra (one workgroup barrier per utterance, no frame is recomputed).
#pragma once

#include "kernels_mfma512.h"
#include "mfma512t_tables.h"

#ifndef M512T_WAVES
#define M512T_WAVES 12
#endif

// Diagnostic build (-DM512T_STAMPS, tools/kbench_m512t.py): per-phase shader-clock sums, one slot per wave.
#ifdef M512T_STAMPS
#define M512T_NSTAMP 16
__device__ unsigned int m512t_stamp_sum[M512T_NSTAMP * 4096];
#define M512T_STAMP(i) do { const unsigned int now_ = m512_clock_t(); stamp_acc_[i] += now_ - stamp_prev_; stamp_prev_ = m512_clock_t(); } while (0)
__device__ __forceinline__ unsigned int m512_clock_t() {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return (unsigned int)t;
}
#else
#define M512T_STAMP(i) do {} while (0)
#endif

struct M512TParams {
    const uint8_t* tables;
    M512TLayout lay;
    int32_t S, C;
    float preemph;
    int32_t delta_n;       // 0: cepstra only ([sum T, ld_out]); 1, 2: rows [sum T, 3 C]
    float inv_den;
    int64_t ld_out;
    int32_t n_utt;
    int32_t samples, frames;    // per utterance
    int32_t tiles;              // ceil(frames / 16)
    int32_t rounds_full;        // rounds of one utterance per wave
    int32_t parts;              // last round: waves per utterance (divides the waves of a workgroup); 0: no last round
};

struct Mfma512TPlan {
    uint8_t* d_tables;
    M512TLayout lay;
};

template <int NMT, int PAT>
__host__ __device__ constexpr int m512t_block_index(int step, int tile) {   // position in the stored order
    int k = 0;
    for (int st = 0; st <= 8; ++st)
        for (int t = 0; t < NMT; ++t) {
            if (st == step && t == tile) return k;
            if (st == 8 || m512t_pattern_has(PAT, NMT, st >> 1, t)) ++k;
        }
    return k;
}

// fp16 (hi, lo) of a pair: hi = the value with its low 13 mantissa bits cleared (exact in fp16 inside the tile scale),
// lo = f16(y - hi), the residual exact in fp32.  Four plain vector instructions per pair (and, and, sub, sub) and two
// packed conversions; v_fma_mixlo / mixhi_f16, which do the same in three, issue at a quarter of the rate
// (tools/probes/valu_enc2.hip: 3.7 ns per wave instruction against 1.1 for v_and / v_sub and 2.0 for v_cvt_pk).
__device__ __forceinline__ void m512t_split2(float y0, float y1, uint32_t& hi, uint32_t& lo) {
    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, y0) & 0xffffe000u);
    const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, y1) & 0xffffe000u);
    const m512_f2 hv = {h0, h1}, lv = {y0 - h0, y1 - h1};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(hv, m512_h2));
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(lv, m512_h2));
}

__device__ __forceinline__ void m512t_split_bf16(float a, float b, uint32_t& hi, uint32_t& lo) {
    const m512_f2 v = {a, b};
    const uint32_t hu = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, m512_b2));
    const float ha = __builtin_bit_cast(float, hu << 16), hb = __builtin_bit_cast(float, hu & 0xffff0000u);
    const m512_f2 r = {a - ha, b - hb};
    hi = hu;
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, m512_b2));
}

// four consecutive samples as floats (one 16-byte load of fp32, one 8-byte load of int16), bounds-checked per dword
template <int DTYPE>
__device__ __forceinline__ m512_f4 m512t_buf_load4(__amdgpu_buffer_rsrc_t rs, uint32_t voff) {
    if constexpr (DTYPE == DSP_WAVE_I16) {
        const m512_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int32_t)voff, 0, 0);
        return m512_f4{(float)(int16_t)(v[0] & 0xffffu), (float)(int16_t)(v[0] >> 16), (float)(int16_t)(v[1] & 0xffffu), (float)(int16_t)(v[1] >> 16)};
    } else {
        return __builtin_bit_cast(m512_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int32_t)voff, 0, 0));
    }
}

// HS: hop in rows of 16 samples (10 for a 160-sample hop).  ND: delta window (base.py:70-79), 0 = cepstra only.
// FPB: frames per straight-line block (2: the products of one frame run beside the vector work of the other).
// PF: the next half tile's samples are requested while this one's frames run (35 registers).
template <int HS, int DTYPE, int NMT, int PAT, int ND, int WAVES, int FPB, bool PF>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void mfcc512t_kernel(M512TParams P, const void* __restrict__ wave,
                                                                          float* __restrict__ out) {
    constexpr bool ROWS = ND > 0;
    constexpr int R = 108;                            // rows per image column: >= 7 HS + 32, R / 4 odd (conflict-free reads)
    static_assert(7 * HS + 32 <= R && (R % 4) == 0 && ((R / 4) & 1) == 1 && (HS % 2) == 0, "image column");
    constexpr int IMG_BYTES = 16 * R * 4, C0_BYTES = 1024, EX_BYTES = 8192;
    static_assert(IMG_BYTES + C0_BYTES <= EX_BYTES, "image + column-0 packets share the exchange buffer");
    constexpr int CB_ROWS = 24;
    constexpr int WAVE_BYTES = EX_BYTES + CB_ROWS * 64 + 8 * 64;
    constexpr int NBLK = m512t_block_index<NMT, PAT>(9, 0);
    constexpr int TAB_BYTES = NBLK * 2048 + 4096;
    constexpr int ESZ = DTYPE == DSP_WAVE_I16 ? 2 : 4;
    constexpr int NVEC = 4 * (7 * HS + 32);           // 16-byte vectors of a half tile's samples: 7 per lane, the last one partial
    extern __shared__ __attribute__((aligned(16))) uint8_t m512t_smem[];
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* const sW = m512t_smem;
    uint8_t* const sM0 = m512t_smem + NBLK * 2048;
    uint8_t* const sWave = m512t_smem + TAB_BYTES + wv * WAVE_BYTES;
    uint8_t* const img = sWave;                                              // sample image of a half tile ...
    uint8_t* const c0 = sWave + IMG_BYTES;                                   // ... + its column-0 packets,
    uint8_t* const ex = sWave;                                               // then the power exchange, then
    float* const db = reinterpret_cast<float*>(sWave);                       // the deltas of an edge tile
    float* const cb = reinterpret_cast<float*>(sWave + EX_BYTES);            // cepstra of frames t0 - 8 .. t0 + 15
    float* const head = cb + CB_ROWS * 16;                                   // the part's first eight cepstra
    const int C = P.C, T = P.frames, Nsamp = P.samples;

    // ---- tables -> LDS (mel blocks | column-0 matrix), every load in flight before the first store
    {
        constexpr int NK = (TAB_BYTES / 16 + 64 * WAVES - 1) / (64 * WAVES);
        m512_u4 tv[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int o = (tid + 64 * WAVES * k) * 16;
            const uint8_t* src = o < NBLK * 2048 ? P.tables + P.lay.off_w + o : P.tables + P.lay.off_m0 + (o - NBLK * 2048);
            if (o < TAB_BYTES) tv[k] = *reinterpret_cast<const m512_u4*>(src);
        }
        m512_u4* z = reinterpret_cast<m512_u4*>(sWave);
        const m512_u4 zero = {0, 0, 0, 0};
        for (int i = lane; i < WAVE_BYTES / 16; i += 64) z[i] = zero;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int o = (tid + 64 * WAVES * k) * 16;
            if (o < TAB_BYTES) *reinterpret_cast<m512_u4*>(m512t_smem + o) = tv[k];
        }
    }
    // ---- register-resident: both DFT matrices, the lane's twiddles and window values, the row sums of the DCT
    m512_h8 f1[2][2], f2[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            f1[u][hl] = reinterpret_cast<const m512_h8*>(P.tables + P.lay.off_f1 + (u * 2 + hl) * 1024)[lane];
            f2[u][hl] = reinterpret_cast<const m512_h8*>(P.tables + P.lay.off_f2 + (u * 2 + hl) * 1024)[lane];
        }
    float twc[4], tws[4], win[8], rowsum[4];
    {
        const m512_f4* tp = reinterpret_cast<const m512_f4*>(P.tables + P.lay.off_tw) + 2 * lane;
        const m512_f4* wp = reinterpret_cast<const m512_f4*>(P.tables + P.lay.off_win) + 2 * lane;
        const m512_f4 t0_ = tp[0], t1_ = tp[1], w0_ = wp[0], w1_ = wp[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            twc[i] = t0_[i]; tws[i] = t1_[i]; win[i] = w0_[i]; win[4 + i] = w1_[i];
            rowsum[i] = reinterpret_cast<const float*>(P.tables + P.lay.off_rowsum)[4 * (lane >> 4) + i];
        }
    }
    __syncthreads();

    // the lane's stage-1 operand in the image: column n, rows 2 g (+ 8 q), eight bytes per read; volatile keeps hipcc
    // from pairing neighbours into ds_read2_b64 (twice the LDS cycles per byte)
    typedef __attribute__((address_space(3))) const volatile m512_f2* lds_cvf2;
    const lds_cvf2 aB = reinterpret_cast<lds_cvf2>((uint32_t)reinterpret_cast<uintptr_t>(img + 4 * ((lane & 15) * R + 2 * (lane >> 4))));
    float* const imgw = reinterpret_cast<float*>(img) + 4 * (lane & 3) * R + (lane >> 2);   // staging: vector lane + 64 i
    const float ncpre = -P.preemph;
    const int Wd = ROWS ? 3 * C : (int)P.ld_out;
    const float inv = P.inv_den;

#ifdef M512T_STAMPS
    unsigned int stamp_acc_[M512T_NSTAMP] = {0}, stamp_prev_ = m512_clock_t();
    const unsigned int stamp_t0_ = stamp_prev_;
    unsigned long long stamp_rt0_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_rt0_));
#endif
    // Rounds 0 .. rounds_full - 1: one utterance per wave.  The last round spreads what is left over the chip: `parts`
    // waves per utterance (a batch smaller than the chip's wave slots has only this round).
    const int n_rounds = P.rounds_full + (P.parts > 0 ? 1 : 0);
    for (int round = 0; round < n_rounds; ++round) {
        const bool full = round < P.rounds_full;
        const int parts = full ? 1 : P.parts;
        const int G = WAVES / parts;                                   // utterances of a workgroup in this round
        const int u_local = wv / parts, part = wv - u_local * parts;
        const int utt = full ? (round * (int)gridDim.x + (int)blockIdx.x) * WAVES + wv
                             : P.rounds_full * (int)gridDim.x * WAVES + (int)blockIdx.x * G + u_local;
        const bool active = u_local < G && utt < P.n_utt;
        const int tile_lo = (int)((int64_t)part * P.tiles / parts), tile_hi = (int)((int64_t)(part + 1) * P.tiles / parts);
        const int f_lo = 16 * tile_lo, f_hi = 16 * tile_hi < T ? 16 * tile_hi : T;
        const bool inner_left = f_lo > 0, inner_right = f_hi < T;
        const int J = tile_hi - tile_lo;
        __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);
        if (active) {
            const uint8_t* ubase = reinterpret_cast<const uint8_t*>(wave) + (int64_t)utt * Nsamp * ESZ;
            // x[u]: u < N; x[u - 1] through rs_p: 1 <= u <= N - 1.  Outside both read as zero: y[0] = x[0], and a zero
            // tail AFTER pre-emphasis (sigproc.py:79-91 pads y, not x)
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(ubase), 0, (int32_t)(Nsamp * ESZ), 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(ubase), 0, (int32_t)((Nsamp - 1) * ESZ), 0x00020000);
            // rows [f_lo, f_hi) of the utterance through a descriptor: stores outside are dropped (a negative offset wraps)
            rs_o = __builtin_amdgcn_make_buffer_rsrc(out + ((int64_t)utt * T + f_lo) * (int64_t)Wd, 0, (int32_t)((f_hi - f_lo) * Wd * 4), 0x00020000);

            // a half tile's samples: vectors lane + 64 i of four samples, and the sample before each (through rs_p)
            m512_f4 x4[7];
            float pv[7];
            auto fetch = [&](int s_half) {
                const uint32_t off0 = (uint32_t)(s_half + 4 * lane) * (uint32_t)ESZ;
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    const uint32_t off = off0 + (uint32_t)(256 * i * ESZ);
                    x4[i] = m512t_buf_load4<DTYPE>(rs_x, off);
                    pv[i] = m512_buf_load<DTYPE>(rs_p, (int32_t)(off - (uint32_t)ESZ), 0);   // sample 0 of the utterance wraps: reads 0
                }
            };
            if constexpr (PF) fetch(f_lo * P.S);

            for (int j = 0; j < J; ++j) {
                asm volatile("" : "+v"(lane));   // (keeps lane-derived addresses from being hoisted out of the loops: VGPRs)
                const int g = lane >> 4, n = lane & 15;
                const int t0 = f_lo + 16 * j;
                const int nv = f_hi - t0 < 16 ? f_hi - t0 : 16;     // frames of this tile
                uint32_t Ph[16][2], Pl[16][2];                        // powers of the 16 frames: (g, k1) = lane, values r = 0..3
                uint32_t b0h[4] = {0u, 0u, 0u, 0u}, b0l[4] = {0u, 0u, 0u, 0u};   // column-0 operand: (g, frame n)
                float corr_h[2] = {0.f, 0.f};

                auto half = [&](auto hc_) {
                    constexpr int h = decltype(hc_)::value;
                    const int s_half = (t0 + 8 * h) * P.S;
                    if (8 * h >= nv) {
#pragma unroll
                        for (int fl = 0; fl < 8; ++fl) { Ph[8 * h + fl][0] = Ph[8 * h + fl][1] = Pl[8 * h + fl][0] = Pl[8 * h + fl][1] = 0u; }
                        if constexpr (PF) fetch(s_half + 8 * P.S);   // (keeps the request unconditional: registers)
                        return;
                    }
                    // ------------------------------------------------------------ 1. the half's samples
                    if constexpr (!PF) fetch(s_half);
                    M512T_STAMP(0);
                    float mx = fabsf(pv[0]);
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(x4[i][0]), "v"(x4[i][1]));
                        asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(x4[i][2]), "v"(x4[i][3]));
                    }
                    // wave maximum of non-negative floats = maximum of their bit patterns
                    uint32_t mu = __builtin_bit_cast(uint32_t, mx);
                    mu = max(mu, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mu, 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
                    mu = max(mu, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mu, 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
                    mu = max(mu, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mu, 0x141, 0xf, 0xf, true));   // row_half_mirror
                    mu = max(mu, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mu, 0x140, 0xf, 0xf, true));   // row_mirror
                    M512T_STAMP(1);
                    const uint32_t mbits = max(max((uint32_t)__builtin_amdgcn_readlane((int)mu, 0), (uint32_t)__builtin_amdgcn_readlane((int)mu, 16)),
                                               max((uint32_t)__builtin_amdgcn_readlane((int)mu, 32), (uint32_t)__builtin_amdgcn_readlane((int)mu, 48)));
                    int esc = 0;
                    if (mbits != 0) {
                        const int exq = (int)((mbits >> 23) & 255u) - 127;    // 2^ex <= max < 2^(ex+1); |y| < 2^(ex+2)
                        esc = M512_XBITS - exq - 2;
                        esc = esc > 120 ? 120 : (esc < -120 ? -120 : esc);
                    }
                    // the half's scale rides on the window (a power of two: exact): the image holds plain y = x[n] - c x[n - 1]
                    const float sc = __builtin_bit_cast(float, (uint32_t)(esc + 127) << 23);
                    float wsc[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) wsc[i] = win[i] * sc;
                    corr_h[h] = (float)(2 * esc + M512_WSH);
                    // the sample at index N (the first one past the utterance) has a predecessor but must be zero
                    const bool has_end = s_half + 16 * (7 * HS + 32) > Nsamp;
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        float y[4];
                        y[0] = fmaf(pv[i], ncpre, x4[i][0]);
#pragma unroll
                        for (int e = 1; e < 4; ++e) y[e] = fmaf(x4[i][e - 1], ncpre, x4[i][e]);
                        if (has_end) {
                            const int d = Nsamp - (s_half + 4 * (lane + 64 * i));
#pragma unroll
                            for (int e = 1; e < 4; ++e)
                                if (d == e) y[e] = 0.f;
                        }
                        if (i < 6 || lane < NVEC - 384) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) imgw[e * R + 16 * i] = y[e];
                        }
                    }
                    M512_LDS_FENCE();
                    if constexpr (PF) fetch(s_half + 8 * P.S);   // the next half tile (of this tile or the next): zeros past the utterance
                    M512T_STAMP(2);

                    // ------------------------------------------------------------ 2. the frames of the half, FPB at a time
                    auto frames = [&](auto bc_) {
                        constexpr int fl0 = FPB * decltype(bc_)::value, f0 = 8 * h + fl0;
                        if (f0 >= nv) {
#pragma unroll
                            for (int k = 0; k < FPB; ++k) Ph[f0 + k][0] = Ph[f0 + k][1] = Pl[f0 + k][0] = Pl[f0 + k][1] = 0u;
                            return;
                        }
                        uint32_t ah[FPB][4], al[FPB][4];
#pragma unroll
                        for (int k = 0; k < FPB; ++k) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const m512_f2 v = aB[(HS * (fl0 + k)) / 2 + 4 * q];   // bytes 4 HS fl + 32 q
                                m512t_split2(v[0] * wsc[2 * q], v[1] * wsc[2 * q + 1], ah[k][q], al[k][q]);
                            }
                        }
                        const m512_f4 zero = {0.f, 0.f, 0.f, 0.f};
                        m512_f4 d1[FPB][2];
#pragma unroll
                        for (int k = 0; k < FPB; ++k) d1[k][0] = d1[k][1] = zero;
#pragma unroll
                        for (int p = 0; p < 3; ++p)
#pragma unroll
                            for (int k = 0; k < FPB; ++k)
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    const m512_h8 A = p < 2 ? m512_as_h8(ah[k][0], ah[k][1], ah[k][2], ah[k][3]) : m512_as_h8(al[k][0], al[k][1], al[k][2], al[k][3]);
                                    d1[k][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, f1[u][p == 1 ? 1 : 0], d1[k][u], 0, 0, 0);
                                }
                        // twiddle (column 0: the pair (Y0, Y16) passes: c = 1, s = 0), then (hi, lo): the stage-2 operand
                        uint32_t bh[FPB][4], bl[FPB][4];
#pragma unroll
                        for (int k = 0; k < FPB; ++k) {
                            float y[8];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                y[r] = fmaf(-d1[k][1][r], tws[r], d1[k][0][r] * twc[r]);
                                y[4 + r] = fmaf(d1[k][0][r], tws[r], d1[k][1][r] * twc[r]);
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) m512t_split2(y[2 * q], y[2 * q + 1], bh[k][q], bl[k][q]);
                        }
                        if (n == 0) {   // column 0: the packed real rows k1 = 0 / 16 of these frames, for the tile's own product
#pragma unroll
                            for (int k = 0; k < FPB; ++k) {
                                *reinterpret_cast<m512_u4*>(c0 + (g * 8 + fl0 + k) * 16) = m512_u4{bh[k][0], bh[k][1], bh[k][2], bh[k][3]};
                                *reinterpret_cast<m512_u4*>(c0 + 512 + (g * 8 + fl0 + k) * 16) = m512_u4{bl[k][0], bl[k][1], bl[k][2], bl[k][3]};
                            }
                        }
                        m512_f4 z2[FPB][2];
#pragma unroll
                        for (int k = 0; k < FPB; ++k) z2[k][0] = z2[k][1] = zero;
#pragma unroll
                        for (int p = 0; p < 3; ++p)
#pragma unroll
                            for (int k = 0; k < FPB; ++k)
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    const m512_h8 B = p == 1 ? m512_as_h8(bl[k][0], bl[k][1], bl[k][2], bl[k][3]) : m512_as_h8(bh[k][0], bh[k][1], bh[k][2], bh[k][3]);
                                    z2[k][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f2[u][p == 2 ? 1 : 0], B, z2[k][u], 0, 0, 0);
                                }
#pragma unroll
                        for (int k = 0; k < FPB; ++k) {
                            float pw[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) pw[r] = fmaf(z2[k][0][r], z2[k][0][r], z2[k][1][r] * z2[k][1][r]);
                            m512t_split_bf16(pw[0], pw[1], Ph[f0 + k][0], Pl[f0 + k][0]);
                            m512t_split_bf16(pw[2], pw[3], Ph[f0 + k][1], Pl[f0 + k][1]);
                        }
                    };
                    m512_static_for<0, 8 / FPB>(frames);
                    M512_LDS_FENCE();
                    M512T_STAMP(3);
                    if ((n >> 3) == h) {   // the half's column-0 packets -> operand of lane (g, frame n)
                        const m512_u4 vh = *reinterpret_cast<const m512_u4*>(c0 + (g * 8 + (n & 7)) * 16);
                        const m512_u4 vl = *reinterpret_cast<const m512_u4*>(c0 + 512 + (g * 8 + (n & 7)) * 16);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { b0h[i] = vh[i]; b0l[i] = vl[i]; }
                    }
                    M512_LDS_FENCE();
                };
                half(std::integral_constant<int, 0>{});
                half(std::integral_constant<int, 1>{});

                // ---------------------------------------------------------------- 3. bins 16 m (column 0 of the 16 frames)
                uint32_t p0h[2], p0l[2];
                {
                    m512_h8 m0[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int hl = 0; hl < 2; ++hl) m0[u][hl] = reinterpret_cast<const m512_h8*>(sM0 + (u * 2 + hl) * 1024)[lane];
                    const m512_h8 Bh = m512_as_h8(b0h[0], b0h[1], b0h[2], b0h[3]), Bl = m512_as_h8(b0l[0], b0l[1], b0l[2], b0l[3]);
                    const m512_f4 zero = {0.f, 0.f, 0.f, 0.f};
                    const m512_f4 zre = m512_mma3(m0[0][0], m0[0][1], Bh, Bl, zero);
                    const m512_f4 zim = m512_mma3(m0[1][0], m0[1][1], Bh, Bl, zero);
                    float pw[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pw[r] = fmaf(zre[r], zre[r], zim[r] * zim[r]);
                    m512t_split_bf16(pw[0], pw[1], p0h[0], p0l[0]);
                    m512t_split_bf16(pw[2], pw[3], p0h[1], p0l[1]);
                }
                M512T_STAMP(4);
                // the DCT matrix (single use per tile) comes from L2 while the mel products run
                m512_h8 dmt[2][2];
#pragma unroll
                for (int u = 0; u < (NMT > 2 ? 2 : 1); ++u)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl)
                        dmt[u][hl] = reinterpret_cast<const m512_h8*>(P.tables + P.lay.off_dm + (u * 2 + hl) * 1024)[lane];

                // ---------------------------------------------------------------- 4. mel: the powers onto the N side
                // exchange slot of packet pair q = p >> 1 and frame f: 16 bytes at (16 q + (f ^ (q & 15))) * 16
                m512_f4 eacc[NMT];
#pragma unroll
                for (int t = 0; t < NMT; ++t) eacc[t] = m512_f4{0.f, 0.f, 0.f, 0.f};
                const uint32_t wr_base = (uint32_t)(((lane >> 1) * 16 + ((lane >> 1) & 15)) * 16 + 8 * (lane & 1));
                // both halves of the powers go through the buffer first (the registers of the powers become the operands'),
                // then every mel block is read ONCE and multiplies the hi and the lo operand of its step
                m512_b8 Bs[2][8];
                auto exchange = [&](auto partc_) {
                    constexpr int lo_part = decltype(partc_)::value;
#pragma unroll
                    for (int f = 0; f < 16; ++f) {
                        const m512_u2 v = lo_part ? m512_u2{Pl[f][0], Pl[f][1]} : m512_u2{Ph[f][0], Ph[f][1]};
                        *reinterpret_cast<m512_u2*>(ex + (wr_base ^ (uint32_t)(16 * f))) = v;
                    }
                    M512_LDS_FENCE();
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        if (m512t_pattern_has(PAT, NMT, s >> 1, 0) || m512t_pattern_has(PAT, NMT, s >> 1, 1) || m512t_pattern_has(PAT, NMT, s >> 1, 2)) {
                            const int q = 4 * s + g;
                            Bs[lo_part][s] = *reinterpret_cast<const m512_b8*>(ex + (q * 16 + (n ^ (q & 15))) * 16);
                        }
                    }
                    M512_LDS_FENCE();
                };
                exchange(std::integral_constant<int, 0>{});
                exchange(std::integral_constant<int, 1>{});
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int t = 0; t < NMT; ++t) {
                        if (m512t_pattern_has(PAT, NMT, s >> 1, t)) {
                            const int bi = m512t_block_index<NMT, PAT>(s, t);
                            const m512_b8 wh = reinterpret_cast<const m512_b8*>(sW + bi * 2048)[lane];
                            const m512_b8 wl = reinterpret_cast<const m512_b8*>(sW + bi * 2048 + 1024)[lane];
                            eacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, Bs[0][s], eacc[t], 0, 0, 0);
                            eacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, Bs[1][s], eacc[t], 0, 0, 0);
                            eacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, Bs[0][s], eacc[t], 0, 0, 0);
                        }
                    }
                {
                    const m512_b8 B = m512_as_b8(p0h[0], p0h[1], p0l[0], p0l[1]);
#pragma unroll
                    for (int t = 0; t < NMT; ++t) {
                        const int bi = m512t_block_index<NMT, PAT>(8, t);
                        const m512_b8 wh = reinterpret_cast<const m512_b8*>(sW + bi * 2048)[lane];
                        const m512_b8 wl = reinterpret_cast<const m512_b8*>(sW + bi * 2048 + 1024)[lane];
                        eacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, B, eacc[t], 0, 0, 0);
                        eacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, B, eacc[t], 0, 0, 0);
                    }
                }

                M512T_STAMP(5);
                // ---------------------------------------------------------------- 5. log2, DCT * lifter, correction
                const float corr = n < 8 ? corr_h[0] : corr_h[1];
                const float zval = P.lay.z_log2_eps + corr;
                uint32_t leh[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}}, lel[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
#pragma unroll
                for (int t = 0; t < NMT; ++t) {
                    float le[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) le[i] = eacc[t][i] == 0.f ? zval : __builtin_amdgcn_logf(eacc[t][i]);
                    m512t_split2(le[0], le[1], leh[t][0], lel[t][0]);
                    m512t_split2(le[2], le[3], leh[t][1], lel[t][1]);
                }
                m512_f4 cep = {0.f, 0.f, 0.f, 0.f};
                cep = m512_mma3(dmt[0][0], dmt[0][1], m512_as_h8(leh[0][0], leh[0][1], leh[1][0], leh[1][1]),
                                m512_as_h8(lel[0][0], lel[0][1], lel[1][0], lel[1][1]), cep);
                if constexpr (NMT > 2) {
                    cep = m512_mma3(dmt[1][0], dmt[1][1], m512_as_h8(leh[2][0], leh[2][1], 0u, 0u), m512_as_h8(lel[2][0], lel[2][1], 0u, 0u), cep);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) cep[i] = fmaf(-corr, rowsum[i], cep[i]);

                M512T_STAMP(6);
                // one quad of a row: coefficients 4 cq .. 4 cq + 3 of part `prt` (0 x, 1 delta, 2 delta-delta) of frame f
                auto store_quad = [&](int f, int cq, int prt, m512_f4 v) {
                    const int col = 4 * cq;
                    const int32_t off = ((f - f_lo) * Wd + prt * C + col) * 4;
                    const m512_u4 u = __builtin_bit_cast(m512_u4, v);
                    if (col + 3 < C) {
                        __builtin_amdgcn_raw_buffer_store_b128(u, rs_o, off, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            if (col + i < C) __builtin_amdgcn_raw_buffer_store_b32(u[i], rs_o, off + 4 * i, 0, 0);
                    }
                };
                if constexpr (!ROWS) {
                    if (n < nv) store_quad(t0 + n, g, 0, cep);
                } else {
                    // ------------------------------------------------------------ 6. delta, delta-delta, rows
                    // cb: cepstra of frames t0 - 8 .. t0 + 15 in rows 0 .. 23 (rows 0..7: the previous tile's last eight)
                    const int fo = lane & 15, cq = lane >> 4;
                    *reinterpret_cast<m512_f4*>(cb + (8 + n) * 16 + 4 * g) = cep;
                    M512_LDS_FENCE();
                    if (j == 0 && inner_left && lane < 32) {   // the part's first eight rows: the four rows either side of
                        const int r = lane >> 2, qd = lane & 3;   // the boundary are finished after the barrier
                        *reinterpret_cast<m512_f4*>(head + r * 16 + 4 * qd) = *reinterpret_cast<const m512_f4*>(cb + (8 + r) * 16 + 4 * qd);
                    }
                    const bool first = t0 == 0, last = t0 + 16 >= T;
                    if (!first && !last) {
                        // interior tile: frame t0 - 4 + fo (row 4 + fo) with its whole window inside the utterance; delta of
                        // delta written out as ONE nine-tap filter (coefficients = the delta taps convolved with themselves).
                        // The first tile of a part that starts inside the utterance has no history: its rows t0 .. t0 + 3
                        // (and the neighbour's last four) wait for the barrier.
                        m512_f4 c9[9];
#pragma unroll
                        for (int r = 0; r < 9; ++r) c9[r] = *reinterpret_cast<const m512_f4*>(cb + r * 16 + fo * 16 + 4 * cq);
                        m512_f4 d, dd;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if constexpr (ND == 2) {
                                d[i] = (c9[5][i] - c9[3][i] + 2.f * (c9[6][i] - c9[2][i])) * inv;
                                dd[i] = (4.f * ((c9[8][i] + c9[0][i]) + (c9[7][i] + c9[1][i]) - (c9[5][i] + c9[3][i])) + (c9[6][i] + c9[2][i]) - 10.f * c9[4][i]) * (inv * inv);
                            } else {
                                d[i] = (c9[5][i] - c9[3][i]) * inv;
                                dd[i] = ((c9[6][i] + c9[2][i]) - 2.f * c9[4][i]) * (inv * inv);
                            }
                        }
                        const int f = t0 - 4 + fo;
                        if (j > 0 || fo >= 8) {
                            store_quad(f, cq, 0, c9[4]);
                            store_quad(f, cq, 1, d);
                            store_quad(f, cq, 2, dd);
                        }
                    } else {
                        // first and / or last tile of the utterance: the windows are clamped to [0, T - 1] (edge padding of
                        // base.py:73, once for delta and once more for delta of delta); deltas go through db
                        const int dlo = first ? 0 : t0 - 6, dhi = last ? T : t0 + 14;
                        const int olo = first ? 0 : t0 - 4, ohi = last ? T : t0 + 12;
                        for (int f0 = dlo; f0 < dhi; f0 += 16) {
                            const int f = f0 + fo;
                            if (f < dhi) {
                                m512_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int k = 1; k <= ND; ++k) {
                                    const int ra = 8 + min(f + k, T - 1) - t0, rb = 8 + max(f - k, 0) - t0;
                                    const m512_f4 a = *reinterpret_cast<const m512_f4*>(cb + ra * 16 + 4 * cq);
                                    const m512_f4 bq = *reinterpret_cast<const m512_f4*>(cb + rb * 16 + 4 * cq);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) d[i] = fmaf((float)k, a[i] - bq[i], d[i]);
                                }
#pragma unroll
                                for (int i = 0; i < 4; ++i) d[i] *= inv;
                                *reinterpret_cast<m512_f4*>(db + (8 + f - t0) * 16 + 4 * cq) = d;
                            }
                        }
                        M512_LDS_FENCE();
                        for (int f0 = olo; f0 < ohi; f0 += 16) {
                            const int f = f0 + fo;
                            if (f < ohi) {
                                m512_f4 dd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int k = 1; k <= ND; ++k) {
                                    const int ra = 8 + min(f + k, T - 1) - t0, rb = 8 + max(f - k, 0) - t0;
                                    const m512_f4 a = *reinterpret_cast<const m512_f4*>(db + ra * 16 + 4 * cq);
                                    const m512_f4 bq = *reinterpret_cast<const m512_f4*>(db + rb * 16 + 4 * cq);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) dd[i] = fmaf((float)k, a[i] - bq[i], dd[i]);
                                }
#pragma unroll
                                for (int i = 0; i < 4; ++i) dd[i] *= inv;
                                store_quad(f, cq, 0, *reinterpret_cast<const m512_f4*>(cb + (8 + f - t0) * 16 + 4 * cq));
                                store_quad(f, cq, 1, *reinterpret_cast<const m512_f4*>(db + (8 + f - t0) * 16 + 4 * cq));
                                store_quad(f, cq, 2, dd);
                            }
                        }
                    }
                    M512_LDS_FENCE();
                    if (!last && lane < 32) {   // the next tile's history: rows 16..23 -> 0..7
                        const int r = lane >> 2, qd = lane & 3;
                        *reinterpret_cast<m512_f4*>(cb + r * 16 + 4 * qd) = *reinterpret_cast<const m512_f4*>(cb + (16 + r) * 16 + 4 * qd);
                    }
                    M512_LDS_FENCE();
                }
                M512T_STAMP(7);
#ifdef M512T_STAMPS
                stamp_acc_[12] += 1;
#endif
            }
        }
        if constexpr (ROWS) {
            if (parts > 1) {
                // ---- the rows either side of a boundary between two parts: four from each wave's head, four from its tail,
                //      nine-tap windows over its own and its neighbour's cepstra (both parts hold 16 frames or more)
                __syncthreads();
                if (active && lane < 32) {
                    const int fo = lane & 7, cq = lane >> 3;
                    const bool tail = fo >= 4;
                    const int f = tail ? f_hi - 8 + fo : f_lo + fo;
                    if (tail ? inner_right : inner_left) {
                        const float* const cbl = cb - WAVE_BYTES / 4;           // left neighbour's rows (frame f_lo - 1 = its row 23)
                        const float* const hdr = head + WAVE_BYTES / 4;         // right neighbour's first rows
                        m512_f4 c9[9];
#pragma unroll
                        for (int r = 0; r < 9; ++r) {
                            const int x = f - 4 + r;
                            const float* row = x < f_lo ? cbl + (24 + x - f_lo) * 16
                                                        : (x >= f_hi ? hdr + (x - f_hi) * 16 : (tail ? cb + (24 + x - f_hi) * 16 : head + (x - f_lo) * 16));
                            c9[r] = *reinterpret_cast<const m512_f4*>(row + 4 * cq);
                        }
                        m512_f4 d, dd;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if constexpr (ND == 2) {
                                d[i] = (c9[5][i] - c9[3][i] + 2.f * (c9[6][i] - c9[2][i])) * inv;
                                dd[i] = (4.f * ((c9[8][i] + c9[0][i]) + (c9[7][i] + c9[1][i]) - (c9[5][i] + c9[3][i])) + (c9[6][i] + c9[2][i]) - 10.f * c9[4][i]) * (inv * inv);
                            } else {
                                d[i] = (c9[5][i] - c9[3][i]) * inv;
                                dd[i] = ((c9[6][i] + c9[2][i]) - 2.f * c9[4][i]) * (inv * inv);
                            }
                        }
                        const int col = 4 * cq;
#pragma unroll
                        for (int prt = 0; prt < 3; ++prt) {
                            const m512_f4 v = prt == 0 ? c9[4] : (prt == 1 ? d : dd);
                            const int32_t off = ((f - f_lo) * Wd + prt * C + col) * 4;
                            const m512_u4 u = __builtin_bit_cast(m512_u4, v);
                            if (col + 3 < C) {
                                __builtin_amdgcn_raw_buffer_store_b128(u, rs_o, off, 0, 0);
                            } else {
#pragma unroll
                                for (int i = 0; i < 3; ++i)
                                    if (col + i < C) __builtin_amdgcn_raw_buffer_store_b32(u[i], rs_o, off + 4 * i, 0, 0);
                            }
                        }
                    }
                }
                if (round + 1 < n_rounds) __syncthreads();   // the neighbours' rows are read: the next utterance may overwrite them
                M512T_STAMP(8);
            }
        }
    }
#ifdef M512T_STAMPS
    {
        const unsigned int t1_ = m512_clock_t();
        unsigned long long rt1_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1_));
        stamp_acc_[13] = t1_ - stamp_t0_;
        stamp_acc_[14] = (unsigned int)(rt1_ - stamp_rt0_);
        stamp_acc_[15] = 1;
        const int wglob = blockIdx.x * WAVES + wv;
        if (lane == 0 && wglob < 4096)
            for (int i = 0; i < M512T_NSTAMP; ++i) atomicAdd(&m512t_stamp_sum[wglob * M512T_NSTAMP + i], stamp_acc_[i]);
    }
#endif
}

// --------------------------------------------------------------------------------------------- host side
#ifndef M512_KERNEL_ONLY

static inline int mfma512t_plan_init(dsp_plan* p, const dsp_plan_desc* d) {
    p->d_mfmat = nullptr;
    if (d->nfft != 512 || d->nfilt < 1 || d->numcep < 1 || !d->h_dct || !d->h_window) return DSP_OK;
    if (d->frame_step != 160) return DSP_OK;                 // instantiated hop: 10 rows of 16 samples
    std::vector<uint8_t> blob;
    M512TLayout lay;
    const int rc = m512t_build_tables(d->frame_len, d->frame_step, d->nfft, d->nfilt, d->numcep, d->append_energy, d->h_window,
                                      d->h_mel_start, d->h_mel_count, d->h_mel_weights, d->h_dct, blob, lay);
    if (rc != 0) return DSP_OK;                               // not served: the other kernels take the plan
    Mfma512TPlan* mp = new Mfma512TPlan();
    mp->lay = lay;
    if (dsp_table_alloc_copy(reinterpret_cast<void**>(&mp->d_tables), blob.data(), blob.size()) != hipSuccess) { delete mp; return DSP_EHIP; }
    p->d_mfmat = mp;
    return DSP_OK;
}

static inline void mfma512t_plan_free(dsp_plan* p) {
    if (!p->d_mfmat) return;
    Mfma512TPlan* mp = static_cast<Mfma512TPlan*>(p->d_mfmat);
    dsp_table_free(mp->d_tables, p->dry_run);
    delete mp;
    p->d_mfmat = nullptr;
}

// dense batches only: every utterance `uniform_samples` long
static inline bool mfma512t_applicable(const dsp_plan* p, const BatchGeom& bg, int dtype, int delta_n) {
    if (!p->d_mfmat || mfma512_mode() != 2) return false;
    if (bg.uniform_samples <= 0 || bg.seg) return false;
    if (dtype != DSP_WAVE_F32 && dtype != DSP_WAVE_I16) return false;
    if (dtype == DSP_WAVE_I16 && (bg.uniform_samples & 1)) return false;   // dword range checks: an odd int16 utterance would lose its last sample
    if (delta_n < 0 || delta_n > 2) return false;
    if (bg.uniform_samples * 4 >= ((int64_t)1 << 30) || bg.uniform_frames * 64 * 4 >= ((int64_t)1 << 31) || (int64_t)bg.n_utt * 64 >= ((int64_t)1 << 31)) return false;   // 32-bit descriptor ranges and offsets
    return true;
}

#ifndef M512T_FPB
#define M512T_FPB 1
#endif
template <int DTYPE, int NMT, int PAT, int ND>
static int mfma512t_launch_k(M512TParams& P, const void* d_wave, float* d_out, hipStream_t st) {
    constexpr int HS = 10;
    constexpr int NBLK = m512t_block_index<NMT, PAT>(9, 0);
    // three waves per SIMD where the mel blocks leave room for twelve waves' buffers, two otherwise
    constexpr int WAVES = (size_t)NBLK * 2048 + 4096 + (size_t)M512T_WAVES * (8192 + 24 * 64 + 8 * 64) <= 163840 ? M512T_WAVES : 8;
#ifdef M512T_PREFETCH
    constexpr bool PF = M512T_PREFETCH != 0;
#else
    constexpr bool PF = WAVES <= 8;
#endif
    constexpr size_t lds = (size_t)NBLK * 2048 + 4096 + (size_t)WAVES * (8192 + 24 * 64 + 8 * 64);
    static_assert(lds <= 163840, "LDS budget");
    auto kern = mfcc512t_kernel<HS, DTYPE, NMT, PAT, ND, WAVES, M512T_FPB, PF>;
    static std::atomic<unsigned long long> attr_set{0};   // per instantiation, one bit per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DSP_EHIP;
    if (dev >= 64 || !((attr_set.load() >> dev) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess)
            return 1;   // not served on this device: the caller falls through to the vector-pipe kernels
        if (dev < 64) attr_set.fetch_or(1ull << dev);
    }
    // Whole utterances per wave while they fill the chip's wave slots; what is left (or a batch smaller than the chip)
    // goes out in one last round with `parts` waves per utterance: as many as fill the slots, each two tiles or more,
    // dividing the workgroup.
    const int cus = mfma512_device_cus();
    const int64_t slots = (int64_t)cus * WAVES;
    P.rounds_full = (int32_t)(P.n_utt / slots);
    const int64_t rest = P.n_utt - (int64_t)P.rounds_full * slots;
    int grid = P.rounds_full > 0 ? cus : 0;
    P.parts = 0;
    if (rest > 0) {
        int parts = 1;
        for (int c = 2; c <= WAVES; ++c) {
            if (WAVES % c) continue;
            if (rest * c > slots || P.tiles / c < 2) break;
            parts = c;
        }
        P.parts = parts;
        const int G = WAVES / parts;
        const int64_t groups = (rest + G - 1) / G;
        if (grid < groups) grid = (int)groups;   // <= cus: rest * parts <= slots
    }
    kern<<<grid, 64 * WAVES, lds, st>>>(P, d_wave, d_out);
    return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
}

// 0 = launched; 1 = not served (caller falls through); < 0 error
static inline int mfma512t_launch(const dsp_plan* p, const void* d_wave, int dtype, const BatchGeom& bg, int delta_n,
                                  float* d_out, int64_t ld_out, hipStream_t st) {
    const Mfma512TPlan* mp = static_cast<const Mfma512TPlan*>(p->d_mfmat);
    M512TParams P;
    memset(&P, 0, sizeof(P));
    P.tables = mp->d_tables;
    P.lay = mp->lay;
    P.S = p->S; P.C = p->C;
    P.preemph = p->preemph;
    P.delta_n = delta_n;
    int den = 0;
    for (int i = 1; i <= delta_n; ++i) den += i * i;
    P.inv_den = den ? (float)(1.0 / (2.0 * den)) : 0.f;
    P.ld_out = ld_out;
    P.n_utt = bg.n_utt;
    P.samples = (int32_t)bg.uniform_samples;
    P.frames = (int32_t)bg.uniform_frames;
    P.tiles = (int32_t)((bg.uniform_frames + 15) / 16);
    const int nmt = mp->lay.n_mtiles, pat = mp->lay.pattern;
#define M512T_LAUNCH_R(DT_, NMT_, PAT_) \
    do { if (delta_n == 0) return mfma512t_launch_k<DT_, NMT_, PAT_, 0>(P, d_wave, d_out, st); \
         if (delta_n == 1) return mfma512t_launch_k<DT_, NMT_, PAT_, 1>(P, d_wave, d_out, st); \
         return mfma512t_launch_k<DT_, NMT_, PAT_, 2>(P, d_wave, d_out, st); } while (0)
#define M512T_LAUNCH_P(DT_) \
    do { if (nmt <= 2) M512T_LAUNCH_R(DT_, 2, 0); else if (pat == 1) M512T_LAUNCH_R(DT_, 3, 1); else M512T_LAUNCH_R(DT_, 3, 0); } while (0)
    if (dtype == DSP_WAVE_I16) M512T_LAUNCH_P(DSP_WAVE_I16); else M512T_LAUNCH_P(DSP_WAVE_F32);
    return 1;
}
#endif  // M512_KERNEL_ONLY
