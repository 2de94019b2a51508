// NFFT = 512 feature kernel on the matrix pipe (gfx950): waveform -> MFCC (+ delta, delta-delta) rows.
// Replaces, for dense batches, the reference's per-utterance chain  sigproc.preemphasis (sigproc.py:178-185) ->
// framesig * window (sigproc.py:66-98) -> powspec (sigproc.py:136-158) -> fbank / mfcc (base.py:8-32, 60-68) ->
// delta (base.py:70-79), like kernels_fast512.h, but with the DFT, the mel filterbank and the DCT as dense products
// on v_mfma_f32_16x16x32_{f16,bf16} (tables and operand maps: mfma512_tables.h; numerics study:
// tools/mfma_numerics.py, tools/mfma512_emul.py).
//
// Why: the vector-pipe kernel needs 12.5 k lane-ops per frame and is latency bound at 0.23 of the HBM roofline
// (DESIGN 4.3).  Here the butterflies are gone; what the vector pipe still does per frame is the fp16 / bf16
// (hi, lo) splitting of the operands, the power spectrum and the delta rows.
//
// Work unit: a TILE = 16 consecutive frames of ONE utterance; lane l = (frame n = l & 15, group g = l >> 4).
// Two waves per SIMD (256 registers each; the stage-1 results of a tile, 128 packed registers, never leave the register
// file); a wave owns whole ROW RANGES of utterances (whole utterances, or two ranges per utterance -- a long and a
// short one per SIMD -- when the batch has fewer utterances than wave slots), so nothing is exchanged between waves and
// the delta window never leaves the wave.  The kernel is OPT-IN (dsp_debug_use_mfma512 / DSP_MFMA512=1): it does less
// than half the vector work of the default kernel and runs level with it.
//  1. staging: the tile's samples (15 S + L, read ONCE per tile through a bounds-checked buffer descriptor: samples
//     before the utterance and beyond its end read as zero, which IS the reference's zero padding), pre-emphasised,
//     scaled by a power of two from the tile's largest sample, split into fp16 hi + lo and stored TRANSPOSED in LDS:
//     plane n2 holds samples 16 r + n2, so a frame's K operand (n1 = 0..31) is contiguous; the image holds 8 of the
//     16 planes at a time (LDS: 64 KB of stage-1 matrices + 28 KB of mel blocks + 8 KB + 7.5 KB per wave = 160 KB);
//  2. stage 1: per column n2 six MFMAs (2 row tiles x 3 products) -> (re, im) pairs packed to fp16 hi / lo;
//  3. a 4 x 4 transpose across the lane groups (v_permlane32_swap / v_permlane16_swap), so that stage 2 finds the
//     n2 index on the K side;
//  4. stage 2 per DFT row (six MFMAs), power, bf16 hi / lo, mel blocks (only the non-zero 16 x 32 blocks);
//  5. log2, DCT * lifter (six MFMAs), scale correction, cepstra into a 24-row LDS buffer (8 rows of history);
//  6. delta / delta-delta from that buffer (rows trail the computation by 4 frames; interior tiles: delta-delta as one
//     nine-tap filter), rows stored through a bounds-checked descriptor over the range's rows.
#pragma once

#include "dsp_common.h"
#include "mfma512_tables.h"

#include <type_traits>

typedef _Float16 m512_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 m512_b8 __attribute__((ext_vector_type(8)));
typedef float m512_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t m512_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t m512_u2 __attribute__((ext_vector_type(2)));
typedef _Float16 m512_h2 __attribute__((ext_vector_type(2)));
typedef float m512_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 m512_b2 __attribute__((ext_vector_type(2)));

#define M512_WAVES 8
#ifndef M512_AHEAD
#define M512_AHEAD 1          // stage 1: operand columns in flight (2: +1.5 %, 6 spilled registers)
#endif
#define M512_NBUF (M512_AHEAD + 1)
// Lanes of a wave hand data to each other through LDS (the LDS queue of a wave is in order).  hipcc reasons per
// lane: a load whose address it can prove different from an earlier store's IN THE SAME LANE may be hoisted above
// that store -- this keeps program order at the hand-over points.
#define M512_LDS_FENCE() asm volatile("" ::: "memory")

struct M512Params {
    const uint8_t* tables;
    M512Layout lay;
    int32_t L, S, C, append_energy;
    float preemph;
    int32_t delta_n;       // 0: cepstra only ([sum T, ld_out]); 1, 2: rows [sum T, 3 C]
    float inv_den;
    int64_t ld_out;
    int32_t n_utt;
    int64_t samples;       // per utterance
    int64_t frames;        // per utterance
    int32_t stagger;       // steps of 512 cycles between the start of consecutive waves of a workgroup
    int32_t items_q, items_r;   // n_utt * splits = items_q * waves + items_r
    int32_t splits;        // work items per utterance (row ranges of equal size): > 1 when the batch has fewer utterances than waves
};

struct Mfma512Plan {
    uint8_t* d_tables;
    M512Layout lay;
};

__device__ __forceinline__ void m512_split_f16(float a, float b, uint32_t& hi, uint32_t& lo) {
    // hi = (f16(a), f16(b)); lo = (f16(a - hi.a), f16(b - hi.b)); the residuals are exact in fp32.
    // v_fma_mixlo/hi_f16 read the fp16 half directly: 3 instructions per pair (hipcc needs 6).
    // No reader of these registers may be an MFMA within two issue slots (the callers pack many pairs first).
    // The conversion is compiler-visible (it is the first reader of an MFMA result or of a load: hipcc pads the
    // hazard and waits for the data); the two mix instructions depend on it.
    uint32_t l;
    const m512_f2 ab = {a, b};
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(ab, m512_h2));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
    hi = h;
    lo = l;
}

// Diagnostic build (-DM512_STAMPS, tools/kbench_m512.py): per-phase shader-clock sums, one slot per wave.
#ifdef M512_STAMPS
#define M512_NSTAMP 16
__device__ unsigned int m512_stamp_sum[M512_NSTAMP * 2048];
__device__ __forceinline__ unsigned int m512_clock() {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return (unsigned int)t;
}
#define M512_STAMP(i) do { const unsigned int now_ = m512_clock(); stamp_acc_[i] += now_ - stamp_prev_; stamp_prev_ = m512_clock(); } while (0)
#define M512_PIN(x) asm volatile("" : "+v"(x))
#else
#define M512_STAMP(i) do {} while (0)
#define M512_PIN(x) do {} while (0)
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void m512_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        m512_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ m512_h8 m512_as_h8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    m512_u4 u = {a, b, c, d};
    return __builtin_bit_cast(m512_h8, u);
}
__device__ __forceinline__ m512_b8 m512_as_b8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    m512_u4 u = {a, b, c, d};
    return __builtin_bit_cast(m512_b8, u);
}

__device__ __forceinline__ m512_f4 m512_mma3(m512_h8 ah, m512_h8 al, m512_h8 bh, m512_h8 bl, m512_f4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    return c;
}

template <int DTYPE>
__device__ __forceinline__ float m512_buf_load(__amdgpu_buffer_rsrc_t rs, int32_t voff, int32_t soff) {
    if constexpr (DTYPE == DSP_WAVE_I16) {
        return (float)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0);
    } else {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    }
}

// Eight consecutive samples as floats (two 16-byte loads of fp32, one of int16), bounds-checked per dword.
template <int DTYPE>
__device__ __forceinline__ void m512_buf_load8(__amdgpu_buffer_rsrc_t rs, uint32_t voff, float* dst) {
    if constexpr (DTYPE == DSP_WAVE_I16) {
        const m512_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int32_t)voff, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[2 * k] = (float)(int16_t)(v[k] & 0xffffu);
            dst[2 * k + 1] = (float)(int16_t)(v[k] >> 16);
        }
    } else {
        // (whole-vector casts: an element-wise bit_cast of the builtin's result makes hipcc 7.2 narrow the load to one dword)
        const m512_f4 a = __builtin_bit_cast(m512_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int32_t)voff, 0, 0));
        const m512_f4 b = __builtin_bit_cast(m512_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int32_t)(voff + 16u), 0, 0));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst[k] = a[k];
            dst[4 + k] = b[k];
        }
    }
}

// Which mel blocks exist is a compile-time pattern (a run-time test per block costs register copies at every join):
// two row tiles: all 16 blocks; three row tiles: tile 1 (which carries the energy row) in every step, tile 0 only for
// the lowest quarter of the spectrum (steps 0, 1), tile 2 only for the upper half (steps 4..7): 14 blocks.  The table
// builder refuses a filterbank with weight outside the pattern (mfma512_tables.h).
template <int NMT>
__host__ __device__ constexpr bool m512_has_block(int step, int tile) {
    return NMT <= 2 ? true : (tile == 1 || (tile == 0 && step < 2) || (tile == 2 && step >= 4));
}
template <int NMT>
__host__ __device__ constexpr int m512_block_index(int step, int tile) {
    int k = 0;
    for (int st = 0; st < 8; ++st)
        for (int t = 0; t < NMT; ++t) {
            if (st == step && t == tile) return k;
            if (m512_has_block<NMT>(st, t)) ++k;
        }
    return k;
}

// The blocks of mel steps 2 ip + h (the steps whose operands an octet of slots completes), in multiplication order.
template <int NMT>
__host__ __device__ constexpr int m512_octet_blocks(int h) {
    int c = 0;
    for (int ip = 0; ip < 4; ++ip)
        for (int t = 0; t < NMT; ++t)
            if (m512_has_block<NMT>(2 * ip + h, t)) ++c;
    return c;
}
template <int NMT>
__host__ __device__ constexpr int m512_octet_block(int h, int k) {   // -> ip * 4 + tile of the k-th block
    int c = 0;
    for (int ip = 0; ip < 4; ++ip)
        for (int t = 0; t < NMT; ++t)
            if (m512_has_block<NMT>(2 * ip + h, t)) {
                if (c == k) return ip * 4 + t;
                ++c;
            }
    return 0;
}

// HS: hop in rows of 16 samples (10 for a 160-sample hop).  WAVES: 8 (two per SIMD, 256 registers each) or 4.
// The image of a tile holds 8 of the 16 sample planes at a time (n2 = 0..7, then 8..15), rows 0..191.
template <int HS, int DTYPE, int NMT, int ND, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void mfcc512m_kernel(M512Params P, const void* __restrict__ wave,
                                                                          float* __restrict__ out) {
    constexpr int PS = 192;                       // rows per plane (15 HS + 32 <= 192)
    static_assert(15 * HS + 32 <= PS, "hop too long for the image");
    constexpr int IMG_BYTES = 8 * PS * 2;         // one half (8 planes) of fp16
    constexpr int CB_ROWS = 24;
    constexpr int WAVE_BYTES = 2 * IMG_BYTES + CB_ROWS * 64;
    constexpr int ESZ = DTYPE == DSP_WAVE_I16 ? 2 : 4;
    constexpr bool ROWS = ND > 0;                 // ND: delta window (base.py:70-79), 0 = cepstra only
    constexpr int NWB = m512_octet_blocks<NMT>(0) + m512_octet_blocks<NMT>(1);
    // the single-use matrices (stage 2 of slot 0, DCT) live in what LDS is left: 4 KB always, 4 more when they fit
    constexpr bool DM_LDS = 65536 + NWB * 2048 + 8192 + WAVES * WAVE_BYTES <= 163840;
    constexpr int XTRA = DM_LDS ? 8192 : 4096;
    extern __shared__ __attribute__((aligned(16))) uint8_t m512_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* const sA1 = m512_smem;
    uint8_t* const sW = m512_smem + 65536;
    uint8_t* const sA2p = m512_smem + 65536 + NWB * 2048;
    uint8_t* const sDm = sA2p + 4096;
    uint8_t* const sWave = m512_smem + 65536 + NWB * 2048 + XTRA + wv * WAVE_BYTES;
    // the image interleaves the two fp16 parts: 8 bytes per ROW PAIR (2 r, 2 r + 1) = [hi(2 r), hi(2 r + 1), lo(2 r), lo(2 r + 1)],
    // 96 pairs = 768 bytes per plane -- a frame's K octet is four 8-byte aligned ds_read_b64 (hi dword, lo dword) instead
    // of eight ds_read_b32, and a staged row pair one ds_write_b64 instead of two ds_write_b32
    constexpr int PLANE = PS * 4;
    uint8_t* const img = sWave;
    float* const cb = reinterpret_cast<float*>(sWave + 2 * IMG_BYTES);   // cepstra of frames t0 - 8 .. t0 + 15
    float* const db = reinterpret_cast<float*>(img);                      // deltas (edge tiles only): over the image,
                                                                          // dead between stage 1 and the next staging
    const int g = lane >> 4, n = lane & 15;
    const int C = P.C;
    const int T = (int)P.frames;
    const int splits = P.splits;

    // ---- this wave's work items: (utterance, part) -> rows [o_lo, o_hi) of the utterance
    const int nw_total = gridDim.x * WAVES, wglob = blockIdx.x * WAVES + wv;
    const int64_t n_items = (int64_t)P.n_utt * splits;
    // exactly one item per wave and two items per utterance: pair the ranges inside the workgroup (see open_item)
    const bool paired = WAVES == 8 && splits == 2 && n_items == (int64_t)nw_total;
    const int it_base = blockIdx.x * WAVES;
    // contiguous runs: P.items_q items per wave and P.items_r waves with one more, spread evenly over the grid
    // (n_items = items_q * waves + items_r from the host; a remainder dealt to the FIRST waves would load whole CUs)
    const uint32_t ex_lo = (uint32_t)wglob * (uint32_t)P.items_r / (uint32_t)nw_total;
    const uint32_t ex_hi = (uint32_t)(wglob + 1) * (uint32_t)P.items_r / (uint32_t)nw_total;
    const int it_lo = paired ? it_base + wv : wglob * P.items_q + (int)ex_lo;
    const int it_hi = paired ? it_lo + 1 : (wglob + 1) * P.items_q + (int)ex_hi;

    // raw samples of one tile: half h (planes 8 h .. 8 h + 7): rows 2 l, 2 l + 1 (xa) and row 128 + l (xb) of the lane
    float xa[2][16], xb[2][8], pa[2][2], pb[2];
    __amdgpu_buffer_rsrc_t rs_x, rs_p;
    auto fetch = [&](int t0s) {   // t0s: first frame of the tile
        // the whole offset travels in the VGPR / immediate (the descriptor's range check covers those two only)
        const uint32_t base = (uint32_t)(t0s * P.S) * (uint32_t)ESZ;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t oa = base + (uint32_t)((32 * lane + 8 * h) * ESZ), ob = base + (uint32_t)((16 * (128 + lane) + 8 * h) * ESZ);
            m512_buf_load8<DTYPE>(rs_x, oa, &xa[h][0]);
            m512_buf_load8<DTYPE>(rs_x, oa + (uint32_t)(16 * ESZ), &xa[h][8]);
            m512_buf_load8<DTYPE>(rs_x, ob, &xb[h][0]);
            pa[h][0] = m512_buf_load<DTYPE>(rs_p, (int32_t)(oa - (uint32_t)ESZ), 0);   // sample 0 of the utterance wraps: reads 0
            pa[h][1] = m512_buf_load<DTYPE>(rs_p, (int32_t)(oa + (uint32_t)(15 * ESZ)), 0);
            pb[h] = m512_buf_load<DTYPE>(rs_p, (int32_t)(ob - (uint32_t)ESZ), 0);
        }
    };
    auto open_item = [&](int item, int& utt, int& o_lo, int& o_hi, int& c_lo, int& J) {
        int part;
        if (paired) {   // two ranges per utterance, one range per wave: waves w and w + 4 (one SIMD) take a long and a short one
            const int wl = item - it_base;            // == wave index in the workgroup
            utt = (it_base >> 1) + (wl & 3);
            part = wl >> 2;
        } else {
            utt = item / splits;
            part = item - utt * splits;
        }
        // range boundaries at 16 k - 4: the range before ends with a full tile, and the tiles of both are as few as can be
        auto bound = [&](int i) -> int {
            if (i <= 0) return 0;
            if (i >= splits) return T;
            int b = ((int)(((int64_t)T * i / splits + 4 + 8) >> 4) << 4) - 4;
            return b < 0 ? 0 : (b > T ? T : b);
        };
        o_lo = bound(part);
        o_hi = bound(part + 1);
        c_lo = o_lo >= 8 ? o_lo - 8 : 0;                        // eight frames of history for the delta windows
        const int c_hi = o_hi + 4 < T ? o_hi + 4 : T;
        J = (c_hi - c_lo + 15) >> 4;
        const uint8_t* ubase = reinterpret_cast<const uint8_t*>(wave) + (int64_t)utt * P.samples * ESZ;
        // x[u]: u < N; x[u - 1] through rs_p: 1 <= u <= N - 1.  Outside both read as zero: y[0] = x[0], and a zero tail
        // AFTER pre-emphasis (sigproc.py:79-91 pads y, not x)
        rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(ubase), 0, (int32_t)(P.samples * ESZ), 0x00020000);
        rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(ubase), 0, (int32_t)((P.samples - 1) * ESZ), 0x00020000);
    };
    int utt = 0, o_lo = 0, o_hi = 0, c_lo = 0, J = 0;

    // ---- tables -> LDS in one flat pass (A1 | mel blocks | slot-0 matrix, DCT): every load is in flight before the
    //      first store; the wave's own region zeroed once (every image row is rewritten by every staging pass)
    {
        constexpr int TOTAL = 65536 + NWB * 2048 + XTRA, NK = (TOTAL / 16 + 64 * WAVES - 1) / (64 * WAVES);
        m512_u4 tv[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int o = (tid + 64 * WAVES * k) * 16;
            const uint8_t* src = o < 65536 ? P.tables + P.lay.off_a1 + o
                                           : (o < 65536 + NWB * 2048 ? P.tables + P.lay.off_w + (o - 65536)
                                                                     : P.tables + P.lay.off_a2p + (o - 65536 - NWB * 2048));
            if (o < TOTAL) tv[k] = *reinterpret_cast<const m512_u4*>(src);
        }
        m512_u4* z = reinterpret_cast<m512_u4*>(sWave);
        const m512_u4 zero = {0, 0, 0, 0};
        for (int i = lane; i < WAVE_BYTES / 16; i += 64) z[i] = zero;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int o = (tid + 64 * WAVES * k) * 16;
            if (o < TOTAL) *reinterpret_cast<m512_u4*>(m512_smem + o) = tv[k];
        }
    }
    // ---- register-resident: the stage-2 matrix of DFT rows 1..15 and the row sums of the DCT
    m512_h8 a2[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
            a2[u][hl] = reinterpret_cast<const m512_h8*>(P.tables + P.lay.off_a2 + (u * 2 + hl) * 1024)[lane];
    float rowsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowsum[i] = reinterpret_cast<const float*>(P.tables + P.lay.off_rowsum)[4 * g + i];
    __syncthreads();

    // the lane's K octet of plane 0 as an LDS pointer: constant offsets fold into the ds_read_b64's offset field, and
    // volatile keeps hipcc from pairing neighbours into ds_read2_b64 (twice the LDS cycles per byte); a frame starts on
    // a row-pair boundary (hop 160 = 10 rows per plane), i.e. 8-byte aligned in the interleaved image -- not 16
    const uint32_t b_loff = 2u * (uint32_t)(HS * n + 8 * g);   // 2 bytes per row of ONE part: the image has twice that
    typedef __attribute__((address_space(3))) const volatile m512_u2* lds_cvu64;
    const lds_cvu64 bB = reinterpret_cast<lds_cvu64>((uint32_t)reinterpret_cast<uintptr_t>(img + 2 * b_loff));
    const float cpre = P.preemph;

#ifdef M512_STAMPS
    unsigned int stamp_acc_[M512_NSTAMP] = {0}, stamp_prev_ = m512_clock();
    const unsigned int stamp_t0_ = stamp_prev_;
    unsigned long long stamp_rt0_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_rt0_));
#endif
    // the waves of a SIMD pair / of a CU start apart, so that they do not run the LDS-heavy stage 1 in lockstep
    for (int w = 0; w < wv * P.stagger; ++w) __builtin_amdgcn_s_sleep(8);   // 8 x 64 cycles per step

    for (int item = it_lo; item < it_hi; ++item) {
        open_item(item, utt, o_lo, o_hi, c_lo, J);
        const int64_t row0 = (int64_t)utt * T;                        // first output row of the utterance
        // rows [o_lo, o_hi) of the utterance through a descriptor: stores outside are dropped (a negative offset wraps)
        const int Wd = ROWS ? 3 * C : (int)P.ld_out;
        const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out + (row0 + o_lo) * (int64_t)Wd, 0,
                                                                               (int32_t)((o_hi - o_lo) * Wd * 4), 0x00020000);
        const int Nsamp = (int)P.samples;

        for (int j = 0; j < J; ++j) {
            const int t0 = c_lo + 16 * j;
            // Two waves per SIMD (256 registers each): the 54 registers of a tile's raw samples cannot stay live through
            // stage 2, so every tile requests its samples here and the other wave of the SIMD computes meanwhile (any
            // later request point spills, and a spill reload waits for the outstanding sample loads).  One wave per
            // SIMD (512 registers): the next tile's samples are requested right after stage 1, below.
            if (WAVES == 8 || j == 0) fetch(t0);
            M512_STAMP(0);
            // ---------------------------------------------------------------- 1. scale of the tile
            float mx = fmaxf(fabsf(pa[0][0]), fabsf(pa[1][0]));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(xa[h][e]), "v"(xa[h][e + 1]));
#pragma unroll
                for (int e = 0; e < 8; e += 2) asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(xb[h][e]), "v"(xb[h][e + 1]));
            }
            // wave maximum of non-negative floats = maximum of their bit patterns: four DPP steps inside each row of 16
            // lanes, then the four rows on the scalar side
            uint32_t mu = __builtin_bit_cast(uint32_t, mx);
            mu = max(mu, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mu, 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
            mu = max(mu, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mu, 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
            mu = max(mu, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mu, 0x141, 0xf, 0xf, true));   // row_half_mirror
            mu = max(mu, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mu, 0x140, 0xf, 0xf, true));   // row_mirror
            const uint32_t mbits = max(max((uint32_t)__builtin_amdgcn_readlane((int)mu, 0), (uint32_t)__builtin_amdgcn_readlane((int)mu, 16)),
                                       max((uint32_t)__builtin_amdgcn_readlane((int)mu, 32), (uint32_t)__builtin_amdgcn_readlane((int)mu, 48)));
            int esc = 0;
            if (mbits != 0) {
                int ex = (int)((mbits >> 23) & 255u) - 127;           // 2^ex <= max < 2^(ex+1); |y| < 2^(ex+2)
                esc = M512_XBITS - ex - 2;
                esc = esc > 120 ? 120 : (esc < -120 ? -120 : esc);
            }
            const float sc = __builtin_bit_cast(float, (uint32_t)(esc + 127) << 23);
            const float csc = -cpre * sc;
            const float corr = (float)(2 * esc + M512_WSH);
            // the sample at index N (the first one past the utterance) has a predecessor but must be zero: tiles that
            // reach the end of the utterance clear it
            const bool has_end = t0 * P.S + 16 * PS > Nsamp;

            // ---------------------------------------------------------------- 2. staging + stage 1, half by half
            uint32_t Rh[4][4][4], Rl[4][4][4];                        // [n2 >> 2][n2 & 3][i]: (re, im) of row 4 g + i
            m512_f4 accp[2];
            uint32_t bh[M512_NBUF][4], bl[M512_NBUF][4];
            m512_h8 ah[M512_NBUF][2], al[M512_NBUF][2];
            auto stage_half = [&](auto hc_) {
                constexpr int h = decltype(hc_)::value;
                // rows 2 l and 2 l + 1: the two rows of a plane share a dword
                float ya[16], yb[8];
#pragma unroll
                for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        ya[8 * r2 + e] = fmaf(e == 0 ? pa[h][r2] : xa[h][8 * r2 + e - 1], csc, xa[h][8 * r2 + e] * sc);
#pragma unroll
                for (int e = 0; e < 8; ++e) yb[e] = fmaf(e == 0 ? pb[h] : xb[h][e - 1], csc, xb[h][e] * sc);
                if (has_end) {
                    const int da = Nsamp - (t0 * P.S + 32 * lane + 8 * h), dbb = Nsamp - (t0 * P.S + 16 * (128 + lane) + 8 * h);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (da == e) ya[e] = 0.f;
                        if (da == 16 + e) ya[8 + e] = 0.f;
                        if (dbb == e) yb[e] = 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    uint32_t hh, ll;
                    m512_split_f16(ya[e], ya[8 + e], hh, ll);
                    *reinterpret_cast<m512_u2*>(img + e * PLANE + 8 * lane) = m512_u2{hh, ll};
                }
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    uint32_t hh, ll;
                    m512_split_f16(yb[e], yb[e + 1], hh, ll);
                    // row 128 + l: pair 64 + (l >> 1), half l & 1
                    uint8_t* const q0 = img + e * PLANE + 8 * (64 + (lane >> 1)) + 2 * (lane & 1);
                    *reinterpret_cast<uint16_t*>(q0) = (uint16_t)hh;
                    *reinterpret_cast<uint16_t*>(q0 + PLANE) = (uint16_t)(hh >> 16);
                    *reinterpret_cast<uint16_t*>(q0 + 4) = (uint16_t)ll;
                    *reinterpret_cast<uint16_t*>(q0 + PLANE + 4) = (uint16_t)(ll >> 16);
                }
            };
            // the operands of column n2 + M512_AHEAD are requested while the products of column n2 run (the compiler pulls
            // a column's products up to one column forward, so one column ahead in the source is less than an LDS
            // latency in the instruction stream)
            auto load_ops = [&](auto nc_) {
                constexpr int n2 = decltype(nc_)::value;
                constexpr int sl = n2 % M512_NBUF, pl = n2 & 7;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const m512_u2 hl = bB[(pl * PLANE + 8 * d) / 8];
                    bh[sl][d] = hl[0];
                    bl[sl][d] = hl[1];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[sl][t] = reinterpret_cast<const m512_h8*>(sA1 + ((n2 * 2 + t) * 2 + 0) * 1024)[lane];
                    al[sl][t] = reinterpret_cast<const m512_h8*>(sA1 + ((n2 * 2 + t) * 2 + 1) * 1024)[lane];
                }
            };
            auto column = [&](auto nc_) {
                constexpr int n2 = decltype(nc_)::value;
                constexpr int sl = n2 % M512_NBUF;
                if constexpr ((n2 & 7) + M512_AHEAD <= 7) load_ops(std::integral_constant<int, ((n2 & 7) + M512_AHEAD <= 7 ? n2 + M512_AHEAD : 0)>{});
                const m512_h8 Bh = m512_as_h8(bh[sl][0], bh[sl][1], bh[sl][2], bh[sl][3]);
                const m512_h8 Bl = m512_as_h8(bl[sl][0], bl[sl][1], bl[sl][2], bl[sl][3]);
                const m512_f4 zero = {0.f, 0.f, 0.f, 0.f};
                const m512_f4 acc0 = m512_mma3(ah[sl][0], al[sl][0], Bh, Bl, zero);
                const m512_f4 acc1 = m512_mma3(ah[sl][1], al[sl][1], Bh, Bl, zero);
                if constexpr (n2 > 0) {   // the previous column's results are packed while this column's products run
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        m512_split_f16(accp[0][i], accp[1][i], Rh[(n2 - 1) >> 2][(n2 - 1) & 3][i], Rl[(n2 - 1) >> 2][(n2 - 1) & 3][i]);
                }
                accp[0] = acc0;
                accp[1] = acc1;
            };
            stage_half(std::integral_constant<int, 0>{});
            M512_LDS_FENCE();
            M512_STAMP(1);
            load_ops(std::integral_constant<int, 0>{});
            if constexpr (M512_AHEAD > 1) load_ops(std::integral_constant<int, 1>{});
            m512_static_for<0, 8>(column);
            M512_LDS_FENCE();
            stage_half(std::integral_constant<int, 1>{});
            M512_LDS_FENCE();
            load_ops(std::integral_constant<int, 8>{});
            if constexpr (M512_AHEAD > 1) load_ops(std::integral_constant<int, 9>{});
            m512_static_for<8, 16>(column);
#pragma unroll
            for (int i = 0; i < 4; ++i) m512_split_f16(accp[0][i], accp[1][i], Rh[3][3][i], Rl[3][3][i]);
            M512_LDS_FENCE();
#ifdef M512_STAMPS
#pragma unroll
            for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 4; ++b_)
#pragma unroll
                    for (int c_ = 0; c_ < 4; ++c_) { M512_PIN(Rh[a_][b_][c_]); M512_PIN(Rl[a_][b_][c_]); }
#endif
            M512_STAMP(2);
            if constexpr (WAVES == 4) {
                if (j + 1 < J) fetch(t0 + 16);
            }

            // ---------------------------------------------------------------- 3. transpose n2-block <-> lane group
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#define M512_SWAP32(a, b) { auto s_ = __builtin_amdgcn_permlane32_swap(a, b, false, false); a = s_[0]; b = s_[1]; }
#define M512_SWAP16(a, b) { auto s_ = __builtin_amdgcn_permlane16_swap(a, b, false, false); a = s_[0]; b = s_[1]; }
                    M512_SWAP32(Rh[0][qq][i], Rh[2][qq][i]);
                    M512_SWAP32(Rh[1][qq][i], Rh[3][qq][i]);
                    M512_SWAP16(Rh[0][qq][i], Rh[1][qq][i]);
                    M512_SWAP16(Rh[2][qq][i], Rh[3][qq][i]);
                    M512_SWAP32(Rl[0][qq][i], Rl[2][qq][i]);
                    M512_SWAP32(Rl[1][qq][i], Rl[3][qq][i]);
                    M512_SWAP16(Rl[0][qq][i], Rl[1][qq][i]);
                    M512_SWAP16(Rl[2][qq][i], Rl[3][qq][i]);
                }
#ifdef M512_STAMPS
#pragma unroll
            for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < 4; ++b_)
#pragma unroll
                    for (int c_ = 0; c_ < 4; ++c_) { M512_PIN(Rh[a_][b_][c_]); M512_PIN(Rl[a_][b_][c_]); }
#endif
            M512_STAMP(3);
            // now R?[sig][qq][i] in lane group gam = (re, im) of slot 4 sig + i at n2 = 4 gam + qq

            // ---------------------------------------------------------------- 4. stage 2, power, mel
            m512_f4 eacc[NMT];
#pragma unroll
            for (int t = 0; t < NMT; ++t) eacc[t] = m512_f4{0.f, 0.f, 0.f, 0.f};
            uint32_t Ph[8][4], Pl[8][4];   // [mel step 2 ip + (s >> 3)][dword (s & 7) >> 1]: powers of slots (s, s + 1) as bf16
            float pwp[4];                   // powers of the even slot of a pair
            m512_f4 rep, imp;               // stage-2 results of the previous slot
            // power + bf16 split of slot s (from rep / imp), and, when an octet of slots is complete, its mel blocks
            auto finish_slot = [&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                float pw[4];
#pragma unroll
                for (int ip = 0; ip < 4; ++ip) pw[ip] = fmaf(rep[ip], rep[ip], imp[ip] * imp[ip]);
                if constexpr ((s & 1) == 0) {
#pragma unroll
                    for (int ip = 0; ip < 4; ++ip) pwp[ip] = pw[ip];
                } else {
#pragma unroll
                    for (int ip = 0; ip < 4; ++ip) {
                        const m512_f2 v = {pwp[ip], pw[ip]};
                        const uint32_t hu = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, m512_b2));
                        const float ha = __builtin_bit_cast(float, hu << 16), hbv = __builtin_bit_cast(float, hu & 0xffff0000u);
                        const m512_f2 r = {v[0] - ha, v[1] - hbv};
                        Ph[2 * ip + (s >> 3)][(s & 7) >> 1] = hu;
                        Pl[2 * ip + (s >> 3)][(s & 7) >> 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, m512_b2));
                    }
                }
                if constexpr ((s & 7) == 7) {
                    constexpr int h = s >> 3;
                    // the blocks of steps 2 ip + h in order; the weights of block k + 1 are read while block k multiplies
                    constexpr int NB = m512_octet_blocks<NMT>(h);
                    m512_b8 wh[2], wl[2];
                    auto load_w = [&](auto kc_) {
                        constexpr int k = decltype(kc_)::value;
                        constexpr int it = m512_octet_block<NMT>(h, k), ip = it >> 2, t = it & 3;
                        constexpr int bi = m512_block_index<NMT>(2 * ip + h, t);
                        wh[k & 1] = reinterpret_cast<const m512_b8*>(sW + bi * 2048)[lane];
                        wl[k & 1] = reinterpret_cast<const m512_b8*>(sW + bi * 2048 + 1024)[lane];
                    };
                    load_w(std::integral_constant<int, 0>{});
                    auto mul_w = [&](auto kc_) {
                        constexpr int k = decltype(kc_)::value;
                        if constexpr (k + 1 < NB) load_w(std::integral_constant<int, (k + 1 < NB ? k + 1 : 0)>{});
                        constexpr int it = m512_octet_block<NMT>(h, k), ip = it >> 2, t = it & 3;
                        constexpr int step = 2 * ip + h;
                        const m512_b8 ph = m512_as_b8(Ph[step][0], Ph[step][1], Ph[step][2], Ph[step][3]);
                        const m512_b8 pl = m512_as_b8(Pl[step][0], Pl[step][1], Pl[step][2], Pl[step][3]);
                        eacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[k & 1], ph, eacc[t], 0, 0, 0);
                        eacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[k & 1], pl, eacc[t], 0, 0, 0);
                        eacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[k & 1], ph, eacc[t], 0, 0, 0);
                    };
                    m512_static_for<0, NB>(mul_w);
                }
            };
            // slot 0 (DFT rows 0 and 16) has its own matrix, read from LDS for this one use
            m512_h8 a2p[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    a2p[u][hl] = reinterpret_cast<const m512_h8*>(sA2p + (u * 2 + hl) * 1024)[lane];
            auto run_slot = [&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                constexpr int sig = s >> 2, i = s & 3;
                const m512_h8 Bh = m512_as_h8(Rh[sig][0][i], Rh[sig][1][i], Rh[sig][2][i], Rh[sig][3][i]);
                const m512_h8 Bl = m512_as_h8(Rl[sig][0][i], Rl[sig][1][i], Rl[sig][2][i], Rl[sig][3][i]);
                const m512_f4 zero = {0.f, 0.f, 0.f, 0.f};
                m512_f4 re, im;
                if constexpr (s == 0) {
                    re = m512_mma3(a2p[0][0], a2p[0][1], Bh, Bl, zero);
                    im = m512_mma3(a2p[1][0], a2p[1][1], Bh, Bl, zero);
                } else {
                    re = m512_mma3(a2[0][0], a2[0][1], Bh, Bl, zero);
                    im = m512_mma3(a2[1][0], a2[1][1], Bh, Bl, zero);
                }
                if constexpr (s > 0) finish_slot(std::integral_constant<int, (s > 0 ? s - 1 : 0)>{});
                rep = re;
                imp = im;
            };
            m512_static_for<0, 16>(run_slot);
            finish_slot(std::integral_constant<int, 15>{});
#ifdef M512_STAMPS
#pragma unroll
            for (int t_ = 0; t_ < NMT; ++t_)
#pragma unroll
                for (int c_ = 0; c_ < 4; ++c_) M512_PIN(eacc[t_][c_]);
#endif
            M512_STAMP(4);

            // ---------------------------------------------------------------- 5. log2, DCT * lifter, correction
            m512_h8 dmt[2][2];
#pragma unroll
            for (int u = 0; u < (NMT > 2 ? 2 : 1); ++u)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    dmt[u][hl] = DM_LDS ? reinterpret_cast<const m512_h8*>(sDm + (u * 2 + hl) * 1024)[lane]
                                        : reinterpret_cast<const m512_h8*>(P.tables + P.lay.off_dm + (u * 2 + hl) * 1024)[lane];
            const float zval = P.lay.z_log2_eps + corr;
            uint32_t leh[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}}, lel[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
#pragma unroll
            for (int t = 0; t < NMT; ++t) {
                float le[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) le[i] = eacc[t][i] == 0.f ? zval : __builtin_amdgcn_logf(eacc[t][i]);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const m512_f2 v = {le[2 * k], le[2 * k + 1]};
                    const m512_h2 hh = __builtin_convertvector(v, m512_h2);
                    const m512_f2 r = {v[0] - (float)hh[0], v[1] - (float)hh[1]};
                    const m512_h2 ll = __builtin_convertvector(r, m512_h2);
                    leh[t][k] = __builtin_bit_cast(uint32_t, hh);
                    lel[t][k] = __builtin_bit_cast(uint32_t, ll);
                }
            }
            m512_f4 cep = {0.f, 0.f, 0.f, 0.f};
            cep = m512_mma3(dmt[0][0], dmt[0][1], m512_as_h8(leh[0][0], leh[0][1], leh[1][0], leh[1][1]),
                            m512_as_h8(lel[0][0], lel[0][1], lel[1][0], lel[1][1]), cep);
            if constexpr (NMT > 2)
                cep = m512_mma3(dmt[1][0], dmt[1][1], m512_as_h8(leh[2][0], leh[2][1], 0u, 0u), m512_as_h8(lel[2][0], lel[2][1], 0u, 0u), cep);
#pragma unroll
            for (int i = 0; i < 4; ++i) cep[i] = fmaf(-corr, rowsum[i], cep[i]);
#ifdef M512_STAMPS
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) M512_PIN(cep[c_]);
#endif
            M512_STAMP(5);

            // one quad of a row: coefficients 4 cq .. 4 cq + 3 of part `part` (0 x, 1 delta, 2 delta-delta) of frame f
            auto store_quad = [&](int f, int cq, int part, m512_f4 v) {
                const int col = 4 * cq;
                const int32_t off = ((f - o_lo) * Wd + part * C + col) * 4;
                const m512_u4 u = __builtin_bit_cast(m512_u4, v);   // whole-vector cast (element-wise bit_casts: see m512_buf_load8)
                if (col + 3 < C) {
                    __builtin_amdgcn_raw_buffer_store_b128(u, rs_o, off, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (col + i < C) __builtin_amdgcn_raw_buffer_store_b32(u[i], rs_o, off + 4 * i, 0, 0);
                }
            };
            if constexpr (!ROWS) {
                // ------------------------------------------------------------ 6a. cepstra only
                if (t0 + n < T) store_quad(t0 + n, g, 0, cep);
            } else {
                // ------------------------------------------------------------ 6b. delta, delta-delta, rows
                // cb: cepstra of frames t0 - 8 .. t0 + 15 in rows 0 .. 23 (rows 0..7: the previous tile's last eight)
                const int fo = lane & 15, cq = lane >> 4;
                const float inv = P.inv_den;
                *reinterpret_cast<m512_f4*>(cb + (8 + n) * 16 + 4 * g) = cep;
                M512_LDS_FENCE();
                const bool first = t0 == 0, last = t0 + 16 >= T;
                if (!first && !last) {
                    // interior tile: frame t0 - 4 + fo (row 4 + fo) with its whole window inside the utterance; delta of
                    // delta written out as ONE nine-tap filter (coefficients = the delta taps convolved with themselves)
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
                    store_quad(f, cq, 0, c9[4]);
                    store_quad(f, cq, 1, d);
                    store_quad(f, cq, 2, dd);
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
            M512_STAMP(6);
#ifdef M512_STAMPS
            stamp_acc_[7] += 1;
#endif
        }
    }
#ifdef M512_STAMPS
    {
        const unsigned int t1_ = m512_clock();
        unsigned long long rt1_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1_));
        stamp_acc_[8] = t1_ - stamp_t0_;
        stamp_acc_[9] = (unsigned int)(rt1_ - stamp_rt0_);
        stamp_acc_[10] = 1;
        if (lane == 0 && wglob < 2048)
            for (int i = 0; i < M512_NSTAMP; ++i) atomicAdd(&m512_stamp_sum[wglob * M512_NSTAMP + i], stamp_acc_[i]);
    }
#endif
}

// --------------------------------------------------------------------------------------------- host side
#ifndef M512_KERNEL_ONLY

// The matrix-pipe kernel is an opt-in path (measured slower than the vector-pipe kernel at this stage):
// DSP_MFMA512=1 in the environment, or dsp_debug_use_mfma512(1) for the calling thread.
thread_local int g_use_mfma512 = -1;   // -1: follow the environment
// 0: vector-pipe kernels; 1: this kernel (16 frames per product); 2: kernels_mfma512t.h (one frame per product)
static inline int mfma512_mode() {
    if (g_use_mfma512 >= 0) return g_use_mfma512;
    static const int env = [] { const char* e = getenv("DSP_MFMA512"); return e && (e[0] == '1' || e[0] == '2') ? e[0] - '0' : 0; }();   // read once, thread-safe
    return env;
}
static inline bool mfma512_enabled() { return mfma512_mode() == 1; }

static inline int mfma512_plan_init(dsp_plan* p, const dsp_plan_desc* d) {
    p->d_mfma = nullptr;
    if (d->nfft != 512 || d->nfilt < 1 || d->numcep < 1 || !d->h_dct || !d->h_window) return DSP_OK;
    if (d->frame_step != 160) return DSP_OK;                 // instantiated hop: 10 rows of 16 samples
    std::vector<uint8_t> blob;
    M512Layout lay;
    const int rc = m512_build_tables(d->frame_len, d->frame_step, d->nfft, d->nfilt, d->numcep, d->append_energy, d->h_window,
                                     d->h_mel_start, d->h_mel_count, d->h_mel_weights, d->h_dct, blob, lay);
    if (rc != 0) return DSP_OK;                               // not served: the other kernels take the plan
    Mfma512Plan* mp = new Mfma512Plan();
    mp->lay = lay;
    if (dsp_table_alloc_copy(reinterpret_cast<void**>(&mp->d_tables), blob.data(), blob.size()) != hipSuccess) { delete mp; return DSP_EHIP; }
    p->d_mfma = mp;
    return DSP_OK;
}

static inline void mfma512_plan_free(dsp_plan* p) {
    if (!p->d_mfma) return;
    Mfma512Plan* mp = static_cast<Mfma512Plan*>(p->d_mfma);
    dsp_table_free(mp->d_tables, p->dry_run);
    delete mp;
    p->d_mfma = nullptr;
}

static inline int mfma512_device_cus() {
    // (one device model per process: the first device asked answers for all)
    static const int cus = [] {
        int dev = 0, n = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n > 0 ? n : 256;
    }();
    return cus;
}

// dense batches only: every utterance `uniform_samples` long, enough frames to give every wave slot of the chip a tile
static inline bool mfma512_applicable(const dsp_plan* p, const BatchGeom& bg, int dtype, int delta_n) {
    if (!p->d_mfma || !mfma512_enabled()) return false;
    if (bg.uniform_samples <= 0 || bg.seg) return false;
    if (dtype != DSP_WAVE_F32 && dtype != DSP_WAVE_I16) return false;
    if (dtype == DSP_WAVE_I16 && (bg.uniform_samples & 1)) return false;   // dword range checks: an odd int16 utterance would lose its last sample
    if (delta_n < 0 || delta_n > 2) return false;
    if (bg.uniform_samples * 4 >= ((int64_t)1 << 31) || bg.uniform_frames * 64 * 4 >= ((int64_t)1 << 31) || (int64_t)bg.n_utt * 64 >= ((int64_t)1 << 31)) return false;   // 32-bit descriptor ranges and item counts
    if (bg.total_frames < (int64_t)mfma512_device_cus() * M512_WAVES * 16) return false;
    return true;
}

template <int DTYPE, int NMT, int ND, int WAVES>
static int mfma512_launch_k(const M512Params& P, const void* d_wave, float* d_out, hipStream_t st) {
    constexpr int HS = 10;
    constexpr int WAVE_BYTES = 2 * 8 * 192 * 2 + 24 * 64;
    constexpr int NWB = m512_octet_blocks<NMT>(0) + m512_octet_blocks<NMT>(1);
    constexpr bool DM_LDS = 65536 + NWB * 2048 + 8192 + WAVES * WAVE_BYTES <= 163840;
    constexpr int XTRA = DM_LDS ? 8192 : 4096;
    const size_t lds = 65536 + (size_t)NWB * 2048 + XTRA + (size_t)WAVES * WAVE_BYTES;
    static_assert(65536 + NWB * 2048 + XTRA + WAVES * WAVE_BYTES <= 163840, "LDS budget");
    auto kern = mfcc512m_kernel<HS, DTYPE, NMT, ND, WAVES>;
    static std::atomic<unsigned long long> attr_set{0};   // per instantiation, one bit per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DSP_EHIP;
    if (dev >= 64 || !((attr_set.load() >> dev) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess)
            return 1;   // not served on this device: the caller falls through to the vector-pipe kernels
        if (dev < 64) attr_set.fetch_or(1ull << dev);
    }
    int grid = mfma512_device_cus();
    const int64_t items = (int64_t)P.n_utt * P.splits;
    const int need = (int)((items + WAVES - 1) / WAVES);
    if (grid > need) grid = need;
    M512Params Q = P;
    Q.items_q = (int32_t)(items / ((int64_t)grid * WAVES));
    Q.items_r = (int32_t)(items % ((int64_t)grid * WAVES));
    kern<<<grid, 64 * WAVES, lds, st>>>(Q, d_wave, d_out);
    return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
}

// 0 = launched; 1 = not served (caller falls through); < 0 error
static inline int mfma512_launch(const dsp_plan* p, const void* d_wave, int dtype, const BatchGeom& bg, int delta_n,
                                 float* d_out, int64_t ld_out, hipStream_t st) {
    const Mfma512Plan* mp = static_cast<const Mfma512Plan*>(p->d_mfma);
    M512Params P;
    memset(&P, 0, sizeof(P));
    P.tables = mp->d_tables;
    P.lay = mp->lay;
    P.L = p->L; P.S = p->S; P.C = p->C; P.append_energy = p->append_energy;
    P.preemph = p->preemph;
    P.delta_n = delta_n;
    int den = 0;
    for (int i = 1; i <= delta_n; ++i) den += i * i;
    P.inv_den = den ? (float)(1.0 / (2.0 * den)) : 0.f;
    P.ld_out = ld_out;
    P.n_utt = bg.n_utt;
    P.samples = bg.uniform_samples;
    P.frames = bg.uniform_frames;
    // A/B knobs, read once (function-local statics: initialised thread-safely)
    static const int stg = [] { const char* e = getenv("DSP_M512_STAGGER"); return e ? atoi(e) : 2; }();
    static const int waves = [] { const char* e = getenv("DSP_M512_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();
    static const int forced_splits = [] { const char* e = getenv("DSP_M512_SPLITS"); return e ? atoi(e) : 0; }();
    P.stagger = stg < 0 ? 0 : (stg > 64 ? 64 : stg);
    // fewer utterances than wave slots: every utterance is cut into row ranges (each range recomputes 8 + 4 frames
    // of its neighbours for the delta windows), as many as keep a range at two tiles or more
    {
        const int64_t slots = (int64_t)mfma512_device_cus() * waves;
        int splits = 1;
        while ((int64_t)bg.n_utt * splits < slots && bg.uniform_frames / (splits + 1) >= 32 && splits < 64) ++splits;
        if (forced_splits > 0 && forced_splits <= 64) splits = forced_splits;
        P.splits = splits;
    }
    const int nmt = mp->lay.n_mtiles;
#define M512_LAUNCH(DT_, NMT_, ND_) \
    do { if (waves == 4) return mfma512_launch_k<DT_, NMT_, ND_, 4>(P, d_wave, d_out, st); \
         return mfma512_launch_k<DT_, NMT_, ND_, 8>(P, d_wave, d_out, st); } while (0)
#define M512_LAUNCH_NMT(DT_, ND_) \
    do { if (nmt <= 2) M512_LAUNCH(DT_, 2, ND_); else M512_LAUNCH(DT_, 3, ND_); } while (0)
#define M512_LAUNCH_DT(ND_) \
    do { if (dtype == DSP_WAVE_I16) M512_LAUNCH_NMT(DSP_WAVE_I16, ND_); else M512_LAUNCH_NMT(DSP_WAVE_F32, ND_); } while (0)
    if (delta_n == 0) M512_LAUNCH_DT(0); else if (delta_n == 1) M512_LAUNCH_DT(1); else M512_LAUNCH_DT(2);
    return 1;
}
#endif  // M512_KERNEL_ONLY
