// Specialised fused MFCC kernel for NFFT = 512 (the BASELINE metric configuration and every other
// plan with L <= 512): waveform -> MFCC without touching HBM in between.
//
// Work decomposition (gfx950, wave64):
//   * one wavefront = 8 consecutive frames of one utterance, 8 lanes per frame (lane = 8 f + c);
//   * the 512-point real FFT of a frame is split n = 16 n1 + n2:
//       pass 1  lane c owns columns n2 = 2c, 2c+1: ONE in-register complex FFT32 over n1 of
//               col_a + i col_b, untangled to rows k1 = 0..16 (real-input symmetry), twiddled;
//       exchange through LDS (2 x 1 KB per frame, XOR-swizzled, conflict-free b128 both ways);
//       pass 2  lane c owns rows k1 = c and c + 8: TWO in-register complex FFT16 over n2
//               -> X[k1 + 32 k2]; rows 0 and 16 (both real) ride together as one packed 32-point
//               real FFT in lane 0's first slot, so every lane runs the same two FFT16s;
//   * |X|^2 -> LDS (one 264-float row per frame) -> table-driven sparse mel triangles (lane c owns
//     filters c, c+8, ...) -> log -> DCT*lifter partial sums -> 3-step DPP all-reduce over the
//     frame's 8 lanes -> energy swap -> store.
//   * the 8 frames' samples (7 S + L of them) are read from HBM once, coalesced 16 B per lane,
//     pre-emphasised on the fly and staged in LDS; overlapping frames re-read LDS, not HBM.
// No MFMA: the path has no dense contraction (mel is 96 % sparse, the DCT is 13 x 40).
#pragma once

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "dsp_common.h"
#include "workspace.h"
#include "fft_inreg.h"

#define F512_WAVE_FLOATS 2112  // per-wave LDS: 8 frames x 264 floats (staging / exchange alias it)
#define F512_PS_STRIDE 264     // == 8 (mod 32): the 8 frames' rows start on distinct bank octets
#define F512_MAX_NI 6   // filter iterations of the catch-all instantiation: up to 48 filters
#ifndef F512_WAVES
#define F512_WAVES 8
#endif
#ifndef F512_MIN_WAVES_PER_SIMD
#define F512_MIN_WAVES_PER_SIMD 4
#endif

struct F512Params {
    const float* tables;   // device blob copied to LDS by every workgroup
    int32_t tab_floats;    // multiple of 64 floats (256 B)
    int32_t off_tw1, off_dct, off_melw, off_mels, off_bias, off_coop;
    int32_t melw_row;      // floats per lane row of mel weights (multiple of 4, /4 odd: conflict-free b128)
    int32_t L, S, M, C, append_energy;
    float preemph;
    int32_t span_vec;      // ceil((7 S + L) / 4): 16-byte vectors staged per wave
    int32_t nb4[F512_MAX_NI];  // 16-byte blocks (4 taps) per filter iteration, the longest of its 8 slots
    int64_t groups_per_utt, total_groups;   // uniform batches
    uint32_t t_magic, t_shift;               // x / uniform_frames   = (uint64) x * t_magic >> t_shift   for x < 2^30
    uint32_t g_magic, g_shift;               // x / groups_per_utt, likewise
    int32_t fd_n;          // fused delta (FD instantiations): delta window N (1 or 2), see "Fused delta" below
    float fd_inv_den;      // 1 / (2 sum n^2)
    int32_t flat;          // uniform batches: groups are cut from the FLAT frame sequence (a group may span two utterances)
    int32_t seam_off;      // ... then frames of the second utterance sit this many floats further into the LDS image
    const int32_t* group_off;                // ragged: [B+1] prefix of ceil(T_b / 8)
    const int32_t* group_utt;                // ragged: utterance of every frame group
};

struct Fast512Plan {
    float* d_tables;
    F512Params P;
    int variant;           // which <NROWS, NI> instantiation serves this plan
    int caps;              // ... and which compile-time mel block table (0: counts from P.nb4)
};

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// Sum over the 8 lanes of a frame (lanes 8f .. 8f+7); every lane ends with the total.
__device__ __forceinline__ float frame_allreduce(float v) {
    v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);  // row_half_mirror: lane i <-> 7 - i inside each 8-lane half row
    return v;
}

// Compile-time mel block counts (16-byte blocks of 4 taps per filter iteration) of the two common plans, so
// that the mel loop unrolls with immediate offsets.  A plan whose own counts fit under a table uses it (the
// extra blocks carry zero weights); any other plan runs the CAPS = 0 instantiation with counts from P.nb4.
//   CAPS 1: 40 filters, 16 kHz (the metric configuration; 33..40 filters at 8/16 kHz fit)   CAPS 2: 26 filters (base.py defaults)
__host__ __device__ constexpr int f512_cap(int caps, int i) {
    constexpr int c1[5] = {2, 3, 4, 5, 8}, c2[4] = {3, 4, 6, 12};
    return caps == 1 ? c1[i] : c2[i];
}
__host__ __device__ constexpr int f512_cap_prefix(int caps, int i) {
    int s = 0;
    for (int k = 0; k < i; ++k) s += f512_cap(caps, k);
    return s;
}
template <int I, int N, typename F>
__device__ __forceinline__ void f512_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        f512_static_for<I + 1, N>(f);
    }
}
// Which cepstral coefficient register r of lane c accumulates: r ^ f512_cbase(c).  The map is chosen so that the
// three reduce-scatter steps (partners c^7, c^2, c^1: row_half_mirror and two quad_perms) need no selects:
// cbase(c ^ 7) = cbase(c) ^ 8, cbase(c ^ 2) = cbase(c) ^ 4, cbase(c ^ 1) = cbase(c) ^ 2.
__host__ __device__ constexpr int f512_cbase(int c) { return ((c & 1) ? 2 : 0) ^ ((c & 2) ? 4 : 0) ^ ((c & 4) ? 14 : 0); }
typedef float f512_f2u __attribute__((ext_vector_type(2), aligned(4)));

#define F512_FENCE() asm volatile("" ::: "memory")

// Diagnostic build (-DF512_STAMPS, see tools/kbench.py): every wave accumulates the shader-clock span of
// each phase of its groups in registers and adds them to a per-wave slot at exit; phases are pinned with
// empty volatile asms so the compiler cannot move arithmetic across a stamp.  Never in the product library.
#ifdef F512_STAMPS
#define F512_NSTAMP 16
__device__ unsigned int f512_stamp_sum[F512_NSTAMP * 8192];   // [wave slot][phase]
__device__ __forceinline__ unsigned int f512_clock() {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return (unsigned int)t;
}
template <typename T, int N>
__device__ __forceinline__ void f512_pin(T (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if constexpr (sizeof(T) == 8) asm volatile("" : "+v"(a[i].x), "+v"(a[i].y));
        else asm volatile("" : "+v"(a[i]));
    }
}
#ifdef F512_STAMP_ROUNDS   // per-ROUND timing instead of per phase: slot r = lifetime of the wave's r-th group, slot 4 + r = count
#define F512_STAMP(i)                                                         \
    do {                                                                      \
        if ((i) == 0) stamp_prev_ = f512_clock();                             \
        if ((i) == 10) {                                                      \
            stamp_acc_[r < 4 ? r : 3] += f512_clock() - stamp_prev_;          \
            stamp_acc_[4 + (r < 4 ? r : 3)] += 1;                             \
        }                                                                     \
    } while (0)
#else
#define F512_STAMP(i)                                     \
    do {                                                  \
        const unsigned int now_ = f512_clock();           \
        stamp_acc_[i] += now_ - stamp_prev_;              \
        stamp_prev_ = f512_clock();                       \
    } while (0)
#endif
#define F512_PIN(a) f512_pin(a)
#else
#define F512_STAMP(i) do {} while (0)
#define F512_PIN(a) do {} while (0)
#endif

typedef float f512_v2 __attribute__((ext_vector_type(2)));
// Pass-1 reads are plain ds_read_b64 (2 LDS cycles per wave, 64-bank addressing) issued from inline
// asm: hipcc would pair neighbouring b64 reads into ds_read2_b64, which costs 8 cycles and uses
// 32-bank addressing -- with frames 160 floats apart that is a 2-way conflict on top.
__device__ __forceinline__ uint32_t f512_lds_addr(const void* p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));
}

// Rows ROW0 .. ROW0+7 of the frame's column pair and of the window: 16 plain ds_read_b64 and their
// wait in ONE asm statement, so the outputs are really valid where the compiler believes they are.
// (With the wait in a separate statement hipcc may spill or move a destination register before
// its data has arrived -- seen once register pressure forced spills in the 32-row instantiation.)
template <int ROW0>
__device__ __forceinline__ void f512_load_rows8(uint32_t fr, uint32_t wn, f512_v2 (&xv)[8], f512_v2 (&wv)[8]) {
    asm volatile(
        "ds_read_b64 %0, %16 offset:%18\n\tds_read_b64 %8, %17 offset:%18\n\t"
        "ds_read_b64 %1, %16 offset:%19\n\tds_read_b64 %9, %17 offset:%19\n\t"
        "ds_read_b64 %2, %16 offset:%20\n\tds_read_b64 %10, %17 offset:%20\n\t"
        "ds_read_b64 %3, %16 offset:%21\n\tds_read_b64 %11, %17 offset:%21\n\t"
        "ds_read_b64 %4, %16 offset:%22\n\tds_read_b64 %12, %17 offset:%22\n\t"
        "ds_read_b64 %5, %16 offset:%23\n\tds_read_b64 %13, %17 offset:%23\n\t"
        "ds_read_b64 %6, %16 offset:%24\n\tds_read_b64 %14, %17 offset:%24\n\t"
        "ds_read_b64 %7, %16 offset:%25\n\tds_read_b64 %15, %17 offset:%25\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3]), "=&v"(xv[4]), "=&v"(xv[5]), "=&v"(xv[6]), "=&v"(xv[7]),
          "=&v"(wv[0]), "=&v"(wv[1]), "=&v"(wv[2]), "=&v"(wv[3]), "=&v"(wv[4]), "=&v"(wv[5]), "=&v"(wv[6]), "=&v"(wv[7])
        : "v"(fr), "v"(wn), "n"(64 * (ROW0 + 0)), "n"(64 * (ROW0 + 1)), "n"(64 * (ROW0 + 2)), "n"(64 * (ROW0 + 3)),
          "n"(64 * (ROW0 + 4)), "n"(64 * (ROW0 + 5)), "n"(64 * (ROW0 + 6)), "n"(64 * (ROW0 + 7))
        : "memory");
}
template <int ROW0>
__device__ __forceinline__ void f512_load_rows1(uint32_t fr, uint32_t wn, f512_v2& xv, f512_v2& wv) {
    asm volatile("ds_read_b64 %0, %2 offset:%4\n\tds_read_b64 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(xv), "=&v"(wv) : "v"(fr), "v"(wn), "n"(64 * ROW0) : "memory");
}
// z[ROW0 ..] = x * w for rows ROW0 .. NROWS-1, eight at a time
template <int NROWS, int ROW0>
__device__ __forceinline__ void f512_window_rows(uint32_t fr, uint32_t wn, cpx (&z)[32]) {
    if constexpr (ROW0 + 8 <= NROWS) {
        f512_v2 xv[8], wv[8];
        f512_load_rows8<ROW0>(fr, wn, xv, wv);
#pragma unroll
        for (int r = 0; r < 8; ++r) z[ROW0 + r] = {xv[r].x * wv[r].x, xv[r].y * wv[r].y};
        f512_window_rows<NROWS, ROW0 + 8>(fr, wn, z);
    } else if constexpr (ROW0 < NROWS) {
        f512_v2 xv, wv;
        f512_load_rows1<ROW0>(fr, wn, xv, wv);
        z[ROW0] = {xv.x * wv.x, xv.y * wv.y};
        f512_window_rows<NROWS, ROW0 + 1>(fr, wn, z);
    }
}

// One staged vector = 4 consecutive samples as they sit in HBM (16 B of fp32 or 8 B of int16).
template <int DTYPE> struct F512Raw { float4 v; };
template <> struct F512Raw<DSP_WAVE_I16> { short4 v; };

template <int DTYPE>
__device__ __forceinline__ F512Raw<DTYPE> f512_load_raw(const void* __restrict__ wave, int64_t idx) {
    // idx: element index (multiple of 4) in the concatenated waveform buffer
    F512Raw<DTYPE> r;
    if constexpr (DTYPE == DSP_WAVE_I16) r.v = *reinterpret_cast<const short4*>(reinterpret_cast<const int16_t*>(wave) + idx);
    else r.v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(wave) + idx);
    return r;
}
// same, from any element address (4-byte / 2-byte aligned): interior groups of ragged batches
typedef float f512_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef short f512_s4u __attribute__((ext_vector_type(4), aligned(2)));
template <int DTYPE>
__device__ __forceinline__ F512Raw<DTYPE> f512_load_raw_unaligned(const void* __restrict__ wave, int64_t idx) {
    F512Raw<DTYPE> r;
    if constexpr (DTYPE == DSP_WAVE_I16) {
        const f512_s4u t = *reinterpret_cast<const f512_s4u*>(reinterpret_cast<const int16_t*>(wave) + idx);
        r.v = make_short4(t.x, t.y, t.z, t.w);
    } else {
        const f512_f4u t = *reinterpret_cast<const f512_f4u*>(reinterpret_cast<const float*>(wave) + idx);
        r.v = make_float4(t.x, t.y, t.z, t.w);
    }
    return r;
}
template <int DTYPE>
__device__ __forceinline__ void f512_unpack(const F512Raw<DTYPE>& r, float (&x)[4]) {
    x[0] = (float)r.v.x; x[1] = (float)r.v.y; x[2] = (float)r.v.z; x[3] = (float)r.v.w;
}
// lane l receives lane l-1's value; lane 0 receives `left`.  DPP wave_shr:1 -- one VALU op, no LDS
// (verified on gfx950 hardware with tools/probes/dpp_probe.hip).
__device__ __forceinline__ float f512_shift_in(float v, float left) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(left), __float_as_int(v), 0x138, 0xF, 0xF, false));
}

// floor(x / d) for 0 <= x < 2^30 as a multiply and a shift: magic = ceil(2^(30 + l) / d), l = ceil(log2 d).
static inline void f512_magic(uint32_t d, uint32_t& magic, uint32_t& shift) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    shift = 30 + l;
    magic = (uint32_t)(((1ull << shift) + d - 1) / d);
}
__device__ __forceinline__ int f512_div(int x, uint32_t magic, uint32_t shift) {
    return (int)(((uint64_t)(uint32_t)x * magic) >> shift);
}

// Where a frame group lives (all wave-uniform).
struct F512Group {
    int utt, t0, T, nsamp;
    int nf1;               // frames of the group that belong to `utt` (8 unless the group spans a seam)
    int64_t s0, row0;
};

template <bool RAGGED>
__device__ __forceinline__ F512Group f512_locate(const F512Params& P, const BatchGeom& bg, int G) {
    F512Group g;
    if constexpr (RAGGED) {
        g.utt = P.group_utt[G];
        g.t0 = (G - P.group_off[g.utt]) * 8;
        g.s0 = bg.sample_off[g.utt];
        g.nsamp = (int)(bg.sample_off[g.utt + 1] - g.s0);
        if (bg.seg != nullptr) {   // the utterance is a segment of that range, read in place
            const int64_t lo = bg.seg[2 * g.utt];
            g.nsamp = (int)(bg.seg[2 * g.utt + 1] - lo);
            g.s0 += lo;
        }
        g.row0 = bg.frame_off[g.utt];
        g.T = (int)(bg.frame_off[g.utt + 1] - g.row0);
        g.nf1 = 8;
    } else {
        g.nsamp = (int)bg.uniform_samples;
        g.T = (int)bg.uniform_frames;
        if (P.flat) {
            // groups of 8 cut from the flat frame sequence: no frame slot is wasted at the end of an utterance
            const int F0 = 8 * G;
            g.utt = f512_div(F0, P.t_magic, P.t_shift);
            g.t0 = F0 - g.utt * g.T;
            g.nf1 = g.T - g.t0 < 8 ? g.T - g.t0 : 8;
        } else {
            const int gpu = (int)P.groups_per_utt;
            g.utt = f512_div(G, P.g_magic, P.g_shift);
            g.t0 = (G - g.utt * gpu) * 8;
            g.nf1 = 8;
        }
        g.s0 = (int64_t)g.utt * bg.uniform_samples;
        g.row0 = (int64_t)g.utt * bg.uniform_frames;
    }
    return g;
}

// Fused delta: the frames a workgroup owns and the run of groups wave `w` takes from them (wave-uniform).
struct F512Run {
    int wg_first, wg_end;   // flat frames [wg_first, wg_end) of the workgroup: whole utterances
    int n_groups;           // groups of this wave (>= 1: the host only launches FD grids whose workgroups hold >= WAVES groups)
    int first_frame;        // first flat frame of the wave's first group
};
__device__ __forceinline__ F512Run f512_run(const BatchGeom& bg, int b, int nb, int w, int waves) {
    F512Run r;
    const int T = (int)bg.uniform_frames;
    const int u0 = (int)((int64_t)b * bg.n_utt / nb), u1 = (int)((int64_t)(b + 1) * bg.n_utt / nb);
    r.wg_first = u0 * T;
    r.wg_end = u1 * T;
    const int ng = (r.wg_end - r.wg_first + 7) >> 3;
    const int q = ng / waves, rem = ng - q * waves;
    r.n_groups = q + (w < rem ? 1 : 0);
    r.first_frame = r.wg_first + 8 * (w * q + (w < rem ? w : rem));
    return r;
}

// Fused-delta instantiations cut their groups from a workgroup's own frame range: a group is named by its first
// (flat) frame, which need not be a multiple of 8.
__device__ __forceinline__ F512Group f512_locate_frame(const F512Params& P, const BatchGeom& bg, int F0) {
    F512Group g;
    g.nsamp = (int)bg.uniform_samples;
    g.T = (int)bg.uniform_frames;
    g.utt = f512_div(F0, P.t_magic, P.t_shift);
    g.t0 = F0 - g.utt * g.T;
    g.nf1 = g.T - g.t0 < 8 ? g.T - g.t0 : 8;
    g.s0 = (int64_t)g.utt * bg.uniform_samples;
    g.row0 = (int64_t)g.utt * bg.uniform_frames;
    return g;
}

// RAGGED = false: dense [B, N] batch, N % 4 == 0 (every 16-byte vector is all-valid or all-padding).
// RAGGED = true : concatenated utterances of any length at any offset; loads stay 16-byte aligned
//                 (the LDS image starts at the aligned sample below the group's first one).
//
// Fused delta (FD = delta window N = 1 or 2, dense batches only; base.py:70-79 applied twice): the kernel writes whole [x | d | dd] rows.
//   * Workgroup b owns whole utterances [b U / nb, (b + 1) U / nb), so nothing is needed from another workgroup
//     (the delta window is edge-replicated at utterance ends); its frames are cut into groups of 8 and wave w takes
//     a CONTIGUOUS run of them.
//   * A wave keeps the cepstra of its previous group in two registers.  After group j it writes previous + current
//     group (16 frames) into its own LDS buffer -- dead at that point -- and emits the rows of the 8 frames in the
//     middle of that window (delta of the edge-replicated x, then delta of the edge-replicated delta, exactly as the
//     reference composes them), i.e. rows trail the computation by 2 N <= 4 frames.
//   * The last 4 rows of a wave's run need the first group of the NEXT wave: every wave publishes its first group's
//     cepstra in a small LDS area, ONE barrier follows the first iteration, and a final iteration without
//     computation emits the tail.  The first wave of the workgroup also emits its own first rows.
template <int NROWS, int NI, int CAPS, int NSTAGE, int DTYPE, int WAVES, bool RAGGED, int FD = 0>
__global__ __launch_bounds__(64 * WAVES, F512_MIN_WAVES_PER_SIMD) void mfcc512_kernel(F512Params P, BatchGeom bg,
                                                             const void* __restrict__ wave,
                                                             float* __restrict__ out, int64_t ld_out) {
    extern __shared__ __attribute__((aligned(256))) float smem_f[];
    float* const smem = smem_f;
    const int tid = threadIdx.x;
#ifdef F512_STAMPS
    const unsigned int stamp_entry_ = f512_clock();
    const unsigned int stamp_rt0_ = (unsigned int)__builtin_amdgcn_s_memrealtime();
#endif
    // Prologue.  The table loads are issued FIRST, then (dense batches) one dword per 16-byte vector of the wave's first
    // group: the copy to LDS and the barrier wait for the tables only (vmcnt leaves the younger touches in flight),
    // so the workgroup's eight waves are not coupled to the slowest of their first HBM fetches -- at the start of a
    // launch every wave of the chip asks for its first 6 KB at once, a 25 MB burst.
    // (Inline asm for the table loads: the compiler would otherwise sink them below the touches and wait for everything.
    // Older than every load the compiler issues afterwards, they never make its own vmcnt waits too short.)
    constexpr int TABV = 3;
    typedef float f512_v4 __attribute__((ext_vector_type(4)));
    f512_v4 tabv_[TABV];
#pragma unroll
    for (int k = 0; k < TABV; ++k) {
        const int i = tid * 4 + 64 * WAVES * 4 * k;
        const float* src = P.tables + (i < P.tab_floats ? i : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(tabv_[k]) : "v"(src) : "memory");
    }
    if constexpr (!RAGGED) {
        float warm_[NSTAGE];
        int64_t e;
        if constexpr (FD) {
            const F512Run run = f512_run(bg, (int)blockIdx.x, (int)gridDim.x, tid >> 6, WAVES);
            const int u0 = f512_div(run.first_frame, P.t_magic, P.t_shift);
            e = (int64_t)u0 * bg.uniform_samples + (int64_t)(run.first_frame - u0 * (int)bg.uniform_frames) * P.S;
        } else {
            int G0 = (int)blockIdx.x * WAVES + (tid >> 6);
            G0 = G0 < (int)P.total_groups ? G0 : (int)P.total_groups - 1;
            const int u0 = P.flat ? f512_div(8 * G0, P.t_magic, P.t_shift) : f512_div(G0, P.g_magic, P.g_shift);
            const int t00 = P.flat ? 8 * G0 - u0 * (int)bg.uniform_frames : (G0 - u0 * (int)P.groups_per_utt) * 8;
            e = (int64_t)u0 * bg.uniform_samples + (int64_t)t00 * P.S;
        }
        const int64_t lim = (int64_t)bg.n_utt * bg.uniform_samples - 4;
#pragma unroll
        for (int q = 0; q < NSTAGE; ++q) {
            int64_t idx = e + 4 * (tid & 63) + 256 * q;
            idx = idx < lim ? idx : lim;
            warm_[q] = dsp_load_sample<DTYPE>(wave, idx);
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTAGE) : "memory");   // the tables; the NSTAGE touches stay in flight
#pragma unroll
        for (int k = 0; k < TABV; ++k) {
            const int i = tid * 4 + 64 * WAVES * 4 * k;
            if (i < P.tab_floats) *reinterpret_cast<f512_v4*>(smem + i) = tabv_[k];
        }
        for (int i = tid * 4 + 64 * WAVES * 4 * TABV; i < P.tab_floats; i += 64 * WAVES * 4)
            *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(P.tables + i);
        __syncthreads();
        // not used: this only keeps the touches alive; every wave waits for its OWN first fetch here
#pragma unroll
        for (int q = 0; q < NSTAGE; ++q) asm volatile("" :: "v"(warm_[q]));
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < TABV; ++k) {
            const int i = tid * 4 + 64 * WAVES * 4 * k;
            if (i < P.tab_floats) *reinterpret_cast<f512_v4*>(smem + i) = tabv_[k];
        }
        for (int i = tid * 4 + 64 * WAVES * 4 * TABV; i < P.tab_floats; i += 64 * WAVES * 4)
            *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(P.tables + i);
        __syncthreads();
    }
    const float* s_win = smem;
    const float4* s_tw1 = reinterpret_cast<const float4*>(smem + P.off_tw1);
    const float* s_dct = smem + P.off_dct;
    const float* s_melw = smem + P.off_melw;

    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* wbuf = smem + P.tab_floats + wid * F512_WAVE_FLOATS;
    const int total_groups = RAGGED ? P.group_off[bg.n_utt] : (int)P.total_groups;
    const int gstride = (int)gridDim.x * WAVES;
    // Full rounds: wave w of block b takes group r * gstride + 8 b + w (the eight waves of a workgroup
    // hold eight consecutive groups, whose overlapping samples are re-read through one CU's L1 / L2).
    // The last, partial round is dealt wave-major instead (group base + b + gridDim * w): its groups
    // go to wave 0 of every workgroup first, so no SIMD carries more than one group above the average.
    const int nfull = total_groups / gstride;
#ifdef F512_STAMPS
    unsigned int stamp_acc_[F512_NSTAMP] = {};
    unsigned int stamp_prev_ = f512_clock();
    const unsigned int stamp_loop0_ = stamp_prev_;
#endif

    // Fused delta: this wave's run of groups, its previous group's cepstra (two per lane) and the halo area behind
    // the wave buffers where every wave leaves its first group (8 frames x 14 coefficients) for its predecessor.
    F512Run run = {0, 0, 0, 0};
    float fd_p0 = 0.f, fd_p1 = 0.f;
    float* const fd_halo = smem + P.tab_floats + WAVES * F512_WAVE_FLOATS;
    if constexpr (FD) run = f512_run(bg, (int)blockIdx.x, (int)gridDim.x, wid, WAVES);
    const int n_iter = FD ? run.n_groups + 1 : nfull + 1;   // FD: one more, computation-free iteration emits the tail rows

    for (int r = 0; r < n_iter; ++r) {
        int G = 0;
        if constexpr (!FD) {
            if (r < nfull) {
                G = __builtin_amdgcn_readfirstlane(r * gstride + (int)blockIdx.x * WAVES + wid);
            } else {
                G = __builtin_amdgcn_readfirstlane(nfull * gstride + (int)blockIdx.x + (int)gridDim.x * wid);
                if (G >= total_groups) break;
            }
        }
        // Lane-derived addresses are recomputed every iteration on purpose: hoisted out of the loop
        // they would pin ~30 VGPRs for the whole kernel (the opaque asm stops the hoisting).
        F512_STAMP(0);
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int f = lane >> 3, c = lane & 7;
        // (F512_COOP, an experiment that is NOT compiled into the product library: rows 0 and 16 by two FFT8 across the
        //  frame's lanes instead of lane 0's packed FFT16.  Measured on the MI355X: 6 % fewer vector instructions per
        //  group (1330 instead of 1420), parity green -- and no gain in time at any batch size (39.4 vs 40.2 us at
        //  configs[1], 414 vs 418 us at 12 500 utterances): the 21 extra DPP operands with their wait states and the
        //  serial butterfly chain cost what the idle lanes did.  Kept as a build flag for the record.)
#ifdef F512_COOP
        // Column pair of this lane in pass 1: 0, 1, 2, 3, 7, 6, 5, 4 -- chosen so that the three stages of a
        // decimation-in-frequency FFT8 ACROSS the frame's lanes (rows 0 and 16, below) pair lane c with c ^ 7, c ^ 2
        // and c ^ 1: row_half_mirror and two quad_perms, one DPP operand each.
        const int np = c < 4 ? c : 11 - c;
#else
        const int np = c;
#endif
        float v0 = 0.f, v1 = 0.f;   // the frame's two cepstral coefficients this lane ends up with
        do {
        if constexpr (FD) { if (r >= run.n_groups) break; }
        const int sigma_hi = ((f >> 1) & 1) << 2;  // exchange swizzle: slot ^= (u >> 1) ^ sigma_hi
        const F512Group grp = FD ? f512_locate_frame(P, bg, run.first_frame + 8 * r) : f512_locate<RAGGED>(P, bg, G);
        const int utt = grp.utt, t0 = grp.t0, T = grp.T, nsamp = grp.nsamp;
        (void)utt;
        const int base = t0 * P.S;                            // first sample of the group, utterance relative
        const int64_t g0 = grp.s0 + base;                     // ... in the concatenated buffer
        // A group that lies entirely inside its utterance (12 of the 13 groups of a 1 s clip) is staged with no
        // per-vector bookkeeping at all: one base pointer, immediate offsets, no masks.  Ragged batches load
        // such a group from its first sample as it stands (4-byte aligned vector loads, nothing outside the
        // utterance is touched), so its LDS image needs no alignment shift either.
        const bool fast_stage = (NSTAGE * 256 <= F512_WAVE_FLOATS) && grp.nf1 == 8 && base + NSTAGE * 256 <= nsamp;
        const int d = (RAGGED && !fast_stage) ? (int)(g0 & 3) : 0;   // LDS image starts d samples earlier (aligned)

        // Unit-variance statistics of segments read in place (bg.stats): every group adds the sums of its own share
        // [base, base + 8 S) of the utterance -- the last group takes everything up to the end -- of (x - x0) and
        // (x - x0)^2, x0 = the utterance's first sample (shift-invariant, and no cancellation under a DC offset).
        float st_s = 0.f, st_q = 0.f, st_ref = 0.f;
        int st_lim = 0;
        bool do_stats = false;
        if constexpr (RAGGED) {
            do_stats = bg.stats != nullptr && nsamp > 0;
            if (do_stats) {
                st_ref = dsp_load_sample<DTYPE>(wave, grp.s0);
                st_lim = t0 + 8 >= T ? nsamp : base + 8 * P.S;
            }
        }
        // ---- stage 7 S + L (+ d) samples: coalesced aligned 16 B loads, all issued before first use;
        //      pre-emphasis, zero fill outside the utterance. ----
        if (fast_stage) {
            {
                asm volatile("; F512_FAST_STAGE (tools/asm_count.py counts from here to the loop's back edge)");
                F512Raw<DTYPE> raw[NSTAGE];
                float prev[NSTAGE];
                const int64_t e0 = g0 + 4 * lane;
#pragma unroll
                for (int r = 0; r < NSTAGE; ++r)
                    raw[r] = RAGGED ? f512_load_raw_unaligned<DTYPE>(wave, e0 + 256 * r) : f512_load_raw<DTYPE>(wave, e0 + 256 * r);
                // the sample before each vector: one more (cache-resident) dword per lane instead of a DPP shift,
                // a v_readlane and a move per vector
#ifndef F512_STAGE_DPP
#pragma unroll
                for (int r = 1; r < NSTAGE; ++r) prev[r] = dsp_load_sample<DTYPE>(wave, e0 + 256 * r - 1);
#endif
                // an utterance's first sample is not filtered (sigproc.py:185): y[0] = x[0] - c * 0
                if (base > 0) prev[0] = dsp_load_sample<DTYPE>(wave, e0 - 1);
                else prev[0] = lane > 0 ? dsp_load_sample<DTYPE>(wave, e0 - 1) : 0.f;
#ifdef F512_STAGE_DPP   // A/B: the round-2 form (previous sample through DPP wave_shr + v_readlane)
                float left = prev[0];
#pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    float x[4];
                    f512_unpack<DTYPE>(raw[r], x);
                    prev[r] = f512_shift_in(x[3], left);
                    left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[3]), 63));
                }
#endif
#pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    float x[4];
                    f512_unpack<DTYPE>(raw[r], x);
                    float4 y;
                    y.x = fmaf(-P.preemph, prev[r], x[0]);
                    y.y = fmaf(-P.preemph, x[0], x[1]);
                    y.z = fmaf(-P.preemph, x[1], x[2]);
                    y.w = fmaf(-P.preemph, x[2], x[3]);
                    *reinterpret_cast<float4*>(wbuf + 4 * lane + 256 * r) = y;
                    if constexpr (RAGGED) {
                        if (do_stats) {
                            const int round_in = st_lim - (base + 256 * r);    // samples of this round inside the share (wave-uniform)
                            if (round_in >= 256) {                              // all of it: the common case, no masks
                                const float d0 = x[0] - st_ref, d1 = x[1] - st_ref, d2 = x[2] - st_ref, d3 = x[3] - st_ref;
                                st_s += (d0 + d1) + (d2 + d3);
                                st_q = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, st_q))));
                            } else if (round_in > 0) {
                                const int left_in_share = round_in - 4 * lane;  // samples of this vector inside
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
            }
        } else if (!RAGGED) {
            // Dense batch, end of an utterance: zero padding past its last sample and, for flat grouping, the
            // first frames of the NEXT utterance as a second segment `seam_off` floats behind where they would
            // sit otherwise (frame slot f >= nf1 reads from f S + seam_off), so the two utterances' samples
            // never overlap in LDS.  Everything is a multiple of 4 here: vectors are all-valid or all-padding.
            const int nf1 = grp.nf1;
            const int X = nf1 * P.S + P.seam_off;               // LDS float offset of segment 2
            const bool seg2 = nf1 < 8 && utt + 1 < bg.n_utt;
            const int span_vec = P.span_vec + (P.flat ? (P.seam_off >> 2) : 0);
            constexpr int NR = NSTAGE + 1;
            F512Raw<DTYPE> raw[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int v = lane + 64 * r, p = 4 * v;
                const bool in2 = nf1 < 8 && p >= X;
                const int rel = in2 ? p - X : base + p;
                const bool ok = v < span_vec && rel < nsamp && (!in2 || seg2);
                raw[r] = f512_load_raw<DTYPE>(wave, ok ? grp.s0 + (in2 ? (int64_t)nsamp : 0) + rel : 0);
            }
            float left = base > 0 ? dsp_load_sample<DTYPE>(wave, g0 - 1) : 0.f;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int v = lane + 64 * r, p = 4 * v;
                const bool in2 = nf1 < 8 && p >= X;
                const int rel = in2 ? p - X : base + p;
                const bool ok = v < span_vec && rel < nsamp && (!in2 || seg2);
                float x[4];
                f512_unpack<DTYPE>(raw[r], x);
                const float prev = f512_shift_in(x[3], left);
                left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[3]), 63));
                float4 y;
                y.x = (in2 && rel == 0) ? x[0] : fmaf(-P.preemph, prev, x[0]);   // an utterance's first sample is not filtered
                y.y = fmaf(-P.preemph, x[0], x[1]);
                y.z = fmaf(-P.preemph, x[1], x[2]);
                y.w = fmaf(-P.preemph, x[2], x[3]);
                const uint32_t m = ok ? 0xffffffffu : 0u;
                y.x = __uint_as_float(__float_as_uint(y.x) & m);
                y.y = __uint_as_float(__float_as_uint(y.y) & m);
                y.z = __uint_as_float(__float_as_uint(y.z) & m);
                y.w = __uint_as_float(__float_as_uint(y.w) & m);
                if (v < span_vec) *reinterpret_cast<float4*>(wbuf + p) = y;
            }
        } else {
            {
                const int64_t a0 = g0 - d;                         // aligned element index of LDS slot 0
                const int span_vec = RAGGED ? P.span_vec + 1 : P.span_vec;
                F512Raw<DTYPE> raw[NSTAGE];
    #pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    const int v = lane + 64 * r;
                    const int rel = base - d + 4 * v;             // utterance-relative position of element 0
                    const bool touch = v < span_vec && rel + 3 >= 0 && rel < nsamp;  // >= 1 valid element
                    // an aligned vector holding >= 1 valid sample never leaves a mapped page; idle lanes
                    // re-read the start of the buffer
                    raw[r] = f512_load_raw<DTYPE>(wave, touch ? a0 + 4 * v : 0);
                }
                float left = (base - d > 0) ? dsp_load_sample<DTYPE>(wave, a0 - 1) : 0.f;
    #pragma unroll
                for (int r = 0; r < NSTAGE; ++r) {
                    const int v = lane + 64 * r;
                    const int rel = base - d + 4 * v;
                    float x[4];
                    f512_unpack<DTYPE>(raw[r], x);
                    const float prev = f512_shift_in(x[3], left);   // sample rel - 1
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
                        // the utterance may start / end inside this vector: first sample is not filtered,
                        // everything outside [0, nsamp) is zero
                        if (rel + 0 == 0) y.x = x[0];
                        if (rel + 1 == 0) y.y = x[1];
                        if (rel + 2 == 0) y.z = x[2];
                        if (rel + 3 == 0) y.w = x[3];
                        if (rel + 0 < 0 || rel + 0 >= nsamp) y.x = 0.f;
                        if (rel + 1 < 0 || rel + 1 >= nsamp) y.y = 0.f;
                        if (rel + 2 < 0 || rel + 2 >= nsamp) y.z = 0.f;
                        if (rel + 3 < 0 || rel + 3 >= nsamp) y.w = 0.f;
                    } else {
                        const uint32_t m = (v < span_vec && rel < nsamp) ? 0xffffffffu : 0u;  // one select + 4 ANDs
                        y.x = __uint_as_float(__float_as_uint(y.x) & m);
                        y.y = __uint_as_float(__float_as_uint(y.y) & m);
                        y.z = __uint_as_float(__float_as_uint(y.z) & m);
                        y.w = __uint_as_float(__float_as_uint(y.w) & m);
                    }
                    // rounds past the span write zeros inside this wave's own buffer when it is large enough
                    if (NSTAGE * 256 <= F512_WAVE_FLOATS || v < span_vec) *reinterpret_cast<float4*>(wbuf + 4 * v) = y;
                }
            }
        }
        if constexpr (RAGGED) {
            if (do_stats) {
                // sums over each row of 16 lanes with DPP (vector pipe, no LDS traffic)
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
                    unsafeAtomicAdd(bg.stats + 2 * utt, ds);
                    unsafeAtomicAdd(bg.stats + 2 * utt + 1, dq);
                }
            }
        }
        F512_FENCE();
        F512_STAMP(1);

        // ---- pass 1: window, complex FFT32 over n1 of (column 2c) + i (column 2c+1) ----
        cpx z[32];
        {
            const float* frp = wbuf + d + f * P.S + 2 * np + ((!RAGGED && f >= grp.nf1) ? P.seam_off : 0);
            const uint32_t fr = f512_lds_addr(frp);
            const uint32_t wn = f512_lds_addr(s_win + 2 * np);
            if (RAGGED && (d & 1)) {
                // odd offset inside the aligned LDS image: the column pair is not 8-byte aligned
#pragma unroll
                for (int n1 = 0; n1 < NROWS; ++n1) {
                    const float2 wv = *reinterpret_cast<const float2*>(s_win + 2 * np + 16 * n1);
                    z[n1] = {frp[16 * n1] * wv.x, frp[16 * n1 + 1] * wv.y};
                }
            } else {
                f512_window_rows<NROWS, 0>(fr, wn, z);
            }
#pragma unroll
            for (int n1 = NROWS; n1 < 32; ++n1) z[n1] = {0.f, 0.f};
        }
        FFTReg<32>::template run<NROWS>(z);
        F512_FENCE();
        F512_PIN(z);
        F512_STAMP(2);
        F512_PIN(z);

#ifdef F512_COOP
        // ---- rows 0 and 16 of the column transforms are real (z[0] = plain, z[16] = alternating column sums): their
        //      transforms over n2 are the 17 bins X[16 j].  Instead of packing them into one lane's FFT16 and finishing
        //      in that lane alone (1 lane in 8 busy for ~170 instructions), the frame's 8 lanes run two complex FFT8
        //      ACROSS lanes (DPP butterflies) and each lane ends with one even and one odd bin:
        //        X[32 k]      = F[k],  F = real FFT16 of the row-0 values   (y_n = a[2n] + i a[2n+1], FFT8, untangle with k <-> 8 - k)
        //        X[16 + 32 k] = O[k],  O[k] = sum_m b[m] W32^(m (2k+1))     (w_n = (b[2n] + i b[2n+1]) W16^n, FFT8, untangle with k <-> 7 - k)
        //      Both untangles have the form |s + t|^2 with s = A_k + conj(A_partner), t = W' (A_k - conj(A_partner)); the
        //      values carry the same factor 2 as the other rows. ----
        float coop_e, coop_o, coop_e8;
        int coop_k;
        {
            const float4* ct = reinterpret_cast<const float4*>(smem + P.off_coop + 16 * c);
            const float4 c0 = ct[0], c1 = ct[1], c2 = ct[2], c3 = ct[3];
            // c0 = (sign1, sign2, sign3, -), c1 = (tw1, tw2), c2 = (W16^n, -i W16^k), c3 = (-i W32^(2k+1), k as int, -)
            cpx y = z[0];
            cpx w = cmulc(z[16], c2.x, c2.y);
            // stage 1: partner c ^ 7
            {
                const cpx ty = {dpp_f32<0x141>(y.x), dpp_f32<0x141>(y.y)}, tw = {dpp_f32<0x141>(w.x), dpp_f32<0x141>(w.y)};
                y = cmulc({fmaf(y.x, c0.x, ty.x), fmaf(y.y, c0.x, ty.y)}, c1.x, c1.y);
                w = cmulc({fmaf(w.x, c0.x, tw.x), fmaf(w.y, c0.x, tw.y)}, c1.x, c1.y);
            }
            // stage 2: partner c ^ 2
            {
                const cpx ty = {dpp_f32<0x4E>(y.x), dpp_f32<0x4E>(y.y)}, tw = {dpp_f32<0x4E>(w.x), dpp_f32<0x4E>(w.y)};
                y = cmulc({fmaf(y.x, c0.y, ty.x), fmaf(y.y, c0.y, ty.y)}, c1.z, c1.w);
                w = cmulc({fmaf(w.x, c0.y, tw.x), fmaf(w.y, c0.y, tw.y)}, c1.z, c1.w);
            }
            // stage 3: partner c ^ 1
            {
                const cpx ty = {dpp_f32<0xB1>(y.x), dpp_f32<0xB1>(y.y)}, tw = {dpp_f32<0xB1>(w.x), dpp_f32<0xB1>(w.y)};
                y = {fmaf(y.x, c0.z, ty.x), fmaf(y.y, c0.z, ty.y)};
                w = {fmaf(w.x, c0.z, tw.x), fmaf(w.y, c0.z, tw.y)};
            }
            // lane c now holds output k(c) of both transforms: k = 0, 4, 2, 6, 7, 3, 5, 1.
            // even bins: partner output (8 - k) mod 8 sits in lane {0, 1, 3, 2 | 7, 6, 5, 4}[c]
            cpx py, pw;
            {
                int r0 = __builtin_amdgcn_update_dpp(0, __float_as_int(y.x), 0xB4, 0xF, 0x5, false);      // quad_perm [0,1,3,2], quads 0 and 2
                r0 = __builtin_amdgcn_update_dpp(r0, __float_as_int(y.x), 0x1B, 0xF, 0xA, false);         // quad_perm [3,2,1,0], quads 1 and 3
                int r1 = __builtin_amdgcn_update_dpp(0, __float_as_int(y.y), 0xB4, 0xF, 0x5, false);
                r1 = __builtin_amdgcn_update_dpp(r1, __float_as_int(y.y), 0x1B, 0xF, 0xA, false);
                py = {__int_as_float(r0), __int_as_float(r1)};
                // odd bins: partner output 7 - k sits in lane c ^ 4 = (c ^ 7) ^ 3
                pw = {dpp_f32<0x1B>(dpp_f32<0x141>(w.x)), dpp_f32<0x1B>(dpp_f32<0x141>(w.y))};
            }
            {
                const cpx sE = {y.x + py.x, y.y - py.y}, dE = {y.x - py.x, y.y + py.y};
                const cpx tE = cmulc(dE, c2.z, c2.w);
                const cpx ep = {sE.x + tE.x, sE.y + tE.y}, em = {sE.x - tE.x, sE.y - tE.y};
                coop_e = fmaf(ep.x, ep.x, ep.y * ep.y);
                coop_e8 = fmaf(em.x, em.x, em.y * em.y);
                const cpx sO = {w.x + pw.x, w.y - pw.y}, dO = {w.x - pw.x, w.y + pw.y};
                const cpx tO = cmulc(dO, c3.x, c3.y);
                const cpx op = {sO.x + tO.x, sO.y + tO.y};
                coop_o = fmaf(op.x, op.x, op.y * op.y);
            }
            coop_k = __float_as_int(c3.z);
        }
#endif
        // untangle the two real columns (rows k1 = 0..16, factor 2 kept) and twiddle by W512^(n2 k1)
        cpx ra[16], rb[16];  // index k1 = 1..15 used
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) {
            const cpx zk = z[k1], zm = z[32 - k1];
            const cpx a = {zk.x + zm.x, zk.y - zm.y};
            const cpx b = {zk.y + zm.y, zm.x - zk.x};
            const float4 t = s_tw1[(k1 - 1) * 8 + c];
            ra[k1] = cmulc(a, t.x, t.y);
            rb[k1] = cmulc(b, t.z, t.w);
        }
#ifndef F512_COOP
        const float qa = 2.f * z[0].x, qb = 2.f * z[0].y, pa = 2.f * z[16].x, pb = 2.f * z[16].y;
#endif

        F512_PIN(ra); F512_PIN(rb);
        F512_STAMP(3);
        F512_PIN(ra); F512_PIN(rb);
        // ---- exchange round A: units 0..7 (unit 0 = packed rows 0/16, units 1..7 = rows 1..7) ----
        float* xb = wbuf + f * 256;
        cpx u0[16], u1[16];
        {
#ifndef F512_COOP
            const int s0x = sigma_hi;  // unit 0: (0 >> 1) ^ sigma_hi
            *reinterpret_cast<float2*>(xb + 4 * ((c >> 1) ^ s0x) + 2 * (c & 1)) = make_float2(qa + pa, qb + pb);
            *reinterpret_cast<float2*>(xb + 4 * ((4 + (c >> 1)) ^ s0x) + 2 * (c & 1)) = make_float2(qa - pa, qb - pb);
#endif
            // (slot 0 of lane 0 -- "row 0" -- is not written with the cooperative unit: that FFT16 runs on whatever the
            //  buffer holds and its outputs are overwritten / left out below)
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) {
                const int sg = (k1 >> 1) ^ sigma_hi;
                *reinterpret_cast<float4*>(xb + k1 * 32 + 4 * (np ^ sg)) = make_float4(ra[k1].x, ra[k1].y, rb[k1].x, rb[k1].y);
            }
            F512_FENCE();
            const int sg = (c >> 1) ^ sigma_hi;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(xb + c * 32 + 4 * (i ^ sg));
                u0[2 * i] = {t.x, t.y};
                u0[2 * i + 1] = {t.z, t.w};
            }
            F512_FENCE();
            // ---- round B: units 8..15 = rows 8..15 (slot u - 8) ----
#pragma unroll
            for (int k1 = 8; k1 < 16; ++k1) {
                const int u = k1 - 8;
                const int sw = (u >> 1) ^ sigma_hi;
                *reinterpret_cast<float4*>(xb + u * 32 + 4 * (np ^ sw)) = make_float4(ra[k1].x, ra[k1].y, rb[k1].x, rb[k1].y);
            }
            F512_FENCE();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(xb + c * 32 + 4 * (i ^ sg));
                u1[2 * i] = {t.x, t.y};
                u1[2 * i + 1] = {t.z, t.w};
            }
            F512_FENCE();
        }

        F512_PIN(u0); F512_PIN(u1);
        F512_STAMP(4);
        F512_PIN(u0); F512_PIN(u1);
        // ---- pass 2: two complex FFT16 over n2 ----
        FFTReg<16>::run(u0);
        FFTReg<16>::run(u1);
        F512_PIN(u0); F512_PIN(u1);
        F512_STAMP(5);
        F512_PIN(u0); F512_PIN(u1);

        // power spectrum |X|^2 / 512: rows carry a factor 2 -> 1/2048.  That power of two is applied
        // (exactly) to the mel weights at plan time and to the energy sum once, not to every bin.
        float p0[16], p1[16];
        constexpr float S1 = 1.0f / 2048.0f;
#ifndef F512_COOP
        constexpr float S2 = 1.0f / 16.0f;   // the packed unit carries 8, not 2
#endif
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            p0[k] = fmaf(u0[k].x, u0[k].x, u0[k].y * u0[k].y);
            p1[k] = fmaf(u1[k].x, u1[k].x, u1[k].y * u1[k].y);
        }
#ifdef F512_COOP
        // Slot layout of p0 (unit 0): p0[k] -> bin c + 32 k (k < 8), bin 512 - 32 k - c (k >= 8).  Lane 0's slot 0 is the
        // unused "row 0": its values are garbage, the 9 bins 32 k they land on are overwritten by the cooperative unit
        // below, and they stay out of the energy.
        float e0 = p0[0], e1 = p1[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) { e0 += p0[k]; e1 += p1[k]; }
        e0 = c == 0 ? coop_e8 : e0;                      // lane 0: bin 256 instead
        e0 += coop_e + coop_o;
#else
        // Slot layout of p0 (unit 0): p0[k] -> bin c + 32 k (k < 8), bin 512 - 32 k - c (k >= 8).
        // Lane 0 (c = 0) owns bins 16 j instead: the even ones (32 k) fit the same slots, bin 256 is
        // slot 8, slots 9..15 repeat bins 224..32, and the 8 odd ones (16, 48, .., 240) go to podd[].
        float podd[8];
        // energy: every lane sums its 32 bins; lane 0's first unit is replaced by its 17 distinct bins below
        float e0 = p0[0], e1 = p1[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) { e0 += p0[k]; e1 += p1[k]; }
#pragma unroll
        for (int m = 0; m < 8; ++m) podd[m] = 0.f;
        if (c == 0) {
            // u0 = FFT16 of r[2m] + i r[2m+1]; finish the 32-point real FFT R[j] = X[16 j], j = 0..16
            // (factor 8 carried: 1/16 relative to the other units)
            float R[17];
            const float f0 = u0[0].x + u0[0].y, f16 = u0[0].x - u0[0].y;
            R[0] = S2 * 4.f * f0 * f0;
            R[16] = S2 * 4.f * f16 * f16;
            R[8] = S2 * 4.f * fmaf(u0[8].x, u0[8].x, u0[8].y * u0[8].y);
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                const cpx zj = u0[j], zq = u0[16 - j];
                const cpx e = {zj.x + zq.x, zj.y - zq.y};
                const cpx d = {zj.x - zq.x, zj.y + zq.y};
                const cpx o = {d.y, -d.x};
                const cpx tw = cmulc(o, DSP_COS32[j], -DSP_SIN32[j]);  // W32^j * o
                const cpx rp = {e.x + tw.x, e.y + tw.y};
                const cpx rm = {e.x - tw.x, e.y - tw.y};
                R[j] = S2 * fmaf(rp.x, rp.x, rp.y * rp.y);
                R[16 - j] = S2 * fmaf(rm.x, rm.x, rm.y * rm.y);
            }
            float es = R[0];
#pragma unroll
            for (int j = 1; j < 17; ++j) es += R[j];
            e0 = es;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                p0[k] = R[2 * k];
                p0[8 + k] = R[16 - 2 * k];
                podd[k] = R[2 * k + 1];
            }
        }
#endif
        // |X|^2 / 512 with the factor 2 of the rows = 1 / 2048; the mel weights and this sum both carry an extra
        // 2^32 (removed again after the logarithm), so that v_log_f32 never sees a denormal
        float energy = (S1 * 4294967296.0f) * frame_allreduce(e0 + e1);

        F512_PIN(p0); F512_PIN(p1);
        F512_STAMP(6);
        F512_PIN(p0); F512_PIN(p1);
        // ---- power spectrum -> LDS row of this frame: two base registers, immediate offsets ----
        float* ps = wbuf + f * F512_PS_STRIDE;
        {
            float* lo = ps + c;           // bins c + 32 k           (+8 for unit 1)
            float* hi = ps + 32 - c;      // bins 512 - 32 k - c = (32 - c) + 32 (15 - k)   (-8 for unit 1)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                lo[32 * k] = p0[k];
                hi[32 * (7 - k)] = p0[8 + k];
                lo[32 * k + 8] = p1[k];
                hi[32 * (7 - k) - 8] = p1[8 + k];
            }
#ifdef F512_COOP
            F512_FENCE();                                // after lane 0's garbage slot: the bins 32 k are rewritten here
            ps[32 * coop_k] = coop_e;                    // X[32 k]
            ps[32 * coop_k + 16] = coop_o;               // X[16 + 32 k]
            // bin 256 from lane 0; the other lanes clear the row padding 257..263 (zero-weight mel taps may read it, so
            // it must hold finite values; nothing else ever writes these slots in the last frame's row)
            ps[256 + c] = c == 0 ? coop_e8 : 0.f;
#else
            if (c == 0) {
#pragma unroll
                for (int m = 0; m < 8; ++m) ps[32 * m + 16] = podd[m];
            } else {
                ps[256 + c] = 0.f;
            }
#endif
        }
        F512_FENCE();
        F512_STAMP(7);

        // ---- sparse mel triangles, log2.  Lane c owns one filter slot per iteration i; its weights sit in one LDS
        //      row, the spectrum bins of a filter are contiguous from a 16-byte aligned start, so both are read as
        //      b128 blocks of 4 taps.  Weights carry 2^21 = 2^32 / 2048; an all-zero filter gives log2(eps 2^32) = -20
        //      (base.py:30, eps = 2^-52). ----
        float lm[NI];
        {
            const float4* wrow = reinterpret_cast<const float4*>(s_melw + c * P.melw_row);
            f512_static_for<0, NI>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const float4* pb = reinterpret_cast<const float4*>(ps + __float_as_int(smem[P.off_mels + i * 8 + c]));
                float acc0 = 0.f, acc1 = 0.f;
                if constexpr (CAPS != 0) {
                    constexpr int nb = f512_cap(CAPS, i), w0 = f512_cap_prefix(CAPS, i);
#pragma unroll
                    for (int b = 0; b < nb; ++b) {
                        const float4 w = wrow[w0 + b], q = pb[b];
                        acc0 = fmaf(w.x, q.x, acc0);
                        acc1 = fmaf(w.y, q.y, acc1);
                        acc0 = fmaf(w.z, q.z, acc0);
                        acc1 = fmaf(w.w, q.w, acc1);
                    }
                } else {
                    const int nb = P.nb4[i];
#pragma unroll 2
                    for (int b = 0; b < nb; ++b) {
                        const float4 w = wrow[b], q = pb[b];
                        acc0 = fmaf(w.x, q.x, acc0);
                        acc1 = fmaf(w.y, q.y, acc1);
                        acc0 = fmaf(w.z, q.z, acc0);
                        acc1 = fmaf(w.w, q.w, acc1);
                    }
                    wrow += nb;
                }
                const float acc = acc0 + acc1;
                const float l2 = __builtin_amdgcn_logf(acc);
                lm[i] = acc == 0.f ? -20.f : l2;
            });
        }
        F512_FENCE();
        F512_PIN(lm);
        F512_STAMP(8);
        F512_PIN(lm);

        // ---- DCT-II * lifter.  The filter of slot (c, u) and the one of slot (7 - c, NI - 1 - u) are mirror images
        //      (j and M - 1 - j), and DCT-II rows are (anti)symmetric under that mirror: even coefficients see the sum
        //      of the two log energies, odd ones the difference -- half the multiply-adds.  Register r of lane c holds
        //      coefficient r ^ cbase(c) (even r <-> even coefficient), see f512_cbase. ----
        float cep[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) cep[k] = 0.f;
        constexpr int NU = (NI + 1) / 2;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float a = lm[u], b = dpp_f32<0x141>(lm[NI - 1 - u]);
            const float sm = a + b, df = a - b;
            const float4* dr = reinterpret_cast<const float4*>(s_dct + (u * 8 + c) * 20);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = dr[q];
                cep[4 * q + 0] = fmaf(t.x, sm, cep[4 * q + 0]);
                cep[4 * q + 1] = fmaf(t.y, df, cep[4 * q + 1]);
                cep[4 * q + 2] = fmaf(t.z, sm, cep[4 * q + 2]);
                cep[4 * q + 3] = fmaf(t.w, df, cep[4 * q + 3]);
            }
        }
        // reduce-scatter over the frame's 8 lanes: 8 + 4 + 2 DPP adds, lane c ends with coefficients cbase(c), cbase(c) + 1
        float h8[8], h4[4];
#pragma unroll
        for (int r = 0; r < 8; ++r) h8[r] = cep[r] + dpp_f32<0x141>(cep[8 + r]);   // partner c ^ 7
#pragma unroll
        for (int r = 0; r < 4; ++r) h4[r] = h8[r] + dpp_f32<0x4E>(h8[4 + r]);      // partner c ^ 2
        v0 = h4[0] + dpp_f32<0xB1>(h4[2]);                                         // partner c ^ 1
        v1 = h4[1] + dpp_f32<0xB1>(h4[3]);
        {
            const float2 bias = *reinterpret_cast<const float2*>(smem + P.off_bias + 2 * c);   // removes the 2^32
            v0 += bias.x;
            v1 += bias.y;
        }
        if (P.append_energy) {
            constexpr float LN2 = 0.69314718055994530942f, LN_EPS = -36.04365338911715f;   // ln(2^-52), base.py:26
            const float le = fmaf(__builtin_amdgcn_logf(energy), LN2, -32.f * LN2);
            if (c == 0) v0 = energy == 0.f ? LN_EPS : le;
        }

        F512_STAMP(9);
        if constexpr (!FD) {
            // ---- store: lane c writes its two adjacent coefficients ----
            const int cb = ((c & 1) << 1) ^ ((c & 2) << 1) ^ ((c & 4) ? 14 : 0);
            const int t = t0 + f;
            // flat grouping: rows of the next utterance follow directly, valid while inside the batch
            const bool row_ok = (!RAGGED && P.flat) ? (grp.row0 + t < bg.total_frames) : (t < T);
            if (row_ok) {
                float* o = out + (grp.row0 + t) * ld_out + cb;
                if (cb + 1 < P.C) {
                    f512_f2u v;
                    v.x = v0;
                    v.y = v1;
                    *reinterpret_cast<f512_f2u*>(o) = v;
                } else if (cb < P.C) {
                    o[0] = v0;
                }
            }
        }
        } while (0);
        if constexpr (FD) {
            // ---- fused delta: window rows 0..7 = previous group, 8..15 = this group (or, in the final iteration,
            //      the next wave's first group from the halo area); rows 4..11 are emitted ----
            const int pr = (((c & 1) << 1) ^ ((c & 2) << 1) ^ ((c & 4) ? 14 : 0)) >> 1;   // coefficient pair of this lane; 7 = none
            const bool tail = r >= run.n_groups;
            if (r == 0) {   // leave the first group for the wave before this one; the workgroup's only barrier
                if (pr < 7) *reinterpret_cast<float2*>(fd_halo + (wid * 8 + f) * 14 + 2 * pr) = make_float2(v0, v1);
                __syncthreads();
            }
            if (tail) {
                const int nw = wid + 1 < WAVES ? wid + 1 : wid;   // no next wave: rows 8.. lie outside the workgroup, never used
                const float2 hv = *reinterpret_cast<const float2*>(fd_halo + (nw * 8 + f) * 14 + 2 * (pr < 7 ? pr : 0));
                v0 = hv.x;
                v1 = hv.y;
            }
            const bool emit = r > 0 || run.first_frame == run.wg_first;   // first iteration: only the workgroup's first wave has rows due
            if (emit) {
                float* const tx = wbuf;          // [16][16] cepstra of the window
                float* const td = wbuf + 256;    // [16][16] their deltas (rows 2..13)
                *reinterpret_cast<float2*>(tx + f * 16 + 2 * pr) = make_float2(fd_p0, fd_p1);
                *reinterpret_cast<float2*>(tx + (8 + f) * 16 + 2 * pr) = make_float2(v0, v1);
                const int Fw = run.first_frame + 8 * (r - 1);                 // flat frame of window row 0 (>= -8)
                // row of the window where an utterance starts (16 or more: none): edge replication stops there
                const int Tn = (int)bg.uniform_frames;
                const int tstart = (Fw + Tn) - f512_div(Fw + Tn, P.t_magic, P.t_shift) * Tn;   // (Fw mod T), Fw >= -8 > -T
                const int sb = tstart == 0 ? 0 : Tn - tstart;
                constexpr int N = FD;
                const float inv_den = P.fd_inv_den;
                const int C = P.C;
                const int Fi = Fw + 4 + f;                                    // the row this lane's frame slot emits
                float* const o = out + (int64_t)Fi * (3 * C) + 2 * pr;
                float2 ox, od, odd;
                bool row_ok;
                F512_FENCE();
                if ((sb == 0 || sb >= 16) && Fw + 4 >= run.wg_first && Fw + 12 <= run.wg_end) {
                    // No utterance boundary inside the window (23 of 25 windows of a 2 x 99-frame workgroup): nothing to
                    // clamp, rows at immediate offsets, and ONE LDS round trip -- the lane reads rows i - 2N .. i + 2N
                    // and forms the deltas of rows i - N .. i + N itself (same operations, same order as below).
                    const float* b = tx + f * 16 + 2 * pr;                    // window row (4 + f) - 4
                    float2 xr[4 * N + 1];
#pragma unroll
                    for (int k = 0; k <= 4 * N; ++k) xr[k] = *reinterpret_cast<const float2*>(b + (4 - 2 * N + k) * 16);
                    float2 dl[2 * N + 1];
#pragma unroll
                    for (int k = 0; k <= 2 * N; ++k) {                        // delta of row i - N + k: centre xr[N + k]
                        float a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int n = 1; n <= N; ++n) {
                            a0 = fmaf((float)n, xr[N + k + n].x - xr[N + k - n].x, a0);
                            a1 = fmaf((float)n, xr[N + k + n].y - xr[N + k - n].y, a1);
                        }
                        dl[k] = make_float2(a0 * inv_den, a1 * inv_den);
                        // rounded here, as when the value goes through memory: keeps -ffp-contract from folding the product
                        // into the subtraction below, so the rows equal the two-kernel path bit for bit
                        asm volatile("" : "+v"(dl[k].x), "+v"(dl[k].y));
                    }
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int n = 1; n <= N; ++n) {
                        a0 = fmaf((float)n, dl[N + n].x - dl[N - n].x, a0);
                        a1 = fmaf((float)n, dl[N + n].y - dl[N - n].y, a1);
                    }
                    ox = xr[2 * N];
                    od = dl[N];
                    odd = make_float2(a0 * inv_den, a1 * inv_den);
                    row_ok = true;
                } else {
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {   // delta of rows 2..9, then 10..13 (rows above 13 repeat row 13)
                        const int i = pass == 0 ? 2 + f : (10 + f < 13 ? 10 + f : 13);
                        const int lo = i >= sb ? sb : 0, hi = i < sb ? sb - 1 : 15;
                        float a0 = 0.f, a1 = 0.f;
#pragma unroll
                        for (int n = 1; n <= N; ++n) {
                            const int ip = i + n < hi ? i + n : hi, im = i - n > lo ? i - n : lo;
                            const float2 xp = *reinterpret_cast<const float2*>(tx + ip * 16 + 2 * pr);
                            const float2 xm = *reinterpret_cast<const float2*>(tx + im * 16 + 2 * pr);
                            a0 = fmaf((float)n, xp.x - xm.x, a0);
                            a1 = fmaf((float)n, xp.y - xm.y, a1);
                        }
                        *reinterpret_cast<float2*>(td + i * 16 + 2 * pr) = make_float2(a0 * inv_den, a1 * inv_den);
                    }
                    F512_FENCE();
                    const int i = 4 + f;
                    const int lo = i >= sb ? sb : 0, hi = i < sb ? sb - 1 : 15;
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int n = 1; n <= N; ++n) {
                        const int ip = i + n < hi ? i + n : hi, im = i - n > lo ? i - n : lo;
                        const float2 dp = *reinterpret_cast<const float2*>(td + ip * 16 + 2 * pr);
                        const float2 dm = *reinterpret_cast<const float2*>(td + im * 16 + 2 * pr);
                        a0 = fmaf((float)n, dp.x - dm.x, a0);
                        a1 = fmaf((float)n, dp.y - dm.y, a1);
                    }
                    ox = *reinterpret_cast<const float2*>(tx + i * 16 + 2 * pr);
                    od = *reinterpret_cast<const float2*>(td + i * 16 + 2 * pr);
                    odd = make_float2(a0 * inv_den, a1 * inv_den);
                    row_ok = Fi >= run.wg_first && Fi < run.wg_end;
                }
                if (row_ok) {
                    if (2 * pr + 1 < C) {
                        f512_f2u v;
                        v.x = ox.x; v.y = ox.y;
                        *reinterpret_cast<f512_f2u*>(o) = v;
                        v.x = od.x; v.y = od.y;
                        *reinterpret_cast<f512_f2u*>(o + C) = v;
                        v.x = odd.x; v.y = odd.y;
                        *reinterpret_cast<f512_f2u*>(o + 2 * C) = v;
                    } else if (2 * pr < C) {
                        o[0] = ox.x;
                        o[C] = od.x;
                        o[2 * C] = odd.x;
                    }
                }
            }
            fd_p0 = v0;
            fd_p1 = v1;
        }
        F512_FENCE();
        F512_STAMP(10);
    }
#ifdef F512_STAMPS
    stamp_acc_[11] = f512_clock() - stamp_entry_;
    stamp_acc_[12] = (unsigned int)__builtin_amdgcn_s_memrealtime() - stamp_rt0_;
    stamp_acc_[13] = stamp_loop0_ - stamp_entry_;
    stamp_acc_[14] = 1;
    if ((tid & 63) == 0) {
        const int slot = ((int)blockIdx.x * WAVES + wid) & 8191;
        for (int i = 0; i < F512_NSTAMP; ++i) f512_stamp_sum[slot * F512_NSTAMP + i] += stamp_acc_[i];
    }
#endif
}

// Ragged batches: group_off[b] = sum_{i<b} ceil(T_i / 2^shift) (exclusive prefix, single block), then the
// utterance of every group.  Both are tiny next to the main kernel and run on the same stream.
__global__ __launch_bounds__(1024) void f512_group_prefix_kernel(const int64_t* __restrict__ frame_off, int32_t n_utt,
                                                                 int32_t shift, int32_t* __restrict__ group_off,
                                                                 int32_t* __restrict__ group_utt = nullptr,
                                                                 int32_t tile_shift = 0,
                                                                 int64_t* __restrict__ tile_off = nullptr,
                                                                 double* __restrict__ zero_stats = nullptr) {
    // optional second table in the same launch: tile_off[b] = sum_{i<b} ceil(T_i / 2^tile_shift) (the delta pass)
    __shared__ int32_t wsum[16], wsum_t[16];
    const int tid = threadIdx.x;
    const int per = (n_utt + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, n_utt);
    const int64_t rnd = ((int64_t)1 << shift) - 1, rnd_t = ((int64_t)1 << tile_shift) - 1;
    int32_t sum = 0, sum_t = 0;
    for (int b = lo; b < hi; ++b) {
        const int64_t T = frame_off[b + 1] - frame_off[b];
        sum += (int32_t)((T + rnd) >> shift);
        sum_t += (int32_t)((T + rnd_t) >> tile_shift);
        if (zero_stats != nullptr) { zero_stats[2 * b] = 0.0; zero_stats[2 * b + 1] = 0.0; }
    }
    // inclusive scan of the per-thread sums: inside each wave with shuffles, across the 16 waves through LDS
    int32_t inc = sum, inc_t = sum_t;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t v = __shfl_up(inc, off, 64), vt = __shfl_up(inc_t, off, 64);
        if (lane >= off) { inc += v; inc_t += vt; }
    }
    if (lane == 63) { wsum[w] = inc; wsum_t[w] = inc_t; }
    __syncthreads();
    int32_t before = 0, before_t = 0, total = 0, total_t = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int32_t a = wsum[k], at = wsum_t[k];
        if (k < w) { before += a; before_t += at; }
        total += a;
        total_t += at;
    }
    inc += before;
    inc_t += before_t;
    int32_t run = inc - sum;
    int64_t run_t = inc_t - sum_t;
    for (int b = lo; b < hi; ++b) {
        group_off[b] = run;
        const int64_t T = frame_off[b + 1] - frame_off[b];
        const int32_t n = (int32_t)((T + rnd) >> shift);
        if (group_utt != nullptr)   // small batches: fill the group -> utterance table in the same launch
            for (int32_t g = 0; g < n; ++g) group_utt[run + g] = b;
        run += n;
        if (tile_off != nullptr) {
            tile_off[b] = run_t;
            run_t += (T + rnd_t) >> tile_shift;
        }
    }
    if (tid == 1023) {
        group_off[n_utt] = total;
        if (tile_off != nullptr) tile_off[n_utt] = total_t;
    }
}

// Builds both ragged index tables on `st`: one launch for small batches, prefix + parallel fill otherwise.
static inline void f512_build_group_tables(const int64_t* frame_off, int32_t n_utt, int32_t shift,
                                           int32_t* group_off, int32_t* group_utt, hipStream_t st,
                                           int64_t* tile_off = nullptr, double* zero_stats = nullptr);

__global__ __launch_bounds__(256) void f512_group_fill_kernel(const int32_t* __restrict__ group_off, int32_t n_utt,
                                                              int32_t* __restrict__ group_utt) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < n_utt; b += gridDim.x * blockDim.x)
        for (int g = group_off[b]; g < group_off[b + 1]; ++g) group_utt[g] = b;
}

static inline void f512_build_group_tables(const int64_t* frame_off, int32_t n_utt, int32_t shift,
                                           int32_t* group_off, int32_t* group_utt, hipStream_t st,
                                           int64_t* tile_off, double* zero_stats) {
    if (n_utt <= 4096) {
        f512_group_prefix_kernel<<<1, 1024, 0, st>>>(frame_off, n_utt, shift, group_off, group_utt, DT_SHIFT, tile_off, zero_stats);
        return;
    }
    f512_group_prefix_kernel<<<1, 1024, 0, st>>>(frame_off, n_utt, shift, group_off, nullptr, DT_SHIFT, tile_off, zero_stats);
    const int fill_blocks = (int)((n_utt + 255) / 256 < 1024 ? (n_utt + 255) / 256 : 1024);
    f512_group_fill_kernel<<<fill_blocks, 256, 0, st>>>(group_off, n_utt, group_utt);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// More than 48 filters on 257 bins means filters of one or two bins (the first one is the DC bin alone at 16 kHz):
// their logarithms amplify the fp32 noise floor of single bins, where this kernel measured 1.5e-4 against the
// generic kernel's 5e-5 on the same data (profiles/r2_parity_measured.json) -- such plans stay on the generic kernel.
static inline bool fast512_shape_ok(const dsp_plan_desc* d) {
    return d->nfft == 512 && d->frame_len <= 512 && (d->frame_step % 2) == 0 && d->frame_step >= 2 &&
           7 * d->frame_step + 512 + 16 <= F512_WAVE_FLOATS && d->nfilt >= 1 &&
           d->nfilt <= 8 * F512_MAX_NI && d->numcep >= 1 && d->numcep <= 16;
}

// The DCT half of the kernel relies on D[k][M - 1 - j] = (-1)^k D[k][j] (true of every DCT-II * lifter matrix);
// a caller-supplied matrix without that symmetry runs on the generic kernel.
static inline bool fast512_dct_is_mirror_symmetric(const dsp_plan_desc* d) {
    const int M = d->nfilt, C = d->numcep;
    for (int k = 0; k < C; ++k) {
        double mx = 0.0;
        for (int j = 0; j < M; ++j) mx = std::max(mx, (double)std::fabs(d->h_dct[(size_t)k * M + j]));
        const double sgn = (k & 1) ? -1.0 : 1.0;
        for (int j = 0; j < M; ++j)
            if (std::fabs((double)d->h_dct[(size_t)k * M + (M - 1 - j)] - sgn * (double)d->h_dct[(size_t)k * M + j]) > 2e-6 * mx) return false;
    }
    return true;
}

static inline int fast512_plan_init(dsp_plan* p, const dsp_plan_desc* d, const int32_t* mel_off) {
    p->d_fast = nullptr;
    if (!fast512_shape_ok(d) || !d->h_dct || !fast512_dct_is_mirror_symmetric(d)) return DSP_OK;
    const int M = d->nfilt, C = d->numcep, L = d->frame_len;
    const int ni_real = (M + 7) / 8, nrows = (L + 15) / 16;
    int variant, NI;  // template instantiation: <NROWS, NI, CAPS>
    // The wave stages every sample its pass-1 rows will touch (16 x NROWS per frame, window zero beyond
    // L), so no multiply ever sees stale LDS.  +1 vector: ragged batches stage from an aligned start.
    const int span25 = 7 * d->frame_step + 400, span32 = 7 * d->frame_step + 512;
    const int nstage = ((span25 + 3) / 4 + 1 + 63) / 64;
    if (nrows <= 25 && ni_real <= 4 && nstage <= 6) { variant = 0; NI = 4; }
    else if (nrows <= 25 && ni_real <= 5 && nstage <= 6) { variant = 1; NI = 5; }
    else { variant = 2; NI = F512_MAX_NI; }
    if (const char* fv = getenv("DSP_F512_FORCE_CATCHALL")) {  // debugging aid: run any plan on the catch-all instantiation
        if (fv[0] == '1') { variant = 2; NI = F512_MAX_NI; }
    }
    std::vector<float> win(512, 0.f);
    for (int n = 0; n < L; ++n) win[n] = d->h_window[n];
#ifdef F512_COOP
    auto col_pair = [](int c) { return c < 4 ? c : 11 - c; };   // pass-1 column pair of lane c (see the kernel)
#else
    auto col_pair = [](int c) { return c; };
#endif
    std::vector<float> tw1(15 * 8 * 4);
    for (int k1 = 1; k1 < 16; ++k1)
        for (int c = 0; c < 8; ++c)
            for (int h = 0; h < 2; ++h) {
                const double a = -2.0 * M_PI * (double)((2 * col_pair(c) + h) * k1) / 512.0;
                tw1[((k1 - 1) * 8 + c) * 4 + 2 * h] = (float)cos(a);
                tw1[((k1 - 1) * 8 + c) * 4 + 2 * h + 1] = (float)sin(a);
            }
    // constants of the cross-lane FFT8s of rows 0 and 16, 16 floats per lane (layout: see the kernel)
    std::vector<float> coop(8 * 16, 0.f);
    for (int c = 0; c < 8; ++c) {
        const int n = col_pair(c), n2 = n >> 2, n1 = (n >> 1) & 1, n0 = n & 1;
        const int k = n0 * 4 + n1 * 2 + n2;                       // DIF: element n ends as output bit-reverse(n)
        float* t = &coop[(size_t)c * 16];
        t[0] = n2 ? -1.f : 1.f;
        t[1] = n1 ? -1.f : 1.f;
        t[2] = n0 ? -1.f : 1.f;
        const double a1 = 2.0 * M_PI * (double)(n & 3) / 8.0, a2 = 2.0 * M_PI * (double)n0 / 4.0;
        t[4] = n2 ? (float)cos(a1) : 1.f;  t[5] = n2 ? (float)-sin(a1) : 0.f;     // W8^(n & 3) on the upper half
        t[6] = n1 ? (float)cos(a2) : 1.f;  t[7] = n1 ? (float)-sin(a2) : 0.f;     // W4^(n & 1)
        const double ap = 2.0 * M_PI * (double)n / 16.0, ae = 2.0 * M_PI * (double)k / 16.0,
                     ao = 2.0 * M_PI * (double)(2 * k + 1) / 32.0;
        t[8] = (float)cos(ap);   t[9] = (float)-sin(ap);                           // W16^n
        t[10] = (float)-sin(ae); t[11] = (float)-cos(ae);                          // -i W16^k
        t[12] = (float)-sin(ao); t[13] = (float)-cos(ao);                          // -i W32^(2k+1)
        memcpy(&t[14], &k, 4);
    }
    // Filter slots: lane c, iteration i.  Filters are dealt in mirror pairs (j, M - 1 - j): pair p sits in slot
    // (p % 8, p / 8) and its mirror image in slot (7 - p % 8, NI - 1 - p / 8); with an odd NI the middle iteration holds
    // 4 pairs (lanes c and 7 - c).  The middle filter of an odd M fills both slots of its pair (half weight each).
    // Iteration i so holds 8 filters of similar length: 8 i .. 8 i + 7 from below or their mirror images from above.
    std::vector<int> slot((size_t)8 * NI, -1);
    const int n_pairs = (M + 1) / 2, full = NI / 2;
    if (n_pairs > 4 * NI) return DSP_OK;
    for (int pr = 0; pr < n_pairs; ++pr) {
        const int ja = pr, jb = M - 1 - pr;
        int c, u;
        if (pr < 8 * full) { c = pr % 8; u = pr / 8; }
        else { c = pr - 8 * full; u = full; }
        slot[(size_t)u * 8 + c] = ja;
        slot[(size_t)(NI - 1 - u) * 8 + (7 - c)] = jb;
    }
    const int NU = (NI + 1) / 2;
    const double LN2 = 0.69314718055994530942;
    std::vector<float> dct((size_t)NU * 8 * 20, 0.f);  // [unit][lane] rows of 16 (+4 pad: 5 x 16 B, conflict-free b128)
    std::vector<double> bias_acc(16, 0.0);
    for (int u = 0; u < NU; ++u)
        for (int c = 0; c < 8; ++c) {
            const int ja = slot[(size_t)u * 8 + c], jb = slot[(size_t)(NI - 1 - u) * 8 + (7 - c)];
            if (ja < 0) continue;
            // the pair is visited from both of its lanes when it lies in the middle iteration of an odd NI, and a
            // filter that is its own mirror image enters as a + b = 2 a
            const bool both = (NI & 1) && u == NI / 2;
            const double wgt = (both ? 0.5 : 1.0) * (ja == jb ? 0.5 : 1.0);
            for (int r = 0; r < 16; ++r) {
                const int k = r ^ f512_cbase(c);
                if (k >= C) continue;
                const float t = (float)(wgt * LN2 * (double)d->h_dct[(size_t)k * M + ja]);
                dct[((size_t)u * 8 + c) * 20 + r] = t;
                // every log2 carries +32 (weights scaled by 2^32): sums see +64, differences nothing
                if ((r & 1) == 0) bias_acc[k] -= 64.0 * (double)t;
            }
        }
    std::vector<float> bias(16, 0.f);
    for (int c = 0; c < 8; ++c)
        for (int r = 0; r < 2; ++r) bias[2 * c + r] = (float)bias_acc[f512_cbase(c) + r];
    Fast512Plan* fp = new Fast512Plan();
    memset(fp, 0, sizeof(*fp));
    // mel: per lane one row of weights (iterations back to back); every filter starts on a bin that is a multiple
    // of 4 (leading zero weights absorb the misalignment) and every iteration is padded to whole 16-byte blocks.
    // Padding weights are 0; padded reads stay inside the 264-float frame row.
    int caps = variant == 0 ? 2 : (variant == 1 ? 1 : 0);
    int row_blocks = 0;
    for (int i = 0; i < NI; ++i) {
        int len = 0;
        for (int c = 0; c < 8; ++c) {
            const int j = slot[(size_t)i * 8 + c];
            if (j < 0) continue;
            const int need = (d->h_mel_start[j] & 3) + d->h_mel_count[j];
            if (need > len) len = need;
        }
        fp->P.nb4[i] = (len + 3) / 4;
        if (caps != 0 && fp->P.nb4[i] > f512_cap(caps, i)) caps = 0;
    }
    if (const char* fc = getenv("DSP_F512_NOCAPS")) {  // A/B aid: run-time block counts for every plan
        if (fc[0] == '1') caps = 0;
    }
    for (int i = 0; i < NI; ++i) {
        if (caps != 0) fp->P.nb4[i] = f512_cap(caps, i);
        row_blocks += fp->P.nb4[i];
    }
    int melw_row = row_blocks > 0 ? 4 * row_blocks : 4;
    if (((melw_row / 4) & 1) == 0) melw_row += 4;  // odd number of 16-byte slots per row
    std::vector<float> melw((size_t)8 * melw_row, 0.f), mels((size_t)NI * 8, 0.f);
    for (int c = 0; c < 8; ++c) {
        int pos = 0;
        for (int i = 0; i < NI; ++i) {
            const int j = slot[(size_t)i * 8 + c], len = 4 * fp->P.nb4[i];
            int32_t start = 0;
            if (j >= 0) {
                start = d->h_mel_start[j] & ~3;
                int lead = d->h_mel_start[j] - start;
                if (start + len > F512_PS_STRIDE) {  // keep the padded read inside the row
                    const int shift = (start + len - F512_PS_STRIDE + 3) / 4 * 4;
                    start -= shift;
                    lead += shift;
                }
                for (int s2 = 0; s2 < d->h_mel_count[j]; ++s2)
                    melw[(size_t)c * melw_row + pos + lead + s2] = d->h_mel_weights[mel_off[j] + s2] * 2097152.0f;  // 2^32 / 2048
            }
            memcpy(&mels[(size_t)i * 8 + c], &start, 4);
            pos += len;
        }
    }
    auto pad64 = [](size_t n) { return (n + 63) / 64 * 64; };
    const size_t o_tw1 = 512, o_dct = o_tw1 + tw1.size(), o_melw = pad64(o_dct + dct.size());
    const size_t o_mels = o_melw + melw.size(), o_bias = o_mels + mels.size();
    const size_t o_coop = (o_bias + bias.size() + 3) / 4 * 4;
    const size_t total = pad64(o_coop + coop.size());
    std::vector<float> blob(total, 0.f);
    memcpy(blob.data(), win.data(), 512 * 4);
    memcpy(blob.data() + o_tw1, tw1.data(), tw1.size() * 4);
    memcpy(blob.data() + o_dct, dct.data(), dct.size() * 4);
    memcpy(blob.data() + o_melw, melw.data(), melw.size() * 4);
    memcpy(blob.data() + o_mels, mels.data(), mels.size() * 4);
    memcpy(blob.data() + o_bias, bias.data(), bias.size() * 4);
    memcpy(blob.data() + o_coop, coop.data(), coop.size() * 4);
    if (dsp_table_alloc_copy(reinterpret_cast<void**>(&fp->d_tables), blob.data(), total * 4) != hipSuccess) {
        delete fp;
        return DSP_EHIP;
    }
    fp->P.tables = fp->d_tables;
    fp->P.tab_floats = (int32_t)total;
    fp->P.off_tw1 = (int32_t)o_tw1; fp->P.off_dct = (int32_t)o_dct;
    fp->P.off_melw = (int32_t)o_melw; fp->P.off_mels = (int32_t)o_mels; fp->P.off_bias = (int32_t)o_bias;
    fp->P.off_coop = (int32_t)o_coop;
    fp->P.melw_row = melw_row;
    fp->P.L = L; fp->P.S = d->frame_step; fp->P.M = M; fp->P.C = C;
    fp->P.append_energy = d->append_energy ? 1 : 0;
    fp->P.preemph = d->preemph;
    fp->P.span_vec = ((variant == 2 ? span32 : span25) + 3) / 4;
    fp->variant = variant;
    fp->caps = caps;
    p->d_fast = fp;
    return DSP_OK;
}

static inline void fast512_plan_free(dsp_plan* p) {
    Fast512Plan* fp = static_cast<Fast512Plan*>(p->d_fast);
    if (!fp) return;
    dsp_table_free(fp->d_tables, p->dry_run);
    delete fp;
    p->d_fast = nullptr;
}

// The fast kernel serves (a) dense batches with N % 4 == 0 and (b) ragged batches (any lengths and
// offsets); the buffer itself must start 16-byte aligned (8 for int16) either way.
static inline bool fast512_applicable(const dsp_plan* p, const BatchGeom& bg, const void* d_wave, int dtype) {
    if (!p->d_fast) return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_wave);
    if ((a % (dtype == DSP_WAVE_I16 ? 8 : 16)) != 0) return false;
    if (bg.uniform_samples > 0) {
        if ((bg.uniform_samples % 4) != 0) return false;
        return bg.uniform_samples <= 0x3fffffff && ((bg.uniform_frames + 7) / 8) * bg.n_utt <= 0x3fffffff;  // 32-bit indexing
    }
    return bg.total_frames / 8 + bg.n_utt <= 0x3fffffff;
}

template <int NROWS, int NI, int CAPS, int NSTAGE, int DTYPE, bool RAGGED>
static int fast512_launch_k(const F512Params& P, const void* d_wave, const BatchGeom& bg, float* d_out,
                            int64_t ld_out, int64_t groups_bound, hipStream_t st) {
    const size_t lds = ((size_t)P.tab_floats + (size_t)F512_WAVES * F512_WAVE_FLOATS) * sizeof(float);
    // balanced persistent grid: every wave runs the same number of groups (no ragged last round)
    const int64_t cap = (int64_t)dsp_cu_count() * (16 / F512_WAVES);  // CUs x resident workgroups (<= 16 waves per CU)
    int64_t blocks = (groups_bound + F512_WAVES - 1) / F512_WAVES;
    static const int grid_mode = [] { const char* e = getenv("DSP_F512_GRID"); return e ? atoi(e) : 1; }();
    if (blocks > cap) {
        if (grid_mode == 0) {   // every wave the same number of groups (fewer, fuller workgroups)
            const int64_t rounds = (blocks + cap - 1) / cap;
            blocks = (blocks + rounds - 1) / rounds;
        } else {
            blocks = cap;       // every CU fully occupied; the partial last round is dealt wave-major
        }
    }
    auto k = mfcc512_kernel<NROWS, NI, CAPS, NSTAGE, DTYPE, F512_WAVES, RAGGED>;
    static size_t granted[DSP_MAX_DEVICES] = {};  // dynamic-LDS limit already raised, per device
    if (dsp_ensure_dynamic_lds((const void*)k, lds, granted) != 0) return DSP_EHIP;
    k<<<(int)blocks, 64 * F512_WAVES, lds, st>>>(P, bg, d_wave, d_out, ld_out);
    return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
}

// Ragged index tables built by the caller in one launch together with its own (dsp_mfcc_delta_batch).
struct DspRaggedTables {
    int32_t* group_off = nullptr;   // [n_utt + 1] prefix of ceil(T_b / 2^shift)
    int32_t* group_utt = nullptr;   // utterance of every group
    int shift = 0;                  // 3: NFFT=512 kernel (8 frames per wave), 2: NFFT=1536 kernel
    // a second set for another group size (a dsp_layout holds the tables of the int16 VAD kernel's 8-frame groups beside
    // those of the 4-frame groups the other VAD kernels use, when the two differ)
    int32_t* group_off2 = nullptr;
    int32_t* group_utt2 = nullptr;
    int shift2 = 0;
};

template <int NROWS, int NI, int CAPS, int NSTAGE>
static int fast512_launch_t(F512Params P, const void* d_wave, int dtype, const BatchGeom& bg, float* d_out,
                            int64_t ld_out, hipStream_t st, const DspRaggedTables* pre = nullptr) {
    if (bg.uniform_samples > 0) {
        P.groups_per_utt = (bg.uniform_frames + 7) / 8;
        P.total_groups = P.groups_per_utt * bg.n_utt;
        f512_magic((uint32_t)bg.uniform_frames, P.t_magic, P.t_shift);
        f512_magic((uint32_t)P.groups_per_utt, P.g_magic, P.g_shift);
        // Flat grouping (groups of 8 cut from the flat frame sequence, a group may span the seam between two
        // utterances) wastes no frame slots at the end of an utterance: 99 frames are 13 groups of 8 otherwise
        // (5 % idle slots).  Needs vector-aligned hops and room for the seam's second segment in the wave buffer.
        // (the seam is as long as the samples pass 1 READS per frame, 16 x NROWS >= L: a frame's rows beyond L meet a
        // zero window, but 0 x NaN of the next utterance's samples would still poison the frame)
        const int seam = (16 * NROWS - P.S + 3) / 4 * 4;
        static const bool no_flat = getenv("DSP_F512_NOFLAT") != nullptr;   // A/B switch for tools/kbench.py
        if (!no_flat && (P.S % 4) == 0 && P.L > P.S && 16 * NROWS > P.S && bg.uniform_frames >= 8 && (bg.uniform_frames % 8) != 0 &&
            7 * P.S + 16 * NROWS + seam + 4 <= F512_WAVE_FLOATS && 7 * P.S + 16 * NROWS + seam <= 256 * (NSTAGE + 1) &&
            bg.total_frames + 8 <= 0x3fffffff) {
            P.flat = 1;
            P.seam_off = seam;
            P.total_groups = (bg.total_frames + 7) / 8;
        }
        if (dtype == DSP_WAVE_I16)
            return fast512_launch_k<NROWS, NI, CAPS, NSTAGE, DSP_WAVE_I16, false>(P, d_wave, bg, d_out, ld_out, P.total_groups, st);
        return fast512_launch_k<NROWS, NI, CAPS, NSTAGE, DSP_WAVE_F32, false>(P, d_wave, bg, d_out, ld_out, P.total_groups, st);
    }
    // ragged: build the group tables in a pooled, event-guarded workspace (no host sync)
    const int64_t bound = bg.total_frames / 8 + bg.n_utt;  // >= sum ceil(T_b / 8)
    DspWorkspace* w = nullptr;
    if (pre != nullptr && pre->shift == 3) {
        P.group_off = pre->group_off;
        P.group_utt = pre->group_utt;
    } else {
        const size_t ws_bytes = ((size_t)bg.n_utt + 1 + (size_t)bound) * sizeof(int32_t);
        w = dsp_workspace_pool().acquire(ws_bytes, st);
        if (!w) return DSP_EHIP;
        int32_t* group_off = static_cast<int32_t*>(w->ptr);
        int32_t* group_utt = group_off + bg.n_utt + 1;
        f512_build_group_tables(bg.frame_off, bg.n_utt, 3, group_off, group_utt, st);
        P.group_off = group_off;
        P.group_utt = group_utt;
    }
    int rc;
    if (dtype == DSP_WAVE_I16)
        rc = fast512_launch_k<NROWS, NI, CAPS, NSTAGE, DSP_WAVE_I16, true>(P, d_wave, bg, d_out, ld_out, bound, st);
    else
        rc = fast512_launch_k<NROWS, NI, CAPS, NSTAGE, DSP_WAVE_F32, true>(P, d_wave, bg, d_out, ld_out, bound, st);
    if (w != nullptr && dsp_workspace_pool().release(w, st) != 0 && rc == DSP_OK) rc = DSP_EHIP;
    return rc;
}

static inline int fast512_launch(const dsp_plan* p, const void* d_wave, int dtype, const BatchGeom& bg,
                                 float* d_out, int64_t ld_out, hipStream_t st, const DspRaggedTables* pre = nullptr) {
    const Fast512Plan* fp = static_cast<const Fast512Plan*>(p->d_fast);
    // exact instantiations for the common shapes, a padded catch-all otherwise (chosen at plan init)
    if (fp->variant == 0 && fp->caps == 2) return fast512_launch_t<25, 4, 2, 6>(fp->P, d_wave, dtype, bg, d_out, ld_out, st, pre);
    if (fp->variant == 0) return fast512_launch_t<25, 4, 0, 6>(fp->P, d_wave, dtype, bg, d_out, ld_out, st, pre);
    if (fp->variant == 1 && fp->caps == 1) return fast512_launch_t<25, 5, 1, 6>(fp->P, d_wave, dtype, bg, d_out, ld_out, st, pre);
    if (fp->variant == 1) return fast512_launch_t<25, 5, 0, 6>(fp->P, d_wave, dtype, bg, d_out, ld_out, st, pre);
    return fast512_launch_t<32, F512_MAX_NI, 0, 9>(fp->P, d_wave, dtype, bg, d_out, ld_out, st, pre);
}

// ------------------------------------------------------------------------------------------------
// fused MFCC + delta + delta-delta (FD instantiations): one launch writes the whole [sum T, 3 C] rows
// ------------------------------------------------------------------------------------------------
#define F512_FD_HALO_FLOATS (F512_WAVES * 8 * 14)

template <int NROWS, int NI, int CAPS, int NSTAGE>
static int fast512_launch_fused_t(F512Params P, const void* d_wave, int dtype, const BatchGeom& bg, int delta_n,
                                  float* d_out, hipStream_t st) {
    const int64_t T = bg.uniform_frames;
    const int64_t blocks = (int64_t)dsp_cu_count() * (16 / F512_WAVES);
    // every workgroup owns whole utterances and at least F512_WAVES groups of 8 frames; a window of 16 frames meets at
    // most one utterance boundary; the delta window reaches 2 N <= 4 frames back
    if (delta_n < 1 || delta_n > 2 || P.C > 14 || T < 8 * F512_WAVES - 7 || bg.n_utt < blocks) return 1;
    if ((P.S % 4) != 0 || P.L <= P.S || 16 * NROWS <= P.S) return 1;
    const int seam = (16 * NROWS - P.S + 3) / 4 * 4;
    if (7 * P.S + 16 * NROWS + seam + 4 > F512_WAVE_FLOATS || 7 * P.S + 16 * NROWS + seam > 256 * (NSTAGE + 1)) return 1;
    if (bg.total_frames + T + 16 > 0x3fffffff) return 1;
    P.flat = 1;
    P.seam_off = seam;
    P.groups_per_utt = (T + 7) / 8;
    P.total_groups = (bg.total_frames + 7) / 8;
    f512_magic((uint32_t)T, P.t_magic, P.t_shift);
    f512_magic((uint32_t)P.groups_per_utt, P.g_magic, P.g_shift);
    P.fd_n = delta_n;
    int den = 0;
    for (int i = 1; i <= delta_n; ++i) den += i * i;
    P.fd_inv_den = (float)(1.0 / (2.0 * den));
    const size_t lds = ((size_t)P.tab_floats + (size_t)F512_WAVES * F512_WAVE_FLOATS + F512_FD_HALO_FLOATS) * sizeof(float);
    if (lds > 80 * 1024) return 1;   // two workgroups per CU
    const int64_t ld = 3 * (int64_t)P.C;
    const int nblk = (int)blocks;
#define F512_FD_LAUNCH(DT, FDN)                                                                          \
    do {                                                                                                 \
        auto k = mfcc512_kernel<NROWS, NI, CAPS, NSTAGE, DT, F512_WAVES, false, FDN>;                    \
        static size_t granted[DSP_MAX_DEVICES] = {};                                                     \
        if (dsp_ensure_dynamic_lds((const void*)k, lds, granted) != 0) return DSP_EHIP;                  \
        k<<<nblk, 64 * F512_WAVES, lds, st>>>(P, bg, d_wave, d_out, ld);                                 \
    } while (0)
    if (dtype == DSP_WAVE_I16) {
        if (delta_n == 1) F512_FD_LAUNCH(DSP_WAVE_I16, 1); else F512_FD_LAUNCH(DSP_WAVE_I16, 2);
    } else {
        if (delta_n == 1) F512_FD_LAUNCH(DSP_WAVE_F32, 1); else F512_FD_LAUNCH(DSP_WAVE_F32, 2);
    }
#undef F512_FD_LAUNCH
    return hipGetLastError() == hipSuccess ? DSP_OK : DSP_EHIP;
}

// DSP_OK: launched; 1: this batch is not one the fused form serves (the caller takes the two-kernel path); < 0: HIP error.
static inline int fast512_launch_fused(const dsp_plan* p, const void* d_wave, int dtype, const BatchGeom& bg, int delta_n,
                                       float* d_out, hipStream_t st) {
    static const bool off = getenv("DSP_F512_NOFUSE") != nullptr;   // A/B switch
    if (off || !fast512_applicable(p, bg, d_wave, dtype) || bg.uniform_samples <= 0) return 1;
    const Fast512Plan* fp = static_cast<const Fast512Plan*>(p->d_fast);
    if (fp->variant == 0 && fp->caps == 2) return fast512_launch_fused_t<25, 4, 2, 6>(fp->P, d_wave, dtype, bg, delta_n, d_out, st);
    if (fp->variant == 0) return fast512_launch_fused_t<25, 4, 0, 6>(fp->P, d_wave, dtype, bg, delta_n, d_out, st);
    if (fp->variant == 1 && fp->caps == 1) return fast512_launch_fused_t<25, 5, 1, 6>(fp->P, d_wave, dtype, bg, delta_n, d_out, st);
    if (fp->variant == 1) return fast512_launch_fused_t<25, 5, 0, 6>(fp->P, d_wave, dtype, bg, delta_n, d_out, st);
    return fast512_launch_fused_t<32, F512_MAX_NI, 0, 9>(fp->P, d_wave, dtype, bg, delta_n, d_out, st);
}
