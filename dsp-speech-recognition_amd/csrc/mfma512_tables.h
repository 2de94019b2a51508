// Host-side tables of the matrix-pipe NFFT = 512 kernel (kernels_mfma512.h).  Plain C++ (no HIP): also compiled
// by tools/mfma512_tables_c.cpp for the CPU emulation of the kernel's operand maps (tools/mfma512_emul.py).
//
// The 512-point real DFT of a frame (sigproc.py:136-158 after :66-98) is two dense stages on
// v_mfma_f32_16x16x32_f16, every operand an fp16 (hi, lo) pair, three products per stage (hi*hi, hi*lo, lo*hi):
//   n = 16 n1 + n2, k = k1 + 32 k2
//   stage 1 (column n2):  Y[k1, n2] = sum_n1 A1[n2][k1, n1] x[16 n1 + n2],  A1 = window * W32^(n1 k1) * W512^(n2 k1)
//   stage 2 (row k1):     X[k1 + 32 k2] = sum_n2 W16^(n2 k2) Y[k1, n2]
// then |X|^2 -> mel (+ an all-ones "energy" row) as a bf16 (hi, lo) product, log2, and the DCT * lifter as one more
// fp16 product (base.py:8-32).  A tile is 16 consecutive frames of one utterance: lane l of the wave is
// (frame n = l & 15, group g = l >> 4); operand maps as in the MI355X guide (A[row l&15][k = 8 (l>>4) + j],
// B[k = 8 (l>>4) + j][col l&15], D[row 4 (l>>4) + i][col l&15]).
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#define M512_XBITS 10      // tile scale: |pre-emphasised sample| < 2^XBITS
#define M512_WSH 16        // mel weights carry 2^WSH / (NFFT * SA1^2): the log argument is never a denormal
#define M512_MAX_WBLOCKS 16

// row rho = 4 g + i of a stage-2 / power tile holds k2 = KAP[g][i]: the four lane groups of one register index
// cover one quarter of the spectrum (keeps the mel blocks sparse)
static const int M512_KAP[4][4] = {{0, 2, 4, 6}, {15, 13, 11, 9}, {1, 3, 5, 7}, {14, 12, 10, 8}};

// FFT bin of (slot s, k2 index kr).  Slot s >= 1 is DFT row k1 = s; slot 0 packs the two real rows k1 = 0 and 16.
// (0, 0) stands for bins 0 AND 256 together (their powers are summed; both carry weight zero in every eligible
// filterbank and weight one in the frame energy).
static inline int m512_bin(int s, int kr) {
    if (s == 0) return kr <= 7 ? 32 * kr : 16 + 32 * (15 - kr);
    return kr <= 7 ? s + 32 * kr : 512 - s - 32 * kr;
}

struct M512Layout {
    int32_t off_a1, off_a2, off_a2p, off_dm, off_w, off_rowsum;  // byte offsets into the blob (16-byte aligned)
    int32_t bytes;
    int32_t n_wblocks;                 // mel blocks stored (2 KB each: hi KB, lo KB): always M512_MAX_WBLOCKS
    int32_t wblock_step[M512_MAX_WBLOCKS], wblock_tile[M512_MAX_WBLOCKS];
    int32_t erow;                      // row of the all-ones energy filter (-1: none); filters follow around it
    int32_t n_mtiles;                  // row tiles of 16 filters (+ energy row): 2 or 3
    int32_t sa1_log2;                  // A1 carries 2^sa1_log2
    int32_t KR;                        // rows of 16 samples a frame occupies: ceil(min(L, 512) / 16)
    float z_log2_eps;                  // log2(eps) (base.py:26,30)
};

// The mel blocks the kernel multiplies (compile-time there: kernels_mfma512.h m512_has_block).
static inline bool m512_pattern_has(int n_mtiles, int step, int tile) {
    return n_mtiles <= 2 ? true : (tile == 1 || (tile == 0 && step < 2) || (tile == 2 && step >= 4));
}

static inline uint16_t m512_f2h(float f) {  // round-to-nearest-even, subnormals kept
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to >= 65520 -> inf
    if (x < 0x33000001u) return (uint16_t)sign;                // < 2^-25: rounds to zero
    int32_t e = (int32_t)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = e >= -14 ? 13 : 13 + (-14 - e);               // bits dropped from the 24-bit significand
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
    uint32_t h;
    if (e >= -14) h = ((uint32_t)(e + 15) << 10) + (q - 0x400u);  // q may carry into the exponent: still right
    else h = q;
    return (uint16_t)(sign | h);
}

static inline float m512_h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 31, m = h & 0x3ffu;
    float f;
    uint32_t x;
    if (e == 0) { f = ldexpf((float)m, -24); memcpy(&x, &f, 4); x |= sign; memcpy(&f, &x, 4); return f; }
    if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    memcpy(&f, &x, 4);
    return f;
}

static inline uint16_t m512_f2bf(float f) {  // round-to-nearest-even
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((x >> 16) | 0x40);
    x += 0x7fffu + ((x >> 16) & 1);
    return (uint16_t)(x >> 16);
}

static inline float m512_bf2f(uint16_t b) {
    uint32_t x = (uint32_t)b << 16;
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// One 16 x 32 operand block in lane order: lane l = (row l & 15, octet l >> 4) holds elements (octet, 0..7).
template <typename F>
static inline void m512_fill_block_f16(uint8_t* hi_kb, uint8_t* lo_kb, F value) {
    uint16_t* hi = reinterpret_cast<uint16_t*>(hi_kb);
    uint16_t* lo = reinterpret_cast<uint16_t*>(lo_kb);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const double v = value(l & 15, l >> 4, j);
            const float f = (float)v;
            const uint16_t h = m512_f2h(f);
            hi[l * 8 + j] = h;
            lo[l * 8 + j] = m512_f2h((float)(v - (double)m512_h2f(h)));
        }
}

template <typename F>
static inline void m512_fill_block_bf16(uint8_t* hi_kb, uint8_t* lo_kb, F value) {
    uint16_t* hi = reinterpret_cast<uint16_t*>(hi_kb);
    uint16_t* lo = reinterpret_cast<uint16_t*>(lo_kb);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
            const double v = value(l & 15, l >> 4, j);
            const uint16_t h = m512_f2bf((float)v);
            hi[l * 8 + j] = h;
            lo[l * 8 + j] = m512_f2bf((float)(v - (double)m512_bf2f(h)));
        }
}

// Returns 0 and fills blob / lay, or a negative reason when the plan does not fit the kernel:
//  -1 shape, -2 filterbank weight on bin 0 / 256, -3 too many mel blocks.
// window[L], mel as CSR (start / count / weights over FFT bins 0..256), dct [C, M] row-major with the lifter in it.
static inline int m512_build_tables(int L, int S, int nfft, int M, int C, int append_energy, const float* window,
                                    const int32_t* mel_start, const int32_t* mel_count, const float* mel_w,
                                    const float* dct, std::vector<uint8_t>& blob, M512Layout& lay) {
    memset(&lay, 0, sizeof(lay));
    if (nfft != 512 || L < 1 || S < 16 || (S % 16) != 0 || M < 1 || C < 1 || C > 16 || C > M) return -1;
    const int Lf = L < 512 ? L : 512;
    const int rows_e = M + (append_energy ? 1 : 0);
    if (rows_e > 48) return -1;
    lay.n_mtiles = rows_e <= 32 ? 2 : 3;   // the kernel is instantiated for two and three row tiles (unused rows: zero weights)
    lay.KR = (Lf + 15) / 16;
    lay.z_log2_eps = (float)log2(2.220446049250313e-16);
    const double PI = 3.14159265358979323846;

    // dense mel weights over bins, by ROW: filters in order with the energy row (all ones) at row erow -- row 31 when
    // there are more than 31 filters, so that it lives in row tile 1, the tile every step multiplies anyway
    lay.erow = append_energy ? (M >= 31 ? 31 : M) : -1;
    auto row_filter = [&](int r) -> int {   // -1: energy row, -2: no such row
        if (r >= rows_e) return -2;
        if (lay.erow < 0) return r;
        return r == lay.erow ? -1 : (r < lay.erow ? r : r - 1);
    };
    std::vector<double> Wf((size_t)M * 257, 0.0);
    {
        size_t o = 0;
        for (int f = 0; f < M; ++f) {
            for (int c = 0; c < mel_count[f]; ++c) {
                const int b = mel_start[f] + c;
                if (b < 0 || b > 256) return -1;
                Wf[(size_t)f * 257 + b] = mel_w[o + c];
            }
            o += mel_count[f];
            if (Wf[(size_t)f * 257 + 0] != 0.0 || Wf[(size_t)f * 257 + 256] != 0.0) return -2;
        }
    }
    std::vector<double> W((size_t)48 * 257, 0.0);
    for (int r = 0; r < rows_e; ++r) {
        const int f = row_filter(r);
        for (int b = 0; b <= 256; ++b) W[(size_t)r * 257 + b] = f == -1 ? 1.0 : Wf[(size_t)f * 257 + b];
    }

    // A1 scale: the largest power of two that keeps every stage-1 output below the fp16 range
    double worst = 0.0;
    for (int n2 = 0; n2 < 16; ++n2) {
        double s = 0.0;
        for (int n1 = 0; n1 < 32; ++n1) {
            const int n = 16 * n1 + n2;
            if (n < Lf) s += fabs((double)window[n]);
        }
        if (s > worst) worst = s;
    }
    if (!(worst > 0.0) || !std::isfinite(worst)) return -1;
    int sa1 = 0;
    while (ldexp(worst, sa1 + 1 + M512_XBITS) <= 60000.0 && sa1 < 8) ++sa1;
    while (ldexp(worst, sa1 + M512_XBITS) > 60000.0 && sa1 > -8) --sa1;
    lay.sa1_log2 = sa1;
    const double SA1 = ldexp(1.0, sa1);

    // the mel blocks of the pattern, in the kernel's order; weight outside the pattern: plan not served
    lay.n_wblocks = 0;
    for (int step = 0; step < 8; ++step)
        for (int tile = 0; tile < lay.n_mtiles; ++tile) {
            const int ip = step >> 1, h = step & 1;
            bool any = false;
            for (int m = 0; m < 16 && !any; ++m)
                for (int g = 0; g < 4 && !any; ++g)
                    for (int j = 0; j < 8 && !any; ++j)
                        if (W[(size_t)(16 * tile + m) * 257 + m512_bin(8 * h + j, M512_KAP[g][ip])] != 0.0) any = true;
            if (m512_pattern_has(lay.n_mtiles, step, tile)) {
                if (lay.n_wblocks >= M512_MAX_WBLOCKS) return -3;
                lay.wblock_step[lay.n_wblocks] = step;
                lay.wblock_tile[lay.n_wblocks] = tile;
                ++lay.n_wblocks;
            } else if (any) {
                return -3;
            }
        }

    int off = 0;
    lay.off_a1 = off; off += 16 * 2 * 2 * 1024;
    lay.off_a2 = off; off += 2 * 2 * 1024;
    lay.off_a2p = off; off += 2 * 2 * 1024;
    lay.off_dm = off; off += 2 * 2 * 1024;
    lay.off_w = off; off += M512_MAX_WBLOCKS * 2 * 1024;
    lay.off_rowsum = off; off += 64;
    lay.bytes = off;
    blob.assign((size_t)off, 0);
    uint8_t* B = blob.data();

    // ---- stage 1: block (n2, t) at off_a1 + ((n2 * 2 + t) * 2 + hl) KB
    for (int n2 = 0; n2 < 16; ++n2)
        for (int t = 0; t < 2; ++t) {
            uint8_t* hi = B + lay.off_a1 + ((n2 * 2 + t) * 2 + 0) * 1024;
            m512_fill_block_f16(hi, hi + 1024, [&](int m, int oct, int j) -> double {
                const int n1 = 8 * oct + j, n = 16 * n1 + n2;
                if (n >= Lf) return 0.0;
                const double w = (double)window[n] * SA1;
                if (m == 0) return t == 0 ? w : ((n1 & 1) ? -w : w);
                const double phi = 2.0 * PI * ((double)((n1 * m) % 32) / 32.0 + (double)((n2 * m) % 512) / 512.0);
                return t == 0 ? w * cos(phi) : -w * sin(phi);
            });
        }
    // ---- stage 2, rows k1 = 1..15: tile u (0 re, 1 im), K element (oct, j): n2 = 4 oct + (j >> 1), part j & 1
    for (int u = 0; u < 2; ++u) {
        uint8_t* hi = B + lay.off_a2 + (u * 2) * 1024;
        m512_fill_block_f16(hi, hi + 1024, [&](int rho, int oct, int j) -> double {
            const int kr = M512_KAP[rho >> 2][rho & 3], n2 = 4 * oct + (j >> 1), part = j & 1;
            const double th = 2.0 * PI * (double)((n2 * kr) % 16) / 16.0;
            if (u == 0) return part == 0 ? cos(th) : sin(th);
            return part == 0 ? -sin(th) : cos(th);
        });
    }
    // ---- stage 2, slot 0: parts are (R0[n2], R16[n2]) = the real rows k1 = 0 and 16 (the latter without its twiddle)
    for (int u = 0; u < 2; ++u) {
        uint8_t* hi = B + lay.off_a2p + (u * 2) * 1024;
        m512_fill_block_f16(hi, hi + 1024, [&](int rho, int oct, int j) -> double {
            const int kr = M512_KAP[rho >> 2][rho & 3], n2 = 4 * oct + (j >> 1), part = j & 1;
            if (kr <= 7) {  // bin 32 kr from R0 (kr = 0, tile 1: bin 256)
                if (part != 0) return 0.0;
                if (u == 0) return cos(2.0 * PI * (double)((n2 * kr) % 16) / 16.0);
                if (kr == 0) return (n2 & 1) ? -1.0 : 1.0;
                return -sin(2.0 * PI * (double)((n2 * kr) % 16) / 16.0);
            }
            if (part != 1) return 0.0;  // bin 16 + 32 m from R16, m = 15 - kr
            const int m = 15 - kr;
            const double th = 2.0 * PI * ((double)n2 / 32.0 + (double)((n2 * m) % 16) / 16.0);
            return u == 0 ? cos(th) : -sin(th);
        });
    }
    // ---- mel blocks
    const double wscale = ldexp(1.0, M512_WSH - 9 - 2 * sa1);  // 2^WSH / (NFFT * SA1^2)
    for (int b = 0; b < lay.n_wblocks; ++b) {
        const int step = lay.wblock_step[b], tile = lay.wblock_tile[b], ip = step >> 1, h = step & 1;
        uint8_t* hi = B + lay.off_w + b * 2048;
        m512_fill_block_bf16(hi, hi + 1024, [&](int m, int oct, int j) -> double {
            return W[(size_t)(16 * tile + m) * 257 + m512_bin(8 * h + j, M512_KAP[oct][ip])] * wscale;
        });
    }
    // ---- DCT * lifter on log2 values: step 0 element (oct, j) is filter 16 (j >> 2) + 4 oct + (j & 3),
    //      step 1 element j < 4 is filter 32 + 4 oct + j; row 0 reads the energy row when append_energy
    const double LN2 = 0.6931471805599453;
    std::vector<double> rowsum(16, 0.0);
    auto dm = [&](int c, int r) -> double {   // coefficient c from ROW r of the log-mel tiles
        const int f = row_filter(r);
        if (c >= C || f == -2) return 0.0;
        if (append_energy && c == 0) return f == -1 ? LN2 : 0.0;
        if (f == -1) return 0.0;
        return LN2 * (double)dct[(size_t)c * M + f];
    };
    for (int step = 0; step < 2; ++step) {
        uint8_t* hi = B + lay.off_dm + (step * 2) * 1024;
        m512_fill_block_f16(hi, hi + 1024, [&](int c, int oct, int j) -> double {
            int f;
            if (step == 0) f = 16 * (j >> 2) + 4 * oct + (j & 3);
            else { if (j >= 4) return 0.0; f = 32 + 4 * oct + j; }
            return dm(c, f);
        });
    }
    for (int c = 0; c < 16; ++c)
        for (int f = 0; f < rows_e; ++f) rowsum[c] += dm(c, f);
    float* rs = reinterpret_cast<float*>(B + lay.off_rowsum);
    for (int c = 0; c < 16; ++c) rs[c] = (float)rowsum[c];
    return 0;
}
