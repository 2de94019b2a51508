// Host-side tables of the frame-per-product matrix-pipe NFFT = 512 kernel (kernels_mfma512t.h).  Plain C++ (no HIP):
// also compiled by tools/mfma512t_tables_c.cpp for the CPU emulation of the kernel's operand maps
// (tools/mfma512t_emul.py, tests/test_mfma512t_tables.py).
//
// The 512-point real DFT of ONE frame (sigproc.py:136-158 after :66-98) as two dense products on
// v_mfma_f32_16x16x32_f16 whose N side is the DFT row k1, not the frame (kernels_mfma512.h puts 16 frames there):
//   n = 16 n1 + n2, k = k1 + 32 k2
//   stage 1:  Y[n2, k1] = sum_n1 (w x)[16 n1 + n2] W32^(n1 k1)      A = the frame's windowed samples [M = n2][K = n1],
//                                                                   B = ONE register-resident DFT-32 matrix [K = n1][N = k1]
//   twiddle:  Y'[n2, k1] = W512^(n2 k1) Y[n2, k1]                   (vector pipe, per-lane constants)
//   stage 2:  X[k1 + 32 k2] = sum_n2 W16^(n2 k2) Y'[n2, k1]         A = ONE DFT-16 matrix [M = k2 (re | im)][K = n2 (re, im)],
//                                                                   B = Y' as it comes out of stage 1 (rows -> K: no lane movement)
// Column k1 = 0 of stage 1 carries the two REAL rows k1 = 0 and k1 = 16; their bins 16 m are one more product per 16
// frames (X[16 m] = sum_n2 W32^(n2 m) (m even ? Y0 : Y16)[n2]).  |X|^2 -> bf16 (hi, lo) -> mel (+ the all-ones energy
// row, base.py:18-32) with 16 frames on the N side after one exchange through LDS, log2, DCT * lifter (base.py:8-16,
// 60-68).  Every fp16 operand is a (hi, lo) pair and every product three MFMAs (hi hi, hi lo, lo hi), as in
// mfma512_tables.h, whose conversion helpers this header reuses.
//
// Operand maps (MI355X guide): lane l = (row / column l & 15, octet g = l >> 4) holds K elements 8 g + j, j = 0..7;
// D[row 4 g + r][column l & 15] in register r.
#pragma once

#include "mfma512_tables.h"

#define M512T_MAX_WBLOCKS 27   // mel blocks (2 KB each: hi KB, lo KB): 8 steps + column 0, up to three row tiles

// stage-1 K element (g, j) -> n1.  Rows (2 g, 2 g + 1) + 8 q: the four 8-byte reads of a lane's operand (one per q)
// are conflict free over the column-major sample image (kernels_mfma512t.h).
static inline int m512t_n1(int g, int j) { return 8 * (j >> 1) + 2 * g + (j & 1); }
// stage-2 / col-0 K element (g, j) -> (n2, part): part 0 = re (or Y0), part 1 = im (or Y16)
static inline int m512t_n2(int g, int j) { return 4 * g + (j & 3); }
static inline int m512t_part(int j) { return j >> 2; }
// stage-2 output row rho = 4 g + r -> k2: lane group g covers FFT bins [64 g, 64 g + 64) (keeps the mel blocks sparse)
static inline int m512t_k2(int rho) {
    const int g = rho >> 2, r = rho & 3;
    return r == 0 ? 2 * g : (r == 1 ? 2 * g + 1 : (r == 2 ? 14 - 2 * g : 15 - 2 * g));
}
// FFT bin of power value r of lane (g, k1), k1 = 1..15
static inline int m512t_bin(int g, int k1, int r) {
    const int k = k1 + 32 * m512t_k2(4 * g + r);
    return k <= 256 ? k : 512 - k;
}
// FFT bin of column-0 power value r of lane group g: 16 (4 g + r); (0, 0) stands for bins 0 AND 256 together
static inline int m512t_bin0(int g, int r) { return 16 * (4 * g + r); }

// The mel blocks the kernel multiplies for bin range rg = [64 rg, 64 rg + 64) and row tile `tile` (compile-time there).
// Pattern 0: every block.  Pattern 1 (three row tiles, e.g. 40 filters + energy over 0..8 kHz): tile 0 only in the
// lowest range, tile 1 (which carries the energy row) everywhere, tile 2 only in the upper half.
static inline constexpr bool m512t_pattern_has(int pattern, int n_mtiles, int rg, int tile) {
    return tile < n_mtiles && (pattern == 0 || tile == 1 || (tile == 0 && rg == 0) || (tile == 2 && rg >= 2));
}

struct M512TLayout {
    int32_t off_f1, off_f2, off_m0, off_dm, off_tw, off_win, off_rowsum, off_w;  // byte offsets into the blob (16-byte aligned)
    int32_t bytes;
    int32_t n_wblocks;                 // mel blocks stored (2 KB each: hi KB, lo KB), in multiplication order
    int8_t wblock_step[M512T_MAX_WBLOCKS], wblock_tile[M512T_MAX_WBLOCKS];   // step 0..7: packets 8 s .. 8 s + 7; 8: column 0
    int32_t pattern;                   // which (bin range, row tile) blocks the kernel multiplies: m512t_pattern_has
    int32_t erow;                      // row of the all-ones energy filter (-1: none)
    int32_t n_mtiles;                  // row tiles of 16 filters (+ energy row): 2 or 3
    float z_log2_eps;                  // log2(eps) (base.py:26,30)
};

// Returns 0 and fills blob / lay, or a negative reason when the plan does not fit the kernel:
//  -1 shape, -2 filterbank weight on bin 0 / 256, -3 too many mel blocks.
static inline int m512t_build_tables(int L, int S, int nfft, int M, int C, int append_energy, const float* window,
                                     const int32_t* mel_start, const int32_t* mel_count, const float* mel_w,
                                     const float* dct, std::vector<uint8_t>& blob, M512TLayout& lay) {
    memset(&lay, 0, sizeof(lay));
    if (nfft != 512 || L < 1 || S < 16 || (S % 16) != 0 || M < 1 || C < 1 || C > 16 || C > M) return -1;
    const int Lf = L < 512 ? L : 512;
    const int rows_e = M + (append_energy ? 1 : 0);
    if (rows_e > 48) return -1;
    lay.n_mtiles = rows_e <= 32 ? 2 : 3;   // the kernel is instantiated for two and three row tiles (unused rows: zero weights)
    lay.z_log2_eps = (float)log2(2.220446049250313e-16);
    const double PI = 3.14159265358979323846;
    // fp16 range: the tile scale keeps |y| < 2^XBITS, so a stage-1 sum is below 2^XBITS * (largest column sum of |w|)
    // and a twiddled value below sqrt(2) times that: the sums of any window with |w| <= 1 over 25 rows fit (36 k < 65 504)
    for (int n2 = 0; n2 < 16; ++n2) {
        double s = 0.0;
        for (int n = n2; n < Lf; n += 16) {
            if (!std::isfinite(window[n])) return -1;
            s += fabs((double)window[n]);
        }
        if (ldexp(s * 1.4142135623730951, M512_XBITS) > 60000.0) return -1;
    }

    // dense mel weights over bins, by ROW: filters in order with the energy row (all ones) at row erow
    lay.erow = append_energy ? (M >= 31 ? 31 : M) : -1;
    auto row_filter = [&](int r) -> int {   // -1: energy row, -2: no such row
        if (r >= rows_e) return -2;
        if (lay.erow < 0) return r;
        return r == lay.erow ? -1 : (r < lay.erow ? r : r - 1);
    };
    std::vector<double> W((size_t)48 * 257, 0.0);
    {
        size_t o = 0;
        std::vector<double> Wf((size_t)M * 257, 0.0);
        for (int f = 0; f < M; ++f) {
            for (int c = 0; c < mel_count[f]; ++c) {
                const int b = mel_start[f] + c;
                if (b < 0 || b > 256) return -1;
                Wf[(size_t)f * 257 + b] = mel_w[o + c];
            }
            o += mel_count[f];
            if (Wf[(size_t)f * 257 + 0] != 0.0 || Wf[(size_t)f * 257 + 256] != 0.0) return -2;
        }
        for (int r = 0; r < rows_e; ++r) {
            const int f = row_filter(r);
            for (int b = 0; b <= 256; ++b) W[(size_t)r * 257 + b] = f == -1 ? 1.0 : Wf[(size_t)f * 257 + b];
        }
    }
    // mel K elements.  Steps 0..7 (the exchange: once with the hi halves of the powers, once with the lo halves):
    // element (gq, j) of step s is power value r = j & 3 of packet p = 8 s + 2 gq + (j >> 2), packet p = lane p of the
    // stage-2 result = (g = p >> 4, k1 = p & 15); step 8: column 0, element j and j + 4 are the hi and lo half of value j.
    auto wval = [&](int tile, int m, int step, int gq, int j) -> double {
        const int r = j & 3;
        if (step < 8) {
            const int p_ = 8 * step + 2 * gq + (j >> 2), g = p_ >> 4, k1 = p_ & 15;
            if (k1 == 0) return 0.0;                // column 0 of the main path holds the packed rows 0 / 16: not a bin
            return W[(size_t)(16 * tile + m) * 257 + m512t_bin(g, k1, r)];
        }
        return W[(size_t)(16 * tile + m) * 257 + m512t_bin0(gq, r)];
    };
    // which (bin range, row tile) pairs carry weight; the kernel is instantiated for two patterns (m512t_pattern_has)
    bool need[4][3] = {{false}};
    for (int step = 0; step < 8; ++step)
        for (int tile = 0; tile < lay.n_mtiles; ++tile)
            for (int m = 0; m < 16; ++m)
                for (int gq = 0; gq < 4; ++gq)
                    for (int j = 0; j < 8; ++j)
                        if (wval(tile, m, step, gq, j) != 0.0) need[step >> 1][tile] = true;
    lay.pattern = lay.n_mtiles == 3 ? 1 : 0;
    for (int rg = 0; rg < 4; ++rg)
        for (int tile = 0; tile < lay.n_mtiles; ++tile)
            if (need[rg][tile] && !m512t_pattern_has(lay.pattern, lay.n_mtiles, rg, tile)) lay.pattern = 0;
    lay.n_wblocks = 0;
    for (int step = 0; step <= 8; ++step)
        for (int tile = 0; tile < lay.n_mtiles; ++tile) {
            if (step < 8 && !m512t_pattern_has(lay.pattern, lay.n_mtiles, step >> 1, tile)) continue;
            if (lay.n_wblocks >= M512T_MAX_WBLOCKS) return -3;
            lay.wblock_step[lay.n_wblocks] = (int8_t)step;
            lay.wblock_tile[lay.n_wblocks] = (int8_t)tile;
            ++lay.n_wblocks;
        }

    int off = 0;
    lay.off_f1 = off; off += 2 * 2 * 1024;
    lay.off_f2 = off; off += 2 * 2 * 1024;
    lay.off_m0 = off; off += 2 * 2 * 1024;
    lay.off_dm = off; off += 2 * 2 * 1024;
    lay.off_tw = off; off += 64 * 8 * 4;
    lay.off_win = off; off += 64 * 8 * 4;
    lay.off_rowsum = off; off += 256;
    lay.off_w = off; off += lay.n_wblocks * 2048;
    lay.bytes = off;
    blob.assign((size_t)off, 0);
    uint8_t* B = blob.data();

    // ---- stage 1 (B operand): column block u (0: re of k1 = 0..15; 1: column 0 = the real row k1 = 16, columns 1..15 im)
    for (int u = 0; u < 2; ++u) {
        uint8_t* hi = B + lay.off_f1 + (u * 2) * 1024;
        m512_fill_block_f16(hi, hi + 1024, [&](int k1, int g, int j) -> double {
            const int n1 = m512t_n1(g, j);
            if (u == 0) return cos(2.0 * PI * (double)((n1 * k1) % 32) / 32.0);
            if (k1 == 0) return (n1 & 1) ? -1.0 : 1.0;
            return -sin(2.0 * PI * (double)((n1 * k1) % 32) / 32.0);
        });
    }
    // ---- stage 2 (A operand): row tile u (0 re, 1 im)
    for (int u = 0; u < 2; ++u) {
        uint8_t* hi = B + lay.off_f2 + (u * 2) * 1024;
        m512_fill_block_f16(hi, hi + 1024, [&](int rho, int g, int j) -> double {
            const int k2 = m512t_k2(rho), n2 = m512t_n2(g, j), part = m512t_part(j);
            const double th = 2.0 * PI * (double)((n2 * k2) % 16) / 16.0;
            if (u == 0) return part == 0 ? cos(th) : sin(th);
            return part == 0 ? -sin(th) : cos(th);
        });
    }
    // ---- column 0 (A operand): row m of tile u holds bin 16 m (re | im); row 0: bin 0 (re tile) and bin 256 (im tile)
    for (int u = 0; u < 2; ++u) {
        uint8_t* hi = B + lay.off_m0 + (u * 2) * 1024;
        m512_fill_block_f16(hi, hi + 1024, [&](int m, int g, int j) -> double {
            const int n2 = m512t_n2(g, j), part = m512t_part(j);
            if (m == 0) {
                if (part != 0) return 0.0;
                return u == 0 ? 1.0 : ((n2 & 1) ? -1.0 : 1.0);
            }
            if (part != (m & 1)) return 0.0;
            const double th = 2.0 * PI * (double)((n2 * m) % 32) / 32.0;
            return u == 0 ? cos(th) : -sin(th);
        });
    }
    // ---- twiddles W512^(n2 k1) = c + i s, per lane (g, k1): [c(r = 0..3), s(r = 0..3)], n2 = 4 g + r
    {
        float* tw = reinterpret_cast<float*>(B + lay.off_tw);
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int n2 = 4 * (l >> 4) + r, k1 = l & 15;
                const double ph = 2.0 * PI * (double)((n2 * k1) % 512) / 512.0;
                tw[l * 8 + r] = (float)cos(ph);
                tw[l * 8 + 4 + r] = (float)(-sin(ph));
            }
    }
    // ---- window in operand order: lane (g, c) element j is sample 16 n1(g, j) + c of the frame (zero past L)
    {
        float* wo = reinterpret_cast<float*>(B + lay.off_win);
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int n = 16 * m512t_n1(l >> 4, j) + (l & 15);
                wo[l * 8 + j] = n < Lf ? window[n] : 0.f;
            }
    }
    // ---- mel blocks (A operand, bf16): hi KB then lo KB
    const double wscale = ldexp(1.0, M512_WSH - 9);  // 2^WSH / NFFT
    for (int b = 0; b < lay.n_wblocks; ++b) {
        uint8_t* hi = B + lay.off_w + b * 2048;
        const int step = lay.wblock_step[b], tile = lay.wblock_tile[b];
        m512_fill_block_bf16(hi, hi + 1024, [&](int m, int gq, int j) -> double { return wval(tile, m, step, gq, j) * wscale; });
    }
    // ---- DCT * lifter on log2 values: step 0 element (g, j) is row 16 (j >> 2) + 4 g + (j & 3) of the log-mel tiles,
    //      step 1 element j < 4 is row 32 + 4 g + j; coefficient 0 reads the energy row when append_energy
    const double LN2 = 0.6931471805599453;
    auto dm = [&](int c, int r) -> double {
        const int f = row_filter(r);
        if (c >= C || f == -2) return 0.0;
        if (append_energy && c == 0) return f == -1 ? LN2 : 0.0;
        if (f == -1) return 0.0;
        return LN2 * (double)dct[(size_t)c * M + f];
    };
    for (int step = 0; step < 2; ++step) {
        uint8_t* hi = B + lay.off_dm + (step * 2) * 1024;
        m512_fill_block_f16(hi, hi + 1024, [&](int c, int g, int j) -> double {
            int f;
            if (step == 0) f = 16 * (j >> 2) + 4 * g + (j & 3);
            else { if (j >= 4) return 0.0; f = 32 + 4 * g + j; }
            return dm(c, f);
        });
    }
    float* rs = reinterpret_cast<float*>(B + lay.off_rowsum);
    for (int c = 0; c < 16; ++c) {
        double s = 0.0;
        for (int f = 0; f < rows_e; ++f) s += dm(c, f);
        rs[c] = (float)s;
    }
    return 0;
}
