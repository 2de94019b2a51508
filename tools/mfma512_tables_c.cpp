// C wrapper around csrc/mfma512_tables.h for tools/mfma512_emul.py (CPU only; g++ -shared).
#include "../dsp-speech-recognition_amd/csrc/mfma512_tables.h"

extern "C" int m512_tables(int L, int S, int nfft, int M, int C, int append_energy, const float* window,
                           const int32_t* mel_start, const int32_t* mel_count, const float* mel_w, const float* dct,
                           uint8_t* out, int out_cap, int32_t* lay_out /* 64 ints */) {
    std::vector<uint8_t> blob;
    M512Layout lay;
    const int rc = m512_build_tables(L, S, nfft, M, C, append_energy, window, mel_start, mel_count, mel_w, dct, blob, lay);
    if (rc != 0) return rc;
    if ((int)blob.size() > out_cap) return -100;
    memcpy(out, blob.data(), blob.size());
    int k = 0;
    lay_out[k++] = lay.off_a1; lay_out[k++] = lay.off_a2; lay_out[k++] = lay.off_a2p; lay_out[k++] = lay.off_dm;
    lay_out[k++] = lay.off_w; lay_out[k++] = lay.off_rowsum; lay_out[k++] = lay.bytes; lay_out[k++] = lay.n_wblocks;
    lay_out[k++] = lay.n_mtiles; lay_out[k++] = lay.sa1_log2; lay_out[k++] = lay.KR;
    for (int b = 0; b < M512_MAX_WBLOCKS; ++b) { lay_out[k++] = lay.wblock_step[b]; lay_out[k++] = lay.wblock_tile[b]; }
    lay_out[k++] = lay.erow;
    return 0;
}

extern "C" uint16_t m512_half(float f) { return m512_f2h(f); }
