// Specialised NFFT=512 MFCC kernel (placeholder until the register-resident FFT lands).
#pragma once

#include "dsp_common.h"

static inline int fast512_plan_init(dsp_plan* p, const dsp_plan_desc*, const int32_t*) {
    p->d_fast = nullptr;
    return DSP_OK;
}
static inline void fast512_plan_free(dsp_plan*) {}
static inline bool fast512_applicable(const dsp_plan*) { return false; }
static inline int fast512_launch(const dsp_plan*, const void*, int, const BatchGeom&, float*, int64_t, hipStream_t) {
    return DSP_EINVAL;
}
