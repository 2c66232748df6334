// Internal: forward pass shared by the inference entry point (capi.cpp) and the learner
// (learner_api.cpp).
#pragma once
#include "kernels.hpp"

namespace mctsq {

// Training-time record of one encoder slot (workspace pointers; fields used depend on the kind).
struct SlotSave {
    // GRU-attention (kernels.hpp: GruSlot)
    float *h, *r, *z, *n, *hn, *score, *stat, *pool, *x;
    float *g, *dsc, *p;
};

struct HeadOut {
    float* q;
    float* vmax;
    int32_t* argmax;
    float* prior;
};

void fill_gru_args(GruFwdArgs& ga, const NetLayout& L, const float* params, const float* const x[3], int64_t B,
                   int64_t Bp, float* feats, const SlotSave* sv);
int forward_impl(const NetLayout& L, const float* params, const float* const x[3], int64_t B, int64_t Bp,
                 float* feats, const HeadOut& out, const SlotSave* sv, float* const* sv_act, hipStream_t st,
                 void* ev_encoders_done);

}  // namespace mctsq
