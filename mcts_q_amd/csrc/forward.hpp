// Internal: forward pass shared by the inference entry point (capi.cpp) and the learner
// (learner_api.cpp).
#pragma once
#include "kernels.hpp"

namespace mctsq {

// Training-time record of one encoder slot (workspace pointers; fields used depend on the kind).
struct SlotSave {
    // shared by the attention-pooled encoders (GRU, conv)
    float *score, *stat, *pool, *x, *dsc, *p;
    // GRU-attention (kernels.hpp: GruSlot)
    float *h, *r, *z, *n, *hn, *wih, *wrz, *wn, *aw, *ln;
    // Conv-attention (kernels.hpp: ConvSlot)
    float *c_h1, *c_h2, *c_d2, *c_d1;
    // gas/EUA MLP (kernels.hpp: MlpArgs)
    float *m_x, *m_m0, *m_d1, *m_d0;
    // Transformer (kernels.hpp: TfArgs): per-stage token buffers and records
    TfLayerBuf t_buf[2];
    float *t_mean, *t_xe;
};

struct HeadOut {
    float* q;
    float* vmax;
    int32_t* argmax;
    float* prior;
};

void fill_gru_args(GruFwdArgs& ga, const NetLayout& L, const float* params, const void* prep,
                   const float* const x[3], int64_t B, int64_t Bp, float* feats, const SlotSave* sv);
int64_t gru_prepared_bytes(const NetLayout& L);
void fill_conv_args(ConvFwdArgs& ca, const NetLayout& L, const float* params, const void* prep,
                    const float* const x[3], int64_t B, int64_t Bp, float* feats, const SlotSave* sv);
bool fill_mlp_args(MlpArgs& ma, const NetLayout& L, const float* params, const float* const x[3], int64_t B,
                   int64_t Bp, float* feats, const SlotSave* sv);
// floats of forward scratch: feature rows [Bp][192] + token buffers of the transformer slots + the
// prepared (split) recurrent weights of the GRU slots
int64_t forward_ws_floats(const NetLayout& L, int64_t Bp);
int64_t prepared_bytes(const NetLayout& L);
int prepare_impl(const NetLayout& L, const float* params, void* prep, hipStream_t st);
void fill_tf_args(TfArgs& t, const EncOff& e, const float* params, const float* x, int slot, int64_t B,
                  int64_t Bp, float* feats, float* hbuf, const SlotSave* sv);
// prep: prepared weights of `params` (prepare_impl), or NULL to prepare them into the workspace first
// policy forward (records kept) + target forward as one launch; 1 = not applicable, nothing enqueued
int forward_pair_impl(const NetLayout& L, const float* params, const void* prep, const float* const x[3], float* ws_p,
                      const HeadOut& out_p, const SlotSave* sv, float* const* sv_act, const float* tparams,
                      const void* tprep, const float* const xt[3], float* ws_t, const HeadOut& out_t, int64_t B,
                      int64_t Bp, hipStream_t st);
int forward_impl(const NetLayout& L, const float* params, const void* prep, const float* const x[3], int64_t B,
                 int64_t Bp, float* ws, const HeadOut& out, const SlotSave* sv, float* const* sv_act, hipStream_t st,
                 void* ev_encoders_done);

}  // namespace mctsq
