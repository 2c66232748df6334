// Kernel argument records + launcher prototypes (internal to libmctsq_hip.so).
#pragma once
#include "common.hpp"

namespace mctsq {

// ---------------- GRU-attention encoder ----------------
struct GruSlot {
    const float* x;                      // [B][S][Fd]
    const float *wih, *whh, *bih, *bhh, *aw, *ab, *lnw, *lnb;
    int S, Fd, F, tf, out_col;
    // training-time activation record (NULL for inference), Bp = padded batch
    float* sv_h;                         // [S+1][Bp][64]: block 0 = h_{-1} = 0, block t+1 = h_t
    float *sv_r, *sv_z, *sv_n, *sv_hn;   // [S][Bp][64]
    float* sv_score;                     // [S][Bp], log2 units
    float* sv_stat;                      // [Bp][4] = running max, running sum, LN mean, LN rstd
    float* sv_pool;                      // [Bp][64] pooled vector before LayerNorm
    float* sv_x;                         // [S][Bp][8] time-major window incl. time features, zero padded
    // backward outputs (learner.hip)
    float* bw_g;                         // [S][Bp][256] = d(pre-activation) r | z | n | n*r
    float* bw_dsc;                       // [S][Bp] d(score)
    float* bw_p;                         // [Bp][64] dfeat * xhat (LayerNorm weight gradient terms)
};
struct GruFwdArgs {
    GruSlot slot[3];
    int nslots;
    int64_t B, Bp;
    float* feats;                        // [Bp][192]
};
int launch_gru_fwd(const GruFwdArgs& args, bool save, hipStream_t st);

// ---------------- head ----------------
struct HeadArgs {
    const float* feats;                  // [Bp][192]
    const float* w[5];
    const float* b[5];
    int n_fc, in[5], out[5], act, A;
    int64_t B, Bp;
    float *q, *vmax, *prior;             // [B][A], [B], [B][A]; vmax/prior may be NULL
    int32_t* argmax;                     // [B] or NULL
    float* sv_act[4];                    // training: post-activation hidden layers [Bp][H] or NULL
};
int launch_head_fwd(const HeadArgs& args, hipStream_t st);

// gru/gru/gru + 192-128-128-A head in one launch (fused_gru_head.hip)
bool fused_gru_head_applicable(const GruFwdArgs& ga, const HeadArgs& ha);
int launch_gru3_head_fwd(const GruFwdArgs& ga, const HeadArgs& ha, bool save, hipStream_t st);

}  // namespace mctsq
