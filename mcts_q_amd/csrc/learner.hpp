// Learner-side records: TD loss, head / encoder backward, weight-gradient GEMM jobs.
// DQNModel.update (dqn_per.py:455-472) between the batch gather and the optimizer step.
#pragma once
#include "kernels.hpp"

namespace mctsq {

constexpr int DQ_LD = 16;       // dQ rows are padded to 16 action columns (zeros beyond A)
constexpr int MAX_WJOBS = 40;   // jobs per wgrad launch (kernel-argument budget: 40 x 80 B)

// ---- TD loss (dqn_per.py:455-467, 474) ----
struct TdArgs {
    const float* q;             // [B][A] policy Q(s)
    const float* qnext;         // [B] max_a' Q_target(s')
    const int32_t* act;
    const float *rew, *done, *isw;
    float gamma, grad_scale;    // grad_scale = loss_scale * 2 / B
    double loss_norm;           // loss = loss_norm * sum w td^2, loss_norm = loss_scale / B
    int64_t B, Bp;
    int A;
    float* dq;                  // [Bp][16]
    float *loss, *abs_td, *qsa, *target;
};
int launch_td_loss(const TdArgs& a, hipStream_t st);

// ---- head backward ----
struct HeadBwdArgs {
    const float* w[5];          // W[l] [out][in]
    int n_fc, in[5], out[5], act;
    int64_t Bp;
    const float* dq;            // [Bp][16]
    const float* sv_act[4];     // post-activation hidden layers [Bp][H]
    float* d_act[4];            // out: d(pre-activation) of hidden layer l, [Bp][H]
    float* dfeat;               // out: [Bp][192]
    const void* wprep_bwd;      // 192-128-128-A head only: transposed split-bf16 fragments (head_prepare_bwd_kernel), or NULL
    // 192-128-128-A head only: the TD loss (td_loss_kernel's arithmetic) evaluated in this kernel's prologue for the
    // tile's 16 rows instead of a launch of its own; td.dq is then an OUTPUT (the last layer's weight-gradient
    // job reads it) and the tile's loss sum goes to loss_parts[tile] (summed in tile order by wgrad_reduce_kernel)
    bool fuse_td;
    TdArgs td;
    double* loss_parts;
};
int launch_head_bwd(const HeadBwdArgs& a, hipStream_t st);
// The head's weights as B operands of d(in) = d(out) . W (k = output unit, n = input unit), split into bf16 planes,
// lane-major: u32x4 [W2: tile 8][plane 3], [W1: tile 8][K chunk 4][plane 3], [W0: tile 12][K chunk 4][plane 3], 64 lanes each
constexpr int64_t HEAD_PREP_BWD_BYTES = (int64_t)(8 * 3 + 8 * 4 * 3 + 12 * 4 * 3) * 64 * 16;
int launch_head_prepare_bwd(const float* w0, const float* w1, const float* w2, int A, void* prep, hipStream_t st);

// ---- GRU-attention backward: LayerNorm, attention pooling, BPTT and the weight gradients (per-workgroup partials) ----
struct GruBwdArgs {
    GruSlot slot[3];
    int nslots;
    int64_t Bp;
    const float* dfeat;         // [Bp][192]
    int half;                   // 1: one workgroup per 8 samples (gru_bwd_kernel<2>), bw_aw partials [Bp/8][65]
    int serial;                 // 1: one workgroup walks all slots of its tile, one after the other (fewer, longer workgroups)
};
// W_hh [192][64] as the B operand of the BPTT product delta [16 x 192] . W_hh: u32x4 [wave 4][K chunk 6][plane 3][lane 64],
// lane (jl, q) of wave w holds W_hh[32c + 8q .. +7][16w + jl]; GRU_PREP_BYTES per slot
int launch_gru_prepare_bwd(const float* const whh[3], int n, void* const out[3], hipStream_t st);
int launch_gru_bwd(const GruBwdArgs& a, hipStream_t st);
int64_t gru_bwd_blocks(const GruBwdArgs& a);
// both groups of windows in one launch (each with its own tile mode); either may be empty
int launch_gru_bwd_pair(const GruBwdArgs& a, const GruBwdArgs& b, hipStream_t st);

// ---- Conv-attention backward (enc_conv.hip) and gas/EUA MLP backward (enc_mlp.hip) ----
struct ConvBwdArgs {
    ConvSlot slot[2];
    int nslots;
    int64_t Bp;
    const float* dfeat;         // [Bp][192]
};
int launch_conv_bwd(const ConvBwdArgs& a, hipStream_t st);
int launch_mlp_bwd(const MlpArgs& a, hipStream_t st);

// ---- Transformer encoder backward (enc_tf_bwd.hip) ----
struct TfBwdArgs {
    TfArgs f;                   // forward arguments incl. the training record
    const float* dfeat;         // [Bp][192]
    int nsb, ntb;               // whole samples per attention-backward workgroup; FFN token blocks
    float *w1t[2], *w2t[2];     // W1^T [64][2048], W2^T [2048][64] per layer
    float *dy2[2], *dx2[2], *p2[2], *dy1[2], *dx1[2], *p1[2];   // [Tp][64] per layer
    float* dqkv[2];             // [Tp][192]
    float* dhpart;              // [16][Tp][64] FFN input-gradient slabs, one per hidden group
    float *wpart1[2], *wpart2[2];   // FFN weight-gradient partials per token block
    float* pfin;                // [Bp][64] dfeat * xhat of the final LayerNorm
    float* dh0;                 // [Tp][64] gradient of the embedding output
};
int launch_tf_bwd(const TfBwdArgs& a, hipStream_t st);
int tf_bwd_token_blocks(int64_t Tp);

// ---- weight gradients: dW[m][n] = sum_k A[k][m] * Bm[k][n], bias[m] = sum_k A[k][m] ----
// Split over K into `nsplit` chunks whose partial results are reduced in a fixed order, so the
// gradient is bitwise reproducible (no float atomics).
struct WJob {                   // 80 bytes: a whole network's job list fits one kernel-argument block
    int64_t a_off;              // A = ws + a_off: [K][lda], M columns (column offset folded in); < 0: the
                                // partials of this job are written by another kernel (reduce only)
    int64_t b_off;              // Bm = ws + b_off: [K][ldb], N columns; < 0: bias (column sums) only
    int64_t part_off;           // float offset of this job's partials: [nsplit][M*N + M]
    int32_t out_w, out_b;       // gradient offsets (< 0: none); weight index = out_w + m*out_ldm + n*out_sn
    int32_t out_ldm, out_sn;
    int32_t lda, ldb, K, kchunk, M, N, nsplit, mblocks, nblocks, blk0;
};
struct WJobList {
    WJob job[MAX_WJOBS];
    int njobs, nblocks_total;
    // first reduced entry of every job (filled by launch_wgrad_reduce): the reduction finds an entry's job from this
    // compact table (three cache lines of the kernel-argument block) instead of walking the 80-byte job records
    int32_t red_e0[MAX_WJOBS + 1];
};
int launch_wgrad(const WJobList& jl, const float* ws, float* partials, hipStream_t st);
// sumsq_parts (optional): one fp64 sum of squares of the reduced gradient entries per block of this launch
// (block order), the pieces of the global gradient norm; returns the number of blocks through *nblocks.
// loss_parts (optional): per-tile loss sums of the fused head backward, folded in order into *loss_out * loss_norm
struct ReduceExtras {
    double* sumsq_parts;
    const double* loss_parts;
    int n_loss_parts;
    double loss_norm;
    float* loss_out;
};
int wgrad_reduce_blocks(const WJobList& jl);
int launch_wgrad_reduce(const WJobList& jl, const float* partials, float* grad, const ReduceExtras* ex, hipStream_t st);

// mctsq_clip_adam_polyak_ex with the data-parallel exchange block (optim.hip): d_exchange [2 * world] or NULL
int clip_adam_polyak_impl(float* d_params, const float* d_grad, float* d_m, float* d_v, float* d_target, int64_t n,
                          int64_t step, double lr, double beta1, double beta2, double eps, double max_norm, double tau,
                          float* d_gnorm, const double* d_sumsq_parts, int64_t nparts, void* d_ws, int64_t ws_bytes,
                          const float* d_exchange, int world, float* d_grad_scaled, float* d_loss_io, void* stream);

}  // namespace mctsq
