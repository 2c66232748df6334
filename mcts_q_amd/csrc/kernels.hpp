// Kernel argument records + launcher prototypes (internal to libmctsq_hip.so).
#pragma once
#include "common.hpp"

namespace mctsq {

// ---------------- GRU-attention encoder ----------------
struct GruSlot {
    const float* x;                      // [B][S][Fd]
    const float *wih, *whh, *bih, *bhh, *aw, *ab, *lnw, *lnb;
    const void* wprep;                   // W_hh as split-fp16 fragments + row exponents (gru_prepare_kernel)
    const void* wprep_t;                 // training: W_hh as B operand of delta . W_hh, split, lane-major (gru_prepare_bwd_kernel)
    int S, Fd, F, tf, out_col;
    // training-time activation record (NULL for inference), Bp = padded batch.  The per-step records are LANE-MAJOR:
    // [step][tile = row / 16][wave 4][lane 64][r 4] holds the value of row 16 tile + 4 (lane >> 4) + r, unit
    // 16 wave + (lane & 15) -- the forward's and the backward's own register layout, so both move a step's record
    // with one 16-byte access per lane (gru_rec_index() addresses single elements)
    float* sv_h;                         // [S+1] step blocks: block 0 = h_{-1} = 0, block t+1 = h_t
    float *sv_r, *sv_z, *sv_n, *sv_hn;   // [S] step blocks
    float* sv_score;                     // [S][Bp], log2 units
    float* sv_stat;                      // [Bp][4] = running max, running sum, LN mean, LN rstd
    float* sv_pool;                      // [Bp][64] pooled vector before LayerNorm
    float* sv_x;                         // [S][Bp][8] time-major window incl. time features, zero padded
    // backward outputs (learner.hip)
    // per-workgroup partials of the weight gradients, accumulated inside the backward kernel, in the layout of three
    // reduction jobs: W_ih + b_ih [192*F + 192], W_hh / b_hh rows r, z [128*64 + 128], rows n [64*64 + 64]
    float *bw_wih, *bw_wrz, *bw_wn;
    float* bw_dsc;                       // [S][Bp] d(score)
    float* bw_p;                         // [Bp][64] dfeat * xhat (LayerNorm weight gradient terms)
    float* bw_aw;                        // [workgroups][65] per-workgroup partials of the score Linear(64,1) gradient
    float* bw_ln;                        // [workgroups][64] per-workgroup partials of the LayerNorm weight gradient
};
// element (row, unit) of step block t of a lane-major record with Bp rows
inline __host__ __device__ int64_t gru_rec_index(int64_t t, int64_t Bp, int64_t row, int unit) {
    const int64_t tile = row >> 4;
    const int i = (int)(row & 15);
    return (((t * (Bp >> 4) + tile) * 4 + (unit >> 4)) * 64 + ((i >> 2) * 16 + (unit & 15))) * 4 + (i & 3);
}
struct GruFwdArgs {
    GruSlot slot[3];
    int nslots;
    int64_t B, Bp;
    float* feats;                        // [Bp][192]
};
int launch_gru_fwd(const GruFwdArgs& args, bool save, hipStream_t st);
// W_hh [192][64] fp32 -> split-fp16 fragments as the recurrence kernels load them (prepare_core.hpp: prep_gru_fwd):
// u32x4 [wave 4][gate 3][K chunk 2][plane 2][lane 64], rows scaled by powers of two, then the 192 row exponents
constexpr int64_t GRU_PREP_BYTES = 4 * 3 * 2 * 3 * 64 * 16;     // per slot; the backward (three bf16 planes) fills it
constexpr int64_t GRU_FWD_FRAG_BYTES = 4 * 3 * 2 * 2 * 64 * 16; // forward: two fp16 planes, then int32 k[192] (prepare_core.hpp)
static_assert(GRU_FWD_FRAG_BYTES + 192 * 4 <= GRU_PREP_BYTES, "forward GRU fragments + row exponents fit the slot");
int launch_gru_prepare(const float* const whh[3], int n, void* prep, hipStream_t st);

// ---------------- Conv-attention encoder (dqn_per.py:21-55) ----------------
struct ConvSlot {
    const float* x;                      // [B][S][Fd]
    const float *w1, *b1, *w2, *b2, *aw, *ab, *lnw, *lnb;
    const void* wprep;                   // conv2 weights as split-fp16 fragments + channel exponents (conv_prepare_kernel)
    int S, Fd, F, tf, out_col;
    // training record (NULL for inference)
    float* sv_h1;                        // [S+2][Bp][64] post-ReLU conv1, zero blocks at both ends
    float* sv_h2;                        // [S][Bp][64] post-ReLU conv2
    float* sv_score;                     // [S][Bp], log2 units
    float* sv_stat;                      // [Bp][4] = max, sum, LN mean, LN rstd
    float* sv_pool;                      // [Bp][64]
    float* sv_x;                         // [S+2][Bp][8] time-major window, zero blocks at both ends
    // backward outputs
    float *bw_d2, *bw_d1;                // [S][Bp][64] d(pre-activation) of conv2 / conv1
    float* bw_dsc;                       // [S][Bp]
    float* bw_p;                         // [Bp][64]
};
struct ConvFwdArgs {
    ConvSlot slot[2];
    int nslots;
    int64_t B, Bp;
    float* feats;                        // [Bp][192]
};
struct MlpArgs;
// mlp != NULL: the gas/EUA MLP encoder of the same network rides in the same launch (enc_conv.hip)
int launch_conv_fwd(const ConvFwdArgs& args, bool save, hipStream_t st, const MlpArgs* mlp = nullptr);
// conv2 weight [64 out][64 in][3 taps] fp32 -> split-fp16 fragments in the order the forward kernel loads them:
// u32x4 [wave 4][K chunk 6 = 2 tap + half][plane 2][lane 64]; lane (jl, q) holds W2[16 wave + jl][32 half + 8q .. +7][tap]
// times the output channel's power of two; the 64 channel exponents follow
constexpr int64_t CONV_PREP_BYTES = 4 * 6 * 3 * 64 * 16;      // per slot (sized for three planes)
constexpr int64_t CONV_FRAG_BYTES = 4 * 6 * 2 * 64 * 16;      // two fp16 planes, then int32 s2[64] (conv_prepare_kernel)
static_assert(CONV_FRAG_BYTES + 64 * 4 <= CONV_PREP_BYTES, "conv fragments + channel exponents fit the slot");
int launch_conv_prepare(const float* const w2[2], int n, void* prep, hipStream_t st);

// ---------------- gas/EUA MLP encoder (dqn_per.py:108-114, 128-130) ----------------
struct MlpArgs {
    const float* x;                      // [B][3*Fd]
    const float *w0, *b0, *w1, *b1;
    int Fd, F, tf, out_col;              // F = Fd (+2 time features per step)
    int64_t B, Bp;
    float* feats;                        // [Bp][192]
    float* sv_x;                         // [Bp][16] flattened window incl. time features, zero padded
    float* sv_m0;                        // [Bp][64] post-ReLU hidden layer
    float *bw_d1, *bw_d0;                // [Bp][64] d(pre-activation) of layer 1 / layer 0
    const float* dfeat;                  // backward input [Bp][192]
};
int launch_mlp_fwd(const MlpArgs& a, bool save, hipStream_t st);

// ---------------- Transformer encoder (dqn_per.py:80-101) ----------------
struct TfLayerW {
    const float *inw, *inb, *ow, *ob, *l1w, *l1b, *l2w, *l2b, *n1w, *n1b, *n2w, *n2b;
};
struct TfLayerBuf {
    float *hin, *hmid, *hout;
    // training record (NULL for inference)
    float* qkv;                          // [Tp][192] in_proj output (q already scaled by 1/4)
    float* ctx;                          // [Tp][64] attention output before out_proj
    float *x1, *st1;                     // [Tp][64] pre-LayerNorm1 sum, [Tp][2] (mean, rstd)
    float *x2, *st2;                     // same for LayerNorm2
};
struct TfArgs {
    const float* x;                      // [B][S][Fd]
    const float *pos, *ew, *eb;          // positional table [500][64], embedding Linear(F,64)
    TfLayerW layer[2];
    const float *lnw, *lnb;
    int S, Fd, F, tf, out_col, NS;       // NS = whole samples per attention workgroup (NS*S <= 64)
    int64_t B, Bp, T, Tp;                // samples / tokens the block kernels process (training: B = Bp)
    int64_t Bvalid;                      // samples present in x (rows beyond are treated as zero windows)
    // token activations [Tp][64]: layer l reads hin, attention writes hmid, FFN writes hout.
    // Inference ping-pongs two buffers; training gives every stage its own (kept for backward).
    TfLayerBuf buf[2];
    float* feats;                        // [Bp][192]
    float* sv_mean;                      // training: [Bp][64] mean over time before the final LayerNorm
    float* sv_xe;                        // training: [Tp][8] embedded-input rows incl. time features
    const void* ffn_prep[2];             // per layer: FFN (split-fp16 fragments + scale tables), then in_proj / out_proj as split-fp16 fragments (tf_prepare_kernel)
};
int64_t tf_padded_tokens(int64_t B, int S);
int launch_tf_fwd(const TfArgs& a, hipStream_t st);
int launch_tf_fwd_pair(const TfArgs& a, const TfArgs& b, hipStream_t st);   // two encoders of one network, FFN grids merged
// FFN weights of one layer (linear1 [2048][64], linear2 [64][2048]) -> split-fp16 fragments in the
// order tf_ffn_kernel consumes them: u32x4 [pair of hidden tiles 64][fragment 16][lane 64] (1 MB), then the scale
// tables of the split (enc_tf.hip: TfFfnTables) inside the same area
constexpr int64_t TF_FFN_PREP_BYTES = 64 * 24 * 64 * 16;
// ... followed by the attention block's in_proj [192][64] and out_proj [64][64] as split-fp16 B fragments (rows scaled by
// powers of two, inverses in the scale tables): u32x4 [in_proj column tile 12][K chunk 2][plane 2][lane 64], then
// [out_proj tile 4][K chunk 2][plane 2][lane 64]
constexpr int64_t TF_ATTN_PREP_BYTES = (12 + 4) * 2 * 2 * 64 * 16;
constexpr int64_t TF_LAYER_PREP_BYTES = TF_FFN_PREP_BYTES + TF_ATTN_PREP_BYTES;
int launch_tf_prepare(const TfLayerW& w, void* prep, hipStream_t st);   // reads inw, ow, l1w, l1b, l2w, n1w, n1b

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
    const void* wprep;                   // fused kernel only: split-bf16 weight fragments (head_prepare_kernel)
};
int launch_head_fwd(const HeadArgs& args, hipStream_t st);

// gru/gru/gru + 192-128-128-A head in one launch (fused_gru_head.hip)
bool fused_gru_head_applicable(const GruFwdArgs& ga, const HeadArgs& ha);
// the three head layers as split-bf16 B fragments: 8x6x3 + 8x4x3 + 4x3 fragments of 64 lanes x 16 B
constexpr int64_t HEAD_PREP_BYTES = (int64_t)(8 * 6 * 3 + 8 * 4 * 3 + 4 * 3) * 64 * 16;
int launch_head_prepare(const float* w0, const float* w1, const float* w2, int A, void* prep, hipStream_t st);
int launch_gru3_head_fwd(const GruFwdArgs& ga, const HeadArgs& ha, bool save, hipStream_t st);
// (ga0, ha0): activation-saving forward, (ga1, ha1): plain forward, as one launch (the learner's policy / target pair)
int launch_gru3_head_fwd_pair(const GruFwdArgs& ga0, const HeadArgs& ha0, const GruFwdArgs& ga1, const HeadArgs& ha1,
                              hipStream_t st);

}  // namespace mctsq
