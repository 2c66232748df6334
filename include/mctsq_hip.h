/* mctsq_hip.h - C ABI of libmctsq_hip.so: the MI355X (gfx950) hot path of MCTS_Q.
 *
 * The reference has no FFI; the boundary it replaces is the Python class API of
 * /root/reference/src/mctsq_config_dqn_per.py (and mctsq_config_dqn.py).  Each entry point below
 * names the reference lines whose arithmetic it performs.  Conventions:
 *   - every function returns 0 on success, <0 on error; mctsq_last_error() gives the message
 *     (thread-local).  No exceptions cross the boundary.
 *   - the CALLER owns every buffer; pointers named d_* are DEVICE pointers (HBM), others host.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *   - one process per GPU; calls on different streams of one device may overlap only if they
 *     touch disjoint buffers.
 *   - no entry point allocates or frees device memory; scratch comes from the `d_ws` workspace
 *     argument sized by the matching *_workspace_bytes() query.
 */
#ifndef MCTSQ_HIP_H
#define MCTSQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCTSQ_ABI_VERSION 5

/* encoder kinds (dqn_per.py:181-203) */
enum { MCTSQ_ENC_CONV = 1, MCTSQ_ENC_GRU = 2, MCTSQ_ENC_TRANSFORMER = 3, MCTSQ_ENC_MLP = 4 };
/* head activation (dqn_per.py:132-143) */
enum { MCTSQ_ACT_RELU = 0, MCTSQ_ACT_TANH = 1 };
/* time features appended in-kernel for the non-_per flavour (mctsq_config_dqn.py:550-587) */
enum { MCTSQ_TF_NONE = 0, MCTSQ_TF_MINUTE = 1, MCTSQ_TF_GAS_HOURS = 2 };

typedef struct {
    int32_t kind;      /* MCTSQ_ENC_* */
    int32_t seq_len;   /* S: 25 (price), 10 (process), 3 (gas/EUA) at the shipped config */
    int32_t data_dim;  /* features per step in the caller's tensor: 1 / 6 / 2 */
    int32_t time_feat; /* MCTSQ_TF_*: in-kernel [sin,cos] columns appended after data_dim */
} mctsq_encoder_desc;

/* TripleEncoderDQN shape (dqn_per.py:146-214). embed_dim must be 64, action_dim <= 16,
 * hidden_units a multiple of 16 and <= 256, hidden_layers in 1..4. */
typedef struct {
    mctsq_encoder_desc enc[3]; /* price, process, gas_eua */
    int32_t embed_dim;
    int32_t hidden_layers;
    int32_t hidden_units;
    int32_t action_dim;
    int32_t activation;
    int32_t reserved;
} mctsq_net_desc;

int mctsq_abi_version(void);
const char* mctsq_last_error(void);

/* ---- parameter layout ------------------------------------------------------------------
 * Parameters live in ONE flat fp32 device buffer, tensors in torch named_parameters() order and
 * native torch layout, each tensor start padded to a multiple of 16 floats (pads stay zero).
 * Fills offs[i]/sizes[i] (in floats) for tensor i; returns the tensor count (<0 on error) and
 * the padded total in *total.  Replaces nothing in the reference: it is what lets
 * DQNModel.policy_net.parameters() (dqn_per.py:383-398) be views into the buffer the kernels,
 * Adam and the RCCL all-reduce operate on. */
int mctsq_param_layout(const mctsq_net_desc* net, int64_t* offs, int64_t* sizes, int cap,
                       int64_t* total);

/* [S,2] fp32 sin/cos columns the non-_per flavour appends (mctsq_config_dqn.py:550-587 with the
 * defaults its callers use); host-side, for tests and documentation. */
int mctsq_time_features(int mode, int seq_len, float* out);

/* ---- Q-network forward: TripleEncoderDQN.forward (dqn_per.py:216-236) fused with the leaf
 * epilogue of MCTSQ._evaluate / _select (mctsq_config_mcts_per.py:291-296, 367-370).
 *   d_price [B,S0,F0], d_process [B,S1,F1], d_gas [B,S2,F2] fp32, contiguous.
 *   d_q [B,A] fp32 (required); d_vmax [B] fp32 = max_a Q; d_argmax [B] int32 = greedy action;
 *   d_prior [B,A] fp32 = softmax(Q) - each of the last three may be NULL.
 *   d_ws: workspace of mctsq_q_forward_workspace_bytes(net, B) bytes. */
int64_t mctsq_q_forward_workspace_bytes(const mctsq_net_desc* net, int64_t B);
int mctsq_q_forward(const mctsq_net_desc* net, const float* d_params, const float* d_price,
                    const float* d_process, const float* d_gas, int64_t B, float* d_q,
                    float* d_vmax, int32_t* d_argmax, float* d_prior, void* d_ws, int64_t ws_bytes,
                    void* stream);

/* Same, and records the caller's hipEvent_t `ev_encoders_done` (may be NULL) on the stream
 * between the encoder kernels and the head kernel, so bench.py can time the dominant kernel
 * with HIP events (no reference counterpart: measurement hook). */
int mctsq_q_forward_ex(const mctsq_net_desc* net, const float* d_params, const float* d_price,
                       const float* d_process, const float* d_gas, int64_t B, float* d_q,
                       float* d_vmax, int32_t* d_argmax, float* d_prior, void* d_ws,
                       int64_t ws_bytes, void* stream, void* ev_encoders_done);

/* ---- prepared weights.  The GRU recurrence (W_hh), the conv encoder's second convolution and the
 * transformer FFN multiply through the fp16 matrix instruction on a two-way split of every fp32 weight,
 * rows scaled by powers of two into the fp16 range; the attention projections and the head of the fused
 * gru/gru/gru kernel through the bf16 matrix instruction on an exact three-way split (DESIGN.md
 * section 3; both forms are fp32-accurate).  The buffer's contents are private to the library
 * version that wrote them.  mctsq_q_forward splits
 * the weights into its workspace on every call; a caller whose parameters stay fixed over many
 * evaluations (MCTS leaf evaluation between learner steps, mctsq_config_mcts_per.py:291-296,
 * 367-370) prepares them once and calls mctsq_q_forward_prepared.  d_prep: mctsq_q_prepared_bytes
 * bytes, 16-byte aligned; it must be re-prepared after every change of d_params.  No reference
 * counterpart (torch keeps one weight tensor). */
int64_t mctsq_q_prepared_bytes(const mctsq_net_desc* net);
int mctsq_q_prepare(const mctsq_net_desc* net, const float* d_params, void* d_prep, int64_t prep_bytes,
                    void* stream);
int mctsq_q_forward_prepared(const mctsq_net_desc* net, const float* d_params, const void* d_prep,
                             const float* d_price, const float* d_process, const float* d_gas,
                             int64_t B, float* d_q, float* d_vmax, int32_t* d_argmax, float* d_prior,
                             void* d_ws, int64_t ws_bytes, void* stream, void* ev_encoders_done);

/* ---- one encoder alone: ConvAttentionEnc / GRUAttentionEnc / TransformerEnc / GasEUAEncoder.forward
 * (dqn_per.py:39-55, 66-77, 90-101, 122-130; module-level names of the reference boundary).
 * d_enc_params holds ONLY that encoder's tensors, in its named_parameters() order and native torch layout,
 * each tensor start padded to 16 floats: mctsq_encoder_param_layout fills offs/sizes (floats) and returns
 * the tensor count.  d_x [B, seq_len, data_dim] fp32 -> d_out [B, 64] fp32.  MCTSQ_ENC_MLP is the gas/EUA
 * Linear(3 F)-ReLU-Linear-ReLU block (seq_len must be 3).  d_ws: mctsq_encoder_workspace_bytes bytes. */
int mctsq_encoder_param_layout(const mctsq_encoder_desc* enc, int64_t* offs, int64_t* sizes, int cap,
                               int64_t* total);
int64_t mctsq_encoder_workspace_bytes(const mctsq_encoder_desc* enc, int64_t B);
int mctsq_encoder_forward(const mctsq_encoder_desc* enc, const float* d_enc_params, const float* d_x,
                          int64_t B, float* d_out, void* d_ws, int64_t ws_bytes, void* stream);

/* ---- PER-DQN learner step on an already gathered batch: DQNModel.update (dqn_per.py:455-472)
 * = policy forward (activations kept), target forward + max, TD target and IS-weighted MSE,
 * full backward into a flat gradient buffer laid out like the parameters.
 *   d_s* / d_n*: state / next-state windows as for mctsq_q_forward; d_act int32 [B];
 *   d_rew, d_done, d_isw fp32 [B]; gamma = discount.
 *   Outputs: d_grad [total] fp32 (overwritten, UNclipped, already divided by B as the mean
 *   does), d_loss [1] fp32, d_abs_td [B] fp32 (new priorities, dqn_per.py:474),
 *   d_qsa [B] and d_target [B] fp32 (optional, may be NULL).
 * loss_scale: multiplies the loss AND its gradient.  Data-parallel: pass B_local / B_global on every rank and
 * all-reduce(SUM) gradient and loss - both then are those of the global batch mean; else 1.
 * Besides `stream`, the call uses two helper streams the library keeps per (device, stream), forked from and joined
 * back to `stream` with events inside the call (also on every error path). */
int64_t mctsq_dqn_grad_workspace_bytes(const mctsq_net_desc* net, int64_t B);
int mctsq_dqn_grad(const mctsq_net_desc* net, const float* d_params, const float* d_target_params,
                   const float* d_sprice, const float* d_sprocess, const float* d_sgas,
                   const float* d_nprice, const float* d_nprocess, const float* d_ngas,
                   const int32_t* d_act, const float* d_rew, const float* d_done,
                   const float* d_isw, float gamma, float loss_scale, int64_t B, float* d_grad,
                   float* d_loss, float* d_abs_td, float* d_qsa, float* d_target, void* d_ws,
                   int64_t ws_bytes, void* stream);

/* ---- the same step with the weights prepared once per parameter change instead of inside every call.
 * mctsq_dqn_prepare fills, in one launch for the shipped gru/gru/gru network: d_prep_policy and d_prep_target
 * (mctsq_q_prepared_bytes each; the mctsq_q_prepare layout, so the policy's buffer also serves
 * mctsq_q_forward_prepared between learner steps) and d_prep_bwd (mctsq_dqn_prepared_bwd_bytes: the fragments
 * the backward pass multiplies by).  Call it after every change of the parameters (Adam, Polyak, load).
 * mctsq_dqn_grad_prepared = mctsq_dqn_grad on those buffers; additionally, if d_sumsq_parts is not NULL
 * (mctsq_dqn_grad_sumsq_parts doubles), it receives the pieces of the gradient's sum of squares, produced for
 * free by the ordered reduction, *n_sumsq_parts their count: pass both to mctsq_clip_adam_polyak_ex, which
 * then needs no pass over the gradient for the norm (single GPU; a data-parallel step all-reduces the gradient
 * in between and passes NULL). */
int64_t mctsq_dqn_prepared_bwd_bytes(const mctsq_net_desc* net);
int mctsq_dqn_prepare(const mctsq_net_desc* net, const float* d_params, const float* d_target_params,
                      void* d_prep_policy, void* d_prep_target, void* d_prep_bwd, void* stream);
int64_t mctsq_dqn_grad_sumsq_parts(const mctsq_net_desc* net);
int mctsq_dqn_grad_prepared(const mctsq_net_desc* net, const float* d_params, const float* d_target_params,
                            const void* d_prep_policy, const void* d_prep_target, const void* d_prep_bwd,
                            const float* d_sprice, const float* d_sprocess, const float* d_sgas,
                            const float* d_nprice, const float* d_nprocess, const float* d_ngas,
                            const int32_t* d_act, const float* d_rew, const float* d_done,
                            const float* d_isw, float gamma, float loss_scale, int64_t B, float* d_grad,
                            float* d_loss, float* d_abs_td, float* d_qsa, float* d_target,
                            double* d_sumsq_parts, int64_t* n_sumsq_parts, void* d_ws, int64_t ws_bytes,
                            void* stream);

/* ---- DQNModel.update() (dqn_per.py:401-476) as ONE call on a device-resident replay ring: sample (uniforms
 * d_u) -> batch assembly -> loss + backward -> clip_grad_norm_ + Adam (+ Polyak when tau >= 0; the reference syncs
 * the target right after update(), mctsq_config_mcts_per.py:181) -> weight fragments of the new parameters ->
 * priority write-back.  Same kernels, same results as mctsq_per_sample_batch, mctsq_dqn_grad_prepared,
 * mctsq_clip_adam_polyak_ex, mctsq_dqn_prepare and mctsq_per_update called in that order; it exists because at
 * the training batch size the HOST (five binding calls, per-step buffers) bounds the step otherwise.
 * All pointers are device pointers owned by the caller; d_ws: mctsq_per_dqn_step_workspace_bytes, 256-byte
 * aligned (the sampled batch, indices, weights, |td| and all scratch live there); d_loss [1] fp32 out;
 * d_gnorm [1] fp32 out (optional); d_wmax [1] fp64 out (optional).  adam_step is 1-based. */
typedef struct {
    /* replay ring (see "prioritized replay" below) */
    uint64_t* d_tree; double* d_prio; int32_t* d_stamp;
    const float* d_states; const float* d_next_states; const int32_t* d_actions; const float* d_rewards;
    const float* d_dones;
    int64_t capacity, head, size, mass_capacity;
    double alpha, beta;
    void* d_per_ws; int64_t per_ws_bytes;
    /* learner state: flat parameter-layout buffers and the prepared fragments (mctsq_dqn_prepare) */
    float* d_params; float* d_target_params; float* d_grad; float* d_m; float* d_v;
    void* d_prep_policy; void* d_prep_target; void* d_prep_bwd;
    double gamma, lr, beta1, beta2, eps, max_norm, tau;
    int64_t adam_step;
    /* this step */
    const double* d_u; int64_t B;
    float* d_loss; float* d_gnorm; double* d_wmax;
    void* d_ws; int64_t ws_bytes;
    /* 0: the fragments are re-split at the END of the call (they are current when it returns, as above).
     * 1: deferred re-split for back-to-back steps: the call re-splits the fragments at its START, on a helper lane
     *    under the sampling launch (neither depends on the other), and leaves them STALE at its end - pass 1 again
     *    on the next call, or call mctsq_dqn_prepare / mctsq_q_prepare before using the fragments elsewhere.  Same
     *    results; the re-split launch and its boundary leave the step's serial chain. */
    int64_t resplit_first;
    /* ---- data-parallel form only (mctsq_per_dqn_step_begin / _finish below); zero for mctsq_per_dqn_step.
     * world ranks each own one replay shard; B (above) is the number of GLOBAL draws in d_u (identical on every
     * rank), bcap the rows this rank provisions for its share of them (a multiple of 16; unused rows are padding). */
    int64_t world, rank, bcap;
    const uint64_t* d_shard_mass; const int64_t* d_shard_size; /* [world]: all-gathered mctsq_per_total_mass pairs */
    int32_t* d_count;        /* [2] out: draws this rank kept / draws that did not fit bcap (must stay 0) */
    int32_t* d_draw;         /* [bcap] out: positions in d_u of the rows (-1 = padding) */
    float* d_exchange;       /* [2 * world] fp32, directly behind the loss word's 16-float block (see below) */
} mctsq_step_args;
int64_t mctsq_per_dqn_step_workspace_bytes(const mctsq_net_desc* net, int64_t B);
int mctsq_per_dqn_step(const mctsq_net_desc* net, const mctsq_step_args* args, void* stream);

/* ---- the same step for W data-parallel ranks (no reference counterpart: SURVEY.md section 8e), as TWO calls around
 * the one gradient collective.  Per step and rank:
 *     mctsq_per_total_mass -> all-gather of the (mass, size) pairs             [collective 1: 16 bytes per rank]
 *     mctsq_per_dqn_step_begin : shard selection of the B global draws (mctsq_per_shard_select), bcap rows sampled and
 *         gathered from this rank's shard with the totals of the whole buffer (mctsq_per_sample_shard), loss and
 *         backward with loss_scale = bcap / B on the RAW importance weights (N p_i)^-beta, priority write-back.
 *         Leaves ONE contiguous fp32 block ready to be summed over ranks:
 *             d_grad [total] | d_loss [16] (word 0 used) | d_exchange [2 * world]
 *         (the caller lays the three out in that order: d_loss = d_grad + total, d_exchange = d_loss + 16), where
 *         d_exchange[2 r], [2 r + 1] = (hi, lo) fp32 halves of rank r's largest raw weight, the other ranks' slots 0.
 *     all-reduce(SUM) over that block                                          [collective 2: RCCL]
 *     mctsq_per_dqn_step_finish : max over the exchanged slots = the global weight maximum the reference divides by
 *         (dqn_per.py:301); loss and gradient are linear in 1 / max, so the division is applied here: loss word and
 *         gradient are scaled in place, then clip_grad_norm_ / Adam (/ Polyak when tau >= 0) as in the one-call form.
 * Every rank ends with identical parameters.  Against mctsq_per_dqn_step on the concatenated buffer: same indices,
 * same |td| priorities; the weights enter as fp32(w_raw) / max instead of fp32(w_raw / max), i.e. parameters agree
 * to fp32 rounding (tests/test_gpu_dataparallel.py).  d_ws: mctsq_per_dqn_step_workspace_bytes(net, bcap). */
int mctsq_per_dqn_step_begin(const mctsq_net_desc* net, const mctsq_step_args* args, void* stream);
int mctsq_per_dqn_step_finish(const mctsq_net_desc* net, const mctsq_step_args* args, void* stream);

/* ---- clip_grad_norm_(max_norm) + Adam.step + Polyak target update (dqn_per.py:471-472,
 * 510-516) over the flat buffers, n floats each.  step is 1-based.  tau < 0 skips the target
 * update; tau == 1 is the hard copy of mctsq_config_dqn.py:547.  d_ws: mctsq_optim_workspace_bytes() of scratch.
 * d_gnorm (optional) receives the pre-clip global L2 norm.  Hyper-parameters are doubles because
 * the reference's are Python floats: torch forms 1-beta in fp64 before rounding to fp32. */
int64_t mctsq_optim_workspace_bytes(void);
int mctsq_clip_adam_polyak(float* d_params, const float* d_grad, float* d_m, float* d_v,
                           float* d_target, int64_t n, int64_t step, double lr, double beta1,
                           double beta2, double eps, double max_norm, double tau, float* d_gnorm,
                           void* d_ws, int64_t ws_bytes, void* stream);
int mctsq_clip_adam_polyak_ex(float* d_params, const float* d_grad, float* d_m, float* d_v,
                              float* d_target, int64_t n, int64_t step, double lr, double beta1,
                              double beta2, double eps, double max_norm, double tau, float* d_gnorm,
                              const double* d_sumsq_parts, int64_t nparts, void* d_ws, int64_t ws_bytes,
                              void* stream);
/* Polyak / hard sync alone: target <- tau*policy + (1-tau)*target (dqn_per.py:515-516). */
int mctsq_polyak(const float* d_params, float* d_target, int64_t n, double tau, void* stream);

/* ---- prioritized replay: PrioritizedReplayBuffer (dqn_per.py:238-321) as a device-resident
 * ring + 64-ary integer sum-tree.  Logical index 0 is the OLDEST entry (deque(maxlen)).
 * Priorities are fp64; probability mass m = rint(min(prio^alpha, 256) * 2^shift) as uint64 (deterministic
 * pow) so tree sums are exact and order-independent.  LIMIT: prio^alpha is clamped at 256
 * (|td| above ~1.03e4 at alpha = 0.6; the reference has no clamp, dqn_per.py:286-301) so that capacity * mass
 * cannot overflow 64 bits; the importance weights are computed from the same clamped value, i.e. they stay the
 * exact weights of the law that is actually sampled.
 *
 * The tree is one uint64 device array of mctsq_per_tree_words(capacity) words, caller-zeroed
 * before first use; d_prio is fp64 [capacity] in PHYSICAL slot order; head = physical slot of
 * logical index 0; size = live entries.
 *
 * Workspace of the sampling calls (d_ws, mctsq_per_workspace_bytes(capacity, B) bytes, 16-byte aligned): its
 * first 64 bytes are control words that must be ZERO when the buffer is first handed to the library and are
 * left zero by every call; the caller keeps the buffer per replay ring and does not write to it.  Two sampling
 * calls that share one workspace must not overlap in time. */
int64_t mctsq_per_tree_words(int64_t capacity);
int mctsq_per_mass_shift(int64_t capacity);
/* mass_capacity (push / sample / update): the capacity the fixed-point scale is derived from -
 * pass `capacity` for a single buffer, the SUM of all shard capacities when several shards are
 * sampled as one buffer (sharded entry points below), so the concatenated mass cannot overflow. */
/* push K entries (dqn_per.py:265-267): they take physical slots (head+size+i) % capacity and get
 * priority *d_maxprio (device scalar fp64 holding max over live priorities, default 1.0); the caller
 * copies the state rows itself and advances head/size on the host:
 * size' = min(size+K, cap), head' = (head + max(0,size+K-cap)) % cap. */
int mctsq_per_push(uint64_t* d_tree, double* d_prio, const double* d_maxprio, int64_t capacity,
                   int64_t head, int64_t size, int64_t K, double alpha, int64_t mass_capacity,
                   void* stream);
/* recompute *d_maxprio = max(prio[live]) (1.0 if empty): the O(N) max() of dqn_per.py:265.  d_ws: the ring's
 * sampling workspace (>= mctsq_per_workspace_bytes(capacity, 1) bytes; the control words are not touched). */
int mctsq_per_refresh_max(const double* d_prio, int64_t capacity, int64_t head, int64_t size,
                          double* d_maxprio, void* d_ws, int64_t ws_bytes, void* stream);
/* sample (dqn_per.py:283-301): d_u fp64 [B] uniforms in [0,1) -> d_idx int64 [B] LOGICAL
 * indices, d_phys int64 [B] physical slots, d_w fp32 [B] IS weights (N p_i)^-beta / max.
 * d_wmax (optional, fp64 [1]) receives that max before normalisation.  One launch for B <= 16384. */
int mctsq_per_sample(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head,
                     int64_t size, const double* d_u, int64_t B, double alpha, double beta,
                     int64_t mass_capacity, int64_t* d_idx, int64_t* d_phys, float* d_w,
                     double* d_wmax, void* d_ws, int64_t ws_bytes, void* stream);
/* mctsq_per_sample and mctsq_per_gather in ONE launch (dqn_per.py:283-305 + the batch assembly of
 * :415-452): the wave that finds a slot also copies its transition into the learner's batch tensors. */
int mctsq_per_sample_batch(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head,
                           int64_t size, const double* d_u, int64_t B, double alpha, double beta,
                           int64_t mass_capacity, int64_t* d_idx, int64_t* d_phys, float* d_w,
                           double* d_wmax, const float* d_states, const float* d_next_states,
                           const int32_t* d_actions, const float* d_rewards, const float* d_dones,
                           float* d_sprice, float* d_sprocess, float* d_sgas, float* d_nprice,
                           float* d_nprocess, float* d_ngas, int32_t* d_act, float* d_rew, float* d_done,
                           void* d_ws, int64_t ws_bytes, void* stream);
/* update_priorities (dqn_per.py:307-314): prio[idx[i]] = p[i] sequentially (last duplicate
 * wins); d_idx LOGICAL int64 [B], d_newprio fp32 [B].  d_stamp: int32 [capacity] scratch owned by
 * the buffer, all -1 on entry (the call leaves it all -1 again).  Indices outside [0, size) name no stored
 * transition and are ignored (the reference raises IndexError; host-side callers check before calling).
 * One launch for B <= 2048 (exact last-writer table in LDS); marking + apply launches above. */
int mctsq_per_update(uint64_t* d_tree, double* d_prio, int32_t* d_stamp, int64_t capacity,
                     int64_t head, int64_t size, const int64_t* d_idx, const float* d_newprio,
                     int64_t B, double alpha, int64_t mass_capacity, void* stream);

/* ---- sharded replay (no reference counterpart: SURVEY.md section 8e).  W ranks each own one ring;
 * together they are sampled as ONE buffer whose logical order is the concatenation of the shards
 * in rank order, with the same exact-integer inverse CDF:
 *   1. every rank: mctsq_per_total_mass -> d_mass[0] = its root mass, d_mass[1] = size (the pair the ranks
 *      all-gather, produced on the device so that no host value has to be copied in);
 *   2. every rank: mctsq_per_shard_targets on the SAME B uniforms -> which draws fall into its
 *      interval of the concatenated mass line (d_mine) and their local integer targets;
 *   3. compact its own draws, mctsq_per_sample_at with n_total / mass_total of the whole buffer
 *      -> local indices, raw fp64 IS weights and their local max;
 *   4. all-reduce(MAX) the max, mctsq_per_normalize_weights -> fp32 weights. */
int mctsq_per_total_mass(const uint64_t* d_tree, int64_t capacity, int64_t size, uint64_t* d_mass,
                         void* stream);
int mctsq_per_shard_targets(const double* d_u, int64_t B, const uint64_t* d_shard_mass, int world,
                            int rank, uint64_t* d_targets, int32_t* d_mine, void* stream);
int mctsq_per_sample_at(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head,
                        int64_t size, const uint64_t* d_targets, int64_t B, double alpha, double beta,
                        int64_t mass_capacity, double n_total, double mass_total, int64_t* d_idx,
                        int64_t* d_phys, double* d_w64, double* d_wmax, void* stream);
int mctsq_per_normalize_weights(const double* d_w64, int64_t B, const double* d_wmax, float* d_w,
                                void* stream);
/* The same protocol WITHOUT host round trips (steps 2-3 above need the number of own draws on the host):
 *   mctsq_per_shard_select: the rank's draws compacted to the front of d_targets [bcap] (draw order kept),
 *     d_draw [bcap] = their positions in d_u (-1 = padding), d_count[0] = how many, d_count[1] = how many did
 *     NOT fit bcap (must be 0: the caller sizes bcap for its share of the batch and checks this word lazily);
 *   mctsq_per_sample_shard: bcap rows from those targets with the totals of the whole buffer read from the
 *     all-gathered DEVICE arrays d_shard_mass / d_shard_size [world]; rows >= d_count[0] are padding
 *     (d_idx = -1, d_phys = 0, weight 0: they contribute exactly nothing to loss, gradient and priorities);
 *     also gathers the transitions like mctsq_per_sample_batch.  d_wmax = local max of the raw weights. */
int mctsq_per_shard_select(const double* d_u, int64_t B, const uint64_t* d_shard_mass, int world, int rank,
                           int64_t bcap, uint64_t* d_targets, int32_t* d_draw, int32_t* d_count,
                           void* stream);
int mctsq_per_sample_shard(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head,
                           int64_t size, const uint64_t* d_targets, const int32_t* d_count, int64_t bcap,
                           double alpha, double beta, int64_t mass_capacity,
                           const uint64_t* d_shard_mass, const int64_t* d_shard_size, int world,
                           int64_t* d_idx, int64_t* d_phys, double* d_w64, double* d_wmax,
                           const float* d_states, const float* d_next_states, const int32_t* d_actions,
                           const float* d_rewards, const float* d_dones, float* d_sprice,
                           float* d_sprocess, float* d_sgas, float* d_nprice, float* d_nprocess,
                           float* d_ngas, int32_t* d_act, float* d_rew, float* d_done, void* d_ws,
                           int64_t ws_bytes, void* stream);
/* gather the sampled transitions (dqn_per.py:293, 415-452): rows are 96-float padded
 * [price 25 | process 10x6 | gas 3x2 | pad] records; outputs are the six window tensors the
 * learner consumes plus actions / rewards / dones. */
int mctsq_per_gather(const float* d_states, const float* d_next_states, const int32_t* d_actions,
                     const float* d_rewards, const float* d_dones, const int64_t* d_phys,
                     int64_t B, float* d_sprice, float* d_sprocess, float* d_sgas, float* d_nprice,
                     float* d_nprocess, float* d_ngas, int32_t* d_act, float* d_rew, float* d_done,
                     void* stream);
int64_t mctsq_per_workspace_bytes(int64_t capacity, int64_t B);

/* ---- mctsq_per_sample without a workspace argument, allocating nothing.  NO counterpart in the reference: it
 * exists for callers that cannot allocate or carry scratch memory - code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  Same inputs, outputs and results as mctsq_per_sample; the sampling
 * workspace (control words + one raw fp64 weight per draw) is a fixed array inside the library's device image:
 * 128.06 KiB per GPU, held until the process ends, so B <= mctsq_per_sample_noalloc_max_draws() (16384) - larger
 * batches are refused with an error.  All calls on one GPU share that array: calls that overlap in time on one
 * GPU (different streams) race; serialise them.  The array that is used is the one of the device `stream` belongs to.
 * The first call on a device looks the array's address up (a synchronous runtime query, nothing is enqueued for it):
 * make that call outside stream capture; later calls only launch. */
int64_t mctsq_per_sample_noalloc_max_draws(void);
int mctsq_per_sample_noalloc(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head,
                             int64_t size, const double* d_u, int64_t B, double alpha, double beta,
                             int64_t mass_capacity, int64_t* d_idx, int64_t* d_phys, float* d_w,
                             double* d_wmax, void* stream);

#define MCTSQ_STATE_ROW 96 /* floats per stored state record (91 used) */

#ifdef __cplusplus
}
#endif
#endif /* MCTSQ_HIP_H */
