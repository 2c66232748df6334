/* Host-side helper of the lock-step tree search (mcts_q_amd/lib/libmctsq_tree.so, plain C, no GPU code).
 * Not part of the device boundary (include/mctsq_hip.h): the tree logic stays on the host by design; these two
 * entry points only take its per-iteration loops out of the interpreter and spread K independent trees over the
 * host cores (OpenMP, from K = 256).  Trees are flat arrays indexed by node id: child[node][5] in creation
 * order (-1 = none), parent[node] (-1 = root).  Bound by mcts_q_amd/treesearch.py through ctypes. */
#ifndef MCTSQ_TREE_H
#define MCTSQ_TREE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* threads used by the parallel loops of this library (MCTSQ_TREE_THREADS, else min(OpenMP default, CPUs of this process /
 * LOCAL_WORLD_SIZE)) */
int mctsq_tree_threads(void);

/* Selection (replaces the walk of MCTSQ._select, src/mctsq_config_mcts_per.py:266-308, for K trees at once):
 * from root[k] descend to the child maximising  total/visits + c_puct * prior_a * sqrt(sum visits) / (1 + visits),
 * c_puct = log((1 + sum visits + c_base) / c_base) + c_init, first maximum in creation order, until a terminal node
 * (done), a node at depth >= maxd, or a node with an untried legal action (tried != legal).  cur_out[k] = that node. */
void mctsq_tree_select(int64_t K, const int64_t* root, int64_t maxd, double c_base, double c_init,
                       const uint8_t* done, const int64_t* depth, const int64_t* legal, const int64_t* tried,
                       const int64_t* child, const int64_t* visits, const double* total, const double* prior_a,
                       int64_t* cur_out);

/* Backup (replaces MCTSQ._backpropagate, src/mctsq_config_mcts_per.py:358-381): every node on the path
 * leaf[k] -> root gains one visit and vmax[leaf[k]]. */
void mctsq_tree_backup(int64_t K, const int64_t* leaf, const double* vmax, const int64_t* parent, int64_t* visits,
                       double* total);

/* Sub-tree reuse (replaces the re-rooting of src/mctsq_config_mcts_per.py:153-167 and reset_depths, :236-247): keeps
 * the nodes under keep_roots[k] (>= 0; -1 = drop tree k), compacts all per-node arrays in place in ascending id order
 * and resets depths relative to the new roots.  rel, newid: scratch, n int64 each.  Returns the new node count. */
int64_t mctsq_tree_retain(int64_t K, int64_t n, const int64_t* keep_roots, int64_t* new_roots, int64_t* parent,
                          int64_t* action, int64_t* visits, double* total, int64_t* depth, uint8_t* done,
                          int64_t* legal, int64_t* tried, int64_t* nchild, int64_t* child, double* vmax,
                          double* prior_a, float* rows, int64_t row_w, double* svec, int64_t D, int64_t* rel,
                          int64_t* newid);

/* ---- Vectorised PtG environment step ---------------------------------------------------------
 * PTGEnv.step of the `_per` environment (env/ptg_gym_env_per.py:422-662, 715-992) is a function of (action, state_c)
 * and static tables; here it runs over n state VECTORS at once.  One state is D = 12 + 6*seq_length doubles:
 *   [0] i  [1] j  [2] k  [3] Meth_State  [4] Meth_T_cat  [5] standby_op  [6] startup_op  [7] part_op  [8] full_op
 *   (data-set ids, order of load_data's op_data_files, src/mctsq_utils.py:126-134: 0 startup_cold .. 16 op12_f_p_f_20)
 *   [9] hot_cold  [10] episode offset in days (act_ep_d)  [11] cumulative reward
 *   [12..] the normalised process window [seq_length][6] (T_CAT, H2_in, CH4_syn, H2_res, H2O_DE, Elec_Heating)
 * Deviations from the reference, all of them state the reference keeps on the environment OBJECT and that therefore
 * leaks between tree nodes there: hot_cold (:454-457), the noon-updated gas/EUA columns (:597-603, derived from the
 * clock here) and cum_rew ride in the state vector / are functions of it, i.e. every state sees the values a
 * sequential episode would give it.  Interval averages use extended-precision prefix sums (np.average: pairwise sums;
 * agreement ~1e-15 relative).  The random index offset of a mode change (:811, 825, 849) is an argument (noise[n],
 * already scaled by `noise`), so the step itself is deterministic. */
typedef struct {
    int64_t step_size, seq_length, seq_step, sim_step, eps_sim_steps, price_len, H, Dd, scenario3;
    double convert_mol_to_Nm3, H_u_CH4, H_u_H2, dt_water, cp_water, rho_water, Molar_mass_CO2, Molar_mass_H2O,
        h_H2O_evap, eeg_el_price, heat_price, o2_price, water_price, min_load_electrolyzer, max_h2_volumeflow, eta_CHP;
    double t_cat_standby, t_cat_startup_cold, t_cat_startup_hot;
    double time1_start_p_f, time2_start_f_p, time_p_f, time_f_p, time1_p_f_p, time2_p_f_p, time3_p_f_p, time34_p_f_p,
        time4_p_f_p, time45_p_f_p, time5_p_f_p, time1_f_p_f, time2_f_p_f, time23_f_p_f, time3_f_p_f, time34_f_p_f,
        time4_f_p_f, time45_f_p_f, time5_f_p_f, i_fully_developed, j_fully_developed;
    double lb[9], ub[9];            /* el, gas, eua, T, h2, ch4, h2_res, h2o, heat */
    double rew_l_b, rew_u_b;
} mctsq_ptg_params;

#define MCTSQ_PTG_NDS 17
#define MCTSQ_PTG_STATE_HEAD 12

/* op: the 17 process data sets concatenated, [sum ds_len][7] (time, T_cat, n_h2, n_ch4, n_h2_res, m_DE, Pel);
 * e_r_b [3][price_len][H], g_e [2][2][Dd] as Preprocessing builds them (src/mctsq_utils.py:300-345).  Everything is
 * copied.  Returns NULL on bad geometry (a data set shorter than one step, window longer than one step). */
void* mctsq_ptg_create(const mctsq_ptg_params* p, const double* op, const int64_t* ds_len, const double* e_r_b,
                       const double* g_e);
void mctsq_ptg_destroy(void* env);
/* reset (:664-705, 135-195): states[n][D] for episodes starting at day offsets d0[n] */
void mctsq_ptg_reset(const void* env, int64_t n, const int64_t* d0, double* states);
/* observation rows of n states: price[n][price_len], proc[n][6*seq_length], gas[n][6] (gas, eua interleaved per
 * time), each with its own row stride in floats; meth[n] = Meth_State.  Returns the number of states whose clock
 * points outside the price tables (the reference raises IndexError there; the caller must). */
int64_t mctsq_ptg_observe(const void* env, int64_t n, const double* states, float* price, int64_t ld_price,
                          float* proc, int64_t ld_proc, float* gas, int64_t ld_gas, int64_t* meth);
/* step (:422-662): next[n][D], observation rows as in observe, reward[n] (normalised, :418), term[n] (:707-712),
 * meth[n].  actions[n] in 0..4; noise[n] may be NULL (= 0).  Returns the count of out-of-table clocks / bad actions. */
int64_t mctsq_ptg_step(const void* env, int64_t n, const double* states, const int64_t* actions, const double* noise,
                       double* next, float* price, int64_t ld_price, float* proc, int64_t ld_proc, float* gas,
                       int64_t ld_gas, double* reward, uint8_t* term, int64_t* meth);
/* the evaluation fields of _get_info (:312-340) that are functions of a state: out[n][8] = el_price_act,
 * gas_price_act, eua_price_act, Pot_Reward, Part_Full, cum_reward, Meth_T_cat, Meth_Hot_Cold */
int64_t mctsq_ptg_info(const void* env, int64_t n, const double* states, double* out);
/* D of this environment; and the step of ONE state into caller-provided slots (what mctsq_ptg_step loops over) */
int64_t mctsq_ptg_state_dim(const void* env);
int mctsq_ptg_step_one(const void* env, const double* st, int64_t action, double nz, double* nx, float* price,
                       float* proc, float* gas, double* reward, uint8_t* term, int64_t* meth);

/* ---- One search iteration of many trees in ONE call (lock-step driver over the vectorised environment) ----------
 * Selection, the choice of an untried action, the environment step and the new node, for trees k_lo <= k < k_hi, then
 * (after the caller evaluated the new leaves on the GPU) value / prior assignment and backup.  Same arithmetic as the
 * separate calls above and as treesearch.py's numpy path, which restate _select / _expand / _evaluate /
 * _backpropagate (src/mctsq_config_mcts_per.py:266-381).  The untried action of tree k in iteration `iteration` is
 * the floor(u * n_untried)-th set bit of legal & ~tried, u from a counter-based generator over (seed, iteration, k):
 * same distribution as the reference's random.choice (:328), a reproducible stream that does not depend on how
 * the trees are split over calls.  The caller guarantees f->cap >= f->n + (k_hi - k_lo). */
typedef struct {
    int64_t K, cap, n, D, row_w;
    int64_t *root, *parent, *action, *visits, *depth, *legal, *tried, *nchild, *child;
    double *total, *vmax, *prior_a, *svec, *reward;
    uint8_t* done;
    float* rows;                /* [cap][row_w] observation row of every node */
} mctsq_forest;

/* cur[k] (k_lo <= k < k_hi) = the leaf of this iteration (new node, or the terminal / depth-limit node selection stopped
 * at).  slot_node[s] = id of the s-th new node; its observation is also written to slot s of `stage`, three blocks
 * price [stage_cap][P] | process [stage_cap][6L] | gas [stage_cap][6] (the input layout of mctsq_q_forward).  Returns
 * the number of new nodes, or -1 - (count of failed environment steps). */
int64_t mctsq_lockstep_expand(mctsq_forest* f, const void* env, int64_t maxd, double c_base, double c_init,
                              uint64_t seed, uint64_t iteration, int64_t k_lo, int64_t k_hi, double noise_sigma,
                              int64_t* cur, int64_t* slot_node, float* stage, int64_t stage_cap, int64_t* treedepth);
/* vmax_out [n_new], prior_out [n_new][n_act]: the network outputs for the slots; then backup from cur[k]. */
void mctsq_lockstep_backup(mctsq_forest* f, int64_t k_lo, int64_t k_hi, const int64_t* cur, const int64_t* slot_node,
                           int64_t n_new, const float* vmax_out, const float* prior_out, int64_t n_act);
/* ---- all iterations of one search in ONE call: the loop over (iteration, block of trees) around expand / backup, with
 * the leaf batches launched through function pointers the caller supplies (this library links no GPU runtime):
 * `forward` = mctsq_q_forward_prepared (include/mctsq_hip.h), `event_record` / `event_sync` / `memcpy_async` =
 * hipEventRecord / hipEventSynchronize / hipMemcpyAsync, or host functions of the same shape in tests.  Block h of
 * `halves` (1..4) covers trees [h*K/halves, (h+1)*K/halves) and owns staging area, outputs, workspace and event h;
 * while block h's leaves are on the device, the other blocks are advanced on the host.  cur, slots: K int64 of scratch each.  stage[h]: pinned host memory in
 * the forward's input layout (three blocks of `cap` rows); dev_in[h]: device copy of it, or NULL to let the kernel
 * read the pinned staging area itself; vmax[h] / prior[h]: host-visible outputs.  prof (NULL or 4 doubles) receives
 * seconds spent waiting for the device, in backup, in expand, in launching.  Returns 0, or < 0: -1 - n for n failed
 * environment steps / capacity, -1000 - rc for a failing forward / copy / event call. */
typedef struct {
    void *forward, *event_record, *event_sync, *memcpy_async;
    const void *net, *params, *prep;
    void* stream[4];            /* one stream per block: the blocks' leaf batches (a few CUs each) then overlap on the device */
    int64_t halves, cap, ws_bytes, n_act;
    float *stage[4], *dev_in[4], *q[4], *vmax[4], *prior[4];
    int32_t* argmax[4];
    void *ws[4], *event[4];
    /* optional: completion through a word in pinned host memory instead of an event.  stream_write32 =
     * hipStreamWriteValue32 (or a host function of that shape): behind every leaf batch of block h the stream writes the
     * next sequence number to *done_flag[h] (4-byte aligned, device-accessible host memory; its current value is the
     * starting point), and the search waits by reading that word.  NULL: event_record / event_sync as above. */
    void* stream_write32;
    uint32_t* done_flag[4];
} mctsq_leaf_eval;
int64_t mctsq_lockstep_search(mctsq_forest* f, const void* env, int64_t maxd, double c_base, double c_init,
                              uint64_t seed, uint64_t iteration0, int64_t iterations, double noise_sigma,
                              const mctsq_leaf_eval* ev, int64_t* cur, int64_t* slots, int64_t* treedepth,
                              double* prof);
/* the uniform in [0, 1) of (seed, iteration, k, stream): stream 0 = action choice, 1 / 2 = the Gaussian index noise */
double mctsq_lockstep_uniform(uint64_t seed, uint64_t iteration, uint64_t k, uint64_t stream);

#ifdef __cplusplus
}
#endif
#endif
