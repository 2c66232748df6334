/* One lock-step search iteration per call (include/mctsq_tree.h): selection, untried-action choice, vectorised
 * environment step and node creation for a block of trees, and value assignment + backup after the leaf batch was
 * evaluated.  Restates the loop body of MCTSQ.predict (/root/reference/src/mctsq_config_mcts_per.py:249-254) with
 * _select (:266-308), _expand (:312-356), the value / prior reads of _evaluate (:291-296, 367-370) and _backpropagate
 * (:372-381) for many trees; per-tree arithmetic is that of tree_ops.c / treesearch.py. */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <stdlib.h>
#include <omp.h>
#include "../../include/mctsq_tree.h"

#define N_ACT 5
static const int64_t LEGAL_OF_STATE[5] = {0x07, 0x07, 0x1B, 0x1B, 0x1B};     /* get_legal_actions (:524-546) */

#define tree_threads mctsq_tree_threads

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

double mctsq_lockstep_uniform(uint64_t seed, uint64_t iteration, uint64_t k, uint64_t stream) {
    const uint64_t x = mix64(seed ^ mix64(iteration * 0xD1342543DE82EF95ULL + stream) ^ mix64(k + 0x632BE59BD9B4E019ULL));
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/* c_puct(tv) = log((1 + tv + c_base) / c_base) + c_init and sqrt(tv) for tv < PUCT_TAB, tabulated with the very
 * expressions of the walk below (same bits), rebuilt when (c_base, c_init) change: a logarithm and a square root per
 * level and tree were half of the selection's arithmetic.  One table per process, filled by the calling thread before
 * the parallel region (the search driver is single-threaded). */
#define PUCT_TAB 8192
static double g_puct[PUCT_TAB], g_sqrt[PUCT_TAB], g_tab_base = -1.0, g_tab_init = -1.0;
static void puct_table(double c_base, double c_init) {
    if (g_tab_base == c_base && g_tab_init == c_init) return;
    for (int64_t tv = 0; tv < PUCT_TAB; ++tv) {
        g_puct[tv] = log((1.0 + (double)tv + c_base) / c_base) + c_init;
        g_sqrt[tv] = sqrt((double)tv);
    }
    g_tab_base = c_base; g_tab_init = c_init;
}

static inline int64_t select_leaf(const mctsq_forest* f, int64_t cur, int64_t maxd, double c_base, double c_init) {
    for (;;) {
        if (f->done[cur] || f->depth[cur] >= maxd || f->tried[cur] != f->legal[cur]) return cur;
        const int64_t* ch = f->child + cur * N_ACT;
        int64_t tv = 0;
        for (int a = 0; a < N_ACT; ++a)
            if (ch[a] >= 0) tv += f->visits[ch[a]];
        const double c_puct = tv < PUCT_TAB ? g_puct[tv] : log((1.0 + (double)tv + c_base) / c_base) + c_init;
        const double sq = tv < PUCT_TAB ? g_sqrt[tv] : sqrt((double)tv);
        double best = -INFINITY;
        int64_t pick = -1;
        for (int a = 0; a < N_ACT; ++a) {
            const int64_t c = ch[a];
            if (c < 0) continue;
            const int64_t v = f->visits[c];
            const double vf = (double)v;
            const double mean = v > 0 ? f->total[c] / vf : 0.0;
            const double score = mean + c_puct * f->prior_a[c] * sq / (1.0 + vf);
            if (pick < 0 || score > best) { best = score; pick = c; }
        }
        if (pick < 0) return cur;
        cur = pick;
    }
}

/* value assignment + backup of ONE tree (the body of mctsq_lockstep_backup) */
static inline void backup_tree(mctsq_forest* f, int64_t node, int64_t first, const float* vmax_out, const float* prior_out,
                               int64_t n_act) {
    if (node >= first) {                                            /* this iteration's new leaf: ids are consecutive */
        const int64_t s = node - first;
        f->vmax[node] = (double)vmax_out[s];                        /* q.max().item() (:370) */
        f->prior_a[node] = (double)prior_out[s * n_act + f->action[node]];   /* prior[idx, action].item() (:296) */
    }
    const double value = f->vmax[node];
    while (node >= 0) {
        f->visits[node] += 1;
        f->total[node] += value;
        node = f->parent[node];
    }
}

/* bk_* (optional): the backup of the block's PREVIOUS iteration, done by every thread for its own trees at the start
   of the same parallel region (the partition of the trees over the threads is the same in both phases, so a tree's
   backup and its next selection need no barrier between them): one fork / join per block and iteration instead of two */
static int64_t expand_impl(mctsq_forest* f, const void* env, int64_t maxd, double c_base, double c_init,
                           uint64_t seed, uint64_t iteration, int64_t k_lo, int64_t k_hi, double noise_sigma,
                           int64_t* cur, int64_t* slot_node, float* stage, int64_t stage_cap, int64_t* treedepth,
                           int64_t bk_n_new, const float* bk_vmax, const float* bk_prior, int64_t bk_n_act) {
    const int64_t nk = k_hi - k_lo, D = f->D, W = f->row_w, P = W - 6 - (D - MCTSQ_PTG_STATE_HEAD);
    const int64_t L6 = D - MCTSQ_PTG_STATE_HEAD;
    if (nk <= 0) return 0;
    puct_table(c_base, c_init);
    /* ONE parallel region, one barrier: every thread owns a contiguous run of trees, selects for it and counts its
       expansions; after the barrier it knows its first slot (prefix over the threads' counts), so slots - and with
       them node ids - are in tree order whatever the thread count */
    int64_t n_new = 0, bad = 0, fail = 0;
    const int64_t base = f->n;
    const int64_t bk_first = bk_n_new > 0 ? slot_node[0] : INT64_MAX;      /* read before this call reuses the slots */
    int64_t counts[64];
    int T = tree_threads();
    if (T > 64) T = 64;
    if (nk < 128) T = 1;
#pragma omp parallel num_threads(T) reduction(+ : bad)
    {
        const int nt = omp_get_num_threads(), tid = omp_get_thread_num();
        const int64_t lo = k_lo + nk * tid / nt, hi = k_lo + nk * (tid + 1) / nt;
        int64_t c = 0;
        if (bk_n_new >= 0)
            for (int64_t k = lo; k < hi; ++k) backup_tree(f, cur[k], bk_first, bk_vmax, bk_prior, bk_n_act);
        for (int64_t k = lo; k < hi; ++k) {
            const int64_t leaf = select_leaf(f, f->root[k], maxd, c_base, c_init);
            cur[k] = leaf;
            c += !(f->done[leaf] || f->depth[leaf] >= maxd);
        }
        counts[tid] = c;
#pragma omp barrier
        int64_t s = 0, total = 0;
        for (int t = 0; t < nt; ++t) { if (t < tid) s += counts[t]; total += counts[t]; }
        const int over = total > stage_cap || base + total > f->cap;
        if (tid == 0) { n_new = total; fail = over; }
        if (!over) {
            for (int64_t k = lo; k < hi; ++k) {
                const int64_t par = cur[k];
                if (f->done[par] || f->depth[par] >= maxd) continue;
                const int64_t id = base + s;
                const int64_t mask = f->legal[par] & ~f->tried[par] & 0x1F;
                int nb = 0, bits[N_ACT];
                for (int a = 0; a < N_ACT; ++a)
                    if (mask >> a & 1) bits[nb++] = a;
                int pick = (int)(mctsq_lockstep_uniform(seed, iteration, (uint64_t)k, 0) * nb);
                if (pick > nb - 1) pick = nb - 1;
                const int64_t a = bits[pick];
                double nz = 0.0;
                if (noise_sigma != 0.0) {
                    const double u1 = mctsq_lockstep_uniform(seed, iteration, (uint64_t)k, 1);
                    const double u2 = mctsq_lockstep_uniform(seed, iteration, (uint64_t)k, 2);
                    nz = noise_sigma * sqrt(-2.0 * log(1.0 - u1)) * cos(6.283185307179586 * u2);
                }
                float* row = f->rows + id * W;
                uint8_t term = 0;
                int64_t meth = 0;
                double rew = 0.0;
                bad += mctsq_ptg_step_one(env, f->svec + par * D, a, nz, f->svec + id * D, row, row + P, row + P + L6,
                                          &rew, &term, &meth);
                memcpy(stage + s * P, row, sizeof(float) * (size_t)P);
                memcpy(stage + stage_cap * P + s * L6, row + P, sizeof(float) * (size_t)L6);
                memcpy(stage + stage_cap * (P + L6) + s * 6, row + P + L6, sizeof(float) * 6);
                f->parent[id] = par; f->action[id] = a; f->visits[id] = 0; f->total[id] = 0.0;
                f->depth[id] = f->depth[par] + 1; f->done[id] = term; f->reward[id] = rew;
                f->legal[id] = meth >= 0 && meth < 5 ? LEGAL_OF_STATE[meth] : 0x1F;
                f->tried[id] = 0; f->nchild[id] = 0;
                for (int q = 0; q < N_ACT; ++q) f->child[id * N_ACT + q] = -1;
                f->vmax[id] = 0.0; f->prior_a[id] = 0.0;
                f->child[par * N_ACT + f->nchild[par]] = id;       /* one expansion per tree: no two slots share a parent */
                f->nchild[par] += 1;
                f->tried[par] |= (int64_t)1 << a;
                if (f->depth[id] > treedepth[k]) treedepth[k] = f->depth[id];
                cur[k] = id;
                slot_node[s] = id;
                ++s;
            }
        }
    }
    if (fail) return -1 - n_new;
    if (bad) {
        /* A step failed (clock outside the price tables): take this call's nodes back out of their parents, so the
           forest is exactly what it was before the call (f->n is not advanced; the rows of the unused ids are scratch) */
        for (int64_t s = 0; s < n_new; ++s) {
            const int64_t id = base + s, par = f->parent[id];
            f->nchild[par] -= 1;
            f->child[par * N_ACT + f->nchild[par]] = -1;
            f->tried[par] &= ~((int64_t)1 << f->action[id]);
        }
        for (int64_t k = k_lo; k < k_hi; ++k)
            if (cur[k] >= base) cur[k] = f->parent[cur[k]];
        return -1 - bad;
    }
    f->n = base + n_new;
    return n_new;
}

int64_t mctsq_lockstep_expand(mctsq_forest* f, const void* env, int64_t maxd, double c_base, double c_init,
                              uint64_t seed, uint64_t iteration, int64_t k_lo, int64_t k_hi, double noise_sigma,
                              int64_t* cur, int64_t* slot_node, float* stage, int64_t stage_cap, int64_t* treedepth) {
    return expand_impl(f, env, maxd, c_base, c_init, seed, iteration, k_lo, k_hi, noise_sigma, cur, slot_node, stage,
                       stage_cap, treedepth, -1, NULL, NULL, 0);
}

void mctsq_lockstep_backup(mctsq_forest* f, int64_t k_lo, int64_t k_hi, const int64_t* cur, const int64_t* slot_node,
                           int64_t n_new, const float* vmax_out, const float* prior_out, int64_t n_act) {
    /* Same partition of the trees over the threads as in expand: a tree's nodes are then read and written by one
       thread in both phases and stay in that core's cache (with the backup on the calling thread alone, every line it
       touched had to travel back to the selecting thread: 11 us per 256 trees against 3 us of arithmetic). */
    const int64_t nk = k_hi - k_lo, first = n_new > 0 ? slot_node[0] : INT64_MAX;
    int T = tree_threads();
    if (T > 64) T = 64;
    if (nk < 128) T = 1;
#pragma omp parallel num_threads(T)
    {
        const int nt = omp_get_num_threads(), tid = omp_get_thread_num();
        const int64_t lo = k_lo + nk * tid / nt, hi = k_lo + nk * (tid + 1) / nt;
        for (int64_t k = lo; k < hi; ++k) backup_tree(f, cur[k], first, vmax_out, prior_out, n_act);
    }
}

/* ---- the whole search ------------------------------------------------------------------------------------------- */
typedef int (*forward_fn)(const void*, const float*, const void*, const float*, const float*, const float*, int64_t,
                          float*, float*, int32_t*, float*, void*, int64_t, void*, void*);
typedef int (*event_record_fn)(void*, void*);
typedef int (*event_sync_fn)(void*);
typedef int (*memcpy_fn)(void*, const void*, size_t, int, void*);
typedef int (*write32_fn)(void*, void*, uint32_t, unsigned);

/* wait until the stream has written `want` (or a later sequence number: the comparison is wrap-safe) to the pinned word
 * (mctsq_leaf_eval.done_flag); gives up after MCTSQ_SEARCH_TIMEOUT_S seconds (default 20) */
static double flag_limit_s(void) {
    static double limit = -1.0;
    if (limit < 0.0) {
        const char* e = getenv("MCTSQ_SEARCH_TIMEOUT_S");
        const double v = e ? atof(e) : 0.0;
        limit = v > 0.0 ? v : 20.0;
    }
    return limit;
}
static int wait_flag(const uint32_t* flag, uint32_t want) {
    const double LIMIT = flag_limit_s();
    for (int64_t spin = 0;; ++spin) {
        if ((int32_t)(__atomic_load_n(flag, __ATOMIC_ACQUIRE) - want) >= 0) return 0;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((spin & 0xFFFFF) == 0xFFFFF) {
            static __thread double t_start = 0.0;
            struct timespec t;
            clock_gettime(CLOCK_MONOTONIC, &t);
            const double now = (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
            if (spin == 0xFFFFF) t_start = now;
            else if (now - t_start > LIMIT) return 9999;
        }
    }
}

static inline double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int64_t mctsq_lockstep_search(mctsq_forest* f, const void* env, int64_t maxd, double c_base, double c_init,
                              uint64_t seed, uint64_t iteration0, int64_t iterations, double noise_sigma,
                              const mctsq_leaf_eval* ev, int64_t* cur, int64_t* slots, int64_t* treedepth,
                              double* prof) {
    const int64_t nh = ev->halves < 1 ? 1 : (ev->halves > 4 ? 4 : ev->halves), K = f->K, cap = ev->cap;
    const int64_t L6 = f->D - MCTSQ_PTG_STATE_HEAD, P = f->row_w - 6 - L6;
    int64_t pend[4] = {-1, -1, -1, -1};
    const forward_fn forward = (forward_fn)ev->forward;
    const event_record_fn ev_record = (event_record_fn)ev->event_record;
    const event_sync_fn ev_sync = (event_sync_fn)ev->event_sync;
    const memcpy_fn copy = (memcpy_fn)ev->memcpy_async;
    const write32_fn write32 = (write32_fn)ev->stream_write32;
    uint32_t seq[4] = {0, 0, 0, 0};
    if (write32)
        for (int64_t h = 0; h < nh; ++h) {
            if (!ev->done_flag[h]) return -3;
            seq[h] = *ev->done_flag[h];
        }
    double t_wait = 0, t_backup = 0, t_expand = 0, t_launch = 0;
    int64_t rc = 0;
    for (int64_t it = 0; it <= iterations && rc == 0; ++it) {          /* the extra round only drains */
        for (int64_t h = 0; h < nh && rc == 0; ++h) {
            const int64_t lo = h * K / nh, hi = (h + 1) * K / nh;
            if (hi - lo > cap) return -2;
            if (pend[h] >= 0) {
                double t0 = prof ? now_s() : 0.0;
                if (pend[h] > 0) {
                    const int e = write32 ? wait_flag(ev->done_flag[h], seq[h]) : ev_sync(ev->event[h]);
                    if (e) { rc = -1000 - e; break; }
                }
                double t1 = prof ? now_s() : 0.0;
                if (it == iterations) {                              /* the draining round: nothing follows the backup */
                    mctsq_lockstep_backup(f, lo, hi, cur, slots + lo, pend[h], ev->vmax[h], ev->prior[h], ev->n_act);
                    pend[h] = -1;
                }
                if (prof) { t_wait += t1 - t0; t_backup += now_s() - t1; }
            }
            if (it == iterations) continue;
            double t0 = prof ? now_s() : 0.0;
            /* the block's pending backup rides at the head of this expand's parallel region */
            const int64_t n_new = expand_impl(f, env, maxd, c_base, c_init, seed, iteration0 + (uint64_t)it, lo, hi,
                                              noise_sigma, cur, slots + lo, ev->stage[h], cap, treedepth, pend[h],
                                              ev->vmax[h], ev->prior[h], ev->n_act);
            pend[h] = -1;
            if (n_new < 0) { rc = n_new; break; }
            double t1 = prof ? now_s() : 0.0;
            if (n_new > 0) {
                const float* in = ev->stage[h];
                int e = 0;
                if (ev->dev_in[h]) {
                    in = ev->dev_in[h];
                    e = copy(ev->dev_in[h], ev->stage[h], sizeof(float) * (size_t)(cap * f->row_w), 1, ev->stream[h]);
                }
                if (!e)
                    e = forward(ev->net, ev->params, ev->prep, in, in + cap * P, in + cap * (P + L6), n_new, ev->q[h],
                                ev->vmax[h], ev->argmax[h], ev->prior[h], ev->ws[h], ev->ws_bytes, ev->stream[h], NULL);
                if (!e) e = write32 ? write32(ev->stream[h], ev->done_flag[h], ++seq[h], 0u) : ev_record(ev->event[h], ev->stream[h]);
                if (e) { rc = -1000 - (e < 0 ? -e : e); break; }
            }
            pend[h] = n_new;
            if (prof) { t_expand += t1 - t0; t_launch += now_s() - t1; }
        }
    }
    if (rc != 0)                                            /* leave nothing in flight behind an error */
        for (int64_t h = 0; h < nh; ++h) {
            if (pend[h] <= 0) continue;
            /* a real drain where the runtime offers one (an event behind everything queued on the block's stream): a flag
               that timed out once would only time out again */
            if (ev_record && ev_sync && ev->event[h] && (!write32 || rc <= -1000)) {
                if (write32) ev_record(ev->event[h], ev->stream[h]);
                ev_sync(ev->event[h]);
            } else if (write32) {
                wait_flag(ev->done_flag[h], seq[h]);
            }
        }
    if (prof) { prof[0] += t_wait; prof[1] += t_backup; prof[2] += t_expand; prof[3] += t_launch; }
    return rc;
}
