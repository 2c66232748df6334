/* One lock-step search iteration per call (include/mctsq_tree.h): selection, untried-action choice, vectorised
 * environment step and node creation for a block of trees, and value assignment + backup after the leaf batch was
 * evaluated.  Restates the loop body of MCTSQ.predict (/root/reference/src/mctsq_config_mcts_per.py:249-254) with
 * _select (:266-308), _expand (:312-356), the value / prior reads of _evaluate (:291-296, 367-370) and _backpropagate
 * (:372-381) for many trees; per-tree arithmetic is that of tree_ops.c / treesearch.py. */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../../include/mctsq_tree.h"

#define N_ACT 5
static const int64_t LEGAL_OF_STATE[5] = {0x07, 0x07, 0x1B, 0x1B, 0x1B};     /* get_legal_actions (:524-546) */

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

static inline int64_t select_leaf(const mctsq_forest* f, int64_t cur, int64_t maxd, double c_base, double c_init) {
    for (;;) {
        if (f->done[cur] || f->depth[cur] >= maxd || f->tried[cur] != f->legal[cur]) return cur;
        const int64_t* ch = f->child + cur * N_ACT;
        int64_t tv = 0;
        for (int a = 0; a < N_ACT; ++a)
            if (ch[a] >= 0) tv += f->visits[ch[a]];
        const double c_puct = log((1.0 + (double)tv + c_base) / c_base) + c_init;
        const double sq = sqrt((double)tv);
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

int64_t mctsq_lockstep_expand(mctsq_forest* f, const void* env, int64_t maxd, double c_base, double c_init,
                              uint64_t seed, uint64_t iteration, int64_t k_lo, int64_t k_hi, double noise_sigma,
                              int64_t* cur, int64_t* slot_node, float* stage, int64_t stage_cap, int64_t* treedepth) {
    const int64_t nk = k_hi - k_lo, D = f->D, W = f->row_w, P = W - 6 - (D - MCTSQ_PTG_STATE_HEAD);
    const int64_t L6 = D - MCTSQ_PTG_STATE_HEAD;
    if (nk <= 0) return 0;
    /* one parallel region for both loops: selection, then (one thread) the slots in tree order - node ids do not depend
       on the thread count - then action choice / environment step / node creation */
    int64_t n_new = 0, bad = 0, fail = 0;
    const int64_t base = f->n;
#pragma omp parallel if (nk >= 128)
    {
#pragma omp for schedule(static)
        for (int64_t k = k_lo; k < k_hi; ++k) cur[k] = select_leaf(f, f->root[k], maxd, c_base, c_init);
#pragma omp single
        {
            for (int64_t k = k_lo; k < k_hi; ++k) {
                const int64_t c = cur[k];
                if (!(f->done[c] || f->depth[c] >= maxd)) slot_node[n_new++] = k;       /* tree index for now */
            }
            if (n_new > stage_cap || f->n + n_new > f->cap) fail = 1;
        }
        if (!fail) {
#pragma omp for schedule(static) reduction(+ : bad)
            for (int64_t s = 0; s < n_new; ++s) {
                const int64_t k = slot_node[s], par = cur[k], id = base + s;
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
                for (int c = 0; c < N_ACT; ++c) f->child[id * N_ACT + c] = -1;
                f->vmax[id] = 0.0; f->prior_a[id] = 0.0;
                f->child[par * N_ACT + f->nchild[par]] = id;       /* one expansion per tree: no two slots share a parent */
                f->nchild[par] += 1;
                f->tried[par] |= (int64_t)1 << a;
                if (f->depth[id] > treedepth[k]) treedepth[k] = f->depth[id];
                cur[k] = id;
            }
        }
    }
    if (fail) return -1 - n_new;
    for (int64_t s = 0; s < n_new; ++s) slot_node[s] = base + s;
    f->n = base + n_new;
    return bad ? -1 - bad : n_new;
}

void mctsq_lockstep_backup(mctsq_forest* f, int64_t k_lo, int64_t k_hi, const int64_t* cur, const int64_t* slot_node,
                           int64_t n_new, const float* vmax_out, const float* prior_out, int64_t n_act) {
    for (int64_t s = 0; s < n_new; ++s) {
        const int64_t id = slot_node[s];
        f->vmax[id] = (double)vmax_out[s];                                  /* q.max().item() (:370) */
        f->prior_a[id] = (double)prior_out[s * n_act + f->action[id]];      /* prior[idx, action].item() (:296) */
    }
#pragma omp parallel for schedule(static) if (k_hi - k_lo >= 256)
    for (int64_t k = k_lo; k < k_hi; ++k) {
        int64_t node = cur[k];
        const double value = f->vmax[node];
        while (node >= 0) {
            f->visits[node] += 1;
            f->total[node] += value;
            node = f->parent[node];
        }
    }
}
