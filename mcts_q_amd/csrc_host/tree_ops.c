/* Host-side inner loops of the lock-step tree search (mcts_q_amd/treesearch.py): PUCT selection and value
 * backup over K independent trees stored as flat arrays.  The serial tree logic stays on the host by design
 * (north star); this file only takes the two per-iteration loops out of numpy and spreads the trees over the
 * host cores.  Arithmetic and tie-breaking are those of the numpy path, which restates
 * mctsq_config_mcts_per.py:266-308 (select) and :358-381 (backup); tests/test_treesearch_cpu.py runs both paths
 * against each other.  Built by mcts_q_amd/csrc/Makefile into mcts_q_amd/lib/libmctsq_tree.so (gcc, no GPU code). */
#define _GNU_SOURCE
#include <sched.h>
#include <omp.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/mctsq_tree.h"

#define N_ACT 5

/* Threads of every parallel loop of this library: MCTSQ_TREE_THREADS if set; else OpenMP's default, capped by the CPUs
 * this process may run on divided by the ranks sharing the node (LOCAL_WORLD_SIZE, set by torch.distributed.run).
 * The loops are short (tens of microseconds) and the teams spin between them, so more threads than cores - e.g. 8 ranks
 * x 16 threads on a 64-core share - does not merely waste time, it collapses (measured: 2 ranks x 16 threads on 16 CPUs
 * ran the search 200x slower). */
int mctsq_tree_threads(void) {
    static int n = -1;
    if (n < 0) {
        const char* e = getenv("MCTSQ_TREE_THREADS");
        int v = e ? atoi(e) : 0;
        if (v <= 0) {
            v = omp_get_max_threads();
            cpu_set_t set;
            int cpus = 0;
            if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
            const char* w = getenv("LOCAL_WORLD_SIZE");
            const int ranks = w && atoi(w) > 0 ? atoi(w) : 1;
            if (cpus > 0 && v > cpus / ranks) v = cpus / ranks;
            if (v < 1) v = 1;
        }
        n = v;
    }
    return n;
}

/* cur_out[k] = the node selection stops at for tree k: a terminal node, a node at the depth limit, or the first
 * node with an untried legal action. */
void mctsq_tree_select(int64_t K, const int64_t* root, int64_t maxd, double c_base, double c_init,
                       const uint8_t* done, const int64_t* depth, const int64_t* legal, const int64_t* tried,
                       const int64_t* child, const int64_t* visits, const double* total, const double* prior_a,
                       int64_t* cur_out) {
#pragma omp parallel for schedule(static) if (K >= 256) num_threads(mctsq_tree_threads())
    for (int64_t k = 0; k < K; ++k) {
        int64_t cur = root[k];
        for (;;) {
            if (done[cur] || depth[cur] >= maxd || tried[cur] != legal[cur]) break;
            const int64_t* ch = child + cur * N_ACT;
            int64_t tv = 0;
            for (int a = 0; a < N_ACT; ++a)
                if (ch[a] >= 0) tv += visits[ch[a]];
            const double c_puct = log((1.0 + (double)tv + c_base) / c_base) + c_init;   /* :273 */
            const double sq = sqrt((double)tv);
            double best = -INFINITY;
            int64_t pick = -1;
            for (int a = 0; a < N_ACT; ++a) {                 /* creation order, first maximum wins */
                const int64_t c = ch[a];
                if (c < 0) continue;
                const int64_t v = visits[c];
                const double vf = (double)v;
                const double mean = v > 0 ? total[c] / vf : 0.0;
                const double explore = c_puct * prior_a[c] * sq / (1.0 + vf);
                const double score = mean + explore;
                if (pick < 0 || score > best) { best = score; pick = c; }
            }
            if (pick < 0) break;                              /* no child: nothing to descend into */
            cur = pick;
        }
        cur_out[k] = cur;
    }
}

/* every node on the path leaf -> root of tree k gains one visit and the leaf's value */
void mctsq_tree_backup(int64_t K, const int64_t* leaf, const double* vmax, const int64_t* parent, int64_t* visits,
                       double* total) {
#pragma omp parallel for schedule(static) if (K >= 256) num_threads(mctsq_tree_threads())
    for (int64_t k = 0; k < K; ++k) {
        int64_t node = leaf[k];
        if (node < 0) continue;
        const double value = vmax[node];
        while (node >= 0) {
            visits[node] += 1;
            total[node] += value;
            node = parent[node];
        }
    }
}

/* The two wide per-node arrays (observation row, state vector: ~1 KB per node together) are compacted through a
 * grow-only scratch buffer in two parallel passes (gather the kept records, copy them back), since an in-place move is
 * only safe in ascending order, i.e. serial.  `old_of` (n int64 scratch) receives the kept old ids in order.  The
 * scratch buffer is shared by all forests of the process: calls must not overlap (the search driver is single-threaded). */
static char* g_wide = NULL;
static size_t g_wide_bytes = 0;

static void compact_wide(int64_t n, int64_t m, const int64_t* newid, int64_t* old_of, float* rows, int64_t row_w,
                         double* svec, int64_t D) {
    int64_t first = m;                                   /* nodes before the first moved one stay where they are */
    for (int64_t i = 0; i < n; ++i)
        if (newid[i] >= 0) {
            old_of[newid[i]] = i;
            if (newid[i] != i && newid[i] < first) first = newid[i];
        }
    const int64_t moved = m - first;
    if (moved <= 0) return;
    const size_t rb = sizeof(float) * (size_t)row_w, sb = sizeof(double) * (size_t)D, rec = rb + sb;
    if ((size_t)moved * rec > g_wide_bytes) {
        free(g_wide);
        g_wide_bytes = (size_t)moved * rec * 2;
        g_wide = (char*)malloc(g_wide_bytes);
        if (!g_wide) {                                   /* no scratch: the serial ascending move */
            g_wide_bytes = 0;
            for (int64_t j = first; j < m; ++j) {
                memmove(rows + j * row_w, rows + old_of[j] * row_w, rb);
                memmove(svec + j * D, svec + old_of[j] * D, sb);
            }
            return;
        }
    }
#pragma omp parallel if (moved >= 2048) num_threads(mctsq_tree_threads())
    {
#pragma omp for schedule(static)
        for (int64_t j = first; j < m; ++j) {
            char* t = g_wide + (size_t)(j - first) * rec;
            memcpy(t, rows + old_of[j] * row_w, rb);
            memcpy(t + rb, svec + old_of[j] * D, sb);
        }
#pragma omp for schedule(static)
        for (int64_t j = first; j < m; ++j) {
            const char* t = g_wide + (size_t)(j - first) * rec;
            memcpy(rows + j * row_w, t, rb);
            memcpy(svec + j * D, t + rb, sb);
        }
    }
}

/* Sub-tree reuse (mctsq_config_mcts_per.py:153-167, 236-247): keep only the nodes under keep_roots[k] (>= 0), compact
 * every per-node array in place in ascending id order (children are always created after their parents, so the
 * order -- and with it every id the numpy path would assign -- is preserved), re-root the kept trees at depth 0.
 * rel / newid: scratch of n int64 each.  Returns the new node count; new_roots[k] = new id of keep_roots[k] or -1. */
int64_t mctsq_tree_retain(int64_t K, int64_t n, const int64_t* keep_roots, int64_t* new_roots, int64_t* parent,
                          int64_t* action, int64_t* visits, double* total, int64_t* depth, uint8_t* done,
                          int64_t* legal, int64_t* tried, int64_t* nchild, int64_t* child, double* vmax,
                          double* prior_a, float* rows, int64_t row_w, double* svec, int64_t D, int64_t* rel,
                          int64_t* newid) {
    for (int64_t i = 0; i < n; ++i) rel[i] = -1;
    /* breadth-first marking; newid doubles as the queue until the ids are assigned */
    int64_t qh = 0, qt = 0;
    for (int64_t k = 0; k < K; ++k)
        if (keep_roots[k] >= 0) { rel[keep_roots[k]] = 0; newid[qt++] = keep_roots[k]; }
    while (qh < qt) {
        const int64_t node = newid[qh++];
        const int64_t* ch = child + node * N_ACT;
        for (int a = 0; a < N_ACT; ++a)
            if (ch[a] >= 0) { rel[ch[a]] = rel[node] + 1; newid[qt++] = ch[a]; }
    }
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) newid[i] = rel[i] >= 0 ? m++ : -1;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t j = newid[i];
        if (j < 0) continue;
        action[j] = action[i]; visits[j] = visits[i]; total[j] = total[i]; done[j] = done[i];
        legal[j] = legal[i]; tried[j] = tried[i]; nchild[j] = nchild[i]; vmax[j] = vmax[i]; prior_a[j] = prior_a[i];
        depth[j] = rel[i];
        const int64_t p = parent[i];
        parent[j] = p >= 0 ? newid[p] : -1;
        for (int a = 0; a < N_ACT; ++a) {
            const int64_t c = child[i * N_ACT + a];
            child[j * N_ACT + a] = c >= 0 ? newid[c] : -1;
        }
    }
    compact_wide(n, m, newid, rel, rows, row_w, svec, D);
    for (int64_t k = 0; k < K; ++k) {
        new_roots[k] = keep_roots[k] >= 0 ? newid[keep_roots[k]] : -1;
        if (new_roots[k] >= 0) parent[new_roots[k]] = -1;      /* detach from the previous root (:162) */
    }
    return m;
}
