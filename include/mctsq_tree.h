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

#ifdef __cplusplus
}
#endif
#endif
