// Prioritized-replay device structures (internal).
#pragma once
#include "common.hpp"

namespace mctsq {

// 64-ary sum tree stored level by level in one uint64 array: level 0 = leaves (one per physical
// ring slot), level l+1 entry g = sum of level l entries [64g, 64g+64); the last level has <= 64
// entries whose sum is the total mass.  Every level is padded with zeros to a multiple of 64.
struct PerTree {
    int nlev;
    int64_t off[8], cnt[8];
    int64_t words;
};
int per_tree_desc(int64_t cap, PerTree* T);
int per_mass_shift(int64_t cap);

struct PerGatherArgs {
    const float *states, *next_states;      // [cap][96]
    const int32_t* actions;
    const float *rewards, *dones;
    const int64_t* phys;                    // [B] physical slots
    int64_t B;
    float *sprice, *sprocess, *sgas, *nprice, *nprocess, *ngas;
    int32_t* act;
    float *rew, *done;
};

int launch_per_sample(const unsigned long long* tree, const double* prio, int64_t cap, int64_t head,
                      int64_t size, const double* u, int64_t B, double alpha, double beta, int64_t* idx,
                      int64_t* phys, float* w, double* wmax, double* ws_w64, hipStream_t st);
int launch_per_update(unsigned long long* tree, double* prio, int32_t* stamp, int64_t cap, int64_t head, const int64_t* idx,
                      const float* newp, int64_t B, double alpha, hipStream_t st);
int launch_per_push(unsigned long long* tree, double* prio, const double* maxprio, int64_t cap, int64_t head,
                    int64_t size, int64_t K, double alpha, hipStream_t st);
int launch_per_refresh_max(const double* prio, int64_t n_live, double* out, double* ws_part, int nparts_cap,
                           hipStream_t st);
int launch_per_gather(const PerGatherArgs& a, hipStream_t st);

}  // namespace mctsq
