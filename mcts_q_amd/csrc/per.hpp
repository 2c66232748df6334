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

typedef unsigned long long per_u64;
// sample workspace: control words first (the normalisation ticket), raw fp64 weights behind them
constexpr int64_t PER_WS_CTRL_BYTES = 64;
// ga != NULL: the sampled records are gathered by the same launch (ga->phys is ignored).
// Passengers (learner step only): extra workgroups of the same launch that do small jobs the start of a step needs and
// that depend on nothing the sampling does -- re-splitting the weight fragments the previous step left
// (prepare_core.hpp: dqn_prepare_block) and zeroing the flat gradient buffer.  Each would otherwise be a launch (and
// ~5 us of host time per runtime call) of its own.
struct DqnPrepArgs;
struct SamplePassengers {
    const DqnPrepArgs* prep;    // or NULL
    float* zero;                // or NULL: zero_floats floats (a multiple of 4, 16-byte aligned) are set to 0
    int64_t zero_floats;
};
int launch_per_sample(const per_u64* tree, const double* prio, int64_t cap, int64_t head, int64_t size,
                      const double* u, int64_t B, double alpha, double beta, int64_t mass_cap, int64_t* idx,
                      int64_t* phys, float* w, double* wmax, void* ws, const PerGatherArgs* ga, hipStream_t st,
                      const SamplePassengers* extra = nullptr);
// mctsq_per_sample_batch with the optional passenger (capi.cpp)
int per_sample_batch_impl(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head, int64_t size,
                          const double* d_u, int64_t B, double alpha, double beta, int64_t mass_capacity, int64_t* d_idx,
                          int64_t* d_phys, float* d_w, double* d_wmax, const float* d_states, const float* d_next_states,
                          const int32_t* d_actions, const float* d_rewards, const float* d_dones, float* d_sprice,
                          float* d_sprocess, float* d_sgas, float* d_nprice, float* d_nprocess, float* d_ngas,
                          int32_t* d_act, float* d_rew, float* d_done, void* d_ws, int64_t ws_bytes,
                          const SamplePassengers* extra, void* stream);
int launch_per_sample_at(const per_u64* tree, const double* prio, int64_t cap, int64_t head, int64_t size,
                         const per_u64* targets, int64_t B, double alpha, double beta, int64_t mass_cap,
                         double n_total, double mass_total, int64_t* idx, int64_t* phys, double* w64, double* wmax,
                         hipStream_t st);
// raw != NULL (data-parallel learner step): additionally fp32 raw weights per row and the (hi, lo) halves of their
// maximum in this rank's pair of `slots` [2 * world] (the other pairs are written as zeros)
struct ShardWeightOut { float* w_raw; float* slots; int world, rank; };
int launch_per_sample_shard(const per_u64* tree, const double* prio, int64_t cap, int64_t head, int64_t size,
                            const per_u64* targets, const int32_t* count, int64_t bcap, double alpha, double beta,
                            int64_t mass_cap, const per_u64* shard_mass, const int64_t* shard_size, int world,
                            int64_t* idx, int64_t* phys, double* w64, double* wmax, void* ws,
                            const PerGatherArgs* ga, hipStream_t st, const SamplePassengers* extra = nullptr,
                            const ShardWeightOut* raw = nullptr);
int launch_per_shard_select(const double* u, int64_t B, const per_u64* shard_mass, int world, int rank, int64_t bcap,
                            per_u64* targets, int32_t* draw, int32_t* out, hipStream_t st);
int launch_per_normalize(const double* w64, int64_t B, const double* wmax, float* w, hipStream_t st);
int launch_per_shard_targets(const double* u, int64_t B, const per_u64* shard_mass, int world, int rank,
                             per_u64* targets, int32_t* mine, hipStream_t st);
int launch_per_total_mass(const per_u64* tree, int64_t cap, int64_t size, per_u64* out, hipStream_t st);
int launch_per_update(per_u64* tree, double* prio, int32_t* stamp, int64_t cap, int64_t head, int64_t size,
                      const int64_t* idx, const float* newp, int64_t B, double alpha, int64_t mass_cap,
                      hipStream_t st);
int launch_per_push(per_u64* tree, double* prio, const double* maxprio, int64_t cap, int64_t head,
                    int64_t size, int64_t K, double alpha, int64_t mass_cap, hipStream_t st);
int launch_per_refresh_max(const double* prio, int64_t n_live, double* out, double* ws_part, int nparts_cap,
                           hipStream_t st);
int launch_per_gather(const PerGatherArgs& a, hipStream_t st);

}  // namespace mctsq
