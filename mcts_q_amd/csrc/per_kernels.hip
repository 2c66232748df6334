// Prioritized replay on the device: PrioritizedReplayBuffer (dqn_per.py:238-321) as a ring buffer
// plus a 64-ary integer sum-tree (one wavefront scans one tree node).
//
// Compiled with -ffp-contract=off: mctsq_pow_det() must execute exactly the IEEE operations the
// numpy oracle (oracle/per_numpy.py: pow_det) executes, in the same order, so that probability
// masses - and with them every sampled index - are bit-identical on CPU and GPU.
#include "device_utils.hpp"
#include "per.hpp"

#pragma clang fp contract(off)

namespace mctsq {

typedef unsigned long long u64;

// ------------------------------------------------------------------ deterministic pow
__device__ __forceinline__ double mctsq_pow_det(double x, double a) {
    if (a == 0.0) return 1.0;
    if (x == 0.0) return a > 0.0 ? 0.0 : INFINITY;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double INV_LN2 = 1.44269504088896338700e+00, SQRT_HALF = 0.70710678118654752440;
    int e;
    double m = frexp(x, &e);
    if (m < SQRT_HALF) { m = m * 2.0; e -= 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double poly = 1.0 / 23.0;
    poly = poly * s2 + 1.0 / 21.0; poly = poly * s2 + 1.0 / 19.0; poly = poly * s2 + 1.0 / 17.0;
    poly = poly * s2 + 1.0 / 15.0; poly = poly * s2 + 1.0 / 13.0; poly = poly * s2 + 1.0 / 11.0;
    poly = poly * s2 + 1.0 / 9.0;  poly = poly * s2 + 1.0 / 7.0;  poly = poly * s2 + 1.0 / 5.0;
    poly = poly * s2 + 1.0 / 3.0;  poly = poly * s2 + 1.0;
    const double lnm = (2.0 * s) * poly;
    const double ed = (double)e;
    const double L = ed * LN2_HI + (ed * LN2_LO + lnm);
    const double y = a * L;
    double n = rint(y * INV_LN2);
    const double r = (y - n * LN2_HI) - n * LN2_LO;
    double ex = 1.0 / 6227020800.0;                       // 1/13!
    ex = ex * r + 1.0 / 479001600.0; ex = ex * r + 1.0 / 39916800.0; ex = ex * r + 1.0 / 3628800.0;
    ex = ex * r + 1.0 / 362880.0;    ex = ex * r + 1.0 / 40320.0;    ex = ex * r + 1.0 / 5040.0;
    ex = ex * r + 1.0 / 720.0;       ex = ex * r + 1.0 / 120.0;      ex = ex * r + 1.0 / 24.0;
    ex = ex * r + 1.0 / 6.0;         ex = ex * r + 1.0 / 2.0;        ex = ex * r + 1.0;
    ex = ex * r + 1.0;
    n = fmin(fmax(n, -2000.0), 2000.0);
    return ldexp(ex, (int)n);
}

__device__ __forceinline__ u64 mctsq_mass(double prio, double alpha, int shift) {
    const double p = mctsq_pow_det(prio, alpha);
    double m = rint(ldexp(fmin(p, 256.0), shift));
    if (p > 0.0 && m < 1.0) m = 1.0;
    return (u64)m;
}

// ------------------------------------------------------------------ wave-wide u64 helpers
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_up_u64(u64 v, int d) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_up(lo, d, 64); hi = __shfl_up(hi, d, 64);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 wave_incl_scan(u64 v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 o = shfl_up_u64(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// mass of physical leaves [0, n) : partial group sums level by level
__device__ u64 prefix_mass(const u64* tree, const PerTree& T, int64_t n, int lane) {
    u64 acc = 0;
    int64_t idx = n;
    for (int l = 0; l < T.nlev; ++l) {
        const int64_t g = idx >> 6;
        const int within = (int)(idx & 63);
        u64 v = 0;
        if (lane < within) v = tree[T.off[l] + g * 64 + lane];
        acc += v;
        idx = g;
        if (l == T.nlev - 1) break;
    }
    // reduce lanes
    const u64 total = wave_incl_scan(acc, lane);
    return shfl_u64(total, 63);
}

// ------------------------------------------------------------------ sample
__global__ __launch_bounds__(256) void per_sample_kernel(const u64* __restrict__ tree, PerTree T,
                                                         const double* __restrict__ prio, int64_t cap,
                                                         int64_t head, int64_t size,
                                                         const double* __restrict__ u,
                                                         const u64* __restrict__ targets, int64_t B,
                                                         double alpha, double beta, int shift, double n_eff,
                                                         double mass_eff,
                                                         int64_t* __restrict__ idx_out,
                                                         int64_t* __restrict__ phys_out,
                                                         double* __restrict__ w64) {
    const int lane = threadIdx.x & 63;
    const int64_t draw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (draw >= B) return;                                   // wave-uniform
    const int top = T.nlev - 1;
    u64 v = lane < T.cnt[top] ? tree[T.off[top] + lane] : 0;
    u64 sc = wave_incl_scan(v, lane);
    const u64 M = shfl_u64(sc, 63);
    // u = k / 2^53 exactly (numpy random_sample); target t = floor(k * M / 2^53)
    u64 t;
    if (targets) {
        t = targets[draw];                                   // sharded: local integer target
    } else {
        const u64 k = (u64)(u[draw] * 9007199254740992.0);
        const u64 hi = __umul64hi(k, M), lo = k * M;
        t = (hi << 11) | (lo >> 53);
    }
    if (head != 0) {                                         // logical order starts at `head`
        const u64 H = prefix_mass(tree, T, head, lane);
        const u64 tail = M - H;
        t = t < tail ? t + H : t - tail;
    }
    int64_t g = 0;
    for (int l = top; l >= 0; --l) {
        if (l != top) {
            v = tree[T.off[l] + g * 64 + lane];
            sc = wave_incl_scan(v, lane);
        }
        const unsigned long long ball = __ballot(sc > t);
        int c = ball ? __ffsll((long long)ball) - 1 : 63;
        const u64 excl = shfl_u64(sc - v, c);
        t -= excl;
        g = g * 64 + c;
    }
    if (g >= cap) g = cap - 1;                               // unreachable for consistent trees
    if (lane == 0) {
        int64_t logical = g - head;
        if (logical < 0) logical += cap;
        idx_out[draw] = logical;
        phys_out[draw] = g;
        const double p = mctsq_pow_det(prio[g], alpha);
        const double ratio = ((n_eff > 0.0 ? n_eff : (double)size) * p) * ldexp(1.0, shift) /
                             (mass_eff > 0.0 ? mass_eff : (double)M);
        w64[draw] = mctsq_pow_det(ratio, -beta);
    }
}

// global targets of a sharded buffer: shard `rank` owns [prefix, prefix + mass[rank]) of the
// concatenated mass line; t = floor(k * M / 2^53) exactly as the single-buffer form
__global__ __launch_bounds__(256) void per_shard_targets_kernel(const double* __restrict__ u, int64_t B,
                                                                const u64* __restrict__ shard_mass, int world,
                                                                int rank, u64* __restrict__ targets,
                                                                int32_t* __restrict__ mine) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    u64 M = 0, prefix = 0;
    for (int r = 0; r < world; ++r) {
        if (r < rank) prefix += shard_mass[r];
        M += shard_mass[r];
    }
    const u64 k = (u64)(u[i] * 9007199254740992.0);
    const u64 hi = __umul64hi(k, M), lo = k * M;
    const u64 t = (hi << 11) | (lo >> 53);
    const bool m = t >= prefix && t - prefix < shard_mass[rank];
    targets[i] = m ? t - prefix : 0;
    mine[i] = m ? 1 : 0;
}

__global__ __launch_bounds__(64) void per_total_mass_kernel(const u64* __restrict__ tree, PerTree T,
                                                            u64* __restrict__ out) {
    const int lane = threadIdx.x, top = T.nlev - 1;
    u64 v = lane < T.cnt[top] ? tree[T.off[top] + lane] : 0;
    v = wave_incl_scan(v, lane);
    if (lane == 63) *out = v;
}

__global__ __launch_bounds__(1024) void per_weight_max_kernel(const double* __restrict__ w64, int64_t B,
                                                              double* __restrict__ wmax) {
    __shared__ double red[16];
    double mx = 0.0;
    for (int64_t i = threadIdx.x; i < B; i += 1024) mx = fmax(mx, w64[i]);
    for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = red[0];
        for (int i = 1; i < 16; ++i) mx = fmax(mx, red[i]);
        *wmax = mx;
    }
}

__global__ __launch_bounds__(256) void per_weight_scale_kernel(const double* __restrict__ w64, int64_t B,
                                                               const double* __restrict__ wmax,
                                                               float* __restrict__ w) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < B) w[i] = (float)(w64[i] / *wmax);
}

__global__ __launch_bounds__(1024) void per_weight_norm_kernel(const double* __restrict__ w64, int64_t B,
                                                               float* __restrict__ w, double* __restrict__ wmax) {
    __shared__ double red[16];
    double mx = 0.0;
    for (int64_t i = threadIdx.x; i < B; i += 1024) mx = fmax(mx, w64[i]);
    for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int i = 1; i < 16; ++i) mx = fmax(mx, red[i]);
    for (int64_t i = threadIdx.x; i < B; i += 1024) w[i] = (float)(w64[i] / mx);
    if (wmax && threadIdx.x == 0) *wmax = mx;
}

// ------------------------------------------------------------------ update / push
__device__ __forceinline__ void set_leaf(u64* tree, const PerTree& T, int64_t phys, u64 m) {
    const u64 old = tree[T.off[0] + phys];
    tree[T.off[0] + phys] = m;
    const u64 delta = m - old;                               // wraps for decreases; sums stay exact
    if (delta == 0) return;
    int64_t idx = phys;
    for (int l = 1; l < T.nlev; ++l) {
        idx >>= 6;
        atomicAdd(&tree[T.off[l] + idx], delta);
    }
}

// Sequential semantics (dqn_per.py:313-314): for duplicate indices the LAST occurrence wins.
// Pass 1 records, per physical slot, the largest batch position that names it (atomicMax on an
// int32 stamp array that is all -1 between calls); pass 2 lets exactly that position write and
// resets the stamp.  O(B), any B.
__global__ __launch_bounds__(256) void per_update_mark_kernel(int32_t* __restrict__ stamp, int64_t cap,
                                                              int64_t head, const int64_t* __restrict__ idx,
                                                              int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const int64_t li = idx[i];
    if (li < 0 || li >= cap) return;
    int64_t phys = head + li;
    if (phys >= cap) phys -= cap;
    atomicMax(&stamp[phys], (int32_t)i);
}

__global__ __launch_bounds__(256) void per_update_kernel(u64* __restrict__ tree, PerTree T,
                                                         double* __restrict__ prio, int32_t* __restrict__ stamp,
                                                         int64_t cap, int64_t head,
                                                         const int64_t* __restrict__ idx,
                                                         const float* __restrict__ newp, int64_t B,
                                                         double alpha, int shift) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const int64_t li = idx[i];
    if (li < 0 || li >= cap) return;
    int64_t phys = head + li;
    if (phys >= cap) phys -= cap;
    if (stamp[phys] != (int32_t)i) return;
    stamp[phys] = -1;
    const double p = (double)newp[i];
    prio[phys] = p;
    set_leaf(tree, T, phys, mctsq_mass(p, alpha, shift));
}

__global__ __launch_bounds__(256) void per_push_kernel(u64* __restrict__ tree, PerTree T,
                                                       double* __restrict__ prio, int64_t cap,
                                                       int64_t first_slot, int64_t K,
                                                       const double* __restrict__ maxprio, double alpha,
                                                       int shift) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    const int64_t phys = (first_slot + i) % cap;
    const double p = *maxprio;
    prio[phys] = p;
    set_leaf(tree, T, phys, mctsq_mass(p, alpha, shift));
}

// ------------------------------------------------------------------ max over live priorities
__global__ __launch_bounds__(256) void per_max_partial_kernel(const double* __restrict__ prio, int64_t n,
                                                              double* __restrict__ part) {
    __shared__ double red[4];
    double mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        mx = fmax(mx, prio[i]);
    for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__global__ __launch_bounds__(256) void per_max_final_kernel(const double* __restrict__ part, int nparts,
                                                            double* __restrict__ out) {
    __shared__ double red[4];
    double mx = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) mx = fmax(mx, part[i]);
    for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        *out = nparts == 0 || mx == -INFINITY ? 1.0 : mx;    // max(deque, default=1.0)
    }
}

// ------------------------------------------------------------------ gather
__global__ __launch_bounds__(256) void per_gather_kernel(PerGatherArgs a) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t b = gid / 192;                 // 2 x 96-float records per sampled transition
    if (b >= a.B) return;
    const int c = (int)(gid - b * 192);
    const int which = c >= 96;                    // 0 = state, 1 = next state
    const int col = which ? c - 96 : c;
    const int64_t phys = a.phys[b];
    if (col < 91) {
        const float v = (which ? a.next_states : a.states)[phys * MCTSQ_STATE_ROW + col];
        float* dst;
        if (col < 25) dst = (which ? a.nprice : a.sprice) + b * 25 + col;
        else if (col < 85) dst = (which ? a.nprocess : a.sprocess) + b * 60 + (col - 25);
        else dst = (which ? a.ngas : a.sgas) + b * 6 + (col - 85);
        *dst = v;
    } else if (col == 91 && !which) {
        a.act[b] = a.actions[phys];
        a.rew[b] = a.rewards[phys];
        a.done[b] = a.dones[phys];
    }
}

// ------------------------------------------------------------------ launchers
int per_tree_desc(int64_t cap, PerTree* T) {
    MCTSQ_CHECK(cap >= 1 && cap <= (int64_t)1 << 36, "capacity %lld out of range", (long long)cap);
    int64_t cnt = cap, off = 0;
    int l = 0;
    for (;; ++l) {
        MCTSQ_CHECK(l < 8, "tree too deep");
        T->cnt[l] = cnt;
        T->off[l] = off;
        off += round_up(cnt, 64);
        if (cnt <= 64) break;
        cnt = (cnt + 63) / 64;
    }
    T->nlev = l + 1;
    T->words = off;
    return 0;
}

int per_mass_shift(int64_t cap) {
    int bits = 1;
    while (((int64_t)1 << bits) < cap) ++bits;       // == max(1, (cap-1).bit_length())
    int s = 62 - bits - 8;
    return s > 44 ? 44 : s;
}

int launch_per_sample(const u64* tree, const double* prio, int64_t cap, int64_t head, int64_t size,
                      const double* u, int64_t B, double alpha, double beta, int64_t mass_cap, int64_t* idx,
                      int64_t* phys, float* w, double* wmax, double* ws_w64, hipStream_t st) {
    PerTree T;
    if (per_tree_desc(cap, &T)) return -1;
    const int shift = per_mass_shift(mass_cap);
    hipLaunchKernelGGL(per_sample_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, tree, T, prio, cap,
                       head, size, u, (const u64*)nullptr, B, alpha, beta, shift, 0.0, 0.0, idx, phys, ws_w64);
    hipLaunchKernelGGL(per_weight_norm_kernel, dim3(1), dim3(1024), 0, st, ws_w64, B, w, wmax);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_sample_at(const u64* tree, const double* prio, int64_t cap, int64_t head, int64_t size,
                         const u64* targets, int64_t B, double alpha, double beta, int64_t mass_cap, double n_total,
                         double mass_total, int64_t* idx, int64_t* phys, double* w64, double* wmax,
                         hipStream_t st) {
    PerTree T;
    if (per_tree_desc(cap, &T)) return -1;
    hipLaunchKernelGGL(per_sample_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, tree, T, prio, cap,
                       head, size, (const double*)nullptr, targets, B, alpha, beta, per_mass_shift(mass_cap),
                       n_total, mass_total, idx, phys, w64);
    if (wmax) hipLaunchKernelGGL(per_weight_max_kernel, dim3(1), dim3(1024), 0, st, w64, B, wmax);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_normalize(const double* w64, int64_t B, const double* wmax, float* w, hipStream_t st) {
    hipLaunchKernelGGL(per_weight_scale_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, w64, B, wmax, w);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_shard_targets(const double* u, int64_t B, const u64* shard_mass, int world, int rank, u64* targets,
                             int32_t* mine, hipStream_t st) {
    hipLaunchKernelGGL(per_shard_targets_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, u, B,
                       shard_mass, world, rank, targets, mine);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_total_mass(const u64* tree, int64_t cap, u64* out, hipStream_t st) {
    PerTree T;
    if (per_tree_desc(cap, &T)) return -1;
    hipLaunchKernelGGL(per_total_mass_kernel, dim3(1), dim3(64), 0, st, tree, T, out);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_update(u64* tree, double* prio, int32_t* stamp, int64_t cap, int64_t head, const int64_t* idx,
                      const float* newp, int64_t B, double alpha, int64_t mass_cap, hipStream_t st) {
    PerTree T;
    if (per_tree_desc(cap, &T)) return -1;
    const dim3 grid((unsigned)((B + 255) / 256));
    hipLaunchKernelGGL(per_update_mark_kernel, grid, dim3(256), 0, st, stamp, cap, head, idx, B);
    hipLaunchKernelGGL(per_update_kernel, grid, dim3(256), 0, st, tree, T, prio, stamp, cap, head, idx, newp, B,
                       alpha, per_mass_shift(mass_cap));
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_push(u64* tree, double* prio, const double* maxprio, int64_t cap, int64_t head, int64_t size,
                    int64_t K, double alpha, int64_t mass_cap, hipStream_t st) {
    PerTree T;
    if (per_tree_desc(cap, &T)) return -1;
    const int64_t first = (head + size) % cap;
    hipLaunchKernelGGL(per_push_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, tree, T, prio, cap,
                       first, K, maxprio, alpha, per_mass_shift(mass_cap));
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_refresh_max(const double* prio, int64_t n_live, double* out, double* ws_part, int nparts_cap,
                           hipStream_t st) {
    int nparts = (int)((n_live + 4095) / 4096);
    if (nparts > nparts_cap) nparts = nparts_cap;
    if (nparts > 0)
        hipLaunchKernelGGL(per_max_partial_kernel, dim3(nparts), dim3(256), 0, st, prio, n_live, ws_part);
    hipLaunchKernelGGL(per_max_final_kernel, dim3(1), dim3(256), 0, st, ws_part, nparts, out);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_gather(const PerGatherArgs& a, hipStream_t st) {
    if (a.B == 0) return 0;
    hipLaunchKernelGGL(per_gather_kernel, dim3((unsigned)((a.B * 192 + 255) / 256)), dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
