// Prioritized replay on the device: PrioritizedReplayBuffer (dqn_per.py:238-321) as a ring buffer
// plus a 64-ary integer sum-tree (one wavefront scans one tree node).
//
// Compiled with -ffp-contract=off: mctsq_pow_det() must execute exactly the IEEE operations the
// numpy oracle (oracle/per_numpy.py: pow_det) executes, in the same order, so that probability
// masses - and with them every sampled index - are bit-identical on CPU and GPU.
#include "device_utils.hpp"
#include "per.hpp"
#include "prepare_core.hpp"

#include <string.h>

#include <mutex>

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
// value of lane `src` (wave-uniform) in every lane: two scalar lane reads, no LDS crossbar trip
__device__ __forceinline__ u64 readlane_u64(u64 v, int src) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}
// DPP move of both halves; lanes the pattern does not reach (row start, rows masked out) read 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, true);
    return ((u64)hi << 32) | lo;
}
// inclusive prefix sum over the 64 lanes (all active): shifts inside the rows of 16 (row_shr:1,2,4,8), then lane 15 of
// rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2, 3 (row_bcast:31).  ~40 vector instructions;
// the shuffle form (12 LDS-crossbar permutes in a dependent chain) cost ~0.5 us per tree level of every draw.
__device__ __forceinline__ u64 wave_incl_scan(u64 v, int) {
    v += dpp_u64<0x111, 0xf>(v);
    v += dpp_u64<0x112, 0xf>(v);
    v += dpp_u64<0x114, 0xf>(v);
    v += dpp_u64<0x118, 0xf>(v);
    v += dpp_u64<0x142, 0xa>(v);
    v += dpp_u64<0x143, 0xc>(v);
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
    return readlane_u64(total, 63);
}

// ------------------------------------------------------------------ sample
// One workgroup = SMP_DRAWS draws, one per wave (the descent is a chain of dependent node reads and 64-wide integer
// prefix scans: latency, so draws run side by side on as many waves as possible rather than in turn on few).  The
// two top levels of the tree (<= 64 + 4096 nodes) are staged in LDS once per workgroup; below them every level
// costs one 512-byte node read per draw.
// Optional tails, all in the same launch:
//   - gather (ga.phys == this kernel's phys_out): the wave that found a slot copies its 2 x 96-float record into
//     the six window tensors (per_gather_kernel's job);
//   - IS-weight normalisation (a.w != NULL): the workgroup whose ticket shows it is the last one to finish divides
//     all B raw weights by their maximum.  Raw weights travel as write-through (sc1) stores and are read back
//     with sc1 loads behind the ticket, so no cache maintenance is needed (MI355X_MICROARCH.md, hand-off table).
constexpr int SMP_DRAWS = 16, SMP_PER_WAVE = 1, SMP_WAVES = SMP_DRAWS / SMP_PER_WAVE, SMP_THREADS = 64 * SMP_WAVES;
constexpr int SMP_L2MAX = 4096;
constexpr int64_t SMP_FUSED_NORM_MAX = 16384;      // beyond this many draws the normalisation is its own launch

struct PerSampleArgs {
    const u64* tree; PerTree T; const double* prio; int64_t cap, head, size;
    const double* u;              // uniforms (single buffer) or NULL
    const u64* targets;           // explicit integer targets (sharded) or NULL
    const int32_t* count;         // device-side number of live draws (rows >= *count are padding) or NULL
    int64_t B;
    double alpha, beta; int shift;
    double n_eff, mass_eff;       // totals of the whole (sharded) buffer when known on the host, else 0
    const u64* shard_mass; const int64_t* shard_size; int world;   // ... or on the device
    int64_t *idx_out, *phys_out; double* w64;
    float* w; double* wmax; unsigned* ticket;     // fused normalisation (w may be NULL)
    float* w_raw;                 // data-parallel step: fp32 RAW weights (the division by the global maximum happens
    float* wmax_slots;            // behind the gradient all-reduce); [2 * slot_world] fp32: (hi, lo) halves of this
    int slot_world, slot_rank;    // rank's largest raw weight in its own pair of slots, zeros in the others
    PerGatherArgs ga;             // fused gather when ga.states != NULL
    int nsample;                  // workgroups that sample; behind them the passengers (per.hpp): nprep workgroups
    int nprep;                    // re-split weights (prep), the rest zero `zero`
    DqnPrepArgs prep;
    float* zero; int64_t zero_n4;
};

__device__ __forceinline__ void gather_record(const PerGatherArgs& a, int64_t b, int64_t phys, int lane) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const float* src = (which ? a.next_states : a.states) + phys * MCTSQ_STATE_ROW;
        float* pp = which ? a.nprice : a.sprice; float* pc = which ? a.nprocess : a.sprocess; float* pg = which ? a.ngas : a.sgas;
        const float v0 = src[lane];
        const float v1 = lane < 27 ? src[64 + lane] : 0.0f;
        if (lane < 25) pp[b * 25 + lane] = v0; else pc[b * 60 + (lane - 25)] = v0;
        if (lane < 21) pc[b * 60 + 39 + lane] = v1; else if (lane < 27) pg[b * 6 + (lane - 21)] = v1;
    }
    if (lane == 0) { a.act[b] = a.actions[phys]; a.rew[b] = a.rewards[phys]; a.done[b] = a.dones[phys]; }
}

__global__ __launch_bounds__(SMP_THREADS) void per_sample_kernel(PerSampleArgs a) {
    __shared__ u64 s_l2[SMP_L2MAX];
    __shared__ u64 s_head;
    __shared__ double s_red[SMP_WAVES];
    __shared__ unsigned s_last;
    __shared__ double s_pr[SMP_WAVES];
    __shared__ int64_t s_g[SMP_WAVES];
    if ((int)blockIdx.x >= a.nsample) {                      // passenger workgroups
        const int pb = (int)blockIdx.x - a.nsample;
        if (pb < a.nprep) {                                  // four 256-thread prepare blocks each
            dqn_prepare_block(a.prep, pb * (SMP_THREADS / 256) + (int)(threadIdx.x >> 8), (int)(threadIdx.x & 255));
        } else {
            const int64_t i = (int64_t)(pb - a.nprep) * SMP_THREADS + threadIdx.x;
            if (i < a.zero_n4) reinterpret_cast<float4*>(a.zero)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const PerTree& T = a.T;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int top = T.nlev - 1;
    const bool l2_staged = top >= 1 && T.cnt[top - 1] <= SMP_L2MAX;
    // the draw's uniform / target starts its trip together with the tree's top levels
    const int64_t d0 = (int64_t)blockIdx.x * SMP_DRAWS + wv * SMP_PER_WAVE;
    double u_in[SMP_PER_WAVE];
    u64 t_in[SMP_PER_WAVE];
#pragma unroll
    for (int k = 0; k < SMP_PER_WAVE; ++k) {
        const int64_t d = d0 + k;
        u_in[k] = d < a.B && !a.targets ? a.u[d] : 0.0;
        t_in[k] = d < a.B && a.targets ? a.targets[d] : 0;
    }
    if (l2_staged)
        for (int i = tid; i < (int)T.cnt[top - 1]; i += SMP_THREADS) s_l2[i] = a.tree[T.off[top - 1] + i];
    if (a.head != 0 && wv == 0) {                            // logical order starts at `head`
        const u64 H = prefix_mass(a.tree, T, a.head, lane);
        if (lane == 0) s_head = H;
    }
    const int64_t nlive = a.count ? (int64_t)*a.count : a.B;
    double n_eff = a.n_eff, mass_eff = a.mass_eff;
    if (a.shard_mass) {                                      // totals of the concatenated shards, read on the device
        u64 ms = 0; int64_t ns = 0;
        for (int r = 0; r < a.world; ++r) { ms += a.shard_mass[r]; ns += a.shard_size[r]; }
        n_eff = (double)ns; mass_eff = (double)ms;
    }
    u64 v = lane < T.cnt[top] ? a.tree[T.off[top] + lane] : 0;
    const u64 sc_top = wave_incl_scan(v, lane);
    const u64 v_top = v;
    const u64 M = readlane_u64(sc_top, 63);
    __syncthreads();
    const u64 H = a.head != 0 ? s_head : 0;

    u64 t[SMP_PER_WAVE];
    int64_t g[SMP_PER_WAVE];
    bool live[SMP_PER_WAVE];
#pragma unroll
    for (int k = 0; k < SMP_PER_WAVE; ++k) {
        const int64_t d = d0 + k;
        live[k] = d < a.B && d < nlive;                      // wave-uniform
        u64 tt = 0;
        if (live[k]) {
            if (a.targets) {
                tt = t_in[k];                                // sharded: local integer target
            } else {
                // u = k / 2^53 exactly (numpy random_sample); target t = floor(k * M / 2^53)
                const u64 kk = (u64)(u_in[k] * 9007199254740992.0);
                const u64 hi = __umul64hi(kk, M), lo = kk * M;
                tt = (hi << 11) | (lo >> 53);
            }
            if (a.head != 0) { const u64 tail = M - H; tt = tt < tail ? tt + H : tt - tail; }
        }
        const unsigned long long ball = __ballot(sc_top > tt);
        const int c = ball ? __ffsll((long long)ball) - 1 : 63;
        t[k] = tt - readlane_u64(sc_top - v_top, c);
        g[k] = c;
    }
    for (int l = top - 1; l >= 0; --l) {
        u64 vv[SMP_PER_WAVE];
        const bool from_lds = l == top - 1 && l2_staged;
#pragma unroll
        for (int k = 0; k < SMP_PER_WAVE; ++k) {
            const int64_t node = g[k] * 64 + lane;
            vv[k] = node < T.cnt[l] ? (from_lds ? s_l2[node] : a.tree[T.off[l] + node]) : 0;
        }
#pragma unroll
        for (int k = 0; k < SMP_PER_WAVE; ++k) {
            const u64 sc = wave_incl_scan(vv[k], lane);
            const unsigned long long ball = __ballot(sc > t[k]);
            const int c = ball ? __ffsll((long long)ball) - 1 : 63;
            t[k] -= readlane_u64(sc - vv[k], c);
            g[k] = g[k] * 64 + c;
        }
    }
    // ---- tails.  The draw's priority and its two records start their trips together; the fp64 weight arithmetic
    // (two software pows: ~500 dependent instructions) of the workgroup's 16 draws runs on 16 lanes of wave 0 while
    // the records are in flight -- done by lane 0 of every wave it occupied each of the four SIMDs four times over
    // (4 us of the kernel, measured) -- and the record stores follow it.
    static_assert(SMP_PER_WAVE == 1, "one draw per wave");
    const bool lv = live[0];
    const int64_t gm = g[0] < a.cap ? g[0] : a.cap - 1;      // clamp unreachable for consistent trees
    double prv = 0.0;
    if (lane == 0 && lv) prv = a.prio[gm];                    // first in the queue: wave 0 waits for these
    const bool gath = a.ga.states != nullptr && d0 < a.B;     // wave-uniform
    const int64_t phys = lv ? gm : 0;
    float r0[2] = {0.f, 0.f}, r1[2] = {0.f, 0.f}, rrew = 0.f, rdone = 0.f;
    int32_t ract = 0;
    if (gath) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const float* src = (which ? a.ga.next_states : a.ga.states) + phys * MCTSQ_STATE_ROW;
            r0[which] = src[lane];
            r1[which] = lane < 27 ? src[64 + lane] : 0.0f;
        }
        if (lane == 0) { ract = a.ga.actions[phys]; rrew = a.ga.rewards[phys]; rdone = a.ga.dones[phys]; }
    }
    if (lane == 0) {
        s_pr[wv] = prv;
        s_g[wv] = lv ? gm : -1;
    }
    __syncthreads();
    if (wv == 0 && lane < SMP_DRAWS) {
        const int64_t d = (int64_t)blockIdx.x * SMP_DRAWS + lane;
        if (d < a.B) {
            const int64_t gd = s_g[lane];
            double wraw = 0.0;
            if (gd >= 0) {
                int64_t logical = gd - a.head;
                if (logical < 0) logical += a.cap;
                a.idx_out[d] = logical;
                a.phys_out[d] = gd;
                // the mass the sampler used (clamped like mctsq_mass), so weights and sampling law agree
                const double p = fmin(mctsq_pow_det(s_pr[lane], a.alpha), 256.0);
                const double ratio = ((n_eff > 0.0 ? n_eff : (double)a.size) * p) * ldexp(1.0, a.shift) /
                                     (mass_eff > 0.0 ? mass_eff : (double)M);
                wraw = mctsq_pow_det(ratio, -a.beta);
            } else {                                         // padding row: neutral everywhere downstream
                a.idx_out[d] = -1;
                a.phys_out[d] = 0;
            }
            __hip_atomic_store(&a.w64[d], wraw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.w_raw) a.w_raw[d] = (float)wraw;
        }
    }
    if (gath) {
        const int64_t b = d0;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float* pp = which ? a.ga.nprice : a.ga.sprice; float* pc = which ? a.ga.nprocess : a.ga.sprocess;
            float* pg = which ? a.ga.ngas : a.ga.sgas;
            if (lane < 25) pp[b * 25 + lane] = r0[which]; else pc[b * 60 + (lane - 25)] = r0[which];
            if (lane < 21) pc[b * 60 + 39 + lane] = r1[which]; else if (lane < 27) pg[b * 6 + (lane - 21)] = r1[which];
        }
        if (lane == 0) { a.ga.act[b] = ract; a.ga.rew[b] = rrew; a.ga.done[b] = rdone; }
    }
    if (!a.w && !a.wmax) return;
    // ---- the ticket: wave 0 wrote the workgroup's raw weights, so its own wait orders them before its signal ----
    // (release on the ticket: the wave's raw-weight stores above are ordered before it by the memory model, not only by
    // the write-through stores having been acknowledged; acquire in the workgroup that finds itself last)
    if (wv == 0) {
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == (unsigned)a.nsample - 1 ? 1u : 0u;
        }
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double mx = 0.0;
    for (int64_t i = tid; i < a.B; i += SMP_THREADS)
        mx = fmax(mx, __hip_atomic_load(&a.w64[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int dd = 32; dd >= 1; dd >>= 1) mx = fmax(mx, __shfl_xor(mx, dd, 64));
    if (lane == 0) s_red[wv] = mx;
    __syncthreads();
    mx = s_red[0];
#pragma unroll
    for (int i = 1; i < SMP_WAVES; ++i) mx = fmax(mx, s_red[i]);
    if (a.w)
        for (int64_t i = tid; i < a.B; i += SMP_THREADS)
            a.w[i] = mx > 0.0 ? (float)(__hip_atomic_load(&a.w64[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / mx) : 0.0f;
    if (a.wmax_slots && tid < 2 * a.slot_world) {
        const float hi = (float)mx, lo = (float)(mx - (double)hi);
        a.wmax_slots[tid] = (tid >> 1) == a.slot_rank ? ((tid & 1) ? lo : hi) : 0.0f;
    }
    if (tid == 0) {
        if (a.wmax) *a.wmax = mx;
        *a.ticket = 0;                                       // ready for the next launch on this stream
    }
}

// which of the B global draws fall into this rank's interval of the concatenated mass line, compacted to the
// front of d_targets (draw order kept) with their positions; out[0] = count, out[1] = draws dropped because
// they did not fit `bcap`.  One workgroup; t = floor(k * M / 2^53) exactly as the single-buffer form.
__global__ __launch_bounds__(1024) void per_shard_select_kernel(const double* __restrict__ u, int64_t B,
                                                                const u64* __restrict__ shard_mass, int world,
                                                                int rank, int64_t bcap, u64* __restrict__ targets,
                                                                int32_t* __restrict__ draw, int32_t* __restrict__ out) {
    __shared__ int s_wsum[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    u64 M = 0, prefix = 0;
    for (int r = 0; r < world; ++r) {
        if (r < rank) prefix += shard_mass[r];
        M += shard_mass[r];
    }
    const u64 own = shard_mass[rank];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < B; i0 += 1024) {
        const int64_t i = i0 + tid;
        bool m = false; u64 t = 0;
        if (i < B) {
            const u64 k = (u64)(u[i] * 9007199254740992.0);
            const u64 hi = __umul64hi(k, M), lo = k * M;
            t = (hi << 11) | (lo >> 53);
            m = t >= prefix && t - prefix < own;
        }
        const unsigned long long ball = __ballot(m);
        const int within = __popcll(ball & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[wv] = __popcll(ball);
        __syncthreads();
        int before = s_base;
        for (int w2 = 0; w2 < wv; ++w2) before += s_wsum[w2];
        if (m) {
            const int64_t pos = before + within;
            if (pos < bcap) { targets[pos] = t - prefix; draw[pos] = (int32_t)i; }
        }
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int w2 = 0; w2 < 16; ++w2) tot += s_wsum[w2]; s_base += tot; }
        __syncthreads();
    }
    const int total = s_base;
    const int kept = total < bcap ? total : (int)bcap;
    for (int64_t i = kept + tid; i < bcap; i += 1024) { targets[i] = 0; draw[i] = -1; }
    if (tid == 0) { out[0] = kept; out[1] = total - kept; }
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

__global__ __launch_bounds__(64) void per_total_mass_kernel(const u64* __restrict__ tree, PerTree T, int64_t size,
                                                            u64* __restrict__ out) {
    const int lane = threadIdx.x, top = T.nlev - 1;
    u64 v = lane < T.cnt[top] ? tree[T.off[top] + lane] : 0;
    v = wave_incl_scan(v, lane);
    if (lane == 63) { out[0] = v; out[1] = (u64)size; }
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
// resets the stamp.  O(B), any B.  Indices outside [0, size) name no stored transition and are skipped
// (the reference raises IndexError for them; negative ones there address the deque from its end).
__global__ __launch_bounds__(256) void per_update_mark_kernel(int32_t* __restrict__ stamp, int64_t cap,
                                                              int64_t head, int64_t size,
                                                              const int64_t* __restrict__ idx, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const int64_t li = idx[i];
    if (li < 0 || li >= size) return;
    int64_t phys = head + li;
    if (phys >= cap) phys -= cap;
    atomicMax(&stamp[phys], (int32_t)i);
}

// Bulk apply (batches above UPD_FUSED_MAX: replay set-up, checkpoint import).  Every entry that won its slot writes the
// leaf and owes its mass delta to one node per upper level.  With one global atomic per level and entry, a million
// entries queue up on the few words of the two top levels (12 ms per million, measured): a workgroup therefore sums
// its entries' deltas for those two levels in LDS and issues one global add per node it touched; on the
// levels below, a wave whose lanes all name the same node (consecutive indices) adds one wave sum instead of 64.
constexpr int UPD_BULK_THREADS = 1024, UPD_BULK_PER_THREAD = 8;
__global__ __launch_bounds__(UPD_BULK_THREADS) void per_update_kernel(u64* __restrict__ tree, PerTree T,
                                                                      double* __restrict__ prio, int32_t* __restrict__ stamp,
                                                                      int64_t cap, int64_t head, int64_t size,
                                                                      const int64_t* __restrict__ idx,
                                                                      const float* __restrict__ newp, int64_t B,
                                                                      double alpha, int shift, int per_thread) {
    __shared__ u64 s_top[64], s_l2[SMP_L2MAX];
    const int tid = threadIdx.x, lane = tid & 63, top = T.nlev - 1;
    const bool lds_top = top >= 1, lds_l2 = top >= 2 && T.cnt[top - 1] <= SMP_L2MAX;
    if (tid < 64) s_top[tid] = 0;
    if (lds_l2) for (int i = tid; i < (int)T.cnt[top - 1]; i += UPD_BULK_THREADS) s_l2[i] = 0;
    __syncthreads();
    for (int k = 0; k < per_thread; ++k) {
        const int64_t i = ((int64_t)blockIdx.x * per_thread + k) * UPD_BULK_THREADS + tid;
        u64 delta = 0;
        int64_t phys = 0;
        if (i < B) {
            const int64_t li = idx[i];
            if (li >= 0 && li < size) {
                phys = head + li;
                if (phys >= cap) phys -= cap;
                if (stamp[phys] == (int32_t)i) {
                    stamp[phys] = -1;
                    const double p = (double)newp[i];
                    prio[phys] = p;
                    const u64 m = mctsq_mass(p, alpha, shift);
                    const u64 old = tree[T.off[0] + phys];
                    tree[T.off[0] + phys] = m;
                    delta = m - old;                         // wraps for decreases; sums stay exact
                }
            }
        }
        int64_t node = phys;
        for (int l = 1; l <= top; ++l) {                     // wave-uniform trip count
            node >>= 6;
            if (l == top && lds_top) { if (delta) atomicAdd(&s_top[node], delta); continue; }
            if (l == top - 1 && lds_l2) { if (delta) atomicAdd(&s_l2[node], delta); continue; }
            // one node for the whole wave (lanes without a delta go along with 0)?  then one add of the wave's sum
            const int64_t first = __shfl(node, __ffsll((long long)__ballot(delta != 0)) - 1, 64);
            const bool same = __ballot(delta != 0 && node != first) == 0;
            if (same) {
                const u64 tot = readlane_u64(wave_incl_scan(delta, lane), 63);
                if (lane == 0 && tot != 0) atomicAdd(&tree[T.off[l] + first], tot);
            } else if (delta) {
                atomicAdd(&tree[T.off[l] + node], delta);
            }
        }
    }
    __syncthreads();
    if (lds_top && tid < (int)T.cnt[top] && s_top[tid] != 0) atomicAdd(&tree[T.off[top] + tid], s_top[tid]);
    if (lds_l2)
        for (int i = tid; i < (int)T.cnt[top - 1]; i += UPD_BULK_THREADS)
            if (s_l2[i] != 0) atomicAdd(&tree[T.off[top - 1] + i], s_l2[i]);
}

// The whole update in ONE workgroup for batches up to UPD_FUSED_MAX (the training batch).  "Last duplicate wins"
// is resolved in LDS with an exact open-addressing table keyed by the physical slot (linear probing, load <= 1/4):
// every position claims / finds its slot's entry with a 64-bit compare-and-swap and raises the entry's mark with
// atomicMax(position); after the barrier the position that equals the mark is the one that writes.  The mass
// deltas of the two top tree levels (<= 64 + 4096 nodes, where every update of the batch collides) are first
// summed in LDS, so each of those nodes receives one global add per launch instead of one per update.  The
// stamp array is not used on this path.
constexpr int UPD_FUSED_MAX = 2048, UPD_HASH = 8192;
// s_key 64 KB + s_mark 32 KB + s_l2 32 KB + s_top: the 160 KB of LDS per workgroup of gfx950 (64 KB on gfx90a / gfx942)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "per_update_fused_kernel's LDS tables are sized for gfx950 (160 KB per workgroup); shrink UPD_HASH for other targets"
#endif
__global__ __launch_bounds__(1024) void per_update_fused_kernel(u64* __restrict__ tree, PerTree T,
                                                                double* __restrict__ prio, int64_t cap,
                                                                int64_t head, int64_t size,
                                                                const int64_t* __restrict__ idx,
                                                                const float* __restrict__ newp, int64_t B,
                                                                double alpha, int shift) {
    __shared__ u64 s_top[64], s_l2[SMP_L2MAX];
    __shared__ u64 s_key[UPD_HASH];
    __shared__ int s_mark[UPD_HASH];
    const int tid = threadIdx.x, top = T.nlev - 1;
    const bool lds_top = top >= 1, lds_l2 = top >= 2 && T.cnt[top - 1] <= SMP_L2MAX;
    constexpr u64 EMPTY = ~0ull;
    if (tid < 64) s_top[tid] = 0;
    if (lds_l2) for (int i = tid; i < (int)T.cnt[top - 1]; i += 1024) s_l2[i] = 0;
    for (int i = tid; i < UPD_HASH; i += 1024) { s_key[i] = EMPTY; s_mark[i] = -1; }
    __syncthreads();
    // positions tid and tid + 1024 (B <= 2048): physical slot and table entry kept in registers across the barrier
    int64_t physv[2]; int ent[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 1024 * u;
        physv[u] = -1; ent[u] = -1;
        if (i < (int)B) {
            const int64_t li = idx[i];
            if (li >= 0 && li < size) {
                int64_t phys = head + li;
                if (phys >= cap) phys -= cap;
                physv[u] = phys;
                int h = (int)(((u64)phys * 0x9E3779B97F4A7C15ull) >> (64 - 13));
                for (;;) {
                    const u64 prev = atomicCAS(&s_key[h], EMPTY, (u64)phys);
                    if (prev == EMPTY || prev == (u64)phys) break;
                    h = (h + 1) & (UPD_HASH - 1);
                }
                ent[u] = h;
                atomicMax(&s_mark[h], i);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + 1024 * u;
        const int64_t phys = physv[u];
        if (phys < 0 || s_mark[ent[u]] != i) continue;
        const double p = (double)newp[i];
        prio[phys] = p;
        const u64 m = mctsq_mass(p, alpha, shift);
        const u64 old = tree[T.off[0] + phys];
        tree[T.off[0] + phys] = m;
        const u64 delta = m - old;                           // wraps for decreases; sums stay exact
        if (delta == 0) continue;
        int64_t node = phys;
        for (int l = 1; l <= top; ++l) {
            node >>= 6;
            if (l == top && lds_top) atomicAdd(&s_top[node], delta);
            else if (l == top - 1 && lds_l2) atomicAdd(&s_l2[node], delta);
            else atomicAdd(&tree[T.off[l] + node], delta);
        }
    }
    __syncthreads();
    if (lds_top && tid < (int)T.cnt[top] && s_top[tid] != 0) atomicAdd(&tree[T.off[top] + tid], s_top[tid]);
    if (lds_l2)
        for (int i = tid; i < (int)T.cnt[top - 1]; i += 1024)
            if (s_l2[i] != 0) atomicAdd(&tree[T.off[top - 1] + i], s_l2[i]);
}

// Push: the K new entries are one run of ring slots (two when it wraps) and all carry the SAME mass (the current
// maximum priority), so nothing needs an atomic on the lower levels: one workgroup owns one 4096-slot span (= 64
// level-1 nodes = one level-2 node), writes the new leaves, recomputes every level-1 node the run touches from its 64
// children (new mass inside the run, stored leaf outside), its level-2 node from the 64 level-1 values, and hands the
// level-2 delta to the (few-word) levels above with one atomic each.  Spans the run does not touch leave at once.
// (One global atomic per level and entry queued a million pushes on the root word: 12 ms per million, measured.)
constexpr int PUSH_THREADS = 1024, PUSH_SPAN = 4096;
__global__ __launch_bounds__(PUSH_THREADS) void per_push_kernel(u64* __restrict__ tree, PerTree T,
                                                                double* __restrict__ prio, int64_t cap,
                                                                int64_t first_slot, int64_t K,
                                                                const double* __restrict__ maxprio, double alpha,
                                                                int shift) {
    __shared__ u64 s_l1[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, top = T.nlev - 1;
    const int64_t span0 = (int64_t)blockIdx.x * PUSH_SPAN;
    // does the run [first_slot, first_slot + K) mod cap meet this span?  (two plain intervals)
    const int64_t e1 = first_slot + K <= cap ? first_slot + K : cap, e2 = first_slot + K - cap;   // [first, e1) and [0, e2)
    const int64_t hi = span0 + PUSH_SPAN < cap ? span0 + PUSH_SPAN : cap;
    const bool meets = (first_slot < hi && e1 > span0) || (e2 > span0);
    if (!meets) return;
    const double p = *maxprio;
    const u64 m = mctsq_mass(p, alpha, shift);
    for (int j = wv; j < 64; j += PUSH_THREADS / 64) {           // level-1 node j of the span: leaves [64 j, 64 j + 64)
        const int64_t leaf = span0 + (int64_t)j * 64 + lane;
        const bool in = leaf < cap && ((leaf >= first_slot && leaf < e1) || leaf < e2);
        const bool any = __ballot(in) != 0;                      // wave-uniform
        u64 v = 0;
        if (any) {
            if (in) { tree[T.off[0] + leaf] = m; prio[leaf] = p; v = m; }
            else if (leaf < cap) v = tree[T.off[0] + leaf];
            v = readlane_u64(wave_incl_scan(v, lane), 63);
            if (top >= 1 && lane == 0) tree[T.off[1] + (span0 >> 6) + j] = v;
        } else if (top >= 2) {                                   // untouched node: its stored value enters the level-2 sum
            const int64_t n1 = (span0 >> 6) + j;
            v = n1 < T.cnt[1] ? tree[T.off[1] + n1] : 0;
        }
        if (lane == 0) s_l1[j] = v;
    }
    if (top < 2) return;
    __syncthreads();
    if (wv == 0) {
        const u64 tot = readlane_u64(wave_incl_scan(s_l1[lane], lane), 63);
        if (lane == 0) {
            const int64_t n2 = span0 >> 12;
            const u64 old = tree[T.off[2] + n2];
            tree[T.off[2] + n2] = tot;
            const u64 delta = tot - old;
            int64_t node = n2;
            if (delta)
                for (int l = 3; l <= top; ++l) { node >>= 6; atomicAdd(&tree[T.off[l] + node], delta); }
        }
    }
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

static int sample_launch(PerSampleArgs& a, void* ws, hipStream_t st) {
    // workspace: [0, 64) control words (the ticket: zero when first handed over, left zero by every launch),
    // raw fp64 weights behind them unless the caller supplies its own array
    a.ticket = reinterpret_cast<unsigned*>(ws);
    const bool fused_norm = (a.w || a.wmax) && a.B <= SMP_FUSED_NORM_MAX && ws != nullptr;
    float* w = a.w; double* wmax = a.wmax;
    if (!fused_norm) { a.w = nullptr; a.wmax = nullptr; }
    a.nsample = (int)((a.B + SMP_DRAWS - 1) / SMP_DRAWS);
    const int nzero = (int)((a.zero_n4 + SMP_THREADS - 1) / SMP_THREADS);
    hipLaunchKernelGGL(per_sample_kernel, dim3((unsigned)(a.nsample + a.nprep + nzero)), dim3(SMP_THREADS), 0, st, a);
    if (!fused_norm && w) hipLaunchKernelGGL(per_weight_norm_kernel, dim3(1), dim3(1024), 0, st, a.w64, a.B, w, wmax);
    else if (!fused_norm && wmax) hipLaunchKernelGGL(per_weight_max_kernel, dim3(1), dim3(1024), 0, st, a.w64, a.B, wmax);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

static int fill_passengers(PerSampleArgs& a, const SamplePassengers* extra) {
    if (extra && extra->prep) { a.nprep = (DQN_PREP_BLOCKS * 256 + SMP_THREADS - 1) / SMP_THREADS; a.prep = *extra->prep; }
    if (extra && extra->zero) {
        MCTSQ_CHECK(extra->zero_floats % 4 == 0 && ((uintptr_t)extra->zero & 15) == 0 && extra->zero_floats >= 0 &&
                        extra->zero_floats < ((int64_t)1 << 31), "passenger zero-fill: bad buffer");
        a.zero = extra->zero; a.zero_n4 = extra->zero_floats / 4;
    }
    return 0;
}

int launch_per_sample(const u64* tree, const double* prio, int64_t cap, int64_t head, int64_t size,
                      const double* u, int64_t B, double alpha, double beta, int64_t mass_cap, int64_t* idx,
                      int64_t* phys, float* w, double* wmax, void* ws, const PerGatherArgs* ga, hipStream_t st,
                      const SamplePassengers* extra) {
    PerSampleArgs a;
    memset(&a, 0, sizeof(a));
    if (fill_passengers(a, extra)) return -1;
    if (per_tree_desc(cap, &a.T)) return -1;
    a.tree = tree; a.prio = prio; a.cap = cap; a.head = head; a.size = size; a.u = u; a.B = B;
    a.alpha = alpha; a.beta = beta; a.shift = per_mass_shift(mass_cap);
    a.idx_out = idx; a.phys_out = phys; a.w64 = reinterpret_cast<double*>((char*)ws + PER_WS_CTRL_BYTES);
    a.w = w; a.wmax = wmax;
    if (ga) a.ga = *ga;
    return sample_launch(a, ws, st);
}

int launch_per_sample_at(const u64* tree, const double* prio, int64_t cap, int64_t head, int64_t size,
                         const u64* targets, int64_t B, double alpha, double beta, int64_t mass_cap, double n_total,
                         double mass_total, int64_t* idx, int64_t* phys, double* w64, double* wmax,
                         hipStream_t st) {
    PerSampleArgs a;
    memset(&a, 0, sizeof(a));
    if (per_tree_desc(cap, &a.T)) return -1;
    a.tree = tree; a.prio = prio; a.cap = cap; a.head = head; a.size = size; a.targets = targets; a.B = B;
    a.alpha = alpha; a.beta = beta; a.shift = per_mass_shift(mass_cap); a.n_eff = n_total; a.mass_eff = mass_total;
    a.idx_out = idx; a.phys_out = phys; a.w64 = w64; a.wmax = wmax;
    return sample_launch(a, nullptr, st);                    // no control words here: the maximum is its own launch
}

int launch_per_sample_shard(const u64* tree, const double* prio, int64_t cap, int64_t head, int64_t size,
                            const u64* targets, const int32_t* count, int64_t bcap, double alpha, double beta,
                            int64_t mass_cap, const u64* shard_mass, const int64_t* shard_size, int world,
                            int64_t* idx, int64_t* phys, double* w64, double* wmax, void* ws,
                            const PerGatherArgs* ga, hipStream_t st, const SamplePassengers* extra,
                            const ShardWeightOut* raw) {
    PerSampleArgs a;
    memset(&a, 0, sizeof(a));
    if (fill_passengers(a, extra)) return -1;
    if (per_tree_desc(cap, &a.T)) return -1;
    a.tree = tree; a.prio = prio; a.cap = cap; a.head = head; a.size = size; a.targets = targets; a.count = count;
    a.B = bcap; a.alpha = alpha; a.beta = beta; a.shift = per_mass_shift(mass_cap);
    a.shard_mass = shard_mass; a.shard_size = shard_size; a.world = world;
    a.idx_out = idx; a.phys_out = phys; a.w64 = w64; a.wmax = wmax;
    if (raw) {
        MCTSQ_CHECK(raw->w_raw && raw->slots && raw->world >= 1 && raw->world <= 64 && raw->rank >= 0 &&
                        raw->rank < raw->world && wmax && ws && bcap <= SMP_FUSED_NORM_MAX,
                    "raw-weight output needs the fused maximum (bcap <= %lld rows, a workspace, d_wmax)",
                    (long long)SMP_FUSED_NORM_MAX);
        a.w_raw = raw->w_raw; a.wmax_slots = raw->slots; a.slot_world = raw->world; a.slot_rank = raw->rank;
    }
    if (ga) a.ga = *ga;
    return sample_launch(a, ws, st);
}

int launch_per_shard_select(const double* u, int64_t B, const u64* shard_mass, int world, int rank, int64_t bcap,
                            u64* targets, int32_t* draw, int32_t* out, hipStream_t st) {
    hipLaunchKernelGGL(per_shard_select_kernel, dim3(1), dim3(1024), 0, st, u, B, shard_mass, world, rank, bcap,
                       targets, draw, out);
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

int launch_per_total_mass(const u64* tree, int64_t cap, int64_t size, u64* out, hipStream_t st) {
    PerTree T;
    if (per_tree_desc(cap, &T)) return -1;
    hipLaunchKernelGGL(per_total_mass_kernel, dim3(1), dim3(64), 0, st, tree, T, size, out);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_update(u64* tree, double* prio, int32_t* stamp, int64_t cap, int64_t head, int64_t size,
                      const int64_t* idx, const float* newp, int64_t B, double alpha, int64_t mass_cap,
                      hipStream_t st) {
    PerTree T;
    if (per_tree_desc(cap, &T)) return -1;
    const int shift = per_mass_shift(mass_cap);
    if (B <= UPD_FUSED_MAX) {
        hipLaunchKernelGGL(per_update_fused_kernel, dim3(1), dim3(1024), 0, st, tree, T, prio, cap, head, size, idx, newp,
                           B, alpha, shift);
    } else {
        const dim3 grid((unsigned)((B + 255) / 256));
        hipLaunchKernelGGL(per_update_mark_kernel, grid, dim3(256), 0, st, stamp, cap, head, size, idx, B);
        // entries per thread: one while the launch still leaves CUs idle, up to UPD_BULK_PER_THREAD for the big batches
        // (fewer workgroups = fewer flushes of the LDS sums)
        int per_thread = (int)(B / ((int64_t)UPD_BULK_THREADS * 256));
        per_thread = per_thread < 1 ? 1 : (per_thread > UPD_BULK_PER_THREAD ? UPD_BULK_PER_THREAD : per_thread);
        const int64_t per_block = (int64_t)UPD_BULK_THREADS * per_thread;
        hipLaunchKernelGGL(per_update_kernel, dim3((unsigned)((B + per_block - 1) / per_block)), dim3(UPD_BULK_THREADS), 0, st,
                           tree, T, prio, stamp, cap, head, size, idx, newp, B, alpha, shift, per_thread);
    }
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_per_push(u64* tree, double* prio, const double* maxprio, int64_t cap, int64_t head, int64_t size,
                    int64_t K, double alpha, int64_t mass_cap, hipStream_t st) {
    PerTree T;
    if (per_tree_desc(cap, &T)) return -1;
    const int64_t first = (head + size) % cap;
    hipLaunchKernelGGL(per_push_kernel, dim3((unsigned)((cap + PUSH_SPAN - 1) / PUSH_SPAN)), dim3(PUSH_THREADS), 0, st, tree,
                       T, prio, cap, first, K, maxprio, alpha, per_mass_shift(mass_cap));
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

// ------------------------------------------------------------------ sampling without a workspace argument
// mctsq_per_sample_noalloc (include/mctsq_hip.h): mctsq_per_sample for callers that cannot allocate or carry a
// workspace (HIP graph capture, serving loops, harnesses that forbid hipMalloc).  The sampling workspace -- the
// control words and one raw fp64 weight per draw -- lives in this file-scope device array instead; every GPU the
// library runs on holds its own copy (64 B + 8 B x 16384 draws = 128.06 KiB) until the process ends.
// `noalloc_scratch` is only a host-side stand-in for the array: the device address comes from
// hipGetSymbolAddress, once per device, and only that pointer is handed to kernels and memsets.
using namespace mctsq;

constexpr int64_t NOALLOC_MAX_DRAWS = SMP_FUSED_NORM_MAX;
static __device__ __align__(256) unsigned char noalloc_scratch[PER_WS_CTRL_BYTES + 8 * NOALLOC_MAX_DRAWS];

namespace {
std::mutex g_noalloc_mu;
void* g_noalloc_ptr[64];                 // per device ordinal: address of that device's copy of noalloc_scratch

// The array's address on the device the STREAM belongs to (each GPU holds its own copy at its own address), looked
// up once per device with that device current.  No memset: a file-scope __device__ array starts zeroed and every
// sampling launch leaves the control words zero again.  The lookup itself (hipGetSymbolAddress) is a synchronous
// runtime query, so the FIRST call on a device must be made outside stream capture (include/mctsq_hip.h).
int noalloc_scratch_ptr(hipStream_t st, void** out) {
    int dev = 0;
    hipDevice_t sdev = 0;
    if (st != nullptr && hipStreamGetDevice(st, &sdev) == hipSuccess) dev = (int)sdev;
    else MCTSQ_HIP(hipGetDevice(&dev));
    MCTSQ_CHECK(dev >= 0 && dev < 64, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_noalloc_mu);
    if (!g_noalloc_ptr[dev]) {
        int cur = 0;
        MCTSQ_HIP(hipGetDevice(&cur));
        if (cur != dev) MCTSQ_HIP(hipSetDevice(dev));
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(noalloc_scratch));
        if (cur != dev) (void)hipSetDevice(cur);
        MCTSQ_HIP(e);
        MCTSQ_CHECK(p != nullptr, "hipGetSymbolAddress returned NULL for the sampling scratch");
        g_noalloc_ptr[dev] = p;
    }
    *out = g_noalloc_ptr[dev];
    return 0;
}
}  // namespace

extern "C" int64_t mctsq_per_sample_noalloc_max_draws(void) { return NOALLOC_MAX_DRAWS; }

extern "C" int mctsq_per_sample_noalloc(const uint64_t* d_tree, const double* d_prio, int64_t capacity, int64_t head,
                                        int64_t size, const double* d_u, int64_t B, double alpha, double beta,
                                        int64_t mass_capacity, int64_t* d_idx, int64_t* d_phys, float* d_w,
                                        double* d_wmax, void* stream) {
    MCTSQ_CHECK(capacity >= 1 && size >= 1 && size <= capacity && head >= 0 && head < capacity &&
                    (size == capacity || head == 0), "bad ring geometry (capacity %lld, head %lld, size %lld)",
                (long long)capacity, (long long)head, (long long)size);
    MCTSQ_CHECK(mass_capacity >= capacity && mass_capacity <= (int64_t)1 << 36, "mass_capacity out of range");
    MCTSQ_CHECK(B >= 0, "negative batch");
    MCTSQ_CHECK(B <= NOALLOC_MAX_DRAWS, "mctsq_per_sample_noalloc serves at most %lld draws per call (got %lld): its "
                "scratch is a fixed device array; use mctsq_per_sample with a workspace for more",
                (long long)NOALLOC_MAX_DRAWS, (long long)B);
    if (B == 0) return 0;
    MCTSQ_CHECK(d_tree && d_prio && d_u && d_idx && d_phys && d_w, "NULL device pointer");
    void* scratch = nullptr;
    if (noalloc_scratch_ptr((hipStream_t)stream, &scratch)) return -1;
    return launch_per_sample((const unsigned long long*)d_tree, d_prio, capacity, head, size, d_u, B, alpha, beta,
                             mass_capacity, d_idx, d_phys, d_w, d_wmax, scratch, nullptr, (hipStream_t)stream);
}

