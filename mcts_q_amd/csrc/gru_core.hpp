// GRU-attention encoder building block shared by the stand-alone encoder kernel (enc_gru.hip) and
// the fused three-encoder + head kernel (fused_gru_head.hip).
//
// One "group" = 4 waves (256 lanes) advancing one 16-sample tile of one encoder slot; wave wq of the
// group owns hidden units 16wq..16wq+15.  Per step every wave computes its 3 gate tiles
// [16 samples x 16 units] = [x_t | h_{t-1}] . [W_ih | W_hh]^T.  The input projection (K <= 8) uses
// v_mfma_f32_16x16x4_f32 (exact fp32).  The recurrent product (K = 64) uses the 16x-faster fp16
// matrix instruction on a two-way split of both operands (device_utils.hpp: split2h): |h| <= 1 and the
// rows of W_hh are scaled by powers of two into the upper fp16 range, where hi + lo carries 22+ bits of
// the value, and the three cross products hi*hi, hi*lo, lo*hi are accumulated in fp32 -- the dropped
// lo*lo is below 2^-22 of the product; measured error against an fp64 evaluation is that of a plain
// fp32 fma chain (tests/test_gpu_qnet.py).  The accumulators of unit j therefore hold 2^k_j times the
// gate pre-activation (k_j = 14 + the row's shift): biases and input-projection weights are scaled by
// the same power of two when they are fetched and the constant inside the gate's exp2 argument by its
// inverse, all exact, so the step has no instruction for it.  W fragments (pre-split) stay resident in VGPRs, h is
// split by the lane that produces it and exchanged through double-buffered LDS planes.  The four waves of
// a group synchronise through an arrival counter in LDS (GruWave::sync), NOT the workgroup barrier,
// so groups of different window length that share a workgroup advance independently and the short
// windows fill the issue gaps of the long one.  The attention pooling trails the recurrence by two
// steps so that all of its VALU / LDS work sits under the MFMA chains; the softmax over time is
// evaluated online (running max / sum, in base 2) so no [S,16,64] history is kept on chip.
// Reference: dqn_per.py:58-77.
#pragma once
#include "device_utils.hpp"
#include "kernels.hpp"

#ifndef MCTSQ_GRU_PK
#define MCTSQ_GRU_PK 1      // gate arithmetic on packed pairs of rows
#endif

namespace mctsq {

constexpr int HLD = 72;     // LDS row stride of an h plane (bf16 elements): 144 B keeps b128 reads conflict-free
constexpr int XLD = 132;    // LDS row stride of the staged window: >= S * (4 or 8)

struct GruLds {             // per-group LDS carve-out
    unsigned short hb[2][2][TILE * HLD];   // planes (hi, lo) of 2^14 h x double buffer, fp16 bits
    float xs[TILE * XLD];
    float spart[2][4][TILE];
    float dump[4][64 * 4];  // write-only sink of the non-publishing lanes (GruWave::publish)
    float red[2][4][TILE];
    unsigned arrivals;      // monotonically increasing: +1 per wave per sync point
    unsigned pad_[3];
};

// Arrival-counter synchronisation of the 4 waves of one group.  `target` = 4 * (sync points so far).
// Every wave of a workgroup is resident, so the spin always terminates.
__device__ __forceinline__ void group_sync(unsigned* ctr, unsigned target, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // this wave's LDS writes are done
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
        __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// RPL = sample rows per lane.  4: the group advances a 16-sample tile, lane (jl, q) holds rows 4q..4q+3 of its unit.
// 2: an 8-sample HALF tile (sample si of the eight sits in matrix row 4 (si / 2) + si % 2, i.e. lane (jl, q) holds
// samples 2q, 2q+1 as its rows r = 0, 1; matrix rows 4q+2, 4q+3 carry whatever LDS holds and nobody reads their
// outputs: a row of a matrix product depends on that row of the A operand only).  The matrix instructions are the
// same; every per-(row, unit) vector / transcendental instruction, LDS write and record store is issued for two rows
// instead of four.  For batches whose 16-sample tiles leave half the CUs idle (the learner's B = 1024: 2 x 64 tiles;
// the search's leaf batches) twice the workgroups at half the vector work each shorten the per-step chain.  The
// records keep their layout (a half tile owns a contiguous half of every lane-major block, as in gru_bwd_kernel<2>).
template <bool SAVE, int RPL = 4>
struct GruWave {
    static_assert(RPL == 4 || RPL == 2, "rows per lane");
    static constexpr int TR = 4 * RPL;                        // samples per group
    // sample (relative to the group's first row) of this lane's row r, and the matrix row of sample i
    __device__ __forceinline__ int srow(int r) const { return RPL == 4 ? 4 * q + r : 2 * q + r; }
    static __device__ __forceinline__ int mrow(int i) { return RPL == 4 ? i : 4 * (i >> 1) + (i & 1); }
    u32x4 whi[3][2], wlo[3][2];     // 2^sh W_hh[g*64 + j][32c + 8q .. +7], gate g, K chunk c, split-fp16
    float esr, esz, esn, invn;      // exp2-argument constants of the three gates with 2^-k of unit j folded in; 2^-k_n
    float wir[2], wiz[2], win[2];   // W_ih[g*64 + j][q (+4)], zero padded
    f32x4 brv, bzv, binv, bhnv;     // gate biases replicated over the 4 accumulator rows: the accumulators START from them
    float awj, abl;                 // score weight of unit j (log2 units); the score bias rides on one lane's product
    float hh[2][RPL];               // hh[t & 1] = h_t (this wave's 16 units, the lane's RPL rows)
    float att[RPL], mrun[RPL], lrun[RPL];
    bool bounded;                   // scores provably inside +-60 (log2 units): softmax without the running max
    float pooled[RPL], mean[RPL], rstd[RPL];
    float xin[4], pdead;            // staged window elements in flight (stage_issue -> stage_commit); weight of a dead fold
    float ab2, lnwj, lnbj;          // score bias (log2 units), LayerNorm scale / shift of unit j: fetched with the weights
    int lane, wq, jl, q, j, FP;
    unsigned nsync;
    // SAVE: running pointers of this wave's slice of the training record (advanced by one step's stride per step) and
    // of the score rows.  Computed from t and looked up through the slot record per step, the addresses cost 64-bit
    // multiplies and - worse - a scalar load of the score pointer inside the step, whose wait also drains the LDS
    // reads the step keeps in flight.
    float *rp_r, *rp_z, *rp_n, *rp_hn, *rp_h, *rp_sc;
    int64_t rec_stride, sc_stride;
    int rec_lo;                     // this lane's float offset inside a 256-float record block
    __device__ __forceinline__ void bind_records(const GruSlot& s, int64_t row0, int64_t Bp) {
        if (!SAVE) return;
        const int64_t ub0 = ((row0 / TILE) * 4 + wq) * 256;
        // RPL 2: samples 8 half + 2q + {0, 1} of the 16-sample tile = rows 2 (q & 1) + {0, 1} of record lane
        // jl + 16 (2 half + q / 2): one 8-byte store, a wave's 64 lanes one contiguous 512-byte half of the block
        const int half = (int)((row0 >> 3) & 1);
        rec_lo = RPL == 4 ? lane * 4 : (jl + 16 * (2 * half + (q >> 1))) * 4 + 2 * (q & 1);
        rp_r = s.sv_r + ub0; rp_z = s.sv_z + ub0; rp_n = s.sv_n + ub0; rp_hn = s.sv_hn + ub0;
        rp_h = s.sv_h + ub0 + Bp * E;
        rp_sc = s.sv_score + row0;
        rec_stride = (Bp / TILE) * 4 * 256;
        sc_stride = Bp;
    }

    __device__ __forceinline__ void bind(int lane_, int wq_) {
        lane = lane_; wq = wq_; jl = lane & 15; q = lane >> 4; j = 16 * wq + jl;
        nsync = 0;
    }
    // A wave may run several windows one after another.  Per window, in this order:
    //   stage_issue   the window's elements -> registers (HBM round trip starts first)
    //   init_weights  weight fragments -> registers; touches nothing the previous window's LayerNorm reads,
    //                 so both can be issued BEFORE it and their latency hides under it
    //   init_state    running sums (after the previous window's fin2)
    //   stage_commit  registers -> LDS (after every wave of the group is done with the previous window's LDS)
    __device__ __forceinline__ void init_weights(const GruSlot& s) {
        const int F = s.F;
        FP = F > 4 ? 8 : 4;
        const u32x4* wp = reinterpret_cast<const u32x4*>(s.wprep) + (int64_t)wq * (3 * 2 * 2 * 64) + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                whi[g][c] = wp[((g * 2 + c) * 2 + 0) * 64];
                wlo[g][c] = wp[((g * 2 + c) * 2 + 1) * 64];
            }
        // unit j's accumulators carry 2^k times the pre-activation (k per gate row, written by prep_gru_fwd)
        const int* kp = reinterpret_cast<const int*>(reinterpret_cast<const char*>(s.wprep) + GRU_FWD_FRAG_BYTES);
        const int kr = kp[j], kz = kp[E + j], kn = kp[2 * E + j];
        const float pr = pow2i(kr), pz = pow2i(kz), pn = pow2i(kn);
        esr = -LOG2E * pow2i(-kr); esz = -LOG2E * pow2i(-kz); esn = (2.0f * LOG2E) * pow2i(-kn); invn = pow2i(-kn);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k = q + 4 * ks;
            const bool ok = k < F;
            wir[ks] = ok ? s.wih[(int64_t)(0 * E + j) * F + k] * pr : 0.0f;
            wiz[ks] = ok ? s.wih[(int64_t)(1 * E + j) * F + k] * pz : 0.0f;
            win[ks] = ok ? s.wih[(int64_t)(2 * E + j) * F + k] * pn : 0.0f;
        }
        {
            const float br = (s.bih[j] + s.bhh[j]) * pr, bz = (s.bih[E + j] + s.bhh[E + j]) * pz;
            const float bin = s.bih[2 * E + j] * pn, bhn = s.bhh[2 * E + j] * pn;
            brv = (f32x4){br, br, br, br}; bzv = (f32x4){bz, bz, bz, bz};
            binv = (f32x4){bin, bin, bin, bin}; bhnv = (f32x4){bhn, bhn, bhn, bhn};
        }
        const float ab = s.ab[0] * LOG2E;                     // softmax evaluated in base 2
        ab2 = ab; lnwj = s.lnw[j]; lnbj = s.lnb[j];
        awj = s.aw[j] * LOG2E;
        abl = (wq == 0 && jl == 0) ? ab : 0.0f;
        // |h| <= 1, so |score| <= sum_j |aw_j| + |ab|: when that bound is small exp2(score) can neither
        // overflow nor vanish and the softmax over time needs no running maximum (attend<true>).
        // Every lane of every wave evaluates the same expression, so the four waves agree.
        float bound = fabsf(s.aw[jl]) + fabsf(s.aw[16 + jl]) + fabsf(s.aw[32 + jl]) + fabsf(s.aw[48 + jl]);
        bound = (row16_sum(bound) + fabsf(s.ab[0])) * LOG2E;
        bounded = __builtin_amdgcn_readfirstlane(bound < 60.0f ? 1 : 0) != 0;
        // The pooling trails the recurrence by two steps, so the first two folds of a window see no step
        // yet ("dead" folds).  Bounded mode lets them run unconditionally: h_{-1} = h_{-2} = 0 adds nothing
        // to att, both dead scores are exactly ab (stage() seeds the slot the first one reads), and lrun
        // starts at minus their weight: -2p + p + p == 0 exactly.
        pdead = fast_exp2(ab);
    }
    __device__ __forceinline__ void init_state() {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            hh[0][r] = 0.0f; hh[1][r] = 0.0f; att[r] = 0.0f;
            lrun[r] = bounded ? -2.0f * pdead : 0.0f;
            mrun[r] = bounded ? 0.0f : -1e30f;
        }
    }

    __device__ __forceinline__ void sync(GruLds& L) {
        nsync += 4;
        group_sync(&L.arrivals, nsync, lane);
    }

    // quotient e / d for 0 <= e < 2^22 with inv = 1 / d: the float estimate is off by at most one
    static __device__ __forceinline__ int div_small(int e, int d, float inv) {
        int qq = (int)((float)e * inv);
        const int r = e - qq * d;
        qq += r >= d ? 1 : 0;
        qq -= r < 0 ? 1 : 0;
        return qq;
    }
    // Staging, cooperative over the group's 256 lanes (gtid): xs[i][t*FP + f] (zero padded), h planes of
    // buffer 0 = 0.  The tile's 16 windows are one contiguous run of 16*S*Fd floats; the first 1024 of
    // them (all of them for every shipped window shape) travel through xin[].
    __device__ __forceinline__ void stage_issue(const GruSlot& s, int gtid, int64_t row0, int64_t B) {
        const int SFd = s.S * s.Fd, n = TR * SFd;
        const float* src = s.x + row0 * (int64_t)SFd;
        const int64_t lim = (B - row0) * (int64_t)SFd;        // floats present in the batch
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = u * 256 + gtid;
            xin[u] = (e < n && e < lim) ? src[e] : 0.0f;
        }
    }
    // The caller follows with a workgroup barrier (first window: it also publishes the zeroed arrival
    // counter) or the group sync.
    __device__ __forceinline__ void stage_commit(const GruSlot& s, GruLds& L, int gtid, int64_t row0, int64_t B, int64_t Bp,
                                                 bool first = true) {
        const int S = s.S, F = s.F, Fd = s.Fd, SFd = S * Fd, sh = FP == 8 ? 3 : 2;
        const int n = TR * SFd;                               // <= 16 * 32 * 8 = 4096
        // everything that does not need the loaded elements first
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            for (int idx = gtid; idx < TILE * HLD / 2; idx += 256) reinterpret_cast<unsigned*>(L.hb[pl][0])[idx] = 0u;
        if (gtid < 64) L.spart[0][gtid >> 4][gtid & 15] = gtid < 16 ? ab2 : 0.0f;   // score of h_{-2}: ab
        if (first && gtid == 0) L.arrivals = 0;
        // columns Fd .. FP-1: time features of the non-_per flavour, then zero padding
        const int npad = FP - Fd, Snp = S * npad;
        const float inv_snp = 1.0f / (float)max(Snp, 1), inv_np = 1.0f / (float)max(npad, 1);
        for (int idx = gtid; idx < TR * Snp; idx += 256) {
            const int i = div_small(idx, Snp, inv_snp), rem = idx - i * Snp, t = div_small(rem, npad, inv_np);
            const int f = Fd + rem - t * npad;
            L.xs[mrow(i) * XLD + (t << sh) + f] = (f < F && row0 + i < B) ? time_feature(s.tf, t, f - Fd) : 0.0f;
        }
        if (SAVE)                                             // h_{-1}: the tile's whole block 0 (RPL 2: from both halves)
            for (int idx = gtid; idx < TILE * E; idx += 256) s.sv_h[(row0 / TILE) * (int64_t)(TILE * E) + idx] = 0.0f;
        const float inv_sfd = 1.0f / (float)SFd, inv_fd = 1.0f / (float)Fd;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = u * 256 + gtid;
            if (e < n) {
                const int i = div_small(e, SFd, inv_sfd), rem = e - i * SFd, t = div_small(rem, Fd, inv_fd);
                L.xs[mrow(i) * XLD + (t << sh) + rem - t * Fd] = xin[u];
            }
        }
        for (int e0 = 1024; e0 < n; e0 += 1024) {             // longer windows than any shipped configuration
            const float* src = s.x + row0 * (int64_t)SFd;
            const int64_t lim = (B - row0) * (int64_t)SFd;
#pragma unroll 1
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256 + gtid;
                if (e < n) {
                    const int i = div_small(e, SFd, inv_sfd), rem = e - i * SFd, t = div_small(rem, Fd, inv_fd);
                    L.xs[mrow(i) * XLD + (t << sh) + rem - t * Fd] = e < lim ? src[e] : 0.0f;
                }
            }
        }
    }
    // training record of the staged window (time-major, 8 columns); call after the barrier
    __device__ __forceinline__ void save_window(const GruSlot& s, GruLds& L, int gtid, int64_t row0, int64_t Bp) {
        // sv_x[(t * Bp + row) * 8 + f]: one (step, row) pair per thread and pass, its eight floats as two 16-byte stores
        // (index math is shifts only; the element-wise form divided by S * 8 thirteen times per thread)
        const int n = TR * s.S;
        for (int idx = gtid; idx < n; idx += 256) {
            const int t = RPL == 4 ? idx >> 4 : idx >> 3, i = idx & (TR - 1);
            const float* x = &L.xs[mrow(i) * XLD + t * (FP == 8 ? 8 : 4)];
            f32x4 a = {x[0], x[1], x[2], x[3]}, b = {0, 0, 0, 0};
            if (FP == 8) b = (f32x4){x[4], x[5], x[6], x[7]};       // xs is zero padded up to FP features
            float* dst = s.sv_x + ((int64_t)t * Bp + row0 + i) * 8;
            *reinterpret_cast<f32x4*>(dst) = a;
            *reinterpret_cast<f32x4*>(dst + 4) = b;
        }
    }

    // ---- attention pooling, kept OFF the recurrence's critical path ---------------------------
    // The score of h_t needs all 64 units of a row, i.e. all four waves.  Each wave publishes the
    // partial sums of its 16 units one step late (during step t+1, under the MFMA chain) and folds
    // h_t into the running softmax sums two steps late (during step t+2, when the partials are
    // visible behind sync t+1).  spart is indexed by step parity: step t writes parity (t-1)&1 and
    // reads parity t&1; a wave cannot run more than one sync ahead, and its reads are complete
    // before it arrives (the release fence drains lgkmcnt), so the two never collide.
    __device__ __forceinline__ void publish(GruLds& L, const float (&hv)[RPL], int par) {
        float4 ps = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        ps.x = row16_sum(fmaf(awj, hv[0], abl)); ps.y = row16_sum(fmaf(awj, hv[1], abl));
        if constexpr (RPL == 4) { ps.z = row16_sum(fmaf(awj, hv[2], abl)); ps.w = row16_sum(fmaf(awj, hv[3], abl)); }
        // one lane per row group stores; the others write their (identical) sums to a dump slot of
        // their own, which keeps the instruction stream branch-free for the interleaving below
        float* dst = jl == 0 ? &L.spart[par][wq][4 * q] : &L.dump[wq][lane * 4];
        *reinterpret_cast<float4*>(dst) = ps;
    }
    // online softmax update of row r with score sc (log2 units), e = exp2(-|sc - mrun|): one of the
    // two rescale factors is always exp2(0) = 1.  mrun starts at -1e30 (finite): a score of -inf
    // (no step yet) leaves the sums untouched and the first real score resets them
    __device__ __forceinline__ void fold(float sc, float e, float hv, int r) {
        const bool up = sc > mrun[r];
        const float c_old = up ? e : 1.0f, c_new = up ? 1.0f : e;
        att[r] = att[r] * c_old + c_new * hv;
        lrun[r] = lrun[r] * c_old + c_new;
        mrun[r] = up ? sc : mrun[r];
    }
    // fold hv (= h_t of this wave's units, whose score parts are in sp) into the running sums.
    // BND: plain sums of exp2(score) (mrun stays 0); otherwise the online form with a running maximum.
    template <bool BND>
    __device__ __forceinline__ void attend(const GruSlot& s, const float4 (&sp)[4], const float (&hv)[RPL], bool live,
                                           int t, int64_t row0, int64_t Bp) {
        const float p[4][4] = {{sp[0].x, sp[0].y, sp[0].z, sp[0].w}, {sp[1].x, sp[1].y, sp[1].z, sp[1].w},
                               {sp[2].x, sp[2].y, sp[2].z, sp[2].w}, {sp[3].x, sp[3].y, sp[3].z, sp[3].w}};
        if (SAVE) {                                           // the four rows' scores (log2 units) as one 16-byte store
            const f32x4 sc4 = {(p[0][0] + p[1][0]) + (p[2][0] + p[3][0]), (p[0][1] + p[1][1]) + (p[2][1] + p[3][1]),
                               (p[0][2] + p[1][2]) + (p[2][2] + p[3][2]), (p[0][3] + p[1][3]) + (p[2][3] + p[3][3])};
            if (live) {                                       // wave-uniform; live folds come in step order 0, 1, 2, ...
                if (wq == 0 && jl == 0) {
                    if constexpr (RPL == 4) *reinterpret_cast<f32x4*>(rp_sc + 4 * q) = sc4;
                    else *reinterpret_cast<f32x2*>(rp_sc + 2 * q) = (f32x2){sc4[0], sc4[1]};
                }
                rp_sc += sc_stride;
            }
        }
#if MCTSQ_GRU_PK
        if (BND) {                                            // the common path, on packed pairs of rows
#pragma unroll
            for (int pp = 0; pp < RPL / 2; ++pp) {
                const f32x2 sc = ((f32x2){p[0][2 * pp], p[0][2 * pp + 1]} + (f32x2){p[1][2 * pp], p[1][2 * pp + 1]}) +
                                 ((f32x2){p[2][2 * pp], p[2][2 * pp + 1]} + (f32x2){p[3][2 * pp], p[3][2 * pp + 1]});
                const f32x2 e = {fast_exp2(sc.x), fast_exp2(sc.y)};
                const f32x2 a2 = e * (f32x2){hv[2 * pp], hv[2 * pp + 1]} + (f32x2){att[2 * pp], att[2 * pp + 1]};
                const f32x2 l2 = (f32x2){lrun[2 * pp], lrun[2 * pp + 1]} + e;
                att[2 * pp] = a2.x; att[2 * pp + 1] = a2.y; lrun[2 * pp] = l2.x; lrun[2 * pp + 1] = l2.y;
            }
            return;
        }
#endif
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            float sc = (p[0][r] + p[1][r]) + (p[2][r] + p[3][r]);
            if (BND) {
                const float e = fast_exp2(sc);
                att[r] = fmaf(e, hv[r], att[r]);
                lrun[r] += e;
            } else {
                sc = live ? sc : -INFINITY;
                fold(sc, fast_exp2(-fabsf(sc - mrun[r])), hv[r], r);
            }
        }
    }
    __device__ __forceinline__ void load_spart(GruLds& L, int par, float4 (&sp)[4]) {
#pragma unroll
        for (int w = 0; w < 4; ++w) sp[w] = *reinterpret_cast<const float4*>(&L.spart[par][w][4 * q]);
    }
    // the whole recurrence: S steps, each followed by the group sync.  Unrolled by two by hand: the
    // double-buffer parity is a template constant and h_t / h_{t-1} swap roles without register copies
    template <bool FP8, bool BND>
    __device__ __forceinline__ void run_(const GruSlot& s, GruLds& L, int64_t row0, int64_t Bp) {
        int t = 0;
        for (; t + 1 < s.S; t += 2) {
            step<FP8, BND, 0>(s, L, t, row0, Bp); sync(L);
            step<FP8, BND, 1>(s, L, t + 1, row0, Bp); sync(L);
        }
        if (t < s.S) { step<FP8, BND, 0>(s, L, t, row0, Bp); sync(L); }
        // h_{S-1} = hh[(S-1) & 1], h_{S-2} = hh[S & 1]: two more folds, one sync between them
        if (s.S & 1) flush_<BND, 0>(s, L, row0, Bp);
        else flush_<BND, 1>(s, L, row0, Bp);
    }
    __device__ __forceinline__ void run(const GruSlot& s, GruLds& L, int64_t row0, int64_t Bp) {
        bind_records(s, row0, Bp);
        if (FP == 8) { if (bounded) run_<true, true>(s, L, row0, Bp); else run_<true, false>(s, L, row0, Bp); }
        else { if (bounded) run_<false, true>(s, L, row0, Bp); else run_<false, false>(s, L, row0, Bp); }
    }
    template <bool BND, int LAST>                             // LAST = (S - 1) & 1
    __device__ __forceinline__ void flush_(const GruSlot& s, GruLds& L, int64_t row0, int64_t Bp) {
        const int S = s.S;
        float4 sp[4];
        publish(L, hh[LAST], LAST);
        load_spart(L, LAST ^ 1, sp);
        attend<BND>(s, sp, hh[LAST ^ 1], S >= 2, S - 2, row0, Bp);
        sync(L);
        load_spart(L, LAST, sp);
        attend<BND>(s, sp, hh[LAST], true, S - 1, row0, Bp);
    }

    // one recurrence step t (PAR = t & 1); the caller follows with sync().  On entry hh[PAR ^ 1] = h_{t-1},
    // hh[PAR] = h_{t-2}; on exit hh[PAR] = h_t.
    template <bool FP8, bool BND, int PAR>
    __device__ __forceinline__ void step(const GruSlot& s, GruLds& L, int t, int64_t row0, int64_t Bp) {
        constexpr int cur = PAR, nxt = PAR ^ 1;
        // A fragments of 2^14 h_{t-1}: plane pl, K chunk c -> 8 fp16 of row lane&15 at k = 32c + 8q
        u32x4 ah[2], al[2];
        {
            const int o = (lane & 15) * HLD + 8 * q;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                ah[c] = *reinterpret_cast<const u32x4*>(&L.hb[0][cur][o + 32 * c]);
                al[c] = *reinterpret_cast<const u32x4*>(&L.hb[1][cur][o + 32 * c]);
            }
        }
        const float* xa = &L.xs[(lane & 15) * XLD + t * (FP8 ? 8 : 4) + q];
        const float x0 = xa[0];
        const float x1 = FP8 ? xa[4] : 0.0f;
        float4 sp[4];
        load_spart(L, cur, sp);
        // input projection, exact fp32, on top of the biases
        f32x4 accr = mfma16(x0, wir[0], brv), accz = mfma16(x0, wiz[0], bzv), acci = mfma16(x0, win[0], binv);
        if (FP8) {
            accr = mfma16(x1, wir[1], accr); accz = mfma16(x1, wiz[1], accz); acci = mfma16(x1, win[1], acci);
        }
        // recurrent product: three split terms per K chunk, smallest first; r and z chains alternate
#define T3(ACC0, ACC1, G0, G1, C)                                                              \
        ACC0 = mfma_h(al[C], whi[G0][C], ACC0);  ACC1 = mfma_h(al[C], whi[G1][C], ACC1);      \
        ACC0 = mfma_h(ah[C], wlo[G0][C], ACC0);  ACC1 = mfma_h(ah[C], wlo[G1][C], ACC1);      \
        ACC0 = mfma_h(ah[C], whi[G0][C], ACC0);  ACC1 = mfma_h(ah[C], whi[G1][C], ACC1);
        T3(accr, accz, 0, 1, 0)
        T3(accr, accz, 0, 1, 1)
#undef T3
        publish(L, hh[nxt], nxt);                             // pooling, two steps behind: scores of h_{t-1}
        attend<BND>(s, sp, hh[cur], t >= 2, t >= 2 ? t - 2 : 0, row0, Bp);   // ... and the fold of h_{t-2}
        // n gate: K chunk 0 into accn (on top of b_hn), chunk 1 into accm (two independent chains)
        f32x4 accn = mfma_h(al[0], whi[2][0], bhnv), accm = mfma_h(al[1], whi[2][1], (f32x4){0, 0, 0, 0});
        accn = mfma_h(ah[0], wlo[2][0], accn);  accm = mfma_h(ah[1], wlo[2][1], accm);
        accn = mfma_h(ah[0], whi[2][0], accn);  accm = mfma_h(ah[1], whi[2][1], accm);
        // Gate arithmetic on PAIRS of rows (packed fp32 instructions process two values per lane at the rate of one):
        // the step is bound by vector + transcendental issue once the matrix part is halved
        float rg[RPL], zg[RPL];
#if MCTSQ_GRU_PK
#pragma unroll
        for (int p = 0; p < RPL / 2; ++p) {
            const f32x2 er = (f32x2){accr[2 * p], accr[2 * p + 1]} * esr, ez = (f32x2){accz[2 * p], accz[2 * p + 1]} * esz;
            const f32x2 dr = (f32x2){fast_exp2(er.x), fast_exp2(er.y)} + 1.0f, dz = (f32x2){fast_exp2(ez.x), fast_exp2(ez.y)} + 1.0f;
            rg[2 * p] = fast_rcp(dr.x); rg[2 * p + 1] = fast_rcp(dr.y);
            zg[2 * p] = fast_rcp(dz.x); zg[2 * p + 1] = fast_rcp(dz.y);
        }
#else
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            rg[r] = fast_rcp(1.0f + fast_exp2(esr * accr[r]));      // sigmoid of 2^-k accr
            zg[r] = fast_rcp(1.0f + fast_exp2(esz * accz[r]));
        }
#endif
        // training record, lane-major (kernels.hpp: GruSlot): the lane's four rows of one unit are one 16-byte store, a
        // wave's store instruction one contiguous KB.  Uniform 64-bit base (scalar registers) + 32-bit lane offset: no
        // vector address math.  Each field leaves as soon as it exists (r, z here, under the n gate's products; n, W_hn h
        // and h at the end): five stores back to back from every wave of the CU at once queued up in the memory pipe
        // and the last wave's issue stalled behind them on the step's critical path.
        const int lo = rec_lo;
        auto put = [&](float* dst, const float (&v)[RPL]) {   // the lane's RPL rows of one unit: one 16- or 8-byte store
            if constexpr (RPL == 4) *reinterpret_cast<f32x4*>(dst) = (f32x4){v[0], v[1], v[2], v[3]};
            else *reinterpret_cast<f32x2*>(dst) = (f32x2){v[0], v[1]};
        };
        if (SAVE) {
            put(rp_r + lo, rg);
            put(rp_z + lo, zg);
        }
        float svn[RPL], svhn[RPL], svh[RPL];
#if MCTSQ_GRU_PK
#pragma unroll
        for (int p = 0; p < RPL / 2; ++p) {
            const f32x2 hn = (f32x2){accn[2 * p], accn[2 * p + 1]} + (f32x2){accm[2 * p], accm[2 * p + 1]};   // 2^k (W_hn h + b_hn)
            const f32x2 rg2 = {rg[2 * p], rg[2 * p + 1]}, zg2 = {zg[2 * p], zg[2 * p + 1]};
            const f32x2 an = ((f32x2){acci[2 * p], acci[2 * p + 1]} + rg2 * hn) * esn;
            const f32x2 dn = (f32x2){fast_exp2(an.x), fast_exp2(an.y)} + 1.0f;
            const f32x2 ng = 1.0f - 2.0f * (f32x2){fast_rcp(dn.x), fast_rcp(dn.y)};                           // tanh of 2^-k (...)
            const f32x2 hp = {hh[nxt][2 * p], hh[nxt][2 * p + 1]};
            const f32x2 hnew = ng + zg2 * (hp - ng);
            const f32x2 hs = hnew * (float)(1 << F16_TOP);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * p + e, i = 4 * q + r;
                unsigned short u0, u1;
                split2h(hs[e], u0, u1);
                L.hb[0][nxt][i * HLD + j] = u0;
                L.hb[1][nxt][i * HLD + j] = u1;
                if (SAVE) { svn[r] = ng[e]; svhn[r] = hn[e] * invn; svh[r] = hnew[e]; }
                hh[cur][r] = hnew[e];
            }
        }
#else
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = 4 * q + r;
            const float hn = accn[r] + accm[r];                                   // 2^k (W_hn h + b_hn)
            const float ng = 1.0f - 2.0f * fast_rcp(1.0f + fast_exp2(esn * (acci[r] + rg[r] * hn)));   // tanh of 2^-k (...)
            const float hnew = ng + zg[r] * (hh[nxt][r] - ng);
            unsigned short u0, u1;
            split2h(hnew * (float)(1 << F16_TOP), u0, u1);
            L.hb[0][nxt][i * HLD + j] = u0;
            L.hb[1][nxt][i * HLD + j] = u1;
            if (SAVE) { svn[r] = ng; svhn[r] = hn * invn; svh[r] = hnew; }
            hh[cur][r] = hnew;
        }
#endif
        if (SAVE) {
            put(rp_n + lo, svn);
            put(rp_hn + lo, svhn);
            put(rp_h + lo, svh);
            rp_r += rec_stride; rp_z += rec_stride; rp_n += rec_stride; rp_hn += rec_stride; rp_h += rec_stride;
        }
    }

    // LayerNorm over the 64 hidden units (4 waves x 16 lanes) with ONE group sync: every wave publishes the
    // sum and the centred sum of squares of its 16 units, and the four pairs are merged exactly
    // (M2 = sum M2_w + 16 sum (mean_w - mean)^2) -- the accuracy of the two-pass form without its second sync.
    // s, row0: for the training record.  The softmax statistics and the pooled vector leave for global memory HERE,
    // where they are last used, not in fin2: kept alive across the sync they cost register spills whose reloads sat
    // in the middle of the LayerNorm (scratch round trips: ~4 us per window in the saving kernel).
    __device__ __forceinline__ void fin0(GruLds& L, const GruSlot& s, int64_t row0) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            pooled[r] = att[r] / lrun[r];
            if (SAVE) {                                       // uniform base (scalar registers) + 32-bit lane offset
                const int i = srow(r);
                (s.sv_pool + row0 * (int64_t)E)[(unsigned)(i * E + j)] = pooled[r];
                if (wq == 0 && jl == 0) *reinterpret_cast<f32x2*>(s.sv_stat + row0 * 4 + (unsigned)(i * 4)) = (f32x2){mrun[r], lrun[r]};
            }
            const float ssum = row16_sum(pooled[r]);
            const float d = pooled[r] - ssum * (1.0f / 16);
            const float m2 = row16_sum(d * d);
            if (jl == 0) { L.red[0][wq][4 * q + r] = ssum; L.red[1][wq][4 * q + r] = m2; }
        }
    }
    // after the sync: writes the normalised feature to out[i * ld + j] (LDS or global) and the training record
    // (lnw, lnb: this window's lnwj / lnbj, read by the caller before a following window's init_weights)
    __device__ __forceinline__ void fin2(const GruSlot& s, GruLds& L, float* out, int ld, int64_t row0, float lnw, float lnb) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = 4 * q + r, si = srow(r);            // matrix row (LDS), sample (out, records)
            const float s0 = L.red[0][0][i], s1 = L.red[0][1][i], s2 = L.red[0][2][i], s3 = L.red[0][3][i];
            mean[r] = ((s0 + s1) + (s2 + s3)) * (1.0f / E);
            const float d0 = s0 * (1.0f / 16) - mean[r], d1 = s1 * (1.0f / 16) - mean[r];
            const float d2 = s2 * (1.0f / 16) - mean[r], d3 = s3 * (1.0f / 16) - mean[r];
            const float m2 = ((L.red[1][0][i] + L.red[1][1][i]) + (L.red[1][2][i] + L.red[1][3][i])) +
                             16.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            rstd[r] = 1.0f / sqrtf(m2 * (1.0f / E) + LN_EPS);
            out[si * ld + j] = (pooled[r] - mean[r]) * rstd[r] * lnw + lnb;
            if (SAVE && wq == 0 && jl == 0)
                *reinterpret_cast<f32x2*>(s.sv_stat + row0 * 4 + (unsigned)(si * 4 + 2)) = (f32x2){mean[r], rstd[r]};
        }
    }
    // fin2 for a consumer that multiplies on the bf16 matrix instruction: the normalised feature goes to three
    // bf16 planes (hi, mid, lo; `pstride` elements apart, row stride `ld`) at column out_col + j, and -- training --
    // to the global feature rows gfeats [Bp][192]
    __device__ __forceinline__ void fin2_split(const GruSlot& s, GruLds& L, unsigned short* planes, int pstride, int ld,
                                               int64_t row0, float lnw, float lnb, float* gfeats) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = 4 * q + r, si = srow(r);            // matrix row (LDS planes), sample (global rows)
            const float s0 = L.red[0][0][i], s1 = L.red[0][1][i], s2 = L.red[0][2][i], s3 = L.red[0][3][i];
            mean[r] = ((s0 + s1) + (s2 + s3)) * (1.0f / E);
            const float d0 = s0 * (1.0f / 16) - mean[r], d1 = s1 * (1.0f / 16) - mean[r];
            const float d2 = s2 * (1.0f / 16) - mean[r], d3 = s3 * (1.0f / 16) - mean[r];
            const float m2 = ((L.red[1][0][i] + L.red[1][1][i]) + (L.red[1][2][i] + L.red[1][3][i])) +
                             16.0f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            rstd[r] = 1.0f / sqrtf(m2 * (1.0f / E) + LN_EPS);
            const float v = (pooled[r] - mean[r]) * rstd[r] * lnw + lnb;
            unsigned u0, u1, u2;
            split3(v, u0, u1, u2);
            unsigned short* dst = planes + i * ld + s.out_col + j;
            dst[0] = (unsigned short)(u0 >> 16);
            dst[pstride] = (unsigned short)(u1 >> 16);
            dst[2 * pstride] = (unsigned short)(u2 >> 16);
            if (SAVE) {
                if (gfeats) (gfeats + row0 * (int64_t)FEAT)[(unsigned)(si * FEAT + j + s.out_col)] = v;
                if (wq == 0 && jl == 0) *reinterpret_cast<f32x2*>(s.sv_stat + row0 * 4 + (unsigned)(si * 4 + 2)) = (f32x2){mean[r], rstd[r]};
            }
        }
    }
};

}  // namespace mctsq
