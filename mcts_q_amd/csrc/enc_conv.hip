// ConvAttentionEnc (dqn_per.py:21-55) forward and backward.
//
// One workgroup (4 waves) = one 16-sample tile of one conv slot.  Rows of every product are the 16
// samples at ONE time step, so a k=3 convolution multiplies three consecutive rows [sample][channel] of
// the previous layer (zero rows at t = -1 and t = S): conv2 is [16 x 192].[192 x 64] per step.  The forward
// runs it on the fp16 matrix instruction (two-way split, DESIGN.md section 3: prepared weight fragments with a
// power-of-two scale per output channel resident in 48 VGPRs; the scale of the h1 rows follows from the
// workgroup's largest input magnitude and the conv1 weights) and keeps only two h1 rows in LDS; the backward
// works on a full LDS image with fp32 MFMA.
// The softmax over time is folded online (one barrier per step), then LayerNorm.
#include "kernels.hpp"
#include "mlp_core.hpp"
#include "learner.hpp"
#include "pool_core.hpp"

namespace mctsq {

constexpr int CLD = 68;                   // channel row stride of the LDS image (64 + 4)

template <int SMAX>
struct ConvLds {
    float img[(SMAX + 2) * TILE * CLD];   // h1 (forward) / delta2 (backward), rows t+1
    float xs[TILE * ((SMAX + 2) * 8 + 4)];
    PoolLds pool;
    PoolBwdLds pb;
};

constexpr int CONV_SMAX = 26;
constexpr int CXLD = (CONV_SMAX + 2) * 8 + 4;
constexpr int CHLD = 72;                  // row stride of an fp16 plane (elements): 144 B keeps the b128 reads conflict-free
struct ConvFwdLds {
    // h1 rows in flight as split-fp16 planes (hi, lo) of 2^kh h1: row tp lives in slot tp & 1, written one step before
    // it is read.  The PRODUCING lane splits its four values; as fp32 rows every wave split the whole 64-channel row
    // again (64 of the step's ~190 vector instructions per wave, four times over per workgroup)
    unsigned short rowp[2][2][TILE * CHLD];
    float xs[TILE * CXLD];
    PoolLds pool;
    float bound[3][4];          // per wave: largest |x| staged, largest sum_k |w1_c[k]|, largest |b1_c| (the h1 bound below)
};

// Forward.  h1 never exists as a whole image: at step t every wave computes its 16 channels of row t+3 (conv1 is
// 3-6 fp32 MFMAs), splits them into two fp16 planes and writes those into a two-slot LDS buffer, reads the fragments
// of row t+2 ONCE, and multiplies rows t, t+1, t+2 (taps 0, 1, 2 of conv2) from three rotating register sets.
// ~20 KB of LDS and ~180 VGPRs let two workgroups (e.g. the price and the process window of one tile) share a CU.
// Passenger: the first n_mlp workgroups of the grid run the gas/EUA MLP encoder of the same network instead (one 16-sample
// tile each, mlp_core.hpp: a few microseconds, no dependency either way) - as a launch of its own behind this one it
// cost its 4.6 us plus a launch boundary of the conv/conv/mlp forward's 40.
template <bool SAVE>
__global__ __launch_bounds__(256, 2) void conv_enc_fwd_kernel(ConvFwdArgs args, MlpArgs mlp, int n_mlp) {
    __shared__ __attribute__((aligned(16))) ConvFwdLds L;
    static_assert(sizeof(ConvFwdLds) >= MLP_LDS_FLOATS * sizeof(float), "the passenger's LDS fits the conv encoder's");
    if ((int)blockIdx.x < n_mlp) {
        float* lds = reinterpret_cast<float*>(&L);
        mlp_enc_fwd_body<SAVE>(mlp, lds, lds + TILE * MXLD, (int)blockIdx.x);
        return;
    }
    constexpr int XLD = CXLD;
    const int cbid = (int)blockIdx.x - n_mlp;
    const int slot = cbid % args.nslots;
    const int64_t row0 = (int64_t)(cbid / args.nslots) * TILE, Bp = args.Bp;
    const ConvSlot& s = args.slot[slot];
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6, jl = lane & 15, q = lane >> 4;
    const int j = 16 * wq + jl;
    const int S = s.S, F = s.F, Fd = s.Fd, FP = F > 4 ? 8 : 4;

    // ---- stage the window (zero rows at both ends) ---------------------------------------------
    float xmax = 0.0f;
    for (int idx = tid; idx < TILE * (S + 2) * FP; idx += 256) {
        const int i = idx / ((S + 2) * FP), rem = idx - i * ((S + 2) * FP), tp = rem / FP, f = rem - tp * FP;
        const int t = tp - 1;
        float v = 0.0f;
        if (t >= 0 && t < S && row0 + i < args.B && f < F)
            v = f < Fd ? s.x[(row0 + i) * (int64_t)(S * Fd) + t * Fd + f] : time_feature(s.tf, t, f - Fd);
        L.xs[i * XLD + rem] = v;
        xmax = fmaxf(xmax, fabsf(v));
        if (SAVE) s.sv_x[((int64_t)tp * Bp + row0 + i) * 8 + f] = v;
    }
    if (SAVE && FP == 4)
        for (int idx = tid; idx < TILE * (S + 2) * 4; idx += 256) {
            const int i = idx / ((S + 2) * 4), rem = idx - i * ((S + 2) * 4);
            s.sv_x[((int64_t)(rem >> 2) * Bp + row0 + i) * 8 + 4 + (rem & 3)] = 0.0f;
        }
    if (SAVE)                                                    // zero pad rows of the h1 record
        for (int idx = tid; idx < 2 * TILE * E; idx += 256) {
            const int e = idx & (TILE * E - 1), i = e >> 6, c = e & 63;
            const int tp = idx < TILE * E ? 0 : S + 1;
            s.sv_h1[((int64_t)tp * Bp + row0 + i) * E + c] = 0.0f;
        }
    // ---- weight fragments ----------------------------------------------------------------------
    const int nk1 = 3 * FP / 4;                                  // 3 or 6 k-steps for conv1
    float w1f[6];
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
        const int k = 4 * kk + q, tap = k / FP, f = k - tap * FP;
        w1f[kk] = (kk < nk1 && f < F) ? s.w1[(int64_t)j * F * 3 + f * 3 + tap] : 0.0f;
    }
    // conv2 weights: prepared split-fp16 B fragments, chunk = 2 tap + half (32 input channels each), output channel j
    // scaled by 2^s2[j] (conv_prepare_kernel; the exponents sit behind the fragments)
    u32x4 w2p[6][2];
    {
        const u32x4* wp = reinterpret_cast<const u32x4*>(s.wprep) + (int64_t)wq * (6 * 2 * 64) + lane;
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w2p[c][pl] = wp[(c * 2 + pl) * 64];
    }
    const int s2j = reinterpret_cast<const int*>(reinterpret_cast<const char*>(s.wprep) + CONV_FRAG_BYTES)[j];
    const float b1j = s.b1[j], b2j = s.b2[j];
    PoolCore<SAVE> pc;
    pc.init(s.aw, s.ab, lane, wq);
    // Bound of the h1 rows (the A operand of conv2): |h1_c| <= |b1_c| + xmax sum_k |w1_c[k]|, with xmax the largest
    // input magnitude of THIS workgroup's windows (time features included).  Every lane folds its share, the waves
    // exchange two numbers each, and all lanes derive the same power of two 2^kh for the rows.
    {
        float wsum = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) wsum += fabsf(w1f[kk]);
        wsum += __shfl_xor(wsum, 16, 64);
        wsum += __shfl_xor(wsum, 32, 64);                           // sum |w1_j[:]| of this lane's channel j
        float xm = xmax, wm = wsum, bm = fabsf(b1j);
        for (int o = 32; o > 0; o >>= 1) {
            xm = fmaxf(xm, __shfl_xor(xm, o, 64)); wm = fmaxf(wm, __shfl_xor(wm, o, 64)); bm = fmaxf(bm, __shfl_xor(bm, o, 64));
        }
        if (lane == 0) { L.bound[0][wq] = xm; L.bound[1][wq] = wm; }
        if (lane == 1) L.bound[2][wq] = bm;
    }
    __syncthreads();
    const float hbound = fmaxf(fmaxf(L.bound[2][0], L.bound[2][1]), fmaxf(L.bound[2][2], L.bound[2][3])) +
                         fmaxf(fmaxf(L.bound[0][0], L.bound[0][1]), fmaxf(L.bound[0][2], L.bound[0][3])) *
                             fmaxf(fmaxf(L.bound[1][0], L.bound[1][1]), fmaxf(L.bound[1][2], L.bound[1][3]));
    const int kh = f16_row_shift(hbound);
    const float sh = pow2i(kh), inv2 = pow2i(-(kh + s2j));          // row scale; un-scale of output channel j
    // the rows live in LDS already scaled: conv1's weights and bias take the factor once (a power of two: the rows
    // are the same bits times 2^kh), instead of sixteen multiplies per step and wave where the rows are split
    const float b1s = b1j * sh, ish = pow2i(-kh);
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) w1f[kk] *= sh;

    // conv1 + ReLU for image row tp (1..S = time step tp-1; S+1 = the zero pad) -> LDS slot tp & 1
    auto conv1_row = [&](int tp) {
        f32x4 acc = {0, 0, 0, 0};
        const bool real = tp <= S;
        if (real) {
            const float* xa = &L.xs[(lane & 15) * XLD + (tp - 1) * FP + q];
#pragma unroll
            for (int kk = 0; kk < 6; ++kk)
                if (kk < nk1) acc = mfma16(xa[4 * kk], w1f[kk], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = real ? fmaxf(acc[r] + b1s, 0.0f) : 0.0f;
            unsigned short u0, u1;
            split2h(v, u0, u1);
            L.rowp[tp & 1][0][(4 * q + r) * CHLD + j] = u0;
            L.rowp[tp & 1][1][(4 * q + r) * CHLD + j] = u1;
            if (SAVE && real) s.sv_h1[((int64_t)tp * Bp + row0 + 4 * q + r) * E + j] = v * ish;
        }
    };
    // ---- conv2 + ReLU, attention pooling folded online ---------------------------------------------
    // [16 x 192] . [192 x 64] per step on the fp16 matrix instruction over the two-way split of both operands
    // (three cross terms, fp32 accumulate: DESIGN.md section 3).
    struct RowFrag { u32x4 h[2], l[2]; };
    auto load_row = [&](int tp, RowFrag& f) {
        const int o = (lane & 15) * CHLD + 8 * q;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f.h[hf] = *reinterpret_cast<const u32x4*>(&L.rowp[tp & 1][0][o + 32 * hf]);
            f.l[hf] = *reinterpret_cast<const u32x4*>(&L.rowp[tp & 1][1][o + 32 * hf]);
        }
    };
#define CONV_MM6(ACC, F, HF, W)                                                                           \
    ACC = mfma_h(F.l[HF], (W)[0], ACC); ACC = mfma_h(F.h[HF], (W)[1], ACC); ACC = mfma_h(F.h[HF], (W)[0], ACC);
    auto step = [&](int t, RowFrag& ra, RowFrag& rb, RowFrag& rc) {
        const int cur = t & 1, nxt = cur ^ 1;
        if (t + 3 <= S + 1) conv1_row(t + 3);                    // for the next step, into the other slot
        load_row(t + 2, rc);
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
        CONV_MM6(acc0, ra, 0, w2p[0]) CONV_MM6(acc1, rb, 0, w2p[2]) CONV_MM6(acc2, rc, 0, w2p[4])
        CONV_MM6(acc0, ra, 1, w2p[1]) CONV_MM6(acc1, rb, 1, w2p[3]) CONV_MM6(acc2, rc, 1, w2p[5])
        if (t > 0) pc.attend(L.pool, cur, t - 1, s.sv_score, row0, Bp);
        float v[4];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {                         // packed pairs of rows: the step is bound by vector issue
            const f32x2 sm = ((f32x2){acc0[2 * pp], acc0[2 * pp + 1]} + (f32x2){acc1[2 * pp], acc1[2 * pp + 1]}) +
                             (f32x2){acc2[2 * pp], acc2[2 * pp + 1]};
            const f32x2 y = sm * inv2 + b2j;
            v[2 * pp] = fmaxf(y.x, 0.0f); v[2 * pp + 1] = fmaxf(y.y, 0.0f);
        }
        if (SAVE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s.sv_h2[((int64_t)t * Bp + row0 + 4 * q + r) * E + j] = v[r];
        }
        pc.publish(L.pool, nxt, v);
        __syncthreads();
    };
    {
        RowFrag f0, f1, f2;
        const u32x4 z = {0, 0, 0, 0};
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) { f0.h[hf] = z; f0.l[hf] = z; }   // row 0: the zero pad at t = -1
        conv1_row(1);
        __syncthreads();
        load_row(1, f1);
        conv1_row(2);
        __syncthreads();
        // whole triples first: after three steps the three register sets are back in their roles, so the loop's back
        // edge needs no register moves (with the window's tail handled inside the loop the compiler rotated 34
        // registers per step); then the one or two steps that are left
        int t = 0;
        for (; t + 2 < S; t += 3) {
            step(t, f0, f1, f2);
            step(t + 1, f1, f2, f0);
            step(t + 2, f2, f0, f1);
        }
        if (t < S) step(t, f0, f1, f2);
        if (t + 1 < S) step(t + 1, f1, f2, f0);
    }
#undef CONV_MM6
    pc.attend(L.pool, S & 1, S - 1, s.sv_score, row0, Bp);
    pc.fin0(L.pool);
    __syncthreads();
    pc.fin1(L.pool);
    __syncthreads();
    pc.fin2(L.pool, s.lnw, s.lnb, args.feats + row0 * FEAT + s.out_col, FEAT, s.sv_pool, s.sv_stat, row0);
}

// one thread per (slot, wave, chunk, lane): 8 input channels of one tap -> 2 x 8 fp16, output channel (row of
// conv2.weight [64][64][3]) scaled by its own power of two; the 64 exponents follow the fragments
struct ConvPrepArgs { const float* w2[2]; u32x4* out; };
__global__ __launch_bounds__(256) void conv_prepare_kernel(ConvPrepArgs a) {
    const int tid = blockIdx.x * 256 + threadIdx.x;           // grid = n * 6 blocks of 256
    const int lane = tid & 63, rest = tid >> 6, c = rest % 6, wq = (rest / 6) & 3, slot = rest / 24;
    const int jl = lane & 15, q = lane >> 4, tap = c >> 1, cin0 = 32 * (c & 1) + 8 * q;
    const int ch = 16 * wq + jl;
    const float4* row = reinterpret_cast<const float4*>(a.w2[slot] + (int64_t)ch * (E * 3));
    float amax = 0.0f;
#pragma unroll 8
    for (int i = 0; i < E * 3 / 4; ++i) {
        const float4 v = row[i];
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    const int sh = f16_row_shift(amax);
    const float sc = pow2i(sh);
    const float* w = a.w2[slot] + (int64_t)ch * (E * 3) + tap;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[(cin0 + e) * 3] * sc;
    u32x4 hi, lo;
    split8h(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
    u32x4* base = a.out + (int64_t)slot * (CONV_PREP_BYTES / 16);
    u32x4* o = base + ((wq * 6 + c) * 2) * 64 + lane;
    o[0] = hi; o[64] = lo;
    if (c == 0 && q == 0) reinterpret_cast<int*>(base + CONV_FRAG_BYTES / 16)[ch] = sh;
}
int launch_conv_prepare(const float* const w2[2], int n, void* prep, hipStream_t st) {
    if (n == 0) return 0;
    ConvPrepArgs a;
    for (int i = 0; i < 2; ++i) a.w2[i] = w2[i < n ? i : 0];
    a.out = reinterpret_cast<u32x4*>(prep);
    hipLaunchKernelGGL(conv_prepare_kernel, dim3((unsigned)(n * 6)), dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_conv_fwd(const ConvFwdArgs& a, bool save, hipStream_t st, const MlpArgs* mlp) {
    if (a.nslots == 0 || a.B == 0) return mlp ? launch_mlp_fwd(*mlp, save, st) : 0;
    int smax = 0;
    for (int i = 0; i < a.nslots; ++i)
        MCTSQ_CHECK(a.slot[i].wprep && ((uintptr_t)a.slot[i].wprep & 15) == 0, "conv encoder needs prepared weights");
    for (int i = 0; i < a.nslots; ++i) {
        MCTSQ_CHECK(a.slot[i].S <= 26 && a.slot[i].F <= MAX_F, "conv encoder window %dx%d unsupported (S <= 26)",
                    a.slot[i].S, a.slot[i].F);
        smax = a.slot[i].S > smax ? a.slot[i].S : smax;
    }
    MCTSQ_CHECK(a.Bp % TILE == 0 && a.Bp >= a.B, "padded batch must be a multiple of %d", TILE);
    MlpArgs m = MlpArgs();
    int n_mlp = 0;
    if (mlp) {
        MCTSQ_CHECK(3 * mlp->F <= 16 && mlp->Bp == a.Bp, "gas/EUA mlp passenger: input width %d > 16 or batch mismatch", 3 * mlp->F);
        m = *mlp;
        n_mlp = (int)(a.Bp / TILE);
    }
    dim3 grid((unsigned)(a.Bp / TILE * a.nslots + n_mlp)), blk(256);
    (void)smax;
    if (save) hipLaunchKernelGGL(conv_enc_fwd_kernel<true>, grid, blk, 0, st, a, m, n_mlp);
    else hipLaunchKernelGGL(conv_enc_fwd_kernel<false>, grid, blk, 0, st, a, m, n_mlp);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

// ======================================================================================= backward
// dfeat -> LayerNorm/attention backward -> delta2 (pre-activation gradient of conv2, kept in the
// LDS image and written out) -> delta1 = ReLU'(h1) * (delta2 (*) W2 transposed in time).
template <int SMAX>
__global__ __launch_bounds__(256) void conv_enc_bwd_kernel(ConvBwdArgs args) {
    __shared__ __attribute__((aligned(16))) ConvLds<SMAX> L;
    const int slot = blockIdx.x % args.nslots;
    const int64_t row0 = (int64_t)(blockIdx.x / args.nslots) * TILE, Bp = args.Bp;
    const ConvSlot& s = args.slot[slot];
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6, jl = lane & 15, q = lane >> 4;
    const int j = 16 * wq + jl;
    const int S = s.S;

    for (int idx = tid; idx < 2 * TILE * E; idx += 256) {
        const int e = idx & (TILE * E - 1), i = e >> 6, c = e & 63;
        L.img[((idx < TILE * E ? 0 : S + 1) * TILE + i) * CLD + c] = 0.0f;
    }
    // transposed-conv fragments: B[k = (tap, c)][col = c'] = W2[c][c'][tap], k = tap*64 + 16q + e
    float wt[3][16];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int e = 0; e < 16; ++e) wt[tap][e] = s.w2[(int64_t)(16 * q + e) * (E * 3) + j * 3 + tap];
    const float awj = s.aw[j];

    PoolBwdCore pb;
    pb.prepare(L.pb, args.dfeat, s.out_col, s.lnw, s.sv_stat, s.sv_pool, s.bw_p, lane, wq, row0);
    for (int t = 0; t < S; ++t) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = s.sv_h2[((int64_t)t * Bp + row0 + 4 * q + r) * E + j];
        pb.publish_e(L.pb, t, v);
    }
    __syncthreads();
    pb.finish(L.pb);
    // ---- delta2 ------------------------------------------------------------------------------------
    for (int t = 0; t < S; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            const int64_t row = (int64_t)t * Bp + row0 + i;
            float at, dsc;
            pb.step_terms(L.pb, t, r, s.sv_score[row], at, dsc);
            if (wq == 0 && jl == 0) s.bw_dsc[row] = dsc;
            const float h2 = s.sv_h2[row * E + j];
            const float d2 = h2 > 0.0f ? at * pb.dpool[r] + dsc * awj : 0.0f;
            s.bw_d2[row * E + j] = d2;
            L.img[((t + 1) * TILE + i) * CLD + j] = d2;
        }
    }
    __syncthreads();
    // ---- delta1[t][c'] = ReLU'(h1) * sum_{tap,c} delta2[t - tap + 1][c] W2[c][c'][tap] -------------
    for (int t = 0; t < S; ++t) {
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
        const float4* r0 = reinterpret_cast<const float4*>(&L.img[((t + 2) * TILE + (lane & 15)) * CLD + 16 * q]);   // tap 0
        const float4* r1 = reinterpret_cast<const float4*>(&L.img[((t + 1) * TILE + (lane & 15)) * CLD + 16 * q]);   // tap 1
        const float4* r2 = reinterpret_cast<const float4*>(&L.img[((t + 0) * TILE + (lane & 15)) * CLD + 16 * q]);   // tap 2
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 a0 = r0[c], a1 = r1[c], a2 = r2[c];
            acc0 = mfma16(a0.x, wt[0][4 * c + 0], acc0); acc1 = mfma16(a1.x, wt[1][4 * c + 0], acc1); acc2 = mfma16(a2.x, wt[2][4 * c + 0], acc2);
            acc0 = mfma16(a0.y, wt[0][4 * c + 1], acc0); acc1 = mfma16(a1.y, wt[1][4 * c + 1], acc1); acc2 = mfma16(a2.y, wt[2][4 * c + 1], acc2);
            acc0 = mfma16(a0.z, wt[0][4 * c + 2], acc0); acc1 = mfma16(a1.z, wt[1][4 * c + 2], acc1); acc2 = mfma16(a2.z, wt[2][4 * c + 2], acc2);
            acc0 = mfma16(a0.w, wt[0][4 * c + 3], acc0); acc1 = mfma16(a1.w, wt[1][4 * c + 3], acc1); acc2 = mfma16(a2.w, wt[2][4 * c + 3], acc2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = (int64_t)t * Bp + row0 + 4 * q + r;
            const float h1 = s.sv_h1[(row + Bp) * E + j];                  // block t+1
            s.bw_d1[row * E + j] = h1 > 0.0f ? (acc0[r] + acc1[r]) + acc2[r] : 0.0f;
        }
    }
}

int launch_conv_bwd(const ConvBwdArgs& a, hipStream_t st) {
    if (a.nslots == 0) return 0;
    int smax = 0;
    for (int i = 0; i < a.nslots; ++i) {
        MCTSQ_CHECK(a.slot[i].S <= 26, "conv encoder window too long");
        smax = a.slot[i].S > smax ? a.slot[i].S : smax;
    }
    dim3 grid((unsigned)(a.Bp / TILE * a.nslots)), blk(256);
    if (smax <= 12) hipLaunchKernelGGL((conv_enc_bwd_kernel<12>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((conv_enc_bwd_kernel<26>), grid, blk, 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
