// The shipped default network (gru/gru/gru encoders, 192-128-128-A head; config_mctsq.yaml:24-42)
// as ONE kernel: TripleEncoderDQN.forward (dqn_per.py:216-236) + the leaf epilogue
// (mcts_per.py:293, 370).  One workgroup = one 16-sample tile = 8 waves (two per SIMD, 256 VGPRs each):
//   waves 0-3  the longest window (price, 25 steps), waves 4-7 the other two back to back (10 + 3).
// Each 4-wave group synchronises through its own LDS arrival counter (gru_core.hpp), not the
// workgroup barrier, so the shorter windows fill the gaps the long one's gate math and h exchange
// leave.  Waves 0-3 then fetch the head's weight fragments straight into VGPRs while group 1 finishes;
// the feature rows never leave LDS and the head needs no second launch.
#include "gru_core.hpp"

namespace mctsq {

constexpr int FLD = FEAT + 4;    // feature tile row stride in LDS
constexpr int ALD1 = 128 + 4;    // hidden activation row stride

// Diagnostic build (-DMCTSQ_STAMPS): s_memtime stamps of the LAST workgroup, kept in LDS while the kernel runs
// (a global store per stamp queues behind the head-weight loads and distorts what it measures) and
// copied out at the end; read by tools/stamps.py.
#ifdef MCTSQ_STAMPS
__device__ long long g_stamps[4 * 64];
__device__ long long g_iter[192];
#define STAMP(k) do { if (blockIdx.x == gridDim.x - 1 && lane == 0) L.st[(k) * 16 + wave] = __builtin_amdgcn_s_memtime(); } while (0)
#define WSTAMP(win, k) do { if (blockIdx.x == gridDim.x - 1 && gtid == 0) L.wst[(win) * 16 + ((gtid != (int)threadIdx.x) ? 8 : 0) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#define WSTAMP(win, k) do {} while (0)
#endif

struct FusedLds {
    GruLds g[2];
    float feats[TILE * FLD];
    float act[2][TILE * ALD1];
    float qv[TILE * 16];
#ifdef MCTSQ_STAMPS
    long long st[64], wst[64];
#endif
};

template <bool SAVE>
__global__ __launch_bounds__(512) void gru3_head_fwd_kernel(GruFwdArgs ga, HeadArgs ha) {
    __shared__ __attribute__((aligned(16))) FusedLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef MCTSQ_EXP_SWAP
    const int grp = (wave >> 2) ^ 1, wq = wave & 3, gtid = tid & 255;
#else
    const int grp = wave >> 2, wq = wave & 3, gtid = tid & 255;
#endif
    const int jl = lane & 15, q = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * TILE;
    GruLds& G = L.g[grp];
    // group 0 takes the longest window; group 1 runs the other two back to back on the same SIMDs (issue
    // priorities were measured to make no difference: the two waves of a SIMD share it evenly)
    const int S0 = ga.slot[0].S, S1 = ga.slot[1].S, S2 = ga.slot[2].S;
    const int big = (S0 >= S1 && S0 >= S2) ? 0 : (S1 >= S2 ? 1 : 2);
    const int sa = grp == 0 ? big : (big == 0 ? 1 : 0);
    const int sb = big == 2 ? 1 : 2;

#ifdef MCTSQ_EXP_PRIO
    if (grp == 1) __builtin_amdgcn_s_setprio(MCTSQ_EXP_PRIO); else __builtin_amdgcn_s_setprio(0);
#endif
    STAMP(0);
    GruWave<SAVE> g;
    g.bind(lane, wq);
    // One copy of the window code serves both groups and both of group 1's windows (a loop, not two
    // inlined call sites): half the instruction footprint, and the second window runs from a warm
    // instruction cache.  The slot index is wave-uniform.
    const int nwin = grp == 0 ? 1 : 2;
    int si = __builtin_amdgcn_readfirstlane(sa);
    WSTAMP(0, 0);
    g.stage_issue(ga.slot[si], gtid, row0, ga.B);
    g.init_weights(ga.slot[si]);
    g.init_state();
    g.stage_commit(ga.slot[si], G, gtid, row0, ga.B, ga.Bp, true);
    WSTAMP(0, 1);
    __syncthreads();                                  // also publishes the zeroed arrival counters
    for (int wi = 0;; ++wi) {
        const GruSlot& s = ga.slot[si];
        WSTAMP(wi, 2);
        if (SAVE) g.save_window(s, G, gtid, row0, ga.Bp);
        g.run(s, G, row0, ga.Bp);
        WSTAMP(wi, 3);
        const bool more = wi + 1 < nwin;
        const float lnw = g.lnwj, lnb = g.lnbj;
        const int sn = __builtin_amdgcn_readfirstlane(sb);
        // the next window's elements and weight fragments start their trip now, under this window's
        // LayerNorm (init_weights touches none of its inputs)
        if (more) { g.stage_issue(ga.slot[sn], gtid, row0, ga.B); g.init_weights(ga.slot[sn]); }
        WSTAMP(wi, 5);
        g.fin0(G);
        WSTAMP(wi, 6);
        g.sync(G);
        WSTAMP(wi, 7);
        g.fin2(s, G, L.feats + s.out_col, FLD, row0, lnw, lnb);
        WSTAMP(wi, 4);
        if (wi == 0) STAMP(1);
        if (!more) break;
        g.sync(G);                                    // every wave is done with this window's LDS
        si = sn;
        g.init_state();
        g.stage_commit(ga.slot[si], G, gtid, row0, ga.B, ga.Bp, false);
        WSTAMP(1, 1);
        g.sync(G);
    }
    // The GRU registers are dead from here on.  Group 0 finishes first (25 steps against 10 + 3 and a
    // second staging): its waves fetch the head's weight fragments (two 16-column tiles of layers 0
    // and 1 each; wave 0 also the output layer) while group 1 is still stepping:
    // B[k = q*K4 + kk][col] = W[col][k].
    const int ht = wave;
    const bool head_wave = wave < 4;
    float4 w0f[2][12], w1f[2][8], w2f[8];
    if (head_wave) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const float4* p0 = reinterpret_cast<const float4*>(ha.w[0] + (int64_t)(16 * (ht + 4 * tt) + jl) * FEAT + q * 48);
            const float4* p1 = reinterpret_cast<const float4*>(ha.w[1] + (int64_t)(16 * (ht + 4 * tt) + jl) * 128 + q * 32);
#pragma unroll
            for (int c = 0; c < 12; ++c) w0f[tt][c] = p0[c];
#pragma unroll
            for (int c = 0; c < 8; ++c) w1f[tt][c] = p1[c];
        }
        if (wave == 0) {
            const float4* p2 = reinterpret_cast<const float4*>(ha.w[2] + (int64_t)(jl < ha.A ? jl : 0) * 128 + q * 32);
#pragma unroll
            for (int c = 0; c < 8; ++c) w2f[c] = jl < ha.A ? p2[c] : make_float4(0, 0, 0, 0);
        }
    }
    __syncthreads();                                  // all three feature blocks are in LDS
    STAMP(2);

    if (SAVE && ga.feats) {                           // training keeps the feature rows
        for (int idx = tid; idx < TILE * FEAT; idx += 512) {
            const int i = idx / FEAT, c = idx - i * FEAT;
            ga.feats[(row0 + i) * (int64_t)FEAT + c] = L.feats[i * FLD + c];
        }
    }

    // ---- head layer 0: [16 x 192] . W0^T, two column tiles per wave 0..3 ------------------
    if (head_wave) {
        const float4* xa = reinterpret_cast<const float4*>(&L.feats[(lane & 15) * FLD + q * 48]);
        f32x4 acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
#pragma unroll
        for (int c = 0; c < 12; c += 2) {
            const float4 a = xa[c], b = xa[c + 1];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                acc[tt][0] = mfma16(a.x, w0f[tt][c].x, acc[tt][0]); acc[tt][1] = mfma16(b.x, w0f[tt][c + 1].x, acc[tt][1]);
                acc[tt][0] = mfma16(a.y, w0f[tt][c].y, acc[tt][0]); acc[tt][1] = mfma16(b.y, w0f[tt][c + 1].y, acc[tt][1]);
                acc[tt][0] = mfma16(a.z, w0f[tt][c].z, acc[tt][0]); acc[tt][1] = mfma16(b.z, w0f[tt][c + 1].z, acc[tt][1]);
                acc[tt][0] = mfma16(a.w, w0f[tt][c].w, acc[tt][0]); acc[tt][1] = mfma16(b.w, w0f[tt][c + 1].w, acc[tt][1]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int col = 16 * (ht + 4 * tt) + jl;
            const float bias = ha.b[0][col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = apply_act((acc[tt][0][r] + acc[tt][1][r]) + bias, ha.act);
                L.act[0][(4 * q + r) * ALD1 + col] = v;
                if (SAVE) ha.sv_act[0][(row0 + 4 * q + r) * (int64_t)128 + col] = v;
            }
        }
    }
    __syncthreads();
    // ---- head layer 1: [16 x 128] . W1^T ---------------------------------------------------
    if (head_wave) {
        const float4* xa = reinterpret_cast<const float4*>(&L.act[0][(lane & 15) * ALD1 + q * 32]);
        f32x4 acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            const float4 a = xa[c], b = xa[c + 1];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                acc[tt][0] = mfma16(a.x, w1f[tt][c].x, acc[tt][0]); acc[tt][1] = mfma16(b.x, w1f[tt][c + 1].x, acc[tt][1]);
                acc[tt][0] = mfma16(a.y, w1f[tt][c].y, acc[tt][0]); acc[tt][1] = mfma16(b.y, w1f[tt][c + 1].y, acc[tt][1]);
                acc[tt][0] = mfma16(a.z, w1f[tt][c].z, acc[tt][0]); acc[tt][1] = mfma16(b.z, w1f[tt][c + 1].z, acc[tt][1]);
                acc[tt][0] = mfma16(a.w, w1f[tt][c].w, acc[tt][0]); acc[tt][1] = mfma16(b.w, w1f[tt][c + 1].w, acc[tt][1]);
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int col = 16 * (ht + 4 * tt) + jl;
            const float bias = ha.b[1][col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = apply_act((acc[tt][0][r] + acc[tt][1][r]) + bias, ha.act);
                L.act[1][(4 * q + r) * ALD1 + col] = v;
                if (SAVE) ha.sv_act[1][(row0 + 4 * q + r) * (int64_t)128 + col] = v;
            }
        }
    }
    __syncthreads();
    // ---- output layer: [16 x 128] . W2^T (A <= 16 columns), wave 0 -------------------------
    if (wave == 0) {
        const float4* xa = reinterpret_cast<const float4*>(&L.act[1][(lane & 15) * ALD1 + q * 32]);
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            const float4 a = xa[c], b = xa[c + 1];
            acc0 = mfma16(a.x, w2f[c].x, acc0); acc1 = mfma16(b.x, w2f[c + 1].x, acc1);
            acc0 = mfma16(a.y, w2f[c].y, acc0); acc1 = mfma16(b.y, w2f[c + 1].y, acc1);
            acc0 = mfma16(a.z, w2f[c].z, acc0); acc1 = mfma16(b.z, w2f[c + 1].z, acc1);
            acc0 = mfma16(a.w, w2f[c].w, acc0); acc1 = mfma16(b.w, w2f[c + 1].w, acc1);
        }
        const float bias = jl < ha.A ? ha.b[2][jl] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) L.qv[(4 * q + r) * 16 + jl] = (acc0[r] + acc1[r]) + bias;
    }
    __syncthreads();
    // ---- leaf epilogue: one lane per sample ---------------------------------------------
    if (tid < TILE && row0 + tid < ha.B) {
        const float* qv = &L.qv[tid * 16];
        const int64_t row = row0 + tid;
        float best = qv[0];
        int arg = 0;
        for (int k = 1; k < ha.A; ++k)
            if (qv[k] > best) { best = qv[k]; arg = k; }       // first maximum, like torch.argmax
        float den = 0.0f;
        for (int k = 0; k < ha.A; ++k) den += expf(qv[k] - best);
        for (int k = 0; k < ha.A; ++k) {
            ha.q[row * ha.A + k] = qv[k];
            if (ha.prior) ha.prior[row * ha.A + k] = expf(qv[k] - best) / den;
        }
        if (ha.vmax) ha.vmax[row] = best;
        if (ha.argmax) ha.argmax[row] = arg;
    }
    STAMP(3);
#ifdef MCTSQ_STAMPS
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1 && tid < 64) { g_stamps[tid] = L.st[tid]; g_iter[tid] = L.wst[tid]; }
#endif
}

#ifdef MCTSQ_STAMPS
extern "C" int mctsq_debug_stamps(long long* host_out) {
    (void)hipMemcpyFromSymbol(host_out + 64, HIP_SYMBOL(g_iter), sizeof(long long) * 192);
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(long long) * 4 * 16);
}
#endif

int check_gru_args(const GruFwdArgs& args);

bool fused_gru_head_applicable(const GruFwdArgs& ga, const HeadArgs& ha) {
    return ga.nslots == 3 && ha.n_fc == 3 && ha.in[0] == FEAT && ha.out[0] == 128 && ha.in[1] == 128 &&
           ha.out[1] == 128 && ha.in[2] == 128 && ha.A <= 16 && ga.slot[0].out_col == 0 &&
           ga.slot[1].out_col == E && ga.slot[2].out_col == 2 * E;
}

int launch_gru3_head_fwd(const GruFwdArgs& ga, const HeadArgs& ha, bool save, hipStream_t st) {
    if (ga.B == 0) return 0;
    if (check_gru_args(ga)) return -1;
    MCTSQ_CHECK(fused_gru_head_applicable(ga, ha), "fused GRU+head kernel needs gru/gru/gru and a 192-128-128-A head");
    dim3 grid((unsigned)(ga.Bp / TILE));
    if (save) hipLaunchKernelGGL(gru3_head_fwd_kernel<true>, grid, dim3(512), 0, st, ga, ha);
    else hipLaunchKernelGGL(gru3_head_fwd_kernel<false>, grid, dim3(512), 0, st, ga, ha);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
