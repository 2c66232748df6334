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

constexpr int FPL = FEAT + 8;    // row stride (bf16 elements) of a feature plane: 400 B, b128 reads conflict-free
constexpr int APL = 128 + 8;     // same for a hidden-activation plane (272 B)
// prepared head weights (head_prepare_kernel), u32x4 fragments of 64 lanes each:
//   layer 0 [tile 8][K chunk 6][plane 3], layer 1 [tile 8][K chunk 4][plane 3], output layer [K chunk 4][plane 3]
constexpr int H0_FR = 8 * 6 * 3, H1_FR = 8 * 4 * 3, H2_FR = 4 * 3;
static_assert(HEAD_PREP_BYTES == (int64_t)(H0_FR + H1_FR + H2_FR) * 64 * 16, "prepared head size");

// six cross terms of the split product, smallest first
#define HEAD_MM6(ACC, XH, XM, XL, W)                                                               \
    ACC = mfma_bf(XL, (W)[0], ACC); ACC = mfma_bf(XH, (W)[2], ACC); ACC = mfma_bf(XM, (W)[1], ACC); \
    ACC = mfma_bf(XM, (W)[0], ACC); ACC = mfma_bf(XH, (W)[1], ACC); ACC = mfma_bf(XH, (W)[0], ACC);

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
    unsigned short fpl[3][TILE * FPL];       // the 192 feature columns as bf16 planes (hi, mid, lo)
    unsigned short apl[2][3][TILE * APL];    // hidden layers 0 and 1, same form
#ifdef MCTSQ_STAMPS
    long long st[64], wst[64];
#endif
};

template <bool SAVE>
__global__ __launch_bounds__(512) void gru3_head_fwd_kernel(GruFwdArgs ga, HeadArgs ha) {
    __shared__ __attribute__((aligned(16))) FusedLds L;
    // the wave index (and everything derived from it: group, unit block, head tile) lives in scalar registers
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3, gtid = tid & 255;
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
        if (wi + 1 >= nwin) break;
        const float lnw = g.lnwj, lnb = g.lnbj;
        const int sn = __builtin_amdgcn_readfirstlane(sb);
        // the next window's elements and weight fragments start their trip now, under this window's
        // LayerNorm (init_weights touches none of its inputs)
        g.stage_issue(ga.slot[sn], gtid, row0, ga.B);
        g.init_weights(ga.slot[sn]);
        g.fin0(G);
        g.sync(G);
        g.fin2_split(s, G, &L.fpl[0][0], TILE * FPL, FPL, row0, lnw, lnb, SAVE ? ga.feats : nullptr);
        WSTAMP(wi, 4);
        STAMP(1);
        g.sync(G);                                    // every wave is done with this window's LDS
        si = sn;
        g.init_state();
        g.stage_commit(ga.slot[si], G, gtid, row0, ga.B, ga.Bp, false);
        WSTAMP(1, 1);
        g.sync(G);
    }
    // The GRU weight registers are dead from here on: the head's fragments (prepared, lane-major, so every
    // load instruction reads 1 KB in a row) take their place and travel under the last LayerNorm.
    //   waves 0-3: layer 0, column tiles w and w + 4 (2 x 6 chunks x 3 planes), and the output layer (4 x 3)
    //   waves 4-7: layer 1, column tiles w - 4 and w (2 x 4 x 3), needed only after layer 0 is through
    const u32x4* hp = reinterpret_cast<const u32x4*>(ha.wprep) + lane;
    const int ht = wave & 3;
    u32x4 hw[2][18], hw2[12];
    float hb[2], hb2 = 0.0f;
    if (wave < 4) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int f = 0; f < 18; ++f) hw[tt][f] = hp[((ht + 4 * tt) * 18 + f) * 64];
            hb[tt] = ha.b[0][16 * (ht + 4 * tt) + jl];
        }
    } else {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int f = 0; f < 12; ++f) hw[tt][f] = hp[(H0_FR + (ht + 4 * tt) * 12 + f) * 64];
            hb[tt] = ha.b[1][16 * (ht + 4 * tt) + jl];
        }
    }
    {
        const GruSlot& s = ga.slot[si];
        WSTAMP(nwin - 1, 5);
        g.fin0(G);
        g.sync(G);
        g.fin2_split(s, G, &L.fpl[0][0], TILE * FPL, FPL, row0, g.lnwj, g.lnbj, SAVE ? ga.feats : nullptr);
        WSTAMP(nwin - 1, 4);
        if (nwin == 1) STAMP(1);
    }
    __syncthreads();                                  // all three feature blocks are in LDS
    STAMP(2);

    // ---- head layer 0: [16 x 192] . W0^T on waves 0-3, two column tiles each ----------------
    // All three layers run on the bf16 matrix instruction over the exact three-way split of both operands
    // (six cross terms, fp32 accumulate on top of the bias); activations are split by the lanes that produce them.
    if (wave < 4) {
        f32x4 acc[2][2] = {{{hb[0], hb[0], hb[0], hb[0]}, {0, 0, 0, 0}}, {{hb[1], hb[1], hb[1], hb[1]}, {0, 0, 0, 0}}};
        const int o = (lane & 15) * FPL + 8 * q;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const u32x4 xh = *reinterpret_cast<const u32x4*>(&L.fpl[0][o + 32 * c]);
            const u32x4 xm = *reinterpret_cast<const u32x4*>(&L.fpl[1][o + 32 * c]);
            const u32x4 xl = *reinterpret_cast<const u32x4*>(&L.fpl[2][o + 32 * c]);
            HEAD_MM6(acc[0][c & 1], xh, xm, xl, &hw[0][3 * c])
            HEAD_MM6(acc[1][c & 1], xh, xm, xl, &hw[1][3 * c])
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int col = 16 * (ht + 4 * tt) + jl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = apply_act(acc[tt][0][r] + acc[tt][1][r], ha.act);
                unsigned u0, u1, u2;
                split3(v, u0, u1, u2);
                L.apl[0][0][(4 * q + r) * APL + col] = (unsigned short)(u0 >> 16);
                L.apl[0][1][(4 * q + r) * APL + col] = (unsigned short)(u1 >> 16);
                L.apl[0][2][(4 * q + r) * APL + col] = (unsigned short)(u2 >> 16);
                if (SAVE) ha.sv_act[0][(row0 + 4 * q + r) * (int64_t)128 + col] = v;
            }
        }
        // layer 0's fragments are dead: the output layer's take their registers and arrive while waves 4-7 run layer 1
#pragma unroll
        for (int f = 0; f < 12; ++f) hw2[f] = hp[(H0_FR + H1_FR + f) * 64];
        hb2 = jl < ha.A ? ha.b[2][jl] : 0.0f;
    }
    __syncthreads();
    // ---- head layer 1: [16 x 128] . W1^T on waves 4-7 ----------------------------------------
    if (wave >= 4) {
        f32x4 acc[2][2] = {{{hb[0], hb[0], hb[0], hb[0]}, {0, 0, 0, 0}}, {{hb[1], hb[1], hb[1], hb[1]}, {0, 0, 0, 0}}};
        const int o = (lane & 15) * APL + 8 * q;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4 xh = *reinterpret_cast<const u32x4*>(&L.apl[0][0][o + 32 * c]);
            const u32x4 xm = *reinterpret_cast<const u32x4*>(&L.apl[0][1][o + 32 * c]);
            const u32x4 xl = *reinterpret_cast<const u32x4*>(&L.apl[0][2][o + 32 * c]);
            HEAD_MM6(acc[0][c & 1], xh, xm, xl, &hw[0][3 * c])
            HEAD_MM6(acc[1][c & 1], xh, xm, xl, &hw[1][3 * c])
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int col = 16 * (ht + 4 * tt) + jl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = apply_act(acc[tt][0][r] + acc[tt][1][r], ha.act);
                unsigned u0, u1, u2;
                split3(v, u0, u1, u2);
                L.apl[1][0][(4 * q + r) * APL + col] = (unsigned short)(u0 >> 16);
                L.apl[1][1][(4 * q + r) * APL + col] = (unsigned short)(u1 >> 16);
                L.apl[1][2][(4 * q + r) * APL + col] = (unsigned short)(u2 >> 16);
                if (SAVE) ha.sv_act[1][(row0 + 4 * q + r) * (int64_t)128 + col] = v;
            }
        }
    }
    __syncthreads();
    // ---- output layer: [16 x 128] . W2^T (A <= 16 columns).  Waves 0-3 all form the full product (24 matrix
    // instructions, side by side on the four SIMDs) and each finishes one of the lane's four rows: the
    // epilogue is a serial dependency chain, so four short ones beat one long one.
    if (wave < 4) {
        f32x4 acc0 = {hb2, hb2, hb2, hb2}, acc1 = {0, 0, 0, 0};
        const int o = (lane & 15) * APL + 8 * q;
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            const u32x4 xh = *reinterpret_cast<const u32x4*>(&L.apl[1][0][o + 32 * c]);
            const u32x4 xm = *reinterpret_cast<const u32x4*>(&L.apl[1][1][o + 32 * c]);
            const u32x4 xl = *reinterpret_cast<const u32x4*>(&L.apl[1][2][o + 32 * c]);
            const u32x4 yh = *reinterpret_cast<const u32x4*>(&L.apl[1][0][o + 32 * c + 32]);
            const u32x4 ym = *reinterpret_cast<const u32x4*>(&L.apl[1][1][o + 32 * c + 32]);
            const u32x4 yl = *reinterpret_cast<const u32x4*>(&L.apl[1][2][o + 32 * c + 32]);
            HEAD_MM6(acc0, xh, xm, xl, &hw2[3 * c])
            HEAD_MM6(acc1, yh, ym, yl, &hw2[3 * c + 3])
        }
        // ---- leaf epilogue in registers: lane (jl, q) holds Q[row 4q + r][action jl]; max / first argmax /
        // softmax (mcts_per.py:293, 370) are reductions over the 16 lanes of a row group
        // (lane position re-derived here: values computed at kernel entry would sit in registers across every loop)
        const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), jl = ln & 15, q = ln >> 4;
        const bool colok = jl < ha.A;
        {
            const int r = ht;
            const int64_t row = row0 + 4 * q + r;
            const float v = r == 0 ? acc0[0] + acc1[0] : r == 1 ? acc0[1] + acc1[1] : r == 2 ? acc0[2] + acc1[2] : acc0[3] + acc1[3];
            const float best = row16_max_dpp(colok ? v : -INFINITY);
            const float first = row16_min_dpp((colok && v == best) ? (float)jl : 99.0f);   // like torch.argmax
            const float e = colok ? expf(v - best) : 0.0f;
            const float den = row16_sum(e);
            if (row < ha.B) {
                if (colok) {
                    ha.q[row * ha.A + jl] = v;
                    if (ha.prior) ha.prior[row * ha.A + jl] = e / den;
                }
                if (jl == 0) {
                    if (ha.vmax) ha.vmax[row] = best;
                    if (ha.argmax) ha.argmax[row] = (int)first;
                }
            }
        }
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

// Head weights -> split-bf16 B fragments in the order the fused kernel loads them: lane (jl, q) of fragment
// (tile, chunk c) holds W[16 tile + jl][32c + 8q .. +7]; rows >= A of the output layer are zero.
__global__ __launch_bounds__(256) void head_prepare_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, int A, u32x4* __restrict__ out) {
    const int tid = blockIdx.x * 256 + threadIdx.x, lane = tid & 63, fg = tid >> 6;
    if (fg >= 48 + 32 + 4) return;
    const int jl = lane & 15, q = lane >> 4;
    const float* w;
    int K, tile, c, base;
    bool valid = true;
    if (fg < 48) { w = w0; K = FEAT; tile = fg / 6; c = fg % 6; base = fg * 3; }
    else if (fg < 80) { const int f = fg - 48; w = w1; K = 128; tile = f / 4; c = f % 4; base = H0_FR + f * 3; }
    else { const int f = fg - 80; w = w2; K = 128; tile = 0; c = f; base = H0_FR + H1_FR + f * 3; valid = jl < A; }
    float4 a = make_float4(0, 0, 0, 0), b = a;
    if (valid) {
        const float4* p = reinterpret_cast<const float4*>(w + (int64_t)(16 * tile + jl) * K + 32 * c + 8 * q);
        a = p[0]; b = p[1];
    }
    u32x4 hi, mid, lo;
    split8(a, b, hi, mid, lo);
    out[(base + 0) * 64 + lane] = hi; out[(base + 1) * 64 + lane] = mid; out[(base + 2) * 64 + lane] = lo;
}
int launch_head_prepare(const float* w0, const float* w1, const float* w2, int A, void* prep, hipStream_t st) {
    MCTSQ_CHECK(w0 && w1 && w2 && prep && A >= 1 && A <= 16, "head prepare: bad arguments");
    hipLaunchKernelGGL(head_prepare_kernel, dim3((84 * 64 + 255) / 256), dim3(256), 0, st, w0, w1, w2, A,
                       reinterpret_cast<u32x4*>(prep));
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

bool fused_gru_head_applicable(const GruFwdArgs& ga, const HeadArgs& ha) {
    return ga.nslots == 3 && ha.n_fc == 3 && ha.in[0] == FEAT && ha.out[0] == 128 && ha.in[1] == 128 &&
           ha.out[1] == 128 && ha.in[2] == 128 && ha.A <= 16 && ga.slot[0].out_col == 0 &&
           ga.slot[1].out_col == E && ga.slot[2].out_col == 2 * E;
}

int launch_gru3_head_fwd(const GruFwdArgs& ga, const HeadArgs& ha, bool save, hipStream_t st) {
    if (ga.B == 0) return 0;
    if (check_gru_args(ga)) return -1;
    MCTSQ_CHECK(fused_gru_head_applicable(ga, ha), "fused GRU+head kernel needs gru/gru/gru and a 192-128-128-A head");
    MCTSQ_CHECK(ha.wprep, "fused GRU+head kernel needs the prepared head weights");
    dim3 grid((unsigned)(ga.Bp / TILE));
    if (save) hipLaunchKernelGGL(gru3_head_fwd_kernel<true>, grid, dim3(512), 0, st, ga, ha);
    else hipLaunchKernelGGL(gru3_head_fwd_kernel<false>, grid, dim3(512), 0, st, ga, ha);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
