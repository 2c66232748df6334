// The shipped default network (gru/gru/gru encoders, 192-128-128-A head; config_mctsq.yaml:24-42)
// as ONE kernel: TripleEncoderDQN.forward (dqn_per.py:216-236) + the leaf epilogue
// (mcts_per.py:293, 370).  One workgroup = one 16-sample tile = 8 waves (two per SIMD, 256 VGPRs each):
//   waves 0-3  the longest window (price, 25 steps), waves 4-7 the other two back to back (10 + 3).
// Each 4-wave group synchronises through its own LDS arrival counter (gru_core.hpp), not the
// workgroup barrier, so the shorter windows fill the gaps the long one's gate math and h exchange
// leave.  Waves 0-3 then fetch the head's weight fragments straight into VGPRs while group 1 finishes;
// the feature rows never leave LDS and the head needs no second launch.
#include <cstdlib>
#include "gru_core.hpp"
#include "head_core.hpp"

namespace mctsq {

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
    HeadLds head;
#ifdef MCTSQ_STAMPS
    long long st[64], wst[64];
#endif
};

// RPL = 4: workgroup `tile` = one 16-sample tile; RPL = 2: one 8-sample half tile (gru_core.hpp), chosen by the host for
// batches whose 16-sample tiles would leave half the CUs idle
template <bool SAVE, int RPL>
__device__ __forceinline__ void gru3_head_tile(FusedLds& L, const GruFwdArgs& ga, const HeadArgs& ha, int tile) {
    // the wave index (and everything derived from it: group, unit block, head tile) lives in scalar registers
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3, gtid = tid & 255;
    const int64_t row0 = (int64_t)tile * (4 * RPL);
    GruLds& G = L.g[grp];
    // group 0 takes the longest window; group 1 runs the other two back to back on the same SIMDs (issue
    // priorities were measured to make no difference: the two waves of a SIMD share it evenly)
    const int S0 = ga.slot[0].S, S1 = ga.slot[1].S, S2 = ga.slot[2].S;
    const int big = (S0 >= S1 && S0 >= S2) ? 0 : (S1 >= S2 ? 1 : 2);
    const int sa = grp == 0 ? big : (big == 0 ? 1 : 0);
    const int sb = big == 2 ? 1 : 2;

    STAMP(0);
    GruWave<SAVE, RPL> g;
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
        g.fin0(G, s, row0);
        g.sync(G);
        g.fin2_split(s, G, &L.head.fpl[0][0], TILE * FPL, FPL, row0, lnw, lnb, SAVE ? ga.feats : nullptr);
        WSTAMP(wi, 4);
        STAMP(1);
        g.sync(G);                                    // every wave is done with this window's LDS
        si = sn;
        g.init_state();
        g.stage_commit(ga.slot[si], G, gtid, row0, ga.B, ga.Bp, false);
        WSTAMP(1, 1);
        g.sync(G);
    }
    // The GRU weight registers are dead from here on: the head's fragments take their place and travel under the
    // last LayerNorm (head_core.hpp).
    HeadWave<SAVE, RPL> hd;
    hd.load(ha, lane, wave);
    {
        const GruSlot& s = ga.slot[si];
        WSTAMP(nwin - 1, 5);
        g.fin0(G, s, row0);
        g.sync(G);
        g.fin2_split(s, G, &L.head.fpl[0][0], TILE * FPL, FPL, row0, g.lnwj, g.lnbj, SAVE ? ga.feats : nullptr);
        WSTAMP(nwin - 1, 4);
        if (nwin == 1) STAMP(1);
    }
    __syncthreads();                                  // all three feature blocks are in LDS
    STAMP(2);

    hd.run(ha, L.head, lane, wave, row0);
    STAMP(3);
#ifdef MCTSQ_STAMPS
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1 && tid < 64) { g_stamps[tid] = L.st[tid]; g_iter[tid] = L.wst[tid]; }
#endif
}

template <bool SAVE, int RPL>
__global__ __launch_bounds__(512) void gru3_head_fwd_kernel(GruFwdArgs ga, HeadArgs ha) {
    __shared__ __attribute__((aligned(16))) FusedLds L;
    gru3_head_tile<SAVE, RPL>(L, ga, ha, blockIdx.x);
}

// The learner's two forwards in ONE launch: workgroups [0, n0) run the policy network on s with the activation records
// kept, the others the target network on s' (dqn_per.py:455-463).  Same code per tile as the two separate launches;
// what goes away is the second launch, the helper-stream hand-off around it and the cross-queue wait before the head
// backward.
template <int RPL>
__global__ __launch_bounds__(512) void gru3_head_fwd_pair_kernel(GruFwdArgs ga0, HeadArgs ha0, GruFwdArgs ga1, HeadArgs ha1,
                                                                 int n0) {
    __shared__ __attribute__((aligned(16))) FusedLds L;
    if ((int)blockIdx.x < n0) gru3_head_tile<true, RPL>(L, ga0, ha0, blockIdx.x);
    else gru3_head_tile<false, RPL>(L, ga1, ha1, (int)blockIdx.x - n0);
}

// 8-sample workgroups while they still fit the chip as one wave of workgroups (one per CU: 512 threads at 256 registers)
static bool half_tiles(int64_t tiles16) {
    static const int force = getenv("MCTSQ_FWD_RPL") ? atoi(getenv("MCTSQ_FWD_RPL")) : 0;   // 2 / 4: measurements and tests
    if (force == 2) return true;
    if (force == 4) return false;
    return 2 * tiles16 <= 256;
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
    MCTSQ_CHECK(ha.wprep, "fused GRU+head kernel needs the prepared head weights");
    const int64_t tiles = ga.Bp / TILE;
    if (half_tiles(tiles)) {
        const dim3 grid((unsigned)(2 * tiles));
        if (save) hipLaunchKernelGGL((gru3_head_fwd_kernel<true, 2>), grid, dim3(512), 0, st, ga, ha);
        else hipLaunchKernelGGL((gru3_head_fwd_kernel<false, 2>), grid, dim3(512), 0, st, ga, ha);
    } else {
        const dim3 grid((unsigned)tiles);
        if (save) hipLaunchKernelGGL((gru3_head_fwd_kernel<true, 4>), grid, dim3(512), 0, st, ga, ha);
        else hipLaunchKernelGGL((gru3_head_fwd_kernel<false, 4>), grid, dim3(512), 0, st, ga, ha);
    }
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_gru3_head_fwd_pair(const GruFwdArgs& ga0, const HeadArgs& ha0, const GruFwdArgs& ga1, const HeadArgs& ha1,
                              hipStream_t st) {
    if (check_gru_args(ga0) || check_gru_args(ga1)) return -1;
    MCTSQ_CHECK(fused_gru_head_applicable(ga0, ha0) && fused_gru_head_applicable(ga1, ha1) && ha0.wprep && ha1.wprep,
                "paired forward needs gru/gru/gru, a 192-128-128-A head and prepared weights");
    MCTSQ_CHECK(ga0.B > 0 && ga1.B > 0, "paired forward: empty batch");
    const int n0 = (int)(ga0.Bp / TILE), n1 = (int)(ga1.Bp / TILE);
    if (half_tiles(n0 + n1))
        hipLaunchKernelGGL(gru3_head_fwd_pair_kernel<2>, dim3((unsigned)(2 * (n0 + n1))), dim3(512), 0, st, ga0, ha0, ga1, ha1, 2 * n0);
    else
        hipLaunchKernelGGL(gru3_head_fwd_pair_kernel<4>, dim3((unsigned)(n0 + n1)), dim3(512), 0, st, ga0, ha0, ga1, ha1, n0);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
