// Stand-alone GRUAttentionEnc forward (dqn_per.py:58-77) for encoder mixes that are not all-GRU:
// one workgroup (4 waves) per (16-sample tile, GRU slot), features written to the [Bp,192] buffer
// the head kernel reads.  The all-GRU default goes through fused_gru_head.hip instead.
#include "gru_core.hpp"
#include "prepare_core.hpp"

namespace mctsq {

template <bool SAVE>
__global__ __launch_bounds__(256) void gru_enc_fwd_kernel(GruFwdArgs args) {
    // interleave slots over consecutive workgroups so every CU gets a mix of long and short windows
    const int slot = blockIdx.x % args.nslots;
    const int64_t row0 = (int64_t)(blockIdx.x / args.nslots) * TILE;
    const GruSlot& s = args.slot[slot];
    __shared__ __attribute__((aligned(16))) GruLds L;
    GruWave<SAVE> g;
    g.bind(threadIdx.x & 63, threadIdx.x >> 6);
    g.stage_issue(s, threadIdx.x, row0, args.B);
    g.init_weights(s);
    g.init_state();
    g.stage_commit(s, L, threadIdx.x, row0, args.B, args.Bp);
    __syncthreads();
    if (SAVE) g.save_window(s, L, threadIdx.x, row0, args.Bp);
    g.run(s, L, row0, args.Bp);
    g.fin0(L, s, row0);
    g.sync(L);
    g.fin2(s, L, args.feats + row0 * FEAT + s.out_col, FEAT, row0, g.lnwj, g.lnbj);
}

// one thread per (slot, wave, gate, K chunk, lane): 8 consecutive weights -> 3 x 8 bf16 (prepare_core.hpp)
struct GruPrepArgs { const float* whh[3]; u32x4* out; };
__global__ __launch_bounds__(256) void gru_prepare_kernel(GruPrepArgs a) {
    const int tid = blockIdx.x * 256 + threadIdx.x;           // grid = n * 6 blocks of 256
    const int slot = tid / 1536;
    prep_gru_fwd(a.whh[slot], a.out + (int64_t)slot * (GRU_PREP_BYTES / 16), tid - slot * 1536);
}

int launch_gru_prepare(const float* const whh[3], int n, void* prep, hipStream_t st) {
    if (n == 0) return 0;
    GruPrepArgs a;
    for (int i = 0; i < 3; ++i) a.whh[i] = whh[i < n ? i : 0];
    a.out = reinterpret_cast<u32x4*>(prep);
    hipLaunchKernelGGL(gru_prepare_kernel, dim3((unsigned)(n * 6)), dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int check_gru_args(const GruFwdArgs& args) {
    for (int i = 0; i < args.nslots; ++i)
        MCTSQ_CHECK(args.slot[i].wprep && ((uintptr_t)args.slot[i].wprep & 15) == 0, "GRU encoder needs prepared weights");
    for (int i = 0; i < args.nslots; ++i)
        MCTSQ_CHECK(args.slot[i].S * (args.slot[i].F > 4 ? 8 : 4) <= XLD && args.slot[i].F <= MAX_F,
                    "GRU encoder window %dx%d does not fit the staged tile", args.slot[i].S, args.slot[i].F);
    MCTSQ_CHECK(args.Bp % TILE == 0 && args.Bp >= args.B, "padded batch must be a multiple of %d", TILE);
    return 0;
}

int launch_gru_fwd(const GruFwdArgs& args, bool save, hipStream_t st) {
    if (args.nslots == 0 || args.B == 0) return 0;
    if (check_gru_args(args)) return -1;
    dim3 grid((unsigned)(args.Bp / TILE * args.nslots));
    if (save) hipLaunchKernelGGL(gru_enc_fwd_kernel<true>, grid, dim3(256), 0, st, args);
    else hipLaunchKernelGGL(gru_enc_fwd_kernel<false>, grid, dim3(256), 0, st, args);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
