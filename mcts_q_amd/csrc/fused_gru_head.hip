// The shipped default network (gru/gru/gru encoders, 192-128-128-A head; config_mctsq.yaml:24-42)
// as ONE kernel: TripleEncoderDQN.forward (dqn_per.py:216-236) + the leaf epilogue
// (mcts_per.py:293, 370).  One workgroup = one 16-sample tile = 12 waves:
//   waves 0-3  price GRU (S steps), waves 4-7 process GRU, waves 8-11 gas/EUA GRU,
// all stepping behind one barrier per step, so each SIMD carries one wave of every window length
// (balanced MFMA load per CU whatever the dispatcher does).  Waves whose window is finished fetch
// the head's weight fragments straight into VGPRs while the price recurrence is still running;
// the feature rows never leave LDS and the head needs no second launch.
#include "gru_core.hpp"

namespace mctsq {

constexpr int FLD = FEAT + 4;    // feature tile row stride in LDS
constexpr int ALD1 = 128 + 4;    // hidden activation row stride

#ifdef MCTSQ_STAMPS
__device__ long long g_stamps[4 * 64];
__device__ long long g_iter[64];
#define STAMP(k) do { if (blockIdx.x == 0 && lane == 0) g_stamps[(k) * 16 + wave] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

struct FusedLds {
    GruLds g[3];
    float feats[TILE * FLD];
    float act[2][TILE * ALD1];
    float qv[TILE * 16];
};

template <bool SAVE>
__global__ __launch_bounds__(768) void gru3_head_fwd_kernel(GruFwdArgs ga, HeadArgs ha) {
    __shared__ __attribute__((aligned(16))) FusedLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3, gtid = tid & 255;
    const int jl = lane & 15, q = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * TILE;
    const GruSlot& s = ga.slot[grp];
    GruLds& G = L.g[grp];
    const int S = s.S;
    const int Smax = max(ga.slot[0].S, max(ga.slot[1].S, ga.slot[2].S));

    STAMP(0);
    GruWave<SAVE> g;
    g.init(s, lane, wq);
    g.stage(s, G, gtid, row0, ga.B, ga.Bp);
    __syncthreads();
    STAMP(1);

    // head weight fragments (waves 4..11 own one 16-column tile of layers 0 and 1; wave 4 also the
    // output layer): B[k = q*K4 + kk][col] = W[col][k]
    // own window: S recurrence steps + 3 LayerNorm phases, one barrier each
    for (int it = 0; it <= S + 2; ++it) {
        const int ph = it - S;                        // uniform per wave group
        if (ph < 0) g.step(s, G, it, row0, ga.Bp);
        else if (ph == 0) { g.attend(s, G, S & 1, S - 1, row0, ga.Bp); g.fin0(G); }
        else if (ph == 1) g.fin1(G);
        else g.fin2(s, G, L.feats + s.out_col, FLD, row0);
#ifdef MCTSQ_STAMPS
        if (blockIdx.x == 0 && tid == 0) g_iter[it] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef MCTSQ_STAMPS
        if (blockIdx.x == 0 && tid == 0) g_iter[32 + it] = __builtin_amdgcn_s_memtime();
#endif
    }
    // The GRU registers are dead from here on.  Waves 4..11 fetch their head weight fragments
    // (one 16-column tile of layers 0 and 1 each; wave 4 also the output layer) while the longer
    // windows are still stepping: B[k = q*K4 + kk][col] = W[col][k].
    const int ht = wave - 4;
    float4 w0f[12], w1f[8], w2f[8];
    if (wave >= 4) {
        const float4* p0 = reinterpret_cast<const float4*>(ha.w[0] + (int64_t)(16 * ht + jl) * FEAT + q * 48);
        const float4* p1 = reinterpret_cast<const float4*>(ha.w[1] + (int64_t)(16 * ht + jl) * 128 + q * 32);
#pragma unroll
        for (int c = 0; c < 12; ++c) w0f[c] = p0[c];
#pragma unroll
        for (int c = 0; c < 8; ++c) w1f[c] = p1[c];
        if (wave == 4) {
            const float4* p2 = reinterpret_cast<const float4*>(ha.w[2] + (int64_t)(jl < ha.A ? jl : 0) * 128 + q * 32);
#pragma unroll
            for (int c = 0; c < 8; ++c) w2f[c] = jl < ha.A ? p2[c] : make_float4(0, 0, 0, 0);
        }
    }
    // every wave executes Smax + 3 barriers in total
    for (int it = S + 3; it <= Smax + 2; ++it) __syncthreads();
    STAMP(2);

    if (SAVE && ga.feats) {                           // training keeps the feature rows
        for (int idx = tid; idx < TILE * FEAT; idx += 768) {
            const int i = idx / FEAT, c = idx - i * FEAT;
            ga.feats[(row0 + i) * (int64_t)FEAT + c] = L.feats[i * FLD + c];
        }
    }

    // ---- head layer 0: [16 x 192] . W0^T, one column tile per wave 4..11 ----------------
    if (wave >= 4) {
        const float4* xa = reinterpret_cast<const float4*>(&L.feats[(lane & 15) * FLD + q * 48]);
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 12; c += 2) {
            const float4 a = xa[c], b = xa[c + 1];
            acc0 = mfma16(a.x, w0f[c].x, acc0); acc1 = mfma16(b.x, w0f[c + 1].x, acc1);
            acc0 = mfma16(a.y, w0f[c].y, acc0); acc1 = mfma16(b.y, w0f[c + 1].y, acc1);
            acc0 = mfma16(a.z, w0f[c].z, acc0); acc1 = mfma16(b.z, w0f[c + 1].z, acc1);
            acc0 = mfma16(a.w, w0f[c].w, acc0); acc1 = mfma16(b.w, w0f[c + 1].w, acc1);
        }
        const int col = 16 * ht + jl;
        const float bias = ha.b[0][col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = apply_act((acc0[r] + acc1[r]) + bias, ha.act);
            L.act[0][(4 * q + r) * ALD1 + col] = v;
            if (SAVE) ha.sv_act[0][(row0 + 4 * q + r) * (int64_t)128 + col] = v;
        }
    }
    __syncthreads();
    // ---- head layer 1: [16 x 128] . W1^T ---------------------------------------------------
    if (wave >= 4) {
        const float4* xa = reinterpret_cast<const float4*>(&L.act[0][(lane & 15) * ALD1 + q * 32]);
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            const float4 a = xa[c], b = xa[c + 1];
            acc0 = mfma16(a.x, w1f[c].x, acc0); acc1 = mfma16(b.x, w1f[c + 1].x, acc1);
            acc0 = mfma16(a.y, w1f[c].y, acc0); acc1 = mfma16(b.y, w1f[c + 1].y, acc1);
            acc0 = mfma16(a.z, w1f[c].z, acc0); acc1 = mfma16(b.z, w1f[c + 1].z, acc1);
            acc0 = mfma16(a.w, w1f[c].w, acc0); acc1 = mfma16(b.w, w1f[c + 1].w, acc1);
        }
        const int col = 16 * ht + jl;
        const float bias = ha.b[1][col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = apply_act((acc0[r] + acc1[r]) + bias, ha.act);
            L.act[1][(4 * q + r) * ALD1 + col] = v;
            if (SAVE) ha.sv_act[1][(row0 + 4 * q + r) * (int64_t)128 + col] = v;
        }
    }
    __syncthreads();
    // ---- output layer: [16 x 128] . W2^T (A <= 16 columns), wave 4 -------------------------
    if (wave == 4) {
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
}

#ifdef MCTSQ_STAMPS
extern "C" int mctsq_debug_stamps(long long* host_out) {
    hipMemcpyFromSymbol(host_out + 64, HIP_SYMBOL(g_iter), sizeof(long long) * 64);
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
    if (save) hipLaunchKernelGGL(gru3_head_fwd_kernel<true>, grid, dim3(768), 0, st, ga, ha);
    else hipLaunchKernelGGL(gru3_head_fwd_kernel<false>, grid, dim3(768), 0, st, ga, ha);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
