// Q head forward: fc_layers of TripleEncoderDQN (dqn_per.py:205-214, 235-236) fused with the
// leaf epilogue the callers apply (mcts_per.py:293 softmax prior, :370 max; argmax for greedy
// parity).  One workgroup = 16 samples; every Linear is a [16 x K] . [K x N] product on
// v_mfma_f32_16x16x4_f32 with the activations ping-ponged through LDS and the weight fragments
// streamed from L2 (the whole head is 164 KB).
#include "device_utils.hpp"
#include "kernels.hpp"
#include "head_core.hpp"
#include "prepare_core.hpp"

namespace mctsq {

constexpr int ALD = 260;   // LDS row stride (floats): max width 256 + 4
constexpr int WL_FLOATS = 128 * 196;   // staged weight matrix, rows padded by 4 floats (25,088 floats)

template <bool SAVE>
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jl = lane & 15, q = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * TILE;
    __shared__ __attribute__((aligned(16))) float buf[2][TILE * ALD];
    __shared__ __attribute__((aligned(16))) float wl[WL_FLOATS];

    for (int idx = tid; idx < TILE * (FEAT / 4); idx += 256) {      // feats tile, float4 copies
        const int i = idx / (FEAT / 4), c = idx - i * (FEAT / 4);
        const float4 v = reinterpret_cast<const float4*>(a.feats + (row0 + i) * (int64_t)FEAT)[c];
        *reinterpret_cast<float4*>(&buf[0][i * ALD + 4 * c]) = v;
    }
    __syncthreads();

    for (int l = 0; l < a.n_fc; ++l) {
        const int K = a.in[l], N = a.out[l], K4 = K >> 2, KP = K + 4, kq = K >> 2;
        const float* src = buf[l & 1];
        float* dst = buf[(l + 1) & 1];
        const bool last = l == a.n_fc - 1;
        const float* xa = src + (lane & 15) * ALD + q * K4;
        // W[N][K] goes through LDS in chunks of NC output columns, staged with coalesced 16-byte
        // loads (per-lane fragment reads from global would touch 64 cache lines per instruction);
        // rows are padded to K+4 floats so the ds_read_b128 fragment reads are at most 2-way.
        const int NC = min((WL_FLOATS / KP) & ~15, (N + 15) & ~15);
        for (int n0 = 0; n0 < N; n0 += NC) {
            const int nrows = min(NC, N - n0);
            {   // 8 independent 16-byte loads in flight per lane before the first LDS write
                const float4* gsrc = reinterpret_cast<const float4*>(a.w[l] + (int64_t)n0 * K);
                const int total = nrows * kq;
                for (int base = 0; base < total; base += 256 * 8) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int idx = base + u * 256 + tid;
                        v[u] = idx < total ? gsrc[idx] : make_float4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int idx = base + u * 256 + tid;
                        if (idx < total) {
                            const int rj = idx / kq, c = idx - rj * kq;
                            *reinterpret_cast<float4*>(&wl[rj * KP + 4 * c]) = v[u];
                        }
                    }
                }
            }
            __syncthreads();
            const int nt = (nrows + 15) >> 4;
            for (int t0 = w; t0 < nt; t0 += 8) {
                const int jA = t0 * 16 + jl, jB = (t0 + 4) * 16 + jl;       // rows inside the chunk
                const bool hasB = (t0 + 4) < nt;                           // wave-uniform
                const bool okA = jA < nrows, okB = hasB && jB < nrows;
                const float* wa = wl + (okA ? jA : 0) * KP + q * K4;
                const float* wb = wl + (okB ? jB : 0) * KP + q * K4;
                f32x4 accA = {0, 0, 0, 0}, accB = {0, 0, 0, 0};
                if (hasB) {
#pragma unroll 2
                    for (int kk = 0; kk < K4; kk += 4) {
                        const float4 av = *reinterpret_cast<const float4*>(xa + kk);
                        float4 b1 = *reinterpret_cast<const float4*>(wa + kk);
                        float4 b2 = *reinterpret_cast<const float4*>(wb + kk);
                        if (!okA) b1 = make_float4(0, 0, 0, 0);
                        if (!okB) b2 = make_float4(0, 0, 0, 0);
                        accA = mfma16(av.x, b1.x, accA); accB = mfma16(av.x, b2.x, accB);
                        accA = mfma16(av.y, b1.y, accA); accB = mfma16(av.y, b2.y, accB);
                        accA = mfma16(av.z, b1.z, accA); accB = mfma16(av.z, b2.z, accB);
                        accA = mfma16(av.w, b1.w, accA); accB = mfma16(av.w, b2.w, accB);
                    }
                } else {
#pragma unroll 2
                    for (int kk = 0; kk < K4; kk += 4) {
                        const float4 av = *reinterpret_cast<const float4*>(xa + kk);
                        float4 b1 = *reinterpret_cast<const float4*>(wa + kk);
                        if (!okA) b1 = make_float4(0, 0, 0, 0);
                        accA = mfma16_x4(av, b1, accA);
                    }
                }
                if (okA) {
                    const int jg = n0 + jA;
                    const float bias = a.b[l][jg];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = accA[r] + bias;
                        if (!last) v = apply_act(v, a.act);
                        dst[(4 * q + r) * ALD + jg] = v;
                        if (SAVE && !last) a.sv_act[l][(row0 + 4 * q + r) * (int64_t)N + jg] = v;
                    }
                }
                if (okB) {
                    const int jg = n0 + jB;
                    const float bias = a.b[l][jg];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = accB[r] + bias;
                        if (!last) v = apply_act(v, a.act);
                        dst[(4 * q + r) * ALD + jg] = v;
                        if (SAVE && !last) a.sv_act[l][(row0 + 4 * q + r) * (int64_t)N + jg] = v;
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: one lane per sample ------------------------------------------------
    if (tid < TILE && row0 + tid < a.B) {
        const float* qv = buf[a.n_fc & 1] + tid * ALD;
        const int64_t row = row0 + tid;
        float best = qv[0];
        int arg = 0;
        for (int k = 1; k < a.A; ++k)
            if (qv[k] > best) { best = qv[k]; arg = k; }       // first maximum, like torch.argmax
        float den = 0.0f;
        for (int k = 0; k < a.A; ++k) den += expf(qv[k] - best);
        for (int k = 0; k < a.A; ++k) {
            a.q[row * a.A + k] = qv[k];
            if (a.prior) a.prior[row * a.A + k] = expf(qv[k] - best) / den;
        }
        if (a.vmax) a.vmax[row] = best;
        if (a.argmax) a.argmax[row] = arg;
    }
}

// Head weights -> split-bf16 B fragments in the order the fused kernel loads them: lane (jl, q) of fragment
// (tile, chunk c) holds W[16 tile + jl][32c + 8q .. +7]; rows >= A of the output layer are zero.
__global__ __launch_bounds__(256) void head_prepare_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, int A, u32x4* __restrict__ out) {
    prep_head_fwd(w0, w1, w2, A, out, blockIdx.x * 256 + threadIdx.x);
}
int launch_head_prepare(const float* w0, const float* w1, const float* w2, int A, void* prep, hipStream_t st) {
    MCTSQ_CHECK(w0 && w1 && w2 && prep && A >= 1 && A <= 16, "head prepare: bad arguments");
    hipLaunchKernelGGL(head_prepare_kernel, dim3((84 * 64 + 255) / 256), dim3(256), 0, st, w0, w1, w2, A,
                       reinterpret_cast<u32x4*>(prep));
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

// ---- the 192-128-128-A head on split-bf16 MFMA (head_core.hpp) for feature rows that are in global memory:
// every encoder mix other than gru/gru/gru (whose fused kernel runs the same code on features still in LDS)
template <bool SAVE>
__global__ __launch_bounds__(512) void head_bf16_fwd_kernel(HeadArgs ha) {
    __shared__ __attribute__((aligned(16))) HeadLds L;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t row0 = (int64_t)blockIdx.x * TILE;
    HeadWave<SAVE> hd;
    hd.load(ha, lane, wave);
    for (int idx = tid; idx < TILE * FEAT; idx += 512) {
        const int i = idx / FEAT, c = idx - i * FEAT;
        unsigned u0, u1, u2;
        split3(ha.feats[(row0 + i) * (int64_t)FEAT + c], u0, u1, u2);
        L.fpl[0][i * FPL + c] = (unsigned short)(u0 >> 16);
        L.fpl[1][i * FPL + c] = (unsigned short)(u1 >> 16);
        L.fpl[2][i * FPL + c] = (unsigned short)(u2 >> 16);
    }
    __syncthreads();
    hd.run(ha, L, lane, wave, row0);
}

bool head_bf16_applicable(const HeadArgs& ha) {
    return ha.wprep && ha.n_fc == 3 && ha.in[0] == FEAT && ha.out[0] == 128 && ha.in[1] == 128 && ha.out[1] == 128 &&
           ha.in[2] == 128 && ha.A <= 16;
}

int launch_head_fwd(const HeadArgs& args, hipStream_t st) {
    if (args.B == 0) return 0;
    for (int l = 0; l < args.n_fc; ++l) {
        MCTSQ_CHECK(args.in[l] % 16 == 0 && args.in[l] <= 256, "head layer %d: K=%d unsupported", l, args.in[l]);
        MCTSQ_CHECK(args.out[l] <= 256, "head layer %d: N=%d unsupported", l, args.out[l]);
    }
    const bool save = args.sv_act[0] != nullptr;
    dim3 grid((unsigned)(args.Bp / TILE));
    if (head_bf16_applicable(args)) {
        if (save) hipLaunchKernelGGL(head_bf16_fwd_kernel<true>, grid, dim3(512), 0, st, args);
        else hipLaunchKernelGGL(head_bf16_fwd_kernel<false>, grid, dim3(512), 0, st, args);
        MCTSQ_HIP(hipGetLastError());
        return 0;
    }
    if (save) hipLaunchKernelGGL(head_fwd_kernel<true>, grid, dim3(256), 0, st, args);
    else hipLaunchKernelGGL(head_fwd_kernel<false>, grid, dim3(256), 0, st, args);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
