// GasEUAEncoder "mlp" (dqn_per.py:108-114, 128-130): flatten [B,3,F] -> Linear(3F,64) -> ReLU ->
// Linear(64,64) -> ReLU (no LayerNorm), forward and backward.  One workgroup = 16 samples; wave wq
// owns output units 16wq..16wq+15; both layers are MFMA products with the weight fragments in VGPRs.
#include "device_utils.hpp"
#include "kernels.hpp"
#include "learner.hpp"

namespace mctsq {

constexpr int MXLD = 20;      // flattened window row stride (<= 3*4 = 12 values, padded to 16)
constexpr int MHLD = 68;

template <bool SAVE>
__global__ __launch_bounds__(256) void mlp_enc_fwd_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[TILE * MXLD];
    __shared__ __attribute__((aligned(16))) float hs[TILE * MHLD];
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6, jl = lane & 15, q = lane >> 4;
    const int j = 16 * wq + jl;
    const int64_t row0 = (int64_t)blockIdx.x * TILE;
    const int F = a.F, Fd = a.Fd, K0 = 3 * F;
    {   // one element per thread: 16 samples x 16 padded columns
        const int i = tid >> 4, k = tid & 15;
        float v = 0.0f;
        if (k < K0 && row0 + i < a.B) {
            const int t = k / F, f = k - t * F;
            v = f < Fd ? a.x[(row0 + i) * (int64_t)(3 * Fd) + t * Fd + f] : time_feature(a.tf, t, f - Fd);
        }
        xs[i * MXLD + k] = v;
        if (SAVE) a.sv_x[(row0 + i) * 16 + k] = v;
    }
    float w0f[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = 4 * kk + q;
        w0f[kk] = k < K0 ? a.w0[(int64_t)j * K0 + k] : 0.0f;
    }
    float w1f[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) w1f[e] = a.w1[(int64_t)j * E + 16 * q + e];
    const float b0 = a.b0[j], b1 = a.b1[j];
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = mfma16(xs[(lane & 15) * MXLD + 4 * kk + q], w0f[kk], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = fmaxf(acc[r] + b0, 0.0f);
        hs[(4 * q + r) * MHLD + j] = v;
        if (SAVE) a.sv_m0[(row0 + 4 * q + r) * E + j] = v;
    }
    __syncthreads();
    const float4* ha = reinterpret_cast<const float4*>(&hs[(lane & 15) * MHLD + 16 * q]);
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
        const float4 u = ha[c], v = ha[c + 1];
        c0 = mfma16(u.x, w1f[4 * c + 0], c0); c1 = mfma16(v.x, w1f[4 * c + 4], c1);
        c0 = mfma16(u.y, w1f[4 * c + 1], c0); c1 = mfma16(v.y, w1f[4 * c + 5], c1);
        c0 = mfma16(u.z, w1f[4 * c + 2], c0); c1 = mfma16(v.z, w1f[4 * c + 6], c1);
        c0 = mfma16(u.w, w1f[4 * c + 3], c0); c1 = mfma16(v.w, w1f[4 * c + 7], c1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        a.feats[(row0 + 4 * q + r) * FEAT + a.out_col + j] = fmaxf((c0[r] + c1[r]) + b1, 0.0f);
}

int launch_mlp_fwd(const MlpArgs& a, bool save, hipStream_t st) {
    if (a.B == 0) return 0;
    MCTSQ_CHECK(3 * a.F <= 16, "gas/EUA mlp encoder: input width %d > 16", 3 * a.F);
    dim3 grid((unsigned)(a.Bp / TILE));
    if (save) hipLaunchKernelGGL(mlp_enc_fwd_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(mlp_enc_fwd_kernel<false>, grid, dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

// backward: delta1 = ReLU'(out) * dfeat; delta0 = ReLU'(m0) * (delta1 . W1)
__global__ __launch_bounds__(256) void mlp_enc_bwd_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(16))) float ds[TILE * MHLD];
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6, jl = lane & 15, q = lane >> 4;
    const int j = 16 * wq + jl;
    const int64_t row0 = (int64_t)blockIdx.x * TILE;
    float wt[16];                                         // B[k = c][col = j] = W1[c][j], k = 16q + e
#pragma unroll
    for (int e = 0; e < 16; ++e) wt[e] = a.w1[(int64_t)(16 * q + e) * E + j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 4 * q + r;
        const float out = a.feats[row * FEAT + a.out_col + j];
        const float d1 = out > 0.0f ? a.dfeat[row * FEAT + a.out_col + j] : 0.0f;
        a.bw_d1[row * E + j] = d1;
        ds[(4 * q + r) * MHLD + j] = d1;
    }
    __syncthreads();
    const float4* da = reinterpret_cast<const float4*>(&ds[(lane & 15) * MHLD + 16 * q]);
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
        const float4 u = da[c], v = da[c + 1];
        c0 = mfma16(u.x, wt[4 * c + 0], c0); c1 = mfma16(v.x, wt[4 * c + 4], c1);
        c0 = mfma16(u.y, wt[4 * c + 1], c0); c1 = mfma16(v.y, wt[4 * c + 5], c1);
        c0 = mfma16(u.z, wt[4 * c + 2], c0); c1 = mfma16(v.z, wt[4 * c + 6], c1);
        c0 = mfma16(u.w, wt[4 * c + 3], c0); c1 = mfma16(v.w, wt[4 * c + 7], c1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 4 * q + r;
        a.bw_d0[row * E + j] = a.sv_m0[row * E + j] > 0.0f ? c0[r] + c1[r] : 0.0f;
    }
}

int launch_mlp_bwd(const MlpArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(mlp_enc_bwd_kernel, dim3((unsigned)(a.Bp / TILE)), dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
