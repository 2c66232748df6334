// GasEUAEncoder "mlp" (dqn_per.py:108-114, 128-130): flatten [B,3,F] -> Linear(3F,64) -> ReLU ->
// Linear(64,64) -> ReLU (no LayerNorm), forward and backward.  One workgroup = 16 samples; wave wq
// owns output units 16wq..16wq+15; both layers are MFMA products with the weight fragments in VGPRs.
#include "device_utils.hpp"
#include "kernels.hpp"
#include "learner.hpp"
#include "mlp_core.hpp"

namespace mctsq {

template <bool SAVE>
__global__ __launch_bounds__(256) void mlp_enc_fwd_kernel(MlpArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[MLP_LDS_FLOATS];
    mlp_enc_fwd_body<SAVE>(a, lds, lds + TILE * MXLD, (int)blockIdx.x);
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
