// GasEUAEncoder "mlp" forward body (dqn_per.py:108-114, 128-130), shared by its own kernel (enc_mlp.hip) and by the
// conv encoder's launch, where it rides as extra workgroups (enc_conv.hip): one workgroup of 256 threads = 16 samples.
#pragma once
#include "device_utils.hpp"
#include "kernels.hpp"

namespace mctsq {

constexpr int MXLD = 20;      // flattened window row stride (<= 3*4 = 12 values, padded to 16)
constexpr int MHLD = 68;
constexpr int MLP_LDS_FLOATS = TILE * MXLD + TILE * MHLD;

// xs [TILE * MXLD], hs [TILE * MHLD]: LDS of the calling kernel; bid: tile index
template <bool SAVE>
__device__ __forceinline__ void mlp_enc_fwd_body(const MlpArgs& a, float* xs, float* hs, int bid) {
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6, jl = lane & 15, q = lane >> 4;
    const int j = 16 * wq + jl;
    const int64_t row0 = (int64_t)bid * TILE;
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

}  // namespace mctsq
