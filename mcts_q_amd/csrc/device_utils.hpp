// Device-side helpers shared by the gfx950 kernels.
#pragma once
#include "common.hpp"

namespace mctsq {

// D(16x16) += A(16x4) * B(4x16), exact fp32 (k-ordered fmaf chain).  Lane l supplies
// A[row l&15][k l>>4] and B[k l>>4][col l&15]; D register r of lane l is D[4*(l>>4)+r][l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Accumulate 16 consecutive k-steps: lane holds a[0..15], b[0..15] for k = q*K4 + kk.
__device__ __forceinline__ f32x4 mfma16_x4(const float4& a, const float4& b, f32x4 c) {
    c = mfma16(a.x, b.x, c);
    c = mfma16(a.y, b.y, c);
    c = mfma16(a.z, b.z, c);
    c = mfma16(a.w, b.w, c);
    return c;
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// Hardware-rate transcendental forms (v_exp_f32 / v_rcp_f32, ~1 ulp each).  Absolute error of the
// gates is ~1e-7, the same size as fp32 rounding of the O(1) values they produce; parity with the
// fp64 oracle is checked in tests/test_gpu_qnet.py.
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoid_fast(float x) { return fast_rcp(1.0f + fast_exp2(-LOG2E * x)); }
__device__ __forceinline__ float tanh_fast(float x) {
    return 1.0f - 2.0f * fast_rcp(1.0f + fast_exp2((2.0f * LOG2E) * x));
}

// DPP move within rows of 16 lanes (row_ror:n = 0x120 + n)
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes that share lane>>4 (one MFMA output row group); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_row<0x128>(v);
    v += dpp_row<0x124>(v);
    v += dpp_row<0x122>(v);
    v += dpp_row<0x121>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    v = fmaxf(v, __shfl_xor(v, 2, 64));
    v = fmaxf(v, __shfl_xor(v, 4, 64));
    v = fmaxf(v, __shfl_xor(v, 8, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// time features of the non-_per flavour (mctsq_config_dqn.py:563-567, 582-585): fp32 casts of the
// fp64 sin/cos values, period 4 for the 15-minute step; c = 0 sin, 1 cos.
__device__ __forceinline__ float time_feature(int mode, int t, int c) {
    // fp64 sin/cos of k*pi/2 rounded to fp32 (cos(pi/2) etc. are tiny, not zero)
    const float S4[4] = {0.0f, 1.0f, 1.2246467991473532e-16f, -1.0f};
    const float C4[4] = {1.0f, 6.123233995736766e-17f, -1.0f, -1.8369701987210297e-16f};
    int ph;
    if (mode == MCTSQ_TF_MINUTE) ph = t & 3;          // minutes 0,15,30,45 -> k = 0..3
    else ph = (t == 1) ? 0 : 2;                       // hours 12,0,12 -> pi, 0, pi
    return c == 0 ? S4[ph] : C4[ph];
}

__device__ __forceinline__ float apply_act(float v, int act) {
    return act == MCTSQ_ACT_RELU ? fmaxf(v, 0.0f) : tanhf(v);
}

}  // namespace mctsq
