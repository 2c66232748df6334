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

// ---- split-bf16 matrix arithmetic ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// D(16x16) += A(16x32) * B(32x16) on bf16 inputs, fp32 accumulate.  Lane l supplies A[row l&15][k = 8(l>>4)+e]
// and B[k = 8(l>>4)+e][col l&15], e = 0..7 packed two per register; D layout as mfma16.
__device__ __forceinline__ f32x4 mfma_bf(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// x = hi + mid + lo exactly, each part a bf16 (by truncation: every part takes the next 8 mantissa bits)
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mid);
    lo = __float_as_uint(r2) & 0xffff0000u;
}
// 8 consecutive fp32 -> three packed bf16x8 planes (element e in bits 16(e&1) of word e>>1)
__device__ __forceinline__ void split8(const float4& x0, const float4& x1, u32x4& hi, u32x4& mid, u32x4& lo) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(x[2 * p], h0, m0, l0);
        split3(x[2 * p + 1], h1, m1, l1);
        hi[p] = (h0 >> 16) | h1; mid[p] = (m0 >> 16) | m1; lo[p] = (l0 >> 16) | l1;
    }
}

// ---- split-fp16 matrix arithmetic for operands of known magnitude (the GRU recurrence: |h| <= 1, W_hh scaled per row) ----
// A value scaled into the upper fp16 range is hi + lo with hi = fp16(x) (round to nearest, 11 significant bits)
// and lo = fp16(x - hi) (the difference is exact in fp32; lo carries the next 11 bits): x to ~2^-23 relative, i.e.
// fp32 accuracy, in TWO planes, so a product needs the three terms hi*hi, hi*lo, lo*hi (lo*lo is below 2^-22 of
// the product) -- half the matrix instructions of the three-plane bf16 split.  fp16 has 5 exponent bits: the caller
// scales by a power of two (exact) so that the largest magnitude sits near 2^14; a value 2^-17 of that still has
// a normal lo, smaller ones lose relative accuracy only where their contribution is below 2^-28 of the largest.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int F16_TOP = 14;                  // scaled magnitudes stay below 2^(F16_TOP + 1)
__device__ __forceinline__ f32x4 mfma_h(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split2h(float xs, unsigned short& hi, unsigned short& lo) {
    const _Float16 h = (_Float16)xs;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (_Float16)(xs - (float)h));
}
// 8 consecutive (already scaled) fp32 -> two packed fp16x8 planes (element e in bits 16(e&1) of word e>>1).  Written
// on two-element fp16 vectors so that the compiler converts straight into the halves of the packed word
// (v_fma_mixlo / v_fma_mixhi_f16, with the caller's scale folded in) instead of assembling it with shifts.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8h(const float4& x0, const float4& x1, u32x4& hi, u32x4& lo) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f16x2 h, l;
        h.x = (_Float16)x[2 * p]; h.y = (_Float16)x[2 * p + 1];
        l.x = (_Float16)(x[2 * p] - (float)h.x); l.y = (_Float16)(x[2 * p + 1] - (float)h.y);
        hi[p] = __builtin_bit_cast(unsigned, h); lo[p] = __builtin_bit_cast(unsigned, l);
    }
}
// ReLU, power-of-two scale and two-way fp16 split of 8 accumulator values in 3 vector instructions per value: the
// integer maximum (ReLU), one mixed-precision fma that converts s * relu(x) straight into its half of the packed hi
// word, one that converts s * relu(x) - hi (hi read back as an fp16 source) into its half of the lo word.  Same values
// as split8h(relu(x) * s): s is a power of two, so every product is exact and each plane is one rounding.  (Left to the
// compiler, split8h costs 5.5: it converts s * relu(x) twice, once packed for the hi plane and once as the subtrahend.)
__device__ __forceinline__ void relu_scale_split8h(const f32x4& a, const f32x4& b, const float4& sa, const float4& sb,
                                                   u32x4& hi, u32x4& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    const float s[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r0 = max(__float_as_int(x[2 * p]), 0), r1 = max(__float_as_int(x[2 * p + 1]), 0);
        unsigned h, l;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(s[2 * p]), "v"(r0));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(s[2 * p + 1]), "v"(r1));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(s[2 * p]), "v"(r0), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(s[2 * p + 1]), "v"(r1), "v"(h));
        hi[p] = h; lo[p] = l;
    }
}
// power of two 2^e as a float; e is clamped to the normal range [-126, 127] (exponents outside it only arise from
// weights or inputs beyond 2^+-100, where the fp32 evaluation overflows as well)
__device__ __forceinline__ float pow2i(int e) {
    e = e < -126 ? -126 : (e > 127 ? 127 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
// the power-of-two exponent that brings a row of largest magnitude `amax` into [2^F16_TOP, 2^(F16_TOP+1))
__device__ __forceinline__ int f16_row_shift(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127;      // floor(log2 amax) for normal amax
    const int sh = amax > 0.0f ? F16_TOP - e : 0;
    return sh < -100 ? -100 : (sh > 40 ? 40 : sh);
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
__device__ __forceinline__ float row16_max_dpp(float v) {
    v = fmaxf(v, dpp_row<0x128>(v));
    v = fmaxf(v, dpp_row<0x124>(v));
    v = fmaxf(v, dpp_row<0x122>(v));
    v = fmaxf(v, dpp_row<0x121>(v));
    return v;
}
__device__ __forceinline__ float row16_min_dpp(float v) {
    v = fminf(v, dpp_row<0x128>(v));
    v = fminf(v, dpp_row<0x124>(v));
    v = fminf(v, dpp_row<0x122>(v));
    v = fminf(v, dpp_row<0x121>(v));
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
