// Weight preparation shared by the per-purpose prepare kernels and the learner's one-launch prepare
// (dqn_prepare_kernel): fp32 weights -> 16-bit planes laid out as the B-operand fragments the consuming kernels
// load, u32x4 [fragment][lane 64]: three bf16 planes (hi + mid + lo == w exactly) for the heads and the backward's
// W_hh, two fp16 planes of power-of-two scaled rows for the forward recurrence (device_utils.hpp).  `t` = thread index inside the job.
#pragma once
#include "device_utils.hpp"
#include "kernels.hpp"
#include "learner.hpp"

namespace mctsq {

// W_hh [192][64] of one GRU slot for the forward recurrence, as split-fp16 fragments (device_utils.hpp): every row
// (output unit) is scaled by its own power of two 2^sh so that its largest magnitude lands in [2^14, 2^15):
// [wave 4][gate 3][K chunk 2][plane 2][lane 64] u32x4, then int32 k[192] = 14 + sh per row (the exponent the
// accumulators of that unit are scaled by: h goes in scaled by 2^14).  1536 threads per slot
__device__ __forceinline__ void prep_gru_fwd(const float* __restrict__ whh, u32x4* __restrict__ out, int t) {
    const int lane = t & 63, c = (t >> 6) & 1, rest = t >> 7, g = rest % 3, wq = (rest / 3) & 3;
    const int jl = lane & 15, q = lane >> 4;
    const int row = g * E + 16 * wq + jl;
    const float4* pr = reinterpret_cast<const float4*>(whh + (int64_t)row * E);
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < E / 4; ++i) {
        const float4 v = pr[i];
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    const int sh = f16_row_shift(amax);
    const float sc = pow2i(sh);
    float4 a = pr[8 * c + 2 * q], b = pr[8 * c + 2 * q + 1];
    a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc; b.x *= sc; b.y *= sc; b.z *= sc; b.w *= sc;
    u32x4 hi, lo;
    split8h(a, b, hi, lo);
    u32x4* o = out + (((wq * 3 + g) * 2 + c) * 2) * 64 + lane;
    o[0] = hi; o[64] = lo;
    if (c == 0 && q == 0) reinterpret_cast<int*>(out + GRU_FWD_FRAG_BYTES / 16)[row] = F16_TOP + sh;
}

// W_hh of one GRU slot as the B operand of the BPTT product delta . W_hh: [wave 4][K chunk 6][plane 3][lane 64];
// 1536 threads per slot
__device__ __forceinline__ void prep_gru_bwd(const float* __restrict__ whh, u32x4* __restrict__ out, int t) {
    const int lane = t & 63, rest = t >> 6, c = rest % 6, wq = (rest / 6) & 3;
    const int jl = lane & 15, q = lane >> 4;
    const float* w = whh + (int64_t)(32 * c + 8 * q) * E + 16 * wq + jl;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[e * E];
    u32x4 hi, mid, lo;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, mid, lo);
    u32x4* o = out + ((wq * 6 + c) * 3) * 64 + lane;
    o[0] = hi; o[64] = mid; o[128] = lo;
}

// the 192-128-128-A head for the forward kernels (head_core.hpp): lane (jl, q) of fragment (tile, chunk c) holds
// W[16 tile + jl][32c + 8q .. +7]; rows >= A of the output layer are zero.  84 fragment groups x 64 lanes
constexpr int HEAD_PREP_THREADS = 84 * 64;
__device__ __forceinline__ void prep_head_fwd(const float* __restrict__ w0, const float* __restrict__ w1,
                                              const float* __restrict__ w2, int A, u32x4* __restrict__ out, int t) {
    const int lane = t & 63, fg = t >> 6;
    if (fg >= 84) return;
    const int jl = lane & 15, q = lane >> 4;
    const float* w;
    int K, tile, c, base;
    bool valid = true;
    if (fg < 48) { w = w0; K = FEAT; tile = fg / 6; c = fg % 6; base = fg * 3; }
    else if (fg < 80) { const int f = fg - 48; w = w1; K = 128; tile = f / 4; c = f % 4; base = 8 * 6 * 3 + f * 3; }
    else { const int f = fg - 80; w = w2; K = 128; tile = 0; c = f; base = 8 * 6 * 3 + 8 * 4 * 3 + f * 3; valid = jl < A; }
    float4 a = make_float4(0, 0, 0, 0), b = a;
    if (valid) {
        const float4* p = reinterpret_cast<const float4*>(w + (int64_t)(16 * tile + jl) * K + 32 * c + 8 * q);
        a = p[0]; b = p[1];
    }
    u32x4 hi, mid, lo;
    split8(a, b, hi, mid, lo);
    out[(base + 0) * 64 + lane] = hi; out[(base + 1) * 64 + lane] = mid; out[(base + 2) * 64 + lane] = lo;
}

// the head for the backward kernel (d(in) = d(out) . W, k = output unit): 8 + 32 + 48 fragment groups x 64 lanes
constexpr int HEAD_PREP_BWD_THREADS = 88 * 64;
__device__ __forceinline__ void prep_head_bwd(const float* __restrict__ w0, const float* __restrict__ w1,
                                              const float* __restrict__ w2, int A, u32x4* __restrict__ out, int t) {
    const int lane = t & 63, fg = t >> 6;
    if (fg >= 88) return;
    const int jl = lane & 15, q = lane >> 4;
    constexpr int TB2_ = 8 * 3, TB1_ = 8 * 4 * 3;
    const float* W;
    int N, K, tile, c, base;
    if (fg < 8) { W = w2; N = 128; K = A; tile = fg; c = 0; base = fg * 3; }
    else if (fg < 40) { const int f = fg - 8; W = w1; N = 128; K = 128; tile = f >> 2; c = f & 3; base = TB2_ + f * 3; }
    else { const int f = fg - 40; W = w0; N = FEAT; K = 128; tile = f >> 2; c = f & 3; base = TB2_ + TB1_ + f * 3; }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 32 * c + 8 * q + e;
        v[e] = k < K ? W[(int64_t)k * N + 16 * tile + jl] : 0.0f;
    }
    u32x4 hi, mid, lo;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, mid, lo);
    out[(base + 0) * 64 + lane] = hi; out[(base + 1) * 64 + lane] = mid; out[(base + 2) * 64 + lane] = lo;
}

// Everything the PER-DQN step multiplies by, for an all-GRU network with the 192-128-128-A head, in ONE launch:
// policy and target forward fragments (the mctsq_q_prepare layout), and the policy's backward fragments.
struct DqnPrepArgs {
    const float* whh[2][3];           // [policy, target][slot]
    const float* hw[2][3];            // [policy, target][layer]
    u32x4* gru_fwd[2];                // slot-major, GRU_PREP_BYTES each
    u32x4* head_fwd[2];
    u32x4* gru_bwd[3];                // policy
    u32x4* head_bwd;                  // policy
    int A;
};
int launch_dqn_prepare(const DqnPrepArgs& a, hipStream_t st);

// ---- the one-launch prepare as a function of (256-thread block b of DQN_PREP_BLOCKS, thread t): block ranges
//   [0, 18) policy W_hh forward   [18, 36) target W_hh forward   [36, 54) policy W_hh backward
//   [54, 75) policy head forward  [75, 96) target head forward   [96, 118) policy head backward
// dqn_prepare_kernel runs it as its grid; the learner step's sampling launch carries it in extra workgroups
// (per_kernels.hip), so that back-to-back steps pay no launch for it.
constexpr int DQN_PREP_BLOCKS = 118;
static_assert(HEAD_PREP_THREADS == 21 * 256 && HEAD_PREP_BWD_THREADS == 22 * 256, "block ranges of dqn_prepare_block");
__device__ __forceinline__ void dqn_prepare_block(const DqnPrepArgs& a, int b, int t) {
    if (b < 36) {
        const int net = b / 18, tid = (b - net * 18) * 256 + t, slot = tid / 1536;
        prep_gru_fwd(a.whh[net][slot], a.gru_fwd[net] + (int64_t)slot * (GRU_PREP_BYTES / 16), tid - slot * 1536);
    } else if (b < 54) {
        const int tid = (b - 36) * 256 + t, slot = tid / 1536;
        prep_gru_bwd(a.whh[0][slot], a.gru_bwd[slot], tid - slot * 1536);
    } else if (b < 96) {
        const int net = (b - 54) / 21, tid = (b - 54 - net * 21) * 256 + t;
        prep_head_fwd(a.hw[net][0], a.hw[net][1], a.hw[net][2], a.A, a.head_fwd[net], tid);
    } else if (b < DQN_PREP_BLOCKS) {
        prep_head_bwd(a.hw[0][0], a.hw[0][1], a.hw[0][2], a.A, a.head_bwd, (b - 96) * 256 + t);
    }
}

}  // namespace mctsq
