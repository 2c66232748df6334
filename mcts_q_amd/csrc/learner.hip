// Learner kernels: TD loss, head backward, GRU-attention backward (BPTT), weight-gradient GEMMs.
// Together with the SAVE variants of the forward kernels they are loss.backward() of
// DQNModel.update (dqn_per.py:455-470), written out by hand.
//
// Structure: the backward kernels only produce "delta" arrays (gradients w.r.t. pre-activations)
// tile by tile; every weight/bias gradient is then a K = batch(*time) contraction
// dW = delta^T . input, done by one generic split-K MFMA kernel whose partials are reduced in a
// fixed order (bitwise reproducible, no float atomics).
#include "device_utils.hpp"
#include <type_traits>
#include "learner.hpp"
#include "prepare_core.hpp"
#ifndef EXP_BWD
#define EXP_BWD 0
#endif

#include <string.h>

namespace mctsq {

// ================================================================================ TD loss
__global__ __launch_bounds__(1024) void td_loss_kernel(TdArgs a) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < a.Bp; i += 1024) {
        float g = 0.0f;
        int ac = -1;
        if (i < a.B) {
            ac = a.act[i];
            const bool ok = ac >= 0 && ac < a.A;
            const float qsa = ok ? a.q[i * a.A + ac] : 0.0f;
            // rewards + gamma * next_q * (1 - dones), evaluated left to right as torch does (:464)
            const float y = __fadd_rn(a.rew[i], __fmul_rn(__fmul_rn(a.gamma, a.qnext[i]), 1.0f - a.done[i]));
            const float td = qsa - y;
            const float w = a.isw[i];
            acc += (double)__fmul_rn(w, __fmul_rn(td, td));
            g = ok ? a.grad_scale * w * td : 0.0f;
            if (!ok) ac = -1;
            a.abs_td[i] = fabsf(td);
            if (a.qsa) a.qsa[i] = qsa;
            if (a.target) a.target[i] = y;
        }
        float4* row = reinterpret_cast<float4*>(a.dq + i * DQ_LD);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = make_float4(0, 0, 0, 0);
            if ((ac >> 2) == c) (&v.x)[ac & 3] = g;
            row[c] = v;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        *a.loss = (float)(t * a.loss_norm);
    }
}

int launch_td_loss(const TdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(td_loss_kernel, dim3(1), dim3(1024), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

// ============================================================================ head backward
// One workgroup = 16 samples.  d(out) [16 x K] sits in LDS; d(in)[i][n] = sum_k d(out)[i][k] W[k][n]
// is a [16 x K].[K x N] MFMA product with W read row-wise (coalesced 64-byte segments, L2
// resident), then masked by the activation derivative of the layer below.
constexpr int BLD = 260;

__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float d[2][TILE * BLD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jl = lane & 15, q = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * TILE;
    for (int idx = tid; idx < TILE * DQ_LD; idx += 256) {
        const int i = idx / DQ_LD, c = idx - i * DQ_LD;
        d[0][i * BLD + c] = a.dq[(row0 + i) * DQ_LD + c];
    }
    __syncthreads();
    int cur = 0;
    for (int l = a.n_fc - 1; l >= 0; --l) {
        const int K = a.out[l], N = a.in[l];
        const int Kp = (K + 15) & ~15, K4 = Kp >> 2;            // lane group q covers k in [q*K4, (q+1)*K4)
        const float* W = a.w[l];
        const float* src = d[cur] + (lane & 15) * BLD + q * K4;
        float* dst = d[cur ^ 1];
        for (int nt = w; nt < (N >> 4); nt += 4) {
            const int n = nt * 16 + jl;
            f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            for (int kk = 0; kk < K4; kk += 4) {
                const float4 av = *reinterpret_cast<const float4*>(src + kk);
                const int k = q * K4 + kk;
                const float b0 = k + 0 < K ? W[(int64_t)(k + 0) * N + n] : 0.0f;
                const float b1 = k + 1 < K ? W[(int64_t)(k + 1) * N + n] : 0.0f;
                const float b2 = k + 2 < K ? W[(int64_t)(k + 2) * N + n] : 0.0f;
                const float b3 = k + 3 < K ? W[(int64_t)(k + 3) * N + n] : 0.0f;
                acc0 = mfma16(av.x, b0, acc0); acc1 = mfma16(av.y, b1, acc1);
                acc0 = mfma16(av.z, b2, acc0); acc1 = mfma16(av.w, b3, acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = row0 + 4 * q + r;
                float v = acc0[r] + acc1[r];
                if (l > 0) {
                    const float h = a.sv_act[l - 1][row * N + n];
                    v = a.act == MCTSQ_ACT_RELU ? (h > 0.0f ? v : 0.0f) : v * (1.0f - h * h);
                    a.d_act[l - 1][row * N + n] = v;
                    dst[(4 * q + r) * BLD + n] = v;
                } else {
                    a.dfeat[row * N + n] = v;
                }
            }
        }
        // zero the k padding the next layer reads (N is a multiple of 16, so nothing to pad)
        __syncthreads();
        cur ^= 1;
    }
}

// ---- the 192-128-128-A head backward on split-bf16 MFMA ----------------------------------------------------
// Same arithmetic as the forward head (head_core.hpp): exact three-way split of both operands, six cross terms,
// fp32 accumulate.  8 waves per 16-row tile; every wave fetches its (prepared, transposed, lane-major) weight
// fragments of all three layers up front -- one round trip instead of a dependent global load per k-step --
// and the deltas travel between layers as bf16 planes in LDS.
constexpr int TB2 = 8 * 3, TB1 = 8 * 4 * 3, TB0 = 12 * 4 * 3;     // fragments per section
static_assert(HEAD_PREP_BWD_BYTES == (int64_t)(TB2 + TB1 + TB0) * 64 * 16, "prepared backward head size");
constexpr int DQP = 32 + 8, DPL = 128 + 8;                        // plane row strides (bf16 elements)

#define HB_MM6(ACC, XH, XM, XL, W)                                                                 \
    ACC = mfma_bf(XL, (W)[0], ACC); ACC = mfma_bf(XH, (W)[2], ACC); ACC = mfma_bf(XM, (W)[1], ACC); \
    ACC = mfma_bf(XM, (W)[0], ACC); ACC = mfma_bf(XH, (W)[1], ACC); ACC = mfma_bf(XH, (W)[0], ACC);

__global__ __launch_bounds__(512) void head_bwd_bf16_kernel(HeadBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short pq[3][TILE * DQP];
    __shared__ __attribute__((aligned(16))) unsigned short pd[2][3][TILE * DPL];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jl = lane & 15, q = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * TILE;
    const u32x4* hp = reinterpret_cast<const u32x4*>(a.wprep_bwd) + lane;
    // fragments: W2 tile w; W1 tile w; W0 tiles w and (waves 0-3) w + 8
    u32x4 f2[3], f1[12], f0[2][12];
#pragma unroll
    for (int f = 0; f < 3; ++f) f2[f] = hp[(w * 3 + f) * 64];
#pragma unroll
    for (int f = 0; f < 12; ++f) f1[f] = hp[(TB2 + w * 12 + f) * 64];
#pragma unroll
    for (int f = 0; f < 12; ++f) f0[0][f] = hp[(TB2 + TB1 + w * 12 + f) * 64];
    if (w < 4) {
#pragma unroll
        for (int f = 0; f < 12; ++f) f0[1][f] = hp[(TB2 + TB1 + (w + 8) * 12 + f) * 64];
    }
    // saved activations behind the derivative masks: rows 4q + r, column 16w + jl
    float h1[4], h0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        h1[r] = a.sv_act[1][(row0 + 4 * q + r) * 128 + 16 * w + jl];
        h0[r] = a.sv_act[0][(row0 + 4 * q + r) * 128 + 16 * w + jl];
    }
    // fused TD loss (dqn_per.py:455-467, 474): the arithmetic of td_loss_kernel for this tile's 16 rows
    __shared__ float s_dq[TILE][DQ_LD];
    __shared__ double s_loss[TILE];
    if (a.fuse_td) {
        const TdArgs& t = a.td;
        if (tid < TILE) {
            const int64_t i = row0 + tid;
            float g = 0.0f;
            int ac = -1;
            double l = 0.0;
            if (i < t.B) {
                // the whole Q row travels with the action index (no dependent second round trip)
                float qrow[DQ_LD];
#pragma unroll
                for (int c = 0; c < DQ_LD; ++c) qrow[c] = c < t.A ? t.q[i * t.A + c] : 0.0f;
                ac = t.act[i];
                const float rw = t.rew[i], qn = t.qnext[i], dn = t.done[i];
                const bool ok = ac >= 0 && ac < t.A;
                float qsa = 0.0f;
#pragma unroll
                for (int c = 0; c < DQ_LD; ++c) qsa = c == ac ? qrow[c] : qsa;
                const float y = __fadd_rn(rw, __fmul_rn(__fmul_rn(t.gamma, qn), 1.0f - dn));
                const float td = qsa - y;
                const float w = t.isw[i];
                l = (double)__fmul_rn(w, __fmul_rn(td, td));
                g = ok ? t.grad_scale * w * td : 0.0f;
                if (!ok) ac = -1;
                t.abs_td[i] = fabsf(td);
                if (t.qsa) t.qsa[i] = qsa;
                if (t.target) t.target[i] = y;
            }
            s_loss[tid] = l;
            float4* row = reinterpret_cast<float4*>(t.dq + i * DQ_LD);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 v = make_float4(0, 0, 0, 0);
                if ((ac >> 2) == c) (&v.x)[ac & 3] = g;
                row[c] = v;
                *reinterpret_cast<float4*>(&s_dq[tid][4 * c]) = v;
            }
        }
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int i = 0; i < TILE; ++i) sum += s_loss[i];
            a.loss_parts[blockIdx.x] = sum;
        }
    }
    // dQ [16 x 16] -> planes, k padded to 32 with zeros
    {
        const int i = tid >> 5, c = tid & 31;
        unsigned u0 = 0, u1 = 0, u2 = 0;
        if (c < DQ_LD) split3(a.fuse_td ? s_dq[i][c] : a.dq[(row0 + i) * DQ_LD + c], u0, u1, u2);
        pq[0][i * DQP + c] = (unsigned short)(u0 >> 16);
        pq[1][i * DQP + c] = (unsigned short)(u1 >> 16);
        pq[2][i * DQP + c] = (unsigned short)(u2 >> 16);
    }
    __syncthreads();
    auto mask = [&](float v, float h) { return a.act == MCTSQ_ACT_RELU ? (h > 0.0f ? v : 0.0f) : v * (1.0f - h * h); };
    // ---- output layer: d1 = (dQ . W2) * act'(h1), column tile w --------------------------------------------
    {
        const int o = (lane & 15) * DQP + 8 * q;
        const u32x4 xh = *reinterpret_cast<const u32x4*>(&pq[0][o]);
        const u32x4 xm = *reinterpret_cast<const u32x4*>(&pq[1][o]);
        const u32x4 xl = *reinterpret_cast<const u32x4*>(&pq[2][o]);
        f32x4 acc = {0, 0, 0, 0};
        HB_MM6(acc, xh, xm, xl, f2)
        const int col = 16 * w + jl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = mask(acc[r], h1[r]);
            a.d_act[1][(row0 + 4 * q + r) * 128 + col] = v;
            unsigned u0, u1, u2;
            split3(v, u0, u1, u2);
            pd[1][0][(4 * q + r) * DPL + col] = (unsigned short)(u0 >> 16);
            pd[1][1][(4 * q + r) * DPL + col] = (unsigned short)(u1 >> 16);
            pd[1][2][(4 * q + r) * DPL + col] = (unsigned short)(u2 >> 16);
        }
    }
    __syncthreads();
    // ---- layer 1: d0 = (d1 . W1) * act'(h0), column tile w ---------------------------------------------------
    {
        const int o = (lane & 15) * DPL + 8 * q;
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            const u32x4 xh = *reinterpret_cast<const u32x4*>(&pd[1][0][o + 32 * c]);
            const u32x4 xm = *reinterpret_cast<const u32x4*>(&pd[1][1][o + 32 * c]);
            const u32x4 xl = *reinterpret_cast<const u32x4*>(&pd[1][2][o + 32 * c]);
            const u32x4 yh = *reinterpret_cast<const u32x4*>(&pd[1][0][o + 32 * c + 32]);
            const u32x4 ym = *reinterpret_cast<const u32x4*>(&pd[1][1][o + 32 * c + 32]);
            const u32x4 yl = *reinterpret_cast<const u32x4*>(&pd[1][2][o + 32 * c + 32]);
            HB_MM6(acc0, xh, xm, xl, &f1[3 * c])
            HB_MM6(acc1, yh, ym, yl, &f1[3 * c + 3])
        }
        const int col = 16 * w + jl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = mask(acc0[r] + acc1[r], h0[r]);
            a.d_act[0][(row0 + 4 * q + r) * 128 + col] = v;
            unsigned u0, u1, u2;
            split3(v, u0, u1, u2);
            pd[0][0][(4 * q + r) * DPL + col] = (unsigned short)(u0 >> 16);
            pd[0][1][(4 * q + r) * DPL + col] = (unsigned short)(u1 >> 16);
            pd[0][2][(4 * q + r) * DPL + col] = (unsigned short)(u2 >> 16);
        }
    }
    __syncthreads();
    // ---- layer 0: dfeat = d0 . W0, column tiles w and (waves 0-3) w + 8 --------------------------------------
    {
        const int o = (lane & 15) * DPL + 8 * q;
        f32x4 acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4 xh = *reinterpret_cast<const u32x4*>(&pd[0][0][o + 32 * c]);
            const u32x4 xm = *reinterpret_cast<const u32x4*>(&pd[0][1][o + 32 * c]);
            const u32x4 xl = *reinterpret_cast<const u32x4*>(&pd[0][2][o + 32 * c]);
            HB_MM6(acc[0][c & 1], xh, xm, xl, &f0[0][3 * c])
            if (w < 4) { HB_MM6(acc[1][c & 1], xh, xm, xl, &f0[1][3 * c]) }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a.dfeat[(row0 + 4 * q + r) * FEAT + 16 * w + jl] = acc[0][0][r] + acc[0][1][r];
            if (w < 4) a.dfeat[(row0 + 4 * q + r) * FEAT + 16 * (w + 8) + jl] = acc[1][0][r] + acc[1][1][r];
        }
    }
}
#undef HB_MM6

// one thread per (fragment group, lane): 8 k-values (output units) of one input column -> 3 x 8 bf16
__global__ __launch_bounds__(256) void head_prepare_bwd_kernel(const float* __restrict__ w0, const float* __restrict__ w1,
                                                               const float* __restrict__ w2, int A, u32x4* __restrict__ out) {
    prep_head_bwd(w0, w1, w2, A, out, blockIdx.x * 256 + threadIdx.x);
}
int launch_head_prepare_bwd(const float* w0, const float* w1, const float* w2, int A, void* prep, hipStream_t st) {
    MCTSQ_CHECK(w0 && w1 && w2 && prep && A >= 1 && A <= 16, "head backward prepare: bad arguments");
    hipLaunchKernelGGL(head_prepare_bwd_kernel, dim3((88 * 64 + 255) / 256), dim3(256), 0, st, w0, w1, w2, A,
                       reinterpret_cast<u32x4*>(prep));
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_head_bwd(const HeadBwdArgs& a, hipStream_t st) {
    for (int l = 0; l < a.n_fc; ++l) {
        MCTSQ_CHECK(a.in[l] % 16 == 0 && a.in[l] <= 256, "head backward layer %d: width %d unsupported", l, a.in[l]);
        MCTSQ_CHECK(a.out[l] <= 256, "head backward layer %d: width %d unsupported", l, a.out[l]);
    }
    MCTSQ_CHECK(a.out[a.n_fc - 1] <= DQ_LD, "action_dim > %d", DQ_LD);
    if (a.wprep_bwd && a.n_fc == 3 && a.in[0] == FEAT && a.out[0] == 128 && a.in[1] == 128 && a.out[1] == 128 &&
        a.in[2] == 128) {
        hipLaunchKernelGGL(head_bwd_bf16_kernel, dim3((unsigned)(a.Bp / TILE)), dim3(512), 0, st, a);
        MCTSQ_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)(a.Bp / TILE)), dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

// ================================================================= GRU-attention backward
// dqn_per.py:72-77 differentiated by hand.  One workgroup (4 waves) per (16-sample tile, slot),
// the same ownership as the forward: wave wq holds hidden units 16wq..16wq+15, lane register r is
// sample 4q+r.  Per step the hidden-state gradient needs [16 x 192].[192 x 64] (delta . W_hh):
// delta goes through a double-buffered LDS tile, W_hh columns stay in 48 VGPRs per lane.
constexpr int GPL = 192 + 8;  // row stride (bf16 elements) of a delta plane: 400 B, b128 reads conflict-free

struct GruBwdLds {
    unsigned short gp[2][3][TILE * GPL];   // delta tile [drp | dzp | ghn] as bf16 planes (hi, mid, lo), double buffered
    float4 pe[MAX_SEQ][TILE];       // per-wave partials of e_t = dpool . h_t
    float red[3][4][TILE];
    u32x4 hx[2][4 * 3 * 64];        // weight gradients: every wave's h_{t-1} operand [wave][plane][lane], two buffers
};

// ---- weight gradients inside the BPTT kernel -------------------------------------------------------------------
// dW_hh = G^T H_prev and dW_ih = G^T X contract over (step, sample), and the registers a lane holds anyway ARE matrix
// operands of those products: lane (jl, q) of wave w has, per step, the deltas and h_{t-1} of unit 16w + jl for its
// RPL samples - as the A operand row jl (a gate row of this wave's units) resp. the B operand column jl (a hidden
// column), K block q.  Any assignment of (step, sample) pairs to the K slots of a block is fine as long as both
// operands use the same one, so the operands are filled in place, step by step:
//   RPL 2 (8-sample workgroups): K = 32 = 3 steps x 2 samples per lane block (+ 2 zero slots), one group of
//         products ("flush") every third step - the step loop runs in triples already;
//   RPL 4 (16-sample workgroups): K = 16 = 1 step x 4 samples, the 16x16x16 instruction, a flush every step.
// Both operands go through the exact three-way bf16 split (six cross terms), as the BPTT product does: the deltas
// have no bound.  A wave owns the gate rows of its 16 units against ALL 64 hidden columns, so the four waves swap
// their H operands through LDS (12 KB per flush); X (the step's input row, <= 8 features) is read per wave.
// The record G [K][256] the separate weight-gradient kernel read back (64 MB written + 64 MB read per step at the
// training batch, plus H_prev and X again) and that kernel's launch are gone.
#ifndef MCTSQ_WG_TERMS
#define MCTSQ_WG_TERMS 6        // cross terms per weight-gradient product (build parameter for measurements: 3 = two planes, 0 = none)
#endif
#if MCTSQ_WG_TERMS != 6 && !defined(MCTSQ_MEASUREMENT_BUILD)
#error "MCTSQ_WG_TERMS other than 6 gives inexact (3) or no (0) weight gradients: timing builds only, with -DMCTSQ_MEASUREMENT_BUILD"
#endif
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int RPL> struct WgFrag { typedef u32x4 T; };
template <> struct WgFrag<4> { typedef u32x2 T; };
__device__ __forceinline__ f32x4 wg_mfma(const u32x4& a, const u32x4& b, f32x4 c) { return mfma_bf(a, b, c); }
__device__ __forceinline__ f32x4 wg_mfma(const u32x2& a, const u32x2& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

// RPL = sample rows per lane: 4 = one workgroup per 16-sample tile; 2 = one workgroup per HALF tile (8 samples, the
// rows r in {2h, 2h+1} of every lane's four): the matrix products keep their 16-row shape (the unused rows of the
// delta tile are zero and their outputs are never read), but every per-(row, unit) vector instruction, LDS write
// and global access is issued for two rows instead of four.  At the training batch a 16-sample tiling leaves three
// quarters of the CUs idle while the step's serial chain is bound by exactly those instructions, so twice the
// workgroups at half the vector work each shorten the chain (launch_gru_bwd picks the mode).
template <int RPL>
struct RowVec {
    float v[RPL];
    __device__ float& operator[](int r) { return v[r]; }
    __device__ const float& operator[](int r) const { return v[r]; }
};
template <int RPL>
__device__ __forceinline__ RowVec<RPL> load_rows(const float* p) {
    RowVec<RPL> o;
    if constexpr (RPL == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        o.v[0] = t[0]; o.v[1] = t[1]; o.v[2] = t[2]; o.v[3] = t[3];
    } else {
        const f32x2 t = *reinterpret_cast<const f32x2*>(p);
        o.v[0] = t[0]; o.v[1] = t[1];
    }
    return o;
}
template <int RPL>
__device__ __forceinline__ void store_rows(float* p, const RowVec<RPL>& x) {
    if constexpr (RPL == 4) *reinterpret_cast<f32x4*>(p) = (f32x4){x.v[0], x.v[1], x.v[2], x.v[3]};
    else *reinterpret_cast<f32x2*>(p) = (f32x2){x.v[0], x.v[1]};
}

// the step's RPL values of one operand -> slot POS of its three bf16 planes (weight gradients, above)
template <int RPL, int POS>
__device__ __forceinline__ void wg_put(typename WgFrag<RPL>::T (&f)[3], const RowVec<RPL>& v) {
    unsigned p[3][RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) split3(v[r], p[0][r], p[1][r], p[2][r]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        if constexpr (RPL == 2) {
            f[pl][POS] = (p[pl][0] >> 16) | p[pl][1];
        } else {
            f[pl][0] = (p[pl][0] >> 16) | p[pl][1];
            f[pl][1] = (p[pl][2] >> 16) | p[pl][3];
        }
    }
}

template <int RPL>
__device__ __forceinline__ void gru_bwd_body(GruBwdLds& L, const GruBwdArgs& args, int bid) {
    constexpr int NH = 4 / RPL;                                // workgroups per 16-sample tile
    const int slot = bid % args.nslots;
    const int hb = bid / args.nslots;
    const int64_t tile = hb / NH;
    const int half = hb % NH;
    const int64_t row0 = tile * TILE;
    const GruSlot& s = args.slot[slot];
    const int lane = threadIdx.x & 63, wq = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), jl = lane & 15, q = lane >> 4;
    const int j = 16 * wq + jl;
    const int S = s.S;
    const int64_t Bp = args.Bp;
    // lane-major records (kernels.hpp: GruSlot): block (t, tile, wave) is 256 floats, a lane's four rows one float4
    const int64_t ntile = Bp / TILE;
    // Which samples a lane owns.  RPL == 4: rows 4q .. 4q+3, its whole float4 of a record.  RPL == 2: the workgroup owns the
    // tile's rows 8*half .. 8*half+7 and lane (jl, q) rows 8*half + 2q + {0, 1}, i.e. one half of the float4 of record
    // lane jl + 16 * (2*half + q/2): the wave's 64 lanes together read / write one contiguous 512-byte half of the
    // record block (an interleaved split - rows 2*half + {0, 1} of every lane - made both workgroups of a tile touch
    // every cache line: 128 MB of traffic per launch instead of 35).
    const int rbase = RPL == 4 ? 4 * q : 8 * half + 2 * q;    // first owned row within the tile
    const int lo4 = RPL == 4 ? lane * 4 : (jl + 16 * (2 * half + (q >> 1))) * 4 + 2 * (q & 1);
    auto rec = [&](const float* base, int t) {
        return load_rows<RPL>(base + (((int64_t)t * ntile + tile) * 4 + wq) * 256 + lo4);
    };
    if constexpr (RPL != 4) {                                  // rows of the delta tile this workgroup never writes
        unsigned short* z = &L.gp[0][0][0];
        for (int i = threadIdx.x; i < 2 * 3 * TILE * GPL / 2; i += 256) reinterpret_cast<unsigned*>(z)[i] = 0u;
    }

    // W_hh as B operand of the delta . W_hh product (k = gate row, col = unit j): prepared split-bf16 fragments,
    // 6 K chunks x 3 planes resident in 72 VGPRs.  The product runs on the bf16 matrix instruction over the exact
    // three-way split of both operands (six cross terms, fp32 accumulate); the forward recurrence, whose operands have
    // known bounds, uses the two-way fp16 split instead (the deltas here have none).
    u32x4 wt[6][3];
    {
        const u32x4* wp = reinterpret_cast<const u32x4*>(s.wprep_t) + (int64_t)wq * (6 * 3 * 64) + lane;
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wt[c][pl] = wp[(c * 3 + pl) * 64];
    }
    const float lnw = s.lnw[j], awj = s.aw[j];
    // The hidden states the pooling backward multiplies by start their trip here, all at once and without a condition
    // around the loads (index clamped instead): fetched in rounds of eight behind the LayerNorm part, each round was
    // one more exposed memory round trip (four for the 25-step window) before the first BPTT step could start.
    constexpr int HB = RPL == 2 ? 32 : 8;                      // RPL 4: 32 registers of records in flight, as before
    RowVec<RPL> hv[HB];
#pragma unroll
    for (int u = 0; u < HB; ++u) hv[u] = rec(s.sv_h, (u < S ? u : S - 1) + 1);

    // ---- LayerNorm backward -> dpool -------------------------------------------------------
    float dpool[RPL], pooled[RPL], mrun[RPL], linv[RPL];
    float dlnw = 0.0f;                     // LayerNorm weight gradient of unit j over this workgroup's rows
    {
        float dxh[RPL], xh[RPL], rs[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int64_t row = row0 + rbase + r;
            const float4 st = *reinterpret_cast<const float4*>(s.sv_stat + row * 4);
            mrun[r] = st.x; linv[r] = 1.0f / st.y; rs[r] = st.w;
            pooled[r] = s.sv_pool[row * E + j];
            const float df = args.dfeat[row * FEAT + s.out_col + j];
            xh[r] = (pooled[r] - st.z) * st.w;
            dxh[r] = df * lnw;
            dlnw += df * xh[r];
            const float s1 = row16_sum(dxh[r]), s2 = row16_sum(dxh[r] * xh[r]);
            if (jl == 0) { L.red[0][wq][4 * q + r] = s1; L.red[1][wq][4 * q + r] = s2; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = 4 * q + r;
            const float s1 = ((L.red[0][0][i] + L.red[0][1][i]) + (L.red[0][2][i] + L.red[0][3][i])) * (1.0f / E);
            const float s2 = ((L.red[1][0][i] + L.red[1][1][i]) + (L.red[1][2][i] + L.red[1][3][i])) * (1.0f / E);
            dpool[r] = rs[r] * (dxh[r] - s1 - xh[r] * s2);
        }
    }
    // ---- attention pooling backward: D = dpool . pooled, e_t = dpool . h_t ---------------
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const float v = row16_sum(dpool[r] * pooled[r]);
        if (jl == 0) L.red[2][wq][4 * q + r] = v;
    }
    for (int t0 = 0; t0 < S; t0 += HB) {
        if (t0 > 0) {                                          // windows longer than one round (not the shipped ones)
#pragma unroll
            for (int u = 0; u < HB; ++u) hv[u] = rec(s.sv_h, (t0 + u < S ? t0 + u : S - 1) + 1);
        }
#pragma unroll
        for (int u = 0; u < HB; ++u) {
            if (t0 + u >= S) break;
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const float v = row16_sum(dpool[r] * hv[u][r]);
                if (jl == 0) (&L.pe[t0 + u][4 * q + r].x)[wq] = v;
            }
        }
    }
    __syncthreads();
    float Dd[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int i = 4 * q + r;
        Dd[r] = (L.red[2][0][i] + L.red[2][1][i]) + (L.red[2][2][i] + L.red[2][3][i]);
    }

    // ---- BPTT ----------------------------------------------------------------------------
    // The saved activations of the steps ahead are fetched while the current step's MFMA chain runs, so the
    // recurrence never waits on HBM/L2 latency.
    struct StepIn { RowVec<RPL> rg, zg, ng, hn, hprev, sc, xv; };
    StepIn sb0, sb1, sb2;                  // three steps of saved activations in flight (HBM latency > one step)
    const float* scp = s.sv_score + row0 + rbase;
    const float* xp = s.sv_x + (row0 + rbase) * 8 + (jl & 7);
    const float xkeep = jl < 8 ? 1.0f : 0.0f;                 // X operand columns 8..15 are zero
    // weight-gradient operands under assembly and accumulators (see above): A = deltas of this wave's units
    // (drp, dzp, ghn = dnp * r, dnp), B = h_{t-1} / the input row
    typedef typename WgFrag<RPL>::T Frag;
    Frag fa[4][3], fb[3], fx[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int g = 0; g < 4; ++g) fa[g][pl] = Frag{};
        fb[pl] = Frag{}; fx[pl] = Frag{};
    }
    f32x4 whh[3][4], wih[3];                                  // [drp | dzp | ghn] x hidden column tile; [drp | dzp | dnp] x X
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) whh[g][nt] = f32x4{0, 0, 0, 0};
        wih[g] = f32x4{0, 0, 0, 0};
    }
    float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};                 // column sums of drp, dzp, dnp, ghn (bias gradients)
    // one group of products: every wave's H operand from LDS (its own included: one code path for the four column tiles),
    // six cross terms smallest first, consecutive instructions on different accumulators
    auto wg_flush = [&](int hbuf) {
        const Frag* hx = reinterpret_cast<const Frag*>(&L.hx[hbuf][0]) + lane;
#define WG_TERM(ACC, A, AP, B, BP) ACC = wg_mfma(A[AP], B[BP], ACC);
#if MCTSQ_WG_TERMS == 6
#define WG_SIX(STMT) STMT(2, 0) STMT(0, 2) STMT(1, 1) STMT(1, 0) STMT(0, 1) STMT(0, 0)
#elif MCTSQ_WG_TERMS == 3
#define WG_SIX(STMT) STMT(1, 0) STMT(0, 1) STMT(0, 0)
#else
#define WG_SIX(STMT)
#endif
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            Frag hb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) hb[pl] = hx[(nt * 3 + pl) * 64];
#define WG_HH(AP, BP) WG_TERM(whh[0][nt], fa[0], AP, hb, BP) WG_TERM(whh[1][nt], fa[1], AP, hb, BP) WG_TERM(whh[2][nt], fa[2], AP, hb, BP)
            WG_SIX(WG_HH)
#undef WG_HH
        }
#define WG_IH(AP, BP) WG_TERM(wih[0], fa[0], AP, fx, BP) WG_TERM(wih[1], fa[1], AP, fx, BP) WG_TERM(wih[2], fa[3], AP, fx, BP)
        WG_SIX(WG_IH)
#undef WG_IH
#undef WG_SIX
#undef WG_TERM
    };
    auto fetch = [&](StepIn& d, int t) {
        d.rg = rec(s.sv_r, t); d.zg = rec(s.sv_z, t); d.ng = rec(s.sv_n, t); d.hn = rec(s.sv_hn, t);
        d.hprev = rec(s.sv_h, t);                              // block t = h_{t-1}
        d.sc = load_rows<RPL>(scp + (int64_t)t * Bp);
#pragma unroll
        for (int r = 0; r < RPL; ++r) d.xv[r] = xp[((int64_t)t * Bp + r) * 8];       // feature jl & 7 of the step's input rows
    };
    fetch(sb0, S - 1);
    if (S >= 2) fetch(sb1, S - 2);
    if (S >= 3) fetch(sb2, S - 3);
    // gradient of the score Linear(64,1): d_aw[j] = sum_{t,row} dsc * h_t[row][j], d_ab = sum dsc.  h_t is
    // the h_{t-1} operand of the step handled just before (the loop runs backwards); the workgroup's sums
    // go out as one partial, reduced over workgroups in order by wgrad_reduce_kernel.
    float daw = 0.0f, dab = 0.0f;
    RowVec<RPL> hnext = rec(s.sv_h, S);
    float dhc[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) dhc[r] = 0.0f;
    // LAST (a compile-time flag): step t == 0, whose hidden-state gradient nobody needs.  The step body has no
    // data-dependent branch around its memory operations: the compiler's wait-count bookkeeping stays exact across the
    // loop (with a conditional prefetch it fell back to waiting for ALL outstanding loads at the top of every iteration,
    // i.e. for the records requested one step earlier - the full memory latency, every third step).
    // POS (RPL 2 only): the step's slot in the weight-gradient operands; the products are issued behind slot 2
    auto bptt_step = [&](int t, StepIn& cur, auto last_tag, auto pos_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int POS = RPL == 2 ? decltype(pos_tag)::value : 0;
        constexpr bool FLUSH = RPL == 4 || POS == 2;
        static_assert(!LAST || FLUSH, "step 0 closes a group");
        // two exchange buffers where flushes follow each other step by step (a fast wave must not overwrite its H
        // operand while a slow one still reads the previous flush's): every step of RPL 4, step 0 behind step 1 of RPL 2
        const int hbuf = RPL == 4 ? (t & 1) : (LAST ? 1 : 0);
        const int buf = t & 1;
        // this step's record moves to locals and the buffer's next occupant starts its trip at once (below step 3: a
        // harmless re-read of record 0 into a buffer that is done): the longer the distance to its use, the less of
        // the memory latency the step sees
        const StepIn in = cur;
        if constexpr (!LAST) fetch(cur, t >= 3 ? t - 3 : 0);
        float zs[RPL], dhs[RPL], dsc[RPL], at[RPL];
        // everything read from LDS first, so that no read has to queue behind this step's plane writes
        {
            float4 pe[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) pe[r] = L.pe[t][4 * q + r];
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                at[r] = fast_exp2(in.sc[r] - mrun[r]) * linv[r];
                dsc[r] = at[r] * (((pe[r].x + pe[r].y) + (pe[r].z + pe[r].w)) - Dd[r]);
            }
        }
        RowVec<RPL> g0, g1, g2, g3;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int i = 4 * q + r;
            const float rg = in.rg[r], zg = in.zg[r], ng = in.ng[r], hn = in.hn[r], hprev = in.hprev[r];
            daw = fmaf(dsc[r], hnext[r], daw); dab += dsc[r];
            const float dh = dhc[r] + at[r] * dpool[r] + dsc[r] * awj;
            const float dnp = dh * (1.0f - zg) * (1.0f - ng * ng);
            const float dzp = dh * (hprev - ng) * zg * (1.0f - zg);
            const float drp = dnp * hn * rg * (1.0f - rg);
            const float ghn = dnp * rg;
            g0[r] = drp; g1[r] = dzp; g2[r] = dnp; g3[r] = ghn;
            {
                const float dv[3] = {drp, dzp, ghn};
#pragma unroll
                for (int g3i = 0; g3i < 3; ++g3i) {
                    unsigned u0, u1, u2;
                    split3(dv[g3i], u0, u1, u2);
                    L.gp[buf][0][i * GPL + g3i * E + j] = (unsigned short)(u0 >> 16);
                    L.gp[buf][1][i * GPL + g3i * E + j] = (unsigned short)(u1 >> 16);
                    L.gp[buf][2][i * GPL + g3i * E + j] = (unsigned short)(u2 >> 16);
                }
            }
            zs[r] = zg; dhs[r] = dh;
        }
        hnext = in.hprev;
        {                                                      // the step's slot of the weight-gradient operands
            if constexpr (LAST && RPL == 2) {                  // step 0 is a group of its own: slots 0, 1 are empty
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) { fa[g][pl][0] = 0u; fa[g][pl][1] = 0u; }
                    fb[pl][0] = 0u; fb[pl][1] = 0u; fx[pl][0] = 0u; fx[pl][1] = 0u;
                }
            }
            wg_put<RPL, POS>(fa[0], g0); wg_put<RPL, POS>(fa[1], g1); wg_put<RPL, POS>(fa[2], g3); wg_put<RPL, POS>(fa[3], g2);
            wg_put<RPL, POS>(fb, in.hprev);
            RowVec<RPL> xm;
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                xm[r] = in.xv[r] * xkeep;
                bsum[0] += g0[r]; bsum[1] += g1[r]; bsum[2] += g2[r]; bsum[3] += g3[r];
            }
            wg_put<RPL, POS>(fx, xm);
            if constexpr (FLUSH) {
                Frag* hx = reinterpret_cast<Frag*>(&L.hx[hbuf][0]) + lane;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) hx[(wq * 3 + pl) * 64] = fb[pl];
            }
        }
        if constexpr (LAST) {                                  // dh_{-1} is not needed
            __syncthreads();
            wg_flush(hbuf);
            return;
        }
        __syncthreads();
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        {
            // Two accumulators (even / odd K chunks) issued alternately, the next pair of chunks requested from LDS
            // while the current pair multiplies: 12 fragment registers in use + 12 in flight.  (Three accumulators with
            // nine + nine fragments measured the same per step but cost 24 more registers, and at that pressure the
            // record prefetch below stalls on register reuse.)
            const int o = (lane & 15) * GPL + 8 * q;
            u32x4 xa[2][3], xb[2][3];
#define GB_LOAD(X, C0)                                                                              \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                      \
                X[0][pl] = *reinterpret_cast<const u32x4*>(&L.gp[buf][pl][o + 32 * (C0)]);          \
                X[1][pl] = *reinterpret_cast<const u32x4*>(&L.gp[buf][pl][o + 32 * ((C0) + 1)]);    \
            }
            // cross terms smallest first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); x plane / w plane
#define GB_TERM(X, C0, XP, WP)                                                                      \
            acc0 = mfma_bf(X[0][XP], wt[(C0)][WP], acc0);                                           \
            acc1 = mfma_bf(X[1][XP], wt[(C0) + 1][WP], acc1);
#define GB_PAIR(X, C0)                                                                              \
            GB_TERM(X, C0, 2, 0) GB_TERM(X, C0, 0, 2) GB_TERM(X, C0, 1, 1)                          \
            GB_TERM(X, C0, 1, 0) GB_TERM(X, C0, 0, 1) GB_TERM(X, C0, 0, 0)
            GB_LOAD(xa, 0)
            GB_LOAD(xb, 2)
            GB_PAIR(xa, 0)
            GB_LOAD(xa, 4)
            GB_PAIR(xb, 2)
            GB_PAIR(xa, 4)
#undef GB_PAIR
#undef GB_TERM
#undef GB_LOAD
        }
        if constexpr (FLUSH) wg_flush(hbuf);                   // behind the chain's products: nothing of this step waits for them
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            dhc[r] = dhs[r] * zs[r] + (acc0[r] + acc1[r]);
        }
    };
    {
        using No = std::integral_constant<bool, false>;
        using Yes = std::integral_constant<bool, true>;
        // steps S-1 .. 1 in groups of three (the buffers keep their roles), the (S-1) % 3 odd ones first, then step 0
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using P2 = std::integral_constant<int, 2>;
        int t = S - 1;
        const int rem = t % 3;
        if (rem == 1) {                                        // a leading group of one or two steps: the operands start as zeros
            bptt_step(t, sb0, No{}, P2{});
            StepIn x = sb0; sb0 = sb1; sb1 = sb2; sb2 = x;
            t -= 1;
        } else if (rem == 2) {
            bptt_step(t, sb0, No{}, P1{});
            bptt_step(t - 1, sb1, No{}, P2{});
            StepIn x = sb2; sb2 = sb1; sb1 = sb0; sb0 = x;
            t -= 2;
        }
        for (; t >= 3; t -= 3) {
            bptt_step(t, sb0, No{}, P0{});
            bptt_step(t - 1, sb1, No{}, P1{});
            bptt_step(t - 2, sb2, No{}, P2{});
        }
        bptt_step(0, sb0, Yes{}, P2{});
    }
    // ---- weight-gradient partials of this workgroup, in the layout of three reduction jobs ([M*N + M] each):
    // accumulator register r of lane (jl, q) is [gate row 16 wq + 4q + r of the tile][column jl]
    {
        const int F = s.F;
        float* Pih = s.bw_wih + (int64_t)hb * (G3 * F + G3);
        float* Prz = s.bw_wrz + (int64_t)hb * (128 * E + 128);
        float* Pn = s.bw_wn + (int64_t)hb * (E * E + E);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = 16 * wq + 4 * q + r;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                Prz[u * E + 16 * nt + jl] = whh[0][nt][r];
                Prz[(E + u) * E + 16 * nt + jl] = whh[1][nt][r];
                Pn[u * E + 16 * nt + jl] = whh[2][nt][r];
            }
            if (jl < F) {
                Pih[u * F + jl] = wih[0][r];
                Pih[(E + u) * F + jl] = wih[1][r];
                Pih[(2 * E + u) * F + jl] = wih[2][r];
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) { bsum[g] += __shfl_xor(bsum[g], 16, 64); bsum[g] += __shfl_xor(bsum[g], 32, 64); }
        if (q == 0) {
            Pih[G3 * F + j] = bsum[0]; Pih[G3 * F + E + j] = bsum[1]; Pih[G3 * F + 2 * E + j] = bsum[2];
            Prz[128 * E + j] = bsum[0]; Prz[128 * E + E + j] = bsum[1];
            Pn[E * E + j] = bsum[3];
        }
    }
    daw += __shfl_xor(daw, 16, 64); daw += __shfl_xor(daw, 32, 64);
    dab += __shfl_xor(dab, 16, 64); dab += __shfl_xor(dab, 32, 64);
    dlnw += __shfl_xor(dlnw, 16, 64); dlnw += __shfl_xor(dlnw, 32, 64);
    float* pa = s.bw_aw + (int64_t)hb * 65;                    // one partial per workgroup: [Bp/16 * NH][65]
    if (q == 0) { pa[j] = daw; s.bw_ln[(int64_t)hb * E + j] = dlnw; }
    if (wq == 0 && lane == 0) pa[64] = dab;
}

// workgroup `bid` of one group of windows: its (tile, slot), or with args.serial every slot of its tile in turn
template <int RPL>
__device__ __forceinline__ void gru_bwd_group(GruBwdLds& L, const GruBwdArgs& args, int bid) {
    const int n = args.serial ? args.nslots : 1;
    for (int i = 0; i < n; ++i) {
        gru_bwd_body<RPL>(L, args, args.serial ? bid * args.nslots + i : bid);
        __syncthreads();
    }
}

template <int RPL>
__global__ __launch_bounds__(256) void gru_bwd_kernel(GruBwdArgs args) {
    __shared__ __attribute__((aligned(16))) GruBwdLds L;
    gru_bwd_group<RPL>(L, args, (int)blockIdx.x);
}
// Two groups of windows with their own tile modes in ONE launch (workgroups [0, na) the first): the learner's long
// window on 8-sample tiles next to the short ones on 16-sample tiles.  As two launches on two streams they ran side
// by side as well, but cost the step a second launch and three event operations (~6 us of host time each; the step is
// close to host-bound) and a cross-queue hand-off before the reduction.
template <int RA, int RB>
__global__ __launch_bounds__(256) void gru_bwd_pair_kernel(GruBwdArgs a, GruBwdArgs b, int na) {
    __shared__ __attribute__((aligned(16))) GruBwdLds L;
    if ((int)blockIdx.x < na) gru_bwd_group<RA>(L, a, (int)blockIdx.x);
    else gru_bwd_group<RB>(L, b, (int)blockIdx.x - na);
}

struct GruPrepBwdArgs { const float* whh[3]; u32x4* out[3]; };
__global__ __launch_bounds__(256) void gru_prepare_bwd_kernel(GruPrepBwdArgs a) {
    const int tid = blockIdx.x * 256 + threadIdx.x;           // grid = n * 6 blocks of 256
    const int slot = tid / 1536;
    prep_gru_bwd(a.whh[slot], a.out[slot], tid - slot * 1536);
}

// ---- the learner's one-launch prepare (all-GRU network, 192-128-128-A head): prepare_core.hpp: dqn_prepare_block
__global__ __launch_bounds__(256) void dqn_prepare_kernel(DqnPrepArgs a) { dqn_prepare_block(a, blockIdx.x, threadIdx.x); }
int launch_dqn_prepare(const DqnPrepArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(dqn_prepare_kernel, dim3(DQN_PREP_BLOCKS), dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_gru_prepare_bwd(const float* const whh[3], int n, void* const out[3], hipStream_t st) {
    if (n == 0) return 0;
    GruPrepBwdArgs a;
    for (int i = 0; i < 3; ++i) { a.whh[i] = whh[i < n ? i : 0]; a.out[i] = reinterpret_cast<u32x4*>(out[i < n ? i : 0]); }
    hipLaunchKernelGGL(gru_prepare_bwd_kernel, dim3((unsigned)(n * 6)), dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int launch_gru_bwd(const GruBwdArgs& a, hipStream_t st) {
    if (a.nslots == 0) return 0;
    MCTSQ_CHECK(a.Bp % TILE == 0, "padded batch must be a multiple of %d", TILE);
    for (int i = 0; i < a.nslots; ++i) MCTSQ_CHECK(a.slot[i].S <= MAX_SEQ, "window too long");
    for (int i = 0; i < a.nslots; ++i)
        MCTSQ_CHECK(a.slot[i].wprep_t && ((uintptr_t)a.slot[i].wprep_t & 15) == 0, "GRU backward needs the transposed prepared weights");
    const dim3 grid((unsigned)gru_bwd_blocks(a));
    if (a.half) hipLaunchKernelGGL(gru_bwd_kernel<2>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gru_bwd_kernel<4>, grid, dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int64_t gru_bwd_blocks(const GruBwdArgs& a) { return a.Bp / TILE * (a.half ? 2 : 1) * (a.serial ? (a.nslots > 0) : a.nslots); }

int launch_gru_bwd_pair(const GruBwdArgs& a, const GruBwdArgs& b, hipStream_t st) {
    if (a.nslots == 0) return launch_gru_bwd(b, st);
    if (b.nslots == 0) return launch_gru_bwd(a, st);
    const GruBwdArgs* both[2] = {&a, &b};
    for (const GruBwdArgs* g : both) {
        MCTSQ_CHECK(g->Bp % TILE == 0, "padded batch must be a multiple of %d", TILE);
        for (int i = 0; i < g->nslots; ++i) {
            MCTSQ_CHECK(g->slot[i].S <= MAX_SEQ, "window too long");
            MCTSQ_CHECK(g->slot[i].wprep_t && ((uintptr_t)g->slot[i].wprep_t & 15) == 0,
                        "GRU backward needs the transposed prepared weights");
        }
    }
    const int na = (int)gru_bwd_blocks(a), nb = (int)gru_bwd_blocks(b);
    const dim3 grid((unsigned)(na + nb));
    if (a.half && b.half) hipLaunchKernelGGL((gru_bwd_pair_kernel<2, 2>), grid, dim3(256), 0, st, a, b, na);
    else if (a.half) hipLaunchKernelGGL((gru_bwd_pair_kernel<2, 4>), grid, dim3(256), 0, st, a, b, na);
    else if (b.half) hipLaunchKernelGGL((gru_bwd_pair_kernel<4, 2>), grid, dim3(256), 0, st, a, b, na);
    else hipLaunchKernelGGL((gru_bwd_pair_kernel<4, 4>), grid, dim3(256), 0, st, a, b, na);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

// ====================================================================== weight-gradient GEMM
// dW[m][n] = sum_k A[k][m] * Bm[k][n] over a K chunk: block = 4 waves = a 64 x 64 output tile of one
// (job, K split); wave w owns rows 16w..16w+15 of the tile and all four 16-column tiles.
// The chunk is walked in slabs of 32 k-rows staged through LDS: every lane fetches 8 + 8 dwords (each
// wave instruction = one contiguous 256-byte row segment), the next slab's fetch is in flight while the
// current one feeds the MFMAs (double-buffered LDS, one barrier per slab).  Operand reads from LDS are
// conflict-free with the row stride 80 (4 k-rows land in 4 disjoint groups of 16 banks).  Column sums of
// A (bias gradients) ride along as VALU adds.
constexpr int WG_KT = 32, WG_LD = 80;
__global__ __launch_bounds__(256) void wgrad_kernel(WJobList jl, const float* __restrict__ ws,
                                                    float* __restrict__ partials) {
    __shared__ float As[2][WG_KT * WG_LD], Bs[2][WG_KT * WG_LD];
    int ji = 0;
    for (int k = 1; k < jl.njobs; ++k)
        if ((int)blockIdx.x >= jl.job[k].blk0) ji = k;
    const WJob& J = jl.job[ji];
    if (J.a_off < 0) return;                                  // partials written by another kernel
    const int lb = blockIdx.x - J.blk0, per = J.mblocks * J.nblocks;
    const int split = lb / per, rem = lb - split * per, mb = rem / J.nblocks, nb = rem - mb * J.nblocks;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, jl_ = lane & 15, q = lane >> 4;
    const int M = J.M, N = J.N;
    const int mbase = mb * 64, n0 = nb * 64, m0 = mbase + 16 * w;
    const int k0 = split * J.kchunk;
    const int k1 = k0 + J.kchunk < J.K ? k0 + J.kchunk : J.K;
    const bool do_bias = nb == 0;
    const bool has_b = J.b_off >= 0;
    const int ntile = has_b ? min(4, (N - n0 + 15) >> 4) : 0;
    const bool wave_live = m0 < M;                            // wave-uniform
    // staging role of this lane: column c of rows r, r+4, ... of the slab
    const int c = tid & 63, r = tid >> 6;
    const bool a_ok = mbase + c < M, b_ok = has_b && n0 + c < N;
    const float* ag = ws + J.a_off + (int64_t)(k0 + r) * J.lda + mbase + c;
    const float* bg = has_b ? ws + J.b_off + (int64_t)(k0 + r) * J.ldb + n0 + c : nullptr;
    const int64_t astep = 4 * (int64_t)J.lda, bstep = 4 * (int64_t)J.ldb;
    float ra[8], rb[8];
    auto fetch = [&](int kb) {                                // slab starting at k-row kb -> registers
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const bool kin = kb + r + 4 * p < k1;
            ra[p] = (a_ok && kin) ? ag[p * astep] : 0.0f;
            rb[p] = (b_ok && kin) ? bg[p * bstep] : 0.0f;
        }
        ag += 8 * astep;
        if (has_b) bg += 8 * bstep;
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            As[buf][(r + 4 * p) * WG_LD + c] = ra[p];
            Bs[buf][(r + 4 * p) * WG_LD + c] = rb[p];
        }
    };
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float bsum = 0.0f;
    fetch(k0);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int kb = k0; kb < k1; kb += WG_KT) {
        const bool more = kb + WG_KT < k1;
        if (more) fetch(kb + WG_KT);
        if (wave_live) {
            const float* ap = &As[buf][q * WG_LD + 16 * w + jl_];
            const float* bp = &Bs[buf][q * WG_LD + jl_];
#pragma unroll
            for (int ks = 0; ks < WG_KT / 4; ++ks) {
                const float av = ap[ks * 4 * WG_LD];
                if (ntile > 0) acc[0] = mfma16(av, bp[ks * 4 * WG_LD], acc[0]);
                if (ntile > 1) acc[1] = mfma16(av, bp[ks * 4 * WG_LD + 16], acc[1]);
                if (ntile > 2) acc[2] = mfma16(av, bp[ks * 4 * WG_LD + 32], acc[2]);
                if (ntile > 3) acc[3] = mfma16(av, bp[ks * 4 * WG_LD + 48], acc[3]);
                bsum += av;
            }
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (!wave_live) return;
    float* P = partials + J.part_off + (int64_t)split * ((int64_t)M * N + M);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < ntile) {
            const int n = n0 + 16 * t + jl_;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int m = m0 + 4 * q + rr;
                if (m < M && n < N) P[(int64_t)m * N + n] = acc[t][rr];
            }
        }
    }
    if (do_bias) {                                            // column sums of A: fold the four k groups
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (q == 0 && m0 + jl_ < M) P[(int64_t)M * N + m0 + jl_] = bsum;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WJobList jl, const float* __restrict__ partials,
                                                           float* __restrict__ grad, int64_t total, ReduceExtras ex) {
    __shared__ double s_sq[4];
    // the loss: the per-tile partials folded by one wave in a fixed order (lane l takes parts l, l + 64, ..., then a
    // shuffle tree).  One thread adding them in turn was a chain of n dependent loads -- the compiler waits for each
    // before it issues the next --, 64 of them at B = 1024: half of this kernel's time.
    if (ex.loss_parts && blockIdx.x == 0 && threadIdx.x < 64) {
        double t = 0.0;
        for (int i0 = 0; i0 < ex.n_loss_parts; i0 += 256) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 64 * u + threadIdx.x;
                const double x = ex.loss_parts[i < ex.n_loss_parts ? i : 0];
                v[u] = i < ex.n_loss_parts ? x : 0.0;
            }
            t += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        if (threadIdx.x == 0) *ex.loss_out = (float)(t * ex.loss_norm);
    }
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float written = 0.0f;
    if (e < total) {
        // which job: a scan of the compact start table without an early exit, so its scalar loads go out together (a
        // walk over the job records with a break was ~40 dependent fetches from the kernel-argument block: 8 of this
        // kernel's 11 us)
        int ji = 0;
#pragma unroll 8
        for (int k = 1; k < jl.njobs; ++k)
            if (e >= jl.red_e0[k]) ji = k;
        const WJob& J = jl.job[ji];
        const int le = (int)e - jl.red_e0[ji], mn = J.M * J.N, stride = mn + J.M;
        const float* p = partials + J.part_off + le;
        float acc = 0.0f;
        int sp = 0;
        // summed in split order; 32 (then 8) loads in flight: the chain of dependent round trips, not the bytes, is what
        // this kernel costs (160 partials per element of the long window's gradients: 20 rounds of eight took 12 us)
        for (; sp + 32 <= J.nsplit; sp += 32) {
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = p[(int64_t)(sp + u) * stride];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += v[u];
        }
        for (; sp + 8 <= J.nsplit; sp += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(sp + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; sp < J.nsplit; ++sp) acc += p[(int64_t)sp * stride];
        if (le < mn) {
            if (J.out_w >= 0) {
                const int m = (int)((unsigned)le / (unsigned)J.N), n = le - m * J.N;
                grad[J.out_w + m * (int64_t)J.out_ldm + n * (int64_t)J.out_sn] = acc;
                written = acc;
            }
        } else if (J.out_b >= 0) {
            grad[J.out_b + (le - mn)] = acc;
            written = acc;
        }
    }
    if (ex.sumsq_parts) {                                     // this block's share of |grad|^2, fixed order
        double sq = (double)written * (double)written;
        for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d, 64);
        if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6] = sq;
        __syncthreads();
        if (threadIdx.x == 0) ex.sumsq_parts[blockIdx.x] = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]);
    }
}

int launch_wgrad(const WJobList& jl, const float* ws, float* partials, hipStream_t st) {
    if (jl.njobs == 0 || jl.nblocks_total == 0) return 0;
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)jl.nblocks_total), dim3(256), 0, st, jl, ws, partials);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

static int64_t reduce_total(const WJobList& jl) {
    int64_t total = 0;
    for (int i = 0; i < jl.njobs; ++i) total += (int64_t)jl.job[i].M * jl.job[i].N + jl.job[i].M;
    return total;
}
int wgrad_reduce_blocks(const WJobList& jl) { return jl.njobs == 0 ? 0 : (int)((reduce_total(jl) + 255) / 256); }

int launch_wgrad_reduce(const WJobList& jl, const float* partials, float* grad, const ReduceExtras* ex, hipStream_t st) {
    if (jl.njobs == 0) return 0;
    const int64_t total = reduce_total(jl);
    MCTSQ_CHECK(total < ((int64_t)1 << 31), "too many gradient entries for one reduction launch");
    ReduceExtras e;
    if (ex) e = *ex; else memset(&e, 0, sizeof(e));
    WJobList l = jl;
    int64_t pos = 0;
    for (int i = 0; i < l.njobs; ++i) {
        l.red_e0[i] = (int32_t)pos;
        pos += (int64_t)l.job[i].M * l.job[i].N + l.job[i].M;
    }
    for (int i = l.njobs; i <= MAX_WJOBS; ++i) l.red_e0[i] = (int32_t)pos;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, l, partials,
                       grad, total, e);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
