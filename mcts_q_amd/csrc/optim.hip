// clip_grad_norm_(max_norm) + Adam.step + Polyak target update over flat fp32 buffers
// (dqn_per.py:471-472 and :510-516; hard copy of mctsq_config_dqn.py:547 when tau == 1).
// HBM-bound streaming: 5 reads + 4 writes of 4 bytes per parameter, float4-vectorised.
#include "device_utils.hpp"
#include "kernels.hpp"
#include "learner.hpp"

namespace mctsq {

constexpr int NPART = 256;     // partial sums of squares (fixed order -> bitwise reproducible norm)

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n,
                                                            double* __restrict__ part) {
    __shared__ double red[4];
    const int64_t n4 = n >> 2;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)NPART * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v; float* tgt;
    int64_t n;
    float step_size, bc2_sqrt, one_m_b1, b2, one_m_b2, eps, max_norm, tau, one_m_tau;
    const double* part;         // partial sums of squares of the gradient (another launch made them), fixed order
    int npart;
    float* gnorm_out;
    // data-parallel step: the summed gradient (and loss) still carry the global weight maximum; its candidates are
    // the (hi, lo) fp32 pairs of the exchange block.  NULL: scale 1 (every product below is then exact: x * 1.0f).
    const float* wmax_slots; int wmax_world;
    float* g_scaled_out;        // the gradient with the scale applied (unclipped), written back in place, or NULL
    float* loss_io;             // the summed loss, scaled in place by block 0, or NULL
};

__device__ __forceinline__ float exchange_scale(const AdamArgs& a) {
    if (!a.wmax_slots) return 1.0f;
    double mx = 0.0;
    for (int r = 0; r < a.wmax_world; ++r) mx = fmax(mx, (double)a.wmax_slots[2 * r] + (double)a.wmax_slots[2 * r + 1]);
    return mx > 0.0 ? (float)(1.0 / mx) : 0.0f;
}

// FUSED_NORM: every block computes the global sum of squares itself, all in the same fixed order (thread t takes
// the float4s t, t + 256, ...; wave shuffle; four wave sums), instead of reading partials a separate launch made:
// for a parameter buffer that sits in L2 the redundant read costs ~1 us and saves a launch on the step's chain.
constexpr int64_t FUSED_NORM_MAX = 1 << 18;        // floats (1 MB): gru/gru/gru and conv/conv/mlp networks
constexpr int ADAM_THREADS = 1024;
template <bool FUSED_NORM>
__global__ __launch_bounds__(ADAM_THREADS) void clip_adam_polyak_kernel(AdamArgs a) {
    __shared__ float s_coef;
    __shared__ double s_red[ADAM_THREADS / 64];
    const float gs = exchange_scale(a);
    if (a.loss_io && blockIdx.x == 0 && threadIdx.x == 0 && a.wmax_slots) *a.loss_io = *a.loss_io * gs;
    if (FUSED_NORM) {
        const int64_t n4 = a.n >> 2;
        double acc = 0.0;
        for (int64_t i0 = threadIdx.x; i0 < n4; i0 += 8 * ADAM_THREADS) {     // eight loads in flight, summed in index order
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + u * ADAM_THREADS;
                v[u] = i < n4 ? reinterpret_cast<const float4*>(a.g)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc += (double)v[u].x * v[u].x + (double)v[u].y * v[u].y + (double)v[u].z * v[u].z + (double)v[u].w * v[u].w;
        }
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < ADAM_THREADS / 64; ++i) t += s_red[i];
            const float total = (float)sqrt(t) * gs;
            s_coef = fminf(a.max_norm / (total + 1e-6f), 1.0f);
            if (blockIdx.x == 0 && a.gnorm_out) *a.gnorm_out = total;
        }
    } else if (threadIdx.x < 64) {                // every block folds the same 256 partials, same order
        double acc = 0.0;
        for (int i0 = 0; i0 < a.npart; i0 += 512) {           // eight loads in flight (a plain loop waits for each in turn)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u + threadIdx.x;
                const double x = a.part[i < a.npart ? i : 0];
                v[u] = i < a.npart ? x : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
        if (threadIdx.x == 0) {
            const float total = (float)sqrt(acc) * gs;
            s_coef = fminf(a.max_norm / (total + 1e-6f), 1.0f);
            if (blockIdx.x == 0 && a.gnorm_out) *a.gnorm_out = total;
        }
    }
    __syncthreads();
    const float coef = s_coef;
    const int64_t n4 = a.n >> 2;
    // python-float hyper-parameters: 1 - beta is formed in fp64 and THEN rounded, as torch does
    const float one_m_b1 = a.one_m_b1, one_m_b2 = a.one_m_b2;
    const float tau = a.tau, one_m_tau = a.one_m_tau;
    for (int64_t i = (int64_t)blockIdx.x * ADAM_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * ADAM_THREADS) {
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        const float4 g4 = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        float* pp = &p.x; float* pm = &m.x; float* pv = &v.x; const float* pg = &g4.x;
        if (a.g_scaled_out)
            reinterpret_cast<float4*>(a.g_scaled_out)[i] = make_float4(g4.x * gs, g4.y * gs, g4.z * gs, g4.w * gs);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float g = (pg[c] * gs) * coef;
            pm[c] = pm[c] + (g - pm[c]) * one_m_b1;                      // exp_avg.lerp_(g, 1-b1)
            pv[c] = __fadd_rn(__fmul_rn(pv[c], a.b2), __fmul_rn(__fmul_rn(g, g), one_m_b2));
            const float denom = sqrtf(pv[c]) / a.bc2_sqrt + a.eps;
            pp[c] = pp[c] - __fdiv_rn(__fmul_rn(a.step_size, pm[c]), denom);   // addcdiv_: (value * m) / denom
        }
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        if (a.tgt) {
            float4 t = reinterpret_cast<float4*>(a.tgt)[i];
            float* pt = &t.x;
#pragma unroll
            for (int c = 0; c < 4; ++c) pt[c] = __fadd_rn(__fmul_rn(tau, pp[c]), __fmul_rn(one_m_tau, pt[c]));
            reinterpret_cast<float4*>(a.tgt)[i] = t;
        }
    }
}

__global__ __launch_bounds__(256) void polyak_kernel(const float* __restrict__ p, float* __restrict__ tgt,
                                                     int64_t n4, float tau, float one_m_tau) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 s = reinterpret_cast<const float4*>(p)[i];
        float4 t = reinterpret_cast<float4*>(tgt)[i];
        t.x = __fadd_rn(__fmul_rn(tau, s.x), __fmul_rn(one_m_tau, t.x));
        t.y = __fadd_rn(__fmul_rn(tau, s.y), __fmul_rn(one_m_tau, t.y));
        t.z = __fadd_rn(__fmul_rn(tau, s.z), __fmul_rn(one_m_tau, t.z));
        t.w = __fadd_rn(__fmul_rn(tau, s.w), __fmul_rn(one_m_tau, t.w));
        reinterpret_cast<float4*>(tgt)[i] = t;
    }
}

static unsigned stream_grid(int64_t n4) {
    int64_t g = (n4 + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace mctsq

using namespace mctsq;

extern "C" int64_t mctsq_optim_workspace_bytes(void) { return (int64_t)(NPART * sizeof(double)); }

// d_sumsq_parts != NULL: the pieces of the gradient's sum of squares already exist (mctsq_dqn_grad_prepared made
// them while it reduced the gradient), so no pass over the gradient is needed for the norm
extern "C" int mctsq_clip_adam_polyak_ex(float* d_params, const float* d_grad, float* d_m, float* d_v,
                                         float* d_target, int64_t n, int64_t step, double lr, double beta1,
                                         double beta2, double eps, double max_norm, double tau, float* d_gnorm,
                                         const double* d_sumsq_parts, int64_t nparts, void* d_ws, int64_t ws_bytes,
                                         void* stream) {
    return mctsq::clip_adam_polyak_impl(d_params, d_grad, d_m, d_v, d_target, n, step, lr, beta1, beta2, eps, max_norm, tau,
                                        d_gnorm, d_sumsq_parts, nparts, d_ws, ws_bytes, nullptr, 0, nullptr, nullptr, stream);
}

// exchange != NULL: the data-parallel form (mctsq_per_dqn_step_finish): gradient and loss are divided by the largest
// of the `world` exchanged weight maxima inside the pass; the scaled (unclipped) gradient replaces d_grad
int mctsq::clip_adam_polyak_impl(float* d_params, const float* d_grad, float* d_m, float* d_v, float* d_target,
                                 int64_t n, int64_t step, double lr, double beta1, double beta2, double eps,
                                 double max_norm, double tau, float* d_gnorm, const double* d_sumsq_parts,
                                 int64_t nparts, void* d_ws, int64_t ws_bytes, const float* d_exchange, int world,
                                 float* d_grad_scaled, float* d_loss_io, void* stream) {
    MCTSQ_CHECK(d_params && d_grad && d_m && d_v, "NULL device pointer");
    MCTSQ_CHECK(n > 0 && n % 16 == 0, "flat buffer length must be a positive multiple of 16 (padded layout)");
    MCTSQ_CHECK(step >= 1, "Adam step is 1-based");
    MCTSQ_CHECK(d_sumsq_parts ? (nparts >= 1 && nparts < ((int64_t)1 << 30))
                              : (d_ws && ws_bytes >= mctsq_optim_workspace_bytes()),
                "workspace must hold mctsq_optim_workspace_bytes() (or pass the gradient-norm parts)");
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)d_ws;
    const bool fused = !d_sumsq_parts && n <= FUSED_NORM_MAX;
    if (!fused && !d_sumsq_parts) hipLaunchKernelGGL(sumsq_partial_kernel, dim3(NPART), dim3(256), 0, st, d_grad, n, part);
    AdamArgs a;
    a.p = d_params; a.g = d_grad; a.m = d_m; a.v = d_v; a.tgt = tau >= 0.0 ? d_target : nullptr;
    a.n = n;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.one_m_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_m_b2 = (float)(1.0 - beta2);
    a.eps = (float)eps; a.max_norm = (float)max_norm; a.tau = (float)tau; a.one_m_tau = (float)(1.0 - tau);
    a.part = d_sumsq_parts ? d_sumsq_parts : part; a.npart = d_sumsq_parts ? (int)nparts : NPART;
    a.gnorm_out = d_gnorm;
    a.wmax_slots = d_exchange; a.wmax_world = world; a.g_scaled_out = d_exchange ? d_grad_scaled : nullptr;
    a.loss_io = d_exchange ? d_loss_io : nullptr;
    MCTSQ_CHECK(!d_exchange || (world >= 1 && world <= 64), "bad exchange geometry");
    MCTSQ_CHECK(tau < 0.0 || d_target, "target buffer is NULL but tau >= 0");
    const unsigned agrid = (stream_grid(n >> 2) * 256 + ADAM_THREADS - 1) / ADAM_THREADS;
    if (fused) hipLaunchKernelGGL(clip_adam_polyak_kernel<true>, dim3(agrid), dim3(ADAM_THREADS), 0, st, a);
    else hipLaunchKernelGGL(clip_adam_polyak_kernel<false>, dim3(agrid), dim3(ADAM_THREADS), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

extern "C" int mctsq_clip_adam_polyak(float* d_params, const float* d_grad, float* d_m, float* d_v,
                                      float* d_target, int64_t n, int64_t step, double lr, double beta1,
                                      double beta2, double eps, double max_norm, double tau, float* d_gnorm,
                                      void* d_ws, int64_t ws_bytes, void* stream) {
    return mctsq_clip_adam_polyak_ex(d_params, d_grad, d_m, d_v, d_target, n, step, lr, beta1, beta2, eps, max_norm, tau,
                                     d_gnorm, nullptr, 0, d_ws, ws_bytes, stream);
}

extern "C" int mctsq_polyak(const float* d_params, float* d_target, int64_t n, double tau, void* stream) {
    MCTSQ_CHECK(d_params && d_target, "NULL device pointer");
    MCTSQ_CHECK(n > 0 && n % 16 == 0, "flat buffer length must be a positive multiple of 16 (padded layout)");
    hipLaunchKernelGGL(polyak_kernel, dim3(stream_grid(n >> 2)), dim3(256), 0, (hipStream_t)stream, d_params,
                       d_target, n >> 2, (float)tau, (float)(1.0 - tau));
    MCTSQ_HIP(hipGetLastError());
    return 0;
}
