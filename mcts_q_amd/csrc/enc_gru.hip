// GRUAttentionEnc forward (dqn_per.py:58-77) as one fused gfx950 kernel:
//   GRU recurrence (gate order r,z,n; h0 = 0) -> score = w.h_t + b -> softmax over time ->
//   sum_t a_t h_t -> LayerNorm(64).
// Mapping: one workgroup = 16 samples (MFMA M), 4 waves; wave w owns hidden units 16w..16w+15.
// Per step every wave computes its 3 gate tiles [16 samples x 16 units] = h_{t-1}[16x64] . W_hh^T
// with 48 v_mfma_f32_16x16x4_f32 (exact fp32), W_hh fragments resident in 48 VGPRs for the whole
// kernel, h exchanged through a double-buffered LDS tile (one barrier per step).  The attention
// softmax is evaluated online (running max / sum) so no [S,16,64] history is kept on chip.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace mctsq {

constexpr int HLD = 68;    // LDS row stride of the h tile (floats): 64 + 4 keeps b128 reads aligned
constexpr int XLD = MAX_SEQ * MAX_F / 2 + 4;   // 132 >= max S*F (25*3, 10*8) rounded up

template <bool SAVE>
__global__ __launch_bounds__(256) void gru_enc_fwd_kernel(GruFwdArgs args) {
    const GruSlot& s = args.slot[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int jl = lane & 15, q = lane >> 4;
    const int j = 16 * w + jl;                 // hidden unit owned by this lane
    const int64_t row0 = (int64_t)blockIdx.x * TILE;
    const int64_t B = args.B, Bp = args.Bp;
    const int S = s.S, F = s.F, Fd = s.Fd;

    __shared__ __attribute__((aligned(16))) float hbuf[2][TILE * HLD];
    __shared__ __attribute__((aligned(16))) float spart[2][4][TILE];
    __shared__ __attribute__((aligned(16))) float xs[TILE * XLD];
    __shared__ __attribute__((aligned(16))) float red[2][4][TILE];

    // ---- resident weights --------------------------------------------------------------
    float4 wr[4], wz[4], wn[4];               // W_hh[g*64 + j][q*16 .. q*16+15]
    {
        const float4* pr = reinterpret_cast<const float4*>(s.whh + (int64_t)(0 * E + j) * E + q * 16);
        const float4* pz = reinterpret_cast<const float4*>(s.whh + (int64_t)(1 * E + j) * E + q * 16);
        const float4* pn = reinterpret_cast<const float4*>(s.whh + (int64_t)(2 * E + j) * E + q * 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) { wr[c] = pr[c]; wz[c] = pz[c]; wn[c] = pn[c]; }
    }
    float wir[MAX_F], wiz[MAX_F], win[MAX_F];
#pragma unroll
    for (int f = 0; f < MAX_F; ++f) {
        const bool ok = f < F;
        wir[f] = ok ? s.wih[(int64_t)(0 * E + j) * F + f] : 0.0f;
        wiz[f] = ok ? s.wih[(int64_t)(1 * E + j) * F + f] : 0.0f;
        win[f] = ok ? s.wih[(int64_t)(2 * E + j) * F + f] : 0.0f;
    }
    const float bir = s.bih[j], biz = s.bih[E + j], bin = s.bih[2 * E + j];
    const float bhr = s.bhh[j], bhz = s.bhh[E + j], bhn = s.bhh[2 * E + j];
    const float awj = s.aw[j], ab = s.ab[0];

    // ---- stage the input windows: xs[i][t*F + f] ---------------------------------------------
    const int SF = S * F;
    for (int idx = tid; idx < TILE * SF; idx += 256) {
        const int i = idx / SF, rem = idx - i * SF, t = rem / F, f = rem - t * F;
        float v = 0.0f;
        if (row0 + i < B)
            v = f < Fd ? s.x[(row0 + i) * (int64_t)(S * Fd) + t * Fd + f] : time_feature(s.tf, t, f - Fd);
        xs[i * XLD + rem] = v;
    }
    for (int idx = tid; idx < TILE * HLD; idx += 256) hbuf[0][idx] = 0.0f;
    __syncthreads();

    float h[4] = {0, 0, 0, 0};                // h_{t-1}[4q+r][j]
    float att[4] = {0, 0, 0, 0};              // online sum_t exp(s_t - m) h_t
    float mrun[4], lrun[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) mrun[r] = -INFINITY;

    auto attend = [&](int buf, int t) {       // fold h (= h_t, score parts in spart[buf]) into att
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            const float sc = ((spart[buf][0][i] + spart[buf][1][i]) + (spart[buf][2][i] + spart[buf][3][i])) + ab;
            if (SAVE && w == 0 && jl == 0) s.sv_score[(int64_t)t * Bp + row0 + i] = sc;
            const float mn = fmaxf(mrun[r], sc);
            const float c_old = expf(mrun[r] - mn);      // exp(-inf) = 0 on the first step
            const float c_new = expf(sc - mn);
            att[r] = att[r] * c_old + c_new * h[r];
            lrun[r] = lrun[r] * c_old + c_new;
            mrun[r] = mn;
        }
    };

    for (int t = 0; t < S; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        // A fragments: h_{t-1}[sample lane&15][k = q*16 + 0..15]
        const float4* ha = reinterpret_cast<const float4*>(&hbuf[cur][(lane & 15) * HLD + q * 16]);
        const float4 a0 = ha[0], a1 = ha[1], a2 = ha[2], a3 = ha[3];
        if (t > 0) attend(cur, t - 1);        // overlaps with the LDS reads above
        f32x4 accr = {0, 0, 0, 0}, accz = {0, 0, 0, 0}, accn = {0, 0, 0, 0};
#define GRU_STEP4(A, C)                                  \
        accr = mfma16(A.x, wr[C].x, accr); accz = mfma16(A.x, wz[C].x, accz); accn = mfma16(A.x, wn[C].x, accn); \
        accr = mfma16(A.y, wr[C].y, accr); accz = mfma16(A.y, wz[C].y, accz); accn = mfma16(A.y, wn[C].y, accn); \
        accr = mfma16(A.z, wr[C].z, accr); accz = mfma16(A.z, wz[C].z, accz); accn = mfma16(A.z, wn[C].z, accn); \
        accr = mfma16(A.w, wr[C].w, accr); accz = mfma16(A.w, wz[C].w, accz); accn = mfma16(A.w, wn[C].w, accn);
        GRU_STEP4(a0, 0) GRU_STEP4(a1, 1) GRU_STEP4(a2, 2) GRU_STEP4(a3, 3)
#undef GRU_STEP4
        float ps[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            float gir = bir, giz = biz, gin = bin;
            const float* xr = &xs[i * XLD + t * F];
#pragma unroll
            for (int f = 0; f < MAX_F; ++f)
                if (f < F) {
                    const float xv = xr[f];
                    gir = fmaf(wir[f], xv, gir); giz = fmaf(wiz[f], xv, giz); gin = fmaf(win[f], xv, gin);
                }
            const float rg = sigmoidf_acc(gir + (accr[r] + bhr));
            const float zg = sigmoidf_acc(giz + (accz[r] + bhz));
            const float hn = accn[r] + bhn;
            const float ng = tanhf(gin + rg * hn);
            const float hnew = (1.0f - zg) * ng + zg * h[r];
            if (SAVE) {
                const int64_t o = ((int64_t)t * Bp + row0 + i) * E + j;
                s.sv_r[o] = rg; s.sv_z[o] = zg; s.sv_n[o] = ng; s.sv_hn[o] = hn; s.sv_h[o] = hnew;
            }
            h[r] = hnew;
            hbuf[nxt][i * HLD + j] = hnew;
            ps[r] = row16_sum(awj * hnew);
        }
        if (jl == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) spart[nxt][w][4 * q + r] = ps[r];
        }
        __syncthreads();
    }
    attend(S & 1, S - 1);

    // ---- pooled = att / l ; LayerNorm over the 64 hidden units (4 waves x 16 lanes) -------
    float pooled[4], mean[4], rstd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pooled[r] = att[r] / lrun[r];
        const float ssum = row16_sum(pooled[r]);
        if (jl == 0) red[0][w][4 * q + r] = ssum;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * q + r;
        mean[r] = ((red[0][0][i] + red[0][1][i]) + (red[0][2][i] + red[0][3][i])) * (1.0f / E);
        const float d = pooled[r] - mean[r];
        const float vs = row16_sum(d * d);
        if (jl == 0) red[1][w][i] = vs;
    }
    __syncthreads();
    const float lnw = s.lnw[j], lnb = s.lnb[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * q + r;
        const float var = ((red[1][0][i] + red[1][1][i]) + (red[1][2][i] + red[1][3][i])) * (1.0f / E);
        rstd[r] = 1.0f / sqrtf(var + LN_EPS);
        const float y = (pooled[r] - mean[r]) * rstd[r] * lnw + lnb;
        args.feats[(row0 + i) * (int64_t)FEAT + s.out_col + j] = y;     // feats has Bp rows
        if (SAVE) {
            s.sv_pool[(row0 + i) * (int64_t)E + j] = pooled[r];
            if (w == 0 && jl == 0) {
                float* st = s.sv_stat + (row0 + i) * 4;
                st[0] = mrun[r]; st[1] = lrun[r]; st[2] = mean[r]; st[3] = rstd[r];
            }
        }
    }
}

int launch_gru_fwd(const GruFwdArgs& args, bool save, hipStream_t st) {
    if (args.nslots == 0 || args.B == 0) return 0;
    for (int i = 0; i < args.nslots; ++i)
        MCTSQ_CHECK(args.slot[i].S * args.slot[i].F <= XLD && args.slot[i].F <= MAX_F,
                    "GRU encoder window %dx%d does not fit the staged tile", args.slot[i].S, args.slot[i].F);
    MCTSQ_CHECK(args.Bp % TILE == 0 && args.Bp >= args.B, "padded batch must be a multiple of %d", TILE);
    dim3 grid((unsigned)(args.Bp / TILE), (unsigned)args.nslots);
    if (save) hipLaunchKernelGGL(gru_enc_fwd_kernel<true>, grid, dim3(256), 0, st, args);
    else hipLaunchKernelGGL(gru_enc_fwd_kernel<false>, grid, dim3(256), 0, st, args);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
