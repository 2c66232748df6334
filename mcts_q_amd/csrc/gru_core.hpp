// GRU-attention encoder building block shared by the stand-alone encoder kernel (enc_gru.hip) and
// the fused three-encoder + head kernel (fused_gru_head.hip).
//
// One "group" = 4 waves (256 lanes) advancing one 16-sample tile of one encoder slot; wave wq of the
// group owns hidden units 16wq..16wq+15.  Per step every wave computes its 3 gate tiles
// [16 samples x 16 units] = [x_t | h_{t-1}] . [W_ih | W_hh]^T with v_mfma_f32_16x16x4_f32 (exact fp32),
// W fragments resident in VGPRs, h exchanged through a double-buffered LDS tile (one barrier per
// step, issued by the caller).  The softmax over time is evaluated online (running max / sum, in
// base 2) so no [S,16,64] history is kept on chip.  Reference: dqn_per.py:58-77.
#pragma once
#include "device_utils.hpp"
#include "kernels.hpp"

namespace mctsq {

constexpr int HLD = 68;     // LDS row stride of the h tile (floats): 64 + 4 keeps b128 reads aligned
constexpr int XLD = 132;    // LDS row stride of the staged window: >= S * (4 or 8)

struct GruLds {             // per-group LDS carve-out
    float hbuf[2][TILE * HLD];
    float xs[TILE * XLD];
    float spart[2][4][TILE];
    float red[2][4][TILE];
};

template <bool SAVE>
struct GruWave {
    float4 wr[4], wz[4], wn[4];     // W_hh[g*64 + j][q*16 .. q*16+15]
    float wir[2], wiz[2], win[2];   // W_ih[g*64 + j][q (+4)], zero padded
    float br, bz, bin, bhn, awj, ab;
    float h[4], att[4], mrun[4], lrun[4];
    float pooled[4], mean[4], rstd[4];
    int lane, wq, jl, q, j, FP;

    __device__ __forceinline__ void init(const GruSlot& s, int lane_, int wq_) {
        lane = lane_; wq = wq_; jl = lane & 15; q = lane >> 4; j = 16 * wq + jl;
        const int F = s.F;
        FP = F > 4 ? 8 : 4;
        const float4* pr = reinterpret_cast<const float4*>(s.whh + (int64_t)(0 * E + j) * E + q * 16);
        const float4* pz = reinterpret_cast<const float4*>(s.whh + (int64_t)(1 * E + j) * E + q * 16);
        const float4* pn = reinterpret_cast<const float4*>(s.whh + (int64_t)(2 * E + j) * E + q * 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) { wr[c] = pr[c]; wz[c] = pz[c]; wn[c] = pn[c]; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k = q + 4 * ks;
            const bool ok = k < F;
            wir[ks] = ok ? s.wih[(int64_t)(0 * E + j) * F + k] : 0.0f;
            wiz[ks] = ok ? s.wih[(int64_t)(1 * E + j) * F + k] : 0.0f;
            win[ks] = ok ? s.wih[(int64_t)(2 * E + j) * F + k] : 0.0f;
        }
        br = s.bih[j] + s.bhh[j]; bz = s.bih[E + j] + s.bhh[E + j];
        bin = s.bih[2 * E + j]; bhn = s.bhh[2 * E + j];
        awj = s.aw[j] * LOG2E; ab = s.ab[0] * LOG2E;          // softmax evaluated in base 2
#pragma unroll
        for (int r = 0; r < 4; ++r) { h[r] = 0.0f; att[r] = 0.0f; lrun[r] = 0.0f; mrun[r] = -INFINITY; }
    }

    // cooperative over the group's 256 lanes (gtid): xs[i][t*FP + f] (zero padded), hbuf[0] = 0
    __device__ __forceinline__ void stage(const GruSlot& s, GruLds& L, int gtid, int64_t row0, int64_t B, int64_t Bp) {
        const int S = s.S, F = s.F, Fd = s.Fd, SF = S * FP;
        for (int idx = gtid; idx < TILE * SF; idx += 256) {
            const int i = idx / SF, rem = idx - i * SF, t = rem / FP, f = rem - t * FP;
            float v = 0.0f;
            if (row0 + i < B && f < F)
                v = f < Fd ? s.x[(row0 + i) * (int64_t)(S * Fd) + t * Fd + f] : time_feature(s.tf, t, f - Fd);
            L.xs[i * XLD + rem] = v;
            if (SAVE && f < 8) s.sv_x[((int64_t)t * Bp + row0 + i) * 8 + f] = v;
        }
        if (SAVE) {
            if (FP == 4)
                for (int idx = gtid; idx < TILE * S * 4; idx += 256) {
                    const int i = idx / (S * 4), rem = idx - i * (S * 4), t = rem >> 2, f = 4 + (rem & 3);
                    s.sv_x[((int64_t)t * Bp + row0 + i) * 8 + f] = 0.0f;
                }
            for (int idx = gtid; idx < TILE * E; idx += 256) s.sv_h[row0 * E + idx] = 0.0f;   // h_{-1}
        }
        for (int idx = gtid; idx < TILE * HLD; idx += 256) L.hbuf[0][idx] = 0.0f;
    }

    // fold h (= h_t, its score parts are in spart[buf]) into the running attention sums
    __device__ __forceinline__ void attend(const GruSlot& s, GruLds& L, int buf, int t, int64_t row0, int64_t Bp) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            const float sc = ((L.spart[buf][0][i] + L.spart[buf][1][i]) + (L.spart[buf][2][i] + L.spart[buf][3][i])) + ab;
            if (SAVE && wq == 0 && jl == 0) s.sv_score[(int64_t)t * Bp + row0 + i] = sc;   // log2 units
            // online softmax: one of the two rescale factors is always exp2(0) = 1
            const float d = sc - mrun[r];                     // +inf on the first step
            const float e = fast_exp2(-fabsf(d));             // exp2(-inf) = 0
            const bool up = d > 0.0f;
            const float c_old = up ? e : 1.0f, c_new = up ? 1.0f : e;
            att[r] = att[r] * c_old + c_new * h[r];
            lrun[r] = lrun[r] * c_old + c_new;
            mrun[r] = up ? sc : mrun[r];
        }
    }

    // one recurrence step; the caller issues the barrier afterwards
    __device__ __forceinline__ void step(const GruSlot& s, GruLds& L, int t, int64_t row0, int64_t Bp) {
        const int cur = t & 1, nxt = cur ^ 1;
        const float4* ha = reinterpret_cast<const float4*>(&L.hbuf[cur][(lane & 15) * HLD + q * 16]);
        const float4 a0 = ha[0], a1 = ha[1], a2 = ha[2], a3 = ha[3];
        const float* xa = &L.xs[(lane & 15) * XLD + t * FP + q];
        const float x0 = xa[0];
        f32x4 accr = {0, 0, 0, 0}, accz = {0, 0, 0, 0}, accn = {0, 0, 0, 0}, acci = {0, 0, 0, 0};
        accr = mfma16(x0, wir[0], accr); accz = mfma16(x0, wiz[0], accz); acci = mfma16(x0, win[0], acci);
        if (FP == 8) {                                        // wave-uniform
            const float x1 = xa[4];
            accr = mfma16(x1, wir[1], accr); accz = mfma16(x1, wiz[1], accz); acci = mfma16(x1, win[1], acci);
        }
        if (t > 0) attend(s, L, cur, t - 1, row0, Bp);
#define GRU_STEP4(A, C)                                                                                      \
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
            const float rg = sigmoid_fast(accr[r] + br);
            const float zg = sigmoid_fast(accz[r] + bz);
            const float hn = accn[r] + bhn;
            const float ng = tanh_fast((acci[r] + bin) + rg * hn);
            const float hnew = ng + zg * (h[r] - ng);
            if (SAVE) {
                const int64_t o = ((int64_t)t * Bp + row0 + i) * E + j;
                s.sv_r[o] = rg; s.sv_z[o] = zg; s.sv_n[o] = ng; s.sv_hn[o] = hn; s.sv_h[o + Bp * E] = hnew;
            }
            h[r] = hnew;
            L.hbuf[nxt][i * HLD + j] = hnew;
            ps[r] = row16_sum(awj * hnew);
        }
        if (jl == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) L.spart[nxt][wq][4 * q + r] = ps[r];
        }
    }

    // LayerNorm over the 64 hidden units (4 waves x 16 lanes) in three barrier-separated phases
    __device__ __forceinline__ void fin0(GruLds& L) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pooled[r] = att[r] / lrun[r];
            const float ssum = row16_sum(pooled[r]);
            if (jl == 0) L.red[0][wq][4 * q + r] = ssum;
        }
    }
    __device__ __forceinline__ void fin1(GruLds& L) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            mean[r] = ((L.red[0][0][i] + L.red[0][1][i]) + (L.red[0][2][i] + L.red[0][3][i])) * (1.0f / E);
            const float d = pooled[r] - mean[r];
            const float vs = row16_sum(d * d);
            if (jl == 0) L.red[1][wq][i] = vs;
        }
    }
    // writes the normalised feature to out[i * ld + j] (LDS or global) and the training record
    __device__ __forceinline__ void fin2(const GruSlot& s, GruLds& L, float* out, int ld, int64_t row0) {
        const float lnw = s.lnw[j], lnb = s.lnb[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            const float var = ((L.red[1][0][i] + L.red[1][1][i]) + (L.red[1][2][i] + L.red[1][3][i])) * (1.0f / E);
            rstd[r] = 1.0f / sqrtf(var + LN_EPS);
            out[i * ld + j] = (pooled[r] - mean[r]) * rstd[r] * lnw + lnb;
            if (SAVE) {
                s.sv_pool[(row0 + i) * (int64_t)E + j] = pooled[r];
                if (wq == 0 && jl == 0) {
                    float* st = s.sv_stat + (row0 + i) * 4;
                    st[0] = mrun[r]; st[1] = lrun[r]; st[2] = mean[r]; st[3] = rstd[r];
                }
            }
        }
    }
};

}  // namespace mctsq
