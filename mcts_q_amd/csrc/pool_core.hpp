// Attention pooling over time + LayerNorm, shared by the encoders whose per-step vector lives in
// the MFMA D layout (wave wq owns units 16wq..16wq+15, lane register r is sample 4q+r):
// softmax over time evaluated online in base 2, then LayerNorm over the 64 units (4 waves).
// dqn_per.py:50-55 (conv) and :74-77 (GRU) - the same arithmetic; gru_core.hpp carries its own copy
// inside the tuned recurrence.
#pragma once
#include "device_utils.hpp"

namespace mctsq {

struct PoolLds {
    float spart[2][4][TILE];
    float red[2][4][TILE];
    float dump[4][64 * 4];              // write-only sink of the non-publishing lanes (publish)
};

template <bool SAVE>
struct PoolCore {
    float h[4], att[4], mrun[4], lrun[4], pooled[4], mean[4], rstd[4];
    float awj, ab;                      // score weights, pre-multiplied by log2(e)
    int wq, jl, q, j;

    __device__ __forceinline__ void init(const float* aw, const float* abp, int lane, int wq_) {
        wq = wq_; jl = lane & 15; q = lane >> 4; j = 16 * wq + jl;
        awj = aw[j] * LOG2E; ab = abp[0] * LOG2E;
#pragma unroll
        for (int r = 0; r < 4; ++r) { h[r] = 0.0f; att[r] = 0.0f; lrun[r] = 0.0f; mrun[r] = -INFINITY; }
    }
    // publish the score partials of the vector v (becomes h) into spart[buf]
    __device__ __forceinline__ void publish(PoolLds& L, int buf, const float v[4]) {
        float4 ps;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = v[r];
        const f32x2 a01 = (f32x2){v[0], v[1]} * awj, a23 = (f32x2){v[2], v[3]} * awj;      // packed pairs of rows
        ps.x = row16_sum(a01.x); ps.y = row16_sum(a01.y); ps.z = row16_sum(a23.x); ps.w = row16_sum(a23.y);
        // one lane per row group stores its four sums as 16 bytes; the others write theirs to a slot of their own:
        // no branch around the store (gru_core.hpp: GruWave::publish)
        float* dst = jl == 0 ? &L.spart[buf][wq][4 * q] : &L.dump[wq][(16 * q + jl) * 4];
        *reinterpret_cast<float4*>(dst) = ps;
    }
    // fold h (whose score parts are in spart[buf]) into the running sums; call after a barrier
    __device__ __forceinline__ void attend(PoolLds& L, int buf, int t, float* sv_score, int64_t row0, int64_t Bp) {
        float4 sp[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) sp[w] = *reinterpret_cast<const float4*>(&L.spart[buf][w][4 * q]);
        // the four waves' partial scores summed on packed pairs of rows
        const f32x2 s01 = (((f32x2){sp[0].x, sp[0].y} + (f32x2){sp[1].x, sp[1].y}) + ((f32x2){sp[2].x, sp[2].y} + (f32x2){sp[3].x, sp[3].y})) + ab;
        const f32x2 s23 = (((f32x2){sp[0].z, sp[0].w} + (f32x2){sp[1].z, sp[1].w}) + ((f32x2){sp[2].z, sp[2].w} + (f32x2){sp[3].z, sp[3].w})) + ab;
        const float scs[4] = {s01.x, s01.y, s23.x, s23.y};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            const float sc = scs[r];
            if (SAVE && wq == 0 && jl == 0) sv_score[(int64_t)t * Bp + row0 + i] = sc;
            const float d = sc - mrun[r];
            const float e = fast_exp2(-fabsf(d));
            const bool up = d > 0.0f;
            const float c_old = up ? e : 1.0f, c_new = up ? 1.0f : e;
            att[r] = att[r] * c_old + c_new * h[r];
            lrun[r] = lrun[r] * c_old + c_new;
            mrun[r] = up ? sc : mrun[r];
        }
    }
    __device__ __forceinline__ void fin0(PoolLds& L) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pooled[r] = att[r] / lrun[r];
            const float ssum = row16_sum(pooled[r]);
            if (jl == 0) L.red[0][wq][4 * q + r] = ssum;
        }
    }
    __device__ __forceinline__ void fin1(PoolLds& L) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            mean[r] = ((L.red[0][0][i] + L.red[0][1][i]) + (L.red[0][2][i] + L.red[0][3][i])) * (1.0f / E);
            const float d = pooled[r] - mean[r];
            const float vs = row16_sum(d * d);
            if (jl == 0) L.red[1][wq][i] = vs;
        }
    }
    __device__ __forceinline__ void fin2(PoolLds& L, const float* lnwp, const float* lnbp, float* out, int ld,
                                         float* sv_pool, float* sv_stat, int64_t row0) {
        const float lnw = lnwp[j], lnb = lnbp[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            const float var = ((L.red[1][0][i] + L.red[1][1][i]) + (L.red[1][2][i] + L.red[1][3][i])) * (1.0f / E);
            rstd[r] = 1.0f / sqrtf(var + LN_EPS);
            out[i * ld + j] = (pooled[r] - mean[r]) * rstd[r] * lnw + lnb;
            if (SAVE) {
                sv_pool[(row0 + i) * (int64_t)E + j] = pooled[r];
                if (wq == 0 && jl == 0) {
                    float* st = sv_stat + (row0 + i) * 4;
                    st[0] = mrun[r]; st[1] = lrun[r]; st[2] = mean[r]; st[3] = rstd[r];
                }
            }
        }
    }
};

// Backward of LayerNorm + attention pooling up to the per-step terms (learner side).
// After prepare(): dpool, D = dpool . pooled, and the softmax statistics are in registers; the
// caller supplies e_t = dpool . v_t per step (cross-wave partials through LDS).
struct PoolBwdLds {
    float4 pe[MAX_SEQ][TILE];
    float red[3][4][TILE];
};

struct PoolBwdCore {
    float dpool[4], pooled[4], mrun[4], linv[4], Dd[4];
    int wq, jl, q, j;

    // needs two barriers inside: call from all 256 threads
    __device__ __forceinline__ void prepare(PoolBwdLds& L, const float* dfeat, int out_col, const float* lnwp,
                                            const float* sv_stat, const float* sv_pool, float* bw_p, int lane,
                                            int wq_, int64_t row0) {
        wq = wq_; jl = lane & 15; q = lane >> 4; j = 16 * wq + jl;
        const float lnw = lnwp[j];
        float dxh[4], xh[4], rs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 4 * q + r;
            const float4 st = *reinterpret_cast<const float4*>(sv_stat + row * 4);
            mrun[r] = st.x; linv[r] = 1.0f / st.y; rs[r] = st.w;
            pooled[r] = sv_pool[row * E + j];
            const float df = dfeat[row * FEAT + out_col + j];
            xh[r] = (pooled[r] - st.z) * st.w;
            dxh[r] = df * lnw;
            bw_p[row * E + j] = df * xh[r];
            const float s1 = row16_sum(dxh[r]), s2 = row16_sum(dxh[r] * xh[r]);
            if (jl == 0) { L.red[0][wq][4 * q + r] = s1; L.red[1][wq][4 * q + r] = s2; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            const float s1 = ((L.red[0][0][i] + L.red[0][1][i]) + (L.red[0][2][i] + L.red[0][3][i])) * (1.0f / E);
            const float s2 = ((L.red[1][0][i] + L.red[1][1][i]) + (L.red[1][2][i] + L.red[1][3][i])) * (1.0f / E);
            dpool[r] = rs[r] * (dxh[r] - s1 - xh[r] * s2);
            const float v = row16_sum(dpool[r] * pooled[r]);
            if (jl == 0) L.red[2][wq][i] = v;
        }
    }
    // partial of e_t for the vector v (this wave's 16 units)
    __device__ __forceinline__ void publish_e(PoolBwdLds& L, int t, const float v[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = row16_sum(dpool[r] * v[r]);
            if (jl == 0) (&L.pe[t][4 * q + r].x)[wq] = s;
        }
    }
    // after the barrier that follows the publish_e calls
    __device__ __forceinline__ void finish(PoolBwdLds& L) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * q + r;
            Dd[r] = (L.red[2][0][i] + L.red[2][1][i]) + (L.red[2][2][i] + L.red[2][3][i]);
        }
    }
    // softmax weight a_t and d(score_t) for register r
    __device__ __forceinline__ void step_terms(PoolBwdLds& L, int t, int r, float sc, float& at, float& dsc) {
        at = fast_exp2(sc - mrun[r]) * linv[r];
        const float4 pe = L.pe[t][4 * q + r];
        dsc = at * (((pe.x + pe.y) + (pe.z + pe.w)) - Dd[r]);
    }
};

}  // namespace mctsq
