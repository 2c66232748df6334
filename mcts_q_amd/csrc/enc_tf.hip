// TransformerEnc (dqn_per.py:80-101) forward: Linear embed + learned positional table, two post-LN
// nn.TransformerEncoderLayer(d=64, 4 heads, ff=2048, ReLU, dropout off), mean over time, LayerNorm.
//
// Tokens are rows: token = sample*S + t, activations [tokens][64] fp32 in HBM between the stages
//   embed -> { attention block -> FFN block } x 2 -> mean-pool + LayerNorm.
// The FFN (94 % of the FLOPs) is the MFMA kernel that matters: see tf_ffn_kernel.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace mctsq {

constexpr int FF = 2048;
constexpr int TOK_PER_WG = 256;

// -------------------------------------------------------------------------------- embedding
// One workgroup = 64 tokens x 64 columns; a thread keeps FOUR consecutive columns of the embedding weights in registers
// and walks 4 consecutive tokens (one division per thread, (sample, step) advanced incrementally): every store is 16
// bytes per lane, 1 KB per wave instruction (with one column per thread - 4-byte stores, 16 tokens per workgroup - the
// 16384-leaf launch ran at 1.6-2.3 TB/s of stores).
constexpr int EMB_TOK = 64;
static_assert(TOK_PER_WG % EMB_TOK == 0, "token padding covers whole embedding workgroups");
__device__ __forceinline__ void tf_embed_body(const TfArgs& a, int bid) {
    const int c4 = (threadIdx.x & 15) * 4;                                      // first of this thread's four columns
    const int64_t tok0 = (int64_t)bid * EMB_TOK + (threadIdx.x >> 4) * 4;       // < Tp
    float w[4][MAX_F];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int f = 0; f < MAX_F; ++f) w[j][f] = f < a.F ? a.ew[(c4 + j) * a.F + f] : 0.0f;
    const float4 bias = make_float4(a.eb[c4], a.eb[c4 + 1], a.eb[c4 + 2], a.eb[c4 + 3]);   // parameters are 4-byte aligned only
    int64_t b = tok0 / a.S;
    int t = (int)(tok0 - b * a.S);
    const int64_t nvalid = a.Bvalid * a.S;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t tok = tok0 + u;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (tok < nvalid) {
            v = bias;
#pragma unroll
            for (int f = 0; f < MAX_F; ++f) {
                if (f < a.F) {
                    const float x = f < a.Fd ? a.x[tok * a.Fd + f] : time_feature(a.tf, t, f - a.Fd);
                    v.x = fmaf(x, w[0][f], v.x); v.y = fmaf(x, w[1][f], v.y);
                    v.z = fmaf(x, w[2][f], v.z); v.w = fmaf(x, w[3][f], v.w);
                    if (a.sv_xe && c4 == 0) a.sv_xe[tok * 8 + f] = x;
                }
            }
            if (a.sv_xe && c4 == 4) {                              // columns F .. 7 of the record are zero
#pragma unroll
                for (int f = 0; f < 8; ++f)
                    if (f >= a.F) a.sv_xe[tok * 8 + f] = 0.0f;
            }
            const float* p = a.pos + t * E + c4;
            v.x += p[0]; v.y += p[1]; v.z += p[2]; v.w += p[3];
        } else if (a.sv_xe && c4 < 8) {
            *reinterpret_cast<float4*>(a.sv_xe + tok * 8 + c4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        *reinterpret_cast<float4*>(a.buf[0].hin + tok * E + c4) = v;
        if (++t == a.S) { t = 0; ++b; }
    }
}

__global__ __launch_bounds__(256) void tf_embed_kernel(TfArgs a) { tf_embed_body(a, (int)blockIdx.x); }
// two encoders of one network as one grid (as tf_ffn_pair_kernel)
__global__ __launch_bounds__(256) void tf_embed_pair_kernel(TfArgs a, TfArgs b, int na) {
    if ((int)blockIdx.x < na) tf_embed_body(a, (int)blockIdx.x);
    else tf_embed_body(b, (int)blockIdx.x - na);
}

// LayerNorm over the 64 columns of rows held in the MFMA D layout: x[c][r] is column 16c + jl of
// row 4q + r; the reduction runs over c (registers) and the 16 lanes of the row group.
// Training: sv_x ([rows][64], row stride 64, pointing at the tile's first row, NULL to skip) keeps
// the pre-norm sum and sv_st ([rows][2]) its (mean, rstd); rows >= nrows are not written.
__device__ __forceinline__ void ln_rows(float x[4][4], const float* lnw, const float* lnb, int jl, int q,
                                        float* sv_x = nullptr, float* sv_st = nullptr, int nrows = 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mean = row16_sum((x[0][r] + x[1][r]) + (x[2][r] + x[3][r])) * (1.0f / E);
        float d[4], ss = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) { d[c] = x[c][r] - mean; ss += d[c] * d[c]; }
        const float rstd = 1.0f / sqrtf(row16_sum(ss) * (1.0f / E) + LN_EPS);
        if (sv_x && 4 * q + r < nrows) {
#pragma unroll
            for (int c = 0; c < 4; ++c) sv_x[(4 * q + r) * E + 16 * c + jl] = x[c][r];
            if (jl == 0) { sv_st[(4 * q + r) * 2] = mean; sv_st[(4 * q + r) * 2 + 1] = rstd; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c][r] = d[c] * rstd * lnw[16 * c + jl] + lnb[16 * c + jl];
    }
}

// Prepared weights of one layer (kernels.hpp: TF_LAYER_PREP_BYTES): per pair of hidden tiles a region of FFN_REGION
// u32x4, of which the FFN's split-fp16 fragments fill the first FFN_FRAGS * 64 of every pair laid out back to back
// (64 x 16 KB), the scale tables (TfFfnTables) sit behind them at FFN_TABLES_OFF, and the attention block's split-bf16
// fragments behind the whole FFN area.
constexpr int FFN_FRAGS = 16, FFN_PAIRS = FF / 32, FFN_REGION = 24 * 64;
constexpr int64_t FFN_TABLES_OFF = (int64_t)FFN_PAIRS * FFN_FRAGS * 64;       // u32x4 units
static_assert((int64_t)FFN_PAIRS * FFN_REGION * 16 == TF_FFN_PREP_BYTES, "FFN area of the prepared buffer");
// Scales of the split-fp16 FFN (device_utils.hpp: split2h).  Everything is a power of two:
//   h (LayerNorm1 output, |h| <= hb = max_j 8 |n1w_j| + |n1b_j|) goes in as 2^kh h;
//   row u of linear1 as 2^s1[u] W1[u][:]  -> its accumulator holds 2^(kh + s1[u]) (W1 h + b1)[u]  (b1s = scaled bias);
//   after the ReLU rs[u] = 2^(ku - kh - s1[u]) brings every unit to the common scale 2^ku, chosen from the bound
//   max_u |b1_u| + hb sum_k |W1[u][k]| so that 2^ku u stays inside the fp16 range;
//   row n of linear2 as 2^s2[n] W2[n][:]  -> the output accumulator of feature n holds 2^(ku + s2[n]) times the sum;
//   inv2[n] undoes it in the epilogue.
struct TfFfnTables {
    float b1s[FF], rs[FF], inv2[E];
    int s1[FF], s2[E];
    int kh, ku, pad_[2];
    // attention block (tf_prepare_kernel): rows of in_proj / out_proj are scaled by powers of two into the fp16 range;
    // 2^-shift per output column undoes it behind the accumulation
    float inv_in[G3], inv_out[E];
};
static_assert(FFN_TABLES_OFF * 16 + (int64_t)sizeof(TfFfnTables) <= TF_FFN_PREP_BYTES, "scale tables fit behind the fragments");
// ReLU as ONE integer maximum: a float with the sign bit set is a negative integer (fmaxf costs a second
// instruction that quiets signalling NaNs first)
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// three cross terms of the split-fp16 product, smallest first
#define TF_MM3(ACC, XH, XL, YH, YL) \
    ACC = mfma_h(XL, YH, ACC); ACC = mfma_h(XH, YL, ACC); ACC = mfma_h(XH, YH, ACC);

// ------------------------------------------------------------------------------ attention block
// One workgroup = NS whole samples (NS*S <= 64 token rows), 4 waves, one 16-row tile each.
// h -> in_proj (q|k|v) -> per (sample, head) softmax(q k^T / 4) v -> out_proj -> + h -> LayerNorm1.
// Only q|k|v lives in LDS (49 KB), so three workgroups share a CU and their phases (global fetch, matrix work,
// the latency-bound per-head softmax, LayerNorm) overlap; the A fragments of in_proj and the residual rows come
// straight from global memory.
constexpr int QLD = 196, AROWS = 64;

__device__ __forceinline__ void tf_attn_body(float (&qkv)[AROWS * QLD], const TfArgs& a, int layer, int bid) {
    const TfLayerW& W = a.layer[layer];
    const TfLayerBuf& Bf = a.buf[layer];
    const float* hin = Bf.hin;
    float* hout = Bf.hmid;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, jl = lane & 15, q = lane >> 4;
    const int S = a.S, NS = a.NS;
    const int64_t sample0 = (int64_t)bid * NS;
    const int ns = (int)min((int64_t)NS, a.B - sample0);
    const int rows = ns * S;
    const int64_t tok0 = sample0 * S;

    // ---- in_proj: [64 x 64] . W_in^T [64 x 192] on the two-way fp16 split (three cross terms, fp32 accumulate) from
    // prepared weight fragments whose rows carry a power-of-two scale each (tf_prepare_kernel); the h rows are scaled
    // per ROW TILE by the power of two that brings the tile's largest magnitude to the top of the fp16 range, found
    // by the wave that splits the tile (a wave-local maximum: no extra rendezvous) and passed on with the fragments.
    // Wave w OWNS COLUMN TILES w, w + 4, w + 8 for all four row tiles, so it fetches a quarter of the weight fragments
    // (12 of 48 KB: with a wave per row tile every wave pulled all of them through the CU's vector memory path, which
    // bounded the phase).  The 64 h rows are split once -- wave w its own row tile -- and exchanged as lane-major
    // fragments through the (not yet used) q|k|v region of LDS; every wave then keeps all four row tiles' A fragments
    // in registers.
    const u32x4* aprep = reinterpret_cast<const u32x4*>(a.ffn_prep[layer]) + (int64_t)FFN_PAIRS * FFN_REGION + lane;
    const TfFfnTables& atb = *reinterpret_cast<const TfFfnTables*>(reinterpret_cast<const u32x4*>(a.ffn_prep[layer]) + FFN_TABLES_OFF);
    {
        u32x4* xa = reinterpret_cast<u32x4*>(qkv);                   // [row tile 4][chunk 2][plane 2][lane 64]: 16 KB
        float* xk = qkv + 16 * 64 * 4;                               // behind them: 2^-k of the four row tiles
        {
            const bool ok = 16 * w + jl < rows;
            const float4* p = reinterpret_cast<const float4*>(hin + (tok0 + (ok ? 16 * w + jl : 0)) * E + 8 * q);
            const float4 z = make_float4(0, 0, 0, 0);
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ok ? p[8 * (i >> 1) + (i & 1)] : z;
            float m = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
            m = row16_max_dpp(m);
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const int kx = f16_row_shift(m);
            const float sx = pow2i(kx);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float4 v0 = v[2 * c], v1 = v[2 * c + 1];
                v0.x *= sx; v0.y *= sx; v0.z *= sx; v0.w *= sx; v1.x *= sx; v1.y *= sx; v1.z *= sx; v1.w *= sx;
                u32x4 h2, l2;
                split8h(v0, v1, h2, l2);
                u32x4* dst = xa + ((w * 2 + c) * 2) * 64 + lane;
                dst[0] = h2; dst[64] = l2;
            }
            if (lane == 0) xk[w] = pow2i(-kx);
        }
        __syncthreads();
        u32x4 xf[4][2][2];
        float ikx[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            ikx[rt] = xk[rt];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) xf[rt][c][pl] = xa[((rt * 2 + c) * 2 + pl) * 64 + lane];
        }
        __syncthreads();                                            // the fragments are in registers: q|k|v may be written
#pragma unroll 1
        for (int jt = 0; jt < 3; ++jt) {
            const int nt = w + 4 * jt;
            u32x4 wc[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) wc[f] = aprep[(nt * 4 + f) * 64];
            const float bias = W.inb[16 * nt + jl];
            const float un = atb.inv_in[16 * nt + jl];
            const float sc = nt < 4 ? 0.25f : 1.0f;                  // q / sqrt(head_dim)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                TF_MM3(acc0, xf[rt][0][0], xf[rt][0][1], wc[0], wc[1])
                TF_MM3(acc1, xf[rt][1][0], xf[rt][1][1], wc[2], wc[3])
                const float us = un * ikx[rt];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = ((acc0[r] + acc1[r]) * us + bias) * sc;
                    const int r0 = 16 * rt + 4 * q + r;
                    qkv[r0 * QLD + 16 * nt + jl] = v0;
                    if (Bf.qkv && r0 < rows) Bf.qkv[(tok0 + r0) * G3 + 16 * nt + jl] = v0;
                }
            }
        }
    }
    // the residual rows start their trip before the attention phase
    float res[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + 4 * q + r;
            res[c][r] = row < rows ? hin[(tok0 + row) * E + 16 * c + jl] : 0.0f;
        }
    __syncthreads();
    // ---- attention: one wave per (sample, head) ------------------------------------------------
    const int nt16 = (S + 15) >> 4;                                 // key / query tiles per sample (1 or 2)
    for (int p = w; p < ns * 4; p += 4) {
        const int si = p >> 2, hh = p & 3, rb = si * S;
        for (int qt = 0; qt < nt16; ++qt) {
            const int qrow = min(rb + 16 * qt + jl, AROWS - 1);
            const float4 qv = *reinterpret_cast<const float4*>(&qkv[qrow * QLD + 16 * hh + 4 * q]);
            f32x4 st[2];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                st[kt] = (f32x4){0, 0, 0, 0};
                if (kt < nt16) {
                    const int krow = min(rb + 16 * kt + jl, AROWS - 1);
                    const float4 kv = *reinterpret_cast<const float4*>(&qkv[krow * QLD + E + 16 * hh + 4 * q]);
                    st[kt] = mfma16(kv.x, qv.x, st[kt]); st[kt] = mfma16(kv.y, qv.y, st[kt]);
                    st[kt] = mfma16(kv.z, qv.z, st[kt]); st[kt] = mfma16(kv.w, qv.w, st[kt]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                    // row = key 16kt + 4q + r, column = query jl
                        if (16 * kt + 4 * q + r >= S) st[kt][r] = -INFINITY;
                        mx = fmaxf(mx, st[kt][r]);
                    }
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = kt < nt16 ? fast_exp2((st[kt][r] - mx) * LOG2E) : 0.0f;   // v_exp_f32, ~1 ulp
                    st[kt][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = fast_rcp(sum);
            // ctx^T[d][query] = sum_key V[key][d] P[key][query]: A lane (d = jl, q) at step r holds V[16kt+4q+r][d]
            f32x4 ct = {0, 0, 0, 0};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                if (kt < nt16) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int vrow = min(rb + 16 * kt + 4 * q + r, AROWS - 1);
                        ct = mfma16(qkv[vrow * QLD + 2 * E + 16 * hh + jl], st[kt][r] * inv, ct);
                    }
                }
            }
            // D: ctx^T[d = 4q + r][query = jl] -> the (dead) q slot of this query row
            if (16 * qt + jl < S)
                *reinterpret_cast<float4*>(&qkv[(rb + 16 * qt + jl) * QLD + 16 * hh + 4 * q]) =
                    make_float4(ct[0], ct[1], ct[2], ct[3]);
        }
    }
    __syncthreads();
    // ---- out_proj + residual + LayerNorm1 --------------------------------------------------------
    {
        const int rbase = 16 * w;
        // A operand: the context row rbase + jl, 8 consecutive columns per K chunk, scaled by the power of two of this
        // wave's largest context magnitude (the wave multiplies its own row tile only) and split here
        float4 cv[4];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4* p = reinterpret_cast<const float4*>(&qkv[(rbase + jl) * QLD + 32 * c + 8 * q]);
            cv[2 * c] = p[0]; cv[2 * c + 1] = p[1];
            if (Bf.ctx && rbase + jl < rows) {
                float4* dst = reinterpret_cast<float4*>(Bf.ctx + (tok0 + rbase + jl) * E + 32 * c + 8 * q);
                dst[0] = p[0]; dst[1] = p[1];
            }
        }
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) m = fmaxf(m, fmaxf(fmaxf(fabsf(cv[i].x), fabsf(cv[i].y)), fmaxf(fabsf(cv[i].z), fabsf(cv[i].w))));
        m = row16_max_dpp(m);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const int kc = f16_row_shift(m);
        const float sxc = pow2i(kc), ikc = pow2i(-kc);
        u32x4 xh[2], xl[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float4 v0 = cv[2 * c], v1 = cv[2 * c + 1];
            v0.x *= sxc; v0.y *= sxc; v0.z *= sxc; v0.w *= sxc; v1.x *= sxc; v1.y *= sxc; v1.z *= sxc; v1.w *= sxc;
            split8h(v0, v1, xh[c], xl[c]);
        }
        const u32x4* oprep = aprep + 48 * 64;
        u32x4 wn[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) wn[f] = oprep[f * 64];
        float x[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u32x4 wc[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) wc[f] = wn[f];
            if (c + 1 < 4) {
#pragma unroll
                for (int f = 0; f < 4; ++f) wn[f] = oprep[((c + 1) * 4 + f) * 64];
            }
            f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            TF_MM3(acc0, xh[0], xl[0], wc[0], wc[1])
            TF_MM3(acc1, xh[1], xl[1], wc[2], wc[3])
            const float bias = W.ob[16 * c + jl];
            const float us = atb.inv_out[16 * c + jl] * ikc;
#pragma unroll
            for (int r = 0; r < 4; ++r) x[c][r] = ((acc0[r] + acc1[r]) * us + bias) + res[c][r];
        }
        ln_rows(x, W.n1w, W.n1b, jl, q, Bf.x1 ? Bf.x1 + (tok0 + rbase) * E : nullptr,
                Bf.x1 ? Bf.st1 + (tok0 + rbase) * 2 : nullptr, rows - rbase);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rbase + 4 * q + r;
            if (row < rows) {
#pragma unroll
                for (int c = 0; c < 4; ++c) hout[(tok0 + row) * E + 16 * c + jl] = x[c][r];
            }
        }
    }
}

__global__ __launch_bounds__(256, 3) void tf_attn_kernel(TfArgs a, int layer) {
    __shared__ __attribute__((aligned(16))) float qkv[AROWS * QLD];
    tf_attn_body(qkv, a, layer, (int)blockIdx.x);
}
// the attention blocks of two encoders of one network as one grid (as tf_ffn_pair_kernel below)
__global__ __launch_bounds__(256, 3) void tf_attn_pair_kernel(TfArgs a, TfArgs b, int na, int layer) {
    __shared__ __attribute__((aligned(16))) float qkv[AROWS * QLD];
    if ((int)blockIdx.x < na) tf_attn_body(qkv, a, layer, (int)blockIdx.x);
    else tf_attn_body(qkv, b, layer, (int)blockIdx.x - na);
}

// ------------------------------------------------------------------------------------ FFN block
// hout = LayerNorm2(hin + W2 relu(W1 hin + b1) + b2).  One workgroup = 256 tokens, one wave = NT tiles of 16
// tokens (NT = 2: 8 waves, two per SIMD, so one wave's split / ReLU VALU work overlaps its neighbour's matrix
// work, and every weight fragment a wave reads from LDS serves two token tiles: at NT = 1 the 16 waves' fragment
// reads, 256 KB per pair and CU, took more LDS cycles than the matrix instructions took issue cycles).  The 2048 hidden units are walked 32 at a time ("pair" of 16-unit tiles); the hidden tile never
// leaves registers: the D layout of the first product (u^T = W1 . h^T: rows = hidden units, columns =
// tokens) is exactly the A operand the second product (out = u . W2^T) wants, with the k slots of the
// 16x16x32 instruction assigned to the units a lane already holds (slot e of lane group q: unit 4q+e of the
// first tile for e < 4, unit 16+4q+e-4 of the second) -- the prepared W2 fragments use the same order.
// Both products run on the fp16 matrix instruction over the two-way split of their operands (three cross
// terms, fp32 accumulate; power-of-two scales from TfFfnTables: DESIGN.md section 3): h is split once per wave,
// the weights once per parameter change (tf_prepare_kernel), u after the ReLU.  The pair's 16 weight fragments
// (16 KB) are fetched once per workgroup into double-buffered LDS and shared by its waves.
#ifndef MCTSQ_FFN_SP
#define MCTSQ_FFN_SP 1      // pairs of hidden tiles staged per barrier; measured per 16384-leaf forward: 2.18 ms (1), 2.19 (2), 2.52 (4):
                            // the barrier per pair is not what bounds the kernel
#endif
constexpr int FFN_SP = MCTSQ_FFN_SP;
static_assert(FFN_PAIRS % FFN_SP == 0, "staged groups cover the hidden units");
#ifndef MCTSQ_FFN_NT
#define MCTSQ_FFN_NT 4      // token tiles of 16 per wave; measured per 16384-leaf forward with 8 waves of NT tiles per CU-filling
                            // workgroup: 2.55 ms (1), 2.41 (2), 2.98 (4: one wave per SIMD); 3.07 for (2) squeezed into 128
                            // registers so that two workgroups share a CU.  With FOUR waves per workgroup and two workgroups
                            // per CU (MCTSQ_FFN_WAVES below; 242 registers, two waves per SIMD): 2.12 (2) -> 1.98 (4): every
                            // fragment a wave reads from LDS serves four token tiles, and the fragment reads - which the
                            // barrier per pair keeps from overlapping the matrix instructions - drop from 0.75 to 0.42 of
                            // the matrix time
#endif

#ifndef MCTSQ_FFN_TTMAJOR
#define MCTSQ_FFN_TTMAJOR 1  // pair loop body token tile by token tile (see there); 0: unit tile by unit tile
#endif
#ifndef MCTSQ_FFN_WAVES
#define MCTSQ_FFN_WAVES 4   // waves per FFN workgroup (x 16 NT tokens each).  4: two workgroups share a CU (the two waves of a
                            // SIMD belong to different barrier domains); 8 at NT = 2 measured 2.12 ms against 2.09 for 4
#endif
constexpr int FFN_WAVES = MCTSQ_FFN_WAVES, FFN_TOK = FFN_WAVES * 16 * MCTSQ_FFN_NT;
static_assert(TOK_PER_WG % FFN_TOK == 0, "token padding covers whole FFN workgroups");
#ifndef MCTSQ_FFN_PIPE
#define MCTSQ_FFN_PIPE 0    // 1: software-pipelined pair loop (below); 0: the plain loop (one pair start to finish per iteration).
                            // Measured per 16384-leaf forward: 2.64 ms (1) against 2.14 (0): the compiler keeps V(g) ahead of
                            // A(g + 1) instead of interleaving them (inline-asm conversions are invisible to its scheduling
                            // groups) and the accumulator hand-over costs 16 moves per pair; kept for reference
#endif
constexpr int FFN_LDS_BUFS = MCTSQ_FFN_PIPE ? 3 : 2;

#if MCTSQ_FFN_PIPE
// Software-pipelined form.  Per pair of hidden tiles a wave has three jobs in a dependent chain: A = the first product
// (24 matrix instructions), V = ReLU / rescale / split of its result (vector instructions), B = the second product (24
// matrix instructions).  Run start to finish per pair, the two waves of a SIMD - released together by the workgroup
// barrier that ends every pair - do their A at the same time, their V at the same time and wait for their LDS
// fragments at the same time: the matrix pipe idles through every V and every fragment fetch (MFMA busy 0.54, measured).
// Here iteration g runs V(g) side by side with A(g + 1), which depends on nothing of pair g, so a wave always has matrix
// instructions to issue while its vector work and its fragment reads are in flight: V(g) | A(g + 1), then B(g).  The
// fragments of pairs g and g + 1 are both resident (three LDS buffers; pair g + 2 is staged during the iteration).
template <int NT>
__device__ __forceinline__ void tf_ffn_body(u32x4 (&wl)[FFN_LDS_BUFS][FFN_SP * FFN_FRAGS * 64], const TfArgs& a, int layer, int bid) {
    static_assert(FFN_SP == 1, "the pipelined loop stages one pair per barrier");
    const TfLayerW& W = a.layer[layer];
    const TfLayerBuf& Bf = a.buf[layer];
    const float* hin = Bf.hmid;
    float* hout = Bf.hout;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, jl = lane & 15, q = lane >> 4;
    constexpr int GFR = FFN_FRAGS * 64;                          // b128 words of one pair's fragments
    constexpr int NW = FFN_WAVES, NTHR = 64 * NW, NSTG = (GFR + NTHR - 1) / NTHR;
    const int64_t tok0 = (int64_t)bid * FFN_TOK + w * (16 * NT);
    const u32x4* prep = reinterpret_cast<const u32x4*>(a.ffn_prep[layer]);
    const TfFfnTables& tb = *reinterpret_cast<const TfFfnTables*>(prep + FFN_TABLES_OFF);
    u32x4 hh[NT][2], hl[NT][2];
    {
        const float sh = pow2i(tb.kh);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float4* p = reinterpret_cast<const float4*>(hin + (tok0 + 16 * tt + jl) * E + 32 * c + 8 * q);
                float4 x0 = p[0], x1 = p[1];
                x0.x *= sh; x0.y *= sh; x0.z *= sh; x0.w *= sh; x1.x *= sh; x1.y *= sh; x1.z *= sh; x1.w *= sh;
                split8h(x0, x1, hh[tt][c], hl[tt][c]);
            }
    }
    f32x4 acc[NT][4];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[tt][c] = (f32x4){0, 0, 0, 0};
    // pairs 0 and 1 -> LDS buffers 0 and 1
    u32x4 stg[NSTG];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int i = 0; i < NSTG; ++i)
            if (tid + NTHR * i < GFR) stg[i] = prep[(int64_t)b * GFR + tid + NTHR * i];
#pragma unroll
        for (int i = 0; i < NSTG; ++i)
            if (tid + NTHR * i < GFR) wl[b][tid + NTHR * i] = stg[i];
    }
    const float* b1p = tb.b1s + 4 * q;
    const float* rsp = tb.rs + 4 * q;
    // first product of one pair: u^T = W1 . h^T on top of the scaled bias (fragments 0..7 of the pair's buffer)
    auto first_product = [&](const u32x4* wf, const float4& b10, const float4& b11, f32x4 (&ua)[NT], f32x4 (&ub)[NT]) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            ua[tt] = (f32x4){b10.x, b10.y, b10.z, b10.w};
            ub[tt] = (f32x4){b11.x, b11.y, b11.z, b11.w};
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u32x4 ah = wf[((0 * 2 + c) * 2 + 0) * 64], al = wf[((0 * 2 + c) * 2 + 1) * 64];
            const u32x4 bh = wf[((1 * 2 + c) * 2 + 0) * 64], bl = wf[((1 * 2 + c) * 2 + 1) * 64];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                TF_MM3(ua[tt], ah, al, hh[tt][c], hl[tt][c])
                TF_MM3(ub[tt], bh, bl, hh[tt][c], hl[tt][c])
            }
        }
    };
    float4 b1n0 = *reinterpret_cast<const float4*>(b1p), b1n1 = *reinterpret_cast<const float4*>(b1p + 16);
    float4 rsc0 = *reinterpret_cast<const float4*>(rsp), rsc1 = *reinterpret_cast<const float4*>(rsp + 16);
    __syncthreads();
    f32x4 ua[NT], ub[NT];
    first_product(&wl[0][lane], b1n0, b1n1, ua, ub);             // A(0)
    b1n0 = *reinterpret_cast<const float4*>(b1p + 32); b1n1 = *reinterpret_cast<const float4*>(b1p + 32 + 16);
    int cur = 0;                                                 // LDS buffer of pair g (g % 3 without the division)
#pragma unroll 1
    for (int g = 0; g < FFN_PAIRS; ++g) {
        const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
        const float4 rs0 = rsc0, rs1 = rsc1, b10 = b1n0, b11 = b1n1;
        if (g + 2 < FFN_PAIRS) {                                 // fragments of pair g + 2 start their trip
#pragma unroll
            for (int i = 0; i < NSTG; ++i)
                if (tid + NTHR * i < GFR) stg[i] = prep[(int64_t)(g + 2) * GFR + tid + NTHR * i];
        }
        if (g + 1 < FFN_PAIRS) {                                 // scales of pair g + 1, bias of pair g + 2
            rsc0 = *reinterpret_cast<const float4*>(rsp + (g + 1) * 32); rsc1 = *reinterpret_cast<const float4*>(rsp + (g + 1) * 32 + 16);
        }
        if (g + 2 < FFN_PAIRS) {
            b1n0 = *reinterpret_cast<const float4*>(b1p + (g + 2) * 32); b1n1 = *reinterpret_cast<const float4*>(b1p + (g + 2) * 32 + 16);
        }
        // W2 fragments of pair g: requested first, used last
        const u32x4* wf = &wl[cur][lane];
        u32x4 yh[4], yl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { yh[c] = wf[(8 + c * 2 + 0) * 64]; yl[c] = wf[(8 + c * 2 + 1) * 64]; }
        // V(g): ReLU, common scale 2^ku, split -- vector work, in the shadow of A(g + 1)'s matrix instructions
        u32x4 uh[NT], ul[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) relu_scale_split8h(ua[tt], ub[tt], rs0, rs1, uh[tt], ul[tt]);
        // A(g + 1)
        f32x4 va[NT], vb[NT];
        if (g + 1 < FFN_PAIRS) first_product(&wl[nxt][lane], b10, b11, va, vb);
        // B(g): out += u . W2^T
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) { TF_MM3(acc[tt][c], uh[tt], ul[tt], yh[c], yl[c]) }
        if (g + 2 < FFN_PAIRS) {
#pragma unroll
            for (int i = 0; i < NSTG; ++i)
                if (tid + NTHR * i < GFR) wl[nn][tid + NTHR * i] = stg[i];
        }
        __syncthreads();
        if (g + 1 < FFN_PAIRS) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) { ua[tt] = va[tt]; ub[tt] = vb[tt]; }
        }
        cur = nxt;
    }
    if (tok0 >= a.T) return;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        float x[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float bias = W.l2b[16 * c + jl], inv = tb.inv2[16 * c + jl];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[c][r] = (acc[tt][c][r] * inv + bias) + hin[(tok0 + 16 * tt + 4 * q + r) * E + 16 * c + jl];
        }
        {
            const int64_t t0 = tok0 + 16 * tt;
            const int64_t left = a.T - t0;
            ln_rows(x, W.n2w, W.n2b, jl, q, Bf.x2 ? Bf.x2 + t0 * E : nullptr, Bf.x2 ? Bf.st2 + t0 * 2 : nullptr,
                    (int)(left < 16 ? (left < 0 ? 0 : left) : 16));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t tok = tok0 + 16 * tt + 4 * q + r;
            if (tok < a.T) {
#pragma unroll
                for (int c = 0; c < 4; ++c) hout[tok * E + 16 * c + jl] = x[c][r];
            }
        }
    }
}
#else
template <int NT>
__device__ __forceinline__ void tf_ffn_body(u32x4 (&wl)[FFN_LDS_BUFS][FFN_SP * FFN_FRAGS * 64], const TfArgs& a, int layer, int bid) {
    const TfLayerW& W = a.layer[layer];
    const TfLayerBuf& Bf = a.buf[layer];
    const float* hin = Bf.hmid;
    float* hout = Bf.hout;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, jl = lane & 15, q = lane >> 4;
    constexpr int SP = FFN_SP, GFR = SP * FFN_FRAGS * 64;       // pairs per staged group (one barrier per group), its b128 count
    constexpr int NW = FFN_WAVES, NTHR = 64 * NW, NSTG = (GFR + NTHR - 1) / NTHR;   // waves, threads, staged b128 per thread
    const int64_t tok0 = (int64_t)bid * FFN_TOK + w * (16 * NT);               // < Tp
    const u32x4* prep = reinterpret_cast<const u32x4*>(a.ffn_prep[layer]);
    const TfFfnTables& tb = *reinterpret_cast<const TfFfnTables*>(prep + FFN_TABLES_OFF);
    // B operand of the first product: 2^kh h^T, lane (token jl, k group q) holds h[token][32c + 8q .. +7], split
    u32x4 hh[NT][2], hl[NT][2];
    {
        const float sh = pow2i(tb.kh);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float4* p = reinterpret_cast<const float4*>(hin + (tok0 + 16 * tt + jl) * E + 32 * c + 8 * q);
                float4 x0 = p[0], x1 = p[1];
                x0.x *= sh; x0.y *= sh; x0.z *= sh; x0.w *= sh; x1.x *= sh; x1.y *= sh; x1.z *= sh; x1.w *= sh;
                split8h(x0, x1, hh[tt][c], hl[tt][c]);
            }
    }
    f32x4 acc[NT][4];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[tt][c] = (f32x4){0, 0, 0, 0};

    u32x4 stg[NSTG];
#pragma unroll
    for (int i = 0; i < NSTG; ++i)
        if (tid + NTHR * i < GFR) stg[i] = prep[tid + NTHR * i];
#pragma unroll
    for (int i = 0; i < NSTG; ++i)
        if (tid + NTHR * i < GFR) wl[0][tid + NTHR * i] = stg[i];
    // scaled bias and post-ReLU rescale of hidden units 4q + r (first tile) and 16 + 4q + r (second tile) of a pair
    const float* b1p = tb.b1s + 4 * q;
    const float* rsp = tb.rs + 4 * q;
    float4 b1n[SP][2], rsn[SP][2];
#pragma unroll
    for (int sp = 0; sp < SP; ++sp) {
        b1n[sp][0] = *reinterpret_cast<const float4*>(b1p + sp * 32); b1n[sp][1] = *reinterpret_cast<const float4*>(b1p + sp * 32 + 16);
        rsn[sp][0] = *reinterpret_cast<const float4*>(rsp + sp * 32); rsn[sp][1] = *reinterpret_cast<const float4*>(rsp + sp * 32 + 16);
    }
    __syncthreads();
#pragma unroll 2
    for (int g = 0; g < FFN_PAIRS / SP; ++g) {
        const int buf = g & 1;
        float4 b1c[SP][2], rsc[SP][2];
#pragma unroll
        for (int sp = 0; sp < SP; ++sp) { b1c[sp][0] = b1n[sp][0]; b1c[sp][1] = b1n[sp][1]; rsc[sp][0] = rsn[sp][0]; rsc[sp][1] = rsn[sp][1]; }
        if (g + 1 < FFN_PAIRS / SP) {                               // next group's fragments, biases and scales in flight
#pragma unroll
            for (int i = 0; i < NSTG; ++i)
                if (tid + NTHR * i < GFR) stg[i] = prep[(int64_t)(g + 1) * GFR + tid + NTHR * i];
#pragma unroll
            for (int sp = 0; sp < SP; ++sp) {
                const int pn = (g + 1) * SP + sp;
                b1n[sp][0] = *reinterpret_cast<const float4*>(b1p + pn * 32); b1n[sp][1] = *reinterpret_cast<const float4*>(b1p + pn * 32 + 16);
                rsn[sp][0] = *reinterpret_cast<const float4*>(rsp + pn * 32); rsn[sp][1] = *reinterpret_cast<const float4*>(rsp + pn * 32 + 16);
            }
        }
#pragma unroll
      for (int sp = 0; sp < SP; ++sp) {
        const float4 b10 = b1c[sp][0], b11 = b1c[sp][1], rs0 = rsc[sp][0], rs1 = rsc[sp][1];
        const u32x4* wf = &wl[buf][sp * (FFN_FRAGS * 64) + lane];
        // every weight fragment is read from LDS once per wave and pair and serves all NT token tiles
#if MCTSQ_FFN_TTMAJOR
        // Token tile by token tile: tile tt's first product, then its ReLU / split - vector work that depends only on
        // tile tt's accumulators, so it can issue in the shadow of tile tt + 1's matrix instructions - and the second
        // product's fragments are requested before the first product starts.  (Unit tile by unit tile over all token
        // tiles, all of a pair's vector work sat in one burst between the two products with the matrix pipe idle.)
        u32x4 w1h[2][2], w1l[2][2], yh[4], yl[4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < 2; ++c) { w1h[m][c] = wf[((m * 2 + c) * 2 + 0) * 64]; w1l[m][c] = wf[((m * 2 + c) * 2 + 1) * 64]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) { yh[c] = wf[(8 + c * 2 + 0) * 64]; yl[c] = wf[(8 + c * 2 + 1) * 64]; }
        u32x4 uh[NT], ul[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            f32x4 ua = {b10.x, b10.y, b10.z, b10.w}, ub = {b11.x, b11.y, b11.z, b11.w};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                TF_MM3(ua, w1h[0][c], w1l[0][c], hh[tt][c], hl[tt][c])
                TF_MM3(ub, w1h[1][c], w1l[1][c], hh[tt][c], hl[tt][c])
            }
            relu_scale_split8h(ua, ub, rs0, rs1, uh[tt], ul[tt]);
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int c = 0; c < 4; ++c) { TF_MM3(acc[tt][c], uh[tt], ul[tt], yh[c], yl[c]) }
#else
        // ---- u^T = W1 . h^T for the pair's two unit tiles, on top of the (scaled) bias of hidden unit 4q + r ----
        f32x4 ua[NT], ub[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            ua[tt] = (f32x4){b10.x, b10.y, b10.z, b10.w};
            ub[tt] = (f32x4){b11.x, b11.y, b11.z, b11.w};
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u32x4 ah = wf[((0 * 2 + c) * 2 + 0) * 64], al = wf[((0 * 2 + c) * 2 + 1) * 64];
            const u32x4 bh = wf[((1 * 2 + c) * 2 + 0) * 64], bl = wf[((1 * 2 + c) * 2 + 1) * 64];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                TF_MM3(ua[tt], ah, al, hh[tt][c], hl[tt][c])
                TF_MM3(ub[tt], bh, bl, hh[tt][c], hl[tt][c])
            }
        }
        // ---- ReLU, common scale 2^ku, split: the lane's 8 values are k slots 0..7 of the second product ----
        u32x4 uh[NT], ul[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) relu_scale_split8h(ua[tt], ub[tt], rs0, rs1, uh[tt], ul[tt]);
        // ---- out += u . W2^T ----
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32x4 yh = wf[(8 + c * 2 + 0) * 64], yl = wf[(8 + c * 2 + 1) * 64];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) { TF_MM3(acc[tt][c], uh[tt], ul[tt], yh, yl) }
        }
#endif
      }
        if (g + 1 < FFN_PAIRS / SP) {
#pragma unroll
            for (int i = 0; i < NSTG; ++i)
                if (tid + NTHR * i < GFR) wl[buf ^ 1][tid + NTHR * i] = stg[i];
        }
        __syncthreads();
    }
    if (tok0 >= a.T) return;
    // ---- unscale, + b2 + residual, LayerNorm2 ----------------------------------------------------
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        float x[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float bias = W.l2b[16 * c + jl], inv = tb.inv2[16 * c + jl];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[c][r] = (acc[tt][c][r] * inv + bias) + hin[(tok0 + 16 * tt + 4 * q + r) * E + 16 * c + jl];
        }
        {
            const int64_t t0 = tok0 + 16 * tt;
            const int64_t left = a.T - t0;
            ln_rows(x, W.n2w, W.n2b, jl, q, Bf.x2 ? Bf.x2 + t0 * E : nullptr, Bf.x2 ? Bf.st2 + t0 * 2 : nullptr,
                    (int)(left < 16 ? (left < 0 ? 0 : left) : 16));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t tok = tok0 + 16 * tt + 4 * q + r;
            if (tok < a.T) {
#pragma unroll
                for (int c = 0; c < 4; ++c) hout[tok * E + 16 * c + jl] = x[c][r];
            }
        }
    }
}

#endif

template <int NT>
__global__ __launch_bounds__(64 * FFN_WAVES, 2) void tf_ffn_kernel(TfArgs a, int layer) {
    __shared__ __attribute__((aligned(16))) u32x4 wl[FFN_LDS_BUFS][FFN_SP * FFN_FRAGS * 64];   // 16 KB per staged pair
    tf_ffn_body<NT>(wl, a, layer, (int)blockIdx.x);
}
// The FFN blocks of TWO encoders (price and process window of one network) as one grid: launched one after the
// other, each grid ends with a round in which most CUs idle (1600 and 640 workgroups on 256 CUs: 7 + 3 rounds);
// together they take 9.
template <int NT>
__global__ __launch_bounds__(64 * FFN_WAVES, 2) void tf_ffn_pair_kernel(TfArgs a, TfArgs b, int na, int layer) {
    __shared__ __attribute__((aligned(16))) u32x4 wl[FFN_LDS_BUFS][FFN_SP * FFN_FRAGS * 64];
    if ((int)blockIdx.x < na) tf_ffn_body<NT>(wl, a, layer, (int)blockIdx.x);
    else tf_ffn_body<NT>(wl, b, layer, (int)blockIdx.x - na);
}

// ----------------------------------------------------------------- mean over time + LayerNorm
__device__ __forceinline__ void tf_pool_body(const TfArgs& a, int bid) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)bid * 4 + (threadIdx.x >> 6);
    if (b >= a.Bp) return;
    float v = 0.0f;
    if (b < a.B) {
        float s = 0.0f;
        for (int t = 0; t < a.S; ++t) s += a.buf[1].hout[(b * a.S + t) * E + lane];
        const float m = s / (float)a.S;
        const float mean = wave_sum(m) * (1.0f / E);
        const float d = m - mean;
        const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.0f / E) + LN_EPS);
        v = d * rstd * a.lnw[lane] + a.lnb[lane];
        if (a.sv_mean) a.sv_mean[b * E + lane] = m;
    }
    a.feats[b * FEAT + a.out_col + lane] = v;
}
__global__ __launch_bounds__(256) void tf_pool_kernel(TfArgs a) { tf_pool_body(a, (int)blockIdx.x); }
__global__ __launch_bounds__(256) void tf_pool_pair_kernel(TfArgs a, TfArgs b, int na) {
    if ((int)blockIdx.x < na) tf_pool_body(a, (int)blockIdx.x);
    else tf_pool_body(b, (int)blockIdx.x - na);
}

// Scales of the split-fp16 FFN (TfFfnTables), one workgroup of 1024 threads per layer; runs ahead of tf_prepare_kernel.
__device__ __forceinline__ float block_max_1024(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();                                          // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    return m;
}
__device__ __forceinline__ int clamp_exp(int e) { return e < -120 ? -120 : (e > 120 ? 120 : e); }
__global__ __launch_bounds__(1024) void tf_scales_kernel(const float* __restrict__ l1w, const float* __restrict__ l1b,
                                                         const float* __restrict__ l2w, const float* __restrict__ n1w,
                                                         const float* __restrict__ n1b, TfFfnTables* __restrict__ tb) {
    __shared__ float red[16];
    const int t = threadIdx.x;
    // |LayerNorm1 output| <= sqrt(E - 1) |n1w_j| + |n1b_j| < 8 |n1w_j| + |n1b_j|
    const float hb = block_max_1024(t < E ? 8.0f * fabsf(n1w[t]) + fabsf(n1b[t]) : 0.0f, red);
    const int kh = f16_row_shift(hb);
    // rows of linear1 [2048][64]: 16 lanes per row (one 16-byte load each, a wave reads four whole rows per instruction)
    float ubl = 0.0f;
    const int part = t & 15;
#pragma unroll 8
    for (int it = 0; it < FF / 64; ++it) {
        const int u = it * 64 + (t >> 4);
        const float4 v = reinterpret_cast<const float4*>(l1w + (int64_t)u * E)[part];
        const float amax = row16_max_dpp(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        const float asum = row16_sum((fabsf(v.x) + fabsf(v.y)) + (fabsf(v.z) + fabsf(v.w)));
        if (part == 0) tb->s1[u] = f16_row_shift(amax);
        ubl = fmaxf(ubl, fabsf(l1b[u]) + asum * hb);
    }
    const float ub = block_max_1024(ubl, red);               // its barriers also publish s1[] to the workgroup
    const int ku = f16_row_shift(ub);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = t + 1024 * i;
        const int s1 = tb->s1[u];
        tb->b1s[u] = l1b[u] * pow2i(clamp_exp(kh + s1));
        tb->rs[u] = pow2i(clamp_exp(ku - kh - s1));
    }
    // rows of linear2 [64][2048]: 16 lanes per row, 32 loads of 16 bytes each with four independent maxima
    {
        const int n = t >> 4;
        const float4* pr = reinterpret_cast<const float4*>(l2w + (int64_t)n * FF) + part;
        float m[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < FF / 64; ++k) {
            const float4 v = pr[k * 16];
            m[k & 3] = fmaxf(m[k & 3], fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        const float amax = row16_max_dpp(fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])));
        if (part == 0) {
            const int s2 = f16_row_shift(amax);
            tb->s2[n] = s2;
            tb->inv2[n] = pow2i(clamp_exp(-(ku + s2)));
        }
    }
    if (t == 0) { tb->kh = kh; tb->ku = ku; }
}

// FFN weights -> split-fp16 fragments: one thread per (pair, fragment group, lane); fragment order as
// read by tf_ffn_kernel: W1 [unit tile m 2][K chunk c 2][plane 2], then W2 [out tile n 4][plane 2]; rows scaled by
// the powers of two tf_scales_kernel chose.  The attention block's projections: two fp16 planes per row as well, the row's
// power of two chosen here (inverse in TfFfnTables::inv_in / inv_out).
__global__ __launch_bounds__(256) void tf_prepare_kernel(const float* __restrict__ l1w, const float* __restrict__ l2w,
                                                        const float* __restrict__ inw, const float* __restrict__ ow,
                                                        u32x4* __restrict__ out) {
    const int gid = blockIdx.x * 256 + threadIdx.x;           // FFN_PAIRS * 8 * 64 threads, then 32 * 64 for the attention block
    const int lane = gid & 63, grp = (gid >> 6) & 7, pr = gid >> 9;
    const int jl = lane & 15, q = lane >> 4;
    if (gid >= FFN_PAIRS * 8 * 64) {                          // in_proj / out_proj: (column tile, K chunk) per 64 lanes
        const int g2 = (gid - FFN_PAIRS * 8 * 64) >> 6;       // 0..23 in_proj, 24..31 out_proj
        const float* w = g2 < 24 ? inw : ow;
        const int g = g2 < 24 ? g2 : g2 - 24, tile = g >> 1, c = g & 1, row = 16 * tile + jl;
        // the row's power of two: every thread of the row reads all 64 weights (256 B, 2048 threads in all)
        const float4* rw = reinterpret_cast<const float4*>(w + (int64_t)row * E);
        float amax = 0.0f;
#pragma unroll
        for (int i = 0; i < E / 4; ++i) {
            const float4 v = rw[i];
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        const int sh = f16_row_shift(amax);
        const float sc = pow2i(sh);
        float4 v0 = rw[8 * c + 2 * q], v1 = rw[8 * c + 2 * q + 1];
        v0.x *= sc; v0.y *= sc; v0.z *= sc; v0.w *= sc; v1.x *= sc; v1.y *= sc; v1.z *= sc; v1.w *= sc;
        u32x4 hi, lo;
        split8h(v0, v1, hi, lo);
        u32x4* o = out + (int64_t)FFN_PAIRS * FFN_REGION + lane;
        o[(g2 * 2 + 0) * 64] = hi; o[(g2 * 2 + 1) * 64] = lo;
        if (c == 0 && q == 0) {
            TfFfnTables* tbw = reinterpret_cast<TfFfnTables*>(out + FFN_TABLES_OFF);
            if (g2 < 24) tbw->inv_in[row] = pow2i(-sh); else tbw->inv_out[row] = pow2i(-sh);
        }
        return;
    }
    const TfFfnTables& tb = *reinterpret_cast<const TfFfnTables*>(out + FFN_TABLES_OFF);
    float4 v0, v1;
    float sc;
    int f0;
    if (grp < 4) {                                            // W1: unit tile m, K chunk c
        const int m = grp >> 1, c = grp & 1, u = 32 * pr + 16 * m + jl;
        const float4* p = reinterpret_cast<const float4*>(l1w + (int64_t)u * E + 32 * c + 8 * q);
        v0 = p[0]; v1 = p[1];
        sc = pow2i(tb.s1[u]);
        f0 = (m * 2 + c) * 2;
    } else {                                                  // W2: output tile n; k slot e -> unit 4q+e | 16+4q+e-4
        const int n = grp - 4;
        const float* p = l2w + (int64_t)(16 * n + jl) * FF + 32 * pr + 4 * q;
        v0 = *reinterpret_cast<const float4*>(p); v1 = *reinterpret_cast<const float4*>(p + 16);
        sc = pow2i(tb.s2[16 * n + jl]);
        f0 = 8 + n * 2;
    }
    v0.x *= sc; v0.y *= sc; v0.z *= sc; v0.w *= sc; v1.x *= sc; v1.y *= sc; v1.z *= sc; v1.w *= sc;
    u32x4 hi, lo;
    split8h(v0, v1, hi, lo);
    u32x4* o = out + (int64_t)pr * (FFN_FRAGS * 64) + lane;
    o[(f0 + 0) * 64] = hi; o[(f0 + 1) * 64] = lo;
}

int launch_tf_prepare(const TfLayerW& w, void* prep, hipStream_t st) {
    u32x4* out = reinterpret_cast<u32x4*>(prep);
    hipLaunchKernelGGL(tf_scales_kernel, dim3(1), dim3(1024), 0, st, w.l1w, w.l1b, w.l2w, w.n1w, w.n1b,
                       reinterpret_cast<TfFfnTables*>(out + FFN_TABLES_OFF));
    hipLaunchKernelGGL(tf_prepare_kernel, dim3((FFN_PAIRS * 8 + 32) * 64 / 256), dim3(256), 0, st, w.l1w, w.l2w, w.inw, w.ow, out);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int64_t tf_padded_tokens(int64_t B, int S) { return round_up(B * S, TOK_PER_WG); }

int launch_tf_fwd(const TfArgs& a, hipStream_t st) {
    if (a.B == 0) return 0;
    MCTSQ_CHECK(a.S >= 1 && a.S <= 32, "transformer encoder: seq_len %d outside 1..32", a.S);
    MCTSQ_CHECK(a.Tp % TOK_PER_WG == 0 && a.Tp >= a.T && a.T == a.B * a.S, "transformer encoder: bad token padding");
    MCTSQ_CHECK(a.NS >= 1 && a.NS * a.S <= AROWS, "transformer encoder: samples per workgroup");
    MCTSQ_CHECK(a.ffn_prep[0] && a.ffn_prep[1], "transformer encoder needs prepared FFN weights");
    hipLaunchKernelGGL(tf_embed_kernel, dim3((unsigned)(a.Tp / EMB_TOK)), dim3(256), 0, st, a);
    const unsigned attn_grid = (unsigned)((a.B + a.NS - 1) / a.NS), ffn_grid = (unsigned)(a.Tp / FFN_TOK);
    for (int l = 0; l < 2; ++l) {
        hipLaunchKernelGGL(tf_attn_kernel, dim3(attn_grid), dim3(256), 0, st, a, l);
        hipLaunchKernelGGL(tf_ffn_kernel<MCTSQ_FFN_NT>, dim3(ffn_grid), dim3(64 * FFN_WAVES), 0, st, a, l);
    }
    hipLaunchKernelGGL(tf_pool_kernel, dim3((unsigned)((a.Bp + 3) / 4)), dim3(256), 0, st, a);
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

static int tf_check(const TfArgs& a) {
    MCTSQ_CHECK(a.S >= 1 && a.S <= 32, "transformer encoder: seq_len %d outside 1..32", a.S);
    MCTSQ_CHECK(a.Tp % TOK_PER_WG == 0 && a.Tp >= a.T && a.T == a.B * a.S, "transformer encoder: bad token padding");
    MCTSQ_CHECK(a.NS >= 1 && a.NS * a.S <= AROWS, "transformer encoder: samples per workgroup");
    MCTSQ_CHECK(a.ffn_prep[0] && a.ffn_prep[1], "transformer encoder needs prepared FFN weights");
    return 0;
}

// two transformer encoders of one network: same stages, the FFN of both as one grid per layer
int launch_tf_fwd_pair(const TfArgs& a, const TfArgs& b, hipStream_t st) {
    if (a.B == 0) return 0;
    if (tf_check(a) || tf_check(b)) return -1;
    MCTSQ_CHECK(a.B == b.B, "transformer encoder pair: batch mismatch");
    hipLaunchKernelGGL(tf_embed_pair_kernel, dim3((unsigned)(a.Tp / EMB_TOK + b.Tp / EMB_TOK)), dim3(256), 0, st, a, b, (int)(a.Tp / EMB_TOK));
    const int na = (int)(a.Tp / FFN_TOK), nb = (int)(b.Tp / FFN_TOK);
    for (int l = 0; l < 2; ++l) {
        const int ga = (int)((a.B + a.NS - 1) / a.NS), gb = (int)((b.B + b.NS - 1) / b.NS);
        hipLaunchKernelGGL(tf_attn_pair_kernel, dim3((unsigned)(ga + gb)), dim3(256), 0, st, a, b, ga, l);
        hipLaunchKernelGGL(tf_ffn_pair_kernel<MCTSQ_FFN_NT>, dim3((unsigned)(na + nb)), dim3(64 * FFN_WAVES), 0, st,
                           a, b, na, l);
    }
    hipLaunchKernelGGL(tf_pool_pair_kernel, dim3((unsigned)((a.Bp + 3) / 4 + (b.Bp + 3) / 4)), dim3(256), 0, st, a, b,
                       (int)((a.Bp + 3) / 4));
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

}  // namespace mctsq
