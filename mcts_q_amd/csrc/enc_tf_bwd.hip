// TransformerEnc (dqn_per.py:80-101) backward: hand-written reverse pass of enc_tf.hip.
//
//   pool/LayerNorm bwd -> for layer 1, 0: { LayerNorm2 bwd -> FFN bwd -> LayerNorm1 bwd -> attention bwd }
//
// Like the other encoders, the kernels here produce pre-activation gradients ("deltas") per token;
// the small weight gradients are K = tokens contractions done by the generic wgrad kernel
// (learner.hip).  The two FFN matrices are the exception: their gradients are accumulated inside
// the FFN backward kernel, whose workgroups each own 128 hidden units and walk a block of tokens,
// so dW1/dW2 stay in registers (no atomics) and only per-token-block partials go to memory.
#include "device_utils.hpp"
#include "learner.hpp"

namespace mctsq {

constexpr int FF = 2048;

// ------------------------------------------------------------------------ small helpers
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int rows, int cols) {     // out[c][r] = in[r][c]
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (by + i < rows && bx + tx < cols) tile[i][tx] = in[(int64_t)(by + i) * cols + bx + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (bx + i < cols && by + tx < rows) out[(int64_t)(bx + i) * rows + by + tx] = tile[tx][i];
}

// feats = LayerNorm(mean_t h): dfeat -> gradient of every token of the last block output
__global__ __launch_bounds__(256) void tf_pool_bwd_kernel(TfBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= a.f.B) return;
    const float m = a.f.sv_mean[b * E + lane];
    const float mean = wave_sum(m) * (1.0f / E);
    const float d = m - mean;
    const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.0f / E) + LN_EPS);
    const float xh = d * rstd;
    const float df = a.dfeat[b * FEAT + a.f.out_col + lane];
    a.pfin[b * E + lane] = df * xh;
    const float g = df * a.f.lnw[lane];
    const float s1 = wave_sum(g) * (1.0f / E), s2 = wave_sum(g * xh) * (1.0f / E);
    const float dm = rstd * (g - s1 - xh * s2) / (float)a.f.S;
    for (int t = 0; t < a.f.S; ++t) a.dy2[1][(b * a.f.S + t) * E + lane] = dm;
}

// dy = base (+ sum of `np` partial slabs); dx = LayerNorm backward of dy; p = dy * xhat.
// One wave per token, lane = channel.
__global__ __launch_bounds__(256) void tf_ln_bwd_kernel(const float* __restrict__ base, const float* __restrict__ part,
                                                        int np, int64_t part_stride, const float* __restrict__ x,
                                                        const float* __restrict__ st, const float* __restrict__ lnw,
                                                        float* __restrict__ dy_out, float* __restrict__ dx,
                                                        float* __restrict__ p, int64_t T) {
    const int lane = threadIdx.x & 63;
    const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= T) return;
    float dy = base[tok * E + lane];
    for (int i = 0; i < np; ++i) dy += part[i * part_stride + tok * E + lane];
    if (dy_out) dy_out[tok * E + lane] = dy;
    const float mean = st[tok * 2], rstd = st[tok * 2 + 1];
    const float xh = (x[tok * E + lane] - mean) * rstd;
    p[tok * E + lane] = dy * xh;
    const float g = dy * lnw[lane];
    const float s1 = wave_sum(g) * (1.0f / E), s2 = wave_sum(g * xh) * (1.0f / E);
    dx[tok * E + lane] = rstd * (g - s1 - xh * s2);
}

// ------------------------------------------------------------------------------- FFN backward
// grid (token blocks, 16 hidden groups); wave w of a workgroup owns hidden tiles hg*8 + 2w, +1.
// u is recomputed in both orientations (as u^T for the dh chain, as u for the weight gradients):
// swapping the MFMA operands gives the transposed product from the same registers.
constexpr int FB_TOK = 1024;

struct FfnBwdLds {
    float dh[4][TILE * 68];
};

__global__ __launch_bounds__(256) void tf_ffn_bwd_kernel(TfBwdArgs a, int layer) {
    __shared__ __attribute__((aligned(16))) FfnBwdLds L;
    const TfLayerW& W = a.f.layer[layer];
    const float* hmid = a.f.buf[layer].hmid;
    const float* dx2 = a.dx2[layer];
    const float* w1t = a.w1t[layer];
    const float* w2t = a.w2t[layer];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, jl = lane & 15, q = lane >> 4;
    const int tb = blockIdx.x, hg = blockIdx.y;

    float w1f[2][16], w2tf[2][16];
    float4 w1b[2][4], b1r[2];
    float b1c[2], db1[2] = {0.0f, 0.0f};
    f32x4 accW1[2][4], accW2[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int nt = hg * 8 + 2 * w + i;
        const float4* p1 = reinterpret_cast<const float4*>(W.l1w + (int64_t)(16 * nt + jl) * E + 16 * q);
        const float4* p2 = reinterpret_cast<const float4*>(w2t + (int64_t)(16 * nt + jl) * E + 16 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 u = p1[c], v = p2[c];
            w1f[i][4 * c] = u.x; w1f[i][4 * c + 1] = u.y; w1f[i][4 * c + 2] = u.z; w1f[i][4 * c + 3] = u.w;
            w2tf[i][4 * c] = v.x; w2tf[i][4 * c + 1] = v.y; w2tf[i][4 * c + 2] = v.z; w2tf[i][4 * c + 3] = v.w;
            w1b[i][c] = *reinterpret_cast<const float4*>(w1t + (int64_t)(16 * c + jl) * FF + 16 * nt + 4 * q);
            accW1[i][c] = (f32x4){0, 0, 0, 0};
            accW2[i][c] = (f32x4){0, 0, 0, 0};
        }
        b1r[i] = *reinterpret_cast<const float4*>(W.l1b + 16 * nt + 4 * q);
        b1c[i] = W.l1b[16 * nt + jl];
    }
    const int64_t tok_lo = (int64_t)tb * FB_TOK;
    const int64_t tok_hi = tok_lo + FB_TOK < a.f.Tp ? tok_lo + FB_TOK : a.f.Tp;
    float* dhp = a.dhpart + (int64_t)hg * a.f.Tp * E;
    for (int64_t t0 = tok_lo; t0 < tok_hi; t0 += TILE) {
        float hf[16], df[16], hd[4][4], dd[4][4];
        {
            const float4* ph = reinterpret_cast<const float4*>(hmid + (t0 + jl) * E + 16 * q);
            const float4* pd = reinterpret_cast<const float4*>(dx2 + (t0 + jl) * E + 16 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 u = ph[c], v = pd[c];
                hf[4 * c] = u.x; hf[4 * c + 1] = u.y; hf[4 * c + 2] = u.z; hf[4 * c + 3] = u.w;
                df[4 * c] = v.x; df[4 * c + 1] = v.y; df[4 * c + 2] = v.z; df[4 * c + 3] = v.w;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hd[c][r] = hmid[(t0 + 4 * q + r) * E + 16 * c + jl];
                    dd[c][r] = dx2[(t0 + 4 * q + r) * E + 16 * c + jl];
                }
        }
        f32x4 dhacc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 uT = {b1r[i].x, b1r[i].y, b1r[i].z, b1r[i].w};            // [hidden 4q+r][token jl]
            f32x4 u = {b1c[i], b1c[i], b1c[i], b1c[i]};                     // [token 4q+r][hidden jl]
            f32x4 duT = {0, 0, 0, 0}, du = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                uT = mfma16(w1f[i][e], hf[e], uT);
                u = mfma16(hf[e], w1f[i][e], u);
                duT = mfma16(w2tf[i][e], df[e], duT);
                du = mfma16(df[e], w2tf[i][e], du);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                duT[r] = uT[r] > 0.0f ? duT[r] : 0.0f;
                du[r] = u[r] > 0.0f ? du[r] : 0.0f;
                u[r] = fmaxf(u[r], 0.0f);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                dhacc[c] = mfma16(duT[0], w1b[i][c].x, dhacc[c]); dhacc[c] = mfma16(duT[1], w1b[i][c].y, dhacc[c]);
                dhacc[c] = mfma16(duT[2], w1b[i][c].z, dhacc[c]); dhacc[c] = mfma16(duT[3], w1b[i][c].w, dhacc[c]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accW2[i][c] = mfma16(dd[c][r], u[r], accW2[i][c]);      // dW2[c'][n] += dout[tok][c'] u[tok][n]
                    accW1[i][c] = mfma16(du[r], hd[c][r], accW1[i][c]);     // dW1[n][k] += du[tok][n] h[tok][k]
                }
            }
            db1[i] += (du[0] + du[1]) + (du[2] + du[3]);
        }
        // sum the four waves' dh contributions (their hidden units differ) and store this group's slab
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) L.dh[w][(4 * q + r) * 68 + 16 * c + jl] = dhacc[c][r];
        __syncthreads();
        {
            const int row = tid >> 4, c4 = (tid & 15) * 4;
            const float4 s0 = *reinterpret_cast<const float4*>(&L.dh[0][row * 68 + c4]);
            const float4 s1 = *reinterpret_cast<const float4*>(&L.dh[1][row * 68 + c4]);
            const float4 s2 = *reinterpret_cast<const float4*>(&L.dh[2][row * 68 + c4]);
            const float4 s3 = *reinterpret_cast<const float4*>(&L.dh[3][row * 68 + c4]);
            *reinterpret_cast<float4*>(dhp + (t0 + row) * E + c4) =
                make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y),
                            (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
        }
        __syncthreads();
    }
    // ---- weight-gradient partials of this token block ------------------------------------------------
    float* P1 = a.wpart1[layer] + (int64_t)tb * ((int64_t)FF * E + FF);          // dW1 [2048][64] | db1 [2048]
    float* P2 = a.wpart2[layer] + (int64_t)tb * ((int64_t)E * FF + E);           // dW2 [64][2048] | (unused 64)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int nt = hg * 8 + 2 * w + i;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                P1[(int64_t)(16 * nt + 4 * q + r) * E + 16 * c + jl] = accW1[i][c][r];
                P2[(int64_t)(16 * c + 4 * q + r) * FF + 16 * nt + jl] = accW2[i][c][r];
            }
        float s = db1[i];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (q == 0) P1[(int64_t)FF * E + 16 * nt + jl] = s;
    }
}

// --------------------------------------------------------------------------- attention backward
// One workgroup = NSB whole samples (<= 64 token rows).  dx1 -> d(ctx) = dx1 . W_o; per (sample, head)
// softmax backward on the vector ALU (S <= 32: one lane per query row, then one lane per key row);
// d(qkv) -> out (for the in_proj weight gradient) and d(h_in) = dx1 + d(qkv) . W_in.
constexpr int BROWS = 64, BHLD = 68, BQLD = 196, PLD = 33;

struct AttnBwdLds {
    float dctx[BROWS * BHLD];
    float qkv[BROWS * BQLD];
    float dqkv[BROWS * BQLD];
    union {                              // dx1 is only needed for the d(ctx) product, before P / dS exist
        float dx1[BROWS * BHLD];
        struct { float P[4][32 * PLD]; float dS[4][32 * PLD]; } s;
    } u;
};

__global__ __launch_bounds__(256) void tf_attn_bwd_kernel(TfBwdArgs a, int layer) {
    __shared__ __attribute__((aligned(16))) AttnBwdLds L;
    const TfLayerW& W = a.f.layer[layer];
    const TfLayerBuf& Bf = a.f.buf[layer];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, jl = lane & 15, q = lane >> 4;
    const int S = a.f.S, NS = a.nsb;
    const int64_t sample0 = (int64_t)blockIdx.x * NS;
    const int ns = (int)min((int64_t)NS, a.f.B - sample0);
    const int rows = ns * S;
    const int64_t tok0 = sample0 * S;
    const float* dx1g = a.dx1[layer];
    float* dyn = layer == 0 ? a.dh0 : a.dy2[layer - 1];      // gradient of this block's input

    for (int idx = tid; idx < BROWS * 16; idx += 256) {
        const int r = idx >> 4, c4 = idx & 15;
        float4 v = make_float4(0, 0, 0, 0);
        if (r < rows) v = reinterpret_cast<const float4*>(dx1g + (tok0 + r) * E)[c4];
        *reinterpret_cast<float4*>(&L.u.dx1[r * BHLD + 4 * c4]) = v;
    }
    for (int idx = tid; idx < BROWS * 48; idx += 256) {
        const int r = idx / 48, c4 = idx - r * 48;
        float4 v = make_float4(0, 0, 0, 0);
        if (r < rows) v = reinterpret_cast<const float4*>(Bf.qkv + (tok0 + r) * G3)[c4];
        *reinterpret_cast<float4*>(&L.qkv[r * BQLD + 4 * c4]) = v;
        *reinterpret_cast<float4*>(&L.dqkv[r * BQLD + 4 * c4]) = make_float4(0, 0, 0, 0);
    }
    __syncthreads();
    // ---- d(ctx)[row][k] = sum_c dx1[row][c] W_o[c][k]; wave w owns rows 16w..16w+15 ---------------
    {
        const float4* p = reinterpret_cast<const float4*>(&L.u.dx1[(16 * w + jl) * BHLD + 16 * q]);
        float af[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float4 v = p[c]; af[4 * c] = v.x; af[4 * c + 1] = v.y; af[4 * c + 2] = v.z; af[4 * c + 3] = v.w; }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 16; ++e) acc = mfma16(af[e], W.ow[(int64_t)(16 * q + e) * E + 16 * cc + jl], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) L.dctx[(16 * w + 4 * q + r) * BHLD + 16 * cc + jl] = acc[r];
        }
    }
    __syncthreads();
    // ---- softmax / attention backward per (sample, head) -------------------------------------------
    float* Pw = L.u.s.P[w];
    float* dSw = L.u.s.dS[w];
    for (int p = w; p < NS * 4; p += 4) {                         // uniform trip count (barriers inside)
        const int si = p >> 2, hh = p & 3, rb = si * S;
        const bool live = si < ns;
        if (live && lane < S) {                                   // lane = query i
            const int i = lane;
            float qi[16], dc[16], dq[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                qi[d] = L.qkv[(rb + i) * BQLD + 16 * hh + d];
                dc[d] = L.dctx[(rb + i) * BHLD + 16 * hh + d];
                dq[d] = 0.0f;
            }
            float mx = -INFINITY;
            for (int j = 0; j < S; ++j) {
                const float* kj = &L.qkv[(rb + j) * BQLD + E + 16 * hh];
                float s = 0.0f;
#pragma unroll
                for (int d = 0; d < 16; ++d) s = fmaf(qi[d], kj[d], s);
                Pw[i * PLD + j] = s;
                mx = fmaxf(mx, s);
            }
            float sum = 0.0f;
            for (int j = 0; j < S; ++j) {
                const float e = fast_exp2((Pw[i * PLD + j] - mx) * LOG2E);      // as the forward evaluates it
                Pw[i * PLD + j] = e;
                sum += e;
            }
            const float inv = fast_rcp(sum);
            float Dd = 0.0f;
            for (int j = 0; j < S; ++j) {
                const float* vj = &L.qkv[(rb + j) * BQLD + 2 * E + 16 * hh];
                float dp = 0.0f;
#pragma unroll
                for (int d = 0; d < 16; ++d) dp = fmaf(dc[d], vj[d], dp);
                const float pij = Pw[i * PLD + j] * inv;
                Pw[i * PLD + j] = pij;
                dSw[i * PLD + j] = dp;
                Dd = fmaf(pij, dp, Dd);
            }
            for (int j = 0; j < S; ++j) {
                const float ds = Pw[i * PLD + j] * (dSw[i * PLD + j] - Dd);
                dSw[i * PLD + j] = ds;
                const float* kj = &L.qkv[(rb + j) * BQLD + E + 16 * hh];
#pragma unroll
                for (int d = 0; d < 16; ++d) dq[d] = fmaf(ds, kj[d], dq[d]);
            }
#pragma unroll
            for (int d = 0; d < 16; ++d) L.dqkv[(rb + i) * BQLD + 16 * hh + d] = 0.25f * dq[d];   // q was scaled by 1/4
        }
        __syncthreads();
        if (live && lane < S) {                                   // lane = key j
            const int j = lane;
            float dk[16], dv[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) { dk[d] = 0.0f; dv[d] = 0.0f; }
            for (int i = 0; i < S; ++i) {
                const float ds = dSw[i * PLD + j], pij = Pw[i * PLD + j];
                const float* qi = &L.qkv[(rb + i) * BQLD + 16 * hh];
                const float* dci = &L.dctx[(rb + i) * BHLD + 16 * hh];
#pragma unroll
                for (int d = 0; d < 16; ++d) { dk[d] = fmaf(ds, qi[d], dk[d]); dv[d] = fmaf(pij, dci[d], dv[d]); }
            }
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                L.dqkv[(rb + j) * BQLD + E + 16 * hh + d] = dk[d];
                L.dqkv[(rb + j) * BQLD + 2 * E + 16 * hh + d] = dv[d];
            }
        }
        __syncthreads();
    }
    // ---- d(qkv) out; d(h_in) = dx1 + d(qkv) . W_in ---------------------------------------------------
    for (int idx = tid; idx < BROWS * 48; idx += 256) {
        const int r = idx / 48, c4 = idx - r * 48;
        if (r < rows)
            reinterpret_cast<float4*>(a.dqkv[layer] + (tok0 + r) * G3)[c4] = *reinterpret_cast<const float4*>(&L.dqkv[r * BQLD + 4 * c4]);
    }
    {
        const float4* p = reinterpret_cast<const float4*>(&L.dqkv[(16 * w + jl) * BQLD + 48 * q]);
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int g = 0; g < 12; ++g) {
            const float4 v = p[g];
            const float* wr = W.inw + (int64_t)(48 * q + 4 * g) * E + jl;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                acc[cc] = mfma16(v.x, wr[16 * cc], acc[cc]);         acc[cc] = mfma16(v.y, wr[E + 16 * cc], acc[cc]);
                acc[cc] = mfma16(v.z, wr[2 * E + 16 * cc], acc[cc]); acc[cc] = mfma16(v.w, wr[3 * E + 16 * cc], acc[cc]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * w + 4 * q + r;
                if (row < rows) dyn[(tok0 + row) * E + 16 * cc + jl] = acc[cc][r] + dx1g[(tok0 + row) * E + 16 * cc + jl];
            }
    }
}

int launch_tf_bwd(const TfBwdArgs& a, hipStream_t st) {
    const TfArgs& f = a.f;
    MCTSQ_CHECK(f.S >= 1 && f.S <= 32, "transformer backward: seq_len %d outside 1..32", f.S);
    MCTSQ_CHECK(a.nsb >= 1 && a.nsb * f.S <= BROWS, "transformer backward: samples per workgroup");
    MCTSQ_CHECK(f.T == f.B * f.S && f.T % TILE == 0 && f.Tp % TILE == 0, "transformer backward: token count must be a multiple of 16");
    const unsigned tokw = (unsigned)((f.T + 3) / 4);
    const int ntb = (int)((f.Tp + FB_TOK - 1) / FB_TOK);
    MCTSQ_CHECK(ntb == a.ntb, "transformer backward: token block plan mismatch");
    hipLaunchKernelGGL(tf_pool_bwd_kernel, dim3((unsigned)((f.B + 3) / 4)), dim3(256), 0, st, a);
    for (int l = 1; l >= 0; --l) {
        const TfLayerW& W = f.layer[l];
        hipLaunchKernelGGL(transpose_kernel, dim3(E / 32, FF / 32), dim3(256), 0, st, W.l1w, a.w1t[l], FF, E);
        hipLaunchKernelGGL(transpose_kernel, dim3(FF / 32, E / 32), dim3(256), 0, st, W.l2w, a.w2t[l], E, FF);
        hipLaunchKernelGGL(tf_ln_bwd_kernel, dim3(tokw), dim3(256), 0, st, (const float*)a.dy2[l], (const float*)nullptr,
                           0, (int64_t)0, (const float*)f.buf[l].x2, (const float*)f.buf[l].st2, W.n2w, (float*)nullptr,
                           a.dx2[l], a.p2[l], f.T);
        hipLaunchKernelGGL(tf_ffn_bwd_kernel, dim3((unsigned)ntb, 16), dim3(256), 0, st, a, l);
        hipLaunchKernelGGL(tf_ln_bwd_kernel, dim3(tokw), dim3(256), 0, st, (const float*)a.dx2[l],
                           (const float*)a.dhpart, 16, f.Tp * (int64_t)E, (const float*)f.buf[l].x1,
                           (const float*)f.buf[l].st1, W.n1w, a.dy1[l], a.dx1[l], a.p1[l], f.T);
        hipLaunchKernelGGL(tf_attn_bwd_kernel, dim3((unsigned)((f.B + a.nsb - 1) / a.nsb)), dim3(256), 0, st, a, l);
    }
    MCTSQ_HIP(hipGetLastError());
    return 0;
}

int tf_bwd_token_blocks(int64_t Tp) { return (int)((Tp + FB_TOK - 1) / FB_TOK); }

}  // namespace mctsq
