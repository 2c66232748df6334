// The 192-128-128-A head of the shipped networks on the bf16 matrix instruction (exact three-way split of both
// operands, six cross terms, fp32 accumulate: DESIGN.md section 3), shared by the fused gru/gru/gru kernel
// (fused_gru_head.hip) and the stand-alone head kernel (head.hip).  8 waves per 16-row tile:
//   waves 0-3: layer 0, column tiles w and w + 4 (2 x 6 K chunks x 3 planes), then the output layer (4 x 3)
//   waves 4-7: layer 1, column tiles w - 4 and w (2 x 4 x 3), needed only after layer 0 is through
// Weights come as prepared lane-major fragments (head_prepare_kernel), so every load instruction reads 1 KB in
// a row; activations are split by the lanes that produce them and exchanged through LDS planes.
// Reference: dqn_per.py:216-236 (fc_layers) + mcts_per.py:293, 370 (leaf epilogue).
#pragma once
#include "device_utils.hpp"
#include "kernels.hpp"

namespace mctsq {

constexpr int FPL = FEAT + 8;    // row stride (bf16 elements) of a feature plane: 400 B, b128 reads conflict-free
constexpr int APL = 128 + 8;     // same for a hidden-activation plane (272 B)
// prepared head weights (head_prepare_kernel), u32x4 fragments of 64 lanes each:
//   layer 0 [tile 8][K chunk 6][plane 3], layer 1 [tile 8][K chunk 4][plane 3], output layer [K chunk 4][plane 3]
constexpr int H0_FR = 8 * 6 * 3, H1_FR = 8 * 4 * 3, H2_FR = 4 * 3;
static_assert(HEAD_PREP_BYTES == (int64_t)(H0_FR + H1_FR + H2_FR) * 64 * 16, "prepared head size");

// six cross terms of the split product, smallest first
#define HEAD_MM6(ACC, XH, XM, XL, W)                                                               \
    ACC = mfma_bf(XL, (W)[0], ACC); ACC = mfma_bf(XH, (W)[2], ACC); ACC = mfma_bf(XM, (W)[1], ACC); \
    ACC = mfma_bf(XM, (W)[0], ACC); ACC = mfma_bf(XH, (W)[1], ACC); ACC = mfma_bf(XH, (W)[0], ACC);

struct HeadLds {
    unsigned short fpl[3][TILE * FPL];       // the 192 feature columns as bf16 planes (hi, mid, lo)
    unsigned short apl[2][3][TILE * APL];    // hidden layers 0 and 1, same form
};

// RPL (gru_core.hpp): 4 = a 16-row tile, lane row r is row 4q + r; 2 = an 8-row half tile whose row 2q + r sits in
// matrix row 4q + r (r < 2): the products are the same, the per-row work behind them is done for the real rows only
template <bool SAVE, int RPL = 4>
struct HeadWave {
    static __device__ __forceinline__ int srow(int q, int r) { return RPL == 4 ? 4 * q + r : 2 * q + r; }
    u32x4 hw[2][18], hw2[12];
    float hb[2], hb2;

    // issue the fragment loads of this wave's role (they may travel under whatever the caller does next)
    __device__ __forceinline__ void load(const HeadArgs& ha, int lane, int wave) {
        const int jl = lane & 15;
        const u32x4* hp = reinterpret_cast<const u32x4*>(ha.wprep) + lane;
        const int ht = wave & 3;
        hb2 = 0.0f;
        if (wave < 4) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int f = 0; f < 18; ++f) hw[tt][f] = hp[((ht + 4 * tt) * 18 + f) * 64];
                hb[tt] = ha.b[0][16 * (ht + 4 * tt) + jl];
            }
        } else {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int f = 0; f < 12; ++f) hw[tt][f] = hp[(H0_FR + (ht + 4 * tt) * 12 + f) * 64];
                hb[tt] = ha.b[1][16 * (ht + 4 * tt) + jl];
            }
        }
    }

    // L.fpl holds the tile's feature planes and the workgroup has passed a barrier since they were written
    __device__ __forceinline__ void run(const HeadArgs& ha, HeadLds& L, int lane, int wave, int64_t row0) {
        const int jl = lane & 15, q = lane >> 4, ht = wave & 3;
        const u32x4* hp = reinterpret_cast<const u32x4*>(ha.wprep) + lane;
        // ---- head layer 0: [16 x 192] . W0^T on waves 0-3, two column tiles each ----------------
        // All three layers run on the bf16 matrix instruction over the exact three-way split of both operands
        // (six cross terms, fp32 accumulate on top of the bias); activations are split by the lanes that produce them.
        if (wave < 4) {
            f32x4 acc[2][2] = {{{hb[0], hb[0], hb[0], hb[0]}, {0, 0, 0, 0}}, {{hb[1], hb[1], hb[1], hb[1]}, {0, 0, 0, 0}}};
            const int o = (lane & 15) * FPL + 8 * q;
    #pragma unroll
            for (int c = 0; c < 6; ++c) {
                const u32x4 xh = *reinterpret_cast<const u32x4*>(&L.fpl[0][o + 32 * c]);
                const u32x4 xm = *reinterpret_cast<const u32x4*>(&L.fpl[1][o + 32 * c]);
                const u32x4 xl = *reinterpret_cast<const u32x4*>(&L.fpl[2][o + 32 * c]);
                HEAD_MM6(acc[0][c & 1], xh, xm, xl, &hw[0][3 * c])
                HEAD_MM6(acc[1][c & 1], xh, xm, xl, &hw[1][3 * c])
            }
    #pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int col = 16 * (ht + 4 * tt) + jl;
    #pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const float v = apply_act(acc[tt][0][r] + acc[tt][1][r], ha.act);
                    unsigned u0, u1, u2;
                    split3(v, u0, u1, u2);
                    L.apl[0][0][(4 * q + r) * APL + col] = (unsigned short)(u0 >> 16);
                    L.apl[0][1][(4 * q + r) * APL + col] = (unsigned short)(u1 >> 16);
                    L.apl[0][2][(4 * q + r) * APL + col] = (unsigned short)(u2 >> 16);
                    if (SAVE) ha.sv_act[0][(row0 + srow(q, r)) * (int64_t)128 + col] = v;
                }
            }
            // layer 0's fragments are dead: the output layer's take their registers and arrive while waves 4-7 run layer 1
    #pragma unroll
            for (int f = 0; f < 12; ++f) hw2[f] = hp[(H0_FR + H1_FR + f) * 64];
            hb2 = jl < ha.A ? ha.b[2][jl] : 0.0f;
        }
        __syncthreads();
        // ---- head layer 1: [16 x 128] . W1^T on waves 4-7 ----------------------------------------
        if (wave >= 4) {
            f32x4 acc[2][2] = {{{hb[0], hb[0], hb[0], hb[0]}, {0, 0, 0, 0}}, {{hb[1], hb[1], hb[1], hb[1]}, {0, 0, 0, 0}}};
            const int o = (lane & 15) * APL + 8 * q;
    #pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4 xh = *reinterpret_cast<const u32x4*>(&L.apl[0][0][o + 32 * c]);
                const u32x4 xm = *reinterpret_cast<const u32x4*>(&L.apl[0][1][o + 32 * c]);
                const u32x4 xl = *reinterpret_cast<const u32x4*>(&L.apl[0][2][o + 32 * c]);
                HEAD_MM6(acc[0][c & 1], xh, xm, xl, &hw[0][3 * c])
                HEAD_MM6(acc[1][c & 1], xh, xm, xl, &hw[1][3 * c])
            }
    #pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int col = 16 * (ht + 4 * tt) + jl;
    #pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const float v = apply_act(acc[tt][0][r] + acc[tt][1][r], ha.act);
                    unsigned u0, u1, u2;
                    split3(v, u0, u1, u2);
                    L.apl[1][0][(4 * q + r) * APL + col] = (unsigned short)(u0 >> 16);
                    L.apl[1][1][(4 * q + r) * APL + col] = (unsigned short)(u1 >> 16);
                    L.apl[1][2][(4 * q + r) * APL + col] = (unsigned short)(u2 >> 16);
                    if (SAVE) ha.sv_act[1][(row0 + srow(q, r)) * (int64_t)128 + col] = v;
                }
            }
        }
        __syncthreads();
        // ---- output layer: [16 x 128] . W2^T (A <= 16 columns).  Waves 0-3 all form the full product (24 matrix
        // instructions, side by side on the four SIMDs) and each finishes one of the lane's four rows: the
        // epilogue is a serial dependency chain, so four short ones beat one long one.
        if (wave < 4) {
            f32x4 acc0 = {hb2, hb2, hb2, hb2}, acc1 = {0, 0, 0, 0};
            const int o = (lane & 15) * APL + 8 * q;
    #pragma unroll
            for (int c = 0; c < 4; c += 2) {
                const u32x4 xh = *reinterpret_cast<const u32x4*>(&L.apl[1][0][o + 32 * c]);
                const u32x4 xm = *reinterpret_cast<const u32x4*>(&L.apl[1][1][o + 32 * c]);
                const u32x4 xl = *reinterpret_cast<const u32x4*>(&L.apl[1][2][o + 32 * c]);
                const u32x4 yh = *reinterpret_cast<const u32x4*>(&L.apl[1][0][o + 32 * c + 32]);
                const u32x4 ym = *reinterpret_cast<const u32x4*>(&L.apl[1][1][o + 32 * c + 32]);
                const u32x4 yl = *reinterpret_cast<const u32x4*>(&L.apl[1][2][o + 32 * c + 32]);
                HEAD_MM6(acc0, xh, xm, xl, &hw2[3 * c])
                HEAD_MM6(acc1, yh, ym, yl, &hw2[3 * c + 3])
            }
            // ---- leaf epilogue in registers: lane (jl, q) holds Q[row 4q + r][action jl]; max / first argmax /
            // softmax (mcts_per.py:293, 370) are reductions over the 16 lanes of a row group
            // (lane position re-derived here: values computed at kernel entry would sit in registers across every loop)
            const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), jl = ln & 15, q = ln >> 4;
            const bool colok = jl < ha.A;
            if (ht < RPL) {                                   // (a half tile has two real rows per lane: waves 0, 1)
                const int r = ht;
                const int64_t row = row0 + srow(q, r);
                const float v = r == 0 ? acc0[0] + acc1[0] : r == 1 ? acc0[1] + acc1[1] : r == 2 ? acc0[2] + acc1[2] : acc0[3] + acc1[3];
                const float best = row16_max_dpp(colok ? v : -INFINITY);
                const float first = row16_min_dpp((colok && v == best) ? (float)jl : 99.0f);   // like torch.argmax
                const float e = colok ? expf(v - best) : 0.0f;
                const float den = row16_sum(e);
                if (row < ha.B) {
                    if (colok) {
                        ha.q[row * ha.A + jl] = v;
                        if (ha.prior) ha.prior[row * ha.A + jl] = e / den;
                    }
                    if (jl == 0) {
                        if (ha.vmax) ha.vmax[row] = best;
                        if (ha.argmax) ha.argmax[row] = (int)first;
                    }
                }
            }
        }
    }
};

}  // namespace mctsq
