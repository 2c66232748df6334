// Accuracy of the two split schemes the matrix kernels use, against a double-precision product, and the matrix
// instruction's treatment of fp16 subnormal inputs.  One wave computes D = A(16x64) . B(64x16) as
//   (a) split-bf16: three planes per operand, six cross terms (device_utils.hpp: split3 / mfma_bf)
//   (b) split-fp16: two planes per operand scaled into the upper fp16 range, three cross terms (split2h / mfma_h)
//   (c) a plain fp32 fmaf chain in k order
// for operands drawn like the GRU recurrence's (|h| <= 1, weights ~U(-1/8, 1/8)) and for a few scales of them.
// Build: hipcc --offload-arch=gfx950 -O3 -I mcts_q_amd/csrc tools/split_ubench.hip -o tools/bin/split_ubench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "device_utils.hpp"
using namespace mctsq;

__global__ void prod(const float* A, const float* B, float* Dbf, float* Dh, float* Df, int shA, int shB) {
    const int lane = threadIdx.x, jl = lane & 15, q = lane >> 4;
    f32x4 abf = {0, 0, 0, 0}, ah = {0, 0, 0, 0};
    for (int c = 0; c < 2; ++c) {
        const float4* pa = reinterpret_cast<const float4*>(A + jl * 64 + 32 * c + 8 * q);
        const float4* pb = reinterpret_cast<const float4*>(B + jl * 64 + 32 * c + 8 * q);   // B stored [n][k]
        u32x4 a0, a1, a2, b0, b1, b2;
        split8(pa[0], pa[1], a0, a1, a2);
        split8(pb[0], pb[1], b0, b1, b2);
        abf = mfma_bf(a2, b0, abf); abf = mfma_bf(a0, b2, abf); abf = mfma_bf(a1, b1, abf);
        abf = mfma_bf(a1, b0, abf); abf = mfma_bf(a0, b1, abf); abf = mfma_bf(a0, b0, abf);
        float4 x0 = pa[0], x1 = pa[1], y0 = pb[0], y1 = pb[1];
        const float sa = pow2i(shA), sb = pow2i(shB);
        x0.x *= sa; x0.y *= sa; x0.z *= sa; x0.w *= sa; x1.x *= sa; x1.y *= sa; x1.z *= sa; x1.w *= sa;
        y0.x *= sb; y0.y *= sb; y0.z *= sb; y0.w *= sb; y1.x *= sb; y1.y *= sb; y1.z *= sb; y1.w *= sb;
        u32x4 ahi, alo, bhi, blo;
        split8h(x0, x1, ahi, alo);
        split8h(y0, y1, bhi, blo);
        ah = mfma_h(alo, bhi, ah); ah = mfma_h(ahi, blo, ah); ah = mfma_h(ahi, bhi, ah);
    }
    const float inv = pow2i(-shA - shB);
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * q + r;
        Dbf[i * 16 + jl] = abf[r];
        Dh[i * 16 + jl] = ah[r] * inv;
        float s = 0.0f;
        for (int k = 0; k < 64; ++k) s = fmaf(A[i * 64 + k], B[jl * 64 + k], s);
        Df[i * 16 + jl] = s;
    }
}

// one product with a subnormal fp16 operand: a = 2^-20 (subnormal in fp16), b = 2^10 -> 2^-10 if kept, 0 if flushed
__global__ void denorm(float* out) {
    const int lane = threadIdx.x;
    u32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if ((lane >> 4) == 0) { a[0] = 0x0010u; b[0] = 0x6400u; }      // element k = 0: 2^-20 (0x0010 = 16 * 2^-24), 1024
    f32x4 d = mfma_h(a, b, (f32x4){0, 0, 0, 0});
    if (lane == 0) { out[0] = d[0]; out[1] = (float)(_Float16)(9.5367431640625e-07f); }   // and the conversion of 2^-20
}

static double frand() { return rand() / (double)RAND_MAX; }

int main() {
    float *A, *B, *Dbf, *Dh, *Df, *dn;
    hipMallocManaged(&A, 16 * 64 * 4); hipMallocManaged(&B, 16 * 64 * 4);
    hipMallocManaged(&Dbf, 1024); hipMallocManaged(&Dh, 1024); hipMallocManaged(&Df, 1024); hipMallocManaged(&dn, 16);
    hipLaunchKernelGGL(denorm, dim3(1), dim3(64), 0, 0, dn);
    hipDeviceSynchronize();
    printf("fp16 subnormal input 2^-20 x 1024 -> %g (kept: %g); (fp16)2^-20 -> %g\n", dn[0], 1.0 / 1024, dn[1]);
    const struct { const char* name; double amag, bmag, small_frac; } cases[] = {
        {"h~U(-1,1), w~U(-1/8,1/8)", 1.0, 0.125, 0.0},
        {"h~U(-1,1), w~U(-8,8)", 1.0, 8.0, 0.0},
        {"h~U(-.01,.01), w~U(-1/8,1/8)", 0.01, 0.125, 0.0},
        {"h: 90% scaled 1e-4, w~U(-1/8,1/8)", 1.0, 0.125, 0.9},
    };
    srand(3);
    for (const auto& cs : cases) {
        double ebf = 0, eh = 0, ef = 0, nrm = 0;
        const int trials = 200;
        for (int t = 0; t < trials; ++t) {
            double bmax = 0;
            for (int i = 0; i < 16 * 64; ++i) {
                double a = (2 * frand() - 1) * cs.amag;
                if (frand() < cs.small_frac) a *= 1e-4;
                A[i] = (float)a;
                B[i] = (float)((2 * frand() - 1) * cs.bmag);
                bmax = fmax(bmax, fabs(B[i]));
            }
            int e; frexp(bmax, &e);                              // bmax in [2^(e-1), 2^e)
            hipLaunchKernelGGL(prod, dim3(1), dim3(64), 0, 0, A, B, Dbf, Dh, Df, 14, 14 - (e - 1));
            hipDeviceSynchronize();
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double s = 0, sa = 0;
                    for (int k = 0; k < 64; ++k) { s += (double)A[i * 64 + k] * B[j * 64 + k]; sa += fabs((double)A[i * 64 + k] * B[j * 64 + k]); }
                    ebf = fmax(ebf, fabs(Dbf[i * 16 + j] - s) / sa);
                    eh = fmax(eh, fabs(Dh[i * 16 + j] - s) / sa);
                    ef = fmax(ef, fabs(Df[i * 16 + j] - s) / sa);
                    nrm += 1;
                }
        }
        printf("%-36s max |err| / sum|a b|:  split-bf16 (6 terms) %.2e   split-fp16 (3 terms) %.2e   fp32 fma chain %.2e\n",
               cs.name, ebf, eh, ef);
    }
    return 0;
}
