// Micro-benchmark: does VALU work issued between MFMAs of the SAME wave overlap with the matrix pipe?
// One wave per SIMD (256 threads/block, 256 blocks).  Loop body: 1 MFMA 16x16x4 f32 + N VALU ops.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_ubench.hip -o tools/bin/mfma_valu_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, int KIND>
__global__ void kern(float* out, int iters, float a, float b) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == 0) v[n & 7] = __builtin_fmaf(v[n & 7], b, a);                 // plain VALU
                else if (KIND == 1) v[n & 7] = __builtin_amdgcn_exp2f(v[n & 7]);          // transcendental
                else v[n & 7] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[n & 7]), 0x128, 0xf, 0xf, false));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = acc0[0] + acc0[1] + acc1[2] + acc1[3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, int KIND>
void run(const char* name) {
    float* out;
    (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<N, KIND>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kern<N, KIND>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-6s N=%d: %.2f ns per (MFMA + N ops)\n", name, N, ms * 1e6 / (iters * 8.0));
    (void)hipFree(out);
}

int main() {
    run<0, 0>("valu"); run<2, 0>("valu"); run<4, 0>("valu"); run<6, 0>("valu"); run<8, 0>("valu"); run<12, 0>("valu");
    run<1, 1>("trans"); run<2, 1>("trans"); run<4, 1>("trans");
    run<1, 2>("dpp"); run<2, 2>("dpp"); run<4, 2>("dpp");
    return 0;
}
