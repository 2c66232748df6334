// Micro-benchmark: how much VALU work hides in the shadow of the fp16 matrix instruction v_mfma_f32_16x16x32_f16
// (a) issued by the SAME wave between its MFMAs, (b) with one or two waves per SIMD.
// Loop body: 8 x (1 MFMA + N independent VALU ops); WPS waves per SIMD (block = 256 * WPS threads, 256 blocks).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma16_valu_ubench.hip -o tools/bin/mfma16_valu_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int N, int NACC>
__global__ void kern(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(a + i); y[i] = (_Float16)(b - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[u % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < N; ++n) v[n & 7] = __builtin_fmaf(v[n & 7], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, int NACC>
void run(int wps) {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<N, NACC>), dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kern<N, NACC>), dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD %d  acc chains %d  N=%2d: %6.2f ns per (MFMA + N valu) per wave, %6.2f ns per SIMD-MFMA\n", wps, NACC, N,
           ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) / wps);
    (void)hipFree(out);
}

int main() {
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, 1>(wps); run<0, 4>(wps); run<1, 4>(wps); run<2, 4>(wps); run<3, 4>(wps); run<4, 4>(wps); run<6, 4>(wps); run<8, 4>(wps);
    }
    return 0;
}
