// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_f16 when every instruction reads DIFFERENT operand registers
// (as a real product does) against the same registers every time; 1, 2 waves per SIMD; 4 or 8 accumulator chains;
// dependent runs of 3 on one accumulator (the split product's three terms) against round-robin.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma16_operand_ubench.hip -o tools/bin/mfma16_operand_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int DISTINCT, int RUN3>
__global__ void kern(float* out, int iters, float a, float b) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    f16x8 x[8], y[8];
    for (int k = 0; k < 8; ++k)
        for (int i = 0; i < 8; ++i) { x[k][i] = (_Float16)(a + i + k + threadIdx.x * 0.01f); y[k][i] = (_Float16)(b - i - k); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int ai = RUN3 ? (u / 3) % 8 : u % 8;
            const int xi = DISTINCT ? u % 8 : 0, yi = DISTINCT ? (u * 3 + 1) % 8 : 0;
            acc[ai] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x[xi], y[yi], acc[ai], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int DISTINCT, int RUN3>
void run(int wps) {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 3000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kern<DISTINCT, RUN3>), dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kern<DISTINCT, RUN3>), dim3(256), dim3(256 * wps), 0, 0, out, iters, 1.0f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (iters * 24.0) / wps;
    printf("waves/SIMD %d  operands %s  order %s: %6.2f ns per SIMD-MFMA  -> %.0f TFLOP/s dense fp16 on 1024 SIMDs\n", wps,
           DISTINCT ? "distinct" : "same    ", RUN3 ? "runs of 3 per accumulator" : "round robin              ", per,
           16.0 * 16 * 32 * 2 / per * 1024 / 1e3);
    (void)hipFree(out);
}

int main() {
    for (int wps = 1; wps <= 2; ++wps) { run<0, 0>(wps); run<1, 0>(wps); run<0, 1>(wps); run<1, 1>(wps); }
    return 0;
}
