// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2_f32 with 1..4 independent chains,
// 1 or 3 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_ubench.hip -o build/mfma_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ void k16(float* out, long long* cyc, int iters, float a, float b) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
__global__ void k32(float* out, long long* cyc, int iters, float a, float b) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int threads, int ch, double flop_per_mfma) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double n_per_wave = (double)iters * 8 * ch;
    double waves = 256.0 * threads / 64;
    // s_memtime ticks at 100 MHz constant? report both
    printf("%-28s threads=%4d chains=%d : %.1f ns/MFMA/wave  memtime-ticks/MFMA %.2f  chip %.1f TFLOP/s\n", name,
           threads, ch, ms * 1e6 / n_per_wave, (double)h[0] / n_per_wave, flop_per_mfma * n_per_wave * waves / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 768}) {
        run("mfma_f32_16x16x4 ", k16<1>, threads, 1, 2048);
        run("mfma_f32_16x16x4 ", k16<2>, threads, 2, 2048);
        run("mfma_f32_16x16x4 ", k16<3>, threads, 3, 2048);
        run("mfma_f32_16x16x4 ", k16<4>, threads, 4, 2048);
        run("mfma_f32_32x32x2 ", k32<1>, threads, 1, 4096);
        run("mfma_f32_32x32x2 ", k32<2>, threads, 2, 4096);
    }
    return 0;
}
