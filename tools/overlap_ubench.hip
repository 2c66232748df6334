// Micro-benchmark: do MFMA work of one wave and VALU / transcendental work of ANOTHER wave on the same
// SIMD overlap?  512-thread blocks (2 waves per SIMD): waves 0-3 run an MFMA loop, waves 4-7 a VALU loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/overlap_ubench.hip -o tools/bin/overlap_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MK, int VK, int VPRIO>
__global__ __launch_bounds__(512) void kern(float* out, long long* ticks, int mi, int vi, float a, float b) {
    const int wave = threadIdx.x >> 6;
    float s = 0;
    long long t0 = wall_clock64();
    if (wave < 4) {
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        bf16x8 ab, bb;
        for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)a; bb[e] = (__bf16)b; }
        for (int it = 0; it < mi; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MK == 0) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc1, 0, 0, 0);
                }
            }
        }
        s = acc0[0] + acc1[1];
    } else {
        if (VPRIO) __builtin_amdgcn_s_setprio(3);
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + threadIdx.x * 0.001f + i;
        for (int it = 0; it < vi; ++it) {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                if (VK == 0) v[n & 7] = __builtin_fmaf(v[n & 7], b, a);
                else v[n & 7] = __builtin_amdgcn_exp2f(v[n & 7]);
            }
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}

template <int MK, int VK, int VPRIO = 0>
void run(const char* name, int mi, int vi) {
    float* out; long long* ticks;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 64);
    hipLaunchKernelGGL((kern<MK, VK, VPRIO>), dim3(256), dim3(512), 0, 0, out, ticks, mi, vi, 1.0f, 0.5f);
    hipLaunchKernelGGL((kern<MK, VK, VPRIO>), dim3(256), dim3(512), 0, 0, out, ticks, mi, vi, 1.0f, 0.5f);
    (void)hipDeviceSynchronize();
    long long h[8];
    (void)hipMemcpy(h, ticks, 64, hipMemcpyDeviceToHost);
    // wall_clock64 ticks at 100 MHz -> 10 ns
    printf("%-26s mfma iters %5d valu iters %5d : mfma wave %7.1f us, valu wave %7.1f us\n", name, mi, vi, h[0] * 0.01, h[4] * 0.01);
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    // per iteration: 16 MFMAs (fp32: 512 cycles; bf16: 256 cycles) / 16 VALU ops
    run<0, 0>("f32 mfma alone", 4000, 0);
    run<0, 0>("valu fma alone", 0, 16000);
    run<0, 0>("f32 mfma + valu fma", 4000, 16000);
    run<1, 0>("bf16 mfma alone", 8000, 0);
    run<1, 0>("bf16 mfma + valu fma", 8000, 16000);
    run<0, 1>("trans alone", 0, 8000);
    run<0, 1>("f32 mfma + trans", 4000, 8000);
    run<1, 1>("bf16 mfma + trans", 8000, 8000);
    run<0, 0, 1>("f32 mfma + VALU(prio3)", 4000, 16000);
    run<1, 0, 1>("bf16 mfma + VALU(prio3)", 8000, 16000);
    run<1, 0, 1>("bf16 mfma + VALU(prio3) long", 8000, 64000);
    run<0, 1, 1>("f32 mfma + trans(prio3)", 4000, 8000);
    return 0;
}
