// Diagnostic: forward-pass-like MFMA stream: 5 "layers" x 2 block pairs x 4 k-groups x 8 MFMAs, A from LDS (b128), B from 16 registers
// (the previous layer's outputs), bias from LDS; optional LeakyReLU.  Compare with the real kernel's forward-only launch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
template <int LRELU, int NL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int PW = 68;
    for (int i = threadIdx.x; i < NL * 64 * PW + 512; i += 512) lds[i] = seed * (float)((i * 7) & 15) - seed * 7;
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 15, q = lane >> 4;
    f32x4 h[4];
    for (int b = 0; b < 4; ++b) h[b] = f32x4{seed * lane, seed, -seed, seed * b};
    float r = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float* W = lds + l * 64 * PW;
            f32x4 hn[4];
#pragma unroll
            for (int mb = 0; mb < 4; mb += 2) {
                f32x4 acc0 = *reinterpret_cast<const f32x4*>(lds + NL * 64 * PW + 16 * mb + 4 * q);
                f32x4 acc1 = *reinterpret_cast<const f32x4*>(lds + NL * 64 * PW + 16 * (mb + 1) + 4 * q);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(W + (16 * mb + j) * PW + 16 * kb + 4 * q);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(W + (16 * (mb + 1) + j) * PW + 16 * kb + 4 * q);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { acc0 = MF(a0[t], h[kb][t], acc0); acc1 = MF(a1[t], h[kb][t], acc1); }
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    hn[mb][t] = LRELU ? fmaxf(acc0[t], 0.01f * acc0[t]) : acc0[t];
                    hn[mb + 1][t] = LRELU ? fmaxf(acc1[t], 0.01f * acc1[t]) : acc1[t];
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) h[b] = hn[b];
        }
        r += h[0][0];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r + h[1][1] + h[2][2] + h[3][3];
}
template <int LRELU, int NL> void run(const char* name, int iters) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const size_t sm = (NL * 64 * 68 + 512) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<LRELU, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    hipLaunchKernelGGL((k<LRELU, NL>), dim3(256), dim3(512), sm, 0, out, 10, 1e-3f); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<LRELU, NL>), dim3(256), dim3(512), sm, 0, out, iters, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 256.0 * 8 * iters * NL * 64;
    const double tf = mf * 2048 / (ms * 1e-3) / 1e12;
    printf("%-40s %8.3f ms  %6.1f TFLOP/s (%.1f%% of 157.3)\n", name, ms, tf, 100 * tf / 157.3);
}
int main() {
    run<0, 4>("4 layers 64x64, no activation", 2000); run<1, 4>("4 layers 64x64, LeakyReLU", 2000);
    run<0, 1>("1 layer in LDS (same weights), no act", 8000); run<1, 1>("1 layer in LDS, LeakyReLU", 8000);
    return 0;
}
