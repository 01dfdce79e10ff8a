// Diagnostic: shader clock under different instruction mixes: cycles of s_memtime per 100 MHz s_memrealtime tick.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed, unsigned long long* res) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = seed * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
    float x = seed * lane, y = seed, f0 = seed, f1 = 2 * seed, f2 = 3 * seed, f3 = 4 * seed;
    f32x4 l0 = a0;
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (KIND >= 1) { a0 = MF(x, y, a0); a1 = MF(x, y, a1); }
            if (KIND >= 2) { l0 += *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + s * 256 + it * 64) & 8191)); }
            if (KIND >= 3) { f0 = fmaf(f0, 1.0001f, 0.5f); f1 = fmaf(f1, 1.0001f, 0.5f); f2 = fmaf(f2, 1.0001f, 0.5f); f3 = fmaf(f3, 1.0001f, 0.5f); }
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { res[0] = c1 - c0; res[1] = r1 - r0; }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + l0[0] + l0[2] + f0 + f1 + f2 + f3;
}
template <int KIND> void run(const char* name) {
    float* out; unsigned long long* res;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&res, 16);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, out, 100, 1e-3f, res); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, out, 200000, 1e-3f, res);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long r[2]; hipMemcpy(r, res, 16, hipMemcpyDeviceToHost);
    printf("%-28s wall %.2f ms  s_memtime %.1f M  s_memrealtime %.1f M ticks  -> memtime/wall = %.1f MHz, realtime/wall = %.1f MHz, cycles per iteration %.1f\n", name, ms,
           r[0] / 1e6, r[1] / 1e6, r[0] / (ms * 1e3), r[1] / (ms * 1e3), (double)r[0] / 200000);
}
int main() { run<0>("empty loop"); run<1>("mfma"); run<2>("mfma + lds"); run<3>("mfma + lds + valu"); return 0; }
