// Diagnostic (gfx950): issue interval of v_mfma_f32_4x4x1_16b_f32 as a function of the number of independent accumulator chains
// (1 .. 8) a wave interleaves, one or two waves per SIMD.  Reported: cycles per MFMA per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH, int WAVES>
__global__ __launch_bounds__(256 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k(float* out, int iters, float seed, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{seed * c, 0, 0, 0};
    float a = seed * lane, b = 1.0f + seed;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 4, 3, 0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t0);
    float s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * 256 * WAVES + threadIdx.x] = s;
}
template <int CH, int WAVES>
double run(int iters) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k<CH, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, out, 4, 1e-3f, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL((k<CH, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, out, iters, 1e-3f, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)c / ((double)iters * 16 * CH);
}
template <int CH> void row() { printf("chains %d: 1 wave/SIMD %.1f   2 waves/SIMD %.1f (per wave)\n", CH, run<CH, 1>(2000), run<CH, 2>(2000)); }
int main() {
    printf("cycles per v_mfma_f32_4x4x1_16b_f32 per wave (2 passes = 8 cycles of the matrix pipe)\n");
    row<1>(); row<2>(); row<3>(); row<4>(); row<6>(); row<8>();
    return 0;
}
