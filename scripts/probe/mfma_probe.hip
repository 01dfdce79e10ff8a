// Diagnostic micro-benchmark (not part of the library): issue rate of v_mfma_f32_16x16x4_f32 on gfx950 under the operand /
// dependency patterns of the fused scaler kernel.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

// VARIANT 0: CH independent chains, register operands
// VARIANT 1: A operands from LDS (ds_read_b128 per 4 steps), like the forward pass
// VARIANT 2: variant 1 + LeakyReLU on the accumulators after every 32 MFMAs (reset chain), like a layer seam
template <int CH, int VARIANT>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 68; i += blockDim.x) lds[i] = seed * (float)(i & 7);
    __syncthreads();
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float b[4] = {seed, seed * 2, seed * 3, seed * 4};
    float a = seed * lane;
    const int j = lane & 15, q = lane >> 4;
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 0) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] = MF(a, b[s & 3], acc[c]);
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                f32x4 av[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) av[c] = *reinterpret_cast<const f32x4*>(lds + (16 * c + j) * 68 + 16 * kb + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int c = 0; c < CH; ++c) acc[c] = MF(av[c][t], b[t], acc[c]);
            }
            asm volatile("" ::: "memory");
            if (VARIANT == 2) {
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int t = 0; t < 4; ++t) { float v = acc[c][t]; v = fmaxf(v, 0.01f * v); b[t] = v * 1e-3f + b[t] * 0.5f; acc[c][t] = seed; }
            }
        }
    }
    float r = 0;
    for (int c = 0; c < CH; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + b[0];
}

// VARIANT 3 of the seam: the LeakyReLU of block n is issued in the shadow of block n+1's MFMAs (software pipelined inside the wave)
template <int CH>
__global__ __launch_bounds__(512) void probe_pipe(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 68; i += blockDim.x) lds[i] = seed * (float)(i & 7);
    __syncthreads();
    f32x4 acc[CH], prev[CH];
    for (int c = 0; c < CH; ++c) { acc[c] = f32x4{0, 0, 0, 0}; prev[c] = f32x4{0, 0, 0, 0}; }
    float b[4] = {seed, seed * 2, seed * 3, seed * 4};
    float bn[4] = {seed, seed * 2, seed * 3, seed * 4};
    const int j = lane & 15, q = lane >> 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 av[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) av[c] = *reinterpret_cast<const f32x4*>(lds + (16 * c + j) * 68 + 16 * kb + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] = MF(av[c][t], b[t], acc[c]);
        }
        // seam work of the PREVIOUS block, independent of this block's MFMAs
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) { float v = prev[c][t]; v = fmaxf(v, 0.01f * v); bn[t] = v * 1e-3f + bn[t] * 0.5f; }
#pragma unroll
        for (int g = 0; g < 8 * CH; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // 2 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // 2 VALU
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < CH; ++c) { prev[c] = acc[c]; acc[c] = f32x4{seed, seed, seed, seed}; }
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = bn[t];
    }
    float r = 0;
    for (int c = 0; c < CH; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3] + prev[c][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + b[0];
}

template <int CH>
void run_pipe(const char* name, int threads, int iters) {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sm = 64 * 68 * 4;
    hipLaunchKernelGGL((probe_pipe<CH>), dim3(256), dim3(threads), sm, 0, out, 10, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_pipe<CH>), dim3(256), dim3(threads), sm, 0, out, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)256 * (threads / 64) * iters * 16.0 * CH;
    const double tf = mf * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-44s threads %3d chains %d : %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, threads, CH, ms, tf, 100 * tf / 157.3);
    hipFree(out);
}

template <int CH, int VARIANT>
void run(const char* name, int threads, int iters) {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sm = 64 * 68 * 4;
    hipLaunchKernelGGL((probe<CH, VARIANT>), dim3(256), dim3(threads), sm, 0, out, 10, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<CH, VARIANT>), dim3(256), dim3(threads), sm, 0, out, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)256 * (threads / 64) * iters * 16.0 * CH;
    const double tf = mf * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-44s threads %3d chains %d : %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)\n", name, threads, CH, ms, tf, 100 * tf / 157.3);
    hipFree(out);
}

int main() {
    const int it = 20000;
    run<1, 0>("reg operands", 256, it);  run<2, 0>("reg operands", 256, it);  run<4, 0>("reg operands", 256, it);
    run<1, 0>("reg operands", 512, it);  run<2, 0>("reg operands", 512, it);  run<4, 0>("reg operands", 512, it);
    run<2, 1>("A from LDS b128", 256, it); run<2, 1>("A from LDS b128", 512, it); run<4, 1>("A from LDS b128", 512, it);
    run<2, 2>("A from LDS + lrelu seam / 32 MFMA", 256, it); run<2, 2>("A from LDS + lrelu seam / 32 MFMA", 512, it);
    run<4, 2>("A from LDS + lrelu seam / 64 MFMA", 256, it); run<4, 2>("A from LDS + lrelu seam / 64 MFMA", 512, it);
    run_pipe<2>("pipelined seam (lrelu under next MFMAs)", 256, it); run_pipe<2>("pipelined seam (lrelu under next MFMAs)", 512, it);
    run_pipe<4>("pipelined seam (lrelu under next MFMAs)", 256, it); run_pipe<4>("pipelined seam (lrelu under next MFMAs)", 512, it);
    return 0;
}
