// Diagnostic (gfx950): a width-10 Dense layer with lane = observation on v_mfma_f32_4x4x1_16b_f32 (A = one block of a weight
// register broadcast with CBSZ / ABID, B = the activation register of input feature k, D = four output features of every lane's
// observation).  Reported: cycles per layer of 64 observations, one wave per SIMD.
//   mode 0: forward (33 small MFMAs + LeakyReLU)            mode 1: + backward-like work (30 small MFMAs + 30 vector instructions)
//   mode 2: mode 1 + 16 v_mfma_f32_16x16x4_f32 (the weight gradient)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int W = 10, NC = 3, NLAY = 20;

template <int K>
__device__ __forceinline__ f32x4 step(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, K, 0); }

template <int N, int K = 0>
struct chain {
    static __device__ __forceinline__ void run(const float (&wa)[NC], const float (&h)[W + 1], f32x4 (&acc)[NC]) {
#pragma unroll
        for (int oc = 0; oc < NC; ++oc) acc[oc] = step<K>(wa[oc], h[K], acc[oc]);
        chain<N, K + 1>::run(wa, h, acc);
    }
};
template <int N>
struct chain<N, N> {
    static __device__ __forceinline__ void run(const float (&)[NC], const float (&)[W + 1], f32x4 (&)[NC]) {}
};

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void lanemlp(float* out, int iters, float seed, unsigned long long* cyc) {
    __shared__ float img[2 * NLAY * NC * 64];
    for (int i = threadIdx.x; i < 2 * NLAY * NC * 64; i += 256) img[i] = 0.05f * (float)((i * 7919) % 13 - 6);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float h[W + 1];
#pragma unroll
    for (int i = 0; i < W; ++i) h[i] = seed * (float)(lane + i);
    h[W] = 1.0f;
    f32x4 m0 = {0, 0, 0, 0}, m1 = m0;
    float keep = 0.0f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int l = 0; l < NLAY; ++l) {
            float wa[NC], wb[NC];
#pragma unroll
            for (int oc = 0; oc < NC; ++oc) { wa[oc] = img[(l * NC + oc) * 64 + lane]; wb[oc] = img[((NLAY + l) * NC + oc) * 64 + lane]; }
            f32x4 acc[NC];
#pragma unroll
            for (int oc = 0; oc < NC; ++oc) acc[oc] = f32x4{0, 0, 0, 0};
            chain<W + 1>::run(wa, h, acc);
            float hn[W + 1];
#pragma unroll
            for (int f = 0; f < W; ++f) {
                const float z = acc[f >> 2][f & 3];
                hn[f] = __builtin_fmaxf(z, 0.01f * z);
            }
            hn[W] = 1.0f;
            if (MODE >= 1) {
                f32x4 dh[NC];
#pragma unroll
                for (int oc = 0; oc < NC; ++oc) dh[oc] = f32x4{0, 0, 0, 0};
                float dz[W + 1];
#pragma unroll
                for (int f = 0; f < W; ++f) dz[f] = (h[f] > 0.0f) ? hn[f] : 0.01f * hn[f];
                dz[W] = 0.0f;
                chain<W>::run(wb, dz, dh);
#pragma unroll
                for (int f = 0; f < W; ++f) hn[f] += 1e-3f * dh[f >> 2][f & 3];
            }
            if (MODE >= 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hn[k % W], hn[(k + 1) % W], m0, 0, 0, 0);
                    m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hn[(k + 2) % W], hn[(k + 3) % W], m1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int f = 0; f <= W; ++f) h[f] = hn[f];
        }
        keep += h[0];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t0);
    float s = keep + m0[0] + m1[1];
#pragma unroll
    for (int i = 0; i < W; ++i) s += h[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
double run(int iters) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((lanemlp<MODE>), dim3(256), dim3(256), 0, 0, out, 4, 1e-3f, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL((lanemlp<MODE>), dim3(256), dim3(256), 0, 0, out, iters, 1e-3f, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)c / ((double)iters * NLAY);
}
int main() {
    printf("cycles per width-10 layer of 64 observations, one wave per SIMD (16x16x4 form of today: ~2040 for forward + backward + weight gradient)\n");
    printf("forward (33 x 4x4x1 + LeakyReLU)                        %.0f\n", run<0>(500));
    printf("+ backward-like (30 x 4x4x1 + 40 vector instructions)   %.0f\n", run<1>(500));
    printf("+ weight gradient (16 x 16x16x4)                        %.0f\n", run<2>(500));
    return 0;
}
