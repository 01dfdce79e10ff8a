// Diagnostic (gfx950): can the vector unit run a width-10 Dense layer faster than the matrix unit runs its zero-padded 16 x 16 image?
// lane = observation, activations in registers, weights as wave-uniform SGPR pairs read with scalar loads (chunks of CH rows, one
// chunk ahead), v_pk_fma_f32 on output pairs.  Reported: cycles per layer of 64 observations (the 16x16x4 fp32 MFMA form needs
// 12 MFMAs = 384 cycles for the same products, plus 30 LeakyReLU instructions beside them).
//   mode 0: forward chain only (55 v_pk_fma + bias moves + LeakyReLU per layer)
//   mode 1: + 16 MFMAs per layer in two chains (the weight gradient of the lane = observation design), as one block per layer
//   mode 2: the 16 MFMAs per layer alone
// Usage: ./vecmlp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int W = 10, NLAY = 20, HP = W / 2;
constexpr int ROWS = W + 1;                 // per layer: bias row, then one row per input feature; a row = HP output pairs
constexpr int IMG = ROWS * W;
#ifndef CHROWS
#define CHROWS 3
#endif
constexpr int CH = CHROWS;                  // rows per chunk (10 SGPRs per row), one chunk in flight while one is consumed
constexpr int NCH = (ROWS + CH - 1) / CH;

struct chunk { f32x2 w[CH][HP]; };

typedef const __attribute__((address_space(4))) f32x2* cptr;      // constant address space: uniform loads become s_load
__device__ __forceinline__ cptr opaque(const f32x2* g) {
    cptr p = (cptr)g;
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ void load_chunk(chunk& c, const f32x2* __restrict__ img, int k) {     // k: chunk number in the stream of all layers
    const int l = k / NCH, r0 = (k % NCH) * CH;
    cptr p = opaque(img) + (l * ROWS + r0) * HP;      // (the offset is an immediate of the load)
#pragma unroll
    for (int r = 0; r < CH; ++r)
#pragma unroll
        for (int q = 0; q < HP; ++q)
            if (r0 + r < ROWS) c.w[r][q] = p[r * HP + q];
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(256 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void vecmlp(const float* __restrict__ wimg, float* out, int iters, float seed, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    const f32x2* __restrict__ img = reinterpret_cast<const f32x2*>(wimg);
    float h[W];
#pragma unroll
    for (int i = 0; i < W; ++i) h[i] = seed * (float)(lane + i);
    f32x4 m0 = {0, 0, 0, 0}, m1 = m0;
    float keep = 0.0f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        chunk cb[2];
        if (MODE != 2) load_chunk(cb[0], img, 0);
#pragma unroll
        for (int l = 0; l < NLAY; ++l) {
            if (MODE != 2) {
                f32x2 z[HP];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int k = l * NCH + c;
                    if (k + 1 < NLAY * NCH) load_chunk(cb[(k + 1) & 1], img, k + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    const chunk& cc = cb[k & 1];
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        const int row = c * CH + r;
                        if (row >= ROWS) continue;
                        if (row == 0) {
#pragma unroll
                            for (int q = 0; q < HP; ++q) z[q] = cc.w[r][q];
                        } else {
                            const f32x2 hh = {h[row - 1], h[row - 1]};
#pragma unroll
                            for (int q = 0; q < HP; ++q) z[q] = __builtin_elementwise_fma(cc.w[r][q], hh, z[q]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < HP; ++q) {
                    const f32x2 t = z[q] * 0.01f;
                    h[2 * q] = __builtin_fmaxf(z[q][0], t[0]);
                    h[2 * q + 1] = __builtin_fmaxf(z[q][1], t[1]);
                }
            }
            if (MODE >= 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(h[k % W], h[(k + 1) % W], m0, 0, 0, 0);
                    m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(h[(k + 2) % W], h[(k + 3) % W], m1, 0, 0, 0);
                }
            }
        }
        keep += h[0];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t0);
    float s = keep + m0[0] + m1[1];
#pragma unroll
    for (int i = 0; i < W; ++i) s += h[i];
    out[blockIdx.x * 256 * WAVES + threadIdx.x] = s;
}

template <int MODE, int WAVES>
double run(const float* wimg, int iters) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((vecmlp<MODE, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, wimg, out, 4, 1e-3f, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL((vecmlp<MODE, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, wimg, out, iters, 1e-3f, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)c / ((double)iters * NLAY);
}

int main() {
    float* wimg; (void)hipMalloc(&wimg, NLAY * IMG * 4);
    static float hw[NLAY * IMG];
    for (int i = 0; i < NLAY * IMG; ++i) hw[i] = 0.05f * (float)((i * 7919) % 13 - 6);
    (void)hipMemcpy(wimg, hw, sizeof hw, hipMemcpyHostToDevice);
    printf("cycles per width-10 layer of 64 observations (matrix form: 12 MFMAs = 384 cycles + 30 LeakyReLU instructions)\n");
    printf("forward only            : 1 wave/SIMD %.0f   2 waves/SIMD (per wave) %.0f\n", run<0, 1>(wimg, 500), run<0, 2>(wimg, 500));
    printf("forward + 16 MFMA/layer : 1 wave/SIMD %.0f   2 waves/SIMD (per wave) %.0f\n", run<1, 1>(wimg, 500), run<1, 2>(wimg, 500));
    printf("16 MFMA/layer alone     : 1 wave/SIMD %.0f   2 waves/SIMD (per wave) %.0f\n", run<2, 1>(wimg, 500), run<2, 2>(wimg, 500));
    return 0;
}
