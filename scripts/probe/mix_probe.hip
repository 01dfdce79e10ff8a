// Diagnostic: does VALU work of ONE wave slow down the MFMAs of the OTHER wave on the same SIMD?  (gfx950)
// Workgroup = 512 threads: waves 0-3 (one per SIMD) run 4 independent v_mfma_f32_16x16x4_f32 chains; waves 4-7 run `KIND` work:
//   0 nothing, 1 v_fma_f32 chains, 2 v_pk_fma_f32 chains, 3 v_add_u32 chains, 4 v_mov/xor (logic), 5 ds_read_b128, 6 v_mul_lo_u32 (quarter rate)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int KIND>
__global__ __launch_bounds__(512) void mix(float* out, int iters, float seed, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = seed * i;
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float r = 0;
    if (wv < 4) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = seed * lane, y = seed;
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) { a0 = MF(x, y, a0); a1 = MF(x, y, a1); a2 = MF(x, y, a2); a3 = MF(x, y, a3); }
        }
        unsigned long long t1 = __builtin_readcyclecounter();
        r = a0[0] + a1[1] + a2[2] + a3[3];
        if (lane == 0 && blockIdx.x == 0 && wv == 0) cyc[0] = t1 - t0;
    } else if (KIND != 0) {
        float f0 = seed, f1 = seed * 2, f2 = seed * 3, f3 = seed * 4, f4 = seed * 5, f5 = seed * 6, f6 = seed * 7, f7 = seed * 8;
        f32x2 p0 = {seed, seed}, p1 = p0 * 2.f, p2 = p0 * 3.f, p3 = p0 * 4.f;
        unsigned u0 = lane, u1 = lane * 3, u2 = lane * 5, u3 = lane * 7;
        f32x4 l0 = {0, 0, 0, 0};
        // the side work runs ~ as long as the MFMA loop: 32 MFMAs x 32 cycles = 1024 cycles per iteration
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (KIND == 1) { f0 = fmaf(f0, 1.0001f, 0.5f); f1 = fmaf(f1, 1.0001f, 0.5f); f2 = fmaf(f2, 1.0001f, 0.5f); f3 = fmaf(f3, 1.0001f, 0.5f);
                                 f4 = fmaf(f4, 1.0001f, 0.5f); f5 = fmaf(f5, 1.0001f, 0.5f); f6 = fmaf(f6, 1.0001f, 0.5f); f7 = fmaf(f7, 1.0001f, 0.5f); }
                if (KIND == 2) { p0 = p0 * 1.0001f + 0.5f; p1 = p1 * 1.0001f + 0.5f; p2 = p2 * 1.0001f + 0.5f; p3 = p3 * 1.0001f + 0.5f;
                                 p0 = p0 * 1.0002f + 0.25f; p1 = p1 * 1.0002f + 0.25f; p2 = p2 * 1.0002f + 0.25f; p3 = p3 * 1.0002f + 0.25f; }
                if (KIND == 3) { u0 += u1; u1 += u2; u2 += u3; u3 += u0; u0 += 3; u1 += 5; u2 += 7; u3 += 11; }
                if (KIND == 4) { u0 ^= u1; u1 ^= u2; u2 ^= u3; u3 ^= u0; u0 = ~u0; u1 = ~u1; u2 = ~u2; u3 = ~u3; }
                if (KIND == 5) { l0 += *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + s * 256 + (u0 & 3) * 4) & 4095)); u0++; }
                if (KIND == 6) { u0 *= u1; u1 *= 2654435761u; }
            }
        }
        r = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0[0] + p1[1] + p2[0] + p3[1] + (float)(u0 + u1 + u2 + u3) + l0[0] + l0[3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, int iters) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((mix<KIND>), dim3(256), dim3(512), 0, 0, out, 10, 1e-3f, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<KIND>), dim3(256), dim3(512), 0, 0, out, iters, 1e-3f, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)c / ((double)iters * 32);
    printf("%-34s : MFMA wave sees %.1f cycles per MFMA (32 = full rate); kernel wall %.3f ms = %.0f cycles (2.4 GHz) per iteration [MFMA alone = 1024]\n", name, per, ms, ms * 1e-3 * 2.4e9 / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 20000;
    run<0>("partner idle", it); run<1>("partner v_fma_f32 x8/step", it); run<2>("partner v_pk_fma_f32 x8/step", it);
    run<3>("partner v_add_u32 x8/step", it); run<4>("partner v_xor/v_not x8/step", it); run<5>("partner ds_read_b128 x1/step", it);
    run<6>("partner v_mul_lo_u32 x2/step", it);
    return 0;
}
