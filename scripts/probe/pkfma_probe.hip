// Diagnostic (gfx950): issue rate of v_pk_fma_f32 / v_fma_f32 streams (5 independent accumulator chains), operands from VGPRs or
// from an SGPR pair, 1 or 2 waves per SIMD.  Reported: cycles per instruction per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: v_pk_fma_f32 v, v, v, v    1: v_pk_fma_f32 v, s, v, v    2: v_fma_f32 v, s, v, v    3: v_pk_fma_f32 v, v, v(op_sel bcast), v
// 4: v_pk_mul_f32 v, v, s            5: v_fma_f32 v, v, v, v
template <int KIND, int WAVES>
__global__ __launch_bounds__(256 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void k(float* out, int iters, float seed, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x2 z0 = {seed, seed}, z1 = z0 * 2.0f, z2 = z0 * 3.0f, z3 = z0 * 4.0f, z4 = z0 * 5.0f;
    f32x2 a = {seed * lane, seed}, b = {1.0001f, 0.9999f};
    f32x2 sw = {seed, 1.0f};
    asm volatile("" : "+s"(sw));
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 20; ++r) {
#define STEP(z)                                                                                                        \
    if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(z) : "v"(a), "v"(b));                             \
    if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(z) : "s"(sw), "v"(b));                            \
    if (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z[0]) : "s"(sw[0]), "v"(b[0]));                      \
    if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(z) : "v"(a), "v"(b));           \
    if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(z) : "v"(a), "s"(sw));                                \
    if (KIND == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z[0]) : "v"(a[0]), "v"(b[0]));
            if (KIND == 6) {          // ONE dependent chain of v_fma_f32
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z0[0]) : "v"(a[0]), "v"(b[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z0[0]) : "v"(a[0]), "v"(b[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z0[0]) : "v"(a[0]), "v"(b[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z0[0]) : "v"(a[0]), "v"(b[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z0[0]) : "v"(a[0]), "v"(b[0]));
            } else if (KIND == 7) {   // two chains, alternating
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z0[0]) : "v"(a[0]), "v"(b[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z1[0]) : "v"(a[0]), "v"(b[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z0[0]) : "v"(a[0]), "v"(b[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z1[0]) : "v"(a[0]), "v"(b[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z0[0]) : "v"(a[0]), "v"(b[0]));
            } else if (KIND == 8) {   // v_cmp (vcc) + dependent v_cndmask pairs, adjacent
                asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(z0[0]) : "v"(a[0]), "v"(b[0]) : "vcc");
                asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(z1[0]) : "v"(a[0]), "v"(b[0]) : "vcc");
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(z2[0]) : "v"(a[0]), "v"(b[0]));
            } else if (KIND == 9) {   // the same work, compares into SGPR pairs first, selects later
                unsigned long long m0, m1;
                asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m0) : "v"(a[0]));
                asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m1) : "v"(b[0]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(z2[0]) : "v"(a[0]), "v"(b[0]));
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(z0[0]) : "v"(b[0]), "s"(m0));
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(z1[0]) : "v"(b[0]), "s"(m1));
            } else {
            STEP(z0) STEP(z1) STEP(z2) STEP(z3) STEP(z4)
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t0);
    out[blockIdx.x * 256 * WAVES + threadIdx.x] = z0[0] + z1[1] + z2[0] + z3[1] + z4[0];
}

template <int KIND, int WAVES>
double run(int iters) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k<KIND, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, out, 4, 1e-3f, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL((k<KIND, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, out, iters, 1e-3f, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)c / ((double)iters * 100);
}
template <int KIND> void row(const char* n) { printf("%-48s 1 wave/SIMD %.2f   2 waves/SIMD %.2f\n", n, run<KIND, 1>(2000), run<KIND, 2>(2000)); }
int main() {
    printf("cycles per instruction per wave (5 independent chains)\n");
    row<0>("v_pk_fma_f32 v, v, v, v"); row<1>("v_pk_fma_f32 v, s[pair], v, v"); row<3>("v_pk_fma_f32 v, v, v, v op_sel_hi:[1,0,1]");
    row<2>("v_fma_f32 v, s, v, v"); row<5>("v_fma_f32 v, v, v, v"); row<4>("v_pk_mul_f32 v, v, s[pair]");
    row<6>("v_fma_f32, ONE dependent chain"); row<7>("v_fma_f32, two chains");
    row<8>("[cmp vcc, nop, cndmask] x2 + mul (per 5 slots)"); row<9>("cmp s, cmp s, mul, cnd, cnd (per 5 slots)");
    return 0;
}
