// Diagnostic (gfx950): what does ONE wave pay for vector / LDS instructions placed between its own v_mfma_f32_16x16x4_f32?
// A wave runs 4 independent MFMA chains; between two consecutive MFMAs it issues K instructions of one kind on registers that
// nothing else touches.  Reported: cycles per MFMA (32 = the matrix pipe's own rate).  WAVES = 1 or 2 waves per SIMD running the
// same stream.  Usage: ./coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

// kinds: 0 v_mul_f32, 1 v_max_f32, 2 v_cmp_lt_f32 + v_cndmask (counts as 2), 3 v_ashrrev_i32, 4 v_bfi_b32, 5 v_and_b32, 6 v_mov_b32,
//        7 ds_write_b32, 8 ds_read_b128, 9 v_fma_f32, 10 v_cndmask only (vcc fixed), 11 v_cmp_gt_i32 only, 12 v_xor_b32, 13 s_nop 0,
//        14 v_mul + v_ashr + v_bfi (leaky select in 3), 15 v_lshlrev_b32, 16 v_add_f32, 17 v_exp_f32
template <int KIND>
__device__ __forceinline__ void filler(float& f0, float& f1, float& f2, unsigned& u0, unsigned& u1, float* lds, f32x4& l0, int lane) {
    if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
    if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
    if (KIND == 2) asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(f0) : "v"(f1), "v"(f2) : "vcc");
    if (KIND == 3) asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(u0) : "v"(f1));
    if (KIND == 4) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(f0) : "v"(u1), "v"(f2));
    if (KIND == 5) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u0) : "v"(u1));
    if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(f0) : "v"(f1));
    if (KIND == 7) asm volatile("ds_write_b32 %0, %1" ::"v"(4 * lane), "v"(f1) : "memory");
    if (KIND == 8) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(3)" : "=v"(l0) : "v"(16 * lane) : "memory");
    if (KIND == 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2));
    if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f0) : "v"(f2) : );
    if (KIND == 11) asm volatile("v_cmp_gt_i32 vcc, %0, %1" ::"v"(u0), "v"(u1) : "vcc");
    if (KIND == 12) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u0) : "v"(u1));
    if (KIND == 13) asm volatile("s_nop 0");
    if (KIND == 14) asm volatile("v_mul_f32 %0, %2, %3\n\tv_ashrrev_i32 %1, 31, %4\n\tv_bfi_b32 %0, %1, %0, %3" : "=&v"(f0), "=&v"(u0) : "v"(f1), "v"(f2), "v"(u1));
    if (KIND == 15) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(u0) : "v"(u1));
    if (KIND == 16) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
    if (KIND == 17) asm volatile("v_exp_f32 %0, %1" : "=v"(f0) : "v"(f1));
}

template <int KIND, int K, int WAVES>
__global__ __launch_bounds__(256 * WAVES) void mix(float* out, int iters, float seed, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 256 * WAVES) lds[i] = seed * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, l0 = a0;
    float x = seed * lane, y = seed;
    float f0 = seed, f1 = 1.0001f, f2 = 0.5f;
    unsigned u0 = lane, u1 = lane * 3 + 1;
    asm volatile("v_cmp_gt_i32 vcc, %0, %1" ::"v"(u0), "v"(u1) : "vcc");
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            a0 = MF(x, y, a0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) filler<KIND>(f0, f1, f2, u0, u1, lds, l0, lane);
            __builtin_amdgcn_sched_barrier(0);
            a1 = MF(x, y, a1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) filler<KIND>(f0, f1, f2, u0, u1, lds, l0, lane);
            __builtin_amdgcn_sched_barrier(0);
            a2 = MF(x, y, a2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) filler<KIND>(f0, f1, f2, u0, u1, lds, l0, lane);
            __builtin_amdgcn_sched_barrier(0);
            a3 = MF(x, y, a3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) filler<KIND>(f0, f1, f2, u0, u1, lds, l0, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t0);      // the slowest wave of the workgroup
    out[blockIdx.x * 256 * WAVES + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + (float)u0 + l0[0];
}

template <int KIND, int K, int WAVES>
double run(int iters) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((mix<KIND, K, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, out, 10, 1e-3f, cyc);
    hipDeviceSynchronize();
    hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL((mix<KIND, K, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, out, iters, 1e-3f, cyc);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    hipFree(out); hipFree(cyc);
    return (double)c / ((double)iters * 32);
}

template <int KIND>
void row(const char* name) {
    const int it = 4000;
    printf("%-28s 1 wave/SIMD: K=0 %.1f  K=1 %.1f  K=2 %.1f  K=3 %.1f  K=4 %.1f  K=6 %.1f | 2 waves/SIMD (per wave): K=1 %.1f  K=2 %.1f  K=4 %.1f  K=6 %.1f\n", name,
           run<KIND, 0, 1>(it), run<KIND, 1, 1>(it), run<KIND, 2, 1>(it), run<KIND, 3, 1>(it), run<KIND, 4, 1>(it), run<KIND, 6, 1>(it),
           run<KIND, 1, 2>(it), run<KIND, 2, 2>(it), run<KIND, 4, 2>(it), run<KIND, 6, 2>(it));
}

int main() {
    printf("cycles per MFMA of the SLOWEST wave of a workgroup (4 chains per wave); K fillers of one kind after EVERY MFMA; with 2 waves/SIMD the pipe's floor is 64 per wave\n");
    printf("%-28s 2 waves/SIMD, no fillers: %.1f\n", "(baseline)", run<13, 0, 2>(4000));
    row<0>("v_mul_f32"); row<1>("v_max_f32"); row<16>("v_add_f32"); row<9>("v_fma_f32"); row<2>("v_cmp_lt_f32+v_cndmask (x2)"); row<10>("v_cndmask_b32");
    row<11>("v_cmp_gt_i32"); row<3>("v_ashrrev_i32"); row<15>("v_lshlrev_b32"); row<4>("v_bfi_b32"); row<5>("v_and_b32"); row<12>("v_xor_b32"); row<6>("v_mov_b32");
    row<14>("mul+ashr+bfi (x3)"); row<17>("v_exp_f32"); row<13>("s_nop 0"); row<7>("ds_write_b32"); row<8>("ds_read_b128");
    return 0;
}
