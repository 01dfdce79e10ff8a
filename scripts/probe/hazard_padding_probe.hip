// What hipcc 7.2 pads by itself around an MFMA on gfx950, and what it does not (round 6, NOTEBOOK R6.1).  Compile to assembly and read:
//   hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S hazard_padding_probe.hip -o - | grep -v '^\s*[.;]'
// k1, k4, k5: a compiler-known vector-ALU write in front of an MFMA that reads the register as B or C -> `s_nop 1` (two wait states), and
//             `s_nop 9` / `s_nop 3` behind the MFMA before its result is read;
// k2, k3:     the same pairs with ONE end inside an `asm` statement -> nothing (the inline-assembly v_max_f32 / v_cmp directly behind the MFMA).
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k1(float* p, float s) {
    float a = p[threadIdx.x], b = p[threadIdx.x + 64];
    f32x4 c = {p[1], p[2], p[3], p[4]};
    float a2 = a * s;             // VALU write -> MFMA srcA
    float b2 = b + s;
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, c, 0, 0, 0);
    float r = c[0] * s + c[1];
    p[threadIdx.x] = r;
}
__global__ void k2(float* p, float s) {
    float a = p[threadIdx.x], b = p[threadIdx.x + 64];
    f32x4 c = {p[1], p[2], p[3], p[4]};
    c = c * s;                   // VALU write -> MFMA srcC
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 3, 0);
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(c[0]), "v"(c[1]));
    p[threadIdx.x] = m;
}
__global__ void k3(float* p, float s) {
    float a = p[threadIdx.x], b = p[threadIdx.x + 64];
    f32x4 c = {p[1], p[2], p[3], p[4]};
    unsigned long long m;
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 3, 0);
    asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m) : "v"(c[2]));
    float r;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    p[threadIdx.x] = r;
}
__global__ void k4(f32x4* p, float* q, float s) {
    f32x4 c = p[threadIdx.x];
    float a = q[threadIdx.x], b = q[threadIdx.x + 64];
    __builtin_amdgcn_sched_barrier(0);
    float b2 = b * s;             // VALU write -> MFMA srcB
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b2, c, 4, 3, 0);
    __builtin_amdgcn_sched_barrier(0);
    p[threadIdx.x] = c;
}
__global__ void k5(f32x4* p, float* q, float s) {
    f32x4 c = p[threadIdx.x];
    float a = q[threadIdx.x], b = q[threadIdx.x + 64];
    __builtin_amdgcn_sched_barrier(0);
    float b2 = b * s;             // VALU write -> MFMA srcB
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b2, c, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    p[threadIdx.x] = c;
}
