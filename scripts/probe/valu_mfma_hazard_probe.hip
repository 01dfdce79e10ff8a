// Round 6 (NOTEBOOK R6.1): does gfx950 really need wait states between a vector-ALU write of a register and an MFMA that reads it?
//
// hipcc pads two wait states there when it knows both instructions (hazard_padding_probe.hip); an inline-assembly producer or consumer
// gets nothing.  This program issues the pair itself with 0, 1 and 2 wait states of several kinds in between and counts how often the
// MFMA saw the OLD content of the register: v_max_f32 writes b (a fresh value every iteration), v_mfma_f32_4x4x1 with A = 1 copies
// every lane's b into D, D is compared with what v_max_f32 must have written.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_mfma_hazard_probe valu_mfma_hazard_probe.hip && ./valu_mfma_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PAIR(GAP)                                                                                              \
    asm volatile("v_max_f32 %[b], %[x], %[y]\n\t" GAP "v_mfma_f32_4x4x1_16b_f32 %[d], %[a], %[b], 0 cbsz:4\n\t" \
                 "s_nop 7"                                                                                      \
                 : [d] "=&v"(d), [b] "=&v"(b), [t] "+v"(tp), [u] "+v"(tu)                                      \
                 : [x] "v"(x), [y] "v"(y), [a] "v"(one))
#define PAIR16(GAP)                                                                                           \
    asm volatile("v_max_f32 %[b], %[x], %[y]\n\t" GAP "v_mfma_f32_16x16x4_f32 %[d], %[a], %[b], 0\n\t"        \
                 "s_nop 15"                                                                                     \
                 : [d] "=&v"(d), [b] "=&v"(b), [t] "+v"(tp), [u] "+v"(tu)                                      \
                 : [x] "v"(x), [y] "v"(y), [a] "v"(one))

// mode: what stands between the two; pre: what runs just before the pair (0 nothing, 1 four vector-ALU instructions, 2 an MFMA,
// 3 a long scalar wait -> both pipes idle)
template <int MODE, int PRE>
__global__ void probe(unsigned* bad, unsigned* total, int iters, float* sink, const float* lds_src) {
    __shared__ float sm[256];
    sm[threadIdx.x] = lds_src[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float one = 1.0f;
    unsigned nb = 0;
    f32x4 d;
    float b = -1.0f;
    float acc = 0.0f;
    f32x4 t2 = {0.f, 0.f, 0.f, 0.f};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 tp = {1.0f, 1.0f};
    float tu = 1.0f;
    for (int i = 0; i < iters; ++i) {
        const float x = (float)(i * 64 + lane) * 0.25f + 1.0f, y = 0.5f * x;       // max = x: changes every iteration
        if (PRE == 1) {
            float p0 = x, p1 = y;
            asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %0\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %1, %1, %0" : "+v"(p0), "+v"(p1));
            acc += p0 + p1;
        } else if (PRE == 2) {
            asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %1, %0 cbsz:4" : "+v"(t2) : "v"(one));
        } else if (PRE == 3) {
            asm volatile("s_nop 15\n\ts_nop 15");
        }
        if (MODE == 0) PAIR("");
        else if (MODE == 1) PAIR("s_nop 0\n\t");
        else if (MODE == 2) PAIR("s_nop 1\n\t");
        else if (MODE == 3) PAIR("v_pk_mul_f32 %[t], %[t], %[t]\n\t");             // (reads t2 only: an unrelated vector-ALU instruction)
        else if (MODE == 4) PAIR("s_waitcnt lgkmcnt(0)\n\t");
        else if (MODE == 5) PAIR("v_mov_b32 %[u], %[u]\n\t");
        else if (MODE == 6) PAIR("s_setprio 0\n\t");
        else if (MODE == 7) PAIR16("");
        else if (MODE == 8) PAIR16("s_nop 0\n\t");
        else if (MODE == 9) PAIR16("s_nop 1\n\t");
        else if (MODE == 10) PAIR16("v_pk_mul_f32 %[t], %[t], %[t]\n\t");
        const float want = x;
        float got = d[0];
        if (MODE >= 7) {
            // 16x16x4: D[i][j] = sum_k A[i][k] B[k][j]; A = 1: D[.][j] = sum over the four k of B[k][j] = b of lanes j, j+16, j+32, j+48
            const float s = __shfl(want, lane & 15) + __shfl(want, (lane & 15) + 16) + __shfl(want, (lane & 15) + 32) + __shfl(want, (lane & 15) + 48);
            if (got != s) ++nb;
        } else if (got != want) ++nb;
        acc += b;
    }
    atomicAdd(bad, nb);
    atomicAdd(total, (unsigned)iters);
    if (acc == 12345.678f) sink[0] = acc + t2[0] + tp[0] + tu;
}

template <int MODE, int PRE>
static void run(const char* name, unsigned* dbad, unsigned* dtot, float* sink, float* src) {
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(dbad, 0, 4);
        hipMemset(dtot, 0, 4);
        hipLaunchKernelGGL((probe<MODE, PRE>), dim3(1024), dim3(256), 0, 0, dbad, dtot, 2000, sink, src);
        hipDeviceSynchronize();
        unsigned bad = 0, tot = 0;
        hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
        hipMemcpy(&tot, dtot, 4, hipMemcpyDeviceToHost);
        printf("%-46s pre=%d  stale %10u of %10u lane-iterations (%.4f %%)\n", name, PRE, bad, tot, 100.0 * bad / (double)tot);
    }
}

#define RUNALL(MODE, NAME) run<MODE, 0>(NAME, dbad, dtot, sink, src); run<MODE, 1>(NAME, dbad, dtot, sink, src); run<MODE, 2>(NAME, dbad, dtot, sink, src); run<MODE, 3>(NAME, dbad, dtot, sink, src);

int main() {
    unsigned *dbad, *dtot;
    float *sink, *src;
    hipMalloc(&dbad, 4); hipMalloc(&dtot, 4); hipMalloc(&sink, 4); hipMalloc(&src, 1024);
    hipMemset(src, 0, 1024);
    RUNALL(0, "4x4x1: v_max -> mfma, nothing between");
    RUNALL(1, "4x4x1: s_nop 0 between (1 wait state)");
    RUNALL(2, "4x4x1: s_nop 1 between (2 wait states)");
    RUNALL(3, "4x4x1: v_pk_mul_f32 between (1, vector ALU)");
    RUNALL(4, "4x4x1: s_waitcnt between (1, scalar)");
    RUNALL(5, "4x4x1: v_mov_b32 between (1, vector ALU)");
    RUNALL(6, "4x4x1: s_setprio between (1, scalar)");
    RUNALL(7, "16x16x4: nothing between");
    RUNALL(8, "16x16x4: s_nop 0 between");
    RUNALL(9, "16x16x4: s_nop 1 between");
    RUNALL(10, "16x16x4: v_pk_mul_f32 between");
    return 0;
}
