// Diagnostic (gfx950): cycles per ds_read_b128 for the lane -> address patterns of the GEMM kernels' operand reads (csrc/wide_gemm.hip,
// csrc/elbo_mlp.hip), one wave alone on a CU and 16 waves per CU.  A check of scripts/probe/lds_model.py (which 16 lanes share an LDS cycle): the loop
// is latency-bound (every read is consumed at once), so a conflict shows as ONE more cycle per read in the 16-wave column -- 34 against 33 --
// exactly for the patterns the model calls two-way (profiles/r4_probe_lds_b128.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(const int* __restrict__ offs, float* out, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float* p = sm + offs[lane];
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(uintptr_t)p), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            acc += v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
static double run(const std::vector<int>& offs, int threads) {
    int* d; float* out; unsigned long long* cyc;
    (void)hipMalloc(&d, 64 * 4); (void)hipMalloc(&out, 1024 * 4 * 4); (void)hipMalloc(&cyc, 8);
    (void)hipMemcpy(d, offs.data(), 64 * 4, hipMemcpyHostToDevice);
    const int iters = 4000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 65536, 0, d, out, 10, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemset(cyc, 0, 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(threads), 65536, 0, d, out, iters, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d); (void)hipFree(out); (void)hipFree(cyc);
    return (double)c / ((double)iters * 16);
}
int main() {
    struct P { const char* name; std::vector<int> o; };
    std::vector<P> ps;
    auto mk = [&](const char* name, auto f) { P p; p.name = name; for (int l = 0; l < 64; ++l) p.o.push_back(f(l & 15, l >> 4, l)); ps.push_back(p); };
    mk("linear: lane l -> quad l (every 16 contiguous lanes on 16 different slots)", [](int j, int q, int l) { return 4 * l; });
    mk("all lanes the same address (broadcast)", [](int j, int q, int l) { return 0; });
    mk("row j, pitch 132, k = 4 q     (round-3 weight image)", [](int j, int q, int l) { return j * 132 + 4 * q; });
    mk("row j, pitch 136, k = 4 q", [](int j, int q, int l) { return j * 136 + 4 * q; });
    mk("row j, pitch 128, quad q ^ j  (round-4 experiment)", [](int j, int q, int l) { return j * 128 + 4 * (q ^ j); });
    mk("row j, pitch 36, k = 4 q      (round-3 tiles)", [](int j, int q, int l) { return j * 36 + 4 * q; });
    mk("row j, pitch 36, quad q ^ (j >> 3)  (round-4 transposed tiles)", [](int j, int q, int l) { return j * 36 + 4 * (q ^ ((j >> 3) & 7)); });
    mk("row j, pitch 32, quad q ^ ((j >> 1) & 7)  (round-4 experiment)", [](int j, int q, int l) { return j * 32 + 4 * (q ^ ((j >> 1) & 7)); });
    mk("row j, pitch 20, k = 4 q      (first-layer image)", [](int j, int q, int l) { return j * 20 + 4 * q; });
    mk("two-way by construction: lanes l and l ^ 1 on one slot, different rows", [](int j, int q, int l) { return 4 * (l >> 1) + 256 * (l & 1) * 4; });
    mk("row j, pitch 68 (elbo_mlp 64-wide image), k = 4 q", [](int j, int q, int l) { return j * 68 + 4 * q; });
    printf("cycles per ds_read_b128 (one workgroup on one CU; 8 reads in flight per wave)\n%-78s %8s %8s\n", "pattern", "1 wave", "16 waves");
    for (auto& p : ps) printf("%-78s %8.2f %8.2f\n", p.name, run(p.o, 64), run(p.o, 1024) );
    return 0;
}
