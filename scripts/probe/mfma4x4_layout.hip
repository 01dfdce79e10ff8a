// Diagnostic (gfx950): operand / result layout of v_mfma_f32_4x4x1_16b_f32 and the meaning of CBSZ / ABID (A broadcast).
// Expected: D_b[i][j] = A_b[i] * B_b[j] + C_b[i][j];  A_b[i] in lane 4b + i, B_b[j] in lane 4b + j, D_b[i][j] in VGPR i of lane 4b + j;
// with CBSZ = 4, ABID = n every block takes the A of block n.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void k(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], c, CBSZ, ABID, 0);
    for (int i = 0; i < 4; ++i) d[i * 64 + lane] = c[i];
}
template <int CBSZ, int ABID>
int check(const char* name) {
    float ha[64], hb[64], hd[256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
    float *a, *b, *d;
    (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&d, 1024);
    (void)hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, a, b, d);
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; ++blk)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const int ab = CBSZ == 4 ? ABID : blk;
                const float want = ha[4 * ab + i] * hb[4 * blk + j];
                if (std::fabs(hd[i * 64 + 4 * blk + j] - want) > 1e-3f * std::fabs(want)) ++bad;
            }
    printf("%-28s mismatches: %d   (D[0] lane 5 = %.1f, lane 6 = %.1f; VGPR 1 lane 5 = %.1f)\n", name, bad, hd[5], hd[6], hd[64 + 5]);
    return bad;
}
int main() {
    int bad = check<0, 0>("cbsz 0");
    bad += check<4, 0>("cbsz 4 abid 0");
    bad += check<4, 7>("cbsz 4 abid 7");
    bad += check<4, 15>("cbsz 4 abid 15");
    printf(bad ? "LAYOUT ASSUMPTION WRONG\n" : "layout as expected\n");
    return bad != 0;
}
