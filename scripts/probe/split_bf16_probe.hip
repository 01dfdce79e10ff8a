// Probe (gfx950): one 64 x 64 Dense layer of the scaler - forward, dgrad and wgrad on 128-observation tiles - in two arithmetics:
//   F32   v_mfma_f32_16x16x4_f32, the arithmetic of csrc/elbo_mlp.hip (exact fp32, bit-equal to an fmaf chain)
//   SPLIT v_mfma_f32_16x16x32_bf16 on three-way operand splits: x = x1 + x2 + x3 (bf16 each, round-to-nearest on the running
//         residual: 3 x 8 significant bits hold an fp32's 24 exactly), six products x1y1 + x1y2 + x2y1 + x1y3 + x2y2 + x3y1 into one
//         fp32 accumulator, smallest first.  Weights are pre-split once per launch into three LDS planes; activations and their
//         gradients are split on the vector unit where they are produced (that cost is inside the timed tile).
// The layer replaces careless/models/scaling/nn.py:55-68 (tfk.layers.Dense + LeakyReLU(0.01)); the tile structure is the one of
// csrc/elbo_mlp.hip: 512-thread workgroup, a wave owns 16 observations and carries H^T (feature x observation) in the MFMA accumulator
// layout so that a layer's output IS the next layer's B operand; NL layers of the same weights are chained per tile (activations of
// all layers stay in registers until the backward pass); dZ and H meet the weight gradient through LDS staging tiles (observations
// are its contraction axis); 2 workgroup barriers per layer; the weight gradient accumulates in registers across all tiles.
// SPLIT: ONE weight image serves forward (row reads, ds_read_b128) and dgrad (ds_read_b64_tr_b16 transposed reads); the staging tiles
// are bf16 planes [observation][feature] written in the accumulator layout (ds_write_b64) and read transposed.
// Output: cycles per tile (s_memtime of wave 0 of workgroup 0), wall time per tile, and the error of every pass against fp64.
// Usage: ./split_bf16_probe [tiles_per_wg=64] [check_tiles=1024]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDSP __attribute__((address_space(3)))

constexpr int WD = 64;            // layer width
constexpr int TILE = 128;         // observations per tile
constexpr float LEAK = 0.01f;
#ifndef NL
#define NL 4                      // chained layers per tile (same weights)
#endif
constexpr int PW = 160;           // bytes per weight row of a bf16 plane (64 x 2 + pad): forward b128 reads conflict-free
constexpr int PS = 136;           // bytes per observation row of a bf16 staging plane
constexpr int PWF = 68;           // floats per weight row, fp32 image
constexpr int PSF = 132;          // floats per feature row of an fp32 staging tile [feature][128 + 4]

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// four consecutive features of one observation -> three bf16 planes (two dwords each)
__device__ __forceinline__ void split4(const f32x4 x, u32x2& p1, u32x2& p2, u32x2& p3) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned q1 = cvt_pk(a, b);
        const float ra = a - __uint_as_float(q1 << 16), rb = b - __uint_as_float(q1 & 0xffff0000u);
        const unsigned q2 = cvt_pk(ra, rb);
        const float sa = ra - __uint_as_float(q2 << 16), sb = rb - __uint_as_float(q2 & 0xffff0000u);
        p1[i] = q1; p2[i] = q2; p3[i] = cvt_pk(sa, sb);
    }
}
// LeakyReLU' from the packed plane-1 pair of an activation: h > 0 <=> its leading bf16 plane > 0 (same exponent range, round-to-nearest keeps the sign)
__device__ __forceinline__ bool pos_lo(unsigned q) { return (short)(q & 0xffffu) > 0; }
__device__ __forceinline__ bool pos_hi(unsigned q) { return (int)q > 0xffff; }
__device__ __forceinline__ bf16x8 frag(const u32x2 lo, const u32x2 hi) {
    u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
}
#define MFB(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MFF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
// the six products of one k-step, smallest first
#define SIX(acc, a1, a2, a3, b1, b2, b3) do { acc = MFB(a3, b1, acc); acc = MFB(a2, b2, acc); acc = MFB(a1, b3, acc); \
    acc = MFB(a2, b1, acc); acc = MFB(a1, b2, acc); acc = MFB(a1, b1, acc); } while (0)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Args {
    const float* Wt;      // [64 fo][64 fi]
    const float* X;       // [xtiles][128][64]
    float* Ytop;          // [tiles][128][64] or null
    float* dX;            // [tiles][128][64] or null
    float* Z1;            // first layer's pre-activations [tiles][128][64] or null
    float* dWpart;        // [grid][NL][64][64]
    unsigned long long* cyc;
    int tiles, xtiles;
};

template <bool SPLIT>
__global__ __launch_bounds__(512) void layer_probe(Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    // ---- LDS carve
    unsigned char* sW = lds;                                                 // SPLIT: [3][64][PW] bytes; F32: [64][PWF] floats
    unsigned char* sS = lds + (SPLIT ? 3 * WD * PW : WD * PWF * 4);          // SPLIT: dZ planes [3][128][PS], H planes [3][128][PS]; F32: two [64][PSF] tiles
    // ---- weights -> LDS (pre-split once per launch)
    for (int e = tid; e < WD * WD; e += 512) {
        const int fo = e >> 6, fi = e & 63;
        const float w = A.Wt[e];
        if (SPLIT) {
            const int pos = ((fi >> 5) & 1) * 32 + ((fi >> 2) & 3) * 8 + ((fi >> 4) & 1) * 4 + (fi & 3);
            const unsigned q1 = cvt_pk(w, 0.f);
            const float r = w - __uint_as_float(q1 << 16);
            const unsigned q2 = cvt_pk(r, 0.f);
            const float s = r - __uint_as_float(q2 << 16);
            const unsigned q3 = cvt_pk(s, 0.f);
            *(unsigned short*)(sW + 0 * WD * PW + fo * PW + pos * 2) = (unsigned short)q1;
            *(unsigned short*)(sW + 1 * WD * PW + fo * PW + pos * 2) = (unsigned short)q2;
            *(unsigned short*)(sW + 2 * WD * PW + fo * PW + pos * 2) = (unsigned short)q3;
        } else {
            ((float*)sW)[fo * PWF + fi] = w;
        }
    }
    __syncthreads();

    f32x4 acc[NL][2];
#pragma unroll
    for (int l = 0; l < NL; ++l) { acc[l][0] = f32x4{0, 0, 0, 0}; acc[l][1] = f32x4{0, 0, 0, 0}; }
    const int fbo = wave >> 1, fbi0 = 2 * (wave & 1);        // this wave's weight-gradient blocks: rows fbo, columns fbi0, fbi0 + 1

    // per-lane LDS byte offsets
    const unsigned w_fwd = (unsigned)(j * PW + 16 * q);                                   // + (16 fb) PW + 64 s (+ plane)
    const unsigned w_tr = (unsigned)((4 * q + (j >> 2)) * PW + 16 * (j & 3));             // + 32 s PW + 64 (fb >> 1) + 8 (fb & 1); second read + 16 PW
    const unsigned s_wr = (unsigned)((16 * wave + j) * PS + 8 * q);                       // + 32 kb (+ plane)
    const unsigned s_tr = (unsigned)((8 * q + (j >> 2)) * PS + 8 * (j & 3));              // + 32 ks PS + 32 fb; second read + 4 PS

    unsigned long long t0 = 0;
    if (tid == 0) t0 = __builtin_readcyclecounter();
    for (int tile = blockIdx.x; tile < A.tiles; tile += gridDim.x) {
        f32x4 h[NL + 1][4];
        {
            const float* xp = A.X + ((size_t)(tile % A.xtiles) * TILE + 16 * wave + j) * WD + 4 * q;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) h[0][kb] = *(const f32x4*)(xp + 16 * kb);
        }
        // ------------------------------------------------ forward
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            f32x4 z[4];
            if (SPLIT) {
                u32x2 p1[4], p2[4], p3[4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) split4(h[l][kb], p1[kb], p2[kb], p3[kb]);
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    f32x4 c = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned off = w_fwd + fb * 16 * PW + 64 * s;
                        const bf16x8 a1 = *(const bf16x8*)(sW + off), a2 = *(const bf16x8*)(sW + WD * PW + off), a3 = *(const bf16x8*)(sW + 2 * WD * PW + off);
                        const bf16x8 b1 = frag(p1[2 * s], p1[2 * s + 1]), b2 = frag(p2[2 * s], p2[2 * s + 1]), b3 = frag(p3[2 * s], p3[2 * s + 1]);
                        SIX(c, a1, a2, a3, b1, b2, b3);
                    }
                    z[fb] = c;
                }
            } else {
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    f32x4 c = {0, 0, 0, 0};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const f32x4 a = *(const f32x4*)((const float*)sW + (16 * fb + j) * PWF + 16 * kb + 4 * q);
#pragma unroll
                        for (int t = 0; t < 4; ++t) c = MFF(a[t], h[l][kb][t], c);
                    }
                    z[fb] = c;
                }
            }
            if (l == 0 && A.Z1) {
                float* zp = A.Z1 + ((size_t)tile * TILE + 16 * wave + j) * WD + 4 * q;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) *(f32x4*)(zp + 16 * kb) = z[kb];
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t) h[l + 1][kb][t] = fmaxf(z[kb][t], LEAK * z[kb][t]);
        }
        if (A.Ytop) {
            float* yp = A.Ytop + ((size_t)tile * TILE + 16 * wave + j) * WD + 4 * q;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) *(f32x4*)(yp + 16 * kb) = h[NL][kb];
        }
        // ------------------------------------------------ backward
        f32x4 dh[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) dh[kb] = 0.5f * h[NL][kb] - 0.1f;
#pragma unroll
        for (int l = NL - 1; l >= 0; --l) {
            f32x4 dz[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int t = 0; t < 4; ++t) dz[kb][t] = h[l + 1][kb][t] > 0.f ? dh[kb][t] : LEAK * dh[kb][t];
            if (SPLIT) {
                u32x2 d1[4], d2[4], d3[4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) split4(dz[kb], d1[kb], d2[kb], d3[kb]);
                // stage dZ planes and the layer input's planes (its split is repeated here: the forward pass's planes are not kept)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    u32x2 g1, g2, g3;
                    split4(h[l][kb], g1, g2, g3);
                    *(u32x2*)(sS + 0 * TILE * PS + s_wr + 32 * kb) = d1[kb];
                    *(u32x2*)(sS + 1 * TILE * PS + s_wr + 32 * kb) = d2[kb];
                    *(u32x2*)(sS + 2 * TILE * PS + s_wr + 32 * kb) = d3[kb];
                    *(u32x2*)(sS + 3 * TILE * PS + s_wr + 32 * kb) = g1;
                    *(u32x2*)(sS + 4 * TILE * PS + s_wr + 32 * kb) = g2;
                    *(u32x2*)(sS + 5 * TILE * PS + s_wr + 32 * kb) = g3;
                }
                // dgrad between the barriers: dH^T[fi][obs] = sum_fo W[fi][fo] dZ^T[fo][obs]; A = transposed reads of the weight image
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    f32x4 c = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned off = w_tr + 32 * s * PW + 64 * (fb >> 1) + 8 * (fb & 1);
                        u32x2 r[3][2];
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            r[p][0] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)(sW + p * WD * PW + off)));
                            r[p][1] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)(sW + p * WD * PW + off + 16 * PW)));
                        }
                        const bf16x8 a1 = frag(r[0][0], r[0][1]), a2 = frag(r[1][0], r[1][1]), a3 = frag(r[2][0], r[2][1]);
                        const bf16x8 b1 = frag(d1[2 * s], d1[2 * s + 1]), b2 = frag(d2[2 * s], d2[2 * s + 1]), b3 = frag(d3[2 * s], d3[2 * s + 1]);
                        SIX(c, a1, a2, a3, b1, b2, b3);
                    }
                    dh[fb] = c;
                }
                lds_barrier();
                // wgrad: dW^T[fo][fi] += sum_obs dZ^T[fo][obs] H^T[fi][obs], both operands by transposed reads of the staging planes
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bf16x8 a[3], b[2][3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned oa = p * TILE * PS + s_tr + 32 * ks * PS + 32 * fbo;
                        a[p] = frag(__builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)(sS + oa))),
                                    __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)(sS + oa + 4 * PS))));
#pragma unroll
                        for (int bi = 0; bi < 2; ++bi) {
                            const unsigned ob = (3 + p) * TILE * PS + s_tr + 32 * ks * PS + 32 * (fbi0 + bi);
                            b[bi][p] = frag(__builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)(sS + ob))),
                                            __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)(sS + ob + 4 * PS))));
                        }
                    }
                    SIX(acc[l][0], a[0], a[1], a[2], b[0][0], b[0][1], b[0][2]);
                    SIX(acc[l][1], a[0], a[1], a[2], b[1][0], b[1][1], b[1][2]);
                }
                lds_barrier();
            } else {
                float* sZ = (float*)sS;
                float* sH = sZ + WD * PSF;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        sZ[(16 * kb + 4 * q + t) * PSF + 16 * wave + j] = dz[kb][t];
                        sH[(16 * kb + 4 * q + t) * PSF + 16 * wave + j] = h[l][kb][t];
                    }
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    f32x4 c = {0, 0, 0, 0};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int t = 0; t < 4; ++t) c = MFF(((const float*)sW)[(16 * kb + 4 * q + t) * PWF + 16 * fb + j], dz[kb][t], c);
                    dh[fb] = c;
                }
                lds_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const f32x4 a = *(const f32x4*)(sZ + (16 * fbo + j) * PSF + 16 * i + 4 * q);
                    const f32x4 b0 = *(const f32x4*)(sH + (16 * fbi0 + j) * PSF + 16 * i + 4 * q);
                    const f32x4 b1 = *(const f32x4*)(sH + (16 * fbi0 + 16 + j) * PSF + 16 * i + 4 * q);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { acc[l][0] = MFF(a[t], b0[t], acc[l][0]); acc[l][1] = MFF(a[t], b1[t], acc[l][1]); }
                }
                lds_barrier();
            }
        }
        if (A.dX) {
            float* dp = A.dX + ((size_t)tile * TILE + 16 * wave + j) * WD + 4 * q;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) *(f32x4*)(dp + 16 * kb) = dh[kb];
        } else {
            // keep the result alive without a store per tile
            float s = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s += dh[kb][0] + dh[kb][1] + dh[kb][2] + dh[kb][3];
            if (s == 12345.678f) A.dWpart[0] = s;
        }
    }
    if (tid == 0 && blockIdx.x == 0) *A.cyc = __builtin_readcyclecounter() - t0;
    // flush: C layout row = fo 4q + t, column = fi j
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                A.dWpart[(((size_t)blockIdx.x * NL + l) * WD + 16 * fbo + 4 * q + t) * WD + 16 * (fbi0 + bi) + j] = acc[l][bi][t];
}


// ---------------------------------------------------------------------------------------------------- SPLIT, second version
// What is kept per layer are the bf16 planes (24 registers per layer input instead of 16 fp32): the backward pass stages them as they
// are (no second split) and takes LeakyReLU' from plane 1's sign.  PIPE = 1: the A fragments of MFMA group g + 1 are requested before
// the MFMAs of group g (two fragment buffers), and a finished block's LeakyReLU + split (forward) / the next block's nothing (dgrad) sit
// behind the next block's MFMAs in program order, pinned with sched_barrier.
__device__ __forceinline__ bf16x8 lds_b128(const unsigned char* p) { return *(const bf16x8*)p; }
__device__ __forceinline__ u32x2 lds_tr(const unsigned char* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)p));
}
#define SB() __builtin_amdgcn_sched_barrier(0)
template <int DIAG> __device__ __forceinline__ bf16x8 ldsA(const unsigned char* p) {
    if (DIAG & 2) { u32x4 v = {(unsigned)(size_t)p, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}; return __builtin_bit_cast(bf16x8, v); }
    return lds_b128(p);
}
template <int DIAG> __device__ __forceinline__ bf16x8 ldsA2(const unsigned char* p) {      // a forward fragment as two 8-byte pieces 32 bytes apart
    if (DIAG & 2) { u32x4 v = {(unsigned)(size_t)p, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}; return __builtin_bit_cast(bf16x8, v); }
    const u32x2 lo = *(const u32x2*)p, hi = *(const u32x2*)(p + 32);
    return frag(lo, hi);
}
template <int DIAG> __device__ __forceinline__ u32x2 ldsT(const unsigned char* p) {
    if (DIAG & 2) return u32x2{(unsigned)(size_t)p, 0x3f803f80u};
    return lds_tr(p);
}

// DIAG (timing only, WRONG results): 1 = no split arithmetic (a plane is a copy of the fp32 bits' halves).  (Bit 2 -- the MFMA loops read
// no operand from LDS -- compiles into 400 spilled registers and measures nothing; it is not run.)
template <int DIAG>
__device__ __forceinline__ void split4d(const f32x4 x, u32x2& p1, u32x2& p2, u32x2& p3) {
    if (DIAG & 1) {
        p1[0] = __float_as_uint(x[0]); p1[1] = __float_as_uint(x[1]); p2[0] = __float_as_uint(x[2]); p2[1] = __float_as_uint(x[3]); p3[0] = p1[0] ^ p2[1]; p3[1] = p1[1];
    } else split4(x, p1, p2, p3);
}
template <int PIPE, int DIAG = 0>
__global__ __launch_bounds__(512) void layer_probe_split2(Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    unsigned char* sW = lds;
    unsigned char* sS = lds + 3 * WD * PW;
    // conflict-free layouts (scripts/probe/lds_model.py rules; SQ_LDS_BANK_CONFLICT of the first layouts: 40 % of the LDS cycles):
    //   weights [fo][fi] in natural order, pitch 160 B, 8-byte piece index ^ (((row >> 3) & 1) << 1): forward takes a fragment as two
    //   ds_read_b64 (pieces 8 s + q and 8 s + 4 + q), dgrad as two transposed reads whose four lanes per row read 32 contiguous bytes;
    //   staging [obs][feature], pitch 128 B (no padding), piece ^ f(row), f = row bit 0 -> bit 1, bit 1 -> bit 2, bit 3 -> bit 3.
    for (int e = tid; e < WD * WD; e += 512) {
        const int fo = e >> 6, fi = e & 63;
        const float w = A.Wt[e];
        const int pos = ((((fi >> 2) ^ (((fo >> 3) & 1) << 1)) << 2) | (fi & 3));
        const unsigned q1 = cvt_pk(w, 0.f);
        const float r = w - __uint_as_float(q1 << 16);
        const unsigned q2 = cvt_pk(r, 0.f);
        const float s2 = r - __uint_as_float(q2 << 16);
        const unsigned q3 = cvt_pk(s2, 0.f);
        *(unsigned short*)(sW + 0 * WD * PW + fo * PW + pos * 2) = (unsigned short)q1;
        *(unsigned short*)(sW + 1 * WD * PW + fo * PW + pos * 2) = (unsigned short)q2;
        *(unsigned short*)(sW + 2 * WD * PW + fo * PW + pos * 2) = (unsigned short)q3;
    }
    __syncthreads();
    f32x4 acc[NL][2];
#pragma unroll
    for (int l = 0; l < NL; ++l) { acc[l][0] = f32x4{0, 0, 0, 0}; acc[l][1] = f32x4{0, 0, 0, 0}; }
    const int fbo = wave >> 1, fbi0 = 2 * (wave & 1);
    const unsigned char* w_fwd = sW + (j * PW + 8 * (q ^ (((j >> 3) & 1) << 1)));                  // + 16 fb PW + 64 s, second piece + 32
    const unsigned char* w_tr = sW + ((4 * q + (j >> 2)) * PW + 8 * ((j & 3) ^ (((q >> 1) & 1) << 1)));   // + 32 s PW + 32 fb, second read + 16 PW
    constexpr int PS2 = 128;
    const int fj = ((j & 1) << 1) | (((j >> 1) & 1) << 2) | (((j >> 3) & 1) << 3);                   // f(row) of the staging row this lane WRITES (16 wave + j)
    unsigned char* s_wr[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) s_wr[kb] = sS + (16 * wave + j) * PS2 + 8 * ((4 * kb + q) ^ fj);
    const int rq = j >> 2, rp = j & 3;                                                               // transposed reads: row 8 q + rq (+ 4), piece 4 fb + rp
    const int fr = ((rq & 1) << 1) | (((rq >> 1) & 1) << 2) | ((q & 1) << 3);
    const unsigned char* s_tra = sS + (8 * q + rq) * PS2 + 8 * ((4 * fbo + rp) ^ fr);                // + plane, + 32 ks PS2, second read + 4 PS2
    const unsigned char* s_trb[2] = {sS + (8 * q + rq) * PS2 + 8 * ((4 * fbi0 + rp) ^ fr), sS + (8 * q + rq) * PS2 + 8 * ((4 * (fbi0 + 1) + rp) ^ fr)};

    unsigned long long t0 = 0;
    if (tid == 0) t0 = __builtin_readcyclecounter();
    for (int tile = blockIdx.x; tile < A.tiles; tile += gridDim.x) {
        u32x2 hp[NL][3][4];               // planes of the input of layer l: [plane][kb] = features 16 kb + 4 q + 0..3 of observation j
        f32x4 htop[4];
        {
            const float* xp = A.X + ((size_t)(tile % A.xtiles) * TILE + 16 * wave + j) * WD + 4 * q;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) split4d<DIAG>(*(const f32x4*)(xp + 16 * kb), hp[0][0][kb], hp[0][1][kb], hp[0][2][kb]);
        }
        // ------------------------------------------------ forward
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            f32x4 z[4];
            bf16x8 af[2][3];
            if (PIPE) {
#pragma unroll
                for (int p = 0; p < 3; ++p) af[0][p] = ldsA2<DIAG>(w_fwd + p * WD * PW);
            }
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                f32x4 c = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int g = 2 * fb + s;
                    if (PIPE) {
                        if (g + 1 < 8) {
#pragma unroll
                            for (int p = 0; p < 3; ++p) af[(g + 1) & 1][p] = ldsA2<DIAG>(w_fwd + p * WD * PW + ((g + 1) >> 1) * 16 * PW + 64 * ((g + 1) & 1));
                        }
                        SB();
                    } else {
#pragma unroll
                        for (int p = 0; p < 3; ++p) af[g & 1][p] = ldsA2<DIAG>(w_fwd + p * WD * PW + fb * 16 * PW + 64 * s);
                    }
                    const bf16x8 b1 = frag(hp[l][0][2 * s], hp[l][0][2 * s + 1]), b2 = frag(hp[l][1][2 * s], hp[l][1][2 * s + 1]), b3 = frag(hp[l][2][2 * s], hp[l][2][2 * s + 1]);
                    SIX(c, af[g & 1][0], af[g & 1][1], af[g & 1][2], b1, b2, b3);
                    if (PIPE && s == 1 && fb > 0) {
                        // the block before this one: LeakyReLU, then its planes (or the top activations) -- vector work in the shadow of this block's MFMAs
                        const int fp = fb - 1;
#pragma unroll
                        for (int t = 0; t < 4; ++t) z[fp][t] = fmaxf(z[fp][t], LEAK * z[fp][t]);
                        if (l + 1 < NL) split4d<DIAG>(z[fp], hp[l + 1 < NL ? l + 1 : 0][0][fp], hp[l + 1 < NL ? l + 1 : 0][1][fp], hp[l + 1 < NL ? l + 1 : 0][2][fp]);
                        else htop[fp] = z[fp];
                    }
                    if (PIPE) SB();
                }
                z[fb] = c;
                if (l == 0 && A.Z1) *(f32x4*)(A.Z1 + ((size_t)tile * TILE + 16 * wave + j) * WD + 4 * q + 16 * fb) = c;
            }
#pragma unroll
            for (int fb = (PIPE ? 3 : 0); fb < 4; ++fb) {
#pragma unroll
                for (int t = 0; t < 4; ++t) z[fb][t] = fmaxf(z[fb][t], LEAK * z[fb][t]);
                if (l + 1 < NL) split4d<DIAG>(z[fb], hp[l + 1 < NL ? l + 1 : 0][0][fb], hp[l + 1 < NL ? l + 1 : 0][1][fb], hp[l + 1 < NL ? l + 1 : 0][2][fb]);
                else htop[fb] = z[fb];
            }
        }
        if (A.Ytop) {
            float* yp = A.Ytop + ((size_t)tile * TILE + 16 * wave + j) * WD + 4 * q;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) *(f32x4*)(yp + 16 * kb) = htop[kb];
        }
        // ------------------------------------------------ backward
        f32x4 dh[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) dh[kb] = 0.5f * htop[kb] - 0.1f;
#pragma unroll
        for (int l = NL - 1; l >= 0; --l) {
            u32x2 d1[4], d2[4], d3[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                f32x4 dz;
                if (l == NL - 1) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) dz[t] = htop[kb][t] > 0.f ? dh[kb][t] : LEAK * dh[kb][t];
                } else {
                    const u32x2 m = hp[l + 1 < NL ? l + 1 : 0][0][kb];
                    dz[0] = pos_lo(m[0]) ? dh[kb][0] : LEAK * dh[kb][0];
                    dz[1] = pos_hi(m[0]) ? dh[kb][1] : LEAK * dh[kb][1];
                    dz[2] = pos_lo(m[1]) ? dh[kb][2] : LEAK * dh[kb][2];
                    dz[3] = pos_hi(m[1]) ? dh[kb][3] : LEAK * dh[kb][3];
                }
                split4d<DIAG>(dz, d1[kb], d2[kb], d3[kb]);
                *(u32x2*)(s_wr[kb] + 0 * TILE * PS2) = d1[kb];
                *(u32x2*)(s_wr[kb] + 1 * TILE * PS2) = d2[kb];
                *(u32x2*)(s_wr[kb] + 2 * TILE * PS2) = d3[kb];
                *(u32x2*)(s_wr[kb] + 3 * TILE * PS2) = hp[l][0][kb];
                *(u32x2*)(s_wr[kb] + 4 * TILE * PS2) = hp[l][1][kb];
                *(u32x2*)(s_wr[kb] + 5 * TILE * PS2) = hp[l][2][kb];
            }
            // dgrad between the barriers
            {
                u32x2 r[2][3][2];
                if (PIPE) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) { r[0][p][0] = ldsT<DIAG>(w_tr + p * WD * PW); r[0][p][1] = ldsT<DIAG>(w_tr + p * WD * PW + 16 * PW); }
                }
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    f32x4 c = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int g = 2 * fb + s;
                        if (PIPE) {
                            if (g + 1 < 8) {
                                const int fn = (g + 1) >> 1, sn = (g + 1) & 1;
                                const unsigned off = 32 * sn * PW + 32 * fn;
#pragma unroll
                                for (int p = 0; p < 3; ++p) { r[(g + 1) & 1][p][0] = ldsT<DIAG>(w_tr + p * WD * PW + off); r[(g + 1) & 1][p][1] = ldsT<DIAG>(w_tr + p * WD * PW + off + 16 * PW); }
                            }
                            SB();
                        } else {
                            const unsigned off = 32 * s * PW + 32 * fb;
#pragma unroll
                            for (int p = 0; p < 3; ++p) { r[g & 1][p][0] = ldsT<DIAG>(w_tr + p * WD * PW + off); r[g & 1][p][1] = ldsT<DIAG>(w_tr + p * WD * PW + off + 16 * PW); }
                        }
                        const bf16x8 a1 = frag(r[g & 1][0][0], r[g & 1][0][1]), a2 = frag(r[g & 1][1][0], r[g & 1][1][1]), a3 = frag(r[g & 1][2][0], r[g & 1][2][1]);
                        const bf16x8 b1 = frag(d1[2 * s], d1[2 * s + 1]), b2 = frag(d2[2 * s], d2[2 * s + 1]), b3 = frag(d3[2 * s], d3[2 * s + 1]);
                        SIX(c, a1, a2, a3, b1, b2, b3);
                        if (PIPE) SB();
                    }
                    dh[fb] = c;
                }
            }
            lds_barrier();
            // wgrad
            {
                u32x2 ra[2][3][2], rb[2][2][3][2];
                auto load = [&](int buf, int ks) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned char* pa = s_tra + p * TILE * PS2 + 32 * ks * PS2;
                        ra[buf][p][0] = ldsT<DIAG>(pa); ra[buf][p][1] = ldsT<DIAG>(pa + 4 * PS2);
#pragma unroll
                        for (int bi = 0; bi < 2; ++bi) {
                            const unsigned char* pb = s_trb[bi] + (3 + p) * TILE * PS2 + 32 * ks * PS2;
                            rb[buf][bi][p][0] = ldsT<DIAG>(pb); rb[buf][bi][p][1] = ldsT<DIAG>(pb + 4 * PS2);
                        }
                    }
                };
                if (PIPE) load(0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int cur = PIPE ? (ks & 1) : 0;
                    if (PIPE) { if (ks + 1 < 4) load((ks + 1) & 1, ks + 1); SB(); }
                    else load(0, ks);
                    const bf16x8 a1 = frag(ra[cur][0][0], ra[cur][0][1]), a2 = frag(ra[cur][1][0], ra[cur][1][1]), a3 = frag(ra[cur][2][0], ra[cur][2][1]);
#pragma unroll
                    for (int bi = 0; bi < 2; ++bi) {
                        const bf16x8 b1 = frag(rb[cur][bi][0][0], rb[cur][bi][0][1]), b2 = frag(rb[cur][bi][1][0], rb[cur][bi][1][1]), b3 = frag(rb[cur][bi][2][0], rb[cur][bi][2][1]);
                        SIX(acc[l][bi], a1, a2, a3, b1, b2, b3);
                    }
                    if (PIPE) SB();
                }
            }
            lds_barrier();
        }
        if (A.dX) {
            float* dp = A.dX + ((size_t)tile * TILE + 16 * wave + j) * WD + 4 * q;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) *(f32x4*)(dp + 16 * kb) = dh[kb];
        } else {
            float sm = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) sm += dh[kb][0] + dh[kb][1] + dh[kb][2] + dh[kb][3];
            if (sm == 12345.678f) A.dWpart[0] = sm;
        }
    }
    if (tid == 0 && blockIdx.x == 0) *A.cyc = __builtin_readcyclecounter() - t0;
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                A.dWpart[(((size_t)blockIdx.x * NL + l) * WD + 16 * fbo + 4 * q + t) * WD + 16 * (fbi0 + bi) + j] = acc[l][bi][t];
}

// -------------------------------------------------------------------------------------------------------- host
static void reference(const std::vector<float>& Wt, const std::vector<float>& X, int tiles, std::vector<double>& Y, std::vector<double>& dX,
                      std::vector<double>& Z1, std::vector<double>& dW) {
    const int n = tiles * TILE;
    Y.assign((size_t)n * WD, 0); dX.assign((size_t)n * WD, 0); Z1.assign((size_t)n * WD, 0); dW.assign((size_t)NL * WD * WD, 0);
#pragma omp parallel
    {
        std::vector<double> dWl((size_t)NL * WD * WD, 0.0);
#pragma omp for
        for (int i = 0; i < n; ++i) {
            double h[NL + 1][WD], dh[WD], dz[WD];
            for (int f = 0; f < WD; ++f) h[0][f] = X[(size_t)i * WD + f];
            for (int l = 0; l < NL; ++l)
                for (int fo = 0; fo < WD; ++fo) {
                    double z = 0;
                    for (int fi = 0; fi < WD; ++fi) z += (double)Wt[fo * WD + fi] * h[l][fi];
                    if (l == 0) Z1[(size_t)i * WD + fo] = z;
                    h[l + 1][fo] = z > 0 ? z : (double)LEAK * z;
                }
            for (int f = 0; f < WD; ++f) { Y[(size_t)i * WD + f] = h[NL][f]; dh[f] = 0.5 * h[NL][f] - 0.1; }
            for (int l = NL - 1; l >= 0; --l) {
                for (int f = 0; f < WD; ++f) dz[f] = h[l + 1][f] > 0 ? dh[f] : (double)LEAK * dh[f];
                for (int fo = 0; fo < WD; ++fo)
                    for (int fi = 0; fi < WD; ++fi) dWl[((size_t)l * WD + fo) * WD + fi] += dz[fo] * h[l][fi];
                for (int fi = 0; fi < WD; ++fi) {
                    double s = 0;
                    for (int fo = 0; fo < WD; ++fo) s += (double)Wt[fo * WD + fi] * dz[fo];
                    dh[fi] = s;
                }
            }
            for (int f = 0; f < WD; ++f) dX[(size_t)i * WD + f] = dh[f];
        }
#pragma omp critical
        for (size_t e = 0; e < dWl.size(); ++e) dW[e] += dWl[e];
    }
}

struct Err { double rms, mx, ref_rms; };
static Err err_of(const float* got, const double* ref, size_t n) {
    double s = 0, m = 0, r = 0;
    for (size_t i = 0; i < n; ++i) { const double d = got[i] - ref[i]; s += d * d; m = fmax(m, fabs(d)); r += ref[i] * ref[i]; }
    return {sqrt(s / n), m, sqrt(r / n)};
}

typedef void (*kern_t)(Args);
static void run(kern_t kern, bool SPLIT, const char* name, const std::vector<float>& Wt, const std::vector<float>& X, int check_tiles, int tiles_per_wg, const char* data_name) {
    const int grid = 256;
    const size_t ldsz = SPLIT ? (size_t)3 * WD * PW + (size_t)6 * TILE * PS : (size_t)WD * PWF * 4 + (size_t)2 * WD * PSF * 4;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsz));
    float *dWt, *dXin, *dY, *dDX, *dZ1, *dPart; unsigned long long* dCyc;
    const size_t nchk = (size_t)check_tiles * TILE * WD;
    CHECK(hipMalloc(&dWt, WD * WD * 4)); CHECK(hipMalloc(&dXin, nchk * 4)); CHECK(hipMalloc(&dY, nchk * 4)); CHECK(hipMalloc(&dDX, nchk * 4));
    CHECK(hipMalloc(&dZ1, nchk * 4)); CHECK(hipMalloc(&dPart, (size_t)grid * NL * WD * WD * 4)); CHECK(hipMalloc(&dCyc, 8));
    CHECK(hipMemcpy(dWt, Wt.data(), WD * WD * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dXin, X.data(), nchk * 4, hipMemcpyHostToDevice));
    // ---- correctness + error study on check_tiles tiles
    Args a{dWt, dXin, dY, dDX, dZ1, dPart, dCyc, check_tiles, check_tiles};
    CHECK(hipMemset(dPart, 0, (size_t)grid * NL * WD * WD * 4));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ldsz, 0, a);
    CHECK(hipDeviceSynchronize());
    std::vector<float> Y(nchk), DX(nchk), Z1(nchk), part((size_t)grid * NL * WD * WD);
    CHECK(hipMemcpy(Y.data(), dY, nchk * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(DX.data(), dDX, nchk * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(Z1.data(), dZ1, nchk * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(part.data(), dPart, part.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> rY, rDX, rZ1, rDW;
    reference(Wt, X, check_tiles, rY, rDX, rZ1, rDW);
    std::vector<float> dW((size_t)NL * WD * WD, 0.f);
    {
        std::vector<double> s(dW.size(), 0.0);
        for (int g = 0; g < grid; ++g) for (size_t e = 0; e < s.size(); ++e) s[e] += part[(size_t)g * s.size() + e];
        for (size_t e = 0; e < s.size(); ++e) dW[e] = (float)s[e];
    }
    const Err eZ = err_of(Z1.data(), rZ1.data(), nchk), eY = err_of(Y.data(), rY.data(), nchk), eD = err_of(DX.data(), rDX.data(), nchk), eW = err_of(dW.data(), rDW.data(), dW.size());
    printf("%-6s %-10s | layer-1 dot products (%zu): rms err %.3e max %.3e (rms value %.3e, rel %.3e) | %d-layer output rel rms %.3e | input gradient rel rms %.3e max/rms-value %.3e | weight gradient rel rms %.3e max/rms-value %.3e\n",
           name, data_name, nchk, eZ.rms, eZ.mx, eZ.ref_rms, eZ.rms / eZ.ref_rms, NL, eY.rms / eY.ref_rms, eD.rms / eD.ref_rms, eD.mx / eD.ref_rms, eW.rms / eW.ref_rms, eW.mx / eW.ref_rms);
    // ---- timing: tiles_per_wg tiles per workgroup, inputs re-used (L2-resident), nothing stored per tile
    Args t{dWt, dXin, nullptr, nullptr, nullptr, dPart, dCyc, grid * tiles_per_wg, check_tiles < 64 ? check_tiles : 64};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ldsz, 0, t);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long cyc; CHECK(hipMemcpy(&cyc, dCyc, 8, hipMemcpyDeviceToHost));
        const double flop = 6.0 * WD * WD * TILE * NL * (double)grid * tiles_per_wg;
        printf("%-6s timing rep %d: %.3f ms, %.2f us per tile of %d layers, %.0f cycles (s_memtime ticks) per tile = %.0f per layer; %.1f algorithmic TFLOP/s = %.3f of the fp32 matrix peak (157.3)\n",
               name, rep, ms, 1e3 * ms / tiles_per_wg, NL, (double)cyc / tiles_per_wg, (double)cyc / tiles_per_wg / NL, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
    }
    hipFree(dWt); hipFree(dXin); hipFree(dY); hipFree(dDX); hipFree(dZ1); hipFree(dPart); hipFree(dCyc);
}

int main(int argc, char** argv) {
    const int tiles_per_wg = argc > 1 ? atoi(argv[1]) : 64;
    const int check_tiles = argc > 2 ? atoi(argv[2]) : 1024;
    std::mt19937_64 rng(1234);
    std::normal_distribution<float> nrm(0.f, 1.f);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    std::vector<float> Wt(WD * WD);
    // a trained scaler's kernel: identity-initialised (nn.py:62-67) plus dense structure; the spectral radius stays near one over NL layers
    for (int fo = 0; fo < WD; ++fo) for (int fi = 0; fi < WD; ++fi) Wt[fo * WD + fi] = (fo == fi ? 0.8f : 0.f) + 0.09f * nrm(rng);
    const size_t n = (size_t)check_tiles * TILE * WD;
    std::vector<float> X(n);
    printf("split_bf16_probe: %d chained 64x64 layers per 128-observation tile, 256 workgroups x 512 threads, %d tiles per workgroup timed, %d tiles checked against fp64\n", NL, tiles_per_wg, check_tiles);
    // data set 1: the activations' real range - standardised metadata / LeakyReLU outputs of O(1)
    for (size_t i = 0; i < n; ++i) { const float v = nrm(rng); X[i] = v > 0 ? v : LEAK * v; }
    run(layer_probe<false>, false, "F32", Wt, X, check_tiles, tiles_per_wg, "unit");
    run(layer_probe<true>, true, "SPLIT0", Wt, X, check_tiles, tiles_per_wg, "unit");
    run(layer_probe_split2<0>, true, "SPLIT1", Wt, X, check_tiles, tiles_per_wg, "unit");
    run(layer_probe_split2<1>, true, "SPLIT2", Wt, X, check_tiles, tiles_per_wg, "unit");
    printf("-- diagnostic builds of SPLIT1 (wrong results by construction; read the timing lines only)\n");
    run(layer_probe_split2<0, 1>, true, "S1-nosplit", Wt, X, 8, tiles_per_wg, "unit");
    // data set 2: six decades of magnitudes in one dot product
    for (size_t i = 0; i < n; ++i) X[i] = (uni(rng) < 0.5f ? -1.f : 1.f) * powf(10.f, -3.f + 6.f * uni(rng));
    run(layer_probe<false>, false, "F32", Wt, X, check_tiles < 128 ? check_tiles : 128, 8, "6-decades");
    run(layer_probe_split2<1>, true, "SPLIT2", Wt, X, check_tiles < 128 ? check_tiles : 128, 8, "6-decades");
    return 0;
}
