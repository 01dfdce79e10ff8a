// Scalers wider than the fused kernels hold (hidden or metadata width > 64): the Dense stack of `MLPScaler` layer by layer on
// hand-written fp32-MFMA GEMM kernels (gfx950 / CDNA4 only), activations through HBM.
//
// Reference: `MetadataScaler.call` / `NormalLayer` (careless/models/scaling/nn.py:55-68, 92-120) and `tape.gradient` of it
// (careless/models/merging/variational.py:197-202): h_l = LeakyReLU(h_{l-1} W_l + b_l), o = h_L W_o + b_o.  The reference takes any
// width; up to 64 the whole stack runs inside ONE launch with nothing per-observation in HBM (elbo_mlp.hip).  Past that a
// layer's activations of a 128-observation tile no longer fit a wave's registers next to its weight-gradient blocks, so the
// layers run one GEMM each, in row chunks (engine._data_term_wide), around the same likelihood kernels as the two-pass Laue path.
//
// One kernel template, v_mfma_f32_16x16x4_f32 (exact fp32: the parity bar of the narrow path holds here too), three uses:
//   forward   Y[n][out]  = LeakyReLU(X[n][in] Wt[out][in]^T + b)                 A, B contraction-contiguous      epilogue: bias, LeakyReLU
//   dgrad     dX[n][in]  = (dZ[n][out] Wt[out][in]) * LeakyReLU'(H[n][in])       B contraction-major              epilogue: derivative mask
//   wgrad     dWt[out][in] = dZ[n][out]^T H[n][in], db = column sums of dZ     A, B contraction-major, split over the observations:
//                                   every workgroup writes its partial sums in the layer's flat W^T layout; cl_reduce_partials
//                                   adds them in index order (deterministic, as for the fused kernels)
// Activation buffers are row-major [rows][ld] with ld = the width rounded up to 4.
// Tiled kernel (any width; the weight gradient always): 128 x 128 (or 64) outputs per 256-thread workgroup (4 waves, 32 rows each), 32-deep
// contraction chunks through double-buffered LDS; an operand stored contraction-contiguous is staged [row][32 + 4] and read as one
// ds_read_b128 per four MFMA steps, one stored contraction-major is staged [32][rows + 4] and read as four ds_read_b32.
// Streaming kernel (forward and dgrad of layers up to 128 x 128 -- widths 65 .. 128, the range past the fused kernels that matters):
// the layer's weights stay in LDS for the whole launch, every wave walks 16-row blocks on its own -- its rows' operand straight from
// global memory into the MFMA B-operand layout (one float4 per lane and 16-deep chunk, the next block's in flight), the transposed
// output tile in accumulators, one float4 store per lane and 16 output columns -- with no workgroup barrier in the loop.
// Round 4 -- what a step of a Dense-only scaler of width 65 .. 128 launches: wide_stream2_kernel (layers 0 + 1, the first
// layer's output never stored), wide_sq_kernel forward with the Dense(2) head in the top layer's epilogue, slot_rows_kernel (elbo_laue.hip),
// then top down: weight gradient and dgrad of the top layer with the head's backward pass made where they read dZ (HEADW / HEADB instances),
// plain weight gradients and dgrads in between, the second layer's weight gradient with the recomputed first layer made by MFMAs in the
// staged layout (PREM), its dgrad with the FIRST layer's weight gradient taken from the output block in registers (WG0).  Every
// weight-gradient partial is summed in index order by cl_reduce_partials: the path has no float atomic of its own.
// Roofline: the layers are unfused, so a layer moves 4 (in + out) bytes per observation for 2 in out flops -- at width 128 that is
// 32 flop / B against a ridge of 19.7; measured (per-kernel counters, profiles/r4_pmc_by_kernel_*.txt) the kernels are bound by MFMA issue
// with the matrix pipe 62 - 78 % busy, not by HBM (3 TB/s).  6 P_mm flops per observation when the forward pass keeps the activations
// (the default), 8 when a data set is too large for that and every chunk's forward pass runs again in the backward pass.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include "cl_math.h"
#include "cl_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CL_WIDE_DIAG
#define CL_WIDE_DIAG 0      /* diagnostic builds only (WRONG results): 1 = the stream kernels read no row operand from global memory, 2 = they store no
                               output, 4 = the tiled kernel reads no operand tiles; what is left of a kernel's time is its MFMA + LDS floor */
#endif

namespace {

// gfx950 serves a ds_read_b128 in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md,
// LDS) --, one cycle per group when its lanes touch 16 different 16-byte slots of the 256-byte bank row.  Every operand read of these kernels
// has lane (j, q) on row j of a tile whose pitch is an odd number of quads, at quad q: slot j + q mod 16 -- lanes (11, 1) and (12, 0) of the first group collide, every read takes 8
// cycles instead of 4 (scripts/probe/lds_model.py; the probe scripts/probe/lds_b128_probe.hip sees the same patterns conflict).  Swapping the
// two quads of each pair in rows 4 .. 11 of every 16 (quad ^ swzb(row)) separates them: the lane's base address changes, the per-block and
// per-chunk offsets stay immediates.
__device__ __forceinline__ int swzb(int row) { return ((row + 4) & 8) ? 1 : 0; }
// element k of row o in a weight image of pitch P whose first KW floats (a multiple of 8) are data: the padding behind them stays in place
__device__ __forceinline__ int img_at(int o, int k, int P, int KW) { return o * P + (k < KW ? (k ^ (4 * swzb(o))) : k); }

constexpr int BM = 128, BK = 32;
constexpr int PK = BK + 4;          // pitch of a contraction-contiguous tile [rows][BK]
constexpr int PMA = BM + 4;         // pitch of a contraction-major tile [BK][rows] (rows + 4: the four k-groups of a b32 read hit different banks)
constexpr int SA = (BM * PK > BK * PMA) ? BM * PK : BK * PMA;

enum { EPI_BIAS_LRELU = 0, EPI_DLRELU = 1, EPI_WGRAD = 2 };

// The FIRST Dense layer of the stack, h_0 = LeakyReLU(X_0 Wt_0^T + b_0) on the metadata rows (careless/models/scaling/nn.py:55-68), recomputed
// wherever its output is needed instead of being stored (round 4): it is a (rows x <= 16) x (<= 16 x w) product -- an eighth of a
// 128 x 128 layer's work -- against 4 w bytes per row written once and read three times (the second layer's forward, its weight
// gradient, the mask of its dgrad).  X0 == NULL: not in use.
constexpr int K0MAX = 16;            // metadata columns of one 16-deep chunk: the fused forms' first-layer image [N0][K0MAX] ...
constexpr int K0WG = 15;             // ... of which the recomputed first layer takes up to 15 (column K0 of the chunk carries the ones of the bias
                                     // gradient in cl_wide_dense_dgrad_pre_wgrad0)
constexpr int S0P = K0MAX + 4;       // pitch of the first layer's weight image [N0][K0MAX]
struct PreArgs {
    const float* X0; int ldx0; int K0;      // metadata rows [n][ldx0], K0 columns in use
    const float* W0; const float* b0;       // Wt_0[N0][K0], b_0[N0]
    int N0;
};

struct GemmArgs {
    const float* A; int lda;        // not AK: A[m][k] at A[m * lda + k];  AK: A[k][m] at A[k * lda + m]
    const float* B; int ldb;        // not BK_: B[n][k] at B[n * ldb + k]; BK_: B[k][n] at B[k * ldb + n]
    float* C; int ldc;              // C[m][n] (EPI_WGRAD: the layer's flat W^T slice of this workgroup's partial)
    int M, N, K;
    const float* bias;              // EPI_BIAS_LRELU: [N]
    const float* H; int ldh;        // EPI_DLRELU: activations whose sign selects the derivative, H[m][n]
    float leak;
    int act;                        // EPI_BIAS_LRELU: 0 = no activation
    int ksplit;                     // EPI_WGRAD: contraction range of a z-block; partial stride (floats)
    long long pstride;
    int n_in;                       // EPI_WGRAD: N = n_in (the bias gradient = column sums of A, taken on the way)
    const int* seg;                 // EPI_WGRAD, grouped (per-image layers): z-block g contracts rows seg[g] .. seg[g+1] and writes the group's
    float* Cb; long long bstride;   //   kernel at C + g * pstride, its bias at Cb + g * bstride (NULL: the flat layer layout, bias behind the kernel)
    const int* stop_flag;
    PreArgs pre;                    // EPI_WGRAD: the B operand (the layer's input) is the recomputed first layer (pre.X0 != NULL; B, ldb unused)
    // Grouped forward / dgrad (per-image layers wider than the streaming kernel holds, round 4): x-block i works on the BM rows from
    // tiles[2 i + 1] of group tiles[2 i] (rows seg[g] .. seg[g + 1]), whose B operand sits at B + g * bgs and bias at bias + g * biasgs
    const int* tiles; long long bgs, biasgs;
    // EPI_WGRAD of the TOP layer with the head's backward pass fused (HEADW instances, round 4): A = the top layer's activations h_L[k][m] in
    // place of dZ_L, which is made from them while the tile is staged: dZ_L[k][m] = (g0[k] Wo[0][m] + g1[k] Wo[1][m]) * LeakyReLU'(h_L[k][m]),
    // g0 = dL/dloc, g1 = dL/dsigma * dsigma/draw of observation k; the head's own weight gradient (sums of h_L g over k) is taken on the way
    const float* hd_dO; const float* hd_dsd; const float* hd_W;     // [K][2], [K], the head's flat layout [Wo (2 x M) | bo (2)]
    float* hd_part;                                                 // [z][2 M + 2] partial sums of (dWo | dbo)
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// global -> registers: one tile of an operand (ROWS x BK), zero outside [0, rows_lim) x [0, k_lim)
template <int ROWS, bool KMAJOR>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int row0, int k0, int rows_lim, int k_lim, bool vec_ok,
                                          f32x4 (&r)[ROWS * BK / 1024], int tid) {
    constexpr int NV = ROWS * BK / 1024;          // float4 per thread
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int idx = v * 256 + tid;
        int row, k;
        if (!KMAJOR) { row = row0 + idx / (BK / 4); k = k0 + 4 * (idx % (BK / 4)); }          // 8 threads per row: 128 contiguous bytes
        else { k = k0 + idx / (ROWS / 4); row = row0 + 4 * (idx % (ROWS / 4)); }              // ROWS / 4 threads per contraction index
        f32x4 x = {0.0f, 0.0f, 0.0f, 0.0f};
        if (CL_WIDE_DIAG & 4) { r[v] = f32x4{1.0f, 0.5f, -0.25f, 0.125f}; continue; }
        if (!KMAJOR) {
            if (row < rows_lim) {
                const float* p = P + (size_t)row * ld + k;
                if (vec_ok && k + 3 < k_lim) x = *reinterpret_cast<const f32x4*>(p);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < k_lim) x[e] = p[e];
                }
            }
        } else {
            if (k < k_lim) {
                const float* p = P + (size_t)k * ld + row;
                if (vec_ok && row + 3 < rows_lim) x = *reinterpret_cast<const f32x4*>(p);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (row + e < rows_lim) x[e] = p[e];
                }
            }
        }
        r[v] = x;
    }
}

template <int ROWS, bool KMAJOR>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const f32x4 (&r)[ROWS * BK / 1024], int tid) {
    constexpr int NV = ROWS * BK / 1024;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int idx = v * 256 + tid;
        if (!KMAJOR) *reinterpret_cast<f32x4*>(s + (idx / (BK / 4)) * PK + 4 * ((idx % (BK / 4)) ^ swzb(idx / (BK / 4)))) = r[v];
        else *reinterpret_cast<f32x4*>(s + (idx / (ROWS / 4)) * (ROWS + 4) + 4 * (idx % (ROWS / 4))) = r[v];
    }
}

// Transposing forms for an operand stored contraction-major (the weight gradient's dZ[k][m], H[k][n]; round 4).  The MFMA loop reads an
// operand quad -- four consecutive contraction indices of one row -- with ONE ds_read_b128 when the tile is staged [row][BK + 4]; from a
// [BK][rows + 4] tile it takes four ds_read_b32 with two-way bank conflicts between the k-groups: 40 LDS instructions per 64 MFMAs, which
// is what held the weight-gradient kernel at 63 % of its MFMA time with or without its global loads (scripts/r4_wide_diag.sh).  A thread
// takes ONE item = 4 rows x 4 contraction indices: four float4 loads along the rows (the coalesced pattern of the plain form: 512-byte
// pieces of four consecutive source rows), transposed in registers (free), four ds_write_b128.  The quad column carries the swzb swap of
// the row (conflict-free reads under the hardware's lane groups, above); eight consecutive writers -- rows four apart -- then land
// two by two on the same banks: 16 LDS cycles per ds_write_b128 against the 13 its operand transfer takes anyway (the first layout of
// this round spread the writers over all 32 banks with the row's bits 3..5 and paid two-way conflicts on every READ instead).
// (First attempt, measured and dropped: four dword loads per item down the contraction axis -- 32 instead of 8 vector-memory
// instructions per thread and chunk with their address arithmetic: the kernel went from 0.355 to 0.447 ms per 1 M rows.)
__device__ __forceinline__ int tr_quad(int row, int c4) { return 4 * (c4 ^ swzb(row)); }
template <int ROWS>
__device__ __forceinline__ void load_tile_tr(const float* __restrict__ P, int ld, int row0, int k0, int rows_lim, int k_lim, bool vec_ok, f32x4 (&r)[4], int tid) {
    static_assert(ROWS * BK == 16 * 256, "one item per thread");
    const int row = row0 + 4 * (tid % (ROWS / 4)), k = k0 + 4 * (tid / (ROWS / 4));
    const float* p = P + (size_t)k * ld + row;
    if (vec_ok && row0 + ROWS <= rows_lim && k0 + BK <= k_lim) {
        // the whole tile lies inside the operand (workgroup-uniform; every chunk but the last of a contraction range): four plain float4
        // loads, no per-element tests -- those and their address selects were a fifth of the kernel's vector instructions
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = *reinterpret_cast<const f32x4*>(p + (size_t)e * ld);
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f32x4 x = {0.0f, 0.0f, 0.0f, 0.0f};
        if (k + e < k_lim) {
            const float* pe = p + (size_t)e * ld;
            if (vec_ok && row + 3 < rows_lim) x = *reinterpret_cast<const f32x4*>(pe);
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (row + c < rows_lim) x[c] = pe[c];
            }
        }
        r[e] = x;                      // r[e][c] = source[k + e][row + c]
    }
}
template <int ROWS>
__device__ __forceinline__ void store_tile_tr(float* __restrict__ s, const f32x4 (&r)[4], int tid) {
    const int row = 4 * (tid % (ROWS / 4)), c4 = tid / (ROWS / 4);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        *reinterpret_cast<f32x4*>(s + (row + c) * PK + tr_quad(row + c, c4)) = f32x4{r[0][c], r[1][c], r[2][c], r[3][c]};
}

// BN: 64 or 128 output columns per workgroup (128: the row operand of a layer of width <= 128 is read once)
// PREM (round 4, EPI_WGRAD with the recomputed first layer as B operand, BN = 128): the B tile h_0[32 observations][128 columns] of a chunk is
// made by MFMAs -- D[m = observation][n = column] = X_0 Wt_0^T, four steps per 16 x 16 block, sixteen blocks a chunk, four of them per wave --
// whose result layout, lane (column j, q), element t = h_0[observation 4 q + t][column j], IS a quad of the staged tile [column][k]: one
// ds_write_b128 per block and no per-thread dot products (the vector form cost 2.4 vector instructions per MFMA of the kernel).
template <bool AK, bool BK_, int EPI, int BN, bool HEADW = false, bool PREM = false>
__global__ __launch_bounds__(256) void wide_gemm_kernel(const GemmArgs G) {
    static_assert(!HEADW || (EPI == EPI_WGRAD && AK && BK_), "the fused head backward feeds the weight gradient's transposing loader");
    static_assert(!PREM || (EPI == EPI_WGRAD && AK && BK_ && BN == 128 && !HEADW), "the MFMA-made first layer is the B tile of the 128-column weight gradient");
    if (G.stop_flag != nullptr && *G.stop_flag != 0) return;      // a previous step hit a non-finite gradient norm
    constexpr int PMB = BN + 4, NB = BN / 16;
    constexpr int SB = (BN * PK > BK * PMB) ? BN * PK : BK * PMB;
    constexpr bool TR = (EPI == EPI_WGRAD) && AK && BK_;      // contraction-major operands staged transposed (load_tile_tr): b128 operand reads
    constexpr bool TRB = TR && BN == 128;                     // (a 64-column B tile is half an item per thread: it keeps the [BK][rows + 4] form)
    __shared__ __attribute__((aligned(16))) float sA[2][SA];
    __shared__ __attribute__((aligned(16))) float sB[2][SB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    int m0 = blockIdx.x * BM, Mlim = G.M;
    const int n0 = blockIdx.y * BN;
    const float* __restrict__ Bp = G.B;
    const float* __restrict__ biasp = G.bias;
    if (EPI != EPI_WGRAD && G.tiles != nullptr) {          // (workgroup-uniform)
        const int g = G.tiles[2 * blockIdx.x];
        m0 = G.tiles[2 * blockIdx.x + 1];
        Mlim = G.seg[g + 1];
        Bp = G.B + (size_t)g * (size_t)G.bgs;
        if (biasp != nullptr) biasp += (size_t)g * (size_t)G.biasgs;
    }
    int kbeg = 0, kend = G.K;
    if (EPI == EPI_WGRAD) {
        if (G.seg != nullptr) { kbeg = G.seg[blockIdx.z]; kend = G.seg[blockIdx.z + 1]; }
        else {
            kbeg = (int)min((long long)blockIdx.z * G.ksplit, (long long)G.K);
            kend = (int)min((long long)kbeg + G.ksplit, (long long)G.K);
        }
    }
    const bool vecA = (G.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(G.A) & 15) == 0);
    const bool vecB = (G.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(Bp) & 15) == 0);

    f32x4 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    float bsum = 0.0f;              // EPI_WGRAD: bias gradient of output unit m0 + tid = column sum of the A operand (dZ)

    f32x4 ra[BM * BK / 1024], rb[BN * BK / 1024];
    // PREM: this wave's four blocks of the B tile are columns 16 (2 wv + nb) .. + 15, nb = 0, 1, for both 16-observation halves of the chunk;
    // Wt_0 of those columns as B operand (lane (column j, q), step t = Wt_0[column][4 q + t]) and their biases stay in registers
    f32x4 w0b[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}}, xk[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
    float b0n[2] = {0.0f, 0.0f};
    if (PREM) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int n = n0 + 16 * (2 * wv + nb) + j;
            b0n[nb] = (n < G.pre.N0) ? G.pre.b0[n] : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) w0b[nb][t] = (n < G.pre.N0 && 4 * q + t < G.pre.K0) ? G.pre.W0[(size_t)n * G.pre.K0 + 4 * q + t] : 0.0f;
        }
    }
    // the chunk's metadata rows as A operand: lane (observation j of half kb, q) holds X_0[k][4 q .. 4 q + 3] (rows past the range: zeros --
    // their dZ rows are zero in the A tile, so what the bias makes of them here multiplies nothing)
    auto load_x0 = [&](int kk0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int k = kk0 + 16 * kb + j;
            xk[kb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (k < kend && 4 * q < G.pre.ldx0) xk[kb] = *reinterpret_cast<const f32x4*>(G.pre.X0 + (size_t)k * G.pre.ldx0 + 4 * q);
        }
    };
    // HEADW: the thread's four output units are fixed (as `pcol` on the other operand): the head's two weights of each in registers, the
    // head's weight-gradient sums of those columns over the thread's observations, dL/d(loc, sigma) and dsigma/draw of the item's four rows
    float hw0[4], hw1[4], ha0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ha1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, hsb0 = 0.0f, hsb1 = 0.0f;
    f32x4 rg0 = {0.0f, 0.0f, 0.0f, 0.0f}, rg1 = rg0, rgd = rg0;
    if (HEADW) {
        const int hcol = m0 + 4 * (tid % (BM / 4));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            hw0[c] = (hcol + c < G.M) ? G.hd_W[hcol + c] : 0.0f;
            hw1[c] = (hcol + c < G.M) ? G.hd_W[G.M + hcol + c] : 0.0f;
        }
    }
    auto load_a = [&](int kk0, f32x4 (&r)[BM * BK / 1024]) {
        if constexpr (TR) load_tile_tr<BM>(G.A, G.lda, m0, kk0, Mlim, kend, vecA, r, tid);
        else load_tile<BM, AK>(G.A, G.lda, m0, kk0, Mlim, kend, vecA, r, tid);
        if constexpr (HEADW) {
            const int k = kk0 + 4 * (tid / (BM / 4));                 // the item's four observations (k is a multiple of four: aligned quads)
            if (k + 3 < kend) {
                rg0 = *reinterpret_cast<const f32x4*>(G.hd_dO + 2 * (size_t)k);
                rg1 = *reinterpret_cast<const f32x4*>(G.hd_dO + 2 * (size_t)k + 4);
                rgd = *reinterpret_cast<const f32x4*>(G.hd_dsd + k);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = k + e < kend;
                    const float a0 = in ? G.hd_dO[2 * (size_t)(k + e)] : 0.0f, a1 = in ? G.hd_dO[2 * (size_t)(k + e) + 1] : 0.0f;
                    if (e < 2) { rg0[2 * e] = a0; rg0[2 * e + 1] = a1; } else { rg1[2 * e - 4] = a0; rg1[2 * e - 3] = a1; }
                    rgd[e] = in ? G.hd_dsd[k + e] : 0.0f;
                }
            }
        }
    };
    // HEADW: h_L in `ra` -> dZ_L, in place, just before the tile is staged (the loads were issued a chunk of MFMAs earlier)
    auto head_xform = [&]() {
        if constexpr (HEADW) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g0 = (e < 2) ? rg0[2 * e] : rg1[2 * e - 4];
                const float g1 = ((e < 2) ? rg0[2 * e + 1] : rg1[2 * e - 3]) * rgd[e];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float h = ra[e][c];
                    ha0[c] = fmaf(h, g0, ha0[c]);
                    ha1[c] = fmaf(h, g1, ha1[c]);
                    const float dh = fmaf(g1, hw1[c], g0 * hw0[c]);
                    ra[e][c] = (h > 0.0f) ? dh : G.leak * dh;
                }
                hsb0 += g0; hsb1 += g1;
            }
        }
    };
    auto load_b = [&](int kk0, f32x4 (&r)[BN * BK / 1024]) {
        if constexpr (PREM) { load_x0(kk0); return; }
        if constexpr (TRB) load_tile_tr<BN>(Bp, G.ldb, n0, kk0, G.N, kend, vecB, r, tid);
        else load_tile<BN, BK_>(Bp, G.ldb, n0, kk0, G.N, kend, vecB, r, tid);
    };
    auto stage = [&](float* sa, float* sb) {
        if constexpr (TR) store_tile_tr<BM>(sa, ra, tid); else store_tile<BM, AK>(sa, ra, tid);
        if constexpr (PREM) {
            // h_0 = LeakyReLU(X_0 Wt_0^T + b_0) of this wave's four blocks, in the forward kernels' contraction order (the bias after the steps)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int t = 0; t < 4; ++t) z = mfma4(xk[kb][t], w0b[nb][t], z);
#pragma unroll
                    for (int t = 0; t < 4; ++t) { const float v = z[t] + b0n[nb]; z[t] = fmaxf(v, G.leak * v); }
                    const int col = 16 * (2 * wv + nb) + j;
                    *reinterpret_cast<f32x4*>(sb + col * PK + tr_quad(col, 4 * kb + q)) = z;
                }
        } else if constexpr (TRB) store_tile_tr<BN>(sb, rb, tid); else store_tile<BN, BK_>(sb, rb, tid);
    };
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        load_a(kbeg, ra);
        load_b(kbeg, rb);
        head_xform();
        stage(sA[0], sB[0]);
    }
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
        const int cur = it & 1;
        if (it + 1 < nk) {           // the next chunk's global loads fly under this chunk's MFMAs
            load_a(kbeg + (it + 1) * BK, ra);
            load_b(kbeg + (it + 1) * BK, rb);
        }
        const float* a_s = sA[cur];
        const float* b_s = sB[cur];
        // MFMA step t of 16-deep sub-chunk kc contracts k = 16 kc + 4 q + t (q = lane >> 4), the same map for both operands.  The operand
        // quads of BOTH sub-chunks are requested before the first MFMA (round 4: the kernel holds two waves per SIMD -- LDS-limited --
        // and 256 registers a wave: the second sub-chunk's LDS round trip hides under the first one's 64 MFMAs instead of standing
        // between them)
        f32x4 af[BK / 16][2], bf[BK / 16][NB];
#pragma unroll
        for (int kc = 0; kc < BK / 16; ++kc) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int row = 32 * wv + 16 * a + j;
                if (TR) af[kc][a] = *reinterpret_cast<const f32x4*>(a_s + row * PK + tr_quad(row, 4 * kc + q));
                else if (!AK) af[kc][a] = *reinterpret_cast<const f32x4*>(a_s + row * PK + 16 * kc + 4 * (q ^ swzb(j)));
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) af[kc][a][t] = a_s[(16 * kc + 4 * q + t) * PMA + row];
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int col = 16 * b + j;
                if (TRB) bf[kc][b] = *reinterpret_cast<const f32x4*>(b_s + col * PK + tr_quad(col, 4 * kc + q));
                else if (!BK_) bf[kc][b] = *reinterpret_cast<const f32x4*>(b_s + col * PK + 16 * kc + 4 * (q ^ swzb(j)));
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) bf[kc][b][t] = b_s[(16 * kc + 4 * q + t) * PMB + col];
                }
            }
        }
#pragma unroll
        for (int kc = 0; kc < BK / 16; ++kc)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[a][b] = mfma4(af[kc][a][t], bf[kc][b][t], acc[a][b]);
        if (EPI == EPI_WGRAD && AK && blockIdx.y == 0 && tid < BM) {
            // (thread m sums its output unit's dZ over the chunk's 32 observations: a row of the transposed tile, a column of the other)
            if (TR) {
#pragma unroll
                for (int k4 = 0; k4 < BK / 4; ++k4) {
                    const f32x4 z4 = *reinterpret_cast<const f32x4*>(a_s + tid * PK + 4 * k4);
                    bsum += (z4[0] + z4[1]) + (z4[2] + z4[3]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < BK; ++k) bsum += a_s[k * PMA + tid];
            }
        }
        // (one LDS copy with two barriers per chunk -- 36.9 KB, three workgroups per CU instead of two -- measured equal: 5.19 vs 5.19 ms per step)
        if (it + 1 < nk) { head_xform(); stage(sA[cur ^ 1], sB[cur ^ 1]); }
        __syncthreads();
    }
    if constexpr (HEADW) {
        // the head's weight gradient of this workgroup's observations: eight k-groups of threads hold sums for the same columns; LDS is free
        // after the loop's last barrier.  Layout of the partial: [dWo (2 x M) | dbo (2)]
        float* sh = &sA[0][0];
        const int kg = tid / (BM / 4), c0 = 4 * (tid % (BM / 4));
#pragma unroll
        for (int c = 0; c < 4; ++c) { sh[kg * 2 * BM + c0 + c] = ha0[c]; sh[kg * 2 * BM + BM + c0 + c] = ha1[c]; }
        if (tid % (BM / 4) == 0) { sh[16 * BM + 2 * kg] = hsb0; sh[16 * BM + 2 * kg + 1] = hsb1; }
        __syncthreads();
        float* part = G.hd_part + (size_t)blockIdx.z * (2 * (size_t)G.M + 2);
        for (int i = tid; i < 2 * BM + 2; i += 256) {
            float t = 0.0f;
            if (i < 2 * BM) {
                const int r = i / BM, c = i - r * BM;
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8) t += sh[g8 * 2 * BM + i];
                if (c < G.M) part[(size_t)r * G.M + c] = t;
            } else {
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8) t += sh[16 * BM + 2 * g8 + (i - 2 * BM)];
                part[2 * (size_t)G.M + (i - 2 * BM)] = t;
            }
        }
    }

    // epilogue: accumulator (a, b), element t of lane (j, q) = C[m0 + 32 wv + 16 a + 4 q + t][n0 + 16 b + j]
    if (EPI == EPI_WGRAD && blockIdx.y == 0 && tid < BM && m0 + tid < G.M) {
        if (G.Cb != nullptr) G.Cb[(size_t)blockIdx.z * (size_t)G.bstride + m0 + tid] = bsum;
        else (G.C + (size_t)blockIdx.z * G.pstride)[(size_t)G.M * G.n_in + m0 + tid] = bsum;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int n = n0 + 16 * b + j;
            if (n >= G.N) continue;
            const float bias = (EPI == EPI_BIAS_LRELU) ? biasp[n] : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = m0 + 32 * wv + 16 * a + 4 * q + t;
                if (m >= Mlim) continue;
                float v = acc[a][b][t];
                if (EPI == EPI_BIAS_LRELU) {
                    v += bias;
                    if (G.act) v = fmaxf(v, G.leak * v);
                    G.C[(size_t)m * G.ldc + n] = v;
                } else if (EPI == EPI_DLRELU) {
                    if (G.H != nullptr) v = (G.H[(size_t)m * G.ldh + n] > 0.0f) ? v : G.leak * v;
                    G.C[(size_t)m * G.ldc + n] = v;
                } else {
                    // m = output unit, n = input feature; flat W^T layout [Wt (M x n_in) | b (M)]
                    float* part = G.C + (size_t)blockIdx.z * G.pstride;
                    part[(size_t)m * G.n_in + n] = v;
                }
            }
        }
}

// Dense(2) head, forward: loc = h . Wo[0] + bo[0], sigma = bijector(h . Wo[1] + bo[1]) + eps   (nn.py:22-25, 84-87); a wave per 64 rows,
// lane = row would stride the loads by ld: instead 16 lanes share a row (coalesced 64-byte pieces), 4 rows per wave pass
__global__ __launch_bounds__(256) void wide_head_forward_kernel(const float* __restrict__ H, int ldh, const float* __restrict__ Wo, int n, int w,
                                                                int bij_kind, float eps, float* __restrict__ loc_out, float* __restrict__ sig_out,
                                                                const int* stop_flag) {
    if (stop_flag != nullptr && *stop_flag != 0) return;
    const int sub = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    float o0 = 0.0f, o1 = 0.0f;
    if (row < n) {
        const float* h = H + (size_t)row * ldh;
        for (int k = sub; k < w; k += 16) {
            const float x = h[k];
            o0 = fmaf(x, Wo[k], o0);
            o1 = fmaf(x, Wo[w + k], o1);
        }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { o0 += __shfl_xor(o0, off); o1 += __shfl_xor(o1, off); }
    if (row < n && sub == 0) {
        float d;
        loc_out[row] = o0 + Wo[2 * w];
        sig_out[row] = cl_scale_bij(o1 + Wo[2 * w + 1], bij_kind, eps, &d);
    }
}

// Dense(2) head, backward, from dL/d(loc, sigma) per row: dZ_L = (g Wo) * LeakyReLU'(h_L) per row, and this workgroup's partial
// sums of dWo (2 x w), dbo (2) in the head's flat layout [Wo (2 x w) | bo (2)].  32 lanes share a row (one float4 each per 128
// columns: whole 512-byte pieces), 8 rows per workgroup pass; the lane's columns of dWo accumulate in registers.
// NP: 128-column passes an instance holds (width <= 128 NP)
template <int NP>
__global__ __launch_bounds__(256) void wide_head_backward_kernel(const float* __restrict__ H, int ldh, const float* __restrict__ Wo, const float* __restrict__ dO,
                                                                 int n, int w, int bij_kind, float eps, float leak, int rows_per_block,
                                                                 float* __restrict__ dZ, int lddz, float* __restrict__ partials, const int* stop_flag) {
    if (stop_flag != nullptr && *stop_flag != 0) return;
    extern __shared__ float sh[];                 // [8 row slots][2 w + 2]
    const int sub = threadIdx.x & 31, slot = threadIdx.x >> 5;
    const int P = 2 * w + 2;
    f32x4 w0[NP], w1[NP], a0[NP], a1[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int c = 128 * p + 4 * sub;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            w0[p][e] = (c + e < w) ? Wo[c + e] : 0.0f;
            w1[p][e] = (c + e < w) ? Wo[w + c + e] : 0.0f;
            a0[p][e] = 0.0f; a1[p][e] = 0.0f;
        }
    }
    const float bo1 = Wo[2 * w + 1];
    float sb0 = 0.0f, sb1 = 0.0f;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(r0 + rows_per_block, (long long)n);
    for (long long row = r0 + slot; row < r1; row += 8) {
        f32x4 h[NP];
        float o1 = 0.0f;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = 128 * p + 4 * sub;
            h[p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (c < ldh) h[p] = *reinterpret_cast<const f32x4*>(H + (size_t)row * ldh + c);      // (ld is a multiple of four, the padding columns are zero)
#pragma unroll
            for (int e = 0; e < 4; ++e) o1 = fmaf(h[p][e], w1[p][e], o1);
        }
        // raw sigma again (the forward pass kept sigma only): d sigma / d raw
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) o1 += __shfl_xor(o1, off);
        float dsig_draw;
        (void)cl_scale_bij(o1 + bo1, bij_kind, eps, &dsig_draw);
        const float g0 = dO[2 * (size_t)row], g1 = dO[2 * (size_t)row + 1] * dsig_draw;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = 128 * p + 4 * sub;
            if (c < lddz && c < ((w + 3) & ~3)) {
                f32x4 dz;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dh = g0 * w0[p][e] + g1 * w1[p][e];
                    dz[e] = (h[p][e] > 0.0f) ? dh : leak * dh;
                }
                *reinterpret_cast<f32x4*>(dZ + (size_t)row * lddz + c) = dz;
            }
            a0[p] += h[p] * g0;
            a1[p] += h[p] * g1;
        }
        sb0 += g0; sb1 += g1;                     // (the same in all 32 lanes of the row)
    }
    float* mine = sh + slot * P;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int c = 128 * p + 4 * sub;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < w) { mine[c + e] = a0[p][e]; mine[w + c + e] = a1[p][e]; }
    }
    if (sub == 0) { mine[2 * w] = sb0; mine[2 * w + 1] = sb1; }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += 256) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sh[k * P + i];
        partials[(size_t)blockIdx.x * P + i] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// streaming form: Y^T[N][rows] = W'[N][K] X^T[K][rows], W' resident in LDS (N, K <= 128)
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SMAX = 128;            // largest layer side the streaming kernel holds
constexpr int SKP = SMAX + 4;        // pitch of the weight image [N][K]
struct StreamArgs {
    const float* X; int ldx;         // [n][K] rows (contraction-contiguous)
    const float* W; int ldw;         // WKM = false: W[N][K] at W[o * ldw + k] (forward: Wt[out][in]); true: W[K][N] at W[k * ldw + o] (dgrad: Wt[out][in], N = in)
    float* Y; int ldy;               // [n][N]
    long long n; int N, K;
    const float* bias;               // EPI_BIAS_LRELU
    const float* H; int ldh;         // EPI_DLRELU
    float leak; int act;
    const int* stop_flag;
    // Grouped form (per-image layers, careless/models/scaling/image.py:90-96): the rows are sorted by image, seg[g] .. seg[g+1] are the
    // rows of group g, its weights sit at W + g * wstride and its bias at bias + g * bstride.  A workgroup takes whole groups: it
    // stages the group's weights, its eight waves walk the group's 16-row blocks, then the next group.  seg == NULL: one group = all rows.
    const int* seg; int n_groups;
    long long wstride, bstride;
    // Dense(2) head fused into the forward epilogue of the top layer (EPI_BIAS_LRELU, N <= 128): the rows' activations are in
    // registers there -- loc / sigma come out of the same pass instead of a launch that reads them back (nn.py:22-25, 84-87)
    const float* head_W;             // [2][N] then [2] biases (the head's flat layout) or NULL
    int bij_kind; float eps;
    float* loc_out; float* sig_out;  // [n]
    float* dsd_out;                  // [n] or NULL: d sigma / d raw of the row (the bijector's derivative), for the head's backward pass fused into
                                     // the top layer's dgrad / weight gradient (HEADB instances, cl_wide_dense_wgrad_head)
    const float* dO; const float* dsd;   // HEADB: dL/d(loc, sigma) [n][2] and d sigma / d raw [n]; X = the TOP layer's activations h_L, head_W = the head
    PreArgs pre;                     // EPI_DLRELU: the activations whose sign selects the derivative are the recomputed first layer (H unused)
    // LIK instances (forward with the fused head, rows that are their own slot): the slot likelihood of the call's rows in the same epilogue --
    // sample, predict, log-prob, its gradient back to dz_f / the image scales / dO -- as cl_slot_rows does it (elbo_laue.hip), for the rows'
    // (loc, sigma) sit in registers here and the amplitude-gradient atomics are fire-and-forget under the MFMAs of the next block
    struct Lik {
        const int* refl_id; const int* image_id; const float* iobs; const float* sig;      // per row of the call
        const long long* row_index; long long obs_offset;                                   // noise key of row i: row_index[i] or obs_offset + i
        const float* img; int use_img; const float* z_f; int S;
        int lik_kind; float dof, lik_const, shift, w_ll;
        unsigned long long seed; unsigned step;
        float* dz_f; float* d_img; float* dO; double* scalars;
    } lik;
    float* wg0_part;                 // WG0 instances: [gridDim.x][N K0 + N] partial sums of the FIRST layer's weight gradient (flat W^T layout)
};

// The kernel's argument block again, behind an opaque pointer (as kernargs_again of cl_kernels.h): the likelihood's two dozen scalars are
// read where the epilogue uses them -- kept in scalar registers for the whole launch they spill into vector-register lanes (47 of them) and
// every use in the block loop pays a v_readlane.  Valid in kernels whose single parameter is a StreamArgs by value.
typedef const __attribute__((address_space(4))) StreamArgs* sargs_p;
__device__ __forceinline__ sargs_p sargs_again() {
    sargs_p p = (sargs_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// NAT: 16-column blocks of the output an instance holds (4: N <= 64, 8: N <= 128); GRP: the grouped form (its own instances: the
// group loop costs the plain ones registers)
template <bool WKM, int EPI, int NAT, bool GRP, bool PRE = false>
__global__ __launch_bounds__(512) void wide_stream_kernel(const StreamArgs S) {
    if (S.stop_flag != nullptr && *S.stop_flag != 0) return;
    extern __shared__ __attribute__((aligned(16))) float sW[];      // [16 NA][SKP], zero outside N x K
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int N = S.N, K = S.K;
    const int NA = (N + 15) >> 4, KC = (K + 15) >> 4;               // 16-column blocks of the output, 16-deep chunks of the contraction
    const int tot = 16 * NA * SKP;
    float* const sBias = sW + tot;                                   // [16 NA] (forward), zero past N
    float* const sHead = sBias + 16 * NA;                            // [2][16 NA] (forward with the fused head), zero past N
    float* const sW0 = sHead + 32 * NA;                              // [16 NA][S0P], [16 NA]: the recomputed first layer (dgrad mask), zero outside N x K0
    float* const sB0 = sW0 + 16 * NA * S0P;
    constexpr bool premask = PRE && EPI == EPI_DLRELU && !GRP;        // (its own instance: the recomputation costs the plain one registers)
    const bool vec = (S.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(S.X) & 15) == 0);
    const float* const wrow = sW + j * SKP + 4 * (q ^ swzb(j));
    constexpr bool grouped = GRP;
    const int ngroups = grouped ? S.n_groups : 1;
  for (int grp = grouped ? (int)blockIdx.x : 0; grp < ngroups; grp += grouped ? (int)gridDim.x : 1) {
    const long long row0 = grouped ? (long long)S.seg[grp] : 0, rend = grouped ? (long long)S.seg[grp + 1] : S.n;      // the group's rows
    if (rend <= row0) continue;                                      // (workgroup-uniform)
    const float* __restrict__ Wg = S.W + (size_t)grp * (size_t)S.wstride;
    if (grouped && grp != (int)blockIdx.x) __syncthreads();          // every wave is done with the previous group's weights
    // (eight independent loads in flight per thread: a plain loop waits for every load in turn -- a fixed cost per launch; consecutive
    //  threads walk the contiguous axis of the stored weights)
    for (int base = 0; base < tot; base += 8 * 512) {
        float v[8];
        int at[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 512 + tid;
            int o, k;
            if (!WKM) { o = idx / SKP; k = idx - o * SKP; }
            else { k = idx / (16 * NA); o = idx - k * (16 * NA); }
            at[u] = (idx < tot && k < SKP) ? img_at(o, k, SKP, SMAX) : -1;
            v[u] = 0.0f;
            if (idx < tot && o < N && k < K) v[u] = WKM ? Wg[(size_t)k * S.ldw + o] : Wg[(size_t)o * S.ldw + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (at[u] >= 0) sW[at[u]] = v[u];
    }
    if (EPI == EPI_BIAS_LRELU && tid < 16 * NA) sBias[tid] = (tid < N) ? S.bias[(size_t)grp * (size_t)S.bstride + tid] : 0.0f;
    if (premask) {
        for (int idx = tid; idx < 16 * NA * S0P; idx += 512) {
            const int o = idx / S0P, k = idx - o * S0P;
            sW0[img_at(o, k, S0P, K0MAX)] = (o < N && k < S.pre.K0) ? S.pre.W0[(size_t)o * S.pre.K0 + k] : 0.0f;
        }
        if (tid < 16 * NA) sB0[tid] = (tid < N) ? S.pre.b0[tid] : 0.0f;
    }
    if (EPI == EPI_BIAS_LRELU && S.head_W != nullptr && tid < 32 * NA) {
        const int r = tid / (16 * NA), c = tid - r * 16 * NA;
        sHead[tid] = (c < N) ? S.head_W[r * N + c] : 0.0f;
    }
    __syncthreads();
    const long long nblk = (rend - row0 + 15) >> 4;
    // the rows' operand: lane (row j, k-group q) holds X[row][16 kc + 4 q .. + 3] of chunk kc (MFMA step t contracts k = 16 kc + 4 q + t)
    auto load_x = [&](long long b, int kc) -> f32x4 {
        const long long row = row0 + b * 16 + j;
        f32x4 x = {0.0f, 0.0f, 0.0f, 0.0f};
        if (CL_WIDE_DIAG & 1) return f32x4{1.0f, 0.5f, -0.25f, 0.125f};
        if (row < rend) {
            const int k = 16 * kc + 4 * q;
            const float* p = S.X + (size_t)row * S.ldx + k;
            if (vec && k + 3 < K) x = *reinterpret_cast<const f32x4*>(p);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (k + e < K) x[e] = p[e];
            }
        }
        return x;
    };
    long long blk = grouped ? (long long)wv : (long long)blockIdx.x * 8 + wv;
    const long long bstep = grouped ? 8 : (long long)gridDim.x * 8;
    f32x4 xc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (blk < nblk) xc = load_x(blk, 0);
    for (; blk < nblk; blk += bstep) {
        f32x4 acc[NAT];
#pragma unroll
        for (int a = 0; a < NAT; ++a) acc[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int kc = 0; kc < KC; ++kc) {
            // the next chunk of the rows' operand (or the first chunk of the wave's next block) flies under this chunk's MFMAs
            f32x4 xnx = {0.0f, 0.0f, 0.0f, 0.0f};
            if (kc + 1 < KC) xnx = load_x(blk, kc + 1);
            else if (blk + bstep < nblk) xnx = load_x(blk + bstep, 0);
#pragma unroll
            for (int a = 0; a < NAT; ++a) {
                if (a < NA) {
                    const f32x4 wf = *reinterpret_cast<const f32x4*>(wrow + 16 * a * SKP + 16 * kc);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[a] = mfma4(wf[t], xc[t], acc[a]);
                }
            }
            xc = xnx;
        }
        // accumulator a, element t of lane (j, q) = Y[row = 16 blk + j][column 16 a + 4 q + t]: four consecutive columns of the lane's row
        const long long row = row0 + blk * 16 + j;
        f32x4 hpre[NAT];                         // dgrad with the recomputed first layer: its pre-activations of this lane's row and columns
        if (premask) {
            // as the forward pass made them: one 16-deep chunk of metadata, four MFMAs per block.  ALL lanes take part (an MFMA reads its
            // A operand -- the weights of feature 16 a + j -- from every lane): rows past the end read the last row
            const long long rr = row < rend ? row : rend - 1;
            f32x4 x0 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (4 * q < S.pre.ldx0) x0 = *reinterpret_cast<const f32x4*>(S.pre.X0 + (size_t)rr * S.pre.ldx0 + 4 * q);
#pragma unroll
            for (int a = 0; a < NAT; ++a) {
                hpre[a] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                if (a < NA) {
                    const f32x4 wf = *reinterpret_cast<const f32x4*>(sW0 + (16 * a + j) * S0P + 4 * (q ^ swzb(j)));
                    f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int t = 0; t < 4; ++t) z = mfma4(wf[t], x0[t], z);
                    hpre[a] = z + *reinterpret_cast<const f32x4*>(sB0 + 16 * a + 4 * q);      // (its sign is the sign of LeakyReLU of it)
                }
            }
        }
        if (row < rend) {
            const bool vecy = (S.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(S.Y) & 15) == 0);
            const bool vech = (S.ldh % 4 == 0) && ((reinterpret_cast<uintptr_t>(S.H) & 15) == 0);
            f32x4 hm[NAT];                           // dgrad: the activations whose sign selects the derivative, all requested before the first use
            if (premask) {
#pragma unroll
                for (int a = 0; a < NAT; ++a) hm[a] = hpre[a];
            } else if (EPI == EPI_DLRELU && S.H != nullptr) {
#pragma unroll
                for (int a = 0; a < NAT; ++a) {
                    const int c = 16 * a + 4 * q;
                    hm[a] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                    if (a < NA && c < N) {
                        const float* hp = S.H + (size_t)row * S.ldh + c;
                        if (vech && c + 3 < S.ldh) hm[a] = *reinterpret_cast<const f32x4*>(hp);
                        else {
#pragma unroll
                            for (int t = 0; t < 4; ++t) if (c + t < N) hm[a][t] = hp[t];
                        }
                    }
                }
            }
            float ho0 = 0.0f, ho1 = 0.0f;            // fused head: this lane's share of the row's two dot products
#pragma unroll
            for (int a = 0; a < NAT; ++a) {
                const int c = 16 * a + 4 * q;
                if (a < NA && c < N) {
                    f32x4 v = acc[a];
                    if (EPI == EPI_BIAS_LRELU) {
                        v += *reinterpret_cast<const f32x4*>(sBias + c);
                        if (S.act) {
#pragma unroll
                            for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], S.leak * v[t]);
                        }
                        if (S.head_W != nullptr) {      // (zero weights past N: padding columns add nothing)
                            const f32x4 h0 = *reinterpret_cast<const f32x4*>(sHead + c), h1 = *reinterpret_cast<const f32x4*>(sHead + 16 * NA + c);
#pragma unroll
                            for (int t = 0; t < 4; ++t) { ho0 = fmaf(v[t], h0[t], ho0); ho1 = fmaf(v[t], h1[t], ho1); }
                        }
                    } else if (S.H != nullptr || premask) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = (hm[a][t] > 0.0f) ? v[t] : S.leak * v[t];
                    }
                    float* y = S.Y + (size_t)row * S.ldy + c;
                    if ((CL_WIDE_DIAG & 2) && v[0] != 12345.0f) continue;
                    if (c + 3 < N && vecy) *reinterpret_cast<f32x4*>(y) = v;
                    else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (c + t < N) y[t] = v[t];
                    }
                }
            }
            if (EPI == EPI_BIAS_LRELU && S.head_W != nullptr) {
                // the four lanes (q = 0 .. 3) of a row add up their shares; all 64 lanes are inside this branch or none of a row's is
                // (row < rend depends on j only) -- the shuffles stay among lanes of the same row
                ho0 += __shfl_xor(ho0, 16); ho1 += __shfl_xor(ho1, 16);
                ho0 += __shfl_xor(ho0, 32); ho1 += __shfl_xor(ho1, 32);
                if (q == 0) {
                    float d;
                    S.loc_out[row] = ho0 + S.head_W[2 * N];
                    S.sig_out[row] = cl_scale_bij(ho1 + S.head_W[2 * N + 1], S.bij_kind, S.eps, &d);
                    if (S.dsd_out != nullptr) S.dsd_out[row] = d;
                }
            }
        }
    }
  }
}

// The first two Dense layers in one launch: h_0 = LeakyReLU(X_0 Wt_0^T + b_0) of a wave's 16 rows comes out of four MFMAs per 16-column
// block IN the B-operand layout of the second layer (accumulator a, element t of lane (j, q) = h_0[row j][16 a + 4 q + t] = operand
// chunk a, element t: the property the fused kernels of elbo_mlp.hip chain their layers with) and never leaves the registers.
// Optionally the Dense(2) head in the epilogue (a two-layer scaler).  N0, N1 <= 16 NAT.
struct Stream2Args {
    PreArgs pre;
    const float* W1; const float* b1; int N1;      // Wt_1[N1][N0], b_1[N1]
    float* Y; int ldy; long long n; float leak;
    const float* head_W; int bij_kind; float eps; float* loc_out; float* sig_out;
    const int* stop_flag;
};

// NA: 16-column blocks of BOTH layers' outputs (the hidden widths of a scaler are equal), compile-time: the loops below are straight-line code
// (run-time block counts inside the unrolled loops cost ~1000 spilled registers)
template <int NA>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 8)))      // (two 8-wave workgroups per CU, as the one-layer kernel: <= 128 registers)
void wide_stream2_kernel(const Stream2Args S) {
    if (S.stop_flag != nullptr && *S.stop_flag != 0) return;
    extern __shared__ __attribute__((aligned(16))) float sm2[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int N0 = S.pre.N0, N1 = S.N1, K0 = S.pre.K0;
    constexpr int NA0 = NA, NA1 = NA, NAT = NA;
    float* const sW1 = sm2;                                   // [16 NA1][SKP], zero outside N1 x N0
    float* const sB1 = sW1 + 16 * NA1 * SKP;                  // [16 NA1]
    float* const sHead = sB1 + 16 * NA1;                      // [2][16 NA1]
    float* const sW0 = sHead + 32 * NA1;                      // [16 NA0][S0P], zero outside N0 x K0
    float* const sB0 = sW0 + 16 * NA0 * S0P;                  // [16 NA0]
    for (int base = 0; base < 16 * NA1 * SKP; base += 8 * 512) {       // (eight loads in flight per thread, as in wide_stream_kernel)
        float v[8];
        int at[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 512 + tid;
            const int o = idx / SKP, k = idx - o * SKP;
            at[u] = idx < 16 * NA1 * SKP ? img_at(o, k, SKP, SMAX) : -1;
            v[u] = (idx < 16 * NA1 * SKP && o < N1 && k < N0) ? S.W1[(size_t)o * N0 + k] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (at[u] >= 0) sW1[at[u]] = v[u];
    }
    for (int idx = tid; idx < 16 * NA0 * S0P; idx += 512) {
        const int o = idx / S0P, k = idx - o * S0P;
        sW0[img_at(o, k, S0P, K0MAX)] = (o < N0 && k < K0) ? S.pre.W0[(size_t)o * K0 + k] : 0.0f;
    }
    if (tid < 16 * NA1) sB1[tid] = (tid < N1) ? S.b1[tid] : 0.0f;
    if (tid < 16 * NA0) sB0[tid] = (tid < N0) ? S.pre.b0[tid] : 0.0f;
    if (S.head_W != nullptr && tid < 32 * NA1) {
        const int r = tid / (16 * NA1), c = tid - r * 16 * NA1;
        sHead[tid] = (c < N1) ? S.head_W[r * N1 + c] : 0.0f;
    }
    __syncthreads();
    const long long nblk = (S.n + 15) >> 4;
    auto load_x0 = [&](long long b) -> f32x4 {                // lane (row j, k-group q): X_0[row][4 q .. + 3] (the row is zero-padded to ldx0)
        const long long row = b * 16 + j;
        f32x4 x = {0.0f, 0.0f, 0.0f, 0.0f};
        if (row < S.n && 4 * q < S.pre.ldx0) x = *reinterpret_cast<const f32x4*>(S.pre.X0 + (size_t)row * S.pre.ldx0 + 4 * q);
        return x;
    };
    long long blk = (long long)blockIdx.x * 8 + wv;
    const long long bstep = (long long)gridDim.x * 8;
    f32x4 xc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (blk < nblk) xc = load_x0(blk);
    for (; blk < nblk; blk += bstep) {
        f32x4 xn = {0.0f, 0.0f, 0.0f, 0.0f};
        if (blk + bstep < nblk) xn = load_x0(blk + bstep);
        f32x4 h0[NAT];
#pragma unroll
        for (int a = 0; a < NAT; ++a) {
            const f32x4 wf = *reinterpret_cast<const f32x4*>(sW0 + (16 * a + j) * S0P + 4 * (q ^ swzb(j)));
            f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 4; ++t) z = mfma4(wf[t], xc[t], z);
            z += *reinterpret_cast<const f32x4*>(sB0 + 16 * a + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) h0[a][t] = fmaxf(z[t], S.leak * z[t]);
        }
        f32x4 acc[NAT];
#pragma unroll
        for (int a = 0; a < NAT; ++a) acc[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* const w1row = sW1 + j * SKP + 4 * (q ^ swzb(j));
        f32x4 wf = *reinterpret_cast<const f32x4*>(w1row);
#pragma unroll
        for (int kc = 0; kc < NAT; ++kc) {
#pragma unroll
            for (int a = 0; a < NAT; ++a) {
                // (the next weight quad is requested before this one's MFMAs, as in wide_sq_kernel)
                const int na = (a + 1 < NAT) ? a + 1 : 0, nkc = (a + 1 < NAT) ? kc : kc + 1;
                f32x4 wnx = wf;
                if (nkc < NAT) wnx = *reinterpret_cast<const f32x4*>(w1row + 16 * na * SKP + 16 * nkc);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[a] = mfma4(wf[t], h0[kc][t], acc[a]);
                wf = wnx;
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);       // (unrolled for the register-resident h_0: the fence keeps the weight reads of later chunks from being hoisted)
        }
        const long long row = blk * 16 + j;
        if (row < S.n) {
            const bool vecy = (S.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(S.Y) & 15) == 0);
            float ho0 = 0.0f, ho1 = 0.0f;
#pragma unroll
            for (int a = 0; a < NAT; ++a) {
                const int c = 16 * a + 4 * q;
                if (c < N1) {
                    f32x4 v = acc[a] + *reinterpret_cast<const f32x4*>(sB1 + c);
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], S.leak * v[t]);
                    if (S.head_W != nullptr) {
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(sHead + c), g1 = *reinterpret_cast<const f32x4*>(sHead + 16 * NA1 + c);
#pragma unroll
                        for (int t = 0; t < 4; ++t) { ho0 = fmaf(v[t], g0[t], ho0); ho1 = fmaf(v[t], g1[t], ho1); }
                    }
                    float* y = S.Y + (size_t)row * S.ldy + c;
                    if (c + 3 < N1 && vecy) *reinterpret_cast<f32x4*>(y) = v;
                    else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (c + t < N1) y[t] = v[t];
                    }
                }
            }
            if (S.head_W != nullptr) {
                ho0 += __shfl_xor(ho0, 16); ho1 += __shfl_xor(ho1, 16);
                ho0 += __shfl_xor(ho0, 32); ho1 += __shfl_xor(ho1, 32);
                if (q == 0) {
                    float d;
                    S.loc_out[row] = ho0 + S.head_W[2 * N1];
                    S.sig_out[row] = cl_scale_bij(ho1 + S.head_W[2 * N1 + 1], S.bij_kind, S.eps, &d);
                }
            }
        }
        xc = xn;
    }
}

template <int NA>
int launch_stream2(const Stream2Args& s, hipStream_t st) {
    constexpr int NA0 = NA, NA1 = NA;
    const size_t sm = (size_t)(16 * NA1 * SKP + 3 * 16 * NA1 + 16 * NA0 * (S0P + 1)) * sizeof(float);
    auto kern = wide_stream2_kernel<NA>;
    static std::atomic<size_t> configured{0};
    size_t have = configured.load(std::memory_order_acquire);
    if (have < sm) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        while (have < sm && !configured.compare_exchange_weak(have, sm, std::memory_order_release, std::memory_order_acquire)) {}
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long nblk = (s.n + 15) >> 4;
    long long grid = (nblk + 7) / 8;
    if (grid > 2LL * cus) grid = 2LL * cus;
    if (grid < 1) grid = 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), sm, st, s);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------
// square layers (hidden -> hidden: N and K span the same number NA of 16-column blocks), round 4
// ---------------------------------------------------------------------------------------------------------------------------
// The diagnostic builds of round 4 (scripts/r4_wide_diag.sh, profiles/r4_wide_diag.txt) put numbers on wide_stream_kernel at 128 x 128 and
// 1 M rows: 0.355 ms as shipped, 0.31 ms without its global loads, 0.33 without its stores, 0.28 with neither -- against 0.209 ms of MFMA
// issue.  So a quarter of the time is the memory side NOT overlapping (the next chunk's operand is requested one chunk = 0.4 us of
// MFMAs ahead, less than a loaded HBM round trip) and another quarter is vector / scalar work around the MFMAs (run-time block counts,
// bounds tests, 64-bit address arithmetic per chunk and per store).  This kernel takes both out for the common case:
//   * the block count is a template argument, the block body straight-line code; ld's are multiples of four by contract
//     (cl_wide_ld), padding columns zero: every access is a float4, the only test per access is "inside the row buffer";
//   * the rows' operand of a WHOLE 16-row block sits in NA register quads; a quad is refilled with the NEXT block's chunk right
//     after the MFMAs that read it have been issued (an MFMA reads its operands at issue; the load lands a microsecond later): the
//     prefetch distance is a full block (3.4 us of MFMAs at NA = 8) with no second buffer;
//   * the dgrad mask (the layer's input activations, or the recomputed first layer) is requested at the start of the block.
// Same arithmetic, same order of the contraction as wide_stream_kernel (chunk by chunk, step t inside a chunk): the same bits.
// WG0 (round 4, with PRE on the dgrad of the SECOND layer): the kernel's output dZ_0 = (dZ_1 Wt_1) * LeakyReLU'(h_0) is the first layer's
// pre-activation gradient, whose only use is that layer's weight gradient dWt_0 = dZ_0^T X_0 (the metadata are not trained).  With the
// MFMA operands SWAPPED (A = the rows' operand, B = the weights: the same registers, the same products in the same order) the output
// block comes out as lane (j, q), element t = dZ_0[row 4 q + t][column 16 a + j] -- the B-operand layout of a contraction over the
// block's 16 rows.  Four more MFMAs per 16 columns against A = X_0^T (lane (i, q), step t = X_0[row 4 q + t][i], a row of ones behind
// the metadata for the bias gradient) add the block into per-wave accumulators dWt_0[16 a + j][i = 4 q + t] that live for the whole
// launch: dZ_0 is never stored (512 B per row at width 128) and the separate weight-gradient launch that read it back is gone.
// HEADB (round 4, the dgrad of the TOP layer): its row operand dZ_L = (g Wo) * LeakyReLU'(h_L), g = dL/d(loc, raw sigma) of the row, is a rank-2
// product behind a mask -- made from h_L (the same bytes per row as dZ_L) and the row's two numbers while the chunk is on its way to the
// MFMAs, instead of being written by a launch of its own (cl_wide_head_backward) and read back twice.
template <bool WKM, int EPI, int NA, bool PRE, bool WG0 = false, bool HEADB = false, bool LIK = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 8)))
void wide_sq_kernel(const StreamArgs S) {
    static_assert(!LIK || (EPI == EPI_BIAS_LRELU && !PRE && !WG0 && !HEADB), "the slot likelihood rides on the forward pass of the top layer");
    static_assert(!WG0 || (PRE && EPI == EPI_DLRELU), "the fused first-layer weight gradient rides on the dgrad with the recomputed mask");
    static_assert(!HEADB || (!PRE && EPI == EPI_DLRELU), "the fused head backward feeds the dgrad of the top layer");
    if (S.stop_flag != nullptr && *S.stop_flag != 0) return;
    extern __shared__ __attribute__((aligned(16))) float sWq[];      // [16 NA][SKP], zero outside N x K
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int N = S.N, K = S.K;
    constexpr int tot = 16 * NA * SKP;
    float* const sBias = sWq + tot;                                  // [16 NA]
    float* const sHead = sBias + 16 * NA;                            // [2][16 NA]
    float* const sW0 = sHead + 32 * NA;                              // PRE: [16 NA][S0P], [16 NA]
    float* const sB0 = sW0 + 16 * NA * S0P;
    for (int base = 0; base < tot; base += 8 * 512) {
        float v[8];
        int at[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 512 + tid;
            int o, k;
            if (!WKM) { o = idx / SKP; k = idx - o * SKP; }
            else { k = idx / (16 * NA); o = idx - k * (16 * NA); }
            at[u] = (idx < tot && k < SKP) ? img_at(o, k, SKP, SMAX) : -1;
            v[u] = 0.0f;
            if (idx < tot && o < N && k < K) v[u] = WKM ? S.W[(size_t)k * S.ldw + o] : S.W[(size_t)o * S.ldw + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (at[u] >= 0) sWq[at[u]] = v[u];
    }
    if (EPI == EPI_BIAS_LRELU && tid < 16 * NA) sBias[tid] = (tid < N) ? S.bias[tid] : 0.0f;
    if ((EPI == EPI_BIAS_LRELU || HEADB) && S.head_W != nullptr && tid < 32 * NA) {
        const int r = tid / (16 * NA), c = tid - r * 16 * NA;
        const int Nh = HEADB ? K : N;                                // (the head reads the layer's output: the forward's N, the dgrad's K)
        sHead[tid] = (c < Nh) ? S.head_W[r * Nh + c] : 0.0f;
    }
    if (PRE) {
        for (int idx = tid; idx < 16 * NA * S0P; idx += 512) {
            const int o = idx / S0P, k = idx - o * S0P;
            sW0[img_at(o, k, S0P, K0MAX)] = (o < N && k < S.pre.K0) ? S.pre.W0[(size_t)o * S.pre.K0 + k] : 0.0f;
        }
        if (tid < 16 * NA) sB0[tid] = (tid < N) ? S.pre.b0[tid] : 0.0f;
    }
    __syncthreads();
    const long long nblk = (S.n + 15) >> 4;
    const float* const wrow = sWq + j * SKP + 4 * (q ^ swzb(j));
    const int cq0 = 4 * (q ^ swzb(j));                                // this lane's quad of a first-layer image row
    // per-lane element offsets inside a row: chunk kc of the operand at 16 kc + 4 q, block a of the output / mask at 16 a + 4 q
    const int cq = 4 * q;
    // Addresses = a wave-uniform base (the block's first row: scalar registers, scalar arithmetic) + a 32-bit per-lane element offset (this
    // lane's row inside the block, clamped to the last existing row -- such rows are never stored --, times the pitch, plus its quad): one
    // vector register per operand instead of a 64-bit pointer each, which is what the fused instances spilled (round 4)
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    auto lane_row = [&](long long b) -> int {                        // row j of block b, or the block's last existing row
        const long long left = S.n - b * 16;                         // (uniform)
        return left >= 16 ? j : (j < (int)left ? j : (int)left - 1);
    };
    auto x_base = [&](long long b) -> const float* { return S.X + (size_t)b * 16 * (size_t)S.ldx; };
    long long blk = (long long)blockIdx.x * 8 + wvu;
    const long long bstep = (long long)gridDim.x * 8;
    f32x4 acc0[WG0 ? NA : 1];                    // WG0: dWt_0[16 a + j][4 q + t] of this wave's blocks (column K0 = the bias gradient)
#pragma unroll
    for (int a = 0; a < (WG0 ? NA : 1); ++a) acc0[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int K0 = S.pre.K0;
    // The rows' operand quads are requested HALF a block ahead of their use: quad t = kc + H is refilled behind chunk kc's
    // MFMAs -- with this block's chunk t, or, past the end, with chunk t - NA of the wave's next block -- so a quad is dead between its own
    // chunk and the chunk H later, and only about half of the NA quads are live at any time (a full block of lead kept all of them live: the
    // instances that carry a fusion spilled 18 - 25 registers, and a spill reload waits for every load in flight).  H chunks of MFMAs of four
    // waves are several microseconds: still more than a loaded HBM round trip.
    constexpr int H = (NA + 1) / 2;
    f32x4 xb[NA];
    float gc0 = 0.0f, gc1 = 0.0f;                 // HEADB: g of this lane's row of the current block
    if (blk < nblk) {
        const float* xp = x_base(blk);
        const unsigned xo = (unsigned)lane_row(blk) * (unsigned)S.ldx + cq;
#pragma unroll
        for (int kc = 0; kc < NA; ++kc) xb[kc] = (kc < H && 16 * kc + cq < S.ldx) ? *reinterpret_cast<const f32x4*>(xp + xo + 16 * kc) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (HEADB) {
            const unsigned r = (unsigned)lane_row(blk);
            const float* gp = S.dO + 2 * (size_t)blk * 16;
            gc0 = gp[2 * r];
            gc1 = gp[2 * r + 1] * (S.dsd + (size_t)blk * 16)[r];
        }
    }
    double l_nll = 0.0;                           // LIK: this lane's share of the NLL over all its blocks (fp64 across rows, like slot_rows_kernel)
    for (; blk < nblk; blk += bstep) {
        const long long row = blk * 16 + j;
        const unsigned jc = (unsigned)lane_row(blk);                 // this lane's (clamped) row inside the block
        float l_da = 0.0f;                        // LIK: image-scale term of the row (summed over its lanes in the epilogue)
        bool l_take = false;
        // HEADB: the next block's row numbers are requested two chunks before the end of this block (below)
        float gn0 = 0.0f, gn1 = 0.0f, gnd = 0.0f;
        // the mask of the dgrad epilogue, requested now (the layer's input activations), consumed after the MFMAs
        f32x4 hm[NA];
        if (EPI == EPI_DLRELU && !PRE && S.H != nullptr) {
            const float* hp = S.H + (size_t)blk * 16 * (size_t)S.ldh;
            const unsigned ho = jc * (unsigned)S.ldh + cq;
#pragma unroll
            for (int a = 0; a < NA; ++a) hm[a] = (16 * a + cq < S.ldh) ? *reinterpret_cast<const f32x4*>(hp + ho + 16 * a) : f32x4{1.0f, 1.0f, 1.0f, 1.0f};
        }
        f32x4 x0 = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* x0p = PRE ? S.pre.X0 + (size_t)blk * 16 * (size_t)S.pre.ldx0 : nullptr;
        if (PRE && cq < S.pre.ldx0) x0 = *reinterpret_cast<const f32x4*>(x0p + jc * (unsigned)S.pre.ldx0 + cq);
        // WG0: X_0^T as A operand -- lane (i = j, q), step t = X_0[row 4 q + t][i]; lane K0 carries ones (bias gradient), lanes past it zeros.
        // Requested now (lines the mask's x0 just touched), used after the MFMAs; rows past the end are zeroed at the use point.
        f32x4 xg = {0.0f, 0.0f, 0.0f, 0.0f};
        if (WG0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const long long left = S.n - blk * 16;               // (uniform)
                const unsigned r = (cq + t < left) ? (unsigned)(cq + t) : (unsigned)(left - 1);
                xg[t] = (j == K0) ? 1.0f : 0.0f;
                if (j < K0) xg[t] = x0p[r * (unsigned)S.pre.ldx0 + j];
            }
        }
        // LIK: what the epilogue needs of this lane's row (the four lanes of a row read the same addresses), requested now; the amplitudes of
        // the lane's first two samples follow half a block later, when the reflection id has landed
        int l_rid = 0, l_img = 0;
        float l_io = 0.0f, l_sg = 1.0f, l_aim = 1.0f, l_zf0 = 0.0f, l_zf1 = 0.0f;
        if (LIK) {
            const sargs_p L = sargs_again();
            const size_t r0 = (size_t)blk * 16;
            l_rid = (L->lik.refl_id + r0)[jc];
            if (L->lik.use_img) l_img = (L->lik.image_id + r0)[jc];
            l_io = (L->lik.iobs + r0)[jc]; l_sg = (L->lik.sig + r0)[jc];
        }
        const bool more = blk + bstep < nblk;
        const long long bnext = more ? blk + bstep : blk;
        const float* xnext = x_base(bnext) + ((unsigned)lane_row(bnext) * (unsigned)S.ldx + cq);
        const float* xcur = x_base(blk) + (jc * (unsigned)S.ldx + cq);
        f32x4 acc[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[a] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        // The NEXT weight quad is requested before the current quad's four MFMAs are issued (round 4, scheduler-pinned: the compiler's order was
        // read -> wait for the LDS -> 4 MFMAs, 64 times a block).  The four MFMAs of a quad stay one dependent chain on one accumulator:
        // alternating two accumulators step by step is SLOWER (forward kernel 0.293 -> 0.308 ms per 1 M rows, profiles/r4_wide_mfma_loop_ab.txt).
        f32x4 wf = *reinterpret_cast<const f32x4*>(wrow);       // (block 0, chunk 0 of the weights)
#pragma unroll
        for (int kc = 0; kc < NA; ++kc) {
            f32x4 xv = xb[kc];
            if (HEADB) {
                // dZ_L[row][16 kc + 4 q + t] from h_L (in xb) and the row's g: (g0 Wo[0][c] + g1 Wo[1][c]) * LeakyReLU'(h_L[row][c])
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(sHead + 16 * kc + cq), h1 = *reinterpret_cast<const f32x4*>(sHead + 16 * NA + 16 * kc + cq);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float dh = fmaf(gc1, h1[t], gc0 * h0[t]);
                    xv[t] = (xb[kc][t] > 0.0f) ? dh : S.leak * dh;
                }
            }
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const int na = (a + 1 < NA) ? a + 1 : 0, nkc = (a + 1 < NA) ? kc : kc + 1;
                f32x4 wnx = wf;
                if (nkc < NA) wnx = *reinterpret_cast<const f32x4*>(wrow + 16 * na * SKP + 16 * nkc);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[a] = WG0 ? mfma4(xv[t], wf[t], acc[a]) : mfma4(wf[t], xv[t], acc[a]);
                wf = wnx;
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read (the next quad) ...
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // ... then this quad's four MFMAs
            }
            // this chunk's MFMAs are issued: request the quad that is used H chunks from now
            {
                const int t = kc + H;
                if (t < NA) { if (16 * t + cq < S.ldx) xb[t] = *reinterpret_cast<const f32x4*>(xcur + 16 * t); }
                else if (more && 16 * (t - NA) + cq < S.ldx) xb[t - NA] = *reinterpret_cast<const f32x4*>(xnext + 16 * (t - NA));
            }
            if (LIK && kc == NA / 2) {
                const sargs_p L = sargs_again();
                const int Ss = L->lik.S;
                if (q < Ss) l_zf0 = L->lik.z_f[(size_t)l_rid * Ss + q];
                if (q + 4 < Ss) l_zf1 = L->lik.z_f[(size_t)l_rid * Ss + q + 4];
                if (L->lik.use_img && l_img > 0) l_aim = L->lik.img[l_img - 1];
            }
            if (HEADB && kc == (NA >= 2 ? NA - 2 : 0) && more) {
                const unsigned r = (unsigned)lane_row(bnext);
                const float* gp = S.dO + 2 * (size_t)bnext * 16;
                gn0 = gp[2 * r]; gn1 = gp[2 * r + 1]; gnd = (S.dsd + (size_t)bnext * 16)[r];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (HEADB) { gc0 = gn0; gc1 = gn1 * gnd; }
        if constexpr (WG0) {
            const long long nleft = S.n - (blk * 16 + cq);          // rows 4 q + t < nleft exist
            f32x4 xa;
#pragma unroll
            for (int t = 0; t < 4; ++t) xa[t] = (t < nleft) ? xg[t] : 0.0f;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const f32x4 wf = *reinterpret_cast<const f32x4*>(sW0 + (16 * a + j) * S0P + cq0);
                f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 4; ++t) z = mfma4(x0[t], wf[t], z);      // h_0's pre-activations [row 4 q + t][column 16 a + j], bias below
                const float b0 = sB0[16 * a + j];
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = (z[t] + b0 > 0.0f) ? acc[a][t] : S.leak * acc[a][t];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc0[a] = mfma4(xa[t], v[t], acc0[a]);
            }
            continue;                                                // (nothing is stored per row)
        }
        if (PRE) {
            // the recomputed first layer's pre-activations of this lane's row and columns (every lane takes part: an MFMA reads the
            // weights of feature 16 a + j from all lanes)
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const f32x4 wf = *reinterpret_cast<const f32x4*>(sW0 + (16 * a + j) * S0P + cq0);
                f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 4; ++t) z = mfma4(wf[t], x0[t], z);
                hm[a] = z + *reinterpret_cast<const f32x4*>(sB0 + 16 * a + cq);
            }
        }
        if (row < S.n) {
            float* yp = S.Y + (size_t)blk * 16 * (size_t)S.ldy + ((unsigned)j * (unsigned)S.ldy + cq);
            float ho0 = 0.0f, ho1 = 0.0f;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                f32x4 v = acc[a];
                if (EPI == EPI_BIAS_LRELU) {
                    v += *reinterpret_cast<const f32x4*>(sBias + 16 * a + cq);
                    if (S.act) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], S.leak * v[t]);
                    }
                    if (S.head_W != nullptr) {
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(sHead + 16 * a + cq), g1 = *reinterpret_cast<const f32x4*>(sHead + 16 * NA + 16 * a + cq);
#pragma unroll
                        for (int t = 0; t < 4; ++t) { ho0 = fmaf(v[t], g0[t], ho0); ho1 = fmaf(v[t], g1[t], ho1); }
                    }
                } else if (PRE || S.H != nullptr) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = (hm[a][t] > 0.0f) ? v[t] : S.leak * v[t];
                }
                if (16 * a + cq < S.ldy) *reinterpret_cast<f32x4*>(yp + 16 * a) = v;      // (columns N .. ld - 1 get the zeros the padded weights produce)
            }
            if (EPI == EPI_BIAS_LRELU && S.head_W != nullptr) {
                ho0 += __shfl_xor(ho0, 16); ho1 += __shfl_xor(ho1, 16);
                ho0 += __shfl_xor(ho0, 32); ho1 += __shfl_xor(ho1, 32);
                float dd;
                const float loc = ho0 + S.head_W[2 * N];
                const float sigma = cl_scale_bij(ho1 + S.head_W[2 * N + 1], S.bij_kind, S.eps, &dd);
                if (q == 0) {
                    S.loc_out[row] = loc;
                    S.sig_out[row] = sigma;
                    if (S.dsd_out != nullptr) S.dsd_out[row] = dd;
                }
                if (LIK) {
                    const sargs_p L = sargs_again();
                    // the row's samples over its four lanes: lane q takes s = q, q + 4, ... (reference: mono.py:10-37 on variational.py:167's prediction)
                    const unsigned long long gidx = L->lik.row_index != nullptr ? (unsigned long long)(L->lik.row_index + (size_t)blk * 16)[jc]
                                                                               : (unsigned long long)(L->lik.obs_offset + row);
                    float dl = 0.0f, ds = 0.0f, row_nll = 0.0f;
                    for (int sm = q; sm < L->lik.S; sm += 4) {
                        const float zf = sm == q ? l_zf0 : (sm == q + 4 ? l_zf1 : L->lik.z_f[(size_t)l_rid * L->lik.S + sm]);
                        const float eta = cl_noise_normal(L->lik.seed, L->lik.step, (uint32_t)sm, gidx);
                        const float tq = loc + sigma * eta + L->lik.shift;
                        float dll;
                        const float ll = cl_lik_log_prob(l_aim * tq * zf * zf, l_io, l_sg, L->lik.lik_kind, L->lik.dof, L->lik.lik_const, &dll);
                        row_nll -= ll * L->lik.w_ll;
                        const float gi = -dll * L->lik.w_ll;             // dNLL / d ipred
                        const float dzs = gi * zf * zf;
                        atomicAdd(L->lik.dz_f + (size_t)l_rid * L->lik.S + sm, gi * l_aim * tq * 2.0f * zf);
                        const float dt = dzs * l_aim;
                        dl += dt; ds += dt * eta; l_da += dzs * tq;
                    }
                    l_nll += (double)row_nll;
                    dl += __shfl_xor(dl, 16); ds += __shfl_xor(ds, 16); l_da += __shfl_xor(l_da, 16);
                    dl += __shfl_xor(dl, 32); ds += __shfl_xor(ds, 32); l_da += __shfl_xor(l_da, 32);
                    if (q == 0) { (L->lik.dO + 2 * (size_t)blk * 16)[2 * j] = dl; (L->lik.dO + 2 * (size_t)blk * 16)[2 * j + 1] = ds; }
                    l_take = q == 0 && l_img > 0;
                }
            }
        }
        if (LIK && sargs_again()->lik.use_img) {
            const sargs_p L = sargs_again();
            // image-scale gradients of the wave's sixteen rows (every lane calls: the reduction is wave-wide; image 0 is pinned, image.py:23-25)
            const unsigned long long m = __ballot(l_take);
            if (m != 0ull) {
                const int im0 = __builtin_amdgcn_readlane(l_img, __builtin_ctzll(m));
                if (__all(!l_take || l_img == im0)) {
                    const float v = cl_wave_sum(l_take ? l_da : 0.0f);
                    if (lane == 0) atomicAdd(L->lik.d_img + (im0 - 1), v);
                } else {
                    cl_image_grad_segments(L->lik.d_img, l_img, l_da, l_take, lane);
                }
            }
        }
    }
    if constexpr (LIK) {
        // the NLL of this workgroup's rows: one fp64 atomic (same-address atomics serialise: not one per wave)
        double v = l_nll;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        __syncthreads();
        double* const sNll = reinterpret_cast<double*>(sBias);      // (16 NA floats >= 8 doubles; sBias sits on a 16-byte boundary)
        if (lane == 0) sNll[wv] = v;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int k = 0; k < 8; ++k) t += sNll[k];
            atomicAdd(sargs_again()->lik.scalars + CL_SC_NLL, t);
        }
    }
    if constexpr (WG0) {
        // the eight waves' accumulators -> this workgroup's partial, in the first layer's flat layout [Wt_0 (N x K0) | b_0 (N)]; the weight
        // image's LDS is free once every wave has left the loop (8 x 16 NA x 16 floats fit its 16 NA x 132)
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NA; ++a) *reinterpret_cast<f32x4*>(sWq + ((wv * 16 * NA + 16 * a + j) * 16 + cq)) = acc0[a];
        __syncthreads();
        float* part = S.wg0_part + (size_t)blockIdx.x * ((size_t)N * K0 + N);
        for (int idx = tid; idx < 16 * NA * 16; idx += 512) {
            const int c = idx >> 4, i = idx & 15;
            if (c < N && i <= K0) {
                float t = 0.0f;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) t += sWq[(w8 * 16 * NA + c) * 16 + i];
                if (i < K0) part[(size_t)c * K0 + i] = t; else part[(size_t)N * K0 + c] = t;
            }
        }
    }
}

// 1: the square-layer kernel takes this call (same block count on both sides, row buffers laid out as cl_wide_ld says, 16-byte aligned)
static bool sq_ok(const StreamArgs& s) {
    if (s.seg != nullptr) return false;
    const int NA = (s.N + 15) >> 4, KA = (s.K + 15) >> 4;
    if (NA != KA || NA < 5) return false;                      // (up to 64 the generic instance is as good; 65 .. 128 is what this path is for)
    if (s.ldx % 4 != 0 || s.ldy % 4 != 0 || s.ldx < s.K || s.ldy < s.N || s.ldx > 16 * NA || s.ldy > 16 * NA) return false;
    if ((reinterpret_cast<uintptr_t>(s.X) & 15) != 0 || (reinterpret_cast<uintptr_t>(s.Y) & 15) != 0) return false;
    if (s.H != nullptr && (s.ldh % 4 != 0 || s.ldh > 16 * NA || (reinterpret_cast<uintptr_t>(s.H) & 15) != 0)) return false;
    return true;
}

static long long sq_grid(long long n) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long nblk = (n + 15) >> 4;
    long long grid = (nblk + 7) / 8;
    if (grid > 2LL * cus) grid = 2LL * cus;
    return grid < 1 ? 1 : grid;
}

template <bool WKM, int EPI, int NA, bool PRE, bool WG0 = false, bool HEADB = false, bool LIK = false>
int launch_sq_n(const StreamArgs& s, hipStream_t st) {
    const size_t sm = (size_t)(16 * NA * SKP + 3 * 16 * NA + (PRE ? 16 * NA * (S0P + 1) : 0)) * sizeof(float);
    auto kern = wide_sq_kernel<WKM, EPI, NA, PRE, WG0, HEADB, LIK>;
    static std::atomic<size_t> configured{0};
    size_t have = configured.load(std::memory_order_acquire);
    if (have < sm) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        while (have < sm && !configured.compare_exchange_weak(have, sm, std::memory_order_release, std::memory_order_acquire)) {}
    }
    const long long grid = sq_grid(s.n);
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), sm, st, s);
    return (int)hipGetLastError();
}

template <bool WKM, int EPI, bool PRE, bool WG0 = false, bool HEADB = false, bool LIK = false>
int launch_sq(const StreamArgs& s, hipStream_t st) {
    switch ((s.N + 15) >> 4) {
        case 5: return launch_sq_n<WKM, EPI, 5, PRE, WG0, HEADB, LIK>(s, st);
        case 6: return launch_sq_n<WKM, EPI, 6, PRE, WG0, HEADB, LIK>(s, st);
        case 7: return launch_sq_n<WKM, EPI, 7, PRE, WG0, HEADB, LIK>(s, st);
        default: return launch_sq_n<WKM, EPI, 8, PRE, WG0, HEADB, LIK>(s, st);
    }
}

template <bool WKM, int EPI, int NAT, bool GRP, bool PRE = false>
int launch_stream_n(const StreamArgs& s, hipStream_t st) {
    const int NA = (s.N + 15) >> 4;
    // weights, bias, the fused head's two rows; the dgrad with a recomputed mask adds the first layer's image and bias
    const size_t sm = (size_t)(16 * NA * SKP + 3 * 16 * NA + (PRE ? 16 * NA * (S0P + 1) : 0)) * sizeof(float);
    auto kern = wide_stream_kernel<WKM, EPI, NAT, GRP, PRE>;
    static std::atomic<size_t> configured{0};
    size_t have = configured.load(std::memory_order_acquire);
    if (have < sm) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        while (have < sm && !configured.compare_exchange_weak(have, sm, std::memory_order_release, std::memory_order_acquire)) {}
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long nblk = (s.n + 15) >> 4;
    long long grid = s.seg != nullptr ? (long long)s.n_groups : (nblk + 7) / 8;
    if (grid > 2LL * cus) grid = 2LL * cus;          // two 8-wave workgroups per CU (2 x 67.6 KB of LDS at 128 x 128)
    if (grid < 1) grid = 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), sm, st, s);
    return (int)hipGetLastError();
}

template <bool WKM, int EPI>
int launch_stream(const StreamArgs& s, hipStream_t st) {
    if (s.seg != nullptr) return s.N <= 64 ? launch_stream_n<WKM, EPI, 4, true>(s, st) : launch_stream_n<WKM, EPI, 8, true>(s, st);
    if (s.N <= 64) return launch_stream_n<WKM, EPI, 4, false>(s, st);
    return launch_stream_n<WKM, EPI, 8, false>(s, st);
}

template <bool AK, bool BK_, int EPI>
int launch_gemm_tiles(const GemmArgs& g, int n_tiles, hipStream_t st) {
    if (n_tiles <= 0 || g.N <= 0 || g.K <= 0 || g.tiles == nullptr || g.seg == nullptr) return -1;
    (void)hipGetLastError();
    hipLaunchKernelGGL((wide_gemm_kernel<AK, BK_, EPI, 128>), dim3(n_tiles, (g.N + 127) / 128, 1), dim3(256), 0, st, g);
    return (int)hipGetLastError();
}

template <bool AK, bool BK_, int EPI>
int launch_gemm(const GemmArgs& g, int zsplit, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return -1;
    (void)hipGetLastError();
    if (g.N > 64) hipLaunchKernelGGL((wide_gemm_kernel<AK, BK_, EPI, 128>), dim3((g.M + BM - 1) / BM, (g.N + 127) / 128, zsplit), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((wide_gemm_kernel<AK, BK_, EPI, 64>), dim3((g.M + BM - 1) / BM, 1, zsplit), dim3(256), 0, st, g);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int cl_wide_ld(int width) { return width < 1 ? 0 : ((width + 3) & ~3); }

int cl_wide_dense_forward(const float* X, int ldx, const float* Wt, const float* b, long long n, int n_in, int n_out, float leak, int act,
                          float* Y, int ldy, const int* stop_flag, void* stream) {
    if (X == nullptr || Wt == nullptr || b == nullptr || Y == nullptr || n < 1 || n > 0x7fffffffLL || n_in < 1 || n_out < 1 || ldx < n_in || ldy < n_out) return -1;
    if (n_in <= SMAX && n_out <= SMAX) {
        StreamArgs s = {};
        s.X = X; s.ldx = ldx; s.W = Wt; s.ldw = n_in; s.Y = Y; s.ldy = ldy; s.n = n; s.N = n_out; s.K = n_in;
        s.bias = b; s.leak = leak; s.act = act; s.stop_flag = stop_flag;
        if (sq_ok(s)) return launch_sq<false, EPI_BIAS_LRELU, false>(s, (hipStream_t)stream);
        return launch_stream<false, EPI_BIAS_LRELU>(s, (hipStream_t)stream);
    }
    GemmArgs g = {};
    g.A = X; g.lda = ldx; g.B = Wt; g.ldb = n_in; g.C = Y; g.ldc = ldy;
    g.M = (int)n; g.N = n_out; g.K = n_in; g.bias = b; g.leak = leak; g.act = act; g.stop_flag = stop_flag;
    return launch_gemm<false, false, EPI_BIAS_LRELU>(g, 1, (hipStream_t)stream);
}

/* the top Dense layer with the Dense(2) head in its epilogue: Y as cl_wide_dense_forward (the backward pass needs it), and
 * loc = Y . Wo[0] + bo[0], sigma = bijector(Y . Wo[1] + bo[1]) + eps per row (`head` = the head's flat layout [Wo (2 x n_out) | bo (2)]).
 * Layers up to 128 x 128 (the streaming kernel); -2 otherwise: the caller then runs cl_wide_dense_forward + cl_wide_head_forward.      */
int cl_wide_dense_forward_head(const float* X, int ldx, const float* Wt, const float* b, long long n, int n_in, int n_out, float leak,
                               float* Y, int ldy, const float* head, int bij_kind, float eps, float* loc_out, float* sig_out, float* dsig_draw_out,
                               const int* stop_flag, void* stream) {
    if (X == nullptr || Wt == nullptr || b == nullptr || Y == nullptr || head == nullptr || loc_out == nullptr || sig_out == nullptr || n < 1 ||
        n > 0x7fffffffLL || n_in < 1 || n_out < 1 || ldx < n_in || ldy < n_out)
        return -1;
    if (n_in > SMAX || n_out > SMAX) return -2;
    StreamArgs s = {};
    s.X = X; s.ldx = ldx; s.W = Wt; s.ldw = n_in; s.Y = Y; s.ldy = ldy; s.n = n; s.N = n_out; s.K = n_in;
    s.bias = b; s.leak = leak; s.act = 1; s.stop_flag = stop_flag;
    s.head_W = head; s.bij_kind = bij_kind; s.eps = eps; s.loc_out = loc_out; s.sig_out = sig_out; s.dsd_out = dsig_draw_out;
    if (sq_ok(s)) return launch_sq<false, EPI_BIAS_LRELU, false>(s, (hipStream_t)stream);
    return launch_stream<false, EPI_BIAS_LRELU>(s, (hipStream_t)stream);
}

/* cl_wide_dense_forward_head with the slot likelihood of the call's rows in the same epilogue (round 4): what cl_slot_rows computes from
 * (loc, sigma) per row -- sample, predict, log-prob, gradient -- where those two numbers are made; `lik` as for cl_slot_rows with its row
 * arrays at the call's first row (loc / sigma / iconv unused).  Rows that are their own slot, in-kernel noise, no ipred_out, no
 * Evans-2011 terms, no deterministic stores, the square-layer kernel's envelope: -2 otherwise (the caller then runs
 * cl_wide_dense_forward_head and cl_slot_rows).                                                                                          */
int cl_wide_dense_forward_head_lik(const float* X, int ldx, const float* Wt, const float* b, long long n, int n_in, int n_out, float leak,
                                   float* Y, int ldy, const float* head, int bij_kind, float eps, float* loc_out, float* sig_out, float* dsig_draw_out,
                                   const cl_laue_args* lik, const int* stop_flag, void* stream) {
    if (X == nullptr || Wt == nullptr || b == nullptr || Y == nullptr || head == nullptr || loc_out == nullptr || sig_out == nullptr || lik == nullptr ||
        n < 1 || n > 0x7fffffffLL || n_in < 1 || n_out < 1 || ldx < n_in || ldy < n_out)
        return -1;
    if (n_in > SMAX || n_out > SMAX) return -2;
    const cl_laue_args& a = *lik;
    if (a.harmonic_id != nullptr || a.eta != nullptr || a.ipred_out != nullptr || a.ev11 != nullptr || a.dzf_obs != nullptr || a.nll_part != nullptr) return -2;
    if (a.refl_id == nullptr || a.iobs == nullptr || a.sig == nullptr || a.z_f == nullptr || a.dz_f == nullptr || a.dO == nullptr || a.scalars == nullptr ||
        a.S < 1 || a.n_obs != (int)n || (a.use_img && (a.image_id == nullptr || a.img == nullptr || a.d_img == nullptr)))
        return -1;
    StreamArgs s = {};
    s.X = X; s.ldx = ldx; s.W = Wt; s.ldw = n_in; s.Y = Y; s.ldy = ldy; s.n = n; s.N = n_out; s.K = n_in;
    s.bias = b; s.leak = leak; s.act = 1; s.stop_flag = stop_flag;
    s.head_W = head; s.bij_kind = bij_kind; s.eps = eps; s.loc_out = loc_out; s.sig_out = sig_out; s.dsd_out = dsig_draw_out;
    if (!sq_ok(s)) return -2;
    s.lik.refl_id = a.refl_id; s.lik.image_id = a.image_id; s.lik.iobs = a.iobs; s.lik.sig = a.sig;
    s.lik.row_index = a.row_index; s.lik.obs_offset = a.obs_offset;
    s.lik.img = a.img; s.lik.use_img = a.use_img; s.lik.z_f = a.z_f; s.lik.S = a.S;
    s.lik.lik_kind = a.lik_kind; s.lik.dof = a.dof; s.lik.lik_const = a.lik_const; s.lik.shift = a.shift; s.lik.w_ll = a.w_ll;
    s.lik.seed = a.seed; s.lik.step = a.step;
    s.lik.dz_f = a.dz_f; s.lik.d_img = a.d_img; s.lik.dO = a.dO; s.lik.scalars = a.scalars;
    return launch_sq<false, EPI_BIAS_LRELU, false, false, false, true>(s, (hipStream_t)stream);
}

/* The Dense(2) head's backward pass inside the TOP layer's dgrad and weight gradient (round 4; replaces cl_wide_head_backward + its dZ_L buffer):
 * Htop = the top layer's activations h_L [n][ldt] as cl_wide_dense_forward_head stored them, dO = dL/d(loc, sigma) per row [n][2],
 * dsig_draw = that call's dsig_draw_out, head = the head's flat layout.  dZ_L = (g Wo) * LeakyReLU'(h_L), g = (dO[0], dO[1] dsig_draw), is
 * made where the kernels read their dZ operand.  Width 65 .. 128 on both sides of the layer (cl_wide_head_bwd_supported), else -2.            */
int cl_wide_head_bwd_supported(int n_out, int n_in) {
    const int NA = (n_in + 15) >> 4, KA = (n_out + 15) >> 4;
    return n_in >= 1 && n_out >= 1 && NA == KA && NA >= 5 && NA <= SMAX / 16;
}

int cl_wide_dense_dgrad_head(const float* Htop, int ldt, const float* head, const float* dO, const float* dsig_draw, const float* Wt, long long n,
                             int n_out, int n_in, const float* Hprev, int ldh, float leak, float* dX, int ldo, const int* stop_flag, void* stream) {
    if (Htop == nullptr || head == nullptr || dO == nullptr || dsig_draw == nullptr || Wt == nullptr || dX == nullptr || n < 1 || n > 0x7fffffffLL ||
        n_in < 1 || n_out < 1 || ldt < n_out || ldo < n_in)
        return -1;
    if (!cl_wide_head_bwd_supported(n_out, n_in)) return -2;
    StreamArgs s = {};
    s.X = Htop; s.ldx = ldt; s.W = Wt; s.ldw = n_in; s.Y = dX; s.ldy = ldo; s.n = n; s.N = n_in; s.K = n_out;
    s.H = Hprev; s.ldh = ldh; s.leak = leak; s.stop_flag = stop_flag;
    s.head_W = head; s.dO = dO; s.dsd = dsig_draw;
    if (!sq_ok(s)) return -2;
    return launch_sq<true, EPI_DLRELU, false, false, true>(s, (hipStream_t)stream);
}

int cl_wide_dense_dgrad(const float* dZ, int lddz, const float* Wt, long long n, int n_out, int n_in, const float* Hprev, int ldh, float leak,
                        float* dX, int ldo, const int* stop_flag, void* stream) {
    if (dZ == nullptr || Wt == nullptr || dX == nullptr || n < 1 || n > 0x7fffffffLL || n_in < 1 || n_out < 1 || lddz < n_out || ldo < n_in) return -1;
    if (n_in <= SMAX && n_out <= SMAX) {
        StreamArgs s = {};
        s.X = dZ; s.ldx = lddz; s.W = Wt; s.ldw = n_in; s.Y = dX; s.ldy = ldo; s.n = n; s.N = n_in; s.K = n_out;
        s.H = Hprev; s.ldh = ldh; s.leak = leak; s.stop_flag = stop_flag;
        if (sq_ok(s)) return launch_sq<true, EPI_DLRELU, false>(s, (hipStream_t)stream);
        return launch_stream<true, EPI_DLRELU>(s, (hipStream_t)stream);
    }
    GemmArgs g = {};
    g.A = dZ; g.lda = lddz; g.B = Wt; g.ldb = n_in; g.C = dX; g.ldc = ldo;
    g.M = (int)n; g.N = n_in; g.K = n_out; g.H = Hprev; g.ldh = ldh; g.leak = leak; g.stop_flag = stop_flag;
    return launch_gemm<false, true, EPI_DLRELU>(g, 1, (hipStream_t)stream);
}

/* 1: the first Dense layer of a (n_in0 -> w -> w ...) stack can be recomputed instead of stored (cl_wide_dense2_forward,
 * cl_wide_dense_dgrad_pre, cl_wide_dense_wgrad_pre): metadata of at most 15 columns, hidden width at most 128 */
int cl_wide_pre_supported(int n_in0, int w) { return n_in0 >= 1 && n_in0 <= K0WG && w >= 1 && w <= SMAX; }

static int pre_check(const float* X0, int ldx0, int n_in0, const float* Wt0, const float* b0, int w) {
    if (X0 == nullptr || Wt0 == nullptr || b0 == nullptr) return -1;
    if (!cl_wide_pre_supported(n_in0, w)) return -2;
    if (ldx0 < n_in0 || ldx0 % 4 != 0 || ldx0 > K0MAX || (reinterpret_cast<uintptr_t>(X0) & 15) != 0) return -1;
    return 0;
}

int cl_wide_dense2_forward(const float* X0, int ldx0, int n_in0, const float* Wt0, const float* b0, const float* Wt1, const float* b1, long long n,
                           int w0, int w1, float leak, float* Y, int ldy, const float* head, int bij_kind, float eps, float* loc_out, float* sig_out,
                           const int* stop_flag, void* stream) {
    if (int e = pre_check(X0, ldx0, n_in0, Wt0, b0, w0)) return e;
    if (Wt1 == nullptr || b1 == nullptr || Y == nullptr || n < 1 || n > 0x7fffffffLL || w1 < 1 || ldy < w1) return -1;
    if (w1 > SMAX) return -2;
    if (head != nullptr && (loc_out == nullptr || sig_out == nullptr)) return -1;
    Stream2Args s = {};
    s.pre.X0 = X0; s.pre.ldx0 = ldx0; s.pre.K0 = n_in0; s.pre.W0 = Wt0; s.pre.b0 = b0; s.pre.N0 = w0;
    s.W1 = Wt1; s.b1 = b1; s.N1 = w1; s.Y = Y; s.ldy = ldy; s.n = n; s.leak = leak;
    s.head_W = head; s.bij_kind = bij_kind; s.eps = eps; s.loc_out = loc_out; s.sig_out = sig_out; s.stop_flag = stop_flag;
    hipStream_t st = (hipStream_t)stream;
    switch (((w0 > w1 ? w0 : w1) + 15) >> 4) {         // exact block count: zero-padded blocks would cost real MFMAs
        case 1: return launch_stream2<1>(s, st);
        case 2: return launch_stream2<2>(s, st);
        case 3: return launch_stream2<3>(s, st);
        case 4: return launch_stream2<4>(s, st);
        case 5: return launch_stream2<5>(s, st);
        case 6: return launch_stream2<6>(s, st);
        case 7: return launch_stream2<7>(s, st);
        default: return launch_stream2<8>(s, st);
    }
}

int cl_wide_dense_dgrad_pre(const float* dZ, int lddz, const float* Wt, long long n, int n_out, int n_in, const float* X0, int ldx0, int n_in0,
                            const float* Wt0, const float* b0, float leak, float* dX, int ldo, const int* stop_flag, void* stream) {
    if (int e = pre_check(X0, ldx0, n_in0, Wt0, b0, n_in)) return e;
    if (dZ == nullptr || Wt == nullptr || dX == nullptr || n < 1 || n > 0x7fffffffLL || n_out < 1 || lddz < n_out || ldo < n_in) return -1;
    if (n_out > SMAX) return -2;
    StreamArgs s = {};
    s.X = dZ; s.ldx = lddz; s.W = Wt; s.ldw = n_in; s.Y = dX; s.ldy = ldo; s.n = n; s.N = n_in; s.K = n_out;
    s.leak = leak; s.stop_flag = stop_flag;
    s.pre.X0 = X0; s.pre.ldx0 = ldx0; s.pre.K0 = n_in0; s.pre.W0 = Wt0; s.pre.b0 = b0; s.pre.N0 = n_in;
    if (sq_ok(s)) return launch_sq<true, EPI_DLRELU, true>(s, (hipStream_t)stream);
    return s.N <= 64 ? launch_stream_n<true, EPI_DLRELU, 4, false, true>(s, (hipStream_t)stream) : launch_stream_n<true, EPI_DLRELU, 8, false, true>(s, (hipStream_t)stream);
}

/* The second layer's dgrad and the FIRST layer's weight gradient in one launch (round 4): dZ_0 = (dZ Wt) * LeakyReLU'(h_0) with h_0 recomputed
 * from the metadata as in cl_wide_dense_dgrad_pre, contracted with the metadata rows on the spot -- dZ_0 is not stored.  Writes
 * cl_wide_dgrad_wgrad0_parts(n) partial sums of [dWt_0 (n_in x n_in0) | db_0 (n_in)] into `partials` (add them with cl_reduce_partials).
 * -2: shape not taken (the square-layer kernel's envelope: 65 .. 128 on both sides, same number of 16-column blocks): the caller then
 * runs cl_wide_dense_dgrad_pre + cl_wide_dense_wgrad.                                                                               */
int cl_wide_dgrad_wgrad0_parts(long long n) { return n < 1 ? 0 : (int)sq_grid(n); }

int cl_wide_dense_dgrad_pre_wgrad0(const float* dZ, int lddz, const float* Wt, long long n, int n_out, int n_in, const float* X0, int ldx0, int n_in0,
                                   const float* Wt0, const float* b0, float leak, float* partials, const int* stop_flag, void* stream) {
    if (int e = pre_check(X0, ldx0, n_in0, Wt0, b0, n_in)) return e;
    if (dZ == nullptr || Wt == nullptr || partials == nullptr || n < 1 || n > 0x7fffffffLL || n_out < 1 || lddz < n_out) return -1;
    if (n_out > SMAX) return -2;
    StreamArgs s = {};
    s.X = dZ; s.ldx = lddz; s.W = Wt; s.ldw = n_in; s.Y = const_cast<float*>(dZ); s.ldy = lddz; s.n = n; s.N = n_in; s.K = n_out;   /* (Y: never written; set for sq_ok's layout tests) */
    s.leak = leak; s.stop_flag = stop_flag;
    s.pre.X0 = X0; s.pre.ldx0 = ldx0; s.pre.K0 = n_in0; s.pre.W0 = Wt0; s.pre.b0 = b0; s.pre.N0 = n_in;
    s.wg0_part = partials;
    if (!sq_ok(s)) return -2;
    return launch_sq<true, EPI_DLRELU, true, true>(s, (hipStream_t)stream);
}

int cl_wide_dense_wgrad_pre(const float* dZ, int lddz, const float* X0, int ldx0, int n_in0, const float* Wt0, const float* b0, float leak, long long n,
                            int n_out, int n_in, float* partials, int nsplit, const int* stop_flag, void* stream) {
    if (int e = pre_check(X0, ldx0, n_in0, Wt0, b0, n_in)) return e;
    if (dZ == nullptr || partials == nullptr || n < 1 || n > 0x7fffffffLL || n_out < 1 || nsplit < 1 || lddz < n_out) return -1;
    GemmArgs g = {};
    g.A = dZ; g.lda = lddz; g.B = X0; g.ldb = ldx0; g.C = partials;
    g.M = n_out; g.N = n_in; g.K = (int)n; g.n_in = n_in;
    g.ksplit = (int)((n + nsplit - 1) / nsplit);
    g.ksplit = (g.ksplit + BK - 1) / BK * BK;
    g.pstride = (long long)n_out * n_in + n_out;
    g.leak = leak; g.stop_flag = stop_flag;
    g.pre.X0 = X0; g.pre.ldx0 = ldx0; g.pre.K0 = n_in0; g.pre.W0 = Wt0; g.pre.b0 = b0; g.pre.N0 = n_in;
    // the layer's input tile is made by MFMAs where the kernel stages it: one 128-column tile of a 128-row output (cl_wide_pre_supported
    // bounds both widths by 128; a width <= 64 never comes here -- the fused kernels take it)
    if (n_in > 128 || n_out > 128) return -2;
    (void)hipGetLastError();
    hipLaunchKernelGGL((wide_gemm_kernel<true, true, EPI_WGRAD, 128, false, true>), dim3(1, 1, nsplit), dim3(256), 0, (hipStream_t)stream, g);
    return (int)hipGetLastError();
}

int cl_wide_wgrad_splits(long long n) {
    long long s = (n + 511) / 512;                 // >= 512 observations per split (16 chunks of the contraction), at most 512 splits:
    return (int)(s < 1 ? 1 : (s > 512 ? 512 : s)); // a layer's output is one or a few tiles, the splits are what fills the chip
}

int cl_wide_dense_wgrad(const float* dZ, int lddz, const float* H, int ldh, long long n, int n_out, int n_in, float* partials, int nsplit,
                        const int* stop_flag, void* stream) {
    if (dZ == nullptr || H == nullptr || partials == nullptr || n < 1 || n > 0x7fffffffLL || n_in < 1 || n_out < 1 || nsplit < 1 || lddz < n_out ||
        ldh < n_in)
        return -1;
    GemmArgs g = {};
    g.A = dZ; g.lda = lddz; g.B = H; g.ldb = ldh; g.C = partials;
    g.M = n_out; g.N = n_in; g.K = (int)n; g.n_in = n_in;
    g.ksplit = (int)((n + nsplit - 1) / nsplit);
    g.ksplit = (g.ksplit + BK - 1) / BK * BK;
    g.pstride = (long long)n_out * n_in + n_out;
    g.stop_flag = stop_flag;
    return launch_gemm<true, true, EPI_WGRAD>(g, nsplit, (hipStream_t)stream);
}

/* head_partials[s][2 n_out + 2] = (dWo | dbo) over the s-th row range, next to the layer's own partials as cl_wide_dense_wgrad writes them */
int cl_wide_dense_wgrad_head(const float* Htop, int ldt, const float* head, const float* dO, const float* dsig_draw, float leak, const float* H, int ldh,
                             long long n, int n_out, int n_in, float* partials, float* head_partials, int nsplit, const int* stop_flag, void* stream) {
    if (Htop == nullptr || head == nullptr || dO == nullptr || dsig_draw == nullptr || H == nullptr || partials == nullptr || head_partials == nullptr ||
        n < 1 || n > 0x7fffffffLL || n_in < 1 || n_out < 1 || nsplit < 1 || ldt < n_out || ldh < n_in)
        return -1;
    if (!cl_wide_head_bwd_supported(n_out, n_in)) return -2;
    // (the loader reads dO and dsig_draw as aligned quads of four consecutive rows)
    if ((reinterpret_cast<uintptr_t>(dO) & 15) != 0 || (reinterpret_cast<uintptr_t>(dsig_draw) & 15) != 0) return -1;
    GemmArgs g = {};
    g.A = Htop; g.lda = ldt; g.B = H; g.ldb = ldh; g.C = partials;
    g.M = n_out; g.N = n_in; g.K = (int)n; g.n_in = n_in;
    g.ksplit = (int)((n + nsplit - 1) / nsplit);
    g.ksplit = (g.ksplit + BK - 1) / BK * BK;
    g.pstride = (long long)n_out * n_in + n_out;
    g.leak = leak; g.stop_flag = stop_flag;
    g.hd_dO = dO; g.hd_dsd = dsig_draw; g.hd_W = head; g.hd_part = head_partials;
    (void)hipGetLastError();
    hipLaunchKernelGGL((wide_gemm_kernel<true, true, EPI_WGRAD, 128, true>), dim3(1, 1, nsplit), dim3(256), 0, (hipStream_t)stream, g);
    return (int)hipGetLastError();
}

/* per-image layers (grouped): rows sorted by image, seg[g] .. seg[g+1] = rows of image g of this call (n_groups images);
 * W = the layer's kernels [image][out][in] (w x w each), b = its biases [image][w]; both pointers at the call's first image */
int cl_wide_image_forward(const float* X, int ldx, const float* W, const float* b, const int* seg, int n_groups, long long n, int w, float leak,
                          float* Y, int ldy, const int* stop_flag, void* stream) {
    if (X == nullptr || W == nullptr || b == nullptr || seg == nullptr || Y == nullptr || n_groups < 1 || n < 1 || w < 1 || w > SMAX || ldx < w || ldy < w) return w > SMAX ? -2 : -1;
    StreamArgs s = {};
    s.X = X; s.ldx = ldx; s.W = W; s.ldw = w; s.Y = Y; s.ldy = ldy; s.n = n; s.N = w; s.K = w;
    s.bias = b; s.leak = leak; s.act = 1; s.stop_flag = stop_flag;
    s.seg = seg; s.n_groups = n_groups; s.wstride = (long long)w * w; s.bstride = w;
    return launch_stream<false, EPI_BIAS_LRELU>(s, (hipStream_t)stream);
}

int cl_wide_image_dgrad(const float* dZ, int lddz, const float* W, const int* seg, int n_groups, long long n, int w, const float* Hprev, int ldh, float leak,
                        float* dX, int ldo, const int* stop_flag, void* stream) {
    if (dZ == nullptr || W == nullptr || seg == nullptr || dX == nullptr || n_groups < 1 || n < 1 || w < 1 || w > SMAX || lddz < w || ldo < w) return w > SMAX ? -2 : -1;
    StreamArgs s = {};
    s.X = dZ; s.ldx = lddz; s.W = W; s.ldw = w; s.Y = dX; s.ldy = ldo; s.n = n; s.N = w; s.K = w;
    s.H = Hprev; s.ldh = ldh; s.leak = leak; s.stop_flag = stop_flag;
    s.seg = seg; s.n_groups = n_groups; s.wstride = (long long)w * w;
    return launch_stream<true, EPI_DLRELU>(s, (hipStream_t)stream);
}

/* Per-image layers wider than 128 (round 4; the streaming kernel holds a layer up to 128 x 128): the tiled kernel, one x-block per entry of
 * `tiles` = (group, first row) pairs covering every group's rows in 128-row pieces (the caller builds the list from its row counts) */
int cl_wide_image_forward_tiles(const float* X, int ldx, const float* W, const float* b, const int* seg, const int* tiles, int n_tiles, int w, float leak,
                                float* Y, int ldy, const int* stop_flag, void* stream) {
    if (X == nullptr || W == nullptr || b == nullptr || seg == nullptr || tiles == nullptr || Y == nullptr || n_tiles < 1 || w < 1 || ldx < w || ldy < w) return -1;
    GemmArgs g = {};
    g.A = X; g.lda = ldx; g.B = W; g.ldb = w; g.C = Y; g.ldc = ldy;
    g.M = 0; g.N = w; g.K = w; g.bias = b; g.leak = leak; g.act = 1; g.stop_flag = stop_flag;
    g.seg = seg; g.tiles = tiles; g.bgs = (long long)w * w; g.biasgs = w;
    return launch_gemm_tiles<false, false, EPI_BIAS_LRELU>(g, n_tiles, (hipStream_t)stream);
}

int cl_wide_image_dgrad_tiles(const float* dZ, int lddz, const float* W, const int* seg, const int* tiles, int n_tiles, int w, const float* Hprev, int ldh,
                              float leak, float* dX, int ldo, const int* stop_flag, void* stream) {
    if (dZ == nullptr || W == nullptr || seg == nullptr || tiles == nullptr || dX == nullptr || n_tiles < 1 || w < 1 || lddz < w || ldo < w) return -1;
    GemmArgs g = {};
    g.A = dZ; g.lda = lddz; g.B = W; g.ldb = w; g.C = dX; g.ldc = ldo;
    g.M = 0; g.N = w; g.K = w; g.H = Hprev; g.ldh = ldh; g.leak = leak; g.stop_flag = stop_flag;
    g.seg = seg; g.tiles = tiles; g.bgs = (long long)w * w;
    return launch_gemm_tiles<false, true, EPI_DLRELU>(g, n_tiles, (hipStream_t)stream);
}

/* dW[image][out][in] = dZ_rows^T H_rows, db[image][out] = column sums of dZ_rows, written (not added) for the n_groups images of the call */
int cl_wide_image_wgrad(const float* dZ, int lddz, const float* H, int ldh, const int* seg, int n_groups, long long n, int w, float* dW, float* db,
                        const int* stop_flag, void* stream) {
    if (dZ == nullptr || H == nullptr || seg == nullptr || dW == nullptr || db == nullptr || n_groups < 1 || n < 1 || n > 0x7fffffffLL || w < 1 || lddz < w || ldh < w)
        return -1;
    GemmArgs g = {};
    g.A = dZ; g.lda = lddz; g.B = H; g.ldb = ldh; g.C = dW;
    g.M = w; g.N = w; g.K = (int)n; g.n_in = w;
    g.pstride = (long long)w * w; g.seg = seg; g.Cb = db; g.bstride = w;
    g.stop_flag = stop_flag;
    return launch_gemm<true, true, EPI_WGRAD>(g, n_groups, (hipStream_t)stream);
}

int cl_wide_head_forward(const float* H, int ldh, const float* Wo, long long n, int w, int bij_kind, float eps, float* loc_out, float* sig_out,
                         const int* stop_flag, void* stream) {
    if (H == nullptr || Wo == nullptr || loc_out == nullptr || sig_out == nullptr || n < 1 || w < 1 || ldh < w) return -1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(wide_head_forward_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, H, ldh, Wo, (int)n, w, bij_kind, eps,
                       loc_out, sig_out, stop_flag);
    return (int)hipGetLastError();
}

int cl_wide_head_blocks(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int cl_wide_head_backward(const float* H, int ldh, const float* Wo, const float* dO, long long n, int w, int bij_kind, float eps, float leak,
                          float* dZ, int lddz, float* partials, int nblocks, const int* stop_flag, void* stream) {
    if (H == nullptr || Wo == nullptr || dO == nullptr || dZ == nullptr || partials == nullptr || n < 1 || n > 0x7fffffffLL || w < 1 || nblocks < 1 ||
        ldh < w || lddz < w || ldh % 4 != 0 || lddz % 4 != 0 || (reinterpret_cast<uintptr_t>(H) & 15) != 0 || (reinterpret_cast<uintptr_t>(dZ) & 15) != 0)
        return -1;
    if (w > 1024) return -2;
    const size_t sm = (size_t)8 * (2 * w + 2) * sizeof(float);
    const int rpb = (int)((n + nblocks - 1) / nblocks);
    const int rows = (rpb + 7) / 8 * 8;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
#define CL_HEAD_BWD(NP) \
    hipLaunchKernelGGL(wide_head_backward_kernel<NP>, dim3(nblocks), dim3(256), sm, st, H, ldh, Wo, dO, (int)n, w, bij_kind, eps, leak, rows, dZ, lddz, partials, stop_flag)
    if (w <= 128) CL_HEAD_BWD(1);
    else if (w <= 256) CL_HEAD_BWD(2);
    else if (w <= 512) CL_HEAD_BWD(4);
    else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wide_head_backward_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        CL_HEAD_BWD(8);
    }
#undef CL_HEAD_BWD
    return (int)hipGetLastError();
}

}  // extern "C"
