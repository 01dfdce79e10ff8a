// A wide first layer in front of the default scaler's kernels (gfx950): the "peeled" first Dense layer (round 5).
//
// The lane-per-observation kernel (elbo_lane.hip) holds 20 layers of hidden width <= 10 on at most 31 metadata columns; four encoded keys
// at the default --positional-encoding-frequencies 4 give 37 (careless/args/positional_encoding.py:24-37, args/scaling.py:21-31), and the
// shape fell to the 16-wide instance of elbo_mlp.hip at ~0.18 of the matrix rate.  The first layer h_1 = LeakyReLU(W_0 x + b_0)
// (careless/models/scaling/nn.py:55-68) is the only one that sees the metadata: computed HERE as u = W_0 x + b_0 (w numbers per observation,
// feature-major like the metadata), the fused kernel then runs the SAME scaler with its first layer replaced by the identity on u --
// LeakyReLU(I u + 0) = h_1 bit for bit -- on w <= 10 "metadata columns", the shape it is fastest at, and hands back dL/du = dZ_0 (its
// dX_out); W_0's gradient dZ_0^T X and b_0's are taken here again.  HBM-bound by construction: forward reads 4 d bytes per observation and
// writes 4 w, backward reads 4 (d + w): at 53 columns 0.5 KB per observation and step, ~0.07 ms per million observations at 8 TB/s next to
// the fused kernel's ~0.25 ms.
//
//   cl_peel_forward : u_t[k][i] = b_0[k] + sum_c W_0[k][c] x[c][i]   (fp32 FMA chain, c ascending);  mlp_peel = [I | 0 | layers 1 .. | head]
//   cl_peel_backward: grad W_0[k][c] += sum_i dZ_0[k][i] x[c][i], grad b_0[k] += sum_i dZ_0[k][i]  (per-workgroup partials, summed in index
//                     order: deterministic), and the gradient of layers 1 .. and of the head copied over from the peeled scaler's.
#include <hip/hip_runtime.h>
#include "cl_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PEEL_T = 256;                 // observations of a chunk = threads of a workgroup

// thread = observation; the layer's weights sit in LDS as [c][WQ] (a broadcast ds_read_b128 per four outputs)
template <int WQ>       // outputs rounded up to 4: 4, 8, 12, 16
__global__ __launch_bounds__(PEEL_T) void peel_forward_kernel(const float* __restrict__ meta_t, int n_pad, int d, int w, const float* __restrict__ mlp,
                                                              long long n_tail, float* __restrict__ u_t, int u_rows, float* __restrict__ mlp_peel,
                                                              float* __restrict__ zero_ptr, int zero_n, const int* __restrict__ stop_flag) {
    if (stop_flag != nullptr && *stop_flag != 0) return;
    extern __shared__ __attribute__((aligned(16))) float sw[];         // [d + 1][WQ]: row d = the bias
    const int tid = threadIdx.x;
    for (int e = tid; e < (d + 1) * WQ; e += PEEL_T) {
        const int c = e / WQ, k = e - c * WQ;
        sw[e] = k < w ? (c < d ? mlp[k * d + c] : mlp[w * d + k]) : 0.0f;
    }
    if (blockIdx.x == 0) {
        // the peeled scaler's parameters: identity kernel, zero bias, then everything behind layer 0 as it stands
        for (int e = tid; e < w * w + w; e += PEEL_T) mlp_peel[e] = (e < w * w && e / w == e % w) ? 1.0f : 0.0f;
        for (long long e = tid; e < n_tail; e += PEEL_T) mlp_peel[w * w + w + e] = mlp[(long long)w * d + w + e];
        for (int e = tid; e < zero_n; e += PEEL_T) zero_ptr[e] = 0.0f;
    }
    __syncthreads();
    const long long i = (long long)blockIdx.x * PEEL_T + tid;
    if (i >= n_pad) return;
    f32x4 acc[WQ / 4];
#pragma unroll
    for (int g = 0; g < WQ / 4; ++g) acc[g] = *reinterpret_cast<const f32x4*>(sw + d * WQ + 4 * g);
    const float* xp = meta_t + i;
    for (int c = 0; c < d; ++c) {
        const float x = xp[(size_t)c * n_pad];
#pragma unroll
        for (int g = 0; g < WQ / 4; ++g) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(sw + c * WQ + 4 * g);
            acc[g][0] = fmaf(wv[0], x, acc[g][0]); acc[g][1] = fmaf(wv[1], x, acc[g][1]);
            acc[g][2] = fmaf(wv[2], x, acc[g][2]); acc[g][3] = fmaf(wv[3], x, acc[g][3]);
        }
    }
#pragma unroll
    for (int g = 0; g < WQ / 4; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (4 * g + t < u_rows) u_t[(size_t)(4 * g + t) * n_pad + i] = 4 * g + t < w ? acc[g][t] : 0.0f;
}

// Weight gradient of the peeled layer on the fp32 MFMA (16 x 16 x 4): D[k][c] += sum over four observations of dZ_0[k][i] x[c][i].  Both
// operands come straight from their feature-major images: lane (row, kk) loads a float4 = four consecutive observations of its row (dZ_0
// row k for A, metadata column c for B), MFMA step t of a 16-observation group contracts observations {i0 + 4 kk + t} -- the same map on
// both sides.  The column behind the last metadata column is a column of ones: its product is the bias gradient.  A wave walks
// 64-observation tiles with a stride, its NB accumulator blocks live in registers for the whole launch; the workgroup's four waves
// add up through LDS and leave ONE partial row [W_0^T (w x d) | b_0 (w)].  Padding observations carry dZ_0 = 0 (the fused kernel's own
// weight gradients rely on the same).
template <int NB>       // 16-column blocks of (metadata columns + the ones column)
__global__ __launch_bounds__(PEEL_T) void peel_wgrad_kernel(const float* __restrict__ meta_t, int n_pad, int d, int w, const float* __restrict__ dz_t,
                                                            float* __restrict__ partials, const int* __restrict__ stop_flag) {
    if (stop_flag != nullptr && *stop_flag != 0) return;
    __shared__ __attribute__((aligned(16))) float red[4][NB][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // row bases: A = dZ_0 row j (zero rows past w: the buffer has cl_mlp_meta_rows(w) rows, all written by the fused kernel or zero), B = column 16 b + j
    const bool arow = j < w;
    const float* pa = dz_t + (size_t)(arow ? j : 0) * n_pad + 4 * q;
    const float* pb[NB];
    int kind[NB];                               // 0: a metadata column, 1: the ones column, 2: nothing
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int c = 16 * b + j;
        kind[b] = c < d ? 0 : (c == d ? 1 : 2);
        pb[b] = meta_t + (size_t)(c < d ? c : 0) * n_pad + 4 * q;
    }
    const int ntile = n_pad / 64;
    for (int tile = blockIdx.x * 4 + wv; tile < ntile; tile += gridDim.x * 4) {
        const size_t i0 = (size_t)tile * 64;
        f32x4 av[4], bv[NB][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            av[s] = *reinterpret_cast<const f32x4*>(pa + i0 + 16 * s);
#pragma unroll
            for (int b = 0; b < NB; ++b) bv[b][s] = *reinterpret_cast<const f32x4*>(pb[b] + i0 + 16 * s);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!arow) av[s] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (kind[b] == 1) bv[b][s] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                else if (kind[b] == 2) bv[b][s] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], bv[b][s][t], acc[b], 0, 0, 0);
            }
        }
    }
    // C layout: lane (j, q), element t = D[k = 4 q + t][c = 16 b + j]
#pragma unroll
    for (int b = 0; b < NB; ++b) *reinterpret_cast<f32x4*>(&red[wv][b][lane][0]) = acc[b];
    __syncthreads();
    const int nout = w * d + w;
    float* part = partials + (size_t)blockIdx.x * nout;
    for (int e = tid; e < NB * 256; e += PEEL_T) {
        const int b = e >> 8, l = (e >> 2) & 63, t = e & 3;
        const int k = 4 * (l >> 4) + t, c = 16 * b + (l & 15);
        if (k < w && c <= d) {
            const float v = red[0][b][l][t] + red[1][b][l][t] + red[2][b][l][t] + red[3][b][l][t];
            part[c < d ? k * d + c : w * d + k] = v;
        }
    }
}

// grad(everything behind layer 0) += the peeled scaler's gradient behind ITS layer 0 (layer 0's own part: cl_reduce_partials of the rows above)
__global__ __launch_bounds__(256) void peel_tail_kernel(const float* __restrict__ grad_peel, long long n_tail, float* __restrict__ grad_tail,
                                                        const int* __restrict__ stop_flag) {
    if (stop_flag != nullptr && *stop_flag != 0) return;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < n_tail) grad_tail[e] += grad_peel[e];
}

// dX = W^T dZ per observation, both feature-major like meta_t (rows of four-row groups, n_pad columns): the step from dL/d(pre-activations
// of a block's first layer) -- what the default scaler's kernels store as dZ_0 -- to dL/d(the block's input activations), which the block
// in front of it in a layer-block chain takes as dH_ext (round 6: the last block of a deep narrow scaler on elbo_lane.hip).  One thread per
// observation, the layer's weights through LDS; reads 4 w_out and writes 4 w_in bytes per observation: HBM-bound by construction.
__global__ __launch_bounds__(256) void chain_dx_kernel(const float* __restrict__ dz, const float* __restrict__ Wt, int n_obs, int n_pad, int w_out,
                                                        int w_in, float* __restrict__ dx, const int* __restrict__ stop_flag) {
    if (stop_flag != nullptr && *stop_flag != 0) return;
    __shared__ float sW[16 * 16];
    for (int i = threadIdx.x; i < w_out * w_in; i += 256) sW[i] = Wt[i];         // Wt[out][in]
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_pad) return;
    float z[16], x[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) z[o] = (o < w_out && n < n_obs) ? dz[(size_t)o * n_pad + n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.0f;
    for (int o = 0; o < w_out; ++o) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fmaf(i < w_in ? sW[o * w_in + i] : 0.0f, z[o], x[i]);
    }
    const int rows = (w_in + 3) & ~3;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < rows) dx[(size_t)i * n_pad + n] = i < w_in ? x[i] : 0.0f;
}

}  // namespace

extern "C" {

int cl_chain_dx(const float* dz0_t, const float* Wt, int n_obs, int n_pad, int w_out, int w_in, float* dx_t, const int* stop_flag, void* stream) {
    if (dz0_t == nullptr || Wt == nullptr || dx_t == nullptr || n_obs < 1 || n_pad < n_obs) return -1;
    if (w_out < 1 || w_out > 15 || w_in < 1 || w_in > 15) return -2;
    (void)hipGetLastError();
    hipLaunchKernelGGL(chain_dx_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dz0_t, Wt, n_obs, n_pad, w_out, w_in, dx_t, stop_flag);
    return (int)hipGetLastError();
}

int cl_peel_supported(int d, int w, int L) {
    return d >= 1 && d <= 79 && w >= 1 && w <= 15 && L >= 1 && d > w;          // (five 16-column blocks hold 79 columns + the ones)
}

int cl_peel_parts(long long n_obs) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long long nchunk = (n_obs + PEEL_T - 1) / PEEL_T;            // (a workgroup's four waves take 64-observation tiles)
    long long g = 2LL * cus;
    if (g > nchunk) g = nchunk;
    return (int)(g < 1 ? 1 : g);
}

int cl_peel_forward(const float* meta_t, int n_obs, int n_pad, int d, int w, int L, const float* mlp, float* u_t, float* mlp_peel,
                    float* zero_ptr, int zero_n, const int* stop_flag, void* stream) {
    if (meta_t == nullptr || mlp == nullptr || u_t == nullptr || mlp_peel == nullptr || n_obs < 1 || n_pad < n_obs) return -1;
    if (zero_n < 0 || (zero_n > 0 && zero_ptr == nullptr)) return -1;
    if (!cl_peel_supported(d, w, L)) return -2;
    const long long n_tail = (long long)(L - 1) * ((long long)w * w + w) + 2 * w + 2;
    const int u_rows = (w + 3) & ~3;                                  // cl_mlp_meta_rows(w)
    const dim3 grid((unsigned)((n_pad + PEEL_T - 1) / PEEL_T));
    const int WQ = (w + 3) & ~3;
    const size_t lds = (size_t)(d + 1) * WQ * sizeof(float);
    (void)hipGetLastError();
    hipStream_t st = (hipStream_t)stream;
#define PEEL_FWD(Q) hipLaunchKernelGGL(peel_forward_kernel<Q>, grid, dim3(PEEL_T), lds, st, meta_t, n_pad, d, w, mlp, n_tail, u_t, u_rows, mlp_peel, zero_ptr, zero_n, stop_flag)
    if (WQ == 4) PEEL_FWD(4);
    else if (WQ == 8) PEEL_FWD(8);
    else if (WQ == 12) PEEL_FWD(12);
    else PEEL_FWD(16);
#undef PEEL_FWD
    return (int)hipGetLastError();
}

int cl_peel_backward(const float* meta_t, int n_obs, int n_pad, int d, int w, int L, const float* dz0_t, const float* grad_peel, float* grad_mlp,
                     float* partials, int nparts, const int* stop_flag, void* stream) {
    if (meta_t == nullptr || dz0_t == nullptr || grad_peel == nullptr || grad_mlp == nullptr || partials == nullptr || n_obs < 1 || n_pad < n_obs) return -1;
    if (!cl_peel_supported(d, w, L)) return -2;
    if (nparts < 1 || nparts > cl_peel_parts(n_obs)) return -1;
    if (n_pad % 64 != 0) return -1;
    const int nout = w * d + w;
    const long long n_tail = (long long)(L - 1) * ((long long)w * w + w) + 2 * w + 2;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    const int NB = (d + 1 + 15) / 16;
#define PEEL_WG(B) hipLaunchKernelGGL(peel_wgrad_kernel<B>, dim3(nparts), dim3(PEEL_T), 0, st, meta_t, n_pad, d, w, dz0_t, partials, stop_flag)
    switch (NB) {
        case 1: PEEL_WG(1); break;
        case 2: PEEL_WG(2); break;
        case 3: PEEL_WG(3); break;
        case 4: PEEL_WG(4); break;
        default: PEEL_WG(5); break;
    }
#undef PEEL_WG
    if (int e = (int)hipGetLastError()) return e;
    if (int e = cl_launch_reduce_partials(partials, nparts, nout, grad_mlp, stop_flag, st)) return e;      // fixed order: deterministic
    hipLaunchKernelGGL(peel_tail_kernel, dim3((unsigned)((n_tail + 255) / 256)), dim3(256), 0, st, grad_peel + (w * w + w), n_tail, grad_mlp + nout, stop_flag);
    return (int)hipGetLastError();
}

}  // extern "C"
