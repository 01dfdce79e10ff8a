// Laue (polychromatic) likelihood path of the ELBO step (gfx950): harmonic deconvolution.
//
// Reference: `ConvolvedLikelihood.convolve / .log_prob`, `LaueBase.call` (careless/models/likelihoods/laue.py:9-47):
//     iconv = scatter_nd(harmonic_id, ipred^T, shape (N,S))^T      -- predictions of the rows that share a harmonic id SUM
//     ll    = base.log_prob(iconv)  over ALL N slots               -- slots >= G keep iconv = 0 and add a constant
// where Iobs / SigIobs are valid in slots [0,G) and padded beyond (careless/io/formatter.py:637-640).
//
// The likelihood of a row depends on the other rows of its harmonic group, so the step is split around the group sum:
//   cl_mlp_forward            scaler loc / sigma per row                            (fused MFMA kernel, forward only)
//   laue_predict_kernel       sample, predict, atomicAdd into iconv[harmonic_id][s]
//   laue_likelihood_kernel    log-prob per slot, NLL partial sums, dNLL/diconv written in place of iconv
//   laue_backward_kernel      broadcast dNLL/diconv back to the rows -> dz_f / image-scale atomics, dL/d(loc, sigma) per row
//   cl_mlp_backward_ext       scaler backward from that dL/d(loc, sigma)            (fused MFMA kernel, MODE 2)
// All three kernels here are streaming (HBM-bound, a few tens of bytes per row and sample).
#include <hip/hip_runtime.h>
#include "cl_math.h"
#include "cl_kernels.h"

namespace {
__device__ __forceinline__ float laue_eta(const cl_laue_args& A, int i, int s) {
    if (A.eta) return A.eta[(size_t)i * A.S + s];
    const uint64_t gidx = A.row_index ? (uint64_t)A.row_index[i] : (uint64_t)(A.obs_offset + i);
    return cl_noise_normal(A.seed, A.step, (uint32_t)s, gidx);
}
__device__ __forceinline__ double wave_sum_d2(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
}  // namespace

__global__ __launch_bounds__(256) void laue_predict_kernel(const cl_laue_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long long)A.n_obs * A.S) return;
    const int i = (int)(p / A.S), s = (int)(p - (long long)i * A.S);
    const int rid = A.refl_id[i];
    float aim = 1.0f;
    if (A.use_img) { const int im = A.image_id[i]; if (im > 0) aim = A.img[im - 1]; }
    const float tq = A.loc[i] + A.sigma[i] * laue_eta(A, i, s) + A.shift;
    const float zf = A.z_f[(size_t)rid * A.S + s];
    const float ipred = aim * tq * zf * zf;
    if (A.ipred_out) A.ipred_out[p] = ipred;
    // harmonic_id NULL: every row is its own slot (monochromatic rows on this path: scalers wider than 64) -- a plain store into a
    // buffer the caller need not clear; otherwise the group's sum (laue.py:24: duplicates add up)
    if (A.harmonic_id == nullptr) A.iconv[p] = ipred;
    else atomicAdd(A.iconv + (size_t)A.harmonic_id[i] * A.S + s, ipred);
}

__global__ __launch_bounds__(256) void laue_likelihood_kernel(const cl_laue_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    // grid-stride over the (slot, sample) pairs: the NLL ends in ONE double atomic per workgroup on one address (they
    // serialise at ~12 ns each), so the grid is kept at a few workgroups per CU instead of one per 256 elements
    const long long total = (long long)A.n_obs * A.S;
    double nll = 0.0;
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    cl_ev11 ev = {1.0f, 0.0f, 0.0f};
    float sg0 = 0.0f, sg1 = 0.0f, sg2 = 0.0f;
    if (A.ev11 != nullptr) {
        ev.sdfac = cl_softplus(A.ev11[0]); ev.sdadd = cl_softplus(A.ev11[1]); ev.sdb = cl_softplus(A.ev11[2]);
        sg0 = cl_sigmoid(A.ev11[0]); sg1 = cl_sigmoid(A.ev11[1]); sg2 = cl_sigmoid(A.ev11[2]);
    }
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(p / A.S);
        float dll, ll;
        if (A.ev11 != nullptr) {
            float gf, gb, ga;
            ll = cl_lik_ev11(A.iconv[p], A.iobs[g], A.sig[g], A.lik_kind, A.dof, A.lik_const, ev, &dll, &gf, &gb, &ga);
            g0 -= gf * A.w_ll * sg0; g1 -= ga * A.w_ll * sg1; g2 -= gb * A.w_ll * sg2;
        } else {
            ll = cl_lik_log_prob(A.iconv[p], A.iobs[g], A.sig[g], A.lik_kind, A.dof, A.lik_const, &dll);
        }
        nll -= (double)ll * (double)A.w_ll;
        A.iconv[p] = -dll * A.w_ll;                  // dNLL / d iconv[g][s]
    }
    __shared__ double sh[4];
    nll = wave_sum_d2(nll);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = nll;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (A.nll_part != nullptr) A.nll_part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];      // deterministic mode: summed in index order by cl_det_reduce
        else atomicAdd(A.scalars + CL_SC_NLL, sh[0] + sh[1] + sh[2] + sh[3]);
    }
    if (A.ev11 != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { g0 += __shfl_xor(g0, off); g1 += __shfl_xor(g1, off); g2 += __shfl_xor(g2, off); }
        if ((threadIdx.x & 63) == 0) {
            if (A.ev11_part != nullptr) {            // deterministic mode: this wave's slot, summed in index order by cl_det_reduce
                float* slot = A.ev11_part + 3 * (4 * (size_t)blockIdx.x + (threadIdx.x >> 6));
                slot[0] = g0; slot[1] = g1; slot[2] = g2;
            } else { atomicAdd(A.d_ev11 + 0, g0); atomicAdd(A.d_ev11 + 1, g1); atomicAdd(A.d_ev11 + 2, g2); }
        }
    }
}

// One thread per (row, sample): the S samples of a row are consecutive lanes, so the amplitude-gradient atomics of a row hit S
// consecutive floats of dz_f (one request) -- with a thread per row and a loop over the samples every instruction sent 64 lanes to 64
// different lines, the pattern the memory side serialises (0.52 ms per step at 2 M rows x 4 samples; csrc/elbo_lane.hip has the same
// story).  The row's sums over its samples (dL/dloc, dL/dsigma, image-scale gradient) are segmented lane reductions; the first lane
// of a row's segment inside a wave adds them into dO (zeroed by the caller side of this launch: see cl_launch_laue_backward).
__global__ __launch_bounds__(256) void laue_backward_kernel(const cl_laue_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool act = p < (long long)A.n_obs * A.S;   // no early return: the reductions below are wave-wide
    const int i = act ? (int)(p / A.S) : -1, s = act ? (int)(p - (long long)i * A.S) : 0;
    int im = 0;
    float dloc = 0.0f, dsig = 0.0f, da = 0.0f;
    if (act) {
        const int rid = A.refl_id[i], hid = A.harmonic_id != nullptr ? A.harmonic_id[i] : i;
        float aim = 1.0f;
        if (A.use_img) { im = A.image_id[i]; if (im > 0) aim = A.img[im - 1]; }
        const float loc = A.loc[i], sigma = A.sigma[i];
        const float eta = laue_eta(A, i, s);
        const float tq = loc + sigma * eta + A.shift;
        const float zf = A.z_f[(size_t)rid * A.S + s];
        const float gi = A.iconv[(size_t)hid * A.S + s];
        const float dzs = gi * zf * zf;
        atomicAdd(A.dz_f + (size_t)rid * A.S + s, gi * aim * tq * 2.0f * zf);
        const float dt = dzs * aim;
        dloc = dt;
        dsig = dt * eta;
        da = dzs * tq;
    }
    if (A.dO == nullptr) return;          // (wave-uniform) a frozen scaling model: nobody takes dL/d(loc, sigma) or the image scales' gradient (round 6)
    // segmented sums over the lanes of one row (they are consecutive)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int r2 = __shfl_down(i, off);
        const float a = __shfl_down(dloc, off), b = __shfl_down(dsig, off), c = __shfl_down(da, off);
        if (lane + off < 64 && r2 == i) { dloc += a; dsig += b; da += c; }
    }
    const int prev = __shfl_up(i, 1);
    const bool head = act && (lane == 0 || prev != i);
    if (head) {
        atomicAdd(A.dO + 2 * (size_t)i, dloc);       // (a row that straddles two waves has two heads)
        atomicAdd(A.dO + 2 * (size_t)i + 1, dsig);
    }
    if (A.use_img) {
        // rows are (nearly) ordered by image, so a wave usually holds one image: reduce the row heads' sums in the wave and issue one atomic
        const int key = (head && im > 0) ? im : 0;   // 0: nothing to add (image 0 is pinned to 1, image.py:23-25)
        int kmax = key;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off));
        if (__all(key == kmax || key == 0)) {
            float v = key ? da : 0.0f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && kmax > 0) atomicAdd(A.d_img + (kmax - 1), v);
        } else {
            cl_image_grad_segments(A.d_img, key, da, key > 0, lane);       // the wave spans several images: one reduction + atomic per image
        }
    }
}

// Rows that are their own slot (harmonic_id NULL: monochromatic rows on the layer-by-layer path of scalers wider than 64): predict ->
// likelihood -> gradient of a (row, sample) pair need nothing from another thread, so the three kernels above collapse into ONE pass
// (round 4: 0.31 -> 0.2 ms per step at 2 M rows x 4 samples -- the pair's loads, its Philox draw and the iconv round trip happened three
// times).  Thread = (row, sample) as in laue_backward_kernel (the samples of a row on consecutive lanes: one request per row for the dz_f
// atomics); grid-stride with a bounded grid, every thread running the same number of rounds (the reductions are wave-wide), so the NLL
// ends in one double atomic per workgroup as in laue_likelihood_kernel.  When S divides 64 a row never straddles two waves: its sums
// over the samples are complete in the first lane of its segment and dO is STORED (no memset, no atomics).
__global__ __launch_bounds__(256) void slot_rows_kernel(const cl_laue_args A, int dO_store) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const long long total = (long long)A.n_obs * A.S, stride = (long long)gridDim.x * blockDim.x;
    const long long rounds = (total + stride - 1) / stride;
    const int lane = threadIdx.x & 63;
    double nll = 0.0;
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    cl_ev11 ev = {1.0f, 0.0f, 0.0f};
    float sg0 = 0.0f, sg1 = 0.0f, sg2 = 0.0f;
    if (A.ev11 != nullptr) {
        ev.sdfac = cl_softplus(A.ev11[0]); ev.sdadd = cl_softplus(A.ev11[1]); ev.sdb = cl_softplus(A.ev11[2]);
        sg0 = cl_sigmoid(A.ev11[0]); sg1 = cl_sigmoid(A.ev11[1]); sg2 = cl_sigmoid(A.ev11[2]);
    }
    // everything a pair reads, requested for TWO rounds before the first one is worked on: the kernel is a chain of dependent loads (row ->
    // reflection -> amplitude) with all eight wave slots of a SIMD taken -- memory-level parallelism has to come from inside the thread
    struct Pair { int i, s, rid, im; float loc, sigma, iobs, sig, aim, zf; bool act; };
    auto fetch = [&](long long p) -> Pair {
        Pair P;
        P.act = p < total;
        P.i = P.act ? (int)(p / A.S) : -1;
        P.s = P.act ? (int)(p - (long long)P.i * A.S) : 0;
        const int ic = P.act ? P.i : 0;                  // (inactive lanes read row 0: no branch around the loads)
        P.rid = A.refl_id[ic];
        P.im = A.use_img ? A.image_id[ic] : 0;
        P.loc = A.loc[ic]; P.sigma = A.sigma[ic]; P.iobs = A.iobs[ic]; P.sig = A.sig[ic];
        P.aim = (P.im > 0) ? A.img[P.im - 1] : 1.0f;
        P.zf = A.z_f[(size_t)P.rid * A.S + P.s];
        return P;
    };
    auto work = [&](const Pair& P, long long p) {
        const int i = P.i, s = P.s;
        int im = 0;
        float dloc = 0.0f, dsig = 0.0f, da = 0.0f;
        if (P.act) {
            im = P.im;
            const float aim = P.aim, zf = P.zf;
            const float eta = laue_eta(A, i, s);
            const float tq = P.loc + P.sigma * eta + A.shift;
            const float ipred = aim * tq * zf * zf;
            if (A.ipred_out) A.ipred_out[p] = ipred;
            float dll, ll;
            if (A.ev11 != nullptr) {
                float gf, gb, ga;
                ll = cl_lik_ev11(ipred, P.iobs, P.sig, A.lik_kind, A.dof, A.lik_const, ev, &dll, &gf, &gb, &ga);
                g0 -= gf * A.w_ll * sg0; g1 -= ga * A.w_ll * sg1; g2 -= gb * A.w_ll * sg2;
            } else {
                ll = cl_lik_log_prob(ipred, P.iobs, P.sig, A.lik_kind, A.dof, A.lik_const, &dll);
            }
            nll -= (double)ll * (double)A.w_ll;
            const float gi = -dll * A.w_ll;             // dNLL / d ipred
            const float dzs = gi * zf * zf;
            const float dzf = gi * aim * tq * 2.0f * zf;
            if (A.dzf_obs != nullptr) A.dzf_obs[(size_t)(A.det_slot != nullptr ? A.det_slot[i] : i) * A.S + s] = dzf;      // (summed per reflection by cl_det_reduce)
            else atomicAdd(A.dz_f + (size_t)P.rid * A.S + s, dzf);
            const float dt = dzs * aim;
            dloc = dt;
            dsig = dt * eta;
            da = dzs * tq;
        }
        bool head;
        if (dO_store && A.S <= 16) {                     // (wave-uniform) a row = an aligned group of S lanes of one 16-lane row: sums on the DPP network
            dloc = cl_group_sum(dloc, A.S); dsig = cl_group_sum(dsig, A.S); da = cl_group_sum(da, A.S);
            head = P.act && s == 0;
        } else {
            // segmented sums over the lanes of one row (they are consecutive)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int r2 = __shfl_down(i, off);
                const float a = __shfl_down(dloc, off), b = __shfl_down(dsig, off), c = __shfl_down(da, off);
                if (lane + off < 64 && r2 == i) { dloc += a; dsig += b; da += c; }
            }
            const int prev = __shfl_up(i, 1);
            head = P.act && (lane == 0 || prev != i);
        }
        if (head) {
            if (dO_store) { A.dO[2 * (size_t)i] = dloc; A.dO[2 * (size_t)i + 1] = dsig; }
            else { atomicAdd(A.dO + 2 * (size_t)i, dloc); atomicAdd(A.dO + 2 * (size_t)i + 1, dsig); }      // (a row that straddles two waves has two heads)
        }
        if (A.use_img && A.dimg_obs != nullptr) {
            if (head) A.dimg_obs[i] = im > 0 ? da : 0.0f;      // deterministic mode: summed per image, in row order, by cl_det_reduce
        } else if (A.use_img) {
            // rows are (nearly) ordered by image, so a wave usually holds one image: one wave sum, one atomic
            const bool take = head && im > 0;            // (image 0 is pinned to 1, image.py:23-25)
            const unsigned long long m = __ballot(take);
            if (m != 0ull) {                             // (wave-uniform)
                const int im0 = __builtin_amdgcn_readlane(im, __builtin_ctzll(m));
                if (__all(!take || im == im0)) {
                    const float v = cl_wave_sum(take ? da : 0.0f);
                    if (lane == 0) atomicAdd(A.d_img + (im0 - 1), v);
                } else {
                    cl_image_grad_segments(A.d_img, im, da, take, lane);
                }
            }
        }
    };
    const long long p0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long r = 0; r < rounds; r += 2) {
        const long long pa = r * stride + p0, pb = pa + stride;      // (pb >= total in a last odd round: an inactive pair)
        const Pair Pa = fetch(pa), Pb = fetch(pb);
        work(Pa, pa);
        work(Pb, pb);
    }
    __shared__ double sh[4];
    nll = wave_sum_d2(nll);
    if (lane == 0) sh[threadIdx.x >> 6] = nll;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (A.nll_part != nullptr) A.nll_part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
        else atomicAdd(A.scalars + CL_SC_NLL, sh[0] + sh[1] + sh[2] + sh[3]);
    }
    if (A.ev11 != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { g0 += __shfl_xor(g0, off); g1 += __shfl_xor(g1, off); g2 += __shfl_xor(g2, off); }
        if (lane == 0) {
            if (A.ev11_part != nullptr) {
                float* slot = A.ev11_part + 3 * (4 * (size_t)blockIdx.x + (threadIdx.x >> 6));
                slot[0] = g0; slot[1] = g1; slot[2] = g2;
            } else { atomicAdd(A.d_ev11 + 0, g0); atomicAdd(A.d_ev11 + 1, g1); atomicAdd(A.d_ev11 + 2, g2); }
        }
    }
}

static int laue_check(const cl_laue_args& a) {
    if (a.n_obs <= 0 || a.S <= 0 || a.refl_id == nullptr || a.loc == nullptr || a.sigma == nullptr ||
        a.z_f == nullptr || a.iconv == nullptr)
        return -1;
    if (a.use_img && (a.image_id == nullptr || a.img == nullptr)) return -1;
    return 0;
}

int cl_launch_laue_predict(const cl_laue_args& a, hipStream_t st) {
    if (int e = laue_check(a)) return e;
    const long long n = (long long)a.n_obs * a.S;
    (void)hipGetLastError();
    hipLaunchKernelGGL(laue_predict_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
int cl_launch_laue_likelihood(const cl_laue_args& a, hipStream_t st) {
    // needs the slot arrays only (it is also called on the padded slots alone by the single-pass path)
    if (a.n_obs <= 0 || a.S <= 0 || a.iconv == nullptr || a.iobs == nullptr || a.sig == nullptr || a.scalars == nullptr) return -1;
    const long long n = (long long)a.n_obs * a.S;
    (void)hipGetLastError();
    long long blocks = (n + 255) / 256;
    if (blocks > CL_LAUE_LIK_MAX_BLOCKS) blocks = CL_LAUE_LIK_MAX_BLOCKS;
    hipLaunchKernelGGL(laue_likelihood_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
int cl_launch_slot_rows(const cl_laue_args& a, hipStream_t st) {
    if (int e = laue_check(a)) return e;
    if (a.harmonic_id != nullptr) return -2;           // rows that share slots need the three passes (group sums between them)
    if (a.iobs == nullptr || a.sig == nullptr || a.scalars == nullptr || a.dz_f == nullptr || a.dO == nullptr || (a.use_img && a.d_img == nullptr)) return -1;
    if (a.dzf_obs != nullptr) {          // deterministic mode: stores per (row, sample) / row / workgroup; a row's samples must sit inside one wave
        if (64 % a.S != 0) return -2;
        if (a.nll_part == nullptr || (a.use_img && a.dimg_obs == nullptr) || (a.ev11 != nullptr && a.ev11_part == nullptr)) return -1;
    }
    (void)hipGetLastError();
    const int store = (64 % a.S == 0) ? 1 : 0;
    if (!store) {
        hipError_t e = hipMemsetAsync(a.dO, 0, sizeof(float) * 2 * (size_t)a.n_obs, st);
        if (e != hipSuccess) return (int)e;
    }
    const long long n = (long long)a.n_obs * a.S;
    long long blocks = (n + 255) / 256;
    if (blocks > CL_LAUE_LIK_MAX_BLOCKS) blocks = CL_LAUE_LIK_MAX_BLOCKS;
    hipLaunchKernelGGL(slot_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, store);
    return (int)hipGetLastError();
}
int cl_launch_laue_backward(const cl_laue_args& a, hipStream_t st) {
    if (int e = laue_check(a)) return e;
    if (a.dz_f == nullptr || (a.dO != nullptr && a.use_img && a.d_img == nullptr)) return -1;
    (void)hipGetLastError();
    // dO receives the rows' sums by atomics: cleared here, on the same stream (part of the call).  dO NULL (round 6): the scaling model
    // is frozen -- only dz_f is taken, no clearing, no per-row reductions
    if (a.dO != nullptr) {
        hipError_t e = hipMemsetAsync(a.dO, 0, sizeof(float) * 2 * (size_t)a.n_obs, st);
        if (e != hipSuccess) return (int)e;
    }
    const long long n = (long long)a.n_obs * a.S;
    hipLaunchKernelGGL(laue_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
