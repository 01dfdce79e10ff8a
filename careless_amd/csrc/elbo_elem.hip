// Per-reflection and per-parameter kernels of the ELBO step (gfx950): HBM-streaming, one pass each.
//
//   tn_forward_kernel     q(F) parameter transform + reparameterised truncated-normal sample z_f (S,R), and the
//                         sample-based KL partial sums  sum_s [log q(z) - log p(z)]
//                         [surrogate_posteriors.py:50-53,104-131; variational.py:123-139,154,173; wilson.py:50-57]
//   tn_backward_kernel    chain rule from dL/dz_f (data term, accumulated by the fused MLP kernel) plus the KL
//                         term's own derivatives back to the raw q parameters (a = log loc, b = log(scale - eps))
//   grad_sqnorm_kernel    global (and per-tensor) squared L2 norm of the flat gradient   [variational.py:205]
//   adam_kernel           non-finite -> 0, optional clipping, tf_keras Adam update       [variational.py:208-209, manager.py:494-501]
//   finalize_kernel       per-step history record + the sticky stop flag               [variational.py:262-274]
// Each kernel is bandwidth-trivial next to the fused MLP kernel (R = N/32); they exist to keep the whole step on
// the device with no host round trip.
#include <hip/hip_runtime.h>
#include "cl_math.h"
#include "cl_kernels.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// block-wide double sum -> one atomic per block
__device__ __forceinline__ void block_atomic_add_d(double v, double* dst) {
    __shared__ double sh[16];
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
        atomicAdd(dst, s);
    }
}

// uniform of (reflection h, sample s); `blk` caches the Philox block shared by samples 4k .. 4k+3 across loop iterations
__device__ __forceinline__ float tn_uniform(const cl_tn_args& A, int h, int s, cl_u32x4& blk) {
    if (A.u_f) return A.u_f[(size_t)h * A.S + s];
    if ((s & 3) == 0) blk = cl_noise_uniform_block(A.seed, A.step, (uint32_t)s >> 2, (uint64_t)h);
    return cl_noise_uniform_pick(blk, (uint32_t)s);
}

// correlation r of reflection h with its parent: fixed (dw_r) or sigmoid of the trainable per-ASU value
__device__ __forceinline__ float dw_r_of(const cl_tn_args& A, int h) {
    return A.dw_r_raw ? cl_sigmoid(A.dw_r_raw[A.asu_ids[h]]) : A.dw_r[h];
}

// log-density of the Wilson prior and its z-derivative for reflection h
__device__ __forceinline__ float prior_lp(const cl_tn_args& A, int h, float z, float* dlp_dz) {
    const bool c = A.centric[h] != 0;
    const float es = A.es[h];
    *dlp_dz = cl_wilson_dlog_prob_dz(z, c, es);
    return cl_wilson_log_prob(z, c, es);
}

}  // namespace

__global__ __launch_bounds__(256) void tn_forward_kernel(const cl_tn_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const bool part = A.r_end > A.r_begin;               // an owned reflection range (reflection-owner data parallelism)
    const int h = (part ? A.r_begin : 0) + blockIdx.x * blockDim.x + threadIdx.x;
    // the step's accumulators, cleared on the way (instead of a memset launch in front of this one): the flat gradient + scalar block
    // by all threads, the dz_f rows of this launch's reflections by their threads
    if (A.zero_ptr != nullptr) {
        const long long n4 = A.zero_n >> 2;              // (16-byte aligned, a multiple of four floats: the caller's workspace layout)
        f32x4_t* z4 = reinterpret_cast<f32x4_t*>(A.zero_ptr);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) z4[i] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        for (long long i = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.zero_n; i += (long long)gridDim.x * blockDim.x) A.zero_ptr[i] = 0.0f;
    }
    double kl = 0.0;
    if (h < (part ? A.r_end : A.R)) {
        if (A.zero_dzf != nullptr)
            for (int s = 0; s < A.S; ++s) A.zero_dzf[(size_t)h * A.S + s] = 0.0f;
        const float a = A.q_loc_raw[h], b = A.q_scale_raw[h], low = A.low[h];
        const bool in_kl = (h >= A.kl_begin && h < A.kl_end);
        const bool dw_child = (A.prior_kind == CL_PRIOR_DOUBLE_WILSON_) && (A.root[h] == 0);
        cl_u32x4 blk = {};
        for (int s = 0; s < A.S; ++s) {
            const cl_tn_elem t = cl_tn_sample(a, b, low, A.high, A.eps, tn_uniform(A, h, s, blk));
            A.z_f[(size_t)h * A.S + s] = t.z;
            if (in_kl) {
                float dlp;
                const float lq = cl_tn_log_prob(t);
                // non-root reflections of a double-Wilson model get their conditional prior in dw_forward_kernel (needs z_parent)
                const float lp = dw_child ? 0.0f : prior_lp(A, h, t.z, &dlp);
                kl += (double)(lq - lp);
            }
        }
    }
    if (A.kl_part != nullptr) {
        // one store per workgroup; the step's cl_tn_backward adds the parts up (no atomics, fixed order)
        __shared__ double sh[4];
        const double w = wave_sum_d(kl * (double)A.w_kl);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) A.kl_part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    } else {
        block_atomic_add_d(kl * (double)A.w_kl, A.scalars + CL_SC_KL);
    }
}

__global__ __launch_bounds__(256) void tn_backward_kernel(const cl_tn_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const bool part = A.r_end > A.r_begin;
    const int nb_tn = ((part ? A.r_end - A.r_begin : A.R) + 255) / 256;      // workgroups of the reflections; the rest carry the reduction
    if ((int)blockIdx.x >= nb_tn) {
        cl_reduce_partials_block(A.red_partials, A.red_nparts, A.red_P, A.red_out, (int)blockIdx.x - nb_tn);
        return;
    }
    if (A.kl_part != nullptr && blockIdx.x == 0) {
        // the KL sums the forward launch(es) left per workgroup (same grid; the double-Wilson pass always covers all R): added up
        // here, in index order
        const int nb = nb_tn;
        double t = 0.0;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) t += A.kl_part[i];
        if (A.kl_part_dw != nullptr && A.prior_kind == CL_PRIOR_DOUBLE_WILSON_)
            for (int i = threadIdx.x; i < (A.R + 255) / 256; i += blockDim.x) t += A.kl_part_dw[i];
        __shared__ double sh[4];
        const double w = wave_sum_d(t);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) A.scalars[CL_SC_KL] += (sh[0] + sh[1]) + (sh[2] + sh[3]);
    }
    const int h = (part ? A.r_begin : 0) + blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= (part ? A.r_end : A.R)) return;
    const float a = A.q_loc_raw[h], b = A.q_scale_raw[h], low = A.low[h];
    const bool in_kl = (h >= A.kl_begin && h < A.kl_end);
    const float wkl = in_kl ? A.w_kl * A.kl_grad_mult : 0.0f;
    const bool dw_child = (A.prior_kind == CL_PRIOR_DOUBLE_WILSON_) && (A.root[h] == 0);
    float gloc = 0.0f, gscale = 0.0f, loc = 0.0f, scale = 0.0f;
    cl_u32x4 blk = {};
    for (int s = 0; s < A.S; ++s) {
        const cl_tn_elem t = cl_tn_sample(a, b, low, A.high, A.eps, tn_uniform(A, h, s, blk));
        loc = t.loc; scale = t.scale;
        float dq_dz, dq_dloc, dq_dscale, dp_dz;
        cl_tn_log_prob_grads(t, &dq_dz, &dq_dloc, &dq_dscale);
        if (dw_child) {
            const int par = A.parent_ids[h];
            const float zp = (par >= 0) ? A.z_f[(size_t)par * A.S + s] : 0.0f;
            float dzp, dr;
            (void)cl_dw_log_prob(t.z, zp, par >= 0, dw_r_of(A, h), A.centric[h] != 0, A.es[h], &dp_dz, &dzp, &dr);
        } else {
            (void)prior_lp(A, h, t.z, &dp_dz);
        }
        const float gz = A.dz_f[(size_t)h * A.S + s] + wkl * (dq_dz - dp_dz);
        gloc += gz * t.dz_dloc + wkl * dq_dloc;
        gscale += gz * t.dz_dscale + wkl * dq_dscale;
    }
    // raw parameters: loc = exp(a), scale = exp(b) + eps
    A.d_loc_raw[h] += gloc * loc;
    A.d_scale_raw[h] += gscale * (scale - A.eps);
}

// Double-Wilson conditional prior of the non-root reflections: -log p(z_h | z_parent) into the KL, and its derivative
// w.r.t. the parent's sample scattered into dz_f (the child's own derivative is taken in tn_backward_kernel).
__global__ __launch_bounds__(256) void dw_forward_kernel(const cl_tn_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    double kl = 0.0;
    float gr = 0.0f;                     // dL/dr of this reflection (trainable r only)
    int asu = -1;
    if (h < A.R && A.root[h] == 0 && h >= A.kl_begin && h < A.kl_end) {
        const int par = A.parent_ids[h];
        const float r = dw_r_of(A, h), es = A.es[h];
        const bool c = A.centric[h] != 0;
        const float wg = A.w_kl * A.kl_grad_mult;
        for (int s = 0; s < A.S; ++s) {
            const float z = A.z_f[(size_t)h * A.S + s];
            const float zp = (par >= 0) ? A.z_f[(size_t)par * A.S + s] : 0.0f;
            float dz, dzp, dr;
            const float lp = cl_dw_log_prob(z, zp, par >= 0, r, c, es, &dz, &dzp, &dr);
            kl -= (double)lp;
            gr -= wg * dr;
            if (par >= 0 && A.dw_child_seg == nullptr) atomicAdd(A.dz_f_out + (size_t)par * A.S + s, -wg * dzp);      // (deterministic mode: dw_parent_pull_kernel)
        }
        if (A.dw_r_raw) { asu = A.asu_ids[h]; gr *= r * (1.0f - r); }      // d sigmoid(raw) / d raw
    }
    if (A.dw_r_raw) {
        // reflections are ordered by ASU, so a wave almost always holds one ASU: wave-reduce, one atomic per wave
        const int a0 = __builtin_amdgcn_readfirstlane(asu);
        if (__all(asu == a0 || asu < 0)) {
            float v = (asu >= 0) ? gr : 0.0f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            const int any = __builtin_amdgcn_readfirstlane(__any(asu >= 0) ? 1 : 0);
            // a0 may be -1 when lane 0 is inactive: take the ASU of the first active lane instead
            int au = asu;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) au = max(au, __shfl_xor(au, off));
            if (any && (threadIdx.x & 63) == 0) atomicAdd(A.d_dw_r_raw + au, v);
        } else if (asu >= 0) {
            atomicAdd(A.d_dw_r_raw + asu, gr);
        }
    }
    if (A.kl_part_dw != nullptr && A.kl_part != nullptr) {          // (kl_part: the step's cl_tn_backward will add the parts up)
        __shared__ double sh[4];
        const double w = wave_sum_d(kl * (double)A.w_kl);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
        __syncthreads();
        if (threadIdx.x == 0) A.kl_part_dw[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    } else {
        block_atomic_add_d(kl * (double)A.w_kl, A.scalars + CL_SC_KL);
    }
}

// ---------------------------------------------------------------------------------------------------------
// gradient norm, Adam
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int seg_of(const int* __restrict__ seg_off, int nseg, int i) {
    int lo = 0, hi = nseg;               // seg_off has nseg + 1 entries
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ g, int n, const int* __restrict__ seg_off,
                                                          int nseg, double* __restrict__ seg_sq, double* scalars,
                                                          const unsigned char* __restrict__ frozen, const int* stop_flag) {
    if (stop_flag != nullptr && *stop_flag != 0) return;
    double acc = 0.0, sane = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        // (a tensor that is not trainable is not among the gradients the reference takes the norm of: tape.gradient(loss,
        //  self.trainable_variables), variational.py:201-205)
        if (frozen != nullptr && frozen[seg_of(seg_off, nseg, i)] != 0) continue;
        const float v = g[i];
        const double v2 = (double)v * (double)v;
        acc += v2;                                           // raw: a NaN gradient gives a NaN norm (variational.py:205)
        const double s2 = isfinite(v) ? v2 : 0.0;           // what the optimizer's clipping sees (after :208)
        sane += s2;
        if (seg_sq != nullptr) atomicAdd(seg_sq + seg_of(seg_off, nseg, i), s2);
    }
    block_atomic_add_d(acc, scalars + CL_SC_GNORM2);
    __syncthreads();
    block_atomic_add_d(sane, scalars + CL_SC_GNORM2_SANE);
}

// Reflection-owner data parallelism: squared norm of this rank's own part of the surrogate-posterior gradient -- d a and d b of
// reflections [r0, r1) -- as four floats of the step's message (raw, sanitised, sanitised per tensor).  Double accumulation through
// scratch[0..3]; the last workgroup to arrive (ticket in scratch[4]) converts.  The caller zeroes scratch once per step.
__global__ __launch_bounds__(256) void owner_qnorm_kernel(const float* __restrict__ g, int R, int r0, int r1, float* __restrict__ out,
                                                          double* scratch, const int* stop_flag) {
    if (stop_flag != nullptr && *stop_flag != 0) return;
    const int nr = r1 - r0;
    double raw = 0.0, sa = 0.0, sb = 0.0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < 2 * nr; t += gridDim.x * blockDim.x) {
        const bool isb = t >= nr;
        const float v = g[isb ? R + r0 + (t - nr) : r0 + t];
        const double v2 = (double)v * (double)v;
        raw += v2;
        const double s2 = isfinite(v) ? v2 : 0.0;
        if (isb) sb += s2; else sa += s2;
    }
    block_atomic_add_d(raw, scratch + 0);
    __syncthreads();
    block_atomic_add_d(sa, scratch + 2);
    __syncthreads();
    block_atomic_add_d(sb, scratch + 3);
    __shared__ unsigned last;
    if (threadIdx.x == 0) {
        // release / acquire at agent scope on the ticket: the block's three sums are visible to whoever draws the last ticket (on a
        // multi-XCD part relaxed ordering guarantees nothing across L2s).  At most 64 workgroups take a ticket, so the L2 write-back
        // the release implies costs microseconds at worst (the same fence per workgroup of a 1 221-workgroup launch: tn_forward
        // 20 -> 30 us, which is why the big launches store per-workgroup parts instead)
        last = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(scratch + 4), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (last == gridDim.x - 1 && threadIdx.x == 0) {
        const double qa = atomicAdd(scratch + 2, 0.0), qb = atomicAdd(scratch + 3, 0.0), qr = atomicAdd(scratch + 0, 0.0);
        scratch[1] = qa + qb;
        out[0] = (float)qr;
        out[1] = (float)(qa + qb);
        out[2] = (float)qa;
        out[3] = (float)qb;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(const cl_adam_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    float gscale = 1.0f;
    if (A.global_clipnorm > 0.0f) {
        // tf.clip_by_global_norm over the sanitised gradients [3P]: g * clip / max(norm, clip)
        const float nrm = (float)sqrt(A.scalars[CL_SC_GNORM2_SANE]);
        gscale = A.global_clipnorm / fmaxf(nrm, A.global_clipnorm);
    }
    double acc = 0.0, sane = 0.0;
    // the indices this call updates: the whole vector, or (reflection-owner data parallelism) up to three ranges -- the rank's own
    // reflections' a and b, whose norm over all ranks arrives in norm_extra, and the replicated tail
    const int len0 = A.n_ranges > 0 ? A.range_end[0] - A.range_begin[0] : A.n;
    const int len1 = A.n_ranges > 1 ? A.range_end[1] - A.range_begin[1] : 0;
    const int len2 = A.n_ranges > 2 ? A.range_end[2] - A.range_begin[2] : 0;
    const int total = len0 + len1 + len2;
    if (A.norm_out != nullptr && A.norm_extra != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        acc = (double)A.norm_extra[0];
        sane = (double)A.norm_extra[1];
    }
    // Four elements per thread and round, all their loads issued before the first store: p / m / v / g may alias as far as the compiler
    // knows, so the one-element loop was a chain of dependent memory round trips (15 - 23 us for 6.4e5 parameters; the arithmetic is
    // nothing).  Elements of a round are a grid stride apart: coalesced as before.
    const int stride = gridDim.x * blockDim.x;
    constexpr int U = 4;
    for (int t0 = blockIdx.x * blockDim.x + threadIdx.x; t0 < total; t0 += U * stride) {
        int ii[U], rk[U];
        float g_[U], m_[U], v_[U], p_[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u * stride;
            on[u] = t < total;
            int i = on[u] ? t : 0, r = 0;
            if (A.n_ranges > 0 && on[u]) {
                if (t < len0) { i = A.range_begin[0] + t; }
                else if (t < len0 + len1) { i = A.range_begin[1] + (t - len0); r = 1; }
                else { i = A.range_begin[2] + (t - len0 - len1); r = 2; }
            }
            ii[u] = i; rk[u] = r;
            g_[u] = A.g[i]; m_[u] = A.m[i]; v_[u] = A.v[i]; p_[u] = A.p[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!on[u]) continue;
            const int i = ii[u];
            float g = g_[u];
            // (a frozen tensor is neither updated nor part of the norm: the reference's gradients are those of trainable_variables)
            if (A.frozen != nullptr && A.frozen[seg_of(A.seg_off, A.nseg, i)] != 0) continue;
            if (A.norm_out != nullptr && rk[u] >= A.norm_skip_ranges) {    // fused tf.linalg.global_norm (variational.py:205)
                const double v2 = (double)g * (double)g;
                acc += v2;
                sane += isfinite(g) ? v2 : 0.0;
            }
            if (!isfinite(g)) g = 0.0f;                                   // variational.py:208
            if (A.clipnorm > 0.0f) {                                      // per-tensor tf.clip_by_norm [3P]
                const float nrm = (float)sqrt(A.seg_sq[seg_of(A.seg_off, A.nseg, i)]);
                if (nrm > A.clipnorm) g *= A.clipnorm / nrm;
            }
            g *= gscale;
            if (A.clipvalue > 0.0f) g = fminf(fmaxf(g, -A.clipvalue), A.clipvalue);
            float m = m_[u], v = v_[u];
            m += (g - m) * (1.0f - A.beta1);
            v += (g * g - v) * (1.0f - A.beta2);
            A.m[i] = m;
            A.v[i] = v;
            A.p[i] = p_[u] - m * A.alpha / (sqrtf(v) + A.adam_eps);
        }
    }
    if (A.norm_out != nullptr && A.norm_part != nullptr) {
        // one store pair per workgroup; cl_step_finalize adds them up (the whole grid is resident at once: its same-address atomics
        // would queue up at the end of the kernel)
        __shared__ double sh[2][4];
        const double wa = wave_sum_d(acc), ws = wave_sum_d(sane);
        if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = wa; sh[1][threadIdx.x >> 6] = ws; }
        __syncthreads();
        if (threadIdx.x < 2) A.norm_part[2 * blockIdx.x + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
    } else if (A.norm_out != nullptr) {
        block_atomic_add_d(acc, A.norm_out + CL_SC_GNORM2);
        __syncthreads();
        block_atomic_add_d(sane, A.norm_out + CL_SC_GNORM2_SANE);
    }
}

// one wave: add up what cl_adam_step left per workgroup (norm_part), then lane 0 writes the history record of this step and updates
// the sticky stop flag
__global__ void finalize_kernel(double* scalars, float kl_weight_or_one, double* history, int step_index,
                                int hist_stride, int* stop_flag, const double* __restrict__ norm_part, int n_norm_part) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    double* rec = history + (size_t)step_index * hist_stride;
    if (stop_flag != nullptr && *stop_flag != 0) {
        if (threadIdx.x == 0) {
            rec[0] = rec[1] = rec[2] = rec[3] = 0.0;
            rec[4] = 1.0;                  // skipped
        }
        return;
    }
    double gn2 = scalars[CL_SC_GNORM2];
    if (norm_part != nullptr) {
        double a = 0.0, b = 0.0;
        for (int i = threadIdx.x; i < n_norm_part; i += 64) { a += norm_part[2 * i]; b += norm_part[2 * i + 1]; }
        a = wave_sum_d(a); b = wave_sum_d(b);
        gn2 += a;
        if (threadIdx.x == 0) { scalars[CL_SC_GNORM2] = gn2; scalars[CL_SC_GNORM2_SANE] += b; }
    }
    if (threadIdx.x != 0) return;
    const double nll = scalars[CL_SC_NLL], kl = scalars[CL_SC_KL];
    const double gn = sqrt(gn2);
    rec[0] = nll + (double)kl_weight_or_one * kl;   // loss
    rec[1] = kl;                                    // "F KLDiv"
    rec[2] = nll;                                   // "NLL"
    rec[3] = gn;                                    // "Grad Norm"
    rec[4] = 0.0;
    if (stop_flag != nullptr && !isfinite(gn)) *stop_flag = 1;     // variational.py:271-274
}

// Output step (include/careless_hip.h: cl_tn_moments): mean, standard deviation and fourth raw moment of q(F_h) = Normal(loc, scale)
// truncated to [low, high], from the raw vectors; reference surrogate_posteriors.py:55-73 (_tf_moment_4: the closed form of Orjebin's
// note) and the TFP truncated-normal moments behind .mean() / .stddev().  fp64: once per run over R values, and the caller subtracts
// <F^2>^2 from <F^4>.  With alpha = (low - loc) / scale <= ~0 (loc = exp(a) > 0 >= low) the normaliser Phi(beta) - Phi(alpha) is
// taken as (erfc(alpha / sqrt 2) - erfc(beta / sqrt 2)) / 2: no cancellation on the side that matters.
__global__ __launch_bounds__(256) void tn_moments_kernel(const float* __restrict__ qa, const float* __restrict__ qb, const float* __restrict__ low, int R,
                                                         double high, double high4, float eps, float* __restrict__ mean, float* __restrict__ sd,
                                                         double* __restrict__ m4) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const double mu = (double)expf(qa[r]), sg = (double)(expf(qb[r]) + eps), a = (double)low[r];
    constexpr double RSQRT2 = 0.70710678118654752440, RSQRT2PI = 0.39894228040143267794;
    const double za = (a - mu) / sg;
    const double pa = RSQRT2PI * exp(-0.5 * za * za);
    if (mean != nullptr || sd != nullptr) {
        const bool open = !(high < 1e30);
        const double zb = open ? 0.0 : (high - mu) / sg;
        const double pb = open ? 0.0 : RSQRT2PI * exp(-0.5 * zb * zb);
        const double zn = 0.5 * (erfc(za * RSQRT2) - (open ? 0.0 : erfc(zb * RSQRT2)));
        const double ratio = (pa - pb) / zn;
        if (mean != nullptr) mean[r] = (float)(mu + sg * ratio);
        if (sd != nullptr) {
            const double bpb = pb > 0.0 ? zb * pb : 0.0;
            sd[r] = (float)sqrt(sg * sg * (1.0 + (za * pa - bpb) / zn - ratio * ratio));
        }
    }
    if (m4 != nullptr) {
        const bool open = !(high4 < 1e30);
        double bterm = 0.0, cb = 0.0;
        if (!open) {
            const double b = high4, zb = (b - mu) / sg;
            bterm = (b * b * b + b * b * mu + b * mu * mu + sg * sg * (3.0 * b + 5.0 * mu) + mu * mu * mu) * RSQRT2PI * exp(-0.5 * zb * zb);
            cb = erfc(zb * RSQRT2);
        }
        const double aterm = (a * a * a + a * a * mu + a * mu * mu + sg * sg * (3.0 * a + 5.0 * mu) + mu * mu * mu) * pa;
        const double den = 0.5 * (erfc(za * RSQRT2) - cb);
        m4[r] = mu * mu * mu * mu + 6.0 * mu * mu * sg * sg + 3.0 * sg * sg * sg * sg - sg * (bterm - aterm) / den;
    }
}

// Output step, per observation (reference VariationalMergingModel.prediction_mean_stddev, careless/models/merging/variational.py:80-121): with
// the scale's moments (smean, sstd) of the row and <F^2> = mean^2 + std^2, <F^4> of its reflection, E[I] = smean <F^2> and
// var[I] = <F^4> (smean^2 + sstd^2) - E[I]^2, in fp64 (the subtraction cancels).  HBM-bound: 12 bytes in, 16 out per row + two gathers.
__global__ __launch_bounds__(256) void predict_moments_kernel(const float* __restrict__ smean, const float* __restrict__ sstd, const int* __restrict__ refl_id,
                                                              long long n, const float* __restrict__ fmean, const float* __restrict__ fstd,
                                                              const double* __restrict__ fm4, int R, double* __restrict__ iexp, double* __restrict__ ivar) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = refl_id[i];
        const bool ok = r >= 0 && r < R;
        const double m = ok ? (double)fmean[r] : 0.0, s = ok ? (double)fstd[r] : 0.0, m4 = ok ? fm4[r] : 0.0;
        const double sm = (double)smean[i], ss = (double)sstd[i];
        const double e = sm * (m * m + s * s);
        iexp[i] = e;
        ivar[i] = m4 * (sm * sm + ss * ss) - e * e;
    }
}

// debug / test aid: the noise the kernels would draw for (seed, step)
__global__ void noise_kernel(unsigned long long seed, unsigned step, int S, long long n, long long offset, int kind,
                             float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int s = 0; s < S; ++s)
        out[(size_t)i * S + s] = (kind == 0) ? cl_noise_uniform(seed, step, (uint32_t)s, (uint64_t)(offset + i))
                                             : cl_noise_normal(seed, step, (uint32_t)s, (uint64_t)(offset + i));
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
int cl_launch_tn_forward(const cl_tn_args& a, hipStream_t st) {
    if (a.R <= 0 || a.S <= 0) return -1;
    (void)hipGetLastError();   // drop any stale error of an unrelated earlier runtime call
    const int nr = a.r_end > a.r_begin ? a.r_end - a.r_begin : a.R;
    hipLaunchKernelGGL(tn_forward_kernel, dim3((nr + 255) / 256), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
int cl_launch_tn_backward(const cl_tn_args& a, hipStream_t st) {
    if (a.R <= 0 || a.S <= 0) return -1;
    (void)hipGetLastError();   // drop any stale error of an unrelated earlier runtime call
    const int nr = a.r_end > a.r_begin ? a.r_end - a.r_begin : a.R;
    const int nred = (a.red_partials != nullptr) ? (a.red_P + 31) / 32 : 0;
    hipLaunchKernelGGL(tn_backward_kernel, dim3((nr + 255) / 256 + nred), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
// Deterministic mode of the double-Wilson prior: what dw_forward_kernel's children scatter into their parents with float atomics, pulled
// by the parent instead -- thread p walks its children (CSR list, ascending) and adds -w dlogp(z_child | z_p) / dz_p to dz_f[p][s] in
// list order (recomputed: the term needs both samples, which are final).  Only children whose KL this rank owns count (row split).
__global__ __launch_bounds__(256) void dw_parent_pull_kernel(const cl_tn_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.R) return;
    const int k0 = A.dw_child_seg[p], k1 = A.dw_child_seg[p + 1];
    if (k1 <= k0) return;
    const float wg = A.w_kl * A.kl_grad_mult;
    for (int s = 0; s < A.S; ++s) {
        const float zp = A.z_f[(size_t)p * A.S + s];
        float acc = 0.0f;
        for (int k = k0; k < k1; ++k) {
            const int h = A.dw_child_ids[k];
            if (A.root[h] != 0 || h < A.kl_begin || h >= A.kl_end) continue;
            float dz, dzp, dr;
            (void)cl_dw_log_prob(A.z_f[(size_t)h * A.S + s], zp, true, dw_r_of(A, h), A.centric[h] != 0, A.es[h], &dz, &dzp, &dr);
            acc += -wg * dzp;
        }
        A.dz_f_out[(size_t)p * A.S + s] += acc;
    }
}

int cl_launch_dw_forward(const cl_tn_args& a, hipStream_t st) {
    if (a.R <= 0 || a.S <= 0) return -1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(dw_forward_kernel, dim3((a.R + 255) / 256), dim3(256), 0, st, a);
    if (a.dw_child_seg != nullptr) hipLaunchKernelGGL(dw_parent_pull_kernel, dim3((a.R + 255) / 256), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
int cl_launch_grad_sqnorm(const float* g, int n, const int* seg_off, int nseg, double* seg_sq, double* scalars,
                          const unsigned char* frozen, const int* stop_flag, hipStream_t st) {
    if (n <= 0) return -1;
    int grid = (n + 255) / 256;
    if (grid > 1024) grid = 1024;
    (void)hipGetLastError();   // drop any stale error of an unrelated earlier runtime call
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(grid), dim3(256), 0, st, g, n, seg_off, nseg, seg_sq, scalars, frozen, stop_flag);
    return (int)hipGetLastError();
}
int cl_adam_grid_of(const cl_adam_args& a) {
    if (a.n <= 0) return -1;
    int work = a.n;
    if (a.n_ranges > 0) {
        work = 0;
        for (int k = 0; k < a.n_ranges; ++k) work += a.range_end[k] - a.range_begin[k];
        if (work < 1) work = 1;
    }
    int grid = (work + 1023) / 1024;          // four elements per thread and round
    // without norm_part a workgroup ends in two same-address fp64 atomics (fused gradient norm), ~12 ns each and serialised: few
    // workgroups then for the usual ~1e6 parameters, more when the vector is long enough for streaming to dominate (per-image layers:
    // 4e7 parameters)
    const int cap = (a.n >= (1 << 22) || a.norm_out == nullptr || a.norm_part != nullptr) ? 1024 : 256;
    if (grid > cap) grid = cap;
    return grid;
}
int cl_launch_adam(const cl_adam_args& a, hipStream_t st) {
    const int grid = cl_adam_grid_of(a);
    if (grid < 1) return -1;
    (void)hipGetLastError();   // drop any stale error of an unrelated earlier runtime call
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
int cl_launch_owner_qnorm(const float* g, int R, int r_begin, int r_end, float* out, double* scratch, const int* stop_flag, hipStream_t st) {
    int grid = (2 * (r_end - r_begin) + 1023) / 1024;
    if (grid > 64) grid = 64;
    if (grid < 1) grid = 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(owner_qnorm_kernel, dim3(grid), dim3(256), 0, st, g, R, r_begin, r_end, out, scratch, stop_flag);
    return (int)hipGetLastError();
}
int cl_launch_finalize(double* scalars, float klw, double* history, int step_index, int hist_stride, int* stop_flag,
                       const double* norm_part, int n_norm_part, hipStream_t st) {
    (void)hipGetLastError();   // drop any stale error of an unrelated earlier runtime call
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, st, scalars, klw, history, step_index, hist_stride, stop_flag, norm_part, n_norm_part);
    return (int)hipGetLastError();
}
int cl_launch_tn_moments(const float* a, const float* b, const float* low, int R, double high, double high4, float eps, float* mean, float* sd,
                         double* m4, hipStream_t st) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(tn_moments_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, a, b, low, R, high, high4, eps, mean, sd, m4);
    return (int)hipGetLastError();
}
int cl_launch_predict_moments(const float* smean, const float* sstd, const int* refl_id, long long n, const float* fmean, const float* fstd,
                              const double* fm4, int R, double* iexp, double* ivar, hipStream_t st) {
    long long blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    (void)hipGetLastError();
    hipLaunchKernelGGL(predict_moments_kernel, dim3((unsigned)blocks), dim3(256), 0, st, smean, sstd, refl_id, n, fmean, fstd, fm4, R, iexp, ivar);
    return (int)hipGetLastError();
}
int cl_launch_noise(unsigned long long seed, unsigned step, int S, long long n, long long offset, int kind, float* out,
                    hipStream_t st) {
    if (n <= 0) return -1;
    (void)hipGetLastError();   // drop any stale error of an unrelated earlier runtime call
    hipLaunchKernelGGL(noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, seed, step, S, n, offset, kind, out);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// deterministic mode (include/careless_hip.h: cl_det_args): fixed-order sums of the per-observation stores of elbo_mlp.hip (-DCL_DET=1)
// ---------------------------------------------------------------------------------------------------------
// dz_f[r][s] += sum over the observations of reflection r in a FIXED order: sixteen lanes share a reflection -- lane l takes the rows
// l, l + 16, ... of its (row-ordered) list, the sixteen lane sums combine in a fixed butterfly.  (Round 3 walked the list with one
// thread per (reflection, sample): ~30 dependent random 4-byte gathers per thread, 0.39 ms per step at 10 M observations -- a fifth of
// the default scaler's step; the order of the sum is as fixed this way, and the gathers of a reflection are in flight together.)
__global__ __launch_bounds__(256) void det_refl_kernel(const cl_det_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const int sub = threadIdx.x & 15;
    const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool on = r < A.R;
    const int k0 = on ? A.seg_refl[r] : 0, k1 = on ? A.seg_refl[r + 1] : 0;
    // the sixteen lanes of a reflection = (row slot, sample): the S' = 1, 2, 4, 8 or 16 samples of a row are consecutive lanes (a row's
    // record is S consecutive floats: one request), 16 / S' rows are in flight per pass; more than 16 samples go in chunks of 16
    const int Sp = A.S <= 1 ? 1 : (A.S <= 2 ? 2 : (A.S <= 4 ? 4 : (A.S <= 8 ? 8 : 16)));
    const int ss = sub & (Sp - 1), slot = sub / Sp, nslot = 16 / Sp;
    for (int s0 = 0; s0 < A.S; s0 += 16) {
        const int s = s0 + ss;
        float acc = 0.0f;
        if (s < A.S) {
            if (A.perm_refl != nullptr) {
                for (int k = k0 + slot; k < k1; k += nslot) acc += A.dzf_obs[(size_t)A.perm_refl[k] * A.S + s];
            } else {            // the records were stored in reflection order (cl_mlp_args.det_slot): a contiguous read
                for (int k = k0 + slot; k < k1; k += nslot) acc += A.dzf_obs[(size_t)k * A.S + s];
            }
        }
        for (int off = 8; off >= Sp; off >>= 1) acc += __shfl_xor(acc, off);      // over the row slots, fixed order
        if (on && slot == 0 && s < A.S) A.dz_f[(size_t)r * A.S + s] += acc;
    }
}

// d_img[m - 1] += sum over the observations of image m (m >= 1), one wave per image: lane l takes the rows l, l + 64, ... of the
// image's range in order, the 64 lane sums combine in a fixed butterfly
__global__ __launch_bounds__(256) void det_img_kernel(const cl_det_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const int m = 1 + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= A.n_images) return;
    float acc = 0.0f;
    for (int k = A.seg_img[m] + lane; k < A.seg_img[m + 1]; k += 64) acc += A.dimg_obs[A.perm_img[k]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) A.d_img[m - 1] += acc;
}

// scalars[NLL] += the workgroups' NLL in index order (one wave; lane l sums parts l, l + 64, ..., then the fixed butterfly)
__global__ __launch_bounds__(64) void det_nll_kernel(const cl_det_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    double acc = 0.0;
    for (int k = threadIdx.x; k < A.nparts; k += 64) acc += A.nll_part[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) A.scalars[CL_SC_NLL] += acc;
}

// d_ev11[c] += the waves' shares of dL/d raw (Evans-2011 error model) in slot order: one wave, lane l sums slots l, l + 64, ..., then the fixed butterfly
__global__ __launch_bounds__(64) void det_ev11_kernel(const cl_det_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    for (int k = threadIdx.x; k < A.n_ev11; k += 64) { a0 += A.ev11_part[3 * k]; a1 += A.ev11_part[3 * k + 1]; a2 += A.ev11_part[3 * k + 2]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); }
    if (threadIdx.x == 0) { A.d_ev11[0] += a0; A.d_ev11[1] += a1; A.d_ev11[2] += a2; }
}

int cl_launch_det_reduce(const cl_det_args& a, hipStream_t st) {
    (void)hipGetLastError();
    if (a.ev11_part != nullptr) hipLaunchKernelGGL(det_ev11_kernel, dim3(1), dim3(64), 0, st, a);
    hipLaunchKernelGGL(det_refl_kernel, dim3((unsigned)((a.R + 15) / 16)), dim3(256), 0, st, a);
    if (a.d_img != nullptr && a.n_images > 1) hipLaunchKernelGGL(det_img_kernel, dim3((a.n_images - 1 + 3) / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(det_nll_kernel, dim3(1), dim3(64), 0, st, a);
    return (int)hipGetLastError();
}
