// Internal launch interface between the C-ABI layer (cl_api.hip) and the kernels.  The argument structs are the
// public ones of include/careless_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/careless_hip.h"

int cl_launch_mlp(const cl_mlp_args& a, int mode, int grid, hipStream_t st);
int cl_launch_mlp_imgl(const cl_mlp_args& a, int mode, int grid, hipStream_t st);   // elbo_mlp.hip compiled with -DCL_IMGL=1
int cl_launch_mlp_packed(const cl_mlp_args& a, int mode, int grid, hipStream_t st); // elbo_mlp.hip compiled with -DCL_IMGL=2
int cl_launch_mlp_chain(const cl_mlp_args& a, int mode, int grid, hipStream_t st);  // elbo_mlp.hip compiled with -DCL_CHAIN=1
int cl_launch_mlp_det(const cl_mlp_args& a, int mode, int grid, hipStream_t st);    // elbo_mlp.hip compiled with -DCL_DET=1 (no atomics)
int cl_launch_mlp_packed_det(const cl_mlp_args& a, int mode, int grid, hipStream_t st);   // ... with -DCL_IMGL=2 -DCL_DET=1 (single-pass Laue, no atomics)
int cl_launch_mlp_chain_det(const cl_mlp_args& a, int mode, int grid, hipStream_t st);    // ... with -DCL_CHAIN=1 -DCL_DET=1 (a chain's last block, no atomics)
int cl_launch_det_reduce(const cl_det_args& a, hipStream_t st);                     // elbo_elem.hip: fixed-order sums of the deterministic mode
int cl_narrow_supports(const cl_mlp_args& a);                                       // elbo_narrow.hip: width <= 15, metadata <= 15, plain layout
int cl_launch_narrow(const cl_mlp_args& a, int grid, hipStream_t st);               // ... the full ELBO step on that kernel
int cl_lane_supports(const cl_mlp_args& a);                                         // elbo_lane.hip: lane = observation; 20 layers, width <= 10, metadata <= 31 columns
int cl_lane_kernel_name(const cl_mlp_args& a, char* out, size_t n);
int cl_narrow_kernel_name(const cl_mlp_args& a, char* out, size_t n);
int cl_mlp_kernel_name_of(const cl_mlp_args& a, int mode, char* out, size_t n);     // elbo_mlp.hip: the routing of cl_launch_mlp, as a name
int cl_lane_imgl_supports(const cl_mlp_args& a);                                    // ... with one or two per-image layers on top (round 5)
int cl_launch_lane_imgl(const cl_mlp_args& a, int grid, hipStream_t st);
int cl_lane_imgl_kernel_name(const cl_mlp_args& a, char* out, size_t n);
int cl_lane_block_supports(const cl_mlp_args& a, int mode);                          // ... a head-less layer block's forward / backward launch (round 6)
int cl_launch_lane_block(const cl_mlp_args& a, int mode, int grid, hipStream_t st);
int cl_launch_lane(const cl_mlp_args& a, int grid, hipStream_t st);                 // ... the full ELBO step on that kernel
int cl_launch_reduce_partials(const float* partials, int nparts, int P, float* out, const int* stop_flag, hipStream_t st);
int cl_launch_tn_forward(const cl_tn_args& a, hipStream_t st);
int cl_launch_tn_backward(const cl_tn_args& a, hipStream_t st);
int cl_launch_dw_forward(const cl_tn_args& a, hipStream_t st);
int cl_launch_grad_sqnorm(const float* g, int n, const int* seg_off, int nseg, double* seg_sq, double* scalars,
                          const unsigned char* frozen, const int* stop_flag, hipStream_t st);
int cl_launch_adam(const cl_adam_args& a, hipStream_t st);
int cl_launch_owner_qnorm(const float* g, int R, int r_begin, int r_end, float* out, double* scratch, const int* stop_flag, hipStream_t st);
int cl_launch_finalize(double* scalars, float klw, double* history, int step_index, int hist_stride, int* stop_flag,
                       const double* norm_part, int n_norm_part, hipStream_t st);
int cl_adam_grid_of(const cl_adam_args& a);
int cl_launch_predict_moments(const float* smean, const float* sstd, const int* refl_id, long long n, const float* fmean, const float* fstd,
                              const double* fm4, int R, double* iexp, double* ivar, hipStream_t st);      // elbo_elem.hip: output step, per observation
int cl_launch_tn_moments(const float* a, const float* b, const float* low, int R, double high, double high4, float eps, float* mean, float* sd,
                         double* m4, hipStream_t st);                                // elbo_elem.hip: moments of q for the output step
int cl_launch_noise(unsigned long long seed, unsigned step, int S, long long n, long long offset, int kind, float* out,
                    hipStream_t st);
int cl_launch_laue_predict(const cl_laue_args& a, hipStream_t st);
int cl_launch_laue_likelihood(const cl_laue_args& a, hipStream_t st);
int cl_launch_laue_backward(const cl_laue_args& a, hipStream_t st);
int cl_launch_slot_rows(const cl_laue_args& a, hipStream_t st);
int cl_launch_frozen_rows(const cl_frozen_args& a, hipStream_t st);      // elbo_frozen.hip (round 6)

// The kernel arguments, re-read from the kernarg segment behind an opaque pointer.  hipcc loads every field of a by-value argument
// struct at kernel entry and keeps it in SGPRs for the whole kernel (more than the ~100 there are: it then parks them in VGPR lanes
// and pays v_readlane / v_writelane in the per-tile code); fields that only one phase of a tile uses are loaded there instead, by
// scalar loads that cannot be hoisted.  Only valid in kernels whose single parameter is a `cl_mlp_args` by value.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) cl_mlp_args* cl_args_p;
__device__ __forceinline__ cl_args_p kernargs_again() {
    cl_args_p p = (cl_args_p)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// Cross-lane sums on the DPP network: no LDS round trip, one vector instruction per step (a __shfl_xor step is a ds_bpermute_b32
// plus its address arithmetic and an LDS latency in the middle of a dependent chain).
#define CL_DPP_ADD(x, ctrl, rmask) \
    ((x) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (rmask), 0xF, false)))
// sum over each group of four consecutive lanes, left in all four
__device__ __forceinline__ float cl_quad_sum(float v) {
    v = CL_DPP_ADD(v, 0xB1, 0xF);   // quad_perm [1,0,3,2]
    v = CL_DPP_ADD(v, 0x4E, 0xF);   // quad_perm [2,3,0,1]
    return v;
}
// sum over each aligned group of S consecutive lanes (S = 1, 2, 4, 8 or 16: inside a 16-lane row), left in all lanes of the group
__device__ __forceinline__ float cl_group_sum(float v, int S) {
    if (S >= 2) v = CL_DPP_ADD(v, 0xB1, 0xF);
    if (S >= 4) v = CL_DPP_ADD(v, 0x4E, 0xF);
    if (S >= 8) v = CL_DPP_ADD(v, 0x141, 0xF);    // row_half_mirror
    if (S >= 16) v = CL_DPP_ADD(v, 0x140, 0xF);   // row_mirror
    return v;
}
// sum over the 64 lanes of a wave, returned as a wave-uniform value: a butterfly inside the 16-lane rows, then the rows through row_bcast
__device__ __forceinline__ float cl_wave_sum(float v) {
    v = cl_quad_sum(v);
    v = CL_DPP_ADD(v, 0x141, 0xF);  // row_half_mirror
    v = CL_DPP_ADD(v, 0x140, 0xF);  // row_mirror: every lane now holds its row's sum
    v = CL_DPP_ADD(v, 0x142, 0xA);  // row_bcast15 into rows 1 and 3
    v = CL_DPP_ADD(v, 0x143, 0xC);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// out[i] += sum over the `nparts` per-workgroup partials of element i, in index order (deterministic): the work of ONE 256-thread
// workgroup -- 32 consecutive elements x 8 chunks of the partial list; a thread sums its chunk (coalesced 128-B rows, eight rows in
// flight: a one-at-a-time loop is a chain of dependent HBM / MALL latencies), the 8 chunk sums are combined through LDS in chunk order.
// Shared by reduce_partials_kernel (elbo_mlp.hip) and the extra workgroups of tn_backward_kernel (elbo_elem.hip).
__device__ __forceinline__ void cl_reduce_partials_block(const float* __restrict__ partials, int nparts, int P, float* __restrict__ out, int block) {
    __shared__ float sh[8][33];
    const int e = threadIdx.x & 31, c = threadIdx.x >> 5;
    const int i = block * 32 + e;
    const int per = (nparts + 7) / 8;
    float s = 0.0f;
    if (i < P) {
        const int g1 = min(nparts, (c + 1) * per);
        int g = c * per;
        for (; g + 8 <= g1; g += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partials[(size_t)(g + k) * P + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; g < g1; ++g) s += partials[(size_t)g * P + i];
    }
    sh[c][e] = s;
    __syncthreads();
    if (c == 0 && i < P) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sh[k][e];
        out[i] += t;
    }
}

// Image-scale gradients of a wave whose observations belong to MORE than one image (the rows of an image end inside the wave:
// stills with a few dozen reflections per image, or a reflection-owner shard that holds an eighth of every image's rows).  Image ids
// are sorted, so a wave holds a few images: one wave reduction and ONE atomic per image, image by image (up to four; a wave with more
// -- unsorted input -- finishes lane by lane).  `take`: this lane carries a term of a trainable image (img > 0); d_img[img - 1] += v.
__device__ __forceinline__ void cl_image_grad_segments(float* __restrict__ d_img, int img, float v, bool take, int lane) {
    unsigned long long m = __ballot(take);
    for (int it = 0; it < 4 && m != 0ull; ++it) {                      // wave-uniform
        const int first = __builtin_ctzll(m);
        const int im = __builtin_amdgcn_readlane(img, first);
        const bool mine = take && img == im;
        const float s = cl_wave_sum(mine ? v : 0.0f);
        if (lane == 0) atomicAdd(d_img + (im - 1), s);
        take = take && !mine;
        m = __ballot(take);
    }
    if (take) atomicAdd(d_img + (img - 1), v);
}
#else
typedef const cl_mlp_args* cl_args_p;                                        // (host pass over the kernel bodies: never executed)
__host__ __device__ inline cl_args_p kernargs_again() { return nullptr; }
__host__ __device__ inline float cl_quad_sum(float v) { return v; }
__host__ __device__ inline float cl_wave_sum(float v) { return v; }
__host__ __device__ inline float cl_group_sum(float v, int) { return v; }
__host__ __device__ inline void cl_image_grad_segments(float*, int, float, bool, int) {}
__host__ __device__ inline void cl_reduce_partials_block(const float*, int, int, float*, int) {}
#endif
