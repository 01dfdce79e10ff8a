// extern "C" boundary of libcareless_hip.so (declared in include/careless_hip.h): argument checks + launches.
#include <hip/hip_runtime.h>
#include "cl_kernels.h"

extern "C" {

const char* cl_version(void) { return "careless_hip 0.1.0 (gfx950)"; }

void cl_abi_sizes(size_t out[5]) {
    out[0] = sizeof(cl_tn_args);
    out[1] = sizeof(cl_mlp_args);
    out[2] = sizeof(cl_adam_args);
    out[3] = sizeof(cl_laue_args);
    out[4] = sizeof(cl_det_args);
}

int cl_det_reduce(const cl_det_args* a, void* stream) {
    if (a == nullptr || a->dzf_obs == nullptr || a->seg_refl == nullptr || a->dz_f == nullptr || a->R < 1 || a->S < 1 ||
        a->nll_part == nullptr || a->nparts < 1 || a->scalars == nullptr)
        return -1;
    if (a->d_img != nullptr && (a->dimg_obs == nullptr || a->perm_img == nullptr || a->seg_img == nullptr || a->n_images < 1)) return -1;
    if (a->ev11_part != nullptr && (a->d_ev11 == nullptr || a->n_ev11 < 1)) return -1;
    return cl_launch_det_reduce(*a, (hipStream_t)stream);
}

int cl_laue_predict(const cl_laue_args* a, void* stream) { return a ? cl_launch_laue_predict(*a, (hipStream_t)stream) : -1; }
int cl_laue_likelihood(const cl_laue_args* a, void* stream) { return a ? cl_launch_laue_likelihood(*a, (hipStream_t)stream) : -1; }
int cl_laue_backward(const cl_laue_args* a, void* stream) { return a ? cl_launch_laue_backward(*a, (hipStream_t)stream) : -1; }
int cl_slot_rows(const cl_laue_args* a, void* stream) { return a ? cl_launch_slot_rows(*a, (hipStream_t)stream) : -1; }
int cl_frozen_rows(const cl_frozen_args* a, void* stream) { return a ? cl_launch_frozen_rows(*a, (hipStream_t)stream) : -1; }
size_t cl_frozen_args_size(void) { return sizeof(cl_frozen_args); }

int cl_mlp_default_grid(void) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return cus;
}

int cl_mlp_max_layers(int w) {
    if (w < 1 || w > 64) return 0;
    return w <= 16 ? CL_MLP_LMAX_W16 : (w <= 32 ? CL_MLP_LMAX_W32 : CL_MLP_LMAX_W64);      // (width 16: the 16-wide instance with explicit biases, round 5)
}

int cl_mlp_max_layers_imgl(int w) {
    if (w < 1 || w > 64) return 0;
    return w <= 15 ? CL_MLP_LMAX_W16_IMGL : (w <= 32 ? CL_MLP_LMAX_W32 : CL_MLP_LMAX_W64);
}

int cl_mlp_meta_rows(int d) { return d < 1 ? 0 : ((d + 3) & ~3); }

size_t cl_mlp_param_count(int d, int w, int L) {
    if (d < 1 || w < 1 || L < 1) return 0;
    return (size_t)w * d + w + (size_t)(L - 1) * ((size_t)w * w + w) + 2 * (size_t)w + 2;
}

static int check_mlp(const cl_mlp_args* a) {
    if (a == nullptr) return -1;
    if (a->meta_t == nullptr || a->mlp == nullptr) return -1;
    if (a->n_obs <= 0 || a->n_pad < a->n_obs || a->n_pad % CL_MLP_TILE != 0) return -1;
    return 0;
}

int cl_elbo_mono_fwd_bwd(const cl_mlp_args* a, int grid, void* stream) {
    if (int e = check_mlp(a)) return e;
    if (a->refl_id == nullptr || a->iobs == nullptr || a->sig == nullptr || a->z_f == nullptr || a->dz_f == nullptr ||
        a->partials == nullptr || a->scalars == nullptr || a->S < 1 || a->R < 1)
        return -1;
    if (a->use_img && (a->image_id == nullptr || a->img == nullptr || a->d_img == nullptr)) return -1;
    return cl_launch_mlp(*a, 0, grid, (hipStream_t)stream);
}

int cl_mlp_forward(const cl_mlp_args* a, int grid, void* stream) {
    if (int e = check_mlp(a)) return e;
    if (a->act_out == nullptr && (a->loc_out == nullptr || a->sig_out == nullptr)) return -1;
    return cl_launch_mlp(*a, 1, grid, (hipStream_t)stream);
}

int cl_mlp_backward_ext(const cl_mlp_args* a, int grid, void* stream) {
    if (int e = check_mlp(a)) return e;
    if ((a->dO_ext == nullptr && a->dH_ext == nullptr) || a->partials == nullptr) return -1;
    return cl_launch_mlp(*a, 2, grid, (hipStream_t)stream);
}

int cl_mlp_kernel_name(const cl_mlp_args* a, int mode, char* out, size_t n) {
    if (a == nullptr || out == nullptr || n == 0 || mode < 0 || mode > 2) return -1;
    return cl_mlp_kernel_name_of(*a, mode, out, n);
}

int cl_reduce_partials(const float* partials, int nparts, int P, float* grad_mlp, const int* stop_flag, void* stream) {
    if (partials == nullptr || grad_mlp == nullptr || nparts < 1 || P < 1) return -1;
    return cl_launch_reduce_partials(partials, nparts, P, grad_mlp, stop_flag, (hipStream_t)stream);
}

static int check_tn(const cl_tn_args* a) {
    if (a == nullptr || a->q_loc_raw == nullptr || a->q_scale_raw == nullptr || a->low == nullptr ||
        a->centric == nullptr || a->es == nullptr || a->R < 1 || a->S < 1)
        return -1;
    if (a->prior_kind == CL_PRIOR_DOUBLE_WILSON_ && (a->parent_ids == nullptr || a->root == nullptr)) return -1;
    if (a->prior_kind == CL_PRIOR_DOUBLE_WILSON_ && a->dw_r == nullptr && (a->dw_r_raw == nullptr || a->asu_ids == nullptr)) return -1;
    if (a->prior_kind != CL_PRIOR_DOUBLE_WILSON_ && a->prior_kind != CL_PRIOR_WILSON_) return -1;
    if (a->r_end > a->r_begin && (a->r_begin < 0 || a->r_end > a->R)) return -1;
    // an owned reflection range and the double-Wilson prior do not go together: a child's parent may belong to another rank
    if (a->r_end > a->r_begin && a->prior_kind == CL_PRIOR_DOUBLE_WILSON_) return -2;
    return 0;
}

int cl_tn_forward(const cl_tn_args* a, void* stream) {
    if (int e = check_tn(a)) return e;
    if (a->z_f == nullptr || a->scalars == nullptr) return -1;
    if (a->zero_ptr != nullptr && (a->zero_n < 1 || (reinterpret_cast<uintptr_t>(a->zero_ptr) & 15) != 0)) return -1;
    // the KL must not be added into a block this launch is clearing: the parts go to kl_part then
    if (a->zero_ptr != nullptr && a->kl_part == nullptr) return -1;
    return cl_launch_tn_forward(*a, (hipStream_t)stream);
}

int cl_tn_backward(const cl_tn_args* a, void* stream) {
    if (int e = check_tn(a)) return e;
    if (a->dz_f == nullptr || a->d_loc_raw == nullptr || a->d_scale_raw == nullptr) return -1;
    if (a->red_partials != nullptr && (a->red_out == nullptr || a->red_nparts < 1 || a->red_P < 1)) return -1;
    return cl_launch_tn_backward(*a, (hipStream_t)stream);
}

int cl_dw_prior_forward(const cl_tn_args* a, void* stream) {
    if (int e = check_tn(a)) return e;
    if (a->prior_kind != CL_PRIOR_DOUBLE_WILSON_ || a->z_f == nullptr || a->dz_f_out == nullptr || a->scalars == nullptr) return -1;
    if (a->dw_r_raw != nullptr && (a->d_dw_r_raw == nullptr || a->n_asu < 1)) return -1;
    if ((a->dw_child_seg != nullptr) != (a->dw_child_ids != nullptr)) return -1;
    if (a->dw_child_seg != nullptr && a->dw_r_raw != nullptr) return -2;          // deterministic mode: fixed r only
    return cl_launch_dw_forward(*a, (hipStream_t)stream);
}

int cl_grad_sqnorm(const float* g, int n, const int* seg_off, int nseg, double* seg_sq, double* scalars,
                   const unsigned char* frozen, const int* stop_flag, void* stream) {
    if (g == nullptr || scalars == nullptr || n < 1) return -1;
    if ((seg_sq != nullptr || frozen != nullptr) && (seg_off == nullptr || nseg < 1)) return -1;
    return cl_launch_grad_sqnorm(g, n, seg_off, nseg, seg_sq, scalars, frozen, stop_flag, (hipStream_t)stream);
}

int cl_adam_step(const cl_adam_args* a, void* stream) {
    if (a == nullptr || a->p == nullptr || a->g == nullptr || a->m == nullptr || a->v == nullptr || a->n < 1) return -1;
    if ((a->clipnorm > 0.0f || a->frozen != nullptr) && (a->seg_off == nullptr || a->nseg < 1)) return -1;
    if (a->clipnorm > 0.0f && a->seg_sq == nullptr) return -1;
    if (a->global_clipnorm > 0.0f && a->scalars == nullptr) return -1;
    if (a->n_ranges < 0 || a->n_ranges > 3 || a->norm_skip_ranges < 0 || a->norm_skip_ranges > a->n_ranges) return -1;
    for (int k = 0; k < a->n_ranges; ++k)
        if (a->range_begin[k] < 0 || a->range_end[k] > a->n || a->range_end[k] < a->range_begin[k]) return -1;
    return cl_launch_adam(*a, (hipStream_t)stream);
}

int cl_owner_qnorm(const float* g, int R, int r_begin, int r_end, float* out, double* scratch, const int* stop_flag, void* stream) {
    if (g == nullptr || out == nullptr || scratch == nullptr || R < 1 || r_begin < 0 || r_end > R || r_end <= r_begin) return -1;
    return cl_launch_owner_qnorm(g, R, r_begin, r_end, out, scratch, stop_flag, (hipStream_t)stream);
}

int cl_step_finalize(double* scalars, float kl_weight_or_one, double* history, int step_index, int* stop_flag,
                     const double* norm_part, int n_norm_part, void* stream) {
    if (scalars == nullptr || history == nullptr || step_index < 0) return -1;
    if (norm_part != nullptr && n_norm_part < 1) return -1;
    return cl_launch_finalize(scalars, kl_weight_or_one, history, step_index, CL_HIST_STRIDE, stop_flag, norm_part, n_norm_part, (hipStream_t)stream);
}

int cl_adam_grid(const cl_adam_args* a) { return a == nullptr ? -1 : cl_adam_grid_of(*a); }

int cl_tn_moments(const float* q_loc_raw, const float* q_scale_raw, const float* low, int R, double high_moments, double high_m4, float eps,
                  float* mean, float* std, double* m4, void* stream) {
    if (q_loc_raw == nullptr || q_scale_raw == nullptr || low == nullptr || R < 1) return -1;
    if (mean == nullptr && std == nullptr && m4 == nullptr) return -1;
    return cl_launch_tn_moments(q_loc_raw, q_scale_raw, low, R, high_moments, high_m4, eps, mean, std, m4, (hipStream_t)stream);
}

int cl_predict_moments(const float* scale_mean, const float* scale_std, const int* refl_id, long long n, const float* f_mean, const float* f_std,
                       const double* f_m4, int R, double* iexp, double* ivar, void* stream) {
    if (scale_mean == nullptr || scale_std == nullptr || refl_id == nullptr || f_mean == nullptr || f_std == nullptr || f_m4 == nullptr ||
        iexp == nullptr || ivar == nullptr || n < 1 || R < 1)
        return -1;
    return cl_launch_predict_moments(scale_mean, scale_std, refl_id, n, f_mean, f_std, f_m4, R, iexp, ivar, (hipStream_t)stream);
}

int cl_debug_noise(unsigned long long seed, unsigned step, int S, long long n, long long offset, int kind, float* out,
                   void* stream) {
    if (out == nullptr || S < 1 || n < 1) return -1;
    return cl_launch_noise(seed, step, S, n, offset, kind, out, (hipStream_t)stream);
}

}  // extern "C"
