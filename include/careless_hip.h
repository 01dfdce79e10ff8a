/* careless_hip.h -- C-ABI of the MI355X (gfx950) ELBO engine: libcareless_hip.so
 *
 * This is the drop-in boundary for the per-step Monte-Carlo ELBO of rs-station/careless.  The reference has no FFI on
 * this path: the work is a chain of TensorFlow / TFP ops issued from Python.  Each entry point below replaces the
 * op chain of the cited reference lines (paths relative to the reference checkout) and is what a maintainer would bind
 * from Python with ctypes.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc'ed / torch.Tensor.data_ptr()), row-major, contiguous, owned by the caller and never
 *     retained after the call returns -- EXCEPT in the cl_host_* group at the end of this file (the formatting step: HOST pointers, host
 *     threads, no stream, no device);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); every call only enqueues;
 *   - return value: 0 = ok, < 0 = invalid argument (-1 shape, -2 unsupported scaler geometry, -3 LDS budget, -4 shard >= 4 GiB),
 *     > 0 = hipError_t from the launch;  nothing throws, nothing allocates persistent device memory;
 *   - no global mutable state besides the loaded code object: calls are re-entrant across streams;
 *   - a NULL noise pointer (u_f / eta) selects the in-kernel counter-based generator keyed by
 *     (seed, step, sample, global element index), so results are independent of the number of GPUs.
 *
 * Device data layout (what the host mirror `careless_amd` prepares once per data set)
 *   z_f, dz_f, u_f       [R][S]   reflection-major, MC sample fastest
 *   eta, ipred_out       [N][S]
 *   meta_t               [cl_mlp_meta_rows(d)][n_pad] feature-major metadata (rows >= d and observations >= N are zero),
 *                        n_pad = N rounded up to CL_MLP_TILE
 *   scaler parameters    "W^T layout": per Dense layer the kernel transposed (rows = output units) then its bias:
 *                        Wt0[w][d] b0[w] | Wtl[w][w] bl[w] (l = 1..L-1) | Wto[2][w] bo[2]
 *   flat parameters      [ q_loc_raw (R) | q_scale_raw (R) | scaler (P) | image scales (M-1) ]
 */
#ifndef CARELESS_HIP_H
#define CARELESS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CL_MLP_TILE 128   /* observations per workgroup tile */
/* deepest scaler the fused kernel is instantiated for, by hidden width (cl_mlp_max_layers) */
#ifndef CL_MLP_LMAX_W16
#define CL_MLP_LMAX_W16 20
#endif /* width <= 16 (the careless CLI default is 20 layers x width 10); width <= 15: padded feature 15 carries the bias gradient */
#ifndef CL_MLP_LMAX_W16_IMGL
#define CL_MLP_LMAX_W16_IMGL 24
#endif /* width <= 15 with per-image layers: Dense + image layers (the default 20 + --image-layers <= 4) */
#define CL_MLP_LMAX_W32 10 /* width 17 .. 32 */
#define CL_MLP_LMAX_W64 5  /* width <= 64 */
#define CL_HIST_STRIDE 8  /* doubles per history record: loss, F KLDiv, NLL, Grad Norm, skipped, 3 spare */

/* indices into the double-precision scalar block of one step */
enum { CL_SC_NLL = 0, CL_SC_KL = 1, CL_SC_GNORM2 = 2, CL_SC_GNORM2_SANE = 3, CL_SC_COUNT = 4 };

/* --- surrogate posterior + prior ------------------------------------------------------------------------------
 * replaces: TruncatedNormal.from_loc_and_scale/.sample/.log_prob  (careless/models/merging/surrogate_posteriors.py:104-131, 50-53, 20-21)
 *           WilsonPrior.log_prob                                  (careless/models/priors/wilson.py:50-57)
 *           VariationalMergingModel.add_kl_div                    (careless/models/merging/variational.py:123-139)      */
typedef struct cl_tn_args {
    const float* q_loc_raw;     /* [R] a = log(loc)                        */
    const float* q_scale_raw;   /* [R] b = log(scale - eps)                */
    const float* low;           /* [R] lower truncation (io/manager.py:434) */
    const unsigned char* centric; /* [R] 0/1                               */
    const float* es;            /* [R] multiplicity * Sigma                */
    int R, S;
    float high, eps;
    float w_kl;                 /* weight of every KL element in the reported KL: 1/S (sum) or 1/(S R) (mean)   */
    float kl_grad_mult;         /* extra weight of the KL in the loss: 1, or --kl-weight                        */
    int kl_begin, kl_end;       /* reflection range whose KL this rank owns (data-parallel: counted once)       */
    int r_begin, r_end;         /* cl_tn_forward / cl_tn_backward work on reflections [r_begin, r_end) only -- a rank that OWNS a
                                 * reflection range (all observations of those reflections are its own: no other rank reads their
                                 * samples or adds to their gradients); r_end <= r_begin means all of [0, R)                     */
    const float* u_f;           /* [R][S] injected uniforms or NULL                                             */
    unsigned long long seed; unsigned step;
    float* z_f;                 /* [R][S] out (forward)                                                          */
    const float* dz_f;          /* [R][S] in  (backward): dL/dz_f of the data term                               */
    float* d_loc_raw;           /* [R] += (backward)                                                             */
    float* d_scale_raw;         /* [R] += (backward)                                                             */
    double* scalars;            /* [CL_SC_COUNT]: forward adds the weighted KL into scalars[CL_SC_KL]            */
    double* kl_part;            /* optional [ceil(R / 256)] workspace: cl_tn_forward then STORES every workgroup's KL sum here instead of
                                 * adding it to scalars[CL_SC_KL] (a 312 k-reflection launch is resident all at once: its 1 200 same-address
                                 * fp64 atomics queue up at the end, 9 of the kernel's 19 us), and the cl_tn_backward of the same step adds
                                 * them up, in index order, into scalars[CL_SC_KL]                                                       */
    double* kl_part_dw;         /* optional [ceil(R / 256)]: the same for cl_dw_prior_forward (also summed by cl_tn_backward)               */
    /* Fewer launches per step (each small kernel costs ~4.5 us of launch + drain, a quarter of the step of a 250 k-observation data set):
     * cl_tn_forward can clear the step's accumulators on its way, cl_tn_backward can carry the scaler-gradient reduction.            */
    float* zero_ptr;            /* optional: cl_tn_forward zeroes zero_ptr[0 .. zero_n) (the caller's flat gradient + scalar block) ...     */
    long long zero_n;
    float* zero_dzf;            /* ... and, non-NULL, the rows [R][S] of this dz_f buffer that belong to the reflections it works on        */
    const float* red_partials;  /* optional: cl_tn_backward also runs cl_reduce_partials(red_partials, red_nparts, red_P, red_out) in extra   */
    int red_nparts, red_P;      /*           workgroups of the same launch                                                                 */
    float* red_out;
    const int* stop_flag;       /* optional device int: non-zero => skip (numerical failure in an earlier step)  */
    /* double-Wilson prior (careless/models/priors/wilson.py:82-175); all NULL / 0 for the plain Wilson prior        */
    int prior_kind;             /* CL_PRIOR_WILSON_ | CL_PRIOR_DOUBLE_WILSON_                                       */
    const int* parent_ids;      /* [R] reflection id of the parent in the parent ASU, -1 = absent (`reflids`)       */
    const unsigned char* root;  /* [R] 1 = reflection of a root ASU (plain Wilson prior)                            */
    const float* dw_r;          /* [R] correlation r of the reflection's ASU with its parent (`r[asu_ids]`)         */
    float* dz_f_out;            /* [R][S] += -w dlogp/dz_parent (cl_dw_prior_forward); the same buffer as dz_f      */
    /* --optimize-double-wilson-r (wilson.py:105-110): r = sigmoid(raw) per ASU is trainable; NULL = fixed r (dw_r)      */
    const float* dw_r_raw;      /* [n_asu] pre-sigmoid r                                                            */
    const int* asu_ids;         /* [R] ASU of every reflection                                                      */
    float* d_dw_r_raw;          /* [n_asu] += dL/d raw (cl_dw_prior_forward)                                        */
    int n_asu;
    /* Deterministic mode (round 4): with the children of every reflection given as a CSR list -- child_ids[child_seg[p] ..
     * child_seg[p+1]) are the reflections whose parent is p, ascending -- cl_dw_prior_forward issues NO atomic: a second launch of the
     * call lets every parent add up its children's -w dlogp/dz_parent itself, in list order (the term is recomputed there).
     * Fixed r only (the trainable r's gradient is a per-wave atomic).  NULL: children scatter with float atomics.                      */
    const int* dw_child_seg;    /* [R + 1] */
    const int* dw_child_ids;    /* [number of reflections with a parent] */
} cl_tn_args;
enum { CL_PRIOR_WILSON_ = 0, CL_PRIOR_DOUBLE_WILSON_ = 1 };

int cl_tn_forward(const cl_tn_args* args, void* stream);
int cl_tn_backward(const cl_tn_args* args, void* stream);
/* double-Wilson only, after cl_tn_forward: adds -w log p(z_h | z_parent) of every non-root reflection to the KL and scatters
 * its derivative w.r.t. the parent's sample into dz_f_out (replaces DoubleWilsonPrior.log_prob, wilson.py:146-175)          */
int cl_dw_prior_forward(const cl_tn_args* args, void* stream);

/* --- scaler + likelihood -----------------------------------------------------------------------------------------
 * replaces: MLPScaler.call / MetadataScaler / NormalLayer        (careless/models/scaling/nn.py:10-120)
 *           ImageScaler.call / HybridImageScaler.call            (careless/models/scaling/image.py:27-63)
 *           VariationalMergingModel.call gather + predict        (careless/models/merging/variational.py:156-167)
 *           NormalLikelihood / StudentTLikelihood log_prob       (careless/models/likelihoods/mono.py:10-37)
 *           tape.gradient of all of the above                    (careless/models/merging/variational.py:197-202) */
typedef struct cl_mlp_args {
    const int* refl_id;         /* [n_obs]                                    */
    const int* image_id;        /* [n_obs]                                    */
    const float* meta_t;        /* [cl_mlp_meta_rows(d)][n_pad]               */
    const float* iobs;          /* [n_obs]                                    */
    const float* sig;           /* [n_obs]                                    */
    int n_obs, n_pad;
    long long obs_offset;       /* global index of the shard's first observation (noise key) */
    const float* mlp;           /* scaler parameters, W^T layout              */
    int d, w, L;
    float leak;
    const float* img;           /* [M-1] trainable image scales (image 0 pinned to 1, image.py:23-25) */
    int use_img;
    const float* z_f;           /* [R][S]                                      */
    int R, S;
    int lik_kind; float dof, lik_const;   /* CL_LIK_*; lik_const = lgamma((nu+1)/2) - lgamma(nu/2) - log(nu pi)/2 */
    int bij_kind; float eps, shift;       /* CL_BIJ_*; sigma = f(raw) + eps; shift = tfb.Shift(std(Iobs)) (nn.py:84-87) */
    float w_ll;                 /* weight of every log-likelihood term: 1/S or 1/(S N_total) */
    const float* eta;           /* [n_obs][S] injected normals or NULL         */
    unsigned long long seed; unsigned step;
    float* dz_f;                /* [R][S] += dL/dz_f                           */
    float* d_img;               /* [M-1]  += dL/d(image scale)                 */
    float* partials;            /* [grid][P] per-workgroup scaler gradient partials (workspace) */
    double* scalars;            /* [CL_SC_COUNT]: adds NLL into scalars[CL_SC_NLL] */
    float* ipred_out;           /* optional [n_obs][S]                         */
    float* loc_out;             /* cl_mlp_forward: [n_obs]                     */
    float* sig_out;             /* cl_mlp_forward: [n_obs]                     */
    const float* dO_ext;        /* cl_mlp_backward_ext: [n_obs][2] dL/d(loc, sigma) */
    const int* stop_flag;
    /* Evans-2011 error model (--refine-uncertainties; careless/models/likelihoods/mono.py:39-73): NULL = off */
    const float* ev11;          /* [3] raw (pre-softplus) Sdfac, Sdadd, SdB                                   */
    float* d_ev11;              /* [3] += dL/d raw                                                             */
    /* NeuralImageScaler (--image-layers K; careless/models/scaling/image.py:66-125): after the L Dense layers, K layers
     * h <- LeakyReLU(W[image] h + b[image]) with one (w x w) matrix and bias per image.  n_imgl = 0 / NULL = off.
     * Layout of imgl / d_imgl: K blocks of [ W: n_images x (w x w), row-major (out, in) | b: n_images x w ].
     * The observation axis must then be PACKED: every 128-observation tile holds rows of ONE image (tile_img), rows of an
     * image are padded to whole tiles with refl_id = -1, sig = 1, metadata 0; n_obs == n_pad.  row_map gives the caller's
     * row of every packed row (-1: padding): eta / ipred_out / loc_out / sig_out / dO_ext and the noise key
     * (obs_offset + row) stay in the caller's row order.                                                          */
    const float* imgl;
    float* d_imgl;              /* += dL/d(imgl)                                                               */
    int n_imgl, n_images;
    const int* tile_img;        /* [n_pad / CL_MLP_TILE] image of every tile (only with n_imgl > 0)            */
    const int* row_map;         /* [n_pad]; non-NULL selects the packed layout also without image layers      */
    /* Single-pass Laue likelihood (careless/models/likelihoods/laue.py:9-47) inside cl_elbo_mono_fwd_bwd: packed layout in
     * which the rows of one harmonic group are consecutive and never straddle a 16-row boundary (a wave's observations), so
     * the group sum of the predictions is a lane reduction in the epilogue.  gmeta[row] = member index | (group size << 8);
     * iobs / sig hold the GROUP's observed intensity on every member row; the group's likelihood is counted on member 0.
     * tile_gmax[tile] = largest group size in the tile.  The padded slots [G, N) are the caller's job (cl_laue_likelihood on
     * one zero slot with w_ll scaled by their number).  NULL = off.                                                   */
    const int* gmeta;
    const int* tile_gmax;
    const int* noise_row;       /* optional [n_pad]: GLOBAL row of every packed row for the in-kernel noise key (shards that are
                                 * not a contiguous row range); NULL = obs_offset + row_map[row]                        */
    /* Scalers deeper than cl_mlp_max_layers(w) run as a CHAIN of layer blocks, each block one launch whose "metadata" is the
     * previous block's output (same feature-major [cl_mlp_meta_rows(w)][n_pad] layout, d = w):
     *   cl_mlp_forward      with act_out : `mlp` holds L Dense layers and NO Dense(2) head; writes the last layer's activations
     *   cl_mlp_backward_ext with dH_ext  : same parameters; gradient w.r.t. those activations comes in (instead of dO_ext)
     *   dX_out (any backward launch)     : also write dL/d(metadata) -- the dH_ext of the block before
     * The scaler-gradient partials of a head-less block have cl_mlp_param_count(d, w, L) - 2 w - 2 entries.               */
    float* act_out;             /* [cl_mlp_meta_rows(w)][n_pad]                                                        */
    const float* dH_ext;        /* [cl_mlp_meta_rows(w)][n_pad]                                                        */
    float* dX_out;              /* [cl_mlp_meta_rows(d)][n_pad]                                                        */
    /* Deterministic mode (no reference counterpart: the reference's CPU path is deterministic, fp32 atomics in arbitrary order are
     * not).  With dzf_obs non-NULL cl_elbo_mono_fwd_bwd (plain layout, width <= 64, no Evans-2011 terms) issues NO atomic: the
     * amplitude gradient of every (observation, sample) is STORED in dzf_obs, the image-scale gradient of every observation in
     * dimg_obs, every workgroup's NLL in nll_part; cl_det_reduce then sums them per reflection / per image / per launch in a fixed
     * order.  Two runs on the same inputs give bit-identical gradients.  Per-image layers (n_imgl > 0; round 6): only where the launch
     * runs the lane kernel's instances (cl_mlp_kernel_name says "elbo_lane_kernel<...> (image layers)": n_imgl <= 3 on 2 .. 20 Dense layers
     * of width <= 10, <= 2 on 19) -- one wave then holds all tiles of an image, its gradient is ONE addition per element onto the cleared d_imgl;
     * every other shape with per-image layers returns -2 in this mode.                                                          */
    float* dzf_obs;             /* [n_obs][S]                                                                          */
    float* dimg_obs;            /* [n_obs]                                                                             */
    double* nll_part;           /* [grid]                                                                              */
    const int* det_slot;        /* optional [n_obs] (round 4): observation i's record goes to dzf_obs[det_slot[i]][S] instead of dzf_obs[i][S] --
                                   the caller's reflection-sorted position, so that cl_det_reduce (perm_refl NULL) reads every reflection's
                                   records contiguously instead of gathering them; the scattered STORES cost the kernel nothing it waits for */
    float* dZ0_out;             /* optional [cl_mlp_meta_rows(w)][n_pad] (round 5): cl_elbo_mono_fwd_bwd also STORES dL/d(pre-activations of the FIRST Dense
                                   layer), feature-major like meta_t -- what cl_peel_backward turns into a peeled first layer's weight gradient.
                                   The default scaler's kernels only (elbo_lane.hip, elbo_narrow.hip): any other routing returns -2            */
    float* ev11_part;           /* deterministic mode with the Evans-2011 error model (round 4): every wave of the launch STORES its share of dL/d raw
                                   (Sdfac, Sdadd, SdB) at ev11_part[3 * (CL_EV11_WAVES * workgroup + wave)] instead of three float atomics on d_ev11;
                                   [3 * CL_EV11_WAVES * grid] floats, cl_det_reduce adds them in index order                                          */
} cl_mlp_args;

#define CL_EV11_WAVES 8          /* wave slots per workgroup in ev11_part (the kernels run four or eight waves) */
enum { CL_LIK_NORMAL_ = 0, CL_LIK_STUDENTT_ = 1 };
enum { CL_BIJ_EXP_ = 0, CL_BIJ_SOFTPLUS_ = 1 };

int cl_mlp_default_grid(void);                       /* workgroups of a persistent launch = CUs of the current device */
size_t cl_mlp_param_count(int d, int w, int L);      /* P */
int cl_mlp_max_layers_imgl(int w);                   /* hidden layers (Dense + per-image) one launch holds with n_imgl > 0         */
int cl_mlp_max_layers(int w);                        /* Dense layers ONE launch holds at hidden width w (0: width unsupported); deeper
                                                      * scalers are chained (act_out / dH_ext / dX_out)                          */
int cl_mlp_meta_rows(int d);                         /* rows of meta_t: d rounded up to a multiple of 4 (one MFMA k-step) */
int cl_elbo_mono_fwd_bwd(const cl_mlp_args* args, int grid, void* stream);
int cl_mlp_forward(const cl_mlp_args* args, int grid, void* stream);
int cl_mlp_backward_ext(const cl_mlp_args* args, int grid, void* stream);
/* Diagnostics: the name of the kernel instance the three calls above run for these arguments (mode 0 = cl_elbo_mono_fwd_bwd,
 * 1 = cl_mlp_forward, 2 = cl_mlp_backward_ext), e.g. "elbo_lane_kernel<10, 0, false>": what a rocprofv3 kernel trace lists.
 * Writes at most n bytes (NUL-terminated), returns the length of the name or < 0 for bad arguments.  No reference counterpart. */
int cl_mlp_kernel_name(const cl_mlp_args* args, int mode, char* out, size_t n);
/* grad_mlp[P] += sum over the `nparts` workgroup partials, in index order (deterministic) */
int cl_reduce_partials(const float* partials, int nparts, int P, float* grad_mlp, const int* stop_flag, void* stream);

/* Deterministic mode, second half: fixed-order sums of what cl_elbo_mono_fwd_bwd stored (dzf_obs / dimg_obs / nll_part).
 * perm_refl lists the shard's observations sorted by reflection (stable), seg_refl[r] .. seg_refl[r+1] is reflection r's range;
 * perm_img / seg_img the same by image.  dz_f[r][s] += sum in that order; d_img[m-1] += ... (image 0 is pinned); scalars[NLL] += sum
 * of nll_part in index order.                                                                                              */
typedef struct cl_det_args {
    const float* dzf_obs; const int* perm_refl; const int* seg_refl; int R, S; float* dz_f;      /* perm_refl NULL: records already in reflection order (det_slot) */
    const float* dimg_obs; const int* perm_img; const int* seg_img; int n_images; float* d_img;      /* d_img NULL: no image scales */
    const double* nll_part; int nparts; double* scalars;
    const int* stop_flag;
    const float* ev11_part; int n_ev11; float* d_ev11;      /* optional: n_ev11 wave slots of three floats (cl_mlp_args / cl_laue_args.ev11_part), d_ev11[3] += their sums */
} cl_det_args;
int cl_det_reduce(const cl_det_args* args, void* stream);

/* --- scalers wider than 64 ---------------------------------------------------------------------------------------------
 * replaces: the Dense stack of MetadataScaler / MLPScaler (careless/models/scaling/nn.py:55-68, 92-120) and its gradient
 *           (careless/models/merging/variational.py:197-202) when the hidden or the metadata width exceeds what the fused kernels
 *           hold (64): one fp32-MFMA GEMM per layer and direction, activations in HBM, in row chunks chosen by the caller.
 * Activation buffers are row-major [rows][ld], ld = cl_wide_ld(width) = the width rounded up to 4.
 * Weights in the W^T layout of the flat parameter vector (Wt[out][in], then b[out]).
 * Call order per row chunk, generic form (any width): cl_wide_dense_forward x L -> cl_wide_head_forward -> [likelihood: cl_slot_rows when
 * every row is its own slot, cl_laue_predict / _likelihood / _backward otherwise] -> cl_wide_dense_forward x L again unless the activations
 * were kept -> cl_wide_head_backward -> per layer, top down: cl_wide_dense_wgrad (+ cl_reduce_partials into the layer's gradient slice),
 * cl_wide_dense_dgrad.  Widths 65 .. 128 take the fused forms declared below instead (round 4): cl_wide_dense2_forward (layers 0 + 1),
 * cl_wide_dense_forward_head (top layer + head), cl_wide_dense_wgrad_head / _dgrad_head (top layer's backward with the head's inside),
 * cl_wide_dense_wgrad_pre and cl_wide_dense_dgrad_pre_wgrad0 (second layer's backward with the first layer recomputed and its weight
 * gradient taken on the way).                                                                                                       */
int cl_wide_ld(int width);
/* Y[n][n_out] = act(X[n][n_in] Wt^T + b), act = LeakyReLU(leak) or identity */
int cl_wide_dense_forward(const float* X, int ldx, const float* Wt, const float* b, long long n, int n_in, int n_out, float leak, int act,
                          float* Y, int ldy, const int* stop_flag, void* stream);
/* The TOP Dense layer with the Dense(2) head in its epilogue (round 4; replaces cl_wide_dense_forward + cl_wide_head_forward for layers up
 * to 128 x 128, -2 beyond): Y as above -- the backward pass needs it -- and loc = Y . Wo[0] + bo[0], sigma = bijector(Y . Wo[1] + bo[1]) + eps
 * per row from the registers the activations are in (reference: NormalLayer, careless/models/scaling/nn.py:10-25, 84-87).
 * head = [Wo^T (2 x n_out) | bo (2)].  dsig_draw_out (or NULL): the bijector's derivative d sigma / d raw per row, which the fused head
 * backward below takes instead of recomputing the head.                                                                               */
int cl_wide_dense_forward_head(const float* X, int ldx, const float* Wt, const float* b, long long n, int n_in, int n_out, float leak,
                               float* Y, int ldy, const float* head, int bij_kind, float eps, float* loc_out, float* sig_out, float* dsig_draw_out,
                               const int* stop_flag, void* stream);
/* ... and with the slot likelihood of the call's rows in the same epilogue (round 4): what cl_slot_rows computes from (loc, sigma) per row
 * -- sample, predict, log-prob (careless/models/likelihoods/mono.py:10-37 on variational.py:167's prediction), gradient -- where the two
 * numbers are made.  `lik` as for cl_slot_rows, its per-row arrays (refl_id, image_id, iobs, sig, dO, row_index) at the call's first row,
 * obs_offset = that row's global number, n_obs = n (loc / sigma / iconv unused).  Rows that are their own slot, in-kernel noise, no
 * ipred_out, no Evans-2011 terms, no deterministic stores, widths 65 .. 128 with the same block count on both sides: -2 otherwise (the
 * caller then runs cl_wide_dense_forward_head and cl_slot_rows).                                                                    */
struct cl_laue_args;
int cl_wide_dense_forward_head_lik(const float* X, int ldx, const float* Wt, const float* b, long long n, int n_in, int n_out, float leak,
                                   float* Y, int ldy, const float* head, int bij_kind, float eps, float* loc_out, float* sig_out, float* dsig_draw_out,
                                   const struct cl_laue_args* lik, const int* stop_flag, void* stream);
/* The head's backward pass inside the TOP layer's weight gradient and dgrad (round 4; replaces cl_wide_head_backward and the dZ_L buffer it
 * wrote: tape.gradient through NormalLayer, careless/models/scaling/nn.py:10-25, variational.py:197-202).  dZ_L = (g Wo) * LeakyReLU'(h_L) with
 * g = (dO[row][0], dO[row][1] * dsig_draw[row]) is a rank-2 product behind a mask: both kernels make it from Htop = h_L (the same bytes per
 * row) where they read their dZ operand.  cl_wide_dense_wgrad_head also writes the head's own partial sums head_partials[s][2 n_out + 2] =
 * (dWo | dbo).  Layers of width 65 .. 128 with the same number of 16-column blocks on both sides (cl_wide_head_bwd_supported), -2 otherwise. */
int cl_wide_head_bwd_supported(int n_out, int n_in);
int cl_wide_dense_wgrad_head(const float* Htop, int ldt, const float* head, const float* dO, const float* dsig_draw, float leak, const float* H, int ldh,
                             long long n, int n_out, int n_in, float* partials, float* head_partials, int nsplit, const int* stop_flag, void* stream);
int cl_wide_dense_dgrad_head(const float* Htop, int ldt, const float* head, const float* dO, const float* dsig_draw, const float* Wt, long long n,
                             int n_out, int n_in, const float* Hprev, int ldh, float leak, float* dX, int ldo, const int* stop_flag, void* stream);
/* dX[n][n_in] = (dZ[n][n_out] Wt) * LeakyReLU'(Hprev[n][n_in])   (Hprev = the layer's input = the previous layer's output; NULL: no mask) */
int cl_wide_dense_dgrad(const float* dZ, int lddz, const float* Wt, long long n, int n_out, int n_in, const float* Hprev, int ldh, float leak,
                        float* dX, int ldo, const int* stop_flag, void* stream);
/* The FIRST Dense layer recomputed instead of stored (round 4): with at most 15 metadata columns and a hidden width of at most 128
 * (cl_wide_pre_supported) its output h_0 = LeakyReLU(X0 Wt0^T + b0) is an eighth of a 128 x 128 layer's work, so it is made again
 * wherever it is needed -- 4 w bytes per row are never written, nor read three times:
 *   cl_wide_dense2_forward   layers 0 and 1 in one launch (h_0 stays in registers, in the operand layout of layer 1); `head` non-NULL:
 *                            the Dense(2) head in the epilogue as in cl_wide_dense_forward_head (a two-layer scaler);
 *   cl_wide_dense_dgrad_pre  layer 1's dgrad, the mask LeakyReLU'(h_0) recomputed from the metadata (replaces Hprev);
 *   cl_wide_dense_wgrad_pre  layer 1's weight gradient, its input operand h_0 recomputed while the tiles are staged (replaces H);
 *   cl_wide_dense_dgrad_pre_wgrad0   layer 1's dgrad AND layer 0's weight gradient in one launch: the dgrad's output dZ_0 is contracted
 *                            with the metadata rows where it is produced and never stored.  Writes cl_wide_dgrad_wgrad0_parts(n) partial sums
 *                            of [dWt_0 (n_in x n_in0) | db_0 (n_in)] into `partials` (cl_reduce_partials adds them); -2 outside the
 *                            square-layer kernel's envelope (widths 65 .. 128, same number of 16-column blocks on both sides): the caller
 *                            then runs cl_wide_dense_dgrad_pre + cl_wide_dense_wgrad.
 * X0 = metadata rows [n][ldx0] (ldx0 = cl_wide_ld(n_in0), 16-byte aligned, padding columns zero).                                   */
int cl_wide_pre_supported(int n_in0, int w);
int cl_wide_dense2_forward(const float* X0, int ldx0, int n_in0, const float* Wt0, const float* b0, const float* Wt1, const float* b1, long long n,
                           int w0, int w1, float leak, float* Y, int ldy, const float* head, int bij_kind, float eps, float* loc_out, float* sig_out,
                           const int* stop_flag, void* stream);
int cl_wide_dense_dgrad_pre(const float* dZ, int lddz, const float* Wt, long long n, int n_out, int n_in, const float* X0, int ldx0, int n_in0,
                            const float* Wt0, const float* b0, float leak, float* dX, int ldo, const int* stop_flag, void* stream);
int cl_wide_dense_wgrad_pre(const float* dZ, int lddz, const float* X0, int ldx0, int n_in0, const float* Wt0, const float* b0, float leak, long long n,
                            int n_out, int n_in, float* partials, int nsplit, const int* stop_flag, void* stream);
int cl_wide_dgrad_wgrad0_parts(long long n);
int cl_wide_dense_dgrad_pre_wgrad0(const float* dZ, int lddz, const float* Wt, long long n, int n_out, int n_in, const float* X0, int ldx0, int n_in0,
                                   const float* Wt0, const float* b0, float leak, float* partials, const int* stop_flag, void* stream);
/* partials[s][n_out * n_in + n_out] = (dWt | db) over the s-th of nsplit row ranges; H is the layer's input */
int cl_wide_wgrad_splits(long long n);
int cl_wide_dense_wgrad(const float* dZ, int lddz, const float* H, int ldh, long long n, int n_out, int n_in, float* partials, int nsplit,
                        const int* stop_flag, void* stream);
/* Per-image layers of NeuralImageScaler (careless/models/scaling/image.py:66-125) on this path: rows sorted by image, seg[g] ..
 * seg[g+1] = the rows of the g-th image of the call; W / b / dW / db point at that first image's kernel (w x w, (out, in)) and bias.
 * Width <= 128 (-2 beyond).  cl_wide_image_wgrad WRITES the images' gradients (the caller keeps an image's rows in one call). */
int cl_wide_image_forward(const float* X, int ldx, const float* W, const float* b, const int* seg, int n_groups, long long n, int w, float leak,
                          float* Y, int ldy, const int* stop_flag, void* stream);
int cl_wide_image_dgrad(const float* dZ, int lddz, const float* W, const int* seg, int n_groups, long long n, int w, const float* Hprev, int ldh, float leak,
                        float* dX, int ldo, const int* stop_flag, void* stream);
int cl_wide_image_wgrad(const float* dZ, int lddz, const float* H, int ldh, const int* seg, int n_groups, long long n, int w, float* dW, float* db,
                        const int* stop_flag, void* stream);
/* ... wider than 128 (round 4: the reference has no limit, careless/models/scaling/image.py:66-125): forward and dgrad on the tiled kernel, one
 * workgroup column per entry of `tiles` = n_tiles (group, first row) pairs covering every group's rows in 128-row pieces (the caller
 * builds the list from its row counts); cl_wide_image_wgrad takes any width as it is */
int cl_wide_image_forward_tiles(const float* X, int ldx, const float* W, const float* b, const int* seg, const int* tiles, int n_tiles, int w, float leak,
                                float* Y, int ldy, const int* stop_flag, void* stream);
int cl_wide_image_dgrad_tiles(const float* dZ, int lddz, const float* W, const int* seg, const int* tiles, int n_tiles, int w, const float* Hprev, int ldh,
                              float leak, float* dX, int ldo, const int* stop_flag, void* stream);
/* Dense(2) head: Wo = [Wo^T (2 x w) | bo (2)]; forward writes loc and sigma = bijector(raw) + eps per row; backward takes
 * dO[n][2] = dL/d(loc, sigma), writes dZ of the top layer and partials[nblocks][2 w + 2] of the head's gradient              */
int cl_wide_head_forward(const float* H, int ldh, const float* Wo, long long n, int w, int bij_kind, float eps, float* loc_out, float* sig_out,
                         const int* stop_flag, void* stream);
int cl_wide_head_blocks(long long n);
int cl_wide_head_backward(const float* H, int ldh, const float* Wo, const float* dO, long long n, int w, int bij_kind, float eps, float leak,
                          float* dZ, int lddz, float* partials, int nblocks, const int* stop_flag, void* stream);

/* --- Laue harmonic deconvolution -----------------------------------------------------------------------------------
 * replaces: ConvolvedLikelihood.convolve / .log_prob, LaueBase.call (careless/models/likelihoods/laue.py:9-47) and their gradient.
 * Call order inside a step: cl_mlp_forward -> cl_laue_predict -> cl_laue_likelihood -> cl_laue_backward -> cl_mlp_backward_ext.
 * iobs / sig hold one entry per SLOT: valid in [0,G), padding beyond (careless/io/formatter.py:637-640); every slot counts.   */
typedef struct cl_laue_args {
    const int* refl_id;         /* [n_obs]                                   */
    const int* image_id;        /* [n_obs]                                   */
    const int* harmonic_id;     /* [n_obs] in [0, n_obs); NULL: row i is slot i (monochromatic rows), iconv needs no clearing */
    const float* loc;           /* [n_obs] scaler mean  (cl_mlp_forward)     */
    const float* sigma;         /* [n_obs] scaler sigma (cl_mlp_forward)     */
    const float* iobs;          /* [n_obs] per slot                          */
    const float* sig;           /* [n_obs] per slot                          */
    int n_obs;
    long long obs_offset;
    const float* img; int use_img;
    const float* z_f; int R, S;
    int lik_kind; float dof, lik_const;
    float shift, w_ll;
    const float* eta;           /* [n_obs][S] injected normals or NULL       */
    unsigned long long seed; unsigned step;
    float* iconv;               /* [n_obs][S] zeroed by the caller; predict accumulates the group sums, likelihood overwrites
                                   them with dNLL/diconv                     */
    float* dz_f;                /* [R][S] +=                                 */
    float* d_img;               /* [M-1] +=                                  */
    float* dO;                  /* [n_obs][2] dL/d(loc, sigma) for cl_mlp_backward_ext; cl_laue_backward takes NULL (a frozen scaling model:
                                   dz_f only, d_img untouched) */
    double* scalars;
    float* ipred_out;           /* optional [n_obs][S]                       */
    const int* stop_flag;
    const float* ev11;          /* [3] raw Sdfac, Sdadd, SdB or NULL (laue.py:49-65) */
    float* d_ev11;              /* [3] +=                                     */
    const long long* row_index; /* optional [n_obs]: global row number of every local row (noise key) when the shard is not a
                                   contiguous range (data-parallel Laue keeps harmonic groups on one rank); NULL = obs_offset + i */
    double* nll_part;           /* optional [CL_LAUE_LIK_MAX_BLOCKS] (deterministic mode): cl_laue_likelihood / cl_slot_rows STORE every
                                   workgroup's NLL here -- slots past the grid are left alone -- instead of adding it to scalars with an atomic */
    /* deterministic mode of cl_slot_rows (all three or none; S must divide 64: a row's samples inside one wave): the amplitude gradient
     * of (row i, sample s) is STORED at dzf_obs[(det_slot ? det_slot[i] : i) * S + s] and the row's image-scale term at dimg_obs[i];
     * cl_det_reduce sums them per reflection / image in a fixed order (as cl_mlp_args.dzf_obs / dimg_obs / det_slot)               */
    float* dzf_obs; float* dimg_obs; const int* det_slot;
    float* ev11_part;           /* deterministic mode with Evans-2011: cl_laue_likelihood / cl_slot_rows store every wave's share of dL/d raw at
                                   ev11_part[3 * (4 * workgroup + wave)] (256-thread workgroups) instead of adding it to d_ev11 with atomics   */
} cl_laue_args;
#define CL_LAUE_LIK_MAX_BLOCKS 2048

int cl_laue_predict(const cl_laue_args* args, void* stream);
int cl_laue_likelihood(const cl_laue_args* args, void* stream);
int cl_laue_backward(const cl_laue_args* args, void* stream);
/* The three calls above in ONE launch for rows that are their own slot (harmonic_id NULL: the monochromatic likelihood of
 * careless/models/likelihoods/mono.py:10-73 on the layer-by-layer path of scalers wider than 64): predict, log-prob and its gradient back to
 * dz_f / d_img / dO per (row, sample) with nothing in between; iconv is not touched.  -2 when harmonic_id is set.                    */
int cl_slot_rows(const cl_laue_args* args, void* stream);

/* --- the data term of a step whose scaling model is frozen (round 6) -------------------------------------------------------------------
 * replaces: VariationalMergingModel.call + the likelihood's log_prob + tape.gradient over the TRAINABLE variables only
 *           (careless/models/merging/variational.py:156-181, 197-202; models/likelihoods/mono.py:10-73) in the trainings whose scaling
 *           model has trainable = False: `--freeze-scales` and the half-dataset trainings of `--merge-half-datasets`
 *           (careless/careless.py:48-50, 102-128).  The scaler's output is then a constant of the training: the caller takes (loc, sigma)
 *           and the image scale of every row once and SORTS THE ROWS BY REFLECTION (any order is as good as another for a constant).
 * Per step: sample the scale (Philox keyed by the row's global number `key`: the fused kernels' draws), predict, log-prob into
 * scalars[NLL], amplitude gradient into dz_f -- equal-reflection runs are summed inside the wave and leave as ONE plain store per
 * (reflection, sample); runs that cross a wave border go through `edge_val / edge_rid` and a second small launch of the same call.  No
 * float atomic touches dz_f: the result does not depend on the run (accumulate = 1 -- several calls share dz_f, e.g. the pieces of a
 * shard -- adds with atomics instead of storing).  Rows with refl_id < 0 are skipped; monochromatic rows only (every row its own slot). */
typedef struct cl_frozen_args {
    const int* refl_id;         /* [n] ascending                                                                   */
    const float* loc;           /* [n] scaler mean per row (cl_mlp_forward / the wide path), in the sorted order   */
    const float* sigma;         /* [n] scaler sigma per row                                                        */
    const float* aim;           /* [n] image scale per row, or NULL (= 1): image.py:53-63 with the scale frozen    */
    const float* iobs;          /* [n]                                                                             */
    const float* sig;           /* [n]                                                                             */
    const int* key;             /* [n] global row number of every row: noise key, row of eta / ipred_out (minus obs_offset); NULL: obs_offset + i */
    long long obs_offset;
    long long n;
    int R, S;
    const float* z_f;           /* [R][S]                                                                          */
    float* dz_f;                /* [R][S] stored (accumulate = 0) or += (accumulate = 1) for the reflections that have rows */
    int accumulate;
    int lik_kind; float dof, lik_const;
    float shift, w_ll;
    const float* eta;           /* optional [rows][S] injected normals (parity tests)                              */
    unsigned long long seed; unsigned step;
    double* scalars;
    float* ipred_out;           /* optional [rows][S]                                                              */
    const int* stop_flag;
    const float* ev11;          /* [3] raw Sdfac, Sdadd, SdB or NULL (mono.py:39-73)                               */
    float* d_ev11;              /* [3] +=                                                                          */
    int* edge_rid;              /* [2 * ceil(n / 64)] workspace                                                    */
    float* edge_val;            /* [cl_frozen_edge_floats(n, S)] workspace                                         */
    double* nll_part;           /* optional [cl_frozen_grid(n)]: every workgroup STORES its NLL (no fp64 atomic)   */
    float* ev11_part;           /* optional [3 * 4 * cl_frozen_grid(n)]: every wave stores its Evans-2011 terms    */
    /* Harmonic groups (Laue data, careless/models/likelihoods/laue.py:9-47: the predictions of a group's rows sum before the likelihood) take
     * TWO calls, because the rows of a group belong to different reflections:
     *   1. gmeta != NULL: rows in the packed order of the single-pass kernels (a group inside a 16-row granule; gmeta[i] = member index |
     *      group size << 8; padding rows refl_id < 0; iobs / sig of the group replicated on its rows): group sums over shuffles, likelihood,
     *      NLL (member 0), the row's amplitude gradients STORED at gbuf[src ? src[i] : i][s] (src[i] < 0: nowhere) -- with src[i] = the row's
     *      position in reflection order the second call reads gbuf front to back.  refl_id need not be sorted; dz_f / edge_* unused.
     *   2. gmeta == NULL, gbuf != NULL: refl_id ascending over the rows that have a reflection; row i's gradients at gbuf[src ? src[i] : i]: summed
     *      per reflection and stored into dz_f as for monochromatic rows (loc / sigma / iobs / sig / z_f / scalars unused).             */
    const int* gmeta;
    float* gbuf;                /* [rows][S]                                                                        */
    const int* src;
} cl_frozen_args;
int cl_frozen_rows(const cl_frozen_args* args, void* stream);
int cl_frozen_edge_floats(long long n, int S);     /* floats of edge_val */
int cl_frozen_grid(long long n);                   /* workgroups of the launch (nll_part / ev11_part slots) */
size_t cl_frozen_args_size(void);                  /* sizeof(cl_frozen_args) as the library was compiled (binding check, like cl_abi_sizes) */

/* --- gradient norm, sanitise, clip, Adam -------------------------------------------------------------------------
 * replaces: tf.linalg.global_norm, tf.where(is_finite), optimizer.apply_gradients (variational.py:202-209)
 *           tfk.optimizers.Adam(lr, b1, b2, clipnorm, clipvalue, global_clipnorm) (careless/io/manager.py:494-501) */
typedef struct cl_adam_args {
    float* p; const float* g; float* m; float* v;   /* [n] each */
    int n;
    float alpha;                /* lr sqrt(1-b2^t)/(1-b1^t), computed on the host from the step count */
    float beta1, beta2, adam_eps;
    float clipnorm, clipvalue, global_clipnorm;     /* <= 0: off */
    const int* seg_off;         /* [nseg+1] tensor boundaries in the flat buffer */
    int nseg;
    const double* seg_sq;       /* [nseg] per-tensor squared norms (only read when clipnorm > 0) */
    const unsigned char* frozen; /* optional [nseg]: 1 = tensor is not trainable (--freeze-*) */
    const double* scalars;
    const int* stop_flag;
    double* norm_out;           /* optional [CL_SC_COUNT]: also accumulate the squared gradient norm here (replaces a separate
                                   cl_grad_sqnorm launch when no norm-dependent clipping is configured)                      */
    /* Reflection-owner data parallelism: a rank updates only the index ranges it owns (its reflections' a and b, then the replicated
     * tail) and learns the other ranks' share of the gradient norm from the step's message.  n_ranges = 0: the whole vector.      */
    int n_ranges;               /* 0..3 */
    int range_begin[3], range_end[3];   /* absolute index ranges [begin, end) of p / g / m / v this call updates                 */
    int norm_skip_ranges;       /* the first k ranges stay out of the norm fused into this call (norm_out) ...                     */
    const float* norm_extra;    /* ... because their squared norm over ALL ranks arrives here: [2] = raw, sanitised (cl_owner_qnorm
                                   + the all-reduce); NULL = nothing to add                                                       */
    double* norm_part;          /* optional [2 * cl_adam_grid(args)] workspace: the workgroups' two norm sums are STORED here (raw, sanitised per
                                   workgroup) instead of added to norm_out by same-address atomics, and cl_step_finalize adds them up     */
} cl_adam_args;

/* frozen (optional [nseg], as cl_adam_args.frozen): tensors that are not trainable stay out of the norm -- the reference takes
 * tf.linalg.global_norm of tape.gradient(loss, self.trainable_variables) (variational.py:201-205); cl_adam_step's fused norm does the same */
int cl_grad_sqnorm(const float* g, int n, const int* seg_off, int nseg, double* seg_sq, double* scalars,
                   const unsigned char* frozen, const int* stop_flag, void* stream);
int cl_adam_step(const cl_adam_args* args, void* stream);
/* Reflection-owner data parallelism (no reference counterpart; the reference is single-process): the squared norm of THIS rank's
 * part of the surrogate-posterior gradient -- g[r_begin .. r_end) (d a) and g[R + r_begin .. R + r_end) (d b) -- as four floats the
 * step's all-reduce sums over the ranks next to the replicated tail: out[0] = raw (NaN / inf propagate, variational.py:205),
 * out[1] = sanitised (what the optimizer's clipping sees, :208), out[2] / out[3] = sanitised, per tensor (clipnorm).  Accumulated in
 * double in scratch[0..3] (scratch[4] is the block ticket; the caller zeroes all five per step).                               */
int cl_owner_qnorm(const float* g, int R, int r_begin, int r_end, float* out, double* scratch, const int* stop_flag, void* stream);
/* history[step_index] = {loss, F KLDiv, NLL, Grad Norm, skipped}; sets *stop_flag when the norm is not finite
 * (careless/models/merging/variational.py:262-274).  norm_part (optional, n_norm_part workgroups' pairs left by cl_adam_step) is added,
 * in index order, to scalars[CL_SC_GNORM2 / _SANE] first. */
int cl_step_finalize(double* scalars, float kl_weight_or_one, double* history, int step_index, int* stop_flag,
                     const double* norm_part, int n_norm_part, void* stream);
/* workgroups cl_adam_step launches for these arguments (the length / 2 of norm_part) */
int cl_adam_grid(const cl_adam_args* args);

/* --- a wide first layer in front of the default scaler's kernels ("peeled" first layer, round 5) -----------------------------------
 * replaces: the FIRST Dense layer of MetadataScaler / MLPScaler (careless/models/scaling/nn.py:55-68: tfk.layers.Dense on the metadata) and
 *           its gradient (careless/models/merging/variational.py:197-202) when the metadata width exceeds what the lane / narrow kernels
 *           hold -- e.g. 20 x 10 on the 37 columns of four positionally encoded keys (careless/args/positional_encoding.py:24-37).
 * cl_peel_forward : u_t[k][i] = b_0[k] + sum_c W_0[k][c] x[c][i]  -- the layer's PRE-activations, feature-major [cl_mlp_meta_rows(w)][n_pad]
 *                   like meta_t -- and mlp_peel = the scaler's parameters with layer 0 replaced by (identity w x w, zero bias)
 *                   (cl_mlp_param_count(w, w, L) floats).  cl_elbo_mono_fwd_bwd on (meta_t = u_t, d = w, mlp = mlp_peel, dZ0_out = dz0_t)
 *                   then computes the same step: LeakyReLU(I u + 0) is the original first activation, and dL/du = dZ_0 comes back in dZ0_out
 *                   (elbo_lane.hip / elbo_narrow.hip store it).  zero_ptr[0 .. zero_n): cleared on the way (the caller's grad_peel).
 * cl_peel_backward: grad_mlp[layer 0] += dZ_0^T X, sum dZ_0 (per-workgroup partials [nparts][w d + w], nparts <= cl_peel_parts(n_obs), summed
 *                   in index order), grad_mlp[behind layer 0] += grad_peel[behind its layer 0] (grad_peel = cl_reduce_partials of the launch).
 * `mlp` / `grad_mlp` point at the scaler's slice of the flat parameter / gradient vector (W^T layout).  cl_peel_supported: d > w, w <= 15,
 * w (d + 1) <= 1280.                                                                                                                     */
int cl_peel_supported(int d, int w, int L);
int cl_peel_parts(long long n_obs);
int cl_peel_forward(const float* meta_t, int n_obs, int n_pad, int d, int w, int L, const float* mlp, float* u_t, float* mlp_peel,
                    float* zero_ptr, int zero_n, const int* stop_flag, void* stream);
int cl_peel_backward(const float* meta_t, int n_obs, int n_pad, int d, int w, int L, const float* dz0_t, const float* grad_peel, float* grad_mlp,
                     float* partials, int nparts, const int* stop_flag, void* stream);
/* The last block of a layer-block chain on the default scaler's kernels (round 6): a scaler of more than 20 layers at width <= 10 runs its
 * last 20 layers on elbo_lane.hip, which stores dZ_0 = dL/d(pre-activations of the block's first layer) (cl_mlp_args.dZ0_out); the block in
 * front takes dL/d(its output activations) as dH_ext: dX = W_0^T dZ_0 per observation.  dz0_t [cl_mlp_meta_rows(w_out)][n_pad] and dx_t
 * [cl_mlp_meta_rows(w_in)][n_pad] feature-major like meta_t, Wt [w_out][w_in] as in the flat layout; widths <= 15 (-2 beyond).
 * replaces: the tape's step through the first Dense kernel of the block (careless/models/scaling/nn.py:55-68, variational.py:197-202).        */
int cl_chain_dx(const float* dz0_t, const float* Wt, int n_obs, int n_pad, int w_out, int w_in, float* dx_t, const int* stop_flag, void* stream);

/* --- output step: merged amplitudes ---------------------------------------------------------------------------------
 * replaces: TruncatedNormal.mean / .stddev (tfd.TruncatedNormal moments behind SurrogatePosterior.mean / .stddev,
 *           careless/models/merging/surrogate_posteriors.py:23-27, 45-48) and TruncatedNormal._tf_moment_4 / moment_4
 *           (careless/models/merging/surrogate_posteriors.py:55-102), as DataManager.get_results consumes them
 *           (careless/io/manager.py:188-197: F = mean, SigF = stddev, var(I) = <F^4> - <F^2>^2).
 * From the raw trainable vectors a = log(loc), b = log(scale - eps) and the lower truncation point, per reflection:
 *   mean[r]  = E[z],  std[r] = sqrt(Var[z]),  m4[r] = E[z^4]   for z ~ Normal(loc, scale) truncated to [low, high].
 * high >= 1e30 (or inf) is "no upper truncation", the reference's moment_4 default (high = np.inf).  The arithmetic is fp64 on the
 * device (R values once per run; the subtraction <F^4> - <F^2>^2 downstream cancels): mean / std are stored as fp32 like the
 * reference's fp32 TFP moments, m4 as fp64 like scipy.stats.truncnorm.moment.  Any output pointer may be NULL.                     */
int cl_tn_moments(const float* q_loc_raw, const float* q_scale_raw, const float* low, int R, double high_moments, double high_m4, float eps,
                  float* mean, float* std, double* m4, void* stream);

/* --- output step: posterior predictive moments per observation -------------------------------------------------------------------
 * replaces: VariationalMergingModel.prediction_mean_stddev (careless/models/merging/variational.py:80-121), consumed by
 *           DataManager.get_predictions (careless/io/manager.py:89-161): per observation i of reflection r = refl_id[i], with the scale's
 *           moments of the row (scale_mean, scale_std: the scaler's forward pass) and the posterior's per reflection (cl_tn_moments),
 *             iexp[i] = scale_mean[i] (f_mean[r]^2 + f_std[r]^2),   ivar[i] = f_m4[r] (scale_mean[i]^2 + scale_std[i]^2) - iexp[i]^2
 *           in fp64 (Ipred = iexp, SigIpred = sqrt(ivar); Laue data: the caller sums both over a harmonic group's rows first, :113-119).
 *           refl_id outside [0, R): zeros.                                                                                              */
int cl_predict_moments(const float* scale_mean, const float* scale_std, const int* refl_id, long long n, const float* f_mean, const float* f_std,
                       const double* f_m4, int R, double* iexp, double* ivar, void* stream);

/* --- formatting step: symmetry bookkeeping of the reflection tables (HOST pointers, host threads; no stream) ---------------------------
 * replaces: DataSet.remove_absences(), DataSet.hkl_to_asu(anomalous=...) and the centric / multiplicity labels the reference takes from
 *           reciprocalspaceship / gemmi (C++) while formatting -- careless/io/formatter.py:285-302, 319 (MonoFormatter.prep_dataset),
 *           :540-562 (LaueFormatter), careless/io/asu.py:27-56 (ReciprocalASU) -- and pandas' groupby(...).ngroup() behind the image ids
 *           (careless/io/formatter.py:203).
 * cl_host_asu_map: for every Miller index hkl[i] (int32 [n][3]) under the operators rot[o] (int32 [nops][3][3], acting on row vectors:
 *           h' = h R) and trans[o] (fractions of the cell, [nops][3]):
 *             eps[i]     = number of operators with h R = h                              (multiplicity of the reflection)
 *             absent[i]  = any such operator with h . t not an integer (tolerance 1e-6)   (systematic absence)
 *             centric[i] = any operator with h R = -h
 *             hasu[i]    = the reciprocal-ASU representative: the first member of (h R_0 .. h R_{nops-1}, -h R_0 .. -h R_{nops-1}) inside
 *                          the CCP4 inequality set `asu_case` (0 .. 9: -1, 2/m, mmm, 4/m and 6/m, 4/mmm and 6/mmm, -3, -31m, -3m1, m-3, m-3m
 *                          in their reference settings; gemmi's ReciprocalAsu::is_in), or with asu_case = -1 the member with the largest
 *                          (h, k, l) in lexicographic order; anomalous != 0: a representative that no ROTATION image reaches is a
 *                          Friedel-minus and is stored negated, as hkl_to_asu(anomalous=True) does.
 *           Any output pointer may be NULL (absent needs trans).  nthreads <= 0: the cores the process may run on, at most 32.
 * cl_host_dense_ids: ids[i] = rank of key[i] among the DISTINCT keys in ascending order (groupby(...).ngroup() of one integer key),
 *           *n_groups = their number; all keys inside [key_min, key_max].  A presence table over the range (4 bytes per slot): returns -2 when
 *           the range exceeds 2^31 slots, -3 when the table cannot be allocated (the caller sorts then).                                                                                  */
int cl_host_asu_map(const int32_t* hkl, long long n, const int32_t* rot, const double* trans, int nops, int asu_case, int anomalous,
                    int32_t* hasu, uint8_t* centric, int32_t* eps, uint8_t* absent, int nthreads);
int cl_host_dense_ids(const int64_t* key, long long n, int64_t key_min, int64_t key_max, int64_t* ids, long long* n_groups, int nthreads);
/* CrystFEL `.stream` files (replaces: rs.read_crystfel behind careless/io/formatter.py:179-184 -- serial-crystallography input, text files of
 * 10^7 .. 10^8 reflection lines): the indexed reflection lists of every crystal in the buffer `buf` (the file's bytes, e.g. a read-only
 * memory map) as one table.  cl_host_crystfel_count returns the number of rows (list lines with at least nine blank-separated fields; < 0:
 * error) and the number of "--- Begin crystal" lines; cl_host_crystfel_parse fills cols[10][n_rows] (column-major, fp32):
 * h, k, l, I, sigma(I), peak, background, fs/px, ss/px, BATCH = number of the crystal (0-based).  Integers as int(), reals as float() of the
 * field (strtoll / strtod) stored as fp32; -5: a field that is not a number.  Lists are parsed in parallel on host threads.              */
long long cl_host_crystfel_count(const char* buf, long long nbytes, long long* n_crystals, int nthreads);
int cl_host_crystfel_parse(const char* buf, long long nbytes, long long n_rows, float* cols, int nthreads);

/* --- diagnostics -------------------------------------------------------------------------------------------------- */
const char* cl_version(void);
/* sizeof(cl_tn_args), sizeof(cl_mlp_args), sizeof(cl_adam_args), sizeof(cl_laue_args), sizeof(cl_det_args): lets a binding verify its mirrors */
void cl_abi_sizes(size_t out[5]);
/* out[n][S]: kind 0 = the uniforms of cl_tn_*, kind 1 = the normals of cl_elbo_mono_fwd_bwd, for (seed, step) */
int cl_debug_noise(unsigned long long seed, unsigned step, int S, long long n, long long offset, int kind, float* out,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CARELESS_HIP_H */
