// Fused scaling-model + likelihood kernel of the ELBO step (gfx950 / CDNA4 only).
//
// One launch does, for every observation of this rank's shard (reference call sites in brackets):
//   forward   metadata -> L x [Dense(w) + LeakyReLU] -> Dense(2)            [careless/models/scaling/nn.py:92-120]
//   sample    z_Sigma = a_img (loc + sigma eta + shift)                      [variational.py:157, image.py:53-63]
//   predict   ipred = z_Sigma * z_f[refl_id]^2                               [variational.py:167]
//   likeli.   Normal / Student-T log-prob, NLL partial sums                  [likelihoods/mono.py:10-37, variational.py:174-181]
//   backward  d/d ipred -> atomics into dz_f and d(image scales); dO -> dgrad + wgrad of every Dense layer
//                                                                            [variational.py:197-202 tape.gradient]
// Nothing per-observation is written to HBM: activations stay in registers between forward and backward.
//
// Work decomposition (MI355X-first):
//   * persistent workgroups (one per CU), 8 waves = two per SIMD, <= 256 registers each, so one wave's VALU / LDS
//     work (LeakyReLU, staging, epilogue) overlaps its SIMD partner's MFMAs;
//   * a workgroup walks tiles of 128 observations; in forward / dgrad a wave owns 16 observations and keeps the
//     transposed activations H^T (feature x observation) in MFMA accumulator layout, so each layer's output tile
//     IS the next layer's B operand (v_mfma_f32_16x16x4_f32: exact fp32; the k-grouping of every step is chosen
//     to match the C layout: step (kb,t) contracts features 16kb + 4q + t, q = lane>>4);
//   * weights live in LDS as W^T with row pitch w+4 (ds_read_b128 feeds 4 forward steps, ds_read_b32 feeds dgrad);
//   * wgrad contracts over observations, which sit on the MFMA lane axis, so dZ_l and H_{l-1} are staged once
//     through two LDS tiles [feature][128 obs (+4)]; each wave then accumulates its own 16x16 blocks of dW_l^T over
//     the whole tile, in registers, across ALL tiles of the kernel (no per-tile global traffic for weight grads);
//   * per-workgroup weight-gradient partials are written once at kernel end; a tiny second kernel sums them in a
//     fixed order (deterministic);
//   * after every workgroup barrier the eight waves restart in lockstep, so a latency that one wave waits for is waited for
//     by all of them: LDS operands are double-buffered by hand one MFMA group ahead (CL_SCHED_FENCE) and the post-barrier
//     latency windows are filled with independent work (bias-gradient reads, the next layer's dZ, staging writes between the
//     dgrad MFMAs) -- DESIGN.md section 4.1.
// The file is compiled eight times (build.py): plain; packed layout + per-image layers (-DCL_IMGL=1, NeuralImageScaler); packed layout only
// (-DCL_IMGL=2, single-pass Laue: harmonic group sums as lane reductions in the epilogue); layer-block chains (-DCL_CHAIN=1, scalers deeper
// than one launch holds); and the plain, packed and chain forms once more with the epilogue's atomics turned into stores (-DCL_DET=1, the
// deterministic mode of include/careless_hip.h: dzf_obs / dimg_obs / nll_part).
// Roofline: fp32 MFMA (157.3 TFLOP/s); algorithmic flops per observation 6 (d w + (L-1) w^2 + 2 w).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "cl_math.h"
#include "cl_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef CL_META_RANGE
#define CL_META_RANGE 2   /* largest KS1 (metadata k-steps) that takes the branch-free metadata prefetch */
#endif
#ifndef CL_ACC_WP
#define CL_ACC_WP 16   /* widest instance with LDS-resident accumulators (32: measured -0.8 % on a 10 x 32 scaler) */
#endif
#ifndef CL_DET
#define CL_DET 0             // 1: the deterministic compilation (build.py: elbo_mlp_det) -- the epilogue's float atomics (dz_f, image scales) and
#endif                       // the NLL's fp64 atomic become per-observation / per-workgroup STORES that cl_det_reduce sums in a fixed order
#define CL_TILE CL_MLP_TILE
#define CL_NW 8              // waves per workgroup
#define CL_WOBS 16           // observations per wave in forward / dgrad
#define CL_PB 136            // pitch of the [feature][obs] staging tiles (128 + 8): conflict-free ds_read_b128 in wgrad;
                             // the 2-way conflict it leaves on the ds_write_b32 staging writes costs no LDS cycles
#define CL_SCR 32            // floats of per-wave scratch (the dO tile)
// compiler-level fence for memory operations: keeps hipcc from hoisting a whole layer of LDS operand reads
// above the MFMAs that consume them
#define CL_PIN() asm volatile("" ::: "memory")
// Full scheduling fence.  Used to pin software-prefetched LDS operand reads ABOVE the MFMA group that runs while they are in
// flight: a wave issues in order and an MFMA issue blocks until the matrix pipe accepts it, so reads placed after a group of
// MFMAs only start when that group has drained; hipcc by itself keeps one operand buffer and emits exactly that order.
#define CL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Diagnostic build only (-DCL_STAMPS, scripts/stamps.py): per-wave cycle shares of the phases of a tile.  The shipped
// library is built without it and executes no stamp.
#ifdef CL_STAMPS
#define CL_NPH 16
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
        st_acc[k] += t_ - st_last;                                                                 \
        st_last = t_;                                                                              \
    } while (0)
#define STAMP_VM(k)                                                                                \
    do {                                                                                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
        STAMP(k);                                                                                  \
    } while (0)
#else
#define STAMP(k)
#define STAMP_VM(k)
#endif

namespace {

// LDS-only workgroup barrier: do not drain outstanding global atomics / loads (vmcnt) at every layer seam.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// ordering of one wave's own LDS traffic (cross-lane exchange through LDS inside a wave)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// (wave-uniform base pointer) + (32-bit per-lane BYTE offset): the form hipcc lowers to `global_* v, v_off, s[base:base+1]`
// with no 64-bit per-lane address arithmetic (and nothing to keep live or spill across the tile loop)
template <class T>
__device__ __forceinline__ T ld_uo(const T* base, unsigned byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ T* ptr_uo(T* base, unsigned byte_off) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}
// make a wave-uniform int opaque to loop-strength reduction (keeps per-tile base pointers in SGPRs, recomputed per tile)
__device__ __forceinline__ int opaque_uniform(int v) {
    v = __builtin_amdgcn_readfirstlane(v);
    asm volatile("" : "+s"(v));
    return v;
}

// LeakyReLU as max(x, leak x) with a bare v_max_f32: fmaxf() makes hipcc canonicalise x first (a second v_max per element) -- unless
// the unit is compiled with -fno-honor-nans (build.py), which it is since round 6.  Before that the bare instruction was inline assembly:
// opaque to hipcc's hazard recognizer, while its result is an MFMA operand of the next layer and gfx950 wants two wait states between a
// vector-ALU write and an MFMA reading it (NOTEBOOK R6.1; scripts/check_lane_isa.py holds the library to the rule).
__device__ __forceinline__ float lrelu(float x, float leak) {
    const float m = leak * x;
#ifdef CL_LRELU_ASM
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
    return r;
#else
    return __builtin_fmaxf(x, m);
#endif
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int WP, int DP, int LMAX>
struct SmemLayout {
    static constexpr int PW = WP + 4;
    static constexpr int PW1 = DP + 4;
    static constexpr int HR = (WP > DP ? WP : DP);
    static constexpr int oW1 = 0;
    static constexpr int oW = oW1 + WP * PW1;
    static constexpr int oWo = oW + (LMAX - 1) * WP * PW;
    static constexpr int oB = oWo + 2 * WP;
    static constexpr int oBo = oB + LMAX * WP;
    static constexpr int oZ = oBo + 4;
    static constexpr int oH = oZ + WP * CL_PB;
    static constexpr int oS = oH + HR * CL_PB;
    static constexpr int total = oS + CL_NW * CL_SCR;
};

// which 16x16 blocks of dW^T (OB x IB blocks) a wave accumulates, and over which part of the 128-observation tile
template <int OB, int IB>
struct WgradPlan {
    static constexpr int NBK = OB * IB;
    static constexpr int BPW = (NBK >= CL_NW) ? NBK / CL_NW : 1;      // blocks per wave (same o-block, consecutive i-blocks)
    static constexpr int KPARTS = (NBK >= CL_NW) ? 1 : CL_NW / NBK;   // split of the observation axis
    static constexpr int GROUPS = NBK / BPW;                          // distinct block groups
    static constexpr int KLEN = CL_TILE / KPARTS;
    static_assert(IB % BPW == 0, "blocks of one wave must share their o-block");
    static_assert(BPW <= 2, "at most two accumulator blocks per wave and layer");
};

// Width <= 15 (the WP = 16 instance): the eight waves of a workgroup all accumulate the SAME 16x16 block of every layer's dW^T (each over its own 16
// observations), so the accumulators alone are 4 registers x LMAX per wave, next to 4 x LMAX registers of activations: the
// 20-layer instance (the CLI default) spilled ~80 registers to scratch and was bound by that traffic.  The accumulators of the
// top NACC layers therefore live in LDS instead: one private 1-KiB slot per (layer, wave), lane-major float4, read before and
// written after that layer's four wgrad MFMAs.  NACC is whatever the LDS left over by the weight images and staging tiles holds.
template <int WP, int DP, int LMAX, int MODE, bool ILAY>
struct AccPlan {
    using SL = SmemLayout<WP, DP, LMAX>;
    static constexpr int FLUSH = ((WP * DP + WP + (LMAX - 1) * (WP * WP + WP) + 2 * WP + 2 + 3) & ~3) + 4 +
                                 CL_NW * (LMAX * WP + 2 * WP + 2);                         // floats the gradient flush may touch
    static constexpr int OFF = ((SL::total > FLUSH ? SL::total : FLUSH) + 3) & ~3;           // start of the accumulator slots
    static constexpr int SLOT = CL_NW * 256;                                                // floats per layer
    static constexpr int ROOM = (160 * 1024 / 4 - OFF) / SLOT;
    static constexpr bool ON = (WP <= CL_ACC_WP) && (MODE != 1) && (LMAX > 5);
    static constexpr int NACC = ON ? (ROOM < LMAX - 1 ? (ROOM > 0 ? ROOM : 0) : LMAX - 1) : 0;
    static constexpr int LREG = LMAX - NACC;                                                // layers < LREG keep register accumulators
    static constexpr int total = OFF + NACC * SLOT;
};

}  // namespace

// MODE 0: full ELBO step (mono likelihood in the epilogue);  MODE 1: forward only (loc, sigma per observation);
// MODE 2: forward recompute + backward from an externally supplied dL/d(loc, sigma) (Laue two-pass path).
// IMGL: NeuralImageScaler (careless/models/scaling/image.py:66-125): the Dense layers are followed by A.n_imgl layers whose
// (w x w) kernel and bias belong to the IMAGE of the observation.  The observation axis is packed so that a tile holds one
// image (A.tile_img); a workgroup walks a CONTIGUOUS range of tiles, keeps the current image's matrices in the LDS slots
// of layers [A.L, A.L + n_imgl) and its weight-gradient sums in the same register accumulators as any other layer, and
// swaps both (atomicAdd of the sums into the image's gradient, reload) only when the image changes.
// CHAIN: layer-block chaining for scalers deeper than LMAX (third compilation of this file): a block may have no Dense(2) head
// (forward writes its last activations, A.act_out; backward starts from their gradient, A.dH_ext) and may return the gradient
// w.r.t. its input (A.dX_out: the first layer gets a dgrad too).
// ILAY: the packed unit is compiled with (ILAY) and without the per-image-layer code: single-pass Laue only needs the packed layout
// KS: narrow kernel only -- number of 4-feature MFMA steps the hidden width needs (2, 3 or 4; see KPERM); 5 = hidden width EXACTLY 16
//     (round 5): four steps and NO constant-one feature -- slot 15 is a real feature, the bias gradient is the row sum of dZ as in the
//     wider instances (`BONE` off).  Width 16 used to pad to the 32-wide instance: (16 / 32)^2 of its MFMAs useful, slower than width 15
//     in absolute time and no faster than width 32 (profiles/r5_envelope_widths_16_64.txt).
// DET: names the deterministic compilation's instances (their bodies differ by preprocessor: without a template argument of their own
// they would be the same symbols as the plain unit's and the linker would keep one of the two)
template <int WP, int DP, int LMAX, int MODE, bool IMGL, bool CHAIN = false, bool ILAY = IMGL, int KS = 4, bool DET = (CL_DET != 0)>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void elbo_mlp_kernel(const cl_mlp_args A) {
    using SL = SmemLayout<WP, DP, LMAX>;
    constexpr int FB = WP / 16;          // 16-feature blocks of a hidden layer
    constexpr int KS1 = DP / 4;          // MFMA k-steps of the first layer (4 metadata features per step)
    constexpr int IB1 = (DP + 15) / 16;  // 16-feature blocks of the metadata
    constexpr int PW = SL::PW, PW1 = SL::PW1, PB = CL_PB;
    // Width <= 15: a layer has ONE 16x16 weight-gradient block and the plan below splits its 128-observation contraction into
    // eight parts of 16 -- wave k contracts exactly the 16 columns of the staging tiles that wave k wrote itself.  The waves
    // then never read each other's LDS data and the workgroup barriers of the backward pass reduce to wave-local ordering
    // (layer 0 too when the metadata fits one block).  This is the geometry of the careless CLI default (20 layers x width d).
    constexpr bool WLOC = (FB == 1);                 // hidden layers are wave-local
    constexpr bool WLOC0 = WLOC && (IB1 == 1);       // ... and so is the first layer
    static_assert(!WLOC || (WgradPlan<1, 1>::KPARTS == CL_NW && WgradPlan<1, 1>::GROUPS == 1), "wave-local wgrad plan");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sW1 = smem + SL::oW1;
    float* const sW = smem + SL::oW;
    float* const sWo = smem + SL::oWo;
    float* const sB = smem + SL::oB;
    float* const sBo = smem + SL::oBo;
    float* const sZ = smem + SL::oZ;
    float* const sH = smem + SL::oH;

    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;   // a previous step hit a non-finite gradient norm

    // Narrow kernel (w <= 15): padded feature 15 of every hidden activation is held at 1.0 (bias image 1.0 on a zero weight row, and
    // LeakyReLU(1) = 1), and nothing reads it: the next layer's weight column 15 is zero padding.  Column 15 of a layer's dW^T
    // accumulator is then sum_obs dZ[o][obs] * 1 = the BIAS gradient, for free inside the wgrad MFMAs: no LMAX bias registers, no
    // re-read of the dZ tile (a third of this kernel's LDS traffic), no row sums.  Layer 0 (metadata input) keeps its own sum.
    constexpr bool BONE = WLOC && (KS != 5);
    // Narrow kernel: hidden feature f lives in accumulator row ("slot") 4 (f & 3) + (f >> 2) instead of row f (an involution; every
    // LDS image and the gradient flush use it consistently).  MFMA step t of a layer then contracts the features 4t .. 4t+3 rather
    // than {t, 4+t, 8+t, 12+t}, so for a width-w layer only ceil(w / 4) of the four forward and dgrad steps have anything to
    // multiply: the CLI default (w = 10) runs 3 + 3 + 4 MFMAs per layer instead of 12.  Slot 15 stays feature 15 (the bias ones).
    constexpr bool KPERM = WLOC;
    auto sl = [](int f) { return (KPERM && f < 16) ? (((f & 3) << 2) | (f >> 2)) : f; };
    using AP = AccPlan<WP, DP, LMAX, MODE, ILAY>;
    constexpr int LREG = AP::LREG;       // layers >= LREG accumulate their weight gradient in LDS (narrow kernel only)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15;            // observation within the wave / MFMA row-or-column index
    const int q = lane >> 4;            // k-group of the MFMA step
    float* const sS = smem + SL::oS + wv * CL_SCR;      // per-wave dO tile: (dL/dloc, dL/draw) of the wave's 16 observations
    // this lane's float4 of the LDS-resident dW^T accumulator of layer l (l >= LREG)
    f32x4* const sAcc = reinterpret_cast<f32x4*>(smem + AP::OFF) + wv * 64 + lane;
    auto acc_slot = [&](int l) -> f32x4& { return sAcc[(l >= LREG ? l - LREG : 0) * (CL_NW * 64)]; };

    const int d = A.d, w = A.w;
    const int Ld = A.L;                              // Dense layers (parameters in A.mlp)
    const int L = ILAY ? A.L + A.n_imgl : A.L;       // all hidden layers (Dense + per-image)
    const float leak = A.leak;
    const bool no_head = CHAIN && ((MODE == 1 && A.act_out != nullptr) || (MODE == 2 && A.dH_ext != nullptr));

    // ---- stage the weights (global W^T layout, see cl_kernels.h) into padded LDS images, zero-filled ---------
    {
        const float* __restrict__ P = A.mlp;
        for (int base = 0; base < WP * PW1; base += 8 * 512) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 512 + tid;
                const int o = idx / PW1, i = idx - o * PW1;
                v[u] = (idx < WP * PW1 && sl(o) < w && i < d) ? P[sl(o) * d + i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 512 + tid;
                if (idx < WP * PW1) sW1[idx] = v[u];
            }
        }
        for (int idx = tid; idx < LMAX * WP; idx += 512) {
            const int l = idx / WP, o = idx - l * WP;
            float v = 0.0f;
            if (l < Ld && sl(o) < w) v = (l == 0) ? P[w * d + sl(o)] : P[w * d + w + (l - 1) * (w * w + w) + w * w + sl(o)];
            if (BONE && o == 15) v = 1.0f;
            sB[idx] = v;
        }
        // eight independent loads in flight per thread (a plain loop waits for every load before issuing the next one, ~40
        // serial round trips per launch: a fixed cost that matters once a GPU's shard is small)
        for (int l = 1; l < Ld; ++l) {
            const float* __restrict__ Wl = P + w * d + w + (l - 1) * (w * w + w);
            float* dst = sW + (l - 1) * WP * PW;
            for (int base = 0; base < WP * PW; base += 8 * 512) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 512 + tid;
                    const int o = idx / PW, i = idx - o * PW;
                    v[u] = (idx < WP * PW && sl(o) < w && i < WP && sl(i) < w) ? Wl[sl(o) * w + sl(i)] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 512 + tid;
                    if (idx < WP * PW) dst[idx] = v[u];
                }
            }
        }
        const float* __restrict__ Wo = P + w * d + w + (Ld - 1) * (w * w + w);
        for (int idx = tid; idx < 2 * WP; idx += 512) {
            const int c = idx / WP, i = idx - c * WP;
            sWo[idx] = (sl(i) < w && !no_head) ? Wo[c * w + sl(i)] : 0.0f;
        }
        if (tid < 4) sBo[tid] = (tid < 2 && !no_head) ? Wo[2 * w + tid] : 0.0f;
    }
    __syncthreads();

    // ---- accumulators that live across all tiles of this workgroup -----------------------------------------
    constexpr int WB = (FB >= 4) ? 2 : 1;     // 16x16 blocks of dW^T a wave accumulates per layer (WgradPlan::BPW)
    f32x4 wacc[LMAX][WB];
    float bacc[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        bacc[l] = 0.0f;
#pragma unroll
        for (int b = 0; b < WB; ++b)
#pragma unroll
            for (int t = 0; t < 4; ++t) wacc[l][b][t] = 0.0f;
    }
    if (AP::NACC > 0) {
#pragma unroll
        for (int l = LREG; l < LMAX; ++l) acc_slot(l) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    float woacc0 = 0.0f, woacc1 = 0.0f, boacc0 = 0.0f, boacc1 = 0.0f;
    float nll_acc = 0.0f;
    // Evans-2011 error model: Sd* = softplus(raw); per-lane gradient accumulators
    cl_ev11 ev = {1.0f, 0.0f, 0.0f};
    float ev_g0 = 0.0f, ev_g1 = 0.0f, ev_g2 = 0.0f;
    const bool use_ev11 = (MODE == 0) && (A.ev11 != nullptr);
    if (use_ev11) { ev.sdfac = cl_softplus(A.ev11[0]); ev.sdadd = cl_softplus(A.ev11[1]); ev.sdb = cl_softplus(A.ev11[2]); }

    const int ntiles = A.n_pad / CL_TILE;
    const int S = A.S;
    // MFMA steps of a hidden layer's forward / dgrad that hold real features: a template parameter, because a run-time test around
    // single MFMAs costs more than the skipped ones save (-8 % against +10 %)
    constexpr int ks = KPERM ? (KS == 5 ? 4 : KS) : 4;     // MFMA steps of a hidden layer's forward / dgrad that hold real features
#ifdef CL_STAMPS
    unsigned long long st_acc[CL_NPH];
#pragma unroll
    for (int k = 0; k < CL_NPH; ++k) st_acc[k] = 0;
    unsigned long long st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif

    // static priority for the younger half (the second wave of every SIMD loses VALU arbitration otherwise)
    if (wv >= CL_NW / 2) __builtin_amdgcn_s_setprio(1);

    // per-observation inputs of a tile, loaded one tile ahead so their HBM latency hides under the backward pass
    float h0n[KS1];
    int ridn = -1, imgn = 0;
    float ion = 0.0f, sgn = 1.0f;
    // epilogue lane map: lane -> (observation je = lane>>2, sample slot qe = lane&3): the MC samples of one observation sit
    // in adjacent lanes, so z_f / eta / ipred accesses and the dz_f atomics of an observation coalesce into one request
    const int je = lane >> 2, qe = lane & 3;
    // All global accesses below are written as (wave-uniform pointer)[32-bit lane offset] so hipcc emits the
    // SGPR-base + VGPR-offset addressing form.  (With 64-bit per-lane addresses it kept one address pair per load live
    // across the tile, spilled them, and every reload cost an s_waitcnt vmcnt(0) behind the epilogue's atomics.)
    // Streaming inputs go through buffer loads: SGPR descriptor + 32-bit VGPR byte offset + SGPR tile offset.  No 64-bit
    // per-lane addresses exist (hipcc otherwise keeps one address pair per load live across the tile loop, spills them, and
    // every reload costs an s_waitcnt vmcnt(0) behind the epilogue's atomics), and the hardware range check returns 0 for
    // the padded metadata rows (feature >= d) and for observations past n_obs.
    const unsigned lane_obs = CL_WOBS * wv + j;                 // observation of this lane inside the tile (MFMA map)
    const unsigned lane_obs_e = CL_WOBS * wv + je;               // ... in the epilogue lane map
    const unsigned lane_obs_eb = 4u * lane_obs_e;
    const unsigned n_pad_u = (unsigned)A.n_pad;
    const int d4 = (d + 3) & ~3;                                // meta_t has d4 rows (rows >= d are zero): cl_mlp_meta_rows()
    const __amdgpu_buffer_rsrc_t r_meta = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A.meta_t), 0, (int)(4u * (unsigned)d4 * n_pad_u), 0x00020000);
    const unsigned lane_meta_b = 4u * ((unsigned)q * n_pad_u + lane_obs);   // the ONE per-lane byte offset of all metadata loads
    const int row4_b = 16 * A.n_pad;                            // bytes between feature groups (4 rows)
    auto load_meta = [&](int tile, float (&dst)[KS1]) {
        const int soff = tile * (CL_TILE * 4);
        const int d4m = opaque_uniform(d4);
#pragma unroll
        for (int t = 0; t < KS1; ++t) {
            // the feature-group offset goes into the per-lane offset, which the hardware range-checks against the 4 d4 n_pad bytes of
            // the descriptor: groups past the padded metadata rows return 0 without a branch (a `4 t < d4` test per load costs a
            // spilled SGPR mask and a descriptor reload each -- ~100 v_readlane per tile at the end of the backward pass)
            if (KS1 <= CL_META_RANGE) {      // (d <= 8 instance; with eight loads the unconditional form costs registers: -9 % on cfg3)
                dst[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_meta, (int)(lane_meta_b + (unsigned)(t * row4_b)), soff, 0));
            } else {
                dst[t] = 0.0f;
                if (4 * t < d4m)                                // wave-uniform
                    dst[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_meta, (int)lane_meta_b, soff + t * row4_b, 0));
            }
        }
    };
    const int nb_obs = 4 * A.n_obs;
    const __amdgpu_buffer_rsrc_t r_rid = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(A.refl_id), 0, MODE == 0 ? nb_obs : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_io = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.iobs), 0, MODE == 0 ? nb_obs : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_sg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.sig), 0, MODE == 0 ? nb_obs : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(A.image_id), 0, (MODE == 0 && A.use_img) ? nb_obs : 0, 0x00020000);
    auto prefetch = [&](int tile_in) {
        const int tile = __builtin_amdgcn_readfirstlane(tile_in);
        const int soff = tile * (CL_TILE * 4);                  // byte offset of the tile inside a per-observation array
        load_meta(tile, h0n);
        if (MODE == 0) {
            const bool ok = tile * CL_TILE + (int)lane_obs_e < A.n_obs;
            const int rr = (int)__builtin_amdgcn_raw_buffer_load_b32(r_rid, (int)lane_obs_eb, soff, 0);
            ridn = ok ? rr : -1;
            ion = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_io, (int)lane_obs_eb, soff, 0));
            const float ss = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_sg, (int)lane_obs_eb, soff, 0));
            sgn = ok ? ss : 1.0f;
            imgn = (int)__builtin_amdgcn_raw_buffer_load_b32(r_img, (int)lane_obs_eb, soff, 0);
        }
    };
    const int pf_layer = (L > 1) ? 1 : 0;
    // tiles of this workgroup: strided over the grid, or (IMGL) one contiguous range so that image changes are rare
    const int tile_begin = IMGL ? (int)((long long)blockIdx.x * ntiles / (int)gridDim.x) : (int)blockIdx.x;
    const int tile_end = IMGL ? (int)((long long)(blockIdx.x + 1) * ntiles / (int)gridDim.x) : ntiles;
    const int tile_step = IMGL ? 1 : (int)gridDim.x;
    if (tile_begin < tile_end) prefetch(tile_begin);

    // ---- per-image layers: gradient flush and weight reload on an image change ---------------------------------
    int cur_img = -1;
    const size_t imgl_blk = ILAY ? (size_t)A.n_images * (size_t)(w * w + w) : 0;      // floats per image layer
    auto imgl_flush = [&](int im) {
#pragma unroll
        for (int l = 1; l < LMAX; ++l) {
            if (l >= Ld && l < L) {
                using WG = WgradPlan<FB, FB>;
                float* __restrict__ gW = A.d_imgl + (size_t)(l - Ld) * imgl_blk + (size_t)im * (size_t)(w * w);
                float* __restrict__ gB = A.d_imgl + (size_t)(l - Ld) * imgl_blk + (size_t)A.n_images * (size_t)(w * w) + (size_t)im * w;
                const int grp = wv % WG::GROUPS;
                const int ob = (grp * WG::BPW) / FB, ib0 = (grp * WG::BPW) - ob * FB;
#pragma unroll
                for (int b = 0; b < WB; ++b) {
                    if (b < WG::BPW) {
                        const int i = 16 * (ib0 + b) + j;
                        const bool in_lds = (AP::NACC > 0) && (l >= LREG);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int o = 16 * ob + 4 * q + t;
                            const float g = in_lds ? acc_slot(l)[t] : wacc[l < LREG ? l : 0][b][t];
                            if (sl(o) < w && sl(i) < w) atomicAdd(gW + sl(o) * w + sl(i), g);
                            if (BONE && sl(o) < w && i == 15) atomicAdd(gB + sl(o), g);
                            if (!in_lds) wacc[l < LREG ? l : 0][b][t] = 0.0f;
                        }
                        if (in_lds) acc_slot(l) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
                }
                if (!BONE && lane < w) atomicAdd(gB + lane, bacc[l]);
                bacc[l] = 0.0f;
            }
        }
    };
    auto imgl_load = [&](int im) {
        for (int l = Ld; l < L; ++l) {
            const float* __restrict__ Wg = A.imgl + (size_t)(l - Ld) * imgl_blk + (size_t)im * (size_t)(w * w);
            const float* __restrict__ Bg = A.imgl + (size_t)(l - Ld) * imgl_blk + (size_t)A.n_images * (size_t)(w * w) + (size_t)im * w;
            float* dst = sW + (l - 1) * WP * PW;
            for (int idx = tid; idx < WP * PW; idx += 512) {
                const int o = idx / PW, i = idx - o * PW;
                dst[idx] = (sl(o) < w && i < WP && sl(i) < w) ? Wg[sl(o) * w + sl(i)] : 0.0f;
            }
            if (tid < WP) sB[l * WP + tid] = (sl(tid) < w) ? Bg[sl(tid)] : ((BONE && tid == 15) ? 1.0f : 0.0f);
        }
    };

    for (int tile = tile_begin; tile < tile_end; tile += tile_step) {
        // the layer count, made opaque once per tile: otherwise hipcc hoists every `l == L - 1` / `l < L` test of the unrolled layer
        // loops out of the tile loop, runs out of SGPRs, parks the masks in VGPR lanes and pays v_readlane / v_writelane round trips
        // (plus bool -> VGPR -> bool conversions) inside every layer
        const int Lt = opaque_uniform(L);
        const int gobs = tile * CL_TILE + CL_WOBS * wv + j;      // this lane's observation (all four k-groups)
        if (ILAY && A.n_imgl > 0) {
            const int im = __builtin_amdgcn_readfirstlane(A.tile_img[tile]);
            if (im != cur_img) {             // workgroup-uniform
                lds_barrier();               // every wave is done with the previous image's matrices
                if (MODE != 1 && cur_img >= 0) imgl_flush(cur_img);
                imgl_load(im);
                cur_img = im;
                lds_barrier();
            }
        }

        // ================= forward =======================================================================
        float h0[KS1];                     // metadata^T as B operand: step t holds feature 4t + q
#pragma unroll
        for (int t = 0; t < KS1; ++t) h0[t] = h0n[t];
        const int rid = ridn, img = imgn;
        const float io = ion, sg = sgn;
        // gathers that depend on the prefetched ids: issued now, consumed in the epilogue (latency hides under forward)
        float aim = 1.0f, zf0 = 0.0f, zf1 = 0.0f, et0 = 0.0f, et1 = 0.0f;
        cl_args_p E0 = kernargs_again();     // arguments of the gathers below, re-read here (see kernargs_again, cl_kernels.h)
        const int gobs_e = tile * CL_TILE + CL_WOBS * wv + je;   // the observation this lane handles in the epilogue
        const unsigned zoff = 4u * (unsigned)rid * (unsigned)S;     // z_f / dz_f BYTE offset of this lane's reflection
        const unsigned eoff_t = 4u * lane_obs_e * (unsigned)S;      // eta / ipred BYTE offset inside the tile
        const int tile_u = opaque_uniform(tile);
        // IMGL: the packed row -> the caller's row (eta / ipred_out / noise key are in the caller's order)
        int rme = 0;
        long long nkey = 0;                // noise key of this lane's row
        if (IMGL && MODE == 0 && rid >= 0) {
            rme = E0->row_map[gobs_e];
            nkey = (E0->noise_row != nullptr) ? (long long)E0->noise_row[gobs_e] : E0->obs_offset + rme;
        }
        const unsigned eoff = IMGL ? 4u * (unsigned)rme * (unsigned)S : eoff_t;
#if CL_DET
        // deterministic mode: byte offset of this observation's record in dzf_obs -- its own row, or the slot the caller assigns it
        // (det_slot: reflection order, so that cl_det_reduce reads contiguously)
        const int drow = IMGL ? rme : gobs_e;                      // (packed layouts: the caller's row, as for eta / ipred_out)
        unsigned dzo = 4u * (unsigned)drow * (unsigned)S;
        if (MODE == 0 && rid >= 0 && E0->det_slot != nullptr) dzo = 4u * (unsigned)E0->det_slot[drow] * (unsigned)S;
#endif
        const float* __restrict__ eta_t = E0->eta ? (IMGL ? E0->eta : E0->eta + (size_t)tile_u * CL_TILE * S) : nullptr;
        float* __restrict__ ipred_t = E0->ipred_out ? (IMGL ? E0->ipred_out : E0->ipred_out + (size_t)tile_u * CL_TILE * S) : nullptr;
        if (MODE == 0 && rid >= 0) {
            if (E0->use_img && img > 0) aim = ld_uo(E0->img, 4u * (unsigned)(img - 1));
            if (qe < S) zf0 = ld_uo(E0->z_f, zoff + 4u * qe);
            if (qe + 4 < S) zf1 = ld_uo(E0->z_f, zoff + 4u * (qe + 4));
            if (eta_t != nullptr) {
                if (qe < S) et0 = ld_uo(eta_t, eoff + 4u * qe);
                if (qe + 4 < S) et1 = ld_uo(eta_t, eoff + 4u * (qe + 4));
            } else if (!IMGL && E0->noise_row != nullptr) {
                // plain layout over rows that are not a contiguous range of the caller's (a rank that owns a reflection range takes
                // every observation of those reflections): the row's GLOBAL number, the key of the in-kernel noise, rides in the
                // register the injected noise would use
                et0 = __builtin_bit_cast(float, ld_uo(E0->noise_row, 4u * (unsigned)gobs_e));
            }
        }
        STAMP_VM(0);
        f32x4 hs[LMAX][FB];                // post-activation H_l^T: block fb, reg t = feature 16fb + 4q + t, obs j
        float o0 = 0.0f, o1 = 0.0f;
        // Width <= 15: a layer is ONE chain of four dependent MFMAs, so there is no k-block loop to prefetch inside; the weight
        // operand and the bias of the NEXT layer are requested before this layer's MFMAs instead
        constexpr bool PFN = (FB == 1);
        f32x4 pfw = {0.0f, 0.0f, 0.0f, 0.0f}, pfb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < Lt) {
                // two output blocks at a time (when the layer has them): two independent accumulator chains keep the MFMA
                // pipe at its issue rate (one 16x16x4 chain is paced by the 40-cycle dependent latency, not the 32-cycle issue)
                constexpr int MBS = (FB >= 2) ? 2 : 1;
#pragma unroll
                for (int mb = 0; mb < FB; mb += MBS) {
                    constexpr int MB1 = MBS - 1;           // offset of the second block of the pair (0: no second block)
                    f32x4 acc0 = (PFN && l > 0) ? pfb : *reinterpret_cast<const f32x4*>(sB + l * WP + 16 * mb + 4 * q);         // bias
                    f32x4 acc1 = *reinterpret_cast<const f32x4*>(sB + l * WP + 16 * (mb + MB1) + 4 * q);
                    f32x4 nxw = pfw, nxb = pfb;
                    if (PFN && l + 1 < LMAX) {
                        nxw = *reinterpret_cast<const f32x4*>(sW + l * WP * PW + j * PW + 4 * q);
                        nxb = *reinterpret_cast<const f32x4*>(sB + (l + 1) * WP + 4 * q);
                    }
                    if (l == 0) {
#pragma unroll
                        for (int t = 0; t < KS1; ++t) {
                            acc0 = mfma4(sW1[(16 * mb + j) * PW1 + 4 * t + q], h0[t], acc0);
                            if (MBS == 2) acc1 = mfma4(sW1[(16 * (mb + MB1) + j) * PW1 + 4 * t + q], h0[t], acc1);
                        }
                    } else {
                        const float* Wl = sW + (l > 0 ? l - 1 : 0) * WP * PW;
                        const float* pa0 = Wl + (16 * mb + j) * PW + 4 * q;
                        const float* pa1 = Wl + (16 * (mb + MB1) + j) * PW + 4 * q;
                        // two operand buffers used alternately (compile-time index after unrolling: no register copies)
                        f32x4 oa[2], ob[2];
                        oa[0] = PFN ? pfw : *reinterpret_cast<const f32x4*>(pa0);
                        ob[0] = *reinterpret_cast<const f32x4*>(pa1);
#pragma unroll
                        for (int kb = 0; kb < FB; ++kb) {
                            if (kb + 1 < FB) {       // operands of the NEXT k-block, in flight under this block's MFMAs
                                oa[(kb + 1) & 1] = *reinterpret_cast<const f32x4*>(pa0 + 16 * (kb + 1));
                                ob[(kb + 1) & 1] = *reinterpret_cast<const f32x4*>(pa1 + 16 * (kb + 1));
                            }
                            CL_SCHED_FENCE();
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                if (!KPERM || t < ks) acc0 = mfma4(oa[kb & 1][t], hs[l > 0 ? l - 1 : 0][kb][t], acc0);
                                if (MBS == 2) acc1 = mfma4(ob[kb & 1][t], hs[l > 0 ? l - 1 : 0][kb][t], acc1);
                            }
                            CL_SCHED_FENCE();
                        }
                    }
                    CL_PIN();
                    if (PFN) { pfw = nxw; pfb = nxb; }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        hs[l][mb][t] = lrelu(acc0[t], leak);
                        if (MBS == 2) hs[l][mb + MB1][t] = lrelu(acc1[t], leak);
                    }
                }
                if (l == Lt - 1) {
                    // final Dense(2): every k-group holds a quarter of the features of observation j
#pragma unroll
                    for (int mb = 0; mb < FB; ++mb) {
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sWo + 16 * mb + 4 * q);
                        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sWo + WP + 16 * mb + 4 * q);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            o0 = fmaf(w0[t], hs[l][mb][t], o0);
                            o1 = fmaf(w1[t], hs[l][mb][t], o1);
                        }
                    }
                }
            }
        }
        o0 += __shfl_xor(o0, 16);
        o1 += __shfl_xor(o1, 16);
        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o0 += sBo[0];
        o1 += sBo[1];

        STAMP(1);
        const bool valid = gobs < A.n_obs;
        float dsig_draw;
        const float sigma = cl_scale_bij(o1, A.bij_kind, A.eps, &dsig_draw);

        if (MODE == 1 && no_head) {
            // block of a chain: the last layer's activations, feature-major like the metadata (they ARE the next block's metadata)
            if (valid) {
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
                    if (l == Lt - 1) {
#pragma unroll
                        for (int mb = 0; mb < FB; ++mb)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const int f = sl(16 * mb + 4 * q + t);
                                if (f < ((w + 3) & ~3)) A.act_out[(size_t)f * A.n_pad + gobs] = (BONE && f >= w) ? 0.0f : hs[l][mb][t];
                            }
                    }
            }
            if (tile + tile_step < tile_end) prefetch(tile + tile_step);
            continue;
        }
        if (MODE == 1) {
            if (q == 0 && valid) {
                const int row = IMGL ? A.row_map[gobs] : gobs;
                if (row >= 0) {
                    A.loc_out[row] = o0;
                    A.sig_out[row] = sigma;
                }
            }
            if (tile + tile_step < tile_end) prefetch(tile + tile_step);
            continue;
        }

        // ================= epilogue: sample, predict, likelihood, dL/dO =====================================
        cl_args_p E = kernargs_again();      // the epilogue's arguments are loaded here, once per tile, not held in SGPRs across the tile
        float dloc, draw;
        if (MODE == 0) {
            // move the scaler outputs from the MFMA lane map (observation lane&15) to the epilogue lane map (lane>>2)
            const float o0e = __shfl(o0, je);
            const float sige = __shfl(sigma, je);
            float pdl = 0.0f, pds = 0.0f, pda = 0.0f;
            STAMP(11);
            if (IMGL && E->gmeta != nullptr) {
                // ---- single-pass Laue: the predictions of the rows of one harmonic group SUM before the likelihood
                //      (ConvolvedLikelihood.convolve, careless/models/likelihoods/laue.py:20-34).  The members of a group sit in
                //      consecutive rows of this wave; lane (row je, slot qe) collects the group's total with shuffles, every
                //      member evaluates the same likelihood derivative, member 0 alone counts the group's log-likelihood.
                const int gm = (rid >= 0) ? E->gmeta[gobs_e] : 0;
                const int mem = gm & 0xff, cnt = gm >> 8;
                const int lfirst = 4 * (je - mem) + qe;
                const int gmax = __builtin_amdgcn_readfirstlane(E->tile_gmax[tile]);
                // hardware reciprocal and logarithm (1 ulp): sigma is an input, its log enters the NLL additively
                const float inv_sg = 1.0f / sg;
                const float log_sg = logf(sg);
                float eta_sin = 0.0f;
                const int K = (S + 3) >> 2;
                for (int k = 0; k < K; ++k) {                     // wave-uniform trip count: all lanes take part in the shuffles
                    const int s = qe + 4 * k;
                    const bool act = (rid >= 0) && (s < S);
                    float eta = 0.0f, zf = 0.0f;
                    if (act) {
                        if (eta_t != nullptr) {
                            eta = (k == 0) ? et0 : ((k == 1) ? et1 : ld_uo(eta_t, eoff + 4u * s));
                        } else if ((k & 1) == 0) {
                            cl_noise_normal_pair(E->seed, E->step, (uint32_t)s, (uint64_t)nkey, &eta, &eta_sin);
                        } else {
                            eta = eta_sin;
                        }
                        zf = (k == 0) ? zf0 : ((k == 1) ? zf1 : ld_uo(E->z_f, zoff + 4u * s));
                    }
                    const float tq = o0e + sige * eta + E->shift;
                    const float zs = aim * tq;
                    const float ipred = act ? zs * zf * zf : 0.0f;
                    if (act && ipred_t) *ptr_uo(ipred_t, eoff + 4u * s) = ipred;
                    float tot = 0.0f;
                    for (int mm = 0; mm < gmax; ++mm) {
                        const float v = __shfl(ipred, (lfirst + 4 * mm) & 63);
                        tot += (mm < cnt) ? v : 0.0f;
                    }
                    if (act) {
                        float dll, ll;
                        if (use_ev11) {
                            float gf, gb, ga;
                            ll = cl_lik_ev11(tot, io, sg, E->lik_kind, E->dof, E->lik_const, ev, &dll, &gf, &gb, &ga);
                            if (mem == 0) { ev_g0 -= gf * E->w_ll; ev_g1 -= ga * E->w_ll; ev_g2 -= gb * E->w_ll; }
                        } else {
                            ll = cl_lik_log_prob2(tot, io, inv_sg, log_sg, E->lik_kind, E->dof, E->lik_const, &dll);
                        }
                        if (mem == 0) nll_acc -= ll * E->w_ll;
                        const float gi = -dll * E->w_ll;                 // dNLL / d iconv = dNLL / d ipred of every member
                        const float dzs = gi * zf * zf;
#if CL_DET
                        *ptr_uo(E->dzf_obs, dzo + 4u * s) = gi * zs * 2.0f * zf;      // summed per reflection, in a fixed order, by cl_det_reduce
#else
                        atomicAdd(ptr_uo(E->dz_f, zoff + 4u * s), gi * zs * 2.0f * zf);
#endif
                        const float dt = dzs * aim;
                        pdl += dt;
                        pds += dt * eta;
                        pda += dzs * tq;
                    }
                }
            } else if (rid >= 0) {
                // hardware reciprocal and logarithm (1 ulp): sigma is an input, its log enters the NLL additively
                const float inv_sg = 1.0f / sg;
                const float log_sg = logf(sg);
                int k = 0;
                float eta_sin = 0.0f;
                for (int s = qe; s < S; s += 4, ++k) {
                    float eta;
                    if (eta_t != nullptr) {
                        eta = (k == 0) ? et0 : ((k == 1) ? et1 : ld_uo(eta_t, eoff + 4u * s));
                    } else if ((k & 1) == 0) {       // one Philox block + Box-Muller pair serves samples s and s + 4
                        const long long key_plain = (E->noise_row != nullptr) ? (long long)__builtin_bit_cast(int, et0) : E->obs_offset + gobs_e;
                        cl_noise_normal_pair(E->seed, E->step, (uint32_t)s, (uint64_t)(IMGL ? nkey : key_plain), &eta, &eta_sin);
                    } else {
                        eta = eta_sin;
                    }
                    const float zf = (k == 0) ? zf0 : ((k == 1) ? zf1 : ld_uo(E->z_f, zoff + 4u * s));
                    const float tq = o0e + sige * eta + E->shift;
                    const float zs = aim * tq;
                    const float ipred = zs * zf * zf;
                    if (ipred_t) *ptr_uo(ipred_t, eoff + 4u * s) = ipred;
                    float dll, ll;
                    if (use_ev11) {
                        float gf, gb, ga;
                        ll = cl_lik_ev11(ipred, io, sg, E->lik_kind, E->dof, E->lik_const, ev, &dll, &gf, &gb, &ga);
                        ev_g0 -= gf * E->w_ll; ev_g1 -= ga * E->w_ll; ev_g2 -= gb * E->w_ll;     // order: Sdfac, Sdadd, SdB
                    } else {
                        ll = cl_lik_log_prob2(ipred, io, inv_sg, log_sg, E->lik_kind, E->dof, E->lik_const, &dll);
                    }
                    nll_acc -= ll * E->w_ll;
                    const float gi = -dll * E->w_ll;                 // dNLL / d ipred
                    const float dzs = gi * zf * zf;
#if CL_DET
                    *ptr_uo(E->dzf_obs, dzo + 4u * s) = gi * zs * 2.0f * zf;      // summed per reflection, in row order, by cl_det_reduce
#else
                    atomicAdd(ptr_uo(E->dz_f, zoff + 4u * s), gi * zs * 2.0f * zf);
#endif
                    const float dt = dzs * aim;
                    pdl += dt;
                    pds += dt * eta;
                    pda += dzs * tq;
                }
            }
            pdl = cl_quad_sum(pdl); pds = cl_quad_sum(pds); pda = cl_quad_sum(pda);      // the four sample slots of an observation
            STAMP(12);
#if CL_DET
            if (E->use_img) {
                if (qe == 0 && rid >= 0) E->dimg_obs[IMGL ? rme : gobs_e] = pda;      // summed per image, in row order, by cl_det_reduce
            }
#else
            if (E->use_img) {
                // image ids are sorted, so the 16 observations of a wave almost always share one image: reduce in the
                // wave and issue ONE atomic instead of 16 same-address ones (which serialise in the L2 atomic unit)
                const int img0 = __builtin_amdgcn_readfirstlane(img);
                if (__all(img == img0 || rid < 0)) {
                    float v = (qe == 0 && rid >= 0) ? pda : 0.0f;
                    v = cl_wave_sum(v);
                    if (lane == 0 && img0 > 0) atomicAdd(ptr_uo(E->d_img, 4u * (unsigned)(img0 - 1)), v);
                } else {
                    cl_image_grad_segments(E->d_img, img, pda, qe == 0 && rid >= 0 && img > 0, lane);
                }
            }
#endif
            // back to the MFMA lane map: lane (j, q) needs dL/dloc and dL/dsigma of observation j, held by lanes 4j..4j+3
            dloc = __shfl(pdl, 4 * j);
            draw = __shfl(pds, 4 * j) * dsig_draw;
            if (q == 0) {
                boacc0 += dloc;
                boacc1 += draw;
            }
            STAMP(13);
        } else if (no_head) {
            dloc = 0.0f; draw = 0.0f;                     // head-less block of a chain: the gradient arrives as dH_ext below
        } else {
            const int row = (IMGL && valid) ? E->row_map[gobs] : gobs;
            const bool have = valid && row >= 0;
            dloc = have ? E->dO_ext[2 * (size_t)row] : 0.0f;
            draw = have ? E->dO_ext[2 * (size_t)row + 1] * dsig_draw : 0.0f;   // external grad is w.r.t. sigma
            if (q == 0) {
                boacc0 += dloc; boacc1 += draw;
            }
        }
        STAMP(14);

        // ================= backward =======================================================================
        // tile seam: every wave must be done reading the staging tiles of the previous tile's last wgrad
        STAMP(2);
        if (WLOC0) wave_lds_sync(); else lds_barrier();
        STAMP(3);
        // the dO tile for the Dense(2) wgrad (wave-private), interleaved: (dL/dloc, dL/draw) of observation j at [2j, 2j+1]
        if (q == 0) *reinterpret_cast<f32x2*>(sS + 2 * j) = f32x2{dloc, draw};

        f32x4 dH[FB];
        // Narrow kernel (WPIPE): a layer step is two chains of four dependent MFMAs (dgrad, wgrad), each behind an LDS round trip,
        // and with two waves per SIMD that latency IS the run time.  The wgrad of layer l is therefore issued one iteration late:
        // its operand reads are requested right after the layer's staging writes (one wave's LDS operations execute in order, and
        // the wave only reads the 16 columns it wrote itself, so no wait is needed in between, nor before the next layer's writes),
        // and its MFMAs run interleaved with the dgrad MFMAs of layer l-1, when the operands have long arrived.  The weight
        // operands of the dgrad are requested one iteration ahead as well.  With no wgrad pending (top layer) the operands are
        // zero and the MFMAs add exactly 0 to an accumulator, so the pipeline needs no branches.
        constexpr bool WPIPE = WLOC;
        f32x4 pa4 = {0.0f, 0.0f, 0.0f, 0.0f}, pb4 = {0.0f, 0.0f, 0.0f, 0.0f}, pacc = {0.0f, 0.0f, 0.0f, 0.0f};
        float r0w[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int l = LMAX - 1; l >= 0; --l) {
            if (l == Lt - 1 && no_head) {
                // head-less block of a chain: dL/dH_L comes from the next block (feature-major, like the metadata)
#pragma unroll
                for (int mb = 0; mb < FB; ++mb)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int f = sl(16 * mb + 4 * q + t);
                        dH[mb][t] = (valid && f < ((w + 3) & ~3)) ? A.dH_ext[(size_t)f * A.n_pad + gobs] : 0.0f;
                    }
            } else if (l == Lt - 1) {
                // dH_L = W_o^T dO ; Dense(2) wgrad from the wave-private columns of the H staging tile
#pragma unroll
                for (int mb = 0; mb < FB; ++mb) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(sWo + 16 * mb + 4 * q);
                    const f32x4 w1 = *reinterpret_cast<const f32x4*>(sWo + WP + 16 * mb + 4 * q);
                    dH[mb] = w0 * dloc + w1 * draw;                      // whole-vector form: packed fp32 math on aligned register pairs
#pragma unroll
                    for (int t = 0; t < 4; ++t) sH[(16 * mb + 4 * q + t) * PB + CL_WOBS * wv + j] = hs[l][mb][t];
                }
                wave_lds_sync();
                if (lane < WP) {
                    // (dWo[0][i], dWo[1][i]) of feature i = lane as ONE packed accumulator: h * (dloc, draw) pairs straight from LDS
                    f32x2 wo = {woacc0, woacc1};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x4 h4 = *reinterpret_cast<const f32x4*>(sH + lane * PB + CL_WOBS * wv + 4 * e);
                        const f32x4 da = *reinterpret_cast<const f32x4*>(sS + 8 * e);          // observations 4e, 4e+1
                        const f32x4 db = *reinterpret_cast<const f32x4*>(sS + 8 * e + 4);      // observations 4e+2, 4e+3
                        wo += f32x2{da[0], da[1]} * h4[0];
                        wo += f32x2{da[2], da[3]} * h4[1];
                        wo += f32x2{db[0], db[1]} * h4[2];
                        wo += f32x2{db[2], db[3]} * h4[3];
                    }
                    woacc0 = wo[0]; woacc1 = wo[1];
                }
                wave_lds_sync();
                STAMP(4);
            }
            if (l < Lt) {
                // next tile's inputs: issued two layers before the end of the backward pass -- early enough to cover the HBM
                // latency, late enough that the registers of the upper layers' activations are free again
                if (l == pf_layer && tile + tile_step < tile_end) prefetch(tile + tile_step);
                // dZ_l = dH_l * lrelu'(H_l)   (sign of the post-activation == sign of the pre-activation).  Only the top layer
                // does it here: for the others it was done one step earlier, in the shadow of the wgrad operand reads (below)
                if (l == Lt - 1) {
#pragma unroll
                    for (int mb = 0; mb < FB; ++mb)
#pragma unroll
                        for (int t = 0; t < 4; ++t) dH[mb][t] = (hs[l][mb][t] > 0.0f) ? dH[mb][t] : leak * dH[mb][t];
                }

                STAMP(5);
                // accumulator of the pending wgrad (layer l+1): registers, or the LDS slot read when its operands were requested
                const int LP = (l + 1 < LMAX) ? l + 1 : LMAX - 1;               // (compile-time after unrolling)
                const bool LP_LDS = (AP::NACC > 0) && (LP >= LREG);
              if (WPIPE && l > 0) {
                const float* wq = sW + (l > 0 ? l - 1 : 0) * WP * PW + (4 * q) * PW + j;
                float (&rw)[4] = r0w[l & 1];                       // weight operands of this layer (two buffers, by layer parity)
                if (l == Lt - 1) {                                 // top layer: nothing was requested ahead
#pragma unroll
                    for (int t = 0; t < 4; ++t) rw[t] = wq[t * PW];
                }
                float* const stz = sZ + (4 * q) * PB + CL_WOBS * wv + j;
                float* const sth = sH + (4 * q) * PB + CL_WOBS * wv + j;
                f32x4 accd = {0.0f, 0.0f, 0.0f, 0.0f};
                f32x4 accw = LP_LDS ? pacc : wacc[LP < LREG ? LP : 0][0];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!KPERM || t < ks) accd = mfma4(rw[t], dH[0][t], accd);         // dgrad of layer l (steps with features < w only)
                    if (l + 1 < LMAX) accw = mfma4(pa4[t], pb4[t], accw);              // wgrad of layer l+1 (operands from last iteration)
                    stz[t * PB] = dH[0][t];                                            // staging: dZ_l and H_{l-1}, this wave's columns
                    sth[t * PB] = hs[l > 0 ? l - 1 : 0][0][t];
                }
                // width 16 (no constant-one feature): layer l+1's bias gradient = row sums of its dZ -- the operand quad this lane just
                // used is four observations of row j; the four lanes of a row add up in the flush
                if (!BONE && l + 1 < LMAX) bacc[LP] += (pa4[0] + pa4[1]) + (pa4[2] + pa4[3]);
                if (l + 1 < LMAX) { if (LP_LDS) acc_slot(LP) = accw; else wacc[LP < LREG ? LP : 0][0] = accw; }
                CL_SCHED_FENCE();
                // requested now, consumed one iteration from now: the operands (and LDS accumulator) of this layer's wgrad and the
                // weight operands of the next dgrad.  Issued AFTER this iteration's MFMAs: the wait in front of those covers every
                // LDS operation in flight (the counter is in order), so anything requested before them would be waited for at once
                pa4 = *reinterpret_cast<const f32x4*>(sZ + j * PB + CL_WOBS * wv + 4 * q);
                pb4 = *reinterpret_cast<const f32x4*>(sH + j * PB + CL_WOBS * wv + 4 * q);
                if ((AP::NACC > 0) && (l >= LREG)) pacc = acc_slot(l);
                if (l >= 2) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) r0w[(l - 1) & 1][t] = (wq - WP * PW)[t * PW];
                }
                CL_SCHED_FENCE();
#pragma unroll
                for (int t = 0; t < 4; ++t) dH[0][t] = (hs[l > 0 ? l - 1 : 0][0][t] > 0.0f) ? accd[t] : leak * accd[t];
                STAMP(9);
              } else {
                if (WPIPE && LMAX > 1) {
                    // layer 0 takes the staged path below; first retire the pending wgrad of layer 1 (zero operands if Lt == 1)
                    f32x4 accw = LP_LDS ? pacc : wacc[LP < LREG ? LP : 0][0];
#pragma unroll
                    for (int t = 0; t < 4; ++t) accw = mfma4(pa4[t], pb4[t], accw);
                    if (LP_LDS) acc_slot(LP) = accw; else wacc[LP < LREG ? LP : 0][0] = accw;
                    if (!BONE) bacc[LP] += (pa4[0] + pa4[1]) + (pa4[2] + pa4[3]);     // (width 16: layer 1's bias gradient, as above)
#pragma unroll
                    for (int t = 0; t < 4; ++t) { pa4[t] = 0.0f; pb4[t] = 0.0f; }
                }
                // barrier A: the previous layer's wgrad reads are complete
                if (l < Lt - 1) { if (WLOC) wave_lds_sync(); else lds_barrier(); }
                STAMP(6);
                // staging writes of this wave's 16 columns: dZ_l into sZ and H_{l-1} into sH.  With a dgrad to run (l > 0) they
                // are issued BETWEEN its MFMAs (both only read registers), so the LDS write phase costs no matrix-pipe time
                float* const stz = sZ + (4 * q) * PB + CL_WOBS * wv + j;
                float* const sth = sH + (4 * q) * PB + CL_WOBS * wv + j;
                if (l == 0) {
#pragma unroll
                    for (int mb = 0; mb < FB; ++mb)
#pragma unroll
                        for (int t = 0; t < 4; ++t) stz[(16 * mb + t) * PB] = dH[mb][t];
                    // the metadata stay in registers through the backward pass: re-reading the tile here (an L2 hit) still puts a
                    // vector-memory round trip in front of the staging writes of every wave at the same time (measured: +2.2 %
                    // on cfg3 without it, at the price of ~20 more spilled registers in the d <= 32 instance)
#pragma unroll
                    for (int t = 0; t < KS1; ++t) sH[(4 * t + q) * PB + CL_WOBS * wv + j] = h0[t];
                    if (CHAIN && A.dX_out != nullptr) {
                        // block of a chain: the gradient w.r.t. this block's input, dX = W_1 dZ_0 (a dgrad for the first layer too),
                        // written feature-major -- it is the dH_ext of the block before
#pragma unroll
                        for (int ib = 0; ib < IB1; ++ib) {
                            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                            for (int kb = 0; kb < FB; ++kb)
#pragma unroll
                                for (int t = 0; t < 4; ++t)
                                    acc = mfma4(sW1[(16 * kb + 4 * q + t) * PW1 + (16 * ib + j < DP ? 16 * ib + j : 0)], dH[kb][t], acc);
                            if (valid) {
#pragma unroll
                                for (int t = 0; t < 4; ++t) {
                                    const int f = 16 * ib + 4 * q + t;
                                    if (f < d4) A.dX_out[(size_t)f * A.n_pad + gobs] = acc[t];
                                }
                            }
                        }
                    }
                }
                // ---- dgrad: dH_{l-1} = W_l dZ_l.  Register + weight-image only, so it runs BEFORE barrier B and
                //      overlaps the other waves' staging writes ---------------------------------------------------
                f32x4 dn[FB];
                if (l > 0) {
                    const float* Wl = sW + (l > 0 ? l - 1 : 0) * WP * PW;
                    constexpr int MBS = (FB >= 2) ? 2 : 1;
                    constexpr int MB1 = MBS - 1;
                    // A operands: W rows 16 kb + 4 q + t, one ds_read_b32 per MFMA.  Those of the next k-block (or of the next
                    // pair of output blocks) are requested before this block's MFMAs are issued (see CL_SCHED_FENCE)
                    const float* wq = Wl + (4 * q) * PW + j;
                    float r0[4], r1[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { r0[t] = wq[t * PW]; r1[t] = wq[t * PW + 16 * MB1]; }
#pragma unroll
                    for (int mb = 0; mb < FB; mb += MBS) {
                        f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int kb = 0; kb < FB; ++kb) {
                            float n0[4], n1[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                n0[t] = r0[t]; n1[t] = r1[t];
                                if (kb + 1 < FB) {
                                    n0[t] = wq[(16 * (kb + 1) + t) * PW + 16 * mb]; n1[t] = wq[(16 * (kb + 1) + t) * PW + 16 * (mb + MB1)];
                                } else if (mb + MBS < FB) {
                                    n0[t] = wq[t * PW + 16 * (mb + MBS)]; n1[t] = wq[t * PW + 16 * (mb + MBS + MB1)];
                                }
                            }
                            CL_SCHED_FENCE();
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                acc0 = mfma4(r0[t], dH[kb][t], acc0);
                                if (MBS == 2) acc1 = mfma4(r1[t], dH[kb][t], acc1);
                            }
                            {   // this group's share of the staging writes, interleaved with its MFMAs
                                constexpr int IPG = 8 * MBS / FB;                  // write items per (pair, k-block) group
                                const int gi = (mb / MBS) * FB + kb;
#pragma unroll
                                for (int e = IPG * gi; e < IPG * (gi + 1); ++e) {
                                    // (plain stores are merged into ds_write2_b32, whose 8-bit offsets cost a v_add_u32 of the base per pair -- a vector
                                    //  instruction beside the MFMAs -- where single ds_write_b32 take the whole offset as an immediate: +0.4 % on the bench line)
                                    // single ds_write_b32 with the whole offset as an immediate (no ds_write2 merge, no re-basing add)
                                    if (e < 4 * FB) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"((unsigned)(size_t)stz), "v"(dH[(e / 4) % FB][e % 4]), "n"(4 * (16 * (e / 4) + (e % 4)) * PB) : "memory");
                                    else asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"((unsigned)(size_t)sth), "v"(hs[l > 0 ? l - 1 : 0][((e - 4 * FB) / 4) % FB][e % 4]), "n"(4 * (16 * ((e - 4 * FB) / 4) + (e % 4)) * PB) : "memory");
                                }
#pragma unroll
                                for (int k = 0; k < IPG; ++k) {
                                    __builtin_amdgcn_sched_group_barrier(0x008, (4 * MBS + IPG - 1) / IPG, 0);   // MFMA
                                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                            // DS write
                                }
                            }
                            CL_SCHED_FENCE();
#pragma unroll
                            for (int t = 0; t < 4; ++t) { r0[t] = n0[t]; r1[t] = n1[t]; }
                        }
                        dn[mb] = acc0;
                        if (MBS == 2) dn[mb + MB1] = acc1;
                    }
                }
                STAMP(7);
                // barrier B: staging tiles complete for all 128 observations
                if (WLOC && (l > 0 || WLOC0)) wave_lds_sync(); else lds_barrier();
                STAMP(8);

                // ---- wgrad: 16x16 blocks of dW_l^T[o][i] = sum_obs dZ[o][obs] H_in[i][obs] ---------------------
                // step (g,t) contracts observations kbase + 16g + 4q + t: one ds_read_b128 feeds four steps
                if (l == 0) {
                    using WG = WgradPlan<FB, IB1>;
                    const int grp = wv % WG::GROUPS, kp = wv / WG::GROUPS;
                    const int ob = (grp * WG::BPW) / IB1, ib0 = (grp * WG::BPW) - ob * IB1;
                    const float* pa = sZ + (16 * ob + j) * PB + kp * WG::KLEN + 4 * q;
                    const float* pb = sH + (16 * ib0 + j) * PB + kp * WG::KLEN + 4 * q;
                    static_assert(WG::BPW <= WB, "accumulator blocks");
                    f32x4 acc0 = wacc[l][0], acc1 = wacc[l][WB - 1];
                    // operands of the next 16 observations are requested before this step's MFMAs are issued (see CL_SCHED_FENCE)
                    f32x4 a4 = *reinterpret_cast<const f32x4*>(pa);
                    f32x4 b4 = *reinterpret_cast<const f32x4*>(pb);
                    f32x4 c4 = (WG::BPW == 2) ? *reinterpret_cast<const f32x4*>(pb + 16 * PB) : b4;
                    // bias gradient = row sums of this wave's own 16 columns of dZ: read here, summed under the first MFMA group
                    f32x4 z4[4];
                    const float* pz = sZ + (lane < WP ? lane : 0) * PB + CL_WOBS * wv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) z4[e] = *reinterpret_cast<const f32x4*>(pz + 4 * e);
#pragma unroll
                    for (int g = 0; g < WG::KLEN / 16; ++g) {
                        f32x4 na = a4, nb = b4, nc = c4;
                        if (g + 1 < WG::KLEN / 16) {
                            na = *reinterpret_cast<const f32x4*>(pa + 16 * (g + 1));
                            nb = *reinterpret_cast<const f32x4*>(pb + 16 * (g + 1));
                            if (WG::BPW == 2) nc = *reinterpret_cast<const f32x4*>(pb + 16 * PB + 16 * (g + 1));
                        }
                        CL_SCHED_FENCE();
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            acc0 = mfma4(a4[t], b4[t], acc0);
                            if (WG::BPW == 2) acc1 = mfma4(a4[t], c4[t], acc1);
                        }
                        CL_SCHED_FENCE();
                        if (g == 0) {
                            float sb = 0.0f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) sb += (z4[e][0] + z4[e][1]) + (z4[e][2] + z4[e][3]);
                            if (lane < WP) bacc[l] += sb;
                        }
                        a4 = na; b4 = nb; c4 = nc;
                    }
                    wacc[l][0] = acc0;
                    if (WG::BPW == 2) wacc[l][WB - 1] = acc1;
                } else {
                    using WG = WgradPlan<FB, FB>;
                    const int grp = wv % WG::GROUPS, kp = wv / WG::GROUPS;
                    const int ob = (grp * WG::BPW) / FB, ib0 = (grp * WG::BPW) - ob * FB;
                    const float* pa = sZ + (16 * ob + j) * PB + kp * WG::KLEN + 4 * q;
                    const float* pb = sH + (16 * ib0 + j) * PB + kp * WG::KLEN + 4 * q;
                    static_assert(WG::BPW <= WB, "accumulator blocks");
                    const bool acc_lds = (AP::NACC > 0) && (l >= LREG);               // compile-time after unrolling
                    f32x4 acc0 = acc_lds ? acc_slot(l) : wacc[l < LREG ? l : 0][0];
                    f32x4 acc1 = wacc[l < LREG ? l : 0][WB - 1];
                    // operands of the next 16 observations are requested before this step's MFMAs are issued (see CL_SCHED_FENCE)
                    f32x4 a4 = *reinterpret_cast<const f32x4*>(pa);
                    f32x4 b4 = *reinterpret_cast<const f32x4*>(pb);
                    f32x4 c4 = (WG::BPW == 2) ? *reinterpret_cast<const f32x4*>(pb + 16 * PB) : b4;
                    // bias gradient = row sums of this wave's own 16 columns of dZ: read here, summed under the first MFMA group
                    // (narrow kernel: it is column 15 of the accumulator instead, see BONE)
                    f32x4 z4[4];
                    const float* pz = sZ + (lane < WP ? lane : 0) * PB + CL_WOBS * wv;
                    if (!BONE) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) z4[e] = *reinterpret_cast<const f32x4*>(pz + 4 * e);
                    }
                    // all waves left barrier B together and wait for these reads together: the VALU work of the NEXT layer step
                    // (dH_{l-1} = dn, dZ_{l-1} = dH_{l-1} * lrelu'(H_{l-1})) goes here, under the LDS latency
                    CL_SCHED_FENCE();
#pragma unroll
                    for (int mb = 0; mb < FB; ++mb)
#pragma unroll
                        for (int t = 0; t < 4; ++t) dH[mb][t] = (hs[l > 0 ? l - 1 : 0][mb][t] > 0.0f) ? dn[mb][t] : leak * dn[mb][t];
#pragma unroll
                    for (int g = 0; g < WG::KLEN / 16; ++g) {
                        f32x4 na = a4, nb = b4, nc = c4;
                        if (g + 1 < WG::KLEN / 16) {
                            na = *reinterpret_cast<const f32x4*>(pa + 16 * (g + 1));
                            nb = *reinterpret_cast<const f32x4*>(pb + 16 * (g + 1));
                            if (WG::BPW == 2) nc = *reinterpret_cast<const f32x4*>(pb + 16 * PB + 16 * (g + 1));
                        }
                        CL_SCHED_FENCE();
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            acc0 = mfma4(a4[t], b4[t], acc0);
                            if (WG::BPW == 2) acc1 = mfma4(a4[t], c4[t], acc1);
                        }
                        CL_SCHED_FENCE();
                        if (g == 0 && !BONE) {
                            float sb = 0.0f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) sb += (z4[e][0] + z4[e][1]) + (z4[e][2] + z4[e][3]);
                            if (lane < WP) bacc[l] += sb;
                        }
                        a4 = na; b4 = nb; c4 = nc;
                    }
                    if (acc_lds) acc_slot(l) = acc0; else wacc[l < LREG ? l : 0][0] = acc0;
                    if (WG::BPW == 2) wacc[l < LREG ? l : 0][WB - 1] = acc1;
                }
                STAMP(9);
              }

                STAMP(10);
            }
        }
    }
#ifdef CL_STAMPS
    if (MODE == 0 && A.loc_out != nullptr && lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(A.loc_out) + ((size_t)blockIdx.x * CL_NW + wv) * CL_NPH;
#pragma unroll
        for (int k = 0; k < CL_NPH; ++k) dbg[k] = st_acc[k];
    }
#endif

    if (MODE == 1) return;
    if (ILAY && cur_img >= 0) imgl_flush(cur_img);

    // ================= flush the weight-gradient accumulators: LDS staging -> per-workgroup partial =========
    __syncthreads();
    const int offWo = w * d + w + (Ld - 1) * (w * w + w);
    const int Ptot = no_head ? offWo : offWo + 2 * w + 2;
    float bo0 = boacc0, bo1 = boacc1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bo0 += __shfl_xor(bo0, off);
        bo1 += __shfl_xor(bo1, off);
    }
    // (a) kernels.  A 16x16 block of dW_l^T is held by GROUPS waves when the layer has at least eight blocks (every element then has
    //     exactly ONE owner: all waves store at once), or by KPARTS waves that split the observation axis (narrow layers): part 0
    //     stores, the other parts add in part order.  Deterministic either way, and no wave waits for seven others in turn.
    constexpr int NBK0 = FB * IB1, NBKH = FB * FB;
    constexpr int KP0 = (NBK0 >= CL_NW) ? 1 : CL_NW / NBK0, KPH = (NBKH >= CL_NW) ? 1 : CL_NW / NBKH;
    constexpr int NPASS = (KP0 > KPH) ? KP0 : KPH;
    for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < Ld) {
                const int IBn = (l == 0) ? IB1 : FB;
                const int NBK = FB * IBn;
                const int BPW = (NBK >= CL_NW) ? NBK / CL_NW : 1;
                const int GROUPS = NBK / BPW;
                const int grp = wv % GROUPS, kp = wv / GROUPS;
                const int ob = (grp * BPW) / IBn, ib0 = (grp * BPW) - ob * IBn;
                const int in_dim = (l == 0) ? d : w;
                const int offW = (l == 0) ? 0 : (w * d + w + (l - 1) * (w * w + w));
                if (kp == pass) {
#pragma unroll
                    for (int b = 0; b < WB; ++b) {
                        if (b < BPW) {
                            const int islot = 16 * (ib0 + b) + j;
                            const int i = (l == 0) ? islot : sl(islot);        // input feature: metadata column, or hidden feature of a slot
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const int o = sl(16 * ob + 4 * q + t);
                                if (o < w && i < in_dim) {
                                    float* dst = smem + offW + o * in_dim + i;
                                    const float g = (AP::NACC > 0 && l >= LREG) ? acc_slot(l)[t] : wacc[l < LREG ? l : 0][b][t];
                                    *dst = (pass == 0) ? g : *dst + g;
                                }
                                if (BONE && l > 0 && o < w && islot == 15) {      // the bias gradient rides in column (slot) 15
                                    float* dst = smem + offW + w * in_dim + o;
                                    const float g = (AP::NACC > 0 && l >= LREG) ? acc_slot(l)[t] : wacc[l < LREG ? l : 0][b][t];
                                    *dst = (pass == 0) ? g : *dst + g;
                                }
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // (b) biases and the Dense(2) head: every wave holds a partial (its own 16 observations).  Each wave parks its values in a
    //     private scratch row; one pass then adds the eight rows in wave order.
    float* const scr = smem + ((Ptot + 3) & ~3);
    constexpr int SCRW = LMAX * WP + 2 * WP + 2;
    {
        float* mine = scr + wv * SCRW;
        if (WLOC && !BONE) {             // width 16: the four lanes (j, q) of a row hold four observations' share each (layer 0: lanes q = 0 only)
#pragma unroll
            for (int l = 1; l < LMAX; ++l) { bacc[l] += __shfl_xor(bacc[l], 16); bacc[l] += __shfl_xor(bacc[l], 32); }
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
            if (l < Ld && lane < WP) mine[l * WP + lane] = (BONE && l > 0) ? 0.0f : bacc[l];
        if (lane < WP) { mine[LMAX * WP + lane] = woacc0; mine[LMAX * WP + WP + lane] = woacc1; }
        if (lane == 0) { mine[LMAX * WP + 2 * WP] = bo0; mine[LMAX * WP + 2 * WP + 1] = bo1; }
    }
    __syncthreads();
    for (int idx = tid; idx < SCRW; idx += 512) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < CL_NW; ++k) t += scr[k * SCRW + idx];
        if (idx < LMAX * WP) {
            const int l = idx / WP, o = sl(idx - l * WP);           // row (slot) of the scratch image -> hidden feature
            if (l < Ld && o < w && !(BONE && l > 0))
                smem[((l == 0) ? 0 : (w * d + w + (l - 1) * (w * w + w))) + w * ((l == 0) ? d : w) + o] = t;
        } else if (!no_head) {
            const int r = idx - LMAX * WP;
            if (r < WP) { if (sl(r) < w) smem[offWo + sl(r)] = t; }
            else if (r < 2 * WP) { if (sl(r - WP) < w) smem[offWo + w + sl(r - WP)] = t; }
            else smem[offWo + 2 * w + (r - 2 * WP)] = t;
        }
    }
    __syncthreads();
    float* __restrict__ part = kernargs_again()->partials + (size_t)blockIdx.x * Ptot;       // (re-read: not held across the tile loop)
    for (int idx = tid; idx < Ptot; idx += 512) part[idx] = smem[idx];

    if (MODE == 0) {
        float v = nll_acc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        // one fp64 atomic per workgroup (same-address atomics serialise at ~12 ns each): combine the 8 wave sums through LDS
        __syncthreads();
        if (lane == 0) smem[wv] = v;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int k = 0; k < CL_NW; ++k) t += (double)smem[k];
#if CL_DET
            kernargs_again()->nll_part[blockIdx.x] = t;          // (every workgroup of the launch writes its slot: workgroups without tiles write 0)
#else
            atomicAdd(kernargs_again()->scalars + CL_SC_NLL, t);
#endif
        }
        if (use_ev11) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                ev_g0 += __shfl_xor(ev_g0, off); ev_g1 += __shfl_xor(ev_g1, off); ev_g2 += __shfl_xor(ev_g2, off);
            }
            if (lane == 0) {                         // d softplus(raw)/d raw = sigmoid(raw)
                // (the flush's pointers are re-read from the kernel-argument block HERE: read through `A` they are loaded at the top of the
                // kernel and held in scalar registers across the tile loop -- three more spilled registers of each kind in the 64-wide
                // instance, 0.2 - 0.6 % of the headline step: profiles/r5_mlp_r3_vs_r4_ab.txt)
                cl_args_p F = kernargs_again();
                const float* ev = F->ev11;
                const float e0 = ev_g0 * cl_sigmoid(ev[0]), e1 = ev_g1 * cl_sigmoid(ev[1]), e2 = ev_g2 * cl_sigmoid(ev[2]);
                float* part = F->ev11_part;
                if (part != nullptr) {               // deterministic mode: this wave's slot, summed in index order by cl_det_reduce
                    float* slot = part + 3 * (CL_EV11_WAVES * (size_t)blockIdx.x + wv);
                    slot[0] = e0; slot[1] = e1; slot[2] = e2;
                } else {
                    float* dst = F->d_ev11;
                    atomicAdd(dst + 0, e0); atomicAdd(dst + 1, e1); atomicAdd(dst + 2, e2);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// sum of the per-workgroup partials in a fixed order -> MLP slice of the flat gradient buffer
// ---------------------------------------------------------------------------------------------------------
// (the work of a workgroup is cl_reduce_partials_block, cl_kernels.h: shared with the step's cl_tn_backward launch, which can carry it)
#if !CL_IMGL && !CL_CHAIN && !CL_DET
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, int nparts, int P,
                                                               float* __restrict__ out, const int* stop_flag) {
    if (stop_flag != nullptr && *stop_flag != 0) return;
    cl_reduce_partials_block(partials, nparts, P, out, (int)blockIdx.x);
}
#endif

// ---------------------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------------------
#ifndef CL_IMGL
#define CL_IMGL 0            // the file is compiled three times: plain (-DCL_IMGL=0 -DCL_CHAIN=0), packed layouts / per-image layers
#endif                       // (-DCL_IMGL=1) and layer-block chains (-DCL_CHAIN=1)
#ifndef CL_CHAIN
#define CL_CHAIN 0
#endif

template <int WP, int DP, int LMAX, int MODE, int KS = 4>
static int launch_one(const cl_mlp_args& a, int grid, hipStream_t st) {
    using SL = SmemLayout<WP, DP, LMAX>;
    const size_t sm_tiles = (size_t)SL::total * sizeof(float);
    const size_t P = (size_t)a.w * a.d + a.w + (size_t)(a.L - 1) * (a.w * a.w + a.w) + 2 * a.w + 2;
    size_t sm = sm_tiles;
    const size_t flush = (P + 4 + (size_t)CL_NW * (LMAX * WP + 2 * WP + 2)) * sizeof(float);      // gradient image + per-wave bias rows
    if (MODE != 1 && flush > sm) sm = flush;
    using AP = AccPlan<WP, DP, LMAX, MODE, (CL_IMGL == 1)>;
    if (AP::NACC > 0) sm = (size_t)AP::total * sizeof(float);                                       // + LDS-resident accumulators
    if (sm > 160 * 1024) return -3;
    auto kern = elbo_mlp_kernel<WP, DP, LMAX, MODE, (CL_IMGL != 0), (CL_CHAIN != 0), (CL_IMGL == 1), KS, (CL_DET != 0)>;
    // largest dynamic-LDS size this instance has been configured for (one process drives one device; host threads may race here:
    // setting the attribute twice is harmless, publishing a size that was not set is not, hence set first, then raise the mark)
    static std::atomic<size_t> configured{0};
    size_t have = configured.load(std::memory_order_acquire);
    if (have < sm) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        while (have < sm && !configured.compare_exchange_weak(have, sm, std::memory_order_release, std::memory_order_acquire)) {}
    }
    (void)hipGetLastError();   // drop any stale error of an unrelated earlier runtime call
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), sm, st, a);
    return (int)hipGetLastError();
}

// Instantiated geometries: the padded width WP fixes how many layers of activations + weight-gradient blocks fit in the
// 256-register budget of a wave: w <= 15 -> up to 20 layers (the CLI default scaler is 20 x 10), w <= 32 -> 10, w <= 64 -> 5.
template <int WP, int LMAX, int MODE, int KS = 4>
static int launch_dp(const cl_mlp_args& a, int grid, hipStream_t st) {
    if (a.L + (CL_IMGL == 1 ? a.n_imgl : 0) > LMAX) return -2;
    const int dp = (a.d <= 8) ? 8 : (a.d <= 32 ? 32 : 64);
    if (dp == 8) return launch_one<WP, 8, LMAX, MODE, KS>(a, grid, st);
    if (dp == 32) return launch_one<WP, 32, LMAX, MODE, KS>(a, grid, st);
    return launch_one<WP, 64, LMAX, MODE, KS>(a, grid, st);
}

template <int MODE>
static int launch_mode(const cl_mlp_args& a, int grid, hipStream_t st) {
    if (a.L < 1 || a.w < 1 || a.d < 1) return -2;
    if (a.w > 64 || a.d > 64) return -2;
    // Per-image layers on more than 32 metadata columns at hidden width <= 15: the 16-wide instance <16, 64, 24, IMGL> (864 - 880 bytes of
    // scratch per lane, the largest of the library) ends in a GPU memory fault when a workgroup walks more than one tile and crosses an
    // image border (found by a random draw late in round 6: scripts/probe/imgl_abort_probe.py, NOTEBOOK R6.4; the forward-only launch
    // is not affected).  Those shapes take the 32-wide instance, which holds up to CL_MLP_LMAX_W32 layers (deeper: the caller's
    // layer-by-layer path, as for any scaler deeper than one launch).
    const bool imgl_d64 = (CL_IMGL == 1) && MODE != 1 && a.n_imgl > 0 && a.d > 32;
    if (a.w <= 15 && !imgl_d64) {
        // the narrow instance comes in three step counts (hidden width <= 8, <= 12, <= 15); the forward-only launch keeps all four
        constexpr int L16 = (CL_IMGL == 1 ? CL_MLP_LMAX_W16_IMGL : CL_MLP_LMAX_W16);
        if (MODE != 1 && a.w <= 8) return launch_dp<16, L16, MODE, 2>(a, grid, st);
        if (MODE != 1 && a.w <= 12) return launch_dp<16, L16, MODE, 3>(a, grid, st);
        return launch_dp<16, L16, MODE, 4>(a, grid, st);
    }
#if CL_IMGL != 1
    if (a.w == 16) return launch_dp<16, CL_MLP_LMAX_W16, MODE, 5>(a, grid, st);       // (per-image layers keep the 32-wide instance: cl_mlp_max_layers_imgl)
#endif
    if (a.w <= 32) return (a.L + (CL_IMGL == 1 ? a.n_imgl : 0) <= 5) ? launch_dp<32, 5, MODE>(a, grid, st) : launch_dp<32, CL_MLP_LMAX_W32, MODE>(a, grid, st);
    return launch_dp<64, CL_MLP_LMAX_W64, MODE>(a, grid, st);
}

#if !CL_IMGL && !CL_CHAIN && !CL_DET
static bool lane_enabled() {            // CARELESS_HIP_LANE=0 keeps narrow scalers off the lane-per-observation kernel (A/B runs)
    static const bool on = [] { const char* e = getenv("CARELESS_HIP_LANE"); return !(e != nullptr && e[0] == '0'); }();
    return on;
}
static bool narrow_enabled() {          // CARELESS_HIP_NARROW=0 keeps narrow scalers on the eight-wave instance of this file (A/B runs)
    static const bool on = [] { const char* e = getenv("CARELESS_HIP_NARROW"); return !(e != nullptr && e[0] == '0'); }();
    return on;
}
#endif

#if CL_DET && CL_IMGL == 2
// packed layout (single-pass Laue) without float atomics: the seventh compilation of this file (build.py: elbo_mlp_packed_det)
int cl_launch_mlp_packed_det(const cl_mlp_args& a, int mode, int grid, hipStream_t st) {
    if (mode != 0 || a.row_map == nullptr || a.n_obs != a.n_pad || a.n_imgl != 0 || (a.ev11 != nullptr && a.ev11_part == nullptr)) return -2;
    if (a.gmeta != nullptr && a.tile_gmax == nullptr) return -1;
    if (a.dzf_obs == nullptr || a.nll_part == nullptr || (a.use_img && a.dimg_obs == nullptr)) return -1;
    if (4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
#elif CL_DET && CL_CHAIN
// the LAST block of a layer-block chain (the one launch of a chained scaler with an epilogue) without float atomics: the eighth compilation
// of this file (build.py: elbo_mlp_chain_det).  The chain's forward-only and backward-only launches have no atomics and keep the chain unit.
int cl_launch_mlp_chain_det(const cl_mlp_args& a, int mode, int grid, hipStream_t st) {
    if (mode != 0 || a.row_map != nullptr || a.n_imgl > 0 || a.act_out != nullptr || a.dH_ext != nullptr || a.dX_out == nullptr || (a.ev11 != nullptr && a.ev11_part == nullptr)) return -2;
    if (a.dzf_obs == nullptr || a.nll_part == nullptr || (a.use_img && a.dimg_obs == nullptr)) return -1;
    if (4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
#elif CL_DET
int cl_launch_mlp_det(const cl_mlp_args& a, int mode, int grid, hipStream_t st) {
    // plain layout, full step, every store target present (with the Evans-2011 terms: their per-wave slots)
    if (mode != 0 || a.row_map != nullptr || a.n_imgl > 0 || a.act_out != nullptr || a.dH_ext != nullptr || a.dX_out != nullptr || (a.ev11 != nullptr && a.ev11_part == nullptr)) return -2;
    if (a.dzf_obs == nullptr || a.nll_part == nullptr || (a.use_img && a.dimg_obs == nullptr)) return -1;
    if (4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
#elif CL_CHAIN
int cl_launch_mlp_chain(const cl_mlp_args& a, int mode, int grid, hipStream_t st) {
    if (a.row_map != nullptr || a.n_imgl > 0) return -2;                       // chains use the plain layout
    if (a.act_out != nullptr && mode != 1) return -1;
    if (a.dH_ext != nullptr && mode != 2) return -1;
#elif CL_IMGL == 2
int cl_launch_mlp_packed(const cl_mlp_args& a, int mode, int grid, hipStream_t st) {
    if (a.row_map == nullptr || a.n_obs != a.n_pad || a.n_imgl != 0) return -1;
    if (a.gmeta != nullptr && (a.tile_gmax == nullptr || mode != 0)) return -1;
    if ((a.eta != nullptr || a.ipred_out != nullptr) && 4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
#elif CL_IMGL
int cl_launch_mlp_imgl(const cl_mlp_args& a, int mode, int grid, hipStream_t st) {
    if (a.row_map == nullptr || a.n_obs != a.n_pad || a.n_imgl < 0) return -1;
    if (a.n_imgl > 0 && (a.imgl == nullptr || a.tile_img == nullptr || a.n_images < 1 || a.use_img)) return -1;
    if (a.n_imgl > 0 && mode != 1 && a.d_imgl == nullptr) return -1;
    if (a.gmeta != nullptr && (a.tile_gmax == nullptr || mode != 0)) return -1;
    // eta / ipred_out are addressed with 32-bit byte offsets from their base in this variant
    if ((a.eta != nullptr || a.ipred_out != nullptr) && 4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
#else
int cl_launch_mlp(const cl_mlp_args& a, int mode, int grid, hipStream_t st) {
    // per-image layers on the lane kernel: ONE predicate for the dZ0_out guard below and for the dispatch (a launch that fell through
    // to the IMGL instances of this file would ignore dZ0_out and leave the peeled layer a zero gradient without any error)
    const bool lane_imgl_route = a.n_imgl > 0 && mode == 0 && cl_lane_imgl_supports(a) && lane_enabled() && a.n_pad > 0 && a.n_pad % CL_TILE == 0 && grid >= 1;
    // dZ0_out (the launch behind a peeled first layer, elbo_peel.hip) is stored by the default scaler's kernels only
    if (a.dZ0_out != nullptr && !(mode == 0 && a.act_out == nullptr && a.dH_ext == nullptr && a.dX_out == nullptr &&
                                  (a.dzf_obs == nullptr || a.ev11 == nullptr || a.ev11_part != nullptr) &&
                                  ((a.n_imgl == 0 && ((cl_lane_supports(a) && lane_enabled()) || (cl_narrow_supports(a) && narrow_enabled()))) || lane_imgl_route)))
        return -2;
    if (a.dzf_obs != nullptr && (mode == 0 || (a.act_out == nullptr && a.dH_ext == nullptr))) {
        // deterministic mode: the default scaler's shapes keep their own kernels (round 4: elbo_lane.hip / elbo_narrow.hip store per
        // observation when dzf_obs is given), every other width <= 64 runs the deterministic compilation of this file.  (The forward-only
        // and backward-only launches of a layer-block chain have no float atomics: they take the chain unit below.)
        // per-image layers (round 6): the lane kernel's instances only (one wave per image); the IMGL instances of this file keep their float atomics
        if (a.n_imgl > 0) return lane_imgl_route ? cl_launch_lane_imgl(a, grid > a.n_pad / CL_TILE ? a.n_pad / CL_TILE : grid, st) : -2;
        if (mode == 0 && a.dX_out != nullptr) return cl_launch_mlp_chain_det(a, mode, grid, st);       // the chain's last block: the one with the epilogue
        if (mode == 0 && (a.ev11 == nullptr || a.ev11_part != nullptr) && a.n_pad > 0 && a.n_pad % CL_TILE == 0 && grid >= 1) {
            const int g = grid > a.n_pad / CL_TILE ? a.n_pad / CL_TILE : grid;
            if (cl_lane_supports(a) && lane_enabled()) return cl_launch_lane(a, g, st);
            if (cl_narrow_supports(a) && narrow_enabled()) return cl_launch_narrow(a, g, st);
        }
        if (a.row_map != nullptr && a.n_imgl == 0) return cl_launch_mlp_packed_det(a, mode, grid, st);      // single-pass Laue, wider than 15
        return cl_launch_mlp_det(a, mode, grid, st);
    }
    // the first block of a chained narrow scaler: the lane kernel's forward-only / external-gradient instances (round 6)
    if ((a.act_out != nullptr || a.dH_ext != nullptr) && cl_lane_block_supports(a, mode) && lane_enabled() && a.n_pad > 0 && a.n_pad % CL_TILE == 0 && grid >= 1)
        return cl_launch_lane_block(a, mode, grid > a.n_pad / CL_TILE ? a.n_pad / CL_TILE : grid, st);
    if (a.act_out != nullptr || a.dH_ext != nullptr || a.dX_out != nullptr) return cl_launch_mlp_chain(a, mode, grid, st);
    if (a.n_imgl > 0) {                                                        // packed layout + per-image layers
        // the default scaler's depth and width with one or two per-image layers: the lane-per-observation kernel (elbo_lane.hip, round 5)
        if (lane_imgl_route) return cl_launch_lane_imgl(a, grid > a.n_pad / CL_TILE ? a.n_pad / CL_TILE : grid, st);
        return cl_launch_mlp_imgl(a, mode, grid, st);
    }
    if (a.row_map != nullptr && !(mode == 0 && ((cl_narrow_supports(a) && narrow_enabled()) || (cl_lane_supports(a) && lane_enabled()))))
        return cl_launch_mlp_packed(a, mode, grid, st);                          // packed layout (single-pass Laue)
#endif
    if (a.n_pad % CL_TILE != 0 || a.n_pad <= 0) return -1;
    // 32-bit byte offsets / buffer sizes inside the kernel: metadata image < 4 GiB, z_f < 4 GiB (shard further across GPUs otherwise)
    const unsigned long long meta_bytes = 4ull * (unsigned long long)((a.d + 3) & ~3) * (unsigned long long)a.n_pad;
    if (meta_bytes >= (1ull << 32) || 4ull * (unsigned long long)a.R * (unsigned long long)a.S >= (1ull << 32)) return -4;
    const int ntiles = a.n_pad / CL_TILE;
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) return -1;
#if !CL_IMGL && !CL_CHAIN && !CL_DET
    // hidden width <= 15 (the careless CLI default): the full step runs on the one-wave-per-SIMD kernel of elbo_narrow.hip
    // (CARELESS_HIP_NARROW=0 keeps the eight-wave instance below: A/B measurements)
    if (mode == 0 && cl_lane_supports(a) && lane_enabled()) return cl_launch_lane(a, grid, st);      // the default scaler's shape: lane = observation (elbo_lane.hip)
    if (mode == 0 && cl_narrow_supports(a) && narrow_enabled()) return cl_launch_narrow(a, grid, st);
#endif
    switch (mode) {
        case 0: return launch_mode<0>(a, grid, st);
        case 1: return launch_mode<1>(a, grid, st);
        case 2: return launch_mode<2>(a, grid, st);
    }
    return -1;
}

#if !CL_IMGL && !CL_CHAIN && !CL_DET
// Name of the kernel instance cl_launch_mlp(a, mode, ...) runs -- the same routing, restated once, here, next to it (bench.py and the
// profiling scripts label their rows with it instead of guessing).  Returns the length written (snprintf semantics).
int cl_mlp_kernel_name_of(const cl_mlp_args& a, int mode, char* out, size_t n) {
    const char* unit = "";
    bool packed = false;
    if (a.dzf_obs != nullptr && (mode == 0 || (a.act_out == nullptr && a.dH_ext == nullptr))) {
        if (a.n_imgl > 0) return (mode == 0 && cl_lane_imgl_supports(a) && lane_enabled()) ? cl_lane_imgl_kernel_name(a, out, n) : snprintf(out, n, "(unsupported)");
        if (mode == 0 && (a.ev11 == nullptr || a.ev11_part != nullptr) && cl_lane_supports(a) && lane_enabled()) return cl_lane_kernel_name(a, out, n);
        if (mode == 0 && (a.ev11 == nullptr || a.ev11_part != nullptr) && cl_narrow_supports(a) && narrow_enabled()) return cl_narrow_kernel_name(a, out, n);
        unit = (mode == 0 && a.dX_out != nullptr) ? ", chain deterministic" : ((a.row_map != nullptr && a.n_imgl == 0) ? ", packed deterministic" : ", deterministic");
    } else if (a.act_out != nullptr || a.dH_ext != nullptr || a.dX_out != nullptr) unit = ", chain";
    else if (a.n_imgl > 0) {
        if (mode == 0 && cl_lane_imgl_supports(a) && lane_enabled())
            return cl_lane_imgl_kernel_name(a, out, n);
        unit = ", image layers";
    } else if (a.row_map != nullptr) { unit = ", packed"; packed = true; }
    if (a.dzf_obs == nullptr && (unit[0] == 0 || packed)) {
        if (mode == 0 && cl_lane_supports(a) && lane_enabled()) return cl_lane_kernel_name(a, out, n);
        if (mode == 0 && cl_narrow_supports(a) && narrow_enabled()) return cl_narrow_kernel_name(a, out, n);
    }
    if (a.w < 1 || a.w > 64 || a.d < 1 || a.d > 64) return snprintf(out, n, "(unsupported)");
    const int imgl = a.n_imgl > 0 ? a.n_imgl : 0;
    const bool w16 = a.w == 16 && imgl == 0;                     // width exactly 16: the 16-wide instance without the constant-one feature (KS = 5)
    const bool imgl_d64 = imgl > 0 && mode != 1 && a.d > 32;     // (launch_mode: these shapes take the 32-wide instance)
    const int WP = ((a.w <= 15 && !imgl_d64) || w16) ? 16 : (a.w <= 32 ? 32 : 64);
    const int DP = a.d <= 8 ? 8 : (a.d <= 32 ? 32 : 64);
    const int LM = WP == 16 ? (imgl ? CL_MLP_LMAX_W16_IMGL : CL_MLP_LMAX_W16) : (WP == 32 ? (a.L + imgl <= 5 ? 5 : CL_MLP_LMAX_W32) : CL_MLP_LMAX_W64);
    const int KS = w16 ? 5 : ((WP == 16 && mode != 1) ? (a.w <= 8 ? 2 : (a.w <= 12 ? 3 : 4)) : 4);
    return snprintf(out, n, "elbo_mlp_kernel<%d, %d, %d, %d%s, KS=%d>", WP, DP, LM, mode, unit, KS);
}

int cl_launch_reduce_partials(const float* partials, int nparts, int P, float* out, const int* stop_flag, hipStream_t st) {
    (void)hipGetLastError();   // drop any stale error of an unrelated earlier runtime call
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((P + 31) / 32), dim3(256), 0, st, partials, nparts, P, out, stop_flag);
    return (int)hipGetLastError();
}
#endif
