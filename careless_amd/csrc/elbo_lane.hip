// Lane-per-observation instance of the fused ELBO step for narrow scalers (gfx950 / CDNA4 only): hidden width <= 10 (12 with the metadata
// in registers), metadata width <= 31, any number of MC samples -- the geometry of the careless CLI default (--mlp-layers 20, --mlp-width 10,
// --mc-samples 1: careless/args/scaling.py:21-31), also with the 16 extra metadata columns of --positional-encoding-keys X,Y
// (careless/utils/positional_encoding.py:3-17).  The depth is a compile-time constant of the unit: the default build has 20 Dense layers,
// build.py compiles the file again for every depth 2 .. 19 (round 6: -DCL_LANE_NL); widths 13 - 15 and one-layer scalers run on elbo_narrow.hip.
//
// Same arithmetic and the same reference lines as elbo_mlp.hip (scaler forward / sample / predict / likelihood / backward:
// careless/models/scaling/nn.py:92-120, image.py:53-63, models/merging/variational.py:156-181, 197-202,
// models/likelihoods/mono.py:10-73), a third work decomposition.  A width-10 layer on v_mfma_f32_16x16x4_f32 (elbo_narrow.hip)
// issues 16 x 16 x 12 multiply-adds per 16 observations for 10 x 11 useful ones, behind an LDS round trip per layer, and its
// LeakyReLUs contend with the MFMAs for the one fp32 datapath of the SIMD.  Here
//   * a lane IS an observation for the whole tile: a wave carries 64 observations, feature f of a layer is one register;
//   * a Dense layer is a chain of v_mfma_f32_4x4x1_16b_f32: the instruction's 16 blocks are 16 groups of four observations, step k
//     multiplies input feature k (B operand = the activation register as it stands) into four output features (D = four
//     registers of the lane).  The A operand of step k is block k of ONE register that holds W[4c .. 4c+3][0 .. 15] of output
//     chunk c, broadcast to all blocks with CBSZ = 4 / ABID = k: a layer's forward weights are ceil(w / 4) registers, read from
//     LDS with one ds_read_b32 each.  Steps = input width + 1 (the bias rides on block 15 with a register of ones), no padding
//     of the contraction, 4-row granularity on the outputs: 33 instructions of 8 cycles for 64 observations of a width-10 layer
//     (12 of 32 cycles in the 16 x 16 form).  dgrad is the same chain on the transposed image;
//   * activations never visit LDS on the forward pass; LeakyReLU, its derivative, the sampling epilogue are plain per-lane code;
//   * the weight gradient contracts over observations, i.e. over lanes: dZ_l and the layer's input are staged [feature][observation]
//     in LDS (conflict-free: a lane writes its own column) and read back as operands of 16 v_mfma_f32_16x16x4_f32 per layer;
//     all accumulators (two 16 x 16 blocks per layer, in accumulator registers) stay put for the whole launch; the LDS traffic of
//     layer l - 1 rides in the shadow of layer l's weight-gradient MFMAs;
//   * the Dense(2) head is one more 4-row chunk (its outputs are the lane's loc and raw sigma); its weight gradient is 2 (w + 1)
//     per-lane sums, reduced across lanes once at the end of the launch;
//   * ONE wave per SIMD with the whole 512-register file: 20 layers x w activations per observation stay in registers.
// Roofline: fp32 MFMA; algorithmic flops per observation 6 (d w + (L-1) w^2 + 2 w).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "cl_math.h"
#include "cl_kernels.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef CL_LANE_COALESCE
#define CL_LANE_COALESCE 2      /* smallest number of MC samples whose amplitude gradients leave through LDS (0: never) */
#endif

namespace {

// Dense layers of the scaler: a compile-time constant of the unit.  The default build has the careless default depth (20 = the most one
// launch holds); round 6 compiles the file again with -DCL_LANE_NL=2 .. 19 (build.py: one part per depth, 16 - 25 s each, the widest
// instances only) for `--mlp-layers` below the default -- a run-time depth test per unrolled layer costs this one-wave-per-SIMD kernel its
// register allocation (NOTEBOOK R6.3: 120 - 380 spilled registers and accumulator copies behind the inline-assembly MFMAs).
#ifndef CL_LANE_NL
#define CL_LANE_NL CL_MLP_LMAX_W16
#endif
constexpr int NL = CL_LANE_NL;
constexpr int NWV = 4;                // waves of a workgroup (one per SIMD)
constexpr int NT = 64 * NWV;
constexpr int WT = 64;                // observations of a wave tile
#ifndef CL_LANE_PIT
#define CL_LANE_PIT 68
#endif
constexpr int PIT = CL_LANE_PIT;               // row pitch of a staging tile [16 features][64 observations]
constexpr int DMAX_ALL = 15;          // metadata columns of the register instances (block 15 of a weight register is the bias): every column is a register of the lane
constexpr int DMAX_LX = 31;           // ... of the LDS-row instances (DMAX = 0): two input blocks of layer 0, the ones row in block 0
constexpr int SPRE = 8;               // MC samples of a batch: their amplitudes are gathered into LDS together, their amplitude gradients leave through it
constexpr int ONE = 15;               // block of a weight register (= row of a staging tile) that belongs to the constant-one feature

template <int K, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, N>(f);
    }
}

// D[4 features][64 observations] += A(block K of `a`, broadcast) x B(`b`: one input feature of every lane's observation)
template <int K>
__device__ __forceinline__ f32x4 mfma_bk(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, K, 0); }
// The weight-gradient accumulators live in accumulator registers for the whole launch and nothing but these MFMAs touches them
// (two per layer, so that consecutive MFMAs never depend on each other): written as inline assembly, because with the compiler's
// own choice every layer of every tile pays copies between the two register files.  (Same-destination MFMAs need no software
// wait states between them; the only other reader is the flush, a barrier later.)
// Diagnostic switches of round 6 (scripts/probe/build_lane_variants.py; NOTEBOOK R6.1): CL_LANE_MFMA_BUILTIN / CL_LANE_SEL_C replace one
// kind of inline-assembly statement by code the compiler knows, CL_LANE_LRELU_ASM brings the inline-assembly LeakyReLU back; CL_LANE_PAD_PRE / CL_LANE_PAD_POST pad the MFMA statement.
// (The unit of the three-per-image-layer instances -- CL_LANE_PART 5, compiled WITHOUT -amdgpu-mfma-vgpr-form: see there -- spills accumulator
//  registers, and the compiler's reload copies land one instruction in front of the statement whose MFMA reads them: two wait states in front of
//  every such MFMA, in that unit only.)
#if !defined(CL_LANE_PAD_PRE) && defined(CL_LANE_PART) && CL_LANE_PART == 5
#define CL_LANE_PAD_PRE "s_nop 1\n\t"
#endif
#ifndef CL_LANE_PAD_PRE
#define CL_LANE_PAD_PRE ""
#endif
#ifndef CL_LANE_PAD_POST
#define CL_LANE_PAD_POST ""
#endif
__device__ __forceinline__ void mfma16_acc(f32x4& acc, float a, float b) {
#ifdef CL_LANE_MFMA_BUILTIN
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#else
    asm volatile(CL_LANE_PAD_PRE "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" CL_LANE_PAD_POST : "+a"(acc) : "v"(a), "v"(b));
#endif
}

// max(x, leak x), leak x given.  ONE v_max_f32 that the compiler KNOWS: this unit is compiled with -fno-honor-nans (build.py), under
// which fmaxf needs no canonicalising v_max_f32 x, x in front (signalling NaNs are the only inputs that would tell the difference).
// Until round 6 this was an inline-assembly v_max_f32 -- opaque to hipcc's hazard recognizer.  Its result is the B operand of the next
// layer's MFMAs, gfx950 wants two wait states between a vector-ALU write and an MFMA reading it, and hipcc pads them only between
// instructions it knows: wherever the scheduler left ONE instruction between the two, the MFMA read the register's old content on
// some launches (NOTEBOOK R6.1: the run-to-run defect of round 5's two withdrawn dZ_0-storing instances, and four latent sites in
// shipped ones).  scripts/check_lane_isa.py holds every code object of the library to the rule; -DCL_LANE_LRELU_ASM rebuilds the old form.
__device__ __forceinline__ float lrelu2(float x, float lx) {
#ifdef CL_LANE_LRELU_ASM
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(lx));
    return r;
#else
    return __builtin_fmaxf(x, lx);
#endif
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T ld_uo(const T* base, unsigned byte_off) {       // (wave-uniform pointer)[32-bit per-lane byte offset]
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

#define LFENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef CL_STAMPS
#define LSTAMP(k)                                                                                  \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
        st_acc[k] += t_ - st_last;                                                                 \
        st_last = t_;                                                                              \
    } while (0)
#else
#define LSTAMP(k)
#endif

// LX: the metadata of a wave tile do not pass through registers at all (16 .. 31 columns: with every column a register of the lane,
// twice with the prefetch, the 512-register file is over).  The next tile's rows are copied global -> LDS by the DMA path
// (global_load_lds: no registers in between) into the second of two row buffers while the backward pass of the current tile runs,
// layer 0 reads its B operands row by row from the current buffer, and its weight gradient -- two 16-row input blocks, the ones
// row of the bias in block 0 -- reads the same rows back transposed.  The staging tiles of dZ / H then come in ONE copy instead of
// one per layer parity (a wave's LDS operations execute in order, so the second copy only ever bought scheduling freedom): that
// is what pays for the row buffers.
// NI > 0 (round 5): per-image layers on top of the NL Dense ones (NeuralImageScaler): every WAVE holds the weight images of its current
// image's layers (forward + transposed) and parks the activations of the top PK layers in LDS across the sampling epilogue, where the
// register file is fullest; the staging tiles come in one copy (as for LX) to pay for both.
template <int W, bool LX, int NI = 0>
struct LSmem {
    static constexpr int NC = (W + 3) / 4;                    // 4-feature chunks of a layer's outputs
    static constexpr int IMG = NC * 64;                       // one layer's weight registers: [chunk][lane]
    static constexpr int NF = NL + 1 + (LX ? 1 : 0);          // forward images: layers 0 .. NL-1, the head (NL), LX: layer 0's second input block (NL + 1)
    static constexpr int TPAR = (LX || NI > 0) ? 1 : 2;       // copies of the dZ / H staging tiles
    static constexpr int PK = NI > 0 ? 4 : 0;                 // top layers whose activations wait in LDS during the epilogue
    static constexpr int oF = 0;
    static constexpr int oK = oF + NF * IMG;                  // transposed (dgrad) images, layers 0 .. NL-1 and the head
    static constexpr int oT = oK + (NL + 1) * IMG;            // per wave: sZ[TPAR], sH[TPAR] (by layer parity), sX (not LX)   [16][PIT] each
    static constexpr int oS = (2 * TPAR + (LX ? 0 : 1)) * 16 * PIT;   // (within a wave's region) sS: the lane's SPRE sampled amplitudes / amplitude gradients; sQ
    static constexpr int oI = oS + (2 * SPRE + 1) * 64;       // (sS, sQ, sE: the batch's scale noise); NI: the wave's image-layer weight images [fwd | transposed][NI][IMG]
    static constexpr int oP = oI + 2 * NI * IMG;              // NI: parked activations [PK][W][64]
    static constexpr int oX = oP + PK * W * 64;               // LX: two buffers of `xrows` metadata rows [row][PIT]
    static constexpr int TWF = oX;                            // floats of a wave's region without the row buffers
    static constexpr int NACCB = NL + (LX ? 1 : 0);           // 16 x 16 accumulator blocks a wave parks in the flush
    static constexpr int REG = NACCB * 256;
    static constexpr int flush_total = (NWV / 2) * REG + NWV * 2 * 16;
    // rows of a metadata buffer: input k sits in row k (k < 15) or k + 1 (row 15 holds the ones of the bias), whole groups of four
    // inputs are read by layer 0 (the rows past d stay zero)
    static constexpr int xrows(int d) {
        if (!LX) return 0;
        int klast = ((d + 3) & ~3) - 1;                       // last input layer 0 reads (input 31 never is: d <= 31)
        if (klast > 30) klast = 30;
        const int r = (klast < 15 ? klast : klast + 1) + 1;
        return r < 16 ? 16 : r;
    }
    static constexpr int tw(int d) { return TWF + 2 * xrows(d) * PIT; }
    // (the transposed reads of the second buffer's upper block run up to row 31: kept inside the allocation)
    static constexpr int main_total(int d) { return oT + NWV * tw(d) + (LX ? (32 - xrows(d)) * PIT : 0); }
    static constexpr int total(int d) { return main_total(d) > flush_total ? main_total(d) : flush_total; }
};

}  // namespace

// PACKED: the packed observation layout of include/careless_hip.h (row_map; single-pass Laue: gmeta / tile_gmax / noise_row): rows the
// engine ordered so that a harmonic group sits inside a 16-row granule, padding rows have refl_id = -1, n_obs == n_pad; everything
// the caller indexes by row (eta, ipred_out, the noise key) goes through row_map.
// The scaler has all NL layers (the careless default; cl_lane_supports): no per-layer depth tests -- a lone wave pays every taken or
// untaken branch in full (a generic-depth instance of this kernel lost to elbo_narrow.hip at every depth below NL: 12 x 10 at 4 M
// observations 1.27 against 0.80 ms per step) -- and a backward pass whose LDS traffic rides in the shadow of the previous layer's
// weight-gradient MFMAs.
// DMAX: metadata columns the instance holds in registers (8 or 15: the registers of seven more columns cost the common narrow case
// 8 %); 0 = the LDS-row instance (LX, see LSmem) for 16 .. 31 columns.
// FULL: every optional input / output of the launch (injected scale noise `eta`, `ipred_out`, the Evans-2011 error model).  The
// training step of a production run uses none of them: its instance (FULL = false) carries neither their branches -- a lone wave
// pays every one -- nor their scalar registers (the kernel spills ~90 of them into lanes of a vector register and reads them back
// with v_readlane at the use).
// DXO: a production instance (FULL = false) that also stores dZ_0 (cl_mlp_args.dZ0_out: the launch behind a peeled first layer).
// NI (round 5): per-image layers (NeuralImageScaler, careless/models/scaling/image.py:66-125: `--image-layers NI` on the default scaler) --
// the NL Dense layers are followed by NI layers whose (w x w) kernel and bias belong to the IMAGE of the observation; same contract as the
// IMGL instances of elbo_mlp.hip (A.imgl / d_imgl / n_imgl / n_images / tile_img: a 128-row tile of the packed layout holds one image).
// A wave walks a CONTIGUOUS range of wave tiles, keeps its current image's weight images (forward + transposed) in its own LDS region and
// the layers' weight-gradient sums in accumulator registers like any other layer's; on an image change it adds the sums to the image's
// gradient (atomics) and reloads.  The activations of the top PK layers wait in LDS across the sampling epilogue (22 x 10 activations +
// the epilogue's state do not fit the 512 registers; the compiler would spill to scratch, which a lone wave waits for in full).
// DEPTH: the unit's NL again, as a template argument -- the instances of the units compiled for other depths need names of their own.
// MODE (round 6): 0 = the fused step; the two launches of a head-less layer block in front of another one (a scaler deeper than one launch:
// include/careless_hip.h, act_out / dH_ext) -- 1 = forward only, the top layer's activations stored feature-major at act_out;
// 2 = backward from an external dL/d(top activations) (dH_ext), the forward pass recomputed, weight-gradient partials as in the fused step
// (no head, no sampling epilogue, nothing per reflection).
template <int W, int DMAX, bool PACKED, bool FULL, bool DXO = false, int NI = 0, int DEPTH = NL, int MODE = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1)))
void elbo_lane_kernel(const cl_mlp_args A) {
    static_assert(DEPTH == NL, "one depth per compilation unit");
    static_assert(MODE == 0 || (!PACKED && !FULL && !DXO && NI == 0 && DMAX != 0), "layer-block launches: plain layout, metadata in registers");
    constexpr bool LX = (DMAX == 0);
    constexpr int DREG = LX ? 1 : DMAX;               // metadata registers of the lane (LX: none; arrays keep one element)
    constexpr int NLT = NL + NI;                      // hidden layers: Dense + per-image
    static_assert(NI == 0 || (PACKED && !LX), "per-image layers: packed layout, metadata in registers");
    using SM = LSmem<W, LX, NI>;
    constexpr int PK = SM::PK;
    constexpr int DGMAX = LX ? 8 : (DMAX + 3) / 4;
    constexpr int TPAR = SM::TPAR;
    constexpr int NC = SM::NC;
    constexpr int IMG = SM::IMG;
    // LeakyReLU derivative: compares into scalar-register pairs in groups of SELG, then the group's selects (equal groups of at
    // most five: a scalar register written by a vector instruction wants two instructions before a vector instruction reads it)
    constexpr int SELG = (W + (W + 4) / 5 - 1) / ((W + 4) / 5);
    static_assert(W >= 1 && W <= 15, "block 15 of a weight register is the bias");
    static_assert(CL_MLP_TILE % WT == 0, "a wave tile must not straddle the end of the padded observation axis");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;   // a previous step hit a non-finite gradient norm
#ifdef CL_STAMPS
    unsigned long long st_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0)::"memory");
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = uniform(tid >> 6);
    const int d = A.d, w = A.w, L = A.L;
    const float leak = A.leak;

    // ---- weight images: register `c` of layer l holds, in lane 4 b + i, W_l[4 c + i][b] (b = 15: the bias); the dgrad image the
    //      transposed weights W_l[b][4 c + i] ---------------------------------------------------------------------------------
    {
        const float* __restrict__ P = A.mlp;
        // One "row" of the images = the 64 lanes of one register (image, chunk): rows * 64 floats in all, row rho at smem[64 rho].
        // Wave `sub` of the workgroup takes rows sub, sub + NWV, ...; what depends on the lane -- block b, feature-in-chunk -- is
        // computed once, what depends on the row is wave-uniform.  No branch around a load: every lane loads from a valid address
        // (offset 0 when its element is padding) and selects afterwards -- with a branch per condition the compiler built ~150
        // exec-mask regions here, a third of the launch's fixed cost for this one-wave-per-SIMD kernel.
        constexpr int NROW = (SM::NF + NL + 1) * NC, NIT = (NROW + NWV - 1) / NWV;
        static_assert(NT == 64 * NWV, "one wave per image row");
        float v_[NIT];
        const int offWo = w * d + w + (NL - 1) * (w * w + w);         // (cl_lane_supports: the scaler has exactly NL layers)
        const int b = lane >> 2, fq = lane & 3;
        // (all loads of a thread are issued before the first LDS store: see elbo_narrow.hip)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rho = it * NWV + wv;                            // wave-uniform
            const int img = rho / NC, c = rho - img * NC;
            const int tr = img >= SM::NF ? 1 : 0, l = img - tr * SM::NF;
            const int f = 4 * c + fq;
            const int in_dim = (l == 0) ? d : w;
            const int base = (l == 0) ? 0 : (w * d + w + (l - 1) * (w * w + w));
            int off;
            bool ok;
            if (tr == 0) {                                            // (wave-uniform selects: scalar code)
                if (LX && l == NL + 1) {                              // layer 0, second input block: block b = metadata column 15 + b
                    off = f * d + 15 + b;
                    ok = f < w && 15 + b < d;
                } else if (l < NL) {
                    off = b == ONE ? base + w * in_dim + f : base + f * in_dim + b;
                    ok = f < w && (b == ONE || b < in_dim);
                } else {                                              // the Dense(2) head (a head-less block has none)
                    off = b == ONE ? offWo + 2 * w + f : offWo + f * w + b;
                    ok = MODE == 0 && f < 2 && (b == ONE || b < w);
                }
            } else {
                if (l < NL) {
                    off = base + b * w + f;
                    ok = l > 0 && b < w && f < w;
                } else {
                    off = offWo + b * w + f;
                    ok = MODE == 0 && b < 2 && f < w;
                }
            }
            ok = ok && rho < NROW;
            const float v = P[ok ? off : 0];
            v_[it] = ok ? v : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rho = it * NWV + wv;
            if (rho < NROW) smem[SM::oF + rho * 64 + lane] = v_[it];
        }
        // staging tiles: rows that are never written hold their constants (zero; row 15 of the input tiles = the ones)
        static_assert(SM::oT % 4 == 0 && SM::TWF % 4 == 0 && PIT % 4 == 0 && SM::oS % PIT == 0, "16-byte fill");
        const int TWr = SM::tw(d), xrows = SM::xrows(d);
        // (wave region by wave region: no division by the run-time region size)
        for (int wr = 0; wr < NWV; ++wr) {
            float* const reg = smem + SM::oT + wr * TWr;
            for (int off = 4 * tid; off < TWr; off += 4 * NT) {
                bool one;
                if (off < SM::oS) {
                    const int row = off / PIT;                          // sZ[TPAR], sH[TPAR], (not LX) sX: 16 rows each
                    one = row >= TPAR * 16 && (row & 15) == ONE;
                } else {
                    const int xr = (off - SM::oX) / PIT;                // LX: the two row buffers (negative before them: sS / sQ / sE)
                    one = LX && off >= SM::oX && (xr == ONE || xr == xrows + ONE);
                }
                const float v = one ? 1.0f : 0.0f;
                *reinterpret_cast<f32x4*>(reg + off) = f32x4{v, v, v, v};
            }
        }
        if (LX) {                                                       // (tail that the upper-block reads of the last wave's second buffer touch)
            for (int idx = 4 * tid; idx < (32 - xrows) * PIT; idx += 4 * NT) *reinterpret_cast<f32x4*>(smem + SM::oT + NWV * TWr + idx) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    __syncthreads();

    const float* const sF = smem + SM::oF + lane;      // forward weight register c of layer l: sF[l * IMG + c * 64]
    const float* const sK = smem + SM::oK + lane;
    const int xrows = SM::xrows(d);                    // LX: rows of a metadata buffer
    float* const sZ = smem + SM::oT + wv * SM::tw(d);  // dZ_l                       [feature][observation]
    float* const sH = sZ + TPAR * 16 * PIT;            // the layer's input (layers >= 1)
    float* const sX = sH + TPAR * 16 * PIT;            // (not LX) the metadata of the tile (layer 0's input)
    float* const sS = sZ + SM::oS;                     // sampled amplitudes of a batch of SPRE samples, then their gradients   [sample][lane]
    unsigned* const sQ = reinterpret_cast<unsigned*>(sS + SPRE * 64);       // byte offset of the lane's reflection in dz_f (~0: none)
    float* const sE = sS + (SPRE + 1) * 64;            // the standard normals of the lane's observation for the samples of a batch   [sample][lane]
    float* const sXb = sZ + SM::oX;                    // LX: metadata rows of the current / the next tile: buffer (tile parity) x [xrows][PIT]
    float* const sFi = sZ + SM::oI + lane;             // NI: forward weight register c of image layer li: sFi[li * IMG + c * 64]
    float* const sKi = sFi + NI * IMG;                 //     transposed
    float* const sP = sZ + SM::oP + lane;              // NI: activation f of parked layer p (= layer NLT - PK + p): sP[(p * W + f) * 64]
    // transposed weight image of hidden layer l (l == NLT: the head)
    auto kimg = [&](auto lc) -> const float* {
        constexpr int l = decltype(lc)::value;
        if constexpr (l < NL) return sK + l * IMG;
        else if constexpr (l < NLT) return sKi + (l - NL) * IMG;
        else return sK + NL * IMG;
    };
    constexpr int PAR = (TPAR - 1) * 16 * PIT;         // second copy of sZ / sH (layers alternate between the two; LX: one copy)

    // ---- accumulators that live across all tiles of this wave ----------------------------------------------------------
    f32x4 wacc[NLT], wacd[NLT];         // dW_l = wacc + wacd: lane (j, q), element t = dW_l[out 4 q + t][in j]  (in 15: the bias)
#pragma unroll
    for (int l = 0; l < NLT; ++l) { wacc[l] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; wacd[l] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    f32x4 wacc0b = {0.0f, 0.0f, 0.0f, 0.0f}, wacd0b = {0.0f, 0.0f, 0.0f, 0.0f};      // LX: layer 0's second input block (columns 15 .. 30)
    f32x2 hacc[W + 1];                  // head: per-lane sums of (dloc, draw) x top activation k; [W]: the bias
#pragma unroll
    for (int k = 0; k <= W; ++k) hacc[k] = f32x2{0.0f, 0.0f};
    float nll_acc = 0.0f;
    cl_ev11 ev = {1.0f, 0.0f, 0.0f};
    float ev_g0 = 0.0f, ev_g1 = 0.0f, ev_g2 = 0.0f;
    const bool use_ev11 = FULL && A.ev11 != nullptr;
    const bool has_eta = FULL && A.eta != nullptr;          // injected scale noise (parity tests)
    const bool det = FULL && A.dzf_obs != nullptr;          // deterministic mode: stores per (observation, sample) instead of float atomics
    // dL/d(pre-activations of layer 0) out (round 5: the launch behind a peeled first layer, cl_peel_*: a row per feature,
    // [cl_mlp_meta_rows(w)][n_pad] like meta_t)
    const bool has_dxo = (FULL || DXO) && A.dZ0_out != nullptr;
    if (use_ev11) { ev.sdfac = cl_softplus(A.ev11[0]); ev.sdadd = cl_softplus(A.ev11[1]); ev.sdb = cl_softplus(A.ev11[2]); }

    const int n_wt = (A.n_obs + WT - 1) / WT;                        // wave tiles
    const int wt_step = (int)gridDim.x * NWV;
    const int dg = (d + 3) >> 2;                                     // 4-row groups of the metadata

    // per-observation inputs of a wave tile, loaded one tile ahead (a wave tile never leaves the padded metadata rows because 64
    // divides CL_MLP_TILE, the per-observation arrays are clamped to their last element)
    float xn[DREG];
#pragma unroll
    for (int k = 0; k < DREG; ++k) xn[k] = 0.0f;
    int ridn = -1, imgn = 0;
    float ion = 0.0f, sgn = 1.0f;
    // `xbuf`: LX only -- the row buffer the tile's metadata go to
    auto prefetch = [&](int wt_in, int xbuf) {
        const int wt = uniform(wt_in);
        const int base = wt * WT;
        const unsigned n_pad_u = (unsigned)A.n_pad;
        const int last_obs = A.n_obs - 1;
        const float* __restrict__ mt = A.meta_t + base;
        const int dd = A.d;
        if constexpr (LX) {
            // rows of meta_t (cl_mlp_meta_rows(d) of them: whole groups of four, zero past d) -> rows of the buffer, by the DMA path:
            // lane i of a row instruction fetches observation i, the data land at (row base) + 4 i
            float* const dst = sXb + uniform(xbuf) * xrows * PIT;
            static_for<0, DGMAX>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (4 * g < dd) {                                        // wave-uniform
                    static_for<4 * g, (4 * g + 4 < DMAX_LX ? 4 * g + 4 : DMAX_LX)>([&](auto kc) {
                        constexpr int k = decltype(kc)::value, r = k < ONE ? k : k + 1;
                        __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(mt + (size_t)k * n_pad_u) + 4u * (unsigned)lane,
                                                         (__attribute__((address_space(3))) void*)(dst + r * PIT), 4, 0, 0);
                    });
                }
            });
        } else {
#pragma unroll
            for (int g = 0; g < DGMAX; ++g) {
                if (4 * g < dd) {                                            // wave-uniform
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int k = 4 * g + t;
                        if (k < DMAX) xn[k] = ld_uo(mt + (size_t)k * n_pad_u, 4u * (unsigned)lane);    // (rows d .. 4 dg - 1: zeroed at the use)
                    }
                }
            }
        }
        // (no arithmetic on the loaded values here: it would wait for them; observations past the end are masked at the use)
        if constexpr (MODE == 0) {
            const unsigned ob = 4u * (unsigned)min(base + lane, last_obs);
            ridn = ld_uo(A.refl_id, ob);
            ion = ld_uo(A.iobs, ob);
            sgn = ld_uo(A.sig, ob);
            imgn = A.use_img ? ld_uo(A.image_id, ob) : 0;
        }
    };
    // wave tiles of this wave: strided over all waves of the launch, or (NI) one contiguous range so that image changes are rare
    const int gwv = (int)blockIdx.x * NWV + wv;
    int wt_begin = NI > 0 ? (int)((long long)gwv * n_wt / wt_step) : gwv;
    int wt_end = NI > 0 ? (int)((long long)(gwv + 1) * n_wt / wt_step) : n_wt;
    const int wt_inc = NI > 0 ? 1 : wt_step;
    if constexpr (NI > 0 && FULL) {
        // Deterministic mode (round 6): both borders of the range move up to the next image border (tile_img ascends: obs.pack_by_image /
        // pack_laue), so that ONE wave holds all tiles of an image and its per-image gradient leaves as one addition per element onto the
        // cleared buffer -- no order for the float atomics to depend on.  (The same function of the even split on both sides: the ranges
        // still partition the tiles.)
        if (det) {
            auto img_of = [&](int t) { return uniform(A.tile_img[(t * WT) / CL_MLP_TILE]); };
            auto align = [&](int t) -> int {
                if (t <= 0 || t >= n_wt) return t < n_wt ? t : n_wt;
                const int prev = img_of(t - 1);
                if (img_of(t) != prev) return t;
                int lo = t, hi = n_wt;               // first wave tile past t whose image is not `prev`
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (img_of(mid) != prev) hi = mid; else lo = mid + 1;
                }
                return lo;
            };
            wt_begin = align(wt_begin);
            wt_end = align(wt_end);
        }
    }
    if (wt_begin < wt_end) prefetch(wt_begin, 0);

    // ---- per-image layers: this wave's gradient flush and weight reload on an image change ------------------------------------------
    int cur_img = -1;
    const size_t imgl_blk = NI > 0 ? (size_t)A.n_images * (size_t)(w * w + w) : 0;      // floats per image layer: [W: n_images x (w x w) | b: n_images x w]
    auto imgl_flush = [&](int im) {
        asm volatile("s_nop 15\n\ts_nop 15");      // (inline-assembly MFMAs: results land eight passes after the issue)
#pragma unroll
        for (int li = 0; li < NI; ++li) {
            float* __restrict__ gW = A.d_imgl + (size_t)li * imgl_blk + (size_t)im * (size_t)(w * w);
            float* __restrict__ gB = A.d_imgl + (size_t)li * imgl_blk + (size_t)A.n_images * (size_t)(w * w) + (size_t)im * w;
            const f32x4 v = wacc[NL + li] + wacd[NL + li];
            const int fi = lane & 15;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int fo = 4 * (lane >> 4) + t;
                if (fo < w && fi < w) atomicAdd(gW + fo * w + fi, v[t]);
                if (fo < w && fi == ONE) atomicAdd(gB + fo, v[t]);
            }
            wacc[NL + li] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            wacd[NL + li] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    auto imgl_load = [&](int im) {
        const int b = lane >> 2, fq = lane & 3;
#pragma unroll
        for (int li = 0; li < NI; ++li) {
            const float* __restrict__ Wg = A.imgl + (size_t)li * imgl_blk + (size_t)im * (size_t)(w * w);
            const float* __restrict__ Bg = A.imgl + (size_t)li * imgl_blk + (size_t)A.n_images * (size_t)(w * w) + (size_t)im * w;
            float vf[NC], vk[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int f = 4 * c + fq;
                const bool okf = f < w && (b == ONE || b < w), okk = b < w && f < w;
                const float xf = (b == ONE) ? Bg[f < w ? f : 0] : Wg[okf ? f * w + b : 0];
                const float xk = Wg[okk ? b * w + f : 0];
                vf[c] = okf ? xf : 0.0f;
                vk[c] = okk ? xk : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                sFi[li * IMG + c * 64] = vf[c];
                sKi[li * IMG + c * 64] = vk[c];
            }
        }
    };

    const float ones = 1.0f;
    const int rr16 = lane & 15, kq = lane >> 4;
    const float* const rdZ = sZ + rr16 * PIT + 4 * kq;       // wgrad operands: row (lane & 15), observations 16 c + 4 kq .. + 3
    const float* const rdH = sH + rr16 * PIT + 4 * kq;
    const float* const rdX = (LX ? sXb : sX) + rr16 * PIT + 4 * kq;      // (LX: + the current buffer)

#ifdef CL_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
    st_acc[6] = st_last - st_t0;                                  // launch prologue: weight images, tile fill, first prefetch issue
#endif
    int xcur = 0;                                             // LX: buffer of the current tile's metadata rows
    for (int wt = wt_begin; wt < wt_end; wt += wt_inc, xcur ^= 1) {
        if constexpr (NI > 0) {
            const int im = uniform(A.tile_img[(wt * WT) / CL_MLP_TILE]);
            if (im != cur_img) {                 // wave-uniform
                if (cur_img >= 0) imgl_flush(cur_img);
                imgl_load(im);
                cur_img = im;
            }
        }
        float x0[DREG];
#pragma unroll
        for (int k = 0; k < DREG; ++k) x0[k] = xn[k];          // (rows >= d of meta_t are zero by contract, groups past them were never loaded)
        // layer 0's input, staged for its weight gradient at the end of the backward pass
        if constexpr (!LX) {
#pragma unroll
            for (int g = 0; g < DGMAX; ++g) {
                if (g < dg) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (4 * g + t < DMAX) sX[(4 * g + t) * PIT + lane] = x0[4 * g + t];
                }
            }
        }
        const int xoff = LX ? uniform(xcur) * xrows * PIT : 0;   // the current tile's row buffer
        const bool in_range = wt * WT + lane < A.n_obs;
        const int rid = in_range ? ridn : -1, img = imgn;
        const float io = ion, sg = in_range ? sgn : 1.0f;
        // Gathers that depend on the prefetched ids: issued now, consumed in the epilogue.  Every lane loads from a valid (clamped)
        // address and nothing is done with the values here: a select or a merge under a divergent branch would make this lone wave
        // wait for each gather in turn.
        float aim_raw = 1.0f, zf0 = 0.0f;
        int rme = 0, gm = 0;                 // PACKED: the caller's row of this lane's packed row; (member index | group size << 8)
        long long nkey = 0;                  // PACKED: noise key of this lane's row
        if constexpr (MODE == 0) {
            const unsigned zb = 4u * (unsigned)max(rid, 0) * (unsigned)A.S;
            if (A.use_img) aim_raw = ld_uo(A.img, 4u * (unsigned)max(img - 1, 0));
            zf0 = ld_uo(A.z_f, zb);
            // samples 1 .. SPRE-1 of this lane's reflection go straight to LDS (global_load_lds: no registers in between; a lone
            // wave cannot afford a gather per sample inside the sample loop)
            if (A.S > 1) {
#pragma unroll
                for (int j = 1; j < SPRE; ++j)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(A.z_f) + (zb + 4u * (unsigned)min(j, A.S - 1)),
                                                     (__attribute__((address_space(3))) void*)(sS + j * 64), 4, 0, 0);
            }
            if (PACKED) {
                const unsigned pb = 4u * (unsigned)(wt * WT + lane);
                rme = ld_uo(A.row_map, pb);
                if (A.gmeta != nullptr) gm = ld_uo(A.gmeta, pb);
                nkey = (A.noise_row != nullptr) ? (long long)ld_uo(A.noise_row, pb) : A.obs_offset + rme;
            } else if (A.noise_row != nullptr) {
                // plain layout over rows that are not a contiguous range of the caller's (reflection-owner shard): the row's global number
                nkey = (long long)ld_uo(A.noise_row, 4u * (unsigned)(wt * WT + lane));
            }
        }
        // deterministic mode with caller-assigned record slots (det_slot: reflection order): this lane's slot -- a plain-layout lane's
        // row is its position, a packed one's comes from row_map (rme), both known here for in-range rows only
        int dslot_raw = 0;
        if (det && A.det_slot != nullptr && !PACKED) dslot_raw = ld_uo(A.det_slot, 4u * (unsigned)min(wt * WT + lane, A.n_obs - 1));
        LSTAMP(0);
        // ================= forward ==========================================================================================
        // (activations and dZ in aligned register pairs: the packed fp32 instructions of the backward pass take them as they stand)
        static_assert(W % 2 == 0, "feature pairs");
        f32x2 hsp[NLT][W / 2];
#define HS(l, f) hsp[l][(f) >> 1][(f) & 1]
        float wan[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) wan[c] = sF[c * 64];
        // (The Dense layers in one loop, the per-image layers in a loop of their own behind it: round 5 kept them apart because a single
        //  static_for over all NLT layers "made" the dZ_0-storing instance unrepeatable -- it had only moved the inline-assembly LeakyReLU
        //  next to an MFMA, NOTEBOOK R6.1; the split stays because the instruction streams are the measured ones.)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            {
                float wa[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) wa[c] = wan[c];
                // the next layer's weight registers, in flight under this layer's MFMAs (after the top layer: the head's)
#pragma unroll
                for (int c = 0; c < NC; ++c) wan[c] = (NI > 0 && l + 1 == NL) ? sFi[c * 64] : sF[(l + 1) * IMG + c * 64];
                f32x4 acc[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (l == 0 && LX) {
                    // metadata column k: row k (k < 15) or k + 1 of the tile's row buffer, block k of the first / block k - 15 of the
                    // second weight register of a chunk; whole groups of four rows (the ones past d are zero)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the rows were requested a backward pass ago
                    const float* const xr = sXb + xoff + lane;
                    float wb[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) wb[c] = sF[(NL + 1) * IMG + c * 64];
                    static_for<0, DGMAX>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        if (g < dg) {                                    // wave-uniform
                            constexpr int k1 = (4 * g + 4 < DMAX_LX ? 4 * g + 4 : DMAX_LX);
                            float xv[4];
                            static_for<4 * g, k1>([&](auto kc) {
                                constexpr int k = decltype(kc)::value;
                                xv[k & 3] = xr[(k < ONE ? k : k + 1) * PIT];
                            });
                            static_for<4 * g, k1>([&](auto kc) {
                                constexpr int k = decltype(kc)::value;
#pragma unroll
                                for (int c = 0; c < NC; ++c) acc[c] = mfma_bk<(k < ONE ? k : k - ONE)>(k < ONE ? wa[c] : wb[c], xv[k & 3], acc[c]);
                            });
                        }
                    });
                } else if (l == 0) {
                    static_for<0, DGMAX>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        if (g < dg) {                                    // wave-uniform
                            static_for<4 * g, (4 * g + 4 < DREG ? 4 * g + 4 : DREG)>([&](auto kc) {
                                constexpr int k = decltype(kc)::value;
#pragma unroll
                                for (int c = 0; c < NC; ++c) acc[c] = mfma_bk<k>(wa[c], x0[k], acc[c]);
                            });
                        }
                    });
                } else {
                    static_for<0, W>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
#pragma unroll
                        for (int c = 0; c < NC; ++c) acc[c] = mfma_bk<k>(wa[c], HS(l > 0 ? l - 1 : 0, k), acc[c]);
                    });
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = mfma_bk<ONE>(wa[c], ones, acc[c]);
                {
                    f32x4 lk[NC];                    // leak * z, two elements per multiply
#pragma unroll
                    for (int c = 0; c < NC; ++c) lk[c] = acc[c] * leak;
#pragma unroll
                    for (int f = 0; f < W; ++f) HS(l, f) = lrelu2(acc[f >> 2][f & 3], lk[f >> 2][f & 3]);
                }
                if (PK > 0 && l >= NLT - PK) {       // parked: the registers are free from the head on, the backward pass reads the copy
#pragma unroll
                    for (int f = 0; f < W; ++f) sP[((l - (NLT - PK)) * W + f) * 64] = HS(l, f);
                }
            }
        }
        if constexpr (NI > 0) {
            // the per-image layers: the same chain on the wave's own weight images (after the last one: the head's, as above)
#pragma unroll
            for (int li = 0; li < NI; ++li) {
                const int l = NL + li;
                float wa[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) wa[c] = wan[c];
#pragma unroll
                for (int c = 0; c < NC; ++c) wan[c] = (li + 1 < NI) ? sFi[(li + 1) * IMG + c * 64] : sF[NL * IMG + c * 64];
                f32x4 acc[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                static_for<0, W>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[c] = mfma_bk<k>(wa[c], HS(l - 1, k), acc[c]);
                });
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = mfma_bk<ONE>(wa[c], ones, acc[c]);
                f32x4 lk[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) lk[c] = acc[c] * leak;
#pragma unroll
                for (int f = 0; f < W; ++f) HS(l, f) = lrelu2(acc[f >> 2][f & 3], lk[f >> 2][f & 3]);
                if (l >= NLT - PK) {
#pragma unroll
                    for (int f = 0; f < W; ++f) sP[((l - (NLT - PK)) * W + f) * 64] = HS(l, f);
                }
            }
        }
#define TOP(k) HS(NLT - 1, k)                            /* the head's input */
        LSTAMP(1);
        float dloc = 0.0f, draw = 0.0f;
        if constexpr (MODE == 1) {
            // a head-less block, forward only: the top layer's activations out (feature-major like meta_t: the next block's "metadata")
            float* const ao = A.act_out + (size_t)(wt * WT) + lane;
            const size_t np = (size_t)A.n_pad;
            static_for<0, W>([&](auto fc_) {
                constexpr int f = decltype(fc_)::value;
                if (f < w) ao[f * np] = TOP(f);
            });
            if (wt + wt_inc < wt_end) prefetch(wt + wt_inc, xcur ^ 1);
            continue;
        }
        if constexpr (MODE == 0) {
        // Dense(2) head: outputs 0, 1 of one more chunk
        float o0, o1;
        {
            const float wh = sF[NL * IMG];
            f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
            static_for<0, W>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                a = mfma_bk<k>(wh, TOP(k), a);
            });
            a = mfma_bk<ONE>(wh, ones, a);
            o0 = a[0];
            o1 = a[1];
        }

        // ================= epilogue: sample, predict, likelihood, dL/d(loc, raw) of this lane's observation ==================
        // (the kernel arguments stay in scalar registers / register lanes for the whole launch: re-reading them from the kernarg
        //  segment at the point of use, as the two-waves-per-SIMD kernels do, would leave a lone wave waiting ~150 cycles per read)
        const int S = A.S;
        const float w_ll = A.w_ll;
        const float aim = (A.use_img && img > 0 && rid >= 0) ? aim_raw : 1.0f;
        const long long gobs = PACKED ? (long long)rme : (long long)wt * WT + lane;      // this lane's observation in the caller's order
        const unsigned zoff = 4u * (unsigned)(rid < 0 ? 0 : rid) * (unsigned)S;
        // record of this lane's observation in dzf_obs (deterministic mode)
        const unsigned drec = (det && A.det_slot != nullptr) ? (unsigned)(PACKED ? (rid >= 0 ? A.det_slot[gobs] : 0) : dslot_raw) : (unsigned)gobs;
        float dsig_draw;
        const float sigma = cl_scale_bij(o1, A.bij_kind, A.eps, &dsig_draw);
        float pdl = 0.0f, pds = 0.0f, pda = 0.0f;
        // single-pass Laue (careless/models/likelihoods/laue.py:20-34): the predictions of the rows of one harmonic group SUM before
        // the likelihood.  The members of a group are consecutive lanes; every lane collects its group's total with shuffles over a
        // wave-uniform member count, every member evaluates the same likelihood derivative, member 0 alone counts the log-likelihood.
        const bool laue = PACKED && A.gmeta != nullptr;                    // wave-uniform
        const int mem = gm & 0xff, cnt = gm >> 8;
        const int gmax = laue ? uniform(A.tile_gmax[(wt * WT) / CL_MLP_TILE]) : 0;
        // The S amplitude gradients of an observation are S consecutive floats of dz_f.  Issued as they are computed -- one atomic
        // instruction per sample, 64 lanes on 64 different lines, the same lines again a sample later -- they cost this kernel 0.12 ms
        // per sample at 4 M observations (the memory side executes one request per lane and serialises the ones that hit a line in
        // flight).  From two samples on they go through LDS instead and leave as eight instructions in which eight consecutive lanes
        // carry the eight samples (of a batch) of one observation: one 32-byte request per observation.
        const bool coal = CL_LANE_COALESCE && S >= CL_LANE_COALESCE;      // wave-uniform
        // The samples go in batches of SPRE (one batch unless S > SPRE): a batch's amplitudes wait in LDS (the first batch's were
        // gathered at the start of the tile), its amplitude gradients leave through it.
        // hardware reciprocal and logarithm (1 ulp): sigma is an input, its log enters the NLL additively
        const float inv_sg = cl_fast_rcp(sg);
        const float log_sg = cl_fast_log(sg);
        // (per-lane 64-bit addresses of the optional arrays are formed where they are used: test / output paths only)
        float* __restrict__ dzf_p = A.dz_f;
        const int lik_kind = A.lik_kind;
        const float dof = A.dof, lik_const = A.lik_const, shift = A.shift;
        const float inv_dof = (lik_kind == CL_LIK_STUDENTT) ? 1.0f / dof : 0.0f;       // wave-uniform
        const bool act = rid >= 0;
        // (The amplitudes wait in LDS: inside the loop every wait on a global load would also wait for the previous sample's
        //  atomics -- one in-order counter --, a few microseconds each for a lone wave.)
        // one MC sample of this lane's observation, given its noise and its sampled amplitude (all lanes take part in the Laue shuffles)
        auto sample = [&](int s, float eta, float zf) {
            const float tq = o0 + sigma * eta + shift;
            const float zs = aim * tq;
            const float ipred = act ? zs * zf * zf : 0.0f;
            if (FULL && A.ipred_out != nullptr && act) A.ipred_out[(size_t)gobs * S + s] = ipred;
            float lin = ipred;                                   // what the likelihood sees: the prediction, or its group's total
            if (laue) {
                lin = 0.0f;
                for (int mm = 0; mm < gmax; ++mm) {
                    const float v = __shfl(ipred, (lane - mem + mm) & 63);
                    lin += (mm < cnt) ? v : 0.0f;
                }
            }
            if (act) {
                const bool counts = !laue || mem == 0;
                float dll, ll;
                if (use_ev11) {
                    float gf, gb, ga;
                    ll = cl_lik_ev11(lin, io, sg, lik_kind, dof, lik_const, ev, &dll, &gf, &gb, &ga);
                    if (counts) { ev_g0 -= gf * w_ll; ev_g1 -= ga * w_ll; ev_g2 -= gb * w_ll; }     // order: Sdfac, Sdadd, SdB
                } else {
                    // (Student-T: 1/nu hoisted, the per-sample division as reciprocal + Newton step -- a lone wave pays ~8 cycles per
                    //  instruction of the two IEEE divisions)
                    ll = cl_lik_log_prob3(lin, io, inv_sg, log_sg, lik_kind, dof, inv_dof, lik_const, &dll);
                }
                if (counts) nll_acc -= ll * w_ll;
                const float gi = -dll * w_ll;                 // dNLL / d ipred (of every member of the group)
                const float dzs = gi * zf * zf;
                if (coal) sS[(s & (SPRE - 1)) * 64 + lane] = gi * zs * 2.0f * zf;      // (the sample's slot: its amplitude was read at the start of this sample)
                else if (det) *reinterpret_cast<float*>(reinterpret_cast<char*>(A.dzf_obs) + 4u * (drec * (unsigned)S + (unsigned)s)) = gi * zs * 2.0f * zf;
                else atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(dzf_p) + zoff + 4u * s), gi * zs * 2.0f * zf);
                const float dt = dzs * aim;
                pdl += dt;
                pds += dt * eta;
                pda += dzs * tq;
            }
        };
        // (deterministic mode: the observation's own record in dzf_obs instead of its reflection's row of dz_f)
        if (coal) sQ[lane] = (rid >= 0) ? (det ? 4u * drec * (unsigned)S : zoff) : 0xFFFFFFFFu;
        int sb = 0;                                      // first sample of the batch
        do {
            const int se = (sb + SPRE < S) ? sb + SPRE : S;
            if (sb > 0) {
                // a later batch (S > SPRE): its amplitudes, gathered here (every lane, clamped addresses); the previous batch's
                // gradients have left the slots (their reads fed atomics that are issued)
#pragma unroll
                for (int j = 0; j < SPRE; ++j)
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(A.z_f) + (zoff + 4u * (unsigned)min(sb + j, S - 1)),
                                                     (__attribute__((address_space(3))) void*)(sS + j * 64), 4, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (!has_eta) {
                // the batch's in-kernel noise, drawn up front into LDS: one Philox block + Box-Muller pair serves samples s and s + 4
                // (cl_math.h).  Drawn inside the sample loop, the pair's second half had to be parked in registers selected by
                // s & 3 -- a dozen scalar branches per sample, each paid in full by a lone wave
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (sb + p < S) {                            // wave-uniform
                        float ec, es;
                        cl_noise_normal_pair(A.seed, A.step, (uint32_t)(sb + p), (uint64_t)((PACKED || A.noise_row != nullptr) ? nkey : A.obs_offset + gobs), &ec, &es);
                        sE[p * 64 + lane] = ec;
                        sE[(p + 4) * 64 + lane] = es;
                    }
                }
            }
            if (laue || rid >= 0) {
                // Two loops over the batch (wave-uniform trip counts), so that the common one -- in-kernel noise -- contains no
                // global load at all: any load in the loop makes the compiler wait on the one in-order memory counter, i.e.
                // for the previous sample's atomics, a few microseconds each for a lone wave.
                if (has_eta) {
                    const float* __restrict__ eta_p = A.eta + (size_t)gobs * S;
                    for (int s = sb; s < se; ++s) {
                        const float zf = (s > 0) ? sS[(s & (SPRE - 1)) * 64 + lane] : zf0;
                        sample(s, act ? eta_p[s] : 0.0f, act ? zf : 0.0f);
                    }
                } else {
                    for (int s = sb; s < se; ++s) {
                        const float zf = (s > 0) ? sS[(s & (SPRE - 1)) * 64 + lane] : zf0;
                        sample(s, sE[(s & (SPRE - 1)) * 64 + lane], act ? zf : 0.0f);
                    }
                }
            }
            if (coal) {
                const int ss = lane & 7, oj = lane >> 3;             // this lane's sample of the batch, its observation within a group of eight
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned zq = sQ[8 * j + oj];
                    const float g = sS[ss * 64 + 8 * j + oj];
                    if (sb + ss < S && zq != 0xFFFFFFFFu) {
                        if (det) *reinterpret_cast<float*>(reinterpret_cast<char*>(A.dzf_obs) + zq + 4u * (unsigned)(sb + ss)) = g;
                        else atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(A.dz_f) + zq + 4u * (unsigned)(sb + ss)), g);
                    }
                }
            }
            sb += SPRE;
        } while (sb < S);
        if (A.use_img && det) {
            if (rid >= 0) A.dimg_obs[gobs] = pda;                        // summed per image, in row order, by cl_det_reduce
        } else if (A.use_img) {
            // image ids are sorted: the observations of a wave tile almost always share one image -> ONE atomic per wave
            const int img0 = uniform(img);
            if (__all(img == img0 || rid < 0)) {
                const float v = cl_wave_sum(rid >= 0 ? pda : 0.0f);
                if (lane == 0 && img0 > 0) atomicAdd(A.d_img + (img0 - 1), v);
            } else {
                cl_image_grad_segments(A.d_img, img, pda, rid >= 0 && img > 0, lane);
            }
        }
        dloc = pdl;
        draw = pds * dsig_draw;              // zero for padding observations
        }   // MODE == 0

        LSTAMP(2);
        // next tile's inputs: their latency hides under the backward pass
        if (wt + wt_inc < wt_end) prefetch(wt + wt_inc, xcur ^ 1);
        LSTAMP(3);

        // ================= backward =========================================================================================
        // NI: a parked layer's activations come back from LDS right before their first use on the way down (through an offset the
        // compiler cannot see through: it would otherwise keep the registers across the epilogue instead of the copy)
        auto unpark = [&](auto lc) {
            constexpr int l = decltype(lc)::value;
            if constexpr (PK > 0 && l >= NLT - PK && l < NLT) {
                int zero = 0;
                asm volatile("" : "+v"(zero));
                const float* const sp = sP + zero;
#pragma unroll
                for (int f = 0; f < W; ++f) HS(l, f) = sp[((l - (NLT - PK)) * W + f) * 64];
            }
        };
        unpark(std::integral_constant<int, NLT - 1>{});
        unpark(std::integral_constant<int, (NLT >= 2 ? NLT - 2 : 0)>{});
        f32x4 dH[NC];
        if constexpr (MODE == 2) {
            // a head-less block: dL/d(top activations) comes from the block behind it (feature-major, like act_out)
            const float* const dh = A.dH_ext + (size_t)(wt * WT) + lane;
            const size_t np = (size_t)A.n_pad;
#pragma unroll
            for (int c = 0; c < NC; ++c) dH[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            static_for<0, W>([&](auto fc_) {
                constexpr int f = decltype(fc_)::value;
                if (f < w) dH[f >> 2][f & 3] = in_range ? dh[f * np] : 0.0f;
            });
        } else {
        // head: its weight gradient is per-lane sums; its dgrad two steps (dloc, draw) per input chunk
        {
            const f32x2 dd2 = {dloc, draw};
#pragma unroll
            for (int k = 0; k < W; ++k) hacc[k] += dd2 * f32x2{TOP(k), TOP(k)};
            hacc[W] += dd2;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float wk = sK[NL * IMG + c * 64];
            f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
            a = mfma_bk<0>(wk, dloc, a);
            a = mfma_bk<1>(wk, draw, a);
            dH[c] = a;
        }
        }
        LSTAMP(4);
            // dZ of a layer: dH where the activation is positive, leak dH otherwise.  As the compiler writes the select (compare
            // into VCC, wait states, select, per element) a lone wave pays ~9 cycles per instruction; compares into scalar
            // register pairs first and the selects after them issue back to back (scripts/probe/pkfma_probe.hip: 5.4 cycles).
            auto dz_of = [&](f32x2 (&dzp)[W / 2], const f32x2 (&hp)[W / 2], const f32x4 (&dh)[NC]) {
                f32x4 lk[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) lk[c] = dh[c] * leak;
#ifdef CL_LANE_SEL_C
#pragma unroll
                for (int f = 0; f < W; ++f) dzp[f >> 1][f & 1] = hp[f >> 1][f & 1] > 0.0f ? dh[f >> 2][f & 3] : lk[f >> 2][f & 3];
                if (true) return;
#endif
#pragma unroll
                for (int f0 = 0; f0 < W; f0 += SELG) {
                    unsigned long long m[SELG];
#pragma unroll
                    for (int i = 0; i < SELG; ++i)
                        if (f0 + i < W) asm volatile("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(m[i]) : "v"(hp[(f0 + i < W ? f0 + i : 0) >> 1][(f0 + i < W ? f0 + i : 0) & 1]));
#pragma unroll
                    for (int i = 0; i < SELG; ++i) {
                        const int f = f0 + i < W ? f0 + i : 0;
                        if (f0 + i < W)
                            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(dzp[f >> 1][f & 1]) : "v"(lk[f >> 2][f & 3]), "v"(dh[f >> 2][f & 3]), "s"(m[i]));
                    }
                }
            };
            // LDS operation number i of layer ll: staging writes (dZ, the layer's input: feature pairs) into the tiles of the
            // layer's parity, then the transposed reads of its weight-gradient operands; past those, the dgrad weights of the
            // layer below.  One wave's LDS operations execute in order and a wave only touches its own tiles.
            constexpr int WP = (W + 1) / 2, NOPS = 2 * WP + 8;
            f32x4 pa[2][4], pb[2][4];
            f32x2 dzp[2][W / 2];
#define DZ(q, f) dzp[q][(f) >> 1][(f) & 1]
            float wk[2][NC];
            auto lds_op = [&](auto llc, auto ic_) {
                constexpr int ll = decltype(llc)::value, i = decltype(ic_)::value, q = ll & 1;
                if constexpr (i < WP) {
                    sZ[q * PAR + (2 * i) * PIT + lane] = DZ(q, 2 * i);
                    if constexpr (2 * i + 1 < W) sZ[q * PAR + (2 * i + 1) * PIT + lane] = DZ(q, 2 * i + 1);
                } else if constexpr (i < 2 * WP) {
                    constexpr int f = 2 * (i - WP);
                    if constexpr (ll > 0) {
                        sH[q * PAR + f * PIT + lane] = HS(ll > 0 ? ll - 1 : 0, f);
                        if constexpr (f + 1 < W) sH[q * PAR + (f + 1) * PIT + lane] = HS(ll > 0 ? ll - 1 : 0, f + 1);
                    }
                } else if constexpr (i < NOPS) {
                    constexpr int j = i - 2 * WP, c = j >> 1;
                    if constexpr ((j & 1) == 0) pa[q][c] = *reinterpret_cast<const f32x4*>(rdZ + q * PAR + 16 * c);
                    else pb[q][c] = *reinterpret_cast<const f32x4*>((ll == 0 ? rdX + xoff : rdH + q * PAR) + 16 * c);
                } else if constexpr (i < NOPS + NC) {
                    if constexpr (ll > 0) wk[q][i - NOPS] = kimg(llc)[(i - NOPS) * 64];       // dgrad weights of layer ll (input side: layer ll - 1)
                }
            };
            // top layer: nothing to hide behind
            dz_of(dzp[(NLT - 1) & 1], hsp[NLT - 1], dH);
            static_for<0, NOPS + NC>([&](auto ic_) { lds_op(std::integral_constant<int, NLT - 1>{}, ic_); });
            static_for<0, NLT>([&](auto lc) {
                constexpr int l = NLT - 1 - decltype(lc)::value, q = l & 1;
                if constexpr (l >= 2) unpark(std::integral_constant<int, (l >= 2 ? l - 2 : 0)>{});      // (layer l - 1's staging, issued under this layer's MFMAs, reads layer l - 2's activations)
                if constexpr (l > 0) {
                    // dgrad of layer l: dH of layer l - 1, then its dZ
#pragma unroll
                    for (int c = 0; c < NC; ++c) dH[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    static_for<0, W>([&](auto oc_) {
                        constexpr int o = decltype(oc_)::value;
#pragma unroll
                        for (int c = 0; c < NC; ++c) dH[c] = mfma_bk<o>(wk[q][c], DZ(q, o), dH[c]);
                    });
                    dz_of(dzp[q ^ 1], hsp[l > 0 ? l - 1 : 0], dH);
                }
                if constexpr (l == 0) {
                    if (has_dxo) {                                   // wave-uniform
                        float* const dxo = A.dZ0_out + (size_t)(wt * WT) + lane;
                        const size_t np = (size_t)A.n_pad;
                        static_for<0, W>([&](auto fc_) {
                            constexpr int f = decltype(fc_)::value;
                            if (f < w) dxo[f * np] = DZ(0, f);
                        });
                    }
                }
                LFENCE();
                // weight gradient of layer l; in the shadow of its MFMAs (two LDS instructions each are free for a lone wave) the
                // staging writes, operand reads and dgrad weights of layer l - 1
                static_for<0, 16>([&](auto ic_) {
                    constexpr int i = decltype(ic_)::value, c = (i >> 3) * 2, t = (i >> 1) & 3;
                    if constexpr (l >= NL) {
                        // per-image layers: their sums are read and cleared at every image change, and around that branch the register
                        // allocator copies the accumulators; behind an inline-assembly MFMA such a copy reads before the result has
                        // landed (it does not know the instruction) -- the builtin lets it place the wait states
                        if constexpr ((i & 1) == 0) wacc[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[q][c][t], pb[q][c][t], wacc[l], 0, 0, 0);
                        else wacd[l] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[q][c + 1][t], pb[q][c + 1][t], wacd[l], 0, 0, 0);
                    } else if constexpr ((i & 1) == 0) mfma16_acc(wacc[l], pa[q][c][t], pb[q][c][t]);
                    else mfma16_acc(wacd[l], pa[q][c + 1][t], pb[q][c + 1][t]);
                    if constexpr (l > 0) {
                        lds_op(std::integral_constant<int, (l > 0 ? l - 1 : 0)>{}, std::integral_constant<int, 2 * i>{});
                        lds_op(std::integral_constant<int, (l > 0 ? l - 1 : 0)>{}, std::integral_constant<int, 2 * i + 1>{});
                    } else if constexpr (LX && i < 4) {
                        // layer 0's second input block (rows 16 .. 31 of the row buffer), read in the shadow of the first block's
                        // MFMAs into the operand registers of the other layer parity (layer 1 is done with them)
                        pb[1][i] = *reinterpret_cast<const f32x4*>(rdX + xoff + 16 * PIT + 16 * i);
                    }
                    LFENCE();
                });
                static_assert(NOPS + NC <= 32, "LDS operations of a layer fit the shadow of sixteen MFMAs");
                if constexpr (LX && l == 0) {
                    static_for<0, 16>([&](auto ic_) {
                        constexpr int i = decltype(ic_)::value, c = (i >> 3) * 2, t = (i >> 1) & 3;
                        if constexpr ((i & 1) == 0) mfma16_acc(wacc0b, pa[0][c][t], pb[1][c][t]);
                        else mfma16_acc(wacd0b, pa[0][c + 1][t], pb[1][c + 1][t]);
                        LFENCE();
                    });
                }
            });
        LSTAMP(5);
    }
    if constexpr (MODE == 1) return;                             // (forward only: nothing accumulated)
    // ================= flush: sum the waves' accumulators, scatter into the flat W^T layout of this workgroup's partial ====
    if constexpr (NI > 0) { if (cur_img >= 0) imgl_flush(cur_img); }        // the last image's layers of this wave
    asm volatile("s_nop 15\n\ts_nop 15");     // (the compiler does not know that the inline-assembly MFMAs' results take eight passes to land)
    __syncthreads();
    const int offWo = w * d + w + (L - 1) * (w * w + w);
    const int Ptot = offWo + (MODE == 0 ? 2 * w + 2 : 0);        // (a head-less block's partial ends with its last Dense layer)
    // fixed binary tree over the waves (deterministic): at stride s the waves with (wv & (2s - 1)) == s park their sums in the
    // region of wave wv - s, which adds them to its own
    constexpr int REG = SM::REG;
#pragma unroll
    for (int l = 0; l < NL; ++l) wacc[l] += wacd[l];
    wacc0b += wacd0b;
    float* const sHead = smem + (NWV / 2) * REG;            // [wave][2 * 16]: the head's sums of every wave
    {
        // head: wave sums of the per-lane sums
#pragma unroll
        for (int k = 0; k <= W; ++k) {
            const float s0 = cl_wave_sum(hacc[k][0]), s1 = cl_wave_sum(hacc[k][1]);
            if (lane == 0) {
                sHead[wv * 32 + k] = s0;
                sHead[wv * 32 + 16 + k] = s1;
            }
        }
    }
#pragma unroll
    for (int s2 = 1; s2 < NWV; s2 <<= 1) {
        f32x4* reg = reinterpret_cast<f32x4*>(smem + ((wv & ~(2 * s2 - 1)) / (2 * s2)) * REG) + lane;
        if ((wv & (2 * s2 - 1)) == s2) {
#pragma unroll
            for (int l = 0; l < NL; ++l) reg[l * 64] = wacc[l];
            if (LX) reg[NL * 64] = wacc0b;
        }
        __syncthreads();
        if ((wv & (2 * s2 - 1)) == 0) {
#pragma unroll
            for (int l = 0; l < NL; ++l) wacc[l] += reg[l * 64];
            if (LX) wacc0b += reg[NL * 64];
        }
        __syncthreads();
    }
    if (wv == 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) reinterpret_cast<f32x4*>(smem + l * 256)[lane] = wacc[l];
        if (LX) reinterpret_cast<f32x4*>(smem + NL * 256)[lane] = wacc0b;
    }
    __syncthreads();
    float* __restrict__ part = A.partials + (size_t)blockIdx.x * Ptot;
    for (int idx = tid; idx < SM::NACCB * 256; idx += NT) {
        const int l = idx >> 8, r = idx & 255;                     // accumulator l: Dense layer l (LX: NL = layer 0's second input block)
        const int ln = r >> 2, t = r & 3;
        const int fo = 4 * (ln >> 4) + t, fi = ln & 15;            // output feature (row), input feature (column; 15: the ones) of this element
        const float v = smem[idx];
        if (LX && l == NL) {
            if (fo < w && ONE + fi < d) part[fo * d + ONE + fi] = v;                      // metadata columns 15 .. 30
        } else if (l < L) {
            const int in_dim = (l == 0) ? d : w;
            const int off = (l == 0) ? 0 : (w * d + w + (l - 1) * (w * w + w));
            if (fo < w && fi < in_dim && fi < ONE) part[off + fo * in_dim + fi] = v;
            if (fo < w && fi == ONE) part[off + w * in_dim + fo] = v;                     // bias gradient: the ones column
        }
    }
    if (MODE == 0 && tid < 32) {
        const int c = tid >> 4, k = tid & 15;                      // head: row c (loc / raw sigma), input feature k; k == W: the bias
        float v = 0.0f;
        for (int q = 0; q < NWV; ++q) v += sHead[q * 32 + tid];
        if (k < w) part[offWo + c * w + k] = v;
        if (k == W) part[offWo + 2 * w + c] = v;
    }

    if constexpr (MODE == 0) {
        float v = cl_wave_sum(nll_acc);
        __syncthreads();
        if (lane == 0) smem[wv] = v;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int k = 0; k < NWV; ++k) t += (double)smem[k];
            if (det) A.nll_part[blockIdx.x] = t;            // every workgroup stores its slot, cl_det_reduce adds them in index order
            else atomicAdd(A.scalars + CL_SC_NLL, t);
        }
        if (use_ev11) {
            ev_g0 = cl_wave_sum(ev_g0); ev_g1 = cl_wave_sum(ev_g1); ev_g2 = cl_wave_sum(ev_g2);
            if (lane == 0) {                         // d softplus(raw)/d raw = sigmoid(raw)
                const float e0 = ev_g0 * cl_sigmoid(A.ev11[0]), e1 = ev_g1 * cl_sigmoid(A.ev11[1]), e2 = ev_g2 * cl_sigmoid(A.ev11[2]);
                if (A.ev11_part != nullptr) {        // deterministic mode: this wave's slot, summed in index order by cl_det_reduce
                    float* slot = A.ev11_part + 3 * (CL_EV11_WAVES * (size_t)blockIdx.x + wv);
                    slot[0] = e0; slot[1] = e1; slot[2] = e2;
                } else {
                    atomicAdd(A.d_ev11 + 0, e0); atomicAdd(A.d_ev11 + 1, e1); atomicAdd(A.d_ev11 + 2, e2);
                }
            }
        }
    }
#ifdef CL_STAMPS
    {
        unsigned long long t_;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        st_acc[7] = t_ - st_last;                                 // flush (st_last: end of the last tile)
        if (A.loc_out != nullptr && lane == 0) {
            unsigned long long* dbg = reinterpret_cast<unsigned long long*>(A.loc_out) + ((size_t)blockIdx.x * NWV + wv) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) dbg[k] = st_acc[k];
        }
    }
#endif
}

// The instances are spread over four compilations of this file (build.py: -DCL_LANE_PART=0 .. 3, in parallel; a part takes 25 - 50 s):
// part 0 = plain layout, metadata in registers (+ the dispatch); 1 = packed layout, registers; 2 = plain, LDS rows (LX); 3 = packed, LX.
#ifndef CL_LANE_PART
#define CL_LANE_PART 0
#endif

template <int W, int DMAX, bool PACKED, bool FULL, bool DXO = false, int NI = 0, int MODE = 0>
static int launch_lane_inst(const cl_mlp_args& a, int grid, hipStream_t st) {
    using SM = LSmem<W, DMAX == 0, NI>;
    const size_t sm = (size_t)SM::total(a.d) * sizeof(float);
    if (sm > 160 * 1024) return -3;
    auto kern = elbo_lane_kernel<W, DMAX, PACKED, FULL, DXO, NI, NL, MODE>;
    static std::atomic<size_t> configured{0};
    size_t have = configured.load(std::memory_order_acquire);
    if (have < sm) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        while (have < sm && !configured.compare_exchange_weak(have, sm, std::memory_order_release, std::memory_order_acquire)) {}
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), sm, st, a);
    return (int)hipGetLastError();
}

// the plain layout has a second instance without the optional inputs / outputs (the training step of a production run); the packed
// layout (single-pass Laue) keeps the one full instance
static inline bool lane_wants_full(const cl_mlp_args& a) { return a.eta != nullptr || a.ipred_out != nullptr || a.ev11 != nullptr || a.dzf_obs != nullptr; }
template <int W, int DMAX, bool PACKED>
static int launch_lane_one(const cl_mlp_args& a, int grid, hipStream_t st) {
    if constexpr (PACKED) return launch_lane_inst<W, DMAX, true, true>(a, grid, st);
    else {
        if (lane_wants_full(a)) return launch_lane_inst<W, DMAX, false, true>(a, grid, st);
        // the production step behind a peeled first layer: metadata = the peeled layer's w pre-activations, in registers (DMAX = 8 or 15)
        if constexpr (DMAX != 0) { if (a.dZ0_out != nullptr) return launch_lane_inst<W, DMAX, false, false, true>(a, grid, st); }
        else if (a.dZ0_out != nullptr) return launch_lane_inst<W, DMAX, false, true>(a, grid, st);
        return launch_lane_inst<W, DMAX, false, false>(a, grid, st);
    }
}

#ifndef CL_LANE_WMAX
#define CL_LANE_WMAX 10
#endif
#define CL_LANE_IMGL_MAX 2          /* per-image layers of the lane instances with all three forms (production, full, dZ_0 out) ... */
#define CL_LANE_IMGL_MAX_NL 3       /* ... a third one: production and full form; at the default depth in a unit of its own (CL_LANE_PART = 5) */
#define CL_LANE_IMGL3_DEPTH_MAX 18  /* ... the third one on 2 .. 18 and on 20 Dense layers (19: see CL_LANE_PART 9) */
#ifndef CL_LANE_DEPTH_WMIN
#define CL_LANE_DEPTH_WMIN 5        /* narrowest scaler the other-depth instances (compiled at widths 8 and 10) take */
#endif

// the instance whose compile-time width is the smallest one that holds the scaler (zero-padded features cost MFMA steps)
#define CL_LANE_WIDTHS(CASE)      \
    if (a.w <= 4) return CASE(4); \
    if (a.w <= 6) return CASE(6); \
    if (a.w <= 8) return CASE(8); \
    return CASE(10);
// ... with the metadata in registers also widths 11 and 12 (round 6: three output chunks like width 10, two more activation registers per
// layer -- 22 .. 72 spilled registers at the default depth, none from 16 layers down; still ahead of elbo_narrow.hip: profiles/r6_envelope_w12.txt)
#define CL_LANE_W12 12
#define CL_LANE_WIDTHS_REG(CASE)   \
    if (a.w <= 4) return CASE(4);  \
    if (a.w <= 6) return CASE(6);  \
    if (a.w <= 8) return CASE(8);  \
    if (a.w <= 10) return CASE(10); \
    return CASE(CL_LANE_W12);

int cl_launch_lane_plain_reg(const cl_mlp_args& a, int grid, hipStream_t st);
int cl_launch_lane_packed_reg(const cl_mlp_args& a, int grid, hipStream_t st);
int cl_launch_lane_plain_rows(const cl_mlp_args& a, int grid, hipStream_t st);
int cl_launch_lane_packed_rows(const cl_mlp_args& a, int grid, hipStream_t st);
int cl_launch_lane_imgl_inst(const cl_mlp_args& a, int grid, hipStream_t st);
int cl_launch_lane_imgl3(const cl_mlp_args& a, int grid, hipStream_t st);
// other depths than the default (round 6): one compilation per depth, CL_LANE_PART = 7
#define CL_LANE_DEPTHS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19)
#define CL_LANE_DEPTH_DECL(D) int cl_launch_lane_depth##D(const cl_mlp_args& a, int grid, hipStream_t st);
CL_LANE_DEPTHS(CL_LANE_DEPTH_DECL)
#undef CL_LANE_DEPTH_DECL
// ... and the two launches of a head-less layer block (MODE 1 / 2) at every depth 2 .. 20: the blocks in front of the last one of a chained scaler
#define CL_LANE_BLOCK_DECL(D) int cl_launch_lane_block##D(const cl_mlp_args& a, int mode, int grid, hipStream_t st);
CL_LANE_DEPTHS(CL_LANE_BLOCK_DECL)
CL_LANE_BLOCK_DECL(20)
#undef CL_LANE_BLOCK_DECL
// ... and the per-image-layer instances at every depth 2 .. 19 (round 6; CL_LANE_PART = 9, one compilation per depth)
#define CL_LANE_IMGLD_DECL(D) int cl_launch_lane_imgl_depth##D(const cl_mlp_args& a, int grid, hipStream_t st);
CL_LANE_DEPTHS(CL_LANE_IMGLD_DECL)
#undef CL_LANE_IMGLD_DECL
static inline bool lane_has_depth(int L) {
#define CL_LANE_DEPTH_TEST(D) if (L == D) return true;
    CL_LANE_DEPTHS(CL_LANE_DEPTH_TEST)
#undef CL_LANE_DEPTH_TEST
    return false;
}

#if CL_LANE_PART == 0
int cl_launch_lane_plain_reg(const cl_mlp_args& a, int grid, hipStream_t st) {
#define CL_LANE_CASE(WW) (a.d <= 8 ? launch_lane_one<WW, 8, false>(a, grid, st) : (WW <= CL_LANE_WMAX ? launch_lane_one<(WW <= CL_LANE_WMAX ? WW : 4), DMAX_ALL, false>(a, grid, st) : -2))
    CL_LANE_WIDTHS_REG(CL_LANE_CASE)
#undef CL_LANE_CASE
}

// LDS bytes the LDS-row instance of hidden width w needs for d metadata columns
static size_t lane_rows_lds(int w, int d) {
    const int t = w <= 4 ? LSmem<4, true>::total(d) : (w <= 6 ? LSmem<6, true>::total(d) : (w <= 8 ? LSmem<8, true>::total(d) : LSmem<10, true>::total(d)));
    return (size_t)t * sizeof(float);
}

// 1 = this geometry runs on the lane-per-observation kernel (full ELBO step of a scaler of exactly NL layers -- the default depth --;
// plain observation layout, or the packed one of single-pass Laue).  Up to 15 metadata columns are registers of the lane, 16 .. 31
// are rows of an LDS buffer (LX); any number of MC samples (batches of SPRE).  4 M observations, 20 x 10, Student-T, ms per step
// here / on elbo_narrow.hip (scripts/narrow_samples.py): S = 1 0.96 / 1.11, 2: 1.01 / 1.12, 4: 1.07 / 1.19, 8: 1.16 / 1.34.
int cl_lane_supports(const cl_mlp_args& a) {
    // (CARELESS_HIP_LANE_W12=0: A/B runs of widths 11 and 12 against the narrow kernel they ran on until round 6)
    static const bool w12_on = [] { const char* e = getenv("CARELESS_HIP_LANE_W12"); return !(e != nullptr && e[0] == '0'); }();
    const int wtop = (w12_on && a.d <= DMAX_ALL) ? CL_LANE_W12 : CL_LANE_WMAX;
    if (!(a.w >= 1 && a.w <= wtop && a.S >= 1 && a.d >= 1 && a.d <= DMAX_LX && (a.L == NL || lane_has_depth(a.L)) && a.n_imgl == 0 && a.act_out == nullptr &&
          a.dH_ext == nullptr && a.dX_out == nullptr && (a.row_map != nullptr || a.gmeta == nullptr)))
        return 0;
    // the other depths (round 6): instances at widths 8 and 10 (a narrower scaler pays the padded steps: from width CL_LANE_DEPTH_WMIN on it still
    // beats elbo_narrow.hip), metadata in registers (more columns: behind the engine's peeled first layer, dZ_0 out in the plain layout)
    // (CARELESS_HIP_LANE_DEPTHS=0: A/B runs against the narrow kernel these shapes ran on until round 5)
    static const bool depths_on = [] { const char* e = getenv("CARELESS_HIP_LANE_DEPTHS"); return !(e != nullptr && e[0] == '0'); }();
    if (a.L != NL) return depths_on && a.w >= CL_LANE_DEPTH_WMIN && a.d <= DMAX_ALL && (a.dZ0_out == nullptr || a.row_map == nullptr);
    // widths 11, 12 at the default depth: 22 .. 72 spilled registers -- ahead of the narrow kernel on <= 8 columns (1.05 against 1.11 ms at
    // 4 M observations), behind it on 9 .. 15 (1.17 against 1.12)
    if (a.w > CL_LANE_WMAX) return a.d <= 8;
    return a.d <= DMAX_ALL || lane_rows_lds(a.w, a.d) <= 160 * 1024;
}

// 1 = the training step of a NeuralImageScaler runs on the lane-per-observation kernel (round 5): the default depth and width (NL Dense layers,
// w <= 10) on up to 15 metadata columns with one or two per-image layers (`careless mono | poly --image-layers 1|2`) in the packed-by-image
// layout (Laue data: harmonic groups inside 16-row granules, single pass, as without per-image layers); everything else with per-image
// layers stays on the IMGL instances of elbo_mlp.hip.
// Round 6: the same at 2 .. 19 Dense layers (`--mlp-layers D --image-layers 1|2`: the instances of the per-depth units, width >= CL_LANE_DEPTH_WMIN).
int cl_lane_imgl_supports(const cl_mlp_args& a) {
    static const bool depths_on = [] { const char* e = getenv("CARELESS_HIP_LANE_DEPTHS"); return !(e != nullptr && e[0] == '0'); }();
    const bool depth_ok = a.L == NL || (depths_on && lane_has_depth(a.L) && a.w >= CL_LANE_DEPTH_WMIN);
    return a.n_imgl >= 1 && a.n_imgl <= ((a.L == NL || a.L <= CL_LANE_IMGL3_DEPTH_MAX) ? CL_LANE_IMGL_MAX_NL : CL_LANE_IMGL_MAX) && a.w >= 1 && a.w <= CL_LANE_WMAX && a.S >= 1 && a.d >= 1 && a.d <= DMAX_ALL && depth_ok &&
           a.act_out == nullptr && a.dH_ext == nullptr && a.dX_out == nullptr && a.row_map != nullptr &&
           (a.gmeta == nullptr || a.tile_gmax != nullptr) && !a.use_img && a.imgl != nullptr && a.d_imgl != nullptr && a.tile_img != nullptr && a.n_images >= 1;
}

// 1 = this launch of a head-less layer block (mode 1: forward with act_out; mode 2: backward from dH_ext) runs on the lane kernel (round 6):
// 2 .. 20 Dense layers of width 5 .. 10 on <= 15 input columns in the plain layout, the FIRST block of a chain (no dX_out)
int cl_lane_block_supports(const cl_mlp_args& a, int mode) {
    static const bool on = [] { const char* e = getenv("CARELESS_HIP_LANE_BLOCKS"); return !(e != nullptr && e[0] == '0'); }();
    if (!on || !(mode == 1 || mode == 2)) return 0;
    if (!(a.w >= CL_LANE_DEPTH_WMIN && a.w <= (a.L == NL ? CL_LANE_WMAX : CL_LANE_W12) && a.d >= 1 && a.d <= DMAX_ALL && (a.L == NL || lane_has_depth(a.L)) && a.n_imgl == 0 &&
          a.row_map == nullptr && a.gmeta == nullptr && a.dX_out == nullptr && a.dO_ext == nullptr && a.dZ0_out == nullptr))
        return 0;
    if (mode == 1) return a.act_out != nullptr && a.dH_ext == nullptr && a.loc_out == nullptr && a.sig_out == nullptr;
    return a.dH_ext != nullptr && a.act_out == nullptr && a.partials != nullptr;
}

int cl_launch_lane_block(const cl_mlp_args& a, int mode, int grid, hipStream_t st) {
    if (!cl_lane_block_supports(a, mode)) return -2;
    if (a.n_pad % CL_MLP_TILE != 0 || a.n_pad <= 0 || grid < 1) return -1;
    if (4ull * (unsigned long long)((a.d + 3) & ~3) * (unsigned long long)a.n_pad >= (1ull << 32)) return -4;
#define CL_LANE_BLOCK_CALL(D) if (a.L == D) return cl_launch_lane_block##D(a, mode, grid, st);
    CL_LANE_DEPTHS(CL_LANE_BLOCK_CALL)
    CL_LANE_BLOCK_CALL(20)
#undef CL_LANE_BLOCK_CALL
    return -2;
}

int cl_launch_lane_imgl(const cl_mlp_args& a, int grid, hipStream_t st) {
    if (!cl_lane_imgl_supports(a)) return -2;
    if (a.n_pad % CL_MLP_TILE != 0 || a.n_pad <= 0 || a.n_obs != a.n_pad) return -1;
    if (4ull * (unsigned long long)((a.d + 3) & ~3) * (unsigned long long)a.n_pad >= (1ull << 32) ||
        4ull * (unsigned long long)a.R * (unsigned long long)a.S >= (1ull << 32))
        return -4;
    if ((a.eta != nullptr || a.ipred_out != nullptr) && 4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
    if (grid < 1) return -1;
    if (a.dzf_obs != nullptr) {          // deterministic mode (round 6), as cl_launch_lane; the per-image gradients: one wave per image (elbo_lane_kernel)
        if (a.ev11 != nullptr && a.ev11_part == nullptr) return -2;
        if (a.nll_part == nullptr) return -1;
        if (4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
    }
#define CL_LANE_IMGLD_CALL(D) if (a.L == D) return cl_launch_lane_imgl_depth##D(a, grid, st);
    CL_LANE_DEPTHS(CL_LANE_IMGLD_CALL)
#undef CL_LANE_IMGLD_CALL
    if (a.n_imgl == CL_LANE_IMGL_MAX_NL) return cl_launch_lane_imgl3(a, grid, st);
    return cl_launch_lane_imgl_inst(a, grid, st);
}

// name of the instance cl_launch_lane_imgl runs (cl_mlp_kernel_name)
int cl_lane_imgl_kernel_name(const cl_mlp_args& a, char* out, size_t n) {
    const bool full = lane_wants_full(a) || (a.n_imgl > CL_LANE_IMGL_MAX && a.dZ0_out != nullptr);       // (three layers + dZ_0 out: the full instance)
    const char* dxo = (!full && a.dZ0_out != nullptr) ? "true" : "false";
    const char* det = a.dzf_obs != nullptr ? " (deterministic stores)" : "";
    if (a.L != NL) return snprintf(out, n, "elbo_lane_kernel<%d, %d, true, %s, %s, %d, %d> (image layers)%s", CL_LANE_WMAX, DMAX_ALL, full ? "true" : "false", dxo, a.n_imgl, a.L, det);
    return snprintf(out, n, "elbo_lane_kernel<%d, %d, true, %s, %s, %d> (image layers)%s", CL_LANE_WMAX, (a.d <= 8 && a.n_imgl <= CL_LANE_IMGL_MAX) ? 8 : DMAX_ALL,
                    full ? "true" : "false", dxo, a.n_imgl, det);
}

// name of the instance cl_launch_lane runs (cl_mlp_kernel_name)
int cl_lane_kernel_name(const cl_mlp_args& a, char* out, size_t n) {
    const int W = a.w <= 4 ? 4 : (a.w <= 6 ? 6 : (a.w <= 8 ? 8 : (a.w <= 10 ? 10 : CL_LANE_W12)));
    const int DM = (a.d <= 8 && a.L == NL) ? 8 : (a.d <= DMAX_ALL ? DMAX_ALL : 0);       // (the other depths: one metadata capacity)
    const bool packed = a.row_map != nullptr;
    const bool full = packed || lane_wants_full(a) || (DM == 0 && a.dZ0_out != nullptr);
    if (a.L != NL)
        return snprintf(out, n, "elbo_lane_kernel<%d, %d, %s, %s, %s, 0, %d>%s", a.w <= 8 ? 8 : (a.w <= 10 ? CL_LANE_WMAX : CL_LANE_W12), DM, packed ? "true" : "false", full ? "true" : "false",
                        (!full && a.dZ0_out != nullptr) ? "true" : "false", a.L, a.dzf_obs != nullptr ? " (deterministic stores)" : "");
    return snprintf(out, n, "elbo_lane_kernel<%d, %d, %s, %s%s>%s", W, DM, packed ? "true" : "false", full ? "true" : "false",
                    (!full && a.dZ0_out != nullptr) ? ", true" : "", a.dzf_obs != nullptr ? " (deterministic stores)" : "");
}

int cl_launch_lane(const cl_mlp_args& a, int grid, hipStream_t st) {
    if (!cl_lane_supports(a)) return -2;
    if (a.n_pad % CL_MLP_TILE != 0 || a.n_pad <= 0) return -1;
    if (4ull * (unsigned long long)((a.d + 3) & ~3) * (unsigned long long)a.n_pad >= (1ull << 32) ||
        4ull * (unsigned long long)a.R * (unsigned long long)a.S >= (1ull << 32))
        return -4;
    if (grid < 1) return -1;
    if (a.dzf_obs != nullptr) {          // deterministic mode: stores per (observation, sample) / observation / workgroup / wave (Evans-2011 terms)
        if (a.ev11 != nullptr && a.ev11_part == nullptr) return -2;      // (the Evans-2011 gradients need their per-wave slots)
        if (a.nll_part == nullptr || (a.use_img && a.dimg_obs == nullptr)) return -1;
        if (4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
    }
    if (a.L != NL) {
        if (a.row_map != nullptr) {
            if (a.n_obs != a.n_pad || (a.gmeta != nullptr && a.tile_gmax == nullptr)) return -1;
            if ((a.eta != nullptr || a.ipred_out != nullptr) && 4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
        }
#define CL_LANE_DEPTH_CALL(D) if (a.L == D) return cl_launch_lane_depth##D(a, grid, st);
        CL_LANE_DEPTHS(CL_LANE_DEPTH_CALL)
#undef CL_LANE_DEPTH_CALL
        return -2;
    }
    // metadata as LDS rows from DMAX_ALL + 1 columns on (from 9 on it measured slower than the register instances, round 3)
    const bool rows = a.d > DMAX_ALL;
    if (a.row_map != nullptr) {
        if (a.n_obs != a.n_pad || (a.gmeta != nullptr && a.tile_gmax == nullptr)) return -1;
        if ((a.eta != nullptr || a.ipred_out != nullptr) && 4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
        return rows ? cl_launch_lane_packed_rows(a, grid, st) : cl_launch_lane_packed_reg(a, grid, st);
    }
    return rows ? cl_launch_lane_plain_rows(a, grid, st) : cl_launch_lane_plain_reg(a, grid, st);
}
#elif CL_LANE_PART == 1
int cl_launch_lane_packed_reg(const cl_mlp_args& a, int grid, hipStream_t st) {
#define CL_LANE_CASE(WW) (a.d <= 8 ? launch_lane_one<WW, 8, true>(a, grid, st) : (WW <= CL_LANE_WMAX ? launch_lane_one<(WW <= CL_LANE_WMAX ? WW : 4), DMAX_ALL, true>(a, grid, st) : -2))
    CL_LANE_WIDTHS_REG(CL_LANE_CASE)
#undef CL_LANE_CASE
}
#elif CL_LANE_PART == 2
int cl_launch_lane_plain_rows(const cl_mlp_args& a, int grid, hipStream_t st) {
#define CL_LANE_CASE(WW) launch_lane_one<WW, 0, false>(a, grid, st)
    CL_LANE_WIDTHS(CL_LANE_CASE)
#undef CL_LANE_CASE
}
#elif CL_LANE_PART == 7
// another depth than the default (-DCL_LANE_NL=D): widths 8, 10 and 12, metadata in registers -- plain layout (production / full) and packed
#define CL_LANE_DEPTH_FN2(D) cl_launch_lane_depth##D
#define CL_LANE_DEPTH_FN(D) CL_LANE_DEPTH_FN2(D)
// (one metadata capacity -- 15 columns in registers -- for every depth below the default: the eight-column instances of the default depth buy
//  back registers the shallower units do not miss, and would double the 19 units' compile time)
template <int WW>
static int launch_lane_depth_w(const cl_mlp_args& a, int grid, hipStream_t st) {
    if (a.row_map != nullptr) return launch_lane_inst<WW, DMAX_ALL, true, true>(a, grid, st);
    if (lane_wants_full(a)) return launch_lane_inst<WW, DMAX_ALL, false, true>(a, grid, st);
    // behind a peeled first layer (more than 15 metadata columns: the engine hands over the layer's w pre-activations): dZ_0 out
    if (a.dZ0_out != nullptr) return launch_lane_inst<WW, DMAX_ALL, false, false, true>(a, grid, st);
    return launch_lane_inst<WW, DMAX_ALL, false, false>(a, grid, st);
}
int CL_LANE_DEPTH_FN(CL_LANE_NL)(const cl_mlp_args& a, int grid, hipStream_t st) {
    return a.w <= 8 ? launch_lane_depth_w<8>(a, grid, st) : (a.w <= 10 ? launch_lane_depth_w<CL_LANE_WMAX>(a, grid, st) : launch_lane_depth_w<CL_LANE_W12>(a, grid, st));
}
#endif
#if CL_LANE_PART == 7 || CL_LANE_PART == 8
// the two launches of a head-less layer block of this depth (part 8: the default depth)
template <int WW, int MODE>
static int launch_lane_block_w(const cl_mlp_args& a, int grid, hipStream_t st) {
    if constexpr (NL == CL_MLP_LMAX_W16) return a.d <= 8 ? launch_lane_inst<WW, 8, false, false, false, 0, MODE>(a, grid, st) : launch_lane_inst<WW, DMAX_ALL, false, false, false, 0, MODE>(a, grid, st);
    else return launch_lane_inst<WW, DMAX_ALL, false, false, false, 0, MODE>(a, grid, st);
}
#define CL_LANE_BLOCK_FN2(D) cl_launch_lane_block##D
#define CL_LANE_BLOCK_FN(D) CL_LANE_BLOCK_FN2(D)
int CL_LANE_BLOCK_FN(CL_LANE_NL)(const cl_mlp_args& a, int mode, int grid, hipStream_t st) {
    // (widths 11, 12: the per-depth units only -- at the default depth twelve activations per layer spill)
    if constexpr (NL != CL_MLP_LMAX_W16) {
        if (a.w > CL_LANE_WMAX) return mode == 1 ? launch_lane_block_w<CL_LANE_W12, 1>(a, grid, st) : launch_lane_block_w<CL_LANE_W12, 2>(a, grid, st);
    } else if (a.w > CL_LANE_WMAX) return -2;
    if (mode == 1) return a.w <= 8 ? launch_lane_block_w<8, 1>(a, grid, st) : launch_lane_block_w<CL_LANE_WMAX, 1>(a, grid, st);
    return a.w <= 8 ? launch_lane_block_w<8, 2>(a, grid, st) : launch_lane_block_w<CL_LANE_WMAX, 2>(a, grid, st);
}
#endif
#if CL_LANE_PART == 5
// THREE per-image layers on the default depth (round 6): 23 layers of activations.  With the four parked layers the LDS has room for
// (157.5 of 160 KB) hipcc's `AMDGPU Rewrite AGPR-Copy-MFMA` pass -- what -amdgpu-mfma-vgpr-form=1 switches on -- crashes on these
// instances, so this unit is compiled WITHOUT that option (build.py): 97 .. 145 spilled registers, still ahead of the 16-wide IMGL
// instance of elbo_mlp.hip these scalers ran on.  One metadata capacity; behind a peeled first layer (dZ_0 out) the FULL instance: without
// the option the dZ_0-storing production instance comes out with accumulator-register copies one instruction in front of the inline-assembly
// MFMAs that read them (the build's wait-state scan refuses it: scripts/check_lane_isa.py, rule R1).
int cl_launch_lane_imgl3(const cl_mlp_args& a, int grid, hipStream_t st) {
    if (lane_wants_full(a) || a.dZ0_out != nullptr) return launch_lane_inst<CL_LANE_WMAX, DMAX_ALL, true, true, false, CL_LANE_IMGL_MAX_NL>(a, grid, st);
    return launch_lane_inst<CL_LANE_WMAX, DMAX_ALL, true, false, false, CL_LANE_IMGL_MAX_NL>(a, grid, st);
}
#endif
#if CL_LANE_PART == 9
// per-image layers on another depth than the default (-DCL_LANE_NL=D): the widest instance, one metadata capacity, as the Dense-only units
#define CL_LANE_IMGLD_FN2(D) cl_launch_lane_imgl_depth##D
#define CL_LANE_IMGLD_FN(D) CL_LANE_IMGLD_FN2(D)
int CL_LANE_IMGLD_FN(CL_LANE_NL)(const cl_mlp_args& a, int grid, hipStream_t st) {
    const bool full = lane_wants_full(a);
#define CL_LANE_IMGL_CASE(NI_) (full ? launch_lane_inst<CL_LANE_WMAX, DMAX_ALL, true, true, false, NI_>(a, grid, st) : \
                                (a.dZ0_out != nullptr ? launch_lane_inst<CL_LANE_WMAX, DMAX_ALL, true, false, true, NI_>(a, grid, st) : \
                                                        launch_lane_inst<CL_LANE_WMAX, DMAX_ALL, true, false, false, NI_>(a, grid, st)))
    if (a.n_imgl == 1) return CL_LANE_IMGL_CASE(1);
    if (a.n_imgl == 2) return CL_LANE_IMGL_CASE(2);
#undef CL_LANE_IMGL_CASE
    // three per-image layers: production and full instance (behind a peeled first layer the full one, as at the default depth) -- up to
    // CL_LANE_IMGL3_DEPTH_MAX Dense layers: on 19 the full instance (22 layers) crashes the compiler pass named at CL_LANE_PART 5
#if CL_LANE_NL <= CL_LANE_IMGL3_DEPTH_MAX
    if (full || a.dZ0_out != nullptr) return launch_lane_inst<CL_LANE_WMAX, DMAX_ALL, true, true, false, CL_LANE_IMGL_MAX_NL>(a, grid, st);
    return launch_lane_inst<CL_LANE_WMAX, DMAX_ALL, true, false, false, CL_LANE_IMGL_MAX_NL>(a, grid, st);
#else
    return -2;
#endif
}
#endif
#if CL_LANE_PART == 4
// per-image layers: the widest instance serves every w <= 10 (a narrower scaler pays the padded MFMA steps: --image-layers on a
// non-default width is rare); with and without the optional inputs / outputs, as the plain layout
int cl_launch_lane_imgl_inst(const cl_mlp_args& a, int grid, hipStream_t st) {
    // Behind a peeled first layer (more than 15 metadata columns) the launch stores dZ_0: the production instance <.., false, true, NI>.
    // (Round 5 built it, saw results that moved from run to run and withdrew it for the FULL instance; round 6 found the cause -- the
    //  inline-assembly LeakyReLU one wait state in front of an MFMA, NOTEBOOK R6.1 -- and it is back: 4.44 -> 4.0 ms per step at 10 M
    //  observations and 8 samples.)
    const bool full = lane_wants_full(a);
#define CL_LANE_IMGL_CASE(DM, NI_) (full ? launch_lane_inst<CL_LANE_WMAX, DM, true, true, false, NI_>(a, grid, st) : \
                                    (a.dZ0_out != nullptr ? launch_lane_inst<CL_LANE_WMAX, DM, true, false, true, NI_>(a, grid, st) : \
                                                            launch_lane_inst<CL_LANE_WMAX, DM, true, false, false, NI_>(a, grid, st)))
    if (a.n_imgl == 1) return a.d <= 8 ? CL_LANE_IMGL_CASE(8, 1) : CL_LANE_IMGL_CASE(DMAX_ALL, 1);
    return a.d <= 8 ? CL_LANE_IMGL_CASE(8, 2) : CL_LANE_IMGL_CASE(DMAX_ALL, 2);
#undef CL_LANE_IMGL_CASE
}
#elif CL_LANE_PART == 3
int cl_launch_lane_packed_rows(const cl_mlp_args& a, int grid, hipStream_t st) {
#define CL_LANE_CASE(WW) launch_lane_one<WW, 0, true>(a, grid, st)
    CL_LANE_WIDTHS(CL_LANE_CASE)
#undef CL_LANE_CASE
}
#endif
