// Narrow-scaler instance of the fused ELBO step (gfx950 / CDNA4 only): hidden width <= 15, metadata width <= 15, up to 20
// Dense layers -- the geometry of the careless CLI default (--mlp-layers 20, --mlp-width = metadata width or 10).
//
// Same arithmetic and the same reference lines as elbo_mlp.hip (scaler forward / sample / predict / likelihood / backward:
// careless/models/scaling/nn.py:92-120, image.py:53-63, models/merging/variational.py:156-181, 197-202,
// models/likelihoods/mono.py:10-73), a different work decomposition.  A 16-wide layer is ONE 16x16 MFMA block, so a layer step of
// one 16-observation group is a chain of 3-4 dependent v_mfma_f32_16x16x4_f32 behind an LDS round trip; with two such waves per
// SIMD (elbo_mlp.hip, WP = 16) the matrix pipe is 41 % busy and the waves spend 70 % of their time waiting on dependencies.  Here
//   * a workgroup is FOUR waves, one per SIMD, with the whole 512-register file (arch + accumulator VGPRs) to itself;
//   * a wave carries G = 4 (2) independent 16-observation groups through every layer, so 4 (2) MFMA chains are in flight and a
//     group's vector work (LeakyReLU, its derivative, the staging writes) runs while the other groups' MFMAs execute;
//   * every weight operand read from LDS serves all G groups; all weight-gradient accumulators (21 blocks of 16x16) live in
//     registers -- no LDS accumulator slots, no scratch;
//   * a wave only ever reads LDS staging columns it wrote itself: the main loop has NO workgroup barrier;
//   * the Dense(2) head is one more 16x16 layer whose two output rows sit, for group g, in rows 4g and 4g+1 of ONE shared
//     accumulator: after the head's MFMAs lane (j, g) holds (loc, raw sigma) of observation 16g + j, which is exactly the
//     lane = observation map of the epilogue, and the epilogue's (dL/dloc, dL/draw) are, as they stand, the B operand of the
//     head's dgrad -- no shuffles in either direction;
//   * biases ride on a constant-one feature (slot 15 of every layer input, metadata included): column 15 of a layer's dW^T
//     accumulator is its bias gradient.
// Feature f of a layer lives in accumulator row ("slot") 4 (f & 3) + (f >> 2) (an involution), so MFMA step t contracts the
// features 4t .. 4t+3 and a width-10 layer needs 3 of the 4 steps (KS).
// Roofline: fp32 MFMA; algorithmic flops per observation 6 (d w + (L-1) w^2 + 2 w).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "cl_math.h"
#include "cl_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));



namespace {

constexpr int NL = CL_MLP_LMAX_W16;   // Dense layers one launch holds
constexpr int NPW = 20;               // row pitch of a 16 x 16 weight image

__device__ __forceinline__ int slot_of(int f) { return ((f & 3) << 2) | (f >> 2); }   // feature <-> slot (involution on 0..15)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// (one v_max_f32 the compiler knows: -fno-honor-nans, see elbo_mlp.hip)
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

// dZ = dH * lrelu'(h): dH where h > 0, leak dH otherwise (h == 0 takes the leak branch, like `h > 0 ? ... : ...`; -0.0 cannot
// occur: h = max(x, leak x)).  A compare / conditional-move pair; the multiply + sign mask + bit select form the issue-time probe
// (scripts/probe/coissue_probe.hip) suggested was measured slower -- hipcc makes five instructions of it
// (scripts/patches/r2_narrow_closed_switches.diff, with the start stagger and the static priority of the second wave: no effect).
__device__ __forceinline__ float lrelu_bwd(float h, float dh, float leak) {
    return (h > 0.0f) ? dh : leak * dh;
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int opaque_uniform(int v) {
    v = __builtin_amdgcn_readfirstlane(v);
    asm volatile("" : "+s"(v));
    return v;
}
// "Is layer l the top layer?" as a bit test on an opaque SGPR mask (1 << (L - 1)).  Written as `l == Lt - 1` hipcc replaces the
// unrolled layer number l inside the guarded block by the run-time value Lt - 1, finds the twenty blocks identical, merges them
// into one that indexes `hs` at run time -- and the activations land in scratch memory.
__device__ __forceinline__ unsigned top_layer_mask(int L) {
    unsigned m = 1u << (unsigned)(__builtin_amdgcn_readfirstlane(L) - 1);
    asm volatile("" : "+s"(m));
    return m;
}
template <class T>
__device__ __forceinline__ T ld_uo(const T* base, unsigned byte_off) {       // (wave-uniform pointer)[32-bit per-lane byte offset]
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// pins the order of the hand-interleaved instruction stream: the scheduler may not move anything across it
#define NFENCE() __builtin_amdgcn_sched_barrier(0)

// Diagnostic build only (-DCL_STAMPS): per-wave cycle shares of the phases of a wave tile; the shipped library executes no stamp
#ifdef CL_STAMPS
#define NSTAMP(k)                                                                                  \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
        st_acc[k] += t_ - st_last;                                                                 \
        st_last = t_;                                                                              \
    } while (0)
#else
#define NSTAMP(k)
#endif

template <int G, int NWAVES>
struct NSmem {
    static constexpr int PBW = 16 * G + 4;                    // pitch of a per-wave staging tile [16 slots][16 G observations]
    static constexpr int oW = 0;                              // NL layer images + G head images, [16][NPW] each
    static constexpr int oB = oW + (NL + G) * 16 * NPW;       // (NL + 1) x 16 bias images
    static constexpr int oT = oB + (NL + 1) * 16;             // per wave: sZ [16 rows], sH [16 rows], sD [3 rows: dloc, draw, zeros]
    static constexpr int TW = (2 * 16 + 3) * PBW;
    static constexpr int oA = (oT + NWAVES * TW + 3) & ~3;    // LDS-resident weight-gradient accumulators: [layer - LREG][wave][lane] float4
    static constexpr int SLOT = NWAVES * 256;                 // floats per layer
    static constexpr int ROOM = (160 * 1024 / 4 - oA) / SLOT;
    // one wave per SIMD has the registers for every accumulator; two per SIMD (256 registers each) keep the accumulators of the
    // upper layers in private LDS slots, read before and written after that layer's wgrad MFMAs of a tile
    static constexpr int NACC = (NWAVES <= 4) ? 0 : (ROOM < NL - 8 ? (ROOM > 0 ? ROOM : 0) : NL - 8);
    static constexpr int LREG = NL - NACC;                    // layers < LREG accumulate in registers
    static constexpr int main_total = oA + NACC * SLOT;
    static constexpr int flush_total = (NWAVES / 2 > 0 ? NWAVES / 2 : 1) * (NL + 1) * 256;   // the flush parks half the waves' accumulators
    static constexpr int total = main_total > flush_total ? main_total : flush_total;
};

}  // namespace

// PACKED: the packed observation layout of include/careless_hip.h (row_map; single-pass Laue: gmeta / tile_gmax / noise_row): rows the
// engine ordered so that a harmonic group sits inside a 16-row granule, padding rows have refl_id = -1, n_obs == n_pad; everything
// the caller indexes by row (eta, ipred_out, the noise key) goes through row_map.
template <int G, int KS, int NWAVES, bool PACKED>
__global__ __launch_bounds__(64 * NWAVES) __attribute__((amdgpu_waves_per_eu(NWAVES / 4, NWAVES / 4)))
void elbo_narrow_kernel(const cl_mlp_args A) {
    using SM = NSmem<G, NWAVES>;
    constexpr int NT = 64 * NWAVES;
    constexpr int LREG = SM::LREG;
    constexpr int PBW = SM::PBW;
    constexpr int WT = 16 * G;                        // observations of one wave tile
    static_assert(WT == 32, "the epilogue's lane map (observation = lane & 31, sample parity = lane >> 5) assumes 32-observation wave tiles");
    static_assert(CL_MLP_TILE % WT == 0, "a wave tile must not straddle the end of the padded observation axis");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const sW = smem + SM::oW;
    float* const sB = smem + SM::oB;

    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;   // a previous step hit a non-finite gradient norm

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = uniform(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int d = A.d, w = A.w, L = A.L;
    const float leak = A.leak;
    const int ks1 = (d + 3) >> 2;                     // k-steps of the metadata that hold rows of meta_t (<= KS, see cl_launch_narrow)

    // ---- weight images (W^T layout of cl_kernels.h -> slot-permuted, zero-padded 16 x 16 images) ------------------------
    {
        const float* __restrict__ P = A.mlp;
        // (all loads of a thread are issued before the first LDS store: a plain loop waits for every load in turn, ~14 serial
        // round trips per launch -- a fixed cost that dominates small data sets)
        constexpr int NIMG = (NL + G) * 16 * NPW, NIT = (NIMG + NT - 1) / NT;
        float wv_[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * NT + tid;
            const int l = idx / (16 * NPW), r = idx - l * (16 * NPW);
            const int os = r / NPW, is = r - os * NPW;               // output slot (row), input slot (column)
            float v = 0.0f;
            if (idx < NIMG && is < 16) {
                const int fi = slot_of(is);
                if (l < L) {
                    const int fo = slot_of(os);
                    const int in_dim = (l == 0) ? d : w;
                    const float* Wl = (l == 0) ? P : P + w * d + w + (l - 1) * (w * w + w);
                    if (fo < w && fi < in_dim) v = Wl[fo * in_dim + fi];
                } else if (l >= NL) {                                // head image of group g: rows 4g (loc) and 4g+1 (raw sigma)
                    const int g = l - NL, c = os - 4 * g;
                    const float* Wo = P + w * d + w + (L - 1) * (w * w + w);
                    if ((c == 0 || c == 1) && fi < w) v = Wo[c * w + fi];
                }
            }
            wv_[it] = v;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * NT + tid;
            if (idx < NIMG) sW[idx] = wv_[it];
        }
        for (int idx = tid; idx < (NL + 1) * 16; idx += NT) {
            const int l = idx >> 4, os = idx & 15;
            float v = 0.0f;
            if (l < L) {
                const int fo = slot_of(os);
                if (fo < w) v = (l == 0) ? P[w * d + fo] : P[w * d + w + (l - 1) * (w * w + w) + w * w + fo];
                if (os == 15) v = 1.0f;                              // the constant-one feature of the next layer's input
            } else if (l == NL) {
                const int c = os & 3;
                if (c < 2 && (os >> 2) < G) v = P[w * d + w + (L - 1) * (w * w + w) + 2 * w + c];
            }
            sB[idx] = v;
        }
        // staging tiles: rows that are never written hold their constants (zero; row 15 of the input tile = the ones)
        static_assert(SM::oT % 4 == 0 && SM::TW % 4 == 0 && PBW % 4 == 0, "16-byte zero fill");
        for (int idx = 4 * tid; idx < NWAVES * SM::TW; idx += 4 * NT) {
            const int row = (idx % SM::TW) / PBW;                       // 0-15 sZ, 16-31 sH, 32-34 sD
            const float v = (row == 16 + 15) ? 1.0f : 0.0f;
            *reinterpret_cast<f32x4*>(smem + SM::oT + idx) = f32x4{v, v, v, v};
        }
        for (int idx = 4 * tid; idx < SM::NACC * SM::SLOT; idx += 4 * NT) *reinterpret_cast<f32x4*>(smem + SM::oA + idx) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    __syncthreads();

    float* const sZ = smem + SM::oT + wv * SM::TW;     // dZ_l                          [slot][observation]
    float* const sD = sZ + 2 * 16 * PBW;               // rows 0, 1: dL/dloc, dL/draw; row 2: zeros   (sH = sZ + 16 PBW: the layer's input)

    // ---- accumulators that live across all tiles of this wave ----------------------------------------------------------
    f32x4 wacc[LREG];                   // dW_l^T of layer l < LREG; the upper layers' live in LDS (acc_slot)
    f32x4 wacc_h = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int l = 0; l < LREG; ++l) wacc[l] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4* const sAcc = reinterpret_cast<f32x4*>(smem + SM::oA) + wv * 64 + lane;          // this lane's float4 of layer LREG's slot
    auto acc_slot = [&](int l) -> f32x4& { return sAcc[(l > LREG ? l - LREG : 0) * (NWAVES * 64)]; };
    float nll_acc = 0.0f;
    cl_ev11 ev = {1.0f, 0.0f, 0.0f};
    float ev_g0 = 0.0f, ev_g1 = 0.0f, ev_g2 = 0.0f;
    const bool use_ev11 = A.ev11 != nullptr;
    if (use_ev11) { ev.sdfac = cl_softplus(A.ev11[0]); ev.sdadd = cl_softplus(A.ev11[1]); ev.sdb = cl_softplus(A.ev11[2]); }

    const int n_wt = (A.n_obs + WT - 1) / WT;                        // wave tiles
    const int wt_step = (int)gridDim.x * NWAVES;
    // loop-invariant per-lane byte offsets of the metadata loads (one per group; the row / tile part is wave-uniform)
    unsigned meta_off[G];
#pragma unroll
    for (int g = 0; g < G; ++g) meta_off[g] = 4u * ((unsigned)q * (unsigned)A.n_pad + (unsigned)(16 * g + j));

    // per-observation inputs of a wave tile, loaded one tile ahead (plain loads; a wave tile never leaves the padded metadata rows
    // because 16 G divides CL_MLP_TILE, the per-observation arrays are clamped to their last element)
    float xn[G][KS];
    int ridn = -1, imgn = 0;
    float ion = 0.0f, sgn = 1.0f;
    auto prefetch = [&](int wt_in, cl_args_p E) {
        const int wt = uniform(wt_in);
        const int base = wt * WT;
        const unsigned n_pad_u = (unsigned)E->n_pad;
        const int last_obs = E->n_obs - 1;
        const float* __restrict__ mt = E->meta_t + base;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            // k-steps past the rows of meta_t (cl_mlp_meta_rows) re-read step 0 and are zeroed: no branch, every address valid
            const bool have = t < ks1;                                   // wave-uniform
            const float* row = mt + (have ? (size_t)(4 * t) * n_pad_u : (size_t)0);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float v = ld_uo(row, meta_off[g]);
                xn[g][t] = have ? v : 0.0f;
            }
        }
        const int o = base + (lane & (WT - 1));            // (the upper lanes mirror the lower ones' observation: they take the odd MC samples)
        const bool ok = o <= last_obs;
        const unsigned ob = 4u * (unsigned)min(o, last_obs);
        const int rr = ld_uo(E->refl_id, ob);
        ridn = ok ? rr : -1;
        ion = ld_uo(E->iobs, ob);
        const float ss = ld_uo(E->sig, ob);
        sgn = ok ? ss : 1.0f;
        imgn = E->use_img ? ld_uo(E->image_id, ob) : 0;
    };
    const int wt_begin = (int)blockIdx.x * NWAVES + wv;
    if (wt_begin < n_wt) prefetch(wt_begin, kernargs_again());
    const float* const wrow = sW + j * NPW + 4 * q;        // forward A operands: image row j, slots 4q .. 4q+3 (one ds_read_b128)
    const float* const wcol = sW + (4 * q) * NPW + j;      // dgrad A operands: image rows 4q + t, column j
    float* const stw = sZ + (4 * q) * PBW + j;             // staging writes: rows 4q + t, this lane's observation column (+ 16 g)
    const float* const strd = sZ + j * PBW + 4 * q;        // staging reads: row j, observations 4q .. 4q+3 (+ 16 g): one ds_read_b128
    constexpr int TH = 16 * PBW;                           // offset of the input tile (sH) from sZ
    const float* const strdD = sD + (j < 2 ? j : 2) * PBW + 4 * q;   // head wgrad A operand: row j of [dloc; draw; 0; 0; ...]

#ifdef CL_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    for (int wt = wt_begin; wt < n_wt; wt += wt_step) {
        const int Lt = opaque_uniform(L);
        const unsigned topm = top_layer_mask(L);
        float x0[G][KS];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < KS; ++t) x0[g][t] = (4 * t + q == 15) ? 1.0f : xn[g][t];     // metadata slot 15 = the ones (KS = 4)
        const int rid = ridn, img = imgn;
        const float io = ion, sg = sgn;
        float aim = 1.0f, zf0 = 0.0f;
        int rme = 0, gm = 0;                 // PACKED: the caller's row of this lane's packed row; (member index | group size << 8)
        long long nkey = 0;                  // PACKED: noise key of this lane's row
        {
            // gathers that depend on the prefetched ids: issued now, consumed in the epilogue
            cl_args_p E0 = kernargs_again();
            if (rid >= 0) {
                if (E0->use_img && img > 0) aim = ld_uo(E0->img, 4u * (unsigned)(img - 1));
                const int S0 = E0->S;
                if ((lane >> 5) < S0) zf0 = ld_uo(E0->z_f, 4u * ((unsigned)rid * (unsigned)S0 + (unsigned)(lane >> 5)));   // this lane's first sample
                if (PACKED) {
                    const unsigned pb = 4u * (unsigned)(wt * WT + (lane & (WT - 1)));
                    rme = ld_uo(E0->row_map, pb);
                    if (E0->gmeta != nullptr) gm = ld_uo(E0->gmeta, pb);
                    nkey = (E0->noise_row != nullptr) ? (long long)ld_uo(E0->noise_row, pb) : E0->obs_offset + rme;
                } else if (E0->noise_row != nullptr) {
                    // plain layout over rows that are not a contiguous range of the caller's (reflection-owner shard): the row's global number
                    nkey = (long long)ld_uo(E0->noise_row, 4u * (unsigned)(wt * WT + (lane & (WT - 1))));
                }
            }
        }
        NSTAMP(0);
        // ================= forward ==========================================================================================
        // `top` = the top layer's activations (the head's input), copied out where the depth puts them.
        float hs[NL][G][KS];
        float top[G][KS];
        f32x4 wfn = *reinterpret_cast<const f32x4*>(wrow), biasn = *reinterpret_cast<const f32x4*>(sB + 4 * q);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (l < Lt) {
                const f32x4 wf = wfn, bias = biasn;
                if (l + 1 < NL) {            // the next layer's operands, in flight under this layer's MFMAs (zero images past the depth)
                    wfn = *reinterpret_cast<const f32x4*>(wrow + (l + 1 < NL ? l + 1 : 0) * 16 * NPW);
                    biasn = *reinterpret_cast<const f32x4*>(sB + (l + 1 < NL ? l + 1 : 0) * 16 + 4 * q);
                }
                f32x4 acc[G];
                // the groups' MFMA chains interleaved step by step (dependent MFMAs of one chain are G issue slots apart), then the
                // LeakyReLUs of all groups: the wave's vector phase runs beside its SIMD partner's MFMAs
#pragma unroll
                for (int t = 0; t < KS; ++t) {
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float b = (l == 0) ? x0[g][t] : hs[l > 0 ? l - 1 : 0][g][t];
                        acc[g] = mfma4(wf[t], b, t == 0 ? bias : acc[g]);
                    }
                }
                NFENCE();
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int t = 0; t < KS; ++t) hs[l][g][t] = lrelu(acc[g][t], leak);
                NFENCE();
                if (topm & (1u << l)) {
#pragma unroll
                    for (int g = 0; g < G; ++g)
#pragma unroll
                        for (int t = 0; t < KS; ++t) top[g][t] = hs[l][g][t];
                }
            }
        }
        NSTAMP(1);
        // Dense(2) head: group g's two outputs land in rows 4g, 4g+1; two accumulator chains (even / odd groups)
        f32x4 acc_h = *reinterpret_cast<const f32x4*>(sB + NL * 16 + 4 * q);
        f32x4 acc_h2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 wh = *reinterpret_cast<const f32x4*>(wrow + (NL + g) * 16 * NPW);
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                if (g & 1) acc_h2 = mfma4(wh[t], top[g][t], acc_h2);
                else acc_h = mfma4(wh[t], top[g][t], acc_h);
            }
        }
        acc_h += acc_h2;

        // ================= epilogue: lane = observation; sample, predict, likelihood, dL/d(loc, raw) ==========================
        cl_args_p E = kernargs_again();              // the epilogue's and the prefetch's arguments, loaded here (see kernargs_again)
        const int S = E->S;
        const float w_ll = E->w_ll;
        // lane map: observation = lane & 31 (the head left (loc, raw sigma) of observation e in lane e < 32), MC samples s = half,
        // half + 2, ... with half = lane >> 5: the two lanes of an observation split its samples (one lane half idles when S = 1)
        const int half = lane >> 5;
        const long long gobs = PACKED ? (long long)rme : (long long)wt * WT + (lane & (WT - 1));      // this lane's observation in the caller's order
        const unsigned zoff = 4u * (unsigned)(rid < 0 ? 0 : rid) * (unsigned)S;
        const bool det = E->dzf_obs != nullptr;                            // wave-uniform: deterministic mode (stores instead of float atomics)
        // this observation's record in dzf_obs (< 4 GiB: cl_launch_narrow): its own row, or the slot the caller assigns it (det_slot)
        const unsigned dob = 4u * (unsigned)((det && E->det_slot != nullptr && rid >= 0) ? E->det_slot[gobs] : (int)gobs) * (unsigned)S;
        float o0 = acc_h[0], o1 = acc_h[1];
        if (S > 1) {                                     // wave-uniform
            o0 = __shfl(o0, lane & 31);
            o1 = __shfl(o1, lane & 31);
        }
        float dsig_draw;
        const float sigma = cl_scale_bij(o1, E->bij_kind, E->eps, &dsig_draw);
        float pdl = 0.0f, pds = 0.0f, pda = 0.0f;
        // single-pass Laue (careless/models/likelihoods/laue.py:20-34): the predictions of the rows of one harmonic group SUM before
        // the likelihood.  The members of a group are consecutive lanes; every lane collects its group's total with shuffles over a
        // wave-uniform member count, every member evaluates the same likelihood derivative, member 0 alone counts the log-likelihood.
        const bool laue = PACKED && E->gmeta != nullptr;                    // wave-uniform
        const int mem = gm & 0xff, cnt = gm >> 8;
        const int gmax = laue ? uniform(E->tile_gmax[(wt * WT) / CL_MLP_TILE]) : 0;
        if (laue || rid >= 0) {
            // hardware reciprocal and logarithm (1 ulp): sigma is an input, its log enters the NLL additively
            const float inv_sg = cl_fast_rcp(sg);
            const float log_sg = cl_fast_log(sg);
            const float* __restrict__ eta_p = E->eta ? E->eta + (size_t)gobs * S : nullptr;
            float* __restrict__ ipred_p = E->ipred_out ? E->ipred_out + (size_t)gobs * S : nullptr;
            const float* __restrict__ zf_p = E->z_f;
            float* __restrict__ dzf_p = E->dz_f;
            const int lik_kind = E->lik_kind;
            const float dof = E->dof, lik_const = E->lik_const, shift = E->shift;
            float esin[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const int nS = (S + 1) >> 1;                 // wave-uniform trip count (all lanes take part in the Laue shuffles)
            for (int k = 0; k < nS; ++k) {
                const int s = 2 * k + half;
                const bool act = rid >= 0 && s < S;
                float eta = 0.0f;
                if (!act) {
                } else if (eta_p != nullptr) {
                    eta = eta_p[s];
                } else if (((s >> 2) & 1) == 0) {        // one Philox block + Box-Muller pair serves samples s and s + 4
                    float sn;
                    cl_noise_normal_pair(E->seed, E->step, (uint32_t)s, (uint64_t)((PACKED || E->noise_row != nullptr) ? nkey : E->obs_offset + gobs), &eta, &sn);
                    const int kk = s & 3;
                    if (kk == 0) esin[0] = sn; else if (kk == 1) esin[1] = sn; else if (kk == 2) esin[2] = sn; else esin[3] = sn;
                } else {
                    const int kk = s & 3;
                    eta = (kk == 0) ? esin[0] : (kk == 1) ? esin[1] : (kk == 2) ? esin[2] : esin[3];
                }
                const float zf = !act ? 0.0f : ((k == 0) ? zf0 : ld_uo(zf_p, zoff + 4u * s));
                const float tq = o0 + sigma * eta + shift;
                const float zs = aim * tq;
                const float ipred = act ? zs * zf * zf : 0.0f;
                if (act && ipred_p) ipred_p[s] = ipred;
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
                        ll = cl_lik_log_prob2(lin, io, inv_sg, log_sg, lik_kind, dof, lik_const, &dll);
                    }
                    if (counts) nll_acc -= ll * w_ll;
                    const float gi = -dll * w_ll;                 // dNLL / d ipred (of every member of the group)
                    const float dzs = gi * zf * zf;
                    // deterministic mode (include/careless_hip.h: dzf_obs): the contribution is STORED per (observation, sample) and
                    // cl_det_reduce sums the observations of a reflection in row order; otherwise a float atomic
                    if (det) *reinterpret_cast<float*>(reinterpret_cast<char*>(E->dzf_obs) + dob + 4u * s) = gi * zs * 2.0f * zf;
                    else atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(dzf_p) + zoff + 4u * s), gi * zs * 2.0f * zf);
                    const float dt = dzs * aim;
                    pdl += dt;
                    pds += dt * eta;
                    pda += dzs * tq;
                }
            }
        }
        if (S > 1) {                                     // the two lanes of an observation add up their samples' sums
            pdl += __shfl_xor(pdl, 32);
            pds += __shfl_xor(pds, 32);
            pda += __shfl_xor(pda, 32);
        }
        if (E->use_img && det) {
            if (rid >= 0 && lane < WT) E->dimg_obs[gobs] = pda;          // summed per image, in row order, by cl_det_reduce
        } else if (E->use_img) {
            // image ids are sorted: the observations of a wave tile almost always share one image -> ONE atomic per wave
            const int img0 = uniform(img);
            if (__all(img == img0 || rid < 0)) {
                const float v = cl_wave_sum((rid >= 0 && lane < WT) ? pda : 0.0f);
                if (lane == 0 && img0 > 0) atomicAdd(E->d_img + (img0 - 1), v);
            } else {
                cl_image_grad_segments(E->d_img, img, pda, rid >= 0 && img > 0 && lane < WT, lane);
            }
        }
        const float dloc = pdl, draw = pds * dsig_draw;              // zero for padding observations
        if (lane < WT) {
            sD[lane] = dloc;
            sD[PBW + lane] = draw;
        }

        NSTAMP(2);
        // next tile's inputs: their latency hides under the backward pass
        if (wt + wt_step < n_wt) prefetch(wt + wt_step, E);
        NSTAMP(3);

        // ================= backward =========================================================================================
        // The weight gradient of a layer is issued one layer late: its operands (the staged dZ_l and the layer's input, read
        // back transposed) are requested right after the staging writes and consumed by MFMAs that alternate with the next
        // layer's dgrad MFMAs, when they have long arrived.  One wave's LDS operations execute in order and a wave only
        // touches its own staging columns, so nothing separates the writes of a layer from the reads before them.
        // The vector work of group g+1 (its dZ) and the staging of group g fill the gaps between group g's MFMAs.
        f32x4 pa[G], pb[G];                 // operands of the pending weight gradient
        f32x4 dH[G];                        // dL/d(output of the layer at hand), per group
        float dz[KS];                       // dZ of the group whose MFMAs come next
        // head: stage the top activations, request the head's wgrad operands; dgrad from the epilogue's (dloc, draw) as they
        // stand (8 MFMAs, under which the operands arrive); then the head's wgrad
#pragma unroll
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int t = 0; t < KS; ++t) (stw + TH)[t * PBW + 16 * g] = top[g][t];
            pa[g] = *reinterpret_cast<const f32x4*>(strdD + 16 * g);                // sD (rows >= 2 of the head's "dZ" are zero)
            pb[g] = *reinterpret_cast<const f32x4*>(strd + TH + 16 * g);            // sH
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
            a = mfma4(wcol[(NL + g) * 16 * NPW], dloc, a);
            a = mfma4(wcol[(NL + g) * 16 * NPW + NPW], draw, a);
            dH[g] = a;
        }
        NFENCE();
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) wacc_h = mfma4(pa[g][t], pb[g][t], wacc_h);
#pragma unroll
        for (int t = 0; t < KS; ++t) dz[t] = lrelu_bwd(top[0][t], dH[0][t], leak);
        // the top layer has no pending weight gradient: zero operands, its MFMAs add exactly 0 to an accumulator
#pragma unroll
        for (int g = 0; g < G; ++g) { pa[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; pb[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
        // Requested one layer ahead, into TWO buffers used alternately by layer parity (a single carried buffer costs a register
        // copy per element and layer where the skipped-layer paths join): the LDS-resident accumulator of the pending weight
        // gradient and the dgrad weight operands.  The depth decides which parity the top layer has, so both start out equal.
        f32x4 paccb[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
        if (SM::NACC > 0) { paccb[0] = acc_slot(Lt < NL ? Lt : NL - 1); paccb[1] = paccb[0]; }     // (run-time slot; unused when Lt < LREG or Lt == NL)

        NSTAMP(4);
        float wdb[2][KS];
        {
            const int lt1 = Lt > 1 ? Lt - 1 : 1;
#pragma unroll
            for (int t = 0; t < KS; ++t) { wdb[0][t] = wcol[lt1 * 16 * NPW + t * NPW]; wdb[1][t] = wdb[0][t]; }
        }
#pragma unroll
        for (int l = NL - 1; l >= 0; --l) {
            if (l < Lt) {
                float (&wd)[KS] = wdb[l & 1];
                if (l > 1) {
#pragma unroll
                    for (int t = 0; t < KS; ++t) wdb[(l - 1) & 1][t] = wcol[(l > 1 ? l - 1 : 1) * 16 * NPW + t * NPW];
                }
                // accumulator of the pending weight gradient (layer l+1; above the top layer: the head's, which receives zeros)
                const int lp = l + 1;
                const bool lp_head = lp >= NL;                       // compile-time after unrolling
                const bool lp_lds = !lp_head && lp >= LREG;
                f32x4 wa = lp_head ? wacc_h : (lp_lds ? paccb[lp & 1] : wacc[lp < LREG ? lp : 0]);
                if (SM::NACC > 0 && l >= LREG) paccb[l & 1] = acc_slot(l);              // next layer's slot, in flight under this layer's MFMAs
                f32x4 dHn[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float dzn[KS];
                    f32x4 accd = {0.0f, 0.0f, 0.0f, 0.0f};
                    NFENCE();
                    // MFMAs of group g: the pending wgrad (layer l+1) and this layer's dgrad, alternating chains
                    wa = mfma4(pa[g][0], pb[g][0], wa);
                    if (l > 0) accd = mfma4(wd[0], dz[0], accd);
#pragma unroll
                    for (int t = 0; t < KS; ++t) stw[t * PBW + 16 * g] = dz[t];                              // stage dZ_l
                    if (l == 0) {
                        // dL/d(pre-activations of layer 0) out (round 5: the launch behind a peeled first layer, elbo_peel.hip): lane (j, q),
                        // step t = feature 4 t + q of observation 16 g + j; rows [cl_mlp_meta_rows(w)][n_pad] like meta_t.  Re-read per
                        // tile: one scalar load.
                        cl_args_p E3 = kernargs_again();
                        float* const dxo = E3->dZ0_out;
                        if (dxo != nullptr) {
                            const size_t np = (size_t)E3->n_pad;
                            float* const p0 = dxo + (size_t)(wt * WT + 16 * g + j);
#pragma unroll
                            for (int t = 0; t < KS; ++t)
                                if (4 * t + q < w) p0[(size_t)(4 * t + q) * np] = dz[t];
                        }
                    }
                    NFENCE();
                    wa = mfma4(pa[g][1], pb[g][1], wa);
                    if (l > 0) accd = mfma4(wd[1], dz[1], accd);
#pragma unroll
                    for (int t = 0; t < KS; ++t) (stw + TH)[t * PBW + 16 * g] = (l == 0) ? x0[g][t] : hs[l > 0 ? l - 1 : 0][g][t];   // stage the layer's input
                    NFENCE();
                    wa = mfma4(pa[g][2], pb[g][2], wa);
                    if (l > 0 && KS > 2) accd = mfma4(wd[KS > 2 ? 2 : 0], dz[KS > 2 ? 2 : 0], accd);
                    // dZ of the group that comes next in this layer
                    if (g + 1 < G) {
#pragma unroll
                        for (int t = 0; t < KS; ++t)
                            dzn[t] = lrelu_bwd(hs[l][g + 1 < G ? g + 1 : 0][t], dH[g + 1 < G ? g + 1 : 0][t], leak);
                    }
                    NFENCE();
                    wa = mfma4(pa[g][3], pb[g][3], wa);
                    if (l > 0 && KS > 3) accd = mfma4(wd[KS > 3 ? 3 : 0], dz[KS > 3 ? 3 : 0], accd);
                    pa[g] = *reinterpret_cast<const f32x4*>(strd + 16 * g);                                  // operands of this layer's wgrad
                    pb[g] = *reinterpret_cast<const f32x4*>(strd + TH + 16 * g);
                    NFENCE();
                    dHn[g] = accd;
                    if (g + 1 < G) {
#pragma unroll
                        for (int t = 0; t < KS; ++t) dz[t] = dzn[t];
                    } else if (l > 0) {          // group 0 of the layer below
#pragma unroll
                        for (int t = 0; t < KS; ++t) dz[t] = lrelu_bwd(hs[l > 0 ? l - 1 : 0][0][t], dHn[0][t], leak);
                    }
                }
                if (lp_head) wacc_h = wa;
                else if (lp_lds) acc_slot(lp) = wa;
                else wacc[lp < LREG ? lp : 0] = wa;
#pragma unroll
                for (int g = 0; g < G; ++g) dH[g] = dHn[g];
            }
        }
        NSTAMP(5);
        // the last pending weight gradient: layer 0
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) wacc[0] = mfma4(pa[g][t], pb[g][t], wacc[0]);
        NSTAMP(6);
    }
#ifdef CL_STAMPS
    if (A.loc_out != nullptr && lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(A.loc_out) + ((size_t)blockIdx.x * NWAVES + wv) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) dbg[k] = st_acc[k];
    }
#endif

    // ================= flush: sum the waves' accumulators, scatter into the flat W^T layout of this workgroup's partial ====
    __syncthreads();
    const int offWo = w * d + w + (L - 1) * (w * w + w);
    const int Ptot = offWo + 2 * w + 2;
    // fixed binary tree over the waves (deterministic, log2(NWAVES) + 1 barriers): at stride s the waves with (wv & (2s - 1)) == s
    // park their sums in the region of wave wv - s, which adds them to its own
    constexpr int REG = (NL + 1) * 256;                    // floats of one wave's parked accumulators
    static_assert((NWAVES / 2) * REG * 4 <= 160 * 1024, "flush regions");
    f32x4 facc[NL + 1];
#pragma unroll
    for (int l = 0; l <= NL; ++l) facc[l] = (l == NL) ? wacc_h : ((l >= LREG) ? acc_slot(l) : wacc[l < LREG ? l : 0]);
    __syncthreads();                                       // every wave has read its LDS slots: the regions may overwrite them
#pragma unroll
    for (int s2 = 1; s2 < NWAVES; s2 <<= 1) {
        f32x4* reg = reinterpret_cast<f32x4*>(smem + ((wv & ~(2 * s2 - 1)) / (2 * s2)) * REG) + lane;
        if ((wv & (2 * s2 - 1)) == s2) {
#pragma unroll
            for (int l = 0; l <= NL; ++l) reg[l * 64] = facc[l];
        }
        __syncthreads();
        if ((wv & (2 * s2 - 1)) == 0) {
#pragma unroll
            for (int l = 0; l <= NL; ++l) facc[l] += reg[l * 64];
        }
        __syncthreads();
    }
    if (wv == 0) {
#pragma unroll
        for (int l = 0; l <= NL; ++l) reinterpret_cast<f32x4*>(smem + l * 256)[lane] = facc[l];
    }
    __syncthreads();
    float* __restrict__ part = A.partials + (size_t)blockIdx.x * Ptot;
    for (int idx = tid; idx < (NL + 1) * 256; idx += NT) {
        const int l = idx >> 8, r = idx & 255;                     // accumulator l < NL: Dense layer l; NL: the head
        const int ln = r >> 2, t = r & 3;
        const int os = 4 * (ln >> 4) + t, is = ln & 15;           // output slot (row), input slot (column) of this element
        const float v = smem[idx];
        if (l < L) {
            const int fo = slot_of(os), fi = slot_of(is);
            const int in_dim = (l == 0) ? d : w;
            const int off = (l == 0) ? 0 : (w * d + w + (l - 1) * (w * w + w));
            if (fo < w && fi < in_dim) part[off + fo * in_dim + fi] = v;
            if (fo < w && is == 15) part[off + w * in_dim + fo] = v;                      // bias gradient: the ones column
        } else if (l == NL) {
            const int fi = slot_of(is);
            if (os < 2 && fi < w) part[offWo + os * w + fi] = v;
            if (os < 2 && is == 15) part[offWo + 2 * w + os] = v;
        }
    }

    {
        float v = nll_acc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        __syncthreads();
        if (lane == 0) smem[wv] = v;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int k = 0; k < NWAVES; ++k) t += (double)smem[k];
            if (A.nll_part != nullptr) A.nll_part[blockIdx.x] = t;       // deterministic mode: every workgroup stores its slot, cl_det_reduce adds them in index order
            else atomicAdd(A.scalars + CL_SC_NLL, t);
        }
        if (use_ev11) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                ev_g0 += __shfl_xor(ev_g0, off); ev_g1 += __shfl_xor(ev_g1, off); ev_g2 += __shfl_xor(ev_g2, off);
            }
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
}

template <int G, int KS, int NWAVES, bool PACKED>
static int launch_narrow_one(const cl_mlp_args& a, int grid, hipStream_t st) {
    using SM = NSmem<G, NWAVES>;
    const size_t sm = (size_t)SM::total * sizeof(float);
    if (sm > 160 * 1024) return -3;
    auto kern = elbo_narrow_kernel<G, KS, NWAVES, PACKED>;
    static std::atomic<size_t> configured{0};
    size_t have = configured.load(std::memory_order_acquire);
    if (have < sm) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) return (int)e;
        while (have < sm && !configured.compare_exchange_weak(have, sm, std::memory_order_release, std::memory_order_acquire)) {}
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWAVES), sm, st, a);
    return (int)hipGetLastError();
}

// 1 = this geometry runs on the narrow kernel (full ELBO step; plain observation layout, or the packed one of single-pass Laue)
int cl_narrow_supports(const cl_mlp_args& a) {
    return a.w >= 1 && a.w <= 15 && a.d >= 1 && a.d <= 15 && a.L >= 1 && a.L <= NL && a.n_imgl == 0 && a.act_out == nullptr &&
           a.dH_ext == nullptr && a.dX_out == nullptr && (a.row_map != nullptr || a.gmeta == nullptr);
}

// name of the instance cl_launch_narrow runs (cl_mlp_kernel_name)
int cl_narrow_kernel_name(const cl_mlp_args& a, char* out, size_t n) {
    const int m = a.w > a.d ? a.w : a.d;
    return snprintf(out, n, "elbo_narrow_kernel<2, %d, 8, %s>%s", m <= 8 ? 2 : (m <= 12 ? 3 : 4), a.row_map != nullptr ? "true" : "false",
                    a.dzf_obs != nullptr ? " (deterministic stores)" : "");
}

template <bool PACKED>
static int launch_narrow_ks(const cl_mlp_args& a, int grid, hipStream_t st) {
    const int m = a.w > a.d ? a.w : a.d;               // the metadata layer takes the same number of k-steps as the hidden ones
    if (m <= 8) return launch_narrow_one<2, 2, 8, PACKED>(a, grid, st);
    if (m <= 12) return launch_narrow_one<2, 3, 8, PACKED>(a, grid, st);
    // widths 13 .. 15: four k-steps.  Two waves per SIMD spill ~90 registers at that size; a one-wave-per-SIMD instance (<2, 4, 4>: no spill)
    // measured no faster in round 3 (DESIGN / NOTEBOOK: 20 x 13 1.60 vs 1.56 ms) and is not instantiated any more
    return launch_narrow_one<2, 4, 8, PACKED>(a, grid, st);
}

int cl_launch_narrow(const cl_mlp_args& a, int grid, hipStream_t st) {
    if (!cl_narrow_supports(a)) return -2;
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
    if (a.row_map != nullptr) {
        if (a.n_obs != a.n_pad || (a.gmeta != nullptr && a.tile_gmax == nullptr)) return -1;
        if ((a.eta != nullptr || a.ipred_out != nullptr) && 4ull * (unsigned long long)a.n_pad * (unsigned long long)a.S >= (1ull << 32)) return -4;
        return launch_narrow_ks<true>(a, grid, st);
    }
    return launch_narrow_ks<false>(a, grid, st);
}
