// The data term of a step whose scaling model is FROZEN (gfx950 / CDNA4 only; round 6): `careless mono | poly --freeze-scales` and the
// half-dataset trainings of `--merge-half-datasets` (reference careless/careless.py:48-50, 102-128 -- as many steps again as the main
// training).  With the scaler frozen its output (loc, sigma) and the image scale of every observation are constants of the training:
// the caller takes them once (cl_mlp_forward / the wide path) and per step only what depends on the sampled amplitudes remains --
// sample the scale, predict, log-prob, gradient to dz_f and the Evans-2011 terms:
//   careless/models/merging/variational.py:156-181 (predict), models/likelihoods/mono.py:10-73 (log-prob), variational.py:197-202
//   (gradient of the trainable variables only: the scaler's is not taken).
//
// Round 5 ran this on cl_slot_rows (elbo_laue.hip): a thread per (row, sample) in the caller's row order and one memory-side float atomic
// of ~96 B per (row, sample) into dz_f -- 7 - 10 % of the HBM roof.  Here the CALLER sorts the rows by reflection once (the scaler's
// output is a constant: no tile, image or harmonic-group constraint binds the order) and
//   * a thread owns a ROW and loops over the samples: the row's seven numbers are read once (28 B, coalesced), one Philox block +
//     Box-Muller pair serves samples s and s + 4 as in the fused kernels (the noise is keyed by the row's GLOBAL number: bit for bit
//     the fused step's draws);
//   * the amplitude gradients of equal-reflection runs are summed inside the wave (rows are sorted: a run is a range of lanes; six
//     shuffle steps, the run's first lane ends with its total) and leave as ONE plain store per (reflection, sample) -- no atomics:
//     two runs of the same reflection can only meet at a wave boundary, those partial sums go to an edge buffer that a second small
//     launch adds up in wave order.  The step's data term is therefore DETERMINISTIC by construction;
//   * NLL in fp64 per workgroup, one atomic each (bounded grid), Evans-2011 gradients per wave.
// Roofline: HBM; algorithmic bytes per row 28 (+ 4 S per reflection, read and written once; + 4 S with injected noise).  With 8 samples
// the vector ALU (Philox + Box-Muller + Student-t per (row, sample)) is within a factor two of the HBM time: DESIGN 5.1b.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "cl_math.h"
#include "cl_kernels.h"

namespace {

constexpr int FB = 256;                 // threads of a workgroup = rows of a chunk
// samples of a batch (registers): template parameter SB -- 1 for --mc-samples 1 (the default: no arrays of eight in the register
// budget, twice the waves per SIMD), 8 otherwise
constexpr int THROUGH = 1 << 30;        // edge_rid flag: the wave is ONE run that continues on both sides

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

}  // namespace

template <int SB>
__global__ __launch_bounds__(FB) void frozen_rows_kernel(const cl_frozen_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const int lane = threadIdx.x & 63;
    const int S = A.S;
    const long long n = A.n;
    const long long chunks = (n + FB - 1) / FB;
    double nll = 0.0;
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    cl_ev11 ev = {1.0f, 0.0f, 0.0f};
    const bool use_ev11 = A.ev11 != nullptr;
    if (use_ev11) { ev.sdfac = cl_softplus(A.ev11[0]); ev.sdadd = cl_softplus(A.ev11[1]); ev.sdb = cl_softplus(A.ev11[2]); }
    const float inv_dof = (A.lik_kind == CL_LIK_STUDENTT) ? 1.0f / A.dof : 0.0f;

    // a row's seven numbers + the two reflection ids beside its wave, requested one chunk ahead: the chain row -> reflection -> amplitude is
    // two dependent loads deep and a thread has ~20 chunks to walk
    // `src` given (harmonic groups, second pass): the row's amplitude gradients were made by frozen_laue_kernel and wait in gbuf[src]: this
    // launch only sums them per reflection
    const bool gather = A.gbuf != nullptr;                 // (this kernel is only launched without gmeta: the second pass of harmonic groups)
    struct Row { int rid, rid_before, rid_after; float loc, sigma, io, sg, aim; long long key; };
    auto fetch = [&](long long c) -> Row {
        Row r;
        const long long row = c * FB + threadIdx.x, row0 = (row >> 6) << 6;
        const long long rc = row < n ? row : n - 1;
        r.rid = A.refl_id[rc];
        if (gather) {
            r.loc = r.sigma = r.io = 0.0f; r.sg = r.aim = 1.0f;
            r.key = A.src != nullptr ? (long long)A.src[rc] : rc;
        } else {
            r.loc = A.loc[rc]; r.sigma = A.sigma[rc]; r.io = A.iobs[rc]; r.sg = A.sig[rc];
            r.aim = A.aim != nullptr ? A.aim[rc] : 1.0f;
            r.key = A.key != nullptr ? (long long)A.key[rc] : A.obs_offset + row;
        }
        r.rid_before = (row0 > 0 && row0 <= n) ? A.refl_id[row0 - 1] : -1;
        r.rid_after = row0 + 64 < n ? A.refl_id[row0 + 64] : -1;
        return r;
    };
    Row nxt = fetch(blockIdx.x < chunks ? blockIdx.x : 0);
    for (long long c = blockIdx.x; c < chunks; c += gridDim.x) {
        const Row cur = nxt;
        if (c + gridDim.x < chunks) nxt = fetch(c + gridDim.x);
        const long long row = c * FB + threadIdx.x;
        const long long wv = row >> 6;                       // wave of the launch this row belongs to (edge record)
        const long long row0 = wv << 6;
        const bool in = row < n;
        int rid = cur.rid;
        if (!in) rid = -1;
        const bool act = rid >= 0;
        const float loc = cur.loc, sigma = cur.sigma, io = cur.io, sg = cur.sg, aim = cur.aim;
        const long long key = cur.key;
        // the runs of this wave: a lane adds the lane `off` above it while that lane belongs to the same reflection
        bool m[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int off = 1 << k;
            const int r2 = __shfl_down(rid, off);
            m[k] = (lane + off < 64) && r2 == rid;
        }
        const int prev = __shfl_up(rid, 1);
        const bool head = act && (lane == 0 || prev != rid);
        // ... and across the wave's two borders (wave-uniform)
        const int rid_first = __builtin_amdgcn_readlane(rid, 0), rid_last = __builtin_amdgcn_readlane(rid, 63);
        const int rid_before = cur.rid_before, rid_after = cur.rid_after;
        const bool first_cp = rid_first >= 0 && rid_first == rid_before;             // the first run continues the previous wave's last one
        const bool last_cn = rid_last >= 0 && rid_last == rid_after;                 // the last run continues in the next wave
        const bool single = rid_first == rid_last;
        const bool rec1 = last_cn && !(single && first_cp);                          // the last run STARTS here and goes on: head of a chain
        if (lane == 0 && row0 < n) {
            A.edge_rid[2 * wv] = first_cp ? (rid_first | ((single && last_cn) ? THROUGH : 0)) : -1;
            A.edge_rid[2 * wv + 1] = rec1 ? rid_last : -1;
        }
        // where this lane's run total goes (meaningful on head lanes): 0 dz_f, 1 edge record 0, 2 edge record 1
        const int route = (lane == 0 && first_cp) ? 1 : ((rid == rid_last && rec1) ? 2 : 0);
        const float inv_sg = cl_fast_rcp(sg), log_sg = cl_fast_log(sg);
        const float* __restrict__ eta_p = (A.eta != nullptr && !gather) ? A.eta + (size_t)(key - A.obs_offset) * S : nullptr;
        float* __restrict__ ip_p = (A.ipred_out != nullptr && !gather) ? A.ipred_out + (size_t)(key - A.obs_offset) * S : nullptr;
        const size_t zoff = (size_t)(act ? rid : 0) * S;

        for (int sb = 0; sb < S; sb += SB) {
            float e[SB], g[SB];
            if (gather) {
#pragma unroll
                for (int j = 0; j < SB; ++j) { e[j] = 0.0f; g[j] = (act && sb + j < S) ? A.gbuf[(size_t)key * S + sb + j] : 0.0f; }
            } else if (eta_p == nullptr) {
                if constexpr (SB == 1) {
                    float unused;
                    cl_noise_normal_pair(A.seed, A.step, 0u, (uint64_t)key, &e[0], &unused);
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        if (sb + p < S) cl_noise_normal_pair(A.seed, A.step, (uint32_t)(sb + p), (uint64_t)key, &e[p], &e[p + 4]);
                        else { e[p] = 0.0f; e[p + 4] = 0.0f; }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < SB; ++j) e[j] = (act && sb + j < S) ? eta_p[sb + j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                if (gather) break;                                 // (wave-uniform)
                g[j] = 0.0f;
                if (sb + j < S && act) {                           // (the first test is wave-uniform)
                    const int s = sb + j;
                    const float zf = A.z_f[zoff + s];
                    const float tq = loc + sigma * e[j] + A.shift;
                    const float ipred = aim * tq * zf * zf;
                    if (ip_p != nullptr) ip_p[s] = ipred;
                    float dll, ll;
                    if (use_ev11) {
                        float gf, gb, ga;
                        ll = cl_lik_ev11(ipred, io, sg, A.lik_kind, A.dof, A.lik_const, ev, &dll, &gf, &gb, &ga);
                        g0 -= gf * A.w_ll; g1 -= ga * A.w_ll; g2 -= gb * A.w_ll;      // order: Sdfac, Sdadd, SdB
                    } else {
                        ll = cl_lik_log_prob3(ipred, io, inv_sg, log_sg, A.lik_kind, A.dof, inv_dof, A.lik_const, &dll);
                    }
                    nll -= (double)ll * (double)A.w_ll;
                    g[j] = -dll * A.w_ll * aim * tq * 2.0f * zf;     // dNLL / d z_f[rid][s] of this row
                }
            }
            // run totals: suffix sums inside the runs, the first lane of a run ends with the run's total
#pragma unroll
            for (int k = 0; k < 6; ++k) {
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    if (sb + j < S) {
                        const float v = __shfl_down(g[j], 1 << k);
                        g[j] += m[k] ? v : 0.0f;
                    }
                }
            }
            if (head) {
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    if (sb + j < S) {
                        const int s = sb + j;
                        if (route == 0) {
                            if (A.accumulate) atomicAdd(A.dz_f + zoff + s, g[j]);
                            else A.dz_f[zoff + s] = g[j];
                        } else {
                            A.edge_val[(size_t)(2 * wv + (route - 1)) * S + s] = g[j];
                        }
                    }
                }
            }
        }
    }
    if (gather) return;                                        // (the first pass counted the NLL and the Evans-2011 terms)
    // NLL: one fp64 atomic per workgroup; Evans-2011 terms: one set per wave
    __shared__ double sh[FB / 64];
    nll = wave_sum_d(nll);
    if (lane == 0) sh[threadIdx.x >> 6] = nll;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < FB / 64; ++k) t += sh[k];
        if (A.nll_part != nullptr) A.nll_part[blockIdx.x] = t;
        else atomicAdd(A.scalars + CL_SC_NLL, t);
    }
    if (use_ev11) {
        g0 = cl_wave_sum(g0); g1 = cl_wave_sum(g1); g2 = cl_wave_sum(g2);
        if (lane == 0) {
            const float e0 = g0 * cl_sigmoid(A.ev11[0]), e1 = g1 * cl_sigmoid(A.ev11[1]), e2 = g2 * cl_sigmoid(A.ev11[2]);
            if (A.ev11_part != nullptr) {
                float* slot = A.ev11_part + 3 * ((FB / 64) * (size_t)blockIdx.x + (threadIdx.x >> 6));
                slot[0] = e0; slot[1] = e1; slot[2] = e2;
            } else { atomicAdd(A.d_ev11 + 0, e0); atomicAdd(A.d_ev11 + 1, e1); atomicAdd(A.d_ev11 + 2, e2); }
        }
    }
}

// Harmonic groups (Laue data, reference careless/models/likelihoods/laue.py:9-47: the predictions of the rows of one group SUM before the
// likelihood), first pass -- rows in the packed order of the single-pass kernels (obs.pack_laue: a group inside a 16-row granule, member
// index and size in gmeta, padding rows refl_id < 0), iobs / sig of the group replicated on its rows.  A thread owns a row and loops over
// the samples; the group's total comes over shuffles, every member evaluates the likelihood's derivative on it, member 0 counts the
// log-likelihood; the row's amplitude gradient is STORED at gbuf[row][s].  The rows of a group belong to different reflections, so one order
// cannot serve the group sums and the per-reflection sums: the second pass (frozen_rows_kernel with `src`) gathers gbuf in reflection order.
template <int SB>
__global__ __launch_bounds__(FB) void frozen_laue_kernel(const cl_frozen_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const int lane = threadIdx.x & 63;
    const int S = A.S;
    const long long n = A.n;
    const long long chunks = (n + FB - 1) / FB;
    double nll = 0.0;
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    cl_ev11 ev = {1.0f, 0.0f, 0.0f};
    const bool use_ev11 = A.ev11 != nullptr;
    if (use_ev11) { ev.sdfac = cl_softplus(A.ev11[0]); ev.sdadd = cl_softplus(A.ev11[1]); ev.sdb = cl_softplus(A.ev11[2]); }
    const float inv_dof = (A.lik_kind == CL_LIK_STUDENTT) ? 1.0f / A.dof : 0.0f;
    for (long long c = blockIdx.x; c < chunks; c += gridDim.x) {
        const long long row = c * FB + threadIdx.x;
        const bool in = row < n;
        const long long rc = in ? row : n - 1;
        int rid = A.refl_id[rc];
        if (!in) rid = -1;
        const bool act = rid >= 0;
        const float loc = A.loc[rc], sigma = A.sigma[rc], io = A.iobs[rc], sg = A.sig[rc];
        const float aim = A.aim != nullptr ? A.aim[rc] : 1.0f;
        const long long key = A.key != nullptr ? (long long)A.key[rc] : A.obs_offset + row;
        // where this row's gradients go in gbuf: its position in REFLECTION order when the caller gives one (src: the second pass then reads
        // gbuf front to back -- a scattered 4 S-byte write per row here instead of a gathered 128-byte line per row there), else its own row
        const long long gdst = !in ? -1 : (A.src != nullptr ? (long long)A.src[rc] : row);
        const int gm = act ? A.gmeta[rc] : (1 << 8);
        const int mem = gm & 0xff, cnt = gm >> 8;
        int gmax = cnt;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) gmax = max(gmax, __shfl_xor(gmax, off));
        gmax = __builtin_amdgcn_readfirstlane(gmax);
        const float inv_sg = cl_fast_rcp(sg), log_sg = cl_fast_log(sg);
        const float* __restrict__ eta_p = A.eta != nullptr ? A.eta + (size_t)(key - A.obs_offset) * S : nullptr;
        float* __restrict__ ip_p = A.ipred_out != nullptr ? A.ipred_out + (size_t)(key - A.obs_offset) * S : nullptr;
        const size_t zoff = (size_t)(act ? rid : 0) * S;
        for (int sb = 0; sb < S; sb += SB) {
            float e[SB];
            if (eta_p == nullptr) {
                if constexpr (SB == 1) {
                    float unused;
                    cl_noise_normal_pair(A.seed, A.step, 0u, (uint64_t)key, &e[0], &unused);
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        if (sb + p < S) cl_noise_normal_pair(A.seed, A.step, (uint32_t)(sb + p), (uint64_t)key, &e[p], &e[p + 4]);
                        else { e[p] = 0.0f; e[p + 4] = 0.0f; }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < SB; ++j) e[j] = (act && sb + j < S) ? eta_p[sb + j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                if (sb + j < S) {                                  // (wave-uniform: every lane takes part in the shuffles)
                    const int s = sb + j;
                    const float zf = act ? A.z_f[zoff + s] : 0.0f;
                    const float tq = loc + sigma * e[j] + A.shift;
                    const float ipred = act ? aim * tq * zf * zf : 0.0f;
                    if (ip_p != nullptr && act) ip_p[s] = ipred;
                    float lin = 0.0f;
                    for (int mm = 0; mm < gmax; ++mm) {
                        const float v = __shfl(ipred, (lane - mem + mm) & 63);
                        lin += (mm < cnt) ? v : 0.0f;
                    }
                    float gj = 0.0f;
                    if (act) {
                        float dll, ll;
                        if (use_ev11) {
                            float gf, gb, ga;
                            ll = cl_lik_ev11(lin, io, sg, A.lik_kind, A.dof, A.lik_const, ev, &dll, &gf, &gb, &ga);
                            if (mem == 0) { g0 -= gf * A.w_ll; g1 -= ga * A.w_ll; g2 -= gb * A.w_ll; }
                        } else {
                            ll = cl_lik_log_prob3(lin, io, inv_sg, log_sg, A.lik_kind, A.dof, inv_dof, A.lik_const, &dll);
                        }
                        if (mem == 0) nll -= (double)ll * (double)A.w_ll;
                        gj = -dll * A.w_ll * aim * tq * 2.0f * zf;
                    }
                    if (gdst >= 0) A.gbuf[(size_t)gdst * S + s] = gj;
                }
            }
        }
    }
    __shared__ double sh[FB / 64];
    nll = wave_sum_d(nll);
    if (lane == 0) sh[threadIdx.x >> 6] = nll;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < FB / 64; ++k) t += sh[k];
        if (A.nll_part != nullptr) A.nll_part[blockIdx.x] = t;
        else atomicAdd(A.scalars + CL_SC_NLL, t);
    }
    if (use_ev11) {
        g0 = cl_wave_sum(g0); g1 = cl_wave_sum(g1); g2 = cl_wave_sum(g2);
        if (lane == 0) {
            const float e0 = g0 * cl_sigmoid(A.ev11[0]), e1 = g1 * cl_sigmoid(A.ev11[1]), e2 = g2 * cl_sigmoid(A.ev11[2]);
            if (A.ev11_part != nullptr) {
                float* slot = A.ev11_part + 3 * ((FB / 64) * (size_t)blockIdx.x + (threadIdx.x >> 6));
                slot[0] = e0; slot[1] = e1; slot[2] = e2;
            } else { atomicAdd(A.d_ev11 + 0, e0); atomicAdd(A.d_ev11 + 1, e1); atomicAdd(A.d_ev11 + 2, e2); }
        }
    }
}

// The runs that cross wave borders: thread = (wave whose last run starts a chain, sample); it walks the chain in wave order -- the
// partial of that run, then the first-run partials of the following waves while they continue it -- and stores the total.
__global__ __launch_bounds__(256) void frozen_edges_kernel(const cl_frozen_args A) {
    if (A.stop_flag != nullptr && *A.stop_flag != 0) return;
    const long long n_waves = (A.n + 63) / 64;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int S = A.S;
    if (t >= n_waves * S) return;
    const long long w = t / S;
    const int s = (int)(t - w * S);
    const int r = A.edge_rid[2 * w + 1];
    if (r < 0) return;
    float sum = A.edge_val[(size_t)(2 * w + 1) * S + s];
    for (long long j = w + 1; j < n_waves; ++j) {
        const int r0 = A.edge_rid[2 * j];
        if (r0 < 0 || (r0 & ~THROUGH) != r) break;
        sum += A.edge_val[(size_t)(2 * j) * S + s];
        if (!(r0 & THROUGH)) break;
    }
    if (A.accumulate) atomicAdd(A.dz_f + (size_t)r * S + s, sum);
    else A.dz_f[(size_t)r * S + s] = sum;
}

// workgroups of a launch: as many as the device holds at once (a grid-stride loop over equal shares: a second, partial round of
// workgroups would leave most of the device waiting for it), never more than cl_frozen_grid(n) (the size of nll_part / ev11_part)
template <int TAG, class K>
static int frozen_blocks(K kern, long long n) {
    const int cap = cl_frozen_grid(n);
    static std::atomic<int> resident{0};                 // (asked once per kernel: TAG tells the instances apart)
    int g = resident.load(std::memory_order_relaxed);
    if (g == 0) {
        int dev = 0, cus = 0, per = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return cap;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, FB, 0) != hipSuccess || per < 1 || cus < 1) return cap;
        g = per * cus;
        resident.store(g, std::memory_order_relaxed);
    }
    return g < cap ? g : cap;
}

int cl_frozen_edge_floats(long long n, int S) { return (n <= 0 || S <= 0) ? 0 : (int)(2 * ((n + 63) / 64) * S); }
int cl_frozen_grid(long long n) {
    long long b = (n + FB - 1) / FB;
    if (b > CL_LAUE_LIK_MAX_BLOCKS) b = CL_LAUE_LIK_MAX_BLOCKS;
    return (int)(b < 1 ? 1 : b);
}

int cl_launch_frozen_rows(const cl_frozen_args& a, hipStream_t st) {
    if (a.gmeta != nullptr) {
        // harmonic groups, first pass: per-row amplitude gradients into gbuf (no sums, no edges)
        if (a.n <= 0 || a.S <= 0 || a.refl_id == nullptr || a.loc == nullptr || a.sigma == nullptr || a.iobs == nullptr || a.sig == nullptr ||
            a.z_f == nullptr || a.gbuf == nullptr || a.scalars == nullptr)
            return -1;
        if (a.n >= (1ll << 31) - 64) return -4;
        if (a.ev11 != nullptr && a.d_ev11 == nullptr && a.ev11_part == nullptr) return -1;
        (void)hipGetLastError();
        if (a.S == 1) hipLaunchKernelGGL(frozen_laue_kernel<1>, dim3((unsigned)frozen_blocks<0>(frozen_laue_kernel<1>, a.n)), dim3(FB), 0, st, a);
        else hipLaunchKernelGGL(frozen_laue_kernel<8>, dim3((unsigned)frozen_blocks<1>(frozen_laue_kernel<8>, a.n)), dim3(FB), 0, st, a);
        return (int)hipGetLastError();
    }
    if (a.gbuf != nullptr) {
        // ... second pass: the gradients in reflection order summed per reflection (refl_id ascending; gbuf row i, or src[i] when given)
        if (a.n <= 0 || a.S <= 0 || a.refl_id == nullptr || a.gbuf == nullptr || a.dz_f == nullptr || a.edge_rid == nullptr || a.edge_val == nullptr) return -1;
    } else if (a.n <= 0 || a.S <= 0 || a.refl_id == nullptr || a.loc == nullptr || a.sigma == nullptr || a.iobs == nullptr || a.sig == nullptr ||
               a.z_f == nullptr || a.dz_f == nullptr || a.scalars == nullptr || a.edge_rid == nullptr || a.edge_val == nullptr)
        return -1;
    if (a.n >= (1ll << 31) - 64 || a.R >= THROUGH) return -4;
    if (a.ev11 != nullptr && a.d_ev11 == nullptr && a.ev11_part == nullptr) return -1;
    (void)hipGetLastError();
    if (a.S == 1) hipLaunchKernelGGL(frozen_rows_kernel<1>, dim3((unsigned)frozen_blocks<2>(frozen_rows_kernel<1>, a.n)), dim3(FB), 0, st, a);
    else hipLaunchKernelGGL(frozen_rows_kernel<8>, dim3((unsigned)frozen_blocks<3>(frozen_rows_kernel<8>, a.n)), dim3(FB), 0, st, a);
    const long long t = ((a.n + 63) / 64) * a.S;
    hipLaunchKernelGGL(frozen_edges_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
