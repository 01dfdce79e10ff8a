// Scalar fp32 math of the ELBO path: truncated-normal surrogate posterior, Wilson prior, Normal / Student-T
// likelihood, scale bijectors and the counter-based RNG.  Every kernel in this directory gets its per-element
// arithmetic from here, so the formulas are written (and unit-checked) once.
//
// The functions are plain inline C++ with a CL_HD qualifier so the same text also compiles with g++ for the
// CPU-side formula check in tests/ (tests/host_math_check.cpp); the product only ever runs them on the GPU.
//
// Reference semantics (paths relative to the reference checkout; [3P] = recalled TFP/Keras behaviour):
//   careless/models/merging/surrogate_posteriors.py:45-131   truncated normal q(F)
//   careless/models/priors/wilson.py:13-57                   Wilson prior
//   careless/models/likelihoods/mono.py:10-37                Normal / Student-T likelihood
//   careless/models/scaling/nn.py:10-25                      NormalLayer scale bijector
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define CL_HD __host__ __device__ __forceinline__
#else
#define CL_HD inline
#endif

// Placed inside the EXPENSIVE arm of a branch on a wave-uniform kind / flag: an (empty) side effect hipcc cannot speculate, so the
// arm stays behind a real scalar branch.  Without it hipcc if-converts e.g. `kind == exp ? exp(x) : softplus(x)` into straight-line
// code that evaluates BOTH arms in every lane (the accurate log1p alone is ~80 vector instructions per call).
#if defined(__HIP_DEVICE_COMPILE__)
#define CL_KEEP_BRANCH() asm volatile("")
#else
#define CL_KEEP_BRANCH()
#endif

#define CL_LOG_2PI_F 1.8378770664093453f
#define CL_INV_SQRT2_F 0.70710678118654752f
#define CL_INV_SQRT_2PI_F 0.3989422804014327f
#define CL_TINY_F 1.17549435e-38f          // np.finfo(float32).tiny  [3P: clip in TFP's sample gradient]
#define CL_EPS_F 1.1920929e-07f            // np.finfo(float32).eps

enum { CL_LIK_NORMAL = 0, CL_LIK_STUDENTT = 1 };
enum { CL_BIJ_EXP = 0, CL_BIJ_SOFTPLUS = 1 };
enum { CL_PRIOR_WILSON = 0, CL_PRIOR_DOUBLE_WILSON = 1 };

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. 2011).  counter = (index lo, index hi | stream, sample, step)
// so every draw is a pure function of (seed, step, sample, global element index): results do not depend on the
// number of GPUs or on the launch geometry.
// ---------------------------------------------------------------------------------------------------------
struct cl_u32x4 { uint32_t x, y, z, w; };

CL_HD uint32_t cl_mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

// `rounds` = 10 is the standard generator; 7 is the smallest count that passes BigCrush (Salmon et al., table 2) and is what the
// in-kernel scale-noise draw uses (it is on the critical path of the epilogue)
template <int ROUNDS>
CL_HD cl_u32x4 cl_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;      // one 32x32->64 multiply each (v_mad_u64_u32)
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    cl_u32x4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}

// uniform in the open interval (0,1): 23 random bits, centred in their bin (k + 0.5 is exact in fp32 for k < 2^23, so the
// result never rounds to 0 or 1)
CL_HD float cl_u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f; }

enum { CL_STREAM_QF = 1, CL_STREAM_SCALE = 2 };

// Uniforms for the truncated-normal draws of reflection idx: one Philox block serves the four MC samples 4k .. 4k+3
// (component s & 3 of the block keyed by s >> 2), so a thread that walks s = 0, 1, 2, ... generates a block every 4th sample.
CL_HD cl_u32x4 cl_noise_uniform_block(uint64_t seed, uint32_t step, uint32_t s_block, uint64_t idx) {
    return cl_philox4x32<10>((uint32_t)idx, (uint32_t)(idx >> 32) | ((uint32_t)CL_STREAM_QF << 28), s_block, step,
                             (uint32_t)seed, (uint32_t)(seed >> 32));
}
CL_HD float cl_noise_uniform_pick(const cl_u32x4& r, uint32_t s) {
    const uint32_t c = s & 3u;
    return cl_u01(c == 0 ? r.x : c == 1 ? r.y : c == 2 ? r.z : r.w);
}
CL_HD float cl_noise_uniform(uint64_t seed, uint32_t step, uint32_t s, uint64_t idx) {
    return cl_noise_uniform_pick(cl_noise_uniform_block(seed, step, s >> 2, idx), s);
}

// fast device forms of the transcendental pieces of the noise generator (1-ulp hardware approximations are ample for
// Monte-Carlo noise; the oracle is always fed the numbers the device actually drew, via cl_debug_noise)
#if defined(__HIP_DEVICE_COMPILE__)
// (v_log_f32 is log2 to 1 ulp; `__logf` wraps it in ~10 instructions of denormal scaling and error compensation that the arguments here --
// uniforms >= 2^-24, 1 + x, sigmas -- do not need: every instruction of the sampling epilogue is paid in full by a lone wave)
CL_HD float cl_fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
CL_HD float cl_fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
CL_HD void cl_fast_sincos_rev(float rev, float* s, float* c) { *s = __builtin_amdgcn_sinf(rev); *c = __builtin_amdgcn_cosf(rev); }
CL_HD float cl_fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
CL_HD float cl_fast_log(float x) { return logf(x); }
CL_HD float cl_fast_sqrt(float x) { return sqrtf(x); }
CL_HD void cl_fast_sincos_rev(float rev, float* s, float* c) { *s = sinf(6.283185307179586f * rev); *c = cosf(6.283185307179586f * rev); }
CL_HD float cl_fast_rcp(float x) { return 1.0f / x; }
#endif

// Standard normals for the scale draw of observation idx.  One Philox block + one Box-Muller transform serve TWO
// MC samples: samples s and s+4 (same s & 3, consecutive s >> 2) are the cosine and the sine branch of one pair, because
// a lane of the fused kernel handles exactly the samples s = q, q+4, q+8, ... of its observation.
CL_HD void cl_noise_normal_pair(uint64_t seed, uint32_t step, uint32_t s_even, uint64_t idx, float* n_cos, float* n_sin) {
    // s_even = the sample index of the cosine branch: (s >> 2) even
    const uint32_t key = ((s_even >> 3) << 2) | (s_even & 3u);
    const cl_u32x4 r = cl_philox4x32<7>((uint32_t)idx, (uint32_t)(idx >> 32) | ((uint32_t)CL_STREAM_SCALE << 28), key, step,
                                        (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u1 = cl_u01(r.x), u2 = cl_u01(r.y);
    const float rad = cl_fast_sqrt(-2.0f * cl_fast_log(u1));
    float sn, cs;
    cl_fast_sincos_rev(u2, &sn, &cs);
    *n_cos = rad * cs;
    *n_sin = rad * sn;
}

// standard normal for the scale draw of observation idx, MC sample s
CL_HD float cl_noise_normal(uint64_t seed, uint32_t step, uint32_t s, uint64_t idx) {
    float a, b;
    const uint32_t kk = s >> 2;
    cl_noise_normal_pair(seed, step, ((kk & ~1u) << 2) | (s & 3u), idx, &a, &b);
    return (kk & 1u) ? b : a;
}

// ---------------------------------------------------------------------------------------------------------
// normal cdf pieces
// ---------------------------------------------------------------------------------------------------------
CL_HD float cl_ndtr(float x) { return 0.5f * erfcf(-x * CL_INV_SQRT2_F); }
CL_HD float cl_npdf(float x) { return CL_INV_SQRT_2PI_F * expf(-0.5f * x * x); }

// Inverse normal cdf for p in (0, 0.5]: Wichura's AS241 PPND7 (about 7 significant digits), lower half only.
CL_HD float cl_ndtri_lower(float p) {
    const float q = p - 0.5f;
    if (q >= -0.425f) {
        const float r = 0.180625f - q * q;
        const float num = ((5.9109374720e+01f * r + 1.5929113202e+02f) * r + 5.0434271938e+01f) * r + 3.3871327179e+00f;
        const float den = ((6.7187563600e+01f * r + 7.8757757664e+01f) * r + 1.7895169469e+01f) * r + 1.0f;
        return q * num / den;
    }
    float r = sqrtf(-logf(p));
    float v;
    if (r <= 5.0f) {
        r -= 1.6f;
        v = (((1.7023821103e-01f * r + 1.3067284816e+00f) * r + 2.7568153900e+00f) * r + 1.4234372777e+00f) /
            ((1.2021132975e-01f * r + 7.3700164250e-01f) * r + 1.0f);
    } else {
        r -= 5.0f;
        v = (((1.7337203997e-02f * r + 4.2868294337e-01f) * r + 3.0812263860e+00f) * r + 6.6579051150e+00f) /
            ((1.2258202635e-02f * r + 2.4197894225e-01f) * r + 1.0f);
    }
    return -v;
}

// ---------------------------------------------------------------------------------------------------------
// truncated-normal surrogate posterior, one (reflection, sample) element
// ---------------------------------------------------------------------------------------------------------
struct cl_tn_elem {
    float loc, scale;        // exp(a), exp(b) + eps          (surrogate_posteriors.py:120-130)
    float alpha, beta;       // standardised bounds
    float zn;                // Phi(beta) - Phi(alpha)
    float e;                 // standardised sample
    float y;                 // (z - loc) / scale, taken as e (or alpha when clamped) to avoid the cancellation in z - loc
    float z;                 // max(low, loc + scale e)        (surrogate_posteriors.py:50-53)
    float dz_dloc, dz_dscale;  // pathwise sample gradients [3P: TFP _std_samples_with_gradients]
};

CL_HD cl_tn_elem cl_tn_sample(float a_raw, float b_raw, float low, float high, float eps, float u) {
    cl_tn_elem r;
    r.loc = expf(a_raw);
    r.scale = expf(b_raw) + eps;
    const float inv = 1.0f / r.scale;
    r.alpha = (low - r.loc) * inv;
    r.beta = (high - r.loc) * inv;
    const float up_b = 0.5f * erfcf(r.beta * CL_INV_SQRT2_F);      // Phi(-beta)
    const float up_a = 0.5f * erfcf(r.alpha * CL_INV_SQRT2_F);     // Phi(-alpha)
    r.zn = up_a - up_b;
    const float lo_a = 0.5f * erfcf(-r.alpha * CL_INV_SQRT2_F);    // Phi(alpha)
    const float p = lo_a + u * r.zn;
    const float q = up_b + (1.0f - u) * r.zn;
    float e = (p < 0.5f) ? cl_ndtri_lower(p) : -cl_ndtri_lower(q);
    e = fminf(fmaxf(e, r.alpha), r.beta);
    r.e = e;
    const float s = r.loc + r.scale * e;
    const bool pass = s > low;
    r.z = pass ? s : low;
    r.y = pass ? e : r.alpha;
    float cdf = fminf(fmaxf(u, CL_TINY_F), 1.0f - CL_EPS_F);
    const float dl = expf(0.5f * (e * e - r.alpha * r.alpha) + log1pf(-cdf));
    const float du = expf(0.5f * (e * e - r.beta * r.beta) + logf(cdf));
    r.dz_dloc = pass ? (1.0f - dl - du) : 0.0f;
    r.dz_dscale = pass ? (e - r.alpha * dl - (du > 0.0f ? r.beta * du : 0.0f)) : 0.0f;
    return r;
}

// log q(z) = -(0.5 y^2 + 0.5 log 2pi + log scale + log Z)   [3P: tfd.TruncatedNormal.log_prob]
CL_HD float cl_tn_log_prob(const cl_tn_elem& t) {
    const float y = t.y;
    return -(0.5f * y * y + 0.5f * CL_LOG_2PI_F + logf(t.scale) + logf(t.zn));
}

// partial derivatives of log q(z; loc, scale) at fixed z, and d/dz
CL_HD void cl_tn_log_prob_grads(const cl_tn_elem& t, float* dz, float* dloc, float* dscale) {
    const float inv = 1.0f / t.scale;
    const float y = t.y;
    const float pa = cl_npdf(t.alpha);
    const float pb = cl_npdf(t.beta);
    const float bpb = (pb > 0.0f) ? t.beta * pb : 0.0f;
    const float izn = 1.0f / t.zn;
    *dz = -y * inv;
    *dloc = y * inv - (pa - pb) * inv * izn;
    *dscale = y * y * inv - inv - (t.alpha * pa - bpb) * inv * izn;
}

// ---------------------------------------------------------------------------------------------------------
// Wilson prior (wilson.py:13-57): centric HalfNormal(sqrt(eps Sigma)), acentric Weibull(2, sqrt(eps Sigma))
// es = multiplicity * Sigma
// ---------------------------------------------------------------------------------------------------------
CL_HD float cl_wilson_log_prob(float z, bool centric, float es) {
    if (centric) return -0.5f * z * z / es + 0.5f * logf(0.6366197723675814f) - 0.5f * logf(es);
    return 0.6931471805599453f + logf(z) - logf(es) - z * z / es;
}
CL_HD float cl_wilson_dlog_prob_dz(float z, bool centric, float es) {
    if (centric) return -z / es;
    return 1.0f / z - 2.0f * z / es;
}

// ---------------------------------------------------------------------------------------------------------
// scale bijector of the scaler's NormalLayer (nn.py:22-25; manager.py:450-463): sigma = f(raw) + eps
// ---------------------------------------------------------------------------------------------------------
CL_HD float cl_softplus(float x) { return (x > 20.0f) ? x : log1pf(expf(x)); }
CL_HD float cl_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
CL_HD float cl_scale_bij(float raw, int kind, float eps, float* dsig_draw) {
    if (kind == CL_BIJ_EXP) {
        const float ex = expf(raw);
        *dsig_draw = ex;
        return ex + eps;
    }
    CL_KEEP_BRANCH();
    *dsig_draw = cl_sigmoid(raw);
    return cl_softplus(raw) + eps;
}

// ---------------------------------------------------------------------------------------------------------
// likelihood of one prediction (mono.py:10-37).  Returns log p(ipred) and writes d log p / d ipred.
// lik_const for Student-T = lgamma((nu+1)/2) - lgamma(nu/2) - 0.5 log(nu pi)   (computed on the host)
// ---------------------------------------------------------------------------------------------------------
CL_HD float cl_lik_log_prob(float ipred, float iobs, float sig, int kind, float dof, float lik_const, float* dll) {
    const float inv = 1.0f / sig;
    const float y = (ipred - iobs) * inv;
    if (kind == CL_LIK_NORMAL) {
        *dll = -y * inv;
        return -0.5f * y * y - 0.5f * CL_LOG_2PI_F - logf(sig);
    }
    const float y2 = y * y;
    *dll = -(dof + 1.0f) * y / (dof + y2) * inv;
    return -0.5f * (dof + 1.0f) * log1pf(y2 / dof) - logf(sig) + lik_const;
}

// log(1 + x) for x >= 0 from the hardware log: log(u) + (x - (u - 1)) / u with u = fl(1 + x) (the correction term restores
// the bits of x lost in forming u), relative error ~1e-7
CL_HD float cl_log1p_pos(float x) {
    const float u = 1.0f + x;
    return cl_fast_log(u) + (x - (u - 1.0f)) * cl_fast_rcp(u);
}

// same, with 1/sig and log(sig) hoisted by the caller (they are per-observation, the prediction is per MC sample)
CL_HD float cl_lik_log_prob2(float ipred, float iobs, float inv_sig, float log_sig, int kind, float dof, float lik_const,
                             float* dll) {
    const float y = (ipred - iobs) * inv_sig;
    if (kind == CL_LIK_NORMAL) {
        *dll = -y * inv_sig;
        return -0.5f * y * y - 0.5f * CL_LOG_2PI_F - log_sig;
    }
    CL_KEEP_BRANCH();
    const float y2 = y * y;
    *dll = -(dof + 1.0f) * y / (dof + y2) * inv_sig;
    return -0.5f * (dof + 1.0f) * cl_log1p_pos(y2 / dof) - log_sig + lik_const;
}

// same again with 1/nu hoisted and the one per-sample division of the Student-T derivative as a hardware reciprocal refined by one
// Newton step (relative error ~1e-7): two accurate divisions are ~20 vector instructions per MC sample in the epilogue
CL_HD float cl_lik_log_prob3(float ipred, float iobs, float inv_sig, float log_sig, int kind, float dof, float inv_dof,
                             float lik_const, float* dll) {
    const float y = (ipred - iobs) * inv_sig;
    if (kind == CL_LIK_NORMAL) {
        *dll = -y * inv_sig;
        return -0.5f * y * y - 0.5f * CL_LOG_2PI_F - log_sig;
    }
    CL_KEEP_BRANCH();
    const float y2 = y * y;
    const float den = dof + y2;
    float r = cl_fast_rcp(den);
    r = r * (2.0f - den * r);
    *dll = -(dof + 1.0f) * y * r * inv_sig;
    return -0.5f * (dof + 1.0f) * cl_log1p_pos(y2 * inv_dof) - log_sig + lik_const;
}

// ---------------------------------------------------------------------------------------------------------
// Double-Wilson prior pieces (careless/models/priors/wilson.py:146-175; careless/utils/distributions.py:228-348):
// exponentially scaled Bessel functions by Chebyshev series on the cephes intervals (|x| <= 8, |x| > 8); coefficients are a
// numerical Chebyshev interpolation of scipy.special.i0e / i1e (relative error < 2e-8 before fp32 rounding)
// ---------------------------------------------------------------------------------------------------------
CL_HD float cl_cheb(float t, const float* c, int n) {      // Clenshaw: c0 + c1 T1(t) + ...
    float b1 = 0.0f, b2 = 0.0f;
    for (int k = n - 1; k >= 1; --k) {
        const float b0 = 2.0f * t * b1 - b2 + c[k];
        b2 = b1; b1 = b0;
    }
    return t * b1 - b2 + c[0];
}
CL_HD void cl_i0e_i1e(float x, float* i0e, float* i1e) {
    const float A0[18] = {3.383976372e-01f, -3.046826723e-01f, 1.716209015e-01f, -9.490109705e-02f, 4.930528424e-02f, -2.373741481e-02f,
                          1.054646039e-02f, -4.324309995e-03f, 1.639475617e-03f, -5.763755745e-04f, 1.885028851e-04f, -5.754195009e-05f,
                          1.644844799e-05f, -4.416737873e-06f, 1.117384585e-06f, -2.670621174e-07f, 6.037319241e-08f, -1.248127081e-08f};
    const float B0[8] = {4.022452055e-01f, 3.369116478e-03f, 6.889758345e-05f, 2.891370136e-06f, 2.048911412e-07f, 2.266848402e-08f,
                         3.409447656e-09f, 5.255593183e-10f};
    const float A1[18] = {1.262935932e-01f, -1.764165184e-01f, 1.026436587e-01f, -5.294598121e-02f, 2.472644903e-02f, -1.056408489e-02f,
                          4.156422944e-03f, -1.513572451e-03f, 5.122859562e-04f, -1.617608158e-04f, 4.781565108e-05f, -1.327316366e-05f,
                          3.470251301e-06f, -8.568719762e-07f, 2.003291532e-07f, -4.444860747e-08f, 9.369801317e-09f, -1.820614997e-09f};
    const float B1[8] = {3.892881175e-01f, -9.761097491e-03f, -1.105889387e-04f, -3.882564401e-06f, -2.512229042e-07f, -2.631672406e-08f,
                         -3.849506242e-09f, -5.915003765e-10f};
    const float ax = fabsf(x);
    if (ax <= 8.0f) {
        const float t = 0.5f * (0.5f * ax - 2.0f);
        *i0e = cl_cheb(t, A0, 18);
        *i1e = x * cl_cheb(t, A1, 18);
    } else {
        const float t = 0.5f * (32.0f / ax - 2.0f);
        const float rs = 1.0f / sqrtf(ax);
        *i0e = cl_cheb(t, B0, 8) * rs;
        const float v = cl_cheb(t, B1, 8) * rs;
        *i1e = (x < 0.0f) ? -v : v;
    }
}

// Rice(nu, sigma) log-density at x > 0 (distributions.py:278-283) and its derivatives w.r.t. x, nu and sigma
CL_HD float cl_rice_log_prob(float x, float nu, float sigma, float* dx, float* dnu, float* dsigma) {
    const float is2 = 1.0f / (sigma * sigma);
    const float arg = x * nu * is2;
    float i0e, i1e;
    cl_i0e_i1e(arg, &i0e, &i1e);
    const float ratio = i1e / i0e;                       // I1/I0
    *dx = 1.0f / x - x * is2 + nu * is2 * ratio;
    *dnu = -nu * is2 + x * is2 * ratio;
    *dsigma = (-2.0f + (x * x + nu * nu) * is2 - 2.0f * arg * ratio) / sigma;
    return logf(x) - 2.0f * logf(sigma) - 0.5f * (x * x + nu * nu) * is2 + logf(i0e) + fabsf(arg);
}

// FoldedNormal(loc, scale) log-density at x >= 0 (distributions.py:333-335): log[N(x; loc, scale) + N(-x; loc, scale)]
CL_HD float cl_folded_normal_log_prob(float x, float loc, float scale, float* dx, float* dloc, float* dscale) {
    const float inv = 1.0f / scale;
    const float ya = (x - loc) * inv, yb = (-x - loc) * inv;
    const float la = -0.5f * ya * ya, lb = -0.5f * yb * yb;
    const float m = fmaxf(la, lb);
    const float ea = expf(la - m), eb = expf(lb - m);
    const float den = ea + eb;
    const float wa = ea / den, wb = eb / den;
    // d la/dx = -ya/scale ; d lb/dx = +yb/scale ; d la/dloc = ya/scale ; d lb/dloc = yb/scale ; d l*/dscale = y*^2/scale
    *dx = (-wa * ya + wb * yb) * inv;
    *dloc = (wa * ya + wb * yb) * inv;
    *dscale = (wa * ya * ya + wb * yb * yb - 1.0f) * inv;
    return m + logf(den) - 0.5f * CL_LOG_2PI_F - logf(scale);
}

// conditional prior of a non-root reflection of the double-Wilson model (wilson.py:146-175); dr = d log p / d r
CL_HD float cl_dw_log_prob(float z, float z_parent, bool has_parent, float r, bool centric, float es, float* dz, float* dzp,
                           float* dr) {
    const float loc = has_parent ? z_parent * r : 0.0f;
    const float c = centric ? es : 0.5f * es;
    const float scale = sqrtf(c * (1.0f - r * r));
    float dloc, dscale, lp;
    if (centric) lp = cl_folded_normal_log_prob(z, loc, scale, dz, &dloc, &dscale);
    else lp = cl_rice_log_prob(z, loc, scale, dz, &dloc, &dscale);
    *dzp = has_parent ? dloc * r : 0.0f;
    *dr = (has_parent ? dloc * z_parent : 0.0f) - dscale * c * r / scale;
    return lp;
}

// ---------------------------------------------------------------------------------------------------------
// Evans-2011 error model (`Ev11Likelihood`, careless/models/likelihoods/mono.py:39-73): the scale of the likelihood is
//   sig_c = Sdfac * sqrt(SigIobs^2 + SdB * softplus(ipred) + Sdadd * softplus(ipred)^2),   Sd* = softplus(raw) trainable
// Returns log p(ipred) and d/d ipred (through the location AND the scale) plus d/d(Sdfac, SdB, Sdadd).
// ---------------------------------------------------------------------------------------------------------
struct cl_ev11 { float sdfac, sdb, sdadd; };
CL_HD float cl_lik_ev11(float ipred, float iobs, float sig, int kind, float dof, float lik_const, cl_ev11 p, float* dll,
                        float* g_fac, float* g_b, float* g_add) {
    const float sp = cl_softplus(ipred);
    const float v = sig * sig + p.sdb * sp + p.sdadd * sp * sp;
    const float rv = sqrtf(v);
    const float sc = p.sdfac * rv;
    const float inv = 1.0f / sc;
    const float y = (ipred - iobs) * inv;
    const float y2 = y * y;
    float ll, dx, dsc;
    if (kind == CL_LIK_NORMAL) {
        ll = -0.5f * y2 - 0.5f * CL_LOG_2PI_F - logf(sc);
        dx = -y * inv;
        dsc = (y2 - 1.0f) * inv;
    } else {
        const float t = (dof + 1.0f) / (dof + y2);
        ll = -0.5f * (dof + 1.0f) * log1pf(y2 / dof) - logf(sc) + lik_const;
        dx = -t * y * inv;
        dsc = (t * y2 - 1.0f) * inv;
    }
    const float hrv = 0.5f / rv;
    *dll = dx + dsc * p.sdfac * (p.sdb + 2.0f * p.sdadd * sp) * hrv * cl_sigmoid(ipred);
    *g_fac = dsc * rv;
    *g_b = dsc * p.sdfac * sp * hrv;
    *g_add = dsc * p.sdfac * sp * sp * hrv;
    return ll;
}
