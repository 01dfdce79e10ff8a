// CPU build of careless_amd/csrc/cl_math.h for formula checks against the oracle (TEST INFRASTRUCTURE).
// Compiled with g++ by tests/test_host_math.py; never part of the product library.
#include "../careless_amd/csrc/cl_math.h"
extern "C" {
// out: z, log_q, dz_dloc, dz_dscale, dlogq_dz, dlogq_dloc, dlogq_dscale, loc, scale, e
void hm_tn(int n, const float* a, const float* b, const float* low, float high, float eps, const float* u, float* out) {
    for (int i = 0; i < n; ++i) {
        cl_tn_elem t = cl_tn_sample(a[i], b[i], low[i], high, eps, u[i]);
        float dz, dl, ds;
        cl_tn_log_prob_grads(t, &dz, &dl, &ds);
        float* o = out + 10 * i;
        o[0] = t.z; o[1] = cl_tn_log_prob(t); o[2] = t.dz_dloc; o[3] = t.dz_dscale;
        o[4] = dz; o[5] = dl; o[6] = ds; o[7] = t.loc; o[8] = t.scale; o[9] = t.e;
    }
}
void hm_ndtri_lower(int n, const float* p, float* out) { for (int i = 0; i < n; ++i) out[i] = cl_ndtri_lower(p[i]); }
void hm_wilson(int n, const float* z, const int* centric, const float* es, float* lp, float* dlp) {
    for (int i = 0; i < n; ++i) { lp[i] = cl_wilson_log_prob(z[i], centric[i] != 0, es[i]); dlp[i] = cl_wilson_dlog_prob_dz(z[i], centric[i] != 0, es[i]); }
}
void hm_lik(int n, const float* ip, const float* io, const float* sg, int kind, float dof, float c, float* ll, float* dll) {
    for (int i = 0; i < n; ++i) ll[i] = cl_lik_log_prob(ip[i], io[i], sg[i], kind, dof, c, dll + i);
}
void hm_bij(int n, const float* raw, int kind, float eps, float* sig, float* dsig) {
    for (int i = 0; i < n; ++i) sig[i] = cl_scale_bij(raw[i], kind, eps, dsig + i);
}
void hm_dw(int n, const float* z, const float* zp, const int* has, const float* r, const int* centric, const float* es, float* lp, float* dz, float* dzp, float* dr) {
    for (int i = 0; i < n; ++i) lp[i] = cl_dw_log_prob(z[i], zp[i], has[i] != 0, r[i], centric[i] != 0, es[i], dz + i, dzp + i, dr + i);
}
void hm_bessel(int n, const float* x, float* i0e, float* i1e) { for (int i = 0; i < n; ++i) cl_i0e_i1e(x[i], i0e + i, i1e + i); }
void hm_noise(int n, unsigned long long seed, unsigned step, unsigned s, unsigned long long idx0, float* un, float* nr) {
    for (int i = 0; i < n; ++i) { un[i] = cl_noise_uniform(seed, step, s, idx0 + i); nr[i] = cl_noise_normal(seed, step, s, idx0 + i); }
}
}
