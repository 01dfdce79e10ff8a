// Host side of the formatting step (row f3): the per-observation symmetry bookkeeping of the reflection tables, natively.
//
// Reference: `DataSet.remove_absences()`, `DataSet.hkl_to_asu(anomalous=...)` and the centric / multiplicity labels as
// careless/io/formatter.py:285-302, 319 (mono), :540-562 (Laue) and careless/io/asu.py:27-56 take them from reciprocalspaceship -- which is
// gemmi's C++ underneath: the reference's own formatter is native at this point, and so is this one.  A default run of the command line on
// 5 M observations spent 6 of its 15 s here while the 2 000 training steps took 2.3 s (profiles/r5_e2e_format.txt): the numpy version
// materialised the (2 x operators, rows, 3) int64 orbit of every chunk of rows several times over.
//
// These entry points take HOST pointers and run on host threads (no stream, no device): plain integer work on 12 bytes per row, one pass,
// rows split evenly over the threads.  They are part of libcareless_hip.so so that the package has ONE native library and no second,
// interpreted implementation of the same arithmetic (tests/ref_asu.py restates it in numpy as the checker).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <sched.h>

#include "../../include/careless_hip.h"

namespace {

// The CCP4 reciprocal asymmetric units of the Laue classes in their reference settings (careless_amd/io/asu.py: _CCP4_ASU, same order):
// -1, 2/m (unique b), mmm, 4/m and 6/m, 4/mmm and 6/mmm, -3, -31m, -3m1, m-3, m-3m.
inline bool inside_asu(int c, long long h, long long k, long long l) {
    switch (c) {
        case 0: return (l > 0) || ((l == 0) && ((h > 0) || ((h == 0) && (k >= 0))));
        case 1: return (k >= 0) && ((l > 0) || ((l == 0) && (h >= 0)));
        case 2: return (h >= 0) && (k >= 0) && (l >= 0);
        case 3: return (l >= 0) && (((h >= 0) && (k > 0)) || ((h == 0) && (k == 0)));
        case 4: return (h >= k) && (k >= 0) && (l >= 0);
        case 5: return ((h >= 0) && (k > 0)) || ((h == 0) && (k == 0) && (l >= 0));
        case 6: return (h >= k) && (k >= 0) && ((k > 0) || (l >= 0));
        case 7: return (h >= k) && (k >= 0) && ((h > k) || (l >= 0));
        case 8: return (h >= 0) && (((l >= h) && (k > h)) || ((l == h) && (k == h)));
        case 9: return (k >= l) && (l >= h) && (h >= 0);
        default: return false;
    }
}

// (careless_amd/io/asu.py: _key) -- the order the "no CCP4 set fits" rule maximises
inline long long hkl_key(long long h, long long k, long long l) {
    const long long B = 1LL << 20;
    return (h + B / 2) * B * B + (k + B / 2) * B + (l + B / 2);
}

int pick_threads(long long n, int nthreads) {
    if (nthreads <= 0) {
        cpu_set_t set;
        int avail = 0;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) avail = CPU_COUNT(&set);
        if (avail <= 0) avail = (int)std::thread::hardware_concurrency();
        nthreads = avail < 1 ? 1 : (avail > 32 ? 32 : avail);
    }
    const long long by_rows = n / 32768 + 1;          // a thread is worth starting for >= 32 k rows
    if (nthreads > by_rows) nthreads = (int)by_rows;
    return nthreads < 1 ? 1 : nthreads;
}

template <class F>
void parallel_rows(long long n, int nthreads, F&& body) {
    nthreads = pick_threads(n, nthreads);
    if (nthreads == 1) { body(0LL, n); return; }
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    const long long per = (n + nthreads - 1) / nthreads;
    long long done = 0;                       // rows handed to a thread so far
    for (int t = 0; t < nthreads; ++t) {
        const long long a = t * per, b = a + per < n ? a + per : n;
        if (a >= b) break;
        try {
            pool.emplace_back([&body, a, b] { body(a, b); });
            done = b;
        } catch (...) {                       // (no more threads to be had: the caller's thread takes the rest -- nothing may unwind through the C-ABI)
            break;
        }
    }
    if (done < n) body(done, n);
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

int cl_host_asu_map(const int32_t* hkl, long long n, const int32_t* rot, const double* trans, int nops, int asu_case, int anomalous,
                    int32_t* hasu, uint8_t* centric, int32_t* eps, uint8_t* absent, int nthreads) {
    if (n < 0 || nops < 1 || nops > 192 || rot == nullptr || (n > 0 && hkl == nullptr)) return -1;
    if (asu_case < -1 || asu_case > 9) return -1;
    if (absent != nullptr && trans == nullptr) return -1;
    parallel_rows(n, nthreads, [=](long long a, long long b) {
        for (long long i = a; i < b; ++i) {
            const long long h0 = hkl[3 * i], h1 = hkl[3 * i + 1], h2 = hkl[3 * i + 2];
            int n_same = 0;
            bool is_absent = false, is_centric = false;
            // representative: the first orbit member inside the CCP4 set, rotations first, then their negatives (np.argmax of the mask: index
            // 0 when none is inside); without a set the member with the largest key (first of equals)
            bool found = false, rep_is_rot = false;
            long long r0 = 0, r1 = 0, r2 = 0, best_key = 0;
            for (int pass = 0; pass < 2 && (hasu != nullptr || pass == 0); ++pass) {
                for (int o = 0; o < nops; ++o) {
                    const int32_t* R = rot + 9 * o;
                    long long g0 = h0 * R[0] + h1 * R[3] + h2 * R[6];
                    long long g1 = h0 * R[1] + h1 * R[4] + h2 * R[7];
                    long long g2 = h0 * R[2] + h1 * R[5] + h2 * R[8];
                    if (pass == 0) {
                        if (g0 == h0 && g1 == h1 && g2 == h2) {
                            ++n_same;
                            if (absent != nullptr) {
                                const double ph = (double)h0 * trans[3 * o] + (double)h1 * trans[3 * o + 1] + (double)h2 * trans[3 * o + 2];
                                if (std::fabs(ph - std::nearbyint(ph)) > 1e-6) is_absent = true;
                            }
                        }
                        if (g0 == -h0 && g1 == -h1 && g2 == -h2) is_centric = true;
                    } else { g0 = -g0; g1 = -g1; g2 = -g2; }
                    if (hasu == nullptr) continue;
                    const bool first = (pass == 0 && o == 0);
                    if (asu_case >= 0) {
                        if (first) { r0 = g0; r1 = g1; r2 = g2; }
                        if (!found && inside_asu(asu_case, g0, g1, g2)) { found = true; r0 = g0; r1 = g1; r2 = g2; }
                    } else {
                        const long long key = hkl_key(g0, g1, g2);
                        if (first || key > best_key) { best_key = key; r0 = g0; r1 = g1; r2 = g2; }
                    }
                }
            }
            if (hasu != nullptr) {
                if (anomalous) {
                    // Friedel-minus: the representative is not among the rotation images -> keep its negative
                    for (int o = 0; o < nops && !rep_is_rot; ++o) {
                        const int32_t* R = rot + 9 * o;
                        rep_is_rot = (h0 * R[0] + h1 * R[3] + h2 * R[6] == r0) && (h0 * R[1] + h1 * R[4] + h2 * R[7] == r1) &&
                                     (h0 * R[2] + h1 * R[5] + h2 * R[8] == r2);
                    }
                    if (!rep_is_rot) { r0 = -r0; r1 = -r1; r2 = -r2; }
                }
                hasu[3 * i] = (int32_t)r0; hasu[3 * i + 1] = (int32_t)r1; hasu[3 * i + 2] = (int32_t)r2;
            }
            if (centric != nullptr) centric[i] = is_centric ? 1 : 0;
            if (eps != nullptr) eps[i] = n_same;
            if (absent != nullptr) absent[i] = is_absent ? 1 : 0;
        }
    });
    return 0;
}

int cl_host_dense_ids(const int64_t* key, long long n, int64_t key_min, int64_t key_max, int64_t* ids, long long* n_groups, int nthreads) {
    if (n < 0 || (n > 0 && (key == nullptr || ids == nullptr)) || key_max < key_min) return -1;
    const unsigned long long range = (unsigned long long)(key_max - key_min) + 1ULL;
    if (range > (1ULL << 31)) return -2;                // the caller sorts instead
    std::vector<int32_t> slot;                          // (ranks fit 31 bits: at most `range` distinct keys)
    try { slot.assign((size_t)range, 0); } catch (...) { return -3; }      // no memory for the table: the caller sorts instead
    bool bad = false;
    for (long long i = 0; i < n; ++i) {                // presence (serial: 8 bytes per row, and the table is shared)
        const int64_t k = key[i];
        if (k < key_min || k > key_max) { bad = true; break; }
        slot[(size_t)(k - key_min)] = 1;
    }
    if (bad) return -1;
    int64_t next = 0;
    for (size_t s = 0; s < (size_t)range; ++s) { const int32_t present = slot[s]; slot[s] = (int32_t)next; next += present; }
    const int32_t* sl = slot.data();
    parallel_rows(n, nthreads, [=](long long a, long long b) {
        for (long long i = a; i < b; ++i) ids[i] = sl[(size_t)(key[i] - key_min)];
    });
    if (n_groups != nullptr) *n_groups = (long long)next;
    return 0;
}

}  // extern "C"
