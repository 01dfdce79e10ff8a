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
#include <cstdlib>
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

// `grain`: units of work a thread is worth starting for (rows of a table: 32 k; reflection lists of a stream: a few)
int pick_threads(long long n, int nthreads, long long grain = 32768) {
    if (nthreads <= 0) {
        cpu_set_t set;
        int avail = 0;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) avail = CPU_COUNT(&set);
        if (avail <= 0) avail = (int)std::thread::hardware_concurrency();
        nthreads = avail < 1 ? 1 : (avail > 32 ? 32 : avail);
    }
    const long long by_rows = n / grain + 1;
    if (nthreads > by_rows) nthreads = (int)by_rows;
    return nthreads < 1 ? 1 : nthreads;
}

template <class F>
void parallel_rows(long long n, int nthreads, F&& body, long long grain = 32768) {
    nthreads = pick_threads(n, nthreads, grain);
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

// ---- CrystFEL stream files -------------------------------------------------------------------------------------------------------------
// The indexed reflection lists of a `.stream` file (text; serial-crystallography runs write 10^7 .. 10^8 lines of them) as one unmerged
// table.  Reference: `rs.read_crystfel` behind careless/io/formatter.py:179-184.  Line rules as the package's former Python loop
// (tests/ref_crystfel.py): a line "--- Begin crystal" starts the next crystal (BATCH), "Reflections measured after indexing" opens a list and
// is followed by one column header line, "End of reflections" closes it; a list line with at least nine blank-separated fields is a row
// (h k l I sigma(I) peak background fs/px ss/px), anything shorter is skipped.
namespace {

struct RefBlock { const char* b; const char* e; long long batch; long long rows; long long first; };

inline bool starts(const char* p, const char* end, const char* lit, size_t n) { return (size_t)(end - p) >= n && std::memcmp(p, lit, n) == 0; }
inline const char* line_end(const char* p, const char* end) { const void* q = std::memchr(p, '\n', (size_t)(end - p)); return q ? (const char*)q : end; }

// Python's str.split() separators that can occur inside a line of such a file
inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

inline int count_fields(const char* p, const char* e) {
    int n = 0;
    while (p < e) {
        while (p < e && is_blank(*p)) ++p;
        if (p >= e) break;
        ++n;
        while (p < e && !is_blank(*p)) ++p;
    }
    return n;
}

// [+-]digits[.digits] with at most 15 significant digits and at most 22 decimals: mantissa and power of ten are exact doubles, their
// quotient is the correctly rounded value (Clinger's fast path) -- what strtod returns, without its cost.  Anything else: false (strtod).
inline bool fast_decimal(const char* t, size_t len, double* out) {
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    size_t i = 0;
    bool neg = false;
    if (i < len && (t[i] == '-' || t[i] == '+')) { neg = t[i] == '-'; ++i; }
    unsigned long long m = 0;
    int digits = 0, decimals = 0;
    bool any = false, dot = false;
    for (; i < len; ++i) {
        const char ch = t[i];
        if (ch >= '0' && ch <= '9') {
            any = true;
            if (m != 0 || ch != '0') ++digits;
            if (digits > 15) return false;
            m = m * 10ULL + (unsigned long long)(ch - '0');
            if (dot) ++decimals;
        } else if (ch == '.' && !dot) dot = true;
        else return false;
    }
    if (!any || decimals > 22) return false;
    const double v = (double)m / P10[decimals];
    *out = neg ? -v : v;
    return true;
}

void find_blocks(const char* buf, long long nbytes, std::vector<RefBlock>& blocks) {
    const char* const end = buf + nbytes;
    static const char kCell0[] = "----- Begin unit cell", kCell1[] = "----- End unit cell", kCrystal[] = "--- Begin crystal",
                      kRefl[] = "Reflections measured after indexing", kEnd[] = "End of reflections";
    long long batch = -1;
    bool in_cell = false, in_refl = false;
    const char* open = nullptr;                 // first byte of the open list's rows of the current crystal
    auto close = [&](const char* at) {
        if (open != nullptr && at > open) blocks.push_back(RefBlock{open, at, batch, 0, 0});
        open = nullptr;
    };
    for (const char* p = buf; p < end;) {
        const char* e = line_end(p, end);
        const char* next = e < end ? e + 1 : end;
        if (starts(p, e, kCell0, sizeof(kCell0) - 1)) { close(p); in_cell = true; if (in_refl) open = next; }
        else if (starts(p, e, kCell1, sizeof(kCell1) - 1)) { close(p); in_cell = false; if (in_refl) open = next; }
        else if (in_cell) { close(p); if (in_refl) open = next; }                  // (lines of a unit-cell block are never rows)
        else if (starts(p, e, kCrystal, sizeof(kCrystal) - 1)) { close(p); ++batch; if (in_refl) open = next; }
        else if (starts(p, e, kRefl, sizeof(kRefl) - 1)) {
            close(p);
            in_refl = true;
            const char* h = next < end ? line_end(next, end) : end;                 // the column header line
            next = h < end ? h + 1 : end;
            open = next;
        } else if (starts(p, e, kEnd, sizeof(kEnd) - 1)) { close(p); in_refl = false; }
        p = next;
    }
    if (in_refl) close(end);
}

}  // namespace

long long cl_host_crystfel_count(const char* buf, long long nbytes, long long* n_crystals, int nthreads) {
    if (buf == nullptr || nbytes < 0) return -1;
    std::vector<RefBlock> blocks;
    try { find_blocks(buf, nbytes, blocks); } catch (...) { return -3; }
    RefBlock* B = blocks.data();
    parallel_rows((long long)blocks.size(), nthreads > 0 ? nthreads : 0, [=](long long a, long long b) {
        for (long long k = a; k < b; ++k) {
            long long rows = 0;
            for (const char* p = B[k].b; p < B[k].e;) {
                const char* e = line_end(p, B[k].e);
                if (count_fields(p, e) >= 9) ++rows;
                p = e < B[k].e ? e + 1 : B[k].e;
            }
            B[k].rows = rows;
        }
    }, 4);
    long long total = 0, last = -1;
    for (auto& bl : blocks) { total += bl.rows; if (bl.batch > last) last = bl.batch; }
    if (n_crystals != nullptr) {
        // crystals = "--- Begin crystal" lines in the whole file (also those without a reflection list)
        long long c = 0;
        const char* const end = buf + nbytes;
        for (const char* p = buf; p < end;) {
            const char* e = line_end(p, end);
            if (starts(p, e, "--- Begin crystal", 17)) ++c;
            p = e < end ? e + 1 : end;
        }
        *n_crystals = c;
    }
    (void)last;
    return total;
}

int cl_host_crystfel_parse(const char* buf, long long nbytes, long long n_rows, float* cols, int nthreads) {
    if (buf == nullptr || nbytes < 0 || n_rows < 0 || (n_rows > 0 && cols == nullptr)) return -1;
    std::vector<RefBlock> blocks;
    try { find_blocks(buf, nbytes, blocks); } catch (...) { return -3; }
    RefBlock* B = blocks.data();
    const long long nb = (long long)blocks.size();
    // rows per list, then every list's first row in the table
    parallel_rows(nb, nthreads > 0 ? nthreads : 0, [=](long long a, long long b) {
        for (long long k = a; k < b; ++k) {
            long long rows = 0;
            for (const char* p = B[k].b; p < B[k].e;) {
                const char* e = line_end(p, B[k].e);
                if (count_fields(p, e) >= 9) ++rows;
                p = e < B[k].e ? e + 1 : B[k].e;
            }
            B[k].rows = rows;
        }
    }, 4);
    long long total = 0;
    for (auto& bl : blocks) { bl.first = total; total += bl.rows; }
    if (total != n_rows) return -1;
    std::vector<int> bad((size_t)(nb > 0 ? nb : 1), 0);
    int* badp = bad.data();
    parallel_rows(nb, nthreads > 0 ? nthreads : 0, [=](long long a, long long b) {
        char tok[64];
        for (long long k = a; k < b; ++k) {
            long long r = B[k].first;
            for (const char* p = B[k].b; p < B[k].e;) {
                const char* e = line_end(p, B[k].e);
                if (count_fields(p, e) >= 9) {
                    const char* q = p;
                    for (int c = 0; c < 9; ++c) {
                        while (q < e && is_blank(*q)) ++q;
                        const char* t0 = q;
                        while (q < e && !is_blank(*q)) ++q;
                        const size_t len = (size_t)(q - t0);
                        if (len == 0 || len >= sizeof(tok)) { badp[k] = 1; break; }
                        std::memcpy(tok, t0, len);
                        tok[len] = 0;
                        char* endp = nullptr;
                        double v;
                        if (c < 3) v = (double)std::strtoll(tok, &endp, 10);          // int(): h, k, l
                        else if (!fast_decimal(tok, len, &v)) v = std::strtod(tok, &endp), (void)0;      // float(): correctly rounded, then stored as fp32
                        else endp = tok + len;
                        if (endp != tok + len) { badp[k] = 1; break; }
                        cols[(size_t)c * (size_t)n_rows + (size_t)r] = (float)v;
                    }
                    cols[(size_t)9 * (size_t)n_rows + (size_t)r] = (float)B[k].batch;
                    ++r;
                }
                p = e < B[k].e ? e + 1 : B[k].e;
            }
        }
    }, 4);
    for (long long k = 0; k < nb; ++k) if (bad[(size_t)k]) return -5;                 // a field that is not a number
    return 0;
}

}  // extern "C"
