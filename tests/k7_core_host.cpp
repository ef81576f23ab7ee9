// TEST INFRASTRUCTURE: polyfuzz_amd/csrc/k7_core.h (K7's per-pair score and its upper bound) compiled for the CPU, so that
// tests/test_k7_core_cpu.py can hold both against the oracle on a box without a GPU.  Nothing in the product links this.
//
// k7_host_pairs(): for every (from i, to j) of two prepared lists -- prepared by the TEST in Python (forms as symbol
// ranks, tokens, tags, class histograms: the layout k7_fuzz.hip's device kernels build) -- the exact score under `mode`
// with running best cur[i], and the float32 upper bound (signature test, then the refined bound, as the kernel's sweep).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define PFZ_HD inline
#include "../polyfuzz_amd/csrc/k7_core.h"

using namespace pfz;

// 0: fz_score sweeps its windows itself; > 0: the kernel's way -- the sweeps deferred, every form's windows shared out in
// runs of this many, each run swept on its own, the pair's score the maximum of the parts
static int g_share = 0;
extern "C" void k7_host_set_share(int share) { g_share = share; }

// symbol-presence masks of the two lists ([n][kFuzzPresWords]; NULL: the bound as the kernel computes it today)
static const uint32_t *g_pres_a = nullptr, *g_pres_b = nullptr;
extern "C" void k7_host_set_presence(const uint32_t *a, const uint32_t *b) { g_pres_a = a, g_pres_b = b; }

// the window sweeps' experiment: live masks (FuzzSweep::has_live), and a count of the windows swept
static int g_live = 0;
static long long g_windows = 0;
extern "C" void k7_host_set_live(int on) { g_live = on; }
extern "C" long long k7_host_windows() { const long long n = g_windows; g_windows = 0; return n; }

template <int W>
static void set_live(FuzzSweep &S, const FuzzFrom<W> &F, const FuzzTo &T, int v)
{
    if (!g_live || F.la[v] > 64 || T.lb[v] > 64) return;
    S.has_live = true;
    for (int p = 0; p < T.lb[v]; ++p) {
        const uint64_t pmv = F.pm[((size_t)T.sym[v][p] * 3 + v) * W];      // (forms within 64 symbols: the first word is all)
        S.live_from |= pmv;
        S.live_to |= (uint64_t)(pmv != 0) << p;
    }
}

template <int W>
static double score_pair(const FuzzFrom<W> &F, FuzzTo &T, int mode, double cur)
{
    if (g_share <= 0) return fz_score<W>(F, T, mode, cur);
    int want = 0;
    double sc = fz_score<W, true>(F, T, mode, cur, &want);
    for (int v = 0; v < 3; ++v) {
        if (!((want >> v) & 1)) continue;
        const int n_win = fz_n_windows(F.la[v], T.lb[v]);
        const double f = fz_sweep_factor(mode, v, F.la[0], T.lb[0]);
        for (int w0 = 0; w0 < n_win; w0 += g_share) {
            FuzzSweep S;
            fz_sweep_begin(S, v, F.la[v], T.lb[v], w0, w0 + g_share < n_win ? w0 + g_share : n_win, T.sym[v], nullptr, 0);
            set_live<W>(S, F, T, v);
            if (S.has_live) {
                while (!fz_sweep_window<W, true>(S, F, f, cur - 1e-6)) {}
                g_windows += S.n_swept;
            }
            else {
                ++g_windows;
                while (!fz_sweep_window<W>(S, F, f, cur - 1e-6)) ++g_windows;
            }
            const double part = fz_sweep_score(mode, v, F.la[0], T.lb[0], fz_ratio_of(S.bl, S.bs));
            if (part > sc) sc = part;
        }
    }
    return sc;
}

struct List {
    int64_t n;
    const uint16_t *sym[3];
    const int64_t *off[3];
    const uint8_t *tag;            // form 2
    const int64_t *tok_off;
    const int32_t *tok_id, *tok_len;
    const uint32_t *hist;          // [n][kFuzzHistWords]
    const int32_t *usum;
};

template <int W>
static void run(int n_sym, const List &A, const List &B, int mode, const double *cur, double *out_score, float *out_ub)
{
    std::vector<uint64_t> pm((size_t)(n_sym + 1) * 3 * W);
    for (int64_t i = 0; i < A.n; ++i) {
        std::fill(pm.begin(), pm.end(), 0ull);
        FuzzFrom<W> F;
        for (int v = 0; v < 3; ++v) {
            const int64_t a0 = A.off[v][i];
            F.la.set(v, (int)(A.off[v][i + 1] - a0));
            for (int p = 0; p < F.la[v]; ++p) {
                const int sy = A.sym[v][a0 + p];
                if (sy) pm[((size_t)sy * 3 + v) * W + (p >> 6)] |= 1ull << (p & 63);
            }
        }
        const int64_t t0 = A.tok_off[i];
        F.ta = (int)(A.tok_off[i + 1] - t0);
        std::vector<int32_t> tid(F.ta + 1), tlen(F.ta + 1);
        std::vector<uint64_t> tmask((size_t)(F.ta + 1) * W), smask((size_t)(F.ta + 1) * W);
        int start = 0;
        FuzzSummary sa;
        sa.sig = 0;
        for (int t = 0; t < F.ta; ++t) {
            tid[t] = A.tok_id[t0 + t];
            tlen[t] = A.tok_len[t0 + t];
            uint64_t tm[W], sm[W];
            fz_range_mask<W>(tm, start, start + tlen[t]);
            fz_range_mask<W>(sm, start + tlen[t], t + 1 < F.ta ? start + tlen[t] + 1 : start + tlen[t]);
            for (int w = 0; w < W; ++w) {
                tmask[(size_t)t * W + w] = tm[w];
                smask[(size_t)t * W + w] = sm[w];
            }
            start += tlen[t] + 1;
            sa.sig |= fz_sig_bit(tid[t]);
        }
        F.pm = pm.data();
        F.tid = tid.data();
        F.tlen = tlen.data();
        F.tmask = tmask.data();
        F.smask = smask.data();
        for (int v = 0; v < 3; ++v) sa.len[v] = F.la[v];
        sa.ntok = F.ta;
        memcpy(sa.hist, A.hist + (size_t)i * kFuzzHistWords, sizeof(sa.hist));
        sa.usum = A.usum[i];
        for (int64_t j = 0; j < B.n; ++j) {
            FuzzTo T;
            FuzzSummary sb;
            sb.sig = 0;
            for (int v = 0; v < 3; ++v) {
                T.sym.set(v, B.sym[v] + B.off[v][j]);
                T.lb.set(v, sb.len[v] = (int)(B.off[v][j + 1] - B.off[v][j]));
            }
            T.tag = B.tag + B.off[2][j];
            T.tok_id = B.tok_id + B.tok_off[j];
            T.tok_len = B.tok_len + B.tok_off[j];
            T.stage = nullptr;
            T.stage_stride = 0;
            T.staged = -1;
            T.n_windows = 0;
            T.tb = sb.ntok = (int)(B.tok_off[j + 1] - B.tok_off[j]);
            for (int t = 0; t < T.tb; ++t) sb.sig |= fz_sig_bit(T.tok_id[t]);
            memcpy(sb.hist, B.hist + (size_t)j * kFuzzHistWords, sizeof(sb.hist));
            sb.usum = B.usum[j];
            out_score[i * B.n + j] = score_pair<W>(F, T, mode, cur[i]);
            // the sweep's two-step bound: signatures first, the exact intersection only when they meet
            const int u = fz_common_chars(sa, sb);
            const bool maybe = (sa.sig & sb.sig) != 0;
            int miss_a = 0, miss_b = 0;
            if (g_pres_a) fz_presence_miss(g_pres_a + (size_t)i * kFuzzPresWords, g_pres_b + (size_t)j * kFuzzPresWords, miss_a, miss_b);
            float ub = fz_upper_bound(sa, sb, mode, u, maybe ? -1 : 0, -1.0f, miss_a, miss_b);
            if (maybe) {
                uint32_t ca, cb;
                fz_intersect<W>(F, T, ca, cb);
                const float ub2 = fz_upper_bound(sa, sb, mode, u, ca ? 1 : 0,
                                                 ca ? fz_token_set_bound<W>(F, ca, T.lb[2], T.tb, u, miss_a, miss_b) : -1.0f, miss_a, miss_b);
                if (ub2 > ub + 1e-3f) ub = -1000.0f;      // the refined bound must never exceed the coarse one (flagged for the test)
                else ub = ub2;
            }
            out_ub[i * B.n + j] = ub;
        }
    }
}

extern "C" int k7_host_pairs(int W, int n_sym, int64_t na, const uint16_t *a0, const int64_t *ao0, const uint16_t *a1,
                             const int64_t *ao1, const uint16_t *a2, const int64_t *ao2, const uint8_t *a_tag,
                             const int64_t *a_tok_off, const int32_t *a_tok_id, const int32_t *a_tok_len, const uint32_t *a_hist,
                             const int32_t *a_usum, int64_t nb, const uint16_t *b0, const int64_t *bo0, const uint16_t *b1,
                             const int64_t *bo1, const uint16_t *b2, const int64_t *bo2, const uint8_t *b_tag,
                             const int64_t *b_tok_off, const int32_t *b_tok_id, const int32_t *b_tok_len, const uint32_t *b_hist,
                             const int32_t *b_usum, int mode, const double *cur, double *out_score, float *out_ub)
{
    const List A{na, {a0, a1, a2}, {ao0, ao1, ao2}, a_tag, a_tok_off, a_tok_id, a_tok_len, a_hist, a_usum};
    const List B{nb, {b0, b1, b2}, {bo0, bo1, bo2}, b_tag, b_tok_off, b_tok_id, b_tok_len, b_hist, b_usum};
    if (W == 1) run<1>(n_sym, A, B, mode, cur, out_score, out_ub);
    else if (W == 2) run<2>(n_sym, A, B, mode, cur, out_score, out_ub);
    else if (W == 4) run<4>(n_sym, A, B, mode, cur, out_score, out_ub);
    else return 1;
    return 0;
}
