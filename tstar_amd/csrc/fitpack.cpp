// Attribution: the algorithm restated here is FITPACK (P. Dierckx, K.U. Leuven; distributed through netlib -- netlib.org/dierckx --
// and wrapped by scipy.interpolate, BSD-licensed there).  Routine and variable names (fpcurf, fpknot, fpdisc, fpgivs, fprota, fprati,
// fpbspl, fpback; store, dd, stor1 ...) follow the library so that the two can be read side by side; the code itself was
// written for this tree from the published description and from scipy's observable behaviour.
//
// Host-side smoothing-spline fit of the T* sampling distribution: a C++ restatement of P. Dierckx's FITPACK `curfit`
// (fpcurf / fpknot / fpdisc / fpgivs / fprota / fprati / fpbspl / fpback) for the one call the reference makes --
//     scipy.interpolate.UnivariateSpline(visited_indices, observed_scores, s=0.5)
// (/root/reference/TStar/interface_searcher.py:265: k = 3, unit weights, xb = x[0], xe = x[m-1], tol = 0.001, maxit = 20, and
// scipy's own call sequence: fpcurf0 with the f2py default nest = max(m / 2, 8), then -- when that stops with "nest too
// small" -- fpcurf1 continuing with nest = m + k + 1, which restarts the knot increment at one; the knots depend on this
// sequence) -- with every floating-point operation in the library's order, compiled
// without contraction (-ffp-contract=off), so that the knots t, the coefficients c and the residual fp are BIT-IDENTICAL
// to scipy's (tests/test_host_logic.py::test_native_curfit_bit_identical_to_scipy: all 63 fits of a reference-default
// search, golden G4's problems, randomized problems; algorithm: Dierckx, "Curve and Surface Fitting with Splines", 1993,
// restated from its published description -- the library's Fortran is not in this image).
//
// Why restate it: with the reference's default 4x4 grid a search makes 63 fits on a growing set of visited frames
// (16 .. 1008 points, up to ~800 knots) and each sits on the critical path between two iterations.  The library spends
// almost all of a late fit in the smoothing-parameter iteration: every one of the n - 8 discontinuity rows is rotated
// through the WHOLE remaining band matrix (O(n^2) dependent Givens rotations per trial value of p, 3-8 trials):
// 75 ms at m = 1008.  The rotations of consecutive rows form a systolic pipeline -- row R can process column j as soon
// as row R - 1 has left it -- so W rows run in lock-step, skewed by two columns, each lane executing exactly the scalar
// operation sequence of its row on exactly the values the sequential loop would see: same IEEE operations on the same
// operands, hence the same bits, at ~1/W of the dependent-chain length (AVX2: W = 4, AVX-512: W = 8).  In the steady state
// of a batch the column data move through the lanes like a shift register (one scalar load and one scalar store per array and
// step); a Givens rotation costs three divisions and one square root (operands selected BEFORE the division, which is what
// the library's two branches compute).  The widest form the CPU supports is picked at run time; all forms give the same bits.
// Round 5: two consecutive batches run their steady states in ONE loop (rotate_pair: the second batch 2 W columns behind the first),
// i.e. two independent division / root chains in flight per iteration: the 63 fits of a reference-default search 179 -> 146 ms on the
// GPU box's EPYC 9575F (AVX-512), 392 -> 324 ms on the build container's Xeon; same bits (tests run both forms).
//
// Not on the GPU: the fit is a few hundred KB of sequential, latency-bound float64 work per search iteration.
#if defined(__x86_64__)
#include <immintrin.h>
#define TSTAR_FITPACK_SIMD 1
#else
#define TSTAR_FITPACK_SIMD 0          // other hosts: the sequential form only (same bits)
#endif
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline void fpgivs(double piv, double& ww, double& cs, double& sn) {
    const double one = 1.0;
    const double store = std::fabs(piv);
    double dd;
    if (store >= ww) { const double r = ww / piv; dd = store * std::sqrt(one + r * r); }
    else { const double r = piv / ww; dd = ww * std::sqrt(one + r * r); }
    cs = ww / dd;
    sn = piv / dd;
    ww = dd;
}

inline void fprota(double cs, double sn, double& a, double& b) {
    const double stor1 = a, stor2 = b;
    b = cs * stor2 + sn * stor1;
    a = cs * stor1 - sn * stor2;
}

// the k + 1 = 4 non-zero cubic B-splines at x, t[l] <= x < t[l+1] (1-based l as in the library)
inline void fpbspl3(const double* t /*1-based*/, double x, int l, double* h /*1-based, >= 5*/) {
    double hh[5];
    h[1] = 1.0;
    for (int j = 1; j <= 3; ++j) {
        for (int i = 1; i <= j; ++i) hh[i] = h[i];
        h[1] = 0.0;
        for (int i = 1; i <= j; ++i) {
            const int li = l + i, lj = li - j;
            if (t[li] == t[lj]) { h[i + 1] = 0.0; continue; }
            const double f = hh[i] / (t[li] - t[lj]);
            h[i] = h[i] + f * (t[li] - x);
            h[i + 1] = f * (x - t[lj]);
        }
    }
}

// back substitution of an n x n upper-triangular band system of bandwidth k; A(i, l) = acc(i, l) (1-based)
template <class Acc>
inline void fpback(Acc A, const double* z, int n, int k, double* c) {
    const int k1 = k - 1;
    c[n] = z[n] / A(n, 1);
    int i = n - 1;
    if (i == 0) return;
    for (int j = 2; j <= n; ++j) {
        double store = z[i];
        int i1 = k1;
        if (j <= k1) i1 = j - 1;
        int mm = i;
        for (int l = 1; l <= i1; ++l) {
            ++mm;
            store = store - c[mm] * A(i, l + 1);
        }
        c[i] = store / A(i, 1);
        --i;
    }
}

// discontinuity jumps of the k-th derivative of the B-splines at the interior knots (k2 = 5)
inline void fpdisc5(const double* t, int n, double* b /* b(i,j) at b[(i)*5 + j - 1], i 1-based */) {
    const int k2 = 5, k1 = 4, k = 3;
    const int nk1 = n - k1, nrint = nk1 - k;
    const double an = nrint;
    const double fac = an / (t[nk1 + 1] - t[k1]);
    double h[13];
    for (int l = k2; l <= nk1; ++l) {
        const int lmk = l - k1;
        for (int j = 1; j <= k1; ++j) {
            const int ik = j + k1, lj = l + j, lk = lj - k2;
            h[j] = t[l] - t[lk];
            h[ik] = t[l] - t[lj];
        }
        int lp = lmk;
        for (int j = 1; j <= k2; ++j) {
            int jk = j;
            double prod = h[j];
            for (int i = 1; i <= k; ++i) {
                ++jk;
                prod = prod * h[jk] * fac;
            }
            const int lk = lp + k1;
            b[lmk * 5 + j - 1] = (t[lk] - t[lp]) / prod;
            ++lp;
        }
    }
}

inline void fpknot(const double* x, double* t, int& n, double* fpint, int* nrdata, int& nrint, int istart) {
    const int k = (n - nrint - 1) / 2;
    double fpmax = 0.0;
    int jbegin = istart, number = 0, maxpt = 0, maxbeg = 0;
    for (int j = 1; j <= nrint; ++j) {
        const int jpoint = nrdata[j];
        if (!(fpmax >= fpint[j] || jpoint == 0)) {
            fpmax = fpint[j];
            number = j;
            maxpt = jpoint;
            maxbeg = jbegin;
        }
        jbegin = jbegin + jpoint + 1;
    }
    const int ihalf = maxpt / 2 + 1;
    const int nrx = maxbeg + ihalf;
    const int next = number + 1;
    if (next <= nrint) {
        for (int j = next; j <= nrint; ++j) {
            const int jj = next + nrint - j;
            fpint[jj + 1] = fpint[jj];
            nrdata[jj + 1] = nrdata[jj];
            const int jk = jj + k;
            t[jk + 1] = t[jk];
        }
    }
    nrdata[number] = ihalf - 1;
    nrdata[next] = maxpt - ihalf;
    const double am = maxpt;
    double an = nrdata[number];
    fpint[number] = fpmax * an / am;
    an = nrdata[next];
    fpint[next] = fpmax * an / am;
    const int jk = next + k;
    t[jk] = x[nrx];
    n = n + 1;
    nrint = nrint + 1;
}

inline double fprati(double& p1, double& f1, double p2, double f2, double& p3, double& f3) {
    double p;
    if (p3 > 0.0) {
        const double h1 = f1 * (f2 - f3);
        const double h2 = f2 * (f3 - f1);
        const double h3 = f3 * (f1 - f2);
        p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
    } else {
        p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
    }
    if (f2 < 0.0) { p3 = p2; f3 = f2; }
    else { p1 = p2; f1 = f2; }
    return p;
}

// ---- the O(n^2) step: rows 1..n8 of b (weighted by 1/p) rotated into the band matrix g (5 columns) and the right-hand side c.
// g column i (1-based) lives at g + (i - 1) * gs + PAD, row j (1-based) at [.. + j]; PAD elements of slack on both sides.
constexpr int PAD = 16;
static bool env_off(const char* k) { const char* v = std::getenv(k); return v && v[0] == '0'; }
// two batches per steady loop; TSTAR_FITPACK_PAIR=0 or tstar_curfit_pairing(0) turn it off (tests / timing: same bits either way)
static bool g_pair_batches = !env_off("TSTAR_FITPACK_PAIR");

// sequential form (the library's loop), used for the rows a full batch does not cover and as the reference of the skewed form
inline void rotate_rows_seq(const double* b, double pinv, double* g, int gs, double* c, int it0, int it1, int nk1, int n8) {
    double h[7];
    for (int it = it0; it <= it1; ++it) {
        for (int i = 1; i <= 5; ++i) h[i] = b[it * 5 + i - 1] * pinv;
        double yi = 0.0;
        for (int j = it; j <= nk1; ++j) {
            const double piv = h[1];
            double cs, sn;
            fpgivs(piv, g[PAD + j], cs, sn);
            fprota(cs, sn, yi, c[j]);
            if (j == nk1) break;
            int i2 = 4;
            if (j > n8) i2 = nk1 - j;
            for (int i = 1; i <= i2; ++i) {
                fprota(cs, sn, h[i + 1], g[i * gs + PAD + j]);
                h[i] = h[i + 1];
            }
            h[i2 + 1] = 0.0;
        }
    }
}

// W rows it0 .. it0 + W - 1 in lock-step, row r two columns behind row r - 1: at time T lane l (row it0 + W - 1 - l) works on
// column base + l, base = it0 + T - W + 1.  Lanes outside their row's life (before its start, past column nk1) compute on
// padding and are masked out of every store.  The per-row state (h1..h5, yi) lives in `st`, so the time range can be cut
// into a masked generic part (pipeline fill, the last five columns) and the register-resident steady state below.
template <int W>
struct RowState { double h1[W], h2[W], h3[W], h4[W], h5[W], yi[W]; };

template <int W>
inline void rotate_rows_skewed(RowState<W>& st, double* g, int gs, double* c, int it0, int nk1, int n8, int T0, int T1) {
    double (&h1)[W] = st.h1; double (&h2)[W] = st.h2; double (&h3)[W] = st.h3; double (&h4)[W] = st.h4; double (&h5)[W] = st.h5;
    double (&yi)[W] = st.yi;
    double* g1 = g + PAD;
    double* g2 = g + gs + PAD;
    double* g3 = g + 2 * gs + PAD;
    double* g4 = g + 3 * gs + PAD;
    double* g5 = g + 4 * gs + PAD;
    for (int T = T0; T <= T1; ++T) {
        const int base = it0 + T - W + 1;                // may be below 1 for the first steps: padding absorbs it
        double act[W], m1[W], m2[W], m3[W], m4[W];       // 1.0 / 0.0 masks
        for (int l = 0; l < W; ++l) {
            const int r = W - 1 - l, j = base + l;
            const bool a = (T >= 2 * r) && (j <= nk1);
            const int i2 = !a ? -1 : (j == nk1 ? 0 : (j > n8 ? nk1 - j : 4));
            act[l] = a ? 1.0 : 0.0;
            m1[l] = i2 >= 1 ? 1.0 : 0.0; m2[l] = i2 >= 2 ? 1.0 : 0.0; m3[l] = i2 >= 3 ? 1.0 : 0.0; m4[l] = i2 >= 4 ? 1.0 : 0.0;
        }
        double cs[W], sn[W];
        for (int l = 0; l < W; ++l) {                    // fpgivs, branch-free: both forms evaluated, one selected
            const double piv = h1[l], ww = g1[base + l];
            const double store = std::fabs(piv);
            const bool ge = store >= ww;                 // one division and one root on the selected operands: the library's two forms
            const double r = (ge ? ww : piv) / (ge ? piv : ww);
            const double dd = (ge ? store : ww) * std::sqrt(1.0 + r * r);
            cs[l] = ww / dd;
            sn[l] = piv / dd;
            g1[base + l] = act[l] != 0.0 ? dd : ww;
        }
        for (int l = 0; l < W; ++l) {                    // right-hand side
            const double s1 = yi[l], s2 = c[base + l];
            const double nb = cs[l] * s2 + sn[l] * s1;
            const double na = cs[l] * s1 - sn[l] * s2;
            c[base + l] = act[l] != 0.0 ? nb : s2;
            yi[l] = act[l] != 0.0 ? na : s1;
        }
        for (int l = 0; l < W; ++l) {                    // left-hand side: (h2, g2) -> h1, (h3, g3) -> h2, (h4, g4) -> h3, (h5, g5) -> h4, h5 = 0
            const double c_ = cs[l], s_ = sn[l];
            const double a2 = h2[l], b2 = g2[base + l];
            const double a3 = h3[l], b3 = g3[base + l];
            const double a4 = h4[l], b4 = g4[base + l];
            const double a5 = h5[l], b5 = g5[base + l];
            const double nb2 = c_ * b2 + s_ * a2, na2 = c_ * a2 - s_ * b2;
            const double nb3 = c_ * b3 + s_ * a3, na3 = c_ * a3 - s_ * b3;
            const double nb4 = c_ * b4 + s_ * a4, na4 = c_ * a4 - s_ * b4;
            const double nb5 = c_ * b5 + s_ * a5, na5 = c_ * a5 - s_ * b5;
            g2[base + l] = m1[l] != 0.0 ? nb2 : b2;
            g3[base + l] = m2[l] != 0.0 ? nb3 : b3;
            g4[base + l] = m3[l] != 0.0 ? nb4 : b4;
            g5[base + l] = m4[l] != 0.0 ? nb5 : b5;
            const bool on = act[l] != 0.0;
            h1[l] = on ? (m1[l] != 0.0 ? na2 : 0.0) : h1[l];
            h2[l] = on ? (m2[l] != 0.0 ? na3 : 0.0) : h2[l];
            h3[l] = on ? (m3[l] != 0.0 ? na4 : 0.0) : h3[l];
            h4[l] = on ? (m4[l] != 0.0 ? na5 : 0.0) : h4[l];
            h5[l] = on ? 0.0 : h5[l];
        }
    }
}

// Steady state of a batch (every lane inside its row, all columns <= n8, i.e. four left-hand rotations per step): the column data
// move through the lanes like a shift register -- column j enters at the top lane (the batch's first row), is handed one lane down
// per step and leaves, final for this batch, at lane 0 -- so a step costs one scalar load and one scalar store per array instead of
// six overlapping vector round trips through memory.  Same operations, same operands, same order as rotate_rows_seq per (row, column).
#define TSTAR_STEADY_BODY(VT, LOADU, STOREU, SET1, MUL, ADD, SUB, DIV, SQRT, ABS, SELECT_GE, LANE0, SHIFT_IN, ZERO)                        \
    VT H1 = LOADU(st.h1), H2 = LOADU(st.h2), H3 = LOADU(st.h3), H4 = LOADU(st.h4), H5 = LOADU(st.h5), YI = LOADU(st.yi);                     \
    int base = it0 + T0 - W + 1;                                                                                                            \
    VT G1 = LOADU(g1 + base), G2 = LOADU(g2 + base), G3 = LOADU(g3 + base), G4 = LOADU(g4 + base), G5 = LOADU(g5 + base), CC = LOADU(c + base); \
    const VT one = SET1(1.0);                                                                                                               \
    for (int T = T0; T <= T1; ++T, ++base) {                                                                                                \
        const VT piv = H1, ww = G1;                                                                                                         \
        const VT sa = ABS(piv);                                                                                                             \
        /* fpgivs: |piv| >= ww ? |piv| sqrt(1 + (ww / piv)^2) : ww sqrt(1 + (piv / ww)^2) -- operands selected first, one division, one root */ \
        const VT rr = DIV(SELECT_GE(sa, ww, ww, piv), SELECT_GE(sa, ww, piv, ww));                                                          \
        const VT dd = MUL(SELECT_GE(sa, ww, sa, ww), SQRT(ADD(one, MUL(rr, rr))));                                                          \
        const VT cs = DIV(ww, dd), sn = DIV(piv, dd);                                                                                       \
        G1 = dd;                                                                                                                            \
        const VT nc = ADD(MUL(cs, CC), MUL(sn, YI));                                                                                        \
        YI = SUB(MUL(cs, YI), MUL(sn, CC));                                                                                                 \
        CC = nc;                                                                                                                            \
        const VT n2 = ADD(MUL(cs, G2), MUL(sn, H2)); H1 = SUB(MUL(cs, H2), MUL(sn, G2)); G2 = n2;                                           \
        const VT n3 = ADD(MUL(cs, G3), MUL(sn, H3)); H2 = SUB(MUL(cs, H3), MUL(sn, G3)); G3 = n3;                                           \
        const VT n4 = ADD(MUL(cs, G4), MUL(sn, H4)); H3 = SUB(MUL(cs, H4), MUL(sn, G4)); G4 = n4;                                           \
        const VT n5 = ADD(MUL(cs, G5), MUL(sn, H5)); H4 = SUB(MUL(cs, H5), MUL(sn, G5)); G5 = n5;                                           \
        H5 = ZERO();                                                                                                                        \
        g1[base] = LANE0(G1); g2[base] = LANE0(G2); g3[base] = LANE0(G3); g4[base] = LANE0(G4); g5[base] = LANE0(G5); c[base] = LANE0(CC);  \
        G1 = SHIFT_IN(G1, g1[base + W]); G2 = SHIFT_IN(G2, g2[base + W]); G3 = SHIFT_IN(G3, g3[base + W]);                                  \
        G4 = SHIFT_IN(G4, g4[base + W]); G5 = SHIFT_IN(G5, g5[base + W]); CC = SHIFT_IN(CC, c[base + W]);                                   \
    }                                                                                                                                       \
    /* hand the register-resident columns [base, base + W - 2] back (the entering one at the top lane was only loaded) */                   \
    STOREU(g1 + base, G1); STOREU(g2 + base, G2); STOREU(g3 + base, G3); STOREU(g4 + base, G4); STOREU(g5 + base, G5); STOREU(c + base, CC); \
    STOREU(st.h1, H1); STOREU(st.h2, H2); STOREU(st.h3, H3); STOREU(st.h4, H4); STOREU(st.h5, H5); STOREU(st.yi, YI);

// Two batches in one loop (round 5): batch A = rows it0 .. it0 + W - 1 at time T, batch B = the next W rows at time T - 2 W, i.e. B's
// columns are A's shifted down by W (baseB = baseA - W): the column that leaves A at lane 0 in a step is the one B takes in at its top
// lane at the end of the same step, so B never touches a column A has not finished -- the sequential order per (row, column) is kept
// and so are the bits.  A step of one batch is ONE dependent chain (division -> root -> two divisions -> the next pivot: ~60-90
// cycles of latency per column); two independent chains per loop iteration let the out-of-order core overlap them.
#define TSTAR_STEADY_STEP(S, VT, SET1, MUL, ADD, SUB, DIV, SQRT, ABS, SELECT_GE, LANE0, SHIFT_IN, ZERO)                                     \
    {                                                                                                                                       \
        const VT piv = H1##S, ww = G1##S;                                                                                                   \
        const VT sa = ABS(piv);                                                                                                             \
        const VT rr = DIV(SELECT_GE(sa, ww, ww, piv), SELECT_GE(sa, ww, piv, ww));                                                          \
        const VT dd = MUL(SELECT_GE(sa, ww, sa, ww), SQRT(ADD(one, MUL(rr, rr))));                                                          \
        const VT cs = DIV(ww, dd), sn = DIV(piv, dd);                                                                                       \
        G1##S = dd;                                                                                                                         \
        const VT nc = ADD(MUL(cs, CC##S), MUL(sn, YI##S));                                                                                  \
        YI##S = SUB(MUL(cs, YI##S), MUL(sn, CC##S));                                                                                        \
        CC##S = nc;                                                                                                                         \
        const VT n2 = ADD(MUL(cs, G2##S), MUL(sn, H2##S)); H1##S = SUB(MUL(cs, H2##S), MUL(sn, G2##S)); G2##S = n2;                         \
        const VT n3 = ADD(MUL(cs, G3##S), MUL(sn, H3##S)); H2##S = SUB(MUL(cs, H3##S), MUL(sn, G3##S)); G3##S = n3;                         \
        const VT n4 = ADD(MUL(cs, G4##S), MUL(sn, H4##S)); H3##S = SUB(MUL(cs, H4##S), MUL(sn, G4##S)); G4##S = n4;                         \
        const VT n5 = ADD(MUL(cs, G5##S), MUL(sn, H5##S)); H4##S = SUB(MUL(cs, H5##S), MUL(sn, G5##S)); G5##S = n5;                         \
        H5##S = ZERO();                                                                                                                     \
        g1[base##S] = LANE0(G1##S); g2[base##S] = LANE0(G2##S); g3[base##S] = LANE0(G3##S); g4[base##S] = LANE0(G4##S);                     \
        g5[base##S] = LANE0(G5##S); c[base##S] = LANE0(CC##S);                                                                              \
        G1##S = SHIFT_IN(G1##S, g1[base##S + W]); G2##S = SHIFT_IN(G2##S, g2[base##S + W]); G3##S = SHIFT_IN(G3##S, g3[base##S + W]);       \
        G4##S = SHIFT_IN(G4##S, g4[base##S + W]); G5##S = SHIFT_IN(G5##S, g5[base##S + W]); CC##S = SHIFT_IN(CC##S, c[base##S + W]);        \
    }
#define TSTAR_STEADY2_BODY(VT, LOADU, STOREU, SET1, MUL, ADD, SUB, DIV, SQRT, ABS, SELECT_GE, LANE0, SHIFT_IN, ZERO)                        \
    VT H1A = LOADU(sa_.h1), H2A = LOADU(sa_.h2), H3A = LOADU(sa_.h3), H4A = LOADU(sa_.h4), H5A = LOADU(sa_.h5), YIA = LOADU(sa_.yi);         \
    VT H1B = LOADU(sb_.h1), H2B = LOADU(sb_.h2), H3B = LOADU(sb_.h3), H4B = LOADU(sb_.h4), H5B = LOADU(sb_.h5), YIB = LOADU(sb_.yi);         \
    int baseA = it0 + T0 - W + 1, baseB = baseA - W;                                                                                        \
    /* B's register-resident columns first: they end where A's begin (B's entering column baseB + W - 1 is A's baseA - 1, final) */        \
    VT G1B = LOADU(g1 + baseB), G2B = LOADU(g2 + baseB), G3B = LOADU(g3 + baseB), G4B = LOADU(g4 + baseB), G5B = LOADU(g5 + baseB),         \
       CCB = LOADU(c + baseB);                                                                                                              \
    VT G1A = LOADU(g1 + baseA), G2A = LOADU(g2 + baseA), G3A = LOADU(g3 + baseA), G4A = LOADU(g4 + baseA), G5A = LOADU(g5 + baseA),         \
       CCA = LOADU(c + baseA);                                                                                                              \
    const VT one = SET1(1.0);                                                                                                               \
    for (int T = T0; T <= T1; ++T, ++baseA, ++baseB) {                                                                                      \
        TSTAR_STEADY_STEP(A, VT, SET1, MUL, ADD, SUB, DIV, SQRT, ABS, SELECT_GE, LANE0, SHIFT_IN, ZERO)                                     \
        TSTAR_STEADY_STEP(B, VT, SET1, MUL, ADD, SUB, DIV, SQRT, ABS, SELECT_GE, LANE0, SHIFT_IN, ZERO)                                     \
    }                                                                                                                                       \
    /* A's columns back first, then B's (B's top lane holds column baseB + W - 1 = baseA - 1 as loaded: the same value A stored) */         \
    STOREU(g1 + baseA, G1A); STOREU(g2 + baseA, G2A); STOREU(g3 + baseA, G3A); STOREU(g4 + baseA, G4A); STOREU(g5 + baseA, G5A);            \
    STOREU(c + baseA, CCA);                                                                                                                 \
    STOREU(g1 + baseB, G1B); STOREU(g2 + baseB, G2B); STOREU(g3 + baseB, G3B); STOREU(g4 + baseB, G4B); STOREU(g5 + baseB, G5B);            \
    STOREU(c + baseB, CCB);                                                                                                                 \
    STOREU(sa_.h1, H1A); STOREU(sa_.h2, H2A); STOREU(sa_.h3, H3A); STOREU(sa_.h4, H4A); STOREU(sa_.h5, H5A); STOREU(sa_.yi, YIA);           \
    STOREU(sb_.h1, H1B); STOREU(sb_.h2, H2B); STOREU(sb_.h3, H3B); STOREU(sb_.h4, H4B); STOREU(sb_.h5, H5B); STOREU(sb_.yi, YIB);

#if TSTAR_FITPACK_SIMD
__attribute__((target("avx512f,avx512dq"))) inline void steady8(RowState<8>& st, double* g, int gs, double* c, int it0, int T0, int T1) {
    constexpr int W = 8;
    double* g1 = g + PAD; double* g2 = g + gs + PAD; double* g3 = g + 2 * gs + PAD; double* g4 = g + 3 * gs + PAD; double* g5 = g + 4 * gs + PAD;
#define SEL512(a, b, x, y) _mm512_mask_blend_pd(_mm512_cmp_pd_mask(a, b, _CMP_GE_OQ), y, x)
#define SHIFT512(v, s) _mm512_castsi512_pd(_mm512_alignr_epi64(_mm512_castpd_si512(_mm512_set1_pd(s)), _mm512_castpd_si512(v), 1))
#define LANE512(v) _mm_cvtsd_f64(_mm512_castpd512_pd128(v))
    TSTAR_STEADY_BODY(__m512d, _mm512_loadu_pd, _mm512_storeu_pd, _mm512_set1_pd, _mm512_mul_pd, _mm512_add_pd, _mm512_sub_pd, _mm512_div_pd,
                      _mm512_sqrt_pd, _mm512_abs_pd, SEL512, LANE512, SHIFT512, _mm512_setzero_pd)
}
// batch A (rows it0 ..) at times T0 .. T1 together with batch B (rows it0 + 8 ..) at times T0 - 16 .. T1 - 16
__attribute__((target("avx512f,avx512dq"))) inline void steady8x2(RowState<8>& sa_, RowState<8>& sb_, double* g, int gs, double* c, int it0, int T0, int T1) {
    constexpr int W = 8;
    double* g1 = g + PAD; double* g2 = g + gs + PAD; double* g3 = g + 2 * gs + PAD; double* g4 = g + 3 * gs + PAD; double* g5 = g + 4 * gs + PAD;
    TSTAR_STEADY2_BODY(__m512d, _mm512_loadu_pd, _mm512_storeu_pd, _mm512_set1_pd, _mm512_mul_pd, _mm512_add_pd, _mm512_sub_pd, _mm512_div_pd,
                       _mm512_sqrt_pd, _mm512_abs_pd, SEL512, LANE512, SHIFT512, _mm512_setzero_pd)
}

__attribute__((target("avx2"))) inline void steady4(RowState<4>& st, double* g, int gs, double* c, int it0, int T0, int T1) {
    constexpr int W = 4;
    double* g1 = g + PAD; double* g2 = g + gs + PAD; double* g3 = g + 2 * gs + PAD; double* g4 = g + 3 * gs + PAD; double* g5 = g + 4 * gs + PAD;
#define ABS256(v) _mm256_andnot_pd(_mm256_set1_pd(-0.0), v)
#define SEL256(a, b, x, y) _mm256_blendv_pd(y, x, _mm256_cmp_pd(a, b, _CMP_GE_OQ))
#define SHIFT256(v, s) _mm256_blend_pd(_mm256_permute4x64_pd(v, 0xF9), _mm256_set1_pd(s), 0x8)
#define LANE256(v) _mm_cvtsd_f64(_mm256_castpd256_pd128(v))
    TSTAR_STEADY_BODY(__m256d, _mm256_loadu_pd, _mm256_storeu_pd, _mm256_set1_pd, _mm256_mul_pd, _mm256_add_pd, _mm256_sub_pd, _mm256_div_pd,
                      _mm256_sqrt_pd, ABS256, SEL256, LANE256, SHIFT256, _mm256_setzero_pd)
}
__attribute__((target("avx2"))) inline void steady4x2(RowState<4>& sa_, RowState<4>& sb_, double* g, int gs, double* c, int it0, int T0, int T1) {
    constexpr int W = 4;
    double* g1 = g + PAD; double* g2 = g + gs + PAD; double* g3 = g + 2 * gs + PAD; double* g4 = g + 3 * gs + PAD; double* g5 = g + 4 * gs + PAD;
    TSTAR_STEADY2_BODY(__m256d, _mm256_loadu_pd, _mm256_storeu_pd, _mm256_set1_pd, _mm256_mul_pd, _mm256_add_pd, _mm256_sub_pd, _mm256_div_pd,
                       _mm256_sqrt_pd, ABS256, SEL256, LANE256, SHIFT256, _mm256_setzero_pd)
}

#endif

template <int W>
inline void rotate_batch(const double* b, double pinv, double* g, int gs, double* c, int it0, int nk1, int n8) {
    RowState<W> st;
    for (int l = 0; l < W; ++l) {
        const int it = it0 + W - 1 - l;
        st.h1[l] = b[it * 5 + 0] * pinv; st.h2[l] = b[it * 5 + 1] * pinv; st.h3[l] = b[it * 5 + 2] * pinv;
        st.h4[l] = b[it * 5 + 3] * pinv; st.h5[l] = b[it * 5 + 4] * pinv;
        st.yi[l] = 0.0;
    }
    const int t_end = nk1 - it0 + W - 1;                 // last time step: the last row (lane 0) reaches column nk1
    const int s0 = 2 * (W - 1), s1 = n8 - it0;           // steady state: every lane started, top lane's column it0 + T <= n8
    if (s1 >= s0) {
        rotate_rows_skewed<W>(st, g, gs, c, it0, nk1, n8, 0, s0 - 1);
#if TSTAR_FITPACK_SIMD
        if constexpr (W == 8) steady8(st, g, gs, c, it0, s0, s1);
        else steady4(st, g, gs, c, it0, s0, s1);
#endif
        rotate_rows_skewed<W>(st, g, gs, c, it0, nk1, n8, s1 + 1, t_end);
    } else {
        rotate_rows_skewed<W>(st, g, gs, c, it0, nk1, n8, 0, t_end);
    }
}

// Two consecutive batches (rows it0 .. it0 + 2 W - 1) with their steady states run as one loop: A fills and runs 2 W steps ahead,
// B fills behind it, both run together while A is in its steady state, A drains, B finishes alone.  Any interleaving in which B
// never touches a column A has not left is the sequential order per (row, column); each piece below is one of the single-batch forms.
template <int W>
inline void rotate_pair(const double* b, double pinv, double* g, int gs, double* c, int it0, int nk1, int n8) {
    RowState<W> sa, sb;
    auto init = [&](RowState<W>& st, int i0) {
        for (int l = 0; l < W; ++l) {
            const int it = i0 + W - 1 - l;
            st.h1[l] = b[it * 5 + 0] * pinv; st.h2[l] = b[it * 5 + 1] * pinv; st.h3[l] = b[it * 5 + 2] * pinv;
            st.h4[l] = b[it * 5 + 3] * pinv; st.h5[l] = b[it * 5 + 4] * pinv;
            st.yi[l] = 0.0;
        }
    };
    init(sa, it0); init(sb, it0 + W);
    const int itb = it0 + W;
    const int s0 = 2 * (W - 1), s1a = n8 - it0, s1b = n8 - itb;            // steady ranges [s0, s1] of the two batches (own time)
    const int tea = nk1 - it0 + W - 1, teb = nk1 - itb + W - 1;            // their last time steps
    const int lag = 2 * W, j0 = s0 + lag;                                    // joint range of A's time: [j0, s1a]; B's = A's - lag
#if TSTAR_FITPACK_SIMD
    auto steady1 = [&](RowState<W>& st, int i0, int T0, int T1) {
        if (T1 < T0) return;
        if constexpr (W == 8) steady8(st, g, gs, c, i0, T0, T1); else steady4(st, g, gs, c, i0, T0, T1);
    };
    rotate_rows_skewed<W>(sa, g, gs, c, it0, nk1, n8, 0, s0 - 1);           // A fills
    steady1(sa, it0, s0, j0 - 1);                                            // A gets 2 W columns ahead
    rotate_rows_skewed<W>(sb, g, gs, c, itb, nk1, n8, 0, s0 - 1);           // B fills (columns <= itb + s0 - 1 < A's lowest)
    if constexpr (W == 8) steady8x2(sa, sb, g, gs, c, it0, j0, s1a); else steady4x2(sa, sb, g, gs, c, it0, j0, s1a);
    rotate_rows_skewed<W>(sa, g, gs, c, it0, nk1, n8, s1a + 1, tea);        // A drains
    steady1(sb, itb, s1a - lag + 1, s1b);                                    // B's remaining steady steps
    rotate_rows_skewed<W>(sb, g, gs, c, itb, nk1, n8, s1b + 1, teb);        // B drains
#else
    (void)s0; (void)s1a; (void)s1b; (void)tea; (void)teb; (void)lag; (void)j0; (void)sa; (void)sb; (void)g; (void)gs; (void)c; (void)nk1;
#endif
}

template <int W>
inline void rotate_all(const double* b, double pinv, double* g, int gs, double* c, int nk1, int n8) {
    int it = 1;
    if constexpr (W > 1) {
#if TSTAR_FITPACK_SIMD
        // pairs while the joint steady range is long enough to pay for B's separate start (>= 4 W joint steps)
        if (g_pair_batches)
            for (; it + 2 * W - 1 <= n8 && (n8 - it) - (2 * (W - 1) + 2 * W) >= 4 * W; it += 2 * W) rotate_pair<W>(b, pinv, g, gs, c, it, nk1, n8);
#endif
        // a batch pays ~2 W steps of pipeline fill: worth it while the rows are much longer than that
        for (; it + W - 1 <= n8 && nk1 - it > 6 * W; it += W) rotate_batch<W>(b, pinv, g, gs, c, it, nk1, n8);
    }
    if (it <= n8) rotate_rows_seq(b, pinv, g, gs, c, it, n8, nk1, n8);
}

template <int W>
int curfit_impl(const double* x0, const double* y0, int m, double s, double* t_out, double* c_out, int* n_out, double* fp_out, int* iters_out) {
    const int k = 3, k1 = 4, k2 = 5;
    const int nest_max = m + k1;                         // scipy's second pass (UnivariateSpline._reset_nest)
    int nest = nest_max;
    const double tol = 0.001;
    const int maxit = 20;
    const double one = 1.0, con1 = 0.1, con9 = 0.9, con4 = 0.04, half = 0.5;
    const double xb = x0[0], xe = x0[m - 1];
    // 1-based views
    const double* x = x0 - 1;
    const double* y = y0 - 1;
    std::vector<double> tv(nest_max + 2), cv(nest_max + 2 + 2 * PAD), fpint(nest_max + 2), z(nest_max + 2), a((size_t)(nest_max + 2) * 4), b((size_t)(nest_max + 2) * 5), q((size_t)(m + 1) * 4);
    const int gs = nest_max + 2 + 2 * PAD;
    std::vector<double> g((size_t)5 * gs);
    std::vector<int> nrdata(nest_max + 2);
    double* t = tv.data();
    double* c = cv.data() + PAD;                         // c[j] with slack on both sides for the skewed lanes
    auto A = [&](int i, int j) -> double& { return a[(size_t)i * 4 + j - 1]; };
    auto Q = [&](int i, int j) -> double& { return q[(size_t)i * 4 + j - 1]; };
    int ier = 0, n = 0, nplus = 0, nrint = 0, nk1 = 0, npl1 = 0, l = 0;
    double fp = 0.0, fpold = 0.0, fp0 = 0.0, fpms = 0.0, acc = 0.0, p = 0.0;
    double h[8];
    const int nmin = 2 * k1;
    acc = tol * s;
    const int nmax = m + k1;
    int p_iters = 0;
    auto place_interpolation_knots = [&]() {
        const int mk1 = m - k1;
        if (mk1 != 0) {
            const int k3 = k / 2;
            int i = k2, j = k3 + 2;
            if (k3 * 2 != k) {
                for (int ll = 1; ll <= mk1; ++ll) { t[i] = x[j]; ++i; ++j; }
            } else {
                for (int ll = 1; ll <= mk1; ++ll) { t[i] = (x[j] + x[j - 1]) * half; ++i; ++j; }
            }
        }
    };
    // scipy's call sequence: fpcurf0 with nest = (s == 0 ? m + k + 1 : max(m / 2, 2 (k + 1))) (the f2py default), and, when that
    // ends with ier = 1 ("nest too small"), fpcurf1 = the same routine CONTINUING (iopt = 1, ier = 1 on entry) with
    // nest = m + k + 1: it re-reads fp0 / fpold / nplus from fpint(n), fpint(n-1), nrdata(n), recomputes the least-squares
    // spline on the knots found so far and -- because ier is non-zero on entry -- restarts the knot increment at nplus = 1.
    nest = s == 0.0 ? nest_max : (m / 2 > nmin ? m / 2 : nmin);
    if (nest > nest_max) nest = nest_max;
    for (int pass = 0; pass < 2; ++pass) {
    const int iopt = pass;
    bool interpolating = false;
    bool fresh = true;
    if (iopt == 1 && n != nmin) {
        fp0 = fpint[n];
        fpold = fpint[n - 1];
        nplus = nrdata[n];
        if (fp0 > s) fresh = false;
    }
    if (fresh) {
        if (s > 0.0) {
            n = nmin;
            fpold = 0.0;
            nplus = 0;
            nrdata[1] = m - 2;
        } else {
            n = nmax;
            interpolating = true;
        }
    }
    if (interpolating) place_interpolation_knots();
    bool accepted = false;                               // fp <= s reached: go on to part 2
    bool done = false;                                   // finished with the least-squares spline (ier <= 0 or error)
    for (int iter = 1; iter <= m && !done && !accepted; ++iter) {
        if (n == nmin) ier = -2;
        nrint = n - nmin + 1;
        nk1 = n - k1;
        {
            int i = n;
            for (int j = 1; j <= k1; ++j) { t[j] = xb; t[i] = xe; --i; }
        }
        fp = 0.0;
        for (int i = 1; i <= nk1; ++i) {
            z[i] = 0.0;
            for (int j = 1; j <= k1; ++j) A(i, j) = 0.0;
        }
        l = k1;
        for (int it = 1; it <= m; ++it) {
            const double xi = x[it];
            const double wi = 1.0;
            double yi = y[it] * wi;
            while (!(xi < t[l + 1] || l == nk1)) ++l;
            fpbspl3(t, xi, l, h);
            for (int i = 1; i <= k1; ++i) { Q(it, i) = h[i]; h[i] = h[i] * wi; }
            int j = l - k1;
            for (int i = 1; i <= k1; ++i) {
                ++j;
                const double piv = h[i];
                if (piv == 0.0) continue;
                double cs, sn;
                fpgivs(piv, A(j, 1), cs, sn);
                fprota(cs, sn, yi, z[j]);
                if (i == k1) break;
                int i2 = 1;
                for (int i1 = i + 1; i1 <= k1; ++i1) {
                    ++i2;
                    fprota(cs, sn, h[i1], A(j, i2));
                }
            }
            fp = fp + yi * yi;
        }
        if (ier == -2) fp0 = fp;
        fpint[n] = fp0;
        fpint[n - 1] = fpold;
        nrdata[n] = nplus;
        fpback([&](int i, int j) -> double { return A(i, j); }, z.data(), nk1, k1, c);
        fpms = fp - s;
        if (std::fabs(fpms) < acc) { done = true; break; }
        if (fpms < 0.0) { accepted = true; break; }
        if (n == nmax) { ier = -1; done = true; break; }
        if (n == nest) { ier = 1; done = true; break; }
        if (ier == 0) {
            npl1 = nplus * 2;
            const double rn = nplus;
            if (fpold - fp > acc) npl1 = (int)(rn * fpms / (fpold - fp));
            int mx = npl1 > nplus / 2 ? npl1 : nplus / 2;
            if (mx < 1) mx = 1;
            nplus = nplus * 2 < mx ? nplus * 2 : mx;
        } else {
            nplus = 1;
            ier = 0;
        }
        fpold = fp;
        double fpart = 0.0;
        int i = 1;
        l = k2;
        int nw = 0;
        for (int it = 1; it <= m; ++it) {
            if (!(x[it] < t[l] || l > nk1)) { nw = 1; ++l; }
            double term = 0.0;
            int l0 = l - k2;
            for (int j = 1; j <= k1; ++j) { ++l0; term = term + c[l0] * Q(it, j); }
            const double d = 1.0 * (term - y[it]);
            term = d * d;
            fpart = fpart + term;
            if (nw == 0) continue;
            const double store = term * half;
            fpint[i] = fpart - store;
            ++i;
            fpart = store;
            nw = 0;
        }
        fpint[nrint] = fpart;
        bool restart_interp = false;
        for (int ll = 1; ll <= nplus; ++ll) {
            fpknot(x, t, n, fpint.data(), nrdata.data(), nrint, 1);
            if (n == nmax) { restart_interp = true; break; }
            if (n == nest) break;
        }
        if (restart_interp) place_interpolation_knots();
    }
    if (accepted && ier != -2) {
        // part 2: the smoothing spline for the knots found
        fpdisc5(t, n, b.data());
        double p1 = 0.0, f1 = fp0 - s, p3 = -one, f3 = fpms;
        p = 0.0;
        for (int i = 1; i <= nk1; ++i) p = p + A(i, 1);
        double rn = nk1;
        p = rn / p;
        int ich1 = 0, ich3 = 0;
        const int n8 = n - nmin;
        bool ok = false;
        for (int iter = 1; iter <= maxit; ++iter) {
            ++p_iters;
            const double pinv = one / p;
            for (int i = 1; i <= nk1; ++i) {
                c[i] = z[i];
                g[(size_t)4 * gs + PAD + i] = 0.0;
                for (int j = 1; j <= k1; ++j) g[(size_t)(j - 1) * gs + PAD + i] = A(i, j);
            }
            rotate_all<W>(b.data(), pinv, g.data(), gs, c, nk1, n8);
            {
                const double* gp = g.data();
                // c is both right-hand side and result (the library passes the same array twice): z(i) is read before c(i) is written
                fpback([&](int i, int j) -> double { return gp[(size_t)(j - 1) * gs + PAD + i]; }, c, nk1, k2, c);
            }
            fp = 0.0;
            l = k2;
            for (int it = 1; it <= m; ++it) {
                if (!(x[it] < t[l] || l > nk1)) ++l;
                int l0 = l - k2;
                double term = 0.0;
                for (int j = 1; j <= k1; ++j) { ++l0; term = term + c[l0] * Q(it, j); }
                const double d = 1.0 * (term - y[it]);
                fp = fp + d * d;
            }
            fpms = fp - s;
            if (std::fabs(fpms) < acc) { ok = true; break; }
            if (iter == maxit) { ier = 3; ok = true; break; }
            const double p2 = p, f2 = fpms;
            bool next = false;
            if (ich3 == 0) {
                if (!((f2 - f3) > acc)) {
                    p3 = p2; f3 = f2;
                    p = p * con4;
                    if (p <= p1) p = p1 * con9 + p2 * con1;
                    next = true;
                } else if (f2 < 0.0) ich3 = 1;
            }
            if (next) continue;
            if (ich1 == 0) {
                if (!((f1 - f2) > acc)) {
                    p1 = p2; f1 = f2;
                    p = p / con4;
                    if (p3 < 0.0) continue;
                    if (p >= p3) p = p2 * con1 + p3 * con9;
                    continue;
                } else if (f2 > 0.0) ich1 = 1;
            }
            if (f2 >= f1 || f2 <= f3) { ier = 2; ok = true; break; }
            p = fprati(p1, f1, p2, f2, p3, f3);
        }
        (void)ok;
    }
    if (!(ier == 1 && pass == 0 && nest < nest_max)) break;
    nest = nest_max;                                     // second pass: continue with room for every knot
    }
    std::memcpy(t_out, t + 1, sizeof(double) * n);
    std::memcpy(c_out, c + 1, sizeof(double) * n);       // c(n-3..n) hold whatever the library leaves there: zero-initialised here, never read by splev
    *n_out = n;
    *fp_out = fp;
    if (iters_out) *iters_out = p_iters;
    return ier;
}

#if TSTAR_FITPACK_SIMD
__attribute__((target("avx512f,avx512dq"))) int curfit_avx512(const double* x, const double* y, int m, double s, double* t, double* c, int* n, double* fp, int* it) {
    return curfit_impl<8>(x, y, m, s, t, c, n, fp, it);
}
__attribute__((target("avx2"))) int curfit_avx2(const double* x, const double* y, int m, double s, double* t, double* c, int* n, double* fp, int* it) {
    return curfit_impl<4>(x, y, m, s, t, c, n, fp, it);
}
#endif
int curfit_scalar(const double* x, const double* y, int m, double s, double* t, double* c, int* n, double* fp, int* it) {
    return curfit_impl<1>(x, y, m, s, t, c, n, fp, it);
}

}  // namespace

extern "C" {

// x (strictly increasing), y: m > 3 points; s >= 0.  t, c: room for m + 4 doubles each.  Returns FITPACK's ier
// (0, -1, -2 = normal; 1, 2, 3 = the library's warnings: the caller falls back to scipy to raise them as scipy does).
// lanes: 0 = widest the CPU supports, 1 / 4 / 8 = force the sequential / AVX2 / AVX-512 form (tests: all three give the same bits).
int tstar_curfit(const double* x, const double* y, int m, double s, int lanes, double* t, double* c, int* n, double* fp, int* p_iterations) {
    if (!x || !y || !t || !c || !n || !fp || m < 4 || !(s >= 0.0)) return 10;
    for (int i = 1; i < m; ++i) if (!(x[i] > x[i - 1])) return 10;
#if TSTAR_FITPACK_SIMD
    __builtin_cpu_init();
    if (lanes == 0) lanes = (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq")) ? 8 : (__builtin_cpu_supports("avx2") ? 4 : 1);
    if (lanes == 8 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq")) return curfit_avx512(x, y, m, s, t, c, n, fp, p_iterations);
    if (lanes >= 4 && __builtin_cpu_supports("avx2")) return curfit_avx2(x, y, m, s, t, c, n, fp, p_iterations);
#endif
    (void)lanes;
    return curfit_scalar(x, y, m, s, t, c, n, fp, p_iterations);
}

// 1 (default): consecutive batches of the rotation pipeline run their steady states as one loop (two dependent chains in flight);
// 0: one batch at a time (round 4's form).  Same bits either way; process-wide, for tests and timing.
void tstar_curfit_pairing(int on) { g_pair_batches = on != 0; }

int tstar_curfit_lanes(void) {
#if TSTAR_FITPACK_SIMD
    __builtin_cpu_init();
    return (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq")) ? 8 : (__builtin_cpu_supports("avx2") ? 4 : 1);
#else
    return 1;
#endif
}

}  // extern "C"
