/* TEST INFRASTRUCTURE — CPU restatement (plain C) of the reference's multi-view solver path.
 *
 *      *** parity unpinned ***   
 *
 * Same content as oracle/lfr_ref.py (which is the readable statement and the
 * cross-check of this file), written in C so it runs BASELINE.json's full sizes in
 * seconds and can serve as the timed CPU baseline ("port") of bench.py.
 *
 * Follows, with file:line of /root/reference:
 *   multi-view-refinement/solve.cc:453-481   node/edge creation            (graph_build)
 *   multi-view-refinement/solve.cc:489-549   constrained max spanning forest (build_tracks)
 *   multi-view-refinement/solve.cc:552-582   root selection                (select_roots)
 *   multi-view-refinement/solve.cc:252-373   meta graph + components       (split_components)
 *   multi-view-refinement/solve.cc:79-160    problem assembly + ceres::Solve (solve_component)
 *   multi-view-refinement/cost.cc:13-48      biquadratic interpolator      (interpolate)
 *   multi-view-refinement/cost.cc:78-90      residual functor              (eval_edge)
 * and upstream Ceres Solver (not in the repo, version unpinned: CMakeLists.txt:9) for
 * the trust-region Levenberg-Marquardt loop, loss functions, corrector, bounds
 * projection and the Armijo line search — restated from its published algorithm.
 * COLMAP/Graclus normalized cut (solve.cc:192) cannot be restated: components above
 * the size cap are only accepted with a caller-supplied component assignment.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>
#include <time.h>

/* ---- constants of the numerical contract (solve.cc:89,111,120,147-154 + Ceres defaults) ---- */
#define BOUND 1.0
#define CAUCHY_A 0.25
#define TUKEY_A 0.0625
#define MAX_NUM_ITERATIONS 100
#define MAX_CONSECUTIVE_INVALID 10
#define FUNCTION_TOLERANCE 1e-4
#define GRADIENT_TOLERANCE 1e-8
#define PARAMETER_TOLERANCE 1e-4
#define INITIAL_RADIUS 1e4
#define MAX_RADIUS 1e16
#define MIN_RADIUS 1e-32
#define MIN_RELATIVE_DECREASE 1e-3
#define MIN_LM_DIAGONAL 1e-6
#define MAX_LM_DIAGONAL 1e32
#define LS_SUFFICIENT_DECREASE 1e-4
#define LS_MAX_STEP_CONTRACTION 1e-3
#define LS_MIN_STEP_CONTRACTION 0.6
#define LS_MAX_ITERATIONS 20
#define LS_MIN_STEP_SIZE 1e-9

enum { TERM_CONVERGENCE = 0, TERM_NO_CONVERGENCE = 1, TERM_FAILURE = 2 };
enum { KIND_INTRA = 0, KIND_INTER = 1 };
enum { ERR_OK = 0, ERR_NEEDS_CUT = -2, ERR_NOMEM = -3 };

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------------------------------ */
/* A1: BiquadraticInterpolator::Evaluate, cost.cc:13-48                                          */
/* ------------------------------------------------------------------------------------------ */
static void interpolate(const float *flow, double row, double col, int want_deriv,
                        double f[2], double dfdrow[2], double dfdcol[2]) {
    const double row_ = row, col_ = col;
    row = fmax(fmin(row, 0.5), -0.5);
    col = fmax(fmin(col, 0.5), -0.5);
    const double lr[3] = {2. * row * (row - .5), (-4.) * (row - .5) * (row + .5), 2. * row * (row + 0.5)};
    const double dlr[3] = {2. * row + 2. * (row - .5), (-4.) * (row - .5) + (-4.) * (row + .5),
                           2. * row + 2. * (row + 0.5)};
    const double lc[3] = {2. * col * (col - .5), (-4.) * (col - .5) * (col + .5), 2. * col * (col + 0.5)};
    const double dlc[3] = {2. * col + 2. * (col - .5), (-4.) * (col - .5) + (-4.) * (col + .5),
                           2. * col + 2. * (col + 0.5)};
    for (int k = 0; k < 2; ++k) {
        f[k] = 0.;
        if (want_deriv) { dfdrow[k] = 0.; dfdcol[k] = 0.; }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double d = (double)flow[2 * (i * 3 + j) + k];
                f[k] += lr[i] * lc[j] * d;
                if (want_deriv) {
                    if (row_ == row) dfdrow[k] += dlr[i] * lc[j] * d;
                    if (col_ == col) dfdcol[k] += lr[i] * dlc[j] * d;
                }
            }
    }
}

/* A8: ScaledLoss(CauchyLoss | TukeyLoss, sim) — Ceres loss_function.cc */
static void scaled_loss(int kind, double s, double w, int tukey_variant, double rho[3]) {
    if (kind == KIND_INTRA) {
        const double b = CAUCHY_A * CAUCHY_A, c = 1.0 / b;
        const double sum = 1.0 + s * c, inv = 1.0 / sum;
        rho[0] = b * log(sum);
        rho[1] = fmax(DBL_MIN, inv);
        rho[2] = -c * (inv * inv);
    } else {
        const double a2 = TUKEY_A * TUKEY_A;
        if (s <= a2) {
            const double v = 1.0 - s / a2, v2 = v * v;
            if (tukey_variant == 1) { rho[0] = a2 / 6.0 * (1.0 - v2 * v); rho[1] = 0.5 * v2; rho[2] = -1.0 / a2 * v; }
            else                    { rho[0] = a2 / 3.0 * (1.0 - v2 * v); rho[1] = v2;       rho[2] = -2.0 / a2 * v; }
        } else {
            rho[0] = (tukey_variant == 1) ? a2 / 6.0 : a2 / 3.0;
            rho[1] = 0.0; rho[2] = 0.0;
        }
    }
    rho[0] *= w; rho[1] *= w; rho[2] *= w;
}

/* one kept directed edge of a component */
typedef struct {
    int32_t src, dst;      /* variable-node index in the component, or -1 = constant (root) */
    int32_t kind;
    float sim;
    const float *flow;     /* 18 floats */
} OEdge;

/* A2 + Ceres ResidualBlock::Evaluate + Corrector (simple branch: rho'' <= 0 always) */
static double eval_edge(const OEdge *e, const double x1[2], const double x2[2], int want_jac,
                        int tukey_variant, double r[2], double J1[4], double *j2) {
    double f[2], dr[2], dc[2];
    interpolate(e->flow, x1[0], x1[1], want_jac, f, dr, dc);
    const double r0 = x2[0] - x1[0] - f[0];
    const double r1 = x2[1] - x1[1] - f[1];
    const double s = r0 * r0 + r1 * r1;
    double rho[3];
    scaled_loss(e->kind, s, (double)e->sim, tukey_variant, rho);
    const double sq = sqrt(rho[1]);
    if (want_jac) {
        J1[0] = (-1.0 - dr[0]) * sq; J1[1] = (-dc[0]) * sq;
        J1[2] = (-dr[1]) * sq;       J1[3] = (-1.0 - dc[1]) * sq;
    }
    r[0] = r0 * sq; r[1] = r1 * sq;
    *j2 = sq;
    return 0.5 * rho[0];
}

/* ------------------------------------------------------------------------------------------ */
/* component problem                                                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int nv;               /* variable nodes */
    int ne;               /* kept edges with at least one variable end */
    const OEdge *edges;
    int tukey_variant;
    /* work: per edge corrected jacobian/residual of the last jacobian evaluation */
    double *eJ1, *ej2, *er;
    long n_cost_evals, n_jac_evals;
} Problem;

static inline double clampb(double v) { return fmin(fmax(v, -BOUND), BOUND); }

/* cost (+ optionally residuals/jacobians/gradient) at x */
static double problem_eval(Problem *p, const double *x, int want_jac, double *g /* 2nv or NULL */) {
    static const double zero2[2] = {0.0, 0.0};
    double cost = 0.0;
    if (want_jac) { p->n_jac_evals++; if (g) memset(g, 0, sizeof(double) * 2 * p->nv); }
    else p->n_cost_evals++;
    for (int e = 0; e < p->ne; ++e) {
        const OEdge *ed = &p->edges[e];
        const double *x1 = ed->src >= 0 ? x + 2 * ed->src : zero2;
        const double *x2 = ed->dst >= 0 ? x + 2 * ed->dst : zero2;
        double r[2], J1[4], j2;
        cost += eval_edge(ed, x1, x2, want_jac, p->tukey_variant, r, J1, &j2);
        if (want_jac) {
            memcpy(p->eJ1 + 4 * e, J1, sizeof J1);
            p->ej2[e] = j2;
            p->er[2 * e] = r[0]; p->er[2 * e + 1] = r[1];
            if (g) {
                if (ed->src >= 0) {
                    g[2 * ed->src]     += J1[0] * r[0] + J1[2] * r[1];
                    g[2 * ed->src + 1] += J1[1] * r[0] + J1[3] * r[1];
                }
                if (ed->dst >= 0) {
                    g[2 * ed->dst]     += j2 * r[0];
                    g[2 * ed->dst + 1] += j2 * r[1];
                }
            }
        }
    }
    return cost;
}

/* ---- Ceres polynomial.cc ------------------------------------------------------------------ */
static double ipow(double x, int n) {   /* x^n, n >= 0 (Ceres uses pow()) */
    double v = 1.0;
    for (int i = 0; i < n; ++i) v *= x;
    return v;
}

static double polyval(const double *p, int n, double x) {
    double v = 0.0;
    for (int i = 0; i < n; ++i) v = v * x + p[i];
    return v;
}

/* Real roots of q (degree m = 2..4, coefficients highest first, q[0] != 0) inside [lo, hi]: the critical points of q
 * cut the interval into monotone pieces, a sign change brackets one root, a safeguarded Newton iteration polishes it.
 * Writes exactly m ascending values inside [lo, hi]; a piece without a root contributes its left end (x_min or an
 * inflection point of the interpolant: a harmless extra candidate for the minimum). */
static double horner(const double *q, int m, double x) {
    double v = q[0];
    for (int i = 1; i <= m; ++i) v = v * x + q[i];
    return v;
}
static void real_roots_in(const double *q, int m, double lo, double hi, double *out) {
    if (m == 2) {
        const double A = q[0], B = q[1], C = q[2], D = B * B - 4 * A * C;
        double r0 = lo, r1 = lo;
        if (D >= 0) {
            const double sD = sqrt(D), t = B >= 0 ? -B - sD : -B + sD;
            const double u = t / (2.0 * A), v = t != 0.0 ? (2.0 * C) / t : u;
            r0 = fmin(u, v); r1 = fmax(u, v);
            if (!(r0 == r0)) r0 = lo;
            if (!(r1 == r1)) r1 = lo;
        }
        out[0] = fmin(fmax(r0, lo), hi); out[1] = fmin(fmax(r1, lo), hi);
        return;
    }
    double dq[4], bp[3];
    for (int i = 0; i < m; ++i) dq[i] = q[i] * (m - i);
    real_roots_in(dq, m - 1, lo, hi, bp);
    for (int i = 0; i < m; ++i) {
        double a = i == 0 ? lo : bp[i - 1], b = i == m - 1 ? hi : bp[i];
        const double fa = horner(q, m, a), fb = horner(q, m, b);
        double r = a;
        if (fa != 0.0 && fb == 0.0) r = b;
        else if ((fa < 0.0 && fb > 0.0) || (fa > 0.0 && fb < 0.0)) {
            double x = 0.5 * (a + b);
            for (int it = 0; it < 200; ++it) {
                const double fx = horner(q, m, x);
                if (fx == 0.0) break;
                if ((fx < 0.0) == (fa < 0.0)) a = x; else b = x;
                double xn = x - fx / horner(dq, m - 1, x);
                if (!(xn > a && xn < b)) xn = 0.5 * (a + b);
                if (!(xn > a && xn < b)) break;
                if (fabs(xn - x) <= 2.220446049250313e-16 * fabs(xn)) { x = xn; break; }
                x = xn;
            }
            r = x;
        }
        out[i] = r;
    }
}

/* Candidate abscissae for the minimum of the interpolant over [lo, hi]: Ceres takes the real parts of all roots of the
 * derivative (FindPolynomialRoots: companion-matrix eigenvalues) and keeps those inside the interval.  Degree <= 2 in
 * closed form exactly as Ceres' FindLinear/FindQuadraticPolynomialRoots (a complex pair contributes its real part);
 * degree 3 and 4: the real roots inside the interval - only a real critical point can be the minimum, and unlike a
 * simultaneous complex iteration (the first version of this oracle: Aberth-Ehrlich from a circle of radius
 * 1 + max |a_i / a_0|) the bracketing stays accurate when the leading coefficient is tiny.  Pinned against
 * numpy.roots in tests/test_oracle_kat.py. */
static int poly_root_real_parts(const double *pin, int n, double lo, double hi, double *out) {
    while (n > 0 && pin[0] == 0.0) { ++pin; --n; }
    const int deg = n - 1;
    if (deg <= 0) return 0;
    if (deg == 1) { out[0] = -pin[1] / pin[0]; return 1; }
    if (deg == 2) {
        const double a = pin[0], b = pin[1], c = pin[2];
        const double D = b * b - 4 * a * c, sD = sqrt(fabs(D));
        if (D >= 0) {
            if (b >= 0) { out[0] = (-b - sD) / (2.0 * a); out[1] = (2.0 * c) / (-b - sD); }
            else        { out[0] = (2.0 * c) / (-b + sD); out[1] = (-b + sD) / (2.0 * a); }
        } else { out[0] = -b / (2.0 * a); out[1] = -b / (2.0 * a); }
        return 2;
    }
    real_roots_in(pin, deg, lo, hi, out);
    return deg;
}

typedef struct { double x, value, gradient; int value_valid, gradient_valid; } Sample;

static int solve_dense(double *A, double *b, int n) {   /* partial pivoting; A row-major n x n */
    for (int k = 0; k < n; ++k) {
        int piv = k;
        for (int i = k + 1; i < n; ++i) if (fabs(A[i * n + k]) > fabs(A[piv * n + k])) piv = i;
        if (A[piv * n + k] == 0.0) return -1;
        if (piv != k) {
            for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
            double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        for (int i = k + 1; i < n; ++i) {
            const double f = A[i * n + k] / A[k * n + k];
            for (int j = k; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
            b[i] -= f * b[k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = b[k];
        for (int j = k + 1; j < n; ++j) s -= A[k * n + j] * b[j];
        b[k] = s / A[k * n + k];
    }
    return 0;
}

/* Ceres MinimizeInterpolatingPolynomial (polynomial.cc) */
static double minimize_interpolating_polynomial(const Sample *s, int ns, double x_min, double x_max) {
    int ncons = 0;
    for (int i = 0; i < ns; ++i) ncons += s[i].value_valid + s[i].gradient_valid;
    const int deg = ncons - 1;
    double lhs[36], poly[6];
    memset(lhs, 0, sizeof lhs);
    int row = 0;
    for (int i = 0; i < ns; ++i) {
        if (s[i].value_valid) {
            for (int j = 0; j <= deg; ++j) lhs[row * ncons + j] = ipow(s[i].x, deg - j);
            poly[row++] = s[i].value;
        }
        if (s[i].gradient_valid) {
            for (int j = 0; j < deg; ++j) lhs[row * ncons + j] = (deg - j) * ipow(s[i].x, deg - j - 1);
            poly[row++] = s[i].gradient;
        }
    }
    double best_x = (x_min + x_max) / 2.0;
    if (solve_dense(lhs, poly, ncons) != 0) return best_x;
    double best_v = polyval(poly, ncons, best_x), v;
    v = polyval(poly, ncons, x_min); if (v < best_v) { best_v = v; best_x = x_min; }
    v = polyval(poly, ncons, x_max); if (v < best_v) { best_v = v; best_x = x_max; }
    if (ncons > 2) {
        double deriv[5], roots[4];
        for (int i = 0; i < deg; ++i) deriv[i] = poly[i] * (deg - i);
        const int nr = poly_root_real_parts(deriv, deg, x_min, x_max, roots);
        for (int i = 0; i < nr; ++i) {
            if (roots[i] < x_min || roots[i] > x_max) continue;
            v = polyval(poly, ncons, roots[i]);
            if (v < best_v) { best_v = v; best_x = roots[i]; }
        }
    }
    for (int i = 0; i < ns; ++i) {
        if (s[i].x < x_min || s[i].x > x_max) continue;
        v = polyval(poly, ncons, s[i].x);
        if (v < best_v) { best_v = v; best_x = s[i].x; }
    }
    return best_x;
}

/* TrustRegionMinimizer::DoLineSearch -> ArmijoLineSearch::DoSearch (line_search.cc), CUBIC */
static int armijo_line_search(Problem *p, const double *x, const double *delta, double cost0,
                              double g0_dot_delta, double *xs, double *gs, double *alpha_out, long *n_evals) {
    const int n = 2 * p->nv;
    double dir_max = 0.0;
    for (int i = 0; i < n; ++i) dir_max = fmax(dir_max, fabs(delta[i]));
    Sample initial = {0.0, cost0, g0_dot_delta, 1, 1}, previous = {0, 0, 0, 0, 0}, current;
    double alpha = 1.0;
    int n_iter = 0;
    for (;;) {
        for (int i = 0; i < n; ++i) xs[i] = clampb(x[i] + alpha * delta[i]);
        const double c = problem_eval(p, xs, 1, gs);
        ++*n_evals;
        current.x = alpha; current.value = c; current.value_valid = isfinite(c);
        current.gradient = 0.0; current.gradient_valid = 0;
        if (current.value_valid) {
            double gd = 0.0;
            for (int i = 0; i < n; ++i) gd += delta[i] * gs[i];
            current.gradient = gd; current.gradient_valid = isfinite(gd);
        }
        if (current.value_valid && !(current.value > cost0 + LS_SUFFICIENT_DECREASE * g0_dot_delta * current.x)) {
            *alpha_out = current.x;
            return 1;
        }
        if (++n_iter >= LS_MAX_ITERATIONS) return 0;
        const double lo = LS_MAX_STEP_CONTRACTION * current.x, hi = LS_MIN_STEP_CONTRACTION * current.x;
        double step;
        if (!current.value_valid) step = fmin(fmax(current.x * 0.5, lo), hi);
        else {
            Sample s[3]; int ns = 0;
            s[ns++] = initial; s[ns++] = current;
            if (previous.value_valid) s[ns++] = previous;
            step = minimize_interpolating_polynomial(s, ns, lo, hi);
        }
        if (step * dir_max < LS_MIN_STEP_SIZE) return 0;
        previous = current;
        alpha = step;
    }
}

typedef struct {
    int32_t iterations, termination, n_successful, n_ls_evals;
    int64_t n_cost_evals, n_jac_evals;
    double final_cost, initial_cost;
} CompInfo;

typedef struct {              /* optional per-iteration trace (tests) */
    int cap, n;
    double *rows;             /* n x 8: it, cost, cost_cand, rel, radius, alpha, gmax, flags */
} Trace;

static void trace_push(Trace *t, double it, double cost, double cost_cand, double rel, double radius,
                       double alpha, double gmax, double flags) {
    if (!t || t->n >= t->cap) return;
    double *r = t->rows + 8 * (t->n++);
    r[0] = it; r[1] = cost; r[2] = cost_cand; r[3] = rel; r[4] = radius; r[5] = alpha; r[6] = gmax; r[7] = flags;
}

/* dense lower Cholesky in place (row-major n x n), returns 0 ok / -1 not PD */
static int cholesky(double *A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}

/* Systems above ENVELOPE_MIN_ROWS rows: the reference asks Ceres for SPARSE_NORMAL_CHOLESKY (solve.cc:147).  A dense n^3/3
 * factorization would make this restatement a straw man as a timed CPU baseline for large sparse components, so they are factored
 * inside the ENVELOPE (profile) of a reverse Cuthill-McKee order of the variable nodes: row i keeps columns first[i] .. i, an
 * LL^T without pivoting never fills outside.  Same arithmetic up to the summation order; used for every purpose (parity and timing).
 * set LFRO_DENSE_ONLY=1 in the environment to force the dense path (A/B). */
#define ENVELOPE_MIN_ROWS 193
static int cholesky_profile(double *A, int n, const int *first) {
    for (int i = 0; i < n; ++i) {
        double *Ai = A + (size_t)i * n;
        for (int j = first[i]; j <= i; ++j) {
            const double *Aj = A + (size_t)j * n;
            double s = Ai[j];
            const int k0 = first[i] > first[j] ? first[i] : first[j];
            for (int k = k0; k < j; ++k) s -= Ai[k] * Aj[k];
            if (j < i) Ai[j] = s / Aj[j];
            else { if (!(s > 0.0)) return -1; Ai[i] = sqrt(s); }
        }
    }
    return 0;
}
typedef struct { int deg, node; } DegNode;
static int degnode_cmp(const void *a, const void *b) {
    const DegNode *x = (const DegNode *)a, *y = (const DegNode *)b;
    return x->deg != y->deg ? (x->deg < y->deg ? -1 : 1) : (x->node < y->node ? -1 : (x->node > y->node ? 1 : 0));
}
/* reverse Cuthill-McKee over the variable nodes: pos[node] = position; scratch allocated here (once per large component) */
static void rcm_order(int nv, int ne, const int *esrc, const int *edst, int *pos) {
    int *off = (int *)calloc((size_t)nv + 1, sizeof(int));
    for (int e = 0; e < ne; ++e) if (esrc[e] >= 0 && edst[e] >= 0 && esrc[e] != edst[e]) { ++off[esrc[e] + 1]; ++off[edst[e] + 1]; }
    for (int i = 0; i < nv; ++i) off[i + 1] += off[i];
    int *adj = (int *)malloc(sizeof(int) * (size_t)(off[nv] > 0 ? off[nv] : 1)), *cur = (int *)malloc(sizeof(int) * (size_t)(nv + 1));
    memcpy(cur, off, sizeof(int) * (size_t)nv);
    for (int e = 0; e < ne; ++e) if (esrc[e] >= 0 && edst[e] >= 0 && esrc[e] != edst[e]) { adj[cur[esrc[e]]++] = edst[e]; adj[cur[edst[e]]++] = esrc[e]; }
    int *order = (int *)malloc(sizeof(int) * (size_t)(nv > 0 ? nv : 1)), n_ord = 0;
    char *seen = (char *)calloc((size_t)nv + 1, 1);
    DegNode *nb = (DegNode *)malloc(sizeof(DegNode) * (size_t)(nv > 0 ? nv : 1));
    for (;;) {
        int start = -1;                                     /* the unvisited node of the smallest degree starts the next piece */
        for (int v = 0; v < nv; ++v) if (!seen[v] && (start < 0 || off[v + 1] - off[v] < off[start + 1] - off[start])) start = v;
        if (start < 0) break;
        const int piece0 = n_ord;
        order[n_ord++] = start; seen[start] = 1;
        for (int h = piece0; h < n_ord; ++h) {
            const int v = order[h];
            int m = 0;
            for (int k = off[v]; k < off[v + 1]; ++k) { const int u = adj[k]; if (!seen[u]) { seen[u] = 1; nb[m].deg = off[u + 1] - off[u]; nb[m].node = u; ++m; } }
            qsort(nb, (size_t)m, sizeof(DegNode), degnode_cmp);
            for (int i = 0; i < m; ++i) order[n_ord++] = nb[i].node;
        }
        for (int i = piece0, j = n_ord - 1; i < j; ++i, --j) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    }
    for (int i = 0; i < nv; ++i) pos[order[i]] = i;
    free(off); free(adj); free(cur); free(order); free(seen); free(nb);
}

/* Per-thread scratch that survives from one component to the next (the timed CPU baseline solves ~150 k small components:
 * a calloc/free pair per buffer and component was a visible part of its time).  Slots grow, never shrink. */
enum { TL_BUF = 0, TL_EJ, TL_VIDX, TL_EDGES, TL_X, TL_SLOTS };
static __thread void *tl_ptr[TL_SLOTS];
static __thread size_t tl_cap[TL_SLOTS];
static void *tl_get(int slot, size_t bytes, int zero) {
    if (bytes > tl_cap[slot]) {
        free(tl_ptr[slot]);
        size_t cap = tl_cap[slot] ? tl_cap[slot] : 4096;
        while (cap < bytes) cap *= 2;
        tl_ptr[slot] = malloc(cap);
        tl_cap[slot] = cap;
    }
    if (zero) memset(tl_ptr[slot], 0, bytes);
    return tl_ptr[slot];
}
static void tl_release(void) { for (int i = 0; i < TL_SLOTS; ++i) { free(tl_ptr[i]); tl_ptr[i] = NULL; tl_cap[i] = 0; } }

/* A10: ceres::Solve on one reduced program.  x_out: 2*nv (zeros when the solve FAILS). */
static void solve_problem(Problem *p, double *x_out, CompInfo *info, Trace *tr) {
    const int n = 2 * p->nv;
    memset(info, 0, sizeof *info);
    if (n == 0) return;
    double *buf = (double *)tl_get(TL_BUF, ((size_t)n * n + 12 * (size_t)n) * sizeof(double), 1);
    double *H = buf, *x = H + (size_t)n * n, *g = x + n, *scale = g + n, *diagonal = scale + n, *D = diagonal + n,
           *rhs = D + n, *step = rhs + n, *delta = step + n, *xc = delta + n, *gs = xc + n, *best = gs + n,
           *colsq = best + n;
    p->eJ1 = (double *)tl_get(TL_EJ, sizeof(double) * 7 * (size_t)(p->ne > 0 ? p->ne : 1), 0);
    p->ej2 = p->eJ1 + 4 * (size_t)p->ne;
    p->er = p->ej2 + p->ne;

    /* large systems: envelope order (perm[node] = position; prow(i) = row of variable i in the permuted matrix) */
    int *perm = NULL, *first = NULL;
    double *rhs_p = NULL;
    {
        const char *dense_only = getenv("LFRO_DENSE_ONLY");
        if (n >= ENVELOPE_MIN_ROWS && !(dense_only && dense_only[0] == '1')) {
            perm = (int *)malloc(sizeof(int) * (size_t)p->nv);
            first = (int *)malloc(sizeof(int) * (size_t)n);
            rhs_p = (double *)malloc(sizeof(double) * (size_t)n);
            int *es = (int *)malloc(sizeof(int) * (size_t)(p->ne > 0 ? p->ne : 1)), *ed = (int *)malloc(sizeof(int) * (size_t)(p->ne > 0 ? p->ne : 1));
            for (int e = 0; e < p->ne; ++e) { es[e] = p->edges[e].src; ed[e] = p->edges[e].dst; }
            rcm_order(p->nv, p->ne, es, ed, perm);
            for (int i = 0; i < n; ++i) first[i] = i & ~1;               /* the node's own 2x2 block */
            for (int e = 0; e < p->ne; ++e) {
                if (es[e] < 0 || ed[e] < 0) continue;
                const int pa = perm[es[e]], pb = perm[ed[e]], hi = pa > pb ? pa : pb, lo = pa > pb ? pb : pa;
                if (2 * lo < first[2 * hi]) first[2 * hi] = 2 * lo;
                if (2 * lo < first[2 * hi + 1]) first[2 * hi + 1] = 2 * lo;
            }
            free(es); free(ed);
        }
    }
#define PNODE(a) (perm ? perm[(a)] : (a))
#define PROW(i) (perm ? 2 * perm[(i) >> 1] + ((i) & 1) : (i))
    for (int i = 0; i < n; ++i) x[i] = clampb(0.0);
    double x_norm = 0.0;
    double cost = problem_eval(p, x, 1, g);
    info->initial_cost = cost;
    /* jacobi scaling from the (corrected) jacobian at iteration 0 */
    memset(colsq, 0, sizeof(double) * n);
    for (int e = 0; e < p->ne; ++e) {
        const OEdge *ed = &p->edges[e];
        const double *J1 = p->eJ1 + 4 * e; const double j2 = p->ej2[e];
        if (ed->src >= 0) { colsq[2 * ed->src] += J1[0] * J1[0] + J1[2] * J1[2]; colsq[2 * ed->src + 1] += J1[1] * J1[1] + J1[3] * J1[3]; }
        if (ed->dst >= 0) { colsq[2 * ed->dst] += j2 * j2; colsq[2 * ed->dst + 1] += j2 * j2; }
    }
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + sqrt(colsq[i]));

    double gmax = 0.0;
    for (int i = 0; i < n; ++i) gmax = fmax(gmax, fabs(x[i] - clampb(x[i] - g[i])));
    memcpy(best, x, sizeof(double) * n);
    double min_cost = cost, radius = INITIAL_RADIUS, decrease_factor = 2.0;
    int reuse_diagonal = 0, n_invalid = 0, step_successful = 1, iteration = 0, term = TERM_CONVERGENCE;
    trace_push(tr, 0, cost, cost, 0, radius, 0, gmax, 1);

    for (;;) {
        if (iteration >= MAX_NUM_ITERATIONS) { term = TERM_NO_CONVERGENCE; break; }
        if (step_successful && gmax <= GRADIENT_TOLERANCE) { term = TERM_CONVERGENCE; break; }
        if (radius <= MIN_RADIUS) { term = TERM_CONVERGENCE; break; }
        ++iteration;
        step_successful = 0;

        /* H = Js^T Js (lower), rhs = Js^T r, column norms of Js */
        if (first) { for (int i = 0; i < n; ++i) memset(H + (size_t)i * n + first[i], 0, sizeof(double) * (size_t)(i - first[i] + 1)); }
        else memset(H, 0, sizeof(double) * (size_t)n * n);
        memset(rhs, 0, sizeof(double) * n);
        for (int e = 0; e < p->ne; ++e) {
            const OEdge *ed = &p->edges[e];
            const double *J1 = p->eJ1 + 4 * e, *r = p->er + 2 * e;
            const int a = ed->src, b = ed->dst;
            double A[4] = {0, 0, 0, 0}, bs[2] = {0, 0};     /* scaled blocks */
            if (a >= 0) {
                A[0] = J1[0] * scale[2 * a]; A[1] = J1[1] * scale[2 * a + 1];
                A[2] = J1[2] * scale[2 * a]; A[3] = J1[3] * scale[2 * a + 1];
                const int pa = PNODE(a);
                double *Haa = H + (size_t)(2 * pa) * n + 2 * pa;
                Haa[0] += A[0] * A[0] + A[2] * A[2];
                Haa[n] += A[1] * A[0] + A[3] * A[2];
                Haa[n + 1] += A[1] * A[1] + A[3] * A[3];
                rhs[2 * a] += A[0] * r[0] + A[2] * r[1];
                rhs[2 * a + 1] += A[1] * r[0] + A[3] * r[1];
            }
            if (b >= 0) {
                bs[0] = p->ej2[e] * scale[2 * b]; bs[1] = p->ej2[e] * scale[2 * b + 1];
                const int pb = PNODE(b);
                double *Hbb = H + (size_t)(2 * pb) * n + 2 * pb;
                Hbb[0] += bs[0] * bs[0];
                Hbb[n + 1] += bs[1] * bs[1];
                rhs[2 * b] += bs[0] * r[0];
                rhs[2 * b + 1] += bs[1] * r[1];
            }
            if (a >= 0 && b >= 0) {
                /* cross block (rows of the larger index): B^T A or A^T B */
                const int pa = PNODE(a), pb = PNODE(b);
                if (pb > pa) {
                    double *Hba = H + (size_t)(2 * pb) * n + 2 * pa;
                    Hba[0] += bs[0] * A[0]; Hba[1] += bs[0] * A[1];
                    Hba[n] += bs[1] * A[2]; Hba[n + 1] += bs[1] * A[3];
                } else {
                    double *Hab = H + (size_t)(2 * pa) * n + 2 * pb;
                    Hab[0] += A[0] * bs[0]; Hab[1] += A[2] * bs[1];
                    Hab[n] += A[1] * bs[0]; Hab[n + 1] += A[3] * bs[1];
                }
            }
        }
        if (!reuse_diagonal)
            for (int i = 0; i < n; ++i) { const int r = PROW(i); diagonal[i] = fmin(fmax(H[(size_t)r * n + r], MIN_LM_DIAGONAL), MAX_LM_DIAGONAL); }
        for (int i = 0; i < n; ++i) { const int r = PROW(i); D[i] = sqrt(diagonal[i] / radius); H[(size_t)r * n + r] += D[i] * D[i]; }
        reuse_diagonal = 1;

        int valid = (first ? cholesky_profile(H, n, first) : cholesky(H, n)) == 0;
        double model_cost_change = 0.0;
        if (valid && first) {                      /* the same two triangular solves inside the envelope, in the permuted order */
            for (int i = 0; i < n; ++i) rhs_p[PROW(i)] = rhs[i];
            for (int i = 0; i < n; ++i) {
                double sacc = rhs_p[i];
                for (int k = first[i]; k < i; ++k) sacc -= H[(size_t)i * n + k] * rhs_p[k];
                rhs_p[i] = sacc / H[(size_t)i * n + i];
            }
            for (int i = n - 1; i >= 0; --i) {
                const double y = rhs_p[i] / H[(size_t)i * n + i];
                rhs_p[i] = y;
                for (int k = first[i]; k < i; ++k) rhs_p[k] -= H[(size_t)i * n + k] * y;
            }
            for (int i = 0; i < n; ++i) { step[i] = -rhs_p[PROW(i)]; if (!isfinite(step[i])) valid = 0; }
        } else if (valid) {
            for (int i = 0; i < n; ++i) {          /* L z = rhs */
                double s = rhs[i];
                for (int k = 0; k < i; ++k) s -= H[(size_t)i * n + k] * step[k];
                step[i] = s / H[(size_t)i * n + i];
            }
            for (int i = n - 1; i >= 0; --i) {     /* L^T y = z */
                double s = step[i];
                for (int k = i + 1; k < n; ++k) s -= H[(size_t)k * n + i] * step[k];
                step[i] = s / H[(size_t)i * n + i];
            }
            for (int i = 0; i < n; ++i) { step[i] = -step[i]; if (!isfinite(step[i])) valid = 0; }
        }
        if (valid) {
            /* model_cost_change = -(Js step)^T (r + Js step / 2) */
            for (int e = 0; e < p->ne; ++e) {
                const OEdge *ed = &p->edges[e];
                const double *J1 = p->eJ1 + 4 * e, *r = p->er + 2 * e;
                double m0 = 0.0, m1 = 0.0;
                if (ed->src >= 0) {
                    const double s0 = step[2 * ed->src] * scale[2 * ed->src], s1 = step[2 * ed->src + 1] * scale[2 * ed->src + 1];
                    m0 += J1[0] * s0 + J1[1] * s1; m1 += J1[2] * s0 + J1[3] * s1;
                }
                if (ed->dst >= 0) {
                    m0 += p->ej2[e] * (step[2 * ed->dst] * scale[2 * ed->dst]);
                    m1 += p->ej2[e] * (step[2 * ed->dst + 1] * scale[2 * ed->dst + 1]);
                }
                model_cost_change -= m0 * (r[0] + m0 / 2.0) + m1 * (r[1] + m1 / 2.0);
            }
            valid = model_cost_change > 0.0;
        }
        if (!valid) {
            if (++n_invalid >= MAX_CONSECUTIVE_INVALID) { term = TERM_FAILURE; break; }
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            reuse_diagonal = 1;
            trace_push(tr, iteration, cost, cost, 0, radius, 0, gmax, 4);
            continue;
        }
        n_invalid = 0;
        double g_dot_delta = 0.0;
        for (int i = 0; i < n; ++i) { delta[i] = step[i] * scale[i]; g_dot_delta += g[i] * delta[i]; }

        double alpha = 0.0;
        long ls_evals = 0;
        const int ls_ok = armijo_line_search(p, x, delta, cost, g_dot_delta, xc, gs, &alpha, &ls_evals);
        info->n_ls_evals += (int32_t)ls_evals;
        if (ls_ok) for (int i = 0; i < n; ++i) delta[i] *= alpha;

        for (int i = 0; i < n; ++i) xc[i] = clampb(x[i] + delta[i]);
        double cost_cand = problem_eval(p, xc, 0, NULL);
        if (!isfinite(cost_cand)) cost_cand = DBL_MAX;

        double step_norm = 0.0;
        for (int i = 0; i < n; ++i) step_norm += (x[i] - xc[i]) * (x[i] - xc[i]);
        step_norm = sqrt(step_norm);
        if (step_norm <= PARAMETER_TOLERANCE * (x_norm + PARAMETER_TOLERANCE)) {
            trace_push(tr, iteration, cost, cost_cand, 0, radius, ls_ok ? alpha : -1, gmax, 8);
            term = TERM_CONVERGENCE; break;
        }
        const double cost_change = cost - cost_cand;
        if (fabs(cost_change) <= FUNCTION_TOLERANCE * cost) {
            trace_push(tr, iteration, cost, cost_cand, 0, radius, ls_ok ? alpha : -1, gmax, 16);
            term = TERM_CONVERGENCE; break;
        }
        const double rel = cost_change / model_cost_change;
        if (rel > MIN_RELATIVE_DECREASE) {
            memcpy(x, xc, sizeof(double) * n);
            x_norm = 0.0;
            for (int i = 0; i < n; ++i) x_norm += x[i] * x[i];
            x_norm = sqrt(x_norm);
            cost = problem_eval(p, x, 1, g);
            gmax = 0.0;
            for (int i = 0; i < n; ++i) gmax = fmax(gmax, fabs(x[i] - clampb(x[i] - g[i])));
            step_successful = 1;
            info->n_successful++;
            radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
            radius = fmin(MAX_RADIUS, radius);
            decrease_factor = 2.0;
            reuse_diagonal = 0;
            if (cost < min_cost) { min_cost = cost; memcpy(best, x, sizeof(double) * n); }
        } else {
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            reuse_diagonal = 1;
        }
        trace_push(tr, iteration, cost, cost_cand, rel, radius, ls_ok ? alpha : -1, gmax, step_successful ? 1 : 2);
    }
    info->iterations = iteration;
    info->termination = term;
    info->final_cost = min_cost;
    info->n_cost_evals = p->n_cost_evals;
    info->n_jac_evals = p->n_jac_evals;
    if (term == TERM_FAILURE) memset(x_out, 0, sizeof(double) * n);
    else memcpy(x_out, best, sizeof(double) * n);
    free(perm); free(first); free(rhs_p);
#undef PNODE
#undef PROW
}


/* ------------------------------------------------------------------------------------------ */
/* size cap: recursive_graph_cut (solve.cc:185-250) around a caller-supplied two-way cut          */
/* ------------------------------------------------------------------------------------------ */
/* The reference's primitive is colmap::ComputeNormalizedMinGraphCut(edges, weights, 2) = Graclus (solve.cc:192),
 * which cannot be restated.  A checker may install a substitute (tests install the product's own
 * lfr_bisect_graph): writes the distinct node ids of `edges` and their side (0/1), returns their count.
 * Everything around the primitive - meta edges, integer weights, recursion, orphans, dropping cut edges,
 * BFS re-labelling (solve.cc:311-364) - is restated here independently of the product. */
typedef int64_t (*lfro_bisect_fn)(int64_t n_edges, const int32_t *edge_a, const int32_t *edge_b, const int32_t *weights,
                                  int32_t *nodes, int32_t *part);
static lfro_bisect_fn g_bisect = NULL;
void lfro_set_bisect(lfro_bisect_fn fn) { g_bisect = fn; }

/* final_subset[node] (indexed by meta node = track id) receives base + the subset index local to this call;
 * returns the number of subset indices used.  Mirrors the reference's control flow; its unordered_map
 * iteration orders only permute subset NUMBERS, never the partition (the caller re-labels by BFS). */
static int64_t recursive_graph_cut(int64_t ne, const int32_t *ea, const int32_t *eb, const int32_t *w, const int64_t *node_weights,
                                   int64_t max_subset_weight, int64_t base, int64_t *final_subset) {
    int32_t *nodes = (int32_t *)malloc(sizeof(int32_t) * (2 * ne + 1)), *part = (int32_t *)malloc(sizeof(int32_t) * (2 * ne + 1));
    const int64_t nn = g_bisect(ne, ea, eb, w, nodes, part);            /* solve.cc:192 (substitute) */
    int64_t subset_weights[2] = {0, 0};
    for (int64_t k = 0; k < nn; ++k) { subset_weights[part[k]] += node_weights[nodes[k]]; final_subset[nodes[k]] = -1; }
    int64_t max_subset_idx = 0;
    for (int subset_idx = 0; subset_idx < 2; ++subset_idx) {
        if (subset_weights[subset_idx] <= max_subset_weight) {                          /* solve.cc:205-211 */
            for (int64_t k = 0; k < nn; ++k) if (part[k] == subset_idx) final_subset[nodes[k]] = base + max_subset_idx;
            ++max_subset_idx;
            continue;
        }
        int64_t nsub = 0;
        int32_t *sa = (int32_t *)malloc(sizeof(int32_t) * (ne + 1)), *sb = (int32_t *)malloc(sizeof(int32_t) * (ne + 1)), *sw = (int32_t *)malloc(sizeof(int32_t) * (ne + 1));
        for (int64_t k = 0; k < ne; ++k) {                                              /* solve.cc:213-227 */
            int64_t ia = 0, ib = 0, hi;
            hi = nn - 1; while (ia < hi) { const int64_t m = (ia + hi) / 2; if (nodes[m] < ea[k]) ia = m + 1; else hi = m; }
            hi = nn - 1; while (ib < hi) { const int64_t m = (ib + hi) / 2; if (nodes[m] < eb[k]) ib = m + 1; else hi = m; }
            if (part[ia] == subset_idx && part[ib] == subset_idx) { sa[nsub] = ea[k]; sb[nsub] = eb[k]; sw[nsub] = w[k]; ++nsub; }
        }
        if (nsub > 0)                                                                   /* solve.cc:229-238 */
            max_subset_idx += recursive_graph_cut(nsub, sa, sb, sw, node_weights, max_subset_weight, base + max_subset_idx, final_subset);
        free(sa); free(sb); free(sw);
        for (int64_t k = 0; k < nn; ++k)                                                /* solve.cc:240-246: orphans -> singletons */
            if (part[k] == subset_idx && final_subset[nodes[k]] < 0) final_subset[nodes[k]] = base + max_subset_idx++;
    }
    free(nodes); free(part);
    return max_subset_idx;
}

/* ------------------------------------------------------------------------------------------ */
/* graph stage                                                                                  */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int32_t dst; float sim; const float *flow; } GEdge;

typedef struct lfro {
    int n_images;
    int64_t n_matches, n_nodes;
    int32_t *node_image; uint32_t *node_feat;
    int64_t *out_off; GEdge *out;          /* CSR of directed out-edges in insertion order */
    int64_t *track, *comp; uint8_t *is_root;
    int64_t n_tracks, max_track_size, n_components, max_component_size, n_oversized;
    double *positions;                     /* 2 * n_nodes */
    CompInfo *infos;                       /* per component (zeros for size-1 components) */
    int32_t *comp_nvar, *comp_nedges;
    double graph_ms, cut_ms, solver_ms, total_ms;
    /* solve work */
    int tukey_variant;
    int64_t *comp_off, *comp_nodes;        /* nodes per component, ascending node idx */
    int64_t *order;                        /* components sorted by size descending */
    int64_t next;                          /* work queue cursor (atomic fetch-add) */
    pthread_mutex_t mu;
    int64_t trace_comp; Trace *trace;
} lfro;

/* (image, feature) -> node hash map */
typedef struct { uint64_t *keys; int64_t *vals; uint64_t mask; } NodeMap;
static uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

static int64_t nodemap_get(NodeMap *m, uint64_t key, int64_t *n_nodes) {
    uint64_t h = mix64(key) & m->mask;
    for (;;) {
        if (m->vals[h] < 0) { m->keys[h] = key; m->vals[h] = (*n_nodes)++; return m->vals[h]; }
        if (m->keys[h] == key) return m->vals[h];
        h = (h + 1) & m->mask;
    }
}

typedef struct { double sim; int64_t a, b; } KEdge;
static int kedge_desc(const void *pa, const void *pb) {   /* std::sort ascending + std::reverse */
    const KEdge *x = (const KEdge *)pa, *y = (const KEdge *)pb;
    if (x->sim != y->sim) return x->sim > y->sim ? -1 : 1;
    if (x->a != y->a) return x->a > y->a ? -1 : 1;
    if (x->b != y->b) return x->b > y->b ? -1 : 1;
    return 0;
}
typedef struct { double score; int64_t node; } Score;
static int score_desc(const void *pa, const void *pb) {
    const Score *x = (const Score *)pa, *y = (const Score *)pb;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (x->node != y->node) return x->node > y->node ? -1 : 1;
    return 0;
}
typedef struct { int64_t size, idx; } SizeIdx;
static int sizeidx_desc(const void *pa, const void *pb) {
    const SizeIdx *x = (const SizeIdx *)pa, *y = (const SizeIdx *)pb;
    if (x->size != y->size) return x->size > y->size ? -1 : 1;
    if (x->idx != y->idx) return x->idx > y->idx ? -1 : 1;
    return 0;
}

static int64_t uf_root(int64_t *parent, int64_t i) {
    int64_t r = i;
    while (parent[r] != -1) r = parent[r];
    while (parent[i] != -1) { int64_t nx = parent[i]; parent[i] = r; i = nx; }
    return r;
}

static void solve_one_component(lfro *o, int64_t c) {
    const int64_t lo = o->comp_off[c], hi = o->comp_off[c + 1];
    const int64_t nn = hi - lo;
    if (nn <= 1) return;                                           /* solve.cc:619-622 */
    const int64_t *nodes = o->comp_nodes + lo;
    /* will_be_optimized + variable numbering (roots are constant, solve.cc:131-143) */
    int64_t ne_all = 0;
    for (int64_t k = 0; k < nn; ++k) ne_all += o->out_off[nodes[k] + 1] - o->out_off[nodes[k]];
    int32_t *vidx = (int32_t *)tl_get(TL_VIDX, sizeof(int32_t) * nn, 0);
    OEdge *edges = (OEdge *)tl_get(TL_EDGES, sizeof(OEdge) * (ne_all > 0 ? ne_all : 1), 0);
    int nv = 0;
    for (int64_t k = 0; k < nn; ++k) {
        const int64_t n = nodes[k];
        int opt = 0;
        for (int64_t e = o->out_off[n]; e < o->out_off[n + 1]; ++e) {
            const int64_t d = o->out[e].dst;
            if (o->track[n] == o->track[d] || o->comp[n] == o->comp[d]) opt = 1;
        }
        vidx[k] = (opt && !o->is_root[n]) ? nv++ : -1;
    }
    /* local index of a node of this component: binary search in the sorted node list */
    int ne = 0;
    for (int64_t k = 0; k < nn; ++k) {
        const int64_t n = nodes[k];
        for (int64_t e = o->out_off[n]; e < o->out_off[n + 1]; ++e) {
            const int64_t d = o->out[e].dst;
            int kind;
            if (o->track[n] == o->track[d]) kind = KIND_INTRA;
            else if (o->comp[n] == o->comp[d]) kind = KIND_INTER;
            else continue;
            int64_t a = 0, b = nn - 1;
            while (a < b) { int64_t m = (a + b) / 2; if (nodes[m] < d) a = m + 1; else b = m; }
            const int32_t vs = vidx[k], vd = vidx[a];
            if (vs < 0 && vd < 0) continue;          /* both constant: not in the reduced program */
            edges[ne].src = vs; edges[ne].dst = vd; edges[ne].kind = kind;
            edges[ne].sim = o->out[e].sim; edges[ne].flow = o->out[e].flow;
            ++ne;
        }
    }
    Problem p; memset(&p, 0, sizeof p);
    p.nv = nv; p.ne = ne; p.edges = edges; p.tukey_variant = o->tukey_variant;
    double *x = (double *)tl_get(TL_X, (size_t)(2 * nv + 1) * sizeof(double), 1);
    solve_problem(&p, x, &o->infos[c], (o->trace && o->trace_comp == c) ? o->trace : NULL);
    o->comp_nvar[c] = nv; o->comp_nedges[c] = ne;
    for (int64_t k = 0; k < nn; ++k)
        if (vidx[k] >= 0) { o->positions[2 * nodes[k]] = x[2 * vidx[k]]; o->positions[2 * nodes[k] + 1] = x[2 * vidx[k] + 1]; }
}

static void *worker(void *arg) {
    lfro *o = (lfro *)arg;
    for (;;) {
        const int64_t i = __atomic_fetch_add(&o->next, 1, __ATOMIC_RELAXED);     /* (round 2: a mutex around the cursor) */
        if (i >= o->n_components) break;
        solve_one_component(o, o->order[i]);
    }
    tl_release();
    return NULL;
}

void lfro_free(lfro *o) {
    if (!o) return;
    free(o->node_image); free(o->node_feat); free(o->out_off); free(o->out); free(o->track); free(o->comp);
    free(o->is_root); free(o->positions); free(o->infos); free(o->comp_nvar); free(o->comp_nedges);
    free(o->comp_off); free(o->comp_nodes); free(o->order);
    free(o);
}

/* Build the graph + tracks + roots + components (solve.cc:453-606).
 * match_img1/2: image index per match (already filtered for banned images by the caller).
 * disp1/disp2: M x 18 float32 (zero padded).  comp_override: NULL, or per-node component ids
 * replacing solve.cc:586 (needed when a component exceeds the cap: Graclus is not restatable). */
int lfro_build(int n_images_seen, int64_t n_matches, const int32_t *match_img1, const int32_t *match_img2,
               const uint32_t *feat1, const uint32_t *feat2, const float *sim, const float *disp1,
               const float *disp2, const int64_t *comp_override, lfro **out) {
    lfro *o = (lfro *)calloc(1, sizeof(lfro));
    *out = o;
    o->n_images = n_images_seen; o->n_matches = n_matches;
    pthread_mutex_init(&o->mu, NULL);
    const double t0 = now_ms();
    /* nodes in order of first appearance, node1 before node2 (solve.cc:474-475) */
    NodeMap nm; uint64_t cap = 16; while (cap < (uint64_t)(4 * n_matches + 16)) cap <<= 1;
    nm.mask = cap - 1; nm.keys = (uint64_t *)malloc(cap * 8); nm.vals = (int64_t *)malloc(cap * 8);
    memset(nm.vals, 0xff, cap * 8);
    int64_t *ma = (int64_t *)malloc(sizeof(int64_t) * (n_matches + 1)), *mb = (int64_t *)malloc(sizeof(int64_t) * (n_matches + 1));
    int64_t n_nodes = 0;
    for (int64_t m = 0; m < n_matches; ++m) {
        ma[m] = nodemap_get(&nm, ((uint64_t)(uint32_t)match_img1[m] << 32) | feat1[m], &n_nodes);
        mb[m] = nodemap_get(&nm, ((uint64_t)(uint32_t)match_img2[m] << 32) | feat2[m], &n_nodes);
    }
    o->n_nodes = n_nodes;
    o->node_image = (int32_t *)malloc(sizeof(int32_t) * (n_nodes + 1));
    o->node_feat = (uint32_t *)malloc(sizeof(uint32_t) * (n_nodes + 1));
    for (uint64_t h = 0; h < cap; ++h) if (nm.vals[h] >= 0) {
        o->node_image[nm.vals[h]] = (int32_t)(nm.keys[h] >> 32); o->node_feat[nm.vals[h]] = (uint32_t)nm.keys[h];
    }
    free(nm.keys); free(nm.vals);
    /* out-edge CSR in insertion order (solve.cc:477-478, graph.cc:17-23) */
    o->out_off = (int64_t *)calloc(n_nodes + 2, sizeof(int64_t));
    for (int64_t m = 0; m < n_matches; ++m) { o->out_off[ma[m] + 1]++; o->out_off[mb[m] + 1]++; }
    for (int64_t i = 0; i < n_nodes; ++i) o->out_off[i + 1] += o->out_off[i];
    o->out = (GEdge *)malloc(sizeof(GEdge) * (2 * n_matches + 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (n_nodes + 1));
    memcpy(cur, o->out_off, sizeof(int64_t) * (n_nodes + 1));
    for (int64_t m = 0; m < n_matches; ++m) {
        GEdge *e1 = &o->out[cur[ma[m]]++]; e1->dst = (int32_t)mb[m]; e1->sim = sim[m]; e1->flow = disp2 + 18 * m;
        GEdge *e2 = &o->out[cur[mb[m]]++]; e2->dst = (int32_t)ma[m]; e2->sim = sim[m]; e2->flow = disp1 + 18 * m;
    }
    free(cur);
    o->track = (int64_t *)malloc(sizeof(int64_t) * (n_nodes + 1));
    o->comp = (int64_t *)malloc(sizeof(int64_t) * (n_nodes + 1));
    o->is_root = (uint8_t *)calloc(n_nodes + 1, 1);
    o->positions = (double *)calloc(2 * n_nodes + 2, sizeof(double));
    if (n_nodes == 0) { free(ma); free(mb); return ERR_OK; }

    const double t_start = now_ms();            /* solve.cc:487 */
    /* ---- tracks: constrained maximum spanning forest (solve.cc:489-541) ---- */
    KEdge *ke = (KEdge *)malloc(sizeof(KEdge) * (n_matches + 1));
    for (int64_t m = 0; m < n_matches; ++m) { ke[m].sim = (double)sim[m]; ke[m].a = ma[m]; ke[m].b = mb[m]; }
    free(ma); free(mb);
    qsort(ke, n_matches, sizeof(KEdge), kedge_desc);
    int64_t *parent = (int64_t *)malloc(sizeof(int64_t) * n_nodes);
    int32_t **imgs = (int32_t **)malloc(sizeof(int32_t *) * n_nodes);     /* image set per root */
    int32_t *imgs_n = (int32_t *)malloc(sizeof(int32_t) * n_nodes), *imgs_cap = (int32_t *)malloc(sizeof(int32_t) * n_nodes);
    int64_t *stamp = (int64_t *)calloc(n_images_seen + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n_nodes; ++i) {
        parent[i] = -1; imgs[i] = (int32_t *)malloc(sizeof(int32_t) * 2); imgs[i][0] = o->node_image[i]; imgs_n[i] = 1; imgs_cap[i] = 2;
    }
    for (int64_t k = 0; k < n_matches; ++k) {
        int64_t r1 = uf_root(parent, ke[k].a), r2 = uf_root(parent, ke[k].b);
        if (r1 == r2) continue;
        int conflict = 0;
        for (int32_t i = 0; i < imgs_n[r1]; ++i) stamp[imgs[r1][i]] = k + 1;
        for (int32_t i = 0; i < imgs_n[r2]; ++i) if (stamp[imgs[r2][i]] == k + 1) { conflict = 1; break; }
        if (conflict) continue;
        int64_t big, small;
        if (imgs_n[r1] < imgs_n[r2]) { parent[r1] = r2; big = r2; small = r1; }
        else { parent[r2] = r1; big = r1; small = r2; }
        if (imgs_n[big] + imgs_n[small] > imgs_cap[big]) {
            imgs_cap[big] = 2 * (imgs_n[big] + imgs_n[small]);
            imgs[big] = (int32_t *)realloc(imgs[big], sizeof(int32_t) * imgs_cap[big]);
        }
        memcpy(imgs[big] + imgs_n[big], imgs[small], sizeof(int32_t) * imgs_n[small]);
        imgs_n[big] += imgs_n[small]; imgs_n[small] = 0;
    }
    for (int64_t i = 0; i < n_nodes; ++i) free(imgs[i]);
    free(imgs); free(imgs_n); free(imgs_cap); free(stamp); free(ke);
    int64_t n_tracks = 0;
    for (int64_t i = 0; i < n_nodes; ++i) o->track[i] = -1;
    for (int64_t i = 0; i < n_nodes; ++i) if (parent[i] == -1) o->track[i] = n_tracks++;
    for (int64_t i = 0; i < n_nodes; ++i) if (o->track[i] == -1) o->track[i] = o->track[uf_root(parent, i)];
    free(parent);
    o->n_tracks = n_tracks;
    int64_t *tsize = (int64_t *)calloc(n_tracks + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n_nodes; ++i) tsize[o->track[i]]++;
    for (int64_t t = 0; t < n_tracks; ++t) if (tsize[t] > o->max_track_size) o->max_track_size = tsize[t];
    /* ---- roots (solve.cc:552-582) ---- */
    Score *sc = (Score *)malloc(sizeof(Score) * n_nodes);
    for (int64_t i = 0; i < n_nodes; ++i) {
        double s = 0.;
        for (int64_t e = o->out_off[i]; e < o->out_off[i + 1]; ++e)
            if (o->track[i] == o->track[o->out[e].dst]) s += (double)o->out[e].sim;
        sc[i].score = s; sc[i].node = i;
    }
    qsort(sc, n_nodes, sizeof(Score), score_desc);
    uint8_t *has_root = (uint8_t *)calloc(n_tracks + 1, 1);
    for (int64_t k = 0; k < n_nodes; ++k) {
        const int64_t i = sc[k].node;
        if (has_root[o->track[i]]) continue;
        o->is_root[i] = 1; has_root[o->track[i]] = 1;
    }
    free(sc); free(has_root);
    /* ---- components (solve.cc:252-373) ---- */
    const double t_cut = now_ms();
    int rc = ERR_OK;
    if (comp_override) {
        memcpy(o->comp, comp_override, sizeof(int64_t) * n_nodes);
        int64_t mx = -1; for (int64_t i = 0; i < n_nodes; ++i) if (o->comp[i] > mx) mx = o->comp[i];
        o->n_components = mx + 1;
    } else {
        /* connected components of the track meta-graph: union-find over inter-track edges,
         * labelled in order of first meta node (== the BFS labelling of solve.cc:292-300) */
        int64_t *mp = (int64_t *)malloc(sizeof(int64_t) * n_tracks);
        for (int64_t t = 0; t < n_tracks; ++t) mp[t] = -1;
        for (int64_t i = 0; i < n_nodes; ++i)
            for (int64_t e = o->out_off[i]; e < o->out_off[i + 1]; ++e) {
                const int64_t ta = uf_root(mp, o->track[i]), tb = uf_root(mp, o->track[o->out[e].dst]);
                if (ta != tb) mp[ta > tb ? ta : tb] = ta > tb ? tb : ta;
            }
        int64_t *label = (int64_t *)malloc(sizeof(int64_t) * n_tracks), nc = 0;
        for (int64_t t = 0; t < n_tracks; ++t) label[t] = -1;
        for (int64_t t = 0; t < n_tracks; ++t) { const int64_t r = uf_root(mp, t); if (label[r] < 0) label[r] = nc++; label[t] = label[r]; }
        int64_t *csize = (int64_t *)calloc(nc + 1, sizeof(int64_t));
        for (int64_t t = 0; t < n_tracks; ++t) csize[label[t]] += tsize[t];
        for (int64_t c = 0; c < nc; ++c) if (csize[c] > n_images_seen) o->n_oversized++;      /* solve.cc:314 */
        if (o->n_oversized > 0 && !g_bisect) rc = ERR_NEEDS_CUT;
        int64_t *final_label = label, n_final = nc;
        if (o->n_oversized > 0 && g_bisect) {
            /* meta edges of the oversized components (solve.cc:268-289): per ordered track pair, sum of similarities in
             * node order / out-edge order; collected as (ta, tb, sim) triples, sorted by (ta, tb) with a stable sort */
            typedef struct { int64_t ta, tb; double sim; int64_t seq; } MEdge;
            int64_t nme = 0, cap_me = 1024;
            MEdge *me = (MEdge *)malloc(sizeof(MEdge) * cap_me);
            for (int64_t i = 0; i < n_nodes; ++i) {
                const int64_t ta = o->track[i];
                if (csize[label[ta]] <= n_images_seen) continue;
                for (int64_t e = o->out_off[i]; e < o->out_off[i + 1]; ++e) {
                    const int64_t tb = o->track[o->out[e].dst];
                    if (ta == tb) continue;
                    if (nme == cap_me) { cap_me *= 2; me = (MEdge *)realloc(me, sizeof(MEdge) * cap_me); }
                    me[nme].ta = ta; me[nme].tb = tb; me[nme].sim = (double)o->out[e].sim; me[nme].seq = nme; ++nme;
                }
            }
            /* sort by (ta, tb, seq): the sums below then run in insertion order, as meta_edges[ta][tb] += sim does */
            for (int64_t gap = nme / 2; gap > 0; gap /= 2)            /* shell sort: no comparator context needed, sizes are small */
                for (int64_t i = gap; i < nme; ++i) {
                    MEdge t = me[i]; int64_t j = i;
                    while (j >= gap && (me[j - gap].ta > t.ta || (me[j - gap].ta == t.ta && (me[j - gap].tb > t.tb || (me[j - gap].tb == t.tb && me[j - gap].seq > t.seq))))) { me[j] = me[j - gap]; j -= gap; }
                    me[j] = t;
                }
            /* unique (ta, tb) with summed weight */
            int64_t nu = 0;
            for (int64_t k = 0; k < nme;) {
                int64_t j = k; double sum = 0.;
                while (j < nme && me[j].ta == me[k].ta && me[j].tb == me[k].tb) { sum += me[j].sim; ++j; }
                me[nu].ta = me[k].ta; me[nu].tb = me[k].tb; me[nu].sim = sum; ++nu;
                k = j;
            }
            int64_t *gc = (int64_t *)calloc(n_tracks + 1, sizeof(int64_t)), ngc = 0;
            int64_t *fs = (int64_t *)malloc(sizeof(int64_t) * (n_tracks + 1));
            for (int64_t c = 0; c < nc; ++c) {
                if (csize[c] <= n_images_seen) { for (int64_t t = 0; t < n_tracks; ++t) if (label[t] == c) gc[t] = ngc; ++ngc; continue; }
                int64_t ne = 0;
                for (int64_t k = 0; k < nu; ++k) if (label[me[k].ta] == c && me[k].ta < me[k].tb) ++ne;
                int32_t *ea = (int32_t *)malloc(sizeof(int32_t) * (ne + 1)), *eb = (int32_t *)malloc(sizeof(int32_t) * (ne + 1)), *ew = (int32_t *)malloc(sizeof(int32_t) * (ne + 1));
                ne = 0;
                for (int64_t k = 0; k < nu; ++k) if (label[me[k].ta] == c && me[k].ta < me[k].tb) {     /* solve.cc:325-331 */
                    ea[ne] = (int32_t)me[k].ta; eb[ne] = (int32_t)me[k].tb; ew[ne] = (int)(100 * me[k].sim); ++ne;
                }
                for (int64_t t = 0; t < n_tracks; ++t) if (label[t] == c) fs[t] = -1;
                const int64_t used = recursive_graph_cut(ne, ea, eb, ew, tsize, n_images_seen, 0, fs);
                for (int64_t t = 0; t < n_tracks; ++t) if (label[t] == c) gc[t] = ngc + fs[t];          /* solve.cc:337-341 */
                ngc += used;
                free(ea); free(eb); free(ew);
            }
            /* drop cut meta edges, re-split (solve.cc:345-364): union-find over the kept meta edges, labelled in order of
             * first meta node (== BFS labelling) */
            int64_t *mp2 = (int64_t *)malloc(sizeof(int64_t) * (n_tracks + 1));
            for (int64_t t = 0; t < n_tracks; ++t) mp2[t] = -1;
            for (int64_t i = 0; i < n_nodes; ++i)
                for (int64_t e = o->out_off[i]; e < o->out_off[i + 1]; ++e) {
                    const int64_t ta = o->track[i], tb = o->track[o->out[e].dst];
                    if (ta == tb || gc[ta] != gc[tb]) continue;
                    const int64_t ra = uf_root(mp2, ta), rb = uf_root(mp2, tb);
                    if (ra != rb) mp2[ra > rb ? ra : rb] = ra > rb ? rb : ra;
                }
            final_label = (int64_t *)malloc(sizeof(int64_t) * (n_tracks + 1));
            for (int64_t t = 0; t < n_tracks; ++t) final_label[t] = -1;
            n_final = 0;
            for (int64_t t = 0; t < n_tracks; ++t) { const int64_t r = uf_root(mp2, t); if (final_label[r] < 0) final_label[r] = n_final++; final_label[t] = final_label[r]; }
            free(me); free(gc); free(fs); free(mp2);
        }
        for (int64_t i = 0; i < n_nodes; ++i) o->comp[i] = final_label[o->track[i]];
        o->n_components = n_final;
        if (final_label != label) free(final_label);
        free(mp); free(label); free(csize);
    }
    free(tsize);
    o->cut_ms = now_ms() - t_cut;
    /* nodes per component (solve.cc:594-604) */
    const int64_t nc = o->n_components;
    o->comp_off = (int64_t *)calloc(nc + 2, sizeof(int64_t));
    for (int64_t i = 0; i < n_nodes; ++i) o->comp_off[o->comp[i] + 1]++;
    for (int64_t c = 0; c < nc; ++c) { if (o->comp_off[c + 1] > o->max_component_size) o->max_component_size = o->comp_off[c + 1]; o->comp_off[c + 1] += o->comp_off[c]; }
    o->comp_nodes = (int64_t *)malloc(sizeof(int64_t) * (n_nodes + 1));
    int64_t *cc = (int64_t *)malloc(sizeof(int64_t) * (nc + 1));
    memcpy(cc, o->comp_off, sizeof(int64_t) * (nc + 1));
    for (int64_t i = 0; i < n_nodes; ++i) o->comp_nodes[cc[o->comp[i]]++] = i;
    free(cc);
    SizeIdx *si = (SizeIdx *)malloc(sizeof(SizeIdx) * (nc + 1));
    for (int64_t c = 0; c < nc; ++c) { si[c].size = o->comp_off[c + 1] - o->comp_off[c]; si[c].idx = c; }
    qsort(si, nc, sizeof(SizeIdx), sizeidx_desc);
    o->order = (int64_t *)malloc(sizeof(int64_t) * (nc + 1));
    for (int64_t c = 0; c < nc; ++c) o->order[c] = si[c].idx;
    free(si);
    o->infos = (CompInfo *)calloc(nc + 1, sizeof(CompInfo));
    o->comp_nvar = (int32_t *)calloc(nc + 1, sizeof(int32_t));
    o->comp_nedges = (int32_t *)calloc(nc + 1, sizeof(int32_t));
    o->graph_ms = now_ms() - t_start;
    (void)t0;
    return rc;
}

/* solve.cc:614-635: thread pool over components, largest first */
int lfro_solve(lfro *o, int n_threads, int tukey_variant, int64_t trace_comp, double *trace_rows, int trace_cap, int *trace_n) {
    if (o->n_nodes == 0) return ERR_OK;
    Trace tr; tr.cap = trace_cap; tr.n = 0; tr.rows = trace_rows;
    o->trace = trace_rows ? &tr : NULL; o->trace_comp = trace_comp;
    o->tukey_variant = tukey_variant;
    o->next = 0;
    memset(o->positions, 0, sizeof(double) * 2 * o->n_nodes);
    const double t1 = now_ms();
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, worker, o);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
    o->solver_ms = now_ms() - t1;
    o->total_ms = o->graph_ms + o->solver_ms;
    if (trace_n) *trace_n = tr.n;
    o->trace = NULL;
    return ERR_OK;
}

/* accessors */
int64_t lfro_n_nodes(const lfro *o) { return o->n_nodes; }
int64_t lfro_n_tracks(const lfro *o) { return o->n_tracks; }
int64_t lfro_max_track_size(const lfro *o) { return o->max_track_size; }
int64_t lfro_n_components(const lfro *o) { return o->n_components; }
int64_t lfro_max_component_size(const lfro *o) { return o->max_component_size; }
int64_t lfro_n_oversized(const lfro *o) { return o->n_oversized; }
double lfro_graph_ms(const lfro *o) { return o->graph_ms; }
double lfro_solver_ms(const lfro *o) { return o->solver_ms; }
const int32_t *lfro_node_image(const lfro *o) { return o->node_image; }
const uint32_t *lfro_node_feat(const lfro *o) { return o->node_feat; }
const int64_t *lfro_track(const lfro *o) { return o->track; }
const int64_t *lfro_comp(const lfro *o) { return o->comp; }
const uint8_t *lfro_is_root(const lfro *o) { return o->is_root; }
const double *lfro_positions(const lfro *o) { return o->positions; }
const int32_t *lfro_comp_nvar(const lfro *o) { return o->comp_nvar; }
const int32_t *lfro_comp_nedges(const lfro *o) { return o->comp_nedges; }
int lfro_info_size(void) { return (int)sizeof(CompInfo); }
const void *lfro_infos(const lfro *o) { return o->infos; }

/* unit-level entry points for known-answer tests */
void lfro_interpolate(const float *flow, double row, double col, double *out6) {
    interpolate(flow, row, col, 1, out6, out6 + 2, out6 + 4);
}
void lfro_loss(int kind, double s, double w, int tukey_variant, double *rho3) { scaled_loss(kind, s, w, tukey_variant, rho3); }
double lfro_eval_edge(const float *flow, float sim, int kind, const double *x1, const double *x2, int tukey_variant, double *out7) {
    OEdge e; e.src = 0; e.dst = 1; e.kind = kind; e.sim = sim; e.flow = flow;
    return eval_edge(&e, x1, x2, 1, tukey_variant, out7, out7 + 2, out7 + 6);
}
double lfro_minimize_poly(const double *samples /* ns x 5 */, int ns, double x_min, double x_max) {
    Sample s[3];
    for (int i = 0; i < ns && i < 3; ++i) { s[i].x = samples[5 * i]; s[i].value = samples[5 * i + 1]; s[i].gradient = samples[5 * i + 2];
        s[i].value_valid = samples[5 * i + 3] != 0; s[i].gradient_valid = samples[5 * i + 4] != 0; }
    return minimize_interpolating_polynomial(s, ns, x_min, x_max);
}
