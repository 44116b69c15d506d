/*
 * Oracle B -- closed-form CPU restatement of the reference's per-timestep field update.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product library never links or calls it.
 *
 * Parity status: the reference (Haskell) cannot be built or run here.  This file is
 * checked bit-for-bit against oracle_a.py (a literal restatement of the reference's
 * term algebra); the reference itself pins only two helper-level known answers, so for
 * the sweep PARITY IS UNPINNED BY THE REFERENCE.
 *
 * What is restated (all citations are into the reference tree):
 *   sweep          applyDiffEq_adj            src/Main.hs:202-226
 *   equations      continuity .. energy       src/Main.hs:89-175, order src/Main.hs:180-187
 *   time filter    correctPrevTime            src/Main.hs:228-258
 *   step           runSingleStep              src/Main.hs:192-200
 *   reduction      totalDensity               src/Main.hs:269-270
 *   integration    integSurface_adj etc.      src/CalculusUtils.hs:68-118
 *   solve          solveUnknown               src/CalculusUtils.hs:36-52
 *   derivative     approximateDerivative_adj  src/SolutionDomain.hs:506-538
 *   face values    propLimitedSlope_adj       src/SolutionDomain.hs:644-659
 *   derived props  prop_adj (Mew, Pressure)   src/SolutionDomain.hs:569-574
 *   limiter        superbee                   src/SolutionDomain.hs:869-897
 *   neighbours     offsetPosition (clamp)     src/SolutionDomain.hs:434-448
 *
 * The symbolic layer collapses because every term the equations build is one of three
 * shapes (time term, first-order flux pair, second-order pair); each shape is evaluated
 * here with the reference's operand order, and each equation adds its constants in the
 * reference's list order, so results are bit-identical to the literal restatement.
 * A cell's face reconstructions are pure functions of the source slot, so they are
 * tabulated once per sweep (pass 1) and looked up by the term evaluators (pass 2).
 *
 * pow_mode: 0 = t**1.5 through libm pow (what GHC's (**) calls);
 *           1 = t*sqrt(t) (two correctly rounded IEEE operations, reproducible on any
 *               IEEE machine; differs from mode 0 by at most a few ulp of mu).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (see Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { S_E = 0, S_W, S_N, S_S, S_T, S_B, S_C, NSIDE };          /* sideIndex order  */
enum { Q_U = 0, Q_V, Q_W, Q_RHO, Q_T, Q_DYE, Q_MU, Q_P, NQ };   /* tabulated values */
enum { D_X = 0, D_Y, D_Z };

typedef struct {
    int32_t nx, ny, nz;
    double dx, dy, dz, dt;
    double lambda, suth, R, cv, k, dye_f;
    int32_t pow_mode;
    int32_t nthreads;
} ob_params;

typedef struct {
    ob_params p;
    size_t n;
    double *slot[3][6];      /* U V W rho T Dye per time slot */
    double *tmp[6];          /* destination of a non-advancing sweep */
    double *mew;             /* stored Mew: written at init, never read by the update */
    uint8_t *active;
    double *tab;             /* [NQ][NSIDE][n] face table of the source slot */
    int cmt, cat;
} ob_state;

/* ---- geometry ------------------------------------------------------------------- */
static inline size_t cidx(const ob_params *p, int x, int y, int z) {
    return ((size_t)x * p->ny + y) * p->nz + z;
}

/* ---- limiter: src/SolutionDomain.hs:869-897 -------------------------------------- */
static inline double hs_min(double x, double y) { return x <= y ? x : y; }
static inline double hs_max(double x, double y) { return x <= y ? y : x; }

static double superbee(double a, double b) {
    /* maxmod [a,b], minmod [2a,2b], then minmod of the two; any Zero or mixed sign -> 0 */
    int pa = a > 0, pb = b > 0, za = a == 0, zb = b == 0;
    if (za || zb || pa != pb) return 0.0;
    double m1, m2;
    if (pa) { m1 = hs_max(a, b); m2 = hs_min(2 * a, 2 * b); }
    else    { m1 = hs_min(a, b); m2 = hs_max(2 * a, 2 * b); }
    /* outer minmod [m1,m2]; m1 and m2 carry the sign of a */
    if (m1 == 0 || m2 == 0) return 0.0;
    if ((m1 > 0) != (m2 > 0)) return 0.0;
    return (m1 > 0) ? hs_min(m1, m2) : hs_max(m1, m2);
}

static inline double mu_of(const ob_params *p, double t) {
    double t15 = p->pow_mode ? t * sqrt(t) : pow(t, 1.5);
    return (p->lambda * t15) / (t + p->suth);
}

/* ---- pass 1: face table ----------------------------------------------------------- */
static void build_table(ob_state *s, double *const src[6]) {
    const ob_params *p = &s->p;
    const size_t n = s->n;
    const int nx = p->nx, ny = p->ny, nz = p->nz;
#pragma omp parallel for num_threads(p->nthreads) schedule(static)
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                size_t c = cidx(p, x, y, z);
                size_t nb[6];
                nb[S_E] = cidx(p, x + 1 < nx ? x + 1 : x, y, z);
                nb[S_W] = cidx(p, x > 0 ? x - 1 : x, y, z);
                nb[S_N] = cidx(p, x, y + 1 < ny ? y + 1 : y, z);
                nb[S_S] = cidx(p, x, y > 0 ? y - 1 : y, z);
                nb[S_T] = cidx(p, x, y, z + 1 < nz ? z + 1 : z);
                nb[S_B] = cidx(p, x, y, z > 0 ? z - 1 : z);
                for (int q = 0; q < 6; ++q) {
                    const double *f = src[q];
                    double vc = f[c];
                    double *t = s->tab + (size_t)q * NSIDE * n;
                    for (int d = 0; d < 3; ++d) {
                        double up = f[nb[2 * d]], lo = f[nb[2 * d + 1]];
                        double ave = superbee(up - vc, vc - lo);
                        t[(size_t)(2 * d) * n + c] = vc + 0.5 * ave;
                        t[(size_t)(2 * d + 1) * n + c] = vc - 0.5 * ave;
                    }
                    t[(size_t)S_C * n + c] = vc;
                }
                for (int sd = 0; sd < NSIDE; ++sd) {
                    double tf = s->tab[((size_t)Q_T * NSIDE + sd) * n + c];
                    double rf = s->tab[((size_t)Q_RHO * NSIDE + sd) * n + c];
                    s->tab[((size_t)Q_MU * NSIDE + sd) * n + c] = mu_of(p, tf);
                    s->tab[((size_t)Q_P * NSIDE + sd) * n + c] = (p->R * tf) * rf;
                }
            }
}

/* ---- pass 2: term evaluators ------------------------------------------------------ */
typedef struct {
    const ob_state *s;
    size_t c;              /* this cell */
    size_t nb[6];          /* clamped neighbours, sideIndex order */
    double area[6];        /* face areas */
    double hsame[3];       /* average [h, h_nb]                 */
    double hcross[3];      /* 2 * average [h, h_n1, h_n2]       */
    double vol, dt;
} cellctx;

static inline double tabv(const ob_state *s, int q, int side, size_t c) {
    return s->tab[((size_t)q * NSIDE + side) * s->n + c];
}

/* multProps_adj: left product starting from 1 (src/CalculusUtils.hs:231-239) */
static inline double Pi(const ob_state *s, const int *props, int np, size_t c, int side) {
    double acc = 1.0;
    for (int i = 0; i < np; ++i) acc = acc * tabv(s, props[i], side, c);
    return acc;
}

/* first-order term integrated over space then time: two constants (up, lo) */
static inline void flux_pair(const cellctx *k, const int *props, int np, double f, int d, double out[2]) {
    int su = 2 * d, sl = 2 * d + 1;
    out[0] = ((Pi(k->s, props, np, k->c, su) * f) * k->area[su]) * k->dt;
    out[1] = (((Pi(k->s, props, np, k->c, sl) * f) * (-1.0)) * k->area[sl]) * k->dt;
}

/* second-order term d/d(dout)( fout * d(Pi*fin)/d(din) ): two constants (up, lo).
 * The outer multiplier of ddfm_ never reaches the arithmetic (src/CalculusUtils.hs:80,104). */
static inline void dd_pair(const cellctx *k, const int *props, int np, double fin, int din,
                           double fout, int dout, double out[2]) {
    const ob_state *s = k->s;
    for (int half = 0; half < 2; ++half) {
        int side = 2 * dout + half;
        double A = k->area[side];
#define VV(cell, sd) (half == 0 ? ((((Pi(s, props, np, (cell), (sd)) * fin) * fout)) * A) * k->dt \
                                : (((((Pi(s, props, np, (cell), (sd)) * fin) * fout)) * (-1.0)) * A) * k->dt)
        if (din == dout) {
            size_t nbr = k->nb[side];
            double vn = VV(nbr, S_C), vt = VV(k->c, S_C);
            double diff = (half == 0) ? (vn - vt) : (vt - vn);
            out[half] = (diff * 1.0) / k->hsame[din];
        } else {
            size_t n1 = k->nb[2 * din], n2 = k->nb[2 * din + 1];
            double v1 = VV(n1, side), v2 = VV(n2, side);
            out[half] = ((v1 - v2) * 1.0) / k->hcross[din];
        }
#undef VV
    }
}

/* time term: unknown coefficient and constant (src/CalculusUtils.hs:82-90, then * volume) */
static inline void time_term(const cellctx *k, const int *props, int np, double f, int unknown,
                             double *aU, double *cP) {
    double F = Pi(k->s, props, np, k->c, S_C) * f;
    double phi = tabv(k->s, unknown, S_C, k->c);
    double u0 = F * (1.0 / phi);
    if (isnan(u0)) u0 = 1.0;
    *aU = u0 * k->vol;
    *cP = ((F * (-1.0)) * 1.0) * k->vol;
}

#define ADD2(acc, pr) do { (acc) = (acc) + (pr)[0]; (acc) = (acc) + (pr)[1]; } while (0)

static double eq_rho(const cellctx *k) {
    double aU, cP, pr[2], lhs = 0.0, rhs = 0.0;
    const int t[] = { Q_RHO };
    time_term(k, t, 1, 1.0, Q_RHO, &aU, &cP);
    lhs = lhs + 0.0; lhs = lhs + cP;
    for (int d = 0; d < 3; ++d) {
        const int pf[] = { Q_RHO, Q_U + d };
        flux_pair(k, pf, 2, 1.0, d, pr); ADD2(lhs, pr);
    }
    rhs = rhs + 0.0;
    return (rhs - lhs) / (aU - 0.0);
}

static double eq_mom(const cellctx *k, int comp) {      /* comp = 0,1,2 for U,V,W; direction = comp */
    double aU, cP, pr[2], lhs = 0.0, rhs = 0.0;
    const int qc = Q_U + comp;
    const int t[] = { Q_RHO, qc };
    time_term(k, t, 2, 1.0, qc, &aU, &cP);
    lhs = lhs + 0.0; lhs = lhs + cP;
    for (int d = 0; d < 3; ++d) {
        const int pf[] = { Q_RHO, qc, Q_U + d };
        flux_pair(k, pf, 3, 1.0, d, pr); ADD2(lhs, pr);
    }
    { const int pp[] = { Q_P }; flux_pair(k, pp, 1, 1.0, comp, pr); ADD2(lhs, pr); }
    { const int pm[] = { Q_MU, qc };                                        /* divGrad [Mew,comp] 1 */
      for (int d = 0; d < 3; ++d) { dd_pair(k, pm, 2, 1.0, d, 1.0, d, pr); ADD2(rhs, pr); } }
    for (int d = 0; d < 3; ++d) {                                           /* divergence [dmew?_d(comp)] */
        const int pm[] = { Q_MU, Q_U + d };
        dd_pair(k, pm, 2, 1.0, comp, 1.0, d, pr); ADD2(rhs, pr);
    }
    for (int d = 0; d < 3; ++d) {                                           /* -2/3 d/d(comp) of div(mew u) */
        const int pm[] = { Q_MU, Q_U + d };
        dd_pair(k, pm, 2, 1.0, d, -(2.0 / 3.0), comp, pr); ADD2(rhs, pr);
    }
    return (rhs - lhs) / (aU - 0.0);
}

static double eq_dye(const cellctx *k) {
    double aU, cP, pr[2], lhs = 0.0, rhs = 0.0;
    const int t[] = { Q_RHO, Q_DYE };
    time_term(k, t, 2, 1.0, Q_DYE, &aU, &cP);
    lhs = lhs + 0.0; lhs = lhs + cP;
    for (int d = 0; d < 3; ++d) {
        const int pf[] = { Q_RHO, Q_DYE, Q_U + d };
        flux_pair(k, pf, 3, 1.0, d, pr); ADD2(lhs, pr);
    }
    const int pm[] = { Q_MU, Q_DYE };
    for (int d = 0; d < 3; ++d) { dd_pair(k, pm, 2, k->s->p.dye_f, d, 1.0, d, pr); ADD2(rhs, pr); }
    return (rhs - lhs) / (aU - 0.0);
}

static void square_derivative(const cellctx *k, int q, double c, int d, double *rhs) {
    double pr[2];
    const int p4[] = { Q_MU, q, Q_MU, q }, p2[] = { Q_MU, q };
    dd_pair(k, p4, 4, 1.0, d, c / 2.0, d, pr); ADD2(*rhs, pr);
    dd_pair(k, p2, 2, 1.0, d, c * (-1.0), d, pr); ADD2(*rhs, pr);
}

static void paired_mult(const cellctx *k, const int *p1, int n1, const int *p2, int n2, int d1, int d2, double *rhs) {
    double pr[2];
    int p12[8];
    for (int i = 0; i < n1; ++i) p12[i] = p1[i];
    for (int i = 0; i < n2; ++i) p12[n1 + i] = p2[i];
    dd_pair(k, p12, n1 + n2, 1.0, d1, 1.0, d2, pr); ADD2(*rhs, pr);
    dd_pair(k, p1, n1, 1.0, d1, -1.0, d2, pr); ADD2(*rhs, pr);
    dd_pair(k, p2, n2, 1.0, d1, -1.0, d2, pr); ADD2(*rhs, pr);
}

static double eq_energy(const cellctx *k) {
    const ob_params *p = &k->s->p;
    double aU, cP, pr[2], lhs = 0.0, rhs = 0.0;
    const int t[] = { Q_RHO, Q_T };
    time_term(k, t, 2, p->cv, Q_T, &aU, &cP);
    lhs = lhs + 0.0; lhs = lhs + cP;
    for (int d = 0; d < 3; ++d) {
        const int pf[] = { Q_RHO, Q_T, Q_U + d };
        flux_pair(k, pf, 3, p->cv, d, pr); ADD2(lhs, pr);
    }
    for (int d = 0; d < 3; ++d) {
        const int pf[] = { Q_P, Q_U + d };
        flux_pair(k, pf, 2, 1.0, d, pr); ADD2(lhs, pr);
    }
    { const int pt[] = { Q_T };
      for (int d = 0; d < 3; ++d) { dd_pair(k, pt, 1, p->k, d, 1.0, d, pr); ADD2(rhs, pr); } }
    const double c83 = 8.0 / 3.0;
    square_derivative(k, Q_U, c83, D_X, &rhs);
    square_derivative(k, Q_V, c83, D_Y, &rhs);
    square_derivative(k, Q_W, c83, D_Z, &rhs);
    square_derivative(k, Q_U, 1.0, D_Y, &rhs);
    square_derivative(k, Q_V, 1.0, D_X, &rhs);
    square_derivative(k, Q_U, 1.0, D_Z, &rhs);
    square_derivative(k, Q_W, 1.0, D_X, &rhs);
    square_derivative(k, Q_V, 1.0, D_Z, &rhs);
    square_derivative(k, Q_W, 1.0, D_Y, &rhs);
    { const int mu_u[] = { Q_MU, Q_U }, mu_w[] = { Q_MU, Q_W }, v[] = { Q_V }, w[] = { Q_W };
      paired_mult(k, mu_u, 2, v, 1, D_X, D_Y, &rhs);
      paired_mult(k, mu_u, 2, w, 1, D_X, D_Z, &rhs);
      paired_mult(k, mu_w, 2, v, 1, D_Z, D_Y, &rhs); }
    return (rhs - lhs) / (aU - 0.0);
}

/* ---- sweep ------------------------------------------------------------------------ */
static void sweep_into(ob_state *s, double *const src[6], double *const dst[6]) {
    const ob_params *p = &s->p;
    const int nx = p->nx, ny = p->ny, nz = p->nz;
    build_table(s, src);
    double area[6];
    area[S_E] = area[S_W] = (1.0 * p->dy) * p->dz;      /* faceToDirections East = [Y,Z] */
    area[S_N] = area[S_S] = (1.0 * p->dx) * p->dz;      /* North = [X,Z] */
    area[S_T] = area[S_B] = (1.0 * p->dx) * p->dy;      /* Top = [X,Y]   */
    const double len[3] = { p->dx, p->dy, p->dz };
    double hsame[3], hcross[3];
    for (int d = 0; d < 3; ++d) {
        hsame[d] = (0.0 + len[d] / 2.0) + len[d] / 2.0;                               /* average [sl,sln] */
        hcross[d] = 2.0 * (((0.0 + len[d] / 3.0) + len[d] / 3.0) + len[d] / 3.0);     /* 2*average [..3] */
    }
    const double vol = ((1.0 * p->dx) * p->dy) * p->dz;
#pragma omp parallel for num_threads(p->nthreads) schedule(static)
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                cellctx k;
                k.s = s; k.c = cidx(p, x, y, z);
                k.nb[S_E] = cidx(p, x + 1 < nx ? x + 1 : x, y, z);
                k.nb[S_W] = cidx(p, x > 0 ? x - 1 : x, y, z);
                k.nb[S_N] = cidx(p, x, y + 1 < ny ? y + 1 : y, z);
                k.nb[S_S] = cidx(p, x, y > 0 ? y - 1 : y, z);
                k.nb[S_T] = cidx(p, x, y, z + 1 < nz ? z + 1 : z);
                k.nb[S_B] = cidx(p, x, y, z > 0 ? z - 1 : z);
                for (int i = 0; i < 6; ++i) k.area[i] = area[i];
                for (int d = 0; d < 3; ++d) {
                    /* a clamped neighbour is the cell itself: same length, same formula */
                    k.hsame[d] = hsame[d]; k.hcross[d] = hcross[d];
                }
                k.vol = vol; k.dt = p->dt;
                const int act = s->active[k.c];
                dst[3][k.c] = eq_rho(&k);                                     /* alwaysRun */
                dst[0][k.c] = act ? eq_mom(&k, 0) : src[0][k.c];
                dst[1][k.c] = act ? eq_mom(&k, 1) : src[1][k.c];
                dst[2][k.c] = act ? eq_mom(&k, 2) : src[2][k.c];
                dst[4][k.c] = eq_energy(&k);                                  /* alwaysRun */
                dst[5][k.c] = act ? eq_dye(&k) : src[5][k.c];
            }
}

/* ---- public API --------------------------------------------------------------------- */
/* fields7 is in propIndex order: U V W Density Temperature Mew Dye */
ob_state *ob_create(const ob_params *p, const double *const fields7[7], const uint8_t *active) {
    ob_state *s = (ob_state *)calloc(1, sizeof *s);
    s->p = *p;
    if (s->p.nthreads < 1) s->p.nthreads = 1;
    s->n = (size_t)p->nx * p->ny * p->nz;
    static const int map6[6] = { 0, 1, 2, 3, 4, 6 };
    for (int t = 0; t < 3; ++t)
        for (int q = 0; q < 6; ++q) {
            s->slot[t][q] = (double *)malloc(s->n * sizeof(double));
            memcpy(s->slot[t][q], fields7[map6[q]], s->n * sizeof(double));
        }
    for (int q = 0; q < 6; ++q) s->tmp[q] = (double *)malloc(s->n * sizeof(double));
    s->mew = (double *)malloc(s->n * sizeof(double));
    memcpy(s->mew, fields7[5], s->n * sizeof(double));
    s->active = (uint8_t *)malloc(s->n);
    memcpy(s->active, active, s->n);
    s->tab = (double *)malloc((size_t)NQ * NSIDE * s->n * sizeof(double));
    s->cmt = 0; s->cat = 0;
    return s;
}

void ob_destroy(ob_state *s) {
    if (!s) return;
    for (int t = 0; t < 3; ++t) for (int q = 0; q < 6; ++q) free(s->slot[t][q]);
    for (int q = 0; q < 6; ++q) free(s->tmp[q]);
    free(s->mew); free(s->active); free(s->tab); free(s);
}

/* applyDiffEq_adj: the source is always slot cmt; with step_time the result becomes slot cmt+1,
 * otherwise it replaces slot cmt (computed from the snapshot, then swapped in). */
void ob_sweep(ob_state *s, int step_time) {
    if (step_time) {
        int nc = (s->cmt + 1) % 3;
        sweep_into(s, s->slot[s->cmt], s->slot[nc]);
        s->cmt = nc; s->cat += 1;
    } else {
        sweep_into(s, s->slot[s->cmt], s->tmp);
        for (int q = 0; q < 6; ++q) { double *t = s->slot[s->cmt][q]; s->slot[s->cmt][q] = s->tmp[q]; s->tmp[q] = t; }
    }
}

/* correctPrevTime: slot cmt-1 <- pp + 0.5*(c - pp), c = slot cmt, pp = slot cmt-2 */
void ob_time_filter(ob_state *s) {
    if (!(s->cat > 1)) return;
    int tb = (s->cmt + 2) % 3, tc = (tb + 2) % 3;
    static const int always[6] = { 0, 0, 0, 1, 1, 0 };
    for (int q = 0; q < 6; ++q) {
        const double *c = s->slot[s->cmt][q], *pp = s->slot[tc][q];
        double *o = s->slot[tb][q];
#pragma omp parallel for num_threads(s->p.nthreads) schedule(static)
        for (size_t i = 0; i < s->n; ++i)
            if (always[q] || s->active[i]) o[i] = pp[i] + (0.5 * (c[i] - pp[i]));
    }
}

void ob_step(ob_state *s, int nsteps) {
    for (int i = 0; i < nsteps; ++i)
        for (int next = 0; next < 3; ++next) { ob_sweep(s, next == 0); ob_time_filter(s); }
}

/* totalDensity: sequential left fold over slot cmt-1 in node order */
double ob_total_density(const ob_state *s) {
    const double *r = s->slot[(s->cmt + 2) % 3][3];
    double acc = 0.0;
    for (size_t i = 0; i < s->n; ++i) acc = acc + r[i];
    return acc;
}

/* slot_rel: 0 = cmt, 1 = cmt-1, 2 = cmt-2; out7 in propIndex order */
void ob_read_fields(const ob_state *s, int slot_rel, double *const out7[7]) {
    int t = ((s->cmt - slot_rel) % 3 + 3) % 3;
    static const int map6[6] = { 0, 1, 2, 3, 4, 6 };
    for (int q = 0; q < 6; ++q) memcpy(out7[map6[q]], s->slot[t][q], s->n * sizeof(double));
    memcpy(out7[5], s->mew, s->n * sizeof(double));
}

void ob_time(const ob_state *s, int *cmt, int *cat) { *cmt = s->cmt; *cat = s->cat; }

/* Address of one live field (0..5 = U V W Density Temperature Dye) of slot cmt-slot_rel, so that a
 * test harness can overwrite halo planes in place. */
double *ob_slot_ptr(ob_state *s, int slot_rel, int q) {
    int t = ((s->cmt - slot_rel) % 3 + 3) % 3;
    return s->slot[t][q];
}

/* Output property of every cell of slot cmt-slot_rel: `prop_adj Nondirectional prp c Prev` as adjGraphToGrid
 * evaluates it (src/Main.hs:339-346, src/SolutionDomain.hs:540-575).  Side Prev resolves to the position's own
 * slot, and positions come from allPositionsCurrentTime, so the reference reads slot cmt (slot_rel = 0).
 * prop follows the reference's Property enum: 0 Speed, 1 Vorticity, 2 Pressure, 3 U, 4 V, 5 W, 6 Density,
 * 7 Temperature, 8 Mew, 9 Dye.  Vorticity differentiates at the East face ("east is arbitrary here", :561):
 * a derivative along x is the centre difference to the East neighbour (:509-521), one along y or z the
 * difference of the East-face values of the two neighbours in that direction (:523-536). */
static double face_east(const ob_params *p, const double *f, int x, int y, int z) {
    double vc = f[cidx(p, x, y, z)];
    double up = f[cidx(p, x + 1 < p->nx ? x + 1 : x, y, z)], lo = f[cidx(p, x > 0 ? x - 1 : x, y, z)];
    return vc + 0.5 * superbee(up - vc, vc - lo);
}

void ob_property(const ob_state *s, int prop, int slot_rel, double *out) {
    const ob_params *p = &s->p;
    const int nx = p->nx, ny = p->ny, nz = p->nz;
    const int t = ((s->cmt - slot_rel) % 3 + 3) % 3;
    const double *U = s->slot[t][0], *V = s->slot[t][1], *W = s->slot[t][2], *rho = s->slot[t][3],
                 *T = s->slot[t][4], *dye = s->slot[t][5];
    const double len[3] = { p->dx, p->dy, p->dz };
    double hsame[3], hcross[3];
    for (int d = 0; d < 3; ++d) {
        hsame[d] = (0.0 + len[d] / 2.0) + len[d] / 2.0;
        hcross[d] = 2.0 * (((0.0 + len[d] / 3.0) + len[d] / 3.0) + len[d] / 3.0);
    }
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                const size_t c = cidx(p, x, y, z);
                double v;
                switch (prop) {
                case 0: v = sqrt(((0.0 + U[c] * U[c]) + V[c] * V[c]) + W[c] * W[c]); break;
                case 1: {
                    const int xe = x + 1 < nx ? x + 1 : x;
                    const int yn = y + 1 < ny ? y + 1 : y, ys = y > 0 ? y - 1 : y;
                    const int zt = z + 1 < nz ? z + 1 : z, zb = z > 0 ? z - 1 : z;
                    const size_t e = cidx(p, xe, y, z);
                    /* pairs (U,V), (U,W), (V,W): d_dir(a) b - d_dir(b) a */
                    double c1 = ((V[e] - V[c]) * 1.0) / hsame[0]
                              - ((face_east(p, U, x, yn, z) - face_east(p, U, x, ys, z)) * 1.0) / hcross[1];
                    double c2 = ((W[e] - W[c]) * 1.0) / hsame[0]
                              - ((face_east(p, U, x, y, zt) - face_east(p, U, x, y, zb)) * 1.0) / hcross[2];
                    double c3 = ((face_east(p, W, x, yn, z) - face_east(p, W, x, ys, z)) * 1.0) / hcross[1]
                              - ((face_east(p, V, x, y, zt) - face_east(p, V, x, y, zb)) * 1.0) / hcross[2];
                    v = sqrt(((0.0 + c1 * c1) + c2 * c2) + c3 * c3);
                    break;
                }
                case 2: v = (p->R * T[c]) * rho[c]; break;
                case 3: v = U[c]; break;
                case 4: v = V[c]; break;
                case 5: v = W[c]; break;
                case 6: v = rho[c]; break;
                case 7: v = T[c]; break;
                case 8: v = mu_of(p, T[c]); break;
                default: v = dye[c]; break;
                }
                out[c] = v;
            }
}

/* The active mask (one byte per cell), writable: a slab harness fills in its ghost planes. */
uint8_t *ob_active_ptr(ob_state *s) { return s->active; }

/* exposed for unit tests */
double ob_superbee(double a, double b) { return superbee(a, b); }
double ob_mu(const ob_params *p, double t) { return mu_of(p, t); }
