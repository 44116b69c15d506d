// Host build of the kernels' per-cell arithmetic (cfd_amd/csrc/sweep_math.h) so that the
// operation order can be checked bit-for-bit against the CPU oracle without a GPU.
// Test scaffolding only: the product library never runs this code on the host.
#include <cstdint>
#include <vector>
#include "../../cfd_amd/csrc/sweep_math.h"

using namespace cfd;

extern "C" void hm_sweep2d(int nx, int ny, double dx, double dy, double dz, double dt, double lambda,
                           double suth, double R, double cv, double k, double dyef,
                           const double* const in[5], const uint8_t* active, double* const out[5]) {
    SweepConsts K = make_consts(dx, dy, dz, dt, lambda, suth, R, cv, k, dyef);
    std::vector<PhaseA2> pa((size_t)nx * ny);
    auto id = [&](int x, int y) { return (size_t)x * ny + y; };
    auto cl = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            double c[5], e[5], w[5], n[5], s[5];
            for (int q = 0; q < 5; ++q) {
                c[q] = in[q][id(x, y)];
                e[q] = in[q][id(cl(x + 1, nx - 1), y)];
                w[q] = in[q][id(cl(x - 1, nx - 1), y)];
                n[q] = in[q][id(x, cl(y + 1, ny - 1))];
                s[q] = in[q][id(x, cl(y - 1, ny - 1))];
            }
            double hx[5];
            x_half_slopes2(c, e, w, hx);
            phaseA2(K, c, hx, n, s, pa[id(x, y)]);
        }
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            double o[5];
            const PhaseA2& C = pa[id(x, y)];
            phaseB2(K, C, pa[id(cl(x + 1, nx - 1), y)].sx, pa[id(cl(x - 1, nx - 1), y)].sx,
                    pa[id(x, cl(y + 1, ny - 1))].sy, pa[id(x, cl(y - 1, ny - 1))].sy, o);
            const bool act = active[id(x, y)] != 0;
            out[F2_RHO][id(x, y)] = o[E2_RHO];
            out[F2_T][id(x, y)] = o[E2_T];
            out[F2_U][id(x, y)] = act ? o[E2_U] : in[F2_U][id(x, y)];
            out[F2_V][id(x, y)] = act ? o[E2_V] : in[F2_V][id(x, y)];
            out[F2_DYE][id(x, y)] = act ? o[E2_DYE] : in[F2_DYE][id(x, y)];
        }
}

extern "C" void hm_sweep3d(int nx, int ny, int nz, double dx, double dy, double dz, double dt, double lambda,
                           double suth, double R, double cv, double k, double dyef,
                           const double* const in[6], const uint8_t* active, double* const out[6]) {
    SweepConsts K = make_consts(dx, dy, dz, dt, lambda, suth, R, cv, k, dyef);
    std::vector<PhaseA3> pa((size_t)nx * ny * nz);
    auto id = [&](int x, int y, int z) { return ((size_t)x * ny + y) * nz + z; };
    auto cl = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                double c[6], nb[6][6];
                const size_t n6[6] = {id(cl(x + 1, nx - 1), y, z), id(cl(x - 1, nx - 1), y, z),
                                      id(x, cl(y + 1, ny - 1), z), id(x, cl(y - 1, ny - 1), z),
                                      id(x, y, cl(z + 1, nz - 1)), id(x, y, cl(z - 1, nz - 1))};
                for (int q = 0; q < 6; ++q) {
                    c[q] = in[q][id(x, y, z)];
                    for (int s = 0; s < 6; ++s) nb[s][q] = in[q][n6[s]];
                }
                phaseA3(K, c, nb, pa[id(x, y, z)]);
            }
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                double o[6];
                const size_t i = id(x, y, z);
                phaseB3(K, pa[i], pa[id(cl(x + 1, nx - 1), y, z)].sx, pa[id(cl(x - 1, nx - 1), y, z)].sx,
                        pa[id(x, cl(y + 1, ny - 1), z)].sy, pa[id(x, cl(y - 1, ny - 1), z)].sy,
                        pa[id(x, y, cl(z + 1, nz - 1))].sz, pa[id(x, y, cl(z - 1, nz - 1))].sz, o);
                const bool act = active[i] != 0;
                out[F3_RHO][i] = o[E3_RHO];
                out[F3_T][i] = o[E3_T];
                out[F3_U][i] = act ? o[E3_U] : in[F3_U][i];
                out[F3_V][i] = act ? o[E3_V] : in[F3_V][i];
                out[F3_W][i] = act ? o[E3_W] : in[F3_W][i];
                out[F3_DYE][i] = act ? o[E3_DYE] : in[F3_DYE][i];
            }
}

extern "C" void hm_sweep2d_fast(int nx, int ny, double dx, double dy, double dz, double dt, double lambda,
                                double suth, double R, double cv, double k, double dyef,
                                const double* const in[5], const uint8_t* active, double* const out[5]) {
    SweepConsts K = make_consts(dx, dy, dz, dt, lambda, suth, R, cv, k, dyef);
    FastConsts F = make_fast_consts(K, dx, dy);
    std::vector<PhaseAF> pa((size_t)nx * ny);
    auto id = [&](int x, int y) { return (size_t)x * ny + y; };
    auto cl = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            double c[5], e[5], w[5], n[5], s[5];
            for (int q = 0; q < 5; ++q) {
                c[q] = in[q][id(x, y)];
                e[q] = in[q][id(cl(x + 1, nx - 1), y)];
                w[q] = in[q][id(cl(x - 1, nx - 1), y)];
                n[q] = in[q][id(x, cl(y + 1, ny - 1))];
                s[q] = in[q][id(x, cl(y - 1, ny - 1))];
            }
            double hx[5];
            x_half_slopes2(c, e, w, hx);
            phaseA2f(F, c, hx, n, s, pa[id(x, y)]);
        }
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            double o[5];
            phaseB2f(F, pa[id(x, y)], pa[id(cl(x + 1, nx - 1), y)].sx, pa[id(cl(x - 1, nx - 1), y)].sx,
                     pa[id(x, cl(y + 1, ny - 1))].sy, pa[id(x, cl(y - 1, ny - 1))].sy, o);
            const bool act = active[id(x, y)] != 0;
            out[F2_RHO][id(x, y)] = o[E2_RHO];
            out[F2_T][id(x, y)] = o[E2_T];
            out[F2_U][id(x, y)] = act ? o[E2_U] : in[F2_U][id(x, y)];
            out[F2_V][id(x, y)] = act ? o[E2_V] : in[F2_V][id(x, y)];
            out[F2_DYE][id(x, y)] = act ? o[E2_DYE] : in[F2_DYE][id(x, y)];
        }
}

extern "C" double hm_half_superbee(double a, double b) { return half_superbee(a, b); }

// Brute-force check of div_by against the IEEE division for n pseudo-random numerators (xorshift,
// magnitudes spread over 2^-60 .. 2^60); returns the number of mismatching bit patterns.
extern "C" long hm_div_by_mismatches(double h, long n, unsigned long long seed) {
    const double rh = 1.0 / h, rl = recip_low(h, rh);
    unsigned long long s = seed ? seed : 88172645463325252ull;
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double mant = 1.0 + (double)(s >> 12) * (1.0 / 4503599627370496.0);   // [1,2), 52 random bits
        const int e = (int)((s >> 3) % 121) - 60;
        double x = ldexp(mant, e);
        if (s & 1) x = -x;
        const double a = x / h, b = div_by(x, h, rh, rl);
        if (__builtin_memcmp(&a, &b, 8) != 0) ++bad;
    }
    return bad;
}

extern "C" void hm_sweep3d_fast(int nx, int ny, int nz, double dx, double dy, double dz, double dt, double lambda,
                                double suth, double R, double cv, double k, double dyef,
                                const double* const in[6], const uint8_t* active, double* const out[6]) {
    SweepConsts K = make_consts(dx, dy, dz, dt, lambda, suth, R, cv, k, dyef);
    FastConsts F = make_fast_consts(K, dx, dy, dz);
    std::vector<PhaseA3F> pa((size_t)nx * ny * nz);
    auto id = [&](int x, int y, int z) { return ((size_t)x * ny + y) * nz + z; };
    auto cl = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                double c[6], nb[6][6];
                const size_t n6[6] = {id(cl(x + 1, nx - 1), y, z), id(cl(x - 1, nx - 1), y, z),
                                      id(x, cl(y + 1, ny - 1), z), id(x, cl(y - 1, ny - 1), z),
                                      id(x, y, cl(z + 1, nz - 1)), id(x, y, cl(z - 1, nz - 1))};
                for (int q = 0; q < 6; ++q) {
                    c[q] = in[q][id(x, y, z)];
                    for (int s = 0; s < 6; ++s) nb[s][q] = in[q][n6[s]];
                }
                phaseA3f(F, c, nb, pa[id(x, y, z)]);
            }
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                double o[6];
                const size_t i = id(x, y, z);
                phaseB3f(F, pa[i], pa[id(cl(x + 1, nx - 1), y, z)].sx, pa[id(cl(x - 1, nx - 1), y, z)].sx,
                         pa[id(x, cl(y + 1, ny - 1), z)].sy, pa[id(x, cl(y - 1, ny - 1), z)].sy,
                         pa[id(x, y, cl(z + 1, nz - 1))].sz, pa[id(x, y, cl(z - 1, nz - 1))].sz, o);
                const bool act = active[i] != 0;
                out[F3_RHO][i] = o[E3_RHO];
                out[F3_T][i] = o[E3_T];
                out[F3_U][i] = act ? o[E3_U] : in[F3_U][i];
                out[F3_V][i] = act ? o[E3_V] : in[F3_V][i];
                out[F3_W][i] = act ? o[E3_W] : in[F3_W][i];
                out[F3_DYE][i] = act ? o[E3_DYE] : in[F3_DYE][i];
            }
}

extern "C" void hm_sweep2d_stream(int nx, int ny, double dx, double dy, double dz, double dt, double lambda,
                                  double suth, double R, double cv, double k, double dyef,
                                  const double* const in[5], const uint8_t* active, double* const out[5]) {
    SweepConsts K = make_consts(dx, dy, dz, dt, lambda, suth, R, cv, k, dyef);
    FastConsts F = make_fast_consts(K, dx, dy);
    std::vector<StreamA2> pa((size_t)nx * ny);
    auto id = [&](int x, int y) { return (size_t)x * ny + y; };
    auto cl = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            double c[5], e[5], w[5], n[5], s[5];
            for (int q = 0; q < 5; ++q) {
                c[q] = in[q][id(x, y)];
                e[q] = in[q][id(cl(x + 1, nx - 1), y)];
                w[q] = in[q][id(cl(x - 1, nx - 1), y)];
                n[q] = in[q][id(x, cl(y + 1, ny - 1))];
                s[q] = in[q][id(x, cl(y - 1, ny - 1))];
            }
            double hx[5];
            x_half_slopes2(c, e, w, hx);
            phaseA2s(F, c, hx, n, s, active[id(x, y)] != 0, pa[id(x, y)]);
        }
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y) {
            const StreamA2& C = pa[id(x, y)];
            const StreamA2& E = pa[id(cl(x + 1, nx - 1), y)];
            const StreamA2& W = pa[id(cl(x - 1, nx - 1), y)];
            const StreamA2& N = pa[id(x, cl(y + 1, ny - 1))];
            const StreamA2& S = pa[id(x, cl(y - 1, ny - 1))];
            double q[4] = {C.q[0], C.q[1], C.q[2], C.q[3]};
            stream_west(q, C.coef, W.mw, W.me[S2_DYE]);
            out[F2_RHO][id(x, y)] = C.rho;
            out[F2_U][id(x, y)] = stream_finish(q[S2_U], C.coef[S2_U], E.me[S2_U], N.ms[S2_U], S.mn[S2_U]);
            out[F2_V][id(x, y)] = stream_finish(q[S2_V], C.coef[S2_V], E.me[S2_V], N.ms[S2_V], S.mn[S2_V]);
            out[F2_T][id(x, y)] = stream_finish(q[S2_T], C.coef[S2_T], E.me[S2_T], N.ms[S2_T], S.ms[S2_T]);
            out[F2_DYE][id(x, y)] = stream_finish(q[S2_DYE], C.coef[S2_DYE], E.me[S2_DYE], N.ms[S2_DYE], S.ms[S2_DYE]);
        }
}

extern "C" void hm_sweep3d_stream(int nx, int ny, int nz, double dx, double dy, double dz, double dt, double lambda,
                                  double suth, double R, double cv, double k, double dyef,
                                  const double* const in[6], const uint8_t* active, double* const out[6]) {
    SweepConsts K = make_consts(dx, dy, dz, dt, lambda, suth, R, cv, k, dyef);
    FastConsts F = make_fast_consts(K, dx, dy, dz);
    std::vector<StreamA3> pa((size_t)nx * ny * nz);
    auto id = [&](int x, int y, int z) { return ((size_t)x * ny + y) * nz + z; };
    auto cl = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                double c[6], nb[6][6], h[3][6];
                const size_t n6[6] = {id(cl(x + 1, nx - 1), y, z), id(cl(x - 1, nx - 1), y, z),
                                      id(x, cl(y + 1, ny - 1), z), id(x, cl(y - 1, ny - 1), z),
                                      id(x, y, cl(z + 1, nz - 1)), id(x, y, cl(z - 1, nz - 1))};
                for (int q = 0; q < 6; ++q) {
                    c[q] = in[q][id(x, y, z)];
                    for (int s = 0; s < 6; ++s) nb[s][q] = in[q][n6[s]];
                }
                half_slopes3(c, nb, h);
                phaseA3s(F, c, h, active[id(x, y, z)] != 0, pa[id(x, y, z)]);
            }
    const int fld[5] = {F3_U, F3_V, F3_W, F3_T, F3_DYE};
    for (int x = 0; x < nx; ++x)
        for (int y = 0; y < ny; ++y)
            for (int z = 0; z < nz; ++z) {
                const size_t i = id(x, y, z);
                const StreamA3& C = pa[i];
                const StreamA3& E = pa[id(cl(x + 1, nx - 1), y, z)];
                const StreamA3& W = pa[id(cl(x - 1, nx - 1), y, z)];
                const StreamA3& N = pa[id(x, cl(y + 1, ny - 1), z)];
                const StreamA3& S = pa[id(x, cl(y - 1, ny - 1), z)];
                const StreamA3& T = pa[id(x, y, cl(z + 1, nz - 1))];
                const StreamA3& B = pa[id(x, y, cl(z - 1, nz - 1))];
                double q[5] = {C.q[0], C.q[1], C.q[2], C.q[3], C.q[4]};
                stream_west3(q, C.coef, W.mw, W.me[S3_DYE]);
                out[F3_RHO][i] = C.rho;
                for (int e = 0; e < 5; ++e) {
                    const double fs = e < 3 ? S.mn[e] : S.ms[e];
                    const double fb = e < 4 ? B.mt[e] : B.mb[e];
                    out[fld[e]][i] = stream_finish3(q[e], C.coef[e], E.me[e], N.ms[e], fs, T.mb[e], fb);
                }
            }
}
