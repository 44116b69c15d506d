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
            phaseA2(K, c, e, w, n, s, pa[id(x, y)]);
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
