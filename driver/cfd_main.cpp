// Compiled host for libcfdhip.so: the reference program's `main` (src/Main.hs:354-356 ->
// runTimeSteps_Print, :278-306) with its solve loop bound to the C ABI, the way the patched Haskell
// host binds it.  Plain C++ (no HIP on the host side); links only against include/cfd_hip.h's
// entry points.
//
//   cfd_main [steps] [--fast] [--nx N --ny N] [--string-domain PROPERTY] [--gpus N | --slabs N]
//
// --gpus N: the grid cut into N x-slabs, one per GPU, halo planes through RCCL (cfd_multi_create with
// CFD_TRANSPORT_RCCL); --slabs N: N slabs on the current GPU, halo planes copied (the same decomposition on a
// one-GPU machine).  The loop and its five lines stay those of the reference; only the density sum's order
// of summation differs from the undecomposed run.
//
// --string-domain k also prints, after every step, the slice of Property k (0 Speed .. 9 Dye) the way the
// reference's stringDomain does (src/Main.hs:349-352; the call is commented out at :302).
//
// Builds the shipped configuration (initialGrid_adj, src/SolutionDomain.hs:262-353, in closed form;
// 301 x 151 x 1 unless --nx/--ny scale it) and prints the reference's five lines per step.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/cfd_hip.h"

// Haskell `show` of a Double (Numeric.showFloat): shortest digits that round-trip; positional
// notation for 0.1 <= |v| < 10^7, otherwise d.ddde<n>.
static std::string hs_show_double(double v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    if (v == 0.0) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    const double a = std::fabs(v);
    if (a >= 0.1 && a < 1e7) {
        auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
        std::string s(buf, r.ptr);
        if (s.find('.') == std::string::npos) s += ".0";
        return s;
    }
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    std::string s(buf, r.ptr);              // d[.ddd]e[+-]XX
    const size_t e = s.find('e');
    std::string mant = s.substr(0, e);
    if (mant.find('.') == std::string::npos) mant += ".0";
    return mant + "e" + std::to_string(std::atoi(s.c_str() + e + 1));
}

// Haskell `round`: to nearest, ties to even
static int hs_round(double x) { return (int)std::nearbyint(x); }

static int die(const char* what) {
    std::fprintf(stderr, "%s: %s\n", what, cfd_last_error());
    return 1;
}

int main(int argc, char** argv) {
    long steps = 5;
    int nx = 301, ny = 151;
    uint32_t flags = 0;
    int show_prop = -1;
    int n_slabs = 0, transport = CFD_TRANSPORT_COPY;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--fast")) flags |= CFD_FLAG_FAST_MATH;
        else if (!std::strcmp(argv[i], "--nx") && i + 1 < argc) nx = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--ny") && i + 1 < argc) ny = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--string-domain") && i + 1 < argc) show_prop = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) { n_slabs = std::atoi(argv[++i]); transport = CFD_TRANSPORT_RCCL; }
        else if (!std::strcmp(argv[i], "--slabs") && i + 1 < argc) { n_slabs = std::atoi(argv[++i]); transport = CFD_TRANSPORT_COPY; }
        else steps = std::atol(argv[i]);
    }
    cfd_params p;
    if (cfd_default_params(&p) != CFD_OK) return die("cfd_default_params");
    // the reference's cell size stays; only the cell counts scale
    const double gx = 1.6, gy = 0.8, dx = gx / 301.0, dy = gy / 151.0;
    p.nx = nx; p.ny = ny; p.nz = 1;
    p.dx = dx; p.dy = dy; p.dz = 1.0;
    p.flags = flags;
    const size_t n = (size_t)nx * ny;
    std::vector<double> U(n, 10.0), V(n, 0.0), W(n, 0.0), rho(n, 0.3), T(n, 223.0), mew(n, 0.1), dye(n, 0.0);
    std::vector<uint8_t> active(n, 1);
    // obstacle corners through coordToPos (src/SolutionDomain.hs:160-166, :214-224), scaled by index fraction
    int x0 = hs_round((gx / 4) / dx), x1 = hs_round((gx / 4 + 0.1) / dx);
    int y0 = hs_round((gy / 2 - 0.05) / dy), y1 = hs_round((gy / 2 + 0.05) / dy);
    if (nx != 301 || ny != 151) {
        x0 = (int)(75.0 / 301 * nx); x1 = (int)(94.0 / 301 * nx);
        y0 = (int)(66.0 / 151 * ny); y1 = (int)(85.0 / 151 * ny);
    }
    for (int y = 0; y < ny; ++y) active[(size_t)0 * ny + y] = 0;                 // inflow column x = 0
    for (int x = x0; x <= x1 && x < nx; ++x)
        for (int y = y0; y <= y1 && y < ny; ++y) {
            const size_t i = (size_t)x * ny + y;
            active[i] = 0; U[i] = V[i] = W[i] = 0.0;
        }
    dye[(size_t)0 * ny + (ny - 1) / 2] = 1.0;
    const double* fields[CFD_NPROP] = {U.data(), V.data(), W.data(), rho.data(), T.data(), mew.data(), dye.data()};
    if (n_slabs > 1) {
        // the same loop over the slab decomposition: runSingleStep -> cfd_multi_step, totalDensity -> cfd_multi_total_density
        cfd_multi* m = nullptr;
        std::vector<int32_t> devs(n_slabs);
        for (int i = 0; i < n_slabs; ++i) devs[i] = transport == CFD_TRANSPORT_RCCL ? i : 0;
        if (cfd_multi_create(&p, n_slabs, devs.data(), transport, fields, active.data(), &m) != CFD_OK) return die("cfd_multi_create");
        std::printf("starting ..... \n");
        for (long step = 0; step < steps; ++step) {
            if (cfd_multi_step(m, 1) != CFD_OK) return die("cfd_multi_step");
            int cmt = 0, cat = 0;
            double total = 0.0;
            if (cfd_multi_time(m, &cmt, &cat) != CFD_OK) return die("cfd_multi_time");
            if (cfd_multi_total_density(m, &total) != CFD_OK) return die("cfd_multi_total_density");
            std::printf("%zu\nstep: %ld\ntimeLevel: %d\ntimeLevelAbsolute: %d\ntotalDensity: %s\n \n", n, step, cmt, cat,
                        hs_show_double(total).c_str());
        }
        cfd_multi_destroy(m);
        return 0;
    }
    cfd_handle* h = nullptr;
    if (cfd_create(&p, fields, active.data(), &h) != CFD_OK) return die("cfd_create");
    std::printf("starting ..... \n");
    for (long step = 0; step < steps; ++step) {
        if (cfd_step(h, 1) != CFD_OK) return die("cfd_step");
        int cmt = 0, cat = 0;
        double total = 0.0;
        if (cfd_time(h, &cmt, &cat) != CFD_OK) return die("cfd_time");
        if (cfd_total_density(h, &total) != CFD_OK) return die("cfd_total_density");
        std::printf("%zu\nstep: %ld\ntimeLevel: %d\ntimeLevelAbsolute: %d\ntotalDensity: %s\n \n", n, step, cmt, cat,
                    hs_show_double(total).c_str());
        if (show_prop >= 0) {
            // adjGraphToGrid next prp (1 + maxPos Y) (quot (maxPos Z) 2), one text row per x
            std::vector<double> slice(n);
            if (cfd_property_slice(h, show_prop, (p.nz - 1) / 2, 0, slice.data()) != CFD_OK)
                return die("cfd_property_slice");
            std::string text;
            for (int x = 0; x < nx; ++x) {
                text += "\n";
                for (int y = 0; y < ny; ++y) text += " " + hs_show_double(slice[(size_t)x * ny + y]);
            }
            std::printf("%s\n", text.c_str());
        }
    }
    cfd_destroy(h);
    return 0;
}
