// Compiled host for libcfdhip.so: the reference program's `main` (src/Main.hs:354-356 ->
// runTimeSteps_Print, :278-306) with its solve loop bound to the C ABI, the way the patched Haskell
// host binds it.  Plain C++ (no HIP on the host side); links only against include/cfd_hip.h's
// entry points.
//
//   cfd_main [steps] [--fast] [--nx N --ny N] [--string-domain PROPERTY] [--gpus N | --slabs N] [--no-preflight]
//
// --gpus N: the grid cut into N x-slabs, one per GPU, halo planes through RCCL (cfd_multi_create with
// CFD_TRANSPORT_RCCL); --slabs N: N slabs on the current GPU, halo planes copied (the same decomposition on a
// one-GPU machine).  The loop and its five lines stay those of the reference; only the density sum's order
// of summation differs from the undecomposed run.  Before the run proper a decomposed run checks itself
// (pre-flight): a small perturbed grid with walls across the slab boundaries is stepped three times undecomposed and
// through the same slab driver, and the three time slots must agree bit for bit (the density sum to 1e-12); a
// difference ends the program with status 4 and names slot, property and plane, a stall with SIGALRM after 180 s.
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

#include <unistd.h>

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

// The decomposition against the undecomposed run on a small grid: 0 = identical, 4 = different, 1 = a call failed.
static int preflight(const cfd_params& shipped, int n_slabs, const int32_t* devs, int transport) {
    cfd_params p = shipped;
    p.nx = 24 * n_slabs; p.ny = 200; p.dt = 1e-7;
    const int nx = p.nx, ny = p.ny;
    const size_t n = (size_t)nx * ny;
    std::vector<double> f[CFD_NPROP];
    for (auto& v : f) v.assign(n, 0.0);
    std::vector<uint8_t> active(n, 1);
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto u = [&]() {        // uniform in (-1, 1), a fixed sequence
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(state >> 11) / 9007199254740992.0 * 2.0 - 1.0;
    };
    for (size_t i = 0; i < n; ++i) {
        f[0][i] = 10.0 + 0.1 * u(); f[1][i] = 0.1 * u(); f[3][i] = 0.3 * (1.0 + 1e-3 * u());
        f[4][i] = 223.0 * (1.0 + 1e-3 * u()); f[5][i] = 0.1; f[6][i] = (i % 97 == 0) ? 1.0 : 0.0;
    }
    for (int y = 0; y < ny; ++y) active[y] = 0;                                     // inflow column
    for (int s = 1; s < n_slabs; ++s)                                               // walls across every slab boundary
        for (int x = nx * s / n_slabs - 2; x < nx * s / n_slabs + 2; ++x)
            for (int y = 20 * s; y < 20 * s + 30 && y < ny; ++y) {
                const size_t i = (size_t)x * ny + y;
                active[i] = 0; f[0][i] = f[1][i] = 0.0;
            }
    const double* fields[CFD_NPROP] = {f[0].data(), f[1].data(), f[2].data(), f[3].data(), f[4].data(), f[5].data(), f[6].data()};
    cfd_handle* h = nullptr;
    cfd_multi* m = nullptr;
    p.device = devs[0];
    if (cfd_create(&p, fields, active.data(), &h) != CFD_OK) return die("pre-flight: cfd_create");
    if (cfd_multi_create(&p, n_slabs, devs, transport, fields, active.data(), &m) != CFD_OK) return die("pre-flight: cfd_multi_create");
    std::vector<double> a[CFD_NPROP], b[CFD_NPROP];
    double* pa[CFD_NPROP]; double* pb[CFD_NPROP];
    for (int k = 0; k < CFD_NPROP; ++k) { a[k].resize(n); b[k].resize(n); pa[k] = a[k].data(); pb[k] = b[k].data(); }
    static const char* names[CFD_NPROP] = {"U", "V", "W", "Density", "Temperature", "Mew", "Dye"};
    int verdict = 0;
    for (int step = 1; step <= 3 && !verdict; ++step) {
        if (cfd_step(h, 1) != CFD_OK) return die("pre-flight: cfd_step");
        if (cfd_multi_step(m, 1) != CFD_OK) return die("pre-flight: cfd_multi_step");
        for (int rel = 0; rel < 3 && !verdict; ++rel) {
            if (cfd_read_fields(h, rel, pa) != CFD_OK) return die("pre-flight: cfd_read_fields");
            if (cfd_multi_read_fields(m, rel, pb) != CFD_OK) return die("pre-flight: cfd_multi_read_fields");
            for (int k = 0; k < CFD_NPROP && !verdict; ++k)
                for (size_t i = 0; i < n; ++i)
                    if (std::memcmp(&a[k][i], &b[k][i], 8) != 0 && !(a[k][i] == 0.0 && b[k][i] == 0.0)) {
                        std::fprintf(stderr, "pre-flight: %d slabs differ from the undecomposed run: step %d, slot -%d, %s, plane %zu "
                                     "cell %zu: %.17g vs %.17g\n", n_slabs, step, rel, names[k], i / ny, i % ny, b[k][i], a[k][i]);
                        verdict = 4;
                        break;
                    }
        }
        double da = 0.0, db = 0.0;
        if (cfd_total_density(h, &da) != CFD_OK || cfd_multi_total_density(m, &db) != CFD_OK) return die("pre-flight: total density");
        if (!verdict && !(std::fabs(da - db) <= 1e-12 * std::fabs(da))) {
            std::fprintf(stderr, "pre-flight: totalDensity %.17g over %d slabs vs %.17g undecomposed (step %d)\n", db, n_slabs, da, step);
            verdict = 4;
        }
    }
    cfd_multi_destroy(m);
    cfd_destroy(h);
    return verdict;
}

int main(int argc, char** argv) {
    long steps = 5;
    int nx = 301, ny = 151;
    uint32_t flags = 0;
    int show_prop = -1;
    int n_slabs = 0, transport = CFD_TRANSPORT_COPY;
    bool want_preflight = true;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--fast")) flags |= CFD_FLAG_FAST_MATH;
        else if (!std::strcmp(argv[i], "--nx") && i + 1 < argc) nx = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--ny") && i + 1 < argc) ny = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--string-domain") && i + 1 < argc) show_prop = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) { n_slabs = std::atoi(argv[++i]); transport = CFD_TRANSPORT_RCCL; }
        else if (!std::strcmp(argv[i], "--slabs") && i + 1 < argc) { n_slabs = std::atoi(argv[++i]); transport = CFD_TRANSPORT_COPY; }
        else if (!std::strcmp(argv[i], "--no-preflight")) want_preflight = false;
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
        if (want_preflight) {
            cfd_params q = p;
            q.flags = flags;
            alarm(180);         // a stalled rendezvous or exchange ends the program instead of hanging it
            const int verdict = preflight(q, n_slabs, devs.data(), transport);
            alarm(0);
            if (verdict) return verdict;
            std::fprintf(stderr, "pre-flight: %d slabs reproduce the undecomposed run bit for bit over 3 steps\n", n_slabs);
        }
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
