// Runs every host entry point of the harness (the kernels' per-cell arithmetic compiled for the CPU) on small
// ragged grids under AddressSanitizer and UndefinedBehaviorSanitizer: out-of-range neighbour indexing, reads of
// uninitialised structure members and signed overflow in the index arithmetic would stop the program here.
// Built and run by `make -C tests/hostmath sanitize` (CPU only).
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>

extern "C" {
void hm_sweep2d(int, int, double, double, double, double, double, double, double, double, double, double,
                const double* const*, const uint8_t*, double* const*);
void hm_sweep2d_fast(int, int, double, double, double, double, double, double, double, double, double, double,
                     const double* const*, const uint8_t*, double* const*);
void hm_sweep2d_stream(int, int, double, double, double, double, double, double, double, double, double, double,
                       const double* const*, const uint8_t*, double* const*);
void hm_sweep3d(int, int, int, double, double, double, double, double, double, double, double, double, double,
                const double* const*, const uint8_t*, double* const*);
void hm_sweep3d_fast(int, int, int, double, double, double, double, double, double, double, double, double, double,
                     const double* const*, const uint8_t*, double* const*);
void hm_sweep3d_stream(int, int, int, double, double, double, double, double, double, double, double, double, double,
                       const double* const*, const uint8_t*, double* const*);
long hm_div_by_mismatches(double, long, unsigned long long);
double hm_half_superbee(double, double);
}

namespace {
struct Grid {
    int nf;
    size_t n;
    std::vector<std::vector<double>> in, out;
    std::vector<uint8_t> active;
    std::vector<const double*> pin;
    std::vector<double*> pout;
    Grid(int nf_, size_t n_, int density_field, int temperature_field) : nf(nf_), n(n_) {
        unsigned long long s = 88172645463325252ull;
        auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
        in.assign(nf, std::vector<double>(n));
        out.assign(nf, std::vector<double>(n, NAN));
        active.resize(n);
        for (int f = 0; f < nf; ++f)
            for (size_t i = 0; i < n; ++i)
                in[f][i] = f == density_field ? 1.0 + 0.1 * rnd() : f == temperature_field ? 223.0 + 5.0 * rnd() : rnd() - 0.5;
        for (size_t i = 0; i < n; ++i) active[i] = rnd() < 0.8;
        for (int f = 0; f < nf; ++f) { pin.push_back(in[f].data()); pout.push_back(out[f].data()); }
    }
    bool all_written() const {
        for (int f = 0; f < nf; ++f)
            for (size_t i = 0; i < n; ++i)
                if (std::isnan(out[f][i])) return false;
        return true;
    }
};
const double dt = 1e-7, lambda = 1.4e-4, suth = 120.0, R = 289.0, cv = 716.0, k = 0.025, dyef = 0.001;
}  // namespace

int main() {
    int bad = 0;
    const int shapes2[][2] = {{1, 1}, {1, 7}, {5, 1}, {6, 9}, {17, 4}};
    for (auto& sh : shapes2) {
        using Fn = decltype(&hm_sweep2d);
        for (Fn fn : {&hm_sweep2d, &hm_sweep2d_fast, &hm_sweep2d_stream}) {
            Grid g(5, (size_t)sh[0] * sh[1], 2, 3);      // U V Density Temperature Dye
            fn(sh[0], sh[1], 0.01, 0.02, 1.0, dt, lambda, suth, R, cv, k, dyef, g.pin.data(), g.active.data(), g.pout.data());
            if (!g.all_written()) { std::printf("2-D %dx%d: cells left unwritten\n", sh[0], sh[1]); ++bad; }
        }
    }
    const int shapes3[][3] = {{1, 1, 1}, {1, 1, 5}, {4, 1, 1}, {3, 4, 5}, {7, 2, 6}};
    for (auto& sh : shapes3) {
        using Fn = decltype(&hm_sweep3d);
        for (Fn fn : {&hm_sweep3d, &hm_sweep3d_fast, &hm_sweep3d_stream}) {
            Grid g(6, (size_t)sh[0] * sh[1] * sh[2], 3, 4);      // U V W Density Temperature Dye
            fn(sh[0], sh[1], sh[2], 0.01, 0.02, 0.03, dt, lambda, suth, R, cv, k, dyef, g.pin.data(), g.active.data(),
               g.pout.data());
            if (!g.all_written()) { std::printf("3-D %dx%dx%d: cells left unwritten\n", sh[0], sh[1], sh[2]); ++bad; }
        }
    }
    if (hm_div_by_mismatches(0.0053, 20000, 7) != 0) { std::printf("div_by mismatch\n"); ++bad; }
    if (hm_half_superbee(1.0, -1.0) != 0.0) { std::printf("limiter\n"); ++bad; }
    std::printf(bad ? "sanitize: FAILED\n" : "sanitize: ok\n");
    return bad ? 1 : 0;
}
