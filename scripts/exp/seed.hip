// Measures the relative error of v_rcp_f64 / v_rsq_f64 seeds and after 1 and 2 Newton steps.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    out[i] = r;
    double e = fma(-v, r, 1.0); double r1 = fma(r, e, r);
    out[n + i] = r1;
    e = fma(-v, r1, 1.0); out[2 * n + i] = fma(r1, e, r1);
    double y = __builtin_amdgcn_rsq(v);
    out[3 * n + i] = y;
    double g = v * y, h = 0.5 * y; double rr = fma(-h, g, 0.5);
    double g1 = fma(g, rr, g), h1 = fma(h, rr, h);
    out[4 * n + i] = g1;
    rr = fma(-h1, g1, 0.5); out[5 * n + i] = fma(g1, rr, g1);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(6 * n);
    for (int i = 0; i < n; ++i) x[i] = 100.0 + 300.0 * (i + 0.5) / n;
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    const char* nm[6] = {"rcp seed", "rcp 1 NR", "rcp 2 NR", "rsq seed", "sqrt 1 it", "sqrt 2 it"};
    for (int c = 0; c < 6; ++c) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            long double ref = c < 3 ? 1.0L / x[i] : (c == 3 ? 1.0L / sqrtl((long double)x[i]) : sqrtl((long double)x[i]));
            long double e = fabsl((o[(size_t)c * n + i] - ref) / ref);
            if (e > worst) worst = e;
        }
        printf("%-10s max rel err %.3Le\n", nm[c], worst);
    }
    return 0;
}
