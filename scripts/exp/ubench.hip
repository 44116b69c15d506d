// Issue-cost microbenchmark for the vector instructions the sweep kernels are made of.
// Each kernel runs 8 independent chains of one instruction; reported: shader cycles per instruction
// per wave, at 1 and 2 waves per SIMD, plus the shader clock under that load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define CHAIN8(INS)                                                             \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)        \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c))

#define I_FMA(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n"
#define I_ADD(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define I_MUL(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
#define I_MAX(n) "v_max_f64 %" #n ", %" #n ", %8\n"
#define I_MIN(n) "v_min_f64 %" #n ", |%" #n "|, |%8|\n"
#define I_RCP(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define I_RSQ(n) "v_rsq_f64 %" #n ", %" #n "\n"
#define I_SQRT(n) "v_sqrt_f64 %" #n ", %" #n "\n"
#define I_RCP32(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMP(n) "v_cmp_gt_f64 vcc, %" #n ", %8\n"
#define I_FMA32(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_DPP(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_LSHL(n) "v_lshlrev_b64 %" #n ", 1, %" #n "\n"
#define I_ADD64(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"

template <int OP>
__global__ void __launch_bounds__(64) kern32(double* out, int iters, long long* clk) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    float b = 1.0000001f, c = 1e-9f;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 8) CHAIN8(I_RCP32);
        if (OP == 9) CHAIN8(I_AND);
        if (OP == 10) CHAIN8(I_MOV);
        if (OP == 11) CHAIN8(I_CND);
        if (OP == 13) CHAIN8(I_FMA32);
        if (OP == 15) CHAIN8(I_DPP);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int OP>
__global__ void __launch_bounds__(64) kern(double* out, int iters, long long* clk) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    double b = 1.0000001, c = 1e-9;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) CHAIN8(I_FMA);
        if (OP == 1) CHAIN8(I_ADD);
        if (OP == 2) CHAIN8(I_MUL);
        if (OP == 3) CHAIN8(I_MAX);
        if (OP == 4) CHAIN8(I_MIN);
        if (OP == 5) CHAIN8(I_RCP);
        if (OP == 6) CHAIN8(I_RSQ);
        if (OP == 7) CHAIN8(I_SQRT);
        if (OP == 12) CHAIN8(I_CMP);
        if (OP == 14) CHAIN8(I_PKFMA);
        if (OP == 16) CHAIN8(I_LSHL);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

// LDS read throughput: 8 independent ds_read_b64 per iteration
__global__ void __launch_bounds__(64) kern_lds(double* out, int iters, long long* clk) {
    __shared__ double l[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) l[i] = i;
    __syncthreads();
    double s = 0;
    const long long t0 = clock64(), w0 = wall_clock64();
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ((volatile double*)l)[(idx + 64 * i) & 1023];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
        idx = (idx + 1) & 1023;
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

typedef void (*kfn)(double*, int, long long*);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("CUs %d clockRate %d kHz\n", cus, prop.clockRate);
    double* out; long long* clk;
    const int maxblocks = cus * 4 * 4;
    CHECK(hipMalloc(&out, (size_t)maxblocks * 64 * 8));
    CHECK(hipMalloc(&clk, (size_t)maxblocks * 16));
    const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_max_f64", "v_min_f64|abs|", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64",
                           "v_rcp_f32", "v_and_b32", "v_mov_b32", "v_cndmask_b32", "v_cmp_gt_f64", "v_fma_f32", "v_pk_fma_f32",
                           "v_mov_b32_dpp", "v_lshlrev_b64", "ds_read_b64(+add)"};
    kfn fns[] = {kern<0>, kern<1>, kern<2>, kern<3>, kern<4>, kern<5>, kern<6>, kern<7>, kern32<8>, kern32<9>, kern32<10>, kern32<11>,
                 kern<12>, kern32<13>, kern<14>, kern32<15>, kern<16>, kern_lds};
    const int iters = 20000;
    std::vector<long long> h(maxblocks * 2);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * 4 * wps;
        for (int k = 0; k < 18; ++k) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(fns[k], dim3(blocks), dim3(64), 0, 0, out, 200, clk);      // warm
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(fns[k], dim3(blocks), dim3(64), 0, 0, out, iters, clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost));
            double cyc = 0, wall = 0;
            for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
            cyc /= blocks; wall /= blocks;
            const double ninstr = (double)iters * 8;
            printf("waves/SIMD %d  %-18s  %6.2f cycles/instr/wave  (SIMD issue %6.2f cyc/instr)  shader clock %.0f MHz  kernel %.3f ms\n", wps,
                   names[k], cyc / ninstr, cyc / ninstr / wps, cyc / wall * 100.0, ms);
        }
    }
    return 0;
}
