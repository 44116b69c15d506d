// Second issue-cost microbenchmark: select forms and bit operations, with the number of waves per SIMD
// pinned by the dynamic LDS request (blocks of 256 threads = one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define CHAIN8(INS)                                                             \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)        \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "v"(c), "s"(m))

#define I_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n"
#define I_CND_VCC(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CND_S(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, %10\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_ASHR(n) "v_ashrrev_i32 %" #n ", 31, %" #n "\n"
#define I_XNOR(n) "v_xnor_b32 %" #n ", %" #n ", %8\n"
#define I_CMPI(n) "v_cmp_lt_i32 vcc, %" #n ", %8\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_MOV64(n) "v_mov_b64 %" #n ", %8\n"

template <int OP>
__global__ void __launch_bounds__(256) kern32(double* out, int iters, long long* clk) {
    extern __shared__ double dyn[];
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    float b = 1.0000001f, c = 1e-9f;
    unsigned long long m = 0x5555555555555555ull;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) CHAIN8(I_CND_VCC);
        if (OP == 1) CHAIN8(I_CND_S);
        if (OP == 2) CHAIN8(I_AND);
        if (OP == 3) CHAIN8(I_BFI);
        if (OP == 4) CHAIN8(I_ANDOR);
        if (OP == 5) CHAIN8(I_ASHR);
        if (OP == 6) CHAIN8(I_XNOR);
        if (OP == 7) CHAIN8(I_CMPI);
        if (OP == 8) CHAIN8(I_ADDU);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 1000) dyn[0] = s;
    if ((threadIdx.x & 63) == 0) { clk[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = t1 - t0; clk[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = w1 - w0; }
}

// f64 stream: OP 0 = fma only; 1 = fma + cndmask pairs interleaved (select of a double); 2 = mix like the kernel:
// 6 f64 ops + 2 int ops
template <int OP>
__global__ void __launch_bounds__(256) kern64(double* out, int iters, long long* clk) {
    extern __shared__ double dyn[];
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
    double b = 1.0000001, c = 1e-9;
    unsigned long long m = 0x5555555555555555ull;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) CHAIN8(I_FMA64);
        if (OP == 1) CHAIN8(I_MOV64);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 1000) dyn[0] = s;
    if ((threadIdx.x & 63) == 0) { clk[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = t1 - t0; clk[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = w1 - w0; }
}

// LDS: conflict-free ds_read_b64, 8 loads in flight, no volatile
__global__ void __launch_bounds__(256) kern_lds(double* out, int iters, long long* clk) {
    extern __shared__ double dyn[];
    for (int i = threadIdx.x; i < 2048; i += 256) dyn[i] = i;
    __syncthreads();
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long t0 = clock64(), w0 = wall_clock64();
    const double* p = dyn + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        double v[8];
        asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:2048\n ds_read_b64 %2, %8 offset:4096\n ds_read_b64 %3, %8 offset:6144\n"
                     "ds_read_b64 %4, %8 offset:8192\n ds_read_b64 %5, %8 offset:10240\n ds_read_b64 %6, %8 offset:12288\n ds_read_b64 %7, %8 offset:14336\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])
                     : "v"((unsigned)(size_t)p & 0xffffu));
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += v[i];
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double r = 0;
    for (int i = 0; i < 8; ++i) r += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) { clk[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = t1 - t0; clk[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = w1 - w0; }
}

typedef void (*kfn)(double*, int, long long*);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* out; long long* clk;
    const int maxblocks = cus * 8;
    CHECK(hipMalloc(&out, (size_t)maxblocks * 256 * 8));
    CHECK(hipMalloc(&clk, (size_t)maxblocks * 4 * 16));
    const char* names[] = {"v_cndmask vcc", "v_cndmask_e64 sgpr", "v_and_b32", "v_bfi_b32", "v_and_or_b32", "v_ashrrev_i32", "v_xnor_b32",
                           "v_cmp_lt_i32", "v_add_u32", "v_fma_f64", "v_mov_b64", "ds_read_b64 x8 + 8 add"};
    kfn fns[] = {kern32<0>, kern32<1>, kern32<2>, kern32<3>, kern32<4>, kern32<5>, kern32<6>, kern32<7>, kern32<8>, kern64<0>, kern64<1>, kern_lds};
    const int nk = 12;
    for (int k = 0; k < nk; ++k) CHECK(hipFuncSetAttribute((const void*)fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int iters = 20000;
    std::vector<long long> h(maxblocks * 8);
    for (int wps = 1; wps <= 4; ++wps) {
        const int blocks = cus * wps;
        const size_t lds = (size_t)(160 * 1024 / wps) - 2048;
        for (int k = 0; k < nk; ++k) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(fns[k], dim3(blocks), dim3(256), lds, 0, out, 200, clk);
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(fns[k], dim3(blocks), dim3(256), lds, 0, out, iters, clk);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipGetLastError());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), clk, (size_t)blocks * 4 * 16, hipMemcpyDeviceToHost));
            double cyc = 0, wall = 0;
            for (int b = 0; b < blocks * 4; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
            cyc /= blocks * 4; wall /= blocks * 4;
            const double ninstr = (double)iters * 8;
            printf("waves/SIMD %d  %-24s %6.2f cycles/instr/wave  (per SIMD %6.2f)  clock %.0f MHz  kernel %.3f ms (wave %.3f ms)\n", wps,
                   names[k], cyc / ninstr, cyc / ninstr / wps, cyc / wall * 100.0, ms, wall / 1e5);
        }
    }
    return 0;
}
