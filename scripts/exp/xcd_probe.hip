// Which XCD does workgroup i of a launch run on?  Two launches on two streams: A (na blocks that hold a compute unit each:
// 155 KB of LDS, 768 threads, ~wait_us microseconds) and, after `pre` one-block kernels on its stream, B (nb blocks of the
// same shape).  Every block records the XCC id it runs on and when it started.  Repeated `reps` times without
// synchronising in between, so that what a queue dealt before is visible in what it deals next.
// build: hipcc --offload-arch=gfx950 -O2 -o xcd_probe xcd_probe.hip     usage: xcd_probe [na nb pre reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(768) hold(int* xcc, long long* t_start, long long wait_ticks) {
    extern __shared__ double lds[];
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        // XCC_ID: hardware register 20, bits 3:0
        xcc[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15);
        t_start[blockIdx.x] = t0;
    }
    lds[threadIdx.x] = (double)threadIdx.x;
    while (wall_clock64() - t0 < wait_ticks) {}
    __syncthreads();
}
__global__ void one_block(int* sink) { if (threadIdx.x == 0 && sink) sink[0] = 1; }

int main(int argc, char** argv) {
    const int na = argc > 1 ? atoi(argv[1]) : 221, nb = argc > 2 ? atoi(argv[2]) : 39, pre = argc > 3 ? atoi(argv[3]) : 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    const size_t lds = 155648;
    CK(hipFuncSetAttribute((const void*)hold, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    int *xa, *xb, *sink; long long *ta, *tb;
    CK(hipMalloc(&xa, reps * na * sizeof(int))); CK(hipMalloc(&xb, reps * nb * sizeof(int))); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&ta, reps * na * sizeof(long long))); CK(hipMalloc(&tb, reps * nb * sizeof(long long)));
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const long long ticks = 30000;      // 100 MHz counter: 300 us
    for (int r = 0; r < reps; ++r) {
        if (r) { CK(hipStreamWaitEvent(sa, eb, 0)); CK(hipStreamWaitEvent(sb, ea, 0)); }
        hipLaunchKernelGGL(hold, dim3(na), dim3(768), lds, sa, xa + r * na, ta + r * na, ticks);
        for (int k = 0; k < pre; ++k) hipLaunchKernelGGL(one_block, dim3(1), dim3(64), 0, sb, sink);
        hipLaunchKernelGGL(hold, dim3(nb), dim3(768), lds, sb, xb + r * nb, tb + r * nb, ticks / 3);
        CK(hipEventRecord(ea, sa)); CK(hipEventRecord(eb, sb));
    }
    CK(hipDeviceSynchronize());
    std::vector<int> ha(reps * na), hb(reps * nb);
    std::vector<long long> hta(reps * na), htb(reps * nb);
    CK(hipMemcpy(ha.data(), xa, ha.size() * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), xb, hb.size() * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hta.data(), ta, hta.size() * sizeof(long long), hipMemcpyDeviceToHost));
    CK(hipMemcpy(htb.data(), tb, htb.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int r = 0; r < reps; ++r) {
        int cnt[16] = {0};
        long long t0 = hta[r * na];
        for (int i = 0; i < na; ++i) { cnt[ha[r * na + i]]++; if (hta[r * na + i] < t0) t0 = hta[r * na + i]; }
        printf("rep %d  A first 16 workgroups on XCC:", r);
        for (int i = 0; i < 16 && i < na; ++i) printf(" %d", ha[r * na + i]);
        printf("   per XCC:");
        for (int k = 0; k < 8; ++k) printf(" %d", cnt[k]);
        printf("\n       B workgroups on XCC (start, us after A's first):");
        for (int i = 0; i < nb; ++i) printf(" %d(%lld)", hb[r * nb + i], (htb[r * nb + i] - t0) / 100);
        printf("\n");
    }
    return 0;
}
