// Which hardware queue does a stream get?  N streams created in a row; each launches a kernel whose grid size names it
// (stream k: k + 1 workgroups), twice, in the order given by argv[2] ("created" = launch right after each create,
// "after" = create all, then launch).  Read the queue ids from rocprofv3 --kernel-trace.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void k(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1; }
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 6;
    const bool launch_at_create = argc > 2 && !strcmp(argv[2], "created");
    const bool use_null_first = argc > 3;
    int* d; hipMalloc(&d, 64);
    if (use_null_first) { hipLaunchKernelGGL(k, dim3(100), dim3(64), 0, 0, d); hipDeviceSynchronize(); }
    hipStream_t s[16];
    for (int i = 0; i < n; ++i) {
        hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        if (launch_at_create) hipLaunchKernelGGL(k, dim3(i + 1), dim3(64), 0, s[i], d);
    }
    for (int rep = 0; rep < 2; ++rep)
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(i + 1), dim3(64), 0, s[i], d);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
