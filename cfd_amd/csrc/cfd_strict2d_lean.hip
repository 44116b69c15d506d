// The LEAN form of the strict 2-D sweep (sweep2d_march2 in cfd_hip.hip) in a translation unit of its own: built with
// -mllvm -amdgpu-sched-strategy=iterative-minreg, under which this one kernel needs 255 registers and no AGPRs
// (256 + 8 with the default strategy) and therefore runs two waves per SIMD; every other kernel is faster with the
// default strategy, which is a per-module option.
#define CFD_KERNEL_TEMPLATES_ONLY
#include "cfd_hip.hip"

extern "C" int cfd_internal_launch_strict2d_lean(const SweepConsts* K, const double* src, double* dst,
                                                 const uint8_t* act, int x0, int x1, int ny, int xlo, int xhi, int ch,
                                                 void* stream, void* ev_begin, void* ev_end) {
    const dim3 block(128), grid((ny + 125) / 126, (x1 - x0 + ch - 1) / ch);
    hipExtLaunchKernelGGL((sweep2d_march2<MathStrict2, 128, 2, false, 126, true>), grid, block, 0, (hipStream_t)stream,
                          (hipEvent_t)ev_begin, (hipEvent_t)ev_end, 0, *K, src, dst, act, x0, x1, ny, xlo, xhi, ch,
                          (const double*)nullptr, (double*)nullptr);
    return (int)hipGetLastError();
}
