// libcfdhip.so -- fp64 HIP kernels (gfx950) and the C ABI declared in include/cfd_hip.h.
//
// Device layout ("plane-interleaved structure of arrays"): one time slot is
//     value(f, x, y, z) = slot[ ((x+2)*NF + f) * (ny*nz) + y*nz + z ]
// i.e. for every x-plane the NF live fields lie back to back, each as ny*nz contiguous
// doubles.  Lanes run along the fastest index (y*nz+z), so accesses coalesce exactly as in a
// plain structure of arrays, while a whole x-plane of all fields is ONE contiguous run: the
// slab halo that a neighbouring GPU needs is a single message with no pack kernel.  Planes -2, -1
// and nx, nx+1 are ghost planes (used only when a neighbouring slab exists: one per side for a
// single sweep, two for the kernel that does two sweeps in one pass, three -- planes -3 and nx+2 too --
// for the kernel that does a whole step in one pass).
//
// Four buffers per handle: the three time slots of the reference plus one scratch slot that
// receives a sweep which does not advance time (the reference rebuilds slot t from a snapshot
// of slot t; here the result lands in scratch and the two buffers trade roles).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <string>
#include <vector>

#include "../../include/cfd_hip.h"
#include "sweep_math.h"

using namespace cfd;

// CFD_KERNEL_TEMPLATES_ONLY: a second translation unit (cfd_strict2d_lean.hip) includes this file for the kernel
// templates alone and instantiates one of them under other compiler options.
#ifndef CFD_KERNEL_TEMPLATES_ONLY
static constexpr int GHOST = 3;      // ghost x-planes allocated on either side of the owned planes

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// for the library's other translation unit (cfd_multi.hip): records the message cfd_last_error returns
extern "C" int cfd_internal_fail(int code, const char* msg) { return fail(code, msg ? msg : ""); }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(CFD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

// ------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------
struct EventPair {
    hipEvent_t a, b;
    int kind;       // 0 = sweep, 1 = sweep fused with the time filter, 2 = two sweeps (+ filter) in one pass,
                    // 3 = a whole step (three sweeps + filter) in one pass
};
static constexpr int NKIND = 4;

struct cfd_handle {
    cfd_params p;
    int nf;                 // live fields on the device: 5 (2-D, W == 0) or 6
    bool two_d;
    size_t plane;           // ny*nz
    size_t slot_elems;      // nf * (nx + 2*GHOST) * plane
    double* buf[4];         // three time slots + scratch
    int map[4];             // map[t] = buffer holding time slot t (t = 0..2); map[3] = scratch
    uint8_t* d_active;      // (nx + 2*GHOST) * plane bytes; ghost planes zero until cfd_set_ghost_active
    int ghost_mask_depth;   // planes per side whose neighbour masks have been supplied (0, 1 or 2)
    double* d_partial;      // reduction partials
    int n_partial;
    std::vector<double> mew;   // stored Mew: set at creation, never read by the update
    hipStream_t stream;
    bool own_stream;
    int device;
    int cmt, cat;
    SweepConsts K;
    FastConsts KF;
    int chunk;              // x-planes per block of the tiled sweep
    bool chunk_auto;        // default 2-D streaming kernel: chunk length chosen per launch (stream_chunk)
    int cus;                // compute units of the device
    int variant;            // tiled-kernel launch shape (tuning knob)
    bool march3_ok;         // a 3-D x-plane fits 32-bit byte offsets (tiled 3-D kernel usable)
    bool march3_strict_ok;  // ... and the strict tiled 3-D kernel's LDS tile fits a workgroup on this device
    int lds_max;            // LDS bytes a workgroup may use on this device
    bool pair_enabled;      // cfd_step runs its second and third sweep in one pass (env CFD_PAIR, default on)
    int pair_chunk;         // x-planes per block of that kernel (env CFD_PAIR_CHUNK)
    int cus_device;         // compute units of the device (cus: those this handle plans for)
    hipStream_t interior_stream;    // the stream the open one-pass step's interior blocks were launched on
    bool xcd_fill;          // a slab's blocks next to a neighbour are placed on the XCDs with room (env CFD_TRIO_XCD_FILL, default on)
    bool trio_enabled;      // cfd_step runs all three sweeps of a step in one pass (env CFD_TRIO, default on)
    int trio_chunk;         // x-planes per block of that kernel (env CFD_TRIO_CHUNK)
    int trio_first;         // x-planes of its first chunk (a slab's chunks next to a neighbour are shorter: see trio_plan)
    int trio_nts;           // its threads per sweep stage: 256 (one block per compute unit) or 128 (two)
    // a sweep split into cfd_sweep_begin / cfd_sweep_end
    struct {
        bool open, step_time, with_filter, fused;
        bool trio;          // the open call is cfd_step_one_pass_begin: c0, c1 = the chunks it launched
        int c0, c1;
        const double* src;
        double* dst;
        const double* fpp;
        double* fprev;
        int x0, x1;         // the range launched by cfd_sweep_begin
    } pend;
    // sweep timing
    bool timing;
    std::vector<EventPair> ev_used, ev_free;
    double t_ms[NKIND];
    int64_t t_launches[NKIND];
};

// field order inside a plane
//   2-D: U V Density Temperature Dye          (sweep_math.h F2_*)
//   3-D: U V W Density Temperature Dye
static const int kProp2[5] = {0, 1, 3, 4, 6};          // device field -> propIndex
static const int kProp3[6] = {0, 1, 2, 3, 4, 6};

static inline double* slot_base(const cfd_handle* h, int t) {
    // points at plane x = 0, field 0
    return h->buf[h->map[t]] + (size_t)GHOST * h->nf * h->plane;
}

#endif  // CFD_KERNEL_TEMPLATES_ONLY

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
// Arithmetic policies for the 2-D kernels: the strict (reference operation order) and the fast
// (re-associated) form of phase A / phase B share all of the kernel plumbing below.
// A struct of doubles to / from a strided LDS column.
template <class S>
__device__ __forceinline__ void lds_put(double* __restrict__ l, int st, const S& s) {
    constexpr int N = sizeof(S) / sizeof(double);
    double v[N];
    __builtin_memcpy(v, &s, sizeof(S));
#pragma unroll
    for (int i = 0; i < N; ++i) l[i * st] = v[i];
}
template <class S>
__device__ __forceinline__ void lds_get(const double* __restrict__ l, int st, S& s) {
    constexpr int N = sizeof(S) / sizeof(double);
    double v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = l[i * st];
    __builtin_memcpy(&s, v, sizeof(S));
}

// The doubles of a struct of doubles that `need` marks, to / from consecutive rows of a strided LDS column
// (a share may carry products that no right-hand side reads through this channel).
template <int N>
__host__ __device__ constexpr int count_needed(bool (*need)(int)) {
    int n = 0;
    for (int i = 0; i < N; ++i) n += need(i) ? 1 : 0;
    return n;
}
template <class S, class Need>
__device__ __forceinline__ void lds_put_needed(double* __restrict__ l, int st, const S& s, Need need) {
    constexpr int N = sizeof(S) / sizeof(double);
    double v[N];
    __builtin_memcpy(v, &s, sizeof(S));
    int slot = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (need(i)) l[(slot++) * st] = v[i];
}
template <class S, class Need>
__device__ __forceinline__ void lds_get_needed(const double* __restrict__ l, int st, S& s, Need need) {
    constexpr int N = sizeof(S) / sizeof(double);
    double v[N];
    int slot = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = need(i) ? l[(slot++) * st] : 0.0;
    __builtin_memcpy(&s, v, sizeof(S));
}

struct MathStrict2 {
    using Consts = SweepConsts;
    using PA = PhaseA2;
    using SX = ShareX2;
    using SY = ShareY2;
    static constexpr int NSY = 12;
    static constexpr bool OWN_FROM_LDS = true;      // phase B re-reads the cell's own y products from LDS
    // stored U, V, Dye of the cell (which = 0, 1, 2)
    __device__ __forceinline__ static double own(const PA& a, int which) { return a.own[which]; }
    __device__ __forceinline__ static void phaseA(const Consts& K, const double* c, const double* hx,
                                                  const double* n, const double* s, PA& o) {
        phaseA2(K, c, hx, n, s, o);
    }
    __device__ __forceinline__ static void phaseB(const Consts& K, const PA& C, const SX& XE, const SX& XW,
                                                  const SY& YN, const SY& YS, double* o) {
        phaseB2(K, C, XE, XW, YN, YS, o);
    }
    __device__ __forceinline__ static void put_sy(double* __restrict__ l, int st, const SY& s) {
        l[0 * st] = s.aY;    l[1 * st] = s.bY;    l[2 * st] = s.bY23;  l[3 * st] = s.tY;
        l[4 * st] = s.b2Y83; l[5 * st] = s.bY83;  l[6 * st] = s.a2Yh;  l[7 * st] = s.gY;
        l[8 * st] = s.bE23;  l[9 * st] = s.bW23;  l[10 * st] = s.aE;   l[11 * st] = s.aW;
    }
    __device__ __forceinline__ static void get_sy(const double* __restrict__ l, int st, SY& s) {
        s.aY = l[0 * st];    s.bY = l[1 * st];    s.bY23 = l[2 * st];  s.tY = l[3 * st];
        s.b2Y83 = l[4 * st]; s.bY83 = l[5 * st];  s.a2Yh = l[6 * st];  s.gY = l[7 * st];
        s.bE23 = l[8 * st];  s.bW23 = l[9 * st];  s.aE = l[10 * st];   s.aW = l[11 * st];
    }
};

struct MathFast2 {
    using Consts = FastConsts;
    using PA = PhaseAF;
    using SX = ShareXF;
    using SY = ShareYF;
    static constexpr int NSY = 6;
    static constexpr bool OWN_FROM_LDS = false;
    __device__ __forceinline__ static double own(const PA& a, int which) {
        return which == 0 ? a.phi[E2_U] : (which == 1 ? a.phi[E2_V] : a.phi[E2_DYE]);
    }
    __device__ __forceinline__ static void phaseA(const Consts& K, const double* c, const double* hx,
                                                  const double* n, const double* s, PA& o) {
        phaseA2f(K, c, hx, n, s, o);
    }
    __device__ __forceinline__ static void phaseB(const Consts& K, const PA& C, const SX& XE, const SX& XW,
                                                  const SY& YN, const SY& YS, double* o) {
        phaseB2f(K, C, XE, XW, YN, YS, o);
    }
    __device__ __forceinline__ static void put_sy(double* __restrict__ l, int st, const SY& s) {
        l[0 * st] = s.a;  l[1 * st] = s.b;  l[2 * st] = s.g;  l[3 * st] = s.py;  l[4 * st] = s.deb;  l[5 * st] = s.dea;
    }
    __device__ __forceinline__ static void get_sy(const double* __restrict__ l, int st, SY& s) {
        s.a = l[0 * st];  s.b = l[1 * st];  s.g = l[2 * st];  s.py = l[3 * st];  s.deb = l[4 * st];  s.dea = l[5 * st];
    }
};

// Row addressing: `x` is uniform across the block, so the plane base stays in scalar registers and
// a lane adds only a 32-bit element offset.
__device__ __forceinline__ void load5(const double* __restrict__ src, long plane5, int ny, int x, int y,
                                      double v[NF2]) {
    const double* __restrict__ row = src + (long)x * plane5;
#pragma unroll
    for (int f = 0; f < NF2; ++f) v[f] = row[(unsigned)(f * ny + y)];
}

__device__ __forceinline__ void store5(const double* __restrict__ src, double* __restrict__ dst,
                                       const uint8_t* __restrict__ active, long plane5, int ny, int x, int y,
                                       const double o[5]) {
    const bool act = (active + (long)x * ny)[(unsigned)y] != 0;
    const double* __restrict__ sp = src + (long)x * plane5;
    double* __restrict__ dp = dst + (long)x * plane5;
    const unsigned iR = F2_RHO * ny + y, iT = F2_T * ny + y, iU = F2_U * ny + y, iV = F2_V * ny + y,
                   iD = F2_DYE * ny + y;
    dp[iR] = o[E2_RHO];
    dp[iT] = o[E2_T];
    dp[iU] = act ? o[E2_U] : sp[iU];
    dp[iV] = act ? o[E2_V] : sp[iV];
    dp[iD] = act ? o[E2_DYE] : sp[iD];
}

// One thread per cell; every thread evaluates phase A for its cell and again for its four face
// neighbours.  Redundant by design: this is the plain reading of the update, kept as the
// on-device cross-check of the tiled kernel.
template <class M>
__device__ __forceinline__ void phaseA2_at(const typename M::Consts& K, const double* __restrict__ src, long plane5,
                                           int ny, int x, int y, int xlo, int xhi, typename M::PA& o) {
    const int xe = x + 1 > xhi ? xhi : x + 1, xw = x - 1 < xlo ? xlo : x - 1;
    const int yn = y + 1 > ny - 1 ? ny - 1 : y + 1, ys = y - 1 < 0 ? 0 : y - 1;
    double c[NF2], e[NF2], w[NF2], n[NF2], s[NF2];
    load5(src, plane5, ny, x, y, c);
    load5(src, plane5, ny, xe, y, e);
    load5(src, plane5, ny, xw, y, w);
    load5(src, plane5, ny, x, yn, n);
    load5(src, plane5, ny, x, ys, s);
    double hx[NF2];
    x_half_slopes2(c, e, w, hx);
    M::phaseA(K, c, hx, n, s, o);
}

template <class M>
__global__ void __launch_bounds__(64)
sweep2d_simple(typename M::Consts K, const double* __restrict__ src, double* __restrict__ dst,
               const uint8_t* __restrict__ active, int x0, int nx, int ny, int xlo, int xhi) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = x0 + (int)blockIdx.y;
    if (y >= ny || x >= nx) return;
    const long plane5 = (long)NF2 * ny;
    const int xe = x + 1 > xhi ? xhi : x + 1, xw = x - 1 < xlo ? xlo : x - 1;
    const int yn = y + 1 > ny - 1 ? ny - 1 : y + 1, ys = y - 1 < 0 ? 0 : y - 1;
    typename M::PA C, T;
    typename M::SX XE, XW;
    typename M::SY YN, YS;
    phaseA2_at<M>(K, src, plane5, ny, x, y, xlo, xhi, C);
    phaseA2_at<M>(K, src, plane5, ny, xe, y, xlo, xhi, T);  XE = T.sx;
    phaseA2_at<M>(K, src, plane5, ny, xw, y, xlo, xhi, T);  XW = T.sx;
    phaseA2_at<M>(K, src, plane5, ny, x, yn, xlo, xhi, T);  YN = T.sy;
    phaseA2_at<M>(K, src, plane5, ny, x, ys, xlo, xhi, T);  YS = T.sy;
    double o[5];
    M::phaseB(K, C, XE, XW, YN, YS, o);
    store5(src, dst, active, plane5, ny, x, y, o);
}

// Arithmetic policies for the 3-D kernels.
// Products of ShareY3 / ShareZ3 that a right-hand side reads from a y / z neighbour (by position in the struct):
// everything but the half-square of the velocity component along the share's own direction.
__host__ __device__ constexpr bool need_sy3(int i) { return i != 8; }      // s.sqh[1]
__host__ __device__ constexpr bool need_sz3(int i) { return i != 9; }      // s.sqh[2]

struct MathStrict3 {
    using Consts = SweepConsts;
    using PA = PhaseA3;
    using SX = ShareX3;
    using SY = ShareY3;
    using SZ = ShareZ3;
    static constexpr int NSY = count_needed<19>(need_sy3), NSZ = count_needed<25>(need_sz3);
    static constexpr bool OWN_FROM_LDS = true;      // phase B re-reads the cell's own y / z products from LDS
    static constexpr const char* name = "strict";
    // stored U, V, W, Dye of the cell (which = 0..3)
    __device__ __forceinline__ static double own(const PA& a, int which) { return a.own[which]; }
    __device__ __forceinline__ static void phaseA(const Consts& K, const double* c, const double (*h)[NF3], PA& o) {
        phaseA3_h(K, c, h, o);
    }
    __device__ __forceinline__ static void phaseB(const Consts& K, const PA& C, const SX& XP, const SX& XM,
                                                  const SY& YP, const SY& YM, const SZ& ZP, const SZ& ZM, double* o) {
        phaseB3(K, C, XP, XM, YP, YM, ZP, ZM, o);
    }
    __device__ __forceinline__ static void put_sy(double* __restrict__ l, int st, const SY& s) {
        lds_put_needed(l, st, s, need_sy3);
    }
    __device__ __forceinline__ static void get_sy(const double* __restrict__ l, int st, SY& s) {
        lds_get_needed(l, st, s, need_sy3);
    }
    __device__ __forceinline__ static void put_sz(double* __restrict__ l, int st, const SZ& s) {
        lds_put_needed(l, st, s, need_sz3);
    }
    __device__ __forceinline__ static void get_sz(const double* __restrict__ l, int st, SZ& s) {
        lds_get_needed(l, st, s, need_sz3);
    }
};

struct MathFast3 {
    using Consts = FastConsts;
    using PA = PhaseA3F;
    using SX = ShareX3F;
    using SY = ShareY3F;
    using SZ = ShareZ3F;
    static constexpr int NSY = sizeof(SY) / sizeof(double), NSZ = sizeof(SZ) / sizeof(double);
    static constexpr bool OWN_FROM_LDS = false;
    static constexpr const char* name = "fast";
    __device__ __forceinline__ static double own(const PA& a, int which) {
        return which == 3 ? a.phi[E3_DYE] : a.phi[E3_U + which];
    }
    __device__ __forceinline__ static void phaseA(const Consts& K, const double* c, const double (*h)[NF3], PA& o) {
        phaseA3f_h(K, c, h, o);
    }
    __device__ __forceinline__ static void phaseB(const Consts& K, const PA& C, const SX& XP, const SX& XM,
                                                  const SY& YP, const SY& YM, const SZ& ZP, const SZ& ZM, double* o) {
        phaseB3f(K, C, XP, XM, YP, YM, ZP, ZM, o);
    }
    __device__ __forceinline__ static void put_sy(double* __restrict__ l, int st, const SY& s) { lds_put(l, st, s); }
    __device__ __forceinline__ static void get_sy(const double* __restrict__ l, int st, SY& s) { lds_get(l, st, s); }
    __device__ __forceinline__ static void put_sz(double* __restrict__ l, int st, const SZ& s) { lds_put(l, st, s); }
    __device__ __forceinline__ static void get_sz(const double* __restrict__ l, int st, SZ& s) { lds_get(l, st, s); }
};

// General 3-D sweep, one thread per cell (lanes along z, then y).  Each thread evaluates phase A
// for its cell and the parts of its six face neighbours' phase A that it differences.
template <class M>
__device__ __forceinline__ void phaseA3_at(const typename M::Consts& K, const double* __restrict__ src, long plane,
                                           int ny, int nz, int x, int y, int z, int xlo, int xhi,
                                           typename M::PA& o) {
    const int xe = x + 1 > xhi ? xhi : x + 1, xw = x - 1 < xlo ? xlo : x - 1;
    const int yn = y + 1 > ny - 1 ? ny - 1 : y + 1, ys = y - 1 < 0 ? 0 : y - 1;
    const int zt = z + 1 > nz - 1 ? nz - 1 : z + 1, zb = z - 1 < 0 ? 0 : z - 1;
    const long o6[7] = {(long)x * NF3 * plane + (long)y * nz + z,
                        (long)xe * NF3 * plane + (long)y * nz + z,  (long)xw * NF3 * plane + (long)y * nz + z,
                        (long)x * NF3 * plane + (long)yn * nz + z,  (long)x * NF3 * plane + (long)ys * nz + z,
                        (long)x * NF3 * plane + (long)y * nz + zt,  (long)x * NF3 * plane + (long)y * nz + zb};
    double c[NF3], nb[6][NF3];
#pragma unroll
    for (int f = 0; f < NF3; ++f) {
        c[f] = src[o6[0] + (long)f * plane];
#pragma unroll
        for (int s = 0; s < 6; ++s) nb[s][f] = src[o6[s + 1] + (long)f * plane];
    }
    double h[3][NF3];
    half_slopes3(c, nb, h);
    M::phaseA(K, c, h, o);
}

template <class M>
__global__ void __launch_bounds__(64)
sweep3d_simple(typename M::Consts K, const double* __restrict__ src, double* __restrict__ dst,
               const uint8_t* __restrict__ active, int x0, int nx, int ny, int nz, int xlo, int xhi) {
    const long plane = (long)ny * nz;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int x = x0 + (int)blockIdx.y;
    if (i >= plane || x >= nx) return;
    const int y = (int)(i / nz), z = (int)(i - (long)y * nz);
    const int xe = x + 1 > xhi ? xhi : x + 1, xw = x - 1 < xlo ? xlo : x - 1;
    const int yn = y + 1 > ny - 1 ? ny - 1 : y + 1, ys = y - 1 < 0 ? 0 : y - 1;
    const int zt = z + 1 > nz - 1 ? nz - 1 : z + 1, zb = z - 1 < 0 ? 0 : z - 1;
    typename M::PA C, T;
    typename M::SX XP, XM;
    typename M::SY YP, YM;
    typename M::SZ ZP, ZM;
    phaseA3_at<M>(K, src, plane, ny, nz, x, y, z, xlo, xhi, C);
    phaseA3_at<M>(K, src, plane, ny, nz, xe, y, z, xlo, xhi, T);  XP = T.sx;
    phaseA3_at<M>(K, src, plane, ny, nz, xw, y, z, xlo, xhi, T);  XM = T.sx;
    phaseA3_at<M>(K, src, plane, ny, nz, x, yn, z, xlo, xhi, T);  YP = T.sy;
    phaseA3_at<M>(K, src, plane, ny, nz, x, ys, z, xlo, xhi, T);  YM = T.sy;
    phaseA3_at<M>(K, src, plane, ny, nz, x, y, zt, xlo, xhi, T);  ZP = T.sz;
    phaseA3_at<M>(K, src, plane, ny, nz, x, y, zb, xlo, xhi, T);  ZM = T.sz;
    double o[NE3];
    M::phaseB(K, C, XP, XM, YP, YM, ZP, ZM, o);
    const bool act = active[(long)x * plane + i] != 0;
    const double* sp = src + (long)x * NF3 * plane + i;
    double* dp = dst + (long)x * NF3 * plane + i;
    dp[(long)F3_RHO * plane] = o[E3_RHO];
    dp[(long)F3_T * plane] = o[E3_T];
    dp[(long)F3_U * plane] = act ? o[E3_U] : sp[(long)F3_U * plane];
    dp[(long)F3_V * plane] = act ? o[E3_V] : sp[(long)F3_V * plane];
    dp[(long)F3_W * plane] = act ? o[E3_W] : sp[(long)F3_W * plane];
    dp[(long)F3_DYE * plane] = act ? o[E3_DYE] : sp[(long)F3_DYE * plane];
}

// Workgroups are dealt to the 8 XCDs (each with its own L2) round-robin by linear id.  This maps the
// linear id so that the workgroups one XCD receives are consecutive tiles (neighbouring tiles share
// halo columns / cache lines, which then hit in that L2).  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned orig, unsigned nwg) {
    const unsigned q = nwg / 8u, r = nwg % 8u, xcd = orig % 8u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + orig / 8u;
}

// A launch that starts while another launch (of `busy` one-block-per-compute-unit workgroups) holds most of the device.
// The hardware deals the workgroups of a launch to the XCDs in turn -- workgroup i of a launch on hardware queue q runs
// on XCD (i + o_q) % 8, o_q a constant of the queue (measured: xcc_offset_of) -- whether or not that XCD has a compute
// unit free, so a workgroup dealt to a full XCD waits for a whole block of the other launch to end while compute
// units idle elsewhere (measured on a slab's one-pass step: 1.07 ms instead of 0.71).  xcd_fill_rank places the work
// where the room is: workgroup `b` of the second launch takes a unit of work only if its XCD still has a compute unit
// free after the first launch's share (`cap` compute units per XCD) and the earlier workgroups of this launch; it
// returns the rank of that unit among the launch's units of work, or -1 for a workgroup that is only there to fill a
// turn (it leaves at once).  oa, ob: the queue constants of the first and of this launch.
// xcd_fill_grid: workgroups to launch for `units` units of work (0: they do not fit beside `busy`).
__host__ __device__ inline int xcd_free(int busy, int cap, int xcd, int oa) {
    const int first = (xcd - oa + 8) % 8;       // the first workgroup of the other launch that runs on this XCD
    const int used = busy > first ? (busy - first + 7) / 8 : 0;
    return cap > used ? cap - used : 0;
}
__host__ __device__ inline int xcd_fill_rank(int b, int busy, int cap, int oa, int ob) {
    const int xcd = (b + ob) % 8, turn = b / 8;
    if (turn >= xcd_free(busy, cap, xcd, oa)) return -1;
    int rank = 0;
    for (int x = 0; x < 8; ++x) {
        const int f = xcd_free(busy, cap, x, oa), before = turn + ((x - ob + 8) % 8 < b % 8 ? 1 : 0);
        rank += f < before ? f : before;
    }
    return rank;
}
inline int xcd_fill_grid(int units, int busy, int cap, int oa, int ob) {
    int total = 0;
    for (int x = 0; x < 8; ++x) total += xcd_free(busy, cap, x, oa);
    if (total < units) return 0;
    int placed = 0;
    for (int b = 0;; ++b)
        if (xcd_fill_rank(b, busy, cap, oa, ob) >= 0 && ++placed == units) return b + 1;
}

// How the tiles of a 2-D tiled launch map to blocks.  A row of ny columns is cut into tiles of wcols; every
// tile but possibly the last is full.  A last tile narrow enough to fit one wave together with its halo lanes
// (nlast + halo <= 64) does not get a block of its own per chunk of x-planes: the nt / 64 waves of one block
// take that tile for nt / 64 consecutive chunks ("packed" blocks; each wave is a complete, independent tile).
// Linear block ids: nfull * nchunk ordinary blocks (tile fastest), then npacked packed ones.
struct TilePlan { int nfull, npacked, pack; };
__host__ __device__ inline TilePlan tile_plan(int ny, int wcols, int nt, int halo, int nchunk) {
    const int ntile = (ny + wcols - 1) / wcols, nlast = ny - (ntile - 1) * wcols, pack = nt / 64;
    const bool p = pack > 1 && nlast + halo <= 64;
    return TilePlan{p ? ntile - 1 : ntile, p ? (nchunk + pack - 1) / pack : 0, pack};
}

// Buffer-resource addressing: one 128-bit descriptor per x-plane (built from block-uniform values,
// so it lives in scalar registers), a per-lane 32-bit byte offset for the column, and a scalar
// byte offset for the field.  No per-lane 64-bit address arithmetic; an out-of-range per-lane
// offset makes the hardware drop the access, which is how non-writing lanes are masked.
typedef int cfd_v2i __attribute__((ext_vector_type(2)));

struct PlaneRef {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ PlaneRef(const void* base, unsigned bytes)
        : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000)) {}
    __device__ __forceinline__ double ld(unsigned voff, unsigned soff) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
    __device__ __forceinline__ unsigned char ldb(unsigned voff) const {
        return __builtin_amdgcn_raw_buffer_load_b8(r, voff, 0, 0);
    }
    __device__ __forceinline__ void st(unsigned voff, unsigned soff, double x) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cfd_v2i, x), r, voff, soff, 0);
    }
    // non-temporal store (written once, not read again by this kernel)
    __device__ __forceinline__ void st_nt(unsigned voff, unsigned soff, double x) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cfd_v2i, x), r, voff, soff, 2);
    }

};

// Tiled sweep, shared-product arrangement ("march"): a block owns WCOLS adjacent columns (lanes run
// along y, the contiguous index) plus one halo column on each side, and walks a chunk of x-planes.
//   * a thread reads only its own column from memory; stored values and phase-A results of rows
//     x-1, x, x+1 rotate through three register sets (the loop is unrolled by three, so nothing is
//     moved): the East/West neighbour products never leave the thread.  The x half-slopes are taken
//     first, which retires the West row, and the load of row j+2 is issued straight into its
//     registers;
//   * the North/South neighbours' stored values and products of the current row cross lanes through
//     double-buffered LDS tiles, one barrier per row.  The two halo lanes have no neighbour on their
//     outer side and use their own value there: what they publish for their inner neighbour (centre
//     products and East/West face products) does not depend on their y-slopes, and they store nothing;
//   * phase A is evaluated exactly once per cell (plus the two halo columns and two extra rows per
//     chunk), and every cell is read from HBM once;
//   * FILTER: the sweep that ends a step also applies the time filter (correctPrevTime) to the
//     values it has just computed: prev <- pp + 0.5*(new - pp), so the new slot is not read back.
//   iteration j:  x half-slopes of row j; issue the loads of row j+2 and of the mask of row j
//                 phase A of row j  ->  publish its y-products in lds[j&1]
//                 phase B of row j-1 (centre = row j-1, East = row j, West = row j-2,
//                                     North/South = lds[(j-1)&1] of the neighbouring lanes)
//                 one barrier
//   LEAN (strict arithmetic, no filter): 39 instead of 54 doubles of LDS per thread, so that four 128-thread blocks
//   fit a compute unit (two waves per SIMD): the neighbours' stored values are single-buffered behind a second
//   barrier per row, and only the left-hand sides are parked in LDS (the coefficients stay in registers).
template <class M, int NT, int MINW, bool FILTER, int WCOLS = NT - 2, bool LEAN = false>
__global__ void __launch_bounds__(NT, MINW)
sweep2d_march2(typename M::Consts K, const double* __restrict__ src, double* __restrict__ dst,
               const uint8_t* __restrict__ active, int x0, int nx, int ny, int xlo, int xhi, int ch,
               const double* __restrict__ fpp, double* __restrict__ fprev) {
    constexpr bool ALWAYS = FILTER;         // finish every row, also the one before the chunk (see emit)
    __shared__ double lds[2][M::NSY][NT];
    __shared__ double lraw[LEAN ? 1 : 2][NF2][NT];
    // OWN_FROM_LDS: a thread's private slot, two rows deep, for the left-hand sides and coefficients of the
    // row it has evaluated but not yet finished (instead of carrying them in registers for a row)
    __shared__ double lown[2][LEAN ? 5 : 10][M::OWN_FROM_LDS ? NT : 1];
    const int t = threadIdx.x;
    // Lanes 0 .. WCOLS-1 hold the block's own columns in order (with WCOLS a multiple of 16 every
    // wave's store covers whole 128-byte lines), lane WCOLS the column above the tile, lane WCOLS+1
    // the column below it; lanes past that idle.  Tiles are taken in XCD-contiguous order.
    const unsigned wg = xcd_contiguous_id(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int bx = (int)(wg % gridDim.x), by = (int)(wg / gridDim.x);
    const int ytile = bx * WCOLS;
    // nv: the tile's own columns that exist (fewer than WCOLS in the last tile of a row); the two halo lanes
    // follow them directly, and waves that hold no used lane leave at once (a barrier counts only the waves
    // of the block that have not ended), so a narrow last tile costs one wave instead of NT / 64.
    const int nv = ny - ytile < WCOLS ? ny - ytile : WCOLS;
    if ((int)(__builtin_amdgcn_readfirstlane(t) & ~63) > nv + 1) return;
    const int y = t < nv ? ytile + t : (t == nv ? ytile + nv : ytile - 1);
    const int cy = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
    const int yn = cy + 1 > ny - 1 ? ny - 1 : cy + 1, ys = cy - 1 < 0 ? 0 : cy - 1;
    auto lane_of = [=](int c) {         // lane that holds column c; outside the tile: this lane (own value)
        const int r = c - ytile;
        return r >= 0 && r < nv ? r : (r == nv ? nv : (r == -1 ? nv + 1 : t));
    };
    const int lnc = lane_of(yn), lsc = lane_of(ys);
    const bool writer = t < nv;
    const int xs = x0 + by * ch;
    const int xe = xs + ch < nx ? xs + ch : nx;
    const int j0 = xs - 1 > xlo ? xs - 1 : xlo;
    const int j1 = xe < xhi ? xe : xhi;
    const long plane5 = (long)NF2 * ny;
    const unsigned pbytes = (unsigned)(plane5 * 8), fstride = (unsigned)ny * 8u;
    const unsigned vc = t <= nv + 1 ? (unsigned)cy * 8u : 0xFFFFFFF0u;     // lanes past the tile load nothing
    const unsigned vst_own = writer ? (unsigned)y * 8u : 0xFFFFFFF0u;      // out of range: store dropped
    auto clampx = [=](int x) { return x < xlo ? xlo : (x > xhi ? xhi : x); };
    auto load_row = [&](int x, unsigned voff, double v[NF2]) {
        const PlaneRef p(src + (long)x * plane5, pbytes);
#pragma unroll
        for (int f = 0; f < NF2; ++f) v[f] = p.ld(voff, (unsigned)f * fstride);
    };
    auto load_act = [&](int x) { return PlaneRef(active + (long)x * ny, (unsigned)ny).ldb((unsigned)cy); };

    double R0[NF2], R1[NF2], R2[NF2];       // stored values of three consecutive rows (own column)
    typename M::PA P0, P1, P2;
    unsigned char act_c = 1;

    // stored values of row x in slot t-2, for the fused filter (issued early, consumed in emit)
    auto load_pp = [&](int x, double pp[NF2]) {
        if (FILTER) {
            const PlaneRef q(fpp + (long)x * plane5, pbytes);
#pragma unroll
            for (int f = 0; f < NF2; ++f) pp[f] = q.ld(vc, (unsigned)f * fstride);
        }
    };
    // kept[]: OWN_FROM_LDS only -- the stored U, V, Dye of the row, fetched by row() for the inactive lanes
    // live: the row belongs to this block's chunk; otherwise (the row before the chunk) everything is computed and
    // every store is dropped by its out-of-range offset.  With the filter, rows are finished unconditionally
    // (ALWAYS): that keeps the compiler's vector-memory wait counts exact -- with the stores behind a branch it has
    // to wait for ALL outstanding operations, the previous row's ten stores included, at the top of every row
    // (-10 % on that launch).  Without the filter the branch stays: five stores drain fast enough, and the
    // unconditional form measured 1 % slower.
    auto emit = [&](int x, bool live, const typename M::PA& C, const typename M::SX& XE, const typename M::SX& XW,
                    const double pp[NF2], const double kept[3]) {
        const unsigned vst = live ? vst_own : 0xFFFFFFF0u;
        const int xr = live ? x : xs;           // a row the planes' descriptors can be built on
        typename M::SY YN, YS;
        M::get_sy(&lds[x & 1][0][lnc], NT, YN);
        M::get_sy(&lds[x & 1][0][lsc], NT, YS);
        double o[5];
        if constexpr (M::OWN_FROM_LDS) {
            typename M::PA Cl = C;
            M::get_sy(&lds[x & 1][0][t], NT, Cl.sy);
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                Cl.lhs[e] = lown[x & 1][e][t];
                if constexpr (!LEAN) Cl.aU[e] = lown[x & 1][5 + e][t];
            }
            M::phaseB(K, Cl, XE, XW, YN, YS, o);
        } else {
            M::phaseB(K, C, XE, XW, YN, YS, o);
        }
        const bool act = act_c != 0;
        const lane_mask am = mask_of(act);
        auto own = [&](int i) { if constexpr (M::OWN_FROM_LDS) return kept[i]; else return M::own(C, i); };
        const double nU = select(am, o[E2_U], own(0)), nV = select(am, o[E2_V], own(1)),
                     nD = select(am, o[E2_DYE], own(2));
        const PlaneRef d(dst + (long)xr * plane5, pbytes);
        d.st_nt(vst, (unsigned)F2_RHO * fstride, o[E2_RHO]);
        d.st_nt(vst, (unsigned)F2_T * fstride, o[E2_T]);
        d.st_nt(vst, (unsigned)F2_U * fstride, nU);
        d.st_nt(vst, (unsigned)F2_V * fstride, nV);
        d.st_nt(vst, (unsigned)F2_DYE * fstride, nD);
        if (FILTER) {
            // Density and Temperature in every cell; U, V, Dye only where the cell is active
            const PlaneRef pv(fprev + (long)xr * plane5, pbytes);
            const unsigned vm = act ? vst : 0xFFFFFFF0u;
            pv.st_nt(vst, (unsigned)F2_RHO * fstride, pp[F2_RHO] + (0.5 * (o[E2_RHO] - pp[F2_RHO])));
            pv.st_nt(vst, (unsigned)F2_T * fstride, pp[F2_T] + (0.5 * (o[E2_T] - pp[F2_T])));
            pv.st_nt(vm, (unsigned)F2_U * fstride, pp[F2_U] + (0.5 * (nU - pp[F2_U])));
            pv.st_nt(vm, (unsigned)F2_V * fstride, pp[F2_V] + (0.5 * (nV - pp[F2_V])));
            pv.st_nt(vm, (unsigned)F2_DYE * fstride, pp[F2_DYE] + (0.5 * (nD - pp[F2_DYE])));
        }
    };
    // stored U, V, Dye of row x for the lanes whose cell is inactive there (the others fetch nothing)
    auto load_kept = [&](int x, double kept[3]) {
        if constexpr (M::OWN_FROM_LDS) {
            const PlaneRef p(src + (long)x * plane5, pbytes);
            const unsigned v = act_c != 0 ? 0xFFFFFFF0u : vc;
            kept[0] = p.ld(v, (unsigned)F2_U * fstride);
            kept[1] = p.ld(v, (unsigned)F2_V * fstride);
            kept[2] = p.ld(v, (unsigned)F2_DYE * fstride);
        }
    };
    // Row j: Rw, Rc, Re hold rows j-1, j, j+1 (clamped); lraw[j&1] holds row j of every lane.
    auto row = [&](int j, double* Rw, const double* Rc, const double* Re, typename M::PA& An,
                   const typename M::PA& Ac, const typename M::PA& Aw, bool first) {
        double hx[NF2], pp[NF2], kept[3];
        x_half_slopes2(Rc, Re, Rw, hx);
        // Rw is dead: fetch row j+2 into it; also this row's mask
        load_row(clampx(j + 2), vc, Rw);
        if (!first && (ALWAYS || j - 1 >= xs)) {        // (first: a compile-time constant at every call)
            const int xr = j - 1 >= xs ? j - 1 : xs;
            load_pp(xr, pp);
            load_kept(xr, kept);
        }
        const unsigned char act_n = load_act(j);
        double rn[NF2], rs[NF2];
        if constexpr (LEAN) {
            // one tile: every lane has read row j before any lane overwrites it with row j+1
#pragma unroll
            for (int f = 0; f < NF2; ++f) {
                rn[f] = lraw[0][f][lnc];
                rs[f] = lraw[0][f][lsc];
            }
            __syncthreads();
#pragma unroll
            for (int f = 0; f < NF2; ++f) lraw[0][f][t] = Re[f];
        } else {
#pragma unroll
            for (int f = 0; f < NF2; ++f) lraw[(j + 1) & 1][f][t] = Re[f];
#pragma unroll
            for (int f = 0; f < NF2; ++f) {
                rn[f] = lraw[j & 1][f][lnc];
                rs[f] = lraw[j & 1][f][lsc];
            }
        }
        M::phaseA(K, Rc, hx, rn, rs, An);
        M::put_sy(&lds[j & 1][0][t], NT, An.sy);
        if constexpr (M::OWN_FROM_LDS) {
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                lown[j & 1][e][t] = An.lhs[e];
                if constexpr (!LEAN) lown[j & 1][5 + e][t] = An.aU[e];
            }
        }
        const int x = j - 1;
        if (!first && (ALWAYS || x >= xs)) emit(x, x >= xs, Ac, An.sx, Aw.sx, pp, kept);
        __syncthreads();
        act_c = act_n;
    };

    load_row(clampx(j0 - 1), vc, R0);
    load_row(j0, vc, R1);
    load_row(clampx(j0 + 1), vc, R2);
#pragma unroll
    for (int f = 0; f < NF2; ++f) lraw[LEAN ? 0 : (j0 & 1)][f][t] = R1[f];
    __syncthreads();

    row(j0, R0, R1, R2, P0, P0, P0, true);
    P2.sx = P0.sx;      // stands in for the row before the first: the clamp-to-self rule
    int j = j0 + 1, last = 0;
    for (;;) {
        if (j > j1) { last = 0; break; }
        row(j, R1, R2, R0, P1, P0, P2, false); ++j;
        if (j > j1) { last = 1; break; }
        row(j, R2, R0, R1, P2, P1, P0, false); ++j;
        if (j > j1) { last = 2; break; }
        row(j, R0, R1, R2, P0, P2, P1, false); ++j;
    }
    if (j1 < xe) {
        double pp[NF2], kept[3];
        load_pp(xe - 1, pp);
        load_kept(xe - 1, kept);
        if (last == 0) emit(xe - 1, true, P0, P0.sx, P2.sx, pp, kept);
        else if (last == 1) emit(xe - 1, true, P1, P1.sx, P0.sx, pp, kept);
        else emit(xe - 1, true, P2, P2.sx, P1.sx, pp, kept);
    }
}

// Streaming form of the tiled 2-D sweep (re-associated arithmetic only; the default 2-D kernel for
// it: 256 threads, 240 stored columns, CFD_VARIANT=20..34 select other shapes).  Same tiling and LDS
// traffic pattern as sweep2d_march2, but a row's phase-A results are turned at once into q, coef
// and the four pre-scaled neighbour messages (phaseA2s), so the only state a thread carries from
// row to row is the pending row's q / coef / new Density, the message for the next row's West side,
// and a ring of RD rows of stored values (RD - 2 rows of prefetch): ~166 registers instead of ~210.
// Every row is finished unconditionally (stores that do not belong to the chunk are dropped by an
// out-of-range offset) and the prologue loads are retired before the loop, which keeps the
// compiler's vector-memory wait counts exact: a row's stores stay in flight under the next row.
// WCOLS: columns stored per block (NT - 2, or a multiple of 16 for tiles aligned to 128-byte lines).
//   iteration j:  x half-slopes of row j; issue the load of row j+RD-1 and of the mask of row j+1
//                 phase A of row j (North/South stored values from lraw[j&1])
//                 fold the West message (row j-1) into q; publish the North/South messages
//                 finish row j-1: East message (row j) + North/South messages (lmsg[(j-1)&1]), store
//                 one barrier
template <int NT, int MINW, bool FILTER, int RD = 3, int WCOLS = NT - 2, bool LIGHT = (RD == 3)>
__global__ void __launch_bounds__(NT, MINW)
sweep2d_stream(FastConsts K, const double* __restrict__ src, double* __restrict__ dst,
               const uint8_t* __restrict__ active, int x0, int nx, int ny, int xlo, int xhi, int ch,
               const double* __restrict__ fpp, double* __restrict__ fprev, int swizzle) {
    __shared__ double lmsg[2][6][NT];
    __shared__ double lraw[2][NF2][NT];
    const int t = threadIdx.x;
    // Lane -> column.  Lanes 0 .. WCOLS-1 hold the block's own columns in order, so that with WCOLS a
    // multiple of 16 every wave's store covers whole, aligned 128-byte lines (a tile that starts one
    // lane late halves the store bandwidth); lane WCOLS holds the column above the tile, lane
    // WCOLS+1 the column below it, lanes past that idle.
    const int nchunk = (nx - x0 + ch - 1) / ch;
    const TilePlan tp = tile_plan(ny, WCOLS, NT, 2, nchunk);
    const unsigned wg = swizzle ? xcd_contiguous_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const int wave0 = (int)(__builtin_amdgcn_readfirstlane(t) & ~63);      // first lane of this wave (scalar)
    int bx, by, lb = 0;
    if (wg < (unsigned)(tp.nfull * nchunk)) {
        bx = (int)(wg % (unsigned)tp.nfull);
        by = (int)(wg / (unsigned)tp.nfull);
    } else {        // packed block: this wave is the row's narrow last tile for one chunk, lanes lb .. lb + 63
        bx = tp.nfull;
        by = (int)(wg - (unsigned)(tp.nfull * nchunk)) * tp.pack + (wave0 >> 6);
        lb = wave0;
        if (by >= nchunk) return;
    }
    const int tl = t - lb;      // lane within the tile
    const int ytile = bx * WCOLS;
    // nv: the tile's own columns that exist (fewer than WCOLS in the last tile of a row); the two halo lanes
    // follow them directly, and waves that hold no used lane leave at once (a barrier counts only the waves
    // of the block that have not ended).
    const int nv = ny - ytile < WCOLS ? ny - ytile : WCOLS;
    if (wave0 - lb > nv + 1) return;
    // lanes past the halo lanes sit out the rest of the kernel (masked off: they would compute on nothing at
    // full power; their wave still takes part in every barrier)
    if (tl > nv + 1) return;
    const int y = tl < nv ? ytile + tl : (tl == nv ? ytile + nv : ytile - 1);
    const int cy = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
    const int yn = cy + 1 > ny - 1 ? ny - 1 : cy + 1, ys = cy - 1 < 0 ? 0 : cy - 1;
    auto lane_of = [=](int c) {         // lane that holds column c; outside the tile: this lane (own value)
        const int r = c - ytile;
        return lb + (r >= 0 && r < nv ? r : (r == nv ? nv : (r == -1 ? nv + 1 : tl)));
    };
    const int lnc = lane_of(yn), lsc = lane_of(ys);
    const bool writer = tl < nv;
    const int xs = x0 + by * ch;
    const int xe = xs + ch < nx ? xs + ch : nx;
    const int j0 = xs - 1 > xlo ? xs - 1 : xlo;
    // Rows j0 .. xe get a phase A; row xe is the East neighbour of the chunk's last row.  At the end
    // of the domain (xe == xhi + 1) it does not exist: the clamp-to-self rule makes the last row its
    // own East neighbour, and phase A of "row xe" is then simply evaluated on the clamped row index
    // (the East message does not depend on the x-slopes, so it comes out as the last row's own).
    const int j1 = xe;
    const long plane5 = (long)NF2 * ny;
    const unsigned pbytes = (unsigned)(plane5 * 8), fstride = (unsigned)ny * 8u;
    const unsigned vc = tl <= nv + 1 ? (unsigned)cy * 8u : 0xFFFFFFF0u;     // lanes past the tile load nothing
    const unsigned vst = writer ? (unsigned)y * 8u : 0xFFFFFFF0u;      // out of range: store dropped
    auto clampx = [=](int x) { return x < xlo ? xlo : (x > xhi ? xhi : x); };
    auto load_row = [&](int x, double v[NF2]) {
        const PlaneRef p(src + (long)x * plane5, pbytes);
#pragma unroll
        for (int f = 0; f < NF2; ++f) v[f] = p.ld(vc, (unsigned)f * fstride);
    };
    auto load_act = [&](int x) { return PlaneRef(active + (long)x * ny, (unsigned)ny).ldb((unsigned)cy); };
    auto load_pp = [&](int x, double pp[NF2]) {
        if (FILTER) {
            const PlaneRef q(fpp + (long)x * plane5, pbytes);
#pragma unroll
            for (int f = 0; f < NF2; ++f) pp[f] = q.ld(vc, (unsigned)f * fstride);
        }
    };

    double R[RD][NF2];                      // stored values of RD consecutive rows (own column), a ring
    double pq[4] = {0, 0, 0, 0}, pcoef[4] = {0, 0, 0, 0}, prho = 0.0;   // pending row: q (West folded in), coef, new Density
    double mwp[3] = {0, 0, 0}, mwpD = 0.0;  // message from the pending row to the row East of it
    unsigned char act_j;                    // mask of the row whose phase A comes next

    // finish row x: pending values + East message me[] + North/South messages in lmsg[x&1]
    // vs: this lane's store offset for the row, out of range where nothing is to be stored
    auto emit = [&](int x, unsigned vs, const double me[4], const double pp[NF2]) {
        const double (*L)[NT] = lmsg[x & 1];
        const double nU = stream_finish(pq[S2_U], pcoef[S2_U], me[S2_U], L[0][lnc], L[4][lsc]);
        const double nV = stream_finish(pq[S2_V], pcoef[S2_V], me[S2_V], L[1][lnc], L[5][lsc]);
        const double nT = stream_finish(pq[S2_T], pcoef[S2_T], me[S2_T], L[2][lnc], L[2][lsc]);
        const double nD = stream_finish(pq[S2_DYE], pcoef[S2_DYE], me[S2_DYE], L[3][lnc], L[3][lsc]);
        const PlaneRef d(dst + (long)x * plane5, pbytes);
        d.st_nt(vs, (unsigned)F2_RHO * fstride, prho);
        d.st_nt(vs, (unsigned)F2_T * fstride, nT);
        d.st_nt(vs, (unsigned)F2_U * fstride, nU);
        d.st_nt(vs, (unsigned)F2_V * fstride, nV);
        d.st_nt(vs, (unsigned)F2_DYE * fstride, nD);
        if (FILTER) {
            // Density and Temperature in every cell; U, V, Dye only where the cell is active
            // (coef of U is zero exactly in the cells that keep their values)
            const PlaneRef pv(fprev + (long)x * plane5, pbytes);
            const unsigned vm = pcoef[S2_U] != 0.0 ? vs : 0xFFFFFFF0u;
            pv.st_nt(vs, (unsigned)F2_RHO * fstride, pp[F2_RHO] + (0.5 * (prho - pp[F2_RHO])));
            pv.st_nt(vs, (unsigned)F2_T * fstride, pp[F2_T] + (0.5 * (nT - pp[F2_T])));
            pv.st_nt(vm, (unsigned)F2_U * fstride, pp[F2_U] + (0.5 * (nU - pp[F2_U])));
            pv.st_nt(vm, (unsigned)F2_V * fstride, pp[F2_V] + (0.5 * (nV - pp[F2_V])));
            pv.st_nt(vm, (unsigned)F2_DYE * fstride, pp[F2_DYE] + (0.5 * (nD - pp[F2_DYE])));
        }
    };
    // Row j: Rw, Rc, Re hold rows j-1, j, j+1 (clamped); lraw[j&1] holds row j of every lane.
    auto row = [&](int j, double* Rw, const double* Rc, const double* Re, bool first) {
        double hx[NF2], pp[NF2];
        x_half_slopes2(Rc, Re, Rw, hx);
        load_row(clampx(j + RD - 1), Rw);   // Rw is dead: the row RD-1 ahead goes into it
        const unsigned char act_n = load_act(clampx(j + 1));
        // Row j-1 is finished in every iteration, unconditionally, so that the stores sit
        // on the one straight-line path (the wait counts before the next row's loads are then exact);
        // for the one row per chunk that belongs to the previous chunk the stores are dropped.
        const int x = j - 1;
        load_pp(x < xs ? xs : x, pp);
#pragma unroll
        for (int f = 0; f < NF2; ++f) lraw[(j + 1) & 1][f][t] = Re[f];
        double rn[NF2], rs[NF2];
#pragma unroll
        for (int f = 0; f < NF2; ++f) {
            rn[f] = lraw[j & 1][f][lnc];
            rs[f] = lraw[j & 1][f][lsc];
        }
        StreamA2 A;
        phaseA2s(K, Rc, hx, rn, rs, mask_of(act_j != 0), A);
        if (first) {    // clamp-to-self on the West side: a real branch (the empty asm keeps it from becoming
            // per-lane conditional moves in every row); no memory operations inside.  The filter variant is
            // bound by memory and at its register limit, where the branch form costs scratch: left to the compiler.
            if (!FILTER) asm volatile("");
            mwp[0] = A.mw[0]; mwp[1] = A.mw[1]; mwp[2] = A.mw[2]; mwpD = A.me[S2_DYE];
        }
        stream_west(A.q, A.coef, mwp, mwpD);
        double (*L)[NT] = lmsg[j & 1];
        L[0][t] = A.ms[S2_U]; L[1][t] = A.ms[S2_V]; L[2][t] = A.ms[S2_T]; L[3][t] = A.ms[S2_DYE];
        L[4][t] = A.mn[S2_U]; L[5][t] = A.mn[S2_V];
        emit(x, x >= xs ? vst : 0xFFFFFFF0u, A.me, pp);
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 4; ++f) { pq[f] = A.q[f]; pcoef[f] = A.coef[f]; }
        prho = A.rho;
        mwp[0] = A.mw[0]; mwp[1] = A.mw[1]; mwp[2] = A.mw[2]; mwpD = A.me[S2_DYE];
        act_j = act_n;
    };

    // West / East messages of a row whose own values this block does not finish: the same phase A with the
    // x half-slopes left out -- nothing that feeds the two messages depends on them, and the compiler drops
    // the rest (x faces, flux divergences, q, coef, the North / South messages), about 70 % of the work.
    auto x_messages = [&](int j, const double* Rc, double me[4], double mw[3]) {
        double rn[NF2], rs[NF2];
        const double hz[NF2] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int f = 0; f < NF2; ++f) {
            rn[f] = lraw[j & 1][f][lnc];
            rs[f] = lraw[j & 1][f][lsc];
        }
        StreamA2 A;
        phaseA2s(K, Rc, hz, rn, rs, mask_of(true), A);
#pragma unroll
        for (int f = 0; f < 4; ++f) me[f] = A.me[f];
        mw[0] = A.mw[0]; mw[1] = A.mw[1]; mw[2] = A.mw[2];
    };

    int j, jfirst;
    if (LIGHT && j0 < xs) {
        // The chunk does not start the domain: row xs-1 belongs to the previous chunk and only sends its
        // message to row xs.  The ring then starts one row later.
        load_row(j0, R[0]);
        load_row(xs, R[1]);
        load_row(clampx(xs + 1), R[2]);
        act_j = load_act(xs);
#pragma unroll
        for (int f = 0; f < NF2; ++f) lraw[j0 & 1][f][t] = R[0][f];
        __syncthreads();
        double me0[4];
        x_messages(j0, R[0], me0, mwp);
        mwpD = me0[S2_DYE];
#pragma unroll
        for (int f = 0; f < NF2; ++f) lraw[xs & 1][f][t] = R[1][f];
        __syncthreads();
        j = xs;
        jfirst = j0 - 1;    // no row of this chunk is the domain's first
    } else {
#pragma unroll
        for (int u = 0; u < RD; ++u) load_row(clampx(j0 - 1 + u), R[u]);
        act_j = load_act(j0);
#pragma unroll
        for (int f = 0; f < NF2; ++f) lraw[j0 & 1][f][t] = R[1][f];
        __syncthreads();
        j = j0;
        jfirst = j0;
    }

    // the ring rotates by one row per iteration; unrolled RD times so that every index is a constant
    // All prologue loads are retired here, so that the only vector-memory operations pending at the
    // top of the loop are those of the previous iteration on either way in (exact wait counts: the
    // stores of a row stay in flight while the next row is computed).
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
    // Row j1 = xe only sends its message to row xe-1: the full rows end at jl, the last one is finished below.
    const int jl = LIGHT ? j1 - 1 : j1;
    int rot = 0;        // ring position of the next row's West values
    while (j <= jl) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            row(j, R[u % RD], R[(u + 1) % RD], R[(u + 2) % RD], j == jfirst);
            ++j;
            rot = (u + 1) % RD;
            if (j > jl) break;
        }
    }
    if (LIGHT) {
        double me1[4], mwu[3], pp[NF2];
        load_pp(xe - 1, pp);
        if (rot == 0) x_messages(j1, R[1 % RD], me1, mwu);
        else if (rot == 1) x_messages(j1, R[2 % RD], me1, mwu);
        else x_messages(j1, R[3 % RD], me1, mwu);
        emit(xe - 1, vst, me1, pp);
    }
}

// Two sweeps in one pass over memory (re-associated arithmetic, streamed messages): dst = S(S(src)),
// the second and third sweep of a step, optionally followed by the time filter.  The intermediate
// field lives only in registers and LDS, so a step moves 81 + 161 bytes per cell instead of 323.
// Pipeline per iteration j (one x-row per stage):
//   stage 1:  phase A of row j on the stored values  ->  v1(j-1), the once-swept values of row j-1
//   stage 2:  phase A of row j-2 on v1               ->  the twice-swept values of row j-3 (stored)
// Lanes: WCOLS own columns first, then the two columns above / below the tile (once-swept values are
// needed there), then the next two (they only feed stage 1's messages); the rest idle.
// Clamp-to-self at the ends of the domain applies to each sweep separately: stage 1 gets it through
// clamped row indices (a duplicated row publishes the messages of the row it duplicates); stage 2
// needs copies of v1 itself: v1(r) for r below xlo1 / above xhi1 is a copy of v1(xlo1) / v1(xhi1), and
// the first row of stage 2 takes its own West message.
//   xlo, xhi:   lowest / highest x-plane that holds stored values (ghost planes included)
//   xlo1, xhi1: lowest / highest x-plane on which once-swept values are defined
// LDS single-buffered (stored values + messages of both stages, 22 doubles per thread), two barriers
// per iteration: everything read was written in the previous iteration.
template <int NT, int MINW, int WCOLS, bool FILTER>
__global__ void __launch_bounds__(NT, MINW)
sweep2d_pair(FastConsts K, const double* __restrict__ src, double* __restrict__ dst,
             const uint8_t* __restrict__ active, int x0, int nx, int ny, int xlo, int xhi, int xlo1, int xhi1,
             int ch, const double* __restrict__ fpp, double* __restrict__ fprev) {
    static_assert(WCOLS + 4 <= NT, "block too small for the tile and its two halo columns per side");
    __shared__ double lraw1[NF2][NT], lmsg1[6][NT], lraw2[NF2][NT], lmsg2[6][NT];
    const int t = threadIdx.x;
    const int nchunk = (nx - x0 + ch - 1) / ch;
    const TilePlan tp = tile_plan(ny, WCOLS, NT, 4, nchunk);
    const unsigned wg = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int wave0 = (int)(__builtin_amdgcn_readfirstlane(t) & ~63);      // first lane of this wave (scalar)
    int bx, by, lb = 0;
    if (wg < (unsigned)(tp.nfull * nchunk)) {
        bx = (int)(wg % (unsigned)tp.nfull);
        by = (int)(wg / (unsigned)tp.nfull);
    } else {        // packed block: this wave is the row's narrow last tile for one chunk (see tile_plan)
        bx = tp.nfull;
        by = (int)(wg - (unsigned)(tp.nfull * nchunk)) * tp.pack + (wave0 >> 6);
        lb = wave0;
        if (by >= nchunk) return;
    }
    const int tl = t - lb;      // lane within the tile
    const int ytile = bx * WCOLS;
    // nv: the tile's own columns that exist; the four halo lanes follow them, and waves without a used lane
    // leave at once (see sweep2d_stream)
    const int nv = ny - ytile < WCOLS ? ny - ytile : WCOLS;
    if (wave0 - lb > nv + 3) return;
    if (tl > nv + 3) return;    // idle lanes are masked off for the rest of the kernel (see sweep2d_stream)
    // lanes: own columns, then +1 above, -1 below, +2 above, -2 below
    const int y = tl < nv ? ytile + tl
                          : (tl == nv ? ytile + nv : (tl == nv + 1 ? ytile - 1 : (tl == nv + 2 ? ytile + nv + 1 : ytile - 2)));
    const int cy = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
    const int yn = cy + 1 > ny - 1 ? ny - 1 : cy + 1, ys = cy - 1 < 0 ? 0 : cy - 1;
    auto lane_of = [=](int c) {         // lane that holds column c; outside the tile and its halo: this lane
        const int r = c - ytile;
        if (r >= 0 && r < nv) return lb + r;
        if (r == nv) return lb + nv;
        if (r == -1) return lb + nv + 1;
        if (r == nv + 1) return lb + nv + 2;
        if (r == -2) return lb + nv + 3;
        return t;
    };
    const int lnc = lane_of(yn), lsc = lane_of(ys);
    const bool writer = tl < nv;
    const int xs = x0 + by * ch;
    const int xe = xs + ch < nx ? xs + ch : nx;
    const long plane5 = (long)NF2 * ny;
    const unsigned pbytes = (unsigned)(plane5 * 8), fstride = (unsigned)ny * 8u;
    const unsigned vc = tl <= nv + 3 ? (unsigned)cy * 8u : 0xFFFFFFF0u;     // idle lanes load nothing
    const unsigned vst = writer ? (unsigned)y * 8u : 0xFFFFFFF0u;
    auto clampx = [=](int x) { return x < xlo ? xlo : (x > xhi ? xhi : x); };
    auto load_row = [&](int x, double v[NF2]) {
        const PlaneRef p(src + (long)x * plane5, pbytes);
#pragma unroll
        for (int f = 0; f < NF2; ++f) v[f] = p.ld(vc, (unsigned)f * fstride);
    };
    auto load_act = [&](int x) { return PlaneRef(active + (long)x * ny, (unsigned)ny).ldb((unsigned)cy); };

    double R[3][NF2];                       // stored values: rows j-1, j, j+1
    double V[3][NF2];                       // once-swept values: rows j-3, j-2, j-1
    double q1[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0}, rho1 = 1.0, mw1[3] = {0, 0, 0}, mw1D = 0.0;   // stage 1 pending row j-1
    double q2[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0}, rho2 = 1.0, mw2[3] = {0, 0, 0}, mw2D = 0.0;   // stage 2 pending row j-3
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int f = 0; f < NF2; ++f) V[u][f] = 1.0;
    const int a2first = xs - 1 > xlo1 ? xs - 1 : xlo1;      // first row whose stage-2 phase A is used

    auto iter = [&](int j, double* Rw, const double* Rc, const double* Re, double* Vw, double* Vc, double* Ve) {
        // ---- stage 1: phase A of row j, then finish row j-1
        double hx[NF2];
        x_half_slopes2<true>(Rc, Re, Rw, hx);
        load_row(clampx(j + 2), Rw);                        // Rw is dead: row j+2 goes into it
        const unsigned char act1 = load_act(clampx(j)), act2 = load_act(clampx(j - 2));
        double pp[NF2];
        const int xo = j - 3;                                // the row stage 2 finishes in this iteration
        if (FILTER) {
            const PlaneRef q(fpp + (long)(xo < xs ? xs : xo) * plane5, pbytes);
#pragma unroll
            for (int f = 0; f < NF2; ++f) pp[f] = q.ld(vc, (unsigned)f * fstride);
        }
        double rn[NF2], rs[NF2];
#pragma unroll
        for (int f = 0; f < NF2; ++f) { rn[f] = lraw1[f][lnc]; rs[f] = lraw1[f][lsc]; }
        StreamA2 A;
        phaseA2s<SelPlain>(K, Rc, hx, rn, rs, act1 != 0, A);
        stream_west(A.q, A.coef, mw1, mw1D);
        double vnew[NF2];                                    // v1(j-1)
        vnew[F2_U] = stream_finish(q1[S2_U], c1[S2_U], A.me[S2_U], lmsg1[0][lnc], lmsg1[4][lsc]);
        vnew[F2_V] = stream_finish(q1[S2_V], c1[S2_V], A.me[S2_V], lmsg1[1][lnc], lmsg1[5][lsc]);
        vnew[F2_T] = stream_finish(q1[S2_T], c1[S2_T], A.me[S2_T], lmsg1[2][lnc], lmsg1[2][lsc]);
        vnew[F2_DYE] = stream_finish(q1[S2_DYE], c1[S2_DYE], A.me[S2_DYE], lmsg1[3][lnc], lmsg1[3][lsc]);
        vnew[F2_RHO] = rho1;
        const int r1 = j - 1;
        // The three domain-end rules below depend on the row index only, so they are real (scalar) branches
        // around plain copies: the empty asm keeps the compiler from turning them into per-lane conditional
        // moves, which every row of every chunk would pay for.
        if (r1 > xhi1) {                             // above the domain: a copy of the last defined row
            asm volatile("");
#pragma unroll
            for (int f = 0; f < NF2; ++f) vnew[f] = Vc[f];
        }
#pragma unroll
        for (int f = 0; f < NF2; ++f) Ve[f] = vnew[f];
        if (r1 == xlo1) {                            // below the domain: copies of the first defined row
            asm volatile("");
#pragma unroll
            for (int f = 0; f < NF2; ++f) { Vc[f] = vnew[f]; Vw[f] = vnew[f]; }
        }
        // ---- stage 2: phase A of row j-2 on the once-swept values, then finish row j-3
        double hx2[NF2], rn2[NF2], rs2[NF2];
        x_half_slopes2<true>(Vc, Ve, Vw, hx2);
#pragma unroll
        for (int f = 0; f < NF2; ++f) { rn2[f] = lraw2[f][lnc]; rs2[f] = lraw2[f][lsc]; }
        StreamA2 B;
        phaseA2s<SelPlain>(K, Vc, hx2, rn2, rs2, act2 != 0, B);
        if (j - 2 == a2first) {     // clamp-to-self on the West side of the second sweep
            asm volatile("");
            mw2[0] = B.mw[0]; mw2[1] = B.mw[1]; mw2[2] = B.mw[2]; mw2D = B.me[S2_DYE];
        }
        stream_west(B.q, B.coef, mw2, mw2D);
        {
            const unsigned vs = (xo >= xs && xo < xe) ? vst : 0xFFFFFFF0u;
            const double nU = stream_finish(q2[S2_U], c2[S2_U], B.me[S2_U], lmsg2[0][lnc], lmsg2[4][lsc]);
            const double nV = stream_finish(q2[S2_V], c2[S2_V], B.me[S2_V], lmsg2[1][lnc], lmsg2[5][lsc]);
            const double nT = stream_finish(q2[S2_T], c2[S2_T], B.me[S2_T], lmsg2[2][lnc], lmsg2[2][lsc]);
            const double nD = stream_finish(q2[S2_DYE], c2[S2_DYE], B.me[S2_DYE], lmsg2[3][lnc], lmsg2[3][lsc]);
            const int xr = xo < xs ? xs : (xo > xe - 1 ? xe - 1 : xo);
            const PlaneRef d(dst + (long)xr * plane5, pbytes);
            d.st_nt(vs, (unsigned)F2_RHO * fstride, rho2);
            d.st_nt(vs, (unsigned)F2_T * fstride, nT);
            d.st_nt(vs, (unsigned)F2_U * fstride, nU);
            d.st_nt(vs, (unsigned)F2_V * fstride, nV);
            d.st_nt(vs, (unsigned)F2_DYE * fstride, nD);
            if (FILTER) {
                const PlaneRef pv(fprev + (long)xr * plane5, pbytes);
                const unsigned vm = c2[S2_U] != 0.0 ? vs : 0xFFFFFFF0u;
                pv.st_nt(vs, (unsigned)F2_RHO * fstride, pp[F2_RHO] + (0.5 * (rho2 - pp[F2_RHO])));
                pv.st_nt(vs, (unsigned)F2_T * fstride, pp[F2_T] + (0.5 * (nT - pp[F2_T])));
                pv.st_nt(vm, (unsigned)F2_U * fstride, pp[F2_U] + (0.5 * (nU - pp[F2_U])));
                pv.st_nt(vm, (unsigned)F2_V * fstride, pp[F2_V] + (0.5 * (nV - pp[F2_V])));
                pv.st_nt(vm, (unsigned)F2_DYE * fstride, pp[F2_DYE] + (0.5 * (nD - pp[F2_DYE])));
            }
        }
        __syncthreads();            // every read of the previous iteration's tiles is done
#pragma unroll
        for (int f = 0; f < NF2; ++f) { lraw1[f][t] = Re[f]; lraw2[f][t] = vnew[f]; }
        lmsg1[0][t] = A.ms[S2_U]; lmsg1[1][t] = A.ms[S2_V]; lmsg1[2][t] = A.ms[S2_T]; lmsg1[3][t] = A.ms[S2_DYE];
        lmsg1[4][t] = A.mn[S2_U]; lmsg1[5][t] = A.mn[S2_V];
        lmsg2[0][t] = B.ms[S2_U]; lmsg2[1][t] = B.ms[S2_V]; lmsg2[2][t] = B.ms[S2_T]; lmsg2[3][t] = B.ms[S2_DYE];
        lmsg2[4][t] = B.mn[S2_U]; lmsg2[5][t] = B.mn[S2_V];
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 4; ++f) { q1[f] = A.q[f]; c1[f] = A.coef[f]; q2[f] = B.q[f]; c2[f] = B.coef[f]; }
        rho1 = A.rho; rho2 = B.rho;
        mw1[0] = A.mw[0]; mw1[1] = A.mw[1]; mw1[2] = A.mw[2]; mw1D = A.me[S2_DYE];
        mw2[0] = B.mw[0]; mw2[1] = B.mw[1]; mw2[2] = B.mw[2]; mw2D = B.me[S2_DYE];
    };

    // The first three iterations (rows xs-2, xs-1, xs of stage 1): stage 2 would work on rows before a2first, of
    // which nothing is used, so these iterations run stage 1 only.  Three iterations bring the rings back to their
    // starting rotation.  Same LDS protocol and barriers as iter.
    auto iter_stage1 = [&](int j, double* Rw, const double* Rc, const double* Re, double* Vw, double* Vc, double* Ve) {
        double hx[NF2];
        x_half_slopes2<true>(Rc, Re, Rw, hx);
        load_row(clampx(j + 2), Rw);
        const unsigned char act1 = load_act(clampx(j));
        double rn[NF2], rs[NF2];
#pragma unroll
        for (int f = 0; f < NF2; ++f) { rn[f] = lraw1[f][lnc]; rs[f] = lraw1[f][lsc]; }
        StreamA2 A;
        phaseA2s<SelPlain>(K, Rc, hx, rn, rs, act1 != 0, A);
        stream_west(A.q, A.coef, mw1, mw1D);
        double vnew[NF2];
        vnew[F2_U] = stream_finish(q1[S2_U], c1[S2_U], A.me[S2_U], lmsg1[0][lnc], lmsg1[4][lsc]);
        vnew[F2_V] = stream_finish(q1[S2_V], c1[S2_V], A.me[S2_V], lmsg1[1][lnc], lmsg1[5][lsc]);
        vnew[F2_T] = stream_finish(q1[S2_T], c1[S2_T], A.me[S2_T], lmsg1[2][lnc], lmsg1[2][lsc]);
        vnew[F2_DYE] = stream_finish(q1[S2_DYE], c1[S2_DYE], A.me[S2_DYE], lmsg1[3][lnc], lmsg1[3][lsc]);
        vnew[F2_RHO] = rho1;
        const int r1 = j - 1;
        if (r1 > xhi1) {
#pragma unroll
            for (int f = 0; f < NF2; ++f) vnew[f] = Vc[f];
        }
#pragma unroll
        for (int f = 0; f < NF2; ++f) Ve[f] = vnew[f];
        if (r1 == xlo1) {
#pragma unroll
            for (int f = 0; f < NF2; ++f) { Vc[f] = vnew[f]; Vw[f] = vnew[f]; }
        }
        __syncthreads();
#pragma unroll
        for (int f = 0; f < NF2; ++f) { lraw1[f][t] = Re[f]; lraw2[f][t] = vnew[f]; }
        lmsg1[0][t] = A.ms[S2_U]; lmsg1[1][t] = A.ms[S2_V]; lmsg1[2][t] = A.ms[S2_T]; lmsg1[3][t] = A.ms[S2_DYE];
        lmsg1[4][t] = A.mn[S2_U]; lmsg1[5][t] = A.mn[S2_V];
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 4; ++f) { q1[f] = A.q[f]; c1[f] = A.coef[f]; }
        rho1 = A.rho;
        mw1[0] = A.mw[0]; mw1[1] = A.mw[1]; mw1[2] = A.mw[2]; mw1D = A.me[S2_DYE];
    };

    const int jb = xs - 2, je = xe + 2;     // stage-1 rows; their indices are clamped where they are used
#pragma unroll
    for (int u = 0; u < 3; ++u) load_row(clampx(jb - 1 + u), R[u]);
#pragma unroll
    for (int f = 0; f < NF2; ++f) { lraw1[f][t] = R[1][f]; lraw2[f][t] = 1.0; }
    for (int e = 0; e < 6; ++e) { lmsg1[e][t] = 0.0; lmsg2[e][t] = 0.0; }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): only the previous iteration's operations are pending in the loop
    int j = jb;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        iter_stage1(j, R[u % 3], R[(u + 1) % 3], R[(u + 2) % 3], V[u % 3], V[(u + 1) % 3], V[(u + 2) % 3]);
        ++j;
    }
    while (j <= je) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            iter(j, R[u % 3], R[(u + 1) % 3], R[(u + 2) % 3], V[u % 3], V[(u + 1) % 3], V[(u + 2) % 3]);
            ++j;
            if (j > je) break;
        }
    }
}

// A whole step in one pass over memory (re-associated arithmetic, streamed messages): dst = S(S(S(src))), the three
// sweeps of runSingleStep, optionally followed by the time filter.  The two intermediate fields live only in LDS.
// A step then moves 81 + 80 bytes per cell (source + mask + filter input read, new values + filtered values written)
// instead of 81 + 161.
//
// One group of NTS threads per sweep ("stage"): stage 0 reads the stored values from memory, stage 1 works on the
// once-swept values, stage 2 on the twice-swept values and stores.  Every stage runs the row loop of sweep2d_stream
// on its own input, so a thread carries one sweep's state only (the register budget of the single sweep, three waves
// per SIMD), and the block's 3 * NTS / 64 waves put one wave of every stage on each SIMD.  A stage hands a finished
// row to the next one through a ring of three rows in LDS: written in iteration i, read by the consumer in
// iterations i+1 (as the row East of its centre row; own column) and i+2 (as its centre row: the North / South
// neighbours' values), overwritten in i+3.  One barrier per iteration, all stages in lock-step; iteration i has
// stage s evaluate row xs - 3 + i - 3 s of its input and finish the row before it.
//
// Lanes of a stage group: the tile's own columns first, then the columns +1, -1, +2, -2, +3, -3 beside the tile
// (stage 0 needs all six, stage 1 the first four, stage 2 the first two: the outermost pair of a stage only sends
// messages and stored values inwards); further lanes idle.  Tiles: widths are multiples of 8 columns (64 bytes), as
// equal as possible (trio_tiles) and up to 248 of a 256-lane stage: 4096 columns are 17 tiles of 240 or 248.  (Whole
// 128-byte lines per tile -- multiples of 16, at most 240 -- need an 18th tile there and take 3 % longer; widths that
// are not a multiple of 8 are no faster than that.)
// Clamp-to-self at the ends of the domain applies to each sweep separately; a stage gets it, as sweep2d_stream
// does, from clamped row indices into its input (the producer writes only rows on which its output is defined):
//   stage s reads input rows lo_s .. hi_s,  lo_s = min(lo0 + s, 0),  hi_s = max(hi0 - s, nx - 1).
#ifndef CFD_TRIO_PRIO
#define CFD_TRIO_PRIO 1
#endif
static constexpr bool TRIO_PRIO = CFD_TRIO_PRIO != 0;
struct TrioTiles { int ntile, lines, gran; };
__host__ __device__ inline TrioTiles trio_tiles(int ny, int wmax, int gran) {
    const int lines = (ny + gran - 1) / gran, per = wmax / gran;
    return TrioTiles{(lines + per - 1) / per, lines, gran};
}
__host__ __device__ inline int trio_tile_start(const TrioTiles& tp, int i) {
    return (int)(((long)i * tp.lines) / tp.ntile) * tp.gran;
}

// LDS of the three-sweeps kernel, in rows of NTS doubles: messages [3 stages][2][6], stored values of stage 0's
// current and next row [2][NF2], rings of once- and twice-swept rows [3][NF2] each.
static constexpr int TRIO_LDS_ROWS = 36 + 2 * NF2 + 2 * 3 * NF2;

// Issue priority by progress through the row (see NoPhaseHook): 3 at the top of a row, down to 0 at its end.
struct TrioPrioHook {
    __device__ __forceinline__ static void at(int k) {
        if (k == 1) __builtin_amdgcn_s_setprio(2);
        else if (k == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
};

template <int STAGE, int NTS, bool FILTER>
__device__ __forceinline__ void trio_stage(const FastConsts& K, const double* __restrict__ src, double* __restrict__ dst,
                                           const uint8_t* __restrict__ active, int nx, int ny, int lo0, int hi0, int xs,
                                           int xe, int ytile, int nv, int tl, int wave0, const double* __restrict__ fpp,
                                           double* __restrict__ fprev, double* __restrict__ smem) {
    // waves without a used lane leave at once (a barrier counts only the waves that have not ended); unused lanes
    // of the other waves are masked off for the rest of the kernel
    constexpr int HALO = 6 - 2 * STAGE;
    if (wave0 > nv + HALO - 1) return;
    double (*const lmsg)[6][NTS] = reinterpret_cast<double (*)[6][NTS]>(smem + (size_t)STAGE * 12 * NTS);
    double (*const raw0)[NF2][NTS] = reinterpret_cast<double (*)[NF2][NTS]>(smem + (size_t)36 * NTS);
    double* const ring1 = smem + (size_t)(36 + 2 * NF2) * NTS;
    double* const ring2 = ring1 + (size_t)3 * NF2 * NTS;
    const double* const ring_in = STAGE == 1 ? ring1 : ring2;      // (stage 0 reads memory)
    double* const ring_out = STAGE == 0 ? ring1 : ring2;           // (stage 2 stores to memory)
    // what this stage publishes starts out defined, in the slots of its unused lanes too (an outermost column reads
    // them as its outer neighbour's; nothing it hands inwards depends on them)
#pragma unroll
    for (int e = 0; e < 6; ++e) { lmsg[0][e][tl] = 0.0; lmsg[1][e][tl] = 0.0; }
    if constexpr (STAGE < 2) {
#pragma unroll
        for (int u = 0; u < 3 * NF2; ++u) ring_out[(size_t)u * NTS + tl] = 1.0;
    }
    if (tl > nv + HALO - 1) return;
    // lane -> column: own columns, then +1 above, -1 below, +2, -2, +3, -3
    const int k = tl - nv;
    const int y = k < 0 ? ytile + tl : ((k & 1) ? ytile - 1 - (k >> 1) : ytile + nv + (k >> 1));
    const int cy = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
    const int yn = cy + 1 > ny - 1 ? ny - 1 : cy + 1, ys = cy - 1 < 0 ? 0 : cy - 1;
    auto lane_of = [=](int c) {         // lane that holds column c; beyond the tile's halo: this lane (own value)
        const int r = c - ytile;
        if (r >= 0 && r < nv) return r;
        if (r >= nv && r < nv + 3) return nv + 2 * (r - nv);
        if (r < 0 && r >= -3) return nv + 2 * (-r - 1) + 1;
        return tl;
    };
    const int lnc = lane_of(yn), lsc = lane_of(ys);
    const bool writer = tl < nv;
    const long plane5 = (long)NF2 * ny;
    const unsigned pbytes = (unsigned)(plane5 * 8), fstride = (unsigned)ny * 8u;
    const unsigned vc = (unsigned)cy * 8u;
    const unsigned vst = writer ? (unsigned)y * 8u : 0xFFFFFFF0u;       // out of range: store dropped
    // rows this stage's input is defined on
    const int lo = lo0 + STAGE > 0 ? 0 : lo0 + STAGE, hi = hi0 - STAGE < nx - 1 ? nx - 1 : hi0 - STAGE;
    // rows that get a phase A: the first and the last one only send their East / West message
    const int afirst = xs - 3 + STAGE > lo ? xs - 3 + STAGE : lo;
    const int alast = xe + 2 - STAGE < hi + 1 ? xe + 2 - STAGE : hi + 1;
    const int nit = xe - xs + 10;           // iterations of the block, the same for every stage
    auto clampr = [=](int r) { return r < lo ? lo : (r > hi ? hi : r); };
    auto slot3 = [](int r) { return (int)((unsigned)(r + 3072) % 3u); };
    auto load_row = [&](int x, double v[NF2]) {
        const PlaneRef p(src + (long)x * plane5, pbytes);
#pragma unroll
        for (int f = 0; f < NF2; ++f) v[f] = p.ld(vc, (unsigned)f * fstride);
    };
    auto ring_row = [&](int r, double v[NF2]) {
        const double* p = ring_in + (size_t)slot3(clampr(r)) * (NF2 * NTS) + tl;
#pragma unroll
        for (int f = 0; f < NF2; ++f) v[f] = p[f * NTS];
    };
    auto load_act = [&](int x) { return PlaneRef(active + (long)x * ny, (unsigned)ny).ldb((unsigned)cy); };

    double R[3][NF2];
    double pq[4] = {0, 0, 0, 0}, pcoef[4] = {0, 0, 0, 0}, prho = 0.0;   // pending row: q (West folded in), coef, new Density
    double mwp[3] = {0, 0, 0}, mwpD = 0.0;                              // message from the pending row to the row East of it
    unsigned char act_j = load_act(afirst);

    // Row a: phase A of input row a, then finish row a - 1.  Rw, Rc, Re: input rows a-1, a, a+1 (clamped).
    auto row = [&](int a, double* Rw, double* Rc, double* Re, bool first) {
        double rn[NF2], rs[NF2], hx[NF2], pp[NF2];
        if (TRIO_PRIO) __builtin_amdgcn_s_setprio(3);
        if constexpr (STAGE > 0) {
            if (first) {
                // The row before the first: at the start of the domain the clamp-to-self rule makes it the first row
                // itself; elsewhere the first row only sends its East message, which does not depend on the x slopes.
                asm volatile("");
                ring_row(a, Rc);
#pragma unroll
                for (int f = 0; f < NF2; ++f) Rw[f] = Rc[f];
            }
            ring_row(a + 1, Re);
        }
        x_half_slopes2(Rc, Re, Rw, hx);
        if constexpr (STAGE == 0) {
            load_row(clampr(a + 2), Rw);        // Rw is dead: the row two ahead goes into it
#pragma unroll
            for (int f = 0; f < NF2; ++f) raw0[(a + 1) & 1][f][tl] = Re[f];
#pragma unroll
            for (int f = 0; f < NF2; ++f) { rn[f] = raw0[a & 1][f][lnc]; rs[f] = raw0[a & 1][f][lsc]; }
        } else {
            const double* cen = ring_in + (size_t)slot3(clampr(a)) * (NF2 * NTS);
#pragma unroll
            for (int f = 0; f < NF2; ++f) { rn[f] = cen[f * NTS + lnc]; rs[f] = cen[f * NTS + lsc]; }
        }
        const unsigned char act_n = load_act(clampr(a + 1));
        const int r = a - 1;
        const int rr = r < xs ? xs : r;         // a row the planes' descriptors can be built on
        if constexpr (STAGE == 2 && FILTER) {
            const PlaneRef q(fpp + (long)rr * plane5, pbytes);
#pragma unroll
            for (int f = 0; f < NF2; ++f) pp[f] = q.ld(vc, (unsigned)f * fstride);
        }
        StreamA2 A;
        if (TRIO_PRIO) phaseA2s<SelMask, TrioPrioHook>(K, Rc, hx, rn, rs, mask_of(act_j != 0), A);
        else phaseA2s(K, Rc, hx, rn, rs, mask_of(act_j != 0), A);
        if (first) {        // clamp-to-self on the West side (a real branch: see sweep2d_stream)
            asm volatile("");
            mwp[0] = A.mw[0]; mwp[1] = A.mw[1]; mwp[2] = A.mw[2]; mwpD = A.me[S2_DYE];
        }
        stream_west(A.q, A.coef, mwp, mwpD);
        double (*L)[NTS] = lmsg[a & 1];
        L[0][tl] = A.ms[S2_U]; L[1][tl] = A.ms[S2_V]; L[2][tl] = A.ms[S2_T]; L[3][tl] = A.ms[S2_DYE];
        L[4][tl] = A.mn[S2_U]; L[5][tl] = A.mn[S2_V];
        // finish row r: pending values + East message + North / South messages of row r
        const double (*P)[NTS] = lmsg[r & 1];
        const double nU = stream_finish(pq[S2_U], pcoef[S2_U], A.me[S2_U], P[0][lnc], P[4][lsc]);
        const double nV = stream_finish(pq[S2_V], pcoef[S2_V], A.me[S2_V], P[1][lnc], P[5][lsc]);
        const double nT = stream_finish(pq[S2_T], pcoef[S2_T], A.me[S2_T], P[2][lnc], P[2][lsc]);
        const double nD = stream_finish(pq[S2_DYE], pcoef[S2_DYE], A.me[S2_DYE], P[3][lnc], P[3][lsc]);
        if constexpr (STAGE < 2) {
            // Written unconditionally (straight-line code keeps the row's registers short-lived).  A row on which this
            // stage's output is not defined -- the one before the first, one beyond the domain's end -- lands in the
            // ring slot of a row three further on or back, which the consumer is not reading at that time: the slot
            // is rewritten before row r + 3 is read, and row r - 3 was last read an iteration earlier.
            double* o = ring_out + (size_t)slot3(r) * (NF2 * NTS) + tl;
            o[F2_U * NTS] = nU; o[F2_V * NTS] = nV; o[F2_RHO * NTS] = prho; o[F2_T * NTS] = nT; o[F2_DYE * NTS] = nD;
        } else {
            const unsigned vs = (!first && r >= xs) ? vst : 0xFFFFFFF0u;
            const PlaneRef d(dst + (long)rr * plane5, pbytes);
            d.st_nt(vs, (unsigned)F2_RHO * fstride, prho);
            d.st_nt(vs, (unsigned)F2_T * fstride, nT);
            d.st_nt(vs, (unsigned)F2_U * fstride, nU);
            d.st_nt(vs, (unsigned)F2_V * fstride, nV);
            d.st_nt(vs, (unsigned)F2_DYE * fstride, nD);
            if (FILTER) {
                // Density and Temperature filtered in every cell; U, V, Dye where the cell is active (coef of U is
                // zero exactly in the cells that keep their values) and otherwise the kept value itself: the slot
                // that receives the filtered values is a buffer of its own here, not the slot the step started from
                const PlaneRef pv(fprev + (long)rr * plane5, pbytes);
                const lane_mask am = mask_of(pcoef[S2_U] != 0.0);
                pv.st_nt(vs, (unsigned)F2_RHO * fstride, pp[F2_RHO] + (0.5 * (prho - pp[F2_RHO])));
                pv.st_nt(vs, (unsigned)F2_T * fstride, pp[F2_T] + (0.5 * (nT - pp[F2_T])));
                pv.st_nt(vs, (unsigned)F2_U * fstride, select(am, pp[F2_U] + (0.5 * (nU - pp[F2_U])), nU));
                pv.st_nt(vs, (unsigned)F2_V * fstride, select(am, pp[F2_V] + (0.5 * (nV - pp[F2_V])), nV));
                pv.st_nt(vs, (unsigned)F2_DYE * fstride, select(am, pp[F2_DYE] + (0.5 * (nD - pp[F2_DYE])), nD));
            }
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) { pq[f] = A.q[f]; pcoef[f] = A.coef[f]; }
        prho = A.rho;
        mwp[0] = A.mw[0]; mwp[1] = A.mw[1]; mwp[2] = A.mw[2]; mwpD = A.me[S2_DYE];
        act_j = act_n;
    };

    if constexpr (STAGE == 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) load_row(clampr(afirst - 1 + u), R[u]);
#pragma unroll
        for (int f = 0; f < NF2; ++f) { raw0[afirst & 1][f][tl] = R[1][f]; raw0[(afirst + 1) & 1][f][tl] = R[1][f]; }
    }
    __syncthreads();
    if constexpr (STAGE == 0) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): see sweep2d_stream
    int done = 0;
    const int lead = afirst - (xs - 3 - 3 * STAGE);     // iterations before this stage's first row
    for (; done < lead; ++done) __syncthreads();
    int a = afirst;
    if (a <= alast) {
        row(a, R[0], R[1], R[2], true);
        __syncthreads();
        ++a; ++done;
        while (a <= alast) {
            row(a, R[1], R[2], R[0], false);
            __syncthreads();
            ++a; ++done;
            if (a > alast) break;
            row(a, R[2], R[0], R[1], false);
            __syncthreads();
            ++a; ++done;
            if (a > alast) break;
            row(a, R[0], R[1], R[2], false);
            __syncthreads();
            ++a; ++done;
        }
    }
    for (; done < nit; ++done) __syncthreads();
}

template <int NTS, int MINW, bool FILTER>
__global__ void __launch_bounds__(3 * NTS, MINW)
sweep2d_trio(FastConsts K, const double* __restrict__ src, double* __restrict__ dst, const uint8_t* __restrict__ active,
             int nx, int ny, int lo0, int hi0, int ch, int x_first, int by0, int n_first, int by1, int wmax, int gran,
             int busy, int xcd_cap, int xcd_off, const double* __restrict__ fpp, double* __restrict__ fprev) {
    extern __shared__ double smem[];
    const int t = threadIdx.x;
    const int first_lane = (int)__builtin_amdgcn_readfirstlane(t) & ~63;     // first lane of this wave (scalar)
    const int stage = first_lane / NTS;
    const int tl = t - stage * NTS, wave0 = first_lane - stage * NTS;
    const TrioTiles tp = trio_tiles(ny, wmax, gran);
    // busy > 0: this launch starts beside one of `busy` blocks and takes the compute units that one leaves (xcd_fill_rank)
    int placed = 0;
    if (busy > 0) {
        placed = xcd_fill_rank((int)blockIdx.x, busy, xcd_cap, xcd_off & 7, (xcd_off >> 4) & 7);
        if (placed < 0) return;
    }
    const unsigned wg = busy > 0 ? (unsigned)placed : xcd_contiguous_id(blockIdx.x, gridDim.x);
    // the launch covers the chunks by0 .. by0 + n_first - 1 and, after them, by1 .. (a slab launches the chunks next to
    // its two neighbours together)
    const int bl = (int)(wg / (unsigned)tp.ntile);
    const int bx = (int)(wg % (unsigned)tp.ntile), by = bl < n_first ? by0 + bl : by1 + (bl - n_first);
    const int ytile = trio_tile_start(tp, bx);
    const int yend = bx + 1 < tp.ntile ? trio_tile_start(tp, bx + 1) : ny;
    const int nv = (yend < ny ? yend : ny) - ytile;
    // chunk 0 holds x_first planes, the others ch (the last one what is left)
    const int xs = by == 0 ? 0 : x_first + (by - 1) * ch;
    const int xe = by == 0 ? (x_first < nx ? x_first : nx) : (xs + ch < nx ? xs + ch : nx);
    if (stage == 0)
        trio_stage<0, NTS, FILTER>(K, src, dst, active, nx, ny, lo0, hi0, xs, xe, ytile, nv, tl, wave0, fpp, fprev, smem);
    else if (stage == 1)
        trio_stage<1, NTS, FILTER>(K, src, dst, active, nx, ny, lo0, hi0, xs, xe, ytile, nv, tl, wave0, fpp, fprev, smem);
    else
        trio_stage<2, NTS, FILTER>(K, src, dst, active, nx, ny, lo0, hi0, xs, xe, ytile, nv, tl, wave0, fpp, fprev, smem);
}

// Tiled 3-D sweep.  A block owns a (TY-2) x (TZ-2) patch of (y, z) columns plus a one-cell halo
// ring (lanes run along z, the contiguous index) and walks a chunk of x-planes.  As in the 2-D
// tiled kernel each thread keeps its column's stored values and phase-A results for three
// consecutive rows in registers, so East/West neighbours never leave the thread.  The North /
// South / Top / Bottom neighbours' stored values and their y- and z-products cross lanes through
// LDS (single-buffered, two barriers per row: reads of row j-1's products and row j's stored
// values, barrier, writes of row j's products and row j+1's stored values, barrier).  Halo
// threads have no neighbour on their outer side and use their own value there: what they publish
// towards the patch does not depend on the slopes along that direction, and they store nothing.
// LDS (dynamic): NF3 + NSY + NSZ doubles per thread (+ 4 * NE3 where the policy parks left-hand sides there).
// FILTER: the sweep that ends a step also applies the time filter to the values it has just computed.
template <class M, int TY, int TZ, int MINW, bool FILTER = false>
__global__ void __launch_bounds__(TY * TZ, MINW)
sweep3d_march(typename M::Consts K, const double* __restrict__ src, double* __restrict__ dst,
              const uint8_t* __restrict__ active, int x0, int nx, int ny, int nz, int xlo, int xhi, int ch,
              const double* __restrict__ fpp, double* __restrict__ fprev) {
    constexpr int NT = TY * TZ;
    constexpr int NSY = M::NSY;
    extern __shared__ double smem[];
    // Row stride of the LDS arrays: one slot more than the block, so that two rows are never a multiple of 512 bytes
    // apart and the compiler cannot pair their reads into ds_read2st64_b64, which moves half as many bytes per LDS
    // cycle as two ds_read_b64 do.
    constexpr int LS = NT + 1;
    double* const lraw = smem;                      // [NF3][LS]
    double* const lsy = smem + NF3 * LS;            // [NSY][LS]
    double* const lsz = lsy + NSY * LS;             // [NSZ][LS]
    double* const lown = lsz + M::NSZ * LS;         // [2][2 * NE3][LS], OWN_FROM_LDS only: see row()
    const int t = threadIdx.x, tz = t % TZ, ty = t / TZ;
    const int zbase = (int)blockIdx.x * (TZ - 2) - 1, ybase = (int)blockIdx.y * (TY - 2) - 1;
    const int y = ybase + ty, z = zbase + tz;
    const int cy = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y), cz = z < 0 ? 0 : (z > nz - 1 ? nz - 1 : z);
    // thread that holds column (yy, zz); outside the tile: this thread itself
    auto lane_of = [=](int yy, int zz) {
        const int ly = yy - ybase, lz = zz - zbase;
        return (ly >= 0 && ly < TY && lz >= 0 && lz < TZ) ? ly * TZ + lz : t;
    };
    const int lN = lane_of(cy + 1 > ny - 1 ? ny - 1 : cy + 1, cz), lS = lane_of(cy - 1 < 0 ? 0 : cy - 1, cz);
    const int lT = lane_of(cy, cz + 1 > nz - 1 ? nz - 1 : cz + 1), lB = lane_of(cy, cz - 1 < 0 ? 0 : cz - 1);
    const bool writer = ty >= 1 && ty <= TY - 2 && tz >= 1 && tz <= TZ - 2 && y < ny && z < nz;
    const int xs = x0 + (int)blockIdx.z * ch;
    const int xe = xs + ch < nx ? xs + ch : nx;
    const int j0 = xs - 1 > xlo ? xs - 1 : xlo;
    const int j1 = xe < xhi ? xe : xhi;
    const long plane = (long)ny * nz, plane6 = (long)NF3 * plane;
    const unsigned pbytes = (unsigned)(plane6 * 8), fstride = (unsigned)(plane * 8);
    const unsigned cell = (unsigned)cy * (unsigned)nz + (unsigned)cz;
    const unsigned vc = cell * 8u;
    const unsigned vst = writer ? vc : 0xFFFFFFF0u;      // out of range: store dropped
    auto clampx = [=](int x) { return x < xlo ? xlo : (x > xhi ? xhi : x); };
    auto load_row = [&](int x, double v[NF3]) {
        const PlaneRef p(src + (long)x * plane6, pbytes);
#pragma unroll
        for (int f = 0; f < NF3; ++f) v[f] = p.ld(vc, (unsigned)f * fstride);
    };
    auto load_act = [&](int x) { return PlaneRef(active + (long)x * plane, (unsigned)plane).ldb(cell); };

    double R0[NF3], R1[NF3], R2[NF3];
    typename M::PA P0, P1, P2;
    unsigned char act_c = 1;

    // kept[]: OWN_FROM_LDS only -- the stored U, V, W, Dye of the row, fetched by row() for the inactive lanes
    auto emit = [&](int x, const typename M::PA& C, const typename M::SX& XP, const typename M::SX& XM,
                    const double kept[4], const double pp[NF3]) {
        typename M::SY YP, YM;
        typename M::SZ ZP, ZM;
        M::get_sy(lsy + lN, LS, YP);
        M::get_sy(lsy + lS, LS, YM);
        M::get_sz(lsz + lT, LS, ZP);
        M::get_sz(lsz + lB, LS, ZM);
        double o[NE3];
        if constexpr (M::OWN_FROM_LDS) {
            // the cell's own y / z products are still in LDS, and row() has parked its left-hand sides and
            // coefficients there: read them back instead of carrying them in registers for a row
            typename M::PA Cl = C;
            M::get_sy(lsy + t, LS, Cl.sy);
            M::get_sz(lsz + t, LS, Cl.sz);
            const double* lp = lown + (size_t)(x & 1) * (2 * NE3 * LS) + t;
#pragma unroll
            for (int e = 0; e < NE3; ++e) {
                Cl.lhs[e] = lp[e * LS];
                Cl.aU[e] = lp[(NE3 + e) * LS];
            }
            M::phaseB(K, Cl, XP, XM, YP, YM, ZP, ZM, o);
        } else {
            M::phaseB(K, C, XP, XM, YP, YM, ZP, ZM, o);
        }
        const lane_mask am = mask_of(act_c != 0);
        const PlaneRef d(dst + (long)x * plane6, pbytes);
        d.st(vst, (unsigned)F3_RHO * fstride, o[E3_RHO]);
        d.st(vst, (unsigned)F3_T * fstride, o[E3_T]);
        auto own = [&](int i) { if constexpr (M::OWN_FROM_LDS) return kept[i]; else return M::own(C, i); };
        const double nU = select(am, o[E3_U], own(0)), nV = select(am, o[E3_V], own(1)),
                     nW = select(am, o[E3_W], own(2)), nD = select(am, o[E3_DYE], own(3));
        d.st(vst, (unsigned)F3_U * fstride, nU);
        d.st(vst, (unsigned)F3_V * fstride, nV);
        d.st(vst, (unsigned)F3_W * fstride, nW);
        d.st(vst, (unsigned)F3_DYE * fstride, nD);
        if (FILTER) {
            // Density and Temperature in every cell; U, V, W, Dye only where the cell is active
            const PlaneRef pv(fprev + (long)x * plane6, pbytes);
            const unsigned vm = act_c != 0 ? vst : 0xFFFFFFF0u;
            pv.st(vst, (unsigned)F3_RHO * fstride, pp[F3_RHO] + (0.5 * (o[E3_RHO] - pp[F3_RHO])));
            pv.st(vst, (unsigned)F3_T * fstride, pp[F3_T] + (0.5 * (o[E3_T] - pp[F3_T])));
            pv.st(vm, (unsigned)F3_U * fstride, pp[F3_U] + (0.5 * (nU - pp[F3_U])));
            pv.st(vm, (unsigned)F3_V * fstride, pp[F3_V] + (0.5 * (nV - pp[F3_V])));
            pv.st(vm, (unsigned)F3_W * fstride, pp[F3_W] + (0.5 * (nW - pp[F3_W])));
            pv.st(vm, (unsigned)F3_DYE * fstride, pp[F3_DYE] + (0.5 * (nD - pp[F3_DYE])));
        }
    };
    // stored values of row x in slot t-2, for the fused filter (issued early, consumed in emit)
    auto load_pp = [&](int x, double pp[NF3]) {
        if (FILTER) {
            const PlaneRef q(fpp + (long)x * plane6, pbytes);
#pragma unroll
            for (int f = 0; f < NF3; ++f) pp[f] = q.ld(vc, (unsigned)f * fstride);
        }
    };
    // stored U, V, W, Dye of row x for the lanes whose cell is inactive there (the others fetch nothing)
    auto load_kept = [&](int x, double kept[4]) {
        if constexpr (M::OWN_FROM_LDS) {
            const PlaneRef p(src + (long)x * plane6, pbytes);
            const unsigned v = act_c != 0 ? 0xFFFFFFF0u : vc;
            kept[0] = p.ld(v, (unsigned)F3_U * fstride);
            kept[1] = p.ld(v, (unsigned)F3_V * fstride);
            kept[2] = p.ld(v, (unsigned)F3_W * fstride);
            kept[3] = p.ld(v, (unsigned)F3_DYE * fstride);
        }
    };
    // Row j: Rw, Rc, Re hold rows j-1, j, j+1 (clamped); lraw holds row j of every thread, lsy / lsz
    // the products of row j-1.
    auto row = [&](int j, double* Rw, const double* Rc, const double* Re, typename M::PA& An,
                   const typename M::PA& Ac, const typename M::PA& Aw, bool first) {
        double h[3][NF3];
#pragma unroll
        for (int f = 0; f < NF3; ++f) h[0][f] = half_slope(Re[f], Rc[f], Rw[f]);
        load_row(clampx(j + 2), Rw);        // Rw is dead: fetch row j+2 into it
        const unsigned char act_n = load_act(j);
        double kept[4], pp[NF3];
        if (!first && j - 1 >= xs) { load_kept(j - 1, kept); load_pp(j - 1, pp); }
#pragma unroll
        for (int f = 0; f < NF3; ++f) {
            h[1][f] = half_slope(lraw[f * LS + lN], Rc[f], lraw[f * LS + lS]);
            h[2][f] = half_slope(lraw[f * LS + lT], Rc[f], lraw[f * LS + lB]);
        }
        M::phaseA(K, Rc, h, An);
        if constexpr (M::OWN_FROM_LDS) {
            // this thread's own slot, two rows deep: no other thread reads it, so no barrier is involved
            double* lp = lown + (size_t)(j & 1) * (2 * NE3 * LS) + t;
#pragma unroll
            for (int e = 0; e < NE3; ++e) {
                lp[e * LS] = An.lhs[e];
                lp[(NE3 + e) * LS] = An.aU[e];
            }
        }
        const int x = j - 1;
        // block-uniform branch.  (Finishing every row unconditionally, as sweep2d_march2 does with the filter, costs this
        // kernel 12 % without and 20 % with the filter: it has no registers left for the different schedule.)
        if (!first && x >= xs) emit(x, Ac, An.sx, Aw.sx, kept, pp);
        __syncthreads();
#pragma unroll
        for (int f = 0; f < NF3; ++f) lraw[f * LS + t] = Re[f];
        M::put_sy(lsy + t, LS, An.sy);
        M::put_sz(lsz + t, LS, An.sz);
        __syncthreads();
        act_c = act_n;
    };

    load_row(clampx(j0 - 1), R0);
    load_row(j0, R1);
    load_row(clampx(j0 + 1), R2);
#pragma unroll
    for (int f = 0; f < NF3; ++f) lraw[f * LS + t] = R1[f];
    __syncthreads();

    row(j0, R0, R1, R2, P0, P0, P0, true);
    P2.sx = P0.sx;      // stands in for the row before the first: the clamp-to-self rule
    int j = j0 + 1, last = 0;
    for (;;) {
        if (j > j1) { last = 0; break; }
        row(j, R1, R2, R0, P1, P0, P2, false); ++j;
        if (j > j1) { last = 1; break; }
        row(j, R2, R0, R1, P2, P1, P0, false); ++j;
        if (j > j1) { last = 2; break; }
        row(j, R0, R1, R2, P0, P2, P1, false); ++j;
    }
    // last owned row of the domain: its clamped East neighbour is the cell itself
    if (j1 < xe) {
        double kept[4], pp[NF3];
        load_kept(xe - 1, kept);
        load_pp(xe - 1, pp);
        if (last == 0) emit(xe - 1, P0, P0.sx, P2.sx, kept, pp);
        else if (last == 1) emit(xe - 1, P1, P1.sx, P0.sx, kept, pp);
        else emit(xe - 1, P2, P2.sx, P1.sx, kept, pp);
    }
}

// Streaming form of the tiled 3-D sweep (re-associated arithmetic only): phaseA3s' messages instead of
// shared products, the row-to-row state of sweep2d_stream (pending q / coef / Density, West message,
// ring of RD stored rows) and its unconditional row finishing.
// Tile: TYW x TZW own (y, z) columns, lanes along z; threads 0 .. TYW*TZW-1 hold them in order, so with
// TZW = 16 every 16 lanes store one whole 128-byte line; the threads after them hold the halo ring
// (row above, row below, column after, column before, four corners -- the corners only feed the
// slopes of halo cells), the rest idle.  East/West neighbours stay in registers; the four in-plane
// neighbours' stored values and messages go through LDS (dynamic, single-buffered, two barriers per
// row): 6 stored values + 8 y-message + 9 z-message doubles per thread.
// FILTER: the sweep that ends a step also applies the time filter to the values it has just computed.
template <int TYW, int TZW, int NT, int MINW, bool FILTER, int RD = 3>
__global__ void __launch_bounds__(NT, MINW)
sweep3d_stream(FastConsts K, const double* __restrict__ src, double* __restrict__ dst,
               const uint8_t* __restrict__ active, int x0, int nx, int ny, int nz, int xlo, int xhi, int ch,
               int swizzle, const double* __restrict__ fpp, double* __restrict__ fprev) {
    constexpr int NW = TYW * TZW;           // own columns
    static_assert(NW + 2 * TYW + 2 * TZW + 4 <= NT, "block too small for the tile and its halo ring");
    extern __shared__ double smem[];
    // Row stride of the LDS arrays: one slot more than the block, so that two rows of one array are never a
    // multiple of 512 bytes apart and the compiler cannot pair their reads into ds_read2st64_b64, which moves
    // half as many bytes per LDS cycle as two ds_read_b64 do (-3 % on the launch).
    constexpr int LS = NT + 1;
    double* const lraw = smem;                  // [NF3][LS]
    double* const lmy = smem + NF3 * LS;        // [8][LS]: ms[0..4], mn[0..2]
    double* const lmz = lmy + 8 * LS;           // [9][LS]: mb[0..4], mt[0..3]
    const int t = threadIdx.x;
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned wg = swizzle ? xcd_contiguous_id(lin, nwg) : lin;
    const int bz = (int)(wg % gridDim.x), by = (int)((wg / gridDim.x) % gridDim.y), bxc = (int)(wg / (gridDim.x * gridDim.y));
    const int ytile = by * TYW, ztile = bz * TZW;
    // thread -> (row, column) relative to the tile
    int ry, rz;
    if (t < NW) { ry = t / TZW; rz = t % TZW; }
    else if (t < NW + TZW) { ry = TYW; rz = t - NW; }
    else if (t < NW + 2 * TZW) { ry = -1; rz = t - NW - TZW; }
    else if (t < NW + 2 * TZW + TYW) { ry = t - NW - 2 * TZW; rz = TZW; }
    else if (t < NW + 2 * TZW + 2 * TYW) { ry = t - NW - 2 * TZW - TYW; rz = -1; }
    else { const int c = (t - NW - 2 * TZW - 2 * TYW) & 3; ry = (c & 2) ? -1 : TYW; rz = (c & 1) ? -1 : TZW; }
    const int y = ytile + ry, z = ztile + rz;
    const int cy = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y), cz = z < 0 ? 0 : (z > nz - 1 ? nz - 1 : z);
    auto lane_of = [=](int yy, int zz) {        // thread that holds column (yy, zz); outside the tile and its ring: this thread
        const int a = yy - ytile, b = zz - ztile;
        const bool ain = a >= 0 && a < TYW, bin = b >= 0 && b < TZW;
        if (ain && bin) return a * TZW + b;
        if (bin && a == TYW) return NW + b;
        if (bin && a == -1) return NW + TZW + b;
        if (ain && b == TZW) return NW + 2 * TZW + a;
        if (ain && b == -1) return NW + 2 * TZW + TYW + a;
        if ((a == TYW || a == -1) && (b == TZW || b == -1))
            return NW + 2 * TZW + 2 * TYW + (a == -1 ? 2 : 0) + (b == -1 ? 1 : 0);
        return t;
    };
    // LDS slot of a thread: own columns are rotated by their row number inside their row's TZW slots.
    // Without that, the halo threads beside the tile read one tile column -- a stride of TZW doubles,
    // two banks for the whole column -- and serialise; rotated, the column lands in distinct banks while
    // a row still occupies one aligned run of TZW slots (any permutation of it is conflict-free).
    auto slot_of = [=](int lane) { return lane < NW ? (lane / TZW) * TZW + ((lane % TZW + lane / TZW) % TZW) : lane; };
    const int lN = slot_of(lane_of(cy + 1 > ny - 1 ? ny - 1 : cy + 1, cz)), lS = slot_of(lane_of(cy - 1 < 0 ? 0 : cy - 1, cz));
    const int lT = slot_of(lane_of(cy, cz + 1 > nz - 1 ? nz - 1 : cz + 1)), lB = slot_of(lane_of(cy, cz - 1 < 0 ? 0 : cz - 1));
    const int tw = slot_of(t);      // where this thread publishes
    const bool writer = t < NW && y < ny && z < nz;
    const int xs = x0 + bxc * ch;
    const int xe = xs + ch < nx ? xs + ch : nx;
    const int j0 = xs - 1 > xlo ? xs - 1 : xlo;
    const int j1 = xe;      // at the end of the domain row xe is the clamped last row (see sweep2d_stream)
    const long plane = (long)ny * nz, plane6 = (long)NF3 * plane;
    const unsigned pbytes = (unsigned)(plane6 * 8), fstride = (unsigned)(plane * 8);
    const unsigned cell = (unsigned)cy * (unsigned)nz + (unsigned)cz;
    const unsigned vc = t < NW + 2 * TZW + 2 * TYW + 4 ? cell * 8u : 0xFFFFFFF0u;   // idle threads load nothing
    const unsigned vst = writer ? cell * 8u : 0xFFFFFFF0u;
    auto clampx = [=](int x) { return x < xlo ? xlo : (x > xhi ? xhi : x); };
    auto load_row = [&](int x, double v[NF3]) {
        const PlaneRef p(src + (long)x * plane6, pbytes);
#pragma unroll
        for (int f = 0; f < NF3; ++f) v[f] = p.ld(vc, (unsigned)f * fstride);
    };
    auto load_act = [&](int x) { return PlaneRef(active + (long)x * plane, (unsigned)plane).ldb(cell); };

    double R[RD][NF3];
    double pq[NS3] = {0, 0, 0, 0, 0}, pcoef[NS3] = {0, 0, 0, 0, 0}, prho = 0.0;
    double mwp[4] = {0, 0, 0, 0}, mwpD = 0.0;
    unsigned char act_j;

    // finish row x: pending values + East message me[] + the in-plane neighbours' messages in LDS; stores
    // dropped for a row that belongs to the previous chunk
    auto finish = [&](int x, const double me[NS3], const double pp[NF3]) {
        const unsigned vs = x >= xs ? vst : 0xFFFFFFF0u;
        const double nU = stream_finish3(pq[S3_U], pcoef[S3_U], me[S3_U], lmy[0 * LS + lN], lmy[5 * LS + lS],
                                         lmz[0 * LS + lT], lmz[5 * LS + lB]);
        const double nV = stream_finish3(pq[S3_V], pcoef[S3_V], me[S3_V], lmy[1 * LS + lN], lmy[6 * LS + lS],
                                         lmz[1 * LS + lT], lmz[6 * LS + lB]);
        const double nW = stream_finish3(pq[S3_W], pcoef[S3_W], me[S3_W], lmy[2 * LS + lN], lmy[7 * LS + lS],
                                         lmz[2 * LS + lT], lmz[7 * LS + lB]);
        const double nT = stream_finish3(pq[S3_T], pcoef[S3_T], me[S3_T], lmy[3 * LS + lN], lmy[3 * LS + lS],
                                         lmz[3 * LS + lT], lmz[8 * LS + lB]);
        const double nD = stream_finish3(pq[S3_DYE], pcoef[S3_DYE], me[S3_DYE], lmy[4 * LS + lN], lmy[4 * LS + lS],
                                         lmz[4 * LS + lT], lmz[4 * LS + lB]);
        const PlaneRef d(dst + (long)(x < xlo ? xlo : x) * plane6, pbytes);
        d.st_nt(vs, (unsigned)F3_RHO * fstride, prho);
        d.st_nt(vs, (unsigned)F3_T * fstride, nT);
        d.st_nt(vs, (unsigned)F3_U * fstride, nU);
        d.st_nt(vs, (unsigned)F3_V * fstride, nV);
        d.st_nt(vs, (unsigned)F3_W * fstride, nW);
        d.st_nt(vs, (unsigned)F3_DYE * fstride, nD);
        if (FILTER) {
            // Density and Temperature in every cell; U, V, W, Dye only where the cell is active
            // (coef of U is zero exactly in the cells that keep their values)
            const PlaneRef pv(fprev + (long)(x < xlo ? xlo : x) * plane6, pbytes);
            const unsigned vm = pcoef[S3_U] != 0.0 ? vs : 0xFFFFFFF0u;
            pv.st_nt(vs, (unsigned)F3_RHO * fstride, pp[F3_RHO] + (0.5 * (prho - pp[F3_RHO])));
            pv.st_nt(vs, (unsigned)F3_T * fstride, pp[F3_T] + (0.5 * (nT - pp[F3_T])));
            pv.st_nt(vm, (unsigned)F3_U * fstride, pp[F3_U] + (0.5 * (nU - pp[F3_U])));
            pv.st_nt(vm, (unsigned)F3_V * fstride, pp[F3_V] + (0.5 * (nV - pp[F3_V])));
            pv.st_nt(vm, (unsigned)F3_W * fstride, pp[F3_W] + (0.5 * (nW - pp[F3_W])));
            pv.st_nt(vm, (unsigned)F3_DYE * fstride, pp[F3_DYE] + (0.5 * (nD - pp[F3_DYE])));
        }
    };

    auto row = [&](int j, double* Rw, const double* Rc, const double* Re, bool first) {
        double hx[NF3];
#pragma unroll
        for (int f = 0; f < NF3; ++f) hx[f] = half_slope(Re[f], Rc[f], Rw[f]);
        load_row(clampx(j + RD - 1), Rw);
        const unsigned char act_n = load_act(clampx(j + 1));
        double pp[NF3];
        if (FILTER) {   // stored values of row j-1 in slot t-2, issued early, consumed when the row is finished
            const int xr = j - 1 < xs ? xs : j - 1;
            const PlaneRef q(fpp + (long)xr * plane6, pbytes);
#pragma unroll
            for (int f = 0; f < NF3; ++f) pp[f] = q.ld(vc, (unsigned)f * fstride);
        }
        StreamA3 A;
        // the y and z slopes are formed when their direction's turn comes (the neighbours' values are read
        // from LDS then), which keeps the registers a direction needs from piling up in front of the first one
        phaseA3s_by_direction(K, Rc, [&](int d, double hd[NF3]) {
#pragma unroll
            for (int f = 0; f < NF3; ++f)
                hd[f] = d == 0 ? hx[f]
                      : d == 1 ? half_slope(lraw[f * LS + lN], Rc[f], lraw[f * LS + lS])
                               : half_slope(lraw[f * LS + lT], Rc[f], lraw[f * LS + lB]);
        }, mask_of(act_j != 0), A);
        if (first) {    // clamp-to-self on the West side (kept a real branch by the empty asm)
            asm volatile("");
            mwp[0] = A.mw[0]; mwp[1] = A.mw[1]; mwp[2] = A.mw[2]; mwp[3] = A.mw[3]; mwpD = A.me[S3_DYE];
        }
        stream_west3(A.q, A.coef, mwp, mwpD);
        finish(j - 1, A.me, pp);     // stores dropped for the row that belongs to the previous chunk
        __syncthreads();        // every read of row j's stored values and row j-1's messages is done
#pragma unroll
        for (int f = 0; f < NF3; ++f) lraw[f * LS + tw] = Re[f];
#pragma unroll
        for (int e = 0; e < NS3; ++e) { lmy[e * LS + tw] = A.ms[e]; lmz[e * LS + tw] = A.mb[e]; }
#pragma unroll
        for (int e = 0; e < 3; ++e) lmy[(5 + e) * LS + tw] = A.mn[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) lmz[(5 + e) * LS + tw] = A.mt[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NS3; ++e) { pq[e] = A.q[e]; pcoef[e] = A.coef[e]; }
        prho = A.rho;
        mwp[0] = A.mw[0]; mwp[1] = A.mw[1]; mwp[2] = A.mw[2]; mwp[3] = A.mw[3]; mwpD = A.me[S3_DYE];
        act_j = act_n;
    };

    // West / East messages of a row whose own values this block does not finish (the first and the last row
    // of a chunk): phase A with zero x half-slopes, of which only the two messages are used -- they depend on
    // the y and z faces and the centre products alone, and the compiler drops the rest.  LIGHT: see sweep2d_stream.
    constexpr bool LIGHT = RD == 3;
    auto x_messages = [&](const double* Rc, double me[NS3], double mw[4]) {
        double h[3][NF3];
#pragma unroll
        for (int f = 0; f < NF3; ++f) {
            h[0][f] = 0.0;
            h[1][f] = half_slope(lraw[f * LS + lN], Rc[f], lraw[f * LS + lS]);
            h[2][f] = half_slope(lraw[f * LS + lT], Rc[f], lraw[f * LS + lB]);
        }
        StreamA3 A;
        phaseA3s(K, Rc, h, mask_of(true), A);
#pragma unroll
        for (int e = 0; e < NS3; ++e) me[e] = A.me[e];
        mw[0] = A.mw[0]; mw[1] = A.mw[1]; mw[2] = A.mw[2]; mw[3] = A.mw[3];
    };

    // the message tiles are read (and the result dropped) in the first iteration: define them
    for (int e = 0; e < 8; ++e) lmy[e * LS + tw] = 0.0;
    for (int e = 0; e < 9; ++e) lmz[e * LS + tw] = 0.0;
    int j, jfirst;
    if (LIGHT && j0 < xs) {
        // row xs-1 belongs to the previous chunk and only sends its message to row xs
        load_row(j0, R[0]);
        load_row(xs, R[1]);
        load_row(clampx(xs + 1), R[2]);
        act_j = load_act(xs);
#pragma unroll
        for (int f = 0; f < NF3; ++f) lraw[f * LS + tw] = R[0][f];
        __syncthreads();
        double me0[NS3];
        x_messages(R[0], me0, mwp);
        mwpD = me0[S3_DYE];
        __syncthreads();            // every read of row xs-1's stored values is done
#pragma unroll
        for (int f = 0; f < NF3; ++f) lraw[f * LS + tw] = R[1][f];
        __syncthreads();
        j = xs;
        jfirst = j0 - 1;
    } else {
#pragma unroll
        for (int u = 0; u < RD; ++u) load_row(clampx(j0 - 1 + u), R[u]);
        act_j = load_act(j0);
#pragma unroll
        for (int f = 0; f < NF3; ++f) lraw[f * LS + tw] = R[1][f];
        __syncthreads();
        j = j0;
        jfirst = j0;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): see sweep2d_stream
    const int jl = LIGHT ? j1 - 1 : j1;     // row j1 = xe only sends its message to row xe-1
    int rot = 0;
    while (j <= jl) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            row(j, R[u % RD], R[(u + 1) % RD], R[(u + 2) % RD], j == jfirst);
            ++j;
            rot = (u + 1) % RD;
            if (j > jl) break;
        }
    }
    if (LIGHT) {
        double me1[NS3], mwu[4], pp[NF3];
        if (FILTER) {
            const PlaneRef q(fpp + (long)(xe - 1) * plane6, pbytes);
#pragma unroll
            for (int f = 0; f < NF3; ++f) pp[f] = q.ld(vc, (unsigned)f * fstride);
        }
        if (rot == 0) x_messages(R[1 % RD], me1, mwu);
        else if (rot == 1) x_messages(R[2 % RD], me1, mwu);
        else x_messages(R[3 % RD], me1, mwu);
        finish(xe - 1, me1, pp);
    }
}

#ifndef CFD_KERNEL_TEMPLATES_ONLY
// correctPrevTime: prev <- pp + 0.5*(c - pp); Density and Temperature in every cell, the
// other fields only where the cell is active.  Grid: (ceil(plane/256), nf, nx).
__global__ void __launch_bounds__(256)
time_filter_kernel(const double* __restrict__ cur, const double* __restrict__ pp, double* __restrict__ prev,
                   const uint8_t* __restrict__ active, long plane, int nf, unsigned always_mask) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const int f = blockIdx.y;
    const long x = blockIdx.z;
    const long o = (x * nf + f) * plane + i;
    const bool always = (always_mask >> f) & 1u;
    if (always || active[x * plane + i]) {
        const double c = cur[o], p = pp[o];
        prev[o] = p + (0.5 * (c - p));
    }
}

// Output property of the cells of one z-level of a time slot: `prop_adj Nondirectional prp c Prev`
// as adjGraphToGrid evaluates it (src/Main.hs:339-346, src/SolutionDomain.hs:540-575).  prop is
// the reference's Property enum (CFD_PROPERTY_*).  Stored properties are copied; Speed, Pressure
// and Mew are cell-centre formulas; Vorticity differentiates at the East face: along x the centre
// difference to the East neighbour, along y / z the difference of the East-face (limited) values
// of the two neighbours in that direction (approximateDerivative_adj, :506-538).  The reference's
// operation order throughout.  One thread per cell of the slice; out[x*ny + y].
struct SliceFields {
    const double* base;     // plane x = 0, field 0 of the slot
    long plane;             // ny*nz
    int nf, nz;
    int iU, iV, iW, iRho, iT, iDye;     // field index inside a plane; iW < 0: W is identically +0
    int xlo, xhi;           // lowest / highest readable x-plane (ghost planes included)
    __device__ double at(int f, int x, int y, int z) const {
        return base[((long)x * nf + f) * plane + (long)y * nz + z];
    }
    // value at the East face of cell (x, y, z): vc + 0.5 * superbee(up - vc, vc - lo)
    __device__ double east(int f, int x, int y, int z) const {
        const double vc = at(f, x, y, z);
        const double up = at(f, x + 1 > xhi ? xhi : x + 1, y, z), lo = at(f, x - 1 < xlo ? xlo : x - 1, y, z);
        return vc + half_slope(up, vc, lo);
    }
};

__global__ void __launch_bounds__(256)
property_slice_kernel(SweepConsts K, SliceFields F, int nx, int ny, int zl, int prop, double* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)nx * ny) return;
    const int x = (int)(i / ny), y = (int)(i % ny), z = zl;
    const int nz = F.nz;
    double v;
    switch (prop) {
    case CFD_PROPERTY_SPEED: {
        const double u = F.at(F.iU, x, y, z), w = F.at(F.iV, x, y, z), q = F.iW >= 0 ? F.at(F.iW, x, y, z) : 0.0;
        v = sqrt(((0.0 + u * u) + w * w) + q * q);
        break;
    }
    case CFD_PROPERTY_VORTICITY: {
        const int xe = x + 1 > F.xhi ? F.xhi : x + 1;
        const int yn = y + 1 < ny ? y + 1 : y, ys = y > 0 ? y - 1 : y;
        const int zt = z + 1 < nz ? z + 1 : z, zb = z > 0 ? z - 1 : z;
        // components for the pairs (U,V), (U,W), (V,W)
        const double c1 = ((F.at(F.iV, xe, y, z) - F.at(F.iV, x, y, z)) * 1.0) / K.hx -
                          ((F.east(F.iU, x, yn, z) - F.east(F.iU, x, ys, z)) * 1.0) / K.Hy;
        double c2, c3;
        if (F.iW >= 0) {
            c2 = ((F.at(F.iW, xe, y, z) - F.at(F.iW, x, y, z)) * 1.0) / K.hx -
                 ((F.east(F.iU, x, y, zt) - F.east(F.iU, x, y, zb)) * 1.0) / K.Hz;
            c3 = ((F.east(F.iW, x, yn, z) - F.east(F.iW, x, ys, z)) * 1.0) / K.Hy -
                 ((F.east(F.iV, x, y, zt) - F.east(F.iV, x, y, zb)) * 1.0) / K.Hz;
        } else {
            // one z-level and W == +0: every difference above is of identical values, 0/h - 0/h = 0
            c2 = 0.0;
            c3 = 0.0;
        }
        v = sqrt(((0.0 + c1 * c1) + c2 * c2) + c3 * c3);
        break;
    }
    case CFD_PROPERTY_PRESSURE: v = (K.R * F.at(F.iT, x, y, z)) * F.at(F.iRho, x, y, z); break;
    case CFD_PROPERTY_U: v = F.at(F.iU, x, y, z); break;
    case CFD_PROPERTY_V: v = F.at(F.iV, x, y, z); break;
    case CFD_PROPERTY_W: v = F.iW >= 0 ? F.at(F.iW, x, y, z) : 0.0; break;
    case CFD_PROPERTY_DENSITY: v = F.at(F.iRho, x, y, z); break;
    case CFD_PROPERTY_TEMPERATURE: v = F.at(F.iT, x, y, z); break;
    case CFD_PROPERTY_MEW: v = mu_of(K, F.at(F.iT, x, y, z)); break;
    default: v = F.at(F.iDye, x, y, z); break;
    }
    out[i] = v;
}

// Sum of one field over the owned planes.  Stage 1: fixed grid, grid-stride, wavefront shuffle,
// then LDS across the block's waves; one partial per block.  Stage 2: one block folds the
// partials.  Fixed launch geometry makes the result reproducible from run to run.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ void __launch_bounds__(256)
sum_field_stage1(const double* __restrict__ base, long plane, int nf, int f, int nx, double* __restrict__ partial) {
    __shared__ double wsum[4];
    const long total = (long)nx * plane;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long x = i / plane, r = i - x * plane;
        acc += base[(x * nf + f) * plane + r];
    }
    acc = wave_sum(acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) wsum[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

__global__ void __launch_bounds__(256)
sum_field_stage2(const double* __restrict__ partial, int n, double* __restrict__ out) {
    __shared__ double wsum[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[i];
    acc = wave_sum(acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) wsum[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

// Partial sums of cfd_total_density_noalloc: one array per GPU, part of the loaded code object (nothing is
// allocated for it at run time; each GPU the library runs on holds its copy until the process ends).  In host
// code the name `scratch` is only a handle for hipGetSymbolAddress: the address the kernels get is the
// per-device pointer kept in g_scratch.
static constexpr int SCRATCH_PARTIALS = 1024;
static __device__ __align__(256) double scratch[SCRATCH_PARTIALS];
static constexpr int MAX_DEVICES = 64;
static double* g_scratch[MAX_DEVICES] = {};      // device ordinal -> address of that GPU's copy of `scratch`

// Looks up this GPU's copy of `scratch` (once per device; the device must be current).
static int scratch_setup(int device) {
    if (device < 0 || device >= MAX_DEVICES) return fail(CFD_E_ARG, "device ordinal beyond the library's table of 64");
    if (g_scratch[device]) return CFD_OK;
    double* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    g_scratch[device] = ptr;
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// fpp / fprev non-null: also apply the time filter to the freshly computed values (kernels that
// support it; *did_fuse tells the caller whether it happened).
template <class M>
static size_t march3_lds_bytes(int nt) {
    return (size_t)(NF3 + M::NSY + M::NSZ + (M::OWN_FROM_LDS ? 4 * NE3 : 0)) * (nt + 1) * sizeof(double);
}

// x0, x1: the range of owned x-planes to compute (x1 < 0: all of them).
// x-planes per block for the default 2-D streaming launch (256 threads, 240 columns) over nplanes planes.
// Three of its blocks fit a compute unit and every block of a launch takes the same time, so a launch runs in
// ceil(blocks / (3 * CUs)) rounds of chunk full rows plus two message-only rows (about 0.3 of a full row each);
// measured launch times follow that product.  Chunks between 24 and 40 planes: longer ones finish in
// fewer rounds but leave the last round's tail unbalanced.
static int stream_chunk(const cfd_handle* h, int nplanes) {
    const long slots = 3L * h->cus;
    int best = 32;
    long best_cost = -1;
    for (int ch = 24; ch <= 40; ++ch) {
        const int nchunk = (nplanes + ch - 1) / ch;
        const TilePlan tp = tile_plan(h->p.ny, 240, 256, 2, nchunk);
        const long blocks = (long)tp.nfull * nchunk + tp.npacked;
        const long cost = ((blocks + slots - 1) / slots) * (10 * ch + 6);
        if (best_cost < 0 || cost < best_cost) { best = ch; best_cost = cost; }
    }
    return best;
}

// x-planes per block for the strict tiled kernels (one wave per SIMD: `slots` blocks run at once, every block of a
// launch takes the same time per row and evaluates two rows more than it stores).  A launch runs in
// ceil(blocks / slots) rounds of (chunk + 2) rows plus a prologue of about one row; the chunk that minimises that
// product fills whole rounds (4096 x 4096: 134 planes = 31 chunks x 33 tiles = 1023 blocks on 512 slots, two full
// rounds, against nine rounds of 32 planes with the ninth a quarter full: -5 % plain, -8 % with the filter).
static int march_chunk(int nplanes, long tiles, long slots) {
    int best = 32;
    long best_cost = -1;
    for (int ch = 8; ch <= 256; ++ch) {
        const long nchunk = (nplanes + ch - 1) / ch;
        const long cost = ((tiles * nchunk + slots - 1) / slots) * (ch + 3);
        if (best_cost < 0 || cost < best_cost) { best = ch; best_cost = cost; }
    }
    return best;
}

// x-planes per block of the three-sweeps kernel.  One block (nts == 256) or two (128) run on a compute unit at a time,
// every block of a launch takes the same time per iteration, and a block of `ch` planes runs ch + 10 iterations
// (the three stages follow one another three rows apart): the launch takes ceil(blocks / slots) rounds of that.
// columns a tile's width is a multiple of
static int trio_gran() {
    static const int g = [] { const char* e = getenv("CFD_TRIO_GRAN"); const int v = e ? atoi(e) : 8; return v >= 1 && v <= 16 ? v : 8; }();
    return g;
}
static int trio_wmax(int nts) { return trio_gran() == 16 ? nts - 16 : nts - 6; }
static int trio_chunk_for(int nx, int ny, int nts, int cus) {
    const long slots = (nts == 128 ? 2L : 1L) * cus, tiles = trio_tiles(ny, trio_wmax(nts), trio_gran()).ntile;
    int best = nx;
    long best_cost = -1;
    for (int nchunk = 1; nchunk <= nx; ++nchunk) {
        const int ch = (nx + nchunk - 1) / nchunk;
        if (ch < 8 && nchunk > 1) break;
        const long cost = ((tiles * ((nx + ch - 1) / ch) + slots - 1) / slots) * (ch + 10);
        if (best_cost < 0 || cost < best_cost) { best = ch; best_cost = cost; }
    }
    return best;
}

// Number of chunks of the three-sweeps kernel and the planes [xs, xe) of chunk c.
static int trio_nchunk(const cfd_handle* h) {
    const int nx = h->p.nx;
    return nx <= h->trio_first ? 1 : 1 + (nx - h->trio_first + h->trio_chunk - 1) / h->trio_chunk;
}
static void trio_span(const cfd_handle* h, int c, int* xs, int* xe) {
    const int nx = h->p.nx, ch = h->trio_chunk, f = h->trio_first;
    *xs = c == 0 ? 0 : f + (c - 1) * ch;
    *xe = c == 0 ? (f < nx ? f : nx) : (*xs + ch < nx ? *xs + ch : nx);
}

// Chunk lengths of the three-sweeps kernel for this handle.  Undecomposed: uniform (trio_chunk_for).  A slab launches
// the chunks next to a neighbour only once the neighbour's planes have arrived, i.e. later than the others by the
// duration of the exchange, and all blocks of a step should end together: those chunks get `margin` planes fewer
// (env CFD_TRIO_EDGE_MARGIN, default 32 = about 77 us of the kernel's 2.4 us per plane: a three-plane exchange between
// two processes is expected to take 30-60 us, and ten planes too many cost the other chunks half a percent, ten too
// few cost the step 24 us), the others share the rest.
static void trio_plan(cfd_handle* h) {
    const cfd_params& p = h->p;
    const int ch0 = trio_chunk_for(p.nx, p.ny, h->trio_nts, h->cus);
    h->trio_chunk = h->trio_first = ch0;
    const int edges = (p.ghost_lo ? 1 : 0) + (p.ghost_hi ? 1 : 0), nchunk = (p.nx + ch0 - 1) / ch0;
    int margin = getenv("CFD_TRIO_EDGE_MARGIN") ? atoi(getenv("CFD_TRIO_EDGE_MARGIN")) : 32;
    if (margin > ch0 / 4) margin = ch0 / 4;
    if (!edges || nchunk < 3 || margin < 1) return;
    // blocks that need more than one round of the device are not launched in two parts (cfd_internal_one_pass_blocks)
    if ((long)nchunk * trio_tiles(p.ny, trio_wmax(h->trio_nts), trio_gran()).ntile > (long)(h->trio_nts == 128 ? 2 : 1) * h->cus) return;
    // edges * (ch - margin) + (nchunk - edges) * ch >= nx, with the last chunk (what is left) no longer than ch - margin
    // where it is next to a neighbour
    int ch = (p.nx + edges * margin + nchunk - 1) / nchunk;
    const int first = p.ghost_lo ? ch - margin : ch;
    if (p.ghost_hi) {
        // grow ch until the last chunk is short enough
        while (p.nx - first - (nchunk - 2) * ch > ch - margin) ++ch;
    }
    if (first < 3 || p.nx - first - (nchunk - 2) * ch < 1) return;     // keep the uniform plan
    h->trio_chunk = ch;
    h->trio_first = p.ghost_lo ? ch - margin : ch;
}

// Chunk lengths of the kernels that fuse sweeps, for h->cus compute units (the device's, or the handle's share of
// them where several handles run on one device at the same time).
static void plan_chunks(cfd_handle* h) {
    const cfd_params* p = &h->p;
    if (getenv("CFD_PAIR_CHUNK")) {
        h->pair_chunk = atoi(getenv("CFD_PAIR_CHUNK"));
    } else {
        // The two-sweeps kernel computes five extra rows per chunk, so chunks should be long; two of its
        // blocks fit a CU, and the whole grid running as one round of blocks was the fastest shape measured.
        const int cus = h->cus;
        // most chunks whose blocks (tile_plan: the narrow last tile of a row shares blocks) fit one round
        int chunks = 1;
        for (int c = 2 * cus; c > 1; --c) {
            const TilePlan tp = tile_plan(p->ny, 240, 256, 4, c);
            if (tp.nfull * c + tp.npacked <= 2 * cus) { chunks = c; break; }
        }
        h->pair_chunk = (p->nx + chunks - 1) / chunks;
        if (h->pair_chunk < 32) h->pair_chunk = 32;
    }
    if (h->pair_chunk < 1) h->pair_chunk = 1;
    if (getenv("CFD_TRIO_CHUNK")) {
        h->trio_chunk = atoi(getenv("CFD_TRIO_CHUNK"));
        if (h->trio_chunk < 1) h->trio_chunk = 1;
        h->trio_first = h->trio_chunk;
    } else {
        trio_plan(h);
    }
}

// The strict 2-D sweep without the filter, LEAN form, instantiated in cfd_strict2d_lean.hip (compiled with the
// minimum-register scheduling strategy, under which it fits 256 registers: two waves per SIMD).
extern "C" int cfd_internal_launch_strict2d_lean(const SweepConsts* K, const double* src, double* dst,
                                                 const uint8_t* act, int x0, int x1, int ny, int xlo, int xhi, int ch,
                                                 void* stream, void* ev_begin, void* ev_end);

static int launch_sweep(cfd_handle* h, const double* src, double* dst, const double* fpp = nullptr,
                        double* fprev = nullptr, bool* did_fuse = nullptr, int x0 = 0, int x1 = -1) {
    bool fused = false;
    const cfd_params& p = h->p;
    if (x1 < 0) x1 = p.nx;
    if (x1 <= x0) return CFD_OK;        // empty range: nothing launched, *did_fuse untouched
    const int xlo = p.ghost_lo ? -1 : 0;
    const int xhi = p.ghost_hi ? p.nx : p.nx - 1;
    EventPair ev{};
    if (h->timing) {
        if (h->ev_free.empty()) {
            HIP_TRY(hipEventCreate(&ev.a));
            HIP_TRY(hipEventCreate(&ev.b));
        } else {
            ev = h->ev_free.back();
            h->ev_free.pop_back();
        }
    }
    // With timing on, the start/stop events are attached to the kernel dispatch itself, so their
    // difference is the kernel's own begin-to-end time (what a kernel trace reports), free of
    // queueing gaps.
    hipEvent_t ea = h->timing ? ev.a : nullptr, eb = h->timing ? ev.b : nullptr;
    const bool fast = (p.flags & CFD_FLAG_FAST_MATH) != 0;
    const uint8_t* act = h->d_active + GHOST * h->plane;
    if (h->two_d && (p.flags & CFD_FLAG_SIMPLE_KERNEL)) {
        dim3 block(64), grid((p.ny + 63) / 64, x1 - x0);
        if (fast)
            hipExtLaunchKernelGGL(sweep2d_simple<MathFast2>, grid, block, 0, h->stream, ea, eb, 0, h->KF, src, dst,
                                  act, x0, x1, p.ny, xlo, xhi);
        else
            hipExtLaunchKernelGGL(sweep2d_simple<MathStrict2>, grid, block, 0, h->stream, ea, eb, 0, h->K, src, dst,
                                  act, x0, x1, p.ny, xlo, xhi);
    } else if (h->two_d) {
        static const bool strict_lean = !getenv("CFD_STRICT_LEAN") || atoi(getenv("CFD_STRICT_LEAN")) != 0;
        const int ch = !(h->chunk_auto && h->variant < 0) ? h->chunk
                       : fast ? stream_chunk(h, x1 - x0)
                              : march_chunk(x1 - x0, (p.ny + 125) / 126, (fpp || !strict_lean ? 2L : 4L) * h->cus);   // 128 threads, two (LEAN: four) blocks per CU
#define LAUNCH_MARCH2(M, KK, NT, MINW) LAUNCH_MARCH2W(M, KK, NT, MINW, (NT - 2))
#define LAUNCH_MARCH2W(M, KK, NT, MINW, WC)                                                         \
    do {                                                                                           \
        dim3 block(NT), grid((p.ny + WC - 1) / WC, (x1 - x0 + ch - 1) / ch);                       \
        if (fpp)                                                                                   \
            hipExtLaunchKernelGGL((sweep2d_march2<M, NT, MINW, true, WC>), grid, block, 0, h->stream, \
                                  ea, eb, 0, KK, src, dst, act, x0, x1, p.ny, xlo, xhi, ch, fpp,     \
                                  fprev);                                                          \
        else                                                                                       \
            hipExtLaunchKernelGGL((sweep2d_march2<M, NT, MINW, false, WC>), grid, block, 0, h->stream, \
                                  ea, eb, 0, KK, src, dst, act, x0, x1, p.ny, xlo, xhi, ch, fpp,     \
                                  fprev);                                                          \
        fused = fpp != nullptr;                                                                    \
    } while (0)
        static const int swz = getenv("CFD_SWIZZLE") ? atoi(getenv("CFD_SWIZZLE")) : 1;     // XCD-contiguous tile order (tuning knob)
#define LAUNCH_STREAM(NT, MINW, WC) LAUNCH_STREAM_RD(NT, MINW, WC, 3)
#define LAUNCH_STREAM_RD(NT, MINW, WC, RDV)                                                         \
    do {                                                                                           \
        const int nchunk = (x1 - x0 + ch - 1) / ch;                                                \
        const TilePlan tp = tile_plan(p.ny, WC, NT, 2, nchunk);                                    \
        dim3 block(NT), grid(tp.nfull * nchunk + tp.npacked);                                      \
        if (fpp)                                                                                   \
            hipExtLaunchKernelGGL((sweep2d_stream<NT, MINW, true, RDV, WC>), grid, block, 0,       \
                                  h->stream, ea, eb, 0, h->KF, src, dst, act, x0, x1, p.ny, xlo,     \
                                  xhi, ch, fpp, fprev, swz);                                       \
        else                                                                                       \
            hipExtLaunchKernelGGL((sweep2d_stream<NT, MINW, false, RDV, WC>), grid, block, 0,      \
                                  h->stream, ea, eb, 0, h->KF, src, dst, act, x0, x1, p.ny, xlo,     \
                                  xhi, ch, fpp, fprev, swz);                                       \
        fused = fpp != nullptr;                                                                    \
    } while (0)
        // CFD_VARIANT (tuning / cross-checks): streamed messages 20..31, shared products 8..16
        if (fast) {
            switch (h->variant < 0 ? 30 : h->variant) {
                case 20: LAUNCH_STREAM(128, 3, 126); break;
                case 21: LAUNCH_STREAM(128, 2, 126); break;
                case 22: LAUNCH_STREAM(64, 3, 62); break;
                case 23: LAUNCH_STREAM(256, 3, 254); break;
                case 24: LAUNCH_STREAM(256, 3, 240); break;
                case 28: LAUNCH_STREAM_RD(256, 2, 240, 5); break;
                case 31: LAUNCH_STREAM(128, 2, 112); break;
                case 8: LAUNCH_MARCH2(MathFast2, h->KF, 128, 2); break;
                case 12: LAUNCH_MARCH2(MathFast2, h->KF, 256, 2); break;
                case 13: LAUNCH_MARCH2W(MathFast2, h->KF, 256, 2, 240); break;
                default: LAUNCH_STREAM(256, 2, 240); break;     // 30: whole 128-byte lines per wave
            }
        } else {
            switch (h->variant < 0 ? 8 : h->variant) {
                case 9: LAUNCH_MARCH2(MathStrict2, h->K, 128, 2); break;
                case 12: LAUNCH_MARCH2(MathStrict2, h->K, 256, 1); break;
                case 13: LAUNCH_MARCH2W(MathStrict2, h->K, 256, 1, 240); break;
                case 14: LAUNCH_MARCH2W(MathStrict2, h->K, 128, 1, 112); break;
                default:        // 8
                    if (!fpp && strict_lean && h->variant < 0) {
                        HIP_TRY((hipError_t)cfd_internal_launch_strict2d_lean(&h->K, src, dst, act, x0, x1, p.ny, xlo, xhi,
                                                                              ch, (void*)h->stream, (void*)ea, (void*)eb));
                        fused = false;
                    } else {
                        LAUNCH_MARCH2(MathStrict2, h->K, 128, 1);
                    }
                    break;
            }
        }
#undef LAUNCH_MARCH2
#undef LAUNCH_MARCH2W
#undef LAUNCH_STREAM
#undef LAUNCH_STREAM_RD
    } else if ((p.flags & CFD_FLAG_SIMPLE_KERNEL) || !h->march3_ok || (!fast && !h->march3_strict_ok && h->variant < 0)) {
        dim3 block(64), grid((unsigned)((h->plane + 63) / 64), x1 - x0);
        if (fast)
            hipExtLaunchKernelGGL(sweep3d_simple<MathFast3>, grid, block, 0, h->stream, ea, eb, 0, h->KF, src, dst,
                                  act, x0, x1, p.ny, p.nz, xlo, xhi);
        else
            hipExtLaunchKernelGGL(sweep3d_simple<MathStrict3>, grid, block, 0, h->stream, ea, eb, 0, h->K, src, dst,
                                  act, x0, x1, p.ny, p.nz, xlo, xhi);
    } else {
        // strict default: 16 x 16 threads (14 x 14 own columns), one block per CU
        const int ch = (h->chunk_auto && !fast && h->variant < 0)
                           ? march_chunk(x1 - x0, (long)((p.nz + 13) / 14) * ((p.ny + 13) / 14), h->cus)
                           : h->chunk;
#define LAUNCH_MARCH3(M, KK, TY, TZ, MINW)                                                          \
    do {                                                                                           \
        auto kern = sweep3d_march<M, TY, TZ, MINW>;                                                \
        const size_t lds = march3_lds_bytes<M>(TY * TZ);                                           \
        static bool attr_set[64] = {};                                                             \
        if (lds > 64 * 1024 && !attr_set[h->device & 63]) {                                        \
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                \
            attr_set[h->device & 63] = true;                                                       \
        }                                                                                          \
        dim3 block(TY * TZ), grid((p.nz + TZ - 3) / (TZ - 2), (p.ny + TY - 3) / (TY - 2),          \
                                  (x1 - x0 + ch - 1) / ch);                                           \
        hipExtLaunchKernelGGL(kern, grid, block, lds, h->stream, ea, eb, 0, KK, src, dst, act,     \
                              x0, x1, p.ny, p.nz, xlo, xhi, ch, (const double*)nullptr, (double*)nullptr); \
    } while (0)
    // the strict default shape, with the time filter fused in where asked for
#define LAUNCH_MARCH3_STRICT()                                                                      \
    do {                                                                                           \
        auto kern = fpp ? sweep3d_march<MathStrict3, 16, 16, 1, true> : sweep3d_march<MathStrict3, 16, 16, 1, false>; \
        const size_t lds = march3_lds_bytes<MathStrict3>(256);                                     \
        static bool attr_set[2][64] = {};                                                          \
        if (lds > 64 * 1024 && !attr_set[fpp ? 1 : 0][h->device & 63]) {                           \
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                \
            attr_set[fpp ? 1 : 0][h->device & 63] = true;                                          \
        }                                                                                          \
        dim3 block(256), grid((p.nz + 13) / 14, (p.ny + 13) / 14, (x1 - x0 + ch - 1) / ch);        \
        hipExtLaunchKernelGGL(kern, grid, block, lds, h->stream, ea, eb, 0, h->K, src, dst, act,   \
                              x0, x1, p.ny, p.nz, xlo, xhi, ch, fpp, fprev);                        \
        fused = fpp != nullptr;                                                                    \
    } while (0)
#define LAUNCH_STREAM3(TYW, TZW, NT, MINW, RDV)                                                     \
    do {                                                                                           \
        auto kern = fpp ? sweep3d_stream<TYW, TZW, NT, MINW, true, RDV>                            \
                        : sweep3d_stream<TYW, TZW, NT, MINW, false, RDV>;                          \
        const size_t lds = (size_t)(NF3 + 8 + 9) * (NT + 1) * sizeof(double);                      \
        static bool attr_set[2][64] = {};                                                          \
        if (lds > 64 * 1024 && !attr_set[fpp ? 1 : 0][h->device & 63]) {                           \
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds));                                                \
            attr_set[fpp ? 1 : 0][h->device & 63] = true;                                          \
        }                                                                                          \
        dim3 block(NT), grid((p.nz + TZW - 1) / TZW, (p.ny + TYW - 1) / TYW, (x1 - x0 + ch - 1) / ch); \
        hipExtLaunchKernelGGL(kern, grid, block, lds, h->stream, ea, eb, 0, h->KF, src, dst, act,  \
                              x0, x1, p.ny, p.nz, xlo, xhi, ch, swz3, fpp, fprev);                 \
        fused = fpp != nullptr;                                                                    \
    } while (0)
        static const int swz3 = getenv("CFD_SWIZZLE") ? atoi(getenv("CFD_SWIZZLE")) : 1;
        if (fast) {
            switch (h->variant < 0 ? 0 : h->variant) {
                case 11: LAUNCH_STREAM3(12, 16, 256, 1, 3); break;
                case 12: LAUNCH_STREAM3(5, 32, 256, 2, 3); break;
                case 13: LAUNCH_STREAM3(14, 14, 256, 2, 3); break;      // the unaligned 14 x 14 tile
                case 14: LAUNCH_STREAM3(12, 16, 256, 2, 4); break;
                case 15: LAUNCH_STREAM3(4, 16, 128, 2, 3); break;
                case 16: LAUNCH_STREAM3(26, 16, 512, 2, 3); break;
                case 1: LAUNCH_MARCH3(MathFast3, h->KF, 8, 32, 2); break;
                case 2: LAUNCH_MARCH3(MathFast3, h->KF, 16, 32, 1); break;
                case 3: LAUNCH_MARCH3(MathFast3, h->KF, 8, 16, 2); break;
                case 4: LAUNCH_MARCH3(MathFast3, h->KF, 16, 16, 2); break;
                case 5: LAUNCH_MARCH3(MathFast3, h->KF, 16, 16, 1); break;
                default: LAUNCH_STREAM3(12, 16, 256, 2, 3); break;
            }
        } else {
            switch (h->variant < 0 ? 0 : h->variant) {
                case 1: LAUNCH_MARCH3(MathStrict3, h->K, 8, 32, 1); break;
                case 3: LAUNCH_MARCH3(MathStrict3, h->K, 8, 16, 1); break;
                case 9:         // the per-cell kernel
                    hipExtLaunchKernelGGL(sweep3d_simple<MathStrict3>, dim3((unsigned)((h->plane + 63) / 64), x1 - x0),
                                          dim3(64), 0, h->stream, ea, eb, 0, h->K, src, dst, act, x0, x1, p.ny, p.nz,
                                          xlo, xhi);
                    break;
                case 2: LAUNCH_MARCH3(MathStrict3, h->K, 16, 16, 1); break;       // never fused with the filter
                default: LAUNCH_MARCH3_STRICT(); break;
            }
        }
#undef LAUNCH_MARCH3
#undef LAUNCH_MARCH3_STRICT
#undef LAUNCH_STREAM3
    }
    HIP_TRY(hipGetLastError());
    if (h->timing) { ev.kind = fused ? 1 : 0; h->ev_used.push_back(ev); }
    if (did_fuse) *did_fuse = fused;
    return CFD_OK;
}

// Two in-place sweeps (+ filter) in one pass; only where pair_ok(h).
static bool pair_ok(const cfd_handle* h) {
    const cfd_params& p = h->p;
    // beside the default launch shape only; a slab needs the masks of its ghost planes and at least two planes
    // (the planes it sends two-deep to a neighbour must be its own)
    const bool ghosts = p.ghost_lo || p.ghost_hi;
    return h->two_d && (p.flags & CFD_FLAG_FAST_MATH) && !(p.flags & CFD_FLAG_SIMPLE_KERNEL) && h->pair_enabled &&
           h->variant < 0 && (!ghosts || (h->ghost_mask_depth >= 1 && p.nx >= 2));
}

static int launch_pair(cfd_handle* h, const double* src, double* dst, const double* fpp, double* fprev) {
    const cfd_params& p = h->p;
    EventPair ev{};
    if (h->timing) {
        if (h->ev_free.empty()) {
            HIP_TRY(hipEventCreate(&ev.a));
            HIP_TRY(hipEventCreate(&ev.b));
        } else {
            ev = h->ev_free.back();
            h->ev_free.pop_back();
        }
    }
    hipEvent_t ea = h->timing ? ev.a : nullptr, eb = h->timing ? ev.b : nullptr;
    const uint8_t* act = h->d_active + GHOST * h->plane;
    const int ch = h->pair_chunk;
    constexpr int NT = 256, WC = 240;
    const int nchunk = (p.nx + ch - 1) / ch;
    const TilePlan tp = tile_plan(p.ny, WC, NT, 4, nchunk);
    dim3 block(NT), grid(tp.nfull * nchunk + tp.npacked);
    // with a neighbouring slab: stored values reach two ghost planes out, once-swept values one
    const int xlo = p.ghost_lo ? -2 : 0, xhi = p.ghost_hi ? p.nx + 1 : p.nx - 1;
    const int xlo1 = p.ghost_lo ? -1 : 0, xhi1 = p.ghost_hi ? p.nx : p.nx - 1;
    if (fpp)
        hipExtLaunchKernelGGL((sweep2d_pair<NT, 2, WC, true>), grid, block, 0, h->stream, ea, eb, 0, h->KF, src, dst,
                              act, 0, p.nx, p.ny, xlo, xhi, xlo1, xhi1, ch, fpp, fprev);
    else
        hipExtLaunchKernelGGL((sweep2d_pair<NT, 2, WC, false>), grid, block, 0, h->stream, ea, eb, 0, h->KF, src, dst,
                              act, 0, p.nx, p.ny, xlo, xhi, xlo1, xhi1, ch, fpp, fprev);
    HIP_TRY(hipGetLastError());
    if (h->timing) { ev.kind = 2; h->ev_used.push_back(ev); }
    return CFD_OK;
}

// All three sweeps of a step (+ filter) in one pass; only where trio_ok(h).
static bool trio_ok(const cfd_handle* h) {
    const cfd_params& p = h->p;
    // a slab needs the masks of two ghost planes per side and at least three planes (the planes it sends
    // three-deep to a neighbour must be its own)
    const bool ghosts = p.ghost_lo || p.ghost_hi;
    return h->two_d && (p.flags & CFD_FLAG_FAST_MATH) && !(p.flags & CFD_FLAG_SIMPLE_KERNEL) && h->trio_enabled &&
           h->variant < 0 && (!ghosts || (h->ghost_mask_depth >= 2 && p.nx >= 3));
}

#ifndef CFD_KERNEL_TEMPLATES_ONLY
// The XCD the first workgroup of a launch on `stream` runs on (the queue constant o_q of xcd_fill_rank), measured
// once per stream with a one-workgroup kernel that reads the hardware's XCC_ID register; -1 if it cannot be measured.
__global__ void xcc_probe_kernel(int* out) {
    if (threadIdx.x == 0) out[0] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15);      // XCC_ID: register 20, bits 3:0
}
static std::mutex g_xcc_mu;
static std::map<std::pair<int, hipStream_t>, int> g_xcc_known;
// A stream that is destroyed takes its measurement with it (a later stream may get the same handle value and
// another hardware queue).
// For cfd_multi.hip: blocks of the handle's one-pass step and the compute units it plans for.  A step whose blocks do
// not fit the device at once is better not split into an interior and a late launch: the interior launch then holds
// every compute unit for a whole block's time, and neither the exchange's kernel nor the late blocks can start.
extern "C" int cfd_internal_one_pass_blocks(cfd_handle* h, int* blocks, int* cus) {
    if (!h || !blocks || !cus) return fail(CFD_E_ARG, "cfd_internal_one_pass_blocks: null pointer");
    *blocks = trio_nchunk(h) * trio_tiles(h->p.ny, trio_wmax(h->trio_nts), trio_gran()).ntile;
    *cus = (h->trio_nts == 128 ? 2 : 1) * h->cus;
    return CFD_OK;
}
// For the tests: the placement of `units` blocks beside a launch of `busy` blocks (xcd_fill_grid / xcd_fill_rank);
// rank_of[b] for b < the returned grid size (at most `max` entries written).
extern "C" int cfd_internal_xcd_fill(int units, int busy, int cap, int oa, int ob, int* rank_of, int max) {
    const int grid = xcd_fill_grid(units, busy, cap, oa, ob);
    for (int b = 0; b < grid && b < max; ++b) rank_of[b] = xcd_fill_rank(b, busy, cap, oa, ob);
    return grid;
}
extern "C" void cfd_internal_forget_stream(int device, void* stream) {
    std::lock_guard<std::mutex> lock(g_xcc_mu);
    g_xcc_known.erase(std::make_pair(device, (hipStream_t)stream));
}
static int xcc_offset_of(int device, hipStream_t stream) {
    std::mutex& mu = g_xcc_mu;
    std::map<std::pair<int, hipStream_t>, int>& known = g_xcc_known;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(device, stream);
    auto it = known.find(key);
    if (it != known.end()) return it->second;
    int off = -1, *d = nullptr;
    if (hipMalloc((void**)&d, sizeof(int)) == hipSuccess) {
        // three launches must agree: the constant belongs to the queue, not to the launch
        int v[3] = {-1, -2, -3};
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {
            hipLaunchKernelGGL(xcc_probe_kernel, dim3(1), dim3(64), 0, stream, d);
            ok = hipMemcpyAsync(&v[k], d, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                 hipStreamSynchronize(stream) == hipSuccess;
        }
        if (ok && v[0] == v[1] && v[1] == v[2] && v[0] >= 0 && v[0] < 8) off = v[0];
        (void)hipFree(d);
    }
    known[key] = off;
    return off;
}
#endif

// src: the slot the step starts from; dst: the slot that receives the step's result; fpp / fprev (both or neither):
// the filter's older time level and the buffer that receives the filtered values of EVERY owned cell.
// c0, c1: the range of chunks (of h->trio_chunk planes) to launch; c1 < 0: all of them.
// d0, d1: a second range of chunks in the same launch (empty by default).
// busy: blocks of a launch of this step that is running beside this one (0: none): see xcd_fill_rank.
static int launch_trio(cfd_handle* h, const double* src, double* dst, const double* fpp, double* fprev, int c0 = 0,
                       int c1 = -1, int d0 = 0, int d1 = 0, int busy = 0) {
    const cfd_params& p = h->p;
    EventPair ev{};
    if (h->timing) {
        if (h->ev_free.empty()) {
            HIP_TRY(hipEventCreate(&ev.a));
            HIP_TRY(hipEventCreate(&ev.b));
        } else {
            ev = h->ev_free.back();
            h->ev_free.pop_back();
        }
    }
    hipEvent_t ea = h->timing ? ev.a : nullptr, eb = h->timing ? ev.b : nullptr;
    const uint8_t* act = h->d_active + GHOST * h->plane;
    const int ch = h->trio_chunk, nts = h->trio_nts, wmax = trio_wmax(nts), gran = trio_gran();
    const TrioTiles tp = trio_tiles(p.ny, wmax, gran);
    const int nchunk = trio_nchunk(h);
    if (c1 < 0) c1 = nchunk;
    if (c1 < c0) c1 = c0;
    if (d1 < d0) d1 = d0;
    const int nblk = (c1 - c0) + (d1 - d0);
    if (nblk <= 0) return CFD_OK;       // empty ranges: nothing launched
    // beside a running launch: one block per compute unit, eight XCDs of cus / 8 compute units
    const int xcd_cap = h->cus_device / 8;
    int grid = tp.ntile * nblk;
    int xcd_off = 0;
    // (eight XCDs: the whole device, not a partition of it)
    if (busy > 0 && nts == 256 && h->cus == h->cus_device && h->cus_device % 8 == 0 && h->cus_device >= 128 && h->xcd_fill) {
        const int oa = xcc_offset_of(h->device, h->interior_stream), ob = xcc_offset_of(h->device, h->stream);
        const int g = oa < 0 || ob < 0 ? 0 : xcd_fill_grid(grid, busy, xcd_cap, oa, ob);
        if (g > 0) { grid = g; xcd_off = oa | (ob << 4); } else busy = 0;
    } else {
        busy = 0;
    }
    const size_t lds = (size_t)TRIO_LDS_ROWS * nts * sizeof(double);
    // with a neighbouring slab the stored values reach three ghost planes out
    const int lo0 = p.ghost_lo ? -3 : 0, hi0 = p.ghost_hi ? p.nx + 2 : p.nx - 1;
#define LAUNCH_TRIO(NTS, MINW, FILT)                                                               \
    do {                                                                                           \
        auto kern = sweep2d_trio<NTS, MINW, FILT>;                                                 \
        static bool attr_set[64] = {};                                                             \
        if (lds > 64 * 1024 && !attr_set[h->device & 63]) {                                        \
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            attr_set[h->device & 63] = true;                                                       \
        }                                                                                          \
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(3 * NTS), lds, h->stream, ea, eb, 0, h->KF,             \
                              src, dst, act, p.nx, p.ny, lo0, hi0, ch, h->trio_first, c0, c1 - c0, d0, wmax, gran, \
                              busy, xcd_cap, xcd_off, fpp, fprev);                                 \
    } while (0)
    if (nts == 128) { if (fpp) LAUNCH_TRIO(128, 3, true); else LAUNCH_TRIO(128, 3, false); }
    else { if (fpp) LAUNCH_TRIO(256, 3, true); else LAUNCH_TRIO(256, 3, false); }
#undef LAUNCH_TRIO
    HIP_TRY(hipGetLastError());
    if (h->timing) { ev.kind = 3; h->ev_used.push_back(ev); }
    return CFD_OK;
}

// runSingleStep as one launch: the first sweep advances the time slot, the other two stay in it, then the filter
// writes slot t-1.  That slot is the buffer the step started from, which other blocks still read, so the filtered
// values go to the scratch buffer, which then takes that slot's place (the step's source buffer becomes scratch).
// c0, c1: chunk range as in launch_trio; finish: this call completes the step (slot rotation, time counters).
static int step_trio(cfd_handle* h, int c0 = 0, int c1 = -1, bool finish = true, int d0 = 0, int d1 = 0, int busy = 0) {
    const int nc = (h->cmt + 1) % 3;
    const double* src = slot_base(h, h->cmt);
    const bool filter = h->cat + 1 > 1;
    int rc = launch_trio(h, src, slot_base(h, nc), filter ? slot_base(h, (h->cmt + 2) % 3) : nullptr,
                         filter ? slot_base(h, 3) : nullptr, c0, c1, d0, d1, busy);
    if (rc || !finish) return rc;
    if (filter) {
        const int t = h->map[h->cmt];
        h->map[h->cmt] = h->map[3];
        h->map[3] = t;
    }
    h->cmt = nc;
    h->cat += 1;
    return CFD_OK;
}

extern "C" {

const char* cfd_last_error(void) { return g_err.c_str(); }

int cfd_default_params(cfd_params* p) {
    if (!p) return fail(CFD_E_ARG, "cfd_default_params: null pointer");
    std::memset(p, 0, sizeof *p);
    p->nx = 301; p->ny = 151; p->nz = 1;
    p->dx = 1.6 / 301.0; p->dy = 0.8 / 151.0; p->dz = 1.0 / 1.0;
    p->dt = 0.00001;
    p->suth = 120.0;
    {
        // sutherlandLambda = initialMew*(initialTemperature + S)/(initialTemperature**1.5)
        const double t0 = 223.0;
        p->lambda = (0.1 * (t0 + 120.0)) / pow(t0, 1.5);
    }
    p->cv = 716.0;
    p->R = 1005.0 - 716.0;
    p->k = 0.025;
    p->dye_f = 0.001;
    p->device = -1;
    p->flags = CFD_FLAG_DEFAULT;
    p->ghost_lo = p->ghost_hi = 0;
    return CFD_OK;
}

void cfd_destroy(cfd_handle* h) {
    if (!h) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& e : h->ev_used) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : h->ev_free) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int i = 0; i < 4; ++i)
        if (h->buf[i]) (void)hipFree(h->buf[i]);
    if (h->d_active) (void)hipFree(h->d_active);
    if (h->d_partial) (void)hipFree(h->d_partial);
    if (h->stream && h->own_stream) { cfd_internal_forget_stream(h->device, (void*)h->stream); (void)hipStreamDestroy(h->stream); }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete h;
}

// For the library's other translation unit (cfd_multi.hip): cfd_create on a stream the caller owns, so that a slab
// never creates a stream of its own only to destroy it again (a process that keeps creating and destroying streams
// hands hardware queues on between them).
static int create_impl(const cfd_params* p, const double* const fields[CFD_NPROP], const uint8_t* active,
                       bool own_stream, hipStream_t stream, cfd_handle** out);

int cfd_internal_create_on(const cfd_params* p, const double* const fields[CFD_NPROP], const uint8_t* active,
                           void* stream, cfd_handle** out) {
    return create_impl(p, fields, active, false, (hipStream_t)stream, out);
}

// For cfd_multi.hip: this handle runs beside others on its device and should plan its launches for `cus` compute units.
int cfd_internal_set_cu_share(cfd_handle* h, int cus) {
    if (!h || cus < 1) return fail(CFD_E_ARG, "cfd_internal_set_cu_share: bad argument");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_internal_set_cu_share: a split sweep or step is open");
    h->cus = cus;
    plan_chunks(h);
    return CFD_OK;
}

int cfd_create(const cfd_params* p, const double* const fields[CFD_NPROP], const uint8_t* active,
               cfd_handle** out) {
    return create_impl(p, fields, active, true, nullptr, out);
}

}  // extern "C"

static int create_impl(const cfd_params* p, const double* const fields[CFD_NPROP], const uint8_t* active,
                       bool own_stream, hipStream_t stream, cfd_handle** out) {
    if (!p || !fields || !active || !out) return fail(CFD_E_ARG, "cfd_create: null pointer");
    *out = nullptr;
    if (p->nx < 1 || p->ny < 1 || p->nz < 1) return fail(CFD_E_ARG, "cfd_create: grid extents must be >= 1");
    if (!(p->dx > 0) || !(p->dy > 0) || !(p->dz > 0) || !(p->dt > 0))
        return fail(CFD_E_ARG, "cfd_create: dx, dy, dz, dt must be > 0");
    for (int i = 0; i < CFD_NPROP; ++i)
        if (!fields[i]) return fail(CFD_E_ARG, "cfd_create: null field pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(CFD_E_DEVICE, "cfd_create: no HIP device available (this library has no CPU path)");
    int dev = p->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(CFD_E_ARG, "cfd_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(dev));

    cfd_handle* h = new cfd_handle();
    h->p = *p;
    h->device = dev;
    h->plane = (size_t)p->ny * p->nz;
    const size_t n = (size_t)p->nx * h->plane;
    // 2-D specialisation applies when nz == 1 and W is +0 in every owned cell.
    bool w_zero = true;
    {
        const uint64_t* wb = reinterpret_cast<const uint64_t*>(fields[2]);
        for (size_t i = 0; i < n; ++i)
            if (wb[i] != 0) { w_zero = false; break; }
    }
    h->two_d = (p->nz == 1) && w_zero && !(p->flags & CFD_FLAG_FORCE_3D);
    h->nf = h->two_d ? 5 : 6;
    h->slot_elems = (size_t)h->nf * (p->nx + 2 * GHOST) * h->plane;
    h->K = make_consts(p->dx, p->dy, p->dz, p->dt, p->lambda, p->suth, p->R, p->cv, p->k, p->dye_f);
    h->KF = make_fast_consts(h->K, p->dx, p->dy, p->dz);
    h->march3_ok = (double)NF3 * (double)h->plane * 8.0 < 4.0e9;
    h->cus = 256;
    if (hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || h->cus < 1) h->cus = 256;
    h->cus_device = h->cus;
    h->xcd_fill = !(getenv("CFD_TRIO_XCD_FILL") && atoi(getenv("CFD_TRIO_XCD_FILL")) == 0);
    // kernels with large LDS tiles run only where a workgroup may have that much (160 KiB on gfx950); elsewhere the
    // handle falls back to the forms that need little: the per-cell strict 3-D kernel, the two-sweeps 2-D kernel
    if (hipDeviceGetAttribute(&h->lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) != hipSuccess || h->lds_max < 1)
        h->lds_max = 64 * 1024;
    h->march3_strict_ok = h->march3_ok && march3_lds_bytes<MathStrict3>(256) <= (size_t)h->lds_max;
    h->pair_enabled = getenv("CFD_PAIR") ? atoi(getenv("CFD_PAIR")) != 0 : true;
    h->trio_enabled = h->pair_enabled && (getenv("CFD_TRIO") ? atoi(getenv("CFD_TRIO")) != 0 : true);
    h->trio_nts = getenv("CFD_TRIO_NTS") && atoi(getenv("CFD_TRIO_NTS")) == 128 ? 128 : 256;
    if ((size_t)TRIO_LDS_ROWS * h->trio_nts * sizeof(double) > (size_t)h->lds_max) h->trio_enabled = false;
    plan_chunks(h);
    h->cmt = 0; h->cat = 0;
    h->chunk = 32;
    h->chunk_auto = true;
    if (const char* e = getenv("CFD_CHUNK")) { int v = atoi(e); if (v >= 1) { h->chunk = v; h->chunk_auto = false; } }
    h->variant = -1;
    if (const char* e = getenv("CFD_VARIANT")) h->variant = atoi(e);
    for (int i = 0; i < 4; ++i) { h->buf[i] = nullptr; h->map[i] = i; }
    h->ghost_mask_depth = 0;
    h->d_active = nullptr; h->d_partial = nullptr; h->stream = stream; h->own_stream = own_stream;
    h->timing = false;
    for (int kd = 0; kd < NKIND; ++kd) { h->t_ms[kd] = 0.0; h->t_launches[kd] = 0; }
    h->mew.assign(fields[5], fields[5] + n);

#define CREATE_TRY(expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            std::string m = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
            cfd_destroy(h);                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? CFD_E_MEMORY : CFD_E_DEVICE, m);           \
        }                                                                                      \
    } while (0)

    if (int rc = scratch_setup(dev)) { cfd_destroy(h); return rc; }
    if (own_stream) CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; ++i) {
        CREATE_TRY(hipMalloc((void**)&h->buf[i], h->slot_elems * sizeof(double)));
        CREATE_TRY(hipMemsetAsync(h->buf[i], 0, h->slot_elems * sizeof(double), h->stream));
    }
    CREATE_TRY(hipMalloc((void**)&h->d_active, (size_t)(p->nx + 2 * GHOST) * h->plane));
    CREATE_TRY(hipMemsetAsync(h->d_active, 0, (size_t)(p->nx + 2 * GHOST) * h->plane, h->stream));
    h->n_partial = SCRATCH_PARTIALS;     // cfd_total_density_noalloc uses the same geometry
    CREATE_TRY(hipMalloc((void**)&h->d_partial, (h->n_partial + 1) * sizeof(double)));
    CREATE_TRY(hipMemcpyAsync(h->d_active + GHOST * h->plane, active, n, hipMemcpyHostToDevice, h->stream));
    // slot 0 from the host fields, plane-interleaved; then slots 1, 2 and scratch as device copies
    const int* prop = h->two_d ? kProp2 : kProp3;
    const size_t row = h->plane * sizeof(double);
    for (int f = 0; f < h->nf; ++f) {
        double* d = h->buf[0] + ((size_t)GHOST * h->nf + f) * h->plane;
        CREATE_TRY(hipMemcpy2DAsync(d, (size_t)h->nf * row, fields[prop[f]], row, row, p->nx,
                                    hipMemcpyHostToDevice, h->stream));
    }
    for (int i = 1; i < 4; ++i)
        CREATE_TRY(hipMemcpyAsync(h->buf[i], h->buf[0], h->slot_elems * sizeof(double), hipMemcpyDeviceToDevice,
                                  h->stream));
    CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
    *out = h;
    return CFD_OK;
}

static int sweep_in_place(cfd_handle* h, bool with_filter, bool* did_fuse);

extern "C" {

int cfd_sweep(cfd_handle* h, int step_time) {
    if (!h) return fail(CFD_E_ARG, "cfd_sweep: null handle");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_sweep: a split sweep is open (call cfd_sweep_end first)");
    HIP_TRY(hipSetDevice(h->device));
    const double* src = slot_base(h, h->cmt);
    if (step_time) {
        const int nc = (h->cmt + 1) % 3;
        int rc = launch_sweep(h, src, slot_base(h, nc));
        if (rc) return rc;
        h->cmt = nc;
        h->cat += 1;
    } else {
        return sweep_in_place(h, false, nullptr);
    }
    return CFD_OK;
}

int cfd_time_filter(cfd_handle* h) {
    if (!h) return fail(CFD_E_ARG, "cfd_time_filter: null handle");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_time_filter: a split sweep is open (call cfd_sweep_end first)");
    if (!(h->cat > 1)) return CFD_OK;
    HIP_TRY(hipSetDevice(h->device));
    const int tb = (h->cmt + 2) % 3, tc = (tb + 2) % 3;
    // Density and Temperature are always filtered; U, V, W, Dye only in active cells
    const unsigned always = h->two_d ? ((1u << F2_RHO) | (1u << F2_T)) : ((1u << 3) | (1u << 4));
    dim3 block(256), grid((unsigned)((h->plane + 255) / 256), h->nf, h->p.nx);
    hipLaunchKernelGGL(time_filter_kernel, grid, block, 0, h->stream, slot_base(h, h->cmt), slot_base(h, tc),
                       slot_base(h, tb), h->d_active + GHOST * h->plane, (long)h->plane, h->nf, always);
    HIP_TRY(hipGetLastError());
    return CFD_OK;
}

// A sweep that does not advance time, optionally fused with the time filter.
static int sweep_in_place(cfd_handle* h, bool with_filter, bool* did_fuse) {
    const double* src = slot_base(h, h->cmt);
    const double* fpp = nullptr;
    double* fprev = nullptr;
    if (with_filter && h->cat > 1) {
        const int tb = (h->cmt + 2) % 3, tc = (tb + 2) % 3;
        fpp = slot_base(h, tc);
        fprev = slot_base(h, tb);
    }
    int rc = launch_sweep(h, src, slot_base(h, 3), fpp, fprev, did_fuse);
    if (rc) return rc;
    const int t = h->map[h->cmt];
    h->map[h->cmt] = h->map[3];
    h->map[3] = t;
    return CFD_OK;
}

int cfd_sweep_and_filter(cfd_handle* h) {
    if (!h) return fail(CFD_E_ARG, "cfd_sweep_and_filter: null handle");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_sweep_and_filter: a split sweep is open (call cfd_sweep_end first)");
    HIP_TRY(hipSetDevice(h->device));
    bool fused = false;
    int rc = sweep_in_place(h, true, &fused);
    if (rc) return rc;
    return fused ? CFD_OK : cfd_time_filter(h);
}

int cfd_sweep_pair_and_filter(cfd_handle* h) {
    if (!h) return fail(CFD_E_ARG, "cfd_sweep_pair_and_filter: null handle");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_sweep_pair_and_filter: a split sweep is open (call cfd_sweep_end first)");
    HIP_TRY(hipSetDevice(h->device));
    if (!pair_ok(h)) {
        // Two separate sweeps need the ghost planes refreshed in between, which only the caller can do.
        if (h->p.ghost_lo || h->p.ghost_hi)
            return fail(CFD_E_STATE, "cfd_sweep_pair_and_filter: not available for this slab (see cfd_pair_available); "
                                     "run cfd_sweep and cfd_sweep_and_filter with an exchange before each");
        int rc = cfd_sweep(h, 0);
        return rc ? rc : cfd_sweep_and_filter(h);
    }
    const double* fpp = nullptr;
    double* fprev = nullptr;
    if (h->cat > 1) {
        const int tb = (h->cmt + 2) % 3, tc = (tb + 2) % 3;
        fpp = slot_base(h, tc);
        fprev = slot_base(h, tb);
    }
    int rc = launch_pair(h, slot_base(h, h->cmt), slot_base(h, 3), fpp, fprev);
    if (rc) return rc;
    const int t = h->map[h->cmt];
    h->map[h->cmt] = h->map[3];
    h->map[3] = t;
    return CFD_OK;
}

int cfd_sweep_begin(cfd_handle* h, int step_time, int with_filter) {
    return cfd_sweep_begin_depth(h, step_time, with_filter, 1);
}

int cfd_sweep_begin_depth(cfd_handle* h, int step_time, int with_filter, int edge_planes) {
    if (!h) return fail(CFD_E_ARG, "cfd_sweep_begin: null handle");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_sweep_begin: a split sweep is already open");
    if (step_time && with_filter) return fail(CFD_E_ARG, "cfd_sweep_begin: the filter follows a sweep that does not advance time");
    if (edge_planes < 1) return fail(CFD_E_ARG, "cfd_sweep_begin_depth: edge_planes must be >= 1");
    HIP_TRY(hipSetDevice(h->device));
    auto& q = h->pend;
    q.trio = false;
    q.step_time = step_time != 0;
    q.with_filter = with_filter != 0;
    q.fused = false;
    q.src = slot_base(h, h->cmt);
    q.dst = slot_base(h, step_time ? (h->cmt + 1) % 3 : 3);
    q.fpp = nullptr;
    q.fprev = nullptr;
    if (with_filter && h->cat > 1) {
        const int tb = (h->cmt + 2) % 3, tc = (tb + 2) % 3;
        q.fpp = slot_base(h, tc);
        q.fprev = slot_base(h, tb);
    }
    // planes 1 .. nx-2 read owned planes only; plane 0 / nx-1 also read a ghost plane if there is one.
    // edge_planes > 1 leaves that many planes on every side with a neighbour to cfd_sweep_end.
    q.x0 = h->p.ghost_lo ? (edge_planes < h->p.nx ? edge_planes : h->p.nx) : 0;
    q.x1 = h->p.ghost_hi ? h->p.nx - edge_planes : h->p.nx;
    if (q.x1 < q.x0) q.x1 = q.x0;
    int rc = launch_sweep(h, q.src, q.dst, q.fpp, q.fprev, &q.fused, q.x0, q.x1);
    if (rc) return rc;
    q.open = true;
    return CFD_OK;
}

int cfd_sweep_end(cfd_handle* h) {
    if (!h) return fail(CFD_E_ARG, "cfd_sweep_end: null handle");
    return cfd_sweep_end_on(h, (void*)h->stream);
}

int cfd_sweep_end_on(cfd_handle* h, void* stream) {
    if (!h) return fail(CFD_E_ARG, "cfd_sweep_end: null handle");
    if (!h->pend.open || h->pend.trio) return fail(CFD_E_STATE, "cfd_sweep_end: no split sweep is open");
    HIP_TRY(hipSetDevice(h->device));
    auto& q = h->pend;
    if (q.with_filter && (hipStream_t)stream != h->stream)
        return fail(CFD_E_ARG, "cfd_sweep_end_on: a sweep with the filter completes on the handle's own stream");
    // the edge launches go to the stream named by the caller; the handle's own stream is untouched
    const hipStream_t own = h->stream;
    h->stream = (hipStream_t)stream;
    int rc = launch_sweep(h, q.src, q.dst, q.fpp, q.fprev, &q.fused, 0, q.x0);
    if (!rc) rc = launch_sweep(h, q.src, q.dst, q.fpp, q.fprev, &q.fused, q.x1, h->p.nx);
    h->stream = own;
    if (rc) return rc;
    q.open = false;
    if (q.step_time) {
        h->cmt = (h->cmt + 1) % 3;
        h->cat += 1;
    } else {
        const int t = h->map[h->cmt];
        h->map[h->cmt] = h->map[3];
        h->map[3] = t;
    }
    if (q.with_filter && !(q.fpp && q.fused)) return cfd_time_filter(h);
    return CFD_OK;
}

int cfd_step(cfd_handle* h, int nsteps) {
    if (!h) return fail(CFD_E_ARG, "cfd_step: null handle");
    if (nsteps < 0) return fail(CFD_E_ARG, "cfd_step: nsteps must be >= 0");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_step: a split sweep is open (call cfd_sweep_end first)");
    // A slab's ghost planes must be refreshed by the host before every sweep (cfd_halo_planes*), which a
    // whole step in one call cannot do: its second sweep would read planes the neighbour has not sent.
    if (h->p.ghost_lo || h->p.ghost_hi)
        return fail(CFD_E_STATE, "cfd_step: this handle is a slab (ghost_lo / ghost_hi); drive it sweep by sweep with a "
                                 "halo exchange before each (cfd_halo_planes, cfd_sweep_begin / cfd_sweep_end)");
    HIP_TRY(hipSetDevice(h->device));
    for (int i = 0; i < nsteps; ++i) {
        // The filter after the first and second sweep of a step writes values that the filter
        // after the third overwrites from the same inputs (slot t-2 and the mask do not change,
        // slot t-1 is not read by any sweep), so only the last of the three is applied -- inside
        // the third sweep's kernel where that kernel supports it, as its own launch otherwise.
        int rc;
        if (trio_ok(h)) {
            if ((rc = step_trio(h))) return rc;
            continue;
        }
        if ((rc = cfd_sweep(h, 1))) return rc;
        if ((rc = cfd_sweep_pair_and_filter(h))) return rc;     // falls back to two launches where it must
    }
    return CFD_OK;
}

int cfd_total_density(cfd_handle* h, double* out) {
    if (!h || !out) return fail(CFD_E_ARG, "cfd_total_density: null pointer");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_total_density: a split sweep is open (call cfd_sweep_end first)");
    HIP_TRY(hipSetDevice(h->device));
    const int t = (h->cmt + 2) % 3;
    const int f = h->two_d ? F2_RHO : 3;
    const long total = (long)h->p.nx * (long)h->plane;
    int nb = (int)((total + 255) / 256);
    if (nb > h->n_partial) nb = h->n_partial;
    hipLaunchKernelGGL(sum_field_stage1, dim3(nb), dim3(256), 0, h->stream, slot_base(h, t), (long)h->plane, h->nf,
                       f, h->p.nx, h->d_partial);
    hipLaunchKernelGGL(sum_field_stage2, dim3(1), dim3(256), 0, h->stream, h->d_partial, nb,
                       h->d_partial + h->n_partial);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->d_partial + h->n_partial, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CFD_OK;
}

int cfd_total_density_noalloc(cfd_handle* h, double* d_out) {
    if (!h || !d_out) return fail(CFD_E_ARG, "cfd_total_density_noalloc: null pointer");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_total_density_noalloc: a split sweep is open (call cfd_sweep_end first)");
    const long total = (long)h->p.nx * (long)h->plane;
    if (total > CFD_NOALLOC_MAX_CELLS)
        return fail(CFD_E_ARG, "cfd_total_density_noalloc: more than CFD_NOALLOC_MAX_CELLS (2^31) owned cells");
    double* part = g_scratch[h->device];      // set by cfd_create on this device
    if (!part) return fail(CFD_E_STATE, "cfd_total_density_noalloc: the device's scratch array was not set up");
    HIP_TRY(hipSetDevice(h->device));
    const int t = (h->cmt + 2) % 3;
    const int f = h->two_d ? F2_RHO : 3;
    int nb = (int)((total + 255) / 256);        // the same launch geometry as cfd_total_density: the same bits
    if (nb > SCRATCH_PARTIALS) nb = SCRATCH_PARTIALS;
    hipLaunchKernelGGL(sum_field_stage1, dim3(nb), dim3(256), 0, h->stream, slot_base(h, t), (long)h->plane, h->nf,
                       f, h->p.nx, part);
    hipLaunchKernelGGL(sum_field_stage2, dim3(1), dim3(256), 0, h->stream, part, nb, d_out);
    HIP_TRY(hipGetLastError());
    return CFD_OK;
}

int cfd_read_fields(cfd_handle* h, int slot_rel, double* const out[CFD_NPROP]) {
    if (!h || !out) return fail(CFD_E_ARG, "cfd_read_fields: null pointer");
    if (slot_rel < 0 || slot_rel > 2) return fail(CFD_E_ARG, "cfd_read_fields: slot_rel must be 0, 1 or 2");
    for (int i = 0; i < CFD_NPROP; ++i)
        if (!out[i]) return fail(CFD_E_ARG, "cfd_read_fields: null output pointer");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_read_fields: a split sweep is open (call cfd_sweep_end first)");
    HIP_TRY(hipSetDevice(h->device));
    const int t = ((h->cmt - slot_rel) % 3 + 3) % 3;
    const int* prop = h->two_d ? kProp2 : kProp3;
    const size_t row = h->plane * sizeof(double);
    const size_t n = (size_t)h->p.nx * h->plane;
    for (int f = 0; f < h->nf; ++f) {
        const double* s = slot_base(h, t) + (size_t)f * h->plane;
        HIP_TRY(hipMemcpy2DAsync(out[prop[f]], row, s, (size_t)h->nf * row, row, h->p.nx, hipMemcpyDeviceToHost,
                                 h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->two_d) std::memset(out[2], 0, n * sizeof(double));   // W is exactly +0 throughout
    std::memcpy(out[5], h->mew.data(), n * sizeof(double));
    return CFD_OK;
}

int cfd_write_fields(cfd_handle* h, int slot_rel, const double* const in[CFD_NPROP]) {
    if (!h || !in) return fail(CFD_E_ARG, "cfd_write_fields: null pointer");
    if (slot_rel < 0 || slot_rel > 2) return fail(CFD_E_ARG, "cfd_write_fields: slot_rel must be 0, 1 or 2");
    for (int i = 0; i < CFD_NPROP; ++i)
        if (!in[i]) return fail(CFD_E_ARG, "cfd_write_fields: null input pointer");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_write_fields: a split sweep is open (call cfd_sweep_end first)");
    const size_t n = (size_t)h->p.nx * h->plane;
    if (h->two_d) {
        // the handle holds no W at all: a W that is not identically +0 cannot be represented
        const uint64_t* wb = reinterpret_cast<const uint64_t*>(in[2]);
        for (size_t i = 0; i < n; ++i)
            if (wb[i] != 0)
                return fail(CFD_E_STATE, "cfd_write_fields: W is not identically +0 but the handle was created in its "
                                         "2-D form (nz == 1, W == 0); create it with CFD_FLAG_FORCE_3D");
    }
    HIP_TRY(hipSetDevice(h->device));
    const int t = ((h->cmt - slot_rel) % 3 + 3) % 3;
    const int* prop = h->two_d ? kProp2 : kProp3;
    const size_t row = h->plane * sizeof(double);
    for (int f = 0; f < h->nf; ++f) {
        double* d = slot_base(h, t) + (size_t)f * h->plane;
        HIP_TRY(hipMemcpy2DAsync(d, (size_t)h->nf * row, in[prop[f]], row, row, h->p.nx, hipMemcpyHostToDevice,
                                 h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (slot_rel == 0) h->mew.assign(in[5], in[5] + n);     // stored Mew: one copy serves every slot (never updated)
    return CFD_OK;
}

int cfd_set_time(cfd_handle* h, int curr_mod_time, int curr_abs_time) {
    if (!h) return fail(CFD_E_ARG, "cfd_set_time: null handle");
    if (curr_mod_time < 0 || curr_mod_time > 2) return fail(CFD_E_ARG, "cfd_set_time: curr_mod_time must be 0, 1 or 2");
    if (curr_abs_time < 0) return fail(CFD_E_ARG, "cfd_set_time: curr_abs_time must be >= 0");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_set_time: a split sweep is open (call cfd_sweep_end first)");
    h->cmt = curr_mod_time;
    h->cat = curr_abs_time;
    return CFD_OK;
}

int cfd_property_slice(cfd_handle* h, int property, int z_level, int slot_rel, double* out) {
    if (!h || !out) return fail(CFD_E_ARG, "cfd_property_slice: null pointer");
    if (property < CFD_PROPERTY_SPEED || property > CFD_PROPERTY_DYE)
        return fail(CFD_E_ARG, "cfd_property_slice: property must be 0..9 (Speed .. Dye)");
    if (z_level < 0 || z_level >= h->p.nz) return fail(CFD_E_ARG, "cfd_property_slice: z_level outside the grid");
    if (slot_rel < 0 || slot_rel > 2) return fail(CFD_E_ARG, "cfd_property_slice: slot_rel must be 0, 1 or 2");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_property_slice: a split sweep is open (call cfd_sweep_end first)");
    HIP_TRY(hipSetDevice(h->device));
    const int t = ((h->cmt - slot_rel) % 3 + 3) % 3;
    SliceFields F;
    F.base = slot_base(h, t);
    F.plane = (long)h->plane;
    F.nf = h->nf;
    F.nz = h->p.nz;
    if (h->two_d) { F.iU = F2_U; F.iV = F2_V; F.iW = -1; F.iRho = F2_RHO; F.iT = F2_T; F.iDye = F2_DYE; }
    else { F.iU = 0; F.iV = 1; F.iW = 2; F.iRho = 3; F.iT = 4; F.iDye = 5; }
    F.xlo = h->p.ghost_lo ? -1 : 0;
    F.xhi = h->p.ghost_hi ? h->p.nx : h->p.nx - 1;
    // the scratch slot is free between sweeps and larger than a slice: it stages the result
    double* stage = h->buf[h->map[3]];
    const long n = (long)h->p.nx * h->p.ny;
    hipLaunchKernelGGL(property_slice_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->K, F,
                       h->p.nx, h->p.ny, z_level, property, stage);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, stage, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CFD_OK;
}

int cfd_time(cfd_handle* h, int* cmt, int* cat) {
    if (!h || !cmt || !cat) return fail(CFD_E_ARG, "cfd_time: null pointer");
    *cmt = h->cmt;
    *cat = h->cat;
    return CFD_OK;
}

int cfd_halo_planes_depth(cfd_handle* h, int depth, void** send_lo, void** send_hi, void** recv_lo, void** recv_hi,
                          size_t* bytes) {
    if (!h) return fail(CFD_E_ARG, "cfd_halo_planes_depth: null handle");
    if (depth < 1 || depth > GHOST) return fail(CFD_E_ARG, "cfd_halo_planes_depth: depth must be 1, 2 or 3");
    if (depth > h->p.nx) return fail(CFD_E_ARG, "cfd_halo_planes_depth: the slab has fewer planes than the depth asked for");
    double* base = slot_base(h, h->cmt);
    const size_t pe = (size_t)h->nf * h->plane;
    if (send_lo) *send_lo = base;
    if (send_hi) *send_hi = base + (size_t)(h->p.nx - depth) * pe;
    if (recv_lo) *recv_lo = h->p.ghost_lo ? (void*)(base - (size_t)depth * pe) : nullptr;
    if (recv_hi) *recv_hi = h->p.ghost_hi ? (void*)(base + (size_t)h->p.nx * pe) : nullptr;
    if (bytes) *bytes = (size_t)depth * pe * sizeof(double);
    return CFD_OK;
}

int cfd_halo_planes(cfd_handle* h, void** send_lo, void** send_hi, void** recv_lo, void** recv_hi,
                    size_t* plane_bytes) {
    if (!h) return fail(CFD_E_ARG, "cfd_halo_planes: null handle");
    return cfd_halo_planes_depth(h, 1, send_lo, send_hi, recv_lo, recv_hi, plane_bytes);
}

int cfd_set_ghost_active(cfd_handle* h, const uint8_t* lo_plane, const uint8_t* hi_plane) {
    return cfd_set_ghost_active_depth(h, 1, lo_plane, hi_plane);
}

int cfd_set_ghost_active_depth(cfd_handle* h, int depth, const uint8_t* lo_planes, const uint8_t* hi_planes) {
    if (!h) return fail(CFD_E_ARG, "cfd_set_ghost_active: null handle");
    if (depth < 1 || depth > 2) return fail(CFD_E_ARG, "cfd_set_ghost_active_depth: depth must be 1 or 2");
    if ((h->p.ghost_lo && !lo_planes) || (h->p.ghost_hi && !hi_planes))
        return fail(CFD_E_ARG, "cfd_set_ghost_active: mask planes are needed on every side that has a ghost plane");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_set_ghost_active: a split sweep is open (call cfd_sweep_end first)");
    HIP_TRY(hipSetDevice(h->device));
    uint8_t* own = h->d_active + GHOST * h->plane;
    const size_t bytes = (size_t)depth * h->plane;
    // both in ascending x: lo_planes = planes -depth .. -1, hi_planes = planes nx .. nx + depth - 1
    if (h->p.ghost_lo) HIP_TRY(hipMemcpyAsync(own - bytes, lo_planes, bytes, hipMemcpyHostToDevice, h->stream));
    if (h->p.ghost_hi)
        HIP_TRY(hipMemcpyAsync(own + (size_t)h->p.nx * h->plane, hi_planes, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->ghost_mask_depth = depth;
    return CFD_OK;
}

int cfd_pair_available(cfd_handle* h, int* available) {
    if (!h || !available) return fail(CFD_E_ARG, "cfd_pair_available: null pointer");
    *available = pair_ok(h) ? 1 : 0;
    return CFD_OK;
}

int cfd_one_pass_step_available(cfd_handle* h, int* available) {
    if (!h || !available) return fail(CFD_E_ARG, "cfd_one_pass_step_available: null pointer");
    *available = trio_ok(h) ? 1 : 0;
    return CFD_OK;
}

int cfd_step_one_pass(cfd_handle* h) {
    if (!h) return fail(CFD_E_ARG, "cfd_step_one_pass: null handle");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_step_one_pass: a split sweep or step is open");
    if (!trio_ok(h)) return fail(CFD_E_STATE, "cfd_step_one_pass: not available for this handle (see cfd_one_pass_step_available)");
    HIP_TRY(hipSetDevice(h->device));
    return step_trio(h);
}

// Chunks of the one-pass step whose blocks read owned planes only: a block of planes [xs, xe) reads planes xs - 3 .. xe + 2.
static void trio_interior_chunks(const cfd_handle* h, int* c0, int* c1) {
    const int nx = h->p.nx, nchunk = trio_nchunk(h);
    int a = 0, b = nchunk, xs, xe;
    if (h->p.ghost_lo)
        for (; a < nchunk; ++a) { trio_span(h, a, &xs, &xe); if (xs >= 3) break; }
    if (h->p.ghost_hi)
        for (; b > a; --b) { trio_span(h, b - 1, &xs, &xe); if (xe + 2 <= nx - 1) break; }
    *c0 = a;
    *c1 = b;
}

int cfd_step_one_pass_begin(cfd_handle* h) {
    if (!h) return fail(CFD_E_ARG, "cfd_step_one_pass_begin: null handle");
    if (h->pend.open) return fail(CFD_E_STATE, "cfd_step_one_pass_begin: a split sweep or step is already open");
    if (!trio_ok(h)) return fail(CFD_E_STATE, "cfd_step_one_pass_begin: not available for this handle (see cfd_one_pass_step_available)");
    HIP_TRY(hipSetDevice(h->device));
    auto& q = h->pend;
    trio_interior_chunks(h, &q.c0, &q.c1);
    h->interior_stream = h->stream;
    int rc = step_trio(h, q.c0, q.c1, false);
    if (rc) return rc;
    q.trio = true;
    q.open = true;
    return CFD_OK;
}

int cfd_step_one_pass_end_on(cfd_handle* h, void* stream) {
    if (!h) return fail(CFD_E_ARG, "cfd_step_one_pass_end_on: null handle");
    if (!h->pend.open || !h->pend.trio) return fail(CFD_E_STATE, "cfd_step_one_pass_end_on: no split step is open");
    HIP_TRY(hipSetDevice(h->device));
    auto& q = h->pend;
    const int nchunk = trio_nchunk(h);
    const hipStream_t own = h->stream;
    h->stream = (hipStream_t)stream;
    // the chunks next to the lower and to the upper neighbour in ONE launch: on one stream two launches would run one
    // after the other (empty ranges still complete the step)
    const int interior_blocks = (q.c1 - q.c0) * trio_tiles(h->p.ny, trio_wmax(h->trio_nts), trio_gran()).ntile;
    int rc = step_trio(h, 0, q.c0, true, q.c1, nchunk, interior_blocks);
    h->stream = own;
    if (rc) return rc;
    q.open = false;
    q.trio = false;
    return CFD_OK;
}

int cfd_stream(cfd_handle* h, void** stream) {
    if (!h || !stream) return fail(CFD_E_ARG, "cfd_stream: null pointer");
    *stream = (void*)h->stream;
    return CFD_OK;
}

int cfd_set_stream(cfd_handle* h, void* stream) {
    if (!h) return fail(CFD_E_ARG, "cfd_set_stream: null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream) { cfd_internal_forget_stream(h->device, (void*)h->stream); HIP_TRY(hipStreamDestroy(h->stream)); }
    h->stream = (hipStream_t)stream;
    h->own_stream = false;
    return CFD_OK;
}

const char* cfd_kernel_name(cfd_handle* h) {
    if (!h) return "";
    const bool fast = (h->p.flags & CFD_FLAG_FAST_MATH) != 0;
    if (!h->two_d) {
        if ((h->p.flags & CFD_FLAG_SIMPLE_KERNEL) || !h->march3_ok || (!fast && (h->variant == 9 || (!h->march3_strict_ok && h->variant < 0))))
            return fast ? "sweep3d_simple<fast>" : "sweep3d_simple<strict>";
        if (fast && (h->variant < 1 || h->variant > 5)) return "sweep3d_stream<fast>";
        return fast ? "sweep3d_march<fast>" : "sweep3d_march<strict>";
    }
    if (h->p.flags & CFD_FLAG_SIMPLE_KERNEL) return fast ? "sweep2d_simple<fast>" : "sweep2d_simple<strict>";
    const int v = h->variant < 0 ? (fast ? 30 : 8) : h->variant;
    if (fast && !(v == 8 || v == 12 || v == 13)) return "sweep2d_stream<fast>";
    return fast ? "sweep2d_march2<fast>" : "sweep2d_march2<strict>";
}

int cfd_sync(cfd_handle* h) {
    if (!h) return fail(CFD_E_ARG, "cfd_sync: null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return CFD_OK;
}

int cfd_set_timing(cfd_handle* h, int enable) {
    if (!h) return fail(CFD_E_ARG, "cfd_set_timing: null handle");
    h->timing = enable != 0;
    return CFD_OK;
}

static int collect_timing(cfd_handle* h) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto& e : h->ev_used) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
        h->t_ms[e.kind] += ms;
        h->t_launches[e.kind] += 1;
        h->ev_free.push_back(e);
    }
    h->ev_used.clear();
    return CFD_OK;
}

int cfd_kernel_timing(cfd_handle* h, int kind, double* total_ms, int64_t* launches, int reset) {
    if (!h) return fail(CFD_E_ARG, "cfd_kernel_timing: null handle");
    if (kind < 0 || kind >= NKIND) return fail(CFD_E_ARG, "cfd_kernel_timing: kind must be 0, 1, 2 or 3");
    int rc = collect_timing(h);
    if (rc) return rc;
    if (total_ms) *total_ms = h->t_ms[kind];
    if (launches) *launches = h->t_launches[kind];
    if (reset) { h->t_ms[kind] = 0.0; h->t_launches[kind] = 0; }
    return CFD_OK;
}

int cfd_sweep_timing(cfd_handle* h, double* total_ms, int64_t* launches, int reset) {
    if (!h) return fail(CFD_E_ARG, "cfd_sweep_timing: null handle");
    return cfd_kernel_timing(h, 0, total_ms, launches, reset);
}

int cfd_live_fields(cfd_handle* h, int* nfields) {
    if (!h || !nfields) return fail(CFD_E_ARG, "cfd_live_fields: null pointer");
    *nfields = h->nf;
    return CFD_OK;
}

}  // extern "C"
#endif  // CFD_KERNEL_TEMPLATES_ONLY
