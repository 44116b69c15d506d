// Several slabs behind one handle: the x-slab decomposition of the field update driven from C, on top of the
// single-slab entry points of cfd_hip.hip (cfd_create, cfd_sweep_begin_depth / cfd_sweep_end_on,
// cfd_halo_planes_depth, cfd_sweep_pair_and_filter, cfd_total_density_noalloc).
//
// Two ways to own the slabs:
//   * cfd_multi_create       one host thread, all slabs of the grid in this process: one slab per listed device
//                            (RCCL transport, ncclCommInitAll) or any number of slabs per device (copy
//                            transport: device-to-device / peer copies; several "virtual" slabs on one GPU);
//   * cfd_multi_create_rank  one process per GPU (this process holds slab `rank` of `world`): RCCL transport
//                            with a communicator built from a broadcast ncclUniqueId (ncclCommInitRank).
// Transport = who moves the ghost planes: RCCL over xGMI (grouped ncclSend / ncclRecv between x-neighbours,
// ncclAllReduce for the density sum) or plain copies.  A plane of all live fields is one contiguous run in a
// time slot, so a halo is ONE message per neighbour with no pack kernel.
//
// Streams: per device one compute stream and one side stream (both at the default priority), shared by every slab on that
// device (several virtual slabs on one GPU therefore never hold more than two streams).  A step with the
// one-pass tail hides both of its exchanges behind the first sweep's interior planes:
//     side:  wait(previous step) -> exchange 1 plane -> edge planes of sweep 1 -> exchange 2 planes -> event
//     main:  interior planes of sweep 1 ....................................... wait(event) -> sweeps 2+3+filter
// Every cross-stream wait is on an event recorded BEFORE the wait is queued, by work that is already queued.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cfd_hip.h"

extern "C" int cfd_internal_fail(int code, const char* msg);     // cfd_hip.hip: sets cfd_last_error's text
// cfd_hip.hip: cfd_create on a stream the caller owns (the slab launches there from the start)
extern "C" int cfd_internal_one_pass_blocks(cfd_handle* h, int* blocks, int* cus);    // cfd_hip.hip
extern "C" void cfd_internal_forget_stream(int device, void* stream);   // cfd_hip.hip: per-stream measurement dropped
extern "C" int cfd_internal_set_cu_share(cfd_handle* h, int cus);   // cfd_hip.hip: plan launches for a share of the device
extern "C" int cfd_internal_create_on(const cfd_params* p, const double* const fields[CFD_NPROP], const uint8_t* active,
                                      void* stream, cfd_handle** out);

namespace {

int fail(int code, const std::string& m) { return cfd_internal_fail(code, m.c_str()); }

#define M_HIP(expr)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(CFD_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define M_NCCL(expr)                                                                             \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) return fail(CFD_E_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define M_CFD(expr)                                                                              \
    do {                                                                                         \
        int rc_ = (expr);                                                                        \
        if (rc_ != CFD_OK) return rc_;                                                           \
    } while (0)

struct DeviceStreams {
    int device;
    hipStream_t main, side;
    std::vector<hipStream_t> rejected;      // candidates for `side` that shared `main`'s hardware queue; kept until the end
};

struct Slab {
    cfd_handle* h = nullptr;
    int rank = 0;               // position of the slab along x (0 = lowest planes)
    int device = 0;
    int x0 = 0, x1 = 0;         // global plane range [x0, x1)
    int ds = 0;                 // index into cfd_multi::streams
    ncclComm_t comm = nullptr;
    hipEvent_t done = nullptr;  // recorded on the compute stream when the slab's current slot is complete
    hipEvent_t edge = nullptr;  // recorded on the side stream when the edge planes of a split sweep are complete
    hipEvent_t side_done = nullptr;     // recorded on the side stream at the end of a step's boundary work
    double* d_sum = nullptr;    // one double: this slab's density sum, all-reduced in place
};

}  // namespace

struct cfd_multi {
    cfd_params p;               // the GLOBAL grid (nx = all planes)
    int world = 1;              // slabs along x, in all processes
    int transport = CFD_TRANSPORT_COPY;
    bool pair = false;          // the step's tail runs as one pass on every slab
    bool trio = false;          // the whole step runs as one pass on every slab (one three-plane exchange per step)
    bool overlap = true;
    // Rehearsal on one GPU (cfd_multi_create_rank with world == 1 and env CFD_MULTI_LOOPBACK=1): the one slab is its own
    // lower and upper neighbour, exchanging with itself through RCCL exactly as a middle rank exchanges with two others
    // (a periodic domain: for timing the step of a rank, not for comparison with the reference).
    bool loopback = false;
    int trio_one_round = -1;    // see trio_fits_one_round
    size_t plane = 0;
    std::vector<Slab> slab;     // the slabs this process holds, in rank order
    std::vector<DeviceStreams> streams;
    int cmt = 0, cat = 0;
    // measurement: duration of the first local slab's ghost-plane exchanges (event pairs around them on their stream)
    bool time_exchange = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> xev_used, xev_free;
    double x_ms = 0.0;
    long x_count = 0;
};

namespace {

hipStream_t main_of(const cfd_multi* m, const Slab& s) { return m->streams[s.ds].main; }
hipStream_t side_of(const cfd_multi* m, const Slab& s) { return m->streams[s.ds].side; }

// Ghost-plane exchange `depth` planes deep between x-neighbours, queued on the side (on_side) or compute stream
// of every slab.  Copy transport: the receiver pulls from its neighbour's send planes once `ready` of that
// neighbour (its event saying those planes are complete) has fired.  RCCL: one group of sends and receives; a
// rank's own stream order covers its send planes, the rendezvous covers the neighbour's.
enum Ready { READY_DONE, READY_EDGE };

int exchange(cfd_multi* m, int depth, bool on_side, Ready ready) {
    if (m->world == 1 && !m->loopback) return CFD_OK;
    struct Planes { void *slo, *shi, *rlo, *rhi; size_t bytes; };
    std::vector<Planes> pl(m->slab.size());
    for (size_t i = 0; i < m->slab.size(); ++i)
        M_CFD(cfd_halo_planes_depth(m->slab[i].h, depth, &pl[i].slo, &pl[i].shi, &pl[i].rlo, &pl[i].rhi, &pl[i].bytes));
    // timed: from the moment the first local slab's stream gets to the exchange (its own planes are complete) to the
    // arrival of its ghost planes, waiting for the neighbours included
    std::pair<hipEvent_t, hipEvent_t> xev{nullptr, nullptr};
    hipStream_t xst = nullptr;
    if (m->time_exchange) {
        const Slab& s0 = m->slab[0];
        M_HIP(hipSetDevice(s0.device));
        xst = on_side ? side_of(m, s0) : main_of(m, s0);
        if (m->xev_free.empty()) {
            M_HIP(hipEventCreate(&xev.first));
            M_HIP(hipEventCreate(&xev.second));
        } else {
            xev = m->xev_free.back();
            m->xev_free.pop_back();
        }
        M_HIP(hipEventRecord(xev.first, xst));
    }
    auto timed_end = [&]() -> int {
        if (!xev.first) return CFD_OK;
        M_HIP(hipSetDevice(m->slab[0].device));
        M_HIP(hipEventRecord(xev.second, xst));
        m->xev_used.push_back(xev);
        return CFD_OK;
    };
    if (m->transport == CFD_TRANSPORT_RCCL) {
        M_NCCL(ncclGroupStart());
        for (size_t i = 0; i < m->slab.size(); ++i) {
            const Slab& s = m->slab[i];
            hipStream_t st = on_side ? side_of(m, s) : main_of(m, s);
            const size_t count = pl[i].bytes / sizeof(double);
            if (m->loopback) {      // sends and receives of one peer match in order: last planes -> lower ghosts, first -> upper
                M_NCCL(ncclSend(pl[i].shi, count, ncclDouble, s.rank, s.comm, st));
                M_NCCL(ncclRecv(pl[i].rlo, count, ncclDouble, s.rank, s.comm, st));
                M_NCCL(ncclSend(pl[i].slo, count, ncclDouble, s.rank, s.comm, st));
                M_NCCL(ncclRecv(pl[i].rhi, count, ncclDouble, s.rank, s.comm, st));
                continue;
            }
            if (s.rank > 0) {
                M_NCCL(ncclSend(pl[i].slo, count, ncclDouble, s.rank - 1, s.comm, st));
                M_NCCL(ncclRecv(pl[i].rlo, count, ncclDouble, s.rank - 1, s.comm, st));
            }
            if (s.rank < m->world - 1) {
                M_NCCL(ncclSend(pl[i].shi, count, ncclDouble, s.rank + 1, s.comm, st));
                M_NCCL(ncclRecv(pl[i].rhi, count, ncclDouble, s.rank + 1, s.comm, st));
            }
        }
        M_NCCL(ncclGroupEnd());
        return timed_end();
    }
    for (size_t i = 0; i < m->slab.size(); ++i) {
        const Slab& s = m->slab[i];
        if (i == 1) M_CFD(timed_end());         // the first slab's copies are queued
        M_HIP(hipSetDevice(s.device));
        hipStream_t st = on_side ? side_of(m, s) : main_of(m, s);
        for (int side = 0; side < 2; ++side) {
            const int nb_rank = side == 0 ? s.rank - 1 : s.rank + 1;
            if (nb_rank < 0 || nb_rank >= m->world) continue;
            const size_t k = i + (side == 0 ? -1 : 1);          // all slabs are local with the copy transport
            const Slab& nb = m->slab[k];
            M_HIP(hipStreamWaitEvent(st, ready == READY_EDGE ? nb.edge : nb.done, 0));
            void* dst = side == 0 ? pl[i].rlo : pl[i].rhi;
            const void* src = side == 0 ? pl[k].shi : pl[k].slo;
            if (nb.device == s.device)
                M_HIP(hipMemcpyAsync(dst, src, pl[i].bytes, hipMemcpyDeviceToDevice, st));
            else
                M_HIP(hipMemcpyPeerAsync(dst, s.device, src, nb.device, pl[i].bytes, st));
        }
    }
    if (m->slab.size() == 1) M_CFD(timed_end());
    return CFD_OK;
}

// folds the finished event pairs of the exchange timing into the accumulators (the streams must have run)
int drain_exchange_events(cfd_multi* m) {
    for (auto& e : m->xev_used) {
        float ms = 0.f;
        M_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        m->x_ms += ms;
        m->x_count += 1;
        m->xev_free.push_back(e);
    }
    m->xev_used.clear();
    return CFD_OK;
}

int record(cfd_multi* m, bool edge_event) {
    for (Slab& s : m->slab) {
        M_HIP(hipSetDevice(s.device));
        if (edge_event) M_HIP(hipEventRecord(s.edge, side_of(m, s)));
        else M_HIP(hipEventRecord(s.done, main_of(m, s)));
    }
    return CFD_OK;
}

// One sweep of every slab with its one-plane exchange.  Overlapped: the exchange runs on the side stream beside the
// launch of the planes that read no ghost plane (cells are independent within a sweep: applyDiffEq_adj maps over the
// nodes of one snapshot, src/Main.hs:211); the planes next to a neighbour follow on the compute stream once the ghost
// planes have arrived.  Otherwise everything is queued on the compute stream: exchange, interior, edge.
int sweep_all(cfd_multi* m, int step_time, int with_filter) {
    if (m->world == 1 && !m->loopback) {
        for (Slab& s : m->slab) M_CFD(with_filter ? cfd_sweep_and_filter(s.h) : cfd_sweep(s.h, step_time));
        return record(m, false);
    }
    if (m->overlap) {
        for (Slab& s : m->slab) {
            M_HIP(hipSetDevice(s.device));
            M_HIP(hipStreamWaitEvent(side_of(m, s), s.done, 0));    // the slot the sweep reads is complete
        }
        M_CFD(exchange(m, 1, true, READY_DONE));
        for (Slab& s : m->slab) M_CFD(cfd_sweep_begin_depth(s.h, step_time, with_filter, 1));
        // the planes next to a neighbour: on the side stream right behind the exchange, beside the interior launch (a
        // sweep fused with the filter completes on the compute stream only: there they follow the interior launch)
        if (!with_filter)
            for (Slab& s : m->slab) M_CFD(cfd_sweep_end_on(s.h, (void*)side_of(m, s)));
        for (Slab& s : m->slab) {
            M_HIP(hipSetDevice(s.device));
            M_HIP(hipEventRecord(s.side_done, side_of(m, s)));
            M_HIP(hipStreamWaitEvent(main_of(m, s), s.side_done, 0));
        }
        if (with_filter)
            for (Slab& s : m->slab) M_CFD(cfd_sweep_end(s.h));
        return record(m, false);
    }
    M_CFD(exchange(m, 1, false, READY_DONE));
    for (Slab& s : m->slab) M_CFD(cfd_sweep_begin_depth(s.h, step_time, with_filter, 1));
    for (Slab& s : m->slab) M_CFD(cfd_sweep_end(s.h));
    return record(m, false);
}

// whether every local slab's one-pass step is a single round of blocks on its device (decided once)
bool trio_fits_one_round(cfd_multi* m) {
    if (m->trio_one_round < 0) {
        m->trio_one_round = 1;
        for (Slab& s : m->slab) {
            int blocks = 0, cus = 0;
            if (cfd_internal_one_pass_blocks(s.h, &blocks, &cus) != CFD_OK || blocks > cus) m->trio_one_round = 0;
        }
    }
    return m->trio_one_round == 1;
}

int step_once(cfd_multi* m) {
    if (m->world == 1 && !m->loopback) {
        for (Slab& s : m->slab) M_CFD(cfd_step(s.h, 1));
        M_CFD(record(m, false));
    } else if (m->trio && !(m->overlap && trio_fits_one_round(m))) {
        // (also where a slab's blocks do not fit its device at once: beside an interior launch that holds every compute
        // unit for a whole block's time neither the exchange's kernel nor the late blocks could start; measured on a
        // 2048 x 16384 slab: 1.545 ms per step split against 1.425 undecomposed)
        M_CFD(exchange(m, 3, false, READY_DONE));
        for (Slab& s : m->slab) M_CFD(cfd_step_one_pass(s.h));
        M_CFD(record(m, false));
    } else if (m->trio) {
        // ONE exchange per step, three planes deep, on the side stream beside the blocks that read owned planes only;
        // the blocks next to a neighbour follow it on the side stream; the compute stream then waits for the side stream
        for (Slab& s : m->slab) {
            M_HIP(hipSetDevice(s.device));
            M_HIP(hipStreamWaitEvent(side_of(m, s), s.done, 0));    // the previous step of this slab
        }
        M_CFD(exchange(m, 3, true, READY_DONE));
        for (Slab& s : m->slab) M_CFD(cfd_step_one_pass_begin(s.h));
        for (Slab& s : m->slab) M_CFD(cfd_step_one_pass_end_on(s.h, (void*)side_of(m, s)));
        for (Slab& s : m->slab) {
            M_HIP(hipSetDevice(s.device));
            M_HIP(hipEventRecord(s.side_done, side_of(m, s)));
            M_HIP(hipStreamWaitEvent(main_of(m, s), s.side_done, 0));
        }
        M_CFD(record(m, false));
    } else if (!m->pair) {
        M_CFD(sweep_all(m, 1, 0));
        M_CFD(sweep_all(m, 0, 0));
        M_CFD(sweep_all(m, 0, 1));
    } else if (!m->overlap) {
        M_CFD(sweep_all(m, 1, 0));
        M_CFD(exchange(m, 2, false, READY_DONE));
        for (Slab& s : m->slab) M_CFD(cfd_sweep_pair_and_filter(s.h));
        M_CFD(record(m, false));
    } else {
        // boundary work of the step on the side stream, beside the interior planes of the first sweep
        for (Slab& s : m->slab) {
            M_HIP(hipSetDevice(s.device));
            M_HIP(hipStreamWaitEvent(side_of(m, s), s.done, 0));    // the previous step of this slab
        }
        M_CFD(exchange(m, 1, true, READY_DONE));
        for (Slab& s : m->slab) M_CFD(cfd_sweep_begin_depth(s.h, 1, 0, 2));       // compute stream: interior
        for (Slab& s : m->slab) M_CFD(cfd_sweep_end_on(s.h, (void*)side_of(m, s)));  // side stream: two planes per side
        M_CFD(record(m, true));
        M_CFD(exchange(m, 2, true, READY_EDGE));
        for (Slab& s : m->slab) {
            M_HIP(hipSetDevice(s.device));
            M_HIP(hipEventRecord(s.side_done, side_of(m, s)));
            M_HIP(hipStreamWaitEvent(main_of(m, s), s.side_done, 0));
        }
        for (Slab& s : m->slab) M_CFD(cfd_sweep_pair_and_filter(s.h));
        M_CFD(record(m, false));
    }
    m->cmt = (m->cmt + 1) % 3;
    m->cat += 1;
    return CFD_OK;
}

bool env_flag(const char* name, bool dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) != 0 : dflt;
}

// Do kernels on streams a and b run side by side?  HIP spreads its streams over a few hardware queues (four here; the
// fourth and the fifth stream of a process share one), and two streams on one queue run their kernels one after the
// other -- measured: a rank's one-pass step 1.16 ms instead of 0.70 when its compute and side stream shared a queue.
// A 300 us one-wave spin on a, a one-wave kernel on b: they overlap iff b's kernel is done while the spin still runs.
__global__ void multi_spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();        // 100 MHz
    // (bounded by a count of its own as well: it must end whatever the counter does)
    for (int i = 0; i < 4000000 && wall_clock64() - t0 < ticks; ++i) {}
}
__global__ void multi_touch_kernel() {}

int streams_overlap(hipStream_t a, hipStream_t b, bool* overlap) {
    hipEvent_t ea = nullptr, eb = nullptr;
    M_HIP(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    M_HIP(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    hipLaunchKernelGGL(multi_touch_kernel, dim3(1), dim3(64), 0, b);       // first launches load code, create queues
    hipLaunchKernelGGL(multi_spin_kernel, dim3(1), dim3(64), 0, a, 1LL);
    M_HIP(hipStreamSynchronize(a));
    M_HIP(hipStreamSynchronize(b));
    hipLaunchKernelGGL(multi_spin_kernel, dim3(1), dim3(64), 0, a, 30000LL);
    M_HIP(hipEventRecord(ea, a));
    hipLaunchKernelGGL(multi_touch_kernel, dim3(1), dim3(64), 0, b);
    M_HIP(hipEventRecord(eb, b));
    M_HIP(hipEventSynchronize(eb));
    *overlap = hipEventQuery(ea) == hipErrorNotReady;
    M_HIP(hipEventSynchronize(ea));
    (void)hipGetLastError();
    M_HIP(hipEventDestroy(ea));
    M_HIP(hipEventDestroy(eb));
    return CFD_OK;
}

// own_pair: a compute / side stream pair for this slab alone (at most two slabs on the device: four hardware queues per
// process); otherwise the slabs of a device share one pair.
int streams_for(cfd_multi* m, int device, int* index, bool own_pair = false) {
    for (size_t i = 0; i < m->streams.size() && !own_pair; ++i)
        if (m->streams[i].device == device) { *index = (int)i; return CFD_OK; }
    DeviceStreams d{device, nullptr, nullptr};
    M_HIP(hipSetDevice(device));
    // Both streams at the default priority, always: a step that mixes stream priorities in its cross-stream waits has
    // stalled on ROCm 7.2 / MI355X (after streams had been created and destroyed earlier in the process); with equal
    // priorities the same sequence of launches, copies and waits never did.
    M_HIP(hipStreamCreateWithFlags(&d.main, hipStreamNonBlocking));
    M_HIP(hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking));
    // a side stream on the compute stream's hardware queue would run nothing beside it: take the next stream that does
    for (int tries = 0; tries < 6 && env_flag("CFD_MULTI_QUEUE_CHECK", true); ++tries) {
        bool overlap = true;
        M_CFD(streams_overlap(d.main, d.side, &overlap));
        if (overlap) break;
        d.rejected.push_back(d.side);
        M_HIP(hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking));
    }
    m->streams.push_back(d);
    *index = (int)m->streams.size() - 1;
    return CFD_OK;
}

// Creates the slab `s` (rank, device, x0, x1 set) from the slab's own fields / mask (owned planes only).
// on_device: how many slabs of this handle run on the slab's device.
int make_slab(cfd_multi* m, Slab& s, const double* const fields[CFD_NPROP], const uint8_t* active, int extra_flags,
              int on_device = 1) {
    cfd_params q = m->p;
    q.nx = s.x1 - s.x0;
    q.device = s.device;
    q.flags = m->p.flags | extra_flags;
    q.ghost_lo = s.rank > 0 || m->loopback;
    q.ghost_hi = s.rank < m->world - 1 || m->loopback;
    // Two slabs on one device run side by side, each on streams of its own and planning its launches for half the
    // compute units; more than two share one pair of streams (their launches then follow one another).
    M_CFD(streams_for(m, s.device, &s.ds, on_device == 2));
    M_CFD(cfd_internal_create_on(&q, fields, active, (void*)main_of(m, s), &s.h));
    M_HIP(hipSetDevice(s.device));
    if (on_device == 2) {
        int cus = 0;
        M_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s.device));
        M_CFD(cfd_internal_set_cu_share(s.h, cus / 2 > 0 ? cus / 2 : 1));
    }
    M_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    M_HIP(hipEventCreateWithFlags(&s.edge, hipEventDisableTiming));
    M_HIP(hipEventCreateWithFlags(&s.side_done, hipEventDisableTiming));
    M_HIP(hipMalloc((void**)&s.d_sum, sizeof(double)));
    M_HIP(hipEventRecord(s.done, main_of(m, s)));
    return CFD_OK;
}

bool w_identically_zero(const double* w, size_t n) {
    const uint64_t* b = reinterpret_cast<const uint64_t*>(w);
    for (size_t i = 0; i < n; ++i)
        if (b[i] != 0) return false;
    return true;
}

}  // namespace

extern "C" {

void cfd_multi_destroy(cfd_multi* m) {
    if (!m) return;
    for (Slab& s : m->slab) {
        (void)hipSetDevice(s.device);
        if (s.h) { (void)cfd_sync(s.h); }
    }
    for (DeviceStreams& d : m->streams) {
        (void)hipSetDevice(d.device);
        if (d.side) (void)hipStreamSynchronize(d.side);
        if (d.main) (void)hipStreamSynchronize(d.main);
    }
    for (Slab& s : m->slab) {
        (void)hipSetDevice(s.device);
        if (s.comm) (void)ncclCommDestroy(s.comm);
        if (s.h) cfd_destroy(s.h);          // the handle does not own the streams it was pointed at
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.edge) (void)hipEventDestroy(s.edge);
        if (s.side_done) (void)hipEventDestroy(s.side_done);
        if (s.d_sum) (void)hipFree(s.d_sum);
    }
    for (auto* v : {&m->xev_used, &m->xev_free})
        for (auto& e : *v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (DeviceStreams& d : m->streams) {
        (void)hipSetDevice(d.device);
        for (hipStream_t r : d.rejected) (void)hipStreamDestroy(r);
        if (d.side) { cfd_internal_forget_stream(d.device, (void*)d.side); (void)hipStreamDestroy(d.side); }
        if (d.main) { cfd_internal_forget_stream(d.device, (void*)d.main); (void)hipStreamDestroy(d.main); }
    }
    delete m;
}

int cfd_multi_create(const cfd_params* p, int n_slabs, const int32_t* devices, int transport,
                     const double* const fields[CFD_NPROP], const uint8_t* active, cfd_multi** out) {
    if (!p || !fields || !active || !out) return fail(CFD_E_ARG, "cfd_multi_create: null pointer");
    *out = nullptr;
    if (n_slabs < 1) return fail(CFD_E_ARG, "cfd_multi_create: n_slabs must be >= 1");
    if (p->nx < n_slabs) return fail(CFD_E_ARG, "cfd_multi_create: fewer x-planes than slabs");
    if (transport != CFD_TRANSPORT_COPY && transport != CFD_TRANSPORT_RCCL)
        return fail(CFD_E_ARG, "cfd_multi_create: transport must be CFD_TRANSPORT_COPY or CFD_TRANSPORT_RCCL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(CFD_E_DEVICE, "cfd_multi_create: no HIP device available (this library has no CPU path)");
    std::vector<int> dev(n_slabs);
    for (int i = 0; i < n_slabs; ++i) {
        dev[i] = devices ? devices[i] : i % ndev;
        if (dev[i] < 0 || dev[i] >= ndev) return fail(CFD_E_ARG, "cfd_multi_create: device ordinal out of range");
    }
    if (transport == CFD_TRANSPORT_RCCL)
        for (int i = 0; i < n_slabs; ++i)
            for (int k = 0; k < i; ++k)
                if (dev[i] == dev[k])
                    return fail(CFD_E_ARG, "cfd_multi_create: the RCCL transport takes one slab per device (several slabs on "
                                           "one device: CFD_TRANSPORT_COPY)");
    cfd_multi* m = new cfd_multi();
    m->p = *p;
    m->world = n_slabs;
    m->transport = transport;
    m->plane = (size_t)p->ny * p->nz;
    m->overlap = env_flag("CFD_MULTI_OVERLAP", true);
    // The 2-D form (no W on the device) is a property of the WHOLE grid: a slab whose own W happens to be zero
    // must not drop W while a neighbour keeps it (their planes would differ in size).
    const bool two_d = p->nz == 1 && !(p->flags & CFD_FLAG_FORCE_3D) && w_identically_zero(fields[2], (size_t)p->nx * m->plane);
    const int extra = two_d ? 0 : (p->nz == 1 ? CFD_FLAG_FORCE_3D : 0);
    m->slab.resize(n_slabs);
    int rc = CFD_OK;
    for (int i = 0; i < n_slabs && rc == CFD_OK; ++i) {
        Slab& s = m->slab[i];
        s.rank = i;
        s.device = dev[i];
        s.x0 = (int)((long)p->nx * i / n_slabs);
        s.x1 = (int)((long)p->nx * (i + 1) / n_slabs);
        const double* f[CFD_NPROP];
        for (int k = 0; k < CFD_NPROP; ++k) f[k] = fields[k] + (size_t)s.x0 * m->plane;
        int on_device = 0;
        for (int k = 0; k < n_slabs; ++k) on_device += dev[k] == dev[i] ? 1 : 0;
        rc = make_slab(m, s, f, active + (size_t)s.x0 * m->plane, extra, on_device);
    }
    if (rc == CFD_OK && transport == CFD_TRANSPORT_RCCL && n_slabs > 1) {
        std::vector<ncclComm_t> comms(n_slabs);
        ncclResult_t r = ncclCommInitAll(comms.data(), n_slabs, dev.data());
        if (r != ncclSuccess) rc = fail(CFD_E_DEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
        else for (int i = 0; i < n_slabs; ++i) m->slab[i].comm = comms[i];
    }
    if (rc == CFD_OK && transport == CFD_TRANSPORT_COPY)
        for (int i = 0; i + 1 < n_slabs && rc == CFD_OK; ++i)
            if (dev[i] != dev[i + 1]) {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, dev[i], dev[i + 1]);
                if (can) {
                    (void)hipSetDevice(dev[i]);     (void)hipDeviceEnablePeerAccess(dev[i + 1], 0);
                    (void)hipSetDevice(dev[i + 1]); (void)hipDeviceEnablePeerAccess(dev[i], 0);
                    (void)hipGetLastError();        // "already enabled" is fine
                }
            }
    // every slab learns its neighbours' masks of the adjoining planes (the one-pass tail needs them), then the
    // slabs agree on the form of a step
    if (rc == CFD_OK && n_slabs > 1) {
        bool pair = true, trio = true;
        // masks two planes deep where every neighbour has two planes to show
        bool deep = true;
        for (int i = 0; i < n_slabs; ++i) deep = deep && (m->slab[i].x1 - m->slab[i].x0 >= 2);
        const int depth = deep ? 2 : 1;
        for (int i = 0; i < n_slabs && rc == CFD_OK; ++i) {
            const Slab& s = m->slab[i];
            const uint8_t* lo = i > 0 ? active + (size_t)(s.x0 - depth) * m->plane : nullptr;
            const uint8_t* hi = i < n_slabs - 1 ? active + (size_t)s.x1 * m->plane : nullptr;
            rc = cfd_set_ghost_active_depth(s.h, depth, lo, hi);
            int ok = 0, ok3 = 0;
            if (rc == CFD_OK) rc = cfd_pair_available(s.h, &ok);
            if (rc == CFD_OK) rc = cfd_one_pass_step_available(s.h, &ok3);
            pair = pair && ok;
            trio = trio && ok3;
        }
        m->pair = pair && env_flag("CFD_MULTI_PAIR", true);
        m->trio = trio && m->pair && env_flag("CFD_MULTI_TRIO", true);
    }
    if (rc != CFD_OK) { cfd_multi_destroy(m); return rc; }
    *out = m;
    return CFD_OK;
}

int cfd_comm_unique_id(void* id128) {
    if (!id128) return fail(CFD_E_ARG, "cfd_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == CFD_COMM_ID_BYTES, "ncclUniqueId is not 128 bytes");
    ncclUniqueId id;
    M_NCCL(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return CFD_OK;
}

int cfd_multi_create_rank(const cfd_params* p, int rank, int world, const void* id128,
                          const double* const fields[CFD_NPROP], const uint8_t* active, cfd_multi** out) {
    if (!p || !fields || !active || !out) return fail(CFD_E_ARG, "cfd_multi_create_rank: null pointer");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(CFD_E_ARG, "cfd_multi_create_rank: rank must be in [0, world)");
    if (world > 1 && !id128) return fail(CFD_E_ARG, "cfd_multi_create_rank: a communicator id is needed for world > 1");
    int dev = p->device;
    if (dev < 0) M_HIP(hipGetDevice(&dev));
    cfd_multi* m = new cfd_multi();
    m->p = *p;                  // nx = this slab's planes; the global extent is not needed here
    m->world = world;
    m->transport = CFD_TRANSPORT_RCCL;
    m->plane = (size_t)p->ny * p->nz;
    m->overlap = env_flag("CFD_MULTI_OVERLAP", true);
    m->loopback = world == 1 && env_flag("CFD_MULTI_LOOPBACK", false);
    m->slab.resize(1);
    Slab& s = m->slab[0];
    s.rank = rank;
    s.device = dev;
    s.x0 = 0;
    s.x1 = p->nx;
    int rc = CFD_OK;
    hipStream_t st = nullptr;
    int* d_flag = nullptr;
    uint8_t *d_mlo = nullptr, *d_mhi = nullptr, *d_first = nullptr, *d_last = nullptr;
    auto cleanup = [&](int code) {
        if (d_flag) (void)hipFree(d_flag);
        if (d_mlo) (void)hipFree(d_mlo);
        if (d_mhi) (void)hipFree(d_mhi);
        if (d_first) (void)hipFree(d_first);
        if (d_last) (void)hipFree(d_last);
        if (code != CFD_OK) cfd_multi_destroy(m);
        return code;
    };
    if (hipSetDevice(dev) != hipSuccess) return cleanup(fail(CFD_E_DEVICE, "cfd_multi_create_rank: hipSetDevice failed"));
    if (m->loopback) {
        ncclUniqueId id;
        ncclResult_t r = ncclGetUniqueId(&id);
        if (r == ncclSuccess) r = ncclCommInitRank(&s.comm, 1, id, 0);
        if (r != ncclSuccess) return cleanup(fail(CFD_E_DEVICE, std::string("ncclCommInitRank (loopback): ") + ncclGetErrorString(r)));
    }
    if (world > 1) {
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof id);
        ncclResult_t r = ncclCommInitRank(&s.comm, world, id, rank);
        if (r != ncclSuccess) return cleanup(fail(CFD_E_DEVICE, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)));
    }
    // agree on the 2-D form: minimum over the ranks of "my W is identically +0"
    // ... and on how deep the neighbours' masks go: two planes where every rank has two planes to show
    int mine[2] = {p->nz == 1 && !(p->flags & CFD_FLAG_FORCE_3D) && w_identically_zero(fields[2], (size_t)p->nx * m->plane),
                   p->nx >= 2};
    int all[2] = {mine[0], mine[1]};
    if ((rc = streams_for(m, dev, &s.ds)) != CFD_OK) return cleanup(rc);
    st = main_of(m, s);
    if (world > 1) {
        if (hipMalloc((void**)&d_flag, 2 * sizeof(int)) != hipSuccess ||
            hipMemcpyAsync(d_flag, mine, 2 * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess)
            return cleanup(fail(CFD_E_DEVICE, "cfd_multi_create_rank: device buffer for the agreement failed"));
        ncclResult_t r = ncclAllReduce(d_flag, d_flag, 2, ncclInt, ncclMin, s.comm, st);
        if (r != ncclSuccess) return cleanup(fail(CFD_E_DEVICE, std::string("ncclAllReduce: ") + ncclGetErrorString(r)));
        if (hipMemcpyAsync(all, d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return cleanup(fail(CFD_E_DEVICE, "cfd_multi_create_rank: reading the agreement back failed"));
    }
    const int extra = all[0] ? 0 : (p->nz == 1 ? CFD_FLAG_FORCE_3D : 0);
    const int depth = all[1] ? 2 : 1;
    if ((rc = make_slab(m, s, fields, active, extra)) != CFD_OK) return cleanup(rc);
    if (m->loopback) {
        // the slab is its own neighbour on both sides: its last planes' masks below, its first planes' above
        if ((rc = cfd_set_ghost_active_depth(s.h, depth, active + (size_t)(p->nx - depth) * m->plane, active)) != CFD_OK)
            return cleanup(rc);
        int ok[2] = {0, 0};
        if ((rc = cfd_pair_available(s.h, &ok[0])) != CFD_OK) return cleanup(rc);
        if ((rc = cfd_one_pass_step_available(s.h, &ok[1])) != CFD_OK) return cleanup(rc);
        m->pair = ok[0] && env_flag("CFD_MULTI_PAIR", true);
        m->trio = ok[1] && m->pair && env_flag("CFD_MULTI_TRIO", true);
    }
    if (world > 1) {
        // neighbours swap the masks of their first / last `depth` planes
        const size_t pb = (size_t)depth * m->plane;
        if (hipMalloc((void**)&d_first, pb) != hipSuccess || hipMalloc((void**)&d_last, pb) != hipSuccess ||
            hipMalloc((void**)&d_mlo, pb) != hipSuccess || hipMalloc((void**)&d_mhi, pb) != hipSuccess ||
            hipMemcpyAsync(d_first, active, pb, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_last, active + (size_t)(p->nx - depth) * m->plane, pb, hipMemcpyHostToDevice, st) != hipSuccess)
            return cleanup(fail(CFD_E_DEVICE, "cfd_multi_create_rank: device buffers for the mask planes failed"));
        ncclResult_t r = ncclGroupStart();
        if (r == ncclSuccess && rank > 0) r = ncclSend(d_first, pb, ncclUint8, rank - 1, s.comm, st);
        if (r == ncclSuccess && rank > 0) r = ncclRecv(d_mlo, pb, ncclUint8, rank - 1, s.comm, st);
        if (r == ncclSuccess && rank < world - 1) r = ncclSend(d_last, pb, ncclUint8, rank + 1, s.comm, st);
        if (r == ncclSuccess && rank < world - 1) r = ncclRecv(d_mhi, pb, ncclUint8, rank + 1, s.comm, st);
        if (r == ncclSuccess) r = ncclGroupEnd();
        if (r != ncclSuccess) return cleanup(fail(CFD_E_DEVICE, std::string("mask exchange: ") + ncclGetErrorString(r)));
        std::vector<uint8_t> lo(pb), hi(pb);
        if ((rank > 0 && hipMemcpyAsync(lo.data(), d_mlo, pb, hipMemcpyDeviceToHost, st) != hipSuccess) ||
            (rank < world - 1 && hipMemcpyAsync(hi.data(), d_mhi, pb, hipMemcpyDeviceToHost, st) != hipSuccess) ||
            hipStreamSynchronize(st) != hipSuccess)
            return cleanup(fail(CFD_E_DEVICE, "cfd_multi_create_rank: reading the mask planes back failed"));
        if ((rc = cfd_set_ghost_active_depth(s.h, depth, rank > 0 ? lo.data() : nullptr,
                                             rank < world - 1 ? hi.data() : nullptr)) != CFD_OK)
            return cleanup(rc);
        int ok[2] = {0, 0};
        if ((rc = cfd_pair_available(s.h, &ok[0])) != CFD_OK) return cleanup(rc);
        if ((rc = cfd_one_pass_step_available(s.h, &ok[1])) != CFD_OK) return cleanup(rc);
        ok[0] = ok[0] && env_flag("CFD_MULTI_PAIR", true);
        ok[1] = ok[1] && ok[0] && env_flag("CFD_MULTI_TRIO", true);
        if (hipMemcpyAsync(d_flag, ok, 2 * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess)
            return cleanup(fail(CFD_E_DEVICE, "cfd_multi_create_rank: agreement on the step form failed"));
        r = ncclAllReduce(d_flag, d_flag, 2, ncclInt, ncclMin, s.comm, st);
        if (r != ncclSuccess) return cleanup(fail(CFD_E_DEVICE, std::string("ncclAllReduce: ") + ncclGetErrorString(r)));
        if (hipMemcpyAsync(ok, d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return cleanup(fail(CFD_E_DEVICE, "cfd_multi_create_rank: agreement on the step form failed"));
        m->pair = ok[0] != 0;
        m->trio = ok[1] != 0;
    }
    cleanup(CFD_OK);
    *out = m;
    return CFD_OK;
}

int cfd_multi_step(cfd_multi* m, int nsteps) {
    if (!m) return fail(CFD_E_ARG, "cfd_multi_step: null handle");
    if (nsteps < 0) return fail(CFD_E_ARG, "cfd_multi_step: nsteps must be >= 0");
    for (int i = 0; i < nsteps; ++i) M_CFD(step_once(m));
    return CFD_OK;
}

int cfd_multi_sync(cfd_multi* m) {
    if (!m) return fail(CFD_E_ARG, "cfd_multi_sync: null handle");
    for (DeviceStreams& d : m->streams) {
        M_HIP(hipSetDevice(d.device));
        M_HIP(hipStreamSynchronize(d.side));
        M_HIP(hipStreamSynchronize(d.main));
    }
    return CFD_OK;
}

int cfd_multi_total_density(cfd_multi* m, double* out) {
    if (!m || !out) return fail(CFD_E_ARG, "cfd_multi_total_density: null pointer");
    if (m->transport == CFD_TRANSPORT_RCCL && m->world > 1) {
        // per-slab tree sums on the device, one all-reduce of a double, one copy to the host
        for (Slab& s : m->slab) M_CFD(cfd_total_density_noalloc(s.h, s.d_sum));
        M_NCCL(ncclGroupStart());
        for (Slab& s : m->slab) M_NCCL(ncclAllReduce(s.d_sum, s.d_sum, 1, ncclDouble, ncclSum, s.comm, main_of(m, s)));
        M_NCCL(ncclGroupEnd());
        Slab& s0 = m->slab[0];
        M_HIP(hipSetDevice(s0.device));
        M_HIP(hipMemcpyAsync(out, s0.d_sum, sizeof(double), hipMemcpyDeviceToHost, main_of(m, s0)));
        return cfd_multi_sync(m);
    }
    double acc = 0.0;
    for (Slab& s : m->slab) {       // slabs in x order: a deterministic sum
        double v = 0.0;
        M_CFD(cfd_total_density(s.h, &v));
        acc += v;
    }
    *out = acc;
    return CFD_OK;
}

int cfd_multi_read_fields(cfd_multi* m, int slot_rel, double* const out[CFD_NPROP]) {
    if (!m || !out) return fail(CFD_E_ARG, "cfd_multi_read_fields: null pointer");
    M_CFD(cfd_multi_sync(m));
    for (Slab& s : m->slab) {
        double* o[CFD_NPROP];
        for (int k = 0; k < CFD_NPROP; ++k) {
            if (!out[k]) return fail(CFD_E_ARG, "cfd_multi_read_fields: null output pointer");
            o[k] = out[k] + (size_t)s.x0 * m->plane;
        }
        M_CFD(cfd_read_fields(s.h, slot_rel, o));
    }
    return CFD_OK;
}

int cfd_multi_write_fields(cfd_multi* m, int slot_rel, const double* const in[CFD_NPROP]) {
    if (!m || !in) return fail(CFD_E_ARG, "cfd_multi_write_fields: null pointer");
    M_CFD(cfd_multi_sync(m));
    for (Slab& s : m->slab) {
        const double* f[CFD_NPROP];
        for (int k = 0; k < CFD_NPROP; ++k) {
            if (!in[k]) return fail(CFD_E_ARG, "cfd_multi_write_fields: null input pointer");
            f[k] = in[k] + (size_t)s.x0 * m->plane;
        }
        M_CFD(cfd_write_fields(s.h, slot_rel, f));
    }
    return CFD_OK;
}

int cfd_multi_set_time(cfd_multi* m, int curr_mod_time, int curr_abs_time) {
    if (!m) return fail(CFD_E_ARG, "cfd_multi_set_time: null handle");
    M_CFD(cfd_multi_sync(m));
    for (Slab& s : m->slab) M_CFD(cfd_set_time(s.h, curr_mod_time, curr_abs_time));
    m->cmt = curr_mod_time;
    m->cat = curr_abs_time;
    return CFD_OK;
}

int cfd_multi_time(cfd_multi* m, int* curr_mod_time, int* curr_abs_time) {
    if (!m || !curr_mod_time || !curr_abs_time) return fail(CFD_E_ARG, "cfd_multi_time: null pointer");
    *curr_mod_time = m->cmt;
    *curr_abs_time = m->cat;
    return CFD_OK;
}

int cfd_multi_info(cfd_multi* m, int* local_slabs, int* one_pass_tail, int* overlapped) {
    if (!m) return fail(CFD_E_ARG, "cfd_multi_info: null handle");
    if (local_slabs) *local_slabs = (int)m->slab.size();
    if (one_pass_tail) *one_pass_tail = m->trio ? 2 : (m->pair ? 1 : 0);
    if (overlapped) *overlapped = (m->overlap && (m->world > 1 || m->loopback)) ? 1 : 0;
    return CFD_OK;
}

int cfd_multi_set_exchange_timing(cfd_multi* m, int enable) {
    if (!m) return fail(CFD_E_ARG, "cfd_multi_set_exchange_timing: null handle");
    m->time_exchange = enable != 0;
    return CFD_OK;
}

int cfd_multi_exchange_timing(cfd_multi* m, double* total_ms, int64_t* exchanges, int reset) {
    if (!m || !total_ms || !exchanges) return fail(CFD_E_ARG, "cfd_multi_exchange_timing: null pointer");
    M_CFD(cfd_multi_sync(m));
    M_HIP(hipSetDevice(m->slab[0].device));
    M_CFD(drain_exchange_events(m));
    *total_ms = m->x_ms;
    *exchanges = m->x_count;
    if (reset) { m->x_ms = 0.0; m->x_count = 0; }
    return CFD_OK;
}

int cfd_multi_slab(cfd_multi* m, int i, cfd_handle** h, int* x0, int* x1) {
    if (!m) return fail(CFD_E_ARG, "cfd_multi_slab: null handle");
    if (i < 0 || i >= (int)m->slab.size()) return fail(CFD_E_ARG, "cfd_multi_slab: no such local slab");
    if (h) *h = m->slab[i].h;
    if (x0) *x0 = m->slab[i].x0;
    if (x1) *x1 = m->slab[i].x1;
    return CFD_OK;
}

// Sends `bytes` of device memory to this rank itself through RCCL (one-rank communicator, grouped send +
// receive) and all-reduces a double: proves on a one-GPU machine that the library's RCCL binding loads, builds
// a communicator and moves data with the call pattern the exchange uses.
int cfd_rccl_selftest(int device, size_t bytes) {
    if (bytes < 8 || bytes % 8) return fail(CFD_E_ARG, "cfd_rccl_selftest: bytes must be a positive multiple of 8");
    M_HIP(hipSetDevice(device));
    const size_t n = bytes / 8;
    std::vector<double> hsrc(n), hdst(n, 0.0);
    for (size_t i = 0; i < n; ++i) hsrc[i] = 1.0 + (double)(i % 1021) * 0.25;
    double *a = nullptr, *b = nullptr, *c = nullptr;
    hipStream_t st = nullptr;
    ncclComm_t comm = nullptr;
    int rc = CFD_OK;
    auto done = [&](int code) {
        if (comm) (void)ncclCommDestroy(comm);
        if (st) (void)hipStreamDestroy(st);
        if (a) (void)hipFree(a);
        if (b) (void)hipFree(b);
        if (c) (void)hipFree(c);
        return code;
    };
    if (hipMalloc((void**)&a, bytes) != hipSuccess || hipMalloc((void**)&b, bytes) != hipSuccess ||
        hipMalloc((void**)&c, 8) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
        return done(fail(CFD_E_MEMORY, "cfd_rccl_selftest: allocation failed"));
    const int devs[1] = {device};
    ncclResult_t r = ncclCommInitAll(&comm, 1, devs);
    if (r != ncclSuccess) return done(fail(CFD_E_DEVICE, std::string("ncclCommInitAll: ") + ncclGetErrorString(r)));
    const double two = 2.0;
    if (hipMemcpyAsync(a, hsrc.data(), bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(b, 0, bytes, st) != hipSuccess || hipMemcpyAsync(c, &two, 8, hipMemcpyHostToDevice, st) != hipSuccess)
        return done(fail(CFD_E_DEVICE, "cfd_rccl_selftest: upload failed"));
    r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclSend(a, n, ncclDouble, 0, comm, st);
    if (r == ncclSuccess) r = ncclRecv(b, n, ncclDouble, 0, comm, st);
    if (r == ncclSuccess) r = ncclGroupEnd();
    if (r == ncclSuccess) r = ncclAllReduce(c, c, 1, ncclDouble, ncclSum, comm, st);
    if (r != ncclSuccess) return done(fail(CFD_E_DEVICE, std::string("RCCL send/recv/all-reduce: ") + ncclGetErrorString(r)));
    double sum = 0.0;
    if (hipMemcpyAsync(hdst.data(), b, bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&sum, c, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return done(fail(CFD_E_DEVICE, "cfd_rccl_selftest: download failed"));
    if (std::memcmp(hsrc.data(), hdst.data(), bytes) != 0 || sum != 2.0)
        rc = fail(CFD_E_STATE, "cfd_rccl_selftest: data came back changed");
    return done(rc);
}

}  // extern "C"
