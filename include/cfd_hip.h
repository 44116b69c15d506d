/*
 * cfd_hip.h -- C ABI of libcfdhip.so: the per-timestep field update of the finite-volume
 * solver (six-equation forward-Euler sweep, three-slot time filter, density reduction) as
 * fp64 HIP kernels for gfx950.
 *
 * The reference program has no FFI layer; these entry points are the seam a host binds in
 * place of its own functions.  Each declaration cites the reference function it replaces
 * (paths are relative to the reference tree).
 *
 * Conventions
 *   - Plain C types only.  The caller owns every pointer it passes; the library copies in
 *     cfd_create and copies out in cfd_read_fields.  The handle owns all device memory.
 *   - Return value 0 = success, negative = error (CFD_E_*); cfd_last_error() returns a
 *     thread-local message for the most recent failure.  Nothing throws or aborts across
 *     the boundary.
 *   - One handle is driven from one host thread at a time (the reference is single-threaded,
 *     CFD.cabal:11 has no -threaded).
 *   - Field order everywhere is the reference's propIndex order
 *     (src/SolutionDomain.hs:77-85): U, V, W, Density, Temperature, Mew, Dye.
 *   - Node order is the reference's: index = (x*ny + y)*nz + z, x slowest
 *     (makePositions, src/SolutionDomain.hs:388-392).
 */
#ifndef CFD_HIP_H
#define CFD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFD_NPROP 7

enum {
    CFD_OK = 0,
    CFD_E_ARG = -1,      /* bad argument (null pointer, non-positive size, bad slot) */
    CFD_E_DEVICE = -2,   /* HIP runtime error, or no usable GPU */
    CFD_E_MEMORY = -3,   /* allocation failed */
    CFD_E_STATE = -4     /* call not valid in the handle's current state */
};

/* Arithmetic.  Default ("strict"): the reference's operation order, no fused contraction, correctly rounded
 * division and square root -- bit-identical to the reference arithmetic (as restated by the CPU oracle) for finite
 * fields.  Divisions and square roots whose operands are Temperature, Density, viscosity or the cell volume use the
 * hardware sequences without their operand scaling, which are the IEEE results for operands in the middle of the
 * exponent range (|x| between about 2^-767 and 2^+970); tests pin bit-equality with the oracle on fields scaled from
 * 1e-300 to 1e+250.  NaN fields, and a -0 numerator of such a division, are outside what is reproduced bit for bit
 * (+0 and -0 are not told apart anywhere).  CFD_FLAG_FAST_MATH: see below. */
/* flags for cfd_params.flags */
enum {
    CFD_FLAG_DEFAULT = 0,
    /* Run the one-thread-per-cell sweep kernel instead of the tiled one.  Same arithmetic;
     * kept so that the tiled kernel can be checked against it at sizes a CPU cannot reach. */
    CFD_FLAG_SIMPLE_KERNEL = 1,
    /* Use the general 3-D kernel even when nz == 1 and W is identically zero. */
    CFD_FLAG_FORCE_3D = 2,
    /* Re-associated arithmetic: folds the geometric constants and replaces divisions by
     * constant intervals with reciprocal multiplies.  Not bit-identical to the strict
     * operation order; agrees with it to <= 1e-13 relative per sweep. */
    CFD_FLAG_FAST_MATH = 4
};

/*
 * Everything the reference hard-codes as top-level constants:
 *   nx,ny,nz   maxPos+1                   src/SolutionDomain.hs:136-141
 *   dx,dy,dz   cellLength                 src/SolutionDomain.hs:150-151
 *   dt         timeStep                   src/SolutionDomain.hs:237-238
 *   lambda     sutherlandLambda           src/SolutionDomain.hs:249-251
 *   suth       sutherlandConstant         src/SolutionDomain.hs:246-247
 *   R          gasConstantR               src/SolutionDomain.hs:253-254
 *   cv         specificHeatCv             src/SolutionDomain.hs:259-260
 *   k          heatConductivityK          src/Main.hs:55-56
 *   dye_f      dye diffusivity factor     src/Main.hs:148
 *
 * Slab decomposition (one handle per GPU): the handle owns nx planes.  ghost_lo / ghost_hi
 * say whether a neighbouring slab exists below x = 0 / above x = nx-1; if so the handle keeps
 * three extra planes on that side, of which the host fills the nearest one before every sweep
 * (cfd_halo_planes), the nearest two before the two-sweeps pass, all three before a one-pass step
 * (cfd_halo_planes_depth), and the clamp-at-the-domain-edge rule of offsetPosition
 * (src/SolutionDomain.hs:434-448) applies only on sides without ghost planes.
 */
typedef struct cfd_params {
    int32_t nx, ny, nz;
    double dx, dy, dz;
    double dt;
    double lambda, suth, R, cv, k, dye_f;
    int32_t device;      /* HIP device ordinal; -1 = the calling thread's current device */
    int32_t flags;       /* CFD_FLAG_* */
    int32_t ghost_lo;    /* 1 if a slab exists on the low-x side */
    int32_t ghost_hi;    /* 1 if a slab exists on the high-x side */
} cfd_params;

typedef struct cfd_handle cfd_handle;

/* Fills *p with the reference's shipped configuration (301 x 151 x 1, dt = 1e-5, air). */
int cfd_default_params(cfd_params *p);

/*
 * Replaces initialGrid_adj's role as the state constructor (src/SolutionDomain.hs:308-353):
 * builds the three time slots as copies of the given fields (":352 V.replicate storedSteps"),
 * currModTime = currAbsTime = 0.  fields[i] has nx*ny*nz doubles (owned planes only, no
 * ghosts); active has nx*ny*nz bytes, non-zero = the node's `active` flag.
 */
int cfd_create(const cfd_params *p, const double *const fields[CFD_NPROP],
               const uint8_t *active, cfd_handle **out);

void cfd_destroy(cfd_handle *h);

/* runSingleStep (src/Main.hs:192-200), nsteps times: three sweeps, the first advancing the
 * time slot, each followed by the time filter.  Only the third filter of a step is live (the
 * first two are overwritten from the same inputs), so it is applied once, fused into the third
 * sweep's kernel; results are identical to cfd_sweep x3 + cfd_time_filter x3.  On 2-D grids with
 * CFD_FLAG_FAST_MATH the whole step is one launch (three sweeps and the filter in one pass over memory).
 * A slab (ghost_lo or ghost_hi set) needs its ghost planes exchanged before every sweep, which one
 * call for a whole step cannot do: CFD_E_STATE; drive a slab sweep by sweep (see the halo section). */
int cfd_step(cfd_handle *h, int nsteps);

/* applyDiffEq_adj calcSteps_adj step_time (src/Main.hs:202-226): one forward-Euler sweep of
 * the six equations from slot currModTime into slot currModTime+1 (step_time != 0) or back
 * into slot currModTime (step_time == 0). */
int cfd_sweep(cfd_handle *h, int step_time);

/* correctPrevTime (src/Main.hs:228-258): slot t-1 <- pp + 0.5*(c - pp) with c = slot t,
 * pp = slot t-2; a no-op while currAbsTime <= 1. */
int cfd_time_filter(cfd_handle *h);

/* applyDiffEq_adj calcSteps_adj False followed by correctPrevTime (the end of a runSingleStep,
 * src/Main.hs:194-197 with next == 2), in one kernel where the sweep kernel supports it. */
int cfd_sweep_and_filter(cfd_handle *h);

/* The tail of runSingleStep (src/Main.hs:194-197 with next == 1 and next == 2): two sweeps that do not
 * advance time, then correctPrevTime.  For 2-D grids with CFD_FLAG_FAST_MATH the two sweeps run in one
 * pass over memory (the intermediate field never reaches HBM); otherwise this is cfd_sweep(h, 0)
 * followed by cfd_sweep_and_filter(h).  Results are identical either way.
 * Slabs (handles with ghost planes): the one-pass form reads TWO ghost planes of the current slot on
 * every side that has a neighbour (cfd_halo_planes_depth with depth 2, exchanged before the call) and the
 * neighbours' masks of the planes next to the slab (cfd_set_ghost_active, once).  cfd_pair_available tells
 * whether it can run; where it cannot, a slab gets CFD_E_STATE here, because two separate sweeps need
 * an exchange in between that only the caller can do. */
int cfd_sweep_pair_and_filter(cfd_handle *h);

/* The same sweep (applyDiffEq_adj, src/Main.hs:202-226) in two halves, for a slab that exchanges
 * ghost planes with its neighbours: cfd_sweep_begin launches every owned plane whose stencil reads
 * owned planes only (all but the first / last plane where a ghost plane exists), so that the
 * exchange of the ghost planes can run at the same time on another stream; cfd_sweep_end launches
 * the remaining planes -- the caller has by then ordered the arrival of the ghost planes before the
 * handle's stream -- and completes the sweep (slot rotation, currAbsTime).  with_filter != 0 (only
 * with step_time == 0): followed by correctPrevTime as in cfd_sweep_and_filter.  Between the two
 * calls every other state-changing entry point returns CFD_E_STATE.  Results are identical to
 * cfd_sweep / cfd_sweep_and_filter. */
int cfd_sweep_begin(cfd_handle *h, int step_time, int with_filter);
int cfd_sweep_end(cfd_handle *h);

/* The split sweep with a deeper edge and the edge on another stream (no counterpart in the reference;
 * for slabs that hide BOTH exchanges of a step behind the first sweep's interior planes).
 * cfd_sweep_begin_depth leaves edge_planes (>= 1) planes on every side that has a neighbour to the
 * second half; cfd_sweep_begin is the case edge_planes == 1.  cfd_sweep_end_on launches that second
 * half on `stream` (a hipStream_t of the handle's device) instead of the handle's own stream and
 * completes the sweep as cfd_sweep_end does; the caller orders `stream` after the arrival of the ghost
 * planes and after whatever wrote the source slot, and orders later work after `stream`.  With
 * edge_planes == 2 the planes cfd_halo_planes_depth(h, 2, ...) names are complete as soon as `stream`
 * has run, while the interior launch may still be running on the handle's stream.  A sweep begun
 * with with_filter != 0 completes on the handle's own stream only (CFD_E_ARG otherwise). */
int cfd_sweep_begin_depth(cfd_handle *h, int step_time, int with_filter, int edge_planes);
int cfd_sweep_end_on(cfd_handle *h, void *stream);

/* totalDensity (src/Main.hs:269-270): sum of Density over slot currModTime-1 (owned planes).
 * The reference folds sequentially; this is a tree reduction, so the last bits may differ. */
int cfd_total_density(cfd_handle *h, double *out);

/* totalDensity for callers that cannot allocate or synchronise (code under HIP graph capture, tight
 * loops, harnesses that forbid hipMalloc): no counterpart in the reference.  The same two kernels as
 * cfd_total_density, queued on the handle's stream; the sum is written to d_out, ONE double in DEVICE
 * memory owned by the caller, complete once the stream has run (nothing is copied to the host, the
 * stream is not synchronised).  The block partials live in a fixed array inside the library's code
 * object (1024 doubles = 8 KiB per GPU, held until the process ends) instead of allocated memory, and the
 * call itself calls no hipMalloc / hipFree or any allocating routine.  Same launch geometry as
 * cfd_total_density, hence the same bits.  Limits: at most CFD_NOALLOC_MAX_CELLS owned cells
 * (CFD_E_ARG beyond); every call on one GPU uses that GPU's one array, so two calls that overlap in time
 * on the same GPU (different handles on different streams) race -- order them. */
#define CFD_NOALLOC_MAX_CELLS 2147483648L
int cfd_total_density_noalloc(cfd_handle *h, double *d_out);

/* Copies one time slot to the host.  slot_rel: 0 = currModTime, 1 = one back, 2 = two back
 * (pushBackTime, src/SolutionDomain.hs:422-423).  out[i] receives nx*ny*nz doubles. */
int cfd_read_fields(cfd_handle *h, int slot_rel, double *const out[CFD_NPROP]);

/* The inverse of cfd_read_fields: overwrites one time slot (owned planes) with host fields, in[i] holding
 * nx*ny*nz doubles in propIndex order.  With cfd_set_time this restores a saved AdjGraph -- its three
 * grids, currModTime and currAbsTime (src/SolutionDomain.hs:53-57); the reference keeps that state only in
 * the accumulator of its solve loop (src/Main.hs:278-306), so a run there cannot be resumed.  To restore:
 * cfd_create with any fields of the right shape, cfd_set_time, then cfd_write_fields for slot_rel 0, 1, 2.
 * Stored Mew is the same in every slot (no equation updates it): the copy given with slot_rel 0 is kept.
 * A handle in its 2-D form (cfd_live_fields == 5) holds no W and rejects a W that is not identically +0
 * with CFD_E_STATE.  Ghost planes are not touched (the host refreshes them before every sweep). */
int cfd_write_fields(cfd_handle *h, int slot_rel, const double *const in[CFD_NPROP]);

/* Sets currModTime (0..2, the slot the next sweep reads) and currAbsTime (the number of time-advancing
 * sweeps so far; the time filter is a no-op while it is <= 1) of the AdjGraph, src/SolutionDomain.hs:55-56.
 * Slots keep their contents: slot_rel 0 afterwards names slot curr_mod_time. */
int cfd_set_time(cfd_handle *h, int curr_mod_time, int curr_abs_time);

/* Output properties, numbered as the reference's Property enum (src/SolutionDomain.hs:30). */
enum {
    CFD_PROPERTY_SPEED = 0, CFD_PROPERTY_VORTICITY = 1, CFD_PROPERTY_PRESSURE = 2, CFD_PROPERTY_U = 3,
    CFD_PROPERTY_V = 4, CFD_PROPERTY_W = 5, CFD_PROPERTY_DENSITY = 6, CFD_PROPERTY_TEMPERATURE = 7,
    CFD_PROPERTY_MEW = 8, CFD_PROPERTY_DYE = 9
};

/* adjGraphToGrid vs prp rowLength zLevel (src/Main.hs:339-346): `prop_adj Nondirectional prp c Prev`
 * (src/SolutionDomain.hs:540-575) of every cell with z == z_level, computed on the device and
 * copied to out[x*ny + y] (nx rows of ny values; the reference calls it with rowLength = ny and
 * zLevel = (nz-1)/2, src/Main.hs:288).  Speed = |(U,V,W)|; Vorticity = magnitude of the three
 * curl components taken at the East face; Pressure = (R T) rho; Mew from Sutherland's law (the
 * stored Mew is not used); the other seven are the stored values.  The reference evaluates it on
 * slot currModTime (side Prev resolves to the position's own slot): slot_rel = 0; 1 and 2 give the
 * filtered history slots.  In a slab run Vorticity reads the ghost planes of that slot: exchange
 * them first. */
int cfd_property_slice(cfd_handle *h, int property, int z_level, int slot_rel, double *out);

/* currModTime and currAbsTime of the AdjGraph (src/SolutionDomain.hs:53-57). */
int cfd_time(cfd_handle *h, int *curr_mod_time, int *curr_abs_time);

const char *cfd_last_error(void);

/* ---- slab halo exchange and overlap (no counterpart in the single-process reference) ---- */

/*
 * Device addresses and size of the planes a host exchanges before a sweep.  send_lo / send_hi
 * are the first / last owned x-plane of the current slot; recv_lo / recv_hi are its ghost
 * planes (NULL where the handle has no ghost on that side).  A plane holds every live field
 * contiguously: plane_bytes = nfields * ny * nz * 8.
 */
int cfd_halo_planes(cfd_handle *h, void **send_lo, void **send_hi, void **recv_lo,
                    void **recv_hi, size_t *plane_bytes);

/* The same for a halo `depth` planes deep (1, 2 or 3): send_lo = the first `depth` owned planes, send_hi =
 * the last `depth`, recv_lo / recv_hi = the `depth` ghost planes next to the slab; bytes = depth *
 * plane_bytes, each range contiguous.  Depth 2 serves cfd_sweep_pair_and_filter, depth 3 the one-pass step
 * (cfd_step_one_pass*).  No counterpart in the reference (single address space). */
int cfd_halo_planes_depth(cfd_handle *h, int depth, void **send_lo, void **send_hi, void **recv_lo,
                          void **recv_hi, size_t *bytes);

/* `active` flags (src/SolutionDomain.hs:51) of the neighbouring slabs' planes next to this slab: lo_plane
 * = the lower neighbour's last plane, hi_plane = the upper neighbour's first plane, ny*nz bytes each, host
 * memory; NULL on a side without a ghost plane.  Needed once, before the first
 * cfd_sweep_pair_and_filter of a slab: the once-swept values of a ghost plane depend on which of its
 * cells keep their velocity and dye (calcSteps_adj, src/Main.hs:180-187). */
int cfd_set_ghost_active(cfd_handle *h, const uint8_t *lo_plane, const uint8_t *hi_plane);

/* *available = 1 if cfd_sweep_pair_and_filter runs as one pass for this handle as it is now. */
int cfd_pair_available(cfd_handle *h, int *available);

/* The same masks two planes deep (depth = 1 or 2): lo_planes = the lower neighbour's last `depth` planes,
 * hi_planes = the upper neighbour's first `depth` planes, each in ascending x, depth*ny*nz bytes.  Depth 2 serves
 * the one-pass step of a slab (cfd_step_one_pass*): the once-swept values of the second ghost plane and the
 * twice-swept values of the first depend on those cells' `active` flags. */
int cfd_set_ghost_active_depth(cfd_handle *h, int depth, const uint8_t *lo_planes, const uint8_t *hi_planes);

/* runSingleStep (src/Main.hs:192-200) as ONE pass over memory: the three sweeps of the step and the live
 * correctPrevTime in one launch, the two intermediate fields held on chip (2-D grids with CFD_FLAG_FAST_MATH;
 * cfd_one_pass_step_available tells whether it can run for the handle as it is now).  cfd_step takes this form
 * by itself on an undecomposed grid; these entry points exist for slabs, which cfd_step refuses:
 *   - the step reads THREE ghost planes of the current slot on every side that has a neighbour
 *     (cfd_halo_planes_depth with depth 3: ONE exchange per step) and needs the neighbours' masks two planes
 *     deep (cfd_set_ghost_active_depth, once);
 *   - cfd_step_one_pass runs the whole step on the handle's stream; cfd_step_one_pass_begin launches the blocks
 *     whose stencils stay inside the owned planes, so that the exchange can run beside them on another stream,
 *     and cfd_step_one_pass_end_on launches the rest on `stream` (ordered by the caller after the arrival of the
 *     ghost planes; later work is ordered after `stream` AND the handle's stream) and completes the step.
 * Results are identical, bit for bit, to cfd_sweep x3 + cfd_time_filter with an exchange before every sweep. */
int cfd_one_pass_step_available(cfd_handle *h, int *available);
int cfd_step_one_pass(cfd_handle *h);
int cfd_step_one_pass_begin(cfd_handle *h);
int cfd_step_one_pass_end_on(cfd_handle *h, void *stream);

/* The HIP stream (hipStream_t) all of the handle's kernels run on. */
int cfd_stream(cfd_handle *h, void **stream);

/* Makes the handle launch on a caller-owned stream (hipStream_t; NULL = the device's default
 * stream) from now on, e.g. the stream a host framework orders its collectives against. */
int cfd_set_stream(cfd_handle *h, void *stream);

/* Name of the sweep kernel the handle launches (for benchmark records). */
const char *cfd_kernel_name(cfd_handle *h);

/* Blocks until everything queued on the handle's stream has finished. */
int cfd_sync(cfd_handle *h);

/* Turns per-launch HIP-event timing of the sweep kernel on (enable != 0) or off (default). */
int cfd_set_timing(cfd_handle *h, int enable);

/*
 * Measurement hook: device time of the sweep kernel, from HIP events recorded on the
 * handle's stream around every sweep launch while timing was enabled.  total_ms = sum of
 * kernel durations, launches = number of sweeps.  reset != 0 clears the accumulators
 * after reading.
 */
int cfd_sweep_timing(cfd_handle *h, double *total_ms, int64_t *launches, int reset);

/* As cfd_sweep_timing, per kernel kind: 0 = plain sweep launches, 1 = sweeps fused with the time
 * filter (the last sweep of a step once currAbsTime > 1), 2 = launches that do the second and third
 * sweep of a step and the filter in one pass (cfd_sweep_pair_and_filter where it applies), 3 = launches
 * that do a whole step -- three sweeps and the filter -- in one pass (cfd_step, cfd_step_one_pass*). */
int cfd_kernel_timing(cfd_handle *h, int kind, double *total_ms, int64_t *launches, int reset);

/* Number of live fields the device holds per slot: 5 when nz == 1 and W is identically
 * zero (W then stays exactly zero, so it is neither read nor written), else 6. */
int cfd_live_fields(cfd_handle *h, int *nfields);

/* ---- several slabs behind one handle (no counterpart in the single-process, single-device reference) ----
 *
 * The x-slab decomposition driven by the library itself: the host keeps the reference's one-thread solve
 * loop (src/Main.hs:278-306) and calls cfd_multi_step / cfd_multi_total_density in place of runSingleStep /
 * totalDensity; the library cuts the grid into slabs of consecutive x-planes (x is the slowest index of the
 * node order, src/SolutionDomain.hs:388-392, so a slab is a contiguous range of every field), exchanges the
 * ghost planes before every sweep and overlaps the exchanges with the interior planes.  Results are
 * identical to the undecomposed run, bit for bit, except the density sum (its order of summation).
 */
typedef struct cfd_multi cfd_multi;

enum {
    CFD_TRANSPORT_COPY = 0,     /* device-to-device / peer copies; any number of slabs per device, one process */
    CFD_TRANSPORT_RCCL = 1      /* grouped ncclSend / ncclRecv between neighbours, ncclAllReduce for the density sum */
};

/* One host thread, every slab in this process.  p describes the WHOLE grid (p->device and ghost_* are
 * ignored); fields / active as for cfd_create.  devices[i] = HIP device of slab i (NULL: slab i on device
 * i modulo the device count).  CFD_TRANSPORT_RCCL: one slab per device (ncclCommInitAll over them).
 * CFD_TRANSPORT_COPY: slabs may share a device (on one GPU this runs the whole decomposition, halo planes
 * copied in place of sent).  Whether W is kept (cfd_live_fields) is decided on the whole grid, so that every
 * slab's planes have the same size. */
int cfd_multi_create(const cfd_params *p, int n_slabs, const int32_t *devices, int transport,
                     const double *const fields[CFD_NPROP], const uint8_t *active, cfd_multi **out);

/* One process per GPU: this process holds slab `rank` of `world` (p, fields, active describe that slab only:
 * p->nx = its planes, p->device = its GPU, -1 = the current one).  The slabs talk through RCCL with a
 * communicator built from id128, the 128 bytes cfd_comm_unique_id wrote on ONE rank and the host sent to all
 * (any host-side channel: MPI, a file, torch.distributed).  Collective: every rank calls it, and afterwards
 * cfd_multi_step / cfd_multi_total_density, the same number of times.
 * Rehearsal on one GPU: with world == 1 and CFD_MULTI_LOOPBACK=1 in the environment the slab is its own lower and
 * upper neighbour (a periodic domain) and exchanges its ghost planes with itself through RCCL every step, as a rank
 * between two others does -- for timing and for testing the exchange, not for comparison with the reference. */
#define CFD_COMM_ID_BYTES 128
int cfd_comm_unique_id(void *id128);
int cfd_multi_create_rank(const cfd_params *p, int rank, int world, const void *id128,
                          const double *const fields[CFD_NPROP], const uint8_t *active, cfd_multi **out);

void cfd_multi_destroy(cfd_multi *m);

/* runSingleStep (src/Main.hs:192-200) nsteps times on every slab: per step three sweeps with a one-plane
 * exchange before each; or -- where every slab can run cfd_sweep_pair_and_filter -- one sweep, a two-plane
 * exchange and the one-pass tail, both exchanges queued on a side stream beside the first sweep's interior
 * planes; or -- where every slab can run cfd_step_one_pass -- ONE three-plane exchange on the side stream beside
 * the blocks of the one-pass step that read owned planes only, then the blocks next to the neighbours.
 * Asynchronous: returns when the work is queued. */
int cfd_multi_step(cfd_multi *m, int nsteps);

/* totalDensity (src/Main.hs:269-270) over all slabs.  RCCL: per-slab sums on the device and one ncclAllReduce
 * of a double (every rank gets the sum); copies: the per-slab sums added on the host in x order. */
int cfd_multi_total_density(cfd_multi *m, double *out);

/* cfd_read_fields of every local slab into the caller's arrays: out[i] holds the planes of all local slabs
 * (the whole grid after cfd_multi_create, the rank's slab after cfd_multi_create_rank). */
int cfd_multi_read_fields(cfd_multi *m, int slot_rel, double *const out[CFD_NPROP]);

int cfd_multi_time(cfd_multi *m, int *curr_mod_time, int *curr_abs_time);

/* State restore of a decomposed run (the reference keeps its state only in the solve loop's accumulator,
 * src/Main.hs:278-306): cfd_write_fields / cfd_set_time on every local slab, in[i] holding the planes of all local
 * slabs as cfd_multi_read_fields returns them; the handle's own time counters follow.  The slabs' ghost planes need
 * no attention: every step begins with an exchange of the current slot.  To restore: create the handle from any
 * fields of the right shape, cfd_multi_set_time, then cfd_multi_write_fields for slot_rel 0, 1, 2.  (Do not call
 * cfd_write_fields / cfd_set_time on the handles cfd_multi_slab returns: the cfd_multi's counters would not follow.) */
int cfd_multi_write_fields(cfd_multi *m, int slot_rel, const double *const in[CFD_NPROP]);
int cfd_multi_set_time(cfd_multi *m, int curr_mod_time, int curr_abs_time);

/* Blocks until everything queued by the handle has finished on every device. */
int cfd_multi_sync(cfd_multi *m);

/* How the handle runs: slabs held by this process; *one_pass_tail = 2 if every slab runs the whole step as one
 * pass (one three-plane exchange per step), 1 if the step's tail (second and third sweep) is one pass (two
 * exchanges per step), 0 for three sweeps with an exchange before each; whether the exchanges run on the side
 * stream beside the interior planes. */
int cfd_multi_info(cfd_multi *m, int *local_slabs, int *one_pass_tail, int *overlapped);

/* Measurement hook: duration of the ghost-plane exchanges of the first local slab, from HIP events recorded on the
 * exchange's stream around every exchange while the timing was enabled -- from the moment that stream gets to the
 * exchange (the slab's own planes are complete) to the arrival of its ghost planes, the wait for the neighbours
 * included.  cfd_multi_exchange_timing synchronises the handle first.  No counterpart in the reference. */
int cfd_multi_set_exchange_timing(cfd_multi *m, int enable);
int cfd_multi_exchange_timing(cfd_multi *m, double *total_ms, int64_t *exchanges, int reset);

/* Local slab i: its single-slab handle (for cfd_set_timing / cfd_kernel_timing / cfd_property_slice; do not
 * step or destroy it) and its plane range [x0, x1) within the arrays given at creation. */
int cfd_multi_slab(cfd_multi *m, int i, cfd_handle **h, int *x0, int *x1);

/* RCCL on one GPU: a one-rank communicator sends `bytes` of device memory to itself (grouped send + receive,
 * the exchange's call pattern) and all-reduces a double; CFD_OK if the data came back unchanged. */
int cfd_rccl_selftest(int device, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* CFD_HIP_H */
