#!/usr/bin/env python3
"""Benchmark of the per-timestep field update on MI355X.

    python bench.py --gpus N --steps K --warmup W

One bench "step" is one reference step (runSingleStep): three forward-Euler sweeps of the six
equations plus the live time-filter pass, over the whole grid.  Workload at N = 1: the shipped
case scaled to a 2-D 4096 x 4096 synthetic grid (cases.synthetic_2d, seed 1).  For N > 1 the
grid is slab-decomposed along x, one process per GPU, 4096 planes per GPU (weak scaling), with
the halo planes exchanged before every sweep; --strong keeps the global grid fixed instead (--nx is
then the global number of planes, e.g. --strong --nx 512 --ny 512 --nz 512).

Time step: the reference ships dt = 1e-5, with which its explicit scheme amplifies every
perturbation until the fields overflow (NaN within ~20 steps on this grid); the benchmark uses
the alternative the reference's source records next to it, dt = 1e-7, with which a 1000-step
run stays finite.  Kernel time does not depend on dt.

Arithmetic: --arith fast (default) runs the re-associated form (<= 1e-13 of the field scale per
sweep against the strict form, see tests); --arith strict runs the reference's operation order
(bit-identical to the CPU oracle).  The strict kernel is also timed on a few steps and reported
under "strict" in the same JSON line.

Launch: with --gpus N > 1 and no WORLD_SIZE in the environment the command starts the N ranks itself (a
child `python -m torch.distributed.run --nproc-per-node N bench.py ...`, started before this process has made
any GPU call), relays rank 0's JSON line and exits with the children's status; under a launcher (WORLD_SIZE
set) it is one rank.  Warm-up: --warmup steps, then more steps until at least --warmup-seconds (0.5 s) of
steps have run, so that a short run is timed at the clock the GPU sustains, not at its boost clock; the extra
steps are reported as "warmup_extra_steps".  --dry-run rehearses the launch path without a GPU (gloo
rendezvous, barrier, max-over-ranks reduction, no compute): its JSON line carries "dry_run": true and no value.

Prints ONE JSON line (rank 0): whole-job cell-updates/s with the inputs already resident in
HBM, the sweep kernel's roofline object (HIP-event kernel time measured in the timed region)
and, at N = 1, the CPU baseline (the C restatement under oracle/, timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak (spec)
BYTES_PER_UPDATE_2D = 81        # 5 fields x 8 B x (read + write) + 1 B mask
BYTES_PER_UPDATE_3D = 97        # 6 fields x 8 B x (read + write) + 1 B mask
SWEEPS_PER_STEP = 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=4096, help="x-planes per GPU")
    ap.add_argument("--ny", type=int, default=4096)
    ap.add_argument("--nz", type=int, default=1)
    ap.add_argument("--case", choices=("synthetic", "cavity"), default="synthetic",
                    help="cavity: lid-driven-cavity style walls, the global grid is (nx*gpus) x ny (2-D only)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --nx is the global number of x-planes, cut into --gpus slabs")
    ap.add_argument("--arith", choices=("fast", "strict"), default="fast")
    ap.add_argument("--dt", type=float, default=1e-7)
    ap.add_argument("--flags", type=int, default=0, help="extra cfd_params.flags (1 = one-thread-per-cell kernel)")
    ap.add_argument("--warmup-seconds", type=float, default=0.5,
                    help="keep warming up until this much time has been spent in steps")
    ap.add_argument("--slab-driver", choices=("c", "torch"), default="c",
                    help="who drives the slabs at N > 1: the library itself (cfd_multi_*: RCCL called from C, the "
                         "process group is gloo and carries only the communicator id, barriers and the timing "
                         "reduction) or cfd_amd.slab on torch.distributed's nccl backend")
    ap.add_argument("--virtual-slabs", type=int, default=1,
                    help="N = 1 only: cut this GPU's grid into K slabs driven by the library (copy transport), to "
                         "measure what the decomposition itself costs")
    ap.add_argument("--loopback", action="store_true",
                    help="N = 1 only, a rehearsal: this GPU's slab as a rank between two others -- it is its own lower and "
                         "upper neighbour and exchanges its ghost planes with itself through the library's RCCL path "
                         "every step (a periodic domain: the step is timed, the fields are not the reference's)")
    ap.add_argument("--no-preflight", action="store_true",
                    help="N > 1: skip the bit-for-bit check of the decomposition against an undecomposed run before timing")
    ap.add_argument("--preflight-timeout", type=float, default=420.0,
                    help="N > 1: seconds the rendezvous, the pre-flight and the creation of the timed slab may take "
                         "together before the rank exits with status 3 (three communicators are built in that time, "
                         "on a fresh machine each library loads its device code first)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse launch + rendezvous + reduction without a GPU (gloo, no compute)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="edge of a square CPU-baseline sample grid of the workload's recipe (0: the workload's own grid "
                         "when it is 2-D and at most 4096 x 4096, else 2048)")
    return ap.parse_args()


def host_cores():
    """(physical cores of the machine, CPUs this process may use).  Physical cores: distinct (physical id, core id)
    pairs of /proc/cpuinfo.  Usable CPUs: the smaller of the scheduler affinity mask and the thread cap the machine's
    operator exports (OMP_NUM_THREADS: a GPU box of the pool hands each job the CPU share of one GPU)."""
    phys = set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    pid = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    cid = line.split(":")[1].strip()
                elif not line.strip():
                    if cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
            if cid is not None:
                phys.add((pid, cid))
    except OSError:
        pass
    physical = len(phys) or (os.cpu_count() or 1)
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    cap = os.environ.get("OMP_NUM_THREADS", "")
    if cap.isdigit() and int(cap) >= 1:
        usable = min(usable, int(cap))
    return physical, usable


def cpu_baseline(edge, fields_of_workload=None):
    """Times the C restatement (oracle/, kind = "port") on the workload's own grid at a reduced step count (or an
    edge x edge sample of the same recipe), and on the reference's own shipped configuration (301 x 151 x 1).
    Threads: one per physical core, within the CPUs this job may use (both numbers are reported)."""
    from cfd_amd import cases
    from oracle import OracleB
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_params
    physical, usable = host_cores()
    cores = max(1, min(physical, usable))

    def timed(params, fields, active, budget_s, max_steps):
        """As many steps as fit the time budget, judged from the (untimed) first step."""
        b = OracleB(oracle_params(params), fields, active, pow_mode=1, nthreads=cores)
        t0 = time.perf_counter()
        b.step(1)                               # untimed: first touch of the table
        first = time.perf_counter() - t0
        nsteps = max(1, min(max_steps, int(budget_s / max(first, 1e-9))))
        t0 = time.perf_counter()
        b.step(nsteps)
        dt = time.perf_counter() - t0
        b.close()
        return params["nx"] * params["ny"] * params["nz"] * SWEEPS_PER_STEP * nsteps / dt, dt, nsteps

    case = fields_of_workload if fields_of_workload is not None else cases.synthetic_2d(edge, edge, seed=1)
    rate, dt, nsteps = timed(*case, 14.0, 50)
    rate1, dt1, nsteps1 = timed(*cases.reference_case(), 3.0, 1000)
    return {"value": rate, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "physical_cores_of_the_host": physical, "cpus_this_job_may_use": usable,
            "sample": "synthetic_2d %dx%d (the workload's recipe and seed), %d steps = %d sweeps + filters, "
                      "C restatement with OpenMP, %d threads, %.1f s" % (case[0]["nx"], case[0]["ny"], nsteps,
                                                                         SWEEPS_PER_STEP * nsteps, cores, dt),
            "shipped_case_301x151": {"value": rate1, "unit": "cell-updates/s", "cores": cores,
                                     "sample": "the reference's configuration, %d steps, %.1f s" % (nsteps1, dt1)}}


def one_step_fast_vs_strict(params, fields, active, flags):
    """The re-associated arithmetic against the strict (reference operation order, bit-identical to the oracle) one
    on the workload itself: one step of each from the same initial state, largest difference per property relative
    to the field scale (largest velocity component for U, V, W)."""
    import numpy as np
    from cfd_amd.solver import AdjGraph
    out = []
    for fl in (flags | 4, flags & ~4):
        g = AdjGraph(dict(params, flags=fl), fields, active)
        g.step(1)
        out.append(g.read_fields(0))
        g.close()
    fast, strict = out
    vs = max(float(np.max(np.abs(strict[k]))) for k in (0, 1, 2))
    worst = 0.0
    for k in (0, 1, 2, 3, 4, 6):
        sc = vs if k < 3 else float(np.max(np.abs(strict[k])))
        d = float(np.max(np.abs(fast[k] - strict[k])))
        if d:
            worst = max(worst, d / sc if sc > 0 else float("inf"))
    return worst


def kernel_source_hash():
    """SHA-256 of the kernel sources and their build options: counter records are valid for the build they were taken
    on only."""
    import hashlib
    h = hashlib.sha256()
    for name in ("cfd_hip.hip", "sweep_math.h", "cfd_strict2d_lean.hip", "cfd_multi.hip", "Makefile"):
        with open(os.path.join(ROOT, "cfd_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def counter_record(workload, kernel):
    """The entry of profiles/traffic.json (separate rocprofv3 --pmc passes, scripts/record_traffic.py) for `kernel`,
    if one was recorded FOR THIS BUILD of the kernels: a record stamped with another source hash is not reported."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get("kernel_source_sha256") != kernel_source_hash():
            return None
        for entry in rec.get(workload, []):
            if kernel.startswith(entry.get("kernel_prefix", "\0")):
                return entry
        return None
    except (OSError, ValueError):
        return None


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the counter record, or None."""
    entry = counter_record(workload, kernel)
    return entry.get("hbm_bytes_per_launch") if entry else None


def valu_issue(workload, kernel, launch_ms, sclk_mhz, n_simd=1024):
    """How busy the launch keeps the fp64 vector pipes: issue slots the counters saw (4 cycles each, all SIMDs) over
    the slots the launch's duration offers at the clock read while the steps ran.  None without a counter record."""
    entry = counter_record(workload, kernel)
    if not entry or not entry.get("valu_issue_slots_per_launch") or not sclk_mhz or not launch_ms:
        return None
    offered = n_simd * (launch_ms * 1e-3) * (sclk_mhz * 1e6) / 4.0
    return {"wave_instructions_per_launch": entry.get("valu_wave_instructions_per_launch"),
            "issue_slots_per_launch": entry["valu_issue_slots_per_launch"], "clock_mhz": sclk_mhz,
            "frac_of_issue_slots": entry["valu_issue_slots_per_launch"] / offered}


def spawn_ranks(a):
    """--gpus N > 1 outside a launcher: start the N ranks as fresh processes (this process has made no GPU
    call and imports no torch), relay rank 0's JSON line, exit with the launcher's status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    json_lines = []
    for line in proc.stdout:
        if line.lstrip().startswith("{"):
            json_lines.append(line.rstrip("\n"))
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if json_lines:
        print(json_lines[-1], flush=True)
    raise SystemExit(rc if rc else (0 if json_lines else 1))


def dry_run(a, world, rank):
    """The launch path without a GPU: rendezvous over gloo, barrier, the max-over-ranks reduction, one line."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "fp64 cell-updates/sec", "value": None, "unit": "cell-updates/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "dry_run": True,
                          "note": "launch-path rehearsal: no GPU, no compute, nothing measured"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def start_watchdog(seconds, what, rank):
    """A timer thread that ends THIS process (exit status 3, no re-exec) if `what` has not finished in time: a rank
    stuck in a rendezvous or in a collective whose peer never arrives must fail the job, not hang it."""
    import threading

    def fire():
        sys.stderr.write("rank %d: %s did not finish within %.0f s; exiting with status 3\n" % (rank, what, seconds))
        sys.stderr.flush()
        os._exit(3)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def sclk_reader(device_index):
    """Returns a function that reads the GPU's current shader clock in MHz from sysfs (None where unreadable)."""
    import glob
    cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if not cands:
        return lambda: None
    path = cands[min(device_index, len(cands) - 1)]

    def read():
        try:
            with open(path) as f:
                for line in f:
                    if line.rstrip().endswith("*"):
                        return int(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        except (OSError, ValueError, IndexError):
            pass
        return None
    return read


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)          # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CFD_BENCH_ONE_DEVICE"):
        # rehearsal of the failure path on a one-GPU machine: every rank on device 0, where no communicator can be
        # built (RCCL refuses two ranks on one device); the job must end with a non-zero status, not hang
        local_rank = 0
    if world != a.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if a.dry_run:
        return dry_run(a, world, rank)
    # Rank 0 prints ONE line on stdout.  Libraries write there too (RCCL prints a version banner when its first
    # communicator is built): from here on file descriptor 1 is stderr, and the line goes to the saved stdout.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU path")
    torch.cuda.set_device(local_rank)
    from cfd_amd import cases
    from cfd_amd.slab import SlabGraph, init_rccl
    if a.loopback and (world > 1 or a.virtual_slabs > 1):
        raise SystemExit("--loopback is a one-rank rehearsal")
    c_driver = a.slab_driver == "c" and (world > 1 or a.virtual_slabs > 1 or a.loopback)
    watchdog = start_watchdog(a.preflight_timeout, "rendezvous + pre-flight", rank) if world > 1 else None
    if world > 1 and c_driver:
        import datetime
        # control plane only; the halo planes go through the library's RCCL
        # (its timeout also bounds how far apart the ranks may reach a control-plane collective)
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))
    elif world > 1:
        init_rccl(torch.device("cuda", local_rank))     # rank, world size and rendezvous from the environment
    red_dev = "cpu" if (world > 1 and c_driver) else "cuda"

    global_nx = a.nx if a.strong else a.nx * world
    if a.strong:
        # this rank's share of the global planes (the first global_nx % world ranks hold one more)
        if global_nx < world:
            raise SystemExit("--strong: fewer x-planes than ranks")
        a.nx = global_nx // world + (1 if rank < global_nx % world else 0)
    two_d = a.nz == 1
    run_flags = a.flags | (4 if a.arith == "fast" else 0)

    def all_agree(ok):
        """True if `ok` holds on every rank (a collective on the current default group)."""
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def library_slab(params, fields, active):
        """This rank's slab behind cfd_multi_create_rank (a collective: rank 0 makes the communicator id)."""
        from cfd_amd.multi import MultiGraph, comm_unique_id
        box = [comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return MultiGraph.for_rank(params, fields, active, rank, world, box[0])

    def preflight_of(make):
        """Runs the pre-flight with slabs made by `make`; every rank gets the same verdict."""
        from cfd_amd import preflight
        rep, err = None, None
        try:
            rep = preflight.run(rank, world, run_flags, a.nz, lambda name, p, f, m, x0, x1: make(p, f, m), device=local_rank)
        except Exception as e:                  # e.g. the communicator could not be created on this rank
            err = "%s: %s" % (type(e).__name__, e)
        ok = all_agree(rep is not None and rep["ok"])
        if rep is None:
            rep = {"ok": False, "cases": [], "steps": 0, "detail": "rank %d: %s" % (rank, err)}
        if not ok and rep["ok"]:
            rep = dict(rep, ok=False, detail="another rank reported a difference")
        if rep["detail"]:
            sys.stderr.write("pre-flight, rank %d: %s\n" % (rank, rep["detail"]))
        return rep

    fallback, preflight_report = None, None
    if world > 1 and not a.no_preflight:
        if c_driver:
            preflight_report = dict(preflight_of(library_slab), driver="library (cfd_multi_*, RCCL)")
            if not preflight_report["ok"]:
                # one failure sends every rank to the torch.distributed driver, on a rendezvous port of its own
                fallback = "pre-flight of the library's slab driver failed: %s" % preflight_report["detail"]
                dist.destroy_process_group()
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
                init_rccl(torch.device("cuda", local_rank))
                c_driver, red_dev = False, "cuda"
        if not c_driver:
            first = preflight_report
            preflight_report = dict(preflight_of(lambda p, f, m: SlabGraph(p, f, m, rank=rank, world=world)),
                                    driver="cfd_amd.slab over torch.distributed (nccl backend)")
            if first is not None:
                preflight_report["library_driver"] = first
            if not preflight_report["ok"]:
                sys.stderr.write("rank %d: the decomposed run does not reproduce the undecomposed one; not timing it\n" % rank)
                sys.stderr.flush()
                os._exit(4)

    if two_d and a.case == "cavity":
        first = sum(global_nx // world + (1 if r < global_nx % world else 0) for r in range(rank)) if a.strong \
            else a.nx * rank
        params, fields, active = cases.cavity_2d_slab(global_nx, a.ny, first, first + a.nx)
    elif two_d:
        params, fields, active = cases.synthetic_2d(a.nx, a.ny, seed=1 + rank)
    else:
        params, fields, active = cases.synthetic_3d(a.nx, a.ny, a.nz, seed=3 + rank)
    params["device"] = local_rank
    params["dt"] = a.dt
    params["flags"] = run_flags
    if c_driver:
        from cfd_amd.multi import MultiGraph

        class LibrarySlabs:
            """The slabs of this process behind cfd_multi_*, with the few verbs the loop below uses."""

            def __init__(self):
                if world > 1:
                    self.m = library_slab(params, fields, active)
                elif a.loopback:
                    os.environ["CFD_MULTI_LOOPBACK"] = "1"
                    # what a pre-flight leaves behind in a rank's process: two rounds of an undecomposed graph and a
                    # slab (communicator, streams) created, stepped and destroyed before the timed slab exists
                    from cfd_amd.solver import AdjGraph
                    for seed in (11, 12):
                        sp, sf, sa = cases.synthetic_2d(48, 528, seed=seed) if two_d else cases.synthetic_3d(12, 20, 18, seed=seed)
                        sp = dict(sp, dt=a.dt, flags=run_flags, device=local_rank)
                        ref, sl = AdjGraph(sp, sf, sa), MultiGraph.for_rank(sp, sf, sa, 0, 1, None)
                        ref.step(3)
                        sl.step(3)
                        sl.total_density()
                        sl.close()
                        ref.close()
                    self.m = MultiGraph.for_rank(params, fields, active, 0, 1, None)
                    del os.environ["CFD_MULTI_LOOPBACK"]
                else:
                    self.m = MultiGraph(params, fields, active, a.virtual_slabs, devices=[local_rank] * a.virtual_slabs)
                self.graph = self.m.slab(0)

            def step(self, n=1):
                self.m.step(n)

            def sync(self):
                self.m.sync()

            def total_density(self):
                return self.m.total_density()

            def close(self):
                self.m.close()

        g, err = None, None
        try:
            g = LibrarySlabs()
        except Exception as e:          # e.g. the library's RCCL communicator could not be created
            if world == 1:
                raise
            err = "%s: %s" % (type(e).__name__, e)
        if world > 1 and not all_agree(err is None):
            # every rank must take the same path: one failure sends all of them to the torch.distributed driver
            if g is not None:
                g.close()
                g = None
            sys.stderr.write("rank %d: library slab driver unavailable (%s); using cfd_amd.slab over "
                             "torch.distributed\n" % (rank, err or "another rank failed"))
            fallback = err or "another rank could not create its slab"
            dist.destroy_process_group()
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
            init_rccl(torch.device("cuda", local_rank))
            c_driver, red_dev = False, "cuda"
    else:
        g = None
    if g is None:
        g = SlabGraph(params, fields, active, rank=rank, world=world)
    if watchdog is not None:
        watchdog.cancel()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        g.sync()

    tw = time.perf_counter()
    for _ in range(a.warmup):
        g.step(1)
    barrier()
    # warm up by time as well: every rank derives the same number of extra steps from the slowest rank's time
    tw = time.perf_counter() - tw
    if world > 1:
        t = torch.tensor([tw], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tw = float(t.item())
    warmup_extra = 0
    if tw < a.warmup_seconds:
        per_step = tw / a.warmup if a.warmup > 0 else 1e-3
        warmup_extra = min(100000, int((a.warmup_seconds - tw) / max(per_step, 1e-6)) + 1)
        for _ in range(warmup_extra):
            g.step(1)
        barrier()
    read_sclk = sclk_reader(local_rank)
    sclk = []
    g.graph.set_timing(os.environ.get("CFD_BENCH_KERNEL_TIMING", "1") != "0")
    g.graph.sweep_timing(reset=True)
    timed_exchange = c_driver and hasattr(g, "m") and os.environ.get("CFD_BENCH_EXCHANGE_TIMING", "1") != "0"
    if timed_exchange:
        g.m.exchange_timing(reset=True)
        g.m.set_exchange_timing(True)
    t0 = time.perf_counter()
    for i in range(a.steps):
        g.step(1)
    sclk.append(read_sclk())        # the host has queued the steps and the GPU is still running them
    barrier()
    elapsed = time.perf_counter() - t0
    sclk.append(read_sclk())
    kernel_ms, launches = g.graph.sweep_timing(reset=True)
    fused_ms, fused_launches = g.graph.kernel_timing(1, reset=True)
    pair_ms, pair_launches = g.graph.kernel_timing(2, reset=True)
    trio_ms, trio_launches = g.graph.kernel_timing(3, reset=True)
    g.graph.set_timing(False)
    exchange = None
    if timed_exchange:
        g.m.set_exchange_timing(False)
        x_ms, x_n = g.m.exchange_timing(reset=True)
        if x_n:
            # rank 0's figure: from the moment its stream gets to the exchange to the arrival of its ghost planes
            exchange = {"per_step": x_n / a.steps, "avg_ms": x_ms / x_n, "ms_per_step": x_ms / a.steps}
    base_name = g.graph.kernel_name()
    rank_spread = None
    if world > 1:
        own = elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([own], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        # how far apart the ranks finished the timed steps (the value uses the slowest)
        rank_spread = {"fastest_rank_ms_per_step": float(t.item()) / a.steps * 1e3,
                       "slowest_rank_ms_per_step": elapsed / a.steps * 1e3}

    if c_driver and a.virtual_slabs > 1:
        # launches were timed on the first of the K slabs only: scale to the whole GPU
        scale = a.virtual_slabs
        kernel_ms, fused_ms, pair_ms, trio_ms = kernel_ms * scale, fused_ms * scale, pair_ms * scale, trio_ms * scale
    cells_per_gpu = a.nx * a.ny * a.nz                   # this rank's cells (rank 0 reports)
    total_cells = global_nx * a.ny * a.nz
    final_density = g.total_density()

    strict = None
    if a.arith == "fast" and world == 1 and not c_driver:
        # the strict-order kernel on the same workload, a few steps, for the record
        g.close()
        params["flags"] = a.flags
        gs = SlabGraph(params, fields, active, rank=rank, world=world)
        barrier_s = lambda: (torch.cuda.synchronize(), gs.sync())
        # warm up for 0.3 s as well, so that the leg is timed at the clock the GPU sustains
        tw0 = time.perf_counter()
        while True:
            gs.step(3)
            barrier_s()
            if time.perf_counter() - tw0 >= min(0.3, a.warmup_seconds):
                break
        gs.graph.set_timing(True)
        gs.graph.sweep_timing(reset=True)
        ts = time.perf_counter()
        nstrict = max(2, min(30, a.steps))
        gs.step(nstrict)
        barrier_s()
        es = time.perf_counter() - ts
        kms, kl = gs.graph.sweep_timing(reset=True)
        gs.graph.kernel_timing(1, reset=True)
        strict = {"value": cells_per_gpu * SWEEPS_PER_STEP * nstrict / es, "unit": "cell-updates/s",
                  "steps": nstrict, "kernel": gs.graph.kernel_name(), "avg_launch_ms": kms / max(kl, 1),
                  "achieved_GBps": cells_per_gpu * (BYTES_PER_UPDATE_2D if gs.graph.live_fields == 5
                                                    else BYTES_PER_UPDATE_3D) / (kms / 1e3 / max(kl, 1)) / 1e9}
        gs.close()
    fast_vs_strict = None
    if a.arith == "fast" and world == 1 and not c_driver and cells_per_gpu <= 4096 * 4096:
        fast_vs_strict = one_step_fast_vs_strict(params, fields, active, a.flags)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        own = two_d and a.case == "synthetic" and not a.cpu_sample and cells_per_gpu <= 4096 * 4096
        cpu = cpu_baseline(a.cpu_sample or 2048, (params, fields, active) if own else None)
    del fields
    updates = total_cells * SWEEPS_PER_STEP * a.steps
    bytes_per_update = BYTES_PER_UPDATE_2D if two_d else BYTES_PER_UPDATE_3D
    # Launch kinds in the timed region, per step: three plain sweeps; or two plus one fused with the filter; or one
    # plain sweep plus one launch that does two sweeps and the filter in one pass; or ONE launch that does the whole
    # step (three sweeps and the filter) in one pass.  At N > 1 a sweep is two or three launches (interior planes,
    # then the planes next to a ghost plane); their times are summed per sweep.
    plain_sweeps = (0 if trio_launches else 1 if pair_launches else (2 if fused_launches else 3)) * a.steps

    def kind_record(name, total_ms, nlaunch, passes, updates_per_cell, filters_per_cell, side_by_side=False):
        """Roofline figures of one launch kind.  Algorithmic bytes count the cell-update units only
        (bytes_per_update each); the filter pass a fused launch also does is listed but not credited.
        (For the two-sweeps launch that figure, 2 x 81 B per cell in 2-D, is also within 1 B of what the launch
        itself has to move: source fields + mask + one filter input read, two time levels written = 161 B.)"""
        avg_s = total_ms / 1e3 / max(passes, 1)
        if side_by_side:
            # a slab's one-pass step is two launches that run side by side (the blocks next to a neighbour on the
            # side stream): their durations overlap, so the pass is priced at the wall clock of a step
            avg_s = elapsed / a.steps
        alg = cells_per_gpu * bytes_per_update * updates_per_cell
        rec = {"kernel": name, "avg_pass_ms": avg_s * 1e3, "passes": passes, "launches": nlaunch,
               "total_ms": total_ms, "cell_updates_per_cell": updates_per_cell,
               "filter_passes_per_cell": filters_per_cell, "algorithmic_bytes": alg,
               "achieved": alg / avg_s / 1e9 if passes else 0.0}
        if side_by_side:
            rec["timing"] = "wall clock per step: this pass is %d launches side by side" % (nlaunch // max(passes, 1))
        rec["frac"] = rec["achieved"] / HBM_PEAK_GBPS
        return rec

    kinds = []
    untimed = not (launches or fused_launches or pair_launches or trio_launches)
    if untimed:
        # per-launch timing was off (a decomposed run: see where set_timing is called): the step at the wall clock
        kinds.append(kind_record("whole step (%s)" % base_name, elapsed * 1e3, 0, a.steps, SWEEPS_PER_STEP, 1))
        kinds[0]["timing"] = "wall clock per step (per-launch events off)"
    elif plain_sweeps:
        kinds.append(kind_record(base_name, kernel_ms, launches, plain_sweeps, 1, 0))
    if trio_launches and not untimed:
        kinds.append(kind_record("sweep2d_trio (three sweeps + filter in one pass)", trio_ms, trio_launches, a.steps, 3, 1,
                                 side_by_side=trio_launches > a.steps and a.virtual_slabs == 1))
    if fused_launches and not untimed:
        kinds.append(kind_record(base_name + " + filter", fused_ms, fused_launches, a.steps, 1, 1))
    if pair_launches and not untimed:
        kinds.append(kind_record("sweep2d_pair (two sweeps + filter in one pass)", pair_ms, pair_launches, a.steps, 2, 1))
    if a.virtual_slabs > 1:
        # several slabs on one GPU run side by side: per-launch durations overlap and say little; one record for the
        # whole step, priced at the wall clock
        per_launch = kinds
        kinds = [kind_record("whole step, %d slabs side by side on this GPU" % a.virtual_slabs, elapsed * 1e3,
                             sum(r["launches"] for r in per_launch), a.steps, SWEEPS_PER_STEP, 1)]
        kinds[0]["timing"] = "wall clock per step"
        kinds[0]["launch_kinds_on_the_first_slab"] = [r["kernel"] for r in per_launch]
    dominant = max(kinds, key=lambda r: r["total_ms"])
    sweep_ms_per_step = sum(r["total_ms"] for r in kinds) / a.steps
    workload = ("cavity_2d %dx%dx1 per GPU" % (a.nx, a.ny)) if (two_d and a.case == "cavity") else \
        ("synthetic_2d %dx%dx1 per GPU" % (a.nx, a.ny)) if two_d else \
        ("synthetic_3d %dx%dx%d per GPU" % (a.nx, a.ny, a.nz))
    if a.strong:
        workload += " (rank 0's share of %d global x-planes)" % global_nx
    # the kernel the timed steps ran in; cfd_kernel_name names the single sweep's kernel of the handle
    kernel_name = dominant["kernel"].split(" ")[0] if dominant["kernel"].startswith(("sweep2d_trio", "sweep2d_pair")) \
        else base_name
    out = {
        "metric": "fp64 cell-updates/sec",
        "value": updates / elapsed,
        "unit": "cell-updates/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "warmup_extra_steps": warmup_extra,
        "sclk_mhz": {"while_queued_steps_run": sclk[0], "after_the_run": sclk[1],
                     "source": "pp_dpm_sclk (the active DPM level; the in-kernel clock under fp64 load reads lower)"},
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if a.strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "cells_per_gpu": cells_per_gpu, "sweeps_per_step": SWEEPS_PER_STEP,
                   "filter_passes_per_step": 1,
                   "decomposition": "x-slabs, %d rank(s)%s%s" % (
                       world, ", %d slabs on this GPU" % a.virtual_slabs if a.virtual_slabs > 1 else "",
                       (", driven by the library (cfd_multi_*: %s)" % (
                           "RCCL" if world > 1 else "RCCL loopback: the slab is its own neighbour on both sides, a "
                           "periodic domain" if a.loopback else "device copies"))
                       if c_driver else (", driven by cfd_amd.slab over torch.distributed" if world > 1 else "")),
                   "kernel": kernel_name, "arithmetic": a.arith, "dt": a.dt,
                   "parity": "strict arithmetic: bit-exact against the CPU oracle (tests); fast (re-associated) arithmetic, "
                             "which this value is measured on unless --arith strict: <= 1e-13 of the field scale per "
                             "sweep against strict; the strict kernels' rate is under \"strict\"",
                   "fast_vs_strict_one_step_on_this_workload": fast_vs_strict,
                   "halo_exchange": exchange,
                   "ranks": rank_spread,
                   "preflight": preflight_report if world > 1 else "not applicable at N = 1 (no decomposition)",
                   "fields_finite_at_end": bool(final_density == final_density and abs(final_density) != float("inf"))},
        # the launch kind that takes most of the timed region; every kind is listed under "kernels"
        "roofline": {"bound": "hbm", "achieved": dominant["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": dominant["frac"], "traffic": measured_traffic(workload, dominant["kernel"]),
                     "kernel": dominant["kernel"], "bytes_per_cell_update": bytes_per_update,
                     "cell_updates_per_cell_per_launch": dominant["cell_updates_per_cell"],
                     "avg_launch_ms": dominant["avg_pass_ms"], "launches": dominant["launches"],
                     "kernels": kinds,
                     # the other limit of the same launch: fp64 vector issue slots in use (counter record of this build)
                     "valu": valu_issue(workload, dominant["kernel"], dominant["avg_pass_ms"], sclk[0]),
                     # all sweep-type launches of a step together, cell-update bytes only
                     "step": {"sweep_kernel_ms": sweep_ms_per_step,
                              "achieved": SWEEPS_PER_STEP * cells_per_gpu * bytes_per_update / (sweep_ms_per_step / 1e3) / 1e9,
                              "frac": SWEEPS_PER_STEP * cells_per_gpu * bytes_per_update / (sweep_ms_per_step / 1e3) / 1e9
                              / HBM_PEAK_GBPS}},
    }
    if world > 1:
        out["roofline"]["note"] = ("slab run: a plain sweep is three launches (interior planes on the main stream, the "
                                   "planes next to the neighbours on a side stream beside it, where they share the GPU "
                                   "with the interior launch and the RCCL transfers); their durations are summed per sweep")
    if fallback:
        out["config"]["driver_fallback"] = fallback
    if strict is not None:
        out["strict"] = strict
    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        os.write(line_fd, (json.dumps(out) + "\n").encode())
    if strict is None:
        g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
