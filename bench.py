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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="edge of the square CPU-baseline sample grid")
    return ap.parse_args()


def cpu_baseline(edge):
    """Times the C restatement (oracle/, kind = "port") on a bounded sample of the same recipe, and on the
    reference's own shipped configuration (301 x 151 x 1)."""
    from cfd_amd import cases
    from oracle import OracleB
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_params
    cores = min(16, os.cpu_count() or 1)

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

    rate, dt, nsteps = timed(*cases.synthetic_2d(edge, edge, seed=1), 12.0, 50)
    rate1, dt1, nsteps1 = timed(*cases.reference_case(), 3.0, 1000)
    return {"value": rate, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": "synthetic_2d %dx%d (same recipe and seed as the workload), %d steps = %d sweeps + filters, "
                      "C restatement with OpenMP, %.1f s" % (edge, edge, nsteps, SWEEPS_PER_STEP * nsteps, dt),
            "shipped_case_301x151": {"value": rate1, "unit": "cell-updates/s", "cores": cores,
                                     "sample": "the reference's configuration, %d steps, %.1f s" % (nsteps1, dt1)}}


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from a separate rocprofv3 --pmc pass, if one was recorded."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        for entry in rec.get(workload, []):
            if kernel.startswith(entry.get("kernel_prefix", "\0")):
                return entry.get("hbm_bytes_per_launch")
        return None
    except (OSError, ValueError):
        return None


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (a.gpus, a.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU path")
    torch.cuda.set_device(local_rank)
    from cfd_amd import cases
    from cfd_amd.slab import SlabGraph, init_rccl
    if world > 1:
        init_rccl(torch.device("cuda", local_rank))     # rank, world size and rendezvous from the environment

    global_nx = a.nx if a.strong else a.nx * world
    if a.strong:
        # this rank's share of the global planes (the first global_nx % world ranks hold one more)
        if global_nx < world:
            raise SystemExit("--strong: fewer x-planes than ranks")
        a.nx = global_nx // world + (1 if rank < global_nx % world else 0)
    two_d = a.nz == 1
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
    params["flags"] = a.flags | (4 if a.arith == "fast" else 0)
    g = SlabGraph(params, fields, active, rank=rank, world=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        g.sync()

    for _ in range(a.warmup):
        g.step(1)
    barrier()
    g.graph.set_timing(True)
    g.graph.sweep_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g.step(1)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = g.graph.sweep_timing(reset=True)
    fused_ms, fused_launches = g.graph.kernel_timing(1, reset=True)
    pair_ms, pair_launches = g.graph.kernel_timing(2, reset=True)
    g.graph.set_timing(False)
    base_name = g.graph.kernel_name()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    cells_per_gpu = a.nx * a.ny * a.nz                   # this rank's cells (rank 0 reports)
    total_cells = global_nx * a.ny * a.nz
    final_density = g.total_density()

    strict = None
    if a.arith == "fast" and world == 1:
        # the strict-order kernel on the same workload, a few steps, for the record
        g.close()
        params["flags"] = a.flags
        gs = SlabGraph(params, fields, active, rank=rank, world=world)
        gs.step(1)
        barrier_s = lambda: (torch.cuda.synchronize(), gs.sync())
        barrier_s()
        gs.graph.set_timing(True)
        gs.graph.sweep_timing(reset=True)
        ts = time.perf_counter()
        nstrict = max(2, min(5, a.steps))
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
    del fields
    updates = total_cells * SWEEPS_PER_STEP * a.steps
    bytes_per_update = BYTES_PER_UPDATE_2D if two_d else BYTES_PER_UPDATE_3D
    # Launch kinds in the timed region, per step: three plain sweeps; or two plus one fused with the filter; or one
    # plain sweep plus one launch that does two sweeps and the filter in one pass.  At N > 1 a sweep is two or three
    # launches (interior planes, then the planes next to a ghost plane); their times are summed per sweep.
    plain_sweeps = (1 if pair_launches else (2 if fused_launches else 3)) * a.steps

    def kind_record(name, total_ms, nlaunch, passes, updates_per_cell, filters_per_cell):
        """Roofline figures of one launch kind.  Algorithmic bytes count the cell-update units only
        (bytes_per_update each); the filter pass a fused launch also does is listed but not credited.
        (For the two-sweeps launch that figure, 2 x 81 B per cell in 2-D, is also within 1 B of what the launch
        itself has to move: source fields + mask + one filter input read, two time levels written = 161 B.)"""
        avg_s = total_ms / 1e3 / max(passes, 1)
        alg = cells_per_gpu * bytes_per_update * updates_per_cell
        rec = {"kernel": name, "avg_pass_ms": avg_s * 1e3, "passes": passes, "launches": nlaunch,
               "total_ms": total_ms, "cell_updates_per_cell": updates_per_cell,
               "filter_passes_per_cell": filters_per_cell, "algorithmic_bytes": alg,
               "achieved": alg / avg_s / 1e9 if passes else 0.0}
        rec["frac"] = rec["achieved"] / HBM_PEAK_GBPS
        return rec

    kinds = [kind_record(base_name, kernel_ms, launches, plain_sweeps, 1, 0)]
    if fused_launches:
        kinds.append(kind_record(base_name + " + filter", fused_ms, fused_launches, a.steps, 1, 1))
    if pair_launches:
        kinds.append(kind_record("sweep2d_pair (two sweeps + filter in one pass)", pair_ms, pair_launches, a.steps, 2, 1))
    dominant = max(kinds, key=lambda r: r["total_ms"])
    sweep_ms_per_step = sum(r["total_ms"] for r in kinds) / a.steps
    workload = ("cavity_2d %dx%dx1 per GPU" % (a.nx, a.ny)) if (two_d and a.case == "cavity") else \
        ("synthetic_2d %dx%dx1 per GPU" % (a.nx, a.ny)) if two_d else \
        ("synthetic_3d %dx%dx%d per GPU" % (a.nx, a.ny, a.nz))
    if a.strong:
        workload += " (rank 0's share of %d global x-planes)" % global_nx
    kernel_name = base_name
    out = {
        "metric": "fp64 cell-updates/sec",
        "value": updates / elapsed,
        "unit": "cell-updates/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if a.strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "cells_per_gpu": cells_per_gpu, "sweeps_per_step": SWEEPS_PER_STEP,
                   "filter_passes_per_step": 1, "decomposition": "x-slabs, %d rank(s)" % world,
                   "kernel": kernel_name, "arithmetic": a.arith, "dt": a.dt,
                   "fields_finite_at_end": bool(final_density == final_density and abs(final_density) != float("inf"))},
        # the launch kind that takes most of the timed region; every kind is listed under "kernels"
        "roofline": {"bound": "hbm", "achieved": dominant["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": dominant["frac"], "traffic": measured_traffic(workload, dominant["kernel"]),
                     "kernel": dominant["kernel"], "bytes_per_cell_update": bytes_per_update,
                     "cell_updates_per_cell_per_launch": dominant["cell_updates_per_cell"],
                     "avg_launch_ms": dominant["avg_pass_ms"], "launches": dominant["launches"],
                     "kernels": kinds,
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
    if strict is not None:
        out["strict"] = strict
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_sample)
        print(json.dumps(out), flush=True)
    if strict is None:
        g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
