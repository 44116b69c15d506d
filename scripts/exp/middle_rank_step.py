"""What the slab form of the one-pass step costs a rank that has a neighbour on both sides, measured on ONE GPU: a
4096 x 4096 slab with ghost planes on both sides whose "neighbours" are itself (its last three planes go into its lower
ghost planes, its first three into the upper ones, copied on a side stream), stepped exactly as the slab driver steps a
rank: exchange on the side stream beside the blocks that read owned planes only, the blocks next to a ghost plane after
it on the side stream.  Printed beside the undecomposed step, alternating.  DELAY_US adds a busy-wait kernel of that many
microseconds in front of the copies, in place of the latency of a real exchange between two processes.
usage: middle_rank_step.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cfd_amd import cases
from cfd_amd.solver import AdjGraph

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
delay_us = float(os.environ.get("DELAY_US", "0"))
nz = int(os.environ.get("NZ", "1"))        # NZ > 1: an NZ^3 grid, three sweeps per step with a one-plane exchange each
if nz > 1:
    nx = ny = nz
    params, fields, active = cases.synthetic_3d(nx, ny, nz, seed=3)
else:
    nx = ny = 4096
    params, fields, active = cases.synthetic_2d(nx, ny, seed=1)
params = dict(params, dt=1e-7, flags=int(os.environ.get("FLAGS", "4")))
und = AdjGraph(params, fields, active)
g = AdjGraph(dict(params, ghost_lo=1, ghost_hi=1), fields, active)
if nz == 1:
    act = np.ascontiguousarray(active, dtype=np.uint8).reshape(nx, ny)
    g.set_ghost_active(act[-2:], act[:2], depth=2)
    assert g.one_pass_step_available()
main = torch.cuda.ExternalStream(g.stream())
side = torch.cuda.Stream()


class Span:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


_spans = {}


def span(ptr, nbytes):
    if ptr not in _spans:
        _spans[ptr] = torch.as_tensor(Span(ptr, nbytes), device="cuda")
    return _spans[ptr]


cycles = int(delay_us * float(os.environ.get("CYCLES_PER_US", "100")))     # torch.cuda._sleep's unit is measured below


_events = [torch.cuda.Event() for _ in range(64)]


def slab_steps_3d(n):
    k = 0
    done = _events[k]; k = (k + 1) % 64
    done.record(main)
    for _ in range(n):
        for sweep in range(3):
            side.wait_event(done)
            (slo, shi, rlo, rhi), nbytes = g.halo_planes(1)
            with torch.cuda.stream(side):
                if cycles:
                    torch.cuda._sleep(cycles)
                span(rlo, nbytes).copy_(span(shi, nbytes), non_blocking=True)
                span(rhi, nbytes).copy_(span(slo, nbytes), non_blocking=True)
            g.sweep_begin(sweep == 0, sweep == 2)
            ev = _events[k]; k = (k + 1) % 64
            ev.record(side)
            main.wait_event(ev)
            g.sweep_end()
            done = _events[k]; k = (k + 1) % 64
            done.record(main)


def slab_steps(n):
    if nz > 1:
        return slab_steps_3d(n)
    k = 0
    done = _events[k]; k = (k + 1) % 64
    done.record(main)
    for _ in range(n):
        side.wait_event(done)
        (slo, shi, rlo, rhi), nbytes = g.halo_planes(3)
        with torch.cuda.stream(side):
            if cycles:
                torch.cuda._sleep(cycles)
            span(rlo, nbytes).copy_(span(shi, nbytes), non_blocking=True)
            span(rhi, nbytes).copy_(span(slo, nbytes), non_blocking=True)
        g.step_one_pass_begin()
        g.step_one_pass_end(stream=side.cuda_stream)
        ev = _events[k]; k = (k + 1) % 64
        ev.record(side)
        main.wait_event(ev)
        done = _events[k]; k = (k + 1) % 64
        done.record(main)


def timed(fn, sync):
    sync()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    sync()
    return (time.perf_counter() - t0) / steps * 1e3, (t1 - t0) / steps * 1e3


def sync_all():
    torch.cuda.synchronize()
    g.sync()
    und.sync()


if cycles:      # how long the sleep really is
    torch.cuda._sleep(10); torch.cuda.synchronize()       # the first launch loads the kernel
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); torch.cuda._sleep(cycles); b.record(); torch.cuda.synchronize()
    print("delay kernel: %.1f us" % (a.elapsed_time(b) * 1e3))
warm = 300 if nz == 1 else 20
und.step(warm); slab_steps(warm); sync_all()       # warm up (clock settles under load)
for r in range(3):
    print("undecomposed            %.4f ms/step (host issue %.4f)" % timed(lambda: und.step(steps), sync_all))
    print("slab, neighbours both sides (self-exchange%s)  %.4f ms/step (host issue %.4f)" % (
        (", +%g us" % delay_us if delay_us else "",) + timed(lambda: slab_steps(steps), sync_all)), flush=True)
