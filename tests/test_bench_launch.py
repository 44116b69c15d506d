"""bench.py's own launch path for N > 1, rehearsed on the CPU: `python bench.py --gpus 2 --dry-run` must start
two fresh ranks, rendezvous (gloo), reduce, print ONE JSON line with n_gpus = 2 and exit 0; without --dry-run and
without a GPU the ranks fail loudly and the parent passes their failure on."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                           *extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)


def test_bench_gpus_2_launches_its_own_ranks():
    r = _run("--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["value"] is None


def test_bench_gpus_2_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the real multi-rank run is not a CPU test")
    r = _run()
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [(), ("--virtual-slabs", "2"), ("--nz", "24", "--nx", "40", "--ny", "36"), ("--loopback",),
                                   ("--loopback", "--nz", "24", "--nx", "40", "--ny", "36")])
def test_bench_line_on_the_gpu(extra):
    """The command the round-end driver runs, shortened: ONE JSON line with the contract's keys, the roofline
    and the finite-fields flag -- undecomposed, as two slabs behind the library's slab driver, 3-D, and as a rank
    between two others (the slab exchanges with itself through the library's RCCL path: stdout must still hold the
    one line although RCCL prints a banner, and the line reports the exchanges)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    args = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--warmup-seconds", "0", "--no-cpu-baseline"]
    if "--nz" not in extra:
        args += ["--nx", "256", "--ny", "512"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, *extra], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    if "--loopback" in extra:
        assert r.stdout.strip() == lines[0], r.stdout        # nothing but the line
    rec = json.loads(lines[0])
    if "--loopback" in extra:
        assert rec["config"]["halo_exchange"]["per_step"] in (1.0, 3.0) and rec["config"]["halo_exchange"]["avg_ms"] > 0
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["dtype"] == "f64" and rec["value"] > 0
    assert rec["config"]["fields_finite_at_end"] is True
    assert rec["roofline"]["bound"] == "hbm" and 0 < rec["roofline"]["frac"] < 1
