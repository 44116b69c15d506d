"""Does data written a moment ago come back faster than data streamed from HBM?

Two chained streaming passes (b = a + 1, then c = b * 2) over a large array, done band by band: with
small bands the second pass reads what the first has just written (it may still sit in the memory-side
cache), with one band covering the whole array it cannot.  Prints the aggregate rate per band size.
"""
import sys
import time

import torch


def main():
    dev = torch.device("cuda:0")
    total = 1 << 27  # doubles: 1 GiB per array
    a = torch.rand(total, dtype=torch.float64, device=dev)
    b = torch.empty_like(a)
    c = torch.empty_like(a)
    for band_mb in (8, 16, 32, 64, 128, 256, 512, 1024):
        n = band_mb * (1 << 20) // 8
        bands = [(s, min(total, s + n)) for s in range(0, total, n)]

        def run():
            for s, e in bands:
                torch.add(a[s:e], 1.0, out=b[s:e])
                torch.mul(b[s:e], 2.0, out=c[s:e])

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        moved = 4 * total * 8
        print(f"band {band_mb:5d} MiB: {dt * 1e3:7.3f} ms per pair of passes, {moved / dt / 1e12:5.2f} TB/s", flush=True)
    # the lagged form: pass 2 runs one band behind pass 1 (what a sweep pipeline would do)
    for band_mb in (16, 32, 64, 128):
        n = band_mb * (1 << 20) // 8
        bands = [(s, min(total, s + n)) for s in range(0, total, n)]

        def run_lagged():
            prev = None
            for s, e in bands:
                torch.add(a[s:e], 1.0, out=b[s:e])
                if prev is not None:
                    torch.mul(b[prev[0]:prev[1]], 2.0, out=c[prev[0]:prev[1]])
                prev = (s, e)
            torch.mul(b[prev[0]:prev[1]], 2.0, out=c[prev[0]:prev[1]])

        for _ in range(3):
            run_lagged()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run_lagged()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"lagged band {band_mb:5d} MiB: {dt * 1e3:7.3f} ms, {4 * total * 8 / dt / 1e12:5.2f} TB/s", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
