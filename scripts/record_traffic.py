"""Writes profiles/traffic.json from the counter summary of scripts/pmc.sh (separate rocprofv3 --pmc passes):
vector-ALU instruction counts, and HBM bytes per launch of the 2-D sweep kernels = 2 x FETCH_SIZE (gfx950 tallies 128-B read requests as 64 B;
checked in the same passes on time_filter_kernel, whose byte counts are known) + WRITE_SIZE, both in KB.
The record is stamped with the SHA-256 of the kernel sources; bench.py reports it only for that build.
usage: python scripts/record_traffic.py gpurun_out/<tag>/pmc2d_summary.txt [profiles/<round>/<copy>.txt]"""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
kern, cur = {}, None
for line in open(src):
    if not line.startswith(" "):
        cur = line.strip(); kern[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+mean\s+(\S+)", line)
        if m and cur: kern[cur][m.group(1)] = float(m.group(2))
h = hashlib.sha256()
for name in ("cfd_hip.hip", "sweep_math.h"):
    h.update(open(os.path.join(ROOT, "cfd_amd", "csrc", name), "rb").read())
cells = 4096 * 4096
out = {"kernel_source_sha256": h.hexdigest(), "source": src if len(sys.argv) < 3 else sys.argv[2],
       "synthetic_2d 4096x4096x1 per GPU": []}
filt = next((v for k, v in kern.items() if k.startswith("time_filter_kernel")), None)
if filt:
    out["calibration_on_time_filter_kernel"] = {"known_read_bytes": 2 * 5 * 8 * cells + cells, "FETCH_SIZE_KB_raw": filt.get("FETCH_SIZE"),
                                                "known_written_bytes": 5 * 8 * cells, "WRITE_SIZE_KB": filt.get("WRITE_SIZE")}
for prefix, alg in (("sweep2d_stream", 81 * cells), ("sweep2d_pair", 2 * 81 * cells)):
    for k, v in kern.items():
        if prefix in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out["synthetic_2d 4096x4096x1 per GPU"].append({
                "kernel_prefix": prefix, "kernel": k, "FETCH_SIZE_KB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
                "hbm_bytes_per_launch": int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024),
                "algorithmic_bytes_per_launch": alg,
                "correction": "FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B)",
                # vector-ALU side of the same launches (another pass of the same script): wave-level instructions and
                # the 4-cycle issue slots they occupied, summed over all SIMDs
                "valu_wave_instructions_per_launch": v.get("SQ_INSTS_VALU"),
                "valu_issue_slots_per_launch": v.get("SQ_ACTIVE_INST_VALU"),
                "wave_slots_per_launch": v.get("SQ_WAVE_CYCLES"), "waves_per_launch": v.get("SQ_WAVES")})
            break
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
