"""Writes profiles/traffic.json from the counter summaries of scripts/pmc.sh (separate rocprofv3 --pmc passes):
vector-ALU instruction counts, and HBM bytes per launch of the sweep kernels = c x FETCH_SIZE + WRITE_SIZE (both in
KB).  c: gfx950 tallies a 128-B read request as 64 B, so FETCH_SIZE under-reports streaming reads by a factor that
is calibrated IN THE SAME PASSES on time_filter_kernel, whose byte counts are known (2-D: c = 2.0x within 1 %; the
3-D access pattern gets its own calibration from its own run).  WRITE_SIZE reads the bytes as they are.
The record is stamped with the SHA-256 of the kernel sources and build options; bench.py reports it only for that build.
usage: python scripts/record_traffic.py <pmc2d_summary.txt> <copy kept under profiles/> [<pmc3d_summary.txt> <copy> <edge>]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash


def parse(path):
    kern, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip(); kern[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+mean\s+(\S+)", line)
            if m and cur: kern[cur][m.group(1)] = float(m.group(2))
    return kern


def entries(kern, cells, nf, table, out_list, calib):
    filt = next((v for k, v in kern.items() if k.startswith("time_filter_kernel")), None)
    factor = 2.0
    if filt and filt.get("FETCH_SIZE"):
        known_read = 2 * nf * 8 * cells + cells
        factor = known_read / (filt["FETCH_SIZE"] * 1024)
        calib.update({"known_read_bytes": known_read, "FETCH_SIZE_KB_raw": filt.get("FETCH_SIZE"),
                      "read_bytes_per_counted_byte": factor,
                      "known_written_bytes": nf * 8 * cells, "WRITE_SIZE_KB": filt.get("WRITE_SIZE")})
    for prefix, alg in table:
        for k, v in kern.items():
            if prefix in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                out_list.append({
                    "kernel_prefix": prefix, "kernel": k, "FETCH_SIZE_KB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
                    "hbm_bytes_per_launch": int(factor * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024),
                    "algorithmic_bytes_per_launch": alg,
                    "correction": "FETCH_SIZE x %.3f (calibrated on time_filter_kernel in the same passes; gfx950 counts 128-B read "
                                  "requests as 64 B)" % factor,
                    # vector-ALU side of the same launches (another pass of the same script): wave-level instructions and
                    # the 4-cycle issue slots they occupied, summed over all SIMDs
                    "valu_wave_instructions_per_launch": v.get("SQ_INSTS_VALU"),
                    "valu_issue_slots_per_launch": v.get("SQ_ACTIVE_INST_VALU"),
                    "wave_slots_per_launch": v.get("SQ_WAVE_CYCLES"), "waves_per_launch": v.get("SQ_WAVES")})
                break


src = sys.argv[1]
out = {"kernel_source_sha256": kernel_source_hash(), "source": sys.argv[2] if len(sys.argv) > 2 else src,
       "synthetic_2d 4096x4096x1 per GPU": [], "calibration_on_time_filter_kernel": {}}
cells = 4096 * 4096
entries(parse(src), cells, 5, (("sweep2d_trio", 3 * 81 * cells), ("sweep2d_stream", 81 * cells), ("sweep2d_pair", 2 * 81 * cells)),
        out["synthetic_2d 4096x4096x1 per GPU"], out["calibration_on_time_filter_kernel"])
if len(sys.argv) > 5:
    edge = int(sys.argv[5])
    key = "synthetic_3d %dx%dx%d per GPU" % (edge, edge, edge)
    out[key], out["calibration_3d_on_time_filter_kernel"] = [], {}
    out["source_3d"] = sys.argv[4]
    entries(parse(sys.argv[3]), edge ** 3, 6, (("sweep3d_stream", 97 * edge ** 3),), out[key], out["calibration_3d_on_time_filter_kernel"])
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
