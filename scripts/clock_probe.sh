# usage: bash scripts/clock_probe.sh  -- shader clock during the sweep kernel (GRBM_GUI_ACTIVE / duration) with and without memory traffic
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/clk
for d in 0 3 1 2; do
  CFD_DBG=$d CFD_VARIANT=${1:-21} FLAGS=4 timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk/d$d -- python3 scripts/prof_target.py > gpurun_out/clk/d$d.log 2>&1
done
python3 - <<'PY'
import csv, glob
for d in (0, 3, 1, 2):
    for f in glob.glob("gpurun_out/clk/d%d/**/*counter_collection.csv" % d, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "sweep2d" in r["Kernel_Name"]]
        for r in rows[-3:]:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            print("dbg", d, r["Kernel_Name"][:40], "GUI_ACTIVE %.0f" % float(r["Counter_Value"]), "dur %d ns" % dur, "clock %.0f MHz" % (float(r["Counter_Value"]) / dur * 1e3))
PY
