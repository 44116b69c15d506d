"""Summarises the counter CSVs written by scripts/pmc.sh: per-kernel mean of every counter per launch."""
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/%s/p*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
