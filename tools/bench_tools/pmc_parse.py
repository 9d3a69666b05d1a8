"""parse rocprofv3 --pmc counter_collection csv files: per kernel name, mean counter value (KB) over dispatches"""
import csv, glob, json, sys
from collections import defaultdict
out = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        out[(r["Counter_Name"], r["Kernel_Name"].split("(")[0][:90])].append(float(r["Counter_Value"]))
res = [{"counter": c, "kernel": k, "dispatches": len(v), "mean_value": sum(v) / len(v)} for (c, k), v in sorted(out.items())]
print(json.dumps(res, indent=1))
