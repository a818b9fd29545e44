#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per launch of the optimise kernel.
usage: pmc_summary.py OUT.json DIR [DIR...]   (each DIR = the -d directory of one --pmc pass)
Averaged: the dispatches of the optimise kernel with the bench's grid size, minus the first one (bench.py's synthetic
targets come from one forward-only launch of the same kernel); first-run, warm-up and timed launches are identical
work."""
import csv, glob, json, os, sys
from collections import Counter, defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
vals = defaultdict(list)
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if "dp_w4_kernel" in r["Kernel_Name"] or "dp_optimize_kernel" in r["Kernel_Name"]]
        if not rows:
            continue
        grid = Counter(r["Grid_Size"] for r in rows).most_common(1)[0][0]
        per_dispatch = defaultdict(lambda: defaultdict(float))
        first = min(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if r["Grid_Size"] == grid and int(r["Dispatch_Id"]) != first:
                per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        for disp in per_dispatch.values():
            for k, v in disp.items():
                vals[k].append(v)
res = {k: sum(v) / len(v) for k, v in sorted(vals.items())}
res["_launches_averaged"] = {k: len(v) for k, v in sorted(vals.items())}
res["_note"] = ("means over the full-size optimise-kernel launches of `bench.py --steps 10 --no-cpu-baseline --no-parity` "
                "(4096 frames x 50 iterations); one rocprofv3 --pmc pass per counter group; FETCH_SIZE/WRITE_SIZE in KB as reported "
                "(FETCH_SIZE under-counts wide reads 2x on gfx950, see MI355X_MICROARCH.md)")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
