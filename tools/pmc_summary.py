"""Mean per launch of every counter found under a tools/pmc_collect.sh output directory (CSV on stdout)."""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, counter) -> dispatch -> value
for path in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            if not name.startswith(("k_", "void k_")):
                continue
            name = name.replace("void ", "")
            # a counter reported per XCD / instance appears once per dimension: sum them per dispatch
            acc[(name, row["Counter_Name"])][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
print("kernel,counter,mean_per_launch,launches")
for (k, c), d in sorted(acc.items()):
    print(f'"{k}",{c},{sum(d.values()) / len(d):.6g},{len(d)}')
