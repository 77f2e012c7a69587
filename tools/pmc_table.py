"""Average every counter of the k_eval2 dispatches found under the given rocprofv3 --pmc output dirs."""
import csv, glob, os, sys
acc = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "k_eval2" not in row["Kernel_Name"]:
                    continue
                a = acc.setdefault(row["Counter_Name"], [0.0, 0])
                a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k:28s} {acc[k][0] / acc[k][1]:16.1f}   ({acc[k][1]} dispatches)")
