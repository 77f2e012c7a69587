"""Average every counter of the k_eval2 dispatches found under the given rocprofv3 --pmc output dirs
(rocpd .db or csv)."""
import csv, glob, os, sqlite3, sys
acc = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "k_eval2" not in row["Kernel_Name"]:
                    continue
                a = acc.setdefault(row["Counter_Name"], [0.0, 0])
                a[0] += float(row["Counter_Value"]); a[1] += 1
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        c = sqlite3.connect(f)
        try:
            rows = c.execute("select counter_name, sum(value), count(*) from counters_collection "
                             "where kernel_name like '%k_eval2%' group by counter_name").fetchall()
        except sqlite3.Error:
            rows = []
        for name, total, n in rows:
            a = acc.setdefault(name, [0.0, 0])
            a[0] += total; a[1] += n
for k in sorted(acc):
    print(f"{k:32s} {acc[k][0] / acc[k][1]:18.1f}   ({acc[k][1]} dispatches)")
