#!/usr/bin/env python
"""rocprofv3 --pmc counter_collection.csv files -> one markdown table: mean counter value per dispatch and kernel.

    python scripts/pmc_table.py <csv> [<csv> ...]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name


def main():
    vals = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> values per dispatch
    dur = defaultdict(dict)                           # kernel -> dispatch id -> ns
    counters = []
    for path in sys.argv[1:]:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k, c = short(row["Kernel_Name"]), row["Counter_Name"]
                vals[k][c].append(float(row["Counter_Value"]))
                if c not in counters:
                    counters.append(c)
                dur[k][(path, row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    print("| kernel | dispatches | avg us (under counters) | " + " | ".join(counters) + " |")
    print("|---|---|---|" + "---|" * len(counters))
    for k in sorted(vals):
        d = list(dur[k].values())
        n = max(len(v) for v in vals[k].values())
        cells = []
        for c in counters:
            v = vals[k].get(c)
            cells.append("%.4g" % (sum(v) / len(v)) if v else "")
        print("| %s | %d | %.1f | %s |" % (k, n, sum(d) / len(d) / 1e3, " | ".join(cells)))


if __name__ == "__main__":
    main()
