#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel name from *_counter_collection.csv files."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    print(name, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in cs.items()})
