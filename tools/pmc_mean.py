#!/usr/bin/env python3
"""Mean per-launch value of every PMC counter per kernel in a rocprofv3 counter_collection.csv tree.
usage: tools/pmc_mean.py gpurun_out/prof_dir [kernel-substring]"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:int(os.environ.get("PMC_NAME_CHARS", "48"))], r["Counter_Name"])].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (k, c), v in sorted(acc.items()):
    if flt in k:
        print("%-50s  %-28s n=%-5d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
