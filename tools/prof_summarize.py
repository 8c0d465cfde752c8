#!/usr/bin/env python3
"""Condenses rocprofv3 output (gpurun_out/prof_{stats,fetch,write}) into the files kept under
profiles/: the --kernel-trace --stats table, and per-kernel HBM traffic from the PMC passes
(FETCH_SIZE and WRITE_SIZE collected in separate runs; both in KiB; FETCH_SIZE doubled as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes for 16 B/lane streaming reads on gfx950).

usage: tools/prof_summarize.py <round-tag> [dir-prefix]   e.g. r01, or r03_chain chainprof (gpurun_out/chainprof_{stats,fetch,write})
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prefix = sys.argv[2] if len(sys.argv) > 2 else "prof"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

newest = lambda fs: sorted(fs, key=os.path.getmtime)[-1:]
stats = newest(glob.glob(os.path.join(src, prefix + "_stats", "**", "*_kernel_stats.csv"), recursive=True))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats.csv" % tag))

traffic = collections.defaultdict(dict)
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = newest(glob.glob(os.path.join(src, "%s_%s" % (prefix, name), "**", "*_counter_collection.csv"), recursive=True))
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        traffic[k][counter + "_KiB_mean"] = sum(v) / len(v)
        traffic[k]["launches_" + name] = len(v)
out = {}
for k, t in traffic.items():
    rd = 2.0 * 1024.0 * t.get("FETCH_SIZE_KiB_mean", 0.0)  # gfx950: FETCH_SIZE counts 128-B requests as 64 B
    wr = 1024.0 * t.get("WRITE_SIZE_KiB_mean", 0.0)
    out[k] = dict(t, hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr,
                  hbm_bytes_per_launch=rd + wr)
# what the profile was taken from: the commit and a hash over the kernel sources (bench.py names both in
# roofline.traffic_source and says so when its own sources differ)
import hashlib
import subprocess
h = hashlib.sha1()
csrc = os.path.join(root, "kaldi-lattice-utils_amd", "csrc")
for name in sorted(os.listdir(csrc)):
    if name.endswith((".hip", ".h")):  # (kernels and the structures they share with the host)
        h.update(name.encode())
        h.update(open(os.path.join(csrc, name), "rb").read())
try:
    commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
    if subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "kaldi-lattice-utils_amd/csrc"], text=True).strip():
        commit += "+uncommitted"
except Exception:
    commit = "?"
out["__profile__"] = {"commit": commit, "source_hash": h.hexdigest()[:12]}
json.dump(out, open(os.path.join(dst, "%s_hbm_traffic.json" % tag), "w"), indent=1, sort_keys=True)
for k, t in out.items():
    if k == "__profile__":
        continue
    print("%-60s read %.3f GB  write %.3f GB" % (k[:60], t["hbm_read_bytes_per_launch"] / 1e9, t["hbm_write_bytes_per_launch"] / 1e9))
