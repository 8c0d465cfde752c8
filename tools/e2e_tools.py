"""End-to-end time of the host tools on a synthetic binary archive (read + flatten + GPU + write),
beside the GPU-only numbers of bench.py / time_accum.py."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kaldi-lattice-utils_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import klat_amd, latfmt
U = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
tmp = sys.argv[3] if len(sys.argv) > 3 else "/tmp"
BIN = os.path.join(ROOT, "kaldi-lattice-utils_amd", "bin")
a = klat_amd.synth(20260001, np.full(U, S, np.int32), np.full(U, 3, np.int32))
txt, ark = os.path.join(tmp, "e2e.txt"), os.path.join(tmp, "e2e.ark")
t0 = time.perf_counter()
lat = np.repeat(np.arange(U), np.diff(a["arc_off"]))
dur = a["state_time"][a["state_off"][lat] + a["dst"]] - a["state_time"][a["state_off"][lat] + a["src"]]
tids = np.array(["_".join(["7"] * d) for d in range(int(dur.max()) + 1)])
cols = [a["src"].astype(str), a["dst"].astype(str), a["ilabel"].astype(str)]
line = cols[0]
for c in cols[1:]:
    line = np.char.add(np.char.add(line, "\t"), c)
w = np.char.add(np.char.add(np.char.mod("%.9g", a["graph"]), ","), np.char.add(np.char.add(np.char.mod("%.9g", a["acoustic"]), ","), tids[dur]))
line = np.char.add(np.char.add(line, "\t"), w)
with open(txt, "w") as f:
    for u in range(U):
        a0, a1 = int(a["arc_off"][u]), int(a["arc_off"][u + 1])
        f.write("utt%06d\n" % u)
        f.write("\n".join(line[a0:a1]))
        f.write("\n%d\n\n" % (a["state_off"][u + 1] - a["state_off"][u] - 1))
print("wrote text archive: %.1f MB in %.1f s" % (os.path.getsize(txt) / 1e6, time.perf_counter() - t0), flush=True)
def run(*cmd):
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(BIN, cmd[0])] + list(cmd[1:]), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return dt
dt = run("klat-lattice-copy", "ark:" + txt, "ark:" + ark)
A = len(a["dst"])
print("klat-lattice-copy text -> binary (%.1f MB): %.2f s" % (os.path.getsize(ark) / 1e6, dt), flush=True)
# (the tools that disambiguate by transcript length -- best-path2, length-dist, the position index --
# are left out: on these 2 000-state chains the reference's own (length, state) expansion explodes)
for tool, extra in (("lattice-to-word-frame-post", []), ("lattice-word-index-segment", [])):
    out = os.path.join(tmp, "e2e.out")
    for rep in range(2):
        dt = run(tool, "--acoustic-scale=0.1", *extra, "ark:" + ark, "ark:" + out)
    print("%-36s %.2f s end to end -> %.1f M arcs/s (output %.1f MB)" % (tool, dt, A / dt / 1e6, os.path.getsize(out) / 1e6), flush=True)
