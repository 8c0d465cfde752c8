"""Checks the sorting networks of csrc/klat_frame.hip (fr_sort16_net, fr_sort12_net, fr_sort8_net) over all 2^n zero-one inputs."""
import re, os
import numpy as np
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi-lattice-utils_amd", "csrc", "klat_frame.hip")).read()
for n in (16, 12, 8):
    body = src[src.index("void fr_sort%d_net" % n):]
    body = body[:body.index("\n}\n")]
    ces = [(int(a), int(b)) for a, b in re.findall(r"ce\(k\[(\d+)\], k\[(\d+)\]\)", body)]
    assert all(a < b < n for a, b in ces)
    x = ((np.arange(1 << n)[:, None] >> np.arange(n)) & 1).astype(np.int8)
    for a, b in ces:
        lo, hi = np.minimum(x[:, a], x[:, b]), np.maximum(x[:, a], x[:, b])
        x[:, a], x[:, b] = lo, hi
    assert (np.diff(x, axis=1) >= 0).all()
    print(n, "keys,", len(ces), "comparators: sorts every zero-one input")
