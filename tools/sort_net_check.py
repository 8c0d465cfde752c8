"""Checks the 16-input sorting network of csrc/klat_frame.hip (fr_sort16_net) over all 2^16 zero-one inputs."""
import re, os
import numpy as np
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi-lattice-utils_amd", "csrc", "klat_frame.hip")).read()
body = src[src.index("void fr_sort16_net"):]
body = body[:body.index("\n}\n")]
ces = [(int(a), int(b)) for a, b in re.findall(r"ce\(k\[(\d+)\], k\[(\d+)\]\)", body)]
x = ((np.arange(1 << 16)[:, None] >> np.arange(16)) & 1).astype(np.int8)
for a, b in ces:
    lo, hi = np.minimum(x[:, a], x[:, b]), np.maximum(x[:, a], x[:, b])
    x[:, a], x[:, b] = lo, hi
assert (np.diff(x, axis=1) >= 0).all()
print(len(ces), "comparators: sorts every zero-one input")
