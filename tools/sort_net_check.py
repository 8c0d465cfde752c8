"""Checks the sorting networks of csrc/klat_frame.hip (fr_sort8_net ... fr_sort32_net) by the zero-one principle: all 2^n
inputs up to 24 keys (bit-parallel), a sample of them plus every input of at most two ones or zeros for 32."""
import re, os
import numpy as np
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaldi-lattice-utils_amd", "csrc", "klat_frame.hip")).read()


def network(n):
    body = src[src.index("void fr_sort%d_net" % n):]
    body = body[:body.index("\n}\n")]
    ces = [(int(a), int(b)) for a, b in re.findall(r"ce\(k\[(\d+)\], k\[(\d+)\]\)", body)]
    assert all(a < b < n for a, b in ces)
    return ces


def sorts_all(ces, n):
    idx = np.arange(1 << n, dtype=np.uint64)
    lines = [np.packbits(((idx >> np.uint64(b)) & np.uint64(1)).astype(np.uint8), bitorder="little") for b in range(n)]
    for a, b in ces:
        lines[a], lines[b] = lines[a] & lines[b], lines[a] | lines[b]
    return not any((lines[b] & ~lines[b + 1]).any() for b in range(n - 1))


def sorts_these(ces, x):
    for a, b in ces:
        lo, hi = np.minimum(x[:, a], x[:, b]), np.maximum(x[:, a], x[:, b])
        x[:, a], x[:, b] = lo, hi
    return bool((np.diff(x, axis=1) >= 0).all())


if __name__ == "__main__":
    for n in (8, 12, 16, 24):
        ces = network(n)
        assert sorts_all(ces, n)
        print(n, "keys,", len(ces), "comparators: sorts every zero-one input")
    ces = network(32)
    rng = np.random.RandomState(1)
    few = np.array([[1 if i in (a, b) else 0 for i in range(32)] for a in range(32) for b in range(32)], np.int8)
    assert sorts_these(ces, rng.randint(0, 2, (2000000, 32)).astype(np.int8)) and sorts_these(ces, few.copy()) and sorts_these(ces, 1 - few)
    print(32, "keys,", len(ces), "comparators: sorts two million random zero-one inputs and all of at most two ones or zeros")
