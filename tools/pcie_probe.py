"""Do host-to-device and device-to-host copies overlap on this box?  Page-locked buffers, two streams (torch is only
the allocator and the stream handle here)."""
import time, torch
n = 1 << 30
h_up = torch.empty(n, dtype=torch.uint8).pin_memory(); h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_up = torch.empty(n, dtype=torch.uint8, device="cuda"); d_dn = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(up, dn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        if up:
            with torch.cuda.stream(s1): d_up.copy_(h_up, non_blocking=True)
        if dn:
            with torch.cuda.stream(s2): h_dn.copy_(d_dn, non_blocking=True)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best
run(True, True)
a, b, c = run(True, False), run(False, True), run(True, True)
print("1 GiB up alone %.1f ms (%.1f GB/s), down alone %.1f ms (%.1f GB/s), both at once %.1f ms" % (a * 1e3, n / a / 1e9, b * 1e3, n / b / 1e9, c * 1e3))
# chunked: does splitting a copy into 8 pieces on one stream cost anything?
def chunks(k):
    torch.cuda.synchronize(); t = time.perf_counter()
    step = n // k
    with torch.cuda.stream(s1):
        for i in range(k): d_up[i * step:(i + 1) * step].copy_(h_up[i * step:(i + 1) * step], non_blocking=True)
    torch.cuda.synchronize(); return time.perf_counter() - t
chunks(8)
print("up in 8 pieces %.1f ms, in 64 pieces %.1f ms" % (chunks(8) * 1e3, chunks(64) * 1e3))
