"""dev helper: what the box's PCIe link gives pinned hipMemcpyAsync, one direction and both at once (the end-to-end pipeline's ceiling)."""
import time, torch
n = 1 << 28  # 256 MiB
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_in = torch.empty(n, dtype=torch.uint8, device="cuda")
d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(h2d, d2h, reps=8):
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps):
        if h2d:
            with torch.cuda.stream(s1):
                d_in.copy_(h_in, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s2):
                h_out.copy_(d_out, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.time() - t
    return reps * n / dt / 1e9
for _ in range(2):
    print("H2D alone %.1f GB/s   D2H alone %.1f GB/s   both at once %.1f + %.1f GB/s" % (run(1, 0), run(0, 1), run(1, 1), run(1, 1)))
# the pipeline's copy sizes: 39 MB up, 46 + 23 MB down per chunk, several streams at once
def sized(nst, reps=40):
    up = [torch.empty(39 << 20, dtype=torch.uint8).pin_memory() for _ in range(nst)]
    dn = [torch.empty(69 << 20, dtype=torch.uint8).pin_memory() for _ in range(nst)]
    dup = [torch.empty(39 << 20, dtype=torch.uint8, device="cuda") for _ in range(nst)]
    ddn = [torch.empty(69 << 20, dtype=torch.uint8, device="cuda") for _ in range(nst)]
    ss = [torch.cuda.Stream() for _ in range(nst)]
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps):
        for i in range(nst):
            with torch.cuda.stream(ss[i]):
                dup[i].copy_(up[i], non_blocking=True)
                dn[i][: 46 << 20].copy_(ddn[i][: 46 << 20], non_blocking=True)
                dn[i][46 << 20:].copy_(ddn[i][46 << 20:], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.time() - t
    print("%d streams, 39 MB up + 46 + 23 MB down per turn: %.1f GB/s up + %.1f GB/s down" % (nst, reps * nst * 39 * 2**20 / dt / 1e9, reps * nst * 69 * 2**20 / dt / 1e9))
for nst in (1, 3, 5):
    sized(nst)
# the same sizes with the two directions on separate streams (n up-streams + n down-streams)
def sized_split(nst, reps=40):
    up = [torch.empty(39 << 20, dtype=torch.uint8).pin_memory() for _ in range(nst)]
    dn = [torch.empty(69 << 20, dtype=torch.uint8).pin_memory() for _ in range(nst)]
    dup = [torch.empty(39 << 20, dtype=torch.uint8, device="cuda") for _ in range(nst)]
    ddn = [torch.empty(69 << 20, dtype=torch.uint8, device="cuda") for _ in range(nst)]
    sa = [torch.cuda.Stream() for _ in range(nst)]
    sb = [torch.cuda.Stream() for _ in range(nst)]
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps):
        for i in range(nst):
            with torch.cuda.stream(sa[i]):
                dup[i].copy_(up[i], non_blocking=True)
            with torch.cuda.stream(sb[i]):
                dn[i][: 46 << 20].copy_(ddn[i][: 46 << 20], non_blocking=True)
                dn[i][46 << 20:].copy_(ddn[i][46 << 20:], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.time() - t
    print("%d + %d streams, directions apart: %.1f GB/s up + %.1f GB/s down" % (nst, nst, reps * nst * 39 * 2**20 / dt / 1e9, reps * nst * 69 * 2**20 / dt / 1e9))
for nst in (1, 3, 5):
    sized_split(nst)
