"""Bandwidth of pageable host <-> device copies by size (torch .cuda() / .cpu()): where does the HIP runtime switch from its staging buffers to
page-locking the buffer in place (GPU_PINNED_MIN_XFER_SIZE)?  Run once as is and once with GPU_PINNED_MIN_XFER_SIZE=1048576."""
import os
import time

import torch

print("GPU_PINNED_MIN_XFER_SIZE =", os.environ.get("GPU_PINNED_MIN_XFER_SIZE", "(unset)"))
torch.cuda.init()
for mb in (1, 2, 4, 8, 16, 32, 64, 96, 127, 129, 160, 256, 512):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8)
    h.fill_(1)
    d = h.cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        d = h.cuda()
    torch.cuda.synchronize()
    up = 3 * n / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    for _ in range(3):
        b = d.cpu()
    down = 3 * n / (time.perf_counter() - t0) / 1e9
    print("%4d MiB  H2D %6.1f GB/s   D2H %6.1f GB/s" % (mb, up, down))
