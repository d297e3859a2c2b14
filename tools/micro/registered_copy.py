"""H2D bandwidth of host arrays page-locked IN PLACE (hipHostRegister: what the engine's pin_host_arrays does to a caller's arrays), by the
number of streams the planes are spread over, against torch's pinned allocations (hipHostMalloc).  11 planes of 28.3 MB = the forcing a resident
call uploads at the config-3 grid."""
import ctypes
import time

import numpy as np
import torch

torch.cuda.init()
rt = torch.cuda.cudart()
NPL, PLANE = 11, 4608 * 1536 * 4
dev = [torch.empty(PLANE, dtype=torch.uint8, device="cuda") for _ in range(NPL)]


def run(host, nstreams, tag, reps=5):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, (h, d) in enumerate(zip(host, dev)):
            with torch.cuda.stream(streams[i % nstreams]):
                d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%-34s %d stream(s): %6.2f ms  %5.1f GB/s" % (tag, nstreams, best * 1e3, NPL * PLANE / best / 1e9))


pinned = [torch.empty(PLANE, dtype=torch.uint8).pin_memory() for _ in range(NPL)]
raw = [np.ones(PLANE, dtype=np.uint8) for _ in range(NPL)]
reg = []
for a in raw:
    rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
    assert int(rc) == 0, rc
    reg.append(torch.from_numpy(a))
for n in (1, 2, 3, 4):
    run(pinned, n, "hipHostMalloc (torch pin_memory)")
for n in (1, 2, 3, 4):
    run(reg, n, "hipHostRegister in place")
# one big registered block, planes as slices
big = np.ones(NPL * PLANE, dtype=np.uint8)
assert int(rt.cudaHostRegister(big.ctypes.data, big.nbytes, 0)) == 0
sl = [torch.from_numpy(big[i * PLANE:(i + 1) * PLANE]) for i in range(NPL)]
for n in (1, 2, 4):
    run(sl, n, "one registered block, slices")


def run2(host_up, host_dn, tag, reps=5):
    """H2D of NPL planes and D2H of NPL planes at the same time on two streams (what the row-chunk pipeline of the host path does)."""
    su, sd = torch.cuda.Stream(), torch.cuda.Stream()
    dev2 = [torch.empty(PLANE, dtype=torch.uint8, device="cuda") for _ in range(NPL)]
    for mode in ("D2H alone", "H2D + D2H together"):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(NPL):
                if mode != "D2H alone":
                    with torch.cuda.stream(su):
                        dev[i].copy_(host_up[i], non_blocking=True)
                with torch.cuda.stream(sd):
                    host_dn[i].copy_(dev2[i], non_blocking=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        nb = NPL * PLANE * (2 if mode != "D2H alone" else 1)
        print("%-34s %-20s %6.2f ms  %5.1f GB/s (sum of both directions)" % (tag, mode, best * 1e3, nb / best / 1e9))


raw2 = [np.ones(PLANE, dtype=np.uint8) for _ in range(NPL)]
reg2 = []
for a in raw2:
    assert int(rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)) == 0
    reg2.append(torch.from_numpy(a))
run2(reg, reg2, "hipHostRegister in place")
pinned2 = [torch.empty(PLANE, dtype=torch.uint8).pin_memory() for _ in range(NPL)]
run2(pinned, pinned2, "hipHostMalloc (torch pin_memory)")
