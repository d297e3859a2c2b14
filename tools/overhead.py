"""Host overhead of the synchronous entry point (wall time per step minus kernel time) at three tile sizes, and the
PCIe-inclusive rate of the host-memory path (what the Fortran shim pays).  Run on the GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noahmp_amd import synth
from noahmp_amd.driver import Engine
from noahmp_amd.tables import load_tables
import torch
T, tb = load_tables("usgs")
eng = Engine(T, device=0)
for ni, nj in ((8, 1), (1024, 64), (1024, 1024)):
    s = synth.config2(tb, ni=ni, nj=nj)
    synth.first_step_fixups(s); synth.diurnal_forcing(s, 12, t_offset=s.t_offset)
    d = s.to_device("cuda:0")
    for it in range(3): eng.noahmplsm(d, it + 1, 2000, 180.0)
    torch.cuda.synchronize()
    t = time.perf_counter(); km = 0
    n = 50
    for it in range(n):
        st = eng.noahmplsm(d, it + 4, 2000, 180.0); km += st.kernel_ms
    torch.cuda.synchronize()
    w = (time.perf_counter() - t) / n * 1e3
    print("tile %dx%d: wall %.3f ms/step, kernel %.3f ms, overhead %.3f ms" % (ni, nj, w, km / n, w - km / n))

# host-memory path (what the Fortran shim pays): single-shot staging, then the row-chunk pipeline, pinned, OUT mirror trusted
def host_path(label, **opts):
    s = synth.config2(tb, ni=1024, nj=1024)
    synth.first_step_fixups(s); synth.diurnal_forcing(s, 12, t_offset=s.t_offset)
    prev = {k: eng.set_option(k, v) for k, v in opts.items()}
    for it in range(3): eng.noahmplsm(s, it + 1, 2000, 180.0)
    t = time.perf_counter(); n = 12
    for it in range(n): st = eng.noahmplsm(s, it + 4, 2000, 180.0)
    w = (time.perf_counter() - t) / n
    if opts.get("lazy_download"):
        t1 = time.perf_counter(); eng.fetch(); print("    fetch of INOUT+OUT arrays: %.1f ms" % ((time.perf_counter() - t1) * 1e3))
    for k in ("deferred_status", "static_inputs", "lazy_download", "resident_state"):
        if k in prev: eng.set_option(k, prev.pop(k))
    for k, v in prev.items(): eng.set_option(k, v)
    print("host-memory path 1024x1024 %-44s %.1f ms/step (kernel %.2f ms) -> %.3e col-steps/s PCIe-inclusive" % (label, w * 1e3, st.kernel_ms, s.ncol / w))
host_path("single shot, pageable:", host_chunks=0)
host_path("single shot, pinned:", host_chunks=0, pin_host_arrays=1)
for nc in (2, 3, 4, 6):
    host_path("%d row chunks, pinned:" % nc, host_chunks=nc, pin_host_arrays=1)
for nc in (2, 3, 4):
    host_path("%d row chunks, pinned, OUT mirror trusted:" % nc, host_chunks=nc, pin_host_arrays=1, trust_out_mirror=1)
host_path("resident state, downloads every call, pageable:", resident_state=1)
host_path("resident state, downloads every call, pinned:", resident_state=1, pin_host_arrays=1)
host_path("resident state, lazy download, pageable:", resident_state=1, lazy_download=1)
host_path("resident state, lazy download, pinned:", resident_state=1, lazy_download=1, pin_host_arrays=1)
host_path("... + static inputs:", resident_state=1, lazy_download=1, pin_host_arrays=1, static_inputs=1)
host_path("... + static inputs + deferred status:", resident_state=1, lazy_download=1, pin_host_arrays=1, static_inputs=1, deferred_status=1)
sys.exit(0)
# (old single measurement): H2D of all arrays + kernel + D2H, 1M columns
s = synth.config2(tb, ni=1024, nj=1024)
synth.first_step_fixups(s); synth.diurnal_forcing(s, 12, t_offset=s.t_offset)
for it in range(2): eng.noahmplsm(s, it + 1, 2000, 180.0)
t = time.perf_counter()
n = 5
for it in range(n):
    st = eng.noahmplsm(s, it + 3, 2000, 180.0)
w = (time.perf_counter() - t) / n
print("host-memory path 1024x1024: %.1f ms/step (kernel %.2f ms) -> %.3e col-steps/s PCIe-inclusive" % (w * 1e3, st.kernel_ms, s.ncol / w))
