"""Pageable host arrays through noahmp_hip_step(NOAHMP_MEM_HOST) at config-3 size (7 M columns, single shot: every array up, INOUT + OUT
back): the engine's own page-locked bounce buffers (nmp_stage.hpp, round 6) by the number of copy threads.  Also a C-host-like process:
GPU_PINNED_MIN_XFER_SIZE is NOT set here and torch is not imported -- the engine is the process's only HIP user.
usage: stage_exp.py [threads ...]   (each count in a process of its own: NMP_COPY_THREADS is read when the pool starts)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import ctypes as C
    from noahmp_amd import synth
    from noahmp_amd.driver import Engine
    from noahmp_amd.state import ModelConfig
    from noahmp_amd.tables import load_tables
    T, tb = load_tables("usgs")
    eng = Engine(T, device=0)
    ni, nj = (int(os.environ.get("NMP_STAGE_NI", "4608")), int(os.environ.get("NMP_STAGE_NJ", "1536")))
    s = synth.config3_tile(tb, ni, nj, cfg=ModelConfig(idveg=3))
    synth.first_step_fixups(s)
    eng.set_option("host_chunks", 0)
    if os.environ.get("NMP_STAGE_CHURN"):
        # fault hunt for a host whose arrays come and go (the pattern that killed round 5's runs when the runtime page-locked pageable
        # buffers in place): every call on a fresh copy of the store, the previous copy freed -- heap addresses are reused across calls;
        # also the resident path (pageable uploads + fetch) and the groundwater / cold-start host paths
        import numpy as np
        from noahmp_amd import init as _init  # noqa: F401
        n = int(os.environ["NMP_STAGE_CHURN"])
        ref = None
        for i in range(n):
            c = s.copy()
            st = eng.noahmplsm(c, 1, 2000, 180.0)
            assert st.code == 0
            if i % 5 == 4:
                eng.set_option("resident_state", 1); eng.set_option("lazy_download", 1)
                c2 = s.copy()
                eng.noahmplsm(c2, 1, 2000, 180.0)
                eng.fetch()
                eng.set_option("lazy_download", 0); eng.set_option("resident_state", 0)
                assert np.array_equal(c2.a["tslb"], c.a["tslb"])
            if ref is None:
                ref = c.a["tslb"].copy()
            assert np.array_equal(ref, c.a["tslb"])
            del c
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        eng.lib.noahmp_hip_debug_copy_stats(C.byref(a), C.byref(b))
        print("churn: %d calls on fresh arrays ok; staged %.2f GB, direct %.2f GB; GPU_PINNED_MIN_XFER_SIZE=%s"
              % (n, a.value / 1e9, b.value / 1e9, os.environ.get("GPU_PINNED_MIN_XFER_SIZE")), flush=True)
        return
    it = 1
    eng.noahmplsm(s, it, 2000, 180.0); it += 1
    t0 = time.perf_counter()
    km, n = 0.0, 3
    for _ in range(n):
        km += eng.noahmplsm(s, it, 2000, 180.0).kernel_ms; it += 1
    dt = (time.perf_counter() - t0) / n
    a, b = C.c_ulonglong(0), C.c_ulonglong(0)
    eng.lib.noahmp_hip_debug_copy_stats(C.byref(a), C.byref(b))
    nbytes = sum(v.nbytes for k, v in s.a.items() if k != "dzs")
    print("copy threads %s: %.1f ms per call = %.3g column-steps/s, kernel %.2f ms; staged %.2f GB, direct %.2f GB over %d calls (arrays: %.2f GB); "
          "GPU_PINNED_MIN_XFER_SIZE=%s torch imported: %s"
          % (os.environ.get("NMP_COPY_THREADS", "default(8)"), dt * 1e3, s.ncol / dt, km / n, a.value / 1e9, b.value / 1e9, n + 1, nbytes / 1e9,
             os.environ.get("GPU_PINNED_MIN_XFER_SIZE"), "torch" in sys.modules), flush=True)


if __name__ == "__main__":
    if os.environ.get("NMP_STAGE_CHILD"):
        one()
    else:
        for th in sys.argv[1:] or ["1", "4", "8", "16"]:
            env = dict(os.environ, NMP_STAGE_CHILD="1", NMP_COPY_THREADS=th)
            env.pop("GPU_PINNED_MIN_XFER_SIZE", None)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
