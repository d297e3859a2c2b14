"""TEST INFRASTRUCTURE: compile + link the generated drop-in Fortran shim with flang.

  libnoahmp_shim.so = noahmp_amd/fortran/module_sf_noahmpdrv_hip.F90 (the product's Fortran side)
                    + tests/fortran/shim_wrap_gen.f90 (bind(C) trampoline for the tests)
                    + tests/fortran/dev_driver.f90 (a miniature device-resident time loop in Fortran)
linked against libnoahmp_hip.so (the engine) and oracle/_ref/libnoahmp_ref.so, which plays the role of
the rest of HRLDAS here: it provides the reference's table modules (module_sf_noahmplsm,
noahmp_rad_parameters) that the shim `use`s, plus wrf_error_fatal / wrf_message."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FC = "/opt/rocm/lib/llvm/bin/flang"
REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libnoahmp_shim.so")


def available():
    return os.path.exists(FC) and os.path.exists(os.path.join(REF, "mod_O0", "module_sf_noahmplsm.mod")) \
        and os.path.exists(os.path.join(REF, "libnoahmp_ref.so"))


def build():
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, "noahmp_amd", "csrc")
    srcs = [os.path.join(ROOT, "noahmp_amd", "fortran", "module_sf_noahmpdrv_hip.F90"),
            os.path.join(HERE, "shim_wrap_gen.f90"), os.path.join(HERE, "dev_driver.f90")]
    if os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in srcs):
        return LIB
    cmd = [FC, "-cpp", "-fPIC", "-shared", "-O1", "-I" + os.path.join(REF, "mod_O0"), "-module-dir", OUT] + srcs + \
          ["-o", LIB, "-L" + REF, "-lnoahmp_ref", "-L" + csrc, "-lnoahmp_hip",
           "-Wl,-rpath," + REF, "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib/llvm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return LIB
