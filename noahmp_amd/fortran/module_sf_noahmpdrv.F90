! Zero-edit drop-in: modules with the REFERENCE'S OWN NAMES, so that driver/module_hrldas_noahmp_driver.F90 compiles unchanged --
!     use module_sf_noahmp_groundwater                                         (hdrv:5)
!     use module_sf_noahmpdrv, only: noahmp_init, noahmplsm, soil_veg_gen_parm (hdrv:6)
! resolve to the modules below.  `noahmplsm` and `WTABLE_mmf_noahmp` come from the generated HIP shims
! (module_sf_noahmpdrv_hip.F90: the reference's dummy lists, ISO_C_BINDING calls into libnoahmp_hip.so); NOAHMP_INIT, the table
! readers and everything else the reference modules export come from the reference's own two files, compiled unchanged but under
! other module names by the preprocessor the reference build already runs (-cpp):
!     flang -cpp ... -Dmodule_sf_noahmp_groundwater=module_sf_noahmp_groundwater_ref   phys/module_sf_noahmp_groundwater.F90
!     flang -cpp ... -Dmodule_sf_noahmp_groundwater=module_sf_noahmp_groundwater_ref \
!                    -Dmodule_sf_noahmpdrv=module_sf_noahmpdrv_ref                     phys/module_sf_noahmpdrv.F90
! (two -D flags in phys/Makefile; no source file of the reference is edited).  INTEGRATION.md section 1b;
! tests/test_fortran_shim.py::test_same_name_modules_compile_the_reference_use_lines builds exactly this.
module module_sf_noahmp_groundwater
  use module_sf_noahmp_groundwater_ref, only : LATERALFLOW, UPDATEWTD           ! (only NOAHMP_INIT's GROUNDWATER_INIT uses LATERALFLOW, drv:1297)
  use module_sf_noahmp_groundwater_hip, only : WTABLE_mmf_noahmp                ! gw:14-22, on the MI355X
  implicit none
  public
end module module_sf_noahmp_groundwater

module module_sf_noahmpdrv
  use module_sf_noahmpdrv_ref, only : noahmp_init, soil_veg_gen_parm, snow_init, groundwater_init   ! drv:847 ..., unchanged reference code
  use module_sf_noahmpdrv_hip, only : noahmplsm, noahmp_hip_fetch_state, noahmp_hip_upload_tables   ! drv:11-844, on the MI355X
  implicit none
  public
end module module_sf_noahmpdrv
