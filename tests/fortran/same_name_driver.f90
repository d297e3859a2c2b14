! TEST INFRASTRUCTURE -- a miniature of driver/module_hrldas_noahmp_driver.F90's dependency on the physics: the two `use` lines are
! the reference's own (hdrv:5-6), character for character; the body only takes the addresses of what the driver calls, so that the
! link proves every name resolves (noahmplsm and WTABLE_mmf_noahmp to the HIP shims, the rest to the reference's code).
module same_name_driver
  use module_sf_noahmp_groundwater
  use module_sf_noahmpdrv, only: noahmp_init, noahmplsm, soil_veg_gen_parm
  implicit none
contains
  subroutine same_name_probe(n) bind(C, name="same_name_probe")
    use iso_c_binding
    integer(c_int), intent(out) :: n
    procedure(noahmplsm), pointer :: p1
    procedure(noahmp_init), pointer :: p2
    procedure(soil_veg_gen_parm), pointer :: p3
    procedure(WTABLE_mmf_noahmp), pointer :: p4
    n = 0
    p1 => noahmplsm;         if (associated(p1)) n = n + 1
    p2 => noahmp_init;       if (associated(p2)) n = n + 1
    p3 => soil_veg_gen_parm; if (associated(p3)) n = n + 1
    p4 => WTABLE_mmf_noahmp; if (associated(p4)) n = n + 1
  end subroutine same_name_probe
end module same_name_driver
