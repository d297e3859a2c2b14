! TEST INFRASTRUCTURE: a miniature device-resident HRLDAS time loop written in Fortran against the generated interfaces
! (modules noahmp_hip_abi / noahmp_hip_device): upload once, then per step forcing preparation (CALC_DECLIN, level
! copies, first-step guesses: hdrv:336-384) and the column step, both only enqueued; one synchronisation and one
! download at the end -- the structure INTEGRATION.md section 4 describes for land_driver_exe (hdrv:309-595).
function dev_driver_run(a, lon2d, rain_rate, nsteps, iday0, zlvl, julian_last) bind(C, name='dev_driver_run') result(rc)
  use iso_c_binding
  use noahmp_hip_abi
  use noahmp_hip_device
  implicit none
  type(noahmp_step_args), intent(in) :: a            ! host arrays
  type(c_ptr), value :: lon2d, rain_rate             ! host planes (ims:ime, jms:jme)
  integer(c_int), value :: nsteps, iday0
  real(c_float), value :: zlvl
  real(c_float), intent(out) :: julian_last
  integer(c_int) :: rc
  type(noahmp_step_args) :: d
  type(noahmp_status) :: st
  type(c_ptr) :: dlon, drain
  integer(c_size_t) :: nb
  integer(c_int) :: n, flags, bad_step
  real(c_float) :: jul

  call noahmp_hip_block_to_device(a, d, rc)
  if (rc /= 0) return
  nb = 4_c_size_t * int(a%ime - a%ims + 1, c_size_t) * int(a%jme - a%jms + 1, c_size_t)
  dlon = noahmp_hip_malloc(nb); drain = noahmp_hip_malloc(nb)
  rc = noahmp_hip_memcpy(dlon, lon2d, nb, 0_c_int)
  if (rc == 0) rc = noahmp_hip_memcpy(drain, rain_rate, nb, 0_c_int)
  do n = 0, nsteps - 1
     if (rc /= 0) exit
     flags = 0
     if (n == 0) flags = NOAHMP_PREP_FIRST_STEP
     rc = noahmp_hip_forcing_prep(d, dlon, drain, iday0 + n / 24, mod(n, 24), 0_c_int, 0_c_int, zlvl, flags, jul, &
                                  1_c_int, c_null_ptr, c_null_ptr)
     if (rc /= 0) exit
     d%itimestep = n + 1
     d%julian = jul
     rc = noahmp_hip_step_async(d, c_null_ptr)
  end do
  julian_last = jul
  if (rc == 0) rc = noahmp_hip_sync(st, bad_step)
  if (rc == 0) call noahmp_hip_block_from_device(d, a, rc)
  call noahmp_hip_block_free(d)
  call noahmp_hip_free(dlon); call noahmp_hip_free(drain)
end function dev_driver_run
