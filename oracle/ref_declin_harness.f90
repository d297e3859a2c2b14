! TEST INFRASTRUCTURE (oracle) -- C-callable door to the reference's CALC_DECLIN (driver/module_hrldas_noahmp_driver.F90:813-863,
! an external subroutine behind `end module`).  oracle/Makefile (`make declin`) compiles THAT subroutine, cut out of the reference
! file by line range into a scratch directory at build time, together with util/module_date_utilities.F and this file into
! oracle/_ref/libnoahmp_declin_ref.so.  Nothing of the reference is copied into the repository.
subroutine ref_calc_declin(iyear, imonth, iday, ihour, iminute, isecond, n, lat, lon, cosz, julian) bind(C, name="ref_calc_declin")
  use iso_c_binding
  implicit none
  integer(c_int), value :: iyear, imonth, iday, ihour, iminute, isecond, n
  real(c_float), intent(in) :: lat(n), lon(n)
  real(c_float), intent(out) :: cosz(n), julian
  character(len=19) :: nowdate
  integer :: i
  external :: CALC_DECLIN
  write(nowdate, '(I4.4,"-",I2.2,"-",I2.2,"_",I2.2,":",I2.2,":",I2.2)') iyear, imonth, iday, ihour, iminute, isecond
  do i = 1, n
     call CALC_DECLIN(nowdate, lat(i), lon(i), cosz(i), julian)
  end do
end subroutine ref_calc_declin
