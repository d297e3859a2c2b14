! TEST INFRASTRUCTURE (oracle) -- not part of the product path.
!
! Hand-written half of the reference-side harness.  Together with the generated
! oracle/ref_harness_gen.f90 it is linked against the UNMODIFIED reference physics
! (compiled in place from /root/reference by oracle/Makefile) into
! oracle/_ref/libnoahmp_ref.so.  It only forwards C-ABI argument blocks to the
! reference's own entry points:
!   ref_read_tables -> read_mp_veg_parameters (lsm:274) + SOIL_VEG_GEN_PARM (drv:1528)
!   ref_noahmp_init -> NOAHMP_INIT (drv:847)   cold start, non-MMF
!   ref_noahmplsm   -> noahmplsm (drv:11)      [generated file]
!   ref_get_tables  -> dump of module tables   [generated file]
! The tables are read from MPTABLE.TBL / VEGPARM.TBL / SOILPARM.TBL / GENPARM.TBL in the
! current working directory (the caller chdir()s to /root/reference/run or to a copy).

! The HRLDAS driver defines this outside any module (driver/module_hrldas_noahmp_driver.F90:806-810);
! the driver layer is not built here (needs NetCDF), so the harness supplies the same one-liner.
function wrf_dm_on_monitor()
  implicit none
  logical :: wrf_dm_on_monitor
  wrf_dm_on_monitor = .true.
end function wrf_dm_on_monitor

subroutine ref_read_tables(use_modis) bind(C, name='ref_read_tables')
  use iso_c_binding
  use module_sf_noahmplsm, only : read_mp_veg_parameters
  use module_sf_noahmpdrv, only : soil_veg_gen_parm
  implicit none
  integer(c_int), value :: use_modis
  if (use_modis /= 0) then
     call read_mp_veg_parameters('MODIFIED_IGBP_MODIS_NOAH')
     call soil_veg_gen_parm('MODIFIED_IGBP_MODIS_NOAH', 'STAS')
  else
     call read_mp_veg_parameters('USGS')
     call soil_veg_gen_parm('USGS', 'STAS')
  end if
end subroutine ref_read_tables

! Cold start through the reference's NOAHMP_INIT (drv:847-1177), iopt_run /= 5 branch.
! Uses the same argument block as a step; the fields NOAHMP_INIT does not take are ignored.
! `fndsnowh` /= 0 means SNOWH is supplied (drv:997).
subroutine ref_noahmp_init(a, iswater, fndsnowh) bind(C, name='ref_noahmp_init')
  use iso_c_binding
  use noahmp_ref_abi
  use module_sf_noahmpdrv, only : noahmp_init
  implicit none
  type(noahmp_step_args), intent(in) :: a
  integer(c_int), value :: iswater, fndsnowh
  integer :: ni, nj, ns
  real(c_float), pointer, dimension(:,:) :: snow, snowh, canwat, tsk, tvxy, tgxy, canicexy, tmn, xice, &
       canliqxy, eahxy, tahxy, cmxy, chxy, fwetxy, sneqvoxy, alboldxy, qsnowxy, wslakexy, zwtxy, waxy, &
       wtxy, lfmassxy, rtmassxy, stmassxy, woodxy, stblcpxy, fastcpxy, xsaixy, t2mvxy, t2mbxy
  real(c_float), pointer, dimension(:,:,:) :: tslb, smois, sh2o, tsnoxy, zsnsoxy, snicexy, snliqxy
  integer(c_int32_t), pointer, dimension(:,:) :: isltyp, ivgtyp, isnowxy
  real(c_float), pointer :: dzs(:)
  real, allocatable :: chstarxy(:,:)
  logical :: lfnd

  ni = a%ime - a%ims + 1; nj = a%jme - a%jms + 1; ns = a%nsoil
  call c_f_pointer(a%snow, snow, [ni,nj]);       call c_f_pointer(a%snowh, snowh, [ni,nj])
  call c_f_pointer(a%canwat, canwat, [ni,nj]);   call c_f_pointer(a%tsk, tsk, [ni,nj])
  call c_f_pointer(a%tvxy, tvxy, [ni,nj]);       call c_f_pointer(a%tgxy, tgxy, [ni,nj])
  call c_f_pointer(a%canicexy, canicexy, [ni,nj]); call c_f_pointer(a%tmn, tmn, [ni,nj])
  call c_f_pointer(a%xice, xice, [ni,nj]);       call c_f_pointer(a%canliqxy, canliqxy, [ni,nj])
  call c_f_pointer(a%eahxy, eahxy, [ni,nj]);     call c_f_pointer(a%tahxy, tahxy, [ni,nj])
  call c_f_pointer(a%cmxy, cmxy, [ni,nj]);       call c_f_pointer(a%chxy, chxy, [ni,nj])
  call c_f_pointer(a%fwetxy, fwetxy, [ni,nj]);   call c_f_pointer(a%sneqvoxy, sneqvoxy, [ni,nj])
  call c_f_pointer(a%alboldxy, alboldxy, [ni,nj]); call c_f_pointer(a%qsnowxy, qsnowxy, [ni,nj])
  call c_f_pointer(a%wslakexy, wslakexy, [ni,nj]); call c_f_pointer(a%zwtxy, zwtxy, [ni,nj])
  call c_f_pointer(a%waxy, waxy, [ni,nj]);       call c_f_pointer(a%wtxy, wtxy, [ni,nj])
  call c_f_pointer(a%lfmassxy, lfmassxy, [ni,nj]); call c_f_pointer(a%rtmassxy, rtmassxy, [ni,nj])
  call c_f_pointer(a%stmassxy, stmassxy, [ni,nj]); call c_f_pointer(a%woodxy, woodxy, [ni,nj])
  call c_f_pointer(a%stblcpxy, stblcpxy, [ni,nj]); call c_f_pointer(a%fastcpxy, fastcpxy, [ni,nj])
  call c_f_pointer(a%xsaixy, xsaixy, [ni,nj]);   call c_f_pointer(a%t2mvxy, t2mvxy, [ni,nj])
  call c_f_pointer(a%t2mbxy, t2mbxy, [ni,nj])
  call c_f_pointer(a%tslb, tslb, [ni,ns,nj]);    call c_f_pointer(a%smois, smois, [ni,ns,nj])
  call c_f_pointer(a%sh2o, sh2o, [ni,ns,nj]);    call c_f_pointer(a%tsnoxy, tsnoxy, [ni,3,nj])
  call c_f_pointer(a%zsnsoxy, zsnsoxy, [ni,ns+3,nj]); call c_f_pointer(a%snicexy, snicexy, [ni,3,nj])
  call c_f_pointer(a%snliqxy, snliqxy, [ni,3,nj])
  call c_f_pointer(a%isltyp, isltyp, [ni,nj]);   call c_f_pointer(a%ivgtyp, ivgtyp, [ni,nj])
  call c_f_pointer(a%isnowxy, isnowxy, [ni,nj]); call c_f_pointer(a%dzs, dzs, [ns])
  allocate(chstarxy(ni,nj))
  lfnd = (fndsnowh /= 0)

  ! ide+1 / jde+1 mirror driver/module_hrldas_noahmp_driver.F90:291 (init loops to min(ite,ide-1), drv:991-992)
  call noahmp_init('USGS', snow, snowh, canwat, isltyp, ivgtyp, a%isurban, &
       tslb, smois, sh2o, dzs, .false., lfnd, a%isice, iswater, &
       tsk, isnowxy, tvxy, tgxy, canicexy, tmn, xice, &
       canliqxy, eahxy, tahxy, cmxy, chxy, &
       fwetxy, sneqvoxy, alboldxy, qsnowxy, wslakexy, zwtxy, waxy, &
       wtxy, tsnoxy, zsnsoxy, snicexy, snliqxy, lfmassxy, rtmassxy, &
       stmassxy, woodxy, stblcpxy, fastcpxy, xsaixy, &
       t2mvxy, t2mbxy, chstarxy, &
       ns, .false., .true., a%iopt_run, &
       a%ids, a%ide+1, a%jds, a%jde+1, a%kds, a%kde, &
       a%ims, a%ime, a%jms, a%jme, a%kms, a%kme, &
       a%its, a%ite, a%jts, a%jte, a%kts, a%kte)
  deallocate(chstarxy)
end subroutine ref_noahmp_init

! Cold start with OPT_RUN = 5: NOAHMP_INIT with its optional groundwater arguments, which makes it call the private
! GROUNDWATER_INIT / EQSMOISTURE (drv:1286-1522).  ZWTXY of the step block and WTD of the MMF block are the same array.
subroutine ref_noahmp_init_mmf(a, w, iswater, fndsnowh, dx, dy, dt) bind(C, name='ref_noahmp_init_mmf')
  use iso_c_binding
  use noahmp_ref_abi
  use module_sf_noahmpdrv, only : noahmp_init
  implicit none
  type(noahmp_step_args), intent(in) :: a
  type(noahmp_wtable_args), intent(in) :: w        ! the MMF planes (optional arguments of NOAHMP_INIT, drv:861-863)
  real(c_float), value :: dx, dy, dt
  integer(c_int), value :: iswater, fndsnowh
  integer :: ni, nj, ns
  real(c_float), pointer, dimension(:,:) :: snow, snowh, canwat, tsk, tvxy, tgxy, canicexy, tmn, xice, &
       canliqxy, eahxy, tahxy, cmxy, chxy, fwetxy, sneqvoxy, alboldxy, qsnowxy, wslakexy, zwtxy, waxy, &
       wtxy, lfmassxy, rtmassxy, stmassxy, woodxy, stblcpxy, fastcpxy, xsaixy, t2mvxy, t2mbxy
  real(c_float), pointer, dimension(:,:,:) :: tslb, smois, sh2o, tsnoxy, zsnsoxy, snicexy, snliqxy
  integer(c_int32_t), pointer, dimension(:,:) :: isltyp, ivgtyp, isnowxy
  real(c_float), pointer :: dzs(:)
  real, allocatable :: chstarxy(:,:), msftx(:,:), msfty(:,:)
  real(c_float), pointer, dimension(:,:) :: smcwtdxy, rechxy, deeprechxy, areaxy, qrfsxy, qspringsxy, qslatxy, &
       fdepthxy, ht, riverbedxy, eqzwt, rivercondxy, pexpxy
  real(c_float), pointer, dimension(:,:,:) :: smoiseq
  integer :: stepwtd
  logical :: lfnd

  ni = a%ime - a%ims + 1; nj = a%jme - a%jms + 1; ns = a%nsoil
  call c_f_pointer(a%snow, snow, [ni,nj]);       call c_f_pointer(a%snowh, snowh, [ni,nj])
  call c_f_pointer(a%canwat, canwat, [ni,nj]);   call c_f_pointer(a%tsk, tsk, [ni,nj])
  call c_f_pointer(a%tvxy, tvxy, [ni,nj]);       call c_f_pointer(a%tgxy, tgxy, [ni,nj])
  call c_f_pointer(a%canicexy, canicexy, [ni,nj]); call c_f_pointer(a%tmn, tmn, [ni,nj])
  call c_f_pointer(a%xice, xice, [ni,nj]);       call c_f_pointer(a%canliqxy, canliqxy, [ni,nj])
  call c_f_pointer(a%eahxy, eahxy, [ni,nj]);     call c_f_pointer(a%tahxy, tahxy, [ni,nj])
  call c_f_pointer(a%cmxy, cmxy, [ni,nj]);       call c_f_pointer(a%chxy, chxy, [ni,nj])
  call c_f_pointer(a%fwetxy, fwetxy, [ni,nj]);   call c_f_pointer(a%sneqvoxy, sneqvoxy, [ni,nj])
  call c_f_pointer(a%alboldxy, alboldxy, [ni,nj]); call c_f_pointer(a%qsnowxy, qsnowxy, [ni,nj])
  call c_f_pointer(a%wslakexy, wslakexy, [ni,nj]); call c_f_pointer(a%zwtxy, zwtxy, [ni,nj])
  call c_f_pointer(a%waxy, waxy, [ni,nj]);       call c_f_pointer(a%wtxy, wtxy, [ni,nj])
  call c_f_pointer(a%lfmassxy, lfmassxy, [ni,nj]); call c_f_pointer(a%rtmassxy, rtmassxy, [ni,nj])
  call c_f_pointer(a%stmassxy, stmassxy, [ni,nj]); call c_f_pointer(a%woodxy, woodxy, [ni,nj])
  call c_f_pointer(a%stblcpxy, stblcpxy, [ni,nj]); call c_f_pointer(a%fastcpxy, fastcpxy, [ni,nj])
  call c_f_pointer(a%xsaixy, xsaixy, [ni,nj]);   call c_f_pointer(a%t2mvxy, t2mvxy, [ni,nj])
  call c_f_pointer(a%t2mbxy, t2mbxy, [ni,nj])
  call c_f_pointer(a%tslb, tslb, [ni,ns,nj]);    call c_f_pointer(a%smois, smois, [ni,ns,nj])
  call c_f_pointer(a%sh2o, sh2o, [ni,ns,nj]);    call c_f_pointer(a%tsnoxy, tsnoxy, [ni,3,nj])
  call c_f_pointer(a%zsnsoxy, zsnsoxy, [ni,ns+3,nj]); call c_f_pointer(a%snicexy, snicexy, [ni,3,nj])
  call c_f_pointer(a%snliqxy, snliqxy, [ni,3,nj])
  call c_f_pointer(a%isltyp, isltyp, [ni,nj]);   call c_f_pointer(a%ivgtyp, ivgtyp, [ni,nj])
  call c_f_pointer(a%isnowxy, isnowxy, [ni,nj]); call c_f_pointer(a%dzs, dzs, [ns])
  allocate(chstarxy(ni,nj), msftx(ni,nj), msfty(ni,nj))
  msftx = 1.0; msfty = 1.0
  call c_f_pointer(w%smoiseq, smoiseq, [ni,ns,nj]); call c_f_pointer(w%smcwtd, smcwtdxy, [ni,nj])
  call c_f_pointer(w%rech, rechxy, [ni,nj]);       call c_f_pointer(w%deeprech, deeprechxy, [ni,nj])
  call c_f_pointer(w%area, areaxy, [ni,nj]);       call c_f_pointer(w%qrfs, qrfsxy, [ni,nj])
  call c_f_pointer(w%qsprings, qspringsxy, [ni,nj]); call c_f_pointer(w%qslat, qslatxy, [ni,nj])
  call c_f_pointer(w%fdepth, fdepthxy, [ni,nj]);   call c_f_pointer(w%topo, ht, [ni,nj])
  call c_f_pointer(w%riverbed, riverbedxy, [ni,nj]); call c_f_pointer(w%eqwtd, eqzwt, [ni,nj])
  call c_f_pointer(w%rivercond, rivercondxy, [ni,nj]); call c_f_pointer(w%pexp, pexpxy, [ni,nj])
  lfnd = (fndsnowh /= 0)

  ! ide+1 / jde+1 mirror driver/module_hrldas_noahmp_driver.F90:291 (init loops to min(ite,ide-1), drv:991-992)
  call noahmp_init('USGS', snow, snowh, canwat, isltyp, ivgtyp, a%isurban, &
       tslb, smois, sh2o, dzs, .false., lfnd, a%isice, iswater, &
       tsk, isnowxy, tvxy, tgxy, canicexy, tmn, xice, &
       canliqxy, eahxy, tahxy, cmxy, chxy, &
       fwetxy, sneqvoxy, alboldxy, qsnowxy, wslakexy, zwtxy, waxy, &
       wtxy, tsnoxy, zsnsoxy, snicexy, snliqxy, lfmassxy, rtmassxy, &
       stmassxy, woodxy, stblcpxy, fastcpxy, xsaixy, &
       t2mvxy, t2mbxy, chstarxy, &
       ns, .false., .true., 5, &
       a%ids, a%ide+1, a%jds, a%jde+1, a%kds, a%kde, &
       a%ims, a%ime, a%jms, a%jme, a%kms, a%kme, &
       a%its, a%ite, a%jts, a%jte, a%kts, a%kte, &
       smoiseq, smcwtdxy, rechxy, deeprechxy, areaxy, dx, dy, msftx, msfty, &
       w%wtddt, stepwtd, dt, qrfsxy, qspringsxy, qslatxy, &
       fdepthxy, ht, riverbedxy, eqzwt, rivercondxy, pexpxy)
  deallocate(chstarxy, msftx, msfty)
end subroutine ref_noahmp_init_mmf
