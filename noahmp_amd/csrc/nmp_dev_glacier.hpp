// Noah-MP column engine for MI355X -- glacier (land-ice) column, NOAHMP_GLACIER
// (reference phys/module_sf_noahmp_glacier.F90:150, "gla").
#pragma once
#include "nmp_dev_common.hpp"

namespace nmp {

template <class A>
NMP_DEV void glacier(const Ctx& c, Col& s, const Lay<A>& y) {
  raise(s, NOAHMP_ERR_GLACIER_ENERGY_BALANCE);   // placeholder until the glacier path lands
}

NMP_DEV void glacier_fill_undefined(Col& s) {}

}  // namespace nmp
