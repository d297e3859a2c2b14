/* TEST INFRASTRUCTURE (oracle): glacier column (NOAHMP_GLACIER, phys/module_sf_noahmp_glacier.F90). */
#include <math.h>
#include "nmp_internal.h"

void nmp_glacier_column(nmp_ctx* c, nmp_column* s, real* fsr_out) {
  (void)s; (void)fsr_out;
  if (!c->err) c->err = NOAHMP_ERR_GLACIER_ENERGY_BALANCE;   /* placeholder until restated */
}
