// Column kernels specialised for the reference's namelist options with DVEG = 4, RUN = 3 (see nmp_engine_fixed.inc).
#include <string.h>
#include "noahmp_hip.h"
#define NMP_FIXED_DVEG 4
#define NMP_FIXED_RUN 3
#include "nmp_engine_fixed.inc"
