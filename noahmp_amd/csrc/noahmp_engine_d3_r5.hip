// Column kernels specialised for the reference's namelist options with DVEG = 3, RUN = 5 (see nmp_engine_fixed.inc).
#include <string.h>
#include "noahmp_hip.h"
#define NMP_FIXED_DVEG 3
#define NMP_FIXED_RUN 5
#include "nmp_engine_fixed.inc"
