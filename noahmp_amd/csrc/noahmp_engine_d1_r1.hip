// Column kernels specialised for the reference's namelist options with DVEG = 1, RUN = 1 (see nmp_engine_fixed.inc).
#include <string.h>
#include "noahmp_hip.h"
#define NMP_FIXED_DVEG 1
#define NMP_FIXED_RUN 1
#define NMP_FIXED_EXPORT_PACK 1
#include "nmp_engine_fixed.inc"
