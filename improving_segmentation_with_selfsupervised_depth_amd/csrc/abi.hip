#include "segsde_common.h"
extern "C" int segsde_abi_version(void) { return SEGSDE_ABI_VERSION; }
