#include "common.h"
#include "espresso_amd.h"
extern "C" int ea_version(void) { return 1; }
