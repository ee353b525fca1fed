#include "common.h"
extern "C" int swin_hip_abi_version(void) { return 1; }
