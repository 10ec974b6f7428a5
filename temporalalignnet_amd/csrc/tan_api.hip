// Library-level entry points.
#include "tan_common.h"

extern "C" int tan_version(void) { return 100; }
