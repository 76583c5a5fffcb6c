// internal view of the public C ABI (include/frhip.h is on the include path)
#pragma once
#include "frhip.h"
