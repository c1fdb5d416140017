// tools/emu: nothing beyond hip_runtime.h is needed by the host functional model
#pragma once
#include <hip/hip_runtime.h>
