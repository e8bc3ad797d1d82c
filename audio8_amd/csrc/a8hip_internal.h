// internal: pulls in the public C ABI next to the device helpers
#pragma once
#include "../../include/a8hip.h"

// a8_set_tuning table (runtime.hip): [1] ln_bwd geometry NW*100 + rows per wave, [2..4] attention fwd / dK,dV / dQ launch geometry,
// [5] min 256 x 256 tiles for the pipelined NT GEMM (0 = off), [6] posconv wgrad loop splits, [9] conv layer 0 on the VALU,
// [14] min M for the asymmetric-ring NT GEMM; the rest spare
extern int g_a8_tuning[32];
