// internal: pulls in the public C ABI next to the device helpers
#pragma once
#include "../../include/a8hip.h"

// a8_set_tuning table (runtime.hip): [0] GEMM variant, [1] ln_bwd geometry NW*100 + rows per wave,
// [2..4] attention fwd / dK,dV / dQ geometry NW*10 + row groups per wave, [5] min tiles for the register-pipelined NT
// GEMM (0 = off), [6] posconv wgrad loop splits, [7] 160-row tile experiment, [8] tall NT GEMM tiles (0 off, 1 by tile count,
// 3 / 4 always 192 / 256 rows), [9..15] spare
extern int g_a8_tuning[32];
