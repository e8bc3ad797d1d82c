// internal: pulls in the public C ABI next to the device helpers
#pragma once
#include "../../include/a8hip.h"
