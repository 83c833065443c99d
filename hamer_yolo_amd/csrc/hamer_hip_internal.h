// Internal helpers shared by the translation units of libhamer_hip.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hamer_hip.h"

int hm_set_error(int code, const char* msg);
int hm_check_launch(const char* what);   // hipGetLastError() -> status
