// ste_err.h — the thread-local error string behind ste_last_error(), shared by the translation units of the UKF ABI.
#pragma once
#include <hip/hip_runtime.h>

namespace ste {
int abi_fail(int code, const char* msg);            // records msg, returns code
int abi_check_hip(hipError_t e, const char* what);  // STE_OK or STE_ELAUNCH with the HIP error text recorded
}  // namespace ste
