// Error plumbing shared by all translation units of librubiks_hip.so.
#pragma once
#include <hip/hip_runtime.h>

namespace rk {

// Records a thread-local message (returned by rk_last_error()) and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace rk

#define RK_HIP(call)                                                                                              \
	do {                                                                                                          \
		hipError_t rk_e_ = (call);                                                                                \
		if (rk_e_ != hipSuccess)                                                                                  \
			return ::rk::fail(-2 /* RK_EHIP */, "%s failed: %s (%s:%d)", #call, hipGetErrorString(rk_e_), __FILE__, __LINE__); \
	} while (0)
