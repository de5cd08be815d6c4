// Internal helpers shared by the translation units of libsd_frontend.so (sd_api.hip: front end + tracker; sd_yolo_api.hip: detector):
// the thread-local error text behind sd_last_error() and the two return-on-error macros.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "sd_frontend.h"

int sd_set_err(int code, const std::string& msg);        // defined in sd_api.hip
#define set_err sd_set_err

#define HIPCHK(call)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return set_err(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? SD_ERR_NO_DEVICE : SD_ERR_HIP, \
                           std::string(#call) + ": " + hipGetErrorString(e_));                            \
    } while (0)

static inline int sd_check_launch(const char* name)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(SD_ERR_HIP, std::string(name) + " launch: " + hipGetErrorString(e));
    return SD_OK;
}
#define LAUNCH_CHECK(name) do { int rc_ = sd_check_launch(name); if (rc_ != SD_OK) return rc_; } while (0)
