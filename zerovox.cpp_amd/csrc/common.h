// common.h — error convention of the boundary.
//
// The reference throws std::runtime_error on load/alloc/compute failures, exit(1)s on a bad KV
// (die_fmt, src/zerovox.h:435-455) and abort()s inside ggml asserts (bad ids etc.).  Here every
// failure is a zv::Error carrying a zv_status; the C-ABI turns it into a status code +
// zv_last_error(), the C++ facade lets it propagate (zv::Error derives from std::runtime_error).
#pragma once

#include <cstdarg>
#include <cstdio>
#include <stdexcept>
#include <string>

#include <hip/hip_runtime.h>

#include "../../include/zerovox_amd.h"

namespace zv
{

class Error : public std::runtime_error
{
  public:
    Error(zv_status st, const std::string &msg) : std::runtime_error(msg), status(st) {}
    zv_status status;
};

[[noreturn]] inline void fail(zv_status st, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(st, buf);
}

#define ZV_HIP(expr)                                                                                        \
    do                                                                                                      \
    {                                                                                                       \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            ::zv::fail(ZV_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

}  // namespace zv
