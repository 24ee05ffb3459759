// cp_error.h -- thread-local last-error message + status helper shared by all translation units.
#pragma once
#include <cstdarg>
#include <cstdio>

namespace cp {

inline char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int status, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return status;
}

}  // namespace cp
