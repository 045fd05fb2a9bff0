// uc_common.h — error plumbing shared by the host side of libunicore_cluster.so.
// Error model mirrors the reference: the only signal that crosses the boundary is a status code
// (/root/reference/src/util/command.rs:10-14); the message is kept thread-local for uc_last_error().
#pragma once
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "unicore_cluster.h"

namespace uc {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char *fmt, ...) {
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

void set_last_error(const std::string &m);

// verbosity-gated logging on the Foldseek 0..3 scale (src/modules/cluster.rs:18 maps Unicore's 0..4 onto it)
extern int g_verbosity;
inline void logf(int level, const char *fmt, ...) {
    if (level > g_verbosity) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(level <= 1 ? stderr : stdout, fmt, ap);
    va_end(ap);
    fflush(level <= 1 ? stderr : stdout);
}

struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
inline const Timer g_library_loaded;      // started when the shared library's initialisers run: UC_TIMING stamps are relative to it

}  // namespace uc

#define UC_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess) ::uc::fail(UC_ERR_DEVICE, "HIP error %s at %s:%d: %s", hipGetErrorName(_e), \
                                         __FILE__, __LINE__, hipGetErrorString(_e));                   \
    } while (0)
