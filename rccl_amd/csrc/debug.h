// Logging in the reference's style: NCCL_DEBUG=WARN|INFO|TRACE selects the level,
// NCCL_DEBUG_FILE redirects (src/debug.cc:29-105).  The most recent warning text is kept
// for ncclGetLastError.
#pragma once
#include <stdarg.h>

enum { RA_LOG_NONE = 0, RA_LOG_WARN = 1, RA_LOG_INFO = 2, RA_LOG_TRACE = 3 };
int raLogLevel();
void raLog(int level, const char* file, int line, const char* fmt, ...) __attribute__((format(printf, 4, 5)));
const char* raLastError();

#define RA_WARN(...) raLog(RA_LOG_WARN, __FILE__, __LINE__, __VA_ARGS__)
#define RA_INFO(...) do { if (raLogLevel() >= RA_LOG_INFO) raLog(RA_LOG_INFO, __FILE__, __LINE__, __VA_ARGS__); } while (0)
#define RA_TRACE(...) do { if (raLogLevel() >= RA_LOG_TRACE) raLog(RA_LOG_TRACE, __FILE__, __LINE__, __VA_ARGS__); } while (0)

// Evaluate a HIP call; on failure log and return ncclUnhandledCudaError from the caller.
#define RA_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      RA_WARN("HIP failure: %s -> %s", #call, hipGetErrorString(e_));                        \
      return ncclUnhandledCudaError;                                                         \
    }                                                                                        \
  } while (0)
#define RA_CHECK(call)                          \
  do {                                          \
    ncclResult_t r_ = (call);                   \
    if (r_ != ncclSuccess) return r_;           \
  } while (0)
