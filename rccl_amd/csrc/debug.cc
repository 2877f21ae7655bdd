#include "debug.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>

#include <mutex>

static std::mutex gLogMutex;
static char gLastError[1024] = "";
static FILE* gLogFile = nullptr;

int raLogLevel() {
  static int level = -1;
  if (level < 0) {
    const char* e = getenv("NCCL_DEBUG");
    int l = RA_LOG_NONE;
    if (e) {
      if (!strcasecmp(e, "WARN")) l = RA_LOG_WARN;
      else if (!strcasecmp(e, "INFO")) l = RA_LOG_INFO;
      else if (!strcasecmp(e, "TRACE")) l = RA_LOG_TRACE;
      else if (!strcasecmp(e, "VERSION")) l = RA_LOG_WARN;
    }
    const char* f = getenv("NCCL_DEBUG_FILE");
    gLogFile = (f && *f) ? fopen(f, "a") : nullptr;
    if (!gLogFile) gLogFile = stderr;
    level = l;
  }
  return level;
}

void raLog(int level, const char* file, int line, const char* fmt, ...) {
  char msg[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof msg, fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> g(gLogMutex);
  if (level == RA_LOG_WARN) { strncpy(gLastError, msg, sizeof gLastError - 1); gLastError[sizeof gLastError - 1] = 0; }
  if (raLogLevel() >= level) {
    const char* base = strrchr(file, '/');
    fprintf(gLogFile, "rccl_amd [%d] %s %s:%d %s\n", (int)getpid(), level == RA_LOG_WARN ? "WARN" : "INFO",
            base ? base + 1 : file, line, msg);
    fflush(gLogFile);
  }
}

const char* raLastError() { return gLastError; }
