// Thread-local error text for the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/vitssl_hip.h"

static thread_local char g_err[512] = "";

void vitssl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vitssl_last_error(void) { return g_err; }
extern "C" int vitssl_version(void) { return 2; }   // 2: vitssl_dino_loss takes the size of its scratch
