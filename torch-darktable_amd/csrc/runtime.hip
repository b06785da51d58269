// runtime.hip -- ABI version and per-thread error message of libtdk_hip.so.
#include "tdk_common.h"

static thread_local char g_last_error[512] = "";

void tdk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

TDK_EXPORT int tdk_abi_version(void) { return TDK_ABI_VERSION; }
TDK_EXPORT const char* tdk_last_error(void) { return g_last_error; }
