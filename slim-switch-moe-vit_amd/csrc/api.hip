// C-ABI bookkeeping: version + thread-local error message.
#include "smoe_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void smoe_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int smoe_abi_version(void) { return 10; }
extern "C" const char* smoe_last_error(void) { return g_err; }
