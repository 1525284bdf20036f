// ff_api.hip -- library-wide pieces of the C ABI (version, error string).
#include "ff_common.h"
#include <string.h>

static thread_local char g_ff_err[256] = "";

void ff_set_error(const char* msg) {
  strncpy(g_ff_err, msg ? msg : "", sizeof(g_ff_err) - 1);
  g_ff_err[sizeof(g_ff_err) - 1] = 0;
}

extern "C" {
int ff_version(void) { return 100; }
const char* ff_last_error(void) { return g_ff_err; }
}
