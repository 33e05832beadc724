// ABI bookkeeping for libape_hip.so.
#include "common.h"
#include <string.h>

namespace ape {
static thread_local char g_err[256] = "";
void set_last_error(const char* what)
{
    strncpy(g_err, what ? what : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
}  // namespace ape

extern "C" int ape_abi_version(void) { return 1; }
extern "C" const char* ape_last_error(void) { return ape::g_err; }
