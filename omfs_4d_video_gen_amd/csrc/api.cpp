// Error reporting and ABI version for libomfs_splat.so.
#include <cstdarg>
#include <cstdio>
#include "../../include/omfs_splat.h"

namespace omfs {
static thread_local char g_last_error[512] = "";
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace omfs

extern "C" int omfs_abi_version(void) { return OMFS_ABI_VERSION; }
extern "C" const char* omfs_last_error(void) { return omfs::g_last_error; }
