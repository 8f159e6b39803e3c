// errors.cpp -- ABI version and the thread-local error string (host only, no HIP calls: part of the sanitizer build, `make asan`).
#include <cstdio>

#include "adt_common.h"

namespace adt {

static thread_local char g_err[512] = "";

char* error_buffer(size_t* size) {
  *size = sizeof(g_err);
  return g_err;
}

int set_error(int code, const char* msg) {
  std::snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

}  // namespace adt

extern "C" int adt_version(void) { return 19; }
extern "C" const char* adt_last_error(void) { return adt::g_err; }
