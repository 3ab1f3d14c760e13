// Host-side ABI plumbing of libltxk: version + thread-local error string.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ltxk.h"

static thread_local char g_err[512] = "";

void ltxk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ltxk_version(void) { return LTXK_VERSION; }
extern "C" const char* ltxk_last_error(void) { return g_err; }

// sizeof of the argument structs as compiled into the library: a binding checks its own layout against these
extern "C" int ltxk_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(ltxk_gemm_args);
    case 1: return (int)sizeof(ltxk_conv3d_args);
    case 2: return (int)sizeof(ltxk_attn_args);
  }
  return -1;
}
