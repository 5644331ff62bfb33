// Stand-in for csrc/runtime.hip's error plumbing in the SANITIZER build of the product's host C++ (oracle/Makefile, target asan):
// octree_build.cpp and octree_proc.cpp are compiled on their own, without the HIP objects, and need only these two symbols of
// the runtime.  Test infrastructure (tests/test_sanitizers.py), never part of the product library.
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void gfn_set_error(const char* fmt, ...) {  // C++ linkage, as csrc/gfn_common.h declares it
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gfn_last_error(void) { return g_err; }
