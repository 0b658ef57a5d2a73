// build_info.cpp -- identity of this build of libxdtts_hip.so.  The Makefile recompiles this file whenever any source of
// the library changes and passes XDTTS_SRC_HASH = sha256 over the sources (sorted by name: csrc/* and include/xdtts.h,
// this file excluded from nothing -- it never changes), so a test can tell whether the .so that was loaded was built from
// the sources next to it (tests/test_abi_cpu.py) -- the .so files are git-ignored and travel prebuilt.
#include "../../include/xdtts.h"

#ifndef XDTTS_SRC_HASH
#define XDTTS_SRC_HASH "unknown"
#endif
#ifndef XDTTS_BUILD_ARCH
#define XDTTS_BUILD_ARCH "unknown"
#endif
#ifndef XDTTS_BUILD_UTC
#define XDTTS_BUILD_UTC "unknown"
#endif

extern "C" const char *xdtts_build_info(void) {
  return "src_sha256=" XDTTS_SRC_HASH " arch=" XDTTS_BUILD_ARCH " built_utc=" XDTTS_BUILD_UTC
#ifdef __clang_version__
         " compiler=clang-" __clang_version__
#endif
      ;
}
