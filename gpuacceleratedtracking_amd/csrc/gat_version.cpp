// gat_version.cpp -- build identity of libgat.so.  Compiled at every link (gpuacceleratedtracking_amd/build.py) with
//   -DGAT_GIT_SHA="<short commit of the last change under csrc/ + include/>[+dirty]"
//   -DGAT_BUILD_FLAGS="<the -D flags of the build beyond the product recipe | none>"
// so that a benchmark record can name the kernels it timed (the reference tags every saved result with the git
// commit: @tagsave, scripts/run_benchmarks_gpsl1.jl:24-27).  A product build reports "flags:none"; development
// builds (-DGAT_DEV: environment knobs, diagnostic kernels that compute wrong results on purpose) name theirs.
#include "gat.h"

#ifndef GAT_GIT_SHA
#define GAT_GIT_SHA "unknown"
#endif
#ifndef GAT_BUILD_FLAGS
#define GAT_BUILD_FLAGS "unrecorded"
#endif

#define GAT_VERSION "0.2.0" /* 0.2.0: gat_last_launch_info gained struct_size (ABI change), gat_set_option added */

extern "C" GAT_API const char *gat_version(void)
{
    return "libgat " GAT_VERSION " (gfx950) git:" GAT_GIT_SHA " flags:" GAT_BUILD_FLAGS;
}
