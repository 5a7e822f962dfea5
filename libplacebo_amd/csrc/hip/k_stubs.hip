// temporary stubs, replaced by k_ortho.hip / k_deband.hip
#include <hip/hip_runtime.h>
#include "plh_device.h"
int plh_launch_ortho(hipStream_t, const plh_pass *) { return -1; }
int plh_launch_deband(hipStream_t, const plh_pass *) { return -1; }

struct plh_errdiff_args;
extern "C" int plh_launch_errdiff(plh_stream, const plh_errdiff_args *) { return -1; }
