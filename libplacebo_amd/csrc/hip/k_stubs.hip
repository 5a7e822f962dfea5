// temporary stubs, replaced by k_ortho.hip / k_deband.hip
#include <hip/hip_runtime.h>
#include "plh_device.h"

struct plh_errdiff_args;
extern "C" int plh_launch_errdiff(plh_stream, const plh_errdiff_args *) { return -1; }
