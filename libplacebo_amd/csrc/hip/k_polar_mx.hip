// k_polar_mx instantiations (k_polar_mx.hiph): RGB and RGBA tiles
#include "k_polar_mx.hiph"

int plh_launch_polar_mx(hipStream_t stream, const plh_pass *pass)
{
    if ((pass->s.comp_mask & 0xf) == 0x7)
        return launch_mx<3>(stream, pass);
    if ((pass->s.comp_mask & 0xf) == 0xf)
        return launch_mx<4>(stream, pass);
    return -1001;
}
