// k_polar_pp instantiations for float LDS tiles, one- and two-component planes (k_polar_pp.hiph)
#define PP_PLANE_MASKS
#include "k_polar_pp.hiph"

int plh_launch_polar_pp_f32_c12(hipStream_t stream, const plh_pass *pass, dim3 grid, dim3 block,
                                 size_t shmem, int n)
{
    return launch_pp_mask<float>(stream, pass, grid, block, shmem, n);
}
