/*
 * libplacebo-hip: MI355X-native implementation of libplacebo's pl_render_image
 * hot path. Public configuration header (written fresh; mirrors the role of the
 * reference's generated src/include/libplacebo/config.h.in).
 */
#ifndef LIBPLACEBO_CONFIG_H_
#define LIBPLACEBO_CONFIG_H_

// API level of the reference this implementation tracks (meson.build:10-15)
#define PL_MAJOR_VER 7
#define PL_API_VER 365
#define PL_HAVE_HIP 1

#define PL_API __attribute__((visibility("default")))

#ifdef __cplusplus
# define PL_API_BEGIN extern "C" {
# define PL_API_END }
#else
# define PL_API_BEGIN
# define PL_API_END
#endif

#endif // LIBPLACEBO_CONFIG_H_
