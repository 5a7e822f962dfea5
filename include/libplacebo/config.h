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
#define PL_FIX_VER (pl_fix_ver())
#define PL_VERSION (pl_version())
#define PL_HAVE_HIP 1

// Members kept only for layout compatibility with older API levels are tagged like the
// reference tags them; they are never read by this implementation unless stated.
#ifndef PL_DEPRECATED_IN
# define PL_DEPRECATED_IN(VER) __attribute__((deprecated))
#endif
#ifndef PL_DEPRECATED_ENUM_IN
# define PL_DEPRECATED_ENUM_IN(VER) PL_DEPRECATED_IN(VER)
#endif

#define PL_API __attribute__((visibility("default")))

#ifdef __cplusplus
# define PL_API_BEGIN extern "C" {
# define PL_API_END }
#else
# define PL_API_BEGIN
# define PL_API_END
#endif

#ifndef __cplusplus
// the parameter macros rely on later designated initialisers overriding the defaults
# pragma GCC diagnostic ignored "-Woverride-init"
#endif

#define PL_TOSTRING_INNER(x) #x
#define PL_TOSTRING(x) PL_TOSTRING_INNER(x)

PL_API_BEGIN
PL_API int pl_fix_ver(void);
PL_API const char *pl_version(void);
PL_API_END

#endif // LIBPLACEBO_CONFIG_H_
