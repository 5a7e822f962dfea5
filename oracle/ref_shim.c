#include <libplacebo/filters.h>
int plref_shim_version(void) { return 1; }
