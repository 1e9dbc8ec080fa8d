/* the binding tools/gimp_box/apply_binding.py inserts into fix-ca.c, reduced to what must compile: the plug-in
 * declares FixCaParams itself (anonymous struct typedef, fix-ca.c:70-82), so the header is used in its
 * FIXCA_TYPES_NO_PARAMS mode and the pointer is cast to the opaque parameter type. */
#include <stdlib.h>
#define FIXCA_TYPES_NO_PARAMS 1
#include <fixca_hip.h>
typedef struct { double blue, red, lens_x, lens_y; int update_preview; int interpolation; double saturation, x_blue, x_red, y_blue, y_red; } FixCaParams;
int gimp_progress_update (double fraction);       /* libgimp's, declared here so that the binding compiles without it */
int f(unsigned char *a, unsigned char *b, FixCaParams *params, int show_progress) {
	if (fixca_region_host (a, b, 10, 10, 4, 1, (const struct FixCaParams_ *) params, 0, 10, 0, 10, show_progress) == FIXCA_OK) {
		if (show_progress)
			gimp_progress_update (1.0);
		return 0;
	}
	return 1;
}
