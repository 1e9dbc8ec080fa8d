/* fixca_wide.h — launch shape of k_wcol (fixca_wide.hip), a kernel argument */
#ifndef FIXCA_WIDE_H
#define FIXCA_WIDE_H

#define WIDE_OC   120   /* output columns per wave: 60 lanes x 2 pixels                      */
#define WIDE_TH_MAX 128 /* most output rows per tile (per-row tables in LDS)                  */
#define WIDE_WIN  128   /* source columns per channel window: 64 lanes x 2                   */

struct WidePlan {
	int oc, th, nstrips, nwaves, nblocks;
};

#endif
