/* fixca_wide.h — launch shape of k_wcol (fixca_wide.hip), a kernel argument */
#ifndef FIXCA_WIDE_H
#define FIXCA_WIDE_H

#define WIDE_OC   120   /* output columns per wave: 60 lanes x 2 pixels                      */
#define WIDE_TH_MAX 128 /* most output rows per tile (per-row tables in LDS)                  */
#define WIDE_WIN  128   /* source columns per channel window: 64 lanes x 2                   */

#ifndef WIDE_SLOTS
#define WIDE_SLOTS 3   /* register queue of row loads: WIDE_SLOTS - 1 rows in flight ahead of the row being computed */
#endif

struct FixcaGeom;
/* waves of the geometry's k_wcol instantiation the device holds at once, 0 if there is no device to ask (fixca_wide.hip) */
extern "C" int fixca_wide_slots(const struct FixcaGeom *g, int fast);
/* the current HIP device (-1: none); k_wcol's tile height depends on it, so the planner's caches are keyed by it */
extern "C" int fixca_wide_device(void);

struct WidePlan {
	int oc, th, nstrips, nwaves, nblocks;
};

#endif
