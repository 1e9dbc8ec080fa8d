/* fixca_move.h — launch shape of k_move (fixca_move.hip), a kernel argument */
#ifndef FIXCA_MOVE_H
#define FIXCA_MOVE_H

#define MOVE_DEPTH   6      /* source rows kept in flight beyond the newest row in use                     */
#define MOVE_NU_MAX  4      /* most 16-byte units a lane moves per row (ring rows of 4 KB)                  */
#define MOVE_LDS_MAX (64 * 1024)   /* most ring bytes per wave                                              */

struct MovePlan {
	int ocpx;           /* output columns per strip (a whole number of 16-byte units)                      */
	int th;             /* output rows per tile                                                            */
	int ring;           /* source rows in the LDS ring                                                     */
	int nstrips, nwaves, nblocks;
};

#endif
