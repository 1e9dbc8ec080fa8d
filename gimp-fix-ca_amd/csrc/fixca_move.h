/* fixca_move.h — launch shape of k_move (fixca_move.hip), a kernel argument */
#ifndef FIXCA_MOVE_H
#define FIXCA_MOVE_H

#define MOVE_DEPTH_MAX 14   /* most source rows kept in flight beyond the newest row in use (the planner picks: MovePlan.depth) */
/* output rows per block: the LDS reads of a block's rows are all issued before the first is used, so a wave exposes one
 * LDS latency per block instead of one per row; RB x NU = 4 units of 16 bytes per lane and block */
#ifndef MOVE_RB1
#define MOVE_RB1 4          /* rows per block with 1-KiB ring rows                                          */
#endif
#ifndef MOVE_RB2
#define MOVE_RB2 2          /* ... with 2-KiB ring rows                                                     */
#endif
#ifndef MOVE_RB4
#define MOVE_RB4 1          /* ... with 4-KiB ring rows                                                     */
#endif
#define MOVE_RB(nu) ((nu) == 1 ? MOVE_RB1 : (nu) == 2 ? MOVE_RB2 : MOVE_RB4)
#ifndef MOVE_BATCH_MAX
#define MOVE_BATCH_MAX 128  /* images per launch (fixca_batch_device): 2 KiB of kernel arguments, as RollBatch.  RGBA u16
                             * cubic 16 / 32 / 64 / 128 per launch: 52.3 / 53.0 / 53.3 / 53.4 %, u16 linear 62.6 / 63.8 / 64.3 /
                             * 64.5 %, float cubic 58.3 / 58.6 / 59.0 (64) %: profiles/r03_sweep_r3y_batch_size.txt */
#endif
#define MOVE_NU_MAX  4      /* most 16-byte units a lane moves per row (ring rows of 4 KB)                  */
#ifndef MOVE_LDS_MAX
#define MOVE_LDS_MAX (64 * 1024)   /* most ring bytes per wave                                              */
#endif

struct MoveBatch {
	const unsigned char *src[MOVE_BATCH_MAX];
	unsigned char       *dst[MOVE_BATCH_MAX];
};

struct MovePlan {
	int ocpx;           /* output columns per strip (a whole number of 16-byte units)                      */
	int th;             /* output rows per tile                                                            */
	int ring;           /* source rows in the LDS ring                                                     */
	int depth;          /* source rows kept in flight beyond the newest row in use                         */
	int nstrips, nwaves, nblocks;
	int edge_wb;        /* k_lin, RGBA: in launches of several images the lanes on a strip's partial end sectors store write-back
	                       (as k_roll: fixca_roll.hip, roll_store); 0 in a launch of one image and in the other kernels */
};

#endif
