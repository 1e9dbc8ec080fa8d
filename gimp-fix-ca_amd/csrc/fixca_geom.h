/*
 * fixca_geom.h — geometry of one fix_ca_region call, shared by the host planner
 * and the gfx950 kernels.  Plain C++ (no HIP types) so the planner also builds
 * and unit-tests on a machine without a GPU.
 */
#ifndef FIXCA_GEOM_H
#define FIXCA_GEOM_H

#include <stdint.h>

/* channel index: 0 = red (byte 0 of a pixel), 1 = blue (byte 2), bpc == 1 */
struct FixcaGeom {
	int    W, H;        /* image size in pixels (orig_width, orig_height)   */
	int    bpp;         /* bytes per pixel: 3 (RGB8) or 4 (RGBA8)           */
	int    xc, yc;      /* lens centre, C-truncated (fix-ca.c:1033-1034)    */
	int    y1, y2;      /* rows to produce                                   */
	int    interp;      /* 0 none, 1 linear, 2 cubic                         */
	int    bpc;         /* color_size() code: 1 u8, 2 u16, 4 u32, -4 float, -8 double */
	int    nch;         /* samples per pixel: 3 (RGB) or 4 (RGBA) = bpp / |bpc| */
	int    preview;     /* show_progress == 0: draw the lens-centre lines afterwards (fix-ca.c:1326) */
	int    pad_;
	double s[2];        /* scale about the centre (fix-ca.c:1044-1045)      */
	double xs[2];       /* directional x shift                               */
	double ys[2];       /* directional y shift                               */
};

/* launch shape chosen by the planner */
struct FixcaPlan {
	int kernel;         /* FIXCA_K_* below */
	int oc;             /* output columns per wave (multiple of 4)           */
	int th;             /* output rows per wave (<= 64)                      */
	int nstrips;        /* ceil(W / oc)                                      */
	int nchunks;        /* ceil((y2 - y1) / th)                              */
	int nwaves;         /* nstrips * nchunks                                 */
	int nblocks;        /* workgroups, padded to a multiple of 8 (256 threads
	                       for kernels 2,3; one 64-thread wave for kernel 4)  */
	int ring;           /* kernels 4, 6, 8: source rows in the LDS ring         */
	int nu;             /* kernels 8, 9, 10: KiB per ring row = 16-byte units a lane moves per output row (1, 2, 4) */
	int depth;          /* kernels 8, 9, 10: source rows kept in flight beyond the newest row in use */
	int wide_fast;      /* 16-bit samples, kernel 7 (CUBIC, fixca_wide.hip) and kernel 9 (LINEAR, fixca_lin.hip): the weight form with
	                       exact recomputation of the samples near a rounding boundary; 0 = operation for operation */
	int edge;           /* kernel 6: some strip has outputs that are the row's first / last pixel beyond the reach of their 5
	                       window pixels: rewritten at the end of each tile (fixca_roll.hip, roll_edge_pixels); informative */
};

#define FIXCA_K_EXACT   1
#define FIXCA_K_NONE    2
#define FIXCA_K_INTERP  3
#define FIXCA_K_SCAN    4
/* 5 was k_scanw (one ring shared by the 4 waves of a workgroup): measured no faster than k_scan, removed in round 3 */
#define FIXCA_K_ROLL    6   /* LDS ring, horizontal-first rolling window (fixca_roll.hip) */
#define FIXCA_K_WCOL    7   /* u16 / u32 / float / double samples: fp64 rolling window (fixca_wide.hip) */
#define FIXCA_K_MOVE    8   /* u16 / u32 / float / double samples, NONE: LDS-ring byte mover (fixca_move.hip) */
#define FIXCA_K_LIN     9   /* RGBA u32 / float / double (u16 on request), LINEAR: LDS ring + fp64 bilinear (fixca_lin.hip) */
#define FIXCA_K_CUB     10  /* RGBA u16 / u32 / float / double, CUBIC: LDS ring + per-lane rolling window (fixca_cub.hip) */

#define FIXCA_WAVE_COLS 256   /* 64 lanes x 4 pixels                           */

#endif
