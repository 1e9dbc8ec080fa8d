/*
 * fixca_roll.h — tables the row-rolling kernel reads (built by k_roll_tables in fixca_roll.hip,
 * once per geometry and parameter block).  Plain structs, shared by the two kernels and the
 * launcher.
 */
#ifndef FIXCA_ROLL_H
#define FIXCA_ROLL_H

#include <stdint.h>

/* the 256-column source window of a strip starts on a multiple of (ROLL_U0_MASK + 1) pixels:
 * 4 pixels = 16 bytes is what the LDS-DMA and the 16-byte LDS reads need */
/* columns of a strip's source window held in a ring row.  256: one 16-byte-per-lane LDS-DMA per
 * row, the last lane's neighbour pixels come over a DPP wave shift, strips up to 248 columns.
 * 272: a second, 4-lane DMA adds 16 columns, every lane reads its 8 pixels itself, strips of 256
 * columns (full 1 KB row stores, 64 producing lanes).  Measured on 6000x4000 cubic: 272 is 4 %
 * slower (36.65 vs 35.1 us per image; the extra DMA and window reads cost more than the wider
 * strips save), so 256 is the default and 272 a build option (-DROLL_WIN=272). */
#ifndef ROLL_WIN
#define ROLL_WIN 256
#endif

/* RGB8's window (fixca_roll_rgb.hip): a build option for experiments; a multiple of 16 pixels (whole 16-byte DMA lanes) */
#ifndef ROLL_WIN_RGB
#define ROLL_WIN_RGB 256
#endif

#ifndef ROLL_U0_MASK
#define ROLL_U0_MASK 3
#endif

/* With a window wider than 256 columns the pixels after a lane's own 4 can be read from LDS by the
 * lane itself (ROLL_NEXT_FROM_LDS 1: strips of 256 columns, every lane produces) or still come from
 * lane L + 1 over DPP (0: strips up to 252 columns; the extra columns only let the window START on a
 * 64-byte boundary: -DROLL_WIN=272 -DROLL_U0_MASK=15 -DROLL_NEXT_FROM_LDS=0). */
#ifndef ROLL_NEXT_FROM_LDS
#define ROLL_NEXT_FROM_LDS (ROLL_WIN > 256)
#endif

/* one strip of `oc` output columns (16 bytes) */
struct RollStrip {
	int U0;             /* first column of the 256-column source window (multiple of 4)          */
	int tapR, tapB;     /* pixel offset, in a ring row, of lane 0's 4-pixel tap window (may be < 0) */
	int gen;            /* bit 0: some output of the strip needs the 5th weight (step or image edge); bit 1 / bit 2: the strip's
	                       first / last outputs are edge pixels out of their window's reach (fixca_roll.hip, roll_edge_pixels):
	                       bits 8-15 how many for red, bits 16-23 for blue */
};

/* one output row y, seen as the first row of a block (32 bytes) */
struct RollBlock {
	int rb[2];          /* first (virtual) source row of the red / blue window: floor(y') - BACK   */
	int keep;           /* oldest source row output row y touches                                  */
	int need1;          /* newest source row output row y touches                                  */
	int need4;          /* newest source row output rows y .. y+3 touch                            */
	int reg4;           /* bit n: both windows of row y+n slide down by exactly one unclamped row   */
	int pad[2];
};

struct RollTables {
	RollStrip *strips;  /* [nstrips]                                                               */
	float     *colw;    /* [nstrips][10][64 lanes][4]: a lane's 5 x 4 (red, blue) tap weights        */
	RollBlock *blocks;  /* [H]                                                                     */
	float     *roww;    /* [H][8]: (red, blue) vertical weights of tap 0..3                          */
};

/* the images of one launch (independent, same geometry and parameters): a kernel argument */
/* Images per launch.  6000x4000 cubic on ONE allocation of 128 images (tools/ab.py; the spread between rounds is
 * 0.1 point): 16 per launch 71.1 %, 32: 71.4, 64: 71.8, 128: 72.1 % of the HBM peak.  128 x 16 bytes of pointers
 * is 2 KiB of the 4 KiB a kernel's arguments may take. */
#ifndef ROLL_BATCH_MAX
#define ROLL_BATCH_MAX 128
#endif
struct RollBatch {
	const uint8_t *src[ROLL_BATCH_MAX];
	uint8_t       *dst[ROLL_BATCH_MAX];
};

/* The tile schedule of one launch (a kernel argument, built by roll_make_sched in fixca_plan.cpp):
 * every image's rows in chunks of `th`, every chunk in strips; image k's tiles are workgroups
 * [k * pool_blocks, (k + 1) * pool_blocks), numbered chunk-major and remapped so that each XCD owns
 * a contiguous range (neighbouring strips and chunks, whose source windows overlap, share an L2).
 * The dispatcher hands out workgroups in order: image k+1's first waves start in the slots image
 * k's last waves leave, so there is no fill/drain bubble per image.
 *
 * Round 2 tried two other schedules on top of this one and dropped both (git history: a75aaf2 ..
 * 601aab9; profiles/NOTES.md): big static pieces plus a pool of small tiles (+4-5 points in the
 * data-movement skeleton, +-0 in the kernel once neighbouring strips share an XCD again, -2 on
 * slow-write boxes), and persistent workgroups pulling tiles from per-XCD queues (+-0 at 32 images
 * per launch, -2 at 16). */
struct RollSched {
	int ncols;          /* n_images * nstrips                                                     */
	int rows;           /* y2 - y1                                                                */
	int pool_chunks;    /* chunks per image: ceil(rows / th)                                      */
	int pool_waves;     /* tiles per image: nstrips * pool_chunks                                 */
	int pool_blocks;    /* workgroups per image: pool_waves rounded up to a multiple of 8          */
	int th;             /* rows per tile                                                          */
	int edge_wb;        /* RGBA8: lanes on a strip's partial end sectors store write-back (roll_store); 1 in launches of
	                       several images.  A launch of one image ends with the write-back of those lines and measures
	                       3.6 points slower with it than without (4500 wide: 53.6 against 57.2 %), a batch 3.6-4.5
	                       points faster (62.5 -> 66.1 %, 65.6 -> 70.1 %)                      */
};

/* fixca_plan.cpp */
struct FixcaGeom;
struct FixcaPlan;
void roll_make_sched(const FixcaGeom *g, const FixcaPlan *pl, int n_images, RollSched *sc);

#endif
