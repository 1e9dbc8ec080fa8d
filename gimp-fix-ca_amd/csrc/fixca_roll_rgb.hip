/*
 * fixca_roll_rgb.hip — the row-rolling kernel for RGB8 (3 bytes per pixel): fixca_roll.hip compiled
 * with ROLL_BPP = 3.  Ring rows are 768 bytes (LDS-DMA moves 12 bytes per lane), a lane's 4-pixel
 * windows are 12 bytes at any byte address (read as 16 aligned bytes and shifted), and the
 * channel bytes sit at fixed positions of the 3 dwords.
 */
#define ROLL_BPP 3
#include "fixca_roll.hip"
