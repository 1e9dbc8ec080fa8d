/*
 * fixca_roll_rgb.hip — the row-rolling kernel for RGB8 (3 bytes per pixel): fixca_roll.hip compiled with
 * ROLL_BPP = 3.  Ring rows are 768 bytes, moved by ONE 16-byte-per-lane LDS-DMA with 48 lanes active; a lane's
 * 4-pixel tap window is 12 bytes at any byte address: the 16 bytes from the dword boundary below are read and the
 * samples are picked by the byte selector of the conversion instruction, one of four instantiations per channel
 * chosen by a scalar branch (fixca_roll.hip, "RGB8 tap windows"); rows are stored 12 bytes per lane.
 */
#define ROLL_BPP 3
#include "fixca_roll.h"
#if ROLL_WIN_RGB != ROLL_WIN
#undef ROLL_WIN
#define ROLL_WIN ROLL_WIN_RGB
#undef ROLL_NEXT_FROM_LDS
#define ROLL_NEXT_FROM_LDS 0
#endif
#include "fixca_roll.hip"
