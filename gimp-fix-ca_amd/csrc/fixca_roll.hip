/*
 * fixca_roll.hip — the row-rolling kernel (RGBA8 and, compiled a second time through fixca_roll_rgb.hip,
 * RGB8; NONE / LINEAR / CUBIC): the main kernel of the path.
 *
 * Same data movement as the LDS-ring scan kernel (fixca_scan.hip):
 *
 *   HBM/L2 --LDS-DMA--> ring of K source rows in LDS --> registers --> HBM
 *
 * one wave (a 64-thread workgroup) owns `oc` output columns x `th` output rows, streams every
 * source row of its 256-column window into LDS once with buffer_load ... lds and waits for
 * exactly the row it needs with a counted s_waitcnt vmcnt.  What differs is the filter:
 *
 *  - HORIZONTAL FIRST, straight from the ring, no exchange through LDS.  The map is
 *    x' = (x - xc)*s + xc - shift with s within a fraction of a percent of 1, so over one strip the
 *    first tap of output column x sits at x + m or x + m + 1 for one strip constant m.  Lane L
 *    reads the 4 ring pixels [P + 4L, P + 4L + 4) (per channel and source row; P = m relative to
 *    the window) and takes the next 4 from lane L + 1 with a DPP wave-shift folded into the
 *    byte->float conversion (v_cvt_f32_ubyteN_dpp wave_shl:1): 8 pixels in registers, no data
 *    crosses LDS a second time.  The "+1" steps, and the reference's edge clamping
 *    (fix-ca.c:1270-1298), are folded into the lane's tap weights: output i uses pixels
 *    i .. i+4 of the 8 with 5 weights of which at most T are non-zero (clamped taps add their
 *    weights on the clamped column).  Waves whose strip has no step and touches no edge (most
 *    of them) run a T-tap loop instead.
 *  - The horizontally filtered row of each channel (4 floats per lane, red and blue packed in
 *    f32x2 for v_pk_fma_f32) enters a rolling window of T rows in registers; the vertical
 *    filter (fix-ca.c:911-920) is T packed FMAs per pixel with wave-uniform weights held in
 *    SGPRs.  This is the reference's own order: cubicY across columns, then cubicX across rows.
 *  - The kernel is bound by vector-instruction issue (one instruction per ~4.3 cycles per SIMD
 *    whatever the mix; measured), not by LDS, waits or HBM, so everything that does not depend
 *    on the pixels is taken out of it: a small table kernel (k_roll_tables, once per geometry
 *    and parameter block, cached by the launcher) computes
 *        per strip   the ring window, the two tap offsets and the general/T-tap choice,
 *        per lane    the 5 x 4 (red, blue) tap weights,
 *        per row     source rows, ring bounds, "slides by one" flags and vertical weights,
 *    with the reference's fp64 coordinate arithmetic, and the row loop reads them with
 *    coalesced vector loads (once per wave) and scalar loads (per block of 4 rows).
 *
 * fp32 sample arithmetic on fp64 coordinates: within 1 LSB of the reference; green and alpha
 * are copied bit-exactly.
 *
 * The planner (fixca_plan.cpp, plan_roll) selects this kernel when W >= 256 (a width that is not a multiple of 4 ends in a
 * lane of 1-3 pixels; RGB8 rows then start at any byte, which gfx950's buffer loads, LDS-DMA and stores take as it comes:
 * tools/skeletons/unaligned_probe.hip), every
 * strip's union window fits 256 columns, its tap offsets drift by at most one column and the
 * rows alive at once fit the ring; everything else goes to the kernels in fixca_scan.hip /
 * fixca_kernels.hip.
 */
#include "fixca_device.h"
#include "fixca_roll.h"
#include "fixca_internal.h"
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <type_traits>

typedef float        f32x2 __attribute__((ext_vector_type(2)));
typedef float        f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

/* This file is compiled twice: for RGBA8 (4 bytes per pixel, the default) and, through
 * fixca_roll_rgb.hip, for RGB8 (3 bytes per pixel).  Everything that depends on the pixel size
 * hangs on ROLL_BPP; entry points and kernels get a suffix so both can live in one library. */
#ifndef ROLL_BPP
#define ROLL_BPP 4
#endif
#if ROLL_BPP == 4
#define ROLL_NAME(x) x
#else
#define ROLL_NAME(x) x##_rgb
#define k_roll        k_roll_rgb
#define k_roll_tables k_roll_tables_rgb
#endif

#define RING_ROW    (ROLL_WIN * ROLL_BPP)   /* bytes per ring row                                  */

namespace ROLL_NAME(fixca_roll_ns) {    /* the two compilations define the same templates differently */
#define RING_PAD_LO 128                 /* a lane window may start up to 32 px before a ring row (the strips at the image's edges: roll_edge_pixels) */
#define RING_PAD_HI 128                 /* ... and the last lanes may read 32 px past it (the plug-in's amounts go to +-30) */
#ifndef ROLL_NRB
#define ROLL_NRB    4                   /* output rows per block of the regular path.  2 (135 instead of 166 VGPRs for RGB8
                                         * cubic) measured 1.5 points slower, 2 with 4 waves per SIMD forced 3.5 points slower */
#endif
#ifndef RDEPTH
#define RDEPTH      6                   /* source rows kept in flight beyond the newest row in use */
#endif
#ifndef ROLL_LD_AUX
#define ROLL_LD_AUX 0                   /* cache policy bits of the ring loads (2 = nt)           */
#endif
#ifndef ROLL_ST_AUX
#define ROLL_ST_AUX 2                   /* cache policy bits of the row stores (2 = nt)           */
#endif

/* v_cvt_pk_u8_f32 rounds to nearest EVEN; the reference rounds half away from zero (set_pixel,
 * fix-ca.c:752): ties must go up, which a bias of 2^-13 (8 fp32 ulps at 255) does. */
#define ROUND_BIAS 0.0001220703125f

template <int N> __device__ __forceinline__ void roll_wait_vm()
{
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int rclampi(int v, int lo, int hi) { return max(lo, min(v, hi)); }

/* scalar (SMEM) loads: the tables are read-only for the whole launch and every address below
 * is wave-uniform; the constant address space makes hipcc emit s_load_dword* */
typedef const __attribute__((address_space(4))) int   *cint_p;
typedef const __attribute__((address_space(4))) float *cfloat_p;

__device__ __forceinline__ RollStrip load_strip(cint_p strips, int i)
{
	cint_p p = strips + (size_t)i * (sizeof(RollStrip) / 4);
	RollStrip h;
	h.U0 = p[0]; h.tapR = p[1]; h.tapB = p[2]; h.gen = p[3];
	return h;
}
__device__ __forceinline__ RollBlock load_block(cint_p blocks, int y)
{
	cint_p p = blocks + (size_t)y * (sizeof(RollBlock) / 4);
	RollBlock b;
	b.rb[0] = p[0]; b.rb[1] = p[1]; b.keep = p[2]; b.need1 = p[3]; b.need4 = p[4]; b.reg4 = p[5];
	b.pad[0] = b.pad[1] = 0;
	return b;
}

/* byte 0 / byte 2 of the dword lane L + 1 holds, as a float (0 in lane 63): the byte->float
 * conversion with a DPP wave shift as its operand modifier.  Written as asm because hipcc folds
 * the shift into the conversion only in some schedules and otherwise spends two extra moves per
 * pixel.  No DPP hazard arises: the source registers are written by LDS loads, not by VALU
 * (tests/test_isa_checks.py looks at the generated code). */
__device__ __forceinline__ float next_lane_ubyte0(uint32_t v)
{
	float f;
	asm("v_cvt_f32_ubyte0_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f) : "v"(v));
	return f;
}
__device__ __forceinline__ float next_lane_ubyte2(uint32_t v)
{
	float f;
	asm("v_cvt_f32_ubyte2_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f) : "v"(v));
	return f;
}

/* Timing instrumentation lives in exp/fixca_roll_exp.h and is compiled in only with -DFIXCA_EXPERIMENTS
 * (never in the shipped library: such a build overwrites the output image with its numbers). */
#if defined(FIXCA_EXPERIMENTS)
#include "exp/fixca_roll_exp.h"
#else
#define ROLL_EXP_FIELDS
#define STAMP(i)                      do { } while (0)
#define ROLL_EXP_BEGIN(st)            do { } while (0)
#define ROLL_EXP_SINK(acc, NR)        do { } while (0)
#define ROLL_EXP_END(st, dst, lane, gen) do { } while (0)
#endif

template <int TAPS, bool P2, int ODD>  /* P2: the ring has a power-of-two number of rows; ODD: see roll_tile */
struct RollState {
	ROLL_EXP_FIELDS
	static constexpr int T = TAPS;
	int      lane;
	int      tap[2];            /* byte offset (may be negative) of my 4-pixel window in a ring row; RGB8: rounded down to a dword */
#if ROLL_BPP == 3
	int      rr[2];             /* RGB8: the window's bytes past that dword boundary, 0..3: the same for every lane (12 bytes per lane) */
#endif
	uint32_t my_idt;            /* byte offset of my 4 output pixels' own source pixels            */
	/* ring */
	uint8_t *ring;
	int      KM, H;             /* KM = K - 1 (a mask when K is a power of two)                 */
	uint32_t kmagic;            /* ceil(2^32 / K): row mod K = row - K * mulhi(row, kmagic), row < 2^24 */
	int      ld;                /* highest source row issued                                   */
	int      lastneed;          /* highest source row this wave ever needs                     */
	/* global memory goes through buffer descriptors: the per-lane part of an address is a
	 * constant 32-bit offset, the row part a scalar, and a lane that must not store gets an
	 * offset outside the buffer (dropped by the range check) instead of a branch */
	rsrc_t   rs_src, rs_dst;
	uint32_t vo_ld;             /* my byte offset inside a source row segment                   */
	uint32_t vo_st;             /* my byte offset inside an output row, past the end if I produce nothing */
	uint32_t so_ld;             /* byte offset of source row ld + 1                             */
	uint32_t so_st;             /* byte offset of the next output row                           */
	uint32_t pitch;
	uint32_t row_phase;         /* address of the strip's first byte in the next output row (only its low 6 bits count) */
	uint32_t strip_bytes;       /* bytes of an output row this wave stores */
	uint32_t dst_bytes;         /* size of the output descriptor: the offset that makes a store a no-op */
	int      edge_wb;           /* RollSched::edge_wb (scalar) */
	int      part;              /* width not a multiple of 4: pixels (1..3) of the strip's last, partial lane; 0 = none (scalar) */
	uint32_t vo_part;           /* ... that lane's byte offset inside an output row; past the end in every other lane */
	bool     idt_al;            /* my_idt is a multiple of 16 bytes (RGB8: of 4) (always, unless the window was pushed back from the right edge of such an image) */
	/* the newest T-1 horizontally filtered rows, (red, blue) of my 4 output columns */
	f32x2    hist[T > 1 ? T - 1 : 1][4];             /* unused when T == 1 */
};

/* ring slot of source row `row` (row >= 0): row mod K */
template <int TAPS, bool P2, int ODD>
__device__ __forceinline__ int roll_slot(const RollState<TAPS, P2, ODD> &st, int row)
{
	if constexpr (P2) return row & st.KM;
	else return row - (st.KM + 1) * (int)__umulhi((uint32_t)row, st.kmagic);
}

template <int TAPS, bool P2, int ODD>
__device__ __forceinline__ void roll_issue(RollState<TAPS, P2, ODD> &st)
{
	++st.ld;
	[[maybe_unused]] const int slot = roll_slot(st, st.ld);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ROLL_ABL_NOLOAD)       /* the host pass of hipcc does not know this builtin; ROLL_ABL_NOLOAD: timing experiment, no ring fill */
#if ROLL_BPP == 4
	__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(st.ring + slot * RING_ROW);
	__builtin_amdgcn_raw_ptr_buffer_load_lds(st.rs_src, (__attribute__((address_space(3))) void *)row, 16, st.vo_ld, st.so_ld, 0, ROLL_LD_AUX);
#if ROLL_WIN > 256
	/* columns 256 .. 271 of the window: 4 lanes x 16 bytes (the instruction offset applies to
	 * the global and to the LDS address alike) */
	if (st.lane < (ROLL_WIN - 256) / 4)
		__builtin_amdgcn_raw_ptr_buffer_load_lds(st.rs_src, (__attribute__((address_space(3))) void *)row, 16, st.vo_ld, st.so_ld, 1024, ROLL_LD_AUX);
#endif
#else
	/* RGB8: the 768 bytes of a ring row (256 pixels x 3) are moved by ONE LDS-DMA of 16 bytes per
	 * lane with 48 lanes active: contiguous in memory and in LDS (lane L's bytes land at 16 L).
	 * Round 1 used three one-dword-per-lane pieces (a 12-byte DMA would leave a 4-byte hole after
	 * every 4 pixels): three times the VMEM instructions of RGBA, the main reason RGB8 ran at 56 %
	 * of the HBM peak against RGBA's 67 %.  The global address is only 4-byte aligned (U0 * 3). */
	__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(st.ring + slot * RING_ROW);
	if (st.lane < RING_ROW / 16)
		__builtin_amdgcn_raw_ptr_buffer_load_lds(st.rs_src, (__attribute__((address_space(3))) void *)row, 16, st.vo_ld, st.so_ld, 0, ROLL_LD_AUX);
#endif
#endif
	st.so_ld += st.pitch;
}

/* keep RDEPTH rows in flight beyond `need` (within the ring's capacity), then wait until row
 * `need` has landed.
 * vmcnt counts loads and stores together.  Loads complete in order among loads and stores among
 * stores, but the compiler's own model of gfx9-family hardware does not order a load against a
 * store (a young store may be acknowledged before an old load has landed), so the wait relies
 * on loads only: rows are loaded in order, one load each, so exactly ld - need
 * rows' loads are younger than the one needed, and "at most ld - need operations outstanding" implies "at most
 * ld - need loads outstanding", which implies the needed row is in LDS.  Stores in flight make
 * the wait stricter, never wrong. */
template <int TAPS, bool P2, int ODD>
__device__ __forceinline__ void roll_fetch(RollState<TAPS, P2, ODD> &st, int keep, int need)
{
	const int target = min(min(need + RDEPTH, keep + st.KM), st.lastneed);
	while (st.ld < target) roll_issue(st);
	constexpr int LPR = 1 + ((ROLL_BPP == 4 && ROLL_WIN > 256) ? 1 : 0);     /* loads per row (RGB8: one DMA whatever the window) */
	const int younger = st.ld - need;          /* <= RDEPTH, and = RDEPTH except at the ring's capacity and at the end of a tile */
	if (younger == RDEPTH && RDEPTH <= 8) roll_wait_vm<RDEPTH * LPR>();
	else if (younger >= 8) roll_wait_vm<8 * LPR>();
	else if (younger >= 6) roll_wait_vm<6 * LPR>();
	else if (younger >= 4) roll_wait_vm<4 * LPR>();
	else if (younger >= 2) roll_wait_vm<2 * LPR>();
	else if (LPR > 1 && younger >= 1) roll_wait_vm<LPR>();
	else roll_wait_vm<0>();
}

/* my 4 output pixels' own source pixels of ring row `row`: 16 bytes at a 16-byte aligned offset
 * (RGBA) or 12 bytes at a 4-byte aligned offset (RGB; .w unused) */
struct __attribute__((packed, aligned(4))) RollPx4 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(4))) RollPx3 { uint32_t a, b, c; };
template <int TAPS, bool P2, int ODD>
__device__ __forceinline__ u32x4 roll_ring_px(const RollState<TAPS, P2, ODD> &st, int row, uint32_t off)
{
#if ROLL_BPP == 4
	if constexpr (ODD) {
		if (!st.idt_al) {                                /* wave-uniform; a misaligned ds_read_b128 is right but takes ~65 LDS cycles */
			const RollPx4 v = *reinterpret_cast<const RollPx4 *>(st.ring + roll_slot(st, row) * RING_ROW + off);
			return u32x4{ v.a, v.b, v.c, v.d };
		}
	}
	return *reinterpret_cast<const u32x4 *>(st.ring + roll_slot(st, row) * RING_ROW + off);
#else
	if constexpr (ODD) {
		if (!st.idt_al) {
			/* wave-uniform: the window was pushed back from the right edge of an image whose width is not a multiple of 4, my
			 * 12 bytes start 1-3 bytes past a dword: the 16 bytes around them, shifted down */
			const RollPx4 w = *reinterpret_cast<const RollPx4 *>(st.ring + roll_slot(st, row) * RING_ROW + (off & ~3u));
			const uint32_t sh = off & 3u;
			return u32x4{ __builtin_amdgcn_alignbyte(w.b, w.a, sh), __builtin_amdgcn_alignbyte(w.c, w.b, sh), __builtin_amdgcn_alignbyte(w.d, w.c, sh), 0u };
		}
	}
	const RollPx3 v = *reinterpret_cast<const RollPx3 *>(st.ring + roll_slot(st, row) * RING_ROW + off);
	return u32x4{ v.a, v.b, v.c, 0u };
#endif
}

/* the 4 pixels of my tap window of ring row `row`, at a pixel aligned offset.
 * RGBA: 16 bytes at a 4-byte aligned address.  A ds_read_b128 from an address that is not 16-byte
 * aligned returns the right data on gfx950 but takes ~65 LDS cycles instead of 4 (measured), so
 * the access is left to the compiler as 4-byte aligned: two ds_read2_b32.
 * RGB: 12 bytes at ANY byte address (3 bytes per pixel): the 16 bytes around them are read from the dword
 * boundary below (st.tap is that boundary); the bytes are picked out where they are converted. */
template <int TAPS, bool P2, int ODD>
__device__ __forceinline__ u32x4 roll_ring_win(const RollState<TAPS, P2, ODD> &st, int row, int off)
{
	const RollPx4 v = *reinterpret_cast<const RollPx4 *>(st.ring + roll_slot(st, row) * RING_ROW + off);
	return u32x4{ v.a, v.b, v.c, v.d };
}

#if ROLL_BPP == 3
/* RGB8 tap windows.  A lane's 4 pixels are 12 bytes starting R = 0..3 bytes past a dword boundary, the next lane's
 * 12 bytes follow directly (its own 16 bytes read starts 12 bytes on): sample c (0 red, 2 blue) of window pixel j is
 * byte R + 3 j + c of my 4 dwords for j < 4 and byte R + 3 (j - 4) + c of lane L + 1's for j >= 4.  R depends on the
 * strip and the channel only (12 bytes per lane keep it the same for every lane), so it is wave-uniform: one scalar
 * branch per block and channel picks the instantiation whose v_cvt_f32_ubyteN carry the right dword and byte as
 * constants.  Round 2 shifted the 16 bytes down by R first (three v_alignbyte per window read) and moved the next
 * lane's dwords over with v_mov_dpp before converting: 86 VALU instructions per wave and row against RGBA8's 69,
 * which is what held RGB8 11 points below RGBA8 (profiles/NOTES.md, finding 6).  Here the conversion IS the byte pick,
 * and the DPP shift rides on it as for RGBA.  The sources are LDS-load results, never fresh VALU results
 * (tests/test_isa_checks.py), so the hand-written DPP forms need no wait states. */
template <int K> __device__ __forceinline__ float own_ubyte(uint32_t v)
{
	return (float)((v >> (8 * K)) & 0xffu);                         /* v_cvt_f32_ubyteK */
}
template <int K> __device__ __forceinline__ float next_lane_ubyte(uint32_t v)
{
	float f;
	if constexpr (K == 0) asm("v_cvt_f32_ubyte0_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f) : "v"(v));
	else if constexpr (K == 1) asm("v_cvt_f32_ubyte1_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f) : "v"(v));
	else if constexpr (K == 2) asm("v_cvt_f32_ubyte2_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f) : "v"(v));
	else asm("v_cvt_f32_ubyte3_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f) : "v"(v));
	return f;
}
/* window pixels 0 .. NP-1 of NR source rows, channel byte C (0 red, 2 blue), misalignment R, into component C / 2 of tp */
template <int R, int C, int NR, int NP>
__device__ __forceinline__ void rgb_cvt_rows(const u32x4 (&p)[NR], f32x2 (&tp)[NR][NP])
{
#pragma unroll
	for (int n = 0; n < NR; ++n) {
#define RGB_PX(J) \
		if constexpr (NP > J) { \
			constexpr int B = R + 3 * ((J) & 3) + C; \
			float f; \
			if constexpr ((J) < 4) f = own_ubyte<B % 4>(p[n][B / 4]); else f = next_lane_ubyte<B % 4>(p[n][B / 4]); \
			if constexpr (C == 0) tp[n][J].x = f; else tp[n][J].y = f; \
		}
		RGB_PX(0) RGB_PX(1) RGB_PX(2) RGB_PX(3) RGB_PX(4) RGB_PX(5) RGB_PX(6) RGB_PX(7)
#undef RGB_PX
	}
}
template <int C, int NR, int NP>
__device__ __forceinline__ void rgb_cvt(int r, const u32x4 (&p)[NR], f32x2 (&tp)[NR][NP])
{
	switch (r) {                                                     /* wave-uniform: a scalar branch */
	case 0:  rgb_cvt_rows<0, C, NR, NP>(p, tp); break;
	case 1:  rgb_cvt_rows<1, C, NR, NP>(p, tp); break;
	case 2:  rgb_cvt_rows<2, C, NR, NP>(p, tp); break;
	default: rgb_cvt_rows<3, C, NR, NP>(p, tp); break;
	}
}
#endif

/* the 8 (NP used) pixels of the windows of NR source rows per channel, as (red, blue) floats.
 * pr / pb: my 4 pixels; with the 272-column window every lane has also read the next 4 (nr / nb),
 * with the 256-column window they come from lane L + 1 over DPP. */
template <int TAPS, bool P2, int ODD, int NR, int NP>
__device__ __forceinline__ void roll_px8(const RollState<TAPS, P2, ODD> &st, const u32x4 (&pr)[NR], const u32x4 (&pb)[NR],
                                         const u32x4 (&nr)[NR], const u32x4 (&nb)[NR], f32x2 (&tp)[NR][NP])
{
#if ROLL_BPP == 4
#pragma unroll
	for (int n = 0; n < NR; ++n) {
#pragma unroll
		for (int k = 0; k < 4; ++k)
			tp[n][k] = f32x2{ (float)(pr[n][k] & 0xffu), (float)((pb[n][k] >> 16) & 0xffu) };       /* v_cvt_f32_ubyte0 / ubyte2 */
#pragma unroll
		for (int k = 4; k < NP; ++k)
#if ROLL_NEXT_FROM_LDS
			tp[n][k] = f32x2{ (float)(nr[n][k - 4] & 0xffu), (float)((nb[n][k - 4] >> 16) & 0xffu) };
#else
			tp[n][k] = f32x2{ next_lane_ubyte0(pr[n][k - 4]), next_lane_ubyte2(pb[n][k - 4]) };
#endif
	}
#else
	static_assert(!ROLL_NEXT_FROM_LDS, "RGB8 takes the neighbour pixels over DPP");
	rgb_cvt<0, NR, NP>(st.rr[0], pr, tp);
	rgb_cvt<2, NR, NP>(st.rr[1], pb, tp);
#endif
}

/* the pixels after my 4-pixel tap window (272-column window only; a dummy otherwise) */
template <int TAPS, bool P2, int ODD>
__device__ __forceinline__ u32x4 roll_ring_next(const RollState<TAPS, P2, ODD> &st, int row, int off)
{
#if ROLL_NEXT_FROM_LDS
	return roll_ring_win(st, row, off + 4 * ROLL_BPP);
#else
	return u32x4{ 0u, 0u, 0u, 0u };
#endif
}

/* horizontal filter of one source row per channel (red from ring row rowR, blue from rowB):
 * h[i] = sum_t wx[i][t] * px8[i + t] for my 4 output columns */
template <int TAPS, bool P2, int ODD, int NW>
__device__ __forceinline__ void roll_hrow(const RollState<TAPS, P2, ODD> &st, const f32x2 (&wx)[4][5],
                                          int rowR, int rowB, f32x2 (&h)[4])
{
	const u32x4 pr[1] = { roll_ring_win(st, rowR, st.tap[0]) }, nr[1] = { roll_ring_next(st, rowR, st.tap[0]) };
	const u32x4 pb[1] = { roll_ring_win(st, rowB, st.tap[1]) }, nb[1] = { roll_ring_next(st, rowB, st.tap[1]) };
	f32x2 tp[1][NW + 3];
	roll_px8<TAPS, P2, ODD, 1, NW + 3>(st, pr, pb, nr, nb, tp);
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		f32x2 a = wx[i][0] * tp[0][i];
#pragma unroll
		for (int t = 1; t < NW; ++t) a = __builtin_elementwise_fma(wx[i][t], tp[0][i + t], a);
		h[i] = a;
	}
}

/* the T (red, blue) vertical weight pairs of output row y: scalar loads, the pairs stay in SGPRs */
template <int TAPS>
__device__ __forceinline__ void roll_wy(cfloat_p roww, int y, f32x2 (&wy)[TAPS])
{
	cfloat_p p = roww + (size_t)y * 8;
#pragma unroll
	for (int t = 0; t < TAPS; ++t) wy[t] = f32x2{ p[2 * t], p[2 * t + 1] };
}

/* ... of NR consecutive rows: one address, constant offsets (with a separate 64-bit address per row the compiler spends
 * four scalar instructions per row on it) */
template <int TAPS, int NR>
__device__ __forceinline__ void roll_wy_rows(cfloat_p roww, int y, f32x2 (&wy)[NR][TAPS])
{
	cfloat_p p = roww + (size_t)y * 8;
#pragma unroll
	for (int n = 0; n < NR; ++n)
#pragma unroll
		for (int t = 0; t < TAPS; ++t) wy[n][t] = f32x2{ p[8 * n + 2 * t], p[8 * n + 2 * t + 1] };
}

/* A row store of a strip whose ends are not on 64-byte boundaries of the address space (an RGBA8 width that is not a
 * multiple of 16 pixels, or an image that does not start on a 64-byte boundary) leaves the strip's first and last
 * 64-byte sector half written; the neighbour strip's wave writes the other half.  Streamed (nt) to memory one half at a
 * time, such a sector costs a read-modify-write each.  Measured with this kernel on 4496-, 4500- and 4512-pixel wide
 * RGBA8 images: 70.5 / 65.7 / 70.4 % of the HBM peak (profiles/r04_pitch_alignment_widths.txt).  So the lanes that
 * touch a partial end sector store write-back instead (the two halves meet in the XCD's L2 and leave it as one
 * sector), everything else stays nt: 4500 wide 65.6 -> 70.1 % (profiles/r04_edge_sector_policy.txt).  Which rows have
 * partial sectors is wave-uniform (a scalar branch); aligned images never take the second store.  Only in launches of
 * several images (RollSched::edge_wb): a launch of one image ends with the write-back of those lines and loses more
 * than the merged sectors gain.
 * RGB8 (12 bytes per lane, 720-byte strips, almost every row unaligned) measured the same with and without
 * (63.1 / 63.3 %): it keeps the single nt store.  ROLL_EDGE_AUX: cache policy bits of the edge stores (0 = write-back;
 * sc0 = 1 measures the same, sc1 = 16 is four times slower). */
#ifndef ROLL_EDGE_AUX
#define ROLL_EDGE_AUX 0
#endif
#ifndef ROLL_EDGE_SPLIT
#define ROLL_EDGE_SPLIT (ROLL_BPP == 4)
#endif

/* saturating round-to-nearest conversion straight into the red and blue bytes, then the row store */
template <int TAPS, bool P2, int ODD>
__device__ __forceinline__ void roll_store(RollState<TAPS, P2, ODD> &st, const f32x2 (&acc)[4], const u32x4 &idt)
{
	/* The row offset is added into the vector offset on purpose.  With the row offset in the
	 * scalar-offset operand hipcc emits no wait state between a 16-byte buffer store and the
	 * next write of its data registers (its hazard table exempts stores with an SGPR soffset);
	 * on MI355X that store then picks up the NEXT row's bytes in the last quarter of each
	 * 16-lane group (seen as sporadic wrong pixels).  With a constant soffset the compiler
	 * handles the hazard. */
#if ROLL_BPP == 4
	u32x4 out;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		uint32_t o = __builtin_amdgcn_cvt_pk_u8_f32(acc[k].x, 0, idt[k]);
		out[k] = __builtin_amdgcn_cvt_pk_u8_f32(acc[k].y, 2, o);
	}
#if ROLL_EDGE_SPLIT
	bool split = false;
	uint32_t a0 = 0, end = 0;
	if (__builtin_expect(st.edge_wb != 0, 0)) {
		/* address (mod 64) of the strip's first byte in this row and one past its last: scalar */
		a0 = st.row_phase & 63u;
		end = a0 + st.strip_bytes;
		st.row_phase += st.pitch;
		split = (a0 | (end & 63u)) != 0;
	}
	if (__builtin_expect(split, 0)) {
		const uint32_t first = (uint32_t)st.lane * 16u + a0;          /* my 16 bytes, relative to the first sector's start */
		const bool edge = (a0 != 0 && first < 64u) || ((end & 63u) != 0 && ((first + 15u) >> 6) >= ((end - 1) >> 6));
		const uint32_t off = st.vo_st + st.so_st, oob = st.dst_bytes;
		__builtin_amdgcn_raw_buffer_store_b128(out, st.rs_dst, edge ? oob : off, 0, ROLL_ST_AUX);
		__builtin_amdgcn_raw_buffer_store_b128(out, st.rs_dst, edge ? off : oob, 0, ROLL_EDGE_AUX);
	} else
#endif
	__builtin_amdgcn_raw_buffer_store_b128(out, st.rs_dst, st.vo_st + st.so_st, 0, ROLL_ST_AUX);
	if ((ODD) && st.part != 0) {
		/* the image's last 1-3 columns: one lane, as many dwords (a 16-byte store would run into the next row) */
		typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
		typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
		const uint32_t off = st.vo_part + st.so_st;
		if (st.part == 1)      __builtin_amdgcn_raw_buffer_store_b32(out[0], st.rs_dst, off, 0, ROLL_ST_AUX);
		else if (st.part == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2{ out[0], out[1] }, st.rs_dst, off, 0, ROLL_ST_AUX);
		else                   __builtin_amdgcn_raw_buffer_store_b96(u32x3{ out[0], out[1], out[2] }, st.rs_dst, off, 0, ROLL_ST_AUX);
	}
#else
	/* [R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3] */
	typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
	uint32_t o0 = idt[0], o1 = idt[1], o2 = idt[2];
	o0 = __builtin_amdgcn_cvt_pk_u8_f32(acc[0].x, 0, o0);
	o0 = __builtin_amdgcn_cvt_pk_u8_f32(acc[0].y, 2, o0);
	o0 = __builtin_amdgcn_cvt_pk_u8_f32(acc[1].x, 3, o0);
	o1 = __builtin_amdgcn_cvt_pk_u8_f32(acc[1].y, 1, o1);
	o1 = __builtin_amdgcn_cvt_pk_u8_f32(acc[2].x, 2, o1);
	o2 = __builtin_amdgcn_cvt_pk_u8_f32(acc[2].y, 0, o2);
	o2 = __builtin_amdgcn_cvt_pk_u8_f32(acc[3].x, 1, o2);
	o2 = __builtin_amdgcn_cvt_pk_u8_f32(acc[3].y, 3, o2);
	__builtin_amdgcn_raw_buffer_store_b96(u32x3{ o0, o1, o2 }, st.rs_dst, st.vo_st + st.so_st, 0, ROLL_ST_AUX);
	if ((ODD) && st.part != 0) {
		/* the image's last 1-3 columns (a width that is not a multiple of 4): one lane, 3 / 6 / 9 bytes, byte by byte
		 * (a 12-byte store would run into the next row) */
		const uint32_t off = st.vo_part + st.so_st;
		const uint32_t d[3] = { o0, o1, o2 };
#pragma unroll
		for (int k = 0; k < 9; ++k)
			if (k < 3 * st.part)
				__builtin_amdgcn_raw_buffer_store_b8((unsigned char)(d[k >> 2] >> (8 * (k & 3))), st.rs_dst, off + k, 0, ROLL_ST_AUX);   /* the other lanes: off >= the descriptor's size */
	}
#endif
	st.so_st += st.pitch;
	/* stores are not counted by the fetch function's wait: see there */
}

/* one output row, windows rebuilt from the ring (first row of a wave, scale steps, rows whose
 * window touches the top or bottom edge: edge rows replicate, fix-ca.c:1218-1256) */
template <int TAPS, bool P2, int ODD, int NW>
__device__ __forceinline__ void roll_row_any(RollState<TAPS, P2, ODD> &st, const f32x2 (&wx)[4][5], const RollBlock &b,
                                             cfloat_p roww, int y)
{
	constexpr int T = TAPS;
	roll_fetch(st, b.keep, b.need1);
	const u32x4 idt = roll_ring_px(st, y, st.my_idt);
	f32x2 rows[T][4];
#pragma unroll
	for (int t = 0; t < T; ++t)
		roll_hrow<TAPS, P2, ODD, NW>(st, wx, rclampi(b.rb[0] + t, 0, st.H - 1), rclampi(b.rb[1] + t, 0, st.H - 1), rows[t]);
	f32x2 v[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) v[k] = f32x2{ ROUND_BIAS, ROUND_BIAS };
	f32x2 wy[T];
	roll_wy<TAPS>(roww, y, wy);
#pragma unroll
	for (int t = 0; t < T; ++t)
#pragma unroll
		for (int k = 0; k < 4; ++k) v[k] = __builtin_elementwise_fma(wy[t], rows[t][k], v[k]);
#pragma unroll
	for (int t = 0; t + 1 < T; ++t)
#pragma unroll
		for (int k = 0; k < 4; ++k) st.hist[t][k] = rows[t + 1][k];
	roll_store(st, v, idt);
}

/* NR (1 or 4) consecutive output rows whose windows each slide down by one unclamped source row.
 * All ring reads of the rows are issued together; the filter arithmetic is written tap-major
 * (for each tap, all rows and outputs) so that consecutive instructions belong to different
 * accumulation chains. */
template <int TAPS, bool P2, int ODD, int NW, int NR>
__device__ __forceinline__ void roll_block(RollState<TAPS, P2, ODD> &st, const f32x2 (&wx)[4][5], const RollBlock &b,
                                           const RollBlock &bn, cfloat_p roww, int y)
{
	constexpr int T = TAPS;
	constexpr int NP = NW + 3;                      /* pixels of the 8 that carry weight */
	STAMP(0);                                       /* everything outside blocks */
	roll_fetch(st, b.keep, NR > 1 ? b.need4 : b.need1);     /* need4 covers rows y .. y+3: right for 4, safe for 2 */
	STAMP(1);                                       /* fetch: DMA issue + wait */
	const int nR = b.rb[0] + T - 1, nB = b.rb[1] + T - 1;   /* newest row of output row y */

	u32x4 idt[NR], pr[NR], pb[NR], nr[NR], nb[NR];
#pragma unroll
	for (int n = 0; n < NR; ++n) {
		pr[n] = roll_ring_win(st, nR + n, st.tap[0]);
		nr[n] = roll_ring_next(st, nR + n, st.tap[0]);
		pb[n] = roll_ring_win(st, nB + n, st.tap[1]);
		nb[n] = roll_ring_next(st, nB + n, st.tap[1]);
		idt[n] = roll_ring_px(st, y + n, st.my_idt);
	}
	f32x2 acc[NR][4], wy[NR][T];
	roll_wy_rows<TAPS, NR>(roww, y, wy);
	/* Everything this block needs from the ring is now in registers (after the wait), so the
	 * slots of rows the NEXT block no longer touches are free already: refill them now, and the
	 * loads travel while this block computes.  Issuing only at block boundaries would leave the
	 * wave with no load in flight for most of a block. */
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	{
		const int target = min(min(bn.need4 + RDEPTH, bn.keep + st.KM), st.lastneed);
		if (target - st.ld == NR && NR > 1) {              /* the usual case: as many rows as the block consumed */
#pragma unroll
			for (int n = 0; n < NR; ++n) roll_issue(st);
		} else
			while (st.ld < target) roll_issue(st);
	}
	STAMP(2);                                       /* ring reads issued and landed */
	f32x2 rows[T - 1 + NR][4];                      /* hist rows, then the NR new rows */
#pragma unroll
	for (int t = 0; t + 1 < T; ++t)
#pragma unroll
		for (int k = 0; k < 4; ++k) rows[t][k] = st.hist[t][k];

	/* horizontal: rows[T-1+n][i] = sum_t wx[i][t] * px8[n][i + t] */
	f32x2 tp[NR][NP];
	roll_px8<TAPS, P2, ODD, NR, NP>(st, pr, pb, nr, nb, tp);
#pragma unroll
	for (int t = 0; t < NW; ++t)
#pragma unroll
		for (int n = 0; n < NR; ++n)
#pragma unroll
			for (int i = 0; i < 4; ++i)
				rows[T - 1 + n][i] = (t == 0) ? wx[i][0] * tp[n][i]
				                              : __builtin_elementwise_fma(wx[i][t], tp[n][i + t], rows[T - 1 + n][i]);

	/* vertical: acc[n][k] = bias + sum_t wy[n][t] * rows[n + t][k] */
#pragma unroll
	for (int t = 0; t < T; ++t)
#pragma unroll
		for (int n = 0; n < NR; ++n)
#pragma unroll
			for (int k = 0; k < 4; ++k)
				acc[n][k] = __builtin_elementwise_fma(wy[n][t], rows[n + t][k], (t == 0) ? f32x2{ ROUND_BIAS, ROUND_BIAS } : acc[n][k]);
#pragma unroll
	for (int t = 0; t + 1 < T; ++t)
#pragma unroll
		for (int k = 0; k < 4; ++k) st.hist[t][k] = rows[NR + t][k];
#ifdef ROLL_EXP_VALU
	/* timing experiment (tools/build_variant.sh NAME -DROLL_EXP_VALU=n, never the shipped library): n more packed
	 * FMAs per output row on four independent chains: what a heavier filter of the same data movement would cost */
	{
		f32x2 d[4] = { acc[0][0], acc[0][1], acc[0][2], acc[0][3] };
#pragma unroll
		for (int i = 0; i < ROLL_EXP_VALU * NR; ++i)
			asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i & 3]) : "v"(wy[0][0]));
		asm volatile("" :: "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
	}
#endif
	ROLL_EXP_SINK(acc, NR);
	STAMP(3);                                       /* filter arithmetic */
#pragma unroll
	for (int n = 0; n < NR; ++n) roll_store(st, acc[n], idt[n]);
	STAMP(4);                                       /* conversions + stores */
}

/* Edge pixels.  An output whose source coordinate falls outside the image is the row's first or last pixel, whatever
 * the lens does to its neighbours (the coordinate is clamped: fix-ca.c:799-819).  In the strips at the image's left and
 * right edge such outputs sit next to outputs whose taps are 5 and more columns away, out of reach of the 5 window
 * pixels an output has: until round 4 that sent every 6000-pixel image with an amount of about -4 and beyond to the scan
 * kernel (44 .. 27 % of the HBM peak instead of 70 %).  Now the tables give such an output no weight at all (and leave it
 * out of the strip's tap offsets), the row loop stores whatever that makes of it, and the wave that owns the tile
 * rewrites those samples when its rows are done: lane k takes row Y0 + k (+ 64, ...), filters the edge column's T
 * samples with the row's vertical weights, in the row loop's own arithmetic, and stores the byte n times, for the strip's
 * n such outputs per channel (the first n of a row, or the last n: the map is monotone; the planner checks).  A few
 * hundred byte accesses per tile, in 2 strips of 25, after a wait for the tile's own stores (the order of two stores of
 * one wave to one address).  An edge output the 5 window pixels do reach is an output like any other.
 * Other forms tried: the window pixel overwritten inside the row loop, as a third instantiation of the kernel, 50 % (181
 * registers); the same in a row-by-row copy of the loop for the edge strips, 68.6 % in a launch of many images but 28 %
 * in a launch of one, where the slow edge tiles are the launch's tail (profiles/r04_edge_pixels.txt). */
template <int TAPS>
__device__ __forceinline__ void roll_edge_pixels(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const FixcaGeom &g,
                                              const RollTables &tb, int gen, int Y0, int nrows, int lane)
{
	constexpr int T = TAPS;
	const int W = g.W, H = g.H;
	const size_t pitch = (size_t)W * ROLL_BPP;
	const int n[2] = { (gen >> 8) & 0xff, (gen >> 16) & 0xff };
	const int ecol = (gen & 2) ? 0 : W - 1;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    /* the tile's own stores of these bytes first */
	for (int r = lane; r < nrows; r += 64) {
		const int y = Y0 + r;
		const RollBlock b = tb.blocks[y];
		const float *wy = tb.roww + (size_t)y * 8;
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			if (n[c] == 0) continue;
			float acc = ROUND_BIAS;
#pragma unroll
			for (int t = 0; t < T; ++t) {
				const int row = rclampi(b.rb[c] + t, 0, H - 1);
				acc = __builtin_fmaf(wy[2 * t + c], (float)src[(size_t)row * pitch + (size_t)ecol * ROLL_BPP + 2 * c], acc);
			}
			const uint8_t v = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(acc, 0, 0u);
			uint8_t *o = dst + (size_t)y * pitch + (size_t)((gen & 2) ? 0 : W - n[c]) * ROLL_BPP + 2 * c;
			for (int k = 0; k < n[c]; ++k) o[(size_t)k * ROLL_BPP] = v;
		}
	}
}

/* one tile: output rows [Y0, Y0 + nrows) of strip `strip` of image `img`.
 * ODD: the tile may end in a lane of 1-3 pixels, or read its own pixels from a ring offset that is not lane-aligned: the
 * last strips of an image whose width is not a multiple of 4.  Such images run their own instantiation of the kernel
 * (wave-uniform branches per row in every strip); for all others the code is compiled out: as scalar branches in the one
 * kernel it cost the single-image launch 1.2 points (RGB8: 2.1), as a second variant inside the one kernel 0.7 (1.9:
 * scalar registers spilled). */
template <int TAPS, bool P2, int ODD>
__device__ __forceinline__ void
roll_tile(uint8_t *lds, const RollBatch &imgs, const FixcaGeom &g, const FixcaPlan &pl, const RollTables &tb,
          int img, int strip, int Y0, int nrows, int edge_wb)
{
	constexpr int T = TAPS;
	const int lane = threadIdx.x;
	const uint8_t *__restrict__ src = imgs.src[img];
	uint8_t *__restrict__ dst = imgs.dst[img];
	const int W = g.W, H = g.H, oc = pl.oc;
	const int X0 = strip * oc;                      /* the last strip may be narrower: its extra lanes store nothing */

	RollState<TAPS, P2, ODD> st;
	st.lane = threadIdx.x;
	ROLL_EXP_BEGIN(st);
	st.KM = pl.ring - 1;
	st.kmagic = (uint32_t)((0x100000000ull + (unsigned)pl.ring - 1) / (unsigned)pl.ring);
	st.H = H;
	st.pitch = (uint32_t)W * ROLL_BPP;
	st.ring = lds + RING_PAD_LO;

	/* ---- tables: strip header, first and last block descriptor (scalar loads) ------------ */
	const cint_p strips = (cint_p)(const int *)tb.strips;
	const cint_p blocks = (cint_p)(const int *)tb.blocks;
	const cfloat_p roww = (cfloat_p)tb.roww;
	const RollStrip hd = load_strip(strips, strip);
	RollBlock b = load_block(blocks, Y0);
	st.lastneed = load_block(blocks, Y0 + nrows - 1).need1;
	const int U0 = hd.U0;

	/* Buffer descriptors of THIS tile's rows: source rows keep .. lastneed, output rows Y0 ..
	 * Y0 + nrows - 1.  Offsets inside them are 32-bit (the planner keeps a tile's rows below 2 GB)
	 * whatever the size of the image.  Built from kernel arguments and scalar loads only, so they
	 * are provably wave-uniform. */
	const uint32_t dst_bytes = (uint32_t)nrows * st.pitch;
	st.rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + (size_t)b.keep * st.pitch, 0,
	                                              (uint32_t)(st.lastneed - b.keep + 1) * st.pitch, 0x00020000);
	st.rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)Y0 * st.pitch, 0, dst_bytes, 0x00020000);

	/* what the tile's last step needs (roll_edge_pixels), parked in vector registers: the row loop has no scalar
	 * register to spare (kept in scalars these five spilled three others in the cubic kernel) */
	uint32_t park[5] = { (uint32_t)(uintptr_t)src, (uint32_t)((uintptr_t)src >> 32), (uint32_t)(uintptr_t)dst, (uint32_t)((uintptr_t)dst >> 32), (uint32_t)hd.gen };
	asm volatile("" : "+v"(park[0]), "+v"(park[1]), "+v"(park[2]), "+v"(park[3]), "+v"(park[4]));

	/* start filling the ring before anything else */
	st.ld = b.keep - 1;
	st.vo_ld = (ROLL_BPP == 4) ? (uint32_t)(U0 + 4 * lane) * 4u : (uint32_t)U0 * 3u + 16u * (uint32_t)lane;
	st.so_ld = 0;
	{
		const int target = min(min(b.need1 + RDEPTH, b.keep + st.KM), st.lastneed);
		while (st.ld < target) roll_issue(st);
	}

	/* ---- per-lane tap weights: 10 coalesced 16-byte loads ---------------------------------- */
	const int wcols = min(oc, W - X0);               /* output columns of this strip */
	const int nact = (wcols + 3) >> 2;               /* lanes that produce output; the last one may hold 1-3 pixels (RGBA8, W % 4 != 0) */
	const int nfull = wcols >> 2;                    /* ... of which these store 4 pixels */
	f32x2 wx[4][5];
	{
		const f32x4 *cw = reinterpret_cast<const f32x4 *>(tb.colw) + (size_t)strip * (10 * 64) + lane;
		float f[40];
#pragma unroll
		for (int q = 0; q < 10; ++q) {
			const f32x4 v = cw[q * 64];
			f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
		}
#pragma unroll
		for (int i = 0; i < 4; ++i)
#pragma unroll
			for (int jj = 0; jj < 5; ++jj) wx[i][jj] = f32x2{ f[(i * 5 + jj) * 2], f[(i * 5 + jj) * 2 + 1] };
	}
	/* lanes past the last producing lane only serve as its DPP neighbour: they all read where
	 * lane `nact` reads, so a narrow last strip never reads far outside its ring row */
	constexpr int NB = ROLL_NEXT_FROM_LDS ? 1 : 0;     /* 256-column window: lane nact serves as lane nact-1's DPP neighbour */
	st.tap[0] = (hd.tapR + 4 * min(lane, nact - NB)) * ROLL_BPP;
	st.tap[1] = (hd.tapB + 4 * min(lane, nact - NB)) * ROLL_BPP;
#if ROLL_BPP == 3
	st.rr[0] = (hd.tapR * 3) & 3;                    /* scalar: the strip header comes from scalar loads */
	st.rr[1] = (hd.tapB * 3) & 3;
	st.tap[0] &= ~3;
	st.tap[1] &= ~3;
#endif
	st.my_idt = (uint32_t)((X0 - U0) + 4 * min(lane, nact - 1)) * ROLL_BPP;
	/* a lane without output stores at the buffer's size: past the end whether or not the range
	 * check adds the row offset (the size is below 2 GB, so the sum cannot wrap) */
	st.vo_st = (lane < nfull) ? (uint32_t)(X0 + 4 * lane) * ROLL_BPP : dst_bytes;
#ifdef ROLL_ABL_NOSTORE            /* timing experiment (never the shipped library): every store out of range */
	st.vo_st = dst_bytes;
#endif
	st.so_st = 0;
	st.part = wcols & 3;
	st.vo_part = (st.part && lane == nfull) ? (uint32_t)(X0 + 4 * lane) * ROLL_BPP : dst_bytes;
	st.idt_al = ((X0 - U0) & 3) == 0;
	st.row_phase = (uint32_t)(uintptr_t)dst + (uint32_t)Y0 * st.pitch + (uint32_t)X0 * ROLL_BPP;      /* mod 2^32: only the low bits are used */
	st.strip_bytes = (uint32_t)wcols * ROLL_BPP;
	st.dst_bytes = dst_bytes;
	st.edge_wb = edge_wb;

	auto run = [&](auto nw_tag) {
		constexpr int NW = decltype(nw_tag)::value;
		int r = 0;
		while (r < nrows) {
			const int y = Y0 + r;
			constexpr int NRB = ROLL_NRB, RMASK = (1 << NRB) - 1;
			const bool four = r > 0 && r + NRB - 1 < nrows && (b.reg4 & RMASK) == RMASK;
			const bool one = r > 0 && (b.reg4 & 1);      /* tails and rows next to an irregular one */
			const int step = four ? NRB : 1;
			/* the next descriptor is requested before this block's work: its scalar load has the
			 * whole block to come back */
			RollBlock bn = b;
			if (r + step < nrows) bn = load_block(blocks, y + step);
			if (four)     roll_block<TAPS, P2, ODD, NW, NRB>(st, wx, b, bn, roww, y);
			else if (one) roll_block<TAPS, P2, ODD, NW, 1>(st, wx, b, bn, roww, y);
			else          roll_row_any<TAPS, P2, ODD, NW>(st, wx, b, roww, y);
			r += step;
			b = bn;
		}
	};
	STAMP(5);                                       /* prologue */
	if (hd.gen & 1) run(std::integral_constant<int, 5>{});
	else            run(std::integral_constant<int, (T == 1) ? 2 : T>{});
	const int gen_end = __builtin_amdgcn_readfirstlane((int)park[4]);
	if (__builtin_expect((gen_end & 6) != 0, 0)) {
		auto addr = [&](uint32_t lo, uint32_t hi) {      /* (readfirstlane returns int: through uint32_t, or a low word with bit 31 set sign-extends into the high one) */
			return (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane(lo) | ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane(hi) << 32);
		};
		const uint8_t *s_end = reinterpret_cast<const uint8_t *>(addr(park[0], park[1]));
		uint8_t *d_end = reinterpret_cast<uint8_t *>(addr(park[2], park[3]));
		roll_edge_pixels<TAPS>(s_end, d_end, g, tb, gen_end, Y0, nrows, lane);
	}
	ROLL_EXP_END(st, dst, lane, (unsigned long long)hd.gen | ((unsigned long long)strip << 8) | ((unsigned long long)Y0 << 20));
}

/* build option: hold the kernel to the registers that let N waves share a SIMD (-DROLL_WAVES_PER_EU=3: 168) */
#ifdef ROLL_WAVES_PER_EU
#define ROLL_OCC __attribute__((amdgpu_waves_per_eu(ROLL_WAVES_PER_EU, ROLL_WAVES_PER_EU)))
#else
#define ROLL_OCC
#endif
template <int TAPS, bool P2, int ODD>   /* TAPS: 1 = NONE (a 1-tap filter with weights 0/1: exact), 2 = LINEAR, 4 = CUBIC; ODD: 1 = the image's width is not a multiple of 4 */
__global__ __launch_bounds__(64) ROLL_OCC void
k_roll(RollBatch imgs, FixcaGeom g, FixcaPlan pl, RollTables tb, RollSched sc)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

	/* One launch covers a batch of independent images; the schedule is described at RollSched
	 * (fixca_roll.h).  Everything below is wave-uniform scalar arithmetic on kernel arguments. */
	const int bx = blockIdx.x;
	const int img = bx / sc.pool_blocks;
	const int bid = bx - img * sc.pool_blocks;
	const int tile = xcd_remap(bid, sc.pool_blocks);
	if (tile >= sc.pool_waves) return;
	const int strip = tile % pl.nstrips, chunk = tile / pl.nstrips;
	const int Y0 = g.y1 + chunk * sc.th;
	roll_tile<TAPS, P2, ODD>(lds, imgs, g, pl, tb, img, strip, Y0, min(sc.th, g.y2 - Y0), sc.edge_wb);
}

/* ======================================================================================
 * k_roll_tables: everything k_roll needs that does not depend on the pixels.
 * Workgroup b < nstrips builds strip b's header and lane weights; workgroup nstrips + k
 * builds the block descriptors and vertical weights of rows 64k .. 64k+63.
 * ====================================================================================== */
template <int TAPS>
__global__ __launch_bounds__(64) void
k_roll_tables(FixcaGeom g, FixcaPlan pl, RollTables tb)
{
	constexpr int T = TAPS;
	constexpr int BACK = (TAPS == 4) ? 1 : 0;
	const int lane = threadIdx.x;
	const int W = g.W, H = g.H, oc = pl.oc;

	if ((int)blockIdx.x >= pl.nstrips) {
		/* ---- rows ------------------------------------------------------------------------- */
		const int y = ((int)blockIdx.x - pl.nstrips) * 64 + lane;
		if (y >= H) return;
		auto row_of = [&](int yy, int c, int &rb, float *wy) {
			if constexpr (T == 1) {                      /* NONE: the rounded source row (fix-ca.c:799-809), weight 1 */
				rb = map_int(yy, g.yc, H, g.s[c], g.ys[c]);
				if (wy) wy[0] = 1.0f;
			} else {
				double yd = map_dbl(yy, g.yc, H, g.s[c], g.ys[c]);
				int y0 = (int)floor(yd);
				if (wy) tap_weights_f32<T>((float)(yd - (double)y0), wy);
				rb = y0 - BACK;
			}
		};
		auto describe = [&](int yy, int &rbR, int &rbB, int &keep, int &need, bool &reg, float (*wy)[T]) {
			int rb[2];
			keep = yy; need = yy; reg = yy > 0;
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				row_of(yy, c, rb[c], wy ? wy[c] : nullptr);
				int prev;
				row_of(max(yy - 1, 0), c, prev, nullptr);
				reg = reg && (rb[c] == prev + 1) && (rb[c] >= 0) && (rb[c] + T - 1 <= H - 1);
				keep = min(keep, rclampi(rb[c], 0, H - 1));
				need = max(need, rclampi(rb[c] + T - 1, 0, H - 1));
			}
			rbR = rb[0]; rbB = rb[1];
		};
		float wy[2][T];
		RollBlock b;
		bool reg;
		describe(y, b.rb[0], b.rb[1], b.keep, b.need1, reg, wy);
		b.need4 = b.need1;
		b.reg4 = reg ? 1 : 0;
		for (int n = 1; n < 4; ++n) {
			if (y + n > H - 1) break;
			int r0, r1, kp, nd;
			bool rg;
			describe(y + n, r0, r1, kp, nd, rg, nullptr);
			b.need4 = max(b.need4, nd);
			b.reg4 |= rg ? (1 << n) : 0;
		}
		b.pad[0] = b.pad[1] = 0;
		tb.blocks[y] = b;
		float *w = tb.roww + (size_t)y * 8;
#pragma unroll
		for (int t = 0; t < 4; ++t) {
			w[2 * t] = t < T ? wy[0][t < T ? t : 0] : 0.f;
			w[2 * t + 1] = t < T ? wy[1][t < T ? t : 0] : 0.f;
		}
		return;
	}

	/* ---- one strip ------------------------------------------------------------------------ */
	const int strip = blockIdx.x;
	const int X0 = strip * oc;
	int lo = X0;
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		if constexpr (T == 1) lo = min(lo, map_int(X0, g.xc, W, g.s[c], g.xs[c]));
		else lo = min(lo, rclampi((int)floor(map_dbl(X0, g.xc, W, g.s[c], g.xs[c])) - BACK, 0, W - 1));
	}
	const int U0 = min(lo & ~ROLL_U0_MASK, W - ROLL_WIN);

	const int nact = (min(oc, W - X0) + 3) >> 2;     /* lanes that produce output (the last may hold 1-3 pixels: RGBA8, W % 4 != 0) */
	const bool active = lane < nact;
	const int xl = X0 + 4 * min(lane, nact - 1);     /* first output column of this lane */

	/* tap columns (clamped as the reference clamps them) and weights of my 4 outputs */
	int   col[2][4][T];
	float wt[2][4][T];
	int   mlo[2] = { 1 << 30, 1 << 30 };             /* min over the strip of (tap column - x), taps with weight */
	int   mall[2] = { 1 << 30, 1 << 30 }, mhi[2] = { -(1 << 30), -(1 << 30) };   /* min and max of the same, edge pixels included */
	int   ecol[2][4];                                /* all of the output's weight is on this edge column (-1: it is not): see roll_edge_pixels */
	auto edge_column = [&](int c, int i, int column) {
		ecol[c][i] = (column == 0 || column == W - 1) ? column : -1;
		return ecol[c][i] >= 0;
	};
#pragma unroll
	for (int c = 0; c < 2; ++c)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			ecol[c][i] = -1;
			const int x = min(xl + i, W - 1);            /* a partial last lane: its outputs past the image repeat the last column's taps and are never stored */
			if constexpr (T == 1) {
				/* NONE: one "tap" of weight 1 on the rounded source column (fix-ca.c:799-809);
				 * the 0/1 weights make the fp32 filter below an exact byte gather */
				col[c][i][0] = map_int(x, g.xc, W, g.s[c], g.xs[c]);
				wt[c][i][0] = 1.0f;
				if (!edge_column(c, i, col[c][i][0])) mlo[c] = min(mlo[c], col[c][i][0] - x);
				mall[c] = min(mall[c], col[c][i][0] - x);
				mhi[c] = max(mhi[c], col[c][i][0] - x);
			} else {
				const double xd = map_dbl(x, g.xc, W, g.s[c], g.xs[c]);
				const int x0 = (int)floor(xd);
				const double fx = xd - (double)x0;
				tap_weights_f32<T>((float)fx, wt[c][i]);
				if constexpr (T == 4) {
					col[c][i][0] = max(x0 - 1, 0);                      /* fix-ca.c:1270-1298 */
					col[c][i][1] = x0;
					col[c][i][2] = min(x0 + 1, W - 1);
					col[c][i][3] = min(col[c][i][2] + 1, W - 1);
				} else {
					col[c][i][0] = x0;                                  /* fix-ca.c:1170-1177 */
					col[c][i][T - 1] = min(x0 + 1, W - 1);
				}
				/* A coordinate that sits exactly on a column (every coordinate clamped to the image
				 * edge does) has all its weight on that column; its other taps, which may lie far
				 * from the rest of the strip's, do not count.  Host and device take this decision on
				 * the fp64 fraction, where they agree bit for bit. */
				constexpr int CT = (T == 4) ? 1 : 0;                    /* the tap with weight 1 at fraction 0 */
				if (fx != 0.0) {
					mlo[c] = min(mlo[c], col[c][i][0] - x);
					mall[c] = min(mall[c], col[c][i][0] - x);
					mhi[c] = max(mhi[c], col[c][i][T - 1] - x);
				} else {
					mall[c] = min(mall[c], col[c][i][CT] - x);
					mhi[c] = max(mhi[c], col[c][i][CT] - x);
					if (!edge_column(c, i, col[c][i][CT])) mlo[c] = min(mlo[c], col[c][i][CT] - x);
#pragma unroll
					for (int t = 0; t < T; ++t) wt[c][i][t] = (t == CT) ? 1.0f : 0.0f;
				}
			}
		}
	/* strip minimum: butterfly over the wave (inactive lanes carry copies of lane nact-1) */
#pragma unroll
	for (int c = 0; c < 2; ++c) {
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) {
			mlo[c] = min(mlo[c], __shfl_xor(mlo[c], m, 64));
			mall[c] = min(mall[c], __shfl_xor(mall[c], m, 64));
			mhi[c] = max(mhi[c], __shfl_xor(mhi[c], m, 64));
		}
		mlo[c] = __builtin_amdgcn_readfirstlane(mlo[c]);
		mall[c] = __builtin_amdgcn_readfirstlane(mall[c]);
		mhi[c] = __builtin_amdgcn_readfirstlane(mhi[c]);
		if (mlo[c] == (1 << 30)) mlo[c] = 0;             /* every output of the strip is an edge pixel */
	}
	/* While the 5 window pixels reach every tap of the strip, edge pixels are outputs like any others.  Otherwise the
	 * window follows the other outputs, and the edge pixels it does not reach are flagged: bit 2i + c first pixel, bit
	 * 8 + 2i + c last pixel (the planner takes the same decisions: roll_cols_fit) */
	uint32_t fmask = 0;
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		if (mhi[c] - mall[c] <= 4) { mlo[c] = mall[c]; continue; }
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int j = ecol[c][i] - (xl + i) - mlo[c];    /* as the fold below places it */
			if (ecol[c][i] >= 0 && (j < 0 || j > 4)) fmask |= (ecol[c][i] == 0 ? 1u : 0x100u) << (2 * i + c);
		}
	}

	/* fold steps and clamped taps into 5 weights per output: pixel i + j of the lane's 8 has column
	 * X0 + mlo + 4*lane + i + j, so tap t of output i lands on j = col - x - mlo, 0 <= j <= 4
	 * for every tap with weight (the planner guarantees the range) */
	float wx[4][5][2];
	bool gen = false;
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int jj = 0; jj < 5; ++jj) {
			float a[2] = { 0.f, 0.f };
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				if (fmask & (0x101u << (2 * i + c))) continue;      /* a flagged edge pixel: no weight, rewritten after the tile's rows (roll_edge_pixels) */
#pragma unroll
				for (int t = 0; t < T; ++t)
					a[c] += (col[c][i][t] - (xl + i) - mlo[c] == jj) ? wt[c][i][t] : 0.f;
			}
			wx[i][jj][0] = a[0];
			wx[i][jj][1] = a[1];
			if (jj >= ((T == 1) ? 2 : T)) gen = gen || (a[0] != 0.f) || (a[1] != 0.f);
		}
	gen = __any(gen && active) != 0;

	/* lane weights: 40 floats in the order (i, jj, channel), as 10 chunks of 16 bytes, chunk q of
	 * all 64 lanes contiguous */
	f32x4 *cw = reinterpret_cast<f32x4 *>(tb.colw) + (size_t)strip * (10 * 64) + lane;
	const bool edge_l = __any((fmask & 0xffu) != 0 && active) != 0, edge_r = __any((fmask & 0xff00u) != 0 && active) != 0;
	int nflag[2];                                    /* flagged outputs of the strip per channel (those inside the image) */
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		nflag[c] = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) nflag[c] += (active && xl + i < W && (fmask & (0x101u << (2 * i + c)))) ? 1 : 0;
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) nflag[c] += __shfl_xor(nflag[c], m, 64);
	}
#pragma unroll
	for (int q = 0; q < 10; ++q) {
		float f[4];
#pragma unroll
		for (int e = 0; e < 4; ++e) {
			const int idx = 4 * q + e;
			f[e] = wx[idx / 10][(idx / 2) % 5][idx & 1];
		}
		cw[q * 64] = f32x4{ f[0], f[1], f[2], f[3] };
	}
	if (lane == 0) {
		RollStrip h;
		h.U0 = U0;
		h.tapR = X0 + mlo[0] - U0;
		h.tapB = X0 + mlo[1] - U0;
		h.gen = (gen ? 1 : 0) | (edge_l ? 2 : 0) | (edge_r ? 4 : 0) | (nflag[0] << 8) | (nflag[1] << 16);   /* see RollStrip */
		tb.strips[strip] = h;
	}
}

/* ---- table cache + launch -------------------------------------------------------------------- */

namespace {
/* Tables live in device memory and are read by kernels that may still be running when the next call arrives, so a
 * geometry keeps its own buffer: a small per-thread cache of entries, least recently used first out.  An entry
 * remembers the stream of its last reader; the event that proves that reader finished is recorded LAZILY: when the
 * entry is evicted, or when a launch on another stream reads it (the event is then recorded behind everything the old
 * stream has been given, which covers every reader there, and the new stream waits for it).  Round 3 recorded the event
 * behind every launch: a recorded event makes the runtime close the kernel with a release to system scope, and a
 * launch of one image (41 us) paid about 4 us for it (profiles/NOTES.md, round 4 §1).  Evicting waits for THAT
 * event only (an interactive preview that drags a slider brings a new parameter block per call: after 8 calls every
 * call evicts, and must not pay a device-wide synchronisation for it).  An entry used inside a stream capture is
 * pinned: the graph may replay the launch at any time, so its tables are never rebuilt (fixca_release_cache() frees
 * them). */
struct TableEntry {
	bool        valid = false;
	bool        pinned = false;             /* read by a captured launch: never evicted */
	bool        used = false;               /* a launch on last_stream has read the tables since last_use was last recorded */
	int         device = -1;
	FixcaGeom   g;                          /* the fields the tables depend on (same_tables) */
	int         oc = 0;
	void       *mem = nullptr;
	size_t      bytes = 0;
	RollTables  tb;
	hipEvent_t  ready = nullptr;            /* behind the table kernel */
	hipEvent_t  last_use = nullptr;         /* recorded on last_stream when somebody needs to know that its readers are done */
	hipStream_t last_stream = nullptr;
	unsigned long long tick = 0;
};
constexpr int TABLE_ENTRIES = 8;
thread_local TableEntry g_tc[TABLE_ENTRIES];
thread_local unsigned long long g_tick = 0;

/* the tables depend on these fields and on nothing else (not on the row range, not on the preview flag) */
bool same_tables(const FixcaGeom &a, const FixcaGeom &b)
{
	return a.W == b.W && a.H == b.H && a.bpp == b.bpp && a.bpc == b.bpc && a.nch == b.nch && a.xc == b.xc && a.yc == b.yc &&
	       a.interp == b.interp && a.s[0] == b.s[0] && a.s[1] == b.s[1] && a.xs[0] == b.xs[0] && a.xs[1] == b.xs[1] &&
	       a.ys[0] == b.ys[0] && a.ys[1] == b.ys[1];
}

bool stream_is_capturing(hipStream_t stream)
{
	hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(stream, &cs) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	return cs != hipStreamCaptureStatusNone;
}
}

extern "C" void ROLL_NAME(fixca_roll_release_tables)(void)
{
	for (TableEntry &e : g_tc) {
		if (e.mem) (void)hipFree(e.mem);
		if (e.ready) (void)hipEventDestroy(e.ready);
		if (e.last_use) (void)hipEventDestroy(e.last_use);
		e = TableEntry();
	}
}

/* Everything last_stream has been given so far (every launch that read entry t's tables among it) is finished once
 * t->last_use is: recorded now, not per launch.  A stream handle the caller has destroyed since (its work is finished
 * then) or one that is capturing cannot take the record: the device is drained instead (never in a steady state). */
static hipError_t tables_close_readers(TableEntry *t, bool *recorded)
{
	*recorded = false;
	if (!t->used) return hipSuccess;
	hipError_t e;
	if (!t->last_use) {
		e = hipEventCreateWithFlags(&t->last_use, hipEventDisableTiming);
		if (e != hipSuccess) return e;
	}
	if (!stream_is_capturing(t->last_stream) && hipEventRecord(t->last_use, t->last_stream) == hipSuccess) {
		*recorded = true;
	} else {
		(void)hipGetLastError();
		e = hipDeviceSynchronize();
		if (e != hipSuccess) return e;
	}
	t->used = false;
	return hipSuccess;
}

/* after a launch that reads entry t's tables on `stream` */
static hipError_t tables_mark_used(TableEntry *t, hipStream_t stream)
{
	if (stream_is_capturing(stream)) {
		t->pinned = true;
		return hipSuccess;
	}
	if (fixca_debug_env("FIXCA_EXP_EAGER_EVENT", 0, 0, 1)) {         /* timing experiment: round 3's record behind every launch */
		if (!t->last_use && hipEventCreateWithFlags(&t->last_use, hipEventDisableTiming) != hipSuccess) return hipErrorOutOfMemory;
		(void)hipEventRecord(t->last_use, stream);
	}
	if (t->used && t->last_stream != stream) {
		/* readers on the other stream may still be running: this stream's later work is put behind them, so that one
		 * record on this stream (at eviction) covers them too */
		bool recorded = false;
		hipError_t e = tables_close_readers(t, &recorded);
		if (e != hipSuccess) return e;
		if (recorded) {
			e = hipStreamWaitEvent(stream, t->last_use, 0);
			if (e != hipSuccess) return e;
		}
	}
	t->used = true;
	t->last_stream = stream;
	return hipSuccess;
}

static hipError_t ensure_tables(const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream, RollTables *out, TableEntry **entry)
{
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	for (TableEntry &t : g_tc) {
		if (t.valid && t.device == dev && t.oc == pl->oc && same_tables(t.g, *g)) {
			/* always ordered after the table kernel: a stream handle can be destroyed and handed out
			 * again, so "the stream that built them" proves nothing, and the wait costs next to nothing */
			e = hipStreamWaitEvent(stream, t.ready, 0);
			if (e != hipSuccess) return e;
			t.tick = ++g_tick;
			*out = t.tb;
			*entry = &t;
			return hipSuccess;
		}
	}
	TableEntry *slot = nullptr;                      /* a free entry, else the least recently used one that no graph holds */
	for (TableEntry &t : g_tc)
		if (!t.valid) { slot = &t; break; }
	if (!slot) {
		for (TableEntry &t : g_tc)
			if (!t.pinned && !(t.used && stream_is_capturing(t.last_stream)) && (!slot || t.tick < slot->tick)) slot = &t;
	}
	if (!slot) {
		fixca_fail(-4, "all %d table entries of this thread are held by captured launches: fixca_release_cache() after the graphs are destroyed", TABLE_ENTRIES);
		return hipErrorOutOfMemory;
	}
	TableEntry &t = *slot;
	if (t.valid) {
		/* eviction: the last kernel that read this entry's tables must have finished (and with it, on its stream, every
		 * earlier one; readers on other streams were chained in front of that stream by tables_mark_used) */
		bool recorded = false;
		e = tables_close_readers(&t, &recorded);
		if (e != hipSuccess) return e;
		if (recorded) {
			e = hipEventSynchronize(t.last_use);
			if (e != hipSuccess) return e;
		}
		e = hipEventSynchronize(t.ready);
		if (e != hipSuccess) return e;
		t.valid = false;
		t.used = false;
	}
	const size_t n_strips = (size_t)pl->nstrips;
	const size_t off_colw = (n_strips * sizeof(RollStrip) + 255) & ~(size_t)255;
	const size_t off_blocks = off_colw + n_strips * 10 * 64 * 16;
	const size_t off_roww = off_blocks + (size_t)g->H * sizeof(RollBlock);
	const size_t need = off_roww + (size_t)g->H * 8 * sizeof(float);
	if (need > t.bytes || t.device != dev) {
		/* (re)allocation happens only for a new or larger geometry: not inside the steady state
		 * of a batch or a stream capture */
		if (t.mem) (void)hipFree(t.mem);
		t.mem = nullptr;
		t.bytes = 0;
		e = hipMalloc(&t.mem, need);
		if (e != hipSuccess) return e;
		t.bytes = need;
	}
	if (!t.ready) {
		e = hipEventCreateWithFlags(&t.ready, hipEventDisableTiming);
		if (e != hipSuccess) return e;
	}
	char *base = static_cast<char *>(t.mem);
	RollTables tb;
	tb.strips = reinterpret_cast<RollStrip *>(base);
	tb.colw = reinterpret_cast<float *>(base + off_colw);
	tb.blocks = reinterpret_cast<RollBlock *>(base + off_blocks);
	tb.roww = reinterpret_cast<float *>(base + off_roww);
	const int grid = pl->nstrips + (g->H + 63) / 64;
	if (g->interp == 0)      hipLaunchKernelGGL(k_roll_tables<1>, dim3(grid), dim3(64), 0, stream, *g, *pl, tb);
	else if (g->interp == 1) hipLaunchKernelGGL(k_roll_tables<2>, dim3(grid), dim3(64), 0, stream, *g, *pl, tb);
	else                     hipLaunchKernelGGL(k_roll_tables<4>, dim3(grid), dim3(64), 0, stream, *g, *pl, tb);
	e = hipGetLastError();
	if (e != hipSuccess) return e;
	e = hipEventRecord(t.ready, stream);
	if (e != hipSuccess) return e;
	t.valid = true;
	t.pinned = false;
	t.device = dev;
	t.g = *g;
	t.oc = pl->oc;
	t.tb = tb;
	t.tick = ++g_tick;
	*out = tb;
	*entry = &t;
	return hipSuccess;
}

/* n_images images (1 .. ROLL_BATCH_MAX) in one launch */
extern "C" hipError_t
ROLL_NAME(fixca_launch_roll_batch)(const void *const *d_src, void *const *d_dst, int n_images, const FixcaGeom *g,
                        const FixcaPlan *pl, hipStream_t stream)
{
	if (n_images < 1 || n_images > ROLL_BATCH_MAX) return hipErrorInvalidValue;
	RollBatch imgs;
	for (int i = 0; i < ROLL_BATCH_MAX; ++i) {
		imgs.src[i] = static_cast<const uint8_t *>(d_src[i < n_images ? i : 0]);
		imgs.dst[i] = static_cast<uint8_t *>(d_dst[i < n_images ? i : 0]);
	}
	RollTables tb;
	TableEntry *entry = nullptr;
	hipError_t e = ensure_tables(g, pl, stream, &tb, &entry);
	if (e != hipSuccess) return e;
	size_t lds = RING_PAD_LO + (size_t)pl->ring * RING_ROW + RING_PAD_HI;
	lds += (size_t)fixca_debug_env("FIXCA_EXP_LDS_PAD", 0, 0, 128 * 1024);      /* timing experiment: fewer waves per CU */
	const bool p2 = (pl->ring & (pl->ring - 1)) == 0, odd = (g->W & 3) != 0 || fixca_debug_env("FIXCA_EXP_FORCE_ODD", 0, 0, 1) != 0;   /* (timing experiment: the any-width instantiation on an aligned image) */
	RollSched sc;
	roll_make_sched(g, pl, n_images, &sc);
	const dim3 grid((unsigned)sc.pool_blocks * (unsigned)n_images);
	auto launch = [&](auto taps_tag, auto p2_tag, auto odd_tag) -> hipError_t {
		constexpr int TP = decltype(taps_tag)::value;
		constexpr bool P2 = decltype(p2_tag)::value;
		constexpr int OD = decltype(odd_tag)::value;
		hipError_t ea = (hipError_t)fixca_dyn_lds((const void *)k_roll<TP, P2, OD>, 96 * 1024);
		if (ea != hipSuccess) return ea;
#ifdef ROLL_EXP_SPLIT
		/* timing experiment (tools/build_variant.sh NAME -DROLL_EXP_SPLIT, never the shipped library): a launch of one image as
		 * two launches of half its tiles, the second on a stream of its own between a fork and a join event: does the second
		 * half's start overlap the first half's tail?  (profiles/NOTES.md, round 4 §10) */
		if (n_images == 1 && (g->y2 - g->y1) >= 4 * sc.th) {
			static thread_local hipStream_t s2 = nullptr;
			static thread_local hipEvent_t fork_ev = nullptr, join_ev = nullptr;
			if (!s2) {
				if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return hipGetLastError();
				if (hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming) != hipSuccess) return hipGetLastError();
				if (hipEventCreateWithFlags(&join_ev, hipEventDisableTiming) != hipSuccess) return hipGetLastError();
			}
			const int chunks = (g->y2 - g->y1 + sc.th - 1) / sc.th;
			FixcaGeom g1 = *g, g2 = *g;
			g1.y2 = g2.y1 = g->y1 + ((chunks + 1) / 2) * sc.th;
			RollSched s1, s2c;
			roll_make_sched(&g1, pl, 1, &s1);
			roll_make_sched(&g2, pl, 1, &s2c);
			s1.th = s2c.th = sc.th;
			if (hipEventRecord(fork_ev, stream) != hipSuccess || hipStreamWaitEvent(s2, fork_ev, 0) != hipSuccess) return hipGetLastError();
			hipLaunchKernelGGL((k_roll<TP, P2, OD>), dim3((unsigned)s1.pool_blocks), dim3(64), lds, stream, imgs, g1, *pl, tb, s1);
			hipLaunchKernelGGL((k_roll<TP, P2, OD>), dim3((unsigned)s2c.pool_blocks), dim3(64), lds, s2, imgs, g2, *pl, tb, s2c);
			if (hipEventRecord(join_ev, s2) != hipSuccess || hipStreamWaitEvent(stream, join_ev, 0) != hipSuccess) return hipGetLastError();
			return hipGetLastError();
		}
#endif
		hipLaunchKernelGGL((k_roll<TP, P2, OD>), grid, dim3(64), lds, stream, imgs, *g, *pl, tb, sc);
		return hipGetLastError();
	};
	auto by_odd = [&](auto taps_tag, auto p2_tag) -> hipError_t {
		return odd ? launch(taps_tag, p2_tag, std::integral_constant<int, 1>{}) : launch(taps_tag, p2_tag, std::integral_constant<int, 0>{});
	};
	auto by_p2 = [&](auto taps_tag) -> hipError_t {
		return p2 ? by_odd(taps_tag, std::true_type{}) : by_odd(taps_tag, std::false_type{});
	};
	if (g->interp == 0)      e = by_p2(std::integral_constant<int, 1>{});
	else if (g->interp == 1) e = by_p2(std::integral_constant<int, 2>{});
	else                     e = by_p2(std::integral_constant<int, 4>{});
	if (e != hipSuccess) return e;
	return tables_mark_used(entry, stream);
}

extern "C" hipError_t
ROLL_NAME(fixca_launch_roll)(const void *d_src, void *d_dst, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	return ROLL_NAME(fixca_launch_roll_batch)(&d_src, &d_dst, 1, g, pl, stream);
}

}   /* namespace */
