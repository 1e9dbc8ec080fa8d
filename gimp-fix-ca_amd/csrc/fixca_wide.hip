/*
 * fixca_wide.hip — the row-rolling kernel for the sample types beyond 8 bits (SURVEY §8f row 1:
 * get_pixel / set_pixel for u16, u32, float and double samples, fix-ca.c:713-774; color_size codes
 * 2, 4, -4, -8): LINEAR and CUBIC on all of them, NONE on the 16-bit ones.  Bit-identical to the CPU oracle.
 *
 * LINEAR / CUBIC: the arithmetic is the reference's, in fp64, operation for operation (the same helpers as
 * the one-thread-per-pixel kernel k_wide, which stays as the checker and the fallback).  What changes is
 * how often each piece is computed.  The map is separable (x' depends on the column only, y' on the row
 * only), so
 *
 *   - get_pixel runs ONCE per source sample and row visit, not once per tap: a lane converts the
 *     samples of its own two window columns and the wave shares the doubles through 2 KB of LDS
 *     (32 get_pixel divisions per output sample in the reference, 1 here);
 *   - the horizontal filter of a source row (cubicY, fix-ca.c:896-909; the inner lerp of
 *     bilinear, fix-ca.c:891) is evaluated once per output column and kept in a rolling window of
 *     TAPS rows in registers: when the output row advances, the window slides by one source row
 *     (clamping is monotone, so the slide is valid at the image edges too) and only the new row is
 *     filtered; the vertical filter (cubicX, fix-ca.c:911-920) runs on the window.  Re-using the
 *     value of an expression is not a change of rounding: every operation still sees the operands
 *     the reference's would.
 *
 * 16-bit CUBIC has a second variant (k_wcol_weights): the two filters as weighted sums, with every sample
 * whose value comes within 1e-6 of a rounding boundary recomputed the reference's way (see cr_weights below
 * for the bound that makes this exact).  NONE on 16-bit samples is the same kernel with one tap and no
 * conversion (the samples are copied through LDS as they are).
 *
 * One wave per tile of `oc` = 120 output columns x `th` rows, two pixels per lane.  Each channel has its
 * own 128-column source window, so any shift fits; only the drift of the tap offset over a strip is limited
 * (the planner checks it and falls back to k_wide).  Memory is read and written with buffer instructions
 * and no branch around them: an output row's loads (the one source row each channel's window gains, and the
 * row's own pixels for green and alpha) sit in a register queue 1-2 rows ahead with counted waits; pixels of
 * 12 bytes and more move as 16-byte units in lane order (full-sector stores), the results reaching the
 * storing lane through LDS.  All tiles of a launch are resident at once (the planner sizes them to the
 * occupancy the runtime reports): the kernel is bound by instruction issue, not by memory.
 */
#include "fixca_device.h"
#include "fixca_codec.h"
#include "fixca_wide.h"
#include "fixca_move.h"
#include <mutex>
#include <type_traits>

namespace {
using fixca_codec::Codec;
using fixca_codec::cr_weights;
using fixca_codec::exact_cubic_sample;

/* ---- pixel moves -------------------------------------------------------------------------------
 * Pixels travel as dwords through raw buffer loads / stores (descriptor = the image for loads, the
 * tile's own rows for stores, so a lane that must not write is given an offset past the range and
 * the instruction stays unconditional: the loop body has no branch around a memory instruction and
 * the compiler can count its waits exactly).  A run of N dwords moves as 16/12/8/4-byte pieces:
 * buffer instructions need dword alignment only, which every pixel has when a pixel is a multiple
 * of 4 bytes or the width is even; RGB u16 rows of odd width move as halfwords. */
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3_t __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void load_dwords(rsrc_t rs, uint32_t voff, uint32_t soff, uint32_t *w)
{
	if constexpr (N >= 4) {
		const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
		w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
		load_dwords<N - 4>(rs, voff + 16, soff, w + 4);
	} else if constexpr (N == 3) {
		const u32x3_t v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, 0);
		w[0] = v.x; w[1] = v.y; w[2] = v.z;
	} else if constexpr (N == 2) {
		const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
		w[0] = v.x; w[1] = v.y;
	} else if constexpr (N == 1) {
		w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
	}
}
template <int N>
__device__ __forceinline__ void store_dwords(rsrc_t rd, uint32_t voff, uint32_t soff, const uint32_t *w)
{
	if constexpr (N >= 4) {
		__builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ w[0], w[1], w[2], w[3] }, rd, voff, soff, 2);
		store_dwords<N - 4>(rd, voff + 16, soff, w + 4);
	} else if constexpr (N == 3) {
		__builtin_amdgcn_raw_buffer_store_b96(u32x3_t{ w[0], w[1], w[2] }, rd, voff, soff, 2);
	} else if constexpr (N == 2) {
		__builtin_amdgcn_raw_buffer_store_b64(u32x2_t{ w[0], w[1] }, rd, voff, soff, 2);
	} else if constexpr (N == 1) {
		__builtin_amdgcn_raw_buffer_store_b32(w[0], rd, voff, soff, 2);
	}
}
/* NH halfwords <-> the low/high halves of consecutive dwords, from halfword `first` on */
template <int NH>
__device__ __forceinline__ void load_halves(rsrc_t rs, uint32_t voff, uint32_t soff, uint32_t *w)
{
#pragma unroll
	for (int i = 0; i < (NH + 1) / 2; ++i) w[i] = 0;
#pragma unroll
	for (int i = 0; i < NH; ++i)
		w[i / 2] |= (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff + 2 * i, soff, 0) << (16 * (i & 1));
}
template <int NH>
__device__ __forceinline__ void store_halves(rsrc_t rd, uint32_t voff, uint32_t soff, const uint32_t *w, int first)
{
#pragma unroll
	for (int i = 0; i < NH; ++i) {
		const int k = first + i;
		__builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w[k / 2] >> (16 * (k & 1))), rd, voff + 2 * i, soff, 2);
	}
}

/* sample k of a run of pixels held as dwords */
template <typename S>
__device__ __forceinline__ S get_sample(const uint32_t *w, int k)
{
	if constexpr (sizeof(S) == 2) return (S)(w[k / 2] >> (16 * (k & 1)));
	else if constexpr (sizeof(S) == 4) return __builtin_bit_cast(S, w[k]);
	else return __builtin_bit_cast(S, (unsigned long long)w[2 * k] | ((unsigned long long)w[2 * k + 1] << 32));
}
template <typename S>
__device__ __forceinline__ void set_sample(uint32_t *w, int k, S v)
{
	if constexpr (sizeof(S) == 2) w[k / 2] = (k & 1) ? ((w[k / 2] & 0x0000FFFFu) | ((uint32_t)v << 16)) : ((w[k / 2] & 0xFFFF0000u) | (uint32_t)v);
	else if constexpr (sizeof(S) == 4) w[k] = __builtin_bit_cast(uint32_t, v);
	else { const unsigned long long b = __builtin_bit_cast(unsigned long long, v); w[2 * k] = (uint32_t)b; w[2 * k + 1] = (uint32_t)(b >> 32); }
}

/* clip_d (fix-ca.c:873-880) ahead of set_pixel.  For the integer sample types the two compares and selects are a max
 * and a min: the operand is finite, and the sign of a zero does not survive the conversion. */
template <bool INTEGER>
__device__ __forceinline__ double clip_for_put(double d)
{
	if constexpr (INTEGER) return __builtin_fmin(__builtin_fmax(d, 0.0), 1.0);
	else return clip01(d);
}

template <int TAPS>
__device__ __forceinline__ double hfilter(const double (&p)[TAPS], double fx)
{
	if constexpr (TAPS == 4) return cr_exact(p[0], p[1], p[2], p[3], fx);              /* cubicY, fix-ca.c:905-907 */
	else return __dadd_rn(p[0], __dmul_rn(fx, __dsub_rn(p[1], p[0])));                   /* fix-ca.c:891, inner */
}
template <int TAPS>
__device__ __forceinline__ double vfilter(const double (&h)[TAPS], double fy)
{
	if constexpr (TAPS == 4) return cr_exact(h[0], h[1], h[2], h[3], fy);              /* cubicX, fix-ca.c:916-918 */
	else return __dadd_rn(__dmul_rn(__dsub_rn(1.0, fy), h[0]), __dmul_rn(fy, h[1]));     /* fix-ca.c:891, outer */
}

/* tap t of the window that starts at integer coordinate c0, clamped as the reference clamps it:
 * cubic  max(c0-1,0), c0, min(c0+1,n-1), min(min(c0+1,n-1)+1,n-1)  (fix-ca.c:1218-1256, 1270-1298)
 * linear c0, (c0 == n-1) ? c0 : c0+1                               (fix-ca.c:1149-1158, 1170-1177)
 * both are clamp(c0 - BACK + t, 0, n-1) because 0 <= c0 <= n-1 */
template <int TAPS>
__device__ __forceinline__ int tap_index(int c0, int t, int n)
{
	constexpr int BACK = (TAPS == 4) ? 1 : 0;
	return max(0, min(c0 - BACK + t, n - 1));
}


/*
 * Q = slots of the register queue of loads; Q - 1 rows of loads are in flight ahead of the row being
 * computed.  Every output row needs, per channel, the one source row its sliding window gains (row
 * tap_index(y0, T-1)) and its own pixels (green and alpha): those 3 loads are issued Q - 1 rows early,
 * always (when the window does not move the row is fetched anyway and not used), so the steady-state
 * body is straight-line code.  Row r uses slot r mod Q in place and the slot is refilled when row
 * r + 1 starts: with the loop unrolled Q times every slot keeps its registers across the back edge
 * (a slot refilled while its old value is still in use would need a copy there, and a copy of a
 * register with a load in flight is a full s_waitcnt).
 */
#ifdef WIDE_WAVES_PER_EU
#define WIDE_OCC __attribute__((amdgpu_waves_per_eu(WIDE_WAVES_PER_EU)))
#else
#define WIDE_OCC
#endif
#ifndef WIDE_PAIR_READS
#define WIDE_PAIR_READS 1      /* 0: each column's window read on its own (rounds 2-3; an A/B build option) */
#endif
#ifndef WIDE_WY_TABLE
#define WIDE_WY_TABLE 1        /* 0: every lane computes the row's vertical weights itself (rounds 2-3; an A/B build option) */
#endif
template <typename S, int NCH, int TAPS, int Q, bool ODDW, bool FAST>
__device__ __forceinline__ void
wcol_tile(const S *__restrict__ src, S *__restrict__ dst, const FixcaGeom &g, const WidePlan &wp, int block)
{
	typedef Codec<S> C;
	constexpr int T = TAPS, BACK = (T == 4) ? 1 : 0;
	static_assert(!FAST || (sizeof(S) == 2 && TAPS == 4), "the weight form is for 16-bit samples, CUBIC");
	constexpr int SB = (int)sizeof(S);
	constexpr int PB = NCH * SB;                          /* bytes per pixel                       */
	constexpr int NW = 2 * PB / 4;                        /* dwords per pixel pair                 */
	constexpr bool PAIRWIN = SB == 2;                     /* window rows fetched as whole pixel pairs (16 or 12 bytes a lane) */
	constexpr int RW = PAIRWIN ? NW : 2 * SB / 4;         /* dwords of one window fetch            */
	constexpr bool ALIGNED = PB % 4 == 0;                 /* every pixel starts on a dword         */
	/* RGBA pixels of 16 or 32 bytes: the strip's bytes of a row move as 16-byte units, unit u by lane u % 64 in
	 * instruction u / 64, so that every load and store instruction covers 1 KB of consecutive bytes.  (A lane's own
	 * two pixels are 32 or 64 bytes: stored by the lane that computed them, each instruction writes a quarter or a half
	 * of every 64-byte sector; measured on 6000x4000 cubic: WRITE_SIZE 1.36x / 2.3x the image, and 253 -> 148 us
	 * (float), 659 -> 228 us (double) with the stores taken out.)  The red and blue results reach the lane that
	 * stores them through LDS. */
	constexpr bool FLAT = NCH == 4 && SB >= 4;
	/* RGB pixels of 12 or 24 bytes move as 16-byte units too; a unit then holds parts of two pixels, so the whole output
	 * row of the strip is put together in LDS: the units as loaded, my four results over them, the units read back. */
	constexpr bool STAGED = NCH == 3 && SB >= 4;
	constexpr bool UNITS = FLAT || STAGED;
	constexpr int NU = UNITS ? (WIDE_OC * PB / 16 + 63) / 64 : 1;   /* 16-byte units per lane and row */
	constexpr int IQ = UNITS ? 4 * NU : NW;               /* dwords a queue slot keeps of the output row */
	__shared__ uint32_t rowbuf[STAGED ? NU * 64 * 4 : 1];
	__shared__ double hbuf[2][WIDE_WIN];                  /* one source row's converted samples per channel */
	__shared__ double row_fy[2][WIDE_TH_MAX];             /* per output row of the tile: fraction ...       */
	__shared__ int    row_y0[2][WIDE_TH_MAX];             /* ... and floor of the source coordinate          */
	/* FAST: the four vertical Catmull-Rom weights of every output row of the tile, per channel.  They are the same for
	 * every lane: computed once per tile by the lane that computes the row's coordinate, instead of by all 64 lanes in
	 * every row (32 of the 70 fp64 instructions a wave spent per row; NOTES round 4 §6).  cr_weights of the same
	 * fraction: the same bits. */
	__shared__ double row_wy[FAST ? 2 * WIDE_TH_MAX * 4 : 1];

	const int lane = threadIdx.x;
	const int W = g.W, H = g.H;
	const int tile = xcd_remap(block, wp.nblocks);
	if (tile >= wp.nwaves) return;
	const int strip = tile % wp.nstrips, chunk = tile / wp.nstrips;
	const int X0 = strip * wp.oc;
	const int Y0 = g.y1 + chunk * wp.th;
	const int nrows = min(wp.th, g.y2 - Y0);
	const uint32_t pitchB = (uint32_t)W * PB;             /* the planner keeps the image below 4 GB */
	const int xe = min(X0 + wp.oc, W);                    /* one past my strip's last output column */
	const int x = X0 + 2 * lane;                          /* my two output columns: x, x + 1 */
	const bool act0 = x < xe, act1 = x + 1 < xe;
	/* Pixel pairs are moved as dwords.  With 6-byte pixels and an odd width (ODDW) the rows, and with them the pairs, start
	 * on halfwords: gfx950's buffer loads and stores take such addresses as they come (tools/skeletons/unaligned_probe.hip;
	 * rounds 2-3 moved those rows halfword by halfword: 5999-wide RGB u16 LINEAR / CUBIC 26.1 / 25.5 % against 55.0 / 41.7 %
	 * at 6000) */
	constexpr bool al4 = true;

	const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<S *>(src), 0, (uint32_t)H * pitchB, 0x00020000);
	const uint32_t dst_range = (uint32_t)nrows * pitchB;  /* also the offset that makes a lane's store a no-op */
	const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(dst) + (size_t)Y0 * pitchB, 0, dst_range, 0x00020000);
	/* UNITS: byte offset in a row of this lane's first unit; its stores' offsets (out of the row for units that are not
	 * wholly inside the strip: = the row descriptor's size, out of range, and no 32-bit wrap when the access size is added) */
	const uint32_t ubase = (uint32_t)X0 * PB + 16u * lane;
	uint32_t ustore[NU];
#pragma unroll
	for (int i = 0; i < NU; ++i) ustore[i] = (lane + 64 * i) * 16 + 16 <= (xe - X0) * PB ? ubase + 1024u * i : pitchB;
	/* the strip's bytes need not be whole units (RGB, last strip): the unit they end in, and how many dwords of it are the strip's */
	const int tail_unit = (xe - X0) * PB / 16, tail_dwords = (xe - X0) * PB % 16 / 4;

	/* ---- per-lane column constants ------------------------------------------------------------
	 * Channel c's window is source columns Uw .. Uw + 127, clamped into the image when they are FETCHED
	 * (hbuf[i] = sample at clamp(Uw + i, 0, W - 1)): the reference's clamped taps
	 * clamp(c0 - BACK + t, 0, W - 1) are then hbuf[c0 - BACK - Uw + t] with no clamp, 4 consecutive doubles. */
	/* what the wave exchanges through LDS: get_pixel's doubles for the filters, the samples themselves for NONE */
	typedef typename std::conditional<T == 1, S, double>::type X;
	typedef const __attribute__((address_space(3))) X *lds_xp;
	X *const xbuf = reinterpret_cast<X *>(&hbuf[0][0]);
	lds_xp   hb[2][2];                                    /* tap 0 in the exchange buffer, per channel and pixel */
	uint32_t wcol[2][2];                                  /* byte offset in a row of the sample (or pair) this lane fetches */
	int      sel[2][2];                                   /* PAIRWIN: which pixel of the fetched pair is window column 2 * lane + j */
	double   fx[2][2];
	double   wx[FAST ? 2 : 1][2][4];                      /* FAST: the horizontal weights of my two columns */
#pragma unroll
	for (int c = 0; c < 2; ++c) {
		/* T == 1 (NONE): the one tap is the nearest source column, scale() of fix-ca.c:799-809 */
		const int first_tap = (T == 1) ? map_int(X0, g.xc, W, g.s[c], g.xs[c]) : (int)floor(map_dbl(X0, g.xc, W, g.s[c], g.xs[c])) - BACK;
		const int Uw = __builtin_amdgcn_readfirstlane(first_tap & ~1);   /* the strip's first column has the lowest tap (the map is monotone); may be -2 */
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const int xx = min(x + j, W - 1);
			int c0;
			if constexpr (T == 1) { c0 = map_int(xx, g.xc, W, g.s[c], g.xs[c]); fx[c][j] = 0.0; }
			else {
				const double xd = map_dbl(xx, g.xc, W, g.s[c], g.xs[c]);
				c0 = (int)floor(xd);
				fx[c][j] = __dsub_rn(xd, (double)c0);
				if constexpr (FAST) cr_weights(fx[c][j], wx[c][j]);
			}
			const int b = max(0, min(c0 - BACK - Uw, WIDE_WIN - T));              /* the planner guarantees the range */
			hb[c][j] = (lds_xp)&xbuf[c * WIDE_WIN] + b;
		}
		/* which two window columns this lane fetches: a pair (2 lane, 2 lane + 1) where the pair is one 12/16-byte load
		 * (16-bit samples); lane and lane + 64 for the sample loads of the wider types, so that a load instruction's 64
		 * samples lie in 64 consecutive pixels (half the cache lines of the pair order) */
		constexpr bool SPLIT = !PAIRWIN;
		const int wi0 = SPLIT ? lane : 2 * lane, wi1 = SPLIT ? lane + 64 : 2 * lane + 1;
		const int col0 = max(0, min(Uw + wi0, W - 1)), col1 = max(0, min(Uw + wi1, W - 1));
		if constexpr (PAIRWIN && al4) {
			const int bp = max(0, min(Uw + 2 * lane, W - 2));
			wcol[c][0] = wcol[c][1] = (uint32_t)bp * PB;
			sel[c][0] = col0 - bp; sel[c][1] = col1 - bp;
		} else {
			wcol[c][0] = (uint32_t)col0 * PB + 2 * c * SB;
			wcol[c][1] = (uint32_t)col1 * PB + 2 * c * SB;
			sel[c][0] = sel[c][1] = 0;
		}
	}
	struct Raw { uint32_t w[RW]; };
	/* the window columns of this lane in source row r, channel c (loads only: nothing waits here) */
	auto fetch = [&](int c, int r, Raw &raw) {
		const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)r * pitchB);
		if constexpr (PAIRWIN) {
			if constexpr (al4) load_dwords<NW>(rs, wcol[c][0], so, raw.w);
			else {
				raw.w[0] = __builtin_amdgcn_raw_buffer_load_b16(rs, wcol[c][0], so, 0);
				raw.w[1] = __builtin_amdgcn_raw_buffer_load_b16(rs, wcol[c][1], so, 0);
			}
		} else {
			load_dwords<SB / 4>(rs, wcol[c][0], so, raw.w);
			load_dwords<SB / 4>(rs, wcol[c][1], so, raw.w + SB / 4);
		}
	};
	/* get_pixel on them, left in LDS for the wave (hbuf[c][i] = window column i of the row) */
	auto convert = [&](int c, const Raw &raw) {
		S s[2];
		if constexpr (PAIRWIN) {
			if constexpr (al4) {
				const S p0 = get_sample<S>(raw.w, 2 * c), p1 = get_sample<S>(raw.w, NCH + 2 * c);
				s[0] = sel[c][0] ? p1 : p0; s[1] = sel[c][1] ? p1 : p0;
				/* every dword the load writes stays allocated until here: the compiler would otherwise hand the ones this
				 * channel does not read to other values while the load is in flight, and wait for the load to do so */
#pragma unroll
				for (int i = 0; i < RW; ++i) asm volatile("" :: "v"(raw.w[i]));
			} else { s[0] = (S)raw.w[0]; s[1] = (S)raw.w[1]; }
		} else { s[0] = get_sample<S>(raw.w, 0); s[1] = get_sample<S>(raw.w, 1); }
		const int i0 = PAIRWIN ? 2 * lane : lane, i1 = PAIRWIN ? 2 * lane + 1 : lane + 64;      /* the window columns fetched, see wcol */
		if constexpr (T == 1) { xbuf[c * WIDE_WIN + i0] = s[0]; xbuf[c * WIDE_WIN + i1] = s[1]; }
		else if constexpr (FAST) { xbuf[c * WIDE_WIN + i0] = (double)s[0]; xbuf[c * WIDE_WIN + i1] = (double)s[1]; }
		else { xbuf[c * WIDE_WIN + i0] = C::get(s[0]); xbuf[c * WIDE_WIN + i1] = C::get(s[1]); }
	};
	/* the horizontal filter of my two columns on the row in hbuf[c].  One wave: its LDS operations complete in order,
	 * the reads see the writes of convert() */
	/* Both of a lane's columns usually start their windows one column apart (no scale step between them, no clamping at an
	 * image edge): then the two windows are T + 1 consecutive samples, read once (5 LDS reads of 8 bytes per channel and row
	 * instead of 8 for CUBIC).  Whether that holds for every lane of the strip is decided once per tile (wave-uniform).
	 * Counters of the 16-bit weight form before this: the CU's LDS busy 69 % of the time, 39 % of that in bank conflicts of
	 * the 8-byte reads (lanes 16 apart share banks); NOTES round 4 §7.  It changes the compiler's register count by up to
	 * +-20, and with it the waves per SIMD, so it is on only where it measured a win on both entries: RGB float CUBIC
	 * (batch 46.3 -> 50.9 %, one image per launch 41.6 -> 44.2 %) and the RGBA 16-bit weight form (47.3 -> 48.6 % one image
	 * per launch, +-0 batched).  Elsewhere (profiles/r04_wcol_pair_reads.txt): RGB u16 LINEAR +1.9 batched but -2.5 alone,
	 * RGBA u16 LINEAR -1.6, RGB u32 CUBIC -0.9 (172 registers instead of 168: a wave per SIMD less), the RGB weight form
	 * -2.3 (spills), RGBA float / u32 CUBIC -0.7 / +0.5, double +-0. */
	constexpr bool PAIR = WIDE_PAIR_READS && T == 4 && ((FAST && NCH == 4) || (std::is_same<S, float>::value && NCH == 3));
	bool pair_ok[2];
#pragma unroll
	for (int c = 0; c < 2; ++c) pair_ok[c] = PAIR && __all(hb[c][1] == hb[c][0] + 1) != 0;
	auto hrow = [&](int c, double (&out)[2]) {
		if constexpr (T > 1) {
			if (pair_ok[c]) {
				double q[T + 1];
#pragma unroll
				for (int t = 0; t < T + 1; ++t) q[t] = hb[c][0][t];
#pragma unroll
				for (int j = 0; j < 2; ++j) {
					double p[T];
#pragma unroll
					for (int t = 0; t < T; ++t) p[t] = q[j + t];
					if constexpr (FAST) out[j] = fma(wx[c][j][3], p[3], fma(wx[c][j][2], p[2], fma(wx[c][j][1], p[1], wx[c][j][0] * p[0])));
					else out[j] = hfilter<T>(p, fx[c][j]);
				}
				return;
			}
#pragma unroll
			for (int j = 0; j < 2; ++j) {
				double p[T];
#pragma unroll
				for (int t = 0; t < T; ++t) p[t] = hb[c][j][t];
				if constexpr (FAST) out[j] = fma(wx[c][j][3], p[3], fma(wx[c][j][2], p[2], fma(wx[c][j][1], p[1], wx[c][j][0] * p[0])));
				else out[j] = hfilter<T>(p, fx[c][j]);
			}
		}
	};
	/* The row coordinates are the same for every lane: lane L computes those of row Y0 + L (and
	 * Y0 + 64 + L) once per tile instead of every lane computing all of them every row. */
	for (int i = lane; i < nrows; i += 64) {
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			if constexpr (T == 1) {
				row_y0[c][i] = map_int(Y0 + i, g.yc, H, g.s[c], g.ys[c]);     /* nearest source row (fix-ca.c:1105-1106) */
				row_fy[c][i] = 0.0;
			} else {
				const double yd = map_dbl(Y0 + i, g.yc, H, g.s[c], g.ys[c]);
				const int y0 = (int)floor(yd);
				row_y0[c][i] = y0;
				row_fy[c][i] = __dsub_rn(yd, (double)y0);
				if constexpr (FAST) {
					double w[4];
					cr_weights(row_fy[c][i], w);
#pragma unroll
					for (int t = 0; t < 4; ++t) row_wy[(c * WIDE_TH_MAX + i) * 4 + t] = w[t];
				}
			}
		}
	}

	/* ---- the queue of loads in flight ----------------------------------------------------------
	 * A slot holds everything output row r needs from memory and from the tables: the loads' registers, and y0 / fy
	 * in scalar registers.  The table entries themselves are read from LDS one issue ahead (tab_*), so that no row
	 * waits for an LDS round trip before it can form an address. */
	Raw      pend[Q][2];                                  /* per channel: the source row output row r's window gains */
	uint32_t idtq[Q][IQ];                                 /* my two pixels of output row r (green and alpha are copied), or my units of it */
	int      sy0[Q][2];                                   /* floor of the source row coordinate, per channel (scalar) */
	double   sfy[Q][2];                                   /* its fraction (scalar) */
	int      tab_y0[2];
	double   tab_fy[2];
	const uint32_t xoffB = (uint32_t)min(x, W - 2) * PB;  /* W >= 2 is guaranteed by the planner */
	auto table_read = [&](int rr) {
		rr = min(rr, nrows - 1);
#pragma unroll
		for (int c = 0; c < 2; ++c) { tab_y0[c] = row_y0[c][rr]; tab_fy[c] = row_fy[c][rr]; }
	};
	auto issue = [&](int k, int rr) {                      /* rows are issued in order: tab_* holds row rr's entries */
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			sy0[k][c] = __builtin_amdgcn_readfirstlane(tab_y0[c]);
			const unsigned long long fb = __builtin_bit_cast(unsigned long long, tab_fy[c]);
			const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)fb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(fb >> 32));
			sfy[k][c] = __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
		}
		table_read(rr + 1);
		rr = min(rr, nrows - 1);                          /* past the tile: a harmless repeat, so that the body has no branch */
#pragma unroll
		for (int c = 0; c < 2; ++c) fetch(c, tap_index<T>(sy0[k][c], T - 1, H), pend[k][c]);
		const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)(Y0 + rr) * pitchB);
		if constexpr (UNITS) {
			/* a descriptor of the row alone: units past the end of the row (last strip) read as 0 instead of running into
			 * the next row, or past the image on its last row */
			const rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(reinterpret_cast<const uint8_t *>(src)) + (size_t)so, 0, pitchB, 0x00020000);
#pragma unroll
			for (int i = 0; i < NU; ++i) load_dwords<4>(rrow, ubase + 1024u * i, 0, idtq[k] + 4 * i);
		} else if constexpr (al4) load_dwords<NW>(rs, xoffB, so, idtq[k]);
		else load_halves<2 * NCH>(rs, xoffB, so, idtq[k]);
	};
	double hh[2][2][T];                                   /* the horizontally filtered window rows of my two columns, per channel */
	int    y0prev[2];
	/* Row 0's windows: all 2 x T source rows are requested at once (one memory latency for the tile's start, not
	 * 2 x T of them one after the other; every wave of the launch is here at the same time). */
	if constexpr (T > 1) {
		Raw first[2][T];
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			y0prev[c] = __builtin_amdgcn_readfirstlane(row_y0[c][0]);
#pragma unroll
			for (int t = 0; t < T; ++t) fetch(c, tap_index<T>(y0prev[c], t, H), first[c][t]);
		}
#pragma unroll
		for (int c = 0; c < 2; ++c)
#pragma unroll
			for (int t = 0; t < T; ++t) {
				double nw[2];
				convert(c, first[c][t]);
				hrow(c, nw);
				hh[c][0][t] = nw[0]; hh[c][1][t] = nw[1];
			}
	}
	table_read(0);
#pragma unroll
	for (int k = 0; k + 1 < Q; ++k) issue(k, k);

	for (int r0 = 0; r0 < nrows; r0 += Q) {
#pragma unroll
		for (int k = 0; k < Q; ++k) {
			const int r = r0 + k;                          /* may run up to Q - 1 rows past the tile: those rows store nothing */
			const bool live = r < nrows;
			issue((k + Q - 1) % Q, r + Q - 1);            /* into the slot row r - 1 used */
			const Raw (&cur)[2] = pend[k];
			uint32_t px[IQ];
#pragma unroll
			for (int i = 0; i < IQ; ++i) px[i] = idtq[k][i];
			const int y0[2] = { sy0[k][0], sy0[k][1] };

			S outv[2][2];                                  /* [channel][pixel]: the samples to store */
			if constexpr (T == 1) {
				/* NONE (fix-ca.c:1100-1121): the sample of the nearest source position, copied as it is.  The queue holds
				 * the source row of each channel; every lane picks its two samples out of the row the wave laid out in LDS. */
				convert(0, cur[0]);
				convert(1, cur[1]);
#pragma unroll
				for (int c = 0; c < 2; ++c)
#pragma unroll
					for (int j = 0; j < 2; ++j) outv[c][j] = hb[c][j][0];
			} else {
				if (__builtin_expect(y0[0] == y0prev[0] + 1 && y0[1] == y0prev[1] + 1, 1)) {
					/* the usual row: both windows slide by one source row (row t of the new window is row t + 1 of the old
					 * one).  Both channels' samples go to LDS before either is read back: one round trip, not two. */
					convert(0, cur[0]);
					convert(1, cur[1]);
#pragma unroll
					for (int c = 0; c < 2; ++c) {
						double nw[2];
						hrow(c, nw);
#pragma unroll
						for (int j = 0; j < 2; ++j) {
#pragma unroll
							for (int t = 0; t + 1 < T; ++t) hh[c][j][t] = hh[c][j][t + 1];
							hh[c][j][T - 1] = nw[j];
						}
					}
				} else {
#pragma unroll
					for (int c = 0; c < 2; ++c) {
						if (y0[c] == y0prev[c] + 1) {
							double nw[2];
							convert(c, cur[c]);
							hrow(c, nw);
#pragma unroll
							for (int j = 0; j < 2; ++j) {
#pragma unroll
								for (int t = 0; t + 1 < T; ++t) hh[c][j][t] = hh[c][j][t + 1];
								hh[c][j][T - 1] = nw[j];
							}
						} else if (y0[c] != y0prev[c]) {
							/* a scale step that skips a row: rebuild the window */
#pragma unroll
							for (int t = 0; t < T; ++t) {
								Raw raw;
								double nw[2];
								fetch(c, tap_index<T>(y0[c], t, H), raw);
								convert(c, raw);
								hrow(c, nw);
								hh[c][0][t] = nw[0]; hh[c][1][t] = nw[1];
							}
						}                                       /* y0 == y0prev: the same window again */
					}
				}
				y0prev[0] = y0[0]; y0prev[1] = y0[1];
				if constexpr (FAST) {
					unsigned long long redo_m[4];               /* [2 * c + j]: lanes that must recompute that sample (scalar registers) */
					const int rr = min(r, nrows - 1);           /* wave-uniform: every lane reads the same 32 bytes (an LDS broadcast) */
#pragma unroll
					for (int c = 0; c < 2; ++c) {
						double wy[4];
#if WIDE_WY_TABLE
#pragma unroll
						for (int t = 0; t < 4; ++t) wy[t] = row_wy[(c * WIDE_TH_MAX + rr) * 4 + t];
#else
						(void)rr;
						cr_weights(sfy[k][c], wy);
#endif
#pragma unroll
						for (int j = 0; j < 2; ++j) {
							const double xv = fma(wy[3], hh[c][j][3], fma(wy[2], hh[c][j][2], fma(wy[1], hh[c][j][1], wy[0] * hh[c][j][0])));
							bool near;
							outv[c][j] = (S)fixca_codec::fast_round_u16(xv, near);
							redo_m[2 * c + j] = __builtin_amdgcn_ballot_w64(near);
						}
					}
					if (__builtin_expect((redo_m[0] | redo_m[1] | redo_m[2] | redo_m[3]) != 0, 0)) {
						/* a few samples per image: one copy of the recomputation, the flagged samples of a lane in turn */
						uint32_t redo = 0;                      /* bit 2 * c + j: recompute that sample */
#pragma unroll
						for (int q = 0; q < 4; ++q) redo |= (uint32_t)((redo_m[q] >> lane) & 1ull) << q;
						while (redo) {
							const int q = __builtin_ctz(redo);
							redo &= redo - 1;
							const S e = exact_cubic_sample<S, NCH>(src, g, min(x + (q & 1), W - 1), Y0 + min(r, nrows - 1), q >> 1);
#pragma unroll
							for (int c = 0; c < 2; ++c)
#pragma unroll
								for (int j = 0; j < 2; ++j) outv[c][j] = (q == 2 * c + j) ? e : outv[c][j];
						}
					}
				} else {
					double v[2][2];                            /* [channel][pixel] */
#pragma unroll
					for (int c = 0; c < 2; ++c)
#pragma unroll
						for (int j = 0; j < 2; ++j) v[c][j] = clip_for_put<C::INTEGER>(vfilter<T>(hh[c][j], sfy[k][c]));
#pragma unroll
					for (int c = 0; c < 2; ++c)
#pragma unroll
						for (int j = 0; j < 2; ++j) outv[c][j] = C::put(v[c][j]);
				}
			}
			/* green and alpha are copied (fix-ca.c:1098), red and blue replaced.
			 * The row offset is added into the per-lane offset and the stores' scalar offset is the constant 0.  With the
			 * offset in a scalar register the compiler places a VALU write to a 16-byte store's data registers straight
			 * behind the store (it takes the two issue cycles of such a store as the wait state that overwrite needs);
			 * on this device that is not enough: the last lanes of each group of 16 were stored with the new value of the
			 * register (seen as 36 wrong samples in a 6000x4000 RGBA double image).  With a constant offset the compiler
			 * inserts the wait state itself. */
			const uint32_t so = (uint32_t)r * pitchB;
			if constexpr (FLAT) {
				/* my four results as samples, pixel-major, into LDS (the converted-sample buffer is free until the next row's
				 * convert(); a wave's LDS operations complete in order); then each lane picks up the red and blue of the units
				 * it holds: unit = pixel for 16-byte pixels; for 32-byte pixels even units are (R, G), odd ones (B, A) */
				/* hbuf is declared double and convert() stores doubles into it; here the same bytes are written and read as
				 * samples.  The compiler orders memory accesses by type, so without the fences it may move the next row's
				 * double stores above this row's sample reads (or these sample stores above the filters' double reads):
				 * the empty asm statements pin the order, as in the STAGED path; the hardware completes a wave's LDS
				 * operations in order. */
				asm volatile("" ::: "memory");
				S *res = reinterpret_cast<S *>(&hbuf[0][0]);
				res[4 * lane + 0] = outv[0][0]; res[4 * lane + 1] = outv[1][0];
				res[4 * lane + 2] = outv[0][1]; res[4 * lane + 3] = outv[1][1];
				asm volatile("" ::: "memory");
				const rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(dst) + (size_t)(Y0 + min(r, nrows - 1)) * pitchB, 0,
				                                                      live ? pitchB : 0u, 0x00020000);
#pragma unroll
				for (int i = 0; i < NU; ++i) {
					/* read back with the type they were written with (the compiler orders accesses by type) */
					const int u = lane + 64 * i;
					if constexpr (SB == 4) { set_sample<S>(px, 4 * i, res[2 * u]); set_sample<S>(px, 4 * i + 2, res[2 * u + 1]); }
					else set_sample<S>(px, 2 * i, res[u]);
					store_dwords<4>(rrow, ustore[i], 0, px + 4 * i);
				}
				asm volatile("" ::: "memory");                  /* the read-back is done before the next row's convert() writes */
			} else if constexpr (STAGED) {
				const rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t *>(dst) + (size_t)(Y0 + min(r, nrows - 1)) * pitchB, 0,
				                                                      live ? pitchB : 0u, 0x00020000);
				/* the three steps touch the same LDS bytes as dwords and as samples: the empty asm statements keep the compiler
				 * from reordering them (it orders accesses by type); the wave's LDS operations themselves complete in order */
#pragma unroll
				for (int i = 0; i < NU; ++i)
#pragma unroll
					for (int d = 0; d < 4; ++d) rowbuf[(lane + 64 * i) * 4 + d] = px[4 * i + d];
				asm volatile("" ::: "memory");
				S *rowS = reinterpret_cast<S *>(&rowbuf[0]);
				rowS[(2 * lane) * 3 + 0] = outv[0][0]; rowS[(2 * lane) * 3 + 2] = outv[1][0];
				rowS[(2 * lane + 1) * 3 + 0] = outv[0][1]; rowS[(2 * lane + 1) * 3 + 2] = outv[1][1];
				asm volatile("" ::: "memory");
#pragma unroll
				for (int i = 0; i < NU; ++i) {
#pragma unroll
					for (int d = 0; d < 4; ++d) px[4 * i + d] = rowbuf[(lane + 64 * i) * 4 + d];
					store_dwords<4>(rrow, ustore[i], 0, px + 4 * i);
				}
				asm volatile("" ::: "memory");
				if (tail_dwords) {                          /* wave-uniform: the last strip of an image whose width is not a multiple of 4 */
#pragma unroll
					for (int i = 0; i < NU; ++i)
						if (tail_unit / 64 == i) {
							const uint32_t o = (lane == tail_unit % 64) ? ubase + 1024u * i : pitchB;
#pragma unroll
							for (int d = 0; d < 3; ++d)
								if (d < tail_dwords) store_dwords<1>(rrow, o == pitchB ? pitchB : o + 4 * d, 0, px + 4 * i + d);
						}
				}
			} else
			if constexpr (!ODDW) {
				set_sample<S>(px, 0, outv[0][0]); set_sample<S>(px, 2, outv[1][0]);
				set_sample<S>(px, NCH, outv[0][1]); set_sample<S>(px, NCH + 2, outv[1][1]);
				store_dwords<NW>(rd, (act0 && live) ? (uint32_t)x * PB + so : dst_range, 0, px);
			} else {
				/* odd width: the last column's lane has one pixel (x == W - 1; px was loaded from column W - 2, so it is px's second) */
				const bool lone = act0 && !act1;
				if (lone) {
#pragma unroll
					for (int i = 0; i < NCH; ++i) set_sample<S>(px, i, get_sample<S>(px, NCH + i));
				}
				set_sample<S>(px, 0, outv[0][0]); set_sample<S>(px, 2, outv[1][0]);
				set_sample<S>(px, NCH, outv[0][1]); set_sample<S>(px, NCH + 2, outv[1][1]);
				const uint32_t o0 = (act0 && live) ? (uint32_t)x * PB + so : dst_range, o1 = (act1 && live) ? (uint32_t)(x + 1) * PB + so : dst_range;
				if constexpr (ALIGNED) {
					store_dwords<PB / 4>(rd, o0, 0, px);
					store_dwords<PB / 4>(rd, o1, 0, px + PB / 4);
				} else {
					/* 6-byte pixels on halfword boundaries: the pair as one 12-byte store, the row's last pixel alone as 4 + 2 bytes */
					static_assert(ALIGNED || PB == 6, "RGB u16 is the one format whose pixels are not whole dwords");
					const uint32_t op = (act1 && live) ? o0 : dst_range, ol = (lone && live) ? o0 : dst_range;
					store_dwords<3>(rd, op, 0, px);
					store_dwords<1>(rd, ol, 0, px);
					__builtin_amdgcn_raw_buffer_store_b16((unsigned short)px[1], rd, ol == dst_range ? dst_range : ol + 4, 0, 2);
					(void)o1;
				}
			}
		}
	}
}

template <typename S, int NCH, int TAPS, int Q, bool ODDW>
__global__ __launch_bounds__(64) WIDE_OCC void
k_wcol(MoveBatch imgs, FixcaGeom g, WidePlan wp)
{
	/* up to MOVE_BATCH_MAX images of one geometry per launch: block b works on tile b % nblocks of image b / nblocks, so
	 * the tiles of the next image take the slots the first finishers of this one leave */
	const int img = blockIdx.x / wp.nblocks;
	wcol_tile<S, NCH, TAPS, Q, ODDW, false>(reinterpret_cast<const S *>(imgs.src[img]), reinterpret_cast<S *>(imgs.dst[img]), g, wp,
	                                        (int)blockIdx.x - img * wp.nblocks);
}
/* The weight form keeps 16 weights per lane on top of the window and sits at 194 registers; held to 168 (3 waves per
 * SIMD) it spills 9-15 dwords and is still faster: 106 vs 113 us on 6000x4000 RGBA u16. */
#ifndef WIDE_WEIGHTS_WAVES
#define WIDE_WEIGHTS_WAVES 3
#endif
#ifndef WIDE_WEIGHTS_Q
#define WIDE_WEIGHTS_Q 2       /* queue slots of the weight form (an A/B build option: deeper queues with 2 waves per SIMD) */
#endif
template <typename S, int NCH, int Q, bool ODDW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WIDE_WEIGHTS_WAVES))) void
k_wcol_weights(MoveBatch imgs, FixcaGeom g, WidePlan wp)
{
	const int img = blockIdx.x / wp.nblocks;
	wcol_tile<S, NCH, 4, Q, ODDW, true>(reinterpret_cast<const S *>(imgs.src[img]), reinterpret_cast<S *>(imgs.dst[img]), g, wp,
	                                    (int)blockIdx.x - img * wp.nblocks);
}

/* the instantiation a geometry runs: sample type x (interpolation, samples per pixel, odd width); `fast`: the weight
 * form with exact recomputation (16-bit samples, CUBIC, when the planner allows it) */
template <typename S>
const void *wcol_kernel(int sel, bool fast)
{
	/* queue slots: 3 where the registers allow it with 3 waves per SIMD (RGBA u16: 162-164), else 2 (measured equal
	 * within 3 % on RGBA u16; a third slot would push RGBA float and RGB u16 past 168 registers) */
#ifndef WIDE_Q_WIDE
#define WIDE_Q_WIDE WIDE_SLOTS   /* 4- and 8-byte samples: their cubic kernels hold 2 waves per SIMD whatever the queue; 3 slots measured 4 % faster on float and double LINEAR, equal elsewhere */
#endif
	constexpr int Q4 = sizeof(S) == 2 ? WIDE_SLOTS : WIDE_Q_WIDE, Q3 = sizeof(S) == 2 ? WIDE_SLOTS - 1 : WIDE_Q_WIDE;
	if constexpr (sizeof(S) == 2) {
		if (fast) switch (sel) {
		case 11: return reinterpret_cast<const void *>(&k_wcol_weights<S, 4, WIDE_WEIGHTS_Q, true>);
		case 10: return reinterpret_cast<const void *>(&k_wcol_weights<S, 4, WIDE_WEIGHTS_Q, false>);
		case 9:  return reinterpret_cast<const void *>(&k_wcol_weights<S, 3, WIDE_WEIGHTS_Q, true>);
		case 8:  return reinterpret_cast<const void *>(&k_wcol_weights<S, 3, WIDE_WEIGHTS_Q, false>);
		default: break;
		}
	}
	switch (sel) {
	case 11: return reinterpret_cast<const void *>(&k_wcol<S, 4, 4, Q4, true>);
	case 10: return reinterpret_cast<const void *>(&k_wcol<S, 4, 4, Q4, false>);
	case 9:  return reinterpret_cast<const void *>(&k_wcol<S, 3, 4, Q3, true>);
	case 8:  return reinterpret_cast<const void *>(&k_wcol<S, 3, 4, Q3, false>);
	case 7:  return reinterpret_cast<const void *>(&k_wcol<S, 4, 2, Q4, true>);
	case 6:  return reinterpret_cast<const void *>(&k_wcol<S, 4, 2, Q4, false>);
	case 5:  return reinterpret_cast<const void *>(&k_wcol<S, 3, 2, Q3, true>);
	case 4:  return reinterpret_cast<const void *>(&k_wcol<S, 3, 2, Q3, false>);
	case 3:  return reinterpret_cast<const void *>(&k_wcol<S, 4, 1, WIDE_SLOTS, true>);
	case 2:  return reinterpret_cast<const void *>(&k_wcol<S, 4, 1, WIDE_SLOTS, false>);
	case 1:  return reinterpret_cast<const void *>(&k_wcol<S, 3, 1, WIDE_SLOTS, true>);
	default: return reinterpret_cast<const void *>(&k_wcol<S, 3, 1, WIDE_SLOTS, false>);
	}
}
int wcol_type_index(int bpc) { return bpc == 2 ? 0 : bpc == 4 ? 1 : bpc == -4 ? 2 : bpc == -8 ? 3 : -1; }
int wcol_selector(const FixcaGeom *g) { return 4 * g->interp + (g->nch == 4 ? 2 : 0) + (g->W & 1); }     /* interp 0, 1, 2 */
const void *wcol_kernel_for(const FixcaGeom *g, bool fast)
{
	const int sel = wcol_selector(g);
	switch (g->bpc) {
	case 2:  return wcol_kernel<uint16_t>(sel, fast);
	case 4:  return wcol_kernel<uint32_t>(sel, fast);
	case -4: return wcol_kernel<float>(sel, fast);
	case -8: return wcol_kernel<double>(sel, fast);
	default: return nullptr;
	}
}

}   /* namespace */

/*
 * Waves of this geometry's kernel the device holds at once (resident waves per CU x CUs), 0 when that cannot be asked
 * (no device).  The planner sizes the tiles from it: the kernel is bound by fp64 issue, every wave of a launch does
 * the same work, so the launch is fastest when all tiles are resident together and every SIMD has the same number.
 */
extern "C" int
fixca_wide_slots(const FixcaGeom *g, int fast)
{
	static std::mutex mu;
	constexpr int MAXDEV = 16;                          /* answers are per device: CU count and occupancy may differ */
	static int cached_all[MAXDEV][2][4][12];
	const int ti = wcol_type_index(g->bpc);
	if (ti < 0 || g->interp < 0 || g->interp > 2) return 0;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	if (dev < 0 || dev >= MAXDEV) return 0;
	int (&cached)[4][12] = cached_all[dev][fast ? 1 : 0];                         /* 0 = not asked yet, -1 = cannot be asked */
	const int sel = wcol_selector(g);
	std::lock_guard<std::mutex> lock(mu);
	if (cached[ti][sel] == 0) {
		int per_cu = 0, cus = 0;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
		    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wcol_kernel_for(g, fast != 0), 64, 0) == hipSuccess && per_cu > 0 && cus > 0)
			cached[ti][sel] = per_cu * cus;
		else {
			(void)hipGetLastError();
			cached[ti][sel] = -1;
		}
	}
	return cached[ti][sel] > 0 ? cached[ti][sel] : 0;
}

/* the current device, -1 when there is none: part of the planner's cache key for the sample types whose tile height
 * comes from fixca_wide_slots */
extern "C" int
fixca_wide_device(void)
{
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess) {
		(void)hipGetLastError();
		return -1;
	}
	return dev;
}

extern "C" hipError_t
fixca_launch_wide_batch(const void *const *d_src, void *const *d_dst, int n_images, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	if (g->y2 <= g->y1) return hipSuccess;
	if (n_images < 1 || n_images > MOVE_BATCH_MAX) return hipErrorInvalidValue;
	if (g->interp < 0 || g->interp > 2) return hipErrorInvalidValue;
	const void *fn = wcol_kernel_for(g, pl->wide_fast != 0);
	if (!fn) return hipErrorInvalidValue;
	WidePlan wp;
	wp.oc = pl->oc; wp.th = pl->th; wp.nstrips = pl->nstrips; wp.nwaves = pl->nwaves; wp.nblocks = pl->nblocks;
	MoveBatch imgs;
	for (int i = 0; i < MOVE_BATCH_MAX; ++i) {
		imgs.src[i] = static_cast<const unsigned char *>(d_src[i < n_images ? i : 0]);
		imgs.dst[i] = static_cast<unsigned char *>(d_dst[i < n_images ? i : 0]);
	}
	FixcaGeom geom = *g;
	void *args[] = { &imgs, &geom, &wp };
	return hipLaunchKernel(fn, dim3((unsigned)wp.nblocks * (unsigned)n_images), dim3(64), args, 0, stream);
}

extern "C" hipError_t
fixca_launch_wide(const void *d_src, void *d_dst, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	return fixca_launch_wide_batch(&d_src, &d_dst, 1, g, pl, stream);
}
