/*
 * fixca_lin.hip — LINEAR on RGBA and RGB images of the wide sample types (u32, float, double; RGBA u16 on request) in the
 * structure of the NONE mover (fixca_move.hip): an LDS-DMA ring of raw source rows, 16-byte units per lane, nt stores,
 * with the reference's bilinear arithmetic (fix-ca.c:882-894, get_pixel / set_pixel :713-774) in fp64, operation for
 * operation: bit-identical to the CPU oracle.
 *
 * Why a second LINEAR kernel next to k_wcol (fixca_wide.hip).  k_wcol keeps a rolling window of horizontally
 * filtered rows in registers and so does every filter operation once; it is built for CUBIC, where that saves 3/4 of
 * the fp64 work.  Its loads sit in a register queue two rows ahead, and a wave that waits for its stores to be
 * acknowledged has nothing else in flight: with its stores sent out of range RGBA float LINEAR runs in 93 instead of
 * 174 us (profiles/NOTES.md).  LINEAR has little arithmetic to save: one output sample is 4 conversions and 10 operations.
 * Recomputing both rows' horizontal interpolation for every output row (twice k_wcol's filter work, no window state)
 * costs ~56 us of fp64 issue per 6000x4000 RGBA float image against 132 us of memory time at 5.8 TB/s, and frees the
 * kernel to move data the way the mover does: source rows prefetched `depth` rows ahead by LDS-DMA (no registers, no
 * coupling to the stores), each fetched once per wave for all its consumers (identity copy, both rows of the red and
 * of the blue interpolation).
 *
 *  - a lane owns NU 16-byte units of the strip's output row: one pixel (4-byte samples), half a pixel (double: red
 *    and green, or blue and alpha) or two pixels (u16);
 *  - per computed sample: column c0 = floor(x'), c1 = c0 + 1 clamped (fix-ca.c:1170-1177), fraction fx: per-lane
 *    constants of the tile (the column map depends on the column only); per output row and channel: r0 = floor(y'),
 *    r1 clamped (fix-ca.c:1149-1158), fy: wave-uniform, from a per-64-rows table read with v_readlane;
 *  - 4 LDS reads of a sample each, get_pixel, the two horizontal and the vertical interpolation in the reference's
 *    order, clip_d, set_pixel; green and alpha come with the unit's own 16 bytes.
 *
 * RGB (k_lin_rgb): 16-byte units straddle 12- and 24-byte pixels, so a unit's samples would change channel from lane to
 * lane (tried: per-row selects of every sample's rows and fractions, green samples computed for nothing: slower than
 * k_wcol on 4-byte samples).  Instead a lane owns 24 BYTES OF WHOLE PIXELS (two pixels of 4-byte samples, one of
 * doubles): red and blue sit at fixed places, green is copied, and the wave's 1536 bytes of an output row are put
 * together in LDS (each lane writes its 24 bytes) and read back as 16-byte units in lane order, so that the stores are
 * the same full-KiB nt stores as everywhere.
 */
#include "fixca_device.h"
#include "fixca_internal.h"
#include "fixca_codec.h"
#include "fixca_move.h"
#include <mutex>
#include <type_traits>

namespace {
using fixca_codec::Codec;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void lin_wait_vm()
{
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int lin_slot(int row, int K, uint32_t kmagic)
{
	return row - K * (int)__umulhi((uint32_t)row, kmagic);
}
__device__ __forceinline__ double lin_readlane_f64(double v, int l)
{
	const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
	const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)b, l), hi = __builtin_amdgcn_readlane((uint32_t)(b >> 32), l);
	return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}

/* FAST (16-bit samples only): the weight form: both interpolations as fused multiply-adds on the integer samples, the
 * result rounded as the reference rounds, and every sample whose fraction is within WIDE_FAST_EPS of a rounding boundary
 * recomputed operation for operation from the source image (exact_linear_sample; bound in fixca_codec.h).  Without it
 * a 16-bit unit costs 16 exact quotients and 70 operations and the kernel is bound by fp64 issue (33-40 % of the HBM
 * peak against k_wcol's 55 %); with it 4 conversions and 5 operations per sample. */
template <typename S, int NU, bool FAST = false>   /* S: sample type; NU: KiB per ring row = units per lane and row */
__global__ __launch_bounds__(64) void
k_lin(MoveBatch imgs, FixcaGeom g, MovePlan mp)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	typedef Codec<S> C;
	static_assert(!FAST || sizeof(S) == 2, "the weight form is for 16-bit samples");
	constexpr int SB = (int)sizeof(S);
	constexpr int PB = 4 * SB;                         /* RGBA                                          */
	constexpr int WINB = NU * 1024;
	constexpr int NC = (SB == 2) ? 4 : (SB == 4) ? 2 : 1;      /* samples computed per unit: (R0 B0 R1 B1) / (R B) / (R | B) */

	const int lane = threadIdx.x;
	const int W = g.W, H = g.H;
	const int img = blockIdx.x / mp.nblocks;
	const int tile = xcd_remap(blockIdx.x - img * mp.nblocks, mp.nblocks);
	if (tile >= mp.nwaves) return;
	const uint8_t *__restrict__ src = imgs.src[img];
	uint8_t *__restrict__ dst = imgs.dst[img];
	const int strip = tile % mp.nstrips, chunk = tile / mp.nstrips;
	const int X0 = strip * mp.ocpx, xe = min(X0 + mp.ocpx, W);
	const int Y0 = g.y1 + chunk * mp.th;
	const int nrows = min(mp.th, g.y2 - Y0);
	const uint32_t pitch = (uint32_t)W * PB;
	const int K = mp.ring;
	const uint32_t kmagic = (uint32_t)((0x100000000ull + (unsigned)K - 1) / (unsigned)K);

	/* the strip's window starts at the lowest column any consumer reads: the strip's own first column or the first
	 * tap of its first column (the maps are monotone), rounded down to a 16-byte boundary of the row */
	int lo = X0;
#pragma unroll
	for (int c = 0; c < 2; ++c) lo = min(lo, (int)floor(map_dbl(X0, g.xc, W, g.s[c], g.xs[c])));
	constexpr int CA = (PB % 16 == 0) ? 1 : 2;         /* u16 RGBA: 8-byte pixels                       */
	const int U0 = __builtin_amdgcn_readfirstlane(lo & ~(CA - 1));

	/* ---- per-lane constants ---------------------------------------------------------------------------------------- */
	const int nunits = ((xe - X0) * PB + 15) >> 4;     /* RGBA rows are whole units except for u16 images of odd width   */
	const int tail_dw = (((xe - X0) * PB) & 15) >> 2;
	uint32_t off0[NU][NC], off1[NU][NC];               /* byte offsets in a ring row of the two taps' samples             */
	double   fx[NU][NC];
	uint32_t idoff[NU], stoff[NU];
	int      blue[NU];                                 /* double: this unit holds (B, A), not (R, G)                      */
	int      pxcol[FAST ? NU : 1][FAST ? NC : 1];      /* FAST: the output column of the sample (for the recomputation)   */
#pragma unroll
	for (int i = 0; i < NU; ++i) {
		const int U = min(lane + 64 * i, nunits - 1);
		idoff[i] = (uint32_t)((X0 - U0) * PB + 16 * U);
		blue[i] = (SB == 8) ? (U & 1) : 0;
#pragma unroll
		for (int j = 0; j < NC; ++j) {
			const int c = (SB == 8) ? blue[i] : (j & 1);                     /* 0 red, 1 blue */
			const int px = min(X0 + ((SB == 8) ? U / 2 : (SB == 4) ? U : 2 * U + j / 2), W - 1);
			if constexpr (FAST) pxcol[i][j] = px;
			const double xd = map_dbl(px, g.xc, W, g.s[c], g.xs[c]);
			const int c0 = (int)floor(xd);
			const int c1 = (c0 == W - 1) ? c0 : c0 + 1;                      /* fix-ca.c:1170-1177 */
			fx[i][j] = __dsub_rn(xd, (double)c0);
			off0[i][j] = (uint32_t)((c0 - U0) * PB + 2 * c * SB);
			off1[i][j] = (uint32_t)((c1 - U0) * PB + 2 * c * SB);
		}
	}

	/* ---- rows: lane i of a 64-row segment holds output row (segment + i)'s source rows and fractions ------------------ */
	int    r0_v[2] = { 0, 0 };
	double fy_v[2] = { 0.0, 0.0 };
	auto row_tables = [&](int rseg) {
		const int y = min(Y0 + rseg + lane, g.y2 - 1);
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			const double yd = map_dbl(y, g.yc, H, g.s[c], g.ys[c]);
			r0_v[c] = (int)floor(yd);
			fy_v[c] = __dsub_rn(yd, (double)r0_v[c]);
		}
	};
	row_tables(0);
	const int yl = Y0 + nrows - 1;
	const int keep0 = __builtin_amdgcn_readfirstlane(min(Y0, min(__builtin_amdgcn_readlane(r0_v[0], 0), __builtin_amdgcn_readlane(r0_v[1], 0))));   /* readfirstlane: see k_cub */
	int lastneed = yl;
#pragma unroll
	for (int c = 0; c < 2; ++c) lastneed = max(lastneed, min((int)floor(map_dbl(yl, g.yc, H, g.s[c], g.ys[c])) + 1, H - 1));
	const int lastneed_s = __builtin_amdgcn_readfirstlane(lastneed);
	const uint32_t dst_bytes = (uint32_t)nrows * pitch;
	const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + (size_t)keep0 * pitch, 0,
	                                                        (uint32_t)(lastneed_s - keep0 + 1) * pitch, 0x00020000);
	const rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)Y0 * pitch, 0, dst_bytes, 0x00020000);
	(void)rs_src;
#pragma unroll
	for (int i = 0; i < NU; ++i) {
		const int U = lane + 64 * i;
		stoff[i] = (U < nunits - (tail_dw ? 1 : 0)) ? (uint32_t)(X0 * PB + 16 * U) : dst_bytes;
	}
	const int tail_u = (nunits - 1) >> 6, tail_lane = (nunits - 1) & 63;
	const int nbytes_row = (xe - X0) * PB;             /* bytes of the strip's output row */
	uint32_t row_phase = (uint32_t)(uintptr_t)dst + (uint32_t)Y0 * pitch + (uint32_t)X0 * PB;      /* mod 2^32: only the low 6 bits are used */

	/* ---- the ring (as in k_move) ---------------------------------------------------------------------------------------- */
	const uint32_t vo_ld = (uint32_t)U0 * PB + 16u * (uint32_t)lane;
	(void)vo_ld;
	int ld = keep0 - 1;
	[[maybe_unused]] uint32_t so_ld = 0;      /* the host pass of hipcc sees no use of it */
	auto issue = [&]() {
		++ld;
#if defined(__HIP_DEVICE_COMPILE__)
		const int slot = lin_slot(ld, K, kmagic);
		__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(ring + slot * WINB);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 0, 0);
		if constexpr (NU > 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 1024, 0);
		if constexpr (NU > 2) {
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 2048, 0);
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 3072, 0);
		}
#endif
		so_ld += pitch;
	};
	auto fetch = [&](int keep, int need) {
		const int target = min(min(need + mp.depth, keep + K - 1), lastneed_s);
		while (ld < target) issue();
		const int younger = ld - need;
		if (younger >= 12) lin_wait_vm<12 * NU>();
		else if (younger >= 10) lin_wait_vm<10 * NU>();
		else if (younger >= 8) lin_wait_vm<8 * NU>();
		else if (younger >= 6) lin_wait_vm<6 * NU>();
		else if (younger >= 4) lin_wait_vm<4 * NU>();
		else if (younger >= 3) lin_wait_vm<3 * NU>();
		else if (younger >= 2) lin_wait_vm<2 * NU>();
		else if (younger >= 1) lin_wait_vm<NU>();
		else lin_wait_vm<0>();
	};

	uint32_t so_st = 0;
	for (int r = 0; r < nrows; ++r) {
		if (r && (r & 63) == 0) row_tables(r);
		const int y = Y0 + r;
		int r0[2], r1[2];
		double fy[2], omfy[2];
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			r0[c] = __builtin_amdgcn_readlane(r0_v[c], r & 63);
			r1[c] = (r0[c] == H - 1) ? r0[c] : r0[c] + 1;                    /* fix-ca.c:1149-1158 */
			fy[c] = lin_readlane_f64(fy_v[c], r & 63);
			omfy[c] = __dsub_rn(1.0, fy[c]);                                /* (1 - dy), fix-ca.c:891 */
		}
		/* every LDS read of the previous row has returned (its stores consumed them): the slots below `keep` are free */
		fetch(min(y, min(r0[0], r0[1])), max(y, max(r1[0], r1[1])));
		const uint32_t bI = (uint32_t)lin_slot(y, K, kmagic) * WINB;
		uint32_t b0[2], b1[2];
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			b0[c] = (uint32_t)lin_slot(r0[c], K, kmagic) * WINB;
			b1[c] = (uint32_t)lin_slot(r1[c], K, kmagic) * WINB;
		}
		u32x4 out[NU];
#pragma unroll
		for (int i = 0; i < NU; ++i) {
			u32x4 v = *reinterpret_cast<const u32x4 *>(ring + bI + idoff[i]);
			S res[NC];
			unsigned long long redo_m[NC];                 /* FAST: lanes whose sample j is within the recheck band (scalar registers) */
#pragma unroll
			for (int j = 0; j < NC; ++j) redo_m[j] = 0;
#pragma unroll
			for (int j = 0; j < NC; ++j) {
				uint32_t a0, a1;
				double wy, wy1;
				if constexpr (SB == 8) {
					a0 = blue[i] ? b0[1] : b0[0]; a1 = blue[i] ? b1[1] : b1[0];
					wy = blue[i] ? fy[1] : fy[0]; wy1 = blue[i] ? omfy[1] : omfy[0];
				} else {
					a0 = b0[j & 1]; a1 = b1[j & 1];
					wy = fy[j & 1]; wy1 = omfy[j & 1];
				}
				const S s00 = *reinterpret_cast<const S *>(ring + a0 + off0[i][j]), s10 = *reinterpret_cast<const S *>(ring + a0 + off1[i][j]);
				const S s01 = *reinterpret_cast<const S *>(ring + a1 + off0[i][j]), s11 = *reinterpret_cast<const S *>(ring + a1 + off1[i][j]);
				if constexpr (FAST) {
					const double h0 = fma(fx[i][j], (double)s10 - (double)s00, (double)s00);
					const double h1 = fma(fx[i][j], (double)s11 - (double)s01, (double)s01);
					bool near;
					res[j] = (S)fixca_codec::fast_round_u16(fma(wy, h1 - h0, h0), near);
					redo_m[j] = __builtin_amdgcn_ballot_w64(near);
				} else {
					const double p00 = C::get(s00), p10 = C::get(s10), p01 = C::get(s01), p11 = C::get(s11);
					/* bilinear(), fix-ca.c:891-893, operation for operation */
					const double a = __dadd_rn(p00, __dmul_rn(fx[i][j], __dsub_rn(p10, p00)));
					const double b = __dadd_rn(p01, __dmul_rn(fx[i][j], __dsub_rn(p11, p01)));
					res[j] = C::put(clip01(__dadd_rn(__dmul_rn(wy1, a), __dmul_rn(wy, b))));
				}
			}
			if constexpr (FAST) {
				unsigned long long any = 0;
#pragma unroll
				for (int j = 0; j < NC; ++j) any |= redo_m[j];
				if (__builtin_expect(any != 0, 0)) {
					/* a few samples per image: recomputed the reference's way from the source image */
					uint32_t redo = 0;
#pragma unroll
					for (int j = 0; j < NC; ++j) redo |= (uint32_t)((redo_m[j] >> lane) & 1ull) << j;
					while (redo) {
						const int q = __builtin_ctz(redo);
						redo &= redo - 1;
						int col = pxcol[i][0];
#pragma unroll
						for (int j = 1; j < NC; ++j) col = (q == j) ? pxcol[i][j] : col;
						const S e = fixca_codec::exact_linear_sample<S, 4>(reinterpret_cast<const S *>(src), g, col, y, q & 1);
#pragma unroll
						for (int j = 0; j < NC; ++j) res[j] = (q == j) ? e : res[j];
					}
				}
			}
			if constexpr (SB == 2) {
				v.x = (v.x & 0xFFFF0000u) | (uint32_t)res[0]; v.y = (v.y & 0xFFFF0000u) | (uint32_t)res[1];
				v.z = (v.z & 0xFFFF0000u) | (uint32_t)res[2]; v.w = (v.w & 0xFFFF0000u) | (uint32_t)res[3];
			} else if constexpr (SB == 4) {
				v.x = __builtin_bit_cast(uint32_t, res[0]);
				v.z = __builtin_bit_cast(uint32_t, res[1]);
			} else {
				const unsigned long long bits = __builtin_bit_cast(unsigned long long, res[0]);
				v.x = (uint32_t)bits; v.y = (uint32_t)(bits >> 32);
			}
			out[i] = v;
		}
		/* scalar offset the constant 0: the store-data hazard of profiles/NOTES.md (tests/test_host_logic.py scans for it).
		 * A row whose strip does not start and end on 64-byte boundaries (a pitch that is not a multiple of 64: RGBA float of
		 * a width that is not a multiple of 4; 6002 wide 68.0 against 72.4 % at 6000) shares its end sectors with the
		 * neighbour strip's wave: in launches of several images the lanes there store write-back, as in k_roll. */
		bool split = false;
		uint32_t a0 = 0, endb = 0;
		if (__builtin_expect(mp.edge_wb != 0, 0)) {
			a0 = row_phase & 63u;
			endb = a0 + (uint32_t)nbytes_row;
			row_phase += pitch;
			split = (a0 | (endb & 63u)) != 0;
		}
		if (__builtin_expect(split, 0)) {
#pragma unroll
			for (int i = 0; i < NU; ++i) {
				const uint32_t first = 16u * (uint32_t)(lane + 64 * i) + a0;
				const bool edge = (a0 != 0 && first < 64u) || ((endb & 63u) != 0 && ((first + 15u) >> 6) >= ((endb - 1) >> 6));
				const uint32_t off = stoff[i] == dst_bytes ? dst_bytes : stoff[i] + so_st;
				__builtin_amdgcn_raw_buffer_store_b128(out[i], rs_dst, edge ? dst_bytes : off, 0, 2);
				__builtin_amdgcn_raw_buffer_store_b128(out[i], rs_dst, edge ? off : dst_bytes, 0, 0);
			}
		} else {
#pragma unroll
			for (int i = 0; i < NU; ++i)
				__builtin_amdgcn_raw_buffer_store_b128(out[i], rs_dst, stoff[i] + so_st, 0, 2);
		}
		if (tail_dw) {                                  /* u16 images of odd width: the row's last unit is half a unit */
#pragma unroll
			for (int i = 0; i < NU; ++i)
				if (i == tail_u) {
					const bool mine = lane == tail_lane;
					const uint32_t o = (uint32_t)(X0 * PB + 16 * (lane + 64 * i)) + so_st;
#pragma unroll
					for (int d = 0; d < 3; ++d)
						if (d < tail_dw) __builtin_amdgcn_raw_buffer_store_b32(out[i][d], rs_dst, mine ? o + 4 * d : dst_bytes, 0, 2);
				}
		}
		so_st += pitch;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the next row's DMA may overwrite what was read here */
	}
}

#define LIN_RGB_LANE_BYTES 24                       /* a lane's whole pixels: 2 x 12 bytes or 1 x 24 bytes */
#define LIN_RGB_ROW_BYTES  (64 * LIN_RGB_LANE_BYTES)   /* 1536: the strip's output row, assembled in LDS */
#define LIN_RGB_NU         2                        /* 2-KiB ring rows: 170 columns of RGB float, 85 of RGB double */

template <typename S>                              /* u32, float, double */
__global__ __launch_bounds__(64) void
k_lin_rgb(MoveBatch imgs, FixcaGeom g, MovePlan mp)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];     /* K ring rows, then the staging row */
	typedef Codec<S> C;
	constexpr int SB = (int)sizeof(S);
	static_assert(SB == 4 || SB == 8, "RGB u16 stays with k_wcol");
	constexpr int PB = 3 * SB;
	constexpr int PXL = LIN_RGB_LANE_BYTES / PB;       /* pixels per lane: 2 or 1                       */
	constexpr int NU = LIN_RGB_NU;
	constexpr int WINB = NU * 1024;
	constexpr int LW = LIN_RGB_LANE_BYTES / 4;         /* dwords a lane contributes to the row          */

	const int lane = threadIdx.x;
	const int W = g.W, H = g.H;
	const int img = blockIdx.x / mp.nblocks;
	const int tile = xcd_remap(blockIdx.x - img * mp.nblocks, mp.nblocks);
	if (tile >= mp.nwaves) return;
	const uint8_t *__restrict__ src = imgs.src[img];
	uint8_t *__restrict__ dst = imgs.dst[img];
	const int strip = tile % mp.nstrips, chunk = tile / mp.nstrips;
	const int X0 = strip * mp.ocpx, xe = min(X0 + mp.ocpx, W);
	const int Y0 = g.y1 + chunk * mp.th;
	const int nrows = min(mp.th, g.y2 - Y0);
	const uint32_t pitch = (uint32_t)W * PB;
	const int K = mp.ring;
	const uint32_t kmagic = (uint32_t)((0x100000000ull + (unsigned)K - 1) / (unsigned)K);
	uint32_t *const stage = reinterpret_cast<uint32_t *>(ring + (size_t)K * WINB);

	int lo = X0;
#pragma unroll
	for (int c = 0; c < 2; ++c) lo = min(lo, (int)floor(map_dbl(X0, g.xc, W, g.s[c], g.xs[c])));
	constexpr int CA = (PB % 8 == 0) ? 2 : 4;          /* columns per 16 bytes of row: the DMA's global address stays 16-byte aligned in the row */
	const int U0 = __builtin_amdgcn_readfirstlane(lo & ~(CA - 1));

	/* ---- per-lane constants: my pixels, the taps of their red and blue samples ---------------------------------------- */
	const int nbytes = (xe - X0) * PB;
	const int nunits = (nbytes + 15) >> 4;
	const int tail_dw = (nbytes & 15) >> 2;
	uint32_t off0[PXL][2], off1[PXL][2], goff[PXL];
	double   fx[PXL][2];
#pragma unroll
	for (int j = 0; j < PXL; ++j) {
		const int px = min(X0 + PXL * lane + j, W - 1);
		goff[j] = (uint32_t)((px - U0) * PB + SB);                         /* my green sample */
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			const double xd = map_dbl(px, g.xc, W, g.s[c], g.xs[c]);
			const int c0 = (int)floor(xd);
			const int c1 = (c0 == W - 1) ? c0 : c0 + 1;                      /* fix-ca.c:1170-1177 */
			fx[j][c] = __dsub_rn(xd, (double)c0);
			off0[j][c] = (uint32_t)((c0 - U0) * PB + 2 * c * SB);
			off1[j][c] = (uint32_t)((c1 - U0) * PB + 2 * c * SB);
		}
	}
	/* the units I read back from the staging row and store: unit lane, and unit 64 + lane for the first 32 lanes */
	uint32_t stoff[2];
#pragma unroll
	for (int i = 0; i < 2; ++i) {
		const int U = lane + 64 * i;
		stoff[i] = (U < nunits - (tail_dw ? 1 : 0)) ? (uint32_t)(X0 * PB + 16 * U) : 0xFFFFFFFFu;
	}

	int    r0_v[2] = { 0, 0 };
	double fy_v[2] = { 0.0, 0.0 };
	auto row_tables = [&](int rseg) {
		const int y = min(Y0 + rseg + lane, g.y2 - 1);
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			const double yd = map_dbl(y, g.yc, H, g.s[c], g.ys[c]);
			r0_v[c] = (int)floor(yd);
			fy_v[c] = __dsub_rn(yd, (double)r0_v[c]);
		}
	};
	row_tables(0);
	const int yl = Y0 + nrows - 1;
	const int keep0 = __builtin_amdgcn_readfirstlane(min(Y0, min(__builtin_amdgcn_readlane(r0_v[0], 0), __builtin_amdgcn_readlane(r0_v[1], 0))));   /* readfirstlane: see k_cub */
	int lastneed = yl;
#pragma unroll
	for (int c = 0; c < 2; ++c) lastneed = max(lastneed, min((int)floor(map_dbl(yl, g.yc, H, g.s[c], g.ys[c])) + 1, H - 1));
	const int lastneed_s = __builtin_amdgcn_readfirstlane(lastneed);
	const uint32_t dst_bytes = (uint32_t)nrows * pitch;
	const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + (size_t)keep0 * pitch, 0,
	                                                        (uint32_t)(lastneed_s - keep0 + 1) * pitch, 0x00020000);
	const rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)Y0 * pitch, 0, dst_bytes, 0x00020000);
	(void)rs_src;
#pragma unroll
	for (int i = 0; i < 2; ++i) stoff[i] = (stoff[i] == 0xFFFFFFFFu) ? dst_bytes : stoff[i];
	const int tail_u = (nunits - 1) >> 6, tail_lane = (nunits - 1) & 63;

	const uint32_t vo_ld = (uint32_t)U0 * PB + 16u * (uint32_t)lane;
	(void)vo_ld;
	int ld = keep0 - 1;
	[[maybe_unused]] uint32_t so_ld = 0;      /* the host pass of hipcc sees no use of it */
	auto issue = [&]() {
		++ld;
#if defined(__HIP_DEVICE_COMPILE__)
		const int slot = lin_slot(ld, K, kmagic);
		__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(ring + slot * WINB);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 0, 0);
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 1024, 0);
#endif
		so_ld += pitch;
	};
	auto fetch = [&](int keep, int need) {
		const int target = min(min(need + mp.depth, keep + K - 1), lastneed_s);
		while (ld < target) issue();
		const int younger = ld - need;
		if (younger >= 12) lin_wait_vm<12 * NU>();
		else if (younger >= 10) lin_wait_vm<10 * NU>();
		else if (younger >= 8) lin_wait_vm<8 * NU>();
		else if (younger >= 6) lin_wait_vm<6 * NU>();
		else if (younger >= 4) lin_wait_vm<4 * NU>();
		else if (younger >= 3) lin_wait_vm<3 * NU>();
		else if (younger >= 2) lin_wait_vm<2 * NU>();
		else if (younger >= 1) lin_wait_vm<NU>();
		else lin_wait_vm<0>();
	};

	uint32_t so_st = 0;
	for (int r = 0; r < nrows; ++r) {
		if (r && (r & 63) == 0) row_tables(r);
		const int y = Y0 + r;
		int r0[2], r1[2];
		double fy[2], omfy[2];
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			r0[c] = __builtin_amdgcn_readlane(r0_v[c], r & 63);
			r1[c] = (r0[c] == H - 1) ? r0[c] : r0[c] + 1;                    /* fix-ca.c:1149-1158 */
			fy[c] = lin_readlane_f64(fy_v[c], r & 63);
			omfy[c] = __dsub_rn(1.0, fy[c]);
		}
		/* the previous row's LDS reads (ring and staging) have returned: its stores consumed them */
		fetch(min(y, min(r0[0], r0[1])), max(y, max(r1[0], r1[1])));
		const uint32_t bI = (uint32_t)lin_slot(y, K, kmagic) * WINB;
		uint32_t b0[2], b1[2];
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			b0[c] = (uint32_t)lin_slot(r0[c], K, kmagic) * WINB;
			b1[c] = (uint32_t)lin_slot(r1[c], K, kmagic) * WINB;
		}
		S mine[PXL][3];                                                    /* my pixels: R, G, B */
#pragma unroll
		for (int j = 0; j < PXL; ++j) {
			mine[j][1] = *reinterpret_cast<const S *>(ring + bI + goff[j]);    /* green is copied (fix-ca.c:1098) */
#pragma unroll
			for (int c = 0; c < 2; ++c) {
				const double p00 = C::get(*reinterpret_cast<const S *>(ring + b0[c] + off0[j][c]));
				const double p10 = C::get(*reinterpret_cast<const S *>(ring + b0[c] + off1[j][c]));
				const double p01 = C::get(*reinterpret_cast<const S *>(ring + b1[c] + off0[j][c]));
				const double p11 = C::get(*reinterpret_cast<const S *>(ring + b1[c] + off1[j][c]));
				/* bilinear(), fix-ca.c:891-893, operation for operation */
				const double a = __dadd_rn(p00, __dmul_rn(fx[j][c], __dsub_rn(p10, p00)));
				const double b = __dadd_rn(p01, __dmul_rn(fx[j][c], __dsub_rn(p11, p01)));
				mine[j][2 * c] = C::put(clip01(__dadd_rn(__dmul_rn(omfy[c], a), __dmul_rn(fy[c], b))));
			}
		}
		/* my 24 bytes into the staging row, as dwords (written and read with one type: the compiler orders accesses by type) */
		uint32_t w[LW];
#pragma unroll
		for (int j = 0; j < PXL; ++j)
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				if constexpr (SB == 4) w[3 * j + k] = __builtin_bit_cast(uint32_t, mine[j][k]);
				else { const unsigned long long q = __builtin_bit_cast(unsigned long long, mine[j][k]); w[2 * k] = (uint32_t)q; w[2 * k + 1] = (uint32_t)(q >> 32); }
			}
		asm volatile("" ::: "memory");
#pragma unroll
		for (int d = 0; d < LW; ++d) stage[LW * lane + d] = w[d];
		asm volatile("" ::: "memory");                  /* one wave: its LDS operations complete in order, the reads below see these writes */
		u32x4 out[2];
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const int U = min(lane + 64 * i, LIN_RGB_ROW_BYTES / 16 - 1);
#pragma unroll
			for (int d = 0; d < 4; ++d) out[i][d] = stage[4 * U + d];
		}
		asm volatile("" ::: "memory");
		/* scalar offset the constant 0: the store-data hazard of profiles/NOTES.md (tests/test_host_logic.py scans for it) */
#pragma unroll
		for (int i = 0; i < 2; ++i)
			__builtin_amdgcn_raw_buffer_store_b128(out[i], rs_dst, stoff[i] == dst_bytes ? dst_bytes : stoff[i] + so_st, 0, 2);
		if (tail_dw) {                                  /* the row's last unit is partial (widths that are not a multiple of 4 / 2 pixels) */
#pragma unroll
			for (int i = 0; i < 2; ++i)
				if (i == tail_u) {
					const bool mine_t = lane == tail_lane;
					const uint32_t o = (uint32_t)(X0 * PB + 16 * (lane + 64 * i)) + so_st;
#pragma unroll
					for (int d = 0; d < 3; ++d)
						if (d < tail_dw) __builtin_amdgcn_raw_buffer_store_b32(out[i][d], rs_dst, mine_t ? o + 4 * d : dst_bytes, 0, 2);
				}
		}
		so_st += pitch;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the next row's DMA and staging writes may overwrite what was read here */
	}
}

template <typename S>
const void *lin_kernel_nu(int nu)
{
	switch (nu) {
	case 1:  return reinterpret_cast<const void *>(&k_lin<S, 1>);
	case 2:  return reinterpret_cast<const void *>(&k_lin<S, 2>);
	default: return reinterpret_cast<const void *>(&k_lin<S, 4>);
	}
}

const void *lin_kernel_for(const FixcaGeom *g, int nu, bool fast)
{
	if (g->nch == 3) {
		if (nu != LIN_RGB_NU) return nullptr;
		switch (g->bpc) {
		case 4:  return reinterpret_cast<const void *>(&k_lin_rgb<uint32_t>);
		case -4: return reinterpret_cast<const void *>(&k_lin_rgb<float>);
		case -8: return reinterpret_cast<const void *>(&k_lin_rgb<double>);
		default: return nullptr;
		}
	}
	switch (g->bpc) {
	case 2:
		if (!fast) return lin_kernel_nu<uint16_t>(nu);
		switch (nu) {
		case 1:  return reinterpret_cast<const void *>(&k_lin<uint16_t, 1, true>);
		case 2:  return reinterpret_cast<const void *>(&k_lin<uint16_t, 2, true>);
		default: return reinterpret_cast<const void *>(&k_lin<uint16_t, 4, true>);
		}
	case 4:  return lin_kernel_nu<uint32_t>(nu);
	case -4: return lin_kernel_nu<float>(nu);
	case -8: return lin_kernel_nu<double>(nu);
	default: return nullptr;
	}
}

}   /* namespace */

extern "C" hipError_t
fixca_launch_lin_batch(const void *const *d_src, void *const *d_dst, int n_images, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	if (g->y2 <= g->y1) return hipSuccess;
	if (n_images < 1 || n_images > MOVE_BATCH_MAX) return hipErrorInvalidValue;
	if (g->interp != 1 || (pl->nu != 1 && pl->nu != 2 && pl->nu != 4) || pl->depth < 1 || pl->depth > MOVE_DEPTH_MAX) return hipErrorInvalidValue;
	const void *fn = lin_kernel_for(g, pl->nu, pl->wide_fast != 0);
	if (!fn) return hipErrorInvalidValue;
	/* + a pad for RGBA (the 16-byte read of a last, partial unit), + the staging row for RGB */
	const size_t lds = (size_t)pl->ring * pl->nu * 1024 + (g->nch == 3 ? LIN_RGB_ROW_BYTES : 64);
	if (lds > MOVE_LDS_MAX + LIN_RGB_ROW_BYTES) return hipErrorInvalidValue;
	hipError_t ea = (hipError_t)fixca_dyn_lds(fn, MOVE_LDS_MAX + LIN_RGB_ROW_BYTES);
	if (ea != hipSuccess) return ea;
	MovePlan mp;
	mp.ocpx = pl->oc; mp.th = pl->th; mp.ring = pl->ring; mp.depth = pl->depth; mp.nstrips = pl->nstrips; mp.nwaves = pl->nwaves; mp.nblocks = pl->nblocks;
	mp.edge_wb = fixca_debug_env("FIXCA_EDGE_WB", n_images > 1 ? 1 : 0, 0, 1);
	MoveBatch imgs;
	for (int i = 0; i < MOVE_BATCH_MAX; ++i) {
		imgs.src[i] = static_cast<const unsigned char *>(d_src[i < n_images ? i : 0]);
		imgs.dst[i] = static_cast<unsigned char *>(d_dst[i < n_images ? i : 0]);
	}
	FixcaGeom geom = *g;
	void *args[] = { &imgs, &geom, &mp };
	return hipLaunchKernel(fn, dim3((unsigned)mp.nblocks * (unsigned)n_images), dim3(64), args, lds, stream);
}

extern "C" hipError_t
fixca_launch_lin(const void *d_src, void *d_dst, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	return fixca_launch_lin_batch(&d_src, &d_dst, 1, g, pl, stream);
}
