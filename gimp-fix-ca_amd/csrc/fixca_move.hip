/*
 * fixca_move.hip — interpolation NONE on the sample types beyond 8 bits (u16, u32, float, double; RGB and RGBA):
 * a byte mover.  NONE never looks at a sample's value: the reference copies the b bytes of the red and of the blue
 * sample from the nearest source position and leaves green and alpha where they are (fix-ca.c:1100-1121, `memcpy`
 * of b bytes at :1117-1120; the positions are scale(), fix-ca.c:799-809).  Bit-exact by construction.
 *
 * Structure of the row-rolling kernel (fixca_roll.hip), on raw bytes:
 *
 *   HBM/L2 --LDS-DMA--> ring of K source rows in LDS --> registers --> HBM (16-byte nt stores)
 *
 *  - one wave (a 64-thread workgroup) per tile of `ocpx` output columns x `th` rows.  A ring row holds the strip's
 *    source window, NU KiB (NU = 1, 2 or 4 LDS-DMA instructions of 16 bytes per lane); every source row is fetched
 *    once per wave and serves all three consumers: the identity copy (green, alpha) of output row y, the red samples
 *    of the output row whose red source row it is, and the blue ones likewise.  (k_wcol's register queue, which ran
 *    NONE on 16-bit samples in round 2, fetched it three times, the one-thread-per-pixel gather issued 3 loads per
 *    pixel: 53 % and 58-63 % of the HBM roofline.)
 *  - a lane owns NU 16-byte UNITS of the strip's output row (unit = lane + 64 i: every store instruction writes 1 KiB
 *    of consecutive bytes).  The column map x' depends on the column only, so the byte offset in a ring row of every
 *    sample a lane gathers is a per-lane constant for the whole tile, computed once with the reference's fp64
 *    expression; the row map y' depends on the row only: per row three wave-uniform ring slots (identity, red, blue).
 *      RGBA: the unit is read whole (ds_read_b128) and its red / blue samples are overwritten by 2-, 4- or 8-byte
 *            LDS reads from the red / blue source row;
 *      RGB:  units straddle pixels (16 is not a multiple of 6, 12 or 24), the channel of sample k of a unit depends on
 *            the unit modulo 3: every sample is gathered, from the row its channel calls for.
 *  - counted s_waitcnt vmcnt on the loads alone, rows prefetched `depth` ahead, as in k_roll; output rows go in
 *    blocks of RB (RB x NU = 4 units per lane): all LDS reads of a block are issued before the first is used.
 *  - a launch takes up to MOVE_BATCH_MAX images of one geometry (fixca_batch_device).
 *
 * The planner (plan_move, fixca_plan.cpp) takes a geometry when every strip's window (identity columns + red and
 * blue source columns) fits a ring row and rows are a whole number of dwords; everything else stays with
 * k_wcol<.., 1> / the one-thread-per-pixel kernel.
 */
#include "fixca_device.h"
#include "fixca_internal.h"
#include "fixca_move.h"
#include <mutex>
#include <type_traits>

namespace {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
/* the dwords at +0 and +8 of a 4-byte aligned LDS address: the compiler makes one ds_read2_b32 of them */
struct __attribute__((packed, aligned(4))) u32x2pk { uint32_t a, pad_, b; };

template <int N> __device__ __forceinline__ void move_wait_vm()
{
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

/* row mod K for any K (wave-uniform operands: scalar multiply-high) */
__device__ __forceinline__ int move_slot(int row, int K, uint32_t kmagic)
{
	return row - K * (int)__umulhi((uint32_t)row, kmagic);
}

/* smallest power of two a such that a * pb is a multiple of 16: ring windows start on such a column, so that the
 * LDS-DMA's global address and the RGBA identity reads are 16-byte aligned relative to the row start */
__host__ __device__ constexpr int move_col_align(int pb)
{
	return (pb % 16 == 0) ? 1 : (pb % 8 == 0) ? 2 : (pb % 4 == 0) ? 4 : 8;
}

template <int SB, int NCH, int NU>     /* SB: bytes per sample (2, 4, 8); NCH: samples per pixel; NU: KiB per ring row */
__global__ __launch_bounds__(64) void
k_move(MoveBatch imgs, FixcaGeom g, MovePlan mp)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	constexpr int PB = SB * NCH;                       /* bytes per pixel                               */
	constexpr int NS = 16 / SB;                        /* samples per unit                              */
	constexpr int WINB = NU * 1024;                    /* bytes per ring row                            */
	constexpr int NG = (NCH == 4) ? (SB == 8 ? 1 : NS / 2) : NS;     /* samples gathered per unit       */

	const int lane = threadIdx.x;
	const int W = g.W, H = g.H;
	/* a launch covers n independent images of one geometry: image k's tiles are workgroups [k nblocks, (k + 1) nblocks),
	 * handed out in order, so image k + 1's first waves start in the slots image k's last waves leave */
	const int img = blockIdx.x / mp.nblocks;
	const int tile = xcd_remap(blockIdx.x - img * mp.nblocks, mp.nblocks);
	if (tile >= mp.nwaves) return;
	const uint8_t *__restrict__ src = imgs.src[img];
	uint8_t *__restrict__ dst = imgs.dst[img];
	constexpr int RB = MOVE_RB(NU);                    /* output rows per block                         */
	const int strip = tile % mp.nstrips, chunk = tile / mp.nstrips;
	const int X0 = strip * mp.ocpx, xe = min(X0 + mp.ocpx, W);
	const int Y0 = g.y1 + chunk * mp.th;
	const int nrows = min(mp.th, g.y2 - Y0);
	const uint32_t pitch = (uint32_t)W * PB;
	const int K = mp.ring;
	const uint32_t kmagic = (uint32_t)((0x100000000ull + (unsigned)K - 1) / (unsigned)K);

	/* ---- the strip's window: the lowest column any consumer reads (the maps are monotone: the first column's) -------- */
	int lo = X0;
#pragma unroll
	for (int c = 0; c < 2; ++c) lo = min(lo, map_int(X0, g.xc, W, g.s[c], g.xs[c]));
	const int U0 = __builtin_amdgcn_readfirstlane(lo & ~(move_col_align(PB) - 1));

	/* ---- per-lane constants: where in a ring row each sample of my units comes from ------------------------------------ */
	const int nbytes = (xe - X0) * PB;                 /* bytes of the strip's output row (a multiple of 4) */
	const int nunits = (nbytes + 15) >> 4;
	const int tail_dw = (nbytes & 15) >> 2;            /* dwords of a last, partial unit (0: none)       */
	uint32_t goff[NU][NG];                             /* byte offset in a ring row of gathered sample j */
	uint32_t idoff[NU];                                /* RGBA: byte offset in a ring row of my unit      */
	uint32_t stoff[NU];                                /* byte offset in an output row of my unit, or out of range */
	int      chsel[NU];                                /* RGB: channel of sample 0 of the unit (0 R, 1 G, 2 B); RGBA double: 0 (R, G) or 2 (B, A) */
#pragma unroll
	for (int i = 0; i < NU; ++i) {
		const int U = min(lane + 64 * i, nunits - 1);  /* units past the strip copy the last one's addresses and store nothing */
		const int s0 = U * NS;                         /* first sample of the unit, counted from the strip's first pixel */
		idoff[i] = (uint32_t)((X0 - U0) * PB + 16 * U);
		chsel[i] = (NCH == 4) ? ((SB == 8) ? 2 * (U & 1) : 0) : s0 % 3;
#pragma unroll
		for (int j = 0; j < NG; ++j) {
			/* sample index within the unit of gathered sample j: RGBA u16 / 4-byte: the red and blue ones (0, 2, 4, 6);
			 * RGBA double: sample 0 (red in even units, blue in odd ones); RGB: every sample */
			const int k = (NCH == 4) ? ((SB == 8) ? 0 : 2 * j) : j;
			const int s = s0 + k;
			const int px = min(X0 + s / NCH, W - 1), ch = s % NCH;
			int sx = px;
			if (ch == 0) sx = map_int(px, g.xc, W, g.s[0], g.xs[0]);
			if (ch == 2) sx = map_int(px, g.xc, W, g.s[1], g.xs[1]);
			goff[i][j] = (uint32_t)((sx - U0) * PB + ch * SB);
		}
	}

	/* RGBA u16: are the two red and the two blue source samples of every unit 8 bytes apart? (decided per strip) */
	bool pair[2] = { false, false };
	if constexpr (NCH == 4 && SB == 2) {
		bool ok[2] = { true, true };
#pragma unroll
		for (int i = 0; i < NU; ++i) {
			ok[0] = ok[0] && goff[i][2] == goff[i][0] + 8;
			ok[1] = ok[1] && goff[i][3] == goff[i][1] + 8;
		}
		pair[0] = __builtin_amdgcn_ballot_w64(!ok[0]) == 0;
		pair[1] = __builtin_amdgcn_ballot_w64(!ok[1]) == 0;
	}

	/* ---- global memory through buffer descriptors of this tile's rows (32-bit offsets whatever the image size) -------- */
	int rowR_v = 0, rowB_v = 0;                        /* lane i: red / blue source row of output row (64-row segment base) + i */
	auto row_tables = [&](int r0) {
		const int y = min(Y0 + r0 + lane, g.y2 - 1);
		rowR_v = map_int(y, g.yc, H, g.s[0], g.ys[0]);           /* fix-ca.c:1105-1106 */
		rowB_v = map_int(y, g.yc, H, g.s[1], g.ys[1]);
	};
	row_tables(0);
	const int yl = Y0 + nrows - 1;
	/* Not forced into a scalar register (k_lin and k_cub do that): the compiler then keeps the source descriptor in vector
	 * registers and wraps every LDS-DMA instruction in a one-pass waterfall loop.  For the mover that measures the same or
	 * better than the plain instruction (one image per launch, RGBA float: 74.0-75.2 against 72.0-73.2 % on three boxes;
	 * batched u16 67.4 against 66.6 %; everything else within 0.2 points; pacing the issues with s_sleep does not
	 * reproduce it: profiles/r03_sweep_r3t_waterfall.txt), so it stays as the compiler builds it. */
	const int keep0 = min(Y0, min(__builtin_amdgcn_readlane(rowR_v, 0), __builtin_amdgcn_readlane(rowB_v, 0)));
	const int lastneed = max(yl, max(map_int(yl, g.yc, H, g.s[0], g.ys[0]), map_int(yl, g.yc, H, g.s[1], g.ys[1])));
	const int lastneed_s = __builtin_amdgcn_readfirstlane(lastneed);
	const uint32_t dst_bytes = (uint32_t)nrows * pitch;
	[[maybe_unused]] const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + (size_t)keep0 * pitch, 0,
	                                                        (uint32_t)(lastneed_s - keep0 + 1) * pitch, 0x00020000);
	const rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)Y0 * pitch, 0, dst_bytes, 0x00020000);
#pragma unroll
	for (int i = 0; i < NU; ++i) {
		const int U = lane + 64 * i;
		const bool whole = U < nunits - (tail_dw ? 1 : 0);
		stoff[i] = whole ? (uint32_t)(X0 * PB + 16 * U) : dst_bytes;
	}
	const int tail_u = (nunits - 1) >> 6, tail_lane = (nunits - 1) & 63;

	/* ---- the ring ------------------------------------------------------------------------------------------------------ */
	[[maybe_unused]] const uint32_t vo_ld = (uint32_t)U0 * PB + 16u * (uint32_t)lane;
	int ld = keep0 - 1;
	[[maybe_unused]] uint32_t so_ld = 0;      /* the host pass of hipcc sees no use of it */
	auto issue = [&]() {
		++ld;
		[[maybe_unused]] const int slot = move_slot(ld, K, kmagic);
#if defined(__HIP_DEVICE_COMPILE__)
		__attribute__((address_space(3))) uint8_t *row = (__attribute__((address_space(3))) uint8_t *)(ring + slot * WINB);
		/* the instruction offset applies to the global and to the LDS address alike: piece i is bytes 1024 i .. of the window */
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 0, 0);
		if constexpr (NU > 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 1024, 0);
		if constexpr (NU > 2) {
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 2048, 0);
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (__attribute__((address_space(3))) void *)row, 16, vo_ld, so_ld, 3072, 0);
		}
#endif
		so_ld += pitch;
	};
	/* rows in flight beyond `need`, within the ring's capacity; then wait until row `need` has landed.  vmcnt counts
	 * loads and stores; the wait relies on the loads alone (rows are loaded in order, NU loads each): stores in flight
	 * make it stricter, never wrong (fixca_roll.hip, roll_fetch) */
	auto fetch = [&](int keep, int need) {
		const int target = min(min(need + mp.depth, keep + K - 1), lastneed_s);
		while (ld < target) issue();
		const int younger = ld - need;
		if (younger >= 12) move_wait_vm<12 * NU>();                              /* vmcnt holds 6 bits: 12 x 4 = 48 */
		else if (younger >= 10) move_wait_vm<10 * NU>();
		else if (younger >= 8) move_wait_vm<8 * NU>();
		else if (younger >= 6) move_wait_vm<6 * NU>();
		else if (younger >= 4) move_wait_vm<4 * NU>();
		else if (younger >= 3) move_wait_vm<3 * NU>();
		else if (younger >= 2) move_wait_vm<2 * NU>();
		else if (younger >= 1) move_wait_vm<NU>();
		else move_wait_vm<0>();
	};

	uint32_t so_st = 0;
	auto rows_loop = [&](auto pair_tag) {
	constexpr bool PAIR = decltype(pair_tag)::value;       /* RGBA u16: both channels' samples of every unit are 8 bytes apart */
	for (int r = 0; r < nrows; r += RB) {
		if (r && (r & 63) == 0) row_tables(r);
		/* the block's rows; rows past the tile repeat its last row (their stores go nowhere) */
		int yy[RB], rR[RB], rB[RB];
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			const int rj = min(r + j, nrows - 1);
			yy[j] = Y0 + rj;
			rR[j] = __builtin_amdgcn_readlane(rowR_v, rj & 63);
			rB[j] = __builtin_amdgcn_readlane(rowB_v, rj & 63);
		}
		/* the maps are monotone: the block's oldest row is its first row's, its newest its last row's.
		 * Every LDS read of the previous block has returned (its stores consumed them): the slots below `keep` are free */
		fetch(min(yy[0], min(rR[0], rB[0])), max(yy[RB - 1], max(rR[RB - 1], rB[RB - 1])));
		u32x4 out[RB][NU];
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			const uint32_t bI = (uint32_t)move_slot(yy[j], K, kmagic) * WINB, bR = (uint32_t)move_slot(rR[j], K, kmagic) * WINB,
			               bB = (uint32_t)move_slot(rB[j], K, kmagic) * WINB;
#pragma unroll
			for (int i = 0; i < NU; ++i) {
				if constexpr (NCH == 4) {
					u32x4 v = *reinterpret_cast<const u32x4 *>(ring + bI + idoff[i]);
					if constexpr (SB == 2) {
						/* A unit is two pixels.  Where the source columns of its two red (blue) samples are neighbours - every
						 * unit of a strip without a scale step inside a unit - the two dwords that hold them are 8 bytes apart:
						 * one ds_read2_b32 instead of two 2-byte reads.  At a lane stride of 16 bytes an LDS read of 2 or 4
						 * bytes per lane runs into 4-way bank conflicts whatever its width, so this halves the LDS time of
						 * the gathers (the kernel's bottleneck on 16-bit samples: 40 -> 24 LDS cycles per unit and row). */
						uint32_t r0, r1, b0, b1;
						if constexpr (PAIR) {
							const u32x2pk t = *reinterpret_cast<const u32x2pk *>(ring + bR + goff[i][0]);
							const u32x2pk u = *reinterpret_cast<const u32x2pk *>(ring + bB + goff[i][1]);
							r0 = t.a; r1 = t.b; b0 = u.a; b1 = u.b;
						} else {
							r0 = *reinterpret_cast<const uint16_t *>(ring + bR + goff[i][0]);
							r1 = *reinterpret_cast<const uint16_t *>(ring + bR + goff[i][2]);
							b0 = *reinterpret_cast<const uint16_t *>(ring + bB + goff[i][1]);
							b1 = *reinterpret_cast<const uint16_t *>(ring + bB + goff[i][3]);
						}
						v.x = (v.x & 0xFFFF0000u) | (r0 & 0xFFFFu); v.y = (v.y & 0xFFFF0000u) | (b0 & 0xFFFFu);
						v.z = (v.z & 0xFFFF0000u) | (r1 & 0xFFFFu); v.w = (v.w & 0xFFFF0000u) | (b1 & 0xFFFFu);
					} else if constexpr (SB == 4) {
						v.x = *reinterpret_cast<const uint32_t *>(ring + bR + goff[i][0]);
						v.z = *reinterpret_cast<const uint32_t *>(ring + bB + goff[i][1]);
					} else {
						const u32x2 s = *reinterpret_cast<const u32x2 *>(ring + (chsel[i] ? bB : bR) + goff[i][0]);
						v.x = s.x; v.y = s.y;
					}
					out[j][i] = v;
				} else {
					/* channel of sample k: (chsel + k) mod 3; its row: red's, this one, blue's */
					uint32_t base[3];
#pragma unroll
					for (int q = 0; q < 3; ++q) {
						const int c = (chsel[i] + q) % 3;
						base[q] = (c == 0) ? bR : (c == 2) ? bB : bI;
					}
					uint32_t w[4];
					if constexpr (SB == 2) {
#pragma unroll
						for (int d = 0; d < 4; ++d) {
							const uint32_t a = *reinterpret_cast<const uint16_t *>(ring + base[(2 * d) % 3] + goff[i][2 * d]);
							const uint32_t b = *reinterpret_cast<const uint16_t *>(ring + base[(2 * d + 1) % 3] + goff[i][2 * d + 1]);
							w[d] = a | (b << 16);
						}
					} else if constexpr (SB == 4) {
#pragma unroll
						for (int d = 0; d < 4; ++d) w[d] = *reinterpret_cast<const uint32_t *>(ring + base[d % 3] + goff[i][d]);
					} else {
#pragma unroll
						for (int d = 0; d < 2; ++d) {
							const u32x2 s = *reinterpret_cast<const u32x2 *>(ring + base[d] + goff[i][d]);
							w[2 * d] = s.x; w[2 * d + 1] = s.y;
						}
					}
					out[j][i] = u32x4{ w[0], w[1], w[2], w[3] };
				}
			}
		}
		/* The row offset goes into the vector offset, the scalar offset stays the constant 0: with a register there the
		 * compiler leaves no wait state between a 16-byte store and the next write of its data registers, and on MI355X the
		 * store then carries the next row's bytes (profiles/NOTES.md, "a hardware hazard"; tests/test_host_logic.py scans for it) */
#pragma unroll
		for (int j = 0; j < RB; ++j) {
			const bool live = r + j < nrows;                /* wave-uniform */
#pragma unroll
			for (int i = 0; i < NU; ++i)
				__builtin_amdgcn_raw_buffer_store_b128(out[j][i], rs_dst, live ? stoff[i] + so_st : dst_bytes, 0, 2);
			if (tail_dw) {                                  /* wave-uniform: a strip whose row does not end on a unit boundary */
#pragma unroll
				for (int i = 0; i < NU; ++i)
					if (i == tail_u) {
						const bool mine = live && lane == tail_lane;
						const uint32_t o = (uint32_t)(X0 * PB + 16 * (lane + 64 * i)) + so_st;
#pragma unroll
						for (int d = 0; d < 3; ++d)
							if (d < tail_dw) __builtin_amdgcn_raw_buffer_store_b32(out[j][i][d], rs_dst, mine ? o + 4 * d : dst_bytes, 0, 2);
					}
			}
			so_st += pitch;
		}
		/* the next block overwrites ring slots: every LDS read above must have returned.  The stores consume the
		 * data, so the compiler has waited already; the explicit wait costs nothing and does not depend on that */
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	};
	if constexpr (NCH == 4 && SB == 2) {
		if (pair[0] && pair[1]) rows_loop(std::true_type{});
		else rows_loop(std::false_type{});
	} else {
		rows_loop(std::false_type{});
	}
}

template <int SB, int NCH>
const void *move_kernel_nu(int nu)
{
	switch (nu) {
	case 1:  return reinterpret_cast<const void *>(&k_move<SB, NCH, 1>);
	case 2:  return reinterpret_cast<const void *>(&k_move<SB, NCH, 2>);
	default: return reinterpret_cast<const void *>(&k_move<SB, NCH, 4>);
	}
}

const void *move_kernel_for(const FixcaGeom *g, int nu)
{
	const int sb = g->bpc < 0 ? -g->bpc : g->bpc;
	if (g->nch == 4) {
		if (sb == 2) return move_kernel_nu<2, 4>(nu);
		if (sb == 4) return move_kernel_nu<4, 4>(nu);
		if (sb == 8) return move_kernel_nu<8, 4>(nu);
	} else if (g->nch == 3) {
		if (sb == 2) return move_kernel_nu<2, 3>(nu);
		if (sb == 4) return move_kernel_nu<4, 3>(nu);
		if (sb == 8) return move_kernel_nu<8, 3>(nu);
	}
	return nullptr;
}

}   /* namespace */

extern "C" hipError_t
fixca_launch_move_batch(const void *const *d_src, void *const *d_dst, int n_images, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	if (g->y2 <= g->y1) return hipSuccess;
	if (n_images < 1 || n_images > MOVE_BATCH_MAX) return hipErrorInvalidValue;
	if (g->interp != 0 || (pl->nu != 1 && pl->nu != 2 && pl->nu != 4) || pl->depth < 1 || pl->depth > MOVE_DEPTH_MAX) return hipErrorInvalidValue;
	const void *fn = move_kernel_for(g, pl->nu);
	if (!fn) return hipErrorInvalidValue;
	const size_t lds = (size_t)pl->ring * pl->nu * 1024 + 64;      /* + a pad: the 16-byte read of a strip's last, partial unit may end past the last ring row */
	if (lds > MOVE_LDS_MAX + 64) return hipErrorInvalidValue;
	/* dynamic LDS above the default limit needs the attribute once per device and function */
	hipError_t ea = (hipError_t)fixca_dyn_lds(fn, MOVE_LDS_MAX + 64);
	if (ea != hipSuccess) return ea;
	MovePlan mp;
	mp.ocpx = pl->oc; mp.th = pl->th; mp.ring = pl->ring; mp.depth = pl->depth; mp.nstrips = pl->nstrips; mp.nwaves = pl->nwaves; mp.nblocks = pl->nblocks;
	mp.edge_wb = 0;
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
fixca_launch_move(const void *d_src, void *d_dst, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	return fixca_launch_move_batch(&d_src, &d_dst, 1, g, pl, stream);
}
