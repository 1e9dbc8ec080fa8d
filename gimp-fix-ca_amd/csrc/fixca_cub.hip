/*
 * fixca_cub.hip — CUBIC on RGBA float and double images in the structure of the NONE mover and of k_lin
 * (fixca_move.hip, fixca_lin.hip): an LDS-DMA ring of raw source rows, 16-byte units per lane, nt stores; the
 * reference's Catmull-Rom arithmetic (cubicY across columns, cubicX across rows, fix-ca.c:896-920; get_pixel /
 * set_pixel :713-774) in fp64, operation for operation: bit-identical to the CPU oracle.
 *
 * k_wcol (fixca_wide.hip) does the same arithmetic; what differs is how the data moves.  k_wcol shares every
 * converted source sample across the wave through LDS and holds its loads in a register queue two rows ahead: a wave
 * that waits for its stores to be acknowledged has nothing else in flight.  Here source rows arrive by LDS-DMA
 * `depth` rows ahead (no registers, no coupling to the stores), and each lane works for itself:
 *
 *  - a lane owns NU 16-byte units of the strip's output row: one pixel (float: red and blue computed) or half a
 *    pixel (double: red, or blue);
 *  - per computed sample the 4 tap columns clamp(c0 - 1 + t) (fix-ca.c:1270-1298) are per-lane ring offsets for the
 *    whole tile, with the fraction fx;
 *  - a ROLLING WINDOW PER LANE: the 4 horizontally filtered rows of every computed sample live in registers (8 doubles
 *    per RGBA float pixel).  When the output row advances and both channels' windows slide by one source row - every
 *    row, except at a scale step of the row map - only the new bottom row is filtered (4 LDS reads, 4 get_pixel,
 *    one cubicY per sample); then cubicX on the window, clip_d, set_pixel.  Otherwise (the first row of a tile, a row
 *    map that repeats or skips a row) the windows are rebuilt from the ring: the rows are clamp(y0 - 1 + t)
 *    (fix-ca.c:1218-1256), and clamping is monotone, so the slide is valid at the image edges too.  Re-using the value
 *    of an expression is not a change of rounding: every operation sees the operands the reference's would.
 *
 * Where it pays (measured against k_wcol on the same buffers, profiles/r03_sweep_r3o.txt): CUBIC is bound by its
 * dependent fp64 chains (two cubics of 19 operations per sample), and a ring of 16 KiB per wave leaves 9 waves per CU
 * where k_wcol runs 8-12 with four chains per lane.  RGBA double: 64.5 % of the HBM peak against 58.0 % (six images per
 * launch), 61.0 against 57.9 (one image); RGBA float: 55.6 against 51.2 % batched, 48.7 against 52.1 one image per
 * launch (profiles/r03_sweep_r3p.txt).  16-bit samples (also in the weight form) and u32 lose 2-8 points and stay with k_wcol: the planner
 * (plan_cub) sends doubles here always and floats for batches.
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

template <int N> __device__ __forceinline__ void cub_wait_vm()
{
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int cub_slot(int row, int K, uint32_t kmagic)
{
	return row - K * (int)__umulhi((uint32_t)row, kmagic);
}
__device__ __forceinline__ double cub_readlane_f64(double v, int l)
{
	const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
	const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)b, l), hi = __builtin_amdgcn_readlane((uint32_t)(b >> 32), l);
	return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ int cub_clamp(int v, int n) { return max(0, min(v, n - 1)); }

template <typename S, int NU>
__global__ __launch_bounds__(64) void
k_cub(MoveBatch imgs, FixcaGeom g, MovePlan mp)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t ring[];
	typedef Codec<S> C;
	static_assert(sizeof(S) == 4 || sizeof(S) == 8, "float and double (u32 compiles too); 16-bit samples stay with k_wcol");
	constexpr int SB = (int)sizeof(S);
	constexpr int PB = 4 * SB;                         /* RGBA                                          */
	constexpr int WINB = NU * 1024;
	constexpr int NC = (SB == 4) ? 2 : 1;              /* samples computed per unit: (R B) / (R | B)     */

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

	/* the strip's window starts at the lowest column any consumer reads: the strip's first column or the first tap
	 * of its first column (the maps are monotone), rounded down to a 16-byte boundary of the row */
	int lo = X0;
#pragma unroll
	for (int c = 0; c < 2; ++c) lo = min(lo, cub_clamp((int)floor(map_dbl(X0, g.xc, W, g.s[c], g.xs[c])) - 1, W));
	constexpr int CA = 1;                              /* 16- and 32-byte pixels: every column starts a unit */
	const int U0 = __builtin_amdgcn_readfirstlane(lo & ~(CA - 1));

	/* ---- per-lane constants ---------------------------------------------------------------------------------------- */
	const int nunits = ((xe - X0) * PB + 15) >> 4;
	uint32_t off[NU][NC][4];                           /* byte offsets in a ring row of the 4 taps' samples              */
	double   fx[NU][NC];
	uint32_t idoff[NU], stoff[NU];
	int      blue[NU];                                 /* double: this unit holds (B, A), not (R, G)                     */
#pragma unroll
	for (int i = 0; i < NU; ++i) {
		const int U = min(lane + 64 * i, nunits - 1);
		idoff[i] = (uint32_t)((X0 - U0) * PB + 16 * U);
		blue[i] = (SB == 8) ? (U & 1) : 0;
#pragma unroll
		for (int j = 0; j < NC; ++j) {
			const int c = (SB == 8) ? blue[i] : (j & 1);
			const int px = min(X0 + ((SB == 8) ? U / 2 : U), W - 1);
			const double xd = map_dbl(px, g.xc, W, g.s[c], g.xs[c]);
			const int c0 = (int)floor(xd);
			fx[i][j] = __dsub_rn(xd, (double)c0);
#pragma unroll
			for (int t = 0; t < 4; ++t)                                      /* fix-ca.c:1270-1298: clamp(c0 - 1 + t) */
				off[i][j][t] = (uint32_t)((cub_clamp(c0 - 1 + t, W) - U0) * PB + 2 * c * SB);
		}
	}

	/* ---- rows: lane i of a 64-row segment holds output row (segment + i)'s first window row and fraction ---------------- */
	int    y0_v[2] = { 0, 0 };
	double fy_v[2] = { 0.0, 0.0 };
	auto row_tables = [&](int rseg) {
		const int y = min(Y0 + rseg + lane, g.y2 - 1);
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			const double yd = map_dbl(y, g.yc, H, g.s[c], g.ys[c]);
			y0_v[c] = (int)floor(yd);
			fy_v[c] = __dsub_rn(yd, (double)y0_v[c]);
		}
	};
	row_tables(0);
	const int yl = Y0 + nrows - 1;
	/* readfirstlane: the value is wave-uniform, but the compiler computes it on the vector unit (it moves the readlane up
	 * to the lane id and redoes the fp64 map there), builds the source descriptor in vector registers and then wraps every
	 * LDS-DMA instruction in a waterfall loop (4 readfirstlane, 2 compares, saveexec, branch per issue) */
	const int keep0 = __builtin_amdgcn_readfirstlane(min(Y0, min(cub_clamp(__builtin_amdgcn_readlane(y0_v[0], 0) - 1, H), cub_clamp(__builtin_amdgcn_readlane(y0_v[1], 0) - 1, H))));
	int lastneed = yl;
#pragma unroll
	for (int c = 0; c < 2; ++c) lastneed = max(lastneed, cub_clamp((int)floor(map_dbl(yl, g.yc, H, g.s[c], g.ys[c])) + 2, H));
	const int lastneed_s = __builtin_amdgcn_readfirstlane(lastneed);
	const uint32_t dst_bytes = (uint32_t)nrows * pitch;
	const rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + (size_t)keep0 * pitch, 0,
	                                                        (uint32_t)(lastneed_s - keep0 + 1) * pitch, 0x00020000);
	const rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)Y0 * pitch, 0, dst_bytes, 0x00020000);
	(void)rs_src;
#pragma unroll
	for (int i = 0; i < NU; ++i) {
		const int U = lane + 64 * i;
		stoff[i] = (U < nunits) ? (uint32_t)(X0 * PB + 16 * U) : dst_bytes;
	}

	/* ---- the ring (as in k_move) ---------------------------------------------------------------------------------------- */
	const uint32_t vo_ld = (uint32_t)U0 * PB + 16u * (uint32_t)lane;
	(void)vo_ld;
	int ld = keep0 - 1;
	[[maybe_unused]] uint32_t so_ld = 0;      /* the host pass of hipcc sees no use of it */
	auto issue = [&]() {
		++ld;
#if defined(__HIP_DEVICE_COMPILE__)
		const int slot = cub_slot(ld, K, kmagic);
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
		if (younger >= 12) cub_wait_vm<12 * NU>();
		else if (younger >= 10) cub_wait_vm<10 * NU>();
		else if (younger >= 8) cub_wait_vm<8 * NU>();
		else if (younger >= 6) cub_wait_vm<6 * NU>();
		else if (younger >= 4) cub_wait_vm<4 * NU>();
		else if (younger >= 3) cub_wait_vm<3 * NU>();
		else if (younger >= 2) cub_wait_vm<2 * NU>();
		else if (younger >= 1) cub_wait_vm<NU>();
		else cub_wait_vm<0>();
	};

	/* Away from the left and right edge of the image the four taps of a sample are four consecutive pixels: one address and
	 * constant offsets (the compiler pairs the reads: ds_read2) instead of four addresses.  Wave-uniform: a tile in which
	 * any lane has a clamped tap takes the general form for all of its rows. */
	bool contig_l = true;
#pragma unroll
	for (int i = 0; i < NU; ++i)
#pragma unroll
		for (int j = 0; j < NC; ++j)
#pragma unroll
			for (int t = 1; t < 4; ++t) contig_l = contig_l && off[i][j][t] == off[i][j][0] + (uint32_t)(t * PB);
	const bool contig = __builtin_amdgcn_ballot_w64(!contig_l) == 0;

	/* cubicY of one source row for computed sample (i, j): the row is given by its ring base (bR / bB: red's / blue's) */
	auto hrow = [&](auto contig_tag, int i, int j, uint32_t bR, uint32_t bB) -> double {
		constexpr bool CONTIG = decltype(contig_tag)::value;
		const uint32_t base = (SB == 8) ? (blue[i] ? bB : bR) : ((j & 1) ? bB : bR);
		S s[4];
		if constexpr (CONTIG) {
			const S *p = reinterpret_cast<const S *>(ring + base + off[i][j][0]);
#pragma unroll
			for (int t = 0; t < 4; ++t) s[t] = p[t * 4];                                         /* 4 samples per RGBA pixel */
		} else {
#pragma unroll
			for (int t = 0; t < 4; ++t) s[t] = *reinterpret_cast<const S *>(ring + base + off[i][j][t]);
		}
		return cr_exact(C::get(s[0]), C::get(s[1]), C::get(s[2]), C::get(s[3]), fx[i][j]);              /* fix-ca.c:905-907 */
	};

	auto rows_loop = [&](auto contig_tag) {
	double hh[NU][NC][4];                              /* the rolling windows: 4 horizontally filtered rows per computed sample */
	int    y0prev[2] = { -1000000, -1000000 };
	uint32_t so_st = 0;
	for (int r = 0; r < nrows; ++r) {
		if (r && (r & 63) == 0) row_tables(r);
		const int y = Y0 + r;
		int y0[2];
		double fy[2];
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			y0[c] = __builtin_amdgcn_readlane(y0_v[c], r & 63);
			fy[c] = cub_readlane_f64(fy_v[c], r & 63);
		}
		/* every LDS read of the previous row has returned (its stores consumed them): the slots below `keep` are free */
		const int keep = min(y, min(cub_clamp(y0[0] - 1, H), cub_clamp(y0[1] - 1, H)));
		const int need = max(y, max(cub_clamp(y0[0] + 2, H), cub_clamp(y0[1] + 2, H)));
		fetch(keep, need);
		const uint32_t bI = (uint32_t)cub_slot(y, K, kmagic) * WINB;
		if (__builtin_expect(y0[0] == y0prev[0] + 1 && y0[1] == y0prev[1] + 1, 1)) {
			/* both windows slide by one source row: row t of the new window is row t + 1 of the old one */
			const uint32_t bR = (uint32_t)cub_slot(cub_clamp(y0[0] + 2, H), K, kmagic) * WINB;
			const uint32_t bB = (uint32_t)cub_slot(cub_clamp(y0[1] + 2, H), K, kmagic) * WINB;
#pragma unroll
			for (int i = 0; i < NU; ++i)
#pragma unroll
				for (int j = 0; j < NC; ++j) {
					const double nw = hrow(contig_tag, i, j, bR, bB);
					hh[i][j][0] = hh[i][j][1]; hh[i][j][1] = hh[i][j][2]; hh[i][j][2] = hh[i][j][3]; hh[i][j][3] = nw;
				}
		} else {
			/* first row of the tile, or a row map that repeats or skips a source row: rebuild the windows */
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				const uint32_t bR = (uint32_t)cub_slot(cub_clamp(y0[0] - 1 + t, H), K, kmagic) * WINB;
				const uint32_t bB = (uint32_t)cub_slot(cub_clamp(y0[1] - 1 + t, H), K, kmagic) * WINB;
#pragma unroll
				for (int i = 0; i < NU; ++i)
#pragma unroll
					for (int j = 0; j < NC; ++j) hh[i][j][t] = hrow(contig_tag, i, j, bR, bB);
			}
		}
		y0prev[0] = y0[0]; y0prev[1] = y0[1];

		u32x4 out[NU];
#pragma unroll
		for (int i = 0; i < NU; ++i) {
			u32x4 v = *reinterpret_cast<const u32x4 *>(ring + bI + idoff[i]);
			S res[NC];
#pragma unroll
			for (int j = 0; j < NC; ++j) {
				const double f = (SB == 8) ? (blue[i] ? fy[1] : fy[0]) : fy[j & 1];
				res[j] = C::put(clip01(cr_exact(hh[i][j][0], hh[i][j][1], hh[i][j][2], hh[i][j][3], f)));       /* fix-ca.c:916-919 */
			}
			if constexpr (SB == 4) {
				v.x = __builtin_bit_cast(uint32_t, res[0]);
				v.z = __builtin_bit_cast(uint32_t, res[1]);
			} else {
				const unsigned long long bits = __builtin_bit_cast(unsigned long long, res[0]);
				v.x = (uint32_t)bits; v.y = (uint32_t)(bits >> 32);
			}
			out[i] = v;
		}
		/* scalar offset the constant 0: the store-data hazard of profiles/NOTES.md (tests/test_host_logic.py scans for it) */
#pragma unroll
		for (int i = 0; i < NU; ++i)
			__builtin_amdgcn_raw_buffer_store_b128(out[i], rs_dst, stoff[i] + so_st, 0, 2);
		so_st += pitch;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* the next row's DMA may overwrite what was read here */
	}
	};
	if (contig) rows_loop(std::true_type{});
	else        rows_loop(std::false_type{});
}

template <typename S>
const void *cub_kernel_nu(int nu)
{
	switch (nu) {
	case 1:  return reinterpret_cast<const void *>(&k_cub<S, 1>);
	case 2:  return reinterpret_cast<const void *>(&k_cub<S, 2>);
	default: return reinterpret_cast<const void *>(&k_cub<S, 4>);
	}
}

const void *cub_kernel_for(const FixcaGeom *g, int nu)
{
	if (g->nch != 4) return nullptr;
	switch (g->bpc) {
	case -4: return cub_kernel_nu<float>(nu);
	case -8: return cub_kernel_nu<double>(nu);
	default: return nullptr;
	}
}

}   /* namespace */

extern "C" hipError_t
fixca_launch_cub_batch(const void *const *d_src, void *const *d_dst, int n_images, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	if (g->y2 <= g->y1) return hipSuccess;
	if (n_images < 1 || n_images > MOVE_BATCH_MAX) return hipErrorInvalidValue;
	if (g->interp != 2 || (pl->nu != 1 && pl->nu != 2 && pl->nu != 4) || pl->depth < 1 || pl->depth > MOVE_DEPTH_MAX) return hipErrorInvalidValue;
	const void *fn = cub_kernel_for(g, pl->nu);
	if (!fn) return hipErrorInvalidValue;
	const size_t lds = (size_t)pl->ring * pl->nu * 1024 + 64;
	if (lds > MOVE_LDS_MAX + 64) return hipErrorInvalidValue;
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
fixca_launch_cub(const void *d_src, void *d_dst, const FixcaGeom *g, const FixcaPlan *pl, hipStream_t stream)
{
	return fixca_launch_cub_batch(&d_src, &d_dst, 1, g, pl, stream);
}
