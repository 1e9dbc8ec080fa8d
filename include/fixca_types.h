/*
 * fixca_types.h — the parameter block and constants shared by every layer
 * (oracle, C host mirror, C-ABI shim, HIP kernels).
 *
 * Layout contract: FixCaParams is byte-for-byte the reference's struct
 * (reference fix-ca.c:70-82; defaults fix-ca.c:85-97): 80 bytes, natural
 * alignment, so a pointer to the plug-in's own struct can be handed straight
 * through the C-ABI.  glib's gdouble/gboolean/enum are double/int/int.
 */
#ifndef FIXCA_TYPES_H
#define FIXCA_TYPES_H

#ifdef __cplusplus
extern "C" {
#endif

/* interpolation codes == GimpInterpolationType NONE/LINEAR/CUBIC (fix-ca.c:156) */
enum {
	FIXCA_INTERP_NONE   = 0,
	FIXCA_INTERP_LINEAR = 1,
	FIXCA_INTERP_CUBIC  = 2
};

/*
 * A translation unit that declares FixCaParams itself -- the plug-in's fix-ca.c does
 * (fix-ca.c:70-82) -- defines FIXCA_TYPES_NO_PARAMS before including this header: the struct is
 * then not declared here and the entry points take `const struct FixCaParams_ *`, an opaque
 * type, so the plug-in passes a cast pointer to its own (byte-identical) struct
 * (tools/gimp_box/apply_binding.py).
 */
#ifdef FIXCA_TYPES_NO_PARAMS
struct FixCaParams_;
#define FIXCA_PARAMS_T struct FixCaParams_
#else
#define FIXCA_PARAMS_T FixCaParams
typedef struct FixCaParams {
	double blue;            /* lateral blue amount, pixels at max_dim      */
	double red;             /* lateral red amount                          */
	double lens_x;          /* lens centre x (truncated to int in the path)*/
	double lens_y;          /* lens centre y                               */
	int    update_preview;  /* dialog only; ignored by the hot path        */
	int    interpolation;   /* FIXCA_INTERP_*                              */
	double saturation;      /* preview only (fix-ca.c:1323)                */
	double x_blue;          /* directional shifts                          */
	double x_red;
	double y_blue;
	double y_red;
} FixCaParams;
#endif

/* fix-ca.c:63-67.  INPUT_MAX is SOURCE_ROWS/4 == 30 wherever it is used. */
#define FIXCA_SOURCE_ROWS 120
#define FIXCA_INPUT_MAX   30

/* bpc codes produced by the reference's color_size() (fix-ca.c:681-711):
 * 1,2,4,8 = unsigned integer bytes per channel; -4 float; -8 double. */

#ifdef __cplusplus
}
#endif

#ifndef FIXCA_TYPES_NO_PARAMS
#if defined(__cplusplus)
static_assert(sizeof(FixCaParams) == 80, "FixCaParams must stay 80 bytes");
#else
_Static_assert(sizeof(FixCaParams) == 80, "FixCaParams must stay 80 bytes");
#endif
#endif

#endif /* FIXCA_TYPES_H */
