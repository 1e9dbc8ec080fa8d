/* fixca_internal.h — shared between the planner and the C-ABI translation unit. */
#ifndef FIXCA_INTERNAL_H
#define FIXCA_INTERNAL_H

#include <stddef.h>
#include "../../include/fixca_types.h"
#include "fixca_geom.h"

/* records the message for fixca_last_error() and returns code */
int fixca_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

int  fixca_make_geom(int W, int H, int bytes, int bpc, const FixCaParams *p,
                     int x1, int x2, int y1, int y2, int show_progress, FixcaGeom *g);
int  fixca_make_plan(const FixcaGeom *g, unsigned flags, FixcaPlan *pl);
/* Tuning and experiment knobs (FIXCA_TH, FIXCA_OC, FIXCA_RING, FIXCA_ABLATE, FIXCA_NO_ROLL, ...) are
 * honoured only when FIXCA_DEBUG is set to a non-zero value: a stray variable in the environment
 * of a GIMP session must not change the kernels' shape.  Returns dflt otherwise. */
int  fixca_debug_env(const char *name, int dflt, int lo, int hi);
bool fixca_debug_flag(const char *name);
/* Dynamic LDS above the 64-KiB default needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel, and the attribute
 * belongs to the (device, function) pair: asked once per pair, whatever device the calling thread has selected.
 * Returns a hipError_t (as int, so that plain C++ units can include this header). */
int  fixca_dyn_lds(const void *fn, int bytes);
/* 1 when the current device moves dwords at any byte address faithfully (probed once per device, fixca_abi.hip), or when
 * there is no device to ask */
extern "C" int fixca_unaligned_ok(void);
void fixca_plan_text(const FixcaGeom *g, const FixcaPlan *pl, char *buf, size_t n);

#endif
