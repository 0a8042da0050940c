/*
 * TEST INFRASTRUCTURE ONLY -- see lago_oracle_impl.h.
 *
 * CPU oracle for the lagomorph deform/interp/metric/adjrep hot path: a scalar,
 * single-threaded C restatement of the reference's CUDA kernels
 * (/root/reference/lagomorph/extension/cuda/{interp,diff,metric,affine}.cu and
 * include/{interp,extrap,diff}.h), instantiated for float and double.
 *
 * Pinning (see oracle/README.md): the known-answer table of SURVEY.md 8(c)
 * (tests/golden/kat_survey.json), the reference's own property tests
 * (testing/test_*.py restated in tests/test_oracle_properties.py), and
 * bit-exact agreement of oracle_affine_interp_forward_cpuref_* with
 * oracle/_ref (the reference's cpu/affine.cpp compiled where it lies).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -mfma: only the explicit LG_FMA sites fuse)
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* Contraction switch, see lago_oracle_impl.h. */
/* Optional OpenMP over independent work items -- output voxels of the forward kernels (interp,
 * jacobian-times-vectorfield, fluid operator), batch items of interp_backward -- so the results do
 * not depend on the thread count.  Only bench.py's cpu_baseline leg raises it above 1. */
static int lg_oracle_threads = 1;
void oracle_set_threads(int n) { lg_oracle_threads = n > 0 ? n : 1; }
int oracle_get_threads(void) { return lg_oracle_threads; }
#ifdef _OPENMP
#define LG_PRAGMA(x) _Pragma(#x)
#define LG_PARALLEL_FOR LG_PRAGMA(omp parallel for schedule(static) num_threads(lg_oracle_threads) if (lg_oracle_threads > 1))
#define LG_PARALLEL_FOR_IF(cond) LG_PRAGMA(omp parallel for schedule(dynamic, 1) num_threads(lg_oracle_threads) if (lg_oracle_threads > 1 && (cond)))
#else
#define LG_PARALLEL_FOR
#define LG_PARALLEL_FOR_IF(cond)
#endif

#ifdef LAGO_ORACLE_STRICT
#define LG_FMAD(a, b, c) ((a) * (b) + (c))
#else
#define LG_FMAD(a, b, c) fma((a), (b), (c))
#endif

#define REAL float
#define SUF _f32
#define LG_SQRT sqrtf
#ifdef LAGO_ORACLE_STRICT
#define LG_FMA(a, b, c) ((a) * (b) + (c))
#else
#define LG_FMA(a, b, c) fmaf((a), (b), (c))
#endif
#include "lago_oracle_impl.h"
#undef REAL
#undef SUF
#undef LG_SQRT
#undef LG_FMA

#define REAL double
#define SUF _f64
#define LG_SQRT sqrt
#ifdef LAGO_ORACLE_STRICT
#define LG_FMA(a, b, c) ((a) * (b) + (c))
#else
#define LG_FMA(a, b, c) fma((a), (b), (c))
#endif
#include "lago_oracle_impl.h"
#undef REAL
#undef SUF
#undef LG_SQRT
#undef LG_FMA

#ifdef LAGO_ORACLE_STRICT
const char *oracle_version(void) { return "lago-oracle 2 (scalar C, 1 thread, unfused a*b+c)"; }
#else
const char *oracle_version(void) { return "lago-oracle 2 (scalar C, 1 thread, documented FMA contraction)"; }
#endif
