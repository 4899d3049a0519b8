/* ntx_oracle.c -- plain-C restatement of the NeRF-Tex render path (float and double builds of the
 * same body).  TEST INFRASTRUCTURE ONLY: loaded by tests/ through ctypes to cross-check the numpy
 * restatement in oracle/nerftex_oracle.py.  Never linked into, or called by, the product. */
#include <math.h>
#include <stdlib.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL double
#define FN(name) CAT(name, _f64)
#define SIN sin
#define COS cos
#define EXP exp
#define SQRT sqrt
#include "ntx_oracle_impl.h"
#undef REAL
#undef FN
#undef SIN
#undef COS
#undef EXP
#undef SQRT

#define REAL float
#define FN(name) CAT(name, _f32)
#define SIN sinf
#define COS cosf
#define EXP expf
#define SQRT sqrtf
#include "ntx_oracle_impl.h"
