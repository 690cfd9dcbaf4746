/*
 * ORACLE (test infrastructure, never shipped / never on the product path): the FLOAT64 build of sdf_grid.c -- the arbiter of the
 * parity tests.  Two float32 implementations of 200 Adam steps (the HIP path and the float32 oracle) end a few 1e-5 m apart because
 * the update m / sqrt(v) amplifies summation-order rounding; the same algorithm in double precision says how far EACH of them is
 * from the exact trajectory (tests/test_gpu_parity.py::test_opt_headline_workload_matches_oracle, bench.py: parity.vs_f64).
 * Same source, same operation order: `float` and the float math functions are renamed before the one include (the system
 * headers are included first, untouched).  Exports ihmr_oracle_sdf_grid_f64 / ..._point_tri_dist2_f64 / ..._ray_hit_px_f64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define float double
#define fmaf fma
#define fabsf fabs
#define sqrtf sqrt
#define ihmr_oracle_sdf_grid ihmr_oracle_sdf_grid_f64
#define ihmr_oracle_point_tri_dist2 ihmr_oracle_point_tri_dist2_f64
#define ihmr_oracle_ray_hit_px ihmr_oracle_ray_hit_px_f64
#include "sdf_grid.c"
