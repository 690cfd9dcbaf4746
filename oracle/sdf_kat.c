/*
 * ORACLE (test infrastructure): known-answer tests of sdf_grid.c as a stand-alone program, so that the oracle's C code can run under
 * AddressSanitizer / UndefinedBehaviorSanitizer (`make -C oracle asan`): the Python tests load the oracle as a shared library into
 * an uninstrumented interpreter, where a sanitizer cannot follow.  The same known answers as tests/test_oracle_golden.py's C-side
 * cases: hand-computed point-triangle distances (every Voronoi region), hand-computed +x ray cases (inside, edge, behind,
 * degenerate), and the grid of a closed convex mesh (an octahedron of radius 0.8): depth at the centre within one voxel of the
 * exact inradius, zero outside, every inside voxel's value = min over faces of the plane distance.
 */
#include <stdio.h>
#include "sdf_grid.c"

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); ++fails; } } while (0)

int main(void) {
    /* ---- point-triangle distance: triangle (0,0,0) (1,0,0) (0,1,0) */
    const float a[3] = {0, 0, 0}, b[3] = {1, 0, 0}, c[3] = {0, 1, 0};
    struct { float p[3], d2; const char* what; } pt[] = {
        {{0.25f, 0.25f, 0.5f}, 0.25f, "face interior"},     {{-1.f, -1.f, 0.f}, 2.0f, "vertex a"},
        {{2.f, -1.f, 0.f}, 2.0f, "vertex b"},               {{-1.f, 2.f, 0.f}, 2.0f, "vertex c"},
        {{0.5f, -1.f, 0.f}, 1.0f, "edge ab"},               {{-1.f, 0.5f, 0.f}, 1.0f, "edge ac"},
        {{1.f, 1.f, 0.f}, 0.5f, "edge bc"},                 {{0.25f, 0.25f, 0.f}, 0.0f, "on the face"}};
    for (unsigned i = 0; i < sizeof(pt) / sizeof(pt[0]); ++i) {
        const float d2 = ihmr_oracle_point_tri_dist2(a, b, c, pt[i].p);
        CHECK(fabsf(d2 - pt[i].d2) <= 1e-6f, "point_tri_dist2 %s: %g, expected %g", pt[i].what, d2, pt[i].d2);
    }
    /* ---- +x ray against the triangle (1,-1,-1) (1,1,-1) (1,0,1) in the plane x = 1 */
    const float ra[3] = {1, -1, -1}, rb[3] = {1, 1, -1}, rc[3] = {1, 0, 1};
    struct { float p[3]; int hit; const char* what; } ray[] = {
        {{0.f, 0.f, 0.f}, 1, "through the interior"},       {{2.f, 0.f, 0.f}, 0, "triangle behind the origin"},
        {{0.f, 5.f, 0.f}, 0, "misses in y"},                 {{0.f, 0.f, -1.f}, 1, "on the edge ab (u + v <= 1 inclusive)"},
        {{1.f, 0.f, 0.f}, 0, "origin in the plane: t = 0 is not a hit"}};
    for (unsigned i = 0; i < sizeof(ray) / sizeof(ray[0]); ++i)
        CHECK(ihmr_oracle_ray_hit_px(ra, rb, rc, ray[i].p) == ray[i].hit, "ray %s", ray[i].what);
    const float da[3] = {1, 0, 0}, db[3] = {2, 0, 0}, dc[3] = {3, 0, 0};                 /* degenerate in yz: never counted */
    const float dp[3] = {0, 0, 0};
    CHECK(ihmr_oracle_ray_hit_px(da, db, dc, dp) == 0, "degenerate triangle");
    /* ---- grid of an octahedron |x| + |y| + |z| <= R: phi = (R - |x| - |y| - |z|) / sqrt(3) inside, 0 outside */
    const float R = 0.8f;
    const float V[6 * 3] = {R, 0, 0, -R, 0, 0, 0, R, 0, 0, -R, 0, 0, 0, R, 0, 0, -R};
    const int32_t F[8 * 3] = {0, 2, 4, 2, 1, 4, 1, 3, 4, 3, 0, 4, 2, 0, 5, 1, 2, 5, 3, 1, 5, 0, 3, 5};
    const int G = 32;
    float* phi = (float*)malloc(sizeof(float) * G * G * G);
    ihmr_oracle_sdf_grid(V, F, 1, 6, 8, G, phi);
    int inside = 0;
    for (int k = 0; k < G; ++k)
        for (int j = 0; j < G; ++j)
            for (int i = 0; i < G; ++i) {
                const float px = (float)(2 * i + 1) / G - 1.0f, py = (float)(2 * j + 1) / G - 1.0f, pz = (float)(2 * k + 1) / G - 1.0f;
                const float s = R - fabsf(px) - fabsf(py) - fabsf(pz);
                const float got = phi[(k * G + j) * G + i];
                if (fabsf(s) < 1e-4f) continue;                                       /* (a voxel centre on the surface: either answer) */
                if (s > 0) { ++inside; CHECK(fabsf(got - s / sqrtf(3.0f)) <= 2e-6f, "octahedron depth at (%d,%d,%d): %g vs %g", k, j, i, got, s / sqrtf(3.0f)); }
                else CHECK(got == 0.0f, "octahedron outside voxel (%d,%d,%d) = %g", k, j, i, got);
            }
    CHECK(inside > 1000, "octahedron: %d inside voxels", inside);
    free(phi);
    printf(fails ? "sdf_kat: %d FAILED\n" : "sdf_kat: ok (%d)\n", fails ? fails : inside);
    return fails ? 1 : 0;
}
