// Host driver for the pure arithmetic helpers of libihmr_hip (ihmr_amd/csrc/ihmr_pure.h): the SAME header the kernels inline, compiled
// by g++ with -fsanitize=address,undefined, so that the GPU-less container can run them and tests/test_pure_host_cpu.py can compare
// them bit for bit with the CPU oracle (oracle/sdf_grid.c) / numpy / torch.  Test infrastructure, not product.
//
//   pure_host_driver <op> <in.bin> <out.bin>      in.bin: int32 n, then n records of float32; out.bin: float32 / uint32 results
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../ihmr_amd/csrc/ihmr_pure.h"

static std::vector<float> read_in(const char* path, int* n, int rec) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    int32_t nn = 0;
    if (fread(&nn, 4, 1, f) != 1) exit(2);
    std::vector<float> v((size_t)nn * rec);
    if (!v.empty() && fread(v.data(), 4, v.size(), f) != v.size()) { fprintf(stderr, "short read\n"); exit(2); }
    fclose(f);
    *n = nn;
    return v;
}
static void write_out(const char* path, const void* p, size_t bytes) {
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}
static uint32_t bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s <op> <in> <out>\n", argv[0]); return 2; }
    const char* op = argv[1];
    int n = 0;
    if (!strcmp(op, "ptd")) {                       // (a, b, c, p) -> squared distance
        auto in = read_in(argv[2], &n, 12);
        std::vector<float> out(n);
        for (int i = 0; i < n; ++i) { const float* r = &in[(size_t)i * 12]; out[i] = sdf_point_tri_dist2(r, r + 3, r + 6, r[9], r[10], r[11]); }
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "raycol")) {             // (a, b, c, col) -> det bits, passed the yz-degeneracy + (u, v) tests, hit mask of all 32 voxels
        auto in = read_in(argv[2], &n, 10);
        std::vector<uint32_t> out((size_t)n * 3);
        for (int i = 0; i < n; ++i) {
            const float* r = &in[(size_t)i * 10];
            const float det = sdf_tri_yz_det(r[1], r[2], r[4], r[5], r[7], r[8]);
            bool uv = false;
            unsigned hits = 0;
            if (fabsf(det) >= 1e-12f) hits = sdf_ray_column_hits(r, r + 3, r + 6, (int)r[9], 0xffffffffu, uv);    // (the kernel's enumeration drops the others)
            out[3 * i] = bits(det); out[3 * i + 1] = uv ? 1u : 0u; out[3 * i + 2] = hits;
        }
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "raycol_need")) {        // (a, b, c, col, need-as-bits) -> hit mask restricted to `need` (must equal all-voxel mask & need)
        auto in = read_in(argv[2], &n, 11);
        std::vector<uint32_t> out(n);
        for (int i = 0; i < n; ++i) {
            const float* r = &in[(size_t)i * 11];
            const float det = sdf_tri_yz_det(r[1], r[2], r[4], r[5], r[7], r[8]);
            uint32_t need; memcpy(&need, &r[10], 4);
            bool uv = false;
            out[i] = (fabsf(det) >= 1e-12f && need) ? sdf_ray_column_hits(r, r + 3, r + 6, (int)r[9], need, uv) : 0u;
        }
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "div")) {                // (a, b) -> a / b through the three-instruction form
        auto in = read_in(argv[2], &n, 2);
        std::vector<float> out(n);
        for (int i = 0; i < n; ++i) out[i] = sdf_div(in[2 * i], sdf_divisor(in[2 * i + 1]));
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "colrange")) {           // (y0, y1, y2, z0, z1, z2) -> j0, j1, k0, k1
        auto in = read_in(argv[2], &n, 6);
        std::vector<int32_t> out((size_t)n * 4);
        for (int i = 0; i < n; ++i) {
            const float* r = &in[(size_t)i * 6];
            int j0, j1, k0, k1;
            tri_col_range(r[0], r[1], r[2], r[3], r[4], r[5], j0, j1, k0, k1);
            out[4 * i] = j0; out[4 * i + 1] = j1; out[4 * i + 2] = k0; out[4 * i + 3] = k1;
        }
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "unnorm")) {             // (x, align_corners) -> grid coordinate; + the voxel centres of ids 0 .. n-1
        auto in = read_in(argv[2], &n, 2);
        std::vector<float> out((size_t)n * 4);
        for (int i = 0; i < n; ++i) {
            out[4 * i] = sdf_unnorm(in[2 * i], (int)in[2 * i + 1]);
            sdf_vox_centre(i % (SDF_G * SDF_G * SDF_G), out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
        }
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "rod_fwd")) {
        auto in = read_in(argv[2], &n, 3);
        std::vector<float> out((size_t)n * 9);
        for (int i = 0; i < n; ++i) rodrigues_fwd(&in[3 * i], &out[9 * i]);
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "rod_bwd")) {            // (r, dR) -> dr
        auto in = read_in(argv[2], &n, 12);
        std::vector<float> out((size_t)n * 3);
        for (int i = 0; i < n; ++i) rodrigues_bwd(&in[12 * i], &in[12 * i + 3], &out[3 * i]);
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "adam")) {               // (x, g, m, v, step_size, bc2_sqrt) -> x', m', v'
        auto in = read_in(argv[2], &n, 6);
        std::vector<float> out((size_t)n * 3);
        for (int i = 0; i < n; ++i) {
            const float* r = &in[(size_t)i * 6];
            float m = r[2], v = r[3];
            out[3 * i] = opt_adam_update(r[0], r[1], m, v, r[4], r[5]);
            out[3 * i + 1] = m; out[3 * i + 2] = v;
        }
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "sgd")) {                // (x, g, m, lr) -> x', m'
        auto in = read_in(argv[2], &n, 4);
        std::vector<float> out((size_t)n * 2);
        for (int i = 0; i < n; ++i) {
            const float* r = &in[(size_t)i * 4];
            float m = r[2];
            out[2 * i] = opt_sgd_update(r[0], r[1], m, r[3]);
            out[2 * i + 1] = m;
        }
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "chain")) {              // (G_parent 12, R 9, J_joint 3, J_parent 3) -> G 12, A 12
        auto in = read_in(argv[2], &n, 27);
        std::vector<float> out((size_t)n * 24);
        for (int i = 0; i < n; ++i) {
            const float* r = &in[(size_t)i * 27];
            float* G = &out[(size_t)i * 24];
            for (int e = 0; e < 12; ++e) G[e] = lbs_chain_elem(r, r + 12, r + 21, r + 24, e / 4, e % 4);
            for (int e = 0; e < 12; ++e) G[12 + e] = lbs_rel_elem(G, r + 21, e / 4, e % 4);
        }
        write_out(argv[3], out.data(), out.size() * 4);
    } else if (!strcmp(op, "misc")) {               // cross3 + align_root: (a, b, w) -> a x b, root
        auto in = read_in(argv[2], &n, 7);
        std::vector<float> out((size_t)n * 4);
        for (int i = 0; i < n; ++i) {
            const float* r = &in[(size_t)i * 7];
            cross3(r, r + 3, &out[4 * i]);
            out[4 * i + 3] = (float)align_root(r[6]);
        }
        write_out(argv[3], out.data(), out.size() * 4);
    } else {
        fprintf(stderr, "unknown op %s\n", op);
        return 2;
    }
    return 0;
}
