// CPU check of the static tables behind k_view's laser_map (img_env_amd/csrc/host_tables.h build_robot_class), without a GPU:
//   1. ray_run: the cells a hit "leaves alone" (agent.cpp:555-560: same row or column as the hit cell) are ONE run of steps right
//      behind the hit -- the property the kernel's packed hit word (first-hit step << 16 | last left-alone step) rests on;
//   2. top_ent / inv_pack / inv_ent / inv_cell agree with each other, and the block of the reach table a cell points at covers
//      all of its lower beams;
//   3. the table-driven laser_map (top beam's verdict, else the first deciding entry of the cell's ray list, behind the reach
//      filter) equals the reference's sequential algorithm (beams in order, later beams overwrite earlier ones) on random
//      occupancies -- i.e. kernel logic restated on the host against the oracle's logic restated on the host.
// usage: k_view_tables_check <view_w> <view_h> <res> <beams> <angle_begin> <angle_end> <seed> ; exit code 0 = all good
#include <stdio.h>

#include <random>

#define WAVE_SZ 64
#include "../../img_env_amd/csrc/host_tables.h"

static int fail(const char* what, long a = 0, long b = 0, long c = 0) {
    printf("FAIL %s (%ld %ld %ld)\n", what, a, b, c);
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 8) return fail("usage");
    imgenv_cfg c;
    memset(&c, 0, sizeof(c));
    c.view_width = (float)atof(argv[1]);
    c.view_height = (float)atof(argv[2]);
    c.view_resolution = (float)atof(argv[3]);
    c.use_laser = 1;
    c.range_total = atoi(argv[4]);
    c.view_angle_begin = (float)atof(argv[5]);
    c.view_angle_end = (float)atof(argv[6]);
    c.view_min_dist = -100.f;
    c.view_max_dist = 100.f;
    const unsigned seed = (unsigned)atoi(argv[7]);
    const ViewGeom g = make_view_geom(c);
    RobotClassHost k;
    k.shape = IMGENV_SHAPE_CIRCLE;
    k.size[0] = 0.f; k.size[1] = 0.f; k.size[2] = 0.17f; k.size[3] = 0.f;
    k.sensor[0] = 0.f; k.sensor[1] = 0.f;
    build_robot_class(k, g);
    if (!k.ok) return fail("class tables overflow");
    if (k.big) { printf("SKIP big class\n"); return 0; }
    const int Hv = g.Hv, Wv = g.Wv, NC = Hv * Wv, B = g.B, S = k.ray_stride;
    // the paths back out of the chunk-major table
    std::vector<std::vector<int>> path(B);
    for (int b = 0; b < B; b++)
        for (int q = 0; q < k.ray_len[b]; q++) path[b].push_back(k.ray_rows[((q / 8) * (size_t)S + b) * 8 + (q % 8)]);
    // 1. ray_run
    for (int b = 0; b < B; b++)
        for (int q = 0; q < (int)path[b].size(); q++) {
            const int x = path[b][q] / Wv, y = path[b][q] % Wv;
            int run = 0;
            bool in_run = true;
            for (int t = q + 1; t < (int)path[b].size(); t++) {
                const bool same = path[b][t] / Wv == x || path[b][t] % Wv == y;
                if (same && !in_run) return fail("left-alone cells are not one run", b, q, t);
                if (same) run++; else in_run = false;
            }
            if (k.ray_run[(size_t)q * S + b] != run) return fail("ray_run", b, q, run);
        }
    // 2. lists
    const uint32_t nb8 = ((uint32_t)B >> 3) + 1, lvl_n[3] = {nb8, (nb8 + 1) >> 1, (nb8 + 3) >> 2}, lvl_off[3] = {0, lvl_n[0], lvl_n[0] + lvl_n[1]};
    for (int cidx = 0; cidx < NC; cidx++) {
        const uint32_t pk = k.inv_pack[cidx], e0 = pk & 0xFFFFFu, cnt = pk >> 20;
        if (k.inv_cell[2 * (size_t)cidx + 1] != pk) return fail("inv_cell.y != inv_pack", cidx);
        if (cnt == 0) {
            if (k.top_ent[cidx] != (((uint32_t)B << 16) | 0xFFFFu)) return fail("top_ent of a cell without beams", cidx);
            continue;
        }
        if (k.top_ent[cidx] != k.inv_ent[e0]) return fail("top_ent != head of the list", cidx);
        for (uint32_t e = 0; e < cnt; e++) {
            const uint32_t ent = k.inv_ent[e0 + e], b = ent >> 16, q = ent & 0xFFFFu;
            if ((int)b >= B || q >= path[b].size() || path[b][q] != cidx) return fail("list entry does not go through its cell", cidx, e);
            if (e > 0 && b >= (k.inv_ent[e0 + e - 1] >> 16)) return fail("list not beam-descending", cidx, e);
        }
        const uint32_t f = k.inv_cell[2 * (size_t)cidx];
        if (cnt >= 2 && ((f >> 13) & 1u) == 0) {
            const uint32_t idx = f & 0x1FFFu;
            int v = idx >= lvl_off[2] ? 2 : (idx >= lvl_off[1] ? 1 : 0);
            const uint32_t i = idx - lvl_off[v], lo = i * (8u << v), hi = (i + 2) * (8u << v);
            uint32_t kkmin = 0xFFFF;
            for (uint32_t e = 1; e < cnt; e++) {
                const uint32_t ent = k.inv_ent[e0 + e], b = ent >> 16;
                if (b < lo || b >= hi) return fail("reach block does not cover a lower beam", cidx, e, idx);
                kkmin = std::min(kkmin, ent & 0xFFFFu);
            }
            if ((f >> 24) != std::min(kkmin, 0xFFu)) return fail("smallest step", cidx);
        }
    }
    // 3. random occupancies: sequential reference against the table-driven composition
    std::mt19937 rng(seed);
    for (int trial = 0; trial < 40; trial++) {
        const double density = trial < 4 ? 0.0 : (trial % 5) * 0.02 + 0.004;
        std::vector<uint8_t> occ(NC, 0);
        for (int q = 0; q < NC; q++) occ[q] = (rng() % 100000) < density * 100000;
        if (trial % 3 == 0)  // an axis-parallel wall: the case with long left-alone runs
            for (int y = 0; y < Wv; y++) occ[(rng() % Hv) * Wv + y] = 1;
        // hits
        std::vector<int> hk(B, -1);
        for (int b = 0; b < B; b++)
            for (int q = 0; q < (int)path[b].size(); q++)
                if (occ[path[b][q]]) { hk[b] = q; break; }
        // reference: beams in order (agent.cpp:419-437, 511-624)
        std::vector<int> ref(NC, 200);
        for (int b = 0; b < B; b++) {
            for (int q = 0; q < (int)path[b].size(); q++) {
                const int cc = path[b][q];
                if (hk[b] < 0 || q < hk[b]) ref[cc] = 255;
                else if (q == hk[b]) ref[cc] = 0;
                else {
                    const int hx = path[b][hk[b]] / Wv, hy = path[b][hk[b]] % Wv;
                    if (cc / Wv == hx || cc % Wv == hy) continue;
                    ref[cc] = 200;
                }
            }
        }
        // tables: hit words, reach table, top beam, filter, list walk
        std::vector<uint32_t> hit(B + 1);
        for (int b = 0; b < B; b++) hit[b] = hk[b] < 0 ? 0xFFFFFFFFu : (((uint32_t)hk[b] << 16) | (uint32_t)(hk[b] + k.ray_run[(size_t)hk[b] * S + b]));
        hit[B] = 0;
        std::vector<uint32_t> reach(lvl_off[2] + lvl_n[2]);
        for (uint32_t i = 0; i < lvl_n[0]; i++) {
            uint32_t m = 0;
            for (int q = 0; q < 16; q++) m = std::max(m, hit[std::min<uint32_t>(8 * i + q, B)]);
            reach[i] = m;
        }
        for (uint32_t i = 0; i < lvl_n[1]; i++) reach[lvl_off[1] + i] = std::max(reach[2 * i], reach[std::min(2 * i + 2, lvl_n[0] - 1)]);
        for (uint32_t i = 0; i < lvl_n[2]; i++) reach[lvl_off[2] + i] = std::max(reach[lvl_off[1] + 2 * i], reach[lvl_off[1] + std::min(2 * i + 2, lvl_n[1] - 1)]);
        for (int cidx = 0; cidx < NC; cidx++) {
            const uint32_t top = k.top_ent[cidx], kk = top & 0xFFFFu, hp = hit[top >> 16], h = hp >> 16;
            int v = kk < h ? 255 : (kk == h ? 0 : 200);
            if (kk > h && kk <= (hp & 0xFFFFu)) {  // left alone by its top beam
                const uint32_t f = k.inv_cell[2 * (size_t)cidx], pk = k.inv_cell[2 * (size_t)cidx + 1], e0 = pk & 0xFFFFFu, cnt = pk >> 20;
                const bool pass = cnt >= 2 && (((f >> 13) & 1u) != 0 || (reach[f & 0x1FFFu] >> 16) >= (f >> 24));
                int walked = 200;
                for (uint32_t e = 1; e < cnt; e++) {
                    const uint32_t ent = k.inv_ent[e0 + e], k2 = ent & 0xFFFFu, hp2 = hit[ent >> 16], h2 = hp2 >> 16;
                    if (k2 < h2) { walked = 255; break; }
                    if (k2 == h2) { walked = 0; break; }
                    if (k2 > (hp2 & 0xFFFFu)) break;
                }
                if (!pass && walked != 200) return fail("the reach filter dropped a cell that changes", trial, cidx, walked);
                v = pass ? walked : 200;
            }
            if (v != ref[cidx]) return fail("laser_map", trial, cidx, v * 1000 + ref[cidx]);
        }
    }
    // 6. the static lists of a STEP's crop / final pass (round 4): every cell on a beam's path lies in a listed group, no beam
    //    crosses a cell of a group that is not listed, the groups are listed once and in ascending order, a group's word carries
    //    its field-of-view and own-footprint bits, and all_groups holds every group
    {
        std::vector<char> listed((NC + 3) / 4, 0);
        uint32_t prev = 0;
        for (size_t q = 0; q < k.dyn_groups.size(); q++) {
            const uint32_t wd = k.dyn_groups[q], c4 = wd & 0xFFFFu;
            if (c4 % 4 != 0 || (int)c4 >= NC) return fail("dyn_groups: first cell", (long)q, c4);
            if (q > 0 && c4 <= prev) return fail("dyn_groups: order", (long)q, c4, prev);
            prev = c4;
            listed[c4 / 4] = 1;
        }
        for (int b = 0; b < B; b++)
            for (int c : path[b])
                if (!listed[c / 4]) return fail("a beam crosses a cell outside the listed groups", b, c);
        std::vector<char> crossed(NC, 0);
        for (int b = 0; b < B; b++)
            for (int c : path[b]) crossed[c] = 1;
        if (B > 0)
            for (int gq = 0; gq < (NC + 3) / 4; gq++) {
                bool any = false;
                for (int c = 4 * gq; c < std::min(4 * gq + 4, NC); c++) any = any || crossed[c];
                if (any != (listed[gq] != 0) && !(k.dyn_groups.size() == 1 && !any && gq == 0)) return fail("dyn_groups: listed without a beam", gq);
            }
        if ((int)k.all_groups.size() != (NC + 3) / 4) return fail("all_groups: size", (long)k.all_groups.size());
        auto check_word = [&](uint32_t wd) {
            const int c4 = (int)(wd & 0xFFFFu);
            for (int q = 0; q < 4 && c4 + q < NC; q++) {
                const int c = c4 + q;
                if (((wd >> (16 + q)) & 1u) != ((k.fov_bits[c >> 5] >> (c & 31)) & 1u)) return false;
                if (((wd >> (20 + q)) & 1u) != ((k.stamp_bits[c >> 5] >> (c & 31)) & 1u)) return false;
            }
            return true;
        };
        for (size_t q = 0; q < k.all_groups.size(); q++)
            if ((k.all_groups[q] & 0xFFFFu) != 4u * (uint32_t)q || !check_word(k.all_groups[q])) return fail("all_groups: word", (long)q);
        for (uint32_t wd : k.dyn_groups)
            if (!check_word(wd)) return fail("dyn_groups: word", wd & 0xFFFFu);
    }
    printf("OK %d x %d cells, %d beams, %zu of %d cell groups in a step\n", Hv, Wv, B, k.dyn_groups.size(), (NC + 3) / 4);
    return 0;
}
