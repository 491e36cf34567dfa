// host_tables.h -- pose-independent tables of a robot / pedestrian class, computed once on the
// host at imgenv_create() with the host libm (the same glibc the reference node links), and the
// host part of reset (obstacle raster, RVO obstacle BSP).
//
// Why tables: in Agent::view (agent.cpp:356-509) the field-of-view gate of the crop, the
// Bresenham path of every beam and the cells of the robot's own footprint depend only on the
// view geometry, the sensor offset and the footprint -- not on the pose.  The reference recomputes
// them per robot per step (one atan2 per view cell, one sin/cos + line walk per beam); here they are
// evaluated once with the reference's formulas and the kernels replay them.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/imgenv.h"
#include "fp_rows.h"
#include "tfm.h"

struct Pts {
    std::vector<double> x, y;
    int n() const { return (int)x.size(); }
};

// Agent::init_shape_circle (agent.cpp:18-30)
static Pts shape_circle(double cx, double cy, double r) {
    Pts p;
    const double resolution = 0.01;
    const int bb = (int)ceil(r / resolution);
    for (int m = -bb; m <= bb; m++)
        for (int n = -bb; n <= bb; n++)
            if (sqrt(m * resolution * m * resolution + n * resolution * n * resolution) <= r) {
                p.x.push_back(m * resolution + cx);
                p.y.push_back(n * resolution + cy);
            }
    return p;
}

// Agent::init_shape_rectangle (agent.cpp:51-62)
static Pts shape_rectangle(const double* s) {
    Pts p;
    const double resolution = 0.01;
    const int x_min = (int)floor(s[0] / resolution), x_max = (int)ceil(s[1] / resolution);
    const int y_min = (int)floor(s[2] / resolution), y_max = (int)ceil(s[3] / resolution);
    for (int m = x_min; m <= x_max; m++)
        for (int n = y_min; n <= y_max; n++) {
            p.x.push_back(m * resolution);
            p.y.push_back(n * resolution);
        }
    return p;
}

// The lattice rows of a sample list (fp_rows.h): runs of consecutive samples (px, cy + 0.01 n), n = n_lo .. n_hi.  Empty when
// the list is not of that form -- it always is for the reference's discs and rectangles; this re-derives the rows from the list
// itself and checks every sample, so a mismatch can only switch the shortcut off -- or when a row can take more steps than
// fp_rows.h handles at this resolution (FPR_MAXC: then nearly every robot would fall back to the sample walk anyway).
static std::vector<FpRow> build_fp_rows(const Pts& p, double cy, double res) {
    std::vector<FpRow> rows;
    const double pitch = FPR_PITCH;
    int q = 0;
    double longest = 0;
    while (q < p.n()) {
        int e = q;
        while (e + 1 < p.n() && p.x[e + 1] == p.x[q]) e++;
        FpRow r;
        r.px = p.x[q];
        r.n_lo = (int)lrint((p.y[q] - cy) / pitch);
        r.n_hi = r.n_lo + (e - q);
        r.q0 = q;
        for (int t = q; t <= e; t++)
            if (p.y[t] != (r.n_lo + (t - q)) * pitch + cy) return {};  // (the generating expression, agent.cpp:26, 58)
        if (!rows.empty() && !(r.px > rows.back().px)) return {};
        longest = std::max(longest, (r.n_hi - r.n_lo) * pitch);
        rows.push_back(r);
        q = e + 1;
    }
    if ((int)floor(longest / res) + 1 > FPR_MAXC) return {};
    return rows;
}

struct ViewGeom {
    double res, view_w, view_h, a_begin, a_end, min_d, max_d;
    int Hv, Wv, B, use_laser, range_total;
    Tf2 view_base, base_view;
};

// Agent::init_view_map (agent.cpp:79-90)
static ViewGeom make_view_geom(const imgenv_cfg& c) {
    ViewGeom g;
    g.res = (double)c.view_resolution;
    g.view_w = (double)c.view_width;
    g.view_h = (double)c.view_height;
    g.a_begin = (double)c.view_angle_begin;
    g.a_end = (double)c.view_angle_end;
    g.min_d = (double)c.view_min_dist;
    g.max_d = (double)c.view_max_dist;
    g.Wv = (int)(g.view_w / g.res);
    g.Hv = (int)(g.view_h / g.res);
    g.use_laser = c.use_laser;
    g.range_total = c.range_total;
    g.B = c.use_laser ? c.range_total : 0;
    g.view_base.ox = g.view_h / 2;
    g.view_base.oy = g.view_w / 2;
    const double half = 3.14159 * 0.5;
    tf_set_rotation_zw(g.view_base, sin(half), cos(half));
    g.base_view = tf_inverse(g.view_base);
    return g;
}

// a bit of the crop bitmap as k_beams_big wants it: the byte address of its 32-bit word in bits 5.., the bit in bits 0-4 -- one
// shift gives the LDS address, v_bfe_u32 takes the bit number from the entry's low bits as it is
static inline uint32_t big_bit_entry(uint32_t bit_address) { return ((bit_address >> 5) << 7) | (bit_address & 31u); }

struct RobotClassHost {
    int shape;
    float size[4], sensor[2];
    double sizes[4], sx, sy;
    Pts fp;
    std::vector<FpRow> fp_rows;  // lattice rows of fp (fp_rows.h), empty: the rasters walk the samples
    double fp_cy = 0;
    std::vector<uint32_t> fov_bits, stamp_bits;
    int ray_maxlen = 0, ray_stride = 0, ray_kpad = 8, box_rad = 0;
    bool ok = true;  // false: a table field overflowed its packing
    bool big = false;  // the view is beyond what k_view packs into 16 / 8 bits: k_view_big and its 32-bit path table instead
    int sensor_x = 0, sensor_y = 0;  // view cell of the laser
    // big views (csrc/view_big.h).  The cropped view lives as a bitmap in 8 x 8 tiles: bit address of view cell (a, b) =
    // ((a / 8) * big_tb + b / 8) * 64 + (a % 8) * 8 + b % 8, so that one wavefront crops one tile and stores one 64-bit ballot
    int big_ta = 0, big_tb = 0;       // tiles per column / per row
    std::vector<uint32_t> big_cells;  // [ray_kpad / 4][ray_stride][4] bitmap bit of step k of beam b as (byte address of its 32-bit word) << 5 | bit (big_bit_entry); past the ray's end: the always-free bit behind the bitmap
    std::vector<uint16_t> ray_end;    // [ray_maxlen][ray_stride] last step behind step k of beam b that shares its row or column (k itself if none)
    std::vector<uint32_t> big_inv;    // [NC][2] rays through a view cell: first entry of inv_ent, count
    std::vector<uint32_t> crop_tiles; // tiles with at least one cell inside the field of view: ta << 16 | tb ...
    std::vector<uint64_t> crop_masks; // ... and their cells inside it (bit = (a % 8) * 8 + b % 8), + 8 zero entries
    int n_crop = 0;
    std::vector<uint32_t> tap_top, tap_inv, tap_addr;  // [16][image_h * image_w] the 4 x 4 source cells of every pixel of the shrunk sensor_map (build_big_taps)
    // k_taps_big in a STEP: the chunks of TAP_CHUNK_PIXELS pixels in which some pixel has a source cell that a beam crosses -- a pixel
    // whose 16 source cells see no beam is a mix of 200 and the own footprint's 100 for the whole episode (the launch of a reset
    // covers every chunk, the launches of the steps only these)
    std::vector<uint16_t> tap_chunks;
    std::vector<uint16_t> ray_rows, ray_len;
    std::vector<float> ray_dist;
    std::vector<uint8_t> ray_run;  // [ray_maxlen][ray_stride] steps behind step k of beam b that share a row or column with it
    // AgentState.hits_x / hits_y / angular_map (agent.cpp:405-438), only with IMGENV_FLAG_AGENT_STATE_EXTRAS
    std::vector<float> ray_hx, ray_hy;  // [ray_maxlen + 1][ray_stride] float32(hit * cos / sin(beam angle)) for a hit at step k; last row: no hit
    std::vector<uint16_t> bin_start;    // [73] first beam of each of the 72 angular_map bins (a beam's bin never decreases)
    std::vector<uint32_t> inv_pack, inv_ent, top_ent;
    std::vector<uint32_t> inv_cell;  // [NC][2] k_view's step (5): filter word, inv_pack
    // k_view's final pass in a STEP: the groups of 4 view cells (first cell c4, c4 % 4 == 0) in which a step can change anything -- a
    // cell no beam crosses (no laser: a cell outside the field of view) is 200, or 100 under the own footprint, for the whole
    // episode: the pass of a reset writes every cell, the passes of the steps only these groups
    std::vector<uint16_t> dyn_c4;
    // ... as k_view reads them: one word per group = c4 | the group's four field-of-view bits << 16 | its four own-footprint bits
    // << 20 (what the kernel would otherwise fetch from fov_bits / stamp_bits once it knows c4: two dependent round trips a round);
    // all_groups: the same for every group of the view (the pass of a reset)
    std::vector<uint32_t> dyn_groups, all_groups;
};

static void build_robot_class(RobotClassHost& k, const ViewGeom& g, bool force_big = false, bool extras = false) {
    for (int j = 0; j < 4; j++) k.sizes[j] = (double)k.size[j];
    k.sx = (double)k.sensor[0];
    k.sy = (double)k.sensor[1];
    k.fp = k.shape == IMGENV_SHAPE_CIRCLE ? shape_circle(k.sizes[0], k.sizes[1], k.sizes[2]) : shape_rectangle(k.sizes);
    k.fp_cy = k.shape == IMGENV_SHAPE_CIRCLE ? k.sizes[1] : 0.0;
    k.fp_rows = build_fp_rows(k.fp, k.fp_cy, g.res);
    const int Hv = g.Hv, Wv = g.Wv, NC = Hv * Wv, NW = (NC + 31) / 32;
    const double res = g.res;
    // crop gate (agent.cpp:373-386)
    k.fov_bits.assign(NW, 0);
    for (int a = 0; a < Hv; a++)
        for (int b = 0; b < Wv; b++) {
            double xb, yb;
            tf_apply(g.view_base, a * res, b * res, xb, yb);
            const double ang = atan2(yb - k.sy, xb - k.sx);
            if (ang <= g.a_begin || ang >= g.a_end || xb < g.min_d || xb > g.max_d) continue;
            const int c = a * Wv + b;
            k.fov_bits[c >> 5] |= 1u << (c & 31);
        }
    // own footprint in the view (agent.cpp:503, 307-312)
    k.stamp_bits.assign(NW, 0);
    double ext = 0;
    for (int q = 0; q < k.fp.n(); q++) {
        double vx, vy;
        tf_apply(g.base_view, k.fp.x[q], k.fp.y[q], vx, vy);
        const int m = w2m(vx, res), n = w2m(vy, res);
        if (m >= 0 && m < Hv && n >= 0 && n < Wv) {
            const int c = m * Wv + n;
            k.stamp_bits[c >> 5] |= 1u << (c & 31);
        }
        ext = std::max(ext, sqrt(k.fp.x[q] * k.fp.x[q] + k.fp.y[q] * k.fp.y[q]));
    }
    k.box_rad = (int)ceil(ext / res) + 2;
    // laser ray paths (agent.cpp:366-369, 405-438, 511-624)
    const int B = g.B;
    k.ray_stride = ((B + WAVE_SZ - 1) / WAVE_SZ) * WAVE_SZ;
    if (k.ray_stride == 0) k.ray_stride = WAVE_SZ;
    k.ray_len.assign(k.ray_stride, 0);
    std::vector<std::vector<uint32_t>> cells(B);
    std::vector<std::vector<float>> dists(B);
    std::vector<std::vector<float>> hxs(B), hys(B);
    std::vector<float> nohit_x(B, 0.f), nohit_y(B, 0.f);
    std::vector<int> beam_bin(B, 0);
    if (B > 0) {
        double sxv, syv;
        tf_apply(g.base_view, k.sx, k.sy, sxv, syv);
        const int x1 = w2m(sxv, res), y1 = w2m(syv, res);
        k.sensor_x = x1;
        k.sensor_y = y1;
        const double x0w = x1 * res, y0w = y1 * res;
        const double map_width = g.base_view.ox, map_height = g.base_view.oy;
        const double max_range = sqrt(map_width * map_width + map_height * map_height);
        const double angle_step = fabs(g.a_end - g.a_begin) / g.range_total;
        for (int b = 0; b < B; b++) {
            const double cur = g.a_begin + angle_step * b;
            const double x = max_range * cos(cur), y = max_range * sin(cur);
            {   // hit_points_x_ / _y_ = hit * cos / sin(cur_angle); angular_map_i = int(angle_step * i / angular_map_step) (agent.cpp:417-435)
                const double angular_map_step = fabs(g.a_end - g.a_begin) / IMGENV_ANGULAR_BINS;
                beam_bin[b] = std::min(IMGENV_ANGULAR_BINS - 1, std::max(0, (int)(angle_step * b / angular_map_step)));
                nohit_x[b] = (float)(6.0 * cos(cur));
                nohit_y[b] = (float)(6.0 * sin(cur));
            }
            double vx, vy;
            tf_apply(g.base_view, x, y, vx, vy);
            const int x2 = w2m(vx, res), y2 = w2m(vy, res);
            int wv = x2 - x1, hv = y2 - y1;
            const int dx = ((wv > 0) << 1) - 1, dy = ((hv > 0) << 1) - 1;
            wv = abs(wv);
            hv = abs(hv);
            const bool steep = !(wv > hv);
            int f, d1, d2;
            if (!steep) {
                f = 2 * hv - wv; d1 = 2 * hv; d2 = (hv - wv) * 2;
            } else {
                f = 2 * wv - hv; d1 = wv * 2; d2 = (wv - hv) * 2;
            }
            int xx = x1, yy = y1;
            while (steep ? (yy != y2) : (xx != x2)) {
                if (!(xx >= 0 && xx < Hv && yy >= 0 && yy < Wv)) break;  // "else return hit"
                cells[b].push_back((uint32_t)(xx * Wv + yy));
                const double cx = xx * res, cy = yy * res;
                const double hit = sqrt((x0w - cx) * (x0w - cx) + (y0w - cy) * (y0w - cy));
                dists[b].push_back((float)hit);
                if (extras) {
                    hxs[b].push_back((float)(hit * cos(cur)));
                    hys[b].push_back((float)(hit * sin(cur)));
                }
                if (f < 0) {
                    f += d1;
                } else {
                    if (steep) xx += dx; else yy += dy;
                    f += d2;
                }
                if (steep) yy += dy; else xx += dx;
            }
            k.ray_len[b] = (uint16_t)cells[b].size();
            k.ray_maxlen = std::max(k.ray_maxlen, (int)cells[b].size());
        }
    }
    if (k.ray_maxlen == 0) k.ray_maxlen = 1;
    if (extras && B > 0) {
        k.ray_hx.assign((size_t)(k.ray_maxlen + 1) * k.ray_stride, 0.f);
        k.ray_hy.assign((size_t)(k.ray_maxlen + 1) * k.ray_stride, 0.f);
        for (int b = 0; b < B; b++) {
            for (size_t q = 0; q < hxs[b].size(); q++) {
                k.ray_hx[q * k.ray_stride + b] = hxs[b][q];
                k.ray_hy[q * k.ray_stride + b] = hys[b][q];
            }
            k.ray_hx[(size_t)k.ray_maxlen * k.ray_stride + b] = nohit_x[b];
            k.ray_hy[(size_t)k.ray_maxlen * k.ray_stride + b] = nohit_y[b];
        }
        k.bin_start.assign(IMGENV_ANGULAR_BINS + 1, (uint16_t)B);
        for (int b = B - 1; b >= 0; b--) k.bin_start[beam_bin[b]] = (uint16_t)b;
        for (int m = IMGENV_ANGULAR_BINS - 1; m >= 0; m--)
            if (k.bin_start[m] == (uint16_t)B || k.bin_start[m] > k.bin_start[m + 1]) k.bin_start[m] = k.bin_start[m + 1];  // an empty bin
        for (int b = 1; b < B; b++)
            if (beam_bin[b] < beam_bin[b - 1]) k.ok = false;
    }
    k.ray_kpad = ((k.ray_maxlen + 7) / 8) * 8;
    if (force_big || NC + 16 > 0xFFFF || k.ray_maxlen > 255 || Hv > 256 || Wv > 256) k.ray_kpad = ((k.ray_maxlen + 31) / 32) * 32;  // k_beams_big: 32 steps a round
    // k_view packs a view cell into 16 bits and (step, row, col) of a hit into 8 bits each; beyond that the class is "big"
    k.big = force_big || NC + 16 > 0xFFFF || k.ray_maxlen > 255 || Hv > 256 || Wv > 256;
    if (k.ray_maxlen > 0xFFFF || B > 0xFFFF) k.ok = false;  // (beam << 16 | step) entries of the per-cell ray lists
    if (k.big && B > 0x7FFF) k.ok = false;                   // k_taps_big keeps a flag in bit 31 of a cell's top entry
    std::vector<std::vector<uint32_t>> inv(NC);
    k.big_ta = (Hv + 7) / 8;
    k.big_tb = (Wv + 7) / 8;
    if (k.big) {
        const uint32_t free_bit = (uint32_t)k.big_ta * (uint32_t)k.big_tb * 64u;  // the word behind the bitmap stays zero
        k.big_cells.assign((size_t)k.ray_kpad * k.ray_stride, big_bit_entry(free_bit));  // k_beams_big walks 32 steps at a time
        k.ray_rows.assign(8, 0);
        k.ray_dist.assign((size_t)k.ray_maxlen * k.ray_stride, 6.0f);
        k.ray_end.assign((size_t)k.ray_maxlen * k.ray_stride, 0);
        k.ray_run.assign(1, 0);
        // the cells a hit leaves alone (agent.cpp:555-560) are one run of steps right behind it (see below): its last step
        for (int b = 0; b < B; b++) {
            const size_t n = cells[b].size();
            for (size_t q = 0; q < n; q++) {
                const uint32_t xq = cells[b][q] / (uint32_t)Wv, yq = cells[b][q] % (uint32_t)Wv;
                // (a path moves along its major axis every step and never back along the minor one: once off the hit's row and
                // column it stays off; tests/test_k_view_tables.py checks that on whole paths)
                size_t run = 0;
                for (size_t t = q + 1; t < n; t++) {
                    if (!(cells[b][t] / (uint32_t)Wv == xq || cells[b][t] % (uint32_t)Wv == yq)) break;
                    run++;
                }
                k.ray_end[q * k.ray_stride + b] = (uint16_t)(q + run);
            }
        }
        for (int ta = 0; ta < k.big_ta; ta++)
            for (int tb = 0; tb < k.big_tb; tb++) {
                uint64_t mask = 0;
                for (int q = 0; q < 64; q++) {
                    const int a = ta * 8 + (q >> 3), b = tb * 8 + (q & 7);
                    if (a >= Hv || b >= Wv) continue;
                    const int c = a * Wv + b;
                    if ((k.fov_bits[c >> 5] >> (c & 31)) & 1u) mask |= 1ull << q;
                }
                if (mask) {
                    k.crop_tiles.push_back(((uint32_t)ta << 16) | (uint32_t)tb);
                    k.crop_masks.push_back(mask);
                }
            }
        if (k.crop_tiles.empty()) {
            k.crop_tiles.push_back(0);
            k.crop_masks.push_back(0);
        }
        k.n_crop = (int)k.crop_tiles.size();
        k.crop_masks.insert(k.crop_masks.end(), 8, 0);  // k_crop_big loads the masks of a whole round (<= 8 tiles) at once
    } else {
        // chunk-major: the 8 steps 8c..8c+7 of beam b sit at ((c * ray_stride) + b) * 8; padding = a free dummy cell behind the view
        k.ray_rows.assign((size_t)(k.ray_kpad / 8) * k.ray_stride * 8, (uint16_t)NC);
        k.ray_dist.assign((size_t)k.ray_maxlen * k.ray_stride, 6.0f);
        k.big_cells.assign(1, 0xFFFFFFFFu);
        // bresenhamLine leaves a cell behind the hit alone when it shares its row or column with the hit cell (agent.cpp:555-560).
        // A path moves one cell along its major axis every step and never back along the minor one, so those cells are the
        // run of steps right behind the hit until the minor coordinate moves: its length per (step, beam), checked here
        k.ray_run.assign((size_t)k.ray_maxlen * k.ray_stride, 0);
        for (int b = 0; b < B; b++) {
            const size_t n = cells[b].size();
            for (size_t q = 0; q < n; q++) {
                const uint32_t xq = cells[b][q] / (uint32_t)Wv, yq = cells[b][q] % (uint32_t)Wv;
                size_t run = 0;
                bool in_run = true;
                for (size_t t = q + 1; t < n; t++) {
                    const bool same = cells[b][t] / (uint32_t)Wv == xq || cells[b][t] % (uint32_t)Wv == yq;
                    if (same && !in_run) k.ok = false;  // not one run: the packing of k_view's hit word would be wrong
                    if (same && in_run) run++;
                    if (!same) in_run = false;
                }
                k.ray_run[q * k.ray_stride + b] = (uint8_t)run;  // ray_maxlen <= 255 here
            }
        }
    }
    for (int b = 0; b < B; b++)
        for (size_t q = 0; q < cells[b].size(); q++) {
            if (k.big) {
                const uint32_t a = cells[b][q] / (uint32_t)Wv, bb = cells[b][q] % (uint32_t)Wv;
                k.big_cells[((q / 4) * (size_t)k.ray_stride + b) * 4 + q % 4] = big_bit_entry(((a >> 3) * (uint32_t)k.big_tb + (bb >> 3)) * 64u + (a & 7u) * 8u + (bb & 7u));
                k.ray_dist[q * k.ray_stride + b] = dists[b][q];
            } else {
                k.ray_rows[((q / 8) * (size_t)k.ray_stride + b) * 8 + (q % 8)] = (uint16_t)cells[b][q];
                k.ray_dist[q * k.ray_stride + b] = dists[b][q];
            }
            inv[cells[b][q]].push_back(((uint32_t)b << 16) | (uint32_t)q);
        }
    k.inv_pack.assign(NC, 0);
    uint32_t off = 0;
    for (int c = 0; c < NC; c++) {
        std::sort(inv[c].begin(), inv[c].end(), [](uint32_t a, uint32_t b) { return a > b; });  // beam descending
        if (!k.big && (off >= (1u << 20) || inv[c].size() >= (1u << 12))) k.ok = false;
        k.inv_pack[c] = off | ((uint32_t)inv[c].size() << 20);
        if (k.big) {
            k.big_inv.push_back(off);
            k.big_inv.push_back((uint32_t)inv[c].size());
        }
        off += (uint32_t)inv[c].size();
    }
    if (k.big_inv.empty()) k.big_inv.assign(2, 0);
    k.inv_ent.reserve(off + 1);
    for (int c = 0; c < NC; c++) k.inv_ent.insert(k.inv_ent.end(), inv[c].begin(), inv[c].end());
    k.top_ent.assign(NC, ((uint32_t)B << 16) | 0xFFFFu);  // no beam: the dummy beam B at a step behind every hit
    for (int c = 0; c < NC; c++)
        if (!inv[c].empty()) {
            k.top_ent[c] = inv[c][0];
        }
    if (k.inv_ent.empty()) k.inv_ent.push_back(0);
    k.dyn_c4.clear();
    if (!k.big)
        for (int c4 = 0; c4 < NC; c4 += 4) {
            bool dyn = false;
            for (int c = c4; c < std::min(c4 + 4, NC); c++)
                dyn = dyn || (B > 0 ? !inv[c].empty() : ((k.fov_bits[c >> 5] >> (c & 31)) & 1u) != 0u);
            if (dyn) k.dyn_c4.push_back((uint16_t)c4);
        }
    if (k.dyn_c4.empty()) k.dyn_c4.push_back(0);
    {
        auto word = [&](int c4) {
            uint32_t fov = 0, st = 0;
            for (int q = 0; q < 4 && c4 + q < NC; q++) {
                const int c = c4 + q;
                fov |= ((k.fov_bits[c >> 5] >> (c & 31)) & 1u) << q;
                st |= ((k.stamp_bits[c >> 5] >> (c & 31)) & 1u) << q;
            }
            return (uint32_t)c4 | (fov << 16) | (st << 20);
        };
        k.dyn_groups.clear();
        k.all_groups.clear();
        for (uint16_t c4 : k.dyn_c4) k.dyn_groups.push_back(word((int)c4));
        if (!k.big)
            for (int c4 = 0; c4 < NC; c4 += 4) k.all_groups.push_back(word(c4));
        if (k.all_groups.empty()) k.all_groups.push_back(0);
    }
    // k_view's step (5) for the cells a top beam leaves alone.  Such a cell keeps its 200 unless one of the lower beams through
    // it gets as far as the cell, so the kernel keeps the largest first-hit step of overlapping blocks of beams -- level v:
    // 16 << v beams starting every 8 << v, v = 0, 1, 2, stored back to back -- and compares the one block around the cell's
    // lower beams with their smallest step.  Per cell, one 8-byte record: block index | no such block << 13 | own footprint
    // << 14 | smallest step << 24, then the cell's inv_pack word.
    k.inv_cell.assign((size_t)NC * 2, 0);
    if (!k.big) {
        const uint32_t nb8 = ((uint32_t)B >> 3) + 1;
        const uint32_t lvl_n[3] = {nb8, (nb8 + 1) >> 1, (nb8 + 3) >> 2};
        const uint32_t lvl_off[3] = {0, lvl_n[0], lvl_n[0] + lvl_n[1]};
        for (int c = 0; c < NC; c++) {
            const uint32_t st = (k.stamp_bits[c >> 5] >> (c & 31)) & 1u;
            k.inv_cell[2 * (size_t)c] = (1u << 13) | (st << 14);
            k.inv_cell[2 * (size_t)c + 1] = k.inv_pack[c];
            if (inv[c].size() < 2) continue;
            uint32_t bmin = 0xFFFFFFFFu, bmax = 0, kkmin = 0xFFFFu;
            for (size_t e = 1; e < inv[c].size(); e++) {
                bmin = std::min(bmin, inv[c][e] >> 16);
                bmax = std::max(bmax, inv[c][e] >> 16);
                kkmin = std::min(kkmin, inv[c][e] & 0xFFFFu);
            }
            uint32_t idx = 0, none = 1;
            for (uint32_t v = 0; v < 3 && none; v++) {
                const uint32_t i = (bmin >> 3) >> v;  // block i of level v covers beams [i * (8 << v), (i + 2) * (8 << v))
                if (bmax < (i + 2) * (8u << v) && lvl_off[v] + i < (1u << 13)) {
                    idx = lvl_off[v] + i;
                    none = 0;
                }
            }
            k.inv_cell[2 * (size_t)c] = idx | (none << 13) | (st << 14) | (std::min(kkmin, 0xFFu) << 24);
        }
    }
}

// cv2.resize(view, image_size, INTER_CUBIC) (yaml_env.py:431-438) reads 4 x 4 view cells per pixel of the sensor_map (no
// anti-aliasing): k_taps_big evaluates exactly those cells.  Per (tap, pixel), tap-major so that the pixels of a wavefront
// read consecutive words: the hot word = the cell's top beam entry | own footprint << 31 (what nearly every tap is decided
// by), and two cold ones -- the cell's ray list {first entry, count} and its bit address in the tiled crop bitmap -- for the
// taps a top beam leaves alone or the own footprint covers.  xofs / yofs: source index of tap 1 per destination column / row
// (csrc/cv_resize.h), borders replicated as OpenCV does.
#define TAP_CHUNK_PIXELS 256  // (= VBT_T, k_taps_big's workgroup)
static void build_big_taps(RobotClassHost& k, const ViewGeom& g, const std::vector<int>& xofs, const std::vector<int>& yofs) {
    const int IW = (int)xofs.size(), IH = (int)yofs.size(), NP = IW * IH;
    std::vector<char> chunk_dyn((size_t)(NP + TAP_CHUNK_PIXELS - 1) / TAP_CHUNK_PIXELS, g.B > 0 ? 0 : 1);  // (no laser: the crop decides everywhere)
    k.tap_top.assign((size_t)16 * NP, 0);
    k.tap_inv.assign((size_t)16 * NP * 2, 0);
    k.tap_addr.assign((size_t)16 * NP, 0);
    for (int dy = 0; dy < IH; dy++)
        for (int dx = 0; dx < IW; dx++)
            for (int kr = 0; kr < 4; kr++)
                for (int j = 0; j < 4; j++) {
                    const int a = std::min(std::max(yofs[dy] - 1 + kr, 0), g.Hv - 1), b = std::min(std::max(xofs[dx] - 1 + j, 0), g.Wv - 1);
                    const uint32_t c = (uint32_t)a * (uint32_t)g.Wv + (uint32_t)b;
                    const uint32_t st = (k.stamp_bits[c >> 5] >> (c & 31)) & 1u;
                    const size_t at = (size_t)(kr * 4 + j) * NP + (size_t)dy * IW + dx;
                    k.tap_top[at] = k.top_ent[c] | (st << 31);  // fewer than 32768 beams in a big view
                    if ((k.top_ent[c] >> 16) != (uint32_t)g.B) chunk_dyn[(size_t)(dy * IW + dx) / TAP_CHUNK_PIXELS] = 1;  // some beam crosses the cell
                    k.tap_inv[2 * at] = k.big_inv[2 * (size_t)c];
                    k.tap_inv[2 * at + 1] = k.big_inv[2 * (size_t)c + 1];
                    k.tap_addr[at] = (((uint32_t)a >> 3) * (uint32_t)k.big_tb + ((uint32_t)b >> 3)) * 64u + ((uint32_t)a & 7u) * 8u + ((uint32_t)b & 7u);
                }
    k.tap_chunks.clear();
    for (size_t q = 0; q < chunk_dyn.size(); q++)
        if (chunk_dyn[q]) k.tap_chunks.push_back((uint16_t)q);
    if (k.tap_chunks.empty()) k.tap_chunks.push_back(0);  // (the robot's first workgroup also commits the collision code)
}

struct PedClassHost {
    int shape;
    float size[6];
    double sizes[6];
    Pts bbox, left, right;
    std::vector<FpRow> bbox_rows, left_rows, right_rows;  // lattice rows (fp_rows.h) of the three sample lists
    double bbox_cy = 0;
    int box_rad = 0;  // cells around the pedestrian's own cell that hold its footprint whatever the gait state (SUM mode's LDS box)
};

// PedAgent::init_shape (agent.cpp:666-685)
static void build_ped_class(PedClassHost& k, double res) {
    for (int j = 0; j < 6; j++) k.sizes[j] = (double)k.size[j];
    if (k.shape == IMGENV_SHAPE_LEG) {
        k.left = shape_circle(0, 0, k.sizes[2]);
        k.right = shape_circle(0, 0, k.sizes[5]);
        k.left_rows = build_fp_rows(k.left, 0.0, res);
        k.right_rows = build_fp_rows(k.right, 0.0, res);
        if (k.left_rows.empty() || k.right_rows.empty()) k.left_rows.clear(), k.right_rows.clear();
    } else if (k.shape == IMGENV_SHAPE_CIRCLE) {
        k.bbox = shape_circle(k.sizes[0], k.sizes[1], k.sizes[2]);
        k.bbox_cy = k.sizes[1];
        k.bbox_rows = build_fp_rows(k.bbox, k.bbox_cy, res);
    } else {
        k.bbox = shape_rectangle(k.sizes);
        k.bbox_rows = build_fp_rows(k.bbox, 0.0, res);
    }
    double ext = 0;
    for (int q = 0; q < k.bbox.n(); q++) ext = std::max(ext, sqrt(k.bbox.x[q] * k.bbox.x[q] + k.bbox.y[q] * k.bbox.y[q]));
    if (k.shape == IMGENV_SHAPE_LEG) {  // leg centres: x in {sizes, -+0.15, -+0.3} (update_bbox, agent.cpp:696-735; step_len_ 0.3), y as given
        const double lx = std::max(0.3, std::max(fabs(k.sizes[0]), fabs(k.sizes[3]))), ly = std::max(fabs(k.sizes[1]), fabs(k.sizes[4]));
        ext = sqrt(lx * lx + ly * ly) + std::max(k.sizes[2], k.sizes[5]) + 0.01;
    }
    k.box_rad = (int)ceil(ext / res) + 2;
}

// Python round(x, 2) (float.__round__ is a correctly rounded decimal rounding)
static double py_round2(double x) {
    char buf[64];
    snprintf(buf, sizeof(buf), "%.2f", x);
    return strtod(buf, nullptr);
}

// numpy float32 -> float16, round to nearest even
static uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const uint32_t shift = (uint32_t)(14 - e);
        uint32_t hm = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(sign | hm);
    }
    const uint32_t hm = m >> 13, rem = m & 0x1fffu;
    uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | hm);
    if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) h++;
    return h;
}

// Agent::get_corners (agent.cpp:626-651)
static void get_corners(int shape, const double* s, const Tf2& bw, double& pax, double& pay, double& pbx, double& pby) {
    if (shape == IMGENV_SHAPE_CIRCLE) {
        tf_apply(bw, s[0] - s[2], s[1] - s[2], pax, pay);
        tf_apply(bw, s[0] + s[2], s[1] + s[2], pbx, pby);
    } else {
        tf_apply(bw, s[0], s[2], pax, pay);
        tf_apply(bw, s[1], s[3], pbx, pby);
    }
}

// ---------------------------------------------------------------------------------------------
// RVO2 obstacle list + BSP tree, host side (RVOSimulator::addObstacle RVOSimulator.cpp:130-170,
// KdTree::buildObstacleTree KdTree.cpp:119-257), float32 like the reference.  Built at reset,
// uploaded as flat arrays; the device replays KdTree::queryObstacleTreeRecursive on them.
struct RvoObstHost {
    float px, py, ux, uy;
    int is_convex, next, prev;
};
struct RvoNodeHost {
    int obstacle, left, right;
};

// (restates RVO2's KdTree::buildObstacleTree -- Copyright 2008 University of North Carolina at Chapel Hill, Apache License 2.0; see NOTICE)
struct RvoObstacles {
    std::vector<RvoObstHost> ob;
    std::vector<RvoNodeHost> nodes;
    int root = -1;

    static float det(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }
    float left_of(int a, int b, int c) const {  // leftOf(a, b, c) = det(a - c, b - a)
        return det(ob[a].px - ob[c].px, ob[a].py - ob[c].py, ob[b].px - ob[a].px, ob[b].py - ob[a].py);
    }

    void clear() {
        ob.clear();
        nodes.clear();
        root = -1;
    }

    void add(const float* xy, int n) {
        const int first = (int)ob.size();
        for (int i = 0; i < n; i++) {
            RvoObstHost o;
            o.px = xy[2 * i];
            o.py = xy[2 * i + 1];
            o.next = o.prev = -1;
            const int k = (int)ob.size();
            if (i != 0) {
                o.prev = k - 1;
                ob[k - 1].next = k;
            }
            const int inext = (i == n - 1 ? 0 : i + 1), iprev = (i == 0 ? n - 1 : i - 1);
            const float dx = xy[2 * inext] - xy[2 * i], dy = xy[2 * inext + 1] - xy[2 * i + 1];
            const float inv = 1.0f / sqrtf(dx * dx + dy * dy);  // normalize(): v / abs(v) multiplies by 1/|v|
            o.ux = dx * inv;
            o.uy = dy * inv;
            if (n == 2) {
                o.is_convex = 1;
            } else {
                // leftOf(prev, cur, next) >= 0
                const float ax = xy[2 * iprev] - xy[2 * inext], ay = xy[2 * iprev + 1] - xy[2 * inext + 1];
                const float bx = xy[2 * i] - xy[2 * iprev], by = xy[2 * i + 1] - xy[2 * iprev + 1];
                o.is_convex = det(ax, ay, bx, by) >= 0.0f;
            }
            ob.push_back(o);
            if (i == n - 1) {
                ob[k].next = first;
                ob[first].prev = k;
            }
        }
    }

    static bool pair_ge(size_t a1, size_t a2, size_t b1, size_t b2) { return (a1 > b1) || (a1 == b1 && a2 >= b2); }

    int build(const std::vector<int>& obs) {
        const float EPS = 0.00001f;
        const size_t n = obs.size();
        if (n == 0) return -1;
        const int node = (int)nodes.size();
        nodes.push_back(RvoNodeHost{-1, -1, -1});
        size_t optimal = 0, min_left = n, min_right = n;
        for (size_t i = 0; i < n; i++) {
            size_t ls = 0, rs = 0;
            const int i1 = obs[i], i2 = ob[i1].next;
            for (size_t j = 0; j < n; j++) {
                if (i == j) continue;
                const int j1 = obs[j], j2 = ob[j1].next;
                const float a = left_of(i1, i2, j1), b = left_of(i1, i2, j2);
                if (a >= -EPS && b >= -EPS) ++ls;
                else if (a <= EPS && b <= EPS) ++rs;
                else { ++ls; ++rs; }
                if (pair_ge(std::max(ls, rs), std::min(ls, rs), std::max(min_left, min_right), std::min(min_left, min_right))) break;
            }
            if (!pair_ge(std::max(ls, rs), std::min(ls, rs), std::max(min_left, min_right), std::min(min_left, min_right))) {
                min_left = ls;
                min_right = rs;
                optimal = i;
            }
        }
        std::vector<int> L, Rr;
        const int i1 = obs[optimal], i2 = ob[i1].next;
        for (size_t j = 0; j < n; j++) {
            if (j == optimal) continue;
            const int j1 = obs[j], j2 = ob[j1].next;
            const float a = left_of(i1, i2, j1), b = left_of(i1, i2, j2);
            if (a >= -EPS && b >= -EPS) {
                L.push_back(j1);
            } else if (a <= EPS && b <= EPS) {
                Rr.push_back(j1);
            } else {
                const float ex = ob[i2].px - ob[i1].px, ey = ob[i2].py - ob[i1].py;
                const float t = det(ex, ey, ob[j1].px - ob[i1].px, ob[j1].py - ob[i1].py) /
                                det(ex, ey, ob[j1].px - ob[j2].px, ob[j1].py - ob[j2].py);
                RvoObstHost no;
                no.px = ob[j1].px + t * (ob[j2].px - ob[j1].px);
                no.py = ob[j1].py + t * (ob[j2].py - ob[j1].py);
                no.prev = j1;
                no.next = j2;
                no.is_convex = 1;
                no.ux = ob[j1].ux;
                no.uy = ob[j1].uy;
                const int nn = (int)ob.size();
                ob.push_back(no);
                ob[j1].next = nn;
                ob[j2].prev = nn;
                if (a > 0.0f) {
                    L.push_back(j1);
                    Rr.push_back(nn);
                } else {
                    Rr.push_back(j1);
                    L.push_back(nn);
                }
            }
        }
        nodes[node].obstacle = i1;
        const int l = build(L);
        const int r = build(Rr);
        nodes[node].left = l;
        nodes[node].right = r;
        return node;
    }

    void process() {
        nodes.clear();
        std::vector<int> all(ob.size());
        for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
        root = build(all);
    }
};


// ---------------------------------------------------------------------------------------------
// glibc rand() for the beep lottery (img_env.cpp:327), in the form k_beep consumes: the last 31 words of the TYPE_3 stream
// of a fresh process (seed 1, 310 discarded draws: glibc stdlib/random_r.c) and the jump table of the linear recurrence
// r[i] = r[i-3] + r[i-31] (mod 2^32).
static void beep_initial_state(uint32_t s[31]) {
    std::vector<uint32_t> r(344);
    r[0] = 1;
    for (int i = 1; i < 31; i++) {
        const long prev = (int32_t)r[i - 1];
        long word = 16807 * (prev % 127773) - 2836 * (prev / 127773);
        if (word < 0) word += 2147483647;
        r[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    for (int i = 34; i < 344; i++) r[i] = r[i - 31] + r[i - 3];
    for (int j = 0; j < 31; j++) s[j] = r[344 - 31 + j];
}
// coef[k][j]: next word k (0-based) = sum_j coef[k][j] * last31[j]
static std::vector<uint32_t> beep_coefficients(int n_words) {
    std::vector<uint32_t> c((size_t)(31 + n_words) * 31, 0u);
    for (int j = 0; j < 31; j++) c[(size_t)j * 31 + j] = 1u;
    for (int k = 31; k < 31 + n_words; k++)
        for (int j = 0; j < 31; j++) c[(size_t)k * 31 + j] = c[(size_t)(k - 31) * 31 + j] + c[(size_t)(k - 3) * 31 + j];
    return std::vector<uint32_t>(c.begin() + 31 * 31, c.end());
}

// ---------------------------------------------------------------------------------------------
// libpedsim / PedScene construction on the host (pedscene.h:57-80, ped_agent.cpp:24-58): the per-agent
// random vmax and the peds' rand() start positions that seed the quadtree.  Both generators are
// third-party (libstdc++ <random>, glibc rand()), restated from their published algorithms; one handle is one
// fresh node process, so both start from their default seeds.
struct PedsimRng {
    unsigned long lcg = 1u;  // std::default_random_engine = minstd_rand0, default seed 1
    int32_t r[34 + 310 + 4096];
    int rk = -1;
    double lcg_next() {
        lcg = (lcg * 16807ul) % 2147483647ul;
        return (double)lcg;
    }
    double canonical() {  // std::generate_canonical<double, 53>: 2 draws of range 2147483646
        const double R = 2147483646.0;
        double sum = 0.0, tmp = 1.0;
        for (int k = 0; k < 2; k++) {
            sum += (lcg_next() - 1.0) * tmp;
            tmp *= R;
        }
        double ret = sum / tmp;
        if (ret >= 1.0) ret = nextafter(1.0, 0.0);
        return ret;
    }
    double normal_fresh(double mean, double stddev) {  // a new normal_distribution per Tagent: polar method, first value
        double x, y, r2;
        do {
            x = 2.0 * canonical() - 1.0;
            y = 2.0 * canonical() - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        const double mult = sqrt(-2 * log(r2) / r2);
        return (y * mult) * stddev + mean;
    }
    int glibc_rand() {  // TYPE_3 additive feedback generator, seed 1 (glibc stdlib/random_r.c)
        if (rk < 0) {
            r[0] = 1;
            for (int i = 1; i < 31; i++) {
                long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
                long word = 16807 * lo - 2836 * hi;
                if (word < 0) word += 2147483647;
                r[i] = (int32_t)word;
            }
            for (int i = 31; i < 34; i++) r[i] = r[i - 31];
            for (int i = 34; i < 344; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
            rk = 344;
        }
        if (rk >= (int)(sizeof(r) / sizeof(r[0]))) {
            memmove(r, r + rk - 34, sizeof(int32_t) * 34);
            rk = 34;
        }
        r[rk] = (int32_t)((uint32_t)r[rk - 31] + (uint32_t)r[rk - 3]);
        return (int)(((uint32_t)r[rk++]) >> 1);
    }
};
