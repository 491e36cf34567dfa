// CPU check of the static tables behind the big-view kernels (img_env_amd/csrc/view_big.h; host_tables.h build_robot_class with
// force_big), without a GPU:
//   1. big_cells: every step of every beam decodes (big_bit_entry: word byte address << 5 | bit) to a bit inside the tiled crop
//      bitmap, consecutive steps are 8-neighbours in the view (Bresenham), the table's cell equals the tiled address of the view
//      cell the per-cell ray lists (big_inv / inv_ent) name for that (beam, step), and the padding behind a beam's end points at
//      the always-free bit behind the bitmap;
//   2. crop_tiles / crop_masks: the masks are exactly the field-of-view cells, every tile once, 8 zero masks behind the list;
//   3. ray_end >= its own step and inside the beam;
//   4. the index arithmetic k_crop_big and the stamps use for crop_map (8 x 8-cell blocks): the six-instruction form of the blocked
//      index against its definition, and the multiply-shift row of a cell index against a division, over a whole map.
// usage: big_view_tables_check <view_w> <view_h> <res> <beams> <angle_begin> <angle_end> <map_h> <map_w> ; exit code 0 = all good
#include <stdio.h>

#define WAVE_SZ 64
#include "../../img_env_amd/csrc/host_tables.h"
#include "../../img_env_amd/csrc/cv_resize.h"

static int fail(const char* what, long a = 0, long b = 0, long c = 0) {
    printf("FAIL %s (%ld %ld %ld)\n", what, a, b, c);
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 9) return fail("usage");
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
    const int Hg = atoi(argv[7]), Wg = atoi(argv[8]);
    const ViewGeom g = make_view_geom(c);
    RobotClassHost k;
    k.shape = IMGENV_SHAPE_CIRCLE;
    k.size[0] = 0.f; k.size[1] = 0.f; k.size[2] = 0.17f; k.size[3] = 0.f;
    k.sensor[0] = 0.f; k.sensor[1] = 0.f;
    build_robot_class(k, g, true);
    if (!k.ok || !k.big) return fail("no big class");
    const int Hv = g.Hv, Wv = g.Wv, NC = Hv * Wv, B = g.B, S = k.ray_stride;
    const uint32_t n_bits = (uint32_t)k.big_ta * k.big_tb * 64u;
    auto tiled = [&](int a, int b) { return (uint32_t)(((a >> 3) * k.big_tb + (b >> 3)) * 64 + (a & 7) * 8 + (b & 7)); };
    // view cell of a tiled bit address
    auto cell_of = [&](uint32_t bit, int& a, int& b) {
        const uint32_t tile = bit >> 6;
        a = (int)(tile / (uint32_t)k.big_tb) * 8 + (int)((bit >> 3) & 7u);
        b = (int)(tile % (uint32_t)k.big_tb) * 8 + (int)(bit & 7u);
    };
    // 1. the paths
    std::vector<std::vector<uint32_t>> step_bit(B);
    for (int b = 0; b < B; b++) {
        const int len = k.ray_len[b];
        if (len > k.ray_kpad) return fail("ray_len beyond the padded table", b, len);
        int pa = 0, pb = 0;
        for (int q = 0; q < k.ray_kpad; q++) {
            const uint32_t e = k.big_cells[((q / 4) * (size_t)S + b) * 4 + q % 4];
            if (e & 0x60u) return fail("entry bits 5-6 not clear", b, q, e);
            const uint32_t bit = (e >> 7) * 32u + (e & 31u);
            if (q >= len) {
                if (bit != n_bits) return fail("padding is not the free bit", b, q, bit);
                continue;
            }
            if (bit >= n_bits) return fail("bit outside the bitmap", b, q, bit);
            int a, bb;
            cell_of(bit, a, bb);
            if (a >= Hv || bb >= Wv) return fail("cell outside the view", b, q, a * Wv + bb);
            if (q > 0 && (abs(a - pa) > 1 || abs(bb - pb) > 1 || (a == pa && bb == pb))) return fail("steps are not 8-neighbours", b, q);
            pa = a; pb = bb;
            step_bit[b].push_back(bit);
            // 3. ray_end
            const int end = k.ray_end[(size_t)q * S + b];
            if (end < q || end >= len) return fail("ray_end", b, q, end);
        }
    }
    // ... against the per-cell ray lists: cell c lists (beam << 16 | step) entries; each must name a step whose bit is c's
    size_t listed = 0;
    for (int cc = 0; cc < NC; cc++) {
        const uint32_t first = k.big_inv[2 * (size_t)cc], count = k.big_inv[2 * (size_t)cc + 1];
        for (uint32_t e = 0; e < count; e++) {
            const uint32_t ent = k.inv_ent[first + e], beam = ent >> 16, step = ent & 0xFFFFu;
            if (beam == (uint32_t)B) continue;  // the dummy beam of cells no beam crosses
            if (beam > (uint32_t)B || step >= step_bit[beam].size()) return fail("ray list entry out of range", cc, beam, step);
            if (step_bit[beam][step] != tiled(cc / Wv, cc % Wv)) return fail("ray list and path table disagree", cc, beam, step);
            listed++;
        }
    }
    size_t steps = 0;
    for (int b = 0; b < B; b++) steps += step_bit[b].size();
    if (listed != steps) return fail("ray lists do not cover the paths", (long)listed, (long)steps);
    // 2. crop tiles
    if ((int)k.crop_tiles.size() != k.n_crop || k.crop_masks.size() != (size_t)k.n_crop + 8) return fail("crop list sizes", k.n_crop);
    for (int q = 0; q < 8; q++)
        if (k.crop_masks[(size_t)k.n_crop + q] != 0) return fail("mask padding", q);
    std::vector<uint8_t> seen((size_t)k.big_ta * k.big_tb, 0);
    size_t fov_cells = 0, mask_cells = 0;
    for (int cc = 0; cc < NC; cc++) fov_cells += (k.fov_bits[cc >> 5] >> (cc & 31)) & 1u;
    for (int t = 0; t < k.n_crop; t++) {
        const uint32_t ta = k.crop_tiles[t] >> 16, tb = k.crop_tiles[t] & 0xFFFFu;
        if (ta >= (uint32_t)k.big_ta || tb >= (uint32_t)k.big_tb) return fail("tile outside the view", t);
        if (fov_cells && seen[(size_t)ta * k.big_tb + tb]++) return fail("tile listed twice", t);
        for (int q = 0; q < 64; q++) {
            if (!((k.crop_masks[t] >> q) & 1ull)) continue;
            const int a = (int)ta * 8 + (q >> 3), b = (int)tb * 8 + (q & 7), cc = a * Wv + b;
            if (a >= Hv || b >= Wv || !((k.fov_bits[cc >> 5] >> (cc & 31)) & 1u)) return fail("mask bit outside the field of view", t, q);
            mask_cells++;
        }
    }
    if (mask_cells != fov_cells) return fail("masks do not cover the field of view", (long)mask_cells, (long)fov_cells);
    // 4. crop_map index arithmetic (kernels.h crop_tiled / crop_mark, view_big.h crop_tiled_at)
    const uint32_t wt = (uint32_t)(Wg + 7) / 8;
    const unsigned long long magic = (((unsigned long long)1 << 40) + (unsigned long long)Wg - 1) / (unsigned long long)Wg;
    for (uint32_t m = 0; m < (uint32_t)Hg; m++)
        for (uint32_t n = 0; n < (uint32_t)Wg; n++) {
            const uint32_t def = ((m >> 3) * wt + (n >> 3)) * 64u + (m & 7u) * 8u + (n & 7u);
            const uint32_t n6 = ((n & ~7u) << 3) | (n & 7u);
            const uint32_t fast = (m << 3) + (((m >> 3) & 0xFFFFFFu) * (((wt - 1u) << 6) & 0xFFFFFFu) + n6);  // __umul24
            const uint32_t plain = (((m >> 3) * wt + (n >> 3)) << 6) | ((m & 7u) << 3) | (n & 7u);
            if (fast != def || plain != def) return fail("blocked index", m, n);
            const unsigned long long cl = (unsigned long long)m * Wg + n;
            if ((uint32_t)((cl * magic) >> 40) != m) return fail("multiply-shift row", m, n);
        }
    // 5. the chunks of pixels k_taps_big covers in a STEP (round 4; 48 x 48 pixels shrunk with INTER_CUBIC as the shipped config
    //    does): a pixel one of whose 4 x 4 source cells is crossed by a beam lies in a listed chunk, a listed chunk holds such
    //    a pixel, the chunks are listed once, in ascending order
    size_t n_chunks_listed = 0;
    {
        const int IW = 48, IH = 48, NP = IW * IH;
        const CvAxis ax = cv_axis(Wv, IW, true, true), ay = cv_axis(Hv, IH, true, false);
        build_big_taps(k, g, ax.ofs, ay.ofs);
        const int n_chunks = (NP + TAP_CHUNK_PIXELS - 1) / TAP_CHUNK_PIXELS;
        std::vector<char> want(n_chunks, 0), got(n_chunks, 0);
        for (int p = 0; p < NP; p++)
            for (int j = 0; j < 16; j++)
                if (((k.tap_top[(size_t)j * NP + p] >> 16) & 0x7FFFu) != (uint32_t)B) want[p / TAP_CHUNK_PIXELS] = 1;
        int prev = -1;
        for (uint16_t q : k.tap_chunks) {
            if ((int)q <= prev || (int)q >= n_chunks) return fail("tap_chunks: order / range", q, prev);
            prev = q;
            got[q] = 1;
        }
        bool any = false;
        for (int q = 0; q < n_chunks; q++) any = any || want[q];
        for (int q = 0; q < n_chunks; q++)
            if (want[q] != got[q] && !(!any && q == 0)) return fail("tap_chunks: chunk", q, want[q], got[q]);
        n_chunks_listed = k.tap_chunks.size();
    }
    printf("OK %d x %d view, %d beams, %d crop tiles, %zu path steps, %zu pixel chunks in a step\n", Hv, Wv, B, k.n_crop, steps, n_chunks_listed);
    return 0;
}
