// CPU check of img_env_amd/csrc/fp_rows.h (the row-by-row footprint raster of k_raster) against the literal walk over every
// footprint sample, which is the definition (Agent::draw, agent.cpp:285-327; GridMap::world2map, grid_map.cpp:40-44).
//   fp_rows_check <shape 0 circle | 1 rectangle | 2 leg disc> <s0> <s1> <s2> <s3> <res> <poses> <seed>
// For every pose: the cells + last sample index of the literal walk, and the same from the certified rows; wherever every row was
// certified the two must be identical.  Prints "OK certified=<fraction> ..." or the first mismatch.  Poses: uniform random, and
// adversarial ones that put lattice samples exactly onto rounding boundaries (round-number positions, axis-parallel headings,
// headings a few ulps off an axis), where the certification has to refuse instead of guessing.
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <random>

#define WAVE_SZ 64
#include "../../img_env_amd/csrc/host_tables.h"

typedef std::map<std::pair<int, int>, uint32_t> CellMap;

static CellMap literal(const Pts& p, const Tf2& bw, double res, double lx, double ly, bool leg) {
    CellMap out;
    Tf2 lb;
    tf_set_rotation_zw(lb, 0.0, 1.0);
    lb.ox = lx;
    lb.oy = ly;
    for (int q = 0; q < p.n(); q++) {
        double bx = p.x[q], by = p.y[q], wx, wy;
        if (leg) tf_apply(lb, p.x[q], p.y[q], bx, by);  // PedAgent::leg2base (agent.cpp:831-837)
        tf_apply(bw, bx, by, wx, wy);
        out[{w2m(wx, res), w2m(wy, res)}] = (uint32_t)q + 1;  // x-major order: later samples overwrite
    }
    return out;
}

static bool by_rows(const std::vector<FpRow>& rows, double cy, const Tf2& bw, double res, double lx, double ly, bool leg, CellMap& out) {
    const FpRowsPose P = fpr_pose(bw.m00, bw.m01, bw.m10, bw.m11, bw.ox, bw.oy, leg ? ly : cy, res);
    bool all = true;
    for (FpRow r : rows) {
        if (leg) r.px = r.px + lx;
        FpAxis ax, ay;
        if (!fpr_row(P, r, ax, ay)) {
            all = false;
            continue;
        }
        for (int i = 0; i <= FPR_MAXC; i++)
            for (int j = 0; j <= FPR_MAXC; j++) {
                int m, n;
                uint32_t last;
                if (fpr_piece(r, ax, ay, i, j, m, n, last)) {
                    uint32_t& v = out[{m, n}];
                    v = std::max(v, last);
                }
            }
    }
    return all;
}

int main(int argc, char** argv) {
    if (argc < 9) return 2;
    const int shape = atoi(argv[1]);
    double s[4] = {atof(argv[2]), atof(argv[3]), atof(argv[4]), atof(argv[5])};
    for (double& v : s) v = (double)(float)v;  // sizes are float32 on the wire
    const double res = (double)(float)atof(argv[6]);
    const long poses = atol(argv[7]);
    std::mt19937_64 rng((unsigned long long)atoll(argv[8]));
    const bool leg = shape == 2;
    const Pts p = shape == 1 ? shape_rectangle(s) : leg ? shape_circle(0, 0, s[2]) : shape_circle(s[0], s[1], s[2]);
    const double cy = shape == 0 ? s[1] : 0.0;
    const std::vector<FpRow> rows = build_fp_rows(p, cy, res);
    if (rows.empty()) {
        printf("OK no rows at this resolution (the class walks its samples)\n");
        return 0;
    }
    std::uniform_real_distribution<double> U(0.0, 1.0);
    long certified = 0, adversarial = 0, adv_certified = 0;
    const double special_yaw[] = {0.0, M_PI / 2, M_PI, -M_PI / 2, 1e-15, -3e-13, 1e-9, 2e-7, M_PI / 4, 0.7853981633974484, 3.14159};
    for (long it = 0; it < poses; it++) {
        double x = 3.0 + 90.0 * U(rng), y = 3.0 + 90.0 * U(rng), yaw = (U(rng) * 2 - 1) * 3.2, lx = 0, ly = 0;
        const bool adv = it % 4 == 3;
        if (adv) {  // positions on multiples of the pitch / the cell, half cells, headings on and next to the axes
            const double grain[] = {res, res / 2, 0.01, 0.005, 0.125, 1.0};
            x = rint(x / grain[it % 6]) * grain[it % 6] + (it % 3 == 0 ? res / 2 : 0.0);
            y = rint(y / grain[(it / 6) % 6]) * grain[(it / 6) % 6] + (it % 5 == 0 ? res / 2 : 0.0);
            yaw = special_yaw[(it / 4) % 11] + ((it / 44) % 3 == 1 ? 1e-16 * (double)(it % 97) : 0.0);
            adversarial++;
        }
        if (leg) {  // the gait's leg offsets (agent.cpp:696-735)
            const double off[] = {0.0, -0.15, 0.15, -0.3, 0.3};
            lx = off[it % 5];
            ly = it % 2 ? 0.1 : -0.1;
        }
        const Tf2 bw = tf_from_pose(x, y, yaw);
        const CellMap want = literal(p, bw, res, lx, ly, leg);
        CellMap got;
        if (!by_rows(rows, cy, bw, res, lx, ly, leg, got)) continue;
        certified++;
        adv_certified += adv;
        if (got != want) {
            printf("MISMATCH pose (%.17g, %.17g, %.17g) leg (%g, %g): literal %zu cells, rows %zu cells\n", x, y, yaw, lx, ly, want.size(), got.size());
            for (auto& kv : want) {
                auto f = got.find(kv.first);
                if (f == got.end() || f->second != kv.second)
                    printf("  cell (%d, %d): literal last %u, rows %s\n", kv.first.first, kv.first.second, kv.second,
                           f == got.end() ? "absent" : std::to_string(f->second).c_str());
            }
            for (auto& kv : got)
                if (!want.count(kv.first)) printf("  cell (%d, %d): rows only (last %u)\n", kv.first.first, kv.first.second, kv.second);
            return 1;
        }
    }
    const double frac = (double)(certified - adv_certified) / (double)std::max(1L, poses - adversarial);
    printf("OK certified=%.6f of random poses, %ld of %ld adversarial ones; %zu rows, %d samples\n", frac, adv_certified, adversarial, rows.size(), p.n());
    return frac > 0.999 ? 0 : 3;
}
