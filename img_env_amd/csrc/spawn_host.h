// spawn_host.h -- random placement of obstacles, robots and pedestrians for an episode, on the host.
//
// The role of the reference's EnvPos.reset (envs/utils/reset_helper.py:104-345) for the pose types the shipped
// robot_nav configs use on plain ranges: `fix`, `rand_angle`, `range` starts and `range` / `range_view` targets.
// The rejection rules are the reference's:
//   * starts keep > clearance (1.0 m, free_check_robo_ped, reset_helper.py:35-43) to every other start and
//     > module size + obstacle radius to every obstacle (free_check_obj, 46-55);
//   * targets keep > target_min_dist to their own start, > clearance to every other target, and clear the obstacles;
//     a `range_view` target lies in the 4 m box around its start but outside the 2.5 m box (random_view, 62-82);
//     after 50 failed targets the start is drawn again (296-300);
//   * pedestrians walk to their target and, with go_back, back to their start (337-342).
// What is NOT the reference's: the random stream (a splitmix64 / xoshiro256** generator seeded per world instead of
// Python's Mersenne Twister), and the neighbour search (a hash grid instead of O(n^2) list scans).  The same rules in
// Python are img_env_amd/spawn.py; tests/test_host_logic.py holds both to them.
//
// Why it is native: a world of a multi-world handle is reset whenever its episode ends -- dozens of worlds on every step --
// and the Python spawn costs 40-170 us per small world, twenty times the device's whole step.
#pragma once
#include <math.h>
#include <stdint.h>

#include <unordered_map>
#include <vector>

#include "../../include/imgenv.h"

struct SpawnRng {  // xoshiro256** seeded through splitmix64
    uint64_t s[4];
    explicit SpawnRng(uint64_t seed) {
        for (int k = 0; k < 4; k++) {
            uint64_t z = (seed += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[k] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0, 1)
    double uniform(double a, double b) { return a + (b - a) * unit(); }
};

struct SpawnGrid {  // points kept apart by `cell`: a hash of cell -> points
    double cell;
    std::unordered_map<int64_t, std::vector<std::pair<double, double>>> buckets;
    explicit SpawnGrid(double c) : cell(c > 1e-6 ? c : 1e-6) {}
    static int64_t key(int64_t i, int64_t j) { return i * 0x100000000ll + j; }
    bool ok(double x, double y, double d) const {
        const int64_t kx = (int64_t)floor(x / cell), ky = (int64_t)floor(y / cell), r = (int64_t)ceil(d / cell);
        const double d2 = d * d;
        for (int64_t i = kx - r; i <= kx + r; i++)
            for (int64_t j = ky - r; j <= ky + r; j++) {
                auto it = buckets.find(key(i, j));
                if (it == buckets.end()) continue;
                for (const auto& p : it->second)
                    if ((p.first - x) * (p.first - x) + (p.second - y) * (p.second - y) <= d2) return false;
            }
        return true;
    }
    void add(double x, double y) { buckets[key((int64_t)floor(x / cell), (int64_t)floor(y / cell))].push_back({x, y}); }
};

struct SpawnOut {  // one world's reset batch, owned
    std::vector<int32_t> obs_shape, ped_traj_len;
    std::vector<float> obs_size;
    std::vector<double> obs_pose, robot_pose, robot_goal, ped_pose, ped_goal, ped_traj;
    imgenv_reset_batch batch;
};

static inline void spawn_pose4(double x, double y, double yaw, double* out4) {  // quaternion_from_euler(0, 0, yaw) (ros_utils.py:22-23)
    out4[0] = x;
    out4[1] = y;
    out4[2] = sin(yaw / 2.0);
    out4[3] = cos(yaw / 2.0);
}

// returns nullptr on success, else what could not be placed
static inline const char* spawn_world(const imgenv_spawn_cfg& c, uint64_t seed, SpawnOut& o) {
    SpawnRng rng(seed);
    const int nr = c.n_robots, np = c.n_peds, n = nr + np, nob = c.n_obstacles;
    auto rand_pose = [&](const double* r, int has_yaw, double* p) {  // _rand_pose (reset_helper.py: uniform box, yaw in +-3.14 by default)
        p[0] = rng.uniform(r[0], r[1]);
        p[1] = rng.uniform(r[2], r[3]);
        p[2] = has_yaw ? rng.uniform(r[4], r[5]) : rng.uniform(-3.14, 3.14);
    };
    // obstacles (reset_helper.py:122-165)
    o.obs_shape.assign(nob, 0);
    o.obs_size.assign((size_t)nob * 4, 0.0f);
    o.obs_pose.assign((size_t)nob * 4, 0.0);
    std::vector<double> obs_range((size_t)nob * 3);  // x, y, radius
    for (int i = 0; i < nob; i++) {
        const imgenv_spawn_obstacle& q = c.obstacles[i];
        double radius;
        if (q.shape == IMGENV_SHAPE_CIRCLE) {
            radius = rng.uniform(q.size_range[0], q.size_range[1]);
            o.obs_size[4 * i + 2] = (float)radius;
        } else {
            radius = sqrt(q.size_range[0] * q.size_range[0] + q.size_range[2] * q.size_range[2]);
            for (int k = 0; k < 4; k++) o.obs_size[4 * i + k] = (float)q.size_range[k];
        }
        o.obs_shape[i] = q.shape == IMGENV_SHAPE_CIRCLE ? IMGENV_SHAPE_CIRCLE : IMGENV_SHAPE_RECTANGLE;
        double p[3] = {q.pose[0], q.pose[1], q.pose[2]};
        if (q.pose_type != IMGENV_POSE_FIX) rand_pose(q.pose, q.pose_type == IMGENV_POSE_RANGE_YAW, p);
        spawn_pose4(p[0], p[1], p[2], &o.obs_pose[4 * i]);
        obs_range[3 * i] = p[0];
        obs_range[3 * i + 1] = p[1];
        obs_range[3 * i + 2] = radius;
    }
    auto free_obj = [&](double x, double y, double r) {  // free_check_obj (reset_helper.py:46-55)
        for (int i = 0; i < nob; i++) {
            if (obs_range[3 * i + 2] == 0.0) continue;
            const double dx = x - obs_range[3 * i], dy = y - obs_range[3 * i + 1];
            if (sqrt(dx * dx + dy * dy) <= r + obs_range[3 * i + 2]) return false;
        }
        return true;
    };
    SpawnGrid starts(c.clearance), goals(c.clearance);
    std::vector<double> init((size_t)n * 3), target((size_t)n * 3);
    std::vector<char> has_init(n, 0), has_target(n, 0);
    for (int i = 0; i < n; i++) {
        const imgenv_spawn_agent& a = c.agents[i];
        if (a.begin_type == IMGENV_POSE_FIX || a.begin_type == IMGENV_POSE_RAND_ANGLE) {
            init[3 * i] = a.begin[0];
            init[3 * i + 1] = a.begin[1];
            init[3 * i + 2] = a.begin_type == IMGENV_POSE_FIX ? a.begin[2] : rng.uniform(a.begin[2], a.begin[3]);
            has_init[i] = 1;
            starts.add(init[3 * i], init[3 * i + 1]);
        }
        if (a.target_type == IMGENV_POSE_FIX || a.target_type == IMGENV_POSE_RAND_ANGLE) {
            target[3 * i] = a.target[0];
            target[3 * i + 1] = a.target[1];
            target[3 * i + 2] = a.target_type == IMGENV_POSE_FIX ? a.target[2] : rng.uniform(a.target[2], a.target[3]);
            has_target[i] = 1;
            goals.add(target[3 * i], target[3 * i + 1]);
        }
    }
    const double tmin2 = c.target_min_dist * c.target_min_dist;
    for (int i = 0; i < n; i++) {
        const imgenv_spawn_agent& a = c.agents[i];
        const bool fixed_start = has_init[i], fixed_target = has_target[i];
        double start[3] = {init[3 * i], init[3 * i + 1], init[3 * i + 2]};
        for (int round = 0;; round++) {
            if (round > 1000) return "no admissible start / target pair";
            if (!fixed_start) {
                bool placed = false;
                for (int t = 0; t < 10000 && !placed; t++) {
                    double p[3];
                    rand_pose(a.begin, a.begin_type == IMGENV_POSE_RANGE_YAW, p);
                    if (starts.ok(p[0], p[1], c.clearance) && free_obj(p[0], p[1], a.module_size)) {
                        start[0] = p[0]; start[1] = p[1]; start[2] = p[2];
                        placed = true;
                    }
                }
                if (!placed) return "could not place a start";
            }
            if (fixed_target) break;
            bool found = false;
            double p[3];
            for (int t = 0; t < 51 && !found; t++) {  // goal_fail > 50 re-draws the start (reset_helper.py:296-300)
                if (a.target_type == IMGENV_POSE_RANGE_VIEW) {  // random_view (reset_helper.py:62-82)
                    const double box[4] = {start[0] - 4.0, start[0] + 4.0, start[1] - 4.0, start[1] + 4.0};
                    for (int guard = 0;; guard++) {
                        if (guard > 100000) return "range_view target range never met";
                        rand_pose(box, 0, p);
                        if (fabs(p[0] - start[0]) <= 2.5 && fabs(p[1] - start[1]) <= 2.5) continue;
                        if (a.target[0] <= p[0] && p[0] <= a.target[1] && a.target[2] <= p[1] && p[1] <= a.target[3]) break;
                    }
                } else {
                    rand_pose(a.target, a.target_type == IMGENV_POSE_RANGE_YAW, p);
                }
                const double dx = start[0] - p[0], dy = start[1] - p[1];
                found = dx * dx + dy * dy > tmin2 && goals.ok(p[0], p[1], c.clearance) && free_obj(p[0], p[1], a.module_size);
            }
            if (found) {
                target[3 * i] = p[0]; target[3 * i + 1] = p[1]; target[3 * i + 2] = p[2];
                break;
            }
            if (fixed_start) return "no admissible target for a fixed start";
        }
        init[3 * i] = start[0]; init[3 * i + 1] = start[1]; init[3 * i + 2] = start[2];
        if (!fixed_start) starts.add(start[0], start[1]);
        if (!fixed_target) goals.add(target[3 * i], target[3 * i + 1]);
    }
    o.robot_pose.assign((size_t)nr * 4, 0.0);
    o.robot_goal.assign((size_t)nr * 2, 0.0);
    o.ped_pose.assign((size_t)(np ? np : 1) * 4, 0.0);
    o.ped_goal.assign((size_t)(np ? np : 1) * 2, 0.0);
    o.ped_traj.assign((size_t)(np ? np : 1) * 2 * 3, 0.0);
    o.ped_traj_len.assign(np ? np : 1, 1);
    for (int i = 0; i < nr; i++) {
        spawn_pose4(init[3 * i], init[3 * i + 1], init[3 * i + 2], &o.robot_pose[4 * i]);
        o.robot_goal[2 * i] = target[3 * i];
        o.robot_goal[2 * i + 1] = target[3 * i + 1];
    }
    for (int j = 0; j < np; j++) {
        const int i = nr + j;
        spawn_pose4(init[3 * i], init[3 * i + 1], init[3 * i + 2], &o.ped_pose[4 * j]);
        o.ped_goal[2 * j] = target[3 * i];
        o.ped_goal[2 * j + 1] = target[3 * i + 1];
        o.ped_traj[6 * j] = target[3 * i];  // walk to the target ...
        o.ped_traj[6 * j + 1] = target[3 * i + 1];
        if (c.go_back == 1 || (c.go_back == 2 && rng.unit() > 0.5)) {  // ... and back (reset_helper.py:337-342)
            o.ped_traj[6 * j + 3] = init[3 * i];
            o.ped_traj[6 * j + 4] = init[3 * i + 1];
            o.ped_traj_len[j] = 2;
        }
    }
    imgenv_reset_batch& b = o.batch;
    b.struct_size = (int32_t)sizeof(imgenv_reset_batch);
    b.n_obstacles = nob;
    b.obs_shape = o.obs_shape.data();
    b.obs_size = o.obs_size.data();
    b.obs_pose = o.obs_pose.data();
    b.robot_pose = o.robot_pose.data();
    b.robot_goal = o.robot_goal.data();
    b.ped_pose = o.ped_pose.data();
    b.ped_goal = o.ped_goal.data();
    b.ped_traj_len = o.ped_traj_len.data();
    b.ped_traj = o.ped_traj.data();
    b.ped_traj_cap = 2;
    b.ignore_obstacle = c.ignore_obstacle;
    b.ped_traj_v = nullptr;
    return nullptr;
}
