// spawn_host.h -- random placement of obstacles, robots and pedestrians for an episode, on the host.
//
// The reference's EnvPos.reset (envs/utils/reset_helper.py:104-345) with the pose types its configs use: `fix`, `rand_angle`,
// `range` (4 or 6 numbers), `range_multi`, `range_circle` / `range_circle_fix` starts; `range`, `range_multi`, `range_view`,
// `range_circle` and `circle_fix` targets.  Control flow and rejection rules are the reference's, loop for loop (the Python
// mirror img_env_amd/spawn.py is pinned bit for bit on episodes the reference's own EnvPos placed; this file follows the same
// structure):
//   * starts keep > clearance (1.0 m, free_check_robo_ped, reset_helper.py:35-43) to every other start -- and to the start they
//     replace, when a start is drawn again -- and > module size + obstacle radius to every obstacle (free_check_obj, 46-55);
//   * targets keep > target_min_dist to their own start, > clearance to every other target, and clear the obstacles; a
//     `range_view` target lies in the 4 m box around its start but outside the 2.5 m box (random_view, 62-82); after 50 failed
//     targets the start is drawn again (296-300);
//   * circle starts sit on a circle of this episode's radius (drawn once from circle_ranges) around their centre, at a random
//     angle -- or, `_fix`, at the agent's own share of the circle -- plus Gaussian noise (sigma 0.5 m), facing the centre; after 50
//     failed circle starts every circle agent is cleared and the pass repeated (249-255); circle targets lie opposite the start;
//   * pedestrians walk to their target and, with go_back, back to their start (337-342).
// What is NOT the reference's: the random stream (a splitmix64 / xoshiro256** generator seeded per world instead of Python's
// Mersenne Twister), and the neighbour search (a hash grid instead of O(n^2) list scans).
//
// Why it is native: a world of a multi-world handle is reset whenever its episode ends -- dozens of worlds on every step --
// and the Python spawn costs 40-170 us per small world, twenty times the device's whole step.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <unordered_map>
#include <vector>

#include "../../include/imgenv.h"

struct SpawnRng {  // xoshiro256** seeded through splitmix64
    uint64_t s[4];
    double gauss_next = 0.0;
    bool has_gauss = false;
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
    int randint(int lo, int hi) { return lo + (int)(next() % (uint64_t)(hi - lo + 1)); }  // inclusive
    double gauss(double mu, double sigma) {  // a pair per two calls (the scheme of Python's random.gauss)
        double z;
        if (has_gauss) {
            z = gauss_next;
            has_gauss = false;
        } else {
            const double x2pi = unit() * 6.283185307179586, g2rad = sqrt(-2.0 * log(1.0 - unit()));
            z = cos(x2pi) * g2rad;
            gauss_next = sin(x2pi) * g2rad;
            has_gauss = true;
        }
        return mu + z * sigma;
    }
};

struct SpawnSet {  // poses of n agents with holes; "is anybody within d of (x, y)?" through a hash of cell -> agents
    double cell;
    std::vector<double> pose;  // [n][3]
    std::vector<char> has;
    std::unordered_map<int64_t, std::vector<int>> buckets;
    SpawnSet(int n, double c) : cell(c > 1e-6 ? c : 1e-6), pose((size_t)n * 3, 0.0), has(n, 0) {}
    int64_t key(double x, double y) const { return (int64_t)floor(x / cell) * 0x100000000ll + (int64_t)floor(y / cell); }
    void clear(int i) {
        if (!has[i]) return;
        std::vector<int>& b = buckets[key(pose[3 * i], pose[3 * i + 1])];
        b.erase(std::remove(b.begin(), b.end(), i), b.end());
        has[i] = 0;
    }
    void set(int i, const double* p) {
        clear(i);
        pose[3 * i] = p[0]; pose[3 * i + 1] = p[1]; pose[3 * i + 2] = p[2];
        has[i] = 1;
        buckets[key(p[0], p[1])].push_back(i);
    }
    bool free_of_all(double x, double y, double d) const {  // free_check_robo_ped (reset_helper.py:35-43)
        const int64_t kx = (int64_t)floor(x / cell), ky = (int64_t)floor(y / cell), r = (int64_t)ceil(d / cell);
        for (int64_t i = kx - r; i <= kx + r; i++)
            for (int64_t j = ky - r; j <= ky + r; j++) {
                auto it = buckets.find(i * 0x100000000ll + j);
                if (it == buckets.end()) continue;
                for (int q : it->second) {
                    const double dx = x - pose[3 * q], dy = y - pose[3 * q + 1];
                    if (sqrt(dx * dx + dy * dy) <= d) return false;
                }
            }
        return true;
    }
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
    auto is_circle = [](int t) { return t == IMGENV_POSE_RANGE_CIRCLE || t == IMGENV_POSE_RANGE_CIRCLE_FIX; };
    auto is_fixed = [](int t) { return t == IMGENV_POSE_FIX || t == IMGENV_POSE_RAND_ANGLE; };
    for (int i = 0; i < n; i++) {
        const imgenv_spawn_agent& a = c.agents[i];
        if (is_fixed(a.begin_type) && !is_fixed(a.target_type))
            return "a fixed start needs a fixed target (the reference never leaves its placement loop otherwise, reset_helper.py:218-300)";
        if (a.begin_type == IMGENV_POSE_RANGE_MULTI && (a.n_begin_multi < 1 || !a.begin_multi)) return "range_multi start without ranges";
        if (a.target_type == IMGENV_POSE_RANGE_MULTI && (a.n_target_multi < 1 || !a.target_multi)) return "range_multi target without ranges";
    }
    SpawnSet init(n, c.clearance), target(n, c.clearance);
    const double circle_range = rng.uniform(c.circle_ranges[0], c.circle_ranges[1]);  // once per episode (reset_helper.py:202)
    for (int i = 0; i < n; i++) {
        const imgenv_spawn_agent& a = c.agents[i];
        if (is_fixed(a.begin_type)) {
            const double p[3] = {a.begin[0], a.begin[1], a.begin_type == IMGENV_POSE_FIX ? a.begin[2] : rng.uniform(a.begin[2], a.begin[3])};
            init.set(i, p);
        }
        if (is_fixed(a.target_type)) {
            const double p[3] = {a.target[0], a.target[1], a.target_type == IMGENV_POSE_FIX ? a.target[2] : rng.uniform(a.target[2], a.target[3])};
            target.set(i, p);
        }
    }
    auto on_circle = [&](const double* centre, double angle, double yaw, double* p) {
        p[0] = circle_range * cos(angle) + centre[0];
        p[1] = circle_range * sin(angle) + centre[1];
        p[2] = yaw;
        p[0] += rng.gauss(0, 0.5);  // random_noise (reset_helper.py:30-32)
        p[1] += rng.gauss(0, 0.5);
    };
    auto from_range = [&](int type, const double* r6, const double* multi, int n_multi, double* p) {
        if (type == IMGENV_POSE_RANGE_MULTI) {  // one of several boxes ([n][6]: a 4-number box carries yaw +-3.14)
            r6 = multi + 6 * (size_t)rng.randint(0, n_multi - 1);
            rand_pose(r6, 1, p);
        } else {
            rand_pose(r6, type == IMGENV_POSE_RANGE_YAW, p);
        }
    };
    const double tmin2 = c.target_min_dist * c.target_min_dist;
    long guard = 0;
    bool circle_ok = false;
    while (!circle_ok) {
        circle_ok = true;
        for (int i = 0; i < n; i++) {
            if (init.has[i] && target.has[i]) continue;
            const imgenv_spawn_agent& a = c.agents[i];
            bool reset_init = true;
            while (reset_init) {
                int goal_fail = 0, circle_fail = 0;
                if (!is_fixed(a.begin_type)) {
                    while (reset_init) {
                        if (++guard > 20000000) return "no admissible placement (starts)";
                        double p[3];
                        if (is_circle(a.begin_type)) {
                            double angle = rng.uniform(-3.14, 3.14);
                            if (a.begin_type == IMGENV_POSE_RANGE_CIRCLE_FIX) angle = -3.14 + (6.28 / n) * i;
                            on_circle(a.begin, angle, angle + 3.14, p);
                        } else {
                            from_range(a.begin_type, a.begin, a.begin_multi, a.n_begin_multi, p);
                        }
                        if (init.free_of_all(p[0], p[1], c.clearance) && free_obj(p[0], p[1], a.module_size)) {
                            init.set(i, p);
                            reset_init = false;
                            break;
                        }
                        if (is_circle(a.begin_type) && ++circle_fail > 50) {
                            circle_ok = false;
                            for (int j = 0; j < n; j++)
                                if (is_circle(c.agents[j].begin_type)) {
                                    init.clear(j);
                                    target.clear(j);
                                }
                        }
                    }
                }
                const double* st = &init.pose[3 * (size_t)i];
                if ((a.target_type == IMGENV_POSE_CIRCLE_FIX || a.target_type == IMGENV_POSE_RANGE_CIRCLE_FIX) && init.has[i]) {
                    const double p[3] = {circle_range * cos(st[2]) + a.target[0], circle_range * sin(st[2]) + a.target[1], st[2] - 3.14};
                    target.set(i, p);
                }
                if (!is_fixed(a.target_type) && a.target_type != IMGENV_POSE_CIRCLE_FIX) {
                    for (;;) {
                        if (++guard > 20000000) return "no admissible placement (targets)";
                        double p[3];
                        if (is_circle(a.target_type)) {
                            on_circle(a.target, st[2], st[2] - 3.14, p);
                        } else if (a.target_type == IMGENV_POSE_RANGE_VIEW) {  // random_view (reset_helper.py:62-82)
                            const double box[4] = {st[0] - 4.0, st[0] + 4.0, st[1] - 4.0, st[1] + 4.0};
                            for (;;) {
                                if (++guard > 20000000) return "range_view target range never met";
                                rand_pose(box, 0, p);
                                if (st[0] - 2.5 <= p[0] && p[0] <= st[0] + 2.5 && st[1] - 2.5 <= p[1] && p[1] <= st[1] + 2.5) continue;
                                if (a.target[0] <= p[0] && p[0] <= a.target[1] && a.target[2] <= p[1] && p[1] <= a.target[3]) break;
                            }
                        } else {
                            from_range(a.target_type, a.target, a.target_multi, a.n_target_multi, p);
                        }
                        const double dx = st[0] - p[0], dy = st[1] - p[1];
                        if (dx * dx + dy * dy > tmin2 && target.free_of_all(p[0], p[1], c.clearance) && free_obj(p[0], p[1], a.module_size)) {
                            target.set(i, p);
                            break;
                        }
                        if (++goal_fail > 50) {  // draw the start again (reset_helper.py:296-300)
                            reset_init = true;
                            break;
                        }
                    }
                }
            }
        }
    }
    const std::vector<double>& initp = init.pose;
    const std::vector<double>& targetp = target.pose;
    o.robot_pose.assign((size_t)nr * 4, 0.0);
    o.robot_goal.assign((size_t)nr * 2, 0.0);
    o.ped_pose.assign((size_t)(np ? np : 1) * 4, 0.0);
    o.ped_goal.assign((size_t)(np ? np : 1) * 2, 0.0);
    o.ped_traj.assign((size_t)(np ? np : 1) * 2 * 3, 0.0);
    o.ped_traj_len.assign(np ? np : 1, 1);
    for (int i = 0; i < nr; i++) {
        spawn_pose4(initp[3 * i], initp[3 * i + 1], initp[3 * i + 2], &o.robot_pose[4 * i]);
        o.robot_goal[2 * i] = targetp[3 * i];
        o.robot_goal[2 * i + 1] = targetp[3 * i + 1];
    }
    for (int j = 0; j < np; j++) {
        const int i = nr + j;
        spawn_pose4(initp[3 * i], initp[3 * i + 1], initp[3 * i + 2], &o.ped_pose[4 * j]);
        o.ped_goal[2 * j] = targetp[3 * i];
        o.ped_goal[2 * j + 1] = targetp[3 * i + 1];
        o.ped_traj[6 * j] = targetp[3 * i];  // walk to the target ...
        o.ped_traj[6 * j + 1] = targetp[3 * i + 1];
        if (c.go_back == 1 || (c.go_back == 2 && rng.unit() > 0.5)) {  // ... and back (reset_helper.py:337-342)
            o.ped_traj[6 * j + 3] = initp[3 * i];
            o.ped_traj[6 * j + 4] = initp[3 * i + 1];
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
