// spawn_device.h -- NeverStopWrapper (base.py:198-211) for a batch of envs without the host in the loop: the worlds whose robots
// are all done are found, given a fresh placement and reset by kernels alone (imgenv_step_autoreset_device).
//
//   k_spawn_fill   1 workgroup / pool slot   EnvPos.reset (envs/utils/reset_helper.py:104-345) for placement number n = a slot's
//                  next serial: the rules and the random stream of csrc/spawn_host.h (xoshiro256** seeded with seed0 + n), then
//                  the obstacle instances for k_reset_obstacles and RVO2's obstacle list + BSP (RVOSimulator.cpp:130-170,
//                  KdTree.cpp:119-257).  Placements depend on their number alone, so the pool is refilled on a side stream
//                  underneath the step, for the serials the previous step consumed.
//   k_finished_dev the finished worlds in ascending order and their count, in device memory; the k-th of them takes
//                  placement (consumed so far) + k.
//   k_respawn      1 workgroup / finished world: what ImgEnv::_reset (img_env.cpp:162-292) receives in ResetEnv.srv, out of the
//                  world's slot: robots, pedestrians, trajectories, RVO obstacles, obstacle instances; its time limit restarts.
// The launches behind them (map restore, obstacle raster, rasters, views, observations, tails) read the list and its length
// from device memory (DevWorld::act_n_dev) and are sized for every world of the handle: blocks beyond the count leave at once.
//
// Differences to the host path (imgenv_step_autoreset): sin / cos / log / atan2 come from the device's libm, so a placement
// may differ from csrc/spawn_host.h's in the last bit (and, where a rejection test is that close, in what follows);
// imgenv_spawn_slot() reads a placement back for checkers.
#pragma once

#define SPAWN_MAX_AGENTS 256  // robots + pedestrians of one world (the distance tests take them 64 at a time)
#define SPAWN_MAX_OBST 24     // obstacles of one world
#define SPAWN_BSP_CAP 256     // RVO obstacle vertices of one world, splits included
#define SPAWN_LIST_CAP 6144   // arena of the BSP's per-node vertex lists
#define SPAWN_GUARD 200000    // draws before a placement is given up (the host gives up after 2e7: a kernel must not spin that long)
#define SPAWN_FILL_PERIOD 2    // the pool is refilled on every second call; it holds 4 W placements: a refill sees the count of
                              // two steps ago at worst, and up to W worlds can finish on each of the steps until the next one

struct DevSpawnAgent {
    int begin_type, target_type;
    double begin[6], target[6];
    double module_size;
    int begin_multi, n_begin_multi, target_multi, n_target_multi;  // offsets (in boxes of 6 doubles) into SpawnDev::multi
};
struct DevSpawnObstacle {
    int shape, pose_type;
    double size_range[4];
    double pose[6];
};
struct SlotAgent {  // a robot or a pedestrian as ResetEnv.srv carries it
    double x, y, qz, qw, gx, gy;
    double traj[2][3];
    int traj_len, pad;
};
struct SlotObstacle {
    double x, y, qz, qw;
    float size[4];
    int shape, pad;
};
struct SpawnDev {
    int n_robots, n_peds, n_obstacles, go_back, ignore_obstacle, rvo;  // rvo: the world has RVO agents (obstacle polygons + BSP wanted)
    double clearance, target_min_dist, circle0, circle1;
    const DevSpawnAgent* agents;
    const DevSpawnObstacle* obstacles;
    const double* multi;
    int S;                             // pool slots
    unsigned long long seed0;
    unsigned long long* slot_serial;   // [S] placement number a slot holds, ~0 = none
    unsigned long long* consumed;      // [0] placements handed out so far, [1] the same before this step's worlds took theirs
    int* slot_status;                  // [S] 0 fine, else why the placement failed
    SlotAgent* s_agents;               // [S][n_robots + n_peds]
    SlotObstacle* s_obst;              // [S][n_obstacles]
    ObstInst* s_inst;                  // [S][n_obstacles] (world unset)
    RvoObstDev* s_rvo;                 // [S][cap_o]
    RvoNodeDev* s_nodes;               // [S][cap_n]
    int* s_rvo_n;                      // [S][4] obstacles, nodes, root
    double* s_seg;                     // [S][n_obstacles][4] pedscene: the obstacles as PedScene::addObs takes them, a segment pa -> pb each (pedscene.h:24-30)
    int sfm;                           // the worlds run libpedsim crowds
    int cap_o, cap_n;
    int* fin_list;                     // [W] finished worlds of this step, ascending
    int* fin_n;                        // [1]
    ObstInst* inst_out;                // [W][n_obstacles] instances of the worlds being reset, for k_reset_obstacles
    SlotAgent* place_agents;           // [W][n_robots + n_peds] the placement each world currently runs (imgenv_world_placement)
    SlotObstacle* place_obst;          // [W][n_obstacles]
    unsigned long long* place_serial;  // [W] its number
    ObstInst* w_inst;                  // [W][n_obstacles] the obstacles drawn on each world's map ...
    int* w_inst_valid;                 // [W] ... where the device knows them (k_restore_maps_dev puts only their cells back)
    int* world_epoch;                  // [W] (writable alias of DevWorld::world_epoch)
    int* n_obst_w;                     // [W] writable aliases of the per-world RVO table
    int* oroot_w;
    RvoObstDev* w_obst;                // [W][cap_o]
    RvoNodeDev* w_nodes;               // [W][cap_n]
    double* traj;                      // [P][traj_cap][3]
    int* traj_len;                     // [P]
    int traj_cap;
};

struct DevRng {  // xoshiro256** seeded through splitmix64 (csrc/spawn_host.h SpawnRng)
    unsigned long long s[4];
    double gauss_next;
    bool has_gauss;
    __device__ void seed(unsigned long long sd) {
        for (int k = 0; k < 4; k++) {
            unsigned long long z = (sd += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[k] = z ^ (z >> 31);
        }
        has_gauss = false;
        gauss_next = 0.0;
    }
    __device__ static unsigned long long rotl(unsigned long long x, int k) { return (x << k) | (x >> (64 - k)); }
    __device__ unsigned long long next() {
        const unsigned long long r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    __device__ double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    __device__ double uniform(double a, double b) { return a + (b - a) * unit(); }
    __device__ int randint(int lo, int hi) { return lo + (int)(next() % (unsigned long long)(hi - lo + 1)); }
    __device__ double gauss(double mu, double sigma) {
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

struct SpawnScratch {  // LDS of one k_spawn_fill workgroup
    double init[SPAWN_MAX_AGENTS][3], target[SPAWN_MAX_AGENTS][3];
    unsigned char has_i[SPAWN_MAX_AGENTS], has_t[SPAWN_MAX_AGENTS];
    double obs_range[SPAWN_MAX_OBST][3];
    RvoObstDev ob[SPAWN_BSP_CAP];
    RvoNodeDev nodes[SPAWN_BSP_CAP];
    int lists[SPAWN_LIST_CAP];
    int frame[SPAWN_BSP_CAP][8];  // list offset, n, node, stage, L offset, L n, R offset, R n
    float fa[SPAWN_BSP_CAP], fb[SPAWN_BSP_CAP];  // leftOf of a node's vertices against its splitting edge
    int n_ob, n_nodes, top, status;
};

// The whole wavefront runs the placement in lockstep -- every lane draws the same numbers -- so that the distance tests can use
// the lanes: lane q looks at agents q, q + 64, ... / obstacle q (at most 24 of those), one ballot answers.
__device__ inline bool sp_free_of_all(const double (*pose)[3], const unsigned char* has, int n, double x, double y, double d) {
    bool hit = false;  // free_check_robo_ped (reset_helper.py:35-43)
    for (int q = lane_id(); q < n; q += WAVE) {
        if (has[q]) {
            const double dx = x - pose[q][0], dy = y - pose[q][1];
            hit = hit || sqrt(dx * dx + dy * dy) <= d;
        }
    }
    return !__any(hit);
}
__device__ inline bool sp_free_obj(const SpawnScratch& L, int nob, double x, double y, double r) {
    const int i = lane_id();  // free_check_obj (reset_helper.py:46-55)
    bool hit = false;
    if (i < nob && L.obs_range[i][2] != 0.0) {
        const double dx = x - L.obs_range[i][0], dy = y - L.obs_range[i][1];
        hit = sqrt(dx * dx + dy * dy) <= r + L.obs_range[i][2];
    }
    return !__any(hit);
}
__device__ inline bool sp_is_circle(int t) { return t == IMGENV_POSE_RANGE_CIRCLE || t == IMGENV_POSE_RANGE_CIRCLE_FIX; }
__device__ inline bool sp_is_fixed(int t) { return t == IMGENV_POSE_FIX || t == IMGENV_POSE_RAND_ANGLE; }
__device__ inline void sp_rand_pose(DevRng& rng, const double* r, int has_yaw, double* p) {
    p[0] = rng.uniform(r[0], r[1]);
    p[1] = rng.uniform(r[2], r[3]);
    p[2] = has_yaw ? rng.uniform(r[4], r[5]) : rng.uniform(-3.14, 3.14);
}

// csrc/spawn_host.h spawn_world, the 64 lanes of one wavefront in lockstep (uniform control flow; every lane writes the same
// values).  Returns 0 or why the cast could not be placed.
__device__ int dev_spawn_world(const SpawnDev& c, unsigned long long seed, SpawnScratch& L, SlotAgent* out_agents, SlotObstacle* out_obst) {
    DevRng rng;
    rng.seed(seed);
    const int nr = c.n_robots, np = c.n_peds, n = nr + np, nob = c.n_obstacles;
    for (int i = 0; i < nob; i++) {  // obstacles (reset_helper.py:122-165)
        const DevSpawnObstacle& q = c.obstacles[i];
        SlotObstacle& o = out_obst[i];
        double radius;
        o.size[0] = o.size[1] = o.size[2] = o.size[3] = 0.0f;
        if (q.shape == IMGENV_SHAPE_CIRCLE) {
            radius = rng.uniform(q.size_range[0], q.size_range[1]);
            o.size[2] = (float)radius;
        } else {
            radius = sqrt(q.size_range[0] * q.size_range[0] + q.size_range[2] * q.size_range[2]);
            for (int k = 0; k < 4; k++) o.size[k] = (float)q.size_range[k];
        }
        o.shape = q.shape == IMGENV_SHAPE_CIRCLE ? IMGENV_SHAPE_CIRCLE : IMGENV_SHAPE_RECTANGLE;
        double p[3] = {q.pose[0], q.pose[1], q.pose[2]};
        if (q.pose_type != IMGENV_POSE_FIX) sp_rand_pose(rng, q.pose, q.pose_type == IMGENV_POSE_RANGE_YAW, p);
        o.x = p[0];
        o.y = p[1];
        o.qz = sin(p[2] / 2.0);  // quaternion_from_euler(0, 0, yaw) (ros_utils.py:22-23)
        o.qw = cos(p[2] / 2.0);
        L.obs_range[i][0] = p[0];
        L.obs_range[i][1] = p[1];
        L.obs_range[i][2] = radius;
    }
    for (int i = 0; i < n; i++) {
        L.has_i[i] = L.has_t[i] = 0;
        const DevSpawnAgent& a = c.agents[i];
        if (sp_is_fixed(a.begin_type) && !sp_is_fixed(a.target_type)) return 1;  // the reference never leaves its placement loop
    }
    const double circle_range = rng.uniform(c.circle0, c.circle1);  // once per episode (reset_helper.py:202)
    for (int i = 0; i < n; i++) {
        const DevSpawnAgent& a = c.agents[i];
        if (sp_is_fixed(a.begin_type)) {
            L.init[i][0] = a.begin[0];
            L.init[i][1] = a.begin[1];
            L.init[i][2] = a.begin_type == IMGENV_POSE_FIX ? a.begin[2] : rng.uniform(a.begin[2], a.begin[3]);
            L.has_i[i] = 1;
        }
        if (sp_is_fixed(a.target_type)) {
            L.target[i][0] = a.target[0];
            L.target[i][1] = a.target[1];
            L.target[i][2] = a.target_type == IMGENV_POSE_FIX ? a.target[2] : rng.uniform(a.target[2], a.target[3]);
            L.has_t[i] = 1;
        }
    }
    const double tmin2 = c.target_min_dist * c.target_min_dist;
    long guard = 0;
    bool circle_ok = false;
    while (!circle_ok) {
        circle_ok = true;
        for (int i = 0; i < n; i++) {
            if (L.has_i[i] && L.has_t[i]) continue;
            const DevSpawnAgent& a = c.agents[i];
            bool reset_init = true;
            while (reset_init) {
                int goal_fail = 0, circle_fail = 0;
                if (!sp_is_fixed(a.begin_type)) {
                    while (reset_init) {
                        if (++guard > SPAWN_GUARD) return 2;
                        double p[3];
                        if (sp_is_circle(a.begin_type)) {
                            double angle = rng.uniform(-3.14, 3.14);
                            if (a.begin_type == IMGENV_POSE_RANGE_CIRCLE_FIX) angle = -3.14 + (6.28 / n) * i;
                            p[0] = circle_range * cos(angle) + a.begin[0];
                            p[1] = circle_range * sin(angle) + a.begin[1];
                            p[2] = angle + 3.14;
                            p[0] += rng.gauss(0, 0.5);  // random_noise (reset_helper.py:30-32)
                            p[1] += rng.gauss(0, 0.5);
                        } else if (a.begin_type == IMGENV_POSE_RANGE_MULTI) {
                            sp_rand_pose(rng, c.multi + 6 * (size_t)(a.begin_multi + rng.randint(0, a.n_begin_multi - 1)), 1, p);
                        } else {
                            sp_rand_pose(rng, a.begin, a.begin_type == IMGENV_POSE_RANGE_YAW, p);
                        }
                        // (a start that is drawn again keeps its distance to the start it replaces, as in the reference)
                        if (sp_free_of_all(L.init, L.has_i, n, p[0], p[1], c.clearance) && sp_free_obj(L, nob, p[0], p[1], a.module_size)) {
                            L.init[i][0] = p[0]; L.init[i][1] = p[1]; L.init[i][2] = p[2];
                            L.has_i[i] = 1;
                            reset_init = false;
                            break;
                        }
                        if (sp_is_circle(a.begin_type) && ++circle_fail > 50) {
                            circle_ok = false;
                            for (int j = 0; j < n; j++)
                                if (sp_is_circle(c.agents[j].begin_type)) L.has_i[j] = L.has_t[j] = 0;
                        }
                    }
                }
                const double* st = L.init[i];
                if ((a.target_type == IMGENV_POSE_CIRCLE_FIX || a.target_type == IMGENV_POSE_RANGE_CIRCLE_FIX) && L.has_i[i]) {
                    L.target[i][0] = circle_range * cos(st[2]) + a.target[0];
                    L.target[i][1] = circle_range * sin(st[2]) + a.target[1];
                    L.target[i][2] = st[2] - 3.14;
                    L.has_t[i] = 1;
                }
                if (!sp_is_fixed(a.target_type) && a.target_type != IMGENV_POSE_CIRCLE_FIX) {
                    for (;;) {
                        if (++guard > SPAWN_GUARD) return 3;
                        double p[3];
                        if (sp_is_circle(a.target_type)) {
                            p[0] = circle_range * cos(st[2]) + a.target[0];
                            p[1] = circle_range * sin(st[2]) + a.target[1];
                            p[2] = st[2] - 3.14;
                            p[0] += rng.gauss(0, 0.5);
                            p[1] += rng.gauss(0, 0.5);
                        } else if (a.target_type == IMGENV_POSE_RANGE_VIEW) {  // random_view (reset_helper.py:62-82)
                            const double box[4] = {st[0] - 4.0, st[0] + 4.0, st[1] - 4.0, st[1] + 4.0};
                            for (;;) {
                                if (++guard > SPAWN_GUARD) return 4;
                                sp_rand_pose(rng, box, 0, p);
                                if (st[0] - 2.5 <= p[0] && p[0] <= st[0] + 2.5 && st[1] - 2.5 <= p[1] && p[1] <= st[1] + 2.5) continue;
                                if (a.target[0] <= p[0] && p[0] <= a.target[1] && a.target[2] <= p[1] && p[1] <= a.target[3]) break;
                            }
                        } else if (a.target_type == IMGENV_POSE_RANGE_MULTI) {
                            sp_rand_pose(rng, c.multi + 6 * (size_t)(a.target_multi + rng.randint(0, a.n_target_multi - 1)), 1, p);
                        } else {
                            sp_rand_pose(rng, a.target, a.target_type == IMGENV_POSE_RANGE_YAW, p);
                        }
                        const double dx = st[0] - p[0], dy = st[1] - p[1];
                        if (dx * dx + dy * dy > tmin2 && sp_free_of_all(L.target, L.has_t, n, p[0], p[1], c.clearance) &&
                            sp_free_obj(L, nob, p[0], p[1], a.module_size)) {
                            L.target[i][0] = p[0]; L.target[i][1] = p[1]; L.target[i][2] = p[2];
                            L.has_t[i] = 1;
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
    for (int i = 0; i < n; i++) {
        SlotAgent& o = out_agents[i];
        o.x = L.init[i][0];
        o.y = L.init[i][1];
        o.qz = sin(L.init[i][2] / 2.0);
        o.qw = cos(L.init[i][2] / 2.0);
        o.gx = L.target[i][0];
        o.gy = L.target[i][1];
        o.traj[0][0] = L.target[i][0];  // pedestrians walk to the target ...
        o.traj[0][1] = L.target[i][1];
        o.traj[0][2] = 0.0;
        o.traj[1][0] = o.traj[1][1] = o.traj[1][2] = 0.0;
        o.traj_len = 1;
        if (i >= nr && (c.go_back == 1 || (c.go_back == 2 && rng.unit() > 0.5))) {  // ... and back (reset_helper.py:337-342)
            o.traj[1][0] = L.init[i][0];
            o.traj[1][1] = L.init[i][1];
            o.traj_len = 2;
        }
    }
    return 0;
}

// ---- RVO2 obstacles + BSP (host twin: RvoObstacles in host_tables.h), one thread, float32 in the reference's order ----
__device__ inline float sp_det(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }
__device__ inline float sp_left_of(const SpawnScratch& L, int a, int b, int c) {  // leftOf(a, b, c) = det(a - c, b - a)
    return sp_det(L.ob[a].px - L.ob[c].px, L.ob[a].py - L.ob[c].py, L.ob[b].px - L.ob[a].px, L.ob[b].py - L.ob[a].py);
}
__device__ inline bool sp_pair_ge(int a1, int a2, int b1, int b2) { return (a1 > b1) || (a1 == b1 && a2 >= b2); }

// RVOSimulator::addObstacle (RVOSimulator.cpp:130-170)
__device__ bool sp_rvo_add(SpawnScratch& L, const float* xy, int n) {
    if (L.n_ob + n > SPAWN_BSP_CAP) return false;
    const int first = L.n_ob;
    for (int i = 0; i < n; i++) {
        RvoObstDev o;
        o.px = xy[2 * i];
        o.py = xy[2 * i + 1];
        o.next = o.prev = -1;
        const int k = L.n_ob;
        if (i != 0) {
            o.prev = k - 1;
            L.ob[k - 1].next = k;
        }
        const int inext = (i == n - 1 ? 0 : i + 1), iprev = (i == 0 ? n - 1 : i - 1);
        const float dx = xy[2 * inext] - xy[2 * i], dy = xy[2 * inext + 1] - xy[2 * i + 1];
        const float inv = 1.0f / sqrtf(dx * dx + dy * dy);
        o.ux = dx * inv;
        o.uy = dy * inv;
        if (n == 2) {
            o.is_convex = 1;
        } else {
            const float ax = xy[2 * iprev] - xy[2 * inext], ay = xy[2 * iprev + 1] - xy[2 * inext + 1];
            const float bx = xy[2 * i] - xy[2 * iprev], by = xy[2 * i + 1] - xy[2 * iprev + 1];
            o.is_convex = sp_det(ax, ay, bx, by) >= 0.0f;
        }
        L.ob[L.n_ob++] = o;
        if (i == n - 1) {
            L.ob[k].next = first;
            L.ob[first].prev = k;
        }
    }
    return true;
}

// (restates RVO2's KdTree::buildObstacleTree -- Copyright 2008 University of North Carolina at Chapel Hill, Apache License 2.0; see NOTICE)
// KdTree::buildObstacleTreeRecursive (KdTree.cpp:131-257) with an explicit stack; returns the root or -1, status on overflow
__device__ int sp_rvo_build(SpawnScratch& L) {
    const float EPS = 0.00001f;
    L.n_nodes = 0;
    L.top = 0;
    const int n_all = L.n_ob;
    if (n_all > SPAWN_LIST_CAP) { L.status = 10; return -1; }
    for (int i = 0; i < n_all; i++) L.lists[i] = i;
    L.top = n_all;
    int sp = 0, ret = -1;
    L.frame[0][0] = 0; L.frame[0][1] = n_all; L.frame[0][2] = -1; L.frame[0][3] = 0;
    while (sp >= 0) {
        int* F = L.frame[sp];
        if (F[3] == 0) {
            const int n = F[1];
            const int* obs = L.lists + F[0];
            if (n == 0) {
                ret = -1;
                sp--;
                continue;
            }
            if (L.n_nodes >= SPAWN_BSP_CAP) { L.status = 11; return -1; }
            const int node = L.n_nodes++;
            L.nodes[node].obstacle = L.nodes[node].left = L.nodes[node].right = -1;
            // the splitting edge: the one that leaves the fewest on its larger side (KdTree.cpp:147-183).  The reference walks
            // the pairs (i, j) one by one and leaves the inner loop early once an edge cannot win any more; the counts only grow,
            // so counting all of them -- lane j takes vertex j -- picks the same edge
            const int lane = lane_id();
            int optimal = 0, min_left = n, min_right = n;
            for (int i = 0; i < n; i++) {
                int ls = 0, rs = 0;
                const int i1 = obs[i], i2 = L.ob[i1].next;
                for (int j0 = 0; j0 < n; j0 += WAVE) {
                    const int j = j0 + lane;
                    bool left = false, right = false;
                    if (j < n && j != i) {
                        const int j1 = obs[j], j2 = L.ob[j1].next;
                        const float a = sp_left_of(L, i1, i2, j1), b = sp_left_of(L, i1, i2, j2);
                        left = a >= -EPS && b >= -EPS;
                        right = !left && a <= EPS && b <= EPS;
                        if (!left && !right) left = right = true;
                    }
                    ls += __popcll(__ballot(left));
                    rs += __popcll(__ballot(right));
                }
                if (!sp_pair_ge(max(ls, rs), min(ls, rs), max(min_left, min_right), min(min_left, min_right))) {
                    min_left = ls;
                    min_right = rs;
                    optimal = i;
                }
            }
            // the two children's lists: at most n - 1 entries each
            if (L.top + 2 * n > SPAWN_LIST_CAP) { L.status = 12; return -1; }
            int* Ll = L.lists + L.top;
            int* Rl = L.lists + L.top + n;
            int nl = 0, nr2 = 0;
            const int i1 = obs[optimal], i2 = L.ob[i1].next;
            for (int j = lane; j < n; j += WAVE) {  // leftOf of every vertex and its successor against the edge, side by side
                const int j1 = obs[j], j2 = L.ob[j1].next;
                L.fa[j] = sp_left_of(L, i1, i2, j1);
                L.fb[j] = sp_left_of(L, i1, i2, j2);
            }
            __syncthreads();
            for (int j = 0; j < n; j++) {  // the lists in the reference's order; edges that straddle are split (every lane alike)
                if (j == optimal) continue;
                const int j1 = obs[j], j2 = L.ob[j1].next;
                const float a = L.fa[j], b = L.fb[j];
                if (a >= -EPS && b >= -EPS) {
                    Ll[nl++] = j1;
                } else if (a <= EPS && b <= EPS) {
                    Rl[nr2++] = j1;
                } else {
                    if (L.n_ob >= SPAWN_BSP_CAP) { L.status = 13; return -1; }
                    const float ex = L.ob[i2].px - L.ob[i1].px, ey = L.ob[i2].py - L.ob[i1].py;
                    const float t = sp_det(ex, ey, L.ob[j1].px - L.ob[i1].px, L.ob[j1].py - L.ob[i1].py) /
                                    sp_det(ex, ey, L.ob[j1].px - L.ob[j2].px, L.ob[j1].py - L.ob[j2].py);
                    RvoObstDev no;
                    no.px = L.ob[j1].px + t * (L.ob[j2].px - L.ob[j1].px);
                    no.py = L.ob[j1].py + t * (L.ob[j2].py - L.ob[j1].py);
                    no.prev = j1;
                    no.next = j2;
                    no.is_convex = 1;
                    no.ux = L.ob[j1].ux;
                    no.uy = L.ob[j1].uy;
                    const int nn = L.n_ob;
                    __syncthreads();  // (every lane has read the old links)
                    L.ob[nn] = no;
                    L.n_ob = nn + 1;
                    L.ob[j1].next = nn;
                    L.ob[j2].prev = nn;
                    __syncthreads();
                    if (a > 0.0f) {
                        Ll[nl++] = j1;
                        Rl[nr2++] = nn;
                    } else {
                        Rl[nr2++] = j1;
                        Ll[nl++] = nn;
                    }
                }
            }
            __syncthreads();
            F[2] = node;
            F[4] = L.top; F[5] = nl; F[6] = L.top + n; F[7] = nr2;
            L.top += 2 * n;
            L.nodes[node].obstacle = i1;
            F[3] = 1;
            if (sp + 1 >= SPAWN_BSP_CAP) { L.status = 14; return -1; }
            sp++;
            L.frame[sp][0] = F[4]; L.frame[sp][1] = F[5]; L.frame[sp][2] = -1; L.frame[sp][3] = 0;
        } else if (F[3] == 1) {
            L.nodes[F[2]].left = ret;
            F[3] = 2;
            sp++;
            L.frame[sp][0] = F[6]; L.frame[sp][1] = F[7]; L.frame[sp][2] = -1; L.frame[sp][3] = 0;
        } else {
            L.nodes[F[2]].right = ret;
            ret = F[2];
            sp--;
        }
    }
    return ret;
}

// One pool slot: placement number n = the smallest n >= consumed with n % S == slot, unless the slot holds it already.
// "consumed" is the count BEFORE the worlds of the last completed step took theirs (consumed[1]): this kernel runs beside a
// step on its own stream, and with the up-to-date count it could recycle a slot whose placement that very step's k_respawn has
// been handed (k_finished_dev advances consumed[0] before k_respawn reads the slots) -- one step's worth of placements is
// redrawn into slots that are not read any more, which the pool's size allows for (SPAWN_FILL_PERIOD).
__global__ __launch_bounds__(WAVE) void k_spawn_fill(SpawnDev c) {
    __shared__ SpawnScratch L;
    const int s = blockIdx.x;
    const unsigned long long done = c.consumed[1];
    const unsigned long long S = (unsigned long long)c.S;
    unsigned long long n = done - done % S + (unsigned long long)s;
    if (n < done) n += S;
    if (c.slot_serial[s] == n) return;
    // one wavefront in lockstep: rejection sampling and the BSP's partition are sequential, their distance tests and pair
    // evaluations use the lanes (every lane computes and writes the same values otherwise)
    const int na = c.n_robots + c.n_peds;
    SlotAgent* ag = c.s_agents + (size_t)s * na;
    SlotObstacle* ob = c.s_obst + (size_t)s * (c.n_obstacles > 0 ? c.n_obstacles : 1);
    L.status = dev_spawn_world(c, c.seed0 + n, L, ag, ob);
    L.n_ob = 0;
    L.n_nodes = 0;
    int root = -1;
    for (int q = 0; q < c.n_obstacles && L.status == 0; q++) {
        // the obstacle as ImgEnv::_reset places it (img_env.cpp:169-207): yaw out of the quaternion, footprint lattice bounds,
        // RVOScene::addObs polygon through the two corners (rvoscene.h:19-26, agent.cpp:626-651)
        const SlotObstacle& o = ob[q];
        double sizes[4];
        for (int j = 0; j < 4; j++) sizes[j] = (double)o.size[j];
        Tf2 rq;
        tf_set_rotation_zw(rq, o.qz, o.qw);
        const double yaw = cr_atan2(rq.m10 / 1.0, rq.m00 / 1.0);  // tf_yaw_from_quaternion_zw with a correctly rounded atan2
        ObstInst oi;
        oi.x = o.x; oi.y = o.y; oi.sh = sin(yaw * 0.5); oi.ch = cos(yaw * 0.5);
        oi.world = -1;
        oi.shape = o.shape;
        oi.cx = sizes[0]; oi.cy = sizes[1]; oi.r = sizes[2];
        if (o.shape == IMGENV_SHAPE_CIRCLE) {
            const int bb = (int)ceil(sizes[2] / 0.01);
            oi.m0 = oi.n0 = -bb;
            oi.m1 = oi.n1 = bb;
        } else {
            oi.m0 = (int)floor(sizes[0] / 0.01); oi.m1 = (int)ceil(sizes[1] / 0.01);
            oi.n0 = (int)floor(sizes[2] / 0.01); oi.n1 = (int)ceil(sizes[3] / 0.01);
        }
        c.s_inst[(size_t)s * c.n_obstacles + q] = oi;
        if ((c.rvo || c.sfm) && !c.ignore_obstacle) {
            const Tf2 bw = tf_from_pose_sc(o.x, o.y, oi.sh, oi.ch);
            double pax, pay, pbx, pby;
            if (o.shape == IMGENV_SHAPE_CIRCLE) {  // Agent::get_corners
                tf_apply(bw, sizes[0] - sizes[2], sizes[1] - sizes[2], pax, pay);
                tf_apply(bw, sizes[0] + sizes[2], sizes[1] + sizes[2], pbx, pby);
            } else {
                tf_apply(bw, sizes[0], sizes[2], pax, pay);
                tf_apply(bw, sizes[1], sizes[3], pbx, pby);
            }
            if (c.sfm) {  // PedScene::addObs: the segment pa -> pb (pedscene.h:24-30)
                double* g = c.s_seg + ((size_t)s * c.n_obstacles + q) * 4;
                g[0] = pax; g[1] = pay; g[2] = pbx; g[3] = pby;
            }
            if (c.rvo) {
                const float v[8] = {(float)pax, (float)pay, (float)pax, (float)pby, (float)pbx, (float)pby, (float)pbx, (float)pay};
                if (!sp_rvo_add(L, v, 4)) L.status = 15;
            }
        }
    }
    if (L.status == 0 && L.n_ob > 0) root = sp_rvo_build(L);
    if (L.status == 0 && (L.n_ob > c.cap_o || L.n_nodes > c.cap_n)) L.status = 16;
    __syncthreads();
    if (L.status == 0) {
        for (int q = lane_id(); q < L.n_ob; q += WAVE) c.s_rvo[(size_t)s * c.cap_o + q] = L.ob[q];
        for (int q = lane_id(); q < L.n_nodes; q += WAVE) c.s_nodes[(size_t)s * c.cap_n + q] = L.nodes[q];
    }
    c.s_rvo_n[4 * s] = L.n_ob;
    c.s_rvo_n[4 * s + 1] = L.n_nodes;
    c.s_rvo_n[4 * s + 2] = root;
    c.slot_status[s] = L.status;
    __threadfence();
    c.slot_serial[s] = n;
}

// The worlds whose robots are all done, ascending, and their count, in device memory (and, for a host that wants to know, in the
// page-locked list k_finished writes); the placements they take are numbered from consumed[1] = the count so far.
__global__ __launch_bounds__(1024) void k_finished_dev(DevWorld w, SpawnDev c) {
    __shared__ int wave_n[16];
    __shared__ int base_sh;
    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int k0 = 0; k0 < w.W; k0 += 1024) {  // uniform trip count
        const int k = k0 + tid;
        bool all_done = k < w.W;
        if (all_done)
            for (int q = 0; q < w.Rw && all_done; q++) all_done = w.dones[(size_t)k * w.Rw + q] != 0;
        if (k < w.W)
            for (int q = 0; q < w.Rw; q++) w.step_all_down[(size_t)k * w.Rw + q] = all_done ? 1 : 0;
        const unsigned long long mask = __ballot(all_done);
        if (lane == 0) wave_n[wv] = __popcll(mask);
        __syncthreads();
        int before = base_sh;
        for (int q = 0; q < wv; q++) before += wave_n[q];
        if (all_done) {
            const int pos = before + __popcll(mask & ((1ull << lane) - 1ull));
            c.fin_list[pos] = k;
            w.finished[1 + pos] = k;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int q = 0; q < 16; q++) tot += wave_n[q];
            base_sh += tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        *c.fin_n = base_sh;
        w.finished[0] = base_sh;
        c.consumed[1] = c.consumed[0];
        c.consumed[0] += (unsigned long long)base_sh;
    }
}

// The q-th finished world receives its placement: what stage_world + k_reset_apply do for a host-made batch.
__global__ __launch_bounds__(WAVE) void k_respawn(DevWorld w, SpawnDev c, int elapsed) {
    const int q = blockIdx.x;
    if (q >= *c.fin_n) return;
    const int world = c.fin_list[q], tid = threadIdx.x;
    const unsigned long long n = c.consumed[1] + (unsigned long long)q;
    const int s = (int)(n % (unsigned long long)c.S);
    if (c.slot_serial[s] != n || c.slot_status[s] != 0) {  // the pool did not hold this placement, or it could not be placed
        if (tid == 0) w.err[2] = c.slot_serial[s] != n ? 100 : c.slot_status[s];
        // The error surfaces at the next API call and the handle is then to be reset by its owner.  Until then the world keeps its
        // poses and its map (no obstacle instances to draw, no map to restore: world = -1 below); it does stay on the finished
        // list, so the chain's rasters / views / solve still treat it as freshly reset (robot agents at rest, every view cell
        // rewritten) -- its outputs between here and the error are those of a reset of the OLD placement, not of a running episode
        for (int e = tid; e < c.n_obstacles; e += WAVE) c.inst_out[(size_t)q * c.n_obstacles + e].world = -1;
        return;
    }
    const int nr = c.n_robots, np = c.n_peds, na = nr + np;
    const SlotAgent* ag = c.s_agents + (size_t)s * na;
    if (tid == 0) c.place_serial[world] = n;
    for (int e = tid; e < c.n_obstacles; e += WAVE) c.place_obst[(size_t)world * c.n_obstacles + e] = c.s_obst[(size_t)s * c.n_obstacles + e];
    for (int a = tid; a < na; a += WAVE) {
        const SlotAgent o = ag[a];
        c.place_agents[(size_t)world * na + a] = o;
        Tf2 rq;
        tf_set_rotation_zw(rq, o.qz, o.qw);
        const double yaw = cr_atan2(rq.m10 / 1.0, rq.m00 / 1.0);  // tf::Matrix3x3(q).getRPY (img_env.cpp:180-183)
        if (a < nr) {  // robots (img_env.cpp:252-282): init_pose, set_goal; Agent::vx, vy persist across resets
            const int i = world * w.Rw + a, l = i - w.r0;
            double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
            r[0] = o.x;
            r[1] = o.y;
            r[2] = yaw;
            r[5] = sin(yaw * 0.5);
            r[6] = cos(yaw * 0.5);
            w.l0v[l] = 0;
            w.l0w[l] = 0;
            w.gx[l] = o.gx;
            w.gy[l] = o.gy;
            w.world_target[l] = tf_inverse(tf_from_pose(o.gx, o.gy, yaw));  // set_goal (agent.cpp:144-154)
            w.is_coll[l] = 0;
            w.is_arr[l] = 0;
        } else {       // pedestrians (img_env.cpp:220-250)
            const int j = world * w.Pw + (a - nr);
            w.ppx[j] = o.x;
            w.ppy[j] = o.y;
            w.pyaw[j] = yaw;
            w.ptraj_idx[j] = 0;
            if (w.NA > 0) {  // setPedPos (rvoscene.h:32-34); the agent's velocity persists
                w.apx[j] = (float)o.x;
                w.apy[j] = (float)o.y;
            }
            if (w.scene == IMGENV_SCENE_PEDSIM) {  // PedScene::setPedPos + setWayPoint (pedscene.h:34-46): position (velocity persists), then
                const size_t m = (size_t)world * w.sfm.n + (size_t)(a - nr);  // the waypoint deque [goal r = 1, trajectory r = z], destination its front
                double* p = w.sfm.p + 3 * m;
                p[0] = o.x;
                p[1] = o.y;
                p[2] = 0.0;
                double *wx = w.sfm.wpx + m * SFM_MAX_WP, *wy = w.sfm.wpy + m * SFM_MAX_WP, *wr = w.sfm.wpr + m * SFM_MAX_WP;
                int* dq = w.sfm.dq + m * SFM_MAX_WP;
                wx[0] = o.gx;
                wy[0] = o.gy;
                wr[0] = 1.0;
                int nw = 1;
                for (int e = 0; e < o.traj_len && nw < SFM_MAX_WP; e++, nw++) {
                    wx[nw] = o.traj[e][0];
                    wy[nw] = o.traj[e][1];
                    wr[nw] = o.traj[e][2];
                }
                for (int e = 0; e < SFM_MAX_WP; e++) {
                    dq[e] = e < nw ? e : 0;
                    if (e >= nw) wx[e] = wy[e] = wr[e] = 0.0;
                }
                w.sfm.dq_n[m] = nw;
                w.sfm.dest[m] = 0;
                w.sfm.last[m] = -1;
            }
            w.ped_state[4 * j] = o.x;
            w.ped_state[4 * j + 1] = o.y;
            w.ped_state[4 * j + 2] = w.pvx[j];
            w.ped_state[4 * j + 3] = w.pvy[j];
            double* tr = c.traj + (size_t)j * c.traj_cap * 3;
            for (int e = 0; e < 2 && e < c.traj_cap; e++) {
                tr[3 * e] = o.traj[e][0];
                tr[3 * e + 1] = o.traj[e][1];
                tr[3 * e + 2] = o.traj[e][2];
            }
            c.traj_len[j] = o.traj_len;
        }
    }
    for (int e = tid; e < c.n_obstacles; e += WAVE) {
        ObstInst oi = c.s_inst[(size_t)s * c.n_obstacles + e];
        oi.world = world;
        c.inst_out[(size_t)q * c.n_obstacles + e] = oi;
    }
    if (c.sfm) {  // the crowd's obstacle segments
        const int nseg = c.ignore_obstacle ? 0 : c.n_obstacles;
        double* dst = w.sfm.obs + (size_t)world * w.sfm.cap_obs * 4;
        for (int e = tid; e < nseg * 4; e += WAVE) dst[e] = c.s_seg[(size_t)s * c.n_obstacles * 4 + e];
        if (tid == 0) const_cast<int*>(w.sfm.n_obs_w)[world] = nseg;
    }
    const int n_ob = c.s_rvo_n[4 * s], n_nodes = c.s_rvo_n[4 * s + 1];
    for (int e = tid; e < n_ob; e += WAVE) c.w_obst[(size_t)world * c.cap_o + e] = c.s_rvo[(size_t)s * c.cap_o + e];
    for (int e = tid; e < n_nodes; e += WAVE) c.w_nodes[(size_t)world * c.cap_n + e] = c.s_nodes[(size_t)s * c.cap_n + e];
    if (tid == 0) {
        c.n_obst_w[world] = n_ob;
        c.oroot_w[world] = c.s_rvo_n[4 * s + 2];
        c.world_epoch[world] = elapsed;  // its TimeLimitWrapper starts over
    }
}

// obs_map_ of every finished world starts from the static map again (img_env.cpp:166-168); STAMP mode: the class layer's base
// classes with it.  The only cells that ever differ from the static map are the ones the world's obstacles were drawn on
// (Agent::draw(obs_map, 0) at reset), so a world whose current obstacles are known (w_inst: every world the device has reset
// before) gets those cells back -- a few thousand -- instead of the whole map (0.5 MB + 2 MB of class words at 733 x 733 cells);
// stamps of the old episode stay on the class layer and expire with their step tags as always.  map_blocks blocks per world,
// sized for every world of the handle.
template <bool POW2>
__global__ __launch_bounds__(256) void k_restore_maps_dev(DevWorld w, SpawnDev c, const uint8_t* __restrict__ static_map, int stamp, int map_blocks) {
    // (the grid covers a guess of the number of finished worlds, not every world of the handle: a block whose world does not
    // exist still costs the dispatcher its nanosecond -- 65 536 of them were most of this kernel's 30 us at 2048 envs)
    const int q0 = blockIdx.x / map_blocks, part = blockIdx.x - q0 * map_blocks, q_stride = (int)gridDim.x / map_blocks;
    for (int q = q0; q < *c.fin_n; q += q_stride) {
        const int world = c.fin_list[q];
        if (c.place_serial[world] != c.consumed[1] + (unsigned long long)q) continue;  // k_respawn could not place it: the old episode's map stays
        uint8_t* map = const_cast<uint8_t*>(w.obs_map) + (size_t)world * w.Gs;
        uint32_t* cell = w.cell + (size_t)world * w.Gs;
        if (c.w_inst_valid[world]) {
            const double resolution = 0.01;
            for (int e = 0; e < c.n_obstacles; e++) {
                const ObstInst o = c.w_inst[(size_t)world * c.n_obstacles + e];
                const Tf2 bw = tf_from_pose_sc(o.x, o.y, o.sh, o.ch);
                const int nn = o.n1 - o.n0 + 1, total = (o.m1 - o.m0 + 1) * nn;
                const bool circle = o.shape == IMGENV_SHAPE_CIRCLE;
                for (int s = part * 256 + (int)threadIdx.x; s < total; s += map_blocks * 256) {  // the footprint samples of k_reset_obstacles
                    const int m = o.m0 + s / nn, n = o.n0 + s % nn;
                    double px = m * resolution, py = n * resolution;
                    if (circle) {
                        if (!(sqrt(m * resolution * m * resolution + n * resolution * n * resolution) <= o.r)) continue;
                        px = px + o.cx;
                        py = py + o.cy;
                    }
                    double wx, wy;
                    tf_apply(bw, px, py, wx, wy);
                    int gm, gn;
                    w2m_pair<POW2>(wx, wy, w.res, w.inv_res, gm, gn);
                    if (gm >= 0 && gm < w.Hg && gn >= 0 && gn < w.Wg) {
                        const size_t at = (size_t)gm * w.Wg + gn;
                        const uint32_t v = static_map[at];
                        map[at] = (uint8_t)v;
                        if (stamp) cell[at] = (v <= 2 ? v : (v < 250 ? CLS_LOW : CLS_HIGH)) | (stamp == 2 ? cell[at] & ~7u : 0u);  // (SUM mode: the counts stay)
                        if (w.crop_map) w.crop_map[(size_t)world * w.crop_ws + crop_tiled(w, (uint32_t)gm, (uint32_t)gn)] = v >= 250 ? 128 : 0;  // (a stamp left on it has expired)
                    }
                }
            }
            continue;
        }
        if (w.crop_map) {
            const uint4* src = (const uint4*)w.static_crop;
            uint4* cm = (uint4*)(w.crop_map + (size_t)world * w.crop_ws);
            for (size_t e = (size_t)part * blockDim.x + threadIdx.x; e < w.crop_ws / 16; e += (size_t)map_blocks * blockDim.x) cm[e] = src[e];
        }
        const size_t n16 = ((size_t)w.Hg * w.Wg + 15) / 16;
        uint4* dst = (uint4*)map;
        uint4* cls = (uint4*)cell;
        for (size_t e = (size_t)part * blockDim.x + threadIdx.x; e < n16; e += (size_t)map_blocks * blockDim.x) {
            const uint4 v = ((const uint4*)static_map)[e];
            dst[e] = v;
            if (stamp) {
                const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t c4[4];
                    const uint4 old = stamp == 2 ? cls[4 * e + k] : make_uint4(0, 0, 0, 0);
                    const uint32_t keep[4] = {old.x & ~7u, old.y & ~7u, old.z & ~7u, old.w & ~7u};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t o = (wd[k] >> (8 * j)) & 0xFFu;
                        c4[j] = (o <= 2 ? o : (o < 250 ? CLS_LOW : CLS_HIGH)) | keep[j];
                    }
                    cls[4 * e + k] = make_uint4(c4[0], c4[1], c4[2], c4[3]);
                }
            }
        }
    }
}

