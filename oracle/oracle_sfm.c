/*
 * THIRD-PARTY NOTICE.  The force model, its constants, the waypoint logic and the quadtree's insert / erase / split rules in this
 * file restate libpedsim ("pedsim - A microscopic pedestrian simulation system.  Copyright (c) by Christian Gloor",
 * http://pedsim.silmaril.org/, distributed under the GNU General Public License) as vendored by the reference in
 * src/3rdparty/pedsimros -- formulas and rule order are necessarily its own, since results must match it.
 * Changes made here: restated in plain C in one translation unit, agents in arrays instead of heap objects; used as a test oracle only.  See NOTICE at the repository root.
 */
/*
 * oracle_sfm.c -- TEST INFRASTRUCTURE (oracle), never linked into the product library.
 *
 * CPU restatement (plain C, float64) of the libpedsim social-force model the reference vendors in
 * src/3rdparty/pedsimros, driven the way PedScene drives it (src/img_env/src/pedscene.h:17-91):
 *   Tagent::Tagent (random vmax)      src/ped_agent.cpp:24-58
 *   Tagent::desiredForce              src/ped_agent.cpp:236-306   (+ Twaypoint::getForce src/ped_waypoint.cpp:81-134)
 *   Tagent::socialForce               src/ped_agent.cpp:316-404
 *   Tagent::obstacleForce             src/ped_agent.cpp:411-429   (+ Tobstacle::closestPoint src/ped_obstacle.cpp:90-113)
 *   Tagent::lookaheadForce            src/ped_agent.cpp:439-480
 *   Tagent::computeForces / move      src/ped_agent.cpp:498-571   (+ Tvector::lineIntersection src/ped_vector.cpp:264-280)
 *   Tscene::moveAgents                src/ped_scene.cpp:167-182
 *   Tvector (3-D: z matters)          src/ped_vector.cpp
 * Pinned against the reference's own sources compiled unmodified (oracle/_ref/libpedsim_ref.so,
 * tests/test_oracle_sfm_ref.py).
 *
 * Neighbour sets.  Tagent::computeForces asks the scene's quadtree for the agents within 20 m
 * (ped_agent.cpp:499-500, ped_scene.cpp:217-252).  PedScene builds that tree as Tscene(0,10,10,10), i.e. over
 * x in [0,10], y in [10,20] (pedscene.h:18, ped_tree.cpp:18-29); agents enter it at addPed / addRobot with a
 * rand() position (peds, pedscene.h:60-61) or (0,0,0) (robots), leaves split above 8 agents
 * (ped_tree.cpp:65-96), and Tagent::move re-buckets through Ttree::moveAgent (ped_tree.cpp:131-137), which
 * re-inserts an agent that left its leaf's rectangle FROM THE ROOT AND THEN ERASES IT FROM THE OLD LEAF.  When
 * the re-insertion lands in the same leaf the agent vanishes from the tree: on maps whose y stays below 10 m
 * every agent is gone after its second move, so social and look-ahead forces act for two steps only.  That is
 * the reference's behaviour and it is restated literally here (tree, treehash, split, moveAgent, query).
 * The force sums run over a std::set ordered by heap address in the reference; here they run in agent order
 * (peds, then robots), which is the allocation order.
 *
 * rand().  glibc's TYPE_3 additive-feedback generator with the default seed 1 (stdlib/random_r.c) gives the
 * peds' initial tree positions; restated because the positions decide the tree shape.
 *
 * Random vmax.  `default_random_engine generator` (minstd_rand0, seed 1) is global: every Tagent() draws one
 * normal_distribution<double>(1.2, 0.2) value from a FRESH distribution object (no saved second value), peds
 * first, then robots (pedscene.h:57-80).  Peds overwrite theirs with setVmax; robots keep it.  Restated from
 * libstdc++ (bits/random.h, bits/random.tcc: linear_congruential_engine, generate_canonical, normal_distribution
 * polar method) -- third-party, not in the reference tree.
 */
#include "oracle_sfm.h"

#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct v3 {
    double x, y, z;
} v3;

static inline v3 V3(double x, double y, double z) {
    v3 r = {x, y, z};
    return r;
}
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscaled(v3 a, double f) { return V3(f * a.x, f * a.y, f * a.z); } /* Tvector::scaled */
static inline double vlen2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static inline double vlen(v3 a) { /* Tvector::length */
    if ((a.x == 0) && (a.y == 0) && (a.z == 0)) return 0;
    return sqrt(vlen2(a));
}
static inline v3 vnormalized(v3 a) { /* Tvector::normalized */
    double len = vlen(a);
    if (len == 0) return V3(0, 0, 0);
    return V3(a.x / len, a.y / len, a.z / len);
}
static inline v3 vdivs(v3 a, double d) { return vscaled(a, 1 / d); } /* operator/ */
static inline double vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y + a.z * b.z); }

/* atan2 as the reference's libm evaluates it (glibc: <= 1 ulp, misrounds ~0.05 % of its inputs), or -- test-only switch
 * sfm_set_cr_atan2(1) -- correctly rounded by an independent route: libquadmath's 113-bit atan2q rounded once to double.
 * Tagent::socialForce branches on the SIGN of a difference of two atan2 of nearly parallel vectors, so on a crowd at rest
 * a last-bit difference flips a full-size force term; the device's atan2 is correctly rounded (csrc/cr_atan2.h), and in CR
 * mode the oracle is too, which makes the two comparable on every field without a tolerance for that coin. */
static int g_cr_atan2 = 0;
void sfm_set_cr_atan2(int on) { g_cr_atan2 = on; }
static double at2(double y, double x) { return g_cr_atan2 ? (double)atan2q((__float128)y, (__float128)x) : atan2(y, x); }

/* Tvector::angleTo via polarAngle = atan2(y, x) */
static double angle_to(v3 a, v3 b) {
    double angle_this = at2(a.y, a.x);
    double angle_other = at2(b.y, b.x);
    double diff = angle_other - angle_this;
    if (diff > M_PI)
        diff -= 2 * M_PI;
    else if (diff <= -M_PI)
        diff += 2 * M_PI;
    return diff;
}

/* Tvector::lineIntersection (2-D) */
static int line_intersection(v3 p0, v3 p1, v3 p2, v3 p3, v3* out) {
    double s1x = p1.x - p0.x, s1y = p1.y - p0.y;
    double s2x = p3.x - p2.x, s2y = p3.y - p2.y;
    double s = (-s1y * (p0.x - p2.x) + s1x * (p0.y - p2.y)) / (-s2x * s1y + s1x * s2y);
    double t = (s2x * (p0.y - p2.y) - s2y * (p0.x - p2.x)) / (-s2x * s1y + s1x * s2y);
    if (s >= 0 && s <= 1 && t >= 0 && t <= 1) {
        out->x = p0.x + (t * s1x);
        out->y = p0.y + (t * s1y);
        return 1;
    }
    return 0;
}

#define SFM_MAX_WP 8

typedef struct sfm_agent {
    v3 p, v, a;
    v3 desiredforce, socialforce, obstacleforce, lookaheadforce, desired_direction;
    double vmax;
    /* waypoints: circular deque of indices into wp[] */
    int n_wp;
    double wpx[SFM_MAX_WP], wpy[SFM_MAX_WP], wpr[SFM_MAX_WP];
    int dq[SFM_MAX_WP], dq_n; /* dq[0] is the front */
    int destination, lastdestination; /* index or -1 */
} sfm_agent;

typedef struct qnode { /* Ped::Ttree */
    int isleaf;
    double x, y, w, h;
    int child[4]; /* tree1..tree4 */
    int* agents;  /* std::set<const Tagent*>, kept sorted by agent index */
    int n_agents, cap_agents;
} qnode;

struct sfm_scene {
    int n_peds, n_robots, n; /* agents in the scene: peds then robots */
    sfm_agent* ag;
    int n_obs, cap_obs;
    double* obs; /* ax ay bx by */
    qnode* nodes;
    int n_nodes, cap_nodes;
    int* treehash; /* agent -> leaf */
    int* nb;       /* neighbour scratch: sorted unique agent indices */
    int n_nb;
    unsigned char* mark;
};

/* ---- Ped::Ttree (ped_tree.cpp) ---- */
static int q_new(sfm_scene* s, double x, double y, double w, double h) {
    if (s->n_nodes == s->cap_nodes) {
        s->cap_nodes = s->cap_nodes ? 2 * s->cap_nodes : 64;
        s->nodes = (qnode*)realloc(s->nodes, sizeof(qnode) * (size_t)s->cap_nodes);
    }
    qnode* q = &s->nodes[s->n_nodes];
    memset(q, 0, sizeof(*q));
    q->isleaf = 1;
    q->x = x; q->y = y; q->w = w; q->h = h;
    q->child[0] = q->child[1] = q->child[2] = q->child[3] = -1;
    return s->n_nodes++;
}
static void q_set_insert(qnode* q, int a) {
    int lo = 0;
    while (lo < q->n_agents && q->agents[lo] < a) lo++;
    if (lo < q->n_agents && q->agents[lo] == a) return;
    if (q->n_agents == q->cap_agents) {
        q->cap_agents = q->cap_agents ? 2 * q->cap_agents : 16;
        q->agents = (int*)realloc(q->agents, sizeof(int) * (size_t)q->cap_agents);
    }
    memmove(q->agents + lo + 1, q->agents + lo, sizeof(int) * (size_t)(q->n_agents - lo));
    q->agents[lo] = a;
    q->n_agents++;
}
static void q_set_erase(qnode* q, int a) {
    for (int k = 0; k < q->n_agents; k++)
        if (q->agents[k] == a) {
            memmove(q->agents + k, q->agents + k + 1, sizeof(int) * (size_t)(q->n_agents - k - 1));
            q->n_agents--;
            return;
        }
}
static void q_add_agent(sfm_scene* s, int node, int a, int depth);
static void q_to_children(sfm_scene* s, int node, int a, int depth) { /* the four non-exclusive tests */
    const double px = s->ag[a].p.x, py = s->ag[a].p.y;
    const double cx = s->nodes[node].x + s->nodes[node].w / 2, cy = s->nodes[node].y + s->nodes[node].h / 2;
    if ((px >= cx) && (py >= cy)) q_add_agent(s, s->nodes[node].child[2], a, depth + 1); /* 3 */
    if ((px <= cx) && (py <= cy)) q_add_agent(s, s->nodes[node].child[0], a, depth + 1); /* 1 */
    if ((px >= cx) && (py <= cy)) q_add_agent(s, s->nodes[node].child[1], a, depth + 1); /* 2 */
    if ((px <= cx) && (py >= cy)) q_add_agent(s, s->nodes[node].child[3], a, depth + 1); /* 4 */
}
static void q_add_agent(sfm_scene* s, int node, int a, int depth) { /* Ttree::addAgent (ped_tree.cpp:65-96) */
    if (depth > 200) return; /* the reference recurses forever on > 8 coincident agents; callers reject that case */
    if (s->nodes[node].isleaf) {
        q_set_insert(&s->nodes[node], a);
        s->treehash[a] = node;
    } else {
        q_to_children(s, node, a, depth);
    }
    if (s->nodes[node].n_agents > 8) {
        s->nodes[node].isleaf = 0;
        const double x = s->nodes[node].x, y = s->nodes[node].y, w = s->nodes[node].w, h = s->nodes[node].h;
        const int c0 = q_new(s, x, y, w / 2, h / 2); /* addChildren (ped_tree.cpp:101-106) */
        const int c1 = q_new(s, x + w / 2, y, w / 2, h / 2);
        const int c2 = q_new(s, x + w / 2, y + h / 2, w / 2, h / 2);
        const int c3 = q_new(s, x, y + h / 2, w / 2, h / 2);
        s->nodes[node].child[0] = c0; s->nodes[node].child[1] = c1; s->nodes[node].child[2] = c2; s->nodes[node].child[3] = c3;
        while (s->nodes[node].n_agents > 0) {
            const int b = s->nodes[node].agents[0];
            q_to_children(s, node, b, depth);
            q_set_erase(&s->nodes[node], b);
        }
    }
}
static void q_move_agent(sfm_scene* s, int a) { /* Tscene::moveAgent -> Ttree::moveAgent (ped_tree.cpp:131-137) */
    const int leaf = s->treehash[a];
    const qnode* q = &s->nodes[leaf];
    const double px = s->ag[a].p.x, py = s->ag[a].p.y;
    if ((px < q->x) || (px > (q->x + q->w)) || (py < q->y) || (py > (q->y + q->h))) {
        q_add_agent(s, 0, a, 0);              /* scene->placeAgent(a): from the root */
        q_set_erase(&s->nodes[leaf], a);      /* agents.erase(a) on the OLD leaf -- even when it is the new one */
    }
}
/* Tscene::getNeighbors (ped_scene.cpp:217-252): leaves whose rectangle the query square touches */
static void q_neighbors(sfm_scene* s, double x, double y, double dist) {
    int stack[256], sp = 0;
    memset(s->mark, 0, (size_t)s->n);
    stack[sp++] = 0;
    while (sp > 0) {
        const qnode* t = &s->nodes[stack[--sp]];
        if (t->isleaf) {
            for (int k = 0; k < t->n_agents; k++) s->mark[t->agents[k]] = 1;
        } else {
            for (int c = 0; c < 4; c++) {
                const qnode* ch = &s->nodes[t->child[c]];
                if (((x + dist) > ch->x) && ((x - dist) < (ch->x + ch->w)) && ((y + dist) > ch->y) && ((y - dist) < (ch->y + ch->h)))
                    if (sp < 256) stack[sp++] = t->child[c];
            }
        }
    }
    s->n_nb = 0;
    for (int a = 0; a < s->n; a++)
        if (s->mark[a]) s->nb[s->n_nb++] = a;
}

/* ---- glibc rand(): TYPE_3 (x^31 + x^3 + 1) additive feedback generator, seed 1 (stdlib/random_r.c) ---- */
static int32_t g_r[34 + 310 + 8192];
static int g_rk = -1;
static void glibc_srand(unsigned seed) {
    g_r[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        long hi = g_r[i - 1] / 127773, lo = g_r[i - 1] % 127773;
        long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        g_r[i] = (int32_t)word;
    }
    for (int i = 31; i < 34; i++) g_r[i] = g_r[i - 31];
    for (int i = 34; i < 344; i++) g_r[i] = (int32_t)((uint32_t)g_r[i - 31] + (uint32_t)g_r[i - 3]);
    g_rk = 344;
}
static int glibc_rand(void) {
    if (g_rk < 0) glibc_srand(1);
    if (g_rk >= (int)(sizeof(g_r) / sizeof(g_r[0]))) { /* slide the window */
        memmove(g_r, g_r + g_rk - 34, sizeof(int32_t) * 34);
        g_rk = 34;
    }
    g_r[g_rk] = (int32_t)((uint32_t)g_r[g_rk - 31] + (uint32_t)g_r[g_rk - 3]);
    return (int)(((uint32_t)g_r[g_rk++]) >> 1);
}

/* ---- libstdc++ minstd_rand0 + generate_canonical<double,53> + normal_distribution (polar) ---- */
static unsigned long g_lcg = 1u; /* default_random_engine() seed */
static double lcg_next(void) {
    g_lcg = (g_lcg * 16807ul) % 2147483647ul;
    return (double)g_lcg;
}
static double canonical(void) {
    /* b = 53 bits, range R = max - min + 1 = 2147483646, k = max(1, ceil(53 / log2(R))) = 2 */
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
static double normal_fresh(double mean, double stddev) {
    double x, y, r2;
    do {
        x = 2.0 * canonical() - 1.0;
        y = 2.0 * canonical() - 1.0;
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    const double mult = sqrt(-2 * log(r2) / r2);
    /* _M_saved = x * mult is discarded with the distribution object; ret = y * mult */
    return (y * mult) * stddev + mean;
}

void sfm_reseed(void) {
    g_lcg = 1u;
    g_rk = -1;
}

sfm_scene* sfm_create(int n_peds, int n_robots, const float* ped_max_speed) {
    sfm_scene* s = (sfm_scene*)calloc(1, sizeof(sfm_scene));
    s->n_peds = n_peds;
    s->n_robots = n_robots;
    s->n = n_peds + n_robots;
    s->ag = (sfm_agent*)calloc((size_t)(s->n > 0 ? s->n : 1), sizeof(sfm_agent));
    s->treehash = (int*)calloc((size_t)(s->n > 0 ? s->n : 1), sizeof(int));
    s->nb = (int*)calloc((size_t)(s->n > 0 ? s->n : 1), sizeof(int));
    s->mark = (unsigned char*)calloc((size_t)(s->n > 0 ? s->n : 1), 1);
    q_new(s, 0, 10, 10, 10); /* Tscene(0,10,10,10) (pedscene.h:18) */
    for (int i = 0; i < s->n; i++) {
        sfm_agent* a = &s->ag[i];
        a->vmax = normal_fresh(1.2, 0.2); /* Tagent() (ped_agent.cpp:41-44) */
        a->destination = a->lastdestination = -1;
        if (i < n_peds) { /* addPed (pedscene.h:57-69) */
            const double pxx = glibc_rand() / 2147483647.0 * 10.0;
            const double pyy = glibc_rand() / 2147483647.0 * 10.0;
            a->p = V3(pxx, pyy, 0);
            a->vmax = (double)ped_max_speed[i];
        }
        q_add_agent(s, 0, i, 0); /* Tscene::addAgent -> tree->addAgent */
    }
    return s;
}

void sfm_destroy(sfm_scene* s) {
    if (!s) return;
    free(s->ag);
    free(s->obs);
    for (int k = 0; k < s->n_nodes; k++) free(s->nodes[k].agents);
    free(s->nodes);
    free(s->treehash);
    free(s->nb);
    free(s->mark);
    free(s);
}

void sfm_clear_obstacles(sfm_scene* s) {
    if (s) s->n_obs = 0;
}

void sfm_add_obstacle(sfm_scene* s, double ax, double ay, double bx, double by) {
    if (!s) return;
    if (s->n_obs == s->cap_obs) {
        s->cap_obs = s->cap_obs ? 2 * s->cap_obs : 16;
        s->obs = (double*)realloc(s->obs, sizeof(double) * 4 * (size_t)s->cap_obs);
    }
    double* o = s->obs + 4 * s->n_obs++;
    o[0] = ax;
    o[1] = ay;
    o[2] = bx;
    o[3] = by;
}

void sfm_set_ped_pos(sfm_scene* s, int j, double x, double y) { /* setPosition(x, y, 0) (pedscene.h:34-36) */
    if (!s) return;
    s->ag[j].p = V3(x, y, 0);
}

/* PedScene::setWayPoint (pedscene.h:38-46): clearWaypoints, then [goal (r = 1), trajectory points (r = z)] */
void sfm_set_waypoints(sfm_scene* s, int j, double gx, double gy, const double* traj_xyz, int n) {
    if (!s) return;
    sfm_agent* a = &s->ag[j];
    a->destination = a->lastdestination = -1;
    a->dq_n = 0;
    a->n_wp = 0;
    a->wpx[0] = gx;
    a->wpy[0] = gy;
    a->wpr[0] = 1;
    a->n_wp = 1;
    for (int k = 0; k < n && a->n_wp < SFM_MAX_WP; k++) {
        a->wpx[a->n_wp] = traj_xyz[3 * k];
        a->wpy[a->n_wp] = traj_xyz[3 * k + 1];
        a->wpr[a->n_wp] = traj_xyz[3 * k + 2];
        a->n_wp++;
    }
    /* addWaypoint: push_back + destination = waypoints.front() -- the front is NOT popped (ped_agent.cpp:97-100) */
    for (int k = 0; k < a->n_wp; k++) a->dq[a->dq_n++] = k;
    a->destination = a->dq[0];
}

void sfm_set_robot_pos(sfm_scene* s, int i, double x, double y) { /* setPosition(px, py, 1) (pedscene.h:52-55) */
    if (!s || i >= s->n_robots) return;
    s->ag[s->n_peds + i].p = V3(x, y, 1);
}

static v3 desired_force(sfm_agent* a) {
    if ((a->destination == -1) && (a->dq_n > 0)) { /* fetch: front, pop, push back (BEHAVIOR_CIRCULAR) */
        a->destination = a->dq[0];
        for (int k = 0; k + 1 < a->dq_n; k++) a->dq[k] = a->dq[k + 1];
        a->dq[a->dq_n - 1] = a->destination;
    }
    if (a->destination == -1) a->desired_direction = V3(0, 0, 0);
    int reached = 0;
    if (a->destination != -1) {
        /* Twaypoint::getForce, TYPE_POINT and TYPE_NORMAL alike (ped_waypoint.cpp:81-134) */
        const int d = a->destination;
        v3 diff = V3(a->wpx[d] - a->p.x, a->wpy[d] - a->p.y, 0);
        reached = vlen(diff) < a->wpr[d];
        a->desired_direction = vnormalized(diff);
    }
    if ((a->destination != -1) && reached) {
        a->lastdestination = a->destination;
        a->destination = -1;
    }
    return vscaled(vnormalized(a->desired_direction), a->vmax); /* normalized() * vmax */
}

static v3 social_force(const sfm_scene* s, int self) {
    const double lambda_importance = 2.0, gamma = 0.35, n = 2, n_prime = 3;
    const sfm_agent* me = &s->ag[self];
    v3 force = V3(0, 0, 0);
    for (int k = 0; k < s->n_nb; k++) {
        const int o = s->nb[k];
        if (o == self) continue;
        const sfm_agent* other = &s->ag[o];
        v3 diff = vsub(other->p, me->p);
        if (vlen2(diff) > 64.0) continue;
        v3 diff_direction = vnormalized(diff);
        v3 vel_diff = vsub(me->v, other->v);
        v3 interaction_vector = vadd(vscaled(vel_diff, lambda_importance), diff_direction);
        double interaction_length = vlen(interaction_vector);
        v3 interaction_direction = vdivs(interaction_vector, interaction_length);
        double theta = angle_to(interaction_direction, diff_direction);
        int theta_sign = (theta == 0) ? (0) : (int)(theta / fabs(theta));
        double B = gamma * interaction_length;
        double force_velocity_amount = -exp(-vlen(diff) / B - (n_prime * B * theta) * (n_prime * B * theta));
        double force_angle_amount = -theta_sign * exp(-vlen(diff) / B - (n * B * theta) * (n * B * theta));
        v3 force_velocity = vscaled(interaction_direction, force_velocity_amount);
        v3 left_normal = V3(-interaction_direction.y, interaction_direction.x, 0);
        v3 force_angle = vscaled(left_normal, force_angle_amount);
        force = vadd(force, vadd(force_velocity, force_angle));
    }
    return force;
}

static v3 obstacle_force(const sfm_scene* s, int self) {
    const sfm_agent* me = &s->ag[self];
    v3 min_diff = V3(0, 0, 0);
    double min_d2 = INFINITY;
    for (int q = 0; q < s->n_obs; q++) {
        const double* o = s->obs + 4 * q;
        v3 start = V3(o[0], o[1], 0), end = V3(o[2], o[3], 0);
        v3 rel_end = vsub(end, start);
        v3 rel_p = vsub(me->p, start);
        double lambda = vdot(rel_p, rel_end) / vlen2(rel_end);
        v3 closest;
        if (lambda <= 0)
            closest = start;
        else if (lambda >= 1)
            closest = end;
        else
            closest = vadd(start, vscaled(rel_end, lambda));
        v3 diff = vsub(me->p, closest);
        double d2 = vlen2(diff);
        if (d2 < min_d2) {
            min_d2 = d2;
            min_diff = diff;
        }
    }
    double distance = sqrt(min_d2) - 0.2; /* agentRadius */
    double force_amount = exp(-distance / 0.8); /* obstacleForceSigma */
    return vscaled(vnormalized(min_diff), force_amount);
}

static v3 lookahead_force(const sfm_scene* s, int self, v3 e) {
    const double pi = 3.14159265;
    const sfm_agent* me = &s->ag[self];
    int count = 0;
    for (int k = 0; k < s->n_nb; k++) {
        const int o = s->nb[k];
        if (o == self) continue;
        const sfm_agent* other = &s->ag[o];
        double dx = other->p.x - me->p.x;
        double dy = other->p.y - me->p.y;
        double dist2 = (dx * dx + dy * dy);
        if (dist2 < 400) {
            double at2v = at2(-e.x, -e.y);
            double at2d = at2(-dx, -dy);
            double at2v2 = at2(-other->v.x, -other->v.y);
            double sdiff = at2d - at2v;
            if (sdiff > pi) sdiff -= 2 * pi;
            if (sdiff < -pi) sdiff += 2 * pi;
            double vv = at2v - at2v2;
            if (vv > pi) vv -= 2 * pi;
            if (vv < -pi) vv += 2 * pi;
            if (fabs(vv) > 2.5) {
                if ((sdiff < 0) && (sdiff > -0.3)) count--;
                if ((sdiff > 0) && (sdiff < 0.3)) count++;
            }
        }
    }
    v3 lf = V3(0, 0, 0);
    if (count < 0) {
        lf.x = 0.5f * e.y;
        lf.y = 0.5f * -e.x;
    }
    if (count > 0) {
        lf.x = 0.5f * -e.y;
        lf.y = 0.5f * e.x;
    }
    return lf;
}

void sfm_move_agents(sfm_scene* s, double h) {
    if (!s) return;
    /* Tscene::moveAgents: all forces from the t-1 state, then all moves */
    for (int i = 0; i < s->n; i++) {
        sfm_agent* a = &s->ag[i];
        q_neighbors(s, a->p.x, a->p.y, 20.0); /* neighborhoodRange (ped_agent.cpp:499-500) */
        a->desiredforce = desired_force(a);
        a->lookaheadforce = lookahead_force(s, i, a->desired_direction);
        a->socialforce = social_force(s, i);
        a->obstacleforce = obstacle_force(s, i);
    }
    for (int i = 0; i < s->n; i++) {
        sfm_agent* a = &s->ag[i];
        v3 p_desired = vadd(a->p, vscaled(a->v, h));
        for (int q = 0; q < s->n_obs; q++) {
            const double* o = s->obs + 4 * q;
            v3 inter = V3(0, 0, 0);
            if (line_intersection(a->p, p_desired, V3(o[0], o[1], 0), V3(o[2], o[3], 0), &inter) == 1) {
                p_desired = vsub(inter, vscaled(vnormalized(vscaled(a->v, h)), 0.1));
            }
        }
        a->p = p_desired;
        a->a = vadd(vadd(vadd(vadd(vscaled(a->desiredforce, 1.0), vscaled(a->socialforce, 2.1)),
                               vscaled(a->obstacleforce, 1.0)),
                          vscaled(a->lookaheadforce, 1.0)),
                     V3(0, 0, 0));
        a->v = vadd(vscaled(a->v, 0.5), vscaled(a->a, h));
        if (vlen(a->v) > a->vmax) a->v = vscaled(vnormalized(a->v), a->vmax);
        q_move_agent(s, i); /* scene->moveAgent(this) */
    }
}

void sfm_get_ped(const sfm_scene* s, int j, double* x, double* y, double* vx, double* vy) {
    if (!s) return;
    *x = s->ag[j].p.x;
    *y = s->ag[j].p.y;
    *vx = s->ag[j].v.x;
    *vy = s->ag[j].v.y;
}

void sfm_get_agent(const sfm_scene* s, int idx, double* out6) {
    const sfm_agent* a = &s->ag[idx];
    out6[0] = a->p.x; out6[1] = a->p.y; out6[2] = a->p.z;
    out6[3] = a->v.x; out6[4] = a->v.y; out6[5] = a->v.z;
}

double sfm_get_vmax(const sfm_scene* s, int idx) { return s->ag[idx].vmax; }

/* test aid: a digest of the quadtree that does not depend on how its nodes are numbered -- node count, member entries, a sum of
 * per-leaf hashes (rectangle + sorted members) and a sum of per-agent hashes (the rectangle its treehash entry points at) */
static uint64_t digest_mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001b3ull; }
static uint64_t digest_f64(double v) {
    uint64_t b;
    memcpy(&b, &v, sizeof(b));
    return b;
}
void sfm_tree_digest(const sfm_scene* s, uint64_t* out8 /* [8] */) { /* out8[4 .. 7]: one bit per agent that is a member of some leaf */
    uint64_t members = 0, leaves = 0, agents = 0;
    out8[4] = out8[5] = out8[6] = out8[7] = 0;
    for (int k = 0; k < s->n_nodes; k++) {
        const qnode* q = &s->nodes[k];
        if (!q->isleaf || q->n_agents == 0) continue;
        uint64_t h = 0xcbf29ce484222325ull;
        h = digest_mix(h, digest_f64(q->x)); h = digest_mix(h, digest_f64(q->y));
        h = digest_mix(h, digest_f64(q->w)); h = digest_mix(h, digest_f64(q->h));
        h = digest_mix(h, (uint64_t)q->n_agents);
        for (int e = 0; e < q->n_agents; e++) {
            h = digest_mix(h, (uint64_t)q->agents[e]);
            if (q->agents[e] < 256) out8[4 + (q->agents[e] >> 6)] |= 1ull << (q->agents[e] & 63);
        }
        leaves += h;
        members += (uint64_t)q->n_agents;
    }
    for (int a = 0; a < s->n; a++) {
        const qnode* q = &s->nodes[s->treehash[a]];
        uint64_t h = 0xcbf29ce484222325ull;
        h = digest_mix(h, (uint64_t)a);
        h = digest_mix(h, digest_f64(q->x)); h = digest_mix(h, digest_f64(q->y));
        h = digest_mix(h, digest_f64(q->w)); h = digest_mix(h, digest_f64(q->h));
        agents += h;
    }
    out8[0] = (uint64_t)s->n_nodes; out8[1] = members; out8[2] = leaves; out8[3] = agents;
}


#include <stdio.h>
void sfm_debug_dump(const sfm_scene* s) { /* test aid: tree leaves and treehash */
    printf("oracle: n_nodes %d\n", s->n_nodes);
    for (int k = 0; k < s->n_nodes; k++)
        if (s->nodes[k].isleaf && s->nodes[k].n_agents) {
            printf(" leaf %d [%.3f %.3f %.3f %.3f]:", k, s->nodes[k].x, s->nodes[k].y, s->nodes[k].w, s->nodes[k].h);
            for (int q = 0; q < s->nodes[k].n_agents; q++) printf(" %d", s->nodes[k].agents[q]);
            printf("\n");
        }
    printf(" treehash:");
    for (int a = 0; a < s->n; a++) printf(" %d", s->treehash[a]);
    printf("\n");
}
