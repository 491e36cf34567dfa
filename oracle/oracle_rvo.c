/*
 * THIRD-PARTY NOTICE.  The ORCA half-plane construction, the linear programs and the obstacle k-d tree walk in this file follow
 * the RVO2 Library (as vendored by the reference in src/3rdparty/ervo_ros, with the ERVO additions) statement for statement --
 * a float32 LP whose result depends on the order of its operations has essentially one spelling if it is to stay bit-exact:
 *
 *   RVO2 Library.  Copyright 2008 University of North Carolina at Chapel Hill.
 *   Licensed under the Apache License, Version 2.0 (the "License"); you may not use this file except in compliance with the
 *   License.  You may obtain a copy of the License at http://www.apache.org/licenses/LICENSE-2.0
 *   Unless required by applicable law or agreed to in writing, software distributed under the License is distributed on an
 *   "AS IS" BASIS, WITHOUT WARRANTIES OR CONDITIONS OF ANY KIND, either express or implied.  See the License for the specific
 *   language governing permissions and limitations under the License.
 *   Authors: Jur van den Berg, Stephen J. Guy, Jamie Snape, Ming C. Lin, Dinesh Manocha -- <http://gamma.cs.unc.edu/RVO2/>
 *
 * Changes made here: restated in plain C, index links instead of pointers, one translation unit, used as a test oracle only.  See NOTICE at the repository root.
 */
/*
 * oracle_rvo.c -- TEST INFRASTRUCTURE (oracle), never linked into the product library.
 *
 * CPU restatement (plain C, float32) of the RVO2 v2.0.x ORCA library + the ERVO add-on the
 * reference vendors in src/3rdparty/ervo_ros:
 *   Agent::computeNeighbors            src/Agent.cpp:50-61
 *   addEvacVelocity                    src/Agent.cpp:63-69
 *   Agent::computeNewVelocity[ForERVO] src/Agent.cpp:72-434, 437-793
 *   insertAgentNeighbor / Obstacle     src/Agent.cpp:795-838
 *   Agent::update                      src/Agent.cpp:840-843
 *   linearProgram1/2/3                 src/Agent.cpp:845-1001
 *   KdTree (agents + obstacle BSP)     src/KdTree.cpp:44-353
 *   RVOSimulator::addObstacle/doStep   src/RVOSimulator.cpp:130-199
 *   Vector2 / Definitions helpers      include/ervo_ros/Vector2.h, Definitions.h
 * Data layout (SoA + index links) and control structure are this repo's own; the float
 * operation order follows the reference so results are bit-identical (checked against
 * oracle/_ref/librvo_ref.so in tests/test_oracle_rvo_ref.py).
 *
 * Build with -ffp-contract=off and without -ffast-math.
 */
#include "oracle_rvo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define RVO_EPS 0.00001f

typedef struct v2 {
    float x, y;
} v2;

static inline v2 V(float x, float y) {
    v2 r = {x, y};
    return r;
}
static inline v2 vadd(v2 a, v2 b) { return V(a.x + b.x, a.y + b.y); }
static inline v2 vsub(v2 a, v2 b) { return V(a.x - b.x, a.y - b.y); }
static inline v2 vneg(v2 a) { return V(-a.x, -a.y); }
static inline float vdot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
static inline v2 vmul(v2 a, float s) { return V(a.x * s, a.y * s); }
static inline v2 smul(float s, v2 a) { return V(s * a.x, s * a.y); }
static inline v2 vdiv(v2 a, float s) { /* Vector2::operator/ multiplies by 1.0f/s */
    const float inv = 1.0f / s;
    return V(a.x * inv, a.y * inv);
}
static inline float vabssq(v2 a) { return vdot(a, a); }
static inline float vabs(v2 a) { return sqrtf(vdot(a, a)); }
static inline float vdet(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }
static inline v2 vnorm(v2 a) { return vdiv(a, vabs(a)); }
static inline float sqrf(float a) { return a * a; }
static inline float left_of(v2 a, v2 b, v2 c) { return vdet(vsub(a, c), vsub(b, a)); }
static inline float fmin_std(float a, float b) { return (b < a) ? b : a; } /* std::min */
static inline float fmax_std(float a, float b) { return (a < b) ? b : a; } /* std::max */

static float dist_sq_point_segment(v2 a, v2 b, v2 c) {
    const float r = vdot(vsub(c, a), vsub(b, a)) / vabssq(vsub(b, a));
    if (r < 0.0f) {
        return vabssq(vsub(c, a));
    } else if (r > 1.0f) {
        return vabssq(vsub(c, b));
    } else {
        return vabssq(vsub(c, vadd(a, smul(r, vsub(b, a)))));
    }
}

static int g_bruteforce = 0;
void rvo_set_bruteforce(int on) { g_bruteforce = on; }

/* ------------------------------------------------------------------ lifetime */

rvo_sim* rvo_create(float time_step) {
    rvo_sim* s = (rvo_sim*)calloc(1, sizeof(rvo_sim));
    s->time_step = time_step;
    s->oroot = -1;
    return s;
}

void rvo_destroy(rvo_sim* s) {
    if (!s) return;
    free(s->px); free(s->py); free(s->vx); free(s->vy); free(s->prefx); free(s->prefy);
    free(s->newvx); free(s->newvy); free(s->radius); free(s->max_speed);
    free(s->neighbor_dist); free(s->time_horizon); free(s->time_horizon_obst);
    free(s->max_neighbors); free(s->obst); free(s->onodes); free(s->order); free(s->atree);
    free(s->lines); free(s->proj); free(s->on_dist); free(s->on_idx);
    free(s);
}

#define GROW(ptr, type, n) ptr = (type*)realloc(ptr, sizeof(type) * (size_t)(n))

int rvo_add_agent(rvo_sim* s, float x, float y, float neighbor_dist, int max_neighbors,
                  float time_horizon, float time_horizon_obst, float radius, float max_speed) {
    if (s->n_agents == s->cap_agents) {
        int c = s->cap_agents ? s->cap_agents * 2 : 64;
        GROW(s->px, float, c); GROW(s->py, float, c); GROW(s->vx, float, c); GROW(s->vy, float, c);
        GROW(s->prefx, float, c); GROW(s->prefy, float, c); GROW(s->newvx, float, c);
        GROW(s->newvy, float, c); GROW(s->radius, float, c); GROW(s->max_speed, float, c);
        GROW(s->neighbor_dist, float, c); GROW(s->time_horizon, float, c);
        GROW(s->time_horizon_obst, float, c); GROW(s->max_neighbors, int32_t, c);
        s->cap_agents = c;
    }
    int i = s->n_agents++;
    s->px[i] = x; s->py[i] = y; s->vx[i] = 0.0f; s->vy[i] = 0.0f;
    s->prefx[i] = 0.0f; s->prefy[i] = 0.0f; s->newvx[i] = 0.0f; s->newvy[i] = 0.0f;
    s->radius[i] = radius; s->max_speed[i] = max_speed; s->neighbor_dist[i] = neighbor_dist;
    s->time_horizon[i] = time_horizon; s->time_horizon_obst[i] = time_horizon_obst;
    s->max_neighbors[i] = max_neighbors > 64 ? 64 : max_neighbors;
    return i;
}

static int new_obstacle(rvo_sim* s) {
    if (s->n_obst == s->cap_obst) {
        s->cap_obst = s->cap_obst ? s->cap_obst * 2 : 64;
        GROW(s->obst, rvo_obstacle, s->cap_obst);
    }
    rvo_obstacle* o = &s->obst[s->n_obst];
    memset(o, 0, sizeof(*o));
    o->next = o->prev = -1;
    return s->n_obst++;
}

int rvo_add_obstacle(rvo_sim* s, const float* xy, int n) {
    if (n < 2) return -1;
    const int first = s->n_obst;
    for (int i = 0; i < n; ++i) {
        int k = new_obstacle(s);
        rvo_obstacle* o = &s->obst[k];
        o->px = xy[2 * i];
        o->py = xy[2 * i + 1];
        if (i != 0) {
            o->prev = k - 1;
            s->obst[k - 1].next = k;
        }
        if (i == n - 1) {
            o->next = first;
            s->obst[first].prev = k;
        }
        const int inext = (i == n - 1 ? 0 : i + 1);
        const int iprev = (i == 0 ? n - 1 : i - 1);
        v2 d = vnorm(vsub(V(xy[2 * inext], xy[2 * inext + 1]), V(xy[2 * i], xy[2 * i + 1])));
        o->ux = d.x;
        o->uy = d.y;
        if (n == 2) {
            o->is_convex = 1;
        } else {
            o->is_convex = (left_of(V(xy[2 * iprev], xy[2 * iprev + 1]), V(xy[2 * i], xy[2 * i + 1]),
                                    V(xy[2 * inext], xy[2 * inext + 1])) >= 0.0f);
        }
    }
    return first;
}

void rvo_clear_obstacles(rvo_sim* s) {
    s->n_obst = 0;
    /* the reference leaves the (now dangling) obstacle tree in place until processObstacles
     * rebuilds it (RVOSimulator.cpp:172-179); it is always rebuilt before the next query
     * (img_env.cpp:166, 283), so dropping it here is equivalent. */
    s->n_onodes = 0;
    s->oroot = -1;
}

/* ---------------------------------------------------------- obstacle BSP tree */

static int new_onode(rvo_sim* s) {
    if (s->n_onodes == s->cap_onodes) {
        s->cap_onodes = s->cap_onodes ? s->cap_onodes * 2 : 64;
        GROW(s->onodes, rvo_onode, s->cap_onodes);
    }
    return s->n_onodes++;
}

static inline v2 opoint(const rvo_sim* s, int i) { return V(s->obst[i].px, s->obst[i].py); }

/* lexicographic std::pair<size_t,size_t> compare on (max, min) */
static inline int pair_ge(size_t a1, size_t a2, size_t b1, size_t b2) {
    return (a1 > b1) || (a1 == b1 && a2 >= b2);
}
static inline size_t zmax(size_t a, size_t b) { return a > b ? a : b; }
static inline size_t zmin(size_t a, size_t b) { return a < b ? a : b; }

static int build_obstacle_tree(rvo_sim* s, const int* obstacles, size_t n) {
    if (n == 0) return -1;
    const int node = new_onode(s);
    size_t optimal = 0, min_left = n, min_right = n;
    for (size_t i = 0; i < n; ++i) {
        size_t left_size = 0, right_size = 0;
        const int i1 = obstacles[i];
        const int i2 = s->obst[i1].next;
        for (size_t j = 0; j < n; ++j) {
            if (i == j) continue;
            const int j1 = obstacles[j];
            const int j2 = s->obst[j1].next;
            const float j1l = left_of(opoint(s, i1), opoint(s, i2), opoint(s, j1));
            const float j2l = left_of(opoint(s, i1), opoint(s, i2), opoint(s, j2));
            if (j1l >= -RVO_EPS && j2l >= -RVO_EPS) {
                ++left_size;
            } else if (j1l <= RVO_EPS && j2l <= RVO_EPS) {
                ++right_size;
            } else {
                ++left_size;
                ++right_size;
            }
            if (pair_ge(zmax(left_size, right_size), zmin(left_size, right_size),
                        zmax(min_left, min_right), zmin(min_left, min_right))) {
                break;
            }
        }
        if (!pair_ge(zmax(left_size, right_size), zmin(left_size, right_size),
                     zmax(min_left, min_right), zmin(min_left, min_right))) {
            min_left = left_size;
            min_right = right_size;
            optimal = i;
        }
    }
    int* left_obs = (int*)malloc(sizeof(int) * (min_left + 1));
    int* right_obs = (int*)malloc(sizeof(int) * (min_right + 1));
    size_t lc = 0, rc = 0;
    const size_t i = optimal;
    const int i1 = obstacles[i];
    const int i2 = s->obst[i1].next;
    for (size_t j = 0; j < n; ++j) {
        if (i == j) continue;
        const int j1 = obstacles[j];
        const int j2 = s->obst[j1].next;
        const float j1l = left_of(opoint(s, i1), opoint(s, i2), opoint(s, j1));
        const float j2l = left_of(opoint(s, i1), opoint(s, i2), opoint(s, j2));
        if (j1l >= -RVO_EPS && j2l >= -RVO_EPS) {
            left_obs[lc++] = j1;
        } else if (j1l <= RVO_EPS && j2l <= RVO_EPS) {
            right_obs[rc++] = j1;
        } else {
            /* split obstacle j */
            const float t = vdet(vsub(opoint(s, i2), opoint(s, i1)), vsub(opoint(s, j1), opoint(s, i1))) /
                            vdet(vsub(opoint(s, i2), opoint(s, i1)), vsub(opoint(s, j1), opoint(s, j2)));
            const v2 sp = vadd(opoint(s, j1), smul(t, vsub(opoint(s, j2), opoint(s, j1))));
            const int nn = new_obstacle(s); /* may move s->obst */
            rvo_obstacle* no = &s->obst[nn];
            no->px = sp.x;
            no->py = sp.y;
            no->prev = j1;
            no->next = j2;
            no->is_convex = 1;
            no->ux = s->obst[j1].ux;
            no->uy = s->obst[j1].uy;
            s->obst[j1].next = nn;
            s->obst[j2].prev = nn;
            if (j1l > 0.0f) {
                left_obs[lc++] = j1;
                right_obs[rc++] = nn;
            } else {
                right_obs[rc++] = j1;
                left_obs[lc++] = nn;
            }
        }
    }
    s->onodes[node].obstacle = i1;
    const int l = build_obstacle_tree(s, left_obs, lc);
    const int r = build_obstacle_tree(s, right_obs, rc);
    s->onodes[node].left = l;
    s->onodes[node].right = r;
    free(left_obs);
    free(right_obs);
    return node;
}

void rvo_process_obstacles(rvo_sim* s) {
    s->n_onodes = 0;
    const int n = s->n_obst;
    int* all = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) all[i] = i;
    s->oroot = build_obstacle_tree(s, all, (size_t)n);
    free(all);
}

/* -------------------------------------------------------------- agent kd-tree */

#define MAX_LEAF_SIZE 10

static void build_agent_tree_rec(rvo_sim* s, size_t begin, size_t end, size_t node) {
    rvo_anode* t = s->atree;
    t[node].begin = begin;
    t[node].end = end;
    t[node].minX = t[node].maxX = s->px[s->order[begin]];
    t[node].minY = t[node].maxY = s->py[s->order[begin]];
    for (size_t i = begin + 1; i < end; ++i) {
        t[node].maxX = fmax_std(t[node].maxX, s->px[s->order[i]]);
        t[node].minX = fmin_std(t[node].minX, s->px[s->order[i]]);
        t[node].maxY = fmax_std(t[node].maxY, s->py[s->order[i]]);
        t[node].minY = fmin_std(t[node].minY, s->py[s->order[i]]);
    }
    if (end - begin > MAX_LEAF_SIZE) {
        const int vertical = (t[node].maxX - t[node].minX > t[node].maxY - t[node].minY);
        const float split = (vertical ? 0.5f * (t[node].maxX + t[node].minX)
                                      : 0.5f * (t[node].maxY + t[node].minY));
        size_t left = begin, right = end;
        while (left < right) {
            while (left < right &&
                   (vertical ? s->px[s->order[left]] : s->py[s->order[left]]) < split) {
                ++left;
            }
            while (right > left &&
                   (vertical ? s->px[s->order[right - 1]] : s->py[s->order[right - 1]]) >= split) {
                --right;
            }
            if (left < right) {
                int32_t tmp = s->order[left];
                s->order[left] = s->order[right - 1];
                s->order[right - 1] = tmp;
                ++left;
                --right;
            }
        }
        if (left == begin) {
            ++left;
            ++right;
        }
        t[node].left = node + 1;
        t[node].right = node + 2 * (left - begin);
        build_agent_tree_rec(s, begin, left, t[node].left);
        build_agent_tree_rec(s, left, end, t[node].right);
    }
}

static void build_agent_tree(rvo_sim* s) {
    if (s->n_order < s->n_agents) {
        GROW(s->order, int32_t, s->n_agents);
        for (int i = s->n_order; i < s->n_agents; ++i) s->order[i] = i;
        s->n_order = s->n_agents;
        GROW(s->atree, rvo_anode, 2 * (size_t)s->n_agents);
    }
    if (s->n_order > 0) build_agent_tree_rec(s, 0, (size_t)s->n_order, 0);
}

static void insert_agent_neighbor(rvo_sim* s, int self, int other, float* range_sq) {
    if (self == other) return;
    const float dist_sq = vabssq(vsub(V(s->px[self], s->py[self]), V(s->px[other], s->py[other])));
    if (dist_sq < *range_sq) {
        const int maxn = s->max_neighbors[self];
        if (s->n_an < maxn) {
            s->an_dist[s->n_an] = dist_sq;
            s->an_idx[s->n_an] = other;
            s->n_an++;
        }
        int i = s->n_an - 1;
        while (i != 0 && dist_sq < s->an_dist[i - 1]) {
            s->an_dist[i] = s->an_dist[i - 1];
            s->an_idx[i] = s->an_idx[i - 1];
            --i;
        }
        s->an_dist[i] = dist_sq;
        s->an_idx[i] = other;
        if (s->n_an == maxn) *range_sq = s->an_dist[s->n_an - 1];
    }
}

static void query_agent_tree(rvo_sim* s, int self, float* range_sq, size_t node) {
    const rvo_anode* t = s->atree;
    if (t[node].end - t[node].begin <= MAX_LEAF_SIZE) {
        for (size_t i = t[node].begin; i < t[node].end; ++i) {
            insert_agent_neighbor(s, self, s->order[i], range_sq);
        }
    } else {
        const float x = s->px[self], y = s->py[self];
        const rvo_anode* L = &t[t[node].left];
        const rvo_anode* R = &t[t[node].right];
        const float dl = sqrf(fmax_std(0.0f, L->minX - x)) + sqrf(fmax_std(0.0f, x - L->maxX)) +
                         sqrf(fmax_std(0.0f, L->minY - y)) + sqrf(fmax_std(0.0f, y - L->maxY));
        const float dr = sqrf(fmax_std(0.0f, R->minX - x)) + sqrf(fmax_std(0.0f, x - R->maxX)) +
                         sqrf(fmax_std(0.0f, R->minY - y)) + sqrf(fmax_std(0.0f, y - R->maxY));
        if (dl < dr) {
            if (dl < *range_sq) {
                query_agent_tree(s, self, range_sq, t[node].left);
                if (dr < *range_sq) query_agent_tree(s, self, range_sq, t[node].right);
            }
        } else {
            if (dr < *range_sq) {
                query_agent_tree(s, self, range_sq, t[node].right);
                if (dl < *range_sq) query_agent_tree(s, self, range_sq, t[node].left);
            }
        }
    }
}

static void insert_obstacle_neighbor(rvo_sim* s, int self, int ob, float range_sq) {
    const int nx = s->obst[ob].next;
    const float dist_sq =
        dist_sq_point_segment(opoint(s, ob), opoint(s, nx), V(s->px[self], s->py[self]));
    if (dist_sq < range_sq) {
        if (s->n_on == s->cap_on) {
            s->cap_on = s->cap_on ? s->cap_on * 2 : 64;
            GROW(s->on_dist, float, s->cap_on);
            GROW(s->on_idx, int32_t, s->cap_on);
        }
        s->on_dist[s->n_on] = dist_sq;
        s->on_idx[s->n_on] = ob;
        s->n_on++;
        int i = s->n_on - 1;
        while (i != 0 && dist_sq < s->on_dist[i - 1]) {
            s->on_dist[i] = s->on_dist[i - 1];
            s->on_idx[i] = s->on_idx[i - 1];
            --i;
        }
        s->on_dist[i] = dist_sq;
        s->on_idx[i] = ob;
    }
}

static void query_obstacle_tree(rvo_sim* s, int self, float range_sq, int node) {
    if (node < 0) return;
    const int o1 = s->onodes[node].obstacle;
    const int o2 = s->obst[o1].next;
    const float agent_left = left_of(opoint(s, o1), opoint(s, o2), V(s->px[self], s->py[self]));
    query_obstacle_tree(s, self, range_sq,
                        (agent_left >= 0.0f ? s->onodes[node].left : s->onodes[node].right));
    const float dist_sq_line = sqrf(agent_left) / vabssq(vsub(opoint(s, o2), opoint(s, o1)));
    if (dist_sq_line < range_sq) {
        if (agent_left < 0.0f) {
            insert_obstacle_neighbor(s, self, o1, range_sq);
        }
        query_obstacle_tree(s, self, range_sq,
                            (agent_left >= 0.0f ? s->onodes[node].right : s->onodes[node].left));
    }
}

static void compute_neighbors(rvo_sim* s, int self) {
    s->n_on = 0;
    float range_sq = sqrf(s->time_horizon_obst[self] * s->max_speed[self] + s->radius[self]);
    query_obstacle_tree(s, self, range_sq, s->oroot);
    s->n_an = 0;
    if (s->max_neighbors[self] > 0) {
        range_sq = sqrf(s->neighbor_dist[self]);
        if (g_bruteforce) {
            for (int j = 0; j < s->n_agents; ++j) insert_agent_neighbor(s, self, j, &range_sq);
        } else {
            query_agent_tree(s, self, &range_sq, 0);
        }
    }
}

/* -------------------------------------------------------------- linear programs */

static int linear_program1(const rvo_line* lines, size_t line_no, float radius, v2 opt,
                           int direction_opt, v2* result) {
    const v2 lp = V(lines[line_no].px, lines[line_no].py);
    const v2 ld = V(lines[line_no].dx, lines[line_no].dy);
    const float dot_product = vdot(lp, ld);
    const float discriminant = sqrf(dot_product) + sqrf(radius) - vabssq(lp);
    if (discriminant < 0.0f) return 0;
    const float sqrt_disc = sqrtf(discriminant);
    float t_left = -dot_product - sqrt_disc;
    float t_right = -dot_product + sqrt_disc;
    for (size_t i = 0; i < line_no; ++i) {
        const v2 ip = V(lines[i].px, lines[i].py);
        const v2 id = V(lines[i].dx, lines[i].dy);
        const float denominator = vdet(ld, id);
        const float numerator = vdet(id, vsub(lp, ip));
        if (fabsf(denominator) <= RVO_EPS) {
            if (numerator < 0.0f) {
                return 0;
            } else {
                continue;
            }
        }
        const float t = numerator / denominator;
        if (denominator >= 0.0f) {
            t_right = fmin_std(t_right, t);
        } else {
            t_left = fmax_std(t_left, t);
        }
        if (t_left > t_right) return 0;
    }
    if (direction_opt) {
        if (vdot(opt, ld) > 0.0f) {
            *result = vadd(lp, smul(t_right, ld));
        } else {
            *result = vadd(lp, smul(t_left, ld));
        }
    } else {
        const float t = vdot(ld, vsub(opt, lp));
        if (t < t_left) {
            *result = vadd(lp, smul(t_left, ld));
        } else if (t > t_right) {
            *result = vadd(lp, smul(t_right, ld));
        } else {
            *result = vadd(lp, smul(t, ld));
        }
    }
    return 1;
}

static size_t linear_program2(const rvo_line* lines, size_t n, float radius, v2 opt,
                              int direction_opt, v2* result) {
    if (direction_opt) {
        *result = vmul(opt, radius);
    } else if (vabssq(opt) > sqrf(radius)) {
        *result = vmul(vnorm(opt), radius);
    } else {
        *result = opt;
    }
    for (size_t i = 0; i < n; ++i) {
        const v2 ip = V(lines[i].px, lines[i].py);
        const v2 id = V(lines[i].dx, lines[i].dy);
        if (vdet(id, vsub(ip, *result)) > 0.0f) {
            const v2 temp = *result;
            if (!linear_program1(lines, i, radius, opt, direction_opt, result)) {
                *result = temp;
                return i;
            }
        }
    }
    return n;
}

static void linear_program3(rvo_sim* s, const rvo_line* lines, size_t n, size_t num_obst_lines,
                            size_t begin_line, float radius, v2* result) {
    float distance = 0.0f;
    rvo_line* proj = s->proj;
    for (size_t i = begin_line; i < n; ++i) {
        const v2 ip = V(lines[i].px, lines[i].py);
        const v2 id = V(lines[i].dx, lines[i].dy);
        if (vdet(id, vsub(ip, *result)) > distance) {
            size_t np = 0;
            for (size_t k = 0; k < num_obst_lines; ++k) proj[np++] = lines[k];
            for (size_t j = num_obst_lines; j < i; ++j) {
                const v2 jp = V(lines[j].px, lines[j].py);
                const v2 jd = V(lines[j].dx, lines[j].dy);
                v2 pt;
                const float determinant = vdet(id, jd);
                if (fabsf(determinant) <= RVO_EPS) {
                    if (vdot(id, jd) > 0.0f) {
                        continue;
                    } else {
                        pt = smul(0.5f, vadd(ip, jp));
                    }
                } else {
                    pt = vadd(ip, vmul(id, vdet(jd, vsub(ip, jp)) / determinant));
                }
                const v2 dir = vnorm(vsub(jd, id));
                proj[np].px = pt.x;
                proj[np].py = pt.y;
                proj[np].dx = dir.x;
                proj[np].dy = dir.y;
                np++;
            }
            const v2 temp = *result;
            if (linear_program2(proj, np, radius, V(-id.y, id.x), 1, result) < np) {
                *result = temp;
            }
            distance = vdet(id, vsub(ip, *result));
        }
    }
}

/* ------------------------------------------------------------ new velocity */

static inline void push_line(rvo_sim* s, size_t* n, v2 p, v2 d) {
    s->lines[*n].px = p.x;
    s->lines[*n].py = p.y;
    s->lines[*n].dx = d.x;
    s->lines[*n].dy = d.y;
    (*n)++;
}

static void compute_new_velocity(rvo_sim* s, int self, const float* ps_xy, const float* rs,
                                 int n_src) {
    const int need = s->n_on + s->n_an + 4;
    if (need > s->cap_lines) {
        s->cap_lines = need * 2;
        GROW(s->lines, rvo_line, s->cap_lines);
        GROW(s->proj, rvo_line, s->cap_lines);
    }
    size_t nl = 0;
    const v2 pos = V(s->px[self], s->py[self]);
    const v2 vel = V(s->vx[self], s->vy[self]);
    const float radius = s->radius[self];
    const float inv_tho = 1.0f / s->time_horizon_obst[self];

    for (int i = 0; i < s->n_on; ++i) {
        int o1 = s->on_idx[i];
        int o2 = s->obst[o1].next;
        const v2 rel1 = vsub(opoint(s, o1), pos);
        const v2 rel2 = vsub(opoint(s, o2), pos);

        int covered = 0;
        for (size_t j = 0; j < nl; ++j) {
            const v2 lp = V(s->lines[j].px, s->lines[j].py);
            const v2 ld = V(s->lines[j].dx, s->lines[j].dy);
            if (vdet(vsub(smul(inv_tho, rel1), lp), ld) - inv_tho * radius >= -RVO_EPS &&
                vdet(vsub(smul(inv_tho, rel2), lp), ld) - inv_tho * radius >= -RVO_EPS) {
                covered = 1;
                break;
            }
        }
        if (covered) continue;

        const float dsq1 = vabssq(rel1);
        const float dsq2 = vabssq(rel2);
        const float rsq = sqrf(radius);
        const v2 ovec = vsub(opoint(s, o2), opoint(s, o1));
        const float sp = vdot(vneg(rel1), ovec) / vabssq(ovec);
        const float dsq_line = vabssq(vsub(vneg(rel1), smul(sp, ovec)));
        const v2 u1 = V(s->obst[o1].ux, s->obst[o1].uy);

        if (sp < 0.0f && dsq1 <= rsq) {
            if (s->obst[o1].is_convex) {
                push_line(s, &nl, V(0.0f, 0.0f), vnorm(V(-rel1.y, rel1.x)));
            }
            continue;
        } else if (sp > 1.0f && dsq2 <= rsq) {
            const v2 u2 = V(s->obst[o2].ux, s->obst[o2].uy);
            if (s->obst[o2].is_convex && vdet(rel2, u2) >= 0.0f) {
                push_line(s, &nl, V(0.0f, 0.0f), vnorm(V(-rel2.y, rel2.x)));
            }
            continue;
        } else if (sp >= 0.0f && sp < 1.0f && dsq_line <= rsq) {
            push_line(s, &nl, V(0.0f, 0.0f), vneg(u1));
            continue;
        }

        v2 left_leg, right_leg;
        if (sp < 0.0f && dsq_line <= rsq) {
            if (!s->obst[o1].is_convex) continue;
            o2 = o1;
            const float leg1 = sqrtf(dsq1 - rsq);
            left_leg = vdiv(V(rel1.x * leg1 - rel1.y * radius, rel1.x * radius + rel1.y * leg1), dsq1);
            right_leg = vdiv(V(rel1.x * leg1 + rel1.y * radius, -rel1.x * radius + rel1.y * leg1), dsq1);
        } else if (sp > 1.0f && dsq_line <= rsq) {
            if (!s->obst[o2].is_convex) continue;
            o1 = o2;
            const float leg2 = sqrtf(dsq2 - rsq);
            left_leg = vdiv(V(rel2.x * leg2 - rel2.y * radius, rel2.x * radius + rel2.y * leg2), dsq2);
            right_leg = vdiv(V(rel2.x * leg2 + rel2.y * radius, -rel2.x * radius + rel2.y * leg2), dsq2);
        } else {
            if (s->obst[o1].is_convex) {
                const float leg1 = sqrtf(dsq1 - rsq);
                left_leg = vdiv(V(rel1.x * leg1 - rel1.y * radius, rel1.x * radius + rel1.y * leg1), dsq1);
            } else {
                left_leg = vneg(u1);
            }
            if (s->obst[o2].is_convex) {
                const float leg2 = sqrtf(dsq2 - rsq);
                right_leg = vdiv(V(rel2.x * leg2 + rel2.y * radius, -rel2.x * radius + rel2.y * leg2), dsq2);
            } else {
                right_leg = u1;
            }
        }

        /* NOTE: after "o1 = o2" / "o2 = o1" above the unit direction used below is that of the
         * (possibly re-assigned) obstacle1, as in the reference. */
        const v2 uo1 = V(s->obst[o1].ux, s->obst[o1].uy);
        const v2 uo2 = V(s->obst[o2].ux, s->obst[o2].uy);
        const int ln = s->obst[o1].prev;
        const v2 uln = V(s->obst[ln].ux, s->obst[ln].uy);
        int left_foreign = 0, right_foreign = 0;
        if (s->obst[o1].is_convex && vdet(left_leg, vneg(uln)) >= 0.0f) {
            left_leg = vneg(uln);
            left_foreign = 1;
        }
        if (s->obst[o2].is_convex && vdet(right_leg, uo2) <= 0.0f) {
            right_leg = uo2;
            right_foreign = 1;
        }

        const v2 left_cutoff = smul(inv_tho, vsub(opoint(s, o1), pos));
        const v2 right_cutoff = smul(inv_tho, vsub(opoint(s, o2), pos));
        const v2 cutoff_vec = vsub(right_cutoff, left_cutoff);

        const float t = (o1 == o2 ? 0.5f
                                  : vdot(vsub(vel, left_cutoff), cutoff_vec) / vabssq(cutoff_vec));
        const float t_left = vdot(vsub(vel, left_cutoff), left_leg);
        const float t_right = vdot(vsub(vel, right_cutoff), right_leg);

        if ((t < 0.0f && t_left < 0.0f) || (o1 == o2 && t_left < 0.0f && t_right < 0.0f)) {
            const v2 unit_w = vnorm(vsub(vel, left_cutoff));
            push_line(s, &nl, vadd(left_cutoff, smul(radius * inv_tho, unit_w)), V(unit_w.y, -unit_w.x));
            continue;
        } else if (t > 1.0f && t_right < 0.0f) {
            const v2 unit_w = vnorm(vsub(vel, right_cutoff));
            push_line(s, &nl, vadd(right_cutoff, smul(radius * inv_tho, unit_w)), V(unit_w.y, -unit_w.x));
            continue;
        }

        const float dsq_cutoff = ((t < 0.0f || t > 1.0f || o1 == o2)
                                      ? INFINITY
                                      : vabssq(vsub(vel, vadd(left_cutoff, smul(t, cutoff_vec)))));
        const float dsq_left = ((t_left < 0.0f)
                                    ? INFINITY
                                    : vabssq(vsub(vel, vadd(left_cutoff, smul(t_left, left_leg)))));
        const float dsq_right = ((t_right < 0.0f)
                                     ? INFINITY
                                     : vabssq(vsub(vel, vadd(right_cutoff, smul(t_right, right_leg)))));

        if (dsq_cutoff <= dsq_left && dsq_cutoff <= dsq_right) {
            const v2 d = vneg(uo1);
            push_line(s, &nl, vadd(left_cutoff, smul(radius * inv_tho, V(-d.y, d.x))), d);
            continue;
        } else if (dsq_left <= dsq_right) {
            if (left_foreign) continue;
            const v2 d = left_leg;
            push_line(s, &nl, vadd(left_cutoff, smul(radius * inv_tho, V(-d.y, d.x))), d);
            continue;
        } else {
            if (right_foreign) continue;
            const v2 d = vneg(right_leg);
            push_line(s, &nl, vadd(right_cutoff, smul(radius * inv_tho, V(-d.y, d.x))), d);
            continue;
        }
    }

    const size_t num_obst_lines = nl;
    const float inv_th = 1.0f / s->time_horizon[self];

    for (int i = 0; i < s->n_an; ++i) {
        const int other = s->an_idx[i];
        const v2 rel_pos = vsub(V(s->px[other], s->py[other]), pos);
        const v2 rel_vel = vsub(vel, V(s->vx[other], s->vy[other]));
        const float dist_sq = vabssq(rel_pos);
        const float comb = radius + s->radius[other];
        const float comb_sq = sqrf(comb);
        v2 dir, u;
        if (dist_sq > comb_sq) {
            const v2 w = vsub(rel_vel, smul(inv_th, rel_pos));
            const float wl_sq = vabssq(w);
            const float dp1 = vdot(w, rel_pos);
            if (dp1 < 0.0f && sqrf(dp1) > comb_sq * wl_sq) {
                const float wl = sqrtf(wl_sq);
                const v2 unit_w = vdiv(w, wl);
                dir = V(unit_w.y, -unit_w.x);
                u = smul(comb * inv_th - wl, unit_w);
            } else {
                const float leg = sqrtf(dist_sq - comb_sq);
                if (vdet(rel_pos, w) > 0.0f) {
                    dir = vdiv(V(rel_pos.x * leg - rel_pos.y * comb, rel_pos.x * comb + rel_pos.y * leg), dist_sq);
                } else {
                    dir = vdiv(vneg(V(rel_pos.x * leg + rel_pos.y * comb, -rel_pos.x * comb + rel_pos.y * leg)), dist_sq);
                }
                const float dp2 = vdot(rel_vel, dir);
                u = vsub(smul(dp2, dir), rel_vel);
            }
        } else {
            const float inv_ts = 1.0f / s->time_step;
            const v2 w = vsub(rel_vel, smul(inv_ts, rel_pos));
            const float wl = vabs(w);
            const v2 unit_w = vdiv(w, wl);
            dir = V(unit_w.y, -unit_w.x);
            u = smul(comb * inv_ts - wl, unit_w);
        }
        push_line(s, &nl, vadd(vel, smul(0.5f, u)), dir);
    }

    v2 nv = V(s->newvx[self], s->newvy[self]);
    const v2 pref = V(s->prefx[self], s->prefy[self]);
    const size_t fail = linear_program2(s->lines, nl, s->max_speed[self], pref, 0, &nv);
    if (fail < nl) {
        linear_program3(s, s->lines, nl, num_obst_lines, fail, s->max_speed[self], &nv);
    }
    for (int k = 0; k < n_src; ++k) { /* ERVO: addEvacVelocity */
        const v2 evac = vsub(pos, V(ps_xy[2 * k], ps_xy[2 * k + 1]));
        const float a = vabs(evac);
        if (a > rs[k] || (double)a < 1e-4) continue;
        nv = vadd(nv, vnorm(evac));
    }
    s->newvx[self] = nv.x;
    s->newvy[self] = nv.y;
}

void rvo_do_step(rvo_sim* s, int n_active, const float* ps_xy, const float* rs, int n_src) {
    if (!g_bruteforce) build_agent_tree(s);
    if (n_active > s->n_agents) n_active = s->n_agents;
    for (int i = 0; i < n_active; ++i) {
        compute_neighbors(s, i);
        compute_new_velocity(s, i, ps_xy, rs, n_src < 0 ? 0 : n_src);
    }
    for (int i = 0; i < n_active; ++i) {
        s->vx[i] = s->newvx[i];
        s->vy[i] = s->newvy[i];
        s->px[i] += s->vx[i] * s->time_step;
        s->py[i] += s->vy[i] * s->time_step;
    }
}
