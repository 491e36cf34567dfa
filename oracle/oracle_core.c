/*
 * oracle_core.c -- TEST INFRASTRUCTURE (oracle), never linked into or loaded by the product.
 *
 * Literal CPU restatement of the per-robot step() path of DRL-Navigation/img_env.  Each
 * function cites the reference file:line it follows (paths relative to /root/reference).
 *
 *   src/img_env/src/agent.cpp      Agent / PedAgent (footprints, cmd, draw, view, bresenham, ...)
 *   src/img_env/src/grid_map.cpp   world2map / map2world / is_in_map / empty_map
 *   src/img_env/src/speed_limit.cpp SpeedLimiter
 *   src/img_env/src/img_env.cpp    ImgEnv::_init/_reset/_step/view_ped/view_robot/get_states
 *   src/img_env/src/rvoscene.h, ervoscene.h   scene adapters
 *   envs/env/yaml_env.py           ImageEnv._step_req/_get_states/_draw_ped_map/step
 *   envs/wrapper/base.py           SensorsPaperReward / TimeLimit / InfoLog / MultiRobotClean
 *
 * PARITY STATUS.  The ORCA pedestrian advance is pinned against the reference's own RVO2 sources and the social-force
 * advance (oracle_sfm.c) against its own pedsim sources (both built into oracle/_ref).  The Python post-processing (_draw_ped_map, _each_r, step_ds, dones) is pinned by
 * golden vectors generated from the reference's Python (tests/golden).  agent.cpp / img_env.cpp /
 * grid_map.cpp need ROS tf + OpenCV + generated message headers that this image lacks, so they
 * cannot be built here: for those rows this oracle is a careful restatement, PARITY UNPINNED
 * (the reference ships no tests or golden data for them).
 *
 * Deliberate resolutions of undefined behaviour in the reference (all flagged in DESIGN.md):
 *   - pedestrian yaw after the first step is an uninitialised local (img_env.cpp:346-349): 0 here.
 *   - PedAgent::arrive() reads trajectory_[cur_traj_index_] without the modulo that
 *     _get_cur_goal() applies (img_env.cpp:314 vs agent.cpp:841): an index past the end is
 *     treated as "not arrived" here.
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math, no -march=native FMA contraction).
 */
#define _GNU_SOURCE
#include "oracle.h"
#include "oracle_resize.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_rvo.h"
#include "oracle_sfm.h"
#include "tfmath.h"

static __thread char g_err[256];
const char* oracle_last_error(void) { return g_err; }
#define FAIL(code, ...)                              \
    do {                                             \
        snprintf(g_err, sizeof(g_err), __VA_ARGS__); \
        return (code);                               \
    } while (0)

typedef struct pts {
    int n;
    double *x, *y;
} pts;

typedef struct rclass { /* robot class = (shape, size, sensor_cfg) */
    int shape;
    float size[4];
    float sensor[2];
    double sizes[4]; /* promoted (img_env.cpp:120-123) */
    double sx, sy;   /* sensor_base_ (img_env.cpp:131-132) */
    pts bbox;
} rclass;

typedef struct pclass { /* ped class = (shape, size) */
    int shape;
    float size[6];
    double sizes[6];
    pts bbox;        /* circle / rectangle */
    pts left, right; /* leg */
} pclass;

/* ------------------------------------------------------------------ glibc rand() (third-party: glibc 2.31+ stdlib/random.c,
 * random_r.c -- TYPE_3, degree 31, separation 3, default seed 1), restated in the library's own front / rear pointer form.
 * img_env.cpp:327 draws rand() once per robot per step for the beep lottery.  One world = one node process = one stream.
 * Pinned against the container's real libc in tests/test_oracle_known_answers.py. */
typedef struct glibc_rand_state {
    int32_t tbl[31];
    int f, r; /* fptr, rptr as indices */
} glibc_rand_state;

static int32_t glibc_rand_next(glibc_rand_state* g) { /* __random_r, TYPE_3 branch */
    uint32_t val = (uint32_t)g->tbl[g->f] + (uint32_t)g->tbl[g->r];
    g->tbl[g->f] = (int32_t)val;
    int32_t result = (int32_t)(val >> 1);
    if (++g->f >= 31) {
        g->f = 0;
        ++g->r;
    } else if (++g->r >= 31) {
        g->r = 0;
    }
    return result;
}

static void glibc_srand(glibc_rand_state* g, unsigned int seed) { /* __srandom_r */
    if (seed == 0) seed = 1;
    g->tbl[0] = (int32_t)seed;
    int32_t word = (int32_t)seed;
    for (int i = 1; i < 31; i++) { /* 16807 * word % 2147483647 without overflow (Schrage) */
        long hi = word / 127773, lo = word % 127773;
        long t = 16807 * lo - 2836 * hi;
        if (t < 0) t += 2147483647;
        word = (int32_t)t;
        g->tbl[i] = word;
    }
    g->f = 3; /* fptr = &state[rand_sep], rptr = &state[0] */
    g->r = 0;
    for (int k = 0; k < 310; k++) (void)glibc_rand_next(g); /* kc = rand_deg * 10 discarded draws */
}

/* unit-test hook: the first n values of rand() after srand(seed) */
void oracle_test_glibc_rand(unsigned int seed, int n, int32_t* out) {
    glibc_rand_state g;
    glibc_srand(&g, seed);
    for (int i = 0; i < n; i++) out[i] = glibc_rand_next(&g);
}

struct oracle_world {
    imgenv_cfg cfg;
    int R, P, r0, r1, RL;
    int Hg, Wg, Hv, Wv, B, Hp, Wp, SD, PV;
    int img_h, img_w; /* sensor_map size after cv2.resize (yaml_env.py:431-438) */
    double res, step_hz, view_w, view_h, a_begin, a_end, min_d, max_d;
    uint8_t *static_map, *obs_map, *peds_map, *priv;
    uint32_t *own_lo, *own_hi;
    tf2d view_base, base_view;
    int n_rcls, n_pcls;
    rclass* rcls;
    pclass* pcls;
    int *robot_cls, *ped_cls;
    double* robot_size_last;
    imgenv_limiter lim_v, lim_w;
    /* world-sized robot records: x y theta vx vy pad */
    double* rec;
    /* local robots */
    double *gx, *gy, *l0v, *l0w, *l1v, *l1w;
    tf2d* world_target;
    int32_t* is_coll;
    uint8_t *is_arr, *py_done, *clean_state;
    uint8_t* view;
    double* hits;
    double *hit_x, *hit_y, *amap; /* hit_points_x_, hit_points_y_, angular_map_ (agent.h:65-67) */
    /* peds */
    double *ppx, *ppy, *pyaw, *plx, *ply, *pvx, *pvy, *prem, *llx, *lly, *rlx, *rly, *pr_round;
    int *pstate, *ptraj_idx, *ptraj_len;
    double* ptraj;
    double* ptraj_v; /* [P][traj_cap][2] Agent.msg trajectory_v, dataset scene */
    int traj_cap;
    float* pmax_speed;
    rvo_sim* rvo;
    sfm_scene* sfm;
    glibc_rand_state lottery; /* the node process's rand() stream as far as the beep lottery consumes it (img_env.cpp:327) */
    int has_reset;
    /* python-side state */
    double* tmp_dist;
    int have_tmp;
    int elapsed;
    /* outputs */
    imgenv_out out;
    uint16_t f16_lut[256];
    float* pedinfo; /* [RL][P][5] AgentState.pedinfo (px, py, vx, vy, r_) in pedestrian order, pre-sort */
};

/* ------------------------------------------------------------------ helpers */

/* GridMap::world2map (grid_map.cpp:40-44): C round(), half away from zero */
static inline int w2m(double v, double res) { return (int)round(v / res); }

static uint16_t f32_to_f16(float f) { /* round-to-nearest-even, as numpy's astype('float16') */
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t hm = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t hm = m >> 13;
    uint32_t rem = m & 0x1fffu;
    uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | hm);
    if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) h++;
    return h;
}

/* Python float floor division a // b (CPython floatobject.c float_floor_div/_float_div_mod) */
static double py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod) {
        if ((wx < 0) != (mod < 0)) {
            mod += wx;
            div -= 1.0;
        }
    }
    double floordiv;
    if (div) {
        floordiv = floor(div);
        if (div - floordiv > 0.5) floordiv += 1.0;
    } else {
        floordiv = copysign(0.0, vx / wx);
    }
    return floordiv;
}

/* Python round(x, 2): correctly rounded decimal (float.__round__ -> dtoa mode 3) */
static double py_round2(double x) {
    char buf[64];
    snprintf(buf, sizeof(buf), "%.2f", x);
    return strtod(buf, NULL);
}

/* Agent::init_shape_circle (agent.cpp:18-30) */
static pts shape_circle(double cx, double cy, double r) {
    double resolution = 0.01;
    int bb = (int)ceil(r / resolution);
    pts p;
    p.n = 0;
    p.x = (double*)malloc(sizeof(double) * (size_t)(2 * bb + 1) * (size_t)(2 * bb + 1));
    p.y = (double*)malloc(sizeof(double) * (size_t)(2 * bb + 1) * (size_t)(2 * bb + 1));
    for (int m = -bb; m <= bb; m++)
        for (int n = -bb; n <= bb; n++) {
            if (sqrt(m * resolution * m * resolution + n * resolution * n * resolution) <= r) {
                p.x[p.n] = m * resolution + cx;
                p.y[p.n] = n * resolution + cy;
                p.n++;
            }
        }
    return p;
}

/* Agent::init_shape_rectangle (agent.cpp:51-62) */
static pts shape_rectangle(const double* s) {
    double resolution = 0.01;
    int x_min = (int)floor(s[0] / resolution);
    int x_max = (int)ceil(s[1] / resolution);
    int y_min = (int)floor(s[2] / resolution);
    int y_max = (int)ceil(s[3] / resolution);
    pts p;
    p.n = 0;
    size_t cap = (size_t)(x_max - x_min + 1) * (size_t)(y_max - y_min + 1);
    if ((long)cap < 1) cap = 1;
    p.x = (double*)malloc(sizeof(double) * cap);
    p.y = (double*)malloc(sizeof(double) * cap);
    for (int m = x_min; m <= x_max; m++)
        for (int n = y_min; n <= y_max; n++) {
            p.x[p.n] = m * resolution;
            p.y[p.n] = n * resolution;
            p.n++;
        }
    return p;
}

static void free_pts(pts* p) {
    free(p->x);
    free(p->y);
    p->x = p->y = NULL;
    p->n = 0;
}

/* Agent::draw on the world grid (agent.cpp:285-327), frame "world_map" */
static int draw_world(uint8_t* grid, int Hg, int Wg, double res, const tf2d* base_world, const pts* bbox,
                      int value) {
    int is_collision = 0;
    for (int k = 0; k < bbox->n; k++) {
        double wx, wy;
        tf_apply(base_world, bbox->x[k], bbox->y[k], &wx, &wy);
        int m = w2m(wx, res), n = w2m(wy, res);
        if (m >= 0 && m < Hg && n >= 0 && n < Wg) {
            uint8_t c = grid[(size_t)m * Wg + n];
            if (c == 0)
                is_collision = 1;
            else if (c == 1)
                is_collision = 2;
            else if (c == 2)
                is_collision = 3;
            else if (value >= 0)
                grid[(size_t)m * Wg + n] = (uint8_t)value;
        }
    }
    return is_collision;
}

/* Agent::get_corners (agent.cpp:626-651) */
static void get_corners(int shape, const double* sizes, const tf2d* base_world, double* pax, double* pay,
                        double* pbx, double* pby) {
    if (shape == IMGENV_SHAPE_CIRCLE) {
        tf_apply(base_world, sizes[0] - sizes[2], sizes[1] - sizes[2], pax, pay);
        tf_apply(base_world, sizes[0] + sizes[2], sizes[1] + sizes[2], pbx, pby);
    } else {
        tf_apply(base_world, sizes[0], sizes[2], pax, pay);
        tf_apply(base_world, sizes[1], sizes[3], pbx, pby);
    }
}

/* ------------------------------------------------------------------ speed limiter */
/* speed_limit.cpp:43-53 */
static inline double clampd(double x, double lo, double hi) { return fmin(fmax(lo, x), hi); }
static inline int signd(double x) { return x == 0 ? 0 : (int)(x / fabs(x)); }

typedef struct limiter {
    int has_v, has_a, has_j;
    double min_v, max_v, min_a, max_a, min_j, max_j;
} limiter;

/* SpeedLimiter(comn_pkg::SpeedLimiter) (speed_limit.cpp:56-65): max_jerk = msg.min_jerk and
 * min_jerk stays uninitialised in the reference; it is 0 here. */
static limiter limiter_from_msg(const imgenv_limiter* m) {
    limiter l;
    l.has_v = m->has_velocity_limits;
    l.has_a = m->has_acceleration_limits;
    l.has_j = m->has_jerk_limits;
    l.max_v = m->max_velocity;
    l.min_v = m->min_velocity;
    l.max_a = m->max_acceleration;
    l.min_a = m->min_acceleration;
    l.max_j = m->min_jerk;
    l.min_j = 0.0;
    return l;
}

/* SpeedLimiter::limit (speed_limit.cpp:92-173): jerk, acceleration, velocity */
static void limiter_limit(const limiter* l, double* v, double v0, double v1, double dt) {
    if (l->has_j) {
        const double dv = *v - v0;
        const double dv0 = v0 - v1;
        const double dt2 = 2. * dt * dt;
        const double da_min = l->min_j * dt2;
        const double da_max = l->max_j * dt2;
        const double da = clampd(dv - dv0, da_min, da_max);
        *v = v0 + dv0 + da;
    }
    if (l->has_a) {
        const double tmp = *v;
        const int v_sign = signd(*v);
        const int v0_sign = signd(v0);
        if (v_sign + v0_sign != 0) {
            const double dv_min = l->min_a * dt;
            const double dv_max = l->max_a * dt;
            double dv = *v - v0;
            const int dv_sign = signd(dv);
            if (dv_sign == v0_sign || dv_sign == v_sign)
                dv = dv_sign * clampd(fabs(dv), dv_min, dv_max);
            else
                dv = dv_sign * fabs(clampd(-fabs(dv), dv_min, dv_max));
            *v = v0 + dv;
        } else {
            const double zero_dt = fabs(v0 / l->min_a);
            if (zero_dt >= dt)
                *v = v0_sign * (fabs(v0) - fabs(l->min_a) * dt);
            else {
                const double v_dt = fabs(*v / l->max_a);
                if (zero_dt + v_dt >= dt)
                    *v = v_sign * fabs(l->max_a * (dt - zero_dt));
                else
                    *v = tmp;
            }
        }
    }
    if (l->has_v) *v = clampd(*v, l->min_v, l->max_v);
}

/* ------------------------------------------------------------------ create */

static int find_rclass(oracle_world* w, int shape, const float* size, const float* sensor) {
    for (int c = 0; c < w->n_rcls; c++) {
        rclass* k = &w->rcls[c];
        if (k->shape == shape && !memcmp(k->size, size, sizeof(float) * 4) &&
            !memcmp(k->sensor, sensor, sizeof(float) * 2))
            return c;
    }
    w->rcls = (rclass*)realloc(w->rcls, sizeof(rclass) * (size_t)(w->n_rcls + 1));
    rclass* k = &w->rcls[w->n_rcls];
    memset(k, 0, sizeof(*k));
    k->shape = shape;
    memcpy(k->size, size, sizeof(float) * 4);
    memcpy(k->sensor, sensor, sizeof(float) * 2);
    for (int j = 0; j < 4; j++) k->sizes[j] = (double)size[j];
    k->sx = (double)sensor[0];
    k->sy = (double)sensor[1];
    if (shape == IMGENV_SHAPE_CIRCLE)
        k->bbox = shape_circle(k->sizes[0], k->sizes[1], k->sizes[2]);
    else
        k->bbox = shape_rectangle(k->sizes);
    return w->n_rcls++;
}

static int find_pclass(oracle_world* w, int shape, const float* size) {
    for (int c = 0; c < w->n_pcls; c++) {
        pclass* k = &w->pcls[c];
        if (k->shape == shape && !memcmp(k->size, size, sizeof(float) * 6)) return c;
    }
    w->pcls = (pclass*)realloc(w->pcls, sizeof(pclass) * (size_t)(w->n_pcls + 1));
    pclass* k = &w->pcls[w->n_pcls];
    memset(k, 0, sizeof(*k));
    k->shape = shape;
    memcpy(k->size, size, sizeof(float) * 6);
    for (int j = 0; j < 6; j++) k->sizes[j] = (double)size[j];
    if (shape == IMGENV_SHAPE_LEG) { /* PedAgent::init_shape (agent.cpp:666-680) */
        k->left = shape_circle(0, 0, k->sizes[2]);
        k->right = shape_circle(0, 0, k->sizes[5]);
    } else if (shape == IMGENV_SHAPE_CIRCLE) {
        k->bbox = shape_circle(k->sizes[0], k->sizes[1], k->sizes[2]);
    } else {
        k->bbox = shape_rectangle(k->sizes);
    }
    return w->n_pcls++;
}

#define ALLOC(ptr, type, n) ptr = (type*)calloc((size_t)((n) > 0 ? (n) : 1), sizeof(type))

int oracle_create(const imgenv_cfg* cfg, const uint8_t* static_map, int32_t Hg, int32_t Wg,
                  oracle_world** out) {
    if (!cfg || !static_map || !out) FAIL(IMGENV_EINVAL, "null argument");
    if (cfg->abi_version != IMGENV_ABI_VERSION || cfg->struct_size != (int32_t)sizeof(imgenv_cfg))
        FAIL(IMGENV_EINVAL, "imgenv_cfg ABI mismatch (version %d size %d, want %d %d)", cfg->abi_version,
             cfg->struct_size, IMGENV_ABI_VERSION, (int)sizeof(imgenv_cfg));
    if (cfg->n_robots < 1 || cfg->n_peds < 0 || Hg < 1 || Wg < 1) FAIL(IMGENV_EINVAL, "bad sizes");
    if (cfg->n_worlds > 1) FAIL(IMGENV_EINVAL, "the oracle is one world: check a batched handle against n_worlds oracles");
    if (cfg->n_peds > cfg->max_ped)
        FAIL(IMGENV_EINVAL, "n_peds %d > max_ped %d (IndexError in yaml_env.py:401)", cfg->n_peds, cfg->max_ped);
    if (cfg->state_dim < 3 || cfg->state_dim > 5) FAIL(IMGENV_EINVAL, "state_dim must be 3, 4 or 5");
    if (cfg->ped_vec_dim != 7) FAIL(IMGENV_EINVAL, "ped_vec_dim must be 7");
    oracle_world* w = (oracle_world*)calloc(1, sizeof(oracle_world));
    w->cfg = *cfg;
    w->R = cfg->n_robots;
    w->P = cfg->n_peds;
    w->r0 = cfg->robot_begin;
    w->r1 = cfg->robot_end;
    if (w->r0 == 0 && w->r1 == 0) w->r1 = w->R;
    if (w->r0 < 0 || w->r1 > w->R || w->r0 >= w->r1) {
        free(w);
        FAIL(IMGENV_EINVAL, "bad robot shard [%d,%d)", cfg->robot_begin, cfg->robot_end);
    }
    w->RL = w->r1 - w->r0;
    if (cfg->ped_scene_type == IMGENV_SCENE_ERVO && cfg->beep_r > 0 && cfg->ped_ca_p > 0 && w->RL != w->R) {
        free(w);
        FAIL(IMGENV_EINVAL, "the beep lottery needs every robot's action: not available in a robot shard");
    }
    /* img_env.cpp:58-81: float32 request fields read into doubles */
    w->res = (double)cfg->view_resolution;
    /* GridMap::read_image (grid_map.cpp:28-38): the image is resized (INTER_LINEAR) to the view resolution */
    w->Hg = Hg;
    w->Wg = Wg;
    if (cfg->global_resolution != cfg->view_resolution) {
        const double resolution_ = (double)cfg->global_resolution;
        w->Wg = (int)(Wg * resolution_ / w->res);
        w->Hg = (int)(Hg * resolution_ / w->res);
        if (w->Hg < 1 || w->Wg < 1) {
            free(w);
            FAIL(IMGENV_EINVAL, "the map would be resized to %d x %d cells", w->Hg, w->Wg);
        }
    }
    w->view_w = (double)cfg->view_width;
    w->view_h = (double)cfg->view_height;
    w->step_hz = (double)cfg->step_hz;
    w->a_begin = (double)cfg->view_angle_begin;
    w->a_end = (double)cfg->view_angle_end;
    w->min_d = (double)cfg->view_min_dist;
    w->max_d = (double)cfg->view_max_dist;
    /* Agent::init_view_map (agent.cpp:79-90) */
    w->Wv = (int)(w->view_w / w->res);
    w->Hv = (int)(w->view_h / w->res);
    w->view_base.ox = w->view_h / 2;
    w->view_base.oy = w->view_w / 2;
    tf_set_yaw(&w->view_base, 3.14159);
    w->base_view = tf_inverse(&w->view_base);
    w->img_w = cfg->image_size[0]; /* cv2.resize(view, (image_size[0], image_size[1])): dsize = (width, height) */
    w->img_h = cfg->image_size[1];
    if (w->img_w < 1 || w->img_h < 1) {
        free(w);
        FAIL(IMGENV_EINVAL, "bad image_size");
    }
    w->B = cfg->use_laser ? cfg->range_total : 0;
    w->Hp = cfg->ped_image_size[0];
    w->Wp = cfg->ped_image_size[1];
    w->SD = cfg->state_dim;
    w->PV = 1 + cfg->ped_vec_dim * cfg->max_ped;

    size_t G = (size_t)w->Hg * w->Wg;
    w->static_map = (uint8_t*)malloc(G);
    if (w->Hg == Hg && w->Wg == Wg)
        memcpy(w->static_map, static_map, G);
    else
        oracle_resize_linear_u8(static_map, Hg, Wg, w->static_map, w->Hg, w->Wg);
    Hg = w->Hg;
    Wg = w->Wg;
    w->obs_map = (uint8_t*)malloc(G);
    memcpy(w->obs_map, w->static_map, G); /* (the resized map: the caller's buffer holds Hg x Wg cells of the SOURCE resolution) */
    w->peds_map = (uint8_t*)malloc(G);
    memcpy(w->peds_map, w->static_map, G);
    w->priv = (uint8_t*)malloc(G);
    w->own_lo = (uint32_t*)malloc(G * 4);
    w->own_hi = (uint32_t*)malloc(G * 4);

    ALLOC(w->robot_cls, int, w->R);
    ALLOC(w->robot_size_last, double, w->R);
    for (int i = 0; i < w->R; i++) {
        int shape = cfg->robot_shape[i];
        if (shape != IMGENV_SHAPE_CIRCLE && shape != IMGENV_SHAPE_RECTANGLE) {
            oracle_destroy(w);
            FAIL(IMGENV_EINVAL, "robot %d: unsupported shape %d", i, shape);
        }
        w->robot_cls[i] = find_rclass(w, shape, cfg->robot_size + 4 * i, cfg->robot_sensor_cfg + 2 * i);
        w->robot_size_last[i] = cfg->robot_size_last ? cfg->robot_size_last[i] : 0.0;
    }
    ALLOC(w->ped_cls, int, w->P);
    ALLOC(w->pmax_speed, float, w->P);
    ALLOC(w->pr_round, double, w->P);
    for (int j = 0; j < w->P; j++) {
        w->ped_cls[j] = find_pclass(w, cfg->ped_shape[j], cfg->ped_size + 6 * j);
        w->pmax_speed[j] = cfg->ped_max_speed[j];
        /* PedInfo.r_ = sizes_[2] as float32 (img_env.cpp:582); ped_r = round(rt.r_, 2) (yaml_env.py:405) */
        w->pr_round[j] = py_round2((double)(float)w->pcls[w->ped_cls[j]].sizes[2]);
    }
    w->lim_v = cfg->limiter_v;
    w->lim_w = cfg->limiter_w;
    /* deep copies are done; drop caller pointers */
    w->cfg.robot_shape = NULL; w->cfg.robot_size = NULL; w->cfg.robot_sensor_cfg = NULL;
    w->cfg.ped_shape = NULL; w->cfg.ped_size = NULL; w->cfg.ped_max_speed = NULL;
    w->cfg.robot_size_last = NULL;

    ALLOC(w->rec, double, (size_t)w->R * IMGENV_RECORD_DOUBLES);
    int RL = w->RL;
    ALLOC(w->gx, double, RL); ALLOC(w->gy, double, RL);
    ALLOC(w->l0v, double, RL); ALLOC(w->l0w, double, RL); ALLOC(w->l1v, double, RL); ALLOC(w->l1w, double, RL);
    ALLOC(w->world_target, tf2d, RL);
    ALLOC(w->is_coll, int32_t, RL); ALLOC(w->is_arr, uint8_t, RL); ALLOC(w->py_done, uint8_t, RL);
    ALLOC(w->clean_state, uint8_t, RL);
    ALLOC(w->view, uint8_t, (size_t)RL * w->Hv * w->Wv);
    ALLOC(w->hits, double, (size_t)RL * (w->B > 0 ? w->B : 1));
    ALLOC(w->hit_x, double, (size_t)RL * (w->B > 0 ? w->B : 1));
    ALLOC(w->hit_y, double, (size_t)RL * (w->B > 0 ? w->B : 1));
    ALLOC(w->amap, double, (size_t)RL * IMGENV_ANGULAR_BINS);
    int P = w->P;
    ALLOC(w->ppx, double, P); ALLOC(w->ppy, double, P); ALLOC(w->pyaw, double, P);
    ALLOC(w->plx, double, P); ALLOC(w->ply, double, P); ALLOC(w->pvx, double, P); ALLOC(w->pvy, double, P);
    ALLOC(w->prem, double, P); ALLOC(w->llx, double, P); ALLOC(w->lly, double, P);
    ALLOC(w->rlx, double, P); ALLOC(w->rly, double, P);
    ALLOC(w->pstate, int, P); ALLOC(w->ptraj_idx, int, P); ALLOC(w->ptraj_len, int, P);
    ALLOC(w->tmp_dist, double, RL);

    /* ImgEnv::_init (img_env.cpp:83-103): scene + addPed + addRobot */
    if (cfg->ped_scene_type == IMGENV_SCENE_RVO || cfg->ped_scene_type == IMGENV_SCENE_ERVO) {
        w->rvo = rvo_create((float)w->step_hz); /* setTimeStep(step_hz_) rvoscene.h:13-16 */
        glibc_srand(&w->lottery, 1); /* a fresh node process: rand() was never seeded */
        for (int j = 0; j < P; j++) /* rvoscene.h:53-58 */
            rvo_add_agent(w->rvo, 0.f, 0.f, 0.5f, 10, 5.f, 5.f, 0.5f, (float)(double)w->pmax_speed[j]);
        if (cfg->relation_ped_robo == 1)
            for (int i = 0; i < w->R; i++) /* rvoscene.h:60-66 */
                rvo_add_agent(w->rvo, 0.f, 0.f, 0.5f, 10, 5.f, 5.f, 0.5f, 0.6f);
    } else if (cfg->ped_scene_type == IMGENV_SCENE_PEDSIM) {
        sfm_reseed(); /* one handle = one fresh node process */
        w->sfm = sfm_create(P, cfg->relation_ped_robo == 1 ? w->R : 0, w->pmax_speed);
    }

    /* outputs (host) */
    imgenv_out* o = &w->out;
    o->struct_size = (int32_t)sizeof(imgenv_out);
    o->n_local = RL; o->view_h = w->Hv; o->view_w = w->Wv; o->n_beams = w->B; o->state_dim = w->SD;
    o->ped_vec_len = w->PV;
    o->image_h = w->img_h; o->image_w = w->img_w; o->grid_h = w->Hg; o->grid_w = w->Wg;
    size_t VW = (size_t)w->Hv * w->Wv;
    ALLOC(o->vector_states, float, (size_t)RL * w->SD);
    ALLOC(o->view_maps, uint8_t, RL * VW);
    ALLOC(o->sensor_maps, uint16_t, (size_t)RL * w->img_h * w->img_w);
    ALLOC(o->lasers_raw, float, (size_t)RL * (w->B > 0 ? w->B : 1));
    ALLOC(o->hits_x, float, (size_t)RL * (w->B > 0 ? w->B : 1));
    ALLOC(o->hits_y, float, (size_t)RL * (w->B > 0 ? w->B : 1));
    ALLOC(o->angular_map, float, (size_t)RL * IMGENV_ANGULAR_BINS);
    ALLOC(o->lasers, double, (size_t)RL * (w->B > 0 ? w->B : 1));
    ALLOC(o->ped_vector_states, float, (size_t)RL * w->PV);
    ALLOC(o->ped_maps, float, (size_t)RL * 3 * w->Hp * w->Wp);
    ALLOC(o->is_collisions, int8_t, RL); ALLOC(o->is_arrives, uint8_t, RL);
    ALLOC(o->step_ds, double, RL); ALLOC(o->ped_min_dists, double, RL);
    ALLOC(o->base_rewards, int32_t, RL); ALLOC(o->base_dones, uint8_t, RL);
    ALLOC(o->rewards, double, RL); ALLOC(o->paper_rewards, double, RL); ALLOC(o->dones, uint8_t, RL); ALLOC(o->dones_info, int32_t, RL);
    ALLOC(o->is_clean, uint8_t, RL);
    ALLOC(o->robot_pose, double, (size_t)RL * 3);
    ALLOC(o->ped_state, double, (size_t)(P > 0 ? P : 1) * 4); /* (the bindings view one zero row when there are no pedestrians) */
    ALLOC(o->counters, int32_t, 4);
    for (int l = 0; l < RL; l++) {
        o->ped_min_dists[l] = INFINITY; /* NearbyPed (reset_helper.py:85-99), never re-initialised */
        o->is_clean[l] = 1;
        w->clean_state[l] = 1;
    }
    /* numpy: uint8.astype('float16') / 255.0 evaluates in float32 and rounds to float16 */
    for (int v = 0; v < 256; v++) w->f16_lut[v] = f32_to_f16((float)v / 255.0f);
    ALLOC(w->pedinfo, float, (size_t)RL * P * 5);
    *out = w;
    return IMGENV_OK;
}

void oracle_destroy(oracle_world* w) {
    if (!w) return;
    for (int c = 0; c < w->n_rcls; c++) free_pts(&w->rcls[c].bbox);
    for (int c = 0; c < w->n_pcls; c++) {
        free_pts(&w->pcls[c].bbox);
        free_pts(&w->pcls[c].left);
        free_pts(&w->pcls[c].right);
    }
    free(w->rcls); free(w->pcls); free(w->robot_cls); free(w->ped_cls); free(w->robot_size_last);
    free(w->static_map); free(w->obs_map); free(w->peds_map); free(w->priv); free(w->own_lo); free(w->own_hi);
    free(w->rec); free(w->gx); free(w->gy); free(w->l0v); free(w->l0w); free(w->l1v); free(w->l1w);
    free(w->world_target); free(w->is_coll); free(w->is_arr); free(w->py_done); free(w->clean_state); free(w->view); free(w->hits); free(w->hit_x); free(w->hit_y); free(w->amap);
    free(w->ppx); free(w->ppy); free(w->pyaw); free(w->plx); free(w->ply); free(w->pvx); free(w->pvy);
    free(w->prem); free(w->llx); free(w->lly); free(w->rlx); free(w->rly); free(w->pr_round);
    free(w->pstate); free(w->ptraj_idx); free(w->ptraj_len); free(w->ptraj); free(w->ptraj_v); free(w->pmax_speed);
    free(w->tmp_dist);
    free(w->pedinfo);
    rvo_destroy(w->rvo);
    sfm_destroy(w->sfm);
    imgenv_out* o = &w->out;
    free(o->vector_states); free(o->view_maps); free(o->sensor_maps); free(o->lasers_raw); free(o->lasers); free(o->hits_x); free(o->hits_y); free(o->angular_map);
    free(o->ped_vector_states); free(o->ped_maps); free(o->is_collisions); free(o->is_arrives);
    free(o->step_ds); free(o->ped_min_dists); free(o->base_rewards); free(o->base_dones); free(o->rewards); free(o->paper_rewards);
    free(o->dones); free(o->dones_info); free(o->is_clean); free(o->robot_pose); free(o->ped_state);
    free(o->counters);
    free(w);
}

/* ------------------------------------------------------------------ rasters */

#define REC(w, i) ((w)->rec + (size_t)(i) * IMGENV_RECORD_DOUBLES)

/* ImgEnv::view_ped (img_env.cpp:594-618) incl. PedAgent::draw_leg (agent.cpp:737-774) */
static void view_ped(oracle_world* w) {
    size_t G = (size_t)w->Hg * w->Wg;
    memcpy(w->peds_map, w->obs_map, G);
    for (int j = 0; j < w->P; j++) {
        pclass* k = &w->pcls[w->ped_cls[j]];
        tf2d bw = tf_from_pose(w->ppx[j], w->ppy[j], w->pyaw[j]);
        if (k->shape == IMGENV_SHAPE_CIRCLE) {
            draw_world(w->peds_map, w->Hg, w->Wg, w->res, &bw, &k->bbox, 1);
        } else if (k->shape == IMGENV_SHAPE_LEG) {
            for (int leg = 0; leg < 2; leg++) {
                const pts* b = leg == 0 ? &k->left : &k->right;
                /* get_leg_base: identity rotation (0,0,0,1), origin = leg (agent.cpp:815-821) */
                tf2d lb;
                tf_set_rotation_zw(&lb, 0.0, 1.0);
                lb.ox = leg == 0 ? w->llx[j] : w->rlx[j];
                lb.oy = leg == 0 ? w->lly[j] : w->rly[j];
                for (int q = 0; q < b->n; q++) {
                    double bx, by, wx, wy;
                    tf_apply(&lb, b->x[q], b->y[q], &bx, &by);
                    tf_apply(&bw, bx, by, &wx, &wy);
                    int m = w2m(wx, w->res), n = w2m(wy, w->res);
                    if (m >= 0 && m < w->Hg && n >= 0 && n < w->Wg) {
                        uint8_t* c = &w->peds_map[(size_t)m * w->Wg + n];
                        if (leg == 0) { /* left leg skips obstacle cells (agent.cpp:751) */
                            if (*c != 0) *c = 1;
                        } else { /* right leg skips pedestrian cells only (agent.cpp:767) */
                            if (*c != 1) *c = 1;
                        }
                    }
                }
            }
        }
        /* rectangle peds are not drawn by view_ped (img_env.cpp:599-616) */
    }
}

/* shared robot-owner layer: min / max index of the robots whose footprint covers a cell */
static void raster_robot_owners(oracle_world* w) {
    size_t G = (size_t)w->Hg * w->Wg;
    memset(w->own_lo, 0xff, G * 4);
    memset(w->own_hi, 0, G * 4);
    for (int j = 0; j < w->R; j++) {
        const double* r = REC(w, j);
        tf2d bw = tf_from_pose(r[0], r[1], r[2]);
        const pts* b = &w->rcls[w->robot_cls[j]].bbox;
        for (int k = 0; k < b->n; k++) {
            double wx, wy;
            tf_apply(&bw, b->x[k], b->y[k], &wx, &wy);
            int m = w2m(wx, w->res), n = w2m(wy, w->res);
            if (m >= 0 && m < w->Hg && n >= 0 && n < w->Wg) {
                size_t c = (size_t)m * w->Wg + n;
                uint32_t id = (uint32_t)j + 1;
                if (w->own_lo[c] > id) w->own_lo[c] = id;
                if (w->own_hi[c] < id) w->own_hi[c] = id;
            }
        }
    }
}

/* value of a cell in robots_[i].global_map_ (img_env.cpp:623-628) from the shared layers */
static inline uint8_t shared_cell(const oracle_world* w, int i, size_t c) {
    uint8_t p = w->peds_map[c];
    if (p <= 2) return p;
    uint32_t id = (uint32_t)i + 1;
    if (w->own_hi[c] != 0 && (w->own_lo[c] != id || w->own_hi[c] != id)) return 2;
    return p;
}

/* literal per-robot grid: global_map_ = peds_map_; draw every other robot with 2 (img_env.cpp:620-629) */
static void build_private(oracle_world* w, int i, uint8_t* dst) {
    memcpy(dst, w->peds_map, (size_t)w->Hg * w->Wg);
    for (int j = 0; j < w->R; j++) {
        if (i == j) continue;
        const double* r = REC(w, j);
        tf2d bw = tf_from_pose(r[0], r[1], r[2]);
        draw_world(dst, w->Hg, w->Wg, w->res, &bw, &w->rcls[w->robot_cls[j]].bbox, 2);
    }
}

int oracle_private_grid(oracle_world* w, int32_t robot, uint8_t* dst) {
    if (!w || robot < 0 || robot >= w->R) FAIL(IMGENV_EINVAL, "bad robot");
    build_private(w, robot, dst);
    return IMGENV_OK;
}

int oracle_grids(oracle_world* w, const uint8_t** obs_map, const uint8_t** peds_map) {
    if (obs_map) *obs_map = w->obs_map;
    if (peds_map) *peds_map = w->peds_map;
    return IMGENV_OK;
}

int oracle_sfm_tree(oracle_world* w, uint64_t* out8) {
    if (!w || !w->sfm || !out8) return IMGENV_EINVAL;
    sfm_tree_digest(w->sfm, out8);
    return IMGENV_OK;
}


/* ------------------------------------------------------------------ Agent::view */

typedef struct gridview {
    const oracle_world* w;
    int robot;            /* world index */
    const uint8_t* priv;  /* non-NULL in private-grid mode */
} gridview;

static inline uint8_t gv_at(const gridview* g, int m, int n) {
    size_t c = (size_t)m * g->w->Wg + n;
    return g->priv ? g->priv[c] : shared_cell(g->w, g->robot, c);
}

/* Agent::bresenhamLine (agent.cpp:511-624).  view_res: both maps share the view resolution. */
static double bresenham(int x1, int y1, int x2, int y2, const uint8_t* src, uint8_t* dst, int Hv, int Wv,
                        double res) {
    double hit = 6;
    double x0w = x1 * res, y0w = y1 * res; /* map2world of the start cell */
    int wv = x2 - x1;
    int hv = y2 - y1;
    int dx = ((wv > 0) << 1) - 1;
    int dy = ((hv > 0) << 1) - 1;
    wv = abs(wv);
    hv = abs(hv);
    int f, y, x, delta1, delta2;
    int line_end = 0;
    int end_x = -1, end_y = -1;
    const int steep = !(wv > hv);
    if (!steep) {
        f = 2 * hv - wv;
        delta1 = 2 * hv;
        delta2 = (hv - wv) * 2;
    } else {
        f = 2 * wv - hv;
        delta1 = wv * 2;
        delta2 = (wv - hv) * 2;
    }
    for (x = x1, y = y1; steep ? (y != y2) : (x != x2);) {
        if (x >= 0 && x < Hv && y >= 0 && y < Wv) {
            int cur = src[x * Wv + y];
            if (!line_end) {
                if (cur != 0)
                    dst[x * Wv + y] = 255;
                else if (end_x == -1) {
                    dst[x * Wv + y] = 0;
                    line_end = 1;
                    end_x = x;
                    end_y = y;
                    double cx = x * res, cy = y * res;
                    hit = sqrt((x0w - cx) * (x0w - cx) + (y0w - cy) * (y0w - cy));
                }
            } else {
                if ((x != end_x) && (y != end_y)) dst[x * Wv + y] = 200;
            }
        } else
            return hit;
        if (f < 0) {
            f += delta1;
        } else {
            if (steep)
                x += dx;
            else
                y += dy;
            f += delta2;
        }
        if (steep)
            y += dy;
        else
            x += dx;
    }
    return hit;
}

/* Agent::view (agent.cpp:356-509) for local robot l */
static void agent_view(oracle_world* w, int l, const gridview* g) {
    const int i = w->r0 + l;
    if (w->is_coll[l] || w->is_arr[l]) {
        w->out.counters[2]++;
        return;
    }
    const rclass* k = &w->rcls[w->robot_cls[i]];
    const double* r = REC(w, i);
    const tf2d base_world = tf_from_pose(r[0], r[1], r[2]);
    const int Hv = w->Hv, Wv = w->Wv;
    const double res = w->res;
    /* is_collision_ = draw(grid_map, -1, "world_map", bbox_) (agent.cpp:361, 285-327) */
    int code = 0;
    for (int q = 0; q < k->bbox.n; q++) {
        double wx, wy;
        tf_apply(&base_world, k->bbox.x[q], k->bbox.y[q], &wx, &wy);
        int m = w2m(wx, res), n = w2m(wy, res);
        if (m >= 0 && m < w->Hg && n >= 0 && n < w->Wg) {
            uint8_t c = gv_at(g, m, n);
            if (c == 0) code = 1;
            else if (c == 1) code = 2;
            else if (c == 2) code = 3;
        }
    }
    w->is_coll[l] = code;
    /* sensor cell (agent.cpp:366-369) */
    double sx, sy;
    tf_apply(&w->base_view, k->sx, k->sy, &sx, &sy);
    const int x0 = w2m(sx, res), y0 = w2m(sy, res);
    uint8_t* view = w->view + (size_t)l * Hv * Wv;
    memset(view, 200, (size_t)Hv * Wv); /* empty_map (grid_map.cpp:57-60) */
    uint8_t* laser = (uint8_t*)malloc((size_t)Hv * Wv);
    memset(laser, 200, (size_t)Hv * Wv); /* GridMap laser_map = view_map_ (agent.cpp:371) */
    const tf2d view_world = tf_mul(&base_world, &w->view_base); /* get_view_world (agent.cpp:128-131) */
    for (int a = 0; a < Hv; a++) {
        for (int b = 0; b < Wv; b++) {
            double xv = a * res, yv = b * res; /* map2world */
            double xb, yb;
            tf_apply(&w->view_base, xv, yv, &xb, &yb);
            double ang = atan2(yb - k->sy, xb - k->sx);
            if (ang <= w->a_begin || ang >= w->a_end || xb < w->min_d || xb > w->max_d) continue;
            double wx, wy;
            tf_apply(&view_world, xv, yv, &wx, &wy);
            int m = w2m(wx, res), n = w2m(wy, res);
            if (m >= 0 && m < w->Hg && n >= 0 && n < w->Wg) {
                uint8_t c = gv_at(g, m, n);
                view[a * Wv + b] = (c < 250) ? 0 : 255;
            }
        }
    }
    if (w->cfg.use_laser) {
        double map_width = w->base_view.ox;
        double map_height = w->base_view.oy;
        double max_range = sqrt(map_width * map_width + map_height * map_height);
        double angle_step = fabs(w->a_end - w->a_begin) / w->cfg.range_total;
        double* hits = w->hits + (size_t)l * w->B;
        /* angular_map_, hit_points_x_ / _y_ (agent.cpp:407-436) */
        const int angular_map_size = IMGENV_ANGULAR_BINS;
        double* amap = w->amap + (size_t)l * angular_map_size;
        for (int m = 0; m < angular_map_size; m++) amap[m] = w->max_d;
        double angular_map_step = fabs(w->a_end - w->a_begin) / angular_map_size;
        for (int b = 0; b < w->cfg.range_total; b++) {
            double cur = w->a_begin + angle_step * b;
            int angular_map_i = (int)(angle_step * b / angular_map_step);
            double x = max_range * cos(cur);
            double y = max_range * sin(cur);
            double vx, vy;
            tf_apply(&w->base_view, x, y, &vx, &vy);
            int x2 = w2m(vx, res), y2 = w2m(vy, res);
            hits[b] = bresenham(x0, y0, x2, y2, view, laser, Hv, Wv, res);
            if (angular_map_i >= 0 && angular_map_i < angular_map_size && hits[b] < amap[angular_map_i]) amap[angular_map_i] = hits[b];
            w->hit_x[(size_t)l * w->B + b] = hits[b] * cos(cur);
            w->hit_y[(size_t)l * w->B + b] = hits[b] * sin(cur);
        }
        memcpy(view, laser, (size_t)Hv * Wv); /* view_map_ = laser_map (agent.cpp:437) */
    }
    free(laser);
    /* draw(view_map_, 100, "view_map", bbox_) (agent.cpp:503, 307-312) */
    for (int q = 0; q < k->bbox.n; q++) {
        double vx, vy;
        tf_apply(&w->base_view, k->bbox.x[q], k->bbox.y[q], &vx, &vy);
        int m = w2m(vx, res), n = w2m(vy, res);
        if (m >= 0 && m < Hv && n >= 0 && n < Wv) {
            uint8_t c = view[m * Wv + n];
            if (c != 0 && c != 1 && c != 2) view[m * Wv + n] = 100;
        }
    }
}

/* ImgEnv::view_agent (img_env.cpp:589-592) */
static void view_agent(oracle_world* w) {
    view_ped(w);
    if (w->cfg.flags & IMGENV_FLAG_PRIVATE_GRIDS) {
        for (int l = 0; l < w->RL; l++) {
            build_private(w, w->r0 + l, w->priv);
            gridview g = {w, w->r0 + l, w->priv};
            agent_view(w, l, &g);
        }
    } else {
        raster_robot_owners(w);
        for (int l = 0; l < w->RL; l++) {
            gridview g = {w, w->r0 + l, NULL};
            agent_view(w, l, &g);
        }
    }
}

/* ------------------------------------------------------------------ states */

typedef struct pedinfo {
    float px, py, vx, vy, r;
    double key;
    int idx;
} pedinfo;

static int cmp_pedinfo(const void* a, const void* b) { /* stable: ties by original index */
    const pedinfo* x = (const pedinfo*)a;
    const pedinfo* y = (const pedinfo*)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return x->idx - y->idx;
}

/* ImgEnv::get_states (img_env.cpp:547-587) + ImageEnv._get_states (yaml_env.py:446-481) */
static void get_states(oracle_world* w) {
    imgenv_out* o = &w->out;
    const int SD = w->SD, Hv = w->Hv, Wv = w->Wv, B = w->B, P = w->P, Hp = w->Hp, Wp = w->Wp;
    pedinfo* pi = (pedinfo*)malloc(sizeof(pedinfo) * (size_t)(P > 0 ? P : 1));
    const double presol = 6.0 / w->cfg.ped_image_size[0]; /* yaml_env.py:164 */
    const double pr = w->cfg.ped_image_r;
    const double pr2 = pow(pr, 2.0); /* self.ped_image_r ** 2 */
    for (int l = 0; l < w->RL; l++) {
        const int i = w->r0 + l;
        const double* r = REC(w, i);
        const tf2d base_world = tf_from_pose(r[0], r[1], r[2]);
        /* Agent::get_state (agent.cpp:156-184) */
        tf2d t = tf_mul(&w->world_target[l], &base_world);
        tf2d target_base = tf_inverse(&t);
        double st[5];
        int ns = 0;
        st[ns++] = target_base.ox;
        st[ns++] = target_base.oy;
        if (SD == 3) {
            st[ns++] = tf_basis_yaw_via_quaternion(&target_base);
        } else if (SD == 4) {
            st[ns++] = w->l0v[l];
            st[ns++] = w->l0w[l];
        } else {
            st[ns++] = tf_basis_yaw_via_quaternion(&target_base);
            st[ns++] = w->l0v[l];
            st[ns++] = w->l0w[l];
        }
        float* vs = o->vector_states + (size_t)l * SD;
        for (int q = 0; q < SD; q++) vs[q] = (float)st[q]; /* float32[] state */
        o->is_collisions[l] = (int8_t)w->is_coll[l];
        o->is_arrives[l] = w->is_arr[l];
        for (int b = 0; b < B; b++) {
            float h = (float)w->hits[(size_t)l * B + b]; /* float32[] laser */
            o->lasers_raw[(size_t)l * B + b] = h;
            o->lasers[(size_t)l * B + b] = w->cfg.laser_norm ? (double)h / w->cfg.laser_max : (double)h;
            o->hits_x[(size_t)l * B + b] = (float)w->hit_x[(size_t)l * B + b]; /* float32[] hits_x, hits_y (img_env.cpp:558-559) */
            o->hits_y[(size_t)l * B + b] = (float)w->hit_y[(size_t)l * B + b];
        }
        if (B > 0)
            for (int m = 0; m < IMGENV_ANGULAR_BINS; m++) /* float32[] angular_map (img_env.cpp:560) */
                o->angular_map[(size_t)l * IMGENV_ANGULAR_BINS + m] = (float)w->amap[(size_t)l * IMGENV_ANGULAR_BINS + m];
        const uint8_t* view = w->view + (size_t)l * Hv * Wv;
        memcpy(o->view_maps + (size_t)l * Hv * Wv, view, (size_t)Hv * Wv);
        /* _trans_cv2_sensor_map (yaml_env.py:431-438): cv2.resize INTER_CUBIC (a copy for equal sizes), float16, / 255 */
        {
            const size_t IS = (size_t)w->img_h * w->img_w;
            uint8_t* img = (uint8_t*)malloc(IS);
            oracle_resize_cubic_u8(view, Hv, Wv, img, w->img_h, w->img_w);
            for (size_t q = 0; q < IS; q++) o->sensor_maps[(size_t)l * IS + q] = w->f16_lut[img[q]];
            free(img);
        }
        o->robot_pose[3 * l] = r[0];
        o->robot_pose[3 * l + 1] = r[1];
        o->robot_pose[3 * l + 2] = r[2];
        /* PedInfo in the robot base frame (img_env.cpp:568-584) */
        tf2d world_base = tf_inverse(&base_world);
        for (int j = 0; j < P; j++) {
            double px, py;
            tf_apply(&world_base, w->ppx[j], w->ppy[j], &px, &py);
            double vx = (world_base.m00 * w->pvx[j] + world_base.m01 * w->pvy[j]) + 0.0;
            double vy = (world_base.m10 * w->pvx[j] + world_base.m11 * w->pvy[j]) + 0.0;
            pi[j].px = (float)px;
            pi[j].py = (float)py;
            pi[j].vx = (float)vx;
            pi[j].vy = (float)vy;
            pi[j].r = (float)w->pcls[w->ped_cls[j]].sizes[2];
            pi[j].key = pow((double)pi[j].px, 2.0) + pow((double)pi[j].py, 2.0); /* yaml_env.py:451 */
            pi[j].idx = j;
            float* dbg = w->pedinfo + ((size_t)l * P + j) * 5;
            dbg[0] = pi[j].px; dbg[1] = pi[j].py; dbg[2] = pi[j].vx; dbg[3] = pi[j].vy; dbg[4] = pi[j].r;
        }
        qsort(pi, (size_t)P, sizeof(pedinfo), cmp_pedinfo);
        /* _draw_ped_map (yaml_env.py:392-429) */
        float* pt = o->ped_vector_states + (size_t)l * w->PV;
        memset(pt, 0, sizeof(float) * (size_t)w->PV);
        float* pm = o->ped_maps + (size_t)l * 3 * Hp * Wp;
        memset(pm, 0, sizeof(float) * 3 * (size_t)Hp * Wp);
        pt[0] = (float)P;
        for (int j = 0; j < P; j++) {
            const pedinfo* rt = &pi[j];
            double dpx = rt->px, dpy = rt->py;
            pt[j * 7 + 1] = rt->px;
            pt[j * 7 + 2] = rt->py;
            pt[j * 7 + 3] = rt->vx;
            pt[j * 7 + 4] = rt->vy;
            double ped_r = w->pr_round[rt->idx];
            pt[j * 7 + 5] = (float)ped_r;
            pt[j * 7 + 6] = (float)(ped_r + w->robot_size_last[i]);
            pt[j * 7 + 7] = (float)sqrt(pow(dpx, 2.0) + pow(dpy, 2.0));
            if (dpx > 3 || dpx < -3 || dpy > 3 || dpy < -3) continue;
            double tmx = -dpx + 3, tmy = -dpy + 3;
            int ax = (int)py_floordiv(tmx - pr, presol), bx = (int)py_floordiv(tmx + pr, presol);
            int ay = (int)py_floordiv(tmy - pr, presol), by = (int)py_floordiv(tmy + pr, presol);
            for (int jj = ax; jj < bx; jj++)
                for (int kk = ay; kk < by; kk++) {
                    if (jj < 0 || jj >= Hp || kk < 0 || kk >= Wp) continue;
                    double d = pow((jj + 0.5) * presol - tmx, 2.0) + pow((kk + 0.5) * presol - tmy, 2.0);
                    if (d < pr2) {
                        pm[(0 * Hp + jj) * Wp + kk] = 1.0f;
                        pm[(1 * Hp + jj) * Wp + kk] = rt->vx;
                        pm[(2 * Hp + jj) * Wp + kk] = rt->vy;
                    }
                }
        }
        if (P != 0) /* nearby_ped.set(i, ped_tmp[7] - ped_tmp[6]) in float32 (yaml_env.py:455-456) */
            o->ped_min_dists[l] = (double)(float)(pt[7] - pt[6]);
        /* distances / step_ds (yaml_env.py:467-471) */
        double dist = sqrt(pow((double)vs[0], 2.0) + pow((double)vs[1], 2.0));
        o->step_ds[l] = w->have_tmp ? w->tmp_dist[l] - dist : 0.0;
        w->tmp_dist[l] = dist;
    }
    w->have_tmp = 1;
    for (int j = 0; j < P; j++) {
        o->ped_state[4 * j] = w->ppx[j];
        o->ped_state[4 * j + 1] = w->ppy[j];
        o->ped_state[4 * j + 2] = w->pvx[j];
        o->ped_state[4 * j + 3] = w->pvy[j];
    }
    free(pi);
}

/* ------------------------------------------------------------------ reset */

int oracle_reset(oracle_world* w, const imgenv_reset_batch* b) {
    if (!w || !b) FAIL(IMGENV_EINVAL, "null argument");
    if (b->struct_size != (int32_t)sizeof(imgenv_reset_batch)) FAIL(IMGENV_EINVAL, "reset batch ABI mismatch");
    size_t G = (size_t)w->Hg * w->Wg;
    /* ImgEnv::_reset (img_env.cpp:162-292) */
    if (w->rvo) rvo_clear_obstacles(w->rvo);
    if (w->sfm) sfm_clear_obstacles(w->sfm);
    memcpy(w->obs_map, w->static_map, G);
    for (int q = 0; q < b->n_obstacles; q++) {
        double sizes[4];
        for (int j = 0; j < 4; j++) sizes[j] = (double)b->obs_size[4 * q + j];
        pts bb = b->obs_shape[q] == IMGENV_SHAPE_CIRCLE ? shape_circle(sizes[0], sizes[1], sizes[2])
                                                          : shape_rectangle(sizes);
        const double* p = b->obs_pose + 4 * q;
        double yaw = tf_yaw_from_quaternion_zw(p[2], p[3]);
        tf2d bw = tf_from_pose(p[0], p[1], yaw);
        draw_world(w->obs_map, w->Hg, w->Wg, w->res, &bw, &bb, 0);
        free_pts(&bb);
        double pax, pay, pbx, pby;
        get_corners(b->obs_shape[q], sizes, &bw, &pax, &pay, &pbx, &pby);
        if (!b->ignore_obstacle) {
            if (w->rvo) { /* rvoscene.h:19-26 */
                float v[8] = {(float)pax, (float)pay, (float)pax, (float)pby,
                              (float)pbx, (float)pby, (float)pbx, (float)pay};
                rvo_add_obstacle(w->rvo, v, 4);
            }
            if (w->sfm) sfm_add_obstacle(w->sfm, pax, pay, pbx, pby); /* pedscene.h:22-26 */
        }
    }
    /* pedestrians (img_env.cpp:220-250) */
    if (b->ped_traj_cap > w->traj_cap) {
        w->traj_cap = b->ped_traj_cap;
        w->ptraj = (double*)realloc(w->ptraj, sizeof(double) * 3 * (size_t)w->traj_cap * (size_t)(w->P > 0 ? w->P : 1));
        w->ptraj_v = (double*)realloc(w->ptraj_v, sizeof(double) * 2 * (size_t)w->traj_cap * (size_t)(w->P > 0 ? w->P : 1));
    }
    if (w->cfg.ped_scene_type == IMGENV_SCENE_DATASET && w->P > 0 && !b->ped_traj_v) return IMGENV_EINVAL;
    for (int j = 0; j < w->P; j++) {
        const double* p = b->ped_pose + 4 * j;
        double yaw = tf_yaw_from_quaternion_zw(p[2], p[3]);
        /* init_pose (agent.cpp:133-142): robot_pose_ = pose; last_robot_pose_ untouched */
        w->ppx[j] = p[0];
        w->ppy[j] = p[1];
        w->pyaw[j] = yaw;
        int len = b->ped_traj_len[j];
        w->ptraj_len[j] = len;
        for (int q = 0; q < len; q++)
            memcpy(w->ptraj + ((size_t)j * w->traj_cap + q) * 3, b->ped_traj + ((size_t)j * b->ped_traj_cap + q) * 3,
                   sizeof(double) * 3);
        if (w->cfg.ped_scene_type == IMGENV_SCENE_DATASET) /* trajectory_v (img_env.cpp:246-247) */
            for (int q = 0; q < len; q++)
                memcpy(w->ptraj_v + ((size_t)j * w->traj_cap + q) * 2, b->ped_traj_v + ((size_t)j * b->ped_traj_cap + q) * 2,
                       sizeof(double) * 2);
        w->ptraj_idx[j] = 0;
        if (w->rvo) { /* setPedPos (rvoscene.h:32-34) */
            w->rvo->px[j] = (float)p[0];
            w->rvo->py[j] = (float)p[1];
        }
        if (w->sfm) { /* setPedPos + setWayPoint (pedscene.h:34-46) */
            sfm_set_ped_pos(w->sfm, j, p[0], p[1]);
            sfm_set_waypoints(w->sfm, j, b->ped_goal[2 * j], b->ped_goal[2 * j + 1],
                              w->ptraj + (size_t)j * w->traj_cap * 3, len);
        }
    }
    /* robots (img_env.cpp:252-282) */
    for (int i = 0; i < w->R; i++) {
        const double* p = b->robot_pose + 4 * i;
        double yaw = tf_yaw_from_quaternion_zw(p[2], p[3]);
        double* r = REC(w, i);
        r[0] = p[0];
        r[1] = p[1];
        r[2] = yaw;
        /* vx, vy of the Agent persist across resets; the scene gets v = 0 (img_env.cpp:279-280) */
        if (w->rvo && w->cfg.relation_ped_robo == 1) {
            int a = w->P + i;
            w->rvo->px[a] = (float)p[0];
            w->rvo->py[a] = (float)p[1];
            w->rvo->vx[a] = 0.0f;
            w->rvo->vy[a] = 0.0f;
        }
        if (w->sfm && w->cfg.relation_ped_robo == 1) sfm_set_robot_pos(w->sfm, i, p[0], p[1]);
        if (i >= w->r0 && i < w->r1) {
            int l = i - w->r0;
            w->l0v[l] = 0;
            w->l0w[l] = 0; /* init_pose resets last0_vw_ only (agent.cpp:137) */
            /* set_goal (agent.cpp:144-154) */
            w->gx[l] = b->robot_goal[2 * i];
            w->gy[l] = b->robot_goal[2 * i + 1];
            tf2d tw = tf_from_pose(w->gx[l], w->gy[l], yaw);
            w->world_target[l] = tf_inverse(&tw);
            w->is_coll[l] = 0;
            w->is_arr[l] = 0;
            w->py_done[l] = 0;       /* self.dones = zeros (yaml_env.py:316) */
            w->clean_state[l] = 1;   /* MultiRobotCleanWrapper.reset (base.py:90-93) */
            w->out.is_clean[l] = 1;
            w->out.base_rewards[l] = 0;
            w->out.base_dones[l] = 0;
            w->out.rewards[l] = 0;
            w->out.paper_rewards[l] = 0;
            w->out.dones[l] = 0;
            w->out.dones_info[l] = 0;
        }
    }
    if (w->rvo) rvo_process_obstacles(w->rvo);
    w->have_tmp = 0; /* self.tmp_distances = None (yaml_env.py:225) */
    w->elapsed = 0;  /* TimeLimitWrapper.reset (base.py:229-231) */
    w->out.counters[0] = 0;
    w->out.counters[1] = 0;
    w->out.counters[2] = 0; /* frozen views since this reset */
    w->has_reset = 1;
    view_agent(w);
    get_states(w);
    return IMGENV_OK;
}

/* ------------------------------------------------------------------ step */

/* PedAgent::update_bbox (agent.cpp:696-735); step_len_ = 0.3 (2-arg ctor, agent.cpp:659-664) */
static void ped_update_bbox(oracle_world* w, int j) {
    const pclass* k = &w->pcls[w->ped_cls[j]];
    if (k->shape != IMGENV_SHAPE_LEG) return;
    const double step_len = 0.3;
    double move = sqrt((w->ppx[j] - w->plx[j]) * (w->ppx[j] - w->plx[j]) +
                       (w->ppy[j] - w->ply[j]) * (w->ppy[j] - w->ply[j]));
    int last = w->pstate[j];
    int st = (int)((move + w->prem[j]) / step_len + last);
    w->prem[j] = move + w->prem[j] - (st - last) * step_len;
    st %= 7;
    w->pstate[j] = st;
    if (st == 0 || st == 4) {
        w->llx[j] = k->sizes[0];
        w->lly[j] = k->sizes[1];
        w->rlx[j] = k->sizes[3];
        w->rly[j] = k->sizes[4];
    } else if (st == 1 || st == 3) {
        w->llx[j] = -step_len / 2;
        w->rlx[j] = step_len / 2;
    } else if (st == 2) {
        w->llx[j] = -step_len;
        w->rlx[j] = step_len;
    } else if (st == 5) {
        w->llx[j] = step_len / 2;
        w->rlx[j] = -step_len / 2;
    } else if (st == 6) {
        w->llx[j] = step_len;
        w->rlx[j] = -step_len;
    }
}

/* ImgEnv::_step_ped_dataset (img_env.cpp:361-386): pedestrians replay recorded positions / velocities, one record
 * per step since the reset (step_ is reset at img_env.cpp:164 and incremented at the end of _step, :518) */
static void step_ped_dataset(oracle_world* w) {
    for (int j = 0; j < w->P; j++) {
        const int len = w->ptraj_len[j];
        const int idx = w->elapsed >= len ? len - 1 : w->elapsed;
        const double* tp = w->ptraj + ((size_t)j * w->traj_cap + idx) * 3;
        const double* tv = w->ptraj_v + ((size_t)j * w->traj_cap + idx) * 2;
        const double vx = tv[0], vy = tv[1];
        w->plx[j] = w->ppx[j]; /* set_position (agent.cpp:691-694) */
        w->ply[j] = w->ppy[j];
        w->ppx[j] = tp[0];
        w->ppy[j] = tp[1];
        w->pyaw[j] = atan2(vy, vx);
        w->pvx[j] = vx;
        w->pvy[j] = vy;
        ped_update_bbox(w, j);
    }
}

/* ImgEnv::_step_ped_normal (img_env.cpp:304-359); actions = this step's request (local robots) */
static void step_ped(oracle_world* w, const float* actions) {
    const int P = w->P;
    if (w->cfg.ped_scene_type == IMGENV_SCENE_DATASET) { /* ImgEnv::_step_ped (img_env.cpp:294-302) */
        step_ped_dataset(w);
        return;
    }
    if (P == 0 && !w->rvo && !w->sfm) return;
    if (w->rvo) {
        for (int j = 0; j < P; j++) {
            /* waypoint logic (img_env.cpp:314-317, agent.cpp:823-829, 839-843) */
            int idx = w->ptraj_idx[j], len = w->ptraj_len[j];
            if (idx < len) {
                const double* tp = w->ptraj + ((size_t)j * w->traj_cap + idx) * 3;
                if ((tp[0] - w->ppx[j]) * (tp[0] - w->ppx[j]) + (tp[1] - w->ppy[j]) * (tp[1] - w->ppy[j]) < 0.04)
                    w->ptraj_idx[j] = ++idx;
            }
            const double* g = w->ptraj + ((size_t)j * w->traj_cap + (len > 0 ? idx % len : 0)) * 3;
            /* RVOScene::step (rvoscene.h:36-46) */
            float gvx = (float)g[0] - w->rvo->px[j];
            float gvy = (float)g[1] - w->rvo->py[j];
            if (gvx * gvx + gvy * gvy > 1.0f) {
                const float inv = 1.0f / sqrtf(gvx * gvx + gvy * gvy);
                gvx = gvx * inv;
                gvy = gvy * inv;
            }
            w->rvo->prefx[j] = gvx;
            w->rvo->prefy[j] = gvy;
        }
        /* beep lottery (img_env.cpp:323-342), one rand() per robot of the request: with probability ped_ca_p a robot whose
         * v_y (the action's `beep`, yaml_env.py:326; 0 for dead robots, 328-331) is positive becomes a source at its pose --
         * robot_pose_ as the PREVIOUS step left it, _step_ped runs before _step_robot (img_env.cpp:423-424) -- with radius
         * beep_r; everybody else contributes ((0,0), 0), for which addEvacVelocity returns early.  Through the reference's Python
         * API both parameters arrive as 0 (yaml_env.py:183-200 never forwards them) and the lottery never fires. */
        if (w->cfg.ped_scene_type == IMGENV_SCENE_ERVO) {
            float* ps = (float*)calloc((size_t)w->R * 2, sizeof(float));
            float* rs = (float*)calloc((size_t)w->R, sizeof(float));
            const double ped_ca_p = (double)w->cfg.ped_ca_p, beep_r = (double)w->cfg.beep_r;
            for (int j = 0; j < w->R; j++) {
                if (glibc_rand_next(&w->lottery) / (double)2147483647 < ped_ca_p) {
                    const int l = j - w->r0;
                    const double beep_radius = (l >= 0 && l < w->RL && w->py_done[l] == 0) ? (double)actions[3 * l + 2] : 0.0;
                    if (beep_radius > 0) {
                        ps[2 * j] = (float)REC(w, j)[0]; /* RVO::Vector2(rpose.x, rpose.y) */
                        ps[2 * j + 1] = (float)REC(w, j)[1];
                        rs[j] = (float)beep_r;
                    }
                }
            }
            rvo_do_step(w->rvo, w->rvo->n_agents, ps, rs, w->R);
            free(ps);
            free(rs);
        } else {
            rvo_do_step(w->rvo, w->rvo->n_agents, NULL, NULL, -1);
        }
    } else if (w->sfm) {
        sfm_move_agents(w->sfm, w->step_hz); /* PedScene::step (pedscene.h:48-50) */
    }
    for (int j = 0; j < P; j++) { /* img_env.cpp:344-358 */
        double nx = w->ppx[j], ny = w->ppy[j], vx = 0, vy = 0;
        if (w->rvo) { /* getNewPosAndVel (rvoscene.h:72-82) */
            nx = (double)w->rvo->px[j];
            ny = (double)w->rvo->py[j];
            vx = (double)w->rvo->vx[j];
            vy = (double)w->rvo->vy[j];
        } else if (w->sfm) { /* pedscene.h:82-91 */
            sfm_get_ped(w->sfm, j, &nx, &ny, &vx, &vy);
        } else {
            continue; /* EmptyScene::getNewPosAndVel leaves everything untouched... */
        }
        w->plx[j] = w->ppx[j]; /* set_position (agent.cpp:691-694) */
        w->ply[j] = w->ppy[j];
        w->ppx[j] = nx;
        w->ppy[j] = ny;
        w->pyaw[j] = 0.0; /* uninitialised `yaw` local in the reference */
        w->pvx[j] = vx;
        w->pvy[j] = vy;
        ped_update_bbox(w, j);
    }
}

/* Agent::cmd (agent.cpp:186-283) for local robot l */
static void agent_cmd(oracle_world* w, int l, double v, double wv, double v_y) {
    const int i = w->r0 + l;
    double* r = REC(w, i);
    const double step_hz = w->step_hz;
    const double control_hz = 0.05; /* agent.cpp:89 */
    limiter lv = limiter_from_msg(&w->lim_v), lw = limiter_from_msg(&w->lim_w);
    limiter_limit(&lv, &v, w->l0v[l], w->l1v[l], step_hz);
    limiter_limit(&lw, &wv, w->l0w[l], w->l1w[l], step_hz);
    w->l1v[l] = w->l0v[l];
    w->l1w[l] = w->l0w[l];
    w->l0v[l] = v;
    w->l0w[l] = wv;
    int is_arrive = 0;
    const double gx = w->gx[l], gy = w->gy[l];
    double ox = r[0], oy = r[1], oz = r[2];
    double cur_control = 0;
    const int omni = w->cfg.robot_ktype == IMGENV_KTYPE_OMNI;
    while (cur_control <= step_hz) {
        if (!omni) {
            ox += v * control_hz * cos(oz);
            oy += v * control_hz * sin(oz);
            r[3] = v * cos(oz);
            r[4] = v * sin(oz);
        } else {
            ox += v * control_hz * cos(oz) - v_y * control_hz * sin(oz);
            oy += v * control_hz * sin(oz) + v_y * control_hz * cos(oz);
        }
        oz += wv * control_hz;
        double cur_dist = sqrt((ox - gx) * (ox - gx) + (oy - gy) * (oy - gy));
        if (cur_dist <= 0.3) {
            is_arrive = 1;
            break;
        }
        cur_control += control_hz;
    }
    double theta = r[2];
    double dt = step_hz;
    if (wv == 0) {
        if (!omni) {
            r[0] += v * dt * cos(theta);
            r[1] += v * dt * sin(theta);
        } else {
            r[0] += v * dt * cos(theta) - v_y * dt * sin(theta);
            r[1] += v * dt * sin(theta) + v_y * dt * cos(theta);
        }
        r[2] += wv * dt;
    } else {
        double vw = v / wv;
        r[0] += -vw * sin(theta) + vw * sin(theta + wv * dt);
        r[1] += vw * cos(theta) - vw * cos(theta + wv * dt);
        if (omni) {
            double v_yw = v_y / wv;
            r[0] += -v_yw * cos(theta) + v_yw * cos(theta + wv * dt);
            r[1] += -v_yw * sin(theta) + v_yw * sin(theta + wv * dt);
        }
        r[2] += wv * dt;
    }
    double cur_dist = sqrt((r[0] - gx) * (r[0] - gx) + (r[1] - gy) * (r[1] - gy));
    if (cur_dist <= 0.3) is_arrive = 1;
    w->is_arr[l] = (uint8_t)is_arrive;
}

int oracle_step_begin(oracle_world* w, const float* actions) {
    if (!w || !actions) FAIL(IMGENV_EINVAL, "null argument");
    if (!w->has_reset) FAIL(IMGENV_ESTATE, "step before reset");
    /* ImgEnv::_step (img_env.cpp:421-425) */
    step_ped(w, actions);
    /* _step_req (yaml_env.py:319-331) + _step_robot (img_env.cpp:388-419) */
    for (int l = 0; l < w->RL; l++) {
        if (w->py_done[l] == 0) {
            /* float32 msg fields promoted to double (Agent.msg:8-10) */
            agent_cmd(w, l, (double)actions[3 * l], (double)actions[3 * l + 1], (double)actions[3 * l + 2]);
        }
    }
    return IMGENV_OK;
}

/* wrapper stack: TimeLimit, SensorsPaperReward, InfoLog, MultiRobotClean (base.py) */
static void wrappers(oracle_world* w) {
    imgenv_out* o = &w->out;
    w->elapsed += 1; /* base.py:224 */
    int ndone = 0;
    for (int l = 0; l < w->RL; l++) {
        /* ImageEnv.step (yaml_env.py:372-377) */
        int coll = o->is_collisions[l];
        int arr = o->is_arrives[l];
        o->base_rewards[l] = arr - coll;
        int d = (coll < -1 ? -1 : (coll > 1 ? 1 : coll)) + arr;
        d = d < 0 ? 0 : (d > 1 ? 1 : d);
        o->base_dones[l] = (uint8_t)d;
        w->py_done[l] = (uint8_t)d;
        /* TimeLimitWrapper.step (base.py:222-227) */
        int timeout = w->elapsed > w->cfg.time_max;
        int done = timeout ? 1 : d;
        int info = timeout ? 10 : 0;
        /* SensorsPaperRewardWrapper._each_r (base.py:164-188) */
        double collision_reward = 0, reach_reward = 0, step_reward = 0, distance_reward = 0, beep_reward = 0;
        double min_dist = o->ped_min_dists[l];
        if (min_dist <= w->cfg.ped_safety_space) collision_reward = -50 * (w->cfg.ped_safety_space - min_dist);
        if (coll > 0) {
            collision_reward = -500;
        } else {
            const float* vs = o->vector_states + (size_t)l * w->SD;
            double dd = sqrt(pow((double)vs[0], 2.0) + pow((double)vs[1], 2.0));
            if (dd < 0.3 || arr) {
                reach_reward = 500.0;
            } else {
                distance_reward = o->step_ds[l] * 200;
                step_reward = -5;
            }
        }
        double reward = collision_reward + reach_reward + step_reward + distance_reward + beep_reward;
        /* InfoLogWrapper.step (base.py:241-254) */
        if (coll > 0) info = coll;
        if (arr == 1) info = 5;
        /* MultiRobotCleanWrapper.step (base.py:79-88): mask uses is_clean from BEFORE this step */
        o->paper_rewards[l] = reward;
        uint8_t clean_before = w->clean_state[l];
        if (!clean_before) reward = 0;
        o->rewards[l] = reward;
        o->dones[l] = (uint8_t)done;
        o->dones_info[l] = info;
        /* info['is_clean'] = the mask used for this step; the state is updated for the next one */
        o->is_clean[l] = clean_before;
        w->clean_state[l] = done > 0 ? 0 : clean_before;
        if (done > 0) ndone++;
    }
    o->counters[0] = w->elapsed;
    o->counters[1] = ndone;
}

int oracle_step_end(oracle_world* w) {
    if (!w) FAIL(IMGENV_EINVAL, "null argument");
    /* _step_robot tail: setRobotPos for every robot (img_env.cpp:411-417) */
    if (w->cfg.relation_ped_robo == 1) {
        for (int i = 0; i < w->R; i++) {
            const double* r = REC(w, i);
            if (w->rvo) { /* rvoscene.h:47-51 */
                int a = w->P + i;
                w->rvo->px[a] = (float)r[0];
                w->rvo->py[a] = (float)r[1];
                w->rvo->vx[a] = (float)r[3];
                w->rvo->vy[a] = (float)r[4];
            }
            if (w->sfm) sfm_set_robot_pos(w->sfm, i, r[0], r[1]); /* pedscene.h:52-55 */
        }
    }
    view_agent(w);
    get_states(w);
    wrappers(w);
    return IMGENV_OK;
}

int oracle_step(oracle_world* w, const float* actions) {
    int rc = oracle_step_begin(w, actions);
    if (rc) return rc;
    return oracle_step_end(w);
}

int oracle_records(oracle_world* w, double** records, int64_t* bytes_per_robot) {
    if (!w) FAIL(IMGENV_EINVAL, "null argument");
    if (records) *records = w->rec;
    if (bytes_per_robot) *bytes_per_robot = IMGENV_RECORD_DOUBLES * (int64_t)sizeof(double);
    return IMGENV_OK;
}

/* AgentState.pedinfo of the local robots as the node would send it (img_env.cpp:568-584) */
int oracle_pedinfo(oracle_world* w, float** pedinfo) {
    if (!w || !pedinfo) FAIL(IMGENV_EINVAL, "null argument");
    *pedinfo = w->pedinfo;
    return IMGENV_OK;
}

int oracle_outputs(oracle_world* w, imgenv_out* out) {
    if (!w || !out) FAIL(IMGENV_EINVAL, "null argument");
    *out = w->out;
    return IMGENV_OK;
}

/* ------------------------------------------------------------------ unit-test hooks (tests/test_oracle_tf.py,
 * tests/test_oracle_known_answers.py): the static restatements above, callable one at a time on hand-made inputs */
double oracle_test_bresenham(int x1, int y1, int x2, int y2, const uint8_t* src, uint8_t* dst, int Hv, int Wv, double res) {
    return bresenham(x1, y1, x2, y2, src, dst, Hv, Wv, res);
}
/* Agent::get_corners (agent.cpp:626-651) of a footprint (shape, sizes[4]) at pose (x, y, yaw): out = pax, pay, pbx, pby */
void oracle_test_corners(int shape, const double* sizes, double x, double y, double yaw, double* out) {
    tf2d bw = tf_from_pose(x, y, yaw);
    get_corners(shape, sizes, &bw, &out[0], &out[1], &out[2], &out[3]);
}
/* op 0: from_pose(x, y, yaw) -> tf | 1: apply(tf[6], x, y) -> (x, y) | 2: inverse(tf) -> tf | 3: mul(tf a, tf b) -> tf
 * | 4: basis yaw via quaternion (tf) -> yaw | 5: yaw from quaternion (z, w) -> yaw | 6: set_rotation_zw(z, w) -> tf (origin 0) */
void oracle_test_tf(int op, const double* in, double* out) {
    tf2d a, b, r;
    memset(&r, 0, sizeof(r));
    if (op == 0) {
        r = tf_from_pose(in[0], in[1], in[2]);
    } else if (op == 1) {
        memcpy(&a, in, sizeof(a));
        tf_apply(&a, in[6], in[7], &out[0], &out[1]);
        return;
    } else if (op == 2) {
        memcpy(&a, in, sizeof(a));
        r = tf_inverse(&a);
    } else if (op == 3) {
        memcpy(&a, in, sizeof(a));
        memcpy(&b, in + 6, sizeof(b));
        r = tf_mul(&a, &b);
    } else if (op == 4) {
        memcpy(&a, in, sizeof(a));
        out[0] = tf_basis_yaw_via_quaternion(&a);
        return;
    } else if (op == 5) {
        out[0] = tf_yaw_from_quaternion_zw(in[0], in[1]);
        return;
    } else if (op == 6) {
        tf_set_rotation_zw(&r, in[0], in[1]);
    }
    memcpy(out, &r, sizeof(r));
}
