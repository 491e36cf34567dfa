/*
 * oracle.h -- TEST INFRASTRUCTURE.  CPU oracle of the img_env step() path: a literal,
 * single-threaded C restatement of the reference algorithm.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product package never does.
 *
 * It consumes the same imgenv_cfg / imgenv_reset_batch structs as the product C ABI
 * (include/imgenv.h) but every pointer -- inputs, actions, outputs -- is a HOST pointer.
 */
#ifndef ORACLE_H_
#define ORACLE_H_

#include "../include/imgenv.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_world oracle_world;

const char* oracle_last_error(void);
int oracle_create(const imgenv_cfg* cfg, const uint8_t* static_map, int32_t Hg, int32_t Wg,
                  oracle_world** out);
void oracle_destroy(oracle_world* w);
int oracle_reset(oracle_world* w, const imgenv_reset_batch* batch);
int oracle_step(oracle_world* w, const float* actions);
int oracle_step_begin(oracle_world* w, const float* actions);
int oracle_step_end(oracle_world* w);
int oracle_records(oracle_world* w, double** records, int64_t* bytes_per_robot);
int oracle_outputs(oracle_world* w, imgenv_out* out); /* host pointers */
int oracle_pedinfo(oracle_world* w, float** pedinfo); /* [n_local][n_peds][5] px py vx vy r_ */
/* class layer as robot `i` (world index) sees it: peds_map + other robots (img_env.cpp:620-629) */
int oracle_private_grid(oracle_world* w, int32_t robot, uint8_t* dst);
/* obs_map_ / peds_map_ (img_env.h:43-46) */
int oracle_grids(oracle_world* w, const uint8_t** obs_map, const uint8_t** peds_map);

/* unit-test hooks: Agent::bresenhamLine (agent.cpp:511-624) and the planar tf operations of tfmath.h on hand-made inputs */
double oracle_test_bresenham(int x1, int y1, int x2, int y2, const uint8_t* src, uint8_t* dst, int Hv, int Wv, double res);
void oracle_test_tf(int op, const double* in, double* out);
void oracle_test_corners(int shape, const double* sizes, double x, double y, double yaw, double* out);
/* the first n values of glibc's rand() after srand(seed), from the oracle's restatement (the beep lottery's stream) */
void oracle_test_glibc_rand(unsigned int seed, int n, int32_t* out);
/* the social-force crowd's quadtree as a digest that does not depend on node numbering (oracle_sfm.c: sfm_tree_digest) */
int oracle_sfm_tree(oracle_world* w, uint64_t* out8);

#ifdef __cplusplus
}
#endif
#endif
