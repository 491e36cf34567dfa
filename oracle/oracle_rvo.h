/*
 * oracle_rvo.h -- TEST INFRASTRUCTURE (oracle).  C restatement of the RVO2 / ERVO library the
 * reference vendors under src/3rdparty/ervo_ros (float32 arithmetic throughout).
 * Pinned against the reference's own sources compiled unmodified (oracle/_ref/librvo_ref.so,
 * see oracle/Makefile and tests/test_oracle_rvo_ref.py).
 */
#ifndef ORACLE_RVO_H_
#define ORACLE_RVO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rvo_line {
    float px, py; /* point */
    float dx, dy; /* direction */
} rvo_line;

typedef struct rvo_obstacle {
    float px, py;
    float ux, uy; /* unitDir_ */
    int32_t is_convex;
    int32_t next, prev;
} rvo_obstacle;

typedef struct rvo_onode { /* KdTree::ObstacleTreeNode */
    int32_t obstacle;
    int32_t left, right; /* -1 = NULL */
} rvo_onode;

typedef struct rvo_anode { /* KdTree::AgentTreeNode */
    size_t begin, end, left, right;
    float maxX, maxY, minX, minY;
} rvo_anode;

typedef struct rvo_sim {
    float time_step;
    /* agents (SoA) */
    int32_t n_agents, cap_agents;
    float *px, *py, *vx, *vy, *prefx, *prefy, *newvx, *newvy;
    float *radius, *max_speed, *neighbor_dist, *time_horizon, *time_horizon_obst;
    int32_t* max_neighbors;
    /* obstacles */
    int32_t n_obst, cap_obst;
    rvo_obstacle* obst;
    int32_t n_onodes, cap_onodes;
    rvo_onode* onodes;
    int32_t oroot;
    /* agent kd-tree (KdTree.cpp:44-117); `order` is KdTree::agents_ and persists across steps */
    int32_t* order;
    int32_t n_order;
    rvo_anode* atree;
    /* scratch */
    rvo_line* lines;
    rvo_line* proj;
    int32_t cap_lines;
    float* on_dist;
    int32_t* on_idx;
    int32_t n_on, cap_on;
    float an_dist[64];
    int32_t an_idx[64];
    int32_t n_an;
} rvo_sim;

rvo_sim* rvo_create(float time_step);
void rvo_destroy(rvo_sim* s);
/* RVOSimulator::addAgent (RVOSimulator.cpp:108-128); velocity = (0,0) */
int rvo_add_agent(rvo_sim* s, float x, float y, float neighbor_dist, int max_neighbors,
                  float time_horizon, float time_horizon_obst, float radius, float max_speed);
/* RVOSimulator::addObstacle (RVOSimulator.cpp:130-170) */
int rvo_add_obstacle(rvo_sim* s, const float* xy, int n_vertices);
void rvo_clear_obstacles(rvo_sim* s);
/* RVOSimulator::processObstacles -> KdTree::buildObstacleTree (KdTree.cpp:119-257) */
void rvo_process_obstacles(rvo_sim* s);
/* RVOSimulator::doStep (RVOSimulator.cpp:180-199) / ERVOSimulator::doStep (ERVOSimulator.cpp:16-35).
 * n_active: agents [0,n_active) get computeNeighbors+computeNewVelocity+update, the rest are
 * only neighbours (pass n_agents for the literal behaviour).  ps/rs (n_src entries) are the
 * ERVO beep sources, n_src = -1 selects the plain RVO doStep. */
void rvo_do_step(rvo_sim* s, int n_active, const float* ps_xy, const float* rs, int n_src);
/* brute-force neighbour search instead of the agent kd-tree (what the HIP kernel does) */
void rvo_set_bruteforce(int on);

#ifdef __cplusplus
}
#endif
#endif
