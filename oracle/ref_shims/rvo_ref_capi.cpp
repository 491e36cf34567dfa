/*
 * rvo_ref_capi.cpp -- TEST INFRASTRUCTURE.  A C API over the REFERENCE's own RVO2 / ERVO
 * simulator, so the oracle's restatement can be checked against it.  This file is this repo's
 * code; the simulator it drives is compiled from /root/reference/src/3rdparty/ervo_ros/src/*.cpp
 * where those files lie (see oracle/Makefile `ref`), through the public RVOSimulator interface
 * only (include/ervo_ros/RVOSimulator.h, ERVOSimulator.h).
 */
#include <vector>

#include "ERVOSimulator.h"
#include "RVO.h"

extern "C" {

void* rvoref_create(float time_step) {
    RVO::ERVOSimulator* sim = new RVO::ERVOSimulator();
    sim->setTimeStep(time_step);
    return sim;
}

void rvoref_destroy(void* h) { delete static_cast<RVO::ERVOSimulator*>(h); }

int rvoref_add_agent(void* h, float x, float y, float neighbor_dist, int max_neighbors, float time_horizon,
                     float time_horizon_obst, float radius, float max_speed) {
    return (int)static_cast<RVO::ERVOSimulator*>(h)->addAgent(RVO::Vector2(x, y), neighbor_dist,
                                                             (size_t)max_neighbors, time_horizon,
                                                             time_horizon_obst, radius, max_speed);
}

int rvoref_add_obstacle(void* h, const float* xy, int n) {
    std::vector<RVO::Vector2> v;
    for (int i = 0; i < n; i++) v.push_back(RVO::Vector2(xy[2 * i], xy[2 * i + 1]));
    return (int)static_cast<RVO::ERVOSimulator*>(h)->addObstacle(v);
}

void rvoref_clear_obstacles(void* h) { static_cast<RVO::ERVOSimulator*>(h)->clearObstacle(); }
void rvoref_process_obstacles(void* h) { static_cast<RVO::ERVOSimulator*>(h)->processObstacles(); }

void rvoref_set_position(void* h, int i, float x, float y) {
    static_cast<RVO::ERVOSimulator*>(h)->setAgentPosition((size_t)i, RVO::Vector2(x, y));
}
void rvoref_set_velocity(void* h, int i, float x, float y) {
    static_cast<RVO::ERVOSimulator*>(h)->setAgentVelocity((size_t)i, RVO::Vector2(x, y));
}
void rvoref_set_pref_velocity(void* h, int i, float x, float y) {
    static_cast<RVO::ERVOSimulator*>(h)->setAgentPrefVelocity((size_t)i, RVO::Vector2(x, y));
}

/* n_src < 0: RVOSimulator::doStep(); otherwise ERVOSimulator::doStep(points, rs) */
void rvoref_do_step(void* h, const float* ps_xy, const float* rs, int n_src) {
    RVO::ERVOSimulator* sim = static_cast<RVO::ERVOSimulator*>(h);
    if (n_src < 0) {
        sim->RVOSimulator::doStep();
    } else {
        std::vector<RVO::Vector2> ps;
        std::vector<float> r;
        for (int i = 0; i < n_src; i++) {
            ps.push_back(RVO::Vector2(ps_xy[2 * i], ps_xy[2 * i + 1]));
            r.push_back(rs[i]);
        }
        sim->doStep(ps, r);
    }
}

/* out: float[n][4] = px, py, vx, vy */
void rvoref_get_state(void* h, float* out) {
    RVO::ERVOSimulator* sim = static_cast<RVO::ERVOSimulator*>(h);
    for (size_t i = 0; i < sim->getNumAgents(); i++) {
        out[4 * i] = sim->getAgentPosition(i).x();
        out[4 * i + 1] = sim->getAgentPosition(i).y();
        out[4 * i + 2] = sim->getAgentVelocity(i).x();
        out[4 * i + 3] = sim->getAgentVelocity(i).y();
    }
}

int rvoref_num_obstacle_vertices(void* h) {
    return (int)static_cast<RVO::ERVOSimulator*>(h)->getNumObstacleVertices();
}

/* neighbour lists of agent i after the last doStep (ids), returns counts */
int rvoref_agent_neighbors(void* h, int i, int* ids, int cap) {
    RVO::ERVOSimulator* sim = static_cast<RVO::ERVOSimulator*>(h);
    int n = (int)sim->getAgentNumAgentNeighbors((size_t)i);
    for (int k = 0; k < n && k < cap; k++) ids[k] = (int)sim->getAgentAgentNeighbor((size_t)i, (size_t)k);
    return n;
}
int rvoref_obstacle_neighbors(void* h, int i, int* ids, int cap) {
    RVO::ERVOSimulator* sim = static_cast<RVO::ERVOSimulator*>(h);
    int n = (int)sim->getAgentNumObstacleNeighbors((size_t)i);
    for (int k = 0; k < n && k < cap; k++) ids[k] = (int)sim->getAgentObstacleNeighbor((size_t)i, (size_t)k);
    return n;
}
}
