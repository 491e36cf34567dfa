/*
 * pedsim_ref_capi.cpp -- TEST INFRASTRUCTURE.  A C API over the REFERENCE's own libpedsim
 * (compiled from /root/reference/src/3rdparty/pedsimros/src/ped_*.cpp where those files lie, see
 * oracle/Makefile `ref`), driven the way the reference's PedScene adapter drives it
 * (src/img_env/src/pedscene.h:17-91): this restates only those ~60 adapter lines because
 * pedscene.h itself includes ROS headers that this image lacks.
 */
#include <cstdlib>
#include <vector>

#include "ped_agent.h"
#include "ped_obstacle.h"
#include "ped_scene.h"
#include "ped_waypoint.h"

struct PedRef {
    Ped::Tscene* scene;
    std::vector<Ped::Tagent*> peds, robots;
    std::vector<Ped::Tobstacle*> obs;
};

extern "C" {

void* pedref_create(int n_peds, int n_robots, int robots_in_scene, const float* ped_max_speed) {
    PedRef* r = new PedRef();
    r->scene = new Ped::Tscene(0, 10, 10, 10); /* pedscene.h:18 */
    for (int i = 0; i < n_peds; i++) {        /* addPed pedscene.h:57-69: rand() start positions (they decide the
                                                 quadtree leaves; setPedPos overwrites them before any step) */
        Ped::Tagent* a = new Ped::Tagent();
        double pxx = rand() / 2147483647.0 * 10.0;
        double pyy = rand() / 2147483647.0 * 10.0;
        a->setPosition(pxx, pyy, 0);
        a->setVmax((double)ped_max_speed[i]);
        r->scene->addAgent(a);
        r->peds.push_back(a);
    }
    for (int i = 0; i < n_robots; i++) { /* addRobot pedscene.h:71-80 */
        Ped::Tagent* a = new Ped::Tagent();
        if (robots_in_scene) r->scene->addAgent(a);
        r->robots.push_back(a);
    }
    return r;
}

void pedref_clear_obstacles(void* h) {
    PedRef* r = static_cast<PedRef*>(h);
    for (size_t i = 0; i < r->obs.size(); i++) r->scene->removeObstacle(r->obs[i]);
    r->obs.clear();
}

void pedref_add_obstacle(void* h, double ax, double ay, double bx, double by) {
    PedRef* r = static_cast<PedRef*>(h);
    Ped::Tobstacle* o = new Ped::Tobstacle(ax, ay, bx, by);
    r->scene->addObstacle(o);
    r->obs.push_back(o);
}

void pedref_set_ped_pos(void* h, int j, double x, double y) { static_cast<PedRef*>(h)->peds[j]->setPosition(x, y, 0); }

void pedref_set_waypoints(void* h, int j, double gx, double gy, const double* traj_xyz, int n) {
    PedRef* r = static_cast<PedRef*>(h);
    r->peds[j]->clearWaypoints();
    r->peds[j]->addWaypoint(new Ped::Twaypoint(gx, gy, 1));
    for (int k = 0; k < n; k++)
        r->peds[j]->addWaypoint(new Ped::Twaypoint(traj_xyz[3 * k], traj_xyz[3 * k + 1], traj_xyz[3 * k + 2]));
}

void pedref_set_robot_pos(void* h, int i, double x, double y) { static_cast<PedRef*>(h)->robots[i]->setPosition(x, y, 1); }

void pedref_move_agents(void* h, double dt) { static_cast<PedRef*>(h)->scene->moveAgents(dt); }

/* out: double[6] = p.x p.y p.z v.x v.y v.z */
void pedref_get_ped(void* h, int j, double* out) {
    Ped::Tagent* a = static_cast<PedRef*>(h)->peds[j];
    Ped::Tvector p = a->getPosition(), v = a->getVelocity();
    out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = v.x; out[4] = v.y; out[5] = v.z;
}
void pedref_get_robot(void* h, int i, double* out) {
    Ped::Tagent* a = static_cast<PedRef*>(h)->robots[i];
    Ped::Tvector p = a->getPosition(), v = a->getVelocity();
    out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = v.x; out[4] = v.y; out[5] = v.z;
}
double pedref_get_vmax(void* h, int is_robot, int i) {
    PedRef* r = static_cast<PedRef*>(h);
    return (is_robot ? r->robots[i] : r->peds[i])->getVmax();
}
}
