/* placeholder until the SFM restatement lands (see oracle_sfm.h) */
#include "oracle_sfm.h"
#include <stddef.h>
sfm_scene* sfm_create(int n_peds, int n_robots, const float* ped_max_speed) { return NULL; }
void sfm_destroy(sfm_scene* s) {}
void sfm_clear_obstacles(sfm_scene* s) {}
void sfm_add_obstacle(sfm_scene* s, double ax, double ay, double bx, double by) {}
void sfm_set_ped_pos(sfm_scene* s, int j, double x, double y) {}
void sfm_set_waypoints(sfm_scene* s, int j, double gx, double gy, const double* traj_xyz, int n) {}
void sfm_set_robot_pos(sfm_scene* s, int i, double x, double y) {}
void sfm_move_agents(sfm_scene* s, double h) {}
void sfm_get_ped(const sfm_scene* s, int j, double* x, double* y, double* vx, double* vy) {}
