/*
 * oracle_sfm.h -- TEST INFRASTRUCTURE (oracle).  C restatement of the libpedsim social-force
 * model the reference vendors under src/3rdparty/pedsimros (float64), driven the way
 * PedScene does (src/img_env/src/pedscene.h:17-91).
 */
#ifndef ORACLE_SFM_H_
#define ORACLE_SFM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfm_scene sfm_scene;

/* PedScene::addPed / addRobot (pedscene.h:57-80): n_peds Tagents with vmax = max_speed, then
 * n_robots Tagents (in the scene iff relation_ped_robo == 1, i.e. n_robots > 0 here). */
sfm_scene* sfm_create(int n_peds, int n_robots, const float* ped_max_speed);
void sfm_destroy(sfm_scene* s);
void sfm_clear_obstacles(sfm_scene* s);
void sfm_add_obstacle(sfm_scene* s, double ax, double ay, double bx, double by);
void sfm_set_ped_pos(sfm_scene* s, int j, double x, double y);
void sfm_set_waypoints(sfm_scene* s, int j, double gx, double gy, const double* traj_xyz, int n);
void sfm_set_robot_pos(sfm_scene* s, int i, double x, double y);
void sfm_move_agents(sfm_scene* s, double h);
void sfm_get_ped(const sfm_scene* s, int j, double* x, double* y, double* vx, double* vy);
/* agent idx (peds, then robots): p.x p.y p.z v.x v.y v.z */
void sfm_get_agent(const sfm_scene* s, int idx, double* out6);
double sfm_get_vmax(const sfm_scene* s, int idx);
/* test aid: node count, member entries, numbering-independent hashes of the leaves and of the treehash (oracle_sfm.c) */
void sfm_tree_digest(const sfm_scene* s, uint64_t* out8 /* [8] */);
/* restart the two process-global random streams (minstd_rand0 for vmax, glibc rand() for the tree positions) */
void sfm_reseed(void);
/* test-only: 1 = every atan2 of the model correctly rounded (libquadmath atan2q rounded once), 0 = the host libm's (default) */
void sfm_set_cr_atan2(int on);

#ifdef __cplusplus
}
#endif
#endif
