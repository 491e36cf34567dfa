// world.h -- device-side data layout of one img_env world (SoA in HBM) and the per-class static
// tables built once at imgenv_create().  See DESIGN.md "Data layout in HBM".
#pragma once
#include <stdint.h>

#include "fp_rows.h"
#include "sfm.h"
#include "tfm.h"

#define WAVE 64
#define OWNER_MULTI 0xFFFFFEu  // 24-bit owner field of the composed layer: several robots cover the cell
#define RC_INLINE 6           // distinct robot classes (shape, size, sensor) per world, carried in the kernel arguments
#define PC_INLINE 4           // distinct pedestrian classes per world
#define BEEP_T 256            // rand() values one round of k_beep produces
#define ORCA_NEAR_CAP 64       // robot agents a pedestrian can have within its 0.5 m neighbour range before k_orca falls back to the full scan

// composed class layer byte (k_compose): low 3 bits = base class, bit 3 = "some robot covers it"
#define CLS_STATIC 0   // occupancy value 0: static map / obstacle      (collision code 1)
#define CLS_PED 1      // value 1: pedestrian                           (collision code 2)
#define CLS_TWO 2      // value 2 already in the static map            (collision code 3)
#define CLS_LOW 3      // free for collision, occupied for the crop (3..249)
#define CLS_HIGH 4     // free (>= 250)
#define CLS_ROBOT 8    // flag: covered by >= 1 robot footprint; owner[] says which

// The same layer in STAMP mode (maps much larger than what their robots and pedestrians cover -- many small worlds): the
// rasters write their stamps straight onto `cell` and nothing is composed or cleared per step.
//   bits  0-2   base class of the obstacle map (written at reset, untouched by the rasters)
//   bits  3-4   what the rasters of ONE step put on the cell: nothing / one robot / several robots / a pedestrian
//   bits  5-12  the tag of that step (1..255, the step count modulo 255): a stamp with another tag has expired; every 255
//               steps one sweep drops the stale stamps before their tag comes round again
//   bits 13-31  the robot of a "one robot" stamp
#define STAMP_ONE 1u
#define STAMP_MANY 2u
#define STAMP_PED 3u
#define STAMP_KIND_SHIFT 3
#define STAMP_TAG_SHIFT 5
#define STAMP_OWNER_SHIFT 13
#define STAMP_MAX_ROBOTS (1 << 19)
#define STAMP_TAGS 255

// The same layer in SUM mode (round 5; dense worlds whose handle owns every robot): `cell` = the base class of the obstacle map
// plus COUNTS of what stands on the cell, kept up to date by the agents themselves with fire-and-forget atomic adds -- every robot
// and pedestrian remembers the cells it has added itself to (fp_cells / pd_cells) and, when it moves, subtracts itself from the
// cells it left and adds itself to the ones it entered (nothing at all while it covers the same cells: a robot turning in
// place, a frozen one, a pedestrian waiting at its goal).  No owner layers, no k_compose over every cell of every world per
// step, no tags to expire.  Layout (field widths per handle: sum_rc_shift, sum_id_shift, sum_pc_mask):
//   bits 0-2                      base class (written by the resets with read-modify-write: the counts stay)
//   bits 3 .. rc_shift - 1        pedestrians on the cell that view_ped would have drawn there (at least bits(peds per world))
//   bits rc_shift .. id_shift - 1 robots on the cell (at least 6 bits: 63 footprints on one cell)
//   bits id_shift .. 31           sum of their indices within the world, modulo: THE robot when the count is 1
// A robot adds (1 << rc_shift) + (index << id_shift); what it later subtracts is the same word, so the arithmetic is exact
// modulo 2^32 whatever the index sums do.  The view's crop still decides a cell with one gather and two compares: free iff the
// word is `CLS_HIGH` or `CLS_HIGH + this robot's own word`.

// Everything of a robot class that does not depend on the pose: footprint samples
// (agent.cpp:18-62), field-of-view mask of the crop (agent.cpp:373-386), Bresenham ray paths of
// the laser (agent.cpp:405-438, 511-624) and the own-footprint stamp (agent.cpp:503).
struct RobotClassDev {
    int n_fp;
    const double2* fp;           // [n_fp] footprint samples (x, y) in the base frame
    const uint32_t* fov_bits;    // [ceil(Hv*Wv/32)] bit = cell passes the angle / distance gate
    const uint32_t* stamp_bits;  // [ceil(Hv*Wv/32)] bit = own footprint covers the view cell
    int ray_maxlen;              // longest ray in cells
    int ray_stride;              // beams padded to a multiple of 64
    int ray_kpad;                // ray_maxlen padded to a multiple of 8
    const uint16_t* ray_rows;    // [ray_kpad / 8][ray_stride][8] view cell index of steps 8c..8c+7 of beam b: one 16-byte
                                 // chunk per lane, consecutive lanes contiguous; padding points at a free dummy cell
    const uint16_t* ray_len;     // [ray_stride] number of in-map steps before the ray leaves / ends
    const float* ray_dist;       // [ray_maxlen][ray_stride] float32(hit distance) if the hit is at step k
    const uint4* ray_fin;        // [ray_maxlen][ray_stride] everything k_view stores for a beam whose first hit is at step k, in ONE 16-byte load:
                                 // {ray_run, float bits of ray_dist, the `lasers` value as a double = (double)ray_dist / laser_max when laser_norm
                                 // (divided on the host: an fp64 division per beam less)}; null for big views
    const uint8_t* ray_run;      // [ray_maxlen][ray_stride] steps right behind step k that share its row or column (left alone by a hit at k)
    const uint32_t* inv_pack;    // [Hv*Wv] rays through a view cell: first entry | count << 20 ...
    const uint32_t* inv_ent;     // ... entries (beam << 16 | k), beam descending
    const uint32_t* top_ent;     // [Hv*Wv] first entry of each cell's list (highest beam) or B << 16 | 0xFFFF
    const uint32_t* dyn_groups;  // [n_dyn] the groups of 4 view cells a step can change (a beam crosses one of them), one word each: first cell c4 |
    int n_dyn;                   // field-of-view bits << 16 | own-footprint bits << 20: what k_view's crop and final pass walk in a step;
    const uint32_t* all_groups;  // [ceil(Hv*Wv / 4)] the same for every group: the pass of a reset (the other cells hold 200 / 100 for the whole episode)
    const uint2* inv_cell;       // [Hv*Wv] k_view's step (5): {block of the reach table | none << 13 | own footprint << 14 | smallest step << 24, inv_pack}
    int box_rad;                 // half-size (cells) of the LDS de-duplication box of the robot raster
    int n_rows;                  // lattice rows of fp (fp_rows.h): the rasters find the covered cells row by row instead of sample
    const FpRow* rows;           // by sample; 0: this class walks its samples (cells much smaller than the footprint)
    double fp_cy;
    // AgentState.hits_x / hits_y / angular_map (IMGENV_FLAG_AGENT_STATE_EXTRAS; null otherwise)
    const float *ray_hx, *ray_hy;   // [ray_maxlen + 1][ray_stride] float32(hit * cos / sin(beam angle)) for a hit at step k; last row: no hit
    const uint16_t* bin_start;      // [73] first beam of each angular_map bin
    int big;                     // views beyond k_view's 16 / 8-bit packing, or shrunk by cv2.resize (the shipped configs: 400 x
                                 // 400 cells, 1000 beams): the kernels of view_big.h and the tables of BigClassDev
};

// A robot class whose view goes through view_big.h.  The cropped view is a bitmap in 8 x 8 tiles (one wavefront crops one tile
// and stores its ballot): bit address of view cell (a, b) = ((a / 8) * tb + b / 8) * 64 + (a % 8) * 8 + b % 8.
struct BigClassDev {
    int ta, tb;                  // tiles per column / row
    int n_crop;                  // tiles that touch the field of view
    const uint32_t* crop_tiles;  // [n_crop] ta << 16 | tb
    const uint64_t* crop_masks;  // [n_crop] the tile's cells inside the field of view
    const uint32_t* cells;       // [ray_kpad / 4][ray_stride][4] bitmap bit of steps 4c..4c+3 of beam b as word byte address << 5 | bit (past the end: the free bit behind the bitmap)
    const uint16_t* ray_end;     // [ray_maxlen][ray_stride] last step behind (k, b) in the row or column of its cell
    const uint2* inv;            // [Hv*Wv] rays through a view cell: {first entry of inv_ent, count}
    // the 4 x 4 source cells of every pixel of a shrunk sensor_map (host_tables.h build_big_taps), [16][img_h * img_w] each
    const uint32_t* tap_top;     // the cell's top beam entry | own footprint << 31
    const uint2* tap_inv;        // its ray list {first entry of inv_ent, count}
    const uint32_t* tap_addr;    // its bit address in the crop bitmap
    const uint16_t* tap_chunks;  // [n_tap_chunks] the chunks of 256 pixels a step can change (some source cell is crossed by a beam): what k_taps_big covers
    int n_tap_chunks;            // in a step; a reset covers every chunk
};

struct PedClassDev {
    int shape;
    int n_bbox;  // circle / rectangle samples
    const double* bx;
    const double* by;
    int n_left, n_right;  // leg samples (circles of radius lr / rr around 0,0)
    const double *lx, *ly, *rx, *ry;
    double sizes[6];
    int box_rad;                             // SUM mode: half-size (cells) of the LDS box around the pedestrian's cell that holds its footprint in every gait state
    int n_brows, n_lrows, n_rrows;           // lattice rows (fp_rows.h) of the three sample lists; 0: walk the samples
    const FpRow *brows, *lrows, *rrows;
    double bbox_cy;
};

struct RvoObstDev {  // RVO::Obstacle (Obstacle.h) with index links
    float px, py, ux, uy;
    int is_convex, next, prev;
};
struct RvoNodeDev {  // KdTree::ObstacleTreeNode
    int obstacle, left, right;
};

struct DevWorld {
    // sizes
    int R, RL, r0, P, NA;  // world robots, local robots, first local robot, peds, RVO agents
    // Independent worlds in one handle (the reference's env_num idiom): W worlds of Rw robots and Pw pedestrians each, world-major
    // robot / pedestrian numbering, one copy of every grid layer per world, Gs cells apart (Gs = Hg Wg rounded up to 16 when W > 1).
    int W, Rw, Pw;
    // A launch covers everything (act_list == nullptr), or the robots / pedestrians / cells of the worlds listed in
    // act_list (a reset of some worlds while the others carry on).
    const int* act_list;  // [act_nw] worlds of this launch
    int act_nw;
    int act_nl, act_ng, act_np;  // local robots, world-wide robots (rasters, RVO records) and pedestrians of this launch
    // device-side auto-reset (spawn_device.h): the list's length lives in device memory, the launches are sized for every world
    // of the handle and blocks beyond the count leave at once (act_nw ... act_np then hold the launch's capacity)
    const int* act_n_dev;
    // the step's per-robot scalars (tail_group in kernels.h) are run by k_view / k_obs wavefronts themselves
    unsigned long long* tail_sig;  // [RL] exchange word: k_obs' min_dist (float bits) << 32 | k_view's collision code << 8 | who has been here (2 view, 1 obs)
    int* tail_cnt;                 // [ceil(RL / 64)][32] (one per 128-byte line) robots of the group whose two wavefronts have both been here
    int tail_fused;                // k_obs runs in this chain of launches (pedestrians exist); otherwise k_view alone hands over
    int tail_is_reset, tail_elapsed;
    int state_in_integrate;        // no pedestrians (no k_side_robots): k_integrate ends with Agent::get_state
    uint32_t Gs;
    size_t act_cells;  // grid cells of an everything-launch (k_compose, k_cell_base)
    const int* world_epoch;                              // [W] global step count at the world's last reset
    uint32_t stamp_tag;  // STAMP mode: tag of the stamps this launch writes and reads
    const int *obst_base, *node_base, *n_obst_w, *oroot_w;  // [W] slices of obst / onodes per world (W > 1)
    int Hg, Wg, Hv, Wv, B, Hp, Wp, SD, PV;
    int scene, relation, ktype, use_laser, laser_norm, time_max;
    double res, inv_res, step_hz, laser_max, ped_safety_space, ped_image_r, ped_image_r2, ped_res;
    double laser_out_nohit;  // the `lasers` value of a beam without a hit: 6.0 (agent.cpp:513), over laser_max when laser_norm
    double ped_inv_res;  // 1 / ped_res when ped_res is a power of two (v // ped_res == floor(v * ped_inv_res) exactly), else 0
    uint32_t wv_magic;  // ceil(2^32 / Wv): c / Wv == __umulhi(c, wv_magic) for c < 65536
    Tf2 view_base, base_view;
    // limiter (speed_limit.cpp)
    int lv_has_v, lv_has_a, lv_has_j, lw_has_v, lw_has_a, lw_has_j;
    double lv_min_v, lv_max_v, lv_min_a, lv_max_a, lv_min_j, lv_max_j;
    double lw_min_v, lw_max_v, lw_min_a, lw_max_a, lw_min_j, lw_max_j;
    // grids
    const uint8_t* obs_map;  // static + obstacles (rebuilt at reset)
    uint8_t* ped_layer;      // 1 where view_ped() would have written a 1 this step
    uint32_t* own_lo;        // min (robot index + 1) covering the cell, 0xFFFFFFFF = none
    uint32_t* own_hi;        // max (robot index + 1) covering the cell, 0 = none
    uint32_t* cell;          // composed layer: class byte | (owning robot or OWNER_MULTI) << 8, one gather per lookup
                             // (STAMP mode: base class | this step's stamp, and the three layers above do not exist)
    // "early observation" steps (imgenv_step on eligible handles, csrc/imgenv_hip.hip): k_obs starts with the step, beside the
    // move, instead of behind it.  It then works out its robot's new pose and the pedestrians' new positions ITSELF, from
    // snapshots nobody writes while it runs: rec_snap = every robot's record as the last k_view found it, ped_snap = every ORCA
    // pedestrian's position and freshly solved velocity as the last k_orca left them -- two buffers each, taken in turns: this
    // step's k_view / k_orca write while this step's k_obs may still read.  (Round 5 kept ONE rec_snap: an early k_obs that was
    // late -- another stream of the process keeping the chip busy -- then read the record its own step's k_view had already
    // replaced and advanced the robot twice; tests/test_gpu_stream_order.py.)
    double* rec_snap_out;        // [RL][IMGENV_RECORD_DOUBLES] where this chain's k_view writes; null: no snapshots
    double* rec_snap_out2;       // a chain over SOME worlds (a reset) writes both buffers
    const double* rec_snap_in;   // what an early k_obs reads
    float4* ped_snap_out;        // [P] (x, y, new vx, new vy): where this chain's k_orca writes; null: no snapshots
    float4* ped_snap_out2;       // a chain over SOME worlds (a reset) writes both buffers: the other worlds' entries stay where the next step reads
    const float4* ped_snap_in;   // [P] what an early k_obs reads
    int obs_early, obs_n_sub;    // this k_obs launch is an early one; sub-steps of Agent::cmd (the heading table's layout)
    // An early k_obs of a step whose actions may still be in the making on the caller's stream (plain imgenv_step) must not read
    // them before that stream has reached the step: the move's first workgroup stores the step's sequence number here -- the
    // caller's stream runs in order, so that store is the witness -- and a one-wavefront kernel in front of k_obs polls it (k_gate).
    // An event recorded in front of the move would say the same, and cost the caller's stream ~6 us per step.  [2], [3]: k_gate_probe.
    uint32_t* sync;
    const float* obs_actions;    // the step's actions
    int layer_sum;           // SUM mode (above): ped_layer / own_lo / own_hi do not exist, k_compose never runs
    // SUM mode in a ROBOT SHARD (round 6).  The rank's own robots keep their cell lists as everywhere; another rank's robot arrives
    // as its record, whose eighth double carries the cells under its footprint as a BITMAP over the (2 (box_rad - 2) + 1)^2 <= 64
    // cells around its centre cell -- made by its owner's raster, which runs in FRONT of the exchange -- and k_remote adds / takes
    // off the difference to the bitmap the robot left a step ago (rm_bits / rm_center): a thread per remote robot and no atomic at
    // all while its cells stay.  Local robots count with index l + 1 in the word's index field, remote ones with 0: "the only
    // robot on the cell is me" stays one compare.  No owner layers, no k_compose, no bounding box of the shard.
    int sum_shard;
    // k_view raises the issue priority of wavefronts that are behind (kernels.h: VIEW_PRIO): on handles whose observation is over when
    // the views start -- no pedestrians, or early-observation steps with an RVO crowd of at most 256 (headline, cfg-2).  Where k_obs or
    // the social-force chain run BESIDE the views (cfg-4, cfg-5) it takes the issue slots they need: cfg-4 108 -> 115, cfg-5 275 -> 281 us
    int view_prio;
    int act_g0;              // first robot (world-wide index) of a raster launch over the robots [act_g0, act_g0 + act_ng)
    int rm_rad;                   // the bitmaps' radius box_rad - 2 when every robot class has the same one, else -1
    unsigned long long* rm_bits;  // [R] the bitmap each robot of ANOTHER rank currently counts itself on ...
    int2* rm_center;              // [R] ... and the centre cell it is relative to
    uint32_t sum_rc_shift, sum_id_shift, sum_pc_mask;
    unsigned long long sum_wg_magic;  // ceil(2^40 / Wg): row of a cell index below 2^24
    uint32_t* pd_cells;      // [P][pd_cap] SUM mode: the cells each pedestrian has added itself to ...
    int* pd_n;               // [P] ... and their number
    int pd_cap, ped_box_cells;
    uint8_t* seg_tag;        // STAMP mode: [cells / 64] tag of the last step that stamped a cell of the segment
    // STAMP mode with big views (view_big.h): everything k_crop_big needs of a cell in ONE byte, in 8 x 8-cell blocks so that a
    // rotated 8 x 8 tile of the view touches 4-6 half cache lines instead of a dozen map rows: bit 7 = the obstacle map leaves
    // the cell free (>= 250), bits 0-6 = crop tag (stamp tag % 127 + 1) of the last stamp on it, 0 = none.  Byte of cell (m, n) of world k:
    // k * crop_ws + ((m / 8) * crop_wt + n / 8) * 64 + (m % 8) * 8 + n % 8.  Written wherever obs_map or a stamp is.
    uint8_t* crop_map;               // null: k_crop_big goes by obs_map + seg_tag
    unsigned long long crop_magic;   // ceil(2^40 / Wg): row of a cell index below 2^24 = (index * crop_magic) >> 40
    uint32_t crop_wt, crop_ws;       // blocks per row of blocks; bytes per world
    const uint8_t* static_crop;      // [crop_ws] crop_map of the static map (every reset starts from it)
    // class records travel by value in the kernel arguments: scalar loads, and their table pointers are known
    // to be global (no flat loads, no reloads after stores)
    RobotClassDev rc[RC_INLINE];
    PedClassDev pc[PC_INLINE];
    const PedClassDev* pc_mem;  // the same records in HBM, for per-lane (divergent) class lookups
    const RobotClassDev* rc_mem;
    // big views (view_big.h): class tables, and per local robot the cropped view as a tiled bitmap (plane 0: occupied, plane 1:
    // outside the map / the field of view; big_words 32-bit words each, the last ones stay zero) and the beams' hit words
    BigClassDev big[RC_INLINE];  // by value like rc[]: scalar loads, pointers known to be global
    uint32_t* big_bits;          // [RL][2][big_words]
    uint32_t* big_hit;           // [RL][big_hit_stride]: B hit words | the dummy beam | 1 = the view is redone by this chain of launches | its collision code
    int big_words, big_hit_stride, big_bits_in_lds;
    int keep_view_maps;          // view_maps (the full-size view) is an output; otherwise a shrunk view is never materialised
    const int* robot_cls;  // [R]
    const int* ped_cls;    // [P]
    const double* robot_size_last;  // [R]
    const double* ped_r_round;      // [P] round(PedInfo.r_, 2)
    const float* ped_r32;           // [P] PedInfo.r_
    const uint16_t* f16_lut;        // [256] float16(v / 255)
    // cv2.resize(view, image_size, INTER_CUBIC) (yaml_env.py:431-438): axis tables built on the host (csrc/cv_resize.h)
    int resize, img_h, img_w;       // resize: image_size differs from the view size
    const int *rs_xofs, *rs_yofs;   // [img_w], [img_h] source index of tap 1
    const short *rs_alpha, *rs_beta;  // [img_w][4], [img_h][4] 11-bit coefficients
    // robots
    double* rec;  // [R][6] x y theta vx vy pad
    double *gx, *gy, *l0v, *l0w, *l1v, *l1w;  // [RL]
    Tf2* world_target;                        // [RL] tf_world_target_ (agent.cpp:144-154)
    int* is_coll;                             // [RL]
    uint8_t *is_arr, *py_done, *clean_state;  // [RL]
    double* tmp_dist;                         // [RL]
    int obs_passes;                           // odd-even passes k_obs tries on last step's order before it sorts in full
    uint16_t* obs_ord;                        // [RL][64 E] last step's pedestrian order of each robot (k_obs starts from it: a few odd-even passes instead of a full sort); null: always sort
    uint16_t* pm_cells;                       // [RL][PM_CAP] ped_map cells written by the last step
    int* pm_n;                                // [RL] their count, -1 = unknown (dense clear needed)
    uint2* fp_cells;                          // [RL][fp_cap] grid cells under the footprint (cell, last sample index + 1), from k_raster
    int* fp_n;                                // [RL] their count, -1 = not available (k_view walks the samples itself)
    double* fp_pose;                          // [RL][3] pose (x, y, theta) the list was made for
    int fp_cap, box_cells;                    // list capacity; LDS box cells of k_raster (max over classes)
    int hit_stride;                           // max ray_stride over classes (LDS layout of k_view)
    // pedestrians
    double *ppx, *ppy, *pyaw, *plx, *ply, *pvx, *pvy, *prem, *llx, *lly, *rlx, *rly;  // [P]
    int *pstate, *ptraj_idx;                                                       // [P]
    const int* ptraj_len;
    const double* ptraj;  // [P][traj_cap][3]
    const double* ptraj_v;  // [P][traj_cap][3] dataset scene: recorded (vx, vy) and the host's atan2(vy, vx)
    int traj_cap;
    // RVO (float32)
    float *apx, *apy, *avx, *avy, *anvx, *anvy;  // [NA]
    const float* amax_speed;                     // [NA]
    const RvoObstDev* obst;
    const RvoNodeDev* onodes;
    int n_obst, n_onodes, oroot;
    int* near_n;     // [P] robot agents within neighborDist of pedestrian j this step (from k_side_robots) ...
    int* near_list;  // [P][ORCA_NEAR_CAP] ... their agent indices, in arrival order
    // ERVO beep lottery (img_env.cpp:323-342) and evacuation term (ervo_ros Agent.cpp:63-69, 430-432)
    int beep_on;                 // scene is ERVO, beep_r > 0 and ped_ca_p > 0: the lottery can fire
    float beep_r;                // rs_ of a source
    double ped_ca_p;
    uint32_t* beep_state;        // [W][31] the last 31 words of each world's rand() stream (glibc TYPE_3)
    const uint32_t* beep_coef;   // [BEEP_T][31] word k of the next BEEP_T as a combination of those 31 (the recurrence is linear)
    uint8_t* beep_flag;          // [R] robot is a beep source this step
    float2* beep_xy;             // [R] its position (float32, as RVO::Vector2) before this step's integrate
    SfmDev sfm;  // social-force crowd (pedscene)
    // robot-sharded worlds: this rank only needs the rasters under its own robots' views.  bbox = ordered-uint32 encoded
    // float min x, min y, max x, max y of the local robots' centres, accumulated by k_integrate / k_reset_robots
    int sharded, region_margin;
    uint32_t* bbox;
    int* err;  // [4] device-side overflow flags
    unsigned long long* prof;  // [16] per-phase cycle counters (IMGENV_PHASE_PROFILE builds)
    unsigned long long* dbg;   // [32] free-form debug marks (profile builds)
    // outputs (imgenv_out)
    float* vector_states;
    uint8_t* view_maps;
    uint16_t* sensor_maps;
    float* lasers_raw;
    double* lasers;
    float* ped_vector_states;
    float* ped_maps;
    int8_t* is_collisions;
    float *hits_x, *hits_y, *angular_map;  // AgentState's remaining fields, null unless asked for
    float view_max_dist32;                 // float32(view_max_dist_): what an angular_map bin holds where nothing was hit
    // the step's own scalars (imgenv_out.step_*): written by a step's tail beside the arrays a reset rewrites
    double* step_rewards; uint8_t* step_dones; int32_t* step_dones_info; uint8_t* step_is_clean; uint8_t* step_is_arrives; int8_t* step_is_collisions; uint8_t* step_all_down;
    int* finished;             // page-locked host memory: [0] number of worlds whose robots are all done, [1..] their indices (k_finished)
    uint8_t* is_arrives;
    double* step_ds;
    double* ped_min_dists;
    int32_t* base_rewards;
    uint8_t* base_dones;
    double* rewards;
    double* paper_rewards;
    uint8_t* dones;
    int32_t* dones_info;
    uint8_t* is_clean;
    double* robot_pose;
    double* ped_state;
    int32_t* counters;
};
