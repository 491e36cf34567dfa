/*
 * imgenv.h -- C ABI of the MI355X-native batched img_env step() path.
 *
 * This is the drop-in boundary.  In the reference the step() path sits behind four ROS
 * services advertised by EnvService (reference src/img_env/src/img_env.cpp:716-754):
 *
 *   ~init_image_env   (src/comn_pkg/srv/InitEnv.srv:1-21)   -> imgenv_create()
 *   ~reset_image_env  (src/comn_pkg/srv/ResetEnv.srv:1-8)   -> imgenv_reset()
 *   ~step_image_env   (src/comn_pkg/srv/StepEnv.srv:1-5)    -> imgenv_step()
 *   ~ep_end_image_env (src/comn_pkg/srv/EndEp.srv)          -> (episode recording: out of scope)
 *
 * and the Python post-processing of the response (reference envs/env/yaml_env.py:392-481,
 * envs/wrapper/base.py:153-254) which this library also performs on the device, so the
 * outputs are the fields of ImageState (envs/state/state.py:4-28) plus rewards/dones.
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t passed as void*
 *    (NULL = default stream).
 *  - every function returns 0 on success, a negative IMGENV_E* code on error; it never
 *    aborts.  imgenv_last_error() gives a thread-local message.  (The reference handlers
 *    always `return true`, img_env.cpp:726-754; Python raises on ServiceException,
 *    yaml_env.py:304-311, 367-370.)
 *  - all *input* pointers are HOST pointers and are copied during the call, except
 *    `actions` of the step functions and `records` (see below) which are DEVICE pointers.
 *  - all *output* pointers handed out by imgenv_outputs() are DEVICE pointers owned by the
 *    handle, valid until imgenv_destroy(); their contents are valid after the stream work of
 *    the last reset/step has completed and are overwritten by the next step.  They are the
 *    library's working copies, READ-ONLY for the caller: a step only rewrites what can change
 *    (a view cell no laser beam crosses holds its 200, or the footprint's 100, from the reset on;
 *    a frozen robot's rows keep their last values, agent.cpp:358-360), so whatever a caller
 *    wrote into them would stay there.  IMGENV_FLAG_CHECK_OUTPUTS detects such writes,
 *    IMGENV_FLAG_FULL_REWRITE hands out copies the caller may do anything to (below).
 *  - one handle is one world (one ImgEnv node, img_env.h:171-172) -- or imgenv_cfg.n_worlds of them,
 *    the reference's env_num nodes batched into one set of launches -- and is single-threaded
 *    like the node (ros::spin, img_env_node.cpp:8); distinct handles are independent.
 *  - fields that are float32 in the ROS messages are `float` here and are promoted to
 *    double inside exactly where the node reads them (img_env.cpp:58-81, 113-160), so the
 *    wire rounding of the reference is reproduced by construction.
 */
#ifndef IMGENV_H_
#define IMGENV_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMGENV_ABI_VERSION 2  /* 2: imgenv_out grew (image / grid sizes, step_* arrays), spawn structs, IMGENV_FLAG_NO_VIEW_MAPS */

/* error codes */
#define IMGENV_OK 0
#define IMGENV_EINVAL (-1)    /* bad argument / unsupported configuration */
#define IMGENV_ENOMEM (-2)    /* host or device allocation failed */
#define IMGENV_EDEVICE (-3)   /* HIP runtime error (no device, launch failure, ...) */
#define IMGENV_ESTATE (-4)    /* call out of order (step before reset, ...) */

/* Agent.msg `shape` (src/comn_pkg/msg/Agent.msg:5; agent.cpp:64-77, 666-685) */
#define IMGENV_SHAPE_CIRCLE 0    /* size = [cx, cy, r]            agent.cpp:18-30  */
#define IMGENV_SHAPE_RECTANGLE 1 /* size = [xmin, xmax, ymin, ymax] agent.cpp:51-62 */
#define IMGENV_SHAPE_LEG 2       /* size = [lx, ly, lr, rx, ry, rr] agent.cpp:666-680 */

/* Env.msg `ped_scene_type` (scenefactory.h:8-24) */
#define IMGENV_SCENE_EMPTY 0
#define IMGENV_SCENE_RVO 1     /* "rvoscene"  rvoscene.h  */
#define IMGENV_SCENE_ERVO 2    /* "ervoscene" ervoscene.h */
#define IMGENV_SCENE_PEDSIM 3  /* "pedscene"  pedscene.h  */
#define IMGENV_SCENE_DATASET 4 /* "dataset": pedestrians replay recorded trajectories (img_env.cpp:295-296, 361-386) */

/* Agent.msg `ktype` of robots (agent.cpp:198, 238) */
#define IMGENV_KTYPE_DIFF 0
#define IMGENV_KTYPE_OMNI 1

/* comn_pkg/msg/SpeedLimiter.msg:1-9 (speed_limit.cpp:56-65) */
typedef struct imgenv_limiter {
    int32_t has_velocity_limits;
    int32_t has_acceleration_limits;
    int32_t has_jerk_limits;
    float min_velocity, max_velocity;
    float min_acceleration, max_acceleration;
    float min_jerk, max_jerk;
} imgenv_limiter;

/*
 * Everything InitEnv.srv carries plus the static ImageEnv / wrapper parameters that the
 * Python side of the reference keeps (yaml_env.py:133-181, base.py:153-231).
 */
typedef struct imgenv_cfg {
    int32_t abi_version;          /* must be IMGENV_ABI_VERSION */
    int32_t struct_size;          /* sizeof(imgenv_cfg), ABI guard */

    /* ---- InitEnv.srv:1-20 (img_env.cpp:58-81) ---- */
    float view_resolution;        /* metres / cell of BOTH grids after load (grid_map.cpp:28-38) */
    float view_width, view_height;/* view extent in metres (agent.cpp:79-90) */
    float step_hz;                /* YAML control_hz: the step dt in seconds (img_env.cpp:80) */
    int32_t state_dim;            /* 3 | 4 | 5 (agent.cpp:163-183) */
    int32_t use_laser;
    int32_t range_total;          /* number of beams */
    float view_angle_begin, view_angle_end;
    float view_min_dist, view_max_dist;
    float beep_r, ped_ca_p;       /* beep lottery (img_env.cpp:323-342): a robot whose action has beep (v_y) > 0 becomes, with
                                   * probability ped_ca_p, a source of radius beep_r that ERVO pedestrians move away from
                                   * (ervo_ros Agent.cpp:63-69).  The reference's Python never forwards either
                                   * (yaml_env.py:183-200 => 0 at the node, the lottery never fires); C-ABI callers may set
                                   * them.  One glibc rand() stream per world, as a fresh node process starts it.  Not
                                   * available in a robot shard (every rank would need every robot's action). */
    int32_t relation_ped_robo;    /* 1: robots are agents of the pedestrian simulator */

    /* ---- Env.msg ---- */
    float global_resolution;      /* metres / pixel of the map handed to imgenv_create; when it differs from view_resolution the
                                   * library resizes the map as GridMap::read_image does (cv::resize INTER_LINEAR to
                                   * int(px * global_resolution / view_resolution), grid_map.cpp:28-38) */
    int32_t ped_scene_type;       /* IMGENV_SCENE_* */
    int32_t n_robots;             /* robots of the WORLD (all shards) */
    int32_t n_peds;
    int32_t robot_ktype;          /* IMGENV_KTYPE_* */
    const int32_t* robot_shape;   /* [n_robots] IMGENV_SHAPE_* */
    const float* robot_size;      /* [n_robots][4]  Agent.msg size[] */
    const float* robot_sensor_cfg;/* [n_robots][2]  Agent.msg sensor_cfg[] */
    imgenv_limiter limiter_v, limiter_w;
    const int32_t* ped_shape;     /* [n_peds] */
    const float* ped_size;        /* [n_peds][6] */
    const float* ped_max_speed;   /* [n_peds] */

    /* ---- Python-side ImageEnv parameters (yaml_env.py:133-181) ---- */
    int32_t image_size[2];        /* sensor_map size (width, height) cv2.resize INTER_CUBIC shrinks the view to (yaml_env.py:433) */
    int32_t ped_image_size[2];
    int32_t max_ped;              /* ped vector has 1 + ped_vec_dim*max_ped entries; n_peds <= max_ped */
    int32_t ped_vec_dim;          /* 7 */
    double ped_image_r;
    double laser_max;
    int32_t laser_norm;
    const double* robot_size_last;/* [n_robots] YAML robot.size[i][-1] as the Python float (yaml_env.py:407) */

    /* ---- wrapper parameters (base.py:153-231) ---- */
    double ped_safety_space;
    int32_t time_max;

    /* ---- robot shard owned by this handle (multi-GPU, one process per GPU) ---- */
    int32_t robot_begin, robot_end; /* [begin, end) within the world's robots; 0,n_robots = unsharded */

    /* ---- implementation knobs ---- */
    int32_t device;               /* HIP device ordinal */
    int32_t flags;                /* IMGENV_FLAG_* */
    /* Optional caller-owned DEVICE arena for every output buffer and the records buffer, so a
     * host framework can alias them zero-copy (e.g. slices of one torch uint8 tensor).  Size it
     * with imgenv_arena_bytes(); NULL = the library allocates (and frees) its own. */
    void* out_arena;
    int64_t out_arena_bytes;

    /* ---- independent worlds in one handle: the reference's env_num processes, batched ---- */
    /* n_worlds copies of the same map and parameters; n_robots and n_peds are TOTALS (multiples of n_worlds), numbered
     * world-major: world k owns robots [k n_robots / n_worlds, (k + 1) n_robots / n_worlds) and the same share of the
     * pedestrians.  Robots and pedestrians only ever see their own world.  0 or 1 = a single world.  Each world is reset
     * on its own with imgenv_reset_world(); its time limit counts from its own reset. */
    int32_t n_worlds;
    int32_t reserved_;
} imgenv_cfg;

#define IMGENV_FLAG_PRIVATE_GRIDS 1 /* oracle only: literal per-robot grid copies (img_env.cpp:620-629) */
/* How the class layer (what a robot's view looks up per map cell) is kept up to date.  Default: the library picks --
 * DENSE: the rasters fill owner layers and every step merges them over every map cell of every world, best when the robots
 * and pedestrians cover a good part of the map; SPARSE: the rasters stamp the class layer directly and stamps expire with
 * their step, nothing per cell, best for big or many maps with few agents each.  The result is the same either way. */
#define IMGENV_FLAG_COMPOSE_DENSE 2
#define IMGENV_FLAG_COMPOSE_SPARSE 4
/* (round 5) Where DENSE used to be the library's pick it now keeps COUNTS on the class layer instead: every robot and pedestrian adds
 * itself to the cells it covers and takes itself off the ones it leaves (fire-and-forget atomics, none at all while it covers
 * the same cells), so no pass over every cell merges anything per step.  IMGENV_FLAG_COMPOSE_DENSE still asks for the owner
 * layers + the per-step merge; IMGENV_FLAG_LAYER_SUM asks for the counting layer wherever it can run (the handle owns every
 * robot, views through k_view), also where the library would have stamped.  Same results in every mode. */
#define IMGENV_FLAG_LAYER_SUM 512
/* imgenv_out.view_maps is not wanted.  It only matters where the view is shrunk into the sensor_map (image_size differs from
 * the view size, every shipped config of the reference: a 400 x 400 view, 160 KB per robot, behind a 48 x 48 sensor_map):
 * the library then evaluates only the 4 x 4 view cells each sensor_map pixel reads and never writes the full-size view;
 * view_maps keeps whatever it held.  ImageState (envs/state/state.py:4-28) has no such field, so img_env_amd's envs set it. */
#define IMGENV_FLAG_NO_VIEW_MAPS 8
/* also produce AgentState.hits_x / hits_y / angular_map (imgenv_out, below) */
#define IMGENV_FLAG_AGENT_STATE_EXTRAS 16
/* Which kernels compute the views.  Default: one wavefront per robot (k_view) wherever it can run; the tiled kernels that
 * spread one robot's view over the chip (csrc/view_big.h) for views k_view cannot pack (beyond 255 x 255 cells or 255 ray
 * steps) and for shrunk sensor_maps.  IMGENV_FLAG_VIEW_TILED asks for the tiled kernels where both can run; the result is the
 * same. */
#define IMGENV_FLAG_VIEW_TILED 32
#define IMGENV_FLAG_VIEW_WAVE 64  /* (reserved: k_view is the default) */
/* Guards for the read-only output arrays (see "Conventions": imgenv_outputs() hands out the kernels' incremental working copies,
 * where the reference returns fresh copies with every service response, img_env.cpp:745-749).
 * IMGENV_FLAG_CHECK_OUTPUTS (debug): every output array is sealed with a checksum when a reset / step has been queued and
 * verified at the start of the next reset / step call; a mismatch fails that call with IMGENV_EINVAL "the caller wrote into
 * imgenv_out.<field>".  The verification synchronises the stream once per call.
 * IMGENV_FLAG_FULL_REWRITE: for callers that cannot promise to leave the arrays alone (in-place normalisation, ...).
 * imgenv_outputs() then hands out a SECOND arena (imgenv_cfg.out_arena if given) whose every byte is rewritten from the
 * kernels' private working copy at the end of every reset / step; whatever the caller does to it never reaches the kernels.
 * Costs one device-to-device copy of imgenv_arena_bytes() per call. */
#define IMGENV_FLAG_CHECK_OUTPUTS 128
#define IMGENV_FLAG_FULL_REWRITE 256
/* IMGENV_FLAG_CHECK_OUTPUTS for the first IMGENV_CHECK_FIRST_CALLS reset / step calls of the handle only; the guard then switches
 * itself off and costs nothing.  A trainer that normalises observations in place does so from its first step: it is told at once,
 * loudly, instead of training on silently corrupted views -- and a correct one pays a synchronisation per call for a fraction of
 * a second.  The Python mirror (img_env_amd.World, ImageEnv, VecImageEnv) creates every handle with it unless told otherwise
 * (params["output_guard"] = "none" | "first" | "check" | "copy"). */
#define IMGENV_FLAG_CHECK_OUTPUTS_FIRST 1024
#define IMGENV_CHECK_FIRST_CALLS 64

/* ResetEnv.srv:1-6 (img_env.cpp:162-292).  Poses are (x, y, qz, qw): geometry_msgs/Pose with a
 * planar orientation; yaw is recovered with tf::Matrix3x3(q).getRPY as the node does. */
typedef struct imgenv_reset_batch {
    int32_t struct_size;
    int32_t n_obstacles;
    const int32_t* obs_shape;     /* [n_obstacles] */
    const float* obs_size;        /* [n_obstacles][4] */
    const double* obs_pose;       /* [n_obstacles][4] */
    const double* robot_pose;     /* [n_robots][4]  (whole world) */
    const double* robot_goal;     /* [n_robots][2] */
    const double* ped_pose;       /* [n_peds][4] */
    const double* ped_goal;       /* [n_peds][2] */
    const int32_t* ped_traj_len;  /* [n_peds] */
    const double* ped_traj;       /* [n_peds][ped_traj_cap][3]  Agent.msg trajectory (x, y, z) */
    int32_t ped_traj_cap;
    int32_t ignore_obstacle;
    const double* ped_traj_v;     /* [n_peds][ped_traj_cap][2]  Agent.msg trajectory_v (vx, vy): IMGENV_SCENE_DATASET only, else NULL */
} imgenv_reset_batch;

/* Device pointers of the per-robot outputs, R = robot_end - robot_begin local robots.
 * Field meaning follows ImageState (envs/state/state.py:4-28) and _get_states
 * (yaml_env.py:446-481). */
typedef struct imgenv_out {
    int32_t struct_size;
    int32_t n_local;              /* R */
    int32_t view_h, view_w;       /* Hv, Wv */
    int32_t n_beams;              /* B (0 when !use_laser) */
    int32_t state_dim;
    int32_t ped_vec_len;          /* 1 + ped_vec_dim*max_ped */
    int32_t image_h, image_w;     /* sensor_maps size: imgenv_cfg.image_size (= the view size unless cv2.resize shrinks it) */
    int32_t grid_h, grid_w;       /* the occupancy grid after the load-time resize (grid_map.cpp:28-38) */
    float* vector_states;         /* [R][state_dim]      AgentState.state (float32 wire) */
    uint8_t* view_maps;           /* [R][Hv][Wv]         AgentState.view_map (8UC1) */
    uint16_t* sensor_maps;        /* [R][image_h][image_w] float16 bits = cv2.resize(view, image_size, INTER_CUBIC) / 255
                                   * (yaml_env.py:431-438; OpenCV copies when the sizes are equal) */
    float* lasers_raw;            /* [R][B]              AgentState.laser */
    double* lasers;               /* [R][B]              _norm_lasers (yaml_env.py:440-444) */
    float* ped_vector_states;     /* [R][ped_vec_len] */
    float* ped_maps;              /* [R][3][Hp][Wp] */
    int8_t* is_collisions;        /* [R]  AgentState.is_collision 0..3 */
    uint8_t* is_arrives;          /* [R] */
    double* step_ds;              /* [R] */
    double* ped_min_dists;        /* [R] (+inf until a ped has been seen) */
    /* ImageEnv.step base outputs (yaml_env.py:372-377) */
    int32_t* base_rewards;        /* [R] arrive - collision code */
    uint8_t* base_dones;          /* [R] */
    /* wrapper stack outputs (base.py:153-254, 69-93) */
    double* rewards;              /* [R] SensorsPaperRewardWrapper, zeroed where !is_clean (the full default stack) */
    double* paper_rewards;        /* [R] SensorsPaperRewardWrapper before the MultiRobotClean mask */
    uint8_t* dones;               /* [R] after TimeLimitWrapper */
    int32_t* dones_info;          /* [R] 0 | 1..3 collision class | 5 arrive | 10 time-out */
    uint8_t* is_clean;            /* [R] MultiRobotCleanWrapper mask used for this step */
    /* simulator state mirrors, handy for tests and GUIs */
    double* robot_pose;           /* [R][3] x, y, theta */
    double* ped_state;            /* [n_peds][4] x, y, vx, vy (world copy) */
    int32_t* counters;            /* [4]: steps since reset (n_worlds > 1: of world 0), #done robots of the last step (local),
                                   * #frozen views since the last imgenv_reset (local), #frozen views since create */
    /* The last STEP's own per-robot scalars: every imgenv_step writes them beside the arrays above, a reset never touches
     * them.  After imgenv_step_autoreset() the rows of the worlds it reset hold, above, the new episode's first observation
     * (NeverStopWrapper, base.py:198-211) and, here, what that step itself returned for them. */
    double* step_rewards;         /* [R] */
    uint8_t* step_dones;          /* [R] */
    int32_t* step_dones_info;     /* [R] */
    uint8_t* step_is_clean;       /* [R] */
    uint8_t* step_is_arrives;     /* [R] */
    int8_t* step_is_collisions;   /* [R] */
    uint8_t* step_all_down;       /* [R] 1 where all robots of the robot's world were done after that step (written by
                                   * imgenv_step_autoreset only) */
    /* The remaining fields of AgentState.msg (src/comn_pkg/msg/AgentState.msg:4-6; filled at agent.cpp:405-438, sent at
     * img_env.cpp:558-560), which ImageEnv never reads: NULL unless the handle was created with IMGENV_FLAG_AGENT_STATE_EXTRAS. */
    float* hits_x;                /* [R][B] hit * cos(beam angle): the hit points in the sensor frame */
    float* hits_y;                /* [R][B] hit * sin(beam angle) */
    float* angular_map;           /* [R][72] nearest hit per 1/72 of the field of view, view_max_dist where nothing was hit */
} imgenv_out;
#define IMGENV_ANGULAR_BINS 72    /* agent.cpp:407 */

typedef struct imgenv imgenv_t;

/* library / build identification: "hip-gfx950" for the product library */
const char* imgenv_backend(void);
/* which build: a hash of the extension's sources and compiler flags (__graft_entry__.source_id), "unstamped" for a library
 * compiled by hand.  profiles/pmc_latest.json carries the id of the library its counters were collected on. */
const char* imgenv_build_id(void);
int32_t imgenv_abi_version(void);
const char* imgenv_last_error(void);

/* init_image_env: builds all per-class static tables (footprints, FOV mask, ray tables) and
 * uploads the static occupancy grid (uint8, row-major [Hg][Wg], rows <-> world x;
 * grid_map.cpp:40-55).  Hg x Wg is the size of `static_map` as given: the map image in its own pixels when
 * global_resolution != view_resolution (the grid the handle works on is then the resized one, imgenv_out.grid_h / grid_w). */
int imgenv_create(const imgenv_cfg* cfg, const uint8_t* static_map, int32_t Hg, int32_t Wg,
                  imgenv_t** out);
/* bytes of output arena a handle created from `cfg` needs (256-byte aligned carve-outs) */
int64_t imgenv_arena_bytes(const imgenv_cfg* cfg);
void imgenv_destroy(imgenv_t* h);

/* reset_image_env: obstacles raster + ORCA obstacle tree + poses, then view + states. */
int imgenv_reset(imgenv_t* h, const imgenv_reset_batch* batch, void* stream);

/* step_image_env + Python post-processing.  actions: DEVICE float[R][3] = (v, w, beep) of the
 * local robots (ContinuousAction, envs/action/action.py:8-20).  Dead robots are zeroed
 * inside (yaml_env.py:319-331).  The call is stream-ordered: actions written by work queued on `stream` in front of it are
 * seen, and outputs of the previous call are not rewritten before work queued on `stream` in front of it has read them. */
int imgenv_step(imgenv_t* h, const float* actions, void* stream);
/* The same with per-call flags.  IMGENV_STEP_ACTIONS_READY: "the actions hold their final values when the call is made".
 * Accepted, and WITHOUT EFFECT since round 6: rounds 4-5 let such a step start its observation kernel on a side stream behind the
 * previous step's views alone, which was not ordered behind the caller's readers of the previous outputs (a replay-buffer copy
 * queued on `stream`, IMGENV_FLAG_FULL_REWRITE's own copy).  Every step now starts that kernel behind a one-wavefront gate that
 * opens when `stream` reaches the step's first kernel -- ordered behind everything queued in front of the call, at the same rate
 * (92.3 against 92.0 us per headline step). */
#define IMGENV_STEP_ACTIONS_READY 1u
int imgenv_step_flags(imgenv_t* h, const float* actions, uint32_t flags, void* stream);

/* The same step split around the one exchange a robot-sharded world needs:
 *   step_begin : pedestrian advance + pose integrate of the local robots, publishes their
 *                records into records[robot_begin:robot_end]
 *   (caller all-gathers `records` in place across ranks, e.g. RCCL ncclAllGather)
 *   step_end   : rasters + per-robot view/observation/reward kernels
 * imgenv_step == step_begin; step_end when the handle owns the whole world. */
/* `actions` must stay as they are until imgenv_step_end has been called: kernels of the step read them on the library's side
 * streams, which are joined into `stream` again at the end of imgenv_step_end (work queued on `stream` BEHIND step_end may
 * overwrite them; work queued between the two calls may not). */
int imgenv_step_begin(imgenv_t* h, const float* actions, void* stream);
int imgenv_step_end(imgenv_t* h, void* stream);
/* records: DEVICE double[n_robots][IMGENV_RECORD_DOUBLES] = x, y, theta, vx, vy, sin(theta/2), cos(theta/2), pad
 * (64 bytes per robot; the half-angle sine / cosine are cached so that the raster / view / observation kernels of
 * every rank build the robot's tf::Transform without re-evaluating them) */
#define IMGENV_RECORD_DOUBLES 8
int imgenv_records(imgenv_t* h, double** records, int64_t* bytes_per_robot);
/* reset counterpart of the exchange (robots' initial records are known to every rank from the
 * batch, so reset needs no collective). */

/* Optional: let the library run the exchange itself.  After imgenv_comm_init() on every rank,
 * imgenv_step() = step_begin; ncclAllGather (RCCL, in place, on the caller's stream); step_end -- no host
 * synchronisation and no framework collective per step.  Needs equal contiguous shards
 * (robot_begin == rank * n_local).  The 128-byte id comes from imgenv_comm_unique_id() on rank 0 and must be
 * broadcast to the other ranks by the caller (any side channel, e.g. torch.distributed).  RCCL is loaded at
 * run time (dlopen); IMGENV_EDEVICE if it is unavailable. */
#define IMGENV_COMM_ID_BYTES 128
int imgenv_comm_unique_id(void* id128);
int imgenv_comm_init(imgenv_t* h, const void* id128, int32_t rank, int32_t n_ranks);
/* what RCCL itself reports for the handle's communicator (ncclCommCount / ncclCommUserRank); IMGENV_ESTATE without one */
int imgenv_comm_info(imgenv_t* h, int32_t* n_ranks, int32_t* rank);

/* out->struct_size on entry: 0 (the caller's imgenv_out is this header's) or the size of the caller's older, shorter struct --
 * then only that many bytes are written (fields are only ever appended at the end). */
int imgenv_outputs(imgenv_t* h, imgenv_out* out);

/* Multi-world handles (imgenv_cfg.n_worlds > 1): reset ONE world while the others keep their state -- what one env
 * process of the reference does when its episode ends (ImageEnv.reset, yaml_env.py:296-317).  The batch holds that
 * world's robots (n_robots / n_worlds), pedestrians and obstacles only.  imgenv_reset() on such a handle resets every
 * world from a batch of all robots / pedestrians (world-major) with one obstacle list shared by all worlds. */
int imgenv_reset_world(imgenv_t* h, int32_t world, const imgenv_reset_batch* batch, void* stream);
/* The same for n distinct worlds at once (batches[q] belongs to worlds[q]): one upload launch and one set of reset / view
 * launches however many worlds ended their episode on this step -- NeverStopWrapper (base.py:198-211) for a whole batch
 * of envs. */
int imgenv_reset_worlds(imgenv_t* h, int32_t n, const int32_t* worlds, const imgenv_reset_batch* batches, void* stream);

/* ---- spawn: the random placement EnvPos.reset does for one episode (envs/utils/reset_helper.py:104-345), natively ----
 * A world of a multi-world handle is reset whenever its episode ends -- dozens of worlds on every step -- so the placement
 * itself has to be cheap.  Rules as in the reference (starts > clearance apart and clear of the obstacles; targets
 * > target_min_dist from their start, > clearance apart, clear of the obstacles; range_view targets in the 4 m box around
 * the start but outside its 2.5 m box; pedestrians walk to their target and, with go_back, back); the random stream is
 * the library's own (seeded per world), not Python's.  Pose types of reset_helper.py:187-300: */
#define IMGENV_POSE_FIX 0         /* [x, y, yaw] */
#define IMGENV_POSE_RAND_ANGLE 1  /* [x, y, yaw_lo, yaw_hi] */
#define IMGENV_POSE_RANGE 2       /* [x_lo, x_hi, y_lo, y_hi], yaw uniform in +-3.14 */
#define IMGENV_POSE_RANGE_YAW 3   /* [x_lo, x_hi, y_lo, y_hi, yaw_lo, yaw_hi] */
#define IMGENV_POSE_RANGE_VIEW 4  /* targets only: [x_lo, x_hi, y_lo, y_hi], drawn around the start (random_view, 62-82) */
#define IMGENV_POSE_RANGE_CIRCLE 5      /* [cx, cy]: on the episode's circle around (cx, cy) at a random angle + noise, facing the
                                         * centre (starts, 223-230); opposite the start (targets, 263-268) */
#define IMGENV_POSE_RANGE_CIRCLE_FIX 6  /* the same at the agent's own share of the circle: angle -3.14 + 6.28 i / n (226-227) */
#define IMGENV_POSE_CIRCLE_FIX 7        /* targets only: exactly opposite the start, no noise, no checks (258-262) */
#define IMGENV_POSE_RANGE_MULTI 8       /* one of several boxes, picked per draw (231-232, 269-270): *_multi / n_*_multi */

typedef struct imgenv_spawn_agent {   /* a robot or a pedestrian */
    int32_t begin_type, target_type;  /* IMGENV_POSE_* */
    double begin[6], target[6];
    double module_size;               /* 2 x the footprint's radius (reset_helper.py:167-186): what must clear the obstacles */
    const double* begin_multi;        /* IMGENV_POSE_RANGE_MULTI: [n_begin_multi][6] boxes x_lo, x_hi, y_lo, y_hi, yaw_lo, yaw_hi */
    const double* target_multi;       /* (a 4-number box of the YAML carries yaw -3.14 .. 3.14) */
    int32_t n_begin_multi, n_target_multi;
} imgenv_spawn_agent;

typedef struct imgenv_spawn_obstacle {
    int32_t shape;                    /* IMGENV_SHAPE_CIRCLE (radius uniform in size_range[0..1]) | _RECTANGLE (size_range = size) */
    int32_t pose_type;                /* IMGENV_POSE_FIX | _RANGE | _RANGE_YAW */
    double size_range[4];
    double pose[6];
} imgenv_spawn_obstacle;

typedef struct imgenv_spawn_cfg {     /* ONE world's cast */
    int32_t struct_size;
    int32_t n_robots, n_peds, n_obstacles;
    const imgenv_spawn_agent* agents;        /* [n_robots + n_peds], robots first */
    const imgenv_spawn_obstacle* obstacles;  /* [n_obstacles] */
    double clearance;                 /* free_check_robo_ped distance (1.0) */
    double target_min_dist;
    double circle_ranges[2];          /* the episode's circle radius is uniform in here (YAML circle_ranges) */
    int32_t go_back;                  /* pedestrians return to their start: 0 no, 1 yes, 2 a coin per pedestrian */
    int32_t ignore_obstacle;
} imgenv_spawn_cfg;

/* One placement into caller-owned host arrays (any of them may be NULL): robot_pose [R][4] (x, y, qz, qw), robot_goal [R][2],
 * ped_pose [P][4], ped_goal [P][2], ped_traj [P][2][3], ped_traj_len [P], obs_shape [O], obs_size [O][4], obs_pose [O][4].
 * Needs no device. */
int imgenv_spawn(const imgenv_spawn_cfg* cfg, uint64_t seed, double* robot_pose, double* robot_goal, double* ped_pose,
                 double* ped_goal, double* ped_traj, int32_t* ped_traj_len, int32_t* obs_shape, float* obs_size, double* obs_pose);
/* imgenv_reset_worlds() with a fresh placement for each listed world, world worlds[q] from seeds[q]. */
int imgenv_reset_worlds_spawn(imgenv_t* h, int32_t n, const int32_t* worlds, const imgenv_spawn_cfg* cfg, const uint64_t* seeds,
                              void* stream);
/* One env step of a handle of n_worlds reference envs the way the trainer runs them (NeverStopWrapper over the default
 * wrapper stack, base.py:198-211, one env process per world): imgenv_step(), then every world whose robots are ALL done
 * (imgenv_out.dones) starts a new episode from a fresh placement, as imgenv_reset_worlds_spawn() would, the k-th such world
 * (ascending world index) from seed seed0 + k.  The list of finished worlds is made on the device and read by the host
 * from page-locked memory, so the call returns with the step complete on `stream` (one host synchronisation, no copy).
 * worlds_out (may be NULL) receives up to cap world indices, *n_out their number. */
int imgenv_step_autoreset(imgenv_t* h, const float* actions, const imgenv_spawn_cfg* cfg, uint64_t seed0, int32_t* worlds_out,
                          int32_t cap, int32_t* n_out, void* stream);

/* imgenv_step_autoreset() without the host in the loop (csrc/spawn_device.h): the finished worlds are found, placed and reset by
 * kernels alone -- placements are drawn ahead into a pool on a side stream with the same rules and random stream as
 * imgenv_spawn() but the device's libm (a placement may differ from the host's in the last bit) -- and the call returns as
 * soon as everything is queued on `stream`: no synchronisation, nothing read back.  The FIRST call with a given spawn cfg
 * fixes seed0: the k-th world reset from then on (in step order, ascending world index within a step) takes placement
 * seed0 + k, as a loop of imgenv_step_autoreset() calls fed seed0 + (worlds reset so far) would hand them out.  RVO, ERVO and
 * empty scenes, and social-force crowds (pedscene) in handles of several worlds; at most 256 agents and 24 obstacles per world. */
int imgenv_step_autoreset_device(imgenv_t* h, const float* actions, const imgenv_spawn_cfg* cfg, uint64_t seed0, void* stream);
/* What the last such call did, for checkers and hosts that do want to know (synchronises `stream`): the worlds it reset
 * (ascending, up to cap), their number, and the placement number of the first of them. */
int imgenv_autoreset_last(imgenv_t* h, int32_t* worlds_out, int32_t cap, int32_t* n_out, uint64_t* first_placement, void* stream);
/* The placement world `world` currently runs as its device-side reset received it -- the arrays of imgenv_spawn(), any may be
 * NULL -- and its number.  Synchronises the device. */
int imgenv_world_placement(imgenv_t* h, int32_t world, uint64_t* placement, double* robot_pose, double* robot_goal, double* ped_pose,
                           double* ped_goal, double* ped_traj, int32_t* ped_traj_len, int32_t* obs_shape, float* obs_size, double* obs_pose);

/* The two OpenCV resizes of the path for one-channel 8-bit images, as the library performs them (OpenCV 4.2.0's generic
 * fixed-point CPU path restated, csrc/cv_resize.h): host buffers, no device needed.  kind 0: INTER_LINEAR, 1: INTER_CUBIC. */
int imgenv_cv_resize_u8(int kind, const uint8_t* src, int32_t sh, int32_t sw, uint8_t* dst, int32_t dh, int32_t dw);

/* number of kernels launched by the last step (bench / profiling aid) */
int imgenv_step_launches(imgenv_t* h);
/* how the handle keeps its class layer and schedules its steps (what imgenv_create decided; tests and probes assert on it):
 * bits 0-1: 0 composed owner layers + k_compose, 1 stamps, 2 counts (SUM); bit 2: robot shard in SUM mode (bitmaps in the
 * records, k_remote); bit 3: early-observation steps; bit 4: the social-force crowd a step ahead */
int imgenv_layer_mode(imgenv_t* h);

/* Live per-kernel timing with HIP events recorded on the stream the kernels are launched on.
 * mode 0: off; 1: every launch of every kernel; 2: every 8th launch of kernel `which` only (sampling keeps the
 * event barriers out of most steps).  Kernel ids (K_PED_UPDATE is part of the K_INTEGRATE launch): */
#define IMGENV_K_ORCA 0
#define IMGENV_K_PED_UPDATE 1
#define IMGENV_K_INTEGRATE 2
#define IMGENV_K_RASTER 3
#define IMGENV_K_COMPOSE 4
#define IMGENV_K_VIEW 5
#define IMGENV_K_OBS 6
#define IMGENV_K_TAIL 7
#define IMGENV_K_CROP 8      /* big views (csrc/view_big.h): k_crop_big; IMGENV_K_VIEW is then k_beams_big */
#define IMGENV_K_FULLVIEW 9  /* k_fullview_big */
#define IMGENV_K_TAPS 10     /* k_taps_big */
#define IMGENV_K_MOVE_RASTER 11 /* k_move_raster: K_INTEGRATE + K_RASTER as one launch (small / pedestrian-free handles) */
#define IMGENV_K_EXCHANGE 12    /* the in-library ncclAllGather of the robot records (imgenv_comm_init), between K_INTEGRATE and K_RASTER */
#define IMGENV_K_REMOTE 13      /* k_remote: a robot shard takes the other ranks' robots from their records, behind the exchange */
#define IMGENV_K_COUNT 14
int imgenv_timing(imgenv_t* h, int mode, int which);
/* synchronises the recorded events and returns accumulated milliseconds / launch counts per kernel
 * id since the last imgenv_timing() call; arrays of IMGENV_K_COUNT entries */
int imgenv_timing_read(imgenv_t* h, double* total_ms, int64_t* launches);
const char* imgenv_kernel_name(int id);

#ifdef __cplusplus
}
#endif
#endif /* IMGENV_H_ */
