// imgenv_hip.hip -- C ABI (include/imgenv.h) of the MI355X-native img_env step() path.
// Host orchestration of ImgEnv::_init/_reset/_step (img_env.cpp:83-160, 162-292, 421-525) as a fixed
// sequence of HIP kernels on one stream; all simulator state lives in HBM for the life of the handle.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off (see __graft_entry__.py)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>
#include <stdio.h>

#include <string>
#include <chrono>
#include <memory>
#include <unordered_map>
#include <vector>

#define WAVE_SZ 64
#include "../../include/imgenv.h"
#include "host_tables.h"
#include "cv_resize.h"
#include "spawn_host.h"
#include "kernels.h"
#include "world.h"

static thread_local char g_err[512] = "";
#define FAIL(code, ...)                              \
    do {                                             \
        snprintf(g_err, sizeof(g_err), __VA_ARGS__); \
        return (code);                               \
    } while (0)
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) FAIL(IMGENV_EDEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct imgenv {
    imgenv_cfg cfg;
    ViewGeom geom;
    int R = 0, P = 0, r0 = 0, r1 = 0, RL = 0, NA = 0, Hg = 0, Wg = 0, PP = 2;
    int W = 1, Rw = 0, Pw = 0;  // independent worlds in this handle, robots / pedestrians per world
    size_t Gs = 0;              // cells between two worlds' copies of a grid layer
    std::vector<int> world_epoch;  // h->elapsed at each world's last reset
    std::vector<char> world_ready;
    int* d_world_epoch = nullptr;
    int* d_wobst = nullptr;        // [4][W]: obstacle base, node base, #obstacles, root per world
    std::vector<int> wobst;
    std::vector<RobotClassHost> rcls;
    std::vector<PedClassHost> pcls;
    std::vector<int> robot_cls, ped_cls;
    std::vector<uint8_t> static_map;
    std::vector<double> rsl;
    std::vector<float> pmax;
    DevWorld d;
    std::vector<void*> allocs;
    unsigned char* arena = nullptr;
    bool own_arena = false;
    imgenv_out out;
    uint8_t* d_obs_map = nullptr;
    RvoObstDev* d_obst = nullptr;
    RvoNodeDev* d_nodes = nullptr;
    int cap_obst = 0, cap_nodes = 0;
    double* d_traj = nullptr;
    double* d_traj_v = nullptr;  // dataset scene
    // the robots / pedestrians of the reset being staged, in list order, in page-locked memory the reset kernel reads
    double *pin_rob3 = nullptr, *pin_ped3 = nullptr;
    void* pin_rr = nullptr;
    int* pin_list = nullptr;
    // pinned host staging of imgenv_reset: copies are truly asynchronous and reset never waits for the stream
    struct Chunk { unsigned char* p; size_t cap, used; };
    // STAGE_GENS generations of chunks, used round-robin: a reset only ever waits for the reset STAGE_GENS calls back (the
    // host may run several steps ahead of the device; waiting for the previous reset would drain that queue every time)
    static constexpr int STAGE_GENS = 4;
    std::vector<Chunk> stage_gen[STAGE_GENS];
    hipEvent_t ev_gen[STAGE_GENS] = {nullptr, nullptr, nullptr, nullptr};
    bool gen_pending[STAGE_GENS] = {false, false, false, false};
    int gen = 0;
    struct StageSegHost { void* dst; const void* src; size_t bytes; };
    std::vector<StageSegHost> segs;  // copies queued for the next stage_flush
    size_t seg_max = 0;
    uint8_t* d_static_map = nullptr;  // the map every reset starts from
    struct ObstInstHost { double x, y, sh, ch, cx, cy, r; int m0, m1, n0, n1, shape, world; };
    std::vector<ObstInstHost> oinst;  // obstacles of the reset being staged
    void* d_oinst = nullptr;
    size_t cap_oinst = 0;
    int* d_act_list = nullptr;
    bool big_view = false;   // the view is beyond k_view's packing, or shrunk by cv2.resize: the kernels of view_big.h
    int big_tap_chunks_dyn = 0;  // k_taps_big workgroups per robot in a step (the largest list of any class: BigClassDev::tap_chunks)
    size_t lds_view_big = 0;
    bool big_bits_in_lds = true;  // the crop bitmap of one robot fits the LDS next to the hit words
    int big_max_crop = 1, big_full_chunks = 1;
    // early-observation steps (world.h): k_obs is launched with the step, beside the move (k_move_raster), from snapshots
    bool early = false;             // the handle can run them (IMGENV_EARLY_OBS=0 in the environment switches them off)
    float4* ped_snap[2] = {nullptr, nullptr};
    unsigned orca_seq = 0;          // k_orca launches so far: launch q writes ped_snap[q & 1]
    double* rec_snap[2] = {nullptr, nullptr};
    unsigned view_seq = 0;          // k_view launches over every world so far: launch q writes rec_snap[q & 1]
    bool early_step = false;
    // an event that only has to say "this kernel is done" rides on the kernel's own dispatch packet (hipExtLaunchKernelGGL's stop
    // event): a hipEventRecord behind the kernel is a packet of its own, and the caller's stream pays ~6 us for each
    uint32_t gate_seq = 0;       // early steps so far: each one's k_obs waits behind a gate (world.h: sync) for its number
    // the social-force crowd a step AHEAD (sfm.h: SfmDev *_out): crowds that ignore the robots (relation_ped_robo = 0) depend on
    // nothing of a step, so k_sfm for step t + 1 runs on a stream of its own underneath step t's rasters and views, reads the
    // crowd as it is and leaves the next state in a second set of arrays; step t + 1 swaps the sets and only publishes
    // (k_sfm_publish).  A reset in between drops what was computed ahead.
    bool sfm_ahead = false;        // the handle can do that
    bool sfm_ahead_valid = false;  // the other set holds the state of the next step (computed or being computed on sfm_stream)
    int sfm_steps_since_reset = 0;
    SfmDev sfm_other;              // the other set of the arrays a step writes (its p / v / dq / dest / last / nodes / n_nodes / treehash)
    hipStream_t sfm_stream = nullptr;
    hipEvent_t ev_sfm = nullptr;
    hipEvent_t ev_sfm_in[2] = {nullptr, nullptr};  // behind a step's last access to the crowd's sets on the caller's stream, by step parity
    int sfm_par = 0;
    int sfm_ahead_wait = -1;  // >= 0: this imgenv_step_begin still owes the crowd's launch ahead, behind ev_sfm_in[that]
    bool gates_work = false;     // k_gate_probe's verdict: kernels of two streams run side by side in this process
    bool fork_on_move = false;   // ev_fork went out with this step's k_integrate
    bool sum = false;        // SUM mode of the class layer (world.h): base class + counts kept by the agents themselves, no k_compose
    bool stamp = false;      // STAMP mode of the class layer (world.h) instead of two owner layers + k_compose
    uint32_t stamp_seq = 0;  // steps so far: the stamps of a step carry tag stamp_seq % STAMP_TAGS + 1
    std::vector<double> tmp_d1;  // scratch of stage_world
    std::vector<int> tmp_i0;
    int* d_traj_len = nullptr;
    int traj_cap = 0;
    int elapsed = 0;
    bool has_reset = false;
    int launches = 0;
    size_t lds_view = 0, lds_obs = 0;
    int obs_E = 0;  // sort slots per lane of k_obs (0: LDS sort)
    int n_sub = 0;  // 0.05 s sub-steps of Agent::cmd per step (agent.cpp:221-236)
    bool pow2 = false;
    // the ORCA solve of step t+1 only needs what exists after the rasters of step t, so it runs on a side
    // stream underneath the view / observation kernels of step t (200 waves alone cannot fill the chip)
    ncclComm_t comm = nullptr;  // optional: native RCCL exchange inside imgenv_step
    int comm_ranks = 0;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t side2 = nullptr;  // pedestrian observation (k_obs) underneath raster / compose / view
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;
    bool serial = false;  // IMGENV_SERIAL=1: no side streams (profiling aid)
    volatile int* err_host = nullptr;  // [8] page-locked flags the kernels raise on overflow; checked at every API call
    volatile int* finished_host = nullptr;  // [1 + W] page-locked: the worlds whose robots are all done (k_finished)
    // imgenv_step_autoreset draws the placements of the next few seeds while the device is still busy with the step
    std::unordered_map<uint64_t, std::unique_ptr<SpawnOut>> spawn_ahead;
    uint64_t spawn_ahead_cfg = 0;  // fingerprint of the spawn cfg the placements were drawn from
    int spawn_ahead_n = 8;
    bool obs_forked = false;  // k_obs of the current step is already in flight (launched by step_begin)
    // imgenv_step on a handle of at most 4096 robots, all local: the move is left to the raster launch (k_move_raster)
    bool in_step = false, move_pending = false;
    const float* move_actions = nullptr;
    double trace_acc[4] = {0, 0, 0, 0};  // IMGENV_TRACE_RESET: host time inside imgenv_step_autoreset
    long trace_calls = 0, trace_resets = 0;
    // device-side auto-reset (csrc/spawn_device.h): pool of placements drawn ahead on a side stream
    bool sd_ready = false, dev_reset_used = false;
    int fill_due = 0;  // calls until the placement pool is refilled again (SPAWN_FILL_PERIOD)
    bool fill_pending = false;  // ev_fill has been recorded behind a k_spawn_fill that nothing has waited for yet
    bool wobst_all = false;  // the per-world RVO table has to be uploaded as a whole (its slices moved)
    // (The chain as a replayed hipGraph was measured in round 4 and dropped in round 6: this runtime issues a replayed graph node
    // by node, 222 us of host time per step against 195 us for the same ~25 plain launches -- docs/HISTORY.md.)
    int act_hint = 0;  // device-side auto-reset: robots the reset chain is expected to cover (picks the small-launch kernel variants)
    uint64_t sd_fp = 0;
    hipStream_t side3 = nullptr;
    hipEvent_t ev_fill = nullptr, ev_consumed = nullptr;
    void* sd_storage = nullptr;  // SpawnDev (defined behind the kernels)
    void (*sd_delete)(void*) = nullptr;
    // output guards (include/imgenv.h: IMGENV_FLAG_CHECK_OUTPUTS / IMGENV_FLAG_FULL_REWRITE)
    unsigned char* pub_arena = nullptr;  // FULL_REWRITE: what imgenv_outputs hands out -- a copy of the working arena made at the end of every chain
    bool own_pub = false;
    size_t arena_bytes = 0;
    imgenv_out pub_out;
    bool guard_check = false, guard_sealed = false;
    int guard_left = -1;  // IMGENV_FLAG_CHECK_OUTPUTS_FIRST: verifications until the guard switches itself off (-1: never)
    OutSpan* d_spans = nullptr;          // CHECK_OUTPUTS: the output arrays as (pointer, bytes, first block) ...
    int n_spans = 0, span_blocks = 0;
    unsigned long long* d_sums = nullptr;  // ... and their checksums: [2][n_spans], sealed | found at the next call
    const char* span_name[40];
    bool chain_open = false;  // a chain of launches that hands over through tail_sig / tail_cnt has started and not been completed
    std::vector<RvoObstacles> rvos;  // one obstacle set per world
    int sfm_cap_obs = 0;
    std::vector<int> sfm_nobs_w;  // pedscene, several worlds: obstacle segments of each world's crowd
    // live timing (imgenv_timing)
    int t_mode = 0, t_which = -1;
    unsigned t_tick = 0;
    std::vector<hipEvent_t> t_ev;
    std::vector<int> t_id;  // kernel id of event pair q
    size_t t_used = 0;
    double t_ms[IMGENV_K_COUNT] = {0};
    int64_t t_cnt[IMGENV_K_COUNT] = {0};
};

// RCCL is resolved at run time so that single-GPU users do not need it: prefer the copy the process already
// loaded (torch ships one), then the ROCm one
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_id = nullptr;
    decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclGetErrorString) err = nullptr;
    decltype(&ncclCommCount) count = nullptr;
    decltype(&ncclCommUserRank) user_rank = nullptr;
};
static RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        void* lib = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);
        for (int q = 0; !lib && q < 3; q++) lib = dlopen(names[q], RTLD_NOW | RTLD_GLOBAL);
        if (lib) {
            api.get_id = (decltype(api.get_id))dlsym(lib, "ncclGetUniqueId");
            api.init_rank = (decltype(api.init_rank))dlsym(lib, "ncclCommInitRank");
            api.all_gather = (decltype(api.all_gather))dlsym(lib, "ncclAllGather");
            api.destroy = (decltype(api.destroy))dlsym(lib, "ncclCommDestroy");
            api.err = (decltype(api.err))dlsym(lib, "ncclGetErrorString");
            api.count = (decltype(api.count))dlsym(lib, "ncclCommCount");
            api.user_rank = (decltype(api.user_rank))dlsym(lib, "ncclCommUserRank");
            if (api.count && api.user_rank && api.get_id && api.init_rank && api.all_gather && api.destroy && api.err) api.lib = lib;
        }
    }
    return api.lib ? &api : nullptr;
}

static const char* const KERNEL_NAMES[IMGENV_K_COUNT] = {"k_orca", "k_ped_update", "k_integrate", "k_raster", "k_compose",
                                                         "k_view", "k_obs", "k_tail", "k_crop_big", "k_fullview_big", "k_taps_big", "k_move_raster", "rccl_all_gather",
                                                         "k_remote"};
extern "C" const char* imgenv_kernel_name(int id) { return (id >= 0 && id < IMGENV_K_COUNT) ? KERNEL_NAMES[id] : ""; }

static int timing_flush(imgenv* h) {
    for (size_t q = 0; q < h->t_used; q++) {
        float ms = 0;
        HIPCHK(hipEventSynchronize(h->t_ev[2 * q + 1]));
        HIPCHK(hipEventElapsedTime(&ms, h->t_ev[2 * q], h->t_ev[2 * q + 1]));
        h->t_ms[h->t_id[q]] += ms;
        h->t_cnt[h->t_id[q]] += 1;
    }
    h->t_used = 0;
    return 0;
}
// mode 1: every launch of every kernel; mode 2: every 8th launch of one kernel (an event pair costs the stream a
// dependency barrier, ~5 us of idle GPU, so the timed bench pass samples instead of bracketing every step)
static inline bool timing_on(imgenv* h, int id) {
    if (h->t_mode == 1) return true;
    return h->t_mode == 2 && h->t_which == id && (h->t_tick++ & 7) == 0;
}
static int timing_mark(imgenv* h, int id, hipStream_t st, int end) {
    if (!end) {
        if (h->t_used * 2 + 2 > h->t_ev.size()) {
            if (h->t_ev.size() >= 16384) {
                if (int rc = timing_flush(h)) return rc;
            } else {
                for (int q = 0; q < 2; q++) {
                    hipEvent_t e;
                    HIPCHK(hipEventCreate(&e));
                    h->t_ev.push_back(e);
                }
                h->t_id.push_back(id);
            }
        }
        h->t_id[h->t_used] = id;
        HIPCHK(hipEventRecord(h->t_ev[2 * h->t_used], st));
    } else {
        HIPCHK(hipEventRecord(h->t_ev[2 * h->t_used + 1], st));
        h->t_used++;
    }
    return 0;
}
// launch wrapper: optional event pair on the launch stream around one kernel
#define TIMED(h, id, st, launch)                                       \
    do {                                                               \
        const bool on_ = timing_on(h, id);                             \
        if (on_) { if (int rc_ = timing_mark(h, id, st, 0)) return rc_; } \
        launch;                                                        \
        if (on_) { if (int rc_ = timing_mark(h, id, st, 1)) return rc_; } \
    } while (0)

extern "C" int imgenv_timing(imgenv_t* h, int mode, int which) {
    if (!h || mode < 0 || mode > 2) FAIL(IMGENV_EINVAL, "bad timing mode");
    if (int rc = timing_flush(h)) return rc;
    if (mode != 0) {  // create the event pool up front: hipEventCreate inside a timed region would be measured
        const size_t want = mode == 1 ? 2 * 64 * IMGENV_K_COUNT : 2 * 1024;
        while (h->t_ev.size() < want) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            h->t_ev.push_back(e);
            if (h->t_ev.size() % 2 == 0) h->t_id.push_back(0);
        }
    }
    h->t_mode = mode;
    h->t_which = which;
    for (int q = 0; q < IMGENV_K_COUNT; q++) {
        h->t_ms[q] = 0;
        h->t_cnt[q] = 0;
    }
    return IMGENV_OK;
}
extern "C" int imgenv_timing_read(imgenv_t* h, double* total_ms, int64_t* launches) {
    if (!h || !total_ms || !launches) FAIL(IMGENV_EINVAL, "null argument");
    if (int rc = timing_flush(h)) return rc;
    for (int q = 0; q < IMGENV_K_COUNT; q++) {
        total_ms[q] = h->t_ms[q];
        launches[q] = h->t_cnt[q];
    }
    return IMGENV_OK;
}

// "hip-gfx950" is the product build only: a library compiled with any work-skipping experiment switch or with the
// profiling instrumentation says so, so that a number measured on it can never pass for the product's
#if defined(IMGENV_EXP_STOP_AFTER)
extern "C" const char* imgenv_backend(void) { return "hip-gfx950-EXPERIMENT-work-skipped"; }
#elif defined(IMGENV_PHASE_PROFILE) || defined(IMGENV_WAVE_TIMELINE) || defined(IMGENV_EXP_RESOLVE_STATS) || defined(IMGENV_EXP_TINY_RESOLVE) || \
    defined(IMGENV_EXP_SKEW) || defined(IMGENV_EXP_EVERY_CELL)
extern "C" const char* imgenv_backend(void) { return "hip-gfx950-profile-instrumented"; }
#else
extern "C" const char* imgenv_backend(void) { return "hip-gfx950"; }
#endif
// which sources + flags this library was compiled from (__graft_entry__.source_id): counter files under profiles/ name the
// build they were collected on, and bench.py only quotes them for the library it is running
#ifndef IMGENV_BUILD_ID
#define IMGENV_BUILD_ID "unstamped"
#endif
// (the id sits behind a marker so that a build script can read it out of the file's bytes without loading the library into its
// own process: a dlopen'ed image is never replaced by a rebuild of the same path -- __graft_entry__._built_id)
extern "C" const char imgenv_build_marker[] = "IMGENV_BUILD_ID=" IMGENV_BUILD_ID;
extern "C" const char* imgenv_build_id(void) { return imgenv_build_marker + 16; }
extern "C" int32_t imgenv_abi_version(void) { return IMGENV_ABI_VERSION; }
extern "C" const char* imgenv_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------- arena
struct OutField {
    size_t offset, bytes;
};
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// LDS of k_taps_big: hit words (+ the dummy beam) | 16 tap values per pixel | list of the taps that need a second look | counter
static size_t taps_lds_bytes(int B) { return 16 * (size_t)((B + 4) / 4) + 16 * (size_t)VBT_T + 2 * 16 * (size_t)VBT_T + 16; }


struct ArenaPlan {
    size_t total = 0;
    size_t off[40];
    int n = 0;
    size_t add(size_t bytes) {
        off[n] = total;
        total = align256(total + (bytes ? bytes : 1));
        return off[n++];
    }
};

static void plan_arena(const imgenv_cfg& c, const ViewGeom& g, int RL, ArenaPlan& p) {
    const size_t R = (size_t)RL, NC = (size_t)g.Hv * g.Wv, B = (size_t)(g.B > 0 ? g.B : 1);
    const size_t PV = 1 + (size_t)c.ped_vec_dim * c.max_ped, NP = (size_t)c.ped_image_size[0] * c.ped_image_size[1];
    const size_t P = (size_t)(c.n_peds > 0 ? c.n_peds : 1);
    p.add(R * c.state_dim * 4);  // 0 vector_states
    p.add(R * NC);               // 1 view_maps
    p.add(R * (size_t)c.image_size[0] * (size_t)c.image_size[1] * 2);  // 2 sensor_maps [image_h][image_w] f16
    p.add(R * B * 4);            // 3 lasers_raw
    p.add(R * B * 8);            // 4 lasers
    p.add(R * PV * 4);           // 5 ped_vector_states
    p.add(R * 3 * NP * 4);       // 6 ped_maps
    p.add(R);                    // 7 is_collisions
    p.add(R);                    // 8 is_arrives
    p.add(R * 8);                // 9 step_ds
    p.add(R * 8);                // 10 ped_min_dists
    p.add(R * 4);                // 11 base_rewards
    p.add(R);                    // 12 base_dones
    p.add(R * 8);                // 13 rewards
    p.add(R);                    // 14 dones
    p.add(R * 4);                // 15 dones_info
    p.add(R);                    // 16 is_clean
    p.add(R * 24);               // 17 robot_pose
    p.add(P * 32);               // 18 ped_state
    p.add(16);                   // 19 counters
    p.add((size_t)c.n_robots * IMGENV_RECORD_DOUBLES * 8);  // 20 records
    p.add(R * 8);                // 21 paper_rewards
    p.add(R * 8);                // 22 step_rewards
    p.add(R);                    // 23 step_dones
    p.add(R * 4);                // 24 step_dones_info
    p.add(R);                    // 25 step_is_clean
    p.add(R);                    // 26 step_is_arrives
    p.add(R);                    // 27 step_is_collisions
    p.add(R);                    // 28 step_all_down
    const bool extras = (c.flags & IMGENV_FLAG_AGENT_STATE_EXTRAS) != 0;
    p.add(extras ? R * B * 4 : 0);  // 29 hits_x
    p.add(extras ? R * B * 4 : 0);  // 30 hits_y
    p.add(extras ? R * IMGENV_ANGULAR_BINS * 4 : 0);  // 31 angular_map
}

static int shard_of(const imgenv_cfg& c, int& r0, int& r1) {
    r0 = c.robot_begin;
    r1 = c.robot_end;
    if (r0 == 0 && r1 == 0) r1 = c.n_robots;
    if (r0 < 0 || r1 > c.n_robots || r0 >= r1) return -1;
    return 0;
}

extern "C" int64_t imgenv_arena_bytes(const imgenv_cfg* cfg) {
    if (!cfg || cfg->struct_size != (int32_t)sizeof(imgenv_cfg)) return -1;
    int r0, r1;
    if (shard_of(*cfg, r0, r1)) return -1;
    ArenaPlan p;
    plan_arena(*cfg, make_view_geom(*cfg), r1 - r0, p);
    return (int64_t)p.total;
}

// ---------------------------------------------------------------------------------------- helpers
template <typename T>
static int dev_alloc(imgenv* h, T** out, size_t n, int fill = 0) {
    void* p = nullptr;
    const size_t bytes = sizeof(T) * (n ? n : 1);
    HIPCHK(hipMalloc(&p, bytes));
    h->allocs.push_back(p);
    HIPCHK(hipMemset(p, fill, bytes));
    *out = (T*)p;
    return 0;
}
// hands one dev_alloc'd block back before the handle dies (the caller has made sure nothing in flight uses it)
template <typename T>
static void dev_free(imgenv* h, T*& p) {
    if (!p) return;
    for (size_t q = 0; q < h->allocs.size(); q++)
        if (h->allocs[q] == (void*)p) {
            h->allocs[q] = h->allocs.back();
            h->allocs.pop_back();
            (void)hipFree((void*)p);
            break;
        }
    p = nullptr;
}
template <typename T>
static int dev_upload(imgenv* h, const T** out, const std::vector<T>& v) {
    T* p = nullptr;
    if (int rc = dev_alloc(h, &p, v.size())) return rc;
    if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    *out = p;
    return 0;
}
// inside imgenv_create once the handle exists: a failing HIP call must not leak the handle and what it has allocated so far
#define HIPCHK_H(expr)                                                                                        \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            snprintf(g_err, sizeof(g_err), "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            imgenv_destroy(h);                                                                                \
            return IMGENV_EDEVICE;                                                                            \
        }                                                                                                     \
    } while (0)
#define TRY(expr)              \
    do {                       \
        int rc_ = (expr);      \
        if (rc_) {             \
            imgenv_destroy(h); \
            return rc_;        \
        }                      \
    } while (0)

extern "C" void imgenv_destroy(imgenv_t* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->comm && rccl_api()) (void)rccl_api()->destroy(h->comm);
    for (hipEvent_t e : h->t_ev) (void)hipEventDestroy(e);
    if (h->side) {
        (void)hipStreamSynchronize(h->side);
        (void)hipStreamDestroy(h->side);
    }
    if (h->side2) {
        (void)hipStreamSynchronize(h->side2);
        (void)hipStreamDestroy(h->side2);
    }
    if (h->err_host) (void)hipHostFree((void*)h->err_host);
    if (h->finished_host) (void)hipHostFree((void*)h->finished_host);
    for (int g = 0; g < imgenv::STAGE_GENS; g++) {
        for (auto& c : h->stage_gen[g]) (void)hipHostFree(c.p);
        if (h->ev_gen[g]) (void)hipEventDestroy(h->ev_gen[g]);
    }
    if (h->side3) {
        (void)hipStreamSynchronize(h->side3);
        (void)hipStreamDestroy(h->side3);
    }
    if (h->sfm_stream) {
        (void)hipStreamSynchronize(h->sfm_stream);
        (void)hipStreamDestroy(h->sfm_stream);
    }
    if (h->ev_sfm) (void)hipEventDestroy(h->ev_sfm);
    for (hipEvent_t e : h->ev_sfm_in)
        if (e) (void)hipEventDestroy(e);
    if (h->ev_fill) (void)hipEventDestroy(h->ev_fill);
    if (h->ev_consumed) (void)hipEventDestroy(h->ev_consumed);
    if (h->sd_storage && h->sd_delete) h->sd_delete(h->sd_storage);
    if (h->ev_fork2) (void)hipEventDestroy(h->ev_fork2);
    if (h->ev_join2) (void)hipEventDestroy(h->ev_join2);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->own_arena && h->arena) (void)hipFree(h->arena);
    if (h->own_pub && h->pub_arena) (void)hipFree(h->pub_arena);
    delete h;
}

// ---------------------------------------------------------------------------------------- create
extern "C" int imgenv_create(const imgenv_cfg* cfg, const uint8_t* static_map, int32_t Hg, int32_t Wg,
                             imgenv_t** out) {
    if (!cfg || !static_map || !out) FAIL(IMGENV_EINVAL, "null argument");
    if (cfg->abi_version != IMGENV_ABI_VERSION || cfg->struct_size != (int32_t)sizeof(imgenv_cfg))
        FAIL(IMGENV_EINVAL, "imgenv_cfg ABI mismatch (version %d size %d, want %d %d)", cfg->abi_version,
             cfg->struct_size, IMGENV_ABI_VERSION, (int)sizeof(imgenv_cfg));
    if (cfg->n_robots < 1 || cfg->n_peds < 0 || Hg < 1 || Wg < 1) FAIL(IMGENV_EINVAL, "bad sizes");
    const int W = cfg->n_worlds > 1 ? cfg->n_worlds : 1;
    if (cfg->n_robots % W || cfg->n_peds % W)
        FAIL(IMGENV_EINVAL, "n_robots %d and n_peds %d must be multiples of n_worlds %d", cfg->n_robots, cfg->n_peds, W);
    if (cfg->n_peds / W > cfg->max_ped)
        FAIL(IMGENV_EINVAL, "n_peds %d > max_ped %d (IndexError in yaml_env.py:401)", cfg->n_peds / W, cfg->max_ped);
    if (cfg->n_peds / W > 65000) FAIL(IMGENV_EINVAL, "more than 65000 pedestrians in one world unsupported");
    if (W > 1 && (((size_t)Hg * Wg + 15) & ~(size_t)15) * (size_t)W >= ((size_t)1 << 32))
        FAIL(IMGENV_EINVAL, "n_worlds x map cells must stay below 2^32");
    if (cfg->n_robots >= (int)OWNER_MULTI) FAIL(IMGENV_EINVAL, "more than 2^24 - 3 robots unsupported");
    if (cfg->image_size[0] < 1 || cfg->image_size[1] < 1) FAIL(IMGENV_EINVAL, "bad image_size");
    if (cfg->state_dim < 3 || cfg->state_dim > 5) FAIL(IMGENV_EINVAL, "state_dim must be 3, 4 or 5");
    if (cfg->ped_vec_dim != 7) FAIL(IMGENV_EINVAL, "ped_vec_dim must be 7");
    if (cfg->ped_scene_type == IMGENV_SCENE_PEDSIM) {
        const int n_sfm = (cfg->n_peds + (cfg->relation_ped_robo == 1 ? cfg->n_robots : 0)) / W;  // one crowd (one PedScene) per world
        if (cfg->relation_ped_robo == 1 && cfg->n_robots / W > 8)
            FAIL(IMGENV_EINVAL, "pedscene with relation_ped_robo=1 and more than 8 robots: the reference node recurses forever in "
                                "Ttree::addAgent (all robot Tagents start at (0,0,0), ped_tree.cpp:65-96)");
        if (n_sfm > SFM_MAX_AGENTS) FAIL(IMGENV_EINVAL, "pedscene crowds larger than %d agents are not supported", SFM_MAX_AGENTS);
    }
    int r0, r1;
    if (shard_of(*cfg, r0, r1)) FAIL(IMGENV_EINVAL, "bad robot shard [%d,%d)", cfg->robot_begin, cfg->robot_end);
    // The lottery only ever has an effect in the ERVO scene (rs_ / ps_ are ignored by the others, img_env.cpp:343).
    const bool beep_on = cfg->ped_scene_type == IMGENV_SCENE_ERVO && cfg->n_peds > 0 && cfg->beep_r > 0 && cfg->ped_ca_p > 0;
    if (beep_on && r1 - r0 != cfg->n_robots)
        FAIL(IMGENV_EINVAL, "beep_r / ped_ca_p > 0 in a robot shard: the beep lottery needs every robot's action on every rank");
    if (W > 1 && r1 - r0 != cfg->n_robots)
        FAIL(IMGENV_EINVAL, "n_worlds > 1 cannot be combined with a robot shard: give each rank whole worlds (its own handle)");
    const ViewGeom g = make_view_geom(*cfg);
    if (g.Hv < 1 || g.Wv < 1 || g.Hv > 4096 || g.Wv > 4096) FAIL(IMGENV_EINVAL, "view of %d x %d cells unsupported", g.Hv, g.Wv);
    if (g.B > 65535) FAIL(IMGENV_EINVAL, "more than 65535 beams unsupported");
    // cv2.resize(view, (image_size[0], image_size[1]), INTER_CUBIC) (yaml_env.py:431-438): dsize = (width, height)
    const bool view_resize = cfg->image_size[0] != g.Wv || cfg->image_size[1] != g.Hv;
    // GridMap::read_image (grid_map.cpp:28-38): the image is resized (INTER_LINEAR) to the view resolution
    std::vector<uint8_t> resized_map;
    if (cfg->global_resolution != cfg->view_resolution) {
        const double resolution_ = (double)cfg->global_resolution, view_res = (double)cfg->view_resolution;
        const double w2d = floor(Wg * resolution_ / view_res), h2d = floor(Hg * resolution_ / view_res);  // (range-checked before the cast)
        if (!(w2d >= 1 && h2d >= 1 && w2d * h2d < 2147483648.0)) FAIL(IMGENV_EINVAL, "the map would be resized to %.0f x %.0f cells", h2d, w2d);
        const int w2 = (int)w2d, h2 = (int)h2d;
        resized_map.resize((size_t)w2 * h2);
        cv_resize_u8(false, static_map, Hg, Wg, resized_map.data(), h2, w2);
        static_map = resized_map.data();
        Hg = h2;
        Wg = w2;
        // the kernels index the worlds' copies of a layer with 32 bits: checked again on the grid the handle really works on
        if (W > 1 && (((size_t)Hg * Wg + 15) & ~(size_t)15) * (size_t)W >= ((size_t)1 << 32))
            FAIL(IMGENV_EINVAL, "n_worlds x map cells must stay below 2^32 (the map is resized to %d x %d cells)", Hg, Wg);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) FAIL(IMGENV_EDEVICE, "no HIP device available");
    if (cfg->device < 0 || cfg->device >= ndev) FAIL(IMGENV_EDEVICE, "device %d out of range (%d)", cfg->device, ndev);
    HIPCHK(hipSetDevice(cfg->device));

    imgenv* h = new imgenv();
    h->cfg = *cfg;
    h->serial = getenv("IMGENV_SERIAL") && getenv("IMGENV_SERIAL")[0] == '1';  // (profiling aid: no side streams, clean per-kernel timings)
    h->geom = g;
    h->R = cfg->n_robots;
    h->P = cfg->n_peds;
    h->r0 = r0;
    h->r1 = r1;
    h->RL = r1 - r0;
    h->Hg = Hg;
    h->Wg = Wg;
    h->W = W;
    h->Rw = h->R / W;
    h->Pw = h->P / W;
    h->Gs = W > 1 ? (((size_t)Hg * Wg + 15) & ~(size_t)15) : (size_t)Hg * Wg;
    h->world_epoch.assign(W, 0);
    h->world_ready.assign(W, 0);
    h->rvos.resize(W);
    const bool rvo = (cfg->ped_scene_type == IMGENV_SCENE_RVO || cfg->ped_scene_type == IMGENV_SCENE_ERVO);
    h->NA = rvo ? h->P + (cfg->relation_ped_robo == 1 ? h->R : 0) : 0;
    h->static_map.assign(static_map, static_map + (size_t)Hg * Wg);
    const int R = h->R, P = h->P, RL = h->RL;

    // classes
    h->robot_cls.resize(R);
    h->rsl.resize(R);
    for (int i = 0; i < R; i++) {
        const int shape = cfg->robot_shape[i];
        if (shape != IMGENV_SHAPE_CIRCLE && shape != IMGENV_SHAPE_RECTANGLE) {
            delete h;
            FAIL(IMGENV_EINVAL, "robot %d: unsupported shape %d", i, shape);
        }
        int found = -1;
        for (size_t c = 0; c < h->rcls.size(); c++)
            if (h->rcls[c].shape == shape && !memcmp(h->rcls[c].size, cfg->robot_size + 4 * i, 16) &&
                !memcmp(h->rcls[c].sensor, cfg->robot_sensor_cfg + 2 * i, 8))
                found = (int)c;
        if (found < 0) {
            RobotClassHost k;
            k.shape = shape;
            memcpy(k.size, cfg->robot_size + 4 * i, 16);
            memcpy(k.sensor, cfg->robot_sensor_cfg + 2 * i, 8);
            // tiled kernels (view_big.h): where k_view cannot run (its packing -- decided in build_robot_class -- or a shrunk
            // sensor_map) and on request.  (Measured on BASELINE cfg-5's 96 x 96 views, 8192 robots beside k_obs<16>: k_view
            // 471 us per step, the tiled kernels 531 -- the same instruction count in three launches; k_view stays the default
            // wherever it can run.)
            const bool tiled = view_resize || ((cfg->flags & IMGENV_FLAG_VIEW_TILED) && !(cfg->flags & IMGENV_FLAG_VIEW_WAVE));
            build_robot_class(k, g, tiled, (cfg->flags & IMGENV_FLAG_AGENT_STATE_EXTRAS) != 0);
            h->rcls.push_back(std::move(k));
            found = (int)h->rcls.size() - 1;
        }
        h->robot_cls[i] = found;
        h->rsl[i] = cfg->robot_size_last ? cfg->robot_size_last[i] : 0.0;
    }
    h->ped_cls.resize(P);
    h->pmax.resize(P);
    std::vector<double> pr_round(P);
    std::vector<float> pr32(P);
    for (int j = 0; j < P; j++) {
        int found = -1;
        for (size_t c = 0; c < h->pcls.size(); c++)
            if (h->pcls[c].shape == cfg->ped_shape[j] && !memcmp(h->pcls[c].size, cfg->ped_size + 6 * j, 24)) found = (int)c;
        if (found < 0) {
            PedClassHost k;
            k.shape = cfg->ped_shape[j];
            memcpy(k.size, cfg->ped_size + 6 * j, 24);
            build_ped_class(k, g.res);
            h->pcls.push_back(std::move(k));
            found = (int)h->pcls.size() - 1;
        }
        h->ped_cls[j] = found;
        h->pmax[j] = cfg->ped_max_speed[j];
        pr32[j] = (float)h->pcls[found].sizes[2];
        pr_round[j] = py_round2((double)pr32[j]);
    }
    h->cfg.robot_shape = nullptr; h->cfg.robot_size = nullptr; h->cfg.robot_sensor_cfg = nullptr;
    h->cfg.ped_shape = nullptr; h->cfg.ped_size = nullptr; h->cfg.ped_max_speed = nullptr;
    h->cfg.robot_size_last = nullptr;

    DevWorld& d = h->d;
    memset(&d, 0, sizeof(d));
    d.R = R; d.RL = RL; d.r0 = r0; d.P = P; d.NA = h->NA;
    d.W = W; d.Rw = h->Rw; d.Pw = h->Pw; d.Gs = (uint32_t)h->Gs;
    d.act_list = nullptr; d.act_nw = W; d.act_nl = RL; d.act_ng = R; d.act_np = P;
    d.act_cells = W > 1 ? h->Gs * W : h->Gs;
    d.Hg = Hg; d.Wg = Wg; d.Hv = g.Hv; d.Wv = g.Wv; d.B = g.B;
    d.Hp = cfg->ped_image_size[0]; d.Wp = cfg->ped_image_size[1];
    d.SD = cfg->state_dim; d.PV = 1 + cfg->ped_vec_dim * cfg->max_ped;
    d.scene = cfg->ped_scene_type; d.relation = cfg->relation_ped_robo; d.ktype = cfg->robot_ktype;
    d.use_laser = cfg->use_laser ? 1 : 0; d.laser_norm = cfg->laser_norm; d.time_max = cfg->time_max;
    d.res = g.res; d.inv_res = 1.0 / g.res;
    {
        int e = 0;
        h->pow2 = (frexp(g.res, &e) == 0.5);  // exact power of two: x * (1/res) == x / res bit for bit
        d.wv_magic = (uint32_t)((0x100000000ull + (uint64_t)g.Wv - 1) / (uint64_t)g.Wv);
    }
    d.step_hz = (double)cfg->step_hz; d.laser_max = cfg->laser_max;
    d.laser_out_nohit = cfg->laser_norm ? (double)6.0f / (double)cfg->laser_max : (double)6.0f;
    d.ped_safety_space = cfg->ped_safety_space; d.ped_image_r = cfg->ped_image_r;
    d.ped_image_r2 = pow(cfg->ped_image_r, 2.0);        // self.ped_image_r ** 2 (yaml_env.py:425)
    {   // while (cur_control <= step_hz) { ...; cur_control += 0.05; } with the loop's own fp64 accumulation
        double cur = 0;
        h->n_sub = 0;
        while (cur <= d.step_hz && h->n_sub < (1 << 20)) {
            h->n_sub++;
            cur += 0.05;
        }
    }
    d.ped_res = 6.0 / cfg->ped_image_size[0];            // yaml_env.py:164
    {
        int e = 0;
        d.ped_inv_res = frexp(d.ped_res, &e) == 0.5 ? 1.0 / d.ped_res : 0.0;
    }
    d.view_base = g.view_base; d.base_view = g.base_view;
    {   // SpeedLimiter(msg) (speed_limit.cpp:56-65): max_jerk <- msg.min_jerk, min_jerk uninitialised (0 here)
        const imgenv_limiter& v = cfg->limiter_v; const imgenv_limiter& ww = cfg->limiter_w;
        d.lv_has_v = v.has_velocity_limits; d.lv_has_a = v.has_acceleration_limits; d.lv_has_j = v.has_jerk_limits;
        d.lv_min_v = v.min_velocity; d.lv_max_v = v.max_velocity; d.lv_min_a = v.min_acceleration;
        d.lv_max_a = v.max_acceleration; d.lv_min_j = 0.0; d.lv_max_j = v.min_jerk;
        d.lw_has_v = ww.has_velocity_limits; d.lw_has_a = ww.has_acceleration_limits; d.lw_has_j = ww.has_jerk_limits;
        d.lw_min_v = ww.min_velocity; d.lw_max_v = ww.max_velocity; d.lw_min_a = ww.min_acceleration;
        d.lw_max_a = ww.max_acceleration; d.lw_min_j = 0.0; d.lw_max_j = ww.min_jerk;
    }
    const size_t G = (size_t)Hg * Wg, Gp = W > 1 ? h->Gs * W : ((G + 15) & ~(size_t)15);

    // grids (one copy per world)
    TRY(dev_alloc(h, &h->d_obs_map, Gp));
    HIPCHK_H(hipMemcpy(h->d_obs_map, static_map, G, hipMemcpyHostToDevice));
    for (size_t filled = 1; filled < (size_t)W; filled *= 2)  // the other worlds' copies: log2(W) device copies instead of W uploads
        HIPCHK_H(hipMemcpy(h->d_obs_map + filled * h->Gs, h->d_obs_map, std::min(filled, (size_t)W - filled) * h->Gs, hipMemcpyDeviceToDevice));
    TRY(dev_alloc(h, &h->d_static_map, (G + 15) & ~(size_t)15));
    HIPCHK_H(hipMemcpy(h->d_static_map, static_map, G, hipMemcpyHostToDevice));
    TRY(dev_alloc(h, &h->d_world_epoch, W));
    d.world_epoch = h->d_world_epoch;
    h->wobst.assign((size_t)4 * W, 0);
    TRY(dev_alloc(h, &h->d_wobst, (size_t)4 * W));
    d.obst_base = h->d_wobst; d.node_base = h->d_wobst + W; d.n_obst_w = h->d_wobst + 2 * W; d.oroot_w = h->d_wobst + 3 * W;
    d.obs_map = h->d_obs_map;
    {   // How is the class layer kept up to date?  Composed: the rasters fill two owner layers and a pedestrian layer with
        // fire-and-forget atomics and k_compose merges (and re-arms) them every step -- 14 bytes a cell, every cell of every
        // world.  Stamped: the rasters merge their stamps straight onto the class layer (compare-and-swap) and stamps expire
        // by their step tag -- nothing per cell, but slower rasters and a slightly longer decode in k_view.  Composed wins
        // where the agents cover a good part of the map (the headline world: 0.133 against 0.138 ms per step), stamped
        // where the maps are much larger than what the agents touch (8192 one-robot worlds: 52 against 43 M robot-steps/s).
        const size_t cells = (size_t)Hg * Wg * W;
        h->stamp = RL == R && cells > (size_t)512 * (R + P);  // measured: composed wins at 277 cells per agent, stamped at 1000
        // ... and handles whose rasters and views are single small launches (at most 1024 blocks: bound by launch latency, not by
        // their work) are better off without the k_compose launch however dense they are (cfg-2, 1024 robots at 156 cells per
        // agent: 38.8 -> 36.3 us per step)
        if (RL == R && R + P <= 1024) h->stamp = true;
        if (cfg->flags & IMGENV_FLAG_COMPOSE_DENSE) h->stamp = false;
        if ((cfg->flags & IMGENV_FLAG_COMPOSE_SPARSE) && RL == R) h->stamp = true;
        if (h->stamp && R >= STAMP_MAX_ROBOTS) h->stamp = false;
        // ... and where the owner layers + k_compose used to be the answer, the counting layer is (round 5: no pass over every cell
        // of every world per step, a robot that covers the same cells as a step ago issues no atomic at all) -- wherever it can
        // run: every footprint fits the rasters' LDS box, views through k_view, and the word has room for the counts: 3 bits of
        // base class, as many pedestrian bits as a world has pedestrians, the robot's index within its world, and at least 6 bits
        // of robot count.  A robot SHARD (round 6; world.h: sum_shard) numbers its own robots 1 .. RL in the index field and needs
        // every footprint's box to fit a 64-bit bitmap, which is how the other ranks' robots arrive.
        auto bits = [](int v) { int b = 0; while ((1 << b) <= v) b++; return b; };  // bits to count up to v
        const bool shard = RL != R;
        const int id_bits = shard ? bits(RL) : std::max(1, bits(h->Rw - 1)), pc_bits = bits(h->Pw);
        bool fits = G < ((size_t)1 << 24) && 3 + pc_bits + 6 + id_bits <= 32;
        for (const RobotClassHost& k : h->rcls)
            fits = fits && !k.big && (2 * k.box_rad + 1) * (2 * k.box_rad + 1) <= RASTER_BOX_CELLS &&
                   (!shard || (2 * k.box_rad - 3) * (2 * k.box_rad - 3) <= WAVE);  // (the bitmap: the box without its margin of two cells)
        for (const PedClassHost& k : h->pcls) fits = fits && (2 * k.box_rad + 1) * (2 * k.box_rad + 1) <= RASTER_BOX_CELLS;
        const bool want_sum = (cfg->flags & IMGENV_FLAG_LAYER_SUM) != 0 || (!h->stamp && !(cfg->flags & IMGENV_FLAG_COMPOSE_DENSE));
        h->sum = want_sum && fits;
        if (h->sum) {
            h->stamp = false;
            d.layer_sum = 1;
            d.sum_shard = shard ? 1 : 0;
            d.sum_pc_mask = (1u << pc_bits) - 1u;
            d.sum_rc_shift = 3u + (uint32_t)pc_bits;
            d.sum_id_shift = 32u - (uint32_t)id_bits;
            d.sum_wg_magic = (((unsigned long long)1 << 40) + (unsigned long long)Wg - 1) / (unsigned long long)Wg;
        }
    }
    d.stamp_tag = 1;
    if (d.sum_shard) {
        d.rm_rad = h->rcls[0].box_rad - 2;
        for (const RobotClassHost& k : h->rcls)
            if (k.box_rad - 2 != d.rm_rad) d.rm_rad = -1;
        TRY(dev_alloc(h, &d.rm_bits, (size_t)R));
        TRY(dev_alloc(h, &d.rm_center, (size_t)R));
    }
    if (!h->stamp && !h->sum) {
        TRY(dev_alloc(h, &d.ped_layer, Gp));
        TRY(dev_alloc(h, &d.own_lo, Gp, 0xFF));
        TRY(dev_alloc(h, &d.own_hi, Gp));
    }
    TRY(dev_alloc(h, &d.cell, Gp));
    TRY(dev_alloc(h, &d.seg_tag, Gp / 64 + 2));
    d.crop_map = nullptr;

    // class tables
    size_t max_stride = WAVE;
    {
        std::vector<RobotClassDev> rc(h->rcls.size());
        for (size_t c = 0; c < h->rcls.size(); c++) {
            const RobotClassHost& k = h->rcls[c];
            RobotClassDev& o = rc[c];
            o.n_fp = k.fp.n();
            {
                std::vector<double2> fp(k.fp.n());
                for (int q = 0; q < k.fp.n(); q++) fp[q] = make_double2(k.fp.x[q], k.fp.y[q]);
                TRY(dev_upload(h, &o.fp, fp));
            }
            TRY(dev_upload(h, &o.fov_bits, k.fov_bits));
            TRY(dev_upload(h, &o.stamp_bits, k.stamp_bits));
            o.ray_maxlen = k.ray_maxlen;
            o.ray_stride = k.ray_stride;
            o.ray_kpad = k.ray_kpad;
            if (!k.ok) {
                imgenv_destroy(h);
                FAIL(IMGENV_EINVAL, "laser ray tables overflow their packing (too many beams x view cells)");
            }
            TRY(dev_upload(h, &o.ray_rows, k.ray_rows));
            TRY(dev_upload(h, &o.ray_len, k.ray_len));
            TRY(dev_upload(h, &o.ray_dist, k.ray_dist));
            if (!k.big) {  // what k_view needs of a beam's first hit, per (step, beam): the host's IEEE division = the device's
                std::vector<uint32_t> fin(4 * k.ray_dist.size());
                for (size_t q = 0; q < k.ray_dist.size(); q++) {
                    const float hd = k.ray_dist[q];
                    const double out = cfg->laser_norm ? (double)hd / (double)cfg->laser_max : (double)hd;
                    fin[4 * q] = k.ray_run[q];
                    memcpy(&fin[4 * q + 1], &hd, 4);
                    memcpy(&fin[4 * q + 2], &out, 8);
                }
                const uint32_t* fin_dev = nullptr;
                TRY(dev_upload(h, &fin_dev, fin));
                o.ray_fin = (const uint4*)fin_dev;
            }
            TRY(dev_upload(h, &o.ray_run, k.ray_run));
            if (!k.ray_hx.empty()) {
                TRY(dev_upload(h, &o.ray_hx, k.ray_hx));
                TRY(dev_upload(h, &o.ray_hy, k.ray_hy));
                TRY(dev_upload(h, &o.bin_start, k.bin_start));
            }
            TRY(dev_upload(h, &o.inv_pack, k.inv_pack));
            TRY(dev_upload(h, &o.inv_ent, k.inv_ent));
            TRY(dev_upload(h, &o.top_ent, k.top_ent));
            TRY(dev_upload(h, &o.dyn_groups, k.dyn_groups));
            o.n_dyn = (int)k.dyn_groups.size();
            TRY(dev_upload(h, &o.all_groups, k.all_groups));
            {
                const uint32_t* cells2 = nullptr;
                TRY(dev_upload(h, &cells2, k.inv_cell));
                o.inv_cell = (const uint2*)cells2;
            }
            o.big = k.big ? 1 : 0;
            h->big_view = h->big_view || k.big;
            o.box_rad = k.box_rad;
            o.n_rows = (int)k.fp_rows.size();
            o.rows = nullptr;
            o.fp_cy = k.fp_cy;
            if (o.n_rows) TRY(dev_upload(h, &o.rows, k.fp_rows));
            max_stride = std::max(max_stride, (size_t)k.ray_stride);
        }
        if (rc.size() > RC_INLINE || h->pcls.size() > PC_INLINE) {
            imgenv_destroy(h);
            FAIL(IMGENV_EINVAL, "more than %d robot or %d pedestrian classes (shape, size, sensor) in one world", RC_INLINE, PC_INLINE);
        }
        for (size_t c = 0; c < rc.size(); c++) d.rc[c] = rc[c];
        TRY(dev_upload(h, &d.rc_mem, rc));
        std::vector<PedClassDev> pc(h->pcls.size());
        for (size_t c = 0; c < h->pcls.size(); c++) {
            const PedClassHost& k = h->pcls[c];
            PedClassDev& o = pc[c];
            memset(&o, 0, sizeof(o));
            o.shape = k.shape;
            o.n_bbox = k.bbox.n();
            o.n_left = k.left.n();
            o.n_right = k.right.n();
            TRY(dev_upload(h, &o.bx, k.bbox.x));
            TRY(dev_upload(h, &o.by, k.bbox.y));
            TRY(dev_upload(h, &o.lx, k.left.x));
            TRY(dev_upload(h, &o.ly, k.left.y));
            TRY(dev_upload(h, &o.rx, k.right.x));
            TRY(dev_upload(h, &o.ry, k.right.y));
            memcpy(o.sizes, k.sizes, sizeof(o.sizes));
            o.box_rad = k.box_rad;
            o.n_brows = (int)k.bbox_rows.size(); o.n_lrows = (int)k.left_rows.size(); o.n_rrows = (int)k.right_rows.size();
            o.bbox_cy = k.bbox_cy;
            if (o.n_brows) TRY(dev_upload(h, &o.brows, k.bbox_rows));
            if (o.n_lrows) TRY(dev_upload(h, &o.lrows, k.left_rows));
            if (o.n_rrows) TRY(dev_upload(h, &o.rrows, k.right_rows));
        }
        for (size_t c = 0; c < pc.size(); c++) d.pc[c] = pc[c];
        if (pc.empty()) pc.resize(1);
        TRY(dev_upload(h, &d.pc_mem, pc));
    }
    TRY(dev_upload(h, &d.robot_cls, h->robot_cls));
    TRY(dev_upload(h, &d.ped_cls, h->ped_cls));
    TRY(dev_upload(h, &d.robot_size_last, h->rsl));
    TRY(dev_upload(h, &d.ped_r_round, pr_round));
    TRY(dev_upload(h, &d.ped_r32, pr32));
    {
        std::vector<uint16_t> lut(256);
        for (int v = 0; v < 256; v++) lut[v] = f32_to_f16((float)v / 255.0f);  // numpy: f16(f32(v)/255)
        TRY(dev_upload(h, &d.f16_lut, lut));
    }
    d.resize = view_resize ? 1 : 0;
    d.img_w = cfg->image_size[0];
    d.img_h = cfg->image_size[1];
    CvAxis ax, ay;
    if (view_resize) {  // axis tables of the bicubic shrink (csrc/cv_resize.h)
        ax = cv_axis(g.Wv, d.img_w, true, true);
        ay = cv_axis(g.Hv, d.img_h, true, false);
        TRY(dev_upload(h, &d.rs_xofs, ax.ofs));
        TRY(dev_upload(h, &d.rs_alpha, ax.coef));
        TRY(dev_upload(h, &d.rs_yofs, ay.ofs));
        TRY(dev_upload(h, &d.rs_beta, ay.coef));
    }
    d.keep_view_maps = (cfg->flags & IMGENV_FLAG_NO_VIEW_MAPS) ? 0 : 1;
    if (h->big_view) {  // view_big.h: tiled crop bitmap + hit words per local robot, static tables per class
        if (h->stamp && G < ((size_t)1 << 24)) {  // view_big.h: the map as k_crop_big wants it (world.h)
            const uint32_t wt = (uint32_t)(Wg + 7) / 8, ht = (uint32_t)(Hg + 7) / 8;
            d.crop_wt = wt;
            d.crop_ws = wt * ht * 64;
            d.crop_magic = (((unsigned long long)1 << 40) + (unsigned long long)Wg - 1) / (unsigned long long)Wg;
            for (int m = 0; m < Hg; m++)  // (the multiply-shift division is exact on this map: row starts and row ends)
                for (int e = 0; e < 2; e++) {
                    const unsigned long long cl = (unsigned long long)m * Wg + (e ? Wg - 1 : 0);
                    if ((int)((cl * d.crop_magic) >> 40) != m) {
                        imgenv_destroy(h);
                        FAIL(IMGENV_EINVAL, "internal: row of cell %llu by multiply-shift", cl);
                    }
                }
            std::vector<uint8_t> tiled((size_t)d.crop_ws, 0);
            for (int m = 0; m < Hg; m++)
                for (int n = 0; n < Wg; n++)
                    tiled[(((size_t)(m >> 3) * wt + (n >> 3)) << 6) | ((m & 7) << 3) | (n & 7)] = static_map[(size_t)m * Wg + n] >= 250 ? 128 : 0;
            uint8_t* sc = nullptr;
            TRY(dev_alloc(h, &sc, (size_t)d.crop_ws));
            HIPCHK_H(hipMemcpy(sc, tiled.data(), tiled.size(), hipMemcpyHostToDevice));
            d.static_crop = sc;
            TRY(dev_alloc(h, &d.crop_map, (size_t)d.crop_ws * W));
            HIPCHK_H(hipMemcpy(d.crop_map, sc, (size_t)d.crop_ws, hipMemcpyDeviceToDevice));
            for (size_t filled = 1; filled < (size_t)W; filled *= 2)
                HIPCHK_H(hipMemcpy(d.crop_map + filled * d.crop_ws, d.crop_map, std::min(filled, (size_t)W - filled) * d.crop_ws, hipMemcpyDeviceToDevice));
        }
        std::vector<BigClassDev> bc(h->rcls.size());
        int max_crop = 1;
        for (size_t c = 0; c < h->rcls.size(); c++) {
            RobotClassHost& k = h->rcls[c];
            BigClassDev& o = bc[c];
            memset(&o, 0, sizeof(o));
            if (!k.big) {
                imgenv_destroy(h);
                FAIL(IMGENV_EINVAL, "robot classes with and without big views in one world");
            }
            o.ta = k.big_ta;
            o.tb = k.big_tb;
            o.n_crop = k.n_crop;
            max_crop = std::max(max_crop, o.n_crop);
            TRY(dev_upload(h, &o.crop_tiles, k.crop_tiles));
            TRY(dev_upload(h, &o.crop_masks, k.crop_masks));
            TRY(dev_upload(h, &o.cells, k.big_cells));
            TRY(dev_upload(h, &o.ray_end, k.ray_end));
            {
                const uint32_t* p2 = nullptr;
                TRY(dev_upload(h, &p2, k.big_inv));
                o.inv = (const uint2*)p2;
            }
            if (view_resize) {
                build_big_taps(k, g, ax.ofs, ay.ofs);
                const uint32_t* p2i = nullptr;
                TRY(dev_upload(h, &o.tap_top, k.tap_top));
                TRY(dev_upload(h, &p2i, k.tap_inv));
                o.tap_inv = (const uint2*)p2i;
                TRY(dev_upload(h, &o.tap_addr, k.tap_addr));
                TRY(dev_upload(h, &o.tap_chunks, k.tap_chunks));
                o.n_tap_chunks = (int)k.tap_chunks.size();
                h->big_tap_chunks_dyn = std::max(h->big_tap_chunks_dyn, o.n_tap_chunks);
                std::vector<uint32_t>().swap(k.tap_top);
                std::vector<uint32_t>().swap(k.tap_inv);
                std::vector<uint32_t>().swap(k.tap_addr);
            }
        }
        for (size_t c = 0; c < bc.size(); c++) d.big[c] = bc[c];
        const size_t tiles = (size_t)h->rcls[0].big_ta * h->rcls[0].big_tb;
        d.big_words = (int)((tiles * 2 + 1 + 3) & ~(size_t)3);  // + the always-free word the padded path entries point at
        d.big_hit_stride = (g.B + 3 + 3) & ~3;  // B hit words, the dummy beam, the flag, the collision code
        TRY(dev_alloc(h, &d.big_bits, (size_t)RL * 2 * d.big_words));
        TRY(dev_alloc(h, &d.big_hit, (size_t)RL * d.big_hit_stride));
        // cells no crop tile covers lie outside the field of view for good: "unknown" in plane 1 (read without the laser only)
        if (!cfg->use_laser)
            HIPCHK_H(hipMemset2D(d.big_bits + d.big_words, (size_t)8 * d.big_words, 0xFF, (size_t)4 * d.big_words, (size_t)RL));
        h->big_max_crop = max_crop;
        h->big_full_chunks = (int)(((size_t)g.Hv * g.Wv + VBF_T * 4 - 1) / (VBF_T * 4));
    }

    // robot / ped state
    TRY(dev_alloc(h, &d.gx, RL)); TRY(dev_alloc(h, &d.gy, RL));
    TRY(dev_alloc(h, &d.l0v, RL)); TRY(dev_alloc(h, &d.l0w, RL)); TRY(dev_alloc(h, &d.l1v, RL)); TRY(dev_alloc(h, &d.l1w, RL));
    TRY(dev_alloc(h, &d.world_target, RL));
    TRY(dev_alloc(h, &d.is_coll, RL)); TRY(dev_alloc(h, &d.is_arr, RL)); TRY(dev_alloc(h, &d.py_done, RL));
    TRY(dev_alloc(h, &d.clean_state, RL, 1));
    TRY(dev_alloc(h, &d.tmp_dist, RL));
    TRY(dev_alloc(h, &d.pm_cells, (size_t)RL * PM_CAP));
    TRY(dev_alloc(h, &d.pm_n, RL));  // 0: the arena starts zeroed, nothing to clear
    {   // k_raster's LDS box and the per-robot footprint cell lists (classes with a huge footprint go without)
        int box = 1, cap = 1;
        for (const RobotClassHost& k : h->rcls) {
            const int side = 2 * k.box_rad + 1;
            if (side * side > RASTER_BOX_CELLS) continue;
            box = std::max(box, side * side);
            cap = std::max(cap, std::min(side * side, k.fp.n()));
        }
        if (h->sum) {  // the pedestrians' boxes share the rasters' LDS, and every pedestrian keeps the list of the cells it counts itself on
            int pbox = 1;
            for (const PedClassHost& k : h->pcls) pbox = std::max(pbox, (2 * k.box_rad + 1) * (2 * k.box_rad + 1));
            d.pd_cap = pbox;
            d.ped_box_cells = pbox;
            box = std::max(box, pbox);
            TRY(dev_alloc(h, &d.pd_cells, (size_t)(P > 0 ? P : 1) * pbox));
            TRY(dev_alloc(h, &d.pd_n, P > 0 ? P : 1));
        }
        d.box_cells = box;
        d.fp_cap = cap;
        TRY(dev_alloc(h, &d.bbox, 4));
        d.sharded = RL != R;
        int max_rad = 0;
        for (const RobotClassHost& k : h->rcls) max_rad = std::max(max_rad, k.box_rad);
        d.region_margin = (int)ceil(0.5 * sqrt((double)g.Hv * g.Hv + (double)g.Wv * g.Wv)) + max_rad + 3;
        TRY(dev_alloc(h, &d.fp_cells, (size_t)RL * cap));
        TRY(dev_alloc(h, &d.fp_n, RL, 0xFF));  // -1 until the first raster
        TRY(dev_alloc(h, &d.fp_pose, (size_t)RL * 3));
    }
    TRY(dev_alloc(h, &d.ppx, P)); TRY(dev_alloc(h, &d.ppy, P)); TRY(dev_alloc(h, &d.pyaw, P));
    TRY(dev_alloc(h, &d.plx, P)); TRY(dev_alloc(h, &d.ply, P)); TRY(dev_alloc(h, &d.pvx, P)); TRY(dev_alloc(h, &d.pvy, P));
    TRY(dev_alloc(h, &d.prem, P)); TRY(dev_alloc(h, &d.llx, P)); TRY(dev_alloc(h, &d.lly, P));
    TRY(dev_alloc(h, &d.rlx, P)); TRY(dev_alloc(h, &d.rly, P));
    TRY(dev_alloc(h, &d.pstate, P)); TRY(dev_alloc(h, &d.ptraj_idx, P));
    TRY(dev_alloc(h, &h->d_traj_len, P));
    d.ptraj_len = h->d_traj_len;
    const int NA = h->NA;
    TRY(dev_alloc(h, &d.apx, NA)); TRY(dev_alloc(h, &d.apy, NA)); TRY(dev_alloc(h, &d.avx, NA)); TRY(dev_alloc(h, &d.avy, NA));
    TRY(dev_alloc(h, &d.anvx, NA)); TRY(dev_alloc(h, &d.anvy, NA));
    TRY(dev_alloc(h, &d.near_n, P)); TRY(dev_alloc(h, &d.near_list, (size_t)(P > 0 ? P : 1) * ORCA_NEAR_CAP));
    {
        std::vector<float> ms(NA > 0 ? NA : 1, 0.6f);  // robots: maxSpeed 0.6 (rvoscene.h:63)
        for (int j = 0; j < P && j < NA; j++) ms[j] = (float)(double)h->pmax[j];
        TRY(dev_upload(h, &d.amax_speed, ms));
    }
    d.beep_on = beep_on ? 1 : 0;
    d.beep_r = cfg->beep_r;
    d.ped_ca_p = (double)cfg->ped_ca_p;
    if (beep_on) {  // one rand() stream per world = per node process of the reference, as a fresh process starts it
        uint32_t s0[31];
        beep_initial_state(s0);
        std::vector<uint32_t> st((size_t)31 * W);
        for (int k = 0; k < W; k++) memcpy(&st[(size_t)31 * k], s0, sizeof(s0));
        const uint32_t* up = nullptr;
        TRY(dev_upload(h, &up, st));
        d.beep_state = const_cast<uint32_t*>(up);
        TRY(dev_upload(h, &d.beep_coef, beep_coefficients(BEEP_T)));
        TRY(dev_alloc(h, &d.beep_flag, R));
        TRY(dev_alloc(h, &d.beep_xy, R));
    }
    {   // the finished-world list of imgenv_step_autoreset: written by k_finished straight into page-locked host memory
        int* f = nullptr;
        HIPCHK_H(hipHostMalloc((void**)&f, (size_t)(1 + W) * sizeof(int), hipHostMallocMapped));
        memset(f, 0, (size_t)(1 + W) * sizeof(int));
        h->finished_host = f;
        int* dev = nullptr;
        HIPCHK_H(hipHostGetDevicePointer((void**)&dev, f, 0));
        d.finished = dev;
    }
    {   // overflow flags live in page-locked host memory the device writes through: no copy, no sync to read them
        int* e = nullptr;
        HIPCHK_H(hipHostMalloc((void**)&e, 8 * sizeof(int), hipHostMallocMapped));
        memset(e, 0, 8 * sizeof(int));
        h->err_host = e;
        int* dev = nullptr;
        HIPCHK_H(hipHostGetDevicePointer((void**)&dev, e, 0));
        d.err = dev;
        d.sfm.err = dev + 4;
    }
    if (cfg->ped_scene_type == IMGENV_SCENE_PEDSIM) {  // PedScene(): Tscene(0,10,10,10), addPed, addRobot (pedscene.h:17-80)
        // one crowd per world (the reference: one node process = one PedScene per env), every array W slices back to back;
        // every world starts as a fresh process does: the same default-seeded draws
        SfmDev& f = d.sfm;
        const int Pw = h->Pw, Rw = h->Rw;
        const int n = Pw + (cfg->relation_ped_robo == 1 ? Rw : 0), n1 = n ? n : 1;
        f.n = n; f.n_peds = Pw; f.n_obs = 0; f.W = W;
        f.cap_nodes = W > 1 ? 2048 : 16384;  // (libpedsim's tree only ever grows: a crowd that outgrows its pool raises the overflow flag)
        f.cap_obs = 0;
        PedsimRng rng;
        std::vector<double> p0((size_t)n1 * 3, 0.0), vmax(n1, 0.0);
        std::vector<SfmNode> nodes(f.cap_nodes);
        std::vector<int> treehash(n1, 0);
        int n_nodes = 0, err = 0;
        sfm_q_new(nodes.data(), &n_nodes, f.cap_nodes, 0, 10, 10, 10);
        // every Tagent() draws its vmax (peds then robots, also robots that stay outside the scene)
        for (int a = 0; a < n; a++) {
            vmax[a] = rng.normal_fresh(1.2, 0.2);
            if (a < Pw) {
                p0[3 * a] = rng.glibc_rand() / 2147483647.0 * 10.0;
                p0[3 * a + 1] = rng.glibc_rand() / 2147483647.0 * 10.0;
                vmax[a] = (double)h->pmax[a];
            }
            sfm_add_agent(nodes.data(), &n_nodes, f.cap_nodes, treehash.data(), p0.data(), a, &err);
        }
        if (err) {
            imgenv_destroy(h);
            FAIL(IMGENV_EINVAL, "pedscene quadtree construction overflowed (%d)", err);
        }
        const size_t Wn = (size_t)W * n1;
        TRY(dev_alloc(h, &f.p, Wn * 3));
        TRY(dev_alloc(h, &f.v, Wn * 3));
        TRY(dev_alloc(h, &f.vmax, Wn)); TRY(dev_alloc(h, &f.wpx, Wn * SFM_MAX_WP));
        TRY(dev_alloc(h, &f.wpy, Wn * SFM_MAX_WP)); TRY(dev_alloc(h, &f.wpr, Wn * SFM_MAX_WP));
        TRY(dev_alloc(h, &f.dq, Wn * SFM_MAX_WP)); TRY(dev_alloc(h, &f.dq_n, Wn));
        TRY(dev_alloc(h, &f.dest, Wn, 0xFF)); TRY(dev_alloc(h, &f.last, Wn, 0xFF));  // -1
        TRY(dev_alloc(h, &f.nodes, (size_t)W * f.cap_nodes)); TRY(dev_alloc(h, &f.n_nodes, W)); TRY(dev_alloc(h, &f.treehash, Wn));
        TRY(dev_alloc(h, &f.pair_f, (size_t)W * n1 * n1 * 3)); TRY(dev_alloc(h, &f.pair_code, (size_t)W * n1 * n1));
        TRY(dev_alloc(h, &f.g_nb, (size_t)W * SFM_MAX_AGENTS * (SFM_MAX_AGENTS / 32))); TRY(dev_alloc(h, &f.g_sh, (size_t)W * 4 * SFM_MAX_AGENTS));
        if (W > 1) {  // room for every world's obstacle segments up front (a slice cannot grow without moving the others)
            f.cap_obs = 64;
            int* nobs = nullptr;
            TRY(dev_alloc(h, &f.obs, (size_t)W * f.cap_obs * 4));
            TRY(dev_alloc(h, &nobs, W));
            f.n_obs_w = nobs;
            h->sfm_cap_obs = f.cap_obs;
            h->sfm_nobs_w.assign(W, 0);
        }
        f.p_out = f.p; f.v_out = f.v; f.dq_out = f.dq; f.dest_out = f.dest; f.last_out = f.last;  // (a step in place)
        f.nodes_out = f.nodes; f.n_nodes_out = f.n_nodes; f.treehash_out = f.treehash;
        h->sfm_ahead = cfg->relation_ped_robo != 1 && n > 0 && !h->serial;
        if (h->sfm_ahead) {
            SfmDev& o = h->sfm_other;
            o = f;
            TRY(dev_alloc(h, &o.p, Wn * 3)); TRY(dev_alloc(h, &o.v, Wn * 3));
            TRY(dev_alloc(h, &o.dq, Wn * SFM_MAX_WP)); TRY(dev_alloc(h, &o.dest, Wn, 0xFF)); TRY(dev_alloc(h, &o.last, Wn, 0xFF));
            TRY(dev_alloc(h, &o.nodes, (size_t)W * f.cap_nodes)); TRY(dev_alloc(h, &o.n_nodes, W)); TRY(dev_alloc(h, &o.treehash, Wn));
            HIPCHK_H(hipStreamCreateWithFlags(&h->sfm_stream, hipStreamNonBlocking));
            HIPCHK_H(hipEventCreateWithFlags(&h->ev_sfm, hipEventDisableTiming | hipEventDisableSystemFence));
            HIPCHK_H(hipEventCreateWithFlags(&h->ev_sfm_in[0], hipEventDisableTiming | hipEventDisableSystemFence));
            HIPCHK_H(hipEventCreateWithFlags(&h->ev_sfm_in[1], hipEventDisableTiming | hipEventDisableSystemFence));
        }
        for (int k = 0; k < W; k++) {
            HIPCHK_H(hipMemcpy(f.p + (size_t)k * n1 * 3, p0.data(), sizeof(double) * 3 * n1, hipMemcpyHostToDevice));
            HIPCHK_H(hipMemcpy(f.vmax + (size_t)k * n1, vmax.data(), sizeof(double) * n1, hipMemcpyHostToDevice));
            HIPCHK_H(hipMemcpy(f.nodes + (size_t)k * f.cap_nodes, nodes.data(), sizeof(SfmNode) * n_nodes, hipMemcpyHostToDevice));
            HIPCHK_H(hipMemcpy(f.n_nodes + k, &n_nodes, sizeof(int), hipMemcpyHostToDevice));
            HIPCHK_H(hipMemcpy(f.treehash + (size_t)k * n1, treehash.data(), sizeof(int) * n1, hipMemcpyHostToDevice));
        }
    }
    TRY(dev_alloc(h, &d.prof, 16 + 12 * (size_t)RL));
    d.state_in_integrate = P == 0 ? 1 : 0;
    TRY(dev_alloc(h, &d.tail_sig, RL));
    TRY(dev_alloc(h, &d.tail_cnt, (size_t)(RL + WAVE - 1) / WAVE * TAIL_CNT_STRIDE));
    TRY(dev_alloc(h, &d.dbg, 32));  // phase sums | per-wave (start, end, hw id, -) of k_view and k_obs

    // output arena
    ArenaPlan plan;
    plan_arena(*cfg, g, RL, plan);
    const bool full_rewrite = (cfg->flags & IMGENV_FLAG_FULL_REWRITE) != 0;
    h->arena_bytes = plan.total;
    if (full_rewrite && RL != R) {  // (a shard's caller all-gathers the records in place: that buffer cannot be a copy)
        imgenv_destroy(h);
        FAIL(IMGENV_EINVAL, "IMGENV_FLAG_FULL_REWRITE is not available in a robot shard");
    }
    if (cfg->out_arena && cfg->out_arena_bytes < (int64_t)plan.total) {
        imgenv_destroy(h);
        FAIL(IMGENV_EINVAL, "out_arena too small: %lld < %zu", (long long)cfg->out_arena_bytes, plan.total);
    }
    if (full_rewrite) {  // the caller's arena (or a second allocation) only ever receives copies; the kernels work on a private one
        if (cfg->out_arena) {
            h->pub_arena = (unsigned char*)cfg->out_arena;
        } else {
            void* p = nullptr;
            if (hipMalloc(&p, plan.total) != hipSuccess) {
                imgenv_destroy(h);
                FAIL(IMGENV_ENOMEM, "hipMalloc(%zu) for the public output arena failed", plan.total);
            }
            h->pub_arena = (unsigned char*)p;
            h->own_pub = true;
        }
        HIPCHK_H(hipMemset(h->pub_arena, 0, plan.total));
    }
    if (cfg->out_arena && !full_rewrite) {
        h->arena = (unsigned char*)cfg->out_arena;
    } else {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, plan.total);
        if (e != hipSuccess) {
            imgenv_destroy(h);
            FAIL(IMGENV_ENOMEM, "hipMalloc(%zu) for the output arena failed: %s", plan.total, hipGetErrorString(e));
        }
        h->arena = (unsigned char*)p;
        h->own_arena = true;
    }
    HIPCHK_H(hipMemset(h->arena, 0, plan.total));
    unsigned char* A = h->arena;
    imgenv_out& o = h->out;
    o.struct_size = (int32_t)sizeof(imgenv_out);
    o.n_local = RL; o.view_h = g.Hv; o.view_w = g.Wv; o.n_beams = g.B; o.state_dim = cfg->state_dim; o.ped_vec_len = d.PV;
    o.image_h = d.img_h; o.image_w = d.img_w; o.grid_h = Hg; o.grid_w = Wg;
    o.vector_states = (float*)(A + plan.off[0]);
    o.view_maps = (uint8_t*)(A + plan.off[1]);
    o.sensor_maps = (uint16_t*)(A + plan.off[2]);
    o.lasers_raw = (float*)(A + plan.off[3]);
    o.lasers = (double*)(A + plan.off[4]);
    o.ped_vector_states = (float*)(A + plan.off[5]);
    o.ped_maps = (float*)(A + plan.off[6]);
    o.is_collisions = (int8_t*)(A + plan.off[7]);
    o.is_arrives = (uint8_t*)(A + plan.off[8]);
    o.step_ds = (double*)(A + plan.off[9]);
    o.ped_min_dists = (double*)(A + plan.off[10]);
    o.base_rewards = (int32_t*)(A + plan.off[11]);
    o.base_dones = (uint8_t*)(A + plan.off[12]);
    o.rewards = (double*)(A + plan.off[13]);
    o.dones = (uint8_t*)(A + plan.off[14]);
    o.dones_info = (int32_t*)(A + plan.off[15]);
    o.is_clean = (uint8_t*)(A + plan.off[16]);
    o.robot_pose = (double*)(A + plan.off[17]);
    o.ped_state = (double*)(A + plan.off[18]);
    o.counters = (int32_t*)(A + plan.off[19]);
    d.rec = (double*)(A + plan.off[20]);
    o.paper_rewards = (double*)(A + plan.off[21]);
    d.paper_rewards = o.paper_rewards;
    o.step_rewards = (double*)(A + plan.off[22]);
    o.step_dones = (uint8_t*)(A + plan.off[23]);
    o.step_dones_info = (int32_t*)(A + plan.off[24]);
    o.step_is_clean = (uint8_t*)(A + plan.off[25]);
    o.step_is_arrives = (uint8_t*)(A + plan.off[26]);
    o.step_is_collisions = (int8_t*)(A + plan.off[27]);
    o.step_all_down = (uint8_t*)(A + plan.off[28]);
    d.step_all_down = o.step_all_down;
    if ((cfg->flags & IMGENV_FLAG_AGENT_STATE_EXTRAS) && g.B > 0) {
        o.hits_x = (float*)(A + plan.off[29]);
        o.hits_y = (float*)(A + plan.off[30]);
        o.angular_map = (float*)(A + plan.off[31]);
    } else {
        o.hits_x = o.hits_y = o.angular_map = nullptr;
    }
    d.hits_x = o.hits_x; d.hits_y = o.hits_y; d.angular_map = o.angular_map;
    d.view_max_dist32 = cfg->view_max_dist;
    d.step_rewards = o.step_rewards; d.step_dones = o.step_dones; d.step_dones_info = o.step_dones_info;
    d.step_is_clean = o.step_is_clean; d.step_is_arrives = o.step_is_arrives; d.step_is_collisions = o.step_is_collisions;
    d.vector_states = o.vector_states; d.view_maps = o.view_maps; d.sensor_maps = o.sensor_maps;
    d.lasers_raw = o.lasers_raw; d.lasers = o.lasers; d.ped_vector_states = o.ped_vector_states;
    d.ped_maps = o.ped_maps; d.is_collisions = o.is_collisions; d.is_arrives = o.is_arrives;
    d.step_ds = o.step_ds; d.ped_min_dists = o.ped_min_dists; d.base_rewards = o.base_rewards;
    d.base_dones = o.base_dones; d.rewards = o.rewards; d.dones = o.dones; d.dones_info = o.dones_info;
    d.is_clean = o.is_clean; d.robot_pose = o.robot_pose; d.ped_state = o.ped_state; d.counters = o.counters;
    {   // NearbyPed starts at +inf and is never re-initialised (reset_helper.py:85-99); is_clean starts True
        std::vector<double> inf(RL, INFINITY);
        HIPCHK_H(hipMemcpy(o.ped_min_dists, inf.data(), sizeof(double) * RL, hipMemcpyHostToDevice));
        HIPCHK_H(hipMemset(o.is_clean, 1, RL));
    }
    if (full_rewrite) {  // the same struct with every pointer moved into the public arena
        h->pub_out = o;
        const ptrdiff_t shift = h->pub_arena - h->arena;
        for (void** f : {(void**)&h->pub_out.vector_states, (void**)&h->pub_out.view_maps, (void**)&h->pub_out.sensor_maps, (void**)&h->pub_out.lasers_raw,
                         (void**)&h->pub_out.lasers, (void**)&h->pub_out.ped_vector_states, (void**)&h->pub_out.ped_maps, (void**)&h->pub_out.is_collisions,
                         (void**)&h->pub_out.is_arrives, (void**)&h->pub_out.step_ds, (void**)&h->pub_out.ped_min_dists, (void**)&h->pub_out.base_rewards,
                         (void**)&h->pub_out.base_dones, (void**)&h->pub_out.rewards, (void**)&h->pub_out.paper_rewards, (void**)&h->pub_out.dones,
                         (void**)&h->pub_out.dones_info, (void**)&h->pub_out.is_clean, (void**)&h->pub_out.robot_pose, (void**)&h->pub_out.ped_state,
                         (void**)&h->pub_out.counters, (void**)&h->pub_out.step_rewards, (void**)&h->pub_out.step_dones, (void**)&h->pub_out.step_dones_info,
                         (void**)&h->pub_out.step_is_clean, (void**)&h->pub_out.step_is_arrives, (void**)&h->pub_out.step_is_collisions,
                         (void**)&h->pub_out.step_all_down, (void**)&h->pub_out.hits_x, (void**)&h->pub_out.hits_y, (void**)&h->pub_out.angular_map})
            if (*f) *f = (unsigned char*)*f + shift;
    }
    if (cfg->flags & (IMGENV_FLAG_CHECK_OUTPUTS | IMGENV_FLAG_CHECK_OUTPUTS_FIRST)) {  // every array of imgenv_out (not the records: the caller's exchange writes those)
        if (!(cfg->flags & IMGENV_FLAG_CHECK_OUTPUTS)) h->guard_left = IMGENV_CHECK_FIRST_CALLS;
        static const char* const names[32] = {"vector_states", "view_maps", "sensor_maps", "lasers_raw", "lasers", "ped_vector_states", "ped_maps",
                                              "is_collisions", "is_arrives", "step_ds", "ped_min_dists", "base_rewards", "base_dones", "rewards", "dones",
                                              "dones_info", "is_clean", "robot_pose", "ped_state", "counters", "records", "paper_rewards", "step_rewards",
                                              "step_dones", "step_dones_info", "step_is_clean", "step_is_arrives", "step_is_collisions", "step_all_down",
                                              "hits_x", "hits_y", "angular_map"};
        std::vector<OutSpan> spans;
        int blocks = 0;
        for (int q = 0; q < plan.n; q++) {
            const size_t bytes = (q + 1 < plan.n ? plan.off[q + 1] : plan.total) - plan.off[q];  // (the padding up to the next array is zero and stays zero)
            if (q == 20 || (q >= 29 && !o.hits_x)) continue;
            OutSpan sp;
            sp.p = A + plan.off[q];
            sp.bytes = bytes;
            sp.first_block = blocks;
            blocks += (int)((bytes + OUT_SUM_CHUNK - 1) / OUT_SUM_CHUNK);
            h->span_name[spans.size()] = names[q];
            spans.push_back(sp);
        }
        h->n_spans = (int)spans.size();
        h->span_blocks = blocks;
        const OutSpan* up = nullptr;
        TRY(dev_upload(h, &up, spans));
        h->d_spans = const_cast<OutSpan*>(up);
        TRY(dev_alloc(h, &h->d_sums, 2 * (size_t)h->n_spans));
        h->guard_check = true;
    }
    h->PP = WAVE;  // sort slots of k_obs: a power of two, 64 * E of them in registers up to 1024 pedestrians
    while (h->PP < h->Pw) h->PP <<= 1;
    h->obs_E = h->PP <= 1024 ? h->PP / WAVE : 0;
    if (h->obs_E >= 2) {
        // last step's pedestrian order per robot (k_obs): any permutation of the slots will do to start from
        std::vector<uint16_t> ident((size_t)h->PP);
        for (int q = 0; q < h->PP; q++) ident[q] = q < h->Pw ? (uint16_t)q : (uint16_t)0xFFFF;
        TRY(dev_alloc(h, &d.obs_ord, (size_t)RL * h->PP));
        std::vector<uint16_t> all((size_t)RL * h->PP);
        for (int r = 0; r < RL; r++) memcpy(&all[(size_t)r * h->PP], ident.data(), sizeof(uint16_t) * (size_t)h->PP);
        HIPCHK_H(hipMemcpy(d.obs_ord, all.data(), sizeof(uint16_t) * all.size(), hipMemcpyHostToDevice));
        // (measured: 200 pedestrians 1 pass 101.8, 2-5 passes 94-96 us per headline step; 1000 pedestrians 1 / 3 / 6 / 12 passes 316 / 303 / 282 / 282 us per cfg-5 step, 294 with the full sort every step)
        d.obs_passes = h->obs_E >= 8 ? 6 : 3;
    }
    const size_t NC = (size_t)g.Hv * g.Wv;
    max_stride += 4;  // + the dummy beam of view cells that no beam crosses (kept a multiple of 16 bytes)
    d.hit_stride = (int)max_stride;
    // src u8 (+ dummy cells) | hit u32 | column terms | cursors of the final pass | largest hit step of blocks of beams (3 levels)
    h->lds_view = ((NC + 16) & ~(size_t)15) + 4 * max_stride + 16 * (size_t)g.Wv + 16 + 4 * (2 * (max_stride / 8 + 1) + 4);
    static_assert(PM_CAP * 2 <= WAVE * 7 * 4, "the touched-cell list reuses the staging buffer");
    h->lds_obs = (h->obs_E == 0 ? (size_t)h->PP * 8 : 0) + (size_t)(h->Pw > 0 ? h->Pw : 1) * 8 + (size_t)h->PP * 4 + WAVE * 7 * 4 + 16;
    if (h->big_view) {  // k_beams_big: the occupied plane of the crop bitmap; k_taps_big: the hit words
        const size_t lds_bits = 4 * (size_t)d.big_words;
        h->big_bits_in_lds = lds_bits <= 150 * 1024;  // beyond that (views above ~1000 x 1000 cells) the beams read the bitmap from HBM
        h->lds_view_big = (h->big_bits_in_lds ? lds_bits : 0) + 16;
        h->lds_view = 16;
        d.big_bits_in_lds = h->big_bits_in_lds ? 1 : 0;
        if (taps_lds_bytes(g.B) > 160 * 1024) {
            imgenv_destroy(h);
            FAIL(IMGENV_EINVAL, "the hit words of %d beams do not fit the 160 KiB LDS", g.B);
        }
        if (taps_lds_bytes(g.B) > 64 * 1024)
            HIPCHK_H(hipFuncSetAttribute((const void*)k_taps_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)taps_lds_bytes(g.B)));
    }
    if (h->lds_view_big > 64 * 1024) {
        for (const void* f : {(const void*)k_beams_big<true, true, true>, (const void*)k_beams_big<true, false, true>,
                              (const void*)k_beams_big<false, true, true>, (const void*)k_beams_big<false, false, true>})
            HIPCHK_H(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_view_big));
    }
    if (h->lds_view > 160 * 1024 || h->lds_obs > 160 * 1024) {
        imgenv_destroy(h);
        FAIL(IMGENV_EINVAL, "view (%zu B) or pedestrian list (%zu B) does not fit the 160 KiB LDS", h->lds_view, h->lds_obs);
    }
    if (h->lds_view > 64 * 1024) {
        for (const void* f : {(const void*)k_view<true, true, false, 1>, (const void*)k_view<true, false, false, 1>,
                              (const void*)k_view<false, true, false, 1>, (const void*)k_view<false, false, false, 1>,
                              (const void*)k_view<true, true, true, 1>, (const void*)k_view<true, false, true, 1>,
                              (const void*)k_view<false, true, true, 1>, (const void*)k_view<false, false, true, 1>,
                              (const void*)k_view<true, true, false, 4>, (const void*)k_view<true, false, false, 4>,
                              (const void*)k_view<false, true, false, 4>, (const void*)k_view<false, false, false, 4>,
                              (const void*)k_view<true, true, true, 4>, (const void*)k_view<true, false, true, 4>,
                              (const void*)k_view<false, true, true, 4>, (const void*)k_view<false, false, true, 4>})
            HIPCHK_H(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_view));
    }
    if (cfg->ped_scene_type == IMGENV_SCENE_PEDSIM)
        HIPCHK_H(hipFuncSetAttribute((const void*)k_sfm, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(SfmNode) * SFM_LDS_NODES)));
    if (h->lds_obs > 64 * 1024)
        HIPCHK_H(hipFuncSetAttribute((const void*)k_obs<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_obs));
    {   // early-observation steps: worlds owned whole, an ORCA crowd that nothing but the solve moves (no beep lottery), no
        // limiter history to carry, views through k_view -- the headline shape and cfg-5; everything else keeps k_obs behind the move
        const bool limiters = cfg->limiter_v.has_velocity_limits || cfg->limiter_v.has_acceleration_limits || cfg->limiter_v.has_jerk_limits ||
                              cfg->limiter_w.has_velocity_limits || cfg->limiter_w.has_acceleration_limits || cfg->limiter_w.has_jerk_limits;
        // The early k_obs always waits behind the GATE (world.h: sync): it reads the step's actions and REWRITES output arrays, so it
        // must run behind everything the caller queued on its stream in front of the step -- a policy that writes the actions, a
        // copy of the last observation into a replay buffer, FULL_REWRITE's own copy.  (Rounds 4-5 also had an ungated variant for
        // callers that promised complete actions, IMGENV_STEP_ACTIONS_READY; it only waited for the last chain's views, which does
        // not cover readers of the outputs -- and measured the same as the gate, 92.0 against 92.3 us per step.  Dropped in round 6.)
        // ... or a social-force crowd that is stepped a step ahead (sfm_ahead): its arrays are published in front of the move and
        // stand still during the step, so the early k_obs reads them as they are (obs_early = 2); the gate also says that the
        // publishing is done
        // (robot shards too, since round 6: k_obs reads the rank's own robots' snapshots and the replicated pedestrians')
        h->early = !h->serial && P > 0 && (h->NA > 0 || h->sfm_ahead) && !d.beep_on && !limiters && !h->big_view &&
                   h->n_sub >= 1 && h->n_sub + 2 <= INT_ITEMS;
        if (h->early) {
            TRY(dev_alloc(h, &d.sync, 8));
            TRY(dev_alloc(h, &h->rec_snap[0], (size_t)RL * IMGENV_RECORD_DOUBLES));
            TRY(dev_alloc(h, &h->rec_snap[1], (size_t)RL * IMGENV_RECORD_DOUBLES));
            TRY(dev_alloc(h, &h->ped_snap[0], (size_t)(h->NA > 0 ? P : 1)));
            TRY(dev_alloc(h, &h->ped_snap[1], (size_t)(h->NA > 0 ? P : 1)));
        }
    }
    HIPCHK_H(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    // (Round 6 tried the observation's stream on a subset of the compute units, hipExtStreamCreateWithCUMask, so that the move's
    // successors find free units at once: such a stream is a BLOCKING one -- it serialises against a caller on the null stream, 95 ->
    // 213 us per step whatever the mask -- and beside a caller on a stream of its own 24 / 28 / 16 of an XCD's 32 units cost 99.2 /
    // 99.5 / 108.3 us against 95.5: the step is bound by the instructions it issues, not by where they run.  docs/HISTORY.md)
    HIPCHK_H(hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking));
    if (h->early) {  // gates (world.h: sync) only where kernels of two streams really run side by side: k_gate_probe
        static int gates_work = -1;  // (per process)
        if (gates_work < 0) {
            k_gate_probe<<<dim3(1), dim3(WAVE), 0, h->side2>>>(d.sync + 2);
            k_gate_probe_set<<<dim3(1), dim3(1), 0, h->side>>>(d.sync + 2);
            HIPCHK_H(hipStreamSynchronize(h->side2));
            HIPCHK_H(hipStreamSynchronize(h->side));
            uint32_t verdict = 0;
            HIPCHK_H(hipMemcpy(&verdict, d.sync + 3, sizeof(verdict), hipMemcpyDeviceToHost));
            gates_work = verdict == 1u ? 1 : 0;
        }
        h->gates_work = gates_work == 1;
    }
    d.view_prio = (P == 0 || (h->early && h->gates_work && h->NA > 0 && h->obs_E >= 1 && h->obs_E <= 4)) ? 1 : 0;  // (world.h)
    HIPCHK_H(hipEventCreateWithFlags(&h->ev_fork2, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCHK_H(hipEventCreateWithFlags(&h->ev_join2, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCHK_H(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCHK_H(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCHK_H(hipDeviceSynchronize());
    *out = h;
    return IMGENV_OK;
}

// ---------------------------------------------------------------------------------------- reset
struct ResetRobot {  // per local robot
    double gx, gy;
    Tf2 world_target;
};

// the t-th robot of a reset (list order): pose3 / rr hold the reset's robots in that order, in page-locked host memory
__device__ __forceinline__ void reset_robot(const DevWorld& w, const int* list, int t, const double* __restrict__ pose3,
                                            const ResetRobot* __restrict__ rr, int whole) {
    const int i = list ? list[t / w.Rw] * w.Rw + t % w.Rw : t;
    double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
    r[0] = pose3[5 * t];  // init_pose (agent.cpp:133-142); Agent::vx, vy persist across resets
    r[1] = pose3[5 * t + 1];
    r[2] = pose3[5 * t + 2];
    r[5] = pose3[5 * t + 3];  // sin / cos of yaw/2, evaluated on the host
    r[6] = pose3[5 * t + 4];
    const int l = i - w.r0;
    if (l >= 0 && l < w.RL) {
        const ResetRobot q = rr[w.W > 1 ? t : l];
        w.l0v[l] = 0;  // last0_vw_ = (0,0); last1_vw_ is not touched by init_pose
        w.l0w[l] = 0;
        w.gx[l] = q.gx;
        w.gy[l] = q.gy;
        w.world_target[l] = q.world_target;
        w.is_coll[l] = 0;
        w.is_arr[l] = 0;
    }
    if (t == 0 && whole) w.counters[2] = 0;  // frozen robot-steps since this reset (of every world)
}

// reset of a robot-sharded world: bounding box of the local robots' new positions (tail_group re-arms it every step)
__global__ void k_reset_bbox(DevWorld w, const double* __restrict__ pose3) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = l < w.RL;
    const double* p = pose3 + 5 * (size_t)(w.r0 + (valid ? l : 0));
    bbox_accumulate(w, valid, p[0], p[1]);
}

__device__ __forceinline__ void reset_ped(const DevWorld& w, const int* list, int t, const double* __restrict__ pose3) {
    const int j = list ? list[t / w.Pw] * w.Pw + t % w.Pw : t;
    const double x = pose3[3 * t], y = pose3[3 * t + 1];
    w.ppx[j] = x;
    w.ppy[j] = y;
    w.pyaw[j] = pose3[3 * t + 2];
    w.ptraj_idx[j] = 0;
    if (w.NA > 0) {  // setPedPos (rvoscene.h:32-34); the agent's velocity persists
        w.apx[j] = (float)x;
        w.apy[j] = (float)y;
    }
    if (w.scene == IMGENV_SCENE_PEDSIM) {  // setPosition(x, y, 0) (pedscene.h:34-36); velocity persists
        const int k = w.W > 1 ? j / w.Pw : 0;
        double* p = w.sfm.p + 3 * ((size_t)k * w.sfm.n + (size_t)(j - k * w.sfm.n_peds));  // its world's crowd
        p[0] = x;
        p[1] = y;
        p[2] = 0.0;
    }
    w.ped_state[4 * j] = x;
    w.ped_state[4 * j + 1] = y;
    w.ped_state[4 * j + 2] = w.pvx[j];
    w.ped_state[4 * j + 3] = w.pvy[j];
}

// device-side overflow flags (raised by kernels, see world.h / sfm.h): turn them into an error at the next API call
static int check_device_flags(imgenv* h) {
    const volatile int* e = h->err_host;
    if (!e) return 0;
    if (e[0]) FAIL(IMGENV_EDEVICE, "ORCA: a pedestrian sees more obstacle segments than the neighbour scratch holds");
    if (e[1]) FAIL(IMGENV_EDEVICE, "ORCA: obstacle BSP walk overflowed its stack");
    if (e[2])
        FAIL(IMGENV_EDEVICE, "device-side auto-reset: a finished world could not be given its placement (code %d: 100 the pool did not hold it; "
                             "1 a fixed start with a random target; 2-4 no admissible placement within 200000 draws; 10-16 the RVO obstacle tree "
                             "outgrew its scratch)", e[2]);
    if (e[7]) FAIL(IMGENV_EDEVICE, "the observation's gate gave up after 60 s: the caller's stream never reached the step's move (the stream is stuck behind something)");
    if (e[6] == 10) FAIL(IMGENV_EDEVICE, "class layer (counts) in a robot shard: a footprint cell fell outside the record's bitmap (internal)");
    if (e[6] == 9)
        FAIL(IMGENV_EDEVICE, "class layer (counts): more robot footprints on one cell than the layer's count field holds (at least 63; robots "
                             "were placed on top of each other) -- create the handle with IMGENV_FLAG_COMPOSE_DENSE, whose owner layers have no such limit");
    if (e[6]) FAIL(IMGENV_EDEVICE, "class layer (counts): a pedestrian's footprint left its raster box or its cell list (code %d)", e[6]);
    if (e[4])
        FAIL(IMGENV_EDEVICE, "pedscene: the social-force quadtree overflowed (code %d: 1 leaf capacity, 2 node pool, 3 / 4 depth, 5 a leaf lock never came free, "
                             "6 an agent a split re-homed is on the other side of its new leaf's edge than its own move assumed) -- more than "
                             "8 agents piled up outside the tree's 10 m x 10 m root square, where the reference recurses forever "
                             "(ped_tree.cpp:65-96)", e[4]);
    return 0;
}

// fork: the pedestrian half of the observation needs the local robots' new poses only, so it starts right behind
// k_integrate (in a sharded world: underneath the record exchange) on its own stream
// what the tails need, whichever kernel ends up running them (tail_group)
static void set_tail_fields(imgenv* h, int is_reset, int tail_elapsed) {
    DevWorld& d = h->d;
    d.tail_fused = h->P > 0 ? 1 : 0;
    d.tail_is_reset = is_reset;
    d.tail_elapsed = tail_elapsed;
}

// The fused tails (tail_group in kernels.h) rely on tail_sig / tail_cnt being zero when a chain of launches starts; the chain
// itself re-zeroes them.  A chain that was abandoned half-way (an error between imgenv_step_begin and imgenv_step_end, a failing
// launch) leaves stale bits behind: the next chain clears the words first.
static int chain_begin(imgenv* h, hipStream_t st) {
    if (h->chain_open) {
        HIPCHK(hipMemsetAsync(h->d.tail_sig, 0, sizeof(unsigned long long) * (size_t)h->RL, st));
        HIPCHK(hipMemsetAsync(h->d.tail_cnt, 0, sizeof(int) * (size_t)(h->RL + WAVE - 1) / WAVE * TAIL_CNT_STRIDE, st));
    }
    h->chain_open = true;
    return 0;
}

// ---- output guards.  The arrays behind imgenv_outputs() are the kernels' incremental working copies (a step rewrites what can
// change), so a caller that writes into them -- an in-place normalisation in a trainer -- would corrupt every later observation
// silently, where the reference hands out fresh copies (ROS responses, img_env.cpp:745-749).  Two opt-in answers:
// IMGENV_FLAG_CHECK_OUTPUTS seals every array with a checksum at the end of a chain of launches and verifies it at the start
// of the next call (debug: it synchronises); IMGENV_FLAG_FULL_REWRITE hands out a second arena that receives a complete copy
// of every array at the end of every chain, so nothing the caller does to it can reach the kernels.
static int outputs_seal(imgenv* h, hipStream_t st) {
    if (h->pub_arena) HIPCHK(hipMemcpyAsync(h->pub_arena, h->arena, h->arena_bytes, hipMemcpyDeviceToDevice, st));
    if (h->guard_check) {
        HIPCHK(hipMemsetAsync(h->d_sums, 0, sizeof(unsigned long long) * (size_t)h->n_spans, st));
        k_out_sum<<<dim3(h->span_blocks), dim3(256), 0, st>>>(h->d_spans, h->n_spans, h->d_sums);
        h->guard_sealed = true;
    }
    return 0;
}
static int outputs_verify(imgenv* h, hipStream_t st) {
    if (!h->guard_check || !h->guard_sealed) return 0;
    unsigned long long* found = h->d_sums + h->n_spans;
    HIPCHK(hipMemsetAsync(found, 0, sizeof(unsigned long long) * (size_t)h->n_spans, st));
    k_out_sum<<<dim3(h->span_blocks), dim3(256), 0, st>>>(h->d_spans, h->n_spans, found);
    k_out_verify<<<dim3(1), dim3(64), 0, st>>>(h->d_sums, found, h->n_spans, h->d.err);
    HIPCHK(hipStreamSynchronize(st));
    const int f = h->err_host ? h->err_host[5] : 0;
    if (h->guard_left > 0 && --h->guard_left == 0) {  // (IMGENV_FLAG_CHECK_OUTPUTS_FIRST: this was the last look)
        h->guard_check = false;
        h->guard_sealed = false;
    }
    if (f) {
        h->err_host[5] = 0;
        h->guard_sealed = false;  // (the next chain seals what it finds; the caller has been told)
        FAIL(IMGENV_EINVAL, "the caller wrote into imgenv_out.%s since the last call: the output arrays are the library's working copies "
                            "and read-only (include/imgenv.h); create the handle with IMGENV_FLAG_FULL_REWRITE to receive copies instead",
             h->span_name[f - 1]);
    }
    return 0;
}

static int launch_obs_kernel(imgenv* h, hipStream_t s_obs) {
    DevWorld& d = h->d;
    const dim3 go(d.act_nl), bo(WAVE);
    const size_t lds_obs = h->lds_obs;
    switch (h->obs_E) {
        case 1: TIMED(h, IMGENV_K_OBS, s_obs, (k_obs<1><<<go, bo, lds_obs, s_obs>>>(d, h->PP))); break;
        case 2: TIMED(h, IMGENV_K_OBS, s_obs, (k_obs<2><<<go, bo, lds_obs, s_obs>>>(d, h->PP))); break;
        case 4: TIMED(h, IMGENV_K_OBS, s_obs, (k_obs<4><<<go, bo, lds_obs, s_obs>>>(d, h->PP))); break;
        case 8: TIMED(h, IMGENV_K_OBS, s_obs, (k_obs<8><<<go, bo, lds_obs, s_obs>>>(d, h->PP))); break;
        // (four wavefronts per robot for 513 .. 1024 pedestrians -- three times the occupancy, a third of the LDS per wavefront --
        // were measured in round 6 and LOSE: cfg-5 277 -> 325 us per step, 321 with two, 406 with eight: the step is bound by the
        // instructions it issues, k_obs beside k_view, not by this kernel's occupancy.  docs/HISTORY.md)
        case 16: TIMED(h, IMGENV_K_OBS, s_obs, (k_obs<16><<<go, bo, lds_obs, s_obs>>>(d, h->PP))); break;
        default: TIMED(h, IMGENV_K_OBS, s_obs, (k_obs<0><<<go, bo, lds_obs, s_obs>>>(d, h->PP))); break;
    }
    h->launches += 1;
    return 0;
}

static int launch_obs(imgenv* h, hipStream_t st) {
    if (int rc = chain_begin(h, st)) return rc;
    DevWorld& d = h->d;
    const bool overlap = !h->serial;
    hipStream_t s_obs = overlap ? h->side2 : st;
    if (overlap) {
        if (!h->fork_on_move) HIPCHK(hipEventRecord(h->ev_fork, st));
        h->fork_on_move = false;
        HIPCHK(hipStreamWaitEvent(s_obs, h->ev_fork, 0));
    }
    if (int rc = launch_obs_kernel(h, s_obs)) return rc;
    h->obs_forked = true;
    return 0;
}

// The rasters of a chain of launches (in front of them, in STAMP mode, every STAMP_TAGS steps the sweep): every robot of the launch
// and every pedestrian -- or, local_only (a step of a robot shard in SUM mode, world.h: sum_shard), this rank's robots and the
// pedestrians: the other ranks' robots follow behind the exchange (k_remote)
static int launch_rasters(imgenv* h, hipStream_t st, int is_reset, bool moved, bool local_only, bool from_begin = false) {
    DevWorld& d = h->d;
    const int keep_ng = d.act_ng;
    if (local_only) {
        d.act_ng = h->RL;
        d.act_g0 = h->r0;
    }
    const int n_g = d.act_ng, n_p = d.act_np;
    // k_compose / k_cell_base: 4 cells per thread over everything, or a fixed number of 256-thread blocks per listed world
    const unsigned compose_blocks = d.act_list ? (unsigned)(((h->Gs / 4 + 255) / 256) * d.act_nw) : (unsigned)((d.act_cells / 4 + 255) / 256 + 1);
    // STAMP mode: no compose.  A reset has given the worlds it covers their base classes together with their obstacle maps
    // (k_reset_apply, k_reset_obstacles); every STAMP_TAGS steps one sweep drops all stamps before their tags come round again.
    if (h->stamp && !is_reset && h->stamp_seq % STAMP_TAGS == 0)
        TIMED(h, IMGENV_K_COMPOSE, st, (k_cell_base<<<dim3(compose_blocks), dim3(256), 0, st>>>(d)));
    const int n_blocks = n_p > n_g ? n_p : n_g;
    // four wavefronts per robot / pedestrian when the launch cannot fill the machine (device-side auto-reset: by the expected
    // number of robots, the grid itself is sized for every world)
    const bool small = (d.act_n_dev ? std::min(n_blocks, h->act_hint) : n_blocks) <= 1024;
    // robots and pedestrians in blocks of their own while all of them fit the chip at once (8192 wavefronts): a robot and a
    // pedestrian one behind the other in one block is twice a block's chain of memory round trips
    // (1024 envs x (4 + 3): k_raster 34 -> 22 us; the headline's 8192 + 200 stay as they are: a second, nearly empty round)
    const bool roomy = !small && n_g + n_p <= 8192;
    const int split = (small || roomy) && n_g > 0 && n_p > 0 ? n_g : 0;
    const dim3 gr(split || moved ? n_g + n_p : n_blocks), br(small ? 4 * WAVE : WAVE);
    const size_t lds = 4 * (size_t)d.box_cells + 16;
    const int variant = (h->pow2 ? 3 : 0) + (h->stamp ? 1 : h->sum ? 2 : 0);
    // (k_move_raster: the step's move in the same launch -- the RVO / recorded pedestrians' too; a social-force crowd has moved in k_sfm)
    // (imgenv_step_end has counted the step when it launches this; imgenv_step_begin has not yet)
    const int move_peds = h->P > 0 && (h->NA > 0 || h->cfg.ped_scene_type == IMGENV_SCENE_DATASET) ? 1 : 0, step_now = from_begin ? h->elapsed : h->elapsed - 1;
#define RASTER_CASE(N, P2, LM)                                                                                        \
    case N:                                                                                                           \
        if (moved && small) TIMED(h, IMGENV_K_MOVE_RASTER, st, (k_move_raster<P2, LM, 4><<<gr, br, lds, st>>>(d, h->move_actions, h->n_sub, step_now, move_peds))); \
        else if (moved) TIMED(h, IMGENV_K_MOVE_RASTER, st, (k_move_raster<P2, LM, 1><<<gr, br, lds, st>>>(d, h->move_actions, h->n_sub, step_now, move_peds))); \
        else if (small) TIMED(h, IMGENV_K_RASTER, st, (k_raster<P2, LM, 4><<<gr, br, lds, st>>>(d, is_reset, split)));     \
        else TIMED(h, IMGENV_K_RASTER, st, (k_raster<P2, LM, 1><<<gr, br, lds, st>>>(d, is_reset, split)));               \
        break;
    switch (variant) {
        RASTER_CASE(5, true, 2)
        RASTER_CASE(4, true, 1)
        RASTER_CASE(3, true, 0)
        RASTER_CASE(2, false, 2)
        RASTER_CASE(1, false, 1)
        RASTER_CASE(0, false, 0)
    }
#undef RASTER_CASE
    d.act_ng = keep_ng;
    d.act_g0 = 0;
    return 0;
}

static int launch_views(imgenv* h, hipStream_t st, int is_reset) {
    DevWorld& d = h->d;
    const int n_g = d.act_ng, n_p = d.act_np, n_l = d.act_nl;
    // k_compose: 4 cells per thread over everything, or a fixed number of 256-thread blocks per listed world
    set_tail_fields(h, is_reset, h->elapsed);
    if (h->P == 0)
        if (int rc = chain_begin(h, st)) return rc;
    const unsigned compose_blocks = d.act_list ? (unsigned)(((h->Gs / 4 + 255) / 256) * d.act_nw) : (unsigned)((d.act_cells / 4 + 255) / 256 + 1);
    // the rasters: in a step whose move was left to them (imgenv_step_begin: k_move_raster) they come first and the side stream forks
    // behind them, otherwise behind the side launches; a robot shard in SUM mode has drawn its own robots in imgenv_step_begin, in
    // front of the exchange, and takes the other ranks' from their records now (k_remote)
    const bool moved = h->move_pending;
    h->move_pending = false;
    const bool remote_only = h->d.sum_shard && !is_reset;
    // what the side streams read of the OTHER ranks' robots: their RVO agents (k_side_robots, k_orca) -- nothing when the crowd
    // ignores the robots or is no RVO crowd
    const bool side_reads_all = h->P > 0 && h->NA > 0 && d.relation == 1;
    auto rasters = [&]() -> int {
        if (!remote_only) return launch_rasters(h, st, is_reset, moved, false);
        return 0;  // (k_remote went out in front of the side launches: below)
    };
    if (remote_only) {
        // the other ranks' robots, right behind the exchange; "the records are complete" for the side streams rides on this kernel's
        // dispatch packet (a hipEventRecord behind the exchange is a packet of its own on the caller's stream: ~6 us)
        const unsigned nb = (unsigned)((h->R - h->RL + 255) / 256);
        const hipEvent_t ev = side_reads_all && !h->serial ? h->ev_fork2 : nullptr;
        if (h->pow2) TIMED(h, IMGENV_K_REMOTE, st, (hipExtLaunchKernelGGL(k_remote<true>, dim3(nb), dim3(256), 0, st, nullptr, ev, 0, d)));
        else TIMED(h, IMGENV_K_REMOTE, st, (hipExtLaunchKernelGGL(k_remote<false>, dim3(nb), dim3(256), 0, st, nullptr, ev, 0, d)));
        h->launches += 1;
    }
    if (moved)
        if (int rc = rasters()) return rc;
    if (h->P > 0) {
        // One fork and one join per step on the caller's stream (every event operation costs it a ~6 us dependency
        // bubble).  Beside the rasters, compose and view run, on two side streams, the pedestrian half of the
        // observation and the next step's _step_ped_normal solve (img_env.cpp:304-343); both need poses only.  The
        // observation stream finally waits for the solve, so its join event covers both.
        // (IMGENV_SERIAL=1 in the environment keeps everything on the caller's stream: clean per-kernel timings.)
        const bool overlap = !h->serial;
        // Handles of at most 4096 robots keep ONE side stream: observation, robot records and the solve one behind the other (they
        // fit underneath the rasters + views with room to spare: 31 us against 72 at 1024 envs x (4 + 3)), which saves three of the
        // seven event operations of a phase -- such shapes are bound by the host's call rate (tools/host_issue_probe.py:
        // ~6 us per launch or event call, ~30 calls per step with a device-side reset)
        const bool one_side = overlap && !d.sharded && h->RL <= 4096;
        hipStream_t s_orca = overlap ? (one_side ? h->side2 : h->side) : st;
        if (!h->obs_forked)
            if (int rc = launch_obs(h, st)) return rc;
        h->obs_forked = false;
        hipStream_t s_obs = overlap ? h->side2 : st;
        if (h->early_step && one_side) HIPCHK(hipStreamWaitEvent(s_orca, h->ev_fork, 0));  // (k_obs went out with the step, in front of the move: the solve waits for it)
        if (overlap && !one_side) {
            if (d.sharded && side_reads_all) {  // the solve needs every rank's robots: a second fork behind the exchange
                if (!remote_only) HIPCHK(hipEventRecord(h->ev_fork2, st));  // (k_remote carries it otherwise)
                HIPCHK(hipStreamWaitEvent(s_orca, h->ev_fork2, 0));
            } else {
                HIPCHK(hipStreamWaitEvent(s_orca, h->ev_fork, 0));
            }
        }
        // handles of several worlds with RVO crowds: k_orca does k_side_robots' part for its world itself (one launch less per phase)
        const bool fold_side = h->W > 1 && h->NA > 0;
        if (!fold_side) {
            // (slices of >= 48 pedestrians, four at most: 8192 robots x 200 pedestrians = 128 x 4 wavefronts.  More of them -- cfg-5's
            // 1000 pedestrians in 16 slices -- only take issue slots from the rasters and the views: 281-286 us per step against 276-278)
            const int rvo_agents = h->NA > 0 && d.relation == 1, slices = rvo_agents && h->W == 1 ? std::min(4, std::max(1, (h->P + 47) / 48)) : 1;
            k_side_robots<<<dim3((unsigned)((n_g + WAVE - 1) / WAVE) * (unsigned)slices), dim3(WAVE), 0, s_orca>>>(d, is_reset, rvo_agents, slices);
            h->launches += 1;
        }
        if (h->NA > 0) {
            // groups of up to 4 pedestrians of one world per wavefront; an agent's LDS scratch sized by the largest obstacle table
            // any world of the handle can hold, the table itself staged into LDS when it fits 256 segments
            OrcaLaunch L;
            const int per_world = h->W > 1 ? h->Pw : h->P, cap = std::max(std::max(h->cap_obst, d.n_obst), 1);
            L.G = per_world > 2 ? 4 : per_world;  // a row of 16 lanes per agent
            L.groups = (per_world + L.G - 1) / L.G;
            L.cap_on = std::max(std::min(ORCA_MAX_ON, cap), ORCA_ROW - ORCA_MAX_AN);  // (a round's 16 candidate lines borrow the projection area)
            L.cap_stack = std::min(ORCA_STACK, cap + 1);
            L.fold_side = fold_side ? 1 : 0;
            L.zero_vel = is_reset;
            L.stage_obst = std::min(cap, 256);  // (a world with more segments than that is solved out of HBM: the kernel checks its count)
            const unsigned blocks = (unsigned)((n_p / per_world) * L.groups);
            // (world.h: what the next step's early k_obs reads.  A launch over every world writes one buffer and the next one the
            // other; a launch over some worlds -- a reset chain, nothing else is in flight -- writes both and does not take a turn)
            const bool partial = d.act_list != nullptr;
            d.ped_snap_out = h->early ? h->ped_snap[h->orca_seq & 1] : nullptr;
            d.ped_snap_out2 = h->early && partial ? h->ped_snap[(h->orca_seq + 1) & 1] : nullptr;
            TIMED(h, IMGENV_K_ORCA, s_orca, (k_orca<<<dim3(blocks), dim3(WAVE), orca_lds_bytes(L), s_orca>>>(d, L)));
            if (!partial) h->orca_seq += 1;
            h->launches += 1;
        }
        if (overlap) {
            if (!one_side) {
                HIPCHK(hipEventRecord(h->ev_join, s_orca));
                HIPCHK(hipStreamWaitEvent(s_obs, h->ev_join, 0));
            }
            HIPCHK(hipEventRecord(h->ev_join2, s_obs));
        }
    }
    if (h->P == 0 && is_reset) {  // no side streams: after a reset Agent::get_state gets its own small launch (in a step k_integrate does it, with pedestrians k_side_robots)
        k_state<<<dim3((n_l + 127) / 128), dim3(128), 0, st>>>(d);
        h->launches += 1;
    }
    if (!moved)
        if (int rc = rasters()) return rc;
    if (!h->stamp && !h->sum) TIMED(h, IMGENV_K_COMPOSE, st, (k_compose<<<dim3(compose_blocks), dim3(256), 0, st>>>(d)));
    if (h->big_view) {  // view_big.h: crop (tiles of every robot spread over the chip) -> beams (a workgroup per robot and 256
                        // beams) -> the shrunk sensor_map (a thread per pixel) -> the full view, only where it is an output
        const int quarters = std::max(1, (d.B + VBB_T - 1) / VBB_T), tap_chunks = (d.img_w * d.img_h + VBT_T - 1) / VBT_T;
        const bool full = d.keep_view_maps || !d.resize;
        // tiles per wavefront: 8 while the launch is a handful of robots (a reset of a few worlds: every robot on ~40 workgroups), 32-64
        // once there are enough robots to fill the chip anyway
        // (measured again after the kernel's gathers stopped binding it: a wavefront's prologue -- pose, fixed-point terms, its tiles'
        // corner records -- is worth ~8 tiles, so even 256 robots want 32 tiles per wavefront: 36 -> 29 us; 2048 robots 64: 152 -> 133.
        // Handing the fixed-point terms over from the robot's raster instead of recomputing them per wavefront was measured too:
        // 155 us at 2048 robots, i.e. worse -- the prologue's cost is its loads, not its arithmetic.)
        const int n_eff = d.act_n_dev ? std::min(n_l, h->act_hint) : n_l;
        const int tpw = n_eff >= 1024 ? 64 : n_eff >= 48 ? 32 : 8, crop_chunks = (h->big_max_crop + (VBC_T / WAVE) * tpw - 1) / ((VBC_T / WAVE) * tpw);
        // (2048 robots x 1000 beams: one block of 256 beams per workgroup 98 us, two 87, four 87 -- but end to end two win: 3.85 M robot-steps/s
        // against 3.79 / 3.78: the first workgroup of a robot also hands the collision code to the step's tail)
        const int qpw = n_eff >= 1024 ? std::min(2, quarters) : 1;
        const dim3 gc((unsigned)((n_l + 7) / 8 * 8) * (unsigned)crop_chunks), gb((unsigned)n_l * (unsigned)((quarters + qpw - 1) / qpw));
        const dim3 gf((unsigned)n_l * (unsigned)h->big_full_chunks);
        static_assert(VBT_T == TAP_CHUNK_PIXELS, "host_tables.h lists k_taps_big's chunks");
        if (h->stamp && d.crop_map) TIMED(h, IMGENV_K_CROP, st, (k_crop_big<true, true><<<gc, dim3(VBC_T), 0, st>>>(d, crop_chunks, n_l, tpw)));
        else if (h->stamp) TIMED(h, IMGENV_K_CROP, st, (k_crop_big<true, false><<<gc, dim3(VBC_T), 0, st>>>(d, crop_chunks, n_l, tpw)));
        else TIMED(h, IMGENV_K_CROP, st, (k_crop_big<false, false><<<gc, dim3(VBC_T), 0, st>>>(d, crop_chunks, n_l, tpw)));
        const int variant = (h->pow2 ? 4 : 0) | (h->stamp ? 2 : 0) | (h->big_bits_in_lds ? 1 : 0);
#define BEAMS_CASE(N, P2, ST, LB) \
    case N: TIMED(h, IMGENV_K_VIEW, st, (k_beams_big<P2, ST, LB><<<gb, dim3(VBB_T), h->lds_view_big, st>>>(d, quarters, qpw))); break;
        switch (variant) {
            BEAMS_CASE(7, true, true, true)
            BEAMS_CASE(6, true, true, false)
            BEAMS_CASE(5, true, false, true)
            BEAMS_CASE(4, true, false, false)
            BEAMS_CASE(3, false, true, true)
            BEAMS_CASE(2, false, true, false)
            BEAMS_CASE(1, false, false, true)
            BEAMS_CASE(0, false, false, false)
        }
#undef BEAMS_CASE
        // the last kernel of the chain commits the robots' new is_collision_
        // (a step only runs the chunks of pixels a beam can reach: static list per class; the chunks behind the sensor hold their
        // 200 / 100 since the reset)
        const bool listed = !is_reset && h->big_tap_chunks_dyn > 0 && h->big_tap_chunks_dyn < tap_chunks;
        const int tap_wgs = listed ? h->big_tap_chunks_dyn : tap_chunks;
        if (d.resize) TIMED(h, IMGENV_K_TAPS, st, (k_taps_big<<<dim3((unsigned)n_l * (unsigned)tap_wgs), dim3(VBT_T), taps_lds_bytes(d.B), st>>>(d, tap_wgs, full ? 0 : 1, listed ? 1 : 0)));
        if (full) TIMED(h, IMGENV_K_FULLVIEW, st, (k_fullview_big<<<gf, dim3(VBF_T), 16 * (size_t)((d.B + 4) / 4), st>>>(d, h->big_full_chunks, 1)));
        h->launches += (d.resize ? 1 : 0) + (full ? 1 : 0);
    } else {
        // one wavefront per robot when the launch fills the machine, four when it is small (a reset of a few worlds): then
        // the single wavefront's latency is all there is
        // ... and whenever a view's LDS (crop + hit words + column table: 15 KB at 96 x 96 cells and 720 beams) would leave a
        // compute unit with 16 or fewer one-wavefront workgroups -- four or fewer wavefronts per SIMD where the registers allow
        // eight: four wavefronts then share one view's LDS (cfg-5, 8192 robots: k_view 252 -> 179 us alone, the step 472 -> 377 us;
        // at 48 x 48 cells, 5 KB and 32 workgroups per unit, it loses: 63 -> 87 us)
        const bool lds_bound = (160 * 1024) / ((h->lds_view + 1279) / 1280 * 1280) <= 16;
        const bool small = lds_bound || (d.act_n_dev ? std::min(n_l, h->act_hint) : n_l) <= 1024;
        // two wavefronts per robot in between (1025-4096 robots: every wavefront still resident at once; 1024 envs x 4: 39 -> 28 us)
        const int n_view = d.act_n_dev ? std::min(n_l, h->act_hint) : n_l;
        const bool two = !small && n_view <= 4096;
        // ... and eight where a launch is at most 1024 robots and the view small (48 x 48 cells and 360 beams are then ONE round of groups
        // and ONE round of beams per wavefront: cfg-2 k_view 21.6 -> 20.4 us)
        const bool eight = small && !lds_bound;
        const dim3 gv(n_l), bv(eight ? 8 * WAVE : small ? 4 * WAVE : two ? 2 * WAVE : WAVE);
        {   // (world.h: what the next step's early k_obs reads; turns as for ped_snap)
            const bool partial = d.act_list != nullptr;
            d.rec_snap_out = h->early ? h->rec_snap[h->view_seq & 1] : nullptr;
            d.rec_snap_out2 = h->early && partial ? h->rec_snap[(h->view_seq + 1) & 1] : nullptr;
            if (!partial) h->view_seq += 1;
        }
        const int variant = (h->pow2 ? 4 : 0) | (h->geom.Wv % 4 == 0 ? 2 : 0) | (h->stamp ? 1 : 0);
#define VIEW_CASE(N, P2, A4_, ST)                                                                                               \
    case N:                                                                                                                     \
        if (eight) TIMED(h, IMGENV_K_VIEW, st, (hipExtLaunchKernelGGL((k_view<P2, A4_, ST, 8>), gv, bv, (uint32_t)h->lds_view, st, nullptr, nullptr, 0, d)));      \
        else if (small) TIMED(h, IMGENV_K_VIEW, st, (hipExtLaunchKernelGGL((k_view<P2, A4_, ST, 4>), gv, bv, (uint32_t)h->lds_view, st, nullptr, nullptr, 0, d))); \
        else if (two) TIMED(h, IMGENV_K_VIEW, st, (hipExtLaunchKernelGGL((k_view<P2, A4_, ST, 2>), gv, bv, (uint32_t)h->lds_view, st, nullptr, nullptr, 0, d)));   \
        else TIMED(h, IMGENV_K_VIEW, st, (hipExtLaunchKernelGGL((k_view<P2, A4_, ST, 1>), gv, bv, (uint32_t)h->lds_view, st, nullptr, nullptr, 0, d)));            \
        break;
        switch (variant) {
            VIEW_CASE(7, true, true, true)
            VIEW_CASE(6, true, true, false)
            VIEW_CASE(5, true, false, true)
            VIEW_CASE(4, true, false, false)
            VIEW_CASE(3, false, true, true)
            VIEW_CASE(2, false, true, false)
            VIEW_CASE(1, false, false, true)
            VIEW_CASE(0, false, false, false)
        }
#undef VIEW_CASE
    }
    // no launch for the per-robot scalars: the k_view / k_obs wavefront that completes a group of 64 robots runs them
    // (tail_group).  The caller's stream ends the step behind both side streams
    h->early_step = false;
    if (h->P > 0 && !h->serial) HIPCHK(hipStreamWaitEvent(st, h->ev_join2, 0));
    h->launches += 3;
    HIPCHK(hipGetLastError());
    h->chain_open = false;
    return outputs_seal(h, st);
}

// ---- pinned staging for reset: everything a reset uploads goes through page-locked chunks owned by the handle and
// reaches the device in ONE launch (a table of segments), however many worlds the reset covers ----
struct StageSeg {
    unsigned char* dst;
    const unsigned char* src;
    size_t bytes;
};
// a kernel pulls the bytes out of the page-locked chunks: unlike hipMemcpyAsync (which was seen to block the host for
// several milliseconds on a busy stream once a copy exceeds a few hundred KB) a launch never waits
__device__ __forceinline__ void copy_segment(const StageSeg g, size_t t, size_t stride) {
    if ((((uintptr_t)g.dst | (uintptr_t)g.src) & 15) == 0) {
        const size_t n16 = g.bytes / 16;
        for (size_t q = t; q < n16; q += stride) ((uint4*)g.dst)[q] = ((const uint4*)g.src)[q];
        for (size_t q = n16 * 16 + t; q < g.bytes; q += stride) g.dst[q] = g.src[q];
    } else if ((((uintptr_t)g.dst | (uintptr_t)g.src | g.bytes) & 3) == 0) {
        for (size_t q = t; q < g.bytes / 4; q += stride) ((uint32_t*)g.dst)[q] = ((const uint32_t*)g.src)[q];
    } else {
        for (size_t q = t; q < g.bytes; q += stride) g.dst[q] = g.src[q];
    }
}
__global__ __launch_bounds__(256) void k_stage_copy(const StageSeg* __restrict__ table, int per_seg) {
    copy_segment(table[blockIdx.x / per_seg], (size_t)(blockIdx.x % per_seg) * blockDim.x + threadIdx.x, (size_t)per_seg * blockDim.x);
}

// Everything of a reset that only depends on the staged bytes, in ONE launch (each dependent launch of the reset chain
// costs the stream ~10 us, whatever its size).  Workgroups, in order:
//   n_seg * per_seg : the segment copies (trajectories, RVO polygons, per-world tables, the world list)
//   n_worlds * MAP_BLOCKS : obs_map_ of each world being reset starts from the static map again (img_env.cpp:166-168)
//   robots, pedestrians : init_pose / set_goal / setPedPos ..., straight from the page-locked host blocks
#define MAP_BLOCKS 8
struct ResetArgs {
    const StageSeg* table;
    int n_seg, per_seg;
    const int* list;  // the worlds of this reset (page-locked host copy), nullptr = every world
    int n_worlds;
    const uint8_t* static_map;
    const double* rob3;
    const ResetRobot* rr;
    const double* ped3;
    int n_robots, n_peds, whole;
    int stamp;  // STAMP mode: the class layer gets its base classes here too (k_reset_obstacles keeps it in step)
};
static int stage_begin(imgenv* h) {
    h->gen = (h->gen + 1) % imgenv::STAGE_GENS;
    const int g = h->gen;
    if (!h->ev_gen[g]) HIPCHK(hipEventCreateWithFlags(&h->ev_gen[g], hipEventDisableTiming));
    if (h->gen_pending[g]) {  // the copies of the reset STAGE_GENS calls ago (long finished in practice) still own these chunks
        HIPCHK(hipEventSynchronize(h->ev_gen[g]));
        h->gen_pending[g] = false;
    }
    for (auto& c : h->stage_gen[g]) c.used = 0;
    h->segs.clear();
    h->seg_max = 0;
    return 0;
}
static int stage_room(imgenv* h, size_t bytes, unsigned char** out) {
    imgenv::Chunk* use = nullptr;
    std::vector<imgenv::Chunk>& chunks = h->stage_gen[h->gen];
    for (auto& c : chunks)
        if (c.cap - c.used >= bytes) {
            use = &c;
            break;
        }
    if (!use) {
        imgenv::Chunk c{nullptr, std::max(bytes, (size_t)1 << 20), 0};
        HIPCHK(hipHostMalloc((void**)&c.p, c.cap, hipHostMallocDefault));
        chunks.push_back(c);
        use = &chunks.back();
    }
    *out = use->p + use->used;
    use->used += (bytes + 255) & ~(size_t)255;
    if (use->used > use->cap) use->used = use->cap;
    return 0;
}
static int stage_put(imgenv* h, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return 0;
    unsigned char* p = nullptr;
    if (int rc = stage_room(h, bytes, &p)) return rc;
    memcpy(p, src, bytes);
    h->segs.push_back(imgenv::StageSegHost{dst, p, bytes});
    h->seg_max = std::max(h->seg_max, bytes);
    return 0;
}
// launch the copies queued so far
static int stage_flush(imgenv* h, hipStream_t st) {
    if (h->segs.empty()) return 0;
    static_assert(sizeof(imgenv::StageSegHost) == sizeof(StageSeg), "layout");
    unsigned char* table = nullptr;
    if (int rc = stage_room(h, h->segs.size() * sizeof(StageSeg), &table)) return rc;
    memcpy(table, h->segs.data(), h->segs.size() * sizeof(StageSeg));
    const int per_seg = (int)std::min<size_t>((h->seg_max / 16 + 255) / 256 + 1, 16);
    k_stage_copy<<<dim3((unsigned)(h->segs.size() * per_seg)), dim3(256), 0, st>>>((const StageSeg*)table, per_seg);
    HIPCHK(hipGetLastError());
    h->segs.clear();
    h->seg_max = 0;
    return 0;
}
static int stage_end(imgenv* h, hipStream_t st) {
    HIPCHK(hipEventRecord(h->ev_gen[h->gen], st));
    h->gen_pending[h->gen] = true;
    return 0;
}

// (errors inside reset leave the handle alive)
#define RTRY(expr)                   \
    do {                             \
        if (int rc_ = (expr)) return rc_; \
    } while (0)

// a launch covers every world (list == nullptr) or the robots, pedestrians and cells of the n listed worlds
static void set_active(imgenv* h, const int* list, int n) {
    DevWorld& d = h->d;
    d.act_cells = h->W > 1 ? h->Gs * h->W : h->Gs;
    if (!list) {
        d.act_list = nullptr; d.act_nw = h->W; d.act_nl = h->RL; d.act_ng = h->R; d.act_np = h->P;
    } else {
        d.act_list = list; d.act_nw = n; d.act_nl = d.act_ng = n * h->Rw; d.act_np = n * h->Pw;
    }
}

__global__ void k_restride3(double* __restrict__ dst, const double* __restrict__ src, int n, int old_cap, int new_cap) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * old_cap * 3) return;
    const size_t j = t / ((size_t)old_cap * 3), rem = t - j * (size_t)old_cap * 3;
    dst[j * (size_t)new_cap * 3 + rem] = src[t];
}

__global__ __launch_bounds__(256) void k_reset_apply(DevWorld w, ResetArgs a) {
    int b = blockIdx.x;
    if (b < a.n_seg * a.per_seg) {
        copy_segment(a.table[b / a.per_seg], (size_t)(b % a.per_seg) * blockDim.x + threadIdx.x, (size_t)a.per_seg * blockDim.x);
        return;
    }
    b -= a.n_seg * a.per_seg;
    if (b < a.n_worlds * MAP_BLOCKS) {
        const int q = b / MAP_BLOCKS, world = a.list ? a.list[q] : q;
        const size_t n16 = ((size_t)w.Hg * w.Wg + 15) / 16;  // (both buffers are padded to 16 bytes)
        uint4* dst = (uint4*)(const_cast<uint8_t*>(w.obs_map) + (size_t)world * w.Gs);
        uint4* cls = (uint4*)(w.cell + (size_t)world * w.Gs);
        for (size_t e = (size_t)(b - q * MAP_BLOCKS) * blockDim.x + threadIdx.x; e < n16; e += (size_t)MAP_BLOCKS * blockDim.x) {
            const uint4 v = ((const uint4*)a.static_map)[e];
            dst[e] = v;
            if (a.stamp) {  // base class of 16 cells, no stamp (2, SUM mode: the counts on the cells stay -- their owners take them off)
                const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t c4[4];
                    const uint4 old = a.stamp == 2 ? cls[4 * e + k] : make_uint4(0, 0, 0, 0);
                    const uint32_t keep[4] = {old.x & ~7u, old.y & ~7u, old.z & ~7u, old.w & ~7u};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t o = (wd[k] >> (8 * j)) & 0xFFu;
                        c4[j] = (o <= 2 ? o : (o < 250 ? CLS_LOW : CLS_HIGH)) | keep[j];
                    }
                    cls[4 * e + k] = make_uint4(c4[0], c4[1], c4[2], c4[3]);
                }
            }
        }
        if (w.crop_map) {  // view_big.h's one-byte-per-cell summary of the map starts over with it
            const uint4* src = (const uint4*)w.static_crop;
            uint4* cm = (uint4*)(w.crop_map + (size_t)world * w.crop_ws);
            for (size_t e = (size_t)(b - q * MAP_BLOCKS) * blockDim.x + threadIdx.x; e < w.crop_ws / 16; e += (size_t)MAP_BLOCKS * blockDim.x) cm[e] = src[e];
        }
        return;
    }
    b -= a.n_worlds * MAP_BLOCKS;
    const int robot_blocks = (a.n_robots + 255) / 256;
    if (b < robot_blocks) {
        const int t = b * 256 + threadIdx.x;
        if (t < a.n_robots) reset_robot(w, a.list, t, a.rob3, a.rr, a.whole);
        return;
    }
    const int t = (b - robot_blocks) * 256 + threadIdx.x;
    if (t < a.n_peds) reset_ped(w, a.list, t, a.ped3);
}
// Every obstacle is then drawn into its world's map with value 0: Agent::draw(obs_map, 0, "world_map") (img_env.cpp:169-193,
// agent.cpp:285-327) writes unless the cell holds 0 / 1 / 2 -- and only ever writes 0, so the order does not matter.
// One workgroup per obstacle.  Its footprint samples (agent.cpp:18-30, 51-62: a 0.01 m lattice, for circles the points within
// the radius) are generated on the fly from the lattice bounds the host computed: random obstacle radii (EnvPos draws a fresh
// one per episode, reset_helper.py:131-133) then cost nothing -- no per-size sample list to upload, cache or free.
struct ObstInst {
    double x, y, sh, ch;  // pose; sin / cos of yaw/2, evaluated on the host
    double cx, cy, r;     // circle: centre offset and radius (sizes[0..2]); rectangle: unused
    int m0, m1, n0, n1;   // lattice range: circle -bb..bb with bb = ceil(r / 0.01); rectangle floor(min / 0.01)..ceil(max / 0.01)
    int shape, world;
};
template <bool POW2>
__global__ __launch_bounds__(256) void k_reset_obstacles(DevWorld w, const ObstInst* __restrict__ inst, int stamp, const int* n_dev, int per_world, int parts,
                                                        ObstInst* keep, int* keep_valid) {
    // parts workgroups share an obstacle's samples (a reset of a few worlds is a handful of obstacles: their latency is the launch's)
    const int t0 = (int)blockIdx.x / parts, part = (int)blockIdx.x - t0 * parts;
    // device-side auto-reset: per_world instances for each finished world, the grid sized for a guess of their number
    for (int t = t0; t < (n_dev ? *n_dev * per_world : t0 + 1); t += (int)gridDim.x / parts) {
        const ObstInst o = inst[t];
        if (o.world < 0) continue;  // (a finished world whose placement failed: k_respawn)
        if (keep && part == 0 && threadIdx.x == 0) {  // ... which remembers what each world now carries, for the next restore (k_restore_maps_dev)
            keep[(size_t)o.world * per_world + (t % per_world)] = o;
            keep_valid[o.world] = 1;
        }
        const Tf2 bw = tf_from_pose_sc(o.x, o.y, o.sh, o.ch);
        uint8_t* map = const_cast<uint8_t*>(w.obs_map) + (size_t)o.world * w.Gs;
        const double resolution = 0.01;
        const int nn = o.n1 - o.n0 + 1, total = (o.m1 - o.m0 + 1) * nn;
        const bool circle = o.shape == IMGENV_SHAPE_CIRCLE;
        for (int q = part * (int)blockDim.x + (int)threadIdx.x; q < total; q += parts * (int)blockDim.x) {
            const int m = o.m0 + q / nn, n = o.n0 + q % nn;
            double px = m * resolution, py = n * resolution;
            if (circle) {
                if (!(sqrt(m * resolution * m * resolution + n * resolution * n * resolution) <= o.r)) continue;
                px = px + o.cx;
                py = py + o.cy;
            }
            double wx, wy;
            tf_apply(bw, px, py, wx, wy);
            int gm, gn;
            w2m_pair<POW2>(wx, wy, w.res, w.inv_res, gm, gn);
            if (gm >= 0 && gm < w.Hg && gn >= 0 && gn < w.Wg) {
                const size_t at = (size_t)gm * w.Wg + gn;
                if (map[at] > 2) {
                    map[at] = 0;
                    if (stamp == 1) w.cell[(size_t)o.world * w.Gs + at] = CLS_STATIC;  // the class layer's base class follows
                    else if (stamp == 2) w.cell[(size_t)o.world * w.Gs + at] &= ~7u;        // (SUM mode: the counts stay; every writer of this cell writes the same word)
                    if (w.crop_map) w.crop_map[(size_t)o.world * w.crop_ws + crop_tiled(w, (uint32_t)gm, (uint32_t)gn)] = 0;  // ... and view_big.h's byte: not free, no stamp
                }
            }
        }
    }
}

#include "spawn_device.h"  // device-side auto-reset: k_spawn_fill, k_finished_dev, k_respawn, k_restore_maps_dev

// the obstacle list of one world's reset batch: instances for k_reset_obstacles, the RVO polygons with their BSP
// (RVOScene::addObs + processObs, rvoscene.h:19-26, img_env.cpp:283) and the social-force segments (pedscene.h:22-26)
static int world_obstacles(imgenv* h, int k, const imgenv_reset_batch* b, std::vector<double>& sfm_obs) {
    RvoObstacles& rvo = h->rvos[k];
    rvo.clear();
    for (int q = 0; q < b->n_obstacles; q++) {
        const int shape = b->obs_shape[q];
        double sizes[4];
        for (int j = 0; j < 4; j++) sizes[j] = (double)b->obs_size[4 * q + j];
        const double* p = b->obs_pose + 4 * q;
        const double yaw = tf_yaw_from_quaternion_zw(p[2], p[3]);
        if (shape != IMGENV_SHAPE_CIRCLE && shape != IMGENV_SHAPE_RECTANGLE) FAIL(IMGENV_EINVAL, "obstacle %d: unsupported shape %d", q, shape);
        imgenv::ObstInstHost oi;
        oi.x = p[0]; oi.y = p[1]; oi.sh = sin(yaw * 0.5); oi.ch = cos(yaw * 0.5);
        oi.world = k;
        oi.shape = shape;
        oi.cx = sizes[0]; oi.cy = sizes[1]; oi.r = sizes[2];
        if (shape == IMGENV_SHAPE_CIRCLE) {  // init_shape_circle (agent.cpp:18-30)
            const int bb = (int)ceil(sizes[2] / 0.01);
            oi.m0 = oi.n0 = -bb;
            oi.m1 = oi.n1 = bb;
        } else {                             // init_shape_rectangle (agent.cpp:51-62)
            oi.m0 = (int)floor(sizes[0] / 0.01); oi.m1 = (int)ceil(sizes[1] / 0.01);
            oi.n0 = (int)floor(sizes[2] / 0.01); oi.n1 = (int)ceil(sizes[3] / 0.01);
        }
        if ((long long)(oi.m1 - oi.m0 + 1) * (long long)(oi.n1 - oi.n0 + 1) > (1ll << 26) || oi.m1 < oi.m0 || oi.n1 < oi.n0)
            FAIL(IMGENV_EINVAL, "obstacle %d: degenerate or oversized footprint", q);
        h->oinst.push_back(oi);
        const Tf2 bw = tf_from_pose_sc(p[0], p[1], oi.sh, oi.ch);
        double pax, pay, pbx, pby;
        get_corners(shape, sizes, bw, pax, pay, pbx, pby);
        if (!b->ignore_obstacle && h->cfg.ped_scene_type == IMGENV_SCENE_PEDSIM) {  // PedScene::addObs: the segment pa -> pb
            sfm_obs.push_back(pax); sfm_obs.push_back(pay); sfm_obs.push_back(pbx); sfm_obs.push_back(pby);
        }
        if (!b->ignore_obstacle && h->NA > 0) {  // RVOScene::addObs
            const float v[8] = {(float)pax, (float)pay, (float)pax, (float)pby, (float)pbx, (float)pby, (float)pbx, (float)pay};
            rvo.add(v, 4);
        }
    }
    rvo.process();
    return 0;
}

// obstacle segments and their BSP of world k: a slice of cap_obst / cap_nodes entries per world
static int put_world_rvo(imgenv* h, int k) {
    static_assert(sizeof(RvoObstHost) == sizeof(RvoObstDev) && sizeof(RvoNodeHost) == sizeof(RvoNodeDev), "layout");
    const RvoObstacles& r = h->rvos[k];
    bool all = false;
    if (h->dev_reset_used && ((int)r.ob.size() > h->cap_obst || (int)r.nodes.size() > h->cap_nodes))
        FAIL(IMGENV_ESTATE, "world %d: more RVO obstacle vertices than the handle has room for, after imgenv_step_autoreset_device has taken over "
                            "the per-world obstacle tables (reset every world with imgenv_reset first)", k);
    if ((int)r.ob.size() > h->cap_obst) {
        h->cap_obst = (int)r.ob.size() * 2;
        if (int rc = dev_alloc(h, &h->d_obst, (size_t)h->cap_obst * h->W)) return rc;
        all = true;
    }
    if ((int)r.nodes.size() > h->cap_nodes) {
        h->cap_nodes = (int)r.nodes.size() * 2;
        if (int rc = dev_alloc(h, &h->d_nodes, (size_t)h->cap_nodes * h->W)) return rc;
        all = true;
    }
    if (all) h->wobst_all = true;  // every world's slice moved: the whole per-world table goes up with this reset
    for (int q = all ? 0 : k; q < (all ? h->W : k + 1); q++) {  // a grown array is refilled from the host copies
        const RvoObstacles& rq = h->rvos[q];
        if (!rq.ob.empty()) RTRY(stage_put(h, h->d_obst + (size_t)q * h->cap_obst, rq.ob.data(), sizeof(RvoObstDev) * rq.ob.size()));
        if (!rq.nodes.empty()) RTRY(stage_put(h, h->d_nodes + (size_t)q * h->cap_nodes, rq.nodes.data(), sizeof(RvoNodeDev) * rq.nodes.size()));
        h->wobst[q] = q * h->cap_obst;
        h->wobst[h->W + q] = q * h->cap_nodes;
        h->wobst[2 * h->W + q] = (int)rq.ob.size();
        h->wobst[3 * h->W + q] = rq.root;
    }
    DevWorld& d = h->d;
    d.obst = h->d_obst;
    d.onodes = h->d_nodes;
    d.n_obst = (int)h->rvos[0].ob.size();  // (W > 1: the kernels read the per-world table instead)
    d.n_onodes = (int)h->rvos[0].nodes.size();
    d.oroot = h->rvos[0].root;
    return 0;
}

// Host half of one world's reset: its obstacles, pedestrians and robots go into the staging chunks.
static int stage_world(imgenv* h, int k, int q_list, const imgenv_reset_batch* b, hipStream_t st) {
    DevWorld& d = h->d;
    const int W = h->W, Rw = h->Rw, Pw = h->Pw, P = h->P;
    const int g_lo = k * Rw, p_lo = k * Pw;  // first robot / pedestrian of the world
    std::vector<double> sfm_obs;
    RTRY(world_obstacles(h, k, b, sfm_obs));
    RTRY(put_world_rvo(h, k));
    if (h->cfg.ped_scene_type == IMGENV_SCENE_PEDSIM) {
        const int nob = (int)sfm_obs.size() / 4;
        if (W > 1) {  // world k's slice and count
            if (nob > h->sfm_cap_obs) FAIL(IMGENV_EINVAL, "world %d: more than %d obstacles in a pedscene world of a multi-world handle", k, h->sfm_cap_obs);
            if (nob) RTRY(stage_put(h, d.sfm.obs + (size_t)k * d.sfm.cap_obs * 4, sfm_obs.data(), sizeof(double) * sfm_obs.size()));
            h->sfm_nobs_w[k] = nob;
            RTRY(stage_put(h, const_cast<int*>(d.sfm.n_obs_w) + k, &h->sfm_nobs_w[k], sizeof(int)));
        } else {
            if (nob > h->sfm_cap_obs) {
                h->sfm_cap_obs = nob * 2;
                if (int rc = dev_alloc(h, &d.sfm.obs, (size_t)h->sfm_cap_obs * 4)) return rc;
                d.sfm.cap_obs = h->sfm_cap_obs;
            }
            if (nob) RTRY(stage_put(h, d.sfm.obs, sfm_obs.data(), sizeof(double) * sfm_obs.size()));
            d.sfm.n_obs = nob;
        }
    }
    // pedestrians (img_env.cpp:220-250)
    if (Pw > 0) {
        const bool dataset = h->cfg.ped_scene_type == IMGENV_SCENE_DATASET;
        if (dataset && !b->ped_traj_v) FAIL(IMGENV_EINVAL, "dataset scene: ped_traj_v is missing from the reset batch");
        if (b->ped_traj_cap > h->traj_cap) {  // longer trajectories than any before: re-lay the table out (other worlds keep theirs)
            RTRY(stage_flush(h, st));  // (queued copies aim at the old table)
            const int old_cap = h->traj_cap;
            double *old_t = h->d_traj, *old_v = h->d_traj_v;
            h->traj_cap = b->ped_traj_cap;
            if (int rc = dev_alloc(h, &h->d_traj, (size_t)P * h->traj_cap * 3)) return rc;
            if (dataset)
                if (int rc = dev_alloc(h, &h->d_traj_v, (size_t)P * h->traj_cap * 3)) return rc;
            if (W > 1 && old_cap > 0) {
                const size_t n = (size_t)P * old_cap * 3;
                k_restride3<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(h->d_traj, old_t, P, old_cap, h->traj_cap);
                if (dataset) k_restride3<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(h->d_traj_v, old_v, P, old_cap, h->traj_cap);
            }
        }
        double* ped3 = h->pin_ped3 + (size_t)q_list * Pw * 3;
        std::vector<int>& tlen = h->tmp_i0;
        std::vector<double>& traj = h->tmp_d1;
        tlen.resize(Pw);
        traj.assign((size_t)Pw * h->traj_cap * 3, 0.0);
        for (int j = 0; j < Pw; j++) {
            const double* p = b->ped_pose + 4 * j;
            ped3[3 * j] = p[0];
            ped3[3 * j + 1] = p[1];
            ped3[3 * j + 2] = tf_yaw_from_quaternion_zw(p[2], p[3]);
            tlen[j] = b->ped_traj_len[j];
            if (tlen[j] < 1 || tlen[j] > b->ped_traj_cap) FAIL(IMGENV_EINVAL, "ped %d: bad trajectory length %d", p_lo + j, tlen[j]);
            for (int q = 0; q < tlen[j]; q++)
                memcpy(&traj[((size_t)j * h->traj_cap + q) * 3], b->ped_traj + ((size_t)j * b->ped_traj_cap + q) * 3, 24);
        }
        RTRY(stage_put(h, h->d_traj + (size_t)p_lo * h->traj_cap * 3, traj.data(), traj.size() * 8));
        RTRY(stage_put(h, h->d_traj_len + p_lo, tlen.data(), sizeof(int) * Pw));
        if (dataset) {  // trajectory_v (img_env.cpp:246-247) + the yaw _step_ped_dataset derives from it, with the host's libm
            std::vector<double>& tv = h->tmp_d1;
            tv.assign((size_t)Pw * h->traj_cap * 3, 0.0);
            for (int j = 0; j < Pw; j++)
                for (int q = 0; q < tlen[j]; q++) {
                    const double* v = b->ped_traj_v + ((size_t)j * b->ped_traj_cap + q) * 2;
                    double* o = &tv[((size_t)j * h->traj_cap + q) * 3];
                    o[0] = v[0];
                    o[1] = v[1];
                    o[2] = atan2(v[1], v[0]);
                }
            RTRY(stage_put(h, h->d_traj_v + (size_t)p_lo * h->traj_cap * 3, tv.data(), tv.size() * 8));
            d.ptraj_v = h->d_traj_v;
        }
        if (h->cfg.ped_scene_type == IMGENV_SCENE_PEDSIM) {  // PedScene::setWayPoint (pedscene.h:38-46): [goal r=1, trajectory r=z]
            // (this world's pedestrians: the first Pw agents of its crowd)
            std::vector<double> wx((size_t)Pw * SFM_MAX_WP, 0.0), wy(wx), wr(wx);
            std::vector<int> dq((size_t)Pw * SFM_MAX_WP, 0), dqn(Pw, 0), dest(Pw, 0), last(Pw, -1);
            for (int j = 0; j < Pw; j++) {
                int nw = 0;
                wx[(size_t)j * SFM_MAX_WP] = b->ped_goal[2 * j];
                wy[(size_t)j * SFM_MAX_WP] = b->ped_goal[2 * j + 1];
                wr[(size_t)j * SFM_MAX_WP] = 1.0;
                nw = 1;
                for (int q = 0; q < tlen[j] && nw < SFM_MAX_WP; q++, nw++) {
                    const double* tp = b->ped_traj + ((size_t)j * b->ped_traj_cap + q) * 3;
                    wx[(size_t)j * SFM_MAX_WP + nw] = tp[0];
                    wy[(size_t)j * SFM_MAX_WP + nw] = tp[1];
                    wr[(size_t)j * SFM_MAX_WP + nw] = tp[2];
                }
                for (int q = 0; q < nw; q++) dq[(size_t)j * SFM_MAX_WP + q] = q;
                dqn[j] = nw;
                dest[j] = 0;  // addWaypoint leaves destination = waypoints.front() without popping it (ped_agent.cpp:97-100)
            }
            const SfmDev& f = d.sfm;
            const size_t a0 = (size_t)k * f.n;  // first agent of world k's crowd
            RTRY(stage_put(h, f.wpx + a0 * SFM_MAX_WP, wx.data(), wx.size() * 8));
            RTRY(stage_put(h, f.wpy + a0 * SFM_MAX_WP, wy.data(), wy.size() * 8));
            RTRY(stage_put(h, f.wpr + a0 * SFM_MAX_WP, wr.data(), wr.size() * 8));
            RTRY(stage_put(h, f.dq + a0 * SFM_MAX_WP, dq.data(), dq.size() * 4));
            RTRY(stage_put(h, f.dq_n + a0, dqn.data(), dqn.size() * 4));
            RTRY(stage_put(h, f.dest + a0, dest.data(), dest.size() * 4));
            RTRY(stage_put(h, f.last + a0, last.data(), last.size() * 4));
        }
        d.ptraj = h->d_traj;
        d.traj_cap = h->traj_cap;
    }
    // robots (img_env.cpp:252-282)
    const int l_lo = W > 1 ? g_lo : 0, n_l = W > 1 ? Rw : h->RL;  // local robots of this world (a shard exists for W == 1 only)
    double* rob3 = h->pin_rob3 + (size_t)q_list * Rw * 5;
    ResetRobot* rr = (ResetRobot*)h->pin_rr + (size_t)q_list * n_l;
    for (int i = 0; i < Rw; i++) {
        const double* p = b->robot_pose + 4 * i;
        const double yaw = tf_yaw_from_quaternion_zw(p[2], p[3]);
        rob3[5 * i] = p[0];
        rob3[5 * i + 1] = p[1];
        rob3[5 * i + 2] = yaw;
        rob3[5 * i + 3] = sin(yaw * 0.5);
        rob3[5 * i + 4] = cos(yaw * 0.5);
        const int l = g_lo + i - h->r0;
        if (l >= 0 && l < h->RL) {  // set_goal (agent.cpp:144-154)
            ResetRobot& q = rr[l - l_lo];
            q.gx = b->robot_goal[2 * i];
            q.gy = b->robot_goal[2 * i + 1];
            q.world_target = tf_inverse(tf_from_pose(q.gx, q.gy, yaw));
        }
    }
    return 0;
}

// everything that can be wrong with the batches is found here, BEFORE any of them touches the handle's state: a reset
// that fails part-way would leave host and device copies of the earlier worlds out of step
static int reset_checks(imgenv* h, int n, const imgenv_reset_batch* b, int peds_per_batch) {
    if (!h || !b) FAIL(IMGENV_EINVAL, "null argument");
    for (int q = 0; q < n; q++) {
        if (b[q].struct_size != (int32_t)sizeof(imgenv_reset_batch)) FAIL(IMGENV_EINVAL, "reset batch ABI mismatch");
        if (b[q].n_obstacles < 0 || (h->P > 0 && b[q].ped_traj_cap < 1)) FAIL(IMGENV_EINVAL, "bad reset batch");
        if (b[q].n_obstacles > 0 && (!b[q].obs_shape || !b[q].obs_size || !b[q].obs_pose)) FAIL(IMGENV_EINVAL, "reset batch %d: null obstacle arrays", q);
        if (!b[q].robot_pose || !b[q].robot_goal) FAIL(IMGENV_EINVAL, "reset batch %d: null robot arrays", q);
        for (int o = 0; o < b[q].n_obstacles; o++)
            if (b[q].obs_shape[o] != IMGENV_SHAPE_CIRCLE && b[q].obs_shape[o] != IMGENV_SHAPE_RECTANGLE)
                FAIL(IMGENV_EINVAL, "reset batch %d, obstacle %d: unsupported shape %d", q, o, b[q].obs_shape[o]);
        if (peds_per_batch > 0) {
            if (!b[q].ped_pose || !b[q].ped_traj_len || !b[q].ped_traj) FAIL(IMGENV_EINVAL, "reset batch %d: null pedestrian arrays", q);
            if (h->cfg.ped_scene_type == IMGENV_SCENE_DATASET && !b[q].ped_traj_v)
                FAIL(IMGENV_EINVAL, "dataset scene: ped_traj_v is missing from the reset batch");
            if (h->cfg.ped_scene_type == IMGENV_SCENE_PEDSIM && !b[q].ped_goal) FAIL(IMGENV_EINVAL, "pedscene: ped_goal is missing from the reset batch");
            for (int j = 0; j < peds_per_batch; j++)
                if (b[q].ped_traj_len[j] < 1 || b[q].ped_traj_len[j] > b[q].ped_traj_cap)
                    FAIL(IMGENV_EINVAL, "reset batch %d, ped %d: bad trajectory length %d", q, j, b[q].ped_traj_len[j]);
        }
    }
    if (h->obs_forked) {
        // a step that was begun and never ended (the caller's exchange failed in between): its k_obs is in flight on a side
        // stream and will still arrive at the tails' hand-over words; wait for it, the reset's first chain then clears them
        if (h->side2) HIPCHK(hipStreamSynchronize(h->side2));
        h->obs_forked = false;
    }
    HIPCHK(hipSetDevice(h->cfg.device));
    if (int rc = check_device_flags(h)) {  // report what the abandoned episode raised, then start clean
        for (int q = 0; q < 8; q++) h->err_host[q] = 0;
        return rc;
    }
    if (!h->d_act_list) RTRY(dev_alloc(h, &h->d_act_list, (size_t)h->W));
    return 0;
}

// Device half of a reset, for every world (list == nullptr) or the n worlds listed: one upload launch, the obstacle
// maps, the robot / pedestrian state, then view_agent + get_states (img_env.cpp:285-286) for those worlds' robots.
// page-locked blocks for the robots / pedestrians of the n worlds of this reset, in list order (read by k_reset_apply)
static int reset_blocks(imgenv* h, int n, const int* list) {
    unsigned char* p = nullptr;
    const size_t n_local = h->W > 1 ? (size_t)n * h->Rw : (size_t)h->RL;
    RTRY(stage_room(h, sizeof(double) * 5 * n * h->Rw, &p));
    h->pin_rob3 = (double*)p;
    RTRY(stage_room(h, sizeof(ResetRobot) * std::max<size_t>(n_local, 1), &p));
    h->pin_rr = p;
    RTRY(stage_room(h, sizeof(double) * 3 * std::max<size_t>((size_t)n * h->Pw, 1), &p));
    h->pin_ped3 = (double*)p;
    h->pin_list = nullptr;
    if (list) {
        RTRY(stage_room(h, sizeof(int) * n, &p));
        h->pin_list = (int*)p;
        memcpy(h->pin_list, list, sizeof(int) * n);
    }
    return 0;
}

static int sfm_ahead_drop(imgenv* h, hipStream_t st);
static int reset_launch(imgenv* h, const int* list, int n, hipStream_t st, int whole) {
    DevWorld& d = h->d;
    if (int rc = sfm_ahead_drop(h, st)) return rc;  // (a reset writes the live crowd: positions, waypoints)
    if (h->sd_ready) {  // a host-side reset draws obstacles the device-side restore does not know: whole-map restore next time
        static const std::vector<int> zeros(1 << 16, 0);
        SpawnDev& c = *(SpawnDev*)h->sd_storage;
        if (!list) {
            for (int k0 = 0; k0 < h->W; k0 += (int)zeros.size())
                RTRY(stage_put(h, c.w_inst_valid + k0, zeros.data(), sizeof(int) * (size_t)std::min<int>((int)zeros.size(), h->W - k0)));
        } else {
            for (int q = 0; q < n; q++) RTRY(stage_put(h, c.w_inst_valid + list[q], zeros.data(), sizeof(int)));
        }
    }
    if (!list || h->wobst_all) RTRY(stage_put(h, h->d_wobst, h->wobst.data(), sizeof(int) * h->wobst.size()));
    h->wobst_all = false;
    if (!list) {
        RTRY(stage_put(h, h->d_world_epoch, h->world_epoch.data(), sizeof(int) * h->W));
    } else {  // only the listed worlds' entries: the device-side auto-reset keeps the others' up to date itself
        for (int q = 0; q < n; q++) {
            const int k = list[q];
            for (int t = 0; t < 4; t++) RTRY(stage_put(h, h->d_wobst + (size_t)t * h->W + k, &h->wobst[(size_t)t * h->W + k], sizeof(int)));
            RTRY(stage_put(h, h->d_world_epoch + k, &h->world_epoch[k], sizeof(int)));
        }
    }
    if (list) RTRY(stage_put(h, h->d_act_list, list, sizeof(int) * n));  // for the launches after k_reset_apply
    const size_t n_inst = h->oinst.size();
    if (n_inst > h->cap_oinst) {
        h->cap_oinst = n_inst * 2;
        unsigned char* raw = nullptr;
        RTRY(dev_alloc(h, &raw, h->cap_oinst * sizeof(ObstInst)));
        h->d_oinst = raw;
    }
    static_assert(sizeof(imgenv::ObstInstHost) == sizeof(ObstInst), "layout");
    if (n_inst) RTRY(stage_put(h, h->d_oinst, h->oinst.data(), n_inst * sizeof(ObstInst)));
    h->oinst.clear();
    set_active(h, list ? h->d_act_list : nullptr, n);
    {   // one launch: segment copies | map restore | robot state | pedestrian state
        ResetArgs a;
        unsigned char* table = nullptr;
        RTRY(stage_room(h, std::max<size_t>(h->segs.size(), 1) * sizeof(StageSeg), &table));
        memcpy(table, h->segs.data(), h->segs.size() * sizeof(StageSeg));
        a.table = (const StageSeg*)table;
        a.n_seg = (int)h->segs.size();
        a.per_seg = (int)std::min<size_t>((h->seg_max / 16 + 255) / 256 + 1, 16);
        a.list = h->pin_list;
        a.n_worlds = d.act_nw;
        a.static_map = h->d_static_map;
        a.rob3 = h->pin_rob3;
        a.rr = (const ResetRobot*)h->pin_rr;
        a.ped3 = h->pin_ped3;
        a.n_robots = d.act_ng;
        a.n_peds = d.act_np;
        a.whole = whole;
        a.stamp = h->stamp ? 1 : h->sum ? 2 : 0;
        const size_t blocks = (size_t)a.n_seg * a.per_seg + (size_t)a.n_worlds * MAP_BLOCKS + (a.n_robots + 255) / 256 + (a.n_peds + 255) / 256;
        k_reset_apply<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(d, a);
        h->segs.clear();
        h->seg_max = 0;
    }
    if (n_inst) {
        if (h->pow2) k_reset_obstacles<true><<<dim3((unsigned)n_inst), dim3(256), 0, st>>>(d, (const ObstInst*)h->d_oinst, h->stamp ? 1 : h->sum ? 2 : 0, nullptr, 0, 1, nullptr, nullptr);
        else k_reset_obstacles<false><<<dim3((unsigned)n_inst), dim3(256), 0, st>>>(d, (const ObstInst*)h->d_oinst, h->stamp ? 1 : h->sum ? 2 : 0, nullptr, 0, 1, nullptr, nullptr);
    }
    if (d.sharded) k_reset_bbox<<<dim3((h->RL + 255) / 256), dim3(256), 0, st>>>(d, h->pin_rob3);
    HIPCHK(hipGetLastError());
    h->launches = 2;
    const int rc = launch_views(h, st, 1);
    set_active(h, nullptr, 0);
    // k_reset_apply is in flight and reads this generation's page-locked chunks: mark them pending whatever came after
    const int rc_end = stage_end(h, st);
    return rc ? rc : rc_end;
}

extern "C" int imgenv_reset(imgenv_t* h, const imgenv_reset_batch* b, void* stream) {
    if (int rc = reset_checks(h, 1, b, h ? h->P : 0)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = outputs_verify(h, st)) return rc;
    RTRY(stage_begin(h));
    RTRY(reset_blocks(h, h->W, nullptr));
    h->oinst.clear();
    for (int k = 0; k < h->W; k++) {
        imgenv_reset_batch sub = *b;  // world k's robots and pedestrians (world-major numbering); one obstacle list for all
        sub.robot_pose = b->robot_pose + (size_t)4 * k * h->Rw;
        sub.robot_goal = b->robot_goal + (size_t)2 * k * h->Rw;
        if (h->P > 0) {
            sub.ped_pose = b->ped_pose + (size_t)4 * k * h->Pw;
            sub.ped_goal = b->ped_goal ? b->ped_goal + (size_t)2 * k * h->Pw : nullptr;
            sub.ped_traj_len = b->ped_traj_len + (size_t)k * h->Pw;
            sub.ped_traj = b->ped_traj + (size_t)k * h->Pw * b->ped_traj_cap * 3;
            sub.ped_traj_v = b->ped_traj_v ? b->ped_traj_v + (size_t)k * h->Pw * b->ped_traj_cap * 2 : nullptr;
        }
        RTRY(stage_world(h, k, k, &sub, st));
        h->world_epoch[k] = 0;
        h->world_ready[k] = 1;
    }
    if (h->d.sharded) {
        const uint32_t init[4] = {BBOX_INIT_MIN, BBOX_INIT_MIN, BBOX_INIT_MAX, BBOX_INIT_MAX};
        RTRY(stage_put(h, h->d.bbox, init, sizeof(init)));
    }
    h->elapsed = 0;  // TimeLimitWrapper.reset (base.py:229-231)
    h->dev_reset_used = false;  // every world's host copy is current again
    RTRY(reset_launch(h, nullptr, 0, st, 1));  // no host wait: the copies read the handle's pinned chunks
    h->has_reset = true;
    return IMGENV_OK;
}

extern "C" int imgenv_reset_worlds(imgenv_t* h, int32_t n, const int32_t* worlds, const imgenv_reset_batch* batches, void* stream) {
    if (n <= 0) return h ? IMGENV_OK : IMGENV_EINVAL;
    if (!worlds) FAIL(IMGENV_EINVAL, "null argument");
    if (int rc = reset_checks(h, n, batches, h ? h->Pw : 0)) return rc;
    if (h->W == 1) {
        if (n != 1 || worlds[0] != 0) FAIL(IMGENV_EINVAL, "world out of range (n_worlds 1)");
        return imgenv_reset(h, batches, stream);
    }
    std::vector<char> seen(h->W, 0);
    for (int q = 0; q < n; q++) {
        if (worlds[q] < 0 || worlds[q] >= h->W) FAIL(IMGENV_EINVAL, "world %d out of range (n_worlds %d)", worlds[q], h->W);
        if (seen[worlds[q]]) FAIL(IMGENV_EINVAL, "world %d listed twice", worlds[q]);
        seen[worlds[q]] = 1;
    }
    hipStream_t st = (hipStream_t)stream;
    if (int rc = outputs_verify(h, st)) return rc;
    static const bool trace = getenv("IMGENV_TRACE_RESET") != nullptr;
    std::chrono::steady_clock::time_point tp[4];
    if (trace) tp[0] = std::chrono::steady_clock::now();
    RTRY(stage_begin(h));
    RTRY(reset_blocks(h, n, worlds));
    h->oinst.clear();
    if (trace) tp[1] = std::chrono::steady_clock::now();
    for (int q = 0; q < n; q++) {
        RTRY(stage_world(h, worlds[q], q, batches + q, st));
        h->world_epoch[worlds[q]] = h->elapsed;  // its TimeLimitWrapper starts over
        h->world_ready[worlds[q]] = 1;
    }
    if (trace) tp[2] = std::chrono::steady_clock::now();
    RTRY(reset_launch(h, worlds, n, st, 0));
    if (trace) {
        tp[3] = std::chrono::steady_clock::now();
        auto us = [&](int a_, int b_) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(tp[b_] - tp[a_]).count() * 1e-3; };
        static double acc[3] = {0, 0, 0};
        static long calls = 0, worlds_n = 0;
        acc[0] += us(0, 1); acc[1] += us(1, 2); acc[2] += us(2, 3);
        calls++; worlds_n += n;
        if (calls % 200 == 0)
            fprintf(stderr, "[imgenv_reset_worlds] %ld calls, %.1f worlds/call: begin %.1f us, stage worlds %.1f us, launches %.1f us per call\n",
                    calls, (double)worlds_n / calls, acc[0] / calls, acc[1] / calls, acc[2] / calls);
    }
    bool all = true;
    for (char r : h->world_ready) all = all && r;
    h->has_reset = all;
    return IMGENV_OK;
}

static int spawn_cfg_check(const imgenv_spawn_cfg* c) {
    if (!c) FAIL(IMGENV_EINVAL, "null argument");
    if (c->struct_size != (int32_t)sizeof(imgenv_spawn_cfg)) FAIL(IMGENV_EINVAL, "spawn cfg ABI mismatch");
    if (c->n_robots < 0 || c->n_peds < 0 || c->n_obstacles < 0 || (c->n_robots + c->n_peds > 0 && !c->agents) ||
        (c->n_obstacles > 0 && !c->obstacles) || !(c->clearance >= 0))
        FAIL(IMGENV_EINVAL, "bad spawn cfg");
    for (int i = 0; i < c->n_robots + c->n_peds; i++) {
        const imgenv_spawn_agent& a = c->agents[i];
        const bool begin_ok = (a.begin_type >= IMGENV_POSE_FIX && a.begin_type <= IMGENV_POSE_RANGE_YAW) ||
                              a.begin_type == IMGENV_POSE_RANGE_CIRCLE || a.begin_type == IMGENV_POSE_RANGE_CIRCLE_FIX ||
                              a.begin_type == IMGENV_POSE_RANGE_MULTI;
        const bool target_ok = a.target_type >= IMGENV_POSE_FIX && a.target_type <= IMGENV_POSE_RANGE_MULTI;
        if (!begin_ok || !target_ok)
            FAIL(IMGENV_EINVAL, "agent %d: unsupported pose type (%d, %d)", i, a.begin_type, a.target_type);
    }
    return 0;
}

extern "C" int imgenv_spawn(const imgenv_spawn_cfg* cfg, uint64_t seed, double* robot_pose, double* robot_goal, double* ped_pose,
                            double* ped_goal, double* ped_traj, int32_t* ped_traj_len, int32_t* obs_shape, float* obs_size,
                            double* obs_pose) {
    if (int rc = spawn_cfg_check(cfg)) return rc;
    SpawnOut o;
    if (const char* why = spawn_world(*cfg, seed, o)) FAIL(IMGENV_EINVAL, "spawn: %s", why);
    const size_t R = cfg->n_robots, P = cfg->n_peds, O = cfg->n_obstacles;
    if (robot_pose) memcpy(robot_pose, o.robot_pose.data(), sizeof(double) * 4 * R);
    if (robot_goal) memcpy(robot_goal, o.robot_goal.data(), sizeof(double) * 2 * R);
    if (ped_pose) memcpy(ped_pose, o.ped_pose.data(), sizeof(double) * 4 * P);
    if (ped_goal) memcpy(ped_goal, o.ped_goal.data(), sizeof(double) * 2 * P);
    if (ped_traj) memcpy(ped_traj, o.ped_traj.data(), sizeof(double) * 6 * P);
    if (ped_traj_len) memcpy(ped_traj_len, o.ped_traj_len.data(), sizeof(int32_t) * P);
    if (obs_shape) memcpy(obs_shape, o.obs_shape.data(), sizeof(int32_t) * O);
    if (obs_size) memcpy(obs_size, o.obs_size.data(), sizeof(float) * 4 * O);
    if (obs_pose) memcpy(obs_pose, o.obs_pose.data(), sizeof(double) * 4 * O);
    return IMGENV_OK;
}

extern "C" int imgenv_reset_worlds_spawn(imgenv_t* h, int32_t n, const int32_t* worlds, const imgenv_spawn_cfg* cfg,
                                         const uint64_t* seeds, void* stream) {
    if (!h || !worlds || !seeds) FAIL(IMGENV_EINVAL, "null argument");
    if (int rc = spawn_cfg_check(cfg)) return rc;
    if (cfg->n_robots != h->Rw || cfg->n_peds != h->Pw)
        FAIL(IMGENV_EINVAL, "spawn cfg is for %d robots / %d pedestrians, a world of this handle has %d / %d", cfg->n_robots,
             cfg->n_peds, h->Rw, h->Pw);
    if (n <= 0) return IMGENV_OK;
    std::vector<SpawnOut> outs((size_t)n);
    std::vector<imgenv_reset_batch> batches((size_t)n);
    for (int q = 0; q < n; q++) {
        if (const char* why = spawn_world(*cfg, seeds[q], outs[q])) FAIL(IMGENV_EINVAL, "spawn of world %d: %s", worlds[q], why);
        batches[q] = outs[q].batch;
    }
    return imgenv_reset_worlds(h, n, worlds, batches.data(), stream);
}

extern "C" int imgenv_reset_world(imgenv_t* h, int32_t world, const imgenv_reset_batch* b, void* stream) {
    return imgenv_reset_worlds(h, 1, &world, b, stream);
}

// ---------------------------------------------------------------------------------------- step
// k_sfm for one step of every crowd, on stream s: from the live set into `out` (null: in place), with or without the write-back to
// the pedestrians' arrays
static int launch_sfm(imgenv* h, hipStream_t s, const SfmDev* out, int publish) {
    DevWorld d = h->d;
    if (out) {
        d.sfm.p_out = out->p; d.sfm.v_out = out->v; d.sfm.dq_out = out->dq; d.sfm.dest_out = out->dest; d.sfm.last_out = out->last;
        d.sfm.nodes_out = out->nodes; d.sfm.n_nodes_out = out->n_nodes; d.sfm.treehash_out = out->treehash;
    }
    const int n_sfm = d.sfm.n, n_pairs = n_sfm * n_sfm;  // (of one world's crowd; a workgroup per world)
    const unsigned nw = (unsigned)h->W, pb = (unsigned)((n_pairs + SFM_MAX_AGENTS - 1) / SFM_MAX_AGENTS);
    if (n_pairs <= 4096) {  // small crowd: one launch
        TIMED(h, IMGENV_K_ORCA, s, (k_sfm<<<dim3(nw), dim3(SFM_MAX_AGENTS), sizeof(SfmNode) * SFM_LDS_NODES, s>>>(d, 0, publish)));
        h->launches += 1;
    } else {  // the n^2 pair terms (three correctly rounded atan2 each) spread over the chip between two one-workgroup launches
        const bool on = timing_on(h, IMGENV_K_ORCA);
        if (on) { if (int rc_ = timing_mark(h, IMGENV_K_ORCA, s, 0)) return rc_; }
        k_sfm<<<dim3(nw), dim3(SFM_MAX_AGENTS), sizeof(SfmNode) * SFM_LDS_NODES, s>>>(d, 1, publish);
        k_sfm<<<dim3(nw * pb), dim3(SFM_MAX_AGENTS), 0, s>>>(d, 2, publish);
        k_sfm<<<dim3(nw), dim3(SFM_MAX_AGENTS), sizeof(SfmNode) * SFM_LDS_NODES, s>>>(d, 3, publish);
        if (on) { if (int rc_ = timing_mark(h, IMGENV_K_ORCA, s, 1)) return rc_; }
        h->launches += 3;
    }
    return 0;
}
// Something is about to write the live crowd on stream st (a reset, a step in place, a device-side auto-reset): what was computed
// ahead is dropped, and the writer waits for the launch that may still be reading the live set
static int sfm_ahead_drop(imgenv* h, hipStream_t st) {
    if (h->sfm_ahead_valid) HIPCHK(hipStreamWaitEvent(st, h->ev_sfm, 0));
    h->sfm_ahead_valid = false;
    h->sfm_steps_since_reset = 0;
    return 0;
}

// the social-force crowd's step AHEAD, on its own stream (see imgenv_step_begin): queued once the step's own first kernels are out
static int sfm_launch_ahead(imgenv* h) {
    if (h->sfm_ahead_wait < 0) return 0;
    HIPCHK(hipStreamWaitEvent(h->sfm_stream, h->ev_sfm_in[h->sfm_ahead_wait], 0));
    h->sfm_ahead_wait = -1;
    if (int rc = launch_sfm(h, h->sfm_stream, &h->sfm_other, 0)) return rc;
    HIPCHK(hipEventRecord(h->ev_sfm, h->sfm_stream));
    h->sfm_ahead_valid = true;
    return 0;
}

extern "C" int imgenv_step_begin(imgenv_t* h, const float* actions, void* stream) {
    if (!h || !actions) FAIL(IMGENV_EINVAL, "null argument");
    if (!h->has_reset) FAIL(IMGENV_ESTATE, "step before reset");
    if (h->obs_forked) FAIL(IMGENV_ESTATE, "imgenv_step_begin twice without imgenv_step_end (reset the handle to recover from an aborted step)");
    if (int rc = check_device_flags(h)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = outputs_verify(h, st)) return rc;
    // (from here on the step's own kernels write output arrays; the chain's end seals them again.  A step that is begun and never
    // ended -- the caller's exchange failed -- must not read as "the caller wrote into imgenv_out" at the reset that recovers it)
    h->guard_sealed = false;
    DevWorld& d = h->d;
    h->launches = 0;
    // _step_ped_normal (img_env.cpp:304-359): the ORCA solve for this step ran on the side stream during the previous
    // step's views and was joined at the end of that step; its velocities are applied by k_integrate's pedestrian blocks
    if (h->cfg.ped_scene_type == IMGENV_SCENE_PEDSIM && h->d.sfm.n > 0) {  // PedScene::step + write-back (img_env.cpp:343-358)
        if (h->sfm_ahead) {
            // a crowd that ignores the robots: this step's state was computed during the last step (launch_sfm below) -- swap the
            // two sets and publish -- or, right behind a reset, is computed now, in place; then the NEXT step's goes out on its stream
            // Ordering against the caller's stream.  The launch ahead (below) reads the live set and WRITES the other one -- the set
            // that was live a step ago, which THAT step's k_sfm_publish (or in-place step) touched on the caller's stream.  Nothing else
            // orders the crowd's stream behind the caller's: a host that queues steps faster than the device drains them (an
            // asynchronous trainer, bench.py's timed loop) would let the crowd run several steps ahead and overwrite a set before it has
            // been published (rounds 3-5 did).  So every step leaves an event behind its access to the sets on the caller's stream,
            // by step parity, and the launch ahead waits for the LAST step's -- this step's publish only reads the live set, as the
            // launch ahead does, so the crowd may run up to one step ahead of the caller's stream -- or for this step's own when it
            // wrote the live set in place.
            const int cur = h->sfm_par;
            h->sfm_par ^= 1;
            bool in_place = false;
            if (h->sfm_ahead_valid) {
                SfmDev& a = h->d.sfm;
                SfmDev& b = h->sfm_other;
                std::swap(a.p, b.p); std::swap(a.v, b.v); std::swap(a.dq, b.dq); std::swap(a.dest, b.dest); std::swap(a.last, b.last);
                std::swap(a.nodes, b.nodes); std::swap(a.n_nodes, b.n_nodes); std::swap(a.treehash, b.treehash);
                a.p_out = a.p; a.v_out = a.v; a.dq_out = a.dq; a.dest_out = a.dest; a.last_out = a.last;
                a.nodes_out = a.nodes; a.n_nodes_out = a.n_nodes; a.treehash_out = a.treehash;
                HIPCHK(hipStreamWaitEvent(st, h->ev_sfm, 0));
                // (the event rides on the kernel's dispatch packet: a hipEventRecord behind it is a packet of its own, ~6 us of the caller's stream)
                hipExtLaunchKernelGGL(k_sfm_publish, dim3((unsigned)h->W), dim3(SFM_MAX_AGENTS), 0, st, nullptr, h->ev_sfm_in[cur], 0, d);
                h->launches += 1;
            } else {
                if (int rc = launch_sfm(h, st, nullptr, 1)) return rc;
                HIPCHK(hipEventRecord(h->ev_sfm_in[cur], st));
                in_place = true;
            }
            // (not in the first step behind a reset: a handle whose worlds are reset every other step -- many small worlds with
            // their own time limits -- would compute ahead what the next reset drops, and make that reset wait for it)
            // (queued at the END of this call, behind the step's own first launches: its five host calls in front of the move were
            // ~25 us in which the caller's stream sat idle -- sfm_launch_ahead)
            h->sfm_ahead_wait = h->sfm_steps_since_reset >= 1 ? (in_place ? cur : cur ^ 1) : -1;
            h->sfm_steps_since_reset += 1;
        } else {
            if (int rc = sfm_ahead_drop(h, st)) return rc;
            if (int rc = launch_sfm(h, st, nullptr, 1)) return rc;
        }
    }
    if (d.beep_on) {  // beep lottery + ERVO's evacuation term on top of the velocities the solve left (img_env.cpp:323-343)
        k_beep<<<dim3(h->W), dim3(BEEP_T), 0, st>>>(d, actions);
        k_evac<<<dim3(h->P), dim3(WAVE), 0, st>>>(d);
        h->launches += 2;
    }
    // _step_robot (img_env.cpp:388-410)
    // (a robot shard in SUM mode draws its own robots in this call anyway: the move goes into that launch, whoever runs the exchange)
    const bool fuse_ok = h->n_sub >= 1 && h->n_sub + 2 <= INT_ITEMS;
    const bool whole_call = h->in_step && !d.sharded && !h->comm && h->RL == h->R;  // imgenv_step on a handle that owns its world
    const bool fuse_small = (whole_call || d.sum_shard) && fuse_ok && (h->P == 0 ? h->RL <= 4096 : h->RL <= 1024);
    // (the move inside the raster launch of BIG handles, with the early observation gated on that launch, was measured again in round 6:
    // 95.0 -> 103.1 us per headline step, cfg-4 117-123 -> 126, cfg-5 280 -> 282: the move's serial chain in front of every robot's
    // raster costs more than the launch it saves)
    const bool fuse_move = fuse_small;
    // early-observation step (world.h): k_obs goes out with the move, on its side stream, instead of behind it -- behind a gate that
    // opens when the caller's stream reaches this step's move (world.h: sync; only where gates work: k_gate_probe)
    static const int force_early = getenv("IMGENV_EARLY_OBS") ? atoi(getenv("IMGENV_EARLY_OBS")) : -1;  // (measurement switch)
    const bool live_peds = h->NA == 0;  // (a social-force crowd a step ahead: see imgenv_create)
    const bool early_step = h->early && h->gates_work && !fuse_move && !h->chain_open && (live_peds ? h->sfm_ahead : h->orca_seq > 0) &&
                            h->view_seq > 0 && force_early != 0;
    if (early_step) h->gate_seq += 1;
    // k_obs beside the move instead of behind it (world.h): it needs nothing of this step but the actions.  It waits behind a gate
    // that opens when the caller's stream reaches the move (k_gate): everything queued there in front of the step is then complete
    // -- whoever writes the actions, whoever still reads the last step's outputs, the last chain's views.  Queued BEHIND the move --
    // a gate must follow the kernel that opens it in queue order; what the caller's stream forks behind the move is the solve alone.
    auto early_obs = [&]() -> int {
        h->chain_open = true;
        if (!h->fork_on_move) HIPCHK(hipEventRecord(h->ev_fork, st));
        h->fork_on_move = false;
        k_gate<<<dim3(1), dim3(WAVE), 0, h->side2>>>(d.sync, h->gate_seq, d.err);
        h->launches += 1;
        d.obs_early = live_peds ? 2 : 1;
        d.obs_actions = actions;
        d.obs_n_sub = h->n_sub;
        d.ped_snap_in = h->ped_snap[(h->orca_seq - 1) & 1];
        d.rec_snap_in = h->rec_snap[(h->view_seq - 1) & 1];
        const int rc = launch_obs_kernel(h, h->side2);
        d.obs_early = 0;
        if (rc) return rc;
        h->obs_forked = true;
        h->early_step = true;
        return 0;
    };
    if (fuse_move && d.sum_shard) {  // k_move_raster over the shard's own robots and the pedestrians, now (in front of the exchange); the observation behind it
        h->move_actions = actions;
        if (int rc = launch_rasters(h, st, 0, true, true, true)) return rc;
        h->launches += 1;
        set_tail_fields(h, 0, h->elapsed + 1);
        if (h->P > 0)
            if (int rc = launch_obs(h, st)) return rc;
        HIPCHK(hipGetLastError());
        return sfm_launch_ahead(h);
    }
    if (fuse_move) {  // k_move_raster, launched by launch_views (the fork of the side stream with it)
        h->move_pending = true;
        h->move_actions = actions;
        return sfm_launch_ahead(h);
    }
    {   // ... and the pedestrians' move (img_env.cpp:343-358) in the same launch
        const bool peds = h->P > 0 && (h->NA > 0 || h->cfg.ped_scene_type == IMGENV_SCENE_DATASET);
        if (h->n_sub >= 1 && h->n_sub + 2 <= INT_ITEMS) {
            const int nb_robot = (h->RL + INT_ROBOTS - 1) / INT_ROBOTS, nb_ped = peds ? (h->P + INT_G * INT_ROBOTS - 1) / (INT_G * INT_ROBOTS) : 0;
            // (the fork of the side streams follows right behind the move, on its dispatch packet)
            h->fork_on_move = h->P > 0 && !h->serial && !h->chain_open;
            TIMED(h, IMGENV_K_INTEGRATE, st, (hipExtLaunchKernelGGL(k_integrate, dim3(nb_robot + nb_ped), dim3(INT_G * INT_ROBOTS), 0, st, nullptr,
                                                                   h->fork_on_move ? h->ev_fork : nullptr, 0, d, actions, nb_robot, h->n_sub, h->elapsed,
                                                                   early_step ? h->gate_seq : 0u)));
        } else {
            const int nb_robot = (h->RL + 127) / 128, nb_ped = peds ? (h->P + 127) / 128 : 0;
            TIMED(h, IMGENV_K_INTEGRATE, st, (k_integrate_serial<<<dim3(nb_robot + nb_ped), dim3(128), 0, st>>>(d, actions, nb_robot, h->elapsed)));
        }
    }
    h->launches += 1;
    set_tail_fields(h, 0, h->elapsed + 1);  // imgenv_step_end counts the step; k_obs goes out before that
    if (early_step) {
        if (int rc = early_obs()) return rc;
    } else if (h->P > 0) {
        if (int rc = launch_obs(h, st)) return rc;
    }
    // a robot shard in SUM mode (world.h: sum_shard) draws its own robots -- and the pedestrians -- NOW, in front of the exchange:
    // their records then carry the cells they cover to the other ranks
    if (d.sum_shard)
        if (int rc = launch_rasters(h, st, 0, false, true, true)) return rc;
    HIPCHK(hipGetLastError());
    return sfm_launch_ahead(h);
}

extern "C" int imgenv_step_end(imgenv_t* h, void* stream) {
    if (!h) FAIL(IMGENV_EINVAL, "null argument");
    if (!h->has_reset) FAIL(IMGENV_ESTATE, "step before reset");
    h->elapsed += 1;  // TimeLimitWrapper._elapsed_steps (base.py:224)
    if (h->stamp) {  // this step's stamps get a new tag
        h->stamp_seq += 1;
        h->d.stamp_tag = h->stamp_seq % STAMP_TAGS + 1;
    }
    return launch_views(h, (hipStream_t)stream, 0);
}

extern "C" int imgenv_step(imgenv_t* h, const float* actions, void* stream) { return imgenv_step_flags(h, actions, 0u, stream); }

extern "C" int imgenv_step_flags(imgenv_t* h, const float* actions, uint32_t flags, void* stream) {
    (void)flags;  // (IMGENV_STEP_ACTIONS_READY: accepted, and since round 6 without effect -- include/imgenv.h)
    if (h) h->in_step = true;  // (begin and end in one call: the actions outlive the move whoever launches it)
    const int rc_begin = imgenv_step_begin(h, actions, stream);
    if (h) h->in_step = false;
    if (rc_begin) return rc_begin;
    if (h->comm) {  // the one exchange of a robot-sharded world: records of all robots, in place
        const size_t count = (size_t)h->RL * IMGENV_RECORD_DOUBLES;
        ncclResult_t e = ncclSuccess;
        TIMED(h, IMGENV_K_EXCHANGE, (hipStream_t)stream,
              e = rccl_api()->all_gather(h->d.rec + (size_t)h->r0 * IMGENV_RECORD_DOUBLES, h->d.rec, count, ncclDouble, h->comm, (hipStream_t)stream));
        if (e != ncclSuccess) FAIL(IMGENV_EDEVICE, "ncclAllGather: %s", rccl_api()->err(e));
    }
    return imgenv_step_end(h, stream);
}

// FNV-1a over everything spawn_world reads of a spawn cfg
static uint64_t spawn_cfg_fingerprint(const imgenv_spawn_cfg& c) {
    uint64_t hsh = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t q = 0; q < n; q++) hsh = (hsh ^ b[q]) * 1099511628211ull;
    };
    mix(&c.n_robots, sizeof(int32_t) * 3);
    mix(&c.clearance, sizeof(double));
    mix(&c.target_min_dist, sizeof(double));
    mix(c.circle_ranges, sizeof(c.circle_ranges));
    mix(&c.go_back, sizeof(int32_t) * 2);
    for (int a = 0; a < c.n_robots + c.n_peds; a++) {
        const imgenv_spawn_agent& g = c.agents[a];
        mix(&g.begin_type, sizeof(int32_t) * 2);
        mix(g.begin, sizeof(g.begin));
        mix(g.target, sizeof(g.target));
        mix(&g.module_size, sizeof(double));
        mix(&g.n_begin_multi, sizeof(int32_t) * 2);
        if (g.begin_multi) mix(g.begin_multi, sizeof(double) * 6 * (size_t)g.n_begin_multi);
        if (g.target_multi) mix(g.target_multi, sizeof(double) * 6 * (size_t)g.n_target_multi);
    }
    for (int q = 0; q < c.n_obstacles; q++) {
        mix(&c.obstacles[q].shape, sizeof(int32_t) * 2);
        mix(c.obstacles[q].size_range, sizeof(c.obstacles[q].size_range));
        mix(c.obstacles[q].pose, sizeof(c.obstacles[q].pose));
    }
    return hsh;
}

extern "C" int imgenv_step_autoreset(imgenv_t* h, const float* actions, const imgenv_spawn_cfg* cfg, uint64_t seed0, int32_t* worlds_out,
                                     int32_t cap, int32_t* n_out, void* stream) {
    if (!h || !actions || !n_out) FAIL(IMGENV_EINVAL, "null argument");
    if (int rc = spawn_cfg_check(cfg)) return rc;
    if (cfg->n_robots != h->Rw || cfg->n_peds != h->Pw)
        FAIL(IMGENV_EINVAL, "spawn cfg is for %d robots / %d pedestrians, a world of this handle has %d / %d", cfg->n_robots,
             cfg->n_peds, h->Rw, h->Pw);
    if (h->RL != h->R) FAIL(IMGENV_EINVAL, "imgenv_step_autoreset needs all robots of every world on this handle");
    *n_out = 0;
    static const bool trace = getenv("IMGENV_TRACE_RESET") != nullptr;  // where the host's time goes, every 200 calls
    double* acc = h->trace_acc;
    long &calls = h->trace_calls, &resets = h->trace_resets;
    std::chrono::steady_clock::time_point tp[5];
    if (trace) tp[0] = std::chrono::steady_clock::now();
    if (int rc = imgenv_step(h, actions, stream)) return rc;
    hipStream_t st = (hipStream_t)stream;
    k_finished<<<dim3(1), dim3(1024), 0, st>>>(h->d);
    if (int rc = outputs_seal(h, st)) return rc;  // (k_finished writes imgenv_out.step_all_down behind the step's own seal)
    HIPCHK(hipGetLastError());
    {   // while the device works: the placements of the next seeds (a placement depends on its seed alone, not on the world)
        const uint64_t fp = spawn_cfg_fingerprint(*cfg);
        if (fp != h->spawn_ahead_cfg) h->spawn_ahead.clear();
        h->spawn_ahead_cfg = fp;
        for (auto it = h->spawn_ahead.begin(); it != h->spawn_ahead.end();)  // seeds the caller has moved past
            it = (it->first - seed0 >= (uint64_t)h->spawn_ahead_n) ? h->spawn_ahead.erase(it) : std::next(it);
        for (int q = 0; q < h->spawn_ahead_n; q++) {
            std::unique_ptr<SpawnOut>& slot = h->spawn_ahead[seed0 + (uint64_t)q];
            if (slot) continue;
            slot.reset(new SpawnOut);
            if (spawn_world(*cfg, seed0 + (uint64_t)q, *slot)) slot->batch.struct_size = 0;  // could not be placed: reported if used
        }
    }
    if (trace) tp[1] = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(st));  // NeverStopWrapper reads the dones here too (base.py:205)
    if (trace) tp[2] = std::chrono::steady_clock::now();
    const int n = h->finished_host[0];
    if (n < 0 || n > h->W) FAIL(IMGENV_EDEVICE, "finished-world list is corrupt (%d)", n);
    int rc = IMGENV_OK;
    if (n > 0) {
        std::vector<int32_t> worlds((size_t)n);
        for (int q = 0; q < n; q++) worlds[q] = h->finished_host[1 + q];
        std::sort(worlds.begin(), worlds.end());  // the device lists them in no particular order; seeds go by ascending index
        // every placement first: a seed that cannot be placed fails the call BEFORE anything is handed out or taken from the
        // pre-drawn placements (the step itself has been applied; no world has been reset; *n_out stays 0; a repeated call
        // meets the same seeds)
        std::vector<SpawnOut*> use((size_t)n, nullptr);
        std::vector<std::unique_ptr<SpawnOut>> fresh;
        for (int q = 0; q < n; q++) {
            auto it = h->spawn_ahead.find(seed0 + (uint64_t)q);
            if (it != h->spawn_ahead.end() && it->second->batch.struct_size != 0) {
                use[q] = it->second.get();
            } else {
                fresh.emplace_back(new SpawnOut);
                if (const char* why = spawn_world(*cfg, seed0 + (uint64_t)q, *fresh.back()))
                    FAIL(IMGENV_EINVAL, "spawn of world %d (seed %llu): %s", worlds[q], (unsigned long long)(seed0 + (uint64_t)q), why);
                use[q] = fresh.back().get();
            }
        }
        std::vector<imgenv_reset_batch> batches((size_t)n);
        for (int q = 0; q < n; q++) batches[q] = use[q]->batch;
        h->spawn_ahead_n = std::min(256, std::max(8, 2 * n));  // twice what this step needed
        if (trace) tp[3] = std::chrono::steady_clock::now();
        rc = imgenv_reset_worlds(h, n, worlds.data(), batches.data(), stream);
        if (rc == IMGENV_OK) {
            *n_out = n;
            for (int q = 0; q < n && worlds_out && q < cap; q++) worlds_out[q] = worlds[q];
            for (int q = 0; q < n; q++) h->spawn_ahead.erase(seed0 + (uint64_t)q);  // consumed
        }
    } else if (trace) {
        tp[3] = std::chrono::steady_clock::now();
    }
    if (trace) {
        tp[4] = std::chrono::steady_clock::now();
        for (int q = 0; q < 4; q++) acc[q] += std::chrono::duration<double, std::micro>(tp[q + 1] - tp[q]).count();
        resets += n;
        if (++calls % 200 == 0) {
            fprintf(stderr, "[imgenv_step_autoreset] %ld calls, %.1f worlds reset/call: step launches + placements ahead %.1f us, wait for the device %.1f us, "
                            "finished list %.1f us, reset launches %.1f us per call\n", calls, (double)resets / calls, acc[0] / calls, acc[1] / calls,
                    acc[2] / calls, acc[3] / calls);
        }
    }
    return rc;
}


// ---------------------------------------------------------------------------------------- device-side auto-reset
template <typename T>
static int sd_alloc(imgenv* h, T** out, size_t n) {
    return dev_alloc(h, out, n);
}
static int spawn_device_setup(imgenv* h, const imgenv_spawn_cfg* cfg, uint64_t seed0, hipStream_t st) {
    const int na = cfg->n_robots + cfg->n_peds, nob = cfg->n_obstacles;
    if (na > SPAWN_MAX_AGENTS || nob > SPAWN_MAX_OBST)
        FAIL(IMGENV_EINVAL, "device-side auto-reset places at most %d agents and %d obstacles per world", SPAWN_MAX_AGENTS, SPAWN_MAX_OBST);
    if (h->cfg.ped_scene_type == IMGENV_SCENE_DATASET)
        FAIL(IMGENV_EINVAL, "device-side auto-reset: dataset scenes (recorded crowds come with the reset call) are reset by the host (imgenv_step_autoreset)");
    const bool sfm = h->cfg.ped_scene_type == IMGENV_SCENE_PEDSIM && h->d.sfm.n > 0;
    if (sfm && (h->W < 2 || !h->d.sfm.n_obs_w))
        FAIL(IMGENV_EINVAL, "device-side auto-reset of a pedscene world needs a handle of several worlds (n_worlds > 1: the per-world crowd tables)");
    if (sfm && !cfg->ignore_obstacle && nob > h->sfm_cap_obs)
        FAIL(IMGENV_EINVAL, "device-side auto-reset: %d obstacles, a pedscene world of this handle has room for %d", nob, h->sfm_cap_obs);
    if (h->Pw > 0 && (h->traj_cap < 2 || !h->d_traj)) FAIL(IMGENV_ESTATE, "device-side auto-reset: reset every world once first (trajectories of two points)");
    if (!h->sd_storage) {
        h->sd_storage = new SpawnDev();
        h->sd_delete = [](void* p) { delete (SpawnDev*)p; };
    }
    SpawnDev& c = *(SpawnDev*)h->sd_storage;
    if (h->sd_ready) {  // another spawn cfg: the old pool goes (a curriculum that alternates cfgs would otherwise grow until imgenv_destroy)
        HIPCHK(hipStreamSynchronize(h->side3));
        HIPCHK(hipStreamSynchronize(st));
        h->sd_ready = false;
        h->fill_pending = false;
        SpawnDev o = c;
        DevSpawnAgent* oa = (DevSpawnAgent*)o.agents; DevSpawnObstacle* oo = (DevSpawnObstacle*)o.obstacles; double* om = (double*)o.multi;
        dev_free(h, oa); dev_free(h, oo); dev_free(h, om);
        dev_free(h, o.slot_serial); dev_free(h, o.consumed); dev_free(h, o.slot_status); dev_free(h, o.s_agents); dev_free(h, o.s_obst);
        dev_free(h, o.s_inst); dev_free(h, o.s_rvo); dev_free(h, o.s_nodes); dev_free(h, o.s_rvo_n); dev_free(h, o.s_seg);
        dev_free(h, o.fin_list); dev_free(h, o.fin_n); dev_free(h, o.inst_out); dev_free(h, o.place_agents); dev_free(h, o.place_obst);
        dev_free(h, o.place_serial); dev_free(h, o.w_inst); dev_free(h, o.w_inst_valid);
    }
    memset(&c, 0, sizeof(c));
    c.n_robots = cfg->n_robots; c.n_peds = cfg->n_peds; c.n_obstacles = nob; c.go_back = cfg->go_back; c.ignore_obstacle = cfg->ignore_obstacle;
    c.rvo = h->NA > 0 ? 1 : 0;
    c.sfm = sfm ? 1 : 0;
    c.clearance = cfg->clearance; c.target_min_dist = cfg->target_min_dist; c.circle0 = cfg->circle_ranges[0]; c.circle1 = cfg->circle_ranges[1];
    std::vector<DevSpawnAgent> ag((size_t)(na ? na : 1));
    std::vector<double> multi;
    for (int a = 0; a < na; a++) {
        const imgenv_spawn_agent& g = cfg->agents[a];
        DevSpawnAgent& o = ag[a];
        memset(&o, 0, sizeof(o));
        o.begin_type = g.begin_type; o.target_type = g.target_type; o.module_size = g.module_size;
        memcpy(o.begin, g.begin, sizeof(o.begin));
        memcpy(o.target, g.target, sizeof(o.target));
        if (g.begin_type == IMGENV_POSE_RANGE_MULTI) {
            if (g.n_begin_multi < 1 || !g.begin_multi) FAIL(IMGENV_EINVAL, "range_multi start without ranges");
            o.begin_multi = (int)(multi.size() / 6); o.n_begin_multi = g.n_begin_multi;
            multi.insert(multi.end(), g.begin_multi, g.begin_multi + 6 * (size_t)g.n_begin_multi);
        }
        if (g.target_type == IMGENV_POSE_RANGE_MULTI) {
            if (g.n_target_multi < 1 || !g.target_multi) FAIL(IMGENV_EINVAL, "range_multi target without ranges");
            o.target_multi = (int)(multi.size() / 6); o.n_target_multi = g.n_target_multi;
            multi.insert(multi.end(), g.target_multi, g.target_multi + 6 * (size_t)g.n_target_multi);
        }
    }
    if (multi.empty()) multi.push_back(0.0);
    std::vector<DevSpawnObstacle> ob((size_t)(nob ? nob : 1));
    for (int q = 0; q < nob; q++) {
        ob[q].shape = cfg->obstacles[q].shape; ob[q].pose_type = cfg->obstacles[q].pose_type;
        memcpy(ob[q].size_range, cfg->obstacles[q].size_range, sizeof(ob[q].size_range));
        memcpy(ob[q].pose, cfg->obstacles[q].pose, sizeof(ob[q].pose));
    }
    RTRY(dev_upload(h, &c.agents, ag));
    RTRY(dev_upload(h, &c.obstacles, ob));
    RTRY(dev_upload(h, &c.multi, multi));
    const int W = h->W, S = (SPAWN_FILL_PERIOD + 2) * std::max(64, W);  // every world can need a placement in a single step; see SPAWN_FILL_PERIOD
    c.S = S;
    c.seed0 = seed0;
    c.cap_o = std::max(16, std::min(SPAWN_BSP_CAP, 16 * std::max(nob, 1)));
    c.cap_n = c.cap_o;
    RTRY(dev_alloc(h, &c.slot_serial, (size_t)S, 0xFF));
    RTRY(dev_alloc(h, &c.consumed, 2));
    RTRY(dev_alloc(h, &c.slot_status, (size_t)S));
    RTRY(dev_alloc(h, &c.s_agents, (size_t)S * (na ? na : 1)));
    RTRY(dev_alloc(h, &c.s_obst, (size_t)S * (nob ? nob : 1)));
    RTRY(dev_alloc(h, &c.s_inst, (size_t)S * (nob ? nob : 1)));
    RTRY(dev_alloc(h, &c.s_rvo, (size_t)S * c.cap_o));
    RTRY(dev_alloc(h, &c.s_nodes, (size_t)S * c.cap_n));
    RTRY(dev_alloc(h, &c.s_rvo_n, (size_t)S * 4));
    RTRY(dev_alloc(h, &c.s_seg, (size_t)S * (nob ? nob : 1) * 4));
    RTRY(dev_alloc(h, &c.fin_list, (size_t)W));
    RTRY(dev_alloc(h, &c.fin_n, 1));
    RTRY(dev_alloc(h, &c.inst_out, (size_t)W * (nob ? nob : 1)));
    RTRY(dev_alloc(h, &c.place_agents, (size_t)W * (na ? na : 1)));
    RTRY(dev_alloc(h, &c.place_obst, (size_t)W * (nob ? nob : 1)));
    RTRY(dev_alloc(h, &c.place_serial, (size_t)W, 0xFF));
    RTRY(dev_alloc(h, &c.w_inst, (size_t)W * (nob ? nob : 1)));
    RTRY(dev_alloc(h, &c.w_inst_valid, (size_t)W));
    // the per-world RVO tables take over from the host's copies: room for any placement's polygons in every world
    if (h->NA > 0 && (h->cap_obst < c.cap_o || h->cap_nodes < c.cap_n)) {
        HIPCHK(hipStreamSynchronize(st));
        const int co = std::max(h->cap_obst, c.cap_o), cn = std::max(h->cap_nodes, c.cap_n);
        RvoObstDev* no = nullptr;
        RvoNodeDev* nn = nullptr;
        RTRY(dev_alloc(h, &no, (size_t)co * W));
        RTRY(dev_alloc(h, &nn, (size_t)cn * W));
        // every world's slice laid out on the host, then ONE copy per table (two blocking copies per world were 4 144 copies and
        // 21 ms for the 2048 envs of the shipped geometry)
        std::vector<RvoObstHost> all_o((size_t)co * W);
        std::vector<RvoNodeHost> all_n((size_t)cn * W);
        for (int q = 0; q < W; q++) {
            const RvoObstacles& rq = h->rvos[q];
            if ((int)rq.ob.size() > co || (int)rq.nodes.size() > cn) FAIL(IMGENV_ESTATE, "world %d holds more RVO obstacle vertices than a placement can have", q);
            std::copy(rq.ob.begin(), rq.ob.end(), all_o.begin() + (size_t)q * co);
            std::copy(rq.nodes.begin(), rq.nodes.end(), all_n.begin() + (size_t)q * cn);
            h->wobst[q] = q * co;
            h->wobst[h->W + q] = q * cn;
            h->wobst[2 * h->W + q] = (int)rq.ob.size();
            h->wobst[3 * h->W + q] = rq.root;
        }
        HIPCHK(hipMemcpy(no, all_o.data(), sizeof(RvoObstDev) * all_o.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(nn, all_n.data(), sizeof(RvoNodeDev) * all_n.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_wobst, h->wobst.data(), sizeof(int) * h->wobst.size(), hipMemcpyHostToDevice));
        dev_free(h, h->d_obst);  // (st was synchronised above and every side stream is joined to it at the end of a step)
        dev_free(h, h->d_nodes);
        h->d_obst = no; h->d_nodes = nn; h->cap_obst = co; h->cap_nodes = cn;
        h->d.obst = no; h->d.onodes = nn;
    }
    c.cap_o = h->NA > 0 ? h->cap_obst : c.cap_o;  // the worlds' slices and the slots' share one stride
    c.cap_n = h->NA > 0 ? h->cap_nodes : c.cap_n;
    if (h->NA > 0) {  // (slot arrays were sized with the smaller capacity: again with the handle's)
        dev_free(h, c.s_rvo);
        dev_free(h, c.s_nodes);
        RTRY(dev_alloc(h, &c.s_rvo, (size_t)S * c.cap_o));
        RTRY(dev_alloc(h, &c.s_nodes, (size_t)S * c.cap_n));
    }
    c.world_epoch = h->d_world_epoch;
    c.n_obst_w = h->d_wobst + 2 * (size_t)W;
    c.oroot_w = h->d_wobst + 3 * (size_t)W;
    c.w_obst = h->d_obst;
    c.w_nodes = h->d_nodes;
    c.traj = h->d_traj;
    c.traj_len = h->d_traj_len;
    c.traj_cap = h->traj_cap;
    if (!h->side3) {
        HIPCHK(hipStreamCreateWithFlags(&h->side3, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&h->ev_fill, hipEventDisableTiming | hipEventDisableSystemFence));
        HIPCHK(hipEventCreateWithFlags(&h->ev_consumed, hipEventDisableTiming | hipEventDisableSystemFence));
    }
    HIPCHK(hipStreamSynchronize(st));  // (set-up only: the uploads above were synchronous copies)
    k_spawn_fill<<<dim3(S), dim3(WAVE), 0, h->side3>>>(c);
    HIPCHK(hipEventRecord(h->ev_fill, h->side3));
    h->fill_pending = true;  // the next chain waits for this fill in front of its k_respawn, whatever fill_due says
    h->fill_due = 0;
    h->sd_ready = true;
    return 0;
}

// A host-side reset between two device-side steps may have re-laid the tables k_respawn writes into: stage_world grows the
// trajectory table when a batch brings longer waypoint lists, put_world_rvo the per-world RVO slices (after imgenv_reset has
// handed them back to the host).  SpawnDev travels by value with every launch, so its copies of those pointers and strides are
// simply taken from the handle again; a changed RVO stride also moves the pool's slot arrays (slots and worlds share one
// stride), whose placements are then drawn again.
static int spawn_dev_refresh(imgenv* h, hipStream_t st) {
    SpawnDev& c = *(SpawnDev*)h->sd_storage;
    const bool stride = h->NA > 0 && (c.cap_o != h->cap_obst || c.cap_n != h->cap_nodes);
    const bool moved = c.traj != h->d_traj || c.traj_len != h->d_traj_len || c.traj_cap != h->traj_cap ||
                       (h->NA > 0 && (c.w_obst != h->d_obst || c.w_nodes != h->d_nodes));
    if (!stride && !moved) return 0;
    if (stride) {
        HIPCHK(hipStreamSynchronize(h->side3));  // (a fill in flight writes the old slot arrays)
        HIPCHK(hipStreamSynchronize(st));
        c.cap_o = h->cap_obst;
        c.cap_n = h->cap_nodes;
        dev_free(h, c.s_rvo);  // (both streams are idle: nothing reads the old slot arrays any more)
        dev_free(h, c.s_nodes);
        RTRY(dev_alloc(h, &c.s_rvo, (size_t)c.S * c.cap_o));
        RTRY(dev_alloc(h, &c.s_nodes, (size_t)c.S * c.cap_n));
        HIPCHK(hipMemsetAsync(c.slot_serial, 0xFF, sizeof(unsigned long long) * (size_t)c.S, st));  // every slot is drawn again (the chain's own fill, behind this on st's event)
        h->fill_due = 0;
    }
    c.w_obst = h->d_obst;
    c.w_nodes = h->d_nodes;
    c.n_obst_w = h->d_wobst + 2 * (size_t)h->W;
    c.oroot_w = h->d_wobst + 3 * (size_t)h->W;
    c.traj = h->d_traj;
    c.traj_len = h->d_traj_len;
    c.traj_cap = h->traj_cap;
    return 0;
}

// one step + the reset of whatever it finished, everything queued on `st` and the handle's side streams (all joined again)
static int autoreset_device_chain(imgenv* h, const float* actions, hipStream_t st) {
    SpawnDev& c = *(SpawnDev*)h->sd_storage;
    DevWorld& d = h->d;
    const int W = h->W, nob = c.n_obstacles;
    // the pool, underneath the step: the slots whose placements earlier steps handed out -- on every SPAWN_FILL_PERIOD-th call
    // (two event operations and a launch less on the others: each costs the caller's stream a dependency bubble and the host a call)
    const bool fill = h->fill_due <= 0;
    if (fill) {
        HIPCHK(hipEventRecord(h->ev_consumed, st));
        HIPCHK(hipStreamWaitEvent(h->side3, h->ev_consumed, 0));
        k_spawn_fill<<<dim3(c.S), dim3(WAVE), 0, h->side3>>>(c);
        HIPCHK(hipEventRecord(h->ev_fill, h->side3));
        h->fill_pending = true;
        h->fill_due = SPAWN_FILL_PERIOD;
    }
    h->fill_due -= 1;
    if (int rc = imgenv_step(h, actions, st)) return rc;
    k_finished_dev<<<dim3(1), dim3(1024), 0, st>>>(d, c);
    // (also the set-up's fill, which runs on the side stream whatever this call's fill_due was: k_respawn must not meet half-drawn slots)
    if (h->fill_pending) HIPCHK(hipStreamWaitEvent(st, h->ev_fill, 0));
    h->fill_pending = false;
    k_respawn<<<dim3(W), dim3(WAVE), 0, st>>>(d, c, h->elapsed);
    // grids for a guess of the finished worlds (four times the last count; the kernels stride over the rest if there are more)
    const int last_n = h->finished_host[0];  // (page-locked, written by k_finished_dev: stale by a step or two)
    const int guess = std::min(W, std::max(16, 4 * std::max(last_n, 0)));
    const int restore_blocks = 4 * MAP_BLOCKS;  // per world
    if (h->pow2) k_restore_maps_dev<true><<<dim3((unsigned)(guess * restore_blocks)), dim3(256), 0, st>>>(d, c, h->d_static_map, h->stamp ? 1 : h->sum ? 2 : 0, restore_blocks);
    else k_restore_maps_dev<false><<<dim3((unsigned)(guess * restore_blocks)), dim3(256), 0, st>>>(d, c, h->d_static_map, h->stamp ? 1 : h->sum ? 2 : 0, restore_blocks);
    if (nob > 0) {
        const int parts = 4;
        if (h->pow2) k_reset_obstacles<true><<<dim3((unsigned)(guess * nob * parts)), dim3(256), 0, st>>>(d, c.inst_out, h->stamp ? 1 : h->sum ? 2 : 0, c.fin_n, nob, parts, c.w_inst, c.w_inst_valid);
        else k_reset_obstacles<false><<<dim3((unsigned)(guess * nob * parts)), dim3(256), 0, st>>>(d, c.inst_out, h->stamp ? 1 : h->sum ? 2 : 0, c.fin_n, nob, parts, c.w_inst, c.w_inst_valid);
    }
    HIPCHK(hipGetLastError());
    if (h->stamp) {
        // The step's rasters have stamped the finished worlds' agents where they stood BEFORE the reset, with this step's tag,
        // and the map restore above only puts the obstacles' cells back: the reset's own rasters and views get a tag of their
        // own, under which those stamps have expired like any older ones (two tags per step; the sweep comes round accordingly)
        h->stamp_seq += 1;
        d.stamp_tag = h->stamp_seq % STAMP_TAGS + 1;
        if (h->stamp_seq % STAMP_TAGS == 0) {
            const unsigned blocks = (unsigned)((d.act_cells / 4 + 255) / 256 + 1);
            k_cell_base<<<dim3(blocks), dim3(256), 0, st>>>(d);
        }
    }
    // the launches behind: sized for every world, the list and its length read from device memory
    set_active(h, c.fin_list, W);
    d.act_n_dev = c.fin_n;
    d.ptraj = h->d_traj;
    d.traj_cap = h->traj_cap;
    h->launches += 4;
    const int rc = launch_views(h, st, 1);
    set_active(h, nullptr, 0);
    d.act_n_dev = nullptr;
    return rc;
}

extern "C" int imgenv_step_autoreset_device(imgenv_t* h, const float* actions, const imgenv_spawn_cfg* cfg, uint64_t seed0, void* stream) {
    if (!h || !actions) FAIL(IMGENV_EINVAL, "null argument");
    if (int rc = spawn_cfg_check(cfg)) return rc;
    if (cfg->n_robots != h->Rw || cfg->n_peds != h->Pw)
        FAIL(IMGENV_EINVAL, "spawn cfg is for %d robots / %d pedestrians, a world of this handle has %d / %d", cfg->n_robots,
             cfg->n_peds, h->Rw, h->Pw);
    if (h->RL != h->R) FAIL(IMGENV_EINVAL, "imgenv_step_autoreset_device needs all robots of every world on this handle");
    if (!h->has_reset) FAIL(IMGENV_ESTATE, "step before reset");
    if (h->obs_forked) FAIL(IMGENV_ESTATE, "imgenv_step_autoreset_device between imgenv_step_begin and imgenv_step_end");
    hipStream_t st = (hipStream_t)stream;
    if (h->sfm_ahead) {  // kernels reset finished worlds here, crowds included, without the host knowing which: the crowd steps in place
        if (int rc = sfm_ahead_drop(h, st)) return rc;
        h->sfm_ahead = false;
    }
    const uint64_t fp = spawn_cfg_fingerprint(*cfg);
    if (!h->sd_ready || fp != h->sd_fp) {  // the first call fixes seed0: the k-th world reset from now on takes placement seed0 + k
        if (int rc = spawn_device_setup(h, cfg, seed0, st)) return rc;
        h->sd_fp = fp;
    }
    if (int rc = spawn_dev_refresh(h, st)) return rc;
    if (!h->d_act_list) RTRY(dev_alloc(h, &h->d_act_list, (size_t)h->W));
    {   // how many worlds the last steps reset (page-locked, written by k_finished_dev; stale by a step or two: a hint only)
        const int last = h->finished_host[0];
        h->act_hint = std::max(8, 2 * std::max(last, 0)) * std::max(std::max(h->Rw, h->Pw), 1);  // (twice the last count: with four times, 64 worlds of 4 pedestrians sat ON the 1024 threshold and flipped between the kernel variants)
    }
    const int rc = autoreset_device_chain(h, actions, st);
    if (rc == IMGENV_OK) h->dev_reset_used = true;
    return rc;
}

// What the last imgenv_step_autoreset_device did (for checkers and hosts that want to know; synchronises `stream`): the worlds it
// reset, ascending (up to cap of them), their number, and the placement number the first of them took.
extern "C" int imgenv_autoreset_last(imgenv_t* h, int32_t* worlds_out, int32_t cap, int32_t* n_out, uint64_t* first_placement, void* stream) {
    if (!h || !n_out) FAIL(IMGENV_EINVAL, "null argument");
    if (!h->sd_ready) FAIL(IMGENV_ESTATE, "no imgenv_step_autoreset_device yet");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (int rc = check_device_flags(h)) return rc;
    SpawnDev& c = *(SpawnDev*)h->sd_storage;
    const int n = h->finished_host[0];
    *n_out = n;
    for (int q = 0; q < n && worlds_out && q < cap; q++) worlds_out[q] = h->finished_host[1 + q];
    if (first_placement) {
        unsigned long long two[2];
        HIPCHK(hipMemcpy(two, c.consumed, sizeof(two), hipMemcpyDeviceToHost));
        *first_placement = two[1];
    }
    return IMGENV_OK;
}

// The placement world `world` currently runs, as its device-side reset received it (the arrays of imgenv_spawn(); any may be
// NULL), and its number.  Synchronises the device.
extern "C" int imgenv_world_placement(imgenv_t* h, int32_t world, uint64_t* placement, double* robot_pose, double* robot_goal, double* ped_pose,
                                      double* ped_goal, double* ped_traj, int32_t* ped_traj_len, int32_t* obs_shape, float* obs_size, double* obs_pose) {
    if (!h) FAIL(IMGENV_EINVAL, "null argument");
    if (!h->sd_ready) FAIL(IMGENV_ESTATE, "no imgenv_step_autoreset_device yet");
    if (world < 0 || world >= h->W) FAIL(IMGENV_EINVAL, "world out of range");
    HIPCHK(hipDeviceSynchronize());
    SpawnDev& c = *(SpawnDev*)h->sd_storage;
    const int nr = c.n_robots, np = c.n_peds, na = nr + np, nob = c.n_obstacles;
    unsigned long long serial = 0;
    HIPCHK(hipMemcpy(&serial, c.place_serial + world, sizeof(serial), hipMemcpyDeviceToHost));
    if (placement) *placement = serial;
    if (serial == ~0ull) FAIL(IMGENV_ESTATE, "world %d has not been reset by the device yet", world);
    std::vector<SlotAgent> ag((size_t)(na ? na : 1));
    std::vector<SlotObstacle> ob((size_t)(nob ? nob : 1));
    if (na) HIPCHK(hipMemcpy(ag.data(), c.place_agents + (size_t)world * na, sizeof(SlotAgent) * na, hipMemcpyDeviceToHost));
    if (nob) HIPCHK(hipMemcpy(ob.data(), c.place_obst + (size_t)world * nob, sizeof(SlotObstacle) * nob, hipMemcpyDeviceToHost));
    for (int i = 0; i < nr; i++) {
        if (robot_pose) { robot_pose[4 * i] = ag[i].x; robot_pose[4 * i + 1] = ag[i].y; robot_pose[4 * i + 2] = ag[i].qz; robot_pose[4 * i + 3] = ag[i].qw; }
        if (robot_goal) { robot_goal[2 * i] = ag[i].gx; robot_goal[2 * i + 1] = ag[i].gy; }
    }
    for (int j = 0; j < np; j++) {
        const SlotAgent& a = ag[nr + j];
        if (ped_pose) { ped_pose[4 * j] = a.x; ped_pose[4 * j + 1] = a.y; ped_pose[4 * j + 2] = a.qz; ped_pose[4 * j + 3] = a.qw; }
        if (ped_goal) { ped_goal[2 * j] = a.gx; ped_goal[2 * j + 1] = a.gy; }
        if (ped_traj) memcpy(ped_traj + 6 * (size_t)j, a.traj, sizeof(a.traj));
        if (ped_traj_len) ped_traj_len[j] = a.traj_len;
    }
    for (int q = 0; q < nob; q++) {
        if (obs_shape) obs_shape[q] = ob[q].shape;
        if (obs_size) memcpy(obs_size + 4 * (size_t)q, ob[q].size, sizeof(ob[q].size));
        if (obs_pose) { obs_pose[4 * q] = ob[q].x; obs_pose[4 * q + 1] = ob[q].y; obs_pose[4 * q + 2] = ob[q].qz; obs_pose[4 * q + 3] = ob[q].qw; }
    }
    return IMGENV_OK;
}

extern "C" int imgenv_comm_unique_id(void* id128) {
    if (!id128) FAIL(IMGENV_EINVAL, "null argument");
    RcclApi* a = rccl_api();
    if (!a) FAIL(IMGENV_EDEVICE, "RCCL (librccl.so) is not available");
    static_assert(sizeof(ncclUniqueId) == IMGENV_COMM_ID_BYTES, "ncclUniqueId size");
    const ncclResult_t e = a->get_id((ncclUniqueId*)id128);
    if (e != ncclSuccess) FAIL(IMGENV_EDEVICE, "ncclGetUniqueId: %s", a->err(e));
    return IMGENV_OK;
}

extern "C" int imgenv_comm_init(imgenv_t* h, const void* id128, int32_t rank, int32_t n_ranks) {
    if (!h || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) FAIL(IMGENV_EINVAL, "bad argument");
    RcclApi* a = rccl_api();
    if (!a) FAIL(IMGENV_EDEVICE, "RCCL (librccl.so) is not available");
    if (h->RL * n_ranks != h->R || h->r0 != rank * h->RL)
        FAIL(IMGENV_EINVAL, "native exchange needs equal contiguous shards: rank %d of %d owns [%d,%d) of %d robots", rank,
             n_ranks, h->r0, h->r1, h->R);
    HIPCHK(hipSetDevice(h->cfg.device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const ncclResult_t e = a->init_rank(&h->comm, n_ranks, id, rank);
    if (e != ncclSuccess) {
        h->comm = nullptr;
        FAIL(IMGENV_EDEVICE, "ncclCommInitRank: %s", a->err(e));
    }
    h->comm_ranks = n_ranks;
    return IMGENV_OK;
}

extern "C" int imgenv_comm_info(imgenv_t* h, int32_t* n_ranks, int32_t* rank) {
    if (!h) FAIL(IMGENV_EINVAL, "null argument");
    if (!h->comm) FAIL(IMGENV_ESTATE, "the handle has no communicator (imgenv_comm_init was not called)");
    int n = 0, r = 0;
    ncclResult_t e = rccl_api()->count(h->comm, &n);
    if (e == ncclSuccess) e = rccl_api()->user_rank(h->comm, &r);
    if (e != ncclSuccess) FAIL(IMGENV_EDEVICE, "ncclCommCount / ncclCommUserRank: %s", rccl_api()->err(e));
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
    return IMGENV_OK;
}

extern "C" int imgenv_records(imgenv_t* h, double** records, int64_t* bytes_per_robot) {
    if (!h) FAIL(IMGENV_EINVAL, "null argument");
    if (records) *records = h->d.rec;
    if (bytes_per_robot) *bytes_per_robot = IMGENV_RECORD_DOUBLES * (int64_t)sizeof(double);
    return IMGENV_OK;
}

extern "C" int imgenv_outputs(imgenv_t* h, imgenv_out* out) {
    if (!h || !out) FAIL(IMGENV_EINVAL, "null argument");
    // a caller built against an older, shorter imgenv_out says so in struct_size and gets no more than that many bytes
    // (fields are only ever appended); 0 = the caller's struct is this library's
    const int32_t want = out->struct_size;
    if (want < 0 || want > (int32_t)sizeof(imgenv_out)) FAIL(IMGENV_EINVAL, "imgenv_out.struct_size %d (this library's is %d)", want, (int)sizeof(imgenv_out));
    const imgenv_out& src = h->pub_arena ? h->pub_out : h->out;
    if (want == 0) {
        *out = src;
    } else {
        memcpy(out, &src, (size_t)want);
        out->struct_size = want;  // what the caller really holds, not this library's larger size
    }
    return IMGENV_OK;
}

extern "C" int imgenv_cv_resize_u8(int kind, const uint8_t* src, int32_t sh, int32_t sw, uint8_t* dst, int32_t dh, int32_t dw) {
    if (!src || !dst || sh < 1 || sw < 1 || dh < 1 || dw < 1 || (kind != 0 && kind != 1)) FAIL(IMGENV_EINVAL, "bad argument");
    cv_resize_u8(kind == 1, src, sh, sw, dst, dh, dw);
    return IMGENV_OK;
}

extern "C" int imgenv_step_launches(imgenv_t* h) { return h ? h->launches : 0; }
extern "C" int imgenv_layer_mode(imgenv_t* h) {
    if (!h) return -1;
    return (h->stamp ? 1 : h->sum ? 2 : 0) | (h->d.sum_shard ? 4 : 0) | (h->early && h->gates_work ? 8 : 0) | (h->sfm_ahead ? 16 : 0);
}

// debug: read (and clear) the per-phase cycle counters of IMGENV_PHASE_PROFILE builds
extern "C" int imgenv_debug_phases(imgenv_t* h, unsigned long long* out16) {
    if (!h || !out16) FAIL(IMGENV_EINVAL, "null argument");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out16, h->d.prof, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->d.prof, 0, 16 * sizeof(unsigned long long)));
    return IMGENV_OK;
}
// debug: the 32 free-form marks of profile builds
extern "C" int imgenv_debug_marks(imgenv_t* h, unsigned long long* out32) {
    if (!h || !out32) FAIL(IMGENV_EINVAL, "null argument");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out32, h->d.dbg, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return IMGENV_OK;
}
// debug / tests: the quadtree of world `world`'s social-force crowd as a digest that does not depend on how the nodes are numbered --
// node count, member entries, a sum of per-leaf hashes (rectangle + sorted members), a sum of per-agent hashes (the rectangle the
// agent's treehash entry points at).  tests/test_gpu_parity.py holds it to the oracle's tree (oracle_sfm.c: sfm_tree_digest) step by step.
extern "C" int imgenv_debug_sfm_tree(imgenv_t* h, int32_t world, uint64_t* out8 /* [8]: node count, member entries, leaf-hash sum, agent-hash sum, 256 membership bits */) {
    if (!h || !out8) FAIL(IMGENV_EINVAL, "null argument");
    const SfmDev& f = h->d.sfm;
    if (h->cfg.ped_scene_type != IMGENV_SCENE_PEDSIM || f.n <= 0 || world < 0 || world >= f.W) FAIL(IMGENV_EINVAL, "no social-force crowd %d", world);
    HIPCHK(hipDeviceSynchronize());
    int n_nodes = 0;
    HIPCHK(hipMemcpy(&n_nodes, f.n_nodes + world, sizeof(int), hipMemcpyDeviceToHost));
    if (n_nodes < 1 || n_nodes > f.cap_nodes) FAIL(IMGENV_EDEVICE, "quadtree node count %d", n_nodes);
    std::vector<SfmNode> nodes((size_t)n_nodes);
    std::vector<int> hash((size_t)f.n);
    HIPCHK(hipMemcpy(nodes.data(), f.nodes + (size_t)world * f.cap_nodes, sizeof(SfmNode) * (size_t)n_nodes, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hash.data(), f.treehash + (size_t)world * f.n, sizeof(int) * (size_t)f.n, hipMemcpyDeviceToHost));
    auto mix = [](uint64_t a, uint64_t v) { return (a ^ v) * 0x100000001b3ull; };
    auto bits = [](double v) {
        uint64_t b;
        memcpy(&b, &v, sizeof(b));
        return b;
    };
    auto rect = [&](uint64_t a, const SfmNode& q) { return mix(mix(mix(mix(a, bits(q.x)), bits(q.y)), bits(q.w)), bits(q.h)); };
    uint64_t members = 0, leaves = 0, agents = 0;
    out8[4] = out8[5] = out8[6] = out8[7] = 0;  // one bit per agent that is a member of some leaf
    for (const SfmNode& q : nodes) {
        if (!q.isleaf || q.n_agents == 0) continue;
        uint64_t a = mix(rect(0xcbf29ce484222325ull, q), (uint64_t)q.n_agents);
        for (int e = 0; e < q.n_agents; e++) {
            a = mix(a, (uint64_t)q.agents[e]);
            if (q.agents[e] >= 0 && q.agents[e] < 256) out8[4 + (q.agents[e] >> 6)] |= 1ull << (q.agents[e] & 63);
        }
        leaves += a;
        members += (uint64_t)q.n_agents;
    }
    for (int a = 0; a < f.n; a++) {
        if (hash[a] < 0 || hash[a] >= n_nodes) FAIL(IMGENV_EDEVICE, "treehash[%d] = %d", a, hash[a]);
        agents += rect(mix(0xcbf29ce484222325ull, (uint64_t)a), nodes[(size_t)hash[a]]);
    }
    out8[0] = (uint64_t)n_nodes; out8[1] = members; out8[2] = leaves; out8[3] = agents;
    return IMGENV_OK;
}
// debug: per-wave (start, end, hw id, 0) records of the last k_view [0, 4 RL) and k_obs [4 RL, 8 RL) launches
extern "C" int imgenv_debug_waves(imgenv_t* h, unsigned long long* out) {
    if (!h || !out) FAIL(IMGENV_EINVAL, "null argument");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, h->d.prof + 16, 12 * (size_t)h->RL * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return IMGENV_OK;
}
