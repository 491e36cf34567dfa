// kernels.h -- the HIP kernels of one img_env step, one kernel per stage, gfx950 (wave64).
//
//   k_integrate    8 lanes / robot       Agent::cmd (agent.cpp:186-283); + 1 thread / pedestrian: Agent::update,
//                                        write-back, leg gait (img_env.cpp:344-358) or the recorded trajectory (361-386)
//   k_sfm          1 workgroup / world   libpedsim crowd step (pedscene only)
//   k_raster       1 wave / ped + robot  view_ped + the shared robot-owner layers (img_env.cpp:594-629)
//   k_compose      1 thread / 4 cells    composed layer = obstacles + peds + robot owners (k_compose_tiles: only the 8 x 8 tiles
//                                        the rasters touched, for big or many maps)
//   k_view         1 wave / robot        Agent::view: collision, crop, laser, stamp (agent.cpp:356-509)
//   k_obs<E>       1 wave / robot        PedInfo, sorted ped vector, ped_map (img_env.cpp:568-584, yaml_env.py:392-456)
//   k_side_robots  1 thread / robot      RVO robot records + Agent::get_state (agent.cpp:156-184), on a side stream
//   k_orca         1 wave / pedestrian   waypoint logic + ORCA half-planes + LP (img_env.cpp:304-343), on a side stream
//   tail_group     1 lane / robot        step_ds, reward / done wrappers (yaml_env.py:446-481, base.py:153-254): no launch of its
//                                        own, run inside k_view / k_obs by the wavefront that completes a group of 64 robots
//   k_state        1 thread / robot      Agent::get_state after a reset of worlds without pedestrians (no side streams there;
//                                        in their steps k_integrate does it)
// A handle may hold several independent worlds (DevWorld::W): robots and pedestrians are numbered world-major, every grid
// layer exists once per world, and a launch covers everything or the worlds listed in DevWorld::act_list (a reset).
//
// This is gather / raster / scan work on bytes and small integers: no MFMA.  What matters is
// wave-per-robot decomposition, LDS staging of the 48x48 windows and pedestrian lists, coalesced
// dword stores of the outputs, no contended atomics, and all 8192 wavefronts of a launch resident at once.
#pragma once
#include <hip/hip_runtime.h>

#include "orca_device.h"
#include "world.h"
#include "exp_hooks.h"

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// robot class record: by value from the kernel arguments when possible (wave-uniform index)
// a double that is known to be the same in every lane, moved to scalar registers
__device__ __forceinline__ double uniform_f64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---- region of the grid a robot-sharded rank keeps up to date: bounding box of its robots +- view reach ----
__device__ __forceinline__ uint32_t ordered_u32(float f) {  // monotonic float -> uint32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_f32(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
#define BBOX_INIT_MIN 0xFFFFFFFFu
#define BBOX_INIT_MAX 0u
struct Region {
    int m0, m1, n0, n1;  // cells [m0, m1) x [n0, n1)
};
__device__ __forceinline__ Region grid_region(const DevWorld& w) {
    Region g = {0, w.Hg, 0, w.Wg};
    if (w.sharded) {
        const uint32_t u0 = w.bbox[0], u1 = w.bbox[1], u2 = w.bbox[2], u3 = w.bbox[3];
        if (u0 == BBOX_INIT_MIN) {  // no local robot
            g.m1 = g.n1 = 0;
        } else {
            const double inv = 1.0 / w.res;
            g.m0 = max(0, (int)floor((double)ordered_f32(u0) * inv) - w.region_margin);
            g.n0 = max(0, (int)floor((double)ordered_f32(u1) * inv) - w.region_margin);
            g.m1 = min(w.Hg, (int)ceil((double)ordered_f32(u2) * inv) + w.region_margin + 1);
            g.n1 = min(w.Wg, (int)ceil((double)ordered_f32(u3) * inv) + w.region_margin + 1);
        }
    }
    return g;
}
// one lane per robot contributes (valid = false: nothing); one atomic quadruple per wavefront
__device__ __forceinline__ void bbox_accumulate(const DevWorld& w, bool valid, double x, double y) {
    uint32_t lo_x = valid ? ordered_u32((float)x) : BBOX_INIT_MIN, lo_y = valid ? ordered_u32((float)y) : BBOX_INIT_MIN;
    uint32_t hi_x = valid ? ordered_u32((float)x) : BBOX_INIT_MAX, hi_y = valid ? ordered_u32((float)y) : BBOX_INIT_MAX;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo_x = min(lo_x, (uint32_t)__shfl_xor((int)lo_x, off));
        lo_y = min(lo_y, (uint32_t)__shfl_xor((int)lo_y, off));
        hi_x = max(hi_x, (uint32_t)__shfl_xor((int)hi_x, off));
        hi_y = max(hi_y, (uint32_t)__shfl_xor((int)hi_y, off));
    }
    if (lane_id() == 0 && lo_x != BBOX_INIT_MIN) {
        // (only a wavefront that widens the box as it reads it -- the box only ever grows during a launch, so a stale read costs an
        // atomic that changes nothing: four atomics per wavefront on four words were 4096 of them queued up one behind the other)
        if (lo_x < __hip_atomic_load(&w.bbox[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&w.bbox[0], lo_x);
        if (lo_y < __hip_atomic_load(&w.bbox[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&w.bbox[1], lo_y);
        if (hi_x > __hip_atomic_load(&w.bbox[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&w.bbox[2], hi_x);
        if (hi_y > __hip_atomic_load(&w.bbox[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&w.bbox[3], hi_y);
    }
}

// worlds of a multi-world handle (world-major numbering); a single world never divides
__device__ __forceinline__ int world_of_robot(const DevWorld& w, int i) { return w.W > 1 ? i / w.Rw : 0; }
__device__ __forceinline__ int world_of_ped(const DevWorld& w, int j) { return w.W > 1 ? j / w.Pw : 0; }
// ---- STAMP mode (layout in world.h).  The merge is a lattice -- nothing < one robot < several robots < pedestrian -- so the
// final word does not depend on the order in which the rasters of a step arrive. ----
__device__ __forceinline__ bool stamp_is_current(uint32_t v, uint32_t tag) { return ((v >> STAMP_TAG_SHIFT) & 0xFFu) == tag; }
// view_robot (img_env.cpp:620-629): robot i covers the cell
// seg_tag: the tag again, once per 64 consecutive cells of the layer: where it is not this step's, no cell of the segment
// carries a stamp of this step, and the obstacle map alone says what a view sees there (k_crop_big reads a byte per cell
// instead of this layer's word)
// crop_map (world.h): the byte of world-local cell index cl, and the mark a stamp leaves there (every stamp of a step writes
// the same byte: plain stores)
__device__ __forceinline__ uint32_t crop_tiled(const DevWorld& w, uint32_t m, uint32_t n) {
    return (((m >> 3) * w.crop_wt + (n >> 3)) << 6) | ((m & 7u) << 3) | (n & 7u);
}
__device__ __forceinline__ uint32_t crop_tag_of(uint32_t tag) { return tag % 127u + 1u; }
__device__ __forceinline__ void crop_mark(const DevWorld& w, size_t c, int world, uint32_t base, uint32_t tag) {
    if (!w.crop_map) return;
    const uint32_t cl = (uint32_t)(c - (size_t)world * w.Gs), m = (uint32_t)(((unsigned long long)cl * w.crop_magic) >> 40), n = cl - m * (uint32_t)w.Wg;
    w.crop_map[(size_t)world * w.crop_ws + crop_tiled(w, m, n)] = (uint8_t)((base == CLS_HIGH ? 128u : 0u) | crop_tag_of(tag));
}
__device__ __forceinline__ void stamp_robot(const DevWorld& w, size_t c, uint32_t i, uint32_t tag, int world) {
    uint32_t* cell = w.cell + c;
    w.seg_tag[c >> 6] = (uint8_t)tag;
    uint32_t old = *cell;
    for (;;) {
        const uint32_t kind = stamp_is_current(old, tag) ? (old >> STAMP_KIND_SHIFT) & 3u : 0u;
        uint32_t nw;
        if (kind == 0u) nw = (old & 7u) | (STAMP_ONE << STAMP_KIND_SHIFT) | (tag << STAMP_TAG_SHIFT) | (i << STAMP_OWNER_SHIFT);
        else if (kind == STAMP_ONE && (old >> STAMP_OWNER_SHIFT) != i) nw = (old & 7u) | (STAMP_MANY << STAMP_KIND_SHIFT) | (tag << STAMP_TAG_SHIFT);
        else return;  // this robot already, several already, or a pedestrian (whoever stamped first this step marked crop_map)
        const uint32_t seen = atomicCAS(cell, old, nw);
        if (seen == old) {
            if (kind == 0u) crop_mark(w, c, world, old & 7u, tag);
            return;
        }
        old = seen;
    }
}
// view_ped (img_env.cpp:594-618): a pedestrian sample lands on the cell
__device__ __forceinline__ void stamp_ped(const DevWorld& w, size_t c, uint32_t old /* the word as just read */, uint32_t tag, int world) {
    uint32_t* cell = w.cell + c;
    w.seg_tag[c >> 6] = (uint8_t)tag;
    for (;;) {
        if (stamp_is_current(old, tag) && ((old >> STAMP_KIND_SHIFT) & 3u) == STAMP_PED) return;
        const uint32_t seen = atomicCAS(cell, old, (old & 7u) | (STAMP_PED << STAMP_KIND_SHIFT) | (tag << STAMP_TAG_SHIFT));
        if (seen == old) {
            if (!stamp_is_current(old, tag)) crop_mark(w, c, world, old & 7u, tag);
            return;
        }
        old = seen;
    }
}
// The same two merges for SEVERAL cells of one lane at once: the words are read together, the first compare-and-swaps go out
// together, and only a lane that lost a race (another raster stamped the same cell in between) falls back to the loops above.
// A raster that stamps its cells one after the other pays two memory round trips per cell -- in a reset of a few worlds, or
// with a few hundred robots in a launch, that chain of round trips IS the kernel's duration.
template <int U>
__device__ __forceinline__ void stamp_robot_batch(const DevWorld& w, const uint32_t (&c)[U], const bool (&go)[U], uint32_t i, uint32_t tag, int world) {
    uint32_t old[U], nw[U], seen[U];
    bool cas[U];
#pragma unroll
    for (int u = 0; u < U; u++) old[u] = go[u] ? w.cell[c[u]] : 0u;
#pragma unroll
    for (int u = 0; u < U; u++) {
        cas[u] = false;
        seen[u] = 0u;
        nw[u] = 0u;
        if (!__any(go[u])) continue;  // (a round nobody takes part in)
        const uint32_t kind = stamp_is_current(old[u], tag) ? (old[u] >> STAMP_KIND_SHIFT) & 3u : 0u;
        cas[u] = go[u];
        nw[u] = 0;
        if (kind == 0u) nw[u] = (old[u] & 7u) | (STAMP_ONE << STAMP_KIND_SHIFT) | (tag << STAMP_TAG_SHIFT) | (i << STAMP_OWNER_SHIFT);
        else if (kind == STAMP_ONE && (old[u] >> STAMP_OWNER_SHIFT) != i) nw[u] = (old[u] & 7u) | (STAMP_MANY << STAMP_KIND_SHIFT) | (tag << STAMP_TAG_SHIFT);
        else cas[u] = false;  // this robot already, several already, or a pedestrian
        if (go[u]) w.seg_tag[c[u] >> 6] = (uint8_t)tag;
        seen[u] = cas[u] ? atomicCAS(w.cell + c[u], old[u], nw[u]) : old[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (!cas[u]) continue;
        if (seen[u] != old[u]) stamp_robot(w, c[u], i, tag, world);  // lost a race: from the top
        else if (!stamp_is_current(old[u], tag)) crop_mark(w, c[u], world, old[u] & 7u, tag);
    }
}
// (v: the words as just read -- the caller has looked at their base classes already)
template <int U>
__device__ __forceinline__ void stamp_ped_batch(const DevWorld& w, const uint32_t (&c)[U], const bool (&go)[U], const uint32_t (&v)[U], uint32_t tag, int world) {
    uint32_t seen[U];
    bool cas[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        cas[u] = go[u] && !(stamp_is_current(v[u], tag) && ((v[u] >> STAMP_KIND_SHIFT) & 3u) == STAMP_PED);
        if (go[u]) w.seg_tag[c[u] >> 6] = (uint8_t)tag;
        seen[u] = cas[u] ? atomicCAS(w.cell + c[u], v[u], (v[u] & 7u) | (STAMP_PED << STAMP_KIND_SHIFT) | (tag << STAMP_TAG_SHIFT)) : v[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (!cas[u]) continue;
        if (seen[u] != v[u]) stamp_ped(w, c[u], seen[u], tag, world);
        else if (!stamp_is_current(v[u], tag)) crop_mark(w, c[u], world, v[u] & 7u, tag);
    }
}
// value of a cell in robots_[self].global_map_ as a class code 0..4 (img_env.cpp:594-629), from the layer word v: a
// pedestrian reads as 1; another robot (or several) as 2 unless the map already holds 0 / 1 / 2 there (agent.cpp:315-322)
template <bool STAMP>
__device__ __forceinline__ uint32_t cell_seen_class(uint32_t v, uint32_t self, uint32_t tag) {
    if (!STAMP) {
        if ((v & CLS_ROBOT) && (v >> 8) != self) return CLS_TWO;
        return v & 7u;
    }
    const uint32_t base = v & 7u;
    if (!stamp_is_current(v, tag)) return base;
    const uint32_t kind = (v >> STAMP_KIND_SHIFT) & 3u;
    if (kind == STAMP_PED) return CLS_PED;
    if (base >= CLS_LOW && (kind == STAMP_MANY || (kind == STAMP_ONE && (v >> STAMP_OWNER_SHIFT) != self))) return CLS_TWO;
    return base;
}
// ... and in SUM mode (world.h): counts instead of stamps.  self_w: the robot's index within its world
__device__ __forceinline__ uint32_t cell_seen_class_sum(const DevWorld& w, uint32_t v, uint32_t self_w) {
    const uint32_t base = v & 7u;
    if ((v >> 3) & w.sum_pc_mask) return CLS_PED;
    const uint32_t rx = v >> w.sum_rc_shift, mine = 1u | (self_w << (w.sum_id_shift - w.sum_rc_shift));
    if (base >= CLS_LOW && rx != 0u && rx != mine) return CLS_TWO;
    return base;
}
__device__ __forceinline__ uint32_t sum_robot_word(const DevWorld& w, uint32_t self_w) { return (1u << w.sum_rc_shift) + (self_w << w.sum_id_shift); }
// what robot i (world-wide index) puts into the word's index field: its index within its world -- in a robot shard its local
// index + 1, robots of other ranks count with 0 (world.h: sum_shard)
__device__ __forceinline__ uint32_t sum_self_id(const DevWorld& w, int i) {
    return w.sum_shard ? (uint32_t)(i - w.r0 + 1) : (uint32_t)(i - world_of_robot(w, i) * (w.W > 1 ? w.Rw : 0));
}
// SUM mode in a robot shard: robot i of ANOTHER rank now covers the cells `nb` (a bitmap over the box of side x side cells around
// (cm, cn)); take it off the cells it has left since the bitmap this rank holds for it, add it to the ones it has entered.  One lane.
__device__ __forceinline__ void sum_apply_bits(const DevWorld& w, int i, unsigned long long nb, int cm, int cn, int rad, int side, uint32_t cell0) {
    const unsigned long long ob = w.rm_bits[i];
    const int2 oc = w.rm_center[i];
    if (ob == nb && oc.x == cm && oc.y == cn) return;  // (the common case: a robot moves a few centimetres per step, a cell is 25)
    const uint32_t word = 1u << w.sum_rc_shift, rc_mask = (1u << (w.sum_id_shift - w.sum_rc_shift)) - 1u;
    for (unsigned long long t = ob; t != 0ull; t &= t - 1ull) {
        const int b = __ffsll((long long)t) - 1, bm = b / side;
        const int m = oc.x - rad + bm, n = oc.y - rad + (b - bm * side);
        const int dm = m - cm + rad, dn = n - cn + rad;
        const bool still = dm >= 0 && dm < side && dn >= 0 && dn < side && ((nb >> (dm * side + dn)) & 1ull) != 0ull;
        if (!still && m >= 0 && m < w.Hg && n >= 0 && n < w.Wg) atomicAdd(&w.cell[cell0 + (uint32_t)m * (uint32_t)w.Wg + (uint32_t)n], 0u - word);
    }
    for (unsigned long long t = nb; t != 0ull; t &= t - 1ull) {
        const int b = __ffsll((long long)t) - 1, bm = b / side;
        const int m = cm - rad + bm, n = cn - rad + (b - bm * side);
        const int dm = m - oc.x + rad, dn = n - oc.y + rad;
        const bool was = dm >= 0 && dm < side && dn >= 0 && dn < side && ((ob >> (dm * side + dn)) & 1ull) != 0ull;
        if (!was && m >= 0 && m < w.Hg && n >= 0 && n < w.Wg) {
            const uint32_t before = atomicAdd(&w.cell[cell0 + (uint32_t)m * (uint32_t)w.Wg + (uint32_t)n], word);
            if (((before >> w.sum_rc_shift) & rc_mask) == rc_mask) w.err[6] = 9;  // (as in raster_robot)
        }
    }
    w.rm_bits[i] = nb;
    w.rm_center[i] = make_int2(cm, cn);
}
// per-step values travel as kernel arguments
__device__ __forceinline__ int tail_elapsed_of(const DevWorld& w) { return w.tail_elapsed; }
__device__ __forceinline__ uint32_t stamp_tag_of(const DevWorld& w) { return w.stamp_tag; }
// ---- output guard (IMGENV_FLAG_CHECK_OUTPUTS): a position-weighted 64-bit sum per output array
#define OUT_SUM_CHUNK 65536  // bytes one workgroup sums
struct OutSpan {
    const unsigned char* p;  // 256-byte aligned (arena carve-outs)
    unsigned long long bytes;
    int first_block, pad_;
};
__global__ __launch_bounds__(256) void k_out_sum(const OutSpan* spans, int n, unsigned long long* sums) {
    int f = 0;
    while (f + 1 < n && spans[f + 1].first_block <= (int)blockIdx.x) f++;  // (at most 32 spans)
    const OutSpan sp = spans[f];
    const unsigned long long lo = (unsigned long long)((int)blockIdx.x - sp.first_block) * OUT_SUM_CHUNK;
    const unsigned long long hi = lo + OUT_SUM_CHUNK < sp.bytes ? lo + OUT_SUM_CHUNK : sp.bytes;
    unsigned long long acc = 0;
    for (unsigned long long b = lo + 4ull * threadIdx.x; b < hi; b += 4ull * 256) {
        uint32_t v = 0;
        if (b + 4 <= hi) {
            v = *(const uint32_t*)(sp.p + b);
        } else {
            for (unsigned long long q = b; q < hi; q++) v |= (uint32_t)sp.p[q] << (8 * (q - b));
        }
        acc += ((unsigned long long)v + 0x9E3779B97F4A7C15ull) * (2ull * (b >> 2) + 1ull);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane_id() == 0 && acc) atomicAdd(&sums[f], acc);
}
__global__ void k_out_verify(const unsigned long long* sealed, const unsigned long long* found, int n, int* err) {
    const int f = threadIdx.x;
    if (f < n && sealed[f] != found[f]) err[5] = f + 1;  // (page-locked host memory: a plain store; any of the changed arrays names the problem)
}

// robots / pedestrians of a launch: the host's count, or (device-side auto-reset) what k_finished_dev counted
__device__ __forceinline__ int act_count_l(const DevWorld& w) { return w.act_n_dev ? *w.act_n_dev * w.Rw : w.act_nl; }
__device__ __forceinline__ int act_count_g(const DevWorld& w) { return w.act_n_dev ? *w.act_n_dev * w.Rw : w.act_ng; }
__device__ __forceinline__ int act_count_p(const DevWorld& w) { return w.act_n_dev ? *w.act_n_dev * w.Pw : w.act_np; }
// the t-th robot (per_world = Rw) or pedestrian (Pw) of a launch: everything, or the members of the listed worlds
__device__ __forceinline__ int act_member(const DevWorld& w, int per_world, int t) {
    if (!w.act_list) return t;
    const int q = t / per_world;
    return w.act_list[q] * per_world + (t - q * per_world);
}

__device__ __forceinline__ RobotClassDev robot_class(const DevWorld& w, int cls) {
    return w.rc[cls];
}

// The per-robot scalars of a step (tail_group, what k_tail used to be) have no launch of their own.  k_view and k_obs each make
// ONE 64-bit exchange on their robot's word -- k_view brings the collision code, k_obs the pedestrian distance, all the tail
// needs from them, so no store has to be waited for -- and whichever comes second counts the robot in its group of 64; the
// wavefront that completes a group runs the group's tails, one lane per robot.  The exchange and the 64-way counter cost the
// step about 1 us; the launch + join bubble they replace, 10.  k_view arrives right after its collision phase (no store in
// flight yet), k_obs at its end.
// Call tail_arrive_* with all 64 lanes of the first (or only) wavefront of the workgroup at active-list position t.
#define TAIL_CNT_STRIDE 32  // one group counter per 128-byte line: 32 of them in one line queue up behind each other like one (k_view 62 -> 88 us)
__device__ __forceinline__ void tail_group(const DevWorld& w, int g);
__device__ __forceinline__ void tail_count(const DevWorld& w, int t, bool later) {
    int run = 0;
    if (lane_id() == 0 && later) {
        const int g = t >> 6, members = min(WAVE, act_count_l(w) - (g << 6));
        run = atomicAdd(&w.tail_cnt[g * TAIL_CNT_STRIDE], 1) + 1 == members;
    }
    if (__builtin_amdgcn_readfirstlane(run)) tail_group(w, t >> 6);
}
__device__ __forceinline__ void tail_arrive_view(const DevWorld& w, int t, int l, int code) {
    bool later = false;
    if (lane_id() == 0) {
        const unsigned long long prev = atomicOr(&w.tail_sig[l], ((unsigned long long)(uint32_t)code << 8) | 2ull);
        later = !w.tail_fused || (prev & 1ull) != 0ull;  // no pedestrians, no k_obs: the view is all there is
    }
    tail_count(w, t, later);
}
__device__ __forceinline__ void tail_arrive_obs(const DevWorld& w, int t, int l, double min_dist) {  // min_dist holds a float32 value
    bool later = false;
    if (lane_id() == 0) {
        const unsigned long long prev = atomicOr(&w.tail_sig[l], ((unsigned long long)__float_as_uint((float)min_dist) << 32) | 1ull);
        later = (prev & 2ull) != 0ull;
    }
    tail_count(w, t, later);
}

// Optional per-phase cycle accounting (build with -DIMGENV_PHASE_PROFILE): lane 0 of every wave adds
// the shader-clock cycles of each phase to w.prof[slot]; tools/phase_profile.py prints the split.
#if defined(IMGENV_PHASE_PROFILE) || defined(IMGENV_WAVE_TIMELINE)
// (IMGENV_WAVE_TIMELINE alone: the phase marks record when THIS wavefront passed them, 10 ns ticks since its start, 16 bits each,
// in the upper halves of the two hardware-id words: marks 0..3 of the kernel in order)
#define WAVE_T0() const unsigned long long wave_t0_ = wall_clock64(); unsigned long long wave_ph_ = 0; int wave_nph_ = 0; (void)wave_ph_; (void)wave_nph_
#define WAVE_DONE(base)                                                                   \
    do {                                                                                  \
        if (lane_id() == 0) {                                                             \
            unsigned long long* p_ = w.prof + 16 + (size_t)(base) * 4 * w.RL + 4 * (size_t)blockIdx.x; \
            p_[0] = wave_t0_;                                                             \
            p_[1] = wall_clock64();                                                       \
            p_[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((wave_ph_ & 0xFFFFFFFFull) << 32);  \
            p_[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) | (wave_ph_ & 0xFFFFFFFF00000000ull); \
        }                                                                                 \
    } while (0)
#else
#define WAVE_T0() (void)0
#define WAVE_DONE(base) (void)0
#endif
#ifdef IMGENV_PHASE_PROFILE
#define PHASE_BEGIN() long long ph_t_ = clock64(); WAVE_T0()
#define PHASE_MARK(slot)                                                                  \
    do {                                                                                  \
        const long long now_ = clock64();                                                 \
        if (lane_id() == 0) atomicAdd((unsigned long long*)&w.prof[slot], (unsigned long long)(now_ - ph_t_)); \
        ph_t_ = now_;                                                                     \
    } while (0)
#elif defined(IMGENV_WAVE_TIMELINE)
#define PHASE_BEGIN() WAVE_T0()
#define PHASE_MARK(slot)                                                                  \
    do {                                                                                  \
        if (wave_nph_ < 4) wave_ph_ |= (unsigned long long)min(wall_clock64() - wave_t0_, 0xFFFFull) << (16 * wave_nph_); \
        wave_nph_++;                                                                      \
    } while (0)
#else
#define PHASE_BEGIN() WAVE_T0()
#define PHASE_MARK(slot) (void)0
#endif
// k_view's issue priority by phase (world.h: view_prio).  All 8192 wavefronts of a headline launch start within 2 us and the SIMD's
// arbiter serves the oldest first: the eight wavefronts of a SIMD drift apart -- the first is done after 33 us, the last after 49
// (tools/wave_timeline.py) -- and for the last third of the kernel the SIMDs run at falling occupancy, where a wavefront is bound
// by its own round trips.  A wavefront that is BEHIND (in an earlier phase) asks for the issue slot first: collision + crop 3,
// first hits 2, final pass 1, the resolve 0.  Same work, 50.6 -> 46 us (round 6: 94.7 -> 90.2 us per headline step).
// (Not in the four-wavefront variant: its launches -- cfg-5, LDS-bound views -- run beside a long k_obs and never set view_prio;
// the dead branches alone cost cfg-5 2 % in registers spilled elsewhere.)
#define VIEW_PRIO(level)                                                    \
    do {                                                                    \
        if constexpr (NW != 4) {                                            \
            if (w.view_prio) __builtin_amdgcn_s_setprio(level);             \
        }                                                                   \
    } while (0)
// -DIMGENV_PHASE_PROFILE -DIMGENV_PROFILE_RASTER: the marks inside k_raster's robot blocks instead of k_orca's (they share slots)
#if defined(IMGENV_PHASE_PROFILE) && defined(IMGENV_PROFILE_RASTER)
#define RASTER_MARK_BEGIN() long long rm_t_ = clock64()
#define RASTER_MARK(slot)                                                                 \
    do {                                                                                  \
        const long long now_ = clock64();                                                 \
        if (threadIdx.x == 0) atomicAdd((unsigned long long*)&w.prof[slot], (unsigned long long)(now_ - rm_t_)); \
        rm_t_ = now_;                                                                     \
    } while (0)
#define ORCA_MARK(slot) (void)0
#else
#define RASTER_MARK_BEGIN() (void)0
#define RASTER_MARK(slot) (void)0
#define ORCA_MARK(slot) PHASE_MARK(slot)
#endif

// ------------------------------------------------------------------------------------------------
// Pedestrian advance (ORCA)

// PedAgent::arrive + _get_cur_goal + RVOScene::step pref velocity (img_env.cpp:306-319, rvoscene.h:36-46),
// then Agent::computeNeighbors + computeNewVelocity -- for a GROUP of up to 4 pedestrians of one world per wavefront
// (workgroup = (listed world, group)): agent g of the group owns ROW g of the wavefront, lanes 16 g .. 16 g + 15; lane 16 g is
// its home lane.
//   * the world's obstacle segments + BSP nodes are copied into LDS once (a pedestrian's tree walk and half-plane construction
//     are chains of dependent loads: ~100 HBM round trips per agent before, LDS accesses now);
//   * agent neighbours: every candidate is loaded once and tested against each agent of the group (positions and ranges in
//     scalar registers), hits inserted in index order by the agent's home lane;
//   * obstacle neighbours (KdTree::queryObstacleTreeRecursive): the row evaluates every tree node at once (which side the
//     agent is on, whether the far side gets visited, whether the node's segment is inserted and at what distance); the home
//     lane then only replays the traversal on those flags to get the visiting order, and the row sorts the inserted
//     segments by (distance, visiting order) by rank -- what the reference's insertion sort leaves behind;
//   * ORCA lines: one obstacle neighbour per lane (everything but the "already covered" test is independent of the other
//     lines: Agent.cpp:479-671), coverage of i by the candidate lines j < i as a 16 x 16 bit matrix, the accepted lines
//     settled in order on those bits; one agent neighbour per lane;
//   * the linear programs stay on the home lane (sequential and tiny).
// (History: one pedestrian per wavefront with everything on lane 0 out of HBM took 129 us at 2048 shipped envs x 4 pedestrians,
// one lane per agent out of LDS 62 us alone -- a single agent's chain of ~9000 dependent instructions.)
__device__ __forceinline__ void state_robot(const DevWorld& w, int l);  // Agent::get_state, below
struct OrcaLaunch {
    int G;          // agents per wavefront (1, 2 or 4: one row of 16 lanes each)
    int groups;     // wavefronts per world
    int cap_on;     // obstacle neighbours an agent's scratch holds (the handle's largest obstacle table, at most ORCA_MAX_ON)
    int cap_stack;  // tree levels its walk may stack up
    int stage_obst; // obstacle segments / nodes the LDS staging area holds (0: read them from HBM, everything on the home lane)
    int fold_side;  // handles of several worlds: this kernel also does k_side_robots' part for its world (a launch less per phase):
                    // the world's robot agents out of the robot records (setRobotPos, img_env.cpp:411-417), Agent::get_state of
                    // its robots (group 0), and the robots taken as neighbour candidates directly instead of through near lists
    int zero_vel;   // (with fold_side) a reset: robot agents start at rest
};
#define ORCA_GROUP_MAX 4
#define ORCA_ROW 16
__host__ __device__ inline size_t orca_row_bytes(const OrcaLaunch& L) {  // an agent's scratch + per-node records + the row's exchange words
    static_assert(ORCA_ROW * 4 % 16 == 0 && sizeof(RvoObstDev) % 4 == 0, "k_orca's LDS carving keeps its uint4 records 16-byte aligned");
    return orca_scratch_bytes(L.cap_on, L.cap_stack) + (size_t)L.stage_obst * 16 + ORCA_ROW * 4;
}
__host__ __device__ inline size_t orca_lds_bytes(const OrcaLaunch& L) {
    return ((size_t)L.G * orca_row_bytes(L) + (size_t)L.stage_obst * (sizeof(RvoObstDev) + sizeof(RvoNodeDev)) + ORCA_NEAR_CAP * sizeof(int) + 15) &
           ~(size_t)15;
}
__global__ __launch_bounds__(WAVE) void k_orca(DevWorld w, OrcaLaunch L) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id(), row = lane >> 4, li = lane & (ORCA_ROW - 1);
    const int n_p = w.W > 1 ? w.Pw : w.P;  // the pedestrians of one world
    const int q = (int)blockIdx.x / L.groups, gi = (int)blockIdx.x - q * L.groups;
    if (q * n_p >= act_count_p(w)) return;
    const int wld = w.W > 1 ? (w.act_list ? w.act_list[q] : q) : 0;
    const int p_lo = wld * n_p, a0 = gi * L.G;
    PHASE_BEGIN();
    const int G = min(L.G, n_p - a0);    // agents of this group
    const bool row_ok = row < G, mine = row_ok && li == 0;
    const int home = min(row, G - 1) * ORCA_ROW;
    const int j = p_lo + a0 + min(row, G - 1);  // the row's agent
    // LDS: per row (scratch | node records | exchange words), then staged obstacles | staged nodes | near_sorted
    const size_t rb = orca_row_bytes(L);
    unsigned char* row_base = smem + (size_t)min(row, L.G - 1) * rb;
    OrcaScratch s;
    orca_scratch_carve(s, row_base, L.cap_on, L.cap_stack);
    uint4* node_rec = (uint4*)(row_base + orca_scratch_bytes(L.cap_on, L.cap_stack));  // flags | obstacle << 8, distance, near child, far child
    uint32_t* xch = (uint32_t*)(node_rec + L.stage_obst);
    RvoObstDev* l_obst = (RvoObstDev*)(smem + (size_t)L.G * rb);
    RvoNodeDev* l_nodes = (RvoNodeDev*)(l_obst + L.stage_obst);
    int* near_sorted = (int*)(l_nodes + L.stage_obst);
    OrcaObst ob;
    ob.obst = w.obst;
    ob.onodes = w.onodes;
    ob.n_obst = w.n_obst;
    ob.oroot = w.oroot;
    if (w.W > 1) {  // this world's slices
        ob.obst += w.obst_base[wld];
        ob.onodes += w.node_base[wld];
        ob.n_obst = w.n_obst_w[wld];
        ob.oroot = w.oroot_w[wld];
    }
    const bool staged = ob.n_obst > 0 && ob.n_obst <= L.stage_obst;
    if (staged) {  // (every obstacle vertex sits in exactly one tree node: as many nodes as vertices)
        const uint32_t* so = (const uint32_t*)ob.obst;
        const uint32_t* sn = (const uint32_t*)ob.onodes;
        uint32_t *dobs = (uint32_t*)l_obst, *dn = (uint32_t*)l_nodes;
        for (int t = lane; t < ob.n_obst * (int)(sizeof(RvoObstDev) / 4); t += WAVE) dobs[t] = so[t];
        for (int t = lane; t < ob.n_obst * (int)(sizeof(RvoNodeDev) / 4); t += WAVE) dn[t] = sn[t];
        ob.obst = l_obst;
        ob.onodes = l_nodes;
    }
    // everything the solve reads of its agent, requested at once (one HBM round trip instead of a dozen along the way)
    const int idx0 = w.ptraj_idx[j], len = w.ptraj_len[j];
    const double ppx = w.ppx[j], ppy = w.ppy[j];
    f2 pos = F2(w.apx[j], w.apy[j]);
    const f2 vel = F2(w.avx[j], w.avy[j]), nv0 = F2(w.anvx[j], w.anvy[j]);
    const float max_speed = w.amax_speed[j];
    const int n_rob_w = w.W > 1 ? w.Rw : w.R, rob_lo_w = w.P + wld * n_rob_w;  // the robot agents of the world
    // fold_side: every robot of the world is a candidate (in index order, as a sorted near list would hold them); more than 16 of
    // them go through the full scan below (a count beyond ORCA_NEAR_CAP asks for it)
    const int n_near_row = w.NA > w.P ? (L.fold_side ? (n_rob_w <= ORCA_ROW ? n_rob_w : ORCA_NEAR_CAP + 1) : w.near_n[j]) : 0;
    const int near_ent = w.NA > w.P ? (L.fold_side ? rob_lo_w + min(li, n_rob_w - 1) : w.near_list[(size_t)j * ORCA_NEAR_CAP + li]) : 0;  // (the row's first 16 entries, whatever the count)
    if (L.fold_side) {  // k_side_robots' part for this world: every group writes the same robot agents (it reads them back below)
        for (int q = lane; q < n_rob_w; q += WAVE) {
            const int i = wld * n_rob_w + q;
            const double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
            if (w.NA > w.P) {  // (relation_ped_robo = 1: the robots are agents of the crowd)
                const int a = w.P + i;
                w.apx[a] = (float)r[0];
                w.apy[a] = (float)r[1];
                w.avx[a] = L.zero_vel ? 0.0f : (float)r[3];
                w.avy[a] = L.zero_vel ? 0.0f : (float)r[4];
            }
            const int l = i - w.r0;
            if (gi == 0 && l >= 0 && l < w.RL) state_robot(w, l);
        }
        __syncthreads();
    }
    // waypoint + pref velocity: the waypoint in question and the one behind it, both in flight
    f2 pref;
    int idx = idx0;
    {
        const double* tr = w.ptraj + (size_t)j * w.traj_cap * 3;
        const int e0 = idx0 < len ? idx0 : 0, e1 = len > 0 ? (idx0 + 1) % len : 0, e2 = len > 0 ? idx0 % len : 0;
        const double t0x = tr[3 * e0], t0y = tr[3 * e0 + 1], t1x = tr[3 * e1], t1y = tr[3 * e1 + 1], t2x = tr[3 * e2], t2y = tr[3 * e2 + 1];
        bool arrived = false;
        if (idx0 < len) arrived = (t0x - ppx) * (t0x - ppx) + (t0y - ppy) * (t0y - ppy) < 0.04;  // an index past the end (UB in the reference) never "arrives"
        if (arrived) idx++;
        const double gx = arrived ? t1x : t2x, gy = arrived ? t1y : t2y;  // trajectory_[idx % len] (agent.cpp:839-843)
        pref = F2((float)gx, (float)gy) - pos;
        if (abs_sq(pref) > 1.0f) pref = normalize(pref);
        if (mine) w.ptraj_idx[j] = idx;
    }
    ORCA_MARK(5);  // staging, the agent's scalars, waypoint
    __syncthreads();
    // agent neighbours (Agent::computeNeighbors over the kd-tree = the maxNeighbors nearest within neighborDist): candidates in
    // agent index order, 64 per round, each tested against every agent of the group; an agent's hits are inserted in index
    // order by its home lane, with the reference's shrinking range -- first every pedestrian of the world ...
    float range_sq = sqr(0.5f);  // neighborDist (rvoscene.h:57)
    {
        // the group's positions and ranges live in scalar registers (v_readlane with a constant lane): a (round, agent) test is
        // two subtractions, the squares, a compare and a ballot -- no shuffles through LDS in the loop
        float gx[ORCA_GROUP_MAX], gy[ORCA_GROUP_MAX], gr[ORCA_GROUP_MAX];
#pragma unroll
        for (int g = 0; g < ORCA_GROUP_MAX; g++) {
            gx[g] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pos.x), g * ORCA_ROW));
            gy[g] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pos.y), g * ORCA_ROW));
            gr[g] = range_sq;
        }
        // the robots near each agent (rows of up to 16 list entries): their positions requested before the pedestrian scan starts
        f2 near_pos = F2(0.0f, 0.0f);
        if (row_ok && li < min(n_near_row, ORCA_ROW)) near_pos = F2(w.apx[near_ent], w.apy[near_ent]);
        constexpr int SCAN_BATCH = 8;  // rounds whose candidates are loaded together: one HBM / L2 round trip per 512 pedestrians
        for (int base0 = 0; base0 < n_p; base0 += WAVE * SCAN_BATCH) {
            f2 cps[SCAN_BATCH];
#pragma unroll
            for (int u = 0; u < SCAN_BATCH; u++) {
                const int a = p_lo + min(base0 + u * WAVE + lane, n_p - 1);
                cps[u] = F2(w.apx[a], w.apy[a]);
            }
#pragma unroll
            for (int u = 0; u < SCAN_BATCH; u++) {
                const int base = base0 + u * WAVE;
                if (base >= n_p) break;  // uniform
                const int t = base + lane, a = p_lo + min(t, n_p - 1);
                const f2 cp = cps[u];
#pragma unroll
                for (int g = 0; g < ORCA_GROUP_MAX; g++) {
                    if (g >= G) break;  // uniform
                    const float dist_sq = abs_sq(F2(gx[g], gy[g]) - cp);
                    const bool cand = t < n_p && t != a0 + g && dist_sq < gr[g];
                    unsigned long long mask = __ballot(cand);
                    if (mask != 0ull) {  // rare
                        while (mask) {
                            const int src = __ffsll((long long)mask) - 1;
                            mask &= mask - 1;
                            const float d = __shfl(dist_sq, src);
                            const int who = __shfl(a, src);
                            if (lane == g * ORCA_ROW) insert_agent_neighbor(s, d, who, range_sq);
                        }
                        gr[g] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(range_sq), g * ORCA_ROW));
                    }
                }
            }
        }
        // ... then the robots k_side_robots found near each agent, in index order.  Up to 16 of them (nearly always): the row sorts
        // its list by rank, the home lane inserts
        {
            int any_fast = (row_ok && n_near_row > 0 && n_near_row <= ORCA_ROW) ? 1 : 0;
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) any_fast |= __shfl_xor(any_fast, off);
            if (any_fast) {  // uniform
                const bool fast = row_ok && n_near_row > 0 && n_near_row <= ORCA_ROW, have = fast && li < n_near_row;
                float* nd = (float*)s.proj;       // (the projection area is free until the lines are built)
                int* ni = (int*)(nd + ORCA_ROW);
                if (fast) xch[li] = have ? (uint32_t)near_ent : 0x7FFFFFFFu;
                __syncthreads();
                if (have) {
                    int rank = 0;
                    for (int e = 0; e < ORCA_ROW; e++) rank += (int)xch[e] < near_ent ? 1 : 0;
                    nd[rank] = abs_sq(pos - near_pos);
                    ni[rank] = near_ent;
                }
                __syncthreads();
                if (mine && fast)
                    for (int e = 0; e < n_near_row; e++) insert_agent_neighbor(s, nd[e], ni[e], range_sq);
                __syncthreads();
            }
        }
    }
    // ... lists of more than 16 robots one agent at a time (sorted by index; the full robot range if the list overflowed)
    if (w.NA > w.P) {
        const int n_near_mine = mine ? n_near_row : 0;
        const int n_rob = n_rob_w, rob_lo = rob_lo_w;
        for (int g = 0; g < G; g++) {
            const int n_near = __shfl(n_near_mine, g * ORCA_ROW);
            if (n_near <= ORCA_ROW) continue;  // (none, or done above)
            const int jg = p_lo + a0 + g;
            const bool listed = n_near <= ORCA_NEAR_CAP;
            if (listed) {
                const int nm = lane < n_near ? w.near_list[(size_t)jg * ORCA_NEAR_CAP + lane] : 0x7FFFFFFF;
                int rank = 0;
                for (int e = 0; e < n_near; e++) rank += __shfl(nm, e) < nm ? 1 : 0;
                __syncthreads();
                if (lane < n_near) near_sorted[rank] = nm;
                __syncthreads();
            }
            const f2 pg = F2(__shfl(pos.x, g * ORCA_ROW), __shfl(pos.y, g * ORCA_ROW));
            const int n_scan = listed ? n_near : n_rob;
            for (int base = 0; base < n_scan; base += WAVE) {
                const int t = base + lane;
                int a = rob_lo;
                if (t < n_scan) a = listed ? near_sorted[t] : rob_lo + t;
                const float dist_sq = abs_sq(pg - F2(w.apx[a], w.apy[a])), rg = __shfl(range_sq, g * ORCA_ROW);
                const bool cand = t < n_scan && dist_sq < rg;
                unsigned long long mask = __ballot(cand);
                while (mask) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const float d = __shfl(dist_sq, src);
                    const int who = __shfl(a, src);
                    if (lane == g * ORCA_ROW) insert_agent_neighbor(s, d, who, range_sq);
                }
            }
        }
        if (mine && !L.fold_side) w.near_n[j] = 0;  // re-armed for the next step's k_side_robots
    }
    __syncthreads();
    ORCA_MARK(6);  // neighbour scans
    if (!staged && ob.n_obst > 0) {  // an obstacle table too large for LDS: the whole solve on the home lane, out of HBM
        if (mine) {
            query_obstacle_tree(ob, w.err, s, pos, sqr(5.0f * max_speed + 0.5f));
            const f2 nv = compute_new_velocity(w, ob, s, j, pref);
            w.anvx[j] = nv.x;
            w.anvy[j] = nv.y;
            if (w.ped_snap_out) w.ped_snap_out[j] = make_float4(pos.x, pos.y, nv.x, nv.y);
            if (w.ped_snap_out2) w.ped_snap_out2[j] = make_float4(pos.x, pos.y, nv.x, nv.y);
        }
        return;
    }
    const int n_an = __shfl(s.n_an, home);
    // ---- obstacle neighbours: every node of the tree at once, then the traversal replayed on the flags
    int n_on = 0;
    if (staged) {
        const float obst_range_sq = sqr(5.0f * max_speed + 0.5f);  // timeHorizonObst*maxSpeed + radius
        const int n_nodes = ob.n_obst;
        for (int k = li; k < n_nodes; k += ORCA_ROW) {
            const RvoNodeDev nd = ob.onodes[k];
            const int o1 = min(max(nd.obstacle, 0), n_nodes - 1);
            const int o2 = min(max(ob.obst[o1].next, 0), n_nodes - 1);
            const float agent_left = left_of(opoint(ob, o1), opoint(ob, o2), pos);
            const float dist_sq_line = sqr(agent_left) / abs_sq(opoint(ob, o2) - opoint(ob, o1));
            uint32_t f = (agent_left >= 0.0f ? 1u : 0u) | (dist_sq_line < obst_range_sq ? 2u : 0u);
            float dseg = 0.0f;
            if ((f & 2u) != 0u && agent_left < 0.0f) {  // insertObstacleNeighbor(node->obstacle): Agent.cpp:813-838
                dseg = dist_sq_point_segment(opoint(ob, o1), opoint(ob, o2), pos);
                if (dseg < obst_range_sq) f |= 4u;
            }
            // the side the agent is on is walked first; the other one only if the node's line is within range (KdTree.cpp:325-349)
            const int near_child = (f & 1u) ? nd.left : nd.right, far_child = (f & 2u) ? ((f & 1u) ? nd.right : nd.left) : -1;
            if (row_ok) node_rec[k] = make_uint4(f | ((uint32_t)o1 << 8), __float_as_uint(dseg), (uint32_t)near_child, (uint32_t)far_child);
        }
        __syncthreads();
        ORCA_MARK(7);  // tree nodes
        // the traversal itself (near side, the node, far side) on those records alone, one LDS read per node: inserted segments in
        // visiting order into the (still unused) projection area
        float* vis_dist = (float*)s.proj;
        int* vis_idx = (int*)(vis_dist + L.cap_on);
        int nv = 0;
        if (mine) {
            int sp = 0, cur = ob.oroot;
            for (;;) {
                bool overflow = false;
                while (cur >= 0) {  // down the near side
                    if (sp >= s.cap_stack) {
                        overflow = true;
                        break;
                    }
                    s.stack[sp++] = cur;
                    cur = (int)node_rec[cur].z;
                }
                if (overflow) {
                    w.err[1] = 1;
                    break;
                }
                if (sp == 0) break;
                const uint4 rec = node_rec[s.stack[--sp]];
                if (rec.x & 4u) {
                    if (nv >= s.cap_on) {
                        w.err[0] = 1;  // more visible obstacle segments than the scratch holds
                    } else {
                        vis_dist[nv] = __uint_as_float(rec.y);
                        vis_idx[nv] = (int)(rec.x >> 8);
                        nv++;
                    }
                }
                cur = (int)rec.w;
            }
        }
        __syncthreads();
        ORCA_MARK(12);  // traversal replay
        n_on = __shfl(nv, home);
        // insertion in visiting order with a strict "<" (Agent.cpp:824-833) = sorted by (distance, visiting order): by rank
        for (int e = li; e < n_on; e += ORCA_ROW) {
            const float d = vis_dist[e];
            int rank = 0;
            for (int k = 0; k < n_on; k++) {
                const float dk = vis_dist[k];
                rank += (dk < d || (dk == d && k < e)) ? 1 : 0;
            }
            if (row_ok) {
                s.on_dist[rank] = d;
                s.on_idx[rank] = vis_idx[e];
            }
        }
        __syncthreads();
    }
    // ---- ORCA lines of the obstacle neighbours, 16 per round and row
    int nl = 0;
    {
        const float inv_tho = 1.0f / ORCA_TIME_HORIZON_OBST;
        int n_on_max = n_on;  // trip counts are wavefront-uniform: the rows run in lockstep, barriers included
#pragma unroll
        for (int off = 16; off < WAVE; off <<= 1) n_on_max = max(n_on_max, __shfl_xor(n_on_max, off));
        OrcaLine* cand = s.proj;
        for (int base = 0; base < n_on_max; base += ORCA_ROW) {
            const int i = base + li;
            const bool valid = row_ok && i < n_on;
            const int o1 = valid ? s.on_idx[i] : 0;
            OrcaLine ln;
            ln.point = ln.direction = F2(0.0f, 0.0f);
            bool has = false;
            f2 rel1s = F2(0.0f, 0.0f), rel2s = F2(0.0f, 0.0f);
            if (valid) {
                const int o2 = ob.obst[o1].next;
                rel1s = inv_tho * (opoint(ob, o1) - pos);
                rel2s = inv_tho * (opoint(ob, o2) - pos);
                bool covered = false;  // ... by a line of an earlier round
                for (int jj = 0; jj < nl; jj++) covered = covered || obstacle_covered_by(rel1s, rel2s, s.lines[jj]);
                if (!covered) has = obstacle_line(ob, pos, vel, o1, ln);
            }
            if (row_ok) cand[li] = ln;  // (rows without an agent may share a row's scratch: they write nothing)
            __syncthreads();
            uint32_t cov = 0;  // bit jj: candidate line jj of this round covers obstacle i
            if (valid)
                for (int jj = 0; jj < li; jj++) cov |= obstacle_covered_by(rel1s, rel2s, cand[jj]) ? (1u << jj) : 0u;
            if (row_ok) xch[li] = (has ? 1u : 0u) | (cov << 1);
            __syncthreads();
            uint32_t acc = 0;  // the lines of this round that really get pushed, settled in order (every lane of the row alike)
            for (int jj = 0; jj < ORCA_ROW; jj++) {
                const uint32_t x = xch[jj];
                if ((x & 1u) != 0u && ((x >> 1) & acc) == 0u) acc |= 1u << jj;
            }
            if (row_ok && ((acc >> li) & 1u) != 0u) s.lines[nl + __popc(acc & ((1u << li) - 1u))] = ln;
            nl += __popc(acc);
            __syncthreads();
        }
    }
    ORCA_MARK(13);  // rank sort + obstacle lines
    const int num_obst_lines = nl;
    // ---- one agent neighbour per lane
    if (row_ok && li < n_an) s.lines[nl + li] = agent_line(w, pos, vel, s.an_idx[li]);
    nl += n_an;
    __syncthreads();
    ORCA_MARK(14);  // agent lines
    if (mine) {
        const f2 nv = solve_velocity(s, max_speed, nv0, nl, num_obst_lines, pref);
        // ERVO's evacuation term (Agent.cpp:63-69, 430-432) is added by k_evac once the step's actions -- and with them the
        // beep sources -- exist; through the reference's Python API there never are any (yaml_env.py:183-200).
        w.anvx[j] = nv.x;
        w.anvy[j] = nv.y;
        if (w.ped_snap_out) w.ped_snap_out[j] = make_float4(pos.x, pos.y, nv.x, nv.y);  // what the next step's early k_obs moves this pedestrian by
        if (w.ped_snap_out2) w.ped_snap_out2[j] = make_float4(pos.x, pos.y, nv.x, nv.y);
    }
    ORCA_MARK(15);  // linear programs
}

// Beep lottery (img_env.cpp:323-342), one workgroup per world: `rand() / double(RAND_MAX) < ped_ca_p` once per robot in
// robot order, then "v_y > 0" of this step's request (the action's beep; 0 for dead robots, yaml_env.py:328-331) makes the
// robot a source at its pose as the previous step left it (_step_ped runs before _step_robot, img_env.cpp:423-424).
//
// rand() is glibc's TYPE_3 additive-feedback generator: r[i] = r[i-3] + r[i-31] mod 2^32, output r[i] >> 1.  The recurrence
// is linear over Z / 2^32, so word k of the next BEEP_T words is a fixed combination (beep_coef[k][0..30], built on the host
// by running the recurrence on unit vectors) of the last 31 words: every thread produces one rand() value per round with 31
// multiply-adds instead of one lane walking a chain of thousands of dependent additions.
__global__ __launch_bounds__(BEEP_T) void k_beep(DevWorld w, const float* __restrict__ actions) {
    __shared__ uint32_t comb[31 + BEEP_T];
    const int wld = blockIdx.x, t = threadIdx.x;
    const int n_rob = w.W > 1 ? w.Rw : w.R, rob_lo = wld * n_rob;
    uint32_t* st = w.beep_state + 31 * wld;
    if (t < 31) comb[t] = st[t];
    __syncthreads();
    const uint32_t* c = w.beep_coef + 31 * t;
    for (int base = 0; base < n_rob; base += BEEP_T) {
        const int n = min(BEEP_T, n_rob - base);
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 31; j++) v += c[j] * comb[j];
        comb[31 + t] = v;
        if (t < n) {
            const int i = rob_lo + base + t, l = i - w.r0;  // (never sharded: l is the robot's row of `actions`)
            const int rnd = (int)(v >> 1);
            const bool lottery = rnd / double(2147483647) < w.ped_ca_p;
            const double beep_radius = w.py_done[l] ? 0.0 : (double)actions[3 * l + 2];
            w.beep_flag[i] = (lottery && beep_radius > 0) ? 1 : 0;
            const double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
            w.beep_xy[i] = make_float2((float)r[0], (float)r[1]);
        }
        __syncthreads();
        const uint32_t keep = t < 31 ? comb[n + t] : 0u;  // the stream has advanced by n words
        __syncthreads();
        if (t < 31) comb[t] = keep;
        __syncthreads();
    }
    if (t < 31) st[t] = comb[t];
}

// addEvacVelocity for every source, after the LP (ervo_ros Agent.cpp:63-69, 430-432): newVelocity_ += normalize(pa - ps) for
// each source within rs (and farther than 1e-4) of the agent, in robot order -- float32 sums, so the order is kept: 64 lanes
// test 64 robots, the (few) hits are added in index order.  One wavefront per pedestrian; the solve itself ran a step ahead
// on the side stream (k_orca), the sources only exist once this step's actions do.  Unclamped, as in the reference.
__global__ __launch_bounds__(WAVE) void k_evac(DevWorld w) {
    const int j = blockIdx.x, lane = lane_id();
    const int wld = world_of_ped(w, j);
    const int n_rob = w.W > 1 ? w.Rw : w.R, rob_lo = wld * n_rob;
    const f2 pa = F2(w.apx[j], w.apy[j]);
    f2 nv = F2(w.anvx[j], w.anvy[j]);
    const float rs = w.beep_r;
    bool changed = false;
    for (int base = 0; base < n_rob; base += WAVE) {
        const int i = rob_lo + base + lane;
        f2 evac = F2(0.0f, 0.0f);
        bool hit = false;
        if (base + lane < n_rob && w.beep_flag[i]) {
            const float2 ps = w.beep_xy[i];
            evac = pa - F2(ps.x, ps.y);
            const float a = vabs(evac);
            hit = !(a > rs || (double)a < 1e-4);
        }
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            nv = nv + normalize(F2(__shfl(evac.x, src), __shfl(evac.y, src)));
            changed = true;
        }
    }
    if (lane == 0 && changed) {
        w.anvx[j] = nv.x;
        w.anvy[j] = nv.y;
    }
}

// PedAgent::update_bbox (agent.cpp:696-735); step_len_ = 0.3 (2-arg constructor, agent.cpp:659-664)
__device__ __forceinline__ void ped_leg_gait(const DevWorld& w, int j, double x, double y, double ox, double oy) {
    // the class differs from lane to lane: read it from the copy of the class records in HBM -- a per-lane index into
    // the by-value kernel argument would make the compiler spill the whole DevWorld to scratch
    const PedClassDev& k = w.pc_mem[w.ped_cls[j]];
    if (k.shape != IMGENV_SHAPE_LEG) return;
    const double step_len = 0.3;
    const double move = sqrt((x - ox) * (x - ox) + (y - oy) * (y - oy));
    const int last = w.pstate[j];
    int st = (int)((move + w.prem[j]) / step_len + last);
    w.prem[j] = move + w.prem[j] - (st - last) * step_len;
    st %= 7;
    w.pstate[j] = st;
    // (the array pointers fetched once, in front of the branches: merged behind them, the stores would pick the FIELD by a phi of
    // addresses inside the kernel argument -- and a kernel whose argument has its address taken keeps all 3.5 KB of it in scratch
    // memory: k_move_raster, 34 -> 200 us)
    double *const llx = w.llx, *const lly = w.lly, *const rlx = w.rlx, *const rly = w.rly;
    if (st == 0 || st == 4) {
        llx[j] = k.sizes[0];
        lly[j] = k.sizes[1];
        rlx[j] = k.sizes[3];
        rly[j] = k.sizes[4];
    } else if (st == 1 || st == 3) {
        llx[j] = -step_len / 2;
        rlx[j] = step_len / 2;
    } else if (st == 2) {
        llx[j] = -step_len;
        rlx[j] = step_len;
    } else if (st == 5) {
        llx[j] = step_len / 2;
        rlx[j] = -step_len / 2;
    } else if (st == 6) {
        llx[j] = step_len;
        rlx[j] = -step_len;
    }
}

// Agent::update (Agent.cpp:840-843), getNewPosAndVel (rvoscene.h:72-82), set_position + update_bbox
// (agent.cpp:691-735)
__device__ __forceinline__ void ped_update_one(const DevWorld& w, int j) {
    const float ts = (float)w.step_hz;
    const float vx = w.anvx[j], vy = w.anvy[j];
    w.avx[j] = vx;
    w.avy[j] = vy;
    const float nx = w.apx[j] + vx * ts, ny = w.apy[j] + vy * ts;
    w.apx[j] = nx;
    w.apy[j] = ny;
    const double ox = w.ppx[j], oy = w.ppy[j];
    w.plx[j] = ox;
    w.ply[j] = oy;
    const double x = (double)nx, y = (double)ny;
    w.ppx[j] = x;
    w.ppy[j] = y;
    w.pyaw[j] = 0.0;  // uninitialised local `yaw` in the reference (img_env.cpp:346-349)
    w.pvx[j] = (double)vx;
    w.pvy[j] = (double)vy;
    w.ped_state[4 * j] = x;
    w.ped_state[4 * j + 1] = y;
    w.ped_state[4 * j + 2] = (double)vx;
    w.ped_state[4 * j + 3] = (double)vy;
    ped_leg_gait(w, j, x, y, ox, oy);
}

// PedScene::step (pedscene.h:48-50) = Tscene::moveAgents(step_hz) for the whole social-force crowd, then the
// write-back of img_env.cpp:344-358 (getNewPosAndVel pedscene.h:82-91, set_position, update_bbox)
// the write-back of img_env.cpp:344-358 for pedestrian jw of crowd `world`: from the crowd's (new) positions and velocities
__device__ __forceinline__ void sfm_publish_one(const DevWorld& w, int world, int jw, const double* p, const double* v, int n_peds) {
    const int j = world * n_peds + jw;  // the pedestrian's index in the handle
    const double ox = w.ppx[j], oy = w.ppy[j];
    const double x = p[3 * jw], y = p[3 * jw + 1];
    const double vx = v[3 * jw], vy = v[3 * jw + 1];
    w.plx[j] = ox;
    w.ply[j] = oy;
    w.ppx[j] = x;
    w.ppy[j] = y;
    w.pyaw[j] = 0.0;  // uninitialised local `yaw` in the reference (img_env.cpp:346-349)
    w.pvx[j] = vx;
    w.pvy[j] = vy;
    w.ped_state[4 * j] = x;
    w.ped_state[4 * j + 1] = y;
    w.ped_state[4 * j + 2] = vx;
    w.ped_state[4 * j + 3] = vy;
    ped_leg_gait(w, j, x, y, ox, oy);
}
// ... as a launch of its own, for a crowd whose step ran ahead (SfmDev: *_out; the host has swapped the two sets by now)
__global__ __launch_bounds__(SFM_MAX_AGENTS) void k_sfm_publish(DevWorld w) {
    const int world = blockIdx.x, jw = threadIdx.x;
    const SfmDev f = sfm_of_world(w.sfm, world);
    if (jw < f.n_peds) sfm_publish_one(w, world, jw, f.p, f.v, f.n_peds);
}
__global__ __launch_bounds__(SFM_MAX_AGENTS) void k_sfm(DevWorld w, int phase, int publish) {
    __shared__ uint32_t nb_bits[SFM_MAX_AGENTS * (SFM_MAX_AGENTS / 32)];
    __shared__ double sfm_sh[4 * SFM_MAX_AGENTS];
    __shared__ __attribute__((aligned(16))) unsigned short sfm_stk[SFM_WALK_CAP * SFM_MAX_AGENTS];  // (its last row doubles as eight 32-bit words: sfm_step)
    extern __shared__ __attribute__((aligned(16))) unsigned char sfm_dyn[];  // [SFM_LDS_NODES] nodes (phases 0, 1, 3; none in phase 2)
    SfmNode* sfm_nodes = (SfmNode*)sfm_dyn;
    __shared__ int sfm_hash[SFM_MAX_AGENTS];
    __shared__ int sfm_nn;
    // one crowd per world: phases 0, 1, 3 run one workgroup per world, phase 2 (the pair terms) pblocks of them
    const int pblocks = phase == 2 ? (int)gridDim.x / w.sfm.W : 1;
    const int world = (int)blockIdx.x / pblocks, pblock = (int)blockIdx.x - world * pblocks;
    const SfmDev f = sfm_of_world(w.sfm, world);
#ifdef IMGENV_PHASE_PROFILE
    sfm_step(f, w.step_hz, phase, nb_bits, sfm_sh, sfm_stk, sfm_nodes, sfm_hash, &sfm_nn, pblock, pblocks, w.dbg);
#else
    sfm_step(f, w.step_hz, phase, nb_bits, sfm_sh, sfm_stk, sfm_nodes, sfm_hash, &sfm_nn, pblock, pblocks);
#endif
    if (phase == 1 || phase == 2 || !publish) return;
    const int jw = threadIdx.x;
    if (jw < f.n_peds) sfm_publish_one(w, world, jw, f.p_out, f.v_out, f.n_peds);
}

// ------------------------------------------------------------------------------------------------
// Pose integrate

__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(lo, x), hi); }
__device__ __forceinline__ int signd(double x) { return x == 0 ? 0 : (int)(x / fabs(x)); }

// SpeedLimiter::limit (speed_limit.cpp:92-173); max_jerk = msg.min_jerk, min_jerk = 0 (speed_limit.cpp:56-65)
__device__ void limiter_limit(bool has_v, bool has_a, bool has_j, double min_v, double max_v, double min_a, double max_a,
                              double min_j, double max_j, double& v, double v0, double v1, double dt) {
    if (has_j) {
        const double dv = v - v0, dv0 = v0 - v1;
        const double dt2 = 2. * dt * dt;
        const double da = clampd(dv - dv0, min_j * dt2, max_j * dt2);
        v = v0 + dv0 + da;
    }
    if (has_a) {
        const double tmp = v;
        const int v_sign = signd(v), v0_sign = signd(v0);
        if (v_sign + v0_sign != 0) {
            const double dv_min = min_a * dt, dv_max = max_a * dt;
            double dv = v - v0;
            const int dv_sign = signd(dv);
            if (dv_sign == v0_sign || dv_sign == v_sign)
                dv = dv_sign * clampd(fabs(dv), dv_min, dv_max);
            else
                dv = dv_sign * fabs(clampd(-fabs(dv), dv_min, dv_max));
            v = v0 + dv;
        } else {
            const double zero_dt = fabs(v0 / min_a);
            if (zero_dt >= dt)
                v = v0_sign * (fabs(v0) - fabs(min_a) * dt);
            else {
                const double v_dt = fabs(v / max_a);
                if (zero_dt + v_dt >= dt)
                    v = v_sign * fabs(max_a * (dt - zero_dt));
                else
                    v = tmp;
            }
        }
    }
    if (has_v) v = clampd(v, min_v, max_v);
}

// _step_req (yaml_env.py:319-331) + Agent::cmd (agent.cpp:186-283)
__device__ __forceinline__ void integrate_robot(const DevWorld& w, const float* __restrict__ actions, int l) {
    double* r = w.rec + (size_t)(w.r0 + l) * IMGENV_RECORD_DOUBLES;
    double v = (double)actions[3 * l], wv = (double)actions[3 * l + 1];
    const double v_y = (double)actions[3 * l + 2];
    const double step_hz = w.step_hz, control_hz = 0.05;
    limiter_limit(w.lv_has_v, w.lv_has_a, w.lv_has_j, w.lv_min_v, w.lv_max_v, w.lv_min_a, w.lv_max_a, w.lv_min_j,
                  w.lv_max_j, v, w.l0v[l], w.l1v[l], step_hz);
    limiter_limit(w.lw_has_v, w.lw_has_a, w.lw_has_j, w.lw_min_v, w.lw_max_v, w.lw_min_a, w.lw_max_a, w.lw_min_j,
                  w.lw_max_j, wv, w.l0w[l], w.l1w[l], step_hz);
    w.l1v[l] = w.l0v[l];
    w.l1w[l] = w.l0w[l];
    w.l0v[l] = v;
    w.l0w[l] = wv;
    bool is_arrive = false;
    const double gx = w.gx[l], gy = w.gy[l];
    double ox = r[0], oy = r[1], oz = r[2];
    double vx = r[3], vy = r[4];
    const bool omni = w.ktype == IMGENV_KTYPE_OMNI;
    double cur_control = 0;
    while (cur_control <= step_hz) {
        const double c = cos(oz), s = sin(oz);
        if (!omni) {
            ox += v * control_hz * c;
            oy += v * control_hz * s;
            vx = v * c;
            vy = v * s;
        } else {
            ox += v * control_hz * c - v_y * control_hz * s;
            oy += v * control_hz * s + v_y * control_hz * c;
        }
        oz += wv * control_hz;
        const double cur_dist = sqrt((ox - gx) * (ox - gx) + (oy - gy) * (oy - gy));
        if (cur_dist <= 0.3) {
            is_arrive = true;
            break;
        }
        cur_control += control_hz;
    }
    const double theta = r[2], dt = step_hz;
    double x = r[0], y = r[1];
    if (wv == 0) {
        if (!omni) {
            x += v * dt * cos(theta);
            y += v * dt * sin(theta);
        } else {
            x += v * dt * cos(theta) - v_y * dt * sin(theta);
            y += v * dt * sin(theta) + v_y * dt * cos(theta);
        }
    } else {
        const double vw = v / wv;
        x += -vw * sin(theta) + vw * sin(theta + wv * dt);
        y += vw * cos(theta) - vw * cos(theta + wv * dt);
        if (omni) {
            const double v_yw = v_y / wv;
            x += -v_yw * cos(theta) + v_yw * cos(theta + wv * dt);
            y += -v_yw * sin(theta) + v_yw * sin(theta + wv * dt);
        }
    }
    const double th = theta + wv * dt;
    const double cur_dist = sqrt((x - gx) * (x - gx) + (y - gy) * (y - gy));
    if (cur_dist <= 0.3) is_arrive = true;
    r[0] = x;
    r[1] = y;
    r[2] = th;
    r[3] = vx;
    r[4] = vy;
    r[5] = sin(th * 0.5);  // Quaternion::setRPY(0,0,theta) of the new pose, shared by the next kernels
    r[6] = cos(th * 0.5);
    w.is_arr[l] = is_arrive ? 1 : 0;
}

// ImgEnv::_step_ped_dataset (img_env.cpp:361-386): pedestrian j replays record min(step, len - 1) of its trajectory
__device__ __forceinline__ void ped_dataset_one(const DevWorld& w, int j, int step) {
    const int len = w.ptraj_len[j];
    const int idx = step >= len ? len - 1 : step;
    const double* tp = w.ptraj + ((size_t)j * w.traj_cap + idx) * 3;
    const double* tv = w.ptraj_v + ((size_t)j * w.traj_cap + idx) * 3;
    const double ox = w.ppx[j], oy = w.ppy[j];
    const double x = tp[0], y = tp[1], vx = tv[0], vy = tv[1];
    w.plx[j] = ox;  // set_position (agent.cpp:691-694)
    w.ply[j] = oy;
    w.ppx[j] = x;
    w.ppy[j] = y;
    w.pyaw[j] = tv[2];  // atan2(vy, vx), evaluated by the host's libm at reset
    w.pvx[j] = vx;
    w.pvy[j] = vy;
    w.ped_state[4 * j] = x;
    w.ped_state[4 * j + 1] = y;
    w.ped_state[4 * j + 2] = vx;
    w.ped_state[4 * j + 3] = vy;
    ped_leg_gait(w, j, x, y, ox, oy);
}

__device__ __forceinline__ void state_robot(const DevWorld& w, int l);  // Agent::get_state, below

// Serial fallback of the robot update (more sub-steps than INT_ITEMS - 2): one thread per robot.
__global__ void k_integrate_serial(DevWorld w, const float* __restrict__ actions, int nb_robot, int step) {
    if ((int)blockIdx.x >= nb_robot) {
        const int j = ((int)blockIdx.x - nb_robot) * blockDim.x + threadIdx.x;
        if (j < w.P) {
            if (w.scene == IMGENV_SCENE_DATASET) ped_dataset_one(w, j, step - w.world_epoch[world_of_ped(w, j)]);
            else ped_update_one(w, j);
        }
        return;
    }
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = l < w.RL;
    if (valid && !w.py_done[l]) integrate_robot(w, actions, l);
    if (w.state_in_integrate && valid) state_robot(w, l);  // no pedestrians, no side stream: get_state right behind the move
    if (w.sharded) {
        const double* r = w.rec + (size_t)(w.r0 + (valid ? l : 0)) * IMGENV_RECORD_DOUBLES;
        bbox_accumulate(w, valid, r[0], r[1]);
    }
}

// The limited command of local robot l (SpeedLimiter::limit, speed_limit.cpp:92-173) ...
__device__ __forceinline__ void integrate_command(const DevWorld& w, const float* __restrict__ actions, int l, double& v, double& wv, double& v_y) {
    v = (double)actions[3 * l];
    wv = (double)actions[3 * l + 1];
    v_y = (double)actions[3 * l + 2];
    limiter_limit(w.lv_has_v, w.lv_has_a, w.lv_has_j, w.lv_min_v, w.lv_max_v, w.lv_min_a, w.lv_max_a, w.lv_min_j,
                  w.lv_max_j, v, w.l0v[l], w.l1v[l], w.step_hz);
    limiter_limit(w.lw_has_v, w.lw_has_a, w.lw_has_j, w.lw_min_v, w.lw_max_v, w.lw_min_a, w.lw_max_a, w.lw_min_j,
                  w.lw_max_j, wv, w.l0w[l], w.l1w[l], w.step_hz);
}
// ... (cos, sin) of table item `it`: the heading of sub-step it < n_sub (odom.z, rounded as the reference's loop rounds it),
// the new heading (it == n_sub) and its half (it == n_sub + 1) ...
__device__ __forceinline__ double2 integrate_heading(double theta, double wv, double step_hz, int it, int n_sub) {
    double a;
    if (it < n_sub) {
        a = theta;
        for (int q = 0; q < it; q++) a += wv * 0.05;
    } else {
        a = theta + wv * step_hz;
        if (it == n_sub + 1) a = a * 0.5;
    }
    double sn, cs;
    sincos(a, &sn, &cs);  // (one argument reduction for both)
    return make_double2(cs, sn);
}
// the exact pose update of Agent::cmd (agent.cpp:221-236 diff, 238-274 omni) from the sines / cosines of the old and the new heading
// (one function for k_integrate and for the early k_obs, which advances its robot by itself: the same operations in the same order)
__device__ __forceinline__ void pose_arc(bool omni, double v, double wv, double v_y, double dt, double c0, double s0, double c1, double s1,
                                         double& x, double& y) {
    if (wv == 0) {
        if (!omni) {
            x += v * dt * c0;
            y += v * dt * s0;
        } else {
            x += v * dt * c0 - v_y * dt * s0;
            y += v * dt * s0 + v_y * dt * c0;
        }
    } else {
        const double vw = v / wv;
        x += -vw * s0 + vw * s1;
        y += vw * c0 - vw * c1;
        if (omni) {
            const double v_yw = v_y / wv;
            x += -v_yw * c0 + v_yw * c1;
            y += -v_yw * s0 + v_yw * s1;
        }
    }
}
// ... and the position recurrence with the arrive tests plus the exact arc, over that table (one lane)
__device__ __forceinline__ void integrate_finish(const DevWorld& w, int l, double* r, double v, double wv, double v_y, double theta,
                                                 const double2* trig, int n_sub) {
    const double step_hz = w.step_hz, control_hz = 0.05;
    w.l1v[l] = w.l0v[l];
    w.l1w[l] = w.l0w[l];
    w.l0v[l] = v;
    w.l0w[l] = wv;
    bool is_arrive = false;
    const double gx = w.gx[l], gy = w.gy[l];
    double ox = r[0], oy = r[1];
    double vx = r[3], vy = r[4];
    const bool omni = w.ktype == IMGENV_KTYPE_OMNI;
    for (int q = 0; q < n_sub; q++) {
        const double c = trig[q].x, s = trig[q].y;
        if (!omni) {
            ox += v * control_hz * c;
            oy += v * control_hz * s;
            vx = v * c;
            vy = v * s;
        } else {
            ox += v * control_hz * c - v_y * control_hz * s;
            oy += v * control_hz * s + v_y * control_hz * c;
        }
        const double cur_dist = sqrt((ox - gx) * (ox - gx) + (oy - gy) * (oy - gy));
        if (cur_dist <= 0.3) {
            is_arrive = true;
            break;
        }
    }
    const double dt = step_hz;
    const double c0 = trig[0].x, s0 = trig[0].y;          // cos / sin(theta): sub-step 0's heading is theta
    const double c1 = trig[n_sub].x, s1 = trig[n_sub].y;  // cos / sin(theta + w dt)
    double x = r[0], y = r[1];
    pose_arc(omni, v, wv, v_y, dt, c0, s0, c1, s1, x, y);
    const double th = theta + wv * dt;
    const double cur_dist = sqrt((x - gx) * (x - gx) + (y - gy) * (y - gy));
    if (cur_dist <= 0.3) is_arrive = true;
    r[0] = x;
    r[1] = y;
    r[2] = th;
    r[3] = vx;
    r[4] = vy;
    r[5] = trig[n_sub + 1].y;  // Quaternion::setRPY(0,0,theta) of the new pose, shared by the next kernels
    r[6] = trig[n_sub + 1].x;
    w.is_arr[l] = is_arrive ? 1 : 0;
}

#define INT_G 8        // lanes per robot in k_integrate
#define INT_ITEMS 32   // sin / cos pairs per robot: the sub-step headings, the new heading and its half
#define INT_ROBOTS 32  // robots per 256-thread block

// One launch for the two independent per-agent updates of a step: blocks [0, nb_robot) integrate the robots,
// the blocks behind them move the ORCA pedestrians by the velocities k_orca solved for.
//
// Agent::cmd is a chain of ~20 dependent fp64 sin / cos calls per robot (one heading per 0.05 s sub-step, agent.cpp:
// 221-236, then the exact arc).  The headings do not depend on the positions, so 8 lanes per robot evaluate them side by
// side -- each lane re-accumulates `oz += w * 0.05` up to its own sub-step, which keeps the reference's rounding -- and
// lane 0 then runs the (cheap) position recurrence and the arrive tests over the table in LDS.
// One wavefront in front of a side stream's work polls a word of world.h's `sync` until it has reached `want` (sequence numbers
// wrap: signed difference).  Relaxed loads -- an acquire per poll would invalidate this XCD's L2 underneath whatever else runs on
// it, millions of times a second -- and one acquire at the end.  Nothing a gate waits for is queued behind it (the move is launched
// first), so it cannot starve its own signal; a word that never arrives raises the device flag after 60 s instead
// of hanging the queue for good (the caller's stream may well spend a long time in a policy's kernels before it reaches the step:
// the first call of a network that is still being tuned; an event wait would sit there just as long).
__global__ void k_gate(const uint32_t* word, uint32_t want, int* err) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 6000000000ull) {  // (100 MHz)
            err[7] = 1;
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Can a kernel on one stream wait for a kernel on another that is queued BEHIND it?  Not when something runs the process's
// kernels strictly one at a time (rocprofv3's counter collection does): a gate would then sit there until its bound.  Asked once per
// process (imgenv_create): this kernel polls for up to 20 ms (a busy chip may take its time to start the other one),
// k_gate_probe_set -- launched behind it on another stream -- stores the word.
__global__ void k_gate_probe(uint32_t* word) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    uint32_t seen = 0;
    while ((seen = __hip_atomic_load(&word[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && wall_clock64() - t0 < 2000000ull) __builtin_amdgcn_s_sleep(16);
    word[1] = seen ? 1u : 2u;
}
__global__ void k_gate_probe_set(uint32_t* word) { __hip_atomic_store(&word[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(INT_G * INT_ROBOTS) void k_integrate(DevWorld w, const float* __restrict__ actions, int nb_robot, int n_sub, int step, uint32_t seq) {
    __shared__ double2 trig[INT_ROBOTS][INT_ITEMS];  // (cos, sin)
    // the step's critical chain (move -> rasters -> views) runs beside the observation's 8192 wavefronts, which are bound by vector
    // issue: its kernels' wavefronts ask the SIMD's arbiter for the issue slot first (s_setprio; k_obs keeps the default 0)
    __builtin_amdgcn_s_setprio(3);
    // (world.h: sync) the caller's stream has reached this step
    if (seq && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&w.sync[0], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)blockIdx.x >= nb_robot) {
        const int j = ((int)blockIdx.x - nb_robot) * blockDim.x + threadIdx.x;
        if (j < w.P) {
            if (w.scene == IMGENV_SCENE_DATASET) ped_dataset_one(w, j, step - w.world_epoch[world_of_ped(w, j)]);
            else ped_update_one(w, j);
        }
        return;
    }
    const int g = threadIdx.x & (INT_G - 1), rb = threadIdx.x / INT_G;
    const int l = blockIdx.x * INT_ROBOTS + rb;
    const bool valid = l < w.RL;
    const bool alive = valid && !w.py_done[l];  // alive = (dones == 0); dead robots keep their pose (img_env.cpp:392)
    double* r = w.rec + (size_t)(w.r0 + (valid ? l : 0)) * IMGENV_RECORD_DOUBLES;
    double v = 0, wv = 0, v_y = 0, theta = 0;
    if (alive) {
        integrate_command(w, actions, l, v, wv, v_y);
        theta = r[2];
        for (int it = g; it < n_sub + 2; it += INT_G) trig[rb][it] = integrate_heading(theta, wv, w.step_hz, it, n_sub);
    }
    __syncthreads();
    if (alive && g == 0) integrate_finish(w, l, r, v, wv, v_y, theta, trig[rb], n_sub);
    if (w.state_in_integrate && valid && g == 0) state_robot(w, l);  // no pedestrians, no side stream: get_state right behind the move
    // (the shard's box: only where the other ranks' robots are rasterised and clipped to it -- the composed layers; its four contended
    // atomics per wavefront made this kernel 50 us instead of 13 at 8192 robots)
    if (w.sharded && !w.sum_shard) bbox_accumulate(w, valid && g == 0, r[0], r[1]);
}

// ------------------------------------------------------------------------------------------------
// Rasters: blocks [0, P) draw pedestrians, blocks [P, P + R) draw robots.

// view_ped (img_env.cpp:594-618).  Agent::draw(...,1) and PedAgent::draw_leg only ever write the
// value 1 and their skip rules depend on the UNDERLYING obstacle value alone:
//   circle sample : writes unless the cell is 0 / 1 / 2          (agent.cpp:315-322)
//   left leg      : writes unless the cell is 0                   (agent.cpp:751-754)
//   right leg     : writes unless the cell is 1 -> always ends 1  (agent.cpp:767-770)
// so the sequential result is order independent: peds_map = ped_layer ? 1 : obs_map.
// One pedestrian sample on cell c (or on none).  rule: 0 circle sample, 1 left-leg sample, 2 right-leg sample (see above).
// Composed mode writes the pedestrian layer.  STAMP mode reads the base class out of the class layer itself and lets only
// the last lane of each run of equal cells stamp (consecutive samples are lattice neighbours and mostly share their cell:
// ~10 compare-and-swaps per pedestrian instead of 900 dependent loads).  Call from wave-uniform control flow.
template <bool STAMP>
__device__ __forceinline__ void ped_sample(const DevWorld& w, bool in, uint32_t c, int rule, int lane, int world) {
    if (!STAMP) {
        if (in) {
            const uint32_t o = w.obs_map[c];
            if (rule == 2 || (rule == 1 ? o != 0u : o > 2u)) w.ped_layer[c] = 1;
        }
        return;
    }
    const uint32_t ci = in ? c : 0xFFFFFFFFu;
    const uint32_t next = (uint32_t)__shfl_down((int)ci, 1);
    if (in && (lane == WAVE - 1 || next != ci)) {
        const uint32_t v = w.cell[c], base = v & 7u;
        if (rule == 2 || (rule == 1 ? base != CLS_STATIC : base >= CLS_LOW)) stamp_ped(w, c, v, stamp_tag_of(w), world);
    }
}

// PB rounds of samples at once (see stamp_ped_batch): the words of all rounds are read together (the robots' batches, by contrast,
// only pay in small launches: raster_robot)
template <bool STAMP, int PB>
__device__ __forceinline__ void ped_samples(const DevWorld& w, const bool (&in)[PB], const uint32_t (&c)[PB], int rule, int lane, int world) {
    if (!STAMP) {
#pragma unroll
        for (int u = 0; u < PB; u++) ped_sample<false>(w, in[u], c[u], rule, lane, world);
        return;
    }
    bool last[PB], go[PB];
    uint32_t v[PB];
#pragma unroll
    for (int u = 0; u < PB; u++) {  // the last lane of each run of equal cells stamps (ped_sample)
        const uint32_t ci = in[u] ? c[u] : 0xFFFFFFFFu;
        const uint32_t next = (uint32_t)__shfl_down((int)ci, 1);
        last[u] = in[u] && (lane == WAVE - 1 || next != ci);
        v[u] = last[u] ? w.cell[c[u]] : 0u;
    }
#pragma unroll
    for (int u = 0; u < PB; u++) {
        const uint32_t base = v[u] & 7u;
        go[u] = last[u] && (rule == 2 || (rule == 1 ? base != CLS_STATIC : base >= CLS_LOW));
    }
    stamp_ped_batch<PB>(w, c, go, v, stamp_tag_of(w), world);
}

// the literal walk over one sample list of a pedestrian (Agent::draw / PedAgent::draw_leg): thread `first` of `stride` takes
// every stride-th sample, PB rounds at a time (ped_samples)
template <bool POW2, bool STAMP>
__device__ __forceinline__ void ped_walk(const DevWorld& w, const Tf2& bw, const Tf2* lb, const double* sx, const double* sy, int n_s, int rule,
                                         int first, int stride, const Region& g, uint32_t cell0, int world) {
    constexpr int PB = 4;  // rounds of samples per batch (ped_samples)
    const double res = w.res, inv = w.inv_res;
    const int lane = lane_id();
    for (int q0 = 0; q0 < n_s; q0 += stride * PB) {  // wave-uniform trip count (lane shuffles inside)
        uint32_t cc[PB];
        bool in[PB];
#pragma unroll
        for (int u = 0; u < PB; u++) {
            in[u] = false;
            cc[u] = 0u;
            if (q0 + u * stride >= n_s) continue;  // uniform
            const int q = min(q0 + u * stride + first, n_s - 1);
            double bx = sx[q], by = sy[q], wx, wy;
            if (lb) tf_apply(*lb, sx[q], sy[q], bx, by);
            tf_apply(bw, bx, by, wx, wy);
            int m, n;
            w2m_pair<POW2>(wx, wy, res, inv, m, n);
            in[u] = q0 + u * stride + first < n_s && m >= g.m0 && m < g.m1 && n >= g.n0 && n < g.n1;
            cc[u] = cell0 + (uint32_t)(m * w.Wg + n);
        }
        ped_samples<STAMP, PB>(w, in, cc, rule, lane, world);
    }
}

// ... and the same cells from the list's lattice rows, one row per lane (fp_rows.h; raster_robot has the story).  px_off / cy: a
// leg disc's rows are shifted by the leg's offset in the pedestrian's frame.  Returns false when some row of THIS wavefront could
// not be certified: the wavefront then walks the whole list itself (the layer is order independent and idempotent).
template <bool STAMP, int NW>
__device__ __forceinline__ bool ped_rows(const DevWorld& w, const Tf2& bw, const FpRow* rows, int n_rows, double px_off, double cy, int rule,
                                         const Region& g, uint32_t cell0, int world) {
    constexpr int NT = WAVE * NW;
    const int lane = lane_id(), tid = threadIdx.x;
    const FpRowsPose P = fpr_pose(bw.m00, bw.m01, bw.m10, bw.m11, bw.ox, bw.oy, cy, w.res);
    bool bad = false;
    for (int r0 = 0; r0 < n_rows; r0 += NT) {  // wave-uniform trip count
        const int ri = r0 + tid;
        const bool act = ri < n_rows;
        FpRow row = rows[min(ri, n_rows - 1)];
        row.px = row.px + px_off;
        FpAxis ax, ay;
        const bool ok = fpr_row(P, row, ax, ay);
        bad |= act & !ok;
        const bool use = act & ok;
        const int cx = use ? ax.cnt : 0, cyn = use ? ay.cnt : 0;
        const int cxw = (int)__any(cx >= 1) + (int)__any(cx >= 2) + (int)__any(cx >= 3) + (int)__any(cx >= 4);
        const int cyw = (int)__any(cyn >= 1) + (int)__any(cyn >= 2) + (int)__any(cyn >= 3) + (int)__any(cyn >= 4);
#pragma unroll
        for (int i = 0; i <= FPR_MAXC; i++) {
            if (i > cxw) break;
#pragma unroll
            for (int j = 0; j <= FPR_MAXC; j++) {
                if (j > cyw) break;
                int m, n;
                uint32_t last;
                const bool has = fpr_piece(row, ax, ay, i, j, m, n, last) & use;
                const bool in = has && m >= g.m0 && m < g.m1 && n >= g.n0 && n < g.n1;
                ped_sample<STAMP>(w, in, cell0 + (uint32_t)(m * w.Wg + n), rule, lane, world);  // (neighbouring rows mostly share the cell: one stamp per run)
            }
        }
    }
    return !__any(bad);
}

// SUM mode (world.h): the pedestrian's footprint goes into an LDS box around its cell -- which leg covers what (the legs' skip
// rules differ, agent.cpp:751 / 767) --, the box cells that pass their rule are what view_ped would have drawn, and the
// pedestrian's count comes off the cells it no longer draws on and goes onto the new ones.  A pedestrian that stands still
// issues no atomic at all.
template <bool POW2, int NW>
__device__ __forceinline__ void raster_ped_sum(const DevWorld& w, int j, const PedClassDev& k, uint32_t* box) {
    constexpr int NT = WAVE * NW;
    const int world = world_of_ped(w, j), tid = threadIdx.x, lane = lane_id();
    const uint32_t cell0 = (uint32_t)world * w.Gs;
    const double res = w.res, inv = w.inv_res;
    const double px = w.ppx[j], py = w.ppy[j];
    const Tf2 bw = tf_from_pose(px, py, w.pyaw[j]);
    const int rad = k.box_rad, side = 2 * rad + 1, ncell = side * side;
    const int cm = w2m_t<POW2>(px, res, inv), cn = w2m_t<POW2>(py, res, inv);
    uint32_t* n_sh = box + w.box_cells + 1;  // (the words behind the box: see raster_robot)
    for (int q = tid; q < ncell; q += NT) box[q] = 0;
    if (tid == 0) *n_sh = 0;
    __syncthreads();
    const int Hg = w.Hg, Wg = w.Wg;
    int* err = w.err;
    auto mark = [&](int m, int n, uint32_t bit) {
        if (m < 0 || m >= Hg || n < 0 || n >= Wg) return;
        const int dm = m - cm + rad, dn = n - cn + rad;
        if (dm >= 0 && dm < side && dn >= 0 && dn < side) atomicOr(&box[dm * side + dn], bit);
        else err[6] = 1;  // (cannot happen: the box is the footprint's extent in every gait state + 2 cells)
    };
    const bool leg_shape = k.shape == IMGENV_SHAPE_LEG;
    for (int part = 0; part < (leg_shape ? 2 : 1); part++) {
        if (!leg_shape && k.shape != IMGENV_SHAPE_CIRCLE) break;  // (view_ped draws discs and legs only, img_env.cpp:599-616)
        Tf2 lb;  // get_leg_base (agent.cpp:815-821): a pure translation
        tf_set_rotation_zw(lb, 0.0, 1.0);
        lb.ox = !leg_shape ? 0.0 : part == 0 ? w.llx[j] : w.rlx[j];
        lb.oy = !leg_shape ? 0.0 : part == 0 ? w.lly[j] : w.rly[j];
        const FpRow* rows = !leg_shape ? k.brows : part == 0 ? k.lrows : k.rrows;
        const int n_rows = !leg_shape ? k.n_brows : part == 0 ? k.n_lrows : k.n_rrows;
        const double* sx = !leg_shape ? k.bx : part == 0 ? k.lx : k.rx;
        const double* sy = !leg_shape ? k.by : part == 0 ? k.ly : k.ry;
        const int n_s = !leg_shape ? k.n_bbox : part == 0 ? k.n_left : k.n_right;
        const uint32_t bit = part == 0 ? 1u : 2u;
        bool bad = n_rows == 0;
        if (n_rows > 0) {  // lattice rows (fp_rows.h), one per lane
            const FpRowsPose P = fpr_pose(bw.m00, bw.m01, bw.m10, bw.m11, bw.ox, bw.oy, leg_shape ? lb.oy : k.bbox_cy, res);
            for (int r0 = 0; r0 < n_rows; r0 += NT) {
                const int ri = r0 + tid;
                FpRow row = rows[min(ri, n_rows - 1)];
                row.px = row.px + lb.ox;
                FpAxis ax, ay;
                const bool ok = fpr_row(P, row, ax, ay);
                bad |= ri < n_rows && !ok;
                const bool use = ri < n_rows && ok;
                const int cx = use ? ax.cnt : 0, cyn = use ? ay.cnt : 0;
                const int cxw = (int)__any(cx >= 1) + (int)__any(cx >= 2) + (int)__any(cx >= 3) + (int)__any(cx >= 4);
                const int cyw = (int)__any(cyn >= 1) + (int)__any(cyn >= 2) + (int)__any(cyn >= 3) + (int)__any(cyn >= 4);
#pragma unroll
                for (int a = 0; a <= FPR_MAXC; a++) {
                    if (a > cxw) break;
#pragma unroll
                    for (int b = 0; b <= FPR_MAXC; b++) {
                        if (b > cyw) break;
                        int m, n;
                        uint32_t last;
                        if (fpr_piece(row, ax, ay, a, b, m, n, last) && use) mark(m, n, bit);
                    }
                }
            }
        }
        if (__any(bad)) {  // this wavefront walks the whole list itself (marks are idempotent)
            for (int q = lane; q < n_s; q += WAVE) {
                double bx = sx[q], by = sy[q], wx, wy;
                if (leg_shape) tf_apply(lb, sx[q], sy[q], bx, by);
                tf_apply(bw, bx, by, wx, wy);
                int m, n;
                w2m_pair<POW2>(wx, wy, res, inv, m, n);
                mark(m, n, bit);
            }
        }
    }
    __syncthreads();
    // what view_ped draws: a disc skips cells holding 0 / 1 / 2, a left leg cells holding 0, a right leg nothing (agent.cpp:313-320, 751, 767)
    for (int b = tid; b < ncell; b += NT) {
        const uint32_t f = box[b];
        if (f) {
            const int bm = b / side;
            const uint32_t c = cell0 + (uint32_t)(cm - rad + bm) * (uint32_t)Wg + (uint32_t)(cn - rad + (b - bm * side));
            const uint32_t base = w.cell[c] & 7u;
            const bool draws = leg_shape ? ((f & 2u) != 0u || base != CLS_STATIC) : base >= CLS_LOW;
            if (draws) box[b] = f | 4u;
        }
    }
    __syncthreads();
    const uint32_t one = 1u << 3;
    uint32_t* list = w.pd_cells + (size_t)j * w.pd_cap;
    const int n_old = w.pd_n[j];
    for (int e = tid; e < n_old; e += NT) {
        const uint32_t c = list[e], rel = c - cell0;
        const int m = (int)(((unsigned long long)rel * w.sum_wg_magic) >> 40), n = (int)rel - m * Wg;
        const int dm = m - cm + rad, dn = n - cn + rad;
        const bool inb = dm >= 0 && dm < side && dn >= 0 && dn < side;
        const uint32_t f = inb ? box[dm * side + dn] : 0u;
        if (f & 4u) atomicOr(&box[dm * side + dn], 8u);  // still drawn: the count stands
        else atomicAdd(&w.cell[c], 0u - one);
    }
    __syncthreads();
    int n_out = 0;
    for (int b0 = 0; b0 < ncell; b0 += NT) {  // wave-uniform trip count (ballots inside)
        const int b = b0 + tid;
        const uint32_t f = b < ncell ? box[b] : 0u;
        const bool go = (f & 4u) != 0u;
        const int bm = b / side;
        const uint32_t c = cell0 + (uint32_t)(cm - rad + bm) * (uint32_t)Wg + (uint32_t)(cn - rad + (b - bm * side));
        if (go && !(f & 8u)) atomicAdd(&w.cell[c], one);
        const unsigned long long mask = __ballot(go);
        const int cnt = __popcll(mask);
        int base = n_out;
        if (NW > 1) {
            if (lane == 0 && cnt) base = (int)atomicAdd(n_sh, (uint32_t)cnt);
            base = __builtin_amdgcn_readfirstlane(base);
        }
        const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
        if (go && pos < w.pd_cap) list[pos] = c;
        n_out += cnt;
    }
    if (NW > 1) {
        __syncthreads();
        n_out = (int)*n_sh;
    }
    if (tid == 0) {
        w.pd_n[j] = min(n_out, w.pd_cap);
        if (n_out > w.pd_cap) err[6] = 2;
    }
}

template <bool POW2, bool STAMP, int NW>
__device__ __forceinline__ void raster_ped(const DevWorld& w, int j, const PedClassDev& k, const Region& g) {
    constexpr int NT = WAVE * NW;  // NW wavefronts share the samples (see k_raster)
    const int world = world_of_ped(w, j);
    const uint32_t cell0 = (uint32_t)world * w.Gs;  // this world's copy of the layers
    const Tf2 bw = tf_from_pose(w.ppx[j], w.ppy[j], w.pyaw[j]);
    const int lane = lane_id(), tid = threadIdx.x;
    if (k.shape == IMGENV_SHAPE_CIRCLE) {
        if (k.n_brows > 0) {
            if (!ped_rows<STAMP, NW>(w, bw, k.brows, k.n_brows, 0.0, k.bbox_cy, 0, g, cell0, world))
                ped_walk<POW2, STAMP>(w, bw, nullptr, k.bx, k.by, k.n_bbox, 0, lane, WAVE, g, cell0, world);
        } else {
            ped_walk<POW2, STAMP>(w, bw, nullptr, k.bx, k.by, k.n_bbox, 0, tid, NT, g, cell0, world);
        }
    } else if (k.shape == IMGENV_SHAPE_LEG) {
        for (int leg = 0; leg < 2; leg++) {
            Tf2 lb;  // get_leg_base: Quaternion(0,0,0,1), origin = leg (agent.cpp:815-821)
            tf_set_rotation_zw(lb, 0.0, 1.0);
            lb.ox = leg == 0 ? w.llx[j] : w.rlx[j];
            lb.oy = leg == 0 ? w.lly[j] : w.rly[j];
            const int n_s = leg == 0 ? k.n_left : k.n_right;
            const double* sx = leg == 0 ? k.lx : k.rx;
            const double* sy = leg == 0 ? k.ly : k.ry;
            const FpRow* rows = leg == 0 ? k.lrows : k.rrows;
            const int n_rows = leg == 0 ? k.n_lrows : k.n_rrows;
            if (n_rows > 0) {
                // (the leg frame is a pure translation: a disc sample (x, y) sits at (x + leg x, y + leg y) in the pedestrian's frame, exactly)
                if (!ped_rows<STAMP, NW>(w, bw, rows, n_rows, lb.ox, lb.oy, leg + 1, g, cell0, world))
                    ped_walk<POW2, STAMP>(w, bw, &lb, sx, sy, n_s, leg + 1, lane, WAVE, g, cell0, world);
            } else {
                ped_walk<POW2, STAMP>(w, bw, &lb, sx, sy, n_s, leg + 1, tid, NT, g, cell0, world);
            }
        }
    }
}

#define RASTER_BOX_CELLS 2048  // LDS box of a robot raster: up to 45 x 45 cells

// view_robot's inner loop (img_env.cpp:624-628) for ALL robots at once: instead of stamping every
// other robot into a private copy of the grid per robot (O(R * Hg*Wg + R^2 * F)), each robot records
// itself in two shared layers, own_lo = min id and own_hi = max id covering a cell.  Robot i then
// sees "another robot" in a cell iff (lo != i or hi != i).  The 901 footprint samples fall on a few
// cells, so they are de-duplicated in an LDS box around the robot first: ~10-20 global atomics per
// robot.  The box keeps the LAST sample index that landed in each cell; for local robots the
// (cell, last sample) pairs go to fp_cells so that the collision test of k_view (agent.cpp:294-326:
// the last footprint sample on an occupied cell decides) needs one gather per covered cell and no
// second pass over the samples.
// LM: how the class layer is kept (world.h): 0 owner layers + k_compose, 1 stamps, 2 counts (SUM)
template <bool POW2, int LM, int NW>
__device__ __forceinline__ void raster_robot(const DevWorld& w, int i, const RobotClassDev& k, uint32_t* box, const Region& g) {
    constexpr bool STAMP = LM == 1, SUM = LM == 2;
    constexpr int NT = WAVE * NW;  // NW wavefronts share the samples (see k_raster)
    constexpr int UB = NW > 1 ? 4 : 1;   // cells a lane stamps at once (stamp_robot_batch)
    RASTER_MARK_BEGIN();
    const double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
    const int lane = lane_id(), tid = threadIdx.x;
    const Tf2 bw = tf_from_pose_sc(r[0], r[1], r[5], r[6]);
    const double res = w.res, inv = w.inv_res;
    const uint32_t id = (uint32_t)i + 1;
    const int rad = k.box_rad, side = 2 * rad + 1, ncell = side * side;
    const bool use_box = ncell <= w.box_cells;
    const int l = i - w.r0;
    const bool local = l >= 0 && l < w.RL;
    const int world = world_of_robot(w, i);
    const uint32_t cell0 = (uint32_t)world * w.Gs;  // this world's copy of the layers
    const int cm = w2m_t<POW2>(r[0], res, inv), cn = w2m_t<POW2>(r[1], res, inv);
    // view_robot (img_env.cpp:620-629) draws every OTHER robot of the world into a robot's map: the only robot of its world has
    // nobody to be drawn for, so it leaves no stamps (the shipped configs: one robot per env -- its 27 x 27-cell footprint at
    // 0.015 m was 54 lines of class words + crop_map + segment tags fetched and written back per step for nothing); what it still
    // needs is its own (cell, last sample) list for the collision test
    const bool alone = (w.W > 1 ? w.Rw : w.R) == 1;
    // _step_robot tail: setRobotPos for every robot (img_env.cpp:411-417); the RVO scenes get theirs from k_side_robots
    if (tid == 0 && w.relation == 1 && w.scene == IMGENV_SCENE_PEDSIM) {  // PedScene::setRobotPos: setPosition(px, py, 1)
        double* p = w.sfm.p + 3 * ((size_t)world * w.sfm.n + w.sfm.n_peds + (size_t)(i - world * w.Rw));  // the world's crowd: pedestrians, then robots
        p[0] = r[0];
        p[1] = r[1];
        p[2] = 1.0;
    }
    // another rank's robot only matters where this rank's robots can see it (a local robot's footprint is inside the
    // region by construction, so the clip never changes its own cells)
    // (SUM mode in a shard keeps every robot's counts everywhere: they persist, and the shard's box moves)
    const bool whole = local || (SUM && w.sum_shard != 0);
    const int lo_m = whole ? 0 : g.m0, hi_m = whole ? w.Hg : g.m1, lo_n = whole ? 0 : g.n0, hi_n = whole ? w.Wg : g.n1;
    if (!whole && (cm + rad < lo_m || cm - rad >= hi_m || cn + rad < lo_n || cn - rad >= hi_n)) return;
    // A robot that has not moved since its cell list was made (frozen after a collision or an arrival, dead, or simply
    // standing still) covers the same cells: re-stamp them from the list instead of walking the 901 samples again.
    if (local) {
        double* cached = w.fp_pose + 3 * (size_t)l;
        const int n_cached = w.fp_n[l];
        if (n_cached >= 0 && cached[0] == r[0] && cached[1] == r[1] && cached[2] == r[2]) {
            if (alone || SUM) return;  // (the list stands, and there is nobody to stamp for -- or, SUM mode, the robot's counts stand with it)
            const uint2* list = w.fp_cells + (size_t)l * w.fp_cap;
            for (int e0 = 0; e0 < n_cached; e0 += NT * UB) {
                uint32_t c[UB];
                bool go[UB];
#pragma unroll
                for (int u = 0; u < UB; u++) {
                    const int e = e0 + u * NT + tid;
                    go[u] = e < n_cached;
                    c[u] = go[u] ? list[e].x : 0u;
                }
                // (rounds past the end of a short list cost a few predicated-off instructions each)
                if (STAMP) {
                    stamp_robot_batch<UB>(w, c, go, (uint32_t)i, stamp_tag_of(w), world);
                } else {
#pragma unroll
                    for (int u = 0; u < UB; u++)
                        if (go[u]) {
                            atomicMin(&w.own_lo[c[u]], id);
                            atomicMax(&w.own_hi[c[u]], id);
                        }
                }
            }
            return;
        }
        if (NW > 1) __syncthreads();  // every wavefront has compared before the pose is replaced
        if (tid == 0) {
            cached[0] = r[0];
            cached[1] = r[1];
            cached[2] = r[2];
        }
    }
    uint32_t* stray_flag = box + w.box_cells;  // NW > 1: "some sample fell outside the box", seen by any wavefront
    RASTER_MARK(5);  // pose, cached-list test
    if (use_box) {
        for (int q = tid; q < ncell; q += NT) box[q] = 0;
        if (NW > 1 && tid == 0) stray_flag[0] = stray_flag[2] = 0;
        __syncthreads();
    }
    bool stray = false;
    // The lattice rows of the footprint, one per lane (fp_rows.h): the cells a row covers and the last sample in each are PREDICTED
    // from the real-arithmetic model and certified against the rounding boundaries -- a dozen instructions per cell of a row
    // instead of ~50 per sample of it (901 samples on a dozen cells at 0.25 m).  Any row that cannot be certified sends the whole
    // robot through the literal sample walk below, which adds the same values to the box again (atomicMax: idempotent).
    bool rows_done = false;
    if (use_box && k.n_rows > 0) {
        uint32_t* rows_bad = stray_flag + 2;
        const FpRowsPose P = fpr_pose(bw.m00, bw.m01, bw.m10, bw.m11, bw.ox, bw.oy, k.fp_cy, res);
        bool bad = false;
        for (int r0 = 0; r0 < k.n_rows; r0 += NT) {  // wave-uniform trip count
            const int ri = r0 + tid;
            const bool act = ri < k.n_rows;
            const FpRow row = k.rows[min(ri, k.n_rows - 1)];
            FpAxis ax, ay;
            const bool ok = fpr_row(P, row, ax, ay);
            bad |= act & !ok;
            const bool use = act & ok;
            const int cx = use ? ax.cnt : 0, cy = use ? ay.cnt : 0;
            // (the pieces of a row are walked up to the wavefront's largest step counts: scalar loop bounds, constant register indices)
            const int cxw = (int)__any(cx >= 1) + (int)__any(cx >= 2) + (int)__any(cx >= 3) + (int)__any(cx >= 4);
            const int cyw = (int)__any(cy >= 1) + (int)__any(cy >= 2) + (int)__any(cy >= 3) + (int)__any(cy >= 4);
            static_assert(FPR_MAXC == 4, "the wavefront's largest step count is gathered with four ballots");
#pragma unroll
            for (int i = 0; i <= FPR_MAXC; i++) {
                if (i > cxw) break;
#pragma unroll
                for (int j = 0; j <= FPR_MAXC; j++) {
                    if (j > cyw) break;
                    int m, n;
                    uint32_t last;
                    const bool has = fpr_piece(row, ax, ay, i, j, m, n, last) & use;
                    if (has && m >= lo_m && m < hi_m && n >= lo_n && n < hi_n) {
                        const int dm = m - cm + rad, dn = n - cn + rad;
                        if (dm >= 0 && dm < side && dn >= 0 && dn < side) atomicMax(&box[dm * side + dn], last);
                        else bad = true;  // (cannot happen: the box is the footprint's extent + 2 cells; the walk below would stamp it)
                    }
                }
            }
        }
        if (NW > 1) {
            if (bad) *rows_bad = 1;
            __syncthreads();
            rows_done = *rows_bad == 0;
        } else {
            rows_done = !__any(bad);
        }
    }
    // (the next four samples are requested before these four are used: the static table's round trip is then underneath the
    // transforms instead of in front of them, four times per robot)
    double2 fp_next[4];
    const int n_walk = rows_done ? 0 : k.n_fp;  // the literal walk: classes without rows, boxes too small, uncertified poses
#pragma unroll
    for (int u = 0; u < 4; u++) fp_next[u] = rows_done ? make_double2(0.0, 0.0) : k.fp[min(u * NT + tid, k.n_fp - 1)];
    for (int q0 = 0; q0 < n_walk; q0 += NT * 4) {  // wave-uniform trip count (lane shuffles inside), 4 loads in flight
        double2 fp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) fp[u] = fp_next[u];
        if (q0 + NT * 4 < k.n_fp) {
#pragma unroll
            for (int u = 0; u < 4; u++) fp_next[u] = k.fp[min(q0 + NT * 4 + u * NT + tid, k.n_fp - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = q0 + u * NT + tid;
            double wx, wy;
            tf_apply(bw, fp[u].x, fp[u].y, wx, wy);
            int m, n;
            w2m_pair<POW2>(wx, wy, res, inv, m, n);
            int b = -1;  // box cell of this sample
            if (q < k.n_fp && m >= lo_m && m < hi_m && n >= lo_n && n < hi_n) {
                const int dm = m - cm + rad, dn = n - cn + rad;
                if (use_box && dm >= 0 && dm < side && dn >= 0 && dn < side) {
                    b = dm * side + dn;
                } else {
                    const size_t c = (size_t)cell0 + (size_t)m * w.Wg + n;
                    if (alone || SUM) {  // (SUM handles only have classes whose footprint fits the box: imgenv_create)
                    } else if (STAMP) {
                        stamp_robot(w, c, (uint32_t)i, stamp_tag_of(w), world);
                    } else {
                        atomicMin(&w.own_lo[c], id);
                        atomicMax(&w.own_hi[c], id);
                    }
                    stray = true;
                }
            }
            // consecutive samples are lattice neighbours and mostly share their cell: only the last lane of each run of
            // equal cells (it holds the run's highest sample index) touches the LDS box -- a handful of atomics
            // instead of 64 on a few addresses
            const int b_next = __shfl_down(b, 1);
            if (b >= 0 && (lane == WAVE - 1 || b_next != b)) atomicMax(&box[b], (uint32_t)q + 1);
        }
    }
    RASTER_MARK(6);  // footprint samples into the LDS box
    if (use_box) {
        uint32_t* n_sh = stray_flag + 1;  // NW > 1: entries of the cell list so far
        if (NW > 1 && stray) *stray_flag = 1;
        if (NW > 1 && tid == 0) *n_sh = 0;
        __syncthreads();
        // every wavefront turns its share of the box into stamps and list entries, four box cells per lane at a time (the words
        // of all four are read together and their compare-and-swaps go out together: stamp_robot_batch)
        uint2* list = w.fp_cells + (size_t)(local ? l : 0) * w.fp_cap;
        int n_out = 0;
        const uint32_t sum_word = SUM ? sum_robot_word(w, sum_self_id(w, i)) : 0u;
        // The covered cells as a BITMAP: what a robot shard's other ranks need of this robot (world.h: sum_shard), carried by the
        // record's eighth double -- written in every layer mode, so that any handle's records can feed a shard.  It spans the
        // box without its margin of two cells: a sample lies within `ext` of the robot's origin and round(a + b) - round(a) is
        // within +-ceil(|b|), so the covered cells are within box_rad - 2 = ceil(ext / res) of the centre cell (checked below).
        const int brad = rad - 2, bside = 2 * brad + 1;
        unsigned long long fp_bits = 0ull;
        if (bside * bside <= WAVE && tid < WAVE) {
            const int bm = tid / bside, bn = tid - bm * bside;
            fp_bits = __ballot(tid < bside * bside && box[(bm + 2) * side + bn + 2] != 0u);
            if (SUM && !local) {  // another rank's robot, rasterised here (a reset: every rank knows every pose from the batch)
                if (lane == 0) sum_apply_bits(w, i, fp_bits, cm, cn, brad, bside, cell0);
            } else if (local && lane == 0) {
                ((unsigned long long*)w.rec)[(size_t)i * IMGENV_RECORD_DOUBLES + 7] = fp_bits;
            }
        }
        if (SUM && !local) return;  // (uniform)
        if (SUM && !alone) {
            // the cells this robot counted itself on so far: still under its footprint -> marked in the box (top bit: nothing to add),
            // left behind -> its word comes off again.  (Handles in SUM mode own every robot: `local` always holds.)
            const int n_old = max(w.fp_n[l], 0);
            for (int e = tid; e < n_old; e += NT) {
                const uint32_t c = list[e].x, rel = c - cell0;
                const int m = (int)(((unsigned long long)rel * w.sum_wg_magic) >> 40), n = (int)rel - m * w.Wg;
                const int dm = m - cm + rad, dn = n - cn + rad;
                const bool inb = dm >= 0 && dm < side && dn >= 0 && dn < side;
                const uint32_t v = inb ? box[dm * side + dn] : 0u;
                if (v) atomicOr(&box[dm * side + dn], 0x80000000u);
                else atomicAdd(&w.cell[c], 0u - sum_word);
            }
            __syncthreads();
        }
        for (int b0 = 0; b0 < ncell; b0 += NT * UB) {  // wave-uniform trip count (ballots inside)
            uint32_t c[UB], last[UB];
            bool go[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int b = b0 + u * NT + tid;
                last[u] = 0u;
                go[u] = false;
                c[u] = 0u;
                if (b0 + u * NT >= ncell) continue;  // uniform: a small box is one round
                last[u] = b < ncell ? box[b] : 0u;
                const bool counted = SUM && (last[u] >> 31) != 0u;
                if (SUM) last[u] &= 0x7FFFFFFFu;
                go[u] = last[u] != 0u;
                const int bm = b / side;
                const int m = cm - rad + bm, n = cn - rad + (b - bm * side);
                c[u] = go[u] ? cell0 + (uint32_t)m * (uint32_t)w.Wg + (uint32_t)n : 0u;
                if (SUM && go[u] && !counted && !alone) {  // a cell the robot has just entered
                    // (the one returning atomic of the layer: the count field is at least 6 bits wide and reset poses are the caller's --
                    // the robot that would carry it into the index sum says so instead of corrupting the word silently)
                    const uint32_t before = atomicAdd(&w.cell[c[u]], sum_word);
                    const uint32_t rc_mask = (1u << (w.sum_id_shift - w.sum_rc_shift)) - 1u;
                    if (((before >> w.sum_rc_shift) & rc_mask) == rc_mask) w.err[6] = 9;
                }
            }
            if (alone || SUM) {
            } else if (STAMP) {
                stamp_robot_batch<UB>(w, c, go, (uint32_t)i, stamp_tag_of(w), world);
            } else {
#pragma unroll
                for (int u = 0; u < UB; u++)
                    if (go[u]) {
                        atomicMin(&w.own_lo[c[u]], id);
                        atomicMax(&w.own_hi[c[u]], id);
                    }
            }
            if (local) {
#pragma unroll
                for (int u = 0; u < UB; u++) {
                    const unsigned long long mask = __ballot(go[u]);
                    const int cnt = __popcll(mask);
                    int base = n_out;
                    if (NW > 1) {  // (the list's order does not matter: k_view takes a maximum over it)
                        if (lane == 0 && cnt) base = (int)atomicAdd(n_sh, (uint32_t)cnt);
                        base = __builtin_amdgcn_readfirstlane(base);
                    }
                    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
                    if (go[u] && pos < w.fp_cap) list[pos] = make_uint2(c[u], last[u]);
                    n_out += cnt;
                }
            }
        }
        if (NW > 1) {
            __syncthreads();
            n_out = (int)*n_sh;
        }
        RASTER_MARK(7);  // box -> stamps + cell list
        const bool any_stray = NW > 1 ? *stray_flag != 0 : __any(stray);
        if (local && tid == 0) w.fp_n[l] = (n_out <= w.fp_cap && !any_stray) ? n_out : -1;
        // (a covered cell in the box's margin would be missing from the bitmap: never, by the bound above -- said aloud if it happens)
        if (w.sum_shard && local && tid == 0 && __popcll(fp_bits) != n_out) w.err[6] = 10;
    } else if (local && tid == 0) {
        w.fp_n[l] = -1;
    }
}

// NW: wavefronts per workgroup.  1 when a launch fills the machine; 4 in small launches (a reset of a few worlds), where the
// 15 rounds of footprint samples of one wavefront are pure latency.
template <bool POW2, int LM, int NW>
__global__ __launch_bounds__(WAVE * NW) __attribute__((amdgpu_waves_per_eu((LM == 1 || !POW2) ? 6 : LM == 2 ? 7 : 8, 8))) void k_raster(DevWorld w, int zero_vel, int split) {
    constexpr bool STAMP = LM == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // max(R, P) blocks: block b draws robot b and pedestrian b.  (P + R single-purpose blocks would be 200 more
    // than the 8192 wavefronts one MI355X holds at once in the headline configuration: a second, nearly empty round.)
    const int b = blockIdx.x;
    WAVE_T0();
    __builtin_amdgcn_s_setprio(3);  // (critical chain: see k_integrate)
    if (LM != 0 && b == 0 && threadIdx.x == 0) w.counters[1] = 0;  // tail_group tallies this step's dones (k_compose does this otherwise)
    const Region g = grid_region(w);
    // split > 0 (small launches): the first `split` blocks draw robots, the ones behind them pedestrians -- a robot and a
    // pedestrian one after the other in the same block is twice one block's chain of memory round trips
    if (b < act_count_g(w) && (split == 0 || b < split)) {
        const int i = act_member(w, w.Rw, b) + w.act_g0;
        raster_robot<POW2, LM, NW>(w, i, robot_class(w, w.robot_cls[i]), (uint32_t*)smem, g);
    }
    const int bp = split > 0 ? b - split : b;
    if (bp >= 0 && bp < act_count_p(w)) {
        RASTER_MARK_BEGIN();
        const int j = act_member(w, w.Pw, bp);
        if (LM == 2) {
            if (split == 0) __syncthreads();  // (the robot of this block is done with the box)
            raster_ped_sum<POW2, NW>(w, j, w.pc[w.ped_cls[j]], (uint32_t*)smem);
        } else {
            raster_ped<POW2, STAMP, NW>(w, j, w.pc[w.ped_cls[j]], g);
        }
        RASTER_MARK(12);  // the block's pedestrian
    }
    if (b < w.RL) WAVE_DONE(2);
}

// k_integrate and k_raster as ONE launch, for steps of handles that leave the chip room (imgenv_step on at most 4096 robots, all of
// them local): a robot's footprint needs nothing but its own new pose, a pedestrian's nothing but its own move, so the robot's
// block integrates it first -- the headings one per lane (integrate_heading), the position recurrence on lane 0 -- and the
// pedestrian's block moves it first.  One launch and one stream dependency less in front of the views; the side stream's fork
// moves behind this kernel, the observation and the solve then run underneath k_view.  (At 8192 robots that placement loses --
// k_obs and k_view are both bound by vector issue, DESIGN.md section 4 -- so the big launches keep the two kernels.)
// Blocks [0, R): robots, [R, R + P): pedestrians.
template <bool POW2, int LM, int NW>
__global__ __launch_bounds__(WAVE * NW) __attribute__((amdgpu_waves_per_eu((LM == 1 || !POW2) ? 6 : LM == 2 ? 7 : 8, 8)))
void k_move_raster(DevWorld w, const float* __restrict__ actions, int n_sub, int step, int move_peds) {
    constexpr bool STAMP = LM == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double2 trig[INT_ITEMS];  // (cos, sin)
    const int b = blockIdx.x, tid = threadIdx.x;
    if (LM != 0 && b == 0 && tid == 0) w.counters[1] = 0;  // (as k_raster)
    const Region g = grid_region(w);
    const int n_rob = w.act_ng;  // the handle's robots: every robot of the world (r0 = 0), or a shard's own (world.h: sum_shard)
    if (b < n_rob) {
        const int l = b, gi = w.act_g0 + b;
        // (the class index in a scalar register: behind the stores below the compiler would fetch it per lane, and a per-lane index
        // into the class records of the kernel argument sends the whole argument through scratch memory)
        const int cls = __builtin_amdgcn_readfirstlane(w.robot_cls[gi]);
        double* r = w.rec + (size_t)gi * IMGENV_RECORD_DOUBLES;
        if (!w.py_done[l]) {  // alive = (dones == 0); dead robots keep their pose (img_env.cpp:392)
            double v, wv, v_y;
            integrate_command(w, actions, l, v, wv, v_y);
            const double theta = r[2];
            if (tid < n_sub + 2) trig[tid] = integrate_heading(theta, wv, w.step_hz, tid, n_sub);
            __syncthreads();
            if (tid == 0) integrate_finish(w, l, r, v, wv, v_y, theta, trig, n_sub);
        }
        if (w.state_in_integrate && tid == 0) state_robot(w, l);  // no pedestrians, no side stream: get_state right behind the move
        __syncthreads();  // the new record, for every lane
        raster_robot<POW2, LM, NW>(w, gi, robot_class(w, cls), (uint32_t*)smem, g);
    } else if (b - n_rob < w.P) {
        const int j = b - n_rob;
        const int cls = __builtin_amdgcn_readfirstlane(w.ped_cls[j]);
        if (move_peds) {
            if (tid == 0) {
                if (w.scene == IMGENV_SCENE_DATASET) ped_dataset_one(w, j, step - w.world_epoch[world_of_ped(w, j)]);
                else ped_update_one(w, j);
            }
            __syncthreads();
        }
        if (LM == 2) raster_ped_sum<POW2, NW>(w, j, w.pc[cls], (uint32_t*)smem);
        else raster_ped<POW2, STAMP, NW>(w, j, w.pc[cls], g);
    }
}

// SUM mode in a robot shard (world.h: sum_shard): the robots of the OTHER ranks, as the exchange has just delivered them -- a thread each
template <bool POW2>
__global__ __launch_bounds__(256) void k_remote(DevWorld w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= w.R - w.RL) return;
    const int i = t < w.r0 ? t : t + w.RL;
    const double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
    // (the bitmap's radius: one for the whole world when every class has the same, which spares two dependent loads)
    const int rad = (w.rm_rad >= 0 ? w.rm_rad : w.rc_mem[w.robot_cls[i]].box_rad - 2), side = 2 * rad + 1;
    const int cm = w2m_t<POW2>(r[0], w.res, w.inv_res), cn = w2m_t<POW2>(r[1], w.res, w.inv_res);
    sum_apply_bits(w, i, (unsigned long long)__double_as_longlong(r[7]), cm, cn, rad, side, 0u);
}

// `cell` of the cells [c0, min(c0 + 4, G)) from the map and the raster layers; re-arms the raster layers
__device__ __forceinline__ void compose_cells_scalar(const DevWorld& w, size_t c0, size_t G) {
    for (size_t c = c0; c < G; c++) {
        const uint32_t base = w.ped_layer[c] ? 1u : w.obs_map[c];
        uint32_t b = base <= 2 ? base : (base < 250 ? CLS_LOW : CLS_HIGH);
        uint32_t own = 0;
        if (b >= CLS_LOW && w.own_hi[c] != 0) {
            b |= CLS_ROBOT;
            own = (w.own_lo[c] == w.own_hi[c]) ? w.own_lo[c] - 1 : OWNER_MULTI;
        }
        w.cell[c] = b | (own << 8);
        w.ped_layer[c] = 0;
        w.own_lo[c] = 0xFFFFFFFFu;
        w.own_hi[c] = 0;
    }
}
__device__ __forceinline__ void compose_cells(const DevWorld& w, size_t c0, size_t G) {
    if (c0 + 4 <= G) {
        const uint32_t obs = *(const uint32_t*)(w.obs_map + c0);
        const uint32_t ped = *(const uint32_t*)(w.ped_layer + c0);
        const uint4 lo = *(const uint4*)(w.own_lo + c0);
        const uint4 hi = *(const uint4*)(w.own_hi + c0);
        const uint32_t los[4] = {lo.x, lo.y, lo.z, lo.w}, his[4] = {hi.x, hi.y, hi.z, hi.w};
        uint32_t out = 0;
        uint32_t own[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t o = (obs >> (8 * q)) & 0xff, p = (ped >> (8 * q)) & 0xff;
            const uint32_t base = p ? 1u : o;
            uint32_t b = base <= 2 ? base : (base < 250 ? CLS_LOW : CLS_HIGH);
            own[q] = 0;
            if (b >= CLS_LOW && his[q] != 0) {
                b |= CLS_ROBOT;
                own[q] = (los[q] == his[q]) ? los[q] - 1 : OWNER_MULTI;
            }
            out |= b << (8 * q);
        }
        *(uint4*)(w.cell + c0) = make_uint4(((out >> 0) & 0xFFu) | (own[0] << 8), ((out >> 8) & 0xFFu) | (own[1] << 8),
                                             ((out >> 16) & 0xFFu) | (own[2] << 8), ((out >> 24) & 0xFFu) | (own[3] << 8));
        if (ped) *(uint32_t*)(w.ped_layer + c0) = 0;
        if (his[0] | his[1] | his[2] | his[3]) {
            *(uint4*)(w.own_lo + c0) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            *(uint4*)(w.own_hi + c0) = make_uint4(0, 0, 0, 0);
        }
    } else {
        compose_cells_scalar(w, c0, G);
    }
}

// class layer: one byte per cell that a robot's view kernel can decode without touching the three
// raster layers; also re-arms the raster layers for the next step (saves two memsets per step).
__global__ void k_compose(DevWorld w) {
    // every cell of the stacked layers, or the cells of the listed worlds (a fixed number of blocks per world)
    size_t G = w.act_cells, c0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (w.act_list) {
        const unsigned per_world = (w.Gs / 4 + blockDim.x - 1) / blockDim.x, q = blockIdx.x / per_world;
        if (w.act_n_dev && (int)q >= *w.act_n_dev) return;
        const size_t base = (size_t)w.act_list[q] * w.Gs;
        c0 = base + ((size_t)(blockIdx.x - q * per_world) * blockDim.x + threadIdx.x) * 4;
        G = base + w.Gs;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) w.counters[1] = 0;  // tail_group tallies this step's dones
    if (c0 >= G) return;
    if (w.sharded) {  // only the region this rank's rasters were clipped to (everything else is clean and unread)
        const Region g = grid_region(w);
        int m = (int)(c0 / (size_t)w.Wg), n = (int)(c0 - (size_t)m * w.Wg);
        bool any = false;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            any |= m >= g.m0 && m < g.m1 && n >= g.n0 && n < g.n1;
            if (++n == w.Wg) {
                n = 0;
                m++;
            }
        }
        if (!any) return;
    }
    compose_cells(w, c0, G);
}

// STAMP mode: the sweep that drops every stamp once per STAMP_TAGS steps, before their tags come round again (the base classes
// themselves are written at reset, with the obstacle maps).
__global__ __launch_bounds__(256) void k_cell_base(DevWorld w) {
    const size_t G = w.act_cells, c0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (c0 >= G) return;
    if ((c0 & 63) == 0) w.seg_tag[c0 >> 6] = 0;
    if (c0 + 4 <= G) {
        const uint4 v = *(const uint4*)(w.cell + c0);
        *(uint4*)(w.cell + c0) = make_uint4(v.x & 7u, v.y & 7u, v.z & 7u, v.w & 7u);
    } else {
        for (size_t c = c0; c < G; c++) w.cell[c] &= 7u;
    }
}

// ------------------------------------------------------------------------------------------------
// Agent::view (agent.cpp:356-509) for local robot l = blockIdx.x, one wavefront.
//   LDS: src[Hv*Wv + pad] u8  the cropped view (0 / 255 / 200); src[Hv*Wv] is a free dummy cell
//        hit[B] u32           first hit of each beam: step << 16 | row << 8 | col, 0xFFFFFFFF = none
//        colt[Wv] double2     the column terms of the view -> world transform
//
// bresenhamLine (agent.cpp:511-624) writes laser_map beam after beam, later beams overwriting earlier
// ones, so a cell ends with the value of the HIGHEST beam that writes it.  Beam paths are static, so
//   (a) hit[b]: every lane walks the precomputed path of its beams in LDS (8 steps per 16-byte load);
//       a step contributes a 16-bit key = value << 8 | step, two steps side by side in one register, and
//       the minimum key is the first occupied cell (value 0) -- no compares, no selects; the wave leaves
//       the loop as soon as all its beams have hit or ended;
//   (b) every view cell looks at the highest beam through it (static table `top_ent`): 255 before that
//       beam's hit, 0 at the hit, 200 behind it -- unless the cell shares a row or column with the hit
//       cell, where that beam leaves the cell alone (agent.cpp:555-560) and the next lower beam through
//       the cell decides (rare slow path over the static per-cell ray list);
//   (c) the same pass stamps the own footprint and writes both output planes straight from
//       registers: cell values live as 2-bit class indices, v_perm_b32 turns four of them into four
//       uint8 values / four float16 values.
// The kernel is written branch-free inside its loops (selects instead of divergent ifs): with one
// wavefront per robot the instruction issue rate, not memory, bounds it.

// collision code from the footprint samples themselves (classes whose box does not fit k_raster's LDS)
template <bool POW2, bool STAMP>
__device__ __forceinline__ uint32_t collision_from_samples(const DevWorld& w, const RobotClassDev& k, const Tf2& bw, uint32_t self) {
    const int lane = lane_id();
    const double res = w.res, inv = w.inv_res;
    uint32_t best = 0;
    for (int q = lane; q < k.n_fp; q += WAVE) {
        const double2 fp = k.fp[q];
        double wx, wy;
        tf_apply(bw, fp.x, fp.y, wx, wy);
        int m, n;
        w2m_pair<POW2>(wx, wy, res, inv, m, n);
        if (m >= 0 && m < w.Hg && n >= 0 && n < w.Wg) {
            const uint32_t v = w.cell[(size_t)world_of_robot(w, (int)self) * w.Gs + (size_t)m * w.Wg + n];
            const uint32_t cc = (!STAMP && w.layer_sum) ? cell_seen_class_sum(w, v, sum_self_id(w, (int)self))
                                                        : cell_seen_class<STAMP>(v, self, stamp_tag_of(w));
            if (cc <= 2) best = max(best, ((uint32_t)(q + 1) << 2) | (cc + 1));
        }
    }
    return best;
}

// v_pk_min_u16: the minimum of the low halves and of the high halves
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}

// laser_map value of a cell its top beam leaves alone: the next lower beam through the cell that writes decides.
// hit word of a beam: first-hit step << 16 | last step behind the hit that still shares a row or column with the hit cell
// (the beam leaves the steps in between alone, agent.cpp:555-560); a beam without a hit holds 0xFFFFFFFF
__device__ __forceinline__ uint32_t resolve_skipped_cell(const RobotClassDev& k, const uint32_t* hit, uint32_t c) {
    const uint32_t pk = k.inv_pack[c];
    const uint32_t e0 = pk & 0xFFFFFu, cnt = pk >> 20;
    for (uint32_t e = 1; e < cnt; e++) {  // entry 0 is the top beam
        const uint32_t ent = k.inv_ent[e0 + e];
        const uint32_t kk = ent & 0xFFFFu, hp = hit[ent >> 16], hk = hp >> 16;
        if (kk < hk) return 3u;   // 255
        if (kk == hk) return 0u;  // 0
        if (kk > (hp & 0xFFFFu)) break;  // this beam writes 200
    }
    return 2u;
}

// the resolved class of such a cell into the two output planes, unless it is the provisional 200 (class 2) they already hold;
// st: the cell lies under the own footprint (100 over anything but 0)
__device__ __forceinline__ void patch_stamped(uint8_t* out_u8, uint16_t* out_f16, uint32_t h01, uint32_t h3, uint32_t c, uint32_t v, bool st) {
    if (v == 2u) return;
    v = (st && v != 0u) ? 1u : v;
    out_u8[c] = (uint8_t)(v == 0u ? 0u : (v == 1u ? 100u : 255u));
    out_f16[c] = (uint16_t)(v == 0u ? (h01 & 0xFFFFu) : (v == 1u ? (h01 >> 16) : h3));
}

// inclusive prefix sum over the 64 lanes of a wavefront (every lane must be active): four row shifts, two row broadcasts
__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return x;
}

// angular_map_ (agent.cpp:407-433): nearest hit of each 1/72 of the field of view, one bin per thread from the beams' hit words
// (first-hit step << 16 | ..., 0xFFFFFFFF = no hit); a bin's beams are a contiguous range
__device__ __forceinline__ void angular_bins(const DevWorld& w, const RobotClassDev& k, const uint32_t* hit, int l, int tid, int nt) {
    for (int m = tid; m < IMGENV_ANGULAR_BINS; m += nt) {
        float best = w.view_max_dist32;
        for (int b = k.bin_start[m]; b < (int)k.bin_start[m + 1]; b++) {
            const uint32_t hp = hit[b];
            const float hd = hp == 0xFFFFFFFFu ? 6.0f : k.ray_dist[(size_t)(hp >> 16) * k.ray_stride + b];
            best = fminf(best, hd);
        }
        w.angular_map[(size_t)l * IMGENV_ANGULAR_BINS + m] = best;
    }
}

// NW: wavefronts per robot.  1 when a launch fills the machine (instruction issue bounds it); 4 for small launches (a reset of
// a few worlds), where the single wavefront's latency is all there is: the cells and beams are then spread over 256 lanes.
template <bool POW2, bool A4, bool STAMP, int NW>
__global__ __launch_bounds__(WAVE * NW) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_view(DevWorld w) {
    constexpr int NT = WAVE * NW;
    const int tid = threadIdx.x;
    // A4: Wv % 4 == 0 (a lane's 4 consecutive cells share their row and nothing runs over the end of the view)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= act_count_l(w)) return;
    const int l = act_member(w, w.Rw, blockIdx.x);
    const int lane = lane_id();
    // The collision test's chain -- how many cells the footprint covers, the cells, their words -- is three dependent round trips
    // that need nothing but l: on their way from here, underneath the prologue's own (flags, class record, pose), instead of behind them
    const int n_cov = w.fp_n[l];
    const uint2* fp_list = w.fp_cells + (size_t)l * w.fp_cap;
    // (the list entry without waiting for the count -- a lane beyond it reads an old entry or the neighbour's, clamped into the layer;
    // the four-wavefront variant waits for it: on cfg-5's 800 x 800 layer the stray gathers of four wavefronts cost it 2.5 %)
    uint2 ce_first = make_uint2(0u, 0u);
    uint32_t v_first = 0u;
    if (NW != 4 || lane < n_cov) {
        ce_first = fp_list[min(lane, w.fp_cap - 1)];
        v_first = w.cell[min(ce_first.x, (uint32_t)(w.W > 1 ? w.W : 1) * w.Gs - 1u)];
    }
    // the robot's record as this chain leaves it, where the next step's early k_obs finds it while the move rewrites the original (world.h)
    if (w.rec_snap_out && tid < IMGENV_RECORD_DOUBLES) {
        const double v = w.rec[(size_t)(w.r0 + l) * IMGENV_RECORD_DOUBLES + tid];
        w.rec_snap_out[(size_t)l * IMGENV_RECORD_DOUBLES + tid] = v;
        if (w.rec_snap_out2) w.rec_snap_out2[(size_t)l * IMGENV_RECORD_DOUBLES + tid] = v;
    }
    if (w.is_coll[l] || w.is_arr[l]) {  // frozen: every per-robot output keeps its last value (counted in tail_group)
        if (tid < WAVE) tail_arrive_view(w, blockIdx.x, l, w.is_coll[l]);
        return;
    }
    const int i = w.r0 + l;
    const RobotClassDev k = robot_class(w, w.robot_cls[i]);
    const double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
    const Tf2 bw = tf_from_pose_sc(r[0], r[1], r[5], r[6]);
    const int Hv = w.Hv, Wv = w.Wv, NC = Hv * Wv, NCp = (NC + 16) & ~15;
    const int Hg = w.Hg, Wg = w.Wg;
    const double res = w.res, inv = w.inv_res;
    const uint32_t wv_magic = w.wv_magic;
    const bool laser = w.use_laser != 0;
    const uint32_t self = (uint32_t)i;
    // the `cell` values this robot sees as free (>= 250, agent.cpp:394-401).  Composed layer: the plain class, or this robot as
    // the only owner.  STAMP layer: base class HIGH under this robot's own stamp of this step, or under no stamp of this step
    // (tested on the word with base and tag XORed against HIGH and our tag: a non-zero multiple of 32)
    const uint32_t tag = STAMP ? stamp_tag_of(w) : 0u;
    const uint32_t free_plain = CLS_HIGH;
    const bool layer_sum = !STAMP && w.layer_sum != 0;  // (uniform) counts instead of owners: the same two compares per crop cell
    const uint32_t self_w = sum_self_id(w, i);
    const uint32_t free_own = STAMP ? (CLS_HIGH | (STAMP_ONE << STAMP_KIND_SHIFT) | (tag << STAMP_TAG_SHIFT) | (self << STAMP_OWNER_SHIFT))
                              : layer_sum ? (CLS_HIGH + sum_robot_word(w, self_w))
                                          : (CLS_HIGH | CLS_ROBOT | (self << 8));
    const uint32_t base_tag_mask = 7u | (0xFFu << STAMP_TAG_SHIFT), base_tag_ours = CLS_HIGH | (tag << STAMP_TAG_SHIFT);
    const uint32_t cell0 = (uint32_t)world_of_robot(w, i) * w.Gs;  // this world's copy of the layers
    uint8_t* src = smem;
    uint32_t* hit = (uint32_t*)(smem + NCp);
    double2* colt = (double2*)(smem + NCp + 4 * (size_t)w.hit_stride);
    int* skip_cnt = (int*)(colt + Wv);  // [0] list entries (NW > 1), [1] chunk descriptors, [2] result slots of the final pass
    uint32_t* reach_tab = (uint32_t*)(skip_cnt + 4);  // largest hit word of overlapping blocks of 16 / 32 / 64 beams (filter of (5))
    PHASE_BEGIN();
    VIEW_PRIO(3);
    EXP_SKEW();

    // (1) is_collision_ = draw(grid, -1, "world_map", bbox_): the LAST footprint sample that hits decides
    //     the code (agent.cpp:294-326) -> max over (last sample index in the cell, code) of the covered cells
    uint32_t best = 0;
    if (n_cov >= 0) {
        for (int e = lane; e < n_cov; e += WAVE) {
            const uint2 ce = e == lane ? ce_first : fp_list[e];
            const uint32_t v = e == lane ? v_first : w.cell[ce.x];
            const uint32_t cc = layer_sum ? cell_seen_class_sum(w, v, self_w) : cell_seen_class<STAMP>(v, self, stamp_tag_of(w));
            best = max(best, cc <= 2 ? ((ce.y << 2) | (cc + 1)) : 0u);
        }
    } else {
        best = collision_from_samples<POW2, STAMP>(w, k, bw, self);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, off));
    const int code = (int)(best & 3);
    // the collision code is all the step's tail needs from this kernel: hand it over NOW, with no store of this wavefront in
    // flight -- if this wavefront completes its group, the tails' loads do not queue up behind 7 KB of view stores (in-order
    // vmcnt), and they run underneath the other wavefronts instead of behind the last one
    if (tid < WAVE) tail_arrive_view(w, blockIdx.x, l, code);
    PHASE_MARK(0);

    // (2) egocentric crop (agent.cpp:373-404): 4 view cells per lane per round -> one LDS dword.
    //     Grid cell of view cell (a, b) = round(((m00 (a res) + m01 (b res)) + ox) / res), i.e. m00 a + m01 b + ox / res in cells
    //     up to a few ulps.  An fp64 operation, a convert or a multiply costs a SIMD 4.2 cycles per wavefront, an integer add 2.4
    //     (tools/micro/valu_issue.hip), so the fast path runs in 32.32 fixed point: the column terms (+ the origin + 0.5) come
    //     from an LDS table, the row term is shared by the 4 cells of a lane, a cell costs one 64-bit add per axis, its index
    //     is the high word and the low word tells how close the value is to a rounding boundary.  Coefficient rounding (2^-33
    //     each) moves the sum by < 2^-24 cells for views below 256 cells; values within 2^-17 of a boundary -- and exact ties,
    //     where C round() goes away from zero -- take the reference's own fp64 chain.  (Measured: 737 -> ~450 vector
    //     instructions for the crop, the step +0.8 %: the phase waits for its gathers, 4 in flight per lane; 8 in flight need
    //     76 registers -- 6 wavefronts per SIMD instead of 8 -- or spill, and lose 6 %.)
    const Tf2 vw = tf_mul(bw, w.view_base);  // get_view_world (agent.cpp:128-131)
    const double two32 = 4294967296.0;
    const double oxs = vw.ox * inv, oys = vw.oy * inv;
    const long long M00 = (long long)rint(vw.m00 * two32), M10 = (long long)rint(vw.m10 * two32);
    const bool fixed_ok = fabs(oxs) < 1048576.0 && fabs(oys) < 1048576.0;  // (always, for a pose anywhere near its map)
    longlong2* coli = (longlong2*)colt;
    {
        const long long M01 = (long long)rint(vw.m01 * two32), M11 = (long long)rint(vw.m11 * two32);
        const long long OX = (long long)rint(oxs * two32) + (1ll << 31), OY = (long long)rint(oys * two32) + (1ll << 31);
        for (int b = tid; b < Wv; b += NT) coli[b] = make_longlong2(OX + b * M01, OY + b * M11);
    }
    if (tid < 16) src[NC + tid] = 255;  // dummy free cells behind the view (padded path entries)
    if (tid < 3) skip_cnt[tid] = 0;
    __syncthreads();
    // Which groups of four cells?  A reset takes every cell; a step only the groups in which a beam crosses a cell (no laser: a cell
    // inside the field of view) -- static list per class, host_tables.h.  Nobody reads the crop of another cell (the beams walk
    // their paths, the final pass without a laser reads the cells of the field of view), and in the outputs the other cells hold
    // their 200, or the own footprint's 100, since the reset.  (48 x 48 cells, 360 beams over 180 degrees: 241 of 576 groups.)
    const bool every_cell = w.tail_is_reset != 0 || EXP_EVERY_CELL;
    const int n_groups = every_cell ? (NC + 3) >> 2 : k.n_dyn;
    // (a group's word -- c4, field-of-view bits, own-footprint bits -- is fetched a round ahead: a wavefront's lifetime is its chain
    // of dependent round trips, and list -> bits -> cells were three of them per round)
    const uint32_t* glist = every_cell ? k.all_groups : k.dyn_groups;
    uint32_t g_next = tid < n_groups ? glist[tid] : 0u;
    for (int gi = tid; gi < n_groups; gi += NT) {
        const uint32_t g_cur = g_next;
        if (gi + NT < n_groups) g_next = glist[gi + NT];
        const int c4 = (int)(g_cur & 0xFFFFu);
        const uint32_t fov = (g_cur >> 16) & 0xFu;
        uint32_t packed = 200u | (200u << 8) | (200u << 16) | (200u << 24);
        if (fov) {
            const int a0 = (int)__umulhi((uint32_t)c4, wv_magic), b0 = c4 - a0 * Wv;
            const long long rx0 = a0 * M00, ry0 = a0 * M10;
            int m[4], n[4];
            bool risky = !fixed_ok;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                long long rx = rx0, ry = ry0;
                int b = b0 + q;
                if (!A4) {  // the group may run over the end of its row (or of the view)
                    const int c = min(c4 + q, NC - 1);
                    const int a = (int)__umulhi((uint32_t)c, wv_magic);
                    b = c - a * Wv;
                    rx = a * M00;
                    ry = a * M10;
                }
                const longlong2 cc = coli[b];
                const unsigned long long fx = (unsigned long long)(rx + cc.x), fy = (unsigned long long)(ry + cc.y);
                m[q] = (int)(uint32_t)(fx >> 32);
                n[q] = (int)(uint32_t)(fy >> 32);
                const uint32_t G = 1u << 15;
                risky |= ((uint32_t)fx + G < 2u * G) | ((uint32_t)fy + G < 2u * G);
            }
            if (__builtin_expect(__any(risky), 0)) {  // close to a rounding boundary somewhere in the wave: the literal chain
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c = min(c4 + q, NC - 1);
                    const int a = (int)__umulhi((uint32_t)c, wv_magic), b = c - a * Wv;
                    double wx, wy;
                    tf_apply(vw, a * res, b * res, wx, wy);
                    m[q] = w2m_t<POW2>(wx, res, inv);
                    n[q] = w2m_t<POW2>(wy, res, inv);
                }
            }
            uint32_t idx[4], okm[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool ok = (A4 || c4 + q < NC) & (((fov >> q) & 1u) != 0) & ((uint32_t)m[q] < (uint32_t)Hg) & ((uint32_t)n[q] < (uint32_t)Wg);
                okm[q] = ok ? 0xFFu : 0u;
                idx[q] = ok ? cell0 + (uint32_t)(m[q] * Wg + n[q]) : 0u;
            }
            uint32_t v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = w.cell[idx[q]];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // free (>= 250, agent.cpp:394-401) and no other robot on it: the plain class, or this robot as the only owner
                bool free_cell;
                if (STAMP) {
                    const uint32_t x = (v[q] & base_tag_mask) ^ base_tag_ours;
                    free_cell = (v[q] == free_own) | (((x & 7u) == 0u) & (x != 0u));
                } else {
                    free_cell = (v[q] == free_plain) | (v[q] == free_own);
                }
                const uint32_t val = free_cell ? 255u : 0u;
                packed = (packed & ~(okm[q] << (8 * q))) | ((val & okm[q]) << (8 * q));
            }
        }
        *(uint32_t*)(src + c4) = packed;
    }
    __syncthreads();
    PHASE_MARK(1);
    VIEW_PRIO(2);
    EXP_STOP_AFTER(1);  // instruction accounting builds: collision + crop only

    // (3) laser (agent.cpp:405-438): first occupied cell on each beam's precomputed Bresenham path
    if (laser) {
        const uint4* rows = (const uint4*)k.ray_rows;
        const int n_chunks = k.ray_kpad >> 3;
        for (int b0 = 0; b0 < w.B; b0 += NT) {
            const int b = b0 + tid;
            const int bb = min(b, w.B - 1);
            const int len = b < w.B ? (int)k.ray_len[bb] : 0;
            // two 16-bit keys side by side, value << 8 | step (a path has at most 255 steps): even steps in the low half,
            // odd ones in the high half, one packed min for both
            uint32_t found2 = 0xFFFFFFFFu;
            uint4 nxt = rows[bb];
            for (int ch = 0; ch < n_chunks; ch++) {
                const uint4 cur = nxt;
                if (ch + 1 < n_chunks) nxt = rows[(size_t)(ch + 1) * k.ray_stride + bb];  // in flight while this chunk is walked
                const uint32_t wds[4] = {cur.x, cur.y, cur.z, cur.w};
                const uint32_t steps0 = (uint32_t)(8 * ch) * 0x00010001u + 0x00010000u;  // (8 ch + 1) << 16 | 8 ch
#pragma unroll
                for (int j = 0; j < 4; j++) {  // padded entries point at the free dummy cell
                    const uint32_t lo = src[wds[j] & 0xFFFFu], hi = src[wds[j] >> 16];
                    const uint32_t two = __builtin_amdgcn_perm(hi, lo, 0x040C000Cu) | (steps0 + (uint32_t)j * 0x00020002u);
                    found2 = pk_min_u16(found2, two);
                }
                const uint32_t first = min(found2 & 0xFFFFu, found2 >> 16);
                if (__all((first < 0x0100u) | (8 * ch + 8 >= len))) break;
            }
            if (b < w.B) {
                const uint32_t first = min(found2 & 0xFFFFu, found2 >> 16);
                const bool has = first < 0x0100u;  // value 0 in the key's top byte
                const uint32_t hk = first & 0xFFu;
                // how far behind the hit the beam stays in the hit cell's row or column: static per (step, beam)
                // ... and its distance, as float32 and as the `lasers` value (hd / laser_max when laser_norm): one 16-byte record
                const uint4 fin = k.ray_fin[(size_t)(has ? hk : 0u) * k.ray_stride + b];
                const uint32_t run = has ? fin.x : 0u;
                const float hd = has ? __uint_as_float(fin.y) : 6.0f;  // agent.cpp:513
                hit[b] = has ? ((hk << 16) | (hk + run)) : 0xFFFFFFFFu;
                w.lasers_raw[(size_t)l * w.B + b] = hd;
                w.lasers[(size_t)l * w.B + b] = has ? __hiloint2double((int)fin.w, (int)fin.z) : w.laser_out_nohit;
                if (w.hits_x) {  // hit_points_x_ / _y_ (agent.cpp:434-435), static per (step, beam); row ray_maxlen: no hit
                    const size_t at = (size_t)(has ? hk : (uint32_t)k.ray_maxlen) * k.ray_stride + b;
                    w.hits_x[(size_t)l * w.B + b] = k.ray_hx[at];
                    w.hits_y[(size_t)l * w.B + b] = k.ray_hy[at];
                }
            }
        }
        if (tid == 0) hit[w.B] = 0u;  // the dummy beam of cells without any (see the final pass)
        __syncthreads();
        if (w.angular_map) angular_bins(w, k, hit, l, tid, NT);
        // how far the beams get at most, for the filter of (5): blocks of 16 beams every 8, of 32 every 16, of 64 every 32 (a cell's
        // beams lie inside one of them unless they are more than 33), largest hit word = largest first-hit step
        const int nb8 = (w.B >> 3) + 1, n0 = nb8, n1 = (nb8 + 1) >> 1, n2 = (nb8 + 3) >> 2;
        for (int i = tid; i < n0; i += NT) {
            uint32_t m = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) m = max(m, hit[min(8 * i + q, w.B)]);
            reach_tab[i] = m;
        }
        __syncthreads();
        for (int i = tid; i < n1; i += NT) reach_tab[n0 + i] = max(reach_tab[2 * i], reach_tab[min(2 * i + 2, n0 - 1)]);
        __syncthreads();
        for (int i = tid; i < n2; i += NT) reach_tab[n0 + n1 + i] = max(reach_tab[n0 + 2 * i], reach_tab[n0 + min(2 * i + 2, n1 - 1)]);
    }
    PHASE_MARK(2);
    VIEW_PRIO(1);
    EXP_STOP_AFTER(2);  // ... + first hits

    // (4) laser_map (agent.cpp:437) per cell from its top beam, the own footprint stamped 100 (agent.cpp:503),
    //     stored as uint8 and as float16(v / 255) (yaml_env.py:431-438).  Class index: 0 -> 0, 1 -> 100, 2 -> 200, 3 -> 255
    uint8_t* out_u8 = w.view_maps + (size_t)l * NC;
    uint16_t* out_f16 = w.sensor_maps + (size_t)l * NC;
    const uint32_t lut_u8 = 0u | (100u << 8) | (200u << 16) | (255u << 24);
    const uint32_t h0 = w.f16_lut[0], h1 = w.f16_lut[100], h2 = w.f16_lut[200], h3 = w.f16_lut[255];
    const uint32_t lut_lo = (h0 & 0xFFu) | ((h1 & 0xFFu) << 8) | ((h2 & 0xFFu) << 16) | ((h3 & 0xFFu) << 24);
    const uint32_t lut_hi = (h0 >> 8) | ((h1 >> 8) << 8) | ((h2 >> 8) << 16) | ((h3 >> 8) << 24);
    uint32_t* skip_list = (uint32_t*)src;  // the crop is dead once the beams have their hits
    const uint32_t no_beam = ((uint32_t)w.B << 16) | 0xFFFFu;  // top_ent of a cell no beam crosses
    int n_skip = 0;
    g_next = tid < n_groups ? glist[tid] : 0u;
    for (int gi = tid; gi < n_groups; gi += NT) {  // (the groups of the crop)
        const uint32_t g_cur = g_next;
        if (gi + NT < n_groups) g_next = glist[gi + NT];
        const int c4 = (int)(g_cur & 0xFFFFu);
        uint32_t I = 0x02020202u;  // four class indices, one per byte; no beam through a cell: 200
        if (laser) {
            uint32_t top[4] = {no_beam, no_beam, no_beam, no_beam};
            if (A4 || c4 + 4 <= NC) {
                const uint4 t4 = *(const uint4*)(k.top_ent + c4);
                top[0] = t4.x; top[1] = t4.y; top[2] = t4.z; top[3] = t4.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (c4 + q < NC) top[q] = k.top_ent[c4 + q];
            }
            if (__any(min(min(top[0], top[1]), min(top[2], top[3])) < no_beam)) {  // rows behind the sensor see no beam at all
                uint32_t skips = 0;
                I = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // a cell no beam crosses looks up the dummy beam B at step 0xFFFF: that beam "hits" at step 0 and leaves
                    // nothing alone, so the cell is class 200
                    const uint32_t kk = top[q] & 0xFFFFu;
                    const uint32_t hp = hit[top[q] >> 16], hk = hp >> 16;  // hk = 0xFFFF when the beam never hits
                    const uint32_t v = kk < hk ? 3u : (kk == hk ? 0u : 2u);
                    skips |= ((kk > hk) & (kk <= (hp & 0xFFFFu))) ? (1u << q) : 0u;
                    I |= v << (8 * q);
                }
                // left alone by their top beam: provisional 200 now, resolved after the pass.  One list entry per lane and
                // round (c4 << 4 | the four flags): at most NC / 4 entries, which the dead crop always holds
                const unsigned long long mask = __ballot(skips != 0);
                if (mask != 0ull) {
                    int pos = n_skip;
                    if (NW > 1) {  // several wavefronts share the list: one LDS atomic per wavefront and round
                        int first = 0;
                        if (lane == 0) first = atomicAdd(skip_cnt, __popcll(mask));
                        pos = __shfl(first, 0);
                    }
                    pos += __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    if (skips != 0) skip_list[pos] = ((uint32_t)c4 << 4) | skips;
                    n_skip += __popcll(mask);
                }
            }
        } else {
            const uint32_t p = *(const uint32_t*)(src + c4);
            I = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t v = (p >> (8 * q)) & 0xFFu;
                I |= (v == 255u ? 3u : (v == 0u ? 0u : 2u)) << (8 * q);
            }
        }
        // own footprint: class 1 wherever the stamp bit is set and the cell is not 0 (agent.cpp:307-312 skips 0 / 1 / 2)
        const uint32_t stamp = (g_cur >> 20) & 0xFu;
        const uint32_t st = (stamp * 0x00204081u) & 0x01010101u;  // bit q -> bit 8q
        const uint32_t sel = st & (I | (I >> 1));                   // ... and class != 0
        I = (I & ~(sel | (sel << 1))) | sel;
        const uint32_t packed = __builtin_amdgcn_perm(lut_u8, lut_u8, I);
        // two float16 values per v_perm: selector bytes (c, c + 4) pick the low byte out of lut_lo and the high one out of lut_hi
        const uint32_t s01 = __builtin_amdgcn_perm(I, I, 0x01010000u) | 0x04000400u, s23 = __builtin_amdgcn_perm(I, I, 0x03030202u) | 0x04000400u;
        const uint32_t f01 = __builtin_amdgcn_perm(lut_hi, lut_lo, s01), f23 = __builtin_amdgcn_perm(lut_hi, lut_lo, s23);
        if (A4 || c4 + 4 <= NC) {
            *(uint32_t*)(out_u8 + c4) = packed;
            *(uint2*)(out_f16 + c4) = make_uint2(f01, f23);
        } else {
            for (int q = 0; q < 4 && c4 + q < NC; q++) {
                out_u8[c4 + q] = (uint8_t)(packed >> (8 * q));
                out_f16[c4 + q] = (uint16_t)((q < 2 ? f01 : f23) >> (16 * (q & 1)));
            }
        }
    }
    // (5) the cells a top beam left alone: the next lower beam through the cell that writes decides (agent.cpp:555-560), i.e.
    //     the first entry of the cell's static ray list that is not "left alone" again.  Those lists are long next to the
    //     sensor (a cell behind an axis-parallel wall is left alone by most of its beams), so they are cut into chunks of 8
    //     entries and every lane takes one chunk, all 8 loads in flight together:
    //       A  one list entry per lane: every flagged cell that passes the filter below allocates its chunks behind the entry
    //          list (prefix sum over the wavefront) --
    //          a descriptor each (cell, first entry, chunk number, own-footprint bit) -- and, for several chunks, a result
    //          slot (stamped << 30 | list position << 18 | class << 16 | cell; starts as "nobody writes": class 200) in the
    //          dead column table;
    //       B  one chunk per lane: the first deciding entry of the chunk; patched straight into the two output planes for
    //          one-chunk cells (only values other than the provisional 200), else LDS atomicMin into the cell's slot;
    //       C  one slot per lane: patch.
    //     A cell that finds no room (never seen) walks its list alone.
    if (NW > 1) {
        __syncthreads();
        n_skip = skip_cnt[0];
    } else {
        n_skip = __builtin_amdgcn_readfirstlane(n_skip);  // lane 0 ran every round of the loop above
    }
    VIEW_PRIO(0);
    if (n_skip > 0) {
        __builtin_amdgcn_s_waitcnt(0);  // the provisional stores of this wave have landed
        __syncthreads();
        const int n_even = (n_skip + 1) & ~1;
        uint2* desc = (uint2*)(skip_list + n_even);
        uint32_t* slots = (uint32_t*)colt;
        const int cap_d = EXP_RESOLVE_CAP((NCp / 4 - n_even) / 2, 5), cap_r = EXP_RESOLVE_CAP(4 * Wv, 2);  // (test builds: hardly any room)
        const uint2 NOP = make_uint2(0xFFFFFFFFu, 0u);
        const uint32_t h01 = h0 | (h1 << 16);
        EXP_RESOLVE_STATS_CELLS(w, k, hit, skip_list, n_skip, tid, NT);
        int base_d = 0, base_r = 0;  // NW == 1: the two cursors live in scalar registers
        const int n_items = 4 * n_skip;
        for (int tb = 0; tb < n_items; tb += 4 * NT) {  // A: one (list entry, cell of its four) per lane, four rounds' loads in flight
            uint32_t cell_of[4];
            uint2 info[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int t = tb + r * NT + tid;
                cell_of[r] = 0xFFFFFFFFu;
                if (t < n_items) {
                    const uint32_t e = skip_list[t >> 2], q = (uint32_t)t & 3u;
                    if (((e >> q) & 1u) != 0u) cell_of[r] = (e >> 4) + q;
                }
                info[r] = k.inv_cell[min(cell_of[r], (uint32_t)NC - 1u)];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (tb + r * NT >= n_items) break;  // uniform
                const uint32_t c = cell_of[r], f = info[r].x, pk = info[r].y;
                const uint32_t n = (pk >> 20) - 1u, nch = (n + 7u) >> 3;  // entries below the top beam, in chunks of 8
                const bool st = ((f >> 14) & 1u) != 0u;
                // filter: a lower beam decides 255 / 0 only if it gets as far as this cell; when none of the beams of the
                // block around the cell's beams does, whoever writes the cell writes 200, which is what it holds
                const bool pass = c != 0xFFFFFFFFu && nch != 0u && (((f >> 13) & 1u) != 0u || (reach_tab[f & 0x1FFFu] >> 16) >= (f >> 24));
                const uint32_t own = pass ? (nch | (nch > 1u ? 0x10000u : 0u)) : 0u;
                // room for nch descriptors (low half) and, for several chunks, one result slot (high half): a prefix sum
                // over the wavefront instead of one LDS atomic per cell
                const uint32_t incl = wave_prefix_sum(own);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (NW > 1) {
                    int fd = 0, fr = 0;
                    if (lane == 63 && total != 0u) {
                        fd = atomicAdd(&skip_cnt[1], (int)(total & 0xFFFFu));
                        fr = atomicAdd(&skip_cnt[2], (int)(total >> 16));
                    }
                    base_d = __builtin_amdgcn_readlane(fd, 63);
                    base_r = __builtin_amdgcn_readlane(fr, 63);
                }
                if (own != 0u) {
                    const uint32_t excl = incl - own;
                    const int pos = base_d + (int)(excl & 0xFFFFu), slot = base_r + (int)(excl >> 16);
                    const bool multi = nch > 1u;
                    const bool alone = pos + (int)nch > cap_d || (multi && slot >= cap_r);
                    const uint32_t e0 = pk & 0xFFFFFu;
                    if (multi && slot < cap_r) slots[slot] = ((uint32_t)st << 30) | (0xFFFu << 18) | (2u << 16) | c;  // also when it walks alone: C reads every slot
                    const uint32_t lo = c | ((uint32_t)slot << 19) | ((uint32_t)st << 29) | (multi ? 0u : 0x80000000u);
                    for (int j = 0; j < (int)nch && pos + j < cap_d; j++)
                        desc[pos + j] = alone ? NOP : make_uint2(lo | ((min(8u, n - 8u * (uint32_t)j) - 1u) << 16), (e0 + 1u + 8u * (uint32_t)j) | ((uint32_t)j << 20));
                    if (alone) patch_stamped(out_u8, out_f16, h01, h3, c, resolve_skipped_cell(k, hit, c), st);
                }
                if (NW == 1) {
                    base_d += (int)(total & 0xFFFFu);
                    base_r += (int)(total >> 16);
                }
            }
        }
        if (NW == 1 && tid == 0) {
            skip_cnt[1] = base_d;
            skip_cnt[2] = base_r;
        }
        __syncthreads();
        EXP_STOP_AFTER(3);  // ... + step A of the resolve
        const int nd = min(skip_cnt[1], cap_d), nr = min(skip_cnt[2], cap_r);
        EXP_RESOLVE_STATS_ROOM(w, nd, nr, tid);
        for (int t = tid; t < nd; t += NT) {  // B: descriptor = cell | entries - 1 << 16 | slot << 19 | stamped << 29 | one chunk << 31,
            const uint2 d = desc[t];          //                 first entry | chunk << 20
            const uint32_t c = d.x & 0xFFFFu;
            if (c == 0xFFFFu) continue;
            const uint32_t nv = (d.x >> 16) & 7u, first = d.y & 0xFFFFFu, eb = 1u + 8u * (d.y >> 20);
            uint32_t ent[8];
#pragma unroll
            for (int q = 0; q < 8; q++) ent[q] = k.inv_ent[first + min((uint32_t)q, nv)];
            uint32_t key = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t kk = ent[q] & 0xFFFFu, hp = hit[ent[q] >> 16], hk = hp >> 16;
                const uint32_t v = kk < hk ? 3u : (kk == hk ? 0u : 2u);
                const bool decides = !((kk > hk) & (kk <= (hp & 0xFFFFu))) & ((uint32_t)q <= nv);
                key = min(key, decides ? (((eb + q) << 18) | (v << 16)) : 0xFFFFFFFFu);
            }
            if (key != 0xFFFFFFFFu) {
                if ((d.x >> 31) != 0u) patch_stamped(out_u8, out_f16, h01, h3, c, (key >> 16) & 3u, ((d.x >> 29) & 1u) != 0u);
                else atomicMin(&slots[(d.x >> 19) & 0x3FFu], key | c | (((d.x >> 29) & 1u) << 30));
            }
        }
        EXP_STOP_AFTER(4);  // ... + step B
        if (nr > 0) {
            __syncthreads();
            for (int t = tid; t < nr; t += NT) {  // C
                const uint32_t key = slots[t];
                patch_stamped(out_u8, out_f16, h01, h3, key & 0xFFFFu, (key >> 16) & 3u, ((key >> 30) & 1u) != 0u);
            }
        }
    }
    if (tid == 0) w.is_coll[l] = code;
    PHASE_MARK(4);
    WAVE_DONE(0);
}

#include "view_big.h"  // views beyond k_view's packing or shrunk by cv2.resize: k_crop_big, k_beams_big, k_fullview_big

// ------------------------------------------------------------------------------------------------
// Observation + reward / done for local robot l = blockIdx.x, one wavefront.

// Python float floor division (CPython float_floor_div / _float_div_mod)
__device__ double py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod != 0.0) {
        if ((wx < 0) != (mod < 0)) {
            mod += wx;
            div -= 1.0;
        }
    }
    double floordiv;
    if (div != 0.0) {
        floordiv = floor(div);
        if (div - floordiv > 0.5) floordiv += 1.0;
    } else {
        floordiv = copysign(0.0, vx / wx);
    }
    return floordiv;
}

// Agent::get_state (agent.cpp:156-184) of local robot l: the goal in the robot frame (+ heading / speeds)
__device__ __forceinline__ void state_robot(const DevWorld& w, int l) {
    const double* r = w.rec + (size_t)(w.r0 + l) * IMGENV_RECORD_DOUBLES;
    const Tf2 bw = tf_from_pose_sc(r[0], r[1], r[5], r[6]);
    const Tf2 t = tf_mul(w.world_target[l], bw);
    const Tf2 target_base = tf_inverse(t);
    float* vs = w.vector_states + (size_t)l * w.SD;
    vs[0] = (float)target_base.ox;
    vs[1] = (float)target_base.oy;
    if (w.SD == 3) {
        vs[2] = (float)tf_basis_yaw_via_quaternion(target_base);
    } else if (w.SD == 4) {
        vs[2] = (float)w.l0v[l];
        vs[3] = (float)w.l0w[l];
    } else {
        vs[2] = (float)tf_basis_yaw_via_quaternion(target_base);
        vs[3] = (float)w.l0v[l];
        vs[4] = (float)w.l0w[l];
    }
    w.robot_pose[3 * l] = r[0];
    w.robot_pose[3 * l + 1] = r[1];
    w.robot_pose[3 * l + 2] = r[2];
}

__device__ __forceinline__ int tail_robot(const DevWorld& w, int l, int is_reset, int elapsed, int coll, double min_dist) {
    float s0, s1;  // vector_states[0:2], recomputed: whoever writes the state (k_side_robots / k_state) may not have got there yet
    {
        const double* r = w.rec + (size_t)(w.r0 + l) * IMGENV_RECORD_DOUBLES;
        const Tf2 target_base = tf_inverse(tf_mul(w.world_target[l], tf_from_pose_sc(r[0], r[1], r[5], r[6])));
        s0 = (float)target_base.ox;
        s1 = (float)target_base.oy;
    }
    const int arr = w.is_arr[l];
    w.is_collisions[l] = (int8_t)coll;
    w.is_arrives[l] = (uint8_t)arr;
    const double dist = sqrt((double)s0 * (double)s0 + (double)s1 * (double)s1);  // yaml_env.py:467
    const double step_d = is_reset ? 0.0 : w.tmp_dist[l] - dist;
    w.step_ds[l] = step_d;
    w.tmp_dist[l] = dist;
    if (is_reset) {
        w.base_rewards[l] = 0;
        w.base_dones[l] = 0;
        w.py_done[l] = 0;  // self.dones = zeros (yaml_env.py:316)
        w.rewards[l] = 0.0;
        w.paper_rewards[l] = 0.0;
        w.dones[l] = 0;
        w.dones_info[l] = 0;
        w.is_clean[l] = 1;
        w.clean_state[l] = 1;
        if (l == 0) w.counters[0] = 0;
        return 0;
    }
    // ImageEnv.step (yaml_env.py:372-377)
    w.base_rewards[l] = arr - coll;
    int d = (coll > 1 ? 1 : coll) + arr;
    d = d > 1 ? 1 : d;
    w.base_dones[l] = (uint8_t)d;
    w.py_done[l] = (uint8_t)d;
    // TimeLimitWrapper (base.py:222-227)
    const bool timeout = elapsed > w.time_max;
    const int done = timeout ? 1 : d;
    int dinfo = timeout ? 10 : 0;
    // SensorsPaperRewardWrapper._each_r (base.py:164-188)
    double collision_reward = 0, reach_reward = 0, step_reward = 0, distance_reward = 0;
    if (min_dist <= w.ped_safety_space) collision_reward = -50 * (w.ped_safety_space - min_dist);
    if (coll > 0) {
        collision_reward = -500;
    } else {
        if (dist < 0.3 || arr) {
            reach_reward = 500.0;
        } else {
            distance_reward = step_d * 200;
            step_reward = -5;
        }
    }
    double reward = collision_reward + reach_reward + step_reward + distance_reward + 0.0;
    // InfoLogWrapper (base.py:241-254)
    if (coll > 0) dinfo = coll;
    if (arr == 1) dinfo = 5;
    // MultiRobotCleanWrapper (base.py:79-88)
    w.paper_rewards[l] = reward;
    const uint8_t clean_before = w.clean_state[l];
    if (!clean_before) reward = 0;
    w.rewards[l] = reward;
    w.dones[l] = (uint8_t)done;
    w.dones_info[l] = dinfo;
    w.is_clean[l] = clean_before;
    w.step_rewards[l] = reward;  // the same again where a reset behind this step does not reach (imgenv_step_autoreset)
    w.step_dones[l] = (uint8_t)done;
    w.step_dones_info[l] = dinfo;
    w.step_is_clean[l] = clean_before;
    w.step_is_arrives[l] = (uint8_t)arr;
    w.step_is_collisions[l] = (int8_t)coll;
    w.clean_state[l] = done > 0 ? 0 : clean_before;
    if (l == 0) w.counters[0] = elapsed;
    return done;
}

#define PM_CAP 512  // cells of a robot's ped_map that may be non-zero before it falls back to dense clears
#define OBS_DISCS_SERIAL 12  // pedestrians inside a robot's ped_map box up to which k_obs stamps their discs one behind the other

// LDS: (key[PP] f64 sort keys, LDS sort only) | info[P] float2 (px,py) | ord[PP] u16 sorted ped index |
//      inbox[PP] u16 ranks of the pedestrians inside the +-3 m box | stage[64*7] f32, later touched[PM_CAP] u16
// Stable sort of 64 * E (key, index) pairs by (key, index), E pairs per lane in registers: a bitonic network
// whose compare-exchanges run on registers (partner in the same lane) or over lane shuffles (partner lane =
// lane ^ m) -- no LDS traffic, no barriers.  Slot e = lane * E + r; afterwards slot e holds the e-th smallest.
template <int E>
__device__ __forceinline__ void sort_pairs_in_registers(double (&key)[E], uint32_t (&id)[E], int lane) {
#pragma unroll
    for (int size = 2; size <= WAVE * E; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= E) {
                const int lm = stride / E;
                const bool lower = (lane & lm) == 0;
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const bool up = ((lane * E + r) & size) == 0;
                    const double pk = __shfl_xor(key[r], lm);
                    const uint32_t pi = (uint32_t)__shfl_xor((int)id[r], lm);
                    const bool self_gt = (key[r] > pk) | ((key[r] == pk) & (id[r] > pi));
                    const bool take = (lower == up) ? self_gt : !self_gt;  // the lower slot keeps the smaller pair when ascending
                    key[r] = take ? pk : key[r];
                    id[r] = take ? pi : id[r];
                }
            } else {
#pragma unroll
                for (int a = 0; a < E; a++) {
                    const int b = a ^ stride;
                    if (b > a) {
                        const bool up = ((lane * E + a) & size) == 0;
                        const bool gt = (key[a] > key[b]) | ((key[a] == key[b]) & (id[a] > id[b]));
                        const bool sw = gt == up;
                        const double ka = key[a], kb = key[b];
                        const uint32_t ia = id[a], ib = id[b];
                        key[a] = sw ? kb : ka;
                        key[b] = sw ? ka : kb;
                        id[a] = sw ? ib : ia;
                        id[b] = sw ? ia : ib;
                    }
                }
            }
        }
    }
}

// The same network on ONE 64-bit integer per slot: float32(key) bits << 32 | index.  float32 rounding is monotone, so this
// order is the order by (key, index) unless two slots share their float32 surrogate with different float64 keys -- the caller
// looks for equal surrogates in neighbouring slots afterwards and falls back to the exact comparator (about one wavefront per
// step at 8192 robots x 200 pedestrians).  A compare-exchange is one 64-bit compare + two selects (+ two lane shuffles across
// lanes) instead of three compares + three selects (+ three shuffles); two registers per slot instead of three.
template <int E>
__device__ __forceinline__ void sort_packed_in_registers(unsigned long long (&kv)[E], int lane) {
#pragma unroll
    for (int size = 2; size <= WAVE * E; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= E) {
                const int lm = stride / E;
                const bool lower = (lane & lm) == 0;
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const bool up = ((lane * E + r) & size) == 0;
                    const unsigned long long pk = __shfl_xor(kv[r], lm);
                    const bool self_gt = kv[r] > pk;
                    const bool take = (lower == up) ? self_gt : !self_gt;  // the lower slot keeps the smaller value when ascending (equal values: either)
                    kv[r] = take ? pk : kv[r];
                }
            } else {
#pragma unroll
                for (int a = 0; a < E; a++) {
                    const int b = a ^ stride;
                    if (b > a) {
                        const bool up = ((lane * E + a) & size) == 0;
                        const bool gt = kv[a] > kv[b];
                        const bool sw = gt == up;
                        const unsigned long long ka = kv[a], kb = kv[b];
                        kv[a] = sw ? kb : ka;
                        kv[b] = sw ? ka : kb;
                    }
                }
            }
        }
    }
}

// E > 0: PP = 64 * E sort slots held in registers; E == 0: any PP = 2^k, sorted in LDS (more than 1024 pedestrians)
template <int E>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(E <= 8 ? 8 : 4, 8))) void k_obs(DevWorld w, int PP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= act_count_l(w)) return;
    const int l = act_member(w, w.Rw, blockIdx.x), lane = lane_id();
    const int i = w.r0 + l;
    // the pedestrians of this robot's world: indices below are relative to p_lo
    const int P = w.W > 1 ? w.Pw : w.P, p_lo = w.W > 1 ? world_of_robot(w, i) * w.Pw : 0;
    const double *g_ppx = w.ppx + p_lo, *g_ppy = w.ppy + p_lo, *g_pvx = w.pvx + p_lo, *g_pvy = w.pvy + p_lo;
    const double* g_ped_r_round = w.ped_r_round + p_lo;
    const int Hp = w.Hp, Wp = w.Wp, NP = Hp * Wp;
    const int Pa = P > 0 ? P : 1;
    const size_t key_bytes = E == 0 ? (size_t)PP * 8 : 0;
    double* key = (double*)smem;
    float2* info = (float2*)(smem + key_bytes);  // position in the robot frame; the velocity is recomputed where needed
    uint16_t* ord = (uint16_t*)(smem + key_bytes + (size_t)Pa * 8);
    uint16_t* inbox = ord + PP;
    float* stage = (float*)(inbox + PP);
    uint16_t* touched = (uint16_t*)stage;  // the staging buffer is dead by the time discs are stamped

    // The robot's pose of THIS step.  Behind the move: its record.  An early launch (world.h) runs beside the move and advances the
    // robot itself from the snapshot of its record: the command as Agent::cmd takes it (no limiters on such handles), the
    // sines / cosines of the old heading, the new one and its half on three lanes (integrate_heading: k_integrate's own table
    // entries), the exact arc (pose_arc) -- bit for bit what the move writes.  A dead robot keeps its pose (img_env.cpp:392).
    const bool early = w.obs_early != 0;  // (uniform)
    double rx, ry, rsh, rch;
    if (early) {
        const double* sn = w.rec_snap_in + (size_t)l * IMGENV_RECORD_DOUBLES;
        rx = sn[0];
        ry = sn[1];
        rsh = sn[5];
        rch = sn[6];
        if (!w.py_done[l]) {
            const double theta = sn[2];
            const double v = (double)w.obs_actions[3 * l], wv = (double)w.obs_actions[3 * l + 1], v_y = (double)w.obs_actions[3 * l + 2];
            const int n_sub = w.obs_n_sub;
            const double2 tr = integrate_heading(theta, wv, w.step_hz, lane == 0 ? 0 : lane == 1 ? n_sub : n_sub + 1, n_sub);
            const double c0 = __shfl(tr.x, 0), s0 = __shfl(tr.y, 0), c1 = __shfl(tr.x, 1), s1 = __shfl(tr.y, 1);
            rch = __shfl(tr.x, 2);
            rsh = __shfl(tr.y, 2);
            pose_arc(w.ktype == IMGENV_KTYPE_OMNI, v, wv, v_y, w.step_hz, c0, s0, c1, s1, rx, ry);
        }
    } else {
        const double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
        rx = r[0];
        ry = r[1];
        rsh = r[5];
        rch = r[6];
    }
    const Tf2 bw = tf_from_pose_sc(rx, ry, rsh, rch);
    // ... and the pedestrians' state of this step: their arrays behind the move; early, Agent::update (Agent.cpp:840-843) +
    // getNewPosAndVel (rvoscene.h:72-82) applied to the solve's snapshot -- float32 position + velocity * dt, promoted (ped_update_one)
    const float4* g_snap = w.ped_snap_in + p_lo;
    const float ts32 = (float)w.step_hz;
    const bool snap_peds = w.obs_early == 1;  // (2: an early launch beside a crowd whose arrays stand still during the step -- social force, published in front of the move)
    auto ped_pos = [&](int j, double& x, double& y) {
        if (snap_peds) {
            const float4 q = g_snap[j];
            x = (double)(q.x + q.z * ts32);
            y = (double)(q.y + q.w * ts32);
        } else {
            x = g_ppx[j];
            y = g_ppy[j];
        }
    };
    auto ped_vel = [&](int j, double& vx, double& vy) {
        if (snap_peds) {
            const float4 q = g_snap[j];
            vx = (double)q.z;
            vy = (double)q.w;
        } else {
            vx = g_pvx[j];
            vy = g_pvy[j];
        }
    };
    PHASE_BEGIN();
    double min_dist = w.ped_min_dists[l];
    if (P > 0) {
        // PedInfo in the robot base frame, float32 on the wire (img_env.cpp:568-584); sort key yaml_env.py:451
        Tf2 wb = tf_inverse(bw);
        wb.m00 = uniform_f64(wb.m00);  // the same in every lane: scalar registers
        wb.m01 = uniform_f64(wb.m01);
        wb.m10 = uniform_f64(wb.m10);
        wb.m11 = uniform_f64(wb.m11);
        wb.ox = uniform_f64(wb.ox);
        wb.oy = uniform_f64(wb.oy);
        if (E > 0) {
            constexpr int EE = E > 0 ? E : 1;
            unsigned long long kv[EE];  // float32(key) << 32 | index: see sort_packed_in_registers
#pragma unroll
            for (int q = 0; q < EE; q++) {
                const int j = lane + WAVE * q;
                kv[q] = ((unsigned long long)0x7F800000u << 32) | 0xFFFFull;  // padding: +inf, sorts behind everyone
                if (j < P) {
                    double px, py;
                    double gx_, gy_;
                    ped_pos(j, gx_, gy_);
                    tf_apply(wb, gx_, gy_, px, py);
                    const float fx = (float)px, fy = (float)py;
                    info[j] = make_float2(fx, fy);
                    const double key = (double)fx * (double)fx + (double)fy * (double)fy;
                    kv[q] = ((unsigned long long)__float_as_uint((float)key) << 32) | (unsigned long long)(uint32_t)j;
                }
            }
            PHASE_MARK(8);
            // Distances change little from one step to the next, so last step's ORDER is nearly this step's: the slots are filled in
            // that order and a few odd-even transposition passes finish the job -- the result is the one order by (packed key), however
            // it is reached; an order that is still not sorted after w.obs_passes passes (the first step of an episode) goes through the
            // full network.  (E >= 2: the cross-lane pairs of a pass are then disjoint.)
            bool presorted = false;
            if (EE >= 2 && w.obs_ord) {
                const int OBS_PASSES = w.obs_passes;
                __syncthreads();  // every pedestrian's info[] is in LDS
                const uint16_t* prev = w.obs_ord + (size_t)l * (WAVE * EE) + lane * EE;
#pragma unroll
                for (int q = 0; q < EE; q++) {
                    const uint32_t j = prev[q];
                    kv[q] = ((unsigned long long)0x7F800000u << 32) | 0xFFFFull;
                    if (j < (uint32_t)P) {
                        const float2 f = info[j];
                        const double key = (double)f.x * (double)f.x + (double)f.y * (double)f.y;
                        kv[q] = ((unsigned long long)__float_as_uint((float)key) << 32) | (unsigned long long)j;
                    }
                }
                for (int pass = 0; pass <= OBS_PASSES; pass++) {
                    bool ok = true;
#pragma unroll
                    for (int q = 0; q + 1 < EE; q++) ok &= kv[q] <= kv[q + 1];
                    {
                        const unsigned long long nxt = __shfl_down(kv[0], 1);
                        ok &= lane == WAVE - 1 || kv[EE - 1] <= nxt;
                    }
                    if (__all(ok)) {
                        presorted = true;
                        break;
                    }
                    if (pass == OBS_PASSES) break;
#pragma unroll
                    for (int a = 0; a + 1 < EE; a += 2) {  // even pairs
                        const unsigned long long x = kv[a], y = kv[a + 1];
                        kv[a] = x <= y ? x : y;
                        kv[a + 1] = x <= y ? y : x;
                    }
#pragma unroll
                    for (int a = 1; a + 1 < EE; a += 2) {  // odd pairs inside the lane ...
                        const unsigned long long x = kv[a], y = kv[a + 1];
                        kv[a] = x <= y ? x : y;
                        kv[a + 1] = x <= y ? y : x;
                    }
                    {   // ... and the one across the lane boundary: (last slot of lane L, first slot of lane L + 1)
                        const unsigned long long nxt = __shfl_down(kv[0], 1), prv = __shfl_up(kv[EE - 1], 1);
                        const unsigned long long last = kv[EE - 1], first = kv[0];
                        if (lane < WAVE - 1 && last > nxt) kv[EE - 1] = nxt;
                        if (lane > 0 && prv > first) kv[0] = prv;
                    }
                }
            }
            if (!presorted) sort_packed_in_registers<EE>(kv, lane);
            // two neighbouring slots with one surrogate (and a pedestrian in the later one): their float64 keys may differ
            uint32_t clash = 0;  // bit q: slots lane * E + q and its successor
#pragma unroll
            for (int q = 0; q + 1 < EE; q++)
                clash |= ((uint32_t)(kv[q] >> 32) == (uint32_t)(kv[q + 1] >> 32) && (uint32_t)kv[q + 1] != 0xFFFFu) ? (1u << q) : 0u;
            {
                const unsigned long long nxt = __shfl_down(kv[0], 1);
                clash |= (lane < WAVE - 1 && (uint32_t)(kv[EE - 1] >> 32) == (uint32_t)(nxt >> 32) && (uint32_t)nxt != 0xFFFFu) ? (1u << (EE - 1)) : 0u;
            }
#pragma unroll
            for (int q = 0; q < EE; q++) ord[lane * EE + q] = (uint16_t)(uint32_t)kv[q];
            __syncthreads();
            if (__builtin_expect(__any(clash != 0u), 0)) {
                // rare.  Slots with different surrogates are in their final order (rounding is monotone); inside a run of equal
                // surrogates the exact (float64 key, index) order is restored by odd-even transposition over the flagged
                // neighbours, until nothing moves (runs are two or three slots long)
                bool again = true;
                while (again) {
                    bool swapped = false;
                    for (int parity = 0; parity < 2; parity++) {
                        for (int q = 0; q < EE; q++) {
                            const int e = lane * EE + q;
                            if (((clash >> q) & 1u) != 0u && (e & 1) == parity) {
                                const uint32_t ja = ord[e], jb = ord[e + 1];
                                const float2 fa = info[ja], fb = info[jb];
                                const double ka = (double)fa.x * (double)fa.x + (double)fa.y * (double)fa.y;
                                const double kb = (double)fb.x * (double)fb.x + (double)fb.y * (double)fb.y;
                                if (ka > kb || (ka == kb && ja > jb)) {
                                    ord[e] = (uint16_t)jb;
                                    ord[e + 1] = (uint16_t)ja;
                                    swapped = true;
                                }
                            }
                        }
                        __syncthreads();
                    }
                    again = __any(swapped);
                }
            }
            if (EE >= 2 && w.obs_ord) {  // next step starts from this order
                uint16_t* keep = w.obs_ord + (size_t)l * (WAVE * EE) + lane * EE;
#pragma unroll
                for (int q = 0; q < EE; q++) keep[q] = ord[lane * EE + q];
            }
        } else {
            for (int j = lane; j < PP; j += WAVE) {
                if (j < P) {
                    double px, py;
                    double gx_, gy_;
                    ped_pos(j, gx_, gy_);
                    tf_apply(wb, gx_, gy_, px, py);
                    const float fx = (float)px, fy = (float)py;
                    info[j] = make_float2(fx, fy);
                    key[j] = (double)fx * (double)fx + (double)fy * (double)fy;
                    ord[j] = (uint16_t)j;
                } else {
                    key[j] = __builtin_huge_val();
                    ord[j] = 0xFFFF;
                }
            }
            __syncthreads();
            PHASE_MARK(8);
            // stable sort by (key, index): bitonic network over PP = 2^k entries in LDS
            for (int kk = 2; kk <= PP; kk <<= 1) {
                for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                    for (int t2 = lane; t2 < PP / 2; t2 += WAVE) {
                        const int a = ((t2 / jj) * 2 * jj) + (t2 % jj);
                        const int b = a + jj;
                        const bool up = ((a & kk) == 0);
                        const double ka = key[a], kb = key[b];
                        const uint16_t oa = ord[a], ob = ord[b];
                        const bool a_gt_b = (ka > kb) || (ka == kb && oa > ob);
                        if (a_gt_b == up) {
                            key[a] = kb;
                            key[b] = ka;
                            ord[a] = ob;
                            ord[b] = oa;
                        }
                    }
                    __syncthreads();
                }
            }
        }
        PHASE_MARK(9);
        // ped_tmp vector (yaml_env.py:397-408): 7 floats per pedestrian in rank order, staged through LDS so
        // that the global stores are contiguous; pedestrians inside the +-3 m box are collected in rank order
        float* pt = w.ped_vector_states + (size_t)l * w.PV;
        if (lane == 0) pt[0] = (float)P;
        const double rsl = w.robot_size_last[i];
        int n_in = 0;
        for (int q0 = 0; q0 < P; q0 += WAVE) {
            const int q = q0 + lane;
            bool in_box = false;
            if (q < P) {
                const int j = ord[q];
                const float2 f = info[j];
                double pvx, pvy;  // PedInfo velocity in the robot frame (img_env.cpp:576-580)
                ped_vel(j, pvx, pvy);
                const float fvx = (float)((wb.m00 * pvx + wb.m01 * pvy) + 0.0), fvy = (float)((wb.m10 * pvx + wb.m11 * pvy) + 0.0);
                const double dpx = f.x, dpy = f.y;
                const double ped_r = g_ped_r_round[j];
                const float dist = (float)sqrt(dpx * dpx + dpy * dpy);
                // seven dwords per lane, 28 bytes from the neighbour lane's: the seven store instructions of a round fill
                // the same cache lines between them (no staging through LDS, no barriers)
                float* o = pt + 1 + 7 * (size_t)q;
                o[0] = f.x;
                o[1] = f.y;
                o[2] = fvx;
                o[3] = fvy;
                o[4] = (float)ped_r;
                o[5] = (float)(ped_r + rsl);
                o[6] = dist;
                if (q == 0) min_dist = (double)(float)(dist - (float)(ped_r + rsl));  // yaml_env.py:455-456
                in_box = !(dpx > 3 || dpx < -3 || dpy > 3 || dpy < -3);          // yaml_env.py:409-410
            }
            const unsigned long long mask = __ballot(in_box);
            if (in_box) inbox[n_in + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)q;
            n_in += __popcll(mask);
        }
        __syncthreads();
        min_dist = __shfl(min_dist, 0);
        PHASE_MARK(10);

        // ped_map (yaml_env.py:409-427), sparse update: the map is zero except under a few discs, so instead of
        // rewriting 3 x Hp x Wp floats per robot per step, the cells written last step are cleared and the discs
        // of the pedestrians inside the box are stamped in rank order (farther ones overwrite nearer ones).
        float* pm = w.ped_maps + (size_t)l * 3 * NP;
        uint16_t* prev = w.pm_cells + (size_t)l * PM_CAP;
        const int n_prev = w.pm_n[l];
        if (n_prev >= 0) {
            for (int e = lane; e < n_prev; e += WAVE) {
                const int c = prev[e];
                pm[c] = 0.0f;
                pm[NP + c] = 0.0f;
                pm[2 * NP + c] = 0.0f;
            }
        } else {  // unknown contents (first use or overflow): dense clear
            for (int c4 = lane * 4; c4 < 3 * NP; c4 += WAVE * 4) {
                if (c4 + 4 <= 3 * NP) {
                    *(float4*)(pm + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    for (int q = c4; q < 3 * NP; q++) pm[q] = 0.0f;
                }
            }
        }
        int n_new = 0;
        const double pres = w.ped_res, pinv = w.ped_inv_res, pr = w.ped_image_r, pr2 = w.ped_image_r2;
        // A robot that stands IN a crowd has dozens of pedestrians inside its box (cfg-4: up to 200 around the robots near the crowd's
        // 10 m square), and discs one behind the other -- each waits for the last one's stores -- made those few wavefronts the
        // kernel's tail: 93 us where the others take 30.  "Later discs overwrite earlier ones" = every cell ends with its HIGHEST-ranked
        // disc, so: (A) a lane per disc puts its rank on its cells with atomicMax (plane 0 of the map holds the rank for a moment:
        // cleared to 0 above, ranks 1 .. n_in, never the bits of 1.0f); (B) the lanes walk their discs again and the one whose rank a
        // cell holds writes the cell's three values and lists it.  No order between discs is needed any more.  The same map, bit for bit.
        const bool discs_in_parallel = n_in > OBS_DISCS_SERIAL;
        if (discs_in_parallel) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the clears above are out before the first atomic lands on them)
            __builtin_amdgcn_wave_barrier();
            uint32_t* rank_plane = (uint32_t*)pm;
            for (int e0 = 0; e0 < n_in; e0 += WAVE) {  // uniform trip counts (ballots inside)
                const int e = e0 + lane;
                const bool has = e < n_in;
                const int je = ord[inbox[has ? e : 0]];
                const float2 f = info[je];
                double evx, evy;
                ped_vel(je, evx, evy);
                const float fvx = (float)((wb.m00 * evx + wb.m01 * evy) + 0.0), fvy = (float)((wb.m10 * evx + wb.m11 * evy) + 0.0);
                const double tmx = -(double)f.x + 3, tmy = -(double)f.y + 3;
                int ax, bx, ay, by;
                if (pinv != 0.0) {
                    ax = (int)floor((tmx - pr) * pinv);
                    bx = (int)floor((tmx + pr) * pinv);
                    ay = (int)floor((tmy - pr) * pinv);
                    by = (int)floor((tmy + pr) * pinv);
                } else {
                    ax = (int)py_floordiv(tmx - pr, pres);
                    bx = (int)py_floordiv(tmx + pr, pres);
                    ay = (int)py_floordiv(tmy - pr, pres);
                    by = (int)py_floordiv(tmy + pr, pres);
                }
                const int wy = by - ay, cnt = has ? (bx - ax) * wy : 0;
                int cnt_max = cnt;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) cnt_max = max(cnt_max, __shfl_xor(cnt_max, off));
                auto cell_of = [&](int tt, int& c) -> bool {  // the tt-th cell of this lane's disc's bounding box: inside the map and the disc?
                    if (tt >= cnt) return false;
                    const int a = tt / wy, jj = ax + a, kq = ay + (tt - a * wy);
                    if (!(jj >= 0 && jj < Hp && kq >= 0 && kq < Wp)) return false;
                    const double ddx = (jj + 0.5) * pres - tmx, ddy = (kq + 0.5) * pres - tmy;
                    c = jj * Wp + kq;
                    return ddx * ddx + ddy * ddy < pr2;
                };
                for (int tt = 0; tt < cnt_max; tt++) {  // (A)
                    int c = 0;
                    if (cell_of(tt, c)) atomicMax(&rank_plane[c], (uint32_t)(e + 1));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (every rank has landed: the atomics are performed in L2)
            __builtin_amdgcn_wave_barrier();
            for (int e0 = 0; e0 < n_in; e0 += WAVE) {
                const int e = e0 + lane;
                const bool has = e < n_in;
                const int je = ord[inbox[has ? e : 0]];
                const float2 f = info[je];
                double evx, evy;
                ped_vel(je, evx, evy);
                const float fvx = (float)((wb.m00 * evx + wb.m01 * evy) + 0.0), fvy = (float)((wb.m10 * evx + wb.m11 * evy) + 0.0);
                const double tmx = -(double)f.x + 3, tmy = -(double)f.y + 3;
                int ax, bx, ay, by;
                if (pinv != 0.0) {
                    ax = (int)floor((tmx - pr) * pinv);
                    bx = (int)floor((tmx + pr) * pinv);
                    ay = (int)floor((tmy - pr) * pinv);
                    by = (int)floor((tmy + pr) * pinv);
                } else {
                    ax = (int)py_floordiv(tmx - pr, pres);
                    bx = (int)py_floordiv(tmx + pr, pres);
                    ay = (int)py_floordiv(tmy - pr, pres);
                    by = (int)py_floordiv(tmy + pr, pres);
                }
                const int wy = by - ay, cnt = has ? (bx - ax) * wy : 0;
                int cnt_max = cnt;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) cnt_max = max(cnt_max, __shfl_xor(cnt_max, off));
                for (int tt = 0; tt < cnt_max; tt++) {  // (B)
                    bool mine = false;
                    int c = 0;
                    if (tt < cnt) {
                        const int a = tt / wy, jj = ax + a, kq = ay + (tt - a * wy);
                        if (jj >= 0 && jj < Hp && kq >= 0 && kq < Wp) {
                            const double ddx = (jj + 0.5) * pres - tmx, ddy = (kq + 0.5) * pres - tmy;
                            c = jj * Wp + kq;
                            // (a cell another lane has finished already holds the bits of 1.0f: nobody's rank)
                            mine = ddx * ddx + ddy * ddy < pr2 &&
                                   __hip_atomic_load(&rank_plane[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)(e + 1);
                        }
                    }
                    if (mine) {
                        pm[c] = 1.0f;
                        pm[NP + c] = fvx;
                        pm[2 * NP + c] = fvy;
                    }
                    const unsigned long long mask = __ballot(mine);
                    const int pos = n_new + __popcll(mask & ((1ull << lane) - 1ull));
                    if (mine && pos < PM_CAP) touched[pos] = (uint16_t)c;
                    n_new += __popcll(mask);
                }
            }
        }
        for (int e = 0; e < (discs_in_parallel ? 0 : n_in); e++) {
            __syncthreads();  // drains this wave's earlier stores: later discs overwrite earlier ones
            const int je = ord[inbox[e]];
            const float2 f = info[je];
            double evx, evy;
            ped_vel(je, evx, evy);
            const float fvx = (float)((wb.m00 * evx + wb.m01 * evy) + 0.0), fvy = (float)((wb.m10 * evx + wb.m11 * evy) + 0.0);
            const double tmx = -(double)f.x + 3, tmy = -(double)f.y + 3;
            int ax, bx, ay, by;
            if (pinv != 0.0) {  // power-of-two cell: Python's v // res is floor(v * (1 / res)) exactly, without the fmod
                ax = (int)floor((tmx - pr) * pinv);
                bx = (int)floor((tmx + pr) * pinv);
                ay = (int)floor((tmy - pr) * pinv);
                by = (int)floor((tmy + pr) * pinv);
            } else {
                ax = (int)py_floordiv(tmx - pr, pres);
                bx = (int)py_floordiv(tmx + pr, pres);
                ay = (int)py_floordiv(tmy - pr, pres);
                by = (int)py_floordiv(tmy + pr, pres);
            }
            const int wy = by - ay, cnt = (bx - ax) * wy;
            for (int t0 = 0; t0 < cnt; t0 += WAVE) {
                const int tt = t0 + lane;
                bool hitc = false;
                int c = 0;
                if (tt < cnt) {
                    const int jj = ax + tt / wy, kq = ay + tt % wy;
                    if (jj >= 0 && jj < Hp && kq >= 0 && kq < Wp) {
                        const double ddx = (jj + 0.5) * pres - tmx, ddy = (kq + 0.5) * pres - tmy;
                        hitc = ddx * ddx + ddy * ddy < pr2;
                        c = jj * Wp + kq;
                    }
                }
                if (hitc) {
                    pm[c] = 1.0f;
                    pm[NP + c] = fvx;
                    pm[2 * NP + c] = fvy;
                }
                const unsigned long long mask = __ballot(hitc);
                const int pos = n_new + __popcll(mask & ((1ull << lane) - 1ull));
                if (hitc && pos < PM_CAP) touched[pos] = (uint16_t)c;
                n_new += __popcll(mask);
            }
        }
        __syncthreads();
        if (n_new <= PM_CAP) {
            for (int e = lane; e < n_new; e += WAVE) prev[e] = touched[e];
            if (lane == 0) w.pm_n[l] = n_new;
        } else if (lane == 0) {
            w.pm_n[l] = -1;
        }
    }
    PHASE_MARK(11);
    WAVE_DONE(1);
    if (lane == 0) w.ped_min_dists[l] = min_dist;
    tail_arrive_obs(w, blockIdx.x, l, min_dist);
}

// Agent::get_state of the local robots after a reset of worlds without pedestrians (with pedestrians k_side_robots does it on
// the side stream; in the steps of pedestrian-free worlds k_integrate does): its correctly rounded atan2 is a long serial chain
// and needs many registers, so it stays out of the kernels that run the tails.
__global__ void k_state(DevWorld w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < act_count_l(w)) state_robot(w, act_member(w, w.Rw, t));
}

// Per-robot work that needs the new poses only, on the side stream beside the rasters and the view:
//  * _step_robot tail for the RVO scenes: setRobotPos for every robot (img_env.cpp:411-417, rvoscene.h:47-51), straight
//    from the gathered robot records, so that the solve does not have to wait for the raster;
//  * Agent::get_state of the local robots (its correctly rounded atan2 is a long serial chain that the tails, on the
//    critical path, would otherwise run).
//
// One wavefront per 64 robots and SLICE of the pedestrians (`slices` workgroups per robot block; slice 0 also writes the agent
// arrays and runs get_state): the solve waits for this kernel and the step's join for the solve -- with the two event hops of the
// join that chain ends later than the views' -- so it is spread over the chip instead of 4 wavefronts on each of 32 compute units
// walking every pedestrian one after the other (8192 robots x 200 pedestrians: 24-32 us beside k_obs; x 1000: 109 us).
#define SIDE_PED_TILE 1024
__global__ __launch_bounds__(WAVE) void k_side_robots(DevWorld w, int zero_vel, int rvo_agents, int slices) {
    __shared__ float2 ped_xy[SIDE_PED_TILE];
    __builtin_amdgcn_s_setprio(2);  // (on the solve's chain: ahead of the observation's wavefronts, behind the move and the rasters)
    const int rb = (int)blockIdx.x / slices, slice = (int)blockIdx.x - rb * slices;
    const int t = rb * WAVE + threadIdx.x;
    const bool valid = t < act_count_g(w);
    const int i = act_member(w, w.Rw, valid ? t : 0);
    if (rvo_agents) {
        const double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
        const int a = w.P + i;
        const f2 me = F2((float)r[0], (float)r[1]);
        if (valid && slice == 0) {
            w.apx[a] = me.x;
            w.apy[a] = me.y;
            w.avx[a] = zero_vel ? 0.0f : (float)r[3];
            w.avy[a] = zero_vel ? 0.0f : (float)r[4];
        }
        // neighborDist is 0.5 m (rvoscene.h:57): tell the few pedestrians this robot can matter to, so that the solve does
        // not scan every robot of the world for every pedestrian.  The test is the solve's own float expression with a
        // slightly larger bound; the solve re-tests exactly.
        if (w.W > 1) {  // several small worlds: every robot walks its own world's pedestrians (one slice)
            const int p_lo = world_of_robot(w, i) * w.Pw;
            for (int j = p_lo; valid && slice == 0 && j < p_lo + w.Pw; j++)
                if (abs_sq(F2(w.apx[j], w.apy[j]) - me) < 0.2500001f) {
                    const int pos = atomicAdd(&w.near_n[j], 1);
                    if (pos < ORCA_NEAR_CAP) w.near_list[(size_t)j * ORCA_NEAR_CAP + pos] = a;
                }
        } else {  // one big world: this slice's pedestrian positions go through LDS, a tile at a time
            const int per = (w.P + slices - 1) / slices, j_lo = slice * per, j_hi = min(w.P, j_lo + per);
            for (int j0 = j_lo; j0 < j_hi; j0 += SIDE_PED_TILE) {
                const int nt = min(SIDE_PED_TILE, j_hi - j0);
                __syncthreads();
                for (int q = threadIdx.x; q < nt; q += blockDim.x) ped_xy[q] = make_float2(w.apx[j0 + q], w.apy[j0 + q]);
                __syncthreads();
                if (valid) {
                    for (int q = 0; q < nt; q++) {
                        const float2 pp = ped_xy[q];
                        if (abs_sq(F2(pp.x, pp.y) - me) < 0.2500001f) {
                            const int pos = atomicAdd(&w.near_n[j0 + q], 1);
                            if (pos < ORCA_NEAR_CAP) w.near_list[(size_t)(j0 + q) * ORCA_NEAR_CAP + pos] = a;
                        }
                    }
                }
            }
        }
    }
    const int l = i - w.r0;
    if (valid && slice == 0 && l >= 0 && l < w.RL) state_robot(w, l);
}

// NeverStopWrapper's question (base.py:198-211): are all robots of a world done?  One workgroup walks the worlds, a thread
// each; the worlds that are go, unordered, into page-locked host memory: finished[1..] = which, finished[0] = how many.
__global__ __launch_bounds__(1024) void k_finished(DevWorld w) {
    __shared__ int n_sh;
    if (threadIdx.x == 0) n_sh = 0;
    __syncthreads();
    for (int k0 = 0; k0 < w.W; k0 += blockDim.x) {  // uniform trip count (ballot inside)
        const int k = k0 + threadIdx.x;
        bool all_done = k < w.W;
        if (all_done)
            for (int q = 0; q < w.Rw && all_done; q++) all_done = w.dones[(size_t)k * w.Rw + q] != 0;
        if (k < w.W)
            for (int q = 0; q < w.Rw; q++) w.step_all_down[(size_t)k * w.Rw + q] = all_done ? 1 : 0;
        const unsigned long long mask = __ballot(all_done);
        int first = 0;
        if (mask != 0ull && lane_id() == 0) first = atomicAdd(&n_sh, __popcll(mask));
        first = __shfl(first, 0);
        if (all_done) w.finished[1 + first + __popcll(mask & ((1ull << lane_id()) - 1ull))] = k;
    }
    __syncthreads();
    if (threadIdx.x == 0) w.finished[0] = n_sh;
}

// Per-robot scalars, one lane per robot of a group of 64 consecutive active-list positions: Agent::get_state's first two
// components (agent.cpp:156-184), the _get_states distances, ImageEnv.step and the wrapper stack (reward / done).  Run by the
// k_view / k_obs wavefront that completes the group (tail_count).  The collision code and the pedestrian distance of each robot
// are in its exchange word, read at device scope (the wavefronts that wrote them ran on other compute units).
__device__ __forceinline__ void tail_group(const DevWorld& w, int g) {
    const int t = (g << 6) + lane_id();
    const bool valid = t < act_count_l(w);
    const int l = act_member(w, w.Rw, valid ? t : 0);
    const int is_reset = w.tail_is_reset;
    int done = 0;
    // robots whose view was frozen this step (agent.cpp:358-360: collided before this step, or arrived): the collision code
    // the previous step published, the arrive flag this step's integrate left
    const bool frozen = valid && !is_reset && (w.is_collisions[l] != 0 || w.is_arr[l] != 0);
    if (valid) {
        const unsigned long long sig = __hip_atomic_load(&w.tail_sig[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double min_dist = w.tail_fused ? (double)__uint_as_float((uint32_t)(sig >> 32)) : w.ped_min_dists[l];
        // TimeLimitWrapper counts per world: steps since that world's last reset
        done = tail_robot(w, l, is_reset, tail_elapsed_of(w) - w.world_epoch[world_of_robot(w, w.r0 + l)], (int)((sig >> 8) & 0xFFull), min_dist);
        w.tail_sig[l] = 0ull;  // for the next chain of launches
    }
    if (lane_id() == 0) w.tail_cnt[g * TAIL_CNT_STRIDE] = 0;
    if (w.sharded && g == 0 && lane_id() < 4)  // the rasters and k_compose of this step (its only readers) are long done with the box: re-arm it
        w.bbox[lane_id()] = lane_id() < 2 ? BBOX_INIT_MIN : BBOX_INIT_MAX;
    const unsigned long long mask = __ballot(done > 0);  // counters[1] = robots done this step, one atomic per group
    if (mask != 0 && lane_id() == 0) atomicAdd(&w.counters[1], __popcll(mask));
    const unsigned long long fmask = __ballot(frozen);  // counters[2] since the last reset, counters[3] since create
    if (fmask != 0 && lane_id() == 0) {  // two 32-bit adds: a handle that only ever resets single worlds never clears counters[2],
        const int n = __popcll(fmask);   // and a carry out of it must not leak into counters[3]
        atomicAdd(&w.counters[2], n);
        atomicAdd(&w.counters[3], n);
    }
}

