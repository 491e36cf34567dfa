// view_big.h -- Agent::view (agent.cpp:356-509) for views beyond k_view's 16 / 8-bit packing, and for views that
// cv2.resize INTER_CUBIC shrinks (yaml_env.py:431-438).  Every shipped config of the reference is one: 400 x 400 cells of
// 0.015 m, 1000 beams of up to 283 steps, a 48 x 48 sensor_map (envs/cfg/test.yaml:54, 126-132).
//
// A view of 160 000 cells is far too much for one wavefront, and a launch of a few hundred robots (one robot per env,
// a few hundred envs) has to fill 256 compute units, so the work of ONE robot is spread over the chip:
//
//   k_crop_big     1 wavefront / 8 x 8 tile of view cells inside the field of view (static list per class), any number of
//                  tiles per robot in flight.  The view -> grid transform of a cell is two fused multiply-adds per axis on the
//                  cell's integer coordinates (the reference's chain of roundings only matters within 1e-6 of a rounding
//                  boundary; those lanes redo it literally), one gather of the class layer, and the tile's 64 verdicts leave
//                  as ONE ballot: the cropped view is a bitmap of 20 KB per robot in HBM instead of 160 KB of bytes.
//   k_beams_big    1 workgroup of 256 / robot and 256 beams: the bitmap into LDS; one beam per thread walks its static path
//                  (coalesced 16-byte loads of bit addresses, 32 steps in flight) to its first occupied cell -> hit word
//                  (first-hit step << 16 | last step behind it in the hit cell's row or column, as in k_view) and lasers;
//                  the first workgroup of a robot also takes the collision code and hands it to the step's tail.
//   k_taps_big     1 thread / pixel of a shrunk sensor_map: ONLY the 4 x 4 source cells of the pixel are evaluated (static
//                  tap records) and run through OpenCV's fixed-point bicubic -- the full view is never written unless asked for.
//   k_fullview_big 1 thread / 4 cells, only when the full view is an output (imgenv_out.view_maps, or a big view that is not
//                  shrunk): laser_map + own footprint of every cell from the hit words.
//
// laser_map semantics as in k_view: a cell ends with the verdict of the HIGHEST beam through it that writes it (static
// top entry per cell; where that beam leaves the cell alone, agent.cpp:555-560, the cell's static ray list is walked).  A cell
// under the own footprint becomes 100 unless it is 0 (agent.cpp:307-312), and it can only be 0 if the crop found it
// occupied: such cells skip the beams altogether, which also keeps the walks away from the long ray lists around the sensor.
#pragma once

#define VBC_T 256    // k_crop_big: up to 4 wavefronts
#define VBC_U 4      // tiles a wavefront keeps in flight (8: measured slower, 321 against 304 us for 2048 shipped views)
#define VBB_T 256    // k_beams_big: one beam per thread
#define VBT_T 256    // k_taps_big: one sensor_map pixel per thread
#define VBF_T 256    // k_fullview_big

__device__ __forceinline__ long long uniform_i64(long long v) {  // a value known to be the same in every lane, into scalar registers
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(unsigned long long)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// lane q (wave-uniform) of v becomes the wave-uniform value x  (m0 is named as clobbered on purpose -- the compiler may keep a
// value of its own there -- which clang reports as "reserved register on the clobber list")
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void write_lane(uint32_t& v, uint32_t x, int q) {
    const uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(xs), "s"(q) : "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ uint32_t crop_tiled_at(uint32_t wt, uint32_t m, uint32_t n) {  // crop_tiled (kernels.h) with the pitch in a register
    // = ((m >> 3) * wt + (n >> 3)) * 64 + (m & 7) * 8 + (n & 7), in six instructions: m << 3 already is (m >> 3) * 64 + (m & 7) * 8
    const uint32_t n6 = ((n & ~7u) << 3) | (n & 7u);  // (n >> 3) * 64 + (n & 7): v_and, v_lshlrev, v_and_or
    return (m << 3) + (__umul24(m >> 3, (wt - 1u) << 6) + n6);  // v_lshrrev + v_mad_u32_u24 + v_lshl_add
}

// ------------------------------------------------------------------------------------------------
// (2) egocentric crop (agent.cpp:373-404), tiled.  All workgroups of one robot run on the same XCD (blockIdx modulo 8), so the
// robot's window of the map is fetched from HBM once and shared through that XCD's L2.  STAMP mode reads the obstacle map
// itself (one byte per cell) and the class layer's word only inside 64-cell segments some raster stamped this step (seg_tag).
// TILED (STAMP mode with w.crop_map): ONE byte gather per cell from the map's 8 x 8-block copy (free bit + crop tag of the last
// stamp, world.h) -- 4-6 half cache lines per tile whatever the heading instead of a dozen map rows + the segment tags.
template <bool STAMP, bool TILED>
__global__ __launch_bounds__(VBC_T) void k_crop_big(DevWorld w, int chunks, int n_robots, int tpw) {
    // block g: XCD g % 8 takes robots g % 8, g % 8 + 8, ... with all their chunks
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int t = (slot / chunks) * 8 + xcd, chunk = slot % chunks;
    if (t >= n_robots || t >= act_count_l(w)) return;
    const int l = act_member(w, w.Rw, t);
    const bool frozen = w.is_coll[l] || w.is_arr[l];  // the view keeps its last value (agent.cpp:358-360)
    // the kernels behind this one go by this word: is_collision_ itself changes underneath them (committed by the last one)
    if (chunk == 0 && threadIdx.x == 0) w.big_hit[(size_t)l * w.big_hit_stride + w.B + 1] = frozen ? 0u : 1u;
    if (frozen) return;
    const int i = w.r0 + l;
    const BigClassDev k = w.big[__builtin_amdgcn_readfirstlane(w.robot_cls[i])];
    const double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
    const Tf2 bw = tf_from_pose_sc(r[0], r[1], r[5], r[6]);
    const Tf2 vw = tf_mul(bw, w.view_base);  // get_view_world (agent.cpp:128-131)
    const double res = w.res;
    // cell index = round(((m00 (a res) + m01 (b res)) + ox) / res): in cells that is m00 a + m01 b + ox / res up to a few ulps.
    // Every multiply, shift, convert or fp64 operation costs a SIMD 4 cycles per wavefront, an integer add 2.4
    // (tools/micro/valu_issue.hip), so the fast path is 32.32 fixed point: the tile's corner term is wave-uniform (scalar unit),
    // a lane adds its own constant offset -- one 64-bit add per axis, the cell index is the high word (0.5 is folded in), and
    // the low word says how close to a rounding boundary the value is.  The coefficients carry 2^-33 of rounding error each:
    // with view coordinates below 4096 the sum is within 2^-20 of the exact value, the reference's own fp64 chain within 1e-11;
    // lanes within 2^-17 of a boundary take the literal chain.
    const double oxs = vw.ox * w.inv_res, oys = vw.oy * w.inv_res;
    const double two32 = 4294967296.0;
    // (wave-uniform values, moved to scalar registers so that the per-tile products run on the scalar unit)
    const long long M00 = uniform_i64((long long)rint(vw.m00 * two32)), M01 = uniform_i64((long long)rint(vw.m01 * two32));
    const long long M10 = uniform_i64((long long)rint(vw.m10 * two32)), M11 = uniform_i64((long long)rint(vw.m11 * two32));
    const long long OX = uniform_i64((long long)rint(oxs * two32) + (1ll << 31)), OY = uniform_i64((long long)rint(oys * two32) + (1ll << 31));
    const bool fixed_ok = fabs(oxs) < 1048576.0 && fabs(oys) < 1048576.0;  // (always, for a pose anywhere near its map)
    const int Hg = w.Hg, Wg = w.Wg;
    const uint32_t self = (uint32_t)i, tag = STAMP ? stamp_tag_of(w) : 0u;
    const uint32_t free_own = STAMP ? (CLS_HIGH | (STAMP_ONE << STAMP_KIND_SHIFT) | (tag << STAMP_TAG_SHIFT) | (self << STAMP_OWNER_SHIFT))
                                    : (CLS_HIGH | CLS_ROBOT | (self << 8));
    const uint32_t base_tag_mask = 7u | (0xFFu << STAMP_TAG_SHIFT), base_tag_ours = CLS_HIGH | (tag << STAMP_TAG_SHIFT);
    const int world = world_of_robot(w, i);
    const uint32_t cell0 = (uint32_t)world * w.Gs, last_cell = (uint32_t)(Hg * Wg - 1);
    const uint8_t* obs = TILED ? w.crop_map + (size_t)world * w.crop_ws : w.obs_map + cell0;
    const uint32_t ctag = crop_tag_of(tag), Wt = w.crop_wt, last_byte = w.crop_ws - 1u;
    const uint32_t* cls = w.cell + cell0;
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int da = lane >> 3, db = lane & 7;
    long long Lx = da * M00 + db * M01, Ly = da * M10 + db * M11;  // this lane's offset inside any tile
    asm volatile("" : "+v"(Lx), "+v"(Ly));  // (opaque: keeps the compiler from folding it back into a per-lane 64-bit multiply per tile)
    unsigned long long* plane0 = (unsigned long long*)(w.big_bits + (size_t)l * 2 * w.big_words);
    unsigned long long* plane1 = (unsigned long long*)(w.big_bits + ((size_t)l * 2 + 1) * w.big_words);
    const bool want_unknown = w.use_laser == 0;  // with the laser on only "occupied or not" survives into the outputs
    const int n_crop = k.n_crop, tb_n = k.tb;
    const int nw = (int)blockDim.x >> 6;  // wavefronts of this workgroup: 4, or 1 where the robots alone fill the chip
    const int first = chunk * (nw * tpw);  // tpw tiles per wavefront: more in big launches (workgroup dispatch has a price)
    // lane q prepares this wavefront's q-th tile (tpw <= 64): its word in the bitmap stays in the lane (the tile's result comes
    // back to it by v_writelane, one store per wavefront at the end), the corner term of the transform goes through a
    // wavefront-private LDS record that every lane reads back with one broadcast ds_read_b128 per tile; the field-of-view masks
    // of a round's tiles come in one scalar load.
    // How it got here (2048 shipped views, us per launch): 323 with the corner products on the scalar unit (~28 scalar
    // instructions per tile; a compute unit issues one per cycle, tools/micro/gather_rate.hip) -> 304 with the LDS records.
    // Parts switched off one at a time then said: 155 of instruction issue (~40 vector + scalar instructions per tile) + 72 for
    // the gather of map bytes + 44 for the segment tags + 20 for the stores, adding up instead of overlapping, the gathers'
    // price following the cache lines a tile touches (231 with every robot heading along the map's rows, 336 at 45 degrees).
    // -> 196 with crop_map (one byte gather per cell, 8 x 8 blocks: world.h) -> 151 with lane masks kept in scalar registers
    // (a vector load in the rare literal path had turned every mask of the loop into a vector register pair), the guard band
    // folded into the corner, the tiled index in six instructions, masks by scalar load: 17 vector instructions per tile.
    // Measured and dropped: 8 tiles in flight (321 against 304), a per-region LDS window of the map loaded as aligned dwords
    // (400-540: a rotated region's bounding box is twice its area -- more lines per tile than the gathers it replaces).
    __shared__ longlong2 corner[VBC_T / WAVE][WAVE];
    // The corner carries the guard band too: with 2^15 added, "within 2^-17 of a rounding boundary" reads low word < 2^16, and
    // the high word is the cell wherever that test says the fast path holds.
    uint32_t my_word = 0xFFFFFFFFu, my_tile = 0;
    const int wave_first = first + wave * tpw;  // a wavefront takes tpw consecutive tiles of the list
    {
        const int ti = wave_first + lane;
        long long tx = OX + (1ll << 15), ty = OY + (1ll << 15);
        if (lane < tpw && ti < n_crop) {
            const uint32_t tile = k.crop_tiles[ti];
            my_tile = tile;
            my_word = (tile >> 16) * (uint32_t)tb_n + (tile & 0xFFFFu);
            const long long ta8 = (long long)((tile >> 16) * 8u), tb8 = (long long)((tile & 0xFFFFu) * 8u);
            tx += ta8 * M00 + tb8 * M01;
            ty += ta8 * M10 + tb8 * M11;
        }
        corner[wave][lane] = make_longlong2(tx, ty);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t res_lo = 0, res_hi = 0, unk_lo = 0, unk_hi = 0;  // lane q: the bitmap words of tile q
    // the masks of a round's tiles come through the scalar cache (constant address space), in ONE load: the table is padded with zeros
    typedef const unsigned long long __attribute__((address_space(4))) * const_u64_ptr;
    for (int it = 0; it < tpw; it += VBC_U) {
        if (wave_first + it >= n_crop) break;  // uniform
        const const_u64_ptr round = (const_u64_ptr)(uintptr_t)(k.crop_masks + (wave_first + it));
        uint32_t idx[VBC_U];
        int mm[VBC_U], nn[VBC_U];  // (TILED: the cell itself, for the class words of stamped cells)
        unsigned long long look[VBC_U], fovs[VBC_U], risky = 0ull;  // lane masks (scalar registers)
#pragma unroll
        for (int u = 0; u < VBC_U; u++) {
            const unsigned long long fov = round[u];  // 0 past the end of the list
            fovs[u] = fov;
            const longlong2 T = corner[wave][it + u];
            const unsigned long long Fx = (unsigned long long)(T.x + Lx), Fy = (unsigned long long)(T.y + Ly);
            const int m = (int)(uint32_t)(Fx >> 32), n = (int)(uint32_t)(Fy >> 32);
            risky |= fov & __ballot(min((uint32_t)Fx, (uint32_t)Fy) < (1u << 16));
            look[u] = fov & __ballot((uint32_t)m < (uint32_t)Hg) & __ballot((uint32_t)n < (uint32_t)Wg);  // (two compares straight into lane masks)
            if (TILED) {
                mm[u] = m;
                nn[u] = n;
                idx[u] = min(crop_tiled_at(Wt, (uint32_t)m, (uint32_t)n), last_byte);
            } else {
                idx[u] = min((uint32_t)(m * Wg + n), last_cell);  // any valid address for the lanes that do not look
            }
        }
        if (__builtin_expect(risky != 0ull || !fixed_ok, 0)) {  // somebody within 2^-17 of a rounding boundary: the reference's own chain
#pragma unroll
            for (int u = 0; u < VBC_U; u++) {
                // (tile and mask by readlane again, not from memory: a vector load here would turn every lane mask of the loop
                // into a vector register pair and the mask algebra below into vector instructions)
                const uint32_t tile = (uint32_t)__builtin_amdgcn_readlane((int)my_tile, it + u);
                const int a = (int)(tile >> 16) * 8 + da, b = (int)(tile & 0xFFFFu) * 8 + db;
                double wx, wy;
                tf_apply(vw, a * res, b * res, wx, wy);
                const int m = w2m(wx, res), n = w2m(wy, res);
                look[u] = fovs[u] & __ballot((uint32_t)m < (uint32_t)Hg) & __ballot((uint32_t)n < (uint32_t)Wg);
                if (TILED) {
                    mm[u] = m;
                    nn[u] = n;
                    idx[u] = min(crop_tiled_at(Wt, (uint32_t)m, (uint32_t)n), last_byte);
                } else {
                    idx[u] = min((uint32_t)(m * Wg + n), last_cell);
                }
            }
        }
        // free = >= 250 in this robot's private grid (agent.cpp:394-401)
        unsigned long long free_cell[VBC_U];
        if (STAMP) {
            uint32_t o[VBC_U], sg[VBC_U];
            unsigned long long stamped[VBC_U], any_stamped = 0ull;
#pragma unroll
            for (int u = 0; u < VBC_U; u++) {
                o[u] = obs[idx[u]];
                if (!TILED) sg[u] = w.seg_tag[(cell0 + idx[u]) >> 6];
            }
#pragma unroll
            for (int u = 0; u < VBC_U; u++) {
                if (TILED) {
                    free_cell[u] = __ballot(o[u] >= 128u);
                    stamped[u] = look[u] & __ballot((o[u] & 127u) == ctag);
                } else {
                    free_cell[u] = __ballot(o[u] >= 250u);
                    stamped[u] = look[u] & __ballot(sg[u] == tag);
                }
                any_stamped |= stamped[u];
            }
            if (any_stamped != 0ull) {  // near a robot or a pedestrian: class HIGH with nobody else's stamp of this step on it
#pragma unroll
                for (int u = 0; u < VBC_U; u++) {
                    if (stamped[u] == 0ull) continue;
                    const uint32_t v = cls[TILED ? min((uint32_t)(mm[u] * Wg + nn[u]), last_cell) : idx[u]];
                    const uint32_t x = (v & base_tag_mask) ^ base_tag_ours;
                    const unsigned long long f2 = __ballot((v == free_own) | (((x & 7u) == 0u) & (x != 0u)));
                    free_cell[u] = (free_cell[u] & ~stamped[u]) | (f2 & stamped[u]);
                }
            }
        } else {
            uint32_t v[VBC_U];
#pragma unroll
            for (int u = 0; u < VBC_U; u++) v[u] = cls[idx[u]];
#pragma unroll
            for (int u = 0; u < VBC_U; u++) free_cell[u] = __ballot((v[u] == (uint32_t)CLS_HIGH) | (v[u] == free_own));
        }
#pragma unroll
        for (int u = 0; u < VBC_U; u++) {  // the tile's words go to the lane that prepared it
            const unsigned long long occ = look[u] & ~free_cell[u];
            write_lane(res_lo, (uint32_t)occ, it + u);
            write_lane(res_hi, (uint32_t)(occ >> 32), it + u);
            if (want_unknown) {
                write_lane(unk_lo, (uint32_t)~look[u], it + u);
                write_lane(unk_hi, (uint32_t)(~look[u] >> 32), it + u);
            }
        }
    }
    if (my_word != 0xFFFFFFFFu) {  // one store instruction for the wavefront's tiles (consecutive words, mostly)
        plane0[my_word] = ((unsigned long long)res_hi << 32) | res_lo;
        if (want_unknown) plane1[my_word] = ((unsigned long long)unk_hi << 32) | unk_lo;
    }
}

// ------------------------------------------------------------------------------------------------
// laser_map value of a view cell (agent.cpp:419-437, 555-560) from the beams' hit words: the cell's top beam decides, unless
// it leaves the cell alone; then the next lower beam through the cell that writes it does (static list, beam descending).
// big_walk: the list behind its first entry (the top beam), pk = {first entry, count}
__device__ __forceinline__ uint32_t big_walk(const RobotClassDev& rc, const uint32_t* hit, uint2 pk) {
    for (uint32_t e = 1; e < pk.y; e += 4) {  // four entries in flight
        uint32_t ent[4];
#pragma unroll
        for (int q = 0; q < 4; q++) ent[q] = rc.inv_ent[pk.x + min(e + (uint32_t)q, pk.y - 1u)];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (e + (uint32_t)q >= pk.y) break;
            const uint32_t k2 = ent[q] & 0xFFFFu, h2 = hit[ent[q] >> 16], hk2 = h2 >> 16;
            if (k2 < hk2) return 255u;
            if (k2 == hk2) return 0u;
            if (k2 > (h2 & 0xFFFFu)) return 200u;
        }
    }
    return 200u;  // nobody writes the cell: laser_map keeps its initial 200
}
__device__ __forceinline__ uint32_t big_laser_value(const RobotClassDev& rc, const BigClassDev& k, const uint32_t* hit, uint32_t c, uint32_t top) {
    const uint32_t kk = top & 0xFFFFu, hp = hit[top >> 16], hk = hp >> 16;  // hk = 0xFFFF: the beam never hits
    if (kk < hk) return 255u;
    if (kk == hk) return 0u;
    if (kk > (hp & 0xFFFFu)) return 200u;
    return big_walk(rc, hit, k.inv[c]);
}

// view_map value (0 / 100 / 200 / 255) of a cell: bits = the robot's crop bitmap (plane 0 occupied, plane 1 unknown, `words`
// 32-bit words apart), addr = the cell's bit address, st = the own footprint covers it
__device__ __forceinline__ uint32_t big_cell_value(const RobotClassDev& rc, const BigClassDev& k, const uint32_t* hit, const uint32_t* plane0,
                                                   const uint32_t* plane1, bool laser, uint32_t c, uint32_t top, uint32_t addr, bool st) {
    const bool occ = ((plane0[addr >> 5] >> (addr & 31u)) & 1u) != 0u;
    uint32_t v;
    if (!laser) {
        const bool unk = ((plane1[addr >> 5] >> (addr & 31u)) & 1u) != 0u;
        v = occ ? 0u : (unk ? 200u : 255u);
    } else if (st && !occ) {
        return 100u;  // a beam can only write 0 where the crop is occupied
    } else {
        v = big_laser_value(rc, k, hit, c, top);
    }
    return (st && v != 0u) ? 100u : v;  // draw(view_map_, 100) skips 0 / 1 / 2 (agent.cpp:307-312)
}

// ------------------------------------------------------------------------------------------------
// (1) collision and (3) laser.  One workgroup of 256 per robot and 256 beams; LDS: the occupied plane of the crop (LDSBM).
// Every beam walks its static path 32 steps at a time (eight 16-byte loads of bit addresses in flight, consecutive lanes
// contiguous), one LDS bit lookup per step; a wavefront leaves as soon as all its beams have hit or ended.
template <bool POW2, bool STAMP, bool LDSBM>
__global__ __launch_bounds__(VBB_T) void k_beams_big(DevWorld w, int quarters, int qpw) {
    // qpw: blocks of 256 beams one workgroup walks one after the other over ONE copy of the bitmap.  A launch that fills the chip
    // takes two (the 20 KB bitmap leaves a compute unit 8 workgroups: with a workgroup per 256 beams the 8192 workgroups of
    // 2048 robots x 1000 beams passed in four generations, each loading the robot's bitmap again: 98 -> 87 us); a handful of robots
    // (a reset of a few worlds) keeps one block per workgroup: there the single workgroup's latency is the launch's
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the bitmap at LDS address 0 (a path entry >> 5 IS its word's address), one word behind it
    const int tid = threadIdx.x;
    const int groups = (quarters + qpw - 1) / qpw;
    const int t = (int)blockIdx.x / groups, quarter0 = ((int)blockIdx.x - t * groups) * qpw, quarter = quarter0;
    uint32_t& best_sh = *(uint32_t*)(smem + (LDSBM ? 4 * (size_t)w.big_words : 0));
    if (t >= act_count_l(w)) return;
    const int l = act_member(w, w.Rw, t);
    const int B = w.B;
    uint32_t* hit_g = w.big_hit + (size_t)l * w.big_hit_stride;
    if (hit_g[B + 1] == 0u) {  // frozen (k_crop_big looked): every per-robot output keeps its last value
        if (quarter == 0 && tid < WAVE) tail_arrive_view(w, t, l, w.is_coll[l]);
        return;
    }
    const int i = w.r0 + l;
    const int cls = __builtin_amdgcn_readfirstlane(w.robot_cls[i]);
    const RobotClassDev rc = w.rc[cls];
    const BigClassDev k = w.big[cls];
    const uint32_t self = (uint32_t)i;
    uint32_t* bm = (uint32_t*)smem;
    const uint32_t* plane0_g = w.big_bits + (size_t)l * 2 * w.big_words;
    if (LDSBM) {
        const uint4* src = (const uint4*)plane0_g;
        for (int q = tid; q < w.big_words / 4; q += VBB_T) ((uint4*)bm)[q] = src[q];
    }
    if (quarter == 0) {
        // (1) collision: the last footprint sample on an occupied cell decides (agent.cpp:294-326)
        if (tid == 0) best_sh = 0;
        __syncthreads();
        uint32_t best = 0;
        const int n_cov = w.fp_n[l];
        if (n_cov >= 0) {
            const uint2* list = w.fp_cells + (size_t)l * w.fp_cap;
            for (int e = tid; e < n_cov; e += VBB_T) {
                const uint2 ce = list[e];
                const uint32_t cc = cell_seen_class<STAMP>(w.cell[ce.x], self, stamp_tag_of(w));
                best = max(best, cc <= 2 ? ((ce.y << 2) | (cc + 1)) : 0u);
            }
        } else {
            const double* r = w.rec + (size_t)i * IMGENV_RECORD_DOUBLES;
            const Tf2 bw = tf_from_pose_sc(r[0], r[1], r[5], r[6]);
            const size_t cell0 = (size_t)world_of_robot(w, i) * w.Gs;
            for (int q = tid; q < rc.n_fp; q += VBB_T) {
                const double2 fp = rc.fp[q];
                double wx, wy;
                tf_apply(bw, fp.x, fp.y, wx, wy);
                int m, n;
                w2m_pair<POW2>(wx, wy, w.res, w.inv_res, m, n);
                if (m >= 0 && m < w.Hg && n >= 0 && n < w.Wg) {
                    const uint32_t cc = cell_seen_class<STAMP>(w.cell[cell0 + (size_t)m * w.Wg + n], self, stamp_tag_of(w));
                    if (cc <= 2) best = max(best, ((uint32_t)(q + 1) << 2) | (cc + 1));
                }
            }
        }
        if (best) atomicMax(&best_sh, best);
        __syncthreads();
        const int code = (int)(best_sh & 3u);
        // the collision code is all the step's tail needs from the view: hand it over now (see k_view)
        if (tid < WAVE) tail_arrive_view(w, t, l, code);
        if (tid == 0) {
            hit_g[B] = 0u;  // the dummy beam of cells no beam crosses: "hits" at step 0 and leaves nothing alone -> 200
            hit_g[B + 2] = (uint32_t)code;  // is_collision_ of this step: committed by the last kernel of the chain
        }
    } else {
        __syncthreads();
    }
    // (3) laser (agent.cpp:405-438, 511-624): first occupied cell on each beam's precomputed path
    if (w.use_laser == 0) return;
    const uint32_t* plane0 = LDSBM ? (const uint32_t*)bm : plane0_g;
    const int stride = rc.ray_stride, kpad = rc.ray_kpad;
    for (int qq = quarter0; qq < min(quarter0 + qpw, quarters); qq++) {
    const int b = qq * VBB_T + tid, bb = min(b, B - 1);
    const int len = b < B ? (int)rc.ray_len[bb] : 0;
    uint32_t hk = 0xFFFFFFFFu;
    const uint4* col = (const uint4*)k.cells + bb;  // [kpad / 4][stride] four steps per entry (host_tables.h big_bit_entry)
    for (int k0 = 0; k0 < kpad; k0 += 32) {
        uint4 a4[8];
#pragma unroll
        for (int j = 0; j < 8; j++) a4[j] = col[(size_t)((k0 >> 2) + j) * stride];  // padded steps point at the free bit
        uint32_t m = 0;  // step k0 + s ends up in bit 31 - s: three vector instructions a step (address, v_bfe_u32, v_lshl_or_b32)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t ad[4] = {a4[j].x, a4[j].y, a4[j].z, a4[j].w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t word = *(const uint32_t*)((const unsigned char*)plane0 + (ad[q] >> 5));
                m = (m << 1) | __builtin_amdgcn_ubfe(word, ad[q], 1u);
            }
        }
        if (m != 0u && hk == 0xFFFFFFFFu) hk = (uint32_t)k0 + (uint32_t)__builtin_clz(m);
        if (__all((hk != 0xFFFFFFFFu) | (k0 + 32 >= len))) break;
    }
    if (b < B) {
        const bool has = hk != 0xFFFFFFFFu;
        const size_t at = (size_t)(has ? hk : 0u) * stride + b;
        const uint32_t end = has ? (uint32_t)k.ray_end[at] : 0u;
        const float hd = has ? rc.ray_dist[at] : 6.0f;  // agent.cpp:513
        hit_g[b] = has ? ((hk << 16) | end) : 0xFFFFFFFFu;
        w.lasers_raw[(size_t)l * B + b] = hd;
        w.lasers[(size_t)l * B + b] = w.laser_norm ? (double)hd / w.laser_max : (double)hd;
        if (w.hits_x) {  // hit_points_x_ / _y_ (agent.cpp:434-435)
            const size_t hx = (size_t)(has ? hk : (uint32_t)rc.ray_maxlen) * stride + b;
            w.hits_x[(size_t)l * B + b] = rc.ray_hx[hx];
            w.hits_y[(size_t)l * B + b] = rc.ray_hy[hx];
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// cv2.resize(view, image_size, INTER_CUBIC).astype(float16) / 255 (yaml_env.py:431-438), one pixel of the sensor_map per
// thread: the 16 view cells the pixel reads (static tap records) are evaluated from the hit words and the crop bitmap, then
// HResizeCubic of the four source rows and the vertical pass (csrc/cv_resize.h).  The full-size view is not needed.
//   pass 1  every tap from its top beam (one hot word per tap); the few that need more -- left alone by that beam, under the
//           own footprint, or no laser at all -- go on a list in LDS (one prefix sum and one LDS atomic per wavefront);
//   pass 2  the list, spread evenly over the workgroup: crop bit, ray list walk;
//   pass 3  the two resize passes from the 16 values of the pixel.
// LDS: hit[B + 1 .. pad] | vals[VBT_T][16] u8 | list[VBT_T * 16] u16 | counter
// listed: a step -- workgroup `slot` of a robot takes chunk tap_chunks[slot] of its class (the chunks a beam can reach; the others
// hold what the reset's launch, which covers every chunk, left in them)
__global__ __launch_bounds__(VBT_T) void k_taps_big(DevWorld w, int chunks, int commit, int listed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = lane_id();
    const int t = (int)blockIdx.x / chunks, slot = (int)blockIdx.x - t * chunks;
    if (t >= act_count_l(w)) return;
    const int l = act_member(w, w.Rw, t);
    const int B = w.B;
    const uint32_t* hit_g = w.big_hit + (size_t)l * w.big_hit_stride;
    if (hit_g[B + 1] == 0u) return;  // frozen
    if (commit && slot == 0 && tid == 0) w.is_coll[l] = (int)hit_g[B + 2];
    const int cls = __builtin_amdgcn_readfirstlane(w.robot_cls[w.r0 + l]);
    const RobotClassDev rc = w.rc[cls];
    const BigClassDev k = w.big[cls];
    if (commit && slot == 0 && w.angular_map && w.use_laser) angular_bins(w, rc, hit_g, l, tid, VBT_T);
    if (listed && slot >= k.n_tap_chunks) return;
    const int chunk = listed ? (int)k.tap_chunks[slot] : slot;
    const int n_hit4 = (B + 4) / 4;  // B words + the dummy beam, in 16-byte units
    uint32_t* hit = (uint32_t*)smem;
    uint32_t* vals = hit + 4 * n_hit4;                  // [VBT_T][4]: the pixel's 16 tap values, one byte each
    uint16_t* list = (uint16_t*)(vals + 4 * VBT_T);     // pixel of the workgroup << 4 | tap
    int* n_list = (int*)(list + 16 * VBT_T);
    const bool laser = w.use_laser != 0;
    if (laser)
        for (int q = tid; q < n_hit4; q += VBT_T) ((uint4*)hit)[q] = ((const uint4*)hit_g)[q];
    if (tid == 0) *n_list = 0;
    const uint32_t* plane0 = w.big_bits + (size_t)l * 2 * w.big_words;
    const uint32_t* plane1 = plane0 + w.big_words;
    const int IW = w.img_w, IH = w.img_h, NP = IW * IH, vec_end = (IW / 8) * 8;
    const int p = chunk * VBT_T + tid, pc = min(p, NP - 1);
    uint32_t top[16];
#pragma unroll
    for (int j = 0; j < 16; j++) top[j] = k.tap_top[(size_t)j * NP + pc];
    __syncthreads();
    // pass 1
    uint32_t packed[4] = {0u, 0u, 0u, 0u}, more = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (laser) {
            const uint32_t kk = top[j] & 0xFFFFu, hp = hit[(top[j] >> 16) & 0x7FFFu], hk = hp >> 16;
            const uint32_t v = kk < hk ? 255u : (kk == hk ? 0u : 200u);
            packed[j >> 2] |= v << (8 * (j & 3));
            more |= (((kk > hk) & (kk <= (hp & 0xFFFFu))) | ((top[j] >> 31) != 0u)) ? (1u << j) : 0u;
        } else {
            more |= 1u << j;
        }
    }
    if (p >= NP) more = 0;
    *(uint4*)(vals + 4 * tid) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    {
        const uint32_t cnt = (uint32_t)__popc(more), incl = wave_prefix_sum(cnt);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        int base = 0;
        if (lane == 63 && total != 0u) base = atomicAdd(n_list, (int)total);
        base = __builtin_amdgcn_readlane(base, 63);
        int pos = base + (int)(incl - cnt);
        for (uint32_t m = more; m != 0u; m &= m - 1u) list[pos++] = (uint16_t)((tid << 4) | __builtin_ctz(m));
    }
    __syncthreads();
    // pass 2
    const int n_items = *n_list;
    for (int it = tid; it < n_items; it += VBT_T) {
        const uint32_t item = list[it], px = item >> 4, j = item & 15u;
        const size_t at = (size_t)j * NP + (size_t)(chunk * VBT_T) + px;
        const uint32_t tj = k.tap_top[at], addr = k.tap_addr[at];
        const bool st = (tj >> 31) != 0u;
        const bool occ = ((plane0[addr >> 5] >> (addr & 31u)) & 1u) != 0u;
        uint32_t v;
        if (!laser) {
            const bool unk = ((plane1[addr >> 5] >> (addr & 31u)) & 1u) != 0u;
            v = occ ? 0u : (unk ? 200u : 255u);
        } else if (st && !occ) {
            v = 100u;  // a beam can only write 0 where the crop is occupied
        } else {
            const uint32_t kk = tj & 0xFFFFu, hp = hit[(tj >> 16) & 0x7FFFu], hk = hp >> 16;
            v = kk < hk ? 255u : (kk == hk ? 0u : 200u);
            if ((kk > hk) & (kk <= (hp & 0xFFFFu))) v = big_walk(rc, hit, k.tap_inv[at]);
        }
        v = (st && v != 0u) ? 100u : v;  // draw(view_map_, 100) skips 0 / 1 / 2 (agent.cpp:307-312)
        ((uint8_t*)vals)[16 * px + j] = (uint8_t)v;
    }
    __syncthreads();
    // pass 3
    if (p >= NP) return;
    const int dy = p / IW, dx = p - dy * IW;
    const short* al = w.rs_alpha + 4 * dx;
    const short* be = w.rs_beta + 4 * dy;
    const uint4 pv = *(const uint4*)(vals + 4 * tid);
    const uint32_t rows[4] = {pv.x, pv.y, pv.z, pv.w};
    int s[4];
#pragma unroll
    for (int kr = 0; kr < 4; kr++)  // HResizeCubic of the four source rows
        s[kr] = (int)(rows[kr] & 0xFFu) * (int)al[0] + (int)((rows[kr] >> 8) & 0xFFu) * (int)al[1] + (int)((rows[kr] >> 16) & 0xFFu) * (int)al[2] +
                (int)(rows[kr] >> 24) * (int)al[3];
    int o;
    if (dx < vec_end) {  // VResizeCubicVec_32s8u: float32, separately rounded multiply and add, nearest-even
        const float scale = 1.f / (2048 * 2048);
        float v = (float)s[3] * ((float)be[3] * scale);
        v = (float)s[2] * ((float)be[2] * scale) + v;
        v = (float)s[1] * ((float)be[1] * scale) + v;
        v = (float)s[0] * ((float)be[0] * scale) + v;
        o = (int)rintf(v);
    } else {             // VResizeCubic tail: FixedPtCast<int, uchar, 22>
        o = (s[0] * be[0] + s[1] * be[1] + s[2] * be[2] + s[3] * be[3] + (1 << 21)) >> 22;
    }
    w.sensor_maps[(size_t)l * NP + p] = w.f16_lut[min(max(o, 0), 255)];
}

// ------------------------------------------------------------------------------------------------
// (4) the full view: laser_map per cell + own footprint (agent.cpp:419-437, 503), 4 cells per thread, 1024 per workgroup;
// view_maps when it is an output, sensor_maps (float16) when nothing is shrunk.  LDS: the robot's hit words.
__global__ __launch_bounds__(VBF_T) void k_fullview_big(DevWorld w, int chunks, int commit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int t = (int)blockIdx.x / chunks, chunk = (int)blockIdx.x - t * chunks;
    if (t >= act_count_l(w)) return;
    const int l = act_member(w, w.Rw, t);
    const int B = w.B;
    const uint32_t* hit_g = w.big_hit + (size_t)l * w.big_hit_stride;
    if (hit_g[B + 1] == 0u) return;  // frozen
    if (commit && chunk == 0 && tid == 0) w.is_coll[l] = (int)hit_g[B + 2];
    const int cls = __builtin_amdgcn_readfirstlane(w.robot_cls[w.r0 + l]);
    const RobotClassDev rc = w.rc[cls];
    const BigClassDev k = w.big[cls];
    if (commit && chunk == 0 && w.angular_map && w.use_laser) angular_bins(w, rc, hit_g, l, tid, VBF_T);
    const bool laser = w.use_laser != 0;
    uint32_t* hit = (uint32_t*)smem;
    if (laser) {
        for (int q = tid; q < (B + 4) / 4; q += VBF_T) ((uint4*)hit)[q] = ((const uint4*)hit_g)[q];  // B words + the dummy beam
        __syncthreads();
    }
    const int NC = w.Hv * w.Wv, c4 = (chunk * VBF_T + tid) * 4;
    if (c4 >= NC) return;
    const uint32_t* plane0 = w.big_bits + (size_t)l * 2 * w.big_words;
    const uint32_t* plane1 = plane0 + w.big_words;
    const uint32_t Wv = (uint32_t)w.Wv, tb_n = (uint32_t)k.tb;
    const uint32_t stamp = (rc.stamp_bits[c4 >> 5] >> (c4 & 31)) & 0xFu;
    // class index per cell: 0 -> 0, 1 -> 100, 2 -> 200, 3 -> 255 (as in k_view)
    uint32_t I = 0x02020202u;
    if (laser) {
        uint32_t top[4] = {0u, 0u, 0u, 0u};
        if (c4 + 4 <= NC) {
            const uint4 t4 = *(const uint4*)(rc.top_ent + c4);
            top[0] = t4.x; top[1] = t4.y; top[2] = t4.z; top[3] = t4.w;
        } else {
            for (int q = 0; q < 4 && c4 + q < NC; q++) top[q] = rc.top_ent[c4 + q];
        }
        uint32_t alone = 0;
        I = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {  // the top beam's verdict
            const uint32_t kk = top[q] & 0xFFFFu, hp = hit[top[q] >> 16], hk = hp >> 16;
            I |= (kk < hk ? 3u : (kk == hk ? 0u : 2u)) << (8 * q);
            alone |= ((kk > hk) & (kk <= (hp & 0xFFFFu))) ? (1u << q) : 0u;
        }
        for (uint32_t m = alone; m != 0u; m &= m - 1u) {  // cells that beam leaves alone: the list behind it
            const int q = __builtin_ctz(m);
            const uint32_t v = big_walk(rc, hit, k.inv[min(c4 + q, NC - 1)]);
            I = (I & ~(0xFFu << (8 * q))) | ((v == 255u ? 3u : (v == 0u ? 0u : 2u)) << (8 * q));
        }
    }
    if (!laser || stamp != 0u) {  // the crop bit itself is only needed without the laser or under the own footprint
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (laser && !((stamp >> q) & 1u)) continue;
            const uint32_t c = (uint32_t)min(c4 + q, NC - 1), a = c / Wv, b = c - a * Wv;
            const uint32_t addr = ((a >> 3) * tb_n + (b >> 3)) * 64u + (a & 7u) * 8u + (b & 7u);
            const bool occ = ((plane0[addr >> 5] >> (addr & 31u)) & 1u) != 0u;
            uint32_t ci = (I >> (8 * q)) & 0xFFu;
            if (!laser) {
                const bool unk = ((plane1[addr >> 5] >> (addr & 31u)) & 1u) != 0u;
                ci = occ ? 0u : (unk ? 2u : 3u);
            } else if (!occ) {
                ci = 1u;  // under the own footprint a beam can only write 0 where the crop is occupied: 100 (agent.cpp:307-312)
            }
            if (((stamp >> q) & 1u) && ci != 0u) ci = 1u;
            I = (I & ~(0xFFu << (8 * q))) | (ci << (8 * q));
        }
    }
    uint8_t* out_u8 = w.view_maps + (size_t)l * NC;
    uint16_t* out_f16 = w.sensor_maps + (size_t)l * NC;  // only when nothing is shrunk
    const bool vec = (NC & 3) == 0;  // then every robot's rows start on a 4-cell boundary
    const uint32_t lut_u8 = 0u | (100u << 8) | (200u << 16) | (255u << 24);
    if (w.keep_view_maps) {
        const uint32_t packed = __builtin_amdgcn_perm(lut_u8, lut_u8, I);
        if (vec) *(uint32_t*)(out_u8 + c4) = packed;
        else for (int q = 0; q < 4 && c4 + q < NC; q++) out_u8[c4 + q] = (uint8_t)(packed >> (8 * q));
    }
    if (!w.resize) {
        const uint32_t h0 = w.f16_lut[0], h1 = w.f16_lut[100], h2 = w.f16_lut[200], h3 = w.f16_lut[255];
        const uint32_t lut_lo = (h0 & 0xFFu) | ((h1 & 0xFFu) << 8) | ((h2 & 0xFFu) << 16) | ((h3 & 0xFFu) << 24);
        const uint32_t lut_hi = (h0 >> 8) | ((h1 >> 8) << 8) | ((h2 >> 8) << 16) | ((h3 >> 8) << 24);
        // two float16 values per v_perm: selector bytes (c, c + 4) pick the low byte out of lut_lo and the high one out of lut_hi
        const uint32_t s01 = __builtin_amdgcn_perm(I, I, 0x01010000u) | 0x04000400u, s23 = __builtin_amdgcn_perm(I, I, 0x03030202u) | 0x04000400u;
        const uint32_t f01 = __builtin_amdgcn_perm(lut_hi, lut_lo, s01), f23 = __builtin_amdgcn_perm(lut_hi, lut_lo, s23);
        if (vec) *(uint2*)(out_f16 + c4) = make_uint2(f01, f23);
        else for (int q = 0; q < 4 && c4 + q < NC; q++) out_f16[c4 + q] = (uint16_t)((q < 2 ? f01 : f23) >> (16 * (q & 1)));
    }
}
