/*
 * THIRD-PARTY NOTICE.  The ORCA half-plane construction, the linear programs and the obstacle k-d tree walk in this file follow
 * the RVO2 Library (as vendored by the reference in src/3rdparty/ervo_ros, with the ERVO additions) statement for statement --
 * a float32 LP whose result depends on the order of its operations has essentially one spelling if it is to stay bit-exact:
 *
 *   RVO2 Library.  Copyright 2008 University of North Carolina at Chapel Hill.
 *   Licensed under the Apache License, Version 2.0 (the "License"); you may not use this file except in compliance with the
 *   License.  You may obtain a copy of the License at http://www.apache.org/licenses/LICENSE-2.0
 *   Unless required by applicable law or agreed to in writing, software distributed under the License is distributed on an
 *   "AS IS" BASIS, WITHOUT WARRANTIES OR CONDITIONS OF ANY KIND, either express or implied.  See the License for the specific
 *   language governing permissions and limitations under the License.
 *   Authors: Jur van den Berg, Stephen J. Guy, Jamie Snape, Ming C. Lin, Dinesh Manocha -- <http://gamma.cs.unc.edu/RVO2/>
 *
 * Changes made here: restated in HIP device code for gfx950, index links instead of pointers, agents of one group side by side in the rows of a wavefront, obstacle tree evaluated per row out of LDS (the kernel around it, k_orca in kernels.h, is this project's own).  See NOTICE at the repository root.
 */
// orca_device.h -- ORCA (RVO2 v2.0.x) + ERVO for one pedestrian per wavefront, float32.
//
// Reference: src/3rdparty/ervo_ros  Agent::computeNeighbors (src/Agent.cpp:50-61),
// computeNewVelocity[ForERVO] (72-434, 437-793), insertAgentNeighbor / insertObstacleNeighbor
// (795-838), linearProgram1/2/3 (845-1001), KdTree::queryObstacleTreeRecursive (src/KdTree.cpp:
// 310-353), helpers in include/ervo_ros/Vector2.h and Definitions.h.
//
// MI355X mapping: a wavefront takes a GROUP of up to 8 pedestrians of one world.  The agent-neighbour search is a coalesced
// brute-force scan of the SoA agent arrays by all 64 lanes -- each candidate is loaded once and tested against every agent
// of the group (ballot + ordered insertion by the agent's lane, with the reference's shrinking range) -- equivalent to the
// kd-tree query up to the order of exactly equidistant neighbours.  The world's obstacle segments and their BSP tree (built at
// reset with the reference's algorithm) are staged into LDS once per wavefront; the tree is walked iteratively in the
// reference's traversal order.  Half-plane construction and the 2-D linear programs are sequential and tiny (<= 10 agent
// lines): one LANE per agent of the group runs them out of its own LDS scratch (divergent, but 4-8 times the lane use of one
// agent per wavefront, and every pointer chase is an LDS access instead of an HBM round trip).  Every float operation is kept
// in the reference's order (-ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

#include "world.h"

#define RVO_EPS 0.00001f
#define ORCA_MAX_ON 118  // obstacle neighbours kept per agent
#define ORCA_MAX_AN 10   // rvoscene.h:57,63 maxNeighbors
#define ORCA_MAX_LINES (ORCA_MAX_ON + ORCA_MAX_AN)
#define ORCA_STACK 128

struct f2 {
    float x, y;
};
__device__ __forceinline__ f2 F2(float x, float y) {
    f2 r;
    r.x = x;
    r.y = y;
    return r;
}
__device__ __forceinline__ f2 operator+(f2 a, f2 b) { return F2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ f2 operator-(f2 a, f2 b) { return F2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ f2 operator-(f2 a) { return F2(-a.x, -a.y); }
__device__ __forceinline__ float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ f2 operator*(f2 a, float s) { return F2(a.x * s, a.y * s); }
__device__ __forceinline__ f2 operator*(float s, f2 a) { return F2(s * a.x, s * a.y); }
__device__ __forceinline__ f2 vdiv(f2 a, float s) {
    const float inv = 1.0f / s;
    return F2(a.x * inv, a.y * inv);
}
__device__ __forceinline__ float abs_sq(f2 a) { return dot(a, a); }
__device__ __forceinline__ float vabs(f2 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ float det(f2 a, f2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ f2 normalize(f2 a) { return vdiv(a, vabs(a)); }
__device__ __forceinline__ float sqr(float a) { return a * a; }
__device__ __forceinline__ float left_of(f2 a, f2 b, f2 c) { return det(a - c, b - a); }
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }

struct OrcaLine {
    f2 point, direction;
};

// LDS scratch of one agent (one lane of a group): pointers into the workgroup's dynamic LDS, capacities chosen at launch from
// the handle's largest obstacle table (at most ORCA_MAX_ON obstacle neighbours, ORCA_STACK tree levels)
struct OrcaScratch {
    OrcaLine* lines;   // [cap_on + ORCA_MAX_AN]
    OrcaLine* proj;    // [cap_on + ORCA_MAX_AN]
    float* on_dist;    // [cap_on]
    int* on_idx;       // [cap_on]
    float* an_dist;    // [ORCA_MAX_AN]
    int* an_idx;       // [ORCA_MAX_AN]
    int* stack;        // [cap_stack]
    int n_an, n_on, cap_on, cap_stack;
};
__host__ __device__ inline size_t orca_scratch_bytes(int cap_on, int cap_stack) {
    // (a multiple of 16: the per-node records behind the scratch are uint4, and so is every later row's base)
    return ((size_t)(cap_on + ORCA_MAX_AN) * 2 * sizeof(OrcaLine) + (size_t)cap_on * 8 + ORCA_MAX_AN * 8 + (size_t)cap_stack * 4 + 15) & ~(size_t)15;
}
__device__ __forceinline__ void orca_scratch_carve(OrcaScratch& s, unsigned char* base, int cap_on, int cap_stack) {
    const int nl = cap_on + ORCA_MAX_AN;
    s.lines = (OrcaLine*)base;
    s.proj = s.lines + nl;
    s.on_dist = (float*)(s.proj + nl);
    s.on_idx = (int*)(s.on_dist + cap_on);
    s.an_dist = (float*)(s.on_idx + cap_on);
    s.an_idx = (int*)(s.an_dist + ORCA_MAX_AN);
    s.stack = s.an_idx + ORCA_MAX_AN;
    s.n_an = s.n_on = 0;
    s.cap_on = cap_on;
    s.cap_stack = cap_stack;
}

// the obstacle segments and BSP nodes of the agent's world: in LDS when they fit the staging area, else in HBM (generic pointers)
struct OrcaObst {
    const RvoObstDev* obst;
    const RvoNodeDev* onodes;
    int n_obst, oroot;
};

__device__ float dist_sq_point_segment(f2 a, f2 b, f2 c) {
    const float r = dot(c - a, b - a) / abs_sq(b - a);
    if (r < 0.0f) {
        return abs_sq(c - a);
    } else if (r > 1.0f) {
        return abs_sq(c - b);
    } else {
        return abs_sq(c - (a + r * (b - a)));
    }
}

__device__ __forceinline__ f2 opoint(const OrcaObst& w, int i) { return F2(w.obst[i].px, w.obst[i].py); }
__device__ __forceinline__ f2 ounit(const OrcaObst& w, int i) { return F2(w.obst[i].ux, w.obst[i].uy); }

// Agent::insertAgentNeighbor (the agent's lane)
__device__ void insert_agent_neighbor(OrcaScratch& s, float dist_sq, int other, float& range_sq) {
    if (dist_sq < range_sq) {
        if (s.n_an < ORCA_MAX_AN) {
            s.an_dist[s.n_an] = dist_sq;
            s.an_idx[s.n_an] = other;
            s.n_an++;
        }
        int i = s.n_an - 1;
        while (i != 0 && dist_sq < s.an_dist[i - 1]) {
            s.an_dist[i] = s.an_dist[i - 1];
            s.an_idx[i] = s.an_idx[i - 1];
            --i;
        }
        s.an_dist[i] = dist_sq;
        s.an_idx[i] = other;
        if (s.n_an == ORCA_MAX_AN) range_sq = s.an_dist[s.n_an - 1];
    }
}

// Agent::insertObstacleNeighbor (the agent's lane)
__device__ __forceinline__ void insert_obstacle_neighbor(const OrcaObst& w, int* err, OrcaScratch& s, f2 pos, int ob, float range_sq) {
    const int nx = w.obst[ob].next;
    const float dist_sq = dist_sq_point_segment(opoint(w, ob), opoint(w, nx), pos);
    if (dist_sq < range_sq) {
        if (s.n_on >= s.cap_on) {
            err[0] = 1;  // more visible obstacle segments than the scratch holds
            return;
        }
        s.on_dist[s.n_on] = dist_sq;
        s.on_idx[s.n_on] = ob;
        s.n_on++;
        int i = s.n_on - 1;
        while (i != 0 && dist_sq < s.on_dist[i - 1]) {
            s.on_dist[i] = s.on_dist[i - 1];
            s.on_idx[i] = s.on_idx[i - 1];
            --i;
        }
        s.on_dist[i] = dist_sq;
        s.on_idx[i] = ob;
    }
}

// KdTree::queryObstacleTreeRecursive, iteratively, same visiting order (the agent's lane)
__device__ __forceinline__ void query_obstacle_tree(const OrcaObst& w, int* err, OrcaScratch& s, f2 pos, float range_sq) {
    int sp = 0;
    if (w.oroot < 0) return;
    s.stack[sp++] = w.oroot << 1;  // (node << 1) | stage
    while (sp > 0) {
        const int top = s.stack[sp - 1];
        const int node = top >> 1;
        const int o1 = w.onodes[node].obstacle;
        const int o2 = w.obst[o1].next;
        const float agent_left = left_of(opoint(w, o1), opoint(w, o2), pos);
        if ((top & 1) == 0) {
            s.stack[sp - 1] = top | 1;
            const int child = (agent_left >= 0.0f ? w.onodes[node].left : w.onodes[node].right);
            if (child >= 0) {
                if (sp >= s.cap_stack) {
                    err[1] = 1;
                    return;
                }
                s.stack[sp++] = child << 1;
            }
        } else {
            sp--;
            const float dist_sq_line = sqr(agent_left) / abs_sq(opoint(w, o2) - opoint(w, o1));
            if (dist_sq_line < range_sq) {
                if (agent_left < 0.0f) insert_obstacle_neighbor(w, err, s, pos, o1, range_sq);
                const int child = (agent_left >= 0.0f ? w.onodes[node].right : w.onodes[node].left);
                if (child >= 0) s.stack[sp++] = child << 1;
            }
        }
    }
}

__device__ bool linear_program1(const OrcaLine* lines, int line_no, float radius, f2 opt, bool direction_opt,
                                f2& result) {
    const f2 lp = lines[line_no].point, ld = lines[line_no].direction;
    const float dot_product = dot(lp, ld);
    const float discriminant = sqr(dot_product) + sqr(radius) - abs_sq(lp);
    if (discriminant < 0.0f) return false;
    const float sqrt_disc = sqrtf(discriminant);
    float t_left = -dot_product - sqrt_disc;
    float t_right = -dot_product + sqrt_disc;
    for (int i = 0; i < line_no; ++i) {
        const float denominator = det(ld, lines[i].direction);
        const float numerator = det(lines[i].direction, lp - lines[i].point);
        if (fabsf(denominator) <= RVO_EPS) {
            if (numerator < 0.0f) {
                return false;
            } else {
                continue;
            }
        }
        const float t = numerator / denominator;
        if (denominator >= 0.0f) {
            t_right = std_min(t_right, t);
        } else {
            t_left = std_max(t_left, t);
        }
        if (t_left > t_right) return false;
    }
    if (direction_opt) {
        if (dot(opt, ld) > 0.0f) {
            result = lp + t_right * ld;
        } else {
            result = lp + t_left * ld;
        }
    } else {
        const float t = dot(ld, opt - lp);
        if (t < t_left) {
            result = lp + t_left * ld;
        } else if (t > t_right) {
            result = lp + t_right * ld;
        } else {
            result = lp + t * ld;
        }
    }
    return true;
}

__device__ int linear_program2(const OrcaLine* lines, int n, float radius, f2 opt, bool direction_opt, f2& result) {
    if (direction_opt) {
        result = opt * radius;
    } else if (abs_sq(opt) > sqr(radius)) {
        result = normalize(opt) * radius;
    } else {
        result = opt;
    }
    for (int i = 0; i < n; ++i) {
        if (det(lines[i].direction, lines[i].point - result) > 0.0f) {
            const f2 temp = result;
            if (!linear_program1(lines, i, radius, opt, direction_opt, result)) {
                result = temp;
                return i;
            }
        }
    }
    return n;
}

__device__ void linear_program3(OrcaScratch& s, int n, int num_obst_lines, int begin_line, float radius, f2& result) {
    float distance = 0.0f;
    const OrcaLine* lines = s.lines;
    for (int i = begin_line; i < n; ++i) {
        if (det(lines[i].direction, lines[i].point - result) > distance) {
            int np = 0;
            for (int k = 0; k < num_obst_lines; ++k) s.proj[np++] = lines[k];
            for (int j = num_obst_lines; j < i; ++j) {
                OrcaLine line;
                const float determinant = det(lines[i].direction, lines[j].direction);
                if (fabsf(determinant) <= RVO_EPS) {
                    if (dot(lines[i].direction, lines[j].direction) > 0.0f) {
                        continue;
                    } else {
                        line.point = 0.5f * (lines[i].point + lines[j].point);
                    }
                } else {
                    line.point = lines[i].point +
                                 (det(lines[j].direction, lines[i].point - lines[j].point) / determinant) * lines[i].direction;
                }
                line.direction = normalize(lines[j].direction - lines[i].direction);
                s.proj[np++] = line;
            }
            const f2 temp = result;
            if (linear_program2(s.proj, np, radius, F2(-lines[i].direction.y, lines[i].direction.x), true, result) < np) {
                result = temp;
            }
            distance = det(lines[i].direction, lines[i].point - result);
        }
    }
}

// ---- Agent::computeNewVelocity (Agent.cpp:437-793) in pieces, so that a group of lanes can share one agent's work ----
#define ORCA_RADIUS 0.5f        // rvoscene.h:57,63
#define ORCA_TIME_HORIZON 5.0f
#define ORCA_TIME_HORIZON_OBST 5.0f

// "already covered" (Agent.cpp:466-477): both end points of the obstacle segment, scaled by 1 / timeHorizonObst, lie behind line L
__device__ __forceinline__ bool obstacle_covered_by(f2 rel1s, f2 rel2s, const OrcaLine& L) {
    const float inv_tho = 1.0f / ORCA_TIME_HORIZON_OBST, radius = ORCA_RADIUS;
    return det(rel1s - L.point, L.direction) - inv_tho * radius >= -RVO_EPS && det(rel2s - L.point, L.direction) - inv_tho * radius >= -RVO_EPS;
}

// The ORCA line of obstacle neighbour o1 (the segment o1 -> next) for an agent at pos moving with vel, if it yields one
// (Agent.cpp:479-671): everything behind the covered test, which is the only part that looks at the lines pushed so far
__device__ __forceinline__ bool obstacle_line(const OrcaObst& w, f2 pos, f2 vel, int o1, OrcaLine& out) {
    const float radius = ORCA_RADIUS;
    const float inv_tho = 1.0f / ORCA_TIME_HORIZON_OBST;
#define EMIT(P_, D_)            \
    do {                        \
        out.point = (P_);       \
        out.direction = (D_);   \
        return true;            \
    } while (0)
    int o2 = w.obst[o1].next;
    const f2 rel1 = opoint(w, o1) - pos;
    const f2 rel2 = opoint(w, o2) - pos;
    const float dsq1 = abs_sq(rel1), dsq2 = abs_sq(rel2);
    const float rsq = sqr(radius);
    const f2 ovec = opoint(w, o2) - opoint(w, o1);
    const float sp = dot(-rel1, ovec) / abs_sq(ovec);
    const float dsq_line = abs_sq(-rel1 - sp * ovec);
    const f2 u1 = ounit(w, o1);
    if (sp < 0.0f && dsq1 <= rsq) {
        if (w.obst[o1].is_convex) EMIT(F2(0.0f, 0.0f), normalize(F2(-rel1.y, rel1.x)));
        return false;
    } else if (sp > 1.0f && dsq2 <= rsq) {
        if (w.obst[o2].is_convex && det(rel2, ounit(w, o2)) >= 0.0f) EMIT(F2(0.0f, 0.0f), normalize(F2(-rel2.y, rel2.x)));
        return false;
    } else if (sp >= 0.0f && sp < 1.0f && dsq_line <= rsq) {
        EMIT(F2(0.0f, 0.0f), -u1);
    }
    f2 left_leg, right_leg;
    if (sp < 0.0f && dsq_line <= rsq) {
        if (!w.obst[o1].is_convex) return false;
        o2 = o1;
        const float leg1 = sqrtf(dsq1 - rsq);
        left_leg = vdiv(F2(rel1.x * leg1 - rel1.y * radius, rel1.x * radius + rel1.y * leg1), dsq1);
        right_leg = vdiv(F2(rel1.x * leg1 + rel1.y * radius, -rel1.x * radius + rel1.y * leg1), dsq1);
    } else if (sp > 1.0f && dsq_line <= rsq) {
        if (!w.obst[o2].is_convex) return false;
        o1 = o2;
        const float leg2 = sqrtf(dsq2 - rsq);
        left_leg = vdiv(F2(rel2.x * leg2 - rel2.y * radius, rel2.x * radius + rel2.y * leg2), dsq2);
        right_leg = vdiv(F2(rel2.x * leg2 + rel2.y * radius, -rel2.x * radius + rel2.y * leg2), dsq2);
    } else {
        if (w.obst[o1].is_convex) {
            const float leg1 = sqrtf(dsq1 - rsq);
            left_leg = vdiv(F2(rel1.x * leg1 - rel1.y * radius, rel1.x * radius + rel1.y * leg1), dsq1);
        } else {
            left_leg = -u1;
        }
        if (w.obst[o2].is_convex) {
            const float leg2 = sqrtf(dsq2 - rsq);
            right_leg = vdiv(F2(rel2.x * leg2 + rel2.y * radius, -rel2.x * radius + rel2.y * leg2), dsq2);
        } else {
            right_leg = u1;
        }
    }
    const f2 uo1 = ounit(w, o1), uo2 = ounit(w, o2);
    const f2 uln = ounit(w, w.obst[o1].prev);
    bool left_foreign = false, right_foreign = false;
    if (w.obst[o1].is_convex && det(left_leg, -uln) >= 0.0f) {
        left_leg = -uln;
        left_foreign = true;
    }
    if (w.obst[o2].is_convex && det(right_leg, uo2) <= 0.0f) {
        right_leg = uo2;
        right_foreign = true;
    }
    const f2 left_cutoff = inv_tho * (opoint(w, o1) - pos);
    const f2 right_cutoff = inv_tho * (opoint(w, o2) - pos);
    const f2 cutoff_vec = right_cutoff - left_cutoff;
    const float t = (o1 == o2 ? 0.5f : dot(vel - left_cutoff, cutoff_vec) / abs_sq(cutoff_vec));
    const float t_left = dot(vel - left_cutoff, left_leg);
    const float t_right = dot(vel - right_cutoff, right_leg);
    if ((t < 0.0f && t_left < 0.0f) || (o1 == o2 && t_left < 0.0f && t_right < 0.0f)) {
        const f2 unit_w = normalize(vel - left_cutoff);
        EMIT(left_cutoff + radius * inv_tho * unit_w, F2(unit_w.y, -unit_w.x));
    } else if (t > 1.0f && t_right < 0.0f) {
        const f2 unit_w = normalize(vel - right_cutoff);
        EMIT(right_cutoff + radius * inv_tho * unit_w, F2(unit_w.y, -unit_w.x));
    }
    const float inf = __builtin_huge_valf();
    const float dsq_cutoff = ((t < 0.0f || t > 1.0f || o1 == o2) ? inf : abs_sq(vel - (left_cutoff + t * cutoff_vec)));
    const float dsq_left = ((t_left < 0.0f) ? inf : abs_sq(vel - (left_cutoff + t_left * left_leg)));
    const float dsq_right = ((t_right < 0.0f) ? inf : abs_sq(vel - (right_cutoff + t_right * right_leg)));
    if (dsq_cutoff <= dsq_left && dsq_cutoff <= dsq_right) {
        const f2 d = -uo1;
        EMIT(left_cutoff + radius * inv_tho * F2(-d.y, d.x), d);
    } else if (dsq_left <= dsq_right) {
        if (left_foreign) return false;
        const f2 d = left_leg;
        EMIT(left_cutoff + radius * inv_tho * F2(-d.y, d.x), d);
    } else {
        if (right_foreign) return false;
        const f2 d = -right_leg;
        EMIT(right_cutoff + radius * inv_tho * F2(-d.y, d.x), d);
    }
#undef EMIT
}

// The ORCA line against agent neighbour `other` (Agent.cpp:676-770)
__device__ __forceinline__ OrcaLine agent_line(const DevWorld& dw, f2 pos, f2 vel, int other) {
    const float radius = ORCA_RADIUS;
    const float inv_th = 1.0f / ORCA_TIME_HORIZON;
    const f2 rel_pos = F2(dw.apx[other], dw.apy[other]) - pos;
    const f2 rel_vel = vel - F2(dw.avx[other], dw.avy[other]);
    const float dist_sq = abs_sq(rel_pos);
    const float comb = radius + 0.5f;  // every agent has radius 0.5
    const float comb_sq = sqr(comb);
    f2 dir, u;
    if (dist_sq > comb_sq) {
        const f2 ww = rel_vel - inv_th * rel_pos;
        const float wl_sq = abs_sq(ww);
        const float dp1 = dot(ww, rel_pos);
        if (dp1 < 0.0f && sqr(dp1) > comb_sq * wl_sq) {
            const float wl = sqrtf(wl_sq);
            const f2 unit_w = vdiv(ww, wl);
            dir = F2(unit_w.y, -unit_w.x);
            u = (comb * inv_th - wl) * unit_w;
        } else {
            const float leg = sqrtf(dist_sq - comb_sq);
            if (det(rel_pos, ww) > 0.0f) {
                dir = vdiv(F2(rel_pos.x * leg - rel_pos.y * comb, rel_pos.x * comb + rel_pos.y * leg), dist_sq);
            } else {
                dir = vdiv(-F2(rel_pos.x * leg + rel_pos.y * comb, -rel_pos.x * comb + rel_pos.y * leg), dist_sq);
            }
            const float dp2 = dot(rel_vel, dir);
            u = dp2 * dir - rel_vel;
        }
    } else {
        const float inv_ts = 1.0f / (float)dw.step_hz;
        const f2 ww = rel_vel - inv_ts * rel_pos;
        const float wl = vabs(ww);
        const f2 unit_w = vdiv(ww, wl);
        dir = F2(unit_w.y, -unit_w.x);
        u = (comb * inv_ts - wl) * unit_w;
    }
    OrcaLine L;
    L.point = vel + 0.5f * u;
    L.direction = dir;
    return L;
}

// the two linear programs on the lines in s.lines[0, nl): the new velocity (Agent.cpp:772-792)
__device__ __forceinline__ f2 solve_velocity(OrcaScratch& s, float max_speed, f2 nv /* newVelocity_ as the last step left it */, int nl,
                                             int num_obst_lines, f2 pref) {
    const int fail = linear_program2(s.lines, nl, max_speed, pref, false, nv);
    if (fail < nl) linear_program3(s, nl, num_obst_lines, fail, max_speed, nv);
    return nv;
}

// Agent::computeNewVelocity for agent `self` given its neighbour lists in the scratch, all on the agent's own lane
__device__ __forceinline__ f2 compute_new_velocity(const DevWorld& dw, const OrcaObst& w, OrcaScratch& s, int self, f2 pref) {
    const f2 pos = F2(dw.apx[self], dw.apy[self]);
    const f2 vel = F2(dw.avx[self], dw.avy[self]);
    const float inv_tho = 1.0f / ORCA_TIME_HORIZON_OBST;
    int nl = 0;
    OrcaLine* L = s.lines;
    for (int i = 0; i < s.n_on; ++i) {
        const int o1 = s.on_idx[i];
        const int o2 = w.obst[o1].next;
        const f2 rel1s = inv_tho * (opoint(w, o1) - pos), rel2s = inv_tho * (opoint(w, o2) - pos);
        bool covered = false;
        for (int j = 0; j < nl; ++j) {
            if (obstacle_covered_by(rel1s, rel2s, L[j])) {
                covered = true;
                break;
            }
        }
        if (covered) continue;
        OrcaLine ln;
        if (obstacle_line(w, pos, vel, o1, ln)) L[nl++] = ln;
    }
    const int num_obst_lines = nl;
    for (int i = 0; i < s.n_an; ++i) L[nl++] = agent_line(dw, pos, vel, s.an_idx[i]);
    return solve_velocity(s, dw.amax_speed[self], F2(dw.anvx[self], dw.anvy[self]), nl, num_obst_lines, pref);
}
