/*
 * THIRD-PARTY NOTICE.  The force model, its constants, the waypoint logic and the quadtree's insert / erase / split rules in this
 * file restate libpedsim ("pedsim - A microscopic pedestrian simulation system.  Copyright (c) by Christian Gloor",
 * http://pedsim.silmaril.org/, distributed under the GNU General Public License) as vendored by the reference in
 * src/3rdparty/pedsimros -- formulas and rule order are necessarily its own, since results must match it.
 * Changes made here: restated for one 256-thread workgroup per crowd on gfx950 -- pair terms spread over the chip, the tree's surgery decided from per-leaf counts and replayed only where order matters, a correctly rounded atan2 (cr_atan2.h).  See NOTICE at the repository root.
 */
// sfm.h -- libpedsim social-force pedestrians (PedScene, src/img_env/src/pedscene.h:17-91) for the HIP path.
//
// Reference: src/3rdparty/pedsimros  Tagent::desiredForce / socialForce / obstacleForce / lookaheadForce /
// computeForces / move (src/ped_agent.cpp:236-571), Twaypoint::getForce (src/ped_waypoint.cpp:81-134),
// Tobstacle::closestPoint (src/ped_obstacle.cpp:90-113), Tvector (src/ped_vector.cpp, 3-D: z matters because
// robots sit at z = 1, pedscene.h:54), Tscene::moveAgents / getNeighbors (src/ped_scene.cpp:167-252) and the
// quadtree Ttree (src/ped_tree.cpp).
//
// The quadtree is part of the behaviour, not an accelerator: PedScene builds it over x in [0,10], y in [10,20]
// (pedscene.h:18), Ttree::moveAgent re-inserts an agent that left its leaf from the root and THEN erases it
// from the old leaf (ped_tree.cpp:131-137) -- which removes it altogether when both are the same leaf -- and
// the force loops only see agents that are still in the tree.  So the tree, its treehash and its split rule
// are carried in HBM and updated exactly as the reference does (sequentially, in agent order).
//
// The reference can only run small crowds here (more than 8 agents at one position -- e.g. 9 robots, which all
// start at (0,0,0) -- recurse forever in Ttree::addAgent), so one 256-thread workgroup handles the whole
// crowd: thread i owns agent i.  float64 throughout, -ffp-contract=off; force sums run in agent order
// (the reference iterates a std::set ordered by heap address = allocation order).
#pragma once
#include <math.h>
#include <stdint.h>

#include "cr_atan2.h"

#if defined(__HIPCC__)
#define SFM_HD __host__ __device__
#else
#define SFM_HD
#endif

#define SFM_MAX_AGENTS 256
#define SFM_MAX_WP 8
#define SFM_LEAF_CAP 12
#define SFM_MAX_DEPTH 64
#define SFM_WALK_CAP 24  // LDS stack entries per agent of the neighbour walk
#define SFM_FULL_CAP 160   // leaves with members that the partial neighbour scan lists (more: the literal walk)
#define SFM_LDS_NODES 1024  // quadtree nodes mirrored in (dynamic) LDS for a step: 104 KB (larger trees are walked in HBM)

struct SfmNode {  // Ped::Ttree
    double x, y, w, h;
    int isleaf, n_agents;
    int child[4];
    int agents[SFM_LEAF_CAP];  // std::set<const Tagent*>, kept sorted by agent index
};

struct SfmDev {  // one crowd, or W of them back to back (one per world of a multi-world handle: sfm_of_world)
    int n, n_peds, n_obs, cap_nodes;  // agents / pedestrians of ONE crowd
    int W, cap_obs;                   // crowds; obstacle segments a crowd has room for
    const int* n_obs_w;               // [W] obstacle segments per crowd (W > 1; n_obs otherwise)
    double* p;     // [n][3]
    double* v;     // [n][3]
    double* vmax;  // [n]
    double *wpx, *wpy, *wpr;  // [n][SFM_MAX_WP]
    int *dq, *dq_n, *dest, *last;  // waypoint deque (indices), its length, destination, lastdestination
    double* obs;  // [n_obs][4] ax ay bx by
    SfmNode* nodes;
    int* n_nodes;   // [1]
    int* treehash;  // [n]
    int* err;       // [1] overflow flag (node pool / leaf capacity / depth)
    double* pair_f;            // [n (neighbour)][n (agent)][3] social-force term of (agent, neighbour)
    unsigned char* pair_code;  // [n (neighbour)][n (agent)] lookahead vote + 1 | has-term << 2
    uint32_t* g_nb;            // [SFM_MAX_AGENTS][SFM_MAX_AGENTS / 32] neighbour sets, phase 1 -> 2 of a split step
    double* g_sh;              // [4][SFM_MAX_AGENTS] desired direction x / y and two angles per agent, phase 1 -> 2, 3
    // Where a step WRITES the state it changes.  The same arrays as above: the step happens in place.  A second set: the step reads
    // the crowd as it is and leaves the next state beside it (k_sfm run a step AHEAD, on a stream of its own underneath the
    // previous step's rasters and views -- imgenv_hip.hip: sfm_ahead; a reset in between simply drops what was computed ahead)
    double *p_out, *v_out;
    int *dq_out, *dest_out, *last_out;
    SfmNode* nodes_out;
    int *n_nodes_out, *treehash_out;
};

#if defined(__HIPCC__)
// crowd k of a multi-world handle: the same record with every array moved on to its slice
__device__ __forceinline__ SfmDev sfm_of_world(const SfmDev& f, int k) {
    SfmDev g = f;
    const size_t n = (size_t)f.n, a = (size_t)k * n;
    g.p += a * 3; g.v += a * 3; g.vmax += a;
    g.wpx += a * SFM_MAX_WP; g.wpy += a * SFM_MAX_WP; g.wpr += a * SFM_MAX_WP;
    g.dq += a * SFM_MAX_WP; g.dq_n += a; g.dest += a; g.last += a;
    g.obs += (size_t)k * f.cap_obs * 4;
    g.nodes += (size_t)k * f.cap_nodes;
    g.n_nodes += k;
    g.treehash += a;
    g.p_out += a * 3; g.v_out += a * 3;
    g.dq_out += a * SFM_MAX_WP; g.dest_out += a; g.last_out += a;
    g.nodes_out += (size_t)k * f.cap_nodes;
    g.n_nodes_out += k;
    g.treehash_out += a;
    g.pair_f += (size_t)k * n * n * 3;
    g.pair_code += (size_t)k * n * n;
    g.g_nb += (size_t)k * SFM_MAX_AGENTS * (SFM_MAX_AGENTS / 32);
    g.g_sh += (size_t)k * 4 * SFM_MAX_AGENTS;
    if (f.n_obs_w) g.n_obs = f.n_obs_w[k];
    return g;
}
#endif

// ---- Ttree (ped_tree.cpp:18-137) on flat arrays, shared by the host (initial tree) and the device ----
SFM_HD inline int sfm_q_new(SfmNode* nodes, int* n_nodes, int cap, double x, double y, double w, double h) {
    if (*n_nodes >= cap) return -1;
    SfmNode& q = nodes[*n_nodes];
    q.x = x; q.y = y; q.w = w; q.h = h;
    q.isleaf = 1;
    q.n_agents = 0;
    q.child[0] = q.child[1] = q.child[2] = q.child[3] = -1;
    return (*n_nodes)++;
}
SFM_HD inline void sfm_set_insert(SfmNode& q, int a, int* err) {
    int lo = 0;
    while (lo < q.n_agents && q.agents[lo] < a) lo++;
    if (lo < q.n_agents && q.agents[lo] == a) return;
    if (q.n_agents >= SFM_LEAF_CAP) {
        *err = 1;
        return;
    }
    for (int k = q.n_agents; k > lo; k--) q.agents[k] = q.agents[k - 1];
    q.agents[lo] = a;
    q.n_agents++;
}
SFM_HD inline void sfm_set_erase(SfmNode& q, int a) {
    for (int k = 0; k < q.n_agents; k++)
        if (q.agents[k] == a) {
            for (int j = k; j + 1 < q.n_agents; j++) q.agents[j] = q.agents[j + 1];
            q.n_agents--;
            return;
        }
}

// Ttree::addAgent (ped_tree.cpp:65-96), recursion unrolled onto an explicit stack.  Work items are
// (node, agent); an item whose node is a leaf inserts and may split the leaf, pushing the former members.
// `work`: 2 * SFM_MAX_DEPTH * 8 ints for the explicit stack (the device passes LDS: a dynamically indexed private array
// would sit in scratch memory, ~1 us per push / pop); nullptr = a local array (host)
// `start`: the node the descent begins at -- the root, or a node the agent's descent is known to pass through (internal nodes
// never change once they exist, so a descent that was walked on an earlier state of the tree is still a prefix of today's)
// `rehomed`: one flag per agent; a split sets it to 3 for every member it re-homes (unless it is 2), so that a caller that
// decides ahead of time who has left its leaf can re-test exactly the agents whose leaf changed
// `p_old`, `turn`: Tagent::move (ped_agent.cpp:519-571) moves ONE agent and has the scene update the tree (scene->moveAgent(this),
// line 570) before the next agent moves -- so a split during agent `turn`'s update re-homes the leaf's members by where they are
// at that moment: agents up to `turn` at their new positions (p), the ones behind it still at their old ones (p_old, `old_stride`
// doubles per agent).  p_old = nullptr: every position from p (building the initial tree).
SFM_HD inline void sfm_add_agent(SfmNode* nodes, int* n_nodes, int cap, int* treehash, const double* p, int agent, int* err,
                                 int* work = nullptr, int start = 0, unsigned short* rehomed = nullptr, const double* p_old = nullptr,
                                 int old_stride = 3, int turn = 0x7FFFFFFF) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (the device always passes LDS: a 4 KB array per lane here, used or not, is 4 KB of scratch memory per lane for the whole
    // kernel -- 2 GB for a full chip of wavefronts, which the runtime hands out per dispatch through an interrupt to the host)
    int* st_node = work;
#else
    int local_stack[2 * SFM_MAX_DEPTH * 8];
    int* st_node = work ? work : local_stack;
#endif
    int* st_agent = st_node + SFM_MAX_DEPTH * 8;
    int sp = 0;
    st_node[sp] = start;
    st_agent[sp++] = agent;
    while (sp > 0) {
        --sp;
        const int node = st_node[sp], a = st_agent[sp];
        SfmNode& q = nodes[node];
        if (q.isleaf) {
            sfm_set_insert(q, a, err);
            treehash[a] = node;
            if (q.n_agents > 8) {  // split: addChildren + redistribute in set order
                const int c0 = sfm_q_new(nodes, n_nodes, cap, q.x, q.y, q.w / 2, q.h / 2);
                const int c1 = sfm_q_new(nodes, n_nodes, cap, q.x + q.w / 2, q.y, q.w / 2, q.h / 2);
                const int c2 = sfm_q_new(nodes, n_nodes, cap, q.x + q.w / 2, q.y + q.h / 2, q.w / 2, q.h / 2);
                const int c3 = sfm_q_new(nodes, n_nodes, cap, q.x, q.y + q.h / 2, q.w / 2, q.h / 2);
                if (c3 < 0) {
                    *err = 2;
                    return;
                }
                q.isleaf = 0;
                q.child[0] = c0; q.child[1] = c1; q.child[2] = c2; q.child[3] = c3;
                // the reference re-adds members in ascending set order, depth first; pushing them in
                // descending order onto the LIFO stack reproduces that order
                for (int k = q.n_agents - 1; k >= 0; k--) {
                    if (sp >= SFM_MAX_DEPTH * 8) {
                        *err = 3;
                        return;
                    }
                    st_node[sp] = node;
                    st_agent[sp++] = q.agents[k];
                    if (rehomed && rehomed[q.agents[k]] != 2) rehomed[q.agents[k]] = 3;  // (3: its leaf changed -- test it again)
                }
                q.n_agents = 0;
            }
        } else {
            const bool moved = !p_old || a <= turn;
            const double px = moved ? p[3 * a] : p_old[old_stride * a], py = moved ? p[3 * a + 1] : p_old[old_stride * a + 1];
            const double cx = q.x + q.w / 2, cy = q.y + q.h / 2;
            // order of the four non-exclusive tests: tree3, tree1, tree2, tree4 (LIFO: push reversed)
            int tgt[4], nt = 0;
            if ((px >= cx) && (py >= cy)) tgt[nt++] = q.child[2];
            if ((px <= cx) && (py <= cy)) tgt[nt++] = q.child[0];
            if ((px >= cx) && (py <= cy)) tgt[nt++] = q.child[1];
            if ((px <= cx) && (py >= cy)) tgt[nt++] = q.child[3];
            for (int k = nt - 1; k >= 0; k--) {
                if (sp >= SFM_MAX_DEPTH * 8) {
                    *err = 3;
                    return;
                }
                st_node[sp] = tgt[k];
                st_agent[sp++] = a;
            }
        }
    }
}

// Tscene::moveAgent -> Ttree::moveAgent (ped_tree.cpp:131-137); p: the crowd as it stands at this agent's turn
SFM_HD inline void sfm_move_agent(SfmNode* nodes, int* n_nodes, int cap, int* treehash, const double* p, int a, int* err,
                                  int* work = nullptr, int start = 0, unsigned short* rehomed = nullptr) {
    const int leaf = treehash[a];
    const SfmNode& q = nodes[leaf];
    const double px = p[3 * a], py = p[3 * a + 1];
    if ((px < q.x) || (px > (q.x + q.w)) || (py < q.y) || (py > (q.y + q.h))) {
        sfm_add_agent(nodes, n_nodes, cap, treehash, p, a, err, work, start, rehomed);  // scene->placeAgent(a): from the root
        sfm_set_erase(nodes[leaf], a);                            // erased from the OLD leaf, even if it is the new one
    }
}


// ---- Tscene::moveAgent for the whole crowd (ped_tree.cpp:131-137), in stages that k_sfm runs with thread i = agent i and a
// barrier in between, and tests/host/sfm_tree_check.cpp agent by agent against the literal loop (sfm_surgery_literal) ----
//
// Most of the moves commute.  What the reference fixes is the ORDER of the inserts and erases of one leaf (agent order; an
// agent's insert comes before its own erase): a leaf splits the moment an insert makes it 9 agents.  Every agent's two
// operations are known up front -- the leaf T its descent ends at, the node `old` its treehash points at, and whether each
// really changes a set (std::set: an insert of a member and an erase of a non-member do nothing) -- and stay what they are as
// long as neither leaf splits.  So the count of every leaf along the reference's order can be worked out ahead: a leaf whose
// count never passes 8 is QUIET, it cannot split whatever happens elsewhere (induction over the first split of the step), and
// what the step leaves in it does not depend on the order: its erases, then its inserts, are done by all agents at once, a
// lock per leaf (erases first: the count then never passes what the reference's order reaches either; an agent whose two
// operations hit the same leaf ends up outside it -- that is how libpedsim loses agents -- so it only erases).  Only the
// operations on the other leaves -- those an insert really takes to 9 -- are replayed by one thread, in agent order, with the
// splits and the re-tests of the agents a split re-homes as the reference does them.
// (round 3 counted `members + everybody who wants in` instead: in a crowd outside the tree's rectangle -- cfg-4: the tree covers
// y in [10, 20], the crowd lives below it -- EVERY agent has left its leaf every step and inserts + erases itself in the same
// edge leaf, so any stretch of the edge that holds nine agents looked unsafe and half the crowd was replayed serially: ~100 us.)
// What the assumption does not cover sends the whole step to the serial replay, which is the definition: a descent that is
// not unique (a position exactly on a centre line: the reference inserts into several children), a tree that lives in HBM, and
// a leaf about to split that holds an agent whose treehash points elsewhere (left behind by such a multiple insert: the split
// would re-home it and change what its own move does).
struct SfmSurgery {
    SfmNode* nodes;
    int* n_nodes;
    int cap_nodes;
    int* treehash;
    const double* p;      // [n][3] the new positions
    const double* p_old;  // [n][old_stride] the positions the step started from (sfm_add_agent: what a split sees of the agents behind the one that moves)
    int old_stride;
    int n;
    unsigned short* flag;         // [agent] 0 stays, 1 has left its leaf, 2 done ahead of the replay, 3 re-homed by a split (test again)
    unsigned short* leaf;         // [agent] the leaf its descent from the root ends at (0: not unique -- the replay starts at the root)
    unsigned short* need_serial;  // [1] the whole step is replayed literally
    unsigned short* ins_leaf;     // [agent] the leaf its insert really grows, 0xFFFF = none
    unsigned short* ers_leaf;     // [agent] the node its erase really shrinks
    unsigned short* ends;         // [agent] bit 0: its insert is done (or nothing to do), bit 1: its erase, bit 2: it had left its leaf
    unsigned short* loud;         // [node] leaves that are not quiet
    int* arrivals;                // [node] real inserts
    unsigned long long* todo;     // [(n + 63) / 64] the agents the replay has to visit
    int* work;                    // sfm_add_agent's stack
};
struct SfmMove {  // what an agent keeps between the stages (registers on the device)
    int T, old;
    bool mover, in_T, t_quiet, o_quiet, do_erase, do_insert;
    unsigned short il, el;
};

// Whether agent i left its leaf, and where the descent of its new position from the root ends: all agents at once, BEFORE anything moves (internal
// nodes never change once they exist, so a descent walked on an earlier state of the tree is a prefix of the real one; a
// dependent access per level, ~20 levels once the tree has grown: most of the serial section's time when it was walked there)
SFM_HD inline void sfm_surgery_descent(const SfmSurgery& c, int i) {
    const SfmNode* nodes = c.nodes;
    const SfmNode& q = nodes[c.treehash[i]];
    const double px = c.p[3 * i], py = c.p[3 * i + 1];
    const bool left = (px < q.x) || (px > (q.x + q.w)) || (py < q.y) || (py > (q.y + q.h));
    int node = 0;
    {   // (also for an agent that stays: a split may re-home it by its OLD position into a child its new one is outside of)
        for (int depth = 0; depth < SFM_MAX_DEPTH && !nodes[node].isleaf; depth++) {
            const SfmNode& t = nodes[node];
            const double cx = t.x + t.w / 2, cy = t.y + t.h / 2;
            int cnt = 0, nxt = 0;
            if ((px >= cx) && (py >= cy)) { cnt++; nxt = t.child[2]; }
            if ((px <= cx) && (py <= cy)) { cnt++; nxt = t.child[0]; }
            if ((px >= cx) && (py <= cy)) { cnt++; nxt = t.child[1]; }
            if ((px <= cx) && (py >= cy)) { cnt++; nxt = t.child[3]; }
            if (cnt != 1) {
                node = 0;
                break;
            }
            node = nxt;
        }
    }
    c.flag[i] = left ? 1 : 0;
    c.leaf[i] = (unsigned short)(node < 65536 ? node : 0);
}
SFM_HD inline bool sfm_is_member(const SfmNode& t, int a) {
    bool in = false;
    for (int k = 0; k < t.n_agents; k++) in |= t.agents[k] == a;
    return in;
}
// stage 1: agent i's two operations
SFM_HD inline SfmMove sfm_surgery_classify(const SfmSurgery& c, int i) {
    SfmMove m;
    m.mover = i < c.n && c.flag[i] != 0;
    m.T = m.mover ? (int)c.leaf[i] : 0;
    m.old = m.mover ? c.treehash[i] : 0;
    m.in_T = false;
    m.t_quiet = m.o_quiet = m.do_erase = m.do_insert = false;
    m.il = m.el = 0xFFFF;
    if (i < c.n && (c.leaf[i] == 0 || !c.nodes[c.leaf[i]].isleaf)) *c.need_serial = 1;  // (a descent that is not unique -- of an agent that stays, too)
    if (m.mover) {
        if (m.T == 0 || !c.nodes[m.T].isleaf) {
            *c.need_serial = 1;
        } else {
            m.in_T = sfm_is_member(c.nodes[m.T], i);
            const bool in_old = m.old == m.T ? m.in_T : sfm_is_member(c.nodes[m.old], i);
            if (!m.in_T) {
                m.il = (unsigned short)m.T;
#if defined(__HIP_DEVICE_COMPILE__)
                atomicAdd(&c.arrivals[m.T], 1);
#else
                c.arrivals[m.T] += 1;
#endif
            }
            if (m.old == m.T || in_old) m.el = (unsigned short)m.old;  // (behind its insert it is a member of T)
        }
    }
    c.ins_leaf[i] = m.il;
    c.ers_leaf[i] = m.el;
    return m;
}
// stage 2: the leaf's count at this agent's insert, along the reference's order (cheap bound first: every insert in front of every erase)
SFM_HD inline void sfm_surgery_count(const SfmSurgery& c, int i, const SfmMove& m) {
    if (m.il != 0xFFFF && c.nodes[m.T].n_agents + c.arrivals[m.T] > 8) {
        int cnt = c.nodes[m.T].n_agents + 1;
        for (int b = 0; b < i; b++) cnt += (c.ins_leaf[b] == m.il ? 1 : 0) - (c.ers_leaf[b] == m.il ? 1 : 0);
        if (cnt > 8) c.loud[m.T] = 1;
    }
}
// stage 3, per node: a leaf about to split must hold nobody whose treehash points elsewhere
SFM_HD inline void sfm_surgery_census(const SfmSurgery& c, int q) {
    if (!c.loud[q]) return;
    const SfmNode& t = c.nodes[q];
    for (int k = 0; k < t.n_agents; k++)
        if (c.treehash[t.agents[k]] != q) *c.need_serial = 1;
}
// stage 4: which of its operations agent i does itself, ahead of the replay (the caller: erases, a barrier, inserts -- under the leaf's lock)
SFM_HD inline void sfm_surgery_verdict(const SfmSurgery& c, int i, SfmMove& m) {
    const bool ok = m.mover && *c.need_serial == 0;
    m.t_quiet = ok && !c.loud[m.T];
    m.o_quiet = ok && !c.loud[m.old];
    // (same leaf at both ends: a member leaves, a non-member comes and goes -- either way only the erase counts)
    m.do_erase = m.o_quiet && m.el != 0xFFFF && (m.old != m.T || m.in_T);
    m.do_insert = m.t_quiet && m.il != 0xFFFF && m.old != m.T;
}
// stage 5: behind its own operations
SFM_HD inline void sfm_surgery_settle(const SfmSurgery& c, int i, const SfmMove& m) {
    if (m.t_quiet && m.o_quiet) {
        c.treehash[i] = m.T;
        c.flag[i] = 2;  // done: the replay passes it by (also after a split, when it re-tests the split leaf's members)
    }
    c.ends[i] = (unsigned short)((m.t_quiet ? 1 : 0) | (m.o_quiet ? 2 : 0) | (m.mover ? 4 : 0));
}
// stage 6, one thread: what is left, in agent order.  c.todo: one bit per agent whose flag is 1.
SFM_HD inline void sfm_surgery_replay(const SfmSurgery& c, int* lerr) {
    SfmNode* nodes = c.nodes;
    for (int wd = 0; wd < (c.n + 63) / 64 && *lerr == 0; wd++) {
        unsigned long long m = c.todo[wd];
        while (m && *lerr == 0) {
            int bit = 0;
            while (!((m >> bit) & 1ull)) bit++;
            const int a = 64 * wd + bit;
            m &= m - 1;
            const int flag = c.flag[a], e = c.ends[a], start = c.leaf[a], nn_before = *c.n_nodes;
            const int cur = c.treehash[a];
            bool moves = flag == 1;  // left its leaf, and the leaf is still the one that verdict was about: no second test
            if (flag == 3) {         // a split re-homed it: Ttree::moveAgent's test on the leaf it is in now
                const SfmNode& q = nodes[cur];
                const double px = c.p[3 * a], py = c.p[3 * a + 1];
                moves = (px < q.x) || (px > (q.x + q.w)) || (py < q.y) || (py > (q.y + q.h));
                // (the split put it into a child by where it stood BEFORE this step, sfm_add_agent: an agent that stays inside the
                // split leaf may well have left that child.  One that had left the split leaf has left every child of it: the
                // rectangles' corners are exact sums down to depth 48.  Should that ever fail while its operations have been done
                // ahead of its turn, the step is reported instead of replayed wrongly.)
                if (!moves && (e & 4) && *c.need_serial == 0) *lerr = 6;
            }
            if (moves && *lerr == 0) {
                if (e & 1) {  // a quiet leaf: the insert is done
                    c.treehash[a] = start;
                } else {
                    SfmNode& t = nodes[start];
                    if (start != 0 && t.isleaf && t.n_agents < 8) {  // room in the leaf its descent ended at: no stack, no split
                        sfm_set_insert(t, a, lerr);
                        c.treehash[a] = start;
                    } else {
                        sfm_add_agent(nodes, c.n_nodes, c.cap_nodes, c.treehash, c.p, a, lerr, c.work, start, c.flag, c.p_old, c.old_stride, a);
                    }
                }
                if (flag == 3 || !(e & 2)) sfm_set_erase(nodes[cur], a);  // erased from the OLD leaf, even if it is the new one (ped_tree.cpp:131-137)
            }
            if (*c.n_nodes != nn_before) {  // a split: its members behind this agent get their turn
                for (int b = a + 1; b < c.n; b++)
                    if (c.flag[b] == 3) c.todo[b >> 6] |= 1ull << (b & 63);
                m = c.todo[wd] & ~((2ull << bit) - 1ull);
            }
        }
    }
}
#if defined(__HIPCC__)
struct d3 {
    double x, y, z;
};
__device__ __forceinline__ d3 D3(double x, double y, double z) {
    d3 r;
    r.x = x; r.y = y; r.z = z;
    return r;
}
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return D3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return D3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 scaled(d3 a, double f) { return D3(f * a.x, f * a.y, f * a.z); }  // Tvector::scaled
__device__ __forceinline__ double len2(d3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ double len3(d3 a) {
    if ((a.x == 0) && (a.y == 0) && (a.z == 0)) return 0;
    return sqrt(len2(a));
}
__device__ __forceinline__ d3 normalized(d3 a) {
    const double l = len3(a);
    if (l == 0) return D3(0, 0, 0);
    return D3(a.x / l, a.y / l, a.z / l);
}
__device__ __forceinline__ double dot3(d3 a, d3 b) { return (a.x * b.x + a.y * b.y + a.z * b.z); }
__device__ __forceinline__ d3 ld3(const double* a, int i) { return D3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }

__device__ double sfm_angle_to(d3 a, d3 b) {  // Tvector::angleTo
    const double kPi = 3.14159265358979323846;
    // cr_atan2: see cr_atan2.h -- the sign of this difference between two nearly parallel vectors switches a
    // full-magnitude force term, so it must round like the reference's (glibc) atan2
    double diff = cr_atan2(b.y, b.x) - cr_atan2(a.y, a.x);
    if (diff > kPi)
        diff -= 2 * kPi;
    else if (diff <= -kPi)
        diff += 2 * kPi;
    return diff;
}

// one Tscene::moveAgents(h) for the whole crowd: thread i = agent i, one workgroup -- or, for crowds whose n^2 pair terms (three
// correctly rounded atan2 each) would keep that one workgroup busy for a millisecond, three launches:
//   phase 1 (one workgroup)     neighbour sets from the quadtree, desired force, the per-agent angles   -> s.g_nb, s.g_sh
//   phase 2 (n^2 / 256 workgroups) the pair terms of socialForce / lookaheadForce                      -> s.pair_f, s.pair_code
//   phase 3 (one workgroup)     sums in neighbour order, obstacle force, move, the serial quadtree surgery
// phase 0 = everything in one launch (small crowds).
__device__ void sfm_step(const SfmDev& s, double h, int phase, uint32_t* nb_lds /* LDS [SFM_MAX_AGENTS][SFM_MAX_AGENTS/32] */,
                         double* sh_lds /* LDS [4][SFM_MAX_AGENTS]: desired direction x / y and two angles per agent */,
                         unsigned short* stk /* LDS [SFM_WALK_CAP][blockDim.x]: walk stacks; later one "left its leaf" flag per agent */,
                         SfmNode* lnodes /* LDS [SFM_LDS_NODES] */, int* lhash /* LDS [SFM_MAX_AGENTS] */, int* ln_nodes /* LDS [1] */,
                         int pblock /* phase 2: this workgroup of the crowd's pblocks */, int pblocks,
                         unsigned long long* stamp = nullptr /* debug: wall-clock marks of thread 0 */) {
#define SFM_STAMP(q) do { if (stamp && threadIdx.x == 0) stamp[q] = wall_clock64(); } while (0)
    SFM_STAMP(0);
    if (phase == 1) SFM_STAMP(12);
    const int i = threadIdx.x;
    const int n = s.n;
    const int n_cap = SFM_MAX_AGENTS;
    // The tree is pointer-chased many times per step and then edited agent by agent: for the step it lives in LDS (when it
    // fits with room for a few splits), so a dependent access costs ~100 ns instead of ~1 us.
    uint32_t* nb_bits = phase == 0 ? nb_lds : s.g_nb;  // split: what one launch hands the next lives in HBM
    double* sh = phase == 0 ? sh_lds : s.g_sh;
    double* lp = (double*)nb_lds;  // [n][3] new positions; the LDS neighbour bit sets (8 KB) are dead once the pair terms exist
    const int n_nodes0 = *s.n_nodes;
    const bool in_lds = phase != 2 && n_nodes0 + 32 <= SFM_LDS_NODES;
    // (a tree too large for LDS is edited in HBM: in the OUTPUT copy, made first, when the step does not happen in place)
    const bool edits_tree = phase == 0 || phase == 3, hbm_copy = !in_lds && edits_tree && s.nodes_out != s.nodes;
    if (hbm_copy) {
        const int words = n_nodes0 * (int)(sizeof(SfmNode) / 4);
        for (int q = threadIdx.x; q < words; q += blockDim.x) ((uint32_t*)s.nodes_out)[q] = ((const uint32_t*)s.nodes)[q];
        if (i < n) s.treehash_out[i] = s.treehash[i];
        if (i == 0) *s.n_nodes_out = n_nodes0;
        __threadfence_block();
        __syncthreads();
    }
    SfmNode* nodes = in_lds ? lnodes : hbm_copy ? s.nodes_out : s.nodes;
    int* treehash = in_lds ? lhash : hbm_copy ? s.treehash_out : s.treehash;
    int* n_nodes = in_lds ? ln_nodes : hbm_copy ? s.n_nodes_out : s.n_nodes;
    const int cap_nodes = in_lds ? SFM_LDS_NODES : s.cap_nodes;
    if (in_lds) {
        const int words = n_nodes0 * (int)(sizeof(SfmNode) / 4);
        for (int q = threadIdx.x; q < words; q += blockDim.x) ((uint32_t*)lnodes)[q] = ((const uint32_t*)s.nodes)[q];
        if (i < n) lhash[i] = s.treehash[i];
        if (i == 0) *ln_nodes = n_nodes0;
        __syncthreads();
    }
    if (phase == 1) SFM_STAMP(13);
    // Tscene::getNeighbors asks for the agents of every leaf a 40 m square touches, in a scene of 10 m x 10 m: for an agent
    // anywhere near the scene that is EVERY leaf, and the walk (a visit of every node, per agent: most of this phase's time once
    // the tree has grown) returns the same set for all of them -- the agents that are in the tree at all.  That set is gathered
    // once, every thread a few nodes; only an agent whose square does not contain the root's rectangle walks.
    uint32_t* in_tree = (uint32_t*)(stk + (size_t)(SFM_WALK_CAP - 1) * blockDim.x);  // [8]: the last row of the walk stacks' LDS
    // ... and the leaves that hold anybody are listed (in any order) for the agents whose square covers only part of the tree
    // (in the LDS angle table, 8 KB, which is written behind the walk)
    uint32_t* full_mask = (uint32_t*)sh_lds;                                                 // [SFM_FULL_CAP][SFM_MAX_AGENTS / 32] its members, one bit per agent
    unsigned short* full_leaf = (unsigned short*)(full_mask + SFM_FULL_CAP * (SFM_MAX_AGENTS / 32));  // [SFM_FULL_CAP] the leaf's node
    int* n_full = (int*)(full_leaf + SFM_FULL_CAP);  // [0] entries, [1] a leaf thinner than 1e-12 m exists, [2] more leaves than the list holds
    static_assert(SFM_FULL_CAP % 2 == 0 && SFM_FULL_CAP * (4 * (SFM_MAX_AGENTS / 32) + 2) + 12 <= 8 * 4 * SFM_MAX_AGENTS, "the list of occupied leaves lives in the LDS angle table");
    if (phase <= 1) {
        if (i < SFM_MAX_AGENTS / 32) in_tree[i] = 0;
        if (i < 3) n_full[i] = 0;
        __syncthreads();
        const int nn_now = *n_nodes;
        for (int q = threadIdx.x; q < nn_now; q += blockDim.x) {
            const SfmNode& t = nodes[q];
            if (t.isleaf) {
                for (int k = 0; k < t.n_agents; k++) atomicOr(&in_tree[t.agents[k] >> 5], 1u << (t.agents[k] & 31));
                if (t.w < 1e-12 || t.h < 1e-12) n_full[1] = 1;
                if (t.n_agents > 0 && q > 0) {
                    const int slot = atomicAdd(&n_full[0], 1);
                    if (slot < SFM_FULL_CAP && q < 65536) {
                        full_leaf[slot] = (unsigned short)q;
                        uint32_t* m = full_mask + slot * (SFM_MAX_AGENTS / 32);
                        for (int k = 0; k < SFM_MAX_AGENTS / 32; k++) m[k] = 0u;
                        for (int k = 0; k < t.n_agents; k++) m[t.agents[k] >> 5] |= 1u << (t.agents[k] & 31);
                    } else {
                        n_full[2] = 1;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (phase == 1) SFM_STAMP(14);
    d3 desiredforce = D3(0, 0, 0), socialforce = D3(0, 0, 0), obstacleforce = D3(0, 0, 0), lookaheadforce = D3(0, 0, 0);
    d3 me_p = D3(0, 0, 0), me_v = D3(0, 0, 0);
    uint32_t* mine = nb_bits + (size_t)i * (SFM_MAX_AGENTS / 32);
    if (i < n && phase == 3) {  // what phase 1 had in registers
        me_p = ld3(s.p, i);
        me_v = ld3(s.v, i);
        desiredforce = scaled(normalized(D3(sh[i], sh[n_cap + i], 0)), s.vmax[i]);
    }
    if (i < n && phase <= 1) {
        me_p = ld3(s.p, i);
        me_v = ld3(s.v, i);
        // Tscene::getNeighbors(p.x, p.y, 20) (ped_scene.cpp:217-252): agents of the leaves the square touches
        const SfmNode& root = nodes[0];
        const bool whole_tree = ((me_p.x + 20.0) > (root.x + root.w)) && ((me_p.x - 20.0) < root.x) && ((me_p.y + 20.0) > (root.y + root.h)) &&
                                ((me_p.y - 20.0) < root.y);  // the square contains the root's rectangle: every child test below passes
        for (int k = 0; k < SFM_MAX_AGENTS / 32; k++) mine[k] = whole_tree ? in_tree[k] : 0u;
        // An agent whose square does not cover the whole tree -- 20 m beyond an edge of it, or merely below y = 0 with the tree's
        // top row out of reach -- gets the agents of the leaves its square touches.  The reference walks down from the root and
        // tests every child's rectangle; a leaf's ancestors contain it, so (the rectangles' corners being exact sums of halves
        // down to depth ~48) it reaches exactly the leaves whose OWN rectangle passes the test: every thread scans the node array
        // thread scans the list of the leaves that hold anybody, in step with the others (the same leaf for the whole wavefront:
        // broadcast reads, no stack, no pointer chase -- the walk itself was ~0.4 us per node for the slowest agent of the crowd:
        // 80 of the first launch's 90 us in cfg-4, where a few pedestrians always stand below y = 0).  Leaves thinner than
        // 1e-12 m: the literal walk below.
        bool walk = false;
        if (!whole_tree) {
            if (root.isleaf) {  // (the walk pops the root without a test)
                for (int k = 0; k < root.n_agents; k++) mine[root.agents[k] >> 5] |= 1u << (root.agents[k] & 31);
            } else if (n_full[1] || n_full[2]) {
                walk = true;
            } else {
                uint32_t acc[SFM_MAX_AGENTS / 32];
#pragma unroll
                for (int k = 0; k < SFM_MAX_AGENTS / 32; k++) acc[k] = 0u;
                const int n_list = n_full[0];
                for (int e = 0; e < n_list; e++) {
                    const SfmNode& t = nodes[full_leaf[e]];
                    const double tx = t.x, ty = t.y, tw = t.w, th = t.h;  // (fetched together with the members' bits, in front of the test)
                    const uint32_t* m = full_mask + e * (SFM_MAX_AGENTS / 32);
                    const bool touched = ((me_p.x + 20.0) > tx) && ((me_p.x - 20.0) < (tx + tw)) && ((me_p.y + 20.0) > ty) && ((me_p.y - 20.0) < (ty + th));
#pragma unroll
                    for (int k = 0; k < SFM_MAX_AGENTS / 32; k++) acc[k] |= touched ? m[k] : 0u;
                }
                if (!walk) {
#pragma unroll
                    for (int k = 0; k < SFM_MAX_AGENTS / 32; k++) mine[k] = acc[k];
                }
            }
        }
        // depth-first walk with the stack in LDS (a dynamically indexed private array would live in scratch memory):
        // entry k of thread i sits at stk[k * blockDim.x + i]  (SFM_WALK_CAP - 1 entries: the last row holds in_tree)
        int sp = 0;
        if (walk) stk[(sp++) * blockDim.x + i] = 0;
        while (sp > 0) {
            const SfmNode& t = nodes[stk[(--sp) * blockDim.x + i]];
            if (t.isleaf) {
                for (int k = 0; k < t.n_agents; k++) mine[t.agents[k] >> 5] |= 1u << (t.agents[k] & 31);
            } else {
                for (int c = 0; c < 4; c++) {
                    const SfmNode& ch = nodes[t.child[c]];
                    if (((me_p.x + 20.0) > ch.x) && ((me_p.x - 20.0) < (ch.x + ch.w)) && ((me_p.y + 20.0) > ch.y) &&
                        ((me_p.y - 20.0) < (ch.y + ch.h))) {
                        if (sp < SFM_WALK_CAP - 1) stk[(sp++) * blockDim.x + i] = (unsigned short)t.child[c];
                        else *s.err = 4;
                    }
                }
            }
        }
        SFM_STAMP(1);
        // Tagent::desiredForce (ped_agent.cpp:236-306)
        int dest = s.dest[i], last = s.last[i];
        const int dqn = s.dq_n[i];
        const int* dq = s.dq + (size_t)i * SFM_MAX_WP;
        int* dq_o = s.dq_out + (size_t)i * SFM_MAX_WP;
        if ((dest == -1) && (dqn > 0)) {
            dest = dq[0];
            for (int k = 0; k + 1 < dqn; k++) dq_o[k] = dq[k + 1];  // (in place as well: entry k + 1 is read before it is overwritten)
            dq_o[dqn - 1] = dest;
        } else if (dq_o != dq) {
            for (int k = 0; k < dqn; k++) dq_o[k] = dq[k];
        }
        d3 desired_direction = D3(0, 0, 0);
        bool reached = false;
        if (dest != -1) {
            const d3 diff = D3(s.wpx[(size_t)i * SFM_MAX_WP + dest] - me_p.x, s.wpy[(size_t)i * SFM_MAX_WP + dest] - me_p.y, 0);
            reached = len3(diff) < s.wpr[(size_t)i * SFM_MAX_WP + dest];
            desired_direction = normalized(diff);
        }
        if ((dest != -1) && reached) {
            last = dest;
            dest = -1;
        }
        s.dest_out[i] = dest;
        s.last_out[i] = last;
        desiredforce = scaled(normalized(desired_direction), s.vmax[i]);
        SFM_STAMP(2);
        // loop-invariant angles of Tagent::lookaheadForce (ped_agent.cpp:439-480): of my desired direction and of my
        // velocity as somebody else's neighbour
        sh[i] = desired_direction.x;
        sh[n_cap + i] = desired_direction.y;
        sh[2 * n_cap + i] = cr_atan2(-desired_direction.x, -desired_direction.y);
        sh[3 * n_cap + i] = cr_atan2(-me_v.x, -me_v.y);
    }
    if (phase == 1) SFM_STAMP(15);
    if (phase == 1) return;
    __syncthreads();
    // Pair terms of lookaheadForce and socialForce (ped_agent.cpp:316-404, 439-480), one (agent, neighbour) pair per
    // thread and round: the three correctly rounded atan2 of a pair are ~1500 serial instructions, and an agent has up to
    // n - 1 neighbours.  Each term is evaluated exactly as the reference does and parked in HBM; the agents then add
    // their terms in neighbour order, so the sums round as the sequential loops do.
    for (int pq = (phase == 2 ? pblock * blockDim.x : 0) + threadIdx.x; phase != 3 && pq < n * n;
         pq += (phase == 2 ? pblocks : 1) * blockDim.x) {
        const int pi = pq / n, o = pq - pi * n;
        unsigned char code = 0;  // bits 0-1: lookahead vote + 1, bit 2: has a social term
        d3 term = D3(0, 0, 0);
        const uint32_t* bits = nb_bits + (size_t)pi * (SFM_MAX_AGENTS / 32);
        if (o != pi && ((bits[o >> 5] >> (o & 31)) & 1u)) {
            const d3 pp = ld3(s.p, pi), po = ld3(s.p, o);
            {
                const double pi_c = 3.14159265;
                int vote = 0;
                const double dx = po.x - pp.x, dy = po.y - pp.y;
                const double dist2 = (dx * dx + dy * dy);
                if (dist2 < 400) {
                    const double at2v = sh[2 * n_cap + pi];
                    const double at2d = cr_atan2(-dx, -dy);
                    const double at2v2 = sh[3 * n_cap + o];
                    double sd = at2d - at2v;
                    if (sd > pi_c) sd -= 2 * pi_c;
                    if (sd < -pi_c) sd += 2 * pi_c;
                    double vv = at2v - at2v2;
                    if (vv > pi_c) vv -= 2 * pi_c;
                    if (vv < -pi_c) vv += 2 * pi_c;
                    if (fabs(vv) > 2.5) {
                        if ((sd < 0) && (sd > -0.3)) vote--;
                        if ((sd > 0) && (sd < 0.3)) vote++;
                    }
                }
                code = (unsigned char)(vote + 1);
            }
            {
                const double lambda_importance = 2.0, gamma = 0.35, nn = 2, n_prime = 3;
                const d3 diff = po - pp;
                if (!(len2(diff) > 64.0)) {
                    const d3 diff_direction = normalized(diff);
                    const d3 vel_diff = ld3(s.v, pi) - ld3(s.v, o);
                    const d3 interaction_vector = scaled(vel_diff, lambda_importance) + diff_direction;
                    const double interaction_length = len3(interaction_vector);
                    const d3 interaction_direction = scaled(interaction_vector, 1 / interaction_length);
                    const double theta = sfm_angle_to(interaction_direction, diff_direction);
                    const int theta_sign = (theta == 0) ? (0) : (int)(theta / fabs(theta));
                    const double B = gamma * interaction_length;
                    const double fva = -exp(-len3(diff) / B - (n_prime * B * theta) * (n_prime * B * theta));
                    const double faa = -theta_sign * exp(-len3(diff) / B - (nn * B * theta) * (nn * B * theta));
                    const d3 force_velocity = scaled(interaction_direction, fva);
                    const d3 force_angle = scaled(D3(-interaction_direction.y, interaction_direction.x, 0), faa);
                    term = force_velocity + force_angle;
                    code |= 4;
                }
            }
        } else {
            code = 1;
        }
        // stored neighbour-major ([o][agent]): the sums below read agent i's terms one neighbour at a time, thread i = agent i --
        // consecutive threads then read consecutive words (agent-major, every thread walked a 4.8 KB row of its own: 64 cache
        // lines per load instruction, 72 us of the step at 200 agents)
        const size_t at = (size_t)o * n + pi;
        s.pair_code[at] = code;
        if (code & 4) {
            s.pair_f[3 * at] = term.x;
            s.pair_f[3 * at + 1] = term.y;
            s.pair_f[3 * at + 2] = term.z;
        }
    }
    if (phase == 2) return;
    __threadfence_block();
    __syncthreads();
    SFM_STAMP(3);
    if (i < n) {
        {
            int count = 0;
            const d3 e = D3(sh[i], sh[n_cap + i], 0);
            // the terms are added in neighbour order (the sums round as the reference's loop does), but their loads do not
            // depend on each other: eight neighbours' codes and terms are fetched at once, then added one by one.  Only the
            // agent's neighbours are visited (everybody else's code says "no vote, no term"): in ascending order out of its bit
            // set, a word at a time -- a crowd outside the tree's rectangle has a fifth of its agents in the tree at any time
            // (cfg-4: 25 rounds of loads per agent -> 7; 29 -> 9 us)
            const unsigned char* codes = s.pair_code + i;   // [o][agent]
            const double* terms = s.pair_f + 3 * (size_t)i;
#pragma unroll
            for (int k = 0; k < SFM_MAX_AGENTS / 32; k++) {
                uint32_t wb = mine[k] & ~(k == (i >> 5) ? 1u << (i & 31) : 0u);  // (its own entry carries neither a vote nor a term: the pair loop above)
                while (wb) {
                    unsigned char cd[8];
                    d3 tm[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const bool has = wb != 0;
                        const size_t o = (size_t)(has ? 32 * k + __ffs((int)wb) - 1 : 0) * n;
                        wb &= wb - 1;  // (0 stays 0)
                        cd[u] = has ? codes[o] : (unsigned char)1;  // 1: no vote, no term
                        tm[u] = D3(terms[3 * o], terms[3 * o + 1], terms[3 * o + 2]);  // (garbage where the code has no term: unused)
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        count += (int)(cd[u] & 3) - 1;
                        if (cd[u] & 4) socialforce = socialforce + tm[u];
                    }
                }
            }
            if (count < 0) {
                lookaheadforce.x = 0.5f * e.y;
                lookaheadforce.y = 0.5f * -e.x;
            }
            if (count > 0) {
                lookaheadforce.x = 0.5f * -e.y;
                lookaheadforce.y = 0.5f * e.x;
            }
        }
        // Tagent::obstacleForce (ped_agent.cpp:411-429)
        {
            d3 min_diff = D3(0, 0, 0);
            double min_d2 = __builtin_huge_val();
            for (int q = 0; q < s.n_obs; q++) {
                const double* o = s.obs + 4 * q;
                const d3 start = D3(o[0], o[1], 0), end = D3(o[2], o[3], 0);
                const d3 rel_end = end - start;
                const d3 rel_p = me_p - start;
                const double lambda = dot3(rel_p, rel_end) / len2(rel_end);
                d3 closest;
                if (lambda <= 0)
                    closest = start;
                else if (lambda >= 1)
                    closest = end;
                else
                    closest = start + scaled(rel_end, lambda);
                const d3 diff = me_p - closest;
                const double d2 = len2(diff);
                if (d2 < min_d2) {
                    min_d2 = d2;
                    min_diff = diff;
                }
            }
            const double distance = sqrt(min_d2) - 0.2;
            const double force_amount = exp(-distance / 0.8);
            obstacleforce = scaled(normalized(min_diff), force_amount);
        }
    }
    __syncthreads();  // all forces are computed from the t-1 state (ped_scene.cpp:170)
    SFM_STAMP(4);
    if (i < n) {
        // Tagent::move (ped_agent.cpp:519-571)
        d3 p_desired = me_p + scaled(me_v, h);
        for (int q = 0; q < s.n_obs; q++) {
            const double* o = s.obs + 4 * q;
            const double s1x = p_desired.x - me_p.x, s1y = p_desired.y - me_p.y;
            const double s2x = o[2] - o[0], s2y = o[3] - o[1];
            const double ss = (-s1y * (me_p.x - o[0]) + s1x * (me_p.y - o[1])) / (-s2x * s1y + s1x * s2y);
            const double tt = (s2x * (me_p.y - o[1]) - s2y * (me_p.x - o[0])) / (-s2x * s1y + s1x * s2y);
            if (ss >= 0 && ss <= 1 && tt >= 0 && tt <= 1) {
                const d3 inter = D3(me_p.x + (tt * s1x), me_p.y + (tt * s1y), 0);
                p_desired = inter - scaled(normalized(scaled(me_v, h)), 0.1);
            }
        }
        const d3 a = (((scaled(desiredforce, 1.0) + scaled(socialforce, 2.1)) + scaled(obstacleforce, 1.0)) +
                      scaled(lookaheadforce, 1.0)) + D3(0, 0, 0);
        d3 v = scaled(me_v, 0.5) + scaled(a, h);
        if (len3(v) > s.vmax[i]) v = scaled(normalized(v), s.vmax[i]);
        s.p_out[3 * i] = p_desired.x; s.p_out[3 * i + 1] = p_desired.y; s.p_out[3 * i + 2] = p_desired.z;
        lp[3 * i] = p_desired.x;  // the tree surgery below reads positions from LDS
        lp[3 * i + 1] = p_desired.y;
        double* lp_old = (double*)(stk + 11 * blockDim.x);  // [n][2]: rows 11-18 of the walk stacks' LDS (dead since the neighbour walk)
        lp_old[2 * i] = me_p.x;
        lp_old[2 * i + 1] = me_p.y;
        s.v_out[3 * i] = v.x; s.v_out[3 * i + 1] = v.y; s.v_out[3 * i + 2] = v.z;
    }
    __syncthreads();
    SFM_STAMP(5);
    // scene->moveAgent(this) in agent order (ped_tree.cpp:131-137): sfm_surgery_* above (shared with tests/host/sfm_tree_check.cpp,
    // which runs the same functions agent by agent against the literal loop).  Here: thread i = agent i, a barrier between the stages.
    SfmSurgery sg;
    sg.nodes = nodes; sg.n_nodes = n_nodes; sg.cap_nodes = cap_nodes; sg.treehash = treehash; sg.p = lp; sg.n = n;
    sg.p_old = (const double*)(stk + 11 * blockDim.x); sg.old_stride = 2;
    sg.flag = stk;                          // rows of the walk stacks' LDS, [blockDim.x] each
    sg.leaf = stk + blockDim.x;
    sg.need_serial = stk + 2 * blockDim.x;
    sg.ins_leaf = stk + 3 * blockDim.x;
    sg.ers_leaf = stk + 4 * blockDim.x;
    sg.ends = stk + 5 * blockDim.x;
    sg.loud = stk + 6 * blockDim.x;         // [SFM_LDS_NODES]: rows 6-9
    sg.todo = (unsigned long long*)(stk + 10 * blockDim.x);
    sg.arrivals = (int*)sh_lds;             // [SFM_LDS_NODES] (the LDS angle table, 8 KB, is free by now)
    sg.work = (int*)sh_lds;                 // (sfm_add_agent's stack, in the serial replay: the counters are dead by then)
    int* leaf_lock = sg.arrivals + SFM_LDS_NODES;  // [SFM_LDS_NODES]
    static_assert(SFM_WALK_CAP >= 19 && SFM_LDS_NODES <= 4 * SFM_MAX_AGENTS && SFM_LDS_NODES < 0xFFFF, "rows of the walk stacks reused by the tree surgery");
    if (i < n) sfm_surgery_descent(sg, i);
    else if (i < (int)blockDim.x) sg.flag[i] = 0;
    __syncthreads();
    SFM_STAMP(7);
    if (in_lds) {
        for (int q = threadIdx.x; q < 2 * SFM_LDS_NODES; q += blockDim.x) sg.arrivals[q] = 0;
        for (int q = threadIdx.x; q < SFM_LDS_NODES; q += blockDim.x) sg.loud[q] = 0;
        if (i == 0) *sg.need_serial = 0;
        __syncthreads();
        SfmMove mv = sfm_surgery_classify(sg, i);
        __syncthreads();
        sfm_surgery_count(sg, i, mv);
        __syncthreads();
        {
            const int nn_now = *n_nodes;
            for (int q = threadIdx.x; q < nn_now; q += blockDim.x) sfm_surgery_census(sg, q);
        }
        __syncthreads();
        sfm_surgery_verdict(sg, i, mv);
        int lerr = 0;
        // (the loops run until the whole wavefront is through: with a per-lane exit the compiler may sink the critical section
        // behind the loop, where a lane that holds the lock waits for the lanes that spin on it)
        bool done = !mv.do_erase;
        for (int spin = 0; !__all(done) && spin < (1 << 20); spin++) {
            if (!done) {
                int expected = 0;
                if (__hip_atomic_compare_exchange_strong(&leaf_lock[mv.old], &expected, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    sfm_set_erase(nodes[mv.old], i);
                    __hip_atomic_store(&leaf_lock[mv.old], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    done = true;
                }
            }
        }
        if (mv.do_erase && !done) lerr = 5;
        __syncthreads();  // every erase in front of every insert: no leaf ever holds more than it does in the reference's order
        done = !mv.do_insert;
        for (int spin = 0; !__all(done) && spin < (1 << 20); spin++) {
            if (!done) {
                int expected = 0;
                if (__hip_atomic_compare_exchange_strong(&leaf_lock[mv.T], &expected, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    sfm_set_insert(nodes[mv.T], i, &lerr);
                    __hip_atomic_store(&leaf_lock[mv.T], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    done = true;
                }
            }
        }
        if (mv.do_insert && !done) lerr = 5;
        sfm_surgery_settle(sg, i, mv);
        if (lerr) *s.err = lerr;
        const unsigned long long m = __ballot(i < n && sg.flag[i] == 1);
        if ((threadIdx.x & 63) == 0) sg.todo[threadIdx.x >> 6] = m;
        __syncthreads();
        if (stamp && threadIdx.x == 0) {  // (profile builds: how many agents are left to the serial replay, and why)
            stamp[10] = (unsigned long long)(__popcll(sg.todo[0]) + __popcll(sg.todo[1]) + __popcll(sg.todo[2]) + __popcll(sg.todo[3]));
            stamp[11] = *sg.need_serial;
        }
    }
    else {  // the tree in HBM: every agent that left its leaf is replayed, as in a step with a tie
        if (i == 0) *sg.need_serial = 1;
        sg.ends[i] = (unsigned short)(i < n && sg.flag[i] ? 4 : 0);
        const unsigned long long m = __ballot(i < n && sg.flag[i] == 1);
        if ((threadIdx.x & 63) == 0) sg.todo[threadIdx.x >> 6] = m;
        __syncthreads();
    }
    SFM_STAMP(8);
    if (i == 0) {
        int lerr = 0;  // s.err is page-locked host memory: touched only to report
        sfm_surgery_replay(sg, &lerr);
        if (lerr) *s.err = lerr;
    }
    __syncthreads();
    SFM_STAMP(9);
    if (in_lds) {  // back to HBM for the next step (a reset only moves positions, as Tagent::setPosition does: the tree catches up in the next step)
        const int words = *ln_nodes * (int)(sizeof(SfmNode) / 4);
        for (int q = threadIdx.x; q < words; q += blockDim.x) ((uint32_t*)s.nodes_out)[q] = ((const uint32_t*)lnodes)[q];
        if (i < n) s.treehash_out[i] = lhash[i];
        if (i == 0) *s.n_nodes_out = *ln_nodes;
    }
    SFM_STAMP(6);
#undef SFM_STAMP
}
#endif
