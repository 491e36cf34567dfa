// sfm_tree_check.cpp -- the staged quadtree surgery of k_sfm (img_env_amd/csrc/sfm.h: sfm_surgery_*) against the literal loop
// (Ttree::moveAgent for every agent in agent order, ped_tree.cpp:131-137), on the CPU.  The very functions the kernel calls with
// thread i = agent i are called here agent by agent -- the operations the kernel does under a lock per leaf in SHUFFLED order -- on
// crowds that random-walk with a drift through / below / across libpedsim's 10 m root square, optionally snapped onto the tree's
// own centre lines (ties: the reference inserts such an agent into several children).  After every step the two trees must be the
// same tree: node count, every leaf's rectangle and members, every agent's treehash rectangle.
//
//   sfm_tree_check <seed> <agents> <steps> <mode: 0 inside the square, 1 below it (cfg-4), 2 across its lower edge, 3 inside with ties>
//   sfm_tree_check 0 <agents> <steps> 9 <file>   positions from a file of doubles: [agents][2] the tree's initial positions, then
//                                                [agents][2] the crowd at the reset, [steps][agents][2] the crowd after each step
//                                                (a recorded run of the oracle or of the library: tools/sfm_tree_debug.py)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "../../img_env_amd/csrc/sfm.h"

struct Tree {
    std::vector<SfmNode> nodes;
    std::vector<int> hash;
    int n_nodes = 0;
};

typedef std::array<double, 4> Rect;
static Rect rect_of(const SfmNode& q) { return Rect{q.x, q.y, q.w, q.h}; }

static bool same_tree(const Tree& a, const Tree& b, int n, int step) {
    if (a.n_nodes != b.n_nodes) {
        printf("step %d: %d nodes against %d (literal)\n", step, b.n_nodes, a.n_nodes);
        return false;
    }
    std::map<Rect, std::vector<int>> la, lb;
    for (int k = 0; k < a.n_nodes; k++)
        if (a.nodes[k].isleaf && a.nodes[k].n_agents) la[rect_of(a.nodes[k])] = std::vector<int>(a.nodes[k].agents, a.nodes[k].agents + a.nodes[k].n_agents);
    for (int k = 0; k < b.n_nodes; k++)
        if (b.nodes[k].isleaf && b.nodes[k].n_agents) lb[rect_of(b.nodes[k])] = std::vector<int>(b.nodes[k].agents, b.nodes[k].agents + b.nodes[k].n_agents);
    if (la != lb) {
        printf("step %d: leaf members differ\n", step);
        for (auto& kv : la)
            if (!lb.count(kv.first) || lb[kv.first] != kv.second) {
                printf("  leaf [%g %g %g %g] literal:", kv.first[0], kv.first[1], kv.first[2], kv.first[3]);
                for (int v : kv.second) printf(" %d", v);
                printf("  staged:");
                if (lb.count(kv.first)) for (int v : lb[kv.first]) printf(" %d", v);
                printf("\n");
            }
        for (auto& kv : lb)
            if (!la.count(kv.first)) {
                printf("  leaf [%g %g %g %g] staged only:", kv.first[0], kv.first[1], kv.first[2], kv.first[3]);
                for (int v : kv.second) printf(" %d", v);
                printf("\n");
            }
        return false;
    }
    for (int i = 0; i < n; i++)
        if (rect_of(a.nodes[a.hash[i]]) != rect_of(b.nodes[b.hash[i]])) {
            printf("step %d: treehash of agent %d differs\n", step, i);
            return false;
        }
    return true;
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const unsigned seed = (unsigned)atoi(argv[1]);
    const int n = atoi(argv[2]), steps = atoi(argv[3]), mode = atoi(argv[4]);
    const int cap = 16384, W = 256;
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::normal_distribution<double> N(0.0, 1.0);
    std::vector<double> p((size_t)n * 3, 0.0), goal((size_t)n * 2), recorded;
    if (mode == 9) {
        if (argc < 6) return 2;
        FILE* f = fopen(argv[5], "rb");
        if (!f) return 2;
        recorded.resize((size_t)(steps + 2) * n * 2);
        const size_t got = fread(recorded.data(), sizeof(double), recorded.size(), f);
        fclose(f);
        if (got != recorded.size()) {
            printf("short file\n");
            return 2;
        }
    }
    const double y_lo = mode == 1 ? 0.5 : mode == 2 ? 8.0 : 10.3, y_hi = mode == 1 ? 9.5 : mode == 2 ? 12.0 : 19.7;
    auto new_goal = [&](int i) {
        // a few meeting points: crowds pile up there and leaves split
        const int g = (int)(U(rng) * 3);
        goal[2 * i] = 1.5 + 3.5 * g + 0.4 * N(rng);
        goal[2 * i + 1] = y_lo + (y_hi - y_lo) * (0.2 + 0.3 * g) + 0.4 * N(rng);
    };
    auto snap = [&](double v) { return std::round(v / 0.625) * 0.625; };  // the tree's centre lines down to level 4
    for (int i = 0; i < n; i++) {
        p[3 * i] = U(rng) * 10.0;
        p[3 * i + 1] = U(rng) * 10.0;  // (PedScene's own start: rand() positions in [0, 10]^2, i.e. BELOW the tree's rectangle)
        if (mode == 9) {
            p[3 * i] = recorded[2 * i];
            p[3 * i + 1] = recorded[2 * i + 1];
        }
        new_goal(i);
    }
    Tree lit, stg;
    lit.nodes.resize(cap);
    lit.hash.assign(n, 0);
    int err = 0;
    sfm_q_new(lit.nodes.data(), &lit.n_nodes, cap, 0, 10, 10, 10);
    for (int a = 0; a < n; a++) sfm_add_agent(lit.nodes.data(), &lit.n_nodes, cap, lit.hash.data(), p.data(), a, &err);
    if (err) {
        printf("initial tree overflowed (%d)\n", err);
        return 1;
    }
    stg = lit;
    // a reset: everybody somewhere in the mode's band
    for (int i = 0; i < n; i++) {
        p[3 * i] = mode == 9 ? recorded[(size_t)n * 2 + 2 * i] : 0.3 + 9.4 * U(rng);
        p[3 * i + 1] = mode == 9 ? recorded[(size_t)n * 2 + 2 * i + 1] : y_lo + (y_hi - y_lo) * U(rng);
        if (mode == 3 && U(rng) < 0.3) {  // (one coordinate only: nine agents on ONE point recurse forever, in the reference too)
            if (U(rng) < 0.5) p[3 * i] = snap(p[3 * i]);
            else p[3 * i + 1] = 10.0 + snap(p[3 * i + 1] - 10.0);
        }
    }
    std::vector<unsigned short> flag(W), leaf(W), ins(W), ers(W), ends(W), loud(cap), need(1);
    std::vector<int> arrivals(cap), work(2 * SFM_MAX_DEPTH * 8);
    std::vector<unsigned long long> todo(4);
    std::vector<SfmMove> mv(W);
    long serial_total = 0, loud_steps = 0, whole_steps = 0, split_steps = 0;
    std::vector<double> prev = p, cur = p;  // (the first step finds the crowd at the reset's positions: Tagent::setPosition leaves the tree alone, pedscene.h:34-36)
    for (int step = 0; step < steps; step++) {
        prev = p;  // where the step finds the crowd
        if (mode == 9) {
            for (int i = 0; i < n; i++) {
                p[3 * i] = recorded[(size_t)(step + 2) * n * 2 + 2 * i];
                p[3 * i + 1] = recorded[(size_t)(step + 2) * n * 2 + 2 * i + 1];
            }
        } else {  // the crowd walks: towards its goal with noise, a new goal on arrival
            for (int i = 0; i < n; i++) {
                const double dx = goal[2 * i] - p[3 * i], dy = goal[2 * i + 1] - p[3 * i + 1], d = std::sqrt(dx * dx + dy * dy);
                if (d < 0.3) new_goal(i);
                const double v = 0.25 * U(rng) + 0.05;
                p[3 * i] += v * dx / (d + 1e-9) + 0.05 * N(rng);
                p[3 * i + 1] += v * dy / (d + 1e-9) + 0.05 * N(rng);
                if (mode == 3 && U(rng) < 0.05) {
                    if (U(rng) < 0.5) p[3 * i] = snap(p[3 * i]);
                    else p[3 * i + 1] = 10.0 + snap(p[3 * i + 1] - 10.0);
                }
            }
        }
        cur = prev;
        // literal: Tagent::move one agent at a time -- its position, then scene->moveAgent(this) (ped_agent.cpp:519-571), while the
        // agents behind it still stand where the step found them
        {
            const int before = lit.n_nodes;
            for (int a = 0; a < n && !err; a++) {
                cur[3 * a] = p[3 * a];
                cur[3 * a + 1] = p[3 * a + 1];
                sfm_move_agent(lit.nodes.data(), &lit.n_nodes, cap, lit.hash.data(), cur.data(), a, &err);
            }
            split_steps += lit.n_nodes != before;
        }
        // staged
        {
            SfmSurgery c{};
            c.nodes = stg.nodes.data(); c.n_nodes = &stg.n_nodes; c.cap_nodes = cap; c.treehash = stg.hash.data(); c.p = p.data(); c.n = n;
            c.flag = flag.data(); c.leaf = leaf.data(); c.need_serial = need.data(); c.ins_leaf = ins.data(); c.ers_leaf = ers.data();
            c.ends = ends.data(); c.loud = loud.data(); c.arrivals = arrivals.data(); c.todo = todo.data(); c.work = work.data();
            c.p_old = prev.data(); c.old_stride = 3;
            std::fill(flag.begin(), flag.end(), 0);
            for (int i = 0; i < n; i++) sfm_surgery_descent(c, i);
            const bool in_lds = stg.n_nodes + 32 <= SFM_LDS_NODES;
            if (in_lds) {
                std::fill(arrivals.begin(), arrivals.end(), 0);
                std::fill(loud.begin(), loud.end(), 0);
                need[0] = 0;
                for (int i = 0; i < W; i++) mv[i] = sfm_surgery_classify(c, i);
                for (int i = 0; i < W; i++) sfm_surgery_count(c, i, mv[i]);
                for (int q = 0; q < stg.n_nodes; q++) sfm_surgery_census(c, q);
                for (int i = 0; i < W; i++) sfm_surgery_verdict(c, i, mv[i]);
                std::vector<int> order(W);
                for (int i = 0; i < W; i++) order[i] = i;
                std::shuffle(order.begin(), order.end(), rng);
                for (int i : order)
                    if (mv[i].do_erase) sfm_set_erase(stg.nodes[mv[i].old], i);
                std::shuffle(order.begin(), order.end(), rng);
                for (int i : order)
                    if (mv[i].do_insert) sfm_set_insert(stg.nodes[mv[i].T], i, &err);
                for (int i = 0; i < W; i++) sfm_surgery_settle(c, i, mv[i]);
                int nl = 0;
                for (int q = 0; q < stg.n_nodes; q++) nl += loud[q];
                loud_steps += nl > 0;
                whole_steps += need[0];
            } else {
                need[0] = 1;
                for (int i = 0; i < W; i++) ends[i] = (unsigned short)(i < n && flag[i] ? 4 : 0);
            }
            std::fill(todo.begin(), todo.end(), 0ull);
            for (int i = 0; i < n; i++)
                if (flag[i] == 1) {
                    todo[i >> 6] |= 1ull << (i & 63);
                    serial_total++;
                }
            sfm_surgery_replay(c, &err);
        }
        if (err) {
            printf("step %d: error code %d\n", step, err);
            return 1;
        }
        if (!same_tree(lit, stg, n, step)) return 1;
        if (mode == 9) {  // for a comparison with a recorded run: node count, member entries, who is in the tree
            int members = 0;
            unsigned long long in[4] = {0, 0, 0, 0};
            for (int k = 0; k < lit.n_nodes; k++)
                if (lit.nodes[k].isleaf)
                    for (int e = 0; e < lit.nodes[k].n_agents; e++) {
                        members++;
                        in[lit.nodes[k].agents[e] >> 6] |= 1ull << (lit.nodes[k].agents[e] & 63);
                    }
            printf("step %d: %d nodes, %d member entries, in the tree: %016llx %016llx %016llx %016llx\n", step, lit.n_nodes, members, in[0], in[1], in[2], in[3]);
        }
    }
    printf("OK %d steps, %d agents, mode %d: %d nodes at the end, splits on %ld steps, leaves about to split on %ld, whole steps replayed %ld, "
           "%.2f agents replayed per step\n", steps, n, mode, lit.n_nodes, split_steps, loud_steps, whole_steps, (double)serial_total / steps);
    return 0;
}
