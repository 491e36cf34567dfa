// exp_hooks.h -- measurement hooks of k_view.  In the product build (none of the switches defined) every macro here expands to
// nothing, so the kernel reads as what ships.  A library built with any of them names itself accordingly (imgenv_backend()) and
// bench.py refuses it.
//   -DIMGENV_EXP_STOP_AFTER=n   k_view returns after phase n (1 crop, 2 first hits, 3 / 4 steps A / B of the resolve): the phases
//                               before an early return are untouched, so counter differences between builds are phase costs
//                               (tools/run_exp_pmc.sh)
//   -DIMGENV_EXP_TINY_RESOLVE   hardly any room for the chunk descriptors / result slots of the resolve: every robot runs its
//                               "no room" fallbacks; the parity suites run on such a build (tools/check_k_view.sh)
//   -DIMGENV_EXP_RESOLVE_STATS  device-side counts of the cells a top beam leaves alone and of the ray lists behind them
//                               (tools/resolve_stats.py)
//   -DIMGENV_EXP_EVERY_CELL     k_view's crop and final pass over every cell in the steps too (what they did until round 4)
//   -DIMGENV_EXP_SKEW=n         every second k_view workgroup starts n x 64 clocks late: a timing experiment ("do the wavefronts suffer
//                               from all being in the same phase at the same time?" -- no: the kernel gets longer by the delay)
#pragma once

#ifdef IMGENV_EXP_STOP_AFTER
#define EXP_STOP_AFTER(n)                      \
    do {                                       \
        if (IMGENV_EXP_STOP_AFTER == (n)) return; \
    } while (0)
#else
#define EXP_STOP_AFTER(n) (void)0
#endif

#ifdef IMGENV_EXP_SKEW
#define EXP_SKEW()                                                                          \
    do {                                                                                    \
        if (blockIdx.x & 1)                                                                 \
            for (int q_ = 0; q_ < IMGENV_EXP_SKEW; q_++) __builtin_amdgcn_s_sleep(1);       \
    } while (0)
#else
#define EXP_SKEW() (void)0
#endif

#ifdef IMGENV_EXP_EVERY_CELL  // k_view's final pass over every cell in the steps too (what it did until round 4): A/B builds
#define EXP_EVERY_CELL true
#else
#define EXP_EVERY_CELL false
#endif

#ifdef IMGENV_EXP_TINY_RESOLVE
#define EXP_RESOLVE_CAP(product, tiny) (tiny)
#else
#define EXP_RESOLVE_CAP(product, tiny) (product)
#endif

#ifdef IMGENV_EXP_RESOLVE_STATS
// how many cells are left alone, how long the lists behind them are, and where a sequential walk of a list would end: entry
// index of the first deciding beam (bins 1, 2, 3-4, 5-8, 9-16, 17+), or never
#define EXP_RESOLVE_STATS_CELLS(w, k, hit, skip_list, n_skip, tid, NT)                                                           \
    do {                                                                                                                         \
        if ((tid) == 0) {                                                                                                        \
            atomicAdd(&(w).dbg[16], 1ull);                                                                                       \
            atomicAdd(&(w).dbg[17], (unsigned long long)(n_skip));                                                               \
        }                                                                                                                        \
        for (int t_ = (tid); t_ < (n_skip); t_ += (NT)) {                                                                        \
            const uint32_t e_ = (skip_list)[t_];                                                                                 \
            for (uint32_t bits_ = e_ & 15u; bits_ != 0u; bits_ &= bits_ - 1u) {                                                  \
                const uint32_t c_ = (e_ >> 4) + (uint32_t)__builtin_ctz(bits_);                                                  \
                const uint32_t pk_ = (k).inv_pack[c_], e0_ = pk_ & 0xFFFFFu, cnt_ = pk_ >> 20;                                   \
                atomicAdd(&(w).dbg[18], 1ull);                                                                                   \
                atomicAdd(&(w).dbg[19], (unsigned long long)cnt_);                                                               \
                atomicMax(&(w).dbg[20], (unsigned long long)cnt_);                                                               \
                atomicAdd(&(w).dbg[21 + min(cnt_ >> 3, 8u)], 1ull);                                                              \
                uint32_t at_ = 0, verdict_ = 2;                                                                                  \
                for (uint32_t q_ = 1; q_ < cnt_ && at_ == 0; q_++) {                                                             \
                    const uint32_t ent_ = (k).inv_ent[e0_ + q_], kk_ = ent_ & 0xFFFFu, hp_ = (hit)[ent_ >> 16], hk_ = hp_ >> 16; \
                    if (kk_ < hk_) { at_ = q_; verdict_ = 3; }                                                                   \
                    else if (kk_ == hk_) { at_ = q_; verdict_ = 0; }                                                             \
                    else if (kk_ > (hp_ & 0xFFFFu)) at_ = q_;                                                                    \
                }                                                                                                                \
                const int bin_ = at_ == 0 ? 6 : (at_ <= 2 ? (int)at_ - 1 : (at_ <= 4 ? 2 : (at_ <= 8 ? 3 : (at_ <= 16 ? 4 : 5)))); \
                atomicAdd(&(w).dbg[8 + bin_], 1ull);                                                                             \
                if (verdict_ != 2) atomicAdd(&(w).dbg[15], 1ull);                                                                \
                if (at_ == 0) atomicAdd(&(w).dbg[30], (unsigned long long)cnt_); /* entries walked for nothing */                 \
                else atomicAdd(&(w).dbg[31], (unsigned long long)at_);                                                           \
            }                                                                                                                    \
        }                                                                                                                        \
    } while (0)
#define EXP_RESOLVE_STATS_ROOM(w, nd, nr, tid)                      \
    do {                                                            \
        if ((tid) == 0) {                                           \
            atomicAdd(&(w).dbg[6], (unsigned long long)(nd));       \
            atomicAdd(&(w).dbg[7], (unsigned long long)(nr));       \
        }                                                           \
    } while (0)
#else
#define EXP_RESOLVE_STATS_CELLS(w, k, hit, skip_list, n_skip, tid, NT) (void)0
#define EXP_RESOLVE_STATS_ROOM(w, nd, nr, tid) (void)0
#endif
