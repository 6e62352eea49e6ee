// tu_big.hip -- the path for large clusters (n_x > 60: BASELINE config 5) and the fp32 arm of its tolerance study:
// the fused workgroup-per-item Riccati sweep (riccati_big.hpp), for double and float.  (Its forward pass / line search is
// tu_bigfwd.hip: this unit is compiled with machine LICM off -- __graft_entry__.UNIT_FLAGS -- which only the sweep needs.)
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "launch.hpp"
#include "riccati_big.hpp"

namespace dpilqr {

int64_t riccati_big_scratch_elems(int n, int m) { return BigScratch(n, m).total; }

template <typename R>
static int32_t launch_riccati_big_t(const dpilqr_batch_desc& D, const R* X, const R* U, const double* mu, R* K, R* d,
                                    int32_t* singular, const int32_t* items, const int32_t* n_items, int grid_items,
                                    int gains_by_item, void* scratch, hipStream_t st) {
    if (grid_items <= 0) return DPILQR_OK;
    const size_t lds = sizeof(R) * (size_t)BigLds(D.k, D.n_s, D.n_c).total;
    // Few items (config 5 is one problem): a team of workgroups per item, the helpers taking their share of S5 + S6's tile pairs
    // (riccati_big.hpp, BigTeam).  Parts: round 5's rule was as many as give every wavefront at most one pair (nine at cfg5's size) --
    // more only added hand-over cost.  With round 6's hand-overs (one acquire per workgroup, no L2 write-back inside an XCD) and
    // more of the step on the team (S1's block pairs, the plugin stage, the substitution's tiles, S4) twice as many pay: one item
    // 18.0 ms with 9 parts, 17.2 with 13, 16.2 - 16.5 from 16 to 30 (profiles/r06_big_parts.txt), so 2 x that + 2, at most 28 (a
    // team stays inside one XCD's 32 CUs with a margin), as long as all teams fit the chip at one workgroup per CU.
    // DPILQR_BIG_TEAM=0 switches it off (A/B), =N caps the parts.
    const int n = D.k * D.n_s, m = D.k * D.n_c;
    const BigScratch S(n, m);
    int nparts = 1;
#ifndef DPILQR_BIG_S5_SEPARATE
    {
        const int cus = device_cus();      // of the CURRENT device (remembered per device id)
        const int tn = (n + 1 + 15) / 16, npair = tn * (tn + 1) / 2, waves = kBigThreads / 64;
        const int slots8 = ((grid_items + 7) / 8) * 8;
        nparts = 2 * ((npair + waves - 1) / waves) + 2;
        if (nparts > 28) nparts = 28;
        if (nparts > cus / slots8) nparts = cus / slots8;
        // (a debug route: null in a process without DPILQR_DEBUG_ROUTES=1, where nothing is read from the environment here; with
        // the gate open the tests switch it from launch to launch)
        const char* e = route_env("DPILQR_BIG_TEAM");
        if (e && *e) { const int v = std::atoi(e); nparts = v <= 0 ? 1 : (v < nparts ? v : nparts); }
        const int exact = route_int("DPILQR_BIG_TEAM_PARTS", 0);      // experiments: exactly this many parts (more than one pair-wavefront each)
        if (exact > 1) nparts = exact < cus / slots8 ? exact : cus / slots8;
        if (nparts < 2) nparts = 1;
    }
#endif
    // tests (gated like every route switch): DPILQR_BIG_TEAM_LATE=1 helpers report late, =2 helpers join and then stall (fault
    // injection); DPILQR_BIG_SPIN_LOG2=s bounds every wait at 2^s polls instead of 2^22 -- packed into one kernel argument
    const int team_dbg = (route_int("DPILQR_BIG_TEAM_LATE", 0) & 3) | ((route_int("DPILQR_BIG_SPIN_LOG2", 0) & 31) << 8) |
                         (route_flag("DPILQR_BIG_TEAM_AGENT") ? 4 : 0);      // (A/B: agent-scope hand-overs even where the team shares an XCD; riccati_big.hpp, big_arrive)
    const int grid = nparts > 1 ? ((grid_items + 7) / 8) * 8 * nparts : grid_items;
    if (nparts > 1)
        hipLaunchKernelGGL((k_big_team_reset<R>), dim3((grid_items + 255) / 256), dim3(256), 0, st, static_cast<R*>(scratch),
                           (int64_t)S.total, (int64_t)S.oSync, grid_items);
    DISPATCH_FAMILY(D.n_s, {
        int32_t rc = allow_lds(k_riccati_big<R, NS, NC>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((k_riccati_big<R, NS, NC>), dim3(grid), dim3(kBigThreads), lds, st, D, X, U, mu, K, d,
                           singular, items, n_items, gains_by_item, static_cast<R*>(scratch), grid_items, nparts, team_dbg);
    })
    HIP_TRY(hipGetLastError());
    return DPILQR_OK;
}

int32_t launch_riccati_big_f64(const dpilqr_batch_desc& D, const double* X, const double* U, const double* mu, double* K,
                               double* d, int32_t* singular, const int32_t* items, const int32_t* n_items,
                               int grid_items, int gains_by_item, void* scratch, hipStream_t st) {
    return launch_riccati_big_t<double>(D, X, U, mu, K, d, singular, items, n_items, grid_items, gains_by_item, scratch, st);
}
int32_t launch_riccati_big_f32(const dpilqr_batch_desc& D, const float* X, const float* U, const double* mu, float* K,
                               float* d, int32_t* singular, const int32_t* items, const int32_t* n_items,
                               int grid_items, int gains_by_item, void* scratch, hipStream_t st) {
    return launch_riccati_big_t<float>(D, X, U, mu, K, d, singular, items, n_items, grid_items, gains_by_item, scratch, st);
}

}  // namespace dpilqr
