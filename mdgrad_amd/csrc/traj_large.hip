// Fused trajectories for systems too large for one workgroup (BASELINE config #4: 4 096-atom
// LJ liquid; 1 024 < N <= 32 768).  Same semantics as traj_small.hip (NoseHooverChain + NH-Verlet forward,
// torchmd/sovlers.py:110-127 with RHS torchmd/md.py:210-240; adjoint sovlers.py:211-293 with the
// backward branch :129-164), but the state lives in HBM/L2 and every step is a few launches
// enqueued from a C++ host loop (no Python, no host sync):
//
//   large_prep        between two force launches: the element-wise update of the integrator / adjoint that ends in the
//                     positions of the next force evaluation, the cross-workgroup scalars of the previous force launch
//                     (summed in a fixed order: "launch-boundary reduce" -- deterministic, no atomics, no grid barrier),
//                     the thermostat chain, the Verlet-reuse decision and, when a search is due, the cell binning
//   large_search_rows the force launch that SEARCHES (cell-binned, candidates at cutoff + skin stored per atom)
//   large_fwd_tiled / large_adj_tiled     the force (+ Hessian.w + parameter vjp) launches over the STORED candidates,
//                     exact cutoff re-applied (topology_update_freq = 1 semantics with a search every ~7 steps): a
//                     workgroup owns one column of bins of the list's build and stages the 3 x 3 columns' state rows in
//                     LDS (column tiles, round 5; large_*_listed: the same from L2 by atom index, for unbinned boxes)
//   large_force_step / large_adj_force    search + evaluate at every call (boxes that cannot be binned, block = -1)
// Forward step: three launches; adjoint interval: four, NHC or NVE.  The launches of a trajectory are issued for groups
// of replicas on concurrent side streams (lg_streams, round 5): one group's latency-bound prep launches and tails overlap
// another group's force sweep.
#include "common.hpp"
#pragma clang diagnostic ignored "-Wunused-result"

namespace {

constexpr int LG_WAVES = 16;                 // waves (= atoms in flight) per workgroup: all-atom scan (tile staging amortised)
constexpr int LG_WAVES_CELL = 4;             // ... cell-binned scan: small workgroups (no tile; short barriers; 5 resident per CU)
constexpr int LG_BLOCK = LG_WAVES * 64;
constexpr int LG_TILE = 2048;               // positions staged in LDS per pass
constexpr int LG_CAP = 256;                  // per-wave neighbour buffer (entries)
constexpr int LG_LIST = 128;                 // stored neighbour indices per atom and list build
constexpr float LG_SKIN = 0.12f;             // skin of the stored lists, as a fraction of the largest cutoff
constexpr float LG_REUSE = 0.45f;            // the forward pass searches again once an atom has moved this fraction of the skin
                                             // (< 1/2: room for the adjoint's midpoint states between two frames)
constexpr long long LG_LIST_MAX_WORDS = 1ll << 33;   // at most 32 GiB of stored lists (of 288); beyond that every evaluation searches
constexpr int LG_TILE_THREADS = 512;          // workgroup of the column-tile kernels (32 rows of 16 lanes)
constexpr int LG_TILE_MAX = 4096;            // staged atoms of a column tile at most
constexpr int LG_MAX_COLS = 1366;            // bin columns (nbx nby) at most: LG_MAX_CELLS / 3
constexpr int LG_KMAX = MDG_MAX_TERMS * MDG_MAX_THETA;
constexpr int LG_NV = LG_KMAX + 2;           // theta partials, sum p^2/m, sum lambda_v.v

struct LargeArgs {
    MdgTrajParams prm;
    MdgCell cell;
    MdgTerms terms;
    const float* theta; const float* mass; const float* t;
    float *q, *v, *vh, *f;                   // [R][N][3] running state
    float *pv, *ph, *pvh;                    // [R][16]
    float *partA, *partB;                    // [R][nb] kinetic-energy partials (ping / pong)
    float *v_t, *q_t, *pv_t;                 // [R][T][N][3] frames
    int32_t* flags;                          // [0] neighbour-buffer overflow, [1] non-finite, [2] table-gradient range, [3] pair below the table
    const float *g_v, *g_q, *g_pv;           // adjoint: incoming frame gradients (nullable)
    float *lv, *lq, *lvh, *lqh, *dq, *qm, *vm;   // [R][N][3]
    float *wl;                               // [R][N][3] adjoint direction of the coming listed evaluation: lam_v / m (NVE: lam_v)
    float *lp, *lph, *pvm;                   // [R][16]
    float *partN;                            // [R][nb][LG_NV]
    float *gth;                              // [R][K_total]
    unsigned long long* g64;                 // MDG_PAIR_TABLE: fixed-point table gradient, int64 words [R][K_total] (fx64, common.hpp)
    float glim;                              //   a single contribution at or beyond this raises flags[2] (fx64_limit)
    float *adj_v0, *adj_q0, *adj_pv0, *adj_theta;
    int nbF, nbE;                            // workgroups of the per-atom / per-element kernels
    int step;                                // forward: step index k; adjoint: frame index i
    // cell-binned neighbour scan (orthorhombic cell, >= 3 bins of >= cutoff per dimension): positions sorted by
    // (bin, atom index) as (x, y, z, index) and the first slot of every bin, rebuilt before each force evaluation
    float4* spos;                            // [R][N]
    int32_t* bstart;                         // [R][LG_MAX_CELLS + 1]
    int32_t* binslot;                        // [R][N] atom indices in provisional bin order (scratch of the binning)
    int nb[3], ncell;                        // ncell == 0: scan all atoms through LDS tiles
    // neighbour lists of the forward pass, kept for the adjoint (nullptr: not kept).  A forward search uses the cutoff
    // (1 + LG_SKIN) rc and stores the candidate indices; later forward steps and the adjoint's two evaluations per
    // interval gather those candidates and re-apply the exact cutoff test -- the same pair set as a fresh search as
    // long as no atom has moved more than skin/2 since the build (forward frames: by construction, see nl_build; the
    // adjoint's midpoint states: checked on the device by large_prep<3>; flags[5] then asks the caller for an adjoint
    // with fresh searches).
    uint16_t* nl_idx;                        // [R][T][N][LG_LIST]  (16-bit: N <= 16 384; the rows are the HBM traffic of the listed launches)
    int32_t* nl_cnt;                         // [R][T][N]
    int32_t* nl_bad;                         // [R][T]  an atom of this frame had more than LG_LIST candidates
    // Verlet reuse: a list serves every later frame until an atom has moved LG_REUSE x skin from its build positions
    // (checked on the device by large_prep<1>, which then asks the next force launch for a search).  Forces over a
    // stored list re-apply the exact cutoff test: the pair set of every evaluation is the one a fresh search gives
    // (topology_update_freq = 1, sovlers.py:114) -- the search itself runs every few steps only.
    int32_t* nl_build;                       // [R][T]  the frame whose stored rows serve frame f
    int32_t* nl_state;                       // [R][2]  {the coming force launch searches, frame of the current list}
    int nbL;                                 // workgroups (partial rows) of the listed force launches
    float skin;                              // absolute skin (0: lists not kept)
    // COLUMN TILES (round 5; binned boxes with kept lists).  The listed launches gathered every candidate's position and
    // adjoint direction from L2 by atom index -- ~48 fully divergent 12-byte gathers per wave, bound by the CU's address /
    // L1 path at 66 us per adjoint evaluation of 64 x 4 096 atoms.  Now a workgroup owns one (bx, by) column of bins of
    // the list's BUILD (a contiguous range of the build's sorted order), stages the 3 x 3 columns around it -- nine
    // contiguous ranges of a copy of the state kept in that sorted order -- in LDS once, and a stored row holds 16-bit
    // slots INTO THE STAGED TILE: every candidate is an LDS read (ds_read_b96).  The order is fixed for the life of a
    // build (atoms stay within skin / 2 of where they were binned); staging gathers the state rows through the build's
    // permutation -- ~9 gathered rows per atom of the tile instead of ~78 per atom (a first version kept copies of the state
    // in build order, written by the prep launches: their scattered 16-byte stores cost more than these gathers save).
    int32_t* nl_perm;                        // [R][T][N]  atom at sorted slot s of the build of frame b
    int32_t* nl_bst;                         // [R][T][LG_MAX_COLS + 1]  first sorted slot of bin column c of that build
    int rep0;                                // first replica of this launch (replica groups on concurrent streams, see lg_streams)
    int tile_cap;                            // staged atoms of a tile at most (0: no tiles)
    int ncol;                                // bin columns nb[0] nb[1]
    // STALE LISTS (topology_update_freq > 1, round 6; mdg_traj_*_large_stale): the pair set and image flags of the last rebuild,
    // persistent across launches like the reference's nbr_list / offsets attributes.  A row entry is
    // j | image code << 15 | term bits << 20 in ascending j; the host loop knows every call's running count and says per
    // launch whether it rebuilds (search at the exact cutoffs of the terms, generate_nbr_list's tests) or evaluates the stored
    // rows with their frozen flags and no cutoff re-test (interface.py:298-300).
    uint32_t* st_row;                        // [R][N][LG_CAP]
    int32_t* st_cnt;                         // [R][N]
    int st_rebuild;
    int st_bin;                              // prep launches: the coming force launch rebuilds -- bin its positions (else: skip)
};

// ---------------------------------------------------------------------------------------------
// Binning: ONE workgroup of 1 024 threads per replica (large_prep below) counts its atoms into <= 4 096 bins with
// LDS integer atomics (slot inside the bin = the atomic's return value), scans the counts in LDS and scatters
// (x, y, z, index) to start[bin] + slot -- one launch, no global atomics, no counter buffers.
// The atomic's order is only provisional: the final slot inside a bin is the atom's rank by index among its bin
// mates, so the sorted array is the same on every run.  (The wave-per-atom kernels additionally sort every atom's
// compacted neighbour buffer by index, which restores the ascending-j order of the all-atom scan: same sums, same bits.)
constexpr int LG_MAX_CELLS = 4096;

__device__ __forceinline__ int bin_coord_l(float x, float inv, int nb) {
    float fr = x * inv;
    fr -= floorf(fr);
    const int b = (int)(fr * (float)nb);
    return b >= nb ? nb - 1 : (b < 0 ? 0 : b);
}

__device__ __forceinline__ float bath_rhs_l(const MdgTrajParams& p, const float* Q, const float* pv, float ke, int k) {
    const int C = p.n_chains;
    if (k == 0) return 2.f * (ke - p.T * p.n_dof * 0.5f) - pv[0] * pv[1] / Q[1];
    if (k == C - 1) return pv[C - 2] * pv[C - 2] / Q[C - 2] - p.T;
    return (pv[k - 1] * pv[k - 1] / Q[k - 1] - p.T) - pv[k + 1] * pv[k] / Q[k + 1];
}

__device__ __forceinline__ float bath_vjp_l(const MdgTrajParams& p, const float* Q, const float* pv,
                                            const float* lp, float slv, int k) {
    const int C = p.n_chains;
    if (k == 0) return -slv / Q[0] - lp[0] * pv[1] / Q[1] + 2.f * pv[0] * lp[1] / Q[0];
    if (k == C - 1) return -lp[C - 2] * pv[C - 2] / Q[C - 1];
    return -lp[k - 1] * pv[k - 1] / Q[k] - lp[k] * pv[k + 1] / Q[k + 1] + 2.f * pv[k] * lp[k + 1] / Q[k];
}

// sum of column `col` of a [n][stride] partial array by one workgroup (fixed order)
__device__ __forceinline__ float reduce_partials(const float* __restrict__ part, int n, int stride, int col,
                                                 float* red) {
    float s = 0.f;
    for (int b = threadIdx.x; b < n; b += blockDim.x) s += part[(size_t)b * stride + col];
    return block_sum(s, red);
}

// column sums of a [n][LG_NV] partial array by one workgroup: one pass over the rows, fixed order (red: 16 * LG_NV floats)
__device__ __forceinline__ void sum_partial_rows(const float* __restrict__ part, int n, float (&tot)[LG_NV], float* red) {
#pragma unroll
    for (int p = 0; p < LG_NV; ++p) tot[p] = 0.f;
    for (int b = threadIdx.x; b < n; b += blockDim.x) {
#pragma unroll
        for (int p = 0; p < LG_NV; ++p) tot[p] += part[(size_t)b * LG_NV + p];
    }
    block_sum_n<LG_NV>(tot, red);
}

struct __attribute__((packed, aligned(4))) Row3 { float x, y, z; };      // one [3] row of a [N][3] array: global_load_dwordx3
__device__ __forceinline__ Row3 row3(const float* __restrict__ a, int j) { return *reinterpret_cast<const Row3*>(a + 3 * (size_t)j); }
// the same at an element offset / as a store: the element-wise updates of large_prep request ALL rows of an atom in one
// round trip (twelve-byte loads back to back, no store between them) and write its results afterwards -- taken component
// by component through the state arrays, every store could alias the next component's loads and the three components
// became three dependent round trips (18-26 us per launch at 2-3 TB/s)
__device__ __forceinline__ Row3 ld3(const float* a, size_t e) { return *reinterpret_cast<const Row3*>(a + e); }
__device__ __forceinline__ void st3(float* a, size_t e, float x, float y, float z) { *reinterpret_cast<Row3*>(a + e) = Row3{x, y, z}; }

// ------------------------------------------------------------------------------------ per-replica preparation
// Everything between two force launches runs in ONE launch of one 1 024-thread workgroup per replica: the
// elementwise update of the integrator / adjoint (which ends in the positions of the next force evaluation), the
// cross-workgroup scalars of the previous force launch (kinetic energy, sum lam.v, parameter partials: summed here in
// a fixed order), the thermostat chain, and the binning of those positions for the cell-binned scan.
//   PHASE 0  forward, before the initial force: bin q0
//   PHASE 1  forward step k: first RHS with the cached force -> half kick, drift, bath half step
//            (sovlers.py:111-118 / :25-33); bin the new positions
//   PHASE 2  adjoint, before the first evaluation of interval i = A.step: finish interval i + 1 (full adjoint update
//            + dL/dy_i, sovlers.py:156-160, :286 / :100) unless i is the last frame; bin frame i
//   PHASE 3  adjoint, before the midpoint evaluation: midpoint state and half-step adjoint (sovlers.py:132-145 /
//            :42-82); bin the midpoint positions
//   PHASE 4  adjoint, after the last interval: finish interval 1 (A.step = 0), no binning
constexpr int LG_PREP = 1024;
constexpr int LG_PREP_SMALL = 256;           // ... of the phases that bin nothing (the adjoint over stored lists): more, smaller workgroups
constexpr int LG_PREP_ATOMS = 16;            // atoms per thread of the binning workgroup: N <= 16 384 ...
constexpr int LG_PREP_ATOMS_MAX = 32;        // ... and <= 32 768 (round 5: the positions of 32 atoms per thread in registers)

// NA = atoms per thread (compile-time: their positions stay in registers between the update and the binning)
template <int PHASE, int NA>
__global__ __launch_bounds__(LG_PREP) void large_prep(const LargeArgs A) {
    __shared__ int32_t start[LG_MAX_CELLS + 1];
    __shared__ int32_t tsum[LG_PREP];
    __shared__ float red[32];
    __shared__ float redN[16 * LG_NV];
    __shared__ float Qs[MDG_MAX_CHAINS], pvs[MDG_MAX_CHAINS], lps[MDG_MAX_CHAINS];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, C = A.prm.n_chains, rep = blockIdx.y + A.rep0, nc = A.ncell;
    // gridDim.x workgroups share a replica's atoms when nothing has to be binned (the adjoint over stored lists); the
    // scalar work (partial sums, thermostat chain) is repeated by each, written by the first
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    const bool first = blockIdx.x == 0;
    const bool nhc = A.prm.ensemble == 0;
    const size_t so = (size_t)rep * N * 3;
    if (threadIdx.x < MDG_MAX_CHAINS) {
        float qv = 0.f;
#pragma unroll
        for (int c = 0; c < MDG_MAX_CHAINS; ++c) if (threadIdx.x == c) qv = A.prm.Q[c];
        Qs[threadIdx.x] = qv;
    }
    float px[NA], py[NA], pz[NA];       // positions of the next force evaluation

    if constexpr (PHASE == 0) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int a = tid + u * stride;
            if (a < N) { px[u] = A.q[so + 3 * a]; py[u] = A.q[so + 3 * a + 1]; pz[u] = A.q[so + 3 * a + 2]; }
        }
    }
    if constexpr (PHASE == 1) {
        const int k = A.step;
        const float dt = A.t[k + 1] - A.t[k];
        float* pv = A.pv + rep * MDG_MAX_CHAINS;
        const bool lists = A.nl_idx != nullptr;
        // (the previous force launch was a search with nbF workgroups or a listed one with nbL)
        // (... or, in a binned box, by large_search_rows, which shares the listed kernels' launch shape)
        const int rows = (lists && (nc > 0 || !A.nl_state[2 * rep])) ? A.nbL : A.nbF;
        const int bfr = lists ? A.nl_state[2 * rep + 1] : 0;
        const float* qb = A.q_t + ((size_t)rep * T + bfr) * N * 3;      // positions the current list was built at
        float far2 = 0.f;
        if (nhc) {
            const float ke = 0.5f * reduce_partials(A.partA + (size_t)rep * A.nbF, rows, 1, 0, red);
            if (threadIdx.x < C) pvs[threadIdx.x] = pv[threadIdx.x];
            __syncthreads();
            if (first && threadIdx.x < C) {
                const float h = 0.5f * bath_rhs_l(A.prm, Qs, pvs, ke, threadIdx.x) * dt;
                A.ph[rep * MDG_MAX_CHAINS + threadIdx.x] = h;
                A.pvh[rep * MDG_MAX_CHAINS + threadIdx.x] = pvs[threadIdx.x] + h;
            }
        }
        const float pv0 = nhc ? pv[0] : 0.f;
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int a = tid + u * stride;
            if (a < N) {
                const float m = A.mass[a];
                const size_t e3 = so + 3 * (size_t)a;
                const Row3 vr = ld3(A.v, e3), fr = ld3(A.f, e3), qr = ld3(A.q, e3);
                const Row3 qbr = lists ? ld3(qb, 3 * (size_t)a) : Row3{0.f, 0.f, 0.f};
                const float vv[3] = {vr.x, vr.y, vr.z}, ff[3] = {fr.x, fr.y, fr.z}, qq[3] = {qr.x, qr.y, qr.z};
                const float qbv[3] = {qbr.x, qbr.y, qbr.z};
                float qn[3], hv[3], mv2 = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float ve = vv[c], p = ve * m;
                    const float acc = nhc ? (ff[c] - pv0 * p / A.prm.Q[0]) / m : ff[c];     // (NVE: md.py:145-148)
                    const float h = 0.5f * acc * dt;
                    hv[c] = h;
                    qn[c] = qq[c] + (ve + h) * dt;
                    const float ph2 = (ve + h) * m;
                    part += ph2 * ph2 / m;
                    if (lists) { const float mv = qn[c] - qbv[c]; mv2 = fmaf(mv, mv, mv2); }
                }
                st3(A.vh, e3, hv[0], hv[1], hv[2]);
                st3(A.q, e3, qn[0], qn[1], qn[2]);
                far2 = fmaxf(far2, mv2);
                px[u] = qn[0]; py[u] = qn[1]; pz[u] = qn[2];
            }
        }
        part = block_sum(part, red);
        if (threadIdx.x == 0) A.partB[rep] = part;                     // (one workgroup per replica in this phase; nbE = 1: KE(v + vh) for the force launch)
        if (lists) {
            // the current list still holds every pair inside the cutoff unless an atom has left its LG_REUSE x skin ball
            // (or the build was incomplete): then the coming force launch searches, and the positions are binned for it
            // (the last frame has no successor whose build could serve its midpoint: it keeps a wider margin itself)
            const float lim = (k + 2 >= T ? 0.5f * LG_REUSE : LG_REUSE) * A.skin;
            const int search = __syncthreads_or(!(far2 <= lim * lim)) || A.nl_bad[(size_t)rep * T + bfr] != 0;
            if (threadIdx.x == 0) A.nl_state[2 * rep] = search;
            if (!search) return;
        }
    }
    if constexpr (PHASE == 2 || PHASE == 4) {
        const int i_fr = A.step + 1;                                   // the interval being finished
        const bool fin = i_fr <= T - 1;
        const size_t go = ((size_t)rep * T + (fin ? i_fr - 1 : 0)) * N * 3;
        const float h = fin ? A.t[i_fr] - A.t[i_fr - 1] : 0.f;
        float pvm0 = 0.f, lpm0 = 0.f;
        if (fin && nhc) {
            const float* pvm = A.pvm + rep * MDG_MAX_CHAINS;
            const float* lph = A.lph + rep * MDG_MAX_CHAINS;
            float* lp = A.lp + rep * MDG_MAX_CHAINS;
            pvm0 = pvm[0]; lpm0 = lph[0];
            float tot[LG_NV];
            sum_partial_rows(A.partN + (size_t)rep * A.nbF * LG_NV, A.nbF, tot, redN);
            const float slv = tot[LG_KMAX + 1];
            const int KT = A.terms.n_theta_total;
            if (first && threadIdx.x == 0) {
#pragma unroll
                for (int m = 0; m < MDG_MAX_TERMS; ++m)
#pragma unroll
                    for (int p = 0; p < MDG_MAX_THETA; ++p)
                        if (m < A.terms.n_terms && p < A.terms.t[m].n_theta)
                            A.gth[(size_t)rep * KT + A.terms.t[m].theta_off + p] += tot[m * MDG_MAX_THETA + p] * h;   // :160
            }
            if (threadIdx.x < C) { pvs[threadIdx.x] = pvm[threadIdx.x]; lps[threadIdx.x] = lph[threadIdx.x]; }
            __syncthreads();
            if (first && threadIdx.x < C) {
                const float gp = bath_vjp_l(A.prm, Qs, pvs, lps, slv, threadIdx.x);
                float nlp = lp[threadIdx.x] + gp * h;                               // :158
                if (A.g_pv) nlp += A.g_pv[((size_t)rep * T + i_fr - 1) * C + threadIdx.x];
                lp[threadIdx.x] = nlp;
            }
        }
        // the coming listed evaluation (PHASE 2, first evaluation of interval A.step) gathers w = lam_v / m (NVE: lam_v) of
        // its candidates from ONE array (each gather stream of that kernel costs as much as its arithmetic)
        const bool want_w = PHASE == 2 && A.nl_idx != nullptr;
        const float* qf = A.q_t + ((size_t)rep * T + A.step) * N * 3;
        if (fin || want_w) {
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int a = tid + u * stride;
                if (a >= N) break;
                const float m = A.mass[a];
                const size_t e3 = so + 3 * (size_t)a, g3 = go + 3 * (size_t)a;
                float nlv[3];
                if (fin) {
                    const Row3 lvhr = ld3(A.lvh, e3), lqhr = ld3(A.lqh, e3), dqr = ld3(A.dq, e3);
                    const Row3 gvr = A.g_v ? ld3(A.g_v, g3) : Row3{0.f, 0.f, 0.f}, gqr = A.g_q ? ld3(A.g_q, g3) : Row3{0.f, 0.f, 0.f};
                    const float lvh_[3] = {lvhr.x, lvhr.y, lvhr.z}, lqh_[3] = {lqhr.x, lqhr.y, lqhr.z}, dq_[3] = {dqr.x, dqr.y, dqr.z};
                    const float gv_[3] = {gvr.x, gvr.y, gvr.z}, gq_[3] = {gqr.x, gqr.y, gqr.z};
                    float nlq[3];
                    if (nhc) {
                        const Row3 vmr = ld3(A.vm, e3), lvr = ld3(A.lv, e3), lqr = ld3(A.lq, e3);
                        const float vm_[3] = {vmr.x, vmr.y, vmr.z}, lv_[3] = {lvr.x, lvr.y, lvr.z}, lq_[3] = {lqr.x, lqr.y, lqr.z};
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float Gv = -(pvm0 / A.prm.Q[0]) * lvh_[c] + lqh_[c] + 2.f * m * vm_[c] * lpm0;
                            nlv[c] = lv_[c] + Gv * h;                                       // :156
                            nlq[c] = lq_[c] + dq_[c] * h;                                   // :157
                            if (A.g_v) nlv[c] += gv_[c];                                    // :286
                            if (A.g_q) nlq[c] += gq_[c];
                        }
                    } else {
                        // verlet_update backward branch, second half (sovlers.py:100) + dL/dy_{i-1} (:286)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            nlv[c] = lvh_[c];
                            nlq[c] = lqh_[c] + dq_[c] * h * 0.5f;
                            if (A.g_v) nlv[c] += gv_[c];
                            if (A.g_q) nlq[c] += gq_[c];
                        }
                    }
                    st3(A.lv, e3, nlv[0], nlv[1], nlv[2]);
                    st3(A.lq, e3, nlq[0], nlq[1], nlq[2]);
                } else {
                    const Row3 lvr = ld3(A.lv, e3);                                         // (the last frame: nothing to finish)
                    nlv[0] = lvr.x; nlv[1] = lvr.y; nlv[2] = lvr.z;
                }
                if (want_w) {
                    const float im = nhc ? 1.0f / m : 1.0f;
                    const float wx = nlv[0] * im, wy = nlv[1] * im, wz = nlv[2] * im;
                    st3(A.wl, e3, wx, wy, wz);
                }
            }
        }
        if constexpr (PHASE == 2) {
            if (A.nl_idx) return;                                      // (the stored candidates of frame i serve)
#pragma unroll
            for (int u = 0; u < NA; ++u) {
                const int a = tid + u * stride;
                if (a < N) { px[u] = qf[3 * a]; py[u] = qf[3 * a + 1]; pz[u] = qf[3 * a + 2]; }
            }
        }
    }
    if constexpr (PHASE == 3) {
        const int i_fr = A.step;
        const size_t fo = ((size_t)rep * T + i_fr) * N * 3;
        // The reference's backward half step drifts with the state's own velocity (sovlers.py:138 under the negated
        // dynamics): the NHC "midpoint" positions lie about one step AHEAD of frame i -- next to frame i + 1, whose
        // list may be a newer build.  Both builds are candidates; the distance to each is measured here and the
        // midpoint launch (large_adj_listed, second = 1) takes one that holds.
        const int slotA = A.nl_idx ? A.nl_build[(size_t)rep * T + i_fr] : i_fr;
        const int slotB = (A.nl_idx && i_fr + 1 < T) ? A.nl_build[(size_t)rep * T + i_fr + 1] : slotA;
        const float* qbA = A.q_t + ((size_t)rep * T + slotA) * N * 3;
        const float* qbB = A.q_t + ((size_t)rep * T + slotB) * N * 3;
        const float h = A.t[i_fr] - A.t[i_fr - 1];
        float tot[LG_NV];
        sum_partial_rows(A.partN + (size_t)rep * A.nbF * LG_NV, A.nbF, tot, redN);
        float pv0 = 0.f, lp0 = 0.f;
        if (nhc) {
            const float* pvf = A.pv_t + ((size_t)rep * T + i_fr) * C;
            float* lp = A.lp + rep * MDG_MAX_CHAINS;
            const float ke = 0.5f * tot[LG_KMAX], slv = tot[LG_KMAX + 1];
            if (threadIdx.x < C) { pvs[threadIdx.x] = pvf[threadIdx.x]; lps[threadIdx.x] = lp[threadIdx.x]; }
            __syncthreads();
            if (first && threadIdx.x < C) {
                const float pb = bath_rhs_l(A.prm, Qs, pvs, ke, threadIdx.x);
                const float gp = bath_vjp_l(A.prm, Qs, pvs, lps, slv, threadIdx.x);
                A.pvm[rep * MDG_MAX_CHAINS + threadIdx.x] = pvs[threadIdx.x] + 0.5f * (-pb) * h;      // :135
                A.lph[rep * MDG_MAX_CHAINS + threadIdx.x] = lps[threadIdx.x] + gp * 0.5f * h;         // :143
            }
            pv0 = pvs[0]; lp0 = lps[0];
        } else if (first && threadIdx.x == 0) {
            // the parameter term of an NVE interval comes from the first evaluation (sovlers.py:82,101)
            const int KT = A.terms.n_theta_total;
#pragma unroll
            for (int m = 0; m < MDG_MAX_TERMS; ++m)
#pragma unroll
                for (int p = 0; p < MDG_MAX_THETA; ++p)
                    if (m < A.terms.n_terms && p < A.terms.t[m].n_theta)
                        A.gth[(size_t)rep * KT + A.terms.t[m].theta_off + p] += (tot[m * MDG_MAX_THETA + p] * 0.5f * h) * 2.f;
        }
        float far2 = 0.f, far2B = 0.f;                              // largest |q_mid - q_build|^2 of this thread's atoms
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int a = tid + u * stride;
            if (a < N) {
                const float m = A.mass[a];
                const size_t e3 = so + 3 * (size_t)a, f3 = fo + 3 * (size_t)a;
                // (all rows of the atom in one round trip: see ld3)
                const Row3 ver = ld3(A.v_t, f3), qtr = ld3(A.q_t, f3), fr = ld3(A.f, e3), lvr = ld3(A.lv, e3), lqr = ld3(A.lq, e3),
                           dqr = ld3(A.dq, e3), qar = ld3(qbA, 3 * (size_t)a), qbr = ld3(qbB, 3 * (size_t)a);
                const float ve_[3] = {ver.x, ver.y, ver.z}, qt_[3] = {qtr.x, qtr.y, qtr.z}, f_[3] = {fr.x, fr.y, fr.z};
                const float lv_[3] = {lvr.x, lvr.y, lvr.z}, lq_[3] = {lqr.x, lqr.y, lqr.z}, dq_[3] = {dqr.x, dqr.y, dqr.z};
                const float qa_[3] = {qar.x, qar.y, qar.z}, qb_[3] = {qbr.x, qbr.y, qbr.z};
                float qn[3], vmo[3], lvho[3], wlo[3], lqho[3], mv2 = 0.f, mv2B = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float ve = ve_[c];
                    if (nhc) {
                        const float p = ve * m;
                        const float acc = (f_[c] - pv0 * p / A.prm.Q[0]) / m;
                        const float Gv = -(pv0 / A.prm.Q[0]) * lv_[c] + lq_[c] + 2.f * m * ve * lp0;
                        const float vhalf = 0.5f * (-acc) * h;                           // :132
                        qn[c] = qt_[c] + (ve + vhalf) * h;                               // :138 (forward-time sign)
                        vmo[c] = ve + vhalf;
                        const float lvh_ = lv_[c] + Gv * 0.5f * h;                       // :141
                        lvho[c] = lvh_;
                        wlo[c] = lvh_ * (1.0f / m);                                      // (the listed midpoint evaluation's w)
                        lqho[c] = lq_[c] + dq_[c] * 0.5f * h;                            // :142
                    } else {
                        const float vhalf = ve - 0.5f * (-f_[c]) * h;                    // :49-50
                        qn[c] = qt_[c] - vhalf * h;                                      // :51-52
                        vmo[c] = vhalf;
                        const float dx = dq_[c] * h * 0.5f;                              // :71
                        const float lvh_ = lv_[c] + (lq_[c] + dx) * h;                   // :72
                        lvho[c] = lvh_;
                        wlo[c] = lvh_;
                        lqho[c] = lq_[c] + dx;
                    }
                    const float mv = qn[c] - qa_[c], mvB = qn[c] - qb_[c];
                    mv2 = fmaf(mv, mv, mv2);
                    mv2B = fmaf(mvB, mvB, mv2B);
                }
                st3(A.vm, e3, vmo[0], vmo[1], vmo[2]);
                st3(A.lvh, e3, lvho[0], lvho[1], lvho[2]);
                if (A.nl_idx) st3(A.wl, e3, wlo[0], wlo[1], wlo[2]);
                st3(A.lqh, e3, lqho[0], lqho[1], lqho[2]);
                st3(A.qm, e3, qn[0], qn[1], qn[2]);
                far2 = fmaxf(far2, mv2);
                far2B = fmaxf(far2B, mv2B);
                px[u] = qn[0]; py[u] = qn[1]; pz[u] = qn[2];
            }
        }
        if (A.nl_idx) {
            // the stored candidates that serve frame i_fr (searched with rc + skin) stay a superset of the midpoint's pair
            // set while no atom is farther than skin / 2 from where the list was built
            const float lim2 = 0.25f * A.skin * A.skin;
            const int movedA = __syncthreads_or(!(far2 <= lim2)), movedB = __syncthreads_or(!(far2B <= lim2));
            if (threadIdx.x == 0) {                                     // (several workgroups share a replica here)
                if (movedA) atomicOr(&A.nl_state[2 * rep], 1);
                if (movedB) atomicOr(&A.nl_state[2 * rep + 1], 1);
            }
            return;
        }
    }
    if (PHASE == 4 || nc == 0) return;                      // (all-atom scan: nothing to bin)
    if (A.st_row && !A.st_bin) return;                      // (stale lists: the coming evaluation runs over the stored rows)

    // ---- binning of (px, py, pz)
    __syncthreads();
    for (int c = threadIdx.x; c <= LG_MAX_CELLS; c += LG_PREP) start[c] = 0;
    __syncthreads();
    int bsl[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int a = tid + u * stride;
        if (a < N) {
            const int bx = bin_coord_l(px[u], A.cell.inv[0], A.nb[0]);
            const int by = bin_coord_l(py[u], A.cell.inv[4], A.nb[1]);
            const int bz = bin_coord_l(pz[u], A.cell.inv[8], A.nb[2]);
            const int bin = (bx * A.nb[1] + by) * A.nb[2] + bz;
            bsl[u] = (atomicAdd(&start[bin], 1) << 12) | bin;
        }
    }
    __syncthreads();
    // exclusive scan of the bin counts: 4 consecutive bins per thread + scan of the thread totals (16 waves)
    int loc[4], tot = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { loc[u] = start[threadIdx.x * 4 + u]; tot += loc[u]; }
    {
        int x = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if ((int)(threadIdx.x & 63) >= o) x += y; }
        tsum[threadIdx.x] = x;                                           // inclusive within the wave
    }
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += tsum[w * 64 + 63];
    int run = base + tsum[threadIdx.x] - tot;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) { start[threadIdx.x * 4 + u] = run; run += loc[u]; }
    if (threadIdx.x == LG_PREP - 1) start[LG_MAX_CELLS] = run;
    __syncthreads();
    int32_t* bs = A.bstart + (size_t)rep * (LG_MAX_CELLS + 1);
    for (int c = threadIdx.x; c <= nc; c += LG_PREP) bs[c] = start[c];
    // slot inside the bin = the atom's RANK by index among its bin mates (the atomic's order is not reproducible): the
    // sorted array is then the same on every run, and so is every sum taken in its order (large_search_rows)
    int32_t* prov = A.binslot + (size_t)rep * N;
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int a = tid + u * stride;
        if (a < N) prov[start[bsl[u] & 4095] + (bsl[u] >> 12)] = a;
    }
    __syncthreads();                                         // (one workgroup per replica bins: its global writes are visible)
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int a = tid + u * stride;
        if (a < N) {
            const int bin = bsl[u] & 4095, s0 = start[bin], s1 = start[bin + 1];
            int rank = 0;
            for (int l = s0; l < s1; ++l) rank += prov[l] < a;
            A.spos[(size_t)rep * N + s0 + rank] = make_float4(px[u], py[u], pz[u], __int_as_float(a));
        }
    }
}

// Every pair term of one accepted candidate: F_i += phi'/r D; (LEVEL 2) -(H w) and the parameter vjp with
// w_ij = w_i - w_j  (w = lam_v / m for NHC, lam_v for NVE).
template <int LEVEL, int KIND>
__device__ __forceinline__ void pair_terms(const LargeArgs& A, const TermConst (&tc)[MDG_MAX_TERMS], int nt, int N, int i, int j,
                                           float dx, float dy, float dz, float d2, float wxi, float wyi, float wzi, float wjx,
                                           float wjy, float wjz, float gw, int rep, float& fx, float& fy, float& fz,
                                           float& gx, float& gy, float& gz, float (&th)[LG_KMAX]) {
    constexpr int NTC = KIND >= 0 ? 1 : MDG_MAX_TERMS;          // (the single-term specialisations carry one term)
#pragma unroll
    for (int m = 0; m < NTC; ++m) {
        if (m >= nt) break;
        if (!(d2 < tc[m].rc2)) continue;
        const uint8_t* mk = KIND >= 0 ? nullptr : A.terms.t[m].mask;
        if (mk && !mk[(size_t)i * N + j]) continue;
        PairOut o;
        float r, ir;
        pair_eval<LEVEL, KIND>(tc[m], d2, r, ir, o);
        if (KIND < 0 && tc[m].kind == MDG_PAIR_TABLE && d2 < tc[m].k0) A.flags[3] = 1;   // below the first table node
        const float c1 = o.du * ir;
        fx = fmaf(c1, dx, fx); fy = fmaf(c1, dy, fy); fz = fmaf(c1, dz, fz);
        if (LEVEL >= 2) {
            const float rx = -dx * ir, ry = -dy * ir, rz = -dz * ir;
            const float ax = wxi - wjx, ay = wyi - wjy, az = wzi - wjz;
            const float a = rx * ax + ry * ay + rz * az;
            const float c2 = o.d2u * a - c1 * a;
            gx -= c2 * rx + c1 * ax; gy -= c2 * ry + c1 * ay; gz -= c2 * rz + c1 * az;
            if (KIND < 0 && tc[m].kind == MDG_PAIR_TABLE) {
                // table gradient: d(w.F)/dnode = 1/2 (D.w_ij) basis, scattered in fixed point (one int64 word per
                // entry, see traj_small.hip) with integer global atomics: order-independent
                if (gw != 0.f) {
                    const float x = -gw * a * r;                         // D . w_ij = -a r
                    unsigned long long* g = A.g64 + (size_t)rep * A.terms.n_theta_total + 2 * o.tg;
#pragma unroll
                    for (int b_ = 0; b_ < 4; ++b_) {
                        const float val = x * o.tb[b_];
                        if (!(fabsf(val) < A.glim)) atomicOr(&A.flags[2], 1);         // out of range (or not a number)
                        atomicAdd(g + b_, fx64(val));
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < MDG_MAX_THETA; ++p)
                    if (p < A.terms.t[m].n_theta) th[m * MDG_MAX_THETA + p] -= 0.5f * o.ddu_dth[p] * a;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One wave: neighbours of atom i from the LDS-staged tiles, then force (LEVEL 1) or force + HVP +
// parameter vjp (LEVEL 2) over the compact list.  Results valid on every lane after the call.
//   F_i = -dU/dq_i ; dq_i = d(w.F)/dq_i = -(H w)_i with w = lam_v / m (NVE: w = lam_v) ; th += d(w.F)/dtheta partial
// LIST 0: search (cutoff^2 = rc2max).  LIST 1 (forward): search, then store the ascending indices for frame `frame`
// (large_adj_listed is the adjoint's consumer).
template <bool DIAG, int LEVEL, int KIND = -1, int LIST = 0>
__device__ __forceinline__ void wave_neighbours_and_force(
    const LargeArgs& A, const float* __restrict__ q, const float* __restrict__ lam, int i, bool valid,
    float* tile, float4* buf, float& fx, float& fy, float& fz, float& gx, float& gy, float& gz,
    float (&th)[LG_KMAX], const TermConst (&tc)[MDG_MAX_TERMS], float rc2max, float gw = 0.f, int rep = 0, int frame = 0) {
    const int N = A.prm.n_atoms, lane = threadIdx.x & 63;
    const float xi = valid ? q[3 * i] : 0.f, yi = valid ? q[3 * i + 1] : 0.f, zi = valid ? q[3 * i + 2] : 0.f;
    int n = 0;
    if (DIAG && A.ncell > 0) {
        // ---- cell-binned scan: the 3 x 3 stencil columns around atom i's bin, each column's three z-bins being one
        // contiguous range of the (bin, index)-sorted positions (two when it wraps); same pair test as below
        if (valid) {
            const float iv0 = A.cell.inv[0], iv1 = A.cell.inv[4], iv2 = A.cell.inv[8];
            const float h0 = A.cell.h[0], h1 = A.cell.h[4], h2 = A.cell.h[8];
            const float4* sp = A.spos + (size_t)rep * N;
            const int32_t* bs = A.bstart + (size_t)rep * (LG_MAX_CELLS + 1);
            const int nbx = A.nb[0], nby = A.nb[1], nbz = A.nb[2];
            // the atom's bin is the same on every lane: keep the stencil arithmetic on the scalar unit (and off integer
            // division: the wrapped neighbours of a bin are one compare-and-add away)
            const int bx = __builtin_amdgcn_readfirstlane(bin_coord_l(xi, iv0, nbx));
            const int by = __builtin_amdgcn_readfirstlane(bin_coord_l(yi, iv1, nby));
            const int bz = __builtin_amdgcn_readfirstlane(bin_coord_l(zi, iv2, nbz));
            const int zl0 = max(bz - 1, 0), zh0 = min(bz + 1, nbz - 1);          // in-range part of the z column
            const int zw = bz == 0 ? nbz - 1 : (bz == nbz - 1 ? 0 : -1);          // wrapped remainder (or none)
            // one candidate of every range per lane is requested BEFORE any of them is tested (the ranges hold ~57
            // atoms: the nine loads of a batch are in flight together instead of nine dependent round trips to L2);
            // batch 0 = the in-range z-parts of the 9 columns, batch 1 = their wrapped remainders (border bins only)
            auto test = [&](const float4 pj, bool live) {
                bool ok = false;
                float dx = 0.f, dy = 0.f, dz = 0.f;
                const int j = __float_as_int(pj.w);
                if (live) {
                    f32x2 ddx = f32x2{pj.x, 0.f} - xi, ddy = f32x2{pj.y, 0.f} - yi, ddz = f32x2{pj.z, 0.f} - zi;
                    ddx = min_image_diag2(ddx, iv0, h0); ddy = min_image_diag2(ddy, iv1, h1); ddz = min_image_diag2(ddz, iv2, h2);
                    const f32x2 d2 = norm2_ref2(ddx, ddy, ddz);
                    dx = ddx.x; dy = ddy.x; dz = ddz.x;
                    ok = (j != i) & (d2.x < rc2max) & (d2.x != 0.f);
                }
                const unsigned long long bal = __ballot(ok);
                if (ok) {
                    const int k = n + __popcll(bal & ((1ull << lane) - 1ull));
                    if (k < LG_CAP) buf[k] = make_float4(dx, dy, dz, __int_as_float(j));
                }
                n += __popcll(bal);
            };
            for (int part = 0; part < 2; ++part) {
                if (part && zw < 0) break;
                const int zlo = part ? zw : zl0, zhi = part ? zw : zh0;
                int a0[9], a1[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    int cx = bx + c / 3 - 1, cy = by + c % 3 - 1;
                    cx += cx < 0 ? nbx : 0; cx -= cx >= nbx ? nbx : 0;
                    cy += cy < 0 ? nby : 0; cy -= cy >= nby ? nby : 0;
                    const int cb = (cx * nby + cy) * nbz;
                    a0[c] = bs[cb + zlo]; a1[c] = bs[cb + zhi + 1];
                }
                float4 pj[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    const int idx = a0[c] + lane;
                    pj[c] = idx < a1[c] ? sp[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    test(pj[c], a0[c] + lane < a1[c]);
                    for (int a = a0[c] + 64; a < a1[c]; a += 64) {      // ranges longer than a wave (dense bins)
                        const int idx = a + lane;
                        test(idx < a1[c] ? sp[idx] : make_float4(0.f, 0.f, 0.f, 0.f), idx < a1[c]);
                    }
                }
            }
            // ascending neighbour index (entries are distinct): rank sort inside the wave's buffer
            const int m = n < LG_CAP ? n : LG_CAP;
            float4 mine[LG_CAP / 64];
            int rank[LG_CAP / 64];
#pragma unroll
            for (int u = 0; u < LG_CAP / 64; ++u) {
                const int k = lane + 64 * u;
                rank[u] = 0;
                mine[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < m) {
                    mine[u] = buf[k];
                    const int key = __float_as_int(mine[u].w);
                    for (int l = 0; l < m; ++l) rank[u] += __float_as_int(buf[l].w) < key;
                }
            }
#pragma unroll
            for (int u = 0; u < LG_CAP / 64; ++u)
                if (lane + 64 * u < m) buf[rank[u]] = mine[u];
        }
    } else
    for (int t0 = 0; t0 < N; t0 += LG_TILE) {
        const int tn = min(LG_TILE, N - t0);
        __syncthreads();
        for (int e = threadIdx.x; e < 3 * tn; e += blockDim.x) tile[(e % 3) * LG_TILE + e / 3] = q[3 * t0 + e];
        __syncthreads();
        if (valid && DIAG) {
            // two candidates per lane (jl, jl + 64) in packed fp32 -- the same arithmetic, and the same
            // ascending-j order of the compacted list, as the scalar loop below
            const float iv0 = A.cell.inv[0], iv1 = A.cell.inv[4], iv2 = A.cell.inv[8];
            const float h0 = A.cell.h[0], h1 = A.cell.h[4], h2 = A.cell.h[8];
            for (int c0 = 0; c0 < tn; c0 += 128) {                 // (LG_TILE is a multiple of 128: reads stay in the tile)
                const int jlA = c0 + lane, jlB = jlA + 64;
                f32x2 dx = f32x2{tile[jlA], tile[jlB]} - xi, dy = f32x2{tile[LG_TILE + jlA], tile[LG_TILE + jlB]} - yi,
                      dz = f32x2{tile[2 * LG_TILE + jlA], tile[2 * LG_TILE + jlB]} - zi;   // D = x_j - x_i
                dx = min_image_diag2(dx, iv0, h0); dy = min_image_diag2(dy, iv1, h1); dz = min_image_diag2(dz, iv2, h2);
                const f32x2 d2 = norm2_ref2(dx, dy, dz);
                const bool okA = (jlA < tn) & (t0 + jlA != i) & (d2.x < rc2max) & (d2.x != 0.f);
                const bool okB = (jlB < tn) & (t0 + jlB != i) & (d2.y < rc2max) & (d2.y != 0.f);
                const unsigned long long below = (1ull << lane) - 1ull;
                const unsigned long long bA = __ballot(okA);
                if (okA) {
                    const int k = n + __popcll(bA & below);
                    if (k < LG_CAP) buf[k] = make_float4(dx.x, dy.x, dz.x, __int_as_float(t0 + jlA));
                }
                n += __popcll(bA);
                const unsigned long long bB = __ballot(okB);
                if (okB) {
                    const int k = n + __popcll(bB & below);
                    if (k < LG_CAP) buf[k] = make_float4(dx.y, dy.y, dz.y, __int_as_float(t0 + jlB));
                }
                n += __popcll(bB);
            }
        } else if (valid) {
            for (int c0 = 0; c0 < tn; c0 += 64) {
                const int jl = c0 + lane, j = t0 + jl;
                bool ok = false;
                float dx = 0.f, dy = 0.f, dz = 0.f;
                if (jl < tn && j != i) {
                    dx = tile[jl] - xi; dy = tile[LG_TILE + jl] - yi; dz = tile[2 * LG_TILE + jl] - zi;   // D = x_j - x_i
                    min_image<DIAG>(A.cell, dx, dy, dz);
                    const float d2 = norm2_ref(dx, dy, dz);
                    ok = (d2 < rc2max) && (d2 != 0.f);
                }
                const unsigned long long b = __ballot(ok);
                if (ok) {
                    const int k = n + __popcll(b & ((1ull << lane) - 1ull));
                    if (k < LG_CAP) buf[k] = make_float4(dx, dy, dz, __int_as_float(j));
                }
                n += __popcll(b);
            }
        }
    }
    if (n > LG_CAP) { if (lane == 0) atomicMax(&A.flags[0], n); n = LG_CAP; }
    if constexpr (LIST == 1) {
        if (valid && A.nl_idx) {
            // (all-atom scan: the buffer is in ascending index order by construction; cell scan: rank-sorted above)
            const size_t at = ((size_t)rep * A.prm.n_frames + frame) * N + i;
            if (n <= LG_LIST) {
                for (int k = lane; k < n; k += 64) A.nl_idx[at * LG_LIST + k] = (uint16_t)__float_as_int(buf[k].w);
                if (lane == 0) A.nl_cnt[at] = n;
            } else if (lane == 0) { A.nl_bad[(size_t)rep * A.prm.n_frames + frame] = 1; A.flags[4] = 1; }
        }
    }
    fx = fy = fz = gx = gy = gz = 0.f;
    if (!valid) return;
    float wxi = 0.f, wyi = 0.f, wzi = 0.f;
    const bool nhc_w = A.prm.ensemble == 0;
    if (LEVEL >= 2) { const float im = nhc_w ? 1.0f / A.mass[i] : 1.0f; wxi = lam[3 * i] * im; wyi = lam[3 * i + 1] * im; wzi = lam[3 * i + 2] * im; }
    const int nt = A.terms.n_terms;
    for (int k = lane; k < n; k += 64) {
        const float4 e = buf[k];
        const float dx = e.x, dy = e.y, dz = e.z;
        const int j = __float_as_int(e.w);
        const float d2 = norm2_ref(dx, dy, dz);
        float wjx = 0.f, wjy = 0.f, wjz = 0.f;
        if (LEVEL >= 2) {
            const float jm = nhc_w ? 1.0f / A.mass[j] : 1.0f;
            wjx = lam[3 * j] * jm; wjy = lam[3 * j + 1] * jm; wjz = lam[3 * j + 2] * jm;
        }
        pair_terms<LEVEL, KIND>(A, tc, nt, N, i, j, dx, dy, dz, d2, wxi, wyi, wzi, wjx, wjy, wjz, gw, rep, fx, fy, fz, gx, gy, gz, th);
    }
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (LEVEL >= 2) { gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz); }
}

__device__ __forceinline__ float prepare_terms(const LargeArgs& A, TermConst (&tc)[MDG_MAX_TERMS]) {
    float rc2max = 0.f;
#pragma unroll
    for (int m = 0; m < MDG_MAX_TERMS; ++m)
        if (m < A.terms.n_terms) { tc[m] = term_prepare(A.terms.t[m], A.theta); rc2max = fmaxf(rc2max, tc[m].rc2); }
    return rc2max;
}

// ------------------------------------------------------------------------------------ stale lists
// One wave, atom i, topology_update_freq > 1 (torchmd/md.py:200-204).  A.st_rebuild (launch-uniform): the atom's pairs are
// searched at the current positions with generate_nbr_list's tests (topology.py:59-67: minimum image, un-contracted
// d^2 < rc^2 per term, != 0, the term's selection mask), written to its stale row in ascending j and evaluated; otherwise the
// stored row is evaluated with its frozen image flags whatever the distances are now.  The two give the same bits at the same
// positions (same D arithmetic, same order), which is what lets the forward loop keep a force across the two right-hand-side
// calls that share a state.  Built-in pair forms; no table kind.
template <bool DIAG, int LEVEL>
__device__ __forceinline__ void wave_stale_force(
    const LargeArgs& A, const float* __restrict__ q, const float* __restrict__ lam, int i, bool valid, float* tile, float4* buf,
    float& fx, float& fy, float& fz, float& gx, float& gy, float& gz, float (&th)[LG_KMAX],
    const TermConst (&tc)[MDG_MAX_TERMS], int rep) {
    const int N = A.prm.n_atoms, lane = threadIdx.x & 63, nt = A.terms.n_terms;
    const float xi = valid ? q[3 * i] : 0.f, yi = valid ? q[3 * i + 1] : 0.f, zi = valid ? q[3 * i + 2] : 0.f;
    uint32_t* row = A.st_row + ((size_t)rep * N + (valid ? i : 0)) * LG_CAP;
    int n = 0;
    if (A.st_rebuild) {
        auto test = [&](int j, bool live, float pjx, float pjy, float pjz) {
            bool ok = false;
            float dx = 0.f, dy = 0.f, dz = 0.f;
            unsigned code = 0;
            if (live && j != i) {
                dx = pjx - xi; dy = pjy - yi; dz = pjz - zi;                       // D = x_j - x_i
                const int img = min_image<DIAG>(A.cell, dx, dy, dz);
                const float d2 = norm2_ref(dx, dy, dz);
                unsigned bits = 0;
                if (d2 != 0.f) {                                                    // topology.py:67
#pragma unroll
                    for (int m = 0; m < MDG_MAX_TERMS; ++m) {
                        if (m >= nt) break;
                        const uint8_t* mk = A.terms.t[m].mask;
                        if (d2 < tc[m].rc2 && (!mk || mk[(size_t)i * N + j])) bits |= 1u << m;
                    }
                }
                ok = bits != 0;
                code = (unsigned)j | ((unsigned)img << 15) | (bits << 20);
            }
            const unsigned long long bal = __ballot(ok);
            if (ok) {
                const int k = n + __popcll(bal & ((1ull << lane) - 1ull));
                if (k < LG_CAP) buf[k] = make_float4(dx, dy, dz, __uint_as_float(code));
            }
            n += __popcll(bal);
        };
        if (DIAG && A.ncell > 0) {
            // cell-binned scan (the bins of large_prep: width >= the largest cutoff): the 3 x 3 columns around the atom's bin,
            // a column's three z-bins being one contiguous range of the sorted positions (two when it wraps)
            if (valid) {
                const float4* sp = A.spos + (size_t)rep * N;
                const int32_t* bs = A.bstart + (size_t)rep * (LG_MAX_CELLS + 1);
                const int nbx = A.nb[0], nby = A.nb[1], nbz = A.nb[2];
                const int bx = __builtin_amdgcn_readfirstlane(bin_coord_l(xi, A.cell.inv[0], nbx));
                const int by = __builtin_amdgcn_readfirstlane(bin_coord_l(yi, A.cell.inv[4], nby));
                const int bz = __builtin_amdgcn_readfirstlane(bin_coord_l(zi, A.cell.inv[8], nbz));
                const int zl0 = max(bz - 1, 0), zh0 = min(bz + 1, nbz - 1);
                const int zw = bz == 0 ? nbz - 1 : (bz == nbz - 1 ? 0 : -1);
                for (int c = 0; c < 9; ++c) {
                    int cx = bx + c / 3 - 1, cy = by + c % 3 - 1;
                    cx += cx < 0 ? nbx : 0; cx -= cx >= nbx ? nbx : 0;
                    cy += cy < 0 ? nby : 0; cy -= cy >= nby ? nby : 0;
                    const int cb = (cx * nby + cy) * nbz;
                    for (int part = 0; part < 2; ++part) {
                        if (part && zw < 0) break;
                        const int a0 = bs[cb + (part ? zw : zl0)], a1 = bs[cb + (part ? zw : zh0) + 1];
                        for (int a = a0; a < a1; a += 64) {
                            const int idx = a + lane;
                            const float4 pj = idx < a1 ? sp[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
                            test(__float_as_int(pj.w), idx < a1, pj.x, pj.y, pj.z);
                        }
                    }
                }
                // ascending neighbour index (entries are distinct): rank sort inside the wave's buffer
                const int m = n < LG_CAP ? n : LG_CAP;
                float4 mine[LG_CAP / 64];
                int rank[LG_CAP / 64];
#pragma unroll
                for (int u = 0; u < LG_CAP / 64; ++u) {
                    const int k = lane + 64 * u;
                    rank[u] = 0;
                    mine[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < m) {
                        mine[u] = buf[k];
                        const unsigned key = __float_as_uint(mine[u].w) & 32767u;
                        for (int l = 0; l < m; ++l) rank[u] += (__float_as_uint(buf[l].w) & 32767u) < key;
                    }
                }
#pragma unroll
                for (int u = 0; u < LG_CAP / 64; ++u)
                    if (lane + 64 * u < m) buf[rank[u]] = mine[u];
            }
        } else {
            for (int t0 = 0; t0 < N; t0 += LG_TILE) {
                const int tn = min(LG_TILE, N - t0);
                __syncthreads();
                for (int e = threadIdx.x; e < 3 * tn; e += blockDim.x) tile[(e % 3) * LG_TILE + e / 3] = q[3 * t0 + e];
                __syncthreads();
                if (valid)
                    for (int c0 = 0; c0 < tn; c0 += 64) {
                        const int jl = c0 + lane;
                        const bool live = jl < tn;
                        test(t0 + jl, live, live ? tile[jl] : 0.f, live ? tile[LG_TILE + jl] : 0.f, live ? tile[2 * LG_TILE + jl] : 0.f);
                    }
            }
        }
        if (n > LG_CAP) { if (lane == 0) atomicMax(&A.flags[0], n); n = LG_CAP; }
        if (valid) {
            for (int k = lane; k < n; k += 64) row[k] = __float_as_uint(buf[k].w);
            if (lane == 0) A.st_cnt[(size_t)rep * N + i] = n;
        }
    } else if (valid) {
        n = A.st_cnt[(size_t)rep * N + i];
        for (int k = lane; k < n; k += 64) {
            const unsigned code = row[k];
            const int j = (int)(code & 32767u), img = (int)((code >> 15) & 31u);
            const float ox = (float)(img % 3 - 1), oy = (float)((img / 3) % 3 - 1), oz = (float)(img / 9 - 1);
            float dx = q[3 * j] - xi, dy = q[3 * j + 1] - yi, dz = q[3 * j + 2] - zi;
            if (DIAG) {
                dx = fmaf(ox, A.cell.h[0], dx); dy = fmaf(oy, A.cell.h[4], dy); dz = fmaf(oz, A.cell.h[8], dz);
            } else {
                dx += fmaf(oz, A.cell.h[6], fmaf(oy, A.cell.h[3], ox * A.cell.h[0]));
                dy += fmaf(oz, A.cell.h[7], fmaf(oy, A.cell.h[4], ox * A.cell.h[1]));
                dz += fmaf(oz, A.cell.h[8], fmaf(oy, A.cell.h[5], ox * A.cell.h[2]));
            }
            buf[k] = make_float4(dx, dy, dz, __uint_as_float(code));
        }
    }
    fx = fy = fz = gx = gy = gz = 0.f;
    if (!valid) return;
    float wxi = 0.f, wyi = 0.f, wzi = 0.f;
    const bool nhc_w = A.prm.ensemble == 0;
    if (LEVEL >= 2) { const float im = nhc_w ? 1.0f / A.mass[i] : 1.0f; wxi = lam[3 * i] * im; wyi = lam[3 * i + 1] * im; wzi = lam[3 * i + 2] * im; }
    for (int k = lane; k < n; k += 64) {
        const float4 e = buf[k];
        const float dx = e.x, dy = e.y, dz = e.z;
        const unsigned code = __float_as_uint(e.w);
        const int j = (int)(code & 32767u);
        const float d2 = norm2_ref(dx, dy, dz);
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (LEVEL >= 2) {
            const float jm = nhc_w ? 1.0f / A.mass[j] : 1.0f;
            ax = wxi - lam[3 * j] * jm; ay = wyi - lam[3 * j + 1] * jm; az = wzi - lam[3 * j + 2] * jm;
        }
#pragma unroll
        for (int m = 0; m < MDG_MAX_TERMS; ++m) {
            if (m >= nt) break;
            if (!((code >> (20 + m)) & 1u)) continue;
            PairOut o;
            float r, ir;
            pair_eval<LEVEL, -1>(tc[m], d2, r, ir, o);
            const float c1 = o.du * ir;
            fx = fmaf(c1, dx, fx); fy = fmaf(c1, dy, fy); fz = fmaf(c1, dz, fz);
            if (LEVEL >= 2) {
                const float rx = -dx * ir, ry = -dy * ir, rz = -dz * ir;
                const float a = rx * ax + ry * ay + rz * az;
                const float c2 = o.d2u * a - c1 * a;
                gx -= c2 * rx + c1 * ax; gy -= c2 * ry + c1 * ay; gz -= c2 * rz + c1 * az;
#pragma unroll
                for (int p = 0; p < MDG_MAX_THETA; ++p)
                    if (p < A.terms.t[m].n_theta) th[m * MDG_MAX_THETA + p] -= 0.5f * o.ddu_dth[p] * a;
            }
        }
    }
    fx = wave_sum(fx); fy = wave_sum(fy); fz = wave_sum(fz);
    if (LEVEL >= 2) { gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz); }
}

// ------------------------------------------------------------------------------------ forward
// MODE 0: initial force at q0 + frame 0 + KE(v0) partials.   MODE 1: second half of step k.
// STALE (stale lists): wave_stale_force instead of the search; MODE 2: the force alone, again, at the positions of the step
// just finished -- the first right-hand-side call of the next step when that call rebuilds the lists (its force then differs
// from the one the second half of this step used).
template <bool DIAG, int MODE, int KIND, bool STALE = false>
__global__ __launch_bounds__(LG_BLOCK) void large_force_step(const LargeArgs A) {
    // dynamic LDS: [waves][LG_CAP] neighbour buffers, then [3][LG_TILE] position tiles for the all-atom scan (absent in cell mode)
    extern __shared__ __attribute__((aligned(16))) float4 nbuf[];
    float* tile = reinterpret_cast<float*>(nbuf + (blockDim.x >> 6) * LG_CAP);
    __shared__ float red[32];
    __shared__ float Qs[MDG_MAX_CHAINS], pvs[MDG_MAX_CHAINS];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, C = A.prm.n_chains, rep = blockIdx.y + A.rep0;
    if (MODE == 1 && A.nl_idx && !A.nl_state[2 * rep]) return;      // the current list serves this step (large_fwd_listed)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const size_t so = (size_t)rep * N * 3;
    float* q = A.q + so; float* v = A.v + so; float* vh = A.vh + so; float* f = A.f + so;
    float* pv = A.pv + rep * MDG_MAX_CHAINS; float* ph = A.ph + rep * MDG_MAX_CHAINS;
    float* pvh = A.pvh + rep * MDG_MAX_CHAINS;
    const int k = A.step;
    if (A.nl_idx && blockIdx.x == 0 && threadIdx.x == 0) {          // this launch builds the list of frame k + 1 (0)
        const int frame = MODE == 0 ? 0 : k + 1;
        A.nl_build[(size_t)rep * T + frame] = frame;
        A.nl_state[2 * rep + 1] = frame;
        if (MODE == 0) A.nl_state[2 * rep] = 1;
    }
    if (threadIdx.x < MDG_MAX_CHAINS) {
        float qv = 0.f;
#pragma unroll
        for (int c = 0; c < MDG_MAX_CHAINS; ++c) if (threadIdx.x == c) qv = A.prm.Q[c];
        Qs[threadIdx.x] = qv;
    }
    const bool nhc = A.prm.ensemble == 0;
    float dt = 0.f;
    if (MODE == 1) dt = A.t[k + 1] - A.t[k];
    // ---- block 0: finish the bath with KE(v + vh) from the previous launch's partials
    if (MODE == 1 && nhc && blockIdx.x == 0) {
        const float ke = 0.5f * reduce_partials(A.partB + (size_t)rep * A.nbE, A.nbE, 1, 0, red);
        if (threadIdx.x < C) pvs[threadIdx.x] = pvh[threadIdx.x];
        __syncthreads();
        if (threadIdx.x < C) {
            const float b1 = bath_rhs_l(A.prm, Qs, pvs, ke, threadIdx.x);
            const float np = pv[threadIdx.x] + (ph[threadIdx.x] + 0.5f * b1 * dt);
            pv[threadIdx.x] = np;
            A.pv_t[((size_t)rep * T + k + 1) * C + threadIdx.x] = np;
        }
    }
    if (MODE == 0 && nhc && blockIdx.x == 0 && threadIdx.x < C)
        A.pv_t[((size_t)rep * T) * C + threadIdx.x] = pv[threadIdx.x];
    TermConst tc[MDG_MAX_TERMS];
    const float rc2max = prepare_terms(A, tc);
    const int i = blockIdx.x * (blockDim.x >> 6) + wid;
    const bool valid = i < N;
    float fx, fy, fz, gx, gy, gz, th[LG_KMAX];
    const float rs = sqrtf(rc2max) + A.skin;                    // (skin 0 unless the lists are kept for the adjoint)
    if constexpr (STALE) {
        wave_stale_force<DIAG, 1>(A, q, nullptr, i, valid, tile, nbuf + wid * LG_CAP, fx, fy, fz, gx, gy, gz, th, tc, rep);
        if (MODE == 2) {
            if (valid && lane < 3) {
                const float F = lane == 0 ? fx : (lane == 1 ? fy : fz);
                f[3 * i + lane] = F;
                if (!isfinite(F)) A.flags[1] = 1;
            }
            return;
        }
    } else
    wave_neighbours_and_force<DIAG, 1, KIND, 1>(A, q, nullptr, i, valid, tile, nbuf + wid * LG_CAP, fx, fy, fz, gx, gy, gz,
                                                th, tc, rs * rs, 0.f, rep, MODE == 0 ? 0 : k + 1);
    float kepart = 0.f;
    if (valid && lane < 3) {
        const float F = lane == 0 ? fx : (lane == 1 ? fy : fz);
        const int e = 3 * i + lane;
        const float m = A.mass[i];
        float vn;
        if (MODE == 0) vn = v[e];
        else {
            const float vv = v[e] + vh[e];
            const float p = vv * m;
            const float a = nhc ? (F - pvh[0] * p / A.prm.Q[0]) / m : F;      // (NVE: md.py:145-148, no 1/m)
            vn = v[e] + (vh[e] + 0.5f * a * dt);
            v[e] = vn;
        }
        f[e] = F;
        const size_t fr = ((size_t)rep * T + (MODE == 0 ? 0 : k + 1)) * N * 3 + e;
        A.q_t[fr] = q[e];
        A.v_t[fr] = vn;
        const float p = vn * m;
        kepart = p * p / m;
        if (!(isfinite(vn) && isfinite(F))) A.flags[1] = 1;
    }
    kepart = block_sum(kepart, red);
    if (threadIdx.x == 0) A.partA[(size_t)rep * A.nbF + blockIdx.x] = kepart;
}

// Second half of step k over the CURRENT list (large_prep<1> found every atom inside its reuse ball): the same four-
// atoms-per-wave rows as large_adj_listed below, force only, then large_force_step<1>'s epilogue.  Exits at once when
// this step searched instead.

constexpr int LG_ROW_ATOMS = 16;                 // atoms per workgroup of the listed kernels (4 waves x 4 rows)
constexpr int LG_ADJ_GROUPS = 4;                 // ... large_adj_listed: groups of four atoms a wave takes one after the other
constexpr int LG_FWD_GROUPS = 4;                 // ... large_fwd_listed

template <bool DIAG, int KIND>
__global__ __launch_bounds__(256) void large_fwd_listed(const LargeArgs A) {
    constexpr int NP = LG_LIST / 16;
    __shared__ float red[32];
    __shared__ float Qs[MDG_MAX_CHAINS], pvs[MDG_MAX_CHAINS];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, C = A.prm.n_chains, rep = blockIdx.y + A.rep0, k = A.step;
    if (A.nl_state[2 * rep]) return;                                   // this step searched (large_force_step<1>)
    const int slot = A.nl_state[2 * rep + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, s = lane & 15;
    const size_t so = (size_t)rep * N * 3;
    float* q = A.q + so; float* v = A.v + so; float* vh = A.vh + so; float* f = A.f + so;
    float* pv = A.pv + rep * MDG_MAX_CHAINS; float* ph = A.ph + rep * MDG_MAX_CHAINS;
    float* pvh = A.pvh + rep * MDG_MAX_CHAINS;
    const bool nhc = A.prm.ensemble == 0;
    const float dt = A.t[k + 1] - A.t[k];
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) A.nl_build[(size_t)rep * T + k + 1] = slot;
        if (nhc) {                                                      // finish the bath with KE(v + vh), as large_force_step<1>
            if (threadIdx.x < MDG_MAX_CHAINS) {
                float qv = 0.f;
#pragma unroll
                for (int c = 0; c < MDG_MAX_CHAINS; ++c) if (threadIdx.x == c) qv = A.prm.Q[c];
                Qs[threadIdx.x] = qv;
            }
            const float ke = 0.5f * reduce_partials(A.partB + (size_t)rep * A.nbE, A.nbE, 1, 0, red);
            if (threadIdx.x < C) pvs[threadIdx.x] = pvh[threadIdx.x];
            __syncthreads();
            if (threadIdx.x < C) {
                const float b1 = bath_rhs_l(A.prm, Qs, pvs, ke, threadIdx.x);
                const float np = pv[threadIdx.x] + (ph[threadIdx.x] + 0.5f * b1 * dt);
                pv[threadIdx.x] = np;
                A.pv_t[((size_t)rep * T + k + 1) * C + threadIdx.x] = np;
            }
            __syncthreads();
        }
    }
    TermConst tc[MDG_MAX_TERMS];
    if (KIND >= 0) tc[0] = term_prepare(A.terms.t[0], A.theta);
    else prepare_terms(A, tc);
    const int ntl = KIND >= 0 ? 1 : A.terms.n_terms;
    // (a wave takes LG_FWD_GROUPS consecutive groups of four atoms: see large_adj_listed)
    float kepart = 0.f;
#pragma unroll 1
    for (int grp = 0; grp < LG_FWD_GROUPS; ++grp) {
    const int i = ((blockIdx.x * 4 + wid) * LG_FWD_GROUPS + grp) * 4 + (lane >> 4);
    if (__ballot(i < N) == 0ull) break;
    const bool valid = i < N;
    const int ic = valid ? i : N - 1;
    const size_t at = ((size_t)rep * T + slot) * N + ic;
    // (entry s + 16 p of the row: two 16-bit entries per word, lanes 2t and 2t + 1 read the same word)
    const uint32_t* idx = reinterpret_cast<const uint32_t*>(A.nl_idx + at * LG_LIST);
    int jj[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) jj[p] = (int)((idx[(s >> 1) + 8 * p] >> (16 * (s & 1))) & 0xffffu);
    const int n = valid ? min(A.nl_cnt[at], LG_LIST) : 0;
    const Row3 qi = row3(q, ic);
    float fx = 0.f, fy = 0.f, fz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f, th[LG_KMAX];
    if constexpr (KIND == KIND_LJ126 && DIAG) {
        // two candidates per lane and iteration in packed fp32 (see large_adj_listed)
        const TermConst& t0 = tc[0];
        const float sig2 = t0.k0 * t0.k0, e4 = 4.f * t0.k1, cq = t0.c, rc2 = t0.rc2;
        const float m1a = 6.f * e4 * cq, m1b = 12.f * e4;
        const float ivx = A.cell.inv[0], ivy = A.cell.inv[4], ivz = A.cell.inv[8];
        const float hx = A.cell.h[0], hy = A.cell.h[4], hz = A.cell.h[8];
        f32x2 fx2 = {0.f, 0.f}, fy2 = fx2, fz2 = fx2;
        constexpr int NPF = 6;                                         // rows requested up front (see large_adj_listed)
        auto pair2 = [&](const Row3 qA, const Row3 qB) {
            f32x2 dx = f32x2{qA.x, qB.x} - qi.x, dy = f32x2{qA.y, qB.y} - qi.y, dz = f32x2{qA.z, qB.z} - qi.z;
            dx = min_image_diag2(dx, ivx, hx); dy = min_image_diag2(dy, ivy, hy); dz = min_image_diag2(dz, ivz, hz);
            const f32x2 d2 = norm2_ref2(dx, dy, dz);
            const bool ok0 = (d2.x != 0.f) && (d2.x < rc2), ok1 = (d2.y != 0.f) && (d2.y < rc2);
            const f32x2 i2 = {ok0 ? __builtin_amdgcn_rcpf(d2.x) : 0.f, ok1 ? __builtin_amdgcn_rcpf(d2.y) : 0.f};
            const f32x2 s2 = sig2 * i2;
            const f32x2 s6 = s2 * s2 * s2;
            const f32x2 c1 = (m1a * s6 - m1b * (s6 * s6)) * i2;
            fx2 += c1 * dx; fy2 += c1 * dy; fz2 += c1 * dz;
        };
        {
            Row3 qv[NPF];
#pragma unroll
            for (int p = 0; p < NPF; ++p) qv[p] = row3(q, s + 16 * p < n ? jj[p] : ic);
#pragma unroll
            for (int p = 0; p < NPF; p += 2) {
                if (p > 0 && __ballot(s + 16 * p < n) == 0) break;
                pair2(qv[p], qv[p + 1]);
            }
        }
        if (__ballot(s + 16 * NPF < n) != 0) {
            Row3 qv[NP - NPF];
#pragma unroll
            for (int p = NPF; p < NP; ++p) qv[p - NPF] = row3(q, s + 16 * p < n ? jj[p] : ic);
#pragma unroll
            for (int p = 0; p < NP - NPF; p += 2) pair2(qv[p], qv[p + 1]);
        }
        fx = fx2.x + fx2.y; fy = fy2.x + fy2.y; fz = fz2.x + fz2.y;
    } else {
    int jn = s < n ? jj[0] : ic;
    Row3 qn = row3(q, jn);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p > 0 && __ballot(s + 16 * p < n) == 0) break;
        const int j = jn;
        float dx = qn.x - qi.x, dy = qn.y - qi.y, dz = qn.z - qi.z;
        if (p + 1 < NP) {
            jn = s + 16 * (p + 1) < n ? jj[p + 1] : ic;
            qn = row3(q, jn);
        }
        min_image<DIAG>(A.cell, dx, dy, dz);
        const float d2 = norm2_ref(dx, dy, dz);
        if (d2 == 0.f) continue;
        pair_terms<1, KIND>(A, tc, ntl, N, i, j, dx, dy, dz, d2, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, rep, fx, fy, fz, gx, gy, gz, th);
    }
    }
    fx = row16_sum(fx); fy = row16_sum(fy); fz = row16_sum(fz);
    if (valid && s < 3) {
        const float F = s == 0 ? fx : (s == 1 ? fy : fz);
        const int e = 3 * i + s;
        const float m = A.mass[i];
        const float vv = v[e] + vh[e];
        const float p = vv * m;
        const float a = nhc ? (F - pvh[0] * p / A.prm.Q[0]) / m : F;          // (NVE: md.py:145-148, no 1/m)
        const float vn = v[e] + (vh[e] + 0.5f * a * dt);
        v[e] = vn;
        f[e] = F;
        const size_t fr = ((size_t)rep * T + k + 1) * N * 3 + e;
        A.q_t[fr] = q[e];
        A.v_t[fr] = vn;
        const float pn = vn * m;
        kepart += pn * pn / m;
        if (!(isfinite(vn) && isfinite(F))) A.flags[1] = 1;
    }
    }                                                                  // groups of this wave
    kepart = wave_sum_rows(kepart);
    if (lane == 0) red[wid] = kepart;
    __syncthreads();
    // partA holds A.nbL rows for the listed launches (large_search_rows writes one per 16 atoms): this workgroup's sum goes to
    // its first row, its other rows are cleared
    if (threadIdx.x < LG_FWD_GROUPS) {
        const int row = blockIdx.x * LG_FWD_GROUPS + threadIdx.x;
        if (row < A.nbL) A.partA[(size_t)rep * A.nbF + row] = threadIdx.x == 0 ? (red[0] + red[1]) + (red[2] + red[3]) : 0.f;
    }
}

// The forward force launch that SEARCHES (first frame, and every Verlet rebuild) in a binned box with kept lists: the
// rows of the listed kernels -- four atoms per wave, a 16-lane DPP row each, atoms taken in sorted-slot order so that a
// workgroup's sixteen atoms share their stencil.  An atom's 9 stencil columns are contiguous ranges of the sorted
// positions (its three z-bins; the bin reached through a z face in a second round); the 18 bounds of a round are
// requested together, the next column's four 16-entry chunks while the current column is tested.  One sweep does
// everything large_force_step spreads over a search, a rank sort, a store and a second pass over an LDS buffer: a
// candidate inside rc + skin is appended to the atom's stored row (slot = the row's running count + the prefix of the
// row's ballot bits) and, inside the exact cutoff, its force is added on the spot.  The order of a stored row is the
// order of the sweep over the (bin, atom index)-sorted array: reproducible, not ascending -- its consumers do not care.
template <int MODE, int KIND>
__global__ __launch_bounds__(256) void large_search_rows(const LargeArgs A) {
    __shared__ float red[32];
    __shared__ float Qs[MDG_MAX_CHAINS], pvs[MDG_MAX_CHAINS];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, C = A.prm.n_chains, rep = blockIdx.y + A.rep0, k = A.step;
    if (MODE == 1 && !A.nl_state[2 * rep]) return;                  // the current list serves this step (large_fwd_listed)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, s = lane & 15, row = lane >> 4;
    const size_t so = (size_t)rep * N * 3;
    float* q = A.q + so; float* v = A.v + so; float* vh = A.vh + so; float* f = A.f + so;
    float* pv = A.pv + rep * MDG_MAX_CHAINS; float* ph = A.ph + rep * MDG_MAX_CHAINS;
    float* pvh = A.pvh + rep * MDG_MAX_CHAINS;
    const bool nhc = A.prm.ensemble == 0;
    const int frame = MODE == 0 ? 0 : k + 1;
    float dt = 0.f;
    if (MODE == 1) dt = A.t[k + 1] - A.t[k];
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {                                      // this launch builds the list of `frame`
            A.nl_build[(size_t)rep * T + frame] = frame;
            A.nl_state[2 * rep + 1] = frame;
            if (MODE == 0) A.nl_state[2 * rep] = 1;
        }
        if (A.tile_cap) {                                            // the build's bin columns, for the tiles of its consumers
            const int32_t* bs0 = A.bstart + (size_t)rep * (LG_MAX_CELLS + 1);
            int32_t* dst = A.nl_bst + ((size_t)rep * T + frame) * (LG_MAX_COLS + 1);
            for (int c = threadIdx.x; c <= A.ncol; c += blockDim.x) dst[c] = bs0[c * A.nb[2]];
        }
        if (MODE == 1 && nhc) {                                      // finish the bath with KE(v + vh), as large_force_step<1>
            if (threadIdx.x < MDG_MAX_CHAINS) {
                float qv = 0.f;
#pragma unroll
                for (int c = 0; c < MDG_MAX_CHAINS; ++c) if (threadIdx.x == c) qv = A.prm.Q[c];
                Qs[threadIdx.x] = qv;
            }
            const float ke = 0.5f * reduce_partials(A.partB + (size_t)rep * A.nbE, A.nbE, 1, 0, red);
            if (threadIdx.x < C) pvs[threadIdx.x] = pvh[threadIdx.x];
            __syncthreads();
            if (threadIdx.x < C) {
                const float b1 = bath_rhs_l(A.prm, Qs, pvs, ke, threadIdx.x);
                const float np = pv[threadIdx.x] + (ph[threadIdx.x] + 0.5f * b1 * dt);
                pv[threadIdx.x] = np;
                A.pv_t[((size_t)rep * T + k + 1) * C + threadIdx.x] = np;
            }
            __syncthreads();
        }
        if (MODE == 0 && nhc && threadIdx.x < C) A.pv_t[((size_t)rep * T) * C + threadIdx.x] = pv[threadIdx.x];
    }
    TermConst tc[MDG_MAX_TERMS];
    float rc2max;
    if (KIND >= 0) { tc[0] = term_prepare(A.terms.t[0], A.theta); rc2max = tc[0].rc2; }
    else rc2max = prepare_terms(A, tc);
    const int ntl = KIND >= 0 ? 1 : A.terms.n_terms;
    const float rs = sqrtf(rc2max) + A.skin, rs2 = rs * rs;
    const int slot = (blockIdx.x * 4 + wid) * 4 + row;
    const bool valid = slot < N;
    const float4* sp = A.spos + (size_t)rep * N;
    const int32_t* bs = A.bstart + (size_t)rep * (LG_MAX_CELLS + 1);
    const float4 pi = sp[valid ? slot : N - 1];
    const int i = __float_as_int(pi.w);
    // (column tiles: the row lives at the atom's SORTED SLOT and holds staged-tile slots; see LargeArgs)
    const size_t at = ((size_t)rep * T + frame) * N + (A.tile_cap ? (valid ? slot : N - 1) : i);
    uint16_t* lrow = A.nl_idx + at * LG_LIST;
    int cnt = 0;                                                     // candidates of the row's atom so far (row-uniform)
    float fx = 0.f, fy = 0.f, fz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f, th[LG_KMAX];
    // every lane runs this for every chunk (live = it holds a candidate): the row's ballot bits give the slots
    auto take = [&](const float4 pj, bool live, int loc) {
        const int j = __float_as_int(pj.w);
        float dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;            // D = x_j - x_i
        min_image<true>(A.cell, dx, dy, dz);
        const float d2 = norm2_ref(dx, dy, dz);
        const bool in = live && j != i && d2 < rs2;
        const unsigned rowmask = (unsigned)(__ballot(in) >> (16 * row)) & 0xffffu;
        if (in) {
            const int at_ = cnt + __popc(rowmask & ((1u << s) - 1u));
            if (at_ < LG_LIST) lrow[at_] = (uint16_t)(A.tile_cap ? loc : j);
            if (d2 != 0.f)                                                    // topology.py:67
                pair_terms<1, KIND>(A, tc, ntl, N, i, j, dx, dy, dz, d2, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, rep, fx, fy, fz, gx,
                                    gy, gz, th);
        }
        cnt += __popc(rowmask);
    };
    {
        const int nbx = A.nb[0], nby = A.nb[1], nbz = A.nb[2];
        const int bx = bin_coord_l(pi.x, A.cell.inv[0], nbx), by = bin_coord_l(pi.y, A.cell.inv[4], nby);
        const int bz = bin_coord_l(pi.z, A.cell.inv[8], nbz);
        const int wrapped = bz == 0 ? nbz - 1 : (bz == nbz - 1 ? 0 : -1);     // the z bin reached through the face
        int col[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            int cx = bx + c / 3 - 1, cy = by + c % 3 - 1;
            cx = cx < 0 ? cx + nbx : (cx >= nbx ? cx - nbx : cx);
            cy = cy < 0 ? cy + nby : (cy >= nby ? cy - nby : cy);
            col[c] = (cx * nby + cy) * nbz;
        }
        // the row's column tile: the nine stencil columns in stencil order, each the contiguous range [bs[col], bs[col + nbz])
        // of the sorted array; staged slot of sorted slot a of column c = cb[c] + (a - bs[col[c]])
        int shift[9];
        {
            int run = 0;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const int g0 = bs[col[c]], g1 = bs[col[c] + nbz];
                shift[c] = run - g0;
                run += g1 - g0;
            }
            if (A.tile_cap && valid && s == 0) {
                if (run > A.tile_cap || run > 65535) {                        // the tile does not fit the staged capacity
                    A.nl_bad[(size_t)rep * T + frame] = 1; A.flags[4] = 1;
                }
                A.nl_perm[((size_t)rep * T + frame) * N + slot] = i;
            }
        }
#pragma unroll 1
        for (int part = 0; part < 2; ++part) {
            const bool on = valid && (part == 0 || wrapped >= 0);             // (rows without an atom / a face: empty ranges)
            if (part == 1 && !__any(on)) break;
            const int zlo = part ? max(wrapped, 0) : max(bz - 1, 0), zhi = part ? max(wrapped, 0) : min(bz + 1, nbz - 1);
            int a0[9], a1[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) { a0[c] = bs[col[c] + zlo]; a1[c] = bs[col[c] + zhi + 1]; }
#pragma unroll
            for (int c = 0; c < 9; ++c) { a0[c] += s; if (!on) a1[c] = 0; }
            float4 P[4], Q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int a = a0[0] + 16 * u; P[u] = sp[a < a1[0] ? a : 0]; }
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                if (c + 1 < 9) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int a = a0[c + 1] + 16 * u; Q[u] = sp[a < a1[c + 1] ? a : 0]; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (__any(a0[c] + 16 * u < a1[c])) take(P[u], a0[c] + 16 * u < a1[c], a0[c] + 16 * u + shift[c]);
                for (int a = a0[c] + 64; __any(a < a1[c]); a += 16) take(sp[a < a1[c] ? a : 0], a < a1[c], a + shift[c]);   // (dense bins)
#pragma unroll
                for (int u = 0; u < 4; ++u) P[u] = Q[u];
            }
        }
    }
    fx = row16_sum(fx); fy = row16_sum(fy); fz = row16_sum(fz);
    if (valid && s == 0) {
        if (cnt <= LG_LIST) A.nl_cnt[at] = cnt;
        else { A.nl_bad[(size_t)rep * T + frame] = 1; A.flags[4] = 1; }      // (the row does not hold them all: see nl_bad)
    }
    float kepart = 0.f;
    if (valid && s < 3) {
        const float F = s == 0 ? fx : (s == 1 ? fy : fz);
        const int e = 3 * i + s;
        const float m = A.mass[i];
        float vn;
        if (MODE == 0) vn = v[e];
        else {
            const float vv = v[e] + vh[e];
            const float p = vv * m;
            const float a = nhc ? (F - pvh[0] * p / A.prm.Q[0]) / m : F;      // (NVE: md.py:145-148, no 1/m)
            vn = v[e] + (vh[e] + 0.5f * a * dt);
            v[e] = vn;
        }
        f[e] = F;
        const size_t fr = ((size_t)rep * T + frame) * N * 3 + e;
        A.q_t[fr] = q[e];
        A.v_t[fr] = vn;
        const float pn = vn * m;
        kepart = pn * pn / m;
        if (!(isfinite(vn) && isfinite(F))) A.flags[1] = 1;
    }
    kepart = wave_sum_rows(kepart);
    if (lane == 0) red[wid] = kepart;
    __syncthreads();
    if (threadIdx.x == 0) A.partA[(size_t)rep * A.nbF + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    // (the rows the column-tile launches write beyond this grid: cleared, large_prep<1> sums A.nbL of them)
    if (blockIdx.x == 0) for (int row = gridDim.x + threadIdx.x; row < A.nbL; row += blockDim.x) A.partA[(size_t)rep * A.nbF + row] = 0.f;
}

// ------------------------------------------------------------------------------------ adjoint
// force + HVP + parameter vjp at (qsrc, vsrc ; lam) -> f, dq, per-block partials
// by a fresh search (large_adj_listed below evaluates the forward pass's stored candidates instead).
template <bool DIAG, int KIND, bool STALE = false>
__global__ __launch_bounds__(LG_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
void large_adj_force(const LargeArgs A, const int second) {
    // dynamic LDS: [waves][LG_CAP] neighbour buffers, then [3][LG_TILE] position tiles for the all-atom scan (absent in cell mode)
    extern __shared__ __attribute__((aligned(16))) float4 nbuf[];
    float* tile = reinterpret_cast<float*>(nbuf + (blockDim.x >> 6) * LG_CAP);
    __shared__ float red[16 * LG_NV];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, rep = blockIdx.y + A.rep0, i_fr = A.step;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const size_t so = (size_t)rep * N * 3;
    const float* qs = second ? A.qm + so : A.q_t + ((size_t)rep * T + i_fr) * N * 3;
    const float* vs = second ? A.vm + so : A.v_t + ((size_t)rep * T + i_fr) * N * 3;
    const float* lam = second ? A.lvh + so : A.lv + so;
    TermConst tc[MDG_MAX_TERMS];
    const float rc2max = prepare_terms(A, tc);
    const int i = blockIdx.x * (blockDim.x >> 6) + wid;
    const bool valid = i < N;
    float fx, fy, fz, gx, gy, gz, th[LG_KMAX];
#pragma unroll
    for (int p = 0; p < LG_KMAX; ++p) th[p] = 0.f;
    // (table kind: the parameter term of an interval comes from the midpoint evaluation with weight h, :160)
    // (NVE: from the first evaluation, sovlers.py:82,101 -- both with total weight h)
    const bool tab_eval = (A.prm.ensemble == 0) == (second != 0);
    const float gw = (tab_eval && A.g64) ? 0.5f * (A.t[i_fr] - A.t[i_fr - 1]) * A.terms.t[0].c : 0.f;
    if constexpr (STALE)
        wave_stale_force<DIAG, 2>(A, qs, lam, i, valid, tile, nbuf + wid * LG_CAP, fx, fy, fz, gx, gy, gz, th, tc, rep);
    else
    wave_neighbours_and_force<DIAG, 2, KIND, 0>(A, qs, lam, i, valid, tile, nbuf + wid * LG_CAP, fx, fy, fz, gx, gy, gz, th,
                                                tc, rc2max, gw, rep);
    float vals[LG_NV];
#pragma unroll
    for (int p = 0; p < LG_KMAX; ++p) vals[p] = th[p];
    float p1 = 0.f, p2 = 0.f;
    if (valid && lane < 3) {
        const int e = 3 * i + lane;
        A.f[so + e] = lane == 0 ? fx : (lane == 1 ? fy : fz);
        A.dq[so + e] = lane == 0 ? gx : (lane == 1 ? gy : gz);
        const float m = A.mass[i], pp = vs[e] * m;
        p1 = pp * pp / m;
        p2 = lam[e] * vs[e];
    }
    vals[LG_KMAX] = p1; vals[LG_KMAX + 1] = p2;
    block_sum_n<LG_NV>(vals, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int p = 0; p < LG_NV; ++p) A.partN[((size_t)rep * A.nbF + blockIdx.x) * LG_NV + p] = vals[p];
    }
}

// ---------------------------------------------------------------------------------------------
// The adjoint's evaluation over the STORED candidates, four atoms per wave: a DPP row (16 lanes) per atom, lane s of
// the row takes candidates s, s + 16, ...  A liquid's ~60 candidates fill 4 passes of 16 lanes where a whole wave per
// atom ran one full and one nearly empty pass of 64; the per-atom sums stay inside the row (4 DPP adds each, no LDS
// crossbar), and the per-wave fixed work (term constants, partial sums) is shared by four atoms.  The six index words
// of a lane are read up front, the next pass's positions / adjoint directions are requested before the current pass
// is evaluated.  Lanes beyond an atom's count gather the atom itself: D = 0 is skipped like everywhere (topology.py:67).
// The candidates are the stored indices of the build that serves frame A.step; a build that overflowed, or a midpoint
// that moved past the skin (flagged by large_prep<3>), raises flags[5]: the caller repeats the adjoint with searches
// (MdgTrajParams.block = -1).
// Launch: 256 threads = 16 LG_ADJ_GROUPS atoms per workgroup, grid (ceil(N / (16 LG_ADJ_GROUPS)), R) = A.nbF rows of partN.
template <bool DIAG, int KIND>
__global__ __launch_bounds__(256) void large_adj_listed(const LargeArgs A, const int second) {
    constexpr int NP = LG_LIST / 16;                                   // passes a full row takes
    constexpr int NTH = KIND >= 0 ? MDG_MAX_THETA : LG_KMAX;           // live parameter partials
    __shared__ float red[4 * (NTH + 2)];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, rep = blockIdx.y + A.rep0, i_fr = A.step;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, s = lane & 15;
    const size_t so = (size_t)rep * N * 3;
    const float* q = second ? A.qm + so : A.q_t + ((size_t)rep * T + i_fr) * N * 3;
    const float* vs = second ? A.vm + so : A.v_t + ((size_t)rep * T + i_fr) * N * 3;
    const float* lam = second ? A.lvh + so : A.lv + so;
    TermConst tc[MDG_MAX_TERMS];
    if (KIND >= 0) tc[0] = term_prepare(A.terms.t[0], A.theta);
    else prepare_terms(A, tc);
    const int ntl = KIND >= 0 ? 1 : A.terms.n_terms;
    const bool nhc = A.prm.ensemble == 0;
    const bool tab_eval = nhc == (second != 0);                        // (as large_adj_force)
    const float gw = (KIND < 0 && tab_eval && A.g64) ? 0.5f * (A.t[i_fr] - A.t[i_fr - 1]) * A.terms.t[0].c : 0.f;
    // first evaluation: the build that serves frame i_fr (valid there by the forward pass's construction).  Midpoint:
    // that one or frame i_fr + 1's, whichever large_prep<3> found within skin / 2 of the midpoint positions
    // (nl_state[2 rep + {0, 1}] = {first, second} candidate is too far; cleared by the first evaluation's launch).
    int slot = A.nl_build[(size_t)rep * T + i_fr];
    bool bad = A.nl_bad[(size_t)rep * T + slot] != 0;                   // (rows of such a build were not all written)
    if (second) {
        const int slotB = i_fr + 1 < T ? A.nl_build[(size_t)rep * T + i_fr + 1] : slot;
        const bool okA = !bad && !A.nl_state[2 * rep];
        const bool okB = !A.nl_bad[(size_t)rep * T + slotB] && !A.nl_state[2 * rep + 1];
        if (okB) slot = slotB;
        bad = !(okA || okB);
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.nl_state[2 * rep] = 0; A.nl_state[2 * rep + 1] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && bad) A.flags[5] = 1;
    const float* wl = A.wl + so;                                       // w = lam_v / m (NVE: lam_v), written by large_prep<2|3>
    // A wave takes LG_ADJ_GROUPS consecutive groups of four atoms: the launch-constant part of a wave (term constants, list
    // selection) and the wave- and workgroup-level reductions of the parameter / kinetic partial sums -- half of the
    // instructions of a wave that handled ONE group -- are paid once per 16 atoms
    float vals[NTH + 2];
#pragma unroll
    for (int p = 0; p < NTH + 2; ++p) vals[p] = 0.f;
#pragma unroll 1
    for (int grp = 0; grp < LG_ADJ_GROUPS; ++grp) {
    const int i = ((blockIdx.x * 4 + wid) * LG_ADJ_GROUPS + grp) * 4 + (lane >> 4);
    if (__ballot(i < N) == 0ull) break;                                // (wave-uniform: this wave's groups are through)
    const bool valid = i < N;
    const int ic = valid ? i : N - 1;                                  // (rows past the end read the last atom, count 0)
    const size_t at = ((size_t)rep * T + slot) * N + ic;
    // (entry s + 16 p of the row: two 16-bit entries per word, lanes 2t and 2t + 1 read the same word)
    const uint32_t* idx = reinterpret_cast<const uint32_t*>(A.nl_idx + at * LG_LIST);
    int jj[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) jj[p] = (int)((idx[(s >> 1) + 8 * p] >> (16 * (s & 1))) & 0xffffu);
    const int n = (valid && !bad) ? min(A.nl_cnt[at], LG_LIST) : 0;
    const Row3 qi = row3(q, ic), li = row3(wl, ic);
    const float xi = qi.x, yi = qi.y, zi = qi.z;
    const float mi = A.mass[ic];
    const float wxi = li.x, wyi = li.y, wzi = li.z;
    float fx = 0.f, fy = 0.f, fz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f, th[LG_KMAX];
#pragma unroll
    for (int p = 0; p < LG_KMAX; ++p) th[p] = 0.f;
    if constexpr (KIND == KIND_LJ126 && DIAG) {
        // LJ 12-6 in an orthorhombic cell (BASELINE config #4): TWO candidates per lane and iteration -- entries s + 16 p and
        // s + 16 (p + 1) of the row -- as the halves of packed fp32 registers (v_pk_fma / mul / add_f32), the arithmetic of
        // the wave-per-replica kernels (csrc/traj_small.hip force_lj126_packed, traj_ring.hpp): even powers of 1/r from one
        // v_rcp_f32 per pair, no square root, branch-free (the 1/d^2 of a rejected candidate is selected to 0, so every
        // term below is exactly 0 for it), the parameter gradients carried as the two sums they are linear in.  The scalar
        // loop it replaces issued ~180 lane-instructions per accepted pair with the VALU 99 % busy
        // (profiles/pmc_lj4096.json, round 3).
        const TermConst& t0 = tc[0];
        const float sig2 = t0.k0 * t0.k0, e4 = 4.f * t0.k1, cq = t0.c, rc2 = t0.rc2;
        const float m1a = 6.f * e4 * cq, m1b = 12.f * e4;                    // phi'/r  = (m1a s6 - m1b s12) / d2
        const float ka = 48.f * e4 * cq, kb = 168.f * e4;                    // phi'' - phi'/r = (kb s12 - ka s6) / d2
        const float ivx = A.cell.inv[0], ivy = A.cell.inv[4], ivz = A.cell.inv[8];
        const float hx = A.cell.h[0], hy = A.cell.h[4], hz = A.cell.h[8];
        f32x2 fx2 = {0.f, 0.f}, fy2 = fx2, fz2 = fx2, gx2 = fx2, gy2 = fx2, gz2 = fx2, S6 = fx2, S12 = fx2;
        // ALL rows of the first NPF passes (96 candidates: a liquid's rows hold ~78) are requested up front -- one round trip
        // per group of atoms instead of one per pass -- and the rare longer rows take a second batch.  (What bounds the
        // sweep after the packing is neither the arithmetic nor that latency: ~48 fully divergent 12-byte gathers per wave
        // at ~64 cache lines each keep the CU's address / L1 path busy for the kernel's whole duration -- packing, up-front
        // requests and a version software-pipelined across the groups all measured 66 us +- 1 at 64 replicas with the
        // VALU 63-65 % busy, where the scalar loop measured 68 us at 99 %; profiles/pmc_lj4096.json.)
        constexpr int NPF = 6;
        auto pair2 = [&](const Row3 qA, const Row3 qB, const Row3 lA, const Row3 lB) {
            f32x2 dx = f32x2{qA.x, qB.x} - xi, dy = f32x2{qA.y, qB.y} - yi, dz = f32x2{qA.z, qB.z} - zi;   // D = x_j - x_i
            const f32x2 ax = wxi - f32x2{lA.x, lB.x}, ay = wyi - f32x2{lA.y, lB.y}, az = wzi - f32x2{lA.z, lB.z};
            dx = min_image_diag2(dx, ivx, hx); dy = min_image_diag2(dy, ivy, hy); dz = min_image_diag2(dz, ivz, hz);
            const f32x2 d2 = norm2_ref2(dx, dy, dz);
            const bool ok0 = (d2.x != 0.f) && (d2.x < rc2);             // (idle lanes gathered the atom itself: D = 0)
            const bool ok1 = (d2.y != 0.f) && (d2.y < rc2);             // topology.py:67
            const f32x2 i2 = {ok0 ? __builtin_amdgcn_rcpf(d2.x) : 0.f, ok1 ? __builtin_amdgcn_rcpf(d2.y) : 0.f};
            const f32x2 s2 = sig2 * i2;
            const f32x2 s6 = s2 * s2 * s2;
            const f32x2 s12 = s6 * s6;
            const f32x2 c1 = (m1a * s6 - m1b * s12) * i2;
            fx2 += c1 * dx; fy2 += c1 * dy; fz2 += c1 * dz;           // F_i += (phi'/r) D
            const f32x2 b = dx * ax + dy * ay + dz * az;
            const f32x2 bi = b * i2;                                    // (w_ij . D) / d2
            const f32x2 k2 = (kb * s12 - ka * s6) * (bi * i2);          // (phi'' - phi'/r) (w_ij . D) / d2
            gx2 += k2 * dx; gx2 += c1 * ax;                             // (opposite sign: negated once below)
            gy2 += k2 * dy; gy2 += c1 * ay;
            gz2 += k2 * dz; gz2 += c1 * az;
            S6 += s6 * bi; S12 += s12 * bi;
        };
        {
            Row3 qv[NPF], lv_[NPF];
#pragma unroll
            for (int p = 0; p < NPF; ++p) {
                const int j = s + 16 * p < n ? jj[p] : ic;
                qv[p] = row3(q, j); lv_[p] = row3(wl, j);
            }
#pragma unroll
            for (int p = 0; p < NPF; p += 2) {
                if (p > 0 && __ballot(s + 16 * p < n) == 0) break;    // (wave-uniform: every row is through)
                pair2(qv[p], qv[p + 1], lv_[p], lv_[p + 1]);
            }
        }
        if (__ballot(s + 16 * NPF < n) != 0) {                         // rows beyond 96 candidates (dense / hot spots)
            Row3 qv[NP - NPF], lv_[NP - NPF];
#pragma unroll
            for (int p = NPF; p < NP; ++p) {
                const int j = s + 16 * p < n ? jj[p] : ic;
                qv[p - NPF] = row3(q, j); lv_[p - NPF] = row3(wl, j);
            }
#pragma unroll
            for (int p = 0; p < NP - NPF; p += 2) pair2(qv[p], qv[p + 1], lv_[p], lv_[p + 1]);
        }
        fx = fx2.x + fx2.y; fy = fy2.x + fy2.y; fz = fz2.x + fz2.y;
        gx = -(gx2.x + gx2.y); gy = -(gy2.x + gy2.y); gz = -(gz2.x + gz2.y);
        const float a6 = S6.x + S6.y, a12 = S12.x + S12.y;
        th[0] = e4 * t0.k2 * (18.f * cq * a6 - 72.f * a12);             // 1/2 d(w.F)/dsigma, this atom's end of its pairs
        th[1] = 12.f * cq * a6 - 24.f * a12;                            // ... d/depsilon
    } else {
    // software pipeline over the passes: (position, direction) of pass p + 1 in flight while pass p is evaluated
    int jn = s < n ? jj[0] : ic;
    Row3 qn = row3(q, jn), ln = row3(wl, jn);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p > 0 && __ballot(s + 16 * p < n) == 0) break;            // (wave-uniform: every row is through)
        const int j = jn;
        float dx = qn.x - xi, dy = qn.y - yi, dz = qn.z - zi;           // D = x_j - x_i
        const float ax = ln.x, ay = ln.y, az = ln.z;
        if (p + 1 < NP) {
            jn = s + 16 * (p + 1) < n ? jj[p + 1] : ic;
            qn = row3(q, jn); ln = row3(wl, jn);
        }
        min_image<DIAG>(A.cell, dx, dy, dz);
        const float d2 = norm2_ref(dx, dy, dz);
        if (d2 == 0.f) continue;                                        // the atom itself (idle lanes) -- topology.py:67
        pair_terms<2, KIND>(A, tc, ntl, N, i, j, dx, dy, dz, d2, wxi, wyi, wzi, ax, ay, az, gw, rep, fx, fy, fz, gx, gy, gz, th);
    }
    }
    fx = row16_sum(fx); fy = row16_sum(fy); fz = row16_sum(fz);
    gx = row16_sum(gx); gy = row16_sum(gy); gz = row16_sum(gz);
#pragma unroll
    for (int p = 0; p < NTH; ++p) vals[p] += th[p];
    if (valid && s < 3) {
        const int e = 3 * i + s;
        A.f[so + e] = s == 0 ? fx : (s == 1 ? fy : fz);
        A.dq[so + e] = s == 0 ? gx : (s == 1 ? gy : gz);
        const float pp = vs[e] * mi;
        vals[NTH] += pp * pp / mi;
        vals[NTH + 1] += lam[e] * vs[e];
    }
    }                                                                  // groups of this wave
#pragma unroll
    for (int p = 0; p < NTH + 2; ++p) vals[p] = wave_sum_rows(vals[p]);
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < NTH + 2; ++p) red[wid * (NTH + 2) + p] = vals[p];
    }
    __syncthreads();
    if (threadIdx.x < LG_NV) {
        const int p = threadIdx.x;
        const int c = p < LG_KMAX ? (p < NTH ? p : -1) : NTH + (p - LG_KMAX);
        float t_ = 0.f;
        if (c >= 0) t_ = (red[c] + red[(NTH + 2) + c]) + (red[2 * (NTH + 2) + c] + red[3 * (NTH + 2) + c]);
        A.partN[((size_t)rep * A.nbF + blockIdx.x) * LG_NV + p] = t_;
    }
}

// ---------------------------------------------------------------------------------------------
// COLUMN-TILE versions of the two listed launches (see LargeArgs): a workgroup = one (bx, by) column of bins of the
// serving build = a contiguous range of that build's sorted order; the 3 x 3 columns around it staged in LDS once.
struct LTile {
    int cbase[10];                           // first staged slot of stencil column c; [9] = staged atoms
    int gstart[9];                           // ... its first slot in the build's sorted order
    int ok;
};

// header + staging by the whole workgroup: the rows of `q` (and `w`, unless nullptr) of the atoms at the tile's sorted slots,
// gathered through the build's permutation.  Ends with a barrier.
__device__ __forceinline__ void large_stage(const LargeArgs& A, const int32_t* __restrict__ bst, const int32_t* __restrict__ perm,
                                            int tile, const float* __restrict__ q, const float* __restrict__ w, LTile& M,
                                            Row3* tp, Row3* tw, int32_t* tidx) {
    const int nbx = A.nb[0], nby = A.nb[1];
    const int cx = tile / nby, cy = tile - cx * nby;
    if (threadIdx.x < 9) {
        const int c = threadIdx.x;
        int x = cx + c / 3 - 1, y = cy + c % 3 - 1;
        x = x < 0 ? x + nbx : (x >= nbx ? x - nbx : x);
        y = y < 0 ? y + nby : (y >= nby ? y - nby : y);
        const int col = x * nby + y;
        M.gstart[c] = bst[col];
        M.cbase[c + 1] = bst[col + 1] - bst[col];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < 9; ++c) { const int len = M.cbase[c + 1]; M.cbase[c] = run; run += len; }
        M.cbase[9] = run;
        M.ok = run <= A.tile_cap;
    }
    __syncthreads();
    const int total = M.ok ? M.cbase[9] : 0;
    for (int t = threadIdx.x; t < total; t += blockDim.x) {
        int c = 0;
#pragma unroll
        for (int k = 1; k < 9; ++k) c += t >= M.cbase[k];
        const int a = perm[M.gstart[c] + (t - M.cbase[c])];
        tp[t] = row3(q, a);
        tidx[t] = a;
        if (w) tw[t] = row3(w, a);
    }
    __syncthreads();
}

// Second half of step k over the CURRENT list, column tiles.  grid (R, ncol): consecutive workgroups = consecutive
// replicas, so the tiles of a replica run on one XCD (its L2 holds the replica's state rows and candidate rows).
template <int KIND>
__global__ __launch_bounds__(LG_TILE_THREADS) void large_fwd_tiled(const LargeArgs A) {
    constexpr int NP = LG_LIST / 16;
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    __shared__ LTile M;
    __shared__ float red[32];
    __shared__ float Qs[MDG_MAX_CHAINS], pvs[MDG_MAX_CHAINS];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, C = A.prm.n_chains, rep = blockIdx.x + A.rep0, tile = blockIdx.y, k = A.step;
    if (A.nl_state[2 * rep]) return;                                   // this step searched (large_search_rows<1>)
    const int slot = A.nl_state[2 * rep + 1];
    Row3* tp = reinterpret_cast<Row3*>(lds_f);
    int32_t* tidx = reinterpret_cast<int32_t*>(lds_f + 3 * (size_t)A.tile_cap);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, s = lane & 15;
    const size_t so = (size_t)rep * N * 3;
    float* q = A.q + so; float* v = A.v + so; float* vh = A.vh + so; float* f = A.f + so;
    float* pv = A.pv + rep * MDG_MAX_CHAINS; float* ph = A.ph + rep * MDG_MAX_CHAINS;
    float* pvh = A.pvh + rep * MDG_MAX_CHAINS;
    const bool nhc = A.prm.ensemble == 0;
    const float dt = A.t[k + 1] - A.t[k];
    if (tile == 0) {
        if (threadIdx.x == 0) A.nl_build[(size_t)rep * T + k + 1] = slot;
        if (nhc) {                                                      // finish the bath with KE(v + vh), as large_force_step<1>
            if (threadIdx.x < MDG_MAX_CHAINS) {
                float qv = 0.f;
#pragma unroll
                for (int c = 0; c < MDG_MAX_CHAINS; ++c) if (threadIdx.x == c) qv = A.prm.Q[c];
                Qs[threadIdx.x] = qv;
            }
            const float ke = 0.5f * reduce_partials(A.partB + (size_t)rep * A.nbE, A.nbE, 1, 0, red);
            if (threadIdx.x < C) pvs[threadIdx.x] = pvh[threadIdx.x];
            __syncthreads();
            if (threadIdx.x < C) {
                const float b1 = bath_rhs_l(A.prm, Qs, pvs, ke, threadIdx.x);
                const float np = pv[threadIdx.x] + (ph[threadIdx.x] + 0.5f * b1 * dt);
                pv[threadIdx.x] = np;
                A.pv_t[((size_t)rep * T + k + 1) * C + threadIdx.x] = np;
            }
            __syncthreads();
        }
    }
    large_stage(A, A.nl_bst + ((size_t)rep * T + slot) * (LG_MAX_COLS + 1), A.nl_perm + ((size_t)rep * T + slot) * N, tile, q, nullptr, M,
                tp, nullptr, tidx);
    TermConst tc[MDG_MAX_TERMS];
    if (KIND >= 0) tc[0] = term_prepare(A.terms.t[0], A.theta);
    else prepare_terms(A, tc);
    const int ntl = KIND >= 0 ? 1 : A.terms.n_terms;
    const int own0 = M.cbase[4], nown = M.ok ? M.cbase[5] - M.cbase[4] : 0, g0 = M.gstart[4];
    float kepart = 0.f;
#pragma unroll 1
    for (int r0 = wid * 4; r0 < nown; r0 += nw * 4) {
        const int r = r0 + (lane >> 4);
        const bool valid = r < nown;
        const int rc = valid ? r : nown - 1;
        const int ls = own0 + rc;                                      // the row atom's staged slot
        const size_t at = ((size_t)rep * T + slot) * N + g0 + rc;
        const uint32_t* idx = reinterpret_cast<const uint32_t*>(A.nl_idx + at * LG_LIST);
        int jj[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) jj[p] = (int)((idx[(s >> 1) + 8 * p] >> (16 * (s & 1))) & 0xffffu);
        const int n = valid ? min(A.nl_cnt[at], LG_LIST) : 0;
        const Row3 qi = tp[ls];
        const int i = tidx[ls];
        float fx = 0.f, fy = 0.f, fz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f, th[LG_KMAX];
        if constexpr (KIND == KIND_LJ126) {
            const TermConst& t0 = tc[0];
            const float sig2 = t0.k0 * t0.k0, e4 = 4.f * t0.k1, cq = t0.c, rc2 = t0.rc2;
            const float m1a = 6.f * e4 * cq, m1b = 12.f * e4;
            const float ivx = A.cell.inv[0], ivy = A.cell.inv[4], ivz = A.cell.inv[8];
            const float hx = A.cell.h[0], hy = A.cell.h[4], hz = A.cell.h[8];
            f32x2 fx2 = {0.f, 0.f}, fy2 = fx2, fz2 = fx2;
#pragma unroll
            for (int p = 0; p < NP; p += 2) {
                if (p > 0 && __ballot(s + 16 * p < n) == 0) break;    // (wave-uniform: every row is through)
                const Row3 qA = tp[s + 16 * p < n ? jj[p] : ls], qB = tp[s + 16 * (p + 1) < n ? jj[p + 1] : ls];
                f32x2 dx = f32x2{qA.x, qB.x} - qi.x, dy = f32x2{qA.y, qB.y} - qi.y, dz = f32x2{qA.z, qB.z} - qi.z;
                dx = min_image_diag2(dx, ivx, hx); dy = min_image_diag2(dy, ivy, hy); dz = min_image_diag2(dz, ivz, hz);
                const f32x2 d2 = norm2_ref2(dx, dy, dz);
                const bool ok0 = (d2.x != 0.f) && (d2.x < rc2), ok1 = (d2.y != 0.f) && (d2.y < rc2);
                const f32x2 i2 = {ok0 ? __builtin_amdgcn_rcpf(d2.x) : 0.f, ok1 ? __builtin_amdgcn_rcpf(d2.y) : 0.f};
                const f32x2 s2 = sig2 * i2;
                const f32x2 s6 = s2 * s2 * s2;
                const f32x2 c1 = (m1a * s6 - m1b * (s6 * s6)) * i2;
                fx2 += c1 * dx; fy2 += c1 * dy; fz2 += c1 * dz;
            }
            fx = fx2.x + fx2.y; fy = fy2.x + fy2.y; fz = fz2.x + fz2.y;
        } else {
#pragma unroll 1
            for (int p = 0; p < NP; ++p) {
                if (p > 0 && __ballot(s + 16 * p < n) == 0) break;
                if (!(s + 16 * p < n)) continue;
                const int l = jj[p];
                const Row3 qn = tp[l];
                float dx = qn.x - qi.x, dy = qn.y - qi.y, dz = qn.z - qi.z;
                min_image<true>(A.cell, dx, dy, dz);
                const float d2 = norm2_ref(dx, dy, dz);
                if (d2 == 0.f) continue;
                pair_terms<1, KIND>(A, tc, ntl, N, i, tidx[l], dx, dy, dz, d2, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, rep, fx, fy, fz, gx, gy,
                                    gz, th);
            }
        }
        fx = row16_sum(fx); fy = row16_sum(fy); fz = row16_sum(fz);
        if (valid && s < 3) {
            const float F = s == 0 ? fx : (s == 1 ? fy : fz);
            const int e = 3 * i + s;
            const float m = A.mass[i];
            const float vv = v[e] + vh[e];
            const float p = vv * m;
            const float a = nhc ? (F - pvh[0] * p / A.prm.Q[0]) / m : F;          // (NVE: md.py:145-148, no 1/m)
            const float vn = v[e] + (vh[e] + 0.5f * a * dt);
            v[e] = vn;
            f[e] = F;
            const size_t fr = ((size_t)rep * T + k + 1) * N * 3 + e;
            A.q_t[fr] = q[e];
            A.v_t[fr] = vn;
            const float pn = vn * m;
            kepart += pn * pn / m;
            if (!(isfinite(vn) && isfinite(F))) A.flags[1] = 1;
        }
    }
    kepart = wave_sum_rows(kepart);
    if (lane == 0) red[wid] = kepart;
    __syncthreads();
    // partA holds A.nbL rows for the listed launches (large_search_rows writes one per 16 atoms): this tile's sum goes to row
    // `tile`, and the tiles clear the rows beyond ncol between them
    if (threadIdx.x == 0) {
        float t_ = 0.f;
        for (int w = 0; w < nw; ++w) t_ += red[w];
        A.partA[(size_t)rep * A.nbF + tile] = t_;
        for (int row = tile + A.ncol; row < A.nbL; row += A.ncol) A.partA[(size_t)rep * A.nbF + row] = 0.f;
    }
}

// The adjoint's evaluation over the stored candidates, column tiles: (position, w) staged per tile, every candidate two
// ds_read_b96.  Build selection and the flags as large_adj_listed.  grid (R, ncol), partN rows = ncol per replica.
template <int KIND>
__global__ __launch_bounds__(LG_TILE_THREADS) void large_adj_tiled(const LargeArgs A, const int second) {
    constexpr int NP = LG_LIST / 16;
    constexpr int NTH = KIND >= 0 ? MDG_MAX_THETA : LG_KMAX;
    constexpr int NWV = LG_TILE_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) float lds_a[];
    __shared__ LTile M;
    __shared__ float red[NWV * (NTH + 2)];
    const int N = A.prm.n_atoms, T = A.prm.n_frames, rep = blockIdx.x + A.rep0, tile = blockIdx.y, i_fr = A.step;
    Row3* tp = reinterpret_cast<Row3*>(lds_a);
    Row3* tw = reinterpret_cast<Row3*>(lds_a + 3 * (size_t)A.tile_cap);
    int32_t* tidx = reinterpret_cast<int32_t*>(lds_a + 6 * (size_t)A.tile_cap);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6, s = lane & 15;
    const size_t so = (size_t)rep * N * 3;
    const float* vs = second ? A.vm + so : A.v_t + ((size_t)rep * T + i_fr) * N * 3;
    const float* lam = second ? A.lvh + so : A.lv + so;
    TermConst tc[MDG_MAX_TERMS];
    if (KIND >= 0) tc[0] = term_prepare(A.terms.t[0], A.theta);
    else prepare_terms(A, tc);
    const int ntl = KIND >= 0 ? 1 : A.terms.n_terms;
    const bool nhc = A.prm.ensemble == 0;
    const bool tab_eval = nhc == (second != 0);                        // (as large_adj_force)
    const float gw = (KIND < 0 && tab_eval && A.g64) ? 0.5f * (A.t[i_fr] - A.t[i_fr - 1]) * A.terms.t[0].c : 0.f;
    int slot = A.nl_build[(size_t)rep * T + i_fr];
    bool bad = A.nl_bad[(size_t)rep * T + slot] != 0;
    if (second) {
        const int slotB = i_fr + 1 < T ? A.nl_build[(size_t)rep * T + i_fr + 1] : slot;
        const bool okA = !bad && !A.nl_state[2 * rep];
        const bool okB = !A.nl_bad[(size_t)rep * T + slotB] && !A.nl_state[2 * rep + 1];
        if (okB) slot = slotB;
        bad = !(okA || okB);
    }
    const float* q = second ? A.qm + so : A.q_t + ((size_t)rep * T + i_fr) * N * 3;
    large_stage(A, A.nl_bst + ((size_t)rep * T + slot) * (LG_MAX_COLS + 1), A.nl_perm + ((size_t)rep * T + slot) * N, tile, q, A.wl + so, M,
                tp, tw, tidx);
    // (the first evaluation clears the midpoint flags: only after every workgroup of the previous midpoint launch -- an
    //  earlier kernel of the stream -- has read them)
    if (!second && tile == 0 && threadIdx.x == 0) { A.nl_state[2 * rep] = 0; A.nl_state[2 * rep + 1] = 0; }
    if (tile == 0 && threadIdx.x == 0 && (bad || !M.ok)) A.flags[5] = 1;
    const int own0 = M.cbase[4], nown = (M.ok && !bad) ? M.cbase[5] - M.cbase[4] : 0, g0 = M.gstart[4];
    float vals[NTH + 2];
#pragma unroll
    for (int p = 0; p < NTH + 2; ++p) vals[p] = 0.f;
#pragma unroll 1
    for (int r0 = wid * 4; r0 < nown; r0 += nw * 4) {
        const int r = r0 + (lane >> 4);
        const bool valid = r < nown;
        const int rc = valid ? r : nown - 1;
        const int ls = own0 + rc;
        const size_t at = ((size_t)rep * T + slot) * N + g0 + rc;
        const uint32_t* idx = reinterpret_cast<const uint32_t*>(A.nl_idx + at * LG_LIST);
        int jj[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) jj[p] = (int)((idx[(s >> 1) + 8 * p] >> (16 * (s & 1))) & 0xffffu);
        const int n = valid ? min(A.nl_cnt[at], LG_LIST) : 0;
        const Row3 qi = tp[ls], li = tw[ls];
        const int i = tidx[ls];
        const float xi = qi.x, yi = qi.y, zi = qi.z, wxi = li.x, wyi = li.y, wzi = li.z;
        float fx = 0.f, fy = 0.f, fz = 0.f, gx = 0.f, gy = 0.f, gz = 0.f, th[LG_KMAX];
#pragma unroll
        for (int p = 0; p < LG_KMAX; ++p) th[p] = 0.f;
        if constexpr (KIND == KIND_LJ126) {
            // the packed arithmetic of large_adj_listed: two candidates per lane and iteration, even powers of 1/r from one
            // v_rcp_f32 per pair, branch-free, the parameter gradients as the two sums they are linear in
            const TermConst& t0 = tc[0];
            const float sig2 = t0.k0 * t0.k0, e4 = 4.f * t0.k1, cq = t0.c, rc2 = t0.rc2;
            const float m1a = 6.f * e4 * cq, m1b = 12.f * e4;                    // phi'/r  = (m1a s6 - m1b s12) / d2
            const float ka = 48.f * e4 * cq, kb = 168.f * e4;                    // phi'' - phi'/r = (kb s12 - ka s6) / d2
            const float ivx = A.cell.inv[0], ivy = A.cell.inv[4], ivz = A.cell.inv[8];
            const float hx = A.cell.h[0], hy = A.cell.h[4], hz = A.cell.h[8];
            f32x2 fx2 = {0.f, 0.f}, fy2 = fx2, fz2 = fx2, gx2 = fx2, gy2 = fx2, gz2 = fx2, S6 = fx2, S12 = fx2;
#pragma unroll
            for (int p = 0; p < NP; p += 2) {
                if (p > 0 && __ballot(s + 16 * p < n) == 0) break;    // (wave-uniform: every row is through)
                const int lA = s + 16 * p < n ? jj[p] : ls, lB = s + 16 * (p + 1) < n ? jj[p + 1] : ls;
                const Row3 qA = tp[lA], qB = tp[lB], wA = tw[lA], wB = tw[lB];
                f32x2 dx = f32x2{qA.x, qB.x} - xi, dy = f32x2{qA.y, qB.y} - yi, dz = f32x2{qA.z, qB.z} - zi;   // D = x_j - x_i
                const f32x2 ax = wxi - f32x2{wA.x, wB.x}, ay = wyi - f32x2{wA.y, wB.y}, az = wzi - f32x2{wA.z, wB.z};
                dx = min_image_diag2(dx, ivx, hx); dy = min_image_diag2(dy, ivy, hy); dz = min_image_diag2(dz, ivz, hz);
                const f32x2 d2 = norm2_ref2(dx, dy, dz);
                const bool ok0 = (d2.x != 0.f) && (d2.x < rc2);             // (idle lanes read the atom itself: D = 0)
                const bool ok1 = (d2.y != 0.f) && (d2.y < rc2);             // topology.py:67
                const f32x2 i2 = {ok0 ? __builtin_amdgcn_rcpf(d2.x) : 0.f, ok1 ? __builtin_amdgcn_rcpf(d2.y) : 0.f};
                const f32x2 s2 = sig2 * i2;
                const f32x2 s6 = s2 * s2 * s2;
                const f32x2 s12 = s6 * s6;
                const f32x2 c1 = (m1a * s6 - m1b * s12) * i2;
                fx2 += c1 * dx; fy2 += c1 * dy; fz2 += c1 * dz;           // F_i += (phi'/r) D
                const f32x2 b = dx * ax + dy * ay + dz * az;
                const f32x2 bi = b * i2;                                    // (w_ij . D) / d2
                const f32x2 k2 = (kb * s12 - ka * s6) * (bi * i2);          // (phi'' - phi'/r) (w_ij . D) / d2
                gx2 += k2 * dx; gx2 += c1 * ax;                             // (opposite sign: negated once below)
                gy2 += k2 * dy; gy2 += c1 * ay;
                gz2 += k2 * dz; gz2 += c1 * az;
                S6 += s6 * bi; S12 += s12 * bi;
            }
            fx = fx2.x + fx2.y; fy = fy2.x + fy2.y; fz = fz2.x + fz2.y;
            gx = -(gx2.x + gx2.y); gy = -(gy2.x + gy2.y); gz = -(gz2.x + gz2.y);
            const float a6 = S6.x + S6.y, a12 = S12.x + S12.y;
            th[0] = e4 * t0.k2 * (18.f * cq * a6 - 72.f * a12);             // 1/2 d(w.F)/dsigma, this atom's end of its pairs
            th[1] = 12.f * cq * a6 - 24.f * a12;                            // ... d/depsilon
        } else {
#pragma unroll 1
            for (int p = 0; p < NP; ++p) {
                if (p > 0 && __ballot(s + 16 * p < n) == 0) break;
                if (!(s + 16 * p < n)) continue;
                const int l = jj[p];
                const Row3 qn = tp[l], ln = tw[l];
                float dx = qn.x - xi, dy = qn.y - yi, dz = qn.z - zi;       // D = x_j - x_i
                min_image<true>(A.cell, dx, dy, dz);
                const float d2 = norm2_ref(dx, dy, dz);
                if (d2 == 0.f) continue;                                    // topology.py:67
                pair_terms<2, KIND>(A, tc, ntl, N, i, tidx[l], dx, dy, dz, d2, wxi, wyi, wzi, ln.x, ln.y, ln.z, gw, rep, fx, fy, fz,
                                    gx, gy, gz, th);
            }
        }
        fx = row16_sum(fx); fy = row16_sum(fy); fz = row16_sum(fz);
        gx = row16_sum(gx); gy = row16_sum(gy); gz = row16_sum(gz);
#pragma unroll
        for (int p = 0; p < NTH; ++p) vals[p] += th[p];
        if (valid && s < 3) {
            const int e = 3 * i + s;
            A.f[so + e] = s == 0 ? fx : (s == 1 ? fy : fz);
            A.dq[so + e] = s == 0 ? gx : (s == 1 ? gy : gz);
            const float mi = A.mass[i];
            const float pp = vs[e] * mi;
            vals[NTH] += pp * pp / mi;
            vals[NTH + 1] += lam[e] * vs[e];
        }
    }
#pragma unroll
    for (int p = 0; p < NTH + 2; ++p) vals[p] = wave_sum_rows(vals[p]);
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < NTH + 2; ++p) red[wid * (NTH + 2) + p] = vals[p];
    }
    __syncthreads();
    if (threadIdx.x < LG_NV) {
        const int p = threadIdx.x;
        const int c = p < LG_KMAX ? (p < NTH ? p : -1) : NTH + (p - LG_KMAX);
        float t_ = 0.f;
        if (c >= 0)
            for (int w = 0; w < nw; ++w) t_ += red[w * (NTH + 2) + c];
        A.partN[((size_t)rep * A.nbF + tile) * LG_NV + p] = t_;
    }
}

// fixed point -> float table gradient
__global__ void large_table_grad(const unsigned long long* __restrict__ g64, size_t n, float scale, float* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = (float)((double)(long long)g64[k] / (double)scale);
}

// lam(T-1) = dL/dy_{T-1} of every replica: the last frame's rows of the incoming gradients (zeros where none came)
__global__ void large_adj_init(const float* __restrict__ g_v, const float* __restrict__ g_q, const float* __restrict__ g_pv, int N,
                               int T, int C, float* __restrict__ lv, float* __restrict__ lq, float* __restrict__ lp) {
    const int r = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t src = ((size_t)r * T + T - 1) * N * 3, dst = (size_t)r * N * 3;
    if (e < 3 * N) {
        lv[dst + e] = g_v ? g_v[src + e] : 0.f;
        lq[dst + e] = g_q ? g_q[src + e] : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x < MDG_MAX_CHAINS)
        lp[r * MDG_MAX_CHAINS + threadIdx.x] = (g_pv && (int)threadIdx.x < C) ? g_pv[((size_t)r * T + T - 1) * C + threadIdx.x] : 0.f;
}

struct WsLayout {
    size_t q, v, vh, f, lv, lq, lvh, lqh, dq, qm, vm, wl, pv, ph, pvh, lp, lph, pvm, partA, partB, partN, gth, ghi, glo, flags,
        spos, bstart, binslot, nl_idx, nl_cnt, nl_bad, nl_build, nl_state, nl_perm, nl_bst, total;
    bool keep_lists;
};

WsLayout ws_layout(int R, int N, int nb, int KT, int T) {
    WsLayout w{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) / 64 * 64; return r; };
    const size_t s3 = (size_t)R * N * 3, sc = (size_t)R * MDG_MAX_CHAINS;
    w.q = take(s3); w.v = take(s3); w.vh = take(s3); w.f = take(s3);
    w.lv = take(s3); w.lq = take(s3); w.lvh = take(s3); w.lqh = take(s3); w.dq = take(s3); w.qm = take(s3); w.vm = take(s3); w.wl = take(s3);
    w.pv = take(sc); w.ph = take(sc); w.pvh = take(sc); w.lp = take(sc); w.lph = take(sc); w.pvm = take(sc);
    const int nbmax = nb > (3 * N + 255) / 256 ? nb : (3 * N + 255) / 256;
    w.partA = take((size_t)R * nbmax); w.partB = take((size_t)R * nbmax);
    w.partN = take((size_t)R * nbmax * LG_NV);
    w.gth = take((size_t)R * (KT > 0 ? KT : 1));
    w.ghi = take((size_t)2 * R * (KT > 0 ? KT : 1));    // (int64 words of the table kind -- offsets are multiples of 256 B --; a few words otherwise)
    w.glo = w.ghi;
    w.flags = take(16);
    w.spos = take((size_t)R * N * 4);
    w.bstart = take((size_t)R * (LG_MAX_CELLS + 1));
    w.binslot = take((size_t)R * N);
    // neighbour lists of every frame, kept for the adjoint (when they fit the budget)
    const long long lw = (long long)R * T * N * (LG_LIST / 2 + 2) + (long long)R * T * (LG_MAX_COLS + 3) + 2ll * R;
    w.keep_lists = T > 1 && lw <= LG_LIST_MAX_WORDS;
    if (w.keep_lists) {
        w.nl_idx = take((size_t)R * T * N * (LG_LIST / 2)); w.nl_cnt = take((size_t)R * T * N);
        w.nl_bad = take((size_t)R * T); w.nl_build = take((size_t)R * T); w.nl_state = take((size_t)2 * R);
        // column tiles: per build the permutation (sorted slot -> atom) and the bin columns' first slots
        w.nl_perm = take((size_t)R * T * N); w.nl_bst = take((size_t)R * T * (LG_MAX_COLS + 1));
    }
    w.total = o;
    return w;
}

// REPLICA GROUPS ON CONCURRENT STREAMS (round 5).  Between two force launches a trajectory runs a `prep` launch that is pure
// latency: one or four workgroups per replica, a chain of dependent loads, 17-20 us during which most of the chip idles
// (valu_busy 0.08-0.2; 14 % of the 64 x 4 096-atom pass), and every force launch ends in a tail of half-empty CUs.  Replicas
// never interact, so the launches of a trajectory are issued for two halves of the replicas on two side streams (forked from /
// joined into the caller's stream by events): one half's prep and tails overlap the other half's force sweep.  The kernels and
// every number they produce are unchanged -- a replica's launches run in the same order on the same data.
constexpr int LG_MAX_GROUPS = 4;
struct LgStreams {
    int device = -1;
    hipStream_t s[LG_MAX_GROUPS] = {};
    hipEvent_t fork = nullptr, join[LG_MAX_GROUPS] = {};
};

// number of groups for R replicas on stream `st` (MDG_LARGE_STREAMS = 1..4 overrides; 1 while `st` is being captured)
int lg_group_count(int R, hipStream_t st) {
    // (64 x 4 096 atoms, MI355X: 170.3 k steps/s on one stream, 193.4 k with two groups, 195.6 k with three -- about one round
    //  of workgroups per launch --, 173.8 k with four: launches too small to fill the chip)
    int want = R >= 48 ? 3 : (R >= 8 ? 2 : 1);
    if (const char* e = getenv("MDG_LARGE_STREAMS")) want = atoi(e);
    if (want < 1) want = 1;
    if (want > LG_MAX_GROUPS) want = LG_MAX_GROUPS;
    if (want > R) want = R;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) want = 1;
    return want;
}

LgStreams* lg_streams() {
    static thread_local LgStreams pool[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    LgStreams& P = pool[dev];
    if (P.device != dev) {
        for (int g = 0; g < LG_MAX_GROUPS; ++g) {
            if (hipStreamCreateWithFlags(&P.s[g], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&P.join[g], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&P.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        P.device = dev;
    }
    return &P;
}

// MDG_LARGE_TILES=0: the listed launches gather by atom index from L2 (A/B measurements; the pre-round-5 kernels)
bool large_tiles_enabled() {
    const char* e = getenv("MDG_LARGE_TILES");
    return !(e && e[0] == '0');
}

int validate_large(const MdgTrajParams* p, const MdgCell* cell, const MdgTerms* terms) {
    MDG_CHECK_ARG(p && cell && terms, "traj_large: null descriptor");
    MDG_CHECK_ARG(p->n_rep > 0 && p->n_atoms > 1 && p->n_frames >= 1, "traj_large: bad sizes");
    MDG_CHECK_ARG(p->n_atoms <= LG_PREP * LG_PREP_ATOMS_MAX, "traj_large: at most %d atoms", LG_PREP * LG_PREP_ATOMS_MAX);
    MDG_CHECK_ARG(p->ensemble == 0 || p->ensemble == 1, "traj_large: ensemble must be 0 (NHC) or 1 (NVE)");
    MDG_CHECK_ARG(p->ensemble == 1 || (p->n_chains >= 2 && p->n_chains <= MDG_MAX_CHAINS),
                  "traj_large: 2 <= num_chains <= %d", MDG_MAX_CHAINS);
    MDG_CHECK_ARG(terms->n_terms >= 1 && terms->n_terms <= MDG_MAX_TERMS, "traj_large: 1..%d pair terms", MDG_MAX_TERMS);
    for (int m = 0; m < terms->n_terms; ++m) {
        const MdgPairTerm& t = terms->t[m];
        if (t.kind != MDG_PAIR_TABLE) continue;
        MDG_CHECK_ARG(terms->n_terms == 1 && !t.mask, "traj_large: a tabulated pair model must be the only term, unmasked");
        MDG_CHECK_ARG(t.p >= 4 && t.p <= 4096 && t.n_theta == 2 * t.p && t.phi > 0.f && t.c > 0.f,
                      "traj_large: bad table (nodes %d, n_theta %d, du %g, scale %g)", t.p, t.n_theta, t.phi, t.c);
    }
    return MDG_OK;
}

}  // namespace

extern "C" int64_t mdg_traj_large_workspace(int n_rep, int n_atoms, int n_frames, int n_theta_total) {
    if (n_rep <= 0 || n_atoms <= 0 || n_frames <= 0) return -1;
    const int nb = (n_atoms + LG_WAVES_CELL - 1) / LG_WAVES_CELL;       // (the larger of the two workgroup shapes)
    return (int64_t)ws_layout(n_rep, n_atoms, nb, n_theta_total, n_frames).total;
}

extern "C" int mdg_traj_large_list_builds(const float* ws, int n_rep, int n_atoms, int n_frames, int n_theta_total,
                                          int32_t* build_of_frame, void* stream) {
    MDG_CHECK_ARG(ws && build_of_frame && n_rep > 0 && n_atoms > 0 && n_frames > 0, "traj_large_list_builds: bad arguments");
    const WsLayout L = ws_layout(n_rep, n_atoms, (n_atoms + LG_WAVES_CELL - 1) / LG_WAVES_CELL, n_theta_total, n_frames);
    if (!L.keep_lists) return 1;
    MDG_HIP(hipMemcpyAsync(build_of_frame, ws + L.nl_build, sizeof(int32_t) * (size_t)n_rep * n_frames,
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MDG_OK;
}

// (phases that bin need the replica in ONE workgroup; the adjoint over stored lists does not: one atom per thread)
#define LG_PREP_LAUNCH(PH_)                                                                          \
    do {                                                                                             \
        if ((PH_) >= 2 && a.nl_idx)                                                                  \
            hipLaunchKernelGGL((large_prep<PH_, 1>), dim3((N + LG_PREP_SMALL - 1) / LG_PREP_SMALL, Rg), dim3(LG_PREP_SMALL), 0, sg, a); \
        else if (N <= 4 * LG_PREP) hipLaunchKernelGGL((large_prep<PH_, 4>), dim3(1, Rg), dim3(LG_PREP), 0, sg, a); \
        else if (N <= LG_PREP_ATOMS * LG_PREP) hipLaunchKernelGGL((large_prep<PH_, LG_PREP_ATOMS>), dim3(1, Rg), dim3(LG_PREP), 0, sg, a); \
        else hipLaunchKernelGGL((large_prep<PH_, LG_PREP_ATOMS_MAX>), dim3(1, Rg), dim3(LG_PREP), 0, sg, a); \
    } while (0)

// the replica groups of a trajectory: G, their streams (the caller's when G == 1), fork / join around the launch loops
#define LG_GROUPS_BEGIN()                                                                            \
    int G = lg_group_count(R, st);                                                                   \
    LgStreams* LS = G > 1 ? lg_streams() : nullptr;                                                  \
    if (!LS) G = 1;                                                                                  \
    if (G > 1) {                                                                                     \
        MDG_HIP(hipEventRecord(LS->fork, st));                                                       \
        for (int g = 0; g < G; ++g) MDG_HIP(hipStreamWaitEvent(LS->s[g], LS->fork, 0));              \
    }                                                                                                \
    (void)0
#define LG_GROUP(g_)                                                                                 \
    const int r0_ = (int)((long long)R * (g_) / G), Rg = (int)((long long)R * ((g_) + 1) / G) - r0_; \
    hipStream_t sg = G > 1 ? LS->s[g_] : st;                                                         \
    a.rep0 = r0_;                                                                                    \
    (void)0
#define LG_GROUPS_END()                                                                              \
    if (G > 1) {                                                                                     \
        for (int g = 0; g < G; ++g) {                                                                \
            MDG_HIP(hipEventRecord(LS->join[g], LS->s[g]));                                          \
            MDG_HIP(hipStreamWaitEvent(st, LS->join[g], 0));                                         \
        }                                                                                            \
    }                                                                                                \
    a.rep0 = 0;                                                                                      \
    (void)0

#define LG_SETUP()                                                                                   \
    const int R = prm->n_rep, N = prm->n_atoms;                                                      \
    const int nbE = 1;              /* the element-wise work of a replica runs in one workgroup (large_prep) */ \
    const WsLayout L = ws_layout(R, N, (N + LG_WAVES_CELL - 1) / LG_WAVES_CELL, terms->n_theta_total, prm->n_frames); \
    LargeArgs a{};                                                                                   \
    a.prm = *prm; a.cell = *cell; a.terms = *terms; a.theta = theta; a.mass = mass; a.t = t_grid;   \
    a.q = ws + L.q; a.v = ws + L.v; a.vh = ws + L.vh; a.f = ws + L.f;                                \
    a.lv = ws + L.lv; a.lq = ws + L.lq; a.lvh = ws + L.lvh; a.lqh = ws + L.lqh; a.dq = ws + L.dq;    \
    a.qm = ws + L.qm; a.vm = ws + L.vm; a.wl = ws + L.wl;                                            \
    a.pv = ws + L.pv; a.ph = ws + L.ph; a.pvh = ws + L.pvh; a.lp = ws + L.lp; a.lph = ws + L.lph;    \
    a.pvm = ws + L.pvm; a.partA = ws + L.partA; a.partB = ws + L.partB; a.partN = ws + L.partN;      \
    a.gth = ws + L.gth; a.flags = flags; a.nbE = nbE;                                                \
    const bool table = terms->t[0].kind == MDG_PAIR_TABLE;                                           \
    a.g64 = table ? reinterpret_cast<unsigned long long*>(ws + L.ghi) : nullptr;                     \
    a.glim = fx64_limit((double)(prm->n_frames > 1 ? prm->n_frames - 1 : 1) * (double)prm->n_atoms * (double)LG_CAP); \
    hipStream_t st = (hipStream_t)stream;                                                            \
    const bool diag = cell->diag != 0;                                                               \
    a.spos = reinterpret_cast<float4*>(ws + L.spos);                                                 \
    a.bstart = reinterpret_cast<int32_t*>(ws + L.bstart);                                            \
    a.binslot = reinterpret_cast<int32_t*>(ws + L.binslot);                                          \
    a.ncell = 0;                                                                                     \
    if (L.keep_lists && prm->block != -1) {          /* (block = -1: search at every evaluation) */       \
        a.nl_idx = reinterpret_cast<uint16_t*>(ws + L.nl_idx); a.nl_cnt = reinterpret_cast<int32_t*>(ws + L.nl_cnt); \
        a.nl_bad = reinterpret_cast<int32_t*>(ws + L.nl_bad);                                        \
        a.nl_build = reinterpret_cast<int32_t*>(ws + L.nl_build);                                    \
        a.nl_state = reinterpret_cast<int32_t*>(ws + L.nl_state);                                    \
        float rcm = 0.f;                                                                             \
        for (int m = 0; m < terms->n_terms; ++m) rcm = terms->t[m].cutoff > rcm ? terms->t[m].cutoff : rcm; \
        a.skin = LG_SKIN * rcm;                                                                      \
    }                                                                                                \
    if (diag) {                                                                                      \
        float rcmax = 0.f;                                                                           \
        for (int m = 0; m < terms->n_terms; ++m) rcmax = terms->t[m].cutoff > rcmax ? terms->t[m].cutoff : rcmax; \
        int nbx[3];                                                                                  \
        bool ok = rcmax > 0.f;                                                                       \
        for (int d = 0; d < 3 && ok; ++d) { nbx[d] = (int)floorf(cell->h[4 * d] / (rcmax + a.skin)); ok = nbx[d] >= 3; } \
        if (ok && (long long)nbx[0] * nbx[1] * nbx[2] <= LG_MAX_CELLS) {                             \
            a.nb[0] = nbx[0]; a.nb[1] = nbx[1]; a.nb[2] = nbx[2]; a.ncell = nbx[0] * nbx[1] * nbx[2]; \
        }                                                                                            \
    }                                                                                                \
    if (a.ncell && a.nl_idx) {                                                                       \
        /* column tiles: staged capacity = 1.35 x the nine columns of a uniform box (+ slack); a box whose tiles cannot */ \
        /* be staged keeps the L2-gather launches */                                                 \
        a.ncol = a.nb[0] * a.nb[1];                                                                  \
        long long cap = (long long)(9.0 * 1.35 * N / (double)a.ncol) + 64;                           \
        if (cap > N) cap = N;                                                                        \
        cap = (cap + 63) / 64 * 64;                                                                  \
        if (large_tiles_enabled() && cap <= LG_TILE_MAX && a.ncol <= LG_MAX_COLS && a.ncol <= (N + LG_WAVES_CELL - 1) / LG_WAVES_CELL) { \
            a.tile_cap = (int)cap;                                                                   \
            a.nl_perm = reinterpret_cast<int32_t*>(ws + L.nl_perm);                                  \
            a.nl_bst = reinterpret_cast<int32_t*>(ws + L.nl_bst);                                    \
        }          /* (otherwise the rows hold atom indices and the listed launches gather from L2, as in unbinned boxes) */ \
    }                                                                                                \
    const int wpb = a.ncell ? LG_WAVES_CELL : LG_WAVES;                                              \
    const int nbF = (N + wpb - 1) / wpb;                                                             \
    a.nbF = nbF;                                                                                     \
    const dim3 gL((N + LG_ROW_ATOMS - 1) / LG_ROW_ATOMS, R);       /* the listed kernels: 16 atoms per workgroup */ \
    a.nbL = (int)gL.x > a.ncol || !a.tile_cap ? (int)gL.x : a.ncol;   /* partA rows of the listed / searching launches */ \
    const size_t tile_lds = sizeof(float4) * (size_t)wpb * LG_CAP + (a.ncell ? 0 : sizeof(float) * 3 * LG_TILE); \
    const bool lj126 = terms->n_terms == 1 && diag && !terms->t[0].mask && terms->t[0].kind == MDG_PAIR_LJ && \
                       terms->t[0].p == 12 && (terms->t[0].q == 6 || terms->t[0].c == 0.f);          \
    (void)0;

// stale lists (topology_update_freq > 1): frequency, the integrator's call count at the launch's first call, the persistent rows
struct StaleOpt { int freq; long long count0; uint32_t* rows; };
static inline bool stale_due(const StaleOpt* so, long long e) { return (so->count0 + e) % (long long)so->freq == 0; }
static int validate_stale(const MdgTrajParams* prm, const MdgTerms* terms, const StaleOpt* so) {
    MDG_CHECK_ARG(so->freq >= 1 && so->count0 >= 0 && so->rows, "traj_large_stale: bad frequency / counter / list buffer");
    MDG_CHECK_ARG(prm->n_atoms <= 32768, "traj_large_stale: at most 32 768 atoms (15-bit row entries)");
    for (int m = 0; m < terms->n_terms; ++m)
        MDG_CHECK_ARG(terms->t[m].kind != MDG_PAIR_TABLE, "traj_large_stale: a tabulated pair model is not supported");
    return MDG_OK;
}
#define LG_STALE_SETUP()                                                                             \
    if (so) {                                                                                        \
        a.st_row = so->rows;                                                                         \
        a.st_cnt = reinterpret_cast<int32_t*>(so->rows + (size_t)R * N * LG_CAP);                    \
    }                                                                                                \
    (void)0

static int traj_fwd_large_run(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                              const float* theta, const float* mass, const float* t_grid,
                              const float* v0, const float* q0, const float* pv0,
                              float* v_t, float* q_t, float* pv_t, float* ws, int32_t* flags, void* stream,
                              const StaleOpt* so) {
    int rc = validate_large(prm, cell, terms);
    if (rc) return rc;
    MDG_CHECK_ARG(mass && t_grid && v0 && q0 && v_t && q_t && ws && flags, "traj_fwd_large: null buffer");
    MDG_CHECK_ARG(prm->ensemble == 1 || (pv0 && pv_t), "traj_fwd_large: NHC needs pv0/pv_t");
    MdgTrajParams pstale;
    if (so) {                                   // (stale lists: no candidate lists of the Verlet-reuse kind -- block = -1)
        rc = validate_stale(prm, terms, so);
        if (rc) return rc;
        pstale = *prm; pstale.block = -1; prm = &pstale;
    }
    LG_SETUP();
    LG_STALE_SETUP();
    a.v_t = v_t; a.q_t = q_t; a.pv_t = pv_t;
    const int C = prm->n_chains, T = prm->n_frames;
    MDG_HIP(hipMemcpyAsync(a.q, q0, sizeof(float) * (size_t)R * N * 3, hipMemcpyDeviceToDevice, st));
    MDG_HIP(hipMemcpyAsync(a.v, v0, sizeof(float) * (size_t)R * N * 3, hipMemcpyDeviceToDevice, st));
    if (prm->ensemble == 0)
        MDG_HIP(hipMemcpy2DAsync(a.pv, sizeof(float) * MDG_MAX_CHAINS, pv0, sizeof(float) * C, sizeof(float) * C, R,
                                 hipMemcpyDeviceToDevice, st));
    a.step = 0;
    if (a.nl_idx) MDG_HIP(hipMemsetAsync(a.nl_bad, 0, sizeof(int32_t) * (size_t)R * T, st));
#define LG_FORCE_STEP(MODE_)                                                                                    \
    do {                                                                                                        \
        const dim3 gF(nbF, Rg), gLg(gL.x, Rg);                                                                  \
        if (a.nl_idx && a.ncell) {         /* binned box, lists kept: the row-based search (one sweep) */      \
            if (lj126) hipLaunchKernelGGL((large_search_rows<MODE_, KIND_LJ126>), gLg, dim3(256), 0, sg, a);  \
            else hipLaunchKernelGGL((large_search_rows<MODE_, -1>), gLg, dim3(256), 0, sg, a);                 \
        } else                                                                                                  \
        if (lj126) hipLaunchKernelGGL((large_force_step<true, MODE_, KIND_LJ126>), gF, dim3(64 * wpb), tile_lds, sg, a); \
        else if (diag) hipLaunchKernelGGL((large_force_step<true, MODE_, -1>), gF, dim3(64 * wpb), tile_lds, sg, a);    \
        else hipLaunchKernelGGL((large_force_step<false, MODE_, -1>), gF, dim3(64 * wpb), tile_lds, sg, a);            \
    } while (0)
    // stale lists: the first right-hand-side call of step k has the running index 2 k, the second 2 k + 1 (sovlers.py:110-127);
    // calls 2 k + 1 and 2 k + 2 share their positions, so the force of the former serves the latter unless that one rebuilds
#define LG_STALE_STEP(MODE_, REBUILD_)                                                                          \
    do {                                                                                                        \
        const dim3 gF(nbF, Rg);                                                                                 \
        a.st_rebuild = (REBUILD_) ? 1 : 0;                                                                      \
        if (diag) hipLaunchKernelGGL((large_force_step<true, MODE_, -1, true>), gF, dim3(64 * wpb), tile_lds, sg, a);  \
        else hipLaunchKernelGGL((large_force_step<false, MODE_, -1, true>), gF, dim3(64 * wpb), tile_lds, sg, a);      \
    } while (0)
    LG_GROUPS_BEGIN();
    if (so) {
        for (int g = 0; g < G; ++g) {
            LG_GROUP(g);
            a.st_bin = stale_due(so, 0);
            if (a.ncell && a.st_bin) LG_PREP_LAUNCH(0);
            LG_STALE_STEP(0, stale_due(so, 0));
        }
        for (int k = 0; k + 1 < T; ++k) {
            a.step = k;
            const bool again = k + 2 < T && stale_due(so, 2ll * k + 2) && !stale_due(so, 2ll * k + 1);
            for (int g = 0; g < G; ++g) {
                LG_GROUP(g);
                a.st_bin = stale_due(so, 2ll * k + 1) || again;     // (the bins serve both calls at these positions)
                LG_PREP_LAUNCH(1);
                LG_STALE_STEP(1, stale_due(so, 2ll * k + 1));
                if (again) LG_STALE_STEP(2, true);
            }
        }
    } else {
    for (int g = 0; g < G; ++g) {
        LG_GROUP(g);
        if (a.ncell) LG_PREP_LAUNCH(0);
        LG_FORCE_STEP(0);
    }
    for (int k = 0; k + 1 < T; ++k) {
        a.step = k;
        for (int g = 0; g < G; ++g) {
            LG_GROUP(g);
            LG_PREP_LAUNCH(1);                                  // kick + drift + bath half step; search needed? then binning
            LG_FORCE_STEP(1);                                   // (returns at once while the current list serves)
            if (a.tile_cap) {                                   // (returns at once when the step searched)
                const dim3 gT(Rg, a.ncol);
                const size_t lds = sizeof(float) * 4 * (size_t)a.tile_cap;
                if (lj126) hipLaunchKernelGGL((large_fwd_tiled<KIND_LJ126>), gT, dim3(LG_TILE_THREADS), lds, sg, a);
                else hipLaunchKernelGGL((large_fwd_tiled<-1>), gT, dim3(LG_TILE_THREADS), lds, sg, a);
            } else if (a.nl_idx) {
                const dim3 gLF((N + LG_ROW_ATOMS * LG_FWD_GROUPS - 1) / (LG_ROW_ATOMS * LG_FWD_GROUPS), Rg);
                if (lj126) hipLaunchKernelGGL((large_fwd_listed<true, KIND_LJ126>), gLF, dim3(256), 0, sg, a);
                else if (diag) hipLaunchKernelGGL((large_fwd_listed<true, -1>), gLF, dim3(256), 0, sg, a);
                else hipLaunchKernelGGL((large_fwd_listed<false, -1>), gLF, dim3(256), 0, sg, a);
            }
        }
    }
    }
    LG_GROUPS_END();
#undef LG_FORCE_STEP
#undef LG_STALE_STEP
    MDG_CHECK_LAUNCH("traj_fwd_large");
    return MDG_OK;
}

extern "C" int mdg_traj_fwd_large(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                  const float* theta, const float* mass, const float* t_grid,
                                  const float* v0, const float* q0, const float* pv0,
                                  float* v_t, float* q_t, float* pv_t, float* ws, int32_t* flags, void* stream) {
    return traj_fwd_large_run(prm, cell, terms, theta, mass, t_grid, v0, q0, pv0, v_t, q_t, pv_t, ws, flags, stream, nullptr);
}

extern "C" int64_t mdg_traj_large_stale_words(int n_rep, int n_atoms) {
    if (n_rep <= 0 || n_atoms <= 0) return -1;
    return (int64_t)n_rep * n_atoms * (LG_CAP + 1);
}

extern "C" int mdg_traj_fwd_large_stale(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                        const float* theta, const float* mass, const float* t_grid,
                                        const float* v0, const float* q0, const float* pv0,
                                        float* v_t, float* q_t, float* pv_t, float* ws, int32_t* flags,
                                        int freq, int64_t count0, uint32_t* rows, void* stream) {
    const StaleOpt so{freq, (long long)count0, rows};
    return traj_fwd_large_run(prm, cell, terms, theta, mass, t_grid, v0, q0, pv0, v_t, q_t, pv_t, ws, flags, stream, &so);
}

static int traj_adj_large_run(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                              const float* theta, const float* mass, const float* t_grid,
                              const float* v_t, const float* q_t, const float* pv_t,
                              const float* g_v, const float* g_q, const float* g_pv,
                              float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                              float* ws, int32_t* flags, void* stream, const StaleOpt* so) {
    int rc = validate_large(prm, cell, terms);
    if (rc) return rc;
    MDG_CHECK_ARG(mass && t_grid && v_t && q_t && adj_v0 && adj_q0 && ws && flags, "traj_adj_large: null buffer");
    MDG_CHECK_ARG(prm->ensemble == 1 || (pv_t && adj_pv0), "traj_adj_large: NHC needs pv_t/adj_pv0");
    MdgTrajParams pstale;
    if (so) {
        rc = validate_stale(prm, terms, so);
        if (rc) return rc;
        pstale = *prm; pstale.block = -1; prm = &pstale;
    }
    LG_SETUP();
    LG_STALE_SETUP();
    a.v_t = const_cast<float*>(v_t); a.q_t = const_cast<float*>(q_t); a.pv_t = const_cast<float*>(pv_t);
    a.g_v = g_v; a.g_q = g_q; a.g_pv = g_pv;
    const int C = prm->n_chains, T = prm->n_frames, KT = terms->n_theta_total;
    const size_t fr = sizeof(float) * (size_t)N * 3;
    // lam = dL/dy_{T-1}: ONE launch for all replicas (three copies per replica in a host loop were 192 serialised 3-5 us copies
    // per 64-replica adjoint: 0.5-1 ms of a 20 ms pass)
    (void)fr;
    hipLaunchKernelGGL(large_adj_init, dim3((3 * N + 255) / 256, R), dim3(256), 0, st, g_v, g_q, g_pv, N, T, C, a.lv, a.lq, a.lp);
    MDG_HIP(hipMemsetAsync(a.gth, 0, sizeof(float) * (size_t)R * (KT > 0 ? KT : 1), st));
    if (table) {
        MDG_HIP(hipMemsetAsync(a.g64, 0, sizeof(unsigned long long) * (size_t)R * KT, st));
    }
    const int gLAx = (N + LG_ROW_ATOMS * LG_ADJ_GROUPS - 1) / (LG_ROW_ATOMS * LG_ADJ_GROUPS);
    if (a.nl_idx) a.nbF = a.tile_cap ? a.ncol : gLAx;         // (rows of partN the listed launches write, the prep launches sum)
#define LG_ADJ_FORCE(SECOND_)                                                                                       \
    do {                                                                                                            \
        const dim3 gF(nbF, Rg), gLA(gLAx, Rg);                                                                      \
        if (a.tile_cap) {                                                                                           \
            const size_t lds_ = sizeof(float) * 7 * (size_t)a.tile_cap;                                             \
            if (lj126) hipLaunchKernelGGL((large_adj_tiled<KIND_LJ126>), dim3(Rg, a.ncol), dim3(LG_TILE_THREADS), lds_, sg, a, SECOND_); \
            else hipLaunchKernelGGL((large_adj_tiled<-1>), dim3(Rg, a.ncol), dim3(LG_TILE_THREADS), lds_, sg, a, SECOND_); \
        } else if (a.nl_idx) {                                                                                      \
            if (lj126) hipLaunchKernelGGL((large_adj_listed<true, KIND_LJ126>), gLA, dim3(256), 0, sg, a, SECOND_);      \
            else if (diag) hipLaunchKernelGGL((large_adj_listed<true, -1>), gLA, dim3(256), 0, sg, a, SECOND_);           \
            else hipLaunchKernelGGL((large_adj_listed<false, -1>), gLA, dim3(256), 0, sg, a, SECOND_);                    \
        } else if (lj126) hipLaunchKernelGGL((large_adj_force<true, KIND_LJ126>), gF, dim3(64 * wpb), tile_lds, sg, a, SECOND_); \
        else if (diag) hipLaunchKernelGGL((large_adj_force<true, -1>), gF, dim3(64 * wpb), tile_lds, sg, a, SECOND_);    \
        else hipLaunchKernelGGL((large_adj_force<false, -1>), gF, dim3(64 * wpb), tile_lds, sg, a, SECOND_);            \
    } while (0)
#define LG_STALE_ADJ(SECOND_, REBUILD_)                                                                             \
    do {                                                                                                            \
        const dim3 gF(nbF, Rg);                                                                                     \
        a.st_rebuild = (REBUILD_) ? 1 : 0;                                                                          \
        if (diag) hipLaunchKernelGGL((large_adj_force<true, -1, true>), gF, dim3(64 * wpb), tile_lds, sg, a, SECOND_);  \
        else hipLaunchKernelGGL((large_adj_force<false, -1, true>), gF, dim3(64 * wpb), tile_lds, sg, a, SECOND_);      \
    } while (0)
    LG_GROUPS_BEGIN();
    for (int i = T - 1; i >= 1; --i) {
        a.step = i;
        for (int g = 0; g < G; ++g) {
            LG_GROUP(g);
            if (so) a.st_bin = stale_due(so, 3ll * (T - 1 - i)) || stale_due(so, 3ll * (T - 1 - i) + 1);
            LG_PREP_LAUNCH(2);                                                      // finish interval i + 1, bin frame i
            if (so) {
                // an interval makes three calls (sovlers.py:258-266): the dL/dt evaluation at y_i (its result is not used,
                // but it advances the counter and may rebuild), the first augmented evaluation at the same positions, the
                // midpoint evaluation -- running counts c0, c0 + 1, c0 + 2
                const long long c0 = 3ll * (T - 1 - i);
                LG_STALE_ADJ(0, stale_due(so, c0) || stale_due(so, c0 + 1));
                a.st_bin = stale_due(so, c0 + 2);
                LG_PREP_LAUNCH(3);
                LG_STALE_ADJ(1, stale_due(so, c0 + 2));
                continue;
            }
            LG_ADJ_FORCE(0);
            LG_PREP_LAUNCH(3);                                                      // midpoint state, bin it
            LG_ADJ_FORCE(1);
        }
    }
#undef LG_ADJ_FORCE
#undef LG_STALE_ADJ
    a.step = 0;
    for (int g = 0; g < G; ++g) {
        LG_GROUP(g);
        LG_PREP_LAUNCH(4);                                                          // finish interval 1
    }
    LG_GROUPS_END();
    MDG_HIP(hipMemcpyAsync(adj_v0, a.lv, sizeof(float) * (size_t)R * N * 3, hipMemcpyDeviceToDevice, st));
    MDG_HIP(hipMemcpyAsync(adj_q0, a.lq, sizeof(float) * (size_t)R * N * 3, hipMemcpyDeviceToDevice, st));
    if (prm->ensemble == 0)
        MDG_HIP(hipMemcpy2DAsync(adj_pv0, sizeof(float) * C, a.lp, sizeof(float) * MDG_MAX_CHAINS, sizeof(float) * C, R,
                                 hipMemcpyDeviceToDevice, st));
    if (adj_theta && table) {
        const size_t n = (size_t)R * KT;
        hipLaunchKernelGGL(large_table_grad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.g64, n, terms->t[0].c, adj_theta);
    } else if (adj_theta && KT > 0)
        MDG_HIP(hipMemcpyAsync(adj_theta, a.gth, sizeof(float) * (size_t)R * KT, hipMemcpyDeviceToDevice, st));
    MDG_CHECK_LAUNCH("traj_adj_large");
    return MDG_OK;
}

extern "C" int mdg_traj_adj_large(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                  const float* theta, const float* mass, const float* t_grid,
                                  const float* v_t, const float* q_t, const float* pv_t,
                                  const float* g_v, const float* g_q, const float* g_pv,
                                  float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                                  float* ws, int32_t* flags, void* stream) {
    return traj_adj_large_run(prm, cell, terms, theta, mass, t_grid, v_t, q_t, pv_t, g_v, g_q, g_pv, adj_v0, adj_q0, adj_pv0,
                              adj_theta, ws, flags, stream, nullptr);
}

extern "C" int mdg_traj_adj_large_stale(const MdgTrajParams* prm, const MdgCell* cell, const MdgTerms* terms,
                                        const float* theta, const float* mass, const float* t_grid,
                                        const float* v_t, const float* q_t, const float* pv_t,
                                        const float* g_v, const float* g_q, const float* g_pv,
                                        float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                                        float* ws, int32_t* flags, int freq, int64_t count0, uint32_t* rows, void* stream) {
    const StaleOpt so{freq, (long long)count0, rows};
    return traj_adj_large_run(prm, cell, terms, theta, mass, t_grid, v_t, q_t, pv_t, g_v, g_q, g_pv, adj_v0, adj_q0, adj_pv0,
                              adj_theta, ws, flags, stream, &so);
}
