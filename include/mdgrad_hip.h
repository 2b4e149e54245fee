/*
 * mdgrad_hip.h -- C ABI of libmdgrad_hip.so: the MI355X (gfx950) hot path of the
 * differentiable-MD inner loop (force evaluation + integrator + adjoint + RDF + SchNet
 * message passing), hand-written HIP.  This header is the drop-in boundary: plain
 * pointers and sizes, no torch types.  All pointers are DEVICE pointers (HIP, fp32 /
 * int32 unless stated) except where marked `host`; `stream` is a hipStream_t passed as
 * void*.  Every entry point enqueues on `stream` and returns without synchronising;
 * the return value is 0 on success or a negative MDG_E* code (mdg_last_error() gives
 * the message).  Functions are stateless and re-entrant.
 *
 * The reference (torchmd/mdgrad, pure Python on PyTorch) has no FFI; each entry point
 * below names the reference op chain (file:line under /root/reference) it replaces.
 * INTEGRATION.md shows the ctypes binding the reference-side maintainer would add.
 */
#ifndef MDGRAD_HIP_H
#define MDGRAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDG_OK            0
#define MDG_EINVAL       -1   /* bad argument (size / null / unsupported combination) */
#define MDG_ELAUNCH      -2   /* HIP launch error                                     */
#define MDG_ECAPACITY    -3   /* (reported through device flags, see nbr builders)    */

#define MDG_MAX_TERMS     4
#define MDG_MAX_THETA     3   /* per term */
#define MDG_MAX_CHAINS    16

/* pair functional forms -- torchmd/potentials.py */
#define MDG_PAIR_LJ       0   /* 4 eps[(s/r)^p - c (s/r)^q]; theta=(sigma,eps). LennardJones :317-327,
                                 LJFamily :61-73, LennardJones69 :329-339, ExcludedVolume :341-352 (c=0) */
#define MDG_PAIR_MORSE    1   /* ModifiedMorse :75-93; constants a, phi; no parameters */
#define MDG_PAIR_BUCK     2   /* Buck :354-365; theta=(A,B,C) */
#define MDG_PAIR_YUKAWA   3   /* eps exp(-kappa r)/r; theta=(eps,kappa); NOT in the reference */
#define MDG_PAIR_TABLE    4   /* tabulated pair model (user modules such as pairMLP, torchmd/potentials.py:163-206,
                                 or a whole Stack of pair terms with one cutoff): theta = 2 p floats,
                                 theta[2g] = c1(u_g), theta[2g+1] = du * dc1/du(u_g) on the uniform grid
                                 u_g = a + g * phi in u = r^2, where c1(u) = phi'(r)/r; cubic-Hermite
                                 interpolation (force = c1 D; (phi'' - phi'/r)/r^2 = 2 dc1/du).  The adjoint
                                 returns dL/dtheta (the table gradient).  c = fixed-point scale of that
                                 accumulation (a power of two).  Fused small-system kernels only, single
                                 unmasked term, orthorhombic cell. */

typedef struct MdgPairTerm {
    int32_t kind;          /* MDG_PAIR_*                                                    */
    int32_t p, q;          /* LJ-family integer powers (q ignored when c == 0)              */
    float   c;             /* LJ-family attractive coefficient (1 = LJ, 0 = ExcludedVolume) */
    float   a, phi;        /* ModifiedMorse constants                                       */
    float   cutoff;        /* pair kept iff 0 < d^2 < cutoff^2 (topology.py:67)             */
    int32_t theta_off;     /* first parameter of this term inside theta[]                   */
    int32_t n_theta;
    int32_t reserved;
    const uint8_t* mask;   /* optional [N,N] 0/1 selection (index_tuple / ex_pairs,
                              topology.py:15-27,37-53); NULL = all pairs                    */
} MdgPairTerm;

typedef struct MdgTerms {
    int32_t n_terms;
    int32_t n_theta_total;
    MdgPairTerm t[MDG_MAX_TERMS];
} MdgTerms;

/* periodic cell: row vectors h[3][3] and its inverse (host side computes the inverse the
 * way the reference does, `cell.inverse()`, topology.py:59) */
typedef struct MdgCell {
    float h[9];
    float inv[9];
    int32_t diag;          /* 1 when h is diagonal (fast path) */
} MdgCell;

const char* mdg_last_error(void);
int mdg_version(void);

/* ------------------------------------------------------------------------------------
 * K1  neighbour list   (replaces generate_nbr_list, torchmd/topology.py:30-73)
 *
 * Output is a padded per-atom ("ELL") FULL list sorted by neighbour index: row i holds
 * cnt[i] <= max_nbr entries col[i*max_nbr + k] (ascending j, both directions present)
 * and shift[...] = (ox+1) + 3(oy+1) + 9(oz+1), the image flags of the reference's
 * `offsets` for the ordered pair (i,j) (d = x_i - x_j - o.cell).  The reference's
 * half list `nbr_list[P,2]` (i<j, lexicographic) and `offsets[P,3]` are the entries with
 * j > i in row order -- see mdg_nbr_half_from_ell.  *overflow is set to max over rows of
 * the needed count when some row needs more than max_nbr entries (caller re-allocates).
 *
 * mdg_nbr_build_dense: all-pairs minimum image, any (triclinic) cell, O(N^2).
 * mdg_nbr_build_cell : cell list (bin -> sort -> 27-stencil), orthorhombic cells with
 *                      at least 3 bins per side; identical output.
 * `scratch` for the cell build: int32[ 2*N + 2*ncell_max + 8 ] (see mdg_nbr_cell_scratch).
 */
int mdg_nbr_build_dense(const float* pos, int n_atoms, const MdgCell* cell /*host*/,
                        float cutoff, const uint8_t* mask,
                        int32_t* col, int32_t* shift, int32_t* cnt, int max_nbr,
                        int32_t* overflow, void* stream);
/* replica-batched variant: atoms form n_atoms/group independent systems of `group` consecutive
 * atoms sharing one cell; pairs never cross groups; mask (optional) is [group, group]. */
int mdg_nbr_build_dense_groups(const float* pos, int n_atoms, int group, const MdgCell* cell /*host*/,
                               float cutoff, const uint8_t* mask,
                               int32_t* col, int32_t* shift, int32_t* cnt, int max_nbr,
                               int32_t* overflow, void* stream);
int64_t mdg_nbr_cell_scratch(int n_atoms, const MdgCell* cell /*host*/, float cutoff);
int mdg_nbr_build_cell(const float* pos, int n_atoms, const MdgCell* cell /*host*/,
                       float cutoff, const uint8_t* mask,
                       int32_t* col, int32_t* shift, int32_t* cnt, int max_nbr,
                       int32_t* overflow, int32_t* scratch, void* stream);
/* cell list for replica-batched systems (see mdg_nbr_build_dense_groups): every group bins on its own */
int64_t mdg_nbr_cell_scratch_groups(int n_atoms, int group, const MdgCell* cell /*host*/, float cutoff);
int mdg_nbr_build_cell_groups(const float* pos, int n_atoms, int group, const MdgCell* cell /*host*/,
                              float cutoff, const uint8_t* mask,
                              int32_t* col, int32_t* shift, int32_t* cnt, int max_nbr,
                              int32_t* overflow, int32_t* scratch, void* stream);
/* half list in the reference's order: row_base = exclusive scan of per-row (j>i) counts.
 * nbr int64[P,2], offsets f32[P,3]; P must be the value returned in *n_pairs by
 * mdg_nbr_half_count (device int32).  edge_id (optional, int32[N*max_nbr]) receives for
 * every ELL slot the index of its undirected pair in the half list. */
int mdg_nbr_half_count(const int32_t* col, const int32_t* cnt, int n_atoms, int max_nbr,
                       int32_t* row_base /*[N+1]*/, void* stream);
int mdg_nbr_half_fill(const int32_t* col, const int32_t* shift, const int32_t* cnt,
                      const int32_t* row_base, int n_atoms, int max_nbr,
                      int64_t* nbr, float* offsets, int32_t* edge_id, void* stream);
/* Fixed-capacity variant for capture into HIP graphs (no host-side pair count): rows [P, capacity) hold
 * the sentinel pair (-1,-1) -- mdg_edge_diff / mdg_edge_prod return 0 for them -- and the image flag
 * (pad_offset,0,0); n_valid[0] <- min(P, capacity); need[0] <- max(need[0], P) when P > capacity. */
int mdg_nbr_half_fill_padded(const int32_t* col, const int32_t* shift, const int32_t* cnt,
                             const int32_t* row_base, int n_atoms, int max_nbr, int64_t capacity,
                             float pad_offset, int64_t* nbr, float* offsets, int32_t* edge_id,
                             int32_t* n_valid, int32_t* need, void* stream);

/* Verlet reuse of a fixed-capacity list (extension; the reference rebuilds at every call, torchmd/md.py:200-204).  One call per
 * force evaluation, no host synchronisation: the list is searched with list_cutoff = cutoff + skin and kept while every atom
 * stays within half_skin of where it was built -- decided on the device (state[0]); the builder launches return at their
 * first instruction otherwise, so a captured HIP graph replays the same nodes either way.  Consumers re-apply the exact
 * cutoff per pair with the builders' own arithmetic (mdg_edge_geom_masked, mdg_pair_eval_ell_into's recheck bit): the pair
 * set of every evaluation is the one a fresh search at `cutoff` finds.  state int32[4], zero-initialised once (state[3]
 * counts the builds); pos_build [N,3] initialised with NaN; need int32[2] as in the fixed-capacity builders; the remaining
 * buffers are those of mdg_nbr_build_*_groups / mdg_nbr_half_count / mdg_nbr_half_fill_padded, all persistent. */
int mdg_nbr_verlet_rebuild(const float* pos, int n_atoms, int group, const MdgCell* cell /*host*/, float list_cutoff,
                           float half_skin, const uint8_t* mask, int use_cell_list, int32_t* col, int32_t* shift, int32_t* cnt,
                           int max_nbr, int64_t capacity, float pad_offset, int64_t* nbr, float* offsets, int32_t* edge_id,
                           int32_t* n_valid, int32_t* need, float* pos_build, int32_t* state, int32_t* row_base,
                           int32_t* scratch, void* stream);

/* ------------------------------------------------------------------------------------
 * K2-K4  pair energy / gradient / Hessian-vector product over an ELL list
 * (replaces compute_dis + pair form + .sum() and both autograd passes through them:
 *  torchmd/topology.py:5-12, torchmd/interface.py:298-299, torchmd/md.py:227-228,
 *  torchmd/sovlers.py:229-233)
 *
 *   energy[0]      = sum_pairs phi(r)                    (if energy != NULL)
 *   grad[N,3]      = dU/dx                               (if grad   != NULL)  (F = -grad)
 *   gtheta[K]      = dU/dtheta                           (if gtheta != NULL, with grad)
 *   hw[N,3]        = H w                                 (if w != NULL)
 *   gtheta_w[K]    = d(w . dU/dx)/dtheta                 (if w != NULL)
 * One term per call (each term owns its list).  partial: f32 scratch of
 * mdg_pair_partial_size(n_atoms) floats.  Deterministic (no float atomics).
 */
int64_t mdg_pair_partial_size(int n_atoms);
int mdg_pair_eval_ell(const float* pos, int n_atoms, const MdgCell* cell /*host*/,
                      const int32_t* col, const int32_t* shift, const int32_t* cnt, int max_nbr,
                      const MdgPairTerm* term /*host*/, const float* theta,
                      const float* w,
                      float* energy, float* grad, float* gtheta,
                      float* hw, float* gtheta_w,
                      float* partial, void* stream);
/* the same with the per-atom outputs scaled and (accumulate != 0) added onto existing values:
 *   grad = (accumulate ? grad : 0) + out_scale * dU/dx ,  hw likewise -- the force sum of a Stack
 *   (torchmd/interface.py:396-401) without extra launches: F += -dU/dx of this term.  accumulate bit 1 (value 2): the list
 *   was searched with a skin (mdg_nbr_verlet_rebuild); every pair is re-tested with the list builders' own arithmetic
 *   (D = x_j - x_i, reference minimum image, un-contracted d^2 < cutoff^2) before it counts */
int mdg_pair_eval_ell_into(const float* pos, int n_atoms, const MdgCell* cell /*host*/,
                      const int32_t* col, const int32_t* shift, const int32_t* cnt, int max_nbr,
                      const MdgPairTerm* term /*host*/, const float* theta,
                      const float* w,
                      float* energy, float* grad, float* gtheta,
                      float* hw, float* gtheta_w,
                      float* partial, float out_scale, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------
 * K5-K7  fused trajectories for small systems (one workgroup per replica, state in LDS,
 * all-pairs minimum image re-evaluated at every force call = the reference's
 * topology_update_freq=1 semantics).
 *
 * forward  (replaces odeint(NH_verlet|verlet): torchmd/sovlers.py:106-127 / 21-40,
 *           torchmd/tinydiffeq.py:56-76, RHS torchmd/md.py:210-240 / 133-150)
 *   ensemble: 0 = NoseHooverChain, 1 = NVE (dv/dt = F, no 1/m: md.py:145-148)
 *   v0,q0 [R,N,3], pv0 [R,C]; t [T] time grid (dt_k = t[k+1]-t[k] in fp32 like the
 *   reference); outputs v_t,q_t [R,T,N,3], pv_t [R,T,C] (frame 0 = inputs).
 *   The force at q_k is evaluated once and reused by the next step (bit-compatible with
 *   the reference's two calls).
 * adjoint  (replaces OdeintAdjointMethod.backward: torchmd/sovlers.py:211-293 with the
 *           backward branches :129-164 / :42-101)
 *   g_* are dL/d(frames) (any may be NULL = zeros); outputs adj_v0,adj_q0 [R,N,3],
 *   adj_pv0 [R,C], adj_theta [R,K] (sum over R on the caller's side; for an MDG_PAIR_TABLE term only that sum is
 *   defined: the wave-per-replica kernels hand the table gradient of eight replicas to the first one's row).
 * nonfinite (optional int32[R]): set to 1 for replicas whose state became non-finite.
 */
typedef struct MdgTrajParams {
    int32_t n_rep, n_atoms, n_frames, n_chains;
    int32_t ensemble;
    int32_t block;                 /* 0 = choose */
    float   T;                     /* NHC target temperature (energy units) */
    float   n_dof;                 /* N * dim  (md.py:187) */
    float   Q[MDG_MAX_CHAINS];     /* md.py:191-193 */
} MdgTrajParams;

int mdg_traj_fwd_small(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/,
                       const MdgTerms* terms /*host*/, const float* theta,
                       const float* mass, const float* t_grid,
                       const float* v0, const float* q0, const float* pv0,
                       float* v_t, float* q_t, float* pv_t, int32_t* nonfinite, void* stream);
int mdg_traj_adj_small(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/,
                       const MdgTerms* terms /*host*/, const float* theta,
                       const float* mass, const float* t_grid,
                       const float* v_t, const float* q_t, const float* pv_t,
                       const float* g_v, const float* g_q, const float* g_pv,
                       float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                       void* stream);
/* The same two launches for integrators with topology_update_freq > 1 (torchmd/md.py:200-204: the lists are rebuilt only
 * at the calls whose running count is a multiple of the frequency -- the counter advances on EVERY right-hand-side call,
 * the adjoint's included -- and are stale in between: pair set and image flags frozen at the rebuild positions, no cutoff
 * re-test, i.e. PairPotentials.forward over self.nbr_list / self.offsets, interface.py:298-300).
 *   freq     topology_update_freq;  count0 = the integrator's update_count at the launch's first call.  The forward launch
 *            makes 2 (n_frames - 1) calls (sovlers.py:110-127); the adjoint 3 (n_frames - 1): per interval the dL/dt evaluation
 *            (sovlers.py:258), the first augmented evaluation at the same state, the midpoint evaluation
 *   code     uint16 [n_rep][n_atoms][n_atoms] (mdg_traj_stale_words words), persistent across launches like the reference's
 *            nbr_list / offsets attributes: 0 = no pair, else (term bits << 5) | image code; written at rebuilds
 * Built-in pair forms (any number of terms, masks, any cell) on the generic one-workgroup-per-replica kernels. */
int64_t mdg_traj_stale_words(int n_rep, int n_atoms);
int mdg_traj_fwd_small_stale(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/, const MdgTerms* terms /*host*/,
                             const float* theta, const float* mass, const float* t_grid,
                             const float* v0, const float* q0, const float* pv0,
                             float* v_t, float* q_t, float* pv_t, int32_t* nonfinite,
                             int freq, int64_t count0, uint16_t* code, void* stream);
int mdg_traj_adj_small_stale(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/, const MdgTerms* terms /*host*/,
                             const float* theta, const float* mass, const float* t_grid,
                             const float* v_t, const float* q_t, const float* pv_t,
                             const float* g_v, const float* g_q, const float* g_pv,
                             float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                             int freq, int64_t count0, uint16_t* code, void* stream);


/* Fused observable (extension): the radial distribution function of torchmd/observable.py:62-76 evaluated on frames
 * of the trajectory INSIDE the trajectory kernels -- the force sweep already holds every pair distance of a frame, so
 * the forward launch also fills the raw soft histogram (sum over the selected frames of all replicas, the quantity
 * mdg_rdf_fwd_uniform returns for those frames) and the adjoint launch takes g_raw = dL/d(raw) in place of the
 * frame gradients dL/dq_t that mdg_rdf_bwd_uniform would have written to HBM (an additional g_q is still accepted).
 * Available where the wave-per-replica kernels run (one unmasked built-in pair term, orthorhombic cell, N <= 128, and
 * block = 64 or n_rep >= 1024) and for equally spaced centres whose fine grids fit the LDS and start above zero: mdg_traj_rdf_supported() != 0. */
typedef struct MdgRdfFuse {
    const float* mu;               /* device [nbins] centres, equally spaced (GaussianSmearing offsets) */
    int32_t nbins;
    float   coeff;                 /* -0.5 / width^2 */
    float   mu0, spacing;          /* mu[0], mu[1] - mu[0] */
    float   cutoff;                /* pair cutoff of the observable (observable.py:52: r_range end + 0.5) */
    int32_t frame_start, frame_stride;   /* frames frame_start + k frame_stride of every replica */
} MdgRdfFuse;
int mdg_traj_rdf_supported(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/,
                           const MdgTerms* terms /*host*/, const MdgRdfFuse* rdf /*host*/);
int mdg_traj_fwd_small_rdf(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/,
                           const MdgTerms* terms /*host*/, const float* theta,
                           const float* mass, const float* t_grid,
                           const float* v0, const float* q0, const float* pv0,
                           float* v_t, float* q_t, float* pv_t, int32_t* nonfinite,
                           const MdgRdfFuse* rdf /*host*/, float* raw /*[nbins]*/, void* stream);
int mdg_traj_adj_small_rdf(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/,
                           const MdgTerms* terms /*host*/, const float* theta,
                           const float* mass, const float* t_grid,
                           const float* v_t, const float* q_t, const float* pv_t,
                           const float* g_v, const float* g_q, const float* g_pv,
                           float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                           const MdgRdfFuse* rdf /*host*/, const float* g_raw /*[nbins]*/, void* stream);

/* ------------------------------------------------------------------------------------
 * K5-K7 for systems beyond one workgroup (N <= 32768, NoseHooverChain or NVE): same contract as
 * mdg_traj_fwd_small / mdg_traj_adj_small, three launches per step (forward) / four per adjoint
 * interval, enqueued by a host loop; the neighbour search is fused into the force kernel (per-wave LDS list).
 * Verlet reuse: a search uses the cutoff rc + skin (skin = 12 % of the largest cutoff) and keeps the candidate
 * indices in the workspace; later force evaluations -- forward steps until an atom has moved 0.45 skin from where the
 * list was built (checked on the device every step), and the adjoint's two evaluations per interval -- gather those
 * candidates and re-apply the exact cutoff test, so every evaluation sees the pair set of a fresh search
 * (topology_update_freq = 1, torchmd/sovlers.py:114) while the search itself runs every few steps.
 * (MdgTrajParams.block = -1 switches the reuse off: a fresh search at every evaluation; the adjoint run with the
 * lists reports through flags[5] when that is required.)  mdg_traj_large_list_builds() reports which search served
 * each frame (diagnostics / tests).
 * ws: f32 workspace of mdg_traj_large_workspace() floats, shared by the
 * forward and the adjoint call of one trajectory; flags: int32[8], zeroed by the caller = {neighbour buffer overflow
 * (needed entries), non-finite state, table-gradient range, pair below the table, [4] forward: a stored candidate row
 * overflowed -- call the adjoint with block = -1, [5] adjoint: the stored candidates did not cover an evaluation (row
 * overflow, or a midpoint farther than skin/2 from its frame) -- repeat the adjoint with block = -1}.
 */
int64_t mdg_traj_large_workspace(int n_rep, int n_atoms, int n_frames, int n_theta_total);
int mdg_traj_fwd_large(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/,
                       const MdgTerms* terms /*host*/, const float* theta,
                       const float* mass, const float* t_grid,
                       const float* v0, const float* q0, const float* pv0,
                       float* v_t, float* q_t, float* pv_t, float* ws, int32_t* flags, void* stream);
/* after mdg_traj_fwd_large on `ws` (same sizes): build_of_frame[r][f] (device int32 [n_rep][n_frames]) = the frame whose
 * search produced the candidate list that served frame f; returns 1 (nothing written) when the lists are not kept */
int mdg_traj_large_list_builds(const float* ws, int n_rep, int n_atoms, int n_frames, int n_theta_total,
                               int32_t* build_of_frame, void* stream);
int mdg_traj_adj_large(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/,
                       const MdgTerms* terms /*host*/, const float* theta,
                       const float* mass, const float* t_grid,
                       const float* v_t, const float* q_t, const float* pv_t,
                       const float* g_v, const float* g_q, const float* g_pv,
                       float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                       float* ws, int32_t* flags, void* stream);
/* The same two trajectories for integrators with topology_update_freq > 1 (torchmd/md.py:200-204; see mdg_traj_*_small_stale
 * for the call counting: 2 (n_frames - 1) right-hand-side calls per forward launch, 3 (n_frames - 1) per adjoint launch, a
 * rebuild at every call whose running count is a multiple of `freq`, stale pair set + frozen image flags and no cutoff re-test
 * in between, torchmd/interface.py:298-300).  The host loop knows each call's count and launches a searching or a
 * stored-row force kernel accordingly; a forward step whose NEXT first call rebuilds gets one more force launch at the same
 * positions (the two right-hand-side calls that share a state then see different lists).  No Verlet reuse here (the lists ARE
 * the semantics).  Built-in pair forms, any number of terms, masks, any cell, N <= 32 768.
 *   rows  uint32 [mdg_traj_large_stale_words(n_rep, n_atoms)], persistent across launches like the reference's nbr_list /
 *         offsets attributes: per replica and atom 256 entries j | image code << 15 | term bits << 20 in ascending j, then
 *         the int32 counts [n_rep][n_atoms]; flags[0] reports a row overflow as in mdg_traj_fwd_large. */
int64_t mdg_traj_large_stale_words(int n_rep, int n_atoms);
int mdg_traj_fwd_large_stale(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/, const MdgTerms* terms /*host*/,
                             const float* theta, const float* mass, const float* t_grid,
                             const float* v0, const float* q0, const float* pv0,
                             float* v_t, float* q_t, float* pv_t, float* ws, int32_t* flags,
                             int freq, int64_t count0, uint32_t* rows, void* stream);
int mdg_traj_adj_large_stale(const MdgTrajParams* prm /*host*/, const MdgCell* cell /*host*/, const MdgTerms* terms /*host*/,
                             const float* theta, const float* mass, const float* t_grid,
                             const float* v_t, const float* q_t, const float* pv_t,
                             const float* g_v, const float* g_q, const float* g_pv,
                             float* adj_v0, float* adj_q0, float* adj_pv0, float* adj_theta,
                             float* ws, int32_t* flags, int freq, int64_t count0, uint32_t* rows, void* stream);

/* ------------------------------------------------------------------------------------
 * K8  soft-histogram RDF  (replaces rdf.forward and its autograd backward:
 *     torchmd/observable.py:62-76 with GaussianSmearing nff/nn/layers.py:14-31)
 *   xyz [F,N,3]; pairs i<j with 0 < d < cutoff (min image) over all frames;
 *   raw[k] = sum exp(coeff (d - mu_k)^2), mu_k = mu0 + k*dmu, k < nbins.
 *   fwd writes raw[nbins] (normalisation / volume factors are cheap host-side torch ops).
 *   bwd: given g_raw[nbins] = dL/draw, writes g_xyz [F,N,3].
 *   partial: f32 scratch of mdg_rdf_partial_size(...) floats.
 */
int64_t mdg_rdf_partial_size(int n_frames, int n_atoms, int nbins);
int mdg_rdf_fwd(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell /*host*/,
                float cutoff, const uint8_t* mask, const float* mu /*[nbins]*/, float coeff,
                int nbins, float* raw, float* partial, void* stream);
/* same, with the caller's guarantee that mu is an equally spaced grid (mu_k = mu[0] + k*spacing, as
 * torch.linspace gives): enables the recurrence kernels (one lane per pair with wave-private LDS
 * histograms when there are >= 1024 frames, 8 bins per thread otherwise). */
int mdg_rdf_fwd_uniform(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell /*host*/,
                        float cutoff, const uint8_t* mask, const float* mu, float spacing, float coeff,
                        int nbins, float* raw, float* partial, void* stream);
int mdg_rdf_bwd(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell /*host*/,
                float cutoff, const uint8_t* mask, const float* mu, float coeff, int nbins,
                const float* g_raw, float* g_xyz, void* stream);
/* Soft histogram of a system given as a per-atom neighbour list (large systems, few frames; frames stacked as groups of
 * the list): every pair (slot neighbour index above the atom's) counted once on a fine integer grid, then smeared onto
 * the equally spaced centres -- the pair search of torchmd/observable.py:64-66 is the list build.  The gradient is a
 * tabulated pair force (mdg_pair_eval_ell with an MDG_PAIR_TABLE term built from g_raw). */
int mdg_rdf_ell_supported(float spacing, float coeff, int nbins);
int mdg_rdf_fwd_ell(const float* pos, int64_t n_atoms_total, const MdgCell* cell /*host*/, const int32_t* col,
                    const int32_t* shift, const int32_t* cnt, int max_nbr, const float* mu, float spacing, float coeff,
                    int nbins, float* raw, void* stream);
/* The same observable straight from the cell bins, without materialising a neighbour list (the list build costs more
 * than the histogram): per frame, positions sorted by (bin, atom index); a wave per atom walks the 27-bin stencil and
 * counts every pair once on the fine integer grid (forward), or sums the tabulated pair force of
 * phi(d) = sum_k g_k exp(coeff (d - mu_k)^2) (backward: term/theta = the MDG_PAIR_TABLE built from dL/d raw, as for
 * mdg_pair_eval_ell; g_xyz [F][N][3] = dL/dxyz).  Orthorhombic cells of >= 3 `cutoff` per side, N <= 32 768
 * (mdg_rdf_cell_supported); cutoff = the list cutoff (>= the reach of the fine grid).  scratch: int32 words of
 * mdg_rdf_cell_scratch(), 16-byte aligned, filled by the forward call and read by the backward call of the same frames.
 * Replaces torchmd/observable.py:62-76 (generate_nbr_list + GaussianSmearing(...).sum(0)) and its autograd transpose. */
int mdg_rdf_cell_supported(int n_atoms, const MdgCell* cell /*host*/, float cutoff);
int64_t mdg_rdf_cell_scratch(int n_frames, int n_atoms, const MdgCell* cell /*host*/, float cutoff);
int mdg_rdf_fwd_cell(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell /*host*/, float cutoff,
                     const float* mu, float spacing, float coeff, int nbins, float* raw, int32_t* scratch, void* stream);
int mdg_rdf_bwd_cell(int n_frames, int n_atoms, const MdgCell* cell /*host*/, float cutoff,
                     const MdgPairTerm* term /*host*/, const float* theta, const int32_t* scratch, float* g_xyz,
                     void* stream);
/* backward with the same equally-spaced-centres guarantee (mdg_rdf_bwd makes no assumption on mu). */
int mdg_rdf_bwd_uniform(const float* xyz, int n_frames, int n_atoms, const MdgCell* cell /*host*/,
                        float cutoff, const uint8_t* mask, const float* mu, float spacing, float coeff,
                        int nbins, const float* g_raw, float* g_xyz, void* stream);

/* ------------------------------------------------------------------------------------
 * K10  graph gather/scatter for the SchNet continuous-filter convolution
 * (replaces the index / scatter_add_ chains of nff/nn/models/schnet.py:142,
 *  nff/nn/modules.py:564-571, nff/nn/graphconv.py:43-53 and their autograd transposes).
 * nbr = int64 [E,2] half list (i<j); col/eid/cnt = ELL list with undirected edge ids
 * (mdg_nbr_half_fill).  Feature matrices are row-major fp32.
 *   mdg_edge_diff    out[e,c] = x[i_e,c] - x[j_e,c]
 *   mdg_edge_scatter out[n,c] = sum_{slots s of n} (+/-) g[eid_s,c]     (+ when n is i_e)
 *   mdg_cfconv_agg   out[n,f] = sum_{slots s of n} h[col_s,f] * W[eid_s,f]
 *   mdg_edge_prod    out[e,f] = a[i_e,f] b[j_e,f] + a[j_e,f] b[i_e,f]
 */
int mdg_edge_diff(const float* x, const int64_t* nbr, int64_t n_edges, int n_feat, float* out, void* stream);
int mdg_edge_scatter(const float* g, const int32_t* col, const int32_t* eid, const int32_t* cnt,
                     int n_atoms, int max_nbr, int n_feat, float* out, void* stream);
int mdg_cfconv_agg(const float* h, const float* W, const int32_t* col, const int32_t* eid,
                   const int32_t* cnt, int n_atoms, int max_nbr, int n_feat, float* out, void* stream);
int mdg_edge_prod(const float* a, const float* b, const int64_t* nbr, int64_t n_edges, int n_feat,
                  float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * K9  continuous-filter generator on the matrix cores (fp32-input MFMA, exact f32)
 * (replaces GaussianSmearing -> Dense(G,G) -> shifted_softplus -> Dense(G,F):
 *  nff/nn/modules.py:531-541, nff/nn/layers.py:14-31,86-134, nff/nn/activations.py:5-11)
 *   d [E] distances; mu, width [G] (coeff_k = -0.5 / width_k^2); W1 [G,G], b1 [G], W2 [F,G],
 *   b2 [F] in torch.nn.Linear layout ([out,in]); out [E,F].   G <= 64.
 */
int mdg_cfconv_filter(const float* d, int64_t n_edges, const float* mu, const float* width, int n_gauss,
                      const float* W1, const float* b1, const float* W2, const float* b2,
                      int n_filters, float* out, void* stream);

/* bf16-operand variant (v_mfma_f32_16x16x32_bf16, fp32 accumulate / bias / output). */
int mdg_cfconv_filter_bf16(const float* d, int64_t n_edges, const float* mu, const float* width, int n_gauss,
                           const float* W1, const float* b1, const float* W2, const float* b2,
                           int n_filters, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Velocity observables as single fused passes over v_t [T, N, 3]:
 *   mdg_vacf_fwd     out[t] = mean over frames s, atoms, components of v[s+t] v[s], t = 0 .. n_lags-1
 *                    (torchmd/observable.py:153-163 vacf.forward; n_dof_per_frame = N * 3)
 *   mdg_vacf_bwd     g_v = d(sum_t g_out[t] out[t]) / dv
 *   mdg_temperature  out[f] = sum_n m_n |v_n|^2 / n_dof   (torchmd/thermo.py:57-66 on every frame; n_dof = N * dim)
 * workspace: mdg_vacf_workspace(n_lags) floats.
 */
int64_t mdg_vacf_workspace(int n_lags);
int mdg_vacf_fwd(const float* v, int n_frames, int64_t n_dof_per_frame, int n_lags, float* out, float* workspace,
                 void* stream);
int mdg_vacf_bwd(const float* v, const float* g_out, int n_frames, int64_t n_dof_per_frame, int n_lags, float* g_v,
                 void* stream);
int mdg_temperature(const float* v, const float* mass, int n_frames, int n_atoms, float n_dof, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * K9 + K10 fused: the whole SchNet interaction block on the MD path, nothing edge-sized in HBM
 * (replaces SchNetConv.message/aggregate -- nff/nn/modules.py:531-541,564-571, nff/nn/graphconv.py:43-53,
 *  the distance line nff/nn/models/schnet.py:142 -- and what autograd / double autograd derive from them for
 *  the force -dU/dx and for the adjoint's d(w.F)/dx, d(w.F)/dtheta, torchmd/sovlers.py:229-233).
 * Filter network: mu, coef [G] (coef_k = -0.5 / width_k^2), W1 [G,G], b1 [G], W2 [F,G], b2 [F] in
 * torch.nn.Linear layout.  G <= 64; F <= 128, a multiple of 4 (F <= 64) or 8 (mdg_cfconv_supported).
 *
 *   mdg_edge_geom      d_e = |x_i - x_j - o_e|, uhat_e; with w: dd_e = uhat_e . (w_i - w_j), ddel_e = w_i - w_j
 *   mdg_cfconv_fwd     m[n] = sum_{slots s of n} h[col_s] (.) W(d_s)                        (dd == NULL)
 *                      md[n] = sum_s h[col_s] (.) dW/dd(d_s) dd_s + hd[col_s] (.) W(d_s)    (tangent; hd nullable)
 *                      hsum[n] = sum_s h[col_s], hdsum likewise (nullable; feed the bias gradient of Dense2)
 *                      Symmetric in the adjacency: called with (h, hd) := (mdb, mb) it returns the adjoints
 *                      (hdb, hb) of (hd, h) in the reverse sweep.
 *   mdg_cfconv_bwd     with Wdb_e = mdb_i h_j + mdb_j h_i and (dual sweep, mb != NULL)
 *                      Wb_e = mb_i h_j + mb_j h_i + mdb_i hd_j + mdb_j hd_i:
 *                        dd_b[e] += d(Wdb_e . Wd_e)/d(dd_e)        d_b[e] += d(Wdb_e . Wd_e + Wb_e . W_e)/d(d_e)
 *                        gW1, gb1, gW2 <- gradients of sum_e (Wdb_e . Wd_e + Wb_e . W_e)   (gW1 != NULL; dual only)
 *                      Called with mdb := dU/dm and mb == NULL it is the plain reverse sweep: dd_b[e] += dU/dd_e.
 *                      workspace: mdg_cfconv_bwd_workspace() floats when gW1 is requested.
 *   mdg_edge_geom_bwd  force[n] = -sum_s sgn_s dd_b uhat ;  dwf[n] = -sum_s sgn_s [d_b uhat + dd_b/d (ddel - dd uhat)]
 *                      (d_b nullable: force only)
 */
typedef struct {
    const float* mu;
    const float* coef;
    const float* W1;
    const float* b1;
    const float* W2;
    const float* b2;
    int32_t n_gauss, n_filters;
} MdgFilterNet;               /* host struct of DEVICE pointers */

int mdg_cfconv_supported(int n_gauss, int n_filters);
int mdg_edge_geom(const float* x, const float* w, const int64_t* nbr, const float* offsets, int64_t n_edges,
                  float* d, float* uhat, float* dd, float* ddel, void* stream);
/* mdg_edge_geom for a list searched with a skin: pairs that fail the builders' cutoff test at the current positions get
 * d = -1 (and dd = 0) and are skipped by the cfconv kernels like padding */
int mdg_edge_geom_masked(const float* x, const float* w, const int64_t* nbr, const float* offsets, int64_t n_edges,
                         const MdgCell* cell /*host*/, float cutoff, float* d, float* uhat, float* dd, float* ddel, void* stream);
/* mdg_edge_geom (cell == NULL) / mdg_edge_geom_masked and the zero fill of zero_n floats at `zero` (nullable) in one launch */
int mdg_edge_geom_prepare(const float* x, const float* w, const int64_t* nbr, const float* offsets, int64_t n_edges,
                          const MdgCell* cell /*host, nullable*/, float cutoff, float* d, float* uhat, float* dd, float* ddel,
                          float* zero, int64_t zero_n, void* stream);
int mdg_edge_geom_bwd(const float* d_b, const float* dd_b, const float* d, const float* dd, const float* uhat,
                      const float* ddel, const int32_t* col, const int32_t* eid, const int32_t* cnt,
                      int n_atoms, int max_nbr, float* force, float* dwf, void* stream);
int mdg_cfconv_fwd(const MdgFilterNet* net /*host*/, const float* d, const float* dd, const float* h, const float* hd,
                   const int32_t* col, const int32_t* eid, const int32_t* cnt, int n_atoms, int max_nbr,
                   float* m, float* md, float* hsum, float* hdsum, void* stream);
/* bf16-operand MFMA variant of mdg_cfconv_fwd (v_mfma_f32_16x16x32_bf16; fp32 accumulate, biases, activation,
 * products with the node rows and sums): BASELINE config #5's "bf16 cfconv MFMA" on the trajectory path. */
int mdg_cfconv_fwd_bf16(const MdgFilterNet* net /*host*/, const float* d, const float* dd, const float* h,
                        const float* hd, const int32_t* col, const int32_t* eid, const int32_t* cnt, int n_atoms,
                        int max_nbr, float* m, float* md, float* hsum, float* hdsum, void* stream);
int64_t mdg_cfconv_bwd_workspace(int n_gauss, int n_filters, int64_t n_edges);
int mdg_cfconv_bwd(const MdgFilterNet* net /*host*/, const float* d, const float* dd, const int64_t* nbr,
                   int64_t n_edges, const float* h, const float* hd, const float* mb, const float* mdb,
                   float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* workspace,
                   const int32_t* n_valid /*device, nullable: real rows of a capacity-padded list*/, void* stream);
/* bf16-operand MFMA variant of mdg_cfconv_bwd (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x16_bf16; everything else fp32):
 * the reverse half of BASELINE config #5's "bf16 cfconv MFMA, full fwd + adjoint".  Same arguments and workspace. */
int mdg_cfconv_bwd_bf16(const MdgFilterNet* net /*host*/, const float* d, const float* dd, const int64_t* nbr,
                        int64_t n_edges, const float* h, const float* hd, const float* mb, const float* mdb,
                        float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* workspace,
                        const int32_t* n_valid, void* stream);
/* either of the two with the gradients of the radial basis as well (GaussianSmearing(trainable=True), nff/nn/layers.py:34-83:
 * `offsets` and `width` are parameters): gmu[G] = d/d(mu_k), gcoef[G] = d/d(coef_k) of the same scalar; the caller chains
 * coef = -0.5 / width^2.  bf16 != 0: bf16 MFMA operands. */
int mdg_cfconv_bwd_smear(const MdgFilterNet* net /*host*/, const float* d, const float* dd, const int64_t* nbr,
                         int64_t n_edges, const float* h, const float* hd, const float* mb, const float* mdb,
                         float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* gmu, float* gcoef,
                         float* workspace, const int32_t* n_valid, int bf16, void* stream);

/* "rows16": the bf16 variants over bf16 MIRRORS of the node matrices the kernels GATHER per edge (h, hd and, in the reverse
 * sweep, mb, mdb): [n_atoms, n_filters] bf16, dense rows, round-to-nearest-even copies of the f32 matrices (mdg_row_chain
 * writes them next to its f32 outputs, mdg_rows_to_bf16 makes one of any f32 matrix).  A gathered row is then one 16-byte
 * load per lane instead of two and half the cache lines; it widens exactly and every product, sum and output stays f32.
 * This is a PRECISION OPTION of its own on top of the bf16 MFMA operands (the gathered features carry 8 significant bits
 * into the products; nff/nn/modules.py:564-571 multiplies f32 features) -- callers opt in explicitly
 * (SchNet.node_rows_bf16, bench.py --bf16-rows) and tests/test_gpu_schnet_rows16.py states its tolerance.
 * Layers of more than 64 filters (mdg_cfconv_rows16_supported).  Outputs, workspace and the other arguments as in
 * mdg_cfconv_fwd_bf16 / mdg_cfconv_bwd_smear (gmu = gcoef = NULL: no basis gradients). */
int mdg_cfconv_rows16_supported(int n_gauss, int n_filters);
int mdg_cfconv_fwd_rows16(const MdgFilterNet* net /*host*/, const float* d, const float* dd, const uint16_t* h16,
                          const uint16_t* hd16, const int32_t* col, const int32_t* eid, const int32_t* cnt, int n_atoms,
                          int max_nbr, float* m, float* md, float* hsum, float* hdsum, void* stream);
int mdg_cfconv_bwd_rows16(const MdgFilterNet* net /*host*/, const float* d, const float* dd, const int64_t* nbr,
                          int64_t n_edges, int n_atoms, const uint16_t* h16, const uint16_t* hd16, const uint16_t* mb16,
                          const uint16_t* mdb16, float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* gmu,
                          float* gcoef, float* workspace, const int32_t* n_valid, void* stream);
/* The G-wide stash of the filter network (round 6): its first Dense layer -- Gaussians, a = g W1^T + b1, s = ssp(a) and, with
 * dd, the tangent sd = sigmoid(a) a_dot (nff/nn/modules.py:531-541, layers 0-2 of message_edge_filter) -- ONCE per undirected
 * edge and evaluation, as [n_edges][W] bf16 rows (W = mdg_cfconv_stash_width(n_gauss): 32 or 64; 64 bytes per edge and
 * quantity at W = 32), zero rows for pairs beyond the cutoff (d = -1) and padding.  mdg_cfconv_fwd_stashed is
 * mdg_cfconv_fwd_bf16 / _rows16 reading the second layer's operands from it instead of recomputing them per directed slot
 * (every edge from both ends, in each of the sweeps of an evaluation): the same bf16 operands, so bitwise the same outputs.
 * d may be NULL when n_gauss + 2 <= W (the second layer's bias then rides in two spare k columns). */
int mdg_cfconv_stash_width(int n_gauss);
int mdg_cfconv_filter_stash(const MdgFilterNet* net /*host*/, const float* d, const float* dd /*nullable*/, int64_t n_edges,
                            const int32_t* n_valid /*device, nullable*/, uint16_t* st_s, uint16_t* st_sd /*with dd*/, void* stream);
int mdg_cfconv_fwd_stashed(const MdgFilterNet* net /*host*/, const uint16_t* st_s, const uint16_t* st_sd /*nullable: no tangent*/,
                           const float* d /*see above*/, const void* h, const void* hd /*nullable*/, const int32_t* col,
                           const int32_t* eid, const int32_t* cnt, int n_atoms, int max_nbr, float* m, float* md /*with st_sd*/,
                           int rows16, void* stream);
/* The reverse sweep with parameter gradients, every option in one entry.  flags: MDG_CFCONV_BF16 (bf16 MFMA operands as
 * mdg_cfconv_bwd_bf16) | MDG_CFCONV_ROWS16 (with BF16: h, hd, mb, mdb are bf16 mirrors, n_atoms rows each).  gb2[n_filters]
 * (nullable; needs mdg_cfconv_bias_column(n_gauss)): d/d b2 of the same scalar, sum over the edges of the filter output's
 * adjoint rows -- it falls out of a spare padded column of the first filter layer whose activation is pinned to 1
 * (csrc/cfconv_fused.hip: B2COL_BIAS), so the caller needs neither the neighbour sums of mdg_cfconv_fwd (hsum / hdsum) nor a
 * reduction over atoms for it (nff/nn/modules.py:531-541: the bias of Dense(n_gaussians -> n_filters)' second layer).
 * gmu / gcoef (nullable, together) as in mdg_cfconv_bwd_smear. */
enum { MDG_CFCONV_BF16 = 1, MDG_CFCONV_ROWS16 = 2 };
int mdg_cfconv_bias_column(int n_gauss);
int mdg_cfconv_bwd_theta(const MdgFilterNet* net /*host*/, const float* d, const float* dd, const int64_t* nbr,
                         int64_t n_edges, int n_atoms, const void* h, const void* hd, const void* mb, const void* mdb,
                         float* d_b, float* dd_b, float* gW1, float* gb1, float* gW2, float* gb2, float* gmu, float* gcoef,
                         float* workspace, const int32_t* n_valid, int flags, void* stream);
/* dst[r, c] = bf16(src[r * src_stride + c]), dense [n_rows, n_cols] bf16 (n_cols, src_stride multiples of 4) */
int mdg_rows_to_bf16(const float* src, int64_t n_rows, int n_cols, int src_stride, uint16_t* dst, void* stream);

/* ------------------------------------------------------------------------------------
 * K11/K12  node-level Dense layers with fused epilogues on the f32 MFMA
 * (replaces nff/nn/layers.py:86-134 Dense + nff/nn/activations.py:5-11 on [N, .] node features: message_node_filter,
 *  the update MLP + residual nff/nn/modules.py:543-547 / schnet.py:149-151, the readout's first Linear, and the
 *  transposed products of the hand-derived reverse sweeps):
 *     z0 = x0 B + bias0 ;  out0 = act(z0) * mul0 + res0 ;  sig0 = sigmoid(z0)            (act = 1: shifted softplus)
 *     z1 = x1 B          ;  out1 = act'(z0) z1 + res1                                     (x1 nullable)
 *  B[k][m] = W[m*k_dim + k] (trans = 0, torch.nn.Linear layout) or W[k*m_dim + m] (trans = 1).  Any k: layers wider than
 *  256 inputs run as k-slabs of 256 that hand their partial sums on through out0 / out1.
 */
int mdg_dense(const float* W, int trans, int act, int n_rows, int k, int m,
              const float* x0, const float* bias0, const float* mul0, const float* res0, float* out0, float* sig0,
              const float* x1, const float* res1, float* out1, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused elementwise / row-reduction pieces of the hand-derived SchNet passes (each replaces a chain of
 * PyTorch elementwise ops; nff/nn/layers.py:14-31, nff/nn/activations.py:5-11 and their derivatives):
 *   mdg_smear        g = exp(c_k (d - mu_k)^2), phi = 2 c_k (d - mu_k)              [E,G]
 *   mdg_ssp          s = softplus(a) - ln 2 (, sa = sigmoid(a))
 *   mdg_mul_row      o = x * y (* r[row])
 *   mdg_ssp_dual_bwd xdb = sa sdb ; xb = sa (1 - sa) xd sdb + sa sb
 *   mdg_smear_bwd    d_b[e] += sum_k (...), dd_b[e] += sum_k (...)   (see csrc/elem.hip)
 */
int mdg_smear(const float* d, const float* mu, const float* c, int64_t n_edges, int n_gauss, float* g, float* phi,
              void* stream);
int mdg_ssp(const float* a, int64_t n, float* s, float* sa, void* stream);
int mdg_mul_row(const float* x, const float* y, const float* r, int64_t n_rows, int n_cols, float* o, void* stream);
int mdg_ssp_dual_bwd(const float* sa, const float* xd, const float* sdb, const float* sb, int64_t n, float* xdb,
                     float* xb, void* stream);
/* the same with the tangent given as t_dot = sa * x_dot:  xdb = sa sdb ; xb = (1 - sa) t_dot sdb + sa sb */
int mdg_ssp_dual_bwd_t(const float* sa, const float* td, const float* sdb, const float* sb, int64_t n, float* xdb,
                       float* xb, void* stream);
/* head of the reverse sweeps through the readout U = sum_i L2 . ssp(y_i) + l2 (nff/nn/modules.py:761-809, schnet.py:155-158):
 * ydb = sy * L2 ; yb = (1 - sy) * syd * L2, with sy = sigmoid(y) and syd = sy * y_dot [n_rows, n_cols], L2 [n_cols];
 * syd = yb = NULL: first-order pass */
int mdg_readout_head(const float* sy, const float* syd, const float* L2, int64_t n_rows, int n_cols, float* ydb, float* yb,
                     void* stream);
int mdg_smear_bwd(const float* gdb, const float* gb, const float* g, const float* phi, const float* dd,
                  const float* c, int64_t n_edges, int n_gauss, float* d_b, float* dd_b, void* stream);

/* ------------------------------------------------------------------------------------
 * Tall-skinny contraction C[M,N] = A[E,M]^T B[E,N] (split-K on the f32 MFMA, ordered reduction):
 * the weight gradients of edge-wise Dense layers in the adjoint's parameter vjp (autograd of
 * nff/nn/layers.py:86-134 on [E, .] inputs).  workspace: mdg_atb_workspace() floats.
 */
int64_t mdg_atb_workspace(int64_t n_rows, int m, int n);
int mdg_atb(const float* A, const float* B, int64_t n_rows, int m, int n, float* C, float* workspace,
            void* stream);
/* C = A^T B + A2^T B2 (same shapes; A2 = B2 = NULL: mdg_atb): primal + tangent halves of a weight gradient at once */
int mdg_atb2(const float* A, const float* B, const float* A2, const float* B2, int64_t n_rows, int m, int n, float* C,
             float* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * K13  a CHAIN of such Dense layers in one launch (csrc/rowchain.hip): between two continuous-filter convolutions every
 * operation of a SchNet block is local to one atom's feature row -- update MLP + residual (nff/nn/modules.py:543-547,
 * nff/nn/models/schnet.py:149-151), the next block's message_node_filter, the readout (nff/nn/modules.py:761-809) -- and so
 * is the stretch of the hand-derived reverse sweeps from the readout head back to the adjoint of the last aggregation.  A
 * workgroup owns 16 rows and walks the stages; a stage's outputs stay on chip as the next stage's input and are copied to
 * the global buffers the caller names.  Per stage (x = in0 / in1 when given, else the previous stage's outputs):
 *     z0 = x0 B + bias ; z1 = x1 B                        B[k][m] = W[m*K + k] (trans = 0) or W[k*M + m] (trans = 1)
 *     act = 1:          out0 = ssp(z0), sig = sigmoid(z0), out1 = sig z1
 *     MDG_CHAIN_MUL     out0 *= aux0[row, m]
 *     MDG_CHAIN_HEAD    (act = 1) pre0 = out0, pre1 = out1 ; out0 = sig aux0[m] ; out1 = (1 - sig) pre1 aux0[m]
 *     MDG_CHAIN_SSP_BWD out0 = s z0 ; out1 = (1 - s) td z0 + s z1         (s = aux0[row, m], td = aux1[row, m])
 *     then out0 += res0[row, m], out1 += res1[row, m]; out0_h / out1_h (when given) receive the same values rounded to bf16
 * dual = 0: only the "0" operands are touched.  aux0 / aux1 / res may be buffers an EARLIER stage of the same call wrote
 * with the same width M (same owner thread); any other aliasing between a stage's outputs and a later stage's inputs is
 * not allowed.  Widths 1..MDG_CHAIN_MAX_WIDTH, 1..MDG_CHAIN_MAX_STAGES stages.
 * flags: MDG_CHAIN_DUAL (= the former `dual` argument: 0 / 1) | MDG_CHAIN_X3: the stage products as three bf16 MFMAs on
 * operands split into bf16 head + bf16 remainder (x_h w_h + x_h w_l + x_l w_h, f32 accumulate: ~1e-5 relative per product
 * instead of 6e-8, 3/16 of the matrix time) -- the companion of the rows16 precision option, honoured by the compiled
 * n_atom_basis = 64 chains and ignored (f32 products) by every other list.  MDG_CHAIN_X6 (instead): THREE exact bf16 pieces
 * per operand and the six piece products that matter -- the accuracy of the f32 matrix instruction (1.8e-7 of sum |terms|,
 * tools/micro/split_mfma.hip) at 3/8 of its issue time; honoured by every compiled chain.
 */
#define MDG_CHAIN_MAX_STAGES 8
#define MDG_CHAIN_MAX_WIDTH 512
enum { MDG_CHAIN_NONE = 0, MDG_CHAIN_MUL = 1, MDG_CHAIN_HEAD = 2, MDG_CHAIN_SSP_BWD = 3 };
enum { MDG_CHAIN_DUAL = 1, MDG_CHAIN_X3 = 2, MDG_CHAIN_X6 = 4 };
typedef struct {
    const float* W;
    const float* bias;
    const float* in0;
    const float* in1;
    const float* res0;
    const float* res1;
    const float* aux0;
    const float* aux1;
    float* out0;
    float* out1;
    float* sig;
    float* pre0;
    float* pre1;
    uint16_t* out0_h;   /* nullable: bf16 mirrors [n_rows, M] of out0 / out1 (the rows16 kernels' inputs), written with them */
    uint16_t* out1_h;
    int32_t K, M, trans, act, mode, pad_;
} MdgChainStage;
int mdg_row_chain(const MdgChainStage* stages, int n_stages, int n_rows, int flags, void* stream);

/* ------------------------------------------------------------------------------------
 * All parameter-gradient reductions of one SchNet adjoint evaluation in two launches (csrc/gradjobs.hip): what double
 * autograd accumulates into the .grad of the Dense weights / biases and the embedding of nff/nn/modules.py:514-575 and
 * nff/nn/models/schnet.py:113-171 while it differentiates w.F (torchmd/sovlers.py:229-233), times the interval weight of
 * sovlers.py:160.  A job is one reduction over `rows` atoms:
 *   MDG_GRAD_ATB     out[m,n] = A[rows,m]^T B[rows,n] (+ A2^T B2); row r of the result goes to destination row
 *                    row_map[r] when row_map != NULL (rows of the embedding table), else r
 *   MDG_GRAD_COLSUM  out[m]   = sum_rows A (.* B) (+ A2 (.* B2))            (B, B2 optional, together)
 *   MDG_GRAD_AXPY    out[m]   = A (+ A2)                                     (already reduced: cfconv_bwd's gW1, gb1, gW2)
 * and lands at flat[out_off ...]:  flat = (accumulate ? flat : 0) + alpha * (t ? t[*idx] - t[*idx - 1] : 1) * out, with the
 * time grid t and the frame index idx on the device (a captured HIP graph advances idx itself).  Fixed-order sums, no
 * atomics.  workspace: mdg_grad_jobs_workspace() floats.  At most MDG_GRAD_JOBS_MAX jobs per call.
 */
#define MDG_GRAD_JOBS_MAX 32
enum { MDG_GRAD_ATB = 0, MDG_GRAD_COLSUM = 1, MDG_GRAD_AXPY = 2 };
typedef struct {
    const float* A;
    const float* B;
    const float* A2;
    const float* B2;
    const int64_t* row_map;
    int64_t rows;
    int32_t m, n;
    int32_t kind, pad_;
    int64_t out_off;
} MdgGradJob;
int64_t mdg_grad_jobs_workspace(const MdgGradJob* jobs, int n_jobs);
int mdg_grad_jobs(const MdgGradJob* jobs, int n_jobs, float* flat, float alpha, const float* t, const int64_t* idx,
                  int accumulate, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------
 * One SchNet evaluation per call (csrc/schnet_eval.hip, round 6): what GNNPotentials.forward + compute_grad
 * (torchmd/interface.py:86-136, torchmd/md.py:23-31: F = -dU/dx by autograd through nff/nn/models/schnet.py:113-171) and the
 * adjoint's vector-Jacobian product of it (torchmd/sovlers.py:229-233: d(w.F)/dx, d(w.F)/dtheta by double autograd) compute,
 * as the hand-derived sweeps of mdgrad_amd/nn/analytic.py -- primal (+ tangent along w), turn at the readout, reverse -- with
 * every launch of the evaluation (edge geometry, the fused interaction-block kernels, the row chains of the node-level layers,
 * the batched parameter-gradient reductions) enqueued by one C++ loop on `stream`.  The caller describes the network and the
 * topology with DEVICE pointers and hands over one workspace of mdg_schnet_workspace() floats; nothing is allocated inside.
 *
 *   layer[i]   one interaction block (nff/nn/modules.py:514-575): Gaussian basis + filter network (`filt`), message_node_filter
 *              (Wn [F, A], bn), update MLP (U1 [A, F], c1; U2 [A, A], c2), torch.nn.Linear layout.  bf16 / bf16_rev: bf16 MFMA
 *              operands in the forward-type / reverse sweeps of the filter network; rows16: the block's kernels gather bf16
 *              mirrors of the node rows (mdg_cfconv_*_rows16); b2col: d/d b2 from the spare filter column (mdg_cfconv_bwd_theta).
 *              off_*: offset of each parameter in the flat parameter-gradient vector (tinydiffeq.py:106-108 order).
 *   readout    L1 [H, A], l1 [H], L2 [H] (nff/nn/utils.py:56-75: Linear, shifted_softplus, Linear(H -> 1)).
 *   r0, h0     embedding rows atom_embed.weight[z] [N, A] and the first block's filtered rows Wn r0 + bn [N, F] (neither depends
 *              on the positions: persistent buffers of the caller); h0_16: bf16 mirror of h0 (first block rows16).
 *   onehot, uniq, n_species    [N, S] one-hot of the species and the S embedding rows in use (gradient of atom_embed.weight).
 *   topology   half list nbr [E, 2] + image offsets [E, 3] and the per-atom rows col / eid / cnt of the same list
 *              (mdg_nbr_*; capacity-padded lists: n_valid = device count of real rows).  masked: the list was searched with a
 *              skin -- pairs beyond `cutoff` in `cell` at the current positions are skipped (mdg_edge_geom_masked).
 * Results are bitwise those of the launch-by-launch sequence (tests/test_gpu_schnet_plan.py).
 */
#define MDG_SCHNET_MAX_LAYERS 8
typedef struct {
    MdgFilterNet filt;
    const float *Wn, *bn, *U1, *c1, *U2, *c2;
    int64_t off_W1, off_b1, off_W2, off_b2, off_Wn, off_bn, off_U1, off_c1, off_U2, off_c2;
    int32_t bf16, bf16_rev, rows16, b2col;
} MdgSchnetLayer;
typedef struct {
    int32_t n_atoms, n_layers, n_atom_basis, n_readout;
    MdgSchnetLayer layer[MDG_SCHNET_MAX_LAYERS];
    const float *L1, *l1, *L2;
    int64_t off_L1, off_l1, off_L2, off_embed;
    const float *r0, *h0;
    const uint16_t* h0_16;
    const float* onehot;
    const int64_t* uniq;
    int32_t n_species, masked;
    const int64_t* nbr;
    const float* offsets;
    int64_t n_edges;
    const int32_t *col, *eid, *cnt;
    const int32_t* n_valid;
    int32_t max_nbr;
    float cutoff;
    MdgCell cell;
    float* ws;
    int64_t ws_floats;
    int32_t stash, chain_x3;  /* chain_x3: MDG_CHAIN_X3 or MDG_CHAIN_X6 (or 0), handed to mdg_row_chain with every node-level chain.  stash != 0: blocks with bf16 operands run mdg_cfconv_filter_stash once per evaluation and the
                                 stashed forward-type sweeps (mdg_cfconv_fwd_stashed): same results, fewer instructions */
} MdgSchnetPlan;              /* host struct of DEVICE pointers */
int64_t mdg_schnet_plan_sizeof(void);      /* sizeof(MdgSchnetPlan): bindings in other languages check their layout against it */
/* floats of workspace: dual = 0 for mdg_schnet_force, dual = 1 for mdg_schnet_force_vjp (theta != 0: with parameter gradients) */
int64_t mdg_schnet_workspace(const MdgSchnetPlan* plan /*host*/, int dual, int theta);
/* force [N, 3] = -dU/dx.  energy_colsum [H] (nullable): column sums of the readout's activations; U = L2 . colsum + N l2 */
int mdg_schnet_force(const MdgSchnetPlan* plan /*host*/, const float* x, float* force, float* energy_colsum, void* stream);
/* force, dwf [N, 3] = d(w.F)/dx and, theta_flat != NULL, theta_flat += alpha * (t ? t[*idx] - t[*idx - 1] : 1) * dU_dot/dtheta
 * (callers pass alpha = -1: w.F = -U_dot; t / idx: the device-side interval weight of sovlers.py:160, together or NULL) */
int mdg_schnet_force_vjp(const MdgSchnetPlan* plan /*host*/, const float* x, const float* w, float* force, float* dwf,
                         float* theta_flat, float alpha, const float* t, const int64_t* idx, float* energy_colsum, void* stream);

/* ------------------------------------------------------------------------------------
 * Nose-Hoover-chain algebra of the generic (non-fused) integrator path as single launches
 * (replaces the elementwise/reduction ops of NoseHooverChain.forward after the force,
 *  torchmd/md.py:221-240, and the thermostat part of its vjp).
 * State of R stacked replicas: v, f, lv, lq [R*n, 3]; pv, lp [R, C]; mass [R*n]; Q [C].
 *   mdg_nhc_rhs:  a = (f - pv0 m v / Q0) / m ;  dpv = bath right-hand side (KE per replica)
 *   mdg_nhc_vjp:  Gv = -(pv0/Q0) lv + lq + 2 m v lp0 ;  Gp = lam^T d(bath rhs)/d pv - (lv.v)/Q0 e0
 */
int mdg_nhc_rhs(const float* v, const float* f, const float* pv, const float* mass, const float* Q,
                const float* T /*device, [1]*/, float n_dof, int n_rep, int n_atoms, int n_chains, float* a,
                float* dpv, void* stream);
int mdg_nhc_vjp(const float* v, const float* pv, const float* lv, const float* lq, const float* lp,
                const float* mass, const float* Q, int n_rep, int n_atoms, int n_chains, float* Gv,
                float* Gp, void* stream);
/* Whole half-steps of the generic NH-Verlet path (interactions that are not pair-fusable, e.g. SchNet) as single
 * launches: torchmd/sovlers.py:110-127 (forward) and :129-164 with :253-288 (adjoint interval), the force / force-vjp
 * evaluations in between stay separate calls.  Time-major frames [T, R*n, 3] / [T, R, C]; idx: device int64[1], the
 * step k (forward) or frame i (adjoint) -- a captured HIP graph advances it itself; dt = t[k+1] - t[k] is read on the
 * device.
 *   mdg_nhv_kick     dv_h = 1/2 a dt, dp_h = 1/2 b dt, qn = q + (v + dv_h) dt           (rhs at (v, q, pv), cached force f)
 *   mdg_nhv_finish   v += dv_h + 1/2 a1 dt, pv += dp_h + 1/2 b1 dt, q = qn, f = fn ; frame k+1 stored
 *   mdg_nhv_adj_pre  (v, q, pv) <- frame i ;  w = lam_v / m
 *   mdg_nhv_adj_mid  vh = v - a hh, qm = q + vh h, pm = pv - b hh, lam_h = lam + G0 hh, wh = lam_h_v / m
 *   mdg_nhv_adj_end  lam += G1 h + dL/dy_{i-1}
 * A replica's elements are cut over several workgroups; `scratch` (mdg_nhv_scratch_floats(n_rep, n_atoms) floats,
 * zeroed ONCE by the caller, reusable by every launch of the same shape on one stream) carries the per-workgroup
 * partial sums of the kinetic energy / <lam_v, v> and a ticket per replica; the result does not depend on the order
 * the workgroups finish in.  advance != 0 (mdg_nhv_finish / mdg_nhv_adj_end): the launch also moves the counter, idx[0] <- k + 1
 * resp. i - 1, once every workgroup has read the old value -- no separate increment launch per step. */
int64_t mdg_nhv_scratch_floats(int n_rep, int n_atoms);
int mdg_nhv_kick(const float* v, const float* q, const float* pv, const float* f, const float* mass, const float* Q,
                 const float* T, float n_dof, const float* t, const int64_t* idx, int n_rep, int n_atoms, int n_chains,
                 float* dv_h, float* dp_h, float* qn, float* scratch, void* stream);
int mdg_nhv_finish(float* v, float* q, float* pv, float* f, const float* dv_h, const float* dp_h, const float* qn,
                   const float* fn, const float* mass, const float* Q, const float* T, float n_dof, const float* t,
                   int64_t* idx, int advance, int n_rep, int n_atoms, int n_chains, float* out_v, float* out_q, float* out_pv,
                   float* scratch, void* stream);
int mdg_nhv_adj_pre(const float* v_t, const float* q_t, const float* pv_t, const float* lv, const float* mass,
                    const int64_t* idx, int n_rep, int n_atoms, int n_chains, float* v, float* q, float* pv, float* w,
                    void* stream);
int mdg_nhv_adj_mid(const float* v, const float* q, const float* pv, const float* lv, const float* lq, const float* lp,
                    const float* f, const float* dwf, const float* mass, const float* Q, const float* T, float n_dof,
                    const float* t, const int64_t* idx, int n_rep, int n_atoms, int n_chains, float* vh, float* qm, float* pm,
                    float* lvh, float* lqh, float* lph, float* wh, float* scratch, void* stream);
int mdg_nhv_adj_end(const float* vh, const float* pm, const float* lvh, const float* lqh, const float* lph, const float* dwf,
                    const float* mass, const float* Q, const float* t, int64_t* idx, int advance, const float* g_v,
                    const float* g_q, const float* g_pv, int n_rep, int n_atoms, int n_chains, float* lv, float* lq,
                    float* lp, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------
 * f4  bonded terms over a static topology table (csrc/bonded.hip).
 * Replaces torchmd/interface.py:447-455 (BondPotentials.forward: harmonic in the SQUARED bond length,
 * 1/2 k (|b|^2 - ro)^2) and :496-508 (AnglePotentials.forward: 1/2 k (theta - theta0)^2 over triples (i, j, k) centred on
 * j), their image flags topology.get_offsets (topology.py:75-80: -[b >= L/2] + [b < -L/2], L = cell.diag()), the autograd
 * force behind them (torchmd/md.py:228-230) and the double-backward Hessian-vector product of the adjoint
 * (torchmd/sovlers.py:229-233) -- one launch, no atomics.
 *   kind      MDG_BONDED_BOND: top = int32 [n_terms, 2], x0 = ro ;  MDG_BONDED_ANGLE: top = int32 [n_terms, 3], x0 = theta0
 *   cell_len  host float[3], the diagonal of the cell
 *   inc_ptr   int32 [n_atoms + 1], inc int32 [sum of roles]: the incidence list of every atom, entries 4 * term + role
 *             (role = the atom's column in `top`), ascending per atom -- the order the per-atom sums run in
 *   w         nullable [n_atoms, 3]; required for hw
 *   e_atom    nullable [n_atoms]: energy of the terms whose FIRST atom this is (sum = U)
 *   grad, hw  nullable [n_atoms, 3]:  grad = (accumulate ? grad : 0) + out_scale * dU/dx ,  hw likewise with H w
 *             (out_scale = -1: the force and d(w.F)/dx, added onto another Stack member's buffers when accumulate != 0)
 */
#define MDG_BONDED_BOND 0
#define MDG_BONDED_ANGLE 1
int mdg_bonded_eval(const float* pos, int n_atoms, const float* cell_len /*host*/, int kind, const int32_t* top, int n_terms,
                    float k, float x0, const int32_t* inc_ptr, const int32_t* inc, const float* w, float* e_atom,
                    float* grad, float* hw, float out_scale, int accumulate, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* MDGRAD_HIP_H */
