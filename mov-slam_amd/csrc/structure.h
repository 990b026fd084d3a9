// Host-side structure pass of a local-BA window: what g2o derives in
// SparseOptimizer::initializeOptimization + BlockSolver::buildStructure (called from
// /root/reference/src/Optimizer.cc:754), re-designed for the device kernels:
//   * hessian index of every pose (free AND active poses, ascending caller order),
//   * edges grouped by map point (stable), so one lane group owns a point,
//   * the upper-triangle pose pairs (i<=j) that share >= 1 point — the block pattern of
//     the reduced camera system — with, per pair, the list of (edge_i, edge_j) entries
//     that contribute B_il Dinv_l B_jl^T, cut into fixed-size work items,
//   * per block-row gather lists for the symmetric block mat-vec of the PCG.
// Pure C++ (no HIP): unit-tested on CPU through movba_structure_probe.
#pragma once
#include <cstdint>
#include <vector>

#include "movba.h"

namespace movba {

struct Int2 { int32_t x, y; };
struct Int4 { int32_t x, y, z, w; };
struct Item { int32_t pair, begin, end, diag; };   // entries [begin,end) of one pair
// one slot of the k_schur launch schedule: everything a workgroup needs to start, in one 32-byte scalar load
// ONE WAVE SLOT of the k_schur launch (four per workgroup): the entries [begin, end) of work item tag >> 1 this wave takes
struct SchedItem {
    int32_t begin, end, tag /* (item << 1) | diagonal, -1 = padding */, pose_i, pose_j /* pose indices of the pair */;
    // off-diagonal items of single-item pairs, windows with an on-chip PCG: where the item's 6 x 6 block goes in DevWindow::img_b,
    // as stored (dst_a) and transposed (dst_b): 36 k 512 + thread of the PCG lane slot that holds it, or -1.
    // Diagonal items: dst_a = the item's record in DevWindow::rec_d (keyframe * rec_slots + place in the pair)
    int32_t dst_a, dst_b;
    int32_t sub;        // this wave's place among the waves that share the item | their number << 8 (place 0 adds the waves' sums up and stores)
};
struct RowEnt { int32_t block, col, transposed, pad; };

struct Structure {
    int NP = 0, P = 0, E = 0, nfree = 0;
    int npairs = 0, nitems = 0, max_degree = 0;
    int64_t nentries = 0;
    bool already_grouped = true;
    bool reordered = false;             // the free keyframes were renumbered by covisibility (covisibility_order)
    bool no_reorder = false;            // caller's wish: keep the given numbering (tests, comparisons)
    int n_fixed = 0;
    std::vector<int32_t> hidx;          // NP: hessian index or -1
    std::vector<int32_t> free_pose;     // nfree: pose index
    std::vector<int32_t> perm;          // E: grouped position -> caller edge; EMPTY when the caller's order is already grouped (identity)
    std::vector<int32_t> tmp_order; std::vector<uint64_t> tmp_key;   // scratch of finish_pairs
    std::vector<int32_t> pose_edges, pose_slot0;   // scratch of build_basic: edges per pose, first pose-major slot of a free pose
    std::vector<int32_t> pt_start;      // P+1
    std::vector<int32_t> g_pose, g_point;   // E (grouped order); left empty when the caller's order is already grouped:
    const int32_t *gp = nullptr, *gl = nullptr;   // the grouped arrays to read: the caller's own, or the two vectors above
    std::vector<int32_t> pair_i, pair_j;    // npairs (hessian indices, i <= j); pair k<nfree is (k,k)
    std::vector<int32_t> pair_item_start;   // npairs+1
    std::vector<int64_t> pair_ptr;          // npairs+1: first entry of every pair
    std::vector<int32_t> pid;               // nfree x nfree: pair id of (i <= j) or -1
    std::vector<int32_t> slot;          // E: pose-major position of a grouped edge among the edges of FREE poses (-1: fixed pose)
    // entry lists of the OFF-diagonal pairs only (host builder; entry k of the global numbering sits at k - E_free):
    // pose-major slot of the edge of pose i, of the edge of pose j, map point.  The diagonal pair of free pose h lists all
    // its edges in slot order, and the diagonal pairs come first: diagonal entry k IS slot k, nothing is stored for it.
    std::vector<int32_t> ent_i, ent_j, ent_l;
    int E_free = 0;                     // edges of free poses = entries of the diagonal pairs
    std::vector<Item> items;            // nitems
    // k_schur launch schedule: 8 segments (one per XCD) of sched_per_xcd wave slots (four per workgroup)
    std::vector<SchedItem> sched;
    int sched_per_xcd = 0;
    std::vector<int32_t> row_ptr;       // nfree+1
    std::vector<RowEnt> row_ent;        // mat-vec gather list per block row
    // coarse level of the two-level PCG preconditioner (build_coarse): keyframe aggregates, 6 dofs each
    int n_agg = 0;
    std::vector<int32_t> cblk_g, cblk_h, cblk_ptr;   // non-empty coarse blocks (g,h) and their term lists
    std::vector<int32_t> cblk_ent;      // fine blocks summed into the coarse block: (index << 2 | transposed << 1 | source); source 1:
                                        // index = work item whose partial is the block (negated), source 0: index = pair id (materialised block)
    std::vector<int32_t> multi_pairs;   // off-diagonal pairs cut into several work items (materialised by the coarse workgroup)
    std::vector<int32_t> cblk_ij;       // (block row << 16 | block column) of the same term: weights of the linear coarse modes
};

// Aggregate g = block rows [agg_row0[g], agg_row0[g+1]): fills the coarse block term lists of
// A_c = P^T S P (P = piecewise-constant prolongation over the aggregates).
void build_coarse(Structure& s, const int32_t* agg_row0, int n_agg);

#ifndef MOVBA_SCHUR_WAVES
#define MOVBA_SCHUR_WAVES 4
#define MOVBA_SCHUR_WPI 4
#define MOVBA_SCHUR_EPW 512
#endif
#ifndef MOVBA_SCHUR_EPW_DIAG
#define MOVBA_SCHUR_EPW_DIAG 128
#endif
constexpr int kSchurWaves = MOVBA_SCHUR_WAVES;      // waves per k_schur workgroup
constexpr int kSchurWPI = MOVBA_SCHUR_WPI;          // waves that share one work item
constexpr int kSchurChunkDiag = MOVBA_SCHUR_EPW_DIAG * kSchurWPI;   // diagonal pairs (all edges of a keyframe, ~1.5x the work per entry) are cut finer
constexpr int kSchurChunk = MOVBA_SCHUR_EPW * kSchurWPI;       // entries per schur work item (one workgroup of 4 waves each)

// Returns MOVBA_OK / MOVBA_ERR_ARG / MOVBA_EMPTY.  build_structure = build_basic + pair counting + finish_pairs +
// entry filling, all on the host; the upload path normally runs only build_basic and finish_pairs on the host and
// leaves counting / filling to the device (struct_kernels.hip).
int build_structure(const movba_lba_desc& d, Structure& s);
// validation, grouping by point, hessian indices, pose-major slots.  With rank_out (E values) the slots are NOT built: every
// edge's rank among the edges of its keyframe, in caller order, is stored there instead, and pose_slot0 keeps the first slot
// of every free keyframe (-1: fixed or unobserved) — for edges already grouped by point, slot = pose_slot0[pose] + rank, which
// the upload path leaves to the device; build_slots(s) completes the job on the host otherwise.
int build_basic(const movba_lba_desc& d, Structure& s, int32_t* rank_out = nullptr);
void reset_structure(Structure& s, int NP, int P, int E);       // what build_basic starts with ...
void index_poses(const uint8_t* pose_fixed, Structure& s);      // ... and ends with (from s.pose_edges), for a grouping pass that ran on the device
void build_slots(Structure& s);                               // (s.gp must still be valid: the caller's arrays, or s.g_pose)
int finish_pairs(Structure& s, const int32_t* cnt);           // cnt[i*nfree+j] (i <= j) -> pairs, items, gather lists

// Covisibility ordering of the free keyframes.  The reference numbers its pose vertices by KeyFrame::mnId
// (/root/reference/src/Optimizer.cc:557-566), i.e. by time of creation; g2o's CSparse solver then reorders the reduced
// system itself (block AMD).  The PCG's aggregates (runs of consecutive block rows), its coarse modes (linear in the row
// index) and the schur launch schedule (by block row) want keyframes that share points to be neighbours in the numbering,
// which creation order gives only while the camera never comes back and ids are handed out along the path.  From the pair
// counts cnt[i*nf+j] (i <= j) this computes a reverse Cuthill-McKee order of the covisibility graph and says whether it is
// worth taking (its envelope is clearly smaller than the given order's); apply_pose_order then renumbers hessian indices,
// free-pose list, first slots and the counts.  Results do not depend on the numbering beyond rounding.
bool covisibility_order(int nf, const int32_t* cnt, std::vector<int32_t>& new_of_old);
void apply_pose_order(Structure& s, const std::vector<int32_t>& new_of_old, int32_t* cnt /* nf x nf, permuted in place; may be null */);

}  // namespace movba
