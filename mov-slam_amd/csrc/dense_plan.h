// Static schedule of the one-launch direct solver (dense_persist.hip): who owns which 48 x 48 tile of the reduced
// camera system, and in which order every workgroup runs its tasks.
//
// The reference factors S with a sparse direct solver whatever its pattern (LinearSolverCSparse,
// /root/reference/src/Optimizer.cc:535).  Here S is factored as a dense blocked Cholesky in ONE launch: every tile of the
// lower block triangle has an owner workgroup that keeps it in LDS for the whole factorisation and applies the updates
// of the block columns to its left as those columns are published (right-looking, owner computes); the finished tiles
// L(I, K) and the inverse factors W_K = L(K, K)^-T of the diagonal tiles are what travels between workgroups (write-through
// stores + one flag each; the substitutions' 48-vectors likewise, the one on the back substitution's chain as tagged records).  The order of a
// workgroup's tasks is a host-built list sorted by a key under which every task depends only on tasks of smaller keys:
// the globally smallest unfinished task can always run, so the schedule cannot deadlock while its workgroups are
// resident (and every device-side wait is bounded by a clock, dense_persist.hip).
// Pure C++ (no HIP): checked on the CPU by simulating the flags (tests/test_dense_plan_cpu.py).
#pragma once
#include <cstdint>
#include <vector>

namespace movba {

enum DenseOp : int32_t {
    DT_ASM = 0,     // tile (I, K) <- S (and, for a diagonal tile, the right-hand side row) from the schur partials
    DT_UPD = 1,     // tile (I, K) -= L(I, k) L(K, k)^T                      waits F(I, k), F(K, k)
    DT_DIAG = 2,    // block column 0's owner: D_0 -> W_0 = L(0, 0)^-T (sweep over the identity), published   sets PD(0)   (K >= 1: DT_COL)
    DT_OFF = 3,     // L(I, K) = tile (I, K) W_K (a matrix product)           waits PD(K), sets F(I, K)
    DT_RHS = 4,     // diagonal owner: r_K -= L(K, K-1) y_(K-1) (own tile: pad[0] = its slot), y_K = W_K^T r_K   waits FY(K-1), sets FY(K)
    DT_BSX = 5,     // diagonal owner: x_J = W_J (y_J - sum_I c(I, J)), then c(J, J-1) of its own sub-diagonal tile (pad[0] = its
                    // slot)                                               waits FC(I, J), I > J, sets FX(J), FC(J, J-1)
    DT_BSC = 6,     // owner of (I, J): c(I, J) = L(I, J)^T x_I               waits FX(I), sets FC(I, J)
    DT_EPI = 7,     // increments, computeScale's pose part, trial poses      waits FX(*)
    DT_UPD2 = 9,    // diagonal owner, block column k <= K - 2: D_K -= L(K, k) L(K, k)^T and tile (K, K-1) -= L(K, k) L(K-1, k)^T in one
                    // task (one wait, both operands fetched together, fifteen MFMA blocks over the four waves): slot = D_K's,
                    // pad[2] = the sub-diagonal tile's, pad[0] / pad[1] = own slot of L(K, k) / L(K-1, k) or -1       waits F(K, k), F(K-1, k)
    DT_COL = 10,    // diagonal owner K >= 1, block column K - 1 published: L(K, K-1) = tile W_(K-1), D_K -= L L^T, D_K -> W_K in ONE task
                    // (DT_OFF + the last DT_UPD + DT_DIAG of the owner without the task boundaries between them): slot = D_K's,
                    // pad[2] = the sub-diagonal tile's       waits PD(K-1), sets F(K, K-1) and PD(K)
    DT_RUP = 8,     // diagonal owner: r_K -= L(K, k) y_k, k <= K - 2 (ahead of DT_RHS, in the shadow of the factorisation)  waits FY(k)
};

struct DenseTask { int32_t op, slot, I, K, k, pad[3]; };       // 32 bytes; DT_UPD: pad[0] / pad[1] = own LDS slot of L(I, k) / L(K, k), -1 = fetch, -2 = still in
                                                                // the scratch tile from the task before; DT_BSC: pad[0] = 1: x_I is in this workgroup's LDS

constexpr int kDenseMaxSlots = 6;       // tiles a workgroup keeps in LDS (2 scratch tiles beside them: 150 KB)
constexpr int kDenseMaxGroups = 248;    // workgroups of the launch (one per CU, a few CUs to spare)

struct DensePlan {
    int nt = 0, G = 0, slots = 0;       // block columns, workgroups, tile slots per workgroup
    bool ok = false;                    // false: the system does not fit the one-launch solver (multi-launch path instead)
    std::vector<int32_t> task_ptr;      // G + 1
    std::vector<DenseTask> tasks;
    std::vector<int32_t> owner, slot;   // nt (nt + 1) / 2: tile (I, K) at I (I + 1) / 2 + K
};

// flags of the launch (one 32-bit word each, compared with the launch's epoch)
constexpr inline int dense_flag_F(int nt, int I, int K) { return I * nt + K; }                  // I in [0, nt]: row nt = FY
constexpr inline int dense_flag_PD(int nt, int K) { return (nt + 1) * nt + K; }
constexpr inline int dense_flag_FX(int nt, int J) { return (nt + 2) * nt + J; }
constexpr inline int dense_flag_FC(int nt, int I, int J) { return (nt + 3) * nt + I * nt + J; }
constexpr inline int dense_flag_count(int nt) { return (2 * nt + 3) * nt + 8; }
// (behind the flags and the two failure words, 16-byte aligned: the tagged sub-diagonal contributions, 4 words x 48 per block row)
constexpr inline int dense_ctag_word(int nt) { return (dense_flag_count(nt) + 8 + 3) / 4 * 4; }
constexpr inline int dense_flag_words(int nt) { return dense_ctag_word(nt) + nt * 48 * 4; }

void build_dense_plan(int nt, DensePlan &p, int max_groups = kDenseMaxGroups, int max_slots = kDenseMaxSlots);

}  // namespace movba
