#include "dense_plan.h"

#include <algorithm>
#include <queue>
#include <utility>

namespace movba {

namespace {

struct Keyed { uint64_t key; DenseTask t; };

inline uint64_t mk_key(int stage, int cls, int a, int b)
{
    return ((uint64_t)(uint32_t)stage << 40) | ((uint64_t)(uint32_t)cls << 32) | ((uint64_t)(uint32_t)a << 16) | (uint64_t)(uint32_t)b;
}

}  // namespace

// Ownership: workgroup K owns the diagonal tile (K, K) and the tile to its left, (K, K - 1): the chain
//     D_(K-1) published -> L(K, K-1) -> D_K -= L(K, K-1) L(K, K-1)^T -> D_K published
// then crosses workgroups once per block column.  The other tiles go, heaviest first (a tile of block column K takes K
// updates), to the least loaded workgroup with a free slot.
// Task order inside a workgroup: ascending key (stage, class, ...) with stage = block column + 1 for the factorisation and
// nt + 1 + (nt - 1 - J) for the back substitution of block row J; every dependency points to a smaller key (dense_plan.h).
// Inside a stage the keys follow the critical path  L(K, K-1) published -> [workgroup K+1] tile (K+1, K) -= L(K+1, K-1) L(K, K-1)^T
// -> L(K+1, K):
//   * the diagonal owner applies block column k to its diagonal tile and to its sub-diagonal tile in ONE task (DT_UPD2: both
//     operands arrive together, L(K-1, k) from the chain's workgroup): one wait, one round of fetches, fifteen MFMA blocks;
//   * consecutive updates that share an operand keep it in the scratch tile (pad = -2);
//   * D_K is published straight behind its last update, ahead of the same stage's updates of the workgroup's other tiles;
//   * the right-hand side row takes the block columns up to K - 2 one stage behind their y_k (DT_RUP, where the workgroup
//     would otherwise wait): DT_RHS is left with the last column and the factorisation of D_K.
void build_dense_plan(int nt, DensePlan &p, int max_groups, int max_slots)
{
    p = DensePlan{};
    p.nt = nt;
    if (nt < 1 || nt > max_groups || max_slots < 2) return;
    const int ntiles = nt * (nt + 1) / 2;
    p.owner.assign(ntiles, -1); p.slot.assign(ntiles, -1);
    auto tix = [](int I, int K) { return I * (I + 1) / 2 + K; };
    const int n_other = ntiles - nt - (nt - 1);
    const int G = std::min(max_groups, nt + std::max(n_other, 0));
    std::vector<double> load(G, 0.0);
    std::vector<int> used(G, 0);
    for (int K = 0; K < nt; ++K) {
        p.owner[tix(K, K)] = K; p.slot[tix(K, K)] = used[K]++;
        load[K] += K + 8.0 + 0.5 * K;                       // updates, local factorisation, right-hand side row
        if (K >= 1) { p.owner[tix(K, K - 1)] = K; p.slot[tix(K, K - 1)] = used[K]++; load[K] += (K - 1) + 6.5; }
    }
    {
        // heaviest first: block column descending, then the rows nearest the diagonal (needed soonest)
        typedef std::pair<double, int> LW;
        std::priority_queue<LW, std::vector<LW>, std::greater<LW>> heap;
        for (int g = 0; g < G; ++g) if (used[g] < max_slots) heap.push(LW(load[g], g));
        for (int K = nt - 2; K >= 0; --K)
            for (int I = K + 2; I < nt; ++I) {
                if (heap.empty()) return;                   // more tiles than slots: not this solver's size
                const int g = heap.top().second;
                heap.pop();
                p.owner[tix(I, K)] = g; p.slot[tix(I, K)] = used[g]++;
                load[g] += K + 6.5;
                if (used[g] < max_slots) heap.push(LW(load[g], g));
            }
    }
    p.G = G;
    p.slots = *std::max_element(used.begin(), used.end());

    std::vector<std::vector<Keyed>> per(G);
    auto add = [&](int g, uint64_t key, int op, int slot, int I, int K, int k) { per[g].push_back(Keyed{ key, DenseTask{ op, slot, I, K, k, { -1, -1, 0 } } }); };
    // an update reads its operand tiles L(I, k), L(K, k) from the workgroup's own LDS slots when it owns them (pad[0], pad[1])
    auto own_slot = [&](int g, int I, int K) { return p.owner[tix(I, K)] == g ? p.slot[tix(I, K)] : -1; };
    for (int K = 0; K < nt; ++K)
        for (int I = K; I < nt; ++I) {
            const int g = p.owner[tix(I, K)], s = p.slot[tix(I, K)];
            add(g, mk_key(0, 0, K, I), DT_ASM, s, I, K, 0);
            for (int k = 0; k < K; ++k) {
                if (I == K + 1 && k <= K - 1 && p.owner[tix(I, I)] == g) continue;     // (the sub-diagonal tile: updated by the owner's DT_UPD2 tasks)
                if (I == K && k <= K - 2) {
                    add(g, mk_key(k + 1, 3, K - 1 - k, 0), DT_UPD2, s, K, K, k);
                    DenseTask &t = per[g].back().t;
                    t.pad[0] = own_slot(g, K, k); t.pad[1] = own_slot(g, K - 1, k); t.pad[2] = p.slot[tix(K, K - 1)];
                    continue;
                }
                if (I == K && k == K - 1) continue;         // (the diagonal tile's last update: part of the owner's DT_COL)
                add(g, I == K ? mk_key(k + 1, 3, K - 1 - k, 0) : mk_key(k + 1, 3, K - k, I - K), DT_UPD, s, I, K, k);
                per[g].back().t.pad[0] = own_slot(g, I, k); per[g].back().t.pad[1] = own_slot(g, K, k);
            }
            if (I == K) {
                if (K == 0) add(g, mk_key(0, 3, 0, 1), DT_DIAG, s, 0, 0, 0);
                else {
                    // the owner's own chain of block column K - 1 in one task, where its DT_OFF stood
                    add(g, mk_key(K, 1, 1, 0), DT_COL, s, K, K, 0);
                    per[g].back().t.pad[2] = p.slot[tix(K, K - 1)];
                }
                // (behind the pair of updates of block column k + 1, where the workgroup would wait for column k + 2; not behind
                //  the pair of column K - 2, the one in front of L(K, K-1): those two follow D_K's publication)
                for (int k = 0; k + 2 <= K; ++k) add(g, k + 4 <= K ? mk_key(k + 2, 3, K - 2 - k, 2) : mk_key(K, 3, 0, 2 + k), DT_RUP, s, K, K, k);
                add(g, mk_key(K + 1, 2, 0, 0), DT_RHS, s, K, K, 0);
                per[g].back().t.pad[0] = K >= 1 ? p.slot[tix(K, K - 1)] : -1;
                add(g, mk_key(nt + 1 + (nt - 1 - K), 0, 0, 0), DT_BSX, s, K, K, 0);
                per[g].back().t.pad[0] = K >= 1 ? p.slot[tix(K, K - 1)] : -1;      // (x_K's solve also takes c(K, K-1): the owner's own tile)
            } else {
                if (!(I == K + 1 && p.owner[tix(I, I)] == g)) add(g, mk_key(K + 1, 1, I - K, 0), DT_OFF, s, I, K, 0);       // (else: the owner's DT_COL)
                if (I == K + 1 && p.owner[tix(I, I)] == g) continue;           // (the sub-diagonal tile's contribution: inside the owner's DT_BSX)
                add(g, mk_key(nt + 1 + (nt - 1 - I), 1, I - K, 0), DT_BSC, s, I, K, 0);
                per[g].back().t.pad[0] = p.owner[tix(I, I)] == g ? 1 : 0;      // x_I was solved by this workgroup: still in its LDS
            }
        }
    add(p.owner[tix(0, 0)], mk_key(2 * nt + 1, 0, 0, 0), DT_EPI, 0, 0, 0, 0);
    p.task_ptr.assign(G + 1, 0);
    for (int g = 0; g < G; ++g) {
        std::sort(per[g].begin(), per[g].end(), [](const Keyed &a, const Keyed &b) { return a.key < b.key; });
        p.task_ptr[g + 1] = p.task_ptr[g] + (int32_t)per[g].size();
        // an operand the update before fetched into the same scratch tile is not fetched again
        for (size_t q = 1; q < per[g].size(); ++q) {
            const DenseTask &a = per[g][q - 1].t;
            DenseTask &b = per[g][q].t;
            if (a.op != DT_UPD || b.op != DT_UPD || a.k != b.k) continue;
            if (a.I == b.I && a.pad[0] < 0 && b.pad[0] == -1) b.pad[0] = -2;
            if (a.I != a.K && b.I != b.K && a.K == b.K && a.pad[1] < 0 && b.pad[1] == -1) b.pad[1] = -2;
        }
    }
    p.tasks.reserve(p.task_ptr[G]);
    for (int g = 0; g < G; ++g) for (const Keyed &kt : per[g]) p.tasks.push_back(kt.t);
    p.ok = true;
}

}  // namespace movba
