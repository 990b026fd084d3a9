#include "structure.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace movba {

void reset_structure(Structure& s, int NP, int P, int E)
{
    // keep the vectors' capacity across calls (a handle solves window after window: fresh multi-MB allocations
    // would be paid in page faults every time)
    s.nfree = 0; s.npairs = 0; s.nitems = 0; s.max_degree = 0; s.nentries = 0; s.already_grouped = true; s.n_fixed = 0; s.n_agg = 0; s.reordered = false;
    s.free_pose.clear(); s.pair_i.clear(); s.pair_j.clear(); s.items.clear(); s.sched.clear(); s.sched_per_xcd = 0; s.row_ent.clear();
    s.cblk_g.clear(); s.cblk_h.clear(); s.cblk_ptr.clear(); s.cblk_ent.clear(); s.ent_i.clear(); s.ent_j.clear(); s.ent_l.clear(); s.E_free = 0;
    s.NP = NP; s.P = P; s.E = E;
}

// hessian indices, free-pose list and first pose-major slots from the edges per keyframe: free keyframes with at least one
// edge, in caller order (the tail of build_basic; the upload path runs it alone when the device has counted the edges)
void index_poses(const uint8_t* pose_fixed, Structure& s)
{
    const int NP = s.NP;
    s.hidx.assign(NP, -1);
    // pose-major slots: the edges of free pose h occupy [pstart[h], pstart[h+1]) in ascending map-point order, so the
    // schur pass reads the per-edge records of one keyframe as (nearly) contiguous memory
    std::vector<int32_t>& pstart = s.pose_slot0;
    pstart.assign(NP + 1, -1);
    int run = 0;
    for (int i = 0; i < NP; ++i) {
        if (pose_fixed[i]) { s.n_fixed++; continue; }
        if (s.pose_edges[i] > 0) { s.hidx[i] = s.nfree++; s.free_pose.push_back(i); pstart[i] = run; run += s.pose_edges[i]; }
    }
    s.E_free = run;
}

int build_basic(const movba_lba_desc& d, Structure& s, int32_t* rank_out)
{
    const int NP = d.n_poses, P = d.n_points, E = d.n_edges;
    if (NP < 0 || P < 0 || E < 0) return MOVBA_ERR_ARG;
    if ((NP && (!d.poses || !d.pose_fixed)) || (P && !d.points)) return MOVBA_ERR_ARG;
    if (E && (!d.edge_pose || !d.edge_point || !d.obs || !d.inv_sigma2)) return MOVBA_ERR_ARG;
    reset_structure(s, NP, P, E);

    // active vertices = those with >= 1 edge (SparseOptimizer::initializeOptimization).  ONE pass over the caller's
    // edges: validation, edges per pose, edges per point, and whether the edges already come grouped by map point.
    s.pose_edges.assign(NP + 1, 0);
    s.pt_start.assign(P + 2, 0);
    {
        // Grouped edges (the expected case) need no counting: the last edge of point l leaves its end offset in slot l (plain
        // stores, no chain of dependent increments of one counter), and a running maximum fills the points nobody observes.
        bool grouped = true;
        int prev_l = -1;
        int32_t *pe = s.pose_edges.data(), *pend = s.pt_start.data() + 1;
        const int32_t *ep = d.edge_pose, *el = d.edge_point;
        for (int e = 0; e < E; ++e) {
            const int ip = ep[e], l = el[e];
            if ((unsigned)ip >= (unsigned)NP || (unsigned)l >= (unsigned)P) return MOVBA_ERR_ARG;
            const int r = pe[ip]++;
            if (rank_out) rank_out[e] = r;    // (the edge's rank among its keyframe's edges: slot = first slot of the keyframe + rank)
            pend[l] = e + 1;
            grouped &= l >= prev_l;
            prev_l = l;
        }
        s.already_grouped = grouped;      // identity permutation iff the caller's edges are in ascending point order
    }
    if (s.already_grouped) {
        for (int l = 0; l < P; ++l) {
            const int b = s.pt_start[l], en = std::max(s.pt_start[l + 1], b);
            s.pt_start[l + 1] = en;
            s.max_degree = std::max(s.max_degree, en - b);
        }
        s.pt_start.pop_back();            // size P + 1
    } else {
        // counting sort by map point: counts accumulated one slot late, so that the scatter below can advance them in place
        std::fill(s.pt_start.begin(), s.pt_start.end(), 0);
        for (int e = 0; e < E; ++e) s.pt_start[d.edge_point[e] + 2]++;
        for (int l = 0; l < P; ++l) { s.max_degree = std::max(s.max_degree, s.pt_start[l + 2]); s.pt_start[l + 2] += s.pt_start[l + 1]; }
    }
    if (s.already_grouped) {
        // the reference's own edge order (map points in list order, Optimizer.cc:623-672): identity permutation
        // (perm stays empty: the upload path copies the per-edge arrays as they are)
        s.perm.clear();
        // (no copy either: gp / gl alias the caller's arrays, valid for the duration of the upload call, which is their only use)
        s.gp = d.edge_pose; s.gl = d.edge_point;
    } else {
        s.perm.resize(E);
        s.g_pose.resize(E); s.g_point.resize(E);
        for (int e = 0; e < E; ++e) {
            const int pos = s.pt_start[d.edge_point[e] + 1]++;
            s.perm[pos] = e;
        }
        s.pt_start.pop_back();      // now pt_start[l]..pt_start[l+1], size P+1
        for (int g = 0; g < E; ++g) {
            s.g_pose[g] = d.edge_pose[s.perm[g]];
            s.g_point[g] = d.edge_point[s.perm[g]];
        }
        s.gp = s.g_pose.data(); s.gl = s.g_point.data();
    }
    index_poses(d.pose_fixed, s);
    if (!rank_out) build_slots(s);
    if (E == 0) return MOVBA_EMPTY;
    return MOVBA_OK;
}

void build_slots(Structure& s)
{
    s.slot.resize(s.E);
    const int32_t *gp = s.gp;
    int32_t *sl = s.slot.data(), *ps = s.pose_slot0.data();
    for (int g = 0; g < s.E; ++g) { const int i = gp[g]; const int v = ps[i]; sl[g] = v; ps[i] = v + (v >= 0); }
}

// free observers of every point, flattened: (hessian index, grouped edge), ascending hessian index
static int free_lists(const Structure& s, std::vector<int32_t>& fe_start, std::vector<int32_t>& fe_h, std::vector<int32_t>& fe_g)
{
    const int P = s.P;
    fe_start.assign(P + 1, 0); fe_h.clear(); fe_g.clear();
    fe_h.reserve(s.E); fe_g.reserve(s.E);
    for (int l = 0; l < P; ++l) {
        const size_t base = fe_h.size();
        for (int g = s.pt_start[l]; g < s.pt_start[l + 1]; ++g) {
            const int h = s.hidx[s.gp[g]];
            if (h < 0) continue;
            // insertion keeps the (usually already ascending) list sorted and stable
            size_t pos = fe_h.size();
            fe_h.push_back(h); fe_g.push_back(g);
            while (pos > base && fe_h[pos - 1] > h) {
                fe_h[pos] = fe_h[pos - 1]; fe_g[pos] = fe_g[pos - 1];
                --pos;
            }
            fe_h[pos] = h; fe_g[pos] = g;
            if (pos > base && fe_h[pos - 1] == h) return MOVBA_ERR_ARG;          // duplicate observation
            if (pos + 1 < fe_h.size() && fe_h[pos + 1] == h) return MOVBA_ERR_ARG;
        }
        fe_start[l + 1] = (int32_t)fe_h.size();
    }
    return MOVBA_OK;
}

int finish_pairs(Structure& s, const int32_t* cnt)
{
    const int nf = s.nfree;
    // pair ids: the nf diagonal pairs first (pair k == (k,k)), then off-diagonal row-major
    s.pid.assign((size_t)nf * (size_t)nf, -1);
    s.pair_i.clear(); s.pair_j.clear();
    for (int i = 0; i < nf; ++i) { s.pid[(size_t)i * nf + i] = i; s.pair_i.push_back(i); s.pair_j.push_back(i); }
    for (int i = 0; i < nf; ++i)
        for (int j = i + 1; j < nf; ++j)
            if (cnt[(size_t)i * nf + j] > 0) {
                s.pid[(size_t)i * nf + j] = (int32_t)s.pair_i.size();
                s.pair_i.push_back(i); s.pair_j.push_back(j);
            }
    s.npairs = (int)s.pair_i.size();
    s.pair_ptr.assign(s.npairs + 1, 0);
    for (int p = 0; p < s.npairs; ++p) s.pair_ptr[p + 1] = s.pair_ptr[p] + cnt[(size_t)s.pair_i[p] * nf + s.pair_j[p]];
    s.nentries = s.pair_ptr[s.npairs];
    if (s.nentries > (int64_t)0x7fffffff) return MOVBA_ERR_ARG;
    // the diagonal pair of a free pose lists each of its edges once, in slot order: diagonal entry k is slot k (structure.h)
    if (nf > 0 && s.pair_ptr[nf] != (int64_t)s.E_free) return MOVBA_ERR_ARG;

    // ---- work items: chunks of a pair's entries ----
    s.items.clear();
    s.pair_item_start.assign(s.npairs + 1, 0);
    for (int p = 0; p < s.npairs; ++p) {
        // a pair that does not fit one work item is cut into equal parts (not full chunks plus a small remainder)
        const int64_t cap = p < nf ? kSchurChunkDiag : kSchurChunk, len = s.pair_ptr[p + 1] - s.pair_ptr[p];
        const int64_t parts = (len + cap - 1) / cap;
        const int64_t chunk = parts > 0 ? ((len + parts - 1) / parts + 3) / 4 * 4 : cap;      // multiple of 4: whole entries per wave
        s.pair_item_start[p] = (int32_t)s.items.size();
        for (int64_t b = s.pair_ptr[p]; b < s.pair_ptr[p + 1]; b += chunk)
            s.items.push_back(Item{ p, (int32_t)b, (int32_t)std::min<int64_t>(b + chunk, s.pair_ptr[p + 1]), p < nf ? 1 : 0 });
    }
    s.pair_item_start[s.npairs] = (int32_t)s.items.size();
    s.nitems = (int)s.items.size();

    // ---- launch schedule of k_schur: the MI355X dispatcher deals consecutive workgroups round-robin to the 8 XCDs,
    // each with its own L2.  Items are ordered by block row (the pairs (i, *) gather the points keyframe i sees, so
    // a contiguous run of rows shares its map points in one L2) and cut into 8 runs of equal estimated work, heavy
    // items first inside a run; the diagonal items (Hpp, b and the Schur diagonal: ~1.5x the work per entry) are
    // thereby spread over all XCDs instead of filling the first two. ----
    // (this function sits between the device's pair counts and the solve's first launch: no allocations, no general sorts)
    {
        const int ipw = kSchurWaves / kSchurWPI;
        // items by block row, in item order inside a row: the diagonal pair's items, then those of the row's off-diagonal
        // pairs (pairs are numbered diagonal first, then off-diagonal row-major, and items follow their pairs)
        std::vector<int32_t>& order = s.tmp_order;
        order.resize(s.nitems);
        {
            int n = 0, od = nf < s.npairs ? s.pair_item_start[nf] : s.nitems;
            for (int i = 0; i < nf; ++i) {
                for (int k = s.pair_item_start[i]; k < s.pair_item_start[i + 1]; ++k) order[n++] = k;
                while (od < s.nitems && s.pair_i[s.items[od].pair] == i) order[n++] = od++;
            }
        }
        auto weight = [&](int k) { const Item& it = s.items[k]; return (int64_t)(it.end - it.begin) * (it.diag ? 3 : 2) + 128; };
        int64_t total = 0;
        for (int k = 0; k < s.nitems; ++k) total += weight(k);
        // 8 runs of equal estimated work; inside a run the heavy items first (ties in row order: the keys are unique)
        std::vector<uint64_t>& key = s.tmp_key;
        key.resize(s.nitems);
        int seg_begin[9];
        {
            int64_t acc = 0; int x = 0;
            seg_begin[0] = 0;
            for (int n = 0; n < s.nitems; ++n) {
                while (x < 7 && acc >= (total * (x + 1)) / 8) seg_begin[++x] = n;
                const int64_t wgt = weight(order[n]);
                key[n] = ((uint64_t)(0x7fffffff - wgt) << 32) | (uint32_t)n;
                acc += wgt;
            }
            while (x < 8) seg_begin[++x] = s.nitems;
        }
        // Waves per item: a wave takes 128 entries per turn of its loop (64 lanes x 2 in flight), so an item of up to 128 entries
        // is ONE wave's single turn, up to 256 two waves', beyond that four waves share it (diagonal items always: they are cut
        // at 512 entries and cost 1.5x per entry).  Workgroups of four waves are filled in schedule order - four small items, two
        // medium ones, or one large - so that a 50-keyframe window's ~630 items need ~400 workgroups: ONE round on the 512 the
        // chip holds at 232 registers per lane, where every item its own workgroup left 120 tiny items to a second round that
        // started 7 - 9 us in and ended the pass at 14.5 us.
#ifndef MOVBA_SCHUR_T1
#define MOVBA_SCHUR_T1 128
#define MOVBA_SCHUR_T2 256
#endif
#ifndef MOVBA_SCHUR_DW
#define MOVBA_SCHUR_DW 4
#endif
        auto waves_of = [&](int k) { const Item& it = s.items[k]; const int n = it.end - it.begin; return it.diag ? MOVBA_SCHUR_DW : (n > MOVBA_SCHUR_T2 ? 4 : (n > MOVBA_SCHUR_T1 ? 2 : 1)); };
        size_t longest = 1;
        for (int g = 0; g < 8; ++g) {
            std::sort(key.begin() + seg_begin[g], key.begin() + seg_begin[g + 1]);
            size_t slots = 0;
            for (int n = seg_begin[g]; n < seg_begin[g + 1]; ++n) {
                const size_t nw = (size_t)waves_of(order[(uint32_t)key[n]]);
                if ((slots & 3) + nw > 4) slots = (slots + 3) & ~(size_t)3;     // does not fit the open workgroup: the next one
                slots += nw;
            }
            longest = std::max(longest, (slots + 3) & ~(size_t)3);
        }
        (void)ipw;
        s.sched_per_xcd = (int)longest;                     // wave slots per XCD (a multiple of 4)
        s.sched.assign((size_t)8 * s.sched_per_xcd, SchedItem{ 0, 0, -1, 0, 0, -1, -1, 0 });
        for (int g = 0; g < 8; ++g) {
            size_t slots = 0;
            for (int n = seg_begin[g]; n < seg_begin[g + 1]; ++n) {
                const int item = order[(uint32_t)key[n]];
                const Item& it = s.items[item];
                const int nw = waves_of(item);
                if ((slots & 3) + (size_t)nw > 4) slots = (slots + 3) & ~(size_t)3;
                const int len = it.end - it.begin, epw = (len + nw - 1) / nw;          // the item's entries split evenly over its waves
                for (int sub = 0; sub < nw; ++sub) {
                    const int wb = std::min(it.begin + sub * epw, it.end), we = std::min(wb + epw, it.end);
                    s.sched[(size_t)g * s.sched_per_xcd + slots + sub] = SchedItem{ wb, we, (item << 1) | (it.diag ? 1 : 0),
                                                                                   s.free_pose[s.pair_i[it.pair]], s.free_pose[s.pair_j[it.pair]], -1, -1, sub | (nw << 8) };
                }
                slots += (size_t)nw;
            }
        }
    }

    // ---- block-row gather lists for y = S x with S given by its upper blocks: row i lists, by ascending column, the
    // transposed blocks (j, i), j < i, then the diagonal block, then the blocks (i, j), j > i; padded to an even length
    // (lists are consumed in pairs; block -1 = zero block).  The off-diagonal pairs are numbered row-major, so walking them
    // once per part fills every row in column order. ----
    {
        s.row_ptr.assign(nf + 1, 0);
        for (int p = 0; p < s.npairs; ++p) {
            s.row_ptr[s.pair_i[p] + 1]++;
            if (p >= nf) s.row_ptr[s.pair_j[p] + 1]++;
        }
        for (int i = 0; i < nf; ++i) s.row_ptr[i + 1] = s.row_ptr[i] + ((s.row_ptr[i + 1] + 1) & ~1);
        s.row_ent.assign((size_t)s.row_ptr[nf], RowEnt{ -1, 0, 0, 0 });
        std::vector<int32_t>& cur = s.tmp_order;
        cur.assign(s.row_ptr.begin(), s.row_ptr.begin() + nf);
        for (int p = nf; p < s.npairs; ++p) s.row_ent[cur[s.pair_j[p]]++] = RowEnt{ p, s.pair_i[p], 1, 0 };
        for (int i = 0; i < nf; ++i) s.row_ent[cur[i]++] = RowEnt{ i, i, 0, 0 };
        for (int p = nf; p < s.npairs; ++p) s.row_ent[cur[s.pair_i[p]]++] = RowEnt{ p, s.pair_j[p], 0, 0 };
    }
    return MOVBA_OK;
}

bool covisibility_order(int nf, const int32_t* cnt, std::vector<int32_t>& new_of_old)
{
    new_of_old.resize(nf);
    for (int i = 0; i < nf; ++i) new_of_old[i] = i;
    if (nf < 3) return false;
    auto linked = [&](int a, int b) { return a == b ? false : (a < b ? cnt[(size_t)a * nf + b] : cnt[(size_t)b * nf + a]) > 0; };
    std::vector<int32_t> deg(nf, 0);
    for (int i = 0; i < nf; ++i) for (int j = i + 1; j < nf; ++j) if (cnt[(size_t)i * nf + j] > 0) { deg[i]++; deg[j]++; }
    // envelope of an order: sum over rows of (row - first linked column)
    auto envelope = [&](const std::vector<int32_t>& old_of_new) {
        int64_t env = 0;
        for (int r = 0; r < nf; ++r) {
            int first = r;
            for (int c = 0; c < r; ++c) if (linked(old_of_new[r], old_of_new[c])) { first = c; break; }
            env += r - first;
        }
        return env;
    };
    // Cuthill-McKee from a pseudo-peripheral keyframe of every component (smallest degree, then two sweeps to the farthest
    // level's smallest-degree keyframe), neighbours by ascending (degree, index); reversed at the end
    std::vector<int32_t> order, level(nf), queue;
    std::vector<uint8_t> placed(nf, 0);
    auto bfs = [&](int root, std::vector<int32_t>& out) {           // over unplaced keyframes only
        out.clear(); out.push_back(root);
        std::fill(level.begin(), level.end(), -1); level[root] = 0;
        for (size_t h = 0; h < out.size(); ++h) {
            const int v = out[h];
            const size_t first = out.size();
            for (int u = 0; u < nf; ++u) if (!placed[u] && level[u] < 0 && linked(v, u)) { level[u] = level[v] + 1; out.push_back(u); }
            std::sort(out.begin() + first, out.end(), [&](int a, int b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
        }
        return level[out.back()];
    };
    order.reserve(nf);
    for (;;) {
        int root = -1;
        for (int i = 0; i < nf; ++i) if (!placed[i] && (root < 0 || deg[i] < deg[root])) root = i;
        if (root < 0) break;
        for (int sweep = 0; sweep < 2; ++sweep) {
            const int depth = bfs(root, queue);
            int far = -1;
            for (int v : queue) if (level[v] == depth && (far < 0 || deg[v] < deg[far])) far = v;
            if (far == root) break;
            root = far;
        }
        bfs(root, queue);
        for (int v : queue) { placed[v] = 1; order.push_back(v); }
    }
    std::reverse(order.begin(), order.end());
    std::vector<int32_t> natural(nf);
    for (int i = 0; i < nf; ++i) natural[i] = i;
    const int64_t env_nat = envelope(natural), env_rcm = envelope(order);
    // taken only when clearly better: a window already numbered along its path keeps its numbering (and its device-side
    // structure pass as it ran)
    if (!(env_rcm * 5 < env_nat * 4)) return false;
    for (int k = 0; k < nf; ++k) new_of_old[order[k]] = k;
    return true;
}

void apply_pose_order(Structure& s, const std::vector<int32_t>& new_of_old, int32_t* cnt)
{
    const int nf = s.nfree;
    std::vector<int32_t> fp(nf);
    for (int h = 0; h < nf; ++h) fp[new_of_old[h]] = s.free_pose[h];
    s.free_pose = fp;
    for (int h = 0; h < nf; ++h) s.hidx[s.free_pose[h]] = h;
    // pose-major slots follow the hessian order: the diagonal pair of keyframe h lists its edges at its first slot
    int run = 0;
    for (int h = 0; h < nf; ++h) { s.pose_slot0[s.free_pose[h]] = run; run += s.pose_edges[s.free_pose[h]]; }
    if (cnt) {
        std::vector<int32_t> c2((size_t)nf * nf, 0);
        for (int i = 0; i < nf; ++i)
            for (int j = i; j < nf; ++j) {
                const int a = new_of_old[i], b = new_of_old[j];
                c2[(size_t)std::min(a, b) * nf + std::max(a, b)] = cnt[(size_t)i * nf + j];
            }
        std::memcpy(cnt, c2.data(), sizeof(int32_t) * c2.size());
    }
}

int build_structure(const movba_lba_desc& d, Structure& s)
{
    const int rc = build_basic(d, s);
    if (rc != MOVBA_OK) return rc;
    const int P = s.P, nf = s.nfree;

    // ---- pose pairs sharing a point (block pattern of the reduced system), counted and filled on the host ----
    std::vector<int32_t> fe_start, fe_h, fe_g;
    const int rf = free_lists(s, fe_start, fe_h, fe_g);
    if (rf != MOVBA_OK) return rf;
    std::vector<int32_t> cnt((size_t)nf * (size_t)nf, 0);       // upper triangle used
    for (int l = 0; l < P; ++l)
        for (int a = fe_start[l]; a < fe_start[l + 1]; ++a) {
            int32_t *row = &cnt[(size_t)fe_h[a] * nf];
            for (int b = a; b < fe_start[l + 1]; ++b) row[fe_h[b]]++;
        }
    if (nf <= 80 && !s.no_reorder) {
        std::vector<int32_t> new_of_old;
        if (covisibility_order(nf, cnt.data(), new_of_old)) {
            // renumber, then slots, observer lists and counts once more in the new numbering
            apply_pose_order(s, new_of_old, nullptr);
            build_slots(s);
            const int rf2 = free_lists(s, fe_start, fe_h, fe_g);
            if (rf2 != MOVBA_OK) return rf2;
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int l = 0; l < P; ++l)
                for (int a = fe_start[l]; a < fe_start[l + 1]; ++a) {
                    int32_t *row = &cnt[(size_t)fe_h[a] * nf];
                    for (int b = a; b < fe_start[l + 1]; ++b) row[fe_h[b]]++;
                }
            s.reordered = true;
        }
    }
    const int rp = finish_pairs(s, cnt.data());
    if (rp != MOVBA_OK) return rp;
    // off-diagonal entries only: diagonal entry k is slot k (structure.h)
    if (s.pair_ptr[nf] != s.E_free) return MOVBA_ERR_ARG;
    const size_t noff = (size_t)(s.nentries - s.E_free);
    s.ent_i.resize(noff); s.ent_j.resize(noff); s.ent_l.resize(noff);
    {
        std::vector<int32_t> cur(s.npairs);
        for (int p = 0; p < s.npairs; ++p) cur[p] = (int32_t)(s.pair_ptr[p] - s.E_free);
        for (int l = 0; l < P; ++l)
            for (int a = fe_start[l]; a < fe_start[l + 1]; ++a) {
                const int32_t *prow = &s.pid[(size_t)fe_h[a] * nf];
                const int32_t ga = fe_g[a];
                for (int b = a + 1; b < fe_start[l + 1]; ++b) {
                    const int32_t pos = cur[prow[fe_h[b]]]++;
                    s.ent_i[pos] = s.slot[ga]; s.ent_j[pos] = s.slot[fe_g[b]]; s.ent_l[pos] = l;
                }
            }
    }
    return MOVBA_OK;
}

void build_coarse(Structure& s, const int32_t* agg_row0, int n_agg)
{
    const int nf = s.nfree, G = n_agg;
    s.n_agg = G;
    s.cblk_g.clear(); s.cblk_h.clear(); s.cblk_ent.clear(); s.cblk_ij.clear(); s.cblk_ptr.assign(1, 0);
    s.multi_pairs.clear();
    for (int p = nf; p < s.npairs; ++p)
        if (s.pair_item_start[p + 1] - s.pair_item_start[p] > 1) s.multi_pairs.push_back(p);
    std::vector<int32_t> agg_of(nf, 0);
    for (int g = 0; g < G; ++g)
        for (int i = agg_row0[g]; i < agg_row0[g + 1]; ++i) agg_of[i] = g;
    std::vector<std::vector<int32_t>> lists((size_t)G * G), ijs((size_t)G * G);
    for (int i = 0; i < nf; ++i)
        for (int e = s.row_ptr[i]; e < s.row_ptr[i + 1]; ++e)
            if (s.row_ent[e].block >= 0) {
                const size_t b = (size_t)agg_of[i] * G + agg_of[s.row_ent[e].col];
                // term source: a pair held in ONE work item is read straight from that item's partial (value = -partial);
                // diagonal pairs and pairs cut into several items come from the materialised block (coarse_level.h)
                const int pr = s.row_ent[e].block;
                const bool single = pr >= nf && s.pair_item_start[pr + 1] - s.pair_item_start[pr] == 1;
                const int src_idx = single ? s.pair_item_start[pr] : pr;
                lists[b].push_back((src_idx << 2) | (s.row_ent[e].transposed ? 2 : 0) | (single ? 1 : 0));
                ijs[b].push_back((i << 16) | s.row_ent[e].col);
            }
    for (int g = 0; g < G; ++g)
        for (int h = 0; h < G; ++h) {
            const std::vector<int32_t>& l = lists[(size_t)g * G + h];
            if (l.empty()) continue;
            s.cblk_g.push_back(g); s.cblk_h.push_back(h);
            s.cblk_ent.insert(s.cblk_ent.end(), l.begin(), l.end());
            s.cblk_ij.insert(s.cblk_ij.end(), ijs[(size_t)g * G + h].begin(), ijs[(size_t)g * G + h].end());
            s.cblk_ptr.push_back((int32_t)s.cblk_ent.size());
        }
}

}  // namespace movba
