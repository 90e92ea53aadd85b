// lcqp_sparse_general.hpp -- symbolic analysis of the GENERAL sparse LDL' of the sparse arm (round 6): KKT patterns that are neither banded nor
// bordered (a 2-D grid, an unstructured mesh) -- what the reference's OSQP arm factorises with QDLDL whatever the pattern
// (src/SubsolverOSQP.cpp:136-152, osqp_setup; src/LCQProblem.cpp:390-441, 629-723).  Host only, once per pattern; no HIP in this file.
//
// Method: multifrontal LDL' over a nested-dissection tree with DENSE fronts.
//   * Ordering: nested dissection by breadth-first level structures (George 1973): a region larger than `leaf` nodes is cut by the level of
//     a breadth-first search from a pseudo-peripheral node that balances the two sides; both sides are ordered first, the separator last;
//     a region of at most `leaf` nodes is one leaf.  Disconnected regions are handled component by component.
//   * Fronts: every leaf and every separator is ONE front: its nodes are the pivots (contiguous positions of the ordering), its update rows
//     are the boundary of the region it closes -- the later nodes adjacent to the region --, taken as dense (the fill of eliminating a connected
//     region IS the clique on its boundary; a few explicit zeros stand where a separator node does not touch every boundary node).  The
//     front tree is the dissection tree; fronts are numbered in postorder, children before parents.
//   * Numeric (device: sp_general_factor / sp_general_solve in lcqp_sparse.hip; CPU restatement of the same loops for the tests:
//     tests/cpp/general_ldl_test.cpp): a front F (ff x ff, ff = np + nb) is zeroed, takes the entries of K whose column is one of its pivots
//     (asm lists below; entries of E are gated by the working set by VALUE, the pattern never changes), takes the update blocks of its
//     children (extend-add through `rel`), eliminates its np pivots (right-looking, no pivoting: the matrix is quasi-definite for
//     delta, delta2 > 0 and any symmetric permutation of a quasi-definite matrix factorises), stores the panel L (ff x np, column-major) and
//     1/D, and leaves its update block (nb x nb) on a stack whose offsets are fixed here.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <vector>

namespace lcqp_general {

struct Symbolic {
    int N = 0, nF = 0;
    std::vector<int> perm, iperm;             // perm[position] = node, iperm[node] = position
    // per front (postorder)
    std::vector<int> piv0, np, nb, parent;    // first pivot position, pivots, boundary rows, parent front (-1: a root)
    std::vector<int> rowPtr, rows;            // boundary rows of front f: positions rows[rowPtr[f] .. rowPtr[f+1]), ascending, all behind its pivots
    std::vector<int> childPtr, child;         // children of front f
    std::vector<int> rel;                     // rel[rowPtr[f] + a]: local index of boundary row a of front f in its PARENT's front (pivots 0 .. np-1, then boundary)
    std::vector<long long> Loff, CBoff;       // panel of front f in the factor storage (ff x np doubles, column-major, ld = ff); its update block on the stack (nb x nb, ld = nb)
    // assembly: the entries of K whose column (the earlier of the two positions) is a pivot of front f
    std::vector<int> asmPtr, asmSrc, asmGate, asmPos;   // src: k < nnzQ entry k of Q, else nnzQ + k entry k of E (CSR order); gate: row of E or -1; pos: i + ff * j (local row, local column)
    long long Lsize = 0, stackSize = 0;
    int maxFront = 0;
    long long flops = 0, nnzL = 0;
};

// adj: adjacency lists of the KKT graph on N = n + m nodes (sorted, no self loops).  Q given as both triangles (CSC = CSR), E in CSR.
// mergeFront: a front absorbs its LAST child (whose pivots lie right in front of its own) while the merged front stays within this many rows --
// the small separators at the bottom of the dissection tree then ride in their parents' fronts instead of costing a front of their own
// (a front costs a fixed ~20 us in a factorisation and ~5 us per sweep whatever its size; 0: no merging)
inline Symbolic analyze(int n, int m, const std::vector<std::vector<int>>& adj, const int* Qp, const int* Qi, const int* Ep, const int* Ei, int leaf = 32, int mergeFront = 64)
{
    const int N = n + m;
    Symbolic S;
    S.N = N;
    S.perm.reserve(N);
    std::vector<int> mark(N, -1), level(N, 0), queue;      // mark[v] = id of the region v currently belongs to
    queue.reserve(N);
    int regionId = 0;
    std::vector<std::vector<int>> frontChildren;

    // breadth-first search inside the nodes marked `id`, from `start`; fills `queue` (visit order) and level[]; the visited nodes are re-marked `newId`
    auto bfs = [&](int start, int id, int newId) {
        queue.clear();
        queue.push_back(start); mark[start] = newId; level[start] = 0;
        for (size_t h = 0; h < queue.size(); h++) {
            const int v = queue[h];
            for (int u : adj[v]) if (mark[u] == id) { mark[u] = newId; level[u] = level[v] + 1; queue.push_back(u); }
        }
    };
    auto new_front = [&](const std::vector<int>& pivots, const std::vector<int>& children) {
        const int f = S.nF++;
        S.piv0.push_back((int)S.perm.size()); S.np.push_back((int)pivots.size()); S.parent.push_back(-1);
        for (int v : pivots) S.perm.push_back(v);
        frontChildren.push_back(children);
        for (int c : children) S.parent[c] = f;
        return f;
    };
    // recursion over (node set) -> top fronts; the depth is logarithmic for every separator this routine produces (the balanced level)
    std::vector<int> rootTops;
    struct Rec {
        static void run(Symbolic& S, const std::vector<std::vector<int>>& adj, std::vector<int>& mark, std::vector<int>& level, std::vector<int>& queue,
                        int& regionId, int leaf, std::vector<int>& nodes, std::vector<int>& tops,
                        const std::function<void(int, int, int)>& bfs, const std::function<int(const std::vector<int>&, const std::vector<int>&)>& new_front)
        {
            const int id = regionId++;
            for (int v : nodes) mark[v] = id;
            for (size_t s = 0; s < nodes.size(); s++) {
                if (mark[nodes[s]] != id) continue;                   // already taken by a component found earlier
                const int compId = regionId++;
                bfs(nodes[s], id, compId);
                std::vector<int> comp(queue);
                if ((int)comp.size() <= leaf) { tops.push_back(new_front(comp, {})); continue; }
                // pseudo-peripheral start: the last node of the first search
                const int tmpId = regionId++;
                bfs(comp.back(), compId, tmpId);
                comp = queue;
                const int depth = level[comp.back()];
                if (depth < 2) { tops.push_back(new_front(comp, {})); continue; }      // clique-like: no separator to be had
                // the level that balances the two sides (not the first, not the last)
                std::vector<int> cnt(depth + 1, 0);
                for (int v : comp) cnt[level[v]]++;
                int best = 1; long long bestCost = -1, below = cnt[0];
                for (int l = 1; l < depth; l++) {
                    const long long above = (long long)comp.size() - below - cnt[l];
                    const long long cost = std::max(below, above) + 2LL * cnt[l];     // balance, with a price on the separator's size
                    if (bestCost < 0 || cost < bestCost) { bestCost = cost; best = l; }
                    below += cnt[l];
                }
                std::vector<int> A, B, sep;
                for (int v : comp) (level[v] < best ? A : (level[v] > best ? B : sep)).push_back(v);
                std::vector<int> ch;
                run(S, adj, mark, level, queue, regionId, leaf, A, ch, bfs, new_front);
                run(S, adj, mark, level, queue, regionId, leaf, B, ch, bfs, new_front);
                tops.push_back(new_front(sep, ch));
            }
        }
    };
    {
        std::vector<int> all(N);
        for (int v = 0; v < N; v++) all[v] = v;
        std::function<void(int, int, int)> bfsF = bfs;
        std::function<int(const std::vector<int>&, const std::vector<int>&)> nfF = new_front;
        Rec::run(S, adj, mark, level, queue, regionId, leaf, all, rootTops, bfsF, nfF);
    }
    S.iperm.assign(N, 0);
    for (int p = 0; p < N; p++) S.iperm[S.perm[p]] = p;
    const int nF = S.nF;
    // children lists
    S.childPtr.assign(nF + 1, 0);
    for (int f = 0; f < nF; f++) S.childPtr[f + 1] = S.childPtr[f] + (int)frontChildren[f].size();
    S.child.reserve(S.childPtr[nF]);
    for (int f = 0; f < nF; f++) for (int c : frontChildren[f]) S.child.push_back(c);
    // boundaries, bottom-up (postorder = index order): later neighbours of the pivots and what the children hand up
    std::vector<std::vector<int>> bnd(nF);
    std::vector<int> frontOf(N);
    for (int f = 0; f < nF; f++) for (int j = 0; j < S.np[f]; j++) frontOf[S.piv0[f] + j] = f;
    for (int f = 0; f < nF; f++) {
        const int last = S.piv0[f] + S.np[f];
        std::vector<int>& b = bnd[f];
        for (int j = 0; j < S.np[f]; j++) for (int u : adj[S.perm[S.piv0[f] + j]]) { const int pu = S.iperm[u]; if (pu >= last) b.push_back(pu); }
        for (int c : frontChildren[f]) for (int pu : bnd[c]) if (pu >= last) b.push_back(pu);
        std::sort(b.begin(), b.end()); b.erase(std::unique(b.begin(), b.end()), b.end());
    }
    if (mergeFront > 0) {
        std::vector<char> dead(nF, 0);
        for (int f = 0; f < nF; f++) {
            while (!frontChildren[f].empty()) {
                const int c = frontChildren[f].back();
                if (S.piv0[c] + S.np[c] != S.piv0[f]) break;                                   // (the last child in postorder: always adjacent)
                if (S.np[c] + S.np[f] + (int)bnd[f].size() > mergeFront) break;
                S.piv0[f] = S.piv0[c]; S.np[f] += S.np[c];
                frontChildren[f].pop_back();
                for (int ch : frontChildren[c]) { frontChildren[f].push_back(ch); S.parent[ch] = f; }      // (indices between the other children's and c: still ascending)
                dead[c] = 1;
            }
        }
        std::vector<int> newId(nF, -1);
        int live = 0;
        for (int f = 0; f < nF; f++) if (!dead[f]) newId[f] = live++;
        std::vector<int> piv0(live), npv(live), par(live);
        std::vector<std::vector<int>> fc(live), bd(live);
        for (int f = 0; f < nF; f++) {
            if (dead[f]) continue;
            const int g = newId[f];
            piv0[g] = S.piv0[f]; npv[g] = S.np[f]; par[g] = S.parent[f] >= 0 ? newId[S.parent[f]] : -1;
            for (int ch : frontChildren[f]) fc[g].push_back(newId[ch]);
            bd[g].swap(bnd[f]);
        }
        S.piv0.swap(piv0); S.np.swap(npv); S.parent.swap(par); frontChildren.swap(fc); bnd.swap(bd);
        S.nF = live;
        for (int f = 0; f < live; f++) for (int j = 0; j < S.np[f]; j++) frontOf[S.piv0[f] + j] = f;
        S.childPtr.assign(live + 1, 0); S.child.clear();
        for (int f = 0; f < live; f++) { S.childPtr[f + 1] = S.childPtr[f] + (int)frontChildren[f].size(); for (int ch : frontChildren[f]) S.child.push_back(ch); }
    }
    S.nb.resize(S.nF); S.rowPtr.assign(S.nF + 1, 0);
    for (int f = 0; f < S.nF; f++) { S.nb[f] = (int)bnd[f].size(); S.rowPtr[f + 1] = S.rowPtr[f] + S.nb[f]; }
    S.rows.reserve(S.rowPtr[S.nF]);
    for (int f = 0; f < S.nF; f++) for (int p : bnd[f]) S.rows.push_back(p);
    // local index of a position in front f
    auto local = [&](int f, int pos) {
        if (pos >= S.piv0[f] && pos < S.piv0[f] + S.np[f]) return pos - S.piv0[f];
        const int* b0 = S.rows.data() + S.rowPtr[f]; const int* b1 = S.rows.data() + S.rowPtr[f + 1];
        const int* it = std::lower_bound(b0, b1, pos);
        return (it != b1 && *it == pos) ? S.np[f] + (int)(it - b0) : -1;
    };
    S.rel.assign(S.rows.size(), -1);
    for (int f = 0; f < S.nF; f++) {
        const int p = S.parent[f];
        if (p < 0) continue;      // (a root has no boundary: nothing comes after the last region)
        for (int a = 0; a < S.nb[f]; a++) S.rel[S.rowPtr[f] + a] = local(p, S.rows[S.rowPtr[f] + a]);
    }
    // assembly lists
    const int nnzQ = Qp[n];
    std::vector<std::vector<int>> aS(S.nF), aG(S.nF), aP(S.nF);
    auto put = [&](int pa, int pb, int src, int gate) {
        const int lo = std::min(pa, pb), hi = std::max(pa, pb), f = frontOf[lo], ff = S.np[f] + S.nb[f];
        const int j = lo - S.piv0[f], i = local(f, hi);
        aS[f].push_back(src); aG[f].push_back(gate); aP[f].push_back(i + ff * j);      // (i >= 0: hi is a later neighbour of a pivot of f)
    };
    for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) { const int pi = S.iperm[i], pj = S.iperm[Qi[k]]; if (pj <= pi) put(pi, pj, k, -1); }     // one of each symmetric pair
    for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) put(S.iperm[n + r], S.iperm[Ei[k]], nnzQ + k, r);
    S.asmPtr.assign(S.nF + 1, 0);
    for (int f = 0; f < S.nF; f++) S.asmPtr[f + 1] = S.asmPtr[f] + (int)aS[f].size();
    for (int f = 0; f < S.nF; f++) { S.asmSrc.insert(S.asmSrc.end(), aS[f].begin(), aS[f].end()); S.asmGate.insert(S.asmGate.end(), aG[f].begin(), aG[f].end()); S.asmPos.insert(S.asmPos.end(), aP[f].begin(), aP[f].end()); }
    // storage: panels one after the other; the stack of update blocks (a front's block takes the place its children's blocks had)
    S.Loff.resize(S.nF); S.CBoff.resize(S.nF);
    long long sp = 0;
    for (int f = 0; f < S.nF; f++) {
        const long long ff = S.np[f] + S.nb[f];
        S.Loff[f] = S.Lsize; S.Lsize += ff * S.np[f];
        S.maxFront = std::max(S.maxFront, (int)ff);
        if (!frontChildren[f].empty()) sp = S.CBoff[frontChildren[f][0]];
        S.CBoff[f] = sp;
        sp += (long long)S.nb[f] * S.nb[f];
        S.stackSize = std::max(S.stackSize, sp);
        for (int j = 0; j < S.np[f]; j++) { const long long r = ff - j - 1; S.flops += r * (r + 3); S.nnzL += r; }
    }
    return S;
}

}  // namespace lcqp_general
