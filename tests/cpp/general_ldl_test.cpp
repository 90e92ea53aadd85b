// CPU check of the general sparse LDL' of the sparse arm (lcqpow_amd/csrc/lcqp_sparse_general.hpp): the symbolic analysis, and a scalar restatement
// of the numeric loops the device runs (sp_general_factor / sp_general_solve in lcqp_sparse.hip: same fronts, same order of operations), against a
// dense LDL' solve of the same KKT matrix.  usage: general_ldl_test [grid size] [leaf]      (no GPU; run by tests/test_general_ldl.py)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../lcqpow_amd/csrc/lcqp_sparse_general.hpp"

using lcqp_general::Symbolic;

struct Factor { std::vector<double> L, Dinv, stack, F; };

// numeric factorisation: K = [Q + dprim I, Ea'; Ea, -ddual I] in the ordering S.perm; rows with use[r] == 0 are decoupled (diagonal -1)
static void factor(const Symbolic& S, int n, int m, const double* Qx, const double* Ex, int nnzQ, double dprim, double ddual, const std::vector<int>& use, Factor& W)
{
    W.L.assign(S.Lsize, 0.0); W.Dinv.assign(S.N, 0.0); W.stack.assign(S.stackSize ? S.stackSize : 1, 0.0); W.F.assign((size_t)S.maxFront * S.maxFront, 0.0);
    for (int f = 0; f < S.nF; f++) {
        const int np = S.np[f], nb = S.nb[f], ff = np + nb;
        double* F = W.F.data();
        for (int e = 0; e < ff * ff; e++) F[e] = 0.0;
        for (int e = S.asmPtr[f]; e < S.asmPtr[f + 1]; e++) {
            const int src = S.asmSrc[e], gate = S.asmGate[e];
            const double v = (gate >= 0 && !use[gate]) ? 0.0 : (src >= nnzQ ? Ex[src - nnzQ] : Qx[src]);
            F[S.asmPos[e]] += v;
        }
        for (int j = 0; j < np; j++) {
            const int node = S.perm[S.piv0[f] + j];
            if (node < n) F[j + ff * j] += dprim;
            else F[j + ff * j] = use[node - n] ? -ddual : -1.0;
        }
        for (int ci = S.childPtr[f]; ci < S.childPtr[f + 1]; ci++) {
            const int c = S.child[ci], nbc = S.nb[c];
            const double* CB = W.stack.data() + S.CBoff[c];
            const int* rel = S.rel.data() + S.rowPtr[c];
            for (int b = 0; b < nbc; b++) for (int a = b; a < nbc; a++) F[rel[a] + ff * rel[b]] += CB[a + nbc * b];
        }
        for (int j = 0; j < np; j++) {
            const double dinv = 1.0 / F[j + ff * j];
            W.Dinv[S.piv0[f] + j] = dinv;
            for (int k = j + 1; k < ff; k++) {
                const double fkj = F[k + ff * j];
                if (fkj == 0.0) continue;
                for (int i = k; i < ff; i++) F[i + ff * k] -= (F[i + ff * j] * dinv) * fkj;
            }
            for (int i = j + 1; i < ff; i++) F[i + ff * j] *= dinv;
        }
        double* Lp = W.L.data() + S.Loff[f];
        for (int j = 0; j < np; j++) for (int i = j + 1; i < ff; i++) Lp[i + ff * j] = F[i + ff * j];
        double* CB = W.stack.data() + S.CBoff[f];
        for (int b = 0; b < nb; b++) for (int a = b; a < nb; a++) CB[a + nb * b] = F[(np + a) + ff * (np + b)];
    }
}

static void solve(const Symbolic& S, const Factor& W, std::vector<double>& b)
{
    std::vector<double> bl(S.maxFront);
    auto pos = [&](int f, int i) { return i < S.np[f] ? S.piv0[f] + i : S.rows[S.rowPtr[f] + i - S.np[f]]; };
    for (int f = 0; f < S.nF; f++) {
        const int np = S.np[f], ff = np + S.nb[f];
        const double* Lp = W.L.data() + S.Loff[f];
        for (int i = 0; i < ff; i++) bl[i] = b[pos(f, i)];
        for (int j = 0; j < np; j++) { const double yj = bl[j]; for (int i = j + 1; i < ff; i++) bl[i] -= Lp[i + ff * j] * yj; }
        for (int i = 0; i < ff; i++) b[pos(f, i)] = bl[i];
    }
    for (int p = 0; p < S.N; p++) b[p] *= W.Dinv[p];
    for (int f = S.nF - 1; f >= 0; f--) {
        const int np = S.np[f], ff = np + S.nb[f];
        const double* Lp = W.L.data() + S.Loff[f];
        for (int i = 0; i < ff; i++) bl[i] = b[pos(f, i)];
        for (int j = np - 1; j >= 0; j--) { double s = bl[j]; for (int i = j + 1; i < ff; i++) s -= Lp[i + ff * j] * bl[i]; bl[j] = s; }
        for (int j = 0; j < np; j++) b[S.piv0[f] + j] = bl[j];
    }
}

int main(int argc, char** argv)
{
    const int g = argc > 1 ? std::atoi(argv[1]) : 12, leaf = argc > 2 ? std::atoi(argv[2]) : 32, mergeFront = argc > 3 ? std::atoi(argv[3]) : 64;
    // a 2-D grid Hessian (5-point stencil), rows of E that couple vertical and horizontal neighbours, a few denser rows
    const int n = g * g;
    std::vector<std::vector<std::pair<int, double>>> Qrows(n), Erows;
    unsigned long long st = 0x9E3779B97F4A7C15ULL;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((st >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53); };
    for (int r = 0; r < g; r++) for (int c = 0; c < g; c++) {
        const int i = r * g + c;
        Qrows[i].push_back({i, 4.5 + rnd()});
        if (c + 1 < g) { Qrows[i].push_back({i + 1, -1.0}); Qrows[i + 1].push_back({i, -1.0}); }
        if (r + 1 < g) { Qrows[i].push_back({i + g, -1.0}); Qrows[i + g].push_back({i, -1.0}); }
    }
    for (int k = 0; k < n / 3; k++) {
        const int r = (int)(rnd() * (g - 1)), c = (int)(rnd() * (g - 1));
        std::vector<std::pair<int, double>> row;
        row.push_back({r * g + c, 0.5 + rnd()});
        row.push_back({(r + 1) * g + c, -0.5 - rnd()});
        if (k % 7 == 0) row.push_back({r * g + c + 1, rnd()});
        std::sort(row.begin(), row.end());
        Erows.push_back(row);
    }
    const int m = (int)Erows.size(), N = n + m;
    std::vector<int> Qp(n + 1, 0), Qi, Ep(m + 1, 0), Ei;
    std::vector<double> Qx, Ex;
    for (int i = 0; i < n; i++) { std::sort(Qrows[i].begin(), Qrows[i].end()); for (auto& e : Qrows[i]) { Qi.push_back(e.first); Qx.push_back(e.second); } Qp[i + 1] = (int)Qi.size(); }
    for (int r = 0; r < m; r++) { for (auto& e : Erows[r]) { Ei.push_back(e.first); Ex.push_back(e.second); } Ep[r + 1] = (int)Ei.size(); }
    std::vector<std::vector<int>> adj(N);
    for (int i = 0; i < n; i++) for (int k = Qp[i]; k < Qp[i + 1]; k++) if (Qi[k] != i) adj[i].push_back(Qi[k]);
    for (int r = 0; r < m; r++) for (int k = Ep[r]; k < Ep[r + 1]; k++) { adj[n + r].push_back(Ei[k]); adj[Ei[k]].push_back(n + r); }
    for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    Symbolic S = lcqp_general::analyze(n, m, adj, Qp.data(), Qi.data(), Ep.data(), Ei.data(), leaf, mergeFront);
    int fails = 0;
    // structure checks
    { std::vector<int> seen(N, 0); for (int p = 0; p < N; p++) seen[S.perm[p]]++; for (int v = 0; v < N; v++) if (seen[v] != 1) fails++; }
    for (int f = 0; f < S.nF; f++) {
        if (S.parent[f] >= 0 && S.parent[f] <= f) fails++;                                                 // postorder
        for (int a = 0; a < S.nb[f]; a++) {
            if (S.rows[S.rowPtr[f] + a] < S.piv0[f] + S.np[f]) fails++;                                    // boundary behind the pivots
            if (a && S.rows[S.rowPtr[f] + a] <= S.rows[S.rowPtr[f] + a - 1]) fails++;
            if (S.parent[f] >= 0 && S.rel[S.rowPtr[f] + a] < 0) fails++;                                   // every boundary row is a row of the parent's front
        }
        if (S.parent[f] < 0 && S.nb[f] != 0) fails++;
    }
    for (int e = 0; e < (int)S.asmPos.size(); e++) if (S.asmPos[e] < 0) fails++;
    // numeric check against a dense LDL' (two working sets: every row in, every third row out)
    double worst = 0.0;
    for (int pass = 0; pass < 2 && N <= 4000; pass++) {
        std::vector<int> use(m, 1);
        if (pass) for (int r = 0; r < m; r += 3) use[r] = 0;
        const double dprim = 1e-8, ddual = 1e-9;
        Factor W;
        factor(S, n, m, Qx.data(), Ex.data(), (int)Qx.size(), dprim, ddual, use, W);
        std::vector<double> K((size_t)N * N, 0.0);
        for (int i = 0; i < n; i++) { for (int k = Qp[i]; k < Qp[i + 1]; k++) K[(size_t)i * N + Qi[k]] = Qx[k]; K[(size_t)i * N + i] += dprim; }
        for (int r = 0; r < m; r++) {
            if (!use[r]) { K[(size_t)(n + r) * N + n + r] = -1.0; continue; }
            K[(size_t)(n + r) * N + n + r] = -ddual;
            for (int k = Ep[r]; k < Ep[r + 1]; k++) { K[(size_t)(n + r) * N + Ei[k]] = Ex[k]; K[(size_t)Ei[k] * N + n + r] = Ex[k]; }
        }
        std::vector<double> xs(N), rhs(N, 0.0), b(N);
        for (int v = 0; v < N; v++) xs[v] = 2.0 * rnd() - 1.0;
        for (int v = 0; v < N; v++) { double s = 0; for (int u = 0; u < N; u++) s += K[(size_t)v * N + u] * xs[u]; rhs[v] = s; }
        for (int p = 0; p < N; p++) b[p] = rhs[S.perm[p]];
        solve(S, W, b);
        // one step of iterative refinement, as the corrections of the polish are (the regularised KKT matrix has condition 1e9: a raw solve is good to 1e-7)
        std::vector<double> x(N), r(N);
        for (int p = 0; p < N; p++) x[S.perm[p]] = b[p];
        for (int v = 0; v < N; v++) { double s2 = rhs[v]; for (int u = 0; u < N; u++) s2 -= K[(size_t)v * N + u] * x[u]; r[v] = s2; }
        for (int p = 0; p < N; p++) b[p] = r[S.perm[p]];
        solve(S, W, b);
        for (int p = 0; p < N; p++) x[S.perm[p]] += b[p];
        // (duplicated random rows of E make K singular up to ddual: the solution is then only determined through its residual)
        double err = 0.0;
        for (int v = 0; v < N; v++) { double s2 = rhs[v]; for (int u = 0; u < N; u++) s2 -= K[(size_t)v * N + u] * x[u]; err = std::fmax(err, std::fabs(s2)); }
        worst = std::fmax(worst, err);
    }
    std::printf("general LDL': grid %d x %d, n %d, m %d, N %d, leaf %d: fronts %d, largest front %d, nnz(L) %lld (%.1f per row), panel storage %lld doubles, stack %lld doubles, flops %.3g, max |K x - b| after one refinement %.2e, structure failures %d\n",
                g, g, n, m, N, leaf, S.nF, S.maxFront, S.nnzL, (double)S.nnzL / N, S.Lsize, S.stackSize, (double)S.flops, worst, fails);
    return (fails == 0 && worst < 1e-11) ? 0 : 1;
}
