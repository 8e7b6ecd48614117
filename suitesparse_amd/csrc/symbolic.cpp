// symbolic.cpp -- host symbolic analysis (include/mi355_symbolic.h).  Pure C++.
// Integer restatement of the reference's analysis chain for a GIVEN permutation; each
// step cites the reference code it follows.  Verified bit-exact against the reference
// (cholmod_l_analyze with CHOLMOD_GIVEN) in tests/test_symbolic.py.
#include "../../include/mi355_symbolic.h"
#include "../../include/mi355_cholmod.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

typedef int64_t Int;
static const Int EMPTY = -1;

struct mi355_symbolic_s {
    Int n = 0, nsuper = 0, ssize = 0, xsize = 0, maxcsize = 1, maxesize = 1, snnz = 0;
    double fl = 0, lnz = 0;
    std::vector<Int> Perm, ColCount, Parent, Super, Lpi, Lpx, Ls;
};

namespace {

struct Csc { std::vector<Int> p, i; std::vector<double> x; };

// B = A' (n x n); row indices of every column of B come out sorted
void transpose(Int n, const Csc &A, Csc &B, bool values)
{
    B.p.assign(n + 1, 0);
    const Int nz = A.p[n];
    B.i.resize(nz);
    if (values) B.x.resize(nz);
    for (Int k = 0; k < nz; k++) B.p[A.i[k] + 1]++;
    for (Int j = 0; j < n; j++) B.p[j + 1] += B.p[j];
    std::vector<Int> w(B.p.begin(), B.p.end() - 1);
    for (Int j = 0; j < n; j++)
        for (Int k = A.p[j]; k < A.p[j + 1]; k++) {
            const Int q = w[A.i[k]]++;
            B.i[q] = j;
            if (values) B.x[q] = A.x[k];
        }
}

// S = tril(P A P') with sorted columns, from the lower triangle (Ap, Ai, Ax) of A
// (the two-transpose sequence of cholmod_factorize.c:230-236 produces the same arrays)
void permute_lower(Int n, const Int *Ap, const Int *Ai, const double *Ax, const Int *pinv, Csc &S)
{
    Csc U;      // triu(P A P'), unsorted
    U.p.assign(n + 1, 0);
    const Int nz = Ap[n];
    U.i.resize(nz);
    if (Ax) U.x.resize(nz);
    for (Int j = 0; j < n; j++)
        for (Int k = Ap[j]; k < Ap[j + 1]; k++) {
            const Int a = pinv ? pinv[Ai[k]] : Ai[k], b = pinv ? pinv[j] : j;
            U.p[std::max(a, b) + 1]++;
        }
    for (Int j = 0; j < n; j++) U.p[j + 1] += U.p[j];
    std::vector<Int> w(U.p.begin(), U.p.end() - 1);
    for (Int j = 0; j < n; j++)
        for (Int k = Ap[j]; k < Ap[j + 1]; k++) {
            const Int a = pinv ? pinv[Ai[k]] : Ai[k], b = pinv ? pinv[j] : j;
            const Int q = w[std::max(a, b)]++;
            U.i[q] = std::min(a, b);
            if (Ax) U.x[q] = Ax[k];
        }
    transpose(n, U, S, Ax != nullptr);
}

// elimination tree of the matrix whose upper triangle is U (cholmod_etree.c, symmetric case)
void etree(Int n, const Csc &U, std::vector<Int> &parent)
{
    parent.assign(n, EMPTY);
    std::vector<Int> ancestor(n, EMPTY);
    for (Int k = 0; k < n; k++)
        for (Int p = U.p[k]; p < U.p[k + 1]; p++) {
            Int i = U.i[p];
            while (i != EMPTY && i < k) {
                const Int inext = ancestor[i];
                ancestor[i] = k;
                if (inext == EMPTY) parent[i] = k;
                i = inext;
            }
        }
}

// cholmod_postorder.c:131-300 (non-recursive dfs :60-95), optional weights
Int postorder(Int n, const std::vector<Int> &Parent, const Int *Weight, std::vector<Int> &Post)
{
    std::vector<Int> Head(n + 1, EMPTY), Next(n, EMPTY), Pstack(n + 1, EMPTY);
    Post.assign(n, 0);
    if (!Weight) {
        for (Int j = n - 1; j >= 0; j--) {
            const Int p = Parent[j];
            if (p >= 0 && p < n) { Next[j] = Head[p]; Head[p] = j; }
        }
    } else {
        std::vector<Int> &Whead = Pstack;
        for (Int w = 0; w < n; w++) Whead[w] = EMPTY;
        for (Int j = 0; j < n; j++) {
            const Int p = Parent[j];
            if (p >= 0 && p < n) {
                Int w = Weight[j];
                w = std::max<Int>(0, w);
                w = std::min<Int>(w, n - 1);
                Next[j] = Whead[w];
                Whead[w] = j;
            }
        }
        for (Int w = n - 1; w >= 0; w--)
            for (Int j = Whead[w], nextj; j != EMPTY; j = nextj) {
                nextj = Next[j];
                const Int p = Parent[j];
                Next[j] = Head[p];
                Head[p] = j;
            }
    }
    Int k = 0;
    for (Int j = 0; j < n; j++) {
        if (Parent[j] != EMPTY) continue;
        Int phead = 0;
        Pstack[0] = j;
        while (phead >= 0) {
            const Int p = Pstack[phead];
            const Int c = Head[p];
            if (c == EMPTY) { phead--; Post[k++] = p; }
            else { Head[p] = Next[c]; Pstack[++phead] = c; }
        }
    }
    return k;
}

// column counts of L (Gilbert-Ng-Peyton skeleton algorithm; the values are unique, so they
// equal cholmod_rowcolcounts' ColCount).  S = lower pattern by columns.
void colcounts(Int n, const Csc &S, const std::vector<Int> &parent, const std::vector<Int> &post,
               std::vector<Int> &cc)
{
    cc.assign(n, 0);
    std::vector<Int> first(n, EMPTY), maxfirst(n, EMPTY), prevleaf(n, EMPTY), ancestor(n);
    for (Int k = 0; k < n; k++) {
        Int j = post[k];
        cc[j] = (first[j] == EMPTY) ? 1 : 0;
        for (; j != EMPTY && first[j] == EMPTY; j = parent[j]) first[j] = k;
    }
    for (Int i = 0; i < n; i++) ancestor[i] = i;
    for (Int k = 0; k < n; k++) {
        const Int j = post[k];
        if (parent[j] != EMPTY) cc[parent[j]]--;
        for (Int p = S.p[j]; p < S.p[j + 1]; p++) {
            const Int i = S.i[p];
            if (i <= j || first[j] <= maxfirst[i]) continue;
            maxfirst[i] = first[j];
            const Int jprev = prevleaf[i];
            prevleaf[i] = j;
            if (jprev == EMPTY) { cc[j]++; continue; }          // first leaf of row subtree i
            Int q = jprev;
            while (q != ancestor[q]) q = ancestor[q];
            for (Int s = jprev, sp; s != q; s = sp) { sp = ancestor[s]; ancestor[s] = q; }
            cc[j]++;
            cc[q]--;
        }
        if (parent[j] != EMPTY) ancestor[j] = parent[j];
    }
    for (Int k = 0; k < n; k++) {
        const Int j = post[k];      // children before parents
        if (parent[j] != EMPTY) cc[parent[j]] += cc[j];
    }
}

// cholmod_super_symbolic.c:76-127
inline void subtree(Int k, const Csc &U, const std::vector<Int> &SuperMap, const std::vector<Int> &Sparent,
                    Int mark, Int k1, std::vector<Int> &Flag, std::vector<Int> &Ls, std::vector<Int> &Lpi2)
{
    for (Int p = U.p[k]; p < U.p[k + 1]; p++) {
        const Int i = U.i[p];
        if (i < k1) {
            for (Int si = SuperMap[i]; Flag[si] < mark; si = Sparent[si]) {
                Ls[Lpi2[si]++] = k;
                Flag[si] = mark;
            }
        } else {
            break;      // columns of U are sorted
        }
    }
}

}  // namespace

extern "C" {

int mi355_nd_grid_order(int64_t nx, int64_t ny, int64_t nz, int64_t dof, int64_t *perm)
{
    if (nx <= 0 || ny <= 0 || nz <= 0 || dof <= 0 || !perm) return MI355_INVALID;
    struct Box { Int lo[3], hi[3]; int stage; Int mid; int ax; };
    std::vector<Box> stack;
    Int k = 0;
    auto emit_box = [&](const Int lo[3], const Int hi[3]) {
        for (Int z = lo[2]; z < hi[2]; z++)
            for (Int y = lo[1]; y < hi[1]; y++)
                for (Int x = lo[0]; x < hi[0]; x++) {
                    const Int node = (z * ny + y) * nx + x;
                    for (Int d = 0; d < dof; d++) perm[k++] = node * dof + d;
                }
    };
    Box root;
    root.lo[0] = root.lo[1] = root.lo[2] = 0;
    root.hi[0] = nx; root.hi[1] = ny; root.hi[2] = nz;
    root.stage = 0; root.mid = 0; root.ax = 0;
    stack.push_back(root);
    while (!stack.empty()) {
        Box b = stack.back();
        stack.pop_back();
        const Int ext[3] = {b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]};
        if (ext[0] <= 0 || ext[1] <= 0 || ext[2] <= 0) continue;
        if (b.stage == 1) {         // emit the separator plane after both halves
            Int lo[3] = {b.lo[0], b.lo[1], b.lo[2]}, hi[3] = {b.hi[0], b.hi[1], b.hi[2]};
            lo[b.ax] = b.mid; hi[b.ax] = b.mid + 1;
            emit_box(lo, hi);
            continue;
        }
        int ax = 0;
        if (ext[1] > ext[ax]) ax = 1;
        if (ext[2] > ext[ax]) ax = 2;
        if (ext[0] * ext[1] * ext[2] <= 8 || ext[ax] <= 2) { emit_box(b.lo, b.hi); continue; }
        const Int mid = b.lo[ax] + ext[ax] / 2;
        Box sep = b; sep.stage = 1; sep.mid = mid; sep.ax = ax;
        Box left = b, right = b;
        left.hi[ax] = mid; left.stage = 0;
        right.lo[ax] = mid + 1; right.stage = 0;
        stack.push_back(sep);       // popped last
        stack.push_back(right);
        stack.push_back(left);      // popped first
    }
    return (k == nx * ny * nz * dof) ? MI355_OK : MI355_INVALID;
}

int mi355_symbolic_analyze(int64_t n, const int64_t *Ap, const int64_t *Ai, const int64_t *perm,
                           const int64_t *nrelax_in, const double *zrelax_in, mi355_symbolic **out)
{
    if (!out) return MI355_INVALID;
    *out = nullptr;
    if (n < 0 || (n > 0 && (!Ap || !Ai))) return MI355_INVALID;
    mi355_symbolic_s *R = new (std::nothrow) mi355_symbolic_s;
    if (!R) return MI355_OUT_OF_MEMORY;
    R->n = n;
    if (n == 0) { R->Super.assign(1, 0); R->Lpi.assign(1, 0); R->Lpx.assign(1, 0); *out = R; return MI355_OK; }
    for (Int j = 0; j < n; j++) {
        if (Ap[j + 1] < Ap[j]) { delete R; return MI355_INVALID; }
        for (Int k = Ap[j]; k < Ap[j + 1]; k++)
            if (Ai[k] < j || Ai[k] >= n) { delete R; return MI355_INVALID; }   // lower triangle only
    }
    const Int nrelax[3] = {nrelax_in ? nrelax_in[0] : 4, nrelax_in ? nrelax_in[1] : 16, nrelax_in ? nrelax_in[2] : 48};
    double zrelax[3] = {zrelax_in ? zrelax_in[0] : 0.8, zrelax_in ? zrelax_in[1] : 0.1, zrelax_in ? zrelax_in[2] : 0.05};
    for (double &z : zrelax) if (std::isnan(z)) z = 0;

    std::vector<Int> Perm(n), pinv(n, EMPTY);
    for (Int k = 0; k < n; k++) {
        const Int v = perm ? perm[k] : k;
        if (v < 0 || v >= n || pinv[v] != EMPTY) { delete R; return MI355_INVALID; }
        Perm[k] = v; pinv[v] = k;
    }

    // ---- analyze_ordering (cholmod_analyze.c:297-371): permute, etree, postorder, colcounts
    Csc S, U;
    permute_lower(n, Ap, Ai, nullptr, pinv.data(), S);
    transpose(n, S, U, false);
    std::vector<Int> Parent, Post, ColCount;
    etree(n, U, Parent);
    if (postorder(n, Parent, nullptr, Post) != n) { delete R; return MI355_INVALID; }
    colcounts(n, S, Parent, Post, ColCount);
    double fl = 0, lnz = 0;
    for (Int j = 0; j < n; j++) { const double c = (double)ColCount[j]; fl += c * c; lnz += c; }
    R->fl = fl; R->lnz = lnz;

    // ---- weighted postorder folded into the permutation (cholmod_analyze.c:828-880)
    if (postorder(n, Parent, ColCount.data(), Post) != n) { delete R; return MI355_INVALID; }
    {
        std::vector<Int> W(n), InvPost(n);
        for (Int k = 0; k < n; k++) W[k] = Perm[Post[k]];
        Perm = W;
        for (Int k = 0; k < n; k++) W[k] = ColCount[Post[k]];
        ColCount = W;
        for (Int k = 0; k < n; k++) InvPost[Post[k]] = k;
        for (Int c = 0; c < n; c++) {
            const Int op = Parent[Post[c]];
            W[c] = (op == EMPTY) ? EMPTY : InvPost[op];
        }
        Parent = W;
    }
    for (Int k = 0; k < n; k++) pinv[Perm[k]] = k;
    permute_lower(n, Ap, Ai, nullptr, pinv.data(), S);
    transpose(n, S, U, false);
    R->snnz = S.p[n];

    // ---- cholmod_super_symbolic2 (cholmod_super_symbolic.c:391-928), for_whom = Cholesky, no GPU caps
    std::vector<Int> Wi(n, 0), SuperMap(n), Sparent(n), Snz(n), Merged(n), Zeros(n), Super(n + 1);
    for (Int j = 0; j < n; j++) if (Parent[j] != EMPTY) Wi[Parent[j]]++;
    Int nfsuper = 1;
    Super[0] = 0;
    for (Int j = 1; j < n; j++)
        if (Parent[j - 1] != j || ColCount[j - 1] != ColCount[j] + 1 || Wi[j] > 1) Super[nfsuper++] = j;
    Super[nfsuper] = n;
    std::vector<Int> &Nscol = Wi;
    for (Int s = 0; s < nfsuper; s++)
        for (Int k = Super[s]; k < Super[s + 1]; k++) SuperMap[k] = s;
    for (Int s = 0; s < nfsuper; s++) {
        const Int parent = Parent[Super[s + 1] - 1];
        Sparent[s] = (parent == EMPTY) ? EMPTY : SuperMap[parent];
    }
    for (Int s = 0; s < nfsuper; s++) {
        Merged[s] = EMPTY;
        Nscol[s] = Super[s + 1] - Super[s];
        Zeros[s] = 0;
        Snz[s] = ColCount[Super[s]];
    }
    const double Int_max = (double)std::numeric_limits<Int>::max();
    for (Int s = nfsuper - 2; s >= 0; s--) {
        Int ss = Sparent[s];
        if (ss == EMPTY) continue;
        for (ss = Sparent[s]; Merged[ss] != EMPTY; ss = Merged[ss]) {}
        const Int sparent = ss;
        for (Int t = Sparent[s], snext; Merged[t] != EMPTY; t = snext) { snext = Merged[t]; Merged[t] = sparent; }
        if (sparent != s + 1) continue;
        const Int nscol0 = Nscol[s], nscol1 = Nscol[s + 1], ns = nscol0 + nscol1;
        Int totzeros = Zeros[s + 1];
        const double lnz1 = (double)Snz[s + 1];
        bool merge;
        if (ns <= nrelax[0]) {
            merge = true;
        } else {
            const double lnz0 = (double)Snz[s];
            const double xnewzeros = nscol0 * (lnz1 + nscol0 - lnz0);
            const Int newzeros = nscol0 * (Snz[s + 1] + nscol0 - Snz[s]);
            if (xnewzeros == 0) {
                merge = true;
            } else {
                const double xtotzeros = ((double)totzeros) + xnewzeros;
                const double xns = (double)ns;
                const double xtotsize = (xns * (xns + 1) / 2) + xns * (lnz1 - nscol1);
                const double z = xtotzeros / xtotsize;
                totzeros += newzeros;
                merge = ((ns <= nrelax[1] && z < zrelax[0]) || (ns <= nrelax[2] && z < zrelax[1]) || (z < zrelax[2]))
                        && (xtotsize < Int_max / sizeof(double));
            }
        }
        if (merge) {
            Zeros[s] = totzeros;
            Merged[s + 1] = s;
            Snz[s] = nscol0 + Snz[s + 1];
            Nscol[s] += Nscol[s + 1];
        }
    }
    Int nsuper = 0;
    for (Int s = 0; s < nfsuper; s++)
        if (Merged[s] == EMPTY) { Super[nsuper] = Super[s]; Snz[nsuper] = Snz[s]; nsuper++; }
    Super[nsuper] = n;
    for (Int s = 0; s < nsuper; s++)
        for (Int k = Super[s]; k < Super[s + 1]; k++) SuperMap[k] = s;
    for (Int s = 0; s < nsuper; s++) {
        const Int parent = Parent[Super[s + 1] - 1];
        Sparent[s] = (parent == EMPTY) ? EMPTY : SuperMap[parent];
    }
    R->nsuper = nsuper;
    R->Super.assign(Super.begin(), Super.begin() + nsuper + 1);
    R->Lpi.resize(nsuper + 1);
    R->Lpx.resize(nsuper + 1);
    Int ps = 0, px = 0;
    for (Int s = 0; s < nsuper; s++) {
        R->Lpi[s] = ps; R->Lpx[s] = px;
        ps += Snz[s];
        px += (Super[s + 1] - Super[s]) * Snz[s];
    }
    R->Lpi[nsuper] = ps; R->Lpx[nsuper] = px;
    R->ssize = std::max<Int>(1, ps);
    R->xsize = std::max<Int>(1, px);
    R->Ls.assign(R->ssize, 0);
    {
        std::vector<Int> Lpi2(R->Lpi.begin(), R->Lpi.end() - 1), Flag(nsuper, EMPTY);
        Int mark = 0;
        for (Int s = 0; s < nsuper; s++) {
            const Int k1 = Super[s], k2 = Super[s + 1];
            for (Int k = k1; k < k2; k++) R->Ls[Lpi2[s]++] = k;
            for (Int k = k1; k < k2; k++) {
                mark++;
                Flag[s] = mark;
                subtree(k, U, SuperMap, Sparent, mark, k1, Flag, R->Ls, Lpi2);
            }
        }
        for (Int s = 0; s < nsuper; s++)
            if (Lpi2[s] != R->Lpi[s + 1]) { delete R; return MI355_INVALID; }   // ColCount inconsistent
    }
    // maxcsize / maxesize (:889-923)
    Int maxcsize = 1, maxesize = 1;
    for (Int d = 0; d < nsuper; d++) {
        const Int nscol = Super[d + 1] - Super[d];
        Int p = R->Lpi[d] + nscol, plast = p;
        const Int pend = R->Lpi[d + 1];
        maxesize = std::max(maxesize, pend - p);
        Int slast = (p == pend) ? EMPTY : SuperMap[R->Ls[p]];
        for (; p <= pend; p++) {
            const Int s = (p == pend) ? EMPTY : SuperMap[R->Ls[p]];
            if (s != slast) {
                const Int ndrow1 = p - plast, ndrow2 = pend - plast;
                maxcsize = std::max(maxcsize, ndrow2 * ndrow1);
                plast = p;
                slast = s;
            }
        }
    }
    R->maxcsize = maxcsize;
    R->maxesize = maxesize;
    R->Perm = Perm;
    R->ColCount = ColCount;
    R->Parent = Parent;
    *out = R;
    return MI355_OK;
}

int mi355_symbolic_free(mi355_symbolic *S) { delete S; return MI355_OK; }

int mi355_symbolic_info(const mi355_symbolic *S, int64_t *iout, double *dout)
{
    if (!S || !iout || !dout) return MI355_INVALID;
    iout[0] = S->n; iout[1] = S->nsuper; iout[2] = S->ssize; iout[3] = S->xsize;
    iout[4] = S->maxcsize; iout[5] = S->maxesize; iout[6] = S->snnz; iout[7] = 0;
    dout[0] = S->fl; dout[1] = S->lnz;
    return MI355_OK;
}

int mi355_symbolic_get(const mi355_symbolic *S, int64_t *Perm, int64_t *ColCount, int64_t *Parent,
                       int64_t *Super, int64_t *Lpi, int64_t *Lpx, int64_t *Ls)
{
    if (!S) return MI355_INVALID;
    auto cp = [](int64_t *dst, const std::vector<Int> &v) { if (dst && !v.empty()) memcpy(dst, v.data(), v.size() * sizeof(Int)); };
    cp(Perm, S->Perm); cp(ColCount, S->ColCount); cp(Parent, S->Parent);
    cp(Super, S->Super); cp(Lpi, S->Lpi); cp(Lpx, S->Lpx); cp(Ls, S->Ls);
    return MI355_OK;
}

int mi355_symbolic_permute(const mi355_symbolic *S, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                           int64_t *Sp, int64_t *Si, double *Sx)
{
    if (!S || !Ap || !Ai || !Ax || !Sp || !Si || !Sx) return MI355_INVALID;
    const Int n = S->n;
    std::vector<Int> pinv(n);
    for (Int k = 0; k < n; k++) pinv[S->Perm[k]] = k;
    Csc out;
    permute_lower(n, Ap, Ai, Ax, pinv.data(), out);
    memcpy(Sp, out.p.data(), (n + 1) * sizeof(Int));
    memcpy(Si, out.i.data(), out.i.size() * sizeof(Int));
    memcpy(Sx, out.x.data(), out.x.size() * sizeof(double));
    return MI355_OK;
}

}  // extern "C"
