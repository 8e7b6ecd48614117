// symbolic.cpp -- host symbolic analysis (include/mi355_symbolic.h).  Pure C++.
// For a GIVEN permutation it produces the structure the reference's analysis chain produces (Perm after the
// weighted postorder, ColCount, Super, Lpi, Lpx, Ls, maxcsize, maxesize, fl, lnz), verified bit-exact against
// cholmod_l_analyze with CHOLMOD_GIVEN in tests/test_symbolic.py.  The OUTPUT is pinned by the reference; the
// algorithms are formulated here: compressed child tables with cursors for the postorders, supernode chains and a
// contiguous-range amalgamation sweep, bottom-up union of the children's structures for the row lists.
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

// Postorder of the forest `parent`, children visited by ascending key (key = weight clamped to [0, n-1], or none),
// ties by ascending index; roots by ascending index.  That is the order cholmod_postorder produces
// (CHOLMOD/Cholesky/cholmod_postorder.c: its bucket lists come out as exactly this ordering of siblings), here
// through a child table in compressed form: nodes are bucket-sorted by key once (stable in the index), dealt to
// their parents' slots in that order, and the walk keeps one cursor per open node.
Int postorder(Int n, const std::vector<Int> &parent, const Int *weight, std::vector<Int> &post)
{
    post.assign((size_t)n, 0);
    // nodes in sibling order
    std::vector<Int> order((size_t)n);
    if (!weight) {
        for (Int j = 0; j < n; j++) order[(size_t)j] = j;
    } else {
        std::vector<Int> slot((size_t)n + 1, 0);
        auto key = [&](Int j) { return std::min<Int>(std::max<Int>(weight[j], 0), n - 1); };
        for (Int j = 0; j < n; j++) slot[(size_t)key(j) + 1]++;
        for (Int w = 0; w < n; w++) slot[(size_t)w + 1] += slot[(size_t)w];
        for (Int j = 0; j < n; j++) order[(size_t)slot[(size_t)key(j)]++] = j;
    }
    // child table: kids[kid_lo[p] .. kid_lo[p+1]) in sibling order
    std::vector<Int> kid_lo((size_t)n + 1, 0), kids((size_t)n);
    auto has_parent = [&](Int j) { return parent[(size_t)j] >= 0 && parent[(size_t)j] < n; };
    for (Int j = 0; j < n; j++) if (has_parent(j)) kid_lo[(size_t)parent[(size_t)j] + 1]++;
    for (Int p = 0; p < n; p++) kid_lo[(size_t)p + 1] += kid_lo[(size_t)p];
    {
        std::vector<Int> fill(kid_lo.begin(), kid_lo.end() - 1);
        for (Int j : order) if (has_parent(j)) kids[(size_t)fill[(size_t)parent[(size_t)j]]++] = j;
    }
    // walk: a node is numbered when its cursor has run past its last child
    std::vector<Int> open, cursor((size_t)n);
    open.reserve(64);
    Int count = 0;
    for (Int r = 0; r < n; r++) {
        if (parent[(size_t)r] != EMPTY) continue;
        cursor[(size_t)r] = kid_lo[(size_t)r];
        open.push_back(r);
        while (!open.empty()) {
            const Int v = open.back();
            if (cursor[(size_t)v] < kid_lo[(size_t)v + 1]) {
                const Int c = kids[(size_t)cursor[(size_t)v]++];
                cursor[(size_t)c] = kid_lo[(size_t)c];
                open.push_back(c);
            } else {
                post[(size_t)count++] = v;
                open.pop_back();
            }
        }
    }
    return count;
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
    R->snnz = S.p[n];

    // ---- supernodes.  What must come out is what cholmod_super_symbolic2 computes for this (Parent, ColCount)
    // (CHOLMOD/Supernodal/cholmod_super_symbolic.c, for_whom = Cholesky, no GPU caps); how it is computed is ours:
    //   * chains: column j continues the supernode of j-1 iff j is the parent of j-1, its only child, and
    //     their structures nest (one row fewer) -- the fundamental supernodes;
    //   * relaxed amalgamation as a right-to-left sweep over CONTIGUOUS groups: a fundamental supernode can only
    //     ever be absorbed by its left neighbour, so every group is a range [lead, end) of fundamental supernodes and
    //     "the group of s's parent is led by s+1" is just  parent(s) < end(s+1)  -- no merge forest is needed;
    //   * row structures bottom-up: struct(s) = own columns + rows of S below them + the children's structures
    //     beyond the own columns (children precede parents), collected with a marker and sorted.
    struct Group { Int first_col, ncols, nrows, zeros, end; };    // end: one past the last fundamental supernode
    std::vector<Group> G;
    std::vector<Int> fparent;                                     // parent of a fundamental supernode (index into G)
    {
        std::vector<Int> nkids((size_t)n, 0), fund_of((size_t)n);
        for (Int j = 0; j < n; j++) if (Parent[j] != EMPTY) nkids[(size_t)Parent[j]]++;
        for (Int j = 0; j < n; j++) {
            const bool chain = j > 0 && Parent[j - 1] == j && nkids[(size_t)j] <= 1 && ColCount[j - 1] == ColCount[j] + 1;
            if (chain) G.back().ncols++;
            else G.push_back(Group{j, 1, ColCount[j], 0, (Int)G.size() + 1});
            fund_of[(size_t)j] = (Int)G.size() - 1;
        }
        fparent.resize(G.size());
        for (size_t f = 0; f < G.size(); f++) {
            const Int top = Parent[G[f].first_col + G[f].ncols - 1];
            fparent[f] = (top == EMPTY) ? EMPTY : fund_of[(size_t)top];
        }
    }
    // absorb(c, p): may group c take in its right neighbour p (c's parent group)?  On yes, zeros_out = explicit
    // zeros the merged supernode is accounted with.  The arithmetic (incl. which branches do NOT add the new zeros)
    // follows the reference's decision exactly, since the structure must be bit-identical to CHOLMOD's.
    const double entry_cap = (double)std::numeric_limits<Int>::max() / sizeof(double);
    auto absorb = [&](const Group &c, const Group &p, Int &zeros_out) {
        const Int width = c.ncols + p.ncols;
        zeros_out = p.zeros;
        if (width <= nrelax[0]) return true;
        // columns of c are padded down to p's structure: per column (rows of p + ncols of c) - rows of c
        const double pad = (double)c.ncols * ((double)p.nrows + (double)c.ncols - (double)c.nrows);
        if (pad == 0) return true;
        const double w = (double)width;
        const double entries = w * (w + 1) / 2 + w * ((double)p.nrows - (double)p.ncols);
        const double frac = ((double)p.zeros + pad) / entries;
        zeros_out = p.zeros + c.ncols * (p.nrows + c.ncols - c.nrows);
        const bool sparse_enough = (width <= nrelax[1] && frac < zrelax[0]) || (width <= nrelax[2] && frac < zrelax[1]) ||
                                   frac < zrelax[2];
        return sparse_enough && entries < entry_cap;
    };
    const Int nfund = (Int)G.size();
    std::vector<uint8_t> absorbed((size_t)nfund, 0);
    for (Int f = nfund - 2; f >= 0; f--) {
        if (fparent[(size_t)f] == EMPTY || fparent[(size_t)f] >= G[(size_t)f + 1].end) continue;   // parent not in the next group
        Int z = 0;
        if (!absorb(G[(size_t)f], G[(size_t)f + 1], z)) continue;
        Group &c = G[(size_t)f];
        const Group &p = G[(size_t)f + 1];
        c.nrows = c.ncols + p.nrows;
        c.ncols += p.ncols;
        c.zeros = z;
        c.end = p.end;
        absorbed[(size_t)f + 1] = 1;
    }
    std::vector<Int> Super, SuperMap((size_t)n), Sparent;
    R->Lpi.assign(1, 0);
    R->Lpx.assign(1, 0);
    for (Int f = 0; f < nfund; f++) {
        if (absorbed[(size_t)f]) continue;
        const Group &g = G[(size_t)f];
        const Int s = (Int)Super.size();
        Super.push_back(g.first_col);
        for (Int k = g.first_col; k < g.first_col + g.ncols; k++) SuperMap[(size_t)k] = s;
        R->Lpi.push_back(R->Lpi.back() + g.nrows);
        R->Lpx.push_back(R->Lpx.back() + g.ncols * g.nrows);
    }
    const Int nsuper = (Int)Super.size();
    Super.push_back(n);
    Sparent.assign((size_t)nsuper, EMPTY);
    for (Int s = 0; s < nsuper; s++) {
        const Int top = Parent[Super[(size_t)s + 1] - 1];
        if (top != EMPTY) Sparent[(size_t)s] = SuperMap[(size_t)top];
    }
    R->nsuper = nsuper;
    R->Super = Super;
    R->ssize = std::max<Int>(1, R->Lpi.back());
    R->xsize = std::max<Int>(1, R->Lpx.back());
    R->Ls.assign((size_t)R->ssize, 0);
    {
        // child table of the supernodal tree, then the bottom-up union
        std::vector<Int> kid_lo((size_t)nsuper + 1, 0), kids((size_t)nsuper);
        for (Int s = 0; s < nsuper; s++) if (Sparent[(size_t)s] != EMPTY) kid_lo[(size_t)Sparent[(size_t)s] + 1]++;
        for (Int s = 0; s < nsuper; s++) kid_lo[(size_t)s + 1] += kid_lo[(size_t)s];
        {
            std::vector<Int> fill(kid_lo.begin(), kid_lo.end() - 1);
            for (Int s = 0; s < nsuper; s++) if (Sparent[(size_t)s] != EMPTY) kids[(size_t)fill[(size_t)Sparent[(size_t)s]]++] = s;
        }
        std::vector<Int> seen_by((size_t)n, EMPTY);
        for (Int s = 0; s < nsuper; s++) {
            const Int k1 = Super[(size_t)s], k2 = Super[(size_t)s + 1];
            Int *rows = R->Ls.data() + R->Lpi[(size_t)s];
            const Int cap = R->Lpi[(size_t)s + 1] - R->Lpi[(size_t)s];
            Int cnt = 0;
            for (Int k = k1; k < k2; k++) rows[cnt++] = k;
            auto take = [&](Int i) {
                if (i < k2 || seen_by[(size_t)i] == s) return true;
                if (cnt >= cap) return false;
                seen_by[(size_t)i] = s;
                rows[cnt++] = i;
                return true;
            };
            bool fits = true;
            for (Int k = k1; k < k2 && fits; k++)
                for (Int q = S.p[(size_t)k]; q < S.p[(size_t)k + 1] && fits; q++) fits = take(S.i[(size_t)q]);
            for (Int c = kid_lo[(size_t)s]; c < kid_lo[(size_t)s + 1] && fits; c++) {
                const Int kid = kids[(size_t)c];
                const Int *lo = R->Ls.data() + R->Lpi[(size_t)kid] + (Super[(size_t)kid + 1] - Super[(size_t)kid]);
                const Int *hi = R->Ls.data() + R->Lpi[(size_t)kid + 1];
                for (const Int *r = std::lower_bound(lo, hi, k2); r < hi && fits; r++) fits = take(*r);
            }
            if (!fits || cnt != cap) { delete R; return MI355_INVALID; }        // column counts inconsistent with the pattern
            std::sort(rows + (k2 - k1), rows + cnt);
        }
    }
    // largest update workspace any descendant would need in the reference (Common->maxcsize: rows from its first row
    // in an ancestor to its end, times its rows inside that ancestor) and the longest sub-diagonal row list (maxesize)
    Int maxcsize = 1, maxesize = 1;
    for (Int d = 0; d < nsuper; d++) {
        const Int *rows = R->Ls.data();
        const Int stop = R->Lpi[(size_t)d + 1];
        Int at = R->Lpi[(size_t)d] + (Super[(size_t)d + 1] - Super[(size_t)d]);
        maxesize = std::max(maxesize, stop - at);
        while (at < stop) {
            const Int anc = SuperMap[(size_t)rows[at]];
            Int run_end = at + 1;
            while (run_end < stop && SuperMap[(size_t)rows[run_end]] == anc) run_end++;
            maxcsize = std::max(maxcsize, (stop - at) * (run_end - at));
            at = run_end;
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
