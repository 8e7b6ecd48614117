// schedule.cpp -- see schedule.hpp.  Pure host C++.
#include "schedule.hpp"
#include <cmath>
#include <algorithm>
#include <unordered_map>
#include <unordered_set>
#include <cstdio>

namespace mi355 {

Tuning &tuning()
{
    static Tuning t;
    return t;
}

int64_t gemm_tiles(int32_t M, int32_t N, int32_t mode, int32_t tile)
{
    return gemm_tiles(M, N, mode, tile, tile);
}

// tiles of a task that hold at least one entry to compute: all of them (GEMM_OVERWRITE), or those of the
// lower trapezoid (some row index >= some column index inside the tile)
int64_t gemm_tiles(int32_t M, int32_t N, int32_t mode, int32_t tile_m, int32_t tile_n)
{
    if (M <= 0 || N <= 0) return 0;
    const int64_t MT = (M + tile_m - 1) / tile_m, NT = (N + tile_n - 1) / tile_n;
    if (mode == GEMM_OVERWRITE || M < N) return MT * NT;      // lower-trapezoid tasks have M >= N
    int64_t count = 0;
    for (int64_t tm = 0; tm < MT; tm++) {
        const int64_t last_row = std::min<int64_t>(tm * tile_m + tile_m, M) - 1;
        count += std::min<int64_t>(NT, last_row / tile_n + 1);
    }
    return count;
}

std::string validate_symbolic(const Symbolic &S)
{
    char buf[256];
    if (S.n < 0 || S.nsuper < 0) return "negative dimension";
    if (S.n == 0 || S.nsuper == 0) return (S.n == 0 && S.nsuper == 0) ? "" : "n/nsuper mismatch";
    if (!S.Super || !S.Lpi || !S.Lpx || !S.Ls) return "null symbolic array";
    if (S.n >= (int64_t(1) << 31) - 1) return "too large: n does not fit the 32-bit row indices of the device kernels";
    if (S.Super[0] != 0 || S.Super[S.nsuper] != S.n) return "Super does not span 0..n";
    if (S.Lpi[0] != 0 || S.Lpx[0] != 0) return "Lpi[0]/Lpx[0] must be 0";
    for (int64_t s = 0; s < S.nsuper; s++) {
        int64_t nscol = S.Super[s + 1] - S.Super[s];
        int64_t nsrow = S.Lpi[s + 1] - S.Lpi[s];
        if (nscol <= 0 || nsrow < nscol) {
            snprintf(buf, sizeof buf, "supernode %lld: nscol %lld nsrow %lld", (long long)s,
                     (long long)nscol, (long long)nsrow);
            return buf;
        }
        if (S.Lpx[s + 1] - S.Lpx[s] != nscol * nsrow) {
            snprintf(buf, sizeof buf, "supernode %lld: Lpx does not match nsrow*nscol", (long long)s);
            return buf;
        }
        const int64_t *rows = S.Ls + S.Lpi[s];
        for (int64_t k = 0; k < nsrow; k++) {
            int64_t want_min = (k < nscol) ? S.Super[s] + k : rows[k - 1] + 1;
            if ((k < nscol && rows[k] != want_min) || rows[k] < want_min || rows[k] >= S.n) {
                snprintf(buf, sizeof buf, "supernode %lld: row list invalid at %lld", (long long)s,
                         (long long)k);
                return buf;
            }
        }
    }
    if (S.Lpi[S.nsuper] > S.ssize || S.Lpx[S.nsuper] > S.xsize) return "ssize/xsize too small";
    return "";
}

namespace {

struct LevelSteps {
    // per iteration it: tasks of all supernodes of this level
    std::vector<std::vector<PotrfTask>> potrf;
    std::vector<std::vector<GemmTask>> trsm, trail;
    std::vector<std::vector<GemmTask>> rest;      // look-ahead: the far part of the outermost trailing updates
    std::vector<GemmTask> updates;
    // DistTop: is potrf[it][i] (and its trsm task) this rank's; blocks that complete at iteration it
    std::vector<std::vector<int8_t>> mine;
    std::vector<std::vector<BcastPoint>> bcast;
    // panel groups that START at iteration it; blk[b] = position of the group's b-th block in potrf[it + b]
    struct Group { PanelGroup g; size_t blk[4]; };
    std::vector<std::vector<Group>> pgroup;
    void ensure(size_t it)
    {
        if (potrf.size() <= it) {
            potrf.resize(it + 1); trsm.resize(it + 1); trail.resize(it + 1); rest.resize(it + 1); mine.resize(it + 1); bcast.resize(it + 1);
            pgroup.resize(it + 1);
        }
    }
};

GemmTask make_gemm(int64_t a, int64_t b, int64_t c, int64_t rm, int64_t lda, int64_t ldb, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, int32_t mode)
{
    GemmTask t;
    t.a_off = a; t.b_off = b; t.c_off = c; t.rm_off = rm;
    t.lda = (int32_t)lda; t.ldb = (int32_t)ldb; t.ldc = (int32_t)ldc;
    t.M = (int32_t)M; t.N = (int32_t)N; t.K = (int32_t)K;
    t.mode = mode; t.big = 0; t.tile0 = 0; t.fuse_potrf = -1; t.pad = 0;
    return t;
}

}  // namespace

// tiles of one task in super-tile order: G x G groups of tiles (G = tuning().supertile; 1 = plain
// row-major over the lower trapezoid), row-major over the groups
void append_tiles(std::vector<Schedule::TileEntry> &seq, int32_t task, const GemmTask &t, int tile_m, int tile_n)
{
    const int64_t G = std::max(1, tuning().supertile);
    const int64_t MT = (t.M + tile_m - 1) / tile_m, NT = (t.N + tile_n - 1) / tile_n;
    const bool lower = (t.mode != GEMM_OVERWRITE) && t.M >= t.N;
    // columns of tile row tm that hold an entry on or below the diagonal
    auto ncols = [&](int64_t tm) {
        if (!lower) return NT;
        const int64_t last_row = std::min<int64_t>(tm * tile_m + tile_m, t.M) - 1;
        return std::min<int64_t>(NT, last_row / tile_n + 1);
    };
    for (int64_t I = 0; I < MT; I += G)
        for (int64_t J = 0; J < NT; J += G)
            for (int64_t tm = I; tm < std::min(I + G, MT); tm++) {
                const int64_t nc = ncols(tm);
                for (int64_t tn = J; tn < std::min(J + G, nc); tn++) {
                    Schedule::TileEntry e;
                    e.task = task;
                    e.tmtn = (uint32_t)tm | ((uint32_t)tn << 16);
                    seq.push_back(e);
                }
            }
}

// workgroup b runs entry u(b) of the sequence.  With tuning().xcd_deal the sequence is dealt to the
// 8 XCDs in contiguous chunks (workgroups go round-robin over the XCDs, so XCD x sees a contiguous
// eighth of the sequence and the operand panels of neighbouring tiles share its L2).
void deal_tiles(const std::vector<Schedule::TileEntry> &seq, std::vector<Schedule::TileEntry> &tab)
{
    const int64_t n = (int64_t)seq.size(), q = n / 8, r = n % 8;
    const size_t base = tab.size();
    tab.resize(base + (size_t)n);
    for (int64_t b = 0; b < n; b++) {
        int64_t u = b;
        if (tuning().xcd_deal) {
            const int64_t x = b % 8, j = b / 8;
            u = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
        }
        tab[base + (size_t)b] = seq[(size_t)u];
    }
}

namespace {

// append the GEMM tasks as up to two steps (big-tile tasks first, then small-tile tasks: the two
// kernels of one step touch disjoint targets or use atomics, so their order is free); assigns the
// tile0 prefix sums
void flush_gemm(Schedule &out, std::vector<GemmTask> &tasks, int level, const Symbolic &S, int32_t trsm = 0,
                int32_t flags = 0)
{
    if (tasks.empty()) return;
    const Tuning &tu = tuning();
    auto is_tiny = [&](const GemmTask &t) {
        return tu.tiny_max > 0 && !S.cplx && !trsm && t.mode != GEMM_OVERWRITE && t.M <= TILE && t.N <= TILE &&
               (int64_t)t.M * t.N * t.K <= tu.tiny_max;
    };
    // launch classes: 1 big-tile tasks, 0 small-tile tasks, 2 tiny tasks (small-tile geometry, STEP_TINY)
    for (int cls : {1, 0, 2}) {
        const int big = cls == 1 ? 1 : 0;
        Step st;
        st.kind = STEP_GEMM; st.level = level; st.big = big; st.trsm = trsm; st.flags = flags | (cls == 2 ? STEP_TINY : 0); st.nsmall = 0;
        st.first = (int64_t)out.gemm.size(); st.count = 0; st.ntiles = 0; st.stiles = 0;
        st.tmap_off = (int64_t)out.gemm_tile_tab.size();
        std::vector<Schedule::TileEntry> seq;
        for (auto &t : tasks) {
            // a complex factor's big tile is 64 x 128 entries: the row threshold counts real rows
            const int is_big = (!S.no_big && (S.cplx ? 2 : 1) * t.M >= tu.big_min_m && t.N >= tu.big_min_n) ? 1 : 0;
            const int c = is_big ? 1 : (is_tiny(t) ? 2 : 0);
            if (c != cls) continue;
            const int tbig = big ? (S.cplx ? 2 : 1) : 0;
            int64_t nt = gemm_tiles(t.M, t.N, t.mode, tile_rows(tbig), tile_cols(tbig));
            if (nt == 0) continue;
            t.big = tbig;
            t.tile0 = st.ntiles;
            st.ntiles += nt;
            out.gemm.push_back(t);
            append_tiles(seq, (int32_t)st.count, t, tile_rows(tbig), tile_cols(tbig));
            {
                const double Mm = t.M, Nn = t.N, Kk = t.K;
                const double fl = (t.mode == GEMM_OVERWRITE) ? Mm * Nn * Kk : (Nn * Nn + 2.0 * (Mm - Nn) * Nn) * Kk;
                if (big) { out.flops_big += fl; out.tiles_big += nt; } else { out.flops_small += fl; out.tiles_small += nt; }
            }
            st.count++;
        }
        if (st.count > 0) {
            deal_tiles(seq, out.gemm_tile_tab);
            out.steps.push_back(st);
        }
    }
}

}  // namespace

bool check_tile_table(const Schedule &sc, const Step &st)
{
    if (st.kind != STEP_GEMM) return true;
    std::vector<std::vector<uint8_t>> seen((size_t)st.count);
    for (int64_t i = 0; i < st.count; i++) {
        const GemmTask &t = sc.gemm[(size_t)(st.first + i)];
        const int tr = tile_rows(t.big), tc = tile_cols(t.big);
        const int64_t MT = (t.M + tr - 1) / tr, NT = (t.N + tc - 1) / tc;
        seen[(size_t)i].assign((size_t)(MT * NT), 0);
    }
    int64_t count = 0;
    for (int64_t b = 0; b < st.ntiles; b++) {
        const Schedule::TileEntry e = sc.gemm_tile_tab[(size_t)(st.tmap_off + b)];
        if (e.task < 0 || e.task >= st.count) return false;
        const GemmTask &t = sc.gemm[(size_t)(st.first + e.task)];
        const int tr = tile_rows(t.big), tc = tile_cols(t.big);
        const int64_t MT = (t.M + tr - 1) / tr, NT = (t.N + tc - 1) / tc;
        const int64_t tm = e.tmtn & 0xffff, tn = e.tmtn >> 16;
        if (tm >= MT || tn >= NT) return false;
        // a tile of a lower-trapezoid task must hold an entry on or below the diagonal
        if (t.mode != GEMM_OVERWRITE && t.M >= t.N && std::min<int64_t>(tm * tr + tr, t.M) - 1 < tn * tc) return false;
        uint8_t &f = seen[(size_t)e.task][(size_t)(tm * NT + tn)];
        if (f) return false;
        f = 1;
        count++;
    }
    int64_t want = 0;
    for (int64_t i = 0; i < st.count; i++) {
        const GemmTask &t = sc.gemm[(size_t)(st.first + i)];
        want += gemm_tiles(t.M, t.N, t.mode, tile_rows(t.big), tile_cols(t.big));
    }
    return count == want && want == st.ntiles;
}

void build_schedule(const Symbolic &S, int64_t s_limit, int64_t nscol2_limit, Schedule &out,
                    const int8_t *mask, int64_t dinv_base, const DistTop *dist)
{
    out = Schedule();
    out.dinv_size = dinv_base;
    const int64_t ns = S.nsuper;
    out.s_limit = s_limit;
    out.nscol2_limit = nscol2_limit;
    if (ns == 0) return;
    const bool truncated = (s_limit < ns);
    // supernodes processed at all: s < s_end
    const int64_t s_end = truncated ? (nscol2_limit > 0 ? s_limit + 1 : s_limit) : ns;

    // column -> supernode (cholmod_super_numeric.c:276-285)
    out.supermap.resize(S.n);
    for (int64_t s = 0; s < ns; s++)
        for (int64_t k = S.Super[s]; k < S.Super[s + 1]; k++) out.supermap[k] = (int32_t)s;

    // supernodal etree: parent = SuperMap[Ls[psi+nscol]] (worker :1226)
    out.parent.assign(ns, -1);
    out.level.assign(ns, 0);
    for (int64_t s = 0; s < ns; s++) {
        int64_t nscol = S.Super[s + 1] - S.Super[s];
        int64_t nsrow = S.Lpi[s + 1] - S.Lpi[s];
        out.max_nscol = std::max(out.max_nscol, nscol);
        out.max_nsrow = std::max(out.max_nsrow, nsrow);
        if (nsrow > nscol) out.parent[s] = out.supermap[S.Ls[S.Lpi[s] + nscol]];
    }
    for (int64_t s = 0; s < ns; s++) {
        int32_t p = out.parent[s];
        if (p >= 0) out.level[p] = std::max(out.level[p], out.level[s] + 1);
    }
    int32_t nlev = 0;
    for (int64_t s = 0; s < s_end; s++) nlev = std::max(nlev, out.level[s] + 1);
    out.nlevels = nlev;

    std::vector<LevelSteps> L(nlev);
    std::unordered_set<int64_t> grouped_blocks;     // diagonal blocks (offset in Lx) that a panel group factorizes

    // DistTop: owner of outer block jb of masked supernode s = (position of s among the masked supernodes of
    // its level + jb * their number + level) mod world -- blocks that are worked on at the same time (same
    // level, same jb) and consecutive blocks of one supernode land on different ranks
    if (dist && (!mask || truncated || dist->world < 1 || dist->rank < 0 || dist->rank >= dist->world)) dist = nullptr;
    std::vector<int32_t> pos_in_level, n_in_level;
    if (dist) {
        pos_in_level.assign(ns, 0);
        n_in_level.assign(nlev, 0);
        for (int64_t s = 0; s < ns; s++)
            if (mask[s]) pos_in_level[s] = n_in_level[out.level[s]]++;
    }
    auto owner = [&](int64_t s, int64_t jb) {
        const int lev = out.level[s];
        return (int32_t)(((int64_t)pos_in_level[s] + jb * n_in_level[lev] + lev) % dist->world);
    };
    // widest nested block width (the unit of ownership)
    int64_t WB = NB;
    {
        const int64_t w1 = std::max(NB, (tuning().ob / NB) * NB);
        if (w1 > WB) WB = w1;
        const int64_t w2 = (tuning().ob2 / NB) * NB;
        if (w2 > WB && w2 % WB == 0) WB = w2;
        const int64_t w3 = (tuning().ob3 / NB) * NB;
        if (w3 > WB && w3 % WB == 0) WB = w3;
    }

    for (int64_t s = 0; s < s_end; s++) {
        if (mask && !mask[s]) continue;
        const int lev = out.level[s];
        const int64_t nscol = S.Super[s + 1] - S.Super[s];
        const int64_t nsrow = S.Lpi[s + 1] - S.Lpi[s];
        const int64_t psx = S.Lpx[s];
        const int64_t ncol_eff = (truncated && s == s_limit) ? nscol2_limit : nscol;

        // ---- panel: blocked partial Cholesky of the nsrow x ncol_eff panel (a3.9 + a3.10)
        // Right-looking with nested block widths W[0] = NB < W[1] < W[2] ...: when a width-W[i] block
        // [b0, e) is complete, the columns [e, end of the enclosing W[i+1] block) receive its update in ONE
        // GEMM of depth K = e - b0 (the outermost level updates to the end of the supernode).  Each C entry is
        // read and written once per W[i] columns at level i, so the wide fronts do most of their flops at
        // the largest K while the K = NB updates stay inside a W[1]-column window.
        size_t it = 0;
        std::vector<int64_t> W;
        W.push_back(NB);
        {
            const int64_t w1 = std::max(NB, (tuning().ob / NB) * NB);
            if (w1 > W.back()) W.push_back(w1);
            const int64_t w2 = (tuning().ob2 / NB) * NB;
            if (w2 > W.back() && w2 % W.back() == 0) W.push_back(w2);
            const int64_t w3 = (tuning().ob3 / NB) * NB;
            if (w3 > W.back() && w3 % W.back() == 0) W.push_back(w3);
        }
        // panel groups: a window of the first nested width (4 blocks) with at least two blocks is factorized by one
        // STEP_PDIAG + STEP_PTRSM pair at the iteration of its first block
        const int64_t GW = (int64_t)SOLVE_GROUP * NB;
        const bool pg_on = tuning().panel_group > 0 && !S.cplx && !tuning().lookahead && W.size() > 1 && W[1] == GW && (!dist || WB % GW == 0) &&
                           nsrow < (1 << 27);       // (the kernels add 32-bit lane offsets of up to 4 * nsrow entries to uniform bases)
        for (int64_t j = 0; j < ncol_eff; j += NB, it++) {
            const int64_t nb = std::min<int64_t>(NB, ncol_eff - j);
            const int64_t e = j + nb;
            const int64_t below = nsrow - e;
            L[lev].ensure(it);
            const int64_t w0 = (j / GW) * GW, w1 = std::min<int64_t>(w0 + GW, ncol_eff);
            const bool grouped = pg_on && w1 - w0 > NB;
            const size_t it0 = it - (size_t)((j - w0) / NB);
            const bool own_w = !dist || owner(s, j / WB) == dist->rank;     // (a window lies inside one outer block: one owner)
            if (grouped) {
                if (j == w0 && own_w) {
                    LevelSteps::Group G;
                    G.g.a_off = psx + w0 + w0 * nsrow;
                    for (int b = 0; b < 4; b++) { G.g.dinv_off[b] = -1; G.blk[b] = 0; }
                    G.g.wg0 = 0; G.g.lda = (int32_t)nsrow; G.g.ncols = (int32_t)(w1 - w0); G.g.s = (int32_t)s;
                    G.g.col0 = (int32_t)w0; G.g.below = (int32_t)(nsrow - w1); G.g.pad = 0;
                    L[lev].pgroup[it0].push_back(G);
                }
                if (own_w) L[lev].pgroup[it0].back().blk[(size_t)((j - w0) / NB)] = L[lev].potrf[it].size();
                grouped_blocks.insert(psx + j + j * nsrow);
            }
            PotrfTask pt;
            pt.a_off = psx + j + j * nsrow;
            pt.dinv_off = (below > 0 && !grouped) ? 0 : -1;     // slot assigned when the step is flushed; >= 0: a TRSM task follows
            pt.lda = (int32_t)nsrow; pt.nb = (int32_t)nb; pt.s = (int32_t)s; pt.col0 = (int32_t)j;
            pt.ls_off = S.Lpi[s]; pt.stile0 = 0;
            pt.k1 = (int32_t)S.Super[s]; pt.nscol = (int32_t)nscol;
            pt.urow0 = (int32_t)(j + nb);
            // a whole supernode of <= 16 columns: one wave factorizes the block AND solves the rows below it
            pt.fuse_trsm = (tuning().tiny_max > 0 && !S.cplx && ncol_eff <= 16 && below > 0) ? 1 : 0;
            const bool own = !dist || owner(s, j / WB) == dist->rank;
            L[lev].potrf[it].push_back(pt);
            L[lev].mine[it].push_back(own ? 1 : 0);
            if (own) out.flops_panel += (double)nb * nb * nb / 3.0;
            if (dist) {
                // outer block (s, j / WB) is complete after this iteration's TRSM: its columns go to everybody.  With
                // Tuning::dist_pieces they go in pieces of the first nested width (256 columns): a piece is final as
                // soon as its own panels are done -- the rest of the block's chain only touches later columns -- so it
                // travels while the chain runs and only the last piece is left when the block is complete
                const int64_t b0 = (j / WB) * WB;
                const bool block_end = e == std::min<int64_t>(b0 + WB, ncol_eff);
                const int64_t PW = (tuning().dist_pieces && W.size() > 1) ? W[1] : WB;
                const int64_t p0 = std::max<int64_t>(b0, ((e - 1) / PW) * PW);
                if (block_end || e == p0 + PW) {
                    BcastPoint bp;
                    bp.step = bp.wait_step = -1; bp.lo = psx + p0 * nsrow; bp.count = (e - p0) * nsrow;
                    bp.root = owner(s, j / WB); bp.s = (int32_t)s;
                    bp.mine = bp.root == dist->rank ? 1 : 0;
                    bp.last = block_end ? 1 : 0;
                    // (a panel group finishes its window -- POTRFs, the rows below -- at the iteration of its first block)
                    L[lev].bcast[grouped ? it0 : it].push_back(bp);
                }
            }
            if (below <= 0) continue;
            // X = B * inv(L11)^H  (trsm "R","L","C","N", worker :1175)
            GemmTask t = make_gemm(psx + e + j * nsrow, /*b: slot, set later*/ 0,
                                   psx + e + j * nsrow, -1, nsrow, nb, nsrow,
                                   below, nb, nb, GEMM_OVERWRITE);
            if (!grouped) L[lev].trsm[it].push_back(t);
            if (own) out.flops_panel += (double)below * nb * nb;
            // a grouped window's trailing updates (those of the blocks that end where it ends) run right after its
            // STEP_PTRSM, i.e. at the iteration of its first block
            auto &trail_here = grouped ? L[lev].trail[it0] : L[lev].trail[it];
            for (size_t lvl = 0; lvl < W.size(); lvl++) {
                const int64_t b0 = ((e - 1) / W[lvl]) * W[lvl];                       // block of this level ending at e?
                if (std::min<int64_t>(b0 + W[lvl], ncol_eff) != e) break;
                const int64_t enc_end = (lvl + 1 < W.size())
                    ? std::min<int64_t>((b0 / W[lvl + 1]) * W[lvl + 1] + W[lvl + 1], ncol_eff) : ncol_eff;
                if (enc_end <= e) continue;                                           // enclosing block complete too
                const int64_t K = e - b0, Nn = enc_end - e;
                if (grouped && lvl == 0) {
                    // the K = 64 update inside the window: part of the group's kernels (same flops)
                    if (own) out.flops_panel += ((double)Nn * Nn + 2.0 * (double)(below - Nn) * Nn) * (double)K;
                    break;
                }
                if (dist && lvl + 1 == W.size()) {
                    // outermost update, owner-computes: one task per target outer block, kept by its owner
                    // look-ahead (Tuning::dist_lookahead): only the update of the NEXT block (c0 == e) feeds the next
                    // panel chain; the updates of the blocks beyond it are launched after that chain's broadcast has
                    // been issued (flattening below), so that chain and broadcast overlap them
                    for (int64_t c0 = e; c0 < ncol_eff; c0 += WB) {
                        const int64_t c1 = std::min<int64_t>(c0 + WB, ncol_eff);
                        if (owner(s, c0 / WB) != dist->rank) continue;
                        const size_t itq = grouped ? it0 : it;
                        auto &dst = (tuning().dist_lookahead && c0 > e) ? L[lev].rest[itq] : L[lev].trail[itq];
                        dst.push_back(make_gemm(
                            psx + c0 + b0 * nsrow, psx + c0 + b0 * nsrow,
                            psx + c0 + c0 * nsrow, -1, nsrow, nsrow, nsrow,
                            nsrow - c0, c1 - c0, K, GEMM_DIRECT_SUB));
                        const double Nt = (double)(c1 - c0), Mt = (double)(nsrow - c0);
                        out.flops_panel += (Nt * Nt + 2.0 * (Mt - Nt) * Nt) * (double)K;
                    }
                    break;
                }
                if (!own) break;
                if (!dist && tuning().lookahead && lvl > 0 && lvl + 1 == W.size() && e + W[lvl] < ncol_eff) {
                    // look-ahead: columns [e, e1) = the next outer block, on the critical path; [e1, end) = the rest
                    const int64_t e1 = e + W[lvl];
                    L[lev].trail[it].push_back(make_gemm(
                        psx + e + b0 * nsrow, psx + e + b0 * nsrow,
                        psx + e + e * nsrow, -1, nsrow, nsrow, nsrow,
                        below, e1 - e, K, GEMM_DIRECT_SUB));
                    L[lev].rest[it].push_back(make_gemm(
                        psx + e1 + b0 * nsrow, psx + e1 + b0 * nsrow,
                        psx + e1 + e1 * nsrow, -1, nsrow, nsrow, nsrow,
                        nsrow - e1, ncol_eff - e1, K, GEMM_DIRECT_SUB));
                    out.flops_panel += ((double)Nn * Nn + 2.0 * (double)(below - Nn) * Nn) * (double)K;
                    break;
                }
                trail_here.push_back(make_gemm(
                    psx + e + b0 * nsrow, psx + e + b0 * nsrow,
                    psx + e + e * nsrow, -1, nsrow, nsrow, nsrow,
                    below, Nn, K, GEMM_DIRECT_SUB));
                out.flops_panel += ((double)Nn * Nn + 2.0 * (double)(below - Nn) * Nn) * (double)K;
                break;
            }
        }

        // ---- updates pushed by s into its ancestors (a3.4-a3.8), right-looking
        if (truncated && s >= s_limit) continue;     // the repeated supernode pushes nothing
        const int64_t pdi = S.Lpi[s], pdend = S.Lpi[s + 1];
        int64_t p = pdi + nscol;
        while (p < pdend) {
            const int32_t anc = out.supermap[S.Ls[p]];
            int64_t q = p;
            while (q < pdend && out.supermap[S.Ls[q]] == anc) q++;
            const int64_t ndrow1 = q - p, ndrow2 = pdend - p;
            const bool keep = !truncated || anc < s_limit || (anc == s_limit && nscol2_limit > 0);
            if (keep) {
                UpdatePair up;
                up.d = (int32_t)s; up.s = anc;
                up.ls_src = p; up.ls_dst = S.Lpi[anc];
                up.ndrow1 = (int32_t)ndrow1; up.ndrow2 = (int32_t)ndrow2;
                up.nsrow = (int32_t)(S.Lpi[anc + 1] - S.Lpi[anc]);
                up.rm_off = out.relmap_size;
                out.relmap_size += ndrow2;
                out.pairs.push_back(up);
                if (dist) {
                    // owner-computes: the rows [p, q) of s are cut where their column of anc enters another
                    // outer block; each piece updates ONE block of anc and runs on that block's owner
                    int64_t p1 = p;
                    while (p1 < q) {
                        const int64_t tb = (S.Ls[p1] - S.Super[anc]) / WB;
                        int64_t q1 = p1;
                        while (q1 < q && (S.Ls[q1] - S.Super[anc]) / WB == tb) q1++;
                        if (owner(anc, tb) == dist->rank) {
                            const int64_t a1 = psx + (p1 - pdi), n1 = q1 - p1, m1 = pdend - p1;
                            L[lev].updates.push_back(make_gemm(a1, a1, S.Lpx[anc], up.rm_off + (p1 - p), nsrow, nsrow,
                                                               up.nsrow, m1, n1, nscol, GEMM_SCATTER_SUB));
                            out.flops_update += (double)n1 * n1 * nscol + 2.0 * (double)(m1 - n1) * n1 * nscol;
                            out.bytes_assembly += (double)n1 * m1 * 8.0;
                        }
                        p1 = q1;
                    }
                    p = q;
                    continue;
                }
                const int64_t a = psx + (p - pdi);
                L[lev].updates.push_back(make_gemm(a, a, S.Lpx[anc], up.rm_off, nsrow, nsrow, up.nsrow,
                                                   ndrow2, ndrow1, nscol, GEMM_SCATTER_SUB));
                out.flops_update += (double)ndrow1 * ndrow1 * nscol
                                  + 2.0 * (double)(ndrow2 - ndrow1) * ndrow1 * nscol;
                out.bytes_assembly += (double)ndrow1 * ndrow2 * 8.0;
            }
            p = q;
        }
    }

    // ---- flatten into launch order
    size_t awaited = 0;                             // DistTop: broadcasts [0, awaited) have their wait point
    std::unordered_map<int64_t, int64_t> fused_by;  // diagonal block (its offset in Lx) -> the trailing task of the previous
                                                    // iteration that completes it and factorizes it in its tile (0, 0)
    const bool truncated_any = false;               // (the truncated replay fuses as well: same tasks, fewer of them)
    for (int lev = 0; lev < nlev; lev++) {
        LevelSteps &ls = L[lev];
        std::vector<GemmTask> deferred;             // DistTop look-ahead: far updates waiting for the next broadcast point
        // Dinv slots of the whole level first, in iteration order (a panel group that starts at iteration it needs the
        // slots of its later blocks).  Every block keeps its inverse for the whole life of the factor (the solves
        // reuse it): slots are global, sum nb^2 <= 64*n entries.
        std::vector<std::vector<uint8_t>> has_trsm(ls.potrf.size());
        for (size_t it = 0; it < ls.potrf.size(); it++)
            for (PotrfTask &pt : ls.potrf[it]) {
                has_trsm[it].push_back(pt.dinv_off >= 0 ? 1 : 0);
                pt.dinv_off = out.dinv_size;
                out.dinv_size += (int64_t)pt.nb * pt.nb;
            }
        auto nolaunch = [&](const PotrfTask &pt) { return fused_by.count(pt.a_off) != 0 || grouped_blocks.count(pt.a_off) != 0; };
        for (size_t it = 0; it < ls.potrf.size(); it++) {
            // the trsm task i pairs with the i-th potrf task that has one (same push order)
            size_t ti = 0;
            std::vector<PotrfTask> pts;                 // the tasks this schedule runs (all, or this rank's)
            std::vector<GemmTask> trs;
            for (size_t i = 0; i < ls.potrf[it].size(); i++) {
                PotrfTask &pt = ls.potrf[it][i];
                const bool own = ls.mine[it][i] != 0;
                if (has_trsm[it][i]) {
                    ls.trsm[it][ti].b_off = pt.dinv_off;
                    if (own && !pt.fuse_trsm) trs.push_back(ls.trsm[it][ti]);
                    ti++;
                }
                if (own) pts.push_back(pt);
            }
            if (!fused_by.empty() || !grouped_blocks.empty())
                std::stable_partition(pts.begin(), pts.end(), [&](const PotrfTask &pt) { return !nolaunch(pt); });
            int64_t stiles = 0;
            const int64_t tmap_off = (int64_t)out.solve_tile_task.size();
            int32_t ti32 = 0;
            for (auto &pt : pts) {
                const int64_t below = (int64_t)pt.lda - (pt.col0 + pt.nb);
                const int64_t nt = (below + SOLVE_ROWS - 1) / SOLVE_ROWS;
                pt.stile0 = stiles;
                stiles += nt;
                out.solve_tile_task.insert(out.solve_tile_task.end(), (size_t)nt, ti32);
                ti32++;
            }
            Step st;
            st.kind = STEP_POTRF; st.level = lev; st.big = 0; st.trsm = 0; st.flags = 0; st.nsmall = 0;
            st.first = (int64_t)out.potrf.size();
            st.count = (int64_t)pts.size();
            // blocks that the previous iteration's trailing update completes inside k_gemm128 (the workgroup of tile
            // (0, 0) factorizes the block right after its stores): they come last in the step and have no launch here
            for (size_t i = 0; i < pts.size(); i++) {
                if (fused_by.count(pts[i].a_off)) {
                    out.gemm[(size_t)fused_by[pts[i].a_off]].fuse_potrf = (int32_t)(st.first + (int64_t)i);
                    st.nfused++;
                } else if (grouped_blocks.count(pts[i].a_off)) { st.nfused++; st.ngrouped++; }
            }
            if (tuning().tiny_max > 0 && !S.cplx) {
                out.potrf_order.resize(out.potrf.size());           // steps of other schedules' making: none
                for (int pass = 0; pass < 3; pass++)
                    for (size_t i = 0; i < pts.size(); i++) {
                        const int cls = nolaunch(pts[i]) ? 2 : (pts[i].nb <= 16 ? 0 : 1);
                        if (cls == pass) out.potrf_order.push_back((int32_t)i);
                    }
                for (auto &pt : pts) st.nsmall += (pt.nb <= 16 && !nolaunch(pt)) ? 1 : 0;
            }
            st.ntiles = st.count;
            st.stiles = stiles;
            st.tmap_off = tmap_off;
            out.potrf.insert(out.potrf.end(), pts.begin(), pts.end());
            if (st.count > 0) out.steps.push_back(st);
            flush_gemm(out, trs, lev, S, 1);
            if (!ls.pgroup[it].empty()) {
                // the panel groups that start here: their diagonal windows, then the rows below them
                Step sd;
                sd.kind = STEP_PDIAG; sd.level = lev; sd.big = 0; sd.trsm = 0; sd.flags = 0; sd.nsmall = 0;
                sd.first = (int64_t)out.pgroups.size(); sd.count = (int64_t)ls.pgroup[it].size();
                sd.ntiles = sd.count; sd.stiles = 0; sd.tmap_off = 0;
                Step sr = sd;
                sr.kind = STEP_PTRSM; sr.tmap_off = (int64_t)out.pg_wg_task.size();
                // rows of a panel per workgroup: 128 or 64 (k_panel_trsm_reg: the rows in registers), 32 or 16 (k_panel_trsm: in LDS)
                const int pgv = tuning().panel_group;
                const int64_t rows = pgv >= 128 || pgv < 16 ? 128 : (pgv >= 64 ? 64 : (pgv >= 32 ? 32 : 16));
                sr.ngrouped = (int32_t)rows;
                int64_t wgs = 0;
                int32_t gi = 0;
                for (LevelSteps::Group &G : ls.pgroup[it]) {
                    for (int b = 0; b * NB < G.g.ncols; b++) G.g.dinv_off[b] = ls.potrf[it + (size_t)b][G.blk[b]].dinv_off;
                    G.g.wg0 = wgs;
                    const int64_t nw = ((int64_t)G.g.below + rows - 1) / rows;
                    out.pg_wg_task.insert(out.pg_wg_task.end(), (size_t)nw, gi++);
                    wgs += nw;
                    out.pgroups.push_back(G.g);
                }
                sr.ntiles = wgs;
                out.steps.push_back(sd);
                if (wgs > 0) out.steps.push_back(sr);
            }
            bool block_end = false;
            for (BcastPoint bp : ls.bcast[it]) {        // finished blocks (or pieces), before anybody's update reads them
                bp.step = bp.wait_step = (int64_t)out.steps.size();
                block_end = block_end || bp.last;
                out.bcasts.push_back(bp);
            }
            if (dist) {
                // the broadcasts of this iteration are issued here.  Where an outer block ends: the far updates of the
                // PREVIOUS block end run next (they read blocks that arrived long ago), and only then must everything
                // issued since -- this block's earlier pieces and its last one -- have arrived
                if (block_end) {
                    if (!deferred.empty()) {
                        flush_gemm(out, deferred, lev, S, 0, 0);
                        deferred.clear();
                    }
                    for (size_t k = awaited; k < out.bcasts.size(); k++) out.bcasts[k].wait_step = (int64_t)out.steps.size();
                    awaited = out.bcasts.size();
                }
                flush_gemm(out, ls.trail[it], lev, S, 0, 0);
                deferred.insert(deferred.end(), ls.rest[it].begin(), ls.rest[it].end());
                continue;
            }
            // an iteration that ends an outer block: its update of the next block waits for the previous far
            // part (same target columns), its own far part then goes to the side stream
            const size_t g0 = out.gemm.size();
            flush_gemm(out, ls.trail[it], lev, S, 0, ls.rest[it].empty() ? 0 : STEP_JOIN);
            fused_by.clear();
            if (tuning().fuse_potrf && !S.cplx && !tuning().lookahead && !truncated_any && tuning().panel_group <= 0) {
                // The one trailing task of a supernode at this iteration is the last writer of its next diagonal block
                // (target origin c_off = that block): routed to the big-tile kernel, its tile (0, 0) holds the block
                for (size_t g = g0; g < out.gemm.size(); g++) {
                    const GemmTask &t = out.gemm[g];
                    if (t.big == 1 && t.mode == GEMM_DIRECT_SUB) fused_by[t.c_off] = (int64_t)g;
                }
            }
            flush_gemm(out, ls.rest[it], lev, S, 0, STEP_SIDE);
        }
        fused_by.clear();
        if (!deferred.empty()) flush_gemm(out, deferred, lev, S, 0, 0);
        for (size_t k = awaited; k < out.bcasts.size(); k++) out.bcasts[k].wait_step = (int64_t)out.steps.size();
        awaited = out.bcasts.size();
        if (tuning().deterministic && !dist && ls.updates.size() > 1) {
            // Deterministic mode (tuning key 27): the scatter updates of a level add into their ancestors with FP64
            // atomics, and several descendants of the level may hit the same entry of an ancestor at once, in an order
            // that differs from run to run (last bits of L).  Here the updates of the level are dealt into ROUNDS: round
            // r holds, for every ancestor, the r-th of the level's descendants that update it (descendants in index
            // order).  Inside a round every entry of Lx receives at most one addition (one task per target panel; the
            // relative map of a pair is injective), the rounds are launched one after the other, so every entry sees
            // its additions in one fixed order and two runs give identical bits -- as the reference's sequential loop
            // over the descendants does (t_cholmod_super_numeric_worker.c:588-967).  Cost: as many launch groups per
            // level as the busiest ancestor has descendants on it.
            std::stable_sort(ls.updates.begin(), ls.updates.end(), [](const GemmTask &a, const GemmTask &b) {
                return a.c_off != b.c_off ? a.c_off < b.c_off : a.a_off < b.a_off; });   // (a_off grows with the descendant)
            std::vector<std::vector<GemmTask>> rounds;
            size_t r = 0;
            for (size_t k = 0; k < ls.updates.size(); k++) {
                r = (k > 0 && ls.updates[k].c_off == ls.updates[k - 1].c_off) ? r + 1 : 0;
                if (rounds.size() <= r) rounds.resize(r + 1);
                rounds[r].push_back(ls.updates[k]);
            }
            for (auto &rd : rounds) flush_gemm(out, rd, lev, S, 0, STEP_JOIN);
        } else
        flush_gemm(out, ls.updates, lev, S, 0, STEP_JOIN);
    }
}

void build_solve_groups(const Schedule &sc, SolveSchedule &out, int kmax)
{
    out = SolveSchedule();
    out.kmax = kmax;
    // the POTRF steps of one level are its iterations it = 0, 1, 2, ... in order; iteration it holds the
    // block at columns [NB*it, NB*it + nb) of every supernode of the level that is that wide
    size_t i = 0;
    const size_t nst = sc.steps.size();
    while (i < nst) {
        if (sc.steps[i].kind != STEP_POTRF) { i++; continue; }
        const int32_t lev = sc.steps[i].level;
        std::vector<const Step *> its;
        while (i < nst && sc.steps[i].level == lev) {
            if (sc.steps[i].kind == STEP_POTRF) its.push_back(&sc.steps[i]);
            i++;
        }
        for (size_t g0 = 0; g0 < its.size(); g0 += SOLVE_GROUP) {
            SolveStep st;
            st.gfirst = (int64_t)out.groups.size(); st.gcount = 0;
            st.ufirst = (int64_t)out.utasks.size(); st.ucount = 0;
            st.stiles = 0; st.tmap_off = (int64_t)out.tile_task.size();
            st.maxblk = 1; st.pad = 0;
            st.nufirst = (int64_t)out.nutasks.size(); st.nucount = 0;
            st.nstiles = 0; st.ntmap_off = (int64_t)out.ntile_task.size();
            st.wfirst = (int64_t)out.wtasks.size(); st.wcount = 0;
            st.wmaxblk = 1; st.wsplit = 0;
            // groups: one per supernode present at the first iteration of the range
            std::vector<std::pair<int32_t, int64_t>> by_s;         // (supernode, group index), sorted by supernode
            const Step &f = *its[g0];
            for (int64_t ti = f.first; ti < f.first + f.count; ti++) {
                const PotrfTask &pt = sc.potrf[(size_t)ti];
                SolveGroupTask gt;
                gt.a_off = pt.a_off;
                for (int b = 0; b < SOLVE_GROUP; b++) gt.dinv_off[b] = 0;
                gt.dinv_off[0] = pt.dinv_off;
                gt.lda = pt.lda;
                gt.ncols = std::min<int32_t>(SOLVE_GROUP * NB, pt.nscol - pt.col0);
                gt.x0 = pt.k1 + pt.col0;
                gt.pad = 0;
                st.maxblk = std::max<int32_t>(st.maxblk, (gt.ncols + NB - 1) / NB);
                by_s.push_back({pt.s, (int64_t)out.groups.size()});
                out.groups.push_back(gt);
                st.gcount++;
            }
            std::sort(by_s.begin(), by_s.end());
            // update tasks, group-major: collect the blocks of each group over the iterations of the range
            std::vector<std::vector<PotrfTask>> blocks((size_t)st.gcount);
            for (size_t b = 0; b < (size_t)SOLVE_GROUP && g0 + b < its.size(); b++) {
                const Step &sb = *its[g0 + b];
                for (int64_t ti = sb.first; ti < sb.first + sb.count; ti++) {
                    const PotrfTask &pt = sc.potrf[(size_t)ti];
                    auto itp = std::lower_bound(by_s.begin(), by_s.end(), std::make_pair(pt.s, (int64_t)-1));
                    const int64_t gi = itp->second;
                    out.groups[(size_t)gi].dinv_off[b] = pt.dinv_off;
                    blocks[(size_t)(gi - st.gfirst)].push_back(pt);
                }
            }
            // Contraction depth of this step's wide tasks (tuning key 33).  A workgroup of the wide kernels fetches the x image of
            // its task (ncols x 64 right-hand sides: 128 KB at 256 columns) and its tiles of L (128 KB each), and ONE CU draws
            // ~25 GB/s: a step of few tiles -- the tail of every supernode's chain -- is bound by that fetch, not by the chip.
            // Such a step is cut along the columns too (64 or 128 per task) while its tiles still fit one round of workgroups.
            int kstep = kmax;
            if (kmax > NB && tuning().solve_wide_ksplit > 0) {
                int64_t pre = 0;
                for (int64_t gi = 0; gi < st.gcount; gi++) {
                    const SolveGroupTask &gt = out.groups[(size_t)(st.gfirst + gi)];
                    if (gt.ncols <= NB || blocks[(size_t)gi].empty()) continue;
                    const PotrfTask &p0 = blocks[(size_t)gi][0];
                    const int64_t below = (int64_t)p0.lda - ((gt.x0 - p0.k1) + gt.ncols);
                    if (below > 0) pre += (below + SOLVE_ROWS - 1) / SOLVE_ROWS * ((gt.ncols + NB - 1) / NB);   // tiles at 64 columns per task
                }
                if (pre > 0 && pre <= tuning().solve_wide_wgs) kstep = NB;
                else if (pre > 0 && pre <= 2 * (int64_t)tuning().solve_wide_wgs && kmax > 2 * NB) kstep = 2 * NB;
            }
            for (int64_t gi = 0; gi < st.gcount; gi++) {
                const SolveGroupTask &gt = out.groups[(size_t)(st.gfirst + gi)];
                // look-ahead (tuning key 30): a step of ONE group with rows beyond the next group's is split -- the first
                // SOLVE_GROUP tiles below the group (the next group's rows, all the next group kernel waits for) go to the
                // per-block lists, whose kernels take ~10 us on a handful of tiles; the rows beyond stay with the wide lists
                // and run beside the next group on a second stream
                int32_t near_tiles = 0;
                if (tuning().solve_lookahead > 0 && kmax > 0 && st.gcount == 1 && gt.ncols > NB && !blocks[(size_t)gi].empty()) {
                    const PotrfTask &p0 = blocks[(size_t)gi][0];
                    if ((int64_t)p0.lda - ((gt.x0 - p0.k1) + gt.ncols) > (int64_t)SOLVE_GROUP * SOLVE_ROWS) near_tiles = SOLVE_GROUP;
                }
                st.wsplit = near_tiles;
                for (PotrfTask pt : blocks[(size_t)gi]) {
                    const int32_t gcol0 = gt.x0 - pt.k1;            // first column of the group inside s
                    pt.urow0 = gcol0 + gt.ncols;
                    // several blocks of s update the same rows of s in one launch: nscol = 0 makes the update
                    // kernels treat every row as shared (atomic adds, row index from Ls)
                    pt.nscol = 0;
                    const int64_t below = (int64_t)pt.lda - pt.urow0;
                    if (below <= 0) continue;
                    const int64_t nt = (below + SOLVE_ROWS - 1) / SOLVE_ROWS;
                    pt.stile0 = st.stiles;
                    st.stiles += nt;
                    out.tile_task.insert(out.tile_task.end(), (size_t)nt, (int32_t)st.ucount);
                    out.utasks.push_back(pt);
                    st.ucount++;
                    if (kmax > 0 && (gt.ncols <= NB || near_tiles > 0)) {
                        // a group of one block: the per-block kernels, on the lists of the wide path (a split step: its near tiles)
                        const int64_t nn = near_tiles > 0 ? std::min<int64_t>(nt, near_tiles) : nt;
                        pt.stile0 = st.nstiles;
                        st.nstiles += nn;
                        out.ntile_task.insert(out.ntile_task.end(), (size_t)nn, (int32_t)st.nucount);
                        out.nutasks.push_back(pt);
                        st.nucount++;
                    }
                }
                if (kmax > 0 && gt.ncols > NB && !blocks[(size_t)gi].empty()) {
                    const PotrfTask &p0 = blocks[(size_t)gi][0];
                    const int32_t gcol0 = gt.x0 - p0.k1;
                    const int32_t skip = near_tiles * SOLVE_ROWS;       // rows of a split step that the per-block lists hold
                    const int32_t below = p0.lda - (gcol0 + gt.ncols) - skip;
                    for (int32_t c0 = 0; below > 0 && c0 < gt.ncols; c0 += kstep) {
                        SolveGroupUpd wt;
                        wt.lda = p0.lda;
                        wt.ncols = std::min<int32_t>(kstep, gt.ncols - c0);
                        wt.x0 = gt.x0 + c0;
                        wt.nrows = below;
                        // gt.a_off = L[gcol0][gcol0]; the first row below the group is local row gcol0 + ncols
                        wt.a_off = gt.a_off + gt.ncols + skip + (int64_t)c0 * p0.lda;
                        wt.ls_off = p0.ls_off + gcol0 + gt.ncols + skip;
                        st.wmaxblk = std::max<int32_t>(st.wmaxblk, (wt.ncols + NB - 1) / NB);
                        out.wtasks.push_back(wt);
                        st.wcount++;
                    }
                }
            }
            // backward sweep: runs of at most `run` tiles, long enough to amortise the flush (a transpose-reduce
            // and 64 atomics into entries every run of the block hits) and short enough to leave ~512 runs
            auto cut = [&](std::vector<SolveRange> &dst, int64_t run) {
                for (int64_t u = 0; u < st.ucount; u++) {
                    const PotrfTask &pt = out.utasks[(size_t)(st.ufirst + u)];
                    const int64_t nt = ((int64_t)pt.lda - pt.urow0 + SOLVE_ROWS - 1) / SOLVE_ROWS;
                    for (int64_t t0 = 0; t0 < nt; t0 += run) {
                        SolveRange r;
                        r.task = (int32_t)u; r.tile0 = (int32_t)t0; r.ntiles = (int32_t)std::min<int64_t>(run, nt - t0); r.pad = 0;
                        dst.push_back(r);
                    }
                }
            };
            st.rfirst = (int64_t)out.ranges.size();
            cut(out.ranges, std::min<int64_t>(32, std::max<int64_t>(2, st.stiles / std::max(1, tuning().solve_bwd_div))));
            st.rcount = (int64_t)out.ranges.size() - st.rfirst;
            st.rmfirst = (int64_t)out.ranges_mma.size();
            cut(out.ranges_mma, std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / std::max(1, tuning().solve_bwd_div_mma))));
            st.rmcount = (int64_t)out.ranges_mma.size() - st.rmfirst;
            st.nrmfirst = (int64_t)out.nranges_mma.size(); st.nrmcount = 0;
            st.wrfirst = (int64_t)out.wranges.size(); st.wrcount = 0;
            if (kmax > 0) {
                const int64_t run = std::min<int64_t>(16, std::max<int64_t>(1, st.nstiles / std::max(1, tuning().solve_bwd_div_mma)));
                for (int64_t u = 0; u < st.nucount; u++) {
                    const PotrfTask &pt = out.nutasks[(size_t)(st.nufirst + u)];
                    int64_t nt = ((int64_t)pt.lda - pt.urow0 + SOLVE_ROWS - 1) / SOLVE_ROWS;
                    if (st.wsplit > 0) nt = std::min<int64_t>(nt, st.wsplit);
                    for (int64_t t0 = 0; t0 < nt; t0 += run)
                        out.nranges_mma.push_back({(int32_t)u, (int32_t)t0, (int32_t)std::min<int64_t>(run, nt - t0), 0});
                }
                st.nrmcount = (int64_t)out.nranges_mma.size() - st.nrmfirst;
                // wide tasks: one workgroup per CU (the x1 image of a whole group fills the LDS), so the tiles of the step
                // -- weighted by the columns of their task -- are dealt in order to at most solve_wide_wgs workgroups,
                // each a contiguous share that may run over task boundaries (a segment per task; at most WIDE_MAX_RUN = 64
                // tiles per segment, the kernels' row list in LDS).  Cutting every task into runs of one length left the
                // workgroups of a step with 1 : 6 different loads and two rounds of them on the chip.
                double wtotal = 0;
                for (int64_t u = 0; u < st.wcount; u++) {
                    const SolveGroupUpd &wt = out.wtasks[(size_t)(st.wfirst + u)];
                    wtotal += (double)((wt.nrows + 63) / 64) * wt.ncols;
                }
                int64_t wt64 = 0;
                for (int64_t u = 0; u < st.wcount; u++) wt64 += (out.wtasks[(size_t)(st.wfirst + u)].nrows + 63) / 64;
                // look-ahead (tuning key 30 = CUs kept free, 0 = off): a step of one group with one wide task is dealt to that many
                // fewer workgroups, so that the next group's kernel finds CUs while this step's far rows are still being updated
                const bool alone = st.wsplit > 0;
                const int64_t wgs = alone ? std::max<int64_t>(8, (int64_t)tuning().solve_wide_wgs - tuning().solve_lookahead) : tuning().solve_wide_wgs;
                const int64_t G = std::max<int64_t>(1, std::min<int64_t>(wgs, wt64));
                st.wgfirst = (int64_t)out.wwg.size(); st.wgcount = 0;
                if (st.wcount > 0) {
                    out.wwg.push_back(0);
                    double done = 0;                    // weight dealt so far
                    int64_t g = 0;                      // workgroup being filled
                    for (int64_t u = 0; u < st.wcount; u++) {
                        const SolveGroupUpd &wt = out.wtasks[(size_t)(st.wfirst + u)];
                        const int64_t nt = (wt.nrows + 63) / 64;
                        int64_t t0 = 0;
                        while (t0 < nt) {
                            // tiles of this task that still fit workgroup g: up to its share's end, at least one
                            const double end = wtotal * (double)(g + 1) / (double)G;
                            int64_t fit = (int64_t)((end - done) / wt.ncols + 0.5);
                            fit = std::max<int64_t>(1, std::min<int64_t>(fit, std::min<int64_t>(nt - t0, 64)));
                            out.wranges.push_back({(int32_t)u, (int32_t)t0, (int32_t)fit, 0});
                            t0 += fit;
                            done += (double)fit * wt.ncols;
                            if (done >= end - 0.5 * wt.ncols && g + 1 < G) {
                                out.wwg.push_back((int32_t)((int64_t)out.wranges.size() - st.wrfirst));
                                g++;
                            }
                        }
                    }
                    if (out.wwg.back() != (int32_t)((int64_t)out.wranges.size() - st.wrfirst))
                        out.wwg.push_back((int32_t)((int64_t)out.wranges.size() - st.wrfirst));
                    st.wgcount = (int64_t)out.wwg.size() - st.wgfirst - 1;
                }
                st.wrcount = (int64_t)out.wranges.size() - st.wrfirst;
            }
            out.steps.push_back(st);
        }
    }
}

void build_host_copy_plan(const Symbolic &S, const Schedule &sc, int64_t min_entries, std::vector<HostCopyChunk> &out)
{
    out.clear();
    if (S.nsuper <= 0 || S.xsize <= 0) return;
    // steps are in level order: after[l] = number of steps whose level is <= l.  Supernode s (level l) is written by its
    // own panel steps and by the update steps of LOWER levels only, so it is final once after[l] steps have run
    std::vector<int64_t> after((size_t)sc.nlevels + 1, 0);
    for (const Step &st : sc.steps) after[(size_t)st.level]++;
    for (int l = 1; l <= sc.nlevels; l++) after[(size_t)l] += after[(size_t)l - 1];
    const int64_t small = std::max<int64_t>(1, min_entries / 8);
    HostCopyChunk c{S.Lpx[0], S.Lpx[0], 0};
    if (c.lo > 0) out.push_back({0, c.lo, 0});      // (Lpx[0] is 0 in every factor CHOLMOD builds)
    for (int64_t s = 0; s < S.nsuper; s++) {
        const int64_t ready = after[(size_t)sc.level[(size_t)s]];
        const int64_t have = c.hi - c.lo;
        if (have >= min_entries || (have >= small && ready > c.ready_step)) {
            out.push_back(c);
            c = HostCopyChunk{c.hi, c.hi, 0};
        }
        c.hi = S.Lpx[s + 1];
        c.ready_step = std::max(c.ready_step, ready);
    }
    c.hi = S.xsize;
    if (c.hi > c.lo) out.push_back(c);
}

std::string check_host_copy_plan(const Symbolic &S, const Schedule &sc, const std::vector<HostCopyChunk> &plan)
{
    char buf[200];
    int64_t at = 0;
    for (size_t k = 0; k < plan.size(); k++) {
        if (plan[k].lo != at || plan[k].hi <= plan[k].lo) return "host copy plan: the chunks do not tile Lx";
        if (plan[k].ready_step < 0 || plan[k].ready_step > (int64_t)sc.steps.size()) return "host copy plan: ready step out of range";
        at = plan[k].hi;
    }
    if (at != (plan.empty() ? 0 : S.xsize)) return "host copy plan: the chunks do not cover Lx";
    if (plan.empty()) return "";
    // latest step that writes each chunk
    std::vector<int64_t> last_write(plan.size(), -1);
    auto mark = [&](int64_t lo, int64_t hi /* inclusive */, int64_t step) {
        size_t k = (size_t)(std::upper_bound(plan.begin(), plan.end(), lo,
                            [](int64_t v, const HostCopyChunk &c) { return v < c.lo; }) - plan.begin()) - 1;
        for (; k < plan.size() && plan[k].lo <= hi; k++) last_write[k] = std::max(last_write[k], step);
    };
    auto super_of = [&](int64_t off) {          // supernode whose values hold entry off
        return (int64_t)(std::upper_bound(S.Lpx, S.Lpx + S.nsuper + 1, off) - S.Lpx) - 1;
    };
    for (size_t si = 0; si < sc.steps.size(); si++) {
        const Step &st = sc.steps[si];
        for (int64_t ti = st.first; ti < st.first + st.count; ti++) {
            if (st.kind == STEP_POTRF) {
                const PotrfTask &t = sc.potrf[(size_t)ti];
                mark(S.Lpx[t.s], S.Lpx[t.s + 1] - 1, (int64_t)si);         // the block (and, fused, the rows below it)
            } else if (st.kind == STEP_PDIAG || st.kind == STEP_PTRSM) {
                const PanelGroup &g = sc.pgroups[(size_t)ti];
                mark(S.Lpx[g.s], S.Lpx[g.s + 1] - 1, (int64_t)si);         // the window, the rows below it
            } else {
                const GemmTask &t = sc.gemm[(size_t)ti];
                if (t.mode == GEMM_SCATTER_SUB) {
                    const int64_t a = super_of(t.c_off);
                    if (a < 0 || a >= S.nsuper) return "host copy plan: scatter target outside Lx";
                    mark(S.Lpx[a], S.Lpx[a + 1] - 1, (int64_t)si);
                } else {
                    mark(t.c_off, t.c_off + (t.M - 1) + (int64_t)(t.N - 1) * t.ldc, (int64_t)si);
                }
            }
        }
    }
    for (size_t k = 0; k < plan.size(); k++)
        if (last_write[k] >= plan[k].ready_step) {
            snprintf(buf, sizeof buf, "host copy plan: chunk %lld is ready after %lld steps but step %lld writes it",
                     (long long)k, (long long)plan[k].ready_step, (long long)last_write[k]);
            return std::string(buf);
        }
    return "";
}

std::string check_schedule_bounds(const Symbolic &S, const Schedule &sc, const SolveSchedule *ss, int64_t dinv_extent)
{
    char buf[256];
    const int64_t xs = S.xsize, ds = dinv_extent >= 0 ? dinv_extent : sc.dinv_size, rs = sc.relmap_size;
    const int64_t ssz = S.Lpi ? S.Lpi[S.nsuper] : 0;
    auto bad = [&](const char *what, int64_t i) { snprintf(buf, sizeof buf, "%s %lld out of bounds", what, (long long)i); return std::string(buf); };
    // last entry a column-major block (rows x cols, ld) reaches, from off
    auto last = [](int64_t off, int64_t rows, int64_t cols, int64_t ld) { return off + (rows - 1) + (cols - 1) * ld; };
    int64_t nfused_refs = 0, nfused_steps = 0;
    for (size_t i = 0; i < sc.gemm.size(); i++) {
        const GemmTask &t = sc.gemm[i];
        if (t.M <= 0 || t.N <= 0 || t.K <= 0 || t.lda < t.M || t.ldc <= 0) return bad("gemm task shape", (int64_t)i);
        {
            // TileEntry packs the tile row and column into 16 bits each (more than 4.19 M rows of one task with
            // 64-row tiles): a wider task would wrap and compute some tiles twice, others never
            const int tr = tile_rows(t.big), tc = tile_cols(t.big);
            if (((int64_t)t.M + tr - 1) / tr > 65536 || ((int64_t)t.N + tc - 1) / tc > 65536) {
                snprintf(buf, sizeof buf, "too large: gemm task %lld has more than 65536 tile rows or columns", (long long)i);
                return std::string(buf);
            }
        }
        if (t.a_off < 0 || last(t.a_off, t.M, t.K, t.lda) >= xs) return bad("gemm task A", (int64_t)i);
        if (t.mode == GEMM_OVERWRITE) {
            if (t.b_off < 0 || last(t.b_off, t.N, t.K, t.ldb) >= ds) return bad("gemm task B (Dinv)", (int64_t)i);
        } else {
            if (t.N > t.M) return bad("gemm task N > M", (int64_t)i);
            if (t.b_off < 0 || last(t.b_off, t.N, t.K, t.ldb) >= xs) return bad("gemm task B", (int64_t)i);
        }
        if (t.fuse_potrf >= 0) {
            // the block this update completes: a POTRF task at the update's target origin, inside tile (0, 0) of a
            // big-tile, plain (unscattered) update of a real factor
            if (t.fuse_potrf >= (int64_t)sc.potrf.size() || t.mode != GEMM_DIRECT_SUB || t.big != 1 || S.cplx) return bad("gemm task fused potrf", (int64_t)i);
            const PotrfTask &pt = sc.potrf[(size_t)t.fuse_potrf];
            if (pt.a_off != t.c_off || pt.lda != t.ldc || pt.nb > t.N || pt.nb > t.M) return bad("gemm task fused potrf block", (int64_t)i);
            nfused_refs++;
        }
        if (t.mode == GEMM_SCATTER_SUB) {
            if (t.rm_off < 0 || t.rm_off + t.M > rs) return bad("gemm task relative map", (int64_t)i);
            if (t.c_off < 0 || t.c_off >= xs) return bad("gemm task C", (int64_t)i);
        } else {
            if (t.c_off < 0 || last(t.c_off, t.M, t.N, t.ldc) >= xs) return bad("gemm task C", (int64_t)i);
        }
    }
    for (size_t i = 0; i < sc.potrf.size(); i++) {
        const PotrfTask &t = sc.potrf[i];
        if (t.nb <= 0 || t.nb > NB || t.lda < t.nb) return bad("potrf task shape", (int64_t)i);
        if (t.a_off < 0 || last(t.a_off, t.nb, t.nb, t.lda) >= xs) return bad("potrf task A", (int64_t)i);
        if (t.dinv_off >= 0 && t.dinv_off + (int64_t)t.nb * t.nb > ds) return bad("potrf task Dinv", (int64_t)i);
        if (t.s < 0 || t.s >= S.nsuper || t.ls_off < 0 || t.ls_off + t.lda > ssz) return bad("potrf task rows", (int64_t)i);
        if (t.urow0 < 0 || t.urow0 > t.lda || (int64_t)t.k1 + t.col0 + t.nb > S.n) return bad("potrf task columns", (int64_t)i);
        if (t.fuse_trsm) {
            // k_potrf16 then also writes the rows below the block: the whole panel must lie inside Lx
            if (t.nb > 16 || t.col0 != 0 || S.cplx || sc.potrf_order.empty()) return bad("fused potrf task shape", (int64_t)i);
            if (last(t.a_off, t.lda, t.nb, t.lda) >= xs) return bad("fused potrf task panel", (int64_t)i);
        }
    }
    int64_t ngrouped_blocks = 0, ngrouped_steps = 0;
    for (size_t i = 0; i < sc.pgroups.size(); i++) {
        // a panel group is a window of one supernode: everything its kernels index follows from (s, col0, ncols)
        const PanelGroup &g = sc.pgroups[i];
        if (S.cplx || g.s < 0 || g.s >= S.nsuper) return bad("panel group supernode", (int64_t)i);
        const int64_t nscol = S.Super[g.s + 1] - S.Super[g.s], nsrow = S.Lpi[g.s + 1] - S.Lpi[g.s];
        if (g.ncols <= NB || g.ncols > SOLVE_GROUP * NB || g.col0 < 0 || g.col0 % (SOLVE_GROUP * NB) != 0 || (int64_t)g.col0 + g.ncols > nscol)
            return bad("panel group columns", (int64_t)i);
        if (nsrow >= (1 << 27)) return bad("panel group leading dimension", (int64_t)i);
        if (g.lda != nsrow || g.below != nsrow - g.col0 - g.ncols || g.a_off != S.Lpx[g.s] + g.col0 + (int64_t)g.col0 * nsrow)
            return bad("panel group panel", (int64_t)i);
        if (last(g.a_off, (int64_t)g.ncols + g.below, g.ncols, g.lda) >= xs) return bad("panel group extent", (int64_t)i);
        for (int b = 0; b * NB < g.ncols; b++) {
            const int64_t nb = std::min<int64_t>(NB, g.ncols - b * NB);
            if (g.dinv_off[b] < 0 || g.dinv_off[b] + nb * nb > ds) return bad("panel group Dinv", (int64_t)i);
            ngrouped_blocks++;
        }
    }
    for (size_t i = 0; i < sc.steps.size(); i++) {
        const Step &st = sc.steps[i];
        if (st.first < 0 || st.count < 0) return bad("step range", (int64_t)i);
        if (st.kind == STEP_PDIAG || st.kind == STEP_PTRSM) {
            if (st.first + st.count > (int64_t)sc.pgroups.size()) return bad("panel group step range", (int64_t)i);
            if (st.kind == STEP_PDIAG) continue;
            // the rows below the windows, `rows` per workgroup: the table lists every workgroup of every task, in order
            const int64_t rows = st.ngrouped;
            if (rows != 16 && rows != 32 && rows != 64 && rows != 128) return bad("panel group rows per workgroup of step", (int64_t)i);
            if (i == 0 || sc.steps[i - 1].kind != STEP_PDIAG || sc.steps[i - 1].first != st.first || sc.steps[i - 1].count != st.count)
                return bad("panel group solve without its diagonal step", (int64_t)i);
            if (st.tmap_off < 0 || st.tmap_off + st.ntiles > (int64_t)sc.pg_wg_task.size()) return bad("panel group step workgroups", (int64_t)i);
            int64_t at = 0;
            for (int64_t k = 0; k < st.count; k++) {
                const PanelGroup &g = sc.pgroups[(size_t)(st.first + k)];
                if (g.wg0 != at) return bad("panel group workgroup prefix of step", (int64_t)i);
                const int64_t nw = ((int64_t)g.below + rows - 1) / rows;
                for (int64_t b = 0; b < nw; b++)
                    if (at + b >= st.ntiles || sc.pg_wg_task[(size_t)(st.tmap_off + at + b)] != (int32_t)k) return bad("panel group workgroup table of step", (int64_t)i);
                at += nw;
            }
            if (at != st.ntiles) return bad("panel group workgroup count of step", (int64_t)i);
            continue;
        }
        if (st.kind == STEP_POTRF) {
            if (st.first + st.count > (int64_t)sc.potrf.size()) return bad("potrf step range", (int64_t)i);
            if (!sc.potrf_order.empty()) {
                // the order list of the step: a permutation of its tasks, the nsmall blocks of <= 16 columns first
                if (sc.potrf_order.size() != sc.potrf.size() || st.nsmall < 0 || st.nsmall > st.count - st.nfused) return bad("potrf order list", (int64_t)i);
                std::vector<uint8_t> seen((size_t)st.count, 0);
                for (int64_t k = 0; k < st.count; k++) {
                    const int32_t o = sc.potrf_order[(size_t)(st.first + k)];
                    if (o < 0 || o >= st.count || seen[(size_t)o]) return bad("potrf order entry of step", (int64_t)i);
                    seen[(size_t)o] = 1;
                    // the launched tasks: blocks of <= 16 columns first; the fused ones (the last nfused tasks of the step) last
                    if (k < st.count - st.nfused) {
                        if (o >= st.count - st.nfused) return bad("potrf order: fused task among the launched", (int64_t)i);
                        if ((sc.potrf[(size_t)(st.first + o)].nb <= 16) != (k < st.nsmall)) return bad("potrf order class of step", (int64_t)i);
                    } else if (o < st.count - st.nfused) return bad("potrf order: launched task among the fused", (int64_t)i);
                }
            } else if (st.nsmall != 0) return bad("potrf step without order list", (int64_t)i);
            if (st.nfused < 0 || st.nfused > st.count || st.ngrouped < 0 || st.ngrouped > st.nfused) return bad("potrf step fused count", (int64_t)i);
            nfused_steps += st.nfused - st.ngrouped;
            ngrouped_steps += st.ngrouped;
            if (st.tmap_off < 0 || st.tmap_off + st.stiles > (int64_t)sc.solve_tile_task.size()) return bad("potrf step solve tiles", (int64_t)i);
            for (int64_t b = 0; b < st.stiles; b++) {
                const int32_t ti = sc.solve_tile_task[(size_t)(st.tmap_off + b)];
                if (ti < 0 || ti >= st.count) return bad("solve tile table entry of step", (int64_t)i);
            }
        } else {
            if (st.first + st.count > (int64_t)sc.gemm.size()) return bad("gemm step range", (int64_t)i);
            if (st.flags & STEP_TINY) {
                // k_gemm_tiny takes one whole task per wave: one tile per task, update modes only, real factor
                if (st.ntiles != st.count || st.big || st.trsm || S.cplx) return bad("tiny step shape", (int64_t)i);
                for (int64_t k = 0; k < st.count; k++) {
                    const GemmTask &t = sc.gemm[(size_t)(st.first + k)];
                    if (t.big || t.mode == GEMM_OVERWRITE || t.M > TILE || t.N > TILE) return bad("task of tiny step", (int64_t)i);
                }
            }
            if (st.tmap_off < 0 || st.tmap_off + st.ntiles > (int64_t)sc.gemm_tile_tab.size()) return bad("gemm step tiles", (int64_t)i);
            for (int64_t b = 0; b < st.ntiles; b++) {      // every workgroup's (task, tile) inside its task
                const Schedule::TileEntry e = sc.gemm_tile_tab[(size_t)(st.tmap_off + b)];
                if (e.task < 0 || e.task >= st.count) return bad("tile table entry of step", (int64_t)i);
                const GemmTask &t = sc.gemm[(size_t)(st.first + e.task)];
                const int tr = tile_rows(t.big), tc = tile_cols(t.big);
                if ((int64_t)(e.tmtn & 0xffff) * tr >= t.M || (int64_t)(e.tmtn >> 16) * tc >= t.N)
                    return bad("tile of step", (int64_t)i);
            }
        }
    }
    if (nfused_refs != nfused_steps) return bad("fused potrf tasks without an update that runs them", nfused_steps - nfused_refs);
    if (ngrouped_blocks != ngrouped_steps) return bad("diagonal blocks without a launch and without a panel group", ngrouped_steps - ngrouped_blocks);
    for (size_t i = 0; i < sc.bcasts.size(); i++) {
        const BcastPoint &b = sc.bcasts[i];
        if (b.lo < 0 || b.count <= 0 || b.lo + b.count > xs || b.step < 0 || b.step > (int64_t)sc.steps.size() || b.root < 0)
            return bad("broadcast", (int64_t)i);
        if (b.wait_step < b.step || b.wait_step > (int64_t)sc.steps.size()) return bad("broadcast wait point", (int64_t)i);
        if (i > 0 && (b.step < sc.bcasts[i - 1].step || b.wait_step < sc.bcasts[i - 1].wait_step))
            return bad("broadcast order", (int64_t)i);
        // look-ahead: nothing launched between the issue and the wait point may touch the block in flight
        // (the rank that sends may go on reading the block; nobody may write it, and a receiver may not read it)
        auto hits = [&](int64_t off, int64_t rows, int64_t cols, int64_t ld) {
            return off < b.lo + b.count && last(off, rows, cols, ld) >= b.lo;
        };
        for (int64_t si = b.step; si < b.wait_step; si++) {
            const Step &st = sc.steps[(size_t)si];
            if (st.kind == STEP_POTRF) {
                // (the last nfused tasks of a step have no launch here: a panel group or a trailing update factorizes them)
                for (int64_t ti = st.first; ti < st.first + st.count - st.nfused; ti++) {
                    const PotrfTask &t = sc.potrf[(size_t)ti];
                    if (hits(t.a_off, t.nb, t.nb, t.lda)) return bad("broadcast in flight: a diagonal block being factorized", (int64_t)i);
                }
                continue;
            }
            if (st.kind != STEP_GEMM) {
                // a panel group reads and writes its own window and the rows below it: whole columns col0 .. col0 + ncols - 1
                for (int64_t ti = st.first; ti < st.first + st.count; ti++) {
                    const PanelGroup &g = sc.pgroups[(size_t)ti];
                    const int64_t lo_w = S.Lpx[g.s] + (int64_t)g.col0 * g.lda, hi_w = lo_w + (int64_t)g.ncols * g.lda;
                    if (lo_w < b.lo + b.count && hi_w > b.lo) return bad("broadcast in flight: the window of a panel group", (int64_t)i);
                }
                continue;
            }
            for (int64_t ti = st.first; ti < st.first + st.count; ti++) {
                const GemmTask &t = sc.gemm[(size_t)ti];
                if (!b.mine && hits(t.a_off, t.M, t.K, t.lda)) return bad("broadcast in flight: A operand of a task", (int64_t)i);
                if (!b.mine && t.mode != GEMM_OVERWRITE && hits(t.b_off, t.N, t.K, t.ldb)) return bad("broadcast in flight: B operand of a task", (int64_t)i);
                if (t.mode == GEMM_SCATTER_SUB) return bad("broadcast in flight over a scatter update", (int64_t)i);
                if (hits(t.c_off, t.M, t.N, t.ldc)) return bad("broadcast in flight: target of a task", (int64_t)i);
            }
        }
    }
    if (!ss) return "";
    for (size_t i = 0; i < ss->groups.size(); i++) {
        const SolveGroupTask &g = ss->groups[i];
        if (g.ncols <= 0 || g.ncols > SOLVE_GROUP * NB || g.lda < g.ncols) return bad("solve group shape", (int64_t)i);
        if (g.a_off < 0 || last(g.a_off, g.ncols, g.ncols, g.lda) >= xs) return bad("solve group L", (int64_t)i);
        if (g.x0 < 0 || (int64_t)g.x0 + g.ncols > S.n) return bad("solve group rows of X", (int64_t)i);
        for (int b = 0; b * NB < g.ncols; b++) {
            const int64_t nb = std::min<int64_t>(NB, g.ncols - b * NB);
            if (g.dinv_off[b] < 0 || g.dinv_off[b] + nb * nb > ds) return bad("solve group Dinv", (int64_t)i);
        }
    }
    for (size_t i = 0; i < ss->utasks.size(); i++) {
        const PotrfTask &t = ss->utasks[i];
        if (t.nb <= 0 || t.nb > NB || t.urow0 < t.col0 + t.nb || t.urow0 > t.lda) return bad("solve update task shape", (int64_t)i);
        if (t.a_off < 0 || last(t.a_off, (int64_t)t.lda - t.col0, t.nb, t.lda) >= xs) return bad("solve update task L", (int64_t)i);
        if (t.ls_off < 0 || t.ls_off + t.lda > ssz || (int64_t)t.k1 + t.col0 + t.nb > S.n) return bad("solve update task rows", (int64_t)i);
    }
    for (size_t i = 0; i < ss->steps.size(); i++) {
        const SolveStep &st = ss->steps[i];
        if (st.gfirst < 0 || st.gfirst + st.gcount > (int64_t)ss->groups.size()) return bad("solve step groups", (int64_t)i);
        if (st.ufirst < 0 || st.ufirst + st.ucount > (int64_t)ss->utasks.size()) return bad("solve step update tasks", (int64_t)i);
        if (st.tmap_off < 0 || st.tmap_off + st.stiles > (int64_t)ss->tile_task.size()) return bad("solve step tiles", (int64_t)i);
        if (st.maxblk < 1 || st.maxblk > SOLVE_GROUP) return bad("solve step maxblk", (int64_t)i);
        int64_t tiles = 0;
        for (int64_t u = 0; u < st.ucount; u++) {
            const PotrfTask &t = ss->utasks[(size_t)(st.ufirst + u)];
            if (t.stile0 != tiles) return bad("solve step tile prefix", (int64_t)i);
            tiles += ((int64_t)t.lda - t.urow0 + SOLVE_ROWS - 1) / SOLVE_ROWS;
        }
        if (tiles != st.stiles) return bad("solve step tile count", (int64_t)i);
        for (int64_t b = 0; b < st.stiles; b++) {
            const int32_t ti = ss->tile_task[(size_t)(st.tmap_off + b)];
            if (ti < 0 || ti >= st.ucount) return bad("solve step tile entry", (int64_t)i);
        }
        for (int which = 0; which < 2; which++) {
            const std::vector<SolveRange> &rg = which ? ss->ranges_mma : ss->ranges;
            const int64_t r0 = which ? st.rmfirst : st.rfirst, rn = which ? st.rmcount : st.rcount;
            if (r0 < 0 || r0 + rn > (int64_t)rg.size()) return bad("solve step runs", (int64_t)i);
            int64_t covered = 0;
            for (int64_t k = r0; k < r0 + rn; k++) {
                const SolveRange &e = rg[(size_t)k];
                if (e.task < 0 || e.task >= st.ucount || e.tile0 < 0 || e.ntiles <= 0) return bad("solve run", k);
                const PotrfTask &t = ss->utasks[(size_t)(st.ufirst + e.task)];
                const int64_t nt = ((int64_t)t.lda - t.urow0 + SOLVE_ROWS - 1) / SOLVE_ROWS;
                if (e.tile0 + e.ntiles > nt) return bad("solve run tiles", k);
                covered += e.ntiles;
            }
            if (covered != st.stiles) return bad("solve step run coverage", (int64_t)i);
        }
        if (ss->kmax <= 0) continue;
        // the wide path's lists: one-block groups on per-block tasks of their own, wider groups as SolveGroupUpd
        if (st.nufirst < 0 || st.nufirst + st.nucount > (int64_t)ss->nutasks.size()) return bad("solve step narrow tasks", (int64_t)i);
        if (st.ntmap_off < 0 || st.ntmap_off + st.nstiles > (int64_t)ss->ntile_task.size()) return bad("solve step narrow tiles", (int64_t)i);
        if (st.wfirst < 0 || st.wfirst + st.wcount > (int64_t)ss->wtasks.size()) return bad("solve step wide tasks", (int64_t)i);
        if (st.wmaxblk < 1 || st.wmaxblk * NB > std::max<int32_t>(ss->kmax, NB)) return bad("solve step wide depth", (int64_t)i);
        tiles = 0;
        for (int64_t u = 0; u < st.nucount; u++) {
            const PotrfTask &t = ss->nutasks[(size_t)(st.nufirst + u)];
            if (t.stile0 != tiles) return bad("solve step narrow tile prefix", (int64_t)i);
            const int64_t all = ((int64_t)t.lda - t.urow0 + SOLVE_ROWS - 1) / SOLVE_ROWS;
            tiles += st.wsplit > 0 ? std::min<int64_t>(all, st.wsplit) : all;       // (a split step lists its near tiles only)
        }
        if (tiles != st.nstiles) return bad("solve step narrow tile count", (int64_t)i);
        for (int64_t b = 0; b < st.nstiles; b++) {
            const int32_t ti = ss->ntile_task[(size_t)(st.ntmap_off + b)];
            if (ti < 0 || ti >= st.nucount) return bad("solve step narrow tile entry", (int64_t)i);
        }
        if (st.nrmfirst < 0 || st.nrmfirst + st.nrmcount > (int64_t)ss->nranges_mma.size()) return bad("solve step narrow runs", (int64_t)i);
        int64_t ncovered = 0;
        for (int64_t k = st.nrmfirst; k < st.nrmfirst + st.nrmcount; k++) {
            const SolveRange &e = ss->nranges_mma[(size_t)k];
            if (e.task < 0 || e.task >= st.nucount || e.tile0 < 0 || e.ntiles <= 0) return bad("solve narrow run", k);
            const PotrfTask &t = ss->nutasks[(size_t)(st.nufirst + e.task)];
            if (e.tile0 + e.ntiles > ((int64_t)t.lda - t.urow0 + SOLVE_ROWS - 1) / SOLVE_ROWS) return bad("solve narrow run tiles", k);
            ncovered += e.ntiles;
        }
        if (ncovered != st.nstiles) return bad("solve step narrow run coverage", (int64_t)i);
        if (st.wrfirst < 0 || st.wrfirst + st.wrcount > (int64_t)ss->wranges.size()) return bad("solve step wide runs", (int64_t)i);
        int64_t wcovered = 0, wtotal = 0;
        for (int64_t u = 0; u < st.wcount; u++) wtotal += ((int64_t)ss->wtasks[(size_t)(st.wfirst + u)].nrows + 63) / 64;
        for (int64_t k = st.wrfirst; k < st.wrfirst + st.wrcount; k++) {
            const SolveRange &e = ss->wranges[(size_t)k];
            if (e.task < 0 || e.task >= st.wcount || e.tile0 < 0 || e.ntiles <= 0 || e.ntiles > 64) return bad("solve wide run", k);   // 64 = WIDE_MAX_RUN (the kernels' row list in LDS)
            if (e.tile0 + e.ntiles > ((int64_t)ss->wtasks[(size_t)(st.wfirst + e.task)].nrows + 63) / 64) return bad("solve wide run tiles", k);
            wcovered += e.ntiles;
        }
        if (wcovered != wtotal) return bad("solve step wide run coverage", (int64_t)i);
        // every segment belongs to exactly one workgroup, in order; segments of one task are consecutive and in order
        if (st.wcount > 0) {
            if (st.wgfirst < 0 || st.wgcount <= 0 || st.wgfirst + st.wgcount + 1 > (int64_t)ss->wwg.size()) return bad("solve step workgroup table", (int64_t)i);
            if (ss->wwg[(size_t)st.wgfirst] != 0 || ss->wwg[(size_t)(st.wgfirst + st.wgcount)] != st.wrcount) return bad("solve step workgroup table ends", (int64_t)i);
            for (int64_t g = 0; g < st.wgcount; g++)
                if (ss->wwg[(size_t)(st.wgfirst + g)] >= ss->wwg[(size_t)(st.wgfirst + g + 1)]) return bad("solve step workgroup share", (int64_t)i);
            int64_t task = -1, next_tile = 0;
            for (int64_t k = st.wrfirst; k < st.wrfirst + st.wrcount; k++) {
                const SolveRange &e = ss->wranges[(size_t)k];
                if (e.task != task) { if (e.task != task + 1 && !(task < 0)) { /* tasks without rows do not exist */ } task = e.task; next_tile = 0; }
                if (e.tile0 != next_tile) return bad("solve wide segment order", k);
                next_tile += e.ntiles;
            }
            // a split step: one group, whose wide tasks start SOLVE_GROUP tiles below it and whose per-block tasks hold exactly
            // those tiles -- what the look-ahead's event order relies on (engine.hip, solve_dev)
            if (st.wsplit != 0) {
                const SolveGroupTask &gt = ss->groups[(size_t)st.gfirst];
                if (st.wsplit != SOLVE_GROUP || st.gcount != 1 || st.nucount <= 0 || st.nstiles != st.nucount * st.wsplit) return bad("solve step split", (int64_t)i);
                for (int64_t u = 0; u < st.wcount; u++) {
                    const SolveGroupUpd &wt = ss->wtasks[(size_t)(st.wfirst + u)];
                    const int64_t c0 = wt.x0 - gt.x0;
                    if (c0 < 0 || wt.a_off != gt.a_off + gt.ncols + (int64_t)st.wsplit * SOLVE_ROWS + c0 * gt.lda) return bad("solve step split: far rows", (int64_t)i);
                }
            }
        } else if (st.wrcount != 0 || st.wsplit != 0) return bad("solve step wide runs without tasks", (int64_t)i);
    }
    for (size_t i = 0; i < ss->nutasks.size(); i++) {
        const PotrfTask &t = ss->nutasks[i];
        if (t.nb <= 0 || t.nb > NB || t.urow0 < t.col0 + t.nb || t.urow0 > t.lda) return bad("solve narrow task shape", (int64_t)i);
        if (t.a_off < 0 || last(t.a_off, (int64_t)t.lda - t.col0, t.nb, t.lda) >= xs) return bad("solve narrow task L", (int64_t)i);
        if (t.ls_off < 0 || t.ls_off + t.lda > ssz || (int64_t)t.k1 + t.col0 + t.nb > S.n) return bad("solve narrow task rows", (int64_t)i);
    }
    for (size_t i = 0; i < ss->wtasks.size(); i++) {
        const SolveGroupUpd &t = ss->wtasks[i];
        if (t.ncols <= 0 || t.ncols > ss->kmax || t.nrows <= 0 || t.lda < t.nrows) return bad("solve wide task shape", (int64_t)i);
        if (t.a_off < 0 || last(t.a_off, t.nrows, t.ncols, t.lda) >= xs) return bad("solve wide task L", (int64_t)i);
        if (t.ls_off < 0 || t.ls_off + t.nrows > ssz) return bad("solve wide task rows", (int64_t)i);
        if (t.x0 < 0 || (int64_t)t.x0 + t.ncols > S.n) return bad("solve wide task rows of X", (int64_t)i);
    }
    return "";
}

}  // namespace mi355
