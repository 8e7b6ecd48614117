// tests/helpers/sched_emu.cpp -- TEST INFRASTRUCTURE.
// CPU emulator of the device schedule (suitesparse_amd/csrc/schedule.hpp): executes the
// very same task lists the HIP kernels consume, with plain loops, so that the HOST logic
// (levels, task offsets, relative maps, truncated not-posdef rerun, solve sweeps) can be
// checked against the oracle in the GPU-less container.  Never loaded by the product.
#include "hip/hip_runtime.h"
#include "../../suitesparse_amd/csrc/schedule.hpp"
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>
using namespace mi355;

// ---- the product's kernels, from their own source, on the CPU.  tests/helpers/hipcpu/hip/hip_runtime.h stands in for
// the HIP header (the build puts it first on the include path), so kernels.hip.hpp compiles unchanged: kernels
// without cross-lane operations run thread after thread (launch_serial), the others with one fiber per
// lane (launch).  Emulator key 100: 1 = assembly, k_relmap, k_gemm_tiny; 2 = also k_potrf16; 3 = every kernel of
// the real factorization (k_potrf64, k_trsm64, k_gemm64, k_gemm128) with the engine's launch geometry.
#include "../../suitesparse_amd/csrc/kernels.hip.hpp"
#include "../../suitesparse_amd/csrc/solve_kernels.hip.hpp"
#ifdef __clang__
#include "../../suitesparse_amd/csrc/kernels_complex.hip.hpp"      // uses clang's vector swizzles
#else
namespace mi355 {
#include "_host_kernels.inc"            // g++: only k_assemble_c, cut out of kernels_complex.hip.hpp by build()
}
#endif
namespace hostrun {
int use_device_source = 0;
}

namespace {
int find_row(const int64_t *a, int64_t len, int64_t key)
{
    const int64_t *p = std::lower_bound(a, a + len, key);
    return (p != a + len && *p == key) ? (int)(p - a) : -1;
}

struct Emu {
    Symbolic S;
    std::vector<double> Dinv;
    std::vector<int> info;
};

void run(const Symbolic &S, const Schedule &sc, int64_t ncols_asm, const int64_t *Ap, const int64_t *Ai,
         const double *Ax, double beta, double *Lx, std::vector<double> &Dinv, std::vector<int> &info,
         const int8_t *asm_mask = nullptr, bool init = true, int64_t step0 = 0, int64_t step1 = -1,
         bool keep_dinv = false, const int64_t *Fp = nullptr, const int64_t *Fi = nullptr, const double *Fx = nullptr)
{
    if (init) std::fill(Lx, Lx + S.xsize, 0.0);
    info.assign(S.nsuper + 2, 0);
    // what the engine checks before a schedule reaches the device must hold for every schedule run here
    if (!check_schedule_bounds(S, sc).empty()) { info[S.nsuper + 1] = -1; return; }
    if (!keep_dinv) Dinv.assign(sc.dinv_size + 1, 0.0);
    if (step1 < 0) step1 = (int64_t)sc.steps.size();
    const bool dev = hostrun::use_device_source != 0;
    std::vector<int32_t> ls32;
    if (dev) { const int64_t nls = S.Lpi[S.nsuper]; ls32.resize((size_t)nls + 1); for (int64_t i = 0; i < nls; i++) ls32[i] = (int32_t)S.Ls[i]; }
    if (dev && init && ncols_asm > 0) {
        // the engine's assembly launches (run_schedule), same arguments
        const int64_t grid = (ncols_asm + 255) / 256;
        if (Fp) {
            hipcpu::launch_serial(grid, 256, [&] { k_assemble_aat<1>(ncols_asm, Ap, Ai, nullptr, Ax, Fp, Fi, nullptr, Fx, sc.supermap.data(),
                                                                 S.Super, S.Lpi, S.Lpx, ls32.data(), Lx, asm_mask); });
            if (beta != 0.0)
                hipcpu::launch_serial(grid, 256, [&] { k_add_beta<1>(ncols_asm, beta, sc.supermap.data(), S.Super, S.Lpi, S.Lpx, Lx, asm_mask); });
        } else {
            hipcpu::launch_serial(grid, 256, [&] { k_assemble(ncols_asm, Ap, Ai, nullptr, Ax, sc.supermap.data(), S.Super, S.Lpi, S.Lpx,
                                                        ls32.data(), Lx, beta, asm_mask); });
        }
    }
    for (int64_t k = 0; !dev && init && k < ncols_asm; k++) {
        int s = sc.supermap[k];
        if (asm_mask && !asm_mask[s]) continue;
        int64_t k1 = S.Super[s], psi = S.Lpi[s], psx = S.Lpx[s], nsrow = S.Lpi[s + 1] - psi;
        int64_t base = psx + (k - k1) * nsrow;
        // stype == 0 (mi355_factor_set_matrix_unsym): column k of A*F
        for (int64_t pf = Fp ? Fp[k] : 0; Fp && pf < Fp[k + 1]; pf++) {
            const int64_t j = Fi[pf];
            for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                int64_t i = Ai[p];
                if (i >= k) { int im = find_row(S.Ls + psi, nsrow, i); if (im >= 0) Lx[base + im] += Ax[p] * Fx[pf]; }
            }
        }
        for (int64_t p = Ap[k]; !Fp && p < Ap[k + 1]; p++) {
            int64_t i = Ai[p];
            if (i >= k) { int im = find_row(S.Ls + psi, nsrow, i); if (im >= 0) Lx[base + im] = Ax[p]; }
        }
        if (beta != 0) Lx[base + (k - k1)] += beta;
    }
    // relative maps
    std::vector<int32_t> relmap(sc.relmap_size + 1);
    if (dev && sc.relmap_size > 0) {
        // upload_schedule's k_relmap launch
        std::vector<RelmapPair> rp(sc.pairs.size());
        for (size_t i = 0; i < rp.size(); i++) {
            rp[i].ls_src = sc.pairs[i].ls_src; rp[i].ls_dst = sc.pairs[i].ls_dst;
            rp[i].rm_off = sc.pairs[i].rm_off; rp[i].nsrow = sc.pairs[i].nsrow; rp[i].pad = 0;
        }
        int err = 0;
        hipcpu::launch_serial((sc.relmap_size + 255) / 256, 256, [&] { k_relmap(sc.relmap_size, (int64_t)rp.size(), rp.data(), ls32.data(), relmap.data(), &err); });
        if (err) { info[S.nsuper + 1] = -1; return; }
    } else
    for (const UpdatePair &u : sc.pairs)
        for (int i = 0; i < u.ndrow2; i++)
            relmap[u.rm_off + i] = find_row(S.Ls + u.ls_dst, u.nsrow, S.Ls[u.ls_src + i]);
    auto host_potrf = [&](int64_t ti) {
                const PotrfTask &t = sc.potrf[ti];
                double *A = Lx + t.a_off; int nb = t.nb; int64_t lda = t.lda;
                bool failed = false;
                for (int j = 0; j < nb; j++) {
                    double d = A[j + j * lda];
                    if (!(d > 0.0) && !failed) { failed = true; if (info[t.s] == 0) info[t.s] = t.col0 + j + 1; }
                    double ljj = std::sqrt(d);
                    A[j + j * lda] = ljj;
                    for (int i = j + 1; i < nb; i++) A[i + j * lda] /= ljj;
                    for (int k = j + 1; k < nb; k++)
                        for (int i = k; i < nb; i++) A[i + k * lda] -= A[i + j * lda] * A[k + j * lda];
                }
                double *D = Dinv.data() + t.dinv_off;
                for (int c = 0; c < nb; c++) {
                    std::vector<double> x(nb, 0.0); x[c] = 1.0;
                    for (int k = 0; k < nb; k++) {
                        x[k] /= A[k + k * lda];
                        for (int i = k + 1; i < nb; i++) x[i] -= A[i + k * lda] * x[k];
                    }
                    for (int i = 0; i < nb; i++) D[i + c * nb] = (i >= c) ? x[i] : 0.0;
                }
                if (t.fuse_trsm) {
                    // the step's TRSM launch has no task for this block: X = B * inv(L11)^T for the rows below
                    for (int64_t r = nb; r < lda; r++) {
                        std::vector<double> xr(nb, 0.0);
                        for (int c = 0; c < nb; c++)
                            for (int k = 0; k <= c; k++) xr[c] += A[r + k * lda] * D[c + k * nb];
                        for (int c = 0; c < nb; c++) A[r + c * lda] = xr[c];
                    }
                }
    };
    std::vector<double> C;
    for (int64_t si = step0; si < step1; si++) {
        const Step &st = sc.steps[(size_t)si];
        if (st.kind == STEP_POTRF) {
            std::vector<int64_t> todo;
            if (hostrun::use_device_source >= 3) {
                // exactly the engine's POTRF launches (run_schedule)
                if (!sc.potrf_order.empty()) {
                    const int32_t *ord = sc.potrf_order.data() + st.first;
                    if (st.nsmall > 0)
                        hipcpu::launch((st.nsmall + POTRF16_WAVES - 1) / POTRF16_WAVES, 64 * POTRF16_WAVES, [&] {
                            k_potrf16(sc.potrf.data() + st.first, ord, (int64_t)st.nsmall, Lx, Dinv.data(), info.data()); });
                    if (st.count - st.nfused > st.nsmall)
                        hipcpu::launch(st.count - st.nfused - st.nsmall, 256, [&] {
                            k_potrf64(sc.potrf.data() + st.first, Lx, Dinv.data(), info.data(), ord + st.nsmall); });
                } else if (st.count > st.nfused)
                    hipcpu::launch(st.count - st.nfused, 256, [&] { k_potrf64(sc.potrf.data() + st.first, Lx, Dinv.data(), info.data(), nullptr); });
            } else if (hostrun::use_device_source >= 2 && !sc.potrf_order.empty()) {
                // the engine's launches: k_potrf16 (from its own source) over the first nsmall entries of the order
                // list, the wider blocks by the loops below
                const int32_t *ord = sc.potrf_order.data() + st.first;
                if (st.nsmall > 0)
                    hipcpu::launch((st.nsmall + POTRF16_WAVES - 1) / POTRF16_WAVES, 64 * POTRF16_WAVES, [&] {
                        k_potrf16(sc.potrf.data() + st.first, ord, (int64_t)st.nsmall, Lx, Dinv.data(), info.data()); });
                for (int64_t k = st.nsmall; k < st.count - st.nfused; k++) todo.push_back(st.first + ord[k]);
            } else
                for (int64_t ti = st.first; ti < st.first + st.count - st.nfused; ti++) todo.push_back(ti);   // (the fused ones are last)
            for (int64_t ti : todo) host_potrf(ti);
        } else if (st.kind == STEP_PDIAG) {
            if (hostrun::use_device_source >= 3) {
                hipcpu::launch(st.count, 256, [&] { k_panel_diag(sc.pgroups.data() + st.first, Lx, Dinv.data(), info.data()); });
            } else
            for (int64_t gi = st.first; gi < st.first + st.count; gi++) {
                // plain loops: Cholesky of the ncols x ncols window, then the inverse of each 64-column diagonal block
                const PanelGroup &g = sc.pgroups[(size_t)gi];
                double *A = Lx + g.a_off; const int nc = g.ncols; const int64_t lda = g.lda;
                bool failed = false;
                for (int j = 0; j < nc; j++) {
                    double d = A[j + j * lda];
                    if (!(d > 0.0) && !failed) { failed = true; if (info[g.s] == 0) info[g.s] = g.col0 + j + 1; }
                    double ljj = std::sqrt(d);
                    A[j + j * lda] = ljj;
                    for (int i = j + 1; i < nc; i++) A[i + j * lda] /= ljj;
                    for (int k = j + 1; k < nc; k++)
                        for (int i = k; i < nc; i++) A[i + k * lda] -= A[i + j * lda] * A[k + j * lda];
                }
                for (int b = 0; b * NB < nc; b++) {
                    const int nb = std::min(NB, nc - b * NB);
                    const double *Ab = A + b * NB + (int64_t)b * NB * lda;
                    double *D = Dinv.data() + g.dinv_off[b];
                    for (int c = 0; c < nb; c++) {
                        std::vector<double> x(nb, 0.0); x[c] = 1.0;
                        for (int k = 0; k < nb; k++) {
                            x[k] /= Ab[k + k * lda];
                            for (int i = k + 1; i < nb; i++) x[i] -= Ab[i + k * lda] * x[k];
                        }
                        for (int i = 0; i < nb; i++) D[i + c * nb] = (i >= c) ? x[i] : 0.0;
                    }
                }
            }
        } else if (st.kind == STEP_PTRSM) {
            // the workgroup table: every workgroup of every task, in order
            {
                int64_t at = 0;
                for (int64_t k = 0; k < st.count; k++) {
                    const PanelGroup &g = sc.pgroups[(size_t)(st.first + k)];
                    if (g.wg0 != at) info[S.nsuper + 1] = -1;
                    const int64_t nw = ((int64_t)g.below + st.ngrouped - 1) / st.ngrouped;
                    for (int64_t b = 0; b < nw; b++)
                        if (at + b >= st.ntiles || sc.pg_wg_task[(size_t)(st.tmap_off + at + b)] != (int32_t)k) info[S.nsuper + 1] = -1;
                    at += nw;
                }
                if (at != st.ntiles) info[S.nsuper + 1] = -1;
            }
            if (hostrun::use_device_source >= 3) {
                const PanelGroup *pg = sc.pgroups.data() + st.first;
                const int32_t *wt = sc.pg_wg_task.data() + st.tmap_off;
                if (st.ngrouped == 128) hipcpu::launch(st.ntiles, 512, [&] { k_panel_trsm_reg<8>(pg, wt, Dinv.data(), Lx); });
                else if (st.ngrouped == 64) hipcpu::launch(st.ntiles, 256, [&] { k_panel_trsm_reg<4>(pg, wt, Dinv.data(), Lx); });
                else if (st.ngrouped == 32) hipcpu::launch(st.ntiles, 256, [&] { k_panel_trsm<32>(pg, wt, Dinv.data(), Lx); });
                else hipcpu::launch(st.ntiles, 256, [&] { k_panel_trsm<16>(pg, wt, Dinv.data(), Lx); });
            } else
            for (int64_t gi = st.first; gi < st.first + st.count; gi++) {
                // plain loops: X * L_w^T = A for the rows below the window, column by column
                const PanelGroup &g = sc.pgroups[(size_t)gi];
                double *A = Lx + g.a_off; const int nc = g.ncols; const int64_t lda = g.lda;
                for (int j = 0; j < nc; j++) {
                    const double ljj = A[j + j * lda];
                    for (int64_t r = nc; r < nc + g.below; r++) {
                        double v = A[r + j * lda];
                        for (int k = 0; k < j; k++) v -= A[r + k * lda] * A[j + k * lda];
                        A[r + j * lda] = v / ljj;
                    }
                }
            }
        } else if (hostrun::use_device_source >= 3 && !(st.flags & STEP_TINY)) {
            // exactly the engine's GEMM launches of a real factor
            const GemmTask *tk = sc.gemm.data() + st.first;
            const Schedule::TileEntry *tm = sc.gemm_tile_tab.data() + st.tmap_off;
            if (st.big)
                hipcpu::launch(st.ntiles, 256, [&] { k_gemm128<false>(tk, tm, Lx, Dinv.data(), relmap.data(), (unsigned long long *)nullptr,
                                                                       sc.potrf.data(), Dinv.data(), info.data()); });
            else if (st.trsm)
                hipcpu::launch(st.ntiles, 256, [&] { k_trsm64(tk, tm, Lx, Dinv.data()); });
            else
                hipcpu::launch(st.ntiles, 256, [&] { k_gemm64(tk, tm, Lx, Dinv.data(), relmap.data()); });
            if (!check_tile_table(sc, st)) info[S.nsuper + 1] = -1;
        } else if (dev && (st.flags & STEP_TINY)) {
            // the engine's k_gemm_tiny launch of this step
            hipcpu::launch_serial((st.ntiles + TINY_WAVES - 1) / TINY_WAVES, 64 * TINY_WAVES, [&] {
                k_gemm_tiny(sc.gemm.data() + st.first, sc.gemm_tile_tab.data() + st.tmap_off, st.ntiles, Lx, relmap.data()); });
            if (!check_tile_table(sc, st)) info[S.nsuper + 1] = -1;
        } else {
            int64_t tiles = 0;
            for (int64_t ti = st.first; ti < st.first + st.count; ti++) {
                const GemmTask &t = sc.gemm[ti];
                if (t.tile0 != tiles) { info[S.nsuper + 1] = -1; }       // prefix sums must be exact
                tiles += gemm_tiles(t.M, t.N, t.mode, tile_rows(t.big), tile_cols(t.big));
                if (t.big != st.big) info[S.nsuper + 1] = -1;
                const double *A = Lx + t.a_off;
                const double *B = (t.mode == GEMM_OVERWRITE ? Dinv.data() : Lx) + t.b_off;
                C.assign((size_t)t.M * t.N, 0.0);
                for (int k = 0; k < t.K; k++)
                    for (int n = 0; n < t.N; n++) {
                        double b = B[n + (int64_t)k * t.ldb];
                        for (int m = 0; m < t.M; m++) C[m + (size_t)n * t.M] += A[m + (int64_t)k * t.lda] * b;
                    }
                double *T = Lx + t.c_off;
                for (int n = 0; n < t.N; n++)
                    for (int m = 0; m < t.M; m++) {
                        double v = C[m + (size_t)n * t.M];
                        if (t.mode == GEMM_SCATTER_SUB) {
                            if (m >= n) T[relmap[t.rm_off + m] + (int64_t)relmap[t.rm_off + n] * t.ldc] -= v;
                        } else if (t.mode == GEMM_DIRECT_SUB) {
                            if (m >= n) T[m + (int64_t)n * t.ldc] -= v;
                        } else T[m + (int64_t)n * t.ldc] = v;
                    }
            }
            if (tiles != st.ntiles || !check_tile_table(sc, st)) info[S.nsuper + 1] = -1;
            // the diagonal blocks that these trailing updates complete (GemmTask::fuse_potrf): k_gemm128 factorizes them
            // in its tile (0, 0); the host loops do it here, after the step
            for (int64_t ti = st.first; ti < st.first + st.count; ti++)
                if (sc.gemm[(size_t)ti].fuse_potrf >= 0) host_potrf(sc.gemm[(size_t)ti].fuse_potrf);
        }
    }
}
}  // namespace

extern "C" {

// schedule tuning of the emulator's own copy of schedule.cpp (2: outer block width, 7: second outer block width)
void emu_set_tuning(int key, int value)
{
    if (key == 0) tuning().big_min_m = value;
    else if (key == 1) tuning().big_min_n = value;
    else if (key == 2) tuning().ob = value;
    else if (key == 4) tuning().supertile = value;
    else if (key == 5) tuning().xcd_deal = value;
    else if (key == 7) tuning().ob2 = value;
    else if (key == 8) tuning().ob3 = value;
    else if (key == 11) tuning().solve_bwd_div = value;
    else if (key == 13) tuning().solve_bwd_div_mma = value;
    else if (key == 14) tuning().tiny_max = value;
    else if (key == 26) tuning().fuse_potrf = value;
    else if (key == 27) tuning().deterministic = value;
    else if (key == 28) tuning().panel_group = value;
    else if (key == 30) tuning().solve_lookahead = value;
    else if (key == 33) tuning().solve_wide_ksplit = value;
    else if (key == 100) hostrun::use_device_source = value;    // run the product's host-runnable kernels
}

// tile tables of the schedule built for a complex factor (rectangular big tiles): 0 if every GEMM step's table
// lists exactly the tiles that hold an entry on or below the diagonal, once each; else the number of bad steps
int emu_check_tile_tables(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                          const int64_t *Ls, int64_t ssize, int64_t xsize, int cplx, int64_t *nbig_tiles)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls; S.cplx = cplx != 0;
    if (!validate_symbolic(S).empty()) return -4;
    Schedule sc; build_schedule(S, nsuper, 0, sc);
    int bad = 0;
    int64_t nb = 0;
    for (const Step &st : sc.steps) {
        if (st.kind != STEP_GEMM) continue;
        if (!check_tile_table(sc, st)) bad++;
        if (st.big) nb += st.ntiles;
        for (int64_t i = 0; i < st.count; i++) {
            const GemmTask &t = sc.gemm[(size_t)(st.first + i)];
            if (t.big != (st.big ? (cplx ? 2 : 1) : 0)) bad++;
        }
    }
    if (nbig_tiles) *nbig_tiles = nb;
    return bad;
}

// The product's assembly kernels alone, from their own source (E = 1 real, 2 complex interleaved):
//   kind 0: k_assemble / k_assemble_c (S = tril(PAP'));  1: k_assemble_perm<E> + k_add_beta<E> (A + inverse permutation);
//   2: k_assemble_aat<E> + k_add_beta<E> (stype == 0: A n x ncol in Ap/Ai/Ax, F in Fp/Fi/Fx).  Lx (E*xsize) is zeroed first.
int emu_device_assemble(int kind, int E, int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi,
                        const int64_t *Lpx, const int64_t *Ls, int64_t xsize, const int64_t *Ap, const int64_t *Ai,
                        const int64_t *Anz, const double *Ax, const int64_t *Fp, const int64_t *Fi, const int64_t *Fnz,
                        const double *Fx, const int64_t *iperm, double beta, double *Lx)
{
    std::vector<int32_t> supermap((size_t)n + 1), ls32((size_t)Lpi[nsuper] + 1);
    for (int64_t s = 0; s < nsuper; s++) for (int64_t k = Super[s]; k < Super[s + 1]; k++) supermap[k] = (int32_t)s;
    for (int64_t i = 0; i < Lpi[nsuper]; i++) ls32[i] = (int32_t)Ls[i];
    std::fill(Lx, Lx + (size_t)xsize * E, 0.0);
    const int64_t grid = (n + 255) / 256;
    const int8_t *nomask = nullptr;
    if (kind == 0 && E == 1)
        hipcpu::launch_serial(grid, 256, [&] { k_assemble(n, Ap, Ai, Anz, Ax, supermap.data(), Super, Lpi, Lpx, ls32.data(), Lx, beta, nomask); });
    else if (kind == 0)
        hipcpu::launch_serial(grid, 256, [&] { k_assemble_c(n, Ap, Ai, Anz, Ax, supermap.data(), Super, Lpi, Lpx, ls32.data(), Lx, beta, nomask); });
    else if (kind == 1 && E == 1)
        hipcpu::launch_serial(grid, 256, [&] { k_assemble_perm<1>(n, n, Ap, Ai, Anz, Ax, iperm, supermap.data(), Super, Lpi, Lpx, ls32.data(), Lx, nomask); });
    else if (kind == 1)
        hipcpu::launch_serial(grid, 256, [&] { k_assemble_perm<2>(n, n, Ap, Ai, Anz, Ax, iperm, supermap.data(), Super, Lpi, Lpx, ls32.data(), Lx, nomask); });
    else if (kind == 2 && E == 1)
        hipcpu::launch_serial(grid, 256, [&] { k_assemble_aat<1>(n, Ap, Ai, Anz, Ax, Fp, Fi, Fnz, Fx, supermap.data(), Super, Lpi, Lpx, ls32.data(), Lx, nomask); });
    else if (kind == 2)
        hipcpu::launch_serial(grid, 256, [&] { k_assemble_aat<2>(n, Ap, Ai, Anz, Ax, Fp, Fi, Fnz, Fx, supermap.data(), Super, Lpi, Lpx, ls32.data(), Lx, nomask); });
    else return -1;
    if (kind != 0 && beta != 0.0) {
        if (E == 1) hipcpu::launch_serial(grid, 256, [&] { k_add_beta<1>(n, beta, supermap.data(), Super, Lpi, Lpx, Lx, nomask); });
        else hipcpu::launch_serial(grid, 256, [&] { k_add_beta<2>(n, beta, supermap.data(), Super, Lpi, Lpx, Lx, nomask); });
    }
    return 0;
}

// launch classes of the full schedule under the current tuning: out[0] tasks in STEP_TINY launches, out[1] those
// launches, out[2] diagonal blocks of <= 16 columns listed first in potrf_order, out[3] all diagonal blocks,
// out[4] 1 if check_schedule_bounds accepts the schedule
int emu_schedule_classes(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                         const int64_t *Ls, int64_t ssize, int64_t xsize, int64_t *out)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    if (!validate_symbolic(S).empty()) return -4;
    Schedule sc; build_schedule(S, nsuper, 0, sc);
    out[0] = out[1] = out[2] = 0; out[3] = (int64_t)sc.potrf.size();
    for (const Step &st : sc.steps) {
        if (st.kind == STEP_GEMM && (st.flags & STEP_TINY)) { out[0] += st.count; out[1]++; }
        if (st.kind == STEP_POTRF) out[2] += st.nsmall;
    }
    out[4] = check_schedule_bounds(S, sc).empty() ? 1 : 0;
    return 0;
}

// launches of the full schedule under the current tuning, by kind: out[0] POTRF steps that launch anything, out[1] GEMM
// steps, out[2] STEP_PDIAG, out[3] STEP_PTRSM, out[4] 1 if check_schedule_bounds accepts the schedule
int emu_schedule_counts(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                        const int64_t *Ls, int64_t ssize, int64_t xsize, int64_t *out)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    if (!validate_symbolic(S).empty()) return -4;
    Schedule sc; build_schedule(S, nsuper, 0, sc);
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (const Step &st : sc.steps) {
        if (st.kind == STEP_POTRF) out[0] += st.count > st.nfused ? 1 : 0;
        else if (st.kind == STEP_GEMM) out[1]++;
        else if (st.kind == STEP_PDIAG) out[2]++;
        else if (st.kind == STEP_PTRSM) out[3]++;
    }
    out[4] = check_schedule_bounds(S, sc).empty() ? 1 : 0;
    return 0;
}

// check_schedule_bounds on ONE hand-made trailing-update task of M x N (big: tile class) inside a factor large enough
// to hold it: 1 if it is rejected as "too large" (more tile rows / columns than the 16-bit fields of a TileEntry
// hold), 0 if accepted, -1 if rejected for another reason
int emu_check_wide_task(int64_t M, int64_t N, int big)
{
    Symbolic S;
    int64_t lpi[2] = {0, 0};
    S.n = M; S.nsuper = 1; S.Lpi = lpi;
    S.xsize = M * (N + 1) + M;
    Schedule sc;
    GemmTask t;
    t.a_off = 0; t.b_off = 0; t.c_off = M; t.rm_off = -1;
    t.lda = (int32_t)M; t.ldb = (int32_t)M; t.ldc = (int32_t)M;
    t.M = (int32_t)M; t.N = (int32_t)N; t.K = 1; t.mode = GEMM_DIRECT_SUB; t.big = big; t.tile0 = 0; t.fuse_potrf = -1; t.pad = 0;
    sc.gemm.push_back(t);
    const std::string e = check_schedule_bounds(S, sc);
    if (e.empty()) return 0;
    return e.compare(0, 10, "too large:") == 0 ? 1 : -1;
}

// returns 0, or <0 on schedule inconsistency; status/minor as the engine reports them
int emu_super_numeric_f(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                        const int64_t *Ls, int64_t ssize, int64_t xsize, const int64_t *Ap, const int64_t *Ai,
                        const double *Ax, const int64_t *Fp, const int64_t *Fi, const double *Fx,
                        double beta, int quick, double *Lx, double *Dinv_out,
                        int64_t *minor, int *status, int64_t *stats)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    if (!validate_symbolic(S).empty()) return -4;
    Schedule sc; build_schedule(S, nsuper, 0, sc);
    std::vector<double> Dinv; std::vector<int> info;
    run(S, sc, n, Ap, Ai, Ax, beta, Lx, Dinv, info, nullptr, true, 0, -1, false, Fp, Fi, Fx);
    if (info[nsuper + 1] != 0) return -10;
    if (stats) { stats[0] = sc.nlevels; stats[1] = (int64_t)sc.steps.size(); stats[2] = (int64_t)sc.pairs.size();
                 stats[3] = sc.relmap_size; stats[4] = sc.dinv_size; stats[5] = (int64_t)sc.gemm.size();
                 stats[6] = 0; stats[7] = 0;      // tasks in STEP_TINY launches, and those launches
                 for (const Step &st : sc.steps) if (st.kind == STEP_GEMM && (st.flags & STEP_TINY)) { stats[6] += st.count; stats[7]++; } }
    *status = 0; *minor = n;
    int64_t sf = -1;
    for (int64_t s = 0; s < nsuper; s++) if (info[s]) { sf = s; break; }
    if (sf >= 0) {
        int inf = info[sf];
        *status = 1; *minor = Super[sf] + inf - 1;
        int64_t nscol2 = (inf == 1 || quick) ? 0 : inf - 1;
        Schedule tr; build_schedule(S, sf, nscol2, tr);
        std::vector<int> info2;
        run(S, tr, nscol2 > 0 ? Super[sf + 1] : Super[sf], Ap, Ai, Ax, beta, Lx, Dinv, info2, nullptr, true, 0, -1, false, Fp, Fi, Fx);
        if (info2[nsuper + 1] != 0) return -11;
        int64_t nsrow = Lpi[sf + 1] - Lpi[sf];
        int64_t tail = Lpx[sf] + nsrow * nscol2;
        std::fill(Lx + tail, Lx + xsize, 0.0);
    } else if (Dinv_out) {
        std::copy(Dinv.begin(), Dinv.begin() + sc.dinv_size, Dinv_out);
    }
    return 0;
}

int emu_super_numeric(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                      const int64_t *Ls, int64_t ssize, int64_t xsize, const int64_t *Ap, const int64_t *Ai,
                      const double *Ax, double beta, int quick, double *Lx, double *Dinv_out,
                      int64_t *minor, int *status, int64_t *stats)
{
    return emu_super_numeric_f(n, nsuper, Super, Lpi, Lpx, Ls, ssize, xsize, Ap, Ai, Ax, nullptr, nullptr, nullptr,
                               beta, quick, Lx, Dinv_out, minor, status, stats);
}

// one multi-GPU phase (mi355_factor_numeric_phase): masked schedule, optional init/assembly
int emu_numeric_phase(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                      const int64_t *Ls, int64_t ssize, int64_t xsize, const int64_t *Ap, const int64_t *Ai,
                      const double *Ax, double beta, const int8_t *proc_mask, const int8_t *asm_mask, int init,
                      double *Lx, int64_t *first_bad, int64_t *info_out,
                      int64_t dinv_base, double *Dinv_out, int64_t *dinv_end)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    Schedule sc; build_schedule(S, nsuper, 0, sc, proc_mask, dinv_base);
    std::vector<double> Dinv; std::vector<int> info;
    run(S, sc, n, Ap, Ai, Ax, beta, Lx, Dinv, info, asm_mask, init != 0);
    // the block inverses of this phase, kept by the caller for the solve phases (slots dinv_base ...)
    if (Dinv_out) std::copy(Dinv.begin() + dinv_base, Dinv.begin() + sc.dinv_size, Dinv_out + dinv_base);
    if (dinv_end) *dinv_end = sc.dinv_size;
    if (info[nsuper + 1] != 0) return -10;
    *first_bad = -1; *info_out = 0;
    for (int64_t s = 0; s < nsuper; s++) if (info[s]) { *first_bad = s; *info_out = info[s]; break; }
    return 0;
}

// owner-computes top (mi355_factor_top_plan / _top_run): the broadcast points of rank `rank`'s schedule
// (4 int64 each: step, lo, count, root) and the number of steps
int emu_top_plan(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                 const int64_t *Ls, int64_t ssize, int64_t xsize, const int8_t *top_mask, int rank, int world,
                 int64_t dinv_base, int64_t cap, int64_t *plan, int64_t *count, int64_t *nsteps, int64_t *dinv_end)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    DistTop dt; dt.rank = rank; dt.world = world;
    Schedule sc; build_schedule(S, nsuper, 0, sc, top_mask, dinv_base, &dt);
    *count = (int64_t)sc.bcasts.size();
    *nsteps = (int64_t)sc.steps.size();
    *dinv_end = sc.dinv_size;
    if (*count > cap) return -20;
    for (size_t i = 0; i < sc.bcasts.size(); i++) {
        plan[4 * i] = sc.bcasts[i].step; plan[4 * i + 1] = sc.bcasts[i].lo;
        plan[4 * i + 2] = sc.bcasts[i].count; plan[4 * i + 3] = sc.bcasts[i].root;
    }
    for (const Step &st : sc.steps) if (!check_tile_table(sc, st)) return -21;
    return 0;
}

// steps [step0, step1) of that schedule on (Lx, Dinv); Dinv holds dinv_end + 1 entries and is kept
int emu_top_run(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                const int64_t *Ls, int64_t ssize, int64_t xsize, const int8_t *top_mask, int rank, int world,
                int64_t dinv_base, int64_t step0, int64_t step1, double *Lx, double *Dinv_io,
                int64_t *first_bad, int64_t *info_out)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    DistTop dt; dt.rank = rank; dt.world = world;
    Schedule sc; build_schedule(S, nsuper, 0, sc, top_mask, dinv_base, &dt);
    std::vector<double> Dinv(Dinv_io, Dinv_io + sc.dinv_size + 1);
    std::vector<int> info;
    run(S, sc, 0, nullptr, nullptr, nullptr, 0.0, Lx, Dinv, info, nullptr, false, step0, step1, true);
    std::copy(Dinv.begin(), Dinv.begin() + sc.dinv_size, Dinv_io);
    if (info[nsuper + 1] != 0) return -10;
    *first_bad = -1; *info_out = 0;
    for (int64_t s = 0; s < nsuper; s++) if (info[s]) { *first_bad = s; *info_out = info[s]; break; }
    return 0;
}

int64_t emu_dinv_size_masked(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                             const int64_t *Ls, int64_t ssize, int64_t xsize, const int8_t *mask)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    Schedule sc; build_schedule(S, nsuper, 0, sc, mask);
    return sc.dinv_size;
}

int64_t emu_dinv_size(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                      const int64_t *Ls, int64_t ssize, int64_t xsize)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    Schedule sc; build_schedule(S, nsuper, 0, sc);
    return sc.dinv_size;
}

// the solve sweeps exactly as engine.hip's solve_dev walks the POTRF steps
// (proc_mask, dinv_base): the sweep over one masked schedule of a partitioned factor
// (mi355_super_solve_phase_dev); nullptr, 0: the whole factor
int emu_solve_phase(int lower, int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi,
                    const int64_t *Lpx, const int64_t *Ls, int64_t ssize, int64_t xsize, const double *Lx,
                    const double *Dinv, int64_t nrhs, double *X, int64_t d, const int8_t *proc_mask, int64_t dinv_base)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    Schedule sc; build_schedule(S, nsuper, 0, sc, proc_mask, dinv_base);
    std::vector<const Step *> ps;
    for (const Step &st : sc.steps) if (st.kind == STEP_POTRF) ps.push_back(&st);
    if (!lower) std::reverse(ps.begin(), ps.end());
    for (const Step *st : ps) {
        // row-tile prefix sums must be exact
        int64_t tiles = 0;
        for (int64_t ti = st->first; ti < st->first + st->count; ti++) {
            const PotrfTask &t = sc.potrf[ti];
            if (t.stile0 != tiles) return -12;
            tiles += ((int64_t)t.lda - (t.col0 + t.nb) + SOLVE_ROWS - 1) / SOLVE_ROWS;
        }
        if (tiles != st->stiles) return -13;
        for (int phase = 0; phase < 2; phase++) {
            const bool diag = lower ? (phase == 0) : (phase == 1);
            for (int64_t ti = st->first; ti < st->first + st->count; ti++) {
                const PotrfTask &t = sc.potrf[ti];
                const int nb = t.nb;
                const double *D = Dinv + t.dinv_off;
                for (int64_t j = 0; j < nrhs; j++) {
                    double *x = X + j * d;
                    double *x1 = x + t.k1 + t.col0;
                    if (diag) {
                        std::vector<double> y(nb, 0.0);
                        for (int i = 0; i < nb; i++)
                            for (int k = 0; k < nb; k++)
                                y[i] += (lower ? D[i + k * nb] : D[k + i * nb]) * x1[k];
                        for (int i = 0; i < nb; i++) x1[i] = y[i];
                    } else {
                        for (int r = t.col0 + nb; r < t.lda; r++) {
                            int64_t row = (r < t.nscol) ? (int64_t)t.k1 + r : Ls[t.ls_off + r];
                            const double *Lr = Lx + t.a_off + (r - t.col0);
                            for (int c = 0; c < nb; c++) {
                                double l = Lr[(int64_t)c * t.lda];
                                if (lower) x[row] -= l * x1[c]; else x1[c] -= l * x[row];
                            }
                        }
                    }
                }
            }
        }
    }
    return 0;
}

int emu_super_solve(int lower, int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi,
                    const int64_t *Lpx, const int64_t *Ls, int64_t ssize, int64_t xsize, const double *Lx,
                    const double *Dinv, int64_t nrhs, double *X, int64_t d)
{
    return emu_solve_phase(lower, n, nsuper, Super, Lpi, Lpx, Ls, ssize, xsize, Lx, Dinv, nrhs, X, d, nullptr, 0);
}

// the grouped sweeps exactly as engine.hip's solve_dev walks the SolveSchedule (schedule.hpp): per step the
// group tasks (diagonal solves + updates among the SOLVE_GROUP blocks of a group), then the updates of the
// rows below the group, tile by tile (forward) or run by run (backward; mma != 0: the MFMA kernel's runs).
// Returns < 0 if the tile tables or the runs do not cover every row tile exactly once.
int emu_super_solve_grouped(int lower, int mma, int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi,
                            const int64_t *Lpx, const int64_t *Ls, int64_t ssize, int64_t xsize, const double *Lx,
                            const double *Dinv, int64_t nrhs, double *X, int64_t d)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    Schedule sc; build_schedule(S, nsuper, 0, sc);
    SolveSchedule ss; build_solve_groups(sc, ss);
    if (!check_schedule_bounds(S, sc, &ss).empty()) return -40;
    auto group_part = [&](const SolveStep &st) {
        for (int64_t gi = st.gfirst; gi < st.gfirst + st.gcount; gi++) {
            const SolveGroupTask &g = ss.groups[(size_t)gi];
            const int nblk = (g.ncols + NB - 1) / NB;
            if (nblk > st.maxblk) return -30;
            for (int64_t j = 0; j < nrhs; j++) {
                double *x = X + j * d + g.x0;
                auto diag = [&](int b) {
                    const int nb = std::min(NB, g.ncols - NB * b);
                    const double *D = Dinv + g.dinv_off[b];
                    std::vector<double> y(nb, 0.0);
                    for (int i = 0; i < nb; i++)
                        for (int k = 0; k < nb; k++) y[i] += (lower ? D[i + k * nb] : D[k + i * nb]) * x[NB * b + k];
                    for (int i = 0; i < nb; i++) x[NB * b + i] = y[i];
                };
                if (lower) {
                    for (int b = 0; b < nblk; b++) {
                        diag(b);
                        for (int r = NB * (b + 1); r < g.ncols; r++)
                            for (int k = NB * b; k < std::min(NB * (b + 1), g.ncols); k++)
                                x[r] -= Lx[g.a_off + r + (int64_t)k * g.lda] * x[k];
                    }
                } else {
                    for (int b = nblk - 1; b >= 0; b--) {
                        for (int r = NB * (b + 1); r < g.ncols; r++)
                            for (int k = NB * b; k < std::min(NB * (b + 1), g.ncols); k++)
                                x[k] -= Lx[g.a_off + r + (int64_t)k * g.lda] * x[r];
                        diag(b);
                    }
                }
            }
        }
        return 0;
    };
    // one row tile of an update task
    auto tile = [&](const PotrfTask &t, int64_t tl) {
        for (int rr = 0; rr < SOLVE_ROWS; rr++) {
            const int64_t r = t.urow0 + tl * SOLVE_ROWS + rr;
            if (r >= t.lda) break;
            const int64_t row = (r < t.nscol) ? (int64_t)t.k1 + r : Ls[t.ls_off + r];
            for (int64_t j = 0; j < nrhs; j++) {
                double *x = X + j * d;
                for (int c = 0; c < t.nb; c++) {
                    const double l = Lx[t.a_off + (r - t.col0) + (int64_t)c * t.lda];
                    if (lower) x[row] -= l * x[t.k1 + t.col0 + c]; else x[t.k1 + t.col0 + c] -= l * x[row];
                }
            }
        }
    };
    for (size_t ii = 0; ii < ss.steps.size(); ii++) {
        const SolveStep &st = ss.steps[lower ? ii : ss.steps.size() - 1 - ii];
        // every tile of every update task once: prefix sums, tile -> task table, runs
        int64_t tiles = 0;
        for (int64_t u = 0; u < st.ucount; u++) {
            const PotrfTask &t = ss.utasks[(size_t)(st.ufirst + u)];
            if (t.stile0 != tiles) return -31;
            const int64_t nt = ((int64_t)t.lda - t.urow0 + SOLVE_ROWS - 1) / SOLVE_ROWS;
            for (int64_t k = 0; k < nt; k++) if (ss.tile_task[(size_t)(st.tmap_off + tiles + k)] != u) return -32;
            tiles += nt;
        }
        if (tiles != st.stiles) return -33;
        if (lower) {
            if (int rc = group_part(st)) return rc;
            for (int64_t b = 0; b < st.stiles; b++) {
                const PotrfTask &t = ss.utasks[(size_t)(st.ufirst + ss.tile_task[(size_t)(st.tmap_off + b)])];
                tile(t, b - t.stile0);
            }
        } else {
            const std::vector<SolveRange> &rg = mma ? ss.ranges_mma : ss.ranges;
            const int64_t r0 = mma ? st.rmfirst : st.rfirst, rn = mma ? st.rmcount : st.rcount;
            int64_t covered = 0;
            for (int64_t k = r0; k < r0 + rn; k++) {
                const SolveRange &e = rg[(size_t)k];
                if (e.task < 0 || e.task >= st.ucount || e.ntiles <= 0) return -34;
                const PotrfTask &t = ss.utasks[(size_t)(st.ufirst + e.task)];
                const int64_t nt = ((int64_t)t.lda - t.urow0 + SOLVE_ROWS - 1) / SOLVE_ROWS;
                if (e.tile0 < 0 || e.tile0 + e.ntiles > nt) return -35;
                for (int64_t q = 0; q < e.ntiles; q++) tile(t, e.tile0 + q);
                covered += e.ntiles;
            }
            if (covered != st.stiles) return -36;     // (with the bounds above and the results below: exactly once)
            if (int rc = group_part(st)) return rc;
        }
    }
    return 0;
}

// check_schedule_bounds of the schedules one rank of a partition uploads (full, own subtrees, top, distributed top)
// and of their solve work lists; returns 0 or the 1-based index of the first schedule that fails (message -> msg)
int emu_validate_partition(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                           const int64_t *Ls, int64_t ssize, int64_t xsize, int cplx, const int8_t *local_mask,
                           const int8_t *top_mask, int rank, int world, char *msg, int msglen)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls; S.cplx = cplx != 0;
    Schedule full, loc, top, dst;
    build_schedule(S, nsuper, 0, full);
    build_schedule(S, nsuper, 0, loc, local_mask);
    build_schedule(S, nsuper, 0, top, top_mask, loc.dinv_size);
    DistTop dt; dt.rank = rank; dt.world = world;
    build_schedule(S, nsuper, 0, dst, top_mask, loc.dinv_size, &dt);
    // the engine sizes Dinv by the full schedule: the masked ones must fit inside
    if (top.dinv_size > full.dinv_size || dst.dinv_size != top.dinv_size) { snprintf(msg, msglen, "Dinv slots do not fit"); return 9; }
    Schedule *all[4] = {&full, &loc, &top, &dst};
    for (int k = 0; k < 4; k++) {
        SolveSchedule ss; build_solve_groups(*all[k], ss);
        std::string e = check_schedule_bounds(S, *all[k], k < 3 ? &ss : nullptr, full.dinv_size /* the device buffer */);
        if (!e.empty()) { snprintf(msg, msglen, "%s", e.c_str()); return k + 1; }
    }
    return 0;
}

// planning figures of one rank of a partition (no numeric work): out[0] executed flops of the own subtrees (phase 0),
// out[1] of this rank's share of the distributed top, out[2] of the whole top when replicated, out[3] POTRF steps
// (= length of the dependent panel chain) of phase 0, out[4] of the distributed top schedule, out[5] broadcasts,
// out[6] entries broadcast in total, out[7] entries this rank sends as root, out[8] GEMM launches of the
// distributed top, out[9] launches (all kinds) of phase 0
int emu_partition_plan(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                       const int64_t *Ls, int64_t ssize, int64_t xsize, const int8_t *local_mask,
                       const int8_t *top_mask, int rank, int world, double *out)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    Schedule loc, top, dst;
    build_schedule(S, nsuper, 0, loc, local_mask);
    build_schedule(S, nsuper, 0, top, top_mask, loc.dinv_size);
    DistTop dt; dt.rank = rank; dt.world = world;
    build_schedule(S, nsuper, 0, dst, top_mask, loc.dinv_size, &dt);
    auto count = [](const Schedule &sc, int kind) { int64_t c = 0; for (const Step &st : sc.steps) if (st.kind == kind) c++; return (double)c; };
    out[0] = loc.flops_update + loc.flops_panel;
    out[1] = dst.flops_update + dst.flops_panel;
    out[2] = top.flops_update + top.flops_panel;
    out[3] = count(loc, STEP_POTRF);
    out[4] = count(dst, STEP_POTRF);
    out[5] = (double)dst.bcasts.size();
    out[6] = out[7] = 0;
    for (const BcastPoint &b : dst.bcasts) { out[6] += (double)b.count; if (b.root == rank) out[7] += (double)b.count; }
    out[8] = count(dst, STEP_GEMM);
    out[9] = (double)loc.steps.size();
    return 0;
}

// The sweeps of a real factor with the product's solve kernels from their own source: the launch sequence of the engine's
// solve_dev (engine.hip) -- grouped scalar kernels (mma = 0, nrhs <= SOLVE_NR, group != 0), grouped MFMA kernels
// (mma != 0, group != 0), or one diagonal + one update launch per block (group = 0; scalar: any nrhs).  X: n x nrhs,
// column-major, leading dimension d.  Lx / Dinv: a factor and its block inverses (emu_super_numeric).
int emu_device_solve(int lower, int mma, int group, int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi,
                     const int64_t *Lpx, const int64_t *Ls, int64_t ssize, int64_t xsize, const double *Lx,
                     const double *Dinv, int64_t nrhs, double *X, int64_t d)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls;
    Schedule sc; build_schedule(S, nsuper, 0, sc);
    SolveSchedule ss; build_solve_groups(sc, ss);
    if (!check_schedule_bounds(S, sc, &ss).empty()) return -40;
    if (n == 0 || nrhs == 0) return 0;
    std::vector<int32_t> ls32((size_t)Lpi[nsuper] + 1);
    for (int64_t i = 0; i < Lpi[nsuper]; i++) ls32[i] = (int32_t)Ls[i];
    const int nr = (int)nrhs, nrr = nr;
    const PotrfTask *P = sc.potrf.data();
    const int32_t *dLs = ls32.data();
    double *const Xcol = X; const int64_t dcol = d;
    std::vector<double> yrow;
    if (!mma && group && nr > SOLVE_NR) return -41;
    if (!mma && group) {
        for (size_t ii = 0; ii < ss.steps.size(); ii++) {
            const SolveStep &st = ss.steps[lower ? ii : ss.steps.size() - 1 - ii];
            if (st.gcount <= 0) continue;
            const SolveGroupTask *gp = ss.groups.data() + st.gfirst;
            auto group_kernel = [&]() {
#define EMU_SG(T, NRT, MB) hipcpu::launch(st.gcount, 256, [&] { k_solve_group<T, NRT, MB>(gp, Dinv, Lx, X, d, nr); })
#define EMU_SG_MB(T, NRT) do { if (st.maxblk <= 1) EMU_SG(T, NRT, 1); else if (st.maxblk <= 2) EMU_SG(T, NRT, 2); else EMU_SG(T, NRT, SOLVE_GROUP); } while (0)
                if (lower) { if (nr == 1) EMU_SG_MB(false, 1); else EMU_SG_MB(false, SOLVE_NR); }
                else       { if (nr == 1) EMU_SG_MB(true, 1); else EMU_SG_MB(true, SOLVE_NR); }
#undef EMU_SG_MB
#undef EMU_SG
            };
            if (lower) {
                group_kernel();
                if (st.stiles > 0)
                    hipcpu::launch(st.stiles, 256, [&] { k_solve_fwd_update(ss.utasks.data() + st.ufirst, ss.tile_task.data() + st.tmap_off, Lx, dLs, X, d, nr); });
            } else {
                if (st.rcount > 0) {
                    if (nr == 1) hipcpu::launch(st.rcount, 256, [&] { k_solve_bwd_range<1>(ss.utasks.data() + st.ufirst, ss.ranges.data() + st.rfirst, Lx, dLs, X, d, nr); });
                    else hipcpu::launch(st.rcount, 256, [&] { k_solve_bwd_range<SOLVE_NR>(ss.utasks.data() + st.ufirst, ss.ranges.data() + st.rfirst, Lx, dLs, X, d, nr); });
                }
                group_kernel();
            }
        }
        return 0;
    }
    if (mma) {
        const int64_t ldr = ((int64_t)nrr + 15) & ~(int64_t)15;
        yrow.assign((size_t)n * (size_t)ldr, 0.0);
        hipcpu::launch((n + 63) / 64, 256, [&] { k_solve_transpose<double>(n, nr, Xcol, dcol, yrow.data(), ldr, 1); });
        X = yrow.data(); d = ldr;
    }
    if (mma && group) {
        for (size_t ii = 0; ii < ss.steps.size(); ii++) {
            const SolveStep &st = ss.steps[lower ? ii : ss.steps.size() - 1 - ii];
            if (st.gcount <= 0) continue;
            const SolveGroupTask *gp = ss.groups.data() + st.gfirst;
            const unsigned gy = (unsigned)((nrr + SG_CW - 1) / SG_CW);
            auto group_kernel = [&]() {
#define EMU_SGM(T, MB) hipcpu::launch(st.gcount, 256, [&] { k_solve_group_mma<T, 1, MB>(gp, Dinv, Lx, X, d, nrr); }, gy)
#define EMU_SGM_MB(T) do { if (st.maxblk <= 1) EMU_SGM(T, 1); else if (st.maxblk <= 2) EMU_SGM(T, 2); else EMU_SGM(T, SOLVE_GROUP); } while (0)
                if (lower) EMU_SGM_MB(false); else EMU_SGM_MB(true);
#undef EMU_SGM_MB
#undef EMU_SGM
            };
            const int grp = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / std::max(1, tuning().solve_fwd_div_mma)));
            const int64_t grid = (st.stiles + grp - 1) / grp;
            if (lower) {
                group_kernel();
                if (st.stiles > 0)
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipcpu::launch(grid, 256, [&] { k_solve_fwd_mma<1>(ss.utasks.data() + st.ufirst, ss.tile_task.data() + st.tmap_off, st.stiles, grp, Lx, dLs, X, d, nrr, j0); });
            } else {
                if (st.rmcount > 0)
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipcpu::launch(st.rmcount, 256, [&] { k_solve_bwd_mma_run<1>(ss.utasks.data() + st.ufirst, ss.ranges_mma.data() + st.rmfirst, Lx, dLs, X, d, nrr, j0); });
                group_kernel();
            }
        }
    } else if (lower) {
        for (const Step &st : sc.steps) {
            if (st.kind != STEP_POTRF || st.count <= 0) continue;
            if (mma) hipcpu::launch(st.count, 256, [&] { k_solve_diag_mma<false, 1>(P + st.first, Dinv, X, d, nrr); });
            else hipcpu::launch(st.count, 256, [&] { k_solve_diag<false>(P + st.first, Dinv, X, d, nr); });
            if (st.stiles > 0) {
                const int32_t *tm = sc.solve_tile_task.data() + st.tmap_off;
                if (mma) {
                    const int grp = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / 1024));
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipcpu::launch((st.stiles + grp - 1) / grp, 256, [&] { k_solve_fwd_mma<1>(P + st.first, tm, st.stiles, grp, Lx, dLs, X, d, nrr, j0); });
                } else
                    hipcpu::launch(st.stiles, 256, [&] { k_solve_fwd_update(P + st.first, tm, Lx, dLs, X, d, nr); });
            }
        }
    } else {
        for (size_t i = sc.steps.size(); i-- > 0;) {
            const Step &st = sc.steps[i];
            if (st.kind != STEP_POTRF || st.count <= 0) continue;
            if (st.stiles > 0) {
                const int32_t *tm = sc.solve_tile_task.data() + st.tmap_off;
                if (mma) {
                    const int grp = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / 1024));
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipcpu::launch((st.stiles + grp - 1) / grp, 256, [&] { k_solve_bwd_mma<1>(P + st.first, tm, st.stiles, grp, Lx, dLs, X, d, nrr, j0); });
                } else if (nr <= SOLVE_NR) {
                    const int grp = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / 128));
                    const int64_t grid = (st.stiles + grp - 1) / grp;
                    if (nr == 1) hipcpu::launch(grid, 256, [&] { k_solve_bwd_update<1>(P + st.first, tm, st.stiles, grp, Lx, dLs, X, d, nr); });
                    else hipcpu::launch(grid, 256, [&] { k_solve_bwd_update<SOLVE_NR>(P + st.first, tm, st.stiles, grp, Lx, dLs, X, d, nr); });
                } else
                    hipcpu::launch(st.stiles, 256, [&] { k_solve_bwd_update<0>(P + st.first, tm, st.stiles, 1, Lx, dLs, X, d, nr); });
            }
            if (mma) hipcpu::launch(st.count, 256, [&] { k_solve_diag_mma<true, 1>(P + st.first, Dinv, X, d, nrr); });
            else hipcpu::launch(st.count, 256, [&] { k_solve_diag<true>(P + st.first, Dinv, X, d, nr); });
        }
    }
    if (mma)
        hipcpu::launch((n + 63) / 64, 256, [&] { k_solve_transpose<double>(n, nr, Xcol, dcol, yrow.data(), d, 0); });
    return 0;
}

#ifdef __clang__
// A complex (interleaved) factorization entirely by the product's kernels from their own source: k_assemble_c, k_relmap,
// k_potrf64c, k_gemm64c (small tasks and TRSM) and k_gemm128<true> (big tasks through the real view), with the engine's
// launch sequence (run_schedule) and its not-positive-definite protocol (mi355_factor_numeric).  Ax, Lx: E = 2.
int emu_device_numeric_c(int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi, const int64_t *Lpx,
                         const int64_t *Ls, int64_t ssize, int64_t xsize, const int64_t *Ap, const int64_t *Ai,
                         const double *Ax, double beta, int quick, double *Lx, int64_t *minor, int *status, int64_t *nbig,
                         double *Dinv_out /* 2 * emu_dinv_size doubles, or NULL */)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls; S.cplx = true;
    if (!validate_symbolic(S).empty()) return -4;
    std::vector<int32_t> ls32((size_t)Lpi[nsuper] + 1);
    for (int64_t i = 0; i < Lpi[nsuper]; i++) ls32[i] = (int32_t)Ls[i];
    std::vector<int> info;
    std::vector<double> Dinv;
    int64_t big_launches = 0;
    auto run_c = [&](const Schedule &sc, int64_t ncols_asm) -> int {
        if (!check_schedule_bounds(S, sc).empty()) return -10;
        std::fill(Lx, Lx + 2 * xsize, 0.0);
        info.assign(nsuper + 2, 0);
        Dinv.assign(2 * (size_t)sc.dinv_size + 2, 0.0);
        const int8_t *nomask = nullptr;
        if (ncols_asm > 0)
            hipcpu::launch_serial((ncols_asm + 255) / 256, 256, [&] { k_assemble_c(ncols_asm, Ap, Ai, nullptr, Ax, sc.supermap.data(), Super, Lpi, Lpx,
                                                                                   ls32.data(), Lx, beta, nomask); });
        std::vector<int32_t> relmap(sc.relmap_size + 1);
        if (sc.relmap_size > 0) {
            std::vector<RelmapPair> rp(sc.pairs.size());
            for (size_t i = 0; i < rp.size(); i++) {
                rp[i].ls_src = sc.pairs[i].ls_src; rp[i].ls_dst = sc.pairs[i].ls_dst;
                rp[i].rm_off = sc.pairs[i].rm_off; rp[i].nsrow = sc.pairs[i].nsrow; rp[i].pad = 0;
            }
            int err = 0;
            hipcpu::launch_serial((sc.relmap_size + 255) / 256, 256, [&] { k_relmap(sc.relmap_size, (int64_t)rp.size(), rp.data(), ls32.data(), relmap.data(), &err); });
            if (err) return -11;
        }
        for (const Step &st : sc.steps) {
            if (st.count <= 0) continue;
            if (st.kind == STEP_POTRF) {
                hipcpu::launch(st.count, 256, [&] { k_potrf64c(sc.potrf.data() + st.first, Lx, Dinv.data(), info.data()); });
            } else {
                const GemmTask *tk = sc.gemm.data() + st.first;
                const Schedule::TileEntry *tm = sc.gemm_tile_tab.data() + st.tmap_off;
                if (st.big) {
                    hipcpu::launch(st.ntiles, 256, [&] { k_gemm128<true>(tk, tm, Lx, Dinv.data(), relmap.data(), (unsigned long long *)nullptr); });
                    big_launches++;
                } else
                    hipcpu::launch(st.ntiles, 256, [&] { k_gemm64c(tk, tm, Lx, Dinv.data(), relmap.data()); });
                if (!check_tile_table(sc, st)) return -12;
            }
        }
        return 0;
    };
    Schedule sc; build_schedule(S, nsuper, 0, sc);
    if (int rc = run_c(sc, n)) return rc;
    *status = 0; *minor = n;
    int64_t sf = -1;
    for (int64_t s = 0; s < nsuper; s++) if (info[s]) { sf = s; break; }
    if (sf >= 0) {
        const int inf = info[sf];
        *status = 1; *minor = Super[sf] + inf - 1;
        const int64_t nscol2 = (inf == 1 || quick) ? 0 : inf - 1;
        Schedule tr; build_schedule(S, sf, nscol2, tr);
        if (int rc = run_c(tr, nscol2 > 0 ? Super[sf + 1] : Super[sf])) return rc;
        const int64_t nsrow = Lpi[sf + 1] - Lpi[sf], tail = Lpx[sf] + nsrow * nscol2;
        std::fill(Lx + 2 * tail, Lx + 2 * xsize, 0.0);
    }
    if (nbig) *nbig = big_launches;
    if (Dinv_out && sf < 0) std::copy(Dinv.begin(), Dinv.begin() + 2 * sc.dinv_size, Dinv_out);
    return 0;
}

// The sweeps of a COMPLEX factor with the product's complex solve kernels from their own source, in solve_dev's launch
// order: mma = 0: k_solve_diag_c + k_solve_fwd_update_c / k_solve_bwd_update_c per block; mma != 0: the MFMA sweeps
// on the real / imaginary row-major image (k_solve_transpose<double2>, k_solve_group_mma<.., 2, ..> + k_solve_fwd_mma<2> /
// k_solve_bwd_mma_run<2> when group != 0, else k_solve_diag_mma<.., 2> + k_solve_fwd_mma<2> / k_solve_bwd_mma<2>).
// X: n x nrhs complex entries (interleaved), column-major, leading dimension d entries.
int emu_device_solve_c(int lower, int mma, int group, int64_t n, int64_t nsuper, const int64_t *Super, const int64_t *Lpi,
                       const int64_t *Lpx, const int64_t *Ls, int64_t ssize, int64_t xsize, const double *Lx,
                       const double *Dinv, int64_t nrhs, double *X, int64_t d)
{
    Symbolic S; S.n = n; S.nsuper = nsuper; S.ssize = ssize; S.xsize = xsize;
    S.Super = Super; S.Lpi = Lpi; S.Lpx = Lpx; S.Ls = Ls; S.cplx = true;
    Schedule sc; build_schedule(S, nsuper, 0, sc);
    SolveSchedule ss; build_solve_groups(sc, ss);
    if (!check_schedule_bounds(S, sc, &ss).empty()) return -40;
    if (n == 0 || nrhs == 0) return 0;
    std::vector<int32_t> ls32((size_t)Lpi[nsuper] + 1);
    for (int64_t i = 0; i < Lpi[nsuper]; i++) ls32[i] = (int32_t)Ls[i];
    const int nr = (int)nrhs, nrr = 2 * nr;
    const PotrfTask *P = sc.potrf.data();
    const int32_t *dLs = ls32.data();
    double *const Xcol = X; const int64_t dcol = d;
    std::vector<double> yrow;
    if (mma) {
        const int64_t ldr = ((int64_t)nrr + 15) & ~(int64_t)15;
        yrow.assign((size_t)n * (size_t)ldr, 0.0);
        hipcpu::launch((n + 63) / 64, 256, [&] { k_solve_transpose<double2>(n, nr, reinterpret_cast<double2 *>(Xcol), dcol,
                                                                            reinterpret_cast<double2 *>(yrow.data()), ldr / 2, 1); });
        X = yrow.data(); d = ldr;
    }
    if (mma && group) {
        for (size_t ii = 0; ii < ss.steps.size(); ii++) {
            const SolveStep &st = ss.steps[lower ? ii : ss.steps.size() - 1 - ii];
            if (st.gcount <= 0) continue;
            const SolveGroupTask *gp = ss.groups.data() + st.gfirst;
            const unsigned gy = (unsigned)((nrr + SG_CW - 1) / SG_CW);
            auto group_kernel = [&]() {
#define EMU_SGM(T, MB) hipcpu::launch(st.gcount, 256, [&] { k_solve_group_mma<T, 2, MB>(gp, Dinv, Lx, X, d, nrr); }, gy)
#define EMU_SGM_MB(T) do { if (st.maxblk <= 1) EMU_SGM(T, 1); else if (st.maxblk <= 2) EMU_SGM(T, 2); else EMU_SGM(T, SOLVE_GROUP); } while (0)
                if (lower) EMU_SGM_MB(false); else EMU_SGM_MB(true);
#undef EMU_SGM_MB
#undef EMU_SGM
            };
            const int grp = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / std::max(1, tuning().solve_fwd_div_mma)));
            const int64_t grid = (st.stiles + grp - 1) / grp;
            if (lower) {
                group_kernel();
                if (st.stiles > 0)
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipcpu::launch(grid, 256, [&] { k_solve_fwd_mma<2>(ss.utasks.data() + st.ufirst, ss.tile_task.data() + st.tmap_off, st.stiles, grp, Lx, dLs, X, d, nrr, j0); });
            } else {
                if (st.rmcount > 0)
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipcpu::launch(st.rmcount, 256, [&] { k_solve_bwd_mma_run<2>(ss.utasks.data() + st.ufirst, ss.ranges_mma.data() + st.rmfirst, Lx, dLs, X, d, nrr, j0); });
                group_kernel();
            }
        }
    } else if (lower) {
        for (const Step &st : sc.steps) {
            if (st.kind != STEP_POTRF || st.count <= 0) continue;
            if (mma) hipcpu::launch(st.count, 256, [&] { k_solve_diag_mma<false, 2>(P + st.first, Dinv, X, d, nrr); });
            else hipcpu::launch(st.count, 256, [&] { k_solve_diag_c<false>(P + st.first, Dinv, X, d, nr); });
            if (st.stiles > 0) {
                const int32_t *tm = sc.solve_tile_task.data() + st.tmap_off;
                if (mma) {
                    const int grp = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / 1024));
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipcpu::launch((st.stiles + grp - 1) / grp, 256, [&] { k_solve_fwd_mma<2>(P + st.first, tm, st.stiles, grp, Lx, dLs, X, d, nrr, j0); });
                } else
                    hipcpu::launch(st.stiles, SOLVE_ROWS, [&] { k_solve_fwd_update_c(P + st.first, tm, Lx, dLs, X, d, nr); });
            }
        }
    } else {
        for (size_t i = sc.steps.size(); i-- > 0;) {
            const Step &st = sc.steps[i];
            if (st.kind != STEP_POTRF || st.count <= 0) continue;
            if (st.stiles > 0) {
                const int32_t *tm = sc.solve_tile_task.data() + st.tmap_off;
                if (mma) {
                    const int grp = (int)std::min<int64_t>(16, std::max<int64_t>(1, st.stiles / 1024));
                    for (int j0 = 0; j0 < nrr; j0 += 64)
                        hipcpu::launch((st.stiles + grp - 1) / grp, 256, [&] { k_solve_bwd_mma<2>(P + st.first, tm, st.stiles, grp, Lx, dLs, X, d, nrr, j0); });
                } else
                    hipcpu::launch(st.stiles, 256, [&] { k_solve_bwd_update_c(P + st.first, tm, Lx, dLs, X, d, nr); });
            }
            if (mma) hipcpu::launch(st.count, 256, [&] { k_solve_diag_mma<true, 2>(P + st.first, Dinv, X, d, nrr); });
            else hipcpu::launch(st.count, 256, [&] { k_solve_diag_c<true>(P + st.first, Dinv, X, d, nr); });
        }
    }
    if (mma)
        hipcpu::launch((n + 63) / 64, 256, [&] { k_solve_transpose<double2>(n, nr, reinterpret_cast<double2 *>(Xcol), dcol,
                                                                            reinterpret_cast<double2 *>(yrow.data()), d / 2, 0); });
    return 0;
}
#endif
}
