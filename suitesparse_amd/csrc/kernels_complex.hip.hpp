// kernels_complex.hip.hpp -- complex (Hermitian, LL^H) variants of the numeric and solve
// kernels: the device side of the cd_/zd_ instantiations of the reference worker
// (CHOLMOD/Supernodal/t_cholmod_super_numeric_worker.c:35-83: L is stored interleaved
// re,im; zherk/zgemm "N","C"; zpotrf; ztrsm "R","L","C","N") and of the complex solve
// workers (t_cholmod_super_solve_worker.c:127-161,246-266,420-448,520-560).
//
// Same schedule, same task structs, same tiling as the real kernels; every offset in a task
// is an ENTRY offset (16 bytes per entry here).  A complex product A*B^H is four real MFMA
// products on de-interleaved LDS images:  Cr = Ar Br^T + Ai Bi^T,  Ci = Ai Br^T - Ar Bi^T.
// As in LAPACK's zpotrf/zherk the imaginary part of a diagonal entry is never read and is
// written as exactly zero.
#pragma once
#include "kernels.hip.hpp"

namespace mi355 {

typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_assemble_c(
    int64_t ncols, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
    const int64_t *__restrict__ Anz, const double *__restrict__ Ax,
    const int32_t *__restrict__ supermap, const int64_t *__restrict__ Super,
    const int64_t *__restrict__ Lpi, const int64_t *__restrict__ Lpx,
    const int32_t *__restrict__ Ls, double *__restrict__ Lx, double beta,
    const int8_t *__restrict__ smask)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncols) return;
    const int s = supermap[k];
    if (smask && !smask[s]) return;
    const int64_t k1 = Super[s], psi = Lpi[s], psx = Lpx[s];
    const int nsrow = (int)(Lpi[s + 1] - psi);
    const int64_t colbase = psx + (k - k1) * (int64_t)nsrow;
    int64_t p = Ap[k];
    const int64_t pend = Anz ? p + Anz[k] : Ap[k + 1];
    for (; p < pend; p++) {
        const int64_t i = Ai[p];
        if (i >= k) {
            const int imap = find_row(Ls + psi, nsrow, (int)i);
            if (imap >= 0) {
                Lx[2 * (colbase + imap)] = Ax[2 * p];
                Lx[2 * (colbase + imap) + 1] = Ax[2 * p + 1];
            }
        }
    }
    if (beta != 0.0) Lx[2 * (colbase + (k - k1))] += beta;      // only the real part of beta is used (worker :510-520)
}

//------------------------------------------------------------------------------
// k_gemm64c: complex C = A * B^H, 64x64 tile, 4 waves (2x2), 2x2 MFMA blocks x {re,im}
//------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gemm64c(
    const GemmTask *__restrict__ tasks, const Schedule::TileEntry *__restrict__ tile_tab, double *__restrict__ Lx,
    const double *__restrict__ Dinv, const int32_t *__restrict__ relmap)
{
    __shared__ double Ar[G64_KS][G64_LD], Ai_[G64_KS][G64_LD];
    __shared__ double Br[G64_KS][G64_LD], Bi_[G64_KS][G64_LD];
    __shared__ int rmM[TILE], rmN[TILE];

    const int tid = threadIdx.x;
    const int64_t bid = blockIdx.x;
    const Schedule::TileEntry te = tile_tab[bid];
    const GemmTask t = tasks[te.task];
    const int M = t.M, N = t.N, K = t.K, mode = t.mode;
    const int tm = (int)(te.tmtn & 0xffffu), tn = (int)(te.tmtn >> 16);
    const int m0 = tm * TILE, n0 = tn * TILE;

    const v2d *__restrict__ A = (const v2d *)Lx + t.a_off;
    const v2d *__restrict__ B = (const v2d *)(mode == GEMM_OVERWRITE ? Dinv : Lx) + t.b_off;
    const int64_t lda = t.lda, ldb = t.ldb;

    if (mode == GEMM_SCATTER_SUB) {
        if (tid < TILE) rmM[tid] = (m0 + tid < M) ? relmap[t.rm_off + m0 + tid] : 0;
        else if (tid < 2 * TILE) { int j = tid - TILE; rmN[j] = (n0 + j < N) ? relmap[t.rm_off + n0 + j] : 0; }
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    v4d accr[2][2], acci[2][2];                  // [ni][mi]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) { accr[a][b] = (v4d){0.0, 0.0, 0.0, 0.0}; acci[a][b] = accr[a][b]; }

    const int lrow = tid & 63, lk = tid >> 6;
    const bool arow_ok = (m0 + lrow) < M, brow_ok = (n0 + lrow) < N;
    const v2d *Ap = A + (m0 + lrow);
    const v2d *Bp = B + (n0 + lrow);
    const v2d zero = {0.0, 0.0};

    for (int k0 = 0; k0 < K; k0 += G64_KS) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = lk + 4 * q, kk = k0 + k;
            const bool kok = kk < K;
            const v2d av = (arow_ok && kok) ? Ap[(int64_t)kk * lda] : zero;
            const v2d bv = (brow_ok && kok) ? Bp[(int64_t)kk * ldb] : zero;
            Ar[k][lrow] = av.x; Ai_[k][lrow] = av.y;
            Br[k][lrow] = bv.x; Bi_[k][lrow] = bv.y;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int kk = ks * 4 + l4;
            double ar[2], ai[2], br[2], bi[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                ar[i] = Ar[kk][wm * 32 + i * 16 + l15];
                ai[i] = Ai_[kk][wm * 32 + i * 16 + l15];
                br[i] = Br[kk][wn * 32 + i * 16 + l15];
                bi[i] = Bi_[kk][wn * 32 + i * 16 + l15];
            }
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++) {
                    accr[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[ni], ar[mi], accr[ni][mi], 0, 0, 0);
                    accr[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[ni], ai[mi], accr[ni][mi], 0, 0, 0);
                    acci[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(br[ni], ai[mi], acci[ni][mi], 0, 0, 0);
                    acci[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(-bi[ni], ar[mi], acci[ni][mi], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    const int64_t ldc = t.ldc;
    double *__restrict__ C = Lx + 2 * t.c_off;
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
#pragma unroll
        for (int mi = 0; mi < 2; mi++) {
            const int ml = wm * 32 + mi * 16 + l15;
            const int m = m0 + ml;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nl = wn * 32 + ni * 16 + l4 + 4 * r;
                const int n = n0 + nl;
                if (m < M && n < N) {
                    const double vr = accr[ni][mi][r], vi = acci[ni][mi][r];
                    if (mode == GEMM_SCATTER_SUB) {
                        if (m >= n) {
                            double *q = C + 2 * (rmM[ml] + (int64_t)rmN[nl] * ldc);
                            atomic_add_f64(q, -vr);
                            if (m != n) atomic_add_f64(q + 1, -vi);     // zherk: diagonal of C is real
                        }
                    } else if (mode == GEMM_DIRECT_SUB) {
                        if (m >= n) {
                            double *q = C + 2 * (m + (int64_t)n * ldc);
                            q[0] -= vr;
                            if (m != n) q[1] -= vi;
                        }
                    } else {
                        double *q = C + 2 * (m + (int64_t)n * ldc);
                        q[0] = vr; q[1] = vi;
                    }
                }
            }
        }
    }
}

//------------------------------------------------------------------------------
// k_potrf64c: zpotrf of one nb x nb (nb <= 64) diagonal block + its inverse.  Same construction as
// k_potrf64 (kernels.hip.hpp): right-looking over 16-column blocks; the 16 x 16 diagonal block and its
// inverse in the registers of one wave (lane = row, values move between lanes with v_readlane), the block
// TRSM, trailing update and the assembly of the 64 x 64 inverse as 16x16x4 FP64 MFMA products -- four real
// products per complex one, on de-interleaved LDS images.  The imaginary part of a pivot is ignored and
// stored as zero (zpotrf).
//------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf64c(
    const PotrfTask *__restrict__ tasks, double *__restrict__ Lx,
    double *__restrict__ Dinv, int *__restrict__ info)
{
    __shared__ double Lr[NB][NB + 1], Li[NB][NB + 1];         // the block, factorized in place
    __shared__ double Vr[NB][NB + 1], Vi[NB][NB + 1];         // its inverse
    const PotrfTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x;
    const int nb = t.nb;
    const int64_t lda = t.lda;
    v2d *__restrict__ A = (v2d *)Lx + t.a_off;

    const int nblk = (nb + 15) >> 4;                          // 16-column blocks that hold anything
    const int np = nblk * 16;                                 // padded order: identity beyond nb
    for (int idx = tid; idx < np * np; idx += 256) {
        const int i = idx % np, j = idx / np;
        double vr = (i == j) ? 1.0 : 0.0, vi = 0.0;           // identity padding
        if (i < nb && j < nb) {
            if (i >= j) { const v2d v = A[i + (int64_t)j * lda]; vr = v.x; vi = (i == j) ? 0.0 : v.y; }
            else vr = 0.0;
        }
        Lr[i][j] = vr; Li[i][j] = vi;
        Vr[i][j] = 0.0; Vi[i][j] = 0.0;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    // P += X * Y^H on 16 x 16 blocks (rows of X and Y along the first index, contraction along the second):
    // Pr += Xr Yr^T + Xi Yi^T, Pi += Xi Yr^T - Xr Yi^T; result element [row = lane/16 + 4g][col = lane%16]
    auto mma_xyh = [&](const double (*Xr_)[NB + 1], const double (*Xi_)[NB + 1], int xr0, int xc0,
                       const double (*Yr_)[NB + 1], const double (*Yi_)[NB + 1], int yr0, int yc0, v4d &pr, v4d &pi) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double xr = Xr_[xr0 + l15][xc0 + 4 * q + l4], xi = Xi_[xr0 + l15][xc0 + 4 * q + l4];
            const double yr = Yr_[yr0 + l15][yc0 + 4 * q + l4], yi = Yi_[yr0 + l15][yc0 + 4 * q + l4];
            pr = __builtin_amdgcn_mfma_f64_16x16x4f64(xr, yr, pr, 0, 0, 0);
            pr = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, yi, pr, 0, 0, 0);
            pi = __builtin_amdgcn_mfma_f64_16x16x4f64(xi, yr, pi, 0, 0, 0);
            pi = __builtin_amdgcn_mfma_f64_16x16x4f64(-xr, yi, pi, 0, 0, 0);
        }
    };
    int fail_col = 0;                                         // wave 0 only: 1-based column of the first bad pivot
#pragma unroll 1
    for (int jb = 0; jb < nblk; jb++) {
        const int c0 = 16 * jb;
        if (wave == 0) {
            // lane r (mod 16) holds row r of the diagonal block; entries right of the diagonal are never
            // consumed (they collect harmless junk) and are written back as zeros
            double ar[16], ai[16], rs[16], wr[16], wi[16];
#pragma unroll
            for (int k = 0; k < 16; k++) { ar[k] = Lr[c0 + l15][c0 + k]; ai[k] = Li[c0 + l15][c0 + k]; }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const double d = lane_bcast(ar[j], j);        // pivot: real part only
                fail_col = (!(d > 0.0) && fail_col == 0) ? c0 + j + 1 : fail_col;
                const double r = rsqrt(d);
                rs[j] = r;
                ar[j] = (l15 == j) ? d * r : ar[j] * r;
                ai[j] = (l15 == j) ? 0.0 : ai[j] * r;
#pragma unroll
                for (int k = j + 1; k < 16; k++) {             // a(i,k) -= l(i,j) * conj(l(k,j))
                    const double br = lane_bcast(ar[j], k), bi = lane_bcast(ai[j], k);
                    ar[k] -= ar[j] * br + ai[j] * bi;
                    ai[k] -= ai[j] * br - ar[j] * bi;
                }
                __builtin_amdgcn_sched_barrier(0);            // keeps the broadcast values (SGPRs) short-lived
            }
            // W = inv(L_jj): lane c holds column c, w[i] = (delta_ic - sum_{k<i} L[i][k] w[k]) / L[i][i]
#pragma unroll
            for (int i = 0; i < 16; i++) {
                double accr = (l15 == i) ? 1.0 : 0.0, acci = 0.0;
#pragma unroll
                for (int k = 0; k < i; k++) {
                    const double lr = lane_bcast(ar[k], i), li = lane_bcast(ai[k], i);
                    accr -= lr * wr[k] - li * wi[k];
                    acci -= lr * wi[k] + li * wr[k];
                }
                wr[i] = accr * rs[i];
                wi[i] = acci * rs[i];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    Lr[c0 + lane][c0 + k] = (k <= lane) ? ar[k] : 0.0;
                    Li[c0 + lane][c0 + k] = (k <= lane) ? ai[k] : 0.0;
                    Vr[c0 + k][c0 + lane] = wr[k];
                    Vi[c0 + k][c0 + lane] = wi[k];
                }
            }
        }
        __syncthreads();
        // TRSM: X_ib = A(ib,jb) * W^H for the block rows below (one per wave)
        {
            const int ib = jb + 1 + wave;
            if (ib < nblk) {
                v4d xr = {0.0, 0.0, 0.0, 0.0}, xi = {0.0, 0.0, 0.0, 0.0};
                mma_xyh(Lr, Li, 16 * ib, c0, Vr, Vi, c0, c0, xr, xi);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    Lr[16 * ib + l4 + 4 * g][c0 + l15] = xr[g];
                    Li[16 * ib + l4 + 4 * g][c0 + l15] = xi[g];
                }
            }
        }
        __syncthreads();
        // trailing update: A(ib,kb) -= X_ib X_kb^H for jb < kb <= ib < nblk, block pairs dealt to the waves
        {
            const int nrem = nblk - 1 - jb, npairs = nrem * (nrem + 1) / 2;
            for (int p = wave; p < npairs; p += 4) {
                int ib = 0, q = p;
                while (q > ib) { q -= ib + 1; ib++; }          // p -> (ib, q): q <= ib
                const int bi_ = jb + 1 + ib, bk = jb + 1 + q;
                v4d cr = {0.0, 0.0, 0.0, 0.0}, ci = {0.0, 0.0, 0.0, 0.0};
                mma_xyh(Lr, Li, 16 * bi_, c0, Lr, Li, 16 * bk, c0, cr, ci);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    Lr[16 * bi_ + l4 + 4 * g][16 * bk + l15] -= cr[g];
                    Li[16 * bi_ + l4 + 4 * g][16 * bk + l15] -= ci[g];
                }
            }
        }
        __syncthreads();
    }

    if (tid == 0 && fail_col != 0 && info[t.s] == 0) info[t.s] = t.col0 + fail_col;
    for (int idx = tid; idx < nb * nb; idx += 256) {
        const int i = idx % nb, j = idx / nb;
        if (i >= j) { v2d v = {Lr[i][j], (i == j) ? 0.0 : Li[i][j]}; A[i + (int64_t)j * lda] = v; }
    }

    // off-diagonal blocks of the inverse, by distance d from the diagonal:
    // V(i,j) = -W_i * sum_{k=j}^{i-1} L(i,k) V(k,j) (plain complex products); the sum's MFMA result registers
    // are the B operand of W_i * (.)
    for (int d = 1; d < nblk; d++) {
        const int j = wave, i = wave + d;
        if (i < nblk) {
            v4d tr = {0.0, 0.0, 0.0, 0.0}, ti = {0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; k++) {
#pragma unroll
                for (int z = 0; z < 4; z++) {
                    const double lr = Lr[16 * i + l15][16 * k + 4 * z + l4], li = Li[16 * i + l15][16 * k + 4 * z + l4];
                    const double vr = Vr[16 * k + 4 * z + l4][16 * j + l15], vi = Vi[16 * k + 4 * z + l4][16 * j + l15];
                    tr = __builtin_amdgcn_mfma_f64_16x16x4f64(lr, vr, tr, 0, 0, 0);
                    tr = __builtin_amdgcn_mfma_f64_16x16x4f64(-li, vi, tr, 0, 0, 0);
                    ti = __builtin_amdgcn_mfma_f64_16x16x4f64(lr, vi, ti, 0, 0, 0);
                    ti = __builtin_amdgcn_mfma_f64_16x16x4f64(li, vr, ti, 0, 0, 0);
                }
            }
            v4d vr = {0.0, 0.0, 0.0, 0.0}, vi = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int z = 0; z < 4; z++) {
                const double wr = Vr[16 * i + l15][16 * i + 4 * z + l4], wi = Vi[16 * i + l15][16 * i + 4 * z + l4];
                vr = __builtin_amdgcn_mfma_f64_16x16x4f64(wr, tr[z], vr, 0, 0, 0);
                vr = __builtin_amdgcn_mfma_f64_16x16x4f64(-wi, ti[z], vr, 0, 0, 0);
                vi = __builtin_amdgcn_mfma_f64_16x16x4f64(wr, ti[z], vi, 0, 0, 0);
                vi = __builtin_amdgcn_mfma_f64_16x16x4f64(wi, tr[z], vi, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                Vr[16 * i + l4 + 4 * g][16 * j + l15] = -vr[g];
                Vi[16 * i + l4 + 4 * g][16 * j + l15] = -vi[g];
            }
        }
        __syncthreads();
    }
    if (t.dinv_off >= 0) {
        v2d *__restrict__ D = (v2d *)Dinv + t.dinv_off;
        for (int idx = tid; idx < nb * nb; idx += 256) {
            const int i = idx % nb, j = idx / nb;
            v2d v = {0.0, 0.0};
            if (i >= j) { v.x = Vr[i][j]; v.y = Vi[i][j]; }
            D[idx] = v;
        }
    }
}

//------------------------------------------------------------------------------
// complex solves
//------------------------------------------------------------------------------
// x1 <- inv(L11) x1 (TRANS = false)  or  x1 <- inv(L11)^H x1 (TRANS = true)
template <bool TRANS>
__global__ __launch_bounds__(256) void k_solve_diag_c(
    const PotrfTask *__restrict__ tasks, const double *__restrict__ Dinv,
    double *__restrict__ X, int64_t ldx, int nrhs)
{
    __shared__ double Dr[NB][NB + 1], Di[NB][NB + 1];
    __shared__ double xr[NB][8 + 1], xi[NB][8 + 1];
    const PotrfTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, nb = t.nb;
    const v2d *__restrict__ D = (const v2d *)Dinv + t.dinv_off;
    for (int idx = tid; idx < nb * nb; idx += 256) {
        const int i = idx % nb, j = idx / nb;
        const v2d v = D[idx];
        Dr[i][j] = v.x; Di[i][j] = v.y;
    }
    v2d *__restrict__ x1 = (v2d *)X + t.k1 + t.col0;
    for (int j0 = 0; j0 < nrhs; j0 += 8) {
        const int nj = min(8, nrhs - j0);
        __syncthreads();
        for (int idx = tid; idx < nb * nj; idx += 256) {
            const int i = idx % nb, j = idx / nb;
            const v2d v = x1[i + (int64_t)(j0 + j) * ldx];
            xr[i][j] = v.x; xi[i][j] = v.y;
        }
        __syncthreads();
        for (int idx = tid; idx < nb * nj; idx += 256) {
            const int i = idx % nb, j = idx / nb;
            double ar = 0.0, ai = 0.0;
            if (!TRANS) {
                for (int k = 0; k <= i; k++) {
                    const double dr = Dr[i][k], di = Di[i][k];
                    ar += dr * xr[k][j] - di * xi[k][j];
                    ai += dr * xi[k][j] + di * xr[k][j];
                }
            } else {
                for (int k = i; k < nb; k++) {
                    const double dr = Dr[k][i], di = -Di[k][i];       // conj(D(k,i))
                    ar += dr * xr[k][j] - di * xi[k][j];
                    ai += dr * xi[k][j] + di * xr[k][j];
                }
            }
            v2d v = {ar, ai};
            x1[i + (int64_t)(j0 + j) * ldx] = v;
        }
    }
}

__global__ __launch_bounds__(SOLVE_ROWS) void k_solve_fwd_update_c(
    const PotrfTask *__restrict__ tasks, const int32_t *__restrict__ tile_task, const double *__restrict__ Lx,
    const int32_t *__restrict__ Ls, double *__restrict__ X, int64_t ldx, int nrhs)
{
    constexpr int NR = 2;
    __shared__ double yr[NB][NR], yi[NB][NR];
    const PotrfTask t = tasks[tile_task[blockIdx.x]];
    const int tid = threadIdx.x, nb = t.nb;
    const int r = t.col0 + nb + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS + tid;
    const bool ok = r < t.lda;
    const v2d *__restrict__ Lr = (const v2d *)Lx + t.a_off + (r - t.col0);
    int64_t row = 0;
    if (ok) row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
    const v2d *__restrict__ x1 = (const v2d *)X + t.k1 + t.col0;
    for (int j0 = 0; j0 < nrhs; j0 += NR) {
        const int nj = min(NR, nrhs - j0);
        __syncthreads();
        for (int idx = tid; idx < nb * NR; idx += SOLVE_ROWS) {
            const int c = idx % nb, j = idx / nb;
            v2d v = {0.0, 0.0};
            if (j < nj) v = x1[c + (int64_t)(j0 + j) * ldx];
            yr[c][j] = v.x; yi[c][j] = v.y;
        }
        __syncthreads();
        if (ok) {
            double ar[NR], ai[NR];
#pragma unroll
            for (int j = 0; j < NR; j++) { ar[j] = 0.0; ai[j] = 0.0; }
            for (int c = 0; c < nb; c++) {
                const v2d l = Lr[(int64_t)c * t.lda];
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    ar[j] += l.x * yr[c][j] - l.y * yi[c][j];
                    ai[j] += l.x * yi[c][j] + l.y * yr[c][j];
                }
            }
#pragma unroll
            for (int j = 0; j < NR; j++)
                if (j < nj) {
                    double *q = X + 2 * (row + (int64_t)(j0 + j) * ldx);
                    atomic_add_f64(q, -ar[j]);
                    atomic_add_f64(q + 1, -ai[j]);
                }
        }
    }
}

// backward: x1[c] -= sum_r conj(L[r][col0+c]) * X[row(r)]
__global__ __launch_bounds__(256) void k_solve_bwd_update_c(
    const PotrfTask *__restrict__ tasks, const int32_t *__restrict__ tile_task, const double *__restrict__ Lx,
    const int32_t *__restrict__ Ls, double *__restrict__ X, int64_t ldx, int nrhs)
{
    constexpr int NR = 2;
    __shared__ double Tr[64][NB + 1], Ti[64][NB + 1];
    __shared__ double xr[64][NR], xi[64][NR];
    __shared__ double redr[4][NB][NR], redi[4][NB][NR];
    const PotrfTask t = tasks[tile_task[blockIdx.x]];
    const int tid = threadIdx.x, nb = t.nb;
    const int rbase = t.col0 + nb + (int)(blockIdx.x - t.stile0) * SOLVE_ROWS;
    const int c = tid & 63, part = tid >> 6;
    double *__restrict__ x1 = X + 2 * (int64_t)(t.k1 + t.col0);
    for (int j0 = 0; j0 < nrhs; j0 += NR) {
        const int nj = min(NR, nrhs - j0);
        double ar[NR], ai[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) { ar[j] = 0.0; ai[j] = 0.0; }
        for (int sub = 0; sub < SOLVE_ROWS / 64; sub++) {
            const int r0 = rbase + sub * 64;
            if (r0 >= t.lda) break;
            __syncthreads();
            {
                const int rl = tid & 63, r = r0 + rl;
                const bool ok = r < t.lda;
                const v2d *__restrict__ Lrow = (const v2d *)Lx + t.a_off + (r - t.col0);
                for (int cc = (tid >> 6); cc < nb; cc += 4) {
                    v2d v = {0.0, 0.0};
                    if (ok) v = Lrow[(int64_t)cc * t.lda];
                    Tr[rl][cc] = v.x; Ti[rl][cc] = v.y;
                }
                if (tid < 64) {
                    int64_t row = 0;
                    if (ok) row = (r < t.nscol) ? (int64_t)t.k1 + r : (int64_t)Ls[t.ls_off + r];
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        v2d v = {0.0, 0.0};
                        if (ok && j < nj) v = ((const v2d *)X)[row + (int64_t)(j0 + j) * ldx];
                        xr[rl][j] = v.x; xi[rl][j] = v.y;
                    }
                }
            }
            __syncthreads();
            if (c < nb) {
                for (int rr = part * 16; rr < part * 16 + 16; rr++) {
                    const double lr = Tr[rr][c], li = -Ti[rr][c];       // conj
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        ar[j] += lr * xr[rr][j] - li * xi[rr][j];
                        ai[j] += lr * xi[rr][j] + li * xr[rr][j];
                    }
                }
            }
        }
        __syncthreads();
        if (c < nb) {
#pragma unroll
            for (int j = 0; j < NR; j++) { redr[part][c][j] = ar[j]; redi[part][c][j] = ai[j]; }
        }
        __syncthreads();
        if (part == 0 && c < nb) {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                if (j < nj) {
                    const double vr = redr[0][c][j] + redr[1][c][j] + redr[2][c][j] + redr[3][c][j];
                    const double vi = redi[0][c][j] + redi[1][c][j] + redi[2][c][j] + redi[3][c][j];
                    double *q = x1 + 2 * (c + (int64_t)(j0 + j) * ldx);
                    atomic_add_f64(q, -vr);
                    atomic_add_f64(q + 1, -vi);
                }
            }
        }
    }
}

}  // namespace mi355
