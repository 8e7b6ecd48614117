// kernels.hip.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the supernodal
// numeric engine.  Real double precision (E = 1).  Included by engine.hip only.
//
// Kernel            replaces (reference)                                     bound
// k_assemble        A-scatter + beta, worker :438-520                        HBM
// k_assemble_aat    A*F accumulation for stype == 0, worker :471-506          HBM
// k_relmap          Map/RelativeMap, worker :392-399,:896-903 and            HBM (once per symbolic)
//                   kernelCreateMap/kernelCreateRelativeMap (CHOLMOD/GPU/cholmod_gpu_kernels.cu:21-41)
// k_gemm64          dsyrk+dgemm+assembly (worker :769-930; cublasDsyrk/Dgemm +  FP64 MFMA
//                   kernelAddUpdate, CHOLMOD/GPU/t_cholmod_gpu.c:502-582),
//                   trailing updates and TRSM-by-inverse inside dpotrf/dtrsm
//                   (worker :1023,:1175; t_cholmod_gpu.c:784-1135,1151-1393)
// k_potrf64         dpotrf of one <=64 diagonal block + its inverse            latency
// k_trsm64          dtrsm of the rows below a panel (worker :1175) by that inverse    latency
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "schedule.hpp"

namespace mi355 {

typedef double v4d __attribute__((ext_vector_type(4)));

// The kernels from here to k_gemm64 use no wave intrinsic, no LDS and no barrier.
// hardware FP64 atomic add (global_atomic_add_f64); no CAS loop
__device__ __forceinline__ void atomic_add_f64(double *p, double v)
{
    unsafeAtomicAdd(p, v);
}

// position of `key` in the sorted int32 list a[0..len); -1 if absent
__device__ __forceinline__ int find_row(const int32_t *__restrict__ a, int len, int key)
{
    int lo = 0, hi = len - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        int v = a[mid];
        if (v == key) return mid;
        if (v < key) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

//------------------------------------------------------------------------------
// k_assemble: Lx <- A (lower part) + beta*I, one thread per column
//------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assemble(
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
            if (imap >= 0) Lx[colbase + imap] = Ax[p];
        }
    }
    if (beta != 0.0) Lx[colbase + (k - k1)] += beta;
}

//------------------------------------------------------------------------------
// k_assemble_perm: Lx <- tril(P A P') straight from A (stype < 0: its lower part) and the inverse permutation,
// one thread per column of A -- the permuted copy S that cholmod_factorize builds with two ptransposes
// (CHOLMOD/Cholesky/cholmod_factorize.c:225-236) never exists.  Entry A(i,j), i >= j, lands at
// S(max(ki,kj), min(ki,kj)), ki = iperm[i], conjugated if the two swap; entries outside the pattern of L are
// dropped (worker :462-467).  Only the columns kc < climit of S are assembled (the replay of a failed factorization
// stops at the failing supernode): the limit is on the TARGET column, every column of A is walked.  k_add_beta then
// adds beta to the diagonal (worker :510-520).
//------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void k_assemble_perm(
    int64_t ncols, int64_t climit, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
    const int64_t *__restrict__ Anz, const double *__restrict__ Ax, const int64_t *__restrict__ iperm,
    const int32_t *__restrict__ supermap, const int64_t *__restrict__ Super,
    const int64_t *__restrict__ Lpi, const int64_t *__restrict__ Lpx,
    const int32_t *__restrict__ Ls, double *__restrict__ Lx, const int8_t *__restrict__ smask)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    const int64_t kj = iperm[j];
    int64_t p = Ap[j];
    const int64_t pend = Anz ? p + Anz[j] : Ap[j + 1];
    for (; p < pend; p++) {
        const int64_t i = Ai[p];
        if (i < j) continue;
        const int64_t ki = iperm[i];
        const bool swap = ki < kj;
        const int64_t kc = swap ? ki : kj, kr = swap ? kj : ki;
        if (kc >= climit) continue;
        const int s = supermap[kc];
        if (smask && !smask[s]) continue;
        const int64_t k1 = Super[s], psi = Lpi[s];
        const int nsrow = (int)(Lpi[s + 1] - psi);
        const int imap = find_row(Ls + psi, nsrow, (int)kr);
        if (imap < 0) continue;
        const int64_t at = Lpx[s] + (kc - k1) * (int64_t)nsrow + imap;
        if (E == 2) {
            Lx[2 * at] = Ax[2 * p];
            Lx[2 * at + 1] = swap ? -Ax[2 * p + 1] : Ax[2 * p + 1];
        } else {
            Lx[at] = Ax[p];
        }
    }
}

//------------------------------------------------------------------------------
// k_assemble_aat: Lx <- tril(A*F) for stype == 0 (worker :471-506), one thread per column k of L:
// column k = sum over the entries F(j,k) of A(:,j)*F(j,k), rows i >= k, accumulated in the reference's order
// (pf ascending, then p ascending) by the one thread that owns the column -- no atomics, the same sum as the
// reference's.  A is n x ncol, F = A' or A(:,f)' is ncol x n; entries outside the pattern of L are dropped
// (:495-497).  Complex: L_MULTADD is a plain complex product (F already holds the conjugates).  k_add_beta follows.
//------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(256) void k_assemble_aat(
    int64_t ncols, const int64_t *__restrict__ Ap, const int64_t *__restrict__ Ai,
    const int64_t *__restrict__ Anz, const double *__restrict__ Ax,
    const int64_t *__restrict__ Fp, const int64_t *__restrict__ Fi,
    const int64_t *__restrict__ Fnz, const double *__restrict__ Fx,
    const int32_t *__restrict__ supermap, const int64_t *__restrict__ Super,
    const int64_t *__restrict__ Lpi, const int64_t *__restrict__ Lpx,
    const int32_t *__restrict__ Ls, double *__restrict__ Lx, const int8_t *__restrict__ smask)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncols) return;
    const int s = supermap[k];
    if (smask && !smask[s]) return;
    const int64_t k1 = Super[s], psi = Lpi[s];
    const int nsrow = (int)(Lpi[s + 1] - psi);
    const int64_t colbase = Lpx[s] + (k - k1) * (int64_t)nsrow;
    int64_t pf = Fp[k];
    const int64_t pfend = Fnz ? pf + Fnz[k] : Fp[k + 1];
    for (; pf < pfend; pf++) {
        const int64_t j = Fi[pf];
        const double fr = Fx[E * pf], fi = (E == 2) ? Fx[E * pf + E - 1] : 0.0;
        int64_t p = Ap[j];
        const int64_t pend = Anz ? p + Anz[j] : Ap[j + 1];
        for (; p < pend; p++) {
            const int64_t i = Ai[p];
            if (i < k) continue;
            const int imap = find_row(Ls + psi, nsrow, (int)i);
            if (imap < 0) continue;
            const int64_t at = colbase + imap;
            if (E == 2) {
                const double ar = Ax[2 * p], ai = Ax[2 * p + 1];
                Lx[2 * at] += ar * fr - ai * fi;
                Lx[2 * at + 1] += ai * fr + ar * fi;
            } else {
                Lx[at] += Ax[p] * fr;
            }
        }
    }
}

template <int E>
__global__ __launch_bounds__(256) void k_add_beta(
    int64_t ncols, double beta, const int32_t *__restrict__ supermap, const int64_t *__restrict__ Super,
    const int64_t *__restrict__ Lpi, const int64_t *__restrict__ Lpx, double *__restrict__ Lx,
    const int8_t *__restrict__ smask)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncols) return;
    const int s = supermap[k];
    if (smask && !smask[s]) return;
    const int64_t k1 = Super[s], nsrow = Lpi[s + 1] - Lpi[s];
    Lx[E * (Lpx[s] + (k - k1) * nsrow + (k - k1))] += beta;
}

//------------------------------------------------------------------------------
// k_relmap: relmap[rm_off + i] = position of Ls[ls_src + i] in the row list of s
//------------------------------------------------------------------------------
struct RelmapPair { int64_t ls_src, ls_dst, rm_off; int32_t nsrow, pad; };

__global__ __launch_bounds__(256) void k_relmap(
    int64_t total, int64_t npairs, const RelmapPair *__restrict__ pairs,
    const int32_t *__restrict__ Ls, int32_t *__restrict__ relmap, int *__restrict__ err)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int64_t lo = 0, hi = npairs - 1;
    while (lo < hi) {
        int64_t mid = (lo + hi + 1) >> 1;
        if (pairs[mid].rm_off <= e) lo = mid; else hi = mid - 1;
    }
    const RelmapPair pr = pairs[lo];
    const int row = Ls[pr.ls_src + (e - pr.rm_off)];
    const int pos = find_row(Ls + pr.ls_dst, pr.nsrow, row);
    if (pos < 0) { *err = 1; relmap[e] = 0; } else relmap[e] = pos;
}


//------------------------------------------------------------------------------
// k_gemm_tiny: the update tasks of a STEP_TINY launch (schedule.hpp), one WAVE per task, TINY_WAVES tasks per
// workgroup: lane l computes entries l, l + 64, ... of the M x N result (column-major, M, N <= 64) as plain dot
// products over K straight from global memory and applies them like k_gemm64's epilogue (entries on or below the
// diagonal only; atomic scatter-subtract through the relative map, or direct subtract).  No LDS, no barrier, no
// MFMA: for a task of a few dozen entries the 64 x 64 MFMA tile of k_gemm64 is all padding, and what bounds a
// launch of several hundred thousand of them is how many tasks a CU holds in flight.
//------------------------------------------------------------------------------
constexpr int TINY_WAVES = 4;

__global__ __launch_bounds__(64 * TINY_WAVES) void k_gemm_tiny(
    const GemmTask *__restrict__ tasks, const Schedule::TileEntry *__restrict__ tile_tab, int64_t ntiles,
    double *__restrict__ Lx, const int32_t *__restrict__ relmap)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * TINY_WAVES + wave;
    if (e >= ntiles) return;                                // no barrier below: a wave may leave early
    const GemmTask t = tasks[tile_tab[e].task];
    const int M = t.M, N = t.N, K = t.K;
    const double *__restrict__ A = Lx + t.a_off;
    const double *__restrict__ B = Lx + t.b_off;
    double *__restrict__ C = Lx + t.c_off;
    const int64_t lda = t.lda, ldb = t.ldb, ldc = t.ldc;
    const bool scatter = t.mode == GEMM_SCATTER_SUB;
    const int32_t *__restrict__ rm = relmap + (scatter ? t.rm_off : 0);
    for (int idx = lane; idx < M * N; idx += 64) {
        const int n = idx / M, m = idx - n * M;
        if (m < n) continue;
        double acc = 0.0;
#pragma unroll 4
        for (int k = 0; k < K; k++) acc += A[m + (int64_t)k * lda] * B[n + (int64_t)k * ldb];
        if (scatter) atomic_add_f64(C + rm[m] + (int64_t)rm[n] * ldc, -acc);
        else C[m + (int64_t)n * ldc] -= acc;
    }
}

//------------------------------------------------------------------------------
// k_gemm64: batched C = A * B^T on v_mfma_f64_16x16x4_f64, 64x64 tile per workgroup
// (4 waves as 2x2, each wave 32x32 = 2x2 MFMA blocks), K staged through LDS in steps
// of 16, epilogue by mode (atomic scatter-subtract / direct subtract / overwrite).
//
// MFMA operand roles are swapped on purpose: the B-matrix rows feed the MFMA "A"
// operand and the A-matrix rows the MFMA "B" operand, so that in the result the LANE
// index (lane & 15) runs along m -- the direction that is contiguous in Lx -- and each
// store/atomic wave-instruction touches 4 runs of 16 consecutive doubles.
//------------------------------------------------------------------------------
constexpr int G64_KS = 16;          // K per LDS stage
constexpr int G64_LD = 80;          // padded row stride (doubles): 640 B, lanes 16..31 land on banks 32..63

__global__ __launch_bounds__(256) void k_gemm64(
    const GemmTask *__restrict__ tasks, const Schedule::TileEntry *__restrict__ tile_tab, double *__restrict__ Lx,
    const double *__restrict__ Dinv, const int32_t *__restrict__ relmap)
{
    __shared__ double As[G64_KS][G64_LD];
    __shared__ double Bs[G64_KS][G64_LD];
    __shared__ int rmM[TILE], rmN[TILE];

    const int tid = threadIdx.x;
    const int64_t bid = blockIdx.x;
    const Schedule::TileEntry te = tile_tab[bid];
    const GemmTask t = tasks[te.task];
    const int M = t.M, N = t.N, K = t.K, mode = t.mode;
    const int tm = (int)(te.tmtn & 0xffffu), tn = (int)(te.tmtn >> 16);
    const int m0 = tm * TILE, n0 = tn * TILE;

    const double *__restrict__ A = Lx + t.a_off;
    const double *__restrict__ B = (mode == GEMM_OVERWRITE ? Dinv : Lx) + t.b_off;
    const int64_t lda = t.lda, ldb = t.ldb;

    if (mode == GEMM_SCATTER_SUB) {
        if (tid < TILE) rmM[tid] = (m0 + tid < M) ? relmap[t.rm_off + m0 + tid] : 0;
        else if (tid < 2 * TILE) { int j = tid - TILE; rmN[j] = (n0 + j < N) ? relmap[t.rm_off + n0 + j] : 0; }
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    v4d acc[2][2];                               // [ni][mi]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    const int lrow = tid & 63, lk = tid >> 6;
    const bool arow_ok = (m0 + lrow) < M, brow_ok = (n0 + lrow) < N;
    const double *Ap = A + (m0 + lrow);
    const double *Bp = B + (n0 + lrow);

    for (int k0 = 0; k0 < K; k0 += G64_KS) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = lk + 4 * q, kk = k0 + k;
            const bool kok = kk < K;
            As[k][lrow] = (arow_ok && kok) ? Ap[(int64_t)kk * lda] : 0.0;
            Bs[k][lrow] = (brow_ok && kok) ? Bp[(int64_t)kk * ldb] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int kk = ks * 4 + l4;
            const double a0 = As[kk][wm * 32 + l15];
            const double a1 = As[kk][wm * 32 + 16 + l15];
            const double b0 = Bs[kk][wn * 32 + l15];
            const double b1 = Bs[kk][wn * 32 + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue.  D row = (lane>>4) + 4*r  <-> n ; D col = lane & 15 <-> m
    const int64_t ldc = t.ldc;
    double *__restrict__ C = Lx + t.c_off;
    if (mode == GEMM_DIRECT_SUB) {
        // batched read-modify-write: 16 loads in flight before the first use
        double cv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int m = m0 + wm * 32 + ((q >> 2) & 1) * 16 + l15;
            const int n = n0 + wn * 32 + (q >> 3) * 16 + l4 + 4 * (q & 3);
            const bool ok = (m < M) && (n < N) && (m >= n);
            cv[q] = C[ok ? (m + (int64_t)n * ldc) : 0];
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int m = m0 + wm * 32 + ((q >> 2) & 1) * 16 + l15;
            const int n = n0 + wn * 32 + (q >> 3) * 16 + l4 + 4 * (q & 3);
            if ((m < M) && (n < N) && (m >= n)) C[m + (int64_t)n * ldc] = cv[q] - acc[q >> 3][(q >> 2) & 1][q & 3];
        }
        return;
    }
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
                    const double v = acc[ni][mi][r];
                    if (mode == GEMM_SCATTER_SUB) {
                        if (m >= n) atomic_add_f64(C + rmM[ml] + (int64_t)rmN[nl] * ldc, -v);
                    } else if (mode == GEMM_DIRECT_SUB) {
                        if (m >= n) C[m + (int64_t)n * ldc] -= v;
                    } else {
                        C[m + (int64_t)n * ldc] = v;
                    }
                }
            }
        }
    }
}

//------------------------------------------------------------------------------
// k_gemm128: the large-task variant.  128x128 tile per 256-thread workgroup, 4 waves as
// 2x2, each wave 64x64 = 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs), K staged 16
// at a time through a double-buffered LDS image.  Same operand-role swap and epilogues as
// k_gemm64.
//
// The FP64 MFMA pipe runs one 16x16x4 instruction per 64 cycles per SIMD
// (profiles/r01_mfma_f64_probe3.txt: 72-78 TFLOP/s chip-wide from registers), and the two
// workgroups of a CU fall into lock-step, so whatever a wave does OUTSIDE its MFMA stream
// is lost pipe time (profiles/r01_pmc_sq_k_gemm128_before.txt: pipe 69 % busy with the
// loads, LDS writes and barrier clustered between the MFMA runs).  The staging is therefore
// spread through the MFMA stream: one register set, the slice loaded during stage s
// (for stage s+2) is written to LDS during stage s+1 right before the register is
// re-loaded -- a full stage of flight time per load, two loads + two LDS writes behind
// every 8 MFMAs, and a loop body without branches (edge rows are clamped: their products
// only reach C entries that are never stored; the K tail is zeroed on the way into LDS).
//------------------------------------------------------------------------------
constexpr int G128_KS = 16;
constexpr int G128_LD = 144;        // 1152 B row stride: +32 banks per k row => conflict-free ds_read_b64

// CPLX: the same kernel on a complex (interleaved re, im) factor through its REAL view.  A complex column
// of C, A or B is a real column of twice the length (rows 2m = Re, 2m+1 = Im), so  C = A B^H  becomes the real
// product  C'' = A'' B''^T  with a doubled contraction:
//     C''(2m,   n) = sum_k Ar(m,k) Br(n,k) + Ai(m,k) Bi(n,k)
//     C''(2m+1, n) = sum_k Ai(m,k) Br(n,k) - Ar(m,k) Bi(n,k)
// i.e. A'' = [ A~ | pair-swapped A~ with a sign ], B'' = [ Re B | Im B ] (rows 2n / 2n+1 of the real view).
// Only the staging addresses and the index arithmetic of the epilogues change; a tile covers 128 real rows
// = 64 complex rows by 128 complex columns (the host builds the tile table for that shape, GemmTask::big = 2).
// The imaginary part of a diagonal entry of C is left alone (it is zero, as zherk keeps it).
template <bool FRESH>
__device__ __forceinline__ void potrf64_block(
    const PotrfTask &t, double *Lx, double *Dinv, int *info, double (*Ls)[NB + 1], double (*Vs)[NB + 1]);

// Fused POTRF (GemmTask::fuse_potrf, real factors): the trailing update of a panel step is the last writer of the next
// diagonal block, and that block lies in tile (0, 0) of the task.  The workgroup of that tile factorizes it (and builds its
// inverse) right after its own stores -- potrf64_block on the LDS of the main loop -- so the panel chain of a wide
// supernode is two dependent launches per 64 columns (TRSM, trailing update) instead of three, and the 35-45 us of the
// single-workgroup POTRF run beside the other tiles of the update instead of after them.  potrf / dinv_w / info: the
// schedule's POTRF tasks, the block inverses (written here; `Dinv` is the read-only view of the TRSM tasks), LAPACK's info.
#ifndef G128_RMW
#define G128_RMW 4          // 16-row blocks per read-modify-write batch of the DIRECT_SUB epilogue (1, 2 or 4; measured at 150^3 and
                            // elasticity 80^3: 4 is 0.5 - 0.7 % faster than 1 or 2 although its 64 values in flight spill 92 bytes)
#endif
template <bool CPLX>
__global__ __launch_bounds__(256, 2) void k_gemm128(
    const GemmTask *__restrict__ tasks, const Schedule::TileEntry *__restrict__ tile_tab, double *__restrict__ Lx,
    const double *__restrict__ Dinv, const int32_t *__restrict__ relmap,
    unsigned long long *__restrict__ stamps,       // debug builds of the micro-benchmark only: 4 clock stamps per tile
    const PotrfTask *__restrict__ potrf = nullptr, double *dinv_w = nullptr, int *info = nullptr)
{
    const unsigned long long st0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    __shared__ double As[2][G128_KS][G128_LD];
    __shared__ double Bs[2][G128_KS][G128_LD];
    __shared__ int rmM[BTILE], rmN[BTILE];

    const int tid = threadIdx.x;
    const int64_t bid = blockIdx.x;
    const Schedule::TileEntry te = tile_tab[bid];
    const GemmTask t = tasks[te.task];
    constexpr int E = CPLX ? 2 : 1;
    const int N = t.N, mode = t.mode;
    const int M = E * t.M;                           // rows of the real view of A and C
    const int Kc = t.K, K = E * t.K;                 // contraction length: complex entries / real view
    const int tm = (int)(te.tmtn & 0xffffu), tn = (int)(te.tmtn >> 16);
    const int m0 = tm * BTILE, n0 = tn * BTILE;

    const double *__restrict__ A = Lx + E * t.a_off;
    const double *__restrict__ B = (mode == GEMM_OVERWRITE ? Dinv : Lx) + E * t.b_off;
    const int64_t lda = E * (int64_t)t.lda, ldb = E * (int64_t)t.ldb;

    if (mode == GEMM_SCATTER_SUB) {
        if (tid < BTILE) {
            const int m = m0 + tid;
            rmM[tid] = (m < M) ? (CPLX ? 2 * relmap[t.rm_off + (m >> 1)] + (m & 1) : relmap[t.rm_off + m]) : 0;
        } else { int j = tid - BTILE; rmN[j] = (n0 + j < N) ? relmap[t.rm_off + n0 + j] : 0; }
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    // a diagonal tile of a lower-trapezoid task needs nothing from its strictly upper 64x64 quadrant
    const bool skip_mma = (mode != GEMM_OVERWRITE) && (((m0 + wm * 64 + 63) >> (E - 1)) < n0 + wn * 64);

    // acc[ni][mi][r] is C[m = 16*mi + lane%16][n = 16*ni + 4*r + lane/16] of the wave's 64x64 tile
    v4d acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

    // each thread stages 8 k-values (lk, lk+2, ..) of one A row and one B row; lk is wave-uniform
    const int lrow = tid & 127, lk = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int arow = min(m0 + lrow, M - 1), brow = min(n0 + lrow, N - 1);
    double ra[8], rb[8];
    // element kk (< K) of this thread's row of A'' and of B''; kk is wave-uniform
    auto load_a = [&](int kk) -> double {
        if constexpr (!CPLX) {
            return A[arow + (int64_t)kk * lda];
        } else {
            const bool second = kk >= Kc;       // the sign of the pair image is applied on the way into LDS (a_sign)
            return A[(second ? (arow ^ 1) : arow) + (int64_t)(second ? kk - Kc : kk) * lda];
        }
    };
    // value of A'' from the raw loaded one: the second half of the contraction is -Re in the Im rows
    auto a_sign = [&](double v, int kk) -> double {
        if constexpr (CPLX) return (kk >= Kc && (arow & 1)) ? -v : v;
        else return v;
    };
    auto load_b = [&](int kk) -> double {
        if constexpr (!CPLX) {
            return B[brow + (int64_t)kk * ldb];
        } else {
            const bool second = kk >= Kc;
            return B[2 * brow + (second ? 1 : 0) + (int64_t)(second ? kk - Kc : kk) * ldb];
        }
    };

    auto gload1 = [&](int q, int k0) {
        const int kk = min(k0 + lk + 2 * q, K - 1);
        ra[q] = load_a(kk);
        rb[q] = load_b(kk);
    };
    auto lstore1 = [&](int q, int k0, int buf) {
        const int kk = k0 + lk + 2 * q;
        const bool kok = kk < K;
        As[buf][lk + 2 * q][lrow] = kok ? a_sign(ra[q], kk) : 0.0;
        Bs[buf][lk + 2 * q][lrow] = kok ? rb[q] : 0.0;
    };

#pragma unroll
    for (int q = 0; q < 8; q++) gload1(q, 0);
#pragma unroll
    for (int q = 0; q < 8; q++) { lstore1(q, 0, 0); gload1(q, G128_KS); }
    __syncthreads();

    // operand registers, two sets: step ks computes from set ks&1 while set (ks+1)&1 is being read
    double fa[2][4], fb[2][4];
    auto fread = [&](int set, int buf, int ks, int h) {          // h = 0..3: one ds_read2_b64 each
        const int kk = ks * 4 + l4;
        if (h < 2) {
            fa[set][2 * h] = As[buf][kk][wm * 64 + (2 * h) * 16 + l15];
            fa[set][2 * h + 1] = As[buf][kk][wm * 64 + (2 * h + 1) * 16 + l15];
        } else {
            fb[set][2 * h - 4] = Bs[buf][kk][wn * 64 + (2 * h - 4) * 16 + l15];
            fb[set][2 * h - 3] = Bs[buf][kk][wn * 64 + (2 * h - 3) * 16 + l15];
        }
    };
    // staging piece g (0..3) of step ks: element q = 2ks + g/2, matrix A (g even) or B (g odd):
    // stage s+1's value goes to the other LDS buffer, then the register is re-loaded for stage s+2
    auto stage_write = [&](int ks, int g, int k0, int buf) {
        const int q = 2 * ks + (g >> 1);
        const int kk = k0 + G128_KS + lk + 2 * q;
        const bool kok = kk < K;
        if ((g & 1) == 0) As[buf ^ 1][lk + 2 * q][lrow] = kok ? a_sign(ra[q], kk) : 0.0;
        else Bs[buf ^ 1][lk + 2 * q][lrow] = kok ? rb[q] : 0.0;
    };
    auto stage_load = [&](int ks, int g, int k0) {
        const int q = 2 * ks + (g >> 1);
        const int kk = min(k0 + 2 * G128_KS + lk + 2 * q, K - 1);
        if ((g & 1) == 0) ra[q] = load_a(kk);
        else rb[q] = load_b(kk);
    };

    const unsigned long long st1 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    auto mainloop = [&](auto with_mma) {
        constexpr bool MMA = decltype(with_mma)::value;
        int buf = 0;
        if constexpr (MMA) {
#pragma unroll
            for (int h = 0; h < 4; h++) fread(0, 0, 0, h);
        }
        for (int k0 = 0; k0 < K; k0 += G128_KS) {
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const int set = ks & 1;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    // the hand-placed issue order: 2 MFMAs | select + LDS write | MFMA | address + global load |
                    // MFMA | operand read for the next step.  The last 4 MFMAs of a stage are issued AFTER the
                    // barrier, behind the first operand reads of the next stage, whose latency they cover.
                    const bool tail = (ks == 3 && g == 3);
                    if (MMA && !tail) {
                        acc[g][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[set][g], fa[set][0], acc[g][0], 0, 0, 0);
                        acc[g][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[set][g], fa[set][1], acc[g][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    stage_write(ks, g, k0, buf);
                    __builtin_amdgcn_sched_barrier(0);
                    if (MMA && !tail) acc[g][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[set][g], fa[set][2], acc[g][2], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    stage_load(ks, g, k0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (MMA && !tail) acc[g][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[set][g], fa[set][3], acc[g][3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (MMA && ks < 3) fread(set ^ 1, buf, ks + 1, g);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            if constexpr (MMA) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 4; h++) fread(0, buf ^ 1, 0, h);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < 4; mi++)
                    acc[3][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[1][3], fa[1][mi], acc[3][mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
    };
    // (uniform over the workgroup)
    const bool fused = !CPLX && mode == GEMM_DIRECT_SUB && t.fuse_potrf >= 0 && te.tmtn == 0u && potrf != nullptr;
    if (skip_mma) {
        mainloop(std::false_type{});
        if (!fused) return;
    } else {
        mainloop(std::true_type{});
    }
    const unsigned long long st2 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;

    const int64_t ldc = E * (int64_t)t.ldc;
    double *__restrict__ C = Lx + E * t.c_off;
    // on or below the diagonal (row index of the factor = m >> 1 in the real view of a complex one); the
    // imaginary part of a diagonal entry is not touched
    auto lower = [&](int m, int n) {
        if constexpr (CPLX) return (m >> 1) > n || ((m >> 1) == n && !(m & 1));
        else return m >= n;
    };
    if (mode == GEMM_DIRECT_SUB) {
        // read-modify-write in batches of 16: all loads of a batch are issued before the first use, so
        // the wave waits for memory once per batch instead of once per element
        // (G128_RMW row blocks of 16 per batch: the staging registers of the main loop are free by now, and every batch
        // is one memory round trip of the tile's epilogue)
#pragma unroll
        for (int mb = 0; mb < 4; mb += G128_RMW) {
            if (skip_mma) break;                 // (a wave that only came along for the fused POTRF below)
            double cv[G128_RMW][16];
#pragma unroll
            for (int u = 0; u < G128_RMW; u++) {
                const int m = m0 + wm * 64 + (mb + u) * 16 + l15;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int n = n0 + wn * 64 + (q >> 2) * 16 + l4 + 4 * (q & 3);
                    const bool ok = (m < M) && (n < N) && lower(m, n);
                    cv[u][q] = C[ok ? (m + (int64_t)n * ldc) : 0];
                }
            }
#pragma unroll
            for (int u = 0; u < G128_RMW; u++) {
                const int m = m0 + wm * 64 + (mb + u) * 16 + l15;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int n = n0 + wn * 64 + (q >> 2) * 16 + l4 + 4 * (q & 3);
                    if ((m < M) && (n < N) && lower(m, n)) C[m + (int64_t)n * ldc] = cv[u][q] - acc[q >> 2][mb + u][q & 3];
                }
            }
        }
        if (stamps && tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            stamps[4 * bid] = st0; stamps[4 * bid + 1] = st1; stamps[4 * bid + 2] = st2;
            stamps[4 * bid + 3] = __builtin_amdgcn_s_memtime();
        }
        if constexpr (!CPLX) {
            if (fused) {
                // the stores above are complete and visible to this workgroup's loads before anybody reads the block
                __threadfence();
                __syncthreads();
                static_assert(sizeof(As) >= sizeof(double) * NB * (NB + 1) && sizeof(Bs) >= sizeof(double) * NB * (NB + 1),
                              "the POTRF images reuse the staging buffers of the main loop");
                potrf64_block<true>(potrf[t.fuse_potrf], Lx, dinv_w, info,
                                    reinterpret_cast<double (*)[NB + 1]>(&As[0][0][0]),
                                    reinterpret_cast<double (*)[NB + 1]>(&Bs[0][0][0]));
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ni++) {
#pragma unroll
        for (int mi = 0; mi < 4; mi++) {
            const int ml = wm * 64 + mi * 16 + l15;
            const int m = m0 + ml;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int nl = wn * 64 + ni * 16 + l4 + 4 * r;        // D row (n) = lane/16 + 4r
                const int n = n0 + nl;
                if (m < M && n < N) {
                    const double v = acc[ni][mi][r];
                    if (mode == GEMM_SCATTER_SUB) {
                        if (lower(m, n)) atomic_add_f64(C + rmM[ml] + (int64_t)rmN[nl] * ldc, -v);
                    } else {
                        C[m + (int64_t)n * ldc] = v;
                    }
                }
            }
        }
    }
}

//------------------------------------------------------------------------------
// k_trsm64: X = A * inv(L11)^T for the rows below a 64-column panel (the TRSM of a3.10 in the form of a
// product with the block inverse kept by k_potrf64), in place.  Every panel step of the factorization waits
// for this launch, so it is built for latency like k_potrf64: one workgroup per 64-row tile of the
// tile table (tn = 0), the inverse staged in LDS once, the rows of A read straight from global memory in MFMA
// operand layout (16 consecutive rows per k: 128-byte runs), no barrier after the staging, and the result
// stored with the rows along lane%16 (runs of 16 consecutive doubles per column).
//------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trsm64(
    const GemmTask *__restrict__ tasks, const Schedule::TileEntry *__restrict__ tile_tab, double *__restrict__ Lx,
    const double *__restrict__ Dinv)
{
    __shared__ double Ds[NB][NB + 2];                    // Ds[n][k] = inv(L11)(n, k)
    const Schedule::TileEntry te = tile_tab[blockIdx.x];
    const GemmTask t = tasks[te.task];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int nb = t.N;                                  // = K: order of the diagonal block
    const int m0 = (int)(te.tmtn & 0xffffu) * TILE;
    const double *__restrict__ D = Dinv + t.b_off;       // column-major, ld = ldb
    // rows of A for this wave: 16 rows, operand lane = row, k = column
    const int r = m0 + 16 * w + l15;
    const bool ok = r < t.M;
    double *__restrict__ Ar = Lx + t.a_off + (ok ? r : 0);
    const int nq = (nb + 3) >> 2, ncb = (nb + 15) >> 4;
    double b[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int k = 4 * q + l4;
        b[q] = (q < nq) ? Ar[(int64_t)(k < nb ? k : 0) * t.lda] : 0.0;     // k >= nb meets a zero of Ds
    }
    const int np = ncb * 16, kp = nq * 4;
    for (int idx = tid; idx < np * kp; idx += 256) {
        const int n = idx % np, k = idx / np;
        Ds[n][k] = (n < nb && k < nb) ? D[n + (int64_t)k * t.ldb] : 0.0;
    }
    __syncthreads();
    v4d acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; cb++) acc[cb] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 16; q++) {
        if (q >= nq) break;                              // uniform
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            if (cb >= ncb) break;
            acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ds[16 * cb + l15][4 * q + l4], b[q], acc[cb], 0, 0, 0);
        }
    }
    // result: column 16 cb + lane/16 + 4g, row = lane%16 of this wave's 16 rows
    if (ok) {
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = 16 * cb + l4 + 4 * g;
                if (n < nb) Ar[(int64_t)n * t.lda] = acc[cb][g];
            }
        }
    }
}

//------------------------------------------------------------------------------
// k_potrf64: lower Cholesky of one nb x nb (nb <= 64) diagonal block, in LDS, plus
// its inverse (used as the TRSM operand and by the solves).  One workgroup per task.
// info[s] receives the 1-based column (inside s) of the first pivot that is <= 0 or
// NaN -- LAPACK dpotrf's INFO, worker :1023-1027,:1084-1095.
//
// The block is on the critical path of every panel step, so the kernel is built for
// latency: right-looking over four 16-column blocks; the 16x16 diagonal block and its
// inverse are computed by ONE wave entirely in registers (lane = row, pivots and
// multipliers move between lanes with v_readlane -- no LDS traffic, no barrier inside
// the 16 column steps, one reciprocal square root per column instead of a square root
// and a division), the 16-wide TRSM and the trailing update are 16x16x4 FP64 MFMA
// products on LDS operands (3 barriers per 16 columns instead of 3 per column), and the
// 64x64 inverse is assembled from the 16x16 ones by MFMA block products.
//------------------------------------------------------------------------------
__device__ __forceinline__ double lane_bcast(double v, int srclane)      // srclane: compile-time constant
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// The work of one workgroup (256 threads) on one block; Ls / Vs: two LDS images of NB x (NB + 1) doubles.
// FRESH: the block was written by THIS workgroup a moment ago (k_gemm128's fused call: its read-modify-write stores,
// made visible by a fence and a barrier): the loads bypass the vector L1, whose lines may predate those stores.
__device__ __forceinline__ void potrf64_core(
    const PotrfTask &t, double *Lx, double *Dinv, int *info, double (*Ls)[NB + 1], double (*Vs)[NB + 1]);

template <bool FRESH>
__device__ __forceinline__ void potrf64_block(
    const PotrfTask &t, double *Lx, double *Dinv, int *info, double (*Ls)[NB + 1], double (*Vs)[NB + 1])
{
    const int tid = threadIdx.x;
    const int nb = t.nb;
    const int64_t lda = t.lda;
    double *A = Lx + t.a_off;

    const int nblk = (nb + 15) >> 4;                          // 16-column blocks that hold anything
    const int np = nblk * 16;                                 // padded order: identity beyond nb
    for (int idx = tid; idx < np * np; idx += 256) {
        const int i = idx % np, j = idx / np;
        double v = (i == j) ? 1.0 : 0.0;                     // identity padding
        if (i < nb && j < nb) {
            if (i >= j) v = FRESH ? __hip_atomic_load(A + i + (int64_t)j * lda, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : A[i + (int64_t)j * lda];
            else v = 0.0;
        }
        Ls[i][j] = v;
        Vs[i][j] = 0.0;
    }
    __syncthreads();
    potrf64_core(t, Lx, Dinv, info, Ls, Vs);
}

// ... and the factorization proper: Ls holds the block (row i, column j; identity beyond nb up to the next multiple of 16),
// Vs zeros; on return Ls = L, Vs = inv(L), both also written to Lx / Dinv.
__device__ __forceinline__ void potrf64_core(
    const PotrfTask &t, double *Lx, double *Dinv, int *info, double (*Ls)[NB + 1], double (*Vs)[NB + 1])
{
    const int tid = threadIdx.x;
    const int nb = t.nb;
    const int64_t lda = t.lda;
    double *A = Lx + t.a_off;
    const int nblk = (nb + 15) >> 4;
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    int fail_col = 0;                                         // wave 0 only: 1-based column of the first bad pivot
#pragma unroll 1
    for (int jb = 0; jb < nblk; jb++) {
        const int c0 = 16 * jb;
        if (wave == 0) {
            // lane r (mod 16) holds row r of the diagonal block; entries right of the diagonal are never
            // consumed (they collect harmless junk) and are written back as zeros
            double a[16], rs[16], w[16];
#pragma unroll
            for (int k = 0; k < 16; k++) a[k] = Ls[c0 + l15][c0 + k];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const double d = lane_bcast(a[j], j);
                fail_col = (!(d > 0.0) && fail_col == 0) ? c0 + j + 1 : fail_col;
                const double r = rsqrt(d);
                rs[j] = r;
                a[j] = (l15 == j) ? d * r : a[j] * r;
#pragma unroll
                for (int k = j + 1; k < 16; k++) a[k] -= a[j] * lane_bcast(a[j], k);
                __builtin_amdgcn_sched_barrier(0);            // keeps the broadcast values (SGPRs) short-lived
            }
            // W = inv(L_jj): lane c holds column c, w[i] = (delta_ic - sum_{k<i} L[i][k] w[k]) / L[i][i]
#pragma unroll
            for (int i = 0; i < 16; i++) {
                double acc = (l15 == i) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; k++) acc -= lane_bcast(a[k], i) * w[k];
                w[i] = acc * rs[i];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    Ls[c0 + lane][c0 + k] = (k <= lane) ? a[k] : 0.0;
                    Vs[c0 + k][c0 + lane] = w[k];
                }
            }
        }
        __syncthreads();
        // TRSM: X_ib = A(ib,jb) * W^T for the block rows below (one per wave)
        {
            const int ib = jb + 1 + wave;
            if (ib < nblk) {
                v4d x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    x = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[16 * ib + l15][c0 + 4 * q + l4], Vs[c0 + l15][c0 + 4 * q + l4], x, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) Ls[16 * ib + l4 + 4 * r][c0 + l15] = x[r];
            }
        }
        __syncthreads();
        // trailing update: A(ib,kb) -= X_ib X_kb^T for jb < kb <= ib < 4, block pairs dealt to the waves
        {
            const int nrem = nblk - 1 - jb, npairs = nrem * (nrem + 1) / 2;
            for (int p = wave; p < npairs; p += 4) {
                int ib = 0, q = p;
                while (q > ib) { q -= ib + 1; ib++; }          // p -> (ib, q): q <= ib
                const int bi = jb + 1 + ib, bk = jb + 1 + q;
                v4d c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int z = 0; z < 4; z++)
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[16 * bi + l15][c0 + 4 * z + l4], Ls[16 * bk + l15][c0 + 4 * z + l4], c, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) Ls[16 * bi + l4 + 4 * r][16 * bk + l15] -= c[r];
            }
        }
        __syncthreads();
    }

    if (tid == 0 && fail_col != 0 && info[t.s] == 0) info[t.s] = t.col0 + fail_col;
    for (int idx = tid; idx < nb * nb; idx += 256) {
        const int i = idx % nb, j = idx / nb;
        if (i >= j) A[i + (int64_t)j * lda] = Ls[i][j];
    }

    // off-diagonal blocks of the inverse, by distance d from the diagonal:
    // V(i,j) = -W_i * sum_{k=j}^{i-1} L(i,k) V(k,j); the sum's MFMA result registers are the B operand of W_i * (.)
    for (int d = 1; d < nblk; d++) {
        const int j = wave, i = wave + d;
        if (i < nblk) {
            v4d tsum = {0.0, 0.0, 0.0, 0.0};
            for (int k = j; k < i; k++) {
#pragma unroll
                for (int z = 0; z < 4; z++)
                    tsum = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[16 * i + l15][16 * k + 4 * z + l4], Vs[16 * k + 4 * z + l4][16 * j + l15], tsum, 0, 0, 0);
            }
            v4d v = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int z = 0; z < 4; z++)
                v = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[16 * i + l15][16 * i + 4 * z + l4], tsum[z], v, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) Vs[16 * i + l4 + 4 * r][16 * j + l15] = -v[r];
        }
        __syncthreads();
    }
    if (t.dinv_off >= 0) {
        double *D = Dinv + t.dinv_off;
        for (int idx = tid; idx < nb * nb; idx += 256) {
            const int i = idx % nb, j = idx / nb;
            D[idx] = (i >= j) ? Vs[i][j] : 0.0;               // column-major, ld = nb
        }
    }
}

__global__ __launch_bounds__(256) void k_potrf64(
    const PotrfTask *__restrict__ tasks, double *__restrict__ Lx,
    double *__restrict__ Dinv, int *__restrict__ info, const int32_t *__restrict__ order = nullptr)
{
    __shared__ double Ls[NB][NB + 1];       // the block, factorized in place (row i, column j)
    __shared__ double Vs[NB][NB + 1];       // its inverse
    const PotrfTask t = tasks[order ? order[blockIdx.x] : (int)blockIdx.x];
    potrf64_block<false>(t, Lx, Dinv, info, Ls, Vs);
}

//------------------------------------------------------------------------------
// k_panel_diag (STEP_PDIAG): the diagonal window of a panel group -- ncols x ncols, 64 < ncols <= 256, lower -- factorized
// by ONE workgroup, left-looking over its 64-column blocks.  It is a pure latency chain (277 of them in a row make the
// panel chain of the 150^3 root), so every value stays where the next step needs it:
//  * a wave owns up to four 16-row strips of block column b (rows from the diagonal block down) and holds them in
//    registers in MFMA RESULT layout -- which is also the operand layout of the next product -- from their load to their
//    last use;
//  * the updates by the block columns before b (T -= L(:, k) L(b, k)^T) take their 64 x 64 operand block through LDS,
//    one block per barrier pair, and accumulate straight into those registers;
//  * the diagonal block goes from the registers into the LDS image that potrf64_core factorizes (no trip through
//    memory), and the strips below it are multiplied by the inverse that the factorization leaves in LDS (X = T W^T),
//    again from the registers, and stored once.
// Four launches of k_potrf64, four of k_trsm64 and three trailing updates on these rows become one launch; the rows BELOW
// the window follow in k_panel_trsm_reg.  What one phase stores the next one loads after a barrier (one workgroup = one
// CU = one vector L1, written through).
//------------------------------------------------------------------------------
// (MI355_PDIAG_STAMP: defined by tools/probes/panel_diag_probe.hip only, to record clock stamps per phase)
#ifndef MI355_PDIAG_STAMP
#define MI355_PDIAG_STAMP(i)
#endif
__global__ __launch_bounds__(256) void k_panel_diag(
    const PanelGroup *__restrict__ groups, double *Lx, double *Dinv, int *info)
{
    // one LDS image of up to 256 rows x 64 columns: during the updates the earlier block columns' rows from the diagonal
    // block down (all of them at once: (4 - b) x 64 rows of b block columns are 192, 256, 192 rows for b = 1, 2, 3); for the
    // factorization its first 64 rows are the block (Ls) and the next 64 its inverse (Vs)
    __shared__ double P[4 * NB][NB + 1];
    double (*Ls)[NB + 1] = P;
    double (*Vs)[NB + 1] = P + NB;
    const PanelGroup g = groups[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int ncols = g.ncols, nblk = (ncols + NB - 1) / NB;
    const int64_t lda = g.lda;
    double *A = Lx + g.a_off;                                   // A[i + j * lda]: row i, column j of the window
    const uint32_t l4off = (uint32_t)l4 * (uint32_t)lda;        // (lda < 2^27 for a panel group: 32-bit lane offsets)
    // Every batch of loads below is UNCONDITIONAL (addresses clamped into the window, unwanted values dropped afterwards):
    // a load inside a branch is waited for at the end of that branch, and the phases of this kernel are nothing but
    // memory round trips (about 3 us each, measured: tools/probes/panel_diag_probe.hip) -- so each phase issues all its
    // loads at once and pays one.
#pragma unroll 1
    for (int b = 0; b < nblk; b++) {
        MI355_PDIAG_STAMP(6 * b + 0);
        const int c0 = NB * b, nbb = min(NB, ncols - c0);
        const int nrow = ncols - c0;                            // rows from the diagonal block down
        const int nrow64 = NB * (nblk - b);                     // ... in whole blocks
        const int nstrips = (nrow + 15) >> 4;                   // 16-row strips; 0 .. 3 are the diagonal block
        // this wave's strips: w (diagonal block), w + 4, w + 8, w + 12; lane % 16 = row of the strip
        uint32_t roff[4];
        bool live[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            live[i] = w + 4 * i < nstrips;                      // uniform over the wave
            roff[i] = (uint32_t)min(c0 + 16 * (w + 4 * i) + l15, ncols - 1) + l4off;
        }
        // ---- T in result layout: t[i][cb][gq] = entry (row of strip i, column c0 + 16 cb + lane / 16 + 4 gq)
        // (block column 0 has no updates: only its diagonal block is needed here, the strips below are loaded as operands later)
        v4d t[4][4];
        if (nbb == NB) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int cb = 0; cb < 4; cb++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++)
                        t[i][cb][gq] = (A + (int64_t)(c0 + 16 * cb + 4 * gq) * lda)[roff[(i == 0 || b > 0) ? i : 0]];
        } else {
            // the last, narrower block of a window (its diagonal block is all there is): no column beyond it is touched
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int cb = 0; cb < 4; cb++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const int n = 16 * cb + 4 * gq + l4;
                        const double v = A[(roff[0] - l4off) + (int64_t)(c0 + min(n, nbb - 1)) * lda];
                        t[i][cb][gq] = n < nbb ? v : 0.0;
                    }
        }
        // ---- the earlier block columns, rows c0 .. : chunk kc (columns 64 kc ..) -> P[kc * nrow64 + row][k]
        // (thread: P row 64 j + tid % 64, k = tid / 64 + 4 i)
        if (b > 0) {
            double sreg[4][16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int pr0 = NB * j, kc = min(pr0 / nrow64, b - 1), row = (pr0 % nrow64) + (tid & 63);
                const uint32_t voff = (uint32_t)min(row, nrow - 1) + (uint32_t)(tid >> 6) * (uint32_t)lda;
#pragma unroll
                for (int i = 0; i < 16; i++) sreg[j][i] = (A + c0 + (int64_t)(NB * kc + 4 * i) * lda)[voff];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool in = NB * j < b * nrow64 && ((NB * j) % nrow64) + (tid & 63) < nrow;
#pragma unroll
                for (int i = 0; i < 16; i++) P[(tid & 63) + NB * j][(tid >> 6) + 4 * i] = in ? sreg[j][i] : 0.0;
            }
        }
        MI355_PDIAG_STAMP(6 * b + 1);
        if (b > 0) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (live[i]) {
#pragma unroll 1
                    for (int kc = 0; kc < b; kc++) {
                        const double (*Pk)[NB + 1] = P + kc * nrow64;
                        double a[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) a[q] = -Pk[16 * (w + 4 * i) + l15][4 * q + l4];
#pragma unroll
                        for (int q = 0; q < 16; q++)
#pragma unroll
                            for (int cb = 0; cb < 4; cb++)
                                t[i][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pk[16 * cb + l15][4 * q + l4], a[q], t[i][cb], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
        MI355_PDIAG_STAMP(6 * b + 2);
        // ---- the diagonal block: registers -> the LDS image (identity beyond nbb), factorized and inverted there
        {
            const int i_ = 16 * w + l15;
#pragma unroll
            for (int cb = 0; cb < 4; cb++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int j = 16 * cb + l4 + 4 * gq;
                    double v = (i_ == j) ? 1.0 : 0.0;
                    if (i_ < nbb && j < nbb) v = (i_ >= j) ? t[0][cb][gq] : 0.0;
                    Ls[i_][j] = v;
                    Vs[i_][j] = 0.0;
                }
        }
        // the updated strips below the diagonal block go back to memory (they would have to stay in registers across the
        // factorization: 96 of them, next to its own hundred) and return as operands afterwards
        if (b > 0) {
#pragma unroll
            for (int i = 1; i < 4; i++) {
                if (live[i] && c0 + 16 * (w + 4 * i) + l15 < ncols) {
#pragma unroll
                    for (int cb = 0; cb < 4; cb++)
#pragma unroll
                        for (int gq = 0; gq < 4; gq++) (A + (int64_t)(c0 + 16 * cb + 4 * gq) * lda)[roff[i]] = t[i][cb][gq];   // (a block with rows below it is full)
                }
            }
        }
        __syncthreads();
        PotrfTask pt;
        pt.a_off = g.a_off + c0 + (int64_t)c0 * lda; pt.dinv_off = groups[blockIdx.x].dinv_off[b]; pt.ls_off = 0; pt.stile0 = 0;   // (in place: g indexed by b would live in scratch)
        pt.lda = g.lda; pt.nb = nbb; pt.s = g.s; pt.col0 = g.col0 + c0; pt.k1 = 0; pt.nscol = 0; pt.urow0 = 0; pt.fuse_trsm = 0;
        MI355_PDIAG_STAMP(6 * b + 3);
        potrf64_core(pt, Lx, Dinv, info, Ls, Vs);
        MI355_PDIAG_STAMP(6 * b + 4);
        // ---- the strips below it: X = T * W^T (W = Vs, lower), T back from memory in operand layout (column 4 q + lane / 16)
        if (b + 1 < nblk) {
            double a[4][16];
#pragma unroll
            for (int i = 1; i < 4; i++)
#pragma unroll
                for (int q = 0; q < 16; q++) a[i][q] = (A + (int64_t)(c0 + 4 * q) * lda)[roff[i]];
#pragma unroll
            for (int i = 1; i < 4; i++) {
                v4d x[4];
#pragma unroll
                for (int cb = 0; cb < 4; cb++) x[cb] = (v4d){0.0, 0.0, 0.0, 0.0};
                if (live[i]) {
#pragma unroll
                    for (int q = 0; q < 16; q++)
#pragma unroll
                        for (int cb = 0; cb < 4; cb++)
                            if (q <= 4 * cb + 3)                // (on or below W's diagonal; a block with rows below it is full)
                                x[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[16 * cb + l15][4 * q + l4], a[i][q], x[cb], 0, 0, 0);
                    if (c0 + 16 * (w + 4 * i) + l15 < ncols) {
#pragma unroll
                        for (int cb = 0; cb < 4; cb++)
#pragma unroll
                            for (int gq = 0; gq < 4; gq++) (A + (int64_t)(c0 + 16 * cb + 4 * gq) * lda)[roff[i]] = x[cb][gq];
                    }
                }
            }
            __syncthreads();
        }
        MI355_PDIAG_STAMP(6 * b + 5);
    }
}

//------------------------------------------------------------------------------
// k_potrf16: the same for blocks of at most 16 columns (Schedule::potrf_order, Step::nsmall), one WAVE per block and
// POTRF16_WAVES blocks per workgroup: the register factorization of k_potrf64's 16 x 16 diagonal step IS the whole
// job here -- lane r loads row r of the block straight from global memory (coalesced along r for each column),
// the factor goes back the same way, and lane c ends up with column c of the inverse, which is how Dinv stores it.
// No LDS, no barrier.
//------------------------------------------------------------------------------
constexpr int POTRF16_WAVES = 4;

__global__ __launch_bounds__(64 * POTRF16_WAVES) void k_potrf16(
    const PotrfTask *__restrict__ tasks, const int32_t *__restrict__ order, int64_t ntasks,
    double *__restrict__ Lx, double *__restrict__ Dinv, int *__restrict__ info)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15;
    const int64_t e = (int64_t)blockIdx.x * POTRF16_WAVES + wave;
    if (e >= ntasks) return;                                  // no barrier below
    const PotrfTask t = tasks[order[e]];
    const int nb = t.nb;                                      // <= 16 (checked on the host)
    const int64_t lda = t.lda;
    double *__restrict__ A = Lx + t.a_off;

    // lane r (mod 16) holds row r; identity beyond nb; entries right of the diagonal are never consumed
    double a[16], rs[16], w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        double v = (l15 == k) ? 1.0 : 0.0;
        if (l15 < nb && k < nb) v = (l15 >= k) ? A[l15 + (int64_t)k * lda] : 0.0;
        a[k] = v;
    }
    int fail_col = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double d = lane_bcast(a[j], j);
        fail_col = (!(d > 0.0) && fail_col == 0) ? j + 1 : fail_col;
        const double r = rsqrt(d);
        rs[j] = r;
        a[j] = (l15 == j) ? d * r : a[j] * r;
#pragma unroll
        for (int k = j + 1; k < 16; k++) a[k] -= a[j] * lane_bcast(a[j], k);
        __builtin_amdgcn_sched_barrier(0);
    }
    // W = inv(L): lane c holds column c, w[i] = (delta_ic - sum_{k<i} L[i][k] w[k]) / L[i][i]
#pragma unroll
    for (int i = 0; i < 16; i++) {
        double acc = (l15 == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) acc -= lane_bcast(a[k], i) * w[k];
        w[i] = acc * rs[i];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (lane == 0 && fail_col != 0 && info[t.s] == 0) info[t.s] = t.col0 + fail_col;
    if (lane < nb) {
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k <= lane) A[lane + (int64_t)k * lda] = a[k];
        if (t.dinv_off >= 0) {
            double *__restrict__ D = Dinv + t.dinv_off + (int64_t)lane * nb;      // column `lane`, ld = nb
#pragma unroll
            for (int i = 0; i < 16; i++)
                if (i < nb) D[i] = (i >= lane) ? w[i] : 0.0;
        }
    }
    // (the factor and the inverse are stored first: only w stays live through the solve below)
    if (t.fuse_trsm) {
        // the TRSM of the panel step (worker :1175: X = B * inv(L11)^T) for the rows below the block, 64 rows at a
        // time, lane = row: x[c] = sum_{k <= c} B[row][k] * W[c][k], W[c][k] = w[c] of lane k.  The trip count and
        // every cross-lane read are wave-uniform; rows beyond the panel are masked at the loads and stores only.
        const int below = (int)lda - nb;
        for (int r0 = 0; r0 < below; r0 += 64) {
            const int row = nb + r0 + lane;
            const bool ok = row < (int)lda;
            double b[16];                                     // the whole row first: the results overwrite it
#pragma unroll
            for (int k = 0; k < 16; k++) b[k] = (ok && k < nb) ? A[row + (int64_t)k * lda] : 0.0;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k <= c; k++) acc += b[k] * lane_bcast(w[c], k);
                if (ok && c < nb) A[row + (int64_t)c * lda] = acc;
            }
        }
    }
}

}  // namespace mi355
