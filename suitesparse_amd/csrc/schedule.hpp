// schedule.hpp -- host-side (pure C++, no HIP) schedule builder for the MI355X
// supernodal numeric engine.
//
// The reference walks linked lists Head/Next/Lpos while it factorizes
// (CHOLMOD/Supernodal/t_cholmod_super_numeric_worker.c:537-551,585-698,950-967,
// 1222-1235).  That walk depends only on the symbolic structure (Super/Lpi/Ls), so
// here it is replayed ONCE per symbolic factor into flat, level-ordered task lists
// that the device consumes with a handful of batched launches per etree level:
//
//   level l:  [panel steps it = 0,1,..]  POTRF(diag 64-block) -> GEMM(trsm via
//             inverse) -> GEMM(trailing update)          (all supernodes of level l)
//             [update step]  GEMM(scatter) for every (d,s) pair with level(d) = l
//
// Descendant d updates ancestor s  <=>  s is a proper ancestor of d in the
// supernodal etree, hence level(s) > level(d): every panel of level l is complete
// (all its updates applied) when level l starts.
#pragma once
#include <cstdint>
#include <vector>
#include <string>

namespace mi355 {

constexpr int NB = 64;        // diagonal-block / panel width of the device POTRF
constexpr int OB = 256;       // default first outer block width (Tuning::ob); nested widths 64 / 256 / 1024 measured best at 150^3
constexpr int TILE = 64;      // GEMM output tile (TILE x TILE per workgroup), small-tile kernel
constexpr int BTILE = 128;    // big-tile kernel
constexpr int SOLVE_ROWS = 64;   // rows of L per workgroup (tile) in the solve update kernels

enum GemmMode : int32_t {
    GEMM_SCATTER_SUB = 0,     // Lx[c_off + rm[i] + rm[j]*ldc] -= C(i,j), i>=j, atomic   (a3.5-a3.8)
    GEMM_DIRECT_SUB  = 1,     // Lx[c_off + i + j*ldc]         -= C(i,j), i>=j           (inside a3.9/a3.10)
    GEMM_OVERWRITE   = 2      // Lx[c_off + i + j*ldc]          = C(i,j)  (B from the Dinv workspace; TRSM a3.10)
};

// C (M x N) = A (M x K) * B (N x K)^H ; A, B column-major with leading dims lda, ldb.
struct GemmTask {
    int64_t a_off;      // entry offset of A(0,0) in Lx
    int64_t b_off;      // entry offset of B(0,0) in Lx (or in the Dinv workspace, GEMM_OVERWRITE)
    int64_t c_off;      // entry offset of the target origin in Lx
    int64_t rm_off;     // offset of this pair's relative map (GEMM_SCATTER_SUB), else -1
    int32_t lda, ldb, ldc;
    int32_t M, N, K;
    int32_t mode;
    int32_t big;        // 1: tiles counted with BTILE (big-tile kernel), 0: TILE
    int64_t tile0;      // first tile of this task inside its launch
    int32_t fuse_potrf; // >= 0: index (in Schedule::potrf) of the diagonal block that this trailing update completes -- the
                        // workgroup of tile (0, 0) factorizes it right after its stores (k_gemm128), and the next POTRF
                        // step has no launch for it; -1: none
    int32_t pad;
};

struct PotrfTask {
    int64_t a_off;      // entry offset of the diagonal block in Lx
    int64_t dinv_off;   // entry offset of the nb x nb inverse (column-major, ld nb) in Dinv
    int64_t ls_off;     // Lpi[s]: row list of the supernode (solves)
    int64_t stile0;     // first solve row-tile of this block inside its step (solves)
    int32_t lda;        // = nsrow
    int32_t nb;         // block size (<= NB)
    int32_t s;          // supernode
    int32_t col0;       // first column of the block inside s (info = col0 + j + 1)
    int32_t k1;         // Super[s]
    int32_t nscol;      // columns of s
    int32_t urow0;      // first row (inside s) of the solve row tiles: col0 + nb, or the end of the solve group
    int32_t fuse_trsm;  // != 0 (tiny classes on, a real supernode of <= 16 columns): k_potrf16 also solves the rows below
                        // the block (the TRSM of the panel step), and the step's TRSM launch has no task for it
};

// Solve groups (real factor, few right-hand sides): SOLVE_GROUP consecutive 64-column blocks of a supernode are
// swept by ONE workgroup (their diagonal solves and the updates among them), and the rows below the group get
// the updates of all its blocks in ONE launch -- two launches per 256 columns instead of eight; the sweep
// is a chain of dependent launches, so its length is what the solve time is made of.
constexpr int SOLVE_GROUP = 4;
struct SolveGroupTask {
    int64_t a_off;                      // L[col0][col0] of the group's first block
    int64_t dinv_off[SOLVE_GROUP];      // block inverses (ld = block width)
    int32_t lda;                        // = nsrow
    int32_t ncols;                      // columns of the group; blocks of NB, the last may be narrower
    int32_t x0;                         // first row of X the group owns (Super[s] + col0)
    int32_t pad;
};
struct SolveRange { int32_t task, tile0, ntiles, pad; };   // a run of row tiles of one update task (backward sweep)
// Many right-hand sides (round 3): the rows below a solve group take the update of ALL the group's columns in one
// pass (contraction depth up to 256 columns, 128 for a complex factor) instead of one pass per 64-column block, so
// the 64 x nrhs rows of Y that a row tile touches cross HBM once per group, not once per block (measured in round 2:
// the per-block kernels moved twice the algorithmic bytes, half of them rows of Y).  One "wide" task per group of
// more than 64 columns (two when a complex group exceeds 128 columns); groups of one block keep the per-block
// kernels, which hold four workgroups per CU where the wide kernels (x1 of the whole group in LDS) hold one.
struct SolveGroupUpd {
    int64_t a_off;      // entry offset of L[first row below the group][first column of the task]
    int64_t ls_off;     // index into Ls of the first row below the group
    int32_t lda;        // = nsrow
    int32_t ncols;      // columns of the task (<= SolveSchedule::kmax)
    int32_t x0;         // first row of X the task's columns own
    int32_t nrows;      // rows below the group
};
struct SolveStep {
    int64_t gfirst, gcount;             // group tasks
    int64_t ufirst, ucount;             // per-block update tasks (PotrfTask with urow0 = end of the group)
    int64_t stiles, tmap_off;           // their row tiles
    int32_t maxblk, pad;                // most 64-column blocks of any group of the step
    int64_t rfirst, rcount;             // runs of row tiles for the backward sweep (scalar kernels)
    int64_t rmfirst, rmcount;           // ... for the MFMA kernel (shorter: it holds fewer workgroups per CU)
    // wide path (many right-hand sides): the same rows, split into the groups of one block (per-block kernels on
    // lists of their own) and the wider groups (SolveGroupUpd tasks, runs of 64-row tiles)
    int64_t nufirst, nucount;           // per-block update tasks of the one-block groups (in nutasks)
    int64_t nstiles, ntmap_off;         // their row tiles (in ntile_task)
    int64_t nrmfirst, nrmcount;         // their runs for the backward MFMA kernel (in nranges_mma)
    int64_t wfirst, wcount;             // wide tasks (in wtasks)
    int64_t wrfirst, wrcount;           // segments (runs of 64-row tiles of one task) over them (in wranges; task relative to wfirst)
    int64_t wgfirst, wgcount;           // the step's workgroups: wwg[wgfirst + g] .. wwg[wgfirst + g + 1] = segments of workgroup g
                                        // (relative to wrfirst; wgcount + 1 offsets)
    int32_t wmaxblk;                    // most 64-column blocks of any wide task of the step
    int32_t wsplit;                     // look-ahead of the sweeps (tuning key 30): a step of ONE group split into its near rows -- the first
                                        // wsplit (= SOLVE_GROUP) tiles below the group, i.e. the rows of the supernode's next group, listed
                                        // per block in nutasks / ntile_task / nranges_mma -- and the rows beyond them (the wide tasks, which
                                        // then start wsplit tiles lower); 0: the step is not split
};
struct SolveSchedule {
    std::vector<SolveGroupTask> groups;
    std::vector<PotrfTask> utasks;
    std::vector<int32_t> tile_task;
    std::vector<SolveRange> ranges, ranges_mma;
    std::vector<SolveStep> steps;       // forward order
    int32_t kmax = 0;                   // contraction depth of the wide tasks (0: wide lists not built)
    std::vector<PotrfTask> nutasks;
    std::vector<int32_t> ntile_task;
    std::vector<SolveRange> nranges_mma;
    std::vector<SolveGroupUpd> wtasks;
    std::vector<SolveRange> wranges;
    std::vector<int32_t> wwg;
};

// one (descendant d -> ancestor s) update, the unit of a3.4 in SURVEY.md section 8
struct UpdatePair {
    int32_t d, s;
    int64_t ls_src;     // index into Ls of the first row of d that lies in s (pdi1)
    int64_t ls_dst;     // Lpi[s]
    int32_t ndrow1, ndrow2, nsrow;
    int64_t rm_off;     // offset of its relative map (ndrow2 entries)
};

// Panel groups (round 3, Tuning::panel_group): the panel chain of a supernode -- POTRF of a 64-column diagonal block, TRSM
// of the rows below it, K = 64 update of the rest of the 256-column window, three dependent launches per 64 columns and
// the part of a factorization that runs far below the rate of the big updates -- is taken 256 columns at a time:
//   STEP_PDIAG  one workgroup factorizes the whole diagonal window (<= 256 x 256: its up to four diagonal blocks, their
//               inverses and the blocks between them; k_panel_diag), then
//   STEP_PTRSM  the rows below the window are solved against it in ONE pass (k_panel_trsm: the forward chain of the
//               solve groups with the panel's rows as right-hand sides -- column-major L IS the row-major image of them),
// and the window's K = 256 (and, where a 1024-column block ends, K = 1024) trailing updates follow as before: three
// launches per 256 columns instead of thirteen.  The 64-column PotrfTasks stay in the schedule (the solve work lists are
// built over them) as tasks without a launch (Step::nfused counts them at the end of their steps).
struct PanelGroup {
    int64_t a_off;                      // L[col0][col0]: origin of the diagonal window in Lx
    int64_t dinv_off[4];                // inverses of the window's diagonal blocks (ld = block width)
    int64_t wg0;                        // first workgroup of this task in its STEP_PTRSM launch
    int32_t lda;                        // = nsrow
    int32_t ncols;                      // columns of the window (<= 256; blocks of NB, the last may be narrower)
    int32_t s;                          // supernode (info)
    int32_t col0;                       // first column of the window inside s (info = col0 + j + 1)
    int32_t below;                      // rows below the window: lda - col0 - ncols
    int32_t pad;
};
enum StepKind : int32_t { STEP_POTRF = 0, STEP_GEMM = 1, STEP_PDIAG = 2, STEP_PTRSM = 3 };
struct Step {
    int32_t kind;
    int32_t level;
    int32_t big;        // STEP_GEMM: tasks of this step use the big-tile kernel
    int32_t trsm;       // STEP_GEMM: every task is a TRSM by block inverse (GEMM_OVERWRITE, N = K = nb <= 64)
    int64_t first;      // first task (index into potrf[] or gemm[])
    int64_t count;      // number of tasks
    int64_t ntiles;     // STEP_GEMM: total tiles (= grid size); STEP_POTRF: = count
    int64_t stiles;     // STEP_POTRF: row tiles (SOLVE_ROWS rows each) below the blocks, for the solves
    int64_t tmap_off;   // offset of this step's tile -> task table (gemm_tile_task / solve_tile_task)
    int32_t flags;      // STEP_SIDE / STEP_JOIN / STEP_TINY
    int32_t nsmall;     // STEP_POTRF with Schedule::potrf_order: the first nsmall entries of the step's order list are
                        // blocks of at most 16 columns (k_potrf16: one wave per block), the rest go to k_potrf64
    int32_t nfused = 0; // STEP_POTRF: the LAST nfused tasks of the step are factorized by the trailing update that completes
                        // them (GemmTask::fuse_potrf) and have no launch here; the solves still see them as tasks of the step
    int32_t ngrouped = 0; // STEP_POTRF: of those, the blocks that a panel group factorizes (STEP_PDIAG); STEP_PTRSM: rows per workgroup
};
// Look-ahead: the part of an outermost trailing update that lies beyond the NEXT outer block does not feed the
// next block's panel chain, so it runs on a second stream (STEP_SIDE) next to that chain; whatever touches its
// targets or reads all of the supernode afterwards first waits for it (STEP_JOIN).  Off by default
// (Tuning::lookahead): at 150^3 it shortens the factorization by 1.6 % only -- the panel-chain kernels wait
// for workgroup slots behind the update's queued tiles and run twice as long, stream priorities
// notwithstanding -- and the overlapped launches blur the per-kernel durations the roofline is built from.
constexpr int32_t STEP_SIDE = 1, STEP_JOIN = 2;
// STEP_TINY: every task of the step is a real update (not a TRSM) of at most one 64 x 64 tile with
// M*N*K <= Tuning::tiny_max; the engine runs such a step with k_gemm_tiny, one WAVE per task (no LDS, no barrier,
// four tasks per workgroup) instead of one workgroup per task.  The low levels of the tree are made of hundreds of
// thousands of such tasks (2D 2000^2, level 0: 661 299 updates of a few entries each), where a launch is bound by
// workgroup slots and the latency chain inside one workgroup, not by flops.  Tasks and tile table entries are the
// ordinary ones (tile (0, 0) of a 64 x 64-tile task).  On by default since round 2 (tiny_max = 2048 flops M*N*K: 2D 2000^2 27.8 -> 20.2 ms on the MI355X, 3D configs unchanged;
// profiles/r02_tiny_classes.txt); 0 switches the classes off.
constexpr int32_t STEP_TINY = 4;

struct Symbolic {
    int64_t n = 0, nsuper = 0, ssize = 0, xsize = 0;
    const int64_t *Super = nullptr, *Lpi = nullptr, *Lpx = nullptr, *Ls = nullptr;
    bool no_big = false;        // route every GEMM task to the 64x64-tile kernel
    bool cplx = false;          // complex factor: the big-tile kernel covers 64 x 128 complex entries per tile
};

// Owner-computes distribution of the masked (top) supernodes over `world` ranks (multi-GPU): the
// columns of every such supernode are cut into outer blocks of the widest nested width (Tuning::ob2); block
// (s, jb) has one owner, who alone factorizes it and applies every update that targets it.  A finished block
// is then broadcast (BcastPoint) before anybody reads it.  The schedule of rank `rank` holds its tasks only.
struct DistTop { int32_t rank = 0, world = 1; };
struct BcastPoint {
    int64_t step;       // the broadcast is ISSUED before steps[step] (after everything earlier: the block is finished)
    int64_t wait_step;  // ... and must have ARRIVED before steps[wait_step] (>= step).  Between the two lie launches that
                        // do not read the block: with Tuning::dist_lookahead the far part of the PREVIOUS block's
                        // trailing update (its targets beyond the next block), so the broadcast travels while they run
    int64_t lo, count;  // entries of Lx: the block's columns, contiguous
    int32_t root;       // owner of the block
    int32_t s;          // supernode
    int32_t mine;       // root == the rank this schedule belongs to (the sender may go on READING what it sends)
    int32_t last;       // this piece completes its outer block (Tuning::dist_pieces: a block is sent in finished pieces
                        // of the first nested width while its panel chain is still running; else every entry is last)
};

struct Schedule {
    std::vector<BcastPoint> bcasts;     // DistTop schedules only, in the order every rank must issue them
    std::vector<int32_t> supermap;      // column -> supernode
    std::vector<int32_t> parent;        // supernodal etree (-1 = root)
    std::vector<int32_t> level;         // height above the leaves
    int32_t nlevels = 0;
    std::vector<PotrfTask> potrf;
    // With Tuning::tiny_max > 0 and a real factor: per POTRF step (same offsets as potrf[]), the step's task indices
    // (relative to the step) with the blocks of at most 16 columns first -- 98 % of the diagonal blocks of a 2D
    // problem, 85 % of a 3D one: k_potrf64 holds two workgroups per CU (its 64 x 64 LDS images), k_potrf16 keeps such
    // a block in the registers of one wave and twelve of them per CU.  The task order itself is left alone (the
    // solve work lists are built over it).  Empty otherwise.
    std::vector<int32_t> potrf_order;
    std::vector<GemmTask> gemm;
    std::vector<PanelGroup> pgroups;    // STEP_PDIAG / STEP_PTRSM tasks (a PTRSM step runs the groups of the PDIAG step before it)
    std::vector<int32_t> pg_wg_task;    // STEP_PTRSM: workgroup -> task (relative to the step's first), Step::tmap_off / ntiles
    std::vector<UpdatePair> pairs;
    std::vector<Step> steps;
    // One entry per GEMM workgroup, indexed by blockIdx.x inside its launch: the task (index relative
    // to the step's first task) and the tile coordinates.  The host chooses the order: tiles of a task
    // are grouped in 8x8 super-tiles (64 tiles that share 8 A-panels and 8 B-panels), and the sequence
    // is dealt to the 8 XCDs in contiguous chunks (blocks b and b+8 share an XCD and its L2), so the
    // ~64 workgroups resident on one XCD at a time re-read each other's operand panels from L2.
    struct TileEntry { int32_t task; uint32_t tmtn; };      // tm = tmtn & 0xffff, tn = tmtn >> 16
    std::vector<TileEntry> gemm_tile_tab;
    // solve row tiles: tile -> task (relative), one load instead of a binary search
    std::vector<int32_t> solve_tile_task;
    int64_t relmap_size = 0;            // total relative-map entries (sum of ndrow2 over pairs)
    int64_t dinv_size = 0;              // total entries of the block inverses (sum nb^2)
    // truncation (not-positive-definite repeat, worker :1084-1146); s_limit == nsuper: none
    int64_t s_limit = 0;
    int64_t nscol2_limit = 0;
    // statistics
    double flops_update = 0, flops_panel = 0;
    double flops_big = 0, flops_small = 0;  // useful flops of the GEMM tasks routed to k_gemm128 / k_gemm64
    int64_t tiles_big = 0, tiles_small = 0;
    double bytes_assembly = 0;          // sum over pairs of ndrow1*ndrow2*8
    int64_t max_nscol = 0, max_nsrow = 0;
};

// Validates the symbolic factor (monotone pointers, sorted in-range row lists, the
// leading nscol rows of each supernode are its own columns, cholmod_super_symbolic.c:
// 779-782).  Returns "" if fine, else a message.
std::string validate_symbolic(const Symbolic &S);

// Full schedule: s_limit = nsuper.  Truncated schedule for the not-posdef repeat:
// supernodes < s_limit are processed completely, supernode s_limit only in its first
// nscol2_limit columns (0: not at all), later supernodes not at all.
// mask (optional, nsuper entries): only supernodes with mask[s] != 0 get their panel factorized
// and push their updates -- the multi-GPU phases (own subtrees / shared top) of SURVEY 8(e).
// dinv_base: first Dinv slot this schedule may use (two masked schedules of one factor keep their block
// inverses side by side, so that the solves can use both).
// dist (with a mask, full schedules only): the owner-computes schedule of one rank, see DistTop; the Dinv slots are
// numbered as in the undistributed schedule of the same mask.
void build_schedule(const Symbolic &S, int64_t s_limit, int64_t nscol2_limit, Schedule &out,
                    const int8_t *mask = nullptr, int64_t dinv_base = 0, const DistTop *dist = nullptr);

// the grouped solve sweep over the POTRF steps of sc (any schedule: full, or one part of a partition)
struct Schedule;
void build_solve_groups(const Schedule &sc, SolveSchedule &out, int kmax = 0 /* 256 real, 128 complex; 0: no wide lists */);

// tasks at least this large go to the big-tile (BTILE) kernel; settable for experiments
struct Tuning { int big_min_m = 192; int big_min_n = 96; int ob = 256; int supertile = 8; int xcd_deal = 0; int ob2 = 1024; int ob3 = 0; int solve_mma_min = 5; int solve_mma_min_c = 3; int solve_group = 1; int solve_bwd_div = 256; int solve_bwd_div_mma = 4096; int solve_fwd_div_mma = 4096; int lookahead = 0; int tiny_max = 2048; int dist_lookahead = 1; int dist_pieces = 1; int host_chunk = 32 << 20; int dist_reduce_channel = 0; int host_lanes = 4; int host_lane_mb = 32; int host_staged_min_kb = 4096; int solve_wide = 1; int solve_wide_wgs = 256; int solve_wide_1024 = 0; int fuse_potrf = 0; int deterministic = 0; int panel_group = 64; int solve_wide_ring = 2; int solve_lookahead = 0; int solve_group_inv = 4; int solve_wide_ksplit = 1; };
Tuning &tuning();
void append_tiles(std::vector<Schedule::TileEntry> &seq, int32_t task, const GemmTask &t, int tile_m, int tile_n);
void deal_tiles(const std::vector<Schedule::TileEntry> &seq, std::vector<Schedule::TileEntry> &tab);

// Every task of the schedule (and, when given, of its solve work lists) against the extents of the arrays the
// kernels index with it: Lx (xsize entries), Dinv, the relative maps, Ls, X (n rows) and the tables themselves.
// "" if fine, else what is wrong.  The engine runs it before a schedule reaches the device: a task that points
// outside its arrays is a GPU memory fault, and that takes the machine down, not just the process.
std::string check_schedule_bounds(const Symbolic &S, const Schedule &sc, const SolveSchedule *ss = nullptr,
                                  int64_t dinv_extent = -1 /* entries of the Dinv buffer; default: sc.dinv_size */);

// Copy of the factor to the host WHILE it is being computed (mi355_factor_numeric_host): Lx is cut into chunks of whole
// supernodes; a chunk may be copied once the first `ready_step` steps of the schedule have run -- nothing after them
// writes into it (its supernodes' levels are complete).  min_entries: smallest chunk worth a copy of its own.
struct HostCopyChunk { int64_t lo, hi, ready_step; };
void build_host_copy_plan(const Symbolic &S, const Schedule &sc, int64_t min_entries, std::vector<HostCopyChunk> &out);
// "" or why the plan is wrong: the chunks must tile [0, xsize) and no step at or after a chunk's ready_step may write it
std::string check_host_copy_plan(const Symbolic &S, const Schedule &sc, const std::vector<HostCopyChunk> &plan);

// true if the table of step st lists every valid tile of every task of the step exactly once
bool check_tile_table(const Schedule &sc, const Step &st);

// number of TILE-tiles of one task
int64_t gemm_tiles(int32_t M, int32_t N, int32_t mode, int32_t tile);
int64_t gemm_tiles(int32_t M, int32_t N, int32_t mode, int32_t tile_m, int32_t tile_n);
// tile shape (rows, columns, in entries of the factor) by GemmTask::big: 0 = k_gemm64(c) 64 x 64,
// 1 = k_gemm128 128 x 128, 2 = k_gemm128 on a complex factor: 64 x 128 complex entries (128 x 128 reals)
inline int tile_rows(int big) { return big == 1 ? BTILE : TILE; }
inline int tile_cols(int big) { return big == 0 ? TILE : BTILE; }

}  // namespace mi355
