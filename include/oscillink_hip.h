/* oscillink_hip.h -- C ABI of liboscillink_hip.so: the MI355X (gfx950) implementation of the
 * Oscillink lattice "settle" hot path.
 *
 * The reference (Maverick0351a/Oscillink v0.1.13) is pure Python + NumPy and has NO native/FFI
 * boundary: its boundary is the Python class `OscillinkLattice` (oscillink/core/lattice.py) calling
 * the NumPy functions in oscillink/core/{graph,solver,receipts}.py.  Every entry point below
 * replaces one of those Python seams 1:1 (file:line cited per function; paths are relative to the
 * reference checkout).  The Python mirror of the class that binds these symbols with ctypes is
 * `oscillink_amd/lattice.py`; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; all host arrays are row-major, contiguous, caller-owned; fp32 / int32 / int64.
 *   - every call returns 0 on success or a negative OSC_E_* code; osc_last_error() gives the text.
 *   - a handle owns one HIP stream and all its device memory; handles are independent and a handle
 *     must not be used from two threads at once (the reference's cloud runs one lattice per request
 *     thread: cloud/app/main.py:1030-1061).  No process-global mutable state except the last-error
 *     string of failed osc_create calls (thread-local).
 *   - there is NO CPU fallback: without a usable gfx950 device osc_create fails with OSC_E_NODEVICE.
 */
#ifndef OSCILLINK_HIP_H
#define OSCILLINK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct osc_lattice* osc_handle;

enum {
  OSC_OK = 0,
  OSC_E_INVALID = -1,     /* bad argument (the Python layer maps this to ValueError)            */
  OSC_E_NODEVICE = -2,    /* no HIP device / wrong architecture                                  */
  OSC_E_HIP = -3,         /* a HIP runtime call failed                                           */
  OSC_E_STATE = -4,       /* call order problem (e.g. deltaH before any U* solve)                */
  OSC_E_UNSUPPORTED = -5, /* valid request this build cannot serve (e.g. kneighbors > 128)       */
  OSC_E_COMM = -6         /* RCCL failure                                                        */
};

/* precond argument of osc_settle (lattice.py:164,186): "jacobi" -> 1, anything else -> 0 */
enum { OSC_PRECOND_NONE = 0, OSC_PRECOND_JACOBI = 1 };

/* ---- library / device ---------------------------------------------------------------------- */
const char* osc_version(void);
int osc_device_count(int32_t* n);                       /* never initialises a device context     */
int osc_device_name(int32_t device, char* out, int32_t cap);
int osc_device_synchronize(int32_t device);
const char* osc_last_error(osc_handle h);               /* h may be NULL: error of the last failed osc_create on this thread */

/* Pinned host memory for result arrays (what the reference allocates with NumPy for `U`, `Y`, `U*`: lattice.py:54-56,
 * 206, 262-271): a read-back (osc_get_U / osc_get_Y / osc_get_ustar / osc_solve_ustar) into such an array is one DMA at
 * PCIe rate; into ordinary pageable memory it goes through two pinned staging buffers and a threaded host copy.  Freed
 * blocks are parked and reused (pinning is slow).  Needs a HIP runtime, no device context of its own. */
int osc_host_alloc(int64_t bytes, void** out);
int osc_host_free(void* p);

/* ---- construction: OscillinkLattice.__init__ (lattice.py:33-110) ---------------------------- */
/* Copies Y (N x D) to the device, sets U = Y, B = 1, psi = 0, lams = (1.0, 0.5, 4.0).
 * build_graph != 0: builds the mutual-kNN graph on the device -- graph.py:8-66 (mutual_knn_adj),
 * :69-83 (row_sum_cap), :86-93 (normalized_laplacian).  k is clamped to [1, N-1] (lattice.py:60).
 * deterministic != 0 -> ties ordered (similarity desc, index asc) (graph.py:46-49); the
 * non-deterministic reference path leaves ties unspecified (graph.py:59) and this build uses the
 * same total order for it.  seed < 0 means None.  seed >= 0 (`neighbor_seed`): the reference adds a float64
 * uniform(-1e-8, 1e-8) jitter drawn from default_rng(seed) to all N^2 similarities before its argpartition
 * (graph.py:54-58).  fp32 similarities below 0.125 are spaced 7.5e-9 apart (3.7e-9 below 0.0625), so that jitter can
 * reorder exact ties AND candidates within 1-2 ulp of each other at the k-th place; reproducing it bit for bit would
 * need the same N^2 random stream.  This build treats the seed as a tie-break only: the neighbour lists are those of
 * the total order above, i.e. they can differ from the reference's seeded lists only where two candidates for the
 * last list place are 1-2 ulp apart (tests/golden/case_seed_*.npz: identical edge sets on Gaussian anchors).
 * build_graph == 0: no graph yet; call osc_set_csr (from_state / parity tests).
 * Y is read until the call returns and not afterwards.  With build_graph != 0 the anchors of a large lattice (>= 64 MB,
 * unpadded rows; beyond 768 columns from ~130 000 rows) travel to the device in pieces WHILE the build runs on the pieces
 * that have arrived (a copy
 * stream, a second build stream and a few host threads for the duration of the call; OSC_CREATE_STREAM=0: one transfer,
 * then the build): same lattice bit for bit, create 20.8 -> 16.5 ms at N = 100 000, D = 768, k = 32. */
int osc_create(const float* Y, int64_t N, int32_t D, int32_t k, float row_cap, int32_t deterministic,
               int64_t seed, int32_t device, int32_t build_graph, osc_handle* out);
int osc_destroy(osc_handle h);

/* rebuild_graph (lattice.py:760-801) */
int osc_rebuild_graph(osc_handle h, int32_t k, float row_cap, int32_t deterministic, int64_t seed);

/* nnz = stored directed edges (== count of A > 0), max_deg = widest row, build_ms = wall time of the last graph build
 * (lattice.py:76-77 `_graph_build_ms`); after osc_create it includes the anchors' transfer, which the build overlaps */
int osc_graph_stats(osc_handle h, int64_t* nnz, int32_t* max_deg, double* build_ms);

/* how the last device build ran: prefilter != 0 -> fp16-MFMA prefilter + exact fp32 re-scoring (2 = the
 * register-resident-panel GEMM with sampled thresholds, 1 = the 128 x 128 tile with in-kernel lists); fallback_rows = rows whose
 * candidate list could not be proven and were redone by the all-fp32 kernel; small_solves = solves served by the
 * one-launch small-lattice CG since creation */
int osc_build_info(osc_handle h, int32_t* prefilter, int32_t* fallback_rows, int64_t* small_solves);

/* internal row order of the last graph: reordered != 0 -> the rows are stored in BFS (locality-preserving) order, which
 * is invisible at this API (every call speaks the caller's row ids); clustering = sampled local clustering
 * coefficient that decided it (OSC_REORDER=0/1 overrides the automatic choice) */
int osc_order_info(osc_handle h, int32_t* reordered, double* clustering);
/* the internal row order itself (test / diagnostic aid): perm[new] = caller's row id, N entries; the identity when the
 * rows are stored in the caller's order */
int osc_get_row_order(osc_handle h, int32_t* perm);

/* how one operator apply (the CG matvec over this handle's column window) is launched: launches = kernel launches per
 * apply, slab_cols = columns each launch covers, xs_workgroups = 0 for sequential column slabs swept by the whole
 * chip, > 0 for one launch of XCD-affine 32-column slabs with that many workgroups per XCD (measurement aid) */
int osc_spmm_plan(osc_handle h, int32_t* launches, int32_t* slab_cols, int32_t* xs_workgroups);

/* the CG matvec of the last general-path solve: src_blocks = 0 for the plain apply, else the number of source-row blocks
 * the blocked apply walked (chosen when the 32-column slab an XCD gathers from, N x 128 B, is at least OSC_BLK_MB = 2
 * MiB; a chain prior of up to 4096 path rows is applied by a small launch behind it; the block count follows the mean degree;
 * OSC_SPMM_BLOCKED = 0 off / n forces n blocks); blocked_applies = such matvecs enqueued since creation (measurement
 * aid) */
int osc_apply_info(osc_handle h, int32_t* src_blocks, int64_t* blocked_applies);

/* the block-major copy of the graph the blocked matvec walks, built for `nb` source blocks (test / diagnostic aid; no
 * reference counterpart): slot_col / slot_w [nb][N][4] = {neighbour row, W_ij} per (source block, row, slot), unused slots
 * {first row of the block, 0}; rows whose edges exceed 4 nb slots list the rest in over_col / over_w[over_first[i] ..
 * + over_count[i]) (at most over_cap entries in all).  Any output may be NULL.  OSC_E_UNSUPPORTED on a lattice stored in
 * an internal row order. */
int osc_get_blocked_copy(osc_handle h, int32_t nb, int32_t* slot_col, float* slot_w, int32_t* over_first,
                         int32_t* over_count, int32_t* over_col, float* over_w, int32_t over_cap);

/* CSR view of the graph for `.A`, `.L_sym`, `_signature()` (lattice.py:729-744) and export_state
 * (:582-624).  rowptr has N+1 entries; col/a/w have nnz entries, columns ascending within a row;
 * a = capped adjacency A_ij (> 0), w = A_ij / (sqrt_deg_i sqrt_deg_j); sqrt_deg has N entries.
 * Any output pointer may be NULL. */
int osc_get_csr(osc_handle h, int64_t* rowptr, int32_t* col, float* a, float* w, float* sqrt_deg);

/* the first min(cap, nnz) stored edges as int64 (i, j) pairs in row-major order, columns ascending within a row:
 * `argwhere(A > 0)[:cap]` of _signature() (lattice.py:729-744) without moving the whole graph to the host */
int osc_edge_prefix(osc_handle h, int32_t cap, int64_t* pairs, int32_t* n);

/* Inject a (symmetric, zero-diagonal, already capped) adjacency as CSR; recomputes sqrt_deg and W
 * exactly as normalized_laplacian (graph.py:86-93).  Mirrors from_state's `lat.A = A;
 * lat.L_sym, lat.sqrt_deg = normalized_laplacian(lat.A)` (lattice.py:709-713). */
int osc_set_csr(osc_handle h, const int64_t* rowptr, const int32_t* col, const float* a);

/* raw per-row top-k lists of the last device graph build (idx/val: N x k_eff, unsorted within a row;
 * val is the similarity clipped at 0) -- graph.py:59-62.  Test/diagnostic aid. */
int osc_get_knn_lists(osc_handle h, int32_t* idx, float* val, int32_t* k_eff);

/* ---- state setters -------------------------------------------------------------------------- */
/* set_query / set_gates (lattice.py:114-127): psi has D entries; gates (N entries) may be NULL */
int osc_set_query(osc_handle h, const float* psi, const float* gates_or_null);
/* add_chain + build_path_laplacian (lattice.py:129-149, graph.py:96-111); weights may be NULL (all 1) */
int osc_set_chain(osc_handle h, const int32_t* chain, const float* weights_or_null, int32_t len, float lamP);
int osc_clear_chain(osc_handle h);                      /* lattice.py:151-157 */
int osc_set_lams(osc_handle h, float lamG, float lamC, float lamQ);
int osc_get_U(osc_handle h, float* out);                /* N x D */
int osc_get_Y(osc_handle h, float* out);                /* N x D: the device's private copy of the anchors */
int osc_set_U(osc_handle h, const float* U_or_null);    /* NULL -> U = Y (device copy, ordered by the handle's stream: returns without waiting for it) */

/* ---- solves --------------------------------------------------------------------------------- */
/* settle (lattice.py:159-230) + cg_solve (solver.py:6-37): one implicit-Euler step
 * (I + dt M) U+ = U + dt (lamG Y + lamQ B 1 psi^T), Jacobi diagonal 1 + dt (lamG + lamQ B + [lamP]),
 * x0 = Y (warm_start == 0) | U | (1-w) Y + w U with w = clamp(inertia, 0, 1) (lattice.py:751-758).
 * Stop test: max over columns of ||r_c||_2 <= tol, checked before the beta/p update.  Never fails on
 * non-convergence: iters = max_iters and res is the last residual (lattice.py:206-212).
 * ms = host wall time of x0 selection + CG, device-synchronised (the reference's t_ms). */
int osc_settle(osc_handle h, float dt, int32_t max_iters, float tol, int32_t precond, int32_t warm_start,
               float inertia, int32_t* iters, float* res, double* ms);
/* solve_Ustar (lattice.py:232-290): M U* = lamG Y + lamQ B 1 psi^T from x0 = Y, Jacobi lamG + lamQ B + [lamP].
 * U* stays resident on the device for osc_deltaH / receipts; Ustar_out (N x D) may be NULL. */
int osc_solve_ustar(osc_handle h, float tol, int32_t max_iters, float* Ustar_out, int32_t* iters, float* res,
                    double* ms);
/* 1 iff a U* of the current state (graph, psi, gates, lams, chain) is resident on the device */
int osc_has_ustar(osc_handle h, int32_t* yes);
/* copy the resident U* (N x D) to the host; OSC_E_STATE if no solve happened since the last state change */
int osc_get_ustar(osc_handle h, float* out);
/* n selected rows (caller's row ids) of Y (which = 0), U (1) or the resident U* (2) into out (n x D): what
 * chain_receipt (lattice.py:466-528) reads -- the chain nodes and their neighbours -- without moving N x D floats */
int osc_get_rows(osc_handle h, int32_t which, const int32_t* rows, int32_t n, float* out);
/* residual after every iteration of the last solve (solver.py:29), n <= cap entries written */
int osc_residual_history(osc_handle h, float* out, int32_t cap, int32_t* n);

/* screened diffusion solve used by compute_diffusion_gates(method="cg")
 * (preprocess/diffusion.py:130-151): (L_sym + gamma I) h = s, x0 = 0, Jacobi diag(L)+gamma = 1+gamma. */
int osc_cg_single_rhs(osc_handle h, float gamma, const float* s, float tol, int32_t max_iters, float* h_out,
                      int32_t* iters, float* res);
/* cosine of every anchor row with psi (diffusion.py:104-107): out[i] = <Y_i/(|Y_i|+1e-12), psi/(|psi|+1e-12)> */
int osc_cosine_to(osc_handle h, const float* psi, float* out);
/* the same over the rows of the resident U*: bundle()'s alignment term (lattice.py:530-568:
 * align_i = <U*_i / (|U*_i| + 1e-12), psi / (|psi| + 1e-12)>); OSC_E_STATE without a resident U*.  With psi = an
 * anchor row, osc_cosine_to gives the similarity row mmr_diversify needs for one chosen item (graph.py:114-133). */
int osc_ustar_cosine_to(osc_handle h, const float* psi, float* out);
/* out[i] = <Yn_i, Yn_row>: the similarity row of one chosen item for mmr_diversify (graph.py:114-133), from the
 * device's own copy of the anchors */
int osc_cosine_to_row(osc_handle h, int64_t row, float* out);
/* mmr_diversify (graph.py:114-133) over the device's own anchors: greedy selection of min(k, N) rows maximising
 * (1 - lambda_div) * scores[i] - lambda_div * max_{j chosen} cos(Y_i, Y_j), first maximum in row order; scores and
 * out_idx in API row order.  *out_count receives the number of rows written. */
int osc_mmr(osc_handle h, const float* scores, int32_t k, float lambda_div, int32_t* out_idx, int32_t* out_count);

/* ---- receipts ------------------------------------------------------------------------------- */
/* deltaH_trace (receipts.py:10-25) on the resident U and U* */
int osc_deltaH(osc_handle h, double* dH);
/* per_node_components (receipts.py:28-60): three arrays of N entries (any may be NULL) */
int osc_receipt_components(osc_handle h, float* coh_drop, float* anchor_pen, float* query_term);
/* null_points (receipts.py:63-83), sparse restatement: per row the first argmax edge, reported iff
 * residual > 0 and z > z_th.  Output arrays hold N entries; *count rows are written in row order. */
int osc_null_points(osc_handle h, float z_th, int32_t* i_out, int32_t* j_out, float* z_out, float* r_out,
                    int32_t* count);

/* components + null points in ONE pass over the edges (what receipt() in "full" detail needs; lattice.py:320-332) */
int osc_receipt_rows(osc_handle h, float z_th, float* coh_drop, float* anchor_pen, float* query_term, int32_t* i_out,
                     int32_t* j_out, float* z_out, float* r_out, int32_t* count);

/* ---- dynamics snapshot: _compute_dynamics (lattice.py:825-927, env-gated OSCILLINK_RECEIPT_DYNAMICS) -------------- */
/* U_prev <- U on the device; call right before osc_settle (replaces the reference's `U_prev = self.U.copy()`,
 * lattice.py:213-216, without moving N x D floats to the host) */
int osc_dynamics_snapshot(osc_handle h);
/* Metrics of the step U_prev -> U_next, all on the device.  U_prev / U_next: host N x D arrays, or NULL for the
 * snapshot taken by osc_dynamics_snapshot / the resident U.
 *   move2_mean, move2_max : mean and max over nodes of ||U_next_i - U_prev_i||^2 (temperature = move2_mean)
 *   step_deltaH           : deltaH_trace(U_prev, U_next, ...) (receipts.py:10-25), chain term at any N
 *   flow_total, top_*     : per directed edge f = max(0, 0.5 lamC A_ij (||Up_i-Up_j||^2 - ||Un_i-Un_j||^2)) with
 *                           Up = U_prev/(sqrt_deg+1e-12), Un likewise; their sum and the top_cap (<= 32; the reference
 *                           keeps 16) largest in the reference's order (flow desc, then (i, j) asc)
 *   radius                : largest BFS hop distance over the lattice graph from the nodes with
 *                           sqrt(move2 + 1e-12) >= 0.1 * max (0 when nothing moved)
 * Any output pointer may be NULL. */
int osc_dynamics(osc_handle h, const float* U_prev_or_null, const float* U_next_or_null, double* move2_mean,
                 float* move2_max, double* step_deltaH, double* flow_total, int32_t top_cap, int32_t* top_i,
                 int32_t* top_j, double* top_flow, int32_t* top_n, int32_t* radius);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Per-kernel HIP-event timing on the handle's own stream.  which: 0 = operator apply inside the CG loop (SpMM, the
 * CG matvec; one sample per apply = all its launches), 1 = fused x/r update, 2 = p update, 3 = kNN GEMM+top-k,
 * 4 = the initial-residual apply of a solve (same gather plus the rhs / r / p streams).  Returns samples and the
 * summed device time since the last reset.  Enabling adds two event records per sample.
 * Diagnostic slots of the source-blocked matvec (k_apply_blocked): with OSC_BLK_STAMP=1 in the environment at creation
 * and profiling on, its loop-form launches run a cycle-stamping instantiation; which = 8..11 then return, in *total_ms,
 * the shader cycles a gathering wave spent (mean over the waves, summed over the stamped launches) in all / in its gather
 * rounds / at the workgroup barrier / in its epilogues, 12..13 the list wave's cycles fetching slot rows / at the barrier,
 * and *launches = the stamped launches.  which = 14: *launches = the kernel shape the last blocked matvec ran with
 * (0 = two 8-wave workgroups per CU, one gather round in flight; 1..6 = one workgroup per CU, four rounds in flight,
 * 8 / 12 / 16 / 20 / 24 / 28 row groups per wave; OSC_BLK_VARIANT forces one), *total_ms = 0.  which = 15: *launches = the
 * pieces the last graph build received its anchors in (0: they were resident before it started; negative: a streamed build
 * gave up on overflowing hit lists and the whole-array build ran instead), *total_ms = 0.  which = 16: *launches = the main
 * sweep of the last build's thresholds-and-hits prefilter (0: another route built the lists; 1: every rank swept every column
 * tile for its row blocks; 2: the symmetric half sweep, ONE per build whatever the world size -- the ranks of a sharded build
 * split its work items), *total_ms = 0. */
int osc_profile_enable(osc_handle h, int32_t on);
int osc_profile_reset(osc_handle h);
int osc_profile_get(osc_handle h, int32_t which, int64_t* launches, double* total_ms);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) ---------------------------------------- */
/* Column-sharded CG: every rank holds the whole graph and the column slab [c0, c1) of the N x D state;
 * alpha/beta are per column (solver.py:22-36) so the only exchange per iteration is one
 * all-reduce(max) of the stop-test residual.  id is an ncclUniqueId (128 bytes) made by rank 0
 * with osc_comm_unique_id and distributed by the caller. */
int osc_comm_unique_id(char id_out[128]);
/* version code of the RCCL this library is linked against (ncclGetVersion, e.g. 22203 = 2.22.3); touches no device.
 * A bench line carries it so that a scaling record says which collective library produced it. */
int osc_comm_backend_version(int32_t* version);
/* An id for the in-process LOOPBACK backend instead: the ranks are threads of ONE process, each with its own handle on
 * the same GPU, and every collective is device-to-device copies between host barriers.  It runs the multi-rank code
 * paths (unequal column slabs, row-block / halo exchanges, the sharded kNN list all-gather, speculative iterations
 * around collectives) at world > 1 on a single MI355X, where RCCL refuses two ranks on one device.  Test backend:
 * every collective blocks the calling thread until all ranks of the group have entered it (60 s limit, then
 * OSC_E_COMM on every rank; OSC_LOOPBACK_TIMEOUT_S overrides). */
int osc_comm_loopback_id(char id_out[128]);
/* Joins the communicator the id names (RCCL or loopback).  Sets this rank's column window (column-sharded CG, the
 * default) or keeps all columns (OSC_SHARD=row: row-sharded CG).  Collective for RCCL ids. */
int osc_comm_init(osc_handle h, const char id[128], int32_t rank, int32_t world);
int osc_comm_shard(osc_handle h, int32_t* c0, int32_t* c1);
/* rank / world (0 / 1 without a communicator), shard_mode 0 = column-sharded CG, 1 = row-sharded; kind_out receives
 * "none", "rccl" or "loopback" */
int osc_comm_info(osc_handle h, int32_t* rank, int32_t* world, int32_t* shard_mode, char* kind_out, int32_t cap);
/* Row-sharded CG only (OSC_SHARD=row): the halo of this rank -- need_rows = off-partition rows its lattice rows (and
 * chain) reference, i.e. the rows of the search direction it receives every iteration; need_rows_max = the largest
 * such count over the ranks; remote_rows = N - own rows; bytes_per_iteration = inbound bytes of one exchange;
 * full_exchange != 0: the lists cover more than 70 % of the remote rows on some rank (unstructured graph), so whole
 * row blocks are exchanged instead (OSC_HALO=lists|full forces either).  Builds the plan on first use per graph:
 * COLLECTIVE (every rank must call it), OSC_E_STATE without a row-sharded communicator. */
int osc_halo_info(osc_handle h, int64_t* need_rows, int64_t* need_rows_max, int64_t* remote_rows,
                  int64_t* bytes_per_iteration, int32_t* full_exchange);
/* Drains the handle's stream, then all-reduces n host doubles in place over the handle's communicator (op 0 = sum,
 * 1 = max); n = 0 is a pure barrier.  Without a communicator: stream drain only.  What a benchmark harness needs for
 * "barrier + max over ranks" without any other distributed runtime. */
int osc_comm_allreduce_f64(osc_handle h, double* vals, int32_t n, int32_t op);

#ifdef __cplusplus
}
#endif
#endif /* OSCILLINK_HIP_H */
