// Operator parameters, apply plans and the CG drivers of liboscillink_hip.so (see osc_internal.hpp).
#include "osc_internal.hpp"

// ---- operators ------------------------------------------------------------------------------------
bool path_active(const L& h) { return h.chain_present && h.lamP > 0.0f; }

OpParams settle_op(const L& h, float dt, int precond) {
  OpParams o{};
  const float lp_op = path_active(h) ? h.lamP : 0.0f;
  o.cs_const = 1.0f + dt * (h.lamG + h.lamC + lp_op);  // X + dt (lamG X + lamC (X - W X) + lamP (X - Wp X))
  o.cs_B = dt * h.lamQ;
  o.cW = dt * h.lamC;
  o.cP = dt * lp_op;
  o.md_const = 1.0f + dt * (h.lamG + (h.chain_present ? h.lamP : 0.0f));  // lattice.py:187-192
  o.md_B = dt * h.lamQ;
  o.precond = precond;
  o.rbU = 1.0f;
  o.rbY = dt * h.lamG;
  o.rbB = dt * h.lamQ;
  return o;
}
OpParams ustar_op(const L& h) {
  OpParams o{};
  const float lp_op = path_active(h) ? h.lamP : 0.0f;
  o.cs_const = h.lamG + h.lamC + lp_op;
  o.cs_B = h.lamQ;
  o.cW = h.lamC;
  o.cP = lp_op;
  o.md_const = h.lamG + (h.chain_present ? h.lamP : 0.0f);  // lattice.py:257-259
  o.md_B = h.lamQ;
  o.precond = 1;
  o.rbU = 0.0f;
  o.rbY = h.lamG;
  o.rbB = h.lamQ;
  return o;
}

// Operator apply, optionally split into column slabs so the gathered operand slab (N x slab x 4 B) stays resident
// in the 256 MB Infinity Cache while its rows are re-read ~deg times (MI355X_MICROARCH.md, Infinity Cache rule).
int32_t auto_slab(const L& h, int32_t ncols) {
  constexpr int32_t kMaxWindow = 2048;  // widest column window one launch covers (8 x 256 floats per row)
  if (h.spmm_slab > 0) return std::min(h.spmm_slab, kMaxWindow);
  if (h.spmm_slab < 0) return std::min(ncols, kMaxWindow);  // OSC_SPMM_SLAB=-1: split only when it must
  // a lattice stored in a local row order gathers from its XCD's L2 whatever the slab: 256 columns (one 1 KB row piece per
  // wave, eight of them in flight: k_spmm's UDEEP variant) ran fastest on 1000 clusters x 100 rows at N = 100k, D = 768
  // (0.69 ms per apply at 64 columns, 0.45 at 128, 0.41 at 256 and 512, 0.44 at 768)
  if (h.reordered && h.spmm_deep && ncols > 256) return 256;
  // keep the gathered slab (N x slab x 4 B) around 50 MB so it and the streams beside it stay inside 256 MB
  const double budget = 56.0 * 1024 * 1024;
  if ((double)h.N * ncols * 4.0 <= 2.0 * budget) return std::min(ncols, kMaxWindow);
  int32_t slab = 64;
  for (int32_t w : {128, 256, 384, 512, 768, 1024, 2048})
    if ((double)h.N * w * 4.0 <= budget) slab = w;
  return slab;
}

// XCD-affine 32-column slabs (SpmmArgs::xs): one launch covers the window; returns the workgroups per XCD to use, 0 = no.
// Pays when the gathered operand is far larger than an XCD's L2 and the graph has no row locality to exploit: each
// XCD then keeps 4 MB / (N x 128 B) of ITS slab in L2 (31 % at N = 100k) instead of 4 MB / (N x 512 B) of a slab all
// eight share.  With fewer than 8 slabs (or a count that is not a multiple of 8) the XCDs pair up: gcd(8, slabs) slab
// groups, the XCDs of a group split the rows.  Needs 128-byte-aligned rows and the slabs in flight (groups x N x 128 B)
// inside the Infinity Cache: measured 1.11 vs 1.26 ms per apply at N = 100k, D = 768; no gain at N = 200k, D = 1536 with
// 8 slabs (205 MB) in flight, 4 % with 4 (xs_groups_for); 36 % slower at N = 1M, D = 384.
// (the counts themselves: host_logic.hpp)
int xs_groups(int32_t ncols, int cap) { return host::xs_groups(ncols, cap); }
int xs_groups_for(const L& h, int32_t ncols) { return host::xs_groups_for(h.N, ncols, h.xs_groups_cap, h.xs_groups_min); }
int blocked_plan(const L& h, bool with_path);
int xs_plan(const L& h, int32_t ncols, int grid) {
  if (grid < 8 || (grid & 7) != 0) return 0;
  const int nb = std::max(1, std::min(grid / 8, h.xs_nb > 0 ? h.xs_nb : 96));
  if (h.spmm_xs == 0) return 0;
  if (h.spmm_xs == 1) return nb;
  if (h.spmm_slab != 0 || (h.ld & 31) != 0 || (h.c0 & 31) != 0) return 0;
  // A lattice stored in BFS order gathers from its XCD's L2 on the general path already (docs/DESIGN_HISTORY.md section 3), so the slab
  // mode is off for it -- except large narrow ones, where the source-blocked matvec on top of the local order wins
  // (round 4, scripts/exp/r04_bfs_blocked_sweep.py, clustered anchors, per settle: 300k x 128 k 16 2.17 -> 1.95 ms, 300k x
  // 256 k 32 6.15 -> 5.0-5.3, 400k x 256 6.35 -> 5.35, 600k x 128 4.79 -> 3.92, 1M x 128 8.15 -> 6.72; at 384 columns a tie
  // (400k 8.06 / 7.98, 1M 20.4 / 20.7), at 200k rows a loss (128 columns: 1.25 -> 1.32)).
  if (h.reordered) return (h.N >= 300000 && ncols <= 256 && blocked_plan(h, false) > 0) ? nb : 0;
  // from N = 32768 on, and from 6144 (16384 until round 3) for windows of >= 256 columns (N = 20000, D = 256: apply 43.5 -> 31.4 us)
  // narrower windows: 32768 rows, but 12288 where the window is whole groups of four slabs (every XCD pair a slab of its
  // own) and 24576 for other windows of >= 128 columns (scripts/exp/xs_narrow_sweep.py, k = 16, per settle: 20000 x 128
  // 246 -> 223 us, 32000 x 128 347 -> 301, 12000 x 128 192 -> 186, 32000 x 192 484 -> 438, 24000 x 192 393 -> 378, 20000 x
  // 192 345 -> 360; 64 and 32 columns: a tie or a loss up to 32000 rows)
  const int narrow_rows = h.xs_min_rows_narrow > 0 ? h.xs_min_rows_narrow : ncols < 128 ? 32768 : (ncols % 128) == 0 ? 12288 : 24576;
  if (h.N < h.xs_min_rows || (h.N < narrow_rows && ncols < 256) || ncols < h.xs_min_cols) return 0;
  // below 16384 rows (round 3: the floor was 16384) a 32-column slab is at most 2 MB -- it sits in its XCD's L2 whole,
  // where the general path spreads N x window over all eight L2s -- which pays once a row has enough gathers: per settle
  // 6500 x 768 k 32 0.520 -> 0.437 ms, 9000 x 1024 k 32 0.925 -> 0.697, 8192 x 1536 k 32 1.32 -> 0.91, 14000 x 256 k 32 0.406
  // -> 0.329, 9000 x 256 k 16 0.236 -> 0.219, 7000 x 512 k 16 0.299 -> 0.280; at k = 8 it loses (14000 x 320: 0.307 -> 0.329)
  if (h.N < 16384 && (double)h.nnz < 10.0 * (double)h.N) return 0;
  const int xg = xs_groups_for(h, ncols);
  if (xg == 0) return 0;
  // beyond the Infinity-Cache budget only the (wide) blocked matvec keeps the mode: host_logic.hpp, xs_groups_for
  if (h.N > host::kXsBudgetRows && blocked_plan(h, false) == 0) return 0;
  // Two slab groups (262k < N <= 524k: four XCDs share a slab) pay only under the blocked matvec -- measured in round 3
  // against the general path: 300k x 768 k 32 25.96 -> 22.33 ms per settle, 400k x 512 k 32 22.70 -> 19.10, 300k x 768 k 64
  // 43.1 -> 36.3, 500k x 384 k 16 a tie; the plain slab apply at two groups loses (config 5's shape: 57.1 vs 56.4 ms) and one
  // group loses either way (700k x 384: 22.8 -> 24.5, config 4: 32.3 -> 34.4)
  if (xg < 4 && xg != xs_groups(ncols, h.xs_groups_cap) && blocked_plan(h, false) == 0) return 0;
  return nb;
}

// workgroups per XCD a shape of the blocked apply gets resident
int blocked_resident(const L& h, int shape) {
  if (h.blk_resident[shape] < 0) {
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, h.device));
    h.blk_resident[shape] = blocked_resident_per_cu(shape) * std::max(1, prop.multiProcessorCount / 8);
  }
  return h.blk_resident[shape];
}

// Kernel shape of the blocked matvec for a window cut into xg slab groups (cg_kernels.hip: kBlkShapes).  The wide shapes
// (one workgroup per CU, four gather rounds in flight, no tests in the rounds) carry their group count as a template
// constant -- the smallest that holds the lattice's groups is used -- and are taken from 96 000 rows on, where they win
// at every width measured except one slab per XCD below 150k rows; below 96k rows they are within +-2 % of shape 0 with
// single wins and losses of 5-7 % either way, so shape 0 stays there.
int blocked_shape_for(const L& h, int xg, int grid) {
  if (h.blk_variant >= 0) return h.blk_variant;
  const int wide_last = blocked_variants() - 1;
  const host::BlockedGeom g = host::blocked_geometry(h.N, xg, grid, blocked_resident(h, wide_last), blocked_groups_max(wide_last),
                                                     blocked_gather_waves(wide_last));
  // Measured against shape 0 (profiles/r05_blk_shape_sweep.txt, per AP launch, exact-fit group counts): 20k x 768 -7.5 %, 20k x
  // 128 k 16 +5.8 %, 30k-80k x 768 -0.8 ... +2.7 %, 100k x 768 -4.6 %, 100k x 384 k 16 -4.8 %, 100k x 1024 k 48 -4.0 %, 100k x
  // 96 (rank 0 of 8's window of config 3) -10.9 %, 100k x 192 -2.7 %, 160k x 768 -10.9 %, 200k x 768 -9.6 %, 200k x 64 -13.0 %,
  // 260k x 512 -15.0 %, 400k x 384 k 16 -7.1 %; one slab per XCD: 100k x 64 k 16 +5.0 %, 100k x 128 k 16 +1.4 %, 130k x 256
  // +1.4 ... +3.6 % -- there the wide shapes wait for N = 150k.
  const int64_t min_rows = h.blk_wide_min_rows > 0 ? h.blk_wide_min_rows : 96000;
  const int slabs_per_group = ((h.c1 - h.c0 + 31) / 32 + xg - 1) / std::max(1, xg);
  if (h.N < min_rows || (h.blk_wide_min_rows <= 0 && slabs_per_group < 2 && h.N < 150000)) return 0;
  for (int v = 1; v <= wide_last; ++v)
    if (g.groups <= blocked_groups_max(v)) return v;
  return 0;
}

// Source blocks of the blocked CG matvec (k_apply_blocked): 0 = use the plain apply.
int blocked_plan(const L& h, bool with_path) {
  if (h.spmm_blocked == 0 || (with_path && (h.prows < 1 || h.prows > OSC_CHAIN_FIX_MAX_ROWS)) || (int64_t)h.N * h.width >= ((int64_t)1 << 28) || h.N >= ((int64_t)1 << 24) ||
      (int64_t)h.N * h.ld * 4 >= ((int64_t)1 << 32))
    return 0;
  if (h.spmm_blocked > 0) return std::min(h.spmm_blocked, OSC_MAX_SRC_BLOCKS);
  // block count from the mean degree and the lattice size (host_logic.hpp: blocked_edges_per_block)
  // overrides the edges a row should have per block
  const double mean_deg = h.N > 0 ? (double)h.nnz / (double)h.N : 0.0;
  // (a lattice in BFS order: 2.2 edges per block -- x4 of x2 / x3 / x4 / x6 / x8 at mean degree 8.3, x8 of x6 / x8 / x12 at 20.2)
  const int ncols = h.c1 - h.c0, xg0 = xs_groups_for(h, ncols);
  const bool wide = blocked_shape_for(h, xg0 > 0 ? xg0 : xs_groups(ncols, h.xs_groups_cap), cg_grid(h)) > 0;
  const double e = h.blk_edges > 0.0 ? h.blk_edges
                   : h.reordered     ? 2.2
                   : wide            ? host::blocked_edges_per_block_wide(h.N)
                                     : host::blocked_edges_per_block(h.N);
  const int nb = host::blocked_block_count(mean_deg, e, OSC_MAX_SRC_BLOCKS);
  if (h.spmm_blocked == -2) return nb;  // "whenever possible" (experiments)
  // ... and wherever the XCD-affine slab mode itself runs from a 2 MiB slab (N = 16384) on.  Measured against the plain
  // apply (k = 32 unless noted): N = 20k x 768 -7 %, 35k x 768 -26 %, 40k x 256 (k 8) -25 %, 50k x 512 -30 %, 65k x 256
  // (k 16) -30 %, 60k x 1024 (k 24) -29 %, 80k x 768 -39 %, 100k x 768 -39 % (k 16, D 384: -33 %; k 48: -47 %; k 64:
  // -45 %), 100k x 128 (k 16) -35 %, 110k x 768 -40 %, 130k x 256 -43 %; round 3: 160k x 768 -31 %, 200k x 768 -37 %
  // (k 64: -46 %), 260k x 768 -22 % (k 64: -37 %).
  const double slab = (double)h.N * 128.0;
  if (slab < h.blk_mb * 1024.0 * 1024.0) return 0;
  // narrow windows of small lattices: the plain slab apply is ahead (round 4 shape sweep: 16384 x 128 k 16 0.205 vs 0.221 ms per
  // settle; from 20000 rows on a tie or a win)
  if (h.N < 20000 && h.c1 - h.c0 <= 128) return 0;
  return nb;
}

BlockedView blocked_view(L& h, int nb) {
  if (h.blk_nb != nb) {
    DevBuf<unsigned> cnt;
    cnt.alloc(1);
    HIP_CHECK(hipMemsetAsync(cnt.p, 0, 4, h.stream));
    launch_blocked_count(h.ell_col.p, h.deg.p, h.width, (int32_t)h.N, nb, cnt.p, h.stream);
    unsigned over = 0;
    HIP_CHECK(hipMemcpyAsync(&over, cnt.p, 4, hipMemcpyDeviceToHost, h.stream));
    sync(h);
    // the apply's list wave copies whole row groups: up to 8 x gather-waves slot rows past the lattice's end
    // (host_logic.hpp: blocked_list_extent <= N - 1 + 8 x gather waves, swept in tests/host_logic)
    constexpr size_t kPadRows = 8192;
    for (int v = 0; v < blocked_variants(); ++v)
      if ((size_t)blocked_gather_waves(v) * 8 > kPadRows) throw std::runtime_error("blocked graph copy: padding too small");
    const size_t nslots = (size_t)nb * h.N * OSC_BLK_SLOTS, npad = kPadRows * OSC_BLK_SLOTS;
    h.blk_slots.alloc(nslots + npad);
    HIP_CHECK(hipMemsetAsync(h.blk_slots.p + nslots, 0, npad * sizeof(int2), h.stream));  // {row 0, 0.0f}
    h.blk_over.alloc((size_t)over + 1);
    h.blk_rest.alloc((size_t)h.N);
    HIP_CHECK(hipMemsetAsync(cnt.p, 0, 4, h.stream));
    launch_blocked_fill(h.ell_col.p, h.ell_w.p, h.deg.p, h.width, (int32_t)h.N, nb, h.blk_slots.p, h.blk_rest.p, h.blk_over.p, cnt.p,
                        h.stream);
    sync(h);  // cnt goes out of scope
    h.blk_nb = nb;
  }
  BlockedView v{};
  v.slots = h.blk_slots.p;
  v.rest = h.blk_rest.p;
  v.over = h.blk_over.p;
  v.nb = nb;
  return v;
}

void spmm_slabbed(L& h, int mode, SpmmArgs sa, int grid, int iter) {
  const int32_t c0 = sa.c0, c1 = sa.c1;
  sa.deep = (h.reordered && h.spmm_deep) ? 1 : 0;
  ProfScope ps(h, mode == SPMM_INIT ? 4 : 0, iter);  // slot 0: AP applies (the CG matvec); slot 4: the INIT apply
  if (const int nb = xs_plan(h, c1 - c0, grid)) {
    // workgroups per XCD: 3 per CU when the operand is row-major (2: 1.37, 4: 1.15 ms vs 1.11), 4 per CU when it is
    // slab-major (3: 1.09, 4: 1.05 ms)
    sa.xs = (h.xs_nb <= 0 && sa.xblk != 0) ? std::min(grid / 8, 128) : nb;
    const int xg = xs_groups_for(h, c1 - c0);
    sa.xs_groups = xg > 0 ? xg : xs_groups(c1 - c0, h.xs_groups_cap);  // forced mode: natural count
    launch_spmm(mode, sa, grid, h.stream);
    return;
  }
  const int32_t slab = auto_slab(h, c1 - c0);
  for (int32_t s0 = c0; s0 < c1; s0 += slab) {
    sa.c0 = s0;
    sa.c1 = std::min(c1, s0 + slab);
    launch_spmm(mode, sa, grid, h.stream);
  }
}

// elementwise CG kernels cover at most 2048 columns per launch: wider states run as several column windows
template <typename F>
void for_windows(UpdateArgs ua, F&& launch) {
  const int32_t c0 = ua.c0, c1 = ua.c1;
  for (int32_t s0 = c0; s0 < c1; s0 += 2048) {
    ua.c0 = s0;
    ua.c1 = std::min(c1, s0 + 2048);
    launch(ua);
  }
}



// cg_solve (solver.py:6-37) on the device; returns once the last residual is out (what may still be queued then touches
// scratch arrays only, and later calls are ordered behind it by the stream).
// The host enqueues iteration it+1 before it reads iteration it's residual.  On one GPU every kernel of a speculative
// iteration carries a gate (residual of the previous iteration, tol) and is a no-op once the CG has converged; under
// a communicator it carries none and writes scratch arrays only (the x update of an iteration is applied by its
// successor's p update or by the host's order, never speculatively).  Either way the reference's "stop before the
// beta/p update" semantics hold exactly while the stream never drains between iterations.
// Small lattices: the whole solve in ONE launch with the state in LDS (small_kernels.hip).  Returns false when the
// lattice does not fit that path (or its barrier timed out) and the general path must run.
bool run_cg_small(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol,
                  CgResult& out) {
  if (!h.small_path || h.comm != nullptr || b.c0 != 0 || b.c1 != h.dcols || b.ld != h.dcols || max_iters > 4096) return false;
  const int C = small_pick_cols((int32_t)h.N, b.ld);
  if (C <= 0) return false;
  const size_t nslots = (size_t)max_iters + 2;
  const size_t nctl = 2 * nslots + 2;  // [residual slots | arrival counters | status | finish counter]
  ensure_ctrl(h, nctl);
  uint32_t* ctl = ctrl_segment(h, nctl);
  // the kernel's last workgroup publishes residuals + a "done" word into host-mapped memory and the host polls that
  // word
  const bool polled = h.mapped_residual;
  constexpr uint32_t kPending = 0xFFFFFFFFu;
  volatile uint32_t* host_words = reinterpret_cast<volatile uint32_t*>(h.res_host);
  if (polled) host_words[nslots] = kPending;
  SmallArgs a{};
  if (!h.ell_t_ready) {
    h.ell_col_t.alloc((size_t)h.N * h.width);
    h.ell_w_t.alloc((size_t)h.N * h.width);
    launch_transpose_ell(h.ell_col.p, h.ell_w.p, (int32_t)h.N, h.width, h.ell_col_t.p, h.ell_w_t.p, h.stream);
    h.ell_t_ready = true;
  }
  a.g = graph_view(h, with_path);
  a.col_t = h.ell_col_t.p;
  a.w_t = h.ell_w_t.p;
  a.op = op;
  float* xout = b.X;
  if (b.X == b.x0 || b.X == b.rhsU || b.X == b.rhsY) {  // never hand the one-launch kernel an aliased output
    if (!b.Xalt) return false;
    xout = b.Xalt;
  }
  a.x0 = b.x0;
  a.X = xout;
  a.U = b.rhsU;
  a.Y = b.rhsY;
  a.B = b.B;
  a.psi = b.psi;
  a.res_bits = ctl;
  a.arrive = ctl + nslots;
  a.status = ctl + 2 * nslots;
  a.finish = ctl + 2 * nslots + 1;
  a.host_words = polled ? reinterpret_cast<uint32_t*>(h.res_host_dev) : nullptr;
  a.N = (int32_t)h.N;
  a.ld = b.ld;
  a.max_iters = max_iters;
  a.tol = tol;
  launch_settle_small(a, C, h.stream);
  if (polled) {
    const double t_start = now_ms();
    for (uint64_t spin = 1; host_words[nslots] == kPending; ++spin) {
      if ((spin & 0x3FFF) == 0) {  // every ~16k polls: has the stream died or drained without the word?
        const hipError_t q = hipStreamQuery(h.stream);
        if (q != hipSuccess && q != hipErrorNotReady) hip_check(q, "hipStreamQuery (one-launch solve)", __FILE__, __LINE__);
        if (q == hipSuccess && host_words[nslots] == kPending) throw HipError("one-launch solve finished without its done word");
        if (now_ms() - t_start > 120000.0) throw HipError("timeout waiting for the one-launch solve");
      }
      __builtin_ia32_pause();
    }
    if (host_words[nslots] != 0u) {
      sync(h);       // (the kernel's other workgroups are on their way out)
      return false;  // barrier timeout (GPU shared with other persistent work): take the general path
    }
  } else {
    HIP_CHECK(hipMemcpyAsync(h.res_host, ctl, nctl * 4, hipMemcpyDeviceToHost, h.stream));
    sync(h);
    uint32_t st;
    std::memcpy(&st, h.res_host + 2 * nslots, 4);
    if (st != 0) return false;  // barrier timeout (GPU shared with other persistent work): take the general path
  }
  h.history.clear();
  out = CgResult{max_iters, 0.f, xout};
  for (int it = 1; it <= max_iters; ++it) {
    const float res = h.res_host[it];
    h.history.push_back(res);
    out.res = res;
    if ((double)res <= (double)tol) {
      out.iters = it;
      break;
    }
  }
  h.small_solves += 1;
  return true;
}

CgResult run_cg_rows(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol);
bool row_mode(const L& h);

// Arguments of the source-blocked matvec (k_apply_blocked) for the handle's whole column window: X = slab-major input, OUT =
// row-major output, column sums of X . OUT into h.part0 (grid rows, + the chain fix-up's chunks behind them).  ba.nb stays
// 0 where the plan says the plain apply serves this lattice (blocked_plan).
void blocked_setup(L& h, const OpParams& op, const float* X, float* OUT, const float* B, int32_t ld, bool with_path, int grid, BlkArgs& ba,
                   ChainFixArgs& cf, int& blk_shape) {
  const int nb = blocked_plan(h, with_path);
  if (nb == 0) return;
  const BlockedView bv = blocked_view(h, nb);
  ba.X = X;
  ba.OUT = OUT;
  ba.B = B;
  ba.part = h.part0.p;
  ba.slots = bv.slots;
  ba.rest = bv.rest;
  ba.over = bv.over;
  ba.cs_const = op.cs_const;
  ba.cs_B = op.cs_B;
  ba.cW = op.cW;
  ba.N = (int32_t)h.N;
  ba.ld = ld;
  ba.c0 = h.c0;
  ba.c1 = h.c1;
  ba.nb = nb;
  // workgroups per XCD (what is resident at once), slab groups, row groups per wave, destination slices
  const int xg0 = xs_groups_for(h, h.c1 - h.c0), xg = xg0 > 0 ? xg0 : xs_groups(h.c1 - h.c0, h.xs_groups_cap);
  blk_shape = blocked_shape_for(h, xg, grid);
  const host::BlockedGeom geom = host::blocked_geometry(h.N, xg, grid, blocked_resident(h, blk_shape), blocked_groups_max(blk_shape),
                                                       blocked_gather_waves(blk_shape));
  ba.xs = geom.xs;
  ba.xs_groups = geom.xs_groups;
  ba.slices = geom.slices;
  ba.groups = geom.groups;
  if (with_path && op.cP != 0.f) {  // the chain prior's few rows: a small launch behind every blocked apply
    cf.X = X;
    cf.OUT = OUT;
    cf.part = h.part0.p;
    cf.prow = h.prow.p;
    cf.pcol = h.pcol.p;
    cf.pw = h.pw.p;
    cf.pdeg = h.pdeg.p;
    cf.cP = op.cP;
    cf.prows = h.prows;
    cf.pwidth = h.pwidth;
    cf.N = (int32_t)h.N;
    cf.ld = ld;
    cf.c0 = h.c0;
    cf.c1 = h.c1;
    cf.part_row0 = grid;
    cf.chunks = chain_fix_chunks(h.prows);
  }
}

// x . (op x) summed per column into h.part0 for a ROW-major x (N x ld, the handle's whole window) through the blocked matvec:
// x -> slab-major (scratch_slab), one launch (+ the chain fix-up), op x -> scratch_out.  Returns the rows of partial sums
// h.part0 holds, 0 where the blocked matvec does not serve this lattice (the caller takes the plain apply's DOT form).
int blocked_quad_form(L& h, const OpParams& op, const float* x_rows, float* scratch_slab, float* scratch_out, bool with_path,
                      const float* x_sub) {
  const int grid = cg_grid(h);
  if (!(h.p_blocked && xs_plan(h, h.c1 - h.c0, grid) > 0 && (h.ld & 31) == 0 && (h.c0 & 31) == 0)) return 0;
  BlkArgs ba{};
  ChainFixArgs cf{};
  int blk_shape = 0;
  blocked_setup(h, op, scratch_slab, scratch_out, h.B.p, h.ld, with_path, grid, ba, cf, blk_shape);
  if (ba.nb == 0) return 0;
  launch_rows_to_slab(x_rows, scratch_slab, h.N, h.ld, h.c0, h.c1, grid, h.stream, x_sub);
  ba.gate = nullptr;
  launch_apply_blocked(ba, grid, h.stream, nullptr, blk_shape);
  h.blk_applies += 1;
  if (cf.chunks > 0) {
    cf.gate = nullptr;
    launch_chain_fix(cf, h.stream);
  }
  return grid + cf.chunks;
}

CgResult run_cg(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol) {
  if (row_mode(h) && b.ld == h.ld) return run_cg_rows(h, op, b, with_path, max_iters, tol);
  {
    CgResult one{};
    if (run_cg_small(h, op, b, with_path, max_iters, tol, one)) return one;
  }
  const int grid = cg_grid(h);
  const size_t nslots = (size_t)max_iters + 2;
  ensure_ctrl(h, nslots);
  uint32_t* const res_slots = ctrl_segment(h, 2 * nslots);  // zeroed: [residual per iteration | arrival counter per iteration]
  uint32_t* done_ctr = res_slots + nslots;
  // Single GPU: the last workgroup of each iteration's beta reduction writes the residual into host-mapped memory and
  // the host polls that word (no 4-byte copy, event record and event wait per iteration).  Under a communicator the
  // residual first goes through the all-reduce (below; OSC_COMM_OVERLAP=0: in the solve's stream,
  // read back by copy + event).
  const bool mapped = h.comm == nullptr && h.mapped_residual;
  // Sharded (column windows): the stop test needs max over the ranks of the residual -- a 4-byte all-reduce per iteration,
  // tens of microseconds of latency on xGMI next to ~180 us of kernels per iteration in an 8-rank window of config 3.
  // With the x update deferred (below) a speculative iteration writes scratch arrays only (r, p, Ap, alpha, beta), so it
  // needs no gate and the solve's stream never waits for the all-reduce: that goes to a second stream behind an event
  // per iteration, followed by a one-thread kernel that publishes the reduced word into the host-mapped slot the host
  // polls, as on one GPU.  The host alone decides when to stop; a wrong guess of the last iteration costs one iteration
  // of device time instead of five gated-off launches.
  const bool xdefer = h.x_defer;
  // What it costs (one-rank RCCL communicator, all-reduce latency ~0: docs/DESIGN_HISTORY.md section 6): ~17 us once per solve for the
  // second stream's hand-over at the last iteration, and the expected last iteration's own form (an ungated speculative
  // iteration must not touch x): 47 us at 768 columns, 6 at 96.  What it saves: every all-reduce latency but the last.
  // Hence by default from four ranks on (narrow windows, 15-30 us per all-reduce); OSC_COMM_OVERLAP=1 / 0 force it.
  const bool want_overlap = h.comm_overlap == 1 || (h.comm_overlap < 0 && h.world >= 4);
  const bool overlap = h.comm != nullptr && want_overlap && h.mapped_residual && xdefer;
  if (overlap) {
    if (!h.comm_stream) h.comm_stream = acquire_stream(h.device);
    while (h.step_events.size() < nslots) {
      hipEvent_t e;
      HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h.step_events.push_back(e);
    }
    h.comm_stream_busy = true;  // from here on (also if the solve is abandoned half way): drained before the slots are reused
  }
  const bool polled = mapped || overlap;
  constexpr uint32_t kPending = 0xFFFFFFFFu;  // never a residual (those are sqrt(...) >= 0 or a canonical NaN)
  if (polled)
    for (size_t i = 0; i < nslots; ++i) reinterpret_cast<volatile uint32_t*>(h.res_host)[i] = kPending;
  const float* res_dev = reinterpret_cast<const float*>(res_slots);
  SpmmArgs sa{};
  sa.g = graph_view(h, with_path);
  sa.op = op;
  sa.B = b.B;
  sa.psi = b.psi;
  sa.N = h.N;
  sa.ld = b.ld;
  sa.c0 = b.c0;
  sa.c1 = b.c1;
  sa.part = h.part0.p;
  // r = b - A x0 ; z ; p ; rz
  sa.X = b.x0;
  sa.OUT = b.X;
  sa.R = b.R;
  sa.P = b.P;
  sa.U = b.rhsU;
  sa.Y = b.rhsY;
  sa.gate = nullptr;
  // slab-major search direction: only where the XCD-affine slab apply runs (its gathers then read contiguous slabs)
  // and P is private to this solve (N x ld floats either way; needs whole 32-column slabs inside the pitch)
  const bool pblk = h.p_blocked && xs_plan(h, b.c1 - b.c0, grid) > 0 && (b.ld & 31) == 0 && (b.c0 & 31) == 0 &&
                    b.ld == h.ld;
  sa.pblk = pblk ? h.N : 0;
  // source-blocked CG matvec (k_apply_blocked) where the slab an XCD gathers from is far larger than its L2
  BlkArgs ba{};
  ChainFixArgs cf{};
  int blk_shape = 0;
  if (pblk && b.c0 == h.c0 && b.c1 == h.c1) blocked_setup(h, op, b.P, b.AP, b.B, b.ld, with_path, grid, ba, cf, blk_shape);

  // (an inertia start hands over x0 IN the AP array, which the blocked matvec would overwrite with A x0 before
  // init_finish has read x0: such a solve keeps the gathering INIT kernel, which reads x0 completely first)
  int init_part_rows = grid;  // rows of r . z partials the INIT pass leaves (+ the chain fix-up's behind the fused pass)
  float* Pbuf = b.P;   // search direction / operator output: the fused INIT pass below leaves p in the AP array and
  float* APbuf = b.AP;  // swaps the two for the rest of the solve
  // (the rhs rows the fused pass can take besides x0 itself: one -- the state term must be x0 or absent, and if y is a
  // third array the solution array must be x0)
  const bool fuse_u = b.rhsU == b.x0 || op.rbU == 0.f;
  const bool fuse_y = b.rhsY == b.x0 || b.X == b.x0;
  if (ba.nb > 0 && h.blk_init && h.blk_init_fused && fuse_u && fuse_y && b.x0 != b.AP) {
    // r = b - A x0 INSIDE the blocked matvec (the in-place warm-started settle: x0 is also the rhs state term and the
    // solution array; the U* solve: x0 is Y, no state term): x0 -> slab-major (into P), then one launch gathers A x0 and
    // leaves r, z (slab-major, in the AP array), x0 in the solution array and the r . z column sums
    ProfScope ps(h, 4, 0);
    launch_rows_to_slab(b.x0, b.P, h.N, b.ld, b.c0, b.c1, grid, h.stream);
    BlkInit bi{};
    bi.Y = b.rhsY == b.x0 ? nullptr : b.rhsY;
    bi.Xcopy = b.X == b.x0 ? nullptr : b.X;
    bi.R = b.R;
    bi.Z = b.AP;
    bi.psi = b.psi;
    bi.rbU = b.rhsU == b.x0 ? op.rbU : 0.f;
    bi.rbY = op.rbY;
    bi.rbB = op.rbB;
    bi.md_B = op.precond ? op.md_B : 0.f;
    bi.md_const = op.precond ? op.md_const : 1.f;
    ba.gate = nullptr;
    ba.OUT = nullptr;
    launch_apply_blocked(ba, grid, h.stream, &bi, blk_shape);
    if (cf.chunks > 0) {  // the chain prior's rows: their r, z and r . z still lack the chain term
      ChainFixArgs ci = cf;
      ci.gate = nullptr;
      ci.initR = b.R;
      ci.initZ = b.AP;
      ci.B = b.B;
      ci.md_B = bi.md_B;
      ci.md_const = bi.md_const;
      launch_chain_fix(ci, h.stream);
      init_part_rows = grid + cf.chunks;
    }
    std::swap(Pbuf, APbuf);
    ba.X = Pbuf;
    ba.OUT = APbuf;
    cf.X = Pbuf;
    cf.OUT = APbuf;
  } else if (ba.nb > 0 && h.blk_init && b.x0 != b.AP) {
    // r = b - A x0 around the blocked matvec: x0 -> slab-major (into P), A x0 -> AP, then r, z, p = z, r . z
    ProfScope ps(h, 4, 0);
    launch_rows_to_slab(b.x0, b.P, h.N, b.ld, b.c0, b.c1, grid, h.stream);
    ba.gate = nullptr;
    launch_apply_blocked(ba, grid, h.stream, nullptr, blk_shape);
    if (cf.chunks > 0) {
      cf.gate = nullptr;
      launch_chain_fix(cf, h.stream);
    }
    InitFinishArgs fa{};
    fa.AP = b.AP;
    fa.X0 = b.x0;
    fa.X = b.X;
    fa.R = b.R;
    fa.P = b.P;
    fa.U = b.rhsU;
    fa.Y = b.rhsY;
    fa.B = b.B;
    fa.psi = b.psi;
    fa.part = h.part0.p;
    fa.op = op;
    fa.N = h.N;
    fa.pblk = h.N;
    fa.ld = b.ld;
    for (int32_t s0 = b.c0; s0 < b.c1; s0 += 2048) {
      fa.c0 = s0;
      fa.c1 = std::min(b.c1, s0 + 2048);
      launch_init_finish(fa, grid, h.stream);
    }
  } else {
    spmm_slabbed(h, SPMM_INIT, sa, grid);
  }
  launch_reduce_init(h.part0.p, init_part_rows, b.ld, b.c0, b.c1, h.rz.p, h.stream);
  UpdateArgs ua{};
  ua.pblk = pblk ? h.N : 0;
  ua.temporal = 5.0 * (double)h.N * (double)(b.c1 - b.c0) * 4.0 <= h.temporal_mb * 1048576.0;
  ua.X = b.X;
  ua.R = b.R;
  ua.P = Pbuf;
  ua.AP = APbuf;
  ua.B = b.B;
  ua.alpha = h.alpha.p;
  ua.beta = h.beta.p;
  ua.part_rr = h.part0.p;
  ua.part_rz = h.part1.p;
  ua.op = op;
  ua.N = h.N;
  ua.ld = b.ld;
  ua.c0 = b.c0;
  ua.c1 = b.c1;
  sa.X = Pbuf;
  sa.OUT = APbuf;
  sa.xblk = pblk ? h.N : 0;
  sa.pblk = 0;
  h.blk_last = ba.nb;
  h.blk_shape_last = ba.nb > 0 ? blk_shape : 0;
  // Deferred x update: iteration it's x += alpha p is applied by iteration it + 1's p update, which reads p anyway (x, r,
  // p in / x, p out there, r, Ap in / r out in the x-r kernel: 8 array passes per iteration instead of 9), or by
  // finish_x behind an iteration that has no successor enqueued.  The iteration expected to be the last (the count of
  // the handle's previous solve of this kind, or max_iters) takes k_update_xr's "last" form instead: x finished next
  // to the r update, the new r not stored (five passes instead of three there and three in finish_x).  Which launch
  // carries which update is decided by host::CgXSchedule (host_logic.hpp; swept against a model of the device's
  // gating on the CPU box, tests/host_logic/sweep_host_logic.cpp).
  const int stop_guess = h.predicted_iters[b.kind];
  host::CgXSchedule xs;
  xs.xdefer = xdefer;
  xs.last_form = h.x_last_form;
  xs.ungated = overlap;
  xs.stop_guess = stop_guess;
  xs.max_iters = max_iters;
  auto finish_x = [&](int it) {
    ua.gate = nullptr;
    ua.xmode = OSC_XMODE_XR_SKIPS_X | OSC_XMODE_P_APPLIES_X;
    for_windows(ua, [&](const UpdateArgs& w) { launch_update_x(w, grid, h.stream); });
    xs.finished(it);
  };
  auto enqueue_iter = [&](int it, bool speculative) {  // everything of iteration `it` up to its residual, gated on iteration it-1
    // (overlap: no gates -- an iteration writes scratch arrays only until the host has seen its predecessor unconverged)
    const Gate g{it > 1 && !overlap ? res_dev + (it - 1) : nullptr, tol};
    sa.gate = g.p;
    sa.gate_tol = tol;
    ua.gate = g.p;
    ua.gate_tol = tol;
    const host::CgXSchedule::IterForm form = xs.enqueue(it, speculative);
    if (it > 1) {
      ProfScope ps(h, 2, it);
      // p = z + beta p (solver.py:32-36), and iteration it - 1's x += alpha p (solver.py:27) with the p it replaces
      ua.xmode = form.p_applies_x ? OSC_XMODE_P_APPLIES_X : 0;
      for_windows(ua, [&](const UpdateArgs& w) { launch_update_p(w, grid, h.stream); });
    }
    if (ba.nb > 0) {  // Ap and column sums of p.Ap
      ProfScope ps(h, 0, it);
      ba.gate = g.p;
      ba.gate_tol = tol;
      unsigned long long* stamps = nullptr;
      if (h.blk_stamp && h.prof_on) {  // diagnostic: per-wave cycle counters of where the launch's time goes
        const size_t words = (size_t)grid * (size_t)(blocked_gather_waves(blk_shape) + 1) * 4;
        if (h.blk_stamps.n != words || h.blk_stamp_grid != grid) {
          h.blk_stamps.alloc(words);
          HIP_CHECK(hipMemsetAsync(h.blk_stamps.p, 0, words * 8, h.stream));
          h.blk_stamp_launches = 0;
          h.blk_stamp_grid = grid;
        }
        stamps = h.blk_stamps.p;
        h.blk_stamp_launches += 1;
      }
      launch_apply_blocked(ba, grid, h.stream, nullptr, blk_shape, stamps);
      if (cf.chunks > 0) {
        cf.gate = g.p;
        cf.gate_tol = tol;
        launch_chain_fix(cf, h.stream);
      }
      h.blk_applies += 1;
    } else {
      spmm_slabbed(h, SPMM_AP, sa, grid, it);
    }
    launch_reduce_alpha(h.part0.p, grid + (ba.nb > 0 ? cf.chunks : 0), b.ld, b.c0, b.c1, h.rz.p, h.alpha.p, g, h.stream);
    {
      ProfScope ps(h, 1, it);
      ua.xmode = form.xr == host::CgXSchedule::XR_LAST ? OSC_XMODE_XR_LAST : form.xr == host::CgXSchedule::XR_SKIPS_X ? OSC_XMODE_XR_SKIPS_X : 0;
      for_windows(ua, [&](const UpdateArgs& w) { launch_update_xr(w, grid, h.stream); });
    }
    if (mapped) {
      launch_reduce_beta(h.part0.p, h.part1.p, grid, b.ld, b.c0, b.c1, h.rz.p, h.beta.p, res_slots + it, g, h.stream,
                         done_ctr + it, h.res_host_dev + it);
      return;
    }
    launch_reduce_beta(h.part0.p, h.part1.p, grid, b.ld, b.c0, b.c1, h.rz.p, h.beta.p, res_slots + it, g, h.stream);
    if (overlap) {  // max over the shards (solver.py:29) and its way to the host, beside the next iteration's first kernels
      HIP_CHECK(hipEventRecord(h.step_events[(size_t)it], h.stream));
      HIP_CHECK(hipStreamWaitEvent(h.comm_stream, h.step_events[(size_t)it], 0));
      h.comm->allreduce(res_slots + it, 1, COMM_F32, COMM_MAX, h.comm_stream);
      launch_publish_word(res_slots + it, reinterpret_cast<uint32_t*>(h.res_host_dev + it), h.comm_stream);
      return;
    }
    // column-sharded: the stop test is the max over all shards (solver.py:29)
    if (h.comm) h.comm->allreduce(res_slots + it, 1, COMM_F32, COMM_MAX, h.stream);
    HIP_CHECK(hipMemcpyAsync(h.res_host + it, res_slots + it, 4, hipMemcpyDeviceToHost, h.stream));
    HIP_CHECK(hipEventRecord(h.iter_events[(size_t)it], h.stream));
  };
  auto wait_residual = [&](int it) -> float {
    if (!polled) {
      HIP_CHECK(hipEventSynchronize(h.iter_events[(size_t)it]));
      return h.res_host[it];
    }
    volatile uint32_t* slot = reinterpret_cast<volatile uint32_t*>(h.res_host) + it;
    const double t_start = now_ms();
    for (uint64_t spin = 1;; ++spin) {
      const uint32_t bits = *slot;
      if (bits != kPending) {
        float v;
        std::memcpy(&v, &bits, 4);
        return v;
      }
      if ((spin & 0x3FFF) == 0) {  // every ~16k polls: has the stream died or drained without publishing?
        hipError_t q = hipStreamQuery(h.stream);
        if (q == hipSuccess && overlap) q = hipStreamQuery(h.comm_stream);  // the word comes out of the second stream
        if (q != hipSuccess && q != hipErrorNotReady) hip_check(q, "hipStreamQuery (CG residual wait)", __FILE__, __LINE__);
        if (q == hipSuccess && *slot == kPending) throw HipError("CG iteration finished without publishing its residual");
        if (now_ms() - t_start > 120000.0) throw HipError("timeout waiting for a CG residual");
      }
      __builtin_ia32_pause();
    }
  };

  h.history.clear();
  CgResult out{max_iters, 0.f, b.X};
  const size_t prof_mark = h.prof_pending.size();
  // Iteration it + 1 is enqueued before iteration it's residual is read -- except behind the iteration the previous
  // solve of this handle converged in: repeated settles of one lattice take the same count, and the five gated-off
  // launches of a needless speculative iteration cost ~22 us (8 % of a settle at N = 20000, D = 128; ungated under a
  // communicator: a whole iteration).  A wrong guess the other way costs one host round trip: the iteration is then
  // enqueued after its predecessor's residual has been read.
  // (Every rank of a sharded solve sees the same residuals, hence takes the same decisions.)
  int enqueued = 1;
  enqueue_iter(1, false);
  for (int it = 1; it <= max_iters; ++it) {
    if (it < max_iters && it != stop_guess && enqueued == it) {
      enqueue_iter(++enqueued, true);  // speculative: no-ops if `it` converged (overlap: ungated, scratch arrays only)
    } else if (xs.finish_before_wait(it)) {
      // nothing is enqueued behind this iteration for now (the expected last one): its x update goes out at once.  The
      // host has seen iteration it - 1 unconverged, so iteration `it` is a real one whatever its residual will say.
      finish_x(it);
    }
    const float res = wait_residual(it);
    h.history.push_back(res);
    out.res = res;
    if ((double)res <= (double)tol) {
      out.iters = it;
      break;
    }
    if (it < max_iters && enqueued == it) {  // the guess was wrong: go on
      if (xs.restore_r(it)) {  // ... from the r this iteration computed but did not keep
        ua.gate = nullptr;
        ua.xmode = OSC_XMODE_XR_SKIPS_X;
        for_windows(ua, [&](const UpdateArgs& w) { launch_update_xr(w, grid, h.stream); });
      }
      enqueue_iter(++enqueued, false);
    }
  }
  h.predicted_iters[b.kind] = out.iters;
  // the last iteration's x update rode in a gated p update that did not run (the solve converged under a speculative
  // iteration): alpha and p are still that iteration's
  if (xs.finish_at_end(out.iters)) finish_x(out.iters);
  // The solution is complete once the last residual is out; what may still be queued are the gated-off launches of
  // the speculative iteration (they return at once and write nothing).  With the mapped read-back the stream is left
  // to drain on its own -- later calls are ordered behind it anyway; the copy + event path keeps its full wait.
  // (overlap: the same; the second stream is drained by whoever next touches the residual slots it writes -- drain_comm_stream)
  if (!polled || h.prof_on) sync(h);
  for (size_t i = prof_mark; i < h.prof_pending.size(); ++i)  // speculative (gated-off) launches are not samples
    if (h.prof_pending[i].iter > out.iters) h.prof_pending[i].which = -1;
  return out;
}

// Column-sharded runs: every rank owns columns [c0, c1) of an N x ld array.  Make the whole array valid on every
// rank: one ncclBroadcast of each rank's packed slab (slab widths may differ by 4 columns, so not an all-gather).
// Collective: every rank must call it.
void gather_columns(L& h, float* arr) {
  if (!h.comm || h.world <= 1) return;
  const int32_t q = h.dcols / 4;
  int32_t wmax = 0;
  for (int r = 0; r < h.world; ++r) wmax = std::max(wmax, (int32_t)(((int64_t)q * (r + 1) / h.world - (int64_t)q * r / h.world) * 4));
  h.comm_buf.alloc((size_t)h.N * wmax);
  for (int r = 0; r < h.world; ++r) {
    const int32_t lo = (int32_t)((int64_t)q * r / h.world) * 4, hi = (int32_t)((int64_t)q * (r + 1) / h.world) * 4;
    const int32_t w = hi - lo;
    if (w <= 0) continue;
    if (r == h.rank)
      HIP_CHECK(hipMemcpy2DAsync(h.comm_buf.p, (size_t)w * 4, arr + lo, (size_t)h.ld * 4, (size_t)w * 4, (size_t)h.N,
                                 hipMemcpyDeviceToDevice, h.stream));
    h.comm->broadcast_group({CommXfer{h.comm_buf.p, (size_t)h.N * w * 4, r}}, h.stream);
    if (r != h.rank)
      HIP_CHECK(hipMemcpy2DAsync(arr + lo, (size_t)h.ld * 4, h.comm_buf.p, (size_t)w * 4, (size_t)w * 4, (size_t)h.N,
                                 hipMemcpyDeviceToDevice, h.stream));
  }
}

// ---- row-sharded CG (BASELINE north_star wording) --------------------------------------------------------------
// Rank r owns rows [N r/G, N (r+1)/G) of every N x D array and of the lattice graph.  Per iteration: the local rows of
// the search direction p are exchanged so every rank holds all of p for the neighbour gathers ("halo": on i.i.d.
// anchors ~all rows are somebody's neighbour, so the halo is the whole array), and the column sums (p.Ap, then
// [r.r, r.z]) are completed with all-reduces of fp64 D-vectors before alpha / beta / the residual are formed.

std::vector<RowShard> row_shards(const L& h) {
  std::vector<RowShard> v;
  if (h.comm) {
    v.push_back({h.N * h.rank / h.world, h.N * (h.rank + 1) / h.world});
  } else {
    const int V = std::max(1, h.fake_row_shards);
    for (int s = 0; s < V; ++s) v.push_back({h.N * s / V, h.N * (s + 1) / V});
  }
  return v;
}

// make every rank's copy of `arr` complete: each rank broadcasts its own row block (grouped, in place)
void exchange_rows(L& h, float* arr, int32_t ld) {
  if (!h.comm) return;  // (a 1-rank communicator still runs the calls: that is how one GPU exercises this path)
  std::vector<CommXfer> pieces;
  for (int r = 0; r < h.world; ++r) {
    const int64_t a = h.N * r / h.world, b = h.N * (r + 1) / h.world;
    pieces.push_back(CommXfer{arr + (size_t)a * ld, (size_t)(b - a) * ld * 4, r});
  }
  h.comm->broadcast_group(pieces, h.stream);
}

void allreduce_sums(L& h, double* buf, size_t n) {
  if (!h.comm) return;
  h.comm->allreduce(buf, n, COMM_F64, COMM_SUM, h.stream);
}


// ---- halo lists ---------------------------------------------------------------------------------------------------
// Which rows of the search direction a rank needs from its peers: the off-partition column ids its ELL rows (and its
// rows of the chain's path graph) reference.  The adjacency is symmetric (by construction of the build, enforced on
// injection), so "peer q needs my row i" == "my row i has a neighbour in q's row block": both lists of a pair of
// ranks follow from each rank's OWN rows, sorted by row id on both sides, and no index lists are exchanged -- only the
// counts, once, as a consistency check and to take the same full-exchange decision everywhere.
void build_halo_plan(L& h) {
  L::HaloPlan& hp = h.halo;
  const int G = h.world, me = h.rank;
  auto lo = [&](int r) { return host::row_lo(h.N, G, r); };
  const int64_t r0 = lo(me), r1 = lo(me + 1), nloc = r1 - r0;
  std::vector<int32_t> col((size_t)nloc * h.width), deg((size_t)nloc);
  if (nloc > 0) {
    HIP_CHECK(hipMemcpyAsync(col.data(), h.ell_col.p + (size_t)r0 * h.width, col.size() * 4, hipMemcpyDeviceToHost, h.stream));
    HIP_CHECK(hipMemcpyAsync(deg.data(), h.deg.p + r0, (size_t)nloc * 4, hipMemcpyDeviceToHost, h.stream));
  }
  sync(h);
  std::vector<std::pair<int64_t, int64_t>> chain_edges;  // path graph: consecutive chain nodes (graph.py:96-111), device row ids
  if (h.chain_present && h.lamP > 0.0f) {
    auto id = [&](int32_t v) { return permuted(h) ? h.inv_h[(size_t)v] : v; };
    for (size_t t = 0; t + 1 < h.chain_nodes.size(); ++t) chain_edges.emplace_back(id(h.chain_nodes[t]), id(h.chain_nodes[t + 1]));
  }
  host::HaloLists hl = host::build_halo_lists(h.N, G, me, h.width, col.data(), deg.data(), chain_edges);
  hp.give_off = hl.give_off;
  hp.need_off = hl.need_off;
  const std::vector<int32_t>&gi = hl.give_idx, &ni = hl.need_idx;
  hp.give_rows = (int64_t)gi.size();
  hp.need_rows = (int64_t)ni.size();
  // counts of every (rank, peer) pair, all-gathered: row r = [need from 0..G-1 | give to 0..G-1] of rank r
  DevBuf<int32_t> cnt_d;
  cnt_d.alloc((size_t)G * 2 * G);
  std::vector<int32_t> mine((size_t)2 * G), all((size_t)G * 2 * G);
  for (int q = 0; q < G; ++q) {
    mine[(size_t)q] = (int32_t)(hp.need_off[(size_t)q + 1] - hp.need_off[(size_t)q]);
    mine[(size_t)G + q] = (int32_t)(hp.give_off[(size_t)q + 1] - hp.give_off[(size_t)q]);
  }
  HIP_CHECK(hipMemcpyAsync(cnt_d.p + (size_t)me * 2 * G, mine.data(), (size_t)2 * G * 4, hipMemcpyHostToDevice, h.stream));
  h.comm->allgather(cnt_d.p, (size_t)2 * G * 4, h.stream);
  HIP_CHECK(hipMemcpyAsync(all.data(), cnt_d.p, all.size() * 4, hipMemcpyDeviceToHost, h.stream));
  sync(h);
  const host::HaloDecision dec = host::halo_decide(h.N, G, all);
  if (!dec.consistent) throw CommError("halo plan: need / give counts of a rank pair differ (asymmetric lattice graph?)");
  hp.need_rows_max = dec.need_rows_max;
  bool full = dec.full;
  const int force = h.halo_force;  // OSC_HALO = full | lists: force one exchange form (tests, A/B); same on every rank
  if (force == 1) full = true;
  if (force == 2) full = false;
  hp.full = full;
  hp.give_idx.alloc(std::max<size_t>(1, gi.size()));
  hp.need_idx.alloc(std::max<size_t>(1, ni.size()));
  if (!gi.empty()) HIP_CHECK(hipMemcpyAsync(hp.give_idx.p, gi.data(), gi.size() * 4, hipMemcpyHostToDevice, h.stream));
  if (!ni.empty()) HIP_CHECK(hipMemcpyAsync(hp.need_idx.p, ni.data(), ni.size() * 4, hipMemcpyHostToDevice, h.stream));
  if (!full) {
    hp.send.alloc(std::max<size_t>(1, gi.size() * (size_t)h.ld));
    hp.recv.alloc(std::max<size_t>(1, ni.size() * (size_t)h.ld));
  }
  sync(h);
  hp.epoch = h.graph_epoch;
}

// the per-iteration halo exchange of `arr` (N x ld, every rank's own row block current): afterwards the rows this
// rank's operator gathers from are current too
void halo_exchange(L& h, float* arr, int32_t ld) {
  if (!h.comm) return;
  if (h.halo.epoch != h.graph_epoch) build_halo_plan(h);
  L::HaloPlan& hp = h.halo;
  if (hp.full || ld != h.ld) {
    exchange_rows(h, arr, ld);
    return;
  }
  if (hp.give_rows > 0) launch_move_rows(hp.send.p, arr, hp.give_idx.p, hp.give_rows, ld, false, h.stream);  // pack
  std::vector<CommXfer> sends, recvs;
  for (int q = 0; q < h.world; ++q) {
    const int64_t g0 = hp.give_off[(size_t)q], g1 = hp.give_off[(size_t)q + 1];
    const int64_t n0 = hp.need_off[(size_t)q], n1 = hp.need_off[(size_t)q + 1];
    if (g1 > g0) sends.push_back(CommXfer{hp.send.p + (size_t)g0 * ld, (size_t)(g1 - g0) * ld * 4, q});
    if (n1 > n0) recvs.push_back(CommXfer{hp.recv.p + (size_t)n0 * ld, (size_t)(n1 - n0) * ld * 4, q});
  }
  h.comm->exchange(sends, recvs, h.stream);
  if (hp.need_rows > 0) launch_move_rows(arr, hp.recv.p, hp.need_idx.p, hp.need_rows, ld, true, h.stream);  // unpack
}

CgResult run_cg_rows(L& h, const OpParams& op, const CgBuffers& b, bool with_path, int max_iters, float tol) {
  const std::vector<RowShard> shards = row_shards(h);
  const int V = (int)shards.size();
  const int grid = cg_grid(h);
  const size_t pn = (size_t)V * grid * b.ld;  // one block of partial rows per local shard
  if (h.part0.n < pn) h.part0.alloc(pn);
  if (h.part1.n < pn) h.part1.alloc(pn);
  h.sums.alloc((size_t)2 * b.ld);
  double* s0 = h.sums.p;
  double* s1 = h.sums.p + b.ld;
  HIP_CHECK(hipMemsetAsync(h.res_bits.p, 0, ((size_t)max_iters + 2) * 4, h.stream));
  if (h.res_host_n < (size_t)max_iters + 2) {
    if (h.res_host) (void)hipHostFree(h.res_host);
    h.res_host = nullptr;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h.res_host), ((size_t)max_iters + 2) * 4, hipHostMallocDefault));
    h.res_host_n = (size_t)max_iters + 2;
  }
  while (h.iter_events.size() < (size_t)max_iters + 2) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h.iter_events.push_back(e);
  }
  const float* res_dev = reinterpret_cast<const float*>(h.res_bits.p);
  SpmmArgs sa{};
  sa.g = graph_view(h, with_path);
  sa.op = op;
  sa.B = b.B;
  sa.psi = b.psi;
  sa.ld = b.ld;
  sa.c0 = b.c0;
  sa.c1 = b.c1;
  UpdateArgs ua{};
  ua.X = b.X;
  ua.R = b.R;
  ua.P = b.P;
  ua.AP = b.AP;
  ua.B = b.B;
  ua.alpha = h.alpha.p;
  ua.beta = h.beta.p;
  ua.op = op;
  ua.ld = b.ld;
  ua.c0 = b.c0;
  ua.c1 = b.c1;
  auto for_shards_spmm = [&](int mode, int iter) {
    for (int s = 0; s < V; ++s) {
      sa.row0 = shards[(size_t)s].r0;
      sa.N = shards[(size_t)s].r1;
      sa.part = h.part0.p + (size_t)s * grid * b.ld;
      spmm_slabbed(h, mode, sa, grid, iter);
    }
  };
  // r = b - A x0 ; z ; p ; rz
  sa.X = b.x0;
  sa.OUT = b.X;
  sa.R = b.R;
  sa.P = b.P;
  sa.U = b.rhsU;
  sa.Y = b.rhsY;
  sa.gate = nullptr;
  for_shards_spmm(SPMM_INIT, 0);
  launch_reduce_sum(h.part0.p, V * grid, b.ld, b.c0, b.c1, s0, h.stream);
  allreduce_sums(h, s0 + b.c0, (size_t)(b.c1 - b.c0));
  launch_finish_init(s0, b.c0, b.c1, h.rz.p, h.stream);
  halo_exchange(h, b.P, b.ld);
  sa.X = b.P;
  sa.OUT = b.AP;

  auto enqueue_iter = [&](int it) {
    const Gate g{it > 1 ? res_dev + (it - 1) : nullptr, tol};
    sa.gate = g.p;
    sa.gate_tol = tol;
    ua.gate = g.p;
    ua.gate_tol = tol;
    if (it > 1) {
      for (int s = 0; s < V; ++s) {
        ua.row0 = shards[(size_t)s].r0;
        ua.N = shards[(size_t)s].r1;
        ProfScope ps(h, 2, it);
        for_windows(ua, [&](const UpdateArgs& w) { launch_update_p(w, grid, h.stream); });
      }
      halo_exchange(h, b.P, b.ld);  // the halo exchange of this iteration
    }
    for_shards_spmm(SPMM_AP, it);
    launch_reduce_sum_gated(h.part0.p, V * grid, b.ld, b.c0, b.c1, s0, g, h.stream);
    allreduce_sums(h, s0 + b.c0, (size_t)(b.c1 - b.c0));
    launch_finish_alpha(s0, b.c0, b.c1, h.rz.p, h.alpha.p, g, h.stream);
    for (int s = 0; s < V; ++s) {
      ua.row0 = shards[(size_t)s].r0;
      ua.N = shards[(size_t)s].r1;
      ua.part_rr = h.part0.p + (size_t)s * grid * b.ld;
      ua.part_rz = h.part1.p + (size_t)s * grid * b.ld;
      ProfScope ps(h, 1, it);
      for_windows(ua, [&](const UpdateArgs& w) { launch_update_xr(w, grid, h.stream); });
    }
    launch_reduce_sum_gated(h.part0.p, V * grid, b.ld, b.c0, b.c1, s0, g, h.stream);
    launch_reduce_sum_gated(h.part1.p, V * grid, b.ld, b.c0, b.c1, s1, g, h.stream);
    allreduce_sums(h, s0, (size_t)2 * b.ld);  // [r.r | r.z] in one message
    launch_finish_beta(s0, s1, b.c0, b.c1, h.rz.p, h.beta.p, h.res_bits.p + it, g, h.stream);
    HIP_CHECK(hipMemcpyAsync(h.res_host + it, h.res_bits.p + it, 4, hipMemcpyDeviceToHost, h.stream));
    HIP_CHECK(hipEventRecord(h.iter_events[(size_t)it], h.stream));
  };

  h.history.clear();
  CgResult out{max_iters, 0.f, b.X};
  const size_t prof_mark = h.prof_pending.size();
  enqueue_iter(1);
  for (int it = 1; it <= max_iters; ++it) {
    if (it < max_iters) enqueue_iter(it + 1);
    HIP_CHECK(hipEventSynchronize(h.iter_events[(size_t)it]));
    const float res = h.res_host[it];
    h.history.push_back(res);
    out.res = res;
    if ((double)res <= (double)tol) {
      out.iters = it;
      break;
    }
  }
  exchange_rows(h, b.X, b.ld);  // every rank leaves with the whole solution
  sync(h);
  for (size_t i = prof_mark; i < h.prof_pending.size(); ++i)
    if (h.prof_pending[i].iter > out.iters) h.prof_pending[i].which = -1;
  return out;
}

bool row_mode(const L& h) { return h.shard_mode == 1 && (h.comm != nullptr || h.fake_row_shards > 1); }

// The ONE place the per-handle OSC_* switches are read (osc_create, osc_rebuild_graph).  Process-wide ones are read where
// the process-wide object is made: OSC_POOL_MB (device memory pool), OSC_PINNED_DL / OSC_COPY_THREADS (read-back staging),
// OSC_LOOPBACK_TIMEOUT_S / OSC_RCCL_PROXY (communicator backends, comm.hip), OSC_LD (osc_create, before the arrays are sized).
