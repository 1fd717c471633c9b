// The "panel" prefilter of the lattice build (graph.py:35-62, all-pairs cosine similarity -> per-row top-k): an fp16
// similarity GEMM shaped for gfx950's matrix pipe, with NO list maintenance inside the GEMM.
//
//   workgroup = 4 waves, ONE per SIMD (the whole 512-entry register file per wave), 128 query rows (wave w: rows
//   32 w .. 32 w + 31), persistent over work items (row block, column split) drawn from an atomic queue;
//   A: the wave's 32 x D fp16 query panel stays in registers for its whole column sweep (D = 768: 48 half8 = 192
//      VGPRs) -- it never passes through LDS again, which halves the LDS-DMA volume per MFMA of the 128 x 128
//      two-waves-per-SIMD shape (knn_kernels.hip: k_knn_pref), whose K steps ran at LDS-DMA latency;
//   B: 128 columns x 64 halfs per K step (16 KB) by global_load_lds_dwordx4 into a 6-stage ring, two K steps per
//      barrier, the pair fetched during a pair stays in flight across the barrier (counted vmcnt); every address of
//      the K loop is a register set once per tile plus a compile-time immediate (the K offset rides in the DMA
//      instruction's immediate, which the hardware also adds to the LDS destination: M0 carries destination - offset);
//      bank swizzle on the source address, the same xor on the ds_read_b128 fragment reads;
//   one v_mfma_f32_32x32x16_f16 per (k16 slice, 32-column subtile): 16 per K step and wave.
//
// Selection is split off the matrix pipe (at one wave per SIMD every sorted list insert would be fully exposed):
//   phase A  (k_panel<NKT, 0>): the same GEMM against a strided SAMPLE of the columns; epilogue = per (row, sample tile)
//            maximum.  tau_row = the r-th largest of its tile maxima: a lower bound of the r-th best sample score, hence
//            of the row's final keep-th best score whenever at least `keep` columns beat it (checked, not assumed).
//   phase B  (k_panel<NKT, 1>): the full sweep; epilogue = compare against tau_row (one max + compare per query row and
//            tile, ballots only for rows with a hit) and APPEND the few hits (row, column, score) to the work item's
//            private candidate slots: no atomics, no sorted inserts, no threshold updates, so column splits cost
//            nothing and balance the persistent grid.
//   select   (k_panel_select): per row the keep best candidates by fp16 score -> the lists k_knn_rescore re-scores in
//            exact fp32 and proves (unchanged contract: every left-out column has fp16 score <= the list's last).
//            A row with an overflowed split or fewer than keep candidates goes to the exact kernel's row list.
// The neighbour lists that come out are therefore those of the exact fp32 scoring, as with the other routes.
//
// D <= 384 (K depth 6): a wave's panel is only 96 registers, so it carries TWO row groups -- its 32 rows of two
// consecutive 128-row blocks (NRG = 2: 192 panel registers, eight 32 x 32 accumulators).  Every B fragment read from LDS
// then feeds two MFMAs instead of one: with one row group the LDS reads and the MFMAs of a K step take the same number
// of cycles (4 waves x 16 KB / 128 B per clk = 512 clk = 16 MFMAs x 32 clk), which caps that shape near 50-60 % of the
// matrix pipe whatever else is hidden.  Hit lists, thresholds and the select stay per (128-row block, wave): nothing
// outside k_panel knows about the pairing.
#include "knn_gemm.hpp"

#include <algorithm>
#include <cmath>
#include <type_traits>

namespace osc {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {  // f(std::integral_constant<int, I>{}) for I in [I, N): compile-time indices
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int RING = 6;
constexpr unsigned STAGE_BYTES = 16384;  // one K step of a column tile: 128 rows x 128 B
// MODE 1: each wave appends its hits to a private LDS list (ds_write is not on the vmcnt queue the DMA ring is counted
// on: global stores in the epilogue made every tile wait for their completion) and flushes it at the end of the item
constexpr int HB_CAP = 1792;  // entries of 8 bytes per wave: 4 x 14 KB behind the ring
constexpr size_t PANEL_LDS = (size_t)RING * STAGE_BYTES + 2048;
constexpr size_t PANEL_LDS_HITS = PANEL_LDS + (size_t)4 * HB_CAP * 8;
// Symmetric half sweep (SYM, single-process builds): S = Yh Yh^T is symmetric bit for bit (every product a_ik b_jk is
// formed once per pair and the MFMA sums k in the same order whichever operand a row arrives as), so row block I only
// visits column tiles J >= I and tests every accumulator against BOTH thresholds -- tau of its row (the entry is a
// candidate of row i: "row side") and tau of its column (a candidate of row j: "column side", J > I only).  That halves
// the MFMA, LDS-fragment and LDS-DMA work of the sweep for about twice the hit test per visited tile.  The column
// thresholds of an item's tiles wait in LDS (the per-lane tau[column] would otherwise be a vector load per tile on the
// vmcnt queue the DMA ring is counted on): the same 56 KB behind the ring hold TC_TILES x 512 B of thresholds and four
// hit lists of HB_CAP_SYM entries.  A delivered entry carries a flag (bit 26: a candidate of its receiving row).  At the end of
// an item a wave delivers its hits to global BUCKETS, one per group of 32 receiving rows (= one (row block, wave) of
// the select): the row-side ones to its own bucket behind one reservation, the column-side ones to the buckets of
// their columns -- counted per bucket in LDS first (the item's tiles span at most 4 TC_TILES buckets), one returning
// atomic per touched bucket, 64 buckets per wave instruction.  (A first version sent the column-side hits through one
// global pool and counting-sorted it afterwards: 9.6 M same-address atomics made those two passes cost 2.7 ms at
// config 3 -- more than the sweep saved at N = 20 000.)
// Round 4, second step: the epilogue no longer resolves WHICH of a query-row register's four subtile scores passed.  It
// appends one COARSE entry per (row register, lane) whose union test fired -- {local row, tile of the item, lane} and the
// four scores, 20 bytes -- behind ONE ballot per row register, and the flush, which runs once per item with a lane per
// entry, takes the entries apart (row side / column side per score, the diagonal and the ragged tail).  Per visited tile
// that is 16 scalar decisions instead of ~56 and ~8 instead of ~60 instructions per row register with a hit.
// The lists are a SOFT limit: a wave delivers its list whenever the next tile's entries (counted from the union masks before
// they are appended) would not fit -- the counters it needs are its own, behind the threshold window -- so an item is as
// long as the window allows whatever the hit density, and only ONE tile that yields more than a whole list can overflow
// (the overflow path stays: the rows concerned go to the exact kernel).  (Delivering a half-full list AFTER a tile was
// not enough: at N = 40 000, D = 256 on clustered anchors a tile adds 39 entries on average to lists of 260, some tiles
// five times that, and every overflow sends its chunk's 3072 rows to the exact kernel: 8768 fallback rows against 66.)
constexpr int TC_TILES = 24;       // tiles per chunk the threshold window holds
constexpr int HB_CAP_SYM = 520;    // coarse entries per wave
constexpr size_t SYM_CNT_BYTES = (size_t)4 * 8 * TC_TILES * 4;  // per wave [2][4 TC_TILES] ints: bucket counts / cursors, bases
static_assert((size_t)4 * HB_CAP_SYM * 20 + (size_t)TC_TILES * 512 + SYM_CNT_BYTES <= (size_t)4 * HB_CAP * 8, "SYM layout must fit the hit area");
constexpr unsigned ROW_SIDE = 1u << 26, COL_SIDE = 1u << 25, COL_MASK = (1u << 25) - 1u;

struct PanelArgs {
  const _Float16* A;   // query image, npad rows
  const _Float16* B;   // column image, ntileB * 128 rows (the query image itself, or the sample)
  int32_t ldh, N, ntileB, S, tiles_per_split;
  int32_t rb_begin, rb_count;  // query row blocks this launch covers (a rank's share in a sharded build)
  int32_t group_tiles, ngroups;  // MODE 0: tile maxima are folded over groups of consecutive sample tiles
  float* tmax;         // MODE 0: [npad][ngroups]
  const float* tau;    // MODE 1: [npad]
  uint2* hit_list;     // MODE 1: [(list * 4 + wave) * hit_cap + e] = {local row << 27 | column, score bits}; list = split * rb_count + (row block - rb_begin)
  int32_t* hit_cnt;    // MODE 1: [list * 4 + wave] hits of that wave in that list (may exceed hit_cap: overflow)
  int32_t hit_cap;     // entries per list: HB_CAP (SYM: HB_CAP_SYM) / row groups per wave
  unsigned* queue;
  // SYM: column chunks of T tiles, walked from the last to the first; chunk c is swept by the row blocks I < min(nrb, (c + 1) T)
  int32_t T, nchunks;
  uint2* bucket_ent;     // [npad / 32][bucket_cap] entries {local row << 27 | ROW_SIDE | candidate, score bits} per group of 32 receiving rows
  int32_t* bucket_cnt;   // [npad / 32] entries delivered (may exceed bucket_cap: overflow); zeroed by the caller
  int32_t bucket_cap;
  int32_t* flags;        // [c]: an LDS list of chunk c overflowed (hits of its tiles' rows were lost); zeroed by the caller
  int32_t set0, nset;    // k_tile_thr2: the sets of two row blocks this launch sweeps (a group whose image rows stay in the Infinity Cache)
  // half sweep of a SHARDED build: this rank sweeps the work items item_offset, item_offset + item_stride, ... (items of
  // one chunk stay neighbours in every rank's sequence); 1 / 0 otherwise
  int32_t item_stride, item_offset;
  // k_panel's half sweep only: the launch takes the work items [item_begin, item_end) of the queue's order (item_end = 0: all)
  // -- the items of some column chunks, launched by the streamed create as the rows those chunks need arrive
  int32_t item_begin, item_end;
};

// ---- the K loop's instructions, each an asm statement of its own: they issue in the order they are written (round 6) ----
// Until round 5 a k16 slice was [fragment reads][MFMAs back to back][DMA pieces of five instructions] between scheduling
// fences: the 24 issue cycles an MFMA leaves free stayed empty and everything else was issued between the groups, where
// only the last MFMA's tail covers it.  Now every instruction of the loop is placed by hand, at most two single-issue
// instructions behind an MFMA (MI355X_MICROARCH.md: <= 5 hide per v_mfma_f32_32x32x16 gap); hipcc's own s_waitcnt
// insertion knows nothing of these loads, so the counted waits are written out too (LDS returns in order: a wait that
// also covers compiler-issued LDS traffic only waits longer).
#define OSC_RD(DST, ABASE, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ABASE), "n"(OFF))
#define OSC_LGKM(N_) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N_))
#define OSC_VMC(N_) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory")
// LDS-DMA of one 1 KiB piece of K step KT: M0 = LDS destination - K offset (ONE s_add; nothing else in the kernel touches
// M0 -- checked in the disassembly), the K offset rides in the load's immediate, which the hardware adds to the global
// address AND to the LDS destination.  An MFMA (or the prologue's s_nop) sits between the M0 write and the load.
#define OSC_SETM0(KT, STG_, Q) \
  asm volatile("s_add_i32 m0, %0, %1" ::"s"(fill_base), "n"((unsigned)((STG_) * STG + (Q) * 1024 - (KT) * 128)) : "scc")
// (source = scalar base of the pass + the lane's 32-bit byte offset within it: the pass-to-pass advance is scalar arithmetic and a
// lane keeps NT offsets instead of 2 NT pointer pairs -- the two-row-group shape at D = 768 has no vector register to spare)
#define OSC_DMA(VOFF, SBASE, KT) \
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(VOFF), "s"(SBASE), "n"((KT) * 128) : "memory")

template <int NKT, int MODE, int NRG, bool SYM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_panel(const PanelArgs a) {
  static_assert(NRG == 1 || NRG == 2, "row groups per wave");
  static_assert(!SYM || MODE == 1, "the half sweep is a form of the main sweep");
  // Round 6: D <= 768 with TWO row groups.  2 x 192 panel registers + eight accumulators do not fit 512, so the K loop of
  // that shape covers 64 columns per pass (NT = 2 subtiles, four 32 x 32 accumulators, two passes per 128-column tile, the
  // epilogue behind each pass on its two subtiles): every B fragment feeds two MFMAs, half the fragment reads and half the
  // LDS-DMA bytes per MFMA.  The sweep is POWER-bound (scripts/exp/r06/README.md: the shader clock settles near 1.75 GHz
  // and every cycle saved by scheduling alone comes back as a lower clock), so bytes moved per flop are what count.
  constexpr int NT = (NKT == 12 && NRG == 2) ? 2 : 4;   // 32-column subtiles per K-loop pass
  constexpr int NPASS = 4 / NT;                          // passes per 128-column tile
  constexpr unsigned STG = NT * 4096u;                   // bytes of one stage: 32 NT columns x 128 B (one K step)
  constexpr int NSTG = (int)(RING * STAGE_BYTES / STG);  // 6 / 12 stages in the same 96 KB
  constexpr int GK = NT == 4 ? 2 : 3;                    // K steps per barrier group
  constexpr int LA = 2;                                  // groups fetched ahead
  constexpr int NG = NKT / GK;
  constexpr int MM = NRG * NT;                           // MFMAs per k16 slice
  constexpr int APAN = 32;                               // k16 slices of a row group's panel in AGPRs where two groups need 384 registers (the rest in VGPRs)
  constexpr int HS = NT == 4 ? 3 : 8;                    // stages behind the low / high fragment base (16-bit offset field)
  static_assert(NKT % NSTG == 0 || NSTG == NKT, "a pass's K steps must be whole laps of the ring (compile-time stage indices)");
  static_assert(NKT % GK == 0 && (LA + 1) * GK <= NSTG && LA >= 2 && LA < NG, "ring plan");
  constexpr int NK16 = NKT * 4;
  constexpr int HCAP = (SYM ? HB_CAP_SYM : HB_CAP) / NRG;  // entries of one (row group, wave) hit list
  extern __shared__ __attribute__((aligned(1024))) float lds[];  // NSTG stages x [32 NT rows][32 float slots] (+2 KB lead)
  __shared__ int s_item, s_chunk, s_first;
  __shared__ __attribute__((aligned(16))) float s_tau[4][NRG][2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int frow = lane >> 3;
  const int swz = (l31 >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)lds + 2048u;
  const unsigned fill_base = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(8 * NT * wave * 128));
  // fragment read bases (absolute LDS byte addresses) of k16 slice s: row l31, chunk (2 s + h) ^ swz   (+ stage, + subtile * 4096)
  unsigned rlo[4], rhi[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    rlo[s] = lds_base + (unsigned)(l31 * 128 + (((2 * s + h) ^ swz) * 16));
    rhi[s] = rlo[s] + (unsigned)HS * STG;
  }
  const int nsets = (a.rb_count + NRG - 1) / NRG;  // work item = (column split, set of NRG consecutive row blocks)
  // SYM: chunk c (T tiles) is swept by the row blocks I < min(nrb, (c + 1) T), i.e. by all nrb of them in the last chunk
  // and by (c + 1) T in the others, so the item and list tables have closed forms (host copies: knn_panel_sym_tables):
  // lists before chunk c: T c (c + 1) / 2; the queue walks the chunks from the last to the first
  auto chunk_sets = [&](int c) { return (min(a.ntileB, (c + 1) * a.T) + NRG - 1) / NRG; };
  int nitems = nsets * a.S;
  if constexpr (SYM) {
    nitems = 0;
    for (int c = 0; c < a.nchunks; ++c) nitems += chunk_sets(c);
  }
  const size_t ldh = (size_t)a.ldh;
  const size_t tile_stride = (size_t)128 * ldh;  // halfs between column tiles
  // SYM: behind the ring [headers 4 x HB_CAP_SYM x 4 B][scores 4 x HB_CAP_SYM x 16 B][thresholds: tile of the item x lane 0..31 x subtile 0..3]
  unsigned* const s_hdr = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds) + PANEL_LDS);
  v4f* const s_sc = reinterpret_cast<v4f*>(reinterpret_cast<char*>(lds) + PANEL_LDS + (size_t)4 * HB_CAP_SYM * 4);
  float* const s_tc = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + PANEL_LDS + (size_t)4 * HB_CAP_SYM * 20);
  if constexpr (SYM) {
    if (a.item_end > 0) nitems = min(nitems, a.item_end);
  }
  for (;;) {
    if (tid == 0) {
      const int it = (SYM ? a.item_begin : 0) + (int)atomicAdd(a.queue, 1u) * a.item_stride + a.item_offset;
      s_item = it;
      if constexpr (SYM) {  // chunks are queued from the last (every row block sweeps it) to the first (T row blocks): the
        int c = a.nchunks - 1, first = 0;  // short diagonal items come last and fill the tail of the persistent grid
        if (it < nitems)
          while (it >= first + chunk_sets(c)) first += chunk_sets(c), --c;
        s_chunk = c;
        s_first = first;
      }
    }
    __syncthreads();
    const int item = s_item;
    const int chunk = SYM ? s_chunk : 0;
    const int first_item = SYM ? s_first : 0;
    __syncthreads();
    if (item >= nitems) break;
    // items of one split (chunk) are consecutive: the workgroups that start together sweep the same column tiles together
    int split, rbi0, t0, t1;
    if constexpr (SYM) {
      split = chunk;
      rbi0 = (item - first_item) * NRG;
      t1 = min(a.ntileB, (chunk + 1) * a.T);
      t0 = max(chunk * a.T, rbi0);  // row group 0's diagonal tile or the chunk's first tile
    } else {
      split = item / nsets;
      rbi0 = (item - split * nsets) * NRG;  // first row block of the set, relative to rb_begin
      t0 = split * a.tiles_per_split;
      t1 = min(a.ntileB, t0 + a.tiles_per_split);
    }
    int rbv[NRG];     // row block of row group r (the last set of an odd count repeats its first block: computed, not kept)
    bool rok[NRG];
#pragma unroll
    for (int r = 0; r < NRG; ++r) {
      rok[r] = rbi0 + r < a.rb_count;
      rbv[r] = a.rb_begin + (rok[r] ? rbi0 + r : rbi0);
    }
    // list of (this item's split / chunk, row block rbi0 + r): SYM packs the chunks' lists (chunk c has min(nrb, (c + 1) T) of them)
    auto list_of = [&](int r) -> size_t {
      return SYM ? (size_t)a.T * chunk * (chunk + 1) / 2 + rbi0 + r : (size_t)split * a.rb_count + rbi0 + r;
    };
    if (t0 >= t1) {  // an empty column split (the planner never makes one; OSC_KNN_MODE=panel can): no hits, but say so
      if (MODE == 1 && lane == 0) {
#pragma unroll
        for (int r = 0; r < NRG; ++r)
          if (rok[r]) a.hit_cnt[list_of(r) * 4 + wave] = 0;
      }
      continue;
    }
    if constexpr (SYM) {  // the column thresholds of the item's tiles -> LDS, [tile][lane][subtile]; +inf beyond N: no receiver
      const int nq = (t1 - t0) * 32;  // quads of consecutive columns (tau has npad entries: whole quads)
      constexpr int QPT = (TC_TILES * 32 + 255) / 256;
      v4f tq[QPT];
#pragma unroll
      for (int j = 0; j < QPT; ++j) {  // all loads first, then the LDS stores
        const int q = tid + 256 * j;
        tq[j] = q < nq ? *reinterpret_cast<const v4f*>(a.tau + (size_t)t0 * 128 + 4 * q) : v4f{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        const int q = tid + 256 * j;
        if (q < nq) {
          const int e = 4 * q, tl = e >> 7, sub = (e >> 5) & 3, l5 = e & 31;
#pragma unroll
          for (int u = 0; u < 4; ++u) s_tc[tl * 128 + (l5 + u) * 4 + sub] = (t0 * 128 + e + u) < a.N ? tq[j][u] : 3.0e38f;
        }
      }
    }
    half8 areg[NRG][NK16];
#pragma unroll
    for (int r = 0; r < NRG; ++r) {
      const int row = rbv[r] * 128 + 32 * wave + l31;
#pragma unroll
      for (int i = 0; i < NK16; ++i) areg[r][i] = *(const half8*)(a.A + (size_t)row * ldh + i * 16 + h * 8);
    }
    // per query-row register g of this half-wave: local row (g & 3) + 8 (g >> 2) + 4 h of the wave's 32
    float taug[MODE == 0 ? NRG : 1][16];  // MODE 0: running maxima of the current tile group
    // MODE 1: the rows' thresholds wait in LDS, in the order the hit test reads them ([wave][row group][half][16]), and are
    // fetched into registers per tile behind the K loop: held in registers (16 per row group) next to two row groups'
    // accumulators and the 192-register panel they made hipcc spill 43 registers
    int wcnt[NRG];        // MODE 1: hits this wave has appended to its list of row group r in this item (wave-uniform)
    uint2* hitbuf[NRG];
#pragma unroll
    for (int r = 0; r < NRG; ++r) {
      wcnt[r] = 0;
      hitbuf[r] = reinterpret_cast<uint2*>(reinterpret_cast<char*>(lds) + PANEL_LDS) + (wave * NRG + r) * HCAP;  // (SYM: the shorter lists leave the tail of the area to s_tc)
      const int grow0 = rbv[r] * 128 + 32 * wave + 4 * h;
      if constexpr (MODE == 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) taug[r][g] = -3.0e38f;
      }
      if constexpr (MODE == 1) {
        if (l31 < 16) s_tau[wave][r][h][l31] = a.tau[grow0 + (l31 & 3) + 8 * (l31 >> 2)];
      }
    }
    // source of piece q of this wave's share of a pass's columns: row 8 NT wave + 8 q + frow of the pass, swizzled chunk
    unsigned boff[NT];  // byte offset within the pass's columns
#pragma unroll
    for (int q = 0; q < NT; ++q)
      boff[q] = (unsigned)(((size_t)(8 * NT * wave + 8 * q + frow) * ldh + ((lane & 7) ^ (((q & 1) << 2) | (frow >> 1))) * 8) * 2);
    const size_t pass_stride = (size_t)(32 * NT) * ldh;  // halfs between passes
    const _Float16* pbase;  // first column of the pass to run next (wave-uniform, kept scalar)
    {
      const size_t pb = reinterpret_cast<size_t>(a.B + (size_t)t0 * tile_stride);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pb), hi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
      pbase = reinterpret_cast<const _Float16*>(((size_t)hi << 32) | lo);
    }
#pragma unroll
    for (int kt = 0; kt < LA * GK; ++kt)
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        OSC_SETM0(kt, kt, q);
        asm volatile("s_nop 0");
        OSC_DMA(boff[q], pbase, kt);
      }
    OSC_VMC(0);
    __syncthreads();
    // ---- one column tile: the K loop, then the tile's epilogue on its accumulators.  Moving the hit test into the
    // next tile's K loop was tried twice and dropped: with a second accumulator set hipcc spills 51 registers next to the
    // 192-register panel; with the hit row-registers parked in LDS (20 bytes per lane) and taken apart behind the MFMA
    // groups it runs, correctly, no faster (17.6 vs 17.0 ms) -- the K loop of a lone wave per SIMD is bound by its own
    // instruction issue (32 MFMAs, 32 fragment reads, 8 DMA pieces of ~5 instructions per pair), not by the matrix pipe,
    // so VALU work placed there is not hidden.
    f32x16 acc[NRG][NT];
    // tc[t]: SYM, the threshold of this lane's column of subtile t (+inf where the column side does not apply: the
    // diagonal tile, whose pairs are all met on the row side, and columns beyond N)
    auto row_mask = [&](auto GC, const f32x16(&pa)[4], const float(&tg)[16], const float(&tc)[4]) -> unsigned long long {
      constexpr int g = decltype(GC)::value;
      (void)tc;
      return __ballot(fmaxf(fmaxf(pa[0][g], pa[1][g]), fmaxf(pa[2][g], pa[3][g])) > tg[g]);
    };
    // SYM, the union test of a row register: "some score of this (row register, lane) beats its row's threshold or its
    // column's".  Round 6: ONE compare against min(tau_row, the smallest of the lane's column thresholds) instead of a
    // compare per side with a subtraction per subtile (4 VALU per register instead of 11) -- a superset of the exact
    // union (any score above its own column's threshold is above the smallest one), and the flush decides every score
    // against its own two thresholds anyway: the lists gain a few coarse entries whose scores all fail there.
    // (v_max3 / v_min3 written as asm: through fmaxf / fminf hipcc first quiets every operand that "might be a signalling NaN"
    // with a v_max x, x of its own -- two extra instructions per test; MFMA results and loaded thresholds never are, and a NaN
    // score fails the compare either way)
    auto max3r = [](float x, float y, float z) { float m; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(x), "v"(y), "v"(z)); return m; };
    auto max2r = [](float x, float y) { float m; asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y)); return m; };
    auto min3r = [](float x, float y, float z) { float m; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(x), "v"(y), "v"(z)); return m; };
    auto min2r = [](float x, float y) { float m; asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y)); return m; };
    auto row_mask_u = [&](auto GC, const f32x16(&pa)[NT], const float(&tg)[16], const float tcmin) -> unsigned long long {
      constexpr int g = decltype(GC)::value;  // NT = 4: the row register; NT = 2: the PAIR of row registers 2 g, 2 g + 1 (rows rl, rl + 1)
      if constexpr (NT == 4) {
        return __ballot(max2r(max3r(pa[0][g], pa[1][g], pa[2][g]), pa[3][g]) > min2r(tg[g], tcmin));
      } else {
        return __ballot(max2r(max3r(pa[0][2 * g], pa[1][2 * g], pa[0][2 * g + 1]), pa[1][2 * g + 1]) > min3r(tg[2 * g], tg[2 * g + 1], tcmin));
      }
    };
    auto hit_rows = [&](auto GC, const f32x16(&pa)[4], const float(&tg)[16], const float(&tc)[4], int pct, int rb, uint2* hb, int& wc) {  // query-row register g of the tile pct has a hit
      constexpr int g = decltype(GC)::value;
      const int rl = (g & 3) + 8 * (g >> 2) + 4 * h;  // local row of the wave's 32
      const int grow = rb * 128 + 32 * wave + rl;
      const int cbase = pct * 128 + l31;
      const bool special = pct == rb || (pct + 1) * 128 > a.N;  // the tile holds the diagonal or the ragged tail
      unsigned long long mk[4];
      unsigned side[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bool pred = pa[t][g] > tg[g];
        if (special) pred = pred && (cbase + 32 * t) != grow && (cbase + 32 * t) < a.N;  // graph.py:37: no self-similarity
        side[t] = pred ? ROW_SIDE : 0u;
        if constexpr (SYM) {
          const bool cp = pa[t][g] > tc[t];
          side[t] |= cp ? COL_SIDE : 0u;
          pred = pred | cp;
        }
        mk[t] = __ballot(pred);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (mk[t] == 0ull) continue;
        const bool pred = (mk[t] >> lane) & 1ull;
        const int pos = wc + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk[t] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk[t], 0u));
        if (pred && pos < HCAP)
          hb[pos] = make_uint2(((unsigned)rl << 27) | side[t] | (unsigned)(cbase + 32 * t), __float_as_uint(pa[t][g]));
        wc += __popcll(mk[t]);
      }
    };
    auto hit_test_tile = [&](const f32x16(&pa)[4], const float(&tg)[16], const float(&tc)[4], int pct, int rb, uint2* hb, int& wc) {  // all 16 registers, compares batched ahead of the branches
      unsigned long long fm[16];
      static_for<0, 16>([&](auto GC) { fm[decltype(GC)::value] = row_mask(GC, pa, tg, tc); });
      static_for<0, 16>([&](auto GC) {
        if (fm[decltype(GC)::value] != 0ull) hit_rows(GC, pa, tg, tc, pct, rb, hb, wc);
      });
    };
    // ---- one pass of the K loop: 32 NT columns against the wave's NRG row groups, all NKT K steps; hand-placed.
    // Slice (k16) of MM = NRG NT MFMAs, j = r NT + t on fragment t:
    //   NT = 4:  [wait cur 0,1] m0  RD nxt0 nxt1   m1  [M0 if MM = 4]   [wait cur 2,3] m2  RD nxt2 nxt3   m3 ... m(MM-2) [M0]  m(MM-1) [DMA]
    //   NT = 2:  [wait cur 0]   m0  RD nxt0   [wait cur 1] m1  RD nxt1   m2 [M0]   m3 [DMA]      (a piece every second slice)
    // K steps in groups of GK behind one workgroup barrier; the group LA ahead (of this pass or the next) is fetched during a
    // group, one piece per slice (NT = 4) / two slices (NT = 2).  The barrier sits BEFORE the group's last slice, whose
    // fragments are in registers by then (lgkmcnt(0): the wave is done with the group's stages; vmcnt: its own pieces of the
    // next group have landed), so the first slice of the next group is read behind the barrier under that slice's MFMAs.
    // The pass's first fragments are read at its start (nothing is carried across the epilogue).
    auto k_pass = [&](auto LASTC) {
      constexpr bool last_pass = decltype(LASTC)::value;  // the item's last pass: there is no next one to fetch
      const _Float16* const nbase = pbase + pass_stride;
      v4f fa[NT], fb[NT];
#define OSC_RD_AT(DST, ST, SL, T)                                            \
  do {                                                                        \
    if constexpr ((ST) < HS) OSC_RD(DST, rlo[SL], (ST) * STG + (T) * 4096);   \
    else OSC_RD(DST, rhi[SL], ((ST) - HS) * STG + (T) * 4096);                \
  } while (0)
      static_for<0, NT>([&](auto TC) { OSC_RD_AT(fa[decltype(TC)::value], 0, 0, decltype(TC)::value); });
      static_for<0, 4 * NKT>([&](auto SS) {
        constexpr int sx = decltype(SS)::value;
        constexpr int kt = sx >> 2, sl = sx & 3, g = kt / GK, ug = sx % (4 * GK);
        constexpr bool group_end = ug == 4 * GK - 1, has_next = sx + 1 < 4 * NKT;
        constexpr int nst = ((sx + 1) >> 2) % NSTG, nsl = (sx + 1) & 3;
        constexpr bool next_pass = g + LA >= NG;
        constexpr int fg = (g + LA) % NG;  // the group fetched during this one
        constexpr bool fetch = !(next_pass && last_pass);
        // the piece this slice carries: NT = 4 one per slice, NT = 2 one per two slices (the odd ones)
        constexpr bool has_piece = fetch && (NT == 4 || (ug & 1) == 1);
        constexpr int pi = NT == 4 ? ug : ug >> 1;  // piece of the group: K step pi / NT of it, part pi % NT
        constexpr int fk = fg * GK + pi / NT, pq = pi % NT, fst = fk % NSTG;
        constexpr int ai = kt * 4 + sl;
        v4f(&cur)[NT] = (sx & 1) ? fb : fa;
        v4f(&nxt)[NT] = (sx & 1) ? fa : fb;
        const _Float16* const psrc = next_pass ? nbase : pbase;
        // (panel slices in ACCUMULATION registers, accumulators in VECTOR registers: hipcc's own choice was the reverse for
        // the accumulators, which cost a v_accvgpr_read per value in the hit test; where two row groups need 384 panel
        // registers the slices from APAN on live in vector registers.  The pass's first slice starts from C = 0.)
#define OSC_MFMA(J)                                                                                                              \
  do {                                                                                                                           \
    constexpr int r_ = (J) / NT, t_ = (J) % NT;                                                                                  \
    if constexpr (NRG * NK16 * 4 <= 256 || ai < APAN) {                                                                          \
      if constexpr (sx == 0)                                                                                                     \
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc[r_][t_]) : "a"(areg[r_][ai]), "v"(cur[t_]));            \
      else                                                                                                                       \
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r_][t_]) : "a"(areg[r_][ai]), "v"(cur[t_]));            \
    } else {                                                                                                                     \
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r_][t_]) : "v"(areg[r_][ai]), "v"(cur[t_]));              \
    }                                                                                                                            \
  } while (0)
#define OSC_RDN(T)                                            \
  do {                                                        \
    if constexpr (has_next) OSC_RD_AT(nxt[T], nst, nsl, T);   \
  } while (0)
        if constexpr (group_end) {
          OSC_LGKM(0);
          if constexpr (fetch) OSC_VMC((LA - 1) * GK * NT - 1); else OSC_VMC(0);
          // raw barrier: __syncthreads() would make hipcc drain vmcnt, and with it the pieces meant to stay in flight
          __builtin_amdgcn_s_barrier();
        } else {
          OSC_LGKM(NT / 2);
        }
        // (written out per MFMA: asm operands inside a third level of generic lambdas trip clang's implicit captures)
#define OSC_STEP(J)                                                                                               \
  do {                                                                                                            \
    if constexpr ((J) < MM) {                                                                                     \
      if constexpr (!group_end && ((NT == 4 && (J) == 2) || (NT == 2 && (J) == 1))) OSC_LGKM(NT / 2);             \
      OSC_MFMA(J);                                                                                                \
      if constexpr (NT == 4 && (J) == 0) { OSC_RDN(0); OSC_RDN(1); }                                              \
      if constexpr (NT == 4 && (J) == 2) { OSC_RDN(2 % NT); OSC_RDN(3 % NT); }                                         \
      if constexpr (NT == 2 && (J) == 0) OSC_RDN(0);                                                              \
      if constexpr (NT == 2 && (J) == 1) OSC_RDN(1);                                                              \
      if constexpr (has_piece && (J) == (MM == 4 && NT == 4 ? 1 : MM - 2)) OSC_SETM0(fk, fst, pq);                \
      if constexpr (has_piece && (J) == MM - 1) OSC_DMA(boff[pq], psrc, fk);                                    \
    }                                                                                                             \
  } while (0)
        OSC_STEP(0);
        OSC_STEP(1);
        OSC_STEP(2);
        OSC_STEP(3);
        OSC_STEP(4);
        OSC_STEP(5);
        OSC_STEP(6);
        OSC_STEP(7);
      });
      pbase = nbase;
    };
    auto k_loop = [&](int ct, int pass) {
      if (ct + 1 == t1 && pass + 1 == NPASS) k_pass(std::true_type{});
      else k_pass(std::false_type{});
    };
    // SYM: take the coarse entries of row group r apart and deliver the hits to the buckets of their receiving rows (a lane
    // per entry).  Runs at the end of the item, and earlier whenever the list is more than half full.
    const int nbl = (t1 - t0) * 4;  // buckets the item's column tiles span: (t0 * 4 + b), b < nbl <= 4 TC_TILES
    int* const s_cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(lds) + PANEL_LDS + (size_t)4 * HB_CAP_SYM * 20 + (size_t)TC_TILES * 512) +
                       wave * (8 * TC_TILES);  // [2][4 TC_TILES]: counts / cursors, bases -- this wave's own
    int* const s_base = s_cnt + 4 * TC_TILES;
    auto deliver = [&](auto RC) {
      constexpr int r = decltype(RC)::value;

        const int n = min(wcnt[r], HCAP);
        const int rb = rbv[r];
        const int own = rb * 4 + wave;  // bucket of this wave's 32 rows
        if (wcnt[r] > HCAP && lane == 0) {  // dropped hits: row-side ones of these rows, column-side ones of any row of the chunk
          a.flags[chunk] = 1;
          atomicAdd(&a.bucket_cnt[own], a.bucket_cap + 1);
        }
        const unsigned* const hd = s_hdr + (wave * NRG + r) * HCAP;
        const v4f* const sc = s_sc + (wave * NRG + r) * HCAP;
        const float* const taur = &s_tau[wave][r][0][0];  // [half][row register]
        const int grow0 = rb * 128 + 32 * wave;
        // one lane per coarse entry: score t of the entry is a candidate of its row (row side) and / or of its column.
        // NT = 4: the entry is one row register's four subtile scores (row rl, subtiles 0-3).  NT = 2 (64-column passes): the
        // scores of a PAIR of row registers on the pass's two subtiles -- t = 0, 1: row rl, t = 2, 3: row rl + 1, subtile
        // 2 pass + (t & 1) -- so that an entry stands for four scores in either shape (one union test per four scores).
        auto row_of = [&](int rl, int t) { return NT == 4 ? rl : rl + (t >> 1); };
        auto sub_of = [&](int ps, int t) { return NT == 4 ? t : 2 * ps + (t & 1); };
        auto take_apart = [&](int e, int& rl, int& tl, int& l5, int& ps, v4f& sv, bool (&rp)[4], bool (&cp)[4]) {
          const bool valid = e < n;
          const unsigned hdv = valid ? hd[e] : 0u;
          rl = (int)((hdv >> 10) & 31u);
          ps = (int)((hdv >> 15) & 1u);
          tl = (int)((hdv >> 5) & 31u);
          l5 = (int)(hdv & 31u);
          sv = valid ? sc[e] : v4f{0.f, 0.f, 0.f, 0.f};
          const int ct = t0 + tl;
          v4f tcv = v4f{3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
          if (valid && ct > rb) tcv = *reinterpret_cast<const v4f*>(&s_tc[tl * 128 + l5 * 4]);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = row_of(rl, t), sub = sub_of(ps, t);
            const float trow = taur[((row >> 2) & 1) * 16 + (row & 3) + 4 * (row >> 3)];
            const int col = ct * 128 + 32 * sub + l5;
            rp[t] = valid && sv[t] > trow && col != grow0 + row && col < a.N;  // graph.py:37: no self-similarity; the ragged tail
            cp[t] = valid && sv[t] > tcv[sub];
          }
        };
        for (int b = lane; b < nbl; b += 64) s_cnt[b] = 0;
        int nrow = 0;
        for (int e0 = 0; e0 < n; e0 += 64) {  // (a wave's LDS accesses complete in order: no barrier between these passes)
          int rl, tl, l5, ps;
          v4f sv;
          bool rp[4], cp[4];
          take_apart(e0 + lane, rl, tl, l5, ps, sv, rp, cp);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            nrow += __popcll(__ballot(rp[t]));
            if (cp[t]) atomicAdd(&s_cnt[tl * 4 + sub_of(ps, t)], 1);
          }
        }
        int rbase = 0;
        if (lane == 0 && nrow > 0) rbase = atomicAdd(&a.bucket_cnt[own], nrow);
        for (int b = lane; b < nbl; b += 64) {
          const int c = s_cnt[b];
          s_base[b] = c > 0 ? atomicAdd(&a.bucket_cnt[t0 * 4 + b], c) : 0;
          s_cnt[b] = 0;
        }
        rbase = __builtin_amdgcn_readfirstlane(rbase);
        for (int e0 = 0; e0 < n; e0 += 64) {
          int rl, tl, l5, ps;
          v4f sv;
          bool rp[4], cp[4];
          take_apart(e0 + lane, rl, tl, l5, ps, sv, rp, cp);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int row = row_of(rl, t), sub = sub_of(ps, t);
            const unsigned long long m = __ballot(rp[t]);
            const int rpos = rbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (rp[t] && rpos < a.bucket_cap)
              a.bucket_ent[(size_t)own * a.bucket_cap + rpos] =
                  make_uint2(((unsigned)row << 27) | ROW_SIDE | (unsigned)((t0 + tl) * 128 + 32 * sub + l5), __float_as_uint(sv[t]));
            rbase += __popcll(m);
            if (cp[t]) {
              const int b = tl * 4 + sub;
              const int cpos = s_base[b] + atomicAdd(&s_cnt[b], 1);
              if (cpos < a.bucket_cap)
                a.bucket_ent[(size_t)(t0 * 4 + b) * a.bucket_cap + cpos] =
                    make_uint2(((unsigned)l5 << 27) | ROW_SIDE | (unsigned)(grow0 + row), __float_as_uint(sv[t]));
            }
          }
        }
          };
    if constexpr (MODE == 1 && SYM) {
      // One more pass than tiles: the pass behind the last tile only delivers what the lists still hold.  A list is
      // delivered BEFORE a tile's entries are appended whenever they would not fit (their count is known from the union
      // masks), so no entry is ever dropped unless ONE tile alone yields more than a whole list.
      for (int ct = t0; ct <= t1; ++ct) {
        const bool flush_only = ct == t1;
        for (int pass = 0; pass < (flush_only ? 1 : NPASS); ++pass) {  // (NT = 2: the tile's columns 0-63, then 64-127)
        if (!flush_only) {
          k_loop(ct, pass);
          asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // the asm MFMAs' results -> VALU reads: hipcc pads nothing for asm
        }
        static_for<0, NRG>([&](auto RC) {
          constexpr int r = decltype(RC)::value;
          if (!rok[r]) return;
#ifdef OSC_PANEL_NO_EPI  // measurement only (wrong lattice): the sweep without its hit test
          return;
#endif
          const bool test = !flush_only && ct >= rbv[r];  // (second row group of a set: the tile below its diagonal belongs to the first)
          constexpr int NU = NT == 4 ? 16 : 8;  // union tests: one per four scores of a lane
          unsigned long long fm[NU];
          int add = 0;
          if (test) {
            float tg[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const v4f t4 = *reinterpret_cast<const v4f*>(&s_tau[wave][r][h][4 * q]);
              tg[4 * q] = t4[0], tg[4 * q + 1] = t4[1], tg[4 * q + 2] = t4[2], tg[4 * q + 3] = t4[3];
            }
            float tcmin = 3.0e38f;  // the smallest threshold of this lane's columns in the pass (+inf: no column side on the diagonal tile)
            if (ct > rbv[r]) {
              // (the lane id recomputed in place: with two row groups at D = 768 every per-lane value that lives across the K loop
              // without being used in it is spilled, and a scratch reload here costs a vmcnt(0) -- a drain of the DMA ring -- per pass)
              unsigned l31o;
              asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_and_b32 %0, 31, %0" : "=v"(l31o));
              const v4f c4 = *reinterpret_cast<const v4f*>(&s_tc[(ct - t0) * 128 + l31o * 4]);
              if constexpr (NT == 4) tcmin = min2r(min3r(c4[0], c4[1], c4[2]), c4[3]);
              else tcmin = pass == 0 ? min2r(c4[0], c4[1]) : min2r(c4[2], c4[3]);
            }
            static_for<0, NU>([&](auto GC) {
              fm[decltype(GC)::value] = row_mask_u(GC, acc[r], tg, tcmin);
              add += __popcll(fm[decltype(GC)::value]);
            });
          }
          if (wcnt[r] > 0 && ((flush_only && pass == 0) || wcnt[r] + add > HCAP)) {
#ifndef OSC_PANEL_NO_DELIVER  // measurement only (wrong lattice): the lists are emptied, not delivered
            deliver(RC);
#endif
            wcnt[r] = 0;
          }
          if (test && add > 0) {
            // (LDS byte addresses of this (wave, row group)'s header and score lists: the low half of a flat LDS address)
            const unsigned hdb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(s_hdr + (wave * NRG + r) * HCAP));
            const unsigned scb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(s_sc + (wave * NRG + r) * HCAP));
            // (the lane's part of the header -- pass, tile of the item, lane, and the half's row offset 4 h -- in ONE register made
            // opaque to hipcc: left to itself it hoists the sixteen (row register | 4 h) constants out of every loop and, in the
            // two-row-group shape at D = 768, spills them: a scratch reload + vmcnt(0) -- a drain of the DMA ring -- per entry)
            unsigned tag, l31o_scratch;  // lane id -> (h << 12) | l31, recomputed in place (see the threshold fetch above)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %1, 7, %0\n\tv_and_b32 %0, 31, %0\n\tv_and_b32 %1, 0x1000, %1\n\tv_or_b32 %0, %1, %0"
                         : "=&v"(tag), "=&v"(l31o_scratch));
            tag |= (unsigned)(pass << 15) | (unsigned)((ct - t0) << 5);
            asm volatile("" : "+v"(tag));
            static_for<0, NU>([&](auto GC) {
              constexpr int g = NT == 4 ? decltype(GC)::value : 2 * decltype(GC)::value;  // (first) row register of the entry
              const unsigned long long m = fm[decltype(GC)::value];
              if (m != 0ull) {
                // The append of one union mask, 13 instructions written out (round 6; hipcc's own form of the same statement
                // took 21: a per-lane bit test against two lane-mask registers, four moves to pack the scores for one 16-byte
                // store, separate compare / and / saveexec / branch): exec <- the mask; position = list length + rank among the
                // mask's lanes; lanes beyond the list's end drop out; header by one store, the four scores by two ds_write2_b32
                // straight from their accumulator registers.
                constexpr unsigned hconst = (unsigned)((g & 3) + 8 * (g >> 2)) << 10;  // (disjoint bits: local row = (g & 3) + 8 (g >> 2) + 4 h)
                const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
                unsigned long long sv_;
                unsigned p_, a1_, a2_, hv_;
                const float s0_ = acc[r][0][g], s1_ = acc[r][1][g];
                const float s2_ = NT == 4 ? acc[r][2 % NT][g] : acc[r][0][g + 1], s3_ = NT == 4 ? acc[r][3 % NT][g] : acc[r][1][g + 1];
                asm volatile(
                    "s_and_saveexec_b64 %[sv], %[m]\n\t"
                    "v_mbcnt_lo_u32_b32 %[p], %[mlo], 0\n\t"
                    "v_mbcnt_hi_u32_b32 %[p], %[mhi], %[p]\n\t"
                    "v_add_u32 %[p], %[wc], %[p]\n\t"
                    "v_cmp_gt_u32 vcc, %[cap], %[p]\n\t"
                    "s_and_b64 exec, exec, vcc\n\t"
                    "v_lshl_add_u32 %[a1], %[p], 2, %[hdb]\n\t"
                    "v_lshl_add_u32 %[a2], %[p], 4, %[scb]\n\t"
                    "v_add_u32 %[hv], %[hc], %[tag]\n\t"
                    "ds_write_b32 %[a1], %[hv]\n\t"
                    "ds_write2_b32 %[a2], %[s0], %[s1] offset1:1\n\t"
                    "ds_write2_b32 %[a2], %[s2], %[s3] offset0:2 offset1:3\n\t"
                    "s_mov_b64 exec, %[sv]"
                    : [sv] "=&s"(sv_), [p] "=&v"(p_), [a1] "=&v"(a1_), [a2] "=&v"(a2_), [hv] "=&v"(hv_)
                    : [m] "s"(m), [mlo] "s"(mlo), [mhi] "s"(mhi), [wc] "s"(wcnt[r]), [cap] "n"(HCAP), [hdb] "s"(hdb), [scb] "s"(scb),
                      [hc] "n"(hconst), [tag] "v"(tag), [s0] "v"(s0_), [s1] "v"(s1_), [s2] "v"(s2_), [s3] "v"(s3_)
                    : "vcc", "memory");
                wcnt[r] += __popcll(m);
              }
            });
          }
        });
        }
      }
    } else if constexpr (MODE == 1) {
      static_assert(SYM || MODE != 1 || NT == 4, "the full sweep's fine entries are written for 128-column passes");
      for (int ct = t0; ct < t1; ++ct) {
        k_loop(ct, 0);
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // the asm MFMAs' results -> VALU reads: hipcc pads nothing for asm
#pragma unroll
        for (int r = 0; r < NRG; ++r) {
          if (!rok[r]) continue;
          float tg[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4f t4 = *reinterpret_cast<const v4f*>(&s_tau[wave][r][h][4 * q]);
            tg[4 * q] = t4[0], tg[4 * q + 1] = t4[1], tg[4 * q + 2] = t4[2], tg[4 * q + 3] = t4[3];
          }
          const float tc[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
          hit_test_tile(acc[r], tg, tc, ct, rbv[r], hitbuf[r], wcnt[r]);
        }
      }
    } else {
      static_assert(MODE != 0 || NT == 4, "the sample sweep's running maxima are written for 128-column passes");
      for (int ct = t0; ct < t1; ++ct) {
        k_loop(ct, 0);
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        // ---- tile maxima of the sample sweep --------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < NRG; ++r) {
#pragma unroll
          for (int g = 0; g < 16; ++g)
            taug[r][g] = fmaxf(taug[r][g], fmaxf(fmaxf(acc[r][0][g], acc[r][1][g]), fmaxf(acc[r][2][g], acc[r][3][g])));
          if ((ct + 1) % a.group_tiles == 0 || ct + 1 == t1) {  // close the group: maximum over its columns
            const int grp = ct / a.group_tiles;
            const int grow0 = rbv[r] * 128 + 32 * wave + 4 * h;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
              float m = taug[r][g];
#pragma unroll
              for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));  // within the 32 lanes of the half
              if (l31 == 0 && rok[r]) a.tmax[(size_t)(grow0 + (g & 3) + 8 * (g >> 2)) * a.ngroups + grp] = m;
              taug[r][g] = -3.0e38f;
            }
          }
        }
      }
    }
    if constexpr (MODE == 1 && !SYM) {  // flush the wave's hit lists (coalesced 8-byte stores) and their counts
#pragma unroll
      for (int r = 0; r < NRG; ++r) {
        if (!rok[r]) continue;
        const size_t li = list_of(r) * 4 + wave;
        const int n = min(wcnt[r], HCAP);
        uint2* out = a.hit_list + li * (size_t)a.hit_cap;
        for (int e = lane; e < n; e += 64) out[e] = hitbuf[r][e];
        if (lane == 0) a.hit_cnt[li] = wcnt[r];
      }
    }
    __syncthreads();  // every wave is done with the ring before the next item refills it
  }
}

// ---- the same thresholds-and-hits prefilter on a GEMM core that takes ANY depth: D > 768 ------------------------------
// A 32-row panel of more than 768 halfs does not fit a wave's registers, so here BOTH operands go through LDS: the
// 128 x 128 tile of the tile prefilter (knn_kernels.hip: k_knn_pref -- A tile + B tile of one 64-half K step per stage by
// LDS-DMA, two stages, one barrier per K step, the flat (column tile, K step) loop that keeps the next tile's first step
// in flight during the epilogue, two workgroups per CU), with k_panel's epilogues instead of the sorted register lists:
// MODE 0 tile-group maxima of the sample sweep, MODE 1 the symmetric half sweep -- threshold test of every accumulator
// against its row's and its column's tau, hits appended to per-wave LDS lists, delivered to the 32-row buckets at the
// end of the item.  Config 5's build (200k x 1536, k 64) spent 227 of its 257 ms in the list-maintaining full sweep of
// k_knn_pref; this sweep visits half the tiles and inserts nothing.  LDS: 64 KB of stages + 4 x TH_CAP entries + a
// threshold window of TT_TILES tiles = 78 KB, two workgroups per CU.
constexpr int TH_CAP = 320;
constexpr int TT_TILES = 8;
constexpr unsigned TILE_STAGE = (128 + 128) * 128;  // bytes of one stage: A tile, then B tile
constexpr size_t TILE_LDS = (size_t)2 * TILE_STAGE;
constexpr size_t TILE_LDS_HITS = TILE_LDS + (size_t)4 * TH_CAP * 8 + (size_t)TT_TILES * 512;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_tile_thr(const PanelArgs a, const int nkt) {
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  __shared__ int s_item, s_chunk, s_first;
  __shared__ __attribute__((aligned(16))) float s_tau[4][2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int frow = lane >> 3;
  const int swz = (l31 >> 1) & 7;
  const unsigned lds_base = (unsigned)(size_t)lds;
  const char* ldsc = reinterpret_cast<const char*>(lds);
  uint2* const hitbuf = reinterpret_cast<uint2*>(reinterpret_cast<char*>(lds) + TILE_LDS) + wave * TH_CAP;
  float* const s_tc = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + TILE_LDS + (size_t)4 * TH_CAP * 8);
  const size_t ldh = (size_t)a.ldh;
  auto chunk_sets = [&](int c) { return min(a.ntileB, (c + 1) * a.T); };
  int nitems = a.rb_count * a.S;
  if constexpr (MODE == 1) {
    nitems = 0;
    for (int c = 0; c < a.nchunks; ++c) nitems += chunk_sets(c);
  }
  auto glds16 = [&](const _Float16* src, unsigned dst_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst_bytes)
                 : "memory");
  };
  for (;;) {
    if (tid == 0) {
      const int it = (int)atomicAdd(a.queue, 1u) * a.item_stride + a.item_offset;
      s_item = it;
      if constexpr (MODE == 1) {
        int c = a.nchunks - 1, first = 0;
        if (it < nitems)
          while (it >= first + chunk_sets(c)) first += chunk_sets(c), --c;
        s_chunk = c;
        s_first = first;
      }
    }
    __syncthreads();
    const int item = s_item;
    const int chunk = MODE == 1 ? s_chunk : 0;
    const int first_item = MODE == 1 ? s_first : 0;
    __syncthreads();
    if (item >= nitems) break;
    int split, rbi, t0, t1;
    if constexpr (MODE == 1) {
      split = chunk;
      rbi = item - first_item;
      t1 = min(a.ntileB, (chunk + 1) * a.T);
      t0 = max(chunk * a.T, rbi);
    } else {
      split = item / a.rb_count;
      rbi = item - split * a.rb_count;
      t0 = split * a.tiles_per_split;
      t1 = min(a.ntileB, t0 + a.tiles_per_split);
    }
    const int rb = a.rb_begin + rbi;
    if (t0 >= t1) continue;
    float taug[16];
    int wcnt = 0;
    if constexpr (MODE == 0) {
#pragma unroll
      for (int g = 0; g < 16; ++g) taug[g] = -3.0e38f;
    } else {
      if (l31 < 16) s_tau[wave][h][l31] = a.tau[rb * 128 + 32 * wave + 4 * h + (l31 & 3) + 8 * (l31 >> 2)];
      for (int e = tid; e < (t1 - t0) * 128; e += 256) {  // the chunk's column thresholds: [tile][lane][subtile]
        const int col = t0 * 128 + e;
        s_tc[(e >> 7) * 128 + (e & 31) * 4 + ((e >> 5) & 3)] = col < a.N ? a.tau[col] : 3.0e38f;
      }
    }
    // LDS-DMA sources of this wave's pieces: rows 32 wave + 8 q + frow of the A and of the B tile, swizzled chunk
    const _Float16* a_src[4];
    const _Float16* b_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t off = (size_t)(32 * wave + 8 * q + frow) * ldh + ((lane & 7) ^ (((q & 1) << 2) | (frow >> 1))) * 8;
      a_src[q] = a.A + (size_t)rb * 128 * ldh + off;
      b_src[q] = a.B + off;
    }
    auto issue = [&](int stage, int ct, int kt) {
      const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)stage * TILE_STAGE + (unsigned)(32 * wave) * 128u);
      const unsigned b_dst = a_dst + 128u * 128u;
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(a_src[q] + kt * 64, a_dst + (unsigned)(8 * q) * 128u);
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(b_src[q] + (size_t)ct * 128 * ldh + kt * 64, b_dst + (unsigned)(8 * q) * 128u);
    };
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
    // ---- k_panel's hit test (one row group, both sides) -----------------------------------------------------------------
    auto row_mask = [&](auto GC, const float(&tg)[16], const float(&tc)[4]) -> unsigned long long {
      constexpr int g = decltype(GC)::value;
      bool any = fmaxf(fmaxf(acc[0][g], acc[1][g]), fmaxf(acc[2][g], acc[3][g])) > tg[g];
      any = any | (fmaxf(fmaxf(acc[0][g] - tc[0], acc[1][g] - tc[1]), fmaxf(acc[2][g] - tc[2], acc[3][g] - tc[3])) > 0.f);
      return __ballot(any);
    };
    auto hit_rows = [&](auto GC, const float(&tg)[16], const float(&tc)[4], int pct) {
      constexpr int g = decltype(GC)::value;
      const int rl = (g & 3) + 8 * (g >> 2) + 4 * h;
      const int grow = rb * 128 + 32 * wave + rl;
      const int cbase = pct * 128 + l31;
      const bool special = pct == rb || (pct + 1) * 128 > a.N;
      unsigned long long mk[4];
      unsigned side[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bool pred = acc[t][g] > tg[g];
        if (special) pred = pred && (cbase + 32 * t) != grow && (cbase + 32 * t) < a.N;
        const bool cp = acc[t][g] > tc[t];
        side[t] = (pred ? ROW_SIDE : 0u) | (cp ? COL_SIDE : 0u);
        mk[t] = __ballot(pred | cp);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (mk[t] == 0ull) continue;
        const bool pred = (mk[t] >> lane) & 1ull;
        const int pos = wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk[t] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk[t], 0u));
        if (pred && pos < TH_CAP)
          hitbuf[pos] = make_uint2(((unsigned)rl << 27) | side[t] | (unsigned)(cbase + 32 * t), __float_as_uint(acc[t][g]));
        wcnt += __popcll(mk[t]);
      }
    };
    // ---- flat (column tile, K step) pipeline: step s computes from stage s & 1 while step s + 1 lands in the other ------
    const int total = (t1 - t0) * nkt;
    int ct = t0, kt = 0, ict = t0, ikt = 0, issued = 0;
    auto issue_next = [&]() {
      issue(issued & 1, ict, ikt);
      ++issued;
      if (++ikt == nkt) {
        ikt = 0;
        ++ict;
      }
    };
    issue_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int step = 0; step < total; ++step) {
      const int stage = step & 1;
      if (issued < total) issue_next();
      const char* Asw = ldsc + (size_t)stage * TILE_STAGE + (size_t)(32 * wave + l31) * 128;
      const char* Bsw = ldsc + (size_t)stage * TILE_STAGE + 128 * 128 + (size_t)l31 * 128;
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const int co = ((2 * sl + h) ^ swz) * 16;
        const half8 av = *reinterpret_cast<const half8*>(Asw + co);
        half8 bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = *reinterpret_cast<const half8*>(Bsw + (size_t)t * 32 * 128 + co);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv[t], acc[t], 0, 0, 0);
      }
      if (kt == nkt - 1) {  // ---- the tile's accumulators are complete ----
        if constexpr (MODE == 0) {
#pragma unroll
          for (int g = 0; g < 16; ++g)
            taug[g] = fmaxf(taug[g], fmaxf(fmaxf(acc[0][g], acc[1][g]), fmaxf(acc[2][g], acc[3][g])));
          if ((ct + 1) % a.group_tiles == 0 || ct + 1 == t1) {
            const int grp = ct / a.group_tiles;
            const int grow0 = rb * 128 + 32 * wave + 4 * h;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
              float m = taug[g];
#pragma unroll
              for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
              if (l31 == 0) a.tmax[(size_t)(grow0 + (g & 3) + 8 * (g >> 2)) * a.ngroups + grp] = m;
              taug[g] = -3.0e38f;
            }
          }
        } else {
          float tg[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4f t4 = *reinterpret_cast<const v4f*>(&s_tau[wave][h][4 * q]);
            tg[4 * q] = t4[0], tg[4 * q + 1] = t4[1], tg[4 * q + 2] = t4[2], tg[4 * q + 3] = t4[3];
          }
          float tc[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
          if (ct > rb) {
            const v4f c4 = *reinterpret_cast<const v4f*>(&s_tc[(ct - t0) * 128 + l31 * 4]);
            tc[0] = c4[0], tc[1] = c4[1], tc[2] = c4[2], tc[3] = c4[3];
          }
          unsigned long long fm[16];
          static_for<0, 16>([&](auto GC) { fm[decltype(GC)::value] = row_mask(GC, tg, tc); });
          static_for<0, 16>([&](auto GC) {
            if (fm[decltype(GC)::value] != 0ull) hit_rows(GC, tg, tc, ct);
          });
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;
        kt = 0;
        ++ct;
      } else {
        ++kt;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next step has landed (and the epilogue's stores are out)
      __syncthreads();                                   // ... and every wave is done reading this stage
    }
    if constexpr (MODE == 1) {  // deliver the wave's entries to the buckets of their receiving rows (as k_panel's half sweep)
      const int nbl = (t1 - t0) * 4;
      int* const s_cnt = reinterpret_cast<int*>(s_tc) + wave * (8 * TT_TILES);
      int* const s_base = s_cnt + 4 * TT_TILES;
      const int n = min(wcnt, TH_CAP);
      const int own = rb * 4 + wave;
      if (wcnt > TH_CAP && lane == 0) {
        a.flags[chunk] = 1;
        atomicAdd(&a.bucket_cnt[own], a.bucket_cap + 1);
      }
      for (int b = lane; b < nbl; b += 64) s_cnt[b] = 0;
      int nrow = 0;
      for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        const unsigned x = e < n ? hitbuf[e].x : 0u;
        if (x & COL_SIDE) atomicAdd(&s_cnt[(int)((x & COL_MASK) >> 5) - t0 * 4], 1);
        nrow += __popcll(__ballot((x & ROW_SIDE) != 0u));
      }
      int rbase = 0;
      if (lane == 0 && nrow > 0) rbase = atomicAdd(&a.bucket_cnt[own], nrow);
      for (int b = lane; b < nbl; b += 64) {
        const int c = s_cnt[b];
        s_base[b] = c > 0 ? atomicAdd(&a.bucket_cnt[t0 * 4 + b], c) : 0;
        s_cnt[b] = 0;
      }
      rbase = __builtin_amdgcn_readfirstlane(rbase);
      const unsigned irow0 = (unsigned)(rb * 128 + 32 * wave);
      for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        const uint2 v = e < n ? hitbuf[e] : make_uint2(0u, 0u);
        const bool rs = (v.x & ROW_SIDE) != 0u, cs = (v.x & COL_SIDE) != 0u;
        const unsigned long long m = __ballot(rs);
        const int rpos = rbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (rs && rpos < a.bucket_cap) a.bucket_ent[(size_t)own * a.bucket_cap + rpos] = make_uint2((v.x & ~COL_SIDE), v.y);
        rbase += __popcll(m);
        if (cs) {
          const unsigned col = v.x & COL_MASK;
          const int b = (int)(col >> 5) - t0 * 4;
          const int cpos = s_base[b] + atomicAdd(&s_cnt[b], 1);
          if (cpos < a.bucket_cap)
            a.bucket_ent[(size_t)(col >> 5) * a.bucket_cap + cpos] = make_uint2(((col & 31u) << 27) | ROW_SIDE | (irow0 + (v.x >> 27)), v.y);
        }
      }
    }
    __syncthreads();  // the stages, the lists and the threshold window are free for the next item
  }
}

// fp32 unit rows -> fp16 image of 16 * Yn, zero beyond (N, D); image rows [r0, r1)
__global__ void k_panel_image(const float* Yn, int32_t ldn, _Float16* Yh, int32_t ldh, int32_t r0, int32_t r1, int32_t N, int32_t D,
                              int32_t mapped, KnnRowMap map) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 8 halfs
  const int per_row = ldh / 8;
  if (i >= (int64_t)(r1 - r0) * per_row) return;
  const int row = r0 + (int)(i / per_row), c0 = (int)(i % per_row) * 8;
  const int64_t src = row < N ? (mapped ? knn_map_lattice_row(map, N, row) : row) : 0;  // KnnPanelPlan::map
  half8 v;
  if (row < N && c0 + 8 <= D) {  // (the unit rows' pitch is a multiple of 32 floats: two 16-byte loads instead of eight dwords)
    const float4 a = *reinterpret_cast<const float4*>(Yn + (size_t)src * ldn + c0);
    const float4 b = *reinterpret_cast<const float4*>(Yn + (size_t)src * ldn + c0 + 4);
    v[0] = (_Float16)(16.0f * a.x), v[1] = (_Float16)(16.0f * a.y), v[2] = (_Float16)(16.0f * a.z), v[3] = (_Float16)(16.0f * a.w);
    v[4] = (_Float16)(16.0f * b.x), v[5] = (_Float16)(16.0f * b.y), v[6] = (_Float16)(16.0f * b.z), v[7] = (_Float16)(16.0f * b.w);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      v[j] = (row < N && c < D) ? (_Float16)(16.0f * Yn[(size_t)src * ldn + c]) : (_Float16)0.f;
    }
  }
  *(half8*)(Yh + (size_t)row * ldh + c0) = v;
}

__global__ void k_panel_sample(const _Float16* Yh, _Float16* Ys, int32_t ldh, int32_t m, int32_t N, int32_t gsz, int32_t G, int32_t mapped,
                               KnnRowMap map) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = ldh / 8;
  if (i >= (int64_t)m * per_row) return;
  const int r = (int)(i / per_row), c0 = (int)(i % per_row) * 8;
  // knn_rowmap.hpp: which lattice row the sample holds at position r; its fp16 row sits at that row's place in the image
  const int32_t row = knn_sample_lattice_row(knn_sample_index(r, m, gsz, G), m, N);
  const int64_t src = mapped ? knn_map_image_row(map, N, row) : row;
  *(half8*)(Ys + (size_t)r * ldh + c0) = *(const half8*)(Yh + (size_t)src * ldh + c0);
}

__device__ __forceinline__ unsigned order_key(unsigned bits) {  // ascending float order == ascending key order
  return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}

// one wave per row: tau = the rank-th largest of the row's ntile (<= 128) tile maxima -- a bitwise search over the order keys
// (32 ballot counts; round 6: the all-pairs rank count it replaces cost 128 shuffles + 512 compares per row, 0.18 ms at config 3)
__global__ __launch_bounds__(256) void k_panel_tau(const float* tmax, int32_t ntile, int32_t rank, int32_t N, float* tau) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  // (padding lanes: key 0, below the key of every float -- order_key never yields 0)
  const unsigned k0 = lane < ntile ? order_key(__float_as_uint(tmax[(size_t)row * ntile + lane])) : 0u;
  const unsigned k1 = lane + 64 < ntile ? order_key(__float_as_uint(tmax[(size_t)row * ntile + lane + 64])) : 0u;
  const bool two = ntile > 64;  // (wave-uniform)
  unsigned T = 0u;
  for (int b = 31; b >= 0; --b) {
    const unsigned cand = T | (1u << b);
    int ge = __popcll(__ballot(k0 >= cand));
    if (two) ge += __popcll(__ballot(k1 >= cand));
    if (ge >= rank) T = cand;
  }
  // T is the rank-th largest key (rank <= ntile: a real value's); back to the float it encodes
  if (lane == 0) tau[row] = __uint_as_float((T & 0x80000000u) ? (T & 0x7FFFFFFFu) : ~T);
}


// One workgroup per (row block, wave-of-32-rows, sub-range of those rows).  The rows' candidates sit in the S hit lists of
// their (row block, wave): the workgroup counting-sorts the entries of ITS rows by row into LDS (two passes over the
// lists), then each wave takes rows in turn: finds the keep-th largest fp16 score T by a bitwise search (32 ballot
// counts over the row's entries, held in registers) and writes keep candidates -- those above T, then entries equal to
// T -- so the last slot holds the list's minimum, the value k_knn_rescore's proof takes as v_last.  The list need not be
// sorted: the re-scoring ranks by exact score.
constexpr int SEL_CAP = 1024;    // candidates of one row the select can hold (expected: ~5 keep)
constexpr int SORT_CAP = 2560;   // entries one workgroup sorts (20 KB of LDS: several workgroups per CU)
// (Round 6, tried and dropped: ONE workgroup per bucket -- 9600 entries, 75 KB of LDS, the bucket read twice by one workgroup
// instead of twice by each of four: k_panel_select 0.55 -> 0.82 ms at config 3.  The kernel is bound by its per-row ballot
// searches, not by the 920 MB it fetches for 150 MB of entries; two workgroups per CU instead of eight lose more than the reads save.)
// Half-sweep builds (sym): every candidate of the 32 rows of (row block, wave) sits in ONE bucket (k_panel's flush).
struct SelectSym {
  int32_t on, T;
  const uint2* bucket_ent;
  const int32_t* bucket_cnt;
  int32_t bucket_cap;
  const int32_t* flags;
};
__global__ __launch_bounds__(256) void k_panel_select(const uint2* hit_list, const int32_t* hit_cnt, int32_t hit_cap, int32_t S,
                                                      int32_t rb_begin, int32_t rb_count, int32_t nsub, int32_t keep,
                                                      int32_t N, int32_t mapped, KnnRowMap map, float* cval, int32_t* cidx,
                                                      int32_t* fail_rows, int32_t* fail_count, const SelectSym sy, const int32_t sort_cap) {
  extern __shared__ __attribute__((aligned(16))) uint2 sorted[];  // sort_cap entries
  __shared__ int hist[32], start[33], cursor[32], s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = blockIdx.x % nsub, w = (blockIdx.x / nsub) % 4, rbi = blockIdx.x / (4 * nsub);
  const int rows_here = 32 / nsub, rl0 = sub * rows_here;  // local rows [rl0, rl0 + rows_here) of the wave's 32
  const int row_base = (rb_begin + rbi) * 128 + 32 * w;
  if (tid < 32) hist[tid] = 0;
  if (tid == 0) s_bad = sy.on && sy.flags[(rb_begin + rbi) / sy.T] != 0;  // lost column-side hits of this row block's rows
  __syncthreads();
  // the segments that hold this (row block, wave)'s entries: (list, entries, raw count)
  const int seg0 = 0;
  const int nseg = sy.on ? 1 : S;
  auto segment = [&](int sg, int& n, int& raw) -> const uint2* {
    if (sy.on) {
      const int b = (rb_begin + rbi) * 4 + w;  // (buckets are indexed by absolute row block: a sharded build selects per rank)
      raw = sy.bucket_cnt[b];
      n = min(raw, sy.bucket_cap);
      return sy.bucket_ent + (size_t)b * sy.bucket_cap;
    }
    const size_t li = ((size_t)sg * rb_count + rbi) * 4 + w;
    raw = hit_cnt[li];
    n = min(raw, hit_cap);
    return hit_list + li * (size_t)hit_cap;
  };
  // pass 1: entries per row
  for (int s = seg0; s < nseg; ++s) {
    int n, c;
    const uint2* list = segment(s, n, c);
    if (c > n && tid == 0) s_bad = 1;  // the list overflowed: hits were dropped
    for (int e = tid; e < n; e += 256) {
      const unsigned x = list[e].x;
      const int rl = (int)(x >> 27);
      if ((x & ROW_SIDE) && rl >= rl0 && rl < rl0 + rows_here) atomicAdd(&hist[rl], 1);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int r = 0; r < 32; ++r) {
      start[r] = acc;
      cursor[r] = acc;
      acc += hist[r];
    }
    start[32] = acc;
    if (acc > sort_cap) s_bad = 1;
  }
  __syncthreads();
  const bool bad = s_bad != 0;
  // pass 2: scatter into row order
  if (!bad) {
    for (int s = seg0; s < nseg; ++s) {
      int n, c;
      const uint2* list = segment(s, n, c);
      for (int e = tid; e < n; e += 256) {
        const uint2 v = list[e];
        const int rl = (int)(v.x >> 27);
        if ((v.x & ROW_SIDE) && rl >= rl0 && rl < rl0 + rows_here)
          sorted[atomicAdd(&cursor[rl], 1)] = make_uint2(v.x & COL_MASK, v.y);
      }
    }
  }
  __syncthreads();
  for (int rl = rl0 + wave; rl < rl0 + rows_here; rl += 4) {
    const int irow = row_base + rl;  // image row
    if (irow >= N) continue;
    const int row = mapped ? knn_map_lattice_row(map, N, irow) : irow;  // lattice row (KnnPanelPlan::map)
    const int m = bad ? 0 : hist[rl];
    float* ov = cval + (size_t)row * keep;
    int32_t* oi = cidx + (size_t)row * keep;
    if (bad || m < keep || m > SEL_CAP) {  // incomplete, too small or too large a candidate set: the exact kernel redoes it
      if (lane == 0) fail_rows[atomicAdd(fail_count, 1)] = row;
      for (int e = lane; e < keep; e += 64) oi[e] = -1;
      continue;
    }
    // (round 6: the registers a row's candidates take are a template constant of this part -- 4, 8 or 16 -- instead of sixteen
    // predicated rounds for every row: a config-3 row has ~190 candidates, three registers)
    auto select_row = [&](auto MC) {
      constexpr int M = decltype(MC)::value;
      const uint2* ent = sorted + start[rl];
      unsigned key[M], col[M], bits[M];
#pragma unroll
      for (int q = 0; q < M; ++q) {
        const int e = lane + 64 * q;
        key[q] = 0u;  // below every real key (scores here are > tau; order_key never yields 0 for them)
        col[q] = 0u;
        bits[q] = 0u;
        if (e < m) {
          const uint2 v = ent[e];
          col[q] = mapped ? (unsigned)knn_map_lattice_row(map, N, (int)v.x) : v.x;  // image column -> lattice column
          bits[q] = v.y;
          key[q] = order_key(v.y);
        }
      }
      const int mq = (m + 63) / 64;  // registers in use (wave-uniform)
      unsigned T = 0u;
      for (int b = 31; b >= 0; --b) {
        const unsigned cand = T | (1u << b);
        int ge = 0;
#pragma unroll
        for (int q = 0; q < M; ++q)
          if (q < mq) ge += __popcll(__ballot(key[q] >= cand));
        if (ge >= keep) T = cand;
      }
      int n_gt = 0;
#pragma unroll
      for (int q = 0; q < M; ++q)
        if (q < mq) n_gt += __popcll(__ballot(key[q] > T));
      int base_gt = 0, base_eq = n_gt;
#pragma unroll
      for (int q = 0; q < M; ++q) {
        if (q >= mq) continue;
        const bool gt = key[q] > T, eq = key[q] == T && key[q] != 0u;
        const unsigned long long bg = __ballot(gt), be = __ballot(eq);
        const int pg = base_gt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bg >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bg, 0u));
        const int pe = base_eq + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(be >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)be, 0u));
        if (gt) {
          ov[pg] = __uint_as_float(bits[q]);
          oi[pg] = (int32_t)col[q];
        }
        if (eq && pe < keep) {
          ov[pe] = __uint_as_float(bits[q]);
          oi[pe] = (int32_t)col[q];
        }
        base_gt += __popcll(bg);
        base_eq += __popcll(be);
      }
    };
    if (m <= 256) select_row(std::integral_constant<int, 4>{});
    else if (m <= 512) select_row(std::integral_constant<int, 8>{});
    else select_row(std::integral_constant<int, SEL_CAP / 64>{});
  }
}

// ---- the main sweep of the tile core with 64 x 128 wave tiles (round 4) ---------------------------------------------------
// k_tile_thr<1> reads 5 KB of LDS fragments per 4 MFMAs and wave (one A fragment, four B fragments per k16 slice); eight
// waves per CU ask the LDS for ~156 B per clock at full MFMA rate, which it does not deliver -- config 5's sweep ran at 31 %
// MFMA-busy.  Here a wave owns 64 rows x 128 columns (two row groups share every B fragment: 6 KB per 8 MFMAs, ~94 B per
// clock), a workgroup of 8 waves a 256 x 256 tile pair -- two row blocks against two column tiles --, 64 KB per stage, two
// stages, one workgroup per CU (still two waves per SIMD).  Everything else is k_tile_thr<1>: flat (tile pair, K step)
// pipeline, threshold test on both sides, fine 8-byte entries in per-(wave, row group) LDS lists that are delivered to the
// 32-row buckets when the next tile might not fit and at the end of the item.
constexpr int T2_CAP = 160;  // (the planner's measure of a list: fine entries; the kernel's lists hold T2_HC coarse ones since round 6)
constexpr int T2_HC = 72;    // coarse entries (20 bytes: header + a row register's four subtile scores) per (wave, row group) list
constexpr unsigned T2_STAGE = (256 + 256) * 128;
constexpr size_t T2_LDS = (size_t)2 * T2_STAGE;
constexpr size_t T2_LISTS = (size_t)16 * T2_HC * 20;  // [16 lists][T2_HC] headers, then [16][T2_HC] score quads
constexpr size_t T2_LDS_ALL = T2_LDS + T2_LISTS + (size_t)TT_TILES * 512 + (size_t)8 * 8 * TT_TILES * 4;

__global__ __launch_bounds__(512, 1) void k_tile_thr2(const PanelArgs a, const int nkt) {
  extern __shared__ __attribute__((aligned(1024))) float lds[];
  __shared__ int s_item, s_chunk, s_first;
  __shared__ __attribute__((aligned(16))) float s_tau[8][2][2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int frow = lane >> 3;
  const int swz = (l31 >> 1) & 7;
  const int rg = wave >> 1, cg = wave & 1;  // the wave's 64 rows (of the set's 256) and 128 columns (of the tile pair's 256)
  const unsigned lds_base = (unsigned)(size_t)lds;
  const char* ldsc = reinterpret_cast<const char*>(lds);
  unsigned* const t2_hdr = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds) + T2_LDS);
  v4f* const t2_sc = reinterpret_cast<v4f*>(reinterpret_cast<char*>(lds) + T2_LDS + (size_t)16 * T2_HC * 4);
  float* const s_tc = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + T2_LDS + T2_LISTS);
  int* const s_cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(lds) + T2_LDS + T2_LISTS + (size_t)TT_TILES * 512) +
                     wave * (8 * TT_TILES);
  int* const s_base = s_cnt + 4 * TT_TILES;
  const size_t ldh = (size_t)a.ldh;
  const int nrb = a.ntileB;
  auto chunk_sets = [&](int c) { return max(0, min((min(nrb, (c + 1) * a.T) + 1) / 2 - a.set0, a.nset)); };  // sets of this launch that sweep chunk c
  int nitems = 0;
  for (int c = 0; c < a.nchunks; ++c) nitems += chunk_sets(c);
  if (a.item_end > 0) nitems = min(nitems, a.item_end);  // (a window of column chunks: the streamed create)
  auto glds16 = [&](const _Float16* src, unsigned dst_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst_bytes)
                 : "memory");
  };
  for (;;) {
    if (tid == 0) {
      const int it = a.item_begin + (int)atomicAdd(a.queue, 1u) * a.item_stride + a.item_offset;
      int c = a.nchunks - 1, first = 0;
      if (it < nitems)
        while (it >= first + chunk_sets(c)) first += chunk_sets(c), --c;
      s_item = it;
      s_chunk = c;
      s_first = first;
    }
    __syncthreads();
    const int item = s_item, chunk = s_chunk, first_item = s_first;
    __syncthreads();
    if (item >= nitems) break;
    const int rb0 = (a.set0 + item - first_item) * 2;  // the set's row blocks rb0, rb0 + 1
    const int t1 = min(nrb, (chunk + 1) * a.T);
    const int t0 = max(chunk * a.T, rb0);          // (T and rb0 are even: tile pairs never straddle a chunk)
    if (t0 >= t1) continue;
    const int rb = rb0 + (rg >> 1);                // this wave's row block
    const bool rok = rb < nrb;
    const int wrow0 = rb * 128 + 64 * (rg & 1);    // first of the wave's 64 image rows
    int wcnt[2] = {0, 0};
    if (rok && l31 < 16) {
#pragma unroll
      for (int r = 0; r < 2; ++r) s_tau[wave][r][h][l31] = a.tau[wrow0 + 32 * r + 4 * h + (l31 & 3) + 8 * (l31 >> 2)];
    }
    for (int e = tid; e < (t1 - t0) * 128; e += 512) {  // the chunk's column thresholds: [tile][lane][subtile]
      const int col = t0 * 128 + e;
      s_tc[(e >> 7) * 128 + (e & 31) * 4 + ((e >> 5) & 3)] = col < a.N ? a.tau[col] : 3.0e38f;
    }
    // LDS-DMA sources: this wave fills rows 32 wave .. + 31 of the stage's A half (256 rows) and of its B half
    const _Float16* a_src[4];
    const _Float16* b_src[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t off = (size_t)(32 * wave + 8 * q + frow) * ldh + ((lane & 7) ^ (((q & 1) << 2) | (frow >> 1))) * 8;
      a_src[q] = a.A + (size_t)rb0 * 128 * ldh + off;  // (the image carries one zero tile behind its last row block)
      b_src[q] = a.B + off;
    }
    auto issue = [&](int stage, int ct, int kt) {
      const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)stage * T2_STAGE + (unsigned)(32 * wave) * 128u);
      const unsigned b_dst = a_dst + 256u * 128u;
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(a_src[q] + kt * 64, a_dst + (unsigned)(8 * q) * 128u);
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16(b_src[q] + (size_t)ct * 128 * ldh + kt * 64, b_dst + (unsigned)(8 * q) * 128u);
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[r][t][g] = 0.f;
    // Round 6: COARSE entries, as in k_panel's half sweep.  With fine 8-byte entries (a ballot, a rank and an LDS store per
    // subtile of every row register with a hit) the hit test was ~10 % of config 5's main sweep (127-132 ms over two builds;
    // 115.0 without any hit test, the accumulators kept alive); now 122.1, i.e. 6 %.  One union test per row register -- its best score against min(tau_row, the
    // smallest of the lane's four column thresholds) -- and one 20-byte entry {local row, tile of the item, lane; the four
    // subtile scores} per (register, lane) that fired, appended by 13 hand-written instructions; the delivery, which runs when a
    // tile's entries would not fit and at the end of the item with a lane per entry, decides every score against its own two
    // thresholds (row side / column side, the diagonal, the ragged tail) and writes the same bucket entries as before.
    auto deliver = [&](auto RC) {
      constexpr int r = decltype(RC)::value;
      const int nbl = (t1 - t0) * 4;
      const int n = min(wcnt[r], T2_HC);
      const int own = rb * 4 + 2 * (rg & 1) + r;
      if (wcnt[r] > T2_HC && lane == 0) {  // (only ONE tile that yields more than a whole list can overflow)
        a.flags[chunk] = 1;
        atomicAdd(&a.bucket_cnt[own], a.bucket_cap + 1);
      }
      const unsigned* const hd = t2_hdr + (wave * 2 + r) * T2_HC;
      const v4f* const sc = t2_sc + (wave * 2 + r) * T2_HC;
      const float* const taur = &s_tau[wave][r][0][0];  // [half][row register]
      const int grow0 = wrow0 + 32 * r;                 // image row of the list's local row 0
      auto take_apart = [&](int e, int& rl, int& tl, int& l5, v4f& sv, bool (&rp)[4], bool (&cp)[4]) {
        const bool valid = e < n;
        const unsigned hdv = valid ? hd[e] : 0u;
        rl = (int)((hdv >> 10) & 31u);
        tl = (int)((hdv >> 5) & 31u);
        l5 = (int)(hdv & 31u);
        sv = valid ? sc[e] : v4f{0.f, 0.f, 0.f, 0.f};
        const int ct = t0 + tl;
        const float trow = taur[((rl >> 2) & 1) * 16 + (rl & 3) + 4 * (rl >> 3)];
        v4f tcv = v4f{3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
        if (valid && ct > rb) tcv = *reinterpret_cast<const v4f*>(&s_tc[tl * 128 + l5 * 4]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int col = ct * 128 + 32 * t + l5;
          rp[t] = valid && sv[t] > trow && col != grow0 + rl && col < a.N;  // graph.py:37: no self-similarity; the ragged tail
          cp[t] = valid && sv[t] > tcv[t];
        }
      };
      for (int b = lane; b < nbl; b += 64) s_cnt[b] = 0;
      int nrow = 0;
      for (int e0 = 0; e0 < n; e0 += 64) {  // (a wave's LDS accesses complete in order: no barrier between these passes)
        int rl, tl, l5;
        v4f sv;
        bool rp[4], cp[4];
        take_apart(e0 + lane, rl, tl, l5, sv, rp, cp);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          nrow += __popcll(__ballot(rp[t]));
          if (cp[t]) atomicAdd(&s_cnt[tl * 4 + t], 1);
        }
      }
      int rbase = 0;
      if (lane == 0 && nrow > 0) rbase = atomicAdd(&a.bucket_cnt[own], nrow);
      for (int b = lane; b < nbl; b += 64) {
        const int c = s_cnt[b];
        s_base[b] = c > 0 ? atomicAdd(&a.bucket_cnt[t0 * 4 + b], c) : 0;
        s_cnt[b] = 0;
      }
      rbase = __builtin_amdgcn_readfirstlane(rbase);
      for (int e0 = 0; e0 < n; e0 += 64) {
        int rl, tl, l5;
        v4f sv;
        bool rp[4], cp[4];
        take_apart(e0 + lane, rl, tl, l5, sv, rp, cp);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned long long m = __ballot(rp[t]);
          const int rpos = rbase + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (rp[t] && rpos < a.bucket_cap)
            a.bucket_ent[(size_t)own * a.bucket_cap + rpos] =
                make_uint2(((unsigned)rl << 27) | ROW_SIDE | (unsigned)((t0 + tl) * 128 + 32 * t + l5), __float_as_uint(sv[t]));
          rbase += __popcll(m);
          if (cp[t]) {
            const int b = tl * 4 + t;
            const int cpos = s_base[b] + atomicAdd(&s_cnt[b], 1);
            if (cpos < a.bucket_cap)
              a.bucket_ent[(size_t)(t0 * 4 + b) * a.bucket_cap + cpos] =
                  make_uint2(((unsigned)l5 << 27) | ROW_SIDE | (unsigned)(grow0 + rl), __float_as_uint(sv[t]));
          }
        }
      }
      wcnt[r] = 0;
    };
    // the hit test of row group r on the tile pct: union masks, then (behind a delivery if the tile's entries would not fit) the append
    auto hit_test = [&](auto RC, int pct) {
      constexpr int r = decltype(RC)::value;
      float tg[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f t4 = *reinterpret_cast<const v4f*>(&s_tau[wave][r][h][4 * q]);
        tg[4 * q] = t4[0], tg[4 * q + 1] = t4[1], tg[4 * q + 2] = t4[2], tg[4 * q + 3] = t4[3];
      }
      auto max3r = [](float x, float y, float z) { float m; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(x), "v"(y), "v"(z)); return m; };
      auto max2r = [](float x, float y) { float m; asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y)); return m; };
      auto min3r = [](float x, float y, float z) { float m; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(x), "v"(y), "v"(z)); return m; };
      auto min2r = [](float x, float y) { float m; asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y)); return m; };
      float tcmin = 3.0e38f;  // (+inf: no column side on the diagonal tile)
      if (pct > rb) {
        const v4f c4 = *reinterpret_cast<const v4f*>(&s_tc[(pct - t0) * 128 + l31 * 4]);
        tcmin = min2r(min3r(c4[0], c4[1], c4[2]), c4[3]);
      }
      unsigned long long fm[16];
      int add = 0;
      static_for<0, 16>([&](auto GC) {
        constexpr int g = decltype(GC)::value;
        fm[g] = __ballot(max2r(max3r(acc[r][0][g], acc[r][1][g], acc[r][2][g]), acc[r][3][g]) > min2r(tg[g], tcmin));
        add += __popcll(fm[g]);
      });
#ifdef OSC_TILE2_NODELIVER  // measurement only (wrong lattice): full lists are emptied, not delivered
      if (wcnt[r] > 0 && wcnt[r] + add > T2_HC) wcnt[r] = 0;
#else
      if (wcnt[r] > 0 && wcnt[r] + add > T2_HC) deliver(RC);  // (resets the count)
#endif
      if (add == 0) return;
      const unsigned hdb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(t2_hdr + (wave * 2 + r) * T2_HC));
      const unsigned scb = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(t2_sc + (wave * 2 + r) * T2_HC));
      const unsigned tag = (unsigned)(h << 12) | (unsigned)((pct - t0) << 5) | (unsigned)l31;
      static_for<0, 16>([&](auto GC) {
        constexpr int g = decltype(GC)::value;
        const unsigned long long m = fm[g];
        if (m != 0ull) {  // (k_panel's append, instruction for instruction)
          constexpr unsigned hconst = (unsigned)((g & 3) + 8 * (g >> 2)) << 10;  // (disjoint bits: local row = (g & 3) + 8 (g >> 2) + 4 h)
          const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
          unsigned long long sv_;
          unsigned p_, a1_, a2_, hv_;
          const float s0_ = acc[r][0][g], s1_ = acc[r][1][g], s2_ = acc[r][2][g], s3_ = acc[r][3][g];
          asm volatile(
              "s_and_saveexec_b64 %[sv], %[m]\n\t"
              "v_mbcnt_lo_u32_b32 %[p], %[mlo], 0\n\t"
              "v_mbcnt_hi_u32_b32 %[p], %[mhi], %[p]\n\t"
              "v_add_u32 %[p], %[wc], %[p]\n\t"
              "v_cmp_gt_u32 vcc, %[cap], %[p]\n\t"
              "s_and_b64 exec, exec, vcc\n\t"
              "v_lshl_add_u32 %[a1], %[p], 2, %[hdb]\n\t"
              "v_lshl_add_u32 %[a2], %[p], 4, %[scb]\n\t"
              "v_add_u32 %[hv], %[hc], %[tag]\n\t"
              "ds_write_b32 %[a1], %[hv]\n\t"
              "ds_write2_b32 %[a2], %[s0], %[s1] offset1:1\n\t"
              "ds_write2_b32 %[a2], %[s2], %[s3] offset0:2 offset1:3\n\t"
              "s_mov_b64 exec, %[sv]"
              : [sv] "=&s"(sv_), [p] "=&v"(p_), [a1] "=&v"(a1_), [a2] "=&v"(a2_), [hv] "=&v"(hv_)
              : [m] "s"(m), [mlo] "s"(mlo), [mhi] "s"(mhi), [wc] "s"(wcnt[r]), [cap] "n"(T2_HC), [hdb] "s"(hdb), [scb] "s"(scb),
                [hc] "n"(hconst), [tag] "v"(tag), [s0] "v"(s0_), [s1] "v"(s1_), [s2] "v"(s2_), [s3] "v"(s3_)
              : "vcc", "memory");
          wcnt[r] += __popcll(m);
        }
      });
    };
    // ---- flat (tile pair, K step) pipeline ---------------------------------------------------------------------------------
    const int npair = (t1 - t0 + 1) / 2;
    const int total = npair * nkt;
    int ct = t0, kt = 0, ict = t0, ikt = 0, issued = 0;
    auto issue_next = [&]() {
      issue(issued & 1, ict, ikt);
      ++issued;
      if (++ikt == nkt) {
        ikt = 0;
        ict += 2;
      }
    };
    issue_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int step = 0; step < total; ++step) {
      const int stage = step & 1;
#ifndef OSC_TILE2_NODMA  // measurement only (wrong lattice): the sweep on stale stages -- what the staging costs it
      if (issued < total) issue_next();
#endif
      const char* Asw = ldsc + (size_t)stage * T2_STAGE + (size_t)(64 * rg + l31) * 128;
      const char* Bsw = ldsc + (size_t)stage * T2_STAGE + 256 * 128 + (size_t)(128 * cg + l31) * 128;
#ifdef OSC_TILE2_NORD  // measurement only (wrong lattice): fragments read once per K step instead of per k16 slice
      half8 av[2], bv[4];
#endif
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const int co = ((2 * sl + h) ^ swz) * 16;
#ifndef OSC_TILE2_NORD
        half8 av[2], bv[4];
#else
        if (sl == 0) {
#endif
#pragma unroll
        for (int r = 0; r < 2; ++r) av[r] = *reinterpret_cast<const half8*>(Asw + (size_t)r * 32 * 128 + co);
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = *reinterpret_cast<const half8*>(Bsw + (size_t)t * 32 * 128 + co);
#ifdef OSC_TILE2_NORD
        }
#endif
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[r], bv[t], acc[r][t], 0, 0, 0);
      }
      if (kt == nkt - 1) {  // ---- the tile pair's accumulators are complete ----
        const int pct = ct + cg;  // this wave's column tile
#ifndef OSC_TILE2_NOEPI  // measurement only (wrong lattice): the sweep without its hit test
        if (rok && pct >= rb && pct < t1) {
          static_for<0, 2>([&](auto RC) { hit_test(RC, pct); });  // (delivers the list first when the tile's entries would not fit)
        }
#else  // (the accumulators stay "used": without this hipcc deletes the MFMAs and their fragment reads with the hit test)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("" ::"v"(acc[r][t]));
#endif
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[r][t][g] = 0.f;
        kt = 0;
        ct += 2;
      } else {
        ++kt;
      }
#ifndef OSC_TILE2_NOBAR  // measurement only (wrong lattice): no wait for the staged K step, no workgroup barrier
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#endif
    }
    if (rok) {
      static_for<0, 2>([&](auto RC) {
        if (wcnt[decltype(RC)::value] > 0) deliver(RC);
      });
    }
    __syncthreads();  // the stages, the lists and the threshold window are free for the next item
  }
}

// ---- second-stage proof for rows the first re-scoring could not decide (half-sweep builds) ---------------------------
// k_knn_rescore proves a row from its `keep` best candidates: "every left-out column has fp16 score <= the list's last".  On
// clustered anchors that often fails -- the exact k-th score is not delta above the keep-th fp16 score -- and such rows used
// to go to the all-fp32 kernel (4-6 ms for ~1 % of the rows of a 100k lattice).  But the row's BUCKET holds every column
// whose fp16 score beat tau_row, so a far weaker statement is available: "every column outside the bucket has fp16 score
// <= tau_row".  One wave per undecided row re-scores ALL its bucket candidates exactly (same products, same order and
// butterfly as k_knn_rescore: bit-identical scores), takes the k best by (score desc, index asc) and accepts the row iff
// tau_row / 256 + delta < its exact k-th score.  Rows it cannot decide either (fewer than k candidates, > WIDE_CAP, a
// bucket that overflowed, a k-th score within delta of tau) stay on the list for the exact kernel.
constexpr int WIDE_CAP = 1024;  // candidates of one row (= SEL_CAP)
__global__ __launch_bounds__(256) void k_bucket_rescore(const float* __restrict__ Yn, int32_t ldn, int32_t N, const int32_t* rows_in,
                                                        int32_t nrows, const uint2* bucket_ent, const int32_t* bucket_cnt,
                                                        int32_t bucket_cap, const int32_t* flags, int32_t T, const float* tau,
                                                        int32_t mapped, KnnRowMap map, int32_t k, float delta,
                                                        float* out_val, int32_t* out_idx, int32_t* fail_rows, int32_t* fail_count) {
  __shared__ int s_idx[4][WIDE_CAP];
  __shared__ float s_sc[4][WIDE_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = blockIdx.x * 4 + wave;
  if (slot >= nrows) return;
  const int row = rows_in[slot];
  auto give_up = [&]() {
    if (lane == 0) fail_rows[atomicAdd(fail_count, 1)] = row;
  };
  const int irow = mapped ? knn_map_image_row(map, N, row) : row;  // image row of this lattice row (KnnPanelPlan::map)
  const int b = irow >> 5, rl = irow & 31;
  const int raw = bucket_cnt[b];
  if (raw > bucket_cap || flags[(irow >> 7) / T] != 0) {  // hits of this row were lost: nothing can be concluded from the bucket
    give_up();
    return;
  }
  // 1. the row's candidates (image columns -> lattice columns)
  const uint2* ent = bucket_ent + (size_t)b * bucket_cap;
  int n = 0;
  for (int e0 = 0; e0 < raw; e0 += 64) {
    const int e = e0 + lane;
    const unsigned x = e < raw ? ent[e].x : 0u;
    const bool mine = (x & ROW_SIDE) != 0u && (int)(x >> 27) == rl;
    const unsigned long long m = __ballot(mine);
    const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (mine && pos < WIDE_CAP) s_idx[wave][pos] = mapped ? knn_map_lattice_row(map, N, (int)(x & COL_MASK)) : (int)(x & COL_MASK);
    n += __popcll(m);
  }
  if (n < k || n > WIDE_CAP) {
    give_up();
    return;
  }
  // 2. exact scores (k_knn_rescore's arithmetic: lanes stride the row, fmaf chain per lane, butterfly sum)
  const float* yi = Yn + (size_t)row * ldn;
  for (int q0 = 0; q0 < n; q0 += 4) {
    float ss[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ss[u] = 0.f;
      if (q0 + u < n) {
        const float* yj = Yn + (size_t)s_idx[wave][q0 + u] * ldn;
        float acc = 0.f;
        for (int c = lane * 4; c < ldn; c += 256) {
          const float4 a = *reinterpret_cast<const float4*>(yi + c), bb = *reinterpret_cast<const float4*>(yj + c);
          acc = fmaf(a.x, bb.x, acc);
          acc = fmaf(a.y, bb.y, acc);
          acc = fmaf(a.z, bb.z, acc);
          acc = fmaf(a.w, bb.w, acc);
        }
        ss[u] = acc;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss[u] += __shfl_xor(ss[u], o, 64);
      if (lane == 0 && q0 + u < n) s_sc[wave][q0 + u] = ss[u];
    }
  }
  // 3. rank of every candidate among all (score desc, index asc); the k best are the row's list
  float tk = -3.0e38f;
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int c = c0 + lane;
    const float sv = c < n ? s_sc[wave][c] : 0.f;
    const int iv = c < n ? s_idx[wave][c] : 0;
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float ov = s_sc[wave][j];
      const int oi = s_idx[wave][j];
      rank += (ov > sv || (ov == sv && oi < iv)) ? 1 : 0;
    }
    if (c < n && rank < k) {
      out_val[(size_t)row * k + rank] = fmaxf(sv, 0.f);
      out_idx[(size_t)row * k + rank] = iv;
      if (rank == k - 1) tk = sv;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tk = fmaxf(tk, __shfl_xor(tk, o, 64));
  if (!(tau[irow] * (1.0f / 256.0f) + delta < tk)) give_up();
}

template <int MODE>
void launch_tile_thr(const PanelArgs& a, int nkt, int grid, hipStream_t s) {
  constexpr size_t lds_bytes = MODE == 1 ? TILE_LDS_HITS : TILE_LDS;
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_thr<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes));
  hipLaunchKernelGGL((k_tile_thr<MODE>), dim3(grid), dim3(256), lds_bytes, s, a, nkt);
  HIP_CHECK(hipGetLastError());
}

template <int MODE>
void launch_panel(const PanelArgs& a, int nkt, int nrg, bool sym, int grid, hipStream_t s) {
#define OSC_PANEL(NKT, NRG, SYM)                                                                                    \
  do {                                                                                                              \
    constexpr size_t lds_bytes = MODE == 1 ? PANEL_LDS_HITS : PANEL_LDS;                                            \
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<NKT, MODE, NRG, SYM>),                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                     \
    hipLaunchKernelGGL((k_panel<NKT, MODE, NRG, SYM>), dim3(grid), dim3(256), lds_bytes, s, a);                    \
  } while (0)
  if constexpr (MODE == 1) {
    if (sym) {
      if (nkt == 6 && nrg == 2) OSC_PANEL(6, 2, true);
      else if (nkt == 6 && nrg == 1) OSC_PANEL(6, 1, true);
      else if (nkt == 12 && nrg == 1) OSC_PANEL(12, 1, true);
      else if (nkt == 12 && nrg == 2) OSC_PANEL(12, 2, true);
      else throw std::runtime_error("launch_panel: unsupported K depth / row groups");
      HIP_CHECK(hipGetLastError());
      return;
    }
  }
  if (nkt == 6 && nrg == 2) OSC_PANEL(6, 2, false);
  else if (nkt == 6 && nrg == 1) OSC_PANEL(6, 1, false);
  else if (nkt == 12 && nrg == 1) OSC_PANEL(12, 1, false);
  else throw std::runtime_error("launch_panel: unsupported K depth / row groups");
#undef OSC_PANEL
  HIP_CHECK(hipGetLastError());
}

}  // namespace

int knn_panel_nkt(int32_t D) { return D <= 384 ? 6 : D <= 768 ? 12 : 0; }
// K steps of the tile core (both operands through LDS): any depth; used beyond 768 columns, up to 4096
int knn_tile_nkt(int32_t D) { return (D > 768 && D <= 4096) ? (D + 63) / 64 : 0; }

void knn_panel_set_pieces(KnnPanelPlan& p, int32_t N, const int32_t* starts, int npieces) {
  p.map = knn_row_map(N, starts, npieces, p.scatter != 1);
}

KnnPanelPlan knn_panel_plan(int32_t N, int32_t D, int32_t keep, int32_t cus, bool scatter_rows, bool sym,
                            const KnnPanelTune& tune) {
  KnnPanelPlan p{};
  p.sym = sym;
  p.map = knn_row_map(N, nullptr, 1, scatter_rows);
  p.scatter = p.map.a[0];
  p.nkt = knn_panel_nkt(D);
  p.tile_core = false;
  if (p.nkt == 0 && sym && knn_tile_nkt(D) != 0) {  // D > 768: the tile core under the same thresholds / buckets / select
    p.nkt = knn_tile_nkt(D);
    p.tile_core = true;
  }
  p.ldh = 64 * p.nkt;
  // row groups per wave: two where the panel is small enough (D <= 384: 2 x 96 registers), see the file header
  p.npad = ((N + 127) / 128) * 128;
  p.nrb = p.npad / 128;
  // (round 6: two row groups also at K depth 12 in the half sweep -- k_panel<12, 1, 2, true>, 64-column passes -- from 720 row
  // blocks on: gemm_topk at D = 768, one / two groups, profiles/r06_knn_nrg_ab.txt: 40 000 rows 2.05 / 2.23 ms, 60 000 3.64 / 3.73,
  // 80 000 6.19 / 6.20, 100 000 9.50 / 8.97, 140 000 x 640 16.9 / 15.8, 200 000 33.5 / 31.4 -- the items are twice as long and
  // half as many, which smaller lattices pay for at the tail of the persistent grid)
  p.nrg = p.tile_core ? 1 : p.nkt == 6 ? (tune.nrg != 1 ? 2 : 1) : (p.nkt == 12 && sym && (tune.nrg == 2 || (tune.nrg == 0 && p.nrb >= 720))) ? 2 : 1;
  p.nrg_s = p.nkt == 6 ? p.nrg : 1;
  p.keep = keep;
  // Thresholds: the sample holds one column in rho; tau = the 14th largest tile maximum ~ the 15th-17th best sample
  // score, so about rho * 16 columns of the full sweep beat it (gamma-distributed: 0.1 % of the rows see fewer than
  // rho * 6 or more than rho * 32).  rho = keep / 4 puts `keep` at the low tail and 8 keep slots above the high one;
  // OSC_KNN_PANEL_RHO overrides it.
  const double rho_env = tune.rho;
  // (round 3, config 4's shape, keep = 24: rho 6 / 8 / 12 / 16 -> build 798 / 790 / 775 / 784 ms with 6 / 1 / 1 / 129 rows sent
  // to the exact kernel; from 24 on the folded group maxima put tau so low that every hit list overflows.  Config 3,
  // keep = 48: 8 / 12 / 16 -> 26.5 / 22.2 / 26.8 ms.  Hence at least 12.)
  // (round 6: beyond 4800 row blocks at most 400 sample tiles -- 51 200 columns -- while rho stays <= 20.  The thresholds are order statistics of <= 128 group maxima, which a
  // denser sample of a very large lattice does not sharpen, while the sample sweep costs nrb / rho tile passes per row block:
  // config 4 (7813 row blocks) rho 12 / 16 / 20 / 24 -> build 403 / 399 / 394 / 396 ms, 0 fallback rows each
  // (scripts/exp/r06/c4_sweep.py); up to 4800 row blocks nothing changes; never beyond 20: 1.5 M x 256 at rho = 29 sent 111
  // rows to the exact kernel and built slower -- profiles/r06_rho_cap.txt)
  const double rho = rho_env > 0.0 ? rho_env : std::max(std::max(12.0, keep / 4.0), std::min(20.0, p.nrb / 400.0));
  p.sample_tiles = (int32_t)std::max(24.0, std::min(p.nrb / 2.0, std::round(p.nrb / rho)));
  // tile maxima are folded over groups of consecutive sample tiles so that a row has at most 128 of them (the r-th
  // largest group maximum is still a lower bound of the r-th best sample score)
  p.group_tiles = (p.sample_tiles + 127) / 128;
  p.sample_groups = (p.sample_tiles + p.group_tiles - 1) / p.group_tiles;
  // The threshold rank.  tau = the r-th largest of G group maxima lets (N / sample columns) G -ln(1 - r / G) columns of a row
  // through in expectation; the target is 4 keep of them (= 16 rho at rho = keep / 4; at least 192), which is rank 14 at
  // configs 3 / 4 / 5 (G = 65; measured there: rank 16 / 14 / 12 / 10 -> 2 / 2 / 21 / 346 rows short of candidates).  Where
  // the sample cannot be that sparse (small lattices: at least 24 sample tiles) the rank rises with the need, up to 0.9 G;
  // the route is offered as long as even that leaves a row twice `keep` candidates.  (Until round 4 the route was held
  // to N >= 16384 and samples of < keep / 8 density: rows short of candidates went to the exact kernel then; now they are
  // proven from their buckets, and between 8192 and 16384 rows this route builds in half the tile prefilter's time:
  // 12000 x 768, k 16: 1.70 -> 0.79 ms; 15000 x 768, k 32: 3.20 -> 1.30 ms.)
  {
    const double G = (double)p.sample_groups, per_col = (double)p.nrb / p.sample_tiles;
    const double target = std::max(4.0 * keep, 192.0);
    const double rmax = std::floor(0.9 * G);
    double r = std::ceil(G * (1.0 - std::exp(-target / (per_col * G))) - 0.25);
    r = std::max(std::min(14.0, rmax), std::min(r, rmax));
    p.sample_rank = (int32_t)r;
    p.ok = per_col * G * -std::log(1.0 - rmax / G) >= 2.0 * keep;
  }
  if (tune.rank > 0) p.sample_rank = std::max(2, std::min(p.sample_groups, tune.rank));
  // column splits: whatever leaves the smallest idle tail on `cus` persistent workgroups (per-item overhead ~1 %)
  // ... and few enough hits per wave and item for its LDS list: 32 rows x ~5 keep / S <= ~2/3 of HB_CAP
  p.hit_cap = HB_CAP / p.nrg;
  const int s_min = std::max(1, (int)std::ceil(32.0 * std::max(5.0 * keep, 20.0 * rho) / (0.66 * p.hit_cap)));
  const int nsets = (p.nrb + p.nrg - 1) / p.nrg;  // work items per split
  double best = 1e30;
  p.S = s_min;
  for (int S = s_min; S <= s_min + 8; ++S) {
    if (p.nrb / S < 16 && S > s_min) break;
    const double rounds = (double)nsets * S / std::max(1, cus);
    const double cost = std::ceil(rounds) / rounds * (1.0 + 0.01 * S);
    if (cost < best - 1e-9) {
      best = cost;
      p.S = S;
    }
  }
  p.tiles_per_split = (p.nrb + p.S - 1) / p.S;
  // Candidates a row's threshold lets through, in expectation: tau is the r-th largest of G group maxima, so r of G groups
  // hold a sample column above it; with lambda such columns per group, 1 - exp(-lambda) = r / G, and every sample column
  // stands for nrb / sample_tiles columns of the sweep.  G = 65 (config 3): 190 (the "16 rho" of round 2); G = 26 (N = 40 000):
  // 242 -- a bound of max(5 keep, 20 rho) = 240 then sized the 32-row buckets below their mean load and sent whole
  // buckets to the exact kernel (8768 fallback rows on clustered anchors at N = 40 000, D = 256, k = 24).
  {
    const double G = (double)p.sample_groups, r = std::min((double)p.sample_rank, 0.95 * G);
    const double mean_hits = (double)p.nrb / p.sample_tiles * G * -std::log(1.0 - r / G);
    p.hit_bound = std::max(std::max(5.0 * keep, 20.0 * rho), 1.25 * mean_hits);
  }
  if (sym) {
    // half sweep: column chunks of T tiles; a (wave, item) list takes the row-side AND the column-side hits of its tiles,
    // 2 x 32 rows x bound / nrb per tile, and should stay within ~2/3 of its LDS list; the item's column thresholds must
    // fit their LDS window (TC_TILES); T even where a set holds two row blocks (both then meet their diagonal in one chunk)
    // (k_tile_thr2 delivers a list before a tile that might not fit: a tile's mean load must stay well below the 96 entries
    // that check leaves it -- small N with a deep k goes through k_tile_thr<1>, whose lists hold a whole item)
    p.tile_wide = p.tile_core && tune.tile_wide != 0 && 64.0 * p.hit_bound / p.nrb <= 40.0;
    p.hit_cap = p.tile_core ? (p.tile_wide ? T2_CAP : TH_CAP) : HB_CAP_SYM / p.nrg;
    const double bound = p.hit_bound;
    int T = (int)std::floor(0.66 * p.hit_cap * p.nrb / (64.0 * bound));  // (the tile core's lists are a hard limit)
    if (!p.tile_core) T = TC_TILES;  // k_panel delivers a list that is half full: the threshold window alone limits an item
    T = std::max(p.tile_core ? 1 : 2, std::min(p.tile_core ? TT_TILES : TC_TILES, T));
    if (tune.T > 0) T = std::max(p.tile_core ? 1 : 2, std::min(p.tile_core ? TT_TILES : TC_TILES, tune.T));  // (A/B: tiles per chunk)
    if (p.tile_wide) T = TT_TILES;  // (its lists are delivered as they fill: the threshold window alone limits an item)
    if (p.nrg == 2 || p.tile_wide) T &= ~1;
    p.T = T;
    p.S = (p.nrb + T - 1) / T;  // chunks
    p.tiles_per_split = T;
    p.bucket_cap = (int32_t)(32.0 * 1.5 * bound);  // a group of 32 rows receives all its candidates here; 1.5: rows of a group vary, clustered anchors have heavy tails
    p.nitems = 0;
    const int per_item = p.tile_wide ? 2 : p.nrg;  // row blocks of one work item
    if (p.tile_wide) {  // groups of sets whose image rows fit the Infinity Cache next to the passing chunks (equal sizes)
      const int nsets = (p.nrb + 1) / 2;
      const double group_mb = tune.tile_group_mb > 0 ? tune.tile_group_mb : 128.0;
      const int ngroups = std::max(1, (int)std::ceil(nsets * 256.0 * p.ldh * 2.0 / (group_mb * 1048576.0)));
      p.tile_group_sets = (nsets + ngroups - 1) / ngroups;
    }
    for (int c = 0; c < p.S; ++c) p.nitems += (std::min(p.nrb, (c + 1) * T) + per_item - 1) / per_item;
  }
  // phase A: splits of whole tile groups, again for the tail of the persistent grid
  // (round 4: grouping a row's sample columns by (column split, lane) instead -- 32 running maxima per split, no shuffles and
  // no stores inside the sweep -- took 0.14 ms off config 3's 1.8 ms sample sweep but moved the thresholds: 16 instead of 1
  // rows short of candidates at N = 20000, config 4's build 412 -> 419 ms; not kept)
  best = 1e30;
  p.SA = 1;
  const int nsets_s = (p.nrb + p.nrg_s - 1) / p.nrg_s;
  for (int S = 1; S <= 6 && S <= p.sample_groups; ++S) {
    const double rounds = (double)nsets_s * S / std::max(1, cus);
    const double cost = std::ceil(rounds) / rounds * (1.0 + 0.03 * S);
    if (cost < best - 1e-9) {
      best = cost;
      p.SA = S;
    }
  }
  if (tune.sa > 0) p.SA = std::max(1, std::min(tune.sa, p.sample_groups));
  p.sample_tiles_per_split = ((p.sample_groups + p.SA - 1) / p.SA) * p.group_tiles;
  return p;
}

void launch_panel_image(const float* Yn, int32_t ldn, void* Yh, const KnnPanelPlan& p, int32_t N, int32_t D, hipStream_t s, int32_t r0,
                        int32_t r1) {
  if (r1 < 0) r1 = p.npad;
  if (r0 < 0 || r1 > p.npad || r1 <= r0) return;
  const int64_t n = (int64_t)(r1 - r0) * (p.ldh / 8);
  hipLaunchKernelGGL(k_panel_image, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Yn, ldn, static_cast<_Float16*>(Yh), p.ldh, r0, r1,
                     N, D, p.scatter != 1 ? 1 : 0, p.map);
  HIP_CHECK(hipGetLastError());
}

void launch_panel_sample_rows(const float* Yn_rows, int32_t ldn, void* Ys, const KnnPanelPlan& p, int32_t rows, int32_t D, hipStream_t s) {
  if (rows != p.sample_tiles * 128) throw std::runtime_error("launch_panel_sample_rows: one unit row per sample row");
  const int64_t n = (int64_t)rows * (p.ldh / 8);
  hipLaunchKernelGGL(k_panel_image, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Yn_rows, ldn, static_cast<_Float16*>(Ys), p.ldh, 0,
                     rows, rows, D, 0, knn_row_map(rows, nullptr, 1, false));
  HIP_CHECK(hipGetLastError());
}

void launch_panel_sample(const void* Yh, void* Ys, const KnnPanelPlan& p, int32_t N, hipStream_t s) {
  const int32_t m = p.sample_tiles * 128;
  const int64_t n = (int64_t)m * (p.ldh / 8);
  hipLaunchKernelGGL(k_panel_sample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     static_cast<const _Float16*>(Yh), static_cast<_Float16*>(Ys), p.ldh, m, N, p.group_tiles * 128, p.sample_groups,
                     p.scatter != 1 ? 1 : 0, p.map);
  HIP_CHECK(hipGetLastError());
}

void launch_panel_tilemax(const void* Yh, const void* Ys, const KnnPanelPlan& p, int32_t N, int rb_begin, int rb_count,
                          float* tmax, unsigned* queue, int grid, hipStream_t s) {
  if (rb_count <= 0) return;
  PanelArgs a{};
  a.A = static_cast<const _Float16*>(Yh);
  a.B = static_cast<const _Float16*>(Ys);
  a.ldh = p.ldh;
  a.N = N;
  a.rb_begin = rb_begin;
  a.rb_count = rb_count;
  a.ntileB = p.sample_tiles;
  a.S = p.SA;
  a.tiles_per_split = p.sample_tiles_per_split;
  a.group_tiles = p.group_tiles;
  a.ngroups = p.sample_groups;
  a.tmax = tmax;
  a.item_stride = 1;
  a.queue = queue;
  HIP_CHECK(hipMemsetAsync(queue, 0, 4, s));
  if (p.tile_core) {
    launch_tile_thr<0>(a, p.nkt, 2 * grid, s);
    return;
  }
  launch_panel<0>(a, p.nkt, p.nrg_s, false, grid, s);
}

void launch_panel_tau(const float* tmax, const KnnPanelPlan& p, int32_t N, float* tau, hipStream_t s, int32_t r0, int32_t r1) {
  if (r1 < 0) r1 = p.npad;
  if (r0 < 0 || r1 > p.npad || r1 <= r0) return;
  hipLaunchKernelGGL(k_panel_tau, dim3((unsigned)((r1 - r0 + 3) / 4)), dim3(256), 0, s, tmax + (size_t)r0 * p.sample_groups, p.sample_groups,
                     p.sample_rank, r1 - r0, tau + r0);
  (void)N;
  HIP_CHECK(hipGetLastError());
}

void launch_panel_filter(const void* Yh, const KnnPanelPlan& p, int32_t N, int rb_begin, int rb_count, const float* tau,
                         void* hit_list, int32_t* hit_cnt, unsigned* queue, int grid, hipStream_t s, const KnnPanelSymDev* sd,
                         int shard, int shards, int chunk_lo, int chunk_hi) {
  if (rb_count <= 0) return;
  PanelArgs a{};
  a.item_stride = 1;
  const bool window = chunk_hi >= 0;
  if (window) {  // a window of column chunks: the queue walks the chunks from the last to the first
    if (!p.sym || (p.tile_core && !p.tile_wide) || chunk_lo < 0 || chunk_hi > p.S || chunk_lo >= chunk_hi)
      throw std::runtime_error("launch_panel_filter: chunk windows are for the half sweep on the panel core or the wide tile core");
    if (!p.tile_core) {
      auto sets = [&](int c) { return (std::min(p.nrb, (c + 1) * p.T) + p.nrg - 1) / p.nrg; };
      int begin = 0, end = 0;
      for (int c = p.S - 1; c >= chunk_lo; --c) {
        if (c >= chunk_hi) begin += sets(c);
        end += sets(c);
      }
      a.item_begin = begin;
      a.item_end = end;
      grid = std::max(1, std::min(grid, (end - begin + std::max(1, shards) - 1) / std::max(1, shards)));
    }
  }
  if (shards > 1) {
    if (!p.sym || shard < 0 || shard >= shards) throw std::runtime_error("launch_panel_filter: only the half sweep is cut by work items");
    a.item_stride = shards;
    a.item_offset = shard;
  }
  if (p.sym) {
    if (!sd || rb_begin != 0 || rb_count != p.nrb) throw std::runtime_error("launch_panel_filter: the half sweep covers all row blocks");
    a.T = p.T;
    a.nchunks = p.S;
    a.bucket_ent = static_cast<uint2*>(sd->bucket_ent);
    a.bucket_cnt = sd->bucket_cnt;
    a.bucket_cap = p.bucket_cap;
    a.flags = sd->flags;
  }
  a.A = static_cast<const _Float16*>(Yh);
  a.B = a.A;
  a.ldh = p.ldh;
  a.N = N;
  a.rb_begin = rb_begin;
  a.rb_count = rb_count;
  a.ntileB = p.nrb;
  a.S = p.S;
  a.tiles_per_split = p.tiles_per_split;
  a.tau = tau;
  a.hit_list = static_cast<uint2*>(hit_list);
  a.hit_cnt = hit_cnt;
  a.hit_cap = p.hit_cap;
  a.queue = queue;
  HIP_CHECK(hipMemsetAsync(queue, 0, 4, s));
  if (p.tile_core && p.tile_wide) {  // 64 x 128 wave tiles, one workgroup of 8 waves per CU
    // One launch per group of row sets: a group's image rows (sets x 256 rows x ldh halfs) are streamed once per tile
    // pair of every chunk; swept chunk-major over ALL sets (614 MB at config 5) they come from HBM every time, 235 GB per
    // build; a group of <= 128 MB stays in the 256 MB Infinity Cache while its chunks pass.
    const int nsets = (p.nrb + 1) / 2, gs = std::max(1, p.tile_group_sets);
    auto go = [&](auto kern) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T2_LDS_ALL));
      bool first_launch = true;
      for (int g = 0; g * gs < nsets; ++g) {
        a.set0 = g * gs;
        a.nset = std::min(gs, nsets - g * gs);
        int launch_grid = grid;
        if (window) {  // this group's items of the chunks [chunk_lo, chunk_hi) (k_tile_thr2: chunk_sets, chunks from the last)
          auto sets = [&](int c) { return std::max(0, std::min((std::min(p.nrb, (c + 1) * p.T) + 1) / 2 - a.set0, a.nset)); };
          int begin = 0, end = 0;
          for (int c = p.S - 1; c >= chunk_lo; --c) {
            if (c >= chunk_hi) begin += sets(c);
            end += sets(c);
          }
          if (end <= begin) continue;  // (the group's sets lie beyond the rows that have arrived)
          a.item_begin = begin;
          a.item_end = end;
          launch_grid = std::max(1, std::min(grid, end - begin));
        }
        if (!first_launch) HIP_CHECK(hipMemsetAsync(queue, 0, 4, s));
        first_launch = false;
        hipLaunchKernelGGL(kern, dim3(launch_grid), dim3(512), T2_LDS_ALL, s, a, p.nkt);
      }
    };
    go(&k_tile_thr2);
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (p.tile_core) {
    launch_tile_thr<1>(a, p.nkt, 2 * grid, s);  // two workgroups per CU
    return;
  }
  launch_panel<1>(a, p.nkt, p.nrg, p.sym, grid, s);
}

void launch_panel_select(const KnnPanelPlan& p, int rb_begin, int rb_count, int32_t N, const void* hit_list,
                         const int32_t* hit_cnt, float* cval, int32_t* cidx, int32_t* fail_rows, int32_t* fail_count,
                         hipStream_t s, const KnnPanelSymDev* sd) {
  if (rb_count <= 0) return;
  SelectSym sy{};
  if (p.sym) {
    sy.on = 1;
    sy.T = p.T;
    sy.bucket_ent = static_cast<const uint2*>(sd->bucket_ent);
    sy.bucket_cnt = sd->bucket_cnt;
    sy.bucket_cap = p.bucket_cap;
    sy.flags = sd->flags;
  }
  // rows of one wave-of-32 a workgroup sorts: as many as keeps ~5 keep entries per row within 3/4 of its LDS array
  // (per row: the planner's bound on the candidates its threshold lets through, max(5 keep, 20 rho) -- with 5 keep alone a
  // sparse sample (large rho) or a small k overflowed the sort array and sent whole 32-row groups to the exact kernel)
  int nsub = 1;
  const int cap = SORT_CAP;
  while (nsub < 32 && p.hit_bound * (32 / nsub) > 0.75 * SORT_CAP) nsub *= 2;
  const size_t lds_bytes = (size_t)cap * sizeof(uint2);
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel_select), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(k_panel_select, dim3((unsigned)(rb_count * 4 * nsub)), dim3(256), lds_bytes, s,
                     static_cast<const uint2*>(hit_list), hit_cnt, p.hit_cap, p.S, rb_begin, rb_count, nsub, p.keep, N, p.scatter != 1 ? 1 : 0,
                     p.map, cval, cidx, fail_rows, fail_count, sy, (int32_t)cap);
  HIP_CHECK(hipGetLastError());
}

// ---- sharded half sweep: the ranks' partial buckets -> complete buckets of each rank's own rows ---------------------------
// Every rank has swept its share of the work items into buckets of ALL rows.  Rank q selects for the buckets [b0(q), b0(q + 1))
// of its row blocks, so it needs every other rank's entries of those buckets: counts all-gathered, entries packed per
// destination (the bucket ranges are contiguous and in rank order: one prefix sum), exchanged, appended.
__global__ void k_bucket_clamp(const int32_t* cnt, int32_t nb, int32_t cap, int32_t* clamped) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) clamped[b] = min(cnt[b], cap);
}
// one wave per bucket: entries [0, min(cnt, cap)) -> out[off[b] ...]
__global__ __launch_bounds__(256) void k_bucket_pack(const uint2* ent, const int32_t* cnt, const int32_t* off, int32_t nb, int32_t cap,
                                                     uint2* out) {
  const int b = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (b >= nb) return;
  const int n = min(cnt[b], cap);
  for (int e = lane; e < n; e += 64) out[(size_t)off[b] + e] = ent[(size_t)b * cap + e];
}
// one wave per (bucket of my range, source rank): the source's entries of that bucket go behind what the bucket holds
// already -- mine, then the lower ranks' -- up to the bucket's capacity; the last source leaves the summed RAW count (an
// overflow anywhere, or of the sum, stays visible to the select as count > capacity).
// all_cnt [ranks][nb_all] raw counts, src_off [ranks][nb_mine] = offset of (source, bucket) inside that source's segment,
// seg_off [ranks] = start of each source's segment in recv
__global__ __launch_bounds__(256) void k_bucket_merge(uint2* ent, int32_t* cnt, const int32_t* all_cnt, const int32_t* src_off,
                                                      const int64_t* seg_off, const uint2* recv, int32_t b0, int32_t nb_mine,
                                                      int32_t nb_all, int32_t cap, int32_t me, int32_t ranks) {
  const int w = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (w >= nb_mine) return;
  const int b = b0 + w;
  int have = min(all_cnt[(size_t)me * nb_all + b], cap);
  long long raw = all_cnt[(size_t)me * nb_all + b];
  for (int p = 0; p < ranks; ++p) {
    if (p == me) continue;
    const int n = min(all_cnt[(size_t)p * nb_all + b], cap);
    raw += all_cnt[(size_t)p * nb_all + b];
    const uint2* src = recv + seg_off[p] + src_off[(size_t)p * nb_mine + w];
    for (int e = lane; e < n; e += 64)
      if (have + e < cap) ent[(size_t)b * cap + have + e] = src[e];
    have += n;  // (may pass cap: those entries are lost and the raw count says so)
  }
  if (lane == 0) cnt[b] = (int32_t)min(raw, (long long)0x7fffffff);
}
void launch_bucket_clamp(const int32_t* cnt, int32_t nb, int32_t cap, int32_t* clamped, hipStream_t s) {
  if (nb <= 0) return;
  hipLaunchKernelGGL(k_bucket_clamp, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, cnt, nb, cap, clamped);
  HIP_CHECK(hipGetLastError());
}
void launch_bucket_pack(const void* ent, const int32_t* cnt, const int32_t* off, int32_t nb, int32_t cap, void* out, hipStream_t s) {
  if (nb <= 0) return;
  hipLaunchKernelGGL(k_bucket_pack, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, s, static_cast<const uint2*>(ent), cnt, off, nb, cap,
                     static_cast<uint2*>(out));
  HIP_CHECK(hipGetLastError());
}
void launch_bucket_merge(void* ent, int32_t* cnt, const int32_t* all_cnt, const int32_t* src_off, const int64_t* seg_off,
                         const void* recv, int32_t b0, int32_t nb_mine, int32_t nb_all, int32_t cap, int32_t me, int32_t ranks,
                         hipStream_t s) {
  if (nb_mine <= 0) return;
  hipLaunchKernelGGL(k_bucket_merge, dim3((unsigned)((nb_mine + 3) / 4)), dim3(256), 0, s, static_cast<uint2*>(ent), cnt, all_cnt, src_off,
                     seg_off, static_cast<const uint2*>(recv), b0, nb_mine, nb_all, cap, me, ranks);
  HIP_CHECK(hipGetLastError());
}

void launch_bucket_rescore(const KnnPanelPlan& p, const KnnPanelSymDev& sd, const float* Yn, int32_t ldn, int32_t N,
                           const int32_t* rows_in, int32_t nrows, const float* tau, int32_t k, float delta, float* out_val,
                           int32_t* out_idx, int32_t* fail_rows, int32_t* fail_count, hipStream_t s) {
  if (nrows <= 0) return;
  hipLaunchKernelGGL(k_bucket_rescore, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, s, Yn, ldn, N, rows_in, nrows,
                     static_cast<const uint2*>(sd.bucket_ent), sd.bucket_cnt, p.bucket_cap, sd.flags, p.T, tau, p.scatter != 1 ? 1 : 0, p.map, k,
                     delta, out_val, out_idx, fail_rows, fail_count);
  HIP_CHECK(hipGetLastError());
}

}  // namespace osc
