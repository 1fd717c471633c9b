// Lattice build orchestration, chain prior and internal row order of liboscillink_hip.so (see osc_internal.hpp).
#include "osc_internal.hpp"

// ---- graph build --------------------------------------------------------------------------------
void graph_counts(L& h) {
  std::vector<int32_t> d((size_t)h.N);
  HIP_CHECK(hipMemcpyAsync(d.data(), h.deg.p, (size_t)h.N * 4, hipMemcpyDeviceToHost, h.stream));
  sync(h);
  int64_t nnz = 0;
  int32_t mx = 0;
  for (auto v : d) {
    nnz += v;
    mx = std::max(mx, v);
  }
  h.nnz = nnz;
  h.max_deg = mx;
}

void alloc_ell(L& h, int32_t width) {
  h.ell_t_ready = false;
  h.blk_nb = 0;
  ++h.graph_epoch;
  h.width = std::max<int32_t>(1, width);
  const size_t n = (size_t)h.N * h.width;
  h.ell_col.alloc(n);
  h.ell_a.alloc(n);
  h.ell_w.alloc(n);
  h.deg.alloc((size_t)h.N);
  h.sqrt_deg.alloc((size_t)h.N);
  HIP_CHECK(hipMemsetAsync(h.ell_col.p, 0, n * 4, h.stream));
  HIP_CHECK(hipMemsetAsync(h.ell_a.p, 0, n * 4, h.stream));
  HIP_CHECK(hipMemsetAsync(h.ell_w.p, 0, n * 4, h.stream));
  HIP_CHECK(hipMemsetAsync(h.deg.p, 0, (size_t)h.N * 4, h.stream));
}

bool permuted(const L& h) { return !h.perm_h.empty(); }

// path Laplacian structures from the stored chain (graph.py:96-111), in the handle's current row order
void install_chain(L& l) {
  ++l.graph_epoch;
  if (!l.chain_present) return;
  const int32_t len = (int32_t)l.chain_nodes.size();
  auto id = [&](int32_t v) { return permuted(l) ? l.inv_h[(size_t)v] : v; };
  // path adjacency, duplicate edges keep the max weight (graph.py:102-109)
  std::map<std::pair<int32_t, int32_t>, float> adj;
  for (int t = 0; t + 1 < len; ++t) {
    const int32_t i = id(l.chain_nodes[(size_t)t]), j = id(l.chain_nodes[(size_t)t + 1]);
    const float w = l.chain_w.empty() ? 1.0f : l.chain_w[(size_t)t];
    auto put = [&](int32_t r, int32_t c) {
      auto it = adj.find({r, c});
      if (it == adj.end()) adj[{r, c}] = std::max(0.0f, w);
      else it->second = std::max(it->second, w);
    };
    put(i, j);
    put(j, i);
  }
  // normalized_laplacian(A_path) (graph.py:86-93): only rows that own an entry differ from identity
  std::map<int32_t, float> dsum;
  for (auto& kv : adj) dsum[kv.first.first] += kv.second;
  std::map<int32_t, int32_t> slot;
  for (auto& kv : dsum) slot.emplace(kv.first, (int32_t)slot.size());
  std::map<int32_t, int32_t> cnt;
  int32_t pwidth = 1;
  for (auto& kv : adj) pwidth = std::max(pwidth, ++cnt[kv.first.first]);
  const int32_t prows = (int32_t)slot.size();
  std::vector<int32_t> hslot((size_t)l.N, -1), hcol((size_t)prows * pwidth, 0), hdeg((size_t)prows, 0);
  std::vector<float> hw((size_t)prows * pwidth, 0.f);
  auto sd = [&](int32_t r) {
    auto it = dsum.find(r);
    return std::sqrt(std::max(it == dsum.end() ? 0.0f : it->second, 1e-12f));
  };
  std::vector<int32_t> hprow((size_t)prows, 0);
  for (auto& kv : slot) hslot[(size_t)kv.first] = kv.second, hprow[(size_t)kv.second] = kv.first;
  for (auto& kv : adj) {
    const int32_t r = kv.first.first, c = kv.first.second, sl = slot[r];
    const int32_t e = hdeg[(size_t)sl]++;
    hcol[(size_t)sl * pwidth + e] = c;
    hw[(size_t)sl * pwidth + e] = (kv.second * (1.0f / sd(r))) * (1.0f / sd(c));
  }
  l.path_slot.alloc((size_t)l.N);
  l.pcol.alloc(hcol.size());
  l.pw.alloc(hw.size());
  l.pdeg.alloc(hdeg.size());
  l.prow.alloc(hprow.size());
  HIP_CHECK(hipMemcpyAsync(l.prow.p, hprow.data(), hprow.size() * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.path_slot.p, hslot.data(), hslot.size() * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.pcol.p, hcol.data(), hcol.size() * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.pw.p, hw.data(), hw.size() * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.pdeg.p, hdeg.data(), hdeg.size() * 4, hipMemcpyHostToDevice, l.stream));
  sync(l);
  l.prows = prows;
  l.pwidth = pwidth;
}

// move every row-indexed device array between two row orders: new row i takes old row from[i]; ids -> relabel[id]
void move_state(L& l, const int32_t* from_d, const int32_t* relabel_d) {
  const size_t n = (size_t)l.N * l.ld;
  for (DevBuf<float>* b : {&l.Y, &l.U}) {  // AP is scratch between solves
    launch_move_rows(l.AP.p, b->p, from_d, l.N, l.ld, false, l.stream);
    HIP_CHECK(hipMemcpyAsync(b->p, l.AP.p, n * 4, hipMemcpyDeviceToDevice, l.stream));
  }
  DevBuf<float> t1;
  t1.alloc((size_t)l.N);
  for (DevBuf<float>* b : {&l.B, &l.sqrt_deg}) {
    launch_move_f32(t1.p, b->p, from_d, l.N, false, l.stream);
    HIP_CHECK(hipMemcpyAsync(b->p, t1.p, (size_t)l.N * 4, hipMemcpyDeviceToDevice, l.stream));
  }
  const size_t ne = (size_t)l.N * l.width;
  DevBuf<int32_t> col2, deg2;
  DevBuf<float> a2, w2;
  col2.alloc(ne);
  a2.alloc(ne);
  w2.alloc(ne);
  deg2.alloc((size_t)l.N);
  launch_permute_ell(l.ell_col.p, l.ell_a.p, l.ell_w.p, l.deg.p, from_d, relabel_d, l.width, l.N, col2.p, a2.p, w2.p,
                     deg2.p, l.stream);
  sync(l);
  l.ell_col.swap(col2);
  l.ell_a.swap(a2);
  l.ell_w.swap(w2);
  l.deg.swap(deg2);
  l.ell_t_ready = false;
  l.blk_nb = 0;
  l.have_ustar = false;
  l.u_sharded = false;
  ++l.graph_epoch;
}

void drop_order(L& l) {  // back to the API's row order
  if (!permuted(l)) return;
  move_state(l, l.inv_d.p, l.perm_d.p);
  l.perm_h.clear();
  l.inv_h.clear();
  install_chain(l);
}

void apply_order(L& l, const std::vector<int32_t>& perm) {  // perm[new] = old ; state must be in API order
  l.perm_h = perm;
  l.inv_h.assign((size_t)l.N, 0);
  for (int64_t i = 0; i < l.N; ++i) l.inv_h[(size_t)perm[(size_t)i]] = (int32_t)i;
  l.perm_d.alloc((size_t)l.N);
  l.inv_d.alloc((size_t)l.N);
  HIP_CHECK(hipMemcpyAsync(l.perm_d.p, l.perm_h.data(), (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
  HIP_CHECK(hipMemcpyAsync(l.inv_d.p, l.inv_h.data(), (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
  move_state(l, l.perm_d.p, l.inv_d.p);
  install_chain(l);
}

// breadth-first order over the lattice graph (components in order of their smallest node): neighbours end up
// within a narrow band of rows, which is what the XCD-local L2 of the operator apply can hold
std::vector<int32_t> bfs_order(L& l) {
  const size_t ne = (size_t)l.N * l.width;
  std::vector<int32_t> col(ne), deg((size_t)l.N);
  HIP_CHECK(hipMemcpyAsync(col.data(), l.ell_col.p, ne * 4, hipMemcpyDeviceToHost, l.stream));
  HIP_CHECK(hipMemcpyAsync(deg.data(), l.deg.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
  sync(l);
  std::vector<int32_t> order;
  order.reserve((size_t)l.N);
  std::vector<char> seen((size_t)l.N, 0);
  for (int64_t start = 0; start < l.N; ++start) {
    if (seen[(size_t)start]) continue;
    seen[(size_t)start] = 1;
    size_t head = order.size();
    order.push_back((int32_t)start);
    while (head < order.size()) {
      const int32_t u = order[head++];
      const int32_t* cu = col.data() + (size_t)u * l.width;
      for (int e = 0; e < deg[(size_t)u]; ++e) {
        const int32_t v = cu[e];
        if (!seen[(size_t)v]) {
          seen[(size_t)v] = 1;
          order.push_back(v);
        }
      }
    }
  }
  return order;
}

// Re-order the rows when it pays: the BFS + state move cost a few ms at N = 100k and buy ~1.5x on the operator apply
// of a clustered lattice, nothing on an unstructured one.  Auto mode decides on a sampled clustering coefficient.
void maybe_reorder(L& l) {
  l.reordered = false;
  l.clustering = 0.0;
  // under a communicator only the row-sharded CG re-orders (every rank holds the same graph and takes the same
  // deterministic decision and order; the halo lists shrink with locality); the column-sharded default keeps API order
  if (l.reorder == 0 || (l.comm != nullptr && l.shard_mode != 1) || l.N < 2) return;
  if (l.reorder < 0) {
    if (l.N < 8192 || l.nnz == 0) return;  // small lattices run out of LDS / L2 anyway
    constexpr int kSample = 1024;
    DevBuf<unsigned long long> cnt;
    cnt.alloc(2 * kSample);  // (every sampled row writes its own two words: perm_kernels.hip)
    launch_clustering_sample(l.ell_col.p, l.deg.p, l.width, l.N, kSample, cnt.p, l.stream);
    std::vector<unsigned long long> per_row((size_t)2 * kSample, 0ull);
    HIP_CHECK(hipMemcpyAsync(per_row.data(), cnt.p, per_row.size() * 8, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    unsigned long long hc[2] = {0, 0};
    for (int s2 = 0; s2 < kSample; ++s2) hc[0] += per_row[(size_t)2 * s2], hc[1] += per_row[(size_t)2 * s2 + 1];
    l.clustering = hc[1] ? (double)hc[0] / (double)hc[1] : 0.0;
    if (l.clustering < 0.05) return;
  }
  // the order itself: on the device (bfs_order.hip; the same order as the host walk below it, OSC_BFS_HOST=1 forces that)
  bool on_device = false;
  if (!l.bfs_host) {
    DevBuf<int32_t> perm;
    perm.alloc((size_t)l.N);
    if (device_bfs_order(l.ell_col.p, l.deg.p, l.width, (int32_t)l.N, perm.p, l.stream)) {
      std::vector<int32_t> ph((size_t)l.N);
      HIP_CHECK(hipMemcpyAsync(ph.data(), perm.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
      sync(l);
      apply_order(l, ph);
      on_device = true;
    }
  }
  if (!on_device) apply_order(l, bfs_order(l));
  l.reordered = true;
}

// Sharded half sweep: every rank holds partial buckets of ALL rows; rank q needs the other ranks' entries of the buckets of
// ITS row blocks, [4 rb_per q, 4 rb_per (q + 1)).  Raw counts all-gathered (they also carry overflow: a count above the
// capacity stays above it in the sum); each rank packs its buckets behind one another (the ranges are in rank order, so
// one prefix sum gives every destination's segment), grouped send / recv, then the received entries are appended behind
// the rank's own, in rank order.  The chunk overflow flags are combined by max.
void exchange_buckets(L& h, const KnnPanelPlan& pp, const KnnPanelSymDev& sd, int rb_per) {
  const int G = h.world, me = h.rank;
  const int32_t nb_all = pp.npad / 32, nbp = rb_per * 4, stride = nbp * G, cap = pp.bucket_cap;
  auto b0 = [&](int q) { return std::min(nb_all, q * nbp); };
  DevBuf<int32_t> all_cnt, clamped, off, sums, src_off_d;
  DevBuf<int64_t> seg_off_d;
  all_cnt.alloc((size_t)G * stride);
  HIP_CHECK(hipMemsetAsync(all_cnt.p, 0, (size_t)G * stride * 4, h.stream));
  HIP_CHECK(hipMemcpyAsync(all_cnt.p + (size_t)me * stride, sd.bucket_cnt, (size_t)nb_all * 4, hipMemcpyDeviceToDevice, h.stream));
  h.comm->allgather(all_cnt.p, (size_t)stride * 4, h.stream);
  std::vector<int32_t> cnt((size_t)G * stride);
  HIP_CHECK(hipMemcpyAsync(cnt.data(), all_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost, h.stream));
  // my buckets, packed: off[b] = entries before bucket b
  clamped.alloc((size_t)nb_all);
  off.alloc((size_t)nb_all);
  sums.alloc(scan_blocks(nb_all) + 1);
  launch_bucket_clamp(sd.bucket_cnt, nb_all, cap, clamped.p, h.stream);
  exclusive_scan_i32(clamped.p, off.p, nb_all, sums.p, h.stream);
  sync(h);
  auto held = [&](int p, int b) { return (int64_t)std::min(cnt[(size_t)p * stride + b], cap); };
  std::vector<int64_t> seg_start((size_t)G + 1, 0);  // my packed buffer: where each destination's segment starts
  for (int q = 0; q < G; ++q) {
    int64_t n = 0;
    for (int b = b0(q); b < b0(q + 1); ++b) n += held(me, b);
    seg_start[(size_t)q + 1] = seg_start[(size_t)q] + n;
  }
  const int32_t nb_mine = b0(me + 1) - b0(me);
  std::vector<int64_t> seg_off((size_t)G + 1, 0);                    // the receive buffer: one segment per source rank
  std::vector<int32_t> src_off((size_t)G * std::max(1, nb_mine), 0);  // (source, my bucket) -> offset inside that segment
  for (int p = 0; p < G; ++p) {
    int64_t n = 0;
    for (int w = 0; w < nb_mine; ++w) {
      src_off[(size_t)p * nb_mine + w] = (int32_t)n;
      if (p != me) n += held(p, b0(me) + w);
    }
    if (n >= ((int64_t)1 << 31)) throw Unsupported("sharded half sweep: more than 2^31 hits for one rank's rows from one peer");
    seg_off[(size_t)p + 1] = seg_off[(size_t)p] + n;
  }
  DevBuf<unsigned long long> send, recv;
  send.alloc((size_t)std::max<int64_t>(1, seg_start[(size_t)G]));
  recv.alloc((size_t)std::max<int64_t>(1, seg_off[(size_t)G]));
  launch_bucket_pack(sd.bucket_ent, sd.bucket_cnt, off.p, nb_all, cap, send.p, h.stream);
  std::vector<CommXfer> sends, recvs;
  for (int q = 0; q < G; ++q) {
    if (q == me) continue;
    const int64_t ns = seg_start[(size_t)q + 1] - seg_start[(size_t)q], nr = seg_off[(size_t)q + 1] - seg_off[(size_t)q];
    if (ns > 0) sends.push_back(CommXfer{send.p + seg_start[(size_t)q], (size_t)ns * 8, q});
    if (nr > 0) recvs.push_back(CommXfer{recv.p + seg_off[(size_t)q], (size_t)nr * 8, q});
  }
  h.comm->exchange(sends, recvs, h.stream);
  if (nb_mine > 0) {
    src_off_d.alloc(src_off.size());
    seg_off_d.alloc(seg_off.size());
    HIP_CHECK(hipMemcpyAsync(src_off_d.p, src_off.data(), src_off.size() * 4, hipMemcpyHostToDevice, h.stream));
    HIP_CHECK(hipMemcpyAsync(seg_off_d.p, seg_off.data(), seg_off.size() * 8, hipMemcpyHostToDevice, h.stream));
    launch_bucket_merge(sd.bucket_ent, sd.bucket_cnt, all_cnt.p, src_off_d.p, seg_off_d.p, recv.p, b0(me), nb_mine, stride, cap, me, G,
                        h.stream);
  }
  h.comm->allreduce(sd.flags, (size_t)pp.S, COMM_I32, COMM_MAX, h.stream);
  sync(h);  // the host vectors and the temporaries above are in use until here
}

// The streamed create (osc_create -> build_graph(host_Y)).  In the reference's production shape -- one lattice per request,
// cloud/app/main.py:887-947 -- the anchors' way over the bus (6 ms at config 3) used to precede a 14 ms build that needs,
// for most of its work, only part of them: column chunk c of the half sweep reads the image rows below (c + 1) T 128.  So:
//   * the anchors travel in `pieces` of whole column chunks on a second stream (pageable source: the call returns when the
//     piece is on its way; an event per piece);
//   * the column SAMPLE the thresholds come from -- an even stride of lattice rows over the whole array (knn_rowmap.hpp) --
//     is gathered on the host into pinned memory by a few threads while piece 0 travels, and follows it;
//   * behind piece j the build stream runs: U's rows, unit rows, image rows (the piece's own permutation: KnnRowMap), the
//     sample sweep and thresholds of the piece's row blocks, and the main sweep's work items of the piece's column chunks.
// What is left when the last piece has landed is that piece's share of the sweep plus select / re-scoring / graph assembly.
// The lists are those of the whole-array build bit for bit: exact top-k lists under one total order, from the same sample,
// hence the same thresholds, the same hits and the same rows proven (only the order of a bucket's entries differs).
void stream_pieces(L& h, const float* host_Y, const std::vector<int32_t>& starts, const KnnPanelPlan& pp, float* Yn, int32_t ldn, float* p_img,
                   float* p_smp, float* p_tmax, float* p_tau, unsigned* p_queue, const KnnPanelSymDev& sym_dev, int cus, DevBuf<float>& smp_raw,
                   DevBuf<float>& smp_n) {
  const int32_t N = (int32_t)h.N, D = h.D;
  const int pieces = (int)starts.size();
  if (pieces < 1 || pieces > 16) throw std::runtime_error("streamed create: 1 to 16 pieces");  // (p_queue: four counters per piece)
  auto row0 = [&](int j) { return j < pieces ? starts[(size_t)j] : N; };
  const int32_t m_s = pp.sample_tiles * 128, chunk_rows = pp.T * 128;
  const size_t row_bytes = (size_t)D * 4;
  static const int threads = [] {
    const char* e = getenv("OSC_COPY_THREADS");
    const int hw = (int)std::thread::hardware_concurrency();
    return e ? std::max(1, atoi(e)) : std::max(1, std::min(8, hw / 2));
  }();
  // Two build streams take the pieces in turn: a sweep launch is a persistent grid of one workgroup per CU, and on ONE stream
  // piece j + 1's kernels would wait for the last straggler of piece j's sweep.
  hipStream_t up = acquire_stream(h.device), second = acquire_stream(h.device);
  hipStream_t cs[2] = {h.stream, second};
  StagePair sp;
  try {
    sp = acquire_stage(h.device);
  } catch (...) {
    release_stream(h.device, up);
    release_stream(h.device, second);
    throw;
  }
  // events: [j] piece j has landed, [pieces + j] the thresholds of all rows up to piece j's are written, then: everything
  // the caller queued before this call is done / the sample image is written / the second stream has drained
  std::vector<hipEvent_t> ev((size_t)2 * pieces + 4, nullptr);
  hipEvent_t &ev_start = ev[(size_t)2 * pieces], &ev_sample = ev[(size_t)2 * pieces + 1], &ev_done = ev[(size_t)2 * pieces + 2],
             &ev_landed = ev[(size_t)2 * pieces + 3];
  // The host side: a copy from pageable memory returns when the data has left, so the calling thread issues the transfers
  // one by one and queues a piece's kernels behind each; a few threads gather the sample's rows into pinned memory
  // meanwhile (started first: the sample is what the first threshold waits for), and the sample follows the first piece.
  // (Issuing the transfers from a thread of their own closed the 0.08 ms gaps between them and cost 0.4 ms at the start --
  // a new thread's first HIP call -- for a build that is bound by the kernels from the third piece on: not kept.)
  std::vector<std::thread> workers;
  auto cleanup = [&](bool wait) {
    for (auto& t : workers)
      if (t.joinable()) t.join();
    if (wait) {
      (void)hipStreamSynchronize(up);
      (void)hipStreamSynchronize(second);
      (void)hipStreamSynchronize(h.stream);
    }
    for (auto e : ev)
      if (e) (void)hipEventDestroy(e);
    release_stage(h.device, sp);
    release_stream(h.device, up);
    release_stream(h.device, second);
  };
  try {
    // The sample's rows travel through the two pinned staging buffers in FILLS of 32 MB: fill w uses buffer w & 1.  Up to 64 MB
    // that is one gather and two transfers; beyond (round 6: configs 4 and 5 -- 128 / 102 MB of sample rows) the buffers are
    // reused: a buffer is gathered into again once the transfer of its previous fill has left it, and a piece of anchors
    // travels (and its row kernels run) during every such gather, so neither the bus nor the device waits for the host threads.
    const int64_t rows_per_buf = std::max<int64_t>(1, (int64_t)(kStageBytes / row_bytes));
    const int nfill = (int)((m_s + rows_per_buf - 1) / rows_per_buf);
    const int nthr = nfill > 2 ? std::max(1, std::min(16, std::max(threads, (int)std::thread::hardware_concurrency() / 8)))
                               : std::max(1, std::min(threads, 8));
    // The sample is defined in lattice terms (knn_rowmap.hpp: an even stride of lattice rows, dealt to the threshold groups
    // in turn), so it can be put together from the caller's array before a single image row exists -- and it is the very
    // sample the whole-array build copies out of its image: same thresholds, same hits, same rows proven.  (Its first form
    // here took every rho-th IMAGE row, as the build did until round 5: with pieces permuted separately a cluster's sampled
    // mates then sat in the few groups of their own piece, the thresholds of anchors that arrive cluster by cluster fell
    // to the background level and every row went to the exact kernel -- soak_streamed_create.py, 60 000 x 768, k = 8,
    // clusters of 300; two permutations of the sample order later the count of distinct groups was still left to chance.)
    const int32_t gsz = pp.group_tiles * 128, G = pp.sample_groups;
    auto fill_begin = [&](int w) { return (int32_t)std::min<int64_t>(m_s, (int64_t)w * rows_per_buf); };
    // host threads gather the sample rows of fills [w0, w1) (at most two: one per buffer) into the staging buffers
    auto gather_async = [&](int w0, int w1) {
      const int32_t a = fill_begin(w0), b = fill_begin(w1);
      float* const pin[2] = {static_cast<float*>(sp.buf[0]), static_cast<float*>(sp.buf[1])};
      for (int t = 0; t < nthr; ++t)
        workers.emplace_back([=] {
          for (int32_t r = a + (int32_t)((int64_t)(b - a) * t / nthr); r < a + (int32_t)((int64_t)(b - a) * (t + 1) / nthr); ++r) {
            const int32_t row = knn_sample_lattice_row(knn_sample_index(r, m_s, gsz, G), m_s, N);
            const int w = (int)(r / rows_per_buf);
            std::memcpy(pin[w & 1] + (size_t)(r - (int64_t)w * rows_per_buf) * D, host_Y + (size_t)row * D, row_bytes);
          }
        });
    };
    auto join_workers = [&] {
      for (auto& t : workers) t.join();
      workers.clear();
    };
    gather_async(0, std::min(2, nfill));
    // (events are made when first used: twenty-odd creations are 0.1 ms the first transfer need not wait for)
    auto E = [&](hipEvent_t& e) -> hipEvent_t {
      if (e == nullptr) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      return e;
    };
    HIP_CHECK(hipEventRecord(E(ev_start), h.stream));
    HIP_CHECK(hipStreamWaitEvent(second, E(ev_start), 0));
    auto send_piece = [&](int j) {
      const int32_t r0 = row0(j), r1 = row0(j + 1);
      HIP_CHECK(hipMemcpyAsync(h.Y.p + (size_t)r0 * h.ld, host_Y + (size_t)r0 * D, (size_t)(r1 - r0) * row_bytes, hipMemcpyHostToDevice, up));
      HIP_CHECK(hipEventRecord(E(ev[(size_t)j]), up));
    };
    // behind piece j's arrival: U's rows, unit rows, image rows
    auto queue_rows = [&](int j) {
      hipStream_t s = cs[j & 1];
      const int32_t r0 = row0(j), r1 = row0(j + 1);
      HIP_CHECK(hipStreamWaitEvent(s, E(ev[(size_t)j]), 0));
      HIP_CHECK(hipMemcpyAsync(h.U.p + (size_t)r0 * h.ld, h.Y.p + (size_t)r0 * h.ld, (size_t)(r1 - r0) * h.ld * 4, hipMemcpyDeviceToDevice, s));
      launch_normalize_rows(h.Y.p + (size_t)r0 * h.ld, h.ld, Yn + (size_t)r0 * ldn, ldn, r1 - r0, D, s);
      launch_panel_image(Yn, ldn, p_img, pp, N, D, s, r0, j + 1 == pieces ? pp.npad : r1);
    };
    // ... and, once the sample image exists: the thresholds of the piece's row blocks and the sweep of its column chunks
    auto queue_sweep = [&](int j) {
      hipStream_t s = cs[j & 1];
      unsigned* q = p_queue + 4 * j;
      const int32_t r0 = row0(j), r1 = row0(j + 1);
      const bool last = j + 1 == pieces;
      if (j & 1) HIP_CHECK(hipStreamWaitEvent(s, E(ev_sample), 0));  // (the sample image was written on the first stream)
      const int rb0 = r0 / 128, rb1 = last ? pp.nrb : r1 / 128;
      KnnPanelPlan pj = pp;  // (splits of the sample sweep chosen for THIS many row blocks)
      const int nsets = (rb1 - rb0 + pp.nrg_s - 1) / pp.nrg_s;
      double best = 1e30;
      for (int S = 1; S <= 6 && S <= pp.sample_groups; ++S) {
        const double rounds = (double)nsets * S / std::max(1, cus);
        const double cost = std::ceil(rounds) / rounds * (1.0 + 0.03 * S);
        if (cost < best - 1e-9) {
          best = cost;
          pj.SA = S;
        }
      }
      pj.sample_tiles_per_split = ((pp.sample_groups + pj.SA - 1) / pj.SA) * pp.group_tiles;
      launch_panel_tilemax(p_img, p_smp, pj, N, rb0, rb1 - rb0, p_tmax, q, std::max(1, std::min(cus, nsets * pj.SA)), s);
      launch_panel_tau(p_tmax, pp, N, p_tau, s, rb0 * 128, rb1 * 128);
      // (the sweep reads the image rows and thresholds of ALL pieces up to this one: the other stream wrote piece j - 1's)
      if (j > 0) HIP_CHECK(hipStreamWaitEvent(s, E(ev[(size_t)pieces + j - 1]), 0));
      HIP_CHECK(hipEventRecord(E(ev[(size_t)pieces + j]), s));
      const int c0 = r0 / chunk_rows, c1 = last ? pp.S : r1 / chunk_rows;
      if (c1 > c0)
        launch_panel_filter(p_img, pp, N, 0, pp.nrb, p_tau, sym_dev.bucket_ent, sym_dev.bucket_cnt, q + 1, cus, s, &sym_dev, 0, 1, c0, c1);
    };
    const int lead = 1;  // pieces that travel while the sample's first fills are being gathered (25 MB in ~0.5 ms: one piece's time on the bus)
    int next_piece = 0;
    for (; next_piece < lead; ++next_piece) {
      send_piece(next_piece);
      queue_rows(next_piece);
    }
    smp_raw.alloc((size_t)m_s * D);
    smp_n.alloc((size_t)m_s * ldn);
    // fill w: staging buffer -> its rows of the gathered sample; the buffer is free again when sp.ev[w & 1] has fired
    auto ship = [&](int w) {
      const int32_t a = fill_begin(w), b = fill_begin(w + 1);
      HIP_CHECK(hipMemcpyAsync(smp_raw.p + (size_t)a * D, sp.buf[w & 1], (size_t)(b - a) * row_bytes, hipMemcpyHostToDevice, up));
      HIP_CHECK(hipEventRecord(sp.ev[w & 1], up));
    };
    join_workers();
    for (int w = 0; w < std::min(2, nfill); ++w) ship(w);
    for (int w = 2; w < nfill; ++w) {
      HIP_CHECK(hipEventSynchronize(sp.ev[w & 1]));  // (fill w - 2 has left the buffer)
      gather_async(w, w + 1);
      if (next_piece < pieces) {  // the bus and the row kernels work while the host threads gather
        send_piece(next_piece);
        queue_rows(next_piece);
        ++next_piece;
      }
      join_workers();
      ship(w);
    }
    HIP_CHECK(hipEventRecord(E(ev_landed), up));
    HIP_CHECK(hipStreamWaitEvent(h.stream, E(ev_landed), 0));  // the sample image: unit rows of the gathered anchors, in sample order
    launch_normalize_rows(smp_raw.p, D, smp_n.p, ldn, m_s, D, h.stream);
    launch_panel_sample_rows(smp_n.p, ldn, p_smp, pp, m_s, D, h.stream);
    HIP_CHECK(hipEventRecord(E(ev_sample), h.stream));
    for (int j = 0; j < next_piece; ++j) queue_sweep(j);
    for (int j = next_piece; j < pieces; ++j) {
      send_piece(j);
      queue_rows(j);
      queue_sweep(j);
    }
    HIP_CHECK(hipEventRecord(E(ev_done), second));
    HIP_CHECK(hipStreamWaitEvent(h.stream, E(ev_done), 0));
    HIP_CHECK(hipStreamSynchronize(up));  // (the caller's array is free again; the pinned buffer and the events go back)
    HIP_CHECK(hipStreamSynchronize(second));
  } catch (...) {
    cleanup(true);
    throw;
  }
  cleanup(false);
  h.create_pieces = pieces;
}

// host_Y (osc_create only): the caller's anchors, not on the device yet -- the build brings them there, either whole before
// anything else or, where the half sweep on the panel core builds the lists (one process, D <= 768), piece by piece on a
// second stream while the kernels work on the pieces that have arrived (stream_pieces below).
static bool build_graph_once(L& h, const float* host_Y);
void build_graph(L& h, const float* host_Y) {
  const double t0 = now_ms();
  if (!build_graph_once(h, host_Y)) {
    // A streamed build that lost more than a few rows to overflowing lists (anchors grouped in runs longer than a piece's
    // scatter can spread: the rows' chunks are flagged and would go to the all-fp32 kernel, seconds at config 3's size):
    // the anchors are resident now, so the whole-array build -- whose scatter spreads a group over the whole image -- runs
    // instead, for one more prefilter pass (14 ms at config 3).
    const int32_t pieces = h.create_pieces;
    if (!build_graph_once(h, nullptr)) throw std::runtime_error("build_graph: the whole-array build asked for a retry");
    h.create_pieces = -pieces;
  }
  h.build_ms = now_ms() - t0;
}

// false: a streamed build gave up before its exact-kernel fallback (see build_graph); Y and U are on the device then
static bool build_graph_once(L& h, const float* host_Y) {
  const double t0 = now_ms();
  h.create_pieces = 0;
  auto upload_all = [&] {
    if (host_Y == nullptr) return;
    upload_rows(h, h.Y.p, host_Y);
    HIP_CHECK(hipMemcpyAsync(h.U.p, h.Y.p, (size_t)h.N * h.ld * 4, hipMemcpyDeviceToDevice, h.stream));
    host_Y = nullptr;
  };
  drop_order(h);  // the build works on the API's row order
  const int32_t N = (int32_t)h.N;
  h.k_eff = std::min<int32_t>(h.k_eff, std::max<int32_t>(1, N - 1));  // lattice.py:60
  h.have_ustar = false;
  if (N <= 1) {  // graph.py:30-32
    upload_all();
    alloc_ell(h, 1);
    const float one_em6 = 1e-6f;  // sqrt(max(0, 1e-12))
    std::vector<float> sd((size_t)h.N, one_em6);
    HIP_CHECK(hipMemcpyAsync(h.sqrt_deg.p, sd.data(), sd.size() * 4, hipMemcpyHostToDevice, h.stream));
    sync(h);
    h.knn_k = 0;
    h.have_graph = true;
    h.nnz = 0;
    h.max_deg = 0;
    h.build_ms = now_ms() - t0;
    return true;
  }
  const int32_t k = h.k_eff;
  // k <= 128: register-resident streaming lists (exact / prefilter / small-dense routes below).  Larger k (the
  // reference takes any k <= N - 1, lattice.py:60): dense similarity rows in chunks + a radix select per row.
  const bool any_k = k > 128;
  const int32_t ldn = ((h.D + 31) / 32) * 32;
  DevBuf<float> Yn;
  Yn.alloc((size_t)h.N * ldn);
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, h.device));
  const int slots = prop.multiProcessorCount * (k <= 64 ? 2 : 1);
  // multi-GPU: row-block-sharded build -- this rank computes the top-k lists of its 128-row blocks against all
  // columns, then one all-gather of the (idx, sim) lists; mutual test / cap / Laplacian weights run on every rank.
  const int all_rb = (N + 127) / 128;
  // OSC_KNN_FAKE_SHARDS=G (test hook): run the G per-rank passes of a sharded build one after another on this GPU
  const int fake = h.knn_fake_shards;
  const bool sharded = h.comm != nullptr && h.world > 1;
  const int parts = sharded ? h.world : (fake > 1 ? fake : 1);
  const int rb_per = (all_rb + parts - 1) / parts;
  const size_t list_rows = parts > 1 ? (size_t)rb_per * 128 * parts : (size_t)h.N;
  h.knn_val.alloc(list_rows * k);
  h.knn_idx.alloc(list_rows * k);
  h.knn_k = k;
  HIP_CHECK(hipMemsetAsync(h.knn_val.p, 0, list_rows * k * 4, h.stream));
  HIP_CHECK(hipMemsetAsync(h.knn_idx.p, 0xFF, list_rows * k * 4, h.stream));
  // Two ways to the per-row top-k lists (identical results):
  //  exact     : fp32 MFMA similarity tiles + running top-k.
  //  prefilter : fp16 MFMA tiles keep the best KC >= k+16 candidates per row, exact fp32 re-scoring picks the k;
  //              a row is accepted only if the worst-case fp16 error bound proves no left-out column can belong
  //              to its top-k, otherwise the row is redone by the exact kernel.
  // kept candidates per row: k plus a margin; rows whose margin turns out too thin are redone exactly
  const int keep_f = std::min(96, k + std::max(12, k / 2));
  constexpr bool dense_small = true;
  constexpr int dense_max = 8192;
  // small lattices go through the dense similarity matrix (below) unless the panel route takes them; beyond that the fp16
  // prefilter pays
  bool prefilter = (keep_f >= k + 8) && N >= 4096;
  // OSC_KNN_MODE = exact | prefilter | panel: force one route (tests, A/B)
  if (h.knn_mode == 1) prefilter = false;
  if (h.knn_mode == 2) prefilter = (keep_f >= k + 8);
  if (any_k) prefilter = false;
  // The prefilter's GEMM has two shapes: "panel" (knn_gemm.hip: query panel in registers, thresholds from a column
  // sample, hits appended -- D <= 768 and enough row blocks for the sample) and the older 128 x 128 tile with
  // register-resident sorted lists (k_knn_pref), which serves everything else.
  // Where the panel route starts.  Until round 6: behind the dense route, at 8193 rows.  With that round's lighter small kernels
  // (threshold kernel, select, re-scoring tail, row kernels) it is ahead of the dense route from its own lower limit of 6144
  // rows on wide rows (scripts/exp/r06/route_crossover.py, build in ms, dense / panel: 6144 x 768 k 32 0.83 / 0.63, 8192 x 768
  // 1.45 / 0.77, 6144 x 1536 1.33 / 0.73, 8192 x 1536 2.34 / 0.83; 7000 x 384 k 16 0.61 / 0.55, 8192 x 384 0.82 / 0.69, but 6144 x
  // 384 0.48 / 0.55 and 8192 x 128 0.53 / 0.62): from 6144 rows at >= 512 columns, 7168 at >= 320, 8193 below.  (Sharded builds
  // have no dense route and keep their tile prefilter up to 8192 rows.)
  const int panel_min = parts == 1 && h.knn_mode == 0 ? (h.D >= 512 ? 6144 : h.D >= 320 ? 7168 : 8193) : 8193;
  // (a hit entry packs the column index into 25 bits, next to its two side flags)
  // (D > 768: the same route on the tile core, k_tile_thr -- half sweep only, so single-process builds only)
  // (round 5: the half sweep also under sharding -- the ranks split the work ITEMS of the one sweep and exchange the hits of
  // each other's rows, below -- so a sharded build issues the single-GPU build's MFMA work, not twice it, and D > 768 keeps
  // the threshold route instead of falling back to the list-maintaining tile prefilter)
  bool sym_ok = h.knn_sym;
  if (sym_ok && prefilter && N >= panel_min && N < (1 << 25)) {
    // The half sweep delivers every hit to a bucket per 32 receiving rows: (npad / 32) x bucket_cap entries of 8 bytes --
    // 2.9 GB of temporaries at N = 1M (config 4), growing with N x the threshold sample's hit bound (the full sweep's
    // lists: 0.2-0.5 GB).  Beyond a budget, or where the device cannot spare it, the build takes the full sweep (D <= 768)
    // or the tile prefilter (D > 768) instead of failing in the allocator.
    const KnnPanelPlan sp = knn_panel_plan(N, h.D, keep_f, prop.multiProcessorCount, false, true, h.knn_tune);
    const double bucket_bytes = (double)(sp.npad / 32) * (double)sp.bucket_cap * 8.0;
    size_t mem_free = 0, mem_total = 0;
    HIP_CHECK(hipMemGetInfo(&mem_free, &mem_total));
    constexpr double kSymBucketBudget = 12.0 * 1024 * 1024 * 1024;
    if (sp.ok && (bucket_bytes > kSymBucketBudget || bucket_bytes > 0.5 * (double)mem_free)) sym_ok = false;
  }
  const bool depth_ok = knn_panel_nkt(h.D) != 0 || (sym_ok && knn_tile_nkt(h.D) != 0);
  bool panel = prefilter && depth_ok && N >= panel_min && N < (1 << 25) &&
               knn_panel_plan(N, h.D, keep_f, prop.multiProcessorCount, false, sym_ok, h.knn_tune).ok;
  if (h.knn_mode == 0 && dense_small && parts == 1 && N <= dense_max && !panel) prefilter = false;  // the dense route (below)
  if (h.knn_mode == 3) panel = prefilter = (keep_f >= k + 8) && !any_k && depth_ok && N >= 6144 && N < (1 << 25);
  if (h.knn_mode == 2) panel = false;
  h.knn_panel = panel;
  h.knn_sweep = 0;
  DevBuf<float> cand_val, cval, pair_sc;  // (pair_sc / pair_pos: the re-scoring's scratch where it scores every undirected pair once)
  DevBuf<int32_t> pair_pos;
  DevBuf<int32_t> cand_idx, cidx, fail_rows, fail_count;
  DevBuf<float> Yh;  // fp16 image, viewed as float slots
  const int32_t ldh = ((h.D + 63) / 64) * 64;
  h.knn_fallback_rows = 0;
  h.knn_prefilter = prefilter;
  KnnPanelPlan pp{};
  DevBuf<float> p_img, p_smp, p_tmax, p_tau;
  DevBuf<unsigned long long> p_hits;
  DevBuf<int32_t> p_hcnt;
  DevBuf<unsigned> p_queue;
  KnnPanelSymDev sym_dev{};
  std::vector<int32_t> piece_starts;  // not empty: the streamed create (below); first image row of each piece
  DevBuf<float> smp_raw, smp_n;  // ... its column sample: anchors as gathered, unit rows
  if (panel) {
    // (image rows scattered over the lattice rows in single-process builds: knn_gemm.hpp, KnnPanelPlan::scatter)
    // single-process builds sweep only the column tiles J >= I of every row block (knn_gemm.hip: symmetric half sweep);
    // a sharded build's ranks own row blocks and would have to exchange the column-side hits, so they keep the full sweep
    // (OSC_KNN_PANEL_SCATTER=0 / OSC_KNN_PANEL_SYM=0: A/B and tests)
    // (round 5: the row scatter also under a SHARED half sweep -- a rank then owns image row blocks, i.e. lattice rows spread
    // over the whole lattice, and the ranks' lists are combined by sums instead of an all-gather, below)
    pp = knn_panel_plan(N, h.D, keep_f, prop.multiProcessorCount, h.knn_scatter && (parts == 1 || sym_ok), sym_ok, h.knn_tune);
    h.knn_sweep = pp.sym ? 2 : 1;
    p_img.alloc((size_t)(pp.npad + 128) * pp.ldh / 2);  // (+ one zero tile: k_tile_thr2 sweeps row blocks and column tiles in pairs)
    HIP_CHECK(hipMemsetAsync(p_img.p + (size_t)pp.npad * pp.ldh / 2, 0, (size_t)128 * pp.ldh * 2, h.stream));
    p_smp.alloc((size_t)pp.sample_tiles * 128 * pp.ldh / 2);
    p_tmax.alloc((size_t)pp.npad * pp.sample_groups);
    p_tau.alloc(std::max((size_t)pp.npad, (size_t)rb_per * 128 * parts));  // (whole equal chunks for the all-gather of a sharded half sweep)
    p_queue.alloc(64);  // (one counter per launch in flight: the streamed create runs consecutive pieces on two streams)
    // Anchors still on the host (osc_create) and the half sweep on the panel core ahead: cut the image into equal pieces of
    // whole column chunks, >= 24 MB of anchors each and at most 16 of them, each permuted within itself
    // (knn_panel_set_pieces) -- the sweep's chunk c needs the image rows below (c + 1) T 128 and nothing else, i.e. the
    // pieces up to its own.  (Schedules tried at config 3 / 200k x 384 clustered, create in ms: 11 equal pieces 16.5 / 28.6;
    // 1, 1, 2, 2, 3, 4, 6, 8 chunks 16.7 / 30.3 -- the kernels run dry while the last large pieces travel --; 1, 1, 2, 2 then
    // threes 16.7 / 32.3 -- sixteen short sample sweeps fill the CUs badly.  From the third piece on the kernels are the
    // slower side, so nothing is gained by small first pieces either.)
    if (host_Y != nullptr && h.create_stream && parts == 1 && h.comm == nullptr && pp.sym && (!pp.tile_core || pp.tile_wide) && h.ld == h.D) {
      const int64_t row_bytes = (int64_t)h.D * 4, chunk_rows = (int64_t)pp.T * 128;
      const int64_t smp_bytes = (int64_t)pp.sample_tiles * 128 * row_bytes;
      int64_t m = std::max<int64_t>(1, (((int64_t)h.create_piece_mb << 20) / row_bytes + chunk_rows - 1) / chunk_rows);
      m = std::max<int64_t>(m, (pp.S + 15) / 16);
      // A piece's rows are permuted among themselves only, so the hits of anchors that arrive group by group -- up to the
      // threshold's bound per row, all of them inside the row's own piece -- spread over the piece's column tiles and no
      // further; a wave's hit list takes 260 / NRG coarse entries from ONE tile (knn_gemm.hip: HB_CAP_SYM), i.e. 32 rows x
      // bound / tiles must stay below that: pieces of >= 32 x bound rows leave a factor of two.  (An explicit
      // OSC_CREATE_PIECE_MB overrides this: tests of the retry below.)
      // (the wide tile core's lists are shorter and its wave tiles taller -- 64 rows, ~96 entries from one tile: 176 x bound rows;
      // soak_streamed_create.py, 82 785 x 800, k = 16, grouped: pieces of 8192 rows overflowed and the build handed over)
      // (two row groups at K depth 12 append per 64-column PASS: half a tile of entries at a time)
      const double min_rows = pp.tile_core ? 176.0 * pp.hit_bound : 32.0 * pp.hit_bound * (pp.nkt == 12 && pp.nrg == 2 ? 1 : pp.nrg);
      if (!h.create_piece_mb_set) m = std::max<int64_t>(m, ((int64_t)min_rows + chunk_rows - 1) / chunk_rows);
      // (what is left over joins the last piece: a piece's rows are permuted among themselves only, so a short piece of
      // anchors that arrive cluster by cluster packs each cluster into few tiles -- 3072 rows holding 7.7 clusters of 401 gave
      // every row 17 cluster mates per column tile, more than a wave's hit list takes from one tile)
      const int64_t rows = m * chunk_rows, pieces = N / rows;
      // (two pieces from 256 MB of anchors on -- in practice the wide tile core's 176 x bound rows: the second half travels behind
      // the first half's kernels.  Create, two pieces vs whole array (scripts/exp/r06/create_n.py, profiles/r06_two_piece.txt):
      // config 5's 1.2 GB 98.9 vs 111.7 ms, 150 000 x 1152 k 56 48.0 vs 54.7, 132 000 x 800 k 48 30.8 vs 34.6; on the panel core
      // 48 000 x 768 k 64 (147 MB) is a tie, 9.1-9.6 vs 9.4: three pieces stay the rule below)
      const int64_t min_pieces = (int64_t)N * row_bytes >= ((int64_t)h.create_two_mb << 20) ? 2 : 3;
      if (pieces >= min_pieces && (int64_t)N * row_bytes >= ((int64_t)h.create_min_mb << 20) && smp_bytes <= 8 * (int64_t)kStageBytes) {  // (up to eight fills of the two staging buffers: stream_pieces)
        for (int64_t j = 0; j < pieces; ++j) piece_starts.push_back((int32_t)(j * rows));
        knn_panel_set_pieces(pp, N, piece_starts.data(), (int)piece_starts.size());
      }
    }
  }
  const bool streamed = !piece_starts.empty();
  if (!streamed) {
    upload_all();
    launch_normalize_rows(h.Y.p, h.ld, Yn.p, ldn, h.N, h.D, h.stream);
    if (panel) {
      launch_panel_image(Yn.p, ldn, p_img.p, pp, N, h.D, h.stream);
      launch_panel_sample(p_img.p, p_smp.p, pp, N, h.stream);
    }
  }
  if (prefilter) {
    if (!panel) {
      Yh.alloc((size_t)h.N * ldh / 2);
      launch_to_f16(Yn.p, ldn, Yh.p, ldh, h.N, h.D, h.stream);
    }
    cval.alloc((size_t)h.N * keep_f);
    cidx.alloc((size_t)h.N * keep_f);
    fail_rows.alloc((size_t)h.N);
    fail_count.alloc(1);
    HIP_CHECK(hipMemsetAsync(fail_count.p, 0, 4, h.stream));
  }
  // worst-case |fp16-path score - exact score| for unit rows: (2u + u^2) with u = 2^-11, plus fp32 accumulation
  const float delta = 9.8e-4f + 1.2e-7f * (float)h.D;
  // (OSC_KNN_FORCE_EXCHANGE: the same flow with its collectives under a one-rank communicator -- the only form in which the
  // RCCL backend's all-gather / grouped send-recv / int32 max all-reduce of this path can run on a one-GPU box)
  const bool exchange = sharded || (h.knn_force_exchange && h.comm != nullptr);
  const bool sym_sharded = panel && pp.sym && (parts > 1 || exchange);
  if (sym_sharded) {
    // Half sweep of a sharded build (graph.py:35-65 cut over the ranks): thresholds of a rank's own row blocks, all-gathered;
    // then ONE sweep of the tiles J >= I whose work items the ranks take in turn (item = rank, rank + parts, ...: items of a
    // chunk stay neighbours), every rank delivering into buckets of ALL rows; then the entries of each rank's own rows travel
    // to it (exchange_buckets).  OSC_KNN_FAKE_SHARDS runs the ranks' passes one after another into the same buckets.
    ProfScope ps(h, 3);
    for (int part = 0; part < parts; ++part) {
      if (sharded && part != h.rank) continue;
      const int rb_begin = std::min(all_rb, part * rb_per), rb_count = std::max(0, std::min(rb_per, all_rb - rb_begin));
      const int nsets = (rb_count + pp.nrg_s - 1) / pp.nrg_s;  // (work items per column split of the SAMPLE sweep)
      launch_panel_tilemax(p_img.p, p_smp.p, pp, N, rb_begin, rb_count, p_tmax.p, p_queue.p,
                           std::max(1, std::min(prop.multiProcessorCount, nsets * pp.SA)), h.stream);
    }
    launch_panel_tau(p_tmax.p, pp, N, p_tau.p, h.stream);  // (rows of other ranks' blocks: overwritten by the all-gather)
    if (exchange) h.comm->allgather(p_tau.p, (size_t)rb_per * 128 * 4, h.stream);
    const size_t nb = (size_t)pp.npad / 32;
    p_hits.alloc(nb * pp.bucket_cap);
    p_hcnt.alloc(nb + (size_t)pp.S);
    HIP_CHECK(hipMemsetAsync(p_hcnt.p, 0, (nb + (size_t)pp.S) * 4, h.stream));
    sym_dev.bucket_ent = p_hits.p;
    sym_dev.bucket_cnt = p_hcnt.p;
    sym_dev.flags = p_hcnt.p + nb;
    const int sgrid = std::max(1, std::min(prop.multiProcessorCount, (pp.nitems + parts - 1) / parts));
    for (int part = 0; part < parts; ++part) {
      if (sharded && part != h.rank) continue;
      launch_panel_filter(p_img.p, pp, N, 0, pp.nrb, p_tau.p, p_hits.p, p_hcnt.p, p_queue.p, sgrid, h.stream, &sym_dev, part, parts);
    }
    if (exchange) exchange_buckets(h, pp, sym_dev, rb_per);
  }
  for (int part = 0; part < parts; ++part) {
    if (sharded && part != h.rank) continue;
    const int rb_begin = std::min(all_rb, part * rb_per);
    const int rb_count = std::max(0, std::min(rb_per, all_rb - rb_begin));
    if (any_k) {
      // chunks of up to ~1 GiB of similarity rows (multiple of 128 rows)
      const int32_t ldS = ((N + 31) / 32) * 32;
      const int64_t cap_rows = std::max<int64_t>(128, (((int64_t)1 << 28) / ldS) / 128 * 128);
      const int32_t row_lo = rb_begin * 128, row_hi = std::min(N, (rb_begin + rb_count) * 128);
      const int32_t chunk = (int32_t)std::min<int64_t>(cap_rows, ((row_hi - row_lo + 127) / 128) * 128);
      if (row_hi > row_lo) {
        DevBuf<float> Sm;
        Sm.alloc((size_t)chunk * ldS);
        ProfScope ps(h, 3);
        for (int32_t r = row_lo; r < row_hi; r += chunk)
          launch_knn_rows_any(Yn.p, ldn, N, k, r, std::min(chunk, row_hi - r), Sm.p, ldS, h.knn_val.p, h.knn_idx.p, h.stream);
        sync(h);  // Sm goes back to the pool at scope exit
      }
    } else if (panel) {
      const KnnPlan plan = knn_plan(N, keep_f, slots, rb_begin, rb_count, true, h.knn_splits);  // row range + keep for the re-scoring
      const int nsets = (rb_count + pp.nrg - 1) / pp.nrg;  // work items per column split (knn_gemm.hip)
      const int grid = std::max(1, std::min(prop.multiProcessorCount, nsets * pp.S));
      if (streamed) {
        ProfScope ps(h, 3);
        const size_t nb = (size_t)pp.npad / 32;
        p_hits.alloc(nb * pp.bucket_cap);
        p_hcnt.alloc(nb + (size_t)pp.S);
        HIP_CHECK(hipMemsetAsync(p_hcnt.p, 0, (nb + (size_t)pp.S) * 4, h.stream));
        sym_dev.bucket_ent = p_hits.p;
        sym_dev.bucket_cnt = p_hcnt.p;
        sym_dev.flags = p_hcnt.p + nb;
        stream_pieces(h, host_Y, piece_starts, pp, Yn.p, ldn, p_img.p, p_smp.p, p_tmax.p, p_tau.p, p_queue.p, sym_dev, prop.multiProcessorCount,
                      smp_raw, smp_n);
        host_Y = nullptr;
      } else if (!sym_sharded) {  // (a sharded half sweep has its thresholds and buckets already: above)
        ProfScope ps(h, 3);
        launch_panel_tilemax(p_img.p, p_smp.p, pp, N, rb_begin, rb_count, p_tmax.p, p_queue.p,
                             std::max(1, std::min(prop.multiProcessorCount, (rb_count + pp.nrg_s - 1) / pp.nrg_s * pp.SA)), h.stream);
        launch_panel_tau(p_tmax.p, pp, N, p_tau.p, h.stream);
        if (pp.sym) {
          const size_t nb = (size_t)pp.npad / 32;
          p_hits.alloc(nb * pp.bucket_cap);  // one bucket per group of 32 receiving rows
          p_hcnt.alloc(nb + (size_t)pp.S);   // [bucket counts | chunk flags]
          HIP_CHECK(hipMemsetAsync(p_hcnt.p, 0, (nb + (size_t)pp.S) * 4, h.stream));
          sym_dev.bucket_ent = p_hits.p;
          sym_dev.bucket_cnt = p_hcnt.p;
          sym_dev.flags = p_hcnt.p + nb;
          const int sgrid = std::max(1, std::min(prop.multiProcessorCount, pp.nitems));
          launch_panel_filter(p_img.p, pp, N, rb_begin, rb_count, p_tau.p, p_hits.p, p_hcnt.p, p_queue.p, sgrid, h.stream, &sym_dev);
        } else {
          p_hits.alloc((size_t)rb_count * pp.S * 4 * pp.hit_cap);  // one list per (work item, wave)
          p_hcnt.alloc((size_t)rb_count * pp.S * 4);
          launch_panel_filter(p_img.p, pp, N, rb_begin, rb_count, p_tau.p, p_hits.p, p_hcnt.p, p_queue.p, grid, h.stream);
        }
      }
      static const bool knn_debug = getenv("OSC_KNN_DEBUG") != nullptr;  // diagnostic: where the prefilter loses rows
      auto failed_so_far = [&] {
        int32_t n = 0;
        HIP_CHECK(hipMemcpyAsync(&n, fail_count.p, 4, hipMemcpyDeviceToHost, h.stream));
        sync(h);
        return n;
      };
      if (knn_debug && pp.sym) {
        const size_t nb = (size_t)pp.npad / 32;
        std::vector<int32_t> cnt(nb + (size_t)pp.S);
        std::vector<float> tau((size_t)pp.npad);
        HIP_CHECK(hipMemcpyAsync(cnt.data(), p_hcnt.p, cnt.size() * 4, hipMemcpyDeviceToHost, h.stream));
        HIP_CHECK(hipMemcpyAsync(tau.data(), p_tau.p, tau.size() * 4, hipMemcpyDeviceToHost, h.stream));
        sync(h);
        int64_t sum = 0, over = 0, flagged = 0;
        int32_t mx = 0;
        int64_t first_over = -1, first_flag = -1;
        for (size_t b = 0; b < nb; ++b) {
          sum += cnt[b], mx = std::max(mx, cnt[b]), over += cnt[b] > pp.bucket_cap;
          if (cnt[b] > pp.bucket_cap && first_over < 0) first_over = (int64_t)b;
        }
        for (int c = 0; c < pp.S; ++c) {
          flagged += cnt[nb + (size_t)c] != 0;
          if (cnt[nb + (size_t)c] != 0 && first_flag < 0) first_flag = c;
        }
        if (over || flagged) fprintf(stderr, "[knn] first bucket over: %lld (rows from %lld), first chunk flagged: %lld\n", (long long)first_over, (long long)first_over * 32, (long long)first_flag);
        double tsum = 0.0;
        float tmin = 3e38f, tmax = -3e38f;
        for (int32_t r = 0; r < N; ++r) tsum += tau[(size_t)r], tmin = std::min(tmin, tau[(size_t)r]), tmax = std::max(tmax, tau[(size_t)r]);
        fprintf(stderr, "[knn] pieces %d: buckets %zu, entries per 32 rows mean %.0f max %d (cap %d), %lld buckets over, %lld of %d chunks flagged; tau / 256: mean %.4f min %.4f max %.4f; hit bound %.0f keep %d\n",
                pp.map.npieces, nb, (double)sum / (double)nb, mx, pp.bucket_cap, (long long)over, (long long)flagged, pp.S, tsum / N / 256.0,
                tmin / 256.0, tmax / 256.0, pp.hit_bound, pp.keep);
      }
      launch_panel_select(pp, rb_begin, rb_count, N, p_hits.p, p_hcnt.p, cval.p, cidx.p, fail_rows.p, fail_count.p,
                          h.stream, pp.sym ? &sym_dev : nullptr);
      if (knn_debug) fprintf(stderr, "[knn] after the select: %d rows without a candidate list\n", failed_so_far());
      // (single-process builds of rows of >= 384 columns: every undirected candidate pair is scored once -- two launches and two
      // N x keep scratch arrays.  A lookup in the partner's list costs keep x 4 bytes and a dependent round trip per candidate,
      // which short rows do not repay: build with / without, profiles/r06_rescore_ab.txt: config 3 12.07 / 12.54 ms, config 5
      // 91.7 / 96.1; with the cheaper selection tail of the round's last pass also 140 000 x 576 18.24 / 18.80, 200 000 x 512
      // 30.74 / 31.05, config 4 (384 columns) 390.8 / 393.6, 100 000 x 384 6.56 / 6.61; 40 000 x 256 a tie, 2.03 / 2.00)
      if ((h.knn_rescore_pair == 2 || (h.knn_rescore_pair == 1 && h.D >= 384)) && parts == 1 && !exchange && h.comm == nullptr && (int64_t)N * keep_f < ((int64_t)1 << 31)) {
        pair_sc.alloc((size_t)N * keep_f);
        pair_pos.alloc((size_t)N * keep_f);
      }
      launch_knn_rescore(plan, Yn.p, ldn, h.D, N, cidx.p, cval.p, k, delta, h.knn_val.p, h.knn_idx.p, fail_rows.p,
                         fail_count.p, h.stream, pp.scatter != 1 ? &pp.map : nullptr, pair_sc.p, pair_pos.p);
      if (knn_debug) fprintf(stderr, "[knn] after the re-scoring: %d rows unproven\n", failed_so_far());
    } else if (prefilter) {
      const KnnPlan plan = knn_plan(N, keep_f, slots, rb_begin, rb_count, true, h.knn_splits);
      const size_t ncand = (size_t)h.N * plan.S * plan.KC;
      cand_val.alloc(ncand);
      cand_idx.alloc(ncand);
      {
        ProfScope ps(h, 3);
        launch_knn_topk(plan, Yh.p, ldh / 2, N, cand_val.p, cand_idx.p, h.stream);
      }
      HIP_CHECK(hipMemsetAsync(cidx.p, 0xFF, (size_t)h.N * keep_f * 4, h.stream));
      launch_knn_merge(plan, cand_val.p, cand_idx.p, N, keep_f, cval.p, cidx.p, 0, h.stream);
      launch_knn_rescore(plan, Yn.p, ldn, h.D, N, cidx.p, cval.p, k, delta, h.knn_val.p, h.knn_idx.p, fail_rows.p,
                         fail_count.p, h.stream);
    } else if (parts == 1 && N <= dense_max && dense_small) {
      // small lattices: dense S + per-row argmax selection (the streaming kernel's first-tile inserts dominate here)
      const int32_t ldS = ((N + 31) / 32) * 32;
      DevBuf<float> Sm;
      Sm.alloc((size_t)N * ldS);
      ProfScope ps(h, 3);
      launch_knn_dense(Yn.p, ldn, N, k, Sm.p, ldS, h.knn_val.p, h.knn_idx.p, h.stream);
      sync(h);  // Sm goes back to the pool at scope exit
    } else {
      const KnnPlan plan = knn_plan(N, k, slots, rb_begin, rb_count, false, h.knn_splits);
      const size_t ncand = (size_t)h.N * plan.S * plan.KC;
      cand_val.alloc(ncand);
      cand_idx.alloc(ncand);
      {
        ProfScope ps(h, 3);
        launch_knn_topk(plan, Yn.p, ldn, N, cand_val.p, cand_idx.p, h.stream);
      }
      launch_knn_merge(plan, cand_val.p, cand_idx.p, N, k, h.knn_val.p, h.knn_idx.p, 1, h.stream);
    }
  }
  if (prefilter) {
    int32_t nfail = 0;
    HIP_CHECK(hipMemcpyAsync(&nfail, fail_count.p, 4, hipMemcpyDeviceToHost, h.stream));
    sync(h);
    DevBuf<int32_t> fail_rows2, fail_count2;
    int32_t* fail_list = fail_rows.p;
    if (panel && pp.sym && nfail > 0) {  // second-stage proof from the rows' whole buckets (knn_gemm.hip: k_bucket_rescore)
      fail_rows2.alloc((size_t)nfail);
      fail_count2.alloc(1);
      HIP_CHECK(hipMemsetAsync(fail_count2.p, 0, 4, h.stream));
      launch_bucket_rescore(pp, sym_dev, Yn.p, ldn, N, fail_rows.p, nfail, p_tau.p, k, delta, h.knn_val.p, h.knn_idx.p,
                            fail_rows2.p, fail_count2.p, h.stream);
      HIP_CHECK(hipMemcpyAsync(&nfail, fail_count2.p, 4, hipMemcpyDeviceToHost, h.stream));
      sync(h);
      fail_list = fail_rows2.p;
    }
    h.knn_fallback_rows = nfail;
    // (OSC_CREATE_FORCE_RETRY: test hook -- every streamed build gives up here)
    if (streamed && (nfail > std::max(64, N / 256) || h.create_force_retry)) return false;
    bool few_done = false;
    if (nfail > 0 && nfail <= 32) {  // a handful of rows: stream the columns once, select per row (0.15 vs 3.9 ms at N = 100k)
      const int32_t ldS = ((N + 31) / 32) * 32;
      DevBuf<float> Sm;
      Sm.alloc((size_t)nfail * ldS);
      few_done = launch_knn_few_rows(Yn.p, ldn, N, k, fail_list, nfail, Sm.p, ldS, h.knn_val.p, h.knn_idx.p, h.stream);
      if (few_done) sync(h);  // Sm goes back to the pool at scope exit
    }
    if (nfail > 0 && !few_done) {  // redo the unproven rows with the exact kernel (ties / dense clusters of near-equal scores)
      KnnPlan plan = knn_plan(N, k, slots, 0, (nfail + 127) / 128, false, h.knn_splits);
      plan.qrows = fail_list;
      plan.nq = nfail;
      const size_t ncand = (size_t)h.N * plan.S * plan.KC;
      cand_val.alloc(ncand);
      cand_idx.alloc(ncand);
      launch_knn_topk(plan, Yn.p, ldn, N, cand_val.p, cand_idx.p, h.stream);
      launch_knn_merge(plan, cand_val.p, cand_idx.p, N, k, h.knn_val.p, h.knn_idx.p, 1, h.stream);
    }
  }
  if (sharded && panel && pp.scatter != 1) {
    // a rank's rows are spread over the lattice (image row blocks): every row has exactly one writer, the others hold the
    // initial pattern (0.0f / -1), so a sum of the similarities and a max of the indices assemble the lists exactly
    const size_t cnt = (size_t)N * k;
    h.comm->allreduce(h.knn_val.p, cnt, COMM_F32, COMM_SUM, h.stream);
    h.comm->allreduce(h.knn_idx.p, cnt, COMM_I32, COMM_MAX, h.stream);
  } else if (sharded) {
    const size_t cnt = (size_t)rb_per * 128 * k;  // equal chunk per rank, in place
    h.comm->allgather(h.knn_val.p, cnt * 4, h.stream);
    h.comm->allgather(h.knn_idx.p, cnt * 4, h.stream);
  }
  alloc_ell(h, k);
  launch_mutual_ell(h.knn_val.p, h.knn_idx.p, N, k, h.width, h.ell_col.p, h.ell_a.p, h.deg.p, h.stream);
  DevBuf<float> scale;
  scale.alloc((size_t)h.N);
  launch_cap_and_normalize(h.ell_a.p, h.ell_w.p, h.ell_col.p, h.deg.p, h.width, N, h.row_cap, 1, scale.p,
                           h.sqrt_deg.p, h.stream);
  graph_counts(h);  // synchronises
  h.have_graph = true;
  maybe_reorder(h);
  h.build_ms = now_ms() - t0;
  return true;
}

