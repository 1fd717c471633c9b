// C ABI of liboscillink_hip.so (include/oscillink_hip.h): environment switches and the extern "C" entry points.  All
// device work of a handle goes to the handle's own HIP stream.  (The host side behind the entry points: osc_internal.hpp.)
#include "osc_internal.hpp"

thread_local std::string g_create_error;

bool env_num(const char* name, int& out) {
  const char* e = getenv(name);
  if (e) out = atoi(e);
  return e != nullptr;
}

// The switches a handle reads ONCE, at creation: how its solves run and how it is sharded.  They stay what they were when
// the graph is rebuilt (osc_rebuild_graph): the column window, the halo plan and the communicator were laid out for them
// (a handle whose OSC_SHARD changed under it would solve over a partial window).
void read_env_solver(L& h) {
  auto num = env_num;
  int v = 0;
  if (num("OSC_SPMM_SLAB", v)) h.spmm_slab = v < 0 ? -1 : (v / 4) * 4;
  if (num("OSC_SPMM_XS", v)) h.spmm_xs = v != 0 ? 1 : 0;
  if (num("OSC_XS_NB", v)) h.xs_nb = std::max(1, v);
  if (num("OSC_XS_GROUPS", v)) h.xs_groups_cap = v >= 8 ? 8 : v >= 4 ? 4 : v >= 2 ? 2 : 1;
  if (num("OSC_P_BLOCKED", v)) h.p_blocked = v != 0;
  if (num("OSC_SPMM_BLOCKED", v)) h.spmm_blocked = v;
  if (num("OSC_BLK_STAMP", v)) h.blk_stamp = v != 0;
  if (num("OSC_BLK_VARIANT", v)) h.blk_variant = (v >= 0 && v < blocked_variants()) ? v : -1;
  if (num("OSC_BLK_WIDE_MIN_ROWS", v)) h.blk_wide_min_rows = std::max(0, v);
  if (num("OSC_BLK_INIT", v)) {
    h.blk_init = v != 0;
    h.blk_init_fused = v == 1;
  }
  if (num("OSC_SPMM_DEEP", v)) h.spmm_deep = v != 0;
  if (num("OSC_COMM_OVERLAP", v)) h.comm_overlap = v != 0 ? 1 : 0;
  if (num("OSC_X_DEFER", v)) {
    h.x_defer = v != 0;
    h.x_last_form = v == 1;
  }
  if (num("OSC_SMALL_PATH", v)) h.small_path = v != 0;
  if (num("OSC_RECEIPT_PAIR", v)) h.receipt_pair = v != 0;
  if (const char* e = getenv("OSC_SHARD")) h.shard_mode = !strcmp(e, "row") ? 1 : 0;
  if (num("OSC_ROW_FAKE_SHARDS", v)) h.fake_row_shards = std::max(0, v);
  h.fake_col_w = 0;
  if (const char* e = getenv("OSC_FAKE_COL_SHARD")) {  // "r/w"
    int r = 0, w = 1;
    if (sscanf(e, "%d/%d", &r, &w) == 2 && w >= 1 && r >= 0 && r < w) h.fake_col_r = r, h.fake_col_w = w;
  }
}

// The switches of the lattice build: read at creation and again by every rebuild.
void read_env_build(L& h) {
  auto num = env_num;
  int v = 0;
  if (num("OSC_REORDER", v)) h.reorder = v != 0 ? 1 : 0;
  else h.reorder = -1;
  h.knn_mode = 0;
  if (const char* e = getenv("OSC_KNN_MODE")) h.knn_mode = !strcmp(e, "exact") ? 1 : !strcmp(e, "prefilter") ? 2 : !strcmp(e, "panel") ? 3 : 0;
  h.knn_fake_shards = 0;
  if (num("OSC_KNN_FAKE_SHARDS", v)) h.knn_fake_shards = std::max(0, v);
  h.knn_splits = 0;
  if (num("OSC_KNN_SPLITS", v)) h.knn_splits = std::max(1, v);
  h.knn_scatter = !(num("OSC_KNN_PANEL_SCATTER", v) && v == 0);
  h.knn_sym = !(num("OSC_KNN_PANEL_SYM", v) && v == 0);
  h.create_stream = !(num("OSC_CREATE_STREAM", v) && v == 0);
  h.create_force_retry = num("OSC_CREATE_FORCE_RETRY", v) && v != 0;
  h.create_min_mb = num("OSC_CREATE_MIN_MB", v) ? std::max(1, v) : 64;
  h.create_two_mb = num("OSC_CREATE_TWO_PIECE_MB", v) ? std::max(1, v) : 256;
  h.create_piece_mb_set = num("OSC_CREATE_PIECE_MB", v);
  h.create_piece_mb = h.create_piece_mb_set ? std::max(1, std::min(v, 1024)) : 24;
  h.knn_tune = KnnPanelTune{};
  h.knn_rescore_pair = num("OSC_KNN_RESCORE_PAIR", v) ? std::max(0, std::min(v, 2)) : 1;
  if (num("OSC_KNN_PANEL_NRG", v)) h.knn_tune.nrg = v;
  if (const char* e = getenv("OSC_KNN_PANEL_RHO")) h.knn_tune.rho = atof(e);
  if (num("OSC_KNN_PANEL_T", v)) h.knn_tune.T = v;
  if (num("OSC_KNN_PANEL_RANK", v)) h.knn_tune.rank = v;
  if (num("OSC_KNN_PANEL_SA", v)) h.knn_tune.sa = v;
  if (num("OSC_KNN_TILE_WIDE", v)) h.knn_tune.tile_wide = v != 0 ? 1 : 0;
  if (num("OSC_KNN_TILE_GROUP_MB", v)) h.knn_tune.tile_group_mb = v;
  h.knn_force_exchange = num("OSC_KNN_FORCE_EXCHANGE", v) && v != 0;
  h.bfs_host = num("OSC_BFS_HOST", v) && v != 0;
  h.halo_force = 0;
  if (const char* e = getenv("OSC_HALO")) h.halo_force = !strcmp(e, "full") ? 1 : !strcmp(e, "lists") ? 2 : 0;
}

void read_env(L& h) {
  read_env_solver(h);
  read_env_build(h);
}

void require_graph(L& h) {
  if (!h.have_graph) throw StateError("no lattice graph: build it (osc_create build_graph=1) or inject one (osc_set_csr)");
}

template <typename F>
int guarded(osc_handle h, F&& f) {
  if (!h) return OSC_E_INVALID;
  try {
    use_device(*h);
    alloc_ctx() = AllocCtx{h->device, h->stream};
    f(*h);
    return OSC_OK;
  } catch (const Invalid& e) {
    h->err = e.what();
    return OSC_E_INVALID;
  } catch (const StateError& e) {
    h->err = e.what();
    return OSC_E_STATE;
  } catch (const Unsupported& e) {
    h->err = e.what();
    return OSC_E_UNSUPPORTED;
  } catch (const CommError& e) {
    h->err = e.what();
    return OSC_E_COMM;
  } catch (const HipError& e) {
    h->err = e.what();
    return OSC_E_HIP;
  } catch (const std::exception& e) {
    h->err = e.what();
    return OSC_E_HIP;
  }
}



// =====================================================================================================
extern "C" {

const char* osc_version(void) { return "oscillink-hip 0.1.0 (gfx950)"; }

int osc_device_count(int32_t* n) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  if (n) *n = c;
  return OSC_OK;
}

int osc_device_name(int32_t device, char* out, int32_t cap) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return OSC_E_NODEVICE;
  snprintf(out, (size_t)cap, "%s|%s|CUs=%d", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return OSC_OK;
}

int osc_device_synchronize(int32_t device) {
  if (hipSetDevice(device) != hipSuccess) return OSC_E_NODEVICE;
  return hipDeviceSynchronize() == hipSuccess ? OSC_OK : OSC_E_HIP;
}

const char* osc_last_error(osc_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int osc_create(const float* Y, int64_t N, int32_t D, int32_t k, float row_cap, int32_t deterministic, int64_t seed,
               int32_t device, int32_t build, osc_handle* out) {
  if (!out) return OSC_E_INVALID;
  *out = nullptr;
  if (!Y || N < 1 || D < 1 || k < 1 || N >= (int64_t)1 << 31) {
    g_create_error = "osc_create: need Y != NULL, 1 <= N < 2^31, D >= 1, k >= 1";
    return OSC_E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) {
    g_create_error = "osc_create: no usable HIP device (this library has no CPU fallback)";
    return OSC_E_NODEVICE;
  }
  std::unique_ptr<osc_lattice> h(new osc_lattice());
  try {
    h->device = device;
    HIP_CHECK(hipSetDevice(device));
    {
      static std::mutex arch_mu;
      static std::map<int, std::string> arch;  // hipGetDeviceProperties is slow: ask once per device
      std::lock_guard<std::mutex> lk(arch_mu);
      auto it = arch.find(device);
      if (it == arch.end()) {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        it = arch.emplace(device, std::string(prop.gcnArchName)).first;
      }
      if (it->second.find("gfx950") == std::string::npos) {
        g_create_error = std::string("osc_create: device is ") + it->second + ", this build targets gfx950 only";
        return OSC_E_NODEVICE;
      }
    }
    h->stream = acquire_stream(device);
    alloc_ctx() = AllocCtx{device, h->stream};
    h->N = N;
    h->D = D;
    h->dcols = ((D + 3) / 4) * 4;
    h->ld = h->dcols;
    // Row pitch.  Gathered rows that straddle 128-byte lines cost the operator apply (stand-alone: pitch 772 vs 768
    // floats, 1.5-2.2 vs 1.06 ms; in the library at D = 1000: 12.2 ms per settle at pitch 1000, 10.7 at 1024), and a
    // pitch that is a multiple of 4 KB piles the rows of a column slab onto few L2 channels (D = 1000: 9.7 ms at pitch
    // 1056; 768 -> 800 changes nothing), so large lattices get line-aligned rows plus one line when the pitch would
    // be a multiple of 4 KB; small ones keep the dense pitch (their state lives in LDS / L2 anyway).
    // OSC_LD overrides (multiple of 4, >= D).
    if ((int64_t)N * D >= (int64_t)1 << 22) {
      h->ld = ((D + 31) / 32) * 32;
      if ((h->ld * 4) % 4096 == 0) h->ld += 32;
    }
    if (const char* e = getenv("OSC_LD")) {  // (row pitch override: needed before the arrays are sized, hence not in read_env)
      const int v = atoi(e);
      if (v >= h->dcols && v % 4 == 0) h->ld = v;
    }
    h->c0 = 0;
    h->c1 = h->dcols;
    h->k_eff = (int32_t)std::min<int64_t>(k, std::max<int64_t>(1, N - 1));
    h->row_cap = row_cap;
    h->deterministic = deterministic;
    h->seed = seed;
    read_env(*h);
    if (h->fake_col_w > 0) {  // measurement hook: work on rank r's column slab of w, no communicator
      std::tie(h->c0, h->c1) = host::column_shard(h->dcols, h->fake_col_r, h->fake_col_w);
      if (h->c1 <= h->c0) throw Invalid("OSC_FAKE_COL_SHARD: more ranks than 4-column groups");
    }
    const size_t n = (size_t)N * h->ld;
    for (DevBuf<float>* b : {&h->Y, &h->U, &h->X, &h->R, &h->P, &h->AP, &h->Ustar}) b->alloc(n);
    if (h->ld != D) HIP_CHECK(hipMemsetAsync(h->Y.p, 0, n * 4, h->stream));  // (the padding columns; the anchors follow below)
    for (DevBuf<float>* b : {&h->X, &h->R, &h->P, &h->AP, &h->Ustar}) HIP_CHECK(hipMemsetAsync(b->p, 0, n * 4, h->stream));
    h->B.alloc((size_t)N);
    std::vector<float> ones((size_t)N, 1.0f);
    HIP_CHECK(hipMemcpyAsync(h->B.p, ones.data(), (size_t)N * 4, hipMemcpyHostToDevice, h->stream));
    h->psi.alloc((size_t)h->ld);
    HIP_CHECK(hipMemsetAsync(h->psi.p, 0, (size_t)h->ld * 4, h->stream));
    // The anchors' way to the device (Y, and U = Y: lattice.py:56-58).  With a build to follow it is the build's business:
    // where it can it takes them piece by piece and works on what has arrived (osc_graph.hip: stream_pieces).
    if (build) {
      build_graph(*h, Y);
    } else {
      upload_rows(*h, h->Y.p, Y);
      HIP_CHECK(hipMemcpyAsync(h->U.p, h->Y.p, n * 4, hipMemcpyDeviceToDevice, h->stream));
    }
    sync(*h);
  } catch (const Unsupported& e) {
    g_create_error = e.what();
    return OSC_E_UNSUPPORTED;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return OSC_E_HIP;
  }
  *out = h.release();
  return OSC_OK;
}

int osc_host_alloc(int64_t bytes, void** out) {
  if (!out || bytes <= 0) return OSC_E_INVALID;
  *out = host_pool_alloc((size_t)bytes);
  return *out ? OSC_OK : OSC_E_HIP;
}

int osc_host_free(void* p) {
  if (!p) return OSC_OK;
  return host_pool_free(p) ? OSC_OK : OSC_E_INVALID;
}

int osc_destroy(osc_handle h) {
  if (!h) return OSC_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  alloc_ctx() = AllocCtx{h->device, nullptr};  // drained: the blocks go back to the pool without further waits
  delete h;
  return OSC_OK;
}

int osc_rebuild_graph(osc_handle h, int32_t k, float row_cap, int32_t deterministic, int64_t seed) {
  return guarded(h, [&](L& l) {
    if (k < 1) throw Invalid("kneighbors must be >= 1");
    read_env_build(l);  // (the build's switches only: solver and sharding switches are fixed at creation)
    l.k_eff = (int32_t)std::min<int64_t>(k, std::max<int64_t>(1, l.N - 1));
    l.row_cap = row_cap;
    l.deterministic = deterministic;
    l.seed = seed;
    build_graph(l);
  });
}

int osc_graph_stats(osc_handle h, int64_t* nnz, int32_t* max_deg, double* build_ms) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (nnz) *nnz = l.nnz;
    if (max_deg) *max_deg = l.max_deg;
    if (build_ms) *build_ms = l.build_ms;
  });
}

int osc_build_info(osc_handle h, int32_t* prefilter, int32_t* fallback_rows, int64_t* small_solves) {
  return guarded(h, [&](L& l) {
    if (prefilter) *prefilter = l.knn_prefilter ? (l.knn_panel ? 2 : 1) : 0;
    if (fallback_rows) *fallback_rows = l.knn_fallback_rows;
    if (small_solves) *small_solves = l.small_solves;
  });
}

int osc_apply_info(osc_handle h, int32_t* src_blocks, int64_t* blocked_applies) {
  return guarded(h, [&](L& l) {
    if (src_blocks) *src_blocks = l.blk_last;
    if (blocked_applies) *blocked_applies = l.blk_applies;
  });
}

int osc_get_blocked_copy(osc_handle h, int32_t nb, int32_t* slot_col, float* slot_w, int32_t* over_first, int32_t* over_count,
                         int32_t* over_col, float* over_w, int32_t over_cap) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (nb < 1 || nb > OSC_MAX_SRC_BLOCKS) throw Invalid("osc_get_blocked_copy: 1 <= nb <= 32");
    if (permuted(l)) throw Unsupported("osc_get_blocked_copy: the lattice is stored in an internal row order");
    const BlockedView bv = blocked_view(l, nb);
    const size_t ns = (size_t)nb * l.N * OSC_BLK_SLOTS;
    std::vector<int2> hs(ns), hr((size_t)l.N);
    HIP_CHECK(hipMemcpyAsync(hs.data(), bv.slots, ns * sizeof(int2), hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hr.data(), bv.rest, (size_t)l.N * sizeof(int2), hipMemcpyDeviceToHost, l.stream));
    sync(l);
    int64_t total = 0;
    for (int64_t i = 0; i < l.N; ++i) total = std::max<int64_t>(total, (int64_t)hr[(size_t)i].x + hr[(size_t)i].y);
    std::vector<int2> ho((size_t)total);
    if (total > 0) {
      HIP_CHECK(hipMemcpyAsync(ho.data(), bv.over, (size_t)total * sizeof(int2), hipMemcpyDeviceToHost, l.stream));
      sync(l);
    }
    for (size_t t = 0; t < ns; ++t) {
      if (slot_col) slot_col[t] = hs[t].x;
      if (slot_w) std::memcpy(slot_w + t, &hs[t].y, 4);
    }
    for (int64_t i = 0; i < l.N; ++i) {
      if (over_first) over_first[i] = hr[(size_t)i].x;
      if (over_count) over_count[i] = hr[(size_t)i].y;
    }
    if (total > over_cap && (over_col || over_w)) throw Invalid("osc_get_blocked_copy: over_cap too small");
    for (int64_t t = 0; t < total; ++t) {
      if (over_col) over_col[t] = ho[(size_t)t].x;
      if (over_w) std::memcpy(over_w + t, &ho[(size_t)t].y, 4);
    }
  });
}

int osc_order_info(osc_handle h, int32_t* reordered, double* clustering) {
  return guarded(h, [&](L& l) {
    if (reordered) *reordered = l.reordered ? 1 : 0;
    if (clustering) *clustering = l.clustering;
  });
}

int osc_get_row_order(osc_handle h, int32_t* perm) {
  return guarded(h, [&](L& l) {
    if (!perm) throw Invalid("osc_get_row_order: perm is NULL");
    for (int64_t i = 0; i < l.N; ++i) perm[i] = l.perm_h.empty() ? (int32_t)i : l.perm_h[(size_t)i];
  });
}

int osc_spmm_plan(osc_handle h, int32_t* launches, int32_t* slab_cols, int32_t* xs_workgroups) {
  return guarded(h, [&](L& l) {
    const int32_t ncols = l.c1 - l.c0;
    const int nb = xs_plan(l, ncols, cg_grid(l));
    const int32_t slab = nb ? 32 : auto_slab(l, ncols);
    if (launches) *launches = nb ? 1 : (ncols + slab - 1) / slab;
    if (slab_cols) *slab_cols = nb ? ncols : slab;
    if (xs_workgroups) *xs_workgroups = nb;
  });
}

int osc_get_csr(osc_handle h, int64_t* rowptr, int32_t* col, float* a, float* w, float* sqrt_deg) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    const size_t n = (size_t)l.N * l.width;
    std::vector<int32_t> hc(n), hd((size_t)l.N);
    std::vector<float> ha(n), hw(n);
    HIP_CHECK(hipMemcpyAsync(hc.data(), l.ell_col.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(ha.data(), l.ell_a.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hw.data(), l.ell_w.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hd.data(), l.deg.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (sqrt_deg) HIP_CHECK(hipMemcpyAsync(sqrt_deg, l.sqrt_deg.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    std::vector<float> hsd;
    if (sqrt_deg && permuted(l)) {  // downloaded in device order above: put it back into API order
      hsd.assign(sqrt_deg, sqrt_deg + l.N);
      for (int64_t i = 0; i < l.N; ++i) sqrt_deg[l.perm_h[(size_t)i]] = hsd[(size_t)i];
    }
    int64_t pos = 0;
    std::vector<std::pair<int32_t, size_t>> ent;  // (API column id, ELL slot) of one row, sorted by column
    for (int64_t r = 0; r < l.N; ++r) {  // r = API row
      const int64_t i = permuted(l) ? l.inv_h[(size_t)r] : r;  // device row
      if (rowptr) rowptr[r] = pos;
      ent.clear();
      for (int e = 0; e < hd[(size_t)i]; ++e) {
        const size_t o = (size_t)i * l.width + e;
        ent.emplace_back(permuted(l) ? l.perm_h[(size_t)hc[o]] : hc[o], o);
      }
      if (permuted(l)) std::sort(ent.begin(), ent.end());
      for (auto& pe : ent) {
        if (col) col[pos] = pe.first;
        if (a) a[pos] = ha[pe.second];
        if (w) w[pos] = hw[pe.second];
        ++pos;
      }
    }
    if (rowptr) rowptr[l.N] = pos;
  });
}

int osc_edge_prefix(osc_handle h, int32_t cap, int64_t* pairs, int32_t* n_out) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (cap < 0 || (cap > 0 && !pairs) || !n_out) throw Invalid("osc_edge_prefix: bad arguments");
    int32_t n = 0;
    std::vector<int32_t> hc, hd;
    std::vector<int32_t> ent;
    // rows are fetched in chunks until `cap` edges are collected (a few dozen rows at k = 32), not the whole graph
    for (int64_t r0 = 0; r0 < l.N && n < cap;) {
      const int64_t chunk = std::min<int64_t>(l.N - r0, permuted(l) ? 1 : 256);
      const int64_t d0 = permuted(l) ? l.inv_h[(size_t)r0] : r0;  // device row (chunk == 1 when rows are permuted)
      hc.resize((size_t)chunk * l.width);
      hd.resize((size_t)chunk);
      HIP_CHECK(hipMemcpyAsync(hc.data(), l.ell_col.p + (size_t)d0 * l.width, hc.size() * 4, hipMemcpyDeviceToHost, l.stream));
      HIP_CHECK(hipMemcpyAsync(hd.data(), l.deg.p + d0, (size_t)chunk * 4, hipMemcpyDeviceToHost, l.stream));
      sync(l);
      for (int64_t t = 0; t < chunk && n < cap; ++t) {
        ent.clear();
        for (int e = 0; e < hd[(size_t)t]; ++e) {
          const int32_t c = hc[(size_t)t * l.width + e];
          ent.push_back(permuted(l) ? l.perm_h[(size_t)c] : c);
        }
        if (permuted(l)) std::sort(ent.begin(), ent.end());
        for (int32_t c : ent) {
          if (n >= cap) break;
          pairs[2 * (size_t)n] = r0 + t;
          pairs[2 * (size_t)n + 1] = c;
          ++n;
        }
      }
      r0 += chunk;
    }
    *n_out = n;
  });
}

int osc_set_csr(osc_handle h, const int64_t* rowptr, const int32_t* col, const float* a) {
  return guarded(h, [&](L& l) {
    // validated and packed on the host (host_logic.hpp: sorted columns, no diagonal, no duplicates, symmetric)
    host::PackedEll pk;
    try {
      pk = host::pack_csr(l.N, rowptr, col, a);
    } catch (const host::InvalidArg& e) {
      throw Invalid(e.what());
    }
    const size_t W = (size_t)pk.width, n = (size_t)l.N * W;
    const std::vector<int32_t>&hc = pk.col, &hd = pk.deg;
    const std::vector<float>& ha = pk.a;
    // everything above ran on the host: a refused graph leaves the handle untouched
    if (l.have_graph) drop_order(l);  // the injected ids are API ids
    else {
      l.perm_h.clear();
      l.inv_h.clear();
    }
    alloc_ell(l, (int32_t)W);
    HIP_CHECK(hipMemcpyAsync(l.ell_col.p, hc.data(), n * 4, hipMemcpyHostToDevice, l.stream));
    HIP_CHECK(hipMemcpyAsync(l.ell_a.p, ha.data(), n * 4, hipMemcpyHostToDevice, l.stream));
    HIP_CHECK(hipMemcpyAsync(l.deg.p, hd.data(), (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
    launch_cap_and_normalize(l.ell_a.p, l.ell_w.p, l.ell_col.p, l.deg.p, l.width, (int32_t)l.N, 0.f, 0, nullptr,
                             l.sqrt_deg.p, l.stream);
    graph_counts(l);
    l.have_graph = true;
    l.have_ustar = false;
    l.knn_k = 0;
    maybe_reorder(l);
  });
}

int osc_get_knn_lists(osc_handle h, int32_t* idx, float* val, int32_t* k_eff) {
  return guarded(h, [&](L& l) {
    if (k_eff) *k_eff = l.knn_k;
    if (l.knn_k <= 0) return;
    const size_t n = (size_t)l.N * l.knn_k;
    if (idx) HIP_CHECK(hipMemcpyAsync(idx, l.knn_idx.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    if (val) HIP_CHECK(hipMemcpyAsync(val, l.knn_val.p, n * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
  });
}

int osc_set_query(osc_handle h, const float* psi, const float* gates) {
  return guarded(h, [&](L& l) {
    if (psi) {
      HIP_CHECK(hipMemsetAsync(l.psi.p, 0, (size_t)l.ld * 4, l.stream));
      HIP_CHECK(hipMemcpyAsync(l.psi.p, psi, (size_t)l.D * 4, hipMemcpyHostToDevice, l.stream));
    }
    std::vector<float> gp;
    if (gates && permuted(l)) {
      gp.resize((size_t)l.N);
      for (int64_t i = 0; i < l.N; ++i) gp[(size_t)i] = gates[l.perm_h[(size_t)i]];
      gates = gp.data();
    }
    if (gates) HIP_CHECK(hipMemcpyAsync(l.B.p, gates, (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
    sync(l);
    l.have_ustar = false;
  });
}

int osc_set_chain(osc_handle h, const int32_t* chain, const float* weights, int32_t len, float lamP) {
  return guarded(h, [&](L& l) {
    if (lamP < 0) throw Invalid("lamP must be >= 0");
    if (len < 2 || !chain) throw Invalid("chain must contain at least two indices");
    for (int i = 0; i < len; ++i)
      if (chain[i] < 0 || chain[i] >= l.N) throw Invalid("chain indices out of bounds");
    l.chain_nodes.assign(chain, chain + len);
    if (weights) l.chain_w.assign(weights, weights + (len - 1));
    else l.chain_w.clear();
    l.chain_present = true;
    install_chain(l);
    l.lamP = lamP;
    l.have_ustar = false;
  });
}

int osc_clear_chain(osc_handle h) {
  return guarded(h, [&](L& l) {
    l.chain_present = false;
    l.lamP = 0.0f;
    l.have_ustar = false;
    ++l.graph_epoch;
  });
}

int osc_set_lams(osc_handle h, float lamG, float lamC, float lamQ) {
  return guarded(h, [&](L& l) {
    if (!(lamG > 0)) throw Invalid("lamG must be > 0 for SPD");
    if (lamC < 0) throw Invalid("lamC must be >= 0");
    if (lamQ < 0) throw Invalid("lamQ must be >= 0");
    if (l.lamG != lamG || l.lamC != lamC || l.lamQ != lamQ) l.have_ustar = false;  // U* belongs to the old lams
    l.lamG = lamG;
    l.lamC = lamC;
    l.lamQ = lamQ;
  });
}

int osc_get_U(osc_handle h, float* out) {
  return guarded(h, [&](L& l) {
    if (!out) throw Invalid("osc_get_U: out is NULL");
    if (l.u_sharded) {  // collective in column-sharded runs
      gather_columns(l, l.U.p);
      l.u_sharded = false;
    }
    download_api_order(l, out, l.U.p);
  });
}

int osc_get_Y(osc_handle h, float* out) {
  return guarded(h, [&](L& l) {
    if (!out) throw Invalid("osc_get_Y: out is NULL");
    download_api_order(l, out, l.Y.p);
  });
}

int osc_set_U(osc_handle h, const float* U) {
  return guarded(h, [&](L& l) {
    if (U) {
      if (permuted(l)) {  // API order -> device order through the scratch array
        upload_rows(l, l.AP.p, U);
        launch_move_rows(l.U.p, l.AP.p, l.perm_d.p, l.N, l.ld, false, l.stream);
      } else {
        upload_rows(l, l.U.p, U);
      }
      l.u_sharded = false;
    } else if (l.comm && l.world > 1 && l.shard_mode == 0) {  // column-sharded: only this rank's slab is needed
      HIP_CHECK(hipMemcpy2DAsync(l.U.p + l.c0, (size_t)l.ld * 4, l.Y.p + l.c0, (size_t)l.ld * 4,
                                 (size_t)(l.c1 - l.c0) * 4, (size_t)l.N, hipMemcpyDeviceToDevice, l.stream));
      l.u_sharded = true;
    } else {
      HIP_CHECK(hipMemcpyAsync(l.U.p, l.Y.p, (size_t)l.N * l.ld * 4, hipMemcpyDeviceToDevice, l.stream));
      l.u_sharded = false;
    }
    if (U) sync(l);  // (the caller's buffer is free on return; the device-side reset is ordered by the stream)
  });
}

int osc_settle(osc_handle h, float dt, int32_t max_iters, float tol, int32_t precond, int32_t warm_start, float inertia,
               int32_t* iters, float* res, double* ms) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (max_iters < 1) throw Invalid("max_iters must be >= 1");
    ensure_cg_scratch(l, max_iters);
    const OpParams op = settle_op(l, dt, precond == OSC_PRECOND_JACOBI ? 1 : 0);
    sync(l);
    const double t0 = now_ms();
    // x0 (lattice.py:751-758)
    const float* x0 = l.U.p;
    if (!warm_start) {
      x0 = l.Y.p;
    } else {
      const float w = std::max(0.0f, std::min(1.0f, inertia));
      if (w > 0.0f) {
        launch_axpby(l.AP.p, l.Y.p, 1.0f - w, l.U.p, w, (int64_t)l.N * l.ld, l.stream);
        x0 = l.AP.p;  // AP is free until the first operator apply overwrites it (INIT gathers x0 before that)
      }
    }
    // Warm start from U itself (the default): the CG runs in place on U -- the old state is only read by the INIT pass
    // (as x0 and as the rhs term), which then has no x0 copy to write, and there is nothing to swap afterwards.
    const bool in_place = x0 == l.U.p && !row_mode(l);
    CgBuffers b{x0, in_place ? l.U.p : l.X.p, l.R.p, l.P.p, l.AP.p, l.U.p, l.Y.p, l.B.p, l.psi.p, l.ld, l.c0, l.c1};
    if (in_place) b.Xalt = l.X.p;  // free in an in-place solve
    // when x0 aliases AP the INIT pass reads it completely before the first SPMM_AP launch writes AP: same stream
    const CgResult r = run_cg(l, op, b, path_active(l), max_iters, tol);
    if (r.sol == l.X.p) l.U.swap(l.X);  // U <- U+ (lattice.py:206); an in-place solve left it in U already
    if (l.comm && l.world > 1 && l.shard_mode == 0) {
      // the swapped-in buffer only holds this rank's columns; the others are refreshed lazily by osc_get_U.
      l.u_sharded = true;
    }
    if (ms) *ms = now_ms() - t0;
    if (iters) *iters = r.iters;
    if (res) *res = r.res;
  });
}

int osc_solve_ustar(osc_handle h, float tol, int32_t max_iters, float* Ustar_out, int32_t* iters, float* res,
                    double* ms) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (max_iters < 1) throw Invalid("max_iters must be >= 1");
    ensure_cg_scratch(l, max_iters);
    const OpParams op = ustar_op(l);
    sync(l);
    const double t0 = now_ms();
    CgBuffers b{l.Y.p, l.X.p, l.R.p, l.P.p, l.AP.p, l.U.p, l.Y.p, l.B.p, l.psi.p, l.ld, l.c0, l.c1};
    b.kind = 1;
    const CgResult r = run_cg(l, op, b, path_active(l), max_iters, tol);
    l.Ustar.swap(l.X);
    if (l.shard_mode == 0) gather_columns(l, l.Ustar.p);  // receipts read whole rows of U* (row mode: already whole)
    l.have_ustar = true;
    if (ms) *ms = now_ms() - t0;
    if (iters) *iters = r.iters;
    if (res) *res = r.res;
    if (Ustar_out) download_api_order(l, Ustar_out, l.Ustar.p);
  });
}

int osc_has_ustar(osc_handle h, int32_t* yes) {
  return guarded(h, [&](L& l) {
    if (yes) *yes = l.have_ustar ? 1 : 0;
  });
}

int osc_get_ustar(osc_handle h, float* out) {
  return guarded(h, [&](L& l) {
    if (!out) throw Invalid("osc_get_ustar: out is NULL");
    if (!l.have_ustar) throw StateError("osc_get_ustar: no resident U* (call osc_solve_ustar first)");
    download_api_order(l, out, l.Ustar.p);
  });
}

int osc_get_rows(osc_handle h, int32_t which, const int32_t* rows, int32_t n, float* out) {
  return guarded(h, [&](L& l) {
    if (n < 0 || (n > 0 && (!rows || !out))) throw Invalid("osc_get_rows: bad arguments");
    const float* src = which == 0 ? l.Y.p : which == 1 ? l.U.p : which == 2 ? l.Ustar.p : nullptr;
    if (!src) throw Invalid("osc_get_rows: which must be 0 (Y), 1 (U) or 2 (U*)");
    if (which == 2 && !l.have_ustar) throw StateError("osc_get_rows: no resident U* (call osc_solve_ustar first)");
    if (which == 1 && l.u_sharded) {  // collective in column-sharded runs, like osc_get_U
      gather_columns(l, l.U.p);
      l.u_sharded = false;
    }
    if (n == 0) return;
    std::vector<int32_t> dev_rows((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
      if (rows[i] < 0 || rows[i] >= l.N) throw Invalid("osc_get_rows: row out of range");
      dev_rows[(size_t)i] = permuted(l) ? l.inv_h[(size_t)rows[i]] : rows[i];
    }
    DevBuf<int32_t> idx;
    DevBuf<float> tmp;
    idx.alloc((size_t)n);
    tmp.alloc((size_t)n * l.ld);
    HIP_CHECK(hipMemcpyAsync(idx.p, dev_rows.data(), (size_t)n * 4, hipMemcpyHostToDevice, l.stream));
    launch_move_rows(tmp.p, src, idx.p, n, l.ld, false, l.stream);  // tmp[i] = src[rows[i]]
    HIP_CHECK(hipMemcpy2DAsync(out, (size_t)l.D * 4, tmp.p, (size_t)l.ld * 4, (size_t)l.D * 4, (size_t)n,
                               hipMemcpyDeviceToHost, l.stream));
    sync(l);
  });
}

int osc_residual_history(osc_handle h, float* out, int32_t cap, int32_t* n) {
  return guarded(h, [&](L& l) {
    const int32_t m = std::min<int32_t>(cap, (int32_t)l.history.size());
    if (out)
      for (int32_t i = 0; i < m; ++i) out[i] = l.history[(size_t)i];
    if (n) *n = m;
  });
}

int osc_cg_single_rhs(osc_handle h, float gamma, const float* s, float tol, int32_t max_iters, float* h_out,
                      int32_t* iters, float* res) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!(gamma > 0)) throw Invalid("gamma must be > 0 for SPD");
    if (max_iters < 1) throw Invalid("max_iters must be >= 1");
    if (!s || !h_out) throw Invalid("osc_cg_single_rhs: NULL buffer");
    ensure_cg_scratch(l, max_iters);
    // N x 1 problem stored with pitch 4 (columns 1..3 stay zero)
    const int32_t ld1 = 4;
    const size_t n = (size_t)l.N * ld1;
    DevBuf<float> S, X0, X, R, P, AP, psi0;
    for (DevBuf<float>* b : {&S, &X0, &X, &R, &P, &AP}) {
      b->alloc(n);
      HIP_CHECK(hipMemsetAsync(b->p, 0, n * 4, l.stream));
    }
    psi0.alloc(ld1);
    HIP_CHECK(hipMemsetAsync(psi0.p, 0, ld1 * 4, l.stream));
    std::vector<float> sp;
    if (permuted(l)) {  // API order -> device order
      sp.resize((size_t)l.N);
      for (int64_t i = 0; i < l.N; ++i) sp[(size_t)i] = s[l.perm_h[(size_t)i]];
      s = sp.data();
    }
    // host <-> device contiguous through a scratch vector, re-pitched on the device: a strided copy of N 4-byte rows
    // from/to pageable memory costs more than the whole solve
    DevBuf<float> flat;
    flat.alloc((size_t)l.N);
    HIP_CHECK(hipMemcpyAsync(flat.p, s, (size_t)l.N * 4, hipMemcpyHostToDevice, l.stream));
    HIP_CHECK(hipMemcpy2DAsync(S.p, ld1 * 4, flat.p, 4, 4, (size_t)l.N, hipMemcpyDeviceToDevice, l.stream));
    OpParams op{};
    op.cs_const = 1.0f + gamma;  // (L_sym + gamma I) x = (1 + gamma) x - W x
    op.cs_B = 0.f;
    op.cW = 1.0f;
    op.cP = 0.f;
    op.md_const = 1.0f + gamma;  // diag(L_sym) + gamma (diffusion.py:138-139), diag(L_sym) = 1
    op.md_B = 0.f;
    op.precond = 1;
    op.rbU = 0.f;
    op.rbY = 1.0f;
    op.rbB = 0.f;
    CgBuffers b{X0.p, X.p, R.p, P.p, AP.p, S.p, S.p, l.B.p, psi0.p, ld1, 0, ld1};
    b.kind = 2;
    // scratch sized for ld >= 4 already
    std::unique_ptr<Comm> saved = std::move(l.comm);  // the diffusion solve is replicated, not sharded
    CgResult r;
    try {
      r = run_cg(l, op, b, false, max_iters, tol);
    } catch (...) {
      l.comm = std::move(saved);
      throw;
    }
    l.comm = std::move(saved);
    HIP_CHECK(hipMemcpy2DAsync(flat.p, 4, X.p, ld1 * 4, 4, (size_t)l.N, hipMemcpyDeviceToDevice, l.stream));
    HIP_CHECK(hipMemcpyAsync(h_out, flat.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    to_api_order(l, h_out);
    if (iters) *iters = r.iters;
    if (res) *res = r.res;
  });
}

// cosine of every row of a resident N x ld array to one host query (rows and query normalised with the +1e-12 of the
// reference); result in API row order
static void rows_cosine_to(L& l, const float* rows, const float* psi, float* out) {
  l.vec_q.alloc((size_t)l.D);
  l.vec_n.alloc((size_t)l.N);
  std::vector<float> qn((size_t)l.D);
  float ss = 0.f;
  for (int c = 0; c < l.D; ++c) ss += psi[c] * psi[c];
  const float inv = 1.0f / (std::sqrt(ss) + 1e-12f);
  for (int c = 0; c < l.D; ++c) qn[(size_t)c] = psi[c] * inv;
  HIP_CHECK(hipMemcpyAsync(l.vec_q.p, qn.data(), (size_t)l.D * 4, hipMemcpyHostToDevice, l.stream));
  launch_rows_cosine(rows, l.ld, l.vec_q.p, l.vec_n.p, l.N, l.D, l.stream);
  HIP_CHECK(hipMemcpyAsync(out, l.vec_n.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
  sync(l);
  to_api_order(l, out);
}

int osc_cosine_to(osc_handle h, const float* psi, float* out) {
  return guarded(h, [&](L& l) {
    if (!psi || !out) throw Invalid("osc_cosine_to: NULL buffer");
    rows_cosine_to(l, l.Y.p, psi, out);
  });
}

int osc_cosine_to_row(osc_handle h, int64_t row, float* out) {
  return guarded(h, [&](L& l) {
    if (!out) throw Invalid("osc_cosine_to_row: out is NULL");
    if (row < 0 || row >= l.N) throw Invalid("osc_cosine_to_row: row out of range");
    const int64_t dev_row = permuted(l) ? l.inv_h[(size_t)row] : row;
    std::vector<float> q((size_t)l.D);
    HIP_CHECK(hipMemcpyAsync(q.data(), l.Y.p + (size_t)dev_row * l.ld, (size_t)l.D * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    rows_cosine_to(l, l.Y.p, q.data(), out);
  });
}

int osc_mmr(osc_handle h, const float* scores, int32_t k, float lambda_div, int32_t* out_idx, int32_t* out_count) {
  return guarded(h, [&](L& l) {
    if (!scores || !out_idx || !out_count) throw Invalid("osc_mmr: NULL buffer");
    *out_count = 0;
    const int32_t want = (int32_t)std::min<int64_t>(std::max(k, 0), l.N);
    if (want <= 0) return;
    const int32_t N = (int32_t)l.N;
    // (1 - lambda) * score in fp64 and in device row order, like the NumPy float64 arithmetic this replaces
    std::vector<double> base((size_t)N);
    for (int32_t i = 0; i < N; ++i)
      base[(size_t)i] = (1.0 - (double)lambda_div) * (double)scores[permuted(l) ? l.perm_h[(size_t)i] : i];
    DevBuf<double> d_base, d_maxsim, d_pval;
    DevBuf<unsigned char> d_alive;
    DevBuf<int32_t> d_pid, d_prow, d_chosen;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>((N + 255) / 256, 256));
    d_base.alloc((size_t)N);
    d_maxsim.alloc((size_t)N);
    d_alive.alloc((size_t)N);
    const size_t nparts = std::max<size_t>((size_t)nblocks, (size_t)mmr_parts(N));  // (from the second step on: a partial per workgroup of the cosine pass)
    d_pval.alloc(nparts);
    d_pid.alloc(nparts);
    d_prow.alloc(nparts);
    d_chosen.alloc((size_t)want);
    l.vec_q.alloc((size_t)l.D);
    HIP_CHECK(hipMemcpyAsync(d_base.p, base.data(), (size_t)N * 8, hipMemcpyHostToDevice, l.stream));
    HIP_CHECK(hipMemsetAsync(d_alive.p, 1, (size_t)N, l.stream));
    MmrArgs a{};
    a.Y = l.Y.p;
    a.base = d_base.p;
    a.maxsim = d_maxsim.p;
    a.alive = d_alive.p;
    a.api_id = permuted(l) ? l.perm_d.p : nullptr;
    a.q = l.vec_q.p;
    a.pval = d_pval.p;
    a.pid = d_pid.p;
    a.prow = d_prow.p;
    a.chosen_api = d_chosen.p;
    a.N = N;
    a.D = l.D;
    a.ld = l.ld;
    a.nblocks = nblocks;
    a.lambda = (double)lambda_div;
    for (int step = 0; step < want; ++step) launch_mmr_step(a, step, l.stream);
    std::vector<int32_t> chosen((size_t)want);
    HIP_CHECK(hipMemcpyAsync(chosen.data(), d_chosen.p, (size_t)want * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    int32_t n = 0;
    for (; n < want && chosen[(size_t)n] >= 0; ++n) out_idx[n] = chosen[(size_t)n];
    *out_count = n;
  });
}

int osc_ustar_cosine_to(osc_handle h, const float* psi, float* out) {
  return guarded(h, [&](L& l) {
    if (!psi || !out) throw Invalid("osc_ustar_cosine_to: NULL buffer");
    if (!l.have_ustar) throw StateError("osc_ustar_cosine_to: no resident U* (call osc_solve_ustar first)");
    rows_cosine_to(l, l.Ustar.p, psi, out);
  });
}

// sum (A - B) . M (A - B) with the stationary operator M = lamG I + lamC L + lamQ B (+ lamP L_path)  (receipts.py:21-25)
// over this rank's share (column window / row block), completed over the communicator
static double quad_form_of_difference(L& l, const float* A, const float* B) {
  ensure_cg_scratch(l, 1);
  const int grid = cg_grid(l);
  // Large lattices on one process: the blocked matvec with its column sums of x . (M x) on the difference, which is formed on
  // its way into the slab-major operand (P; AP takes the product nobody reads): 0.76 instead of 1.23 ms at config 3, and
  // since round 6 without the row-major copy of the difference (one array pass of three less in front of the matvec).  Same
  // terms per row; a row's sum runs block by block instead of in list order (the matvec's 3e-8 relative on the state, far
  // inside the 1e-4 of the receipt's energies).  (A and B are never the solver's scratch arrays: U / U* / dynamics snapshots.)
  if (!row_mode(l) && l.comm == nullptr) {
    if (const int rows = blocked_quad_form(l, ustar_op(l), A, l.P.p, l.AP.p, path_active(l), B)) {
      launch_reduce_sum(l.part0.p, rows, l.ld, l.c0, l.c1, l.colsum.p, l.stream);
      std::vector<double> cs((size_t)l.ld, 0.0);
      HIP_CHECK(hipMemcpyAsync(cs.data() + l.c0, l.colsum.p + l.c0, (size_t)(l.c1 - l.c0) * 8, hipMemcpyDeviceToHost, l.stream));
      sync(l);
      double tot = 0.0;
      for (int c = l.c0; c < l.c1; ++c) tot += cs[(size_t)c];
      return tot;
    }
  }
  launch_axpby(l.P.p, A, 1.0f, B, -1.0f, (int64_t)l.N * l.ld, l.stream);
  SpmmArgs sa{};
  sa.g = graph_view(l, path_active(l));
  sa.op = ustar_op(l);
  sa.X = l.P.p;
  sa.B = l.B.p;
  sa.psi = l.psi.p;
  sa.ld = l.ld;
  sa.c0 = l.c0;
  sa.c1 = l.c1;
  sa.gate = nullptr;
  int nb = grid;
  if (row_mode(l)) {  // each rank (or fake shard) sums its own rows; the scalar is all-reduced below
    const std::vector<RowShard> shards = row_shards(l);
    nb = (int)shards.size() * grid;
    if (l.part0.n < (size_t)nb * l.ld) l.part0.alloc((size_t)nb * l.ld);
    for (size_t si = 0; si < shards.size(); ++si) {
      sa.row0 = shards[si].r0;
      sa.N = shards[si].r1;
      sa.part = l.part0.p + si * grid * l.ld;
      spmm_slabbed(l, SPMM_DOT, sa, grid);
    }
  } else {
    sa.part = l.part0.p;
    sa.N = l.N;
    spmm_slabbed(l, SPMM_DOT, sa, grid);
  }
  launch_reduce_sum(l.part0.p, nb, l.ld, l.c0, l.c1, l.colsum.p, l.stream);
  std::vector<double> cs((size_t)l.ld, 0.0);
  HIP_CHECK(hipMemcpyAsync(cs.data() + l.c0, l.colsum.p + l.c0, (size_t)(l.c1 - l.c0) * 8, hipMemcpyDeviceToHost,
                           l.stream));
  sync(l);
  double tot = 0.0;
  for (int c = l.c0; c < l.c1; ++c) tot += cs[(size_t)c];
  if (l.comm) {
    DevBuf<double> t;
    t.alloc(1);
    HIP_CHECK(hipMemcpyAsync(t.p, &tot, 8, hipMemcpyHostToDevice, l.stream));
    l.comm->allreduce(t.p, 1, COMM_F64, COMM_SUM, l.stream);
    HIP_CHECK(hipMemcpyAsync(&tot, t.p, 8, hipMemcpyDeviceToHost, l.stream));
    sync(l);
  }
  return tot;
}

int osc_deltaH(osc_handle h, double* dH) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!l.have_ustar) throw StateError("osc_deltaH: no resident U* (call osc_solve_ustar first)");
    if (!dH) throw Invalid("osc_deltaH: dH is NULL");
    *dH = quad_form_of_difference(l, l.U.p, l.Ustar.p);  // diff = U - U*
  });
}

// N x D host array in API row order -> N x ld device array in device row order (AP is scratch between solves)
static void upload_api_order(L& l, float* dst, const float* src) {
  if (permuted(l)) {
    upload_rows(l, l.AP.p, src);
    launch_move_rows(dst, l.AP.p, l.perm_d.p, l.N, l.ld, false, l.stream);
  } else {
    upload_rows(l, dst, src);
  }
}

int osc_dynamics_snapshot(osc_handle h) {
  return guarded(h, [&](L& l) {
    if (l.u_sharded) {  // collective in column-sharded runs, like osc_get_U
      gather_columns(l, l.U.p);
      l.u_sharded = false;
    }
    l.Uprev.alloc((size_t)l.N * l.ld);
    HIP_CHECK(hipMemcpyAsync(l.Uprev.p, l.U.p, (size_t)l.N * l.ld * 4, hipMemcpyDeviceToDevice, l.stream));
    l.have_uprev = true;
  });
}

int osc_dynamics(osc_handle h, const float* U_prev, const float* U_next, double* move2_mean, float* move2_max,
                 double* step_deltaH, double* flow_total, int32_t top_cap, int32_t* top_i, int32_t* top_j,
                 double* top_flow, int32_t* top_n, int32_t* radius) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (top_cap < 0 || top_cap > 32 || (top_cap > 0 && (!top_i || !top_j || !top_flow || !top_n)))
      throw Invalid("osc_dynamics: top_cap must be 0..32 with output arrays");
    const size_t n = (size_t)l.N * l.ld;
    if (U_prev) {
      l.Uprev.alloc(n);
      HIP_CHECK(hipMemsetAsync(l.Uprev.p, 0, n * 4, l.stream));
      upload_api_order(l, l.Uprev.p, U_prev);
      l.have_uprev = true;
    }
    if (!l.have_uprev) throw StateError("osc_dynamics: no previous state (call osc_dynamics_snapshot before the settle)");
    if (l.u_sharded) {
      gather_columns(l, l.U.p);
      l.u_sharded = false;
    }
    const float* next = l.U.p;
    DevBuf<float> next_buf;
    if (U_next) {
      next_buf.alloc(n);
      HIP_CHECK(hipMemsetAsync(next_buf.p, 0, n * 4, l.stream));
      upload_api_order(l, next_buf.p, U_next);
      next = next_buf.p;
    }
    // step energy (lattice.py:843-857): the quadratic form of the stationary operator on U_prev - U_next
    const double dH = quad_form_of_difference(l, l.Uprev.p, next);
    if (step_deltaH) *step_deltaH = dH;
    // movement per node and structural energy drop per edge: the receipt-rows pass with (Y, U*) := (U_prev, U_next)
    DevBuf<float> move2, flows;
    move2.alloc((size_t)l.N);
    const size_t ne = (size_t)l.N * l.width;
    flows.alloc(ne);
    HIP_CHECK(hipMemsetAsync(flows.p, 0, ne * 4, l.stream));
    ReceiptArgs a{};
    a.Y = l.Uprev.p;
    a.Ustar = next;
    a.psi = l.psi.p;
    a.B = l.B.p;
    a.sqrt_deg = l.sqrt_deg.p;
    a.col = l.ell_col.p;
    a.adj = l.ell_a.p;
    a.deg = l.deg.p;
    a.width = l.width;
    a.lamG = 1.0f;  // anchor term of the pass = ||U_next - U_prev||^2 per node
    a.lamC = l.lamC;
    a.lamQ = 0.0f;
    a.z_th = 0.0f;
    a.anchor = move2.p;
    a.N = (int32_t)l.N;
    a.D = l.D;
    a.ld = l.ld;
    a.api_id = nullptr;
    a.edge_flow = flows.p;
    launch_receipt_rows(a, l.stream);
    // reductions: sum / max of the movement, sum of the flows
    const int nb1 = (int)std::max<int64_t>(1, std::min<int64_t>((l.N + 255) / 256, 256));
    const int nb2 = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ne + 255) / 256, 256));
    DevBuf<double> ps;
    DevBuf<float> pm;
    ps.alloc((size_t)nb1 + nb2);
    pm.alloc((size_t)nb1 + nb2);
    launch_sum_max(move2.p, l.N, nb1, ps.p, pm.p, l.stream);
    launch_sum_max(flows.p, (int64_t)ne, nb2, ps.p + nb1, pm.p + nb1, l.stream);
    // top flows: per-block candidates (twice the cap, so both directions of a tied pair survive a block's cut)
    const int K = top_cap > 0 ? 32 : 0;
    const int nb3 = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ne + 4095) / 4096, 256));
    DevBuf<float> tv;
    DevBuf<int64_t> ti;
    DevBuf<int32_t> tc;
    std::vector<float> htv;
    std::vector<int64_t> hti;
    std::vector<int32_t> htc;
    if (K > 0) {
      tv.alloc((size_t)nb3 * K);
      ti.alloc((size_t)nb3 * K);
      tc.alloc((size_t)nb3 * K);
      launch_top_select(flows.p, l.ell_col.p, permuted(l) ? l.perm_d.p : nullptr, l.width, (int64_t)ne, nb3, K, tv.p, ti.p, tc.p,
                        l.stream);
      htv.resize((size_t)nb3 * K);
      hti.resize((size_t)nb3 * K);
      htc.resize((size_t)nb3 * K);
      HIP_CHECK(hipMemcpyAsync(htv.data(), tv.p, htv.size() * 4, hipMemcpyDeviceToHost, l.stream));
      HIP_CHECK(hipMemcpyAsync(hti.data(), ti.p, hti.size() * 8, hipMemcpyDeviceToHost, l.stream));
      HIP_CHECK(hipMemcpyAsync(htc.data(), tc.p, htc.size() * 4, hipMemcpyDeviceToHost, l.stream));
    }
    std::vector<double> hps((size_t)nb1 + nb2);
    std::vector<float> hpm((size_t)nb1 + nb2);
    HIP_CHECK(hipMemcpyAsync(hps.data(), ps.p, hps.size() * 8, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hpm.data(), pm.p, hpm.size() * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    double msum = 0.0, fsum = 0.0;
    float mmax = 0.f;
    for (int b = 0; b < nb1; ++b) {
      msum += hps[(size_t)b];
      mmax = (hpm[(size_t)b] != hpm[(size_t)b] || mmax != mmax) ? NAN : std::max(mmax, hpm[(size_t)b]);
    }
    for (int b = 0; b < nb2; ++b) fsum += hps[(size_t)nb1 + b];
    if (move2_mean) *move2_mean = msum / (double)l.N;
    if (move2_max) *move2_max = mmax;
    if (flow_total) *flow_total = fsum;
    if (K > 0) {
      struct Cand {
        double f;
        int32_t i, j;
      };
      std::vector<Cand> cands;
      for (size_t t = 0; t < hti.size(); ++t) {
        if (hti[t] < 0) continue;
        const int32_t row = (int32_t)(hti[t] / l.width);
        cands.push_back({(double)htv[t], permuted(l) ? l.perm_h[(size_t)row] : row,
                         permuted(l) ? l.perm_h[(size_t)htc[t]] : htc[t]});
      }
      // the reference keeps the flows in argwhere order and sorts them stably by flow, descending (lattice.py:879-881)
      std::sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) {
        if (x.f != y.f) return x.f > y.f;
        if (x.i != y.i) return x.i < y.i;
        return x.j < y.j;
      });
      const int32_t m = (int32_t)std::min<size_t>((size_t)top_cap, cands.size());
      for (int32_t t = 0; t < m; ++t) {
        top_i[t] = cands[(size_t)t].i;
        top_j[t] = cands[(size_t)t].j;
        top_flow[t] = cands[(size_t)t].f;
      }
      *top_n = m;
    } else if (top_n) {
      *top_n = 0;
    }
    // coherence radius (lattice.py:885-892, 905-927): BFS from the nodes that moved at least 10 % of the largest move
    if (radius) {
      *radius = 0;
      const float maxinf = std::sqrt(mmax + 1e-12f);
      if (l.N > 0 && maxinf > 1e-9f) {
        const float thr = (float)(0.1 * (double)maxinf);
        DevBuf<int32_t> dist, changed;
        dist.alloc((size_t)l.N);
        changed.alloc(1);
        launch_bfs_seeds(move2.p, l.N, thr, dist.p, l.stream);
        int32_t level = 0;
        for (; level < l.N; ++level) {
          int32_t ch = 0;
          HIP_CHECK(hipMemsetAsync(changed.p, 0, 4, l.stream));
          launch_bfs_level(l.ell_col.p, l.deg.p, l.width, l.N, dist.p, level, changed.p, l.stream);
          HIP_CHECK(hipMemcpyAsync(&ch, changed.p, 4, hipMemcpyDeviceToHost, l.stream));
          sync(l);
          if (!ch) break;
        }
        *radius = level;
      }
    }
  });
}

// null-point candidates per device row -> compact list in API row order (the reference walks rows 0..N-1)
static int32_t compact_nulls(const L& l, const std::vector<int32_t>& hj, const std::vector<float>& hz,
                             const std::vector<float>& hr, int32_t* i_out, int32_t* j_out, float* z_out, float* r_out) {
  int32_t n = 0;
  for (int64_t r = 0; r < l.N; ++r) {  // r = API row
    const size_t i = (size_t)(l.perm_h.empty() ? r : l.inv_h[(size_t)r]);
    if (hj[i] < 0) continue;
    if (i_out) i_out[n] = (int32_t)r;
    if (j_out) j_out[n] = l.perm_h.empty() ? hj[i] : l.perm_h[(size_t)hj[i]];
    if (z_out) z_out[n] = hz[i];
    if (r_out) r_out[n] = hr[i];
    ++n;
  }
  return n;
}

static void receipt_rows(L& l, float z_th, DevBuf<float>& coh, DevBuf<float>& an, DevBuf<float>& qu,
                         DevBuf<int32_t>& nj, DevBuf<float>& nz, DevBuf<float>& nr, bool want_comp, bool want_null) {
  ReceiptArgs a{};
  a.Y = l.Y.p;
  a.Ustar = l.Ustar.p;
  a.psi = l.psi.p;
  a.B = l.B.p;
  a.sqrt_deg = l.sqrt_deg.p;
  a.col = l.ell_col.p;
  a.adj = l.ell_a.p;
  a.deg = l.deg.p;
  a.width = l.width;
  a.lamG = l.lamG;
  a.lamC = l.lamC;
  a.lamQ = l.lamQ;
  a.z_th = z_th;
  a.N = (int32_t)l.N;
  a.D = l.D;
  a.ld = l.ld;
  a.api_id = permuted(l) ? l.perm_d.p : nullptr;
  if (want_comp) {
    coh.alloc((size_t)l.N);
    an.alloc((size_t)l.N);
    qu.alloc((size_t)l.N);
    a.coh = coh.p;
    a.anchor = an.p;
    a.query = qu.p;
  }
  if (want_null) {
    nj.alloc((size_t)l.N);
    nz.alloc((size_t)l.N);
    nr.alloc((size_t)l.N);
    a.null_j = nj.p;
    a.null_z = nz.p;
    a.null_r = nr.p;
  }
  // the pair form (receipt_kernels.hip): every edge's squared distances computed once -- half the gathers of the one-launch
  // kernel, the same outputs bit for bit (OSC_RECEIPT_PAIR=0: the one-launch kernel; tests, A/B)
  DevBuf<float> pdy, pdu;
  DevBuf<int32_t> pfail;
  const bool pair = l.receipt_pair && (l.ld + 255) / 256 <= 6 && (int64_t)l.N * l.width < ((int64_t)1 << 31) && l.N >= 4096;
  if (pair) {
    pdy.alloc((size_t)l.N * l.width);
    pdu.alloc((size_t)l.N * l.width);
    pfail.alloc(1);
    HIP_CHECK(hipMemsetAsync(pfail.p, 0, 4, l.stream));
    a.pair_dy = pdy.p;
    a.pair_du = pdu.p;
    a.pair_fail = pfail.p;
  }
  launch_receipt_rows(a, l.stream);
  if (pair) {
    int32_t fail = 0;
    HIP_CHECK(hipMemcpyAsync(&fail, pfail.p, 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    if (fail != 0) {  // an edge without its mirror (asymmetric lists: not produced by the build or accepted by the injection)
      a.pair_dy = a.pair_du = nullptr;
      a.pair_fail = nullptr;
      launch_receipt_rows(a, l.stream);
    }
  }
}

int osc_receipt_components(osc_handle h, float* coh, float* anchor, float* query) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!l.have_ustar) throw StateError("osc_receipt_components: no resident U*");
    DevBuf<float> c, a, q, nz, nr;
    DevBuf<int32_t> nj;
    receipt_rows(l, 3.0f, c, a, q, nj, nz, nr, true, false);
    if (coh) HIP_CHECK(hipMemcpyAsync(coh, c.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (anchor) HIP_CHECK(hipMemcpyAsync(anchor, a.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (query) HIP_CHECK(hipMemcpyAsync(query, q.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    to_api_order(l, coh);
    to_api_order(l, anchor);
    to_api_order(l, query);
  });
}

int osc_null_points(osc_handle h, float z_th, int32_t* i_out, int32_t* j_out, float* z_out, float* r_out,
                    int32_t* count) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!l.have_ustar) throw StateError("osc_null_points: no resident U*");
    if (!count) throw Invalid("osc_null_points: count is NULL");
    DevBuf<float> c, a, q, nz, nr;
    DevBuf<int32_t> nj;
    receipt_rows(l, z_th, c, a, q, nj, nz, nr, false, true);
    std::vector<int32_t> hj((size_t)l.N);
    std::vector<float> hz((size_t)l.N), hr((size_t)l.N);
    HIP_CHECK(hipMemcpyAsync(hj.data(), nj.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hz.data(), nz.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hr.data(), nr.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    *count = compact_nulls(l, hj, hz, hr, i_out, j_out, z_out, r_out);
  });
}

int osc_receipt_rows(osc_handle h, float z_th, float* coh, float* anchor, float* query, int32_t* i_out, int32_t* j_out,
                     float* z_out, float* r_out, int32_t* count) {
  return guarded(h, [&](L& l) {
    require_graph(l);
    if (!l.have_ustar) throw StateError("osc_receipt_rows: no resident U*");
    if (!count) throw Invalid("osc_receipt_rows: count is NULL");
    DevBuf<float> c, a, q, nz, nr;
    DevBuf<int32_t> nj;
    receipt_rows(l, z_th, c, a, q, nj, nz, nr, true, true);
    std::vector<int32_t> hj((size_t)l.N);
    std::vector<float> hz((size_t)l.N), hr((size_t)l.N);
    if (coh) HIP_CHECK(hipMemcpyAsync(coh, c.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (anchor) HIP_CHECK(hipMemcpyAsync(anchor, a.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    if (query) HIP_CHECK(hipMemcpyAsync(query, q.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hj.data(), nj.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hz.data(), nz.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    HIP_CHECK(hipMemcpyAsync(hr.data(), nr.p, (size_t)l.N * 4, hipMemcpyDeviceToHost, l.stream));
    sync(l);
    to_api_order(l, coh);
    to_api_order(l, anchor);
    to_api_order(l, query);
    *count = compact_nulls(l, hj, hz, hr, i_out, j_out, z_out, r_out);
  });
}

int osc_profile_enable(osc_handle h, int32_t on) {
  return guarded(h, [&](L& l) {
    if (!on) prof_drain(l);
    l.prof_on = on != 0;
  });
}
int osc_profile_reset(osc_handle h) {
  return guarded(h, [&](L& l) {
    prof_drain(l);
    for (int i = 0; i < 5; ++i) {
      l.prof_count[i] = 0;
      l.prof_ms[i] = 0.0;
    }
    if (l.blk_stamps.n) {
      HIP_CHECK(hipMemsetAsync(l.blk_stamps.p, 0, l.blk_stamps.n * 8, l.stream));
      l.blk_stamp_launches = 0;
    }
  });
}
int osc_profile_get(osc_handle h, int32_t which, int64_t* launches, double* total_ms) {
  return guarded(h, [&](L& l) {
    if (which == 16) {  // main sweep of the last build's prefilter: 0 none, 1 full (per rank), 2 half (one per build)
      if (launches) *launches = l.knn_sweep;
      if (total_ms) *total_ms = 0.0;
      return;
    }
    if (which == 15) {  // pieces the last build received its anchors in (0: they were on the device before it started)
      if (launches) *launches = l.create_pieces;
      if (total_ms) *total_ms = 0.0;
      return;
    }
    if (which == 14) {  // the kernel shape of the last blocked matvec
      if (launches) *launches = l.blk_shape_last;
      if (total_ms) *total_ms = 0.0;
      return;
    }
    if (which >= 8 && which <= 13) {
      // cycle stamps of the blocked matvec (OSC_BLK_STAMP=1): mean over the waves of a role of the shader cycles summed
      // over the stamped launches.  8-11: gathering waves' lifetime / gather rounds / barrier / epilogue; 12-13: the list
      // waves' fetch / barrier.  *launches = stamped launches (speculative, gated-off ones included: they add ~nothing).
      prof_drain(l);
      sync(l);
      const int wpg = blocked_gather_waves(l.blk_shape_last) + 1;
      std::vector<unsigned long long> w(l.blk_stamps.n);
      if (!w.empty()) HIP_CHECK(hipMemcpy(w.data(), l.blk_stamps.p, w.size() * 8, hipMemcpyDeviceToHost));
      double sum = 0.0;
      int64_t cnt = 0;
      for (size_t i = 0; i + 3 < w.size(); i += 4) {
        const bool list_wave = (int)((i / 4) % (size_t)wpg) == wpg - 1;
        if (w[i] == 0 || list_wave != (which >= 12)) continue;  // (workgroups that took no part have no stamps)
        sum += (double)w[i + (which >= 12 ? which - 11 : which - 8)];
        cnt += 1;
      }
      if (launches) *launches = l.blk_stamp_launches;
      if (total_ms) *total_ms = cnt ? sum / (double)cnt : 0.0;
      return;
    }
    if (which < 0 || which > 4) throw Invalid("osc_profile_get: which must be 0..4 (or 8..14: blocked matvec diagnostics)");
    prof_drain(l);
    if (launches) *launches = l.prof_count[which];
    if (total_ms) *total_ms = l.prof_ms[which];
  });
}

int osc_comm_unique_id(char id_out[128]) {
  try {
    comm_rccl_id(id_out);
  } catch (const std::exception&) {
    return OSC_E_COMM;
  }
  return OSC_OK;
}

int osc_comm_backend_version(int32_t* version) {
  if (!version) return OSC_E_INVALID;
  *version = comm_rccl_version();
  return OSC_OK;
}

int osc_comm_loopback_id(char id_out[128]) {
  comm_loopback_id(id_out);
  return OSC_OK;
}

int osc_comm_init(osc_handle h, const char id[128], int32_t rank, int32_t world) {
  return guarded(h, [&](L& l) {
    if (world < 1 || rank < 0 || rank >= world) throw Invalid("osc_comm_init: bad rank/world");
    drain_comm_stream(l);
    l.comm.reset();
    l.rank = rank;
    l.world = world;
    l.fake_window = false;
    std::tie(l.c0, l.c1) = host::column_shard(l.dcols, rank, world);  // slabs in units of 4 floats, as even as possible
    if (l.shard_mode == 1) {  // row-sharded CG: every rank works on all columns of its row block
      l.c0 = 0;
      l.c1 = l.dcols;
    }
    if (world == 1 && l.shard_mode == 0 && l.fake_col_w > 0) {  // measurement hook (OSC_FAKE_COL_SHARD, read at osc_create): one
      std::tie(l.c0, l.c1) = host::column_shard(l.dcols, l.fake_col_r, l.fake_col_w);  // rank's window of a wider solve, now WITH
      l.fake_window = l.fake_col_w > 1;  // the communicator machinery around it; osc_comm_info says so
    }
    if (l.c1 <= l.c0) throw Invalid("osc_comm_init: more ranks than 4-column groups");
    l.comm = comm_create(id, rank, world, l.device);
    ++l.graph_epoch;
    l.have_ustar = false;
  });
}

int osc_comm_allreduce_f64(osc_handle h, double* vals, int32_t n, int32_t op) {
  return guarded(h, [&](L& l) {
    if (n < 0 || (n > 0 && !vals) || (op != 0 && op != 1)) throw Invalid("osc_comm_allreduce_f64: bad arguments");
    sync(l);  // also the barrier use: everything this rank enqueued so far has finished
    if (!l.comm || n == 0) {
      if (l.comm) {  // n == 0: pure barrier
        DevBuf<double> t;
        t.alloc(1);
        HIP_CHECK(hipMemsetAsync(t.p, 0, 8, l.stream));
        l.comm->allreduce(t.p, 1, COMM_F64, COMM_SUM, l.stream);
        sync(l);
      }
      return;
    }
    DevBuf<double> t;
    t.alloc((size_t)n);
    HIP_CHECK(hipMemcpyAsync(t.p, vals, (size_t)n * 8, hipMemcpyHostToDevice, l.stream));
    l.comm->allreduce(t.p, (size_t)n, COMM_F64, op == 0 ? COMM_SUM : COMM_MAX, l.stream);
    HIP_CHECK(hipMemcpyAsync(vals, t.p, (size_t)n * 8, hipMemcpyDeviceToHost, l.stream));
    sync(l);
  });
}

int osc_comm_info(osc_handle h, int32_t* rank, int32_t* world, int32_t* shard_mode, char* kind_out, int32_t cap) {
  return guarded(h, [&](L& l) {
    if (rank) *rank = l.comm ? l.rank : 0;
    if (world) *world = l.comm ? l.world : 1;
    if (shard_mode) *shard_mode = l.shard_mode;
    if (kind_out && cap > 0)
      snprintf(kind_out, (size_t)cap, "%s%s", l.comm ? l.comm->kind() : "none", l.comm && l.fake_window ? "+fake-window" : "");
  });
}

int osc_halo_info(osc_handle h, int64_t* need_rows, int64_t* need_rows_max, int64_t* remote_rows,
                  int64_t* bytes_per_iteration, int32_t* full_exchange) {
  return guarded(h, [&](L& l) {
    if (!l.comm || l.shard_mode != 1) throw StateError("osc_halo_info: no row-sharded communicator on this handle");
    require_graph(l);
    if (l.halo.epoch != l.graph_epoch) build_halo_plan(l);
    const int64_t own = l.N * (l.rank + 1) / l.world - l.N * l.rank / l.world;
    if (need_rows) *need_rows = l.halo.need_rows;
    if (need_rows_max) *need_rows_max = l.halo.need_rows_max;
    if (remote_rows) *remote_rows = l.N - own;
    if (bytes_per_iteration) *bytes_per_iteration = (l.halo.full ? (l.N - own) : l.halo.need_rows) * (int64_t)l.ld * 4;
    if (full_exchange) *full_exchange = l.halo.full ? 1 : 0;
  });
}

int osc_comm_shard(osc_handle h, int32_t* c0, int32_t* c1) {
  return guarded(h, [&](L& l) {
    if (c0) *c0 = l.c0;
    if (c1) *c1 = std::min(l.c1, l.D);
  });
}

}  // extern "C"
