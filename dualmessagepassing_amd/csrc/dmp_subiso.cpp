// Exact subgraph-isomorphism enumeration on the host (the labels of a synthetic dataset in the
// reference's format -- `counts` and `subisomorphisms` of every (pattern, graph) pair, utils/io.py:99-115 -- need an exact
// counter; the reference ships its datasets with these columns precomputed).  Native C++ (no GPU): a depth-first search
// over the pattern's nodes in id order, candidates in ascending graph-node id, so the matches come out in lexicographic
// order of the node map -- the order of dualmessagepassing_amd.harness.enumerate_subisomorphisms, its plain-Python twin.
//
// A match: an injective, node-label-preserving map f of the pattern's nodes into the graph's nodes such that every
// pattern edge (u -> v, label l) has a graph edge (f(u) -> f(v)) with label l (multigraphs: any parallel edge with that
// label; non-induced).  dmp_subiso_batch runs many pairs over a pool of host threads.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/dmp_hip.h"

namespace {

struct Pair {
  int64_t pn, pm, gn, gm;
  const int64_t *ps, *pd, *pvl, *pel, *gs, *gd, *gvl, *gel;
};

struct Searcher {
  const Pair &p;
  std::vector<std::vector<int64_t>> out_labels;                 // graph adjacency: (u, v) -> sorted labels
  std::unordered_map<uint64_t, int> slot;
  std::vector<std::vector<int>> touching;                       // pattern edges whose later endpoint is node k
  std::vector<int64_t> map;
  std::vector<char> used;
  std::vector<std::vector<int64_t>> cand;                       // graph nodes by label, ascending
  int64_t count = 0, limit, *rows, cap;

  Searcher(const Pair &pr, int64_t *rows_, int64_t cap_, int64_t limit_) : p(pr), limit(limit_), rows(rows_), cap(cap_) {
    for (int64_t e = 0; e < p.gm; ++e) {
      const uint64_t key = (uint64_t)p.gs[e] * (uint64_t)p.gn + (uint64_t)p.gd[e];
      auto it = slot.find(key);
      if (it == slot.end()) { it = slot.emplace(key, (int)out_labels.size()).first; out_labels.emplace_back(); }
      out_labels[it->second].push_back(p.gel[e]);
    }
    for (auto &v : out_labels) std::sort(v.begin(), v.end());
    touching.resize((size_t)p.pn);
    for (int64_t e = 0; e < p.pm; ++e) touching[(size_t)std::max(p.ps[e], p.pd[e])].push_back((int)e);
    std::unordered_map<int64_t, int> by_label;
    for (int64_t v = 0; v < p.gn; ++v) {
      auto it = by_label.find(p.gvl[v]);
      if (it == by_label.end()) { it = by_label.emplace(p.gvl[v], (int)cand.size()).first; cand.emplace_back(); }
      cand[it->second].push_back(v);
    }
    label_slot.resize((size_t)p.pn, -1);
    for (int64_t k = 0; k < p.pn; ++k) {
      auto it = by_label.find(p.pvl[k]);
      if (it != by_label.end()) label_slot[(size_t)k] = it->second;
    }
    map.assign((size_t)p.pn, -1);
    used.assign((size_t)p.gn, 0);
  }
  std::vector<int> label_slot;

  bool edge_ok(int e) const {
    const int64_t mu = map[(size_t)p.ps[e]], mv = map[(size_t)p.pd[e]];
    auto it = slot.find((uint64_t)mu * (uint64_t)p.gn + (uint64_t)mv);
    return it != slot.end() && std::binary_search(out_labels[it->second].begin(), out_labels[it->second].end(), p.pel[e]);
  }

  void rec(int64_t k) {
    if (limit >= 0 && count >= limit) return;
    if (k == p.pn) {
      if (rows && count < cap) std::copy(map.begin(), map.end(), rows + count * p.pn);
      ++count;
      return;
    }
    const int ls = label_slot[(size_t)k];
    if (ls < 0) return;
    for (int64_t v : cand[(size_t)ls]) {
      if (used[(size_t)v]) continue;
      map[(size_t)k] = v;
      bool good = true;
      for (int e : touching[(size_t)k])                         // both endpoints are placed now (ids <= k)
        if (!edge_ok(e)) { good = false; break; }
      if (good) {
        used[(size_t)v] = 1;
        rec(k + 1);
        used[(size_t)v] = 0;
      }
      map[(size_t)k] = -1;
    }
  }
};

bool pair_valid(const Pair &p) {
  if (p.pn < 0 || p.pm < 0 || p.gn < 0 || p.gm < 0) return false;
  if ((p.pn > 0 && !p.pvl) || (p.gn > 0 && !p.gvl) || (p.pm > 0 && (!p.ps || !p.pd || !p.pel)) ||
      (p.gm > 0 && (!p.gs || !p.gd || !p.gel)))
    return false;
  for (int64_t e = 0; e < p.pm; ++e)
    if (p.ps[e] < 0 || p.ps[e] >= p.pn || p.pd[e] < 0 || p.pd[e] >= p.pn) return false;
  for (int64_t e = 0; e < p.gm; ++e)
    if (p.gs[e] < 0 || p.gs[e] >= p.gn || p.gd[e] < 0 || p.gd[e] >= p.gn) return false;
  return true;
}

}  // namespace

extern "C" {

int64_t dmp_subiso_enumerate(int64_t pn, int64_t pm, const int64_t *p_src, const int64_t *p_dst, const int64_t *p_vlabel,
                             const int64_t *p_elabel, int64_t gn, int64_t gm, const int64_t *g_src, const int64_t *g_dst,
                             const int64_t *g_vlabel, const int64_t *g_elabel, int64_t *rows, int64_t rows_capacity,
                             int64_t limit) {
  const Pair p{pn, pm, gn, gm, p_src, p_dst, p_vlabel, p_elabel, g_src, g_dst, g_vlabel, g_elabel};
  if (!pair_valid(p) || rows_capacity < 0 || (rows_capacity > 0 && !rows)) return DMP_ERR_BAD_ARG;
  if (pn > gn) return 0;
  Searcher s(p, rows_capacity > 0 ? rows : nullptr, rows_capacity, limit);
  s.rec(0);
  return s.count;
}

int dmp_subiso_count_batch(int64_t num_pairs, const int64_t *p_node_off, const int64_t *p_edge_off, const int64_t *p_src,
                           const int64_t *p_dst, const int64_t *p_vlabel, const int64_t *p_elabel,
                           const int64_t *g_node_off, const int64_t *g_edge_off, const int64_t *g_src, const int64_t *g_dst,
                           const int64_t *g_vlabel, const int64_t *g_elabel, int64_t *counts, int num_threads) {
  if (num_pairs < 0 || (num_pairs > 0 && (!p_node_off || !p_edge_off || !g_node_off || !g_edge_off || !counts)))
    return DMP_ERR_BAD_ARG;
  std::atomic<int64_t> next{0};
  std::atomic<int> bad{0};
  auto work = [&]() {
    for (int64_t i = next.fetch_add(1); i < num_pairs; i = next.fetch_add(1)) {
      const int64_t pn0 = p_node_off[i], pe0 = p_edge_off[i], gn0 = g_node_off[i], ge0 = g_edge_off[i];
      const int64_t c = dmp_subiso_enumerate(p_node_off[i + 1] - pn0, p_edge_off[i + 1] - pe0, p_src + pe0, p_dst + pe0,
                                            p_vlabel + pn0, p_elabel + pe0, g_node_off[i + 1] - gn0, g_edge_off[i + 1] - ge0,
                                            g_src + ge0, g_dst + ge0, g_vlabel + gn0, g_elabel + ge0, nullptr, 0, -1);
      if (c < 0) bad.store(1);
      counts[i] = c;
    }
  };
  const int nt = num_threads > 0 ? num_threads : (int)std::max(1u, std::thread::hardware_concurrency());
  std::vector<std::thread> pool;
  for (int t = 1; t < nt && t < num_pairs; ++t) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
  return bad.load() ? DMP_ERR_BAD_ARG : DMP_OK;
}

}  // extern "C"
