/*
 * ORACLE -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * Plain-C restatement of the reference's integer graph transforms, sequential and in
 * the reference's own loop order.  Pinned against the golden vectors that
 * oracle/make_golden.py emits by running the reference's functions
 * (tests/test_oracle_golden_int.py).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * Paths cited are under /root/reference/SubgraphCountingMatching.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* CSR by key with rows in ascending eid = the order of
 * `incidence_matrix("in")[v]._indices()` (utils/graph.py:104,117,130) and of DGL's
 * in-edge lists; entries are (eid << 1) | flag like the product's. */
void orc_csr_build(const int64_t *key, const uint8_t *flag, int64_t E, int64_t N, int32_t *rowptr,
                   int32_t *ent) {
  memset(rowptr, 0, sizeof(int32_t) * (size_t)(N + 1));
  for (int64_t e = 0; e < E; ++e) rowptr[key[e] + 1]++;
  for (int64_t v = 0; v < N; ++v) rowptr[v + 1] += rowptr[v];
  int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  memcpy(cur, rowptr, sizeof(int32_t) * (size_t)N);
  for (int64_t e = 0; e < E; ++e) ent[cur[key[e]]++] = (int32_t)((e << 1) | (flag ? (flag[e] ? 1 : 0) : 0));
  free(cur);
}

/* add_reversed_edges, GraphAdj branch (train.py:299-327), one graph. */
void orc_add_reversed_edges(const int64_t *src, const int64_t *dst, const int64_t *eid, const int64_t *el,
                            int64_t E, int64_t max_ne, int64_t max_nel, int64_t *o_src, int64_t *o_dst,
                            int64_t *o_eid, int64_t *o_el, uint8_t *o_rev) {
  for (int64_t e = 0; e < E; ++e) {
    o_src[e] = src[e]; o_dst[e] = dst[e]; o_eid[e] = eid[e]; o_el[e] = el[e]; o_rev[e] = 0;
    o_src[E + e] = dst[e]; o_dst[E + e] = src[e];          /* add_edges(v, u, ...)      :307-309 */
    o_eid[E + e] = max_ne + e;                             /* arange(max_nge, +num_ge)  :306     */
    o_el[E + e] = el[e] + max_nel;                         /* label + max_ngel          :312     */
    o_rev[E + e] = 1;                                      /* REVFLAG ones              :313     */
  }
}

/* dgl.batch (dataset.py:1320-1328): node-offset concatenation in list order. */
void orc_collate(const int64_t *ls, const int64_t *ld, const int64_t *nn, const int64_t *ne, int64_t B,
                 int64_t *node_off, int64_t *edge_off, int64_t *src, int64_t *dst, int32_t *edge_graph,
                 int32_t *node_graph) {
  node_off[0] = 0; edge_off[0] = 0;
  for (int64_t g = 0; g < B; ++g) {
    node_off[g + 1] = node_off[g] + nn[g];
    edge_off[g + 1] = edge_off[g] + ne[g];
    for (int64_t e = edge_off[g]; e < edge_off[g + 1]; ++e) {
      src[e] = ls[e] + node_off[g]; dst[e] = ld[e] + node_off[g];
      if (edge_graph) edge_graph[e] = (int32_t)g;
    }
    if (node_graph) for (int64_t v = node_off[g]; v < node_off[g + 1]; ++v) node_graph[v] = (int32_t)g;
  }
}

/* ---- convert_to_dual_graph (utils/graph.py:74-169), one graph ------------------------ */

/* plain branch (:126-134): returns the number of dual edges; arrays sized sum indeg*outdeg */
int64_t orc_line_graph_plain(const int64_t *src, const int64_t *dst, int64_t E, int64_t N, int64_t *dsrc,
                             int64_t *ddst, int64_t *payload) {
  int32_t *ptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
  int32_t *ent = (int32_t *)malloc(sizeof(int32_t) * (size_t)(E > 0 ? E : 1));
  orc_csr_build(dst, NULL, E, N, ptr, ent);
  int64_t m = 0;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t s = src[e];
    for (int32_t k = ptr[s]; k < ptr[s + 1]; ++k) {   /* incident edges of `source`, ascending */
      dsrc[m] = ent[k] >> 1; ddst[m] = e; payload[m] = s; ++m;
    }
  }
  free(ptr); free(ent);
  return m;
}

typedef struct { int64_t a, l, b; } key3;
static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; return x; }

/* id+label branch (:80-95,110-125,161-164).
 * in : src,dst,eid [E], nlabel [N]
 * out: first [K] (K = max eid + 1; -1 for holes), dsrc/ddst in COMPACTED dual-node numbering,
 *      payload; *num_dual_nodes = number of non-hole ids.  Returns the number of dual edges. */
int64_t orc_line_graph_id(const int64_t *src, const int64_t *dst, const int64_t *eid, const int64_t *nlabel,
                          int64_t E, int64_t N, int64_t K, int64_t *first, int64_t *dsrc, int64_t *ddst,
                          int64_t *payload, int64_t *num_dual_nodes) {
  for (int64_t k = 0; k < K; ++k) first[k] = -1;
  for (int64_t e = 0; e < E; ++e)                     /* id2vertex[eid] = min(...)  :83-88 */
    if (first[eid[e]] < 0 || e < first[eid[e]]) first[eid[e]] = e;
  int64_t *newidx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(K > 0 ? K : 1));
  int64_t nk = 0;
  for (int64_t k = 0; k < K; ++k) newidx[k] = first[k] >= 0 ? nk++ : -1;   /* remove_nodes :161-164 */
  *num_dual_nodes = nk;

  int32_t *ptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
  int32_t *ent = (int32_t *)malloc(sizeof(int32_t) * (size_t)(E > 0 ? E : 1));
  orc_csr_build(dst, NULL, E, N, ptr, ent);
  int64_t cap = 16, total = 0;
  for (int64_t e = 0; e < E; ++e) total += ptr[src[e] + 1] - ptr[src[e]];
  while (cap < 2 * total + 2) cap <<= 1;
  key3 *tab = (key3 *)malloc(sizeof(key3) * (size_t)cap);
  uint8_t *used = (uint8_t *)calloc((size_t)cap, 1);
  int64_t m = 0;
  for (int64_t e = 0; e < E; ++e) {
    const int64_t s = src[e], vid = eid[e], lab = nlabel[s];
    for (int32_t k = ptr[s]; k < ptr[s + 1]; ++k) {
      const int64_t uid = eid[ent[k] >> 1];
      uint64_t h = mix((uint64_t)uid * 0x9E3779B97F4A7C15ULL ^ mix((uint64_t)vid + 77) ^ mix((uint64_t)lab * 31 + 5)) &
                   (uint64_t)(cap - 1);
      int seen = 0;
      while (used[h]) {                               /* key in used_keys?          :120-121 */
        if (tab[h].a == uid && tab[h].l == lab && tab[h].b == vid) { seen = 1; break; }
        h = (h + 1) & (uint64_t)(cap - 1);
      }
      if (!seen) {
        used[h] = 1; tab[h].a = uid; tab[h].l = lab; tab[h].b = vid;
        dsrc[m] = newidx[uid]; ddst[m] = newidx[vid]; payload[m] = s; ++m;   /* :122-124 */
      }
    }
  }
  free(tab); free(used); free(ptr); free(ent); free(newidx);
  return m;
}

/* compute_largest_eigenvalues (utils/graph.py:40-71), one graph; degrees from the structure */
void orc_eigen_bounds(const int64_t *src, const int64_t *dst, int64_t E, int64_t N, float *node_eigenv,
                      float *edge_eigenv) {
  int64_t *ind = (int64_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int64_t));
  int64_t *outd = (int64_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(int64_t));
  for (int64_t e = 0; e < E; ++e) { ind[dst[e]]++; outd[src[e]]++; }
  float mn = -1e30f, me = -1e30f;
  for (int64_t e = 0; e < E; ++e) {
    const float a = (float)outd[src[e]] + (float)ind[dst[e]];   /* out_deg[u] + in_deg[v] :49 */
    const float b = (float)ind[src[e]] + (float)outd[dst[e]];   /* in_deg[u] + out_deg[v] :50 */
    if (a > mn) mn = a;
    if (b > me) me = b;
  }
  *node_eigenv = mn; *edge_eigenv = me;
  free(ind); free(outd);
}

/* compute_nodeseq_subisoweights (dataset.py:54-61): how many subisomorphisms use each target node.
 * sub = [rows, width] target node per pattern node. */
void orc_subiso_node_weights(const int64_t *sub, int64_t rows, int64_t width, int64_t num_nodes, int64_t *w) {
  memset(w, 0, sizeof(int64_t) * (size_t)num_nodes);
  for (int64_t r = 0; r < rows; ++r)
    for (int64_t j = 0; j < width; ++j) w[sub[r * width + j]] += 1;
}

static int64_t bisect_left_i64(const int64_t *a, int64_t x, int64_t lo, int64_t hi) { /* dataset.py:22-30 */
  while (lo < hi) {
    int64_t mid = (lo + hi) / 2;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* GraphAdjDataset.calculate_edge_weights (dataset.py:1503-1520) around
 * compute_edgeseq_subisoweights (dataset.py:64-107).  Pattern edges come in eid order, target
 * edges are looked up in (src, dst)-sorted order and the counts are written back by edge id.
 * The pattern's label lists are kept per (u, v) key in a dict that is filled run by run of equal
 * consecutive keys (:79-88): a key that shows up again later REPLACES its earlier run. */
void orc_subiso_edge_weights(const int64_t *p_u, const int64_t *p_v, const int64_t *p_el, int64_t p_len,
                             const int64_t *g_u, const int64_t *g_v, const int64_t *g_el, int64_t g_len,
                             const int64_t *sub, int64_t rows, int64_t width, int64_t *w) {
  memset(w, 0, sizeof(int64_t) * (size_t)g_len);
  if (rows == 0 || p_len == 0 || g_len == 0) return;                    /* counts == 0 -> zeros (:1504) */
  /* all_edges(order="srcdst") (:1509): stable sort of the edge ids by (src, dst) */
  int64_t *ord = (int64_t *)malloc(sizeof(int64_t) * (size_t)g_len);
  for (int64_t i = 0; i < g_len; ++i) ord[i] = i;
  for (int64_t i = 1; i < g_len; ++i) {                                   /* insertion sort: stable */
    int64_t e = ord[i], j = i - 1;
    while (j >= 0 && (g_u[ord[j]] > g_u[e] || (g_u[ord[j]] == g_u[e] && g_v[ord[j]] > g_v[e]))) { ord[j + 1] = ord[j]; --j; }
    ord[j + 1] = e;
  }
  int64_t *su = (int64_t *)malloc(sizeof(int64_t) * (size_t)g_len), *sv = (int64_t *)malloc(sizeof(int64_t) * (size_t)g_len);
  int64_t *sl = (int64_t *)malloc(sizeof(int64_t) * (size_t)g_len), *sw = (int64_t *)calloc((size_t)g_len, sizeof(int64_t));
  for (int64_t i = 0; i < g_len; ++i) { su[i] = g_u[ord[i]]; sv[i] = g_v[ord[i]]; sl[i] = g_el[ord[i]]; }
  int64_t max_v = 0;                                                       /* :77-78 */
  for (int64_t i = 0; i < p_len; ++i) { if (p_u[i] > max_v) max_v = p_u[i]; if (p_v[i] > max_v) max_v = p_v[i]; }
  for (int64_t i = 0; i < g_len; ++i) { if (g_u[i] > max_v) max_v = g_u[i]; if (g_v[i] > max_v) max_v = g_v[i]; }
  int64_t mod = max_v + 1;
  /* dict key -> (start, len) of its LAST run (:79-88) */
  int64_t *dk = (int64_t *)malloc(sizeof(int64_t) * (size_t)p_len), *ds = (int64_t *)malloc(sizeof(int64_t) * (size_t)p_len);
  int64_t *dl = (int64_t *)malloc(sizeof(int64_t) * (size_t)p_len), nd = 0;
  for (int64_t i = 0; i < p_len;) {
    int64_t key = p_u[i] * mod + p_v[i], j = i + 1;
    while (j < p_len && p_u[j] * mod + p_v[j] == key) ++j;
    int64_t slot = nd;
    for (int64_t q = 0; q < nd; ++q) if (dk[q] == key) slot = q;
    dk[slot] = key; ds[slot] = i; dl[slot] = j - i;
    if (slot == nd) ++nd;
    i = j;
  }
  for (int64_t r = 0; r < rows; ++r) {                                   /* :90-106 (key order is immaterial) */
    const int64_t *m = sub + r * width;
    for (int64_t q = 0; q < nd; ++q) {
      int64_t u = m[dk[q] / mod], v = m[dk[q] % mod];
      int64_t u_i = bisect_left_i64(su, u, 0, g_len), u_j = bisect_left_i64(su, u + 1, 0, g_len);
      int64_t v_i = bisect_left_i64(sv, v, u_i, u_j), v_j = bisect_left_i64(sv, v + 1, v_i, u_j);
      for (int64_t k = v_i; k < v_j; ++k)
        for (int64_t t = 0; t < dl[q]; ++t)
          if (p_el[ds[q] + t] == sl[k]) sw[k] += 1;
    }
  }
  for (int64_t i = 0; i < g_len; ++i) w[ord[i]] = sw[i];                 /* edge_weights[g_eid] = ... (:1512) */
  free(ord); free(su); free(sv); free(sl); free(sw); free(dk); free(ds); free(dl);
}
