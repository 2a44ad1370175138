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
