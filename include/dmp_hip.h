/*
 * dmp_hip.h -- C ABI of libdmp_hip.so: the MI355X (gfx950) dual-message-passing
 * hot path.
 *
 * The reference (HKUST-KnowComp/DualMessagePassing) is pure Python; its "native"
 * boundary for this path is the set of DGL calls made by the DMPNN / CompGCN
 * layers and the dataset transforms.  Each entry point below names the
 * reference call site (path:line under /root/reference) it replaces.  A
 * maintainer binds them with ctypes; the shipped host side
 * (dualmessagepassing_amd/) does exactly that.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless marked "host";
 *  - the caller allocates every buffer, the library holds no state;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - nothing synchronises the stream, nothing allocates; safe to capture in a
 *    hipGraph and to call from several host threads on distinct streams;
 *  - return value: DMP_OK (0) or a negative DMP_ERR_* code; never throws.
 *  - feature matrices are row-major fp32 with an explicit leading dimension
 *    (`ld*`, in floats) so column slices of fused GEMM outputs can be passed
 *    without a copy;
 *  - graph indices are int64 at the API where the reference uses DGL's default
 *    idtype (int64) and int32 inside the CSR index arrays built here.
 *
 * "ent" arrays: a CSR entry is a packed int32  (eid << 1) | flag  where flag is
 * the edge's is_reversed bit (possibly flipped, see dmp_incidence_build).  The
 * rows of every CSR built here list their edges in ascending eid, so every
 * per-destination sum has a fixed order and results are run-to-run bit-stable.
 */
#ifndef DMP_HIP_H
#define DMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMP_OK 0
#define DMP_ERR_BAD_ARG (-1)     /* null pointer, negative size, bad enum          */
#define DMP_ERR_UNSUPPORTED (-2) /* shape the kernels do not cover (e.g. E >= 2^30) */
#define DMP_ERR_HIP (-3)         /* a HIP launch failed; see dmp_last_hip_error()   */

/* ABI version of this header; bumped on any signature change. */
#define DMP_ABI_VERSION 84
int dmp_abi_version(void);
/* hipGetLastError() text of the most recent DMP_ERR_HIP on this host thread. */
const char *dmp_last_hip_error(void);

/* ------------------------------------------------------------------------- */
/* Graph index construction (integer, bit-exact)                             */
/* ------------------------------------------------------------------------- */

/* Number of int32 words of scratch dmp_csr_build / dmp_incidence_build need. */
size_t dmp_csr_workspace_words(int64_t num_nodes, int64_t num_edges);

/*
 * CSR by `key` (key = dst gives the in-edge lists DGL builds behind
 * `update_all(..., fn.sum, ...)`: SubgraphCountingMatching/models/dmpnn.py:163,
 * compgcn.py:271, UnsupervisedNodeClassification/Model/DMPNN/src/model.py:271;
 * key = src gives the out-edge lists).  Also yields the degree vector
 * (`graph.in_degrees()` / `graph.out_degrees()`: dmpnn.py:101,
 * compgcn.py:180,190, dataset.py:1222-1236) as int64.
 *
 *   key      [E] int64  node id per edge, in eid order
 *   flag     [E] uint8  is_reversed per edge, or NULL (all 0)
 *   rowptr   [N+1] int32 out
 *   ent      [E]  int32 out, (eid<<1)|flag, ascending eid inside each row
 *   key32    [E]  int32 out, or NULL: key narrowed to int32
 *   degree   [N]  int64 out, or NULL
 *   status   [1]  int32 out: 0, or 1 if any key was outside [0,N)
 *   ws       scratch, dmp_csr_workspace_words() int32 words
 */
int dmp_csr_build(const int64_t *key, const uint8_t *flag, int64_t num_edges,
                  int64_t num_nodes, int32_t *rowptr, int32_t *ent,
                  int32_t *key32, int64_t *degree, int32_t *status,
                  int32_t *ws, void *stream);

/*
 * Both CSRs of a graph -- by destination (in_*) and by source (out_*) -- built side by side: the same arrays two
 * dmp_csr_build calls produce (rows in ascending edge id, entry = eid << 1 | flag), with the dispatches of a build paid
 * once for the pair.  status: [2] (bit 0 of status[0] / status[1]: a dst / src endpoint outside [0, N)).
 * ws: dmp_csr_pair_workspace_words(N) int32 words.
 */
size_t dmp_csr_pair_workspace_words(int64_t num_nodes);
int dmp_csr_build_pair(const int64_t *dst, const int64_t *src, const uint8_t *flag, int64_t num_edges, int64_t num_nodes,
                       int32_t *in_ptr, int32_t *in_ent, int32_t *dst32, int64_t *in_deg,
                       int32_t *out_ptr, int32_t *out_ent, int32_t *src32, int64_t *out_deg,
                       int32_t *status, int32_t *ws, void *stream);

/* dmp_csr_build_pair for a BLOCK-DIAGONAL batch (what dgl.batch makes, dataset.py:1320-1328: the edges of graph g are the
 * rows [edge_off[g], edge_off[g + 1]) and join nodes [node_off[g], node_off[g + 1]) only) in ONE launch: a workgroup per
 * graph counts, scans, fills and sorts with the graph's counters in LDS.  Every graph has at most
 * dmp_csr_build_graphs_max_nodes() nodes (the caller checks: larger graphs take dmp_csr_build_pair) and node_off[B] == N.
 * Same outputs, bit for bit (rows in ascending edge id).  status [2] is only OR-ed into (bit 0: an endpoint outside its
 * graph's node range -- such an edge is left out); the caller clears it. */
int dmp_csr_build_graphs_max_nodes(void);
int dmp_csr_build_graphs(const int64_t *dst, const int64_t *src, const uint8_t *flag, const int64_t *node_off,
                         const int64_t *edge_off, int64_t B, int64_t E, int64_t N, int32_t *in_ptr, int32_t *in_ent,
                         int32_t *dst32, int64_t *in_deg, int32_t *out_ptr, int32_t *out_ent, int32_t *src32,
                         int64_t *out_deg, int32_t *status, void *stream);

/*
 * Incidence CSR: for node w, its in-edges (flag = is_reversed) and its
 * out-edges (flag = !is_reversed), merged by ascending edge id (a self loop lists
 * its in-entry first).  This is the index of the backward of the
 * implicit line-graph edge message (dmpnn.py:112,120: d/dX of
 * X[dst]W_dst - X[src]W_src with the src/dst swap on reversed edges).
 *
 *   in_ptr/in_ent   CSR by dst from dmp_csr_build
 *   out_ptr/out_ent CSR by src from dmp_csr_build
 *   inc_ptr [N+1] int32 out, inc_ent [2E] int32 out
 */
int dmp_incidence_build(const int32_t *in_ptr, const int32_t *in_ent,
                        const int32_t *out_ptr, const int32_t *out_ent,
                        int64_t num_nodes, int64_t num_edges, int32_t *inc_ptr,
                        int32_t *inc_ent, void *stream);

/*
 * Degree coefficient of the edge update, per node:
 *   coef[v] = 2 * (1 + log2(1 + out_deg[v]))          (dmpnn.py:144-146)
 * out_deg is the int64 vector the reference caches in ndata["out_deg"].
 */
int dmp_degree_coef(const int64_t *out_deg, int64_t num_nodes, float *coef,
                    void *stream);

/*
 * Block-diagonal batching of B graphs (`dgl.batch` behind Graph.batch:
 * SubgraphCountingMatching/dataset.py:1320-1328, called from
 * GraphAdjDataset.batchify dataset.py:1604-1636).
 *
 *   local_src/local_dst [E] int64  per-graph local node ids, graphs
 *                                  concatenated in list order
 *   num_nodes/num_edges [B] int64  per-graph sizes (batch_num_nodes/edges)
 *   node_off/edge_off   [B+1] int64 out, exclusive prefix sums
 *   src/dst             [E] int64 out, global ids (local + node_off[graph])
 *   edge_graph [E] int32 out or NULL, node_graph [N] int32 out or NULL:
 *                                  owning graph of each edge / node
 * E and N (totals) are passed by the host, which knows them from the sizes.
 */
int dmp_collate(const int64_t *local_src, const int64_t *local_dst,
                const int64_t *num_nodes, const int64_t *num_edges,
                int64_t batch, int64_t total_nodes, int64_t total_edges,
                int64_t *node_off, int64_t *edge_off, int64_t *src,
                int64_t *dst, int32_t *edge_graph, int32_t *node_graph,
                void *stream);

/* dmp_collate for several batches in the same two launches (a step collates its pattern batch and its target batch).
 * `jobs` is a HOST array; every job has 1 <= B <= 2048 graphs; edge_graph / node_graph may be NULL. */
typedef struct {
  const int64_t *local_src, *local_dst, *num_nodes, *num_edges; int64_t B, N, E;
  int64_t *node_off, *edge_off, *src, *dst; int32_t *edge_graph, *node_graph;
} dmp_collate_job;
#define DMP_COLLATE_MAX_JOBS 4
int dmp_collate_jobs(const dmp_collate_job *jobs, int n, void *stream);

/*
 * `add_reversed_edges` (GraphAdj branch), SubgraphCountingMatching/train.py:299-327:
 * for one graph, append (v->u) after all (u->v) with id = max_ne + arange(E),
 * label += max_nel, is_reversed = 1 (originals 0).  Batched over B graphs laid
 * out back to back: graph g's E_g edges at edge_off[g] become 2*E_g edges at
 * 2*edge_off[g] ([forward | reversed] per graph, the order the layer sees).
 *
 *   src/dst/eid/elabel [E] int64 in;  edge_off [B+1] int64
 *   o_src/o_dst/o_eid/o_elabel [2E] int64 out; o_rev [2E] uint8 out
 */
int dmp_add_reversed_edges(const int64_t *src, const int64_t *dst,
                           const int64_t *eid, const int64_t *elabel,
                           const int64_t *edge_off, int64_t batch,
                           int64_t num_edges, int64_t max_ne, int64_t max_nel,
                           int64_t *o_src, int64_t *o_dst, int64_t *o_eid,
                           int64_t *o_elabel, uint8_t *o_rev, void *stream);

/*
 * Directed line graph, plain branch of `convert_to_dual_graph`
 * (SubgraphCountingMatching/utils/graph.py:126-134): for e = 0..E-1 in eid
 * order, s = src[e], for every in-edge i of s in ascending eid emit the dual
 * edge (i -> e) with payload s.  Two calls:
 *   count: cnt[e] = in_degree(src[e])           (then the caller scans cnt,
 *          or passes `off` = NULL to dmp_line_graph_fill after dmp_exclusive_scan_i64)
 *   fill : writes dual_src/dual_dst/payload [M], M = sum(cnt) = sum_v indeg*outdeg
 */
int dmp_line_graph_count(const int32_t *in_ptr, const int64_t *src,
                         int64_t num_edges, int64_t *cnt, void *stream);
int dmp_line_graph_fill(const int32_t *in_ptr, const int32_t *in_ent,
                        const int64_t *src, const int64_t *off,
                        int64_t num_edges, int64_t *dual_src, int64_t *dual_dst,
                        int64_t *payload, void *stream);

/*
 * id/label branch of `convert_to_dual_graph` (utils/graph.py:80-95,110-125,161-164):
 * dual edges are built in edge-*id* space and only the first occurrence of a key
 * (id[i], node_label[src], id[e]) is kept; dual nodes are the distinct edge ids,
 * each represented by its first edge, holes removed and ids compacted.
 *
 *   dmp_first_edge_of_id: first[k] = min{e : eid[e] = k} or -1, k in [0,K)
 *   dmp_dedupe_first    : keep[m] = 1 iff candidate m is the first (lowest m)
 *                         with its (a,l,b) key.  table: scratch of
 *                         dmp_dedupe_table_words(M) int64 words.
 */
int dmp_first_edge_of_id(const int64_t *eid, int64_t num_edges, int64_t num_ids,
                         int64_t *first, void *stream);
size_t dmp_dedupe_table_words(int64_t num_items);
int dmp_dedupe_first(const int64_t *key_a, const int64_t *key_l,
                     const int64_t *key_b, int64_t num_items, int64_t *table,
                     uint8_t *keep, void *stream);

/* Subisomorphism weights of a batch (the optional `node_weights` / `edge_weights` of
 * GraphAdjDataset.batchify, SubgraphCountingMatching/dataset.py:1618-1634; counters
 * compute_nodeseq_subisoweights :54-61, compute_edgeseq_subisoweights :64-107,
 * calculate_{node,edge}_weights :1491-1520), whole batch per launch, int64, exact.
 *   sub         : the subisomorphism rows of samples 0..B-1 back to back; a row holds, per pattern
 *                 node, the graph-local target node it maps to.  T = total elements.
 *   sample_ptr  : [B+1] first element of each sample in `sub`.
 *   *_node_off / *_edge_off : [B+1] offsets of the batched pattern (p) / target (g) graph.
 *   dmp_subiso_node_weights : out[g_node_off[i] + x] = #rows of sample i containing x.
 *   dmp_pattern_edge_active : active[j] = 1 iff pattern edge j lies in the LAST run of equal
 *                 consecutive (src, dst) pairs of its graph (the reference's per-key dict keeps
 *                 only that run, dataset.py:79-88).
 *   dmp_subiso_edge_weights : out[e] = #{(row, active pattern edge j) : row maps (src_j, dst_j)
 *                 onto the endpoints of target edge e and label_j = label_e}.
 *                 work_ptr [B+1] = exclusive prefix of rows_i * pattern_edges_i;
 *                 g_out_ptr / g_out_ent: CSR by source of the batched target (dmp_csr_build);
 *                 work_hint: host estimate of work_ptr[B] (grid sizing only; 0 = unknown).
 *   status      : optional device flag, set to 1 when a row entry is outside its target graph. */
int dmp_subiso_node_weights(const int64_t *sub, int64_t T, const int64_t *sample_ptr, int64_t B,
                            const int64_t *g_node_off, int64_t *out, int64_t N,
                            int32_t *status, void *stream);
/*
 * Exact subgraph isomorphisms on the HOST (csrc/dmp_subiso.cpp; host pointers, no GPU, no stream): the `counts` /
 * `subisomorphisms` columns of a dataset in the reference's format (utils/io.py:99-115).  A match: an injective,
 * node-label-preserving map of the pattern's nodes into the graph's such that every pattern edge (u -> v, label l) has a
 * graph edge between the images with label l (non-induced; multigraphs: any parallel edge).
 *   dmp_subiso_enumerate  : one pair -> the number of matches (or DMP_ERR_BAD_ARG); the first rows_capacity of them are
 *                           written to rows [capacity, pn] in lexicographic order of the node map; limit >= 0 stops the
 *                           search after that many matches.
 *   dmp_subiso_count_batch: counts of num_pairs pairs given back to back (node / edge offsets [num_pairs + 1], node ids
 *                           local to each graph), over num_threads host threads (0: all cores).
 */
int64_t dmp_subiso_enumerate(int64_t pn, int64_t pm, const int64_t *p_src, const int64_t *p_dst,
                             const int64_t *p_vlabel, const int64_t *p_elabel, int64_t gn, int64_t gm,
                             const int64_t *g_src, const int64_t *g_dst, const int64_t *g_vlabel,
                             const int64_t *g_elabel, int64_t *rows, int64_t rows_capacity, int64_t limit);
int dmp_subiso_count_batch(int64_t num_pairs, const int64_t *p_node_off, const int64_t *p_edge_off,
                           const int64_t *p_src, const int64_t *p_dst, const int64_t *p_vlabel,
                           const int64_t *p_elabel, const int64_t *g_node_off, const int64_t *g_edge_off,
                           const int64_t *g_src, const int64_t *g_dst, const int64_t *g_vlabel,
                           const int64_t *g_elabel, int64_t *counts, int num_threads);

/*
 * UNC mini-batch samplers on the device (UnsupervisedNodeClassification/Model/DMPNN/src/utils.py:279-349, where DGL's
 * dgl.sampling.random_walk / sample_neighbors do this on the host).  Randomness: a counter-based 32-bit mix of
 * (seed, a, b) (oracle/graph_oracle.py::rng_hash restates it), so results are reproducible from `seed`.
 *   dmp_random_walks   : `walks` walks of `depth` steps from every seed over the CSR by source (dmp_csr_build): step t
 *                        of walk w takes the k-th out-edge of the current node, k = (mix(seed, w, t) * out_degree) >> 32;
 *                        a node without out-edges ends the walk.  traces [num_seeds * walks, depth + 1] int64 (-1 after
 *                        the end) and / or visited [N] uint8 (set to 1 for every node on a walk; the caller zeroes it).
 *   dmp_sample_in_edges: mask[e] = 1 for the in-edges kept for every node v with wanted[v] != 0 (NULL: all nodes): all
 *                        of them if v has at most `width`, else the `width` with the smallest
 *                        (mix(seed, e, 0), e) -- a uniform sample without replacement (any width: a thread per node
 *                        with a sorted list up to 64, a wave per node bisecting the key threshold above; the
 *                        reference's default sample width is 128, main.py:294).  mask [E] uint8 is fully written.
 */
int dmp_random_walks(const int32_t *out_ptr, const int32_t *out_ent, const int32_t *dst, const int64_t *seeds,
                     int64_t num_seeds, int walks, int depth, uint64_t seed, int64_t *traces, uint8_t *visited,
                     void *stream);
int dmp_sample_in_edges(const int32_t *in_ptr, const int32_t *in_ent, const uint8_t *wanted, int64_t N,
                        int64_t E, int width, uint64_t seed, uint8_t *mask, void *stream);

/*
 * Pooling index of a batch (ops.PoolIndex): the rows of graph i are the next sizes[i] rows; every graph's range is cut
 * into chunks of `chunk` rows so that per-graph sums (the prediction heads' Sum / Mean pooling, pred.py:93-156) run as
 * two launches of dmp_seg_sum(2): rows -> chunk sums (CSR vptr / vent over V = R / chunk + B chunk rows, unused tail
 * chunks empty) -> graph sums (CSR gptr / gent).  vent[r] = (r << 1) | flag[r] (flag: is_reversed for edge rows, splits
 * the sum; NULL = 0), seg[r] = the graph of row r (optional).  sizes and flags come as TWO pieces each (the pattern graphs
 * then the target graphs of a union pass: B = Ba + Bb, rows [0, rows_a) read flag_a).  off [B+1] int64, gptr [B+1],
 * vptr [V+1], vent [R], gent [V], seg [R] int32.  Two launches, no host synchronisation.
 */
int dmp_pool_index(const int64_t *sizes_a, int64_t Ba, const int64_t *sizes_b, int64_t Bb,
                   const uint8_t *flag_a, const uint8_t *flag_b, int64_t rows_a, int64_t R, int chunk,
                   int64_t *off, int32_t *gptr, int32_t *vptr, int32_t *vent, int32_t *gent, int32_t *seg,
                   void *stream);

/* Several pooling indexes in the same two launches (a step needs the node index and the edge index of its batch), with
 * three optional outputs per index: flag8 [R] = the flags of both pieces back to back (0 / 1), rowmap [R] = seg[r], or -1
 * for a flagged row (the rows the pooled heads mask out, basemodel.py:1521-1531), sizes [Ba + Bb] = both size pieces back
 * to back.  `jobs` is a HOST array. */
typedef struct {
  const int64_t *sizes_a, *sizes_b; int64_t Ba, Bb;
  const uint8_t *flag_a, *flag_b; int64_t rows_a, R; int chunk;
  int64_t *off; int32_t *gptr, *vptr, *vent, *gent, *seg;
  uint8_t *flag8; int32_t *rowmap; int64_t *sizes;
} dmp_pool_job;
#define DMP_POOL_MAX_JOBS 4
int dmp_pool_index_jobs(const dmp_pool_job *jobs, int n, void *stream);
/* Per-graph sums of a row weight over a pooling index: out [B, 1] = sum_{r in graph} w[r], or with flag8 (dmp_pool_index_jobs)
 * out [B, 2] = [sum over the non-flagged rows | sum over the flagged rows]; w NULL = 1 (row counts).  off [B + 1] = the index's
 * row offsets.  Fixed summation order (a wave per graph, lane-strided, butterfly): run-to-run bit-stable. */
int dmp_pool_weight_sums(const float *w, const uint8_t *flag8, const int64_t *off, int64_t B, float *out, void *stream);


/*
 * get_dual_subisomorphisms (utils/graph.py:277-316 as convert_to_dual_data calls it, train.py:417-446) for a whole
 * batch: the node maps of the samples -> per map and per pattern KEY the (sample-local) id of the graph edge it is sent
 * to.  Keys = maximal runs of consecutive pattern edges with equal (src, dst) in edge-id order; a later run of an
 * existing key replaces its label list and keeps its position (dict semantics, graph.py:293-300).  out: rows back to
 * back, sample i contributing rows_i rows of p_edges_i entries (work_ptr [B+1] = exclusive prefix of rows_i * p_edges_i,
 * work_total = work_ptr[B] as a host int); entry k < #keys: the LARGEST edge id among the graph edges
 * map(src_k) -> map(dst_k) whose label is in key k's list (= the last one of the reference's (src, dst)-sorted scan);
 * no such edge, or k >= #keys: the id of the first edge of that sorted order (the reference leaves index 0 there and
 * maps it through g_eid).  first_sorted [B] receives that id per sample.  workspace: 2 * PE + B int32 words.
 * p_src / p_dst / g_dst are batch-global node ids, the sub rows sample-local ones (as stored in the dataset).
 */
int dmp_dual_subisomorphisms(const int64_t *sub, int64_t T, const int64_t *sample_ptr, const int64_t *work_ptr,
                             int64_t work_total, int64_t B, const int64_t *p_node_off, const int64_t *p_edge_off,
                             const int64_t *p_src, const int64_t *p_dst, const int64_t *p_label, int64_t PE,
                             const int64_t *g_node_off, const int64_t *g_edge_off, const int32_t *g_out_ptr,
                             const int32_t *g_out_ent, const int32_t *g_dst, const int64_t *g_label,
                             int32_t *workspace, int64_t *first_sorted, int64_t *out, int32_t *status, void *stream);
int dmp_pattern_edge_active(const int64_t *p_src, const int64_t *p_dst, const int64_t *p_edge_off,
                            const int32_t *p_edge_graph, int64_t PE, uint8_t *active, void *stream);
int dmp_subiso_edge_weights(const int64_t *sub, int64_t T, const int64_t *sample_ptr,
                            const int64_t *work_ptr, int64_t B, const int64_t *p_node_off,
                            const int64_t *p_edge_off, const int64_t *p_src, const int64_t *p_dst,
                            const int64_t *p_label, const uint8_t *active, const int64_t *g_node_off,
                            const int32_t *g_out_ptr, const int32_t *g_out_ent, const int32_t *g_dst,
                            const int64_t *g_label, int64_t *out, int64_t E, int64_t work_hint,
                            int32_t *status, void *stream);

/*
 * Tile list of the class-typed edge kernels, built on the device (no host sync, no order decided by
 * atomics): edges grouped by the degree `deg[dst]` their coefficient is computed from, classes
 * ascending, nodes of a class in ascending id, the in-edges of a node in ascending edge id; every
 * class starts a new 32-slot tile.
 *   deg [N] int64 (the tensor passed to dmp_degree_coef), in_ptr / in_ent: CSR by destination;
 *   num_classes: size of the class table, 2..65536 (degrees >= num_classes - 1 share -- and poison with NaN --
 *   the last class; status is set to 1 then); tiles_bound >= E / 32 + num_classes.
 *   out: slot_edge [tiles_bound * 32] (-1 = padding), tile_scale [tiles_bound] = dmp_degree_coef of the
 *   tile's class (entries past num_tiles are not written), num_tiles [1];
 *   ws: 8-byte aligned scratch of dmp_class_tiles_workspace_words(N, num_classes) int32 words.
 */
size_t dmp_class_tiles_workspace_words(int64_t num_nodes, int num_classes);
int dmp_class_tiles(const int64_t *deg, const int32_t *in_ptr, const int32_t *in_ent, int64_t num_nodes,
                    int64_t num_edges, int num_classes, int64_t tiles_bound, int32_t *ws,
                    int32_t *slot_edge, float *tile_scale, int32_t *num_tiles, void *stream);
/* ... over the edges a 0 / 1 edge gate keeps (gate [E] floats: an edge with gate 0 gets no slot; row_cnt [N]: scratch, the kept
 * in-edges of every node): the class-typed kernels of a gated layer then walk the kept edges' tiles only -- for the launches whose
 * rows under a zero gate are dead (dmp_edge_fwd_typed's output rows, dmp_atb_typed's products, dmp_bwd_z_typed_arow's gradient
 * rows: dmpnn.py:215-277 multiplies all of them by the gate).  Same order, same tile_scale, tiles_bound as dmp_class_tiles. */
int dmp_class_tiles_gated(const int64_t *deg, const int32_t *in_ptr, const int32_t *in_ent, const float *gate, int32_t *row_cnt,
                          int64_t num_nodes, int64_t num_edges, int num_classes, int64_t tiles_bound, int32_t *ws,
                          int32_t *slot_edge, float *tile_scale, int32_t *num_tiles, void *stream);

/* The CSR by destination restricted to the edges a 0 / 1 edge gate keeps: keep_ptr [N + 1], keep_ent [<= entries of in_ent] (the
 * kept entries of every row in their order, packed (eid << 1 | flag) as in in_ent), row_cnt [dmp_csr_keep_scratch_words(N)] scratch.  A segment sum over it
 * (dmp_seg_sum2: the node aggregation of dmpnn.py:92,163) reads the kept rows only -- for summands that are zeros under a zero
 * gate the same sums (x + 0 = x), without the skipped rows' entries in its instruction stream. */
int64_t dmp_csr_keep_scratch_words(int64_t num_nodes);
int dmp_csr_keep(const int32_t *in_ptr, const int32_t *in_ent, const float *gate, int64_t num_nodes, int64_t num_entries,
                 int32_t *row_cnt, int32_t *keep_ptr, int32_t *keep_ent, void *stream);
/* (num_entries: a host-side hint, entries of in_ent or 0 = unknown: from 12 entries per row on average -- a pooling index's chunk
 * table -- a group of 16 lanes walks a row instead of one thread; same arrays, same bits) */

/* Exclusive prefix sum of int64 counts; out has n+1 entries (out[n] = total).
 * ws: scratch of dmp_scan_workspace_words(n) int64 words. */
size_t dmp_scan_workspace_words(int64_t n);
int dmp_exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out,
                           int64_t *ws, void *stream);

/* ------------------------------------------------------------------------- */
/* Aggregation kernels (fp32, HBM-bound)                                     */
/* ------------------------------------------------------------------------- */

/*
 * Segment sum by destination -- `fn.sum(msg, out)` inside `update_all`
 * (dmpnn.py:92,163; compgcn.py:165,271; UNC model.py:202,271):
 *     out[v, :] = sum_{i in [rowptr[v], rowptr[v+1])}  w_i * M[ent[i]>>1, :]
 * with w_i = edge_w[eid] if edge_w != NULL else 1.  Atomics-free, fixed order.
 *   M [E, ldm>=H], out [N, ldo>=H]
 *   rows_shared: dispatch-order hint, speed only.  1 = consecutive CSR rows are kept on
 *   one XCD (private L2): right when M rows are listed by several destinations (the
 *   incidence CSR) and, measured, also when M was produced by the kernel just before
 *   (59 vs 70 us at E=524288, H=128).  0 = plain dispatch order: ~3 us faster for a cold,
 *   read-once stream.  The shipped host side passes 1.  2 = as 1, said of an incidence CSR (every
 *   row of M listed under two destinations): the same code as a separate kernel instantiation,
 *   so that per-kernel profiler statistics keep the two launch kinds apart.  3 (dmp_seg_sum2 without weights): as 1 under a
 *   third kernel name -- for a measurement launch that a profile must keep apart from a step's own launches (bench.py).
 */
int dmp_seg_sum(const float *M, int64_t ldm, const int32_t *rowptr,
                const int32_t *ent, const float *edge_w, int64_t num_nodes,
                int H, float *out, int64_t ldo, int rows_shared, void *stream);

/*
 * Flag-split segment sum: the DMPLayer node aggregation after moving the
 * W_in / W_out products behind the sum (dmpnn.py:113,121,125 + fn.sum):
 *     out[v, 0:H ] = s0 * sum_{i: flag=0} w_i * M[eid_i, :]
 *     out[v, H:2H] = s1 * sum_{i: flag=1} w_i * M[eid_i, :]
 * With the in-CSR and (s0,s1) = (-1,+1) this is [-S_fwd | S_rev] so that
 * node_agg = out @ [W_in ; W_out].  With the incidence CSR and (+1,-1) it is the
 * backward of dmp_edge_combine w.r.t. the projected node features.
 *   M [E, ldm>=H], out [N, ldo>=2H]
 */
/* dmp_seg_sum2 for a block-diagonal batch whose CSR rows share source rows (the incidence CSR of the layer's backward,
 * where every edge row is summed into both endpoints): the graphs are grouped into tiles -- graphs [0, Ba) in groups of
 * ka, then [Ba, Ba + Bb) in groups of kb; node_off / edge_off [Ba + Bb + 1]: first node / edge row of every graph --
 * and a workgroup stages a 32-column slice of its tile's edge rows in LDS once and sums every node row of the tile from
 * there, in CSR order (the bits of dmp_seg_sum2).  Requirements (else DMP_ERR_UNSUPPORTED / wrong results): every CSR
 * entry of a tile's node rows refers to an edge row of the same tile, no tile has more than 512 edge rows, H % 32 == 0. */
int dmp_seg_sum2_tiled(const float *M, int64_t ldm, const int32_t *rowptr, const int32_t *ent,
                       const int64_t *node_off, const int64_t *edge_off, int64_t Ba, int64_t Bb, int ka, int kb,
                       int H, float s0, float s1, float *out, int64_t ldo, void *stream);
int dmp_seg_sum2(const float *M, int64_t ldm, const int32_t *rowptr,
                 const int32_t *ent, const float *edge_w, int64_t num_nodes,
                 int H, float s0, float s1, float *out, int64_t ldo,
                 int rows_shared, void *stream);
/* ... for the destination rows of a LIST only (rowlist [<= num_nodes] ascending row ids, *rowcount of them in device memory:
 * dmp_kept_rows of a 0 / 1 NODE gate): the other rows of `out` are not written.  A node under a zero of the ScalarFilter's
 * node gate is a zero row in every layer (dmpnn.py:245-277), its aggregate feeds only its own (gated) update: dead. */
int dmp_seg_sum2_rows(const float *M, int64_t ldm, const int32_t *rowptr, const int32_t *ent, const int32_t *rowlist,
                      const int32_t *rowcount, int ptr_by_pos, int incidence, int64_t num_nodes, int H, float s0, float s1, float *out,
                      int64_t ldo, void *stream);
/* The incidence CSR (dmp_incidence_build: a node's in-entries and out-entries, flag flipped, merged by ascending edge id) over the
 * edges a 0 / 1 edge gate keeps (gate [E] floats), for the nodes of a list only (list / count: dmp_kept_rows of a 0 / 1 node gate):
 * keep_ptr [*count + 1] is indexed by the POSITION in the list, keep_ent [<= 2 kept edges]; row_cnt: dmp_csr_keep_scratch_words(N)
 * scratch.  out_ptr / out_ent NULL: the in-entries alone -- the kept edges' CSR by destination with a row per list position (the
 * forward node aggregation: the row group of a kept node finds its entries without the node's row pointers).  dmp_seg_sum2_rows(ptr_by_pos = 1) over it is the backward of the gathered node projections (dmpnn.py:111-127) under
 * the ScalarFilter's gates: every kept node's two sums over its kept edges in ascending edge id -- the bits of dmp_seg_sum2 over
 * the full incidence CSR (the left-out addends are zero rows, the left-out nodes' rows dead) from the fewest row reads: an edge
 * row with no kept endpoint is never fetched. */
int dmp_incidence_keep(const int32_t *in_ptr, const int32_t *in_ent, const int32_t *out_ptr, const int32_t *out_ent,
                       const float *gate, const int32_t *list, const int32_t *count, int64_t num_nodes, int32_t *row_cnt,
                       int32_t *keep_ptr, int32_t *keep_ent, void *stream);

/* Gate compaction of a block-diagonal batch (csrc/dmp_compact.hip): the edges a filter gate keeps (gate[e] != 0), graph by
 * graph in ascending eid, as a batch of exactly `cap` edges -- `cap - kept` padding edges (gate 0, self-loops dealt over
 * the graphs and their nodes) follow each graph's kept edges.  An edge with gate 0 is a zero row through the whole rep-net
 * of the reference (basemodel.py:1515-1531, dmpnn.py:215-277: embeddings and every layer update are multiplied by the
 * gate); only the degrees see it, so `out_deg` [N] (optional) receives the out-degrees of the WHOLE graph (dmpnn.py:101).
 * Outputs: src_c / dst_c / eid_map (the edge's eid; 0 for padding) [cap] int64, rev_c [cap] (optional), gate_c [cap],
 * num_edges_c [B], edge_off_c [B + 1]; kept [B] int32 workspace = kept edges per graph; status[0] is OR-ed with: bit 0 = more
 * kept edges than cap (outputs truncated: run the batch as it stands), bit 1 = padding fell on a graph without nodes (the
 * caller clears the word; it is never reset here, so one word can watch over many recorded steps).
 * zero_deg: clear out_deg first (needed when a graph has more than dmp_gate_compact_hist_nodes() nodes).  Two launches. */
int dmp_gate_compact_hist_nodes(void);
int dmp_gate_compact(const float *gate, const int64_t *src, const int64_t *dst, const uint8_t *rev, const int64_t *node_off,
                     const int64_t *edge_off, int64_t B, int64_t N, int64_t E, int64_t cap, int zero_deg, int32_t *kept,
                     int64_t *out_deg, int64_t *src_c, int64_t *dst_c, uint8_t *rev_c, int64_t *eid_map, float *gate_c,
                     int64_t *num_edges_c, int64_t *edge_off_c, int32_t *status, void *stream);
/* Out-degrees of an edge list (graph.out_degrees(), dataset.py:1230-1236) without building its CSR: deg[u] = #{e: src[e] == u}. */
int dmp_out_degrees(const int64_t *src, int64_t E, int64_t N, int64_t *deg, void *stream);

/*
 * The same sums as dmp_seg_sum2 over the incidence CSR -- the backward of the gathered node projections of
 * DMPLayer._node_message_func (dmpnn.py:111-127: dP_d[a_e] += dPre[e], dP_s[b_e] -= dPre[e]) -- from ONE pass over the
 * edge rows (csrc/dmp_segacc.hip):
 *     out[v, 0:H ] = s0 * sum_{e: sel_a[e] = v} M[e, :]        out[v, H:2H] = s1 * sum_{e: sel_b[e] = v} M[e, :]
 * every sum in ascending e (the bits of dmp_seg_sum2 over dmp_incidence_build's CSR when sel_a / sel_b are
 * dmp_edge_select_build's selectors).  For a block-diagonal batch: graphs [0, Ba) are grouped ka to a tile, graphs
 * [Ba, Ba + Bb) kb to a tile; node_off / edge_off [Ba + Bb + 1]: first node / edge row of every graph.  A workgroup
 * streams its tile's edge rows once and keeps the tile's node sums in registers (a wave: 64 columns x 64 nodes x 2 halves).  Requirements: no tile has more than dmp_seg_sum2_graphs_max_nodes() nodes (further nodes of a tile are left
 * unwritten), every edge of a tile has both endpoints among the tile's nodes (others are dropped), H is 64 or 128
 * (DMP_ERR_UNSUPPORTED otherwise: the caller falls back to dmp_seg_sum2).
 *   M [E, ldm>=H], sel_a / sel_b [E] int32 (node ids), out [N, ldo>=2H]
 */
int dmp_seg_sum2_graphs_max_nodes(void);
int dmp_seg_sum2_graphs(const float *M, int64_t ldm, const int32_t *sel_a, const int32_t *sel_b,
                        const int64_t *node_off, const int64_t *edge_off, int64_t Ba, int64_t Bb, int ka, int kb,
                        int H, float s0, float s1, float *out, int64_t ldo, void *stream);
/* ... where rows of M are known to be all zeros: rowmask [ceil(num_rows / 32)] (dmp_row_mask_bits: bit e & 31 of word e >> 5
 * clear = row e is zero, e.g. the dPre rows a 0 / 1 edge gate wiped in dmp_bwd_h1_fused_rows) -- those rows are not fetched.
 * nodemask (optional) [ceil(num_nodes / 32)]: bit v & 31 of word v >> 5 clear = row v of `out` is dead (a node under a zero of
 * a 0 / 1 node gate: its gradient is multiplied by that zero further down, dmpnn.py:245-277) and is NOT STORED -- `out` keeps
 * whatever it held there; with dmp_edge_select_nodes' selectors (-1 for such a node) its addends are dropped too. */
int dmp_seg_sum2_graphs_masked(const float *M, int64_t ldm, const int32_t *sel_a, const int32_t *sel_b,
                               const int64_t *node_off, const int64_t *edge_off, int64_t Ba, int64_t Bb, int ka, int kb,
                               int H, float s0, float s1, float *out, int64_t ldo, const uint32_t *rowmask, int64_t num_rows,
                               const uint32_t *nodemask, int64_t num_nodes, void *stream);

/*
 * Row gather by an int32 index -- `edges.src[k]` / `edges.dst[k]` inside the
 * message UDFs (dmpnn.py:112,120; compgcn.py:227) and the backward of
 * dmp_seg_sum:   out[e, :] = w_e * X[idx[e], :]
 */
int dmp_gather_rows(const float *X, int64_t ldx, const int32_t *idx,
                    const float *edge_w, int64_t num_edges, int H, float *out,
                    int64_t ldo, void *stream);

/*
 * Backward of dmp_seg_sum2 over the in-CSR, optionally accumulating onto `base`:
 *     out[e, :] = base[e, :] + w_e * (flag[e] ? s1 * D[dst[e], H:2H] : s0 * D[dst[e], 0:H])
 *   D [N, ldd>=2H], out [E, ldo>=H], flag may be NULL (all 0), base [E, ldb>=H] or NULL (0)
 */
int dmp_gather_select(const float *D, int64_t ldd, const int32_t *dst,
                      const uint8_t *flag, const float *edge_w,
                      const float *base, int64_t ldb, int64_t num_edges, int H,
                      float s0, float s1, float *out, int64_t ldo, void *stream);

/*
 * DMPLayer edge pre-activation (dmpnn.py:112,120,124,142-151), project-then-gather:
 *     Y[e] = G[e,0:H] + coef[dst e] * G[e,H:2H] + bias
 *            + (flag[e] ? P[src e,0:H] - P[dst e,H:2H]
 *                       : P[dst e,0:H] - P[src e,H:2H])
 * where G = Z @ [W_eloop | W_src - W_dst]  and  P = X @ [W_dst | W_src].
 *   G [E, ldg>=2H], P [N, ldp>=2H], coef [N], bias [H] or NULL, Y [E, ldy>=H]
 * relu != 0 applies the MLP's activation max(y, slope * y) to Y (slope 0: ReLU; slope 1/5.5: the
 * reference's default `leaky_relu`, utils/act.py:27,466; 0 <= slope <= 1, else DMP_ERR_UNSUPPORTED):
 * used when the first Linear of the edge MLP (no non-linearity sits between it and this sum,
 * dmpnn.py:147-152) has been folded into G and P by multiplying the weights, so Y is already
 * the MLP's hidden activation.
 */
int dmp_edge_combine(const float *G, int64_t ldg, const float *P, int64_t ldp,
                     const float *coef, const float *bias, const int32_t *src,
                     const int32_t *dst, const uint8_t *flag,
                     int64_t num_edges, int H, int relu, float slope, float *Y, int64_t ldy,
                     void *stream);

/*
 * Backward of dmp_edge_combine w.r.t. G:
 *     dG[e, 0:H] = dY[e],   dG[e, H:2H] = coef[dst e] * dY[e]
 * (w.r.t. P it is dmp_seg_sum2 over the incidence CSR with (s0,s1) = (+1,-1);
 *  w.r.t. bias a column sum the host side takes.)
 */
int dmp_edge_combine_bwd_g(const float *dY, int64_t ldy, const float *coef,
                           const int32_t *dst, int64_t num_edges, int H,
                           float *dG, int64_t ldg, void *stream);

/*
 * CompGCN message aggregation (compgcn.py:213-238 + fn.sum), sum moved in front
 * of the W_in / W_out products:
 *     m_e = norm_e * comp(X[src e], Z[e]),  comp = sub (0): x - z, mult (1): x * z,
 *           cmul (2): conj(x) * z over rows of interleaved (re, im) pairs (H = 2 x bins, H % 4 == 0) -- the reference's
 *           default composition "corr" (config.py:170-172; compgcn.py:218-222: irfft(conj(rfft h) rfft r)) in the frequency
 *           domain: the DFT is linear, so the transform of the rows is a product with a fixed matrix on either side of the
 *           sum (X D, Z D before; D^-1 [W_in; W_out] folded into the N-row product after) and no per-edge FFT exists
 *     out[v, 0:H ] = sum_{e->v, flag=0} m_e ;  out[v, H:2H] = sum_{e->v, flag=1} m_e
 * so that node_agg = out @ [W_in ; W_out].  norm may be NULL.
 */
int dmp_compgcn_agg(const float *X, int64_t ldx, const float *Z, int64_t ldz,
                    const int32_t *rowptr, const int32_t *ent,
                    const int32_t *src, const float *norm, int64_t num_nodes,
                    int H, int comp, float *out, int64_t ldo, void *stream);

/*
 * Backward of dmp_compgcn_agg w.r.t. Z (per edge, streaming):
 *     g_e = norm_e * D[dst e, flag ? H:2H : 0:H]
 *     dZ[e] = sub: -g_e          mult: g_e * X[src e]          cmul: g_e * X[src e]  (complex product)
 * and the per-edge term of dX:  dXe[e] = sub: g_e   mult: g_e * Z[e]   cmul: conj(g_e) * Z[e]
 * (dX = dmp_seg_sum of dXe over the CSR by src).
 */
int dmp_compgcn_agg_bwd(const float *D, int64_t ldd, const float *X,
                        int64_t ldx, const float *Z, int64_t ldz,
                        const int32_t *src, const int32_t *dst,
                        const uint8_t *flag, const float *norm,
                        int64_t num_edges, int H, int comp, float *dZ,
                        int64_t lddz, float *dXe, int64_t lddxe, void *stream);

/* ------------------------------------------------------------------------- */
/* Row-wise epilogues of the fused layer (fp32, HBM-bound, H % 4 == 0)       */
/* ------------------------------------------------------------------------- */

/* Number of partial rows the *_colsum kernels write for `rows` input rows (<= 1024). */
int64_t dmp_colsum_partial_rows(int64_t rows, int H);

/*
 * Gate + residual of the rep-net loop (dmpnn.py:263-273: `v = v * v_gate`,
 * `v_outputs[-1] + v`):   out[r] = prev[r] + gate[r] * upd[r]
 * prev may be NULL (no residual), gate may be NULL (pattern side: no gate).
 */
int dmp_gate_residual(const float *prev, int64_t ldp, const float *upd, int64_t ldu,
                      const float *gate, int64_t rows, int H, float *out, int64_t ldo,
                      void *stream);

/* First MLP Linear of the node update after the fold (dmpnn.py:129-140):  out = act(a + b + bias),
 * act(y) = max(y, slope * y) (slope 0: ReLU), in one pass -- a: the projected aggregate S Bn, b: the
 * self-loop projection (a column slice of XP), bias [H] or NULL.  out may alias a. */
int dmp_add_bias_relu(const float *a, int64_t lda, const float *b, int64_t ldb, const float *bias,
                      int64_t rows, int H, float slope, float *out, int64_t ldo, void *stream);

/*
 * Backward of the gate and the bias gradient of the layer before it:
 *     dUpd[r] = gate[r] * dOut[r];  partial[b, :] = column sums of dUpd over workgroup b's rows
 * gate NULL: dUpd is dOut itself (pass dUpd = NULL), only the column sums are produced.
 * partial: [dmp_colsum_partial_rows(rows, H), H]; finish with dmp_reduce_partials.
 */
int dmp_scale_rows_colsum(const float *dOut, int64_t ldd, const float *gate, int64_t rows,
                          int H, float *dUpd, int64_t ldu, float *partial, void *stream);

/* (Leaky)ReLU backward on the saved OUTPUT (`threshold_backward` / `leaky_relu_backward(self_is_result)`:
 * dPre = act > 0 ? dH : slope * dH; the MLPs of dmpnn.py:45-60) + column sums of dPre (gradient of the
 * preceding Linear's bias).  dPre may alias dH. */
int dmp_relu_bwd_colsum(const float *dH, int64_t ldh, const float *act, int64_t lda,
                        int64_t rows, int H, float slope, float *dPre, int64_t ldp, float *partial,
                        void *stream);

/*
 * The same with the upstream gradient given through a row map: dH[r] = gate[r] * table[rowmap[r]] (rowmap[r] < 0: a zero
 * row; gate NULL: 1).  The backward of a layer whose output feeds ONLY per-graph sum / mean pooling heads
 * (basemodel.py:1545-1631 -> pred.py:176-214): the gradient of every row of a graph is the same [H] vector, so
 * d(out) W2 is a [graphs, H] product and the [rows, H] gradient tensor never exists (dmpnn.py:142-156 backward).
 */
int dmp_relu_bwd_gathered_colsum(const float *table, int64_t ldt, const int32_t *rowmap, const float *gate, const float *act,
                                 int64_t lda, int64_t R, int H, float slope, float *dPre, int64_t ldp, float *partial, void *stream);

/*
 * The same pass ALSO producing the gated per-chunk sums of the saved activation over a pooling index (ops.PoolIndex: vptr
 * [num_chunks + 1], vent [R] = (row << 1) | flag, chunks of <= 64 rows of one graph): chunk_sums [num_chunks, 2H] =
 * [sum_{flag = 0} gate[r] act[r] | sum_{flag = 1} ...] -- the operand of dW2 = T^T Q for a layer whose output gradient is a
 * per-graph vector -- so that the activation is read once for both.  partial: [dmp_pool_relu_bwd_blocks(...), H].
 */
int64_t dmp_pool_relu_bwd_blocks(int64_t num_chunks, int H);
int dmp_pool_relu_bwd(const float *act, int64_t lda, const float *table, int64_t ldt, const int32_t *rowmap, const float *gate,
                      const int32_t *vptr, const int32_t *vent, int64_t num_chunks, int64_t R, int H, float slope, float *dPre,
                      int64_t ldp, float *chunk_sums, float *partial, void *stream);

/* dmp_edge_combine_bwd_g + column sums of dY (gradient of ebias, dmpnn.py:148-149). */
int dmp_edge_combine_bwd_g_colsum(const float *dY, int64_t ldy, const float *coef,
                                  const int32_t *dst, int64_t num_edges, int H, float *dG,
                                  int64_t ldg, float *partial, void *stream);

/* (Leaky)ReLU backward fused with dmp_edge_combine_bwd_g (for the folded first Linear):
 *     dPre = act > 0 ? dH : slope * dH;  dG = [dPre | coef[dst] * dPre];  partial = column sums of dPre */
int dmp_relu_bwd_g_colsum(const float *dH, int64_t ldh, const float *act, int64_t lda,
                          const float *coef, const int32_t *dst, int64_t num_edges, int H, float slope,
                          float *dG, int64_t ldg, float *partial, void *stream);

/* Column-sum partials of A [rows, H]. */
int dmp_colsum_partials(const float *A, int64_t lda, int64_t rows, int H, float *partial,
                        void *stream);

/* Weight gradient of a narrow input layer with a row gate fused in (the label-embedding products of
 * embed.py:103-120 feeding `e = e * e_gate`, basemodel.py:1515):
 *     partial[b][k, :] = sum over workgroup b's rows r of X[r, k] * gate[r] * D[r, :]
 * X [rows, ldx >= K] (K <= 16 inputs per row: the multihot label encodings), D [rows, ldd >= H] the upstream
 * gradient of gate * (X W), gate [rows] or NULL; H = 128 or 64.
 * partial: [dmp_smallk_atb_blocks(rows), K*H]; finish with dmp_reduce_partials. */
/* Forward of the same layer with its gate:  out[r, :] = gate[r] * (X[r, :K] W)  with W [K, ldw >= H]
 * (embed.py:103-120 + basemodel.py:1515), written where the caller wants the gated rows (ldo >= H). */
int dmp_smallk_embed_gate(const float *X, int64_t ldx, int K, const float *W, int64_t ldw,
                          const float *gate, int64_t rows, int H, float *out, int64_t ldo, void *stream);
/* ... leaving out DEAD rows: a row whose gate is 0 is NOT stored (`out` keeps whatever the buffer held there) -- for a 0 / 1 gate
 * all of whose readers skip the rows under its zeros (the first layer's residual term under the kept edges' tiles). */
int dmp_smallk_embed_live(const float *X, int64_t ldx, int K, const float *W, int64_t ldw, const float *gate, int64_t R,
                          int H, float *out, int64_t ldo, void *stream);
/* The same two over several column blocks in ONE launch: W / out [.., ncols * H] (ncols <= 8 blocks of H columns), and
 * D [rows, ncols * H] plus an optional further block D2 [rows, H] of another matrix (NULL: none).
 * partial: [ncols (+1), dmp_smallk_atb_blocks(rows), K*H] -- one dmp_reduce_partials per column block. */
int dmp_smallk_embed_cols(const float *X, int64_t ldx, int K, const float *W, int64_t ldw, const float *gate,
                          int64_t rows, int H, int ncols, float *out, int64_t ldo, void *stream);
/* ... with a row mask (dmp_row_mask_bits): rows of D / D2 whose bit is 0 -- rows whose X row is all zeros, or whose gate is 0 -- are
 * not fetched. */
int dmp_smallk_atb_cols_masked(const float *X, int64_t ldx, int K, const float *D, int64_t ldd, int ncols, const float *D2,
                               int64_t ldd2, const float *gate, const uint32_t *rowmask, int64_t R, int H, float *partial,
                               void *stream);
/* ... over a LIST of rows (dmp_kept_rows: ascending row ids, their number in device memory; R = the list's capacity): a batch is
 * then four LIVE rows -- the masked form spends a batch's slots on its dead rows too (three of five at the benchmark's node gate). */
int dmp_smallk_atb_cols_rows(const float *X, int64_t ldx, int K, const float *D, int64_t ldd, int ncols, const float *D2,
                             int64_t ldd2, const int32_t *list, const int32_t *count, int64_t R, int H, float *partial, void *stream);
/* Several of these products -- same K and H, one column block each, with a gate and / or a row mask or neither -- in ONE launch
 * (num_jobs <= 4).  The first layer's node side on the label codes needs S0_h^T dPn for both halves h of the code sums and both
 * embedding tables (dmpnn.py:92,163 re-associated: DESIGN 3): four launches of 5-12 us as one.
 * job.partial: [dmp_smallk_atb_blocks(job.R), K*H] (an empty job: one zero block); finish each with dmp_reduce_partials. */
typedef struct {
  const float *X; int64_t ldx;           /* [R, ldx >= K] */
  const float *D; int64_t ldd;           /* [R, ldd >= H] */
  const float *gate;                     /* [R] or NULL */
  const uint32_t *rowmask;               /* dmp_row_mask_bits words (bit 0 = the job's first row) or NULL */
  int64_t R;
  float *partial;
} dmp_smallk_job;
int dmp_smallk_atb_jobs(const dmp_smallk_job *jobs, int num_jobs, int K, int H, void *stream);
int64_t dmp_smallk_atb_blocks(int64_t rows);
int dmp_smallk_atb(const float *X, int64_t ldx, int K, const float *D, int64_t ldd, const float *gate,
                   int64_t rows, int H, float *partial, void *stream);

/* The FIRST layer of a rep-net whose edge input is a label embedding z0 = enc W (basemodel.py:1393-1420 feeding
 * dmpnn.py:111-156): every product of the layer with z0 has rank K <= 16, so it is computed from the label codes
 * (csrc/dmp_layer0.hip).  H = 128 or 64.
 *   dmp_l0_pack      out[r, 0:Kpad] = r < rows_p ? enc_p[r, 0:K] : gate[r - rows_p] * enc_g[r - rows_p, 0:K], zero-padded
 *                    (the union's edge rows: pattern rows first, then the gated target rows; gate may be NULL).  The target
 *                    rows' codes start at column goff: with goff = K the two kinds of rows occupy disjoint columns, so two
 *                    embedding tables stacked to [2K, H] act as one (Kpad >= goff + K; Kpad % 4 == 0 and `out` 16-byte aligned:
 *                    the rows are stored in 16-byte pieces -- DMP_ERR_UNSUPPORTED otherwise).
 *   dmp_l0_edge_fwd_masked  out[r] = act(enc[r] MA + coef_e[r] (enc[r] MB) + P[sel_a[r], 0:H] - P[sel_b[r], H:2H] + bias)
 *                    with M = [MA | MB] = W [A | B]  ([K, ldm >= 2H]); replaces dmp_edge_fwd_typed for this layer.
 *   dmp_l0_bwd_w_masked     partial[b] = [enc^T dPre | (coef_e enc)^T dPre | enc^T dZn] over workgroup b's rows
 *                    ([dmp_l0_bwd_w_blocks(rows), K, (dZn ? 3 : 2) * H]; finish with dmp_reduce_partials): the
 *                    class-typed weight gradient is W^T of the first two blocks, the embedding's gradient needs all three. */
int dmp_l0_pack(const float *enc_p, int64_t ldp, int64_t rows_p, const float *enc_g, int64_t ldg, const float *gate,
                int64_t rows_g, int K, int Kpad, int goff, float *out, void *stream);
/* ... one or two of them in ONE launch (a step packs its edge rows' codes and its node rows' codes). */
typedef struct {
  const float *enc_p; int64_t ldp; int64_t rows_p;
  const float *enc_g; int64_t ldg; const float *gate; int64_t rows_g;
  int K, Kpad, goff;
  float *out;
} dmp_l0_pack_job;
int dmp_l0_pack_jobs(const dmp_l0_pack_job *jobs, int num_jobs, void *stream);
/* ... leaving out DEAD rows: bit r of rowmask[t] == 0 says that every consumer of row 32 t + r of `out` multiplies it by a zero gate
 * and does not fetch it (the `_masked` kernels, dmp_pool_relu_bwd, the weighted dmp_seg_sum): nothing is gathered, computed or
 * STORED for such a row -- `out` keeps whatever the buffer held there. */
int dmp_l0_edge_fwd_masked(const float *enc, int64_t lde, int K, const float *M, int64_t ldm, const float *P, int64_t ldp,
                           const float *bias, const float *coef_e, const int32_t *sel_a, const int32_t *sel_b,
                           const uint32_t *rowmask, int64_t R, int H, float slope, float *out, int64_t ldo, void *stream);
/* The first layer's NODE side from the label codes, one pass (dmpnn.py:113,121,125,129-140 with node rows x = venc WV0 and edge
 * rows z0 = enc W0, so that a node's aggregates are S0_h W0 with S0_h the sums of its in / out edges' codes):
 *   h1[r]          = act([venc[r] | S0[r, 0:K0] | S0[r, Kp:Kp+K0]] W[:, 0:H] + bias)
 *   P[r, 0:2H]     = [venc[r] | ...] W[:, H:3H]                                                (the gathered projections, ldp >= 2H)
 * for the node rows n0 <= r < n1 of one embedding table.  W [dmp_l0_node_pack_rows(), ldw >= 3H], packed by the caller: rows
 * 0 .. VK-1 = WV0 Wx, rows VK .. VK+K0-1 = W0 Bn_in and the next K0 rows W0 Bn_out in the first H columns, every other entry ZERO
 * (VK + 2 K0 <= dmp_l0_node_pack_rows()).  rowmask (bit r & 31 of word r >> 5, absolute node ids; NULL: every row): a DEAD node's h1
 * row is left unwritten (its consumers walk the kept nodes' tiles), its P row is written as zeros (its code row is zero; the first
 * layer's edge kernel gathers P through the plain selectors).  list / count (optional, with rowmask; dmp_kept_rows over ALL nodes,
 * list_bound >= *count known to the host): the launch walks the list's positions from q_begin on -- sixteen live rows per wave and
 * batch -- and acts on the entries inside [n0, n1); q_begin <= n0: a position the caller knows to precede the table's first row
 * (0 is always valid; n0 when every node below n0 is kept). */
int64_t dmp_l0_node_pack_rows(void);
int dmp_l0_node_fwd(const float *venc, int64_t ldv, int VK, const float *S0, int64_t lds, int K0, int Kp, const float *W, int64_t ldw,
                    const float *bias, float slope, const uint32_t *rowmask, const int32_t *list, const int32_t *count, int64_t list_bound,
                    int64_t q_begin, int64_t n0, int64_t n1, int H, float *h1, int64_t ldh, float *P, int64_t ldp, void *stream);
int64_t dmp_l0_bwd_w_blocks(int64_t rows);
/* dmp_l0_bwd_w_masked with a row mask: bit r of rowmask[t] == 0 says the code row 32 t + r is all zeros (the row's gate was 0 when
 * dmp_l0_pack made it), so its dPre / dZn rows -- which would be multiplied by those zeros -- are not fetched. */
int dmp_l0_bwd_w_masked(const float *enc, int64_t lde, int K, const float *coef_e, const float *dPre, int64_t ldd, const float *dZn,
                        int64_t ldz, const uint32_t *rowmask, int64_t R, int H, float *partial, void *stream);
/* ... over a LIST of rows instead of the rows 0 .. R-1 (the kept rows of a 0 / 1 gate, dmp_kept_rows: ascending row ids, their number
 * in device memory): a batch of rows is then kRows kept rows -- the masked forms spend a batch's slots on its dead rows too.
 *   dmp_kept_rows: list [<= R] = the rows r < R whose bit (r & 31) of rowmask[r >> 5] is set, *count = how many;
 *   scratch: dmp_kept_rows_scratch_words(R) int32 words.
 *   tiles != 0: the list is also a slot list of 32-row tiles for the tile kernels (dmp_out_fwd_typed, dmp_bwd_h1_typed: the
 *   node side of a layer over the nodes a 0 / 1 node gate keeps): list [(R + 31) / 32 * 32], the entries from *count to the next
 *   multiple of 32 are -1 (padding slots), count [2] with count[1] = the number of tiles. */
int64_t dmp_kept_rows_scratch_words(int64_t R);
int dmp_kept_rows(const uint32_t *rowmask, int64_t R, int tiles, int32_t *scratch, int32_t *list, int32_t *count, void *stream);
/* Several lists in ONE pair of launches (a step derives three from its two 0 / 1 gates: the kept nodes' tiles, the first layer's
 * kept target edges, the kept edges' tiles in ascending order), and -- selA != NULL -- dmp_edge_select_nodes' three arrays from the
 * count launch (they need the node mask only, as the lists do): what were seven launches of ~5 us in the step's index chain.
 * Every job as dmp_kept_rows (R > 0); same bits. */
#define DMP_KEPT_MAX_JOBS 4
typedef struct {
  const uint32_t *mask; int64_t R; int tiles;
  int32_t *scratch;                    /* dmp_kept_rows_scratch_words(R) words */
  int32_t *list, *count;
} dmp_kept_job;
int dmp_kept_rows_jobs(const dmp_kept_job *jobs, int n, const int32_t *src, const int32_t *dst, const uint8_t *flag,
                       const uint32_t *nodemask, int64_t E, int32_t *selA, int32_t *selB, int32_t *dstM, void *stream);
/* dmp_row_mask_bits for several gates in one launch. */
#define DMP_ROWMASK_MAX_JOBS 4
typedef struct { const float *gate; int64_t R; uint32_t *mask; } dmp_rowmask_job;
int dmp_row_mask_bits_jobs(const dmp_rowmask_job *jobs, int n, void *stream);
int dmp_l0_edge_fwd_rows(const float *enc, int64_t lde, int K, const float *M, int64_t ldm, const float *P, int64_t ldp,
                         const float *bias, const float *coef_e, const int32_t *sel_a, const int32_t *sel_b,
                         const int32_t *list, const int32_t *count, int64_t R, int H, float slope, float *out, int64_t ldo,
                         void *stream);
int dmp_l0_bwd_w_rows(const float *enc, int64_t lde, int K, const float *coef_e, const float *dPre, int64_t ldd, const float *dZn,
                      int64_t ldz, const int32_t *list, const int32_t *count, int64_t R, int H, float *partial, void *stream);

/* BatchNorm1d in TRAINING mode over the rows of x [rows, C] with the activation that follows it fused in (the UNC layers'
 * MLPs: Linear -> BatchNorm1d -> LeakyReLU -> Linear, UNC model.py:145-157; torch.nn.BatchNorm1d semantics: biased variance
 * for the normalisation, running_var from the unbiased one, `momentum` update; running_* may be NULL).  csrc/dmp_bn.hip.
 *   fwd: out = act(gamma (x - mean) / sqrt(var + eps) + beta)  (act != 0: LeakyReLU(slope), slope 0 = ReLU);
 *        stats[0:C] = mean, stats[C:2C] = 1 / sqrt(var + eps) (kept for the backward).
 *   bwd: dx = gamma invstd (dyb - sum(dyb) / rows - xhat sum(dyb xhat) / rows), dyb = act'(y) dy (y: the saved output);
 *        stats[2C:3C] = dbeta, stats[3C:4C] = dgamma.
 * partial: [dmp_bn_partial_rows(rows, C), 2C] scratch; stats: [4C].  C % 4 == 0, 256 % (C / 4) == 0, C <= 1024 (else
 * DMP_ERR_UNSUPPORTED: 0 from dmp_bn_partial_rows); all matrices 16-byte aligned with row strides % 4 == 0. */
int64_t dmp_bn_partial_rows(int64_t rows, int C);
int dmp_bn_train_fwd(const float *x, int64_t ldx, int64_t rows, int C, const float *gamma, const float *beta, float eps,
                     float momentum, float *running_mean, float *running_var, int act, float slope, float *partial,
                     float *stats, float *out, int64_t ldo, void *stream);
int dmp_bn_train_bwd(const float *x, int64_t ldx, const float *y, int64_t ldy, const float *dy, int64_t lddy, int64_t rows,
                     int C, const float *gamma, int act, float slope, float *partial, float *stats, float *dx, int64_t ldo,
                     void *stream);
/* ... with the row count ON THE DEVICE (rows_dev [1], int64, or NULL): only the first min(rows, *rows_dev) rows exist -- a batch padded
 * to a fixed capacity so that its step replays from a recording (UNC's sampled sub-graphs, utils.py:279-349, differ in size from
 * step to step).  The others stay out of the statistics (and out of `rows` in the divisions) and are written as zeros. */
int dmp_bn_train_fwd_rows(const float *x, int64_t ldx, int64_t rows, const int64_t *rows_dev, int C, const float *gamma, const float *beta,
                          float eps, float momentum, float *running_mean, float *running_var, int act, float slope, float *partial,
                          float *stats, float *out, int64_t ldo, void *stream);
int dmp_bn_train_bwd_rows(const float *x, int64_t ldx, const float *y, int64_t ldy, const float *dy, int64_t lddy, int64_t rows,
                          const int64_t *rows_dev, int C, const float *gamma, int act, float slope, float *partial, float *stats, float *dx,
                          int64_t ldo, void *stream);

/* out[l] (+)= sum_s partial[s, l], s in a fixed order; L % 4 == 0.  Also reduces the
 * split-K partial products of the weight gradients. */
int dmp_reduce_partials(const float *partial, int64_t S, int64_t L, float *out,
                        int accumulate, void *stream);

/* n <= DMP_REDUCE_MAX_SEGMENTS independent reductions in one launch (the partials of the parameter
 * gradients of one layer's backward): outs[i][l] = sum_s partials[i][s, l], each in the order of
 * dmp_reduce_partials.  The four arrays are HOST arrays of n entries (device pointers / sizes). */
#define DMP_REDUCE_MAX_SEGMENTS 16
int dmp_reduce_partials_multi(const float *const *partials, const int64_t *S, const int64_t *L,
                              float *const *outs, int n, void *stream);

/*
 * One AdamW step over a flat fp32 buffer (the optimizer of SubgraphCountingMatching/train.py:1231:
 * AdamW(lr, weight_decay, amsgrad=True); torch.optim.AdamW's update, element by element):
 *     p *= 1 - lr * weight_decay;  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g^2;
 *     [vmax = max(vmax, v)];  p -= lr / (1 - beta1^step) * m / (sqrt(vmax or v) / sqrt(1 - beta2^step) + eps)
 * max_exp_avg_sq NULL: no AMSGrad.  step counts from 1.  All buffers [n], 16-byte aligned.
 */
int dmp_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                   float *max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int64_t step, void *stream);
/* The same step, leaving the ranges [skip_lo[k], skip_hi[k]) of the buffer untouched (nskip <= DMP_ADAMW_MAX_SKIP host
 * entries, multiples of 4 floats): the slices of parameters that received no gradient this step, which
 * torch.optim.AdamW skips (no decay, no moment update). */
#define DMP_ADAMW_MAX_SKIP 16
int dmp_adamw_step_skip(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                        float *max_exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int64_t step, const int64_t *skip_lo, const int64_t *skip_hi,
                        int nskip, void *stream);

/* The same update with the step count and the learning rate in DEVICE memory: state = [step, lr] (two doubles).  The
 * call first adds 1 to state[0] (a one-thread launch ordered before the update), then evaluates the bias corrections
 * and lr * weight_decay from the device values -- nothing that changes from step to step is left in the launch
 * arguments, so a training step captured in a HIP graph (hipGraphLaunch / torch.cuda.CUDAGraph.replay) stays
 * correct when replayed; the host writes state[1] before a replay when a scheduler changed the learning rate. */
int dmp_adamw_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                       float *max_exp_avg_sq, int64_t n, double *state, double beta1, double beta2,
                       double eps, double weight_decay, const int64_t *skip_lo, const int64_t *skip_hi,
                       int nskip, void *stream);

/* The AdamW step over a flat buffer that holds P parameter tensors back to back, each with its OWN step count, as
 * torch.optim.AdamW keeps one per tensor (train.py:1231): a tensor that receives its first gradient at optimizer step k starts its
 * bias corrections at 1, a tensor without a gradient this step is left alone and its count stands.  seg_off [P + 1] (device):
 * element offsets of the tensors (multiples of 4); live (HOST, ceil(P / 64) words, or NULL = all): bit s = tensor s has a gradient
 * this step; live_dev (DEVICE floats [P], or NULL): when given it replaces `live` -- tensor s is live where live_dev[s] != 0
 * (data-parallel runs: every rank adds its own 0 / 1 indicators to the gradient all-reduce, so the set is the union over the
 * ranks and the replicas take the same step, dp.FlatGradSync); state (device doubles) [2 + P]: [0] optimizer steps taken,
 * [1] learning rate (written by the host), [2 + s] the step count of tensor s; seg_tab (device floats) [2 P]: scratch.
 * veto / veto_mask as dmp_adamw_step_guarded (NULL: none).
 * Two launches, nothing per step in the arguments except the live set: replays from a HIP graph. */
#define DMP_ADAMW_MAX_SEGMENTS 1024
int dmp_adamw_step_segments(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                            int64_t n, double *state, const int64_t *seg_off, int P, const uint64_t *live, const float *live_dev,
                            float *seg_tab, double beta1, double beta2, double eps, double weight_decay, int32_t *veto,
                            int32_t veto_mask, void *stream);

/* dmp_adamw_step_dev that DROPS the step when a device-side flag says so (as a loss-scaling optimizer drops a step whose
 * gradients overflowed): veto = int32 [4] in device memory.  veto[0] collects flags raised since the last optimizer step
 * (dmp_gate_compact ORs bit 0 into it for a batch that kept more edges than its capacity -- its gradients are wrong);
 * if veto[0] & veto_mask the step count is not advanced and parameters and moments are left as they are.  Either way
 * veto[1] |= veto[0], veto[0] = 0, veto[2] = this step was dropped, veto[3] += dropped.  No host sync: replays. */
int dmp_adamw_step_guarded(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *max_exp_avg_sq,
                           int64_t n, double *state, double beta1, double beta2, double eps, double weight_decay,
                           const int64_t *skip_lo, const int64_t *skip_hi, int nskip, int32_t *veto, int32_t veto_mask,
                           void *stream);

/*
 * ScalarFilter gates of a batch of (pattern, target) pairs (filter.py:6-16 on the pre-padded label matrices,
 * basemodel.py:1394-1423) for several element kinds (node labels, edge labels) in three dispatches:
 *   gate[j] = 1.0 if the label of target row j occurs among the labels of the pattern of the same pair, else 0.0;
 *   label 0 also passes when that pattern is shorter than p_max (its pre-padding zeros take part in the
 *   reference's comparison).
 * p_seg / g_seg: pair id of every pattern / target row (int64, or -- seg_is_i32 != 0 -- the int32 arrays the
 * device collate leaves in node_graph / edge_graph), labels in [0, num_labels) (rows with labels outside
 * that range never match and mark nothing).  p_sizes [B] rows per pattern, or NULL when every pattern has
 * p_max rows.  present: scratch of present_bytes >= sum over jobs of B * num_labels bytes, job i using
 * [present_off, present_off + B * num_labels).  The jobs array is a HOST array.
 */
#define DMP_FILTER_MAX_JOBS 4
typedef struct {
  const void *p_seg; const int64_t *p_label; int64_t num_p;
  const int64_t *p_sizes; int64_t p_max;
  const void *g_seg; const int64_t *g_label; int64_t num_g;
  int64_t num_labels, present_off;
  float *gate;                      /* [num_g] */
  int64_t seg_is_i32;
} dmp_filter_job;
int dmp_scalar_filter_gates(const dmp_filter_job *jobs, int num_jobs, int64_t B, uint8_t *present,
                            int64_t present_bytes, void *stream);

/*
 * Pre-padding masks of a batch and their row counts (utils/dl.py:113-127 batch_convert_len_to_mask(pre_pad=True);
 * basemodel.py:1521-1531: reversed edges leave the edge masks) for several element kinds in one launch:
 *   mask[b, j]  = j >= max_len - sizes[b]  and not rev[off[b] + j - (max_len - sizes[b])]      (uint8 0/1 = torch.bool)
 *   count[b]    = number of set entries of row b (fp32)
 * sizes [B] rows of every graph; off [B] first row of every graph (NULL: b * max_len, i.e. all graphs have
 * max_len rows); rev: one byte per row or NULL.  The jobs array is a HOST array.
 */
#define DMP_MASK_MAX_JOBS 4
typedef struct {
  const int64_t *sizes, *off; int64_t max_len;
  const uint8_t *rev;
  uint8_t *mask;                    /* [B, max_len] */
  float *count;                     /* [B] */
} dmp_mask_job;
int dmp_len_masks(const dmp_mask_job *jobs, int num_jobs, int64_t B, void *stream);

/*
 * Row lookups into small frozen tables (the multi-hot / position encodings of ids and labels, embed.py:199-224:
 * enc = table[ids]) for several (table, ids) pairs in one launch:
 *   out[r, 0:width] = table[idx[r], 0:width]        r < rows
 * idx values outside [0, table_rows) give NaN rows.  The jobs array is a HOST array.
 */
#define DMP_LOOKUP_MAX_JOBS 8
typedef struct {
  const float *table; int64_t ld, table_rows; int width;
  const int64_t *idx; int64_t rows;
  float *out;                       /* [rows, width], dense */
} dmp_lookup_job;
int dmp_table_rows(const dmp_lookup_job *jobs, int num_jobs, void *stream);

/*
 * Several two-way concatenations in one launch (the block-diagonal union of the pattern and target batches,
 * dgl.batch([a, b]) on the structure arrays: edge endpoints of b shifted by a's node count; the gate vectors of the
 * union pass: ones for the pattern rows):
 *   out[0:na] = a (or fill_a when a is NULL, 4-byte elements only),   out[na:na+nb] = b (+ add_b, 8-byte = int64 only)
 * elem_size in {1, 4, 8}.  The jobs array is a HOST array.
 */
#define DMP_CONCAT_MAX_JOBS 12
typedef struct {
  const void *a; int64_t na;
  const void *b; int64_t nb;
  void *out;
  int elem_size;
  int64_t add_b;
  float fill_a;
} dmp_concat_job;
int dmp_concat_pairs(const dmp_concat_job *jobs, int num_jobs, void *stream);

/*
 * Pack n separate fp32 arrays into one flat buffer in one launch (train.py:1231's optimizer and the gradient
 * all-reduce work on ONE flat gradient; autograd hands back one tensor per parameter):
 *   dst[dst_off[i] : dst_off[i] + len[i]] = src[i][0 : len[i]]      i < n
 * src / dst_off / len are HOST arrays; every dst_off[i] a multiple of 4 and dst 16-byte aligned; the sources
 * need no alignment.  src[i] == NULL: the segment is CLEARED (a parameter that received no gradient this step: torch leaves
 * its .grad None; the flat buffer holds zeros there) -- in the same launch, instead of clearing the whole buffer first.
 * pad_to_4 != 0: the up-to-3 floats between a segment's end and the next multiple of 4 are cleared as well (for a buffer whose
 * segments are laid out in 16-byte pieces, padding included: dp.FlatGradSync).  Any n (split over launches of DMP_PACK_MAX_SEGMENTS).
 */
#define DMP_PACK_MAX_SEGMENTS 128
int dmp_pack_segments(const float *const *src, const int64_t *dst_off, const int64_t *len, int n, int pad_to_4, float *dst,
                      void *stream);

/*
 * Parameter algebra of the fused layer for all layers of a rep-net in one launch (H = 128 or 64).
 * The first Linear of the node / edge MLP (dmpnn.py:45-60,129-156) is folded into the projections that
 * feed it:  [W_loop; W_in; W_out; nbias] W0n^T  and  [W_eloop; W_src - W_dst; W_dst; W_src; ebias] W0e^T,
 * written in the layouts the layer kernels read:
 *   Bn [2H,H] = [W_in; W_out] W0n^T,   bn [H] = nbias W0n^T + nb0,
 *   Wx [H,3H] = [W_loop W0n^T | W_dst W0e^T | W_src W0e^T],
 *   Wes [H,2H] = [W_eloop W0e^T | (W_src - W_dst) W0e^T],   be [H] = ebias W0e^T + eb0.
 * All weights [H,H] row-major as the reference stores them (dmpnn.py:97-109; W0: nn.Linear [out,in]),
 * biases [H]; every pointer 16-byte aligned.  The struct arrays are HOST arrays of num_layers entries.
 * dmp_unfold_layers is the backward: from the gradients of the folded tensors to those of the weights
 * (the gradients of nb0 / eb0 are dbn / dbe themselves and are not written).
 */
#define DMP_FOLD_MAX_LAYERS 3   /* layers per launch; more are split over launches */
typedef struct {
  const float *nloop_w, *in_w, *out_w, *nbias, *eloop_w, *src_w, *dst_w, *ebias, *nW0, *nb0, *eW0, *eb0;
  const float *nW2, *eW2, *eye;   /* optional (forward only): the second Linears [H,H] and an [H,H] identity matrix */
} dmp_layer_weights;
typedef struct {
  float *Bn, *bn, *Wx, *Wes, *be;
  float *WesT, *nW2t, *eW2t;      /* optional outputs (NULL to skip): [A'^T | B'^T] [H,2H] and the transposed second Linears
                                     [in,out] -- the layouts dmp_bwd_z_typed_arow / dmp_out_fwd_fused_rows read coalesced */
} dmp_layer_folded;
typedef struct { const float *dBn, *dbn, *dWx, *dWes, *dbe; } dmp_layer_folded_grads;
typedef struct { float *nloop_w, *in_w, *out_w, *nbias, *eloop_w, *src_w, *dst_w, *ebias, *nW0, *eW0; } dmp_layer_weight_grads;
int dmp_fold_layers(const dmp_layer_weights *w, const dmp_layer_folded *f, int num_layers, int H,
                    void *stream);
int dmp_unfold_layers(const dmp_layer_weights *w, const dmp_layer_folded_grads *g,
                      const dmp_layer_weight_grads *d, int num_layers, int H, void *stream);

/*
 * Several SMALL dense products in one launch (csrc/dmp_fold.hip): the parameter-only algebra of the first layer on
 * the label codes (fused.Layer0Codes: W0 [A | B], W0 Bn, WV0 Wx forward; their gradients backward -- a dozen products
 * with 10..20 rows or columns, each a library call of 5..13 us before).  Job j:
 *     C_j [M, N] = sum_{t < num_terms} op(A_t) [M, K_t]  op(B_t) [K_t, N]  (+ C0_j)
 * op(X) = X or X^T (trans flag: the operand is stored transposed, i.e. A as [K, M] / B as [N, K]); leading dimensions
 * in floats; fp32 FMA chains in k order, terms in order, C0 added last.  At most DMP_GEMM_MAX_JOBS jobs of at most
 * DMP_GEMM_MAX_TERMS terms; C must not alias an operand of the same launch.
 */
#define DMP_GEMM_MAX_JOBS 12
#define DMP_GEMM_MAX_TERMS 3
typedef struct {
  const float *A; int64_t lda; const float *B; int64_t ldb; int32_t transA, transB, K, pad;
} dmp_gemm_term;
typedef struct {
  dmp_gemm_term term[DMP_GEMM_MAX_TERMS];
  const float *C0; int64_t ldc0; float *C; int64_t ldc; int32_t num_terms, M, N, pad;
} dmp_gemm_job;
int dmp_small_gemm_jobs(const dmp_gemm_job *jobs, int num_jobs, void *stream);


/*
 * The pooled prediction heads (SubgraphCountingMatching/models/pred.py:93-156 applied to per-graph sums; the node
 * and the edge head of basemodel.py:1477-1498) for all heads in one launch forward and two backward:
 *     p = ps Wp^T + scale_p bp;  g = gs Wg^T + scale_g bg;  s = [pl, gl, 1/pl, 1/gl]
 *     f = [p | g | g - p | g * p | s];  y1 = act(f W1^T + b1);  y = [y1 | s] W2^T + b2
 * act(x) = max(x, slope x): ReLU for slope 0, the reference's default pred_act_func leaky_relu for 1/5.5.
 * H = width of ps / gs and of the hidden layer = 128 or 64.  Weights in nn.Linear layout: Wp, Wg [H,H], W1 [H,4H+4],
 * W2 [1,H+4]; ps / gs [B, ld >= H]; pl / gl [B] (mask counts); scale_p / scale_g: the factor on the bias (the padded
 * length for sum pooling).  Forward writes y [B] and keeps F [B,4H+4] = f and Y1S [B,H+4] = [y1 | s] for backward.
 * Backward: dy [B] (times dy_scale [B] if given: the blend weight of this head) -> dps / dgs [B, ld] (may be NULL),
 * all weight / bias gradients (fully written), through the scratch dY1, dP, dG [B,H].
 * The struct arrays are HOST arrays of num_heads <= DMP_HEADS_MAX entries.
 */
#define DMP_HEADS_MAX 2
typedef struct { const float *Wp, *bp, *Wg, *bg, *W1, *b1, *W2, *b2; } dmp_head_weights;
typedef struct {
  const float *ps; int64_t ld_ps; const float *gs; int64_t ld_gs; const float *pl, *gl; float scale_p, scale_g;
  float *F, *Y1S, *y;
} dmp_head_io;
typedef struct {
  const float *dy, *dy_scale; float *dY1, *dP, *dG; float *dps; int64_t ld_dps; float *dgs; int64_t ld_dgs;
  float *dWp, *dbp, *dWg, *dbg, *dW1, *db1, *dW2, *db2;
} dmp_head_grads;
int dmp_heads_forward(const dmp_head_weights *w, const dmp_head_io *io, int num_heads, int B, int H,
                      float slope, void *stream);

/* The count loss of a training step AND its seed in one launch (train.py:624-628: `bp_crit(F.leaky_relu(pred_c, neg_slp), counts)`,
 * reduction 'mean', followed by `loss.backward()` -- five torch launches around ~1000 numbers):
 *     loss[0] = mean_i crit(leaky_relu(pred[i], neg_slope) - target[i]),     dpred[i] = d loss[0] / d pred[i]
 * kind 0 = MSE, 1 = MAE (L1), 2 = smooth L1 with beta 1 (the reference's "SMSE"); neg_slope in [0, 1] (1: no activation).
 * One workgroup, fixed summation order (bit-stable); n <= 2^22, else DMP_ERR_UNSUPPORTED (the caller uses its tensor ops). */
int dmp_count_loss(const float *pred, const float *target, int64_t n, int kind, float neg_slope, float *loss, float *dpred, void *stream);
int dmp_heads_backward(const dmp_head_weights *w, const dmp_head_io *io, const dmp_head_grads *g,
                       int num_heads, int B, int H, float slope, void *stream);

/*
 * Blend of the heads' counts by the sizes of their target graphs (basemodel.py:1488-1494):
 *   w_i[b] = gl_i[b] / sum_k gl_k[b],   out[b] = sum_i y_i[b] * w_i[b]   (terms added in head order)
 * y / gl / w are HOST arrays of num_heads device pointers ([B] each); the weights w_i are written for the backward
 * (dmp_head_grads.dy_scale).
 */
int dmp_heads_blend(const float *const *y, const float *const *gl, float *const *w, int num_heads, int B, float *out,
                    void *stream);


/* ------------------------------------------------------------------------- */
/* Fused MFMA kernels of the edge chain (fp32 MFMA, exact fp32).  H = 128; the one-panel kernels (dmp_out_fwd_fused_rows,
 * dmp_bwd_h1_fused_rows without coefE), the class-typed kernels (dmp_edge_fwd_typed, dmp_bwd_z_typed_arow, dmp_atb_typed) and the
 * row weight-gradient kernels (dmp_atb_rows_masked, dmp_atb_rows_jobs_h) also H = 64, the reference's shipped hidden_dim
 * (config.py:298-301): DMP_ERR_UNSUPPORTED for any other width. */
/* ------------------------------------------------------------------------- */

/*
 * Per-edge selectors of the edge chain, built once per batched graph (they depend only on the
 * structure and the degrees; dmpnn.py:111-124,144-146):
 *     selA[e] = flag[e] ? src[e] : dst[e]     (node whose W_dst-side projection is added)
 *     selB[e] = flag[e] ? dst[e] : src[e]     (node whose W_src-side projection is subtracted)
 *     coefE[e] = coef[dst[e]]                 (coef = dmp_degree_coef)
 */
int dmp_edge_select_build(const int32_t *src, const int32_t *dst, const uint8_t *flag,
                          const float *coef, int64_t num_edges, int32_t *selA, int32_t *selB,
                          float *coefE, void *stream);
/* ... for a rep-net whose NODE rows under a zero of a 0 / 1 node gate are zeros in every layer (the ScalarFilter gate
 * multiplies the input rows and every layer's update, dmpnn.py:245-277): selA / selB / dstM [E] as above (dstM = dst) with
 * every such node replaced by -1, an index outside any descriptor -- the tile kernels' gathers of its (never written)
 * projection / gradient rows return zeros without touching memory, the one-pass endpoint sums drop its addends.
 *   nodemask [ceil(N / 32)]: bit v & 31 of word v >> 5 set = node v is kept (dmp_row_mask_bits of the node gate). */
int dmp_edge_select_nodes(const int32_t *src, const int32_t *dst, const uint8_t *flag, const uint32_t *nodemask,
                          int64_t num_edges, int32_t *selA, int32_t *selB, int32_t *dstM, void *stream);

/*
 * The E-row projection of the layer and dmp_edge_combine(relu) in one pass
 * (dmpnn.py:112,120,124,142-152 with the first Linear of emlp folded in, see fused.py):
 *     H1[e] = act( Z[e] W[:, 0:H] + coefE[e] * Z[e] W[:, H:2H] + b + P[selA e, 0:H] - P[selB e, H:2H] )
 *   act(y) = max(y, slope * y), 0 <= slope <= 1 (0: ReLU).  Z [E, ldz>=H], W [H, ldw>=2H] row-major ([in, out] layout), P [num_nodes, ldp>=2H], H1 [E, ldh>=H].
 * The [E,2H] product never reaches HBM.  Returns DMP_ERR_UNSUPPORTED unless H == 128, or when
 * num_nodes*ldp*4 or E*4 do not fit 32 bits (the kernels address tables with 32-bit byte offsets).
 */
int dmp_edge_fwd_fused(const float *Z, int64_t ldz, const float *W, int64_t ldw,
                       const float *P, int64_t ldp, int64_t num_nodes, const float *bias,
                       const int32_t *selA, const int32_t *selB, const float *coefE,
                       int64_t num_edges, int H, float slope, float *H1, int64_t ldh, void *stream);

/*
 * Second Linear of the MLP + gate + residual in one pass (dmpnn.py:136,152 + 263-273):
 *     out[r] = R[r] + gate[r] * (Hin[r] W2^T + b2)
 *   W2 [H, ldw>=H] in nn.Linear layout [out, in], or -- w_in_out != 0 -- its transpose [in, out] (every
 *   workgroup then reads its weight panel with coalesced loads); gate [rows] or NULL (1); R [rows, ldr] or NULL (0).
 */

/*
 * Backward of the second Linear, the (Leaky)ReLU and dmp_edge_combine in one pass:
 *     dPre[e] = H1[e] > 0 ? dO[e] W2 : slope * (dO[e] W2) ;   dG[e] = [dPre[e] | coefE[e] * dPre[e]]
 *     partial = column sums of dPre per (workgroup, wave group): dmp_mfma_partial_rows_h(E) rows of H floats
 *   dO [E, ldo>=H] (already gated), W2 [H, ldw>=H] in nn.Linear layout, H1 [E, ldh>=H], dG [E, ldg>=2H].
 *   coefE NULL: only dPre is written (dG [E, ldg>=H]: may be a column slice of a wider matrix), and then
 *   `gate` [E] (or NULL) may carry a row gate: dO is the UNgated output gradient and
 *   dPre[e] = H1[e] > 0 ? gate[e] (dO[e] W2) : 0 -- the separate gate pass of the backward is fused away
 *   (gate must be NULL with coefE).  Also serves the node update's MLP (rows = nodes, gate = v_gate).
 */
            /* H = 128 */
int64_t dmp_mfma_partial_rows_h(int64_t num_edges, int H);   /* H = 128 or 64 */

/* Row masks for gated E-row kernels: bit r of mask[t] = (gate[32 t + r] != 0), mask [(E + 31) / 32] words.  The `_masked` forms of
 * dmp_out_fwd_fused_rows / dmp_bwd_h1_fused_rows take it beside the gate and do not FETCH the operand rows of a masked row (H1 in the
 * forward kernel; dO and H1 in the backward kernel): those rows' products are multiplied by their zero gate
 * (out = R + 0 (h1 W2^T + b2) = R, dPre = act'(.) (0 dO W2) = 0), so the results are the unmasked kernels' -- with 60 % of the rows gated
 * out (a ScalarFilter target batch, basemodel.py:1515-1531) a fifth / two fifths of the kernels' bytes are not moved. */
int dmp_row_mask_bits(const float *gate, int64_t E, uint32_t *mask, void *stream);
/* ... of the rows of a matrix: bit r of mask[t] = (row 32 t + r of X [R, ldx] has a non-zero among its first K entries) -- the label
 * code rows that dmp_l0_pack multiplied by a zero gate; for dmp_l0_bwd_w_masked / dmp_smallk_atb_cols_masked. */
int dmp_row_mask_rows(const float *X, int64_t ldx, int K, int64_t R, uint32_t *mask, void *stream);
/* ... that also hands out the column sums of the rows of dO it FETCHED: partial_rows [dmp_mfma_partial_rows_h(E, H), H] (or NULL), summed by
 * dmp_reduce_partials.  With the row mask of a 0 / 1 gate that is sum_e gate_e dO[e] -- the bias gradient of the Linear behind the gate
 * (dmpnn.py:45-60: db2), for which dmp_atb_rows_masked otherwise carries column sums: the weight gradient can then run ungated
 * (dmp_atb_rows_plain).  Without a mask: the column sums of all of dO. */

/* ... where the caller knows more about the masked-out rows (the rep-net of a ScalarFilter batch: the union's edge rows are
 * [pattern rows | gate * target rows], and zn = z + gate (...) keeps a masked-out row of every layer's input at zero):
 *   dmp_out_fwd_fused_rows, dead_rows bit 0: R's masked-out rows are zeros -- not fetched (the output rows are zeros);
 *                           dead_rows bit 1: the masked-out rows of `out` are not STORED (every reader of `out` leaves them out:
 *                           the next layer's masked kernels) -- `out` keeps whatever it held there;
 *   dmp_bwd_h1_fused_rows, skip_dead_stores: the masked-out rows of dG (zeros) are not stored (its readers -- the class-tile
 *                           kernels with masked slots, dmp_seg_sum2_graphs_masked, dmp_l0_bwd_w_masked -- leave them out).
 * Both need the row mask. */
int dmp_out_fwd_fused_rows(const float *Hin, int64_t ldh, const float *W2, int64_t ldw, const float *bias,
                           const float *gate, const uint32_t *rowmask, int dead_rows, const float *R, int64_t ldr, int64_t E, int H,
                           int w_in_out, float *out, int64_t ldo, void *stream);
int dmp_bwd_h1_fused_rows(const float *dO, int64_t ldo, const float *W2, int64_t ldw, const float *H1, int64_t ldh,
                          const float *coefE, const float *gate, const uint32_t *rowmask, int skip_dead_stores, int64_t E, int H,
                          float slope, float *dG, int64_t ldg, float *partial, float *partial_rows, void *stream);

/*
 * ... and over the tiles of the KEPT edges only (dmp_class_tiles_gated's slot_edge / tile_scale / num_tiles / tiles_bound; the rows
 * are gathered and scattered by edge id as in the class-typed kernels, csrc/dmp_typed.hip), for a 0 / 1 gate whose kept rows
 * have gate 1:
 *   dmp_out_fwd_typed   out[e] = R[e] + R2[e] + (Hin[e] W2^T + bias)   for the kept e; the other rows of `out` are not written
 *                       (W2: nn.Linear's [out, in], or w_in_out: its transpose [in, out]; R, R2 may be NULL; R may be `out`: a row
 *                       is read before it is written; act != 0: out[e] = LeakyReLU_slope(...)).  num_jobs (<= 6) such products over
 *                       the SAME tile list run as ONE launch (grid.y = the job; every workgroup loads one weight panel) -- the
 *                       node side of a layer over the tiles of the nodes a 0 / 1 node gate keeps, dmp_kept_rows(tiles = 1):
 *                       x W_x, act(x W_0 + S B_n + b), x + H1 W2^T + b2, dP_n B_n^T, dxn + dXP W_x^T of dmpnn.py:113,121,129-140
 *                       as 128-wide block products
 *   dmp_bwd_h1_typed    dG[e] = act'(H1[e]) (.) (dO[e] W2)      for the kept e; the other rows of dG are not written;
 *                       partial / partial_rows [dmp_typed_partial_rows(tiles_bound, H), H]: column sums of dG / of the fetched
 *                       rows of dO per workgroup (partial_rows may be NULL), summed by dmp_reduce_partials
 * -- the work of dmp_out_fwd_fused_rows(dead_rows = 3) / dmp_bwd_h1_fused_rows(skip_dead_stores) without the tiles' worth of
 * zero rows in between the kept ones.
 */
int64_t dmp_typed_partial_rows(int64_t tiles_bound, int H);
typedef struct {
  const float *Hin; int64_t ldh;       /* streamed rows [E, ldh >= H] */
  const float *W2; int64_t ldw; int w_in_out;
  const float *bias;                   /* [H] or NULL */
  const float *R; int64_t ldr;         /* addend rows or NULL */
  const float *R2; int64_t ldr2;       /* second addend rows or NULL */
  int act; float slope;
  float *out; int64_t ldo;
} dmp_typed_job;
int dmp_out_fwd_typed(const dmp_typed_job *jobs, int num_jobs, const int32_t *slot_edge, const float *tile_scale,
                      const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H, void *stream);
int dmp_bwd_h1_typed(const float *dO, int64_t ldo, const float *W2, int64_t ldw, const float *H1, int64_t ldh,
                     const int32_t *slot_edge, const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, int64_t E,
                     int H, float slope, float *dG, int64_t ldg, float *partial, float *partial_rows, void *stream);
/* dmp_out_fwd_typed (one job, no second addend) with a K-EXTENSION of the product:
 *     out[e] = (e < r_rows ? R[e] : 0) + (Hin[e] W2^T + (e >= code_row0 ? codes[e] Wc : 0) + bias).
 * The residual rows of a rep-net's FIRST layer are a label embedding, z0[e] = codes[e] W_e with <= 16 code columns
 * (basemodel.py:1393-1420 feeding dmpnn.py:262-275's `zn = z + ...`): given the codes and the table the kernel adds them as one more
 * 16-deep k-group (codes [E, ldc % 4 == 0] fp32, split into bf16 pieces like every operand; Wc [kcodes <= 16, ldwc >= H]), and the [E, H]
 * rows z0 are never written nor read.  The reference builds one table per side (basemodel.py:51-52: share_emb_net only matters in
 * expand()), the union's first edges being the pattern's: code_row0 = their number, R [r_rows, ldr] = their (few) embedded rows, Wc = the
 * target's table; one shared table: code_row0 = 0, R = NULL.  H = 128, bf16x6, arrays below 4 GiB; DMP_ERR_UNSUPPORTED otherwise. */
int dmp_out_fwd_typed_codes(const dmp_typed_job *job, const float *codes, int64_t ldc, int kcodes, const float *Wc, int64_t ldwc,
                            int64_t code_row0, int64_t r_rows, const int32_t *slot_edge, const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, int64_t E,
                            int H, void *stream);

/*
 * The tall-skinny weight gradients over a tile list on the same kind of LDS image (csrc/dmp_h1w.hip::atb2_k): per job
 *     T = sum_e Z[e]^T D[e]      and, partial_B != NULL,      B = sum_e c(e) Z[e]^T D[e]     (c: tile_scale of e's tile)
 * -- dmp_atb_typed's two products (the class-typed edge chain's dA', dB', dmpnn.py:144-156 backward) and dmp_atb_rows_jobs_h's tile
 * form (the node side's weight gradients over the kept nodes' tiles) with every fetched element split into its bf16 pieces ONCE
 * (by the staging thread) and every fragment read through ds_read_b64_tr_b16.  H = 128, bf16x6, arrays below 4 GiB; otherwise
 * DMP_ERR_UNSUPPORTED (callers run dmp_atb_typed / dmp_atb_rows_jobs_h).  num_jobs <= 6 products over the SAME tile list share the
 * launch (grid.y = job); partials [dmp_atb2_blocks(tiles_bound, num_jobs)] per job, `partial_stride` floats apart, rows `ldp`
 * floats apart (partial_B = partial_T + H with ldp = 2 H: the [T | B] layout of dWes), summed by dmp_reduce_partials.
 */
typedef struct {
  const float *Z; int64_t ldz;
  const float *D; int64_t ldd;
  float *partial_T, *partial_B; int64_t partial_stride; int ldp;
} dmp_atb2_job;
int64_t dmp_atb2_blocks(int64_t tiles_bound, int num_jobs);
int dmp_atb2_jobs(const dmp_atb2_job *jobs, int num_jobs, const int32_t *slot_row, const float *tile_scale, const int32_t *num_tiles,
                  int64_t tiles_bound, int64_t rows, int H, void *stream);

/*
 * dmp_bwd_h1_typed AND the second Linear's weight gradient dO^T H1 (dmp_atb_typed's plain form) in ONE launch over the same tile
 * list (csrc/dmp_h1w.hip; the backward of `emlp[2]`, SubgraphCountingMatching/models/dmpnn.py:147-152: both products read the same
 * two [E, H] operands).  H = 128, bf16x6 arithmetic, arrays below 4 GiB; DMP_ERR_UNSUPPORTED otherwise (callers then run the two
 * launches).  A 512-thread workgroup per CU: four waves form dG as dmp_bwd_h1_typed does (bit-identical rows), four waves keep the
 * [128, 128] total of dO^T H1 in registers, both from ONE bf16-piece LDS image per operand (row reads / ds_read_b64_tr_b16).
 *   partial, partial_rows [dmp_bwd_h1_w_blocks(tiles_bound), H]: column sums of dG / of the fetched dO rows per workgroup;
 *   partial_w [dmp_bwd_h1_w_blocks(tiles_bound), H * H]: dO^T H1 per workgroup ([out feature of W2][in feature]: nn.Linear's
 *   layout); all three summed by dmp_reduce_partials.  The sign of H1 (the activation's derivative on the saved output) is read
 *   from its bf16 hi piece: an output of magnitude below 2^-133 counts as not positive.
 */
/* dmp_bwd_z_typed_arow AND dmp_atb_typed in ONE launch over the same class-sorted tile list (csrc/dmp_h1w.hip::dzw_k; both read dPre):
 *     dZ[e] = base[e] + s(flag e) D[dst e, half(flag e)] + dPre[e] W_g(e)^T,      partial_w [dmp_bwd_h1_w_blocks(tiles_bound), H * 2H] =
 *     [sum Z^T dPre | sum c Z^T dPre] per workgroup (the [T | B] layout of dWes; summed by dmp_reduce_partials).  W = [A'^T | B'^T]
 * (the w_transposed form of dmp_bwd_z_typed_arow); arguments as there, slot_arow = slot_edge.  dZ is bit-identical to that kernel's.
 * H = 128, bf16x6, arrays below 4 GiB; DMP_ERR_UNSUPPORTED otherwise (callers run the two launches). */
int dmp_bwd_z_w(const float *dPre, int64_t ldp, const float *Z, int64_t ldz, const float *W, int64_t ldw, const float *D, int64_t ldd,
                int64_t num_nodes, const float *base, int64_t ldb, const int32_t *dst, const uint8_t *flag, float s0, float s1,
                const int32_t *slot_edge, const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H,
                const int32_t *base_map, int64_t base_rows, float *dZ, int64_t ldo, float *partial_w, void *stream);
int64_t dmp_bwd_h1_w_blocks(int64_t tiles_bound);
int dmp_bwd_h1_w(const float *dO, int64_t ldo, const float *W2, int64_t ldw, const float *H1, int64_t ldh,
                 const int32_t *slot_edge, const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H, float slope,
                 float *dG, int64_t ldg, float *partial, float *partial_rows, float *partial_w, void *stream);

/*
 * Input gradient of the edge chain in one pass (replaces dmp_gather_select + the K=2H GEMM):
 *     dZ[e] = base[e] + s(flag e) * D[dst e, (flag e ? H : 0) + :] + dPre[e] W[:, 0:H]^T
 *             + coefE[e] * dPre[e] W[:, H:2H]^T
 *   dPre [E, ldp>=H] (e.g. the first half of dG, ldp = 2H), W [H, ldw>=2H] the forward weight panel,
 *   D [num_nodes, ldd>=2H] the gradient of dmp_seg_sum2's output, base [E, ldb] or NULL, s(0)=s0, s(1)=s1.
 */
int dmp_bwd_z_fused(const float *dPre, int64_t ldp, const float *W, int64_t ldw,
                    const float *D, int64_t ldd, int64_t num_nodes, const float *base, int64_t ldb,
                    const float *coefE, const int32_t *dst, const uint8_t *flag, float s0,
                    float s1, int64_t num_edges, int H, float *dZ, int64_t ldz, void *stream);

/*
 * Class-typed variants (csrc/dmp_typed.hip).  coefE[e] depends on out_deg[dst e] only, so all edges of
 * one degree class share the weight matrix W_g = W[:, 0:H] + c_g W[:, H:2H]; over tiles of 32 edges
 * of equal class the two products of dmp_edge_fwd_fused / dmp_bwd_z_fused collapse into one.
 *   slot_edge  [tiles_bound * 32] int32: edge id per tile slot, tiles sorted by class, -1 = padding
 *   tile_scale [tiles_bound] float     : c_g of each tile
 *   num_tiles  device int32 scalar     : tiles in use (<= tiles_bound, the host-side bound)
 * Rows are gathered / scattered by edge id; outputs equal the untyped kernels' up to fp32 rounding
 * of W_g.  DMP_ERR_UNSUPPORTED additionally when E*ld*4 does not fit 32 bits.
 * dmp_bwd_z_typed_arow, w_transposed != 0: W holds [A'^T | B'^T] (each half transposed) instead of [A' | B'] --
 * the per-class panel of dZ = dPre W_g^T is then read with coalesced loads (a strided panel read costs
 * every workgroup several microseconds per class segment).
 * dmp_bwd_z_typed_arow, base_map != NULL: `base` is a [base_rows, ldb] table and edge e adds row base_map[e] of it (< 0: nothing)
 * instead of row e of an [E, ldb] array -- the residual path of a layer whose output gradient is a per-graph vector
 * (see dmp_relu_bwd_gathered_colsum).
 */
int dmp_edge_fwd_typed(const float *Z, int64_t ldz, const float *W, int64_t ldw, const float *P,
                       int64_t ldp, int64_t num_nodes, const float *bias, const int32_t *selA,
                       const int32_t *selB, const int32_t *slot_edge, const float *tile_scale,
                       const int32_t *num_tiles, int64_t tiles_bound, int64_t num_edges, int H,
                       float slope, float *H1, int64_t ldh, void *stream);
/* Gated class tiles: out[s] = slot[s] if gate[slot[s]] != 0 else -1 (a tile slot list whose gated-out edges read as padding).
 * dmp_atb_typed takes it in place of slot_edge (a gated-out edge has dPre = 0: its z^T dPre term is zero, neither row is
 * fetched); dmp_bwd_z_typed_arow takes it as the row list of its streamed operand dPre (slot_arow; NULL = slot_edge), while the
 * output rows, the gathered term and the base rows still follow slot_edge: dz[e] = base[e] + select(D)[e] + 0 for such an edge. */
int dmp_mask_slots(const int32_t *slot, int64_t n, const float *gate, int64_t E, int32_t *out, void *stream);
int dmp_bwd_z_typed_arow(const float *dPre, int64_t ldp, const float *W, int64_t ldw, const float *D, int64_t ldd,
                         int64_t num_nodes, const float *base, int64_t ldb, const int32_t *dst, const uint8_t *flag,
                         float s0, float s1, const int32_t *slot_edge, const int32_t *slot_arow, const float *tile_scale,
                         const int32_t *num_tiles, int64_t tiles_bound, int64_t E, int H, int w_transposed,
                         const int32_t *base_map, int64_t base_rows, float *dZ, int64_t ldz, void *stream);

/*
 * Weight gradient of the class-typed edge chain:  with G_c = sum over the edges e of class c of
 * Z[e]^T dPre[e]  ([H,H]),  dA' = sum_c G_c  and  dB' = sum_c c_c G_c  (the two halves of dW for
 * W = [A' | B']) from ONE pass and one product's worth of MFMAs.  Every workgroup walks a contiguous
 * range of the class-sorted tiles and writes two [H,H] partials (running total, coefficient-weighted
 * total); partial_T / partial_B: [dmp_atb_typed_blocks_h(tiles_bound), H*H] floats each -- or, when
 * partial_B == partial_T + H, one [blocks, H, 2H] buffer with [T | B] side by side (its reduction is
 * dW = [dA' | dB'] in the layout of W) --, to be summed with dmp_reduce_partials (fixed order:
 * bit-stable for a given tile list).  partial_B NULL: the plain total sum_e Z[e]^T dPre[e] alone (a Linear's weight
 * gradient over the edges of a tile list, e.g. dO^T H1 over dmp_class_tiles_gated's kept edges).
 */
          /* H = 128 */
int64_t dmp_atb_typed_blocks_h(int64_t tiles_bound, int H); /* H = 128 or 64 (the reference's shipped hidden_dim) */
int dmp_atb_typed(const float *Z, int64_t ldz, const float *dPre, int64_t ldp,
                  const int32_t *slot_edge, const float *tile_scale, const int32_t *num_tiles,
                  int64_t tiles_bound, int64_t num_edges, int H, float *partial_T,
                  float *partial_B, void *stream);

/*
 * Tall-skinny weight-gradient product with an optional row gate fused in -- e.g. the second edge Linear's
 * gradient (dmpnn.py:263-273 backward: dO = gate * dOut, dW2 = dO^T H1, db2 = column sums of dO) -- in one
 * pass over plain 32-row tiles:
 *     partial[b]        = (gate (.) A)^T B  over workgroup b's rows   ([M,N])
 *     partial_colsum[b] = column sums of gate (.) A over the same rows ([M]; may be NULL)
 *   A [rows, lda>=M], B [rows, ldb>=N], gate [rows] or NULL (1); M and N multiples of 128 (every 128 x 128
 *   output block is one workgroup column of the launch);
 *   partial: [dmp_atb_rows_blocks_h(rows, M, N), M*N], partial_colsum: [dmp_atb_rows_blocks_h(rows, M, N), M];
 *   finish both with dmp_reduce_partials.
 */
/* The same with H x H output blocks, H = 128 (the two functions above) or 64: M and N multiples of H. */
int64_t dmp_atb_rows_blocks_h(int64_t rows, int M, int N, int H);
/* dmp_atb_rows_masked with a row mask (dmp_row_mask_bits of the gate: the rows under a zero gate, which contribute gate * a = 0 to
 * every sum, are not fetched) and the choice of the matrix pipe: x6 != 0 = bf16x6 (fp32-accurate, dmp_dev_set_exact_fp32
 * overrides), 0 = the f32-input MFMA. */
int dmp_atb_rows_masked(const float *A, int64_t lda, const float *B, int64_t ldb, const float *gate, const uint32_t *rowmask,
                        int x6, int64_t rows, int M, int N, int H, float *partial, float *partial_colsum, void *stream);
/* The rows product without a gate and without column sums, over the rows of a row mask (NULL: all): A^T B restricted to the masked-in
 * rows -- which IS (gate (.) A)^T B for a 0 / 1 gate whose mask it is.  On the bf16 pipe (bf16x6; the gated form above does not fit the
 * register file there); the bias gradient (column sums of the gated A) comes from dmp_smallk_atb_cols_masked with the gate as its
 * one-column X.  partial as dmp_atb_rows_masked (same number of blocks). */
int dmp_atb_rows_plain(const float *A, int64_t lda, const float *B, int64_t ldb, const uint32_t *rowmask, int64_t rows, int M, int N,
                       int H, float *partial, void *stream);

/*
 * Relation-typed products of the relational layers (SubgraphCountingMatching/models/rgcn.py:98-123,
 * rgin.py:100-125: msg_e = X[src e] W[type e] (* norm e), summed by destination; the reference copies the
 * weights to [E, in, out] with index_select + bmm).  Edges are sorted by type and cut into 32-slot tiles
 * that never mix types (slot arrays [tiles * 32], -1 = padding; tile_type [tiles]; both in type order):
 *
 *   dmp_rel_gemm   C[slot_row[s]] = row_scale[slot_row[s]] * A[slot_arow[s]] W[tile type]     (w_transposed: W^T)
 *                  forward (A = X by source node, W) and input gradient (A = d_agg by destination, W^T);
 *                  W [num_rels, 128, ldw >= 128], num_tiles a device scalar <= tiles_bound, row_scale [rows_c] or NULL.
 *   dmp_rel_atb    partial[t][b] = sum over type t's slots s of workgroup b of
 *                                  slot_scale[s] * X[slot_x[s]]^T D[slot_d[s]]               ([128,128] each)
 *                  the per-type weight gradient; type_tile_ptr [num_rels + 1] (device) = first tile of every
 *                  type, tiles = their total; partial [num_rels, dmp_rel_atb_blocks(num_rels), 128*128], every
 *                  entry written (zeros for a type without edges); the sum over b in a fixed order is dW[t].
 * H = in = out = 128 only (DMP_ERR_UNSUPPORTED otherwise; the caller then runs one library GEMM per type).
 */
int dmp_rel_gemm(const float *A, int64_t lda, int64_t rows_a, const float *W, int64_t ldw, int num_rels,
                 int w_transposed, const int32_t *slot_arow, const int32_t *slot_row,
                 const int32_t *tile_type, const int32_t *num_tiles, int64_t tiles_bound,
                 const float *row_scale, int64_t rows_c, int H, float *C, int64_t ldc, void *stream);
int64_t dmp_rel_atb_blocks(int num_rels);
int dmp_rel_atb(const float *X, int64_t ldx, int64_t rows_x, const float *D, int64_t ldd, int64_t rows_d,
                const int32_t *slot_x, const int32_t *slot_d, const float *slot_scale,
                const int32_t *type_tile_ptr, int num_rels, int64_t tiles, int H, float *partial,
                void *stream);

/* Several such products over the SAME rows in one launch (the node update's three weight gradients): every job is
 * one 128 x 128 output block  partial[b] = (gate (.) A[:, 0:128])^T B[:, 0:128]  (callers pass column-offset
 * pointers for the blocks of a wider product), written at partial + b * partial_stride with row stride ldp;
 * partial_colsum + b * cs_ld (or NULL) gets the column sums of gate (.) A.  b < dmp_atb_jobs_blocks_h(rows, num_jobs):
 * the launch's workgroups are shared by all jobs.  `jobs` is a HOST array of num_jobs <= DMP_ATB_MAX_JOBS entries. */
#define DMP_ATB_MAX_JOBS 8
typedef struct {
  const float *A; int64_t lda; const float *B; int64_t ldb; const float *gate;
  float *partial; int64_t partial_stride; int ldp; float *partial_colsum; int cs_ld;
  const uint32_t *rowmask;   /* optional (dmp_row_mask_bits): rows whose bit is 0 -- rows where A (after its gate) or B is known to be
                              * all zeros -- are not fetched */
} dmp_atb_job;
/* H x H output blocks per job, H = 128 (the two functions above) or 64. */
int64_t dmp_atb_jobs_blocks_h(int64_t rows, int num_jobs, int H);
/* slot_row != NULL: the jobs run over a TILE LIST instead of the rows 0 .. rows-1 -- slot_row [tiles_bound * 32] row ids (-1 =
 * padding), tile_scale [tiles_bound] (constant: one "class"), *num_tiles tiles in use (device): dmp_kept_rows(tiles = 1) of a
 * 0 / 1 node gate, i.e. the node side's weight gradients over the kept nodes only (rows of A and B gathered by slot; jobs
 * without gate / column sums / row mask; partials [dmp_atb_tile_jobs_blocks(tiles_bound, num_jobs, H)] per job).  Else NULLs. */
int64_t dmp_atb_tile_jobs_blocks(int64_t tiles_bound, int num_jobs, int H);
int dmp_atb_rows_jobs_h(const dmp_atb_job *jobs, int num_jobs, int64_t rows, int H, const int32_t *slot_row,
                        const float *tile_scale, const int32_t *num_tiles, int64_t tiles_bound, void *stream);
/* (a launch none of whose jobs has a gate or column sums runs on the bf16 pipe, bf16x6: the 0 / 1-gated products of a step come as
 * row masks, their column sums from dmp_bwd_h1_fused_rows (partial_rows)) */

/* Development switch (not part of the product path): 0 = independent 256-thread workgroups (default),
 * 1 = the experimental "ping-pong" driver of csrc/dmp_mfma.hip (two wave groups per 512-thread workgroup
 * held half an iteration apart by barriers).  Process-wide; results are identical. */
void dmp_dev_set_mfma_variant(int variant);


/* Arithmetic of the class-typed kernels' products (dmp_edge_fwd_typed, dmp_bwd_z_typed_arow, dmp_rel_gemm).  Default (0):
 * fp32 operands split into three bf16 pieces each, six piece products per 16-deep k-group on the bf16 matrix pipe,
 * fp32 accumulation ("bf16x6": every partial product carried to 2^-24 of its magnitude, i.e. fp32-accurate; gfx950's
 * f32-input MFMA has 1/16 of the bf16 MFMA rate).  1: the f32-input MFMA (bitwise an fmaf chain), kept for comparison
 * and as the reference point of the parity tests.  Process-wide development switch. */
void dmp_dev_set_exact_fp32(int on);
int dmp_dev_get_exact_fp32(void);

/* Plain C[E, ncols] = A[E,128] B (ncols = 128 or 256; B[k*ldb+j], or B[j*ldb+k] if b_transposed):
 * the bare pipeline of the two kernels above, kept for tests and tuning. */
int dmp_gemm_k128(const float *A, int64_t lda, const float *B, int64_t ldb, int b_transposed,
                  float *C, int64_t ldc, int64_t rows, int ncols, void *stream);
/* C[E, 64] = A[E, 64] B: the same pipeline at the reference's shipped hidden_dim (two waves per workgroup). */
int dmp_gemm_k64(const float *A, int64_t lda, const float *B, int64_t ldb, int b_transposed,
                 float *C, int64_t ldc, int64_t rows, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DMP_HIP_H */
