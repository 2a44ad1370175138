"""Frame keys of the graph objects.

The key STRINGS are part of the drop-in boundary: the reference's drivers and models address ``ndata`` / ``edata`` by
them (SubgraphCountingMatching/constants.py:12-34), so a graph prepared by the reference's dataset code and a graph
prepared here carry the same entries.  Grouped by what writes them."""

# numeric sentinels used by the masking / padding helpers
INF, _INF, EPS = 1e30, -1e30, 1e-8
LEAKY_RELU_A = 1 / 5.5                       # negative slope of the reference's "leaky_relu" (utils/act.py:27)

# dataset passes: ids and labels (nodes and edges share the strings, the frames differ), reversed / loop flags
NODEID = EDGEID = "id"
NODELABEL = EDGELABEL = "label"
REVFLAG, LOOPFLAG = "is_reversed", "is_loop"
NODETYPE, EDGETYPE = "node_type", "edge_type"

# degree bookkeeping the layers cache on the graph, and the normalisers derived from it
INDEGREE, OUTDEGREE = "in_deg", "out_deg"
INNORM, OUTNORM, NORM = "in_norm", "out_norm", "norm"
NODEEIGENV, EDGEEIGENV = "node_eigenv", "edge_eigenv"

# what a message-passing layer reads and leaves behind (features in, messages, aggregates, outputs)
NODEFEAT, EDGEFEAT = "node_feat", "edge_feat"
NODEMSG, EDGEMSG = "node_msg", "edge_msg"
NODEAGG, EDGEAGG = "node_agg", "edge_agg"
NODEOUTPUT, EDGEOUTPUT = "node_out", "edge_out"
