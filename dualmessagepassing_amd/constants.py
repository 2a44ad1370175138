"""Frame keys of the graph objects -- the same strings the reference uses
(SubgraphCountingMatching/constants.py:12-34), because they are part of the
drop-in boundary: drivers and models address ``ndata`` / ``edata`` by them."""

INF = 1e30
_INF = -1e30
EPS = 1e-8

LEAKY_RELU_A = 1 / 5.5

LOOPFLAG = "is_loop"
REVFLAG = "is_reversed"
NORM = "norm"
INDEGREE = "in_deg"
INNORM = "in_norm"
OUTDEGREE = "out_deg"
OUTNORM = "out_norm"
NODEID = "id"
EDGEID = "id"
NODELABEL = "label"
EDGELABEL = "label"
NODEEIGENV = "node_eigenv"
EDGEEIGENV = "edge_eigenv"
NODEFEAT = "node_feat"
EDGEFEAT = "edge_feat"
NODETYPE = "node_type"
EDGETYPE = "edge_type"
NODEMSG = "node_msg"
EDGEMSG = "edge_msg"
NODEAGG = "node_agg"
EDGEAGG = "edge_agg"
NODEOUTPUT = "node_out"
EDGEOUTPUT = "edge_out"
