"""toyslam_amd — MI355X-native drop-in for the Gauss-Newton hot path of ToySlam's remote graph_optimizer.

The product is the C-ABI library built from toyslam_amd/csrc (include/tsgo.h) plus the C++ TCP server
`graph_optimizer`; this package is the thin Python host side used by tests and bench.py:

    graph.py      the OptGraph model (mirrors python/optimizer/{opt_graph,vertices,edges2d}.py)
    optimizer.py  HipOptimizer / GraphOptimizer on top of the C ABI (mirrors graph_optimizer.py)
    remote.py     wire codec + client (mirrors python/remote/*)
    synth.py      synthetic graphs for BASELINE.json configs 2-5
    build.py      hipcc / g++ recipes (in-tree .so files)
"""
from .graph import (EdgeLandmark2d, EdgeOdometry2d, GraphArrays, OptGraph, Vertex2d,  # noqa: F401
                    VertexPose2d)
