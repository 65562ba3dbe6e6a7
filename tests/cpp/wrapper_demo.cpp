// Drives the hot path through include/tsgo.hpp the way reference code drives GraphCpu + IOptimizer:
//   wrapper_demo <request.bin> <iterations> <vertices_out.txt>
// request.bin: a ToySlam wire request with its 4-byte length prefix (tests/golden/c1_request.bin is the reference's
// own graph_to_bytes output).  Writes "id type x y theta" per vertex and the chi^2 trace to the output file.
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

#include "tsgo.hpp"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: wrapper_demo request.bin iterations out.txt\n"); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (bytes.size() < 8) { std::fprintf(stderr, "empty request\n"); return 2; }
    tsgo_wire_graph* w = nullptr;
    if (tsgo_wire_decode(bytes.data() + 4, bytes.size() - 4, &w)) { std::fprintf(stderr, "%s\n", tsgo_last_error()); return 1; }
    tsgo_graph view; tsgo_wire_view(w, &view);

    tsgo::Graph graph;                                   // DeserializeGraph.h:43,52,151,172: AddVertex / AddEdge / FixVertex
    for (int i = 0; i < view.n_vertices; ++i)
        graph.AddVertex(view.v_id[i], (tsgo::VertexType)view.v_type[i], view.v_pos[3 * i], view.v_pos[3 * i + 1], view.v_pos[3 * i + 2]);
    for (int e = 0; e < view.n_edges; ++e)
        graph.AddEdge((tsgo::EdgeType)view.e_type[e], view.e_ids[2 * e], view.e_ids[2 * e + 1], view.e_meas + 9 * (size_t)e, view.e_inf + 3 * (size_t)e);
    for (int i = 0; i < view.n_fixed; ++i) graph.FixVertex(view.fixed[i]);

    tsgo::OptimizerHip optimizer((unsigned)std::atoi(argv[2]));
    optimizer.Optimize(&graph);

    std::FILE* out = std::fopen(argv[3], "w");
    if (!out) return 2;
    for (int i = 0; i < view.n_vertices; ++i) {
        const auto p = graph.GetVertex(view.v_id[i]);
        std::fprintf(out, "v %u %u %.17g %.17g %.17g\n", view.v_id[i], view.v_type[i], p.x, p.y, p.theta);
    }
    const tsgo_stats& st = optimizer.Stats();
    for (int k = 0; k < st.iterations_run && k < TSGO_MAX_TRACE; ++k) std::fprintf(out, "chi2 %d %.17g\n", k, st.chi2[k]);
    std::fprintf(out, "stop %d\n", st.stop_reason);
    std::fclose(out);
    tsgo_wire_free(w);
    return 0;
}
