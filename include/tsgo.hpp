// tsgo.hpp — header-only C++ host side above the C ABI (tsgo.h), shaped like the reference's own in-process
// interfaces so that code written against ToySlam's classes reads the same:
//
//   reference (remote/…)                                        here (namespace tsgo)
//   graph/vertex/VertexType.h:3-7   enum class VertexType       VertexType { Se2 = 0, Point2 = 1 }
//   graph/edge/EdgeType.h:3-7       enum class EdgeType         EdgeType   { Se2 = 0, Se2Point2 = 1 }
//   graph/GraphCpu.h:15-28          AddVertex / AddEdge /       Graph::AddVertex / AddEdge / FixVertex
//                                   FixVertex                   (values instead of unique_ptr<BaseVertexCpu<T>>)
//   graph/GraphCpu.h:45-53          GetVertex(id)               Graph::GetVertex(id) -> (x, y, theta) / (x, y, 0)
//   optimizer/IOptimizer.h:10-26    IOptimizer(iterations,      OptimizerHip(iterations[, config]);
//                                   solver); Optimize(IGraph*)  Optimize(Graph*)  — graph mutated in place, errors
//                                                               printed and the call returns early (OptimizerCpu.h style)
//
// The solver argument of the reference's optimizers has no counterpart: the implicit-Schur PCG is part of the
// library (SolverEigen.h:20 is what it replaces).  Nothing here touches the device directly; link libtsgo_hip.so.
#pragma once

#include <cmath>
#include <cstdint>
#include <iostream>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "tsgo.h"

namespace tsgo {

enum class VertexType : uint32_t { Se2 = 0, Point2 = 1 };
enum class EdgeType : uint32_t { Se2 = 0, Se2Point2 = 1, Se2VirtualPoint2 = 2 };      // 2: behind the C ABI only (include/tsgo.h), not in the reference's enum

class Graph {
public:
    // Se2: (x, y, theta); Point2: (x, y).  Same role as Functions::CreateVertex + GraphCpu::AddVertex
    // (remote/serialization/DeserializeGraphFuncCpu.h:14-25, GraphCpu.h:15-18).
    void AddVertex(unsigned id, VertexType type, double x, double y, double theta = 0.0) {
        if (index.count(id)) throw std::invalid_argument("AddVertex: duplicate vertex id");
        index[id] = v_id.size();
        v_id.push_back(id); v_type.push_back((uint32_t)type);
        v_pos.push_back(x); v_pos.push_back(y); v_pos.push_back(type == VertexType::Se2 ? theta : 0.0);
    }
    // ODOM edge: measurement = relative pose (x, y, theta) — the 3x3 transform of EdgeSe2.h:23-38 is built here as
    // remote/graph/Helper.h:6-19 does; information = its diagonal (the only form the wire carries, DeserializeGraph.h:123-147).
    void AddEdgeSe2(unsigned id1, unsigned id2, double mx, double my, double mtheta, double w0, double w1, double w2) {
        const double c = std::cos(mtheta), s = std::sin(mtheta);
        const double m[9] = {c, -s, mx, s, c, my, 0, 0, 1};
        push_edge(EdgeType::Se2, id1, id2, m, w0, w1, w2);
    }
    // LM edge: measurement = (range, bearing) (EdgeSe2Point2d.h:34-35); information = diag(w0, w1).
    void AddEdgeSe2Point2(unsigned id_pose, unsigned id_landmark, double range, double bearing, double w0, double w1) {
        const double m[9] = {range, bearing, 0, 0, 0, 0, 0, 0, 0};
        push_edge(EdgeType::Se2Point2, id_pose, id_landmark, m, w0, w1, 0.0);
    }
    // Virtual landmark measurement (README.md:53; python/optimizer/edges2d.py:83-121): the same physical point seen from two poses as
    // (range, bearing) each; information = diag(w0, w1).  No wire encoding exists for it.
    void AddEdgeVirtualLandmark(unsigned id_pose_1, unsigned id_pose_2, double range1, double bearing1, double range2, double bearing2, double w0, double w1) {
        const double m[9] = {range1, bearing1, range2, bearing2, 0, 0, 0, 0, 0};
        push_edge(EdgeType::Se2VirtualPoint2, id_pose_1, id_pose_2, m, w0, w1, 0.0);
    }
    // generic form, same argument meaning as Functions::CreateEdge(type, id1, id2, meas, inf)
    // (DeserializeGraphFuncCpu.h:27-38): meas = 9 doubles row-major (ODOM) or (range, bearing, 0...) (LM)
    void AddEdge(EdgeType type, unsigned id1, unsigned id2, const double meas[9], const double inf_diag[3]) {
        push_edge(type, id1, id2, meas, inf_diag[0], inf_diag[1], inf_diag[2]);
    }
    void FixVertex(unsigned id) { fixed.push_back(id); }                                 // GraphCpu.h:25-28

    struct Position { double x, y, theta; };
    Position GetVertex(unsigned id) const {                                              // GraphCpu.h:45-53 (.at throws)
        const size_t i = index.at(id);
        return {v_pos[3 * i], v_pos[3 * i + 1], v_pos[3 * i + 2]};
    }
    size_t VertexCount() const { return v_id.size(); }
    size_t EdgeCount() const { return e_type.size(); }
    const std::vector<unsigned>& GetFixedVertices() const { return fixed; }              // GraphCpu.h:40-43

    tsgo_graph View() const {
        return tsgo_graph{(int32_t)v_id.size(), v_id.data(), v_type.data(), v_pos.data(), (int32_t)e_type.size(), e_type.data(),
                          e_ids.data(), e_meas.data(), e_inf.data(), (int32_t)fixed.size(), fixed.data()};
    }
    void SetPositions(const std::vector<double>& xyt) { v_pos = xyt; }

private:
    void push_edge(EdgeType type, unsigned id1, unsigned id2, const double* m, double w0, double w1, double w2) {
        e_type.push_back((uint32_t)type); e_ids.push_back(id1); e_ids.push_back(id2);
        e_meas.insert(e_meas.end(), m, m + 9);
        e_inf.push_back(w0); e_inf.push_back(w1); e_inf.push_back(w2);
    }
    std::vector<uint32_t> v_id, v_type, e_type, e_ids, fixed;
    std::vector<double> v_pos, e_meas, e_inf;
    std::unordered_map<unsigned, size_t> index;
};

class OptimizerHip {
public:
    explicit OptimizerHip(unsigned iterations, const tsgo_config* config = nullptr) : iterations(iterations) {
        tsgo_config c;
        if (config) c = *config; else tsgo_default_config(&c);
        if (tsgo_create(&c, &handle)) throw std::runtime_error(tsgo_last_error());     // no device: fails loudly, no CPU path
    }
    ~OptimizerHip() { tsgo_destroy(handle); }
    OptimizerHip(const OptimizerHip&) = delete;
    OptimizerHip& operator=(const OptimizerHip&) = delete;

    // IOptimizer<T>::Optimize(IGraph*) (IOptimizer.h:21): in place, no return value; the reference prints and returns
    // on errors (OptimizerCpu.h:28-33) and prints its stop reason and "Summary() error" (:146,:169,:175,:182).
    void Optimize(Graph* graph) {
        if (!graph) return;
        const tsgo_graph g = graph->View();
        if (tsgo_set_graph(handle, &g) || tsgo_optimize(handle, (int)iterations, &stats)) { std::cout << tsgo_last_error() << std::endl; return; }
        std::vector<double> out(graph->VertexCount() * 3);
        if (tsgo_get_vertices(handle, out.data())) { std::cout << tsgo_last_error() << std::endl; return; }
        graph->SetPositions(out);
        if (stats.stop_reason == TSGO_STOP_WORSE) std::cout << "Error is getting worse\n";
        if (stats.stop_reason == TSGO_STOP_PLATEAU) std::cout << "Plateau: NO MORE OPT\n";
        if (stats.stop_reason == TSGO_STOP_CONVERGED) std::cout << "CONVERGED\n";
        const int last = stats.iterations_run > 0 ? (stats.iterations_run < TSGO_MAX_TRACE ? stats.iterations_run : TSGO_MAX_TRACE) - 1 : 0;
        std::cout << "Summary() error = " << stats.chi2[last] << std::endl;
    }
    const tsgo_stats& Stats() const { return stats; }

private:
    unsigned iterations;
    tsgo_optimizer* handle = nullptr;
    tsgo_stats stats{};
};

}  // namespace tsgo
