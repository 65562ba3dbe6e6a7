// codec.cpp — ToySlam's binary graph message, both directions, bounds-checked.
//
// Request  (client -> server): python/remote/graph_to_bytes.py:32-67 writes it,
//          remote/serialization/DeserializeGraph.h:18-173 reads it.
// Response (server -> client): remote/serialization/SerializeGraph.h:17-71 with
//          SerializeGraphFuncCpu.h:10-65 and SerializeHelper.cpp:22-56 write it,
//          python/remote/bytes_to_graph.py:16-108 reads it.  The two forms are NOT symmetric: the
//          reply always carries measurements as rows x cols matrices (ODOM 3x3, LM 2x1).
// All integers/floats are raw little-endian u32 / f32 (SerializeHelper.cpp:5-19).
//
// Unlike the reference (which trusts the buffer and `throw`s C strings on unknown types,
// DeserializeGraph.h:94,114,135,146,159), every read is bounds-checked and errors come back as
// status codes.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/tsgo.h"
#include "errors.h"

struct tsgo_wire_graph {
    std::vector<uint32_t> v_id, v_type;
    std::vector<double> v_pos;            // 3 per vertex
    std::vector<uint32_t> e_type, e_ids;
    std::vector<double> e_meas, e_inf;    // 9 / 3 per edge
    std::vector<float> e_meas_f32;        // 9 per edge: what goes back on the wire
    std::vector<float> e_inf_f32;         // 3 per edge
    std::vector<uint32_t> fixed;
};

namespace {

struct Reader {
    const uint8_t* p; size_t len, off = 0; bool ok = true;
    bool need(size_t n) { if (!ok || len - off < n) { ok = false; return false; } return true; }
    uint32_t u32() { uint32_t v = 0; if (need(4)) { std::memcpy(&v, p + off, 4); off += 4; } return v; }
    float f32() { float v = 0; if (need(4)) { std::memcpy(&v, p + off, 4); off += 4; } return v; }
};

struct Writer {
    uint8_t* buf; size_t cap, off = 0;
    void raw(const void* src, size_t n) { if (buf && off + n <= cap) std::memcpy(buf + off, src, n); off += n; }
    void u32(uint32_t v) { raw(&v, 4); }
    void f32(float v) { raw(&v, 4); }
};

// remote/graph/Helper.h:6-19
void transform_from_xyt(double x, double y, double th, double* m) {
    const double c = std::cos(th), s = std::sin(th);
    m[0] = c; m[1] = -s; m[2] = x; m[3] = s; m[4] = c; m[5] = y; m[6] = 0; m[7] = 0; m[8] = 1;
}

}  // namespace

extern "C" int tsgo_wire_decode(const uint8_t* payload, size_t len, tsgo_wire_graph** out) {
    if (!payload || !out) return tsgo::set_error(-1, "tsgo_wire_decode: null argument");
    auto* w = new tsgo_wire_graph();
    Reader r{payload, len};
    auto fail = [&](const std::string& why) { delete w; *out = nullptr; return tsgo::set_error(-2, "wire decode: " + why); };

    const uint32_t nV = r.u32();                                        // DeserializeGraph.h:20
    if (!r.ok || (size_t)nV > len / 16) return fail("vertex count does not fit the payload");
    w->v_id.resize(nV); w->v_type.resize(nV); w->v_pos.resize((size_t)nV * 3);      // sized once, written by index
    for (uint32_t i = 0; i < nV; ++i) {
        const uint32_t id = r.u32(), type = r.u32();                    // :28-31
        double x, y, th = 0;
        if (type == 0) { x = r.f32(); y = r.f32(); th = r.f32(); }      // :32-44
        else if (type == 1) { x = r.f32(); y = r.f32(); }               // :45-53
        else return fail("unknown vertex type " + std::to_string(type));
        if (!r.ok) return fail("truncated in vertex " + std::to_string(i));
        w->v_id[i] = id; w->v_type[i] = type;
        w->v_pos[3 * (size_t)i] = x; w->v_pos[3 * (size_t)i + 1] = y; w->v_pos[3 * (size_t)i + 2] = th;
    }
    const uint32_t nE = r.u32();                                        // :55
    if (!r.ok || (size_t)nE > len / 36) return fail("edge count does not fit the payload");
    w->e_type.resize(nE); w->e_ids.resize((size_t)nE * 2);
    w->e_meas.resize((size_t)nE * 9); w->e_inf.resize((size_t)nE * 3);
    w->e_meas_f32.resize((size_t)nE * 9); w->e_inf_f32.resize((size_t)nE * 3);
    for (uint32_t i = 0; i < nE; ++i) {
        const uint32_t type = r.u32(), id1 = r.u32(), id2 = r.u32(), rows = r.u32(), cols = r.u32();   // :61-70
        if (!r.ok) return fail("truncated in edge " + std::to_string(i));
        if (type > 1) return fail("unknown edge type " + std::to_string(type));                          // :93-95
        double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        float mf[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (rows == 0) {                                                                                 // :74
            if (type == 0) {                                                                             // :76-85 (x, y, theta)
                const double x = r.f32(), y = r.f32(), th = r.f32();
                transform_from_xyt(x, y, th, m);
                for (int k = 0; k < 9; ++k) mf[k] = (float)m[k];
            } else {                                                                                     // :86-91 (range, bearing)
                mf[0] = r.f32(); mf[1] = r.f32(); m[0] = mf[0]; m[1] = mf[1];
            }
        } else {
            if (type != 0) return fail("LM measurement must use the rows==0 form");                      // :112-115
            if (cols > 3) return fail("ODOM measurement wider than 3x3");
            m[0] = m[4] = m[8] = 1;                                                                      // :101 identity
            for (uint32_t rr = 0; rr < cols; ++rr)
                for (uint32_t cc = 0; cc < cols; ++cc) m[rr * 3 + cc] = r.f32();                         // :103-110
            for (int k = 0; k < 9; ++k) mf[k] = (float)m[k];
        }
        const uint32_t rows_inf = r.u32(), cols_inf = r.u32();                                           // :118-121
        if (!r.ok) return fail("truncated in edge " + std::to_string(i));
        if (rows_inf != 0) return fail("only diagonal information matrices are supported");              // :144-147
        const uint32_t dim = type == 0 ? 3u : 2u;
        if (cols_inf > dim) return fail("information diagonal longer than the residual");
        float wf[3] = {1, 1, 1};                                                                         // :125-131 identity
        if (type == 1) wf[2] = 0;
        for (uint32_t j = 0; j < cols_inf; ++j) wf[j] = r.f32();                                         // :138-142
        if (!r.ok) return fail("truncated in edge " + std::to_string(i));
        w->e_type[i] = type; w->e_ids[2 * (size_t)i] = id1; w->e_ids[2 * (size_t)i + 1] = id2;
        for (int k = 0; k < 9; ++k) { w->e_meas[9 * (size_t)i + k] = m[k]; w->e_meas_f32[9 * (size_t)i + k] = mf[k]; }
        for (int k = 0; k < 3; ++k) { w->e_inf[3 * (size_t)i + k] = wf[k]; w->e_inf_f32[3 * (size_t)i + k] = wf[k]; }
    }
    const uint32_t nF = r.u32();                                        // :163
    if (!r.ok || (size_t)nF > len / 4) return fail("fixed-vertex count does not fit the payload");
    for (uint32_t i = 0; i < nF; ++i) w->fixed.push_back(r.u32());      // :167-172
    if (!r.ok) return fail("truncated in the fixed-vertex list");
    *out = w;
    return 0;
}

extern "C" void tsgo_wire_view(const tsgo_wire_graph* w, tsgo_graph* g) {
    g->n_vertices = (int32_t)w->v_id.size(); g->v_id = w->v_id.data(); g->v_type = w->v_type.data(); g->v_pos = w->v_pos.data();
    g->n_edges = (int32_t)w->e_type.size(); g->e_type = w->e_type.data(); g->e_ids = w->e_ids.data();
    g->e_meas = w->e_meas.data(); g->e_inf = w->e_inf.data();
    g->n_fixed = (int32_t)w->fixed.size(); g->fixed = w->fixed.data();
}

extern "C" void tsgo_wire_free(tsgo_wire_graph* w) { delete w; }

extern "C" int64_t tsgo_wire_encode_response(const tsgo_wire_graph* w, const double* v_pos, uint8_t* buf, size_t cap) {
    if (!w || !v_pos) return tsgo::set_error(-1, "tsgo_wire_encode_response: null argument");
    Writer o{buf, cap};
    o.u32(0);                                                           // size prefix, patched below (SerializeGraph.h:68-69)
    const size_t nV = w->v_id.size(), nE = w->e_type.size();
    o.u32((uint32_t)nV);                                                // :23-24
    for (size_t i = 0; i < nV; ++i) {                                   // SerializeGraphFuncCpu.h:10-41
        o.u32(w->v_id[i]); o.u32(w->v_type[i]);
        o.f32((float)v_pos[3 * i]); o.f32((float)v_pos[3 * i + 1]);
        if (w->v_type[i] == 0) o.f32((float)v_pos[3 * i + 2]);
    }
    o.u32((uint32_t)nE);                                                // SerializeGraph.h:45-46
    for (size_t e = 0; e < nE; ++e) {                                   // SerializeGraphFuncCpu.h:43-63
        o.u32(w->e_type[e]); o.u32(w->e_ids[2 * e]); o.u32(w->e_ids[2 * e + 1]);
        const float* m = &w->e_meas_f32[9 * e];
        const float* d = &w->e_inf_f32[3 * e];
        if (w->e_type[e] == 0) {
            o.u32(3); o.u32(3); for (int k = 0; k < 9; ++k) o.f32(m[k]);            // MatrixToByteArray(rows, cols, .., false)
            o.u32(0); o.u32(3); for (int k = 0; k < 3; ++k) o.f32(d[k]);            // (.., true): 0, rows, diagonal
        } else {
            o.u32(2); o.u32(1); o.f32(m[0]); o.f32(m[1]);                           // Vector2 = 2 rows x 1 col
            o.u32(0); o.u32(2); o.f32(d[0]); o.f32(d[1]);
        }
    }
    o.u32((uint32_t)w->fixed.size());                                   // SerializeGraph.h:59-66
    for (uint32_t id : w->fixed) o.u32(id);
    if (buf && o.off <= cap) { const uint32_t sz = (uint32_t)(o.off - 4); std::memcpy(buf, &sz, 4); }
    if (o.off - 4 > 0xFFFFFFFFull) return tsgo::set_error(-3, "response larger than 4 GiB");
    return (int64_t)o.off;
}

extern "C" int64_t tsgo_wire_encode_request(const tsgo_graph* g, uint8_t* buf, size_t cap) {
    if (!g) return tsgo::set_error(-1, "tsgo_wire_encode_request: null argument");
    Writer o{buf, cap};
    o.u32(0);                                                           // graph_to_bytes.py:67 length prefix
    o.u32((uint32_t)g->n_vertices);                                     // :44
    for (int i = 0; i < g->n_vertices; ++i) {                           // :45-50
        o.u32(g->v_id[i]); o.u32(g->v_type[i]);
        o.f32((float)g->v_pos[3 * (size_t)i]); o.f32((float)g->v_pos[3 * (size_t)i + 1]);
        if (g->v_type[i] == 0) o.f32((float)g->v_pos[3 * (size_t)i + 2]);
    }
    o.u32((uint32_t)g->n_edges);                                        // :53
    for (int e = 0; e < g->n_edges; ++e) {                              // :54-59
        const double* m = g->e_meas + 9 * (size_t)e;
        const double* d = g->e_inf + 3 * (size_t)e;
        o.u32(g->e_type[e]); o.u32(g->e_ids[2 * (size_t)e]); o.u32(g->e_ids[2 * (size_t)e + 1]);
        if (g->e_type[e] == 0) {
            o.u32(3); o.u32(3); for (int k = 0; k < 9; ++k) o.f32((float)m[k]);     // 2-D array: rows, cols (:25-27)
            o.u32(0); o.u32(3); for (int k = 0; k < 3; ++k) o.f32((float)d[k]);     // is_diag (:16-20)
        } else {
            o.u32(0); o.u32(2); o.f32((float)m[0]); o.f32((float)m[1]);             // 1-D array: 0, n (:22-24)
            o.u32(0); o.u32(2); o.f32((float)d[0]); o.f32((float)d[1]);
        }
    }
    o.u32((uint32_t)g->n_fixed);                                        // :62
    for (int i = 0; i < g->n_fixed; ++i) o.u32(g->fixed[i]);
    if (buf && o.off <= cap) { const uint32_t sz = (uint32_t)(o.off - 4); std::memcpy(buf, &sz, 4); }
    return (int64_t)o.off;
}
