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
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/tsgo.h"
#include "errors.h"
#include "parallel.h"

// Arrays are allocated uninitialised (a 100k-pose request decodes into ~120 MB: zero-filling it first costs as much as
// writing it) and every element is written by the decoder.
template <typename V> struct Raw {
    std::unique_ptr<V[]> p; size_t n = 0, cap = 0;
    // grow-only: a handle that is decoded into again (tsgo_wire_decode_into) keeps its pages — fresh pages for 120 MB cost more
    // than parsing into them
    // (a quarter of headroom when it must grow: a front-end that appends a few poses per message then reallocates rarely)
    void alloc(size_t count) { if (count > cap || !p) { cap = count + count / 4 + 1; p.reset(new V[cap]); } n = count; }
    V* data() { return p.get(); } const V* data() const { return p.get(); }
    size_t size() const { return n; }
    V& operator[](size_t k) { return p[k]; } const V& operator[](size_t k) const { return p[k]; }
};

struct tsgo_wire_graph {
    Raw<uint32_t> v_id, v_type;
    Raw<double> v_pos;            // 3 per vertex
    Raw<uint32_t> e_type, e_ids;
    Raw<double> e_meas, e_inf;    // 9 / 3 per edge; every value is exactly an f32 (or computed from f32s the way the reference
                                  // does), so the reply's floats are plain casts of these
    Raw<uint32_t> fixed;
    size_t reply_bytes = 0;       // length of the reply message, prefix included
};

namespace {

constexpr int kChunk = 16384;     // records per work unit of the parallel passes

inline uint32_t rd_u32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline float rd_f32(const uint8_t* p) { float v; std::memcpy(&v, p, 4); return v; }
inline void wr_u32(uint8_t* p, uint32_t v) { std::memcpy(p, &v, 4); }
inline void wr_f32(uint8_t* p, float v) { std::memcpy(p, &v, 4); }

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

inline size_t reply_vertex_bytes(uint32_t type) { return type == 0 ? 20 : 16; }     // SerializeGraphFuncCpu.h:10-41
inline size_t reply_edge_bytes(uint32_t type) { return type == 0 ? 76 : 44; }       // :43-63

}  // namespace

// Two passes.  Pass 1 walks the records once, sequentially, reading only the fields that decide a record's length
// (vertex type; edge type, rows, cols, length of the information diagonal), checks every bound and every type, and
// notes where each chunk of kChunk records starts.  Pass 2 parses the chunks in parallel: all bounds are known to hold.
extern "C" tsgo_wire_graph* tsgo_wire_new(void) { return new tsgo_wire_graph(); }

extern "C" int tsgo_wire_decode(const uint8_t* payload, size_t len, tsgo_wire_graph** out) {
    if (!payload || !out) return tsgo::set_error(-1, "tsgo_wire_decode: null argument");
    *out = nullptr;
    auto* w = new tsgo_wire_graph();
    if (const int rc = tsgo_wire_decode_into(w, payload, len)) { delete w; return rc; }
    *out = w;
    return 0;
}

extern "C" int tsgo_wire_decode_into(tsgo_wire_graph* w, const uint8_t* payload, size_t len) {
    if (!payload || !w) return tsgo::set_error(-1, "tsgo_wire_decode_into: null argument");
    auto fail = [&](const std::string& why) {
        w->v_id.n = w->v_type.n = w->v_pos.n = w->e_type.n = w->e_ids.n = w->e_meas.n = w->e_inf.n = w->fixed.n = 0; w->reply_bytes = 16;
        return tsgo::set_error(-2, "wire decode: " + why);
    };
    size_t off = 0;
    auto need = [&](size_t n) { return len - off >= n; };

    if (!need(4)) return fail("vertex count does not fit the payload");
    const uint32_t nV = rd_u32(payload); off = 4;                       // DeserializeGraph.h:20
    if ((size_t)nV > len / 16) return fail("vertex count does not fit the payload");
    std::vector<size_t> v_start((size_t)nV / kChunk + 2);
    for (uint32_t i = 0; i < nV; ++i) {
        if (i % kChunk == 0) v_start[i / kChunk] = off;
        if (!need(8)) return fail("truncated in vertex " + std::to_string(i));
        const uint32_t type = rd_u32(payload + off + 4);                // :28-31
        if (type > 1) return fail("unknown vertex type " + std::to_string(type));
        const size_t sz = 8 + (type == 0 ? 12 : 8);                     // :32-53
        if (!need(sz)) return fail("truncated in vertex " + std::to_string(i));
        off += sz;
    }
    if (!need(4)) return fail("edge count does not fit the payload");
    const uint32_t nE = rd_u32(payload + off); off += 4;                // :55
    if ((size_t)nE > len / 36) return fail("edge count does not fit the payload");
    std::vector<size_t> e_start((size_t)nE / kChunk + 2);
    for (uint32_t i = 0; i < nE; ++i) {
        if (i % kChunk == 0) e_start[i / kChunk] = off;
        if (!need(20)) return fail("truncated in edge " + std::to_string(i));
        const uint32_t type = rd_u32(payload + off), rows = rd_u32(payload + off + 12), cols = rd_u32(payload + off + 16);   // :61-70
        if (type > 1) return fail("unknown edge type " + std::to_string(type));                          // :93-95
        size_t n_meas;
        if (rows == 0) n_meas = type == 0 ? 3 : 2;                                                       // :74-91
        else {
            if (type != 0) return fail("LM measurement must use the rows==0 form");                      // :112-115
            if (cols > 3) return fail("ODOM measurement wider than 3x3");
            n_meas = (size_t)cols * cols;                                                                // :99-111
        }
        if (!need(20 + 4 * n_meas + 8)) return fail("truncated in edge " + std::to_string(i));
        const uint8_t* q = payload + off + 20 + 4 * n_meas;
        const uint32_t rows_inf = rd_u32(q), cols_inf = rd_u32(q + 4);                                   // :118-121
        if (rows_inf != 0) return fail("only diagonal information matrices are supported");              // :144-147
        if (cols_inf > (type == 0 ? 3u : 2u)) return fail("information diagonal longer than the residual");
        const size_t sz = 20 + 4 * n_meas + 8 + 4 * (size_t)cols_inf;
        if (!need(sz)) return fail("truncated in edge " + std::to_string(i));
        off += sz;
    }
    if (!need(4)) return fail("fixed-vertex count does not fit the payload");
    const uint32_t nF = rd_u32(payload + off); off += 4;                // :163
    if ((size_t)nF > (len - off) / 4) return fail("truncated in the fixed-vertex list");
    const size_t f_start = off;
    static const bool timing = getenv("TSGO_CODEC_TIMING") != nullptr;
    const auto t_scan = std::chrono::steady_clock::now();

    w->v_id.alloc(nV); w->v_type.alloc(nV); w->v_pos.alloc((size_t)nV * 3);
    w->e_type.alloc(nE); w->e_ids.alloc((size_t)nE * 2); w->e_meas.alloc((size_t)nE * 9); w->e_inf.alloc((size_t)nE * 3);
    w->fixed.alloc(nF);
    std::vector<size_t> v_bytes((size_t)nV / kChunk + 1, 0), e_bytes((size_t)nE / kChunk + 1, 0);      // reply bytes per chunk
    const int n_vc = (int)(((size_t)nV + kChunk - 1) / kChunk), n_ec = (int)(((size_t)nE + kChunk - 1) / kChunk);
    tsgo::parallel_chunks(n_vc, [&](int, int cb, int ce) {
        for (int c = cb; c < ce; ++c) {
            const uint8_t* q = payload + v_start[c];
            size_t rb = 0;
            const uint32_t last = (uint32_t)std::min<size_t>((size_t)nV, ((size_t)c + 1) * kChunk);
            for (uint32_t i = (uint32_t)c * kChunk; i < last; ++i) {
                const uint32_t type = rd_u32(q + 4);
                w->v_id[i] = rd_u32(q); w->v_type[i] = type;
                w->v_pos[3 * (size_t)i] = rd_f32(q + 8); w->v_pos[3 * (size_t)i + 1] = rd_f32(q + 12);
                w->v_pos[3 * (size_t)i + 2] = type == 0 ? (double)rd_f32(q + 16) : 0.0;
                q += 8 + (type == 0 ? 12 : 8); rb += reply_vertex_bytes(type);
            }
            v_bytes[c] = rb;
        }
    }, 1);
    tsgo::parallel_chunks(n_ec, [&](int, int cb, int ce) {
        for (int c = cb; c < ce; ++c) {
            const uint8_t* q = payload + e_start[c];
            size_t rb = 0;
            const uint32_t last = (uint32_t)std::min<size_t>((size_t)nE, ((size_t)c + 1) * kChunk);
            for (uint32_t i = (uint32_t)c * kChunk; i < last; ++i) {
                const uint32_t type = rd_u32(q), rows = rd_u32(q + 12), cols = rd_u32(q + 16);
                w->e_type[i] = type; w->e_ids[2 * (size_t)i] = rd_u32(q + 4); w->e_ids[2 * (size_t)i + 1] = rd_u32(q + 8);
                double* m = &w->e_meas[9 * (size_t)i];
                q += 20;
                if (rows == 0) {
                    if (type == 0) {                                                                     // :76-85 (x, y, theta)
                        double full[9];
                        transform_from_xyt(rd_f32(q), rd_f32(q + 4), rd_f32(q + 8), full);
                        for (int k = 0; k < 9; ++k) m[k] = (double)(float)full[k];                       // the reference holds floats (main.cpp:40)
                        q += 12;
                    } else {                                                                             // :86-91 (range, bearing)
                        m[0] = rd_f32(q); m[1] = rd_f32(q + 4);
                        for (int k = 2; k < 9; ++k) m[k] = 0;
                        q += 8;
                    }
                } else {
                    for (int k = 0; k < 9; ++k) m[k] = 0;
                    m[0] = m[4] = m[8] = 1;                                                              // :101 identity
                    for (uint32_t rr = 0; rr < cols; ++rr)
                        for (uint32_t cc = 0; cc < cols; ++cc) { m[rr * 3 + cc] = rd_f32(q); q += 4; }   // :103-110
                }
                const uint32_t cols_inf = rd_u32(q + 4);
                q += 8;
                double* d = &w->e_inf[3 * (size_t)i];
                d[0] = 1; d[1] = 1; d[2] = type == 1 ? 0.0 : 1.0;                                        // :125-131 identity
                for (uint32_t j = 0; j < cols_inf; ++j) { d[j] = rd_f32(q); q += 4; }                    // :138-142
                rb += reply_edge_bytes(type);
            }
            e_bytes[c] = rb;
        }
    }, 1);
    for (uint32_t i = 0; i < nF; ++i) w->fixed[i] = rd_u32(payload + f_start + 4 * (size_t)i);          // :167-172
    size_t total = 4 + 4 + 4 + 4 + 4 * (size_t)nF;
    for (size_t b : v_bytes) total += b;
    for (size_t b : e_bytes) total += b;
    w->reply_bytes = total;
    if (timing) std::fprintf(stderr, "[codec] decode: parse (parallel) %.1f ms after the scan\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scan).count());
    return 0;
}

extern "C" void tsgo_wire_view(const tsgo_wire_graph* w, tsgo_graph* g) {
    g->n_vertices = (int32_t)w->v_id.size(); g->v_id = w->v_id.data(); g->v_type = w->v_type.data(); g->v_pos = w->v_pos.data();
    g->n_edges = (int32_t)w->e_type.size(); g->e_type = w->e_type.data(); g->e_ids = w->e_ids.data();
    g->e_meas = w->e_meas.data(); g->e_inf = w->e_inf.data();
    g->n_fixed = (int32_t)w->fixed.size(); g->fixed = w->fixed.data();
}

extern "C" void tsgo_wire_free(tsgo_wire_graph* w) { delete w; }

// The reply has fixed record sizes per type (vertex 20 / 16 bytes, edge 76 / 44), so its length is known from the decode
// and every chunk of records knows where it starts: the chunks are written in parallel.
extern "C" int64_t tsgo_wire_encode_response(const tsgo_wire_graph* w, const double* v_pos, uint8_t* buf, size_t cap) {
    if (!w || !v_pos) return tsgo::set_error(-1, "tsgo_wire_encode_response: null argument");
    const size_t total = w->reply_bytes;
    if (total - 4 > 0xFFFFFFFFull) return tsgo::set_error(-3, "response larger than 4 GiB");
    if (!buf || cap < total) return (int64_t)total;                     // two-call pattern: the size
    const size_t nV = w->v_id.size(), nE = w->e_type.size(), nF = w->fixed.size();
    const int n_vc = (int)((nV + kChunk - 1) / kChunk), n_ec = (int)((nE + kChunk - 1) / kChunk);
    std::vector<size_t> v_off(n_vc + 1, 0), e_off(n_ec + 1, 0);
    tsgo::parallel_chunks(n_vc, [&](int, int cb, int ce) {
        for (int c = cb; c < ce; ++c) { size_t b = 0; for (size_t i = (size_t)c * kChunk; i < std::min(nV, ((size_t)c + 1) * kChunk); ++i) b += reply_vertex_bytes(w->v_type[i]); v_off[c + 1] = b; }
    }, 1);
    tsgo::parallel_chunks(n_ec, [&](int, int cb, int ce) {
        for (int c = cb; c < ce; ++c) { size_t b = 0; for (size_t i = (size_t)c * kChunk; i < std::min(nE, ((size_t)c + 1) * kChunk); ++i) b += reply_edge_bytes(w->e_type[i]); e_off[c + 1] = b; }
    }, 1);
    wr_u32(buf, (uint32_t)(total - 4));                                 // size prefix (SerializeGraph.h:68-69)
    wr_u32(buf + 4, (uint32_t)nV);                                      // :23-24
    v_off[0] = 8;
    for (int c = 0; c < n_vc; ++c) v_off[c + 1] += v_off[c];
    wr_u32(buf + v_off[n_vc], (uint32_t)nE);                            // :45-46
    e_off[0] = v_off[n_vc] + 4;
    for (int c = 0; c < n_ec; ++c) e_off[c + 1] += e_off[c];
    tsgo::parallel_chunks(n_vc, [&](int, int cb, int ce) {              // SerializeGraphFuncCpu.h:10-41
        for (int c = cb; c < ce; ++c) {
            uint8_t* q = buf + v_off[c];
            for (size_t i = (size_t)c * kChunk; i < std::min(nV, ((size_t)c + 1) * kChunk); ++i) {
                wr_u32(q, w->v_id[i]); wr_u32(q + 4, w->v_type[i]);
                wr_f32(q + 8, (float)v_pos[3 * i]); wr_f32(q + 12, (float)v_pos[3 * i + 1]);
                if (w->v_type[i] == 0) { wr_f32(q + 16, (float)v_pos[3 * i + 2]); q += 20; } else q += 16;
            }
        }
    }, 1);
    tsgo::parallel_chunks(n_ec, [&](int, int cb, int ce) {              // :43-63
        for (int c = cb; c < ce; ++c) {
            uint8_t* q = buf + e_off[c];
            for (size_t e = (size_t)c * kChunk; e < std::min(nE, ((size_t)c + 1) * kChunk); ++e) {
                const double* m = &w->e_meas[9 * e]; const double* d = &w->e_inf[3 * e];
                wr_u32(q, w->e_type[e]); wr_u32(q + 4, w->e_ids[2 * e]); wr_u32(q + 8, w->e_ids[2 * e + 1]);
                if (w->e_type[e] == 0) {
                    wr_u32(q + 12, 3); wr_u32(q + 16, 3); for (int k = 0; k < 9; ++k) wr_f32(q + 20 + 4 * k, (float)m[k]);      // MatrixToByteArray(rows, cols, .., false)
                    wr_u32(q + 56, 0); wr_u32(q + 60, 3); for (int k = 0; k < 3; ++k) wr_f32(q + 64 + 4 * k, (float)d[k]);      // (.., true): 0, rows, diagonal
                    q += 76;
                } else {
                    wr_u32(q + 12, 2); wr_u32(q + 16, 1); wr_f32(q + 20, (float)m[0]); wr_f32(q + 24, (float)m[1]);             // Vector2 = 2 rows x 1 col
                    wr_u32(q + 28, 0); wr_u32(q + 32, 2); wr_f32(q + 36, (float)d[0]); wr_f32(q + 40, (float)d[1]);
                    q += 44;
                }
            }
        }
    }, 1);
    uint8_t* q = buf + e_off[n_ec];
    wr_u32(q, (uint32_t)nF); q += 4;                                    // SerializeGraph.h:59-66
    for (size_t i = 0; i < nF; ++i) { wr_u32(q, w->fixed[i]); q += 4; }
    return (int64_t)total;
}

extern "C" int64_t tsgo_wire_encode_request(const tsgo_graph* g, uint8_t* buf, size_t cap) {
    if (!g) return tsgo::set_error(-1, "tsgo_wire_encode_request: null argument");
    for (int e = 0; e < g->n_edges; ++e)
        if (g->e_type[e] > 1)
            return tsgo::set_error(-2, "tsgo_wire_encode_request: edge " + std::to_string(e) + " has type " + std::to_string(g->e_type[e]) +
                                           "; the wire format carries ODOM (0) and LM (1) edges only (remote/serialization/DeserializeGraph.h:93-95 throws on any other): "
                                           "virtual landmark measurements exist behind the C ABI only");
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
