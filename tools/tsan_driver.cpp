// ThreadSanitizer driver for the host-side builders (tools/sanitize_host.sh): synthetic graph -> slot tables -> multigrid patterns.
#include "../include/tsgo.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main() {
    tsgo_synth_config sc; sc.n_poses = 20000; sc.lm_per_pose = 10; sc.lm_obs_target = 5.0; sc.loop_closures = 100; sc.seed = 3;
    tsgo_synth* sy = nullptr;
    if (tsgo_synth_create(&sc, &sy)) { std::printf("synth: %s\n", tsgo_last_error()); return 1; }
    tsgo_graph g; tsgo_synth_view(sy, &g);
    for (int world = 1; world <= 2; ++world) {
        tsgo_amg_info info; int64_t od = 0;
        if (tsgo_amg_probe_shard(&g, world - 1, world, &info, &od)) { std::printf("probe: %s\n", tsgo_last_error()); return 1; }
        std::printf("world %d: levels %d checksum %llx\n", world, info.n_levels, (unsigned long long)info.checksum);
    }
    // the wire codec's parallel passes: request bytes -> decoded arrays -> reply bytes
    const int64_t n = tsgo_wire_encode_request(&g, nullptr, 0);
    if (n <= 0) { std::printf("encode_request: %s\n", tsgo_last_error()); return 1; }
    std::vector<uint8_t> req((size_t)n);
    tsgo_wire_encode_request(&g, req.data(), req.size());
    tsgo_wire_graph* w = nullptr;
    if (tsgo_wire_decode(req.data() + 4, req.size() - 4, &w))      /* the payload follows the 4-byte length prefix */ { std::printf("decode: %s\n", tsgo_last_error()); return 1; }
    tsgo_graph view; tsgo_wire_view(w, &view);
    const int64_t m = tsgo_wire_encode_response(w, view.v_pos, nullptr, 0);
    std::vector<uint8_t> reply((size_t)m);
    tsgo_wire_encode_response(w, view.v_pos, reply.data(), reply.size());
    std::printf("codec: request %lld bytes, %d vertices / %d edges decoded, reply %lld bytes\n", (long long)n, view.n_vertices, view.n_edges, (long long)m);
    tsgo_wire_free(w);
    tsgo_synth_free(sy);
    return 0;
}
