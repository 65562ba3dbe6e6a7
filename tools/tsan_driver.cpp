// ThreadSanitizer driver for the host-side builders (tools/sanitize_host.sh): synthetic graph -> slot tables -> multigrid patterns.
#include "../include/tsgo.h"
#include <cstdio>
#include <cstdlib>
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
    tsgo_synth_free(sy);
    return 0;
}
