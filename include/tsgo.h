/* tsgo.h — C ABI of the MI355X-native pose-graph optimizer (libtsgo_hip.so / libtsgo_host.so).
 *
 * This is the drop-in boundary for ToySlam's remote `graph_optimizer` hot path.  The reference has
 * no FFI of its own: its optimizer sits behind the in-process C++ interfaces cited per entry point
 * below (paths relative to the ToySlam tree), fed by the TCP codec.  Every entry point takes plain
 * pointers and sizes, returns an int status (0 = ok, <0 = error, text via tsgo_last_error()), and
 * never throws across the boundary.
 *
 * libtsgo_hip.so  : everything here (device entry points need a gfx950 GPU; they fail loudly
 *                   without one — there is no CPU fallback).
 * libtsgo_host.so : the host-only entry points (codec, synthetic graphs, problem layout), so that
 *                   CPU-only tests can exercise the boundary logic.
 */
#ifndef TSGO_H
#define TSGO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the OptGraph as plain arrays ----------------------------------------------------------------
 * Mirrors the reference graph model: remote/graph/GraphCpu.h:12-59 (AddVertex/AddEdge/FixVertex),
 * vertex/VertexType.h:3-7 (Se2 = 0, Point2 = 1), edge/EdgeType.h:3-7 (Se2 = 0 "ODOM", Se2Point2 = 1
 * "LM"), python twin python/optimizer/opt_graph.py:7-18.
 *   v_pos  : 3 doubles per vertex: (x, y, theta) for Se2, (x, y, 0) for Point2
 *   e_type : 0 ODOM, 1 LM — the two the reference's wire format carries — and 2 = VIRTUAL LANDMARK MEASUREMENT, behind this ABI only
 *            (README.md:53 "Further development: ... Virtual Meas."; the reference keeps a sketch commented out,
 *            python/optimizer/edges2d.py:83-121): two Se2 vertices that observed the same physical point, no landmark vertex;
 *            residual e = T1 p1 - T2 p2 (2), Jacobians [I | dR1/dth p1] and -[I | dR2/dth p2], 2 x 2 diagonal information
 *   e_meas : 9 doubles per edge: ODOM = the 3x3 measurement row-major (EdgeSe2.h); LM = (range,
 *            bearing, 0...) (EdgeSe2Point2d.h:34-35); virtual landmark = (range1, bearing1, range2, bearing2, 0...): the point
 *            as seen from id1 and from id2
 *   e_inf  : 3 doubles per edge: the diagonal of the information matrix (the wire format carries
 *            nothing else, DeserializeGraph.h:123-147); LM uses the first two
 *   fixed  : vertex ids given to FixVertex; a repeated id adds the gauge term once per occurrence
 *            (OptimizerCpu.h:132-138 iterates the vector) */
typedef struct tsgo_graph {
    int32_t n_vertices;
    const uint32_t* v_id;
    const uint32_t* v_type;
    const double* v_pos;
    int32_t n_edges;
    const uint32_t* e_type;
    const uint32_t* e_ids;   /* 2 per edge: id1, id2 */
    const double* e_meas;
    const double* e_inf;
    int32_t n_fixed;
    const uint32_t* fixed;
} tsgo_graph;

/* ---- optimizer -----------------------------------------------------------------------------------*/
typedef struct tsgo_optimizer tsgo_optimizer;

typedef struct tsgo_config {
    int32_t device;          /* HIP device ordinal */
    int32_t precision;       /* 64 (default, parity mode) or 32 */
    double pcg_rel_tol;      /* stop PCG when sqrt(r^T D^-1 r) <= tol * sqrt(b^T D^-1 b), D = the 3x3 block diagonal of the reduced pose
                                system (both preconditioners; the residual norm block-Jacobi PCG measures); default 1e-10 */
    int32_t pcg_max_iters;   /* cap per Gauss-Newton iteration; default 20000 */
    int32_t lanes_per_pose;  /* 0 = auto; 1, 2, 4 or 8 lanes cooperate on one pose row */
    int32_t lanes_per_lm;    /* 0 = auto */
    int32_t use_graphs;      /* how the PCG iterations are launched.  0: kernel by kernel (eager).  1: replayed from a captured hipGraph (from the
                                second tsgo_optimize on a structure on).  2 (default): eager while the host thread enqueues an iteration well
                                within the time the device needs to run it, replay once it has been seen not to.  Eager launches on one device
                                are paced by the first kernel of every iteration, which reports to pinned host memory; edge-sharded runs launch
                                eagerly whatever this says (RCCL calls sit between the kernels).  Same answers, bit for bit. */
    int32_t rank, world;     /* edge sharding: this process owns shard `rank` of `world` (default 0, 1) */
    int32_t verbose;
    int32_t preconditioner;  /* 1 (default): smoothed-aggregation multigrid V-cycle on the reduced pose system; 0: block-Jacobi on its
                                3x3 diagonal.  Edge-sharded runs (world > 1) take the same cycle: every rank builds the hierarchy's
                                patterns from the whole graph (host memory: a full-graph layout + the hierarchy on every rank), the level-0
                                blocks are all-reduced (rank 0 contributes the diagonal), everything below is computed redundantly. */
    int32_t xcd_map;         /* 1: workgroup -> slice map gives each XCD a contiguous eighth of the vertices; 0: round-robin */
    int32_t warm_start;      /* 0: PCG starts from zero.  m >= 1: from a prediction of this solve's pose delta made from the deltas of the last
                                (up to m, at most 6) Gauss-Newton iterations: order 1 is (1 - step) * the previous delta (the un-taken
                                remainder of the last step); order k continues the degree-(k-1) trend of delta_j / (1 - step)^j.  Below the
                                cap the engine takes, solve by solve, the order that would have predicted the previous delta best.
                                Default 6.  Same answer to pcg_rel_tol. */
    int32_t rules;           /* 0 (default): the loop of the C++ server, remote/optimizer/OptimizerCpu.h:80-180 (fixed step 0.2, plateau /
                                short-step / getting-worse stops, b untouched at fixed vertices).  1: the loop of the reference's in-process
                                Python optimizer, python/optimizer/graph_optimizer.py:20-92 — Levenberg-Marquardt-style damping H + lambda I
                                (lambda from 1e-3, x1.1 when chi^2 rose, /1.1 otherwise, within [1e-6, 10]; the `lambdaVal` the C++
                                declares and never uses, OptimizerCpu.h:70), step `lr`, b zeroed at fixed vertices, stop on ||lr dx|| < 1e-3 only. */
    double lr;               /* rules = 1: the step scale `lr` of GraphOptimizer.optimize(iterations, lr) (slam_main.py passes 0.2); ignored by rules = 0 */
    int32_t odom_jacobian;   /* 0 (default): the reference's ODOM Jacobians, the constants A = -I, B = +I (remote/graph/edge/EdgeSe2.h:35-37;
                                parity).  1: the analytic Jacobians of the same residual under the reference's vertex update (SURVEY 8f
                                rank 4; README.md:53 "further development") — same fixed points, and pose graphs with loop closures, which
                                diverge under the constants ("Error is getting worse"), converge. */
    int32_t reuse_structure; /* 1 (default): tsgo_set_graph with the SAME vertex ids/types, edge list and fixed list as the graph the
                                handle already holds only refills estimates, measurements and weights (the reference re-creates
                                everything per message, remote/app/ConnectionHandler.h:18-21); 0: always rebuild.  Same results. */
    int32_t cycle_level0;    /* what the two level-0 products INSIDE the multigrid V-cycle read.  0 (default): the implicit Schur passes over the
                                slot tables (current with every linearisation; in an edge-sharded run each ends in an all-reduce).  1: the explicit
                                level-0 matrix of the hierarchy (replicated on every shard, as old as the hierarchy): no all-reduce inside the
                                cycle, a few per cent more PCG iterations; what bench.py --gpus N (N > 1) runs.  PCG's own product is always the
                                implicit one in `precision`.  Same answers. */
    int32_t cycle_storage;   /* 16 (default) or 32: the copies of the hierarchy's matrices that the V-cycle reads as nine half floats with a common
                                power-of-two exponent per 3x3 block (20 bytes) or as nine f32 (36 bytes).  PCG's own operator and every vector stay
                                in `precision`.  Same answers; a structure whose solves take more than 64 iterations (nearly singular systems) is
                                moved to 32 by the engine (tsgo_stats.cycle_storage_now). */
    int32_t warm_requests;   /* 0 (default): every tsgo_set_graph starts the solver from nothing, so a handle's results are bit-identical to a
                                fresh handle's.  1: the warm start's history (the pose deltas of the last Gauss-Newton iterations) survives
                                tsgo_set_graph — kept for the same structure, carried over by vertex id into a grown one, dropped at the first
                                solve when it does not fit the new estimates — for a front-end that resends the graph with the estimates it was
                                returned (python/slam_main.py:215-238; the reference re-creates everything per message,
                                remote/app/ConnectionHandler.h:18-21).  Same answer to pcg_rel_tol.  What graph_optimizer runs, per connection
                                (tsgo_reset_history). */
} tsgo_config;

enum { TSGO_STOP_CAP = 0, TSGO_STOP_WORSE = 1, TSGO_STOP_PLATEAU = 2, TSGO_STOP_CONVERGED = 3, TSGO_STOP_SOLVER = 4 };

#define TSGO_MAX_TRACE 256
typedef struct tsgo_stats {
    int32_t iterations_run;              /* Gauss-Newton linearisations performed */
    int32_t stop_reason;                 /* TSGO_STOP_*; rules of OptimizerCpu.h:140-153,167-177 */
    double chi2[TSGO_MAX_TRACE];         /* robustified chi^2 at each linearisation (`err`, :117); the first TSGO_MAX_TRACE of them */
    int32_t pcg_iters[TSGO_MAX_TRACE];   /* PCG iterations of each solve */
    double last_delta_norm;              /* ||delta||_2 of the last solve (unscaled, :173) */
    double ms_total, ms_linearize, ms_solve, ms_update;   /* device time, hipEvent */
    double ms_setup;                     /* tsgo_set_graph: host layout build + upload, or the refill when the structure was reused */
    int32_t structure_reused;            /* 1 when the last tsgo_set_graph found the same structure and only refilled values */
    int32_t cycle_storage_now;           /* what the multigrid cycle of this structure reads now: 16 (packed halves) or 32 (f32: chosen by
                                            tsgo_config.cycle_storage, or by the engine after a solve of more than 64 iterations — a graph too
                                            ill-conditioned for 11-bit blocks); 0 with block-Jacobi */
    double lambda_last;                  /* rules = 1: the damping used by the last iteration */
    int64_t n_pose, n_lm, n_odom_edges, n_lm_edges;
    int64_t pcg_iters_total;
    int32_t pcg_fallbacks;               /* solves repeated with block-Jacobi after a multigrid breakdown */
    int32_t trace_len;                   /* entries of chi2[] / pcg_iters[] that are valid: min(iterations_run, TSGO_MAX_TRACE) */
    double chi2_last;                    /* chi^2 of the LAST linearisation (`Summary() error`, OptimizerCpu.h:182), also when the
                                            run is longer than the trace */
    int32_t history_carried;             /* tsgo_config.warm_requests: 1 when this run started from the solver history of the handle's previous
                                            request (kept in place or carried over by vertex id), 2 when it did and the first solve dropped it
                                            (it did not fit the new estimates), 0 otherwise */
    int32_t graph_replay;                /* 1 when this run replayed captured hipGraphs of the PCG iteration (tsgo_config.use_graphs), 0: eager launches */
} tsgo_stats;

/* Fills cfg with defaults. */
void tsgo_default_config(tsgo_config* cfg);

/* GPUs visible to this process (hipGetDeviceCount; 0 when none or on error): tsgo_config.device picks one of them per handle.  The
 * server's DEVICE=all spreads its engine pool over them (host/server.cpp); the reference has one device (GraphManager.h:73-122). */
int tsgo_device_count(void);

/* Replaces CreateOptimizer/CreateSolver/CreateGraph (remote/app/GraphManager.h:73-122): one handle
 * holds the device buffers and is reusable across requests. */
int tsgo_create(const tsgo_config* cfg, tsgo_optimizer** out);
void tsgo_destroy(tsgo_optimizer* opt);

/* Replaces the graph-builder policy Functions::CreateVertex/CreateEdge + graph->AddVertex/AddEdge/
 * FixVertex (remote/serialization/DeserializeGraphFuncCpu.h:14-38, DeserializeGraph.h:43,52,151,172)
 * and GraphGpu::ToDevice (remote/cuda/graph/GraphGpu.h:80-187).  Pointers are borrowed for the call. */
int tsgo_set_graph(tsgo_optimizer* opt, const tsgo_graph* g);

/* tsgo_config.warm_requests only: forget the solver history the handle holds, so that the NEXT tsgo_set_graph starts from nothing —
 * what a pool of handles calls when a handle goes to another client than the one whose requests built that history (the
 * reference creates a fresh optimizer per message, remote/app/ConnectionHandler.h:18-21: nothing of one client ever reaches another). */
void tsgo_reset_history(tsgo_optimizer* opt);

/* Replaces IOptimizer<T>::Optimize(IGraph*) (remote/optimizer/IOptimizer.h:21; loop semantics of
 * OptimizerCpu.h:25-183) including ISolver<T>::Solve (remote/solver/ISolver.h:10).  The graph held by
 * the handle is updated in place. */
int tsgo_optimize(tsgo_optimizer* opt, int32_t iterations, tsgo_stats* stats);

/* Replaces GraphGpu::ToHost (remote/cuda/graph/GraphGpu.h:190-224) / the vertex read-out of
 * SerializeGraphFuncCpu::SerializeVertex (remote/serialization/SerializeGraphFuncCpu.h:10-41):
 * v_pos_out has 3 doubles per vertex in the order of tsgo_graph.v_id; theta = atan2(R10, R00).  A shard (world > 1)
 * writes every pose and the landmarks IT owns; the entries of the other shards' landmarks are left untouched. */
int tsgo_get_vertices(tsgo_optimizer* opt, double* v_pos_out);

/* Parity probes (no reference counterpart; they expose what OptimizerCpu.h:82-138 builds).
 * tsgo_linearize runs one linearisation at the current state and returns, per vertex in tsgo_graph
 * order: diag (9 doubles, the dense diagonal block of H incl. the gauge term, row-major, 2x2 blocks
 * use the leading 2x2) and grad (3 doubles, b = -J^T Omega_w e), plus chi2.
 * tsgo_solve_step additionally solves H delta = b and returns delta (3 doubles per vertex, unscaled)
 * without updating the vertices. */
int tsgo_linearize(tsgo_optimizer* opt, double* diag_out, double* grad_out, double* chi2_out);
int tsgo_solve_step(tsgo_optimizer* opt, double* delta_out, double* chi2_out, int32_t* pcg_iters_out);

/* Multi-GPU (one process per GPU).  Rank 0 calls tsgo_comm_unique_id and ships the 128 bytes to the
 * other ranks (any transport); every rank then calls tsgo_comm_init.  Collectives are RCCL. */
int tsgo_comm_unique_id(uint8_t id_out[128]);
int tsgo_comm_init(tsgo_optimizer* opt, const uint8_t id[128]);
/* One element through the all-reduce the solver uses (rank + 1 from every rank, world (world + 1) / 2 expected back) and the
 * communicator's own rank count (ncclCommCount) in *ranks_out (1 without a communicator).  The first collective is where a
 * missing peer shows — as a hang: callers run it under a watchdog (bench.py). */
int tsgo_comm_selftest(tsgo_optimizer* opt, int32_t* ranks_out);
/* Timing probe used by bench.py on more than one GPU: `reps` back-to-back all-reduces (sum) of n_elements numbers of the handle's
 * precision on the handle's communicator and stream — the call the solver makes after a sharded product (3 P + partials), after a
 * linearisation (18 P + partials) and after the level-0 blocks of a hierarchy build; *us_per_call = hipEvent time / reps.  Every rank
 * of the communicator must call it with the same arguments.  0 microseconds without a communicator. */
int tsgo_comm_time_allreduce(tsgo_optimizer* opt, int64_t n_elements, int32_t reps, double* us_per_call);
/* Timing probe used by bench.py: average device time (hipEvent, microseconds) of `reps` back-to-back
 * launches of one kernel on the handle's stream, and the algorithmic bytes one launch moves.
 * which: 0 schur_lm, 1 schur_pose, 2 cg_update, 3 lin_lm, 4 lin_pose, 5 one whole PCG iteration
 * (preconditioner application included), 6 the multigrid numeric setup of one GN iteration. */
int tsgo_time_kernel(tsgo_optimizer* opt, int32_t which, int32_t reps, double* us_per_launch, double* bytes_per_launch);

/* Timing probe for the multigrid V-cycle's coarse levels (bench.py's per-kernel table): for every explicit level below
 * level 0, the average time of one block-Jacobi smoothing sweep (k_bcsr_residual, hipEvent over `reps` back-to-back
 * launches), its algorithmic bytes (the level's 3x3 blocks + column indices once, three vectors and the diagonal
 * inverse) and how many such sweeps one V-cycle runs on that level.  Returns the number of levels written (<= cap). */
typedef struct tsgo_cycle_level {
    int64_t rows, blocks;            /* block rows / 3x3 blocks of the level's matrix */
    int32_t sweeps_per_cycle;        /* k_bcsr_residual launches per V-cycle on this level (smoothing + residual) */
    int32_t lanes_per_row;
    double us_per_sweep, bytes_per_sweep;
} tsgo_cycle_level;
int tsgo_cycle_probe(tsgo_optimizer* opt, int32_t reps, tsgo_cycle_level* out, int32_t cap);

/* In-situ timing of ONE multigrid-preconditioned PCG iteration, kernel by kernel (bench.py's `roofline`): `reps` iterations are
 * launched eagerly on the handle's stream with the stopping test disabled and a hipEvent recorded before every launch; an entry
 * is one kernel instantiation at one place of the iteration (name = kernel symbol without arguments, `where` = level / role).
 * us = average time from that launch's event to the next one's (the kernel in the cache state the solve leaves it in, plus
 * one kernel boundary); bytes = the algorithmic bytes one such launch moves (the byte models of DESIGN.md section 4).  Returns
 * the number of entries written (<= cap), in launch order; < 0 on error.  Block-Jacobi handles return their three kernels. */
typedef struct tsgo_prof_entry {
    char name[64];
    char where[32];
    int32_t launches_per_iteration;
    int32_t reserved;                    /* 0 */
    double us, bytes;
} tsgo_prof_entry;
int tsgo_profile_iteration(tsgo_optimizer* opt, int32_t reps, tsgo_prof_entry* out, int32_t cap);

const char* tsgo_last_error(void);

/* ---- host-only: wire codec (libtsgo_host.so and libtsgo_hip.so) ---------------------------------
 * Request payload = what python/remote/graph_to_bytes.py:32-67 writes and
 * remote/serialization/DeserializeGraph.h:18-173 reads (WITHOUT the 4-byte length prefix).
 * Response = what remote/serialization/SerializeGraph.h:17-71 + SerializeGraphFuncCpu.h:10-65 write
 * (WITH the u32 length prefix) and python/remote/bytes_to_graph.py:49-108 reads. */
typedef struct tsgo_wire_graph tsgo_wire_graph;
int tsgo_wire_decode(const uint8_t* payload, size_t len, tsgo_wire_graph** out);
/* The same into an existing handle (tsgo_wire_new, or one decoded before): its arrays keep their capacity, so a
 * connection that sends graph after graph (remote/app/ConnectionHandler.h:30-32) does not fault in ~2x the payload of
 * fresh pages per message.  On failure the handle stays valid but holds no graph. */
tsgo_wire_graph* tsgo_wire_new(void);
int tsgo_wire_decode_into(tsgo_wire_graph* w, const uint8_t* payload, size_t len);
void tsgo_wire_view(const tsgo_wire_graph* w, tsgo_graph* view);
/* Encodes the reply for the decoded request with vertex positions replaced by v_pos (3 doubles per
 * vertex, request order).  Two-call pattern: buf = NULL returns the size. */
int64_t tsgo_wire_encode_response(const tsgo_wire_graph* w, const double* v_pos, uint8_t* buf, size_t cap);
/* Encodes a REQUEST (client side; byte-identical to graph_to_bytes for the same OptGraph), prefix
 * included.  Two-call pattern. */
int64_t tsgo_wire_encode_request(const tsgo_graph* g, uint8_t* buf, size_t cap);
void tsgo_wire_free(tsgo_wire_graph* w);

/* ---- host-only: synthetic graphs (BASELINE.json configs 2-5; definition in DESIGN.md) ------------*/
typedef struct tsgo_synth_config {
    int64_t n_poses;
    int32_t lm_per_pose;        /* LM edges per pose (k nearest landmarks in range) */
    double lm_obs_target;       /* aimed observations per landmark (sets landmark density) */
    int32_t loop_closures;      /* extra ODOM edges between revisiting poses (config 5) */
    uint64_t seed;
} tsgo_synth_config;
typedef struct tsgo_synth tsgo_synth;
int tsgo_synth_create(const tsgo_synth_config* cfg, tsgo_synth** out);
void tsgo_synth_view(const tsgo_synth* s, tsgo_graph* view);
/* ground-truth vertex positions (3 doubles per vertex), for convergence checks */
const double* tsgo_synth_truth(const tsgo_synth* s);
void tsgo_synth_free(tsgo_synth* s);

/* ---- host-only: layout probe (tests of the SELL builder and of the shard planner) ---------------*/
typedef struct tsgo_layout_info {
    int64_t n_pose, n_lm_local, n_lm_total, n_lm_edges_local, n_odom_slots;
    int64_t rows_by_pose, rows_by_lm, rows_odom;      /* 64-lane rows incl. padding */
    int32_t lanes_per_pose, lanes_per_lm;
    int64_t lm_first, lm_last;                        /* landmark range owned (in landmark order of the graph) */
    int64_t pose_first, pose_last;                    /* pose range whose ODOM rows/gauge this shard owns */
} tsgo_layout_info;
int tsgo_layout_probe(const tsgo_graph* g, int32_t rank, int32_t world, int32_t lanes_per_pose,
                      int32_t lanes_per_lm, tsgo_layout_info* out);

/* ---- host-only: multigrid hierarchy probe (tests + setup timing) ------------------------------------*/
typedef struct tsgo_amg_info {
    int32_t n_levels;               /* matrices in the hierarchy, the dense coarsest one included */
    int64_t rows[8];                /* block rows per level (level 0 = poses) */
    int64_t blocks[8];              /* 3x3 blocks per level */
    int64_t p_blocks[8];            /* blocks of the prolongator leaving each level */
    int64_t schur_contribs;         /* landmark-pair terms summed into the explicit level-0 matrix */
    double ms_layout, ms_symbolic;  /* host time: slot tables / hierarchy patterns */
    int32_t agg_min[8], agg_max[8]; /* smallest / largest aggregate (in nodes of that level) leaving each level */
    uint64_t checksum;              /* FNV-1a over every slot table, numbering, pattern and gather list, in order: equal checksums
                                     * = the device would be handed the same bytes (the build must not depend on the thread count) */
} tsgo_amg_info;
int tsgo_amg_probe(const tsgo_graph* g, tsgo_amg_info* out);
/* The same for shard `rank` of `world` (edge-sharded runs replicate the hierarchy; only the level-0 contribution lists
 * are per shard): out->schur_contribs = landmark-pair terms THIS shard sums, *odom_contribs_out = its odometry terms.
 * Over all ranks both add up to the unsharded counts. */
int tsgo_amg_probe_shard(const tsgo_graph* g, int32_t rank, int32_t world, tsgo_amg_info* out, int64_t* odom_contribs_out);

#ifdef __cplusplus
}
#endif
#endif /* TSGO_H */
