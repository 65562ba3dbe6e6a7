// server.cpp — `graph_optimizer`: drop-in for ToySlam's remote optimizer process.
//
//   graph_optimizer [HOST=127.0.0.1] [PORT=8888] [ITERATIONS=10] [PIPELINE=cpu] [SOLVER=eigen]
//                   [PRECISION=64] [PCG_TOL=1e-10] [DEVICE=0] [ENGINES=2] [RULES=cpp] [ODOM_JACOBIAN=constant] [WARM_REQUESTS=1]
//   DEVICE: one GPU ("0"), a list ("0,1,2,3") or "all": the engine pool then spans the listed GPUs, ENGINES handles on EACH; a request
//   goes to its connection's last handle when that is idle (it holds the connection's structure and solver history), else to the listed
//   GPU with the fewest requests in flight.  One graph per GPU, no collective: how this server uses a node.
//   RULES "python" or "python:LR": the loop of the reference's in-process Python optimizer instead (lambda * I damping, step LR,
//   default 0.2 as slam_main.py passes); ODOM_JACOBIAN "analytic": the extension of tsgo_config.odom_jacobian.  Both default to
//   what the reference's C++ server does.  WARM_REQUESTS 1: tsgo_config.warm_requests (a connection's next request starts its PCG
//   solves from the history of ITS OWN last one when it gets the same engine handle back; a handle that last served another
//   connection forgets that history first, so no client's iteration counts or low-order bits depend on other clients' traffic;
//   same answers to PCG_TOL), 0: every request starts from nothing.
//
// Positional arguments 1-5 are the reference's (remote/app/main.cpp:12-16, README.md:15-18).  The
// reference maps PIPELINE "cpu" -> CPU optimizer and anything else -> GPU, SOLVER "eigen" -> Eigen and
// anything else -> CUDA, and then FORCES both by what the binary was built with (main.cpp:18-29).  This
// binary is built with the HIP pipeline only, so — exactly like a reference build forces its enums —
// every combination runs the HIP Gauss-Newton pipeline with the implicit-Schur PCG solver, and says
// so in the banner.  There is no CPU path in this program.
//
// Protocol (remote/app/ConnectionHandlerGraph.h:20-52, remote/conn/ConnectionHandlerBase.h:45-128):
//   request  = [i32 size][size bytes], reply = [u32 size][size bytes]; the connection is persistent
//   (ConnectionHandler.h:30-32), many graphs per connection, any number of connections.  Transport is
//   POSIX sockets (the reference uses Boost.Asio, which is not in this image); one thread per
//   connection.  The reference optimises one request at a time; here up to ENGINES requests (from different
//   connections) are in flight on the device at once, each on its own engine handle and HIP stream: at
//   <= 100k poses a solve is a chain of short kernels and two overlap to 1.4x the throughput of one.
#include <arpa/inet.h>
#include <malloc.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <signal.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../../include/tsgo.h"

namespace {

struct BlockTimer {    // same output shape as remote/tools/BlockTimer.cpp:6-18
    std::string caption; unsigned level; std::chrono::steady_clock::time_point t0;
    BlockTimer(const std::string& c, unsigned l = 0) : caption(c), level(l), t0(std::chrono::steady_clock::now()) {}
    ~BlockTimer() {
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        std::cout << std::string(level, ' ') << "[" << caption << "] time: " << ms << "ms" << std::endl;
    }
};

bool read_exact(int fd, void* buf, size_t n) {
    char* p = (char*)buf;
    while (n) {
        const ssize_t r = ::recv(fd, p, n, 0);
        if (r == 0) return false;
        if (r < 0) { if (errno == EINTR) continue; std::cerr << "ReadAsync() error: " << std::strerror(errno) << std::endl; return false; }
        p += r; n -= (size_t)r;
    }
    return true;
}
bool write_all(int fd, const void* buf, size_t n) {
    const char* p = (const char*)buf;
    while (n) {
        const ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL);
        if (r < 0) { if (errno == EINTR) continue; std::cerr << "SendSync() error: " << std::strerror(errno) << std::endl; return false; }
        p += r; n -= (size_t)r;
    }
    return true;
}

struct Server {
    tsgo_config cfg;
    int max_engines = 2;                    // per listed device
    std::vector<int> devices{0};            // the pool's GPUs (DEVICE argument); the same GPU may be listed twice (two pools on it)
    struct Slot { int created = 0, busy = 0; };
    std::vector<Slot> slots;                // per entry of `devices`
    std::vector<std::pair<tsgo_optimizer*, int>> slot_of;      // handle -> entry of `devices` (pool_mutex)
    std::vector<tsgo_optimizer*> idle;      // engine handles not in use
    std::vector<std::pair<tsgo_optimizer*, uint64_t>> last_user;   // per handle: the session it served last (pool_mutex)
    uint64_t next_session = 1;
    int created = 0;
    std::mutex pool_mutex;
    std::condition_variable pool_cv;
    int iterations = 10;
    // connection threads are bounded: past kMaxConnections the acceptor stops accepting (the kernel's listen backlog
    // holds the rest) until one ends.  The reference serves every connection from ONE thread (main.cpp:36-42).
    static constexpr int kMaxConnections = 64;
    int live_connections = 0;
    std::mutex conn_mutex;
    std::condition_variable conn_cv;
    void connection_begin() { std::unique_lock<std::mutex> lock(conn_mutex); conn_cv.wait(lock, [this] { return live_connections < kMaxConnections; }); ++live_connections; }
    void connection_end() { { std::lock_guard<std::mutex> lock(conn_mutex); --live_connections; } conn_cv.notify_all(); }
    void drain() { std::unique_lock<std::mutex> lock(conn_mutex); conn_cv.wait(lock, [this] { return live_connections == 0; }); }

    // An engine handle for one request: the one this connection used last when it is idle (it still holds that
    // connection's graph structure: a repeated structure only refills values, tsgo_config::reuse_structure), else an
    // idle one, else a new one while fewer than max_engines exist, else wait.
    int slot_index(tsgo_optimizer* o) const { for (auto& e : slot_of) if (e.first == o) return e.second; return 0; }
    // on_slot >= 0: a handle of that entry of `devices` only (start-up: every listed GPU is warmed once)
    tsgo_optimizer* acquire(tsgo_optimizer* preferred = nullptr, int on_slot = -1) {
        std::unique_lock<std::mutex> lock(pool_mutex);
        if (slots.size() != devices.size()) slots.resize(devices.size());
        for (;;) {
            if (preferred) {
                auto it = std::find(idle.begin(), idle.end(), preferred);
                if (it != idle.end()) { idle.erase(it); ++slots[slot_index(preferred)].busy; return preferred; }
            }
            // the listed GPU with the fewest requests in flight that can take one more (an idle handle, or room for a new one); among its
            // idle handles the least recently released, which leaves the recently used ones (and their cached structures) to the
            // connections that used them
            int best = -1;
            for (int d = 0; d < (int)slots.size(); ++d) {
                if (on_slot >= 0 && d != on_slot) continue;
                const bool has_idle = std::any_of(idle.begin(), idle.end(), [&](tsgo_optimizer* o) { return slot_index(o) == d; });
                if (!has_idle && slots[d].created >= max_engines) continue;
                if (best < 0 || slots[d].busy < slots[best].busy) best = d;
            }
            if (best >= 0) {
                for (auto it = idle.begin(); it != idle.end(); ++it)
                    if (slot_index(*it) == best) { tsgo_optimizer* o = *it; idle.erase(it); ++slots[best].busy; return o; }
                tsgo_config c = cfg; c.device = devices[best];
                tsgo_optimizer* o = nullptr;
                if (tsgo_create(&c, &o)) return nullptr;
                ++created; ++slots[best].created; ++slots[best].busy;
                slot_of.emplace_back(o, best);
                return o;
            }
            pool_cv.wait(lock);
        }
    }
    void release(tsgo_optimizer* o) {
        { std::lock_guard<std::mutex> lock(pool_mutex); idle.push_back(o); --slots[slot_index(o)].busy; }
        pool_cv.notify_all();
    }
    int slot_index_locked(tsgo_optimizer* o) { std::lock_guard<std::mutex> lock(pool_mutex); return slot_index(o); }
    uint64_t new_session() { std::lock_guard<std::mutex> lock(pool_mutex); return next_session++; }
    // The handle now serves `session`: true when the last request it served was another session's (or nobody's).  The solver
    // history a handle keeps under WARM_REQUESTS belongs to the connection whose requests built it.
    bool changes_hands(tsgo_optimizer* o, uint64_t session) {
        std::lock_guard<std::mutex> lock(pool_mutex);
        for (auto& e : last_user) if (e.first == o) { const bool other = e.second != session; e.second = session; return other; }
        last_user.emplace_back(o, session);
        return true;
    }

    // one request: remote/app/ConnectionHandler.h:14-34
    // What a connection keeps from message to message (the reference re-creates everything per message,
    // ConnectionHandler.h:18-21): the receive buffer, the decoded graph's arrays, the reply buffer — grow-only, so a
    // client that resends a growing graph does not make the server fault in fresh pages every time — and the engine it
    // used last.
    // grow-only byte buffer whose new bytes are NOT zeroed (std::vector::resize would write all 57 MB of a 100k-pose message
    // before recv / the encoder write them again: ~10 ms per direction on a connection's first request)
    struct Bytes {
        std::unique_ptr<uint8_t[]> p; size_t cap = 0, n = 0;
        void resize(size_t want) { if (want > cap) { cap = want + want / 4; p.reset(new uint8_t[cap]); } n = want; }
        void shrink_to(size_t keep) { if (cap > keep) { p.reset(); cap = 0; n = 0; } }
        uint8_t* data() { return p.get(); }
        size_t size() const { return n; }
    };
    struct Session {
        Bytes payload, reply;
        std::vector<double> v_pos;
        tsgo_wire_graph* w = tsgo_wire_new();
        tsgo_optimizer* last_engine = nullptr;
        uint64_t id = 0;
        ~Session() { tsgo_wire_free(w); }
    };

    bool handle(int fd, Session& ss) {
        BlockTimer total{"Total"};
        tsgo_wire_graph* w = ss.w;
        {
            BlockTimer t{"DeserializeGraph"};
            if (tsgo_wire_decode_into(w, ss.payload.data(), ss.payload.size())) { std::cerr << tsgo_last_error() << std::endl; return false; }
        }
        tsgo_graph view; tsgo_wire_view(w, &view);
        std::vector<double>& v_pos = ss.v_pos; v_pos.resize((size_t)view.n_vertices * 3);
        Bytes& reply = ss.reply;
        tsgo_optimizer*& last_engine = ss.last_engine;
        bool ok = true;
        {
            tsgo_optimizer* opt = acquire(last_engine);
            if (!opt) { std::cerr << tsgo_last_error() << std::endl; return false; }
            last_engine = opt;
            const bool fresh_hands = changes_hands(opt, ss.id);
            if (fresh_hands) tsgo_reset_history(opt);      // another connection's deltas must not seed this one's solves
            struct Give { Server& s; tsgo_optimizer* o; ~Give() { s.release(o); } } give{*this, opt};
            BlockTimer t{"OptimizeHIP"};
            tsgo_stats st;
            if (tsgo_set_graph(opt, &view) || tsgo_optimize(opt, iterations, &st) || tsgo_get_vertices(opt, v_pos.data())) {
                std::cerr << tsgo_last_error() << std::endl; ok = false;
            } else {
                if (st.stop_reason == TSGO_STOP_WORSE) std::cout << "Error is getting worse\n";       // OptimizerCpu.h:146
                if (st.stop_reason == TSGO_STOP_PLATEAU) std::cout << "Plateau: NO MORE OPT\n";       // :169
                if (st.stop_reason == TSGO_STOP_CONVERGED) std::cout << "CONVERGED\n";                // :175
                std::cout << "Summary() error = " << st.chi2_last << std::endl;                        // :182
                std::cout << " [hip] iterations=" << st.iterations_run << " pcg_iters=" << st.pcg_iters_total
                          << (st.structure_reused ? " structure=reused refill=" : " structure=built setup=") << st.ms_setup << "ms linearize=" << st.ms_linearize << "ms solve=" << st.ms_solve
                          << "ms update=" << st.ms_update << "ms history=" << st.history_carried << " gpu=" << devices[slot_index_locked(opt)] << " pool=" << slot_index_locked(opt) << std::endl;
            }
        }
        if (ok) {
            BlockTimer t{"SerializeGraph"};
            const int64_t n = tsgo_wire_encode_response(w, v_pos.data(), nullptr, 0);
            if (n < 0) { std::cerr << tsgo_last_error() << std::endl; ok = false; }
            else { reply.resize((size_t)n); tsgo_wire_encode_response(w, v_pos.data(), reply.data(), reply.size()); }
        }
        if (!ok) return false;
        BlockTimer t{"Sending"};
        std::cout << "SendSync() data size = " << reply.size() << std::endl;     // ConnectionHandlerBase.h:118
        return write_all(fd, reply.data(), reply.size());
    }

    // The size prefix is a signed 32-bit int (ConnectionHandlerGraph.h:37-43,57): up to 2 GiB on the wire.  A connection's buffers
    // are grow-only and the process keeps freed pages (mallopt in main), so one oversized or hostile prefix would pin gigabytes for
    // good: messages above max_message_bytes are refused (the connection is closed, like any malformed request), and a session
    // whose buffers are more than 4x its last few requests gives them back.
    size_t max_message_bytes = size_t(1) << 30;
    void connection(int fd) {
        std::cout << "\n------ New Connection ------\n";                         // ConnectionHandler.h:10
        int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        try {                                        // bad_alloc and friends end THIS connection, not the server (the thread is detached)
            Session ss; ss.id = new_session();
            Bytes& payload = ss.payload;
            size_t window_max = 0; int window_n = 0;
            for (;;) {
                int32_t size = 0;                                                // ConnectionHandlerGraph.h:37-43,57
                if (!read_exact(fd, &size, sizeof(size))) break;
                if (size <= 0) { std::cerr << "bad graph size " << size << std::endl; break; }
                if ((size_t)size > max_message_bytes) { std::cerr << "graph size " << size << " exceeds the server's limit of " << max_message_bytes << " bytes (TSGO_MAX_MESSAGE_MB)" << std::endl; break; }
                // every 8 messages: buffers more than 4x the largest of them go back to the allocator
                window_max = std::max(window_max, (size_t)size);
                if (++window_n == 8) {
                    if (payload.cap > 4 * window_max) { payload.shrink_to(0); ss.reply.shrink_to(0); std::vector<double>().swap(ss.v_pos); tsgo_wire_free(ss.w); ss.w = tsgo_wire_new(); }
                    window_max = 0; window_n = 0;
                }
                payload.resize((size_t)size);
                if (!read_exact(fd, payload.data(), payload.size())) break;
                if (!handle(fd, ss)) break;
            }
        } catch (const std::exception& e) {
            std::cerr << "connection error: " << e.what() << std::endl;
        } catch (...) {
            std::cerr << "connection error: unknown exception" << std::endl;
        }
        ::close(fd);
    }
};

}  // namespace

int main(int argc, char* argv[]) {
    signal(SIGPIPE, SIG_IGN);
    // A request builds and drops several hundred MB of host arrays.  Left to its defaults glibc hands blocks of that size
    // back to the kernel on free and faults fresh pages in for the next request (tens of ms per request); this process
    // keeps them: no mmap-backed blocks, no trimming of the heap top.
    mallopt(M_MMAP_MAX, 0);
    mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
    std::cout << "HIP (gfx950) is supported\n";
    try {
        const std::string host = argc < 2 ? "127.0.0.1" : argv[1];
        const std::string port = argc < 3 ? "8888" : argv[2];
        const int iters = argc < 4 ? 10 : std::stoi(argv[3]);
        const std::string targetS = argc < 5 ? "cpu" : argv[4];
        const std::string solverS = argc < 6 ? "eigen" : argv[5];
        const int precision = argc < 7 ? 64 : std::stoi(argv[6]);
        const double tol = argc < 8 ? 1e-10 : std::stod(argv[7]);
        const std::string deviceS = argc < 9 ? "0" : argv[8];
        std::vector<int> devices;
        if (deviceS == "all") {
            const int n = tsgo_device_count();
            if (n <= 0) { std::cerr << "ConnectionManager error: DEVICE=all, but no GPU is visible" << std::endl; return 1; }
            for (int d = 0; d < n; ++d) devices.push_back(d);
        } else {
            for (size_t b = 0; b <= deviceS.size();) {
                const size_t e = std::min(deviceS.find(',', b), deviceS.size());
                devices.push_back(std::stoi(deviceS.substr(b, e - b)));
                b = e + 1;
            }
        }
        const int device = devices[0];
        const int engines = argc < 10 ? 2 : std::max(1, std::stoi(argv[9]));
        const std::string rulesS = argc < 11 ? "cpp" : argv[10];
        const std::string odomS = argc < 12 ? "constant" : argv[11];
        const int warm_requests = argc < 13 ? 1 : std::stoi(argv[12]);
        // the reference prints the enums after forcing them to what the build supports (main.cpp:21-34):
        // 0 = EIGEN, 1 = CUDA; here both are always the accelerator pipeline.
        std::cout << "iters: " << iters << ", optimizerType: 1, solverType: 1" << std::endl;
        if (targetS == "cpu" || solverS == "eigen")
            std::cout << "note: PIPELINE=" << targetS << " SOLVER=" << solverS
                      << " requested; this build has the HIP pipeline only (gfx950, implicit-Schur PCG, f" << precision << ")\n";

        tsgo_config cfg; tsgo_default_config(&cfg);
        cfg.device = device; cfg.precision = precision; cfg.pcg_rel_tol = tol;
        if (rulesS.rfind("python", 0) == 0) {
            cfg.rules = 1;
            const size_t colon = rulesS.find(':');
            if (colon != std::string::npos) cfg.lr = std::stod(rulesS.substr(colon + 1));
            std::cout << "rules: python/optimizer/graph_optimizer.py (lambda * I, lr " << cfg.lr << ")\n";
        }
        if (odomS == "analytic") { cfg.odom_jacobian = 1; std::cout << "ODOM Jacobians: analytic (extension)\n"; }
        cfg.warm_requests = warm_requests ? 1 : 0;      // a connection's next request continues from the last one's solver history (tsgo.h)
        Server srv; srv.iterations = iters; srv.cfg = cfg; srv.max_engines = engines; srv.devices = devices;
        if (devices.size() > 1) {
            std::cout << "engine pool over " << devices.size() << " GPUs (";
            for (size_t k = 0; k < devices.size(); ++k) std::cout << (k ? "," : "") << devices[k];
            std::cout << "), " << engines << " handle(s) each\n";
        }
        if (const char* e = getenv("TSGO_MAX_MESSAGE_MB")) srv.max_message_bytes = (size_t)std::max(1, atoi(e)) << 20;
        for (int slot = 0; slot < (int)devices.size(); ++slot) {   // fail at start-up, like the reference, when the pipeline cannot be created at all (on any listed GPU)
            tsgo_optimizer* first = srv.acquire(nullptr, slot);
            if (!first) { std::cerr << "ConnectionManager error: " << tsgo_last_error() << std::endl; return 1; }
            {   // one small solve before the first client: loads every kernel's code object and sizes the runtime's pools,
                // so that the first request does not pay for it (the code objects are per process, not per engine)
                BlockTimer t{"WarmUp"};
                // TSGO_WARMUP_POSES: a front-end that knows the size of its graphs lets the warm-up run at that size, so that the device slabs,
                // the pinned staging and the runtime's pools have their final size before the first request (a 100k-pose first request otherwise
                // spends ~30 ms growing them)
                int warm_poses = 2000;
                if (const char* e = getenv("TSGO_WARMUP_POSES")) warm_poses = std::max(100, std::min(2000000, atoi(e)));
                tsgo_synth_config sc; sc.n_poses = warm_poses; sc.lm_per_pose = 8; sc.lm_obs_target = 5.0; sc.loop_closures = 0; sc.seed = 1;
                tsgo_synth* sy = nullptr;
                if (tsgo_synth_create(&sc, &sy) == 0) {
                    tsgo_graph view; tsgo_synth_view(sy, &view);
                    tsgo_stats st; std::vector<double> v((size_t)view.n_vertices * 3);
                    if (tsgo_set_graph(first, &view) || tsgo_optimize(first, 2, &st) || tsgo_get_vertices(first, v.data()))
                        std::cerr << "warm-up: " << tsgo_last_error() << std::endl;
                    tsgo_synth_free(sy);
                }
            }
            srv.release(first);
        }

        addrinfo hints{}; hints.ai_family = AF_UNSPEC; hints.ai_socktype = SOCK_STREAM; hints.ai_flags = AI_PASSIVE;
        addrinfo* res = nullptr;
        const int gai = getaddrinfo(host.c_str(), port.c_str(), &hints, &res);
        if (gai != 0 || !res) { std::cerr << "ConnectionManager error: resolve: " << gai_strerror(gai) << std::endl; return 1; }
        std::cout << "Endpoints:\n";                                             // ConnectionManagerServer.h:27-31
        for (addrinfo* a = res; a; a = a->ai_next) {
            char h[NI_MAXHOST], s[NI_MAXSERV];
            if (!getnameinfo(a->ai_addr, a->ai_addrlen, h, sizeof(h), s, sizeof(s), NI_NUMERICHOST | NI_NUMERICSERV))
                std::cout << " - " << h << ":" << s << std::endl;
        }
        const int lfd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);   // first endpoint (:33)
        if (lfd < 0) { std::cerr << "ConnectionManager error: socket: " << std::strerror(errno) << std::endl; return 1; }
        int one = 1; setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));      // :35
        if (::bind(lfd, res->ai_addr, res->ai_addrlen) || ::listen(lfd, 16)) {
            std::cerr << "ConnectionManager error: bind/listen: " << std::strerror(errno) << std::endl; return 1;
        }
        freeaddrinfo(res);
        std::cout << "listening" << std::endl;
        for (;;) {                                                               // ConnectionManagerServer.h:46-61
            srv.connection_begin();                                              // waits while kMaxConnections are being served
            const int fd = ::accept(lfd, nullptr, nullptr);
            if (fd < 0) {
                srv.connection_end();
                if (errno == EINTR) continue;
                std::cerr << "accept: " << std::strerror(errno) << std::endl; break;
            }
            std::thread([&srv, fd] { srv.connection(fd); srv.connection_end(); }).detach();
        }
        srv.drain();                                                             // the acceptor failed: let the requests in flight finish
        for (tsgo_optimizer* o : srv.idle) tsgo_destroy(o);
    } catch (std::exception& e) {
        std::cerr << "ConnectionManager error: " << e.what() << std::endl;       // main.cpp:44-47
    }
    return 0;
}
