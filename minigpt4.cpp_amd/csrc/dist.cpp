// Host side of the native weight broadcast (dist.hpp).  No device code here; Engine::init drives it.
#include "dist.hpp"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <ctime>
#include <memory>
#include <mutex>
#include <thread>

#include <hip/hip_runtime_api.h>

namespace mg4 {

static bool parse_int(const char *s, int &v) {
    if (!s || !*s) return false;
    char *end = nullptr;
    const long x = strtol(s, &end, 10);
    if (*end || x < -1000000 || x > 1000000) return false;
    v = (int)x;
    return true;
}
int parse_dist_env(DistEnv &out, std::string &err) {
    out = DistEnv{};
    const char *ws = getenv("MINIGPT4_WORLD_SIZE"), *rk = getenv("MINIGPT4_RANK"), *idf = getenv("MINIGPT4_NCCL_ID_FILE"), *to = getenv("MINIGPT4_DIST_TIMEOUT_S");
    if (ws && !parse_int(ws, out.world)) { err = "MINIGPT4_WORLD_SIZE is not an integer"; return 1; }
    if (rk && !parse_int(rk, out.rank)) { err = "MINIGPT4_RANK is not an integer"; return 1; }
    if (to && (!parse_int(to, out.timeout_s) || out.timeout_s <= 0)) { err = "MINIGPT4_DIST_TIMEOUT_S must be a positive integer"; return 1; }
    if (out.world < 1 || out.world > 1024) { err = "MINIGPT4_WORLD_SIZE must be in 1 .. 1024"; return 1; }
    if (out.rank < 0 || out.rank >= out.world) { err = "MINIGPT4_RANK must be in 0 .. MINIGPT4_WORLD_SIZE - 1"; return 1; }
    if (idf && *idf) out.id_file = idf;
    if (out.world > 1 && out.id_file.empty()) { err = "MINIGPT4_WORLD_SIZE > 1 needs MINIGPT4_NCCL_ID_FILE (the file through which rank 0 hands the ncclUniqueId to the other ranks)"; return 1; }
    return 0;
}

int publish_unique_id(const std::string &path, const uint8_t id[128], std::string &err) {
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, 128, f) != 128) { if (f) fclose(f); err = "cannot write " + tmp; return 1; }
    fclose(f);
    if (rename(tmp.c_str(), path.c_str())) { err = "cannot rename " + tmp + " to " + path; return 1; }
    return 0;
}
long long process_start_epoch_s() {
    // field 22 of /proc/self/stat = start time in clock ticks since boot; boot time = now - uptime
    long long fallback = (long long)time(nullptr) - 1;
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return fallback;
    char buf[2048]; const size_t n = fread(buf, 1, sizeof buf - 1, f); fclose(f); buf[n] = 0;
    const char *p = strrchr(buf, ')');                       // the command name may contain spaces / parentheses: fields are counted after the LAST ')'
    if (!p) return fallback;
    unsigned long long ticks = 0; int field = 2;
    for (p++; *p && field < 22; p++) if (*p == ' ') field++;
    if (field != 22 || sscanf(p, "%llu", &ticks) != 1) return fallback;
    double up = 0.0; FILE *u = fopen("/proc/uptime", "r");
    if (!u) return fallback;
    const int got = fscanf(u, "%lf", &up); fclose(u);
    if (got != 1) return fallback;
    const long hz = sysconf(_SC_CLK_TCK) > 0 ? sysconf(_SC_CLK_TCK) : 100;
    return (long long)((double)time(nullptr) - up + (double)ticks / (double)hz);
}
int await_unique_id(const std::string &path, uint8_t id[128], int timeout_s, std::string &err, long long not_before) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        struct stat st;
        if (!stat(path.c_str(), &st) && st.st_size == 128 && (not_before <= 0 || (long long)st.st_mtime >= not_before - 2)) {   // (2 s of slack for coarse file times)
            FILE *f = fopen(path.c_str(), "rb");
            if (f) { const size_t n = fread(id, 1, 128, f); fclose(f); if (n == 128) return 0; }
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) { err = "no ncclUniqueId appeared in " + path + " within " + std::to_string(timeout_s) + " s (is rank 0 running?)"; return 1; }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}

// the five RCCL entry points, by their C signatures (rccl.h: ncclUniqueId = 128 opaque bytes passed BY VALUE, ncclUint8 = 1)
struct UniqueId { char internal[128]; };
using fn_get_id = int (*)(UniqueId *);
using fn_init = int (*)(void **, int, UniqueId, int);
using fn_bcast = int (*)(const void *, void *, size_t, int, int, void *, void *);
using fn_allreduce = int (*)(const void *, void *, size_t, int, int, void *, void *);   // (send, recv, count, datatype, op, comm, stream)
using fn_destroy = int (*)(void *);
using fn_errstr = const char *(*)(int);

Rccl::~Rccl() { close(); }
std::string Rccl::why(int rc) const { return errstr_ ? std::string(reinterpret_cast<fn_errstr>(errstr_)(rc)) : "ncclResult " + std::to_string(rc); }
int Rccl::open(std::string &err) {
    if (lib_) return 0;
    // A process that already maps an RCCL (a Python rank: torch ships its own torch/lib/librccl.so) must not get a SECOND copy with its own bootstrap state next to it: take
    // the mapped one when there is one (RTLD_NOLOAD finds it by soname), load the system library only otherwise (the C-client path this code exists for).
    for (const char *name : {"librccl.so.1", "librccl.so"}) { lib_ = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); if (lib_) break; }
    if (!lib_) for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib_ = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib_) break; }
    if (!lib_) { err = std::string("librccl.so not found (dlopen: ") + (dlerror() ? dlerror() : "?") + "): the weight broadcast needs RCCL; there is no fallback"; return 1; }
    get_id_ = dlsym(lib_, "ncclGetUniqueId"); init_ = dlsym(lib_, "ncclCommInitRank"); bcast_ = dlsym(lib_, "ncclBroadcast"); destroy_ = dlsym(lib_, "ncclCommDestroy");
    errstr_ = dlsym(lib_, "ncclGetErrorString"); allreduce_ = dlsym(lib_, "ncclAllReduce");
    if (!get_id_ || !init_ || !bcast_ || !destroy_ || !allreduce_) { err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclBroadcast / ncclAllReduce / ncclCommDestroy"; return 1; }
    return 0;
}
int Rccl::unique_id(uint8_t id[128], std::string &err) {
    UniqueId u{};
    const int rc = reinterpret_cast<fn_get_id>(get_id_)(&u);
    if (rc) { err = "ncclGetUniqueId: " + why(rc); return 1; }
    memcpy(id, u.internal, 128);
    return 0;
}
int Rccl::init(int world, int rank, const uint8_t id[128], int timeout_s, std::string &err) {
    // the blocking ncclCommInitRank on a helper thread; this thread waits for it at most timeout_s seconds.  State shared with the helper lives on the heap and is owned by
    // whoever finishes last, so an abandoned helper never touches freed memory.
    struct Shared { std::mutex m; std::condition_variable cv; bool done = false; int rc = 0; void *comm = nullptr; };
    auto sh = std::make_shared<Shared>();
    UniqueId u{}; memcpy(u.internal, id, 128);
    const fn_init f = reinterpret_cast<fn_init>(init_);
    int dev = 0; (void)hipGetDevice(&dev);
    std::thread([sh, f, world, u, rank, dev]() {
        (void)hipSetDevice(dev);                               // the device is a per-thread setting; the communicator binds to the calling thread's device
        void *c = nullptr;
        const int rc = f(&c, world, u, rank);
        std::lock_guard<std::mutex> g(sh->m); sh->rc = rc; sh->comm = c; sh->done = true; sh->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(sh->m);
    if (!sh->cv.wait_for(lk, std::chrono::seconds(std::max(1, timeout_s)), [&] { return sh->done; })) {
        poisoned_ = true;                                      // the helper is still inside RCCL: never dlclose the library, never destroy a half-built communicator
        err = "ncclCommInitRank(" + std::to_string(world) + ", rank " + std::to_string(rank) + ") did not complete within " + std::to_string(timeout_s) + " s: a peer never joined (did it fail to load?)";
        return 1;
    }
    if (sh->rc) { comm_ = nullptr; err = "ncclCommInitRank(" + std::to_string(world) + ", rank " + std::to_string(rank) + "): " + why(sh->rc); return 1; }
    comm_ = sh->comm;
    return 0;
}
int Rccl::allreduce_max_u64(void *p, size_t count, void *stream, std::string &err) {
    const int rc = reinterpret_cast<fn_allreduce>(allreduce_)(p, p, count, /*ncclUint64*/ 5, /*ncclMax*/ 2, comm_, stream);
    if (rc) { err = "ncclAllReduce of " + std::to_string(count) + " words: " + why(rc); return 1; }
    return 0;
}
int Rccl::broadcast(void *p, size_t bytes, int root, void *stream, std::string &err) {
    // few large collectives: a ring broadcast over point-to-point xGMI is bound by one link (~153 GB/s), not by message count
    for (size_t off = 0; off < bytes; off += (size_t)1 << 30) {
        const size_t n = std::min(bytes - off, (size_t)1 << 30);
        char *q = static_cast<char *>(p) + off;
        const int rc = reinterpret_cast<fn_bcast>(bcast_)(q, q, n, /*ncclUint8*/ 1, root, comm_, stream);
        if (rc) { err = "ncclBroadcast of " + std::to_string(n) + " bytes: " + why(rc); return 1; }
    }
    return 0;
}
void Rccl::close() {
    if (poisoned_) { comm_ = nullptr; lib_ = nullptr; return; }   // an abandoned ncclCommInitRank is still running inside the library: leave both alone
    if (comm_ && destroy_) reinterpret_cast<fn_destroy>(destroy_)(comm_);
    comm_ = nullptr;
    if (lib_) dlclose(lib_);
    lib_ = nullptr;
}

// host-only view of the environment for the CPU test tier (include/minigpt4_amd_test.h)
int parse_dist_env_for_test(int *world, int *rank, char *id_file, size_t cap, char *err, size_t err_cap) {
    DistEnv d; std::string e;
    const int rc = parse_dist_env(d, e);
    if (world) *world = d.world;
    if (rank) *rank = d.rank;
    if (id_file && cap) { strncpy(id_file, d.id_file.c_str(), cap - 1); id_file[cap - 1] = 0; }
    if (err && err_cap) { strncpy(err, e.c_str(), err_cap - 1); err[err_cap - 1] = 0; }
    return rc;
}

}  // namespace mg4
