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
#include <thread>

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
int await_unique_id(const std::string &path, uint8_t id[128], int timeout_s, std::string &err) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        struct stat st;
        if (!stat(path.c_str(), &st) && st.st_size == 128) {
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
using fn_destroy = int (*)(void *);
using fn_errstr = const char *(*)(int);

Rccl::~Rccl() { close(); }
std::string Rccl::why(int rc) const { return errstr_ ? std::string(reinterpret_cast<fn_errstr>(errstr_)(rc)) : "ncclResult " + std::to_string(rc); }
int Rccl::open(std::string &err) {
    if (lib_) return 0;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib_ = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib_) break; }
    if (!lib_) { err = std::string("librccl.so not found (dlopen: ") + (dlerror() ? dlerror() : "?") + "): the weight broadcast needs RCCL; there is no fallback"; return 1; }
    get_id_ = dlsym(lib_, "ncclGetUniqueId"); init_ = dlsym(lib_, "ncclCommInitRank"); bcast_ = dlsym(lib_, "ncclBroadcast"); destroy_ = dlsym(lib_, "ncclCommDestroy");
    errstr_ = dlsym(lib_, "ncclGetErrorString");
    if (!get_id_ || !init_ || !bcast_ || !destroy_) { err = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclBroadcast / ncclCommDestroy"; return 1; }
    return 0;
}
int Rccl::unique_id(uint8_t id[128], std::string &err) {
    UniqueId u{};
    const int rc = reinterpret_cast<fn_get_id>(get_id_)(&u);
    if (rc) { err = "ncclGetUniqueId: " + why(rc); return 1; }
    memcpy(id, u.internal, 128);
    return 0;
}
int Rccl::init(int world, int rank, const uint8_t id[128], std::string &err) {
    UniqueId u{}; memcpy(u.internal, id, 128);
    const int rc = reinterpret_cast<fn_init>(init_)(&comm_, world, u, rank);
    if (rc) { comm_ = nullptr; err = "ncclCommInitRank(" + std::to_string(world) + ", rank " + std::to_string(rank) + "): " + why(rc); return 1; }
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
