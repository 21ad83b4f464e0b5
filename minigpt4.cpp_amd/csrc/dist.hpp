// Native multi-GPU load: replicas of one node share ONE read of the weight files -- rank 0 loads, the other ranks receive both weight arenas by RCCL broadcast
// over xGMI inside minigpt4_model_load itself, so a plain C client (examples/main.cpp's flow, tests/c/replay_main.c) can be a rank without Python or torch.  Host-only helpers.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace mg4 {

// MINIGPT4_WORLD_SIZE / MINIGPT4_RANK / MINIGPT4_NCCL_ID_FILE (the file through which rank 0 hands the 128-byte ncclUniqueId to the other ranks) / MINIGPT4_DIST_TIMEOUT_S
struct DistEnv {
    int world = 1, rank = 0;
    std::string id_file;       // non-empty: the RCCL path is taken (also with world == 1: a communicator of one rank -- what the single-GPU test drives)
    int timeout_s = 120;
    bool active() const { return !id_file.empty(); }
};
// 0 = fine (out filled); otherwise the message says which variable is wrong.  No environment at all = {1, 0, "", 120}: an ordinary single-GPU load.
int parse_dist_env(DistEnv &out, std::string &err);

// librccl.so by dlopen (the library must not need RCCL when no broadcast is asked for); every call returns 0 or sets `err`
class Rccl {
public:
    ~Rccl();
    int open(std::string &err);
    int unique_id(uint8_t id[128], std::string &err);
    // ncclCommInitRank under a watchdog: a peer that died before it joined (header error, no librccl, its own time-out) must not leave this rank inside RCCL's bootstrap
    // for ever.  After `timeout_s` the call gives up: the helper thread that is still inside ncclCommInitRank is abandoned and this object is deliberately never closed
    // (poisoned()): the load fails, the process lives.
    int init(int world, int rank, const uint8_t id[128], int timeout_s, std::string &err);
    int broadcast(void *device_ptr, size_t bytes, int root, void *stream, std::string &err);   // in place, <= 1 GiB pieces
    int allreduce_max_u64(void *device_words, size_t count, void *stream, std::string &err);   // in place: every rank ends with the element-wise maximum
    void close();
    bool poisoned() const { return poisoned_; }
    // a collective is (or may be) still in flight on a stream after a bounded wait gave up: from now on close() neither destroys the communicator nor unloads the
    // library (ncclCommDestroy on a communicator with a hung collective can hang or corrupt the surviving rank) -- both are leaked, the load fails, the process lives
    void poison() { poisoned_ = true; }
private:
    void *lib_ = nullptr, *comm_ = nullptr;
    void *get_id_ = nullptr, *init_ = nullptr, *bcast_ = nullptr, *allreduce_ = nullptr, *destroy_ = nullptr, *errstr_ = nullptr;
    bool poisoned_ = false;
    std::string why(int rc) const;
};
// rank 0: writes the id (tmp + rename); others: wait for a 128-byte file that is not older than `not_before` (seconds since the epoch; 0 = any): an id a crashed earlier
// job left behind would bootstrap against a dead address.  0 or error text.
int publish_unique_id(const std::string &path, const uint8_t id[128], std::string &err);
int await_unique_id(const std::string &path, uint8_t id[128], int timeout_s, std::string &err, long long not_before = 0);
long long process_start_epoch_s();   // when this process started (from /proc/self/stat; falls back to "now - 1 s")

}  // namespace mg4
