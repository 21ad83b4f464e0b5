// Helpers shared by the two extern "C" translation units: api.cpp (the product boundary) and test_hooks.cpp (libminigpt4_test.so only).
#pragma once
#include <sys/stat.h>

#include <cstdio>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "engine.hpp"
#include "minigpt4.h"

namespace mg4 {
namespace apiutil {

inline Engine *E_(MiniGPT4Context *c) { return reinterpret_cast<Engine *>(c); }
inline bool file_exists(const char *p) { struct stat st; return p && stat(p, &st) == 0; }

// The HIP runtime keeps the last failed call of the thread as a sticky "last error" that the NEXT user of hipGetLastError sees.  Entry: whatever another HIP user of the
// process (PyTorch next to this library) left there is not ours.  Exit: kernel launches have no return value -- a refused launch (bad configuration, too much LDS) only
// shows up in that slot -- so a leftover error fails the call loudly instead of letting generation continue on stale activations, and the slot is left clean for the
// host application's own launch checks (PyTorch once raised "HIP error: invalid argument" on its first kernel after this library had ignored a refused attribute hint;
// ignorable calls now go through HIP_IGNORE, which clears the slot itself).
struct ClearStickyHipError { ~ClearStickyHipError() { (void)hipGetLastError(); } };
template <typename F> inline int guarded(int on_hip_error, F &&f) {
    (void)hipGetLastError();
    ClearStickyHipError clear_on_exit;
    try {
        const int rc = f();
        const hipError_t left = hipGetLastError();
        // host-only entry points also run where there is no device: the runtime then answers every query with "no device" -- nothing was launched, nothing to report
        if (left != hipSuccess && left != hipErrorNoDevice && left != hipErrorInsufficientDriver && left != hipErrorNotInitialized) {
            char buf[256]; snprintf(buf, sizeof(buf), "HIP error %d (%s) left behind by a kernel launch", (int)left, hipGetErrorString(left));
            set_last_error(buf); MG4_ERR("%s", buf);
            return rc ? rc : on_hip_error;
        }
        return rc;
    }
    catch (const HipError &e) {
        char buf[512]; snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d: %s", (int)e.code, hipGetErrorString(e.code), e.file, e.line, e.what);
        set_last_error(buf); MG4_ERR("%s", buf); return on_hip_error;
    } catch (const std::bad_alloc &) { set_last_error("out of host memory"); return on_hip_error; }
    // nothing may unwind through the extern "C" boundary into a ctypes / C caller (a malformed file can make a container throw std::length_error etc.)
    catch (const std::exception &e) { set_last_error(std::string("internal error: ") + e.what()); MG4_ERR("%s", last_error().c_str()); return on_hip_error; }
    catch (...) { set_last_error("internal error: unknown exception"); MG4_ERR("%s", last_error().c_str()); return on_hip_error; }
}

struct DevBuf { void *p = nullptr; DevBuf(size_t n) { HIP_CHECK(hipMalloc(&p, n ? n : 1)); } ~DevBuf() { if (p) HIP_IGNORE(hipFree(p)); } template <typename T> T *as() { return static_cast<T *>(p); } };
inline void alloc_act(ActQ &A, std::vector<std::unique_ptr<DevBuf>> &keep, size_t N, size_t K) {
    auto mk = [&](size_t bytes) { keep.emplace_back(new DevBuf(bytes + 256)); return keep.back()->p; };
    A.q8k = (int8_t *)mk(N * K); A.q80 = (int8_t *)mk(N * K); A.dk = (float *)mk(N * (K / 256 + 1) * 4); A.bsk = (int16_t *)mk(N * (K / 16 + 1) * 2); A.bsq = (int8_t *)mk(N * (K / 16 + 16)); A.q16 = (__half *)mk(N * K * 2); A.bs16 = (__half *)mk(N * (K / 16 + 16) * 2);
    A.d0 = (float *)mk(N * (K / 32 + 1) * 4); A.d1 = (float *)mk(N * (K / 32 + 1) * 4); A.s1 = (float *)mk(N * (K / 32 + 1) * 4); A.sum0 = (int *)mk(N * (K / 32 + 1) * 4);
    A.xh = (__half *)mk(N * K * 2); A.xf = (float *)mk(N * K * 4);
}

}  // namespace apiutil
}  // namespace mg4
