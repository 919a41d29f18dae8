// lf_common.h -- internal declarations shared by the translation units of liblisflood_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <string>
#include <map>
#include <vector>

#include "../../include/lisflood_amd.h"

#define LF_NEWTON_TOL 1e-12 /* kinematic_wave_parallel_tools.py:26 */
#define LF_MAX_ITERS 3000   /* kinematic_wave_parallel_tools.py:27 */

int lf_set_error(int code, const char *fmt, ...);

#define LF_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return lf_set_error(LF_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                                __LINE__);                                                                \
    } while (0)

#define LF_TRY(call)          \
    do {                      \
        int rc_ = (call);     \
        if (rc_ != LF_OK) return rc_; \
    } while (0)

// Per-device context: one compute stream, a stopwatch event pair.
struct lf_device_ctx {
    bool ready = false;
    hipStream_t stream = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // staging arena of the *_host entry points (lf_soil.hip): one grow-only device allocation carved up per call
    // instead of ~70 hipMalloc / hipFree pairs; a call that needs more gets overflow buffers and the arena grows
    // before the next one
    void *stage_base = nullptr;
    size_t stage_bytes = 0, stage_need = 0;
    void *soil_ws = nullptr; // lists and straggler records of the soil call (lf_soil.hip: soil_ws_layout)
    size_t soil_ws_bytes = 0, soil_ntiles = 0;
    // upload stream for double-buffered inputs (lf_upload_*): copies of the NEXT step's forcing overlap the kernels of
    // the current one; per buffer set an event "copy finished" and an event "last kernel reading the set finished"
    hipStream_t copy_stream = nullptr;
    std::map<void *, std::pair<void *, size_t>> f32_stage; // lf_upload_copy_f32: fp32 staging buffer per destination vector
    hipEvent_t copied[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    bool consumed_valid[2] = {false, false}, copied_valid[2] = {false, false};
    // page-locked staging ring of lf_memcpy_h2d_staged: small per-step uploads that must not stall the host
    struct staged_slot {
        void *host = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
        bool in_flight = false;
    };
    staged_slot staged[8];
    int staged_next = 0;
    // side stream (lf_side_stream_*): a part of a step that the NEXT step's first kernels do not depend on -- the channel
    // wavefront of a model step beside the canopy / soil / overland kernels of the step after it (hotpath.py)
    hipStream_t side_stream = nullptr, main_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_side_done = nullptr;
    bool side_active = false, side_pending = false;
    // lanes (lf_lane_*): streams for independent work items of one device (loopback blocks of the partition)
    std::vector<hipStream_t> lanes;
    std::vector<hipEvent_t> lane_done;
    std::vector<bool> lane_used;
    hipEvent_t lane_fork = nullptr;
    hipStream_t lane_main = nullptr;
    int lane_current = 0;
    bool lane_forked = false;
};
int lf_ctx(int device, lf_device_ctx **out); // makes `device` current, creates the context on first use

// Host-side graph.  Positions ("sweep order") are the engine's internal cell numbering: levels
// ascending (level = max distance to outlet - distance, as kinematic_wave_parallel.py:148-149),
// breadth-first from the outlets inside a level, which makes the upstream cells of position p the
// contiguous positions [ups_ptr[p], ups_ptr[p+1]) in ascending pixel id.
struct lf_graph {
    int H = 0, W = 0;
    int64_t N = 0, NL = 0;
    int K = 1;
    std::vector<int32_t> down;        // [N] downstream pixel id, -1 = none
    std::vector<int32_t> perm;        // [N] position -> pixel
    std::vector<int32_t> ups_ptr;     // [N+1]
    std::vector<int64_t> level_start; // [NL+1]
    // zero-length structure links (lf_graph_create_ex): flag per POSITION, 1 = this pit hangs on a structure cell of
    // its own level.  Such cells sit at the end of their level outside every upstream range, so the LAST cell of
    // the next level finds them behind its own children: only sweeps that write a separate router-output buffer
    // (the fused sub-step wavefront, which stores 0 for them) may run on such a graph.
    std::vector<uint8_t> linked;
    bool has_links = false;
    uint64_t serial = 0; // identity of this graph object (unique per process): routers swept together must share it
    lf_graph() = default;
    lf_graph(const lf_graph &) = delete;
    lf_graph &operator=(const lf_graph &) = delete;
};

template <typename T>
struct lf_dbuf { // owning device buffer
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count)
    {
        release();
        n = count;
        if (count == 0) return LF_OK;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return lf_set_error(LF_E_HIP, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        }
        return LF_OK;
    }
    // `stream`: the stream the kernels that read the buffer run on.  The library's compute stream is non-blocking, i.e. it
    // does not synchronise with the legacy stream a plain hipMemcpy uses, and a pageable H2D copy may return once the
    // source is staged: the copy is therefore enqueued on the consumer's own stream and waited for there.
    int upload(const T *src, size_t count, hipStream_t stream = nullptr)
    {
        LF_TRY(alloc(count));
        if (count) {
            LF_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, stream));
            LF_HIP(hipStreamSynchronize(stream));
        }
        return LF_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~lf_dbuf() { release(); }
    lf_dbuf() = default;
    lf_dbuf(const lf_dbuf &) = delete;
    lf_dbuf &operator=(const lf_dbuf &) = delete;
};

// The [nsteps][N] history of the time-major fused form (k_fused_level_steps) is OPTIONAL working storage: without it the
// skewed wavefront runs.  It can be tens of GB (10000^2, 24 split sub-steps: 2 x 19 GB), so it is taken only inside a
// budget -- LF_FUSED_TIME_MAJOR_MAX_BYTES if set, otherwise half of what hipMemGetInfo reports free (counting what the
// two buffers already hold) -- and a refused size is remembered, so a router that does not fit never pays a multi-GB
// hipMalloc / failed hipMalloc / synchronising hipFree per call.  -> true: both buffers hold `count` doubles (h2 only
// if `two`).
inline bool lf_history_ensure(lf_dbuf<double> &h1, lf_dbuf<double> &h2, size_t &refused_bytes, size_t count, bool two)
{
    const bool ok1 = h1.p && h1.n >= count, ok2 = !two || (h2.p && h2.n >= count);
    if (ok1 && ok2) return true;
    const size_t total = count * sizeof(double) * (two ? 2 : 1);
    if (total >= refused_bytes) return false;
    size_t budget = 0;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
            (void)hipGetLastError();
            refused_bytes = total;
            return false;
        }
        const size_t held = (h1.p ? h1.n : 0) * sizeof(double) + (h2.p ? h2.n : 0) * sizeof(double);
        budget = (free_b + held) / 2;
        static const long long cap = [] {
            const char *e = std::getenv("LF_FUSED_TIME_MAJOR_MAX_BYTES");
            return e ? std::atoll(e) : -1ll;
        }();
        if (cap >= 0) budget = std::min((size_t)cap, free_b + held);
    }
    if (total > budget) {
        refused_bytes = total;
        return false;
    }
    bool ok = true;
    if (!ok1) ok = h1.alloc(count) == LF_OK;
    if (ok && !ok2) ok = h2.alloc(count) == LF_OK;
    if (!ok) {
        (void)hipGetLastError();
        h1.release();
        h2.release();
        refused_bytes = total;
    }
    return ok;
}
