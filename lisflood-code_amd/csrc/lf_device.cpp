// lf_device.cpp -- error reporting, per-device context (stream + stopwatch), memory plumbing.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "lf_common.h"

namespace {
thread_local char g_err[512] = "";
std::mutex g_ctx_mutex;
lf_device_ctx g_ctx[64];
// lf_device_alloc skews the start of successive allocations by a few KiB: kernels that stream dozens of equally
// sized vectors in lock-step (the soil kernel reads ~67) would otherwise present every stream at the same offset
// of a 2 MiB-aligned buffer, i.e. on the same HBM channel at the same time.  user pointer -> hipMalloc base.
std::unordered_map<void *, void *> g_alloc_base;
size_t g_alloc_counter = 0;
constexpr size_t kSkewStep = 4096 + 256, kSkewSlots = 61;
} // namespace

int lf_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int lf_ctx(int device, lf_device_ctx **out)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return lf_set_error(LF_E_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                            e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count || device >= 64)
        return lf_set_error(LF_E_INVALID, "device %d out of range (0..%d)", device, count - 1);
    LF_HIP(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    lf_device_ctx &c = g_ctx[device];
    if (!c.ready) {
        LF_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        LF_HIP(hipEventCreate(&c.t0));
        LF_HIP(hipEventCreate(&c.t1));
        c.ready = true;
    }
    if (out) *out = &c;
    return LF_OK;
}

extern "C" {

const char *lf_last_error(void) { return g_err; }
int lf_version(void) { return 100; }

int lf_struct_sizes(int64_t out[7])
{
    if (!out) return lf_set_error(LF_E_INVALID, "null argument");
    out[0] = (int64_t)sizeof(lf_substep_args);
    out[1] = (int64_t)sizeof(lf_interception_args);
    out[2] = (int64_t)sizeof(lf_soil_args);
    out[3] = (int64_t)sizeof(lf_canopy_args);
    out[4] = (int64_t)sizeof(lf_surface_args);
    out[5] = (int64_t)sizeof(lf_inloop_args);
    out[6] = (int64_t)sizeof(lf_pixel_args);
    return LF_OK;
}

int lf_device_count(int *count)
{
    if (!count) return lf_set_error(LF_E_INVALID, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return LF_OK;
}

int lf_device_name(int device, char *buf, size_t buflen)
{
    if (!buf || buflen == 0) return lf_set_error(LF_E_INVALID, "null argument");
    LF_TRY(lf_ctx(device, nullptr));
    hipDeviceProp_t prop;
    LF_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s", prop.gcnArchName);
    return LF_OK;
}

int lf_device_alloc(int device, size_t bytes, void **ptr_dev)
{
    if (!ptr_dev) return lf_set_error(LF_E_INVALID, "null argument");
    LF_TRY(lf_ctx(device, nullptr));
    *ptr_dev = nullptr;
    if (bytes == 0) return LF_OK;
    static const bool skew_on = [] {
        const char *e = std::getenv("LF_ALLOC_SKEW");
        return !(e && e[0] == '0');
    }();
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    const size_t skew = skew_on ? (g_alloc_counter++ % kSkewSlots) * kSkewStep : 0;
    void *base = nullptr;
    LF_HIP(hipMalloc(&base, bytes + skew));
    *ptr_dev = (char *)base + skew;
    g_alloc_base[*ptr_dev] = base;
    return LF_OK;
}

int lf_device_free(int device, void *ptr_dev)
{
    lf_device_ctx *c = nullptr;
    LF_TRY(lf_ctx(device, &c));
    if (!ptr_dev) return LF_OK;
    if (c) { // the fp32 staging buffer of lf_upload_copy_f32 for this vector goes with it
        auto st = c->f32_stage.find(ptr_dev);
        if (st != c->f32_stage.end()) {
            if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
            if (st->second.first) (void)hipFree(st->second.first);
            c->f32_stage.erase(st);
        }
    }
    void *base = ptr_dev;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        auto it = g_alloc_base.find(ptr_dev);
        if (it != g_alloc_base.end()) {
            base = it->second;
            g_alloc_base.erase(it);
        }
    }
    LF_HIP(hipFree(base));
    return LF_OK;
}

static int side_join(lf_device_ctx *c);

int lf_host_alloc(int device, size_t bytes, void **ptr_host)
{
    if (!ptr_host) return lf_set_error(LF_E_INVALID, "null argument");
    *ptr_host = nullptr;
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    LF_HIP(hipHostMalloc(ptr_host, bytes ? bytes : 1, hipHostMallocDefault));
    return LF_OK;
}

int lf_host_free(int device, void *ptr_host)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (ptr_host) LF_HIP(hipHostFree(ptr_host));
    return LF_OK;
}

int lf_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    LF_TRY(side_join(c));
    if (bytes) {
        LF_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
        LF_HIP(hipStreamSynchronize(c->stream));
    }
    return LF_OK;
}

// A small upload that does not stall the host: the bytes go to a page-locked slot of a ring owned by the context (host
// memcpy, the source is free again on return) and from there by DMA on the stream the library calls currently go to -- main,
// side or lane -- in order with the kernels around it.  lf_memcpy_h2d waits for the stream (its source may be pageable and
// the caller may reuse it at once): for the per-step vectors of a small domain that wait IS the step (LF_ETRS89: two 23 kB
// inflow vectors cost 0.7 of 1.85 ms).  A slot is reused only after its copy has finished (event per slot).
int lf_memcpy_h2d_staged(int device, void *dst_dev, const void *src_host, size_t bytes)
{
    if (!dst_dev || (!src_host && bytes)) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (!bytes) return LF_OK;
    lf_device_ctx::staged_slot &sl = c->staged[c->staged_next];
    c->staged_next = (c->staged_next + 1) % 8;
    if (sl.in_flight) LF_HIP(hipEventSynchronize(sl.done));
    sl.in_flight = false;
    if (sl.cap < bytes) {
        if (sl.host) (void)hipHostFree(sl.host);
        sl.host = nullptr;
        sl.cap = 0;
        LF_HIP(hipHostMalloc(&sl.host, bytes, hipHostMallocDefault));
        sl.cap = bytes;
    }
    if (!sl.done) LF_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    std::memcpy(sl.host, src_host, bytes);
    LF_HIP(hipMemcpyAsync(dst_dev, sl.host, bytes, hipMemcpyHostToDevice, c->stream));
    LF_HIP(hipEventRecord(sl.done, c->stream));
    sl.in_flight = true;
    return LF_OK;
}

// ---- double-buffered uploads on a second stream ------------------------------------------------------------------
// Protocol for buffer set b in {0, 1}:   lf_upload_begin(b); lf_upload_copy(...) x n; lf_upload_end(b)   [any time]
//                                        lf_compute_acquire(b); <kernels reading set b>; lf_compute_release(b)
// The copies wait for the kernels that last read set b, the kernels wait for the copies; neither blocks the other set.
static int upload_ctx(int device, int set, lf_device_ctx **out)
{
    if (set < 0 || set > 1) return lf_set_error(LF_E_INVALID, "buffer set must be 0 or 1");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (!c->copy_stream) {
        LF_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            LF_HIP(hipEventCreateWithFlags(&c->copied[b], hipEventDisableTiming));
            LF_HIP(hipEventCreateWithFlags(&c->consumed[b], hipEventDisableTiming));
        }
    }
    *out = c;
    return LF_OK;
}

int lf_upload_begin(int device, int set)
{
    lf_device_ctx *c;
    LF_TRY(upload_ctx(device, set, &c));
    if (c->consumed_valid[set]) LF_HIP(hipStreamWaitEvent(c->copy_stream, c->consumed[set], 0));
    return LF_OK;
}

int lf_upload_copy(int device, void *dst_dev, const void *src_host, size_t bytes)
{
    if (!dst_dev || (!src_host && bytes)) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(upload_ctx(device, 0, &c));
    if (bytes) LF_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->copy_stream));
    return LF_OK;
}

// fp32 forcing: the reference's meteo files hold float32 (tests/data/LF_ETRS89_UseCase/meteo_1950: "float32 as stored") and
// widens them on the host; widening on the device is the same exact conversion and halves what crosses PCIe.  The fp32
// image goes to a staging buffer kept per destination vector, the widening kernel runs on the copy stream behind it.
__global__ void k_widen_f32(long long n, const float *__restrict__ src, double *__restrict__ dst)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

int lf_upload_copy_f32(int device, double *dst_dev, const float *src_host, size_t count)
{
    if (!dst_dev || (!src_host && count)) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(upload_ctx(device, 0, &c));
    if (!count) return LF_OK;
    auto &st = c->f32_stage[(void *)dst_dev];
    if (st.second < count * sizeof(float)) {
        LF_HIP(hipStreamSynchronize(c->copy_stream)); // (a copy into the old buffer may still be in flight)
        if (st.first) (void)hipFree(st.first);
        st.first = nullptr;
        st.second = 0;
        LF_HIP(hipMalloc(&st.first, count * sizeof(float)));
        st.second = count * sizeof(float);
    }
    LF_HIP(hipMemcpyAsync(st.first, src_host, count * sizeof(float), hipMemcpyHostToDevice, c->copy_stream));
    hipLaunchKernelGGL(k_widen_f32, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->copy_stream, (long long)count,
                       (const float *)st.first, dst_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_upload_end(int device, int set)
{
    lf_device_ctx *c;
    LF_TRY(upload_ctx(device, set, &c));
    LF_HIP(hipEventRecord(c->copied[set], c->copy_stream));
    c->copied_valid[set] = true;
    return LF_OK;
}

// The host side of the protocol: a caller that refills a page-locked source buffer in place (the netCDF reader writing the
// next step's forcing into the arrays of pinned_forcing()) must not do so while the DMA that reads it is in flight.
// Blocks until every copy enqueued between the last lf_upload_begin(set) / lf_upload_end(set) pair has finished; returns at
// once if the set was never uploaded.
int lf_upload_wait(int device, int set)
{
    lf_device_ctx *c;
    LF_TRY(upload_ctx(device, set, &c));
    if (c->copied_valid[set]) LF_HIP(hipEventSynchronize(c->copied[set]));
    return LF_OK;
}

int lf_compute_acquire(int device, int set)
{
    lf_device_ctx *c;
    LF_TRY(upload_ctx(device, set, &c));
    LF_HIP(hipStreamWaitEvent(c->stream, c->copied[set], 0));
    return LF_OK;
}

int lf_compute_release(int device, int set)
{
    lf_device_ctx *c;
    LF_TRY(upload_ctx(device, set, &c));
    LF_HIP(hipEventRecord(c->consumed[set], c->stream));
    c->consumed_valid[set] = true;
    return LF_OK;
}

// ---- lanes: several streams of one device, for work items that do not depend on one another ------------------------------
// The blocks of a row-block partition that live on ONE GPU (the loopback form of the multi-GPU path) are independent inside a
// phase -- on real hardware every block has its own GPU.  lf_lane_select(i > 0) sends the library calls that follow to lane
// stream i (created on first use), which first waits for everything the main stream held at the last lf_lane_fork();
// lf_lane_select(0) returns to the main stream; lf_lane_join() makes the main stream wait for all lanes used since the fork.
//     lf_lane_fork(); for k: lf_lane_select(k + 1); <block k's kernels>;   lf_lane_select(0); lf_lane_join();
int lf_lane_fork(int device)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (c->side_active || c->lane_current != 0) return lf_set_error(LF_E_INVALID, "lf_lane_fork inside a side section or a lane");
    if (c->lane_forked) return lf_set_error(LF_E_INVALID, "lf_lane_fork: the previous fork has not been joined (lf_lane_join)");
    if (!c->lane_fork) LF_HIP(hipEventCreateWithFlags(&c->lane_fork, hipEventDisableTiming));
    LF_HIP(hipEventRecord(c->lane_fork, c->stream));
    c->lane_main = c->stream;
    std::fill(c->lane_used.begin(), c->lane_used.end(), false);
    c->lane_forked = true;
    return LF_OK;
}

int lf_lane_select(int device, int lane)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (lane < 0 || lane > 64) return lf_set_error(LF_E_INVALID, "lane out of range");
    if (!c->lane_forked) return lf_set_error(LF_E_INVALID, "lf_lane_fork first");
    if (lane == 0) {
        c->stream = c->lane_main;
        c->lane_current = 0;
        return LF_OK;
    }
    if ((int)c->lanes.size() < lane) {
        c->lanes.resize(lane, nullptr);
        c->lane_done.resize(lane, nullptr);
        c->lane_used.resize(lane, false);
    }
    if (!c->lanes[lane - 1]) {
        LF_HIP(hipStreamCreateWithFlags(&c->lanes[lane - 1], hipStreamNonBlocking));
        LF_HIP(hipEventCreateWithFlags(&c->lane_done[lane - 1], hipEventDisableTiming));
    }
    if (!c->lane_used[lane - 1]) { // first use since the fork: behind everything the main stream held at the fork
        LF_HIP(hipStreamWaitEvent(c->lanes[lane - 1], c->lane_fork, 0));
        c->lane_used[lane - 1] = true;
    }
    c->stream = c->lanes[lane - 1];
    c->lane_current = lane;
    return LF_OK;
}

int lf_lane_join(int device)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (!c->lane_forked || c->lane_current != 0) return lf_set_error(LF_E_INVALID, "lf_lane_join: select lane 0 first");
    for (size_t i = 0; i < c->lanes.size(); ++i)
        if (c->lane_used[i]) {
            LF_HIP(hipEventRecord(c->lane_done[i], c->lanes[i]));
            LF_HIP(hipStreamWaitEvent(c->stream, c->lane_done[i], 0));
            c->lane_used[i] = false;
        }
    c->lane_forked = false;
    return LF_OK;
}

// ---- side stream ---------------------------------------------------------------------------------------------------------
// lf_side_stream_begin: every library call from now on (kernels, memsets, copies -- they all take the context's stream)
// goes to the side stream, behind everything enqueued on the main stream so far.  lf_side_stream_end: back to the main
// stream; the side work stays in flight BESIDE what the main stream gets next.  lf_side_stream_join: the main stream waits
// for the side work issued so far.  The copies of this file join by themselves when they run on the main stream, so a
// download never sees a vector the side stream is still writing; a caller that enqueues a KERNEL on the main stream which
// touches such a vector joins first.  One thread per device context, as everywhere in this library.
static int side_join(lf_device_ctx *c)
{
    if (c->side_pending && !c->side_active) {
        LF_HIP(hipStreamWaitEvent(c->stream, c->ev_side_done, 0));
        c->side_pending = false;
    }
    return LF_OK;
}

int lf_side_stream_begin(int device)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (c->side_active) return lf_set_error(LF_E_INVALID, "the side stream is already active");
    if (c->lane_current != 0 || c->lane_forked)
        return lf_set_error(LF_E_INVALID, "lf_side_stream_begin between lf_lane_fork and lf_lane_join");
    if (!c->side_stream) {
        LF_HIP(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
        LF_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        LF_HIP(hipEventCreateWithFlags(&c->ev_side_done, hipEventDisableTiming));
    }
    LF_HIP(hipEventRecord(c->ev_fork, c->stream));
    LF_HIP(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
    c->main_stream = c->stream;
    c->stream = c->side_stream;
    c->side_active = true;
    return LF_OK;
}

int lf_side_stream_end(int device)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (!c->side_active) return lf_set_error(LF_E_INVALID, "the side stream is not active");
    LF_HIP(hipEventRecord(c->ev_side_done, c->side_stream));
    c->stream = c->main_stream;
    c->side_active = false;
    c->side_pending = true;
    return LF_OK;
}

int lf_side_stream_join(int device)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (c->side_active) return lf_set_error(LF_E_INVALID, "join from inside the side section");
    return side_join(c);
}

// Releases what the context created on demand and keeps for reuse: the lane streams and their events, the fp32 staging
// buffers of lf_upload_copy_f32, the staging arena of the *_host entry points.  Waits for the device first; refuses inside
// a lane or side section.  Everything is created again on next use.
int lf_device_trim(int device)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (c->side_active || c->lane_current != 0 || c->lane_forked)
        return lf_set_error(LF_E_INVALID, "lf_device_trim inside a lane or side section");
    LF_TRY(lf_device_synchronize(device));
    for (hipStream_t st : c->lanes)
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : c->lane_done)
        if (ev) (void)hipEventDestroy(ev);
    c->lanes.clear();
    c->lane_done.clear();
    c->lane_used.clear();
    for (auto &kv : c->f32_stage)
        if (kv.second.first) (void)hipFree(kv.second.first);
    c->f32_stage.clear();
    if (c->stage_base) (void)hipFree(c->stage_base);
    c->stage_base = nullptr;
    c->stage_bytes = 0;
    if (c->soil_ws) (void)hipFree(c->soil_ws); // lists and straggler records of the soil call: rebuilt by the next one
    c->soil_ws = nullptr;
    c->soil_ws_bytes = 0;
    c->soil_ntiles = 0;
    return LF_OK;
}

int lf_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    LF_TRY(side_join(c));
    if (bytes) {
        LF_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        LF_HIP(hipStreamSynchronize(c->stream));
    }
    return LF_OK;
}

int lf_memcpy_d2d(int device, void *dst_dev, const void *src_dev, size_t bytes)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    LF_TRY(side_join(c));
    if (bytes) LF_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
    return LF_OK;
}

int lf_memset(int device, void *dst_dev, int value, size_t bytes)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    LF_TRY(side_join(c)); // like the copies: a vector the side stream's wavefront still works on must not be cleared under it
    if (bytes) LF_HIP(hipMemsetAsync(dst_dev, value, bytes, c->stream));
    return LF_OK;
}

int lf_device_synchronize(int device)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    LF_HIP(hipStreamSynchronize(c->stream));
    LF_HIP(hipDeviceSynchronize());
    if (!c->side_active) c->side_pending = false;
    return LF_OK;
}

int lf_timer_start(int device)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    LF_HIP(hipEventRecord(c->t0, c->stream));
    return LF_OK;
}

int lf_timer_stop(int device, double *elapsed_ms)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    LF_HIP(hipEventRecord(c->t1, c->stream));
    LF_HIP(hipEventSynchronize(c->t1));
    float ms = 0.f;
    LF_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    if (elapsed_ms) *elapsed_ms = (double)ms;
    return LF_OK;
}

} // extern "C"
