// TEST INFRASTRUCTURE -- not part of the product, never loaded by it unless a test puts this directory first on
// LD_LIBRARY_PATH.
//
// A stand-in for the eight RCCL entry points csrc/lf_dist.hip binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
// ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString) that works between PROCESSES SHARING ONE GPU,
// which RCCL itself refuses ("duplicate GPU").  The test boxes have one MI355X; with this library the row-block
// partition runs as it does on an 8-GPU node -- one process per rank, socket rendezvous, the product's own order of
// group calls, sends, receives, streams and events -- and a protocol mistake (a send nobody receives, counts that differ
// between the two ends, a group whose order only works by luck, a kernel reading ghost slots before the halo landed)
// shows up as a wrong result or as a time-out instead of waiting for the first real multi-GPU run.
//
// Semantics kept from RCCL:
//   * ncclSend / ncclRecv are asynchronous: they enqueue work on the caller's stream and return;
//   * the operations between ncclGroupStart and ncclGroupEnd progress concurrently (one kernel, one workgroup per
//     operation), operations outside a group one after the other in stream order;
//   * a message is matched by (source, destination) in issue order; the receiver checks the byte count.
// How: every rank owns a mailbox per source rank in fine-grained device memory, shared through hipIpc handles published
// in a POSIX shared-memory segment named after the unique id.  A send kernel waits until the mailbox is free, copies the
// payload in chunks of kCap bytes and publishes a sequence number; a receive kernel waits for it, copies out and
// acknowledges.  Flags and payload move with system-scope atomics (no stale L2 lines between XCDs or processes).  Every
// wait gives up after FAKE_RCCL_TIMEOUT_S seconds (default 30) and raises the error flag: the GPU is never left spinning.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace {
constexpr int kMaxRanks = 16;
constexpr size_t kCap = 1u << 20; // payload bytes per chunk
constexpr int kBlock = 256;
constexpr int kMaxGroup = 32;

enum { kSuccess = 0, kUnhandledHip = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };

struct mailbox {
    unsigned long long seq;   // written by the sender: chunks published so far
    unsigned long long ack;   // written by the receiver: chunks consumed so far
    unsigned long long bytes; // size of the message the current chunk belongs to
    unsigned long long pad[5];
    unsigned long long data[kCap / 8];
};

struct control { // the POSIX shared-memory segment
    std::atomic<int> arrived, opened, leaving;
    std::atomic<int> error;
    hipIpcMemHandle_t handle[kMaxRanks];
};

struct comm {
    int nranks = 0, rank = 0, device = 0;
    char shm_name[64] = {0};
    control *ctl = nullptr;
    mailbox *mine = nullptr;            // [nranks] mailboxes this rank receives into (index = source rank)
    mailbox *peer[kMaxRanks] = {};      // peer[d] = rank d's mailbox array (peer[rank] = mine)
    unsigned long long sent[kMaxRanks] = {}, rcvd[kMaxRanks] = {}; // chunks issued per pair
    unsigned int *err = nullptr;        // pinned host word the kernels raise
    long long timeout_ticks = 0;
    // FAKE_RCCL_LOG_DIR: every operation in host issue order -- group number, send / receive, peer, bytes -- written to
    // <dir>/rank<r>.log when the communicator is destroyed.  RCCL matches the operations of a pair of ranks in issue
    // order, so for every pair (a, b) a's sends to b and b's receives from a must be the same sequence of sizes
    // (tests/test_dist_multirank_gpu.py: check_issue_order); a mailbox with equal sizes would not notice a swap.
    struct logged {
        long long group;
        int send, peer;
        unsigned long long bytes;
    };
    std::vector<logged> log;
    long long groups = 0;
};

struct op {
    int send;
    char *user;
    unsigned long long bytes;
    mailbox *mb;
    unsigned long long base;
};
struct op_list {
    int n;
    op ops[kMaxGroup];
    unsigned int *err;
    long long timeout_ticks;
};

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_data(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_data(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// thread 0 waits until *flag == want; false on time-out (or when another operation already failed)
__device__ bool wait_for(const unsigned long long *flag, unsigned long long want, unsigned int *err, long long ticks)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int good = 1;
        while (ld_sys(flag) != want) {
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > ticks) {
                good = 0;
                __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        ok = good;
    }
    __syncthreads();
    const bool r = ok != 0;
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(kBlock) k_ops(op_list L)
{
    const op o = L.ops[blockIdx.x];
    const unsigned long long nchunk = (o.bytes + kCap - 1) / kCap;
    for (unsigned long long c = 0; c < nchunk; ++c) {
        const unsigned long long off = c * kCap;
        const unsigned long long len = (o.bytes - off < kCap) ? o.bytes - off : kCap; // multiple of 8
        const unsigned long long words = len / 8;
        unsigned long long *u = (unsigned long long *)(o.user + off);
        if (o.send) {
            if (!wait_for(&o.mb->ack, o.base + c, L.err, L.timeout_ticks)) return;
            for (unsigned long long i = threadIdx.x; i < words; i += kBlock) st_data(&o.mb->data[i], u[i]);
            __threadfence_system();
            __syncthreads();
            if (threadIdx.x == 0) {
                st_data(&o.mb->bytes, o.bytes);
                st_sys(&o.mb->seq, o.base + c + 1);
            }
        } else {
            if (!wait_for(&o.mb->seq, o.base + c + 1, L.err, L.timeout_ticks)) return;
            if (threadIdx.x == 0 && ld_data(&o.mb->bytes) != o.bytes)
                __hip_atomic_store(L.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // the two ends disagree
            for (unsigned long long i = threadIdx.x; i < words; i += kBlock) u[i] = ld_data(&o.mb->data[i]);
            __threadfence_system();
            __syncthreads();
            if (threadIdx.x == 0) st_sys(&o.mb->ack, o.base + c + 1);
        }
        __syncthreads();
    }
}

thread_local int t_depth = 0;
thread_local std::vector<op> t_ops;
thread_local comm *t_comm = nullptr;
thread_local hipStream_t t_stream = nullptr;
thread_local bool t_have_stream = false;

double now_s()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

bool spin_until(std::atomic<int> &a, int want, double seconds)
{
    const double t0 = now_s();
    while (a.load() < want) {
        if (now_s() - t0 > seconds) return false;
        usleep(200);
    }
    return true;
}

int check_error(comm *c)
{
    if (!c || !c->err) return kSuccess;
    const unsigned int e = __atomic_load_n(c->err, __ATOMIC_RELAXED);
    if (e == 0 && c->ctl->error.load() == 0) return kSuccess;
    if (e) c->ctl->error.store((int)e);
    std::fprintf(stderr, "fake_rccl rank %d: %s\n", c->rank,
                 e == 1 ? "send and receive byte counts differ" : "an operation timed out (or a peer failed)");
    return kInternalError;
}

int launch(comm *c, hipStream_t stream, const op *ops, int n)
{
    if (n == 0) return kSuccess;
    if (n > kMaxGroup) return kInvalidUsage;
    op_list L;
    L.n = n;
    for (int i = 0; i < n; ++i) L.ops[i] = ops[i];
    L.err = c->err;
    L.timeout_ticks = c->timeout_ticks;
    hipLaunchKernelGGL(k_ops, dim3(n), dim3(kBlock), 0, stream, L);
    return hipGetLastError() == hipSuccess ? kSuccess : kUnhandledHip;
}

int enqueue(comm *c, int send, void *buf, size_t count, int dtype, int peer, hipStream_t stream)
{
    if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return kInvalidArgument;
    size_t width;
    switch (dtype) { // ncclDataType_t
    case 0: case 1: width = 1; break;          // int8 / uint8
    case 2: case 3: case 7: width = 4; break;  // int32 / uint32 / float32
    case 4: case 5: case 8: width = 8; break;  // int64 / uint64 / float64
    case 6: case 9: width = 2; break;          // float16 / bfloat16
    default: return kInvalidArgument;
    }
    const unsigned long long bytes = (unsigned long long)count * width;
    if (bytes % 8 != 0 || ((uintptr_t)buf & 7)) return kInvalidArgument; // the shim moves 8-byte words
    if (int e = check_error(c)) return e;
    if (bytes == 0) return kSuccess;
    c->log.push_back({t_depth == 0 ? c->groups++ : c->groups, send, peer, bytes});
    op o;
    o.send = send;
    o.user = (char *)buf;
    o.bytes = bytes;
    const unsigned long long nchunk = (bytes + kCap - 1) / kCap;
    if (send) {
        o.mb = c->peer[peer] + c->rank;
        o.base = c->sent[peer];
        c->sent[peer] += nchunk;
    } else {
        o.mb = c->mine + peer;
        o.base = c->rcvd[peer];
        c->rcvd[peer] += nchunk;
    }
    if (t_depth == 0) return launch(c, stream, &o, 1);
    if (t_comm && t_comm != c) return kInvalidUsage;
    if (t_have_stream && t_stream != stream) {
        std::fprintf(stderr, "fake_rccl: the operations of one group must use one stream\n");
        return kInvalidUsage;
    }
    t_comm = c;
    t_stream = stream;
    t_have_stream = true;
    t_ops.push_back(o);
    return kSuccess;
}
} // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return kInvalidArgument;
    std::memset(id->internal, 0, sizeof id->internal);
    unsigned long long r[2] = {(unsigned long long)getpid(), (unsigned long long)(now_s() * 1e6)};
    if (FILE *f = std::fopen("/dev/urandom", "rb")) {
        if (std::fread(r, sizeof r, 1, f) != 1) r[1] ^= 0x9e3779b97f4a7c15ull;
        std::fclose(f);
    }
    std::snprintf(id->internal, sizeof id->internal, "/lf_fake_rccl_%016llx%016llx", r[0], r[1]);
    return kSuccess;
}

int ncclCommInitRank(void **out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return kInvalidArgument;
    if (std::strncmp(id.internal, "/lf_fake_rccl_", 14) != 0) return kInvalidArgument;
    comm *c = new comm();
    c->nranks = nranks;
    c->rank = rank;
    std::snprintf(c->shm_name, sizeof c->shm_name, "%s", id.internal);
    const char *ts = std::getenv("FAKE_RCCL_TIMEOUT_S");
    const double timeout_s = ts ? std::atof(ts) : 30.0;
    c->timeout_ticks = (long long)(timeout_s * 1e8); // wall_clock64 counts at 100 MHz
    if (hipGetDevice(&c->device) != hipSuccess) return kUnhandledHip;
    const int fd = shm_open(c->shm_name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(control)) != 0) return kSystemError;
    c->ctl = (control *)mmap(nullptr, sizeof(control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->ctl == MAP_FAILED) return kSystemError;
    const size_t bytes = sizeof(mailbox) * (size_t)nranks;
    if (hipExtMallocWithFlags((void **)&c->mine, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc((void **)&c->mine, bytes) != hipSuccess) return kUnhandledHip;
    }
    for (int s = 0; s < nranks; ++s)
        if (hipMemset(c->mine + s, 0, 64) != hipSuccess) return kUnhandledHip; // the flags; payload needs no clearing
    if (hipDeviceSynchronize() != hipSuccess) return kUnhandledHip;
    if (hipHostMalloc((void **)&c->err, 64, hipHostMallocMapped) != hipSuccess) return kUnhandledHip;
    *c->err = 0;
    c->peer[rank] = c->mine;
    if (nranks > 1) {
        if (hipIpcGetMemHandle(&c->ctl->handle[rank], c->mine) != hipSuccess) { // fine-grained memory refused: plain
            (void)hipGetLastError();
            (void)hipFree(c->mine);
            if (hipMalloc((void **)&c->mine, bytes) != hipSuccess) return kUnhandledHip;
            for (int s = 0; s < nranks; ++s)
                if (hipMemset(c->mine + s, 0, 64) != hipSuccess) return kUnhandledHip;
            if (hipDeviceSynchronize() != hipSuccess) return kUnhandledHip;
            c->peer[rank] = c->mine;
            if (hipIpcGetMemHandle(&c->ctl->handle[rank], c->mine) != hipSuccess) {
                std::fprintf(stderr, "fake_rccl rank %d: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)\n", rank);
                return kUnhandledHip;
            }
        }
        c->ctl->arrived.fetch_add(1);
        if (!spin_until(c->ctl->arrived, nranks, 120.0)) return kSystemError;
        for (int d = 0; d < nranks; ++d)
            if (d != rank &&
                hipIpcOpenMemHandle((void **)&c->peer[d], c->ctl->handle[d], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                std::fprintf(stderr, "fake_rccl rank %d: hipIpcOpenMemHandle of rank %d failed\n", rank, d);
                return kUnhandledHip;
            }
        c->ctl->opened.fetch_add(1);
        if (!spin_until(c->ctl->opened, nranks, 120.0)) return kSystemError;
    }
    if (rank == 0) shm_unlink(c->shm_name); // everybody has it mapped
    *out = c;
    return kSuccess;
}

int ncclCommDestroy(void *p)
{
    comm *c = (comm *)p;
    if (!c) return kSuccess;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    const int e = check_error(c);
    if (const char *dir = std::getenv("FAKE_RCCL_LOG_DIR")) {
        char path[512];
        std::snprintf(path, sizeof path, "%s/rank%d.log", dir, c->rank);
        if (FILE *f = std::fopen(path, "a")) { // (a process may make several communicators in a row)
            for (const comm::logged &l : c->log)
                std::fprintf(f, "%lld %s %d %llu\n", l.group, l.send ? "send" : "recv", l.peer, l.bytes);
            std::fclose(f);
        }
    }
    if (c->nranks > 1) { // nobody unmaps a mailbox a peer may still be writing to
        c->ctl->leaving.fetch_add(1);
        spin_until(c->ctl->leaving, c->nranks, 60.0);
        for (int d = 0; d < c->nranks; ++d)
            if (d != c->rank && c->peer[d]) (void)hipIpcCloseMemHandle(c->peer[d]);
    }
    (void)hipFree(c->mine);
    (void)hipHostFree(c->err);
    munmap(c->ctl, sizeof(control));
    delete c;
    return e;
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *c, hipStream_t stream)
{
    return enqueue((comm *)c, 1, const_cast<void *>(buf), count, dtype, peer, stream);
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *c, hipStream_t stream)
{
    return enqueue((comm *)c, 0, buf, count, dtype, peer, stream);
}

int ncclGroupStart()
{
    ++t_depth;
    return kSuccess;
}

int ncclGroupEnd()
{
    if (t_depth <= 0) return kInvalidUsage;
    if (--t_depth > 0) return kSuccess;
    int e = kSuccess;
    if (!t_ops.empty()) {
        e = launch(t_comm, t_stream, t_ops.data(), (int)t_ops.size());
        ++t_comm->groups;
    }
    t_ops.clear();
    t_comm = nullptr;
    t_have_stream = false;
    return e;
}

const char *ncclGetErrorString(int e)
{
    switch (e) {
    case kSuccess: return "no error";
    case kUnhandledHip: return "fake_rccl: unhandled HIP error";
    case kSystemError: return "fake_rccl: system error (shared memory / rendezvous time-out)";
    case kInternalError: return "fake_rccl: an operation failed on the device (count mismatch or time-out, see stderr)";
    case kInvalidArgument: return "fake_rccl: invalid argument";
    case kInvalidUsage: return "fake_rccl: invalid usage";
    default: return "fake_rccl: unknown error";
    }
}

} // extern "C"
