// libsimq: data-parallel gradient exchange over RCCL (xGMI), behind the C-ABI (simq_comm_*).
//
// Replaces the reduce-add half of torch.nn.DataParallel (reference policies.py:39: replicas' gradients are summed onto
// device 0 inside one process) with one process per GPU: every rank sums its flat gradient buffer with the others'.
// The communicator owns ONE non-blocking HIP stream; a collective is enqueued there behind an event recorded on the
// producer's stream, so the backward kernels that follow on the producer stream keep running while the bucket travels;
// simq_comm_wait makes a consumer stream wait for everything enqueued so far.  No host synchronisation anywhere.
//
// librccl.so.1 is bound at run time (dlopen): a host that already loaded RCCL (PyTorch-ROCm does) shares that copy, a
// single-GPU host that never calls simq_comm_* needs no RCCL at all, and a missing library is a loud error, not a fallback.
#include <dlfcn.h>

#include <cstring>

#include "../../include/simq.h"
#include <atomic>

#include "common.h"

namespace {

// the subset of rccl.h this file uses (ABI-stable NCCL 2.x entry points; declared here so the build needs no RCCL headers)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                                   // 0 == ncclSuccess
enum { kNcclInt8 = 0, kNcclFloat32 = 7, kNcclFloat64 = 8 }; // ncclDataType_t
enum { kNcclSum = 0 };                                      // ncclRedOp_t

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return 0;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // the copy the host process already uses, if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        simq::set_error("simq_comm: librccl.so.1 cannot be loaded (%s)", dlerror());
        return -3;
    }
    Rccl r;
    r.handle = h;
#define SIMQ_SYM(field, name)                                                            \
    *reinterpret_cast<void**>(&r.field) = dlsym(h, name);                                 \
    if (!r.field) { simq::set_error("simq_comm: librccl.so.1 lacks %s", name); return -3; }
    SIMQ_SYM(GetUniqueId, "ncclGetUniqueId")
    SIMQ_SYM(CommInitRank, "ncclCommInitRank")
    SIMQ_SYM(CommDestroy, "ncclCommDestroy")
    SIMQ_SYM(AllReduce, "ncclAllReduce")
    SIMQ_SYM(Broadcast, "ncclBroadcast")
    SIMQ_SYM(GetErrorString, "ncclGetErrorString")
#undef SIMQ_SYM
    g_rccl = r;
    return 0;
}

#define SIMQ_CHECK_RCCL(expr)                                                                                     \
    do {                                                                                                          \
        ncclResult_t _r = (expr);                                                                                 \
        if (_r != 0) {                                                                                            \
            simq::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(_r));              \
            return -4;                                                                                            \
        }                                                                                                         \
    } while (0)

constexpr int kEvents = 16;

}  // namespace

struct simq_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;          // library-owned: every collective runs here
    hipStream_t stream_ext = nullptr;      // ... or here: the caller's stream (simq_comm_adopt_stream), never destroyed by the library
    hipStream_t cs() const { return stream_ext ? stream_ext : stream; }
    hipEvent_t ready[kEvents] = {};        // producer stream -> comm stream (ring: one per collective in flight)
    hipEvent_t done = nullptr;             // comm stream -> consumer stream
    // exposed-wait timing (simq_comm_time_waits): a timing-enabled pair around the consumer stream's wait -- what the consumer stream
    // LOST waiting for collectives still in flight (0 when they finished behind the kernels that ran beside them)
    hipEvent_t wait_t0 = nullptr, wait_t1 = nullptr;
    int time_waits = 0, wait_timed = 0;
    int next = 0;
    // progress accounting for hang diagnosis (simq_comm_progress): every collective records `fin` behind itself on the comm stream
    hipEvent_t fin[kEvents] = {};
    // (atomics: simq_comm_progress runs on a watchdog thread while the training thread enqueues; `recorded` = how many `fin` events of the
    // current generation exist -- progress never queries a slot whose event still belongs to the collective kEvents earlier)
    std::atomic<int64_t> enqueued{0}, recorded{0}, completed{0};
    std::atomic<int> last_kind{-1};        // 0 all-reduce fp32, 1 all-reduce fp64, 2 broadcast
    std::atomic<int64_t> last_count{0};
};

namespace simq {

// used by simq_train_step (train_step.hip)
int comm_wait(simq_comm* c, hipStream_t consumer);

int comm_allreduce(simq_comm* c, void* buf, int64_t count, int dtype, hipStream_t producer) {
    SIMQ_REQUIRE(c && buf && count > 0, "comm_allreduce: bad argument");
    SIMQ_REQUIRE(dtype == SIMQ_COMM_F32 || dtype == SIMQ_COMM_F64, "comm_allreduce: dtype %d (SIMQ_COMM_F32 | SIMQ_COMM_F64)", dtype);
    hipEvent_t ev = c->ready[c->next];
    c->next = (c->next + 1) % kEvents;
    SIMQ_CHECK_HIP(hipEventRecord(ev, producer));
    SIMQ_CHECK_HIP(hipStreamWaitEvent(c->cs(), ev, 0));
    const int slot = (int)(c->enqueued.load(std::memory_order_relaxed) % kEvents);
    c->last_kind.store(dtype == SIMQ_COMM_F32 ? 0 : 1, std::memory_order_relaxed); c->last_count.store(count, std::memory_order_relaxed);
    c->enqueued.fetch_add(1, std::memory_order_release);
    SIMQ_CHECK_RCCL(g_rccl.AllReduce(buf, buf, (size_t)count, dtype == SIMQ_COMM_F32 ? kNcclFloat32 : kNcclFloat64, kNcclSum, c->comm,
                                     c->cs()));
    SIMQ_CHECK_HIP(hipEventRecord(c->fin[slot], c->cs()));
    c->recorded.fetch_add(1, std::memory_order_release);
    return 0;
}

int comm_reduce_f64(void* comm, double* buf, int64_t count, void* stream) {
    simq_comm* c = static_cast<simq_comm*>(comm);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = comm_allreduce(c, buf, count, SIMQ_COMM_F64, st)) return rc;
    return comm_wait(c, st);
}

hipStream_t comm_stream(simq_comm* c) { return c ? c->cs() : nullptr; }

int comm_wait(simq_comm* c, hipStream_t consumer) {
    SIMQ_REQUIRE(c, "comm_wait: NULL communicator");
    SIMQ_CHECK_HIP(hipEventRecord(c->done, c->cs()));
    if (c->time_waits) SIMQ_CHECK_HIP(hipEventRecord(c->wait_t0, consumer));
    SIMQ_CHECK_HIP(hipStreamWaitEvent(consumer, c->done, 0));
    if (c->time_waits) { SIMQ_CHECK_HIP(hipEventRecord(c->wait_t1, consumer)); c->wait_timed = 1; }
    return 0;
}

}  // namespace simq

extern "C" {

int simq_comm_unique_id(void* id_out) {
    SIMQ_REQUIRE(id_out, "comm_unique_id: NULL argument");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    SIMQ_CHECK_RCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == SIMQ_COMM_ID_BYTES, "unique id size");
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int simq_comm_init(const void* id, int world_size, int rank, simq_comm** out) {
    SIMQ_REQUIRE(id && out, "comm_init: NULL argument");
    SIMQ_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "comm_init: rank %d of %d", rank, world_size);
    if (int rc = load_rccl()) return rc;
    simq_comm* c = new simq_comm();
    c->world = world_size; c->rank = rank;
    auto fail = [&](int rc) { simq_comm_destroy(c); return rc; };
    if (hipGetDevice(&c->device) != hipSuccess) { simq::set_error("comm_init: hipGetDevice failed"); return fail(-2); }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world_size, uid, rank);
    if (r != 0) { simq::set_error("comm_init: ncclCommInitRank(rank %d of %d) -> %s", rank, world_size, g_rccl.GetErrorString(r)); c->comm = nullptr; return fail(-4); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { simq::set_error("comm_init: stream creation failed"); return fail(-2); }
    for (int i = 0; i < kEvents; ++i)
        if (hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming) != hipSuccess) { simq::set_error("comm_init: event creation failed"); return fail(-2); }
    for (int i = 0; i < kEvents; ++i)
        if (hipEventCreateWithFlags(&c->fin[i], hipEventDisableTiming) != hipSuccess) { simq::set_error("comm_init: event creation failed"); return fail(-2); }
    if (hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) { simq::set_error("comm_init: event creation failed"); return fail(-2); }
    *out = c;
    return 0;
}

int simq_comm_world_size(const simq_comm* comm) { return comm ? comm->world : -1; }
int simq_comm_rank(const simq_comm* comm) { return comm ? comm->rank : -1; }

int simq_comm_allreduce(simq_comm* comm, void* d_buf, int64_t count, int dtype, void* producer_stream) {
    return simq::comm_allreduce(comm, d_buf, count, dtype, static_cast<hipStream_t>(producer_stream));
}

int simq_comm_broadcast(simq_comm* comm, void* d_buf, int64_t bytes, int root, void* producer_stream) {
    SIMQ_REQUIRE(comm && d_buf && bytes > 0 && root >= 0 && root < comm->world, "comm_broadcast: bad argument");
    hipEvent_t ev = comm->ready[comm->next];
    comm->next = (comm->next + 1) % kEvents;
    SIMQ_CHECK_HIP(hipEventRecord(ev, static_cast<hipStream_t>(producer_stream)));
    SIMQ_CHECK_HIP(hipStreamWaitEvent(comm->cs(), ev, 0));
    const int slot = (int)(comm->enqueued.load(std::memory_order_relaxed) % kEvents);
    comm->last_kind.store(2, std::memory_order_relaxed); comm->last_count.store(bytes, std::memory_order_relaxed);
    comm->enqueued.fetch_add(1, std::memory_order_release);
    SIMQ_CHECK_RCCL(g_rccl.Broadcast(d_buf, d_buf, (size_t)bytes, kNcclInt8, root, comm->comm, comm->cs()));
    SIMQ_CHECK_HIP(hipEventRecord(comm->fin[slot], comm->cs()));
    comm->recorded.fetch_add(1, std::memory_order_release);
    return 0;
}

// Hang diagnosis (bench.py's watchdog, tools/mgpu_selftest.py): how many collectives this rank has enqueued, how many of them the
// device has finished (event queries, no synchronisation), and what the last one was.  Callable from any host thread while another
// is blocked in a synchronisation.  out = {enqueued, completed, last kind (0 all-reduce fp32, 1 all-reduce fp64, 2 broadcast), last count}
int simq_comm_progress(simq_comm* comm, int64_t out[4]) {
    SIMQ_REQUIRE(comm && out, "comm_progress: NULL argument");
    const int64_t enq = comm->enqueued.load(std::memory_order_acquire), rec = comm->recorded.load(std::memory_order_acquire);
    int64_t done = comm->completed.load(std::memory_order_relaxed);
    if (rec - done > kEvents) done = rec - kEvents;                                                // (older events were re-recorded)
    while (done < rec && hipEventQuery(comm->fin[done % kEvents]) == hipSuccess) ++done;
    (void)hipGetLastError();                                                                       // (hipErrorNotReady is not an error here)
    comm->completed.store(done, std::memory_order_relaxed);
    out[0] = enq; out[1] = done; out[2] = comm->last_kind.load(std::memory_order_relaxed); out[3] = comm->last_count.load(std::memory_order_relaxed);
    return 0;
}

int simq_comm_reduce_f64(void* comm, double* d_buf, int64_t count, void* stream) { return simq::comm_reduce_f64(comm, d_buf, count, stream); }

int simq_comm_wait(simq_comm* comm, void* consumer_stream) {
    return simq::comm_wait(comm, static_cast<hipStream_t>(consumer_stream));
}

int simq_comm_time_waits(simq_comm* comm, int on) {
    SIMQ_REQUIRE(comm, "comm_time_waits: NULL communicator");
    if (on && !comm->wait_t0) {
        SIMQ_CHECK_HIP(hipEventCreate(&comm->wait_t0));
        SIMQ_CHECK_HIP(hipEventCreate(&comm->wait_t1));
    }
    comm->time_waits = on ? 1 : 0;
    comm->wait_timed = 0;
    return 0;
}

int simq_comm_last_wait_ms(simq_comm* comm, float* ms) {
    SIMQ_REQUIRE(comm && ms, "comm_last_wait_ms: NULL argument");
    SIMQ_REQUIRE(comm->wait_timed, "comm_last_wait_ms: no timed simq_comm_wait yet (simq_comm_time_waits(comm, 1) first)");
    SIMQ_CHECK_HIP(hipEventSynchronize(comm->wait_t1));
    SIMQ_CHECK_HIP(hipEventElapsedTime(ms, comm->wait_t0, comm->wait_t1));
    return 0;
}

int simq_comm_adopt_stream(simq_comm* comm, void* stream) {
    SIMQ_REQUIRE(comm, "comm_adopt_stream: NULL communicator");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (st == comm->stream_ext) return 0;
    // collectives of ONE communicator must stay ordered: everything enqueued on the stream in use so far finishes first
    SIMQ_CHECK_HIP(hipStreamSynchronize(comm->cs()));
    comm->stream_ext = st;
    return 0;
}

int simq_comm_destroy(simq_comm* comm) {
    if (!comm) return 0;
    if (comm->cs()) (void)hipStreamSynchronize(comm->cs());
    if (comm->stream) (void)hipStreamSynchronize(comm->stream);
    if (comm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm->comm);
    for (int i = 0; i < kEvents; ++i)
        if (comm->ready[i]) (void)hipEventDestroy(comm->ready[i]);
    for (int i = 0; i < kEvents; ++i)
        if (comm->fin[i]) (void)hipEventDestroy(comm->fin[i]);
    if (comm->done) (void)hipEventDestroy(comm->done);
    if (comm->wait_t0) (void)hipEventDestroy(comm->wait_t0);
    if (comm->wait_t1) (void)hipEventDestroy(comm->wait_t1);
    if (comm->stream) (void)hipStreamDestroy(comm->stream);
    delete comm;
    return 0;
}

}  // extern "C"
