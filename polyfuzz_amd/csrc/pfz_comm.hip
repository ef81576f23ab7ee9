// Multi-GPU exchange, collectives issued on the context's own stream.  Two transports behind the same
// three seams (comm_allgather_bytes / comm_allreduce_sum_i32 / _i64):
//   - RCCL over xGMI, one process per GPU (pfz_comm_init) -- the production path;
//   - "local": one process driving several contexts -- on different GPUs, or on the same one -- whose ranks
//     run on their own host threads (pfz_comm_init_local).  A collective is a host rendezvous of the ranks
//     followed, on every rank's stream, by event waits and device-to-device copies out of the peers' buffers.
//     It lets the sharded code paths run at world > 1 on a one-GPU box (tests/test_comm_gpu.py).
//
// The reference has no distributed code at all (joblib process pool only,
// polyfuzz/models/_distance.py:77); the hot path shards by from-row with the
// to-side replicated, so the exchanges are few and small:
//   - fit over a sharded corpus: all-gather of the vocabulary bitmaps (32 KiB
//     for cleaned 3-grams) + all-reduce of the document frequencies (V int32);
//   - results: all-gather of the per-shard (idx, score) top-n blocks.
// All of them are far below one xGMI link's bandwidth-time; they are grouped
// into as few collectives as possible rather than pipelined.
#include "pfz_internal.h"

#include <rccl/rccl.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <vector>

// rendezvous state of the ranks of one local communicator
struct pfz_comm_group {
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    int attached = 0;
    std::vector<const void *> send;      // per rank: the buffer it contributes to the running collective
    std::vector<hipEvent_t> ready, done; // per rank: "send buffer final" / "my copies out of the peers are enqueued and run"
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
};

struct pfz_comm {
    pfz_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;           // RCCL transport
    pfz_comm_group *group = nullptr;     // local transport
    int rank = 0, world = 1;
    int32_t *flag = nullptr;  // device, barrier payload
};

namespace pfz {

template <typename T>
__global__ __launch_bounds__(256) void k_sum_ranks(const T *__restrict__ gathered, int64_t n, int world, T *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T s = 0;
    for (int p = 0; p < world; ++p) s += gathered[(int64_t)p * n + i];
    out[i] = s;
}

static int local_allgather(pfz_comm *c, const void *send, void *recv, size_t bytes_per_rank)
{
    pfz_comm_group *g = c->group;
    hipStream_t st = c->ctx->stream;
    PFZ_HIP(hipEventRecord(g->ready[c->rank], st));
    g->send[c->rank] = send;
    g->barrier();                       // every rank has published its buffer and recorded its event
    for (int p = 0; p < c->world; ++p) {
        if (p != c->rank) PFZ_HIP(hipStreamWaitEvent(st, g->ready[p], 0));
        PFZ_HIP(hipMemcpyAsync((char *)recv + (size_t)p * bytes_per_rank, g->send[p], bytes_per_rank, hipMemcpyDefault, st));
    }
    PFZ_HIP(hipEventRecord(g->done[c->rank], st));
    g->barrier();                       // every rank has enqueued its copies
    // what this rank enqueues next may overwrite `send`: not before the peers have read it
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank) PFZ_HIP(hipStreamWaitEvent(st, g->done[p], 0));
    return PFZ_OK;
}

template <typename T> static int local_allreduce_sum(pfz_comm *c, T *buf, size_t n)
{
    T *tmp = nullptr;
    PFZ_TRY(pool_alloc(c->ctx, &tmp, (size_t)c->world * n * sizeof(T)));
    int rc = local_allgather(c, buf, tmp, n * sizeof(T));
    if (rc == PFZ_OK) {
        hipLaunchKernelGGL((k_sum_ranks<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->ctx->stream, tmp, (int64_t)n,
                           c->world, buf);
        if (hipGetLastError() != hipSuccess) rc = PFZ_ERR_HIP;
    }
    pool_free(tmp);                     // stream-ordered reuse: the next user of the block is behind the kernel
    return rc;
}

static int rccl_fail(ncclResult_t r, const char *what, int line)
{
    set_error("RCCL error %d (%s) in %s at pfz_comm.hip:%d", (int)r, ncclGetErrorString(r), what, line);
    return PFZ_ERR_RCCL;
}

#define PFZ_RCCL(call)                                              \
    do {                                                            \
        ncclResult_t _r = (call);                                   \
        if (_r != ncclSuccess) return pfz::rccl_fail(_r, #call, __LINE__); \
    } while (0)

int comm_rank(const pfz_comm *c) { return c ? c->rank : 0; }
int comm_world(const pfz_comm *c) { return c ? c->world : 1; }

int comm_allgather_bytes(pfz_comm *c, const void *send, void *recv, size_t bytes_per_rank)
{
    if (c->group) return local_allgather(c, send, recv, bytes_per_rank);
    PFZ_RCCL(ncclAllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, c->ctx->stream));
    return PFZ_OK;
}

int comm_allreduce_sum_i32(pfz_comm *c, int32_t *buf, size_t n)
{
    if (c->group) return local_allreduce_sum<int32_t>(c, buf, n);
    PFZ_RCCL(ncclAllReduce(buf, buf, n, ncclInt32, ncclSum, c->comm, c->ctx->stream));
    return PFZ_OK;
}

int comm_allreduce_sum_i64(pfz_comm *c, int64_t *buf, size_t n)
{
    if (c->group) return local_allreduce_sum<int64_t>(c, buf, n);
    PFZ_RCCL(ncclAllReduce(buf, buf, n, ncclInt64, ncclSum, c->comm, c->ctx->stream));
    return PFZ_OK;
}

}  // namespace pfz

using namespace pfz;

extern "C" {

int pfz_comm_unique_id(uint8_t id128[128])
{
    PFZ_REQUIRE(id128, "pfz_comm_unique_id: NULL buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId uid;
    PFZ_RCCL(ncclGetUniqueId(&uid));
    memcpy(id128, &uid, sizeof(uid));
    return PFZ_OK;
}

int pfz_comm_init(pfz_ctx *ctx, const uint8_t id128[128], int32_t rank, int32_t world, pfz_comm **out)
{
    PFZ_REQUIRE(ctx && id128 && out, "pfz_comm_init: NULL argument");
    PFZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "pfz_comm_init: rank %d of %d", rank, world);
    PFZ_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id128, sizeof(uid));
    pfz_comm *c = new pfz_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclResult_t r = ncclCommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return rccl_fail(r, "ncclCommInitRank", __LINE__);
    }
    Owner<pfz_comm, pfz_comm_destroy> guard(c);
    PFZ_TRY(pool_alloc(ctx, &c->flag, sizeof(int32_t)));
    PFZ_HIP(hipMemsetAsync(c->flag, 0, sizeof(int32_t), ctx->stream));
    *out = guard.release();
    return PFZ_OK;
}

int pfz_comm_group_create(int32_t world, pfz_comm_group **out)
{
    PFZ_REQUIRE(out && world >= 1 && world <= 64, "pfz_comm_group_create: world must be in 1..64");
    pfz_comm_group *g = new pfz_comm_group();
    g->world = world;
    g->send.assign((size_t)world, nullptr);
    g->ready.assign((size_t)world, nullptr);
    g->done.assign((size_t)world, nullptr);
    *out = g;
    return PFZ_OK;
}

void pfz_comm_group_destroy(pfz_comm_group *g)
{
    if (!g) return;
    for (hipEvent_t e : g->ready)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->done)
        if (e) (void)hipEventDestroy(e);
    delete g;
}

int pfz_comm_init_local(pfz_ctx *ctx, pfz_comm_group *g, int32_t rank, pfz_comm **out)
{
    PFZ_REQUIRE(ctx && g && out, "pfz_comm_init_local: NULL argument");
    PFZ_REQUIRE(rank >= 0 && rank < g->world, "pfz_comm_init_local: rank %d of %d", rank, g->world);
    PFZ_REQUIRE(g->ready[(size_t)rank] == nullptr, "pfz_comm_init_local: rank %d is already attached", rank);
    PFZ_HIP(hipSetDevice(ctx->device));
    PFZ_HIP(hipEventCreateWithFlags(&g->ready[(size_t)rank], hipEventDisableTiming));
    PFZ_HIP(hipEventCreateWithFlags(&g->done[(size_t)rank], hipEventDisableTiming));
    pfz_comm *c = new pfz_comm();
    c->ctx = ctx;
    c->group = g;
    c->rank = rank;
    c->world = g->world;
    *out = c;
    return PFZ_OK;
}

void pfz_comm_destroy(pfz_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->flag) pool_free(c->flag);
    delete c;
}

int pfz_comm_info(const pfz_comm *c, int32_t *rank, int32_t *world)
{
    PFZ_REQUIRE(c, "pfz_comm_info: NULL communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return PFZ_OK;
}

int pfz_comm_allgather_topn(pfz_comm *c, const pfz_topn *local, pfz_topn *global)
{
    PFZ_REQUIRE(c && local && global, "pfz_comm_allgather_topn: NULL argument");
    PFZ_REQUIRE(global->ntop == local->ntop && global->n_rows == local->n_rows * c->world,
                "pfz_comm_allgather_topn: global buffer is %lldx%d, need %lldx%d", (long long)global->n_rows,
                global->ntop, (long long)(local->n_rows * c->world), local->ntop);
    PFZ_HIP(hipSetDevice(c->ctx->device));
    const size_t count = (size_t)local->n_rows * (size_t)local->ntop;
    if (count == 0) return PFZ_OK;
    if (c->group) {
        PFZ_TRY(comm_allgather_bytes(c, local->idx, global->idx, count * sizeof(int32_t)));
        return comm_allgather_bytes(c, local->val, global->val, count * sizeof(float));
    }
    PFZ_RCCL(ncclGroupStart());
    PFZ_RCCL(ncclAllGather(local->idx, global->idx, count, ncclInt32, c->comm, c->ctx->stream));
    PFZ_RCCL(ncclAllGather(local->val, global->val, count, ncclFloat32, c->comm, c->ctx->stream));
    PFZ_RCCL(ncclGroupEnd());
    return PFZ_OK;
}

int pfz_comm_barrier(pfz_comm *c)
{
    PFZ_REQUIRE(c, "pfz_comm_barrier: NULL communicator");
    PFZ_HIP(hipSetDevice(c->ctx->device));
    if (c->group) {
        PFZ_HIP(hipStreamSynchronize(c->ctx->stream));
        c->group->barrier();
        return PFZ_OK;
    }
    PFZ_RCCL(ncclAllReduce(c->flag, c->flag, 1, ncclInt32, ncclSum, c->comm, c->ctx->stream));
    PFZ_HIP(hipStreamSynchronize(c->ctx->stream));
    return PFZ_OK;
}

}  // extern "C"
