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
    std::vector<int> vote;               // per rank: comm_agree
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

// Merge of per-to-shard candidates: row i has world x ntop candidates (idx local to the shard of rank p, -1 = none),
// the result is its ntop best by (score desc, GLOBAL to-index asc).  One wave per row; a lane holds up to 16 keys
// score_bits << 32 | ~global_idx (scores are positive floats: their bit patterns order like the values).
constexpr int kMergeKeysPerLane = 16;
__global__ __launch_bounds__(256) void k_merge_to_shards(const int32_t *__restrict__ g_idx, const float *__restrict__ g_val,
                                                          const int64_t *__restrict__ offsets, int32_t world, int64_t n_rows,
                                                          int32_t ntop, int32_t *__restrict__ out_idx, float *__restrict__ out_val)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int total = world * ntop;
    uint64_t e[kMergeKeysPerLane];
#pragma unroll
    for (int i = 0; i < kMergeKeysPerLane; ++i) {
        const int c = lane + 64 * i;
        uint64_t key = 0ull;
        if (c < total) {
            const int p = c / ntop, r = c - p * ntop;
            const int64_t src = ((int64_t)p * n_rows + row) * ntop + r;
            const int32_t j = g_idx[src];
            if (j >= 0) key = ((uint64_t)__float_as_uint(g_val[src]) << 32) | (uint32_t)(~(uint32_t)(j + offsets[p]));
        }
        e[i] = key;
    }
    for (int r = 0; r < ntop; ++r) {
        uint64_t m = e[0];
#pragma unroll
        for (int i = 1; i < kMergeKeysPerLane; ++i) m = e[i] > m ? e[i] : m;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t lo = __shfl_xor((uint32_t)m, d, 64), hi = __shfl_xor((uint32_t)(m >> 32), d, 64);
            const uint64_t o = ((uint64_t)hi << 32) | lo;
            m = o > m ? o : m;
        }
#pragma unroll
        for (int i = 0; i < kMergeKeysPerLane; ++i)
            if (e[i] == m) e[i] = 0ull;
        if (lane == 0) {
            out_idx[row * ntop + r] = m ? (int32_t)(~(uint32_t)m) : -1;
            out_val[row * ntop + r] = m ? __uint_as_float((uint32_t)(m >> 32)) : 0.f;
        }
    }
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
        char *dst = (char *)recv + (size_t)p * bytes_per_rank;
        if (dst != (const char *)g->send[p])      // (in place: this rank's stretch is where it belongs already)
            PFZ_HIP(hipMemcpyAsync(dst, g->send[p], bytes_per_rank, hipMemcpyDefault, st));
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

// Do ALL ranks say yes?  A host-side agreement (it waits): what a rank decides from its own environment or its own allocations --
// "this job takes the symmetric form" -- must be the same everywhere before any rank enters a collective the others would not.
int comm_agree(pfz_comm *c, bool mine, bool *all)
{
    *all = mine;
    if (!c || c->world == 1) return PFZ_OK;
    if (c->group) {
        pfz_comm_group *g = c->group;
        g->vote[(size_t)c->rank] = mine ? 1 : 0;
        g->barrier();
        bool yes = true;
        for (int v : g->vote) yes = yes && v != 0;
        g->barrier();                   // (nobody votes again before everybody has read)
        *all = yes;
        return PFZ_OK;
    }
    int32_t no = mine ? 0 : 1;
    PFZ_HIP(hipMemcpyAsync(c->flag, &no, sizeof(no), hipMemcpyHostToDevice, c->ctx->stream));
    PFZ_HIP(hipStreamSynchronize(c->ctx->stream));      // (`no` is a stack word)
    PFZ_RCCL(ncclAllReduce(c->flag, c->flag, 1, ncclInt32, ncclSum, c->comm, c->ctx->stream));
    PFZ_HIP(hipMemcpyAsync(&no, c->flag, sizeof(no), hipMemcpyDeviceToHost, c->ctx->stream));
    PFZ_HIP(hipStreamSynchronize(c->ctx->stream));
    PFZ_HIP(hipMemsetAsync(c->flag, 0, sizeof(int32_t), c->ctx->stream));      // (pfz_comm_barrier's payload again)
    *all = no == 0;
    return PFZ_OK;
}

// A rank that fails BEHIND an agreement cannot leave its peers inside a collective: the communicator is torn down, the peers'
// pending RCCL calls end in an error instead of a wait that never ends.
void comm_abort(pfz_comm *c)
{
    if (c && c->comm) {
        (void)ncclCommAbort(c->comm);
        c->comm = nullptr;
    }
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

int pfz_rccl_versions(int32_t *header, int32_t *runtime)
{
    int v = 0;
    PFZ_RCCL(ncclGetVersion(&v));
    if (header) *header = NCCL_VERSION_CODE;
    if (runtime) *runtime = v;
    return PFZ_OK;
}

int pfz_comm_init(pfz_ctx *ctx, const uint8_t id128[128], int32_t rank, int32_t world, pfz_comm **out)
{
    PFZ_REQUIRE(ctx && id128 && out, "pfz_comm_init: NULL argument");
    PFZ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "pfz_comm_init: rank %d of %d", rank, world);
    {   // the library was compiled against one rccl.h and runs on whatever librccl the process resolved (a torch wheel ships its
        // own): structures and enums are stable inside a major version only -- refuse loudly across one
        int v = 0;
        PFZ_RCCL(ncclGetVersion(&v));
        if (v / 10000 != NCCL_VERSION_CODE / 10000) {
            set_error("pfz_comm_init: compiled against rccl.h %d but the process resolved librccl %d: major versions differ",
                      (int)NCCL_VERSION_CODE, v);
            return PFZ_ERR_RCCL;
        }
    }
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
    g->vote.assign((size_t)world, 0);
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

int pfz_comm_merge_to_shards(pfz_comm *c, const pfz_topn *local, int64_t to_offset, pfz_topn *out)
{
    PFZ_REQUIRE(c && local && out, "pfz_comm_merge_to_shards: NULL argument");
    PFZ_REQUIRE(out->ntop == local->ntop && out->n_rows == local->n_rows,
                "pfz_comm_merge_to_shards: result buffer is %lldx%d, need %lldx%d", (long long)out->n_rows, out->ntop,
                (long long)local->n_rows, local->ntop);
    if ((int64_t)c->world * local->ntop > 64 * kMergeKeysPerLane) {
        set_error("pfz_comm_merge_to_shards: %d ranks x top-%d exceed the %d candidates a row merge holds", c->world,
                  local->ntop, 64 * kMergeKeysPerLane);
        return PFZ_ERR_UNSUPPORTED;
    }
    pfz_ctx *ctx = c->ctx;
    PFZ_HIP(hipSetDevice(ctx->device));
    const size_t count = (size_t)local->n_rows * (size_t)local->ntop;
    if (count == 0) return PFZ_OK;
    struct Buf {
        void *p = nullptr;
        ~Buf() { if (p) pool_free(p); }
    } g_idx, g_val, offs;
    PFZ_TRY(pool_alloc(ctx, &g_idx.p, (size_t)c->world * count * sizeof(int32_t)));
    PFZ_TRY(pool_alloc(ctx, &g_val.p, (size_t)c->world * count * sizeof(float)));
    PFZ_TRY(pool_alloc(ctx, &offs.p, (size_t)(c->world + 1) * sizeof(int64_t)));
    int64_t *mine = (int64_t *)offs.p + c->world;
    PFZ_TRY(copy_h2d(ctx, mine, &to_offset, sizeof(int64_t)));
    PFZ_TRY(comm_allgather_bytes(c, mine, offs.p, sizeof(int64_t)));
    PFZ_TRY(comm_allgather_bytes(c, local->idx, g_idx.p, count * sizeof(int32_t)));
    PFZ_TRY(comm_allgather_bytes(c, local->val, g_val.p, count * sizeof(float)));
    hipLaunchKernelGGL(k_merge_to_shards, dim3((unsigned)((local->n_rows + 3) / 4)), dim3(256), 0, ctx->stream,
                       (const int32_t *)g_idx.p, (const float *)g_val.p, (const int64_t *)offs.p, c->world, local->n_rows,
                       local->ntop, out->idx, out->val);
    PFZ_HIP(hipGetLastError());
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
