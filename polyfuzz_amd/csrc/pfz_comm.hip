// Multi-GPU exchange: one process per GPU, RCCL over xGMI, collectives issued on
// the context's own stream.
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

struct pfz_comm {
    pfz_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int32_t *flag = nullptr;  // device, barrier payload
};

namespace pfz {

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
    PFZ_RCCL(ncclAllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, c->ctx->stream));
    return PFZ_OK;
}

int comm_allreduce_sum_i32(pfz_comm *c, int32_t *buf, size_t n)
{
    PFZ_RCCL(ncclAllReduce(buf, buf, n, ncclInt32, ncclSum, c->comm, c->ctx->stream));
    return PFZ_OK;
}

int comm_allreduce_sum_i64(pfz_comm *c, int64_t *buf, size_t n)
{
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
    PFZ_RCCL(ncclAllReduce(c->flag, c->flag, 1, ncclInt32, ncclSum, c->comm, c->ctx->stream));
    PFZ_HIP(hipStreamSynchronize(c->ctx->stream));
    return PFZ_OK;
}

}  // extern "C"
