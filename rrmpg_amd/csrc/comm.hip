// comm.hip -- the job's one collective inside the C-ABI.
//
// The reference has no distributed code; its scalable axis, the loop over
// parameter sets (rrmpg/models/hbvedu.py:199-209, tools/monte_carlo.py:61-71),
// shards into contiguous blocks, one per process / GPU, with no exchange on
// the data path.  What a sharded Monte-Carlo job exchanges is ONE all-gather
// of the per-set scores (8 bytes per set) -- in Python that is
// rrmpg_amd.sharding.allgather_scores over torch.distributed (backend "nccl"
// = RCCL).  A binder that has neither Python nor torch gets the same exchange
// here, over RCCL / xGMI directly:
//
//   rr_comm_unique_id  (rank 0; the 128 bytes travel out of band)
//   rr_comm_init       (every rank, one per GPU: ncclCommInitRank)
//   rr_allgather_metric(comm, local block, n_total scores out, stream)
//   rr_comm_destroy
//
// ... or, for ONE process that drives several GPUs (SURVEY.md 8e's sketch:
// ncclCommInitAll + ncclGroupStart / End):
//
//   rr_comm_init_all   (comms[ndev], one per device of `devices`)
//   rr_comm_group_start; rr_allgather_metric(comms[j], ..., stream_j) for
//   every j from the one thread; rr_comm_group_end
//
// The blocks are rrmpg_amd.sharding.shard_bounds' (contiguous, sizes differ
// by at most one, the first n_total % world ranks hold the longer ones).
// Equal blocks (n_total % world == 0 -- every BASELINE configuration) are ONE
// ncclAllGather; ragged blocks rule out its equal counts, so that exchange is
// one group of `world` broadcasts, rank r the root of block r, each landing
// in its place of `all` -- no padding, no staging buffer, nothing allocated.
// Asynchronous on `stream` like the *_simulate_dev family.
//
// librccl is opened at first use (dlopen "librccl.so.1": the copy a host
// program -- PyTorch, say -- has already loaded is the one that answers),
// so librrhip.so itself carries no link-time dependency on it and every other
// entry point works on a box without RCCL.  Nor does the BUILD need RCCL's
// development headers: the handful of types and the seven entry points used
// are declared here (NCCL's stable C ABI: rccl.h ncclResult_t,
// ncclDataType_t ncclFloat64 = 8, NCCL_UNIQUE_ID_BYTES 128).
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[RR_COMM_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclFloat64 = 8;

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t,
                              int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t,
                              ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
// rrdbg_comm_inject: a table of stand-ins for the entry points (tests: the
// shape of the exchange -- which collective, which offsets, counts and roots
// -- checked on a box without a GPU)
Rccl g_injected;
bool g_use_injected = false;

const Rccl *rccl()
{
    if (g_use_injected) return &g_injected;
    std::call_once(g_rccl_once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) return;
        auto sym = [](const char *n) { return dlsym(g_rccl.handle, n); };
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
        g_rccl.CommInitRank =
            (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
        g_rccl.CommInitAll =
            (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
        g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
        g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
        g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
        g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
        g_rccl.GetErrorString =
            (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
        g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank &&
                    g_rccl.CommDestroy && g_rccl.Broadcast &&
                    g_rccl.AllGather && g_rccl.GroupStart && g_rccl.GroupEnd &&
                    g_rccl.CommInitAll;
    });
    return g_rccl.ok ? &g_rccl : nullptr;
}

struct RrComm {
    ncclComm_t comm;
    int world, rank;
};

int fail(const Rccl *r, const char *what, ncclResult_t rc)
{
    rr_set_error("%s: %s", what,
                 r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
    return RR_E_HIP;
}
const Rccl *need_rccl(const char *who)
{
    const Rccl *r = rccl();
    if (!r) {
        const char *why = dlerror();      // (a second call would return NULL)
        rr_set_error("%s: librccl.so.1 could not be opened (or lacks an entry "
                     "point): %s", who, why ? why : "-");
    }
    return r;
}
}  // namespace

static_assert(sizeof(ncclUniqueId) == RR_COMM_ID_BYTES, "rrhip.h");

// Test hook (not part of include/rrhip.h): table = eight function pointers in
// the order {GetUniqueId, CommInitRank, CommDestroy, Broadcast, AllGather,
// GroupStart, GroupEnd, CommInitAll} that answer instead of librccl from now
// on; NULL restores the library.
extern "C" int rrdbg_comm_inject(void *const *table)
{
    if (!table) {
        g_use_injected = false;
        return RR_OK;
    }
    g_injected = Rccl();
    g_injected.GetUniqueId = (decltype(g_injected.GetUniqueId))table[0];
    g_injected.CommInitRank = (decltype(g_injected.CommInitRank))table[1];
    g_injected.CommDestroy = (decltype(g_injected.CommDestroy))table[2];
    g_injected.Broadcast = (decltype(g_injected.Broadcast))table[3];
    g_injected.AllGather = (decltype(g_injected.AllGather))table[4];
    g_injected.GroupStart = (decltype(g_injected.GroupStart))table[5];
    g_injected.GroupEnd = (decltype(g_injected.GroupEnd))table[6];
    g_injected.CommInitAll = (decltype(g_injected.CommInitAll))table[7];
    g_injected.ok = true;
    g_use_injected = true;
    return RR_OK;
}

extern "C" int rr_comm_unique_id(void *id_out)
{
    if (!id_out) {
        rr_set_error("rr_comm_unique_id: id_out is NULL");
        return RR_E_NULL;
    }
    const Rccl *r = need_rccl("rr_comm_unique_id");
    if (!r) return RR_E_NODEVICE;
    ncclUniqueId id;
    const ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return fail(r, "ncclGetUniqueId", rc);
    memcpy(id_out, id.internal, RR_COMM_ID_BYTES);
    return RR_OK;
}

extern "C" int rr_comm_init(void **comm_out, int world, int rank,
                            const void *id)
{
    if (!comm_out || !id) {
        rr_set_error("rr_comm_init: NULL argument");
        return RR_E_NULL;
    }
    *comm_out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) {
        rr_set_error("rr_comm_init: rank %d of %d", rank, world);
        return RR_E_SIZE;
    }
    const Rccl *r = need_rccl("rr_comm_init");
    if (!r) return RR_E_NODEVICE;
    ncclUniqueId uid;
    memcpy(uid.internal, id, RR_COMM_ID_BYTES);
    RrComm *c = new RrComm{nullptr, world, rank};
    const ncclResult_t rc = r->CommInitRank(&c->comm, world, uid, rank);
    if (rc != ncclSuccess) {
        delete c;
        return fail(r, "ncclCommInitRank", rc);
    }
    *comm_out = c;
    return RR_OK;
}

// One process, several GPUs: a clique of `ndev` communicators, comms_out[j]
// the rank-j communicator on device devices[j] (NULL: devices 0..ndev-1).
// RCCL refuses two ranks on one device, so the devices must differ.
extern "C" int rr_comm_init_all(void **comms_out, int ndev, const int *devices)
{
    if (!comms_out) {
        rr_set_error("rr_comm_init_all: comms_out is NULL");
        return RR_E_NULL;
    }
    if (ndev < 1 || ndev > 1024) {
        rr_set_error("rr_comm_init_all: %d devices", ndev);
        return RR_E_SIZE;
    }
    for (int j = 0; j < ndev; ++j) comms_out[j] = nullptr;
    if (devices)
        for (int j = 0; j < ndev; ++j)
            for (int k = 0; k < j; ++k)
                if (devices[j] == devices[k]) {
                    rr_set_error("rr_comm_init_all: device %d is listed twice "
                                 "(one rank per GPU)", devices[j]);
                    return RR_E_PARAM;
                }
    const Rccl *r = need_rccl("rr_comm_init_all");
    if (!r) return RR_E_NODEVICE;
    ncclComm_t *raw = new ncclComm_t[ndev];
    const ncclResult_t rc = r->CommInitAll(raw, ndev, devices);
    if (rc != ncclSuccess) {
        delete[] raw;
        return fail(r, "ncclCommInitAll", rc);
    }
    for (int j = 0; j < ndev; ++j) comms_out[j] = new RrComm{raw[j], ndev, j};
    delete[] raw;
    return RR_OK;
}

// Brackets the collectives one thread issues for several communicators of a
// clique (ncclGroupStart / ncclGroupEnd); they nest with the group
// rr_allgather_metric opens itself for ragged blocks.
extern "C" int rr_comm_group_start(void)
{
    const Rccl *r = need_rccl("rr_comm_group_start");
    if (!r) return RR_E_NODEVICE;
    const ncclResult_t rc = r->GroupStart();
    return rc == ncclSuccess ? RR_OK : fail(r, "ncclGroupStart", rc);
}
extern "C" int rr_comm_group_end(void)
{
    const Rccl *r = need_rccl("rr_comm_group_end");
    if (!r) return RR_E_NODEVICE;
    const ncclResult_t rc = r->GroupEnd();
    return rc == ncclSuccess ? RR_OK : fail(r, "ncclGroupEnd", rc);
}

extern "C" int rr_comm_destroy(void *comm)
{
    if (!comm) return RR_OK;
    RrComm *c = (RrComm *)comm;
    const Rccl *r = need_rccl("rr_comm_destroy");
    ncclResult_t rc = ncclSuccess;
    if (r) rc = r->CommDestroy(c->comm);
    delete c;
    if (r && rc != ncclSuccess) return fail(r, "ncclCommDestroy", rc);
    return RR_OK;
}

// [first, stop) of rank r's block (rrmpg_amd/sharding.py shard_bounds)
extern "C" int rr_shard_bounds(int64_t n_total, int world, int rank,
                               int64_t *first, int64_t *stop)
{
    if (!first || !stop) {
        rr_set_error("rr_shard_bounds: NULL output");
        return RR_E_NULL;
    }
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world) {
        rr_set_error("rr_shard_bounds: %lld sets, rank %d of %d",
                     (long long)n_total, rank, world);
        return RR_E_SIZE;
    }
    const int64_t base = n_total / world, extra = n_total % world;
    *first = rank * base + (rank < extra ? rank : extra);
    *stop = *first + base + (rank < extra ? 1 : 0);
    return RR_OK;
}

extern "C" int rr_allgather_metric(void *comm, const double *local,
                                   int64_t n_local, double *all,
                                   int64_t n_total, void *stream)
{
    const char *who = "rr_allgather_metric";
    if (!comm || !all || (!local && n_local > 0)) {
        rr_set_error("%s: NULL argument", who);
        return RR_E_NULL;
    }
    RrComm *c = (RrComm *)comm;
    int64_t first = 0, stop = 0;
    int rc = rr_shard_bounds(n_total, c->world, c->rank, &first, &stop);
    if (rc != RR_OK) return rc;
    if (n_local != stop - first) {
        rr_set_error("%s: rank %d of %d holds %lld scores, its block of %lld "
                     "has %lld", who, c->rank, c->world, (long long)n_local,
                     (long long)n_total, (long long)(stop - first));
        return RR_E_SIZE;
    }
    const Rccl *r = need_rccl(who);
    if (!r) return RR_E_NODEVICE;
    hipStream_t st = (hipStream_t)stream;
    if (n_total % c->world == 0) {
        // equal blocks: the one all-gather (in place when local is this
        // rank's block of `all`, as ncclAllGather defines in place)
        if (n_local == 0) return RR_OK;
        const ncclResult_t grc = r->AllGather(local, all, (size_t)n_local,
                                              ncclFloat64, c->comm, st);
        if (grc != ncclSuccess) return fail(r, "ncclAllGather", grc);
        return RR_OK;
    }
    ncclResult_t nrc = r->GroupStart();
    if (nrc != ncclSuccess) return fail(r, "ncclGroupStart", nrc);
    for (int root = 0; root < c->world; ++root) {
        int64_t a = 0, b = 0;
        (void)rr_shard_bounds(n_total, c->world, root, &a, &b);
        if (b == a) continue;
        nrc = r->Broadcast(root == c->rank ? (const void *)local
                                           : (const void *)(all + a),
                           all + a, (size_t)(b - a), ncclFloat64, root,
                           c->comm, st);
        if (nrc != ncclSuccess) {
            (void)r->GroupEnd();
            return fail(r, "ncclBroadcast", nrc);
        }
    }
    nrc = r->GroupEnd();
    if (nrc != ncclSuccess) return fail(r, "ncclGroupEnd", nrc);
    return RR_OK;
}
