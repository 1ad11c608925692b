/* vslam_comm.hip -- the multi-GPU exchange step of the front-end, inside the library.
 *
 * Frames are dealt to the GPUs in BLOCKS: of a step's world x B consecutive frames rank r holds r*B .. (r+1)*B - 1
 * (vi_slam_amd/dist.py: global_frame(rank, slot) = rank * B + slot); extraction and L<->R stereo matching need no
 * collective.  Cross-frame matching (SearchForInitialization, SearchByProjection(Current, Last)) needs the predecessor
 * frame's packed result slot, which is the rank's own previous slot for every frame but the first of its block; slot 0
 * needs the LAST frame of the LEFT neighbour rank (for rank 0: rank world-1's last frame of the previous step).  So the
 * exchange is a ring shift of ONE packed slot per rank and step: one ncclSend to rank+1 and one ncclRecv from rank-1
 * inside ncclGroupStart/End, enqueued on the extractor context's own HIP stream right behind k_pack_slots -- the
 * matcher that follows on the same stream needs no host synchronisation.  vslam_exchange_allgather is the
 * north_star-literal variant (world times the volume).
 *
 * RCCL is resolved at run time (dlopen "librccl.so.1"): a single-GPU consumer of libvslam_fe.so needs only
 * libamdhip64, and inside a PyTorch process the already loaded RCCL is reused instead of a second copy.
 */
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "vslam_ctx.h"

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!api.handle) {
            api.err = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "");
            return;
        }
#define VSLAM_RCCL_SYM(field, name)                                                   \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name));       \
    if (!api.field && api.err.empty()) api.err = std::string("RCCL symbol missing: ") + name;
        VSLAM_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
        VSLAM_RCCL_SYM(CommInitRank, "ncclCommInitRank")
        VSLAM_RCCL_SYM(CommDestroy, "ncclCommDestroy")
        VSLAM_RCCL_SYM(GroupStart, "ncclGroupStart")
        VSLAM_RCCL_SYM(GroupEnd, "ncclGroupEnd")
        VSLAM_RCCL_SYM(Send, "ncclSend")
        VSLAM_RCCL_SYM(Recv, "ncclRecv")
        VSLAM_RCCL_SYM(AllGather, "ncclAllGather")
        VSLAM_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef VSLAM_RCCL_SYM
    });
    return api;
}

int rccl_fail(const char* what, ncclResult_t r) {
    RcclApi& A = rccl();
    g_err = std::string(what) + ": " + (A.GetErrorString ? A.GetErrorString(r) : "RCCL error");
    return VSLAM_ERR_COMM;
}
} // namespace

struct vslam_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

#define RCCLCHK(call)                                      \
    do {                                                   \
        ncclResult_t r_ = (call);                          \
        if (r_ != ncclSuccess) return rccl_fail(#call, r_); \
    } while (0)

static int need_rccl() {
    RcclApi& A = rccl();
    if (!A.err.empty() || !A.handle) {
        g_err = A.err.empty() ? "RCCL unavailable" : A.err;
        return VSLAM_ERR_COMM;
    }
    return VSLAM_OK;
}

extern "C" int vslam_comm_unique_id(uint8_t id[VSLAM_COMM_ID_BYTES]) {
    static_assert(VSLAM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id) return VSLAM_ERR_INVALID;
    int rc = need_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    RCCLCHK(rccl().GetUniqueId(&u));
    memcpy(id, u.internal, VSLAM_COMM_ID_BYTES);
    return VSLAM_OK;
}

extern "C" int vslam_comm_create(int device, int rank, int world, const uint8_t id[VSLAM_COMM_ID_BYTES],
                                 vslam_comm** out) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) {
        g_err = "invalid arguments";
        return VSLAM_ERR_INVALID;
    }
    *out = nullptr;
    int rc = need_rccl();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    ncclUniqueId u;
    memcpy(u.internal, id, VSLAM_COMM_ID_BYTES);
    vslam_comm* c = new vslam_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return rccl_fail("ncclCommInitRank", r);
    }
    *out = c;
    return VSLAM_OK;
}

extern "C" void vslam_comm_destroy(vslam_comm* c) {
    if (!c) return;
    if (c->comm && rccl().CommDestroy) {
        hipSetDevice(c->device);
        rccl().CommDestroy(c->comm);
    }
    delete c;
}

extern "C" int vslam_comm_rank(const vslam_comm* c) { return c ? c->rank : VSLAM_ERR_INVALID; }
extern "C" int vslam_comm_world(const vslam_comm* c) { return c ? c->world : VSLAM_ERR_INVALID; }

static int check_xchg(vslam_fe* fe, vslam_comm* c, const void* s, void* r, size_t bytes) {
    if (!fe || !c || !s || !r || !bytes || fe->p.device != c->device) {
        g_err = "invalid arguments (context and communicator must live on the same device)";
        return VSLAM_ERR_INVALID;
    }
    HIPCHK(hipSetDevice(fe->p.device));
    return VSLAM_OK;
}

/* Ring shift on fe's stream: dev_send -> rank+1, dev_recv <- rank-1.  World size 1 is a self send/recv pair through the
 * same calls (RCCL supports it inside a group), so the code path is the one N ranks take. */
extern "C" int vslam_exchange_ring(vslam_fe* fe, vslam_comm* c, const void* dev_send, void* dev_recv, size_t bytes) {
    int rc = check_xchg(fe, c, dev_send, dev_recv, bytes);
    if (rc) return rc;
    RcclApi& A = rccl();
    const int right = (c->rank + 1) % c->world, left = (c->rank + c->world - 1) % c->world;
    RCCLCHK(A.GroupStart());
    ncclResult_t rs = A.Send(dev_send, bytes, ncclUint8, right, c->comm, fe->stream);
    ncclResult_t rr = A.Recv(dev_recv, bytes, ncclUint8, left, c->comm, fe->stream);
    ncclResult_t re = A.GroupEnd(); /* always close the group, also after a failed Send/Recv */
    if (rs != ncclSuccess) return rccl_fail("ncclSend", rs);
    if (rr != ncclSuccess) return rccl_fail("ncclRecv", rr);
    if (re != ncclSuccess) return rccl_fail("ncclGroupEnd", re);
    return VSLAM_OK;
}

/* north_star-literal variant: every rank receives every rank's block (dev_recv_all: world x bytes_per_rank). */
extern "C" int vslam_exchange_allgather(vslam_fe* fe, vslam_comm* c, const void* dev_send, void* dev_recv_all,
                                        size_t bytes_per_rank) {
    int rc = check_xchg(fe, c, dev_send, dev_recv_all, bytes_per_rank);
    if (rc) return rc;
    RCCLCHK(rccl().AllGather(dev_send, dev_recv_all, bytes_per_rank, ncclUint8, c->comm, fe->stream));
    return VSLAM_OK;
}
