// vgmi_api_rccl.cpp -- the table image over RCCL (vgmi_rccl_unique_id, vgmi_comm_*, vgmi_table_broadcast*)
#include "vgmi_ctx.h"

extern "C" {

// ---- the table image over RCCL, for one process per GPU (the north_star's "single RCCL broadcast of the read-only graph index over
// xGMI"; the reference is single-device: main.cu:221,444 select one).  librccl is loaded on first use: the library itself carries
// no dependency on it, a node without RCCL still runs everything else.
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, ncclUniqueIdBytes, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;      // (optional: a rank that cannot go on ends its peers' collectives with it)
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl* rccl()
{
    static Rccl r = [] {
        Rccl x;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) {
            const char* m = dlerror();      // (once: the call hands the message over and clears it)
            x.err = std::string("librccl.so: ") + (m ? m : "not found");
            return x;
        }
        x.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(x.lib, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<int (*)(void**, int, ncclUniqueIdBytes, int)>(dlsym(x.lib, "ncclCommInitRank"));
        x.Broadcast = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(x.lib, "ncclBroadcast"));
        x.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(x.lib, "ncclAllReduce"));
        x.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(x.lib, "ncclCommDestroy"));
        x.CommAbort = reinterpret_cast<int (*)(void*)>(dlsym(x.lib, "ncclCommAbort"));
        x.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(x.lib, "ncclGetErrorString"));
        if (!x.GetUniqueId || !x.CommInitRank || !x.Broadcast || !x.AllReduce || !x.CommDestroy) x.err = "librccl.so lacks an entry point";
        return x;
    }();
    return &r;
}
}  // namespace

int vgmi_rccl_unique_id(void* id128)
{
    if (!id128) return VGMI_E_INVALID;
    Rccl* r = rccl();
    if (!r->err.empty()) return fail(nullptr, VGMI_E_STATE, r->err);
    const int rc = r->GetUniqueId(id128);
    if (rc) return fail(nullptr, VGMI_E_HIP, std::string("ncclGetUniqueId: ") + (r->GetErrorString ? r->GetErrorString(rc) : "failed"));
    return VGMI_OK;
}

// The communicator on its own: ncclCommInitRank takes seconds (topology, kernels of every rank's device) and needs neither a
// table nor a context -- a rank calls it beside its graph load / table build, and joins the broadcast when both are there.
struct vgmi_comm {
    void* comm = nullptr;
    int device = 0, rank = 0, world = 1;
};

int vgmi_comm_create(int device, int rank, int world, const void* id128, vgmi_comm** out)
{
    if (!out) return VGMI_E_INVALID;
    *out = nullptr;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return VGMI_E_INVALID;
    Rccl* r = rccl();
    if (!r->err.empty()) return fail(nullptr, VGMI_E_STATE, r->err);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, VGMI_E_HIP, "vgmi_comm_create: hipSetDevice failed");
    ncclUniqueIdBytes id;
    memcpy(id.internal, id128, sizeof id.internal);
    vgmi_comm* m = new (std::nothrow) vgmi_comm();
    if (!m) return fail(nullptr, VGMI_E_NOMEM, "out of memory");
    m->device = device;
    m->rank = rank;
    m->world = world;
    const int rc = r->CommInitRank(&m->comm, world, id, rank);
    if (rc) {
        delete m;
        return fail(nullptr, VGMI_E_HIP, std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(rc) : "failed"));
    }
    *out = m;
    return VGMI_OK;
}

void vgmi_comm_destroy(vgmi_comm* m)
{
    if (!m) return;
    if (m->comm) {
        (void)hipSetDevice(m->device);
        (void)rccl()->CommDestroy(m->comm);
    }
    delete m;
}

// Root = rank 0.  Every rank goes through the same three collectives -- the image's size (0: the root has none to give), an agreement
// that every receiver has its buffer (all-reduce, minimum), the image -- so that a rank that cannot go on says so to the others instead
// of leaving them inside a collective: "no table" and "no room" are told INSIDE the collectives and every rank returns together.  A rank
// that fails on its own between them (a HIP call, a collective's own error) ABORTS the communicator (ncclCommAbort) on its way out: its
// peers' pending collectives end with an error instead of waiting for it.  The communicator is unusable after that
// (vgmi_comm_destroy still takes it).
int vgmi_table_broadcast_comm(vgmi_ctx* c, vgmi_comm* m)
{
    if (!c || !m || !m->comm) return VGMI_E_INVALID;
    if (m->device != c->device) return fail(c, VGMI_E_INVALID, "vgmi_table_broadcast_comm: the communicator is on another device than the context");
    Rccl* r = rccl();
    struct AbortGuard {      // armed until this rank has said what it has to say inside the collectives
        Rccl* r; vgmi_comm* m; bool armed;
        ~AbortGuard() { if (armed && m->comm && r->CommAbort) { (void)r->CommAbort(m->comm); m->comm = nullptr; } }
    } guard{r, m, true};
    HIPCHK(c, hipSetDevice(c->device));
    const int rank = m->rank;
    // a stream of its own: a root that sends a snapshot may be counting on the context's streams meanwhile
    hipStream_t st = nullptr;
    HIPCHK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long* d_n = nullptr;      // [0] the size, [1] the receivers' agreement
    uint8_t* d_recv = nullptr;
    struct Cleanup {
        hipStream_t& st; unsigned long long*& d_n; uint8_t*& d_recv;
        ~Cleanup() { if (d_n) (void)hipFree(d_n); if (d_recv) (void)hipFree(d_recv); if (st) (void)hipStreamDestroy(st); }
    } cleanup{st, d_n, d_recv};
    if (!(rank == 0 && c->d_snapshot)) HIPCHK(c, hipStreamSynchronize(c->stream));
    auto nccl_text = [&](const char* what, int rc) { return std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "failed"); };
    // (16 bytes: if even that fails the device is gone, and so is this rank's part in the collectives)
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_n), 16));
    unsigned long long h[2] = {rank == 0 && c->has_table ? (unsigned long long)c->image_bytes : 0ull, 1ull};
    HIPCHK(c, hipMemcpy(d_n, h, 16, hipMemcpyHostToDevice));
    int rc = r->Broadcast(d_n, d_n, 8, /* ncclChar */ 0, 0, m->comm, st);
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = 1;
    if (rc) return fail(c, VGMI_E_HIP, nccl_text("ncclBroadcast (size)", rc));
    HIPCHK(c, hipMemcpy(h, d_n, 8, hipMemcpyDeviceToHost));
    const unsigned long long n = h[0];
    if (n == 0) {      // (every rank reads the same size and leaves here)
        guard.armed = false;
        return fail(c, VGMI_E_STATE, "the root has no table to broadcast");
    }
    uint8_t* d_buf = rank == 0 ? (c->d_snapshot ? c->d_snapshot : c->d_image) : nullptr;
    if (rank != 0) {
        if (hipMalloc(reinterpret_cast<void**>(&d_recv), n) != hipSuccess) {
            d_recv = nullptr;
            h[1] = 0;
            (void)hipGetLastError();
            HIPCHK(c, hipMemcpy(d_n + 1, h + 1, 8, hipMemcpyHostToDevice));
        }
        d_buf = d_recv;
    }
    rc = r->AllReduce(d_n + 1, d_n + 1, 1, /* ncclUint64 */ 5, /* ncclMin */ 3, m->comm, st);
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = 1;
    unsigned long long all_ready = 0;
    if (rc == 0 && hipMemcpy(&all_ready, d_n + 1, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
    if (rc) return fail(c, VGMI_E_HIP, nccl_text("ncclAllReduce (buffers)", rc));
    if (!all_ready) {      // (likewise)
        guard.armed = false;
        return fail(c, VGMI_E_NOMEM, "vgmi_table_broadcast_comm: a rank has no room for the table image");
    }
    rc = r->Broadcast(d_buf, d_buf, n, /* ncclChar */ 0, 0, m->comm, st);
    if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = 1;
    if (rc) return fail(c, VGMI_E_HIP, nccl_text("ncclBroadcast (image)", rc));
    guard.armed = false;
    if (rank == 0 && c->d_snapshot) {
        (void)hipFree(c->d_snapshot);
        c->d_snapshot = nullptr;
    }
    return rank != 0 ? vgmi_table_import(c, d_recv, n) : VGMI_OK;
}

int vgmi_table_broadcast(vgmi_ctx* c, int rank, int world, const void* id128)
{
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return VGMI_E_INVALID;
    vgmi_comm* m = nullptr;
    const int rc = vgmi_comm_create(c->device, rank, world, id128, &m);
    if (rc) return fail(c, rc, vgmi_last_error(nullptr));
    const int out = vgmi_table_broadcast_comm(c, m);
    vgmi_comm_destroy(m);
    return out;
}

}  // extern "C"
