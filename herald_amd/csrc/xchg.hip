// ha_xchg_*: the row exchange of the sharded sparse pull / push (PSAgent::vecPullSparse / vecPushSparse,
// /root/reference/ps-lite/include/ps/worker/PSAgent.h:124-237: U_s keys and U_s x d floats per server) as RCCL point-to-point
// calls made BY THE LIBRARY, on the caller's stream.
//
// The reference's worker sends and receives inside C++ (PSAgent, ZeroMQ vans); Python only enqueues.  Round 4 here went
// through torch.distributed.all_to_all_single: a Python call with split-size lists per exchange, on c10d's own stream with an
// event hop there and back -- 25-40 us of host time and four stream dependencies for a 700 KB message.  This file is the same
// exchange -- one ncclGroupStart / ncclSend + ncclRecv per peer that names rows / ncclGroupEnd, exactly what c10d issues for an
// all-to-all with split sizes -- from one C call, with the per-peer row counts read from the pinned words the routing left:
// no Python, no second stream, capturable into a hipGraph with the launches around it.
//
// RCCL is resolved at RUN TIME (dlopen): the library the process already has (PyTorch ships its own librccl.so; loading a
// second copy beside it would give two sets of RCCL globals) or /opt/rocm's.  A process that never creates an exchange
// never touches RCCL; ha_xchg_available() says whether it could.  The communicator is the library's own (ncclCommInitRank
// with a unique id the caller broadcasts -- ha_xchg_unique_id on rank 0); herald_amd/sharded.py checks the exchange against
// torch.distributed's on its first use and keeps torch's if they disagree.
#include <dlfcn.h>

#include "common.h"

namespace {

constexpr int kIdBytes = 128;      // NCCL_UNIQUE_ID_BYTES
struct UniqueId {
    char internal[kIdBytes];
};
typedef void *Comm;
typedef int Result;                // ncclResult_t (0 = ncclSuccess)
constexpr int kUint8 = 1;          // ncclUint8

struct Api {
    void *lib = nullptr;
    Result (*GetUniqueId)(UniqueId *) = nullptr;
    Result (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    Result (*GroupStart)() = nullptr;
    Result (*GroupEnd)() = nullptr;
    Result (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    Result (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(Result) = nullptr;
    bool ok = false;
};

Api &api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy the process has already loaded, if any (RTLD_NOLOAD finds it by its soname), else the system's
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (a.lib)
                break;
        }
        for (int i = 0; !a.lib && i < 3; ++i)
            a.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!a.lib)
            return;
        auto sym = [&](const char *s) { return dlsym(a.lib, s); };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
        a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv;
    });
    return a;
}

struct Xchg {
    Comm comm = nullptr;
    int world = 0, rank = 0;
};

#define HA_CHECK_NCCL(expr)                                                                                       \
    do {                                                                                                          \
        Result _r = (expr);                                                                                       \
        if (_r != 0) {                                                                                            \
            ::ha::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                                         \
                            api().GetErrorString ? api().GetErrorString(_r) : "RCCL error");                      \
            return -1;                                                                                            \
        }                                                                                                         \
    } while (0)

}  // namespace

using namespace ha;

// 1 if an RCCL library could be resolved in this process
extern "C" int ha_xchg_available(void) { return api().ok ? 1 : 0; }

// rank 0: a fresh unique id (128 bytes) for the other ranks (the caller carries it over: torch.distributed.broadcast)
extern "C" int ha_xchg_unique_id(void *id128) {
    HA_REQUIRE(id128 != nullptr, "ha_xchg_unique_id: null buffer");
    HA_REQUIRE(api().ok, "ha_xchg_unique_id: no RCCL library in this process");
    HA_CHECK_NCCL(api().GetUniqueId(static_cast<UniqueId *>(id128)));
    return 0;
}

// every rank, collectively: the exchange object of `world` ranks on the CURRENT device
extern "C" void *ha_xchg_create(const void *id128, int world, int rank) {
    if (!api().ok || id128 == nullptr || world < 1 || rank < 0 || rank >= world) {
        set_error("ha_xchg_create: bad arguments or no RCCL library");
        return nullptr;
    }
    UniqueId id;
    memcpy(&id, id128, kIdBytes);
    Xchg *x = new Xchg();
    x->world = world;
    x->rank = rank;
    const Result r = api().CommInitRank(&x->comm, world, id, rank);
    if (r != 0) {
        set_error("ha_xchg_create: ncclCommInitRank -> %s", api().GetErrorString ? api().GetErrorString(r) : "RCCL error");
        delete x;
        return nullptr;
    }
    return x;
}

extern "C" int ha_xchg_destroy(void *h) {
    Xchg *x = static_cast<Xchg *>(h);
    if (x == nullptr)
        return 0;
    if (x->comm)
        (void)api().CommDestroy(x->comm);
    delete x;
    return 0;
}

// All-to-all of BYTES with per-peer counts: `send` holds the bytes for peer 0, 1, ... back to back (send_bytes[g] each), `recv`
// receives peer 0's, 1's, ... back to back (recv_bytes[g]).  The counts of this rank itself are normally zero (its own
// rows never enter an exchange); if not, they must be equal and are copied on the stream.  Asynchronous on `stream`.
extern "C" int ha_xchg_bytes(void *h, const void *send, const int64_t *send_bytes, void *recv, const int64_t *recv_bytes,
                             ha_stream_t stream) {
    Xchg *x = static_cast<Xchg *>(h);
    HA_REQUIRE(x != nullptr && x->comm != nullptr && send_bytes && recv_bytes, "ha_xchg_bytes: bad arguments");
    hipStream_t s = as_stream(stream);
    const char *sp = static_cast<const char *>(send);
    char *rp = static_cast<char *>(recv);
    int64_t so = 0, ro = 0;
    bool any = false;
    // every argument is checked BEFORE the group is opened, and an error inside it closes the group first: a thread left
    // inside an open RCCL group queues every later collective of the process (c10d's too) without ever launching it
    for (int g = 0; g < x->world; ++g) {
        HA_REQUIRE(send_bytes[g] >= 0 && recv_bytes[g] >= 0, "ha_xchg_bytes: negative count");
        any = any || (g != x->rank && (send_bytes[g] > 0 || recv_bytes[g] > 0));
    }
    HA_REQUIRE(send_bytes[x->rank] == 0 || send_bytes[x->rank] == recv_bytes[x->rank],
               "ha_xchg_bytes: this rank's own counts differ");
    HA_REQUIRE(!any || (send != nullptr && recv != nullptr), "ha_xchg_bytes: null buffers");
    if (send_bytes[x->rank] > 0) {
        int64_t so0 = 0, ro0 = 0;
        for (int g = 0; g < x->rank; ++g) {
            so0 += send_bytes[g];
            ro0 += recv_bytes[g];
        }
        HA_CHECK_HIP(hipMemcpyAsync(rp + ro0, sp + so0, static_cast<size_t>(send_bytes[x->rank]), hipMemcpyDeviceToDevice, s));
    }
    if (!any)
        return 0;
    HA_CHECK_NCCL(api().GroupStart());
    int bad = 0;
    for (int g = 0; g < x->world && !bad; ++g) {
        if (g != x->rank) {
            if (send_bytes[g] > 0)
                bad = api().Send(sp + so, static_cast<size_t>(send_bytes[g]), kUint8, g, x->comm, s);
            if (!bad && recv_bytes[g] > 0)
                bad = api().Recv(rp + ro, static_cast<size_t>(recv_bytes[g]), kUint8, g, x->comm, s);
        }
        so += send_bytes[g];
        ro += recv_bytes[g];
    }
    const int end = api().GroupEnd();
    HA_REQUIRE(bad == 0, "ha_xchg_bytes: ncclSend / ncclRecv failed (%d); the group was closed", bad);
    HA_CHECK_NCCL(end);
    return 0;
}

// The same with counts in ROWS of `width` floats (what the sized row exchanges of a step carry).
extern "C" int ha_xchg_rows(void *h, const float *send, const int64_t *send_rows, float *recv, const int64_t *recv_rows,
                            int64_t width, ha_stream_t stream) {
    Xchg *x = static_cast<Xchg *>(h);
    HA_REQUIRE(x != nullptr && send_rows && recv_rows && width >= 1 && x->world <= 1024, "ha_xchg_rows: bad arguments");
    int64_t sb[1024], rb[1024];
    for (int g = 0; g < x->world; ++g) {
        sb[g] = send_rows[g] * width * 4;
        rb[g] = recv_rows[g] * width * 4;
    }
    return ha_xchg_bytes(h, send, sb, recv, rb, stream);
}
