// In-node parameter-server surface for embedding tables: the engine behind the libps names
// (ps-lite/src/python_binding.cc:6-151) that python/hetu binds with ctypes -- InitTensor, SparsePull,
// SparsePush, SSPushPull, Pull, Push, Wait, SaveParam, LoadParam, rank, nrank.  csrc/libps_shim.cpp exports
// those names (libherald_ps.so) and forwards here.
//
// Reference: a worker hands (node id, index DLArray, value DLArray) to its PSAgent, which dedups the ids,
// routes them to the servers that own their row ranges (AveragePartitioner, partitioner.h:46-57) and
// scatters / reduces rows (PSAgent.h:124-237); servers keep the shards (PSFHandle.h:101-164, 401-439).
// Here every process owns the row range of its rank in ITS GPU's HBM:
//   * nrank == 1 (or no backend registered): everything is local -- SparsePull is the gather kernel,
//     SparsePush the occurrence-ordered dedup-reduce + `+=` apply (ha_push_apply), both asynchronous on the
//     node's stream; Wait(node) joins it (worker.cc:189-197);
//   * nrank > 1: the exchange is the all-to-all of herald_amd/sharded.py (RCCL over xGMI through
//     torch.distributed); herald_amd/ps.py registers it as this engine's backend (ha_ps_set_backend), the
//     shard itself still lives here (ha_ps_tensor) and is served by the same kernels.
// Index / value arrays may be device arrays (used in place) or host arrays (staged on the node's stream).
#include "common.h"

#include <math.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace ha {

struct PsNode {
    int ptype = 0;
    int64_t len = 0, width = 0, row_start = 0, rows_local = 0;
    float *table = nullptr;
    bool owned = false;
    hipStream_t stream = nullptr;
    void *plan_ws = nullptr;
    size_t plan_cap = 0;
    // staging for host index / value arrays
    float *st_idx = nullptr, *st_val = nullptr;
    size_t st_idx_cap = 0, st_val_cap = 0;
};

static std::mutex g_ps_mu;
static std::unordered_map<int, PsNode> g_ps_nodes;
static int g_ps_rank = 0, g_ps_nrank = 1;
static ha_ps_backend g_ps_backend = {nullptr, nullptr, nullptr};

static PsNode *ps_find(int node) {
    auto it = g_ps_nodes.find(node);
    return it == g_ps_nodes.end() ? nullptr : &it->second;
}

static void ps_partition(int64_t len, int nrank, int rank, int64_t *start, int64_t *rows) {
    const int64_t per = len / nrank, rem = len % nrank;   // partitioner.h:46-57
    *start = per * rank + (rank < rem ? rank : rem);
    *rows = per + (rank < rem ? 1 : 0);
}

// counter-based generator: element i of a tensor depends on (seed, i) only, so a shard initialises its own
// rows to the values the whole table would have (the reference's servers seed std engines per shard,
// param.h; its values are not reproducible across partitionings, these are)
__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float unit_open(uint64_t r) {   // (0, 1)
    return (static_cast<float>(r >> 40) + 0.5f) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void ps_init_kernel(float *__restrict__ t, uint64_t first_elem, uint64_t count,
                                                      int init_type, float a, float b, uint64_t seed) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; e < count; e += stride) {
        const uint64_t g = first_elem + e;
        float v;
        if (init_type == 0) {                       // Constant(a)
            v = a;
        } else if (init_type == 1) {                // Uniform(a, b)
            v = a + (b - a) * unit_open(splitmix(seed ^ splitmix(g)));
        } else {                                    // Normal(mean a, stddev b) / TruncatedNormal (|z| <= 2)
            float z = 0.f;
            for (uint64_t attempt = 0; attempt < 16; ++attempt) {
                const uint64_t r1 = splitmix(seed ^ splitmix(g * 32 + attempt * 2));
                const uint64_t r2 = splitmix(seed ^ splitmix(g * 32 + attempt * 2 + 1));
                z = sqrtf(-2.f * logf(unit_open(r1))) * cosf(6.28318530718f * unit_open(r2));
                if (init_type == 2 || fabsf(z) <= 2.f)
                    break;
            }
            v = a + b * z;
        }
        t[e] = v;
    }
}

static int ps_grow(void **p, size_t *cap, size_t bytes, hipStream_t s) {
    if (*cap >= bytes)
        return 0;
    if (*p) {
        HA_CHECK_HIP(hipStreamSynchronize(s));
        HA_CHECK_HIP(hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    const size_t want = bytes + bytes / 4 + 256;
    HA_CHECK_HIP(hipMalloc(p, want));
    *cap = want;
    return 0;
}

// device view of an index / value DLArray: in place for GPU arrays, staged for host arrays
static int ps_in(PsNode &nd, const DLArray *a, bool is_index, const float **out) {
    HA_REQUIRE(a && a->data, "ps: null array");
    if (a->ctx.device_type == kGPU) {
        *out = static_cast<const float *>(a->data);
        return 0;
    }
    const size_t bytes = static_cast<size_t>(dl_numel(a)) * 4;
    void **buf = reinterpret_cast<void **>(is_index ? &nd.st_idx : &nd.st_val);
    size_t *cap = is_index ? &nd.st_idx_cap : &nd.st_val_cap;
    if (ps_grow(buf, cap, bytes, nd.stream))
        return -1;
    HA_CHECK_HIP(hipMemcpyAsync(*buf, a->data, bytes, hipMemcpyHostToDevice, nd.stream));
    *out = static_cast<const float *>(*buf);
    return 0;
}

static int ps_local_pull(PsNode &nd, const float *ids, int64_t n, float *out) {
    return ha_gather_f32ids(nd.table, nd.rows_local, nd.width, ids, n, out, nd.stream);
}

static int ps_local_push(PsNode &nd, const float *ids, int64_t n, const float *vals) {
    if (n == 0)
        return 0;
    if (ps_grow(&nd.plan_ws, &nd.plan_cap, ha_plan_bytes(n), nd.stream))
        return -1;
    if (ha_plan_sort_f32ids(ids, n, nd.plan_ws, nd.stream))
        return -1;
    // values of equal ids reduced in position order from 0, then `+=` (PSAgent.h:146-160, PSFHandle.h:130-164)
    return ha_push_apply(nd.table, nd.rows_local, nd.width, nd.plan_ws, n, vals, nd.stream);
}

}  // namespace ha

using namespace ha;

extern "C" int ha_ps_configure(int rank, int nrank) {
    HA_REQUIRE(nrank >= 1 && rank >= 0 && rank < nrank, "ha_ps_configure: bad rank %d / %d", rank, nrank);
    std::lock_guard<std::mutex> lk(g_ps_mu);
    HA_REQUIRE(g_ps_nodes.empty(), "ha_ps_configure: tensors exist already");
    g_ps_rank = rank;
    g_ps_nrank = nrank;
    return 0;
}

extern "C" int ha_ps_set_backend(const ha_ps_backend *b) {
    std::lock_guard<std::mutex> lk(g_ps_mu);
    if (b)
        g_ps_backend = *b;
    else
        g_ps_backend = ha_ps_backend{nullptr, nullptr, nullptr};
    return 0;
}

extern "C" int ha_ps_rank(void) { return g_ps_rank; }
extern "C" int ha_ps_nrank(void) { return g_ps_nrank; }

extern "C" int ha_ps_init_tensor(int node, int ptype, int64_t len, int64_t width, int init_type, double a,
                                 double b, uint64_t seed) {
    HA_REQUIRE(len > 0 && width > 0 && init_type >= 0 && init_type <= 3, "InitTensor: bad arguments");
    std::lock_guard<std::mutex> lk(g_ps_mu);
    if (ps_find(node))
        return 0;   // a second worker's InitTensor of the same node is a no-op (try_init_with_no_conflict)
    PsNode nd;
    nd.ptype = ptype;
    nd.len = len;
    nd.width = width;
    ps_partition(len, g_ps_nrank, g_ps_rank, &nd.row_start, &nd.rows_local);
    HA_CHECK_HIP(hipStreamCreateWithFlags(&nd.stream, hipStreamNonBlocking));
    const uint64_t count = static_cast<uint64_t>(nd.rows_local) * static_cast<uint64_t>(width);
    HA_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&nd.table), (count ? count : 1) * 4));
    nd.owned = true;
    if (count) {
        uint64_t blocks = (count + 255) / 256;
        if (blocks > 65536)
            blocks = 65536;
        hipLaunchKernelGGL(ps_init_kernel, dim3((unsigned)blocks), dim3(256), 0, nd.stream, nd.table,
                           static_cast<uint64_t>(nd.row_start) * static_cast<uint64_t>(width), count, init_type,
                           static_cast<float>(a), static_cast<float>(b), seed);
        HA_LAUNCH_CHECK();
    }
    g_ps_nodes[node] = nd;
    return 0;
}

extern "C" int ha_ps_attach_tensor(int node, float *table_dev, int64_t len, int64_t width) {
    HA_REQUIRE(table_dev && len > 0 && width > 0, "ha_ps_attach_tensor: bad arguments");
    std::lock_guard<std::mutex> lk(g_ps_mu);
    HA_REQUIRE(!ps_find(node), "ha_ps_attach_tensor: node %d exists", node);
    PsNode nd;
    nd.ptype = 1;
    nd.len = len;
    nd.width = width;
    ps_partition(len, g_ps_nrank, g_ps_rank, &nd.row_start, &nd.rows_local);
    nd.table = table_dev;
    HA_CHECK_HIP(hipStreamCreateWithFlags(&nd.stream, hipStreamNonBlocking));
    g_ps_nodes[node] = nd;
    return 0;
}

extern "C" int ha_ps_tensor(int node, ha_ps_tensor_info *out) {
    std::lock_guard<std::mutex> lk(g_ps_mu);
    PsNode *nd = ps_find(node);
    HA_REQUIRE(nd && out, "ha_ps_tensor: unknown node %d", node);
    out->table = nd->table;
    out->len = nd->len;
    out->width = nd->width;
    out->row_start = nd->row_start;
    out->rows_local = nd->rows_local;
    out->stream = nd->stream;
    return 0;
}

static int ps_check_value(const PsNode &nd, const DLArray *index, const DLArray *value, const char *what) {
    HA_REQUIRE(index && value && index->data && value->data, "%s: null array", what);
    HA_REQUIRE(dl_numel(value) == dl_numel(index) * nd.width, "%s: value size != index size x width", what);
    return 0;
}

extern "C" int ha_ps_sparse_pull(int node, const DLArray *index, DLArray *value) {
    std::lock_guard<std::mutex> lk(g_ps_mu);
    PsNode *nd = ps_find(node);
    HA_REQUIRE(nd, "SparsePull: unknown node %d", node);
    if (ps_check_value(*nd, index, value, "SparsePull"))
        return -1;
    const int64_t n = dl_numel(index);
    const float *ids = nullptr;
    if (ps_in(*nd, index, true, &ids))
        return -1;
    const bool host_out = value->ctx.device_type != kGPU;
    float *out = static_cast<float *>(value->data);
    if (host_out) {
        if (ps_grow(reinterpret_cast<void **>(&nd->st_val), &nd->st_val_cap, static_cast<size_t>(n) * nd->width * 4,
                    nd->stream))
            return -1;
        out = nd->st_val;
    }
    // with several ranks this process holds ONE row range of the table: global ids can only be served by the
    // registered exchange (herald_amd.ps.attach_sharded); the local kernels would address the shard with them
    HA_REQUIRE(g_ps_nrank == 1 || g_ps_backend.sparse_pull,
               "SparsePull: %d ranks configured but no sharded backend registered (ha_ps_set_backend)", g_ps_nrank);
    int rc;
    if (g_ps_nrank > 1)
        rc = g_ps_backend.sparse_pull(node, ids, n, out, nd->stream);
    else
        rc = ps_local_pull(*nd, ids, n, out);
    if (rc)
        return -1;
    if (host_out)
        HA_CHECK_HIP(hipMemcpyAsync(value->data, out, static_cast<size_t>(n) * nd->width * 4, hipMemcpyDeviceToHost,
                                    nd->stream));
    return 0;
}

extern "C" int ha_ps_sparse_push(int node, const DLArray *index, const DLArray *value) {
    std::lock_guard<std::mutex> lk(g_ps_mu);
    PsNode *nd = ps_find(node);
    HA_REQUIRE(nd, "SparsePush: unknown node %d", node);
    if (ps_check_value(*nd, index, value, "SparsePush"))
        return -1;
    const int64_t n = dl_numel(index);
    const float *ids = nullptr, *vals = nullptr;
    if (ps_in(*nd, index, true, &ids) || ps_in(*nd, value, false, &vals))
        return -1;
    HA_REQUIRE(g_ps_nrank == 1 || g_ps_backend.sparse_push,
               "SparsePush: %d ranks configured but no sharded backend registered (ha_ps_set_backend)", g_ps_nrank);
    if (g_ps_nrank > 1)
        return g_ps_backend.sparse_push(node, ids, n, vals, nd->stream);
    return ps_local_push(*nd, ids, n, vals);
}

extern "C" int ha_ps_dense_pull(int node, DLArray *arr) {
    std::lock_guard<std::mutex> lk(g_ps_mu);
    PsNode *nd = ps_find(node);
    HA_REQUIRE(nd && arr && arr->data, "Pull: unknown node %d or null array", node);
    HA_REQUIRE(g_ps_nrank == 1, "Pull: dense tensors are not sharded by this engine (nrank = %d)", g_ps_nrank);
    HA_REQUIRE(dl_numel(arr) == nd->len * nd->width, "Pull: size mismatch");
    HA_CHECK_HIP(hipMemcpyAsync(arr->data, nd->table, static_cast<size_t>(nd->len) * nd->width * 4,
                                arr->ctx.device_type == kGPU ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                nd->stream));
    return 0;
}

extern "C" int ha_ps_wait(int node) {
    hipStream_t s = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ps_mu);
        PsNode *nd = ps_find(node);
        HA_REQUIRE(nd, "Wait: unknown node %d", node);
        s = nd->stream;
    }
    HA_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

extern "C" int ha_ps_barrier(void) {
    if (g_ps_nrank > 1 && g_ps_backend.barrier)
        return g_ps_backend.barrier();
    return 0;
}

extern "C" int ha_ps_clear(int node) {
    std::lock_guard<std::mutex> lk(g_ps_mu);
    PsNode *nd = ps_find(node);
    if (!nd)
        return 0;
    (void)hipStreamSynchronize(nd->stream);
    if (nd->owned && nd->table)
        (void)hipFree(nd->table);
    if (nd->plan_ws)
        (void)hipFree(nd->plan_ws);
    if (nd->st_idx)
        (void)hipFree(nd->st_idx);
    if (nd->st_val)
        (void)hipFree(nd->st_val);
    (void)hipStreamDestroy(nd->stream);
    g_ps_nodes.erase(node);
    return 0;
}

// `<address>/<node>_<part>.dat`, raw fp32 rows of this rank's range (PSAgent.h:447-476, PSFHandle.h:401-439),
// streamed through a 64 MiB pinned buffer
static int ps_file(int node, const char *address, bool save) {
    PsNode nd;
    {
        std::lock_guard<std::mutex> lk(g_ps_mu);
        PsNode *p = ps_find(node);
        HA_REQUIRE(p && address, "%s: unknown node %d", save ? "SaveParam" : "LoadParam", node);
        nd = *p;
    }
    const std::string path = std::string(address) + "/" + std::to_string(node) + "_" + std::to_string(g_ps_rank) + ".dat";
    FILE *f = fopen(path.c_str(), save ? "wb" : "rb");
    HA_REQUIRE(f, "%s: cannot open %s", save ? "SaveParam" : "LoadParam", path.c_str());
    const size_t total = static_cast<size_t>(nd.rows_local) * nd.width * 4;
    const size_t chunk = 64u << 20;
    void *stage = nullptr;
    if (hipHostMalloc(&stage, chunk < total ? chunk : (total ? total : 4), hipHostMallocDefault) != hipSuccess) {
        fclose(f);
        set_error("SaveParam/LoadParam: cannot allocate the staging buffer");
        return -1;
    }
    int rc = 0;
    char *dev = reinterpret_cast<char *>(nd.table);
    for (size_t off = 0; off < total && rc == 0; off += chunk) {
        const size_t nb = total - off < chunk ? total - off : chunk;
        if (save) {
            if (hipMemcpyAsync(stage, dev + off, nb, hipMemcpyDeviceToHost, nd.stream) != hipSuccess ||
                hipStreamSynchronize(nd.stream) != hipSuccess || fwrite(stage, 1, nb, f) != nb)
                rc = -1;
        } else {
            if (fread(stage, 1, nb, f) != nb ||
                hipMemcpyAsync(dev + off, stage, nb, hipMemcpyHostToDevice, nd.stream) != hipSuccess ||
                hipStreamSynchronize(nd.stream) != hipSuccess)
                rc = -1;
        }
    }
    (void)hipHostFree(stage);
    fclose(f);
    if (rc)
        set_error("%s: I/O error on %s", save ? "SaveParam" : "LoadParam", path.c_str());
    return rc;
}

extern "C" int ha_ps_save(int node, const char *address) { return ps_file(node, address, true); }
extern "C" int ha_ps_load(int node, const char *address) { return ps_file(node, address, false); }
