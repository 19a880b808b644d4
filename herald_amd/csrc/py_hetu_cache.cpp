// pybind11 module `hetu_cache`: the plugin surface of the reference (src/hetu_cache/src/python_api.cc:12-79)
// on top of the C-ABI of libherald_amd.so.  Host code only -- every computation is an ha_cache_* call.
//
//   LRUCache / LFUCache / LFUOptCache(limit, len, width, node_id)
//     .limit .width .perf .pull_bound .push_bound .perf_enabled   bypass() undo_bypass()
//     embedding_lookup(keys u64[N], dest f32[N,w])            host numpy arrays (staged over PCIe)
//     embedding_update(keys, grads)  embedding_update_with_push_keys(keys, push_keys, grads)
//     embedding_lookup_raw(keys_addr, dest_addr, n)  embedding_update_raw(...)            float32 keys,
//     embedding_push_pull_raw(pull, dest, np, push, grads, ns)                            raw addresses
//     embedding_update_with_push_keys_raw / _np_raw          (device or host; detected per pointer)
//     count(k) size() keys() lookup(k) __repr__
//   every batch method returns a `_waittype` whose wait() releases the GIL and joins the stream.
// The server side is bound with bind_store(table_addr, versions_addr, rows, row_start) or looked up by
// node_id in the table registry (register_table), the role InitTensor plays in the reference.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "../../include/herald_amd.h"

namespace py = pybind11;

namespace {

void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void ha_check(int rc, const char *what) {
    if (rc != 0)
        throw std::runtime_error(std::string(what) + ": " + ha_last_error());
}
bool is_device_ptr(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

struct StoreBinding {
    uint64_t table, versions;
    int64_t rows, row_start;
};
std::map<int, StoreBinding> g_tables;

// wait handle: joins an event on the cache's stream, then runs the deferred device->host copies
struct Wait {
    hipEvent_t ev = nullptr;
    std::vector<std::function<void()>> after;
    std::vector<py::object> keep;
    void wait() {
        {
            py::gil_scoped_release release;
            if (ev) {
                hip_check(hipEventSynchronize(ev), "hipEventSynchronize");
                (void)hipEventDestroy(ev);
                ev = nullptr;
            }
        }
        for (auto &f : after)
            f();
        after.clear();
        keep.clear();
    }
    ~Wait() {
        if (ev)
            (void)hipEventDestroy(ev);
    }
};

struct Embedding {
    uint64_t key;
    int64_t version;
    py::array_t<float> data, grad;
    double mean() const {
        double s = 0;
        auto r = data.unchecked<1>();
        for (py::ssize_t i = 0; i < r.shape(0); ++i)
            s += r(i);
        return s / (r.shape(0) ? r.shape(0) : 1);
    }
    double var() const {
        const double m = mean();
        double s = 0;
        auto r = data.unchecked<1>();
        for (py::ssize_t i = 0; i < r.shape(0); ++i)
            s += (r(i) - m) * (r(i) - m);
        return s / (r.shape(0) ? r.shape(0) : 1);
    }
    std::string repr() const {
        std::stringstream ss;
        ss << "<hetu.Embedding : key:" << key << ", len:" << data.size() << ", version:" << version
           << ", mean:" << mean() << ", var:" << var() << ">";
        return ss.str();
    }
};

class Cache {
public:
    Cache(int policy, size_t limit, size_t len, size_t width, int node_id)
        : limit_(limit), len_(len), width_(width), node_id_(node_id) {
        max_batch_ = 1 << 17;
        h_ = ha_cache_create(policy, (int64_t)limit, (int64_t)len, (int64_t)width, max_batch_);
        if (!h_)
            throw std::runtime_error(std::string("ha_cache_create: ") + ha_last_error());
        hip_check(hipStreamCreate(&stream_), "hipStreamCreate");
        auto it = g_tables.find(node_id);
        if (it != g_tables.end())
            bind_store(it->second.table, it->second.versions, it->second.rows, it->second.row_start);
    }
    ~Cache() {
        if (stream_)
            (void)hipStreamSynchronize(stream_);
        for (void *p : stage_)
            (void)hipFree(p);
        if (h_)
            ha_cache_destroy(h_);
        if (stream_)
            (void)hipStreamDestroy(stream_);
    }
    void bind_store(uint64_t table, uint64_t versions, int64_t rows, int64_t row_start) {
        ha_check(ha_cache_bind_store(h_, (float *)table, (int64_t *)versions, rows, row_start), "ha_cache_bind_store");
    }
    size_t limit() const { return limit_; }
    size_t width() const { return width_; }
    int64_t pull_bound() const { return pull_; }
    int64_t push_bound() const { return push_; }
    void set_pull_bound(int64_t b) { pull_ = b; ha_check(ha_cache_set_bounds(h_, pull_, push_), "set_bounds"); }
    void set_push_bound(int64_t b) { push_ = b; ha_check(ha_cache_set_bounds(h_, pull_, push_), "set_bounds"); }
    bool perf_enabled() const { return perf_enabled_; }
    void set_perf_enabled(bool v) {
        perf_enabled_ = v;
        ha_check(ha_cache_set_timing(h_, v ? 1 : 0), "set_timing");      // the perf dict's stage times (cache.cc:99-105)
    }
    py::list perf() const { return perf_; }
    void bypass() { ha_check(ha_cache_set_bypass(h_, 1), "bypass"); }
    void undo_bypass() { ha_check(ha_cache_set_bypass(h_, 0), "undo_bypass"); }

    // ---- numpy entry points (host arrays, uint64 keys) -------------------------------------------------
    std::shared_ptr<Wait> lookup_np(py::array_t<uint64_t> keys, py::array_t<float> dest) {
        check_c(keys, "_keys"); check_c(dest, "_dest");
        const size_t n = keys.size();
        if ((size_t)dest.size() != n * width_)
            throw std::runtime_error("dest has the wrong size");
        return lookup_any(keys.data(), 1, n, dest.mutable_data(), {keys, dest});
    }
    std::shared_ptr<Wait> update_np(py::array_t<uint64_t> keys, py::array_t<float> grads) {
        check_c(keys, "_keys"); check_c(grads, "_grads");
        const size_t n = keys.size();
        if ((size_t)grads.size() != n * width_)
            throw std::runtime_error("grads has the wrong size");
        return update_any(keys.data(), 1, n, grads.data(), nullptr, 0, 0, false, {keys, grads});
    }
    std::shared_ptr<Wait> update_pk_np(py::array_t<uint64_t> keys, py::array_t<uint64_t> pk, py::array_t<float> grads) {
        check_c(keys, "_keys"); check_c(pk, "_push_keys"); check_c(grads, "_grads");
        return update_any(keys.data(), 1, keys.size(), grads.data(), pk.data(), 1, pk.size(), true, {keys, pk, grads});
    }
    // ---- raw entry points (addresses of float32 keys; cache.cc:49-58) ----------------------------------
    std::shared_ptr<Wait> lookup_raw(uint64_t keys, uint64_t dest, size_t n) {
        return lookup_any((const void *)keys, 0, n, (float *)dest, {});
    }
    std::shared_ptr<Wait> update_raw(uint64_t keys, uint64_t grads, size_t n) {
        return update_any((const void *)keys, 0, n, (const float *)grads, nullptr, 0, 0, false, {});
    }
    std::shared_ptr<Wait> update_pk_raw(uint64_t keys, uint64_t pk, uint64_t grads, size_t n, size_t npk) {
        return update_any((const void *)keys, 0, n, (const float *)grads, (const void *)pk, 0, npk, true, {});
    }
    std::shared_ptr<Wait> update_pk_np_raw(uint64_t keys, py::array_t<uint64_t> pk, uint64_t grads, size_t n) {
        check_c(pk, "_push_keys");
        return update_any((const void *)keys, 0, n, (const float *)grads, pk.data(), 1, pk.size(), true, {pk});
    }
    std::shared_ptr<Wait> push_pull_raw(uint64_t pullkeys, uint64_t dest, size_t np, uint64_t pushkeys,
                                        uint64_t grads, size_t ns) {
        auto w = std::make_shared<Wait>();
        const void *pk = to_dev((const void *)pullkeys, np * 4, w);
        const void *sk = to_dev((const void *)pushkeys, ns * 4, w);
        const float *g = (const float *)to_dev((const void *)grads, ns * width_ * 4, w);
        float *d = (float *)dest;
        float *dd = is_device_ptr(d) ? d : (float *)stage(np * width_ * 4);
        ha_check(ha_cache_push_pull(h_, pk, 0, np, dd, sk, 0, ns, g, stream_), "ha_cache_push_pull");
        finish(w, dd, d, np * width_ * 4);
        return w;
    }

    // ---- inspection ----------------------------------------------------------------------------------------
    size_t size() {
        int64_t st[8];
        ha_check(ha_cache_state(h_, st, stream_), "ha_cache_state");
        return (size_t)st[0];
    }
    struct Snap {
        std::vector<uint32_t> keys;
        std::vector<int64_t> version;
        std::vector<int32_t> updates, slots;
    };
    Snap snapshot() {
        const int64_t cap = (int64_t)limit_ + 8;
        uint32_t *k; int64_t *v; int32_t *u; uint64_t *st; int32_t *sl; uint64_t *cnt;
        hip_check(hipMalloc((void **)&k, cap * 4), "hipMalloc");
        hip_check(hipMalloc((void **)&v, cap * 8), "hipMalloc");
        hip_check(hipMalloc((void **)&u, cap * 4), "hipMalloc");
        hip_check(hipMalloc((void **)&st, cap * 8), "hipMalloc");
        hip_check(hipMalloc((void **)&sl, cap * 4), "hipMalloc");
        hip_check(hipMalloc((void **)&cnt, 8), "hipMalloc");
        hip_check(hipMemsetAsync(cnt, 0, 8, stream_), "memset");
        ha_check(ha_cache_snapshot(h_, cap, k, v, u, st, sl, cnt, stream_), "ha_cache_snapshot");
        uint64_t n = 0;
        hip_check(hipMemcpyAsync(&n, cnt, 8, hipMemcpyDeviceToHost, stream_), "memcpy");
        hip_check(hipStreamSynchronize(stream_), "sync");
        Snap s;
        s.keys.resize(n); s.version.resize(n); s.updates.resize(n); s.slots.resize(n);
        if (n) {
            hip_check(hipMemcpy(s.keys.data(), k, n * 4, hipMemcpyDeviceToHost), "memcpy");
            hip_check(hipMemcpy(s.version.data(), v, n * 8, hipMemcpyDeviceToHost), "memcpy");
            hip_check(hipMemcpy(s.updates.data(), u, n * 4, hipMemcpyDeviceToHost), "memcpy");
            hip_check(hipMemcpy(s.slots.data(), sl, n * 4, hipMemcpyDeviceToHost), "memcpy");
        }
        for (void *p : {(void *)k, (void *)v, (void *)u, (void *)st, (void *)sl, (void *)cnt})
            (void)hipFree(p);
        return s;
    }
    py::array_t<uint64_t> keys() {
        Snap s = snapshot();
        std::vector<uint64_t> ks(s.keys.begin(), s.keys.end());
        std::sort(ks.begin(), ks.end());
        return py::array_t<uint64_t>(ks.size(), ks.data());
    }
    int count(uint64_t k) {
        Snap s = snapshot();
        return (int)std::count(s.keys.begin(), s.keys.end(), (uint32_t)k);
    }
    py::object lookup(uint64_t k) {
        Snap s = snapshot();
        for (size_t i = 0; i < s.keys.size(); ++i) {
            if (s.keys[i] != (uint32_t)k)
                continue;
            auto e = std::make_shared<Embedding>();
            e->key = k;
            e->version = s.version[i];
            e->data = py::array_t<float>(width_);
            e->grad = py::array_t<float>(width_);
            hip_check(hipMemcpy(e->data.mutable_data(), ha_cache_data(h_) + (size_t)s.slots[i] * width_, width_ * 4,
                                hipMemcpyDeviceToHost), "memcpy");
            hip_check(hipMemcpy(e->grad.mutable_data(), ha_cache_grad(h_) + (size_t)s.slots[i] * width_, width_ * 4,
                                hipMemcpyDeviceToHost), "memcpy");
            return py::cast(e);
        }
        return py::none();
    }
    // insert(EmbeddingPT) of the policies (lru_cache.cc:9-25): the line enters the cache like a lookup miss would
    // bring it in (same touch / evict bookkeeping), then takes the given version and data
    void insert(std::shared_ptr<Embedding> e) {
        if (!e || (size_t)e->data.size() != width_)
            throw std::runtime_error("insert: the embedding must hold `width` floats");
        uint64_t *k;
        float *row;
        hip_check(hipMalloc((void **)&k, 8), "hipMalloc");
        hip_check(hipMalloc((void **)&row, width_ * 4), "hipMalloc");
        hip_check(hipMemcpyAsync(k, &e->key, 8, hipMemcpyHostToDevice, stream_), "memcpy");
        ha_check(ha_cache_lookup(h_, k, 1, 1, row, stream_), "ha_cache_lookup");
        hip_check(hipMemcpyAsync(row, e->data.data(), width_ * 4, hipMemcpyHostToDevice, stream_), "memcpy");
        ha_check(ha_cache_set_line(h_, (int64_t)e->key, e->version, row, stream_), "ha_cache_set_line");
        hip_check(hipStreamSynchronize(stream_), "sync");
        (void)hipFree(k);
        (void)hipFree(row);
    }
    std::string repr() {
        std::stringstream ss;
        ss << "<Cache : " << size() << "/" << limit_ << " , id:" << node_id_ << " , width:" << width_
           << " , bound:" << pull_ << " " << push_ << ">";
        return ss.str();
    }

private:
    template <class A>
    static void check_c(const A &a, const char *name) {
        if (!a.attr("flags").attr("c_contiguous").template cast<bool>())
            throw std::runtime_error(std::string("Array not continuous in C: ") + name);   // binding.h:51-57
    }
    void *stage(size_t bytes) {
        void *p = nullptr;
        hip_check(hipMalloc(&p, bytes + 256), "hipMalloc(stage)");
        stage_.push_back(p);
        return p;
    }
    // device pointer for `p`: itself if it is device memory, otherwise a staged copy on the stream
    const void *to_dev(const void *p, size_t bytes, std::shared_ptr<Wait> &w) {
        if (bytes == 0 || is_device_ptr(p))
            return p;
        void *d = stage(bytes);
        hip_check(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, stream_), "hipMemcpyAsync(H2D)");
        return d;
    }
    void finish(std::shared_ptr<Wait> &w, float *dev_dest, float *user_dest, size_t bytes) {
        if (dev_dest != user_dest && bytes)
            hip_check(hipMemcpyAsync(user_dest, dev_dest, bytes, hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync(D2H)");
        hip_check(hipEventCreateWithFlags(&w->ev, hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventRecord(w->ev, stream_), "hipEventRecord");
        // staging buffers are released when the call has completed
        std::vector<void *> st;
        st.swap(stage_);
        w->after.push_back([st]() { for (void *p : st) (void)hipFree(p); });
        if (perf_enabled_) {
            int64_t out[8];
            ha_check(ha_cache_perf(h_, out, stream_), "ha_cache_perf");
            py::dict d;
            d["type"] = out[0] == 0 ? "Pull" : "Push";
            d["is_full"] = out[6] != 0;
            d["num_all"] = out[1];
            d["num_unique"] = out[2];
            d["num_miss"] = out[3];
            d["num_transfered"] = out[4];
            if (out[0] == 1)
                d["num_evict"] = out[5];
            double ms[6];
            ha_check(ha_cache_stage_times(h_, ms), "ha_cache_stage_times");
            auto t = [&](int i) { return ms[i] > 0.0 ? ms[i] : 0.0; };
            d["time"] = t(0);
            d["sort_time"] = t(1);
            d["lookup_time"] = t(2);
            d["transfer_time"] = t(4);
            if (out[0] == 0) {
                d["prepare_time"] = 0.0;
                d["copy_time"] = t(3) + t(5);
                d["insert_time"] = 0.0;
            } else {
                d["copy_time"] = t(3);
                d["cleanup_time"] = t(5);
            }
            perf_.append(d);
        }
    }
    std::shared_ptr<Wait> lookup_any(const void *keys, int kind, size_t n, float *dest, std::vector<py::object> keep) {
        if ((int64_t)n > max_batch_)
            throw std::runtime_error("batch larger than max_batch");
        auto w = std::make_shared<Wait>();
        w->keep = std::move(keep);
        const void *k = to_dev(keys, n * (kind == 0 ? 4 : 8), w);
        float *dd = is_device_ptr(dest) ? dest : (float *)stage(n * width_ * 4);
        ha_check(ha_cache_lookup(h_, k, kind, (int64_t)n, dd, stream_), "ha_cache_lookup");
        finish(w, dd, dest, n * width_ * 4);
        return w;
    }
    std::shared_ptr<Wait> update_any(const void *keys, int kind, size_t n, const float *grads, const void *pk,
                                     int pkind, size_t npk, bool with_pk, std::vector<py::object> keep) {
        if ((int64_t)n > max_batch_)
            throw std::runtime_error("batch larger than max_batch");
        auto w = std::make_shared<Wait>();
        w->keep = std::move(keep);
        const void *k = to_dev(keys, n * (kind == 0 ? 4 : 8), w);
        const float *g = (const float *)to_dev(grads, n * width_ * 4, w);
        if (with_pk) {
            const void *p = to_dev(pk, npk * (pkind == 0 ? 4 : 8), w);
            ha_check(ha_cache_update_with_push_keys(h_, k, kind, (int64_t)n, p, pkind, (int64_t)npk, g, stream_),
                     "ha_cache_update_with_push_keys");
        } else {
            ha_check(ha_cache_update(h_, k, kind, (int64_t)n, g, stream_), "ha_cache_update");
        }
        finish(w, nullptr, nullptr, 0);
        return w;
    }

    ha_cache *h_ = nullptr;
    hipStream_t stream_ = nullptr;
    size_t limit_, len_, width_;
    int node_id_;
    int64_t max_batch_;
    int64_t pull_ = 5, push_ = 5;
    bool perf_enabled_ = false;
    py::list perf_;
    std::vector<void *> stage_;
};

template <int POLICY>
struct CacheOf : Cache {
    CacheOf(size_t limit, size_t len, size_t width, int node_id) : Cache(POLICY, limit, len, width, node_id) {}
};

}  // namespace

PYBIND11_MODULE(hetu_cache, m) {
    m.doc() = "hetu cache plugin on libherald_amd (MI355X)";
    py::class_<Wait, std::shared_ptr<Wait>>(m, "_waittype").def("wait", &Wait::wait);
    py::class_<Embedding, std::shared_ptr<Embedding>>(m, "Embedding")
        .def(py::init([](uint64_t key, int64_t version, py::array_t<float> data) {
            auto e = std::make_shared<Embedding>();
            e->key = key;
            e->version = version;
            e->data = data;
            e->grad = py::array_t<float>(data.size());
            return e;
        }))
        .def("mean", &Embedding::mean)
        .def("var", &Embedding::var)
        .def("__repr__", &Embedding::repr)
        .def_readonly("data", &Embedding::data)
        .def_readonly("grad", &Embedding::grad)
        .def_readonly("key", &Embedding::key)
        .def_readwrite("version", &Embedding::version);
    py::class_<Cache>(m, "CacheBase")
        .def_property_readonly("limit", &Cache::limit)
        .def_property_readonly("width", &Cache::width)
        .def_property_readonly("perf", &Cache::perf)
        .def_property("pull_bound", &Cache::pull_bound, &Cache::set_pull_bound)
        .def_property("push_bound", &Cache::push_bound, &Cache::set_push_bound)
        .def_property("perf_enabled", &Cache::perf_enabled, &Cache::set_perf_enabled)
        .def("bypass", &Cache::bypass)
        .def("undo_bypass", &Cache::undo_bypass)
        .def("bind_store", &Cache::bind_store)
        .def("embedding_lookup", &Cache::lookup_np)
        .def("embedding_update", &Cache::update_np)
        .def("embedding_lookup_raw", &Cache::lookup_raw)
        .def("embedding_update_raw", &Cache::update_raw)
        .def("embedding_push_pull_raw", &Cache::push_pull_raw)
        .def("embedding_update_with_push_keys", &Cache::update_pk_np)
        .def("embedding_update_with_push_keys_np_raw", &Cache::update_pk_np_raw)
        .def("embedding_update_with_push_keys_raw", &Cache::update_pk_raw)
        .def("count", &Cache::count)
        .def("lookup", &Cache::lookup)
        .def("insert", &Cache::insert)
        .def("size", &Cache::size)
        .def("keys", &Cache::keys)
        .def("__repr__", &Cache::repr);
    py::class_<CacheOf<0>, Cache>(m, "LRUCache").def(py::init<size_t, size_t, size_t, int>());
    py::class_<CacheOf<1>, Cache>(m, "LFUCache").def(py::init<size_t, size_t, size_t, int>());
    py::class_<CacheOf<2>, Cache>(m, "LFUOptCache").def(py::init<size_t, size_t, size_t, int>());
    m.def("register_table", [](int node_id, uint64_t table, uint64_t versions, int64_t rows, int64_t row_start) {
        g_tables[node_id] = StoreBinding{table, versions, rows, row_start};
    }, py::arg("node_id"), py::arg("table"), py::arg("versions"), py::arg("rows"), py::arg("row_start") = 0);
    m.def("debug", []() { py::print("herald_amd hetu_cache:", ha_version()); });
}
