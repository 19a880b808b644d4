// pybind11 module `laia_cache`: the plugin surface of the reference (laia/src/python_binding.cc:8-23)
// on top of ha_laia_* (libherald_amd.so).  LaiaScheduler().start(...) spawns the background thread of
// LaiaScheduler::launch (laia/src/laia_scheduler.cc:115-169); pop() blocks (GIL released); the stream
// is [plan, dist] per global batch and ends with [0].  TopkScheduler adds the top-k-table scheduler
// (laia/src/topk_scheduler.cc) and its local-shared distribution through ha_shm_ring_*.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <cstring>
#include <chrono>
#include <queue>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../include/herald_amd.h"

namespace py = pybind11;

static const std::vector<int32_t> &topk_table_order(const std::string &dataset) {
    // pre-profiled orders, topk_scheduler.cc:151-165
    static const std::vector<int32_t> criteo{9, 13, 22, 20, 12, 21, 17, 14, 24, 3, 5, 10, 16,
                                             15, 19, 2, 4, 11, 7, 25, 23, 18, 8, 1, 0, 6};
    static const std::vector<int32_t> avazu{1, 2, 4, 5, 15, 7, 6, 16, 12, 0, 17, 8, 14, 10, 9, 11, 13, 3};
    static const std::vector<int32_t> movie{0, 1};
    static const std::vector<int32_t> criteosearch{0, 11, 3, 4, 5, 14, 1, 6, 2, 13, 16, 9, 8, 10, 12, 7, 15};
    if (dataset == "criteo") return criteo;
    if (dataset == "avazu") return avazu;
    if (dataset == "movie") return movie;
    if (dataset == "criteosearch") return criteosearch;
    throw std::runtime_error("dataset not supported");
}

class LaiaScheduler {
public:
    LaiaScheduler() = default;
    ~LaiaScheduler() {
        close_ = true;
        if (thread_.joinable())
            thread_.join();
        if (h_)
            ha_laia_destroy(h_);
        if (my_ring_)
            ha_shm_ring_close(my_ring_);
        for (auto *r : rings_)
            ha_shm_ring_close(r);
    }
    // argument order of the .cc (laia_scheduler.cc:31-33): num_sample, num_table
    void start(py::array_t<uint64_t, py::array::c_style | py::array::forcecast> samples, size_t num_sample,
               size_t num_table, size_t epoch_num, size_t mini_batch_size, size_t batch_num, size_t nrank,
               size_t rank, size_t cache_size, size_t num_threads, size_t top_k_table) {
        (void)num_threads;
        (void)top_k_table;
        start_impl(samples, num_sample, num_table, epoch_num, mini_batch_size, batch_num, nrank, rank, cache_size);
    }
    std::vector<uint64_t> pop() {
        py::gil_scoped_release release;
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty(); });
        auto v = std::move(q_.front());
        q_.pop();
        if (!error_.empty() && v.size() == 1 && v[0] == 0 && q_.empty()) {
            const std::string e = error_;
            lk.unlock();
            py::gil_scoped_acquire acq;
            throw std::runtime_error(e);
        }
        return v;
    }
    // pop() without the per-element conversion: the message as a uint64 array (the plan's keys / the dist's sample indices;
    // the terminator is [0]).  What herald_amd.laia's data loader glue reads.
    py::array_t<uint64_t> pop_arrays() {
        std::vector<uint64_t> v = pop();
        py::array_t<uint64_t> a(static_cast<py::ssize_t>(v.size()));
        if (!v.empty())
            std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(uint64_t));
        return a;
    }
    // {"batches", "us_per_batch" (inside the library call), "thread_wall_us_per_batch" (the scheduler thread's whole loop)}
    py::dict timing() {
        double t[4] = {0, 0, 0, 0};
        if (h_)
            (void)ha_laia_timing(h_, t);
        const double calls = t[0] > 0 ? t[0] : 1.0;
        py::dict d;
        d["batches"] = static_cast<long long>(t[0]);
        d["us_per_batch"] = t[1] / calls;
        d["host_assign_us"] = t[2] / calls;
        d["host_snapshot_us"] = t[3] / calls;
        d["gpu_and_transfer_us"] = (t[1] - t[2] - t[3]) / calls;
        double td[4] = {0, 0, 0, 0};
        if (h_)
            (void)ha_laia_timing_device(h_, td);
        d["issue_us"] = td[1] / calls;       // device-resident mode: enqueueing the next batch / waiting / copying out
        d["wait_us"] = td[2] / calls;
        d["unpack_us"] = td[3] / calls;
        const long long done = done_.load();
        d["thread_wall_us_per_batch"] = done > 0 ? wall_us_.load() / static_cast<double>(done) : 0.0;
        // the same per batch WITHOUT the first kWarmBatches (the first call allocates and clears the scheduler's device state)
        if (warm_set_.load() && done > kWarmBatches && t[0] > warm_t_[0]) {
            const double n = static_cast<double>(done - kWarmBatches), nc = t[0] - warm_t_[0];
            d["steady_from_batch"] = static_cast<long long>(kWarmBatches);
            d["steady_thread_wall_us_per_batch"] = (wall_us_.load() - warm_wall_us_) / n;
            d["steady_us_per_batch"] = (t[1] - warm_t_[1]) / nc;
            d["steady_issue_us"] = (td[1] - warm_td_[1]) / nc;
            d["steady_wait_us"] = (td[2] - warm_td_[2]) / nc;
            d["steady_unpack_us"] = (td[3] - warm_td_[3]) / nc;
        }
        return d;
    }
    // stops the scheduler thread (the destructor does the same)
    void close() {
        close_ = true;
        py::gil_scoped_release release;
        if (thread_.joinable())
            thread_.join();
    }
    size_t length() {
        if (local_shared_)
            return (size_t)ha_shm_ring_pending_words(my_ring_);
        std::lock_guard<std::mutex> lk(mu_);
        return q_.size();
    }

protected:
    void start_impl(py::array_t<uint64_t, py::array::c_style | py::array::forcecast> samples, size_t num_sample,
                    size_t num_table, size_t epoch_num, size_t mini_batch_size, size_t batch_num, size_t nrank,
                    size_t rank, size_t cache_size) {
        if (samples.ndim() != 2)
            throw std::runtime_error("Input should be 2D numpy array");
        if ((size_t)samples.shape(0) != num_sample || (size_t)samples.shape(1) != num_table)
            throw std::runtime_error("samples shape does not match num_sample x num_table");
        uint64_t key_limit = 1;
        const uint64_t *p = samples.data();
        for (size_t i = 0; i < num_sample * num_table; ++i)
            key_limit = std::max<uint64_t>(key_limit, p[i] + 1);
        int dev = 0;
        (void)hipGetDevice(&dev);
        dev_ = dev;
        h_ = ha_laia_create(p, (int64_t)num_sample, (int64_t)num_table, (int64_t)nrank, (int64_t)cache_size,
                            (int64_t)key_limit, (int64_t)(mini_batch_size * nrank));
        if (!h_)
            throw std::runtime_error(std::string("ha_laia_create: ") + ha_last_error());
        epoch_num_ = epoch_num; mini_bs_ = mini_batch_size; batch_num_ = batch_num;
        nrank_ = nrank; rank_ = rank; num_table_ = num_table;
        thread_ = std::thread([this] { launch(); });
    }
    void push(std::vector<uint64_t> v) {
        std::lock_guard<std::mutex> lk(mu_);
        q_.push(std::move(v));
        cv_.notify_one();
    }
    void launch() {
        (void)hipSetDevice(dev_);
        const size_t W = nrank_;
        std::vector<int64_t> dist(W * mini_bs_), off(W + 1);
        const size_t cap = W * mini_bs_ * num_table_ * (W > 1 ? W - 1 : 1) + 16;
        std::vector<uint64_t> plan(cap);
        size_t epoch_id = 0, batch_num = batch_num_;
        const char *ah = getenv("HA_LAIA_AHEAD");       // one batch ahead (ha_laia_hint_next) unless HA_LAIA_AHEAD=0
        const bool ahead_ = !(ah != nullptr && ah[0] == '0');
        const auto t_start = std::chrono::steady_clock::now();
        while (epoch_id < epoch_num_ && !close_) {
            size_t batch_id = 0;
            ++epoch_id;
            if (epoch_id == epoch_num_)
                batch_num += 1;  // one more allocation for the cache prefetch (laia_scheduler.cc:126-128)
            while (batch_id < batch_num && !close_) {
                if (ahead_)     // the batch of the NEXT call: the following one, the first of the next epoch, or none
                    (void)ha_laia_hint_next(h_, batch_id + 1 < batch_num ? (int64_t)(batch_id + 1)
                                                                       : (epoch_id < epoch_num_ ? 0 : -1));
                // (standalone LaiaScheduler: only this rank's plan is queued, laia_scheduler.cc:140-168 -- with the state on the
                // device only its rows come back)
                const int rc = topk_
                    ? ha_laia_next_topk(h_, (int64_t)batch_id, (int64_t)mini_bs_, order_.data(),
                                        (int64_t)order_.size(), (int64_t)num_threads_, dist.data(), plan.data(),
                                        (int64_t)cap, off.data())
                    : ha_laia_next_for_rank(h_, (int64_t)batch_id, (int64_t)mini_bs_, (int64_t)rank_, dist.data(), plan.data(),
                                            (int64_t)cap, off.data());
                if (rc != 0) {
                    error_ = std::string("ha_laia_next: ") + ha_last_error();
                    finish();
                    return;
                }
                // standalone: this rank's stream into the queue; local-shared: every local worker's
                // stream into its ring (topk_scheduler.cc:304-318)
                const size_t nout = local_shared_ ? local_size_ : 1;
                for (size_t i = 0; i < nout; ++i) {
                    const size_t w = rank_ + i;
                    std::vector<uint64_t> p(plan.begin() + off[w], plan.begin() + off[w + 1]);
                    std::vector<uint64_t> d(mini_bs_);
                    for (size_t k = 0; k < mini_bs_; ++k)
                        d[k] = (uint64_t)dist[w * mini_bs_ + k];
                    if (local_shared_) {
                        if (!ring_send(rings_[i], p) || !ring_send(rings_[i], d))
                            return;
                    } else {
                        push(std::move(p));
                        push(std::move(d));
                    }
                }
                ++batch_id;
                const long long done_now = done_.fetch_add(1) + 1;
                wall_us_.store(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count());
                if (done_now == kWarmBatches) {     // what the first batches cost (the device state is allocated in the first call)
                    warm_wall_us_ = wall_us_.load();
                    (void)ha_laia_timing(h_, warm_t_);
                    (void)ha_laia_timing_device(h_, warm_td_);
                    warm_set_ = true;
                }
            }
        }
        finish();
    }
    void finish() {  // notify python to end (laia_scheduler.cc:168, topk_scheduler.cc:347-354)
        if (local_shared_) {
            for (auto *r : rings_)
                ring_send(r, {0});
        } else {
            push({0});
        }
    }
    bool ring_send(ha_shm_ring *r, const std::vector<uint64_t> &v) {
        while (!close_) {  // poll every 10 us like push_to_local_worker (topk_scheduler.cc:204-222)
            const int rc = ha_shm_ring_send(r, v.data(), (int64_t)v.size());
            if (rc == 1)
                return true;
            if (rc < 0) {
                error_ = "message does not fit the shared-memory ring";
                return false;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(10));
        }
        return false;
    }
    ha_laia *h_ = nullptr;
    int dev_ = 0;
    size_t epoch_num_ = 0, mini_bs_ = 0, batch_num_ = 0, nrank_ = 0, rank_ = 0, num_table_ = 0;
    std::thread thread_;
    std::atomic<bool> close_{false};
    std::atomic<long long> done_{0};
    std::atomic<double> wall_us_{0.0};
    static constexpr long long kWarmBatches = 8;
    std::atomic<bool> warm_set_{false};
    double warm_wall_us_ = 0.0, warm_t_[4] = {0, 0, 0, 0}, warm_td_[4] = {0, 0, 0, 0};
    std::mutex mu_;
    std::condition_variable cv_;
    std::queue<std::vector<uint64_t>> q_;
    std::string error_;
    // TopkScheduler state
    bool topk_ = false, local_shared_ = false;
    std::vector<int32_t> order_;
    size_t num_threads_ = 1, local_rank_ = 0, local_size_ = 1;
    std::vector<ha_shm_ring *> rings_;
    ha_shm_ring *my_ring_ = nullptr;
};

class TopkScheduler : public LaiaScheduler {
public:
    void start(py::array_t<uint64_t, py::array::c_style | py::array::forcecast> samples, size_t num_sample,
               size_t num_table, size_t epoch_num, size_t mini_batch_size, size_t batch_num, size_t nrank,
               size_t rank, size_t cache_size, size_t num_threads, const std::string &dataset,
               size_t top_k_table, bool local_shared, size_t local_rank, size_t local_size) {
        const auto &order = topk_table_order(dataset);
        size_t k = top_k_table ? top_k_table : num_table;    // topk_scheduler.cc:131-133
        k = std::min(k, order.size());                        // :167-168
        order_.assign(order.begin(), order.begin() + k);
        topk_ = true;
        num_threads_ = num_threads;
        local_shared_ = local_shared;
        local_rank_ = local_rank;
        local_size_ = local_size;
        if (local_shared_) {
            if (local_rank_ == 0) {  // rank 0 creates every local worker's ring (:68-84)
                for (size_t i = 0; i < local_size_; ++i) {
                    auto *r = ha_shm_ring_open(("laia_cache_" + std::to_string(i)).c_str(), 1, (int64_t)1 << 24);
                    if (!r)
                        throw std::runtime_error(std::string("ha_shm_ring_open: ") + ha_last_error());
                    rings_.push_back(r);
                }
            }
            my_ring_ = ha_shm_ring_open(("laia_cache_" + std::to_string(local_rank_)).c_str(), 0, 0);
            if (!my_ring_)
                throw std::runtime_error(std::string("ha_shm_ring_open: ") + ha_last_error());
            if (local_rank_ != 0)
                return;  // only local rank 0 schedules (:176-180)
        }
        start_impl(samples, num_sample, num_table, epoch_num, mini_batch_size, batch_num, nrank, rank, cache_size);
    }
    std::vector<uint64_t> pop_from_local_worker() {
        if (!local_shared_)
            throw std::runtime_error("pop_from_local_worker needs local_shared");
        py::gil_scoped_release release;
        std::vector<uint64_t> buf(1 << 16);
        for (;;) {  // polls every 10 us (topk_scheduler.cc:236-260)
            int64_t need = 0;
            const int64_t n = ha_shm_ring_recv(my_ring_, buf.data(), (int64_t)buf.size(), &need);
            if (n >= 0) {
                buf.resize((size_t)n);
                return buf;
            }
            if (n == -2)
                buf.resize((size_t)need + 16);
            else
                std::this_thread::sleep_for(std::chrono::microseconds(10));
        }
    }
};

PYBIND11_MODULE(laia_cache, m) {
    m.doc() = "laia scheduler plugin on libherald_amd (MI355X)";
    py::class_<LaiaScheduler>(m, "LaiaScheduler")
        .def(py::init<>())
        .def("start", &LaiaScheduler::start)
        .def("pop", &LaiaScheduler::pop)
        .def("pop_arrays", &LaiaScheduler::pop_arrays)
        .def("timing", &LaiaScheduler::timing)
        .def("close", &LaiaScheduler::close)
        .def("length", &LaiaScheduler::length);
    py::class_<TopkScheduler>(m, "TopkScheduler")
        .def(py::init<>())
        .def("start", &TopkScheduler::start)
        .def("pop", &TopkScheduler::pop)
        .def("pop_arrays", &TopkScheduler::pop_arrays)
        .def("timing", &TopkScheduler::timing)
        .def("close", &TopkScheduler::close)
        .def("pop_from_local_worker", &TopkScheduler::pop_from_local_worker)
        .def("pop_from_local_worker_arrays", [](TopkScheduler &s) {
            std::vector<uint64_t> v = s.pop_from_local_worker();
            py::array_t<uint64_t> a(static_cast<py::ssize_t>(v.size()));
            if (!v.empty())
                std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(uint64_t));
            return a;
        })
        .def("length", &TopkScheduler::length);
}
