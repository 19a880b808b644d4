// pybind11 module `laia_cache`: the plugin surface of the reference (laia/src/python_binding.cc:8-23)
// on top of ha_laia_* (libherald_amd.so).  LaiaScheduler().start(...) spawns the background thread of
// LaiaScheduler::launch (laia/src/laia_scheduler.cc:115-169); pop() blocks (GIL released); the stream
// is [plan, dist] per global batch and ends with [0].
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../include/herald_amd.h"

namespace py = pybind11;

class LaiaScheduler {
public:
    LaiaScheduler() = default;
    ~LaiaScheduler() {
        close_ = true;
        if (thread_.joinable())
            thread_.join();
        if (h_)
            ha_laia_destroy(h_);
    }
    // argument order of the .cc (laia_scheduler.cc:31-33): num_sample, num_table
    void start(py::array_t<uint64_t, py::array::c_style | py::array::forcecast> samples, size_t num_sample,
               size_t num_table, size_t epoch_num, size_t mini_batch_size, size_t batch_num, size_t nrank,
               size_t rank, size_t cache_size, size_t num_threads, size_t top_k_table) {
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
    size_t length() {
        std::lock_guard<std::mutex> lk(mu_);
        return q_.size();
    }

private:
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
        while (epoch_id < epoch_num_ && !close_) {
            size_t batch_id = 0;
            ++epoch_id;
            if (epoch_id == epoch_num_)
                batch_num += 1;  // one more allocation for the cache prefetch (laia_scheduler.cc:126-128)
            while (batch_id < batch_num && !close_) {
                if (ha_laia_next(h_, (int64_t)batch_id, (int64_t)mini_bs_, dist.data(), plan.data(), (int64_t)cap,
                                 off.data()) != 0) {
                    error_ = std::string("ha_laia_next: ") + ha_last_error();
                    push({0});
                    return;
                }
                push(std::vector<uint64_t>(plan.begin() + off[rank_], plan.begin() + off[rank_ + 1]));
                std::vector<uint64_t> d(mini_bs_);
                for (size_t i = 0; i < mini_bs_; ++i)
                    d[i] = (uint64_t)dist[rank_ * mini_bs_ + i];
                push(std::move(d));
                ++batch_id;
            }
        }
        push({0});  // notify python to end (laia_scheduler.cc:168)
    }
    ha_laia *h_ = nullptr;
    int dev_ = 0;
    size_t epoch_num_ = 0, mini_bs_ = 0, batch_num_ = 0, nrank_ = 0, rank_ = 0, num_table_ = 0;
    std::thread thread_;
    std::atomic<bool> close_{false};
    std::mutex mu_;
    std::condition_variable cv_;
    std::queue<std::vector<uint64_t>> q_;
    std::string error_;
};

class TopkScheduler {
public:
    void start(py::args, py::kwargs) {
        throw std::runtime_error("TopkScheduler is not built yet (SURVEY.md 8f.1); use LaiaScheduler");
    }
};

PYBIND11_MODULE(laia_cache, m) {
    m.doc() = "laia scheduler plugin on libherald_amd (MI355X)";
    py::class_<LaiaScheduler>(m, "LaiaScheduler")
        .def(py::init<>())
        .def("start", &LaiaScheduler::start)
        .def("pop", &LaiaScheduler::pop)
        .def("length", &LaiaScheduler::length);
    py::class_<TopkScheduler>(m, "TopkScheduler").def(py::init<>()).def("start", &TopkScheduler::start);
}
