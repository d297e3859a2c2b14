// A small pool of copy threads (plain C++, no HIP: tests/test_host.py compiles it with g++): one memcpy thread moves ~10 GB/s, a PCIe 5 x16
// link 55 GB/s, so the bounce buffers of nmp_stage.hpp are filled / drained by several threads at once.
#pragma once
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace nmp_host {

class CopyPool {
 public:
  void copy(void* dst, const void* src, size_t bytes) {
    const int n = workers();
    if (bytes < (4u << 20) || n == 0) { memcpy(dst, src, bytes); return; }
    const int parts = n + 1;
    size_t slice = (bytes / parts + 4095) & ~(size_t)4095;
    {
      std::lock_guard<std::mutex> lk(m_);
      dst_ = (char*)dst; src_ = (const char*)src; bytes_ = bytes; slice_ = slice;
      pending_ = n;
      generation_++;
    }
    cv_work_.notify_all();
    part(0);
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (auto& t : th_) t.join();
    th_.clear();
    started_ = false; stop_ = false;
  }

 private:
  int workers() {
    if (!started_) {
      started_ = true;
      int want = 8;
      if (const char* e = getenv("NMP_COPY_THREADS")) want = atoi(e);
      cpu_set_t set;
      int cpus = 1;
      if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
      if (want > cpus) want = cpus;
      if (want < 1) want = 1;
      const unsigned g0 = generation_;                 // (only the caller of copy() changes it: jobs posted before a restart are not the new workers')
      for (int i = 1; i < want; i++) th_.emplace_back([this, i, g0] { loop(i, g0); });
    }
    return (int)th_.size();
  }
  void part(int i) {
    const size_t lo = (size_t)i * slice_;
    if (lo >= bytes_) return;
    const size_t n = bytes_ - lo < slice_ ? bytes_ - lo : slice_;
    memcpy(dst_ + lo, src_ + lo, n);
  }
  void loop(int i, unsigned seen) {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      part(i);
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) cv_done_.notify_one();
      }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_work_, cv_done_;
  char* dst_ = nullptr; const char* src_ = nullptr;
  size_t bytes_ = 0, slice_ = 0;
  unsigned generation_ = 0;
  int pending_ = 0;
  bool stop_ = false, started_ = false;
};


}  // namespace nmp_host
