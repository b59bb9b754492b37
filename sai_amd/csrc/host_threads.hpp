// Host-side threading helpers shared by the ingest units and the int8 narrowing.
#pragma once

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

// Worker threads that are always joined, also when an exception unwinds the spawning scope (a
// joinable std::thread that is destroyed calls std::terminate).
struct ThreadGroup {
  std::vector<std::thread> th;
  template <typename F>
  void spawn(F&& f) { th.emplace_back(std::forward<F>(f)); }
  void join() {
    for (auto& t : th)
      if (t.joinable()) t.join();
    th.clear();
  }
  ~ThreadGroup() { join(); }
};


// Persistent workers for loops that fan the same job out batch after batch (a thread costs tens of
// microseconds to create; a 160 MB bgzip file is ~20 batches x 2 phases x 15 threads).  run(nt, fn)
// executes fn(0) .. fn(nt-1), fn(0) on the calling thread, and returns when all are done.  fn must
// not throw (the callers' jobs catch inside).
class WorkerPool {
 public:
  explicit WorkerPool(int n) : n_(n < 1 ? 1 : n) {
    for (int t = 1; t < n_; ++t) threads_.emplace_back([this, t] { loop(t); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_)
      if (t.joinable()) t.join();
  }
  int size() const { return n_; }
  template <typename F>
  void run(int nt, F&& fn) {
    if (nt > n_) nt = n_;
    if (nt <= 1) {
      fn(0);
      return;
    }
    std::function<void(int)> job = [&fn](int t) { fn(t); };
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &job;
      active_ = nt;
      pending_ = nt - 1;
      ++gen_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  void loop(int t) {
    uint64_t seen = 0;
    for (;;) {
      std::function<void(int)>* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        if (t < active_) job = job_;
      }
      if (job) {
        (*job)(t);
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::function<void(int)>* job_ = nullptr;
  int active_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

// A pool shared by all calls of a process (narrow.cpp, text_out.cpp).  A fork()ed child inherits the
// object but not its threads -- a run() there would wait for workers that do not exist -- so the pool
// says whether it still belongs to the calling process and the callers fall back to the calling thread.
#include <unistd.h>

class ProcessPool {
 public:
  explicit ProcessPool(int n) : pool_(n), owner_(getpid()) {}
  bool usable() const { return getpid() == owner_; }
  template <typename F>
  void run(int nt, F&& fn) {
    std::lock_guard<std::mutex> lk(m_);
    pool_.run(nt, std::forward<F>(fn));
  }

 private:
  WorkerPool pool_;
  pid_t owner_;
  std::mutex m_;
};

