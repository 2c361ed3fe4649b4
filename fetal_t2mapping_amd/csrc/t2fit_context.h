// t2fit_context.h -- the host seam's long-lived state ("t2fit_create/destroy for a context holding
// streams ... and pinned staging").  Included by t2fit_kernels.hip only (one translation unit).
//
// A numpy-in / numpy-out fit of one volume is three transfers around one kernel: pageable host memory -> HBM,
// fit, HBM -> pageable host memory.  Done naively (round 1: hipMalloc, three stream creates, 2 x slabs event
// creates and pageable hipMemcpyAsync on every call) that took 57 ms for a 22 ms fit of 256^3 x 8 TE.  The
// context keeps everything that can be kept -- streams, events, a grow-only device arena, two pinned staging slots
// per direction -- and moves the caller's pageable bytes itself: worker threads copy slab k+1 of the echo stack
// into a pinned slot (and slab k-1 of the maps out of one) while slab k is being fitted, so the GPU only ever
// sees DMA from and to page-locked memory.  The maps do not depend on the slab split (voxels are independent).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace t2fit {

// A few threads that copy memory: a single core moves ~10 GB/s, the 0.8 GB of one 256^3 x 8 TE round trip need
// the memory system's bandwidth to stay inside the fit's 22 ms.
class CopyPool {
 public:
  explicit CopyPool(int n_threads) {
    for (int i = 0; i < n_threads; ++i) workers_.emplace_back([this] { run(); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  int size() const { return (int)workers_.size(); }

  // A batch of row copies (dst, src, bytes; src == nullptr: zero-fill), each row cut into pieces of <= 4 MiB; returns
  // when all are done.
  struct Row { void* dst; const void* src; size_t bytes; };
  static void move(const Row& p) {
    if (p.src) std::memcpy(p.dst, p.src, p.bytes);
    else std::memset(p.dst, 0, p.bytes);
  }
  void copy(const std::vector<Row>& rows) {
    constexpr size_t kPiece = (size_t)4 << 20;
    std::vector<Row> pieces;
    for (const Row& r : rows)
      for (size_t off = 0; off < r.bytes; off += kPiece)
        pieces.push_back(Row{(char*)r.dst + off, r.src ? (const char*)r.src + off : nullptr, std::min(kPiece, r.bytes - off)});
    if (pieces.empty()) return;
    if (pieces.size() == 1 || workers_.empty()) {
      for (const Row& p : pieces) move(p);
      return;
    }
    std::unique_lock<std::mutex> g(m_);
    batch_ = &pieces;
    next_ = 0;
    left_ = pieces.size();
    ++generation_;
    cv_.notify_all();
    done_cv_.wait(g, [this] { return left_ == 0; });
    batch_ = nullptr;
  }

 private:
  void run() {
    size_t seen = 0;
    std::unique_lock<std::mutex> g(m_);
    for (;;) {
      cv_.wait(g, [&] { return stop_ || (batch_ && generation_ != seen && next_ < batch_->size()) ; });
      if (stop_) return;
      while (batch_ && next_ < batch_->size()) {
        const Row p = (*batch_)[next_++];
        g.unlock();
        move(p);
        g.lock();
        if (--left_ == 0) done_cv_.notify_all();
      }
      seen = generation_;
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  const std::vector<Row>* batch_ = nullptr;
  size_t next_ = 0, left_ = 0, generation_ = 0;
  bool stop_ = false;
};

}  // namespace t2fit

struct t2fit_context {
  int device = 0;
  hipStream_t s_in = nullptr, s_fit = nullptr, s_out = nullptr;
  char* dev = nullptr;       // device arena, grow-only
  size_t dev_cap = 0;
  char* pin_in[2] = {nullptr, nullptr};   // pinned staging, one slab of input each
  char* pin_out[2] = {nullptr, nullptr};  // pinned staging, one slab of maps each
  size_t pin_in_cap = 0, pin_out_cap = 0;
  std::vector<hipEvent_t> events;  // recycled across calls
  t2fit::CopyPool* pool = nullptr;
  std::mutex busy;                 // one call at a time per context
};
