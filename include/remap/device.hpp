// remap/device.hpp — the only place where the C++ header layer touches the C ABI
// (include/remap_amd.h): context cache, RAII device buffers, error policy.
//
// The reference's hot path has no error channel; conditions that have no
// reference counterpart (no GPU, a workspace limit, unsupported dimensions) surface as
// remap::device_error — there is no CPU fallback behind these headers.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>

#include "../remap_amd.h"

namespace remap {

struct device_error : std::runtime_error {
  int status;
  device_error(int st, std::string const& what) : std::runtime_error(what), status(st) {}
};

inline void check(int st, rmx_ctx* ctx, char const* what) {
  if (st != RMX_OK)
    throw device_error(st, std::string(what) + " failed with rmx_status " + std::to_string(st) +
                               (ctx ? std::string(": ") + rmx_last_error(ctx) : std::string()));
}

// One rmx_ctx per (width, height, grid, overlap, kmax); created on first use, kept for the
// life of the process (contexts are cheap; their workspaces grow on demand).
class context_cache {
public:
  static rmx_ctx* get(std::uint32_t w, std::uint32_t h, std::uint32_t gw, std::uint32_t gh, std::uint32_t ov,
                      std::uint32_t kmax) {
    static context_cache self;
    std::lock_guard<std::mutex> lock(self.m_);
    auto key = std::make_tuple(w, h, gw, gh, ov, kmax);
    auto it = self.ctx_.find(key);
    if (it != self.ctx_.end()) return it->second;
    rmx_ctx* c = nullptr;
    check(rmx_ctx_create(0, w, h, gw, gh, ov, kmax, nullptr, &c), nullptr, "rmx_ctx_create");
    self.ctx_.emplace(key, c);
    return c;
  }
  // ... with a stream of its own (RMX_STREAM_OWN), one per `lane`: contexts whose work overlaps (the blocks a
  // collector keeps in flight: staging and extraction of the next block beside the callbacks of this one)
  static rmx_ctx* get_lane(std::uint32_t w, std::uint32_t h, std::uint32_t gw, std::uint32_t gh, std::uint32_t ov,
                           std::uint32_t kmax, std::uint32_t lane) {
    static context_cache self;
    std::lock_guard<std::mutex> lock(self.m_);
    auto key = std::make_tuple(w, h, gw, gh, ov, kmax, lane);
    auto it = self.lanes_.find(key);
    if (it != self.lanes_.end()) return it->second;
    rmx_ctx* c = nullptr;
    check(rmx_ctx_create(0, w, h, gw, gh, ov, kmax, RMX_STREAM_OWN, &c), nullptr, "rmx_ctx_create");
    self.lanes_.emplace(key, c);
    return c;
  }

private:
  std::mutex m_;
  std::map<std::tuple<std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t>,
           rmx_ctx*>
      ctx_;
  std::map<std::tuple<std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t>,
           rmx_ctx*>
      lanes_;
};

// Freed device blocks are kept and handed out again.  hipFree waits for the whole device — every stream, also the
// staging and extraction of the blocks a collector has in flight — and a fragment's canvas is reallocated every
// time it grows: in steady state no allocation reaches the driver any more.  Sizes are rounded up to 2^k or
// 1.5 * 2^k, so a canvas that grows by a few per cent per block alternates between two blocks of one class.
// The blocks go back to the driver with trim() or at process exit.
class device_pool {
public:
  static std::size_t size_class(std::size_t bytes) {
    std::size_t c = 256;
    while (c < bytes) {
      if (c + c / 2 >= bytes && c >= 512) return c + c / 2;
      c <<= 1;
    }
    return c;
  }
  static void* acquire(rmx_ctx* c, std::size_t bytes, std::size_t& cap) {
    cap = size_class(bytes);
    {
      auto& self = instance();
      std::lock_guard<std::mutex> lock(self.m_);
      auto it = self.free_.find(cap);
      if (it != self.free_.end() && !it->second.empty()) {
        void* p = it->second.back();
        it->second.pop_back();
        return p;
      }
    }
    void* p = nullptr;
    check(rmx_malloc(c, cap, &p), c, "rmx_malloc");
    return p;
  }
  // (the caller has waited for the work that used the block: the destructors below wait for the owner's stream)
  static void release(void* p, std::size_t cap) {
    auto& self = instance();
    std::lock_guard<std::mutex> lock(self.m_);
    self.free_[cap].push_back(p);
  }
  static void trim(rmx_ctx* c) {
    auto& self = instance();
    std::lock_guard<std::mutex> lock(self.m_);
    for (auto& [cap, v] : self.free_)
      for (void* p : v) rmx_free(c, p);
    self.free_.clear();
  }

private:
  static device_pool& instance() {
    static device_pool self;
    return self;
  }
  std::mutex m_;
  std::map<std::size_t, std::vector<void*>> free_;
};

// Device allocation tied to a context.
class device_buffer {
public:
  device_buffer() = default;
  device_buffer(rmx_ctx* c, std::size_t bytes) : ctx_{c}, bytes_{bytes} {
    if (bytes) p_ = device_pool::acquire(c, bytes, cap_);
  }
  device_buffer(device_buffer&& o) noexcept { swap(o); }
  device_buffer& operator=(device_buffer&& o) noexcept {
    device_buffer t{std::move(o)};
    swap(t);
    return *this;
  }
  device_buffer(device_buffer const&) = delete;
  device_buffer& operator=(device_buffer const&) = delete;
  ~device_buffer() {
    if (p_) {
      (void)rmx_sync(ctx_);  // what the owner's stream still had to do with it
      device_pool::release(p_, cap_);
    }
  }
  void swap(device_buffer& o) noexcept {
    std::swap(ctx_, o.ctx_);
    std::swap(p_, o.p_);
    std::swap(bytes_, o.bytes_);
    std::swap(cap_, o.cap_);
  }
  // grow-only
  void reserve(rmx_ctx* c, std::size_t bytes) {
    if (bytes > cap_ || c != ctx_) *this = device_buffer{c, bytes};
    else bytes_ = std::max(bytes_, bytes);
  }
  template <typename T = void>
  T* get() const noexcept { return static_cast<T*>(p_); }
  std::size_t bytes() const noexcept { return bytes_; }
  explicit operator bool() const noexcept { return p_ != nullptr; }

private:
  rmx_ctx* ctx_{nullptr};
  void* p_{nullptr};
  std::size_t bytes_{0}, cap_{0};
};

namespace detail {
// REMAP_PROFILE=1: wall time per phase of a host-side loop, printed to stderr at destruction.
class phase_clock {
public:
  phase_clock() : on_{std::getenv("REMAP_PROFILE") != nullptr}, last_{std::chrono::steady_clock::now()} {}
  void mark(char const* name) {
    if (!on_) return;
    const auto now = std::chrono::steady_clock::now();
    if (name) acc_[name] += std::chrono::duration<double>(now - last_).count();
    last_ = now;
  }
  ~phase_clock() {
    if (!on_) return;
    for (auto const& [k, v] : acc_) std::fprintf(stderr, "[remap] %-44s %8.1f ms\n", k.c_str(), v * 1e3);
  }

private:
  bool on_;
  std::chrono::steady_clock::time_point last_;
  std::map<std::string, double> acc_;
};
}  // namespace detail

// Page-locked host staging tied to a context (grow-only).  Freed blocks are kept for the next buffer of their size
// class (device_pool's classes): page-locking 300 MB costs ~70 ms, and a collector stages three blocks at a time.
class pinned_buffer {
public:
  pinned_buffer() = default;
  pinned_buffer(pinned_buffer const&) = delete;
  pinned_buffer& operator=(pinned_buffer const&) = delete;
  ~pinned_buffer() { drop(); }
  void reserve(rmx_ctx* c, std::size_t bytes) {
    if (bytes <= cap_) return;
    drop();
    ctx_ = c;
    cap_ = device_pool::size_class(bytes);
    {
      std::lock_guard<std::mutex> lock(mutex());
      auto it = cache().find(cap_);
      if (it != cache().end() && !it->second.empty()) {
        p_ = it->second.back();
        it->second.pop_back();
        return;
      }
    }
    check(rmx_malloc_host(c, cap_, &p_), c, "rmx_malloc_host");
  }
  template <typename T = void>
  T* get() const noexcept { return static_cast<T*>(p_); }

private:
  void drop() {
    if (!p_) return;
    std::lock_guard<std::mutex> lock(mutex());
    cache()[cap_].push_back(p_);
    p_ = nullptr;
    cap_ = 0;
  }
  static std::mutex& mutex() {
    static std::mutex m;
    return m;
  }
  static std::map<std::size_t, std::vector<void*>>& cache() {
    static std::map<std::size_t, std::vector<void*>> c;
    return c;
  }
  rmx_ctx* ctx_{nullptr};
  void* p_{nullptr};
  std::size_t cap_{0};
};

}  // namespace remap
