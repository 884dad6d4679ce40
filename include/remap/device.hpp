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
#include <map>
#include <memory>
#include <mutex>
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

private:
  std::mutex m_;
  std::map<std::tuple<std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t, std::uint32_t>,
           rmx_ctx*>
      ctx_;
};

// Device allocation tied to a context.
class device_buffer {
public:
  device_buffer() = default;
  device_buffer(rmx_ctx* c, std::size_t bytes) : ctx_{c}, bytes_{bytes} {
    if (bytes) check(rmx_malloc(c, bytes, &p_), c, "rmx_malloc");
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
    if (p_) rmx_free(ctx_, p_);
  }
  void swap(device_buffer& o) noexcept {
    std::swap(ctx_, o.ctx_);
    std::swap(p_, o.p_);
    std::swap(bytes_, o.bytes_);
  }
  // grow-only
  void reserve(rmx_ctx* c, std::size_t bytes) {
    if (bytes > bytes_ || c != ctx_) *this = device_buffer{c, bytes};
  }
  template <typename T = void>
  T* get() const noexcept { return static_cast<T*>(p_); }
  std::size_t bytes() const noexcept { return bytes_; }
  explicit operator bool() const noexcept { return p_ != nullptr; }

private:
  rmx_ctx* ctx_{nullptr};
  void* p_{nullptr};
  std::size_t bytes_{0};
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

// Page-locked host staging tied to a context (grow-only).
class pinned_buffer {
public:
  pinned_buffer() = default;
  pinned_buffer(pinned_buffer const&) = delete;
  pinned_buffer& operator=(pinned_buffer const&) = delete;
  ~pinned_buffer() {
    if (p_) rmx_free_host(ctx_, p_);
  }
  void reserve(rmx_ctx* c, std::size_t bytes) {
    if (bytes <= bytes_ && c == ctx_) return;
    if (p_) rmx_free_host(ctx_, p_);
    p_ = nullptr;
    bytes_ = 0;
    ctx_ = c;
    check(rmx_malloc_host(c, bytes, &p_), c, "rmx_malloc_host");
    bytes_ = bytes;
  }
  template <typename T = void>
  T* get() const noexcept { return static_cast<T*>(p_); }

private:
  rmx_ctx* ctx_{nullptr};
  void* p_{nullptr};
  std::size_t bytes_{0};
};

}  // namespace remap
