// remap/core.hpp — value types of the reference's L0/L1 layers, written from scratch for the
// drop-in header surface.  Namespaces and member names are the reference's
// (callers such as main.cpp and user feeds/callbacks spell them), the implementation is not.
//   cdt  point / dimensions / limits / region / offset_t / offset_hash   (reference cdt.hpp)
//   mrl  matrix<T, Alloc> and the size-typed aliases                      (reference mrl.hpp)
//   cpl  tagged colour values, the two luminance-rank LUTs                (reference cpl.hpp)
//   sid  image aliases            ifd  frame + feeder concept             (sid.hpp, ifd.hpp)
//   icd  compressed_t + compressor concept                                (icd.hpp)
//   all  bump-allocator family used by feeds (`produce(alloc)`)           (all.hpp)
#pragma once

#include <algorithm>
#include <functional>
#include <array>
#include <compare>
#include <concepts>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <memory>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

namespace cdt {

template <std::integral T>
struct point {
  using value_type = T;
  T x_{};
  T y_{};
  point& operator+=(point const& o) noexcept { return x_ += o.x_, y_ += o.y_, *this; }
  point& operator-=(point const& o) noexcept { return x_ -= o.x_, y_ -= o.y_, *this; }
  template <std::integral U>
  explicit operator point<U>() const noexcept {
    return {static_cast<U>(x_), static_cast<U>(y_)};
  }
  friend point operator+(point a, point const& b) noexcept { return a += b; }
  friend point operator-(point a, point const& b) noexcept { return a -= b; }
  friend point operator-(point const& a) noexcept { return {static_cast<T>(-a.x_), static_cast<T>(-a.y_)}; }
  friend bool operator==(point const&, point const&) noexcept = default;
};

template <std::unsigned_integral T>
struct dimensions {
  using value_type = T;
  T width_{};
  T height_{};
  T area() const noexcept { return width_ * height_; }
};

template <std::integral T>
struct limits {
  using value_type = T;
  T lower_{std::numeric_limits<T>::max()};
  T upper_{std::numeric_limits<T>::min()};
  void update(T v) noexcept {
    if (v > upper_)
      upper_ = v;
    else if (v < lower_)
      lower_ = v;
  }
  T size() const noexcept { return upper_ - lower_; }
};

template <std::integral T>
struct region {
  using value_type = T;
  T left_{}, top_{}, right_{}, bottom_{};
  point<T> left_top() const noexcept { return {left_, top_}; }
  point<T> right_bottom() const noexcept { return {right_, bottom_}; }
  T width() const noexcept { return right_ - left_; }
  T height() const noexcept { return bottom_ - top_; }
  point<T> margins() const noexcept { return {static_cast<T>(left_ + right_), static_cast<T>(top_ + bottom_)}; }
  T area() const noexcept { return width() * height(); }
  bool contains(point<T> const& p) const noexcept {
    return p.x_ >= left_ && p.x_ < right_ && p.y_ >= top_ && p.y_ < bottom_;
  }
};

template <std::integral T>
region<T> from_limits(limits<T> h, limits<T> v) noexcept {
  return {h.lower_, v.lower_, h.upper_, v.upper_};
}

// tuple-like access used by structured bindings in callers
template <std::size_t I, std::integral T>
T const& get(point<T> const& p) noexcept {
  static_assert(I < 2);
  if constexpr (I == 0) return p.x_; else return p.y_;
}
template <std::size_t I, std::integral T>
T& get(point<T>& p) noexcept {
  static_assert(I < 2);
  if constexpr (I == 0) return p.x_; else return p.y_;
}
template <std::size_t I, std::unsigned_integral T>
T const& get(dimensions<T> const& d) noexcept {
  static_assert(I < 2);
  if constexpr (I == 0) return d.width_; else return d.height_;
}
template <std::size_t I, std::unsigned_integral T>
T& get(dimensions<T>& d) noexcept {
  static_assert(I < 2);
  if constexpr (I == 0) return d.width_; else return d.height_;
}

using offset_t = point<std::int32_t>;

// FNV-1a-style mix in 64-bit arithmetic; the int32 components sign-extend (cdt.hpp:158-171).
struct offset_hash {
  std::size_t operator()(offset_t const& o) const noexcept {
    std::size_t h = 2166136261U;
    h = (h ^ static_cast<std::size_t>(o.x_)) * 16777619U;
    h = (h ^ static_cast<std::size_t>(o.y_)) * 16777619U;
    return h;
  }
};

template <std::integral T, std::unsigned_integral U>
std::ptrdiff_t to_index(point<T> p, dimensions<U> const& d) noexcept {
  return static_cast<std::ptrdiff_t>(d.width_) * p.y_ + p.x_;
}

}  // namespace cdt

namespace std {
template <std::integral T>
struct tuple_size<cdt::point<T>> : integral_constant<size_t, 2> {};
template <size_t I, std::integral T>
struct tuple_element<I, cdt::point<T>> { using type = T; };
template <std::unsigned_integral T>
struct tuple_size<cdt::dimensions<T>> : integral_constant<size_t, 2> {};
template <size_t I, std::unsigned_integral T>
struct tuple_element<I, cdt::dimensions<T>> { using type = T; };
}  // namespace std

namespace mrl {

using size_type = std::size_t;
using point_t = cdt::point<size_type>;
using dimensions_t = cdt::dimensions<size_type>;
using limits_t = cdt::limits<size_type>;
using region_t = cdt::region<size_type>;

// Row-major dense matrix over std::vector (mrl.hpp:14-157).  A matrix may BORROW its cells (view()): a
// read-only window on memory someone else owns — frc::collector hands every callback the frame's median
// image (frc.hpp:119) straight out of the page-locked buffer the block came back in, instead of copying
// 76 800 bytes per frame for a callback that, in main.cpp, does not look at it.  Any write access (and any
// copy) turns the view into an owning matrix first.
// What fills the memory a group of views borrows, run once, by whichever view is read first (frc::collector:
// the medians of a block come back from the device in ONE copy, and only if a callback looks at one).
// (read through const accessors, possibly from several threads of a callback — the reference matches regions
// under std::execution::par: the first reader fills, the others wait for it)
struct cell_loader {
  std::function<void()> fill;
  std::once_flag once;
  explicit cell_loader(std::function<void()> f) : fill{std::move(f)} {}
  void ensure() { std::call_once(once, fill); }
};

template <typename T, typename Alloc = std::allocator<T>>
class matrix {
public:
  using value_type = T;
  using allocator_type = Alloc;

  matrix() noexcept : matrix(allocator_type{}) {}
  explicit matrix(allocator_type const& a) noexcept : cells_{a} {}
  matrix(dimensions_t d, allocator_type const& a) : dim_{d}, cells_{a} { cells_.resize(d.area()); }
  matrix(dimensions_t d, T const& v, allocator_type const& a) : dim_{d}, cells_{a} { cells_.resize(d.area(), v); }
  matrix(dimensions_t d) : matrix(d, allocator_type{}) {}
  matrix(dimensions_t d, T const& v) : matrix(d, v, allocator_type{}) {}
  // borrowed cells: valid while `cells` is (the collector's callbacks: for the duration of the call)
  // `loader` (optional, same lifetime as `cells`): run before the cells are first read
  static matrix view(dimensions_t d, T const* cells, allocator_type const& a = allocator_type{}, cell_loader* loader = nullptr) {
    matrix m{a};
    m.dim_ = d;
    m.borrowed_ = cells;
    m.loader_ = loader;
    return m;
  }
  matrix(matrix const& o) : dim_{o.dim_}, cells_{o.cells_.get_allocator()} { cells_.assign(o.data(), o.end()); }
  matrix(matrix&& o) noexcept : dim_{o.dim_}, cells_{std::move(o.cells_)}, borrowed_{o.borrowed_}, loader_{o.loader_} {
    o.dim_ = {}, o.borrowed_ = nullptr, o.loader_ = nullptr;
  }
  matrix& operator=(matrix const& o) {
    if (this != &o) {
      dim_ = o.dim_;
      cells_.assign(o.data(), o.end());
      borrowed_ = nullptr, loader_ = nullptr;
    }
    return *this;
  }
  matrix& operator=(matrix&& o) noexcept {
    dim_ = o.dim_, cells_ = std::move(o.cells_), borrowed_ = o.borrowed_, loader_ = o.loader_;
    o.dim_ = {}, o.borrowed_ = nullptr, o.loader_ = nullptr;
    return *this;
  }

  T& operator[](size_type i) { return own()[i]; }
  T const& operator[](size_type i) const { return data()[i]; }
  T* data() { return own(); }
  T const* data() const {
    if (loader_) loader_->ensure();
    return borrowed_ ? borrowed_ : cells_.data();
  }
  T* end() { return own() + size(); }
  T const* end() const { return data() + size(); }
  size_type width() const noexcept { return dim_.width_; }
  size_type height() const noexcept { return dim_.height_; }
  size_type size() const noexcept { return borrowed_ ? dim_.area() : cells_.size(); }
  dimensions_t const& dimensions() const noexcept { return dim_; }
  allocator_type get_allocator() const noexcept { return cells_.get_allocator(); }
  bool is_view() const noexcept { return borrowed_ != nullptr; }

  template <typename Fn>
    requires std::invocable<Fn, T const&>
  auto map(Fn fn) const {
    using R = std::invoke_result_t<Fn, T const&>;
    matrix<R> out{dim_};
    std::transform(data(), end(), out.data(), fn);
    return out;
  }

  // sub-matrix without `r.left_/top_/right_/bottom_` cells on the respective sides
  matrix crop(region_t r) const {
    matrix out{{dim_.width_ - r.left_ - r.right_, dim_.height_ - r.top_ - r.bottom_}, get_allocator()};
    for (size_type y = 0; y < out.height(); ++y)
      std::copy_n(data() + (y + r.top_) * dim_.width_ + r.left_, out.width(), out.data() + y * out.width());
    return out;
  }
  // matrix grown by `r.*_` default cells on the respective sides
  matrix extend(region_t r) const {
    matrix out{{dim_.width_ + r.left_ + r.right_, dim_.height_ + r.top_ + r.bottom_}, get_allocator()};
    for (size_type y = 0; y < dim_.height_; ++y)
      std::copy_n(data() + y * dim_.width_, dim_.width_, out.data() + (y + r.top_) * out.width() + r.left_);
    return out;
  }

private:
  T* own() {
    if (borrowed_) {
      if (loader_) loader_->ensure();
      cells_.assign(borrowed_, borrowed_ + dim_.area());
      borrowed_ = nullptr, loader_ = nullptr;
    }
    return cells_.data();
  }
  dimensions_t dim_{};
  std::vector<T, Alloc> cells_;
  T const* borrowed_{nullptr};
  cell_loader* loader_{nullptr};
};

}  // namespace mrl

namespace cpl {

struct nat_coded_color_t {};
struct nat_ordered_value_t {};
struct mon_bit_value_t {};
struct rgb_blended_color_t {};

template <typename T, typename Tag>
struct color_t {
  using value_type = T;
  T value;
  friend auto operator<=>(color_t const&, color_t const&) = default;
};
template <typename T, typename Tag>
T value(color_t<T, Tag> const& c) { return c.value; }

using nat_cc = color_t<std::uint8_t, nat_coded_color_t>;   // C64 palette index 0..15
using nat_ov = color_t<std::uint8_t, nat_ordered_value_t>; // luminance rank 0..15
using mon_bv = color_t<std::uint8_t, mon_bit_value_t>;
using rgb_bc = color_t<std::uint32_t, rgb_blended_color_t>;

inline constexpr nat_cc operator"" _cc(unsigned long long v) noexcept { return {static_cast<std::uint8_t>(v)}; }
inline constexpr nat_ov operator"" _ov(unsigned long long v) noexcept { return {static_cast<std::uint8_t>(v)}; }
inline constexpr mon_bv operator"" _bv(unsigned long long v) noexcept { return {static_cast<std::uint8_t>(v)}; }

// C64 palette as 0x00BBGGRR-style blend words (cpl.hpp:77-92) and the luminance-rank LUTs
// (cpl.hpp:163-217; values verified against the reference build).
inline constexpr std::uint32_t palette[16] = {0x000000, 0xFFFFFF, 0x68372B, 0x70A4B2, 0x6F3D86, 0x588D43,
                                              0x352879, 0xB8C76F, 0x6F4F25, 0x433900, 0x9A6759, 0x444444,
                                              0x6C6C6C, 0x9AD284, 0x6C5EB5, 0x959595};
inline constexpr std::uint8_t n2o_lut[16] = {0, 15, 2, 12, 6, 9, 3, 13, 5, 1, 7, 4, 8, 14, 10, 11};
inline constexpr std::uint8_t o2n_lut[16] = {0, 9, 2, 6, 11, 8, 4, 10, 12, 5, 14, 15, 3, 7, 13, 1};

inline constexpr nat_ov native_to_ordered(nat_cc c) noexcept { return {n2o_lut[c.value & 15]}; }
inline constexpr nat_cc ordered_to_native(nat_ov o) noexcept { return {o2n_lut[o.value & 15]}; }
inline constexpr rgb_bc native_to_blend(nat_cc c) noexcept { return {palette[c.value & 15]}; }

}  // namespace cpl

namespace sid {
namespace nat {
using dimg_t = mrl::matrix<cpl::nat_cc>;
template <typename Alloc>
using aimg_t = mrl::matrix<cpl::nat_cc, Alloc>;
}  // namespace nat
namespace mon {
using dimg_t = mrl::matrix<cpl::mon_bv>;
template <typename Alloc>
using aimg_t = mrl::matrix<cpl::mon_bv, Alloc>;
}  // namespace mon
}  // namespace sid

namespace ifd {
template <typename Image>
struct frame {
  std::size_t number_;
  Image image_;
};
// what frc::collector::collect needs from a feed (ifd.hpp:20-28)
template <typename F, typename Alloc>
concept feeder = requires(F f) {
  { f.has_more() } noexcept -> std::same_as<bool>;
  { f.produce(std::declval<Alloc&&>()) } -> std::same_as<frame<sid::nat::aimg_t<Alloc>>>;
};
}  // namespace ifd

namespace icd {
using compressed_t = std::vector<std::uint8_t>;
template <typename C, typename Alloc>
concept compressor = requires(C c) {
  { c(std::declval<sid::nat::aimg_t<Alloc>>()) } -> std::same_as<compressed_t>;
};
// what fdf::filter needs from the codec (icd.hpp:21-27)
template <typename C, typename Alloc>
concept decompressor = requires(C c) {
  { c(std::declval<compressed_t>(), std::declval<mrl::dimensions_t>()) } -> std::same_as<sid::nat::aimg_t<Alloc>>;
};
}  // namespace icd

namespace all {

template <typename Alloc, typename T>
using rebind_alloc_t = typename std::allocator_traits<Alloc>::template rebind_alloc<T>;

// Bump arena: blocks are chained and released together; deallocate is a no-op
// (all.hpp:14-149).  Host-side only — device buffers are owned by the collector.
class memory_pool {
public:
  memory_pool() noexcept = default;
  explicit memory_pool(std::size_t reserve) {
    if (reserve) grow(reserve);
  }
  memory_pool(memory_pool&& o) noexcept { swap(o); }
  memory_pool& operator=(memory_pool&& o) noexcept {
    memory_pool t{std::move(o)};
    swap(t);
    return *this;
  }
  memory_pool(memory_pool const&) = delete;
  memory_pool& operator=(memory_pool const&) = delete;
  ~memory_pool() {
    for (auto& b : blocks_) std::free(b.first);
  }
  char* get(std::size_t size, std::size_t align) {
    auto fits = [&] {
      std::size_t at = (used_ + align - 1) / align * align;
      return !blocks_.empty() && at + size <= blocks_.back().second ? at : std::size_t(-1);
    };
    std::size_t at = fits();
    if (at == std::size_t(-1)) {
      grow(std::max(size + align, total_ / 2));
      at = fits();
    }
    used_ = at + size;
    total_used_ += size;
    return blocks_.back().first + at;
  }
  std::size_t total_used() const noexcept { return total_used_; }
  // forget what was handed out, keep the memory: one block as large as everything that was reserved
  // (pages already touched stay mapped: a pool reused block after block takes no page faults)
  void reset() {
    if (blocks_.size() > 1) {
      const std::size_t all = total_;
      for (auto& b : blocks_) std::free(b.first);
      blocks_.clear();
      total_ = 0;
      grow(all);
    }
    used_ = 0;
    total_used_ = 0;
  }
  void swap(memory_pool& o) noexcept {
    blocks_.swap(o.blocks_);
    std::swap(used_, o.used_);
    std::swap(total_, o.total_);
    std::swap(total_used_, o.total_used_);
  }

private:
  void grow(std::size_t n) {
    blocks_.emplace_back(static_cast<char*>(std::aligned_alloc(64, (n + 63) / 64 * 64)), n);
    used_ = 0;
    total_ += n;
  }
  std::vector<std::pair<char*, std::size_t>> blocks_;
  std::size_t used_{0}, total_{0}, total_used_{0};
};

template <typename T>
class frame_allocator {
public:
  using value_type = T;
  using propagate_on_container_move_assignment = std::true_type;
  using propagate_on_container_swap = std::true_type;
  explicit frame_allocator(memory_pool& p) noexcept : pool_{&p} {}
  template <typename U>
  frame_allocator(frame_allocator<U> const& o) noexcept : pool_{o.pool_} {}
  T* allocate(std::size_t n) { return reinterpret_cast<T*>(pool_->get(n * sizeof(T), alignof(T))); }
  void deallocate(T*, std::size_t) const noexcept {}
  bool operator==(frame_allocator const& o) const noexcept { return pool_ == o.pool_; }
  bool operator!=(frame_allocator const& o) const noexcept { return pool_ != o.pool_; }

private:
  memory_pool* pool_;
  template <typename U>
  friend class frame_allocator;
};

// two arenas used alternately: the previous frame's data stays valid while the current
// one is produced (all.hpp:151-205)
template <typename T>
class memory_stack {
public:
  using allocator_type = frame_allocator<T>;
  void prepare() { *current_ = memory_pool{previous_->total_used() << 1}; }
  void rotate() noexcept { std::swap(previous_, current_); }
  allocator_type current() const noexcept { return allocator_type{*current_}; }
  allocator_type previous() const noexcept { return allocator_type{*previous_}; }

private:
  memory_pool pools_[2];
  memory_pool* previous_{pools_};
  memory_pool* current_{pools_ + 1};
};

template <typename T>
class memory_swing {
public:
  using stack_type = memory_stack<T>;
  using allocator_type = typename stack_type::allocator_type;
  explicit memory_swing(stack_type& s) : stack_{&s} { stack_->prepare(); }
  ~memory_swing() { stack_->rotate(); }
  operator allocator_type() const noexcept { return get(); }
  allocator_type get() const noexcept { return stack_->current(); }

private:
  stack_type* stack_;
};

}  // namespace all
