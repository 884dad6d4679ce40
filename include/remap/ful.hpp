// remap/ful.hpp — namespace ful of the reference (ful.hpp:10-102): fragment files.
//
//   ful::write(dir, first, last)    one file per fragment, named by its index
//   ful::read(dir)                  files sorted by their integer names -> std::vector<fgm::fragment>
//
// Byte layout (LP64 little endian, exactly what the reference streams out): u64 width, u64
// height; width*height dots of 16 x u16; i32 zero x, y; u64 frame count; per frame u64 number,
// i32 x, y, u64 size + image blob, u64 size + median blob.  The canvas lives in HBM here, so
// write() downloads it once per fragment and read() uploads it (fragment's matrix
// constructor); restored fragments get step (1, 1) as in ful.hpp:97-98.  This is the
// checkpoint between collect and splice.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <string>
#include <utility>
#include <stdexcept>
#include <vector>

#include "fragment.hpp"

namespace ful {

namespace details {
template <typename T>
void put(std::ostream& o, T const& v) {
  o.write(reinterpret_cast<char const*>(&v), sizeof(T));
}
}  // namespace details

template <typename Iter>
void write(std::filesystem::path dir, Iter first, Iter last) {
  for (int i = 0; first != last; ++first, ++i) {
    std::fstream output{dir / std::to_string(i), std::ios::out | std::ios::binary};
    auto const dots = first->dots();  // one download
    details::put(output, dots.dimensions());
    output.write(reinterpret_cast<char const*>(dots.data()),
                 std::streamsize(dots.dimensions().area() * sizeof(fgm::dot_type)));
    details::put(output, first->zero());
    details::put(output, std::size_t(first->frames().size()));
    for (auto const& frame : first->frames()) {
      details::put(output, frame.number_);
      details::put(output, frame.position_);
      details::put(output, std::size_t(frame.data_.image_.size()));
      output.write(reinterpret_cast<char const*>(frame.data_.image_.data()), std::streamsize(frame.data_.image_.size()));
      details::put(output, std::size_t(frame.data_.median_.size()));
      output.write(reinterpret_cast<char const*>(frame.data_.median_.data()),
                   std::streamsize(frame.data_.median_.size()));
    }
  }
}

namespace details {
// A fragment file held in memory, consumed front to back.  Every take() is bounds-checked: a truncated
// or foreign file ends the parse (ok() turns false) instead of reading past the buffer — the reference
// streams field by field and trusts the file.
class cursor {
public:
  explicit cursor(std::filesystem::path const& file) {
    std::ifstream in{file, std::ios::binary | std::ios::ate};
    if (!in) return;
    bytes_.resize(static_cast<std::size_t>(in.tellg()));
    in.seekg(0);
    in.read(bytes_.data(), std::streamsize(bytes_.size()));
    good_ = static_cast<bool>(in);
  }
  [[nodiscard]] bool ok() const noexcept { return good_; }
  [[nodiscard]] std::size_t remaining() const noexcept { return good_ ? bytes_.size() - at_ : 0; }
  void fail() noexcept { good_ = false; }
  template <typename T>
  [[nodiscard]] T take() {
    T v{};
    take_into(&v, sizeof(T));
    return v;
  }
  void take_into(void* dst, std::size_t n) {
    if (!good_ || n > bytes_.size() - at_) {
      good_ = false;
      return;
    }
    std::memcpy(dst, bytes_.data() + at_, n);
    at_ += n;
  }
  [[nodiscard]] icd::compressed_t blob() {  // u64 size + bytes
    auto const n = take<std::uint64_t>();
    icd::compressed_t b;
    if (good_ && n <= bytes_.size() - at_) {
      b.resize(static_cast<std::size_t>(n));
      take_into(b.data(), b.size());
    } else {
      good_ = false;
    }
    return b;
  }

private:
  std::vector<char> bytes_;
  std::size_t at_{0};
  bool good_{false};
};

[[nodiscard]] inline bool parse_fragment(cursor& c, std::vector<fgm::fragment>& out) {
  auto const dim = c.take<mrl::dimensions_t>();
  if (!c.ok()) return false;
  // the canvas is what the header says only if the file holds that many dots: checked before anything is
  // allocated (a truncated or foreign file must end the parse, not ask for terabytes), without overflow
  auto const per_row = c.remaining() / sizeof(fgm::dot_type);
  if (dim.width_ != 0 && dim.height_ > per_row / dim.width_) {
    c.fail();
    return false;
  }
  fgm::fragment::matrix_type dots{dim};  // (a fragment without a canvas is a valid file: the reference reads it too)
  c.take_into(dots.data(), dots.size() * sizeof(fgm::dot_type));
  auto const zero = c.take<fgm::point_t>();
  auto const count = c.take<std::uint64_t>();
  std::vector<fgm::frame> frames;
  for (std::uint64_t i = 0; c.ok() && i < count; ++i) {
    fgm::frame f{};
    f.number_ = c.take<decltype(f.number_)>();
    f.position_ = c.take<fgm::point_t>();
    f.data_.image_ = c.blob();
    f.data_.median_ = c.blob();
    frames.push_back(std::move(f));
  }
  if (!c.ok()) return false;
  out.emplace_back(std::move(dots), mrl::dimensions_t{1, 1}, zero, std::move(frames));  // step (1, 1): ful.hpp:97-98
  return true;
}
}  // namespace details

// Fragments of `dir` in the order of their integer file names (the index write() gave them); entries
// whose name is not a number are skipped.  A file that does not parse (truncated, not a fragment file) is an
// error that names the file — the reference streams whatever bytes it finds; a shortened list handed back
// silently would lose fragments without a trace.
[[nodiscard]] inline std::vector<fgm::fragment> read(std::filesystem::path dir) {
  std::vector<std::pair<long, std::filesystem::path>> numbered;
  for (auto const& entry : std::filesystem::directory_iterator(dir)) {
    auto const name = entry.path().filename().string();
    char* end = nullptr;
    long const index = std::strtol(name.c_str(), &end, 10);
    if (end != name.c_str() && *end == '\0') numbered.emplace_back(index, entry.path());
  }
  std::sort(numbered.begin(), numbered.end());
  std::vector<fgm::fragment> fragments;
  fragments.reserve(numbered.size());
  for (auto const& [index, file] : numbered) {
    details::cursor c{file};
    if (!details::parse_fragment(c, fragments))
      throw std::runtime_error("ful::read: " + file.string() + ": truncated or not a fragment file");
  }
  return fragments;
}

}  // namespace ful
