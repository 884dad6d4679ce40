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
#include <filesystem>
#include <fstream>
#include <string>
#include <vector>

#include "fragment.hpp"

namespace ful {

namespace details {
template <typename T>
void put(std::ostream& o, T const& v) {
  o.write(reinterpret_cast<char const*>(&v), sizeof(T));
}
template <typename T>
void get(std::istream& i, T& v) {
  i.read(reinterpret_cast<char*>(&v), sizeof(T));
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

[[nodiscard]] inline std::vector<fgm::fragment> read(std::filesystem::path dir) {
  std::vector<std::filesystem::path> files;
  std::copy(std::filesystem::directory_iterator(dir), std::filesystem::directory_iterator(), std::back_inserter(files));
  std::sort(files.begin(), files.end(),
            [](auto& a, auto& b) { return std::stoi(a.filename().string()) < std::stoi(b.filename().string()); });
  std::vector<fgm::fragment> result;
  for (auto& file : files) {
    std::ifstream input{file, std::ios::in | std::ios::binary};
    mrl::dimensions_t dim{};
    details::get(input, dim);
    fgm::fragment::matrix_type temp{dim};
    input.read(reinterpret_cast<char*>(temp.data()), std::streamsize(dim.area() * sizeof(fgm::dot_type)));
    fgm::point_t zero{};
    details::get(input, zero);
    std::size_t count{};
    details::get(input, count);
    std::vector<fgm::frame> frames{};
    for (std::size_t i = 0; i < count; ++i) {
      fgm::frame frame{};
      details::get(input, frame.number_);
      details::get(input, frame.position_);
      std::size_t size_i{};
      details::get(input, size_i);
      frame.data_.image_.resize(size_i);
      input.read(reinterpret_cast<char*>(frame.data_.image_.data()), std::streamsize(size_i));
      std::size_t size_m{};
      details::get(input, size_m);
      frame.data_.median_.resize(size_m);
      input.read(reinterpret_cast<char*>(frame.data_.median_.data()), std::streamsize(size_m));
      frames.push_back(std::move(frame));
    }
    result.emplace_back(std::move(temp), mrl::dimensions_t{1, 1}, zero, std::move(frames));
  }
  return result;
}

}  // namespace ful
