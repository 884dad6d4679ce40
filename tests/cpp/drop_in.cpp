// drop_in.cpp — a caller written against the REFERENCE's header surface (frc.hpp, fgm.hpp,
// kpe.hpp, kpm.hpp: same includes, type names and concepts as main.cpp uses), compiled
// against the device-backed drop-in headers.  Prints what tests/test_dropin_gpu.py checks.
//
//   g++ -std=c++20 -O1 -Iinclude -Iinclude/remap/compat tests/cpp/drop_in.cpp -Lremap_amd -lremap_amd ...
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <cstdlib>
#include <string>

#include "frc.hpp"
#include "fdf.hpp"
#include "ful.hpp"
#include "nic.hpp"
#include "mpb.hpp"

// same shape as main.cpp's file_feed (main.cpp:16-52), reading from one memory blob
class mem_feed {
public:
  mem_feed(mrl::dimensions_t d, std::vector<std::uint8_t> const& blob) : dim_{d}, blob_{&blob} {}
  bool has_more() const noexcept { return next_ * dim_.area() < blob_->size(); }
  template <typename Alloc>
  auto produce(Alloc const& alloc) {
    using image_type = sid::nat::aimg_t<Alloc>;
    image_type img{dim_, alloc};
    std::memcpy(img.data(), blob_->data() + next_ * dim_.area(), dim_.area());
    return ifd::frame<image_type>{next_++, std::move(img)};
  }

private:
  mrl::dimensions_t dim_;
  std::vector<std::uint8_t> const* blob_;
  std::size_t next_{0};
};

// a feed whose frames are already in host memory: the images are views on the blob (no copy in the feeder; the
// collector's gather into its page-locked staging buffer is the one copy before PCIe)
class mem_view_feed {
public:
  mem_view_feed(mrl::dimensions_t d, std::vector<std::uint8_t> const& blob) : dim_{d}, blob_{&blob} {}
  bool has_more() const noexcept { return next_ * dim_.area() < blob_->size(); }
  template <typename Alloc>
  auto produce(Alloc const& alloc) {
    using image_type = sid::nat::aimg_t<Alloc>;
    auto const* px = reinterpret_cast<typename image_type::value_type const*>(blob_->data() + next_ * dim_.area());
    return ifd::frame<image_type>{next_++, image_type::view(dim_, px, alloc)};
  }

private:
  mrl::dimensions_t dim_;
  std::vector<std::uint8_t> const* blob_;
  std::size_t next_{0};
};

struct noop_compression {
  template <typename Alloc>
  icd::compressed_t operator()(sid::nat::aimg_t<Alloc> const& img) const {
    return icd::compressed_t(1, std::uint8_t(img.size() & 0xff));
  }
};

// an adapter with the members mpb::builder uses on this path (main.cpp:194-244)
struct counting_callbacks {
  std::size_t frames{0};
  std::size_t collected{0}, spliced{0};
  void operator()(fgm::fragment const&, frc::frame_type const&, frc::image_type const&, frc::grid_type const&) noexcept {
    ++frames;
  }
  void operator()(std::string const& tag, std::list<fgm::fragment> const& f) noexcept {
    if (tag == "frc") collected = f.size();
  }
  void operator()(std::string const& tag, std::vector<fgm::fragment> const& f) noexcept {
    if (tag == "spl") spliced = f.size();
  }
};
class mem_adapter {
public:
  using callbacks_type = counting_callbacks;
  using feed_type = mem_feed;
  noop_compression get_compression() const { return {}; }
  callbacks_type& get_callbacks() noexcept { return callbacks_; }

private:
  callbacks_type callbacks_{};
};

static unsigned long long canvas_sum(fgm::fragment const& f) {
  auto dots = f.dots();
  unsigned long long s = 0;
  for (auto const* d = dots.data(); d < dots.end(); ++d)
    for (int b = 0; b < 16; ++b) s = s * 1000003ull + (*d)[b];
  return s;
}

// mpb::builder::collect + splice (mpb.hpp:52-68) over a memory feed
static int run_splice(mrl::dimensions_t dim, std::vector<std::uint8_t> const& blob) {
  mpb::builder<mem_adapter> builder{mem_adapter{}};
  mem_feed feed{dim, blob};
  auto fragments = builder.collect(feed, dim);
  std::printf("collected %zu\n", fragments.size());
  auto spliced = builder.splice(fragments);
  std::printf("spliced %zu\n", spliced.size());
  std::size_t fi = 0;
  for (auto& f : spliced) {
    std::printf("fragment %zu dims %zu %zu zero %d %d frames %zu sum %llu\n", fi, f.dimensions().width_,
                f.dimensions().height_, f.zero().x_, f.zero().y_, f.frames().size(), canvas_sum(f));
    for (auto& fr : f.frames())
      std::printf("frame %zu frag %zu pos %d %d\n", fr.number_, fi, fr.position_.x_, fr.position_.y_);
    ++fi;
  }
  return 0;
}

// ful::read + ful::write (ful.hpp:10-102): every fragment file of `in` through HBM and back
static int run_ful(std::string const& in, std::string const& out) {
  auto fragments = ful::read(in);
  std::printf("read %zu\n", fragments.size());
  for (auto& f : fragments)
    std::printf("fragment dims %zu %zu zero %d %d frames %zu sum %llu\n", f.dimensions().width_, f.dimensions().height_,
                f.zero().x_, f.zero().y_, f.frames().size(), canvas_sum(f));
  ful::write(out, fragments.begin(), fragments.end());
  return 0;
}

// collect with the reference's codec as Comp (main.cpp:113-125): nic::compression is batched
// on the device by the collector; prints every frame's two blobs as hex for the test to compare
static int run_nic(mrl::dimensions_t dim, std::vector<std::uint8_t> const& blob) {
  frc::collector collector{dim};
  auto cb = [](fgm::fragment const&, frc::frame_type const&, frc::image_type const&, frc::grid_type const&) {};
  collector.collect(mem_feed{dim, blob}, nic::compression{}, cb);
  auto fragments = collector.complete();
  auto hex = [](icd::compressed_t const& v) {
    std::string s;
    char b[3];
    for (auto x : v) {
      std::snprintf(b, 3, "%02x", x);
      s += b;
    }
    return s;
  };
  for (auto& f : fragments)
    for (auto& fr : f.frames()) std::printf("blob %zu %s %s\n", fr.number_, hex(fr.data_.image_).c_str(), hex(fr.data_.median_).c_str());
  // the single-image forms round-trip
  sid::nat::dimg_t a{dim};
  std::memcpy(a.data(), blob.data(), dim.area());
  auto pk = nic::compress(a);
  auto back = nic::compression{}(pk, dim);
  std::printf("single %d %zu\n", int(std::memcmp(back.data(), a.data(), dim.area()) == 0), pk.size());
  return 0;
}

// collect (nic codec) -> fdf::filter (fdf.hpp:40-89): per frame kept contours and mask pixels,
// per filtered fragment dims and canvas checksum
static int run_fdf(mrl::dimensions_t dim, std::vector<std::uint8_t> const& blob) {
  frc::collector collector{dim};
  auto cb0 = [](fgm::fragment const&, frc::frame_type const&, frc::image_type const&, frc::grid_type const&) {};
  collector.collect(mem_feed{dim, blob}, nic::compression{}, cb0);
  auto list = collector.complete();
  std::vector<fgm::fragment> fragments;
  for (auto& f : list) fragments.push_back(std::move(f));
  auto cb = [&](fgm::fragment const&, std::size_t idx, sid::nat::dimg_t const&, std::size_t no, sid::nat::dimg_t const&,
                fgm::point_t const&, fdf::contours_t const& foreground, sid::mon::dimg_t const& mask) {
    std::size_t px = 0;
    unsigned long long h = 0;
    for (auto const* m = mask.data(); m < mask.end(); ++m) {
      px += m->value != 0;
      h = h * 1000003ull + (m->value != 0);
    }
    std::printf("frame %zu frag %zu contours %zu maskpx %zu maskhash %llu\n", no, idx, foreground.size(), px, h);
  };
  auto out = fdf::filter(fragments, dim, nic::compression{}, cb);
  std::size_t fi = 0;
  for (auto& f : out)
    std::printf("fragment %zu dims %zu %zu zero %d %d frames %zu sum %llu\n", fi++, f.dimensions().width_,
                f.dimensions().height_, f.zero().x_, f.zero().y_, f.frames().size(), canvas_sum(f));
  return 0;
}

// throughput of frc::collector::collect through the drop-in layer (host feed, callbacks and all)
template <typename Feed>
static int run_time(mrl::dimensions_t dim, std::vector<std::uint8_t> const& blob, bool codec, char const* feed_name) {
  for (int rep = 0; rep < 2; ++rep) {  // the second pass has the context, workspace and pools warm
    frc::collector collector{dim};
    collector.prefetch_feed(std::getenv("REMAP_NO_PREFETCH") == nullptr);  // (the feed and the counting callback share nothing)
    std::size_t calls = 0;
    auto cb = [&](fgm::fragment const&, frc::frame_type const&, frc::image_type const&, frc::grid_type const&) { ++calls; };
    const auto t0 = std::chrono::steady_clock::now();
    if (codec)
      collector.collect(Feed{dim, blob}, nic::compression{}, cb);
    else
      collector.collect(Feed{dim, blob}, noop_compression{}, cb);
    auto frags = collector.complete();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("collect pass %d: %zu frames, %zu fragments, %.3f s, %.0f frames/s (%s, %s)\n", rep, calls + 1, frags.size(), dt,
                double(calls + 1) / dt, codec ? "nic::compression" : "no-op codec", feed_name);
  }
  return 0;
}

// two collect() calls on ONE collector (frc.hpp:55-68 starts each with process_init, which always
// opens a fragment): the second stream must not be blitted into the first one's last fragment
static int run_twice(mrl::dimensions_t dim, std::vector<std::uint8_t> const& blob) {
  frc::collector collector{dim};
  std::size_t calls = 0;
  auto cb = [&](fgm::fragment const&, frc::frame_type const&, frc::image_type const&, frc::grid_type const&) { ++calls; };
  collector.collect(mem_feed{dim, blob}, noop_compression{}, cb);
  collector.collect(mem_feed{dim, blob}, noop_compression{}, cb);
  auto fragments = collector.complete();
  std::printf("calls %zu nfrag %zu\n", calls, fragments.size());
  std::size_t fi = 0;
  for (auto& f : fragments)
    std::printf("fragment %zu dims %zu %zu zero %d %d frames %zu sum %llu\n", fi++, f.dimensions().width_,
                f.dimensions().height_, f.zero().x_, f.zero().y_, f.frames().size(), canvas_sum(f));
  return 0;
}

// exact_callbacks(true): the callback of frame t sees the fragment exactly after frame t
// (frc.hpp:117-119) — prints the canvas checksum it saw
static int run_exact(mrl::dimensions_t dim, std::vector<std::uint8_t> const& blob) {
  frc::collector collector{dim};
  collector.exact_callbacks(true);
  auto cb = [&](fgm::fragment const& fragment, frc::frame_type const& frame, frc::image_type const&, frc::grid_type const&) {
    std::printf("seen %zu frames %zu dims %zu %zu zero %d %d sum %llu\n", frame.number_, fragment.frames().size(),
                fragment.dimensions().width_, fragment.dimensions().height_, fragment.zero().x_, fragment.zero().y_,
                canvas_sum(fragment));
  };
  collector.collect(mem_feed{dim, blob}, noop_compression{}, cb);
  auto fragments = collector.complete();
  std::printf("nfrag %zu\n", fragments.size());
  return 0;
}

// the constants of frc::collector::match_config (frc.hpp:30-44)
struct seq_cfg {
  using allocator_type = std::allocator<char>;
  static constexpr std::size_t weight_switch{10};
  static constexpr std::size_t region_votes{3};
  allocator_type get_allocator() const noexcept { return {}; }
};

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  if (argc > 5 && std::string(argv[4]) == "ful") return run_ful(argv[1], argv[5]);
  const std::size_t W = std::stoul(argv[2]), H = std::stoul(argv[3]);
  std::ifstream in(argv[1], std::ios::binary);
  std::vector<std::uint8_t> blob((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  mrl::dimensions_t dim{W, H};
  if (argc > 4 && std::string(argv[4]) == "splice") return run_splice(dim, blob);
  if (argc > 4 && std::string(argv[4]) == "nic") return run_nic(dim, blob);
  if (argc > 4 && std::string(argv[4]) == "fdf") return run_fdf(dim, blob);
  if (argc > 4 && std::string(argv[4]) == "time") return run_time<mem_feed>(dim, blob, false, "copying feed");
  if (argc > 4 && std::string(argv[4]) == "timenic") return run_time<mem_feed>(dim, blob, true, "copying feed");
  if (argc > 4 && std::string(argv[4]) == "timeview") return run_time<mem_view_feed>(dim, blob, false, "view feed");
  if (argc > 4 && std::string(argv[4]) == "timeviewnic") return run_time<mem_view_feed>(dim, blob, true, "view feed");
  if (argc > 4 && std::string(argv[4]) == "twice") return run_twice(dim, blob);
  if (argc > 4 && std::string(argv[4]) == "exact") return run_exact(dim, blob);

  frc::collector collector{dim};
  std::size_t calls = 0, entries = 0, last_no = 0;
  bool ordered = true;
  unsigned long long med_sum = 0;
  auto cb = [&](fgm::fragment const& fragment, frc::frame_type const& frame, frc::image_type const& median,
                frc::grid_type const& grid) {
    ++calls;
    ordered = ordered && frame.number_ == last_no + 1 && fragment.frames().back().number_ == frame.number_;
    last_no = frame.number_;
    for (auto& r : grid.regions()) entries += r.total_count();
    for (auto const* p = median.data(); p < median.end(); ++p) med_sum += p->value;
  };
  if (char const* b = std::getenv("REMAP_TEST_BLOCK")) collector.block_size(std::uint32_t(std::atoi(b)));  // (several blocks from a short stream)
  if (std::getenv("REMAP_TEST_PREFETCH")) collector.prefetch_feed(true);
  collector.collect(mem_feed{dim, blob}, noop_compression{}, cb);
  auto fragments = collector.complete();

  std::printf("calls %zu ordered %d entries %zu medsum %llu nfrag %zu\n", calls, int(ordered), entries, med_sum,
              fragments.size());
  std::size_t fi = 0;
  for (auto& f : fragments) {
    auto dots = f.dots();
    unsigned long long s = 0;
    for (auto const* d = dots.data(); d < dots.end(); ++d)
      for (int b = 0; b < 16; ++b) s = s * 1000003ull + (*d)[b];
    auto const mg = f.margins();  // fgm.hpp:145-153
    std::printf("fragment %zu dims %zu %zu zero %d %d frames %zu sum %llu margins %zu %zu %zu %zu\n", fi, f.dimensions().width_,
                f.dimensions().height_, f.zero().x_, f.zero().y_, f.frames().size(), s, mg.left_, mg.top_, mg.right_, mg.bottom_);
    for (auto& fr : f.frames())
      std::printf("frame %zu frag %zu pos %d %d packed %zu %zu\n", fr.number_, fi, fr.position_.x_, fr.position_.y_,
                  fr.data_.image_.size(), fr.data_.median_.size());
    ++fi;
  }
  // single-image kpe::extractor and pairwise kpm::match through the same surface
  {
    using grid_t = kpr::grid<4, 2, std::allocator<char>>;
    kpe::extractor<grid_t, 16> ex{dim};
    sid::nat::dimg_t a{dim}, b{dim}, ma{dim}, mb{dim};
    std::memcpy(a.data(), blob.data(), dim.area());
    std::memcpy(b.data(), blob.data() + dim.area(), dim.area());
    auto ga = ex.extract(a, ma, std::allocator<char>{});
    auto gb = ex.extract(b, mb, std::allocator<char>{});
    auto off = kpm::match(seq_cfg{}, ga, gb);
    std::size_t ea = 0;
    for (auto& r : ga.regions()) ea += r.total_count();
    std::printf("pair entries %zu matched %d off %d %d\n", ea, int(bool(off)), off ? off->x_ : 0, off ? off->y_ : 0);
  }
  return 0;
}
