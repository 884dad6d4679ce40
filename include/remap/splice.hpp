// remap/splice.hpp — namespaces fgs and mpb of the reference on top of the C ABI.
//
//   fgs::splice(first, last)                     (fgs.hpp:187-213)
//   mpb::builder<Adapter>::collect / splice / filter  (mpb.hpp:52-77; aws before and arf
//                                                  after them are outside this path)
//
// Pixels and keypoints stay on the device: a snippet is a fragment (canvas in HBM) plus its
// blend mask and single-region descriptor table, both device buffers produced by
// rmx_fgm_blend and rmx_kpe_batch (grid 1x1, overlap 0: fgs.hpp:17,85); candidate pairs are
// scored by rmx_kpm_cell_pairs (cellular kpm::match, kpm.hpp:371-393).  The greedy merge
// loop is sequential by nature and stays on the host, with the reference's list/edge ORDER
// (select_match takes the first maximum, fgs.hpp:142-163).
#pragma once

#include <iterator>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "collector.hpp"
#include "fdf.hpp"

namespace fgs {

namespace details {

// rows of a snippet's descriptor table: extract_single starts with snippet_kmax and, when a
// canvas has more keypoints (rmx_kpe_status reports how many), extracts again with the next
// power of two (the reference has no limit, kpr.hpp:121-124), up to snippet_kmax_limit = the
// largest table a context takes; beyond it splice throws remap::device_error(RMX_ERR_KMAX).
// rmx_kpm_cell_pairs matches pairs of any size (its context only sets the size of the tables
// that are matched in LDS) — INTEGRATION.md §4.
inline constexpr std::uint32_t snippet_kmax = 2048;
inline constexpr std::uint32_t snippet_kmax_limit = 1u << 22;
inline constexpr std::uint32_t cell_ctx_kmax = 4096;
inline constexpr kpm::cell_size_t cell_size{15, 15};  // fgs.hpp:121

struct snippet;
using snippet_list = std::list<snippet>;
using snippet_iterator_t = snippet_list::iterator;

struct delta {
  bool primary_;
  kpm::vote vote_;
  snippet_iterator_t other_;
};

// a snippet's descriptor table: rows inside a block that may be shared with the other snippets of one
// batched extraction (extract_all); the block lives as long as any of its tables
struct table_rows {
  std::shared_ptr<remap::device_buffer> block_;
  std::size_t first_row_{0};
  template <typename T>
  T* get() const noexcept {
    return block_ ? reinterpret_cast<T*>(block_->template get<rmx_desc>() + first_row_) : nullptr;
  }
};

struct snippet {
  fgm::fragment fragment_;
  remap::device_buffer mask_;  // dim.area() bytes, non-zero where the canvas has content
  table_rows table_;           // count_ rmx_desc rows (stride: the kmax it was extracted with)
  std::uint32_t count_{0};
  std::list<delta> edges_;
};

// fgs.hpp:80-103 — blend the canvases, extract the keypoints of the blended images (one region each).
// The reference runs extract_single over the fragments under std::execution::par; here all fragments of
// one canvas size go through ONE rmx_kpe_batch launch and one status / count read-back (a stream with
// hundreds of fragments otherwise pays a device round trip per fragment before the first pair is scored).
inline void extract_group(std::vector<snippet*> const& group, std::uint32_t w, std::uint32_t h) {
  const std::uint32_t n = std::uint32_t(group.size());
  const std::size_t area = std::size_t(w) * h;
  rmx_ctx* c = remap::context_cache::get(w, h, 1, 1, 0, snippet_kmax);
  remap::device_buffer images{c, n * area}, medians{c, n * area}, kcount{c, std::size_t(n) * 4 + 16};
  for (std::uint32_t i = 0; i < n; ++i) {
    snippet& s = *group[i];
    s.mask_ = remap::device_buffer{c, area};
    if (s.fragment_.device_canvas()) {
      remap::check(rmx_fgm_blend(c, s.fragment_.device_canvas(), w, h, images.get<std::uint8_t>() + i * area, s.mask_.get<std::uint8_t>()),
                   c, "rmx_fgm_blend");
    } else {
      remap::check(rmx_memset(c, images.get<std::uint8_t>() + i * area, 0, area), c, "rmx_memset");
      remap::check(rmx_memset(c, s.mask_.get(), 0, area), c, "rmx_memset");
    }
  }
  for (std::uint32_t kmax = snippet_kmax;;) {
    auto block = std::make_shared<remap::device_buffer>(c, std::size_t(n) * kmax * sizeof(rmx_desc));
    remap::check(rmx_kpe_batch(c, images.get<std::uint8_t>(), n, medians.get<std::uint8_t>(), block->get<rmx_desc>(),
                               kcount.get<std::uint32_t>()),
                 c, "rmx_kpe_batch");
    std::uint32_t need = 0;
    const int st = rmx_kpe_status(c, &need);
    if (st == RMX_ERR_KMAX && need <= snippet_kmax_limit) {  // same canvases, larger tables
      while (kmax < need) kmax <<= 1;
      remap::check(rmx_sync(c), c, "rmx_sync");
      c = remap::context_cache::get(w, h, 1, 1, 0, kmax);
      continue;
    }
    if (st == RMX_ERR_KMAX)
      throw remap::device_error(st, "fgs::splice: a fragment of " + std::to_string(w) + "x" + std::to_string(h) + " has " +
                                        std::to_string(need) + " keypoints; a descriptor table holds at most " +
                                        std::to_string(snippet_kmax_limit));
    remap::check(st, c, "rmx_kpe_status");
    std::vector<std::uint32_t> counts(n);
    remap::check(rmx_download(c, counts.data(), kcount.get(), std::size_t(n) * 4), c, "rmx_download");
    for (std::uint32_t i = 0; i < n; ++i) {
      group[i]->table_ = table_rows{block, std::size_t(i) * kmax};
      group[i]->count_ = counts[i];
    }
    break;
  }
}

template <typename It>
void extract_all(It first, It last) {  // snippets whose fragment_ is set
  std::map<std::pair<std::uint32_t, std::uint32_t>, std::vector<snippet*>> by_size;
  for (; first != last; ++first) {
    const auto dim = first->fragment_.dimensions();
    by_size[{std::uint32_t(dim.width_), std::uint32_t(dim.height_)}].push_back(&*first);
  }
  for (auto const& [size, group] : by_size) extract_group(group, size.first, size.second);
}

inline snippet extract_single(fgm::fragment&& fragment) {  // fgs.hpp:80-89
  snippet s{std::move(fragment), {}, {}, 0, {}};
  const auto dim = s.fragment_.dimensions();
  extract_group({&s}, std::uint32_t(dim.width_), std::uint32_t(dim.height_));
  return s;
}

// fgs.hpp:119-133 — head against every snippet of [first, last): one device call for all pairs
template <typename It>
void match_partial(It head, It first, It last) {
  std::vector<It> others;
  for (; first != last; ++first) others.push_back(first);
  if (others.empty()) return;
  std::vector<rmx_desc const*> tables{head->table_.template get<rmx_desc>()};
  std::vector<std::uint32_t> counts{head->count_}, dims, pairs;
  std::vector<std::uint8_t const*> masks{head->mask_.template get<std::uint8_t>()};
  auto push_dims = [&](fgm::fragment const& f) {
    dims.push_back(std::uint32_t(f.dimensions().width_));
    dims.push_back(std::uint32_t(f.dimensions().height_));
  };
  push_dims(head->fragment_);
  for (std::size_t i = 0; i < others.size(); ++i) {
    tables.push_back(others[i]->table_.template get<rmx_desc>());
    counts.push_back(others[i]->count_);
    masks.push_back(others[i]->mask_.template get<std::uint8_t>());
    push_dims(others[i]->fragment_);
    pairs.push_back(0);
    pairs.push_back(std::uint32_t(i + 1));
  }
  rmx_ctx* c = remap::context_cache::get(16, 8, 1, 1, 0, cell_ctx_kmax);
  std::vector<rmx_vote> votes(others.size());
  remap::check(rmx_kpm_cell_pairs(c, tables.data(), counts.data(), masks.data(), dims.data(),
                                  std::uint32_t(tables.size()), pairs.data(), std::uint32_t(others.size()),
                                  cell_size.width_, cell_size.height_, votes.data()),
               c, "rmx_kpm_cell_pairs");
  for (std::size_t i = 0; i < others.size(); ++i)
    if (votes[i].valid) {  // snippet::bind, fgs.hpp:56-65: a primary edge here, the reversed one there
      kpm::vote v{{votes[i].dx, votes[i].dy}, votes[i].count};
      head->edges_.push_back(delta{true, v, others[i]});
      others[i]->edges_.push_back(delta{false, v.reverse(), head});
    }
}

template <typename It>
void match_all(It first, It last) {  // fgs.hpp:135-140
  if (first == last) return;
  for (auto rest = std::next(first); rest != last; ++first, ++rest) match_partial(first, rest, last);
}

inline void unbind(snippet_iterator_t self) {  // fgs.hpp:67-71: drop the mirrored edges held by the others
  for (auto& e : self->edges_) e.other_->edges_.remove_if([&](delta const& d) { return d.other_ == self; });
}

}  // namespace details

template <typename Iter>
std::vector<fgm::fragment> splice(Iter first, Iter last) {
  using namespace details;
  snippet_list snippets;
  for (; first != last; ++first) snippets.push_back(snippet{std::move(*first), {}, {}, 0, {}});
  extract_all(snippets.begin(), snippets.end());
  match_all(snippets.begin(), snippets.end());
  for (;;) {
    // select_match (fgs.hpp:142-163): first maximum of the vote count over primary edges
    snippet_iterator_t left = snippets.end();
    delta* edge = nullptr;
    for (auto it = snippets.begin(); it != snippets.end(); ++it)
      for (auto& e : it->edges_)
        if (e.primary_ && (!edge || e.vote_.count_ > edge->vote_.count_)) {
          left = it;
          edge = &e;
        }
    if (!edge) break;
    // splice_single (fgs.hpp:165-183)
    auto right = edge->other_;
    auto& dst = left->fragment_;
    dst.blit(dst.zero() + edge->vote_.offset_, std::move(right->fragment_));
    dst.normalize();
    snippets.push_front(extract_single(std::move(dst)));
    unbind(right);
    unbind(left);
    snippets.erase(right);
    snippets.erase(left);
    match_partial(snippets.begin(), std::next(snippets.begin()), snippets.end());
  }
  std::vector<fgm::fragment> result;
  result.reserve(snippets.size());
  for (auto& s : snippets) result.push_back(std::move(s.fragment_));
  return result;
}

}  // namespace fgs

namespace mpb {

// The two stages of mpb::builder that lie on the hot path (mpb.hpp:52-68), with the
// reference's adapter contract: get_compression(), get_callbacks().  build() of the
// reference also runs aws (before) and fdf/arf (after), which are outside this path.
template <typename Adapter>
class builder {
public:
  using adapter_type = Adapter;
  using feed_type = typename Adapter::feed_type;

  builder(adapter_type const& adapter) noexcept : adapter_{adapter} {}

  auto collect(feed_type& feed, mrl::dimensions_t const& window) {
    frc::collector collector{window};
    collector.collect(feed, adapter_.get_compression(), cb());
    auto result = collector.complete();
    cb()(std::string{"frc"}, result);
    return result;
  }

  auto splice(std::list<fgm::fragment>& fragments) {
    auto result = fgs::splice(fragments.begin(), fragments.end());
    cb()(std::string{"spl"}, result);
    return result;
  }

  // mpb.hpp:70-77 (needs the adapter's callbacks to accept the fdf per-frame call, main.cpp:150-163)
  auto filter(mrl::dimensions_t const& window, std::vector<fgm::fragment>& fragments) {
    auto result = fdf::filter(fragments, window, adapter_.get_compression(), cb());
    cb()(std::string{"fdf"}, result);
    return result;
  }

private:
  auto& cb() noexcept { return adapter_.get_callbacks(); }
  adapter_type adapter_;
};

}  // namespace mpb
