// archive-index.cc -- see archive-index.h.
#include "archive-index.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstring>
#include <istream>
#include <limits>
#include <streambuf>

namespace klat {

namespace {

struct MemBuf : std::streambuf {
  MemBuf(const char* b, const char* e) {
    char* p = const_cast<char*>(b);
    setg(p, p, const_cast<char*>(e));
  }
};

inline bool IsSpace(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

}  // namespace

bool MappedArchive::Mappable(const std::string& rspecifier) {
  const Specifier sp = ParseSpecifier(rspecifier, false);
  if (sp.filename.empty() || sp.filename == "-" || sp.filename.back() == '|') return false;
  struct stat st;
  return stat(sp.filename.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

MappedArchive::MappedArchive(const std::string& rspecifier) {
  const Specifier sp = ParseSpecifier(rspecifier, false);
  name_ = sp.filename;
  const int fd = open(name_.c_str(), O_RDONLY);
  if (fd < 0) KLAT_ERR("Error opening input stream " << name_ << ": " << std::strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    KLAT_ERR("fstat " << name_ << ": " << std::strerror(errno));
  }
  size_ = (size_t)st.st_size;
  if (size_ > 0) {
    void* p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) {
      close(fd);
      KLAT_ERR("mmap " << name_ << ": " << std::strerror(errno));
    }
    data_ = static_cast<const char*>(p);
    (void)madvise(p, size_, MADV_SEQUENTIAL);
  }
  close(fd);
}

MappedArchive::~MappedArchive() {
  if (data_) munmap(const_cast<char*>(data_), size_);
}

bool MappedArchive::Done() {
  while (pos_ < size_ && IsSpace(data_[pos_])) ++pos_;  // whitespace between entries
  return pos_ >= size_;
}

void MappedArchive::ScanNext(ArchiveEntry* e) {
  if (Done()) KLAT_ERR("ScanNext past the end of " << name_);
  const char* d = data_;
  size_t p = pos_;
  const size_t k0 = p;
  while (p < size_ && !IsSpace(d[p])) ++p;
  e->key.assign(d + k0, p - k0);
  while (p < size_ && (d[p] == ' ' || d[p] == '\t')) ++p;
  auto bad = [&]() { KLAT_ERR("Failed reading lattice for key \"" << e->key << "\""); };
  if (p < size_ && (unsigned char)d[p] == 0xD6) {  // the OpenFst magic number: a binary value
    e->binary = true;
    e->begin = p;
    auto need = [&](size_t n) {
      if (n > size_ - p) bad();  // (p <= size_ always; no sum that could wrap)
    };
    auto i32 = [&]() -> int32_t {
      need(4);
      int32_t v;
      std::memcpy(&v, d + p, 4);
      p += 4;
      return v;
    };
    auto i64 = [&]() -> int64_t {
      need(8);
      int64_t v;
      std::memcpy(&v, d + p, 8);
      p += 8;
      return v;
    };
    (void)i32();  // magic
    const int32_t n1 = i32();
    if (n1 < 0 || n1 > 1024) bad();
    need((size_t)n1);
    p += (size_t)n1;  // fst type
    const int32_t n2 = i32();
    if (n2 < 0 || n2 > 1024) bad();
    need((size_t)n2);
    const bool compact = n2 >= 7 && std::memcmp(d + p, "compact", 7) == 0;
    p += (size_t)n2;  // arc type
    (void)i32();      // version
    const int32_t flags = i32();
    if (flags & 3) KLAT_ERR("Lattice archives with embedded symbol tables are not supported");
    (void)i64();  // properties
    (void)i64();  // start
    const int64_t num_states = i64();
    (void)i64();  // num_arcs
    if (num_states < 0 || num_states > (int64_t)1 << 30) bad();
    auto skip_weight = [&]() {
      need(8);
      p += 8;
      if (compact) {
        const int32_t n = i32();
        if (n < 0) bad();
        need((size_t)n * 4);
        p += (size_t)n * 4;
      }
    };
    for (int64_t s = 0; s < num_states; ++s) {
      skip_weight();
      const int64_t narcs = i64();
      if (narcs < 0) bad();
      if (!compact) {  // a Lattice arc is 20 bytes
        if (narcs > (int64_t)((size_ - p) / 20)) bad();
        need((size_t)narcs * 20);
        p += (size_t)narcs * 20;
      } else {
        for (int64_t a = 0; a < narcs; ++a) {
          need(8);
          p += 8;  // ilabel, olabel
          skip_weight();
          need(4);
          p += 4;  // nextstate
        }
      }
    }
    e->end = p;
  } else {  // text: the rest of the key's line, then lines up to a blank one
    e->binary = false;
    if (p < size_ && d[p] == '\r') ++p;
    if (p < size_ && d[p] == '\n') ++p;
    e->begin = p;
    for (;;) {
      if (p >= size_) break;
      const char* nl = static_cast<const char*>(std::memchr(d + p, '\n', size_ - p));
      const size_t le = nl ? (size_t)(nl - d) : size_;
      bool blank = true;
      for (size_t q = p; q < le; ++q)
        if (!IsSpace(d[q])) {
          blank = false;
          break;
        }
      p = nl ? le + 1 : size_;
      if (blank) break;
    }
    e->end = p;
  }
  pos_ = p;
}

bool MappedArchive::ReadFlat(const ArchiveEntry& e, FlatLattice* lat) const {
  lat->Clear();
  if (!e.binary) return false;
  const char* d = data_;
  size_t p = e.begin;
  const size_t end = e.end;
  auto bad = [&]() { KLAT_ERR("Failed reading lattice for key \"" << e.key << "\""); };
  auto need = [&](size_t n) {
    if (n > end - p) bad();  // (p <= end always; no sum that could wrap)
  };
  auto i32 = [&]() -> int32_t {
    need(4);
    int32_t v;
    std::memcpy(&v, d + p, 4);
    p += 4;
    return v;
  };
  auto i64 = [&]() -> int64_t {
    need(8);
    int64_t v;
    std::memcpy(&v, d + p, 8);
    p += 8;
    return v;
  };
  auto f32 = [&]() -> float {
    need(4);
    float v;
    std::memcpy(&v, d + p, 4);
    p += 4;
    return v;
  };
  (void)i32();  // magic (the scan saw it)
  const int32_t n1 = i32();
  if (n1 < 0 || n1 > 1024) bad();  // (the scan's bounds: an entry need not have come through it)
  need((size_t)n1);
  const bool vector_fst = n1 == 6 && std::memcmp(d + p, "vector", 6) == 0;
  p += (size_t)n1;
  const int32_t n2 = i32();
  if (n2 < 0 || n2 > 1024) bad();
  need((size_t)n2);
  const bool compact = n2 >= 7 && std::memcmp(d + p, "compact", 7) == 0;
  p += (size_t)n2;
  if (!vector_fst || !compact) return false;  // (a Lattice-format entry goes through ConvertLattice)
  (void)i32();  // version
  (void)i32();  // flags (the scan refused symbol tables)
  (void)i64();  // properties
  const int64_t start = i64(), num_states = i64(), num_arcs_hint = i64();
  if (num_states < 0 || num_states > (int64_t)1 << 30) bad();
  lat->row.resize((size_t)num_states + 1);
  lat->final_graph.resize((size_t)num_states);
  lat->final_acoustic.resize((size_t)num_states);
  lat->final_len.resize((size_t)num_states);
  if (num_arcs_hint > 0 && num_arcs_hint < ((int64_t)1 << 31)) {
    const size_t r = (size_t)num_arcs_hint;
    lat->dst.reserve(r), lat->ilabel.reserve(r), lat->olabel.reserve(r), lat->len.reserve(r), lat->graph.reserve(r), lat->acoustic.reserve(r);
  }
  for (int64_t s = 0; s < num_states; ++s) {
    lat->final_graph[(size_t)s] = f32();
    lat->final_acoustic[(size_t)s] = f32();
    int32_t n = i32();
    if (n < 0) bad();
    need((size_t)n * 4);
    p += (size_t)n * 4;
    lat->final_len[(size_t)s] = n;
    const int64_t narcs = i64();
    if (narcs < 0 || narcs > (int64_t)(end - p) / 20) bad();
    lat->row[(size_t)s] = (int32_t)lat->dst.size();
    for (int64_t k = 0; k < narcs; ++k) {
      const int32_t il = i32(), ol = i32();
      const float g = f32(), a = f32();
      n = i32();
      if (n < 0) bad();
      need((size_t)n * 4);
      p += (size_t)n * 4;
      const int32_t next = i32();
      if (next < 0 || next >= num_states) bad();
      lat->ilabel.push_back(il), lat->olabel.push_back(ol), lat->graph.push_back(g), lat->acoustic.push_back(a), lat->len.push_back(n),
          lat->dst.push_back(next);
    }
  }
  lat->row[(size_t)num_states] = (int32_t)lat->dst.size();
  lat->start = num_states > 0 ? (int32_t)start : -1;
  if (num_states > 0 && (start < 0 || start >= num_states)) bad();
  return true;
}

void ToFlat(const CompactLattice& clat, FlatLattice* out) {
  out->Clear();
  const int32_t S = clat.Start() == CompactLattice::kNoStateId ? 0 : clat.NumStates();
  out->start = S > 0 ? clat.Start() : -1;
  out->row.resize((size_t)S + 1);
  out->final_graph.resize((size_t)S), out->final_acoustic.resize((size_t)S), out->final_len.resize((size_t)S);
  for (int32_t s = 0; s < S; ++s) {
    const CompactLatticeState& st = clat.State(s);
    out->row[(size_t)s] = (int32_t)out->dst.size();
    out->final_graph[(size_t)s] = st.final_graph, out->final_acoustic[(size_t)s] = st.final_acoustic, out->final_len[(size_t)s] = st.final_string_len;
    for (const auto& arc : st.arcs) {
      out->dst.push_back(arc.nextstate), out->ilabel.push_back(arc.ilabel), out->olabel.push_back(arc.olabel);
      out->graph.push_back(arc.graph), out->acoustic.push_back(arc.acoustic), out->len.push_back(arc.string_len);
    }
  }
  out->row[(size_t)S] = (int32_t)out->dst.size();
}

bool IsTopSortedFromZero(const FlatLattice& lat) {
  const int32_t S = lat.NumStates();
  if (S == 0) return true;
  if (lat.start != 0) return false;
  for (int32_t s = 0; s < S; ++s)
    for (int32_t i = lat.row[(size_t)s]; i < lat.row[(size_t)s + 1]; ++i)
      if (lat.dst[(size_t)i] <= s) return false;
  return true;
}

int32_t FlatLatticeStateTimes(const FlatLattice& lat, std::vector<int32_t>* times) {
  if (!IsTopSortedFromZero(lat)) KLAT_ERR("Input lattice must be topologically sorted.");
  const int32_t S = lat.NumStates();
  times->clear();
  if (S == 0) return 0;
  times->assign((size_t)S, -1);
  (*times)[0] = 0;
  int32_t utt_len = -1;
  for (int32_t s = 0; s < S; ++s) {
    const int32_t cur = (*times)[(size_t)s];
    for (int32_t i = lat.row[(size_t)s]; i < lat.row[(size_t)s + 1]; ++i) {
      int32_t& t = (*times)[(size_t)lat.dst[(size_t)i]];
      if (t == -1) t = cur + lat.len[(size_t)i];
      else if (t != cur + lat.len[(size_t)i])
        KLAT_ERR("CompactLatticeStateTimes: inconsistent times at state " << lat.dst[(size_t)i] << " (lattice is not aligned)");
    }
    const bool fin = !(lat.final_graph[(size_t)s] == std::numeric_limits<float>::infinity() &&
                       lat.final_acoustic[(size_t)s] == std::numeric_limits<float>::infinity());
    if (fin) {
      const int32_t this_len = cur + lat.final_len[(size_t)s];
      if (utt_len == -1) {
        utt_len = this_len;
      } else if (this_len != utt_len) {
        KLAT_WARN("Utterance does not seem to have a consistent length.");
        utt_len = std::max(utt_len, this_len);
      }
    }
  }
  if (utt_len == -1) {
    KLAT_WARN("Utterance does not have a final-state.");
    return 0;
  }
  return utt_len;
}

void MappedArchive::Read(const ArchiveEntry& e, CompactLattice* clat, bool keep_strings) const {
  MemBuf buf(data_ + e.begin, data_ + e.end);
  std::istream is(&buf);
  const bool ok = e.binary ? ReadCompactLatticeBinary(is, clat, keep_strings) : ReadCompactLatticeText(is, clat, keep_strings);
  if (!ok) KLAT_ERR("Failed reading lattice for key \"" << e.key << "\"");
}

}  // namespace klat
