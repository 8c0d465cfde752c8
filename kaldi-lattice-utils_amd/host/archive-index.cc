// archive-index.cc -- see archive-index.h.
#include "archive-index.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cerrno>
#include <cstring>
#include <istream>
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
      if (p + n > size_) bad();
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

void MappedArchive::Read(const ArchiveEntry& e, CompactLattice* clat, bool keep_strings) const {
  MemBuf buf(data_ + e.begin, data_ + e.end);
  std::istream is(&buf);
  const bool ok = e.binary ? ReadCompactLatticeBinary(is, clat, keep_strings) : ReadCompactLatticeText(is, clat, keep_strings);
  if (!ok) KLAT_ERR("Failed reading lattice for key \"" << e.key << "\"");
}

}  // namespace klat
