// archive-index.h -- entry boundaries of a lattice archive without building the lattices.
//
// The reference reads every lattice exactly once, on the thread that hands it to the TaskSequencer
// (kwsbin2/lattice-word-index-segment.cc:261-274).  With one PROCESS per GPU every rank has to know
// where the entries of the archive begin -- a Kaldi archive has no index and a binary
// CompactLattice no length prefix -- but only the rank that owns an entry has to parse it.  A plain
// file is therefore mapped into memory and SCANNED with pointer arithmetic (key, then the value's
// byte range: a text entry ends at its blank line, a binary one is walked state by state over its
// variable-length weights), a few nanoseconds per arc against ~50 for the parser; the byte length of
// an entry stands in for its arc count when utterances are dealt to the ranks.
#ifndef KLAT_HOST_ARCHIVE_INDEX_H_
#define KLAT_HOST_ARCHIVE_INDEX_H_
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "compact-lattice.h"

namespace klat {

struct ArchiveEntry {
  std::string key;
  size_t begin = 0, end = 0;  // the value's bytes (behind "key "), [begin, end)
  bool binary = false;
};

// A lattice as arrays, no vector per state: what the batch needs of it and what a binary entry holds, read
// by pointer arithmetic straight from the mapped bytes (the object form costs an allocation per state and a
// stream call per field: ~50 ns per arc against ~5).
struct FlatLattice {
  int32_t start = -1;                 // CompactLattice::kNoStateId for an empty lattice
  std::vector<int32_t> row;           // [S + 1] first arc of every state
  std::vector<int32_t> dst, ilabel, olabel, len;  // per arc: nextstate, labels, length of the transition-id string
  std::vector<float> graph, acoustic;
  std::vector<float> final_graph, final_acoustic;  // +inf, +inf: not final
  std::vector<int32_t> final_len;
  int32_t NumStates() const { return row.empty() ? 0 : (int32_t)row.size() - 1; }
  int64_t NumArcs() const { return (int64_t)dst.size(); }
  void Clear() {
    start = -1;
    row.clear(), dst.clear(), ilabel.clear(), olabel.clear(), len.clear();
    graph.clear(), acoustic.clear(), final_graph.clear(), final_acoustic.clear(), final_len.clear();
  }
};
// the same lattice from the object form
void ToFlat(const CompactLattice& clat, FlatLattice* out);
// start state 0 and every arc to a higher state (what TopSortCompactLatticeIfNeeded leaves alone)
bool IsTopSortedFromZero(const FlatLattice& lat);
// CompactLatticeStateTimes on the flat form (same checks, warnings and return value)
int32_t FlatLatticeStateTimes(const FlatLattice& lat, std::vector<int32_t>* times);

class MappedArchive {
 public:
  // an rspecifier that names a plain file ("ark:path"): not standard input, not a command
  static bool Mappable(const std::string& rspecifier);
  explicit MappedArchive(const std::string& rspecifier);
  ~MappedArchive();
  MappedArchive(const MappedArchive&) = delete;
  MappedArchive& operator=(const MappedArchive&) = delete;
  bool Done();                      // nothing left to scan
  void ScanNext(ArchiveEntry* e);   // KLAT_ERR on a malformed entry
  // parses one scanned entry with the archive readers of compact-lattice.h
  void Read(const ArchiveEntry& e, CompactLattice* clat, bool keep_strings = false) const;
  // a binary CompactLattice entry straight into arrays; false for any other entry (text, Lattice format:
  // the caller takes Read())
  bool ReadFlat(const ArchiveEntry& e, FlatLattice* lat) const;
  size_t size() const { return size_; }

 private:
  const char* data_ = nullptr;
  size_t size_ = 0, pos_ = 0;
  std::string name_;
};

}  // namespace klat
#endif
