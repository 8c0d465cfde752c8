// table-writers.h -- the two output holders of the hot path (a13 of SURVEY.md 8a):
//   PosteriorWriter                          (latbin/lattice-to-word-frame-post.cc:66,139)
//   TableWriter<BasicTupleVectorHolder<...>> (util/basic-tuple-vector-holder.h:141-181;
//       segment <int32,int32,int32,double>    kwsbin2/lattice-word-index-segment.cc:77-78,
//       position <int32,int32,int32,int32,double> kwsbin2/lattice-word-index-position.cc:80-82)
// Wire formats: SURVEY.md Appendix B.5 / B.6; records are written in the order Write() is called.
#ifndef KLAT_HOST_TABLE_WRITERS_H_
#define KLAT_HOST_TABLE_WRITERS_H_
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/klat.h"
#include "compact-lattice.h"
#include "kaldi-io.h"

namespace klat {

class ArchiveWriter {
 public:
  explicit ArchiveWriter(const std::string& wspecifier) : spec_(ParseSpecifier(wspecifier, true)), out_(spec_.filename) {}
  void Close() { out_.Close(); }

 protected:
  std::ostream& Begin(const std::string& key) {
    std::ostream& os = out_.Stream();
    os << key << ' ';
    InitKaldiOutputStream(os, !spec_.text);
    return os;
  }
  bool binary() const { return !spec_.text; }
  void End() {
    if (spec_.text || spec_.filename == "-") out_.Stream().flush();
    if (!out_.Stream().good()) KLAT_ERR("Write failure on " << spec_.filename);
  }
  Specifier spec_;
  Output out_;
};

// vector<vector<pair<int32,BaseFloat>>>: one bracket group per frame in text mode.
class PosteriorWriter : public ArchiveWriter {
 public:
  typedef std::vector<std::vector<std::pair<int32_t, float>>> T;
  using ArchiveWriter::ArchiveWriter;
  void Write(const std::string& key, const T& post) {
    std::ostream& os = Begin(key);
    if (binary()) {
      WriteBasicType(os, true, (int32_t)post.size());
      for (const auto& frame : post) {
        WriteBasicType(os, true, (int32_t)frame.size());
        for (const auto& p : frame) {
          WriteBasicType(os, true, p.first);
          WriteBasicType(os, true, p.second);
        }
      }
    } else {
      for (const auto& frame : post) {
        os << "[ ";
        for (const auto& p : frame) os << p.first << ' ' << p.second << ' ';
        os << "] ";
      }
      os << '\n';
    }
    End();
  }
};

// CompactLatticeWriter (latbin/lattice-prune-dyn-beam.cc:158,190): "key " then the lattice, as text
// (a newline, OpenFst text lines, an empty line) or as the binary VectorFst stream.
class CompactLatticeWriter {
 public:
  explicit CompactLatticeWriter(const std::string& wspecifier) : spec_(ParseSpecifier(wspecifier, true)), out_(spec_.filename) {}
  void Write(const std::string& key, const CompactLattice& clat) {
    std::ostream& os = out_.Stream();
    os << key << ' ';
    if (spec_.text) {
      os.precision(9);
      WriteCompactLatticeText(os, clat);
    } else {
      WriteCompactLatticeBinary(os, clat);
    }
    if (spec_.text || spec_.filename == "-") os.flush();
    if (!os.good()) KLAT_ERR("Write failure on " << spec_.filename);
  }
  void Close() { out_.Close(); }

 private:
  Specifier spec_;
  Output out_;
};

// vector<int32> (Int32VectorWriter, latbin/lattice-best-path2.cc:75,199): BasicVectorHolder wire
// format -- text: the values separated by spaces, newline; binary: int32 size, then the values.
class Int32VectorWriter : public ArchiveWriter {
 public:
  using ArchiveWriter::ArchiveWriter;
  void Write(const std::string& key, const std::vector<int32_t>& v) {
    std::ostream& os = Begin(key);
    if (binary()) WriteBasicType(os, true, (int32_t)v.size());
    for (int32_t x : v) WriteBasicType(os, binary(), x);
    if (!binary()) os << '\n';
    End();
  }
};

// vector<tuple<int32,double>> (utterance index, kwsbin2/lattice-word-index-utterance.cc:267-268).
class UtteranceScoreWriter : public ArchiveWriter {
 public:
  using ArchiveWriter::ArchiveWriter;
  void Write(const std::string& key, const std::vector<std::pair<int32_t, double>>& rec) {
    std::ostream& os = Begin(key);
    if (binary()) WriteBasicType(os, true, (int32_t)rec.size());
    for (size_t i = 0; i < rec.size(); ++i) {
      WriteBasicType(os, binary(), rec[i].first);
      WriteBasicType(os, binary(), rec[i].second);
      if (!binary() && i + 1 < rec.size()) os << "; ";
    }
    if (!binary()) os << '\n';
    End();
  }
};

// vector<tuple<int32,int32,int32,double>> (segment index).
class SegmentScoreWriter : public ArchiveWriter {
 public:
  using ArchiveWriter::ArchiveWriter;
  void Write(const std::string& key, const klat_segment_record* rec, int64_t n) {
    std::ostream& os = Begin(key);
    if (binary()) WriteBasicType(os, true, (int32_t)n);
    for (int64_t i = 0; i < n; ++i) {
      WriteBasicType(os, binary(), rec[i].label);
      WriteBasicType(os, binary(), rec[i].t0);
      WriteBasicType(os, binary(), rec[i].t1);
      WriteBasicType(os, binary(), rec[i].logp);
      if (!binary() && i + 1 < n) os << "; ";
    }
    if (!binary()) os << '\n';
    End();
  }
};

// vector<tuple<int32,int32,int32,int32,double>> (position index: label,pos,t0,t1,logp).
class PositionScoreWriter : public ArchiveWriter {
 public:
  using ArchiveWriter::ArchiveWriter;
  void Write(const std::string& key, const klat_position_record* rec, int64_t n) {
    std::ostream& os = Begin(key);
    if (binary()) WriteBasicType(os, true, (int32_t)n);
    for (int64_t i = 0; i < n; ++i) {
      WriteBasicType(os, binary(), rec[i].label);
      WriteBasicType(os, binary(), rec[i].pos);
      WriteBasicType(os, binary(), rec[i].t0);
      WriteBasicType(os, binary(), rec[i].t1);
      WriteBasicType(os, binary(), rec[i].logp);
      if (!binary() && i + 1 < n) os << "; ";
    }
    if (!binary()) os << '\n';
    End();
  }
};

}  // namespace klat
#endif
