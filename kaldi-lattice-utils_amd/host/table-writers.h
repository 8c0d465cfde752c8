// table-writers.h -- the two output holders of the hot path (a13 of SURVEY.md 8a):
//   PosteriorWriter                          (latbin/lattice-to-word-frame-post.cc:66,139)
//   TableWriter<BasicTupleVectorHolder<...>> (util/basic-tuple-vector-holder.h:141-181;
//       segment <int32,int32,int32,double>    kwsbin2/lattice-word-index-segment.cc:77-78,
//       position <int32,int32,int32,int32,double> kwsbin2/lattice-word-index-position.cc:80-82)
// Wire formats: SURVEY.md Appendix B.5 / B.6; records are written in the order Write() is called.
#ifndef KLAT_HOST_TABLE_WRITERS_H_
#define KLAT_HOST_TABLE_WRITERS_H_
#include <cstdint>
#include <sstream>
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
  bool binary() const { return !spec_.text; }
  // "key " and a value some Format() made for this writer's mode (formatting is the expensive part of a
  // text table and is done on several threads; the writes stay in input order)
  void WriteFormatted(const std::string& key, const std::string& value) {
    std::ostream& os = out_.Stream();
    os << key << ' ';
    if (binary()) os.write("\0B", 2);
    os.write(value.data(), (std::streamsize)value.size());
    End();
  }

 protected:
  std::ostream& Begin(const std::string& key) {
    std::ostream& os = out_.Stream();
    os << key << ' ';
    InitKaldiOutputStream(os, !spec_.text);
    return os;
  }
  // Binary records are assembled in one buffer and written with one call: a value is Kaldi's
  // 1-byte size tag + the little-endian bytes (WriteBasicType), and a stream call per value is what
  // the tools' time went into.
  template <class V>
  static void Put(std::string* buf, V v) {
    buf->push_back((char)sizeof(V));
    buf->append(reinterpret_cast<const char*>(&v), sizeof(V));
  }
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
  // the same record straight from klat_frame_post's flat arrays: frames [0, num_frames) of one
  // utterance, entries of frame k at [frame_off[k], frame_off[k + 1])
  static void FormatFlat(bool binary, int32_t num_frames, const int64_t* frame_off, const int32_t* labels, const float* logps,
                         std::string* out) {
    out->clear();
    if (binary) {
      out->reserve((size_t)(frame_off[num_frames] - frame_off[0]) * 10 + (size_t)num_frames * 5 + 16);
      Put(out, num_frames);
      for (int32_t k = 0; k < num_frames; ++k) {
        Put(out, (int32_t)(frame_off[k + 1] - frame_off[k]));
        for (int64_t i = frame_off[k]; i < frame_off[k + 1]; ++i) {
          Put(out, labels[i]);
          Put(out, logps[i]);
        }
      }
    } else {
      std::ostringstream os;
      InitKaldiOutputStream(os, false);
      for (int32_t k = 0; k < num_frames; ++k) {
        os << "[ ";
        for (int64_t i = frame_off[k]; i < frame_off[k + 1]; ++i) os << labels[i] << ' ' << logps[i] << ' ';
        os << "] ";
      }
      os << '\n';
      *out = os.str();
    }
  }
  void WriteFlat(const std::string& key, int32_t num_frames, const int64_t* frame_off, const int32_t* labels,
                 const float* logps) {
    FormatFlat(binary(), num_frames, frame_off, labels, logps, &buf_);
    WriteFormatted(key, buf_);
  }
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

 private:
  std::string buf_;
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

// one BaseFloat per key (Kaldi's BaseFloatWriter): the cost table of lattice-best-path2 --nbest
class BaseFloatWriter : public ArchiveWriter {
 public:
  using ArchiveWriter::ArchiveWriter;
  void Write(const std::string& key, float v) {
    std::ostream& os = Begin(key);
    WriteBasicType(os, binary(), v);
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
  static void Format(bool binary, const klat_segment_record* rec, int64_t n, std::string* out) {
    out->clear();
    if (binary) {
      out->reserve((size_t)n * 24 + 8);
      Put(out, (int32_t)n);
      for (int64_t i = 0; i < n; ++i) Put(out, rec[i].label), Put(out, rec[i].t0), Put(out, rec[i].t1), Put(out, rec[i].logp);
      return;
    }
    std::ostringstream os;
    InitKaldiOutputStream(os, false);
    for (int64_t i = 0; i < n; ++i) {
      WriteBasicType(os, false, rec[i].label);
      WriteBasicType(os, false, rec[i].t0);
      WriteBasicType(os, false, rec[i].t1);
      WriteBasicType(os, false, rec[i].logp);
      if (i + 1 < n) os << "; ";
    }
    os << '\n';
    *out = os.str();
  }
  void Write(const std::string& key, const klat_segment_record* rec, int64_t n) {
    Format(binary(), rec, n, &buf_);
    WriteFormatted(key, buf_);
  }

 private:
  std::string buf_;
};

// vector<tuple<int32,int32,int32,int32,double>> (position index: label,pos,t0,t1,logp).
class PositionScoreWriter : public ArchiveWriter {
 public:
  using ArchiveWriter::ArchiveWriter;
  static void Format(bool binary, const klat_position_record* rec, int64_t n, std::string* out) {
    out->clear();
    if (binary) {
      out->reserve((size_t)n * 29 + 8);
      Put(out, (int32_t)n);
      for (int64_t i = 0; i < n; ++i)
        Put(out, rec[i].label), Put(out, rec[i].pos), Put(out, rec[i].t0), Put(out, rec[i].t1), Put(out, rec[i].logp);
      return;
    }
    std::ostringstream os;
    InitKaldiOutputStream(os, false);
    for (int64_t i = 0; i < n; ++i) {
      WriteBasicType(os, false, rec[i].label);
      WriteBasicType(os, false, rec[i].pos);
      WriteBasicType(os, false, rec[i].t0);
      WriteBasicType(os, false, rec[i].t1);
      WriteBasicType(os, false, rec[i].logp);
      if (i + 1 < n) os << "; ";
    }
    os << '\n';
    *out = os.str();
  }
  void Write(const std::string& key, const klat_position_record* rec, int64_t n) {
    Format(binary(), rec, n, &buf_);
    WriteFormatted(key, buf_);
  }

 private:
  std::string buf_;
};

}  // namespace klat
#endif
