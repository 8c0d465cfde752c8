// compact-lattice.cc -- see compact-lattice.h.  Wire formats: SURVEY.md Appendix B.2-B.4.
#include "compact-lattice.h"

#include <cstdlib>
#include <cstring>
#include <sstream>

namespace klat {

static const int32_t kFstMagicNumber = 2125659606;
static const float kInf = std::numeric_limits<float>::infinity();

int64_t CompactLattice::NumArcs() const {
  int64_t n = 0;
  for (const auto& s : states_) n += (int64_t)s.arcs.size();
  return n;
}

// "g,a,t1_t2_.."  ->  graph, acoustic, number of transition ids.  Missing pieces = One.
static bool ParseCompactWeight(const std::string& tok, float* g, float* a, int32_t* len,
                               std::vector<int32_t>* pool = nullptr, int32_t* off = nullptr) {
  *g = 0.f;
  *a = 0.f;
  *len = 0;
  size_t c1 = tok.find(',');
  const std::string s1 = tok.substr(0, c1);
  char* end = nullptr;
  if (!s1.empty()) {
    *g = std::strtof(s1.c_str(), &end);
    if (end == s1.c_str() || *end) return false;
  }
  if (c1 == std::string::npos) return true;
  size_t c2 = tok.find(',', c1 + 1);
  const std::string s2 = tok.substr(c1 + 1, c2 == std::string::npos ? std::string::npos : c2 - c1 - 1);
  if (!s2.empty()) {
    *a = std::strtof(s2.c_str(), &end);
    if (end == s2.c_str() || *end) return false;
  }
  if (c2 == std::string::npos) return true;
  const std::string s3 = tok.substr(c2 + 1);
  if (!s3.empty()) {
    int32_t n = 1;
    for (char ch : s3) {
      if (ch == '_') ++n;
      else if (!(isdigit((unsigned char)ch) || ch == '-')) return false;
    }
    *len = n;
    if (pool) {  // keep the ids themselves
      *off = (int32_t)pool->size();
      const char* c = s3.c_str();
      for (int32_t i = 0; i < n; ++i) {
        char* e = nullptr;
        pool->push_back((int32_t)std::strtol(c, &e, 10));
        if (e == c) return false;
        c = (*e == '_') ? e + 1 : e;
      }
    }
  }
  return true;
}

static bool IsInt(const std::string& s) {
  if (s.empty()) return false;
  char* end = nullptr;
  std::strtol(s.c_str(), &end, 10);
  return *end == 0;
}

bool ReadCompactLatticeText(std::istream& is, CompactLattice* clat, bool keep_strings) {
  clat->DeleteStates();
  std::vector<int32_t>* pool = keep_strings ? &clat->StringPool() : nullptr;
  std::string line;
  auto need_state = [&](int32_t s) {
    while (s >= clat->NumStates()) clat->AddState();
  };
  while (std::getline(is, line)) {
    std::vector<std::string> t;
    {
      std::istringstream ls(line);
      std::string w;
      while (ls >> w) t.push_back(w);
    }
    if (t.empty()) break;  // blank line terminates the entry (right after the key: empty lattice)
    if (!IsInt(t[0])) return false;
    const int32_t s = (int32_t)std::atol(t[0].c_str());
    if (s < 0) return false;
    need_state(s);
    if (clat->Start() == CompactLattice::kNoStateId) clat->SetStart(s);
    if (t.size() <= 2) {  // final state line: "<state> [weight]"
      CompactLatticeState& st = clat->State(s);
      st.final_graph = 0.f;
      st.final_acoustic = 0.f;
      st.final_string_len = 0;
      if (t.size() == 2 && !ParseCompactWeight(t[1], &st.final_graph, &st.final_acoustic, &st.final_string_len, pool, &st.final_string_off)) return false;
      continue;
    }
    if (!IsInt(t[1]) || !IsInt(t[2])) return false;
    CompactLatticeArc arc;
    arc.nextstate = (int32_t)std::atol(t[1].c_str());
    if (arc.nextstate < 0) return false;
    need_state(arc.nextstate);
    if (t.size() == 3 || (t.size() == 4 && t[3].find(',') != std::string::npos)) {
      // CompactLattice (acceptor) format: src dst label [g,a,tids]
      arc.ilabel = arc.olabel = (int32_t)std::atol(t[2].c_str());
      if (t.size() == 4 && !ParseCompactWeight(t[3], &arc.graph, &arc.acoustic, &arc.string_len, pool, &arc.string_off)) return false;
    } else if (t.size() == 4 || t.size() == 5) {
      // Lattice format: src dst ilabel(transition-id) olabel(word) [g,a].  Converted arc by arc:
      // the word is the olabel, the transition-id string is [ilabel] when ilabel != 0.  This
      // equals ConvertLattice whenever no chain can be merged (every arc carries an olabel).
      if (!IsInt(t[3])) return false;
      const int32_t il = (int32_t)std::atol(t[2].c_str());
      arc.ilabel = arc.olabel = (int32_t)std::atol(t[3].c_str());
      arc.string_len = il != 0 ? 1 : 0;
      if (pool && il != 0) {
        arc.string_off = (int32_t)pool->size();
        pool->push_back(il);
      }
      int32_t dummy;
      if (t.size() == 5 && !ParseCompactWeight(t[4], &arc.graph, &arc.acoustic, &dummy)) return false;
    } else {
      return false;
    }
    clat->State(s).arcs.push_back(arc);
  }
  return true;
}

void WriteCompactLatticeText(std::ostream& os, const CompactLattice& clat) {
  os << '\n';
  auto weight = [&](float g, float a, int32_t len, int32_t off) {
    os << g << ',' << a << ',';
    for (int32_t i = 0; i < len; ++i)  // the ids, or 1s when the reader did not keep them
      os << (i ? "_" : "") << (off >= 0 ? clat.StringPool()[(size_t)(off + i)] : 1);
  };
  // OpenFst text order: start state first
  std::vector<int32_t> order;
  if (clat.Start() != CompactLattice::kNoStateId) order.push_back(clat.Start());
  for (int32_t s = 0; s < clat.NumStates(); ++s)
    if (s != clat.Start()) order.push_back(s);
  for (int32_t s : order) {
    const CompactLatticeState& st = clat.State(s);
    for (const auto& arc : st.arcs) {
      os << s << '\t' << arc.nextstate << '\t' << arc.ilabel << '\t';
      weight(arc.graph, arc.acoustic, arc.string_len, arc.string_off);
      os << '\n';
    }
    if (st.IsFinal()) {
      os << s << '\t';
      weight(st.final_graph, st.final_acoustic, st.final_string_len, st.final_string_off);
      os << '\n';
    }
  }
  os << '\n';
}

// ---- binary: OpenFst VectorFst stream (Appendix B.4) ----
// (straight from the stream buffer: istream::read builds a sentry per call, and a lattice is read
// field by field -- seven reads per arc)
static bool rd_bytes(std::istream& is, void* p, std::streamsize n) {
  if (is.rdbuf()->sgetn(static_cast<char*>(p), n) == n) return true;
  is.setstate(std::ios::failbit | std::ios::eofbit);
  return false;
}
template <class T>
static bool rd(std::istream& is, T* v) {
  return rd_bytes(is, v, (std::streamsize)sizeof(T));
}
static bool rd_string(std::istream& is, std::string* s) {
  int32_t n;
  if (!rd(is, &n) || n < 0 || n > 1024) return false;
  s->resize((size_t)n);
  return n == 0 || rd_bytes(is, &(*s)[0], n);
}
template <class T>
static void wr(std::ostream& os, T v) {
  os.write(reinterpret_cast<const char*>(&v), sizeof(T));
}
static void wr_string(std::ostream& os, const std::string& s) {
  wr<int32_t>(os, (int32_t)s.size());
  os.write(s.data(), (std::streamsize)s.size());
}

bool ReadCompactLatticeBinary(std::istream& is, CompactLattice* clat, bool keep_strings) {
  clat->DeleteStates();
  std::vector<int32_t>* pool = keep_strings ? &clat->StringPool() : nullptr;
  int32_t magic, version, flags;
  std::string fst_type, arc_type;
  uint64_t properties;
  int64_t start, num_states, num_arcs;
  if (!rd(is, &magic) || magic != kFstMagicNumber) return false;
  if (!rd_string(is, &fst_type) || !rd_string(is, &arc_type)) return false;
  if (!rd(is, &version) || !rd(is, &flags) || !rd(is, &properties) || !rd(is, &start) ||
      !rd(is, &num_states) || !rd(is, &num_arcs))
    return false;
  if (fst_type != "vector") KLAT_ERR("Unsupported FST type \"" << fst_type << "\" in lattice archive");
  const bool compact = arc_type.compare(0, 7, "compact") == 0;
  if (!compact && arc_type.compare(0, 7, "lattice") != 0)
    KLAT_ERR("Unsupported arc type \"" << arc_type << "\" in lattice archive");
  if (flags & 3) KLAT_ERR("Lattice archives with embedded symbol tables are not supported");
  if (num_states < 0 || num_states > (int64_t)1 << 30) return false;
  auto read_weight = [&](float* g, float* a, int32_t* len, int32_t* off) -> bool {
    if (!rd(is, g) || !rd(is, a)) return false;
    *len = 0;
    if (compact) {
      int32_t n;
      if (!rd(is, &n) || n < 0) return false;
      if (pool && n > 0) {
        *off = (int32_t)pool->size();
        pool->resize(pool->size() + (size_t)n);
        if (!rd_bytes(is, pool->data() + *off, (std::streamsize)n * 4)) return false;
      } else {
        int32_t skip[64];
        for (int32_t left = n; left > 0; left -= 64)
          if (!rd_bytes(is, skip, (std::streamsize)std::min(left, 64) * 4)) return false;
      }
      *len = n;
    }
    return (bool)is;
  };
  for (int64_t s = 0; s < num_states; ++s) {
    clat->AddState();
    CompactLatticeState& st = clat->State((int32_t)s);
    if (!read_weight(&st.final_graph, &st.final_acoustic, &st.final_string_len, &st.final_string_off)) return false;
    int64_t narcs;
    if (!rd(is, &narcs) || narcs < 0) return false;
    st.arcs.resize((size_t)narcs);
    for (auto& arc : st.arcs) {
      int32_t il, ol;
      if (!rd(is, &il) || !rd(is, &ol)) return false;
      if (!read_weight(&arc.graph, &arc.acoustic, &arc.string_len, &arc.string_off)) return false;
      if (!rd(is, &arc.nextstate)) return false;
      if (compact) {
        arc.ilabel = il;
        arc.olabel = ol;
      } else {  // Lattice: ilabel = transition id, olabel = word (arc-by-arc conversion, see text reader)
        arc.ilabel = arc.olabel = ol;
        arc.string_len = il != 0 ? 1 : 0;
        if (pool && il != 0) {
          arc.string_off = (int32_t)pool->size();
          pool->push_back(il);
        }
      }
    }
  }
  clat->SetStart(num_states > 0 ? (int32_t)start : CompactLattice::kNoStateId);
  return true;
}

void WriteCompactLatticeBinary(std::ostream& os, const CompactLattice& clat) {
  wr<int32_t>(os, kFstMagicNumber);
  wr_string(os, "vector");
  wr_string(os, "compactlattice44");
  wr<int32_t>(os, 2);   // version
  wr<int32_t>(os, 0);   // flags: no symbol tables
  wr<uint64_t>(os, 0);  // properties (unknown)
  wr<int64_t>(os, clat.Start());
  wr<int64_t>(os, clat.NumStates());
  wr<int64_t>(os, clat.NumArcs());
  auto write_weight = [&](float g, float a, int32_t len, int32_t off) {
    wr(os, g);
    wr(os, a);
    wr<int32_t>(os, len);
    for (int32_t i = 0; i < len; ++i) wr<int32_t>(os, off >= 0 ? clat.StringPool()[(size_t)(off + i)] : 1);
  };
  for (int32_t s = 0; s < clat.NumStates(); ++s) {
    const CompactLatticeState& st = clat.State(s);
    write_weight(st.final_graph, st.final_acoustic, st.final_string_len, st.final_string_off);
    wr<int64_t>(os, (int64_t)st.arcs.size());
    for (const auto& arc : st.arcs) {
      wr<int32_t>(os, arc.ilabel);
      wr<int32_t>(os, arc.olabel);
      write_weight(arc.graph, arc.acoustic, arc.string_len, arc.string_off);
      wr<int32_t>(os, arc.nextstate);
    }
  }
}

// ---- sequential archive reader ----
SequentialCompactLatticeReader::SequentialCompactLatticeReader(const std::string& rspecifier, bool keep_strings)
    : keep_strings_(keep_strings) {
  const Specifier sp = ParseSpecifier(rspecifier, false);
  input_.reset(new Input(sp.filename));
  Next();
}

void SequentialCompactLatticeReader::Next() {
  std::istream& is = input_->Stream();
  value_.DeleteStates();
  key_.clear();
  // skip whitespace between entries, then read the key token
  int c;
  while ((c = is.peek()) != EOF && isspace(c)) is.get();
  if (c == EOF) {
    done_ = true;
    return;
  }
  while ((c = is.peek()) != EOF && !isspace(c)) key_.push_back((char)is.get());
  // one separator, then either '\n' (text body follows) or the OpenFst magic number (binary)
  while ((c = is.peek()) == ' ' || c == '\t') is.get();
  bool ok;
  if (c == 0xD6) {
    ok = ReadCompactLatticeBinary(is, &value_, keep_strings_);
  } else {
    if (c == '\r') is.get(), c = is.peek();
    if (c == '\n') is.get();
    ok = ReadCompactLatticeText(is, &value_, keep_strings_);
  }
  if (!ok) KLAT_ERR("Failed reading lattice for key \"" << key_ << "\"");
}

}  // namespace klat
