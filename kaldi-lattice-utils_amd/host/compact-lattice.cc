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

// ---- Lattice (non-compact) entries fed to a CompactLattice reader ----
// The reference does this (kwsbin2/README.md:176 runs lattice-char-index-segment on
// egs/lattice.char.ark.txt, a Lattice-format file); Kaldi's holder then converts with
// ConvertLattice(lat, &clat) [upstream, fstext/lattice-utils-inl.h], i.e. Factor() [upstream,
// fstext/factor-inl.h] followed by a per-arc rewrite.  Restated:
//   * a state is REMOVED if it is neither initial nor final, has exactly one arc in and exactly one
//     arc out, and that outgoing arc carries no output label: it lies inside a linear chain whose
//     word (if any) sits on the chain's first arc;
//   * every arc leaving a kept state absorbs the chain behind it: the weights multiply
//     (LatticeWeight Times = the float sums of the two components, left to right), the non-zero
//     input labels (transition ids) concatenate;
//   * the compact arc is (olabel, olabel, weight, ids); final weights carry an empty string; kept
//     states are renumbered in their original order (ConvertLattice top-sorts afterwards, which the
//     tools do anyway: no output holds a state id).
struct RawArc {
  int32_t ilabel, olabel, nextstate;
  float graph, acoustic;
};
struct RawLattice {
  int32_t start = CompactLattice::kNoStateId;
  std::vector<std::vector<RawArc>> arcs;
  std::vector<float> final_graph, final_acoustic;
  void Need(int32_t s) {
    while (s >= (int32_t)arcs.size()) {
      arcs.emplace_back();
      final_graph.push_back(kInf);
      final_acoustic.push_back(kInf);
    }
  }
};

static bool ConvertLattice(const RawLattice& in, CompactLattice* clat, std::vector<int32_t>* pool) {
  clat->DeleteStates();
  const int32_t S = (int32_t)in.arcs.size();
  if (S == 0 || in.start == CompactLattice::kNoStateId) return true;
  std::vector<int32_t> arcs_in((size_t)S, 0);
  for (int32_t s = 0; s < S; ++s)
    for (const RawArc& a : in.arcs[(size_t)s]) {
      if (a.nextstate < 0 || a.nextstate >= S) return false;
      ++arcs_in[(size_t)a.nextstate];
    }
  std::vector<char> remove((size_t)S, 0);
  for (int32_t s = 0; s < S; ++s) {
    const bool is_final = !(in.final_graph[(size_t)s] == kInf && in.final_acoustic[(size_t)s] == kInf);
    remove[(size_t)s] = s != in.start && !is_final && arcs_in[(size_t)s] == 1 && in.arcs[(size_t)s].size() == 1 &&
                        in.arcs[(size_t)s][0].olabel == 0;
  }
  std::vector<int32_t> map((size_t)S, -1);
  for (int32_t s = 0; s < S; ++s)
    if (!remove[(size_t)s]) map[(size_t)s] = clat->AddState();
  clat->SetStart(map[(size_t)in.start]);
  std::vector<int32_t> ids;
  for (int32_t s = 0; s < S; ++s) {
    if (remove[(size_t)s]) continue;
    CompactLatticeState& st = clat->State(map[(size_t)s]);
    st.final_graph = in.final_graph[(size_t)s];
    st.final_acoustic = in.final_acoustic[(size_t)s];
    for (const RawArc& first : in.arcs[(size_t)s]) {
      CompactLatticeArc arc;
      arc.ilabel = arc.olabel = first.olabel;
      arc.graph = first.graph;
      arc.acoustic = first.acoustic;
      ids.clear();
      if (first.ilabel != 0) ids.push_back(first.ilabel);
      int32_t next = first.nextstate;
      for (int32_t steps = 0; remove[(size_t)next]; ++steps) {
        if (steps > S) return false;  // a cycle of chain states: not a lattice
        const RawArc& a2 = in.arcs[(size_t)next][0];
        arc.graph = arc.graph + a2.graph;  // Times(weight, arc2.weight), float as LatticeWeightTpl<float>
        arc.acoustic = arc.acoustic + a2.acoustic;
        if (a2.ilabel != 0) ids.push_back(a2.ilabel);
        next = a2.nextstate;
      }
      arc.nextstate = map[(size_t)next];
      arc.string_len = (int32_t)ids.size();
      if (pool && !ids.empty()) {
        arc.string_off = (int32_t)pool->size();
        pool->insert(pool->end(), ids.begin(), ids.end());
      }
      st.arcs.push_back(arc);
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
  RawLattice raw;               // filled instead of clat if the entry turns out to be in Lattice format
  int format = 0;               // 0 unknown, 1 CompactLattice lines seen, 2 Lattice lines seen
  auto need_state = [&](int32_t s) {
    while (s >= clat->NumStates()) clat->AddState();
    raw.Need(s);
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
    if (clat->Start() == CompactLattice::kNoStateId) clat->SetStart(s), raw.start = s;
    if (t.size() <= 2) {  // final state line: "<state> [weight]"
      CompactLatticeState& st = clat->State(s);
      st.final_graph = 0.f;
      st.final_acoustic = 0.f;
      st.final_string_len = 0;
      if (t.size() == 2 && !ParseCompactWeight(t[1], &st.final_graph, &st.final_acoustic, &st.final_string_len, pool, &st.final_string_off)) return false;
      raw.final_graph[(size_t)s] = st.final_graph;
      raw.final_acoustic[(size_t)s] = st.final_acoustic;
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
      // Lattice format: src dst ilabel(transition-id) olabel(word) [g,a]; converted once the whole
      // entry is read (ConvertLattice above)
      if (!IsInt(t[3])) return false;
      if (format == 1) return false;  // an entry is in one format or the other
      format = 2;
      RawArc ra;
      ra.ilabel = (int32_t)std::atol(t[2].c_str());
      ra.olabel = (int32_t)std::atol(t[3].c_str());
      ra.nextstate = arc.nextstate;
      ra.graph = ra.acoustic = 0.f;
      int32_t dummy;
      if (t.size() == 5 && !ParseCompactWeight(t[4], &ra.graph, &ra.acoustic, &dummy)) return false;
      raw.arcs[(size_t)s].push_back(ra);
      continue;
    } else {
      return false;
    }
    if (format == 2) return false;
    format = 1;
    clat->State(s).arcs.push_back(arc);
  }
  if (format == 2) return ConvertLattice(raw, clat, pool);
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
  if (!compact) {  // a Lattice: read it raw, then ConvertLattice (chains merge; see above)
    RawLattice raw;
    if (num_states > 0) raw.Need((int32_t)num_states - 1);
    for (int64_t s = 0; s < num_states; ++s) {
      int32_t len, off;
      if (!read_weight(&raw.final_graph[(size_t)s], &raw.final_acoustic[(size_t)s], &len, &off)) return false;
      int64_t narcs;
      if (!rd(is, &narcs) || narcs < 0) return false;
      raw.arcs[(size_t)s].resize((size_t)narcs);
      for (RawArc& ra : raw.arcs[(size_t)s]) {
        if (!rd(is, &ra.ilabel) || !rd(is, &ra.olabel)) return false;
        if (!read_weight(&ra.graph, &ra.acoustic, &len, &off)) return false;
        if (!rd(is, &ra.nextstate)) return false;
      }
    }
    raw.start = num_states > 0 ? (int32_t)start : CompactLattice::kNoStateId;
    return ConvertLattice(raw, clat, pool);
  }
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
      arc.ilabel = il;
      arc.olabel = ol;
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
