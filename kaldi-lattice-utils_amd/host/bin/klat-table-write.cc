// klat-table-write -- writes a table given as text on standard input through this build's OUTPUT HOLDERS
// (host/table-writers.h), text or binary.  Not one of the reference's tools: test plumbing that lets the tests hold
// the writers' bytes against fixtures assembled by hand from the reference's holder
// (util/basic-tuple-vector-holder.h:141-181) without a GPU in the loop.
//   kind = segment    lines "key label t0 t1 logp ; label t0 t1 logp ..."         (kwsbin2/lattice-word-index-segment.cc:77-78)
//          position   lines "key label pos t0 t1 logp ; ..."                       (kwsbin2/lattice-word-index-position.cc:80-82)
//          utterance  lines "key label logp ; ..."                                 (kwsbin2/lattice-word-index-utterance.cc:267)
//          posterior  lines "key [ label logp label logp ] [ ] ..."                (latbin/lattice-to-word-frame-post.cc:66,139)
#include <cstdlib>
#include <iostream>
#include <sstream>

#include "../table-writers.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    ParseOptions po("Write a text table from standard input through the output holders.\n"
                    "Usage: klat-table-write <segment|position|utterance|posterior> <wspecifier>\n");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) {
      po.PrintUsage();
      exit(1);
    }
    const std::string kind = po.GetArg(1);
    SegmentScoreWriter seg(kind == "segment" ? po.GetArg(2) : "ark:/dev/null");
    PositionScoreWriter pos(kind == "position" ? po.GetArg(2) : "ark:/dev/null");
    UtteranceScoreWriter utt(kind == "utterance" ? po.GetArg(2) : "ark:/dev/null");
    PosteriorWriter post(kind == "posterior" ? po.GetArg(2) : "ark:/dev/null");
    if (kind != "segment" && kind != "position" && kind != "utterance" && kind != "posterior") KLAT_ERR("unknown table kind " << kind);
    std::string line;
    int64_t n = 0;
    while (std::getline(std::cin, line)) {
      std::istringstream is(line);
      std::string key, tok;
      if (!(is >> key)) continue;
      std::vector<std::string> toks;
      while (is >> tok) toks.push_back(tok);
      if (kind == "posterior") {
        PosteriorWriter::T frames;
        for (size_t i = 0; i < toks.size(); ++i) {
          if (toks[i] != "[") KLAT_ERR("posterior line: expected [ in entry " << key);
          frames.emplace_back();
          for (++i; i < toks.size() && toks[i] != "]"; i += 2) {
            if (i + 1 >= toks.size()) KLAT_ERR("posterior line: odd number of fields in entry " << key);
            frames.back().emplace_back((int32_t)std::stol(toks[i]), std::stof(toks[i + 1]));
          }
        }
        post.Write(key, frames);
      } else {
        const size_t width = kind == "segment" ? 4 : kind == "position" ? 5 : 2;
        std::vector<std::vector<std::string>> recs(1);
        for (const std::string& t : toks) {
          if (t == ";") recs.emplace_back();
          else recs.back().push_back(t);
        }
        if (recs.back().empty()) recs.pop_back();
        for (const auto& r : recs)
          if (r.size() != width) KLAT_ERR("a " << kind << " record has " << width << " fields (entry " << key << ")");
        if (kind == "segment") {
          std::vector<klat_segment_record> v;
          for (const auto& r : recs) v.push_back(klat_segment_record{(int32_t)std::stol(r[0]), (int32_t)std::stol(r[1]), (int32_t)std::stol(r[2]), 0, std::stod(r[3])});
          seg.Write(key, v.data(), (int64_t)v.size());
        } else if (kind == "position") {
          std::vector<klat_position_record> v;
          for (const auto& r : recs)
            v.push_back(klat_position_record{(int32_t)std::stol(r[0]), (int32_t)std::stol(r[1]), (int32_t)std::stol(r[2]), (int32_t)std::stol(r[3]), std::stod(r[4])});
          pos.Write(key, v.data(), (int64_t)v.size());
        } else {
          std::vector<std::pair<int32_t, double>> v;
          for (const auto& r : recs) v.emplace_back((int32_t)std::stol(r[0]), std::stod(r[1]));
          utt.Write(key, v);
        }
      }
      ++n;
    }
    seg.Close(), pos.Close(), utt.Close(), post.Close();
    KLAT_LOG("Wrote " << n << " entries.");
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
