// klat-archive-scan -- lists the entries of a lattice archive as the --devices tools see them: the
// archive is mapped and scanned for the entry boundaries without building a lattice
// (archive-index.h), then every entry is parsed from its byte range.  Prints per entry
//   key  begin  end  t|b  states  arcs  s|u      (s: start state 0 and every arc to a higher state; u: TopSort has work to do)
// Not one of the reference's tools: test plumbing for the scanner every rank of a multi-GPU run
// relies on to agree on the entries it does not parse.
#include <cstdio>
#include <cstdlib>

#include "../archive-index.h"
#include "../kaldi-io.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    ParseOptions po("List the entries of a lattice archive (mapped and scanned in place).\nUsage: klat-archive-scan <lattice-rspecifier>\n");
    bool parse = true;
    po.Register("parse", &parse, "Also parse every entry from its byte range and print its states and arcs.");
    po.Read(argc, argv);
    if (po.NumArgs() != 1) {
      po.PrintUsage();
      exit(1);
    }
    if (!MappedArchive::Mappable(po.GetArg(1))) KLAT_ERR("not a plain file: " << po.GetArg(1));
    MappedArchive archive(po.GetArg(1));
    int64_t n = 0;
    while (!archive.Done()) {
      ArchiveEntry e;
      archive.ScanNext(&e);
      long long states = -1, arcs = -1;
      char sorted = '-';
      if (parse) {
        CompactLattice clat;
        archive.Read(e, &clat);
        states = clat.NumStates();
        arcs = clat.NumArcs();
        {
          FlatLattice fl;
          ToFlat(clat, &fl);
          sorted = IsTopSortedFromZero(fl) ? 's' : 'u';
        }
        // a binary entry read straight into arrays must be the object form flattened
        FlatLattice direct, via_object;
        if (archive.ReadFlat(e, &direct)) {
          ToFlat(clat, &via_object);
          if (direct.start != via_object.start || direct.row != via_object.row || direct.dst != via_object.dst ||
              direct.ilabel != via_object.ilabel || direct.olabel != via_object.olabel || direct.len != via_object.len ||
              direct.graph != via_object.graph || direct.acoustic != via_object.acoustic || direct.final_graph != via_object.final_graph ||
              direct.final_acoustic != via_object.final_acoustic || direct.final_len != via_object.final_len)
            KLAT_ERR("entry " << e.key << ": the flat reader and the object reader disagree");
          if (IsTopSortedFromZero(direct)) {
            std::vector<int32_t> t;
            (void)FlatLatticeStateTimes(direct, &t);
          }
        } else if (e.binary) {
          KLAT_ERR("entry " << e.key << ": not a binary CompactLattice");
        }
      }
      std::printf("%s %zu %zu %c %lld %lld %c\n", e.key.c_str(), e.begin, e.end, e.binary ? 'b' : 't', states, arcs, sorted);
      ++n;
    }
    KLAT_LOG("Scanned " << n << " entries of " << archive.size() << " bytes.");
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
