// klat-lattice-copy -- copies a CompactLattice archive between text and binary form
// (ark,t: <-> ark:).  Helper for exercising the binary reader of SURVEY.md Appendix B.4; the
// transition-id strings are reduced to their lengths (all ids written as 1).
#include <cstdlib>

#include "../compact-lattice.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    ParseOptions po("Copy a CompactLattice archive.\nUsage: klat-lattice-copy <lattice-rspecifier> <lattice-wspecifier>\n");
    po.Read(argc, argv);
    if (po.NumArgs() != 2) {
      po.PrintUsage();
      exit(1);
    }
    const Specifier ws = ParseSpecifier(po.GetArg(2), true);
    Output out(ws.filename);
    int64_t n = 0;
    for (SequentialCompactLatticeReader reader(po.GetArg(1)); !reader.Done(); reader.Next(), ++n) {
      out.Stream() << reader.Key() << ' ';
      if (ws.text) {
        out.Stream().precision(9);
        WriteCompactLatticeText(out.Stream(), reader.Value());
      } else {
        WriteCompactLatticeBinary(out.Stream(), reader.Value());
      }
    }
    out.Close();
    KLAT_LOG("Copied " << n << " lattices.");
    return 0;
  } catch (const std::exception& e) {
    std::cerr << e.what();
    return -1;
  }
}
