// klat-lattice-copy -- copies a CompactLattice archive between text and binary form
// (ark,t: <-> ark:).  Helper for exercising the binary reader of SURVEY.md Appendix B.4; the
// transition-id strings are reduced to their lengths (all ids written as 1).  --top-sort=true runs
// TopSortCompactLatticeIfNeeded (latbin/lattice-to-word-frame-post.cc:80) on the way: the CPU-side tests'
// handle on the renumbering and on the cyclic-input error.
#include <cstdlib>

#include "../compact-lattice.h"
#include "../lattice-functions.h"

int main(int argc, char* argv[]) {
  try {
    using namespace klat;
    ParseOptions po("Copy a CompactLattice archive.\nUsage: klat-lattice-copy <lattice-rspecifier> <lattice-wspecifier>\n");
    bool top_sort = false;
    po.Register("top-sort", &top_sort, "Run TopSortCompactLatticeIfNeeded on every lattice before it is written (a cyclic one ends the run).");
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
      CompactLattice sorted;
      if (top_sort) {
        sorted = reader.Value();
        if (sorted.Start() != CompactLattice::kNoStateId) TopSortCompactLatticeIfNeeded(&sorted);
      }
      const CompactLattice& clat = top_sort ? sorted : reader.Value();
      if (ws.text) {
        out.Stream().precision(9);
        WriteCompactLatticeText(out.Stream(), clat);
      } else {
        WriteCompactLatticeBinary(out.Stream(), clat);
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
