// makedb_main.cpp -- `makedb -c <fasta|dir> -o <out.dbindex> [-t threads] [-g device]`
// (reference makedb.cpp:87-168), on top of walt_makedb(); -g builds on the GPU (walt_makedb_device).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../../include/walt_amd.h"

int main(int argc, const char** argv) {
  std::string chrom, out;
  int threads = 1, device = -1;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if ((a == "-c" || a == "-chrom") && i + 1 < argc) chrom = argv[++i];
    else if ((a == "-o" || a == "-output") && i + 1 < argc) out = argv[++i];
    else if ((a == "-t" || a == "-thread") && i + 1 < argc) threads = atoi(argv[++i]);
    else if ((a == "-g" || a == "-gpu") && i + 1 < argc) device = atoi(argv[++i]);  // extension: build on this GPU
    else {
      fprintf(stderr, "Usage: makedb -c <chromosomes .fa file or dir> -o <output .dbindex> [-t threads] [-g gpu]\n"
                      "  -g <gpu>  build on this GPU (seconds instead of minutes).  Entries whose 60 care characters are all equal come out\n"
                      "            in ascending position, not in the order the reference's std::sort leaves them: the files map the same\n"
                      "            reads with the same counts, but the position reported for an ambiguous read (last one wins) can differ\n"
                      "            from the reference's; without -g the files are byte-identical to the reference makedb's.\n");
      return EXIT_SUCCESS;
    }
  }
  if (chrom.empty() || out.empty()) { fprintf(stderr, "Usage: makedb -c <chromosomes .fa file or dir> -o <output .dbindex>\n"); return EXIT_SUCCESS; }
  if (out.substr(out.find_last_of(".") + 1) != "dbindex") {  // makedb.cpp:120-123
    fprintf(stderr, "The suffix of the output file should be '.dbindex'\n");
    return EXIT_FAILURE;
  }
  const int rc = device >= 0 ? walt_makedb_device(chrom.c_str(), out.c_str(), device)
                             : walt_makedb(chrom.c_str(), out.c_str(), threads);
  if (rc != WALT_OK) {
    fprintf(stderr, "%s\n", walt_last_error());
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
