// makedb_main.cpp -- `makedb -c <fasta|dir> -o <out.dbindex> [-t threads]`
// (reference makedb.cpp:87-168), on top of walt_makedb().
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../../include/walt_amd.h"

int main(int argc, const char** argv) {
  std::string chrom, out;
  int threads = 1;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if ((a == "-c" || a == "-chrom") && i + 1 < argc) chrom = argv[++i];
    else if ((a == "-o" || a == "-output") && i + 1 < argc) out = argv[++i];
    else if ((a == "-t" || a == "-thread") && i + 1 < argc) threads = atoi(argv[++i]);
    else { fprintf(stderr, "Usage: makedb -c <chromosomes .fa file or dir> -o <output .dbindex> [-t threads]\n"); return EXIT_SUCCESS; }
  }
  if (chrom.empty() || out.empty()) { fprintf(stderr, "Usage: makedb -c <chromosomes .fa file or dir> -o <output .dbindex>\n"); return EXIT_SUCCESS; }
  if (out.substr(out.find_last_of(".") + 1) != "dbindex") {  // makedb.cpp:120-123
    fprintf(stderr, "The suffix of the output file should be '.dbindex'\n");
    return EXIT_FAILURE;
  }
  if (walt_makedb(chrom.c_str(), out.c_str(), threads) != WALT_OK) {
    fprintf(stderr, "%s\n", walt_last_error());
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
