// host_common.h -- host-side declarations shared by the C-ABI translation units.
#ifndef WALT_AMD_HOST_COMMON_H_
#define WALT_AMD_HOST_COMMON_H_

#include <stdint.h>

#include <string>
#include <vector>

#include "core.h"

namespace walt {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

// Literal SEEDPATTERN3 tables (seedpattern.hpp:355-456) and the compare-mask
// table derived from them.
const uint32_t* nocare_row(int seed_i);             // 150 entries
const std::vector<uint32_t>& compare_mask_table();  // [3][39][kMaskWords]

// .dbindex head file (reference.cpp:353-417)
struct IndexHead {
  std::vector<std::string> names;
  std::vector<uint32_t> lengths;
  uint32_t genome_len = 0;
  uint32_t max_index_size = 0;
};
// ReadGenome (reference.cpp:79-129) of a FASTA file or directory: names, lengths and the concatenated
// upper-case sequence with non-ACGT characters replaced by rand()%4 (seeded like walt_makedb)
int read_fasta_genome(const char* fasta_path, std::vector<std::string>& names, std::vector<uint32_t>& lengths,
                      std::vector<uint8_t>& seq);
int read_index_head(const std::string& path, IndexHead& head);
int write_index_head(const std::string& path, const IndexHead& head);

// one strand file (reference.cpp:302-351)
struct StrandFile {
  char strand = '+';
  std::vector<uint8_t> genome;
  std::vector<uint32_t> counter;  // 4^12 + 1
  std::vector<uint32_t> index;
};
int read_strand_file(const std::string& path, uint32_t genome_len, StrandFile& sf);
int write_strand_file(const std::string& path, const StrandFile& sf);

}  // namespace walt
#endif
