// split_naive -- the reference's comparator tool (split_naive.cpp): usage and exit codes as there.
#include "../../include/raft_host.h"

#include <cstdlib>
#include <iostream>
#include <string>

int main(int argc, char **argv)
{
    if (argc < 4) {                                // split_naive.cpp:46-52,57-58
        std::cout << "Purpose: Split input reads naively into non-overlapping subreads. The output format is FASTA\n";
        std::cout << "Usage: split_naive <inputfilename> <outputfilename> SPLITLEN\n";
        std::cout << "Example: split_naive input.fastq output.fragmented.fasta 20000\n";
        return 1;
    }
    int len = 0;
    try { len = std::stoi(argv[3]); } catch (...) { len = 0; }
    const int rc = raft_host_split_naive(argv[1], argv[2], len, nullptr);
    if (rc != RAFT_HOST_OK) {
        std::cout << "ERROR, split_naive, " << (rc == RAFT_HOST_ERR_ARG ? "SPLITLEN must be a positive integer" : "cannot read or write the given files") << "\n";
        return 1;
    }
    return 0;
}
