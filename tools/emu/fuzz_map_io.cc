// Mutation fuzz of the saved-map loader (csrc/ovs_mapio.hip: host code that parses a file it did not write)
// under AddressSanitizer and UndefinedBehaviorSanitizer, built against the stand-in HIP runtime:
//
//   tools/emu/fuzz_map_io.sh [iterations]          (builds /tmp/ovs_emu/fuzz_map_io and runs it on a seed file)
//
// Every iteration copies the seed, applies a few mutations (byte flips, bytes set to MessagePack heads that
// announce long strings / arrays / maps, truncation, duplicated spans) and opens the result; a file the loader
// accepts is read back completely.  The sanitizers end the run at the first invalid access.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/ovs.h"

static uint64_t g_rng = 88172645463325252ull;
static uint64_t rnd() {
    g_rng ^= g_rng << 13;
    g_rng ^= g_rng >> 7;
    g_rng ^= g_rng << 17;
    return g_rng;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: fuzz_map_io <seed file> <scratch file> <iterations>\n");
        return 2;
    }
    std::vector<uint8_t> seed;
    {
        std::FILE* f = std::fopen(argv[1], "rb");
        if (!f) {
            return 2;
        }
        uint8_t buf[4096];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) {
            seed.insert(seed.end(), buf, buf + got);
        }
        std::fclose(f);
    }
    const long iterations = std::atol(argv[3]);
    static const uint8_t heads[] = {0xdb, 0xdd, 0xdf, 0xc6, 0xc9, 0xda, 0xdc, 0xde, 0xc1, 0xcf, 0xd3, 0x9f, 0x8f, 0xbf, 0xff, 0x00};
    long accepted = 0, rows_read = 0;
    for (long it = 0; it < iterations; ++it) {
        std::vector<uint8_t> b = seed;
        const int edits = 1 + (int)(rnd() % 4);
        for (int e = 0; e < edits && !b.empty(); ++e) {
            const size_t at = rnd() % b.size();
            switch (rnd() % 6) {
                case 0: b[at] ^= (uint8_t)(1u << (rnd() % 8)); break;
                case 1: b[at] = heads[rnd() % sizeof(heads)]; break;
                case 2: b.resize(at); break;
                case 3: b[at] = (uint8_t)rnd(); break;
                case 4: { // a length field of all ones behind a 32-bit head
                    b[at] = heads[rnd() % 5];
                    for (size_t k = 1; k <= 4 && at + k < b.size(); ++k) {
                        b[at + k] = (rnd() & 1) ? 0xff : 0x00;
                    }
                    break;
                }
                default: { // duplicate a span
                    const size_t len = rnd() % 64;
                    if (at + len <= b.size()) {
                        b.insert(b.begin() + (long)at, b.begin() + (long)at, b.begin() + (long)(at + len));
                    }
                    break;
                }
            }
        }
        std::FILE* f = std::fopen(argv[2], "wb");
        if (!f) {
            return 2;
        }
        if (!b.empty()) {
            std::fwrite(b.data(), 1, b.size(), f);
        }
        std::fclose(f);
        ovs_map_file* m = nullptr;
        if (ovs_map_open(argv[2], &m) != OVS_OK) {
            continue;
        }
        ++accepted;
        const int nk = ovs_map_num_keyframes(m);
        const int64_t nd = ovs_map_num_descriptors(m);
        std::vector<int32_t> ids((size_t)nk + 1);
        std::vector<int64_t> offs((size_t)nk + 1);
        std::vector<uint8_t> descs((size_t)nd * 32 + 1);
        if (ovs_map_keyframe_rows(m, ids.data(), offs.data()) != OVS_OK || offs[(size_t)nk] != nd || ovs_map_copy_descriptors(m, descs.data()) != OVS_OK) {
            std::fprintf(stderr, "inconsistent map at iteration %ld\n", it);
            return 1;
        }
        rows_read += nd;
        ovs_map_close(m);
    }
    std::printf("%ld files, %ld accepted, %ld descriptor rows read back\nOK\n", iterations, accepted, rows_read);
    return 0;
}
