// Sanitizer fuzz driver for csrc/flac_decode.cpp (compiled together with it under -fsanitize=address,undefined by
// tests/test_data_path_cpu.py): every corpus file is decoded as is (must succeed and give the recorded sample count), then through
// N random mutations -- bit flips, byte overwrites, truncations, inserted garbage, duplicated spans.  The decoder may return any
// error code; what it may not do is read or write out of bounds, overflow a signed integer, shift out of range or ask for an
// absurd allocation: any sanitizer report aborts the process (-fno-sanitize-recover) and fails the test.
//   usage: flac_fuzz <mutations per file> <seed> <file> [<file> ...]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

extern "C" {
typedef struct {
    int32_t sample_rate, channels, bits_per_sample, min_block, max_block;
    int64_t total_samples;
    uint8_t md5[16];
} wj_flac_stream_info;
int wj_flac_info(const uint8_t* buf, int64_t len, wj_flac_stream_info* out);
int64_t wj_flac_decode(const uint8_t* buf, int64_t len, int32_t* pcm, int64_t capacity_frames);
}

static uint64_t rng_state = 1;
static uint64_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}

static int64_t run(const std::vector<uint8_t>& b) {
    wj_flac_stream_info si;
    // exact-size heap copy: an over-read of even one byte is an ASan report
    uint8_t* copy = (uint8_t*)malloc(b.size() ? b.size() : 1);
    memcpy(copy, b.data(), b.size());
    int64_t n = -1;
    if (wj_flac_info(copy, (int64_t)b.size(), &si) == 0 && si.channels >= 1 && si.channels <= 8) {
        const int64_t count = wj_flac_decode(copy, (int64_t)b.size(), nullptr, 0);
        if (count >= 0 && count < (1 << 22)) {
            int32_t* pcm = (int32_t*)malloc((size_t)(count > 0 ? count : 1) * si.channels * 4);
            n = wj_flac_decode(copy, (int64_t)b.size(), pcm, count);
            if (n != count) { fprintf(stderr, "counting pass %lld != decode %lld\n", (long long)count, (long long)n); abort(); }
            free(pcm);
        }
    }
    free(copy);
    return n;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int per_file = atoi(argv[1]);
    rng_state = strtoull(argv[2], nullptr, 10) | 1;
    long ok = 0, rejected = 0;
    for (int f = 3; f < argc; ++f) {
        FILE* fp = fopen(argv[f], "rb");
        if (!fp) return 3;
        std::vector<uint8_t> good(1 << 22);
        good.resize(fread(good.data(), 1, good.size(), fp));
        fclose(fp);
        if (run(good) < 0) { fprintf(stderr, "%s: the unmodified file does not decode\n", argv[f]); return 4; }
        for (int m = 0; m < per_file; ++m) {
            std::vector<uint8_t> b = good;
            const int kind = (int)(rnd() % 6);
            const size_t pos = (size_t)(rnd() % b.size());
            if (kind == 0) { for (int k = 0, K = 1 + (int)(rnd() % 4); k < K; ++k) b[(size_t)(rnd() % b.size())] ^= (uint8_t)(1u << (rnd() % 8)); }
            else if (kind == 1) b[pos] = (uint8_t)rnd();
            else if (kind == 2) b.resize(pos);
            else if (kind == 3) { std::vector<uint8_t> junk(1 + rnd() % 64); for (auto& x : junk) x = (uint8_t)rnd(); b.insert(b.begin() + pos, junk.begin(), junk.end()); }
            else if (kind == 4) { const size_t len = 1 + (size_t)(rnd() % 200); if (pos + len < b.size()) b.insert(b.begin() + pos, good.begin() + pos, good.begin() + pos + len); }
            else { const size_t hdr = b.size() < 64 ? b.size() : 64; b[(size_t)(rnd() % hdr)] = (uint8_t)rnd(); }      // headers / STREAMINFO
            if (b.empty()) continue;
            if (run(b) >= 0) ++ok; else ++rejected;
        }
    }
    printf("mutations decoded: %ld, rejected: %ld\n", ok, rejected);
    return 0;
}
