// Native FLAC decoder for the real data path (reference data_modules/WebAudioDataModule.py:104-109,
// `.decode(wds.torch_audio)` -> torchaudio.load of the shard's .flac members).  Host code (no GPU): built with g++ into
// wavjepa_amd/lib/libwavjepa_io.so and bound through ctypes (wavjepa_amd/audio_io.py); loader threads call it without the GIL.
//
// Implements the FLAC format specification (xiph.org/flac/format.html; RFC 9639) from the format description, not from libFLAC:
// STREAMINFO, all subframe types (CONSTANT, VERBATIM, FIXED 0-4, LPC 1-32), wasted bits, Rice / Rice2 partitions with escape,
// independent / left-side / right-side / mid-side channel assignments, fixed and variable block size, 4-32 bits per sample,
// frame header CRC-8 and frame CRC-16 (both verified).  Output: interleaved int32 PCM; the MD5 of STREAMINFO is returned to the
// caller, who can verify it over the decoded samples (audio_io.decode_flac(..., verify_md5=True)).
//
//   int wj_flac_info(const uint8_t* buf, int64_t len, wj_flac_stream_info* out)           0 or a negative error
//   int64_t wj_flac_decode(const uint8_t* buf, int64_t len, int32_t* pcm, int64_t capacity_frames)
//        -> number of inter-channel frames written (pcm[frame * channels + ch]) or a negative error; with pcm == NULL nothing is
//           written and the number of frames the stream decodes to is returned (sizing pass for streams of unknown length)
#include <stdint.h>
#include <string.h>

#include <vector>

extern "C" {
typedef struct {
    int32_t sample_rate, channels, bits_per_sample, min_block, max_block;
    int64_t total_samples;      // per channel; 0 = unknown
    uint8_t md5[16];
} wj_flac_stream_info;
}

namespace {

enum { ERR_FORMAT = -1, ERR_TRUNCATED = -2, ERR_CRC = -3, ERR_UNSUPPORTED = -4, ERR_CAPACITY = -5 };

// Predictor arithmetic wraps (two's complement) instead of overflowing: a valid stream never comes near 64 bits, a corrupt one
// (caught by its CRC-16 a few lines later) must not be undefined behaviour on the way (found by the sanitizer fuzz test).
inline int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
inline int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
inline int64_t wmul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

struct BitReader {
    const uint8_t* p;
    int64_t n, pos;        // pos in bits
    bool bad;
    BitReader(const uint8_t* b, int64_t len) : p(b), n(len), pos(0), bad(false) {}
    // the next bits of the stream, left-aligned in 64 bits (at least 57 of them valid), zeros beyond the end
    inline uint64_t window() const {
        const int64_t byte = pos >> 3;
        uint64_t w = 0;
        if (byte + 8 <= n) {
            memcpy(&w, p + byte, 8);
            w = __builtin_bswap64(w);
        } else {
            for (int i = 0; i < 8; ++i) w = (w << 8) | (byte + i < n ? p[byte + i] : 0);
        }
        return w << (pos & 7);
    }
    inline void advance(int k) {
        pos += k;
        if (pos > n * 8) bad = true;
    }
    inline uint32_t bit() { return (uint32_t)bits(1); }
    inline uint64_t bits(int k) {          // k <= 57
        if (k == 0) return 0;
        const uint64_t v = window() >> (64 - k);
        advance(k);
        return bad ? 0 : v;
    }
    inline int64_t sbits(int k) {
        if (k == 0) return 0;
        const uint64_t v = bits(k);
        const uint64_t sign = 1ull << (k - 1);
        return (int64_t)((v ^ sign) - sign);
    }
    inline uint32_t unary() {              // zeros before the terminating one
        uint32_t q = 0;
        while (!bad) {
            const uint64_t w = window() >> 7;              // 57 valid bits in the low end
            if (w) {
                const int lz = __builtin_clzll(w) - 7;
                advance(lz + 1);
                return q + (uint32_t)lz;
            }
            q += 57;
            advance(57);
        }
        return q;
    }
    inline void align() { pos = (pos + 7) & ~7ll; }
};

uint8_t crc8(const uint8_t* d, int64_t n) {
    uint8_t c = 0;
    for (int64_t i = 0; i < n; ++i) {
        c ^= d[i];
        for (int b = 0; b < 8; ++b) c = (c & 0x80) ? (uint8_t)((c << 1) ^ 0x07) : (uint8_t)(c << 1);
    }
    return c;
}

// CRC-16 table (polynomial 0x8005), built at compile time: loader threads call the decoder concurrently without the GIL, and a
// lazily initialised static table behind an unsynchronised flag was a data race.
struct Crc16Table {
    uint16_t v[256];
    constexpr Crc16Table() : v() {
        for (int i = 0; i < 256; ++i) {
            uint16_t c = (uint16_t)(i << 8);
            for (int b = 0; b < 8; ++b) c = (c & 0x8000) ? (uint16_t)((c << 1) ^ 0x8005) : (uint16_t)(c << 1);
            v[i] = c;
        }
    }
};
constexpr Crc16Table CRC16_TABLE;

uint16_t crc16(const uint8_t* d, int64_t n) {
    uint16_t c = 0;
    for (int64_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ CRC16_TABLE.v[((c >> 8) ^ d[i]) & 0xff]);
    return c;
}

int parse_streaminfo(const uint8_t* buf, int64_t len, wj_flac_stream_info* si, int64_t* audio_start) {
    int64_t off = 0;
    if (len >= 10 && !memcmp(buf, "ID3", 3)) {     // an ID3v2 tag in front of the stream
        const int64_t sz = ((buf[6] & 0x7f) << 21) | ((buf[7] & 0x7f) << 14) | ((buf[8] & 0x7f) << 7) | (buf[9] & 0x7f);
        off = 10 + sz;
    }
    if (len < off + 4 + 4 + 34 || memcmp(buf + off, "fLaC", 4)) return ERR_FORMAT;
    off += 4;
    bool have = false, last = false;
    while (!last) {
        if (off + 4 > len) return ERR_TRUNCATED;
        last = (buf[off] & 0x80) != 0;
        const int type = buf[off] & 0x7f;
        const int64_t blen = ((int64_t)buf[off + 1] << 16) | ((int64_t)buf[off + 2] << 8) | buf[off + 3];
        off += 4;
        if (off + blen > len) return ERR_TRUNCATED;
        if (type == 0) {
            if (blen < 34) return ERR_FORMAT;
            const uint8_t* s = buf + off;
            si->min_block = (s[0] << 8) | s[1];
            si->max_block = (s[2] << 8) | s[3];
            si->sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            si->channels = ((s[12] >> 1) & 7) + 1;
            si->bits_per_sample = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si->total_samples = ((int64_t)(s[13] & 0xf) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
            memcpy(si->md5, s + 18, 16);
            have = true;
        }
        off += blen;
    }
    if (!have || si->sample_rate <= 0 || si->bits_per_sample < 4) return ERR_FORMAT;
    *audio_start = off;
    return 0;
}

int read_residual(BitReader& br, int64_t* out, int blocksize, int order) {
    const int method = (int)br.bits(2);
    if (method > 1) return ERR_UNSUPPORTED;
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = (int)br.bits(4);
    const int parts = 1 << porder;
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return ERR_FORMAT;
    int idx = order;
    for (int pt = 0; pt < parts; ++pt) {
        int count = blocksize >> porder;
        if (pt == 0) count -= order;
        if (count < 0) return ERR_FORMAT;
        const int k = (int)br.bits(pbits);
        if (k == esc) {
            const int nb = (int)br.bits(5);
            for (int i = 0; i < count; ++i) out[idx++] = br.sbits(nb);
        } else {
            for (int i = 0; i < count; ++i) {
                const uint64_t q = br.unary();
                const uint64_t u = (q << k) | (k ? br.bits(k) : 0);
                out[idx++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
            }
        }
        if (br.bad) return ERR_TRUNCATED;
    }
    return 0;
}

int read_subframe(BitReader& br, int64_t* s, int blocksize, int bps) {
    if (br.bit() != 0) return ERR_FORMAT;
    const int type = (int)br.bits(6);
    int wasted = 0;
    if (br.bit()) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps <= 0) return ERR_FORMAT;
    if (type == 0) {                                    // CONSTANT
        const int64_t v = br.sbits(bps);
        for (int i = 0; i < blocksize; ++i) s[i] = v;
    } else if (type == 1) {                             // VERBATIM
        for (int i = 0; i < blocksize; ++i) s[i] = br.sbits(bps);
    } else if (type >= 8 && type <= 12) {               // FIXED, order type - 8
        const int order = type - 8;
        if (order > blocksize) return ERR_FORMAT;
        for (int i = 0; i < order; ++i) s[i] = br.sbits(bps);
        const int rc = read_residual(br, s, blocksize, order);
        if (rc) return rc;
        for (int i = order; i < blocksize; ++i) {
            switch (order) {
                case 0: break;
                case 1: s[i] = wadd(s[i], s[i - 1]); break;
                case 2: s[i] = wadd(s[i], wsub(wmul(2, s[i - 1]), s[i - 2])); break;
                case 3: s[i] = wadd(s[i], wadd(wsub(wmul(3, s[i - 1]), wmul(3, s[i - 2])), s[i - 3])); break;
                default: s[i] = wadd(s[i], wsub(wadd(wsub(wmul(4, s[i - 1]), wmul(6, s[i - 2])), wmul(4, s[i - 3])), s[i - 4])); break;
            }
        }
    } else if (type >= 32) {                            // LPC, order (type & 31) + 1
        const int order = (type & 31) + 1;
        if (order > blocksize) return ERR_FORMAT;
        for (int i = 0; i < order; ++i) s[i] = br.sbits(bps);
        const int prec = (int)br.bits(4) + 1;
        if (prec == 16) return ERR_FORMAT;
        const int shift = (int)br.sbits(5);
        if (shift < 0) return ERR_UNSUPPORTED;
        int64_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.sbits(prec);
        const int rc = read_residual(br, s, blocksize, order);
        if (rc) return rc;
        for (int i = order; i < blocksize; ++i) {
            int64_t acc = 0;
            for (int j = 0; j < order; ++j) acc = wadd(acc, wmul(coef[j], s[i - 1 - j]));
            s[i] = wadd(s[i], acc >> shift);
        }
    } else {
        return ERR_UNSUPPORTED;                         // reserved subframe types
    }
    if (br.bad) return ERR_TRUNCATED;
    if (wasted)
        for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
    return 0;
}

}  // namespace

extern "C" int wj_flac_info(const uint8_t* buf, int64_t len, wj_flac_stream_info* out) {
    if (!buf || !out || len <= 0) return ERR_FORMAT;
    int64_t start = 0;
    return parse_streaminfo(buf, len, out, &start);
}

extern "C" int64_t wj_flac_decode(const uint8_t* buf, int64_t len, int32_t* pcm, int64_t capacity_frames) {
    if (!buf || len <= 0) return ERR_FORMAT;
    const bool count_only = pcm == nullptr;
    wj_flac_stream_info si;
    int64_t off = 0;
    int rc = parse_streaminfo(buf, len, &si, &off);
    if (rc) return rc;
    static const int BS_TABLE[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
    static const int BPS_TABLE[8] = {0, 8, 12, 0, 16, 20, 24, 32};
    std::vector<int64_t> work;
    int64_t written = 0;
    while (off + 2 <= len) {
        if (!(buf[off] == 0xff && (buf[off + 1] & 0xfe) == 0xf8)) {      // 14-bit sync + reserved 0 (+ blocking strategy bit)
            if (si.total_samples && written >= si.total_samples) break;   // trailing bytes (e.g. an ID3v1 tag)
            return ERR_FORMAT;
        }
        BitReader br(buf + off, len - off);
        br.bits(15);
        br.bit();                                                         // blocking strategy: only the coded number differs
        const int bs_code = (int)br.bits(4), sr_code = (int)br.bits(4), ch_code = (int)br.bits(4), bps_code = (int)br.bits(3);
        if (br.bit() != 0) return ERR_FORMAT;
        // UTF-8-style coded frame / sample number (value itself is not needed)
        {
            const uint32_t first = (uint32_t)br.bits(8);
            int extra = 0;
            if (first >= 0xfe) extra = 6; else if (first >= 0xfc) extra = 5; else if (first >= 0xf8) extra = 4;
            else if (first >= 0xf0) extra = 3; else if (first >= 0xe0) extra = 2; else if (first >= 0xc0) extra = 1;
            else if (first >= 0x80) return ERR_FORMAT;
            for (int i = 0; i < extra; ++i)
                if ((br.bits(8) & 0xc0) != 0x80) return ERR_FORMAT;
        }
        int blocksize = BS_TABLE[bs_code];
        if (bs_code == 0) return ERR_FORMAT;
        if (bs_code == 6) blocksize = (int)br.bits(8) + 1;
        else if (bs_code == 7) blocksize = (int)br.bits(16) + 1;
        if (sr_code == 12) br.bits(8); else if (sr_code == 13 || sr_code == 14) br.bits(16); else if (sr_code == 15) return ERR_FORMAT;
        if (br.bad) return ERR_TRUNCATED;
        const int64_t hdr_bytes = br.pos >> 3;
        if (off + hdr_bytes + 1 > len) return ERR_TRUNCATED;
        if (crc8(buf + off, hdr_bytes) != buf[off + hdr_bytes]) return ERR_CRC;
        br.bits(8);
        const int bps = bps_code == 0 ? si.bits_per_sample : BPS_TABLE[bps_code];
        if (bps == 0) return ERR_FORMAT;
        int channels;
        if (ch_code < 8) channels = ch_code + 1; else if (ch_code <= 10) channels = 2; else return ERR_FORMAT;
        if (channels != si.channels) return ERR_UNSUPPORTED;
        if (!count_only && written + blocksize > capacity_frames) return ERR_CAPACITY;
        work.resize((size_t)channels * blocksize);
        for (int ch = 0; ch < channels; ++ch) {
            const bool side = (ch_code == 8 && ch == 1) || (ch_code == 9 && ch == 0) || (ch_code == 10 && ch == 1);
            rc = read_subframe(br, work.data() + (size_t)ch * blocksize, blocksize, bps + (side ? 1 : 0));
            if (rc) return rc;
        }
        br.align();
        const int64_t body = br.pos >> 3;
        if (off + body + 2 > len) return ERR_TRUNCATED;
        const uint16_t want = (uint16_t)((buf[off + body] << 8) | buf[off + body + 1]);
        if (crc16(buf + off, body) != want) return ERR_CRC;
        int64_t* c0 = work.data();
        int64_t* c1 = work.data() + blocksize;
        if (ch_code == 8) {
            for (int i = 0; i < blocksize; ++i) c1[i] = wsub(c0[i], c1[i]);
        } else if (ch_code == 9) {
            for (int i = 0; i < blocksize; ++i) c0[i] = wadd(c0[i], c1[i]);
        } else if (ch_code == 10) {
            for (int i = 0; i < blocksize; ++i) {
                const int64_t side = c1[i];
                const int64_t mid = (int64_t)(((uint64_t)c0[i] << 1) | (uint64_t)(side & 1));
                c0[i] = wadd(mid, side) >> 1;
                c1[i] = wsub(mid, side) >> 1;
            }
        }
        if (!count_only) {
            int32_t* dst = pcm + written * channels;
            for (int i = 0; i < blocksize; ++i)
                for (int ch = 0; ch < channels; ++ch) dst[(int64_t)i * channels + ch] = (int32_t)work[(size_t)ch * blocksize + i];
        }
        written += blocksize;
        off += body + 2;
    }
    if (si.total_samples && written < si.total_samples) return ERR_TRUNCATED;
    return written;
}
