#!/usr/bin/env python3
"""Writes tests/golden/flac_rfc9639.npz: the three complete example FLAC streams of RFC 9639 (Appendix D: "Examples"), byte for
byte as the RFC prints them, with the samples they decode to.

Why these are independent vectors: the streams were produced by the reference encoder (example 2 carries its vendor string,
"reference libFLAC 1.3.3 20190804"), and each carries THREE redundancies computed by that encoder -- the frame-header CRC-8, the
frame CRC-16 and, in STREAMINFO, the MD5 of the decoded PCM.  This script (no decoder of this repository involved) checks the
CRC-8 and CRC-16 of every frame with its own bitwise implementation, so a transcription error in the bytes cannot pass; the
test then requires the repository's decoder to reproduce samples whose MD5 equals the one in the stream.

Coverage: D.1 two VERBATIM subframes with wasted bits, independent stereo, 1-sample block; D.2 SEEKTABLE + VORBIS_COMMENT + PADDING
metadata, a right/side frame with FIXED order-1 predictors and partitioned Rice residuals, then a 3-sample last block (VERBATIM,
wasted bits); D.3 an 8-bit 32 kHz mono LPC subframe (order 3, Rice).
"""
import hashlib
import os

import numpy as np

EX1 = """
66 4c 61 43 80 00 00 22 10 00 10 00 00 00 0f 00 00 0f 0a c4 42 f0 00 00 00 01 3e 84 b4 18 07 dc 69 03 07 58
6a 3d ad 1a 2e 0f ff f8 69 18 00 00 bf 03 58 fd 03 12 8b aa 9a
"""
EX2 = """
66 4c 61 43 00 00 00 22 00 10 00 10 00 00 17 00 00 44 0a c4 42 f0 00 00 00 13 d5 b0 56 49 75 e9 8b 8d 8b 93
04 22 75 7b 81 03 03 00 00 12 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 00 10 04 00 00 3a 20 00 00 00
72 65 66 65 72 65 6e 63 65 20 6c 69 62 46 4c 41 43 20 31 2e 33 2e 33 20 32 30 31 39 30 38 30 34 01 00 00 00
0e 00 00 00 54 49 54 4c 45 3d d7 a9 d7 9c d7 95 d7 9d 81 00 00 06 00 00 00 00 00 00 ff f8 69 98 00 0f 99 12
08 67 01 62 3d 14 42 99 8f 5d f7 0d 6f e0 0c 17 ca eb 21 00 0e e7 a7 7a 24 a1 59 0c 12 17 b6 03 09 7b 78 4f
aa 9a 33 d2 85 e0 70 ad 5b 1b 48 51 b4 01 0d 99 d2 cd 1a 68 f1 e6 b8 10 ff f8 69 18 01 02 a4 02 c3 82 c4 0b
c1 4a 03 ee 48 dd 03 b6 7c 13 30
"""
EX3 = """
66 4c 61 43 80 00 00 22 10 00 10 00 00 00 1f 00 00 1f 07 d0 00 70 00 00 00 18 f8 f9 e3 96 f5 cb cf c6 dc 80
7f 99 77 90 6b 32 ff f8 68 02 00 17 e9 44 00 4f 6f 31 3d 10 47 d2 27 cb 6d 09 08 31 45 2b dc 28 22 22 80 57
a3
"""
# decoded samples [channel][sample]
PCM1 = [[25588], [10416]]
PCM2 = [[10372, 18041, 14942, 17876, 15627, 17899, 16242, 18077, 16824, 18263, 17295, -14418, -15201, -14508, -15195, -14818, -15486, -15349,
         -16054],
        [6070, 10545, 8743, 10449, 9143, 10463, 9502, 10569, 9840, 10680, 10113, -8428, -8895, -8476, -8896, -8653, -9072, -8958, -9410]]
PCM3 = [[0, 79, 111, 78, 8, -61, -90, -68, -13, 42, 67, 53, 13, -27, -46, -38, -12, 14, 24, 19, 6, -4, -5, 0]]
FRAMES = {"ex1": [(42, 15)], "ex2": [(136, 68), (204, 23)], "ex3": [(42, 31)]}      # (offset, length) of every audio frame


def crc(data: bytes, poly: int, bits: int) -> int:
    c, top, mask = 0, 1 << (bits - 1), (1 << bits) - 1
    for x in data:
        c ^= x << (bits - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def check(name: str, stream: bytes, pcm) -> None:
    assert stream[:4] == b"fLaC"
    si = stream[8:8 + 34]
    bps = (((si[12] & 1) << 4) | (si[13] >> 4)) + 1
    total = ((si[13] & 0xf) << 32) | int.from_bytes(si[14:18], "big")
    assert total == len(pcm[0]), (name, total)
    end = 0
    for off, n in FRAMES[name]:
        f = stream[off:off + n]
        assert f[0] == 0xff and (f[1] & 0xfe) == 0xf8, name
        hdr = 6 if (f[2] >> 4) == 6 else 5                      # 8-bit block size follows the frame number in these examples
        assert crc(f[:hdr], 0x07, 8) == f[hdr], (name, "header CRC-8")
        assert crc(f[:-2], 0x8005, 16) == int.from_bytes(f[-2:], "big"), (name, "frame CRC-16")
        end = off + n
    assert end == len(stream), name
    inter = np.asarray(pcm, dtype=np.int64).T.reshape(-1)
    raw = b"".join(int(v).to_bytes((bps + 7) // 8, "little", signed=True) for v in inter)
    assert hashlib.md5(raw).digest() == si[18:34], (name, "MD5 of the listed samples vs STREAMINFO")


def main() -> None:
    out = {}
    for name, hexs, pcm in (("ex1", EX1, PCM1), ("ex2", EX2, PCM2), ("ex3", EX3, PCM3)):
        stream = bytes.fromhex(hexs.replace("\n", " "))
        check(name, stream, pcm)
        out[name + "_bytes"] = np.frombuffer(stream, dtype=np.uint8)
        out[name + "_pcm"] = np.asarray(pcm, dtype=np.int32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flac_rfc9639.npz")
    np.savez(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
