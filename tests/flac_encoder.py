"""A small FLAC ENCODER used only to make test streams for the native decoder (wavjepa_amd/csrc/flac_decode.cpp).

Written from the format specification, independently of the decoder, and deliberately parameterised so that tests can force every
bitstream feature the decoder implements: subframe types (constant / verbatim / fixed order 0-4 / LPC with given coefficients),
Rice and Rice2 residual coding with chosen partition order, parameters and escape partitions, wasted bits, the four channel
assignments, 8 / 12 / 16 / 20 / 24-bit samples, block-size codes (table values, 8-bit and 16-bit explicit sizes), a short last
frame, explicit sample-rate codes, the STREAMINFO MD5, extra metadata blocks and an ID3v2 prefix."""
import hashlib
import struct

import numpy as np


class BitWriter:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def write(self, value: int, bits: int):
        if bits == 0:
            return
        value &= (1 << bits) - 1
        self.acc = (self.acc << bits) | value
        self.n += bits
        while self.n >= 8:
            self.n -= 8
            self.buf.append((self.acc >> self.n) & 0xff)
        self.acc &= (1 << self.n) - 1

    def unary(self, q: int):
        for _ in range(q // 32):
            self.write(0, 32)
        self.write(0, q % 32)
        self.write(1, 1)

    def align(self):
        if self.n:
            self.write(0, 8 - self.n)

    def bytes(self) -> bytes:
        assert self.n == 0
        return bytes(self.buf)


def crc8(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def crc16(data: bytes) -> int:
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


def utf8_number(v: int) -> bytes:
    if v < 0x80:
        return bytes([v])
    out, n = [], 0
    while True:
        n += 1
        lim = 1 << (6 - n)                     # payload bits left in the first byte
        out.append(0x80 | (v & 0x3f))
        v >>= 6
        if v < lim:
            first = ((0xff << (7 - n)) & 0xff) | v
            return bytes([first] + out[::-1])


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _residual(x, order, coefs, shift):
    x = [int(v) for v in x]
    res = []
    for i in range(order, len(x)):
        pred = sum(c * x[i - 1 - j] for j, c in enumerate(coefs)) >> shift
        res.append(x[i] - pred)
    return res


def _write_residual(bw, res, blocksize, order, porder, rice2, params, escape_parts):
    bw.write(1 if rice2 else 0, 2)
    bw.write(porder, 4)
    pbits, esc = (5, 31) if rice2 else (4, 15)
    idx = 0
    for p in range(1 << porder):
        count = (blocksize >> porder) - (order if p == 0 else 0)
        part = res[idx:idx + count]
        idx += count
        if p in escape_parts:
            bw.write(esc, pbits)
            nb = max([1] + [v.bit_length() + 1 for v in part])
            bw.write(nb, 5)
            for v in part:
                bw.write(v, nb)
            continue
        k = params[p % len(params)] if params else None
        if k is None:                           # a reasonable parameter for this partition
            mean = (sum(abs(v) for v in part) / max(1, len(part))) if part else 0
            k = max(0, min(esc - 1, int(np.log2(mean + 1))))
        bw.write(k, pbits)
        for v in part:
            u = (v << 1) if v >= 0 else ((-v) << 1) - 1
            bw.unary(u >> k)
            bw.write(u & ((1 << k) - 1), k)


def _subframe(bw, x, bps, spec):
    """spec: dict(kind='constant'|'verbatim'|'fixed'|'lpc', order, coefs, shift, precision, porder, rice2, params, escape_parts, wasted)"""
    wasted = spec.get("wasted", 0)
    if wasted:
        assert all((int(v) & ((1 << wasted) - 1)) == 0 for v in x)
        x = [int(v) >> wasted for v in x]
        bps -= wasted
    kind = spec["kind"]
    order = spec.get("order", 0)
    code = {"constant": 0, "verbatim": 1}.get(kind)
    if kind == "fixed":
        code = 8 + order
    elif kind == "lpc":
        code = 32 + order - 1
    bw.write(0, 1)
    bw.write(code, 6)
    if wasted:
        bw.write(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.write(0, 1)
    n = len(x)
    if kind == "constant":
        assert all(v == x[0] for v in x)
        bw.write(int(x[0]), bps)
        return
    if kind == "verbatim":
        for v in x:
            bw.write(int(v), bps)
        return
    for v in x[:order]:
        bw.write(int(v), bps)
    if kind == "fixed":
        coefs, shift = FIXED[order], 0
    else:
        coefs, shift, prec = spec["coefs"], spec["shift"], spec["precision"]
        bw.write(prec - 1, 4)
        bw.write(shift, 5)
        for c in coefs:
            bw.write(int(c), prec)
    res = _residual(x, order, coefs, shift)
    _write_residual(bw, res, n, order, spec.get("porder", 0), spec.get("rice2", False), spec.get("params"), spec.get("escape_parts", ()))


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SR_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
BPS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode(pcm: np.ndarray, sample_rate: int, bps: int, blocksize: int = 4096, stereo: str = "independent", subframes=None,
           sr_in_header: str = "table", bps_in_header: bool = True, variable: bool = False, md5: bool = True, extra_blocks=(),
           id3: bool = False, total_in_header: bool = True) -> bytes:
    """pcm int [samples, channels].  subframes: one spec (all channels / frames), a list per channel, or a callable(frame, ch)."""
    pcm = np.asarray(pcm, dtype=np.int64)
    n, ch = pcm.shape
    frames = bytearray()
    pos, fno = 0, 0
    min_f, max_f = 1 << 24, 0
    while pos < n:
        bs = min(blocksize, n - pos)
        blk = pcm[pos:pos + bs]
        bw = BitWriter()
        bw.write(0b11111111111110, 14)
        bw.write(0, 1)
        bw.write(1 if variable else 0, 1)
        if bs in BS_CODES:
            bs_code = BS_CODES[bs]
        else:
            bs_code = 6 if bs <= 256 else 7
        bw.write(bs_code, 4)
        if sr_in_header == "streaminfo":
            sr_code = 0
        elif sr_in_header == "table" and sample_rate in SR_CODES:
            sr_code = SR_CODES[sample_rate]
        elif sample_rate % 1000 == 0 and sample_rate // 1000 < 256 and sr_in_header != "hz":
            sr_code = 12
        elif sample_rate < 65536:
            sr_code = 13
        else:
            sr_code = 14
        bw.write(sr_code, 4)
        ch_code = {"independent": ch - 1, "left_side": 8, "right_side": 9, "mid_side": 10}[stereo]
        bw.write(ch_code, 4)
        bw.write(BPS_CODES[bps] if bps_in_header else 0, 3)
        bw.write(0, 1)
        for b in utf8_number(pos if variable else fno):
            bw.write(b, 8)
        if bs_code == 6:
            bw.write(bs - 1, 8)
        elif bs_code == 7:
            bw.write(bs - 1, 16)
        if sr_code == 12:
            bw.write(sample_rate // 1000, 8)
        elif sr_code == 13:
            bw.write(sample_rate, 16)
        elif sr_code == 14:
            bw.write(sample_rate // 10, 16)
        bw.write(crc8(bw.bytes()), 8)
        if stereo == "independent":
            chans, widths = [blk[:, c] for c in range(ch)], [bps] * ch
        else:
            left, right = blk[:, 0], blk[:, 1]
            side = left - right
            if stereo == "left_side":
                chans, widths = [left, side], [bps, bps + 1]
            elif stereo == "right_side":
                chans, widths = [side, right], [bps + 1, bps]
            else:
                chans, widths = [(left + right) >> 1, side], [bps, bps + 1]
        for c, (x, w) in enumerate(zip(chans, widths)):
            if callable(subframes):
                spec = subframes(fno, c)
            elif isinstance(subframes, (list, tuple)):
                spec = subframes[c]
            else:
                spec = subframes or dict(kind="fixed", order=2)
            if spec["kind"] in ("fixed", "lpc") and spec.get("order", 0) > len(x):
                spec = dict(kind="verbatim")
            if spec.get("porder", 0) and (bs % (1 << spec["porder"]) or (bs >> spec["porder"]) < spec.get("order", 0)):
                spec = dict(spec, porder=0)
            _subframe(bw, list(x), w, spec)
        bw.align()
        body = bw.bytes()
        frame = body + struct.pack(">H", crc16(body))
        min_f, max_f = min(min_f, len(frame)), max(max_f, len(frame))
        frames += frame
        pos += bs
        fno += 1
    nbytes = (bps + 7) // 8
    sig = b"\0" * 16
    if md5:
        raw = np.ascontiguousarray(pcm.astype("<i4")).tobytes()
        sig = hashlib.md5(np.frombuffer(raw, dtype=np.uint8).reshape(-1, 4)[:, :nbytes].tobytes()).digest()
    si = BitWriter()
    si.write(blocksize if not variable else 16, 16)
    si.write(blocksize, 16)
    si.write(min_f, 24)
    si.write(max_f, 24)
    si.write(sample_rate, 20)
    si.write(ch - 1, 3)
    si.write(bps - 1, 5)
    si.write(n if total_in_header else 0, 36)
    out = bytearray()
    if id3:
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, 10]) + b"\0" * 10
    out += b"fLaC"
    blocks = [(0, si.bytes() + sig)] + list(extra_blocks)
    for i, (btype, payload) in enumerate(blocks):
        last = 0x80 if i == len(blocks) - 1 else 0
        out += bytes([last | btype]) + struct.pack(">I", len(payload))[1:] + payload
    return bytes(out + frames)
