"""TEST HELPER: a small FLAC *encoder* written from the published format, so the host decoder (occ_flac_decode) can be exercised
without libFLAC: every subframe type (constant, verbatim, fixed orders 0-4, LPC), both Rice methods with escape partitions,
partition orders, wasted bits, independent / left-side / side-right / mid-side stereo, explicit 8- and 16-bit block sizes,
CRC-8, CRC-16 and the STREAMINFO MD5.  Not a product file."""
import hashlib

import numpy as np


class BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        v &= (1 << n) - 1
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    def unary(self, q):
        self.bits.extend([0] * q + [1])

    def align(self):
        self.bits.extend([0] * (-len(self.bits) % 8))

    def bytes(self):
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, dtype=np.uint8)).tobytes()


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xff if c & 0x80 else (c << 1) & 0xff
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xffff if c & 0x8000 else (c << 1) & 0xffff
    return c


def _utf8(n):
    if n < 0x80:
        return [n]
    out, lead = [], 0
    while True:
        out.insert(0, 0x80 | (n & 0x3f)); n >>= 6; lead += 1
        if n < (0x40 >> lead):
            out.insert(0, ((0xff << (7 - lead)) & 0xff) | n)
            return out


def _residual(bw, res, bs, order, method, porder, force_escape=False):
    pb, esc = (4, 15) if method == 0 else (5, 31)
    bw.put(method, 2); bw.put(porder, 4)
    idx = 0
    for p in range(1 << porder):
        cnt = (bs >> porder) - (order if p == 0 else 0)
        part = [int(v) for v in res[idx:idx + cnt]]; idx += cnt
        zz = [(v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1 for v in part]
        best_k, best_len = 0, None
        for k in range(esc):
            ln = sum((z >> k) + 1 + k for z in zz)
            if best_len is None or ln < best_len:
                best_k, best_len = k, ln
        if force_escape and p == 0:
            nb = max([0] + [int(v).bit_length() + 1 for v in part])
            bw.put(esc, pb); bw.put(nb, 5)
            for v in part:
                bw.put(v, nb)
            continue
        bw.put(best_k, pb)
        for z in zz:
            bw.unary(z >> best_k)
            if best_k:
                bw.put(z, best_k)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _subframe(bw, x, bps, kind, method=0, porder=0, force_escape=False, lpc=None):
    x = [int(v) for v in x]
    bs = len(x)
    wasted = 0
    if kind.endswith("+wasted"):
        kind = kind[:-7]
        nz = [v for v in x if v]
        wasted = min((v & -v).bit_length() - 1 for v in nz) if nz else 0
        x = [v >> wasted for v in x]
    bw.put(0, 1)
    if kind == "constant":
        bw.put(0, 6)
    elif kind == "verbatim":
        bw.put(1, 6)
    elif kind.startswith("fixed"):
        bw.put(8 + int(kind[5:]), 6)
    else:
        bw.put(32 + len(lpc[0]) - 1, 6)
    if wasted:
        bw.put(1, 1); bw.unary(wasted - 1)
    else:
        bw.put(0, 1)
    b = bps - wasted
    if kind == "constant":
        assert len(set(x)) == 1
        bw.put(x[0], b)
    elif kind == "verbatim":
        for v in x:
            bw.put(v, b)
    elif kind.startswith("fixed"):
        o = int(kind[5:]); c = FIXED[o]
        for v in x[:o]:
            bw.put(v, b)
        res = [0] * o + [x[i] - sum(c[j] * x[i - 1 - j] for j in range(o)) for i in range(o, bs)]
        _residual(bw, res[o:], bs, o, method, porder, force_escape)
    else:
        coefs, prec, shift = lpc
        o = len(coefs)
        for v in x[:o]:
            bw.put(v, b)
        bw.put(prec - 1, 4); bw.put(shift, 5)
        for cf in coefs:
            bw.put(cf, prec)
        res = [x[i] - (sum(coefs[j] * x[i - 1 - j] for j in range(o)) >> shift) for i in range(o, bs)]
        _residual(bw, res, bs, o, method, porder, force_escape)


def encode(pcm, fs=16000, bps=16, blocksize=4096, kinds=("fixed2",), stereo="independent", method=0, porder=0, force_escape=False, lpc=None,
           variable=False, with_md5=True):
    """pcm int array [n, channels].  kinds: subframe kind per frame (cycled).  Returns the file bytes."""
    pcm = np.asarray(pcm).reshape(len(pcm), -1).astype(np.int64)
    n, nch = pcm.shape
    frames, pos, fno = [], 0, 0
    while pos < n:
        bs = min(blocksize, n - pos)
        blk = pcm[pos:pos + bs]
        bw = BitWriter()
        bw.put(0x3ffe, 14); bw.put(0, 1); bw.put(1 if variable else 0, 1)
        table = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
        bsc = table.get(bs, 6 if bs <= 256 else 7)
        bw.put(bsc, 4); bw.put(0, 4)                                        # sample rate: from STREAMINFO
        chan = {"independent": nch - 1, "left_side": 8, "side_right": 9, "mid_side": 10}[stereo]
        bw.put(chan, 4); bw.put({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}.get(bps, 0), 3); bw.put(0, 1)
        for byte in _utf8(pos if variable else fno):
            bw.put(byte, 8)
        if bsc == 6:
            bw.put(bs - 1, 8)
        elif bsc == 7:
            bw.put(bs - 1, 16)
        hdr = bw.bytes()
        bw.put(crc8(hdr), 8)
        chans = [blk[:, c] for c in range(nch)]
        widths = [bps] * nch
        if stereo == "left_side":
            chans = [blk[:, 0], blk[:, 0] - blk[:, 1]]; widths = [bps, bps + 1]
        elif stereo == "side_right":
            chans = [blk[:, 0] - blk[:, 1], blk[:, 1]]; widths = [bps + 1, bps]
        elif stereo == "mid_side":
            chans = [(blk[:, 0] + blk[:, 1]) >> 1, blk[:, 0] - blk[:, 1]]; widths = [bps, bps + 1]
        kind = kinds[fno % len(kinds)]
        po = porder
        while po > 0 and (bs % (1 << po) or (bs >> po) < 5):
            po -= 1
        for c, w in zip(chans, widths):
            k = kind
            if k == "constant" and len(set(int(v) for v in c)) != 1:
                k = "verbatim"
            _subframe(bw, c, w, k, method, po, force_escape, lpc)
        bw.align()
        body = bw.bytes()
        frames.append(body + crc16(body).to_bytes(2, "big"))
        pos += bs; fno += 1
    nbytes = (bps + 7) // 8
    md5 = hashlib.md5(np.ascontiguousarray(pcm.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :nbytes]).tobytes()).digest() if with_md5 else bytes(16)
    si = BitWriter()
    si.put(blocksize, 16); si.put(blocksize, 16); si.put(0, 24); si.put(0, 24)
    si.put(fs, 20); si.put(nch - 1, 3); si.put(bps - 1, 5); si.put(n, 36)
    info = si.bytes() + md5
    pad = bytes([0x81, 0, 0, 4, 0, 0, 0, 0])                                # a PADDING block, flagged last
    return b"fLaC" + bytes([0x00, 0, 0, 34]) + info + pad + b"".join(frames)
