// Host-side FLAC decoder for the input pipeline (no GPU work).  The reference reads the ASVspoof .flac files with
// librosa.load (oc_training.py:214, 234; oc_classifier.py:93; data_utils_SSL.py:66, 91), i.e. libsndfile; neither is available here and the
// data loader should not need them.  This is the FLAC format as published (subset: 4-32 bits per sample, up to 8 channels, every
// subframe type, both Rice methods with escapes, wasted bits, the three stereo decorrelations, fixed or variable block size).
// Frame-header CRC-8 and frame CRC-16 are verified here; the STREAMINFO MD5 of the decoded PCM is verified by the caller
// (occm_amd/data_utils_SSL.py), so a stream this decoder misreads is reported, never silently returned.
#include "occ_common.h"
#include <vector>

namespace {

struct BitReader {
    const uint8_t* p; int64_t n; int64_t bit;               // bit = absolute bit position
    bool ok;
    BitReader(const uint8_t* p_, int64_t n_, int64_t byte0) : p(p_), n(n_), bit(byte0 * 8), ok(true) {}
    inline uint32_t get(int nb) {                           // nb <= 32, MSB first
        uint64_t v = 0;
        if (nb == 0) return 0;
        if (bit + nb > n * 8) { ok = false; bit = n * 8; return 0; }
        int64_t b = bit; int left = nb;
        while (left > 0) {
            const int off = (int)(b & 7), take = 8 - off < left ? 8 - off : left;
            const uint32_t byte = p[b >> 3];
            v = (v << take) | ((byte >> (8 - off - take)) & ((1u << take) - 1u));
            b += take; left -= take;
        }
        bit = b;
        return (uint32_t)v;
    }
    inline int64_t gets(int nb) {                           // signed, nb <= 33 (side channel of 32-bit audio)
        if (nb == 0) return 0;
        uint64_t v;
        if (nb > 32) { v = ((uint64_t)get(nb - 32) << 32) | get(32); }
        else v = get(nb);
        const uint64_t sign = 1ull << (nb - 1);
        return (int64_t)((v ^ sign)) - (int64_t)sign;
    }
    inline uint32_t unary() {                               // zeros up to the terminating one
        uint32_t q = 0;
        while (true) {
            if (bit >= n * 8) { ok = false; return q; }
            const int off = (int)(bit & 7);
            const uint32_t rest = (uint32_t)(p[bit >> 3] << off) & 0xffu;           // remaining bits of this byte, left-aligned
            if (rest == 0) { q += 8 - off; bit += 8 - off; continue; }
            const int lz = __builtin_clz(rest) - 24;
            q += lz; bit += lz + 1;
            return q;
        }
    }
    inline void align() { bit = (bit + 7) & ~7ll; }
    inline int64_t byte_pos() const { return bit >> 3; }
};

uint8_t crc8(const uint8_t* d, int64_t n) {
    uint8_t c = 0;
    for (int64_t i = 0; i < n; ++i) { c ^= d[i]; for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1); }
    return c;
}
uint16_t crc16(const uint8_t* d, int64_t n) {
    uint16_t c = 0;
    for (int64_t i = 0; i < n; ++i) { c ^= (uint16_t)(d[i] << 8); for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1); }
    return c;
}

struct StreamInfo { int64_t first_frame; int sr, channels, bps; int64_t total; uint8_t md5[16]; };

int parse_header(const uint8_t* buf, int64_t n, StreamInfo& si) {
    int64_t pos = 0;
    if (n >= 10 && buf[0] == 'I' && buf[1] == 'D' && buf[2] == '3')      // ID3v2 tag in front of the stream
        pos = 10 + (((int64_t)buf[6] & 0x7f) << 21 | ((int64_t)buf[7] & 0x7f) << 14 | ((int64_t)buf[8] & 0x7f) << 7 | ((int64_t)buf[9] & 0x7f));
    if (pos + 4 > n || buf[pos] != 'f' || buf[pos + 1] != 'L' || buf[pos + 2] != 'a' || buf[pos + 3] != 'C') { occ_set_error("occ_flac: no fLaC marker"); return OCC_EINVAL; }
    pos += 4;
    bool have = false;
    while (true) {
        if (pos + 4 > n) { occ_set_error("occ_flac: truncated metadata"); return OCC_EINVAL; }
        const bool last = buf[pos] & 0x80; const int type = buf[pos] & 0x7f;
        const int64_t len = ((int64_t)buf[pos + 1] << 16) | ((int64_t)buf[pos + 2] << 8) | buf[pos + 3];
        pos += 4;
        if (pos + len > n) { occ_set_error("occ_flac: truncated metadata block"); return OCC_EINVAL; }
        if (type == 0) {
            if (len < 34) { occ_set_error("occ_flac: short STREAMINFO"); return OCC_EINVAL; }
            const uint8_t* s = buf + pos;
            si.sr = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
            si.channels = ((s[12] >> 1) & 7) + 1;
            si.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si.total = ((int64_t)(s[13] & 0xf) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
            for (int i = 0; i < 16; ++i) si.md5[i] = s[18 + i];
            have = true;
        }
        pos += len;
        if (last) break;
    }
    if (!have) { occ_set_error("occ_flac: no STREAMINFO block"); return OCC_EINVAL; }
    si.first_frame = pos;
    return OCC_OK;
}

bool decode_residual(BitReader& br, int64_t* out, int bs, int pred_order) {
    const int method = (int)br.get(2);
    if (method > 1) return false;
    const int pb = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = (int)br.get(4);
    const int parts = 1 << porder;
    if ((bs >> porder) << porder != bs && porder > 0) return false;
    if ((bs >> porder) < pred_order) return false;
    int idx = pred_order;
    for (int p = 0; p < parts; ++p) {
        const int cnt = (bs >> porder) - (p == 0 ? pred_order : 0);
        const int k = (int)br.get(pb);
        if (k == esc) {
            const int nb = (int)br.get(5);
            for (int i = 0; i < cnt; ++i) out[idx++] = br.gets(nb);
        } else {
            for (int i = 0; i < cnt; ++i) {
                const uint32_t q = br.unary();
                const uint64_t v = ((uint64_t)q << k) | (k ? br.get(k) : 0u);
                out[idx++] = (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
            }
        }
        if (!br.ok) return false;
    }
    return true;
}

bool decode_subframe(BitReader& br, int64_t* s, int bs, int bps) {
    if (br.get(1) != 0) return false;
    const int type = (int)br.get(6);
    int wasted = 0;
    if (br.get(1)) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps < 1) return false;
    if (type == 0) {
        const int64_t v = br.gets(bps);
        for (int i = 0; i < bs; ++i) s[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < bs; ++i) s[i] = br.gets(bps);
    } else if (type >= 8 && type <= 12) {
        const int o = type - 8;
        if (o > bs) return false;
        for (int i = 0; i < o; ++i) s[i] = br.gets(bps);
        if (!decode_residual(br, s, bs, o)) return false;
        for (int i = o; i < bs; ++i) {
            switch (o) {
                case 0: break;
                case 1: s[i] += s[i - 1]; break;
                case 2: s[i] += 2 * s[i - 1] - s[i - 2]; break;
                case 3: s[i] += 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
                default: s[i] += 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4]; break;
            }
        }
    } else if (type >= 32) {
        const int o = (type & 31) + 1;
        if (o > bs) return false;
        for (int i = 0; i < o; ++i) s[i] = br.gets(bps);
        const int prec = (int)br.get(4) + 1;
        if (prec == 16) return false;
        const int shift = (int)br.gets(5);
        if (shift < 0) return false;
        int64_t coef[32];
        for (int j = 0; j < o; ++j) coef[j] = br.gets(prec);
        if (!decode_residual(br, s, bs, o)) return false;
        for (int i = o; i < bs; ++i) {
            int64_t acc = 0;
            for (int j = 0; j < o; ++j) acc += coef[j] * s[i - 1 - j];
            s[i] += acc >> shift;
        }
    } else {
        return false;                                        // reserved subframe type
    }
    if (wasted)
        for (int i = 0; i < bs; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
    return br.ok;
}

}  // namespace

// info[0..3] = sample rate, channels, bits per sample, 1 if the STREAMINFO MD5 is set; total = samples per channel (0 = unknown); md5 = 16 bytes
extern "C" int occ_flac_info(const uint8_t* buf, int64_t n, int32_t* info, int64_t* total, uint8_t* md5) {
    OCC_CHECK_ARG(buf && info && total && n > 0, "occ_flac_info: bad argument");
    StreamInfo si;
    const int rc = parse_header(buf, n, si);
    if (rc != OCC_OK) return rc;
    info[0] = si.sr; info[1] = si.channels; info[2] = si.bps;
    int any = 0;
    for (int i = 0; i < 16; ++i) { any |= si.md5[i]; if (md5) md5[i] = si.md5[i]; }
    info[3] = any ? 1 : 0;
    *total = si.total;
    return OCC_OK;
}

// out: interleaved int32 [capacity * channels]; *decoded = samples per channel written.  Stops at the end of the buffer or at `capacity`.
extern "C" int occ_flac_decode(const uint8_t* buf, int64_t n, int32_t* out, int64_t capacity, int64_t* decoded) {
    OCC_CHECK_ARG(buf && out && decoded && n > 0 && capacity >= 0, "occ_flac_decode: bad argument");
    StreamInfo si;
    int rc = parse_header(buf, n, si);
    if (rc != OCC_OK) return rc;
    int64_t pos = si.first_frame, done = 0;
    std::vector<int64_t> ch[8];
    while (pos + 2 <= n && done < capacity) {
        if (!(buf[pos] == 0xff && (buf[pos + 1] & 0xfe) == 0xf8)) {                  // 14 sync bits + reserved 0
            if (si.total && done >= si.total) break;                                // trailing bytes after the last frame
            occ_set_error("occ_flac_decode: lost frame sync at byte %ld", (long)pos); return OCC_EINVAL;
        }
        BitReader br(buf, n, pos);
        br.get(16);
        const int bsc = (int)br.get(4), src = (int)br.get(4), chan = (int)br.get(4), ssc = (int)br.get(3);
        if (br.get(1) != 0) { occ_set_error("occ_flac_decode: reserved header bit set"); return OCC_EINVAL; }
        int lead = 0;                                                                // UTF-8 coded frame / sample number: skip
        { const uint32_t b0 = br.get(8); while (lead < 7 && (b0 & (0x80u >> lead))) ++lead; for (int i = 1; i < lead; ++i) br.get(8); }
        int bs;
        if (bsc == 0) { occ_set_error("occ_flac_decode: reserved block size code"); return OCC_EINVAL; }
        else if (bsc == 1) bs = 192;
        else if (bsc <= 5) bs = 576 << (bsc - 2);
        else if (bsc == 6) bs = (int)br.get(8) + 1;
        else if (bsc == 7) bs = (int)br.get(16) + 1;
        else bs = 256 << (bsc - 8);
        if (src == 12) br.get(8); else if (src == 13 || src == 14) br.get(16); else if (src == 15) { occ_set_error("occ_flac_decode: invalid sample rate code"); return OCC_EINVAL; }
        int bps = si.bps;
        switch (ssc) { case 0: break; case 1: bps = 8; break; case 2: bps = 12; break; case 4: bps = 16; break; case 5: bps = 20; break; case 6: bps = 24; break; case 7: bps = 32; break;
                       default: occ_set_error("occ_flac_decode: reserved sample size code"); return OCC_EINVAL; }
        const int64_t hdr_end = br.byte_pos();
        if (!br.ok || hdr_end + 1 > n || crc8(buf + pos, hdr_end - pos) != buf[hdr_end]) { occ_set_error("occ_flac_decode: frame header CRC mismatch at byte %ld", (long)pos); return OCC_EINVAL; }
        br.get(8);
        const int nch = chan < 8 ? chan + 1 : 2;
        if (chan > 10 || nch != si.channels) { occ_set_error("occ_flac_decode: channel assignment %d does not match STREAMINFO (%d channels)", chan, si.channels); return OCC_EINVAL; }
        for (int c = 0; c < nch; ++c) {
            ch[c].resize(bs);
            const int side = (chan == 8 && c == 1) || (chan == 9 && c == 0) || (chan == 10 && c == 1);
            if (!decode_subframe(br, ch[c].data(), bs, bps + side)) { occ_set_error("occ_flac_decode: bad subframe (frame at byte %ld, channel %d)", (long)pos, c); return OCC_EINVAL; }
        }
        br.align();
        const int64_t end = br.byte_pos();
        if (end + 2 > n || crc16(buf + pos, end - pos) != (uint16_t)((buf[end] << 8) | buf[end + 1])) { occ_set_error("occ_flac_decode: frame CRC mismatch at byte %ld", (long)pos); return OCC_EINVAL; }
        if (chan == 8) for (int i = 0; i < bs; ++i) ch[1][i] = ch[0][i] - ch[1][i];
        else if (chan == 9) for (int i = 0; i < bs; ++i) ch[0][i] = ch[0][i] + ch[1][i];
        else if (chan == 10)
            for (int i = 0; i < bs; ++i) {
                const int64_t side = ch[1][i], mid = (int64_t)((uint64_t)ch[0][i] << 1) | (side & 1);
                ch[0][i] = (mid + side) >> 1; ch[1][i] = (mid - side) >> 1;
            }
        const int64_t take = done + bs <= capacity ? bs : capacity - done;
        for (int c = 0; c < nch; ++c)
            for (int64_t i = 0; i < take; ++i) out[(done + i) * nch + c] = (int32_t)ch[c][i];
        done += take;
        pos = end + 2;
    }
    *decoded = done;
    return OCC_OK;
}
