// pxg_zcodec.cpp -- libpxghost.so, host only: the lossless sample encoding a read bundle may
// carry so that a batch crosses PCIe as ~1.1 bytes per sample instead of 2 (the end-to-end
// pipeline is bound by the H2D copy: 1.2 GB per 10 000 reads at 57 GB/s = 21 ms against
// 15 ms of kernels).  It is the variable-byte stage of ONT's VBZ (zig-zag deltas, one or two
// bytes each) cut into independent 1 024-sample chunks so that the device decodes every
// chunk with one workgroup (k_z_decode, pxg_api.hip):
//   chunk = 128 control bytes (bit i set: sample i took two bytes) + the data bytes of
//   samples 1 .. len-1; sample 0 sits in the chunk record.  Chunks never span reads.
// Encoder (bundle writing, offline) and reference decoder (tests, per-read host access).
#include <cstdint>
#include <cstring>
#include "../../include/pxg.h"
#include "pxg_zcheck.h"

extern "C" int64_t pxg_z_count_chunks(int64_t n_reads, const int64_t* offsets)
{
    int64_t n = 0;
    for (int64_t r = 0; r < n_reads; r++)
        n += (offsets[r + 1] - offsets[r] + PXG_Z_CHUNK - 1) / PXG_Z_CHUNK;
    return n;
}

// PXG_Z_PACKED (round 4): the same zig-zag deltas, bit-packed.  A chunk = 128 header bytes (256 nibbles: the
// bit width of each group of FOUR deltas, code 15 = 16 bits; delta 0 of a chunk is 0, sample 0 sits in the
// record; groups past the chunk's end have width 0) + the groups' 4 x w bits back to back, least significant
// bit first, padded to a whole byte.  Nanopore samples dwell on a level and jump: most groups need 6 bits,
// the jumps 9 - 10, and a width per four samples follows that where a width per byte-code (VBZ's 1-or-2
// bytes) or per sixteen samples cannot: 0.96 instead of 1.19 bytes per sample on the bench signal
// (first-order entropy of its deltas: 0.83).
static inline int z_bit_width(unsigned v)
{
    int w = 0;
    while (v) { w++; v >>= 1; }
    return w == 15 ? 16 : w;
}

static int64_t z_encode_packed(int64_t n_reads, const int16_t* arena, const int64_t* offsets, uint8_t* out,
                               int64_t cap, pxg_z_chunk* chunks)
{
    int64_t at = 0, g = 0;
    uint16_t zz[PXG_Z_CHUNK];
    for (int64_t r = 0; r < n_reads; r++) {
        for (int64_t s0 = offsets[r]; s0 < offsets[r + 1]; s0 += PXG_Z_CHUNK, g++) {
            const int64_t len = (offsets[r + 1] - s0) < PXG_Z_CHUNK ? (offsets[r + 1] - s0) : PXG_Z_CHUNK;
            if (at + PXG_Z_CTRL_BYTES + 2 * PXG_Z_CHUNK + 8 > cap) return PXG_E_NOMEM;
            pxg_z_chunk& c = chunks[g];
            c.data_off = at;
            c.dst = s0;
            c.first = arena[s0];
            c.len = (int16_t)len;
            c.codec = PXG_Z_PACKED;
            zz[0] = 0;
            for (int64_t i = 1; i < len; i++) {
                const int16_t d = (int16_t)((uint16_t)arena[s0 + i] - (uint16_t)arena[s0 + i - 1]);
                zz[i] = (uint16_t)(((uint16_t)d << 1) ^ (uint16_t)(d >> 15));
            }
            for (int64_t i = len; i < PXG_Z_CHUNK; i++) zz[i] = 0;
            uint8_t* hdr = out + at;
            memset(hdr, 0, PXG_Z_CTRL_BYTES);
            uint8_t* p = hdr + PXG_Z_CTRL_BYTES;
            uint64_t acc = 0;
            int have = 0;
            for (int grp = 0; grp < PXG_Z_CHUNK / 4; grp++) {
                const uint16_t* q = zz + 4 * grp;
                const int w = z_bit_width((unsigned)(q[0] | q[1] | q[2] | q[3]));
                hdr[grp >> 1] |= (uint8_t)((w == 16 ? 15 : w) << ((grp & 1) * 4));
                for (int j = 0; j < 4 && w; j++) {
                    acc |= (uint64_t)q[j] << have;
                    have += w;
                    while (have >= 8) { *p++ = (uint8_t)acc; acc >>= 8; have -= 8; }
                }
            }
            if (have) *p++ = (uint8_t)acc;
            at = p - out;
        }
    }
    return at;
}

static void z_decode_packed(const pxg_z_chunk& c, const uint8_t* hdr, const uint8_t* end, int16_t* dst)
{
    const uint8_t* p = hdr + PXG_Z_CTRL_BYTES;
    uint64_t acc = 0;
    int have = 0;
    uint16_t v = (uint16_t)c.first;
    for (int i = 0; i < c.len; i++) {
        const int code = (hdr[i >> 3] >> (((i >> 2) & 1) * 4)) & 15;
        const int w = code == 15 ? 16 : code;
        while (have < w) { acc |= (uint64_t)(p < end ? *p : 0) << have; p++; have += 8; }
        const uint16_t zz = (uint16_t)(acc & ((1u << w) - 1u));
        acc >>= w;
        have -= w;
        const uint16_t d = (uint16_t)((zz >> 1) ^ (uint16_t)(-(int16_t)(zz & 1)));
        v = (uint16_t)(v + d);
        dst[i] = (int16_t)v;
    }
}

extern "C" int64_t pxg_z_encode_as(int64_t n_reads, const int16_t* arena, const int64_t* offsets, uint8_t* out,
                                   int64_t cap, pxg_z_chunk* chunks, int32_t codec)
{
    if (n_reads < 0 || (n_reads && (!offsets || !chunks)) || (!out && cap)) return PXG_E_INVALID;
    if (codec == PXG_Z_PACKED) return z_encode_packed(n_reads, arena, offsets, out, cap, chunks);
    if (codec != PXG_Z_BYTES) return PXG_E_INVALID;
    return pxg_z_encode(n_reads, arena, offsets, out, cap, chunks);
}

extern "C" int64_t pxg_z_encode(int64_t n_reads, const int16_t* arena, const int64_t* offsets, uint8_t* out,
                                int64_t cap, pxg_z_chunk* chunks)
{
    if (n_reads < 0 || (n_reads && (!offsets || !chunks)) || (!out && cap)) return PXG_E_INVALID;
    int64_t at = 0, g = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        for (int64_t s0 = offsets[r]; s0 < offsets[r + 1]; s0 += PXG_Z_CHUNK, g++) {
            const int64_t len = (offsets[r + 1] - s0) < PXG_Z_CHUNK ? (offsets[r + 1] - s0) : PXG_Z_CHUNK;
            if (at + PXG_Z_CTRL_BYTES + 2 * len > cap) return PXG_E_NOMEM;
            pxg_z_chunk& c = chunks[g];
            c.data_off = at;
            c.dst = s0;
            c.first = arena[s0];
            c.len = (int16_t)len;
            c.codec = PXG_Z_BYTES;
            uint8_t* ctrl = out + at;
            memset(ctrl, 0, PXG_Z_CTRL_BYTES);
            uint8_t* p = ctrl + PXG_Z_CTRL_BYTES;
            for (int64_t i = 1; i < len; i++) {
                const int16_t d = (int16_t)((uint16_t)arena[s0 + i] - (uint16_t)arena[s0 + i - 1]);
                const uint16_t zz = (uint16_t)(((uint16_t)d << 1) ^ (uint16_t)(d >> 15));
                *p++ = (uint8_t)zz;
                if (zz > 0xFF) {
                    ctrl[i >> 3] |= (uint8_t)(1u << (i & 7));
                    *p++ = (uint8_t)(zz >> 8);
                }
            }
            at = p - out;
        }
    }
    return at;
}

// (the host decoder trusts its records as before -- pxg_z_validate first on anything read from disk -- but a
// packed chunk's widths come from the stream itself, so it is told where the stream ends: z_bytes_or_0 = 0
// means "the caller vouches for 2 176 readable bytes behind every chunk start")
extern "C" int pxg_z_decode_n(int64_t n_chunks, const uint8_t* z, int64_t z_bytes_or_0, const pxg_z_chunk* chunks,
                              int64_t data_base, int64_t dst_base, int16_t* out)
{
    if (n_chunks < 0 || (n_chunks && (!z || !chunks || !out))) return PXG_E_INVALID;
    for (int64_t g = 0; g < n_chunks; g++) {
        const pxg_z_chunk& c = chunks[g];
        if (c.codec == PXG_Z_PACKED) {
            const uint8_t* hdr = z + (c.data_off - data_base);
            const uint8_t* end = z_bytes_or_0 ? z + z_bytes_or_0 : hdr + PXG_Z_CTRL_BYTES + 2 * PXG_Z_CHUNK;
            z_decode_packed(c, hdr, end, out + (c.dst - dst_base));
            continue;
        }
        if (c.codec != PXG_Z_BYTES) return PXG_E_INVALID;
        const uint8_t* ctrl = z + (c.data_off - data_base);
        const uint8_t* p = ctrl + PXG_Z_CTRL_BYTES;
        int16_t* dst = out + (c.dst - dst_base);
        uint16_t v = (uint16_t)c.first;
        if (c.len > 0) dst[0] = c.first;
        for (int i = 1; i < c.len; i++) {
            uint16_t zz = *p++;
            if (ctrl[i >> 3] & (1u << (i & 7))) zz |= (uint16_t)(*p++) << 8;
            const uint16_t d = (uint16_t)((zz >> 1) ^ (uint16_t)(-(int16_t)(zz & 1)));
            v = (uint16_t)(v + d);
            dst[i] = (int16_t)v;
        }
    }
    return PXG_OK;
}

extern "C" int pxg_z_decode(int64_t n_chunks, const uint8_t* z, const pxg_z_chunk* chunks, int64_t data_base,
                            int64_t dst_base, int16_t* out)
{
    return pxg_z_decode_n(n_chunks, z, 0, chunks, data_base, dst_base, out);
}

extern "C" int pxg_z_validate(int64_t n_chunks, const pxg_z_chunk* chunks, int64_t data_base, int64_t z_bytes,
                              int64_t dst_base, int64_t n_samples)
{
    return pxg_z_check(n_chunks, chunks, data_base, z_bytes, dst_base, n_samples);
}
