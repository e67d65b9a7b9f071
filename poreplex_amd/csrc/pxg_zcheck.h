// pxg_zcheck.h -- structural check of the chunk records of an encoded sample stream
// (include/pxg.h, pxg_z_chunk), shared by libpxg.so (pxg_batch_stage_z) and libpxghost.so
// (pxg_z_validate).  The records come from a file: a truncated or corrupt bundle must be an
// error, not an out-of-bounds read or write on the device.  After this check every chunk
//   * writes exactly its own samples, and the chunks tile [dst_base, dst_base + n_samples) in order;
//   * starts inside the byte stream, and the stream holds at least the bytes the chunk cannot do
//     without (control bytes + one byte per delta; a packed chunk: its width header); the decoders clamp
//     their reads to the stream,
//     so flipped control bits can only produce wrong samples, never a stray access.
#ifndef PXG_ZCHECK_H
#define PXG_ZCHECK_H
#include <stdint.h>
#include "../../include/pxg.h"

static inline int pxg_z_check(int64_t n_chunks, const pxg_z_chunk* chunks, int64_t data_base, int64_t z_bytes,
                              int64_t dst_base, int64_t n_samples)
{
    if (n_chunks < 0 || z_bytes < 0 || n_samples < 0 || (n_chunks && !chunks)) return PXG_E_INVALID;
    int64_t at = 0, byte = 0;                 // next sample, first byte the next chunk may start at
    for (int64_t g = 0; g < n_chunks; g++) {
        const pxg_z_chunk& c = chunks[g];
        if (c.len < 1 || c.len > PXG_Z_CHUNK) return PXG_E_INVALID;
        if (c.dst - dst_base != at) return PXG_E_INVALID;
        const int64_t off = c.data_off - data_base;
        if (off < byte || off > z_bytes) return PXG_E_INVALID;
        if (c.codec != PXG_Z_BYTES && c.codec != PXG_Z_PACKED) return PXG_E_INVALID;
        // (a packed chunk may be all header: every width 0)
        const int64_t least = PXG_Z_CTRL_BYTES + (c.codec == PXG_Z_BYTES ? (int64_t)(c.len - 1) : 0);
        if (z_bytes - off < least) return PXG_E_INVALID;
        byte = off + least;
        at += c.len;
    }
    return at == n_samples ? PXG_OK : PXG_E_INVALID;
}
#endif
